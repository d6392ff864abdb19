// Launch helpers shared between translation units of libsdn_hip (not part of the C ABI).
#pragma once
#include "sdn_common.h"

namespace sdn_int {

// Frame group selection handed (by value) to the loop kernels: with n_frames > 1 ray r belongs to frame r / rays_per_frame,
// marches grid[frame] with the cull grid cull + frame * cull_stride, and every emitted sample's frame goes to slot_frame.
struct FrameSel {
    uint32_t n_frames = 0;        // <= 1: a single frame, the kernels' own grid / cull arguments are used as they are
    uint32_t rays_per_frame = 0;
    uint32_t cull_stride = 0;     // uint32 words between the cull grids of consecutive frames
    uint32_t pad_ = 0;
    const uint8_t *grid[SDN_MAX_GROUP_FRAMES] = {};
    uint8_t *slot_frame = nullptr;
};
FrameSel frame_sel(const SdnRenderCtx *c);

int loop_begin(uint32_t N, uint32_t max_steps, const float *nears, int32_t *alive_a, float *rays_t, float *weights_sum, float *depth,
               float *image, int32_t *state, int32_t *live_counts, uint32_t n_counters, void *mailbox, uint32_t frame_tag, float *rays_tend,
               hipStream_t st);
int loop_cull_start(uint32_t N, const float *rays_o, const float *rays_d, const float *nears, const float *fars, float bound, float dt_gamma,
                    uint32_t C, uint32_t H, const uint32_t *cull, const FrameSel &fs, int32_t *alive_a, int32_t *alive_b, float *rays_tend,
                    int32_t *state, uint32_t *block_totals, int32_t *n_out, int32_t *trace, uint32_t max_steps, float *jump, hipStream_t st);
int loop_march(uint32_t bound_alive, const int32_t *alive_a, const int32_t *alive_b, const float *rays_t, const float *rays_o,
               const float *rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t *grid,
               const float *fars, float *xyzs, float *dirs, float *deltas, const uint32_t *cull, uint32_t *live_idx,
               uint32_t *live_counts, const int32_t *state, const FrameSel &fs, hipStream_t st, const float *jump = nullptr);
int loop_composite_compact(uint32_t bound_alive, float T_thresh, int32_t *alive_a, int32_t *alive_b, float *rays_t, const float *sigmas,
                           const float *rgbs, const float *deltas, float *weights_sum, float *depth, float *image, int32_t *state,
                           uint32_t *block_totals, int32_t *n_out, int32_t *trace, int32_t *snap, hipStream_t st, bool freeze = false);
int loop_steady_begin(uint32_t bound_alive, const int32_t *alive_a, const int32_t *alive_b, const float *rays_t, const float *rays_o,
                      const float *rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t *grid,
                      const float *fars, float *xyzs, float *dirs, float *deltas, const uint32_t *cull, uint32_t *live_idx,
                      uint32_t *live_counts, int32_t *state, const FrameSel &fs, hipStream_t st, bool frozen_already = false);
int loop_composite_march(uint32_t bound_list, float T_thresh, int32_t *alive_a, int32_t *alive_b, float *rays_t, const float *rays_o,
                         const float *rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t *grid,
                         const float *fars, const float *sigmas, const float *rgbs, float *xyzs, float *dirs, float *deltas,
                         float *weights_sum, float *depth, float *image, const uint32_t *cull, uint32_t *live_idx, uint32_t *live_counts,
                         int32_t *state, int32_t *ticket, int32_t *trace, int32_t *snap, const FrameSel &fs, hipStream_t st);
int loop_finish(uint32_t N, const float *nears, const float *fars, const float *weights_sum, const float *depth, const float *image, float bg,
                float *image_out, float *depth_out, hipStream_t st);
int render_begin(const SdnRenderCtx *c, void *mailbox, uint32_t frame_tag, hipStream_t st);
int build_cull(const uint8_t *bitfield, uint32_t *cull_bits, hipStream_t st, bool with_image = true);
int march_rays_train(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound, float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C,
                     uint32_t H, uint32_t M, const float *nears, const float *fars, float *xyzs, float *dirs, float *deltas, int32_t *rays, int32_t *counter,
                     const float *noises, void *scratch, const void *prebuilt_cull, hipStream_t st);
int build_cull_group(const FrameSel &fs, uint32_t *cull_bits, hipStream_t st);   // one cull grid per frame of the group
int copy_cull(const void *const *prebuilt, uint32_t n_frames, uint32_t *cull_bits, hipStream_t st);   // prebuilt grids -> the context's copy
int field_forward_f16(const float *xyzs, const float *dirs, const uint32_t *live_idx, const uint32_t *live_count, const int32_t *state,
                      uint32_t M, const void *weights, const float *bias0, const void *table, const int32_t *offsets_host, float S,
                      uint32_t H, float bound, float density_scale, int zero_deform, float *sigmas, float *rgbs, uint32_t expect_points,
                      const uint8_t *slot_frame, uint32_t n_frames, hipStream_t st);   // n_frames: rows of bias0 (1 without slot_frame)

int field_forward_f32(const float *xyzs, const float *dirs, const uint32_t *live_idx, const uint32_t *live_count, const int32_t *state,
                      uint32_t M, const float *weights, const float *bias0, const float *table, const int32_t *offsets_host, float S,
                      uint32_t H, float bound, float density_scale, int zero_deform, float *sigmas, float *rgbs, float *deform,
                      const uint8_t *slot_frame, uint32_t n_frames, hipStream_t st);   // zero_deform: bit f = frame f is canonical

int field_forward_f32x3(const float *xyzs, const float *dirs, const uint32_t *live_idx, const uint32_t *live_count, const int32_t *state,
                        uint32_t M, const float *weights, const float *bias0, const float *table, const int32_t *offsets_host, float S,
                        uint32_t H, float bound, float density_scale, int zero_deform, float *sigmas, float *rgbs, float *deform,
                        const uint8_t *slot_frame, uint32_t n_frames, hipStream_t st);

int field_cells_f16(const int32_t *cells, const uint32_t *cell_count, uint32_t n, const float *noise, uint32_t seed, uint32_t grid_size,
                    float cas_bound, const void *weights, const float *bias0, const void *table, const int32_t *offsets_host, float S,
                    uint32_t H, float bound, float density_scale, int zero_deform, float *tmp_slice, hipStream_t st);

}  // namespace sdn_int

// Host-side pieces of the fused-MLP operator (ffmlp.hip) that the native training step (train.hip) composes itself: packing of
// several networks in one launch, the fused forward / backward chains on already packed fragments, and any set of weight-gradient
// products  out[M,N] = G[B,ldg]^T X[B,ldx]  (fp16 operands, fp32 split-K partial sums) in one launch + one reduction.
namespace sdn_ffh {
struct PackJob { const void *weights; void *packed; uint32_t in_dim, W, L; int backward, with_last; };
struct DwJob { const void *G; uint32_t ldg, M; const void *X; uint32_t ldx, N; void *out; };
uint32_t total_frags(uint32_t in_dim, uint32_t W, uint32_t L, int backward, int with_last);
int pack_many(const PackJob *jobs, uint32_t n, hipStream_t st);
int forward_packed(const void *inputs, const void *packed, uint32_t B, uint32_t in_dim, uint32_t W, uint32_t L, uint32_t act,
                   void *forward_buffer, void *outputs, hipStream_t st);
int backward_packed(const void *grad, const void *packed, const void *forward_buffer, uint32_t B, uint32_t in_dim, uint32_t W, uint32_t L,
                    uint32_t act, int want_dx, void *backward_buffer, void *grad_inputs, hipStream_t st);
uint64_t dw_jobs_bytes(const DwJob *jobs, uint32_t n, uint32_t B);
int dw_jobs(const DwJob *jobs, uint32_t n, uint32_t B, void *partial, hipStream_t st);
}  // namespace sdn_ffh
