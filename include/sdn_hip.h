/*
 * sdn_hip.h -- C ABI of libsdn_hip.so, the MI355X (gfx950) native backend for the
 * SealD-NeRF / torch-ngp dynamic-NeRF rendering path.
 *
 * Drop-in boundary: every entry point below replaces one function that the
 * reference's pybind11 modules export to its Python autograd wrappers
 * (reference paths relative to the reference repo root):
 *
 *   raymarching/src/raymarching.h:7-17   -> sdn_near_far_from_aabb ... sdn_composite_rays
 *   gridencoder/src/gridencoder.h:12-15  -> sdn_grid_encode_forward / _backward
 *   shencoder/src/shencoder.h:9-10       -> sdn_sh_encode_forward / _backward
 *   freqencoder/src/freqencoder.h:7-10   -> sdn_freq_encode_forward / _backward
 *   ffmlp/src/ffmlp.h:8-14               -> sdn_mlp_* (fused MLP on MFMA)
 *
 * Conventions (differences from the reference are deliberate and listed):
 *   - plain device pointers + sizes, no at::Tensor; the caller owns every buffer,
 *     the library never allocates, never synchronises, keeps no global state;
 *   - every call takes the HIP stream to launch on (the reference always used the
 *     legacy default stream) as an opaque void* (hipStream_t);
 *   - returns 0 on success, a hipError_t value on a launch failure, or a negative
 *     SDN_E_* code for an argument the kernels do not support (the reference threw
 *     std::runtime_error for those, or checked nothing at all);
 *   - all float tensors are fp32 unless a `dtype` argument says otherwise
 *     (SDN_F32 / SDN_F16, the two scalar_t the reference dispatches on that its
 *     callers actually use).
 */
#ifndef SDN_HIP_H
#define SDN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDN_F32 0
#define SDN_F16 1

#define SDN_E_BADARG (-1)      /* null pointer / zero size where not allowed            */
#define SDN_E_UNSUPPORTED (-2) /* D, C, degree ... outside what the reference supports */
#define SDN_E_TIMEOUT (-3)     /* a frame driver waited 20 s for an iteration that never reported back */

/* Library / build identification ("gfx950"), for load checks. */
const char *sdn_version(void);
/* Measurement aid (no reference counterpart): one wave on `stream` that idles for wall_ticks of the 100 MHz wall clock (<= 1e8) and
 * writes {shader-clock ticks, wall ticks, first shader stamp, first wall stamp} to out4 (device, 4 x uint64): the shader clock granted
 * while other streams' kernels run. */
int sdn_debug_shader_clock(unsigned long long *out4, uint32_t wall_ticks, void *stream);

/* ---------------------------------------------------------------------------
 * raymarching  (reference: raymarching/src/raymarching.h:7-17, bindings.cpp:5-19)
 * ------------------------------------------------------------------------- */

/* raymarching.h:7  near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars)
 * rays_o, rays_d [N,3]; aabb [6]; nears, fars [N].  Miss => both FLT_MAX. */
int sdn_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb, uint32_t N,
                           float min_near, float *nears, float *fars, void *stream);

/* raymarching.h:8  sph_from_ray(rays_o, rays_d, radius, N, coords)  coords [N,2] */
int sdn_sph_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N, float *coords,
                     void *stream);

/* Caller-side helper ("next" row): the rays of a whole H x W frame from a camera-to-world pose [4,4] (row-major, device memory)
 * and pinhole intrinsics -- nerf/utils.py:54-137 `get_rays` with N = -1: pixel centres + 0.5, directions normalised.
 * rays_o, rays_d [H*W, 3]. */
int sdn_get_rays(const float *pose, float fx, float fy, float cx, float cy, uint32_t H, uint32_t W, float *rays_o,
                 float *rays_d, void *stream);

/* raymarching.h:9-10  morton3D / morton3D_invert;  coords [N,3] int32, indices [N] int32 */
int sdn_morton3D(const int32_t *coords, uint32_t N, int32_t *indices, void *stream);
int sdn_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords, void *stream);

/* raymarching.h:11  packbits(grid, N, density_thresh, bitfield)  grid [N*8] f32 -> bitfield [N] u8 */
int sdn_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield, void *stream);

/* raymarching.h:13  march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M,
 *                                    nears, fars, xyzs, dirs, deltas, rays, counter, noises)
 * xyzs, dirs [M,3], deltas [M,2] must be zero-filled by the caller; rays [N,3] int32 = (ray id,
 * point offset, point count); counter [2] int32 (points, rays) is ADDED to, as in the reference.
 * Difference: slot allocation is a deterministic prefix scan in ray order (rays[i,0] == i) instead
 * of two racing atomicAdds, so results are reproducible; per ray the samples are identical.
 * scratch: caller-provided workspace of sdn_march_rays_train_scratch_bytes(N, max_steps) bytes, 16-byte aligned (scan
 * workspace, the slice's cull grid, and the parameter t of every sample: each ray is marched ONCE -- the counting pass records
 * t, a parallel pass rebuilds positions / deltas from it -- where the reference marches every ray twice). */
uint64_t sdn_march_rays_train_scratch_bytes(uint32_t N, uint32_t max_steps);
int sdn_march_rays_train(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound,
                         float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                         const float *nears, const float *fars, float *xyzs, float *dirs, float *deltas,
                         int32_t *rays, int32_t *counter, const float *noises, void *scratch, void *stream);

/* raymarching.h:14  composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, T_thresh,
 *                                                weights_sum, depth, image) */
int sdn_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *deltas,
                                     const int32_t *rays, uint32_t M, uint32_t N, float T_thresh,
                                     float *weights_sum, float *depth, float *image, void *stream);

/* raymarching.h:15  composite_rays_train_backward(grad_weights_sum, grad_image, sigmas, rgbs, deltas,
 *                   rays, weights_sum, image, M, N, T_thresh, grad_sigmas, grad_rgbs)
 * grad_sigmas [M], grad_rgbs [M,3] zero-filled by the caller. */
int sdn_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_image,
                                      const float *sigmas, const float *rgbs, const float *deltas,
                                      const int32_t *rays, const float *weights_sum, const float *image,
                                      uint32_t M, uint32_t N, float T_thresh, float *grad_sigmas,
                                      float *grad_rgbs, void *stream);

/* Whole-ray inference compositing (no counterpart in the reference's ABI; the arithmetic is raymarching.h:17 composite_rays'):
 * samples in march_rays_train's (offset, count) layout, composited with the INFERENCE rules (transmittance = 1 - weights_sum,
 * stop when the transmittance in front of a sample is < T_thresh, t running from nears[ray]).  march_rays_train (no perturbation)
 * -> field network -> this call renders a small ray batch in one pass; image and weights_sum equal the iteration loop's
 * (dnerf/renderer.py:333-381) bit for bit, depth to fp32 rounding (the loop re-bases t at every iteration). */
int sdn_composite_whole_rays(const float *sigmas, const float *rgbs, const float *deltas, const int32_t *rays,
                             const float *nears, uint32_t M, uint32_t N, float T_thresh, float *weights_sum,
                             float *depth, float *image, void *stream);

/* raymarching.h:16  march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma,
 *                              max_steps, C, H, grid, nears, fars, xyzs, dirs, deltas, noises)
 * xyzs, dirs [>= n_alive*n_step, 3], deltas [.., 2] zero-filled by the caller. */
int sdn_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t,
                   const float *rays_o, const float *rays_d, float bound, float dt_gamma, uint32_t max_steps,
                   uint32_t C, uint32_t H, const uint8_t *grid, const float *nears, const float *fars,
                   float *xyzs, float *dirs, float *deltas, const float *noises, void *stream);

/* Extension of march_rays for the native render loop (same samples, bit for bit):
 *   M_pad      rows of xyzs/dirs/deltas; the rows [n_alive*n_step, M_pad) are cleared by the same launch
 *              (the reference wrapper memsets all three buffers before every call, raymarching.py:334-336); with no ray alive
 *              (n_alive * n_step == 0) no kernel runs and the M_pad rows are cleared by three asynchronous fills instead;
 *   cull_grid  sdn_cull_grid_bytes() bytes from sdn_build_cull_grid, or NULL: exact early-out for rays whose remaining
 *              segment stays >= 2 voxels away from every occupied voxel (they produce no sample in the reference either);
 *   live_idx / live_count (both or neither): slot indices that received a sample are appended at
 *              live_idx[atomicAdd(live_count, n)] -- unordered; the caller zeroes *live_count. */
int sdn_march_rays_ex(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t,
                      const float *rays_o, const float *rays_d, float bound, float dt_gamma, uint32_t max_steps,
                      uint32_t C, uint32_t H, const uint8_t *grid, const float *fars, float *xyzs, float *dirs,
                      float *deltas, const float *noises, uint32_t M_pad, const uint8_t *cull_grid,
                      uint32_t *live_idx, uint32_t *live_count, void *stream);
uint32_t sdn_cull_grid_bytes(void);
/* bitfield: one 128^3 Morton-ordered occupancy slice (cascade 0), 8-byte aligned.  H must be 128.
 * cull_grid: sdn_cull_grid_bytes() bytes, 16-byte aligned: 32^3 mark bits (bit (z*32+y)*32+x; cell marked iff an occupied voxel lies
 * in its 3x3x3 neighbourhood) followed by 8 ints {x0,y0,z0,x1,y1,z1,-,-}, the inclusive bounding box of the marked cells (the
 * marcher keeps the fine bits of that box in LDS when it fits 32 KiB). */
int sdn_build_cull_grid(const uint8_t *bitfield, uint32_t H, uint8_t *cull_grid, void *stream);

/* raymarching.h:17  composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas,
 *                                  weights_sum, depth, image)   -- mutates the last three, rays_alive, rays_t */
int sdn_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive, float *rays_t,
                       const float *sigmas, const float *rgbs, const float *deltas, float *weights_sum,
                       float *depth, float *image, void *stream);

/* Extension (no reference counterpart; replaces the caller-side torch mask-select
 * `rays_alive[rays_alive >= 0]`, dnerf/renderer.py:372): stable compaction of the non-negative
 * entries of in[0..n) into out, count written to *n_out (device int32).  Single launch, wave-ballot
 * scan; used by the native render loop so it needs one 4-byte read-back per iteration at most.
 * scratch: caller-provided, sdn_compact_alive_scratch_bytes(n) bytes. */
uint64_t sdn_compact_alive_scratch_bytes(uint32_t n);
int sdn_compact_alive(const int32_t *in, uint32_t n, int32_t *out, int32_t *n_out, void *scratch, void *stream);

/* ---------------------------------------------------------------------------
 * gridencoder  (reference: gridencoder/src/gridencoder.h:12-15)
 * ------------------------------------------------------------------------- */

/* gridencoder.h:12  grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx,
 *                                       gridtype, align_corners, interp)
 * inputs [B,D] f32 in [0,1]; embeddings [offsets[L], C] (dtype); offsets [L+1] int32 (device);
 * outputs [L,B,C] (dtype); dy_dx [B,L,D,C] (dtype) or NULL.  D in 2..5, C in {1,2,4,8}.
 * gridtype 0 = hash, 1 = tiled; interp 0 = linear, 1 = smoothstep.
 * offsets_host: the same L+1 offsets in host memory (the reference re-reads them on the device;
 * passing them by value keeps every per-level constant in scalar registers). */
/* The forward for D = 3, C = 2, 16 tiled levels, fp16 (no align_corners, linear interpolation: the dnerf configuration under `-O`) on the
 * QUAD copy of the table (sdn_field_build_quad_table; ref_offsets_host = the reference's level offsets): two 16-byte gathers per
 * (point, level) instead of four 8-byte ones, outputs [16, B, 2] fp16 bit-identical to sdn_grid_encode_forward's.  For inference on a
 * table that does not change between calls; gridencoder/grid.py keeps the copy per table version. */
int sdn_grid_encode_forward_quad_f16(const float *inputs, const void *quad_table, const int32_t *ref_offsets_host, void *outputs,
                                     uint32_t B, float S, uint32_t H, void *stream);
int sdn_grid_encode_forward(const float *inputs, const void *embeddings, const int32_t *offsets_host,
                            void *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                            void *dy_dx, uint32_t gridtype, int align_corners, uint32_t interp, int dtype,
                            void *stream);

/* gridencoder.h:13  grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L,
 *                                        S, H, dy_dx, grad_inputs, gridtype, align_corners, interp)
 * grad [L,B,C] (dtype); grad_embeddings like embeddings, zero-filled by the caller;
 * dy_dx / grad_inputs ([B,D], dtype) may both be NULL. */
int sdn_grid_encode_backward(const void *grad, const float *inputs, const int32_t *offsets_host,
                             void *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                             uint32_t H, const void *dy_dx, void *grad_inputs, uint32_t gridtype,
                             int align_corners, uint32_t interp, int dtype, void *stream);
/* Deterministic form (a mode for bit-exact tests): det_scratch = offsets_host[L] * C 64-bit words of device memory,
 * 8-byte aligned, cleared by the call.  Every corner contribution is added to a fixed-point accumulator (2^-24 units for the fp16 table,
 * 2^-40 for fp32) with an integer atomic -- independent of the order the hardware executes them in -- and the sums are added to
 * grad_embeddings with one rounding per element; NULL = sdn_grid_encode_backward. */
int sdn_grid_encode_backward_det(const void *grad, const float *inputs, const int32_t *offsets_host, void *grad_embeddings, uint32_t B,
                                 uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, const void *dy_dx, void *grad_inputs,
                                 uint32_t gridtype, int align_corners, uint32_t interp, int dtype, void *det_scratch, void *stream);

/* ---------------------------------------------------------------------------
 * shencoder  (reference: shencoder/src/shencoder.h:9-10)   fp32 only, D == 3, 1 <= C <= 8
 * ------------------------------------------------------------------------- */
/* shencoder.h:9   sh_encode_forward(inputs, outputs, B, D, C, dy_dx)  outputs [B,C*C]; dy_dx [B,3,C*C] or NULL */
int sdn_sh_encode_forward(const float *inputs, float *outputs, uint32_t B, uint32_t D, uint32_t C,
                          float *dy_dx, void *stream);
/* shencoder.h:10  sh_encode_backward(grad, inputs, B, D, C, dy_dx, grad_inputs)  grad_inputs [B,3] is ADDED to */
int sdn_sh_encode_backward(const float *grad, const float *inputs, uint32_t B, uint32_t D, uint32_t C,
                           const float *dy_dx, float *grad_inputs, void *stream);

/* ---------------------------------------------------------------------------
 * freqencoder  (reference: freqencoder/src/freqencoder.h:7-10)   fp32 only
 * ------------------------------------------------------------------------- */
/* freqencoder.h:7  freq_encode_forward(inputs, B, D, deg, C, outputs)   C = D + 2*D*deg */
int sdn_freq_encode_forward(const float *inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                            float *outputs, void *stream);
/* freqencoder.h:9  freq_encode_backward(grad, outputs, B, D, deg, C, grad_inputs) */
int sdn_freq_encode_backward(const float *grad, const float *outputs, uint32_t B, uint32_t D, uint32_t deg,
                             uint32_t C, float *grad_inputs, void *stream);

/* ---------------------------------------------------------------------------
 * fused field network  (reference: the op sequence of dnerf/network.py:123-169 under `-O`; the fused-MLP operator
 * shape of ffmlp/src/ffmlp.h:8-14 -- one launch, fp16 weights, activations on chip -- extended over the encoders)
 * ------------------------------------------------------------------------- */
/* Number of 1-KiB weight fragments sdn_field_forward_f16 expects (fragment order: seald-nerf_amd/dnerf_amd/fused.py). */
uint32_t sdn_field_weight_blocks(void);

/* sigma [M], rgb [M,3] of sample points xyzs/dirs [M,3] for the dnerf field network (freq(10) ++ time bias -> 8x128
 * deform MLP -> tiled grid 16x2 (fp16 table) -> 64,16 sigma MLP; SH(4) ++ geo_feat -> 64,64,3 colour MLP), with the
 * reference's autocast numerics.  live_idx/live_count (both or neither): evaluate only the listed slots; the count is read
 * on the device.  bias0 [128] f32: W0[:,63:76] . freq(t, 6).  offsets_host [17]: level row offsets.  The table may be in the
 * reference's layout (level sizes multiples of 8 rows, grid.py:124) or PADDED -- every level followed by one extra row that
 * repeats the level's row 0 (level sizes == 1 mod 8; the layout is recognised from the offsets) -- which takes the wrap
 * bookkeeping of `(index + 1) % hashmap_size` out of the gather path -- or the QUAD layout of sdn_field_build_quad_table
 * (level sizes == 2 mod 8): 16-byte blocks holding the four (x, y) corners of a cell, two gathers per level instead of four
 * (dnerf_amd/fused.py builds it; the product path). */
int sdn_field_forward_f16(const float *xyzs, const float *dirs, const uint32_t *live_idx, const uint32_t *live_count,
                          uint32_t M, const void *weights, const float *bias0, const void *table,
                          const int32_t *offsets_host, float S, uint32_t H, float bound, float density_scale,
                          int zero_deform, float *sigmas, float *rgbs, void *stream);

/* The same network in fp32 (the reference without `-O`; csrc/field_f32.hip: v_mfma_f32_32x32x2_f32, fp32 encoders): weights =
 * sdn_field_weight_floats_f32() floats in the order of dnerf_amd/fused_f32.py:pack_weights_f32, bias0 [128] = W0[:,63:76] . freq(t, 6)
 * in fp32, table = the model's fp32 embeddings [offsets_host[16], 2] in the reference's layout (read in place), offsets_host [17] the
 * reference's level offsets.  deform: NULL, or [M,3] receiving the deformation network's output (zeros when zero_deform: network.py:139-141).
 * Everything else as sdn_field_forward_f16. */
uint32_t sdn_field_weight_floats_f32(void);
int sdn_field_forward_f32(const float *xyzs, const float *dirs, const uint32_t *live_idx, const uint32_t *live_count,
                          uint32_t M, const float *weights, const float *bias0, const float *table,
                          const int32_t *offsets_host, float S, uint32_t H, float bound, float density_scale,
                          int zero_deform, float *sigmas, float *rgbs, float *deform, void *stream);

/* The same fp32 network on the fp16 matrix pipes (csrc/field_f32x3.hip): every fp32 operand split into hi + lo fp16 values, three
 * v_mfma_f32_32x32x16_f16 per product (hi.hi + hi.lo + lo.hi, fp32 accumulation; the dropped lo.lo term is 2^-22 relative) -- the
 * accuracy of the fp32 kernel at 3 / 16 of its matrix time.  weights: sdn_field_weight_floats_f32() * 4 bytes in the order of
 * dnerf_amd/fused_f32.py:pack_weights_f32_split; every other argument as sdn_field_forward_f32. */
int sdn_field_forward_f32x3(const float *xyzs, const float *dirs, const uint32_t *live_idx, const uint32_t *live_count,
                            uint32_t M, const float *weights, const float *bias0, const float *table,
                            const int32_t *offsets_host, float S, uint32_t H, float bound, float density_scale,
                            int zero_deform, float *sigmas, float *rgbs, float *deform, void *stream);

/* Which kernel large launches of the fused field network take: 1 = the persistent two-set ("ping-pong") kernel, the default for launches
 * of at least 4 tiles of 256 points per CU on a QUAD table; 0 = one tile per workgroup for every launch; -1 = environment SDN_FIELD_PP or
 * the default.  Both produce the same bits: a switch for tests and A/B measurements, not a tuning knob. */
void sdn_field_select_kernel(int persistent);
/* Workgroups of a persistent launch: 0 = one per CU (default); a driver that keeps several frames in flight passes fewer (7 / 8 of
 * the CUs) so that the other frames' small kernels find a free CU while a persistent launch runs. */
void sdn_field_persistent_workgroups(int n);

/* The fused kernel's QUAD table from embeddings in the reference's layout (gridencoder/grid.py:118-140; cast to fp16 as grid.py:43-44
 * does under autocast): embeddings [ref_offsets_host[16], 2] of `dtype` (SDN_F32 / SDN_F16), ref_offsets_host [17] the reference's level
 * offsets.  out: (ref_offsets_host[16] + 32) blocks of 16 bytes; block ref_offsets_host[l] + 2 l + r holds rows {r, r+1, r+s1, r+s1+1}
 * of level l (each mod the level's row count; s1 = the row stride of +1 in y of get_grid_index, gridencoder.cu:66-84, 0 if the
 * dimension is dropped) as fp16 pairs.  Pass offsets_host[l] = ref_offsets_host[l] + 2 l to the field entry points. */
int sdn_field_build_quad_table(const void *embeddings, int dtype, const int32_t *ref_offsets_host, float S, uint32_t H, void *out,
                               void *stream);

/* ---------------------------------------------------------------------------
 * density-grid maintenance  (reference: NeRFRenderer.update_extra_state, dnerf/renderer.py:453-555; the network queries of
 * :470-497 / :503-533, the EMA of :536-538, the mean + packbits of :539-545).  No host synchronisation in any of the three.
 * ------------------------------------------------------------------------- */
/* tmp_slice[cell] = density_scale * sigma(jittered centre of `cell`, t) for one time slice and one cascade, through the fused
 * field network (weights / bias0 / table / offsets_host / S / H / bound / zero_deform as for sdn_field_forward_f16; bias0 carries
 * the -- perturbed -- time).  cells [n] Morton indices with the live count on the device (cell_count), or both NULL = the cells
 * 0..n-1.  Point of a cell, in the reference's fp32 operation order (:480-490):
 *   (2 * coord * (1 / (grid_size - 1)) - 1) * (cas_bound - cas_bound / grid_size) + (r * 2 - 1) * cas_bound / grid_size
 * (the division by the host scalar grid_size - 1 is the reciprocal multiplication torch performs for `tensor / scalar` on the device)
 * with r = noise[list position, dim] (uniform [0,1), what torch.rand_like supplies there) or, noise == NULL, a counter-based
 * generator on `seed`.  A cell listed more than once keeps one of its values (as tmp_grid[indices] = sigmas does). */
int sdn_density_query_cells_f16(const int32_t *cells, const uint32_t *cell_count, uint32_t n, const float *noise, uint32_t seed,
                                uint32_t grid_size, float cas_bound, const void *weights, const float *bias0, const void *table,
                                const int32_t *offsets_host, float S, uint32_t H, float bound, float density_scale, int zero_deform,
                                float *tmp_slice, void *stream);
/* The same query for a model trained WITHOUT `-O`: the fp32 network of sdn_field_forward_f32 (weights = its packed floats, bias0 the
 * fp32 bias row of the -- perturbed -- time, table = the model's fp32 embeddings in the reference layout with the reference's
 * offsets), within 1e-4 of the op-by-op fp32 network of dnerf/network.py:171-206. */
int sdn_density_query_cells_f32(const int32_t *cells, const uint32_t *cell_count, uint32_t n, const float *noise, uint32_t seed,
                                uint32_t grid_size, float cas_bound, const float *weights, const float *bias0, const float *table,
                                const int32_t *offsets_host, float S, uint32_t H, float bound, float density_scale, int zero_deform,
                                float *tmp_slice, void *stream);
/* :536-538  density = max(density * decay, tmp) where density >= 0 and tmp >= 0, over n cells (n % 4 == 0, 16-byte aligned);
 * *sum (device, fp64, zeroed by the caller before the first slice) += sum of clamp(density, 0) after the update. */
int sdn_density_grid_ema(float *density_grid, const float *tmp_grid, uint64_t n, float decay, double *sum, void *stream);
/* :539-545  mean = *sum / n; threshold = min(mean, density_thresh); bitfield bit (i % 8) of byte i / 8 = density[i] > threshold
 * (raymarching.cu:268-289) for all n cells of all slices at once (n % 8 == 0).  mean_out (device [2], or NULL) = {mean, threshold}. */
int sdn_density_grid_pack(const float *density_grid, uint64_t n, const double *sum, float density_thresh, float *mean_out,
                          uint8_t *bitfield, void *stream);

/* ---------------------------------------------------------------------------
 * ffmlp: fully fused bias-free MLP on fp16  (reference: ffmlp/src/ffmlp.h:8-14, ffmlp/src/ffmlp.cu:630-894,
 * Python wrapper ffmlp/ffmlp.py:15-168).  Layout as in the reference: inputs [B, input_dim], outputs [B, 16],
 * forward_buffer / backward_buffer [num_layers, B, hidden_dim], all fp16 and point-major; weights flat fp16,
 * row-major [hidden, input] ++ (num_layers - 1) x [hidden, hidden] ++ [16, hidden]  (ffmlp.cu:631).
 * Supported (SDN_E_UNSUPPORTED otherwise, where the reference threw): hidden_dim in {16,32,64,128,256},
 * input_dim % 16 == 0 (<= 512), output_dim == 16 (the wrapper pads, ffmlp.py:120), num_layers >= 2,
 * activation 0..6 = relu, exponential, sine, sigmoid, squareplus, softplus, none (ffmlp.py:88-96),
 * output_activation == 6 (none; "not supported currently", ffmlp.py:108).  B need not be a multiple of 128.
 * `scratch`: sdn_ffmlp_scratch_bytes() bytes, 16-byte aligned, owned by the caller (packed weight fragments and the
 * split-K partial sums; replaces the reference's internal CUTLASS workspaces and allocate_splitk() side streams,
 * ffmlp.cu:711-740).
 * ------------------------------------------------------------------------- */
uint64_t sdn_ffmlp_scratch_bytes(uint32_t B, uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers);

/* ffmlp.h:8  ffmlp_forward(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation,
 * output_activation, forward_buffer, outputs): training forward, keeps every hidden post-activation. */
int sdn_ffmlp_forward(const void *inputs, const void *weights, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                      uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation,
                      void *forward_buffer, void *outputs, void *scratch, void *stream);

/* ffmlp.h:9  ffmlp_inference(...): same result, nothing kept (the reference's inference_buffer is not needed:
 * activations stay in registers). */
int sdn_ffmlp_inference(const void *inputs, const void *weights, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                        uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation,
                        void *outputs, void *scratch, void *stream);

/* ffmlp.h:11  ffmlp_backward(grad, inputs, weights, forward_buffer, B, ..., calc_grad_inputs, backward_buffer,
 * grad_inputs, grad_weights).  grad [B,16]; writes backward_buffer (pre-activation gradients, output side first),
 * grad_weights (flat, same layout as weights) and, if calc_grad_inputs, grad_inputs [B, input_dim].
 * activation 2 (sine) is rejected: its derivative needs pre-activations (the reference silently passes the gradient
 * through, utils.h:552-556). */
int sdn_ffmlp_backward(const void *grad, const void *inputs, const void *weights, const void *forward_buffer, uint32_t B,
                       uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers,
                       uint32_t activation, uint32_t output_activation, int calc_grad_inputs, void *backward_buffer,
                       void *grad_inputs, void *grad_weights, void *scratch, void *stream);

/* ---------------------------------------------------------------------------
 * device-driven inference loop for one frame  (reference: the Python loop dnerf/renderer.py:333-381)
 * ------------------------------------------------------------------------- */
/* All pointers are device pointers owned by the caller except grid_offsets (17 host ints, copied by value).
 * Buffer sizes: per-ray arrays N; sample arrays M_cap >= N + 128 rows; live_counts n_counters >= max_steps + 8;
 * state 16 ints; trace 2 * n_counters + 16 (ints 2 * n_counters .. +8 are a 4-deep ring of {alive rays entering the next iteration, iteration
 * number} snapshots for asynchronous read-back, the int after them a survivor-count scratch word); block_totals ceil(N / 256) + 1; n_out 1 int, zero on entry (ticket counter); cull_bits sdn_cull_grid_bytes(). */
/* Arguments of sdn_seal_bbox_map(_source) / sdn_seal_modify_hsv / sdn_seal_modify_rgb as one record (host memory except `tris`), for SdnRenderCtx.seal. */
typedef struct SdnSealBox {
    float bounds[24];          /* n_bounds x {lo xyz, hi xyz} */
    uint32_t n_bounds, n_tris;
    const float *tris;         /* device, [n_tris][12] */
    float test_dir[3], tinv[12], rinv[9], scale[3], center[3];
    float hsv[3];              /* colour modification of the mapped samples */
    int32_t modify_hsv;        /* 0 = leave colours alone */
    /* `rgb` / `rgbLightOffset` of the seal config (seal_utils.py:55-57, modify_rgb :761-777): tint towards a target colour, applied after
     * the hsv modification as map_color does */
    float rgb[3], rgb_light_offset;
    int32_t modify_rgb;
    /* `mapSource` (seal_utils.py:238-240,269-273): samples strictly inside source_bound {lo xyz, hi xyz} move to map_source */
    int32_t has_map_source;
    float source_bound[6], map_source[3];
    uint32_t reserved_;
    void *scratch;             /* device, 32 bytes, zeroed once by the caller: [0..15] modify_rgb's sum / count, [16..19] the mapSource flag word */
} SdnSealBox;

/* Several frames may be rendered TOGETHER by one loop ("frame group": the shards of consecutive frames of a camera path / of
 * successive time steps on one GPU of a ray-sharded job).  The rays are frame-major -- ray r belongs to frame r / rays_per_frame --
 * and everything that depends on a frame's time is selected per ray: the occupancy slice (dnerf/renderer.py:285), the
 * time-encoding bias of the first deformation layer and the t == 0 rule (dnerf/network.py:130-141).  Per-ray results do not
 * depend on which rays share a loop, so each frame of a group is bit-identical to the frame rendered alone. */
#define SDN_MAX_GROUP_FRAMES 16

/* The time-dependent constants of one frame (entry 0 / bit 0), or of each frame of a group. */
typedef struct SdnFrameTime {
    const uint8_t *bitfield[SDN_MAX_GROUP_FRAMES];   /* occupancy slice per frame */
    const float *field_bias0;                         /* device, [frames][128] */
    uint32_t zero_deform;                             /* bit f: frame f renders the canonical field (time == 0) */
    uint32_t reserved_;
    const void *cull_grid[SDN_MAX_GROUP_FRAMES];     /* optional: sdn_build_cull_grid(bitfield[f]) kept by the caller per occupancy
                                                       * slice (a slice is rendered many times between density updates); when every
                                                       * frame has one, the loop copies them instead of deriving them again */
} SdnFrameTime;

typedef struct SdnRenderCtx {
    const float *rays_o, *rays_d, *nears, *fars;
    const uint8_t *bitfield;
    void *cull_bits;
    int32_t *alive_a, *alive_b;
    float *rays_t, *weights_sum, *depth, *image;
    float *xyzs, *dirs, *deltas, *sigmas, *rgbs;
    uint32_t *live_idx;
    int32_t *live_counts, *state, *trace, *n_out;
    void *block_totals;
    const void *field_weights;
    const float *field_bias0;
    const void *grid_table;
    int32_t grid_offsets[17];
    float grid_S;
    uint32_t grid_H;
    uint32_t N, M_cap, n_counters, max_steps, C, H;
    float bound, dt_gamma, T_thresh, density_scale;
    int32_t zero_deform;
    /* optional: when aabb is non-NULL the frame drivers compute nears / fars themselves (near_far_from_aabb with this box and
     * min_near) into the caller's `nears` / `fars` buffers, on the frame's stream, before the loop starts */
    const float *aabb;
    float min_near;
    int32_t reserved_;
    /* optional [N] scratch: per-ray cache of the cull-grid scan (the parameter beyond which a ray meets no marked cell), so
     * that only a ray's first march of a frame scans; NULL = scan in every iteration.  With it (and the H = 128, C = 1, bound = 1
     * configuration) sdn_render_begin also runs the CULLED START: the scan of all N rays up front, iteration 0 on the compacted
     * list of the rays that may produce a sample, the first march from a certified later point of each ray's step lattice --
     * samples, survivors and the trace (which logs N for iteration 0) are unchanged; `sigmas` holds the per-ray start parameters
     * between sdn_render_begin and the first field launch. */
    float *rays_tend;
    /* optional SealD edit: a bounding-box seal mapper applied to every iteration's samples between the marcher and the field
     * network (sdn_seal_bbox_map) and to the colours of the mapped samples after it (sdn_seal_modify_hsv); NULL = no edit */
    const struct SdnSealBox *seal;
    uint8_t *seal_mask;        /* [M_cap] scratch, required with `seal` */
    /* optional frame group (n_group_frames > 1; N = n_group_frames * rays_per_frame): frame f marches frame_bitfield[f], its field
     * constants are field_bias0 + 128 f and bit f of zero_deform; cull_bits then holds n_group_frames cull grids and slot_frame
     * [M_cap] (scratch, required) receives the frame of every emitted sample.  With n_group_frames <= 1 `bitfield`,
     * `field_bias0` [128] and bit 0 of `zero_deform` describe the one frame. */
    uint32_t n_group_frames, rays_per_frame;
    const uint8_t *frame_bitfield[SDN_MAX_GROUP_FRAMES];
    uint8_t *slot_frame;
    /* optional prebuilt cull grids (sdn_build_cull_grid) of `bitfield` (entry 0) / of frame_bitfield[f]: copied into cull_bits at the
     * start of a frame instead of being derived again; any NULL entry among the frames in use = derive */
    const void *frame_cull[SDN_MAX_GROUP_FRAMES];
    /* 0: the fp16 fused field (`-O`: field_weights / grid_table / grid_offsets are sdn_field_forward_f16's); 1 / 2: the fp32 fused field
     * on fp32 MFMAs (sdn_field_forward_f32) / on split fp16 operands (sdn_field_forward_f32x3)
     * (the reference without `-O`): field_weights = sdn_field_forward_f32's packed floats, field_bias0 [128] the fp32 bias row,
     * grid_table the fp32 embeddings in the reference's layout, grid_offsets the reference's offsets */
    int32_t field_f32;
    int32_t reserved2_;
} SdnRenderCtx;

/* Resets per-ray state (alive = 0..N-1, rays_t = nears, accumulators = 0), the loop record and the counters, and builds
 * the cull grid of `bitfield`. */
int sdn_render_begin(const SdnRenderCtx *ctx, void *stream);
/* Enqueues one loop iteration (march -> fused field -> composite -> compact -> advance).  bound_alive: any upper bound of
 * the current number of alive rays (N is always valid; tighter bounds launch fewer idle workgroups).  The record is
 * state = {n_alive, n_step, steps done, iteration, side, N, max_steps, -}; an iteration with n_alive == 0 is a no-op. */
int sdn_render_step_f16(const SdnRenderCtx *ctx, uint32_t bound_alive, void *stream);
/* Same, recording the two given hipEvent_t (may be NULL) on `stream` immediately before / after the fused-field launch, so
 * a caller can time the dominant kernel in place (bench.py's roofline). */
int sdn_render_step_f16_ev(const SdnRenderCtx *ctx, uint32_t bound_alive, void *ev_field_begin, void *ev_field_end,
                           void *stream);
/* Which launches the `ev_field` timing events of the frame drivers bracket: 0 (default) the fused-field launch of an iteration,
 * 1 the marcher launch of the same iteration (k_march_rays* before steady mode, k_composite_march* in it) -- bench.py's
 * `roofline_secondary`.  Process-wide; set it while no frame is in flight. */
int sdn_render_time_kernel(int which);
/* Whole frame in one call (begin, iterations until no ray is alive, finish).  This is the one entry point that waits on the
 * device: after enqueuing iteration k it blocks on the (side-stream, pinned-memory) read-back of iteration k-1's survivor
 * count, which bounds the grids of iteration k+1 and ends the loop.  ev_main[4] / ev_copy[4]: hipEvent_t created by the
 * caller; host_snap: 8 ints of pinned host memory; ev_field: NULL or 2 * max_field_events timing events recorded around the
 * fused-field launches (iteration k uses 2k, 2k+1); iterations_out: step calls enqueued (incl. the trailing no-op). */
int sdn_render_frame_f16(const SdnRenderCtx *ctx, float bg_color, float *image_out, float *depth_out, void *stream,
                         void *side_stream, void **ev_main, void **ev_copy, int32_t *host_snap, void **ev_field,
                         uint32_t max_field_events, uint32_t *iterations_out);
/* A stream of n_frames frames through n_ctx (<= 8) contexts used in turn (ctxs, streams, side_streams [n_ctx]; host_snap 8 ints
 * and ev_main / ev_copy 4 events per context): the next frame starts as soon as a context is free and the alive rays of the
 * newest frame in flight are <= N / overlap_div (1 = at once), so the latency-bound parts of one frame run under the
 * throughput-bound kernels of another.  Per-frame pointers come from rays_o / rays_d / image_outs / depth_outs [n_frames]; the
 * contexts supply everything else (aabb must be set: the driver computes nears / fars per frame).  ev_field_frames: NULL, or per
 * frame a pointer (may be NULL) to 2 * max_field_events timing events for that frame's field launches.  exclusive_frames: NULL or
 * [n_frames] flags; a flagged frame runs with nothing else in flight (to time its kernels undisturbed).  frame_times: NULL (every
 * frame uses its context's time constants) or [n_frames] records: frame i (a frame group when the contexts are group contexts)
 * is rendered at ITS time -- a D-NeRF test set carries one time per frame (dnerf/utils.py:151-161).  Every frame is
 * bit-identical to sdn_render_frame_f16's; iterations_out [n_frames] or NULL.  done_events: NULL or [n_frames] hipEvent_t: event i is
 * recorded on frame i's stream behind its last kernel, and only THEN iterations_out[i] becomes non-zero (release store) -- a second
 * host thread can poll iterations_out and hand finished frames on (e.g. to the per-frame all-gather) while later frames still
 * render.  Returns SDN_E_TIMEOUT when no iteration of any
 * frame in flight completes within 20 s.  Every OTHER error return synchronises all streams first; the time-out return does NOT
 * (a kernel that never reports back is hung: waiting for it would hang the caller too) -- after SDN_E_TIMEOUT the device still owns
 * every buffer of the contexts and of the frames in flight: do not reuse or free them, end the process. */
int sdn_render_frames_pipelined_f16(const SdnRenderCtx *const *ctxs, uint32_t n_ctx, uint32_t n_frames, const float *const *rays_o,
                                    const float *const *rays_d, float *const *image_outs, float *const *depth_outs, float bg_color,
                                    uint32_t overlap_div, void *const *streams, void *const *side_streams, void **ev_main,
                                    void **ev_copy, int32_t *host_snap, void *const *ev_field_frames, uint32_t max_field_events,
                                    const uint8_t *exclusive_frames, const SdnFrameTime *frame_times, void *const *done_events,
                                    uint32_t *iterations_out);
/* ---------------------------------------------------------------------------
 * SealD-NeRF bounding-box seal mapper on the sample stream  (reference: SealNeRF/seal_utils.py:132-153 map_mask, :245-286
 * SealBBoxMapper.map_to_origin, :638-693 moller_trumbore / points_in_mesh, :747-758 modify_hsv; torch boolean-mask code there)
 * ------------------------------------------------------------------------- */
/* In place on xyzs / dirs [M,3] (device): a sample that is non-zero in every coordinate, strictly inside one of the n_bounds (<= 4)
 * axis-aligned bounds [n_bounds][lo xyz, hi xyz] and inside the mesh (hit by the ray along test_dir AND the opposite ray) is
 * taken back to its origin -- tinv [3x4] applied to (p, 1), then (.. - center) * scale + center -- and its direction rotated by
 * rinv [3x3]; mask [M] u8 receives 1 for mapped samples.  tris [n_tris][12] (device) = v0, E1, E2, N = E1 x E2 per triangle;
 * bounds, test_dir, tinv, rinv, scale, center are host arrays (copied by value). */
int sdn_seal_bbox_map(float *xyzs, float *dirs, uint32_t M, const float *bounds, uint32_t n_bounds, const float *tris,
                      uint32_t n_tris, const float *test_dir, const float *tinv, const float *rinv, const float *scale,
                      const float *center, uint8_t *mask, void *stream);
/* The same with the seal config's `mapSource` option (seal_utils.py:238-240,269-273): in a call that maps at least one sample -- the
 * reference returns early otherwise, :251-252 -- every unmapped sample strictly inside source_bound {lo xyz, hi xyz} (host) moves to
 * map_source [3] (host).  flag: one device word owned by the caller, zeroed once (the kernels only raise it to a per-call tag).
 * "The call's samples" are the M slots, or -- live_idx / live_count (/ state: the count is live_count[state[3]]) given -- the listed
 * slots: the device-driven loop's sample buffers may hold stale slots of earlier iterations beyond the live ones. */
int sdn_seal_bbox_map_source(float *xyzs, float *dirs, uint32_t M, const float *bounds, uint32_t n_bounds, const float *tris,
                             uint32_t n_tris, const float *test_dir, const float *tinv, const float *rinv, const float *scale,
                             const float *center, const float *source_bound, const float *map_source, uint32_t *flag, uint8_t *mask,
                             const uint32_t *live_idx, const uint32_t *live_count, const int32_t *state, void *stream);
/* rgbs [M,3] of the masked samples: rgb -> hsv, + (dh, ds, dv), -> rgb (color_utils.py:31-63), in place. */
int sdn_seal_modify_hsv(float *rgbs, const uint8_t *mask, uint32_t M, float dh, float ds, float dv, void *stream);
/* modify_rgb (seal_utils.py:761-777) on the masked samples, in place: hue and saturation of the target colour (r, g, b), brightness
 * V(target) + (V(sample) - mean V of the masked samples of THIS call) + light_offset clamped to [0, 1].  The mean is summed in fixed
 * point: the same for any order of the samples.  scratch16: 16 bytes of device memory, 8-byte aligned (cleared by the call). */
int sdn_seal_modify_rgb(float *rgbs, const uint8_t *mask, uint32_t M, float r, float g, float b, float light_offset, void *scratch16,
                        const uint32_t *live_idx, const uint32_t *live_count, const int32_t *state, void *stream);

/* Read-back memory for the frame drivers: 32 bytes per ray group of coherent, device-mapped host memory.  When `host_snap`
 * comes from here the loop kernels publish every iteration's survivor count into it with one 64-bit system-scope store and
 * the driver polls it (no event record / stream wait / copy per iteration); any other pinned memory selects the event +
 * copy path described above.  Returns NULL on failure. */
void *sdn_host_mailbox_alloc(uint32_t groups);
int sdn_host_mailbox_free(void *mailbox);
/* image_out [N,3] = image + (1 - weights_sum) * bg; depth_out [N] = clamp(depth - nears, 0) / (fars - nears). */
int sdn_render_finish(const SdnRenderCtx *ctx, float bg_color, float *image_out, float *depth_out, void *stream);

/* ---------------------------------------------------------------------------
 * native training step of the dynamic field  ("next" row of the hot path: the caller that trains what the render path evaluates)
 * reference: dnerf/utils.py:38-125 `train_step` (MSE criterion, main_dnerf.py:103) -> dnerf/renderer.py:260-331 `run_cuda`,
 * training branch (near_far_from_aabb, march_rays_train, network forward, composite_rays_train, background mix) ->
 * dnerf/network.py:123-169 under `-O` (fp16 autocast; nerf/utils.py:880-893 GradScaler) -> torch.optim.Adam(betas (0.9, 0.99),
 * eps 1e-15) as main_dnerf.py:118 builds it -> optionally torch_ema's shadow update (nerf/utils.py:906).
 * ONE call = one optimizer step in ~35 launches: nothing goes back to the host, no autograd graph, no per-step allocation.
 * The arithmetic is the reference's op sequence with its dtypes (fp16 operands / fp32 accumulation in the MLPs, one fp16
 * rounding per layer; fp16 table and table gradient; fp32 encoders, compositing, loss and Adam); results agree with the
 * op-by-op path within fp16 accumulation-order tolerance (tests/test_gpu_train_native.py), not bit for bit.
 * ------------------------------------------------------------------------- */
#define SDN_TRAIN_N_PARAMS 14   /* encoder.embeddings, deform_net.0..7, sigma_net.0..1, color_net.0..2 (dnerf/network.py:260-275) */

typedef struct SdnTrainParam {
    float *param;                 /* fp32 master, the parameter's own storage, [n] */
    float *exp_avg, *exp_avg_sq;  /* Adam moments, [n] */
    float *ema;                   /* torch_ema shadow [n], or NULL */
    uint64_t n;
} SdnTrainParam;

typedef struct SdnTrainStep {
    /* batch (device memory) */
    const float *rays_o, *rays_d;   /* [N,3] */
    const float *target;            /* [N,3] ground-truth colours, already mixed with the background (utils.py:76-79) */
    const float *bg_color;          /* [N,3] per-ray background (utils.py:74), or NULL: bg_value on every channel */
    float bg_value;
    uint32_t N, M;                  /* rays; sample budget = mean_count rounded as raymarching.py:200-203 */
    /* scene */
    const uint8_t *bitfield;        /* occupancy slice of `time` (dnerf/renderer.py:285) */
    const void *cull_grid;          /* optional: sdn_build_cull_grid(bitfield) kept by the caller (a slice is marched many times between two
                                     * density-grid updates); NULL = built inside every step */
    const float *aabb;              /* device, [6] (aabb_train) */
    float bound, min_near, dt_gamma, density_scale, T_thresh;
    float time;                     /* frame time by value; time == 0 is the canonical frame: no deformation, and the
                                     * deformation MLP is left out of the optimizer step (its gradient is None in the reference,
                                     * dnerf/network.py:140) */
    uint32_t cascade, grid_size, max_steps;
    int32_t perturb;                /* 1: per-ray start offsets from a counter-based generator (the reference draws torch.rand) */
    uint64_t noise_seed;
    const float *noises;            /* optional [N]: per-ray offsets in [0,1) to use instead of the generator (replays a torch.rand draw) */
    int32_t *counter;               /* [2] (samples, rays) of this step: the step-counter slot of dnerf/renderer.py:296-298 */
    /* grid encoder geometry (gridencoder/grid.py:96-141) */
    int32_t grid_offsets[17];
    float grid_S;
    uint32_t grid_H;
    /* parameters, in the order of SDN_TRAIN_N_PARAMS */
    SdnTrainParam params[SDN_TRAIN_N_PARAMS];
    /* optimizer */
    double lr_table, lr_net;        /* network.py:260-275: encoder tables at lr, MLPs at lr_net (doubles: torch derives the step
                                     * size and 1 - beta on the host in double precision) */
    double beta1, beta2, eps;
    float *adam_steps;              /* device [2]: optimizer steps taken by {everything but the deformation MLP, the deformation MLP} */
    float *loss_scale;              /* device [1]: GradScaler's scale */
    int32_t *growth_tracker;        /* device [1] */
    float growth_factor, backoff_factor;
    uint32_t growth_interval;
    float ema_decay;                /* this step's effective decay (min(decay, (1 + n) / (10 + n))), for parameters with `ema` */
    /* results */
    float *loss_out;                /* device [1]: the unscaled loss of this step */
    float *image_out;               /* optional [N,3]: predicted colours */
    void *workspace;                /* sdn_train_layout().total_bytes, 256-byte aligned; persistent (holds the fp16 copies of the
                                     * parameters that the kernels read): call sdn_train_refresh once before the first step and
                                     * after every outside change of the parameters */
    int32_t mode;                   /* 0: full step; 1: forward + backward only (gradients stay in the workspace); 2: optimizer only, on
                                     * the gradients in the workspace.  Data-parallel training is 1 -> all-reduce of the fp16 gradient
                                     * buffers (g_table, g_deform .. g_color) -> 2 with grad_divisor = world size */
    int32_t keep_deform;            /* modes 1 / 2 under data parallelism: the deformation MLP takes part in the optimizer step whatever
                                     * `time` is (another rank's batch may carry its gradient); mode 1 then clears its gradient at time == 0 */
    float grad_divisor;             /* gradients are divided by this on top of the loss scale (0 is read as 1) */
    int32_t deform_frozen;          /* 1: the deformation MLP is evaluated but not trained (SealD-NeRF's edit training,
                                     * SealDNeRF/utils.py:692-694): no gradient through it, left out of the optimizer step */
    int32_t phase;                  /* 0: rays -> samples AND the step on them; 1: rays -> samples only (into sample set `sample_set`);
                                     * 2: the step on the samples already in `sample_set`.  Marching needs nothing of the network, so a
                                     * caller that knows the next batch early runs phase 1 for batch k+1 on a second stream beside
                                     * phase 2 of batch k (the optimizer pass and the marcher do not compete for the same unit) */
    int32_t sample_set;             /* 0 / 1: which of the workspace's two sample buffers */
    /* Optional overlap of the optimizer's pass over the embedding table (12.2 M entries, 30 B each: 85 % of the pass, HBM-bound) with
     * the NEXT step's deformation-MLP forward, which does not touch the table: with table_stream set (a second hipStream_t), that
     * part of the pass is enqueued there behind `table_ready` (hipEvent_t, recorded on `stream` after the optimizer prologue) and
     * `table_done` (hipEvent_t) is recorded behind it; the next sdn_train_step_f16 / sdn_train_refresh makes `stream` wait for
     * table_done before its first access to the table.  A caller that touches the table, its moments or its EMA shadow itself must
     * wait for table_done first (hipStreamWaitEvent / sdn_train_flush).  All three NULL: everything on `stream`, as before. */
    void *table_stream, *table_ready, *table_done;
    /* optional deterministic mode: grid_offsets[16] * 2 64-bit words of device memory for the table gradient's order-independent
     * accumulation (sdn_grid_encode_backward_det); NULL = the plain half atomics */
    void *det_scratch;
} SdnTrainStep;

/* Byte offsets into the workspace of what a caller or a test may want to look at.  fp16 "flat" networks are laid out as the fused
 * MLP operator wants them (ffmlp.cu:631): [hidden, in16] ++ (L-1) x [hidden, hidden] ++ [16, hidden], zero padding included. */
typedef struct SdnTrainLayout {
    uint64_t total_bytes;
    uint64_t w_table, w_deform, w_sigma0, w_sigma1, w_color;   /* fp16 copies of the parameters */
    uint64_t g_table, g_deform, g_sigma0, g_sigma1, g_color;   /* fp16 gradients (scaled by the loss scale), same layouts */
    uint64_t xyzs, dirs, deltas, rays;                         /* march_rays_train outputs: [M,3] [M,3] [M,2] f32, [N,3] i32 (sample set 0) */
    uint64_t sample_set_stride;                                /* sample set 1 = the same offsets + this */
    uint64_t sigmas;                                           /* [M] f32 (density_scale applied) */
    uint64_t weights_sum, depth, image;                        /* [N] [N] [N,3] f32, before the background mix */
    uint64_t found_inf;                                        /* f32 [1] */
} SdnTrainLayout;

int sdn_train_layout(uint32_t N, uint32_t M, uint32_t max_steps, const int32_t *grid_offsets, SdnTrainLayout *out);
/* fp16 copies of all parameters from the fp32 masters; clears the gradient accumulator of the table. */
int sdn_train_refresh(const SdnTrainStep *s, void *stream);
int sdn_train_step_f16(const SdnTrainStep *s, void *stream);
/* Makes `stream` wait for the table pass a previous step left on table_stream (no-op without one). */
int sdn_train_flush(const SdnTrainStep *s, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SDN_HIP_H */
