// Shared by the two fp32 fused field kernels (field_f32.hip: fp32 MFMAs; field_f32x3.hip: fp32 operands split into fp16 pairs): the
// argument record, the stage layout of the packed weights (identical byte sizes in both packings), the point prologue, the register
// prefetch / LDS commit of a weight stage, the one-instruction ReLU, and the host-side argument fill.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "cell_points.h"
#include "grid_common.h"
#include "sdn_common.h"
#include "sdn_internal.h"
#include "sh_eval.h"

namespace sdn_f32 {

using sdn_grid::LevelParams;
typedef float float16_t __attribute__((ext_vector_type(16)));

#ifndef SDN_F32_WAVES
#define SDN_F32_WAVES 4
#endif
constexpr int kWaves = SDN_F32_WAVES;          // waves per workgroup (8: one workgroup per CU; 4: two, out of step with each other)
constexpr int kPieces = 16384 / (64 * kWaves * 4);   // 16-byte pieces per thread of a 64-KiB stage
constexpr int kPointsPerWG = 32 * kWaves;
constexpr int kStageFloats = 16384;   // 64 KiB: one 128 x 128 layer
constexpr int kMaxFrames = 16;        // frames of a frame group (SDN_MAX_GROUP_FRAMES)

// packed weights, in stage order: D0 (32 KiB) | D1 .. D6 (64 KiB each) | tail = D7 S0 S1 C0 C1 C2 (64 KiB), as float counts
constexpr int kD0 = 0, kD0Floats = 8192;
constexpr int kD1 = kD0 + kD0Floats;
constexpr int kTail = kD1 + 6 * kStageFloats;
constexpr int kTailFloats = kStageFloats;
constexpr int kTotalFloats = kTail + kTailFloats;

struct F32Args {
    const float *xyzs, *dirs;
    const uint32_t *live_idx, *live_count;
    const int32_t *state;
    uint32_t M;
    const float *weights, *bias0, *table;
    float *sigmas, *rgbs, *deform;      // deform: optional [M,3], the deformation network's output (zeros on the canonical frame)
    float bound, density_scale;
    int zero_deform;              // bit f: frame f is the canonical frame (no deformation)
    const uint8_t *slot_frame;    // frame group: frame of every sample slot (bias0 then holds n_frames rows), or nullptr = one frame
    uint32_t n_frames;
    // density-grid query (CELLS kernels): slot = Morton cell index (live_idx = the cell list or nullptr = cells 0 .. M-1), the point
    // is the cell's jittered centre built in the kernel (cell_points.h), only sigma * density_scale is written (to sigmas[cell])
    const float *cell_noise;      // [count,3] uniform [0,1) by list position, or nullptr = counter-based generator on cell_seed
    uint32_t cell_seed;
    float cell_inv, cell_span, cell_half;
};

// ReLU in ONE instruction: as signed integers, negative floats are negative and non-negative floats keep their order, so max(bits, 0)
// is max(x, 0) (fmaxf compiles to a canonicalising v_max_f32 x, x in front of the real one; -0.0 and negative NaNs become +0.0)
__device__ __forceinline__ float relu1(float x) {
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}

// The point of this lane: lane half h = lane / 32 of point n = lane % 32 of the wave's 32 (both halves hold the same point).
struct Point {
    float x[3], d[3];
    uint32_t slot, fr, h, n, lane;
    bool valid, canonical;
};
// false: the whole workgroup lies beyond the live count (workgroup-uniform, before any barrier)
template <bool CELLS = false>
__device__ __forceinline__ bool load_point(const F32Args &P, Point &p) {
    p.lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    p.h = p.lane >> 5; p.n = p.lane & 31u;
    const uint32_t count = P.state ? P.live_count[P.state[3]] : (P.live_idx ? *P.live_count : P.M);
    if (blockIdx.x * (uint32_t)kPointsPerWG >= count) return false;
    const uint32_t i = blockIdx.x * (uint32_t)kPointsPerWG + wave * 32u + p.n;
    p.valid = i < count;
    p.slot = p.valid ? (P.live_idx ? P.live_idx[i] : i) : 0u;
    p.x[0] = p.x[1] = p.x[2] = 0; p.d[0] = p.d[1] = 0; p.d[2] = 1;
    if constexpr (CELLS) {
        if (p.valid) {
            #pragma unroll
            for (int d = 0; d < 3; d++) {
                const float r = P.cell_noise ? P.cell_noise[(size_t)i * 3 + d] : sdn_cells::cell_uniform(P.cell_seed, i * 3u + (uint32_t)d);
                p.x[d] = sdn_cells::cell_coord(p.slot, d, r, P.cell_inv, P.cell_span, P.cell_half);
            }
        }
    } else if (p.valid) {
        p.x[0] = P.xyzs[(size_t)p.slot * 3]; p.x[1] = P.xyzs[(size_t)p.slot * 3 + 1]; p.x[2] = P.xyzs[(size_t)p.slot * 3 + 2];
        p.d[0] = P.dirs[(size_t)p.slot * 3]; p.d[1] = P.dirs[(size_t)p.slot * 3 + 1]; p.d[2] = P.dirs[(size_t)p.slot * 3 + 2];
    }
    p.fr = (P.slot_frame && p.valid) ? (uint32_t)P.slot_frame[p.slot] : 0u;      // (< n_frames: written by the marcher)
    p.canonical = (P.zero_deform >> p.fr) & 1;
    return true;
}

// Weight stages: the NEXT stage's 16-byte pieces are fetched into registers before a layer's MFMAs start and written to LDS when every
// wave is through with the current stage -- the global latency runs under the layer instead of in front of it.  Seen in the ISA on the
// way: the pieces must travel BY VALUE (an array captured by reference went to scratch), the guard must be a compile-time one (a
// lane-dependent guard made every piece a predicated merge that waited for its load on the spot), and left alone the scheduler sinks the
// loads to the end of the layer, where nothing hides them.
struct Pre { float4 v[kPieces]; };
template <int FLOATS>
__device__ __forceinline__ Pre stage_prefetch(const float *src) {
    static_assert(FLOATS % (64 * kWaves * 4) == 0, "whole pieces");
    Pre r;
    #pragma unroll
    for (int q = 0; q < kPieces; q++)
        r.v[q] = (q * 64 * kWaves * 4 < FLOATS) ? *reinterpret_cast<const float4 *>(src + (q * 64 * kWaves + (int)threadIdx.x) * 4)
                                                : make_float4(0, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    return r;
}
// (a bare s_barrier behind the wave's own LDS traffic: __syncthreads() carries a fence that drains the global loads in flight -- exactly
//  the prefetch -- at the first barrier after they were issued; measured 104 -> 93 TFLOP/s with it)
__device__ __forceinline__ void wg_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}
template <int FLOATS>
__device__ __forceinline__ void stage_commit(float *s_w, const Pre &r) {
    wg_barrier();                                      // every wave has finished reading the previous stage
    #pragma unroll
    for (int q = 0; q < kPieces; q++)
        if (q * 64 * kWaves * 4 < FLOATS) *reinterpret_cast<float4 *>(s_w + (q * 64 * kWaves + (int)threadIdx.x) * 4) = r.v[q];
    wg_barrier();
}

inline int fill_args(F32Args &a, LevelParams &lp, const float *xyzs, const float *dirs, const uint32_t *live_idx, const uint32_t *live_count,
                     const int32_t *state, uint32_t M, const float *weights, const float *bias0, const float *table, const int32_t *offsets_host,
                     float S, uint32_t H, float bound, float density_scale, int zero_deform, float *sigmas, float *rgbs, float *deform,
                     const uint8_t *slot_frame, uint32_t n_frames) {
    int rc = sdn_grid::fill_levels(lp, offsets_host, 16u, S, H);
    if (rc) return rc;
    a.xyzs = xyzs; a.dirs = dirs; a.live_idx = live_idx; a.live_count = live_count; a.state = state; a.M = M;
    a.weights = weights; a.bias0 = bias0; a.table = table; a.sigmas = sigmas; a.rgbs = rgbs; a.deform = deform;
    a.bound = bound; a.density_scale = density_scale; a.zero_deform = zero_deform;
    a.slot_frame = slot_frame; a.n_frames = slot_frame ? (n_frames > (uint32_t)kMaxFrames ? (uint32_t)kMaxFrames : (n_frames ? n_frames : 1u)) : 1u;
    a.cell_noise = nullptr; a.cell_seed = 0; a.cell_inv = a.cell_span = a.cell_half = 0;
    return 0;
}

}  // namespace sdn_f32
