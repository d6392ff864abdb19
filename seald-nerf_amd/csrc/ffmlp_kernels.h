// Fused-MLP kernel templates of libsdn_hip (gfx950), shared by ffmlp.hip (ReLU instantiation, C ABI) and ffmlp_act.hip
// (run-time-dispatched activations).  See ffmlp.hip for the design notes.
#pragma once
#include <math.h>

#include "sdn_common.h"

namespace sdn_ff {


typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kStageFrags = 32;                 // 32 KiB per LDS buffer
constexpr int kStageBytes = kStageFrags * 1024;
constexpr float kAct = 10.0f;                   // utils.h:41 K_ACT

enum { ACT_RELU = 0, ACT_EXP = 1, ACT_SINE = 2, ACT_SIGMOID = 3, ACT_SQUAREPLUS = 4, ACT_SOFTPLUS = 5, ACT_NONE = 6 };

__device__ __forceinline__ f32x16 mfma(half8 a, half8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// Rows of activations live one per lane PAIR: after a 32x32 MFMA chain lane n (h = 0) holds features 8q .. 8q+3 and lane n + 32
// (h = 1) features 8q+4 .. 8q+7 of every 8-feature group q.  Stored as they are that is one 8-byte piece per lane per group: 64
// different cache lines per store instruction, and the address pipe takes ~55 cycles for each (measured: 7 000 cycles per 128-wide
// layer per workgroup, 3.5x the layer's MFMA time).  v_permlane32_swap exchanges the upper half-wave of one register with the
// lower half-wave of another: for a pair of groups (q, q+1) the lower lanes end up with the 8 contiguous features of group q and
// the upper lanes with those of group q+1 -- ONE 16-byte access per lane per pair, half the instructions, twice the run length.
// The same exchange, applied to a 16-byte load, puts the features back where the MFMA layout wants them (it is an involution).
// Must run with the whole wave active.
__device__ __forceinline__ void half_wave_exchange(uint32_t &a, uint32_t &b) {
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}
struct PairRuns { half4 g0, g1; };   // this lane's 4-feature runs of groups q and q+1
__device__ __forceinline__ uint4 pair_to_row16(half4 g0, half4 g1) {
    uint2 a = __builtin_bit_cast(uint2, g0), b = __builtin_bit_cast(uint2, g1);
    half_wave_exchange(a.x, b.x);
    half_wave_exchange(a.y, b.y);
    return uint4{a.x, a.y, b.x, b.y};
}
__device__ __forceinline__ PairRuns row16_to_pair(uint4 v) {
    uint2 a{v.x, v.y}, b{v.z, v.w};
    half_wave_exchange(a.x, b.x);
    half_wave_exchange(a.y, b.y);
    return PairRuns{__builtin_bit_cast(half4, a), __builtin_bit_cast(half4, b)};
}

// Activations act on one accumulator tile's 16 values per lane at a time, with the dispatch outside the element loops
// (ARELU = true: compiled for ReLU only -- the hot case; false: run-time dispatch over the other six).
// utils.h:425-470 (forward) on the fp16-rounded layer output.  The transcendental ones use the hardware
// v_exp_f32 / v_log_f32 / v_sin_f32 forms (relative error ~1e-6, far inside the fp16 rounding of the result;
// sine is accurate to ~1e-5 absolute for |x| < 100 and degrades with |x| like any fp32 argument reduction by 2 pi).
template <bool ARELU>
__device__ __forceinline__ void act_forward16(uint32_t act, _Float16 (&v)[16]) {
    if (ARELU) {
        #pragma unroll
        for (int i = 0; i < 16; i++) v[i] = v[i] > (_Float16)0 ? v[i] : (_Float16)0;
        return;
    }
    switch (act) {
        case ACT_RELU:
            #pragma unroll
            for (int i = 0; i < 16; i++) v[i] = v[i] > (_Float16)0 ? v[i] : (_Float16)0;
            break;
        case ACT_EXP:
            #pragma unroll
            for (int i = 0; i < 16; i++) v[i] = (_Float16)__expf((float)v[i]);
            break;
        case ACT_SINE:
            #pragma unroll
            for (int i = 0; i < 16; i++) v[i] = (_Float16)__sinf((float)v[i]);
            break;
        case ACT_SIGMOID:
            #pragma unroll
            for (int i = 0; i < 16; i++) v[i] = (_Float16)__fdividef(1.0f, 1.0f + __expf(-(float)v[i]));
            break;
        case ACT_SQUAREPLUS:
            #pragma unroll
            for (int i = 0; i < 16; i++) { const float x = (float)v[i] * kAct; v[i] = (_Float16)(0.5f * (x + __fsqrt_rn(x * x + 4.0f)) / kAct); }
            break;
        case ACT_SOFTPLUS:
            #pragma unroll
            for (int i = 0; i < 16; i++) v[i] = (_Float16)(__logf(__expf((float)v[i] * kAct) + 1.0f) / kAct);
            break;
        default: break;
    }
}

// utils.h:538-583 (backward from the stored post-activation value `f`), half arithmetic for the product as there
template <bool ARELU>
__device__ __forceinline__ void act_backward16(uint32_t act, _Float16 (&g)[16], const _Float16 (&f)[16]) {
    if (ARELU) {
        #pragma unroll
        for (int i = 0; i < 16; i++) g[i] = f[i] > (_Float16)0 ? g[i] : (_Float16)0;
        return;
    }
    switch (act) {
        case ACT_RELU:
            #pragma unroll
            for (int i = 0; i < 16; i++) g[i] = f[i] > (_Float16)0 ? g[i] : (_Float16)0;
            break;
        case ACT_EXP:
            #pragma unroll
            for (int i = 0; i < 16; i++) g[i] = g[i] * f[i];
            break;
        case ACT_SIGMOID:
            #pragma unroll
            for (int i = 0; i < 16; i++) g[i] = g[i] * (_Float16)(f[i] * ((_Float16)1.0f - f[i]));
            break;
        case ACT_SQUAREPLUS:
            #pragma unroll
            for (int i = 0; i < 16; i++) { const float y = (float)f[i] * kAct; g[i] = g[i] * (_Float16)__fdividef(y * y, y * y + 1.0f); }
            break;
        case ACT_SOFTPLUS:
            #pragma unroll
            for (int i = 0; i < 16; i++) g[i] = g[i] * (_Float16)(1.0f - __expf(-(float)f[i] * kAct));
            break;
        default: break;   // none; sine is rejected on the host (needs pre-activations the buffers do not hold)
    }
}

__device__ __forceinline__ void stage_load(const unsigned char *__restrict__ g, unsigned char *lds, int nfrags, uint32_t wave, uint32_t lane, int nwaves) {
    for (int c = (int)wave; c < nfrags; c += nwaves) {
        const uint32_t off = __builtin_amdgcn_readfirstlane((uint32_t)c * 1024u);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + off + lane * 16),
                                         (__attribute__((address_space(3))) void *)(lds + off), 16, 0, 0);
    }
}

// direct-to-LDS loads are pending LDS writes on the VM counter; hipcc does not reliably wait for them before a barrier
__device__ __forceinline__ void stage_wait_and_sync() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// Counted form: everything but the wave's N youngest vector-memory operations is done (MI355X_MICROARCH: loads, stores and LDS-DMA
// count together in issue order).  The N youngest are the activation stores a wave issued AFTER the stage's direct-to-LDS loads:
// they need not be waited for -- a plain vmcnt(0) exposes their write latency in front of every stage barrier.  Raw barrier:
// `__syncthreads()` adds its own vmcnt(0).  Only valid when the wave really issued >= N such operations (see `counted` below).
template <int N>
__device__ __forceinline__ void stage_wait_counted_and_sync() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// Weight fragments reach the MFMAs through LDS in one of two ways:
//   streamed (RES = false): <= 32 KiB stages, double-buffered, one barrier per stage, re-fetched (from L2) for every point tile;
//   resident (RES = true): the whole packed network fits the LDS allocation, is loaded once per workgroup, and the workgroup
//   then loops over point tiles with no further barrier -- a stage is just an offset.
template <bool RES, int WV>
struct Stager {
    const unsigned char *g;   // streamed: next fragment to prefetch
    unsigned char *lds;
    int cur;                  // streamed: buffer of the stage being consumed
    uint32_t off, next_off;   // resident: byte offset of the current / following stage
    uint32_t wave, lane;
    __device__ __forceinline__ void prefetch(int nfrags) {
        stage_load(g, lds + (cur ^ 1) * kStageBytes, nfrags, wave, lane, WV);
        g += (size_t)nfrags * 1024;
    }
    // the stage (this_frags) prefetched last becomes current; start fetching the one after it (next_frags, 0 = none).
    // N / counted: the wave issued >= N vector-memory operations after the pending stage's loads that may stay in flight.
    // between(): issued after the barrier and before the next stage's loads (global loads that must be OLDER than those).
    template <int N, typename Fn>
    __device__ __forceinline__ void begin_stage(int this_frags, int next_frags, bool counted, Fn between) {
        if (RES) {
            off = next_off;
            next_off += (uint32_t)this_frags * 1024u;
            between();
        } else {
            if (N > 0 && N < 64 && counted) stage_wait_counted_and_sync<(N > 0 && N < 64) ? N : 0>(); else stage_wait_and_sync();
            cur ^= 1;
            between();
            if (next_frags) prefetch(next_frags);
        }
    }
    __device__ __forceinline__ void begin_stage(int this_frags, int next_frags) { begin_stage<0>(this_frags, next_frags, false, [] {}); }
    __device__ __forceinline__ const unsigned char *buf() const { return RES ? lds + off : lds + cur * kStageBytes; }
};

__device__ __forceinline__ half8 lds_frag(const unsigned char *buf, int blk, uint32_t lane) {
    return *reinterpret_cast<const half8 *>(buf + (size_t)blk * 1024 + lane * 16);
}

__host__ __device__ inline int tiles_per_stage(int KS) { return KS >= kStageFrags ? 1 : kStageFrags / KS; }
__host__ __device__ inline int first_stage_frags(int MT, int KS) { const int g = tiles_per_stage(KS); return (MT < g ? MT : g) * KS; }

struct FfArgs {
    const _Float16 *x;            // forward: inputs [B, K0]; backward: grad [B, 16]
    const unsigned char *packed;  // fragments in consumption order
    _Float16 *buf;                // forward (training): forward_buffer [L, B, W]; backward: backward_buffer [L, B, W]
    const _Float16 *fwd;          // backward: forward_buffer
    _Float16 *out;                // forward: outputs [B, 16]; backward: grad_inputs [B, Mlast] or nullptr
    uint32_t B, K0, Mlast, L, act;
    uint32_t total_frags;         // of the packed network
};

// MODE 0 inference, 1 training forward (stores the post-activations), 2 backward
template <int WIDTH, int NT, int MODE, bool RES, int OCC, int kWaves, bool ARELU>
__global__ void __launch_bounds__(64 * kWaves, OCC) k_ffmlp(FfArgs P) {
    constexpr int KS = WIDTH / 16;               // k-steps of a hidden activation
    constexpr int MT = (WIDTH + 31) / 32;        // M-tiles of a hidden layer
    constexpr int GM = KS >= kStageFrags ? 1 : (kStageFrags / KS < MT ? kStageFrags / KS : MT);   // hidden M-tiles per stage
    extern __shared__ __attribute__((aligned(16))) unsigned char s_w[];   // RES: total_frags KiB; else 2 x kStageBytes

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t n = lane & 31u, h = lane >> 5;
    Stager<RES, kWaves> S{P.packed, s_w, 1, 0, 0, wave, lane};
    if (RES) {
        stage_load(P.packed, s_w, (int)P.total_frags, wave, lane, kWaves);
        stage_wait_and_sync();
    }
    const uint32_t ntiles = (P.B + (uint32_t)(kWaves * 32 * NT) - 1u) / (uint32_t)(kWaves * 32 * NT);
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const uint32_t p0 = (tile * kWaves + wave) * (32u * NT) + n;   // point of n-tile 0
    uint32_t pt[NT];
    bool live[NT];
    #pragma unroll
    for (int t = 0; t < NT; t++) { live[t] = p0 + 32u * t < P.B; pt[t] = live[t] ? p0 + 32u * t : P.B - 1; }

    const int KS0 = (int)P.K0 / 16;
    const int MTL = ((int)P.Mlast + 31) / 32;
    const bool has_last = P.out != nullptr;

    if (RES) {
        S.next_off = 0;
    } else {
        S.g = P.packed; S.cur = 1;
        if (tile != blockIdx.x) __syncthreads();   // (streamed kernels are launched one tile per workgroup; kept for safety)
        S.prefetch(first_stage_frags(MT, KS0));     // stage 0 -> buffer 0
    }

    half8 fa[KS][NT], fb[KS][NT];
    constexpr int NQ = WIDTH >= 32 ? 4 : 2;                              // 4-feature runs of a tile that exist (hidden 16: rows 0..15 only)
    // The latency variant (OCC <= 2: twice the registers; picked by the host when a launch has no more workgroups than CUs, where
    // occupancy buys nothing and every exposed latency is on the critical path of the whole launch):
    //   * a whole M-tile of weight fragments is read from LDS while the previous M-tile's MFMAs run (the throughput variant reads
    //     two fragments, waits, issues two MFMAs: the matrix pipe idles for an LDS round trip per pair);
    //   * backward: the forward activations a layer's epilogues test are fetched at the START of the layer (all M-tiles at once,
    //     older than the next stage's loads), not inside each epilogue (an exposed L2 / HBM round trip per M-tile per layer).
    constexpr bool LA = OCC <= 2 && WIDTH <= 128 && !RES;
    constexpr bool kPrefetchFwd = LA && MODE == 2;
    uint4 fwraw[kPrefetchFwd ? MT : 1][NT][2];
    auto fetch_fwd = [&](int k) __attribute__((always_inline)) {
        if constexpr (kPrefetchFwd) {
            #pragma unroll
            for (int Mt = 0; Mt < MT; Mt++)
                #pragma unroll
                for (int t = 0; t < NT; t++)
                    #pragma unroll
                    for (int p = 0; p < NQ / 2; p++)
                        fwraw[Mt][t][p] = *reinterpret_cast<const uint4 *>(P.fwd + ((size_t)(P.L - 1 - k) * P.B + pt[t]) * WIDTH + 32u * Mt + 16u * p + 8u * h);
        }
    };
    // activation stores per M-tile epilogue, and whether this wave issues all of them (a wave without live points may skip them:
    // then nothing may be assumed about its vector-memory queue and the stage waits drain it completely)
    constexpr int kStoresPerTile = MODE != 0 ? (NQ / 2) * NT : 0;
    bool wave_full = MODE != 0;
    #pragma unroll
    for (int t = 0; t < NT; t++) wave_full = wave_full && __builtin_amdgcn_ballot_w64(live[t]) != 0;

    // hidden-layer epilogue: accumulator tile -> next layer's B fragments (+ buffers)
    auto epilogue = [&](int k, int Mt, const f32x16 (&acc)[NT], half8 (&dst)[KS][NT]) __attribute__((always_inline)) {
        #pragma unroll
        for (int t = 0; t < NT; t++) {
            _Float16 v[16];
            #pragma unroll
            for (int i = 0; i < 16; i++) v[i] = (_Float16)acc[t][i];
            if (MODE == 2) {
                _Float16 fw[16];
                #pragma unroll
                for (int p = 0; p < 2; p++) {
                    PairRuns x{{0, 0, 0, 0}, {0, 0, 0, 0}};
                    if (p < NQ / 2) {
                        uint4 raw;
                        if constexpr (kPrefetchFwd) raw = fwraw[Mt][t][p];
                        else raw = *reinterpret_cast<const uint4 *>(P.fwd + ((size_t)(P.L - 1 - k) * P.B + pt[t]) * WIDTH + 32u * Mt + 16u * p + 8u * h);
                        x = row16_to_pair(raw);
                    }
                    #pragma unroll
                    for (int e = 0; e < 4; e++) { fw[8 * p + e] = x.g0[e]; fw[8 * p + 4 + e] = x.g1[e]; }
                }
                act_backward16<ARELU>(P.act, v, fw);
            } else {
                act_forward16<ARELU>(P.act, v);
            }
            if constexpr (MODE != 0) {
                #pragma unroll
                for (int p = 0; p < NQ / 2; p++) {
                    const uint4 row16 = pair_to_row16(half4{v[8 * p], v[8 * p + 1], v[8 * p + 2], v[8 * p + 3]}, half4{v[8 * p + 4], v[8 * p + 5], v[8 * p + 6], v[8 * p + 7]});
                    if (live[t]) *reinterpret_cast<uint4 *>(P.buf + ((size_t)k * P.B + pt[t]) * WIDTH + 32u * Mt + 16u * p + 8u * h) = row16;
                }
            }
            #pragma unroll
            for (int s = 0; s < 2; s++)
                if (2 * Mt + s < KS) dst[2 * Mt + s][t] = half8{v[8 * s], v[8 * s + 1], v[8 * s + 2], v[8 * s + 3], v[8 * s + 4], v[8 * s + 5], v[8 * s + 6], v[8 * s + 7]};
        }
    };

    // ---- first layer: B operand from global memory (row length K0), M = WIDTH -------------------------------------------------
    {
        const int g0 = tiles_per_stage(KS0);
        const int after = first_stage_frags(P.L > 1 ? MT : MTL, KS);       // first stage of the layer that follows
        const bool preload = KS0 <= KS;
        if (preload) {
            #pragma unroll
            for (int s = 0; s < KS; s++) {
                const int ss = s < KS0 ? s : KS0 - 1;      // k-steps beyond K0 re-read the last one (never multiplied): keeps fb in registers
                #pragma unroll
                for (int t = 0; t < NT; t++) {
                    half4 lo, hi;
                    if constexpr (MODE != 0) {      // (the inference instantiation sits at its register limit: it keeps the 8-byte loads)
                        const PairRuns r = row16_to_pair(*reinterpret_cast<const uint4 *>(P.x + (size_t)pt[t] * P.K0 + 16 * ss + 8 * h));
                        lo = r.g0; hi = r.g1;
                    } else {
                        const _Float16 *row = P.x + (size_t)pt[t] * P.K0 + 16 * ss + 4 * h;
                        lo = *reinterpret_cast<const half4 *>(row); hi = *reinterpret_cast<const half4 *>(row + 8);
                    }
                    fb[s][t] = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
            }
        }
        #pragma unroll
        for (int Mt = 0; Mt < MT; Mt++) {
            if (Mt % g0 == 0) {
                const int left = MT - Mt - g0;
                const bool more = P.L > 1 || has_last;
                S.template begin_stage<0>((MT - Mt < g0 ? MT - Mt : g0) * KS0, left > 0 ? (left < g0 ? left : g0) * KS0 : (more ? after : 0), false,
                                          [&]() __attribute__((always_inline)) { if (Mt == 0) fetch_fwd(0); });
            }
            const unsigned char *base = S.buf() + (size_t)(Mt % g0) * KS0 * 1024;
            f32x16 acc[NT];
            #pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            if (preload) {
                #pragma unroll
                for (int s = 0; s < KS; s++) {
                    if (s < KS0) {
                        const half8 a = lds_frag(base, s, lane);
                        #pragma unroll
                        for (int t = 0; t < NT; t++) acc[t] = mfma(a, fb[s][t], acc[t]);
                    }
                }
            } else {
                for (int s = 0; s < KS0; s++) {
                    const half8 a = lds_frag(base, s, lane);
                    #pragma unroll
                    for (int t = 0; t < NT; t++) {
                        const _Float16 *row = P.x + (size_t)pt[t] * P.K0 + 16 * s + 4 * h;
                        const half4 lo = *reinterpret_cast<const half4 *>(row), hi = *reinterpret_cast<const half4 *>(row + 8);
                        acc[t] = mfma(a, half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}, acc[t]);
                    }
                }
            }
            epilogue(0, Mt, acc, fa);
        }
    }

    // ---- hidden layers 1 .. L-1: WIDTH x WIDTH, operands ping-pong between fa and fb ---------------------------------------------
    auto hidden = [&](int k, const half8 (&src)[KS][NT], half8 (&dst)[KS][NT]) __attribute__((always_inline)) {
        const int after = (k + 1 < (int)P.L) ? GM * KS : (has_last ? first_stage_frags(MTL, KS) : 0);
        half8 ahead[LA ? 2 : 1][LA ? KS : 1];
        #pragma unroll
        for (int Mt = 0; Mt < MT; Mt++) {
            if (Mt % GM == 0) {
                const int left = MT - Mt - GM;
                // the previous stage group's epilogues (of this layer, or of the hidden layer before it: k >= 2) left GM tiles' stores
                S.template begin_stage<GM * kStoresPerTile>((MT - Mt < GM ? MT - Mt : GM) * KS, left > 0 ? (left < GM ? left : GM) * KS : after,
                                                            wave_full && (Mt != 0 || k >= 2),
                                                            [&]() __attribute__((always_inline)) { if (Mt == 0) fetch_fwd(k); });
                if constexpr (LA) {
                    #pragma unroll
                    for (int s = 0; s < KS; s++) ahead[Mt & 1][s] = lds_frag(S.buf() + (size_t)(Mt % GM) * KS * 1024, s, lane);
                }
            }
            const unsigned char *base = S.buf() + (size_t)(Mt % GM) * KS * 1024;
            if constexpr (LA) {
                if (Mt + 1 < MT && (Mt + 1) % GM != 0) {
                    #pragma unroll
                    for (int s = 0; s < KS; s++) ahead[(Mt + 1) & 1][s] = lds_frag(S.buf() + (size_t)((Mt + 1) % GM) * KS * 1024, s, lane);
                }
            }
            f32x16 acc[NT];
            #pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            #pragma unroll
            for (int s = 0; s < KS; s++) {
                half8 a;
                if constexpr (LA) a = ahead[Mt & 1][s]; else a = lds_frag(base, s, lane);
                #pragma unroll
                for (int t = 0; t < NT; t++) acc[t] = mfma(a, src[s][t], acc[t]);
            }
            epilogue(k, Mt, acc, dst);
        }
    };
    // ---- last layer: M = Mlast rows written to global memory ---------------------------------------------------------------------
    auto last = [&](const half8 (&src)[KS][NT]) __attribute__((always_inline)) {
        const int gl = tiles_per_stage(KS);
        for (int Mt = 0; Mt < MTL; Mt++) {
            if (Mt % gl == 0) {
                const int left = MTL - Mt - gl;
                S.template begin_stage<GM * kStoresPerTile>((MTL - Mt < gl ? MTL - Mt : gl) * KS, left > 0 ? (left < gl ? left : gl) * KS : 0,
                                                            wave_full && Mt == 0 && P.L >= 2, [] {});
            }
            const unsigned char *base = S.buf() + (size_t)(Mt % gl) * KS * 1024;
            f32x16 acc[NT];
            #pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            #pragma unroll
            for (int s = 0; s < KS; s++) {
                const half8 a = lds_frag(base, s, lane);
                #pragma unroll
                for (int t = 0; t < NT; t++) acc[t] = mfma(a, src[s][t], acc[t]);
            }
            #pragma unroll
            for (int t = 0; t < NT; t++) {
                #pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t feat = 32u * Mt + 8u * q + 4u * h;
                    if (feat < P.Mlast && live[t]) {
                        half4 v;
                        #pragma unroll
                        for (int e = 0; e < 4; e++) v[e] = (_Float16)acc[t][4 * q + e];
                        *reinterpret_cast<half4 *>(P.out + (size_t)pt[t] * P.Mlast + feat) = v;
                    }
                }
            }
        }
    };

    int k = 1;
    for (; k + 1 < (int)P.L; k += 2) {
        hidden(k, fa, fb);
        hidden(k + 1, fb, fa);
    }
    if (k < (int)P.L) {
        hidden(k, fa, fb);
        if (has_last) last(fb);
    } else if (has_last) {
        last(fa);
    }
    }   // tile loop
}


constexpr uint32_t kResidentMaxFrags = 64;     // 64 KiB of LDS per workgroup: two workgroups per CU

// points per wave (NT x 32) and waves per SIMD the kernel is compiled for, per hidden width (measured choices)
#ifndef SDN_FF_NT128
#define SDN_FF_NT128 1
#endif
#ifndef SDN_FF_OCC128
#define SDN_FF_OCC128 4
#endif
#ifndef SDN_FF_SW128
#define SDN_FF_SW128 8
#endif
#ifndef SDN_FF_NT64
#define SDN_FF_NT64 1
#endif
#ifndef SDN_FF_OCC64
#define SDN_FF_OCC64 4
#endif
#ifndef SDN_FF_SW64
#define SDN_FF_SW64 8
#endif
#ifndef SDN_FF_SW256
#define SDN_FF_SW256 8
#endif

// RW / SW: waves per workgroup of the resident / streamed kernel
template <int WIDTH, int NT, int MODE, int OCC, int RW, int SW, bool ARELU>
int launch_width(const FfArgs &a, hipStream_t st) {
    if (a.total_frags <= kResidentMaxFrags) {
        const uint32_t ntiles = sdn_div_up(a.B, (uint32_t)(RW * 32 * NT));
        const uint32_t lds = a.total_frags * 1024u;
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        uint32_t per_cu = 163840u / (lds > 1024u ? lds : 1024u);
        if (per_cu > (uint32_t)(OCC * 4 / RW)) per_cu = OCC * 4 / RW;     // waves per SIMD the register budget allows
        if (per_cu == 0) per_cu = 1;
        const uint32_t grid = ntiles < (uint32_t)cus * per_cu ? ntiles : (uint32_t)cus * per_cu;
        hipLaunchKernelGGL((k_ffmlp<WIDTH, NT, MODE, true, OCC, RW, ARELU>), dim3(grid), dim3(64 * RW), lds, st, a);
    } else {
        const uint32_t ntiles = sdn_div_up(a.B, (uint32_t)(SW * 32 * NT));
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        if constexpr ((WIDTH == 64 || WIDTH == 128) && OCC > 2) {
            // A launch with no more workgroups than CUs: the latency variant, and as few waves per workgroup as still leave at most two
            // workgroups per CU -- the per-layer critical path (stage barrier, the workgroup's activation stores through ONE address
            // pipe, the MFMAs of the waves sharing a SIMD) shrinks with the rows a workgroup owns; the weights come from L2 anyway.
            if (ntiles <= (uint32_t)cus) {
                if (sdn_div_up(a.B, (uint32_t)(2 * 32 * NT)) <= 2u * (uint32_t)cus)
                    hipLaunchKernelGGL((k_ffmlp<WIDTH, NT, MODE, false, 2, 2, ARELU>), dim3(sdn_div_up(a.B, (uint32_t)(2 * 32 * NT))), dim3(128), 2 * kStageBytes, st, a);
                else if (sdn_div_up(a.B, (uint32_t)(4 * 32 * NT)) <= 2u * (uint32_t)cus)
                    hipLaunchKernelGGL((k_ffmlp<WIDTH, NT, MODE, false, 2, 4, ARELU>), dim3(sdn_div_up(a.B, (uint32_t)(4 * 32 * NT))), dim3(256), 2 * kStageBytes, st, a);
                else
                    hipLaunchKernelGGL((k_ffmlp<WIDTH, NT, MODE, false, 2, SW, ARELU>), dim3(ntiles), dim3(64 * SW), 2 * kStageBytes, st, a);
                return sdn_launch_status();
            }
        }
        hipLaunchKernelGGL((k_ffmlp<WIDTH, NT, MODE, false, OCC, SW, ARELU>), dim3(ntiles), dim3(64 * SW), 2 * kStageBytes, st, a);
    }
    return sdn_launch_status();
}

template <int MODE, bool ARELU>
int launch_fused_t(uint32_t W, const FfArgs &a, hipStream_t st) {
    switch (W) {
        case 16: return launch_width<16, 2, MODE, 4, 4, 4, ARELU>(a, st);
        case 32: return launch_width<32, 2, MODE, 4, 4, 4, ARELU>(a, st);
        case 64: return launch_width<64, SDN_FF_NT64, MODE, SDN_FF_OCC64, 4, SDN_FF_SW64, ARELU>(a, st);
        case 128: return launch_width<128, SDN_FF_NT128, MODE, SDN_FF_OCC128, 4, SDN_FF_SW128, ARELU>(a, st);
        case 256: return launch_width<256, 1, MODE, 2, 4, SDN_FF_SW256, ARELU>(a, st);
        default: return SDN_E_UNSUPPORTED;
    }
}


}  // namespace sdn_ff
