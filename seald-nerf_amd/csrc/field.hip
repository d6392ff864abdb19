// Fused dynamic-NeRF field evaluation on MFMA for gfx950 (MI355X), inference.
//
//   (xyz, dir) --freq--> deform MLP (8 x 128) --> xyz' --tiled grid (16 lv x 2)--> sigma MLP (64, 16)
//                                                    dir --SH(4)--^ ++ geo_feat --> color MLP (64, 64, 3)
//
// Behavioural contract: NeRFNetwork.forward of the reference under `-O` (dnerf/network.py:123-169 with torch
// autocast): every Linear takes fp16 inputs / weights, accumulates in fp32 and rounds its output to fp16; the
// encoders run in fp32 (their custom_fwd casts); the grid table and its output are fp16 with the per-corner
// half rounding of kernel_grid (gridencoder.cu:164-191); sigma = exp(fp32(h0)); rgb = fp16(sigmoid(.)).
// In the reference this is 13 cuBLAS GEMMs + ~25 elementwise / encoder kernels per loop iteration, each
// round-tripping [M,128] activations through HBM.  Here it is ONE launch and activations never leave registers.
//
// Mapping (see DESIGN.md "fused field kernel"):
//   * workgroup = 8 waves = 256 sample points (two workgroups per CU: 4 waves per SIMD, <= 128 VGPRs -- measured faster
//     than 6-wave workgroups at 3 waves per SIMD with deeper LDS prefetch: occupancy wins); one wave = 32 points = the N dimension of
//     v_mfma_f32_32x32x16_f16.  Weights are the A operand (M = output features), activations the B operand
//     (K = input features).  The accumulator of layer i (feature rows in registers, point on the lane) converts
//     in place (ReLU, cvt_pk_f16) into the B operand of layer i+1 -- no cross-lane traffic between layers.  The
//     k-order this implies (element j of lane-half h <-> feature 16s + 8(j>>2) + 4h + (j&3)) is baked into the
//     host-side weight packing (dnerf_amd/fused.py), as are the first layer's freq-feature order, the
//     grid-feature order and the SH / geo_feat order of the colour net.
//   * weights (240 fragments of 1 KiB, fragment order) are streamed through LDS in 8 stages
//     (D0 16 KiB | D1..D6 32 KiB each | D7+S0+S1+C0+C1+C2 32 KiB) with direct-to-LDS loads
//     (global_load_lds_dwordx4), double-buffered, one barrier per stage: stage s+1 lands while stage s feeds
//     the MFMAs through conflict-free ds_read_b128 (lane-linear fragments).  HBM/L2 sees each weight byte once
//     per 256 points instead of once per 32.
//   * the time encoding is the same for every point: its contribution W0[:,63:76] . enc(t) is a per-frame
//     bias vector (computed on the host) loaded as the initial accumulator of the first layer.
//   * the 128 table gathers per point are split over the two lane-halves (levels 0-7 / 8-15); row strides,
//     wrap masks and scales of the tiled levels are host-precomputed (no integer division on the device).
//   * optional live-sample index list: only slots that hold a sample are evaluated (the reference evaluates
//     the network on every padded slot).
#include <math.h>
#include <type_traits>
#include <stdlib.h>

#include "sdn_common.h"
#include "cell_points.h"
#include "grid_common.h"
#include "sh_eval.h"


// Staggered half-workgroups in the hidden layers (waves 4-7 half a layer behind waves 0-3): built and bit-identical, measured neutral
// to -3 % (profiles/r03_rejected_field_stagger.txt) -- compiled out by default; `make -C seald-nerf_amd/csrc stagger` builds the variant.
#ifndef SDN_FIELD_STAGGER
#define SDN_FIELD_STAGGER 0
#endif

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef SDN_STAMPS
// Diagnostic build only (make diag; the product library has no stamp): raw shader-clock stamps of k_field_f16, stored as they are taken
// (nothing is kept in registers: the throughput variant has none to spare) by lane 0 of waves 0 and 7 of the first 2048 workgroups.
// g_field_stamps[(2 wg + w) * 32 + k] (w = 0: wave 0, 1: wave 7):
//   k = 0 entry  1 point loaded + freq features  2 first stage barrier passed  3 D0 issued  4..9 hidden layers D1..D6 issued
//   10 last conversion + tail-stage barrier  11 D7 + deformation  12 grid encode  13 sigma net  14 SH + colour net  15 sigmoid + stores
//   16 HW_ID register (CU / SE / SIMD of the wave)  17 XCC_ID register
constexpr uint32_t kStampWGs = 2048;
__device__ unsigned long long g_field_stamps[kStampWGs * 2 * 32];
#define FSTAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (stamp_on) { stamp_slot[k] = __builtin_amdgcn_s_memtime(); } __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define FSTAMP(k) ((void)0)
#endif

// fragment (1 KiB block) indices inside the packed weight buffer
constexpr int kBlkD0 = 0;                 // 4 Mt x 4 ks
constexpr int kBlkD1 = kBlkD0 + 16;       // 6 layers x (4 Mt x 8 ks)
constexpr int kBlkD7 = kBlkD1 + 6 * 32;   // 1 Mt x 8 ks
constexpr int kBlkS0 = kBlkD7 + 8;        // 2 Mt x 2 ks
constexpr int kBlkS1 = kBlkS0 + 4;        // 1 Mt x 4 ks
constexpr int kBlkC0 = kBlkS1 + 4;        // 2 Mt x 2 ks
constexpr int kBlkC1 = kBlkC0 + 4;        // 2 Mt x 4 ks
constexpr int kBlkC2 = kBlkC1 + 8;        // 1 Mt x 4 ks
constexpr int kBlkTotal = kBlkC2 + 4;     // 240
static_assert(kBlkTotal - kBlkD7 == 32, "the tail stage must be exactly one 32 KiB buffer");

constexpr int kStageBytes = 32768;
constexpr int kWaves = 8;                 // waves per workgroup, 32 points each; two workgroups per CU = 4 waves per SIMD
constexpr int kPointsPerWG = 32 * kWaves;

// tiled-grid level constants (D = 3, align_corners = false), host-precomputed: gridencoder.cu:66-84,138-139
struct TiledLevels {
    uint32_t offset[16];  // first row of the level
    uint32_t s1[16];      // row stride of +1 in y (0 if the dimension is dropped: stride > rows)
    uint32_t s2[16];      // row stride of +1 in z (0 if dropped)
    uint32_t hsize[16];   // rows in the level
    uint32_t mask[16];    // hsize - 1 if hsize is a power of two (wrapping level), else 0xFFFFFFFF (dense level)
    float scale[16];
};

struct FieldArgs {
    const float *xyzs;        // [M,3]
    const float *dirs;        // [M,3]
    const uint32_t *live_idx; // [<=M] slot indices to evaluate, or nullptr = all M slots
    const uint32_t *live_count;
    const int32_t *state;     // device-driven loop: the count is live_count[state[3]] (one counter per iteration), else nullptr
    uint32_t M;
    const unsigned char *weights;  // kBlkTotal KiB, fragment order
    const float *bias0;       // [128] time-encoding contribution to the first deform layer
    const __half *table;      // grid embeddings, fp16 [rows, 2]
    float *sigmas;            // [M]
    float *rgbs;              // [M,3]
    float bound;
    float inv_2bound;         // 1 / (2 bound) if that is a power of two (the division is then an exact multiplication), else 0
    float density_scale;
    int zero_deform;          // bit f: frame f is at t == 0, the canonical frame (dnerf/network.py:140-141); a single frame uses bit 0
    const uint8_t *slot_frame;  // frame group: frame of every sample slot (selects bias0 + 128 f and bit f of zero_deform), or nullptr
    // density-grid query (CELLS variant): the points are jittered centres of occupancy-grid cells, built in the kernel
    const float *cell_noise;  // [count,3] uniform [0,1) by list position, or nullptr = counter-based generator on cell_seed
    uint32_t cell_seed;
    float cell_inv;           // 1 / (grid_size - 1) in fp32: torch divides a tensor by a host scalar as a multiplication by its reciprocal
    float cell_span;          // bound_cas - half_grid   (dnerf/renderer.py:484-488)
    float cell_half;          // half_grid = bound_cas / grid_size
    // workgroup stagger: workgroups [stagger_lo, stagger_hi) -- the second workgroup of every CU in the launch's first wave of
    // workgroups -- start `stagger_cycles` shader cycles late, so that the two workgroups of a CU run their matrix and vector phases out
    // of step (0 = off)
    uint32_t stagger_cycles, stagger_lo, stagger_hi;
    uint32_t n_frames;        // rows of bias0 (frames of a frame group; 1 without slot_frame)
    uint32_t pp_soft;         // persistent kernel: the workgroup count to stay within unless more workgroups save a whole round (0 = gridDim.x)
};

using sdn_cells::cell_uniform;
using sdn_cells::compact_bits3;

__device__ __forceinline__ f32x16 mfma(half8 a, half8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// accumulator tile -> the two B fragments (k-steps) it provides to the next layer
// (the reference rounds the Linear output to fp16 and applies ReLU on the fp16 tensor: round first, then a packed max)
template <bool RELU>
__device__ __forceinline__ void acc_to_frags(const f32x16 &acc, half8 &f0, half8 &f1) {
    #pragma unroll
    for (int j = 0; j < 8; j++) {
        f0[j] = (_Float16)acc[j];
        f1[j] = (_Float16)acc[8 + j];
    }
    if (RELU) {
        const half8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
        f0 = __builtin_elementwise_max(f0, zero);
        f1 = __builtin_elementwise_max(f1, zero);
    }
}

__device__ __forceinline__ float round_h(float v) { return (float)(_Float16)v; }

// fp32 multiply with one operand taken straight from the low / high half of a packed fp16 pair (v_fma_mix_f32):
//   mix_mul_*(w, h2) = w * float(h2.half)        as fma(w, half, -0)  -- identical to the rounded product for every input
__device__ __forceinline__ float mix_mul_lo(float w, uint32_t h2) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]" : "=v"(r) : "v"(w), "v"(h2), "s"(-0.0f));
    return r;
}
__device__ __forceinline__ float mix_mul_hi(float w, uint32_t h2) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(w), "v"(h2), "s"(-0.0f));
    return r;
}
// sin(a) and cos(a) with one shared 3-term Cody-Waite reduction by pi (explicit FMAs: this file is built with -ffp-contract=off):
// r = a - k pi in [-pi/2, pi/2], sin(a) = (-1)^k sin(r), cos(a) = (-1)^k cos(r); odd degree-9 / even degree-10 polynomials,
// ~1.3e-7 absolute for |a| < ~1e4.  The standalone freq_encode kernel uses OCML sinf (<= 1 ulp); the two agree to ~1e-7, far below
// the fp16 rounding the features get as MFMA operands.
__device__ __forceinline__ void fast_sincos(float a, float &sn, float &cs) {
    const float k = rintf(a * 0.31830988618379067f);
    float r = __builtin_fmaf(-k, 3.140625f, a);
    r = __builtin_fmaf(-k, 9.67502593994140625e-4f, r);
    r = __builtin_fmaf(-k, 1.509957990978376e-7f, r);
    const float r2 = r * r;
    float p = __builtin_fmaf(r2, 2.6083159809786593e-6f, -1.9810690719168633e-4f);
    p = __builtin_fmaf(p, r2, 8.3330785855650902e-3f);
    p = __builtin_fmaf(p, r2, -1.6666659712791443e-1f);
    const float s = __builtin_fmaf(r * r2, p, r);
    float q = __builtin_fmaf(r2, -2.6051615e-07f, 2.4760495e-05f);
    q = __builtin_fmaf(q, r2, -1.3888378e-03f);
    q = __builtin_fmaf(q, r2, 4.1666638e-02f);
    q = __builtin_fmaf(q, r2, -0.5f);
    const float c = __builtin_fmaf(q, r2, 1.0f);
    const int sign = ((int)k & 1) << 31;
    sn = __int_as_float(__float_as_int(s) ^ sign);
    cs = __int_as_float(__float_as_int(c) ^ sign);
}

// Stage `nbytes` (multiple of 4 KiB) of packed weights global -> LDS, all 256 threads, 16 B per lane per instruction.
// One wave-instruction moves 1 KiB to a wave-uniform LDS base + lane * 16 (global_load_lds_dwordx4).
__device__ __forceinline__ void stage_load(const unsigned char *__restrict__ g, unsigned char *lds, int nbytes, uint32_t wave, uint32_t lane) {
    for (int c = (int)wave; c < nbytes / 1024; c += kWaves) {
        const uint32_t off = __builtin_amdgcn_readfirstlane((uint32_t)c * 1024u);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + off + lane * 16),
                                         (__attribute__((address_space(3))) void *)(lds + off), 16, 0, 0);
    }
}

// One 1-KiB piece of a stage: piece k of this wave (pieces are dealt to the waves round-robin, as stage_load does).
__device__ __forceinline__ void stage_piece(const unsigned char *__restrict__ g, unsigned char *lds, int k, uint32_t wave, uint32_t lane) {
    const uint32_t off = __builtin_amdgcn_readfirstlane((wave + (uint32_t)k * kWaves) * 1024u);
    // (uniform 64-bit base in SGPRs + one 32-bit lane offset: the saddr form of the load, no 64-bit VGPR address per piece)
    const __attribute__((address_space(1))) unsigned char *base = (const __attribute__((address_space(1))) unsigned char *)(g + off);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (lane << 4)),
                                     (__attribute__((address_space(3))) void *)(lds + off), 16, 0, 0);
}

// A staged buffer may be read once (a) every wave's own direct-to-LDS loads have landed -- they are pending LDS writes on the
// VM counter, and hipcc does NOT reliably emit the vmcnt wait in front of __syncthreads() for them (checked in the ISA:
// only lgkmcnt was waited) -- and (b) the workgroup has met at the barrier.
__device__ __forceinline__ void stage_wait_and_sync() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

__device__ __forceinline__ half8 lds_frag(const unsigned char *buf, int blk, uint32_t lane) {
    return *reinterpret_cast<const half8 *>(buf + (size_t)blk * 1024 + lane * 16);
}

// OCC = waves per SIMD the kernel is compiled for, LA = LDS fragment reads kept in flight ahead of the MFMAs of a hidden layer.
//   <4, 2>: the throughput variant (128 VGPRs, two workgroups per CU) for launches that fill the chip;
//   <2, 8>: the latency variant for launches of at most one workgroup per CU (the tail iterations of a frame, every iteration
//           of a ray-sharded frame): with two waves per SIMD nothing hides an LDS round trip per MFMA, so a whole k-step of
//           fragments is read ahead (256-VGPR budget).
// CELLS: the density-grid query of update_extra_state (dnerf/renderer.py:453-555): slot p is a Morton cell index, the point is
//   that cell's jittered centre, only sigma is evaluated and it is stored at sigmas[p] (a slice of tmp_grid).
// LAYOUT of the fp16 table:
//   kLayoutRef   the reference's (grid.py:118-127): the (x, x+1) pair of a gather is clamped into the level, a wrap is patched;
//   kLayoutPad   every level followed by one extra row that repeats the level's row 0: the (x, x+1) row pair of a gather is always two
//                consecutive rows -- no clamp, no wrap bookkeeping, no patch path; 4 gathers of 8 B per level;
//   kLayoutQuad  one 16-byte BLOCK per row r of a level: rows {r, r+1, r+s1, r+s1+1} (each mod the level's row count) -- the four
//                (x, y) corners of the cell whose low corner is row r (sdn_field_build_quad_table): 2 gathers of 16 B per level.  The
//                phase is bound by the rate at which the texture-address unit takes divergent lane addresses (~1 per cycle and CU:
//                64 gathers per point are as many cycles of that unit as the point's 240 MFMAs are of a matrix pipe), not by bytes:
//                half the gathers, the same values into the same arithmetic.
constexpr int kLayoutRef = 0, kLayoutPad = 1, kLayoutQuad = 2;
template <int OCC, int LA, bool CELLS, int LAYOUT>
__global__ void __launch_bounds__(64 * kWaves, OCC) k_field_f16(FieldArgs P, TiledLevels lv) {
    constexpr bool PAD = LAYOUT != kLayoutRef;   // (no clamp / wrap bookkeeping in either derived layout)
    constexpr int kGridBatch = OCC >= 4 ? 4 : 8;   // grid levels (per lane-half) whose gathers are in flight together (register budget)
    // Two SEPARATE LDS arrays, and a hidden-layer loop unrolled so that every access names its array at compile time: the compiler
    // treats a direct-to-LDS load as a store to LDS and puts `s_waitcnt vmcnt(0)` in front of every later LDS read it cannot prove
    // disjoint -- with one array indexed by `l & 1` that was every layer's first fragment read, i.e. each layer waited for the
    // NEXT layer's 32 KiB to land before it issued its first MFMA (found in the ISA; ~600 cycles per layer).
    __shared__ __attribute__((aligned(16))) unsigned char s_w0[kStageBytes];
    __shared__ __attribute__((aligned(16))) unsigned char s_w1[kStageBytes];
    // (the largest alignment of the three: the LDS layout pass then places it at offset 0, where its 32 per-level reads address it with
    //  the instructions' 16-bit immediate offsets instead of one v_or_b32 each -- behind the two 32 KiB arrays it sat at 0x10000)
    __shared__ __attribute__((aligned(4096))) uint4 s_lv[16][2];   // per grid level: {offset, s1, s2, hsize}, {mask, scale, -, -}
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#ifdef SDN_STAMPS
    const uint32_t wave_u = __builtin_amdgcn_readfirstlane(wave);
    const bool stamp_on = ((wave_u == 0u) | (wave_u == 7u)) && lane == 0 && blockIdx.x < kStampWGs;
    unsigned long long *stamp_slot = g_field_stamps + ((size_t)blockIdx.x * 2 + (wave_u ? 1 : 0)) * 32;
    FSTAMP(0);
    if (stamp_on) {
        stamp_slot[16] = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
        stamp_slot[17] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID
    }
#endif
    if (threadIdx.x < 16) {         // visible to everyone after the first stage barrier below
        const uint32_t l = threadIdx.x;
        s_lv[l][0] = make_uint4(lv.offset[l], lv.s1[l], lv.s2[l], lv.hsize[l]);
        s_lv[l][1] = make_uint4(lv.mask[l], __float_as_uint(lv.scale[l]), 0u, 0u);
    }
    const uint32_t count = P.state ? P.live_count[P.state[3]] : (P.live_idx ? *P.live_count : P.M);
    if (blockIdx.x * (uint32_t)kPointsPerWG >= count) return;  // workgroup-uniform: nothing to do, no barrier touched
    if (P.stagger_cycles != 0u && blockIdx.x >= P.stagger_lo && blockIdx.x < P.stagger_hi) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)P.stagger_cycles) __builtin_amdgcn_s_sleep(64);
    }
    const uint32_t n = lane & 31u, h = lane >> 5;
    const uint32_t i = blockIdx.x * (uint32_t)kPointsPerWG + wave * 32u + n;
    const bool valid = i < count;
    const uint32_t ii = valid ? i : (count - 1);  // idle lanes recompute the last point and store nothing
    const uint32_t p = P.live_idx ? P.live_idx[ii] : ii;

    // stage 0 (D0, 16 KiB) -> buf 0 and stage 1 (D1) -> buf 1 start now and land under the feature computation
    stage_load(P.weights + (size_t)kBlkD0 * 1024, s_w0, 16 * 1024, wave, lane);
    stage_load(P.weights + (size_t)kBlkD1 * 1024, s_w1, kStageBytes, wave, lane);

    float x0, x1, x2, d0 = 0, d1 = 0, d2 = 0;
    if constexpr (CELLS) {
        // dnerf/renderer.py:480-490 (== :517-524), the same fp32 operations in the same order:
        //   xyzs = 2 * coords / (grid_size - 1) - 1;  cas_xyzs = xyzs * (bound - half_grid);  cas_xyzs += (rand * 2 - 1) * half_grid
        float xs[3];
        #pragma unroll
        for (int d = 0; d < 3; d++) {
            const float c = (float)compact_bits3(p >> d);
            const float r = P.cell_noise ? P.cell_noise[(size_t)ii * 3 + d] : cell_uniform(P.cell_seed, ii * 3u + (uint32_t)d);
            xs[d] = ((2.0f * c) * P.cell_inv - 1.0f) * P.cell_span + (r * 2.0f - 1.0f) * P.cell_half;
        }
        x0 = xs[0]; x1 = xs[1]; x2 = xs[2];
    } else {
        x0 = P.xyzs[(size_t)p * 3]; x1 = P.xyzs[(size_t)p * 3 + 1]; x2 = P.xyzs[(size_t)p * 3 + 2];
        d0 = P.dirs[(size_t)p * 3]; d1 = P.dirs[(size_t)p * 3 + 1]; d2 = P.dirs[(size_t)p * 3 + 2];
    }

    // ---------------- deform layer 0: freq features as B fragments ----------------
    // lane-half h owns (freq, dim) pairs 15h .. 15h+14 (sin and cos) plus x0,x1 (h = 0) / x2,pad (h = 1), i.e. the five octaves
    // 2^(5h) .. 2^(5h+4) of every coordinate.  One sine / cosine pair per coordinate at the lane-half's base octave (shared range
    // reduction, two short polynomials), the four higher octaves by angle doubling in fp32:  s' = 2 s c,  c' = 1 - 2 s^2  -- 6
    // polynomial evaluations + 48 multiply-adds per lane instead of 30 sine evaluations.  The doubling error (<= 2^4 x 1e-7) is two
    // orders of magnitude below the fp16 rounding the features get as MFMA operands; kernel_freq (freqencoder.cu:52-56) evaluates
    // cos as sin(x 2^f + float(pi/2)), which is itself off by up to 3e-5 at 2^9 -- the values here are the closer to the exact ones.
    half8 bf[8];
    {
        const float xs[3] = {x0, x1, x2};
        const float fscale = h ? 32.0f : 1.0f;
        float sv[5][3], cv[5][3];
        #pragma unroll
        for (int dd = 0; dd < 3; dd++) {
            // (v_sin_f32 / v_cos_f32 on revolutions here measured no faster and leave 1.6 % of the fp16 features off the exactly rounded
            //  value against 0.14 % for this pair and 0.40 % for the reference's own float form: profiles/r04_field_valu_diet.txt)
            fast_sincos(xs[dd] * fscale, sv[0][dd], cv[0][dd]);
            #pragma unroll
            for (int f = 1; f < 5; f++) {
                const float sp = sv[f - 1][dd], cp = cv[f - 1][dd];
                const float s2 = sp + sp;                         // (2 s) c and 1 - (2 s) s: the same roundings as 2 (s c) and 1 - 2 s^2
                sv[f][dd] = s2 * cp;
                cv[f][dd] = __builtin_fmaf(-s2, sp, 1.0f);
            }
        }
        #pragma unroll
        for (int s = 0; s < 4; s++) {
            #pragma unroll
            for (int j = 0; j < 8; j++) {
                const int q = s * 8 + j;
                float v;
                if (q < 30) {
                    const int pr = q >> 1, f = pr / 3, dd = pr % 3;
                    v = (q & 1) ? cv[f][dd] : sv[f][dd];
                } else if (q == 30) {
                    v = h ? x2 : x0;
                } else {
                    v = h ? 0.0f : x1;
                }
                bf[s][j] = (_Float16)v;
            }
        }
    }
    f32x16 acc[4];
    const uint32_t frame = P.slot_frame ? P.slot_frame[p] : 0u;   // kernel-uniform condition
    {
        const float *__restrict__ b0 = P.bias0 + 128u * frame;
        #pragma unroll
        for (int mt = 0; mt < 4; mt++) {
            #pragma unroll
            for (int r = 0; r < 16; r++) acc[mt][r] = b0[32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h];
        }
    }
    // Wave priority follows the phase.  All vector instructions of a SIMD share one issue port and the arbiter prefers the oldest
    // wave, so a co-resident workgroup that is in its VALU-only grid phase starves a younger one's MFMAs completely (measured:
    // the second workgroup of a CU made no progress until the first had retired -- two tiles took 28 + 23 us, not ~35).  An MFMA
    // needs the port for 8 of its 32 cycles: with the matrix phases at higher priority the other workgroup's VALU work
    // fills the remaining 24 and the two phases overlap.
    FSTAMP(1);
    __builtin_amdgcn_s_setprio(2);
#if SDN_FIELD_STAGGER
    // ---- staggered half-workgroups (cdna_hip_programming.md / MI355X_MICROARCH.md "Two waves per SIMD", item 9) ---------------------------
    // The eight waves of a workgroup used to pass every layer boundary together: accumulator -> operand conversion (~100 VALU
    // instructions), stage barrier, then 32 MFMAs -- the matrix pipe idled through every conversion, and the co-resident workgroup ran
    // the same phases at the same time more often than not (in-kernel stamps of round 2: 3.6 K cycles per hidden layer where the
    // pipe needs 2.0 K).  Now waves 4-7 (a SIMD's second wave of this workgroup) run HALF A LAYER behind waves 0-3: a hidden layer is
    // two units of 16 MFMAs (k-steps 0-3 / 4-7) with one barrier each, and while one half-workgroup converts and starts a layer the
    // other issues the second half of the previous layer's MFMAs.  Waves 4-7 simply pass one barrier more before layer 0 and one
    // less before the tail (different points, no data dependence -- only the two 32 KiB weight buffers are shared):
    //   global phase p:   waves 0-3 run unit p (p = -1: layer 0, 0..11: hidden units, 12: tail), waves 4-7 unit p - 1.
    //   stage s (2..7) refills buffer s & 1 in phase 2 s - 3: its previous content, stage s - 2, was last read by waves 4-7 in phase
    //   2 s - 4, and waves 0-3 first read stage s in phase 2 s - 2, behind every wave's vmcnt(0) + the barrier of that phase.
    const bool late = __builtin_amdgcn_readfirstlane(wave) >= 4u;
    stage_wait_and_sync();            // barrier of phase -1: D0 and D1 are resident
    if (late) stage_wait_and_sync();  // barrier of phase 0 (waves 0-3 arrive at it after their layer 0)
#else
    stage_wait_and_sync();  // D0 and D1 are resident
#endif
    FSTAMP(2);
    #pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        #pragma unroll
        for (int mt = 0; mt < 4; mt++) acc[mt] = mfma(lds_frag(s_w0, mt * 4 + ks, lane), bf[ks], acc[mt]);
    }
    FSTAMP(3);
#if SDN_FIELD_STAGGER
    // One unit: half a hidden layer (k-steps 4 HALF .. 4 HALF + 3), fragments from `cur`; refill (wave-uniform, run time: it depends
    // on the half-workgroup): this wave's four 1-KiB pieces of stage `stage` go to `other`, issued together in front of the MFMAs
    // under ONE scalar branch (a branch per piece between the MFMAs cut the chain into scheduling regions and spilled registers).
    auto unit = [&](auto half_c, bool refill, const unsigned char *cur, unsigned char *other, int stage) __attribute__((always_inline)) {
        constexpr int half = decltype(half_c)::value;
        stage_wait_and_sync();
        if (refill) {
            // (stage 7, the tail, follows D6 in the packed buffer: one formula for every stage)
            const unsigned char *__restrict__ refill_src = P.weights + (size_t)(kBlkD1 + (stage - 1) * 32) * 1024;
            #pragma unroll
            for (int k = 0; k < kStageBytes / 1024 / kWaves; k++) stage_piece(refill_src, other, k, wave, lane);
        }
        if constexpr (half == 0) {
            #pragma unroll
            for (int t = 0; t < 4; t++) acc_to_frags<true>(acc[t], bf[2 * t], bf[2 * t + 1]);
            __builtin_amdgcn_sched_barrier(0);     // (the fragment read-ahead must not be hoisted over the conversion: it spills at 128 VGPRs)
        }
        half8 ring[LA];
        #pragma unroll
        for (int j = 0; j < LA; j++) ring[j] = lds_frag(cur, (j & 3) * 8 + 4 * half + (j >> 2), lane);
        #pragma unroll
        for (int i = 0; i < 16; i++) {
            const int ks = 4 * half + (i >> 2), mt = i & 3;
            const half8 a = ring[i % LA];
            if (i + LA < 16) ring[i % LA] = lds_frag(cur, ((i + LA) & 3) * 8 + 4 * half + ((i + LA) >> 2), lane);
            if (ks == 0) {
                f32x16 z;
                #pragma unroll
                for (int r = 0; r < 16; r++) z[r] = 0.0f;
                acc[mt] = mfma(a, bf[0], z);
            } else {
                acc[mt] = mfma(a, bf[ks], acc[mt]);
            }
        }
        __builtin_amdgcn_sched_group_barrier(0x100, LA, 0);
        #pragma unroll
        for (int i = 0; i < 16 - LA; i++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, LA, 0);
    };
    static_assert(LA <= 16 && kBlkD7 == kBlkD1 + 6 * 32, "look-ahead is per unit of 16 MFMAs; the tail stage follows D6");
    constexpr std::integral_constant<int, 0> H1{};
    constexpr std::integral_constant<int, 1> H2{};
    #pragma unroll 1
    for (int b2 = 0; b2 < 3; b2++) {      // layers 2 b2 + 1 (stage in s_w1) and 2 b2 + 2 (stage in s_w0): units 4 b2 .. 4 b2 + 3
        // waves 0-3 refill in their odd units (stage (u + 3) / 2), waves 4-7 -- one phase behind -- in their even units (stage (u + 4) / 2)
        unit(H1, late, s_w1, s_w0, 2 * b2 + 2);
        unit(H2, !late, s_w1, s_w0, 2 * b2 + 2);
        unit(H1, late, s_w0, s_w1, 2 * b2 + 3);
        unit(H2, !late, s_w0, s_w1, 2 * b2 + 3);
    }
    #pragma unroll
    for (int t = 0; t < 4; t++) acc_to_frags<true>(acc[t], bf[2 * t], bf[2 * t + 1]);
    if (!late) stage_wait_and_sync();   // barrier of phase 12: the tail stage (D7 | S0 | S1 | C0 | C1 | C2) is resident in buffer 1 (waves 4-7 passed it before their last unit)
#else
    // ---------------- deform layers 1..6 (128 -> 128, ReLU): stage l+1 uses buffer (l+1)&1 ----------------
    // One hidden layer: fragments from `cur`, refill of `other` (static arrays: see the comment at their declaration).
    auto hidden_layer = [&](int l, const unsigned char *cur, unsigned char *other) __attribute__((always_inline)) {
        #pragma unroll
        for (int t = 0; t < 4; t++) acc_to_frags<true>(acc[t], bf[2 * t], bf[2 * t + 1]);
        stage_wait_and_sync();  // stage l+1 landed for everyone; everyone is done reading the other buffer (layer l)
        // refill the other buffer with stage l+2 (D(l+2) for l < 5, the tail stage for l == 5); it lands under this layer's MFMAs
        // (its four 1-KiB pieces per wave are issued between this layer's MFMAs, not in front of them: a direct-to-LDS load costs
        // 60-180 cycles of issue)
        const unsigned char *__restrict__ refill_src = P.weights + (size_t)(l < 5 ? kBlkD1 + (l + 1) * 32 : kBlkD7) * 1024;
        // (a software-pipelined variant -- fragments of k-step ks+1 read while the MFMAs of ks run -- needs 32 more VGPRs,
        // i.e. 3 waves per SIMD instead of 4, and measured 20 % slower: latency is hidden by occupancy here)
        // look-ahead: the LDS reads of fragments i+1 .. i+LA are in flight while MFMA i issues
        half8 ring[LA];
        #pragma unroll
        for (int j = 0; j < LA; j++) ring[j] = lds_frag(cur, (j & 3) * 8 + (j >> 2), lane);
        #pragma unroll
        for (int i = 0; i < 32; i++) {
            const int ks = i >> 2, mt = i & 3;
            const half8 a = ring[i % LA];
            if (i + LA < 32) ring[i % LA] = lds_frag(cur, ((i + LA) & 3) * 8 + ((i + LA) >> 2), lane);
            if ((i & 7) == 1 && (i >> 3) < kStageBytes / 1024 / kWaves) stage_piece(refill_src, other, i >> 3, wave, lane);
            if (ks == 0) {
                f32x16 z;
                #pragma unroll
                for (int r = 0; r < 16; r++) z[r] = 0.0f;
                acc[mt] = mfma(a, bf[0], z);
            } else {
                acc[mt] = mfma(a, bf[ks], acc[mt]);
            }
        }
        // pin the interleaving the look-ahead needs (left alone, the scheduler sinks every read to just before its MFMA to save
        // registers, and each MFMA then waits out a full LDS round trip): LA reads, then MFMA / read alternating
        __builtin_amdgcn_sched_group_barrier(0x100, LA, 0);
        #pragma unroll
        for (int i = 0; i < 32 - LA; i++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, LA, 0);
    };
    #pragma unroll 1
    for (int lp = 0; lp < 3; lp++) {      // layers 2 lp (reads s_w1, refills s_w0) and 2 lp + 1 (the other way round)
        hidden_layer(2 * lp, s_w1, s_w0);
        FSTAMP(4 + 2 * lp);
        hidden_layer(2 * lp + 1, s_w0, s_w1);
        FSTAMP(5 + 2 * lp);
    }
    #pragma unroll
    for (int t = 0; t < 4; t++) acc_to_frags<true>(acc[t], bf[2 * t], bf[2 * t + 1]);
    stage_wait_and_sync();  // tail stage (D7 | S0 | S1 | C0 | C1 | C2) resident in buffer 1
    FSTAMP(10);
#endif
    const unsigned char *tail = s_w1;
    constexpr int tD7 = 0, tS0 = kBlkS0 - kBlkD7, tS1 = kBlkS1 - kBlkD7, tC0 = kBlkC0 - kBlkD7, tC1 = kBlkC1 - kBlkD7, tC2 = kBlkC2 - kBlkD7;

    // ---------------- deform layer 7 (128 -> 3) ----------------
    float u[3];
    {
        f32x16 o;
        #pragma unroll
        for (int r = 0; r < 16; r++) o[r] = 0.0f;
        #pragma unroll
        for (int ks = 0; ks < 8; ks++) o = mfma(lds_frag(tail, tD7 + ks, lane), bf[ks], o);
        // rows 0..2 = registers 0..2 of lane-half 0; broadcast to both halves
        float df[3];
        #pragma unroll
        for (int c = 0; c < 3; c++) df[c] = __shfl(round_h(o[c]), (int)n, 64);
        const float xs[3] = {x0, x1, x2};
        #pragma unroll
        for (int c = 0; c < 3; c++) {
            const float xd = ((P.zero_deform >> frame) & 1) ? xs[c] : xs[c] + df[c];
            // GridEncoder.forward (grid.py:149): (x + bound) / (2 bound); for 2 bound a power of two the quotient is the exact product
            u[c] = P.inv_2bound != 0.0f ? (xd + P.bound) * P.inv_2bound : (xd + P.bound) / (2 * P.bound);
        }
    }
    __builtin_amdgcn_s_setprio(0);
    FSTAMP(11);
    // ---------------- grid encode: lane-half h evaluates levels 8h .. 8h+7 ----------------
    half8 gf[2];
    uint32_t gfw[2][4];   // the same 2 x 8 halfs as packed pairs
    {
        const bool oob = (u[0] < 0) | (u[0] > 1) | (u[1] < 0) | (u[1] > 1) | (u[2] < 0) | (u[2] > 1);
        const unsigned char *__restrict__ table_bytes = reinterpret_cast<const unsigned char *>(P.table);
        // Per level: cell coordinates, fractions and the row of the cell's low corner.  The level's constants come from the LDS copy made
        // at kernel start (two 16-byte reads per level instead of six per-lane selects between kernarg values, which the compiler turns
        // into six per-lane global loads).
        //   pos = u scale + 0.5 >= 0.5 for every point that is not zeroed as out of range: the truncating conversion IS floor, and
        //   v_fract_f32 returns pos - floor(pos), which is exact in fp32 -- the reference's `pos -= (float)pos_grid` (gridencoder.cu:147-151);
        //   rows: cell coordinates <= 2049 and strides <= 2049^2 < 2^24: the low 32 bits of the 24 x 24-bit products are the uint32
        //   products of get_grid_index (gridencoder.cu:66-84), wrap-around included (v_mad_u32_u24, not the quarter-rate v_mul_lo_u32);
        //   `index % hashmap_size` without a division: capped levels have a power-of-two row count (AND); dense levels hold every
        //   (res+1)^3 corner, so an in-range point never wraps.
        struct LevelCell { uint32_t offset, s1, s2, hsize, mask, base; };
        auto level_cell = [&](int li, float (&fr)[3]) {
            const uint4 k0 = s_lv[8 * h + li][0], k1 = s_lv[8 * h + li][1];
            LevelCell c;
            c.offset = k0.x; c.s1 = k0.y; c.s2 = k0.z; c.hsize = k0.w; c.mask = k1.x;
            const float scale = __uint_as_float(k1.y);
            uint32_t pg[3];
            #pragma unroll
            for (int d = 0; d < 3; d++) {
                const float q = u[d] * scale + 0.5f;
                pg[d] = (uint32_t)q;
                fr[d] = __builtin_amdgcn_fractf(q);
            }
            c.base = oob ? 0u : pg[0] + __umul24(pg[1], c.s1) + __umul24(pg[2], c.s2);
            return c;
        };
        // kernel_grid (gridencoder.cu:187-189), scalar_t = at::Half:  results[ch] += w * grid[index + ch]  is
        //   t = Half(w * float(val));  results = Half(float(results) + float(t))
        // -- the float product is converted to Half first (the only `Half += x` takes a Half).  The products come from
        // v_fma_mix_f32 reading the fp16 halves of the gathered word in place (w * val as fma(w, val, -0): the individually
        // rounded fp32 product), one v_cvt_pk_f16_f32 rounds both channels, and the Half + Half sum is ONE v_pk_add_f16:
        // for two fp16 operands the fp16-rounded exact sum equals Half(fp32 sum) (24 >= 2 * 11 + 2 bits: no double-rounding
        // case exists).  4 VALU instructions per corner for both channels.  `corner_bits(idx)` = the half2 of corner idx (bit d set:
        // +1 along dimension d), in the reference's corner order.
        auto interpolate = [&](const float (&fr)[3], auto corner_bits) {
            typedef _Float16 half2v __attribute__((ext_vector_type(2)));
            half2v accv = {(_Float16)0.0f, (_Float16)0.0f};
            #pragma unroll
            for (uint32_t idx = 0; idx < 8; idx++) {
                float w = 1;
                #pragma unroll
                for (uint32_t d = 0; d < 3; d++) w *= (idx & (1u << d)) ? fr[d] : 1 - fr[d];
                const uint32_t bits = corner_bits(idx);
                const half2v t = {(_Float16)mix_mul_lo(w, bits), (_Float16)mix_mul_hi(w, bits)};
                accv = accv + t;
            }
            const uint32_t acc2 = __builtin_bit_cast(uint32_t, accv);
            return oob ? 0u : acc2;
        };
        #pragma unroll
        for (int lb = 0; lb < 8 / kGridBatch; lb++) {
            float pos[kGridBatch][3];
            if constexpr (LAYOUT == kLayoutQuad) {
                // one 16-byte block = the four (x, y) corners of a cell: 2 gathers per level (z and z + 1).  All gathers of the batch are
                // issued back to back (independent, L2 / Infinity Cache latency overlapped) before any is consumed.
                uint4 quads[kGridBatch][2];
                #pragma unroll
                for (int lq = 0; lq < kGridBatch; lq++) {
                    const LevelCell c = level_cell(lb * kGridBatch + lq, pos[lq]);
                    const uint32_t r0 = c.base & c.mask, r1 = (c.base + c.s2) & c.mask;
                    // (uniform 64-bit base + 32-bit byte offset: the host refuses tables of 2^28 blocks or more)
                    __builtin_memcpy(&quads[lq][0], table_bytes + ((c.offset + r0) << 4), 16);
                    __builtin_memcpy(&quads[lq][1], table_bytes + ((c.offset + r1) << 4), 16);
                }
                #pragma unroll
                for (int lq = 0; lq < kGridBatch; lq++) {
                    const int li = lb * kGridBatch + lq;
                    gfw[li >> 2][li & 3] = interpolate(pos[lq], [&](uint32_t idx) {
                        const uint4 &q = quads[lq][idx >> 2];
                        return (idx & 3u) == 0u ? q.x : ((idx & 3u) == 1u ? q.y : ((idx & 3u) == 2u ? q.z : q.w));
                    });
                }
            } else {
                // The x and x+1 corners of a (y, z) corner pair are neighbouring table rows, so one 8-byte gather fetches both:
                // 4 gathers per level instead of 8.
                uint2 pairs[kGridBatch][4];
                uint32_t wrapbits = 0;   // bit 4 lq + c: that gather's x corner is the last row of a capped level (x+1 wraps to row 0)
                #pragma unroll
                for (int lq = 0; lq < kGridBatch; lq++) {
                    const LevelCell lc = level_cell(lb * kGridBatch + lq, pos[lq]);
                    #pragma unroll
                    for (uint32_t c = 0; c < 4; c++) {
                        const uint32_t row0 = (lc.base + ((c & 1u) ? lc.s1 : 0u) + ((c & 2u) ? lc.s2 : 0u)) & lc.mask;
                        // the pair (rl, rl + 1) is always inside the level (PAD: row hsize exists and repeats row 0)
                        const uint32_t rl = PAD ? row0 : min(row0, lc.hsize - 2u);
                        __builtin_memcpy(&pairs[lq][c], table_bytes + ((lc.offset + rl) << 2), 8);   // 4-byte aligned 8-byte gather
                        if (!PAD && row0 > rl) wrapbits |= 1u << (4 * lq + (int)c);
                    }
                }
                // On a capped level the x corner may be the LAST row: it is then the second row of the pair that was fetched and the
                // x+1 corner is row 0.  Patched after every gather of the batch has been issued (a branch per gather would make each
                // wait for its own data); practically never taken (1 row in 2^19).
                if (!PAD && __builtin_expect(wrapbits != 0u, 0)) {
                    #pragma unroll
                    for (int lq = 0; lq < kGridBatch; lq++) {
                        #pragma unroll
                        for (uint32_t c = 0; c < 4; c++) {
                            if ((wrapbits >> (4 * lq + (int)c)) & 1u) {
                                const int li = lb * kGridBatch + lq;   // wrapped: row0 == mask, so the x+1 corner is row 0 of the level
                                const uint32_t offset = s_lv[8 * h + li][0].x;
                                pairs[lq][c].x = pairs[lq][c].y;
                                pairs[lq][c].y = *reinterpret_cast<const uint32_t *>(table_bytes + (offset << 2));
                            }
                        }
                    }
                }
                #pragma unroll
                for (int lq = 0; lq < kGridBatch; lq++) {
                    const int li = lb * kGridBatch + lq;
                    gfw[li >> 2][li & 3] = interpolate(pos[lq], [&](uint32_t idx) { return (idx & 1u) ? pairs[lq][idx >> 1].y : pairs[lq][idx >> 1].x; });
                }
            }
            // keep the batches apart: hoisting the next batch's index math and gathers above this batch's interpolation doubles the
            // live registers (spills under the 128-VGPR cap of the throughput variant)
            if (lb + 1 < 8 / kGridBatch) __builtin_amdgcn_sched_barrier(0);
        }
    }

    #pragma unroll
    for (int q = 0; q < 2; q++) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        gf[q] = __builtin_bit_cast(half8, (u32x4){gfw[q][0], gfw[q][1], gfw[q][2], gfw[q][3]});
    }
    FSTAMP(12);
    // ---------------- sigma net: 32 -> 64 (ReLU) -> 16 ----------------
    f32x16 s0[2];
    #pragma unroll
    for (int mt = 0; mt < 2; mt++) {
        #pragma unroll
        for (int r = 0; r < 16; r++) s0[mt][r] = 0.0f;
        #pragma unroll
        for (int ks = 0; ks < 2; ks++) s0[mt] = mfma(lds_frag(tail, tS0 + mt * 2 + ks, lane), gf[ks], s0[mt]);
    }
    half8 sf[4];
    acc_to_frags<true>(s0[0], sf[0], sf[1]);
    acc_to_frags<true>(s0[1], sf[2], sf[3]);
    f32x16 hv;
    #pragma unroll
    for (int r = 0; r < 16; r++) hv[r] = 0.0f;
    #pragma unroll
    for (int ks = 0; ks < 4; ks++) hv = mfma(lds_frag(tail, tS1 + ks, lane), sf[ks], hv);
    // h[0] (lane-half 0, register 0) is the density logit; trunc_exp = exp in fp32 of the fp16 value
    // (v_exp_f32 on h log2(e): 1 ulp of the hardware exponential plus |h| 2^-24 from the product -- the reference's own operators use the
    //  fast intrinsics of their platform here (__expf), and sigma feeds a compositing sum that is compared at 1e-4 / fp16 distance)
    const float sigma = P.density_scale * __builtin_amdgcn_exp2f(round_h(hv[0]) * 1.4426950408889634f);
    if constexpr (CELLS) {
        if (h == 0 && valid) P.sigmas[p] = sigma;   // duplicates in the list: any one of them wins, as with tmp_grid[indices] = sigmas
        return;
    }
    FSTAMP(13);
    // ---------------- colour net: [SH(16) ++ geo_feat(15)] -> 64 -> 64 -> 3 ----------------
    half8 cf[2], dummy;
    acc_to_frags<false>(hv, cf[0], dummy);  // registers 0..7 of every lane = h[0..15]; column of h[0] is zero in the packed weights
    {
        float sh[16];
        float *nul = nullptr;
        sdn_sh::sh_eval<4, false>(d0, d1, d2, sh, nul, nul, nul);
        #pragma unroll
        for (int j = 0; j < 8; j++) {
            float lo = sh[j], hi = sh[8 + j];
            // pin both candidates in VGPRs: otherwise the select of two array elements becomes one dynamically indexed
            // load and the whole array is demoted to LDS
            asm volatile("" : "+v"(lo), "+v"(hi));
            cf[1][j] = (_Float16)(h ? hi : lo);
        }
    }
    f32x16 c0[2];
    #pragma unroll
    for (int mt = 0; mt < 2; mt++) {
        #pragma unroll
        for (int r = 0; r < 16; r++) c0[mt][r] = 0.0f;
        #pragma unroll
        for (int ks = 0; ks < 2; ks++) c0[mt] = mfma(lds_frag(tail, tC0 + mt * 2 + ks, lane), cf[ks], c0[mt]);
    }
    half8 c1f[4];
    acc_to_frags<true>(c0[0], c1f[0], c1f[1]);
    acc_to_frags<true>(c0[1], c1f[2], c1f[3]);
    f32x16 c1[2];
    #pragma unroll
    for (int mt = 0; mt < 2; mt++) {
        #pragma unroll
        for (int r = 0; r < 16; r++) c1[mt][r] = 0.0f;
        #pragma unroll
        for (int ks = 0; ks < 4; ks++) c1[mt] = mfma(lds_frag(tail, tC1 + mt * 4 + ks, lane), c1f[ks], c1[mt]);
    }
    half8 c2f[4];
    acc_to_frags<true>(c1[0], c2f[0], c2f[1]);
    acc_to_frags<true>(c1[1], c2f[2], c2f[3]);
    f32x16 co;
    #pragma unroll
    for (int r = 0; r < 16; r++) co[r] = 0.0f;
    #pragma unroll
    for (int ks = 0; ks < 4; ks++) co = mfma(lds_frag(tail, tC2 + ks, lane), c2f[ks], co);
    FSTAMP(14);
    if (h == 0 && valid) {
        P.sigmas[p] = sigma;
        #pragma unroll
        for (int c = 0; c < 3; c++) {
            const float logit = round_h(co[c]);
            // torch.sigmoid on fp16: fp32 math, fp16 result
            P.rgbs[(size_t)p * 3 + c] = round_h(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(logit * -1.4426950408889634f)));
        }
    }
    FSTAMP(15);
}

// The reference sizes every level to a multiple of 8 rows (grid.py:124).  The two derived layouts are self-describing through the
// offsets they come with: a PADDED table (one extra row per level repeating the level's row 0) has level sizes == 1 mod 8, a QUAD table
// (one 16-byte block per row, two unused blocks behind every level) == 2 mod 8 (dnerf_amd/fused.py builds both).
int table_layout(const int32_t *offsets_host) {
    uint32_t seen = 0;
    for (uint32_t l = 0; l < 16; l++) {
        const uint32_t r = (uint32_t)(offsets_host[l + 1] - offsets_host[l]) & 7u;
        seen |= 1u << r;
    }
    return seen == 2u ? kLayoutPad : (seen == 4u ? kLayoutQuad : kLayoutRef);
}
bool table_is_padded(const int32_t *offsets_host) { return table_layout(offsets_host) == kLayoutPad; }

// 1 / v when v is a (normal) power of two -- x / v == x * (1 / v) exactly for every x -- else 0
float exact_reciprocal(float v) {
    int e = 0;
    if (!(v > 0.0f) || !isfinite(v) || frexpf(v, &e) != 0.5f || e < -100 || e > 100) return 0.0f;
    return 1.0f / v;
}

int fill_tiled_levels(TiledLevels &lv, const int32_t *offsets_host, float S, uint32_t H) {
    const int layout = table_layout(offsets_host);
    for (uint32_t l = 0; l < 16; l++) {
        const uint32_t hsize = (uint32_t)(offsets_host[l + 1] - offsets_host[l]) - (layout == kLayoutPad ? 1u : (layout == kLayoutQuad ? 2u : 0u));
        if (hsize == 0 || (uint32_t)offsets_host[16] >= (1u << 28)) return SDN_E_BADARG;   // (32-bit byte offsets into the table)
        const float scale = exp2f((float)l * S) * (float)H - 1.0f;  // gridencoder.cu:138
        const uint32_t res = (uint32_t)ceil((double)scale) + 1;      // gridencoder.cu:139
        // get_grid_index (gridencoder.cu:66-84) for D = 3, align_corners = false, gridtype = tiled:
        //   stride = 1; for d: if (stride <= hsize) { index += pos[d] * stride; stride *= res + 1; }
        uint32_t stride = 1, s[3] = {0, 0, 0};
        for (int d = 0; d < 3; d++) {
            if (stride <= hsize) { s[d] = stride; stride *= (res + 1); }
        }
        if (s[0] != 1) return SDN_E_BADARG;
        lv.offset[l] = (uint32_t)offsets_host[l];
        lv.s1[l] = s[1];
        lv.s2[l] = s[2];
        lv.hsize[l] = hsize;
        const bool pow2 = (hsize & (hsize - 1)) == 0;
        // a non-power-of-two level must be dense (every corner has its own row), otherwise the AND/min form would be wrong
        if (!pow2 && (s[1] == 0 || s[2] == 0 || (uint64_t)(res + 1) * (res + 1) * (res + 1) > hsize)) return SDN_E_UNSUPPORTED;
        lv.mask[l] = pow2 ? hsize - 1 : 0xFFFFFFFFu;
        lv.scale[l] = scale;
    }
    return 0;
}

#include "field_pp.inc"

// Quad table: block (offset_q[l] + r) of level l = rows {r, r + 1, r + s1, r + s1 + 1} of the level, each mod its row count, as fp16 pairs
// (the corners (x, y), (x+1, y), (x, y+1), (x+1, y+1) of the cell whose low corner is row r: get_grid_index adds 1 / s1 per step and
// takes the sum mod the row count, gridencoder.cu:66-84).  T = float or __half (rounded to fp16 as grid.py:43-44's `.half()` does).
struct QuadLevels { uint32_t src[17]; uint32_t s1[16]; };
template <typename T>
__global__ void k_build_quad_table(const T *__restrict__ emb, QuadLevels q, uint4 *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;   // source row
    if (i >= q.src[16]) return;
    uint32_t l = 0;
    #pragma unroll
    for (uint32_t k = 1; k < 16; k++) l += (i >= q.src[k]) ? 1u : 0u;
    const uint32_t a = q.src[l], hs = q.src[l + 1] - a, r = i - a, s1 = q.s1[l];
    auto row = [&](uint32_t rr) {
        const T *p = emb + (size_t)(a + rr % hs) * 2;
        const __half2 v = __halves2half2(__float2half_rn((float)p[0]), __float2half_rn((float)p[1]));
        return __builtin_bit_cast(uint32_t, v);
    };
    out[i + 2u * l] = make_uint4(row(r), row(r + 1u), row(r + s1), row(r + s1 + 1u));
}

}  // namespace

namespace sdn_int {

static int g_field_pp = -1;   // 1: large launches take the persistent ping-pong kernel (default), 0: never; -1: read SDN_FIELD_PP
static int g_field_pp_wgs = 0;   // workgroups of a persistent launch; 0 = one per CU

// launch used by both the C entry point and the device-driven render loop (render.hip)
int field_forward_f16(const float *xyzs, const float *dirs, const uint32_t *live_idx, const uint32_t *live_count, const int32_t *state,
                      uint32_t M, const void *weights, const float *bias0, const void *table, const int32_t *offsets_host, float S,
                      uint32_t H, float bound, float density_scale, int zero_deform, float *sigmas, float *rgbs, uint32_t expect_points,
                      const uint8_t *slot_frame, uint32_t n_frames, hipStream_t st) {
    TiledLevels lv;
    int rc = fill_tiled_levels(lv, offsets_host, S, H);
    if (rc) return rc;
    FieldArgs a;
    a.xyzs = xyzs; a.dirs = dirs; a.live_idx = live_idx; a.live_count = live_count; a.state = state; a.M = M;
    a.weights = (const unsigned char *)weights; a.bias0 = bias0; a.table = (const __half *)table;
    a.sigmas = sigmas; a.rgbs = rgbs; a.bound = bound; a.density_scale = density_scale; a.zero_deform = zero_deform;
    a.inv_2bound = exact_reciprocal(2 * bound);
    a.slot_frame = slot_frame; a.n_frames = slot_frame ? (n_frames > 16u ? 16u : n_frames) : 1u;
    a.cell_noise = nullptr; a.cell_seed = 0; a.cell_inv = a.cell_span = a.cell_half = 0;
    const uint32_t wgs = sdn_div_up(M, (uint32_t)kPointsPerWG);
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
    }
    // expect_points (0 = unknown): the caller's estimate of the live points when M is only a loose bound of them
    const uint32_t busy = expect_points ? sdn_div_up(expect_points < M ? expect_points : M, (uint32_t)kPointsPerWG) : wgs;
    // latency variant (one workgroup per CU, deep LDS read-ahead) while the launch is at most one workgroup per CU.  (Measured,
    // profiles/r02_field_latency_rounds.txt: running 2+ ROUNDS of lone workgroups instead of one round of co-resident pairs is
    // slower -- 57.6 vs 51.9 us at 126 976 points.)
    bool small = busy <= (uint32_t)cus;
    {   // SDN_FIELD_VARIANT=throughput / latency pins the variant (measurements only)
        static int pin = -1;
        if (pin < 0) {
            const char *e = getenv("SDN_FIELD_VARIANT");
            pin = !e ? 0 : (e[0] == 't' ? 1 : (e[0] == 'l' ? 2 : 0));
        }
        if (pin == 1) small = false;
        if (pin == 2) small = true;
    }
    a.stagger_cycles = 0; a.stagger_lo = a.stagger_hi = 0; a.pp_soft = 0;
    if (!small) {
        static int stag = -1;
        if (stag < 0) {
            const char *e = getenv("SDN_FIELD_STAGGER_CYCLES");
            stag = e ? atoi(e) : 0;
        }
        a.stagger_cycles = (uint32_t)stag; a.stagger_lo = (uint32_t)cus; a.stagger_hi = 2u * (uint32_t)cus;
    }
    const int layout = table_layout(offsets_host);
    // persistent ping-pong form (field_pp.inc): one 16-wave workgroup per CU walking tile pairs, for launches of at least two rounds of
    // the CUs (SDN_FIELD_PP=0 keeps every launch on the one-tile-per-workgroup kernels: measurements only)
    if (g_field_pp < 0) {
        const char *e = getenv("SDN_FIELD_PP");
        g_field_pp = e ? atoi(e) : 1;
    }
    static int pp_min_tiles_per_cu = -1;      // launches of at least this many 256-point tiles per CU take the persistent kernel
    if (pp_min_tiles_per_cu < 0) {
        const char *e = getenv("SDN_FIELD_PP_MIN_TILES");
        pp_min_tiles_per_cu = e ? atoi(e) : 4;
    }
    if (g_field_pp && layout == kLayoutQuad && busy >= (uint32_t)pp_min_tiles_per_cu * (uint32_t)cus && a.n_frames <= kPPMaxFrames && M < (1u << 28)) {
        const uint32_t pairs = sdn_div_up(M, kPPTile);     // (tiles: the unit the kernel deals out)
        // (SDN_FIELD_PP_CUS: workgroups of the persistent launch, default one per CU -- fewer leave CUs to the other frames' small kernels
        //  of a pipelined stream, which cannot share a CU with a 16-wave, 156-KiB workgroup: a measurement knob)
        static int env_cus = -1;
        if (env_cus < 0) {
            const char *e = getenv("SDN_FIELD_PP_CUS");
            env_cus = e ? atoi(e) : 0;
        }
        int pp_cus = env_cus > 0 ? env_cus : (g_field_pp_wgs > 0 ? g_field_pp_wgs : cus);
        if (pp_cus > cus) pp_cus = cus;
        // The launch covers every CU; the kernel, which knows the live count, keeps G of the workgroups (the others leave at once):
        // the fewest that finish in as many rounds as `pp_cus` would need -- or, when one workgroup per CU saves a whole round over
        // `pp_cus`, the fewest that finish in that many (SDN_FIELD_PP_BALANCE=0: exactly pp_cus, as before; 2: never more than pp_cus)
        static int balance = -1;
        if (balance < 0) {
            const char *e = getenv("SDN_FIELD_PP_BALANCE");
            balance = e ? atoi(e) : 0;
        }
        const uint32_t grid = balance == 1 ? (uint32_t)cus : (uint32_t)pp_cus;
        a.pp_soft = (uint32_t)pp_cus | (balance == 0 ? 0x80000000u : (balance == 2 ? 0x40000000u : 0u));
        hipLaunchKernelGGL(k_field_pp_f16, dim3(pairs < grid ? pairs : grid), dim3(64 * kPPWaves), 0, st, a, lv);
        return sdn_launch_status();
    }
    if (layout == kLayoutQuad) {
        if (small) hipLaunchKernelGGL((k_field_f16<2, 8, false, kLayoutQuad>), dim3(wgs), dim3(64 * kWaves), 0, st, a, lv);
        else hipLaunchKernelGGL((k_field_f16<4, 2, false, kLayoutQuad>), dim3(wgs), dim3(64 * kWaves), 0, st, a, lv);
    } else if (layout == kLayoutPad) {
        if (small) hipLaunchKernelGGL((k_field_f16<2, 8, false, kLayoutPad>), dim3(wgs), dim3(64 * kWaves), 0, st, a, lv);
        else hipLaunchKernelGGL((k_field_f16<4, 2, false, kLayoutPad>), dim3(wgs), dim3(64 * kWaves), 0, st, a, lv);
    } else {
        // reference-layout table (the public entry point with a caller's own table): one variant only -- with the wrap bookkeeping
        // the throughput variant does not fit 128 VGPRs without spilling; the derived layouts are the product path
        hipLaunchKernelGGL((k_field_f16<2, 8, false, kLayoutRef>), dim3(wgs), dim3(64 * kWaves), 0, st, a, lv);
    }
    return sdn_launch_status();
}

// sigma * density_scale of jittered occupancy-grid cell centres -> tmp_grid slice (density.hip drives the whole update)
int field_cells_f16(const int32_t *cells, const uint32_t *cell_count, uint32_t n, const float *noise, uint32_t seed, uint32_t grid_size,
                    float cas_bound, const void *weights, const float *bias0, const void *table, const int32_t *offsets_host, float S,
                    uint32_t H, float bound, float density_scale, int zero_deform, float *tmp_slice, hipStream_t st) {
    TiledLevels lv;
    int rc = fill_tiled_levels(lv, offsets_host, S, H);
    if (rc) return rc;
    FieldArgs a;
    a.xyzs = nullptr; a.dirs = nullptr; a.live_idx = (const uint32_t *)cells; a.live_count = cell_count; a.state = nullptr; a.M = n;
    a.weights = (const unsigned char *)weights; a.bias0 = bias0; a.table = (const __half *)table;
    a.sigmas = tmp_slice; a.rgbs = nullptr; a.bound = bound; a.density_scale = density_scale; a.zero_deform = zero_deform ? 1 : 0;
    a.inv_2bound = exact_reciprocal(2 * bound);
    a.slot_frame = nullptr;
    a.cell_noise = noise; a.cell_seed = seed;
    a.stagger_cycles = 0; a.stagger_lo = a.stagger_hi = 0; a.n_frames = 1; a.pp_soft = 0;
    const float half_grid = cas_bound / (float)grid_size;
    a.cell_inv = 1.0f / (float)(grid_size - 1); a.cell_span = cas_bound - half_grid; a.cell_half = half_grid;
    const uint32_t wgs = sdn_div_up(n, (uint32_t)kPointsPerWG);
    const int layout = table_layout(offsets_host);
    if (layout == kLayoutRef) return SDN_E_UNSUPPORTED;   // the density query is only built for the derived layouts
    if (layout == kLayoutQuad) {
        if (wgs <= 256u) hipLaunchKernelGGL((k_field_f16<2, 8, true, kLayoutQuad>), dim3(wgs), dim3(64 * kWaves), 0, st, a, lv);
        else hipLaunchKernelGGL((k_field_f16<4, 2, true, kLayoutQuad>), dim3(wgs), dim3(64 * kWaves), 0, st, a, lv);
    } else {
        if (wgs <= 256u) hipLaunchKernelGGL((k_field_f16<2, 8, true, kLayoutPad>), dim3(wgs), dim3(64 * kWaves), 0, st, a, lv);
        else hipLaunchKernelGGL((k_field_f16<4, 2, true, kLayoutPad>), dim3(wgs), dim3(64 * kWaves), 0, st, a, lv);
    }
    return sdn_launch_status();
}

}  // namespace sdn_int

extern "C" {

uint32_t sdn_field_weight_blocks(void) { return (uint32_t)kBlkTotal; }

// Kernel selection for large launches of the fused field network: 1 = persistent ping-pong kernel (the default), 0 = one tile per
// workgroup for every launch (the two produce identical bits; tests and A/B measurements switch here), -1 = back to SDN_FIELD_PP / default.
void sdn_field_select_kernel(int persistent) { sdn_int::g_field_pp = persistent; }
// Workgroups of a persistent launch (0 = one per CU, the default).  A stream of frames in several loop contexts leaves an eighth of the
// CUs to the other frames' marchers and compositors -- which cannot share a CU with a 16-wave, 156-KiB workgroup and otherwise wait
// for a whole persistent launch to end: 0.410 -> 0.398 ms per frame with 224 of 256 (profiles/r04_field_pp_kernel.txt).
void sdn_field_persistent_workgroups(int n) { sdn_int::g_field_pp_wgs = n > 0 ? n : 0; }

// Builds the fused kernel's QUAD table (16 bytes per row, see kLayoutQuad) from embeddings in the reference layout.
//   embeddings [ref_offsets_host[16], 2] of dtype (SDN_F32 / SDN_F16), ref_offsets_host [17] the reference's level offsets (grid.py:118-127);
//   out: (ref_offsets_host[16] + 32) blocks of 16 bytes -- level l starts at block ref_offsets_host[l] + 2 l (pass those 17 values as
//   `offsets_host` to the field entry points: level sizes == 2 mod 8 announce the layout).
int sdn_field_build_quad_table(const void *embeddings, int dtype, const int32_t *ref_offsets_host, float S, uint32_t H, void *out, void *stream) {
    if (!embeddings || !ref_offsets_host || !out || ((uintptr_t)out & 15u) != 0) return SDN_E_BADARG;
    TiledLevels lv;
    int rc = fill_tiled_levels(lv, ref_offsets_host, S, H);   // (reference offsets: level sizes are multiples of 8)
    if (rc) return rc;
    QuadLevels q;
    for (int l = 0; l < 17; l++) q.src[l] = (uint32_t)ref_offsets_host[l];
    for (int l = 0; l < 16; l++) q.s1[l] = lv.s1[l];
    const uint32_t rows = q.src[16];
    if (rows == 0) return SDN_E_BADARG;
    if (hipMemsetAsync(out, 0, ((size_t)rows + 32) * 16, (hipStream_t)stream) != hipSuccess) return sdn_launch_status();
    if (dtype == SDN_F32)
        hipLaunchKernelGGL(k_build_quad_table<float>, dim3(sdn_div_up(rows, 256u)), dim3(256), 0, (hipStream_t)stream, (const float *)embeddings, q, (uint4 *)out);
    else if (dtype == SDN_F16)
        hipLaunchKernelGGL(k_build_quad_table<__half>, dim3(sdn_div_up(rows, 256u)), dim3(256), 0, (hipStream_t)stream, (const __half *)embeddings, q, (uint4 *)out);
    else
        return SDN_E_BADARG;
    return sdn_launch_status();
}

#ifdef SDN_STAMPS
// diagnostic build only: the persistent kernel's slot stamps (g_pp_stamps: 256 workgroups x 2 sets x 128 words), cleared behind the copy
int sdn_debug_pp_stamps(unsigned long long *out) {
    if (!out) return SDN_E_BADARG;
    if (hipDeviceSynchronize() != hipSuccess) return sdn_launch_status();
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pp_stamps), sizeof(unsigned long long) * 256 * 2 * 128) != hipSuccess) return sdn_launch_status();
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_pp_stamps)) != hipSuccess) return sdn_launch_status();
    return (int)hipMemset(p, 0, sizeof(unsigned long long) * 256 * 2 * 128);
}
// diagnostic build only: copies out and clears the field kernel's stamp sums (see g_field_stamps)
// (out: sdn_debug_field_stamp_words() 64-bit words; the buffer is cleared behind the copy)
uint32_t sdn_debug_field_stamp_words(void) { return kStampWGs * 2 * 32; }
int sdn_debug_field_stamps(unsigned long long *out) {
    if (!out) return SDN_E_BADARG;
    if (hipDeviceSynchronize() != hipSuccess) return sdn_launch_status();
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_field_stamps), sizeof(unsigned long long) * kStampWGs * 2 * 32) != hipSuccess) return sdn_launch_status();
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_field_stamps)) != hipSuccess) return sdn_launch_status();
    return (int)hipMemset(p, 0, sizeof(unsigned long long) * kStampWGs * 2 * 32);
}
#endif

// Fused field forward, `-O` numerics (fp16 MLPs / table, fp32 encoders).  weights: packed by dnerf_amd/fused.py
// (sdn_field_weight_blocks() KiB); bias0 [128] f32; table fp16 [rows,2]; offsets_host [17] (16 levels).
int sdn_field_forward_f16(const float *xyzs, const float *dirs, const uint32_t *live_idx, const uint32_t *live_count, uint32_t M,
                          const void *weights, const float *bias0, const void *table, const int32_t *offsets_host, float S, uint32_t H,
                          float bound, float density_scale, int zero_deform, float *sigmas, float *rgbs, void *stream) {
    if (M == 0) return 0;
    if (!xyzs || !dirs || !weights || !bias0 || !table || !offsets_host || !sigmas || !rgbs) return SDN_E_BADARG;
    if ((live_idx == nullptr) != (live_count == nullptr)) return SDN_E_BADARG;
    if (((uintptr_t)weights & 15u) != 0 || ((uintptr_t)table & 3u) != 0) return SDN_E_BADARG;
    return sdn_int::field_forward_f16(xyzs, dirs, live_idx, live_count, nullptr, M, weights, bias0, table, offsets_host, S, H, bound,
                                      density_scale, zero_deform ? 1 : 0, sigmas, rgbs, 0u, nullptr, 1u, (hipStream_t)stream);
}

}  // extern "C"
