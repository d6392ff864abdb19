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
//   * one wave = 32 sample points = the N dimension of v_mfma_f32_32x32x16_f16; weights are the A operand
//     (M = output features), activations the B operand (K = input features).  The accumulator of layer i
//     (feature rows in registers, point on the lane) converts in place (ReLU, cvt_pk_f16) into the B operand
//     of layer i+1 -- no LDS, no barriers, no cross-lane traffic between layers.  The k-order this implies
//     (element j of lane-half h <-> feature 16s + 8(j>>2) + 4h + (j&3)) is baked into the host-side weight
//     packing (dnerf_amd/fused.py), as are the first layer's freq-feature order, the grid-feature order and
//     the SH / geo_feat order of the colour net.
//   * the time encoding is the same for every point: its contribution W0[:,63:76] . enc(t) is a per-frame
//     bias vector (computed on the host) loaded as the initial accumulator of the first layer.
//   * weights (240 fragments of 1 KiB, fragment order) are streamed from L2 with one 16-byte load per lane
//     per MFMA; every wave of the chip reads the same 240 KiB.
//   * the 128 table gathers per point are split over the two lane-halves (levels 0-7 / 8-15): 64 independent
//     4-byte loads per lane, issued per level before the blend.
//   * optional live-sample index list: only slots that hold a sample are evaluated (the reference evaluates
//     the network on every padded slot).
#include <math.h>

#include "sdn_common.h"
#include "grid_common.h"
#include "sh_eval.h"

namespace {

using namespace sdn_grid;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBlkD0 = 0;                 // 4 Mt x 4 ks
constexpr int kBlkD1 = kBlkD0 + 16;       // 6 layers x (4 Mt x 8 ks)
constexpr int kBlkD7 = kBlkD1 + 6 * 32;   // 1 Mt x 8 ks
constexpr int kBlkS0 = kBlkD7 + 8;        // 2 Mt x 2 ks
constexpr int kBlkS1 = kBlkS0 + 4;        // 1 Mt x 4 ks
constexpr int kBlkC0 = kBlkS1 + 4;        // 2 Mt x 2 ks
constexpr int kBlkC1 = kBlkC0 + 4;        // 2 Mt x 4 ks
constexpr int kBlkC2 = kBlkC1 + 8;        // 1 Mt x 4 ks
constexpr int kBlkTotal = kBlkC2 + 4;     // 240

struct FieldArgs {
    const float *xyzs;        // [M,3]
    const float *dirs;        // [M,3]
    const uint32_t *live_idx; // [<=M] slot indices to evaluate, or nullptr = all M slots
    const uint32_t *live_count;
    uint32_t M;
    const half8 *weights;     // kBlkTotal x 64 lanes x 8 halfs
    const float *bias0;       // [128] time-encoding contribution to the first deform layer
    const __half *table;      // grid embeddings, fp16 [rows, 2]
    float *sigmas;            // [M]
    float *rgbs;              // [M,3]
    float bound;
    float density_scale;
    int zero_deform;          // t == 0: canonical frame (dnerf/network.py:140-141)
};

__device__ __forceinline__ f32x16 mfma(half8 a, half8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

__device__ __forceinline__ half8 ldw(const half8 *w, int blk, uint32_t lane) { return w[(size_t)blk * 64 + lane]; }

// accumulator tile -> the two B fragments (k-steps) it provides to the next layer
template <bool RELU>
__device__ __forceinline__ void acc_to_frags(const f32x16 &acc, half8 &f0, half8 &f1) {
    #pragma unroll
    for (int j = 0; j < 8; j++) {
        float a = acc[j], b = acc[8 + j];
        if (RELU) { a = fmaxf(a, 0.0f); b = fmaxf(b, 0.0f); }
        f0[j] = (_Float16)a;
        f1[j] = (_Float16)b;
    }
}

__device__ __forceinline__ float round_h(float v) { return (float)(_Float16)v; }

// sin(a) for |a| < ~1e4 to ~1e-7 absolute: 3-term Cody-Waite reduction by pi (explicit FMAs: this file is built with
// -ffp-contract=off) + odd degree-9 minimax polynomial on [-pi/2, pi/2].  The standalone freq_encode kernel uses OCML
// sinf (<= 1 ulp); the two agree to ~1e-7, far below the fp16 rounding the features get as MFMA operands.
__device__ __forceinline__ float fast_sin(float a) {
    const float k = rintf(a * 0.31830988618379067f);
    float r = __builtin_fmaf(-k, 3.140625f, a);                  // pi_hi  (8 significant bits: k * pi_hi exact)
    r = __builtin_fmaf(-k, 9.67502593994140625e-4f, r);        // pi_mid
    r = __builtin_fmaf(-k, 1.509957990978376e-7f, r);          // pi_lo
    const float r2 = r * r;
    float p = __builtin_fmaf(r2, 2.6083159809786593e-6f, -1.9810690719168633e-4f);
    p = __builtin_fmaf(p, r2, 8.3330785855650902e-3f);
    p = __builtin_fmaf(p, r2, -1.6666659712791443e-1f);
    const float s = __builtin_fmaf(r * r2, p, r);
    const int ki = (int)k;
    return __int_as_float(__float_as_int(s) ^ ((ki & 1) << 31));
}

__global__ void __launch_bounds__(256) k_field_f16(FieldArgs P, LevelParams lp) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t count = P.live_idx ? *P.live_count : P.M;
    const uint32_t base = gw * 32u;
    if (base >= count) return;  // wave-uniform
    const uint32_t n = lane & 31u, h = lane >> 5;
    const uint32_t i = base + n;
    const bool valid = i < count;
    const uint32_t ii = valid ? i : (count - 1);
    const uint32_t p = P.live_idx ? P.live_idx[ii] : ii;
    const half8 *__restrict__ W = P.weights;

    const float x0 = P.xyzs[(size_t)p * 3], x1 = P.xyzs[(size_t)p * 3 + 1], x2 = P.xyzs[(size_t)p * 3 + 2];
    const float d0 = P.dirs[(size_t)p * 3], d1 = P.dirs[(size_t)p * 3 + 1], d2 = P.dirs[(size_t)p * 3 + 2];

    // ---------------- deform layer 0: freq features as B fragments ----------------
    // lane-half h owns (freq, dim) pairs 15h .. 15h+14 (sin and cos) plus x0,x1 (h = 0) / x2,pad (h = 1)
    half8 bf[8];
    {
        const float xs[3] = {x0, x1, x2};
        const float fscale = h ? 32.0f : 1.0f;  // pair index 15h + q/2  ->  freq = 5h + (q/2)/3
        #pragma unroll
        for (int s = 0; s < 4; s++) {
            #pragma unroll
            for (int j = 0; j < 8; j++) {
                const int q = s * 8 + j;
                float v;
                if (q < 30) {
                    const int pr = q >> 1, f = pr / 3, dd = pr % 3;
                    // kernel_freq (freqencoder.cu:52-56): sin(x * 2^f + (col % 2) * pi/2), same float ops as encoders.hip
                    const float arg = scalbnf(xs[dd], f) * fscale + (float)(q & 1) * (3.141592653589793f / 2);
                    v = fast_sin(arg);
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
    #pragma unroll
    for (int mt = 0; mt < 4; mt++) {
        #pragma unroll
        for (int r = 0; r < 16; r++) acc[mt][r] = P.bias0[32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h];
    }
    #pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        #pragma unroll
        for (int mt = 0; mt < 4; mt++) acc[mt] = mfma(ldw(W, kBlkD0 + mt * 4 + ks, lane), bf[ks], acc[mt]);
    }

    // ---------------- deform layers 1..6 (128 -> 128, ReLU) ----------------
    for (int l = 0; l < 6; l++) {
        #pragma unroll
        for (int t = 0; t < 4; t++) acc_to_frags<true>(acc[t], bf[2 * t], bf[2 * t + 1]);
        #pragma unroll
        for (int mt = 0; mt < 4; mt++) {
            #pragma unroll
            for (int r = 0; r < 16; r++) acc[mt][r] = 0.0f;
        }
        const int blk = kBlkD1 + l * 32;
        #pragma unroll
        for (int ks = 0; ks < 8; ks++) {
            #pragma unroll
            for (int mt = 0; mt < 4; mt++) acc[mt] = mfma(ldw(W, blk + mt * 8 + ks, lane), bf[ks], acc[mt]);
        }
    }
    // ---------------- deform layer 7 (128 -> 3) ----------------
    float u[3];
    {
        #pragma unroll
        for (int t = 0; t < 4; t++) acc_to_frags<true>(acc[t], bf[2 * t], bf[2 * t + 1]);
        f32x16 o;
        #pragma unroll
        for (int r = 0; r < 16; r++) o[r] = 0.0f;
        #pragma unroll
        for (int ks = 0; ks < 8; ks++) o = mfma(ldw(W, kBlkD7 + ks, lane), bf[ks], o);
        // rows 0..2 = registers 0..2 of lane-half 0; broadcast to both halves
        float df[3];
        #pragma unroll
        for (int c = 0; c < 3; c++) df[c] = __shfl(round_h(o[c]), (int)n, 64);
        const float xs[3] = {x0, x1, x2};
        #pragma unroll
        for (int c = 0; c < 3; c++) {
            const float xd = P.zero_deform ? xs[c] : xs[c] + df[c];
            u[c] = (xd + P.bound) / (2 * P.bound);  // GridEncoder.forward (grid.py:149)
        }
    }

    // ---------------- grid encode: lane-half h evaluates levels 8h .. 8h+7 ----------------
    half8 gf[2];
    {
        const bool oob = (u[0] < 0) | (u[0] > 1) | (u[1] < 0) | (u[1] > 1) | (u[2] < 0) | (u[2] > 1);
        #pragma unroll
        for (int li = 0; li < 8; li++) {
            const uint32_t offset = h ? lp.offset[8 + li] : lp.offset[li];
            const uint32_t hsize = h ? lp.hashmap_size[8 + li] : lp.hashmap_size[li];
            const float scale = h ? lp.scale[8 + li] : lp.scale[li];
            const uint32_t res = h ? lp.resolution[8 + li] : lp.resolution[li];
            const __half2 *__restrict__ tab = reinterpret_cast<const __half2 *>(P.table) + offset;
            float pos[3];
            uint32_t pg[3];
            #pragma unroll
            for (int d = 0; d < 3; d++) {
                pos[d] = u[d] * scale + 0.5f;
                pg[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pg[d];
            }
            float2 vals[8];
            float wgt[8];
            #pragma unroll
            for (uint32_t idx = 0; idx < 8; idx++) {
                float w = 1;
                uint32_t pgl[3];
                #pragma unroll
                for (uint32_t d = 0; d < 3; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pgl[d] = pg[d]; }
                    else { w *= pos[d]; pgl[d] = pg[d] + 1; }
                }
                wgt[idx] = w;
                const uint32_t row = oob ? 0u : grid_index<3, 1>(1u, false, hsize, res, pgl);
                vals[idx] = __half22float2(tab[row]);
            }
            float r0 = 0, r1 = 0;
            #pragma unroll
            for (int idx = 0; idx < 8; idx++) {  // kernel_grid: half += float * half, rounded to half each step
                r0 = round_h(r0 + wgt[idx] * vals[idx].x);
                r1 = round_h(r1 + wgt[idx] * vals[idx].y);
            }
            if (oob) { r0 = 0; r1 = 0; }
            gf[li >> 2][2 * (li & 3)] = (_Float16)r0;
            gf[li >> 2][2 * (li & 3) + 1] = (_Float16)r1;
        }
    }

    // ---------------- sigma net: 32 -> 64 (ReLU) -> 16 ----------------
    f32x16 s0[2];
    #pragma unroll
    for (int mt = 0; mt < 2; mt++) {
        #pragma unroll
        for (int r = 0; r < 16; r++) s0[mt][r] = 0.0f;
        #pragma unroll
        for (int ks = 0; ks < 2; ks++) s0[mt] = mfma(ldw(W, kBlkS0 + mt * 2 + ks, lane), gf[ks], s0[mt]);
    }
    half8 sf[4];
    acc_to_frags<true>(s0[0], sf[0], sf[1]);
    acc_to_frags<true>(s0[1], sf[2], sf[3]);
    f32x16 hv;
    #pragma unroll
    for (int r = 0; r < 16; r++) hv[r] = 0.0f;
    #pragma unroll
    for (int ks = 0; ks < 4; ks++) hv = mfma(ldw(W, kBlkS1 + ks, lane), sf[ks], hv);
    // h[0] (lane-half 0, register 0) is the density logit; trunc_exp = exp in fp32 of the fp16 value
    const float sigma = P.density_scale * expf(round_h(hv[0]));

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
        for (int ks = 0; ks < 2; ks++) c0[mt] = mfma(ldw(W, kBlkC0 + mt * 2 + ks, lane), cf[ks], c0[mt]);
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
        for (int ks = 0; ks < 4; ks++) c1[mt] = mfma(ldw(W, kBlkC1 + mt * 4 + ks, lane), c1f[ks], c1[mt]);
    }
    half8 c2f[4];
    acc_to_frags<true>(c1[0], c2f[0], c2f[1]);
    acc_to_frags<true>(c1[1], c2f[2], c2f[3]);
    f32x16 co;
    #pragma unroll
    for (int r = 0; r < 16; r++) co[r] = 0.0f;
    #pragma unroll
    for (int ks = 0; ks < 4; ks++) co = mfma(ldw(W, kBlkC2 + ks, lane), c2f[ks], co);

    if (h == 0 && valid) {
        P.sigmas[p] = sigma;
        #pragma unroll
        for (int c = 0; c < 3; c++) {
            const float logit = round_h(co[c]);
            P.rgbs[(size_t)p * 3 + c] = round_h(1.0f / (1.0f + expf(-logit)));  // torch.sigmoid on fp16: fp32 math, fp16 result
        }
    }
}

}  // namespace

extern "C" {

uint32_t sdn_field_weight_blocks(void) { return (uint32_t)kBlkTotal; }

// Fused field forward, `-O` numerics (fp16 MLPs / table, fp32 encoders).  weights: packed by dnerf_amd/fused.py
// (sdn_field_weight_blocks() KiB); bias0 [128] f32; table fp16 [rows,2]; offsets_host [17] (16 levels).
int sdn_field_forward_f16(const float *xyzs, const float *dirs, const uint32_t *live_idx, const uint32_t *live_count, uint32_t M,
                          const void *weights, const float *bias0, const void *table, const int32_t *offsets_host, float S, uint32_t H,
                          float bound, float density_scale, int zero_deform, float *sigmas, float *rgbs, void *stream) {
    if (M == 0) return 0;
    if (!xyzs || !dirs || !weights || !bias0 || !table || !offsets_host || !sigmas || !rgbs) return SDN_E_BADARG;
    if ((live_idx == nullptr) != (live_count == nullptr)) return SDN_E_BADARG;
    if (((uintptr_t)weights & 15u) != 0) return SDN_E_BADARG;
    LevelParams lp;
    int rc = fill_levels(lp, offsets_host, 16, S, H);
    if (rc) return rc;
    FieldArgs a;
    a.xyzs = xyzs; a.dirs = dirs; a.live_idx = live_idx; a.live_count = live_count; a.M = M;
    a.weights = (const half8 *)weights; a.bias0 = bias0; a.table = (const __half *)table;
    a.sigmas = sigmas; a.rgbs = rgbs; a.bound = bound; a.density_scale = density_scale; a.zero_deform = zero_deform;
    const uint32_t waves = sdn_div_up(M, 32u);
    hipLaunchKernelGGL(k_field_f16, dim3(sdn_div_up(waves, 4u)), dim3(256), 0, (hipStream_t)stream, a, lp);
    return sdn_launch_status();
}

}  // extern "C"
