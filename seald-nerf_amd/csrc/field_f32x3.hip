// Fused field network with fp32 ACCURACY on the fp16 matrix pipes: every fp32 operand is split x = hi + lo (two fp16 values, 22 bits of
// mantissa) and a product is three MFMAs -- hi.hi + hi.lo + lo.hi, accumulated in fp32 (the dropped lo.lo term is 2^-22 relative).
// v_mfma_f32_32x32x16_f16 does 16 k-steps in 32 cycles where v_mfma_f32_32x32x2_f32 does 2 in 64: three of them are 3 / 16 of the fp32
// MFMA time, paid for with three vector instructions per activation (max, convert, subtract-convert).  Same 1e-4 bar against the fp32
// network as field_f32.hip, same encoders (the fp32 operators' expressions), same staging; the operand layout is the fp16 kernel's
// (field.hip): accumulator registers 8 s .. 8 s + 7 of output tile t are k-step (t, s) of the next layer's B operand, the k-order baked
// into the weight packing (dnerf_amd/fused.py kmaps, reused by fused_f32.py: pack_weights_f32_split).
//
// Scaling.  The fp16 MFMA flushes subnormal inputs, and the lo part of a value is 2^-12 of it: unscaled, every activation below 0.25 and
// every weight below 0.25 would lose its lo part (measured: 2e-4 relative errors).  Activations travel as 2^6 x and weights as 2^8 w
// (exact scalings), accumulators hold 2^14 times the layer's output and are brought back by the 2^-8 of the next split / the 2^-14 of an
// output: lo parts stay normal down to |x| = 4e-3 and |w| = 1e-3, the hi parts fit fp16 up to |x| = 1023 and |w| = 255.
//
// Packed weights, per layer: [k-step][lane][m-tile][hi | lo][8 halves] -- one lane reads its 32 bytes per m-tile with two ds_read_b128.
#include "field_f32_common.h"

namespace {

using namespace sdn_f32;

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
struct Split { half8_t hi, lo; };
constexpr float kXS = 64.0f, kWS = 256.0f;            // operand scales (see "Scaling"); accumulators carry kXS * kWS
constexpr float kAccToX = 1.0f / kWS, kAccToOut = 1.0f / (kXS * kWS);
// offsets inside the tail stage (floats): [k-step][lane][m-tile][hi | lo][8 halves] blocks   (fused_f32.py: pack_weights_f32_split)
constexpr int kBlk = 64 * 8;   // floats of one (k-step, m-tile): 64 lanes x 32 bytes
constexpr int kT_D7 = 0, kT_S0 = kT_D7 + 8 * kBlk, kT_S1 = kT_S0 + 2 * 2 * kBlk, kT_C0 = kT_S1 + 4 * kBlk, kT_C1 = kT_C0 + 2 * 2 * kBlk,
              kT_C2 = kT_C1 + 4 * 2 * kBlk;
static_assert(kT_C2 + 4 * kBlk == kTailFloats && 4 * 4 * kBlk == kD0Floats, "stage sizes");

__device__ __forceinline__ Split split8(const float (&x)[8], float scale) {      // x * scale = hi + lo
    Split r;
    #pragma unroll
    for (int j = 0; j < 8; j++) {
        r.hi[j] = (_Float16)(x[j] * scale);
        // (one fused multiply-add that reads the fp16 operand in place -- v_fma_mix_f32 -- instead of convert-back, multiply, subtract;
        //  the scaling is by a power of two, so the product is exact either way)
        r.lo[j] = (_Float16)__builtin_fmaf(x[j], scale, -(float)r.hi[j]);
    }
    return r;
}

// one layer: KS k-steps of 16 (B operands split in registers) against the staged split A operands, MT output tiles of 32 rows
template <int KS, int MT>
__device__ __forceinline__ void layer(const float *s_w, const Split (&b)[KS], float16_t (&acc)[MT], uint32_t lane) {
    #pragma unroll
    for (int ks = 0; ks < KS; ks++) {
        const uint4 *src = reinterpret_cast<const uint4 *>(s_w) + ((size_t)ks * 64 + lane) * MT * 2;
        half8_t ah[MT], al[MT];
        #pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            ah[mt] = __builtin_bit_cast(half8_t, src[2 * mt]);
            al[mt] = __builtin_bit_cast(half8_t, src[2 * mt + 1]);
        }
        #pragma unroll
        for (int mt = 0; mt < MT; mt++) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], b[ks].hi, acc[mt], 0, 0, 0);
        #pragma unroll
        for (int mt = 0; mt < MT; mt++) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], b[ks].lo, acc[mt], 0, 0, 0);
        #pragma unroll
        for (int mt = 0; mt < MT; mt++) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], b[ks].hi, acc[mt], 0, 0, 0);
#ifdef SDN_X3_LOLO
        // (the fourth term, 2^-22 of the product: measured, changes no result at the test's resolution -- what separates this kernel from the
        //  fp32 one is the 22-bit operands, not the dropped term)
        #pragma unroll
        for (int mt = 0; mt < MT; mt++) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], b[ks].lo, acc[mt], 0, 0, 0);
#endif
    }
}

// accumulator registers 8 s .. 8 s + 7 of tile t -> k-step 2 t + s of the next layer (ReLU, then the split)
template <int MT, bool RELU>
__device__ __forceinline__ void operands_from(const float16_t (&acc)[MT], Split (&b)[2 * MT]) {
    #pragma unroll
    for (int t = 0; t < MT; t++)
        #pragma unroll
        for (int sh = 0; sh < 2; sh++) {
            float x[8];
            #pragma unroll
            for (int j = 0; j < 8; j++) x[j] = RELU ? relu1(acc[t][8 * sh + j]) : acc[t][8 * sh + j];
            b[2 * t + sh] = split8(x, kAccToX);       // accumulators hold kXS kWS y: the next operand is kXS y
        }
}

__global__ void __launch_bounds__(64 * kWaves, 8 / kWaves) k_field_f32x3(F32Args P, LevelParams lp) {
    __shared__ __attribute__((aligned(16))) float s_w[kStageFloats];
    __shared__ float s_bias[kMaxFrames * 128];      // the frames' time-encoding bias rows (D0's initial accumulators)
    Point pt;
    if (!load_point(P, pt)) return;                                  // workgroup-uniform, before any barrier
    const uint32_t lane = pt.lane, h = pt.h, n = pt.n, slot = pt.slot, fr = pt.fr;
    const bool valid = pt.valid, canonical = pt.canonical;
    float x[3] = {pt.x[0], pt.x[1], pt.x[2]}, d[3] = {pt.d[0], pt.d[1], pt.d[2]};
    Pre pre = stage_prefetch<kD0Floats>(P.weights + kD0);            // (see field_f32_common.h: the next stage travels under the layer)
    for (uint32_t k = threadIdx.x; k < P.n_frames * 128u; k += 64 * kWaves) s_bias[k] = P.bias0[k];

    // ---- deformation network: freq(x, 10) (time part folded into bias0) -> 128 x 7 -> 3 ----
    // k position (k-step s, lane half h, j), q = 8 s + j: q < 30 -> pair (f, d) = (5 h + (q >> 1) / 3, (q >> 1) % 3), sine for even q, the
    // reference's phase-shifted sine (the cosine) for odd q; q = 30 -> x0 | x2; q = 31 -> x1 | -   (fused.py _d0_kmap)
    Split b8[8];
    {
        float f[32];
        #pragma unroll
        for (int q = 0; q < 30; q++) {
            const int pr = q >> 1;
            const float xa = h ? x[(15 + pr) % 3] : x[pr % 3];
            const int fa = pr / 3, fb = 5 + pr / 3;
            const float arg = h ? scalbnf(xa, fb) : scalbnf(xa, fa);
            f[q] = sinf(arg + (float)(q & 1) * (3.141592653589793f / 2));
        }
        f[30] = h ? x[2] : x[0];
        f[31] = h ? 0.0f : x[1];
        #pragma unroll
        for (int sk = 0; sk < 4; sk++) {
            float xs[8];
            #pragma unroll
            for (int j = 0; j < 8; j++) xs[j] = f[8 * sk + j];
            b8[sk] = split8(xs, kXS);
        }
    }
    float16_t acc[4];
    stage_commit<kD0Floats>(s_w, pre);
    pre = stage_prefetch<kStageFloats>(P.weights + kD1);
    #pragma unroll
    for (int mt = 0; mt < 4; mt++)
        #pragma unroll
        for (int v = 0; v < 16; v++) acc[mt][v] = s_bias[fr * 128u + mt * 32 + (v >> 2) * 8 + h * 4 + (v & 3)] * (kXS * kWS);
    {
        Split b4[4];
        #pragma unroll
        for (int sk = 0; sk < 4; sk++) b4[sk] = b8[sk];
        layer<4, 4>(s_w, b4, acc, lane);
    }
    #pragma unroll 1
    for (int l = 0; l < 6; l++) {
        operands_from<4, true>(acc, b8);
        stage_commit<kStageFloats>(s_w, pre);                                                     // D(l+1), fetched under the previous layer
        pre = stage_prefetch<kStageFloats>(P.weights + kD1 + (size_t)(l + 1) * kStageFloats);      // D(l+2); after D6 the tail stage (kTail follows D6)
        #pragma unroll
        for (int mt = 0; mt < 4; mt++)
            #pragma unroll
            for (int v = 0; v < 16; v++) acc[mt][v] = 0.0f;
        layer<8, 4>(s_w, b8, acc, lane);
    }
    operands_from<4, true>(acc, b8);
    stage_commit<kStageFloats>(s_w, pre);
    float16_t a1[1];
    #pragma unroll
    for (int v = 0; v < 16; v++) a1[0][v] = 0.0f;
    layer<8, 1>(s_w + kT_D7, b8, a1, lane);
    // rows 0..2 of the output live in registers 0..2 of the lower lane half; the upper half evaluates the same point
    if (P.deform && valid && h == 0) {      // dnerf/network.py:139-141: `deform = zeros` on the canonical frame
        #pragma unroll
        for (int k = 0; k < 3; k++) P.deform[(size_t)slot * 3 + k] = canonical ? 0.0f : a1[0][k] * kAccToOut;
    }
    #pragma unroll
    for (int k = 0; k < 3; k++) {
        const float dk = __shfl(a1[0][k], (int)n, 64) * kAccToOut;
        if (!canonical) x[k] = x[k] + dk;
    }

    // ---- sigma network: grid(x') -> 64 -> 16.  Lane half h owns levels 8 h .. 8 h + 7, both channels (fused.py _s0_kmap:
    //      k position (s, h, j) = level 8 h + 4 s + (j >> 1), channel j & 1) ----
    Split b2[2];
    {
        float in[3];
        bool oob = false;
        #pragma unroll
        for (int k = 0; k < 3; k++) {
            in[k] = (x[k] + P.bound) / (2 * P.bound);            // grid.py:149
            if (in[k] < 0 || in[k] > 1) oob = true;
        }
        #pragma unroll
        for (int sk = 0; sk < 2; sk++) {
            float g[8];
            #pragma unroll
            for (int lv = 0; lv < 4; lv++) {
                const uint32_t level = 8u * h + 4u * sk + lv;
                const float *grid = P.table + (size_t)lp.offset[level] * 2;
                const uint32_t hashmap_size = lp.hashmap_size[level], resolution = lp.resolution[level];
                const float scale = lp.scale[level];
                float pos[3];
                uint32_t pg[3];
                #pragma unroll
                for (int k = 0; k < 3; k++) {
                    pos[k] = in[k] * scale + 0.5f;
                    pg[k] = (uint32_t)floorf(pos[k]);
                    pos[k] -= (float)pg[k];
                }
                float r0 = 0, r1 = 0;
                if (!oob) {
                    float2 vals[8];
                    float ws[8];
                    #pragma unroll
                    for (uint32_t idx = 0; idx < 8; idx++) {
                        float w = 1;
                        uint32_t pgl[3];
                        #pragma unroll
                        for (uint32_t k = 0; k < 3; k++) {
                            w *= (idx & (1u << k)) ? pos[k] : 1 - pos[k];
                            pgl[k] = pg[k] + ((idx >> k) & 1u);
                        }
                        ws[idx] = w;
                        vals[idx] = *reinterpret_cast<const float2 *>(grid + sdn_grid::grid_index<3, 2>(1u, false, hashmap_size, resolution, pgl));
                    }
                    #pragma unroll
                    for (uint32_t idx = 0; idx < 8; idx++) { r0 = r0 + ws[idx] * vals[idx].x; r1 = r1 + ws[idx] * vals[idx].y; }
                }
                g[2 * lv] = r0; g[2 * lv + 1] = r1;
            }
            b2[sk] = split8(g, kXS);
        }
    }
    float16_t a2[2];
    #pragma unroll
    for (int mt = 0; mt < 2; mt++)
        #pragma unroll
        for (int v = 0; v < 16; v++) a2[mt][v] = 0.0f;
    layer<2, 2>(s_w + kT_S0, b2, a2, lane);
    Split b4[4];
    operands_from<2, true>(a2, b4);
    #pragma unroll
    for (int v = 0; v < 16; v++) a1[0][v] = 0.0f;
    layer<4, 1>(s_w + kT_S1, b4, a1, lane);
    const float sigma = expf(a1[0][0] * kAccToOut) * P.density_scale;     // row 0 (lower half); trunc_exp's forward is exp

    // ---- colour network: k-step 0 = the sigma net's 16 outputs in accumulator order (raw; the density logit's column is zero),
    //      k-step 1 = SH coefficient 8 h + j (fused.py _c0_kmap) -> 64 -> 64 -> 3 ----
    {
        float xs[8], sh[16], *nul = nullptr;
        #pragma unroll
        for (int j = 0; j < 8; j++) xs[j] = a1[0][j];
        b2[0] = split8(xs, kAccToX);
        sdn_sh::sh_eval<4, false>(d[0], d[1], d[2], sh, nul, nul, nul);
        #pragma unroll
        for (int j = 0; j < 8; j++) xs[j] = h ? sh[8 + j] : sh[j];
        b2[1] = split8(xs, kXS);
    }
    #pragma unroll
    for (int mt = 0; mt < 2; mt++)
        #pragma unroll
        for (int v = 0; v < 16; v++) a2[mt][v] = 0.0f;
    layer<2, 2>(s_w + kT_C0, b2, a2, lane);
    operands_from<2, true>(a2, b4);
    #pragma unroll
    for (int mt = 0; mt < 2; mt++)
        #pragma unroll
        for (int v = 0; v < 16; v++) a2[mt][v] = 0.0f;
    layer<4, 2>(s_w + kT_C1, b4, a2, lane);
    operands_from<2, true>(a2, b4);
    #pragma unroll
    for (int v = 0; v < 16; v++) a1[0][v] = 0.0f;
    layer<4, 1>(s_w + kT_C2, b4, a1, lane);
    if (valid && h == 0) {
        P.sigmas[slot] = sigma;
        #pragma unroll
        for (int k = 0; k < 3; k++) P.rgbs[(size_t)slot * 3 + k] = 1.0f / (1.0f + expf(-a1[0][k] * kAccToOut));
    }
}

}  // namespace

namespace sdn_int {
int field_forward_f32x3(const float *xyzs, const float *dirs, const uint32_t *live_idx, const uint32_t *live_count, const int32_t *state,
       uint32_t M, const float *weights, const float *bias0, const float *table, const int32_t *offsets_host, float S,
       uint32_t H, float bound, float density_scale, int zero_deform, float *sigmas, float *rgbs, float *deform,
       const uint8_t *slot_frame, uint32_t n_frames, hipStream_t st) {
    sdn_f32::LevelParams lp;
    sdn_f32::F32Args a;
    int rc = sdn_f32::fill_args(a, lp, xyzs, dirs, live_idx, live_count, state, M, weights, bias0, table, offsets_host, S, H, bound, density_scale,
                                zero_deform, sigmas, rgbs, deform, slot_frame, n_frames);
    if (rc) return rc;
    hipLaunchKernelGGL(k_field_f32x3, dim3(sdn_div_up(M, (uint32_t)sdn_f32::kPointsPerWG)), dim3(64 * sdn_f32::kWaves), 0, st, a, lp);
    return sdn_launch_status();
}
}  // namespace sdn_int

extern "C" {


int sdn_field_forward_f32x3(const float *xyzs, const float *dirs, const uint32_t *live_idx, const uint32_t *live_count, uint32_t M,
                          const float *weights, const float *bias0, const float *table, const int32_t *offsets_host, float S, uint32_t H,
                          float bound, float density_scale, int zero_deform, float *sigmas, float *rgbs, float *deform, void *stream) {
    if (M == 0) return 0;
    if (!xyzs || !dirs || !weights || !bias0 || !table || !offsets_host || !sigmas || !rgbs) return SDN_E_BADARG;
    if ((live_idx == nullptr) != (live_count == nullptr)) return SDN_E_BADARG;
    if (((uintptr_t)weights & 15u) != 0 || ((uintptr_t)table & 3u) != 0) return SDN_E_BADARG;
    return sdn_int::field_forward_f32x3(xyzs, dirs, live_idx, live_count, nullptr, M, weights, bias0, table, offsets_host, S, H, bound,
                                      density_scale, zero_deform ? 1 : 0, sigmas, rgbs, deform, nullptr, 1u, (hipStream_t)stream);
}

}  // extern "C"
