// Native training step of the dynamic field for gfx950 (MI355X): one C call = one optimizer step.
//
// Behavioural contract (what the step computes, in the reference's own order and dtypes):
//   dnerf/utils.py:38-125       train_step: render the ray batch, MSE against the ground truth (main_dnerf.py:103), mean
//   dnerf/renderer.py:260-331   run_cuda, training branch: near_far_from_aabb -> march_rays_train(perturb, mean_count budget) ->
//                               network -> density_scale -> composite_rays_train -> image + (1 - weights_sum) * bg
//   dnerf/network.py:123-169    freq(x,10) ++ freq(t,6) -> deformation MLP (8 x 128) -> x + deform (0 at t == 0) -> tiled grid ->
//                               sigma MLP (64) -> trunc_exp | SH(d,4) ++ geo -> colour MLP (64,64) -> sigmoid, all bias-free
//                               Linear layers under fp16 autocast (fp16 operands, fp32 accumulation, fp16 result per layer)
//   nerf/utils.py:880-906       GradScaler.scale(loss).backward(); scaler.step(optimizer); scaler.update(); ema.update()
//   main_dnerf.py:118           Adam(betas (0.9, 0.99), eps 1e-15) over network.py:260-275's groups (tables at lr, MLPs at lr_net)
//
// Why a native step: at 4096 rays / ~9000 samples the op-by-op step is ~160 launches of a few microseconds each (autograd glue:
// casts, fills, cats, the per-call weight packing, a 48 -> 24 MB table cast per step, three optimizer passes) -- launch-bound even
// when replayed as one HIP graph.  Here the step is ~35 launches: the existing operators' kernels (marching, fused MLP chains,
// grid encoder, compositing) composed directly, the glue between them fused into a handful of small kernels, every layer's
// weight gradient in one split-K launch, and ONE optimizer pass that also rewrites the fp16 copies the next step reads and clears
// the table's gradient accumulator.  Nothing is read back to the host.
//
// Everything below the MLP chains is per-sample streaming work (a few hundred bytes per sample): those kernels are launch- and
// latency-bound at this batch size, not bandwidth-bound; the optimizer pass over the 12 M-entry table is the one HBM-bound piece
// (30 B per entry: p, m, v read + written fp32, fp16 gradient read + cleared, fp16 copy written).
#include <math.h>
#include <stddef.h>

#include "sdn_common.h"
#include "sdn_internal.h"
#include "sh_eval.h"

namespace {

using namespace sdn_sh;
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// ---- the network's fixed geometry (dnerf/network.py:10-96 defaults; the mirror is dnerf_amd/network.py) ------------------------------
constexpr uint32_t kDefIn = 80, kDefCols = 76, kDefW = 128, kDefL = 7;   // 63 + 13 frequency features (padded to 80) -> 7 hidden layers -> 3
constexpr uint32_t kSigIn = 32, kSigW = 64, kSigOut = 16;               // 16 levels x 2 -> 64 -> 1 + 15
constexpr uint32_t kColIn = 32, kColCols = 31, kColW = 64, kColL = 2;    // SH 16 ++ geo 15 (padded to 32) -> 64 -> 64 -> 3
constexpr uint32_t kDefFlat = kDefW * kDefIn + (kDefL - 1) * kDefW * kDefW + 16 * kDefW;
constexpr uint32_t kColFlat = kColW * kColIn + (kColL - 1) * kColW * kColW + 16 * kColW;
constexpr uint32_t kLevels = 16;
constexpr uint32_t ACT_RELU = 0;

struct Hyper {          // written by the optimizer prologue, read by the update kernel
    float inv_scale;
    float step_size[2][2];   // [table | net][everything else | deformation MLP]
    float bc2_sqrt[2];
    float found_inf;
    int32_t skip;
    int32_t pad_[7];
};

struct Layout {
    uint64_t total;
    uint64_t w_table, w_deform, w_sigma0, w_sigma1, w_color, g_table, hyper;                 // persistent
    uint64_t pk_def_f, pk_def_b, pk_col_f, pk_col_b;
    uint64_t nears, fars, noises, rays, pts, march;     // sample set 0 ...
    uint64_t set_stride;                                //  ... set 1 = + set_stride
    uint64_t dcol_out, dh0;
    uint64_t enc_in, def_hidden, def_out, xdef, grid_out, dy_dx, enc_rm, h1, hout, sigmas, col_in, col_hidden, col_out;
    uint64_t weights_sum, depth, image, sq_err;
    uint64_t col_bwd, dcol_in, dh, dh1, denc, dx16, ddef, def_bwd;
    uint64_t g_deform, g_sigma0, g_sigma1, g_color, dw_partial;
};

void dw_job_list(const Layout &L, unsigned char *ws, uint32_t M, sdn_ffh::DwJob *jobs, uint32_t &n);

Layout make_layout(uint32_t N, uint32_t M, uint32_t max_steps, uint64_t table_entries) {
    Layout L{};
    uint64_t at = 0;
    auto take = [&](uint64_t bytes) { const uint64_t o = at; at = (at + bytes + 255u) & ~(uint64_t)255u; return o; };
    const uint64_t m = M, n = N;
    L.w_table = take(table_entries * 2);  L.w_deform = take(kDefFlat * 2);  L.w_sigma0 = take(kSigW * kSigIn * 2);
    L.w_sigma1 = take(kSigOut * kSigW * 2);  L.w_color = take(kColFlat * 2);  L.g_table = take(table_entries * 2);
    L.hyper = take(sizeof(Hyper));
    L.pk_def_f = take((uint64_t)sdn_ffh::total_frags(kDefIn, kDefW, kDefL, 0, 1) * 1024);
    L.pk_def_b = take((uint64_t)sdn_ffh::total_frags(kDefIn, kDefW, kDefL, 1, 0) * 1024);
    L.pk_col_f = take((uint64_t)sdn_ffh::total_frags(kColIn, kColW, kColL, 0, 1) * 1024);
    L.pk_col_b = take((uint64_t)sdn_ffh::total_frags(kColIn, kColW, kColL, 1, 1) * 1024);
    // two sets of sample buffers (phase 1 of the next batch runs beside phase 2 of this one)
    L.nears = take(n * 4);  L.fars = take(n * 4);  L.noises = take(n * 4);  L.rays = take(n * 12);
    L.pts = take(m * 32);                       // xyzs [M,3] | dirs [M,3] | deltas [M,2], zero-filled by every march
    L.march = take(sdn_march_rays_train_scratch_bytes(N, max_steps));
    L.set_stride = at - L.nears;
    at += L.set_stride;
    L.dcol_out = take(m * 32);                  // the gradient rows the compositing backward fills for the samples rays own:
    L.dh0 = take(m * 2);                        // zero-filled by every step (dh0 must follow dcol_out: one fill)
    L.enc_in = take(m * kDefIn * 2);  L.def_hidden = take(m * kDefL * kDefW * 2);  L.def_out = take(m * 32);
    L.xdef = take(m * 12);  L.grid_out = take(m * kLevels * 4);  L.dy_dx = take(m * kLevels * 12);
    L.enc_rm = take(m * kSigIn * 2);  L.h1 = take(m * kSigW * 2);  L.hout = take(m * kSigOut * 2);  L.sigmas = take(m * 4);
    L.col_in = take(m * kColIn * 2);  L.col_hidden = take(m * kColL * kColW * 2);  L.col_out = take(m * 32);
    L.weights_sum = take(n * 4);  L.depth = take(n * 4);  L.image = take(n * 12);  L.sq_err = take(n * 4);
    L.col_bwd = take(m * kColL * kColW * 2);  L.dcol_in = take(m * kColIn * 2);
    L.dh = take(m * kSigOut * 2);  L.dh1 = take(m * kSigW * 2);  L.denc = take(m * kLevels * 4);  L.dx16 = take(m * 6);
    L.ddef = take(m * 32);  L.def_bwd = take(m * kDefL * kDefW * 2);
    L.g_deform = take(kDefFlat * 2);  L.g_sigma0 = take(kSigW * kSigIn * 2);  L.g_sigma1 = take(kSigOut * kSigW * 2);  L.g_color = take(kColFlat * 2);
    sdn_ffh::DwJob jobs[16];
    uint32_t nj = 0;
    dw_job_list(L, nullptr, M, jobs, nj);
    L.dw_partial = take(sdn_ffh::dw_jobs_bytes(jobs, nj, M));
    L.total = at;
    return L;
}

// all 13 weight-gradient products of the step: deformation MLP (8), colour MLP (3), sigma MLP (2)
void dw_job_list(const Layout &L, unsigned char *ws, uint32_t M, sdn_ffh::DwJob *jobs, uint32_t &n) {
    auto H = [&](uint64_t off) { return (const _Float16 *)(ws + off); };
    auto G = [&](uint64_t off) { return (_Float16 *)(ws + off); };
    n = 0;
    const size_t mw = (size_t)M * kDefW, mc = (size_t)M * kColW;
    // deformation MLP (layout of the flat gradient: [128, 80] ++ 6 x [128, 128] ++ [16, 128]; ffmlp.cu:800-876 for the pairing)
    jobs[n++] = {H(L.ddef), 16, 16, H(L.def_hidden) + (kDefL - 1) * mw, kDefW, kDefW, G(L.g_deform) + kDefW * kDefIn + (kDefL - 1) * kDefW * kDefW};
    for (uint32_t j = kDefL - 1; j >= 1; j--)
        jobs[n++] = {H(L.def_bwd) + (kDefL - 1 - j) * mw, kDefW, kDefW, H(L.def_hidden) + (j - 1) * mw, kDefW, kDefW,
                     G(L.g_deform) + kDefW * kDefIn + (j - 1) * kDefW * kDefW};
    jobs[n++] = {H(L.def_bwd) + (kDefL - 1) * mw, kDefW, kDefW, H(L.enc_in), kDefIn, kDefIn, G(L.g_deform)};
    // colour MLP ([64, 32] ++ [64, 64] ++ [16, 64])
    jobs[n++] = {H(L.dcol_out), 16, 16, H(L.col_hidden) + (kColL - 1) * mc, kColW, kColW, G(L.g_color) + kColW * kColIn + (kColL - 1) * kColW * kColW};
    for (uint32_t j = kColL - 1; j >= 1; j--)
        jobs[n++] = {H(L.col_bwd) + (kColL - 1 - j) * mc, kColW, kColW, H(L.col_hidden) + (j - 1) * mc, kColW, kColW,
                     G(L.g_color) + kColW * kColIn + (j - 1) * kColW * kColW};
    jobs[n++] = {H(L.col_bwd) + (kColL - 1) * mc, kColW, kColW, H(L.col_in), kColIn, kColIn, G(L.g_color)};
    // sigma MLP: dW2 [16, 64] = dh^T h1, dW1 [64, 32] = dh1^T enc
    jobs[n++] = {H(L.dh), kSigOut, kSigOut, H(L.h1), kSigW, kSigW, G(L.g_sigma1)};
    jobs[n++] = {H(L.dh1), kSigW, kSigW, H(L.enc_rm), kSigIn, kSigIn, G(L.g_sigma0)};
}

// ---- small kernels --------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float u01(uint64_t seed, uint32_t i) {   // splitmix64 -> 24 random bits -> [0, 1), like torch.rand's float grid
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1u);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(uint32_t)(z >> 40) * (1.0f / 16777216.0f);
}

__global__ void __launch_bounds__(256) k_train_rays(float *__restrict__ noises, const float *__restrict__ given, uint32_t N, uint64_t seed, int perturb,
                                                    int32_t *counter) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { counter[0] = 0; counter[1] = 0; }                              // renderer.py:296-298 counter.zero_()
    if (i < N) noises[i] = perturb ? (given ? given[i] : u01(seed, i)) : 0.0f;
}

// freq(x, 10) ++ freq(t, 6) as the fp16 input rows of the deformation MLP (freqencoder.cu:30-58's formula; the autocast cast of
// F.linear's input).  One thread per (sample, 16th of a row).
__global__ void __launch_bounds__(256) k_train_encode(const float *__restrict__ xyzs, uint32_t M, float time, _Float16 *__restrict__ enc, Hyper *hyper,
                                                      uint4 *__restrict__ zero_fill, uint32_t zero_n16) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) hyper->found_inf = 0.0f;       // first kernel of a step's own phase: the non-finite flag of this step's gradients
    // the gradient rows the compositing backward fills only for the samples rays own (dcol_out .. dh0: 34 B per sample, i.e. at most
    // three 16-byte words per 16 threads of a sample) are cleared here instead of by a fill of their own
    if (t < zero_n16) zero_fill[t] = make_uint4(0u, 0u, 0u, 0u);
    const uint32_t b = t >> 4, j = t & 15u;
    if (b >= M) return;
    _Float16 *row = enc + (size_t)b * kDefIn;
    const float half_pi = 3.141592653589793f / 2;
    if (j < 10) {
        #pragma unroll
        for (int d = 0; d < 3; d++) {
            const float a = scalbnf(xyzs[(size_t)b * 3 + d], (int)j);
            row[3 + 6 * j + d] = (_Float16)sinf(a + 0.0f);
            row[6 + 6 * j + d] = (_Float16)sinf(a + half_pi);
        }
    } else if (j == 10) {
        #pragma unroll
        for (int d = 0; d < 3; d++) row[d] = (_Float16)xyzs[(size_t)b * 3 + d];
    } else if (j == 11) {
        row[63] = (_Float16)time;
        #pragma unroll
        for (int f = 0; f < 6; f++) {
            const float a = scalbnf(time, f);
            row[64 + 2 * f] = (_Float16)sinf(a + 0.0f);
            row[65 + 2 * f] = (_Float16)sinf(a + half_pi);
        }
    } else if (j == 12) {
        row[76] = 0; row[77] = 0; row[78] = 0; row[79] = 0;
    }
}

// x + deform (network.py:140-145, deform = 0 on the canonical frame), normalised as GridEncoder.forward does (grid.py:146)
__global__ void __launch_bounds__(256) k_train_xdef(const float *__restrict__ xyzs, const _Float16 *__restrict__ def_out, uint32_t M, int zero_deform,
                                                    float bound, float *__restrict__ xdef) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * 3) return;
    const uint32_t b = t / 3, d = t - b * 3;
    const float v = xyzs[t] + (zero_deform ? 0.0f : (float)def_out[(size_t)b * 16 + d]);
    xdef[t] = (v + bound) / (2 * bound);
}

// ---- the sigma MLP on the matrix cores ------------------------------------------------------------------------------------------------
// 32 -> 64 -> 16, 3 072 MACs per sample: one wave = 32 samples = N of v_mfma_f32_32x32x16_f16, weights are the A operand (M = output
// features), 4 + 4 MFMAs forward, 2 + 4 backward.  As in the fused MLP kernels (ffmlp_kernels.h) a layer's accumulator (feature
// rows 8(i>>2) + 4h + (i&3) of tile Mt in registers, the sample on the lane) converts in place into the next layer's B fragments,
// and rows leave / enter as 16-byte pieces through the half-wave exchange.  (A first version ran one LANE per sample on v_dot2 with
// the weights broadcast from LDS: 22 us forward / 13 us backward at 9 000 samples -- a chain of 1 536 dependent dot products per
// lane at one wave per CU.)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 mfma16(h8 a, h8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void half_wave_exchange(uint32_t &a, uint32_t &b) {      // lanes 32-63 of a <-> lanes 0-31 of b (whole wave active)
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}
// this lane's 4-feature runs of two neighbouring 8-feature groups <-> the 16 contiguous bytes of the row it stores / loaded
__device__ __forceinline__ uint4 runs_to_row16(h4 g0, h4 g1) {
    uint2 a = __builtin_bit_cast(uint2, g0), b = __builtin_bit_cast(uint2, g1);
    half_wave_exchange(a.x, b.x);
    half_wave_exchange(a.y, b.y);
    return uint4{a.x, a.y, b.x, b.y};
}
__device__ __forceinline__ void row16_to_runs(uint4 v, h4 &g0, h4 &g1) {
    uint2 a{v.x, v.y}, b{v.z, v.w};
    half_wave_exchange(a.x, b.x);
    half_wave_exchange(a.y, b.y);
    g0 = __builtin_bit_cast(h4, a);
    g1 = __builtin_bit_cast(h4, b);
}
__device__ __forceinline__ h8 join(h4 a, h4 b) { return h8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }
// A fragment of a row-major weight matrix W[rows, ld]: lane (m, h) holds W[m][16 s + 8 (j >> 2) + 4 h + (j & 3)], rows >= n_rows are zero
__device__ __forceinline__ h8 a_frag(const _Float16 *w, uint32_t ld, uint32_t n_rows, uint32_t m, uint32_t s, uint32_t h) {
    if (m >= n_rows) return h8{0, 0, 0, 0, 0, 0, 0, 0};
    const _Float16 *p = w + (size_t)m * ld + 16 * s + 4 * h;
    return join(*reinterpret_cast<const h4 *>(p), *reinterpret_cast<const h4 *>(p + 8));
}
// A fragment of the TRANSPOSE: lane (m, h) holds W[16 s + 8 (j >> 2) + 4 h + (j & 3)][m]
__device__ __forceinline__ h8 a_frag_t(const _Float16 *w, uint32_t ld, uint32_t m, uint32_t s, uint32_t h) {
    h8 v;
    #pragma unroll
    for (int j = 0; j < 8; j++) v[j] = w[(size_t)(16 * s + 8 * (j >> 2) + 4 * h + (j & 3)) * ld + m];
    return v;
}

// forward: grid features -> sigma MLP -> trunc_exp (activation.py:5-17) x density_scale; SH(d, 4); the colour MLP's input row
// [SH 16 | sigma-MLP output 16] (the log-density column meets a zero weight column: `w_color` is kept in that order)
struct SigmaFwd {
    const _Float16 *grid_out;   // [16][M][2]
    const _Float16 *w1, *w2;    // [64,32], [16,64]
    const float *dirs;          // [M,3]
    _Float16 *enc_rm, *h1, *hout, *col_in;
    float *sigmas;
    uint32_t M;
    float density_scale;
};

__global__ void __launch_bounds__(64) k_train_sigma_fwd(SigmaFwd P) {
    const uint32_t lane = threadIdx.x, n = lane & 31u, h = lane >> 5;
    const uint32_t b_raw = blockIdx.x * 32 + n;
    const bool live = b_raw < P.M;
    const uint32_t b = live ? b_raw : P.M - 1;
    h8 B1[2];
    #pragma unroll
    for (uint32_t s = 0; s < 2; s++) {
        h2 e[4];
        #pragma unroll
        for (uint32_t u = 0; u < 2; u++)
            #pragma unroll
            for (uint32_t v = 0; v < 2; v++) e[2 * u + v] = reinterpret_cast<const h2 *>(P.grid_out)[(size_t)(8 * s + 4 * u + 2 * h + v) * P.M + b];
        B1[s] = h8{e[0][0], e[0][1], e[1][0], e[1][1], e[2][0], e[2][1], e[3][0], e[3][1]};
        const uint4 row16 = runs_to_row16(h4{e[0][0], e[0][1], e[1][0], e[1][1]}, h4{e[2][0], e[2][1], e[3][0], e[3][1]});
        if (live) *reinterpret_cast<uint4 *>(P.enc_rm + (size_t)b * kSigIn + 16 * s + 8 * h) = row16;
    }
    h8 B2[4];
    #pragma unroll
    for (uint32_t Mt = 0; Mt < 2; Mt++) {
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        #pragma unroll
        for (uint32_t s = 0; s < 2; s++) acc = mfma16(a_frag(P.w1, kSigIn, kSigW, 32 * Mt + n, s, h), B1[s], acc);
        _Float16 v[16];
        #pragma unroll
        for (int i = 0; i < 16; i++) { const _Float16 x = (_Float16)acc[i]; v[i] = x > (_Float16)0 ? x : (_Float16)0; }
        #pragma unroll
        for (uint32_t p = 0; p < 2; p++) {
            const h4 g0{v[8 * p], v[8 * p + 1], v[8 * p + 2], v[8 * p + 3]}, g1{v[8 * p + 4], v[8 * p + 5], v[8 * p + 6], v[8 * p + 7]};
            B2[2 * Mt + p] = join(g0, g1);
            const uint4 row16 = runs_to_row16(g0, g1);
            if (live) *reinterpret_cast<uint4 *>(P.h1 + (size_t)b * kSigW + 32 * Mt + 16 * p + 8 * h) = row16;
        }
    }
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    #pragma unroll
    for (uint32_t s = 0; s < 4; s++) acc = mfma16(a_frag(P.w2, kSigW, kSigOut, n, s, h), B2[s], acc);
    // rows 0..15 of the output tile: this lane holds o[4h .. 4h+3] (i = 0..3) and o[8+4h .. 8+4h+3] (i = 4..7)
    const h4 o0{(_Float16)acc[0], (_Float16)acc[1], (_Float16)acc[2], (_Float16)acc[3]}, o1{(_Float16)acc[4], (_Float16)acc[5], (_Float16)acc[6], (_Float16)acc[7]};
    const uint4 orow = runs_to_row16(o0, o1);
    float sh[16], u0[1], u1[1], u2[1];
    sh_eval<4, false>(P.dirs[(size_t)b * 3], P.dirs[(size_t)b * 3 + 1], P.dirs[(size_t)b * 3 + 2], sh, u0, u1, u2);
    if (live) {
        *reinterpret_cast<uint4 *>(P.hout + (size_t)b * kSigOut + 8 * h) = orow;
        *reinterpret_cast<uint4 *>(P.col_in + (size_t)b * kColIn + 16 + 8 * h) = orow;
        h8 shh;
        #pragma unroll
        for (int j = 0; j < 8; j++) shh[j] = (_Float16)(h ? sh[8 + j] : sh[j]);
        *reinterpret_cast<h8 *>(P.col_in + (size_t)b * kColIn + 8 * h) = shh;
        if (h == 0) P.sigmas[b] = P.density_scale * expf((float)o0[0]);
    }
}

__device__ __forceinline__ _Float16 sigmoid16(_Float16 c) { return (_Float16)(1.0f / (1.0f + expf(-(float)c))); }

// ---- compositing, one WAVE per ray ------------------------------------------------------------------------------------------------
// composite_rays_train (raymarching.cu:501-577 forward, :602-682 backward) walks a ray's samples one after the other; with one lane per
// ray a batch of 4096 rays is 64 waves whose time is the longest ray's chain of dependent loads and double-precision exps (24 + 30 us
// for ~9000 samples).  Here lane i of a wave takes sample i of the ray: alpha in parallel, transmittance as a prefix PRODUCT across
// the lanes, the ray's sums as wave reductions.  The early stop `T < T_thresh` of the sequential loop becomes a per-sample predicate
// (sample i is composited iff the transmittance in front of it is >= T_thresh: T never grows).  The sums are tree sums, so results
// differ from the sequential operator's in the last bits (the public composite_rays_train op keeps the sequential order and the
// bit-exact contract).  Folded in: the fp16 sigmoid of the colour MLP's output (network.py:166) in front, the background mix and
// the MSE gradient per ray behind (renderer.py:318, utils.py:85), and in the backward kernel the gradients through sigmoid,
// density_scale and trunc_exp (activation.py:12-17) down to the fp16 rows the MLP backward chains start from.
__device__ __forceinline__ float wave_excl_product(float v, uint32_t lane) {   // exclusive prefix product over the 64 lanes
    float inc = v;
    #pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float u = __shfl_up(inc, o); if (lane >= (uint32_t)o) inc *= u; }
    const float up = __shfl_up(inc, 1);
    return lane == 0 ? 1.0f : up;
}
__device__ __forceinline__ float wave_incl_sum(float v, uint32_t lane) {
    #pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float u = __shfl_up(v, o); if (lane >= (uint32_t)o) v += u; }
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

struct CompositeArgs {
    const float *sigmas, *deltas;        // [M] (density_scale applied), [M,2]
    const _Float16 *col_out, *hout;      // [M,16] colour MLP output (pre-sigmoid), [M,16] sigma MLP output (column 0: log density)
    const int32_t *rays;                 // [N,3]
    const float *bg, *gt, *loss_scale;
    float *weights_sum, *depth, *image, *image_out, *sq_err;   // per ray; sq_err [N]: sum over channels of (pred - gt)^2
    _Float16 *dcol_out, *dh0;            // backward: [M,16], [M]
    uint32_t M, N;
    float T_thresh, bg_value, density_scale;
};

// one chunk of <= 64 samples of a ray: everything the forward AND the backward need about sample `i`
struct SampleTerms { bool on; float alpha, T, weight, delta0, t, c[3]; };

__device__ __forceinline__ SampleTerms sample_terms(const CompositeArgs &P, uint32_t idx, bool in_ray, float T_carry, float t_carry, uint32_t lane,
                                                    float &T_next, float &t_next) {
    SampleTerms s{};
    float one_minus = 1.0f, d1 = 0.0f;
    if (in_ray) {
        s.delta0 = P.deltas[(size_t)idx * 2];
        d1 = P.deltas[(size_t)idx * 2 + 1];
        s.alpha = 1.0f - sdn_exp_cr(-P.sigmas[idx] * s.delta0);
        one_minus = 1.0f - s.alpha;
        #pragma unroll
        for (int ch = 0; ch < 3; ch++) s.c[ch] = (float)sigmoid16(P.col_out[(size_t)idx * 16 + ch]);
    }
    s.T = T_carry * wave_excl_product(one_minus, lane);
    s.t = t_carry + wave_incl_sum(d1, lane);
    s.on = in_ray && s.T >= P.T_thresh;
    s.weight = s.on ? s.alpha * s.T : 0.0f;
    T_next = __shfl(s.T * one_minus, 63);
    t_next = __shfl(s.t, 63);
    return s;
}

__global__ void __launch_bounds__(256) k_train_composite_fwd(CompositeArgs P) {
    const uint32_t lane = threadIdx.x & 63u, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= P.N) return;
    const uint32_t index = (uint32_t)P.rays[n * 3], offset = (uint32_t)P.rays[n * 3 + 1], count = (uint32_t)P.rays[n * 3 + 2];
    float ws = 0, d = 0, r = 0, g = 0, b = 0;
    if (count != 0 && offset + count <= P.M) {
        float T_carry = 1.0f, t_carry = 0.0f;
        for (uint32_t c0 = 0; c0 < count && T_carry >= P.T_thresh; c0 += 64) {
            float T_next, t_next;
            const SampleTerms s = sample_terms(P, offset + c0 + lane, c0 + lane < count, T_carry, t_carry, lane, T_next, t_next);
            ws += s.weight; d += s.weight * s.t;
            r += s.weight * s.c[0]; g += s.weight * s.c[1]; b += s.weight * s.c[2];
            T_carry = T_next; t_carry = t_next;
        }
        ws = wave_sum(ws); d = wave_sum(d); r = wave_sum(r); g = wave_sum(g); b = wave_sum(b);
    }
    if (lane == 0) {
        P.weights_sum[index] = ws; P.depth[index] = d;
        const float img[3] = {r, g, b};
        float sq = 0.0f;
        #pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            P.image[(size_t)index * 3 + ch] = img[ch];
            const float bgc = P.bg ? P.bg[(size_t)index * 3 + ch] : P.bg_value;
            const float pred = img[ch] + (1.0f - ws) * bgc;          // renderer.py:318
            const float e = pred - P.gt[(size_t)index * 3 + ch];
            sq += e * e;
            if (P.image_out) P.image_out[(size_t)index * 3 + ch] = pred;
        }
        P.sq_err[index] = sq;
    }
}

// loss = mean over rays of the mean over channels of (pred - gt)^2 (utils.py:85, :125): a deterministic tree sum in ONE workgroup
// (folding it into the last block of the compositing launch, or the optimizer prologue into the last block of the non-finite check,
// was measured and reverted: the agent-scope release every block then needs before its ticket costs ~1.7 us per block, 5 -> 26 us
// and 12 -> 19 us for the two launches)
__global__ void __launch_bounds__(1024) k_train_loss(const float *__restrict__ sq_err, uint32_t N, float *__restrict__ loss_out) {
    __shared__ float s_part[16];
    float acc = 0.0f;
    for (uint32_t r = threadIdx.x; r < N; r += 1024) acc += sq_err[r] / 3.0f;
    acc = wave_sum(acc);
    if ((threadIdx.x & 63u) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
        for (int w = 0; w < 16; w++) t += s_part[w];
        *loss_out = t / (float)N;
    }
}

__global__ void __launch_bounds__(256) k_train_composite_bwd(CompositeArgs P) {
    const uint32_t lane = threadIdx.x & 63u, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= P.N) return;
    const uint32_t index = (uint32_t)P.rays[n * 3], offset = (uint32_t)P.rays[n * 3 + 1], count = (uint32_t)P.rays[n * 3 + 2];
    if (count == 0 || offset + count > P.M) return;
    // gradient of scale * loss with respect to this ray's image and weights_sum (MSE mean over 3 N values; the background mix)
    const float ws_final = P.weights_sum[index];
    const float cg = (*P.loss_scale / (float)P.N) / 3.0f;
    float gi[3], fin[3], gws = 0.0f;
    #pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        fin[ch] = P.image[(size_t)index * 3 + ch];
        const float bgc = P.bg ? P.bg[(size_t)index * 3 + ch] : P.bg_value;
        const float pred = fin[ch] + (1.0f - ws_final) * bgc;
        gi[ch] = ((pred - P.gt[(size_t)index * 3 + ch]) * 2.0f) * cg;
        gws -= gi[ch] * bgc;
    }
    float T_carry = 1.0f, t_carry = 0.0f, acc_c[3] = {0, 0, 0};
    for (uint32_t c0 = 0; c0 < count && T_carry >= P.T_thresh; c0 += 64) {
        float T_next, t_next;
        const uint32_t idx = offset + c0 + lane;
        const SampleTerms s = sample_terms(P, idx, c0 + lane < count, T_carry, t_carry, lane, T_next, t_next);
        // raymarching.cu:655-675: r, g, b are the running sums INCLUDING this sample, T the transmittance BEHIND it
        float run[3];
        #pragma unroll
        for (int ch = 0; ch < 3; ch++) run[ch] = acc_c[ch] + wave_incl_sum(s.weight * s.c[ch], lane);
        if (s.on) {
            const float T_after = s.T * (1.0f - s.alpha);
            const float g_sigma = s.delta0 * (gi[0] * (T_after * s.c[0] - (fin[0] - run[0])) + gi[1] * (T_after * s.c[1] - (fin[1] - run[1])) +
                                              gi[2] * (T_after * s.c[2] - (fin[2] - run[2])) + gws * (1.0f - ws_final));
            _Float16 dc[3];
            #pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const _Float16 g16 = (_Float16)(gi[ch] * s.weight);                       // composite's float32 cast, backwards
                dc[ch] = (_Float16)(((float)g16 * (1.0f - s.c[ch])) * s.c[ch]);             // sigmoid backward on fp16 (a * (1 - y) * y in float)
            }
            *reinterpret_cast<h8 *>(P.dcol_out + (size_t)idx * 16) = h8{dc[0], dc[1], dc[2], 0, 0, 0, 0, 0};
            const float x = fminf(fmaxf((float)P.hout[(size_t)idx * kSigOut], -15.0f), 15.0f);
            P.dh0[idx] = (_Float16)((g_sigma * P.density_scale) * expf(x));              // density_scale, trunc_exp backward, cast to fp16
        }
        #pragma unroll
        for (int ch = 0; ch < 3; ch++) acc_c[ch] = __shfl(run[ch], 63);
        T_carry = T_next; t_carry = t_next;
    }
}

// sigma MLP backward: dh = [trunc_exp gradient | colour MLP's gradient of columns 17..31 of its input], dh1 = (W2^T dh) * relu',
// denc = W1^T dh1 in the grid encoder's [level][sample][2] layout; dh and dh1 also row-major for the weight-gradient products.
struct SigmaBwd {
    const _Float16 *dh0, *dcol_in, *h1, *w1, *w2;
    _Float16 *dh, *dh1, *denc;
    uint32_t M;
};

__global__ void __launch_bounds__(64) k_train_sigma_bwd(SigmaBwd P) {
    const uint32_t lane = threadIdx.x, n = lane & 31u, h = lane >> 5;
    const uint32_t b_raw = blockIdx.x * 32 + n;
    const bool live = b_raw < P.M;
    const uint32_t b = live ? b_raw : P.M - 1;
    // columns 16..31 of the colour MLP's input gradient are d(sigma-MLP output); column 16 (log density: zero weights there) is
    // replaced by the compositing's gradient through trunc_exp
    uint4 raw = *reinterpret_cast<const uint4 *>(P.dcol_in + (size_t)b * kColIn + 16 + 8 * h);
    if (h == 0) raw.x = (raw.x & 0xFFFF0000u) | (uint32_t)__builtin_bit_cast(unsigned short, P.dh0[b]);
    if (live) *reinterpret_cast<uint4 *>(P.dh + (size_t)b * kSigOut + 8 * h) = raw;
    h4 d0, d1;
    row16_to_runs(raw, d0, d1);
    const h8 Bd = join(d0, d1);
    h8 B2[4];
    #pragma unroll
    for (uint32_t Mt = 0; Mt < 2; Mt++) {
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        acc = mfma16(a_frag_t(P.w2, kSigW, 32 * Mt + n, 0, h), Bd, acc);
        #pragma unroll
        for (uint32_t p = 0; p < 2; p++) {
            h4 f0, f1;
            row16_to_runs(*reinterpret_cast<const uint4 *>(P.h1 + (size_t)b * kSigW + 32 * Mt + 16 * p + 8 * h), f0, f1);
            h4 g0, g1;
            #pragma unroll
            for (int e = 0; e < 4; e++) {
                g0[e] = f0[e] > (_Float16)0 ? (_Float16)acc[8 * p + e] : (_Float16)0;
                g1[e] = f1[e] > (_Float16)0 ? (_Float16)acc[8 * p + 4 + e] : (_Float16)0;
            }
            B2[2 * Mt + p] = join(g0, g1);
            const uint4 row16 = runs_to_row16(g0, g1);
            if (live) *reinterpret_cast<uint4 *>(P.dh1 + (size_t)b * kSigW + 32 * Mt + 16 * p + 8 * h) = row16;
        }
    }
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    #pragma unroll
    for (uint32_t s = 0; s < 4; s++) acc = mfma16(a_frag_t(P.w1, kSigIn, n, s, h), B2[s], acc);
    // row m = 8 q + 4 h + (i & 3) of the tile is grid feature m = 2 level + channel
    if (live) {
        #pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
            reinterpret_cast<h2 *>(P.denc)[(size_t)(4 * q + 2 * h) * P.M + b] = h2{(_Float16)acc[4 * q], (_Float16)acc[4 * q + 1]};
            reinterpret_cast<h2 *>(P.denc)[(size_t)(4 * q + 2 * h + 1) * P.M + b] = h2{(_Float16)acc[4 * q + 2], (_Float16)acc[4 * q + 3]};
        }
    }
}

// gradient of the deformation MLP's output: grid input gradient (fp16 -> float), / (2 bound) of the normalisation, the fp16 cast of
// `deform.to(x.dtype)`; zero columns 3..15 of the operator's 16-wide output
__global__ void __launch_bounds__(256) k_train_deform_grad(const _Float16 *__restrict__ dx16, uint32_t M, float bound, _Float16 *__restrict__ ddef) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= M) return;
    _Float16 d[3];
    #pragma unroll
    for (int c = 0; c < 3; c++) d[c] = (_Float16)((float)dx16[(size_t)b * 3 + c] / (2 * bound));
    h8 *row = reinterpret_cast<h8 *>(ddef + (size_t)b * 16);
    row[0] = h8{d[0], d[1], d[2], 0, 0, 0, 0, 0};
    row[1] = h8{0, 0, 0, 0, 0, 0, 0, 0};
}

// ---- optimizer ------------------------------------------------------------------------------------------------------------------
struct CheckArgs { const _Float16 *g[5]; uint32_t n8[5]; uint32_t count; Hyper *hyper; uint32_t col_seg; };

struct Prologue {
    Hyper *hyper;
    float *steps, *scale;
    int32_t *tracker;
    double beta1, beta2, lr_table, lr_net;
    float growth, backoff;
    uint32_t interval;
    int deform_active;
    float divisor;
};

// GradScaler's non-finite check over every gradient of the step (amp's _amp_foreach_non_finite_check_and_unscale_)
__global__ void __launch_bounds__(256) k_train_check(CheckArgs A) {
    bool bad = false;
    for (uint32_t s = 0; s < A.count; s++) {
        const uint4 *p = reinterpret_cast<const uint4 *>(A.g[s]);
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < A.n8[s]; i += gridDim.x * blockDim.x) {
            const uint4 v = p[i];
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
            // input column 16 of the colour MLP's first layer is the log-density column: its weight is structurally zero and its
            // "gradient" (dcol_hidden^T . log-density) has no counterpart in the reference and never reaches Adam (the `split` of
            // build_segments skips it) -- an fp16 overflow there must not skip the step.  Row r, column 16 = half 0 of uint4 4 r + 2.
            if (s == A.col_seg && i < kColW * kColIn / 8 && (i & 3u) == 2u) w[0] &= 0xFFFF0000u;
            #pragma unroll
            for (int k = 0; k < 4; k++) bad |= ((w[k] & 0x7C00u) == 0x7C00u) | ((w[k] & 0x7C000000u) == 0x7C000000u);
        }
    }
    if (bad) A.hyper->found_inf = 1.0f;
}

// One thread: the skip decision, bias corrections in double as torch's Adam computes them on the host (adam.py _single_tensor_adam:
// 1 - beta ** step, lr / bias_correction1, bias_correction2 ** 0.5), and amp_update_scale (AmpKernels.cu) for the NEXT step.
__global__ void k_train_prologue(Prologue P) {
    Hyper *h = P.hyper;
    const bool found = h->found_inf != 0.0f;
    const float scale = *P.scale;
    h->skip = found ? 1 : 0;
    h->inv_scale = (float)(1.0 / ((double)scale * (double)P.divisor));
    if (!found) {
        P.steps[0] += 1.0f;
        if (P.deform_active) P.steps[1] += 1.0f;
    }
    for (int g = 0; g < 2; g++) {
        const double st = (double)P.steps[g] > 0 ? (double)P.steps[g] : 1.0;
        const double bc1 = 1.0 - pow(P.beta1, st), bc2 = 1.0 - pow(P.beta2, st);
        h->step_size[0][g] = (float)(P.lr_table / bc1);
        h->step_size[1][g] = (float)(P.lr_net / bc1);
        h->bc2_sqrt[g] = (float)sqrt(bc2);
    }
    if (found) {
        *P.scale = scale * P.backoff;
        *P.tracker = 0;
    } else {
        const int32_t ok = *P.tracker + 1;
        if (ok == (int32_t)P.interval) {
            const float grown = scale * P.growth;
            if (isfinite(grown)) *P.scale = grown;
            *P.tracker = 0;
        } else {
            *P.tracker = ok;
        }
    }
}

struct AdamSeg {
    float *p, *m, *v, *ema;
    _Float16 *g;        // fp16 gradient, element (r, c) at r * ld + c
    _Float16 *w16;      // fp16 copy, same layout
    uint32_t n, cols, ld, split;      // columns >= split sit one column further right (the colour MLP's input layer: [SH 16 | 0 | geo 15])
    uint32_t lr_idx, group, zero_grad, frozen, blk_begin;
};
struct AdamArgs {
    AdamSeg seg[SDN_TRAIN_N_PARAMS];
    uint32_t nseg;
    const Hyper *hyper;
    float one_minus_b1, b2, one_minus_b2, eps, ema_keep;   // ema_keep = 1 - decay
};

// torch.optim.Adam, single-tensor form (adam.py:_single_tensor_adam; amsgrad / weight_decay / maximize off), on gradients unscaled
// the way GradScaler.unscale_ does (float(grad) * inv_scale):   m.lerp_(g, 1 - b1);  v.mul_(b2).addcmul_(g, g, 1 - b2);
// denom = sqrt(v) / bias_correction2_sqrt + eps;  p.addcdiv_(m, denom, -step_size).   The same pass writes the fp16 copy the next
// step's kernels read (the autocast casts of the reference), clears the table's gradient accumulator, and applies torch_ema's
// shadow -= (1 - decay) * (shadow - p).  A step with a non-finite gradient changes nothing but the accumulator and the shadows.
__global__ void __launch_bounds__(256) k_train_adam(AdamArgs A) {
    uint32_t s = 0;
    for (uint32_t k = 1; k < A.nseg; k++) s = blockIdx.x >= A.seg[k].blk_begin ? k : s;
    const AdamSeg &S = A.seg[s];
    const Hyper &H = *A.hyper;
    const bool update = !H.skip && !S.frozen;
    const float inv_scale = H.inv_scale, step_size = H.step_size[S.lr_idx][S.group], bc2s = H.bc2_sqrt[S.group];
    // the table (12.2 M contiguous entries: the pass's HBM traffic) four entries per lane: 16-byte loads and stores
    const uint32_t first = (blockIdx.x - S.blk_begin) * 1024u;
    if (S.cols == S.ld && first + 1024u <= S.n && !S.ema && update) {
        const uint32_t i = first + 4u * threadIdx.x;
        float4 p = *reinterpret_cast<const float4 *>(S.p + i), m = *reinterpret_cast<const float4 *>(S.m + i), v = *reinterpret_cast<const float4 *>(S.v + i);
        const h4 g16 = *reinterpret_cast<const h4 *>(S.g + i);
        float *pp = &p.x, *mm = &m.x, *vv = &v.x;
        h4 w;
        #pragma unroll
        for (int e = 0; e < 4; e++) {
            const float g = (float)g16[e] * inv_scale;
            mm[e] = mm[e] + A.one_minus_b1 * (g - mm[e]);
            vv[e] = vv[e] * A.b2 + (A.one_minus_b2 * g) * g;
            const float denom = sqrtf(vv[e]) / bc2s + A.eps;
            pp[e] = pp[e] - step_size * (mm[e] / denom);
            w[e] = (_Float16)pp[e];
        }
        *reinterpret_cast<float4 *>(S.m + i) = m; *reinterpret_cast<float4 *>(S.v + i) = v; *reinterpret_cast<float4 *>(S.p + i) = p;
        *reinterpret_cast<h4 *>(S.w16 + i) = w;
        if (S.zero_grad) *reinterpret_cast<h4 *>(S.g + i) = h4{0, 0, 0, 0};
        return;
    }
    const uint32_t base = first + threadIdx.x;
    #pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t i = base + 256u * k;
        if (i >= S.n) break;
        uint32_t j = i;
        if (S.cols != S.ld) { const uint32_t r = i / S.cols, c = i - r * S.cols; j = r * S.ld + c + (c >= S.split ? 1u : 0u); }
        float p = S.p[i];
        if (update) {
            const float g = (float)S.g[j] * inv_scale;
            float m = S.m[i], v = S.v[i];
            m = m + A.one_minus_b1 * (g - m);
            v = v * A.b2 + (A.one_minus_b2 * g) * g;
            const float denom = sqrtf(v) / bc2s + A.eps;
            p = p - step_size * (m / denom);
            S.m[i] = m; S.v[i] = v; S.p[i] = p;
            S.w16[j] = (_Float16)p;
        }
        if (S.zero_grad) S.g[j] = (_Float16)0;
        if (S.ema) { const float e = S.ema[i]; S.ema[i] = e - (e - p) * A.ema_keep; }
    }
}

// fp16 copies from the fp32 masters (sdn_train_refresh)
__global__ void __launch_bounds__(256) k_train_copy16(AdamArgs A) {
    uint32_t s = 0;
    for (uint32_t k = 1; k < A.nseg; k++) s = blockIdx.x >= A.seg[k].blk_begin ? k : s;
    const AdamSeg &S = A.seg[s];
    const uint32_t base = (blockIdx.x - S.blk_begin) * 1024u + threadIdx.x;
    #pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t i = base + 256u * k;
        if (i >= S.n) break;
        uint32_t j = i;
        if (S.cols != S.ld) { const uint32_t r = i / S.cols, c = i - r * S.cols; j = r * S.ld + c + (c >= S.split ? 1u : 0u); }
        S.w16[j] = (_Float16)S.p[i];
    }
}

// parameter i of SDN_TRAIN_N_PARAMS -> its place in the fp16 copies / gradients
uint32_t build_segments(const SdnTrainStep *s, const Layout &L, AdamArgs &A) {
    unsigned char *ws = (unsigned char *)s->workspace;
    auto H = [&](uint64_t off) { return (_Float16 *)(ws + off); };
    struct Place { uint64_t w, g; uint32_t off, rows, cols, ld, lr, group; };
    Place pl[SDN_TRAIN_N_PARAMS];
    const uint32_t table_n = (uint32_t)s->grid_offsets[kLevels] * 2u;
    pl[0] = {L.w_table, L.g_table, 0, table_n / 2, 2, 2, 0, 0};
    pl[1] = {L.w_deform, L.g_deform, 0, kDefW, kDefCols, kDefIn, 1, 1};
    for (uint32_t i = 1; i < kDefL; i++) pl[1 + i] = {L.w_deform, L.g_deform, kDefW * kDefIn + (i - 1) * kDefW * kDefW, kDefW, kDefW, kDefW, 1, 1};
    pl[8] = {L.w_deform, L.g_deform, kDefW * kDefIn + (kDefL - 1) * kDefW * kDefW, 3, kDefW, kDefW, 1, 1};
    pl[9] = {L.w_sigma0, L.g_sigma0, 0, kSigW, kSigIn, kSigIn, 1, 0};
    pl[10] = {L.w_sigma1, L.g_sigma1, 0, kSigOut, kSigW, kSigW, 1, 0};
    pl[11] = {L.w_color, L.g_color, 0, kColW, kColCols, kColIn, 1, 0};
    pl[12] = {L.w_color, L.g_color, kColW * kColIn, kColW, kColW, kColW, 1, 0};
    pl[13] = {L.w_color, L.g_color, kColW * kColIn + kColW * kColW, 3, kColW, kColW, 1, 0};
    uint32_t blk = 0;
    for (int i = 0; i < SDN_TRAIN_N_PARAMS; i++) {
        const SdnTrainParam &q = s->params[i];
        AdamSeg &S = A.seg[i];
        S.p = q.param; S.m = q.exp_avg; S.v = q.exp_avg_sq; S.ema = q.ema;
        S.g = H(pl[i].g) + pl[i].off; S.w16 = H(pl[i].w) + pl[i].off;
        S.n = pl[i].rows * pl[i].cols; S.cols = pl[i].cols; S.ld = pl[i].ld; S.split = i == 11 ? 16u : 0xFFFFFFFFu;
        S.lr_idx = pl[i].lr; S.group = pl[i].group; S.zero_grad = i == 0; S.frozen = 0; S.blk_begin = blk;
        blk += sdn_div_up(S.n, 1024u);
    }
    A.nseg = SDN_TRAIN_N_PARAMS;
    return blk;
}

// Adam over the segments [i0, i1) of a full segment list (workgroup ranges re-based)
void launch_adam(const AdamArgs &all, uint32_t i0, uint32_t i1, hipStream_t st) {
    AdamArgs A = all;
    uint32_t blk = 0;
    for (uint32_t i = i0; i < i1; i++) {
        A.seg[i - i0] = all.seg[i];
        A.seg[i - i0].blk_begin = blk;
        blk += sdn_div_up(all.seg[i].n, 1024u);
    }
    A.nseg = i1 - i0;
    hipLaunchKernelGGL(k_train_adam, dim3(blk), dim3(256), 0, st, A);
}

int table_wait(const SdnTrainStep *s, hipStream_t st) {
    if (s->table_stream && s->table_done && hipStreamWaitEvent(st, (hipEvent_t)s->table_done, 0) != hipSuccess) return sdn_launch_status();
    return 0;
}

bool step_ok(const SdnTrainStep *s) {
    if (!s || !s->workspace || ((uintptr_t)s->workspace & 255u)) return false;
    if (s->mode < 0 || s->mode > 2) return false;
    if ((s->table_stream != nullptr) != (s->table_ready != nullptr) || (s->table_stream != nullptr) != (s->table_done != nullptr)) return false;
    if (s->phase < 0 || s->phase > 2 || (s->mode == 2 && s->phase != 0)) return false;
    if (s->mode != 2 && s->phase != 2 && (!s->rays_o || !s->rays_d || !s->bitfield || !s->aabb || !s->counter)) return false;
    if (s->mode != 2 && s->phase != 1 && (!s->target || !s->loss_out)) return false;
    if (s->N == 0 || s->M == 0 || s->max_steps == 0 || s->bound <= 0) return false;
    for (int i = 0; i < SDN_TRAIN_N_PARAMS; i++) if (!s->params[i].param) return false;
    if (s->mode != 1) {
        if (!s->adam_steps || !s->loss_scale || !s->growth_tracker) return false;
        // (a frozen deformation MLP -- SealD-NeRF's edit training -- has no optimizer state: its Adam segments are never touched)
        for (int i = 0; i < SDN_TRAIN_N_PARAMS; i++) {
            const bool frozen = s->deform_frozen && i >= 1 && i <= (int)kDefL + 1;
            if (!frozen && (!s->params[i].exp_avg || !s->params[i].exp_avg_sq)) return false;
        }
    } else if (!s->loss_scale) {
        return false;
    }
    const uint64_t expect[SDN_TRAIN_N_PARAMS] = {(uint64_t)s->grid_offsets[kLevels] * 2, kDefW * kDefCols, kDefW * kDefW, kDefW * kDefW, kDefW * kDefW, kDefW * kDefW,
                                                 kDefW * kDefW, kDefW * kDefW, 3 * kDefW, kSigW * kSigIn, kSigOut * kSigW, kColW * kColCols, kColW * kColW, 3 * kColW};
    for (int i = 0; i < SDN_TRAIN_N_PARAMS; i++) if (s->params[i].n != expect[i]) return false;
    return true;
}

}  // namespace

extern "C" {

int sdn_train_layout(uint32_t N, uint32_t M, uint32_t max_steps, const int32_t *grid_offsets, SdnTrainLayout *out) {
    if (!grid_offsets || !out || N == 0 || M == 0 || max_steps == 0 || grid_offsets[kLevels] <= 0) return SDN_E_BADARG;
    const Layout L = make_layout(N, M, max_steps, (uint64_t)grid_offsets[kLevels] * 2);
    out->total_bytes = L.total;
    out->w_table = L.w_table; out->w_deform = L.w_deform; out->w_sigma0 = L.w_sigma0; out->w_sigma1 = L.w_sigma1; out->w_color = L.w_color;
    out->g_table = L.g_table; out->g_deform = L.g_deform; out->g_sigma0 = L.g_sigma0; out->g_sigma1 = L.g_sigma1; out->g_color = L.g_color;
    out->xyzs = L.pts; out->dirs = L.pts + (uint64_t)M * 12; out->deltas = L.pts + (uint64_t)M * 24; out->rays = L.rays;
    out->sample_set_stride = L.set_stride;
    out->sigmas = L.sigmas; out->weights_sum = L.weights_sum; out->depth = L.depth; out->image = L.image;
    out->found_inf = L.hyper + offsetof(Hyper, found_inf);
    return 0;
}

int sdn_train_refresh(const SdnTrainStep *s, void *stream) {
    if (!s || !s->workspace || ((uintptr_t)s->workspace & 255u)) return SDN_E_BADARG;
    for (int i = 0; i < SDN_TRAIN_N_PARAMS; i++) if (!s->params[i].param) return SDN_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (int rc = table_wait(s, st)) return rc;
    const Layout L = make_layout(s->N, s->M, s->max_steps, (uint64_t)s->grid_offsets[kLevels] * 2);
    unsigned char *ws = (unsigned char *)s->workspace;
    // zero padding of the flat networks, and the table's gradient accumulator
    if (hipMemsetAsync(ws + L.w_deform, 0, kDefFlat * 2, st) != hipSuccess || hipMemsetAsync(ws + L.w_color, 0, kColFlat * 2, st) != hipSuccess ||
        hipMemsetAsync(ws + L.g_table, 0, (uint64_t)s->grid_offsets[kLevels] * 4, st) != hipSuccess || hipMemsetAsync(ws + L.hyper, 0, sizeof(Hyper), st) != hipSuccess)
        return sdn_launch_status();
    AdamArgs A{};
    const uint32_t blocks = build_segments(s, L, A);
    hipLaunchKernelGGL(k_train_copy16, dim3(blocks), dim3(256), 0, st, A);
    return sdn_launch_status();
}

int sdn_train_flush(const SdnTrainStep *s, void *stream) {
    if (!s) return SDN_E_BADARG;
    return table_wait(s, (hipStream_t)stream);
}

int sdn_train_step_f16(const SdnTrainStep *s, void *stream) {
    if (!step_ok(s)) return SDN_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t N = s->N, M = s->M;
    const Layout L = make_layout(N, M, s->max_steps, (uint64_t)s->grid_offsets[kLevels] * 2);
    unsigned char *ws = (unsigned char *)s->workspace;
    auto F = [&](uint64_t off) { return (float *)(ws + off); };
    auto H = [&](uint64_t off) { return (_Float16 *)(ws + off); };
    Hyper *hyper = (Hyper *)(ws + L.hyper);
    const int zero_deform = s->time == 0.0f;
    const int no_deform_grad = zero_deform || s->deform_frozen;           // nothing flows back through x + deform
    const int freeze_deform = (zero_deform && !s->keep_deform) || s->deform_frozen;
    int rc;
    #define SDN_TRY(x) do { rc = (x); if (rc) return rc; } while (0)
    const uint64_t set_off = s->sample_set ? L.set_stride : 0;
    float *xyzs = F(L.pts + set_off), *dirs = xyzs + (size_t)M * 3, *deltas = xyzs + (size_t)M * 6;
    const int32_t *ray_table = (const int32_t *)(ws + L.rays + set_off);
    if (s->mode != 2 && s->phase != 2) {
        // ---- rays -> samples (renderer.py:283-304) --------------------------------------------------------------------------------
        hipLaunchKernelGGL(k_train_rays, dim3(sdn_div_up(N, 256u)), dim3(256), 0, st, F(L.noises + set_off), s->noises, N, s->noise_seed, s->perturb, s->counter);
        SDN_TRY(sdn_near_far_from_aabb(s->rays_o, s->rays_d, s->aabb, N, s->min_near, F(L.nears + set_off), F(L.fars + set_off), st));
        if (hipMemsetAsync(ws + L.pts + set_off, 0, (uint64_t)M * 32, st) != hipSuccess) return sdn_launch_status();
        SDN_TRY(sdn_int::march_rays_train(s->rays_o, s->rays_d, s->bitfield, s->bound, s->dt_gamma, s->max_steps, N, s->cascade, s->grid_size, M,
                                          F(L.nears + set_off), F(L.fars + set_off), xyzs, dirs, deltas, (int32_t *)(ws + L.rays + set_off), s->counter,
                                          F(L.noises + set_off), ws + L.march + set_off, s->cull_grid, st));
    }
    if (s->phase == 1) return sdn_launch_status();
    if (s->mode != 2) {
    // (dcol_out .. dh0 are cleared by k_train_encode below: both buffers are padded to 256 bytes, so rounding the range up to 16 is safe)
    const uint32_t zero_n16 = (uint32_t)(((L.dh0 - L.dcol_out) + (uint64_t)M * 2 + 15u) / 16u);
    static_assert(sizeof(uint4) == 16, "");
    if (s->mode == 1) {
        SDN_TRY(table_wait(s, st));
        if (hipMemsetAsync(ws + L.g_table, 0, (uint64_t)s->grid_offsets[kLevels] * 4, st) != hipSuccess) return sdn_launch_status();
    }

    // ---- this step's packed weights (both directions of both fused MLPs, one launch) ------------------------------------------------
    const sdn_ffh::PackJob packs[4] = {{ws + L.w_deform, ws + L.pk_def_f, kDefIn, kDefW, kDefL, 0, 1}, {ws + L.w_deform, ws + L.pk_def_b, kDefIn, kDefW, kDefL, 1, 0},
                                       {ws + L.w_color, ws + L.pk_col_f, kColIn, kColW, kColL, 0, 1}, {ws + L.w_color, ws + L.pk_col_b, kColIn, kColW, kColL, 1, 1}};
    SDN_TRY(sdn_ffh::pack_many(packs, 4, st));

    // ---- forward (network.py:123-169) ---------------------------------------------------------------------------------------------
    if (zero_n16 > M * 16u) return SDN_E_BADARG;     // (34 B per sample + padding: never more than one word per thread)
    hipLaunchKernelGGL(k_train_encode, dim3(sdn_div_up(M * 16u, 256u)), dim3(256), 0, st, xyzs, M, s->time, H(L.enc_in), hyper,
                       (uint4 *)(ws + L.dcol_out), zero_n16);
    SDN_TRY(sdn_ffh::forward_packed(H(L.enc_in), ws + L.pk_def_f, M, kDefIn, kDefW, kDefL, ACT_RELU, H(L.def_hidden), H(L.def_out), st));
    hipLaunchKernelGGL(k_train_xdef, dim3(sdn_div_up(M * 3u, 256u)), dim3(256), 0, st, xyzs, H(L.def_out), M, zero_deform, s->bound, F(L.xdef));
    SDN_TRY(table_wait(s, st));       // the previous step's pass over the table (fp16 copy, gradient accumulator), if it ran on table_stream
    SDN_TRY(sdn_grid_encode_forward(F(L.xdef), ws + L.w_table, s->grid_offsets, ws + L.grid_out, M, 3, 2, kLevels, s->grid_S, s->grid_H,
                                    no_deform_grad ? nullptr : ws + L.dy_dx, 1, 0, 0, SDN_F16, st));
    const SigmaFwd sf{H(L.grid_out), H(L.w_sigma0), H(L.w_sigma1), dirs, H(L.enc_rm), H(L.h1), H(L.hout), H(L.col_in), F(L.sigmas), M, s->density_scale};
    hipLaunchKernelGGL(k_train_sigma_fwd, dim3(sdn_div_up(M, 32u)), dim3(64), 0, st, sf);
    SDN_TRY(sdn_ffh::forward_packed(H(L.col_in), ws + L.pk_col_f, M, kColIn, kColW, kColL, ACT_RELU, H(L.col_hidden), H(L.col_out), st));
    // ---- compositing, loss, and their gradients (renderer.py:309-318, utils.py:85-125), one wave per ray ----------------------------
    const CompositeArgs ca{F(L.sigmas), deltas, H(L.col_out), H(L.hout), ray_table, s->bg_color, s->target, s->loss_scale, F(L.weights_sum),
                           F(L.depth), F(L.image), s->image_out, F(L.sq_err), H(L.dcol_out), H(L.dh0), M, N, s->T_thresh, s->bg_value, s->density_scale};
    hipLaunchKernelGGL(k_train_composite_fwd, dim3(sdn_div_up(N, 4u)), dim3(256), 0, st, ca);
    hipLaunchKernelGGL(k_train_loss, dim3(1), dim3(1024), 0, st, F(L.sq_err), N, s->loss_out);

    // ---- backward through the field ---------------------------------------------------------------------------------------------------
    hipLaunchKernelGGL(k_train_composite_bwd, dim3(sdn_div_up(N, 4u)), dim3(256), 0, st, ca);
    SDN_TRY(sdn_ffh::backward_packed(H(L.dcol_out), ws + L.pk_col_b, H(L.col_hidden), M, kColIn, kColW, kColL, ACT_RELU, 1, H(L.col_bwd), H(L.dcol_in), st));
    const SigmaBwd sb{H(L.dh0), H(L.dcol_in), H(L.h1), H(L.w_sigma0), H(L.w_sigma1), H(L.dh), H(L.dh1), H(L.denc), M};
    hipLaunchKernelGGL(k_train_sigma_bwd, dim3(sdn_div_up(M, 32u)), dim3(64), 0, st, sb);
    SDN_TRY(sdn_grid_encode_backward_det(ws + L.denc, F(L.xdef), s->grid_offsets, ws + L.g_table, M, 3, 2, kLevels, s->grid_S, s->grid_H,
                                         no_deform_grad ? nullptr : ws + L.dy_dx, no_deform_grad ? nullptr : ws + L.dx16, 1, 0, 0, SDN_F16,
                                         s->det_scratch, st));
    sdn_ffh::DwJob jobs[16];
    uint32_t nj = 0;
    dw_job_list(L, ws, M, jobs, nj);
    if (!no_deform_grad) {
        hipLaunchKernelGGL(k_train_deform_grad, dim3(sdn_div_up(M, 256u)), dim3(256), 0, st, H(L.dx16), M, s->bound, H(L.ddef));
        SDN_TRY(sdn_ffh::backward_packed(H(L.ddef), ws + L.pk_def_b, H(L.def_hidden), M, kDefIn, kDefW, kDefL, ACT_RELU, 0, H(L.def_bwd), nullptr, st));
        SDN_TRY(sdn_ffh::dw_jobs(jobs, nj, M, ws + L.dw_partial, st));
    } else {
        SDN_TRY(sdn_ffh::dw_jobs(jobs + (kDefL + 1), nj - (kDefL + 1), M, ws + L.dw_partial, st));   // canonical frame: the deformation MLP has no gradient
    }
    if (no_deform_grad && s->keep_deform && !s->deform_frozen && hipMemsetAsync(ws + L.g_deform, 0, kDefFlat * 2, st) != hipSuccess) return sdn_launch_status();
    }   // mode != 2
    if (s->mode == 1) return sdn_launch_status();

    // ---- optimizer (nerf/utils.py:889-906) --------------------------------------------------------------------------------------------
    const uint32_t table_n = (uint32_t)s->grid_offsets[kLevels] * 2u;
    CheckArgs ck{};
    ck.hyper = hyper;
    ck.g[0] = H(L.g_table); ck.n8[0] = table_n / 8;
    ck.g[1] = H(L.g_sigma0); ck.n8[1] = kSigW * kSigIn / 8;
    ck.g[2] = H(L.g_sigma1); ck.n8[2] = kSigOut * kSigW / 8;
    ck.g[3] = H(L.g_color); ck.n8[3] = kColFlat / 8; ck.col_seg = 3;
    ck.count = 4;
    if (!freeze_deform) { ck.g[4] = H(L.g_deform); ck.n8[4] = kDefFlat / 8; ck.count = 5; }
    const Prologue pr{hyper, s->adam_steps, s->loss_scale, s->growth_tracker, s->beta1, s->beta2, s->lr_table, s->lr_net, s->growth_factor, s->backoff_factor,
                      s->growth_interval, !freeze_deform, s->grad_divisor > 0.0f ? s->grad_divisor : 1.0f};
    hipLaunchKernelGGL(k_train_check, dim3(1024), dim3(256), 0, st, ck);
    hipLaunchKernelGGL(k_train_prologue, dim3(1), dim3(1), 0, st, pr);
    AdamArgs A{};
    const uint32_t blocks = build_segments(s, L, A);
    if (freeze_deform) for (uint32_t i = 1; i <= kDefL + 1; i++) A.seg[i].frozen = 1;
    A.hyper = hyper;
    A.one_minus_b1 = (float)(1.0 - s->beta1); A.b2 = (float)s->beta2; A.one_minus_b2 = (float)(1.0 - s->beta2); A.eps = (float)s->eps;
    A.ema_keep = 1.0f - s->ema_decay;
    (void)blocks;
    if (s->table_stream) {
        if (s->mode == 2) SDN_TRY(table_wait(s, st));           // (optimizer-only call: nothing above waited)
        hipStream_t ts = (hipStream_t)s->table_stream;
        if (hipEventRecord((hipEvent_t)s->table_ready, st) != hipSuccess || hipStreamWaitEvent(ts, (hipEvent_t)s->table_ready, 0) != hipSuccess)
            return sdn_launch_status();
        launch_adam(A, 0, 1, ts);                                // the table: 366 MB of traffic, beside whatever `stream` runs next
        if (hipEventRecord((hipEvent_t)s->table_done, ts) != hipSuccess) return sdn_launch_status();
        launch_adam(A, 1, SDN_TRAIN_N_PARAMS, st);               // the MLPs (the next step's first kernels read their fp16 copies)
    } else {
        launch_adam(A, 0, SDN_TRAIN_N_PARAMS, st);
    }
    #undef SDN_TRY
    return sdn_launch_status();
}

}  // extern "C"
