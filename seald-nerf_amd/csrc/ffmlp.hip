// Fully fused MLP operator on MFMA for gfx950 (MI355X): forward / inference / backward of the reference's `ffmlp`
// extension (ffmlp/src/ffmlp.h:8-14, ffmlp/src/ffmlp.cu:331-407 forward, :411-523 backward, :770-894 weight gradients).
//
// Behavioural contract (ffmlp/ffmlp.py:100-168): a bias-free MLP  in -> hidden (x num_layers) -> 16  on fp16 tensors,
//   flat fp16 weights, row-major [hidden, in] ++ (num_layers - 1) x [hidden, hidden] ++ [16, hidden]   (ffmlp.cu:631),
//   inputs [B, in], outputs [B, 16] point-major (the reference's "col-major [dim, B]"), hidden in {16,32,64,128,256},
//   in % 16 == 0, activation on every hidden layer (ReLU / exp / sine / sigmoid / squareplus / softplus / none,
//   utils.h:425-470), none on the output.  The training forward keeps the num_layers post-activation tensors
//   (forward_buffer), the backward derives the activation gradient from them (utils.h:538-583) and keeps the
//   pre-activation gradients (backward_buffer) for the weight-gradient products  dW_l = G_l^T X_l.
// Numerics: products accumulate in fp32 on the matrix cores and are rounded to fp16 once per layer output (the reference
// accumulates in fp16 wmma fragments, which is hardware-defined and cannot be restated; see oracle/ffmlp.py header).
//
// Mapping -- nothing here is a translation of the wmma / CUTLASS structure:
//   * weights are the A operand of v_mfma_f32_32x32x16_f16 (M = output features), activations the B operand
//     (N = 32 points, K = input features).  A layer's accumulator (feature rows in registers, point on the lane) converts in
//     place (fp16 round, activation) into the B fragments of the next layer, so activations never leave registers
//     (the reference round-trips them through shared memory between layers).  The k order this implies
//     (element j of lane-half h <-> feature 16s + 8(j>>2) + 4h + (j&3)) is produced by a tiny pack kernel that
//     rewrites the flat weights into 1-KiB lane-linear fragments ([layer][M-tile][k-step]; transposed for backward).
//   * one wave = NT x 32 points (NT = 2 up to hidden 128: every A fragment read from LDS feeds two MFMAs),
//     workgroup = 4 waves; the packed fragments stream through LDS in <= 32 KiB stages, double-buffered with
//     direct-to-LDS loads (global_load_lds_dwordx4), one barrier per stage.
//   * weight gradients: split-K MFMA kernel, batch rows staged row-major in LDS and read column-wise with the gfx950
//     transposing LDS read (ds_read_b64_tr_b16), fp32 partials per split + a deterministic reduction (no atomics).
#include "ffmlp_kernels.h"
#include "sdn_internal.h"

namespace sdn_ff {
// instantiated in ffmlp_act.hip: every activation other than ReLU (run-time dispatch inside the kernels)
int launch_fused_generic(int mode, uint32_t W, const FfArgs &a, hipStream_t st);
}  // namespace sdn_ff

namespace {

using namespace sdn_ff;

// ---- weight packing ---------------------------------------------------------------------------------------------------------------
struct PackArgs {
    const _Float16 *w;
    unsigned char *packed;
    uint32_t in_dim, W, L;
    int backward;
    int with_last;       // backward: pack W0^T for grad_inputs
};

struct LayerDesc { uint32_t M, K, off; uint32_t sm, sk; };

// layer i of the consumption order (forward: 0..L; backward: 0..L, reversed and transposed)
__device__ __forceinline__ LayerDesc layer_desc(const PackArgs &P, uint32_t i) {
    const uint32_t W = P.W, in = P.in_dim, L = P.L;
    const uint32_t off_mid = W * in, off_out = W * in + (L - 1) * W * W;
    LayerDesc d;
    if (!P.backward) {
        if (i == 0) d = {W, in, 0, in, 1};
        else if (i < L) d = {W, W, off_mid + (i - 1) * W * W, W, 1};
        else d = {16, W, off_out, W, 1};
    } else {
        if (i == 0) d = {W, 16, off_out, 1, W};                              // A[m][k] = Wout[k][m]
        else if (i < L) d = {W, W, off_mid + (L - 1 - i) * W * W, 1, W};     // A = W_{L-i}^T
        else d = {in, W, 0, 1, in};                                          // A = W0^T
    }
    return d;
}

__device__ __forceinline__ void pack_frag(const PackArgs &P, uint32_t frag, uint32_t lane) {
    const uint32_t dst = frag;
    const uint32_t nlayers = P.L + ((P.backward && !P.with_last) ? 0u : 1u);
    LayerDesc d{};
    for (uint32_t i = 0; i < nlayers; i++) {
        d = layer_desc(P, i);
        const uint32_t nf = ((d.M + 31) / 32) * (d.K / 16);
        if (frag < nf) break;
        frag -= nf;
    }
    const uint32_t KSl = d.K / 16, Mt = frag / KSl, s = frag % KSl;
    const uint32_t m = 32 * Mt + (lane & 31u), hh = lane >> 5;
    half8 v;
    #pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t kk = 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3);
        v[j] = m < d.M ? P.w[d.off + (size_t)m * d.sm + (size_t)kk * d.sk] : (_Float16)0;
    }
    *reinterpret_cast<half8 *>(P.packed + (size_t)dst * 1024 + lane * 16) = v;
}

__global__ void k_ffmlp_pack(PackArgs P, uint32_t total_frags) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if ((gid >> 6) >= total_frags) return;
    pack_frag(P, gid >> 6, gid & 63u);
}

// several networks / directions in one launch (blockIdx.y = job): the training step re-packs both directions of its MLPs once per step
constexpr int kMaxPackJobs = 8;
struct PackMany { PackArgs a[kMaxPackJobs]; uint32_t nf[kMaxPackJobs]; };

__global__ void k_ffmlp_pack_many(PackMany Q) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t job = blockIdx.y;
    if ((gid >> 6) >= Q.nf[job]) return;
    pack_frag(Q.a[job], gid >> 6, gid & 63u);
}

uint32_t total_frags_host(uint32_t in_dim, uint32_t W, uint32_t L, int backward, int with_last) {
    const uint32_t MT = (W + 31) / 32, KS = W / 16;
    if (!backward) return MT * (in_dim / 16) + (L - 1) * MT * KS + 1 * KS;
    return MT * 1 + (L - 1) * MT * KS + (with_last ? ((in_dim + 31) / 32) * KS : 0);
}

// ---- weight gradients: dW[m][n] = sum_b G[b][m] X[b][n] ---------------------------------------------------------------------------
constexpr int kDwRows = 64;                     // batch rows per LDS tile
constexpr int kDwStride = (128 + 32) * 2;       // bytes per LDS row: 128 halfs + 64 B pad (4 rows x 32 B blocks of a transposed read land on disjoint banks)

struct DwArgs {
    const _Float16 *G;   // [B, ldg]
    const _Float16 *X;   // [B, ldx]
    float *partial;      // [nsplit][Mpad][Npad]
    uint32_t B, ldg, ldx, M, N, Mpad, Npad, rows_per_split;
};

__device__ __forceinline__ half8 tr_frag(const unsigned char *tile, int s, int c0, uint32_t lane) {
    // 32x32x16 operand with k = batch row 16s + 8h + j and row/col = feature c0 + (lane & 31), from a row-major [row][feature] image
    const uint32_t g = lane >> 4, i = lane & 15u, q = i >> 2, p = i & 3u;
    const uint32_t row = 16u * s + 8u * (g >> 1) + q;
    const unsigned char *a = tile + row * kDwStride + (c0 + 16u * (g & 1u) + 4u * p) * 2u;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(a));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(a + 4 * kDwStride));
    const half4 l4 = __builtin_bit_cast(half4, lo), h4 = __builtin_bit_cast(half4, hi);
    return half8{l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
}

// one workgroup: split `split` of the batch, output block `block` of one layer's dW
__device__ __forceinline__ void dw_block(const DwArgs &P, uint32_t split, uint32_t block, unsigned char *s_g, unsigned char *s_x) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t nbx = (P.N + 127) / 128;
    const uint32_t mb = (block / nbx) * 128, nb = (block % nbx) * 128;             // origin of this block's <= 128 x 128 output
    const uint32_t mcols = min(128u, P.Mpad - mb), ncols = min(128u, P.Npad - nb); // multiples of 32
    const uint32_t wy = wave >> 1, wx = wave & 1u;
    const uint32_t b_begin = split * P.rows_per_split;
    const uint32_t b_end = min(P.B, b_begin + P.rows_per_split);

    f32x16 acc[2][2];
    #pragma unroll
    for (int a = 0; a < 2; a++)
        #pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const bool my_m[2] = {64 * wy < mcols, 64 * wy + 32 < mcols};
    const bool my_n[2] = {64 * wx < ncols, 64 * wx + 32 < ncols};

    // 64 rows x 16 chunks of 16 B per operand = 1024 chunks, 4 per thread
    half8 rg[4], rx[4];
    auto fetch = [&](uint32_t b0) {
        #pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t idx = threadIdx.x + 256u * u, row = idx >> 4, ch = idx & 15u;
            const uint32_t b = b0 + row;
            const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            rg[u] = (b < b_end && ch * 8 < mcols && mb + ch * 8 < P.M) ? *reinterpret_cast<const half8 *>(P.G + (size_t)b * P.ldg + mb + ch * 8) : z;
            rx[u] = (b < b_end && ch * 8 < ncols && nb + ch * 8 < P.N) ? *reinterpret_cast<const half8 *>(P.X + (size_t)b * P.ldx + nb + ch * 8) : z;
        }
    };
    auto stash = [&]() {
        #pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t idx = threadIdx.x + 256u * u, row = idx >> 4, ch = idx & 15u;
            *reinterpret_cast<half8 *>(s_g + row * kDwStride + ch * 16) = rg[u];
            *reinterpret_cast<half8 *>(s_x + row * kDwStride + ch * 16) = rx[u];
        }
    };

    if (b_begin < b_end) fetch(b_begin);
    for (uint32_t b0 = b_begin; b0 < b_end; b0 += kDwRows) {
        __syncthreads();            // previous tile fully consumed
        stash();
        __syncthreads();
        if (b0 + kDwRows < b_end) fetch(b0 + kDwRows);
        #pragma unroll
        for (int s = 0; s < kDwRows / 16; s++) {
            half8 fa[2], fb[2];
            #pragma unroll
            for (int a = 0; a < 2; a++) fa[a] = tr_frag(s_g, s, 64 * wy + 32 * a, lane);
            #pragma unroll
            for (int b = 0; b < 2; b++) fb[b] = tr_frag(s_x, s, 64 * wx + 32 * b, lane);
            #pragma unroll
            for (int a = 0; a < 2; a++)
                #pragma unroll
                for (int b = 0; b < 2; b++)
                    if (my_m[a] && my_n[b]) acc[a][b] = mfma(fa[a], fb[b], acc[a][b]);
        }
    }
    const uint32_t n = lane & 31u, h = lane >> 5;
    float *dst = P.partial + (size_t)split * P.Mpad * P.Npad;
    #pragma unroll
    for (int a = 0; a < 2; a++)
        #pragma unroll
        for (int b = 0; b < 2; b++) {
            if (!(my_m[a] && my_n[b])) continue;
            #pragma unroll
            for (int i = 0; i < 16; i++) {
                const uint32_t m = mb + 64 * wy + 32 * a + 8 * (i >> 2) + 4 * h + (i & 3);
                dst[(size_t)m * P.Npad + nb + 64 * wx + 32 * b + n] = acc[a][b][i];
            }
        }
}

__global__ void __launch_bounds__(256) k_ffmlp_dw(DwArgs P) {
    __shared__ __attribute__((aligned(16))) unsigned char s_g[kDwRows * kDwStride];
    __shared__ __attribute__((aligned(16))) unsigned char s_x[kDwRows * kDwStride];
    dw_block(P, blockIdx.x, blockIdx.y, s_g, s_x);
}

// Every layer's weight gradient in ONE launch (blockIdx.z = layer), each with its own partial-sum region: a training batch of a few
// thousand samples is launch-bound here (2 x (L + 1) launches of < 20 us each for the deformation MLP of the dnerf network).
constexpr int kMaxDw = 17;
struct DwBatch {
    DwArgs a[kMaxDw];
    _Float16 *out[kMaxDw];
    uint32_t nsplit[kMaxDw], blocks[kMaxDw];
};

__global__ void __launch_bounds__(256) k_ffmlp_dw_batched(DwBatch Bt) {
    __shared__ __attribute__((aligned(16))) unsigned char s_g[kDwRows * kDwStride];
    __shared__ __attribute__((aligned(16))) unsigned char s_x[kDwRows * kDwStride];
    const uint32_t z = blockIdx.z;
    if (blockIdx.x >= Bt.nsplit[z] || blockIdx.y >= Bt.blocks[z]) return;   // workgroup-uniform, before any barrier
    dw_block(Bt.a[z], blockIdx.x, blockIdx.y, s_g, s_x);
}

// deterministic second pass: element (m, n) = sum over the splits, 4 partial sums per element combined through LDS
__device__ __forceinline__ void dw_reduce_block(const float *partial, uint32_t nsplit, uint32_t M, uint32_t N, uint32_t Mpad, uint32_t Npad,
                                                _Float16 *out, float (*s_part)[64]) {
    const uint32_t x = threadIdx.x & 63u, y = threadIdx.x >> 6;
    const uint32_t idx = blockIdx.x * 64u + x;
    float s = 0.0f;
    if (idx < M * N) {
        const uint32_t m = idx / N, n = idx % N;
        for (uint32_t k = y; k < nsplit; k += 4) s += partial[((size_t)k * Mpad + m) * Npad + n];
    }
    s_part[y][x] = s;
    __syncthreads();
    if (y == 0 && idx < M * N) out[idx] = (_Float16)((s_part[0][x] + s_part[1][x]) + (s_part[2][x] + s_part[3][x]));
}

__global__ void __launch_bounds__(256) k_ffmlp_dw_reduce(const float *partial, uint32_t nsplit, uint32_t M, uint32_t N, uint32_t Mpad, uint32_t Npad, _Float16 *out) {
    __shared__ float s_part[4][64];
    dw_reduce_block(partial, nsplit, M, N, Mpad, Npad, out, s_part);
}

__global__ void __launch_bounds__(256) k_ffmlp_dw_reduce_batched(DwBatch Bt) {
    __shared__ float s_part[4][64];
    const DwArgs &P = Bt.a[blockIdx.y];
    if (blockIdx.x * 64u >= P.M * P.N) return;                               // workgroup-uniform
    dw_reduce_block(P.partial, Bt.nsplit[blockIdx.y], P.M, P.N, P.Mpad, P.Npad, Bt.out[blockIdx.y], s_part);
}

uint32_t dw_splits(uint32_t B, uint32_t blocks) {
    // 256 batch rows (4 LDS tiles) per workgroup: a training batch of ~9000 samples then spreads over 36 CUs instead of 9
    // (measured 17 us per layer with 1024-row splits, the whole weight-gradient pass of the dnerf deformation MLP 190 us)
    uint32_t ns = sdn_div_up(B, 256u);
    const uint32_t cap = 512u / blocks;       // ~2 workgroups per CU in flight; the second pass reads ns partial tiles per element
    if (ns > cap) ns = cap;
    return ns ? ns : 1;
}

uint32_t dw_blocks(uint32_t M, uint32_t N) { return (((M + 31) / 32 * 32 + 127) / 128) * (((N + 31) / 32 * 32 + 127) / 128); }

uint64_t dw_partial_bytes(uint32_t B, uint32_t M, uint32_t N) {
    return (uint64_t)dw_splits(B, dw_blocks(M, N)) * ((M + 31) / 32 * 32) * ((N + 31) / 32 * 32) * sizeof(float);
}

// All layers' partial sums side by side (one launch for every dW) when that stays small; else one region reused layer by layer.
constexpr uint64_t kDwBatchedBudget = 64ull << 20;
uint64_t dw_batched_bytes(uint32_t B, uint32_t in_dim, uint32_t W, uint32_t L) {
    return dw_partial_bytes(B, 16, W) + (uint64_t)(L - 1) * dw_partial_bytes(B, W, W) + dw_partial_bytes(B, W, in_dim);
}
bool dw_batched(uint32_t B, uint32_t in_dim, uint32_t W, uint32_t L) {
    return L + 1 <= (uint32_t)kMaxDw && dw_batched_bytes(B, in_dim, W, L) <= kDwBatchedBudget;
}

bool dims_ok(uint32_t in_dim, uint32_t out_dim, uint32_t W, uint32_t L) {
    const bool wok = W == 16 || W == 32 || W == 64 || W == 128 || W == 256;
    return wok && in_dim >= 16 && in_dim % 16 == 0 && in_dim <= 512 && out_dim == 16 && L >= 2;
}

// ReLU has its own instantiation (no dispatch in the epilogues); everything else goes through ffmlp_act.hip
template <int MODE>
int launch_fused(uint32_t W, const FfArgs &a, hipStream_t st) {
    return a.act == ACT_RELU ? launch_fused_t<MODE, true>(W, a, st) : launch_fused_generic(MODE, W, a, st);
}

int pack(const void *weights, void *packed, uint32_t in_dim, uint32_t W, uint32_t L, int backward, int with_last, hipStream_t st) {
    PackArgs p{(const _Float16 *)weights, (unsigned char *)packed, in_dim, W, L, backward, with_last};
    const uint32_t nf = total_frags_host(in_dim, W, L, backward, with_last);
    hipLaunchKernelGGL(k_ffmlp_pack, dim3(sdn_div_up(nf * 64u, 256u)), dim3(256), 0, st, p, nf);
    return sdn_launch_status();
}

int forward_common(const void *inputs, const void *weights, uint32_t B, uint32_t in_dim, uint32_t out_dim, uint32_t W, uint32_t L,
                   uint32_t act, uint32_t out_act, void *forward_buffer, void *outputs, void *scratch, hipStream_t st) {
    if (B == 0) return 0;
    if (!inputs || !weights || !outputs || !scratch) return SDN_E_BADARG;
    if (!dims_ok(in_dim, out_dim, W, L) || act > ACT_NONE || out_act != ACT_NONE) return SDN_E_UNSUPPORTED;
    if (((uintptr_t)scratch & 15u) || ((uintptr_t)inputs & 7u) || ((uintptr_t)outputs & 7u) || ((uintptr_t)forward_buffer & 7u)) return SDN_E_BADARG;
    int rc = pack(weights, scratch, in_dim, W, L, 0, 1, st);
    if (rc) return rc;
    FfArgs a{(const _Float16 *)inputs, (const unsigned char *)scratch, (_Float16 *)forward_buffer, nullptr, (_Float16 *)outputs, B, in_dim, 16, L, act,
             total_frags_host(in_dim, W, L, 0, 1)};
    return forward_buffer ? launch_fused<1>(W, a, st) : launch_fused<0>(W, a, st);
}

}  // namespace

namespace sdn_ffh {

uint32_t total_frags(uint32_t in_dim, uint32_t W, uint32_t L, int backward, int with_last) { return total_frags_host(in_dim, W, L, backward, with_last); }

int pack_many(const PackJob *jobs, uint32_t n, hipStream_t st) {
    if (n == 0) return 0;
    if (n > (uint32_t)kMaxPackJobs) return SDN_E_UNSUPPORTED;
    PackMany q;
    uint32_t max_nf = 1;
    for (uint32_t i = 0; i < (uint32_t)kMaxPackJobs; i++) {
        const PackJob &j = jobs[i < n ? i : 0];
        q.a[i] = PackArgs{(const _Float16 *)j.weights, (unsigned char *)j.packed, j.in_dim, j.W, j.L, j.backward, j.with_last};
        q.nf[i] = i < n ? total_frags_host(j.in_dim, j.W, j.L, j.backward, j.with_last) : 0;
        if (q.nf[i] > max_nf) max_nf = q.nf[i];
    }
    hipLaunchKernelGGL(k_ffmlp_pack_many, dim3(sdn_div_up(max_nf * 64u, 256u), n), dim3(256), 0, st, q);
    return sdn_launch_status();
}

int forward_packed(const void *inputs, const void *packed, uint32_t B, uint32_t in_dim, uint32_t W, uint32_t L, uint32_t act,
                   void *forward_buffer, void *outputs, hipStream_t st) {
    if (B == 0) return 0;
    if (!dims_ok(in_dim, 16, W, L) || act > ACT_NONE) return SDN_E_UNSUPPORTED;
    FfArgs a{(const _Float16 *)inputs, (const unsigned char *)packed, (_Float16 *)forward_buffer, nullptr, (_Float16 *)outputs, B, in_dim, 16, L, act,
             total_frags_host(in_dim, W, L, 0, 1)};
    return forward_buffer ? launch_fused<1>(W, a, st) : launch_fused<0>(W, a, st);
}

int backward_packed(const void *grad, const void *packed, const void *forward_buffer, uint32_t B, uint32_t in_dim, uint32_t W, uint32_t L,
                    uint32_t act, int want_dx, void *backward_buffer, void *grad_inputs, hipStream_t st) {
    if (B == 0) return 0;
    if (!dims_ok(in_dim, 16, W, L) || act > ACT_NONE || act == ACT_SINE) return SDN_E_UNSUPPORTED;
    FfArgs a{(const _Float16 *)grad, (const unsigned char *)packed, (_Float16 *)backward_buffer, (const _Float16 *)forward_buffer,
             want_dx ? (_Float16 *)grad_inputs : nullptr, B, 16, in_dim, L, act, total_frags_host(in_dim, W, L, 1, want_dx ? 1 : 0)};
    return launch_fused<2>(W, a, st);
}

uint64_t dw_jobs_bytes(const DwJob *jobs, uint32_t n, uint32_t B) {
    uint64_t t = 0;
    for (uint32_t i = 0; i < n; i++) t += dw_partial_bytes(B, jobs[i].M, jobs[i].N);
    return t;
}

// every job's split-K partial sums in ONE launch (blockIdx.z = job) + one deterministic reduction launch
int dw_jobs(const DwJob *jobs, uint32_t n, uint32_t B, void *partial, hipStream_t st) {
    if (n == 0 || B == 0) return 0;
    if (n > (uint32_t)kMaxDw) return SDN_E_UNSUPPORTED;
    DwBatch bt;
    uint32_t max_ns = 1, max_blocks = 1, max_red = 1;
    float *next = (float *)partial;
    for (uint32_t i = 0; i < n; i++) {
        const DwJob &j = jobs[i];
        DwArgs d;
        d.G = (const _Float16 *)j.G; d.X = (const _Float16 *)j.X; d.partial = next; d.B = B; d.ldg = j.ldg; d.ldx = j.ldx; d.M = j.M; d.N = j.N;
        d.Mpad = (j.M + 31) / 32 * 32; d.Npad = (j.N + 31) / 32 * 32;
        const uint32_t blocks = dw_blocks(j.M, j.N);
        const uint32_t ns = dw_splits(B, blocks);
        d.rows_per_split = sdn_div_up(sdn_div_up(B, ns), (uint32_t)kDwRows) * kDwRows;
        const uint32_t ns_used = sdn_div_up(B, d.rows_per_split);
        bt.a[i] = d; bt.out[i] = (_Float16 *)j.out; bt.nsplit[i] = ns_used; bt.blocks[i] = blocks;
        next += dw_partial_bytes(B, j.M, j.N) / sizeof(float);
        if (ns_used > max_ns) max_ns = ns_used;
        if (blocks > max_blocks) max_blocks = blocks;
        if (sdn_div_up(j.M * j.N, 64u) > max_red) max_red = sdn_div_up(j.M * j.N, 64u);
    }
    for (uint32_t k = n; k < (uint32_t)kMaxDw; k++) { bt.a[k] = bt.a[0]; bt.out[k] = nullptr; bt.nsplit[k] = 0; bt.blocks[k] = 0; }
    hipLaunchKernelGGL(k_ffmlp_dw_batched, dim3(max_ns, max_blocks, n), dim3(256), 0, st, bt);
    hipLaunchKernelGGL(k_ffmlp_dw_reduce_batched, dim3(max_red, n), dim3(256), 0, st, bt);
    return sdn_launch_status();
}

}  // namespace sdn_ffh

extern "C" {

uint64_t sdn_ffmlp_scratch_bytes(uint32_t B, uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers) {
    if (!dims_ok(input_dim, output_dim, hidden_dim, num_layers)) return 0;
    const uint64_t fw = total_frags_host(input_dim, hidden_dim, num_layers, 0, 1), bw = total_frags_host(input_dim, hidden_dim, num_layers, 1, 1);
    const uint64_t packed = (fw > bw ? fw : bw) * 1024;
    uint64_t partial = dw_partial_bytes(B, 16, hidden_dim);
    const uint64_t p1 = dw_partial_bytes(B, hidden_dim, hidden_dim), p2 = dw_partial_bytes(B, hidden_dim, input_dim);
    if (p1 > partial) partial = p1;
    if (p2 > partial) partial = p2;
    if (dw_batched(B, input_dim, hidden_dim, num_layers)) partial = dw_batched_bytes(B, input_dim, hidden_dim, num_layers);
    return packed + partial;
}

int sdn_ffmlp_forward(const void *inputs, const void *weights, uint32_t B, uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim,
                      uint32_t num_layers, uint32_t activation, uint32_t output_activation, void *forward_buffer, void *outputs,
                      void *scratch, void *stream) {
    if (B && !forward_buffer) return SDN_E_BADARG;
    return forward_common(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, forward_buffer,
                          outputs, scratch, (hipStream_t)stream);
}

int sdn_ffmlp_inference(const void *inputs, const void *weights, uint32_t B, uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim,
                        uint32_t num_layers, uint32_t activation, uint32_t output_activation, void *outputs, void *scratch, void *stream) {
    return forward_common(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, nullptr, outputs,
                          scratch, (hipStream_t)stream);
}

int sdn_ffmlp_backward(const void *grad, const void *inputs, const void *weights, const void *forward_buffer, uint32_t B, uint32_t input_dim,
                       uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation,
                       int calc_grad_inputs, void *backward_buffer, void *grad_inputs, void *grad_weights, void *scratch, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const uint32_t W = hidden_dim, L = num_layers, in = input_dim;
    if (B == 0) return 0;
    if (!grad || !inputs || !weights || !forward_buffer || !backward_buffer || !grad_weights || !scratch) return SDN_E_BADARG;
    if (calc_grad_inputs && !grad_inputs) return SDN_E_BADARG;
    if (!dims_ok(in, output_dim, W, L) || activation > ACT_NONE || activation == ACT_SINE || output_activation != ACT_NONE) return SDN_E_UNSUPPORTED;
    if ((uintptr_t)scratch & 15u) return SDN_E_BADARG;

    // 1. pre-activation gradients of every hidden layer (+ grad_inputs): one fused launch
    int rc = pack(weights, scratch, in, W, L, 1, calc_grad_inputs ? 1 : 0, st);
    if (rc) return rc;
    FfArgs a{(const _Float16 *)grad, (const unsigned char *)scratch, (_Float16 *)backward_buffer, (const _Float16 *)forward_buffer,
             calc_grad_inputs ? (_Float16 *)grad_inputs : nullptr, B, 16, in, L, activation, total_frags_host(in, W, L, 1, calc_grad_inputs ? 1 : 0)};
    rc = launch_fused<2>(W, a, st);
    if (rc) return rc;

    // 2. weight gradients (split-K over the batch, deterministic two-pass reduction): every layer in one launch when the partial
    //    sums fit, else layer by layer through one region
    const uint64_t fw = total_frags_host(in, W, L, 0, 1), bw = total_frags_host(in, W, L, 1, 1);
    float *partial = (float *)((unsigned char *)scratch + (fw > bw ? fw : bw) * 1024);
    const _Float16 *fwd = (const _Float16 *)forward_buffer, *bwd = (const _Float16 *)backward_buffer;
    _Float16 *gw = (_Float16 *)grad_weights;
    const size_t BW = (size_t)B * W;
    sdn_ffh::DwJob jobs[kMaxDw];
    uint32_t n = 0;
    if (L + 1 > (uint32_t)kMaxDw) return SDN_E_UNSUPPORTED;
    // output layer: dW_out [16, W] = grad^T X_L                                   (ffmlp.cu:800-811)
    jobs[n++] = {grad, 16, 16, fwd + (L - 1) * BW, W, W, gw + (size_t)W * in + (size_t)(L - 1) * W * W};
    // hidden layers: dW_j [W, W] = G_j^T X_j, G_j = backward_buffer[L - 1 - j]   (ffmlp.cu:845-863)
    for (uint32_t j = L - 1; j >= 1; j--)
        jobs[n++] = {bwd + (size_t)(L - 1 - j) * BW, W, W, fwd + (size_t)(j - 1) * BW, W, W, gw + (size_t)W * in + (size_t)(j - 1) * W * W};
    // input layer: dW_0 [W, in] = G_0^T inputs                                    (ffmlp.cu:866-876)
    jobs[n++] = {bwd + (size_t)(L - 1) * BW, W, W, inputs, in, in, gw};
    if (dw_batched(B, in, W, L)) return sdn_ffh::dw_jobs(jobs, n, B, partial, st);
    for (uint32_t k = 0; k < n; k++) {
        rc = sdn_ffh::dw_jobs(jobs + k, 1, B, partial, st);
        if (rc) return rc;
    }
    return 0;
}

}  // extern "C"
