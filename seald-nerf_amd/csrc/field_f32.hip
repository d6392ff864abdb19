// Fused field network in fp32 (the reference WITHOUT `-O`: dnerf/network.py:123-169 in float32) -- sigma and rgb of sample points in
// one launch, fp32 weights, fp32 table, activations on chip.  The fp16 kernel (field.hip) reproduces autocast's half roundings and
// cannot meet a 1e-4 bar against the fp32 network; this one can: every product and sum is fp32 (v_mfma_f32_32x32x2_f32), the
// encoders are the fp32 operators' own expressions (freqencoder.cu:30-58 with its phase-shifted sine, gridencoder.cu:87-245,
// shencoder.cu:49-121 through sh_eval.h).
//
// Mapping.  A workgroup is 4 waves (two workgroups per CU, out of step with each other: one encodes while the other multiplies; 8 waves
// in one workgroup measured 3 % slower), a wave owns 32 points (N of the MFMA) from its encodings to its outputs and never talks to another
// wave; the workgroup shares the WEIGHTS: each layer's A operands are staged once in LDS (64 KiB for a 128 x 128 layer) and read by
// all its waves.  v_mfma_f32_32x32x2_f32 takes A[m = lane % 32][k = lane / 32] and B[k = lane / 32][n = lane % 32] -- one register
// each -- and leaves C[row = 8 (v / 4) + 4 (lane / 32) + v % 4][n = lane % 32] in accumulator register v.  So accumulator register v
// of output tile mt IS the B operand of the next layer for the k-pair (row, row + 4), row = 32 mt + 8 (v / 4) + v % 4: activations never
// leave the register file and are never permuted; the k-order that makes this true is baked into the weight packing
// (dnerf_amd/fused_f32.py, layout [pair][lane][m-tile] so that one 16-byte LDS read feeds the four MFMAs of a k-pair).
// The encodings are produced in the same shape: lane half h = lane / 32 of point n computes feature (pair, h) -- for the frequency
// encoding the pair is (sin, cos) of one angle, i.e. ONE sinf with the reference's phase shift h * pi/2; for the grid it is the two
// channels of a level; for SH consecutive coefficients.
//
// Bound: 1 920 MFMAs of 64 cycles per 32 points = 123 K matrix-pipe cycles per wave-tile (fp32 MFMA peak 157 TFLOP/s dense on MI355X);
// 235 520 FLOP per point as in the fp16 kernel.
#include "field_f32_common.h"

namespace {

using namespace sdn_f32;

// offsets inside the tail stage (floats): [pair][lane][m-tile] blocks   (dnerf_amd/fused_f32.py: pack_weights_f32)
constexpr int kT_D7 = 0, kT_S0 = kT_D7 + 64 * 64, kT_S1 = kT_S0 + 16 * 64 * 2, kT_C0 = kT_S1 + 32 * 64, kT_C1 = kT_C0 + 16 * 64 * 2,
              kT_C2 = kT_C1 + 32 * 64 * 2;
static_assert(kT_C2 + 32 * 64 == kTailFloats && 32 * 64 * 4 == kD0Floats, "stage sizes");

// one layer: PAIRS k-pairs of B operands (registers) against the staged A operands, MT output tiles of 32 rows
template <int PAIRS, int MT>
__device__ __forceinline__ void layer(const float *s_w, const float (&b)[PAIRS], float16_t (&acc)[MT], uint32_t lane) {
    #pragma unroll
    for (int p = 0; p < PAIRS; p++) {
        float a[MT];
        const float *src = s_w + ((size_t)p * 64 + lane) * MT;
        if constexpr (MT == 4) *reinterpret_cast<float4 *>(a) = *reinterpret_cast<const float4 *>(src);
        else if constexpr (MT == 2) *reinterpret_cast<float2 *>(a) = *reinterpret_cast<const float2 *>(src);
        else a[0] = src[0];
        #pragma unroll
        for (int mt = 0; mt < MT; mt++) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], b[p], acc[mt], 0, 0, 0);
    }
}

template <int MT>
__device__ __forceinline__ void relu_into(const float16_t (&acc)[MT], float (&b)[16 * MT]) {
    #pragma unroll
    for (int mt = 0; mt < MT; mt++)
        #pragma unroll
        for (int v = 0; v < 16; v++) b[mt * 16 + v] = relu1(acc[mt][v]);
}

// CELLS: the density-grid query of update_extra_state (dnerf/renderer.py:453-555) without -O: slot p is a Morton cell index, the point is
// the cell's jittered centre (cell_points.h), the kernel stops behind the sigma network and writes sigma * density_scale only.
template <bool CELLS>
__global__ void __launch_bounds__(64 * kWaves, 8 / kWaves) k_field_f32(F32Args P, LevelParams lp) {
    __shared__ __attribute__((aligned(16))) float s_w[kStageFloats];
    __shared__ float s_bias[kMaxFrames * 128];      // the frames' time-encoding bias rows (D0's initial accumulators)
    Point pt;
    if (!load_point<CELLS>(P, pt)) return;                           // workgroup-uniform, before any barrier
    const uint32_t lane = pt.lane, h = pt.h, n = pt.n, slot = pt.slot, fr = pt.fr;
    const bool valid = pt.valid, canonical = pt.canonical;
    float x[3] = {pt.x[0], pt.x[1], pt.x[2]}, d[3] = {pt.d[0], pt.d[1], pt.d[2]};
    Pre pre = stage_prefetch<kD0Floats>(P.weights + kD0);            // (see field_f32_common.h: the next stage travels under the layer)
    for (uint32_t k = threadIdx.x; k < P.n_frames * 128u; k += 64 * kWaves) s_bias[k] = P.bias0[k];

    // ---- deformation network: freq(x, 10) (time part folded into bias0) -> 128 x 7 -> 3 ----
    float bin[64];
    {
        const float phase = (float)h * (3.141592653589793f / 2);      // freqencoder.cu: the cosine is the sine shifted by pi/2 in fp32
        #pragma unroll
        for (int p = 0; p < 30; p++) bin[p] = sinf(scalbnf(x[p % 3], p / 3) + phase);
        bin[30] = h ? x[1] : x[0];
        bin[31] = h ? 0.0f : x[2];
    }
    float16_t acc[4];
    stage_commit<kD0Floats>(s_w, pre);
    pre = stage_prefetch<kStageFloats>(P.weights + kD1);
    #pragma unroll
    for (int mt = 0; mt < 4; mt++)
        #pragma unroll
        for (int v = 0; v < 16; v++) acc[mt][v] = s_bias[fr * 128u + mt * 32 + (v >> 2) * 8 + h * 4 + (v & 3)];
    {
        float b0[32];
        #pragma unroll
        for (int p = 0; p < 32; p++) b0[p] = bin[p];
        layer<32, 4>(s_w, b0, acc, lane);
    }
    #pragma unroll 1
    for (int l = 0; l < 6; l++) {
        relu_into<4>(acc, bin);
        stage_commit<kStageFloats>(s_w, pre);                                                     // D(l+1), fetched under the previous layer
        pre = stage_prefetch<kStageFloats>(P.weights + kD1 + (size_t)(l + 1) * kStageFloats);      // D(l+2); after D6 the tail stage (kTail follows D6)
        #pragma unroll
        for (int mt = 0; mt < 4; mt++)
            #pragma unroll
            for (int v = 0; v < 16; v++) acc[mt][v] = 0.0f;
        layer<64, 4>(s_w, bin, acc, lane);
    }
    relu_into<4>(acc, bin);
    stage_commit<kStageFloats>(s_w, pre);
    float16_t a1[1];
    #pragma unroll
    for (int v = 0; v < 16; v++) a1[0][v] = 0.0f;
    layer<64, 1>(s_w + kT_D7, bin, a1, lane);
    // rows 0..2 of the output live in registers 0..2 of the lower lane half; the upper half evaluates the same point
    if (P.deform && valid && h == 0) {      // dnerf/network.py:139-141: `deform = zeros` on the canonical frame
        #pragma unroll
        for (int k = 0; k < 3; k++) P.deform[(size_t)slot * 3 + k] = canonical ? 0.0f : a1[0][k];
    }
    #pragma unroll
    for (int k = 0; k < 3; k++) {
        const float dk = __shfl(a1[0][k], (int)n, 64);
        if (!canonical) x[k] = x[k] + dk;
    }

    // ---- sigma network: grid(x') -> 64 -> 16 ----
    float gin[16];
    {
        float in[3];
        bool oob = false;
        #pragma unroll
        for (int k = 0; k < 3; k++) {
            in[k] = (x[k] + P.bound) / (2 * P.bound);            // grid.py:149
            if (in[k] < 0 || in[k] > 1) oob = true;
        }
        #pragma unroll
        for (int level = 0; level < 16; level++) {
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
            float r = 0;
            if (!oob) {
                float vals[8], ws[8];
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
                    vals[idx] = grid[sdn_grid::grid_index<3, 2>(1u, false, hashmap_size, resolution, pgl) + h];    // tiled grid, channel h
                }
                #pragma unroll
                for (uint32_t idx = 0; idx < 8; idx++) r = r + ws[idx] * vals[idx];
            }
            gin[level] = r;
        }
    }
    float16_t a2[2];
    #pragma unroll
    for (int mt = 0; mt < 2; mt++)
        #pragma unroll
        for (int v = 0; v < 16; v++) a2[mt][v] = 0.0f;
    layer<16, 2>(s_w + kT_S0, gin, a2, lane);
    float b32[32];
    relu_into<2>(a2, b32);
    #pragma unroll
    for (int v = 0; v < 16; v++) a1[0][v] = 0.0f;
    layer<32, 1>(s_w + kT_S1, b32, a1, lane);
    const float sigma = expf(a1[0][0]) * P.density_scale;     // row 0 (lower half); trunc_exp's forward is exp
    if constexpr (CELLS) {                                    // (every barrier of the workgroup lies behind this wave)
        if (valid && h == 0) P.sigmas[slot] = sigma;
        return;
    }

    // ---- colour network: SH(d, 4) ++ geo_feat (rows 1..15, raw) -> 64 -> 64 -> 3 ----
    float cin[16];
    {
        float sh[16], *nul = nullptr;
        sdn_sh::sh_eval<4, false>(d[0], d[1], d[2], sh, nul, nul, nul);
        #pragma unroll
        for (int p = 0; p < 8; p++) cin[p] = h ? sh[2 * p + 1] : sh[2 * p];
        #pragma unroll
        for (int v = 0; v < 8; v++) cin[8 + v] = a1[0][v];       // pairs of rows (8 (v / 4) + v % 4, + 4); row 0's weights are zero
    }
    #pragma unroll
    for (int mt = 0; mt < 2; mt++)
        #pragma unroll
        for (int v = 0; v < 16; v++) a2[mt][v] = 0.0f;
    layer<16, 2>(s_w + kT_C0, cin, a2, lane);
    relu_into<2>(a2, b32);
    #pragma unroll
    for (int mt = 0; mt < 2; mt++)
        #pragma unroll
        for (int v = 0; v < 16; v++) a2[mt][v] = 0.0f;
    layer<32, 2>(s_w + kT_C1, b32, a2, lane);
    relu_into<2>(a2, b32);
    #pragma unroll
    for (int v = 0; v < 16; v++) a1[0][v] = 0.0f;
    layer<32, 1>(s_w + kT_C2, b32, a1, lane);
    if (valid && h == 0) {
        P.sigmas[slot] = sigma;
        #pragma unroll
        for (int k = 0; k < 3; k++) P.rgbs[(size_t)slot * 3 + k] = 1.0f / (1.0f + expf(-a1[0][k]));
    }
}

}  // namespace

namespace sdn_int {
int field_forward_f32(const float *xyzs, const float *dirs, const uint32_t *live_idx, const uint32_t *live_count, const int32_t *state,
       uint32_t M, const float *weights, const float *bias0, const float *table, const int32_t *offsets_host, float S,
       uint32_t H, float bound, float density_scale, int zero_deform, float *sigmas, float *rgbs, float *deform,
       const uint8_t *slot_frame, uint32_t n_frames, hipStream_t st) {
    sdn_f32::LevelParams lp;
    sdn_f32::F32Args a;
    int rc = sdn_f32::fill_args(a, lp, xyzs, dirs, live_idx, live_count, state, M, weights, bias0, table, offsets_host, S, H, bound, density_scale,
                                zero_deform, sigmas, rgbs, deform, slot_frame, n_frames);
    if (rc) return rc;
    hipLaunchKernelGGL(k_field_f32<false>, dim3(sdn_div_up(M, (uint32_t)sdn_f32::kPointsPerWG)), dim3(64 * sdn_f32::kWaves), 0, st, a, lp);
    return sdn_launch_status();
}

// sigma * density_scale of jittered occupancy-grid cell centres -> tmp_grid slice, fp32 network (the fp32 twin of field_cells_f16)
int field_cells_f32(const int32_t *cells, const uint32_t *cell_count, uint32_t n, const float *noise, uint32_t seed, uint32_t grid_size,
                    float cas_bound, const float *weights, const float *bias0, const float *table, const int32_t *offsets_host, float S,
                    uint32_t H, float bound, float density_scale, int zero_deform, float *tmp_slice, hipStream_t st) {
    sdn_f32::LevelParams lp;
    sdn_f32::F32Args a;
    int rc = sdn_f32::fill_args(a, lp, nullptr, nullptr, (const uint32_t *)cells, cell_count, nullptr, n, weights, bias0, table, offsets_host, S, H,
                                bound, density_scale, zero_deform ? 1 : 0, tmp_slice, nullptr, nullptr, nullptr, 1u);
    if (rc) return rc;
    a.cell_noise = noise; a.cell_seed = seed;
    const float half_grid = cas_bound / (float)grid_size;
    a.cell_inv = 1.0f / (float)(grid_size - 1); a.cell_span = cas_bound - half_grid; a.cell_half = half_grid;
    hipLaunchKernelGGL(k_field_f32<true>, dim3(sdn_div_up(n, (uint32_t)sdn_f32::kPointsPerWG)), dim3(64 * sdn_f32::kWaves), 0, st, a, lp);
    return sdn_launch_status();
}
}  // namespace sdn_int

extern "C" {

uint32_t sdn_field_weight_floats_f32(void) { return (uint32_t)sdn_f32::kTotalFloats; }

int sdn_field_forward_f32(const float *xyzs, const float *dirs, const uint32_t *live_idx, const uint32_t *live_count, uint32_t M,
                          const float *weights, const float *bias0, const float *table, const int32_t *offsets_host, float S, uint32_t H,
                          float bound, float density_scale, int zero_deform, float *sigmas, float *rgbs, float *deform, void *stream) {
    if (M == 0) return 0;
    if (!xyzs || !dirs || !weights || !bias0 || !table || !offsets_host || !sigmas || !rgbs) return SDN_E_BADARG;
    if ((live_idx == nullptr) != (live_count == nullptr)) return SDN_E_BADARG;
    if (((uintptr_t)weights & 15u) != 0 || ((uintptr_t)table & 3u) != 0) return SDN_E_BADARG;
    return sdn_int::field_forward_f32(xyzs, dirs, live_idx, live_count, nullptr, M, weights, bias0, table, offsets_host, S, H, bound,
                                      density_scale, zero_deform ? 1 : 0, sigmas, rgbs, deform, nullptr, 1u, (hipStream_t)stream);
}

// The density-grid query of update_extra_state for a model trained WITHOUT -O: as sdn_density_query_cells_f16 with the fp32 network
// (weights of sdn_field_weight_floats_f32 floats, the model's fp32 embedding table in the reference layout).
int sdn_density_query_cells_f32(const int32_t *cells, const uint32_t *cell_count, uint32_t n, const float *noise, uint32_t seed,
                                uint32_t grid_size, float cas_bound, const float *weights, const float *bias0, const float *table,
                                const int32_t *offsets_host, float S, uint32_t H, float bound, float density_scale, int zero_deform,
                                float *tmp_slice, void *stream) {
    if (n == 0) return 0;
    if (!weights || !bias0 || !table || !offsets_host || !tmp_slice) return SDN_E_BADARG;
    if ((cells == nullptr) != (cell_count == nullptr)) return SDN_E_BADARG;
    if (grid_size < 2 || grid_size > 1024 || !(cas_bound > 0)) return SDN_E_BADARG;
    if (!cells && (uint64_t)n > (uint64_t)grid_size * grid_size * grid_size) return SDN_E_BADARG;   // without a list, slot p IS the Morton index
    if (((uintptr_t)weights & 15u) != 0 || ((uintptr_t)table & 3u) != 0) return SDN_E_BADARG;
    return sdn_int::field_cells_f32(cells, cell_count, n, noise, seed, grid_size, cas_bound, weights, bias0, table, offsets_host, S, H, bound,
                                    density_scale, zero_deform, tmp_slice, (hipStream_t)stream);
}

}  // extern "C"
