// Multiresolution hash / tiled grid encoder for gfx950 (MI355X).
//
// Behavioural contract: gridencoder/src/gridencoder.cu of the reference
// (kernel_grid :87-245, kernel_grid_backward :248-340, kernel_input_backward :343-369).
// Built with -ffp-contract=off: position / index / weight arithmetic rounds exactly
// as written in the reference source, so grid indices are bit-exact against
// oracle/sdn_oracle.c and fp32 outputs are bit-identical (same corner order);
// the fp16 path rounds to half wherever the reference's at::Half operators do.
//
// MI355X mapping:
//   * launch = (point tiles) x (levels), level is the SLOW grid dimension: the
//     dispatcher walks x first, so at any instant the whole chip works on one or two
//     levels and the live table slice (<= 2 MiB fp16 / 4 MiB fp32 per level) stays in
//     every XCD's 4 MiB L2; the whole 23/47 MiB table stays in the 256 MiB Infinity
//     Cache between frames.
//   * per-level constants (row offset, row count, scale, resolution) are computed
//     on the host and passed by value -> SGPRs, no dependent offsets[] loads.
//   * 256-thread blocks; one lane = one (point, level); inputs are re-read per
//     level from L2 (12 B/point), outputs are written [L,B,C] so a wave's stores
//     are contiguous.
//   * backward: no-return float / packed-half atomics (global_atomic_add_f32 /
//     global_atomic_pk_add_f16), order-free like the reference.
#include <math.h>

#include "sdn_common.h"
#include "grid_common.h"

namespace {

using namespace sdn_grid;

template <typename T> struct Num;
template <> struct Num<float> {
    static __device__ __forceinline__ float ld(const float *p) { return *p; }
    static __device__ __forceinline__ float rnd(float v) { return v; }
    static __device__ __forceinline__ void st(float *p, float v) { *p = v; }
};
template <> struct Num<__half> {
    static __device__ __forceinline__ float ld(const __half *p) { return __half2float(*p); }
    // (the empty asm pins v as an fp32 VALUE: without it the compiler folds `half(a * b)` / `half(a + b)` into v_fma_mixlo_f16, which
    // rounds the exact result ONCE to fp16 -- the reference rounds the float operation to fp32 first and converts that)
    static __device__ __forceinline__ float rnd(float v) { asm volatile("" : "+v"(v)); return __half2float(__float2half_rn(v)); }
    static __device__ __forceinline__ void st(__half *p, float v) { *p = __float2half_rn(v); }
};

// ---------------------------------------------------------------------------
// forward (+ optional dy_dx)            gridencoder.cu:87-245
// ---------------------------------------------------------------------------
template <typename T, uint32_t D, uint32_t C, bool WITH_DYDX>
__global__ void __launch_bounds__(256) k_grid_fwd(const float *__restrict__ inputs, const T *__restrict__ grid_all, T *__restrict__ outputs,
                                                  uint32_t B, uint32_t L, LevelParams lp, T *__restrict__ dy_dx, uint32_t gridtype,
                                                  bool align_corners, uint32_t interp) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    const T *__restrict__ grid = grid_all + (size_t)lp.offset[level] * C;
    const uint32_t hashmap_size = lp.hashmap_size[level];
    const float scale = lp.scale[level];
    const uint32_t resolution = lp.resolution[level];
    T *out = outputs + ((size_t)level * B + b) * C;

    float in[D];
    bool oob = false;
    #pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        in[d] = inputs[(size_t)b * D + d];
        if (in[d] < 0 || in[d] > 1) oob = true;
    }
    if (oob) {
        #pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) Num<T>::st(out + ch, 0.0f);
        if (WITH_DYDX) {
            T *dd = dy_dx + (size_t)b * D * L * C + (size_t)level * D * C;
            #pragma unroll
            for (uint32_t i = 0; i < D * C; i++) Num<T>::st(dd + i, 0.0f);
        }
        return;
    }

    float pos[D], pos_deriv[D];
    uint32_t pos_grid[D];
    #pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        pos[d] = in[d] * scale + (align_corners ? 0.0f : 0.5f);
        pos_grid[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pos_grid[d];
        if (interp == 1) {
            pos_deriv[d] = smoothstep_derivative_(pos[d]);
            pos[d] = smoothstep_(pos[d]);
        } else {
            pos_deriv[d] = 1.0f;
        }
    }

    // issue all row gathers first (independent loads), then blend in the reference's corner order
    float vals[1u << D][C];
    float ws[1u << D];
    #pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        float w = 1;
        #pragma unroll
        for (uint32_t d = 0; d < D; d++) w *= (idx & (1u << d)) ? pos[d] : 1 - pos[d];
        ws[idx] = w;
    }
    // A level whose index is linear in the coordinates (every tiled level, every hash level that fits its table) keeps the x and
    // x+1 corners in neighbouring rows: ONE gather of 2 C elements fetches both (the gather address rate, not bytes, bounds this
    // kernel).  Workgroup-uniform choice; the rare wrap (x corner in the last row of a capped level) is patched after all gathers
    // of the lane have been issued.
    uint32_t full_stride = 1;
    #pragma unroll
    for (uint32_t d = 0; d < D; d++)
        if (full_stride <= hashmap_size) full_stride *= align_corners ? resolution : (resolution + 1);
    // (a row must be a whole number of dwords: the 2-row gather is then at least dword aligned)
    const bool linear = (C * sizeof(T)) % 4 == 0 && !(gridtype == 0 && full_stride > hashmap_size) && hashmap_size >= 2;
    if (linear) {
        constexpr uint32_t NP = 1u << (D - 1);
        T raw[NP][2 * C];
        uint32_t wrapped = 0;
        #pragma unroll
        for (uint32_t pi = 0; pi < NP; pi++) {
            uint32_t pgl[D];
            pgl[0] = pos_grid[0];
            #pragma unroll
            for (uint32_t d = 1; d < D; d++) pgl[d] = pos_grid[d] + ((pi >> (d - 1)) & 1u);
            const uint32_t row0 = grid_index<D, 1>(gridtype, align_corners, hashmap_size, resolution, pgl);
            const uint32_t rl = min(row0, hashmap_size - 2u);
            __builtin_memcpy(raw[pi], grid + (size_t)rl * C, 2 * C * sizeof(T));
            if (row0 > rl) wrapped |= 1u << pi;
        }
        if (__builtin_expect(wrapped != 0u, 0)) {
            #pragma unroll
            for (uint32_t pi = 0; pi < NP; pi++) {
                if ((wrapped >> pi) & 1u) {   // x corner = last row (second half of what was fetched), x+1 corner = row 0
                    #pragma unroll
                    for (uint32_t ch = 0; ch < C; ch++) { raw[pi][ch] = raw[pi][C + ch]; raw[pi][C + ch] = grid[ch]; }
                }
            }
        }
        #pragma unroll
        for (uint32_t idx = 0; idx < (1u << D); idx++) {
            #pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) vals[idx][ch] = Num<T>::ld(&raw[idx >> 1][(idx & 1u) * C + ch]);
        }
    } else {
        #pragma unroll
        for (uint32_t idx = 0; idx < (1u << D); idx++) {
            uint32_t pgl[D];
            #pragma unroll
            for (uint32_t d = 0; d < D; d++) pgl[d] = pos_grid[d] + ((idx >> d) & 1u);
            const uint32_t index = grid_index<D, C>(gridtype, align_corners, hashmap_size, resolution, pgl);
            #pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) vals[idx][ch] = Num<T>::ld(grid + index + ch);
        }
    }
    float results[C];
    #pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) results[ch] = 0;
    #pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        #pragma unroll
        // results[ch] += w * grid[...] with at::Half results: the float product is converted to Half first (the only `Half += x` takes a
        // Half), then Half + Half rounds once more -- two half roundings per corner (gridencoder.cu:187-189, c10 Half.h)
        for (uint32_t ch = 0; ch < C; ch++) results[ch] = Num<T>::rnd(results[ch] + Num<T>::rnd(ws[idx] * vals[idx][ch]));
    }
    #pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) Num<T>::st(out + ch, results[ch]);

    if (WITH_DYDX) {
        T *dd = dy_dx + (size_t)b * D * L * C + (size_t)level * D * C;
        #pragma unroll
        for (uint32_t gd = 0; gd < D; gd++) {
            float rg[C];
            #pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) rg[ch] = 0;
            #pragma unroll
            for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                float w = scale;
                uint32_t corner = 0;  // bit d set <=> +1 along d, gd excluded
                #pragma unroll
                for (uint32_t nd = 0; nd < D - 1; nd++) {
                    const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                    if ((idx & (1u << nd)) == 0) { w *= 1 - pos[d]; }
                    else { w *= pos[d]; corner |= (1u << d); }
                }
                // the left/right rows are two of the 2^D corners gathered above: reuse the registers
                #pragma unroll
                for (uint32_t ch = 0; ch < C; ch++) {
                    const float diff = Num<T>::rnd(vals[corner | (1u << gd)][ch] - vals[corner][ch]);
                    rg[ch] = Num<T>::rnd(rg[ch] + Num<T>::rnd(w * diff * pos_deriv[gd]));
                }
            }
            #pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) Num<T>::st(dd + gd * C + ch, rg[ch]);
        }
    }
}

// ---------------------------------------------------------------------------
// backward: scatter-add into the table gradient     gridencoder.cu:248-340
// one lane = (point, level, channel pair)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void atomic_add_f32(float *p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add_h2(__half *p, float a, float b) {
    unsafeAtomicAdd(reinterpret_cast<__half2 *>(p), __halves2half2(__float2half_rn(a), __float2half_rn(b)));
}
// scalar half add without a native instruction: CAS on the enclosing 32-bit word (C == 1 only; the
// reference's own comment calls this path "very slow ... never use it").
__device__ __forceinline__ void atomic_add_h1(__half *p, float a) {
    const uintptr_t addr = (uintptr_t)p;
    unsigned int *word = (unsigned int *)(addr & ~(uintptr_t)3);
    const bool hi = addr & 2;
    unsigned int old = *word, assumed;
    const __half add = __float2half_rn(a);
    do {
        assumed = old;
        const unsigned short cur = hi ? (unsigned short)(assumed >> 16) : (unsigned short)(assumed & 0xFFFFu);
        const __half nv = __hadd(__ushort_as_half(cur), add);
        const unsigned int nb = __half_as_ushort(nv);
        const unsigned int repl = hi ? ((assumed & 0x0000FFFFu) | (nb << 16)) : ((assumed & 0xFFFF0000u) | nb);
        old = atomicCAS(word, assumed, repl);
    } while (old != assumed);
}

// DET (deterministic mode, sdn_grid_encode_backward_det): every contribution w * grad is added to a 64-bit FIXED-POINT accumulator
// (det_all, one per table element, 2^-kDetFrac units) with an integer atomic -- integer addition is associative, so the sum does not
// depend on the order in which the hardware executes the atomics, which the float / half atomics' rounding does.  k_grid_det_finish adds
// the sums to the gradient table, one rounding per element.
template <typename T> struct DetScale;
template <> struct DetScale<__half> { static constexpr double value = 16777216.0; };          // 2^24: fp16's subnormal step is 2^-24; +-5e11 of range
template <> struct DetScale<float> { static constexpr double value = 1099511627776.0; };       // 2^40: 9e-13 of resolution, +-8e6 of range

template <typename T, uint32_t D, uint32_t C, uint32_t N_C, bool DET>
__global__ void __launch_bounds__(256) k_grid_bwd(const T *__restrict__ grad, const float *__restrict__ inputs, T *__restrict__ grad_grid_all,
                                                  uint32_t B, uint32_t L, LevelParams lp, uint32_t gridtype, bool align_corners,
                                                  uint32_t interp, long long *__restrict__ det_all) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = tid * N_C / C;
    const uint32_t level = blockIdx.y;
    const uint32_t ch = tid * N_C - b * C;
    T *__restrict__ grad_grid = grad_grid_all + (size_t)lp.offset[level] * C;
    const uint32_t hashmap_size = lp.hashmap_size[level];
    const float scale = lp.scale[level];
    const uint32_t resolution = lp.resolution[level];
    const uint32_t lane = threadIdx.x & 63u;

    // No early return: every lane of the wave takes part in the run aggregation below (an idle lane contributes nothing).
    bool active = b < B;
    float pos[D];
    uint32_t pos_grid[D];
    #pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        const float v = active ? inputs[(size_t)b * D + d] : 0.0f;
        if (v < 0 || v > 1) active = false;
        pos[d] = v * scale + (align_corners ? 0.0f : 0.5f);
        pos_grid[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pos_grid[d];
        if (interp == 1) pos[d] = smoothstep_(pos[d]);
    }
    float grad_cur[N_C];
    #pragma unroll
    for (uint32_t c = 0; c < N_C; c++) grad_cur[c] = active ? Num<T>::ld(grad + ((size_t)level * B + b) * C + ch + c) : 0.0f;

    #pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        float w = 1;
        uint32_t pgl[D];
        #pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
            else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
        }
        const uint32_t index = grid_index<D, C>(gridtype, align_corners, hashmap_size, resolution, pgl) + ch;
        if constexpr (DET) {
            if (active) {
                long long *det = det_all + (size_t)lp.offset[level] * C + index;
                #pragma unroll
                for (uint32_t c = 0; c < N_C; c++)
                    atomicAdd((unsigned long long *)(det + c), (unsigned long long)__double2ll_rn((double)(w * grad_cur[c]) * DetScale<T>::value));
            }
            continue;
        }
        // Consecutive lanes are consecutive samples -- in training, neighbours along a ray, a fraction of a coarse cell apart -- so on
        // the coarse levels whole runs of lanes update the SAME row, and a wave's same-address atomics are executed one after the
        // other by the L2 (the 9 000-sample training batch spent 176 us here, nearly all of it in the few hundred hot rows of levels
        // 0-3).  Each run of equal rows is summed across its lanes in fp32 (segmented inclusive scan over run numbers) and its last
        // lane issues ONE atomic.  Runs of one lane -- every fine level, any unordered batch -- behave exactly as before.
        const uint32_t key = active ? index : 0xFFFFFFFFu - lane;            // idle lanes: a private key each
        const uint32_t prev_key = __shfl_up(key, 1, 64);
        const unsigned long long heads = __ballot(lane == 0 || prev_key != key);
        const uint32_t run = (uint32_t)__popcll(heads & (~0ull >> (63u - lane)));   // run number, non-decreasing with the lane
        float acc[N_C];
        #pragma unroll
        for (uint32_t c = 0; c < N_C; c++) acc[c] = w * grad_cur[c];
        if (__popcll(heads) != 64) {   // wave-uniform: something to merge
            #pragma unroll
            for (uint32_t off = 1; off < 64; off <<= 1) {
                const uint32_t r = __shfl_up(run, off, 64);
                #pragma unroll
                for (uint32_t c = 0; c < N_C; c++) {
                    const float o = __shfl_up(acc[c], off, 64);
                    if (lane >= off && r == run) acc[c] += o;
                }
            }
        }
        const bool tail = lane == 63u || ((heads >> (lane + 1u)) & 1ull);
        if (!active || !tail) continue;
        if constexpr (sizeof(T) == 2) {
            if constexpr (N_C == 2) atomic_add_h2((__half *)grad_grid + index, acc[0], acc[1]);
            else atomic_add_h1((__half *)grad_grid + index, acc[0]);
        } else {
            #pragma unroll
            for (uint32_t c = 0; c < N_C; c++) atomic_add_f32((float *)grad_grid + index + c, acc[c]);
        }
    }
}

// deterministic mode, second pass: gradient element += its fixed-point sum (one rounding)
template <typename T>
__global__ void __launch_bounds__(256) k_grid_det_finish(T *__restrict__ grad_grid, const long long *__restrict__ det, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long v = det[i];
    if (v == 0) return;
    Num<T>::st(grad_grid + i, (float)((double)Num<T>::ld(grad_grid + i) + (double)v / DetScale<T>::value));
}

// gridencoder.cu:343-369    grad_inputs[b,d] = sum_{l,c} grad[l,b,c] * dy_dx[b,l,d,c]
template <typename T, uint32_t D, uint32_t C>
__global__ void __launch_bounds__(256) k_grid_input_bwd(const T *__restrict__ grad, const T *__restrict__ dy_dx, T *__restrict__ grad_inputs,
                                                        uint32_t B, uint32_t L) {
    const uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const T *dd = dy_dx + (size_t)b * L * D * C;
    float result = 0;
    for (uint32_t l = 0; l < L; l++) {
        #pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) {
            const float g = Num<T>::ld(grad + ((size_t)l * B + b) * C + ch);
            const float x = Num<T>::ld(dd + (size_t)l * D * C + d * C + ch);
            // at::Half: (Half*Half -> Half) then (Half += Half); float: plain fma-free mul, add
            result = Num<T>::rnd(result + Num<T>::rnd(g * x));
        }
    }
    Num<T>::st(grad_inputs + t, result);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
template <typename T, uint32_t D, uint32_t C>
void launch_fwd(const float *inputs, const T *emb, T *out, uint32_t B, uint32_t L, const LevelParams &lp, T *dy_dx, uint32_t gridtype,
                bool ac, uint32_t interp, hipStream_t st) {
    const dim3 grid(sdn_div_up(B, 256u), L, 1);
    if (dy_dx) hipLaunchKernelGGL((k_grid_fwd<T, D, C, true>), grid, dim3(256), 0, st, inputs, emb, out, B, L, lp, dy_dx, gridtype, ac, interp);
    else hipLaunchKernelGGL((k_grid_fwd<T, D, C, false>), grid, dim3(256), 0, st, inputs, emb, out, B, L, lp, dy_dx, gridtype, ac, interp);
}

template <typename T, uint32_t D>
int dispatch_fwd_c(uint32_t C, const float *inputs, const T *emb, T *out, uint32_t B, uint32_t L, const LevelParams &lp, T *dy_dx,
                   uint32_t gridtype, bool ac, uint32_t interp, hipStream_t st) {
    switch (C) {
        case 1: launch_fwd<T, D, 1>(inputs, emb, out, B, L, lp, dy_dx, gridtype, ac, interp, st); break;
        case 2: launch_fwd<T, D, 2>(inputs, emb, out, B, L, lp, dy_dx, gridtype, ac, interp, st); break;
        case 4: launch_fwd<T, D, 4>(inputs, emb, out, B, L, lp, dy_dx, gridtype, ac, interp, st); break;
        case 8: launch_fwd<T, D, 8>(inputs, emb, out, B, L, lp, dy_dx, gridtype, ac, interp, st); break;
        default: return SDN_E_UNSUPPORTED;  // "GridEncoding: C must be 1, 2, 4, or 8." gridencoder.cu:381
    }
    return 0;
}

template <typename T>
int dispatch_fwd(uint32_t D, uint32_t C, const float *inputs, const T *emb, T *out, uint32_t B, uint32_t L, const LevelParams &lp, T *dy_dx,
                 uint32_t gridtype, bool ac, uint32_t interp, hipStream_t st) {
    switch (D) {
        case 2: return dispatch_fwd_c<T, 2>(C, inputs, emb, out, B, L, lp, dy_dx, gridtype, ac, interp, st);
        case 3: return dispatch_fwd_c<T, 3>(C, inputs, emb, out, B, L, lp, dy_dx, gridtype, ac, interp, st);
        case 4: return dispatch_fwd_c<T, 4>(C, inputs, emb, out, B, L, lp, dy_dx, gridtype, ac, interp, st);
        case 5: return dispatch_fwd_c<T, 5>(C, inputs, emb, out, B, L, lp, dy_dx, gridtype, ac, interp, st);
        default: return SDN_E_UNSUPPORTED;
    }
}

template <typename T, uint32_t D, uint32_t C, uint32_t N_C>
void launch_bwd(const T *grad, const float *inputs, T *gg, uint32_t B, uint32_t L, const LevelParams &lp, const T *dy_dx, T *gi,
                uint32_t gridtype, bool ac, uint32_t interp, hipStream_t st, long long *det) {
    const dim3 grid(sdn_div_up(B * C / N_C, 256u), L, 1);
    if (det) {
        const uint64_t n = ((uint64_t)lp.offset[L - 1] + lp.hashmap_size[L - 1]) * C;      // elements of the whole table
        if (hipMemsetAsync(det, 0, n * sizeof(long long), st) != hipSuccess) return;      // (reported by the caller's sdn_launch_status)
        hipLaunchKernelGGL((k_grid_bwd<T, D, C, N_C, true>), grid, dim3(256), 0, st, grad, inputs, gg, B, L, lp, gridtype, ac, interp, det);
        hipLaunchKernelGGL((k_grid_det_finish<T>), dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, gg, (const long long *)det, n);
    } else {
        hipLaunchKernelGGL((k_grid_bwd<T, D, C, N_C, false>), grid, dim3(256), 0, st, grad, inputs, gg, B, L, lp, gridtype, ac, interp, (long long *)nullptr);
    }
    if (dy_dx && gi) hipLaunchKernelGGL((k_grid_input_bwd<T, D, C>), dim3(sdn_div_up(B * D, 256u)), dim3(256), 0, st, grad, dy_dx, gi, B, L);
}

template <typename T, uint32_t D>
int dispatch_bwd_c(uint32_t C, const T *grad, const float *inputs, T *gg, uint32_t B, uint32_t L, const LevelParams &lp, const T *dy_dx, T *gi,
                   uint32_t gridtype, bool ac, uint32_t interp, hipStream_t st, long long *det) {
    switch (C) {
        case 1: launch_bwd<T, D, 1, 1>(grad, inputs, gg, B, L, lp, dy_dx, gi, gridtype, ac, interp, st, det); break;
        case 2: launch_bwd<T, D, 2, 2>(grad, inputs, gg, B, L, lp, dy_dx, gi, gridtype, ac, interp, st, det); break;
        case 4: launch_bwd<T, D, 4, 2>(grad, inputs, gg, B, L, lp, dy_dx, gi, gridtype, ac, interp, st, det); break;
        case 8: launch_bwd<T, D, 8, 2>(grad, inputs, gg, B, L, lp, dy_dx, gi, gridtype, ac, interp, st, det); break;
        default: return SDN_E_UNSUPPORTED;
    }
    return 0;
}

template <typename T>
int dispatch_bwd(uint32_t D, uint32_t C, const T *grad, const float *inputs, T *gg, uint32_t B, uint32_t L, const LevelParams &lp,
                 const T *dy_dx, T *gi, uint32_t gridtype, bool ac, uint32_t interp, hipStream_t st, long long *det) {
    switch (D) {
        case 2: return dispatch_bwd_c<T, 2>(C, grad, inputs, gg, B, L, lp, dy_dx, gi, gridtype, ac, interp, st, det);
        case 3: return dispatch_bwd_c<T, 3>(C, grad, inputs, gg, B, L, lp, dy_dx, gi, gridtype, ac, interp, st, det);
        case 4: return dispatch_bwd_c<T, 4>(C, grad, inputs, gg, B, L, lp, dy_dx, gi, gridtype, ac, interp, st, det);
        case 5: return dispatch_bwd_c<T, 5>(C, grad, inputs, gg, B, L, lp, dy_dx, gi, gridtype, ac, interp, st, det);
        default: return SDN_E_UNSUPPORTED;
    }
}

// ---------------------------------------------------------------------------
// forward on a QUAD copy of an fp16 table (sdn_field_build_quad_table: block r of a level = rows {r, r+1, r+s1, r+s1+1} mod the level's
// row count -- the four (x, y) corners of a cell): TWO 16-byte gathers per (point, level) instead of four 8-byte ones.  The operator is
// bound by the rate at which a CU takes lane-divergent gather addresses (DESIGN.md "The address-rate roof"), so halving the
// addresses is what counts.  Same values into the same arithmetic as k_grid_fwd<__half, 3, 2> (tiled grid, no align_corners, linear
// interpolation): bit-identical outputs.  For inference on a table that does not change between calls (gridencoder/grid.py keeps the
// copy per table version).
// ---------------------------------------------------------------------------
struct QuadFwdLevels {
    uint32_t offset_q[16], hsize[16], s1[16], s2[16];
    float scale[16];
};
// LV levels per lane (blockIdx.y = level group): the inputs are read once per group and the 2 LV gathers of a lane are in flight together
// -- the launch is short of waves, not of addresses, when every lane waits for two loads (one level per lane: 25 us per 196 352 points,
// like the plain kernel; profiles/r04_grid_quad_forward.txt)
template <int LV>
__global__ void __launch_bounds__(256) k_grid_fwd_quad(const float *__restrict__ inputs, const uint4 *__restrict__ quad, __half *__restrict__ outputs,
                                                       uint32_t B, QuadFwdLevels q) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t level0 = blockIdx.y * LV;
    float in[3];
    bool oob = false;
    #pragma unroll
    for (uint32_t d = 0; d < 3; d++) {
        in[d] = inputs[(size_t)b * 3 + d];
        if (in[d] < 0 || in[d] > 1) oob = true;
    }
    if (oob) {
        #pragma unroll
        for (int k = 0; k < LV; k++) {
            __half *out = outputs + ((size_t)(level0 + k) * B + b) * 2;
            Num<__half>::st(out, 0.0f); Num<__half>::st(out + 1, 0.0f);
        }
        return;
    }
    float pos[LV][3];
    uint4 lo[LV], hi[LV];
    #pragma unroll
    for (int k = 0; k < LV; k++) {
        const uint32_t level = level0 + k;
        const uint32_t hs = q.hsize[level], s1 = q.s1[level], s2 = q.s2[level];
        const float scale = q.scale[level];
        uint32_t pg[3];
        #pragma unroll
        for (uint32_t d = 0; d < 3; d++) {
            pos[k][d] = in[d] * scale + 0.5f;
            pg[d] = (uint32_t)floorf(pos[k][d]);
            pos[k][d] -= (float)pg[d];
        }
        // get_grid_index (gridencoder.cu:66-84) of the cell's low corner and of the corner one step up in z; a dropped dimension has stride 0
        const uint32_t lin = pg[0] + pg[1] * s1 + pg[2] * s2;
        auto reduce = [&](uint32_t i) { return ((hs & (hs - 1u)) == 0u) ? (i & (hs - 1u)) : (i >= hs ? i % hs : i); };
        lo[k] = quad[(size_t)q.offset_q[level] + reduce(lin)];
        hi[k] = quad[(size_t)q.offset_q[level] + reduce(lin + s2)];
    }
    #pragma unroll
    for (int k = 0; k < LV; k++) {
        float ws[8];
        #pragma unroll
        for (uint32_t idx = 0; idx < 8; idx++) {
            float w = 1;
            #pragma unroll
            for (uint32_t d = 0; d < 3; d++) w *= (idx & (1u << d)) ? pos[k][d] : 1 - pos[k][d];
            ws[idx] = w;
        }
        const uint32_t raw[8] = {lo[k].x, lo[k].y, lo[k].z, lo[k].w, hi[k].x, hi[k].y, hi[k].z, hi[k].w};
        float r0 = 0, r1 = 0;
        #pragma unroll
        for (uint32_t idx = 0; idx < 8; idx++) {
            const __half2 v = __builtin_bit_cast(__half2, raw[idx]);
            r0 = Num<__half>::rnd(r0 + Num<__half>::rnd(ws[idx] * __low2float(v)));
            r1 = Num<__half>::rnd(r1 + Num<__half>::rnd(ws[idx] * __high2float(v)));
        }
        __half *out = outputs + ((size_t)(level0 + k) * B + b) * 2;
        Num<__half>::st(out, r0); Num<__half>::st(out + 1, r1);
    }
}

}  // namespace

extern "C" {
// grid_encode forward for D = 3, C = 2, 16 tiled levels, fp16, on the QUAD copy of the table (sdn_field_build_quad_table; ref_offsets_host
// = the reference's 17 level offsets, the copy's level l starts at block ref_offsets_host[l] + 2 l): outputs [16, B, 2] fp16,
// bit-identical to sdn_grid_encode_forward on the fp16 table.
int sdn_grid_encode_forward_quad_f16(const float *inputs, const void *quad_table, const int32_t *ref_offsets_host, void *outputs, uint32_t B,
                                     float S, uint32_t H, void *stream) {
    if (B == 0) return 0;
    if (!inputs || !quad_table || !ref_offsets_host || !outputs || ((uintptr_t)quad_table & 15u) != 0) return SDN_E_BADARG;
    QuadFwdLevels q;
    for (uint32_t l = 0; l < 16; l++) {
        const uint32_t hs = (uint32_t)(ref_offsets_host[l + 1] - ref_offsets_host[l]);
        if (hs == 0) return SDN_E_BADARG;
        const float scale = exp2f((float)l * S) * (float)H - 1.0f;      // gridencoder.cu:138-139
        const uint32_t res = (uint32_t)ceil((double)scale) + 1;
        uint32_t stride = 1, st3[3] = {0, 0, 0};
        for (int d = 0; d < 3; d++)
            if (stride <= hs) { st3[d] = stride; stride *= (res + 1); }
        if (st3[0] != 1) return SDN_E_BADARG;
        q.offset_q[l] = (uint32_t)ref_offsets_host[l] + 2u * l;
        q.hsize[l] = hs; q.s1[l] = st3[1]; q.s2[l] = st3[2]; q.scale[l] = scale;
    }
    static int lv = 0;            // levels per lane: 4 (SDN_GRID_QUAD_LEVELS = 1 | 2 | 4 | 8 | 16 for measurements)
    if (lv == 0) { const char *e = getenv("SDN_GRID_QUAD_LEVELS"); lv = e ? atoi(e) : 4; }
    const dim3 b256(256);
    hipStream_t st = (hipStream_t)stream;
    const uint4 *qt = (const uint4 *)quad_table;
    __half *o = (__half *)outputs;
    switch (lv) {
        case 1: hipLaunchKernelGGL(k_grid_fwd_quad<1>, dim3(sdn_div_up(B, 256u), 16), b256, 0, st, inputs, qt, o, B, q); break;
        case 2: hipLaunchKernelGGL(k_grid_fwd_quad<2>, dim3(sdn_div_up(B, 256u), 8), b256, 0, st, inputs, qt, o, B, q); break;
        case 8: hipLaunchKernelGGL(k_grid_fwd_quad<8>, dim3(sdn_div_up(B, 256u), 2), b256, 0, st, inputs, qt, o, B, q); break;
        case 16: hipLaunchKernelGGL(k_grid_fwd_quad<16>, dim3(sdn_div_up(B, 256u), 1), b256, 0, st, inputs, qt, o, B, q); break;
        default: hipLaunchKernelGGL(k_grid_fwd_quad<4>, dim3(sdn_div_up(B, 256u), 4), b256, 0, st, inputs, qt, o, B, q); break;
    }
    return sdn_launch_status();
}


int sdn_grid_encode_forward(const float *inputs, const void *embeddings, const int32_t *offsets_host, void *outputs, uint32_t B,
                            uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, void *dy_dx, uint32_t gridtype,
                            int align_corners, uint32_t interp, int dtype, void *stream) {
    if (B == 0) return 0;
    if (!inputs || !embeddings || !offsets_host || !outputs) return SDN_E_BADARG;
    if (gridtype > 1 || interp > 1) return SDN_E_UNSUPPORTED;
    LevelParams lp;
    int rc = fill_levels(lp, offsets_host, L, S, H);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SDN_F32)
        rc = dispatch_fwd<float>(D, C, inputs, (const float *)embeddings, (float *)outputs, B, L, lp, (float *)dy_dx, gridtype,
                                 align_corners != 0, interp, st);
    else if (dtype == SDN_F16)
        rc = dispatch_fwd<__half>(D, C, inputs, (const __half *)embeddings, (__half *)outputs, B, L, lp, (__half *)dy_dx, gridtype,
                                  align_corners != 0, interp, st);
    else
        return SDN_E_UNSUPPORTED;
    return rc ? rc : sdn_launch_status();
}

int sdn_grid_encode_backward(const void *grad, const float *inputs, const int32_t *offsets_host, void *grad_embeddings, uint32_t B,
                             uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, const void *dy_dx, void *grad_inputs,
                             uint32_t gridtype, int align_corners, uint32_t interp, int dtype, void *stream) {
    return sdn_grid_encode_backward_det(grad, inputs, offsets_host, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, gridtype, align_corners,
                                        interp, dtype, nullptr, stream);
}

// The same with an order-independent table gradient: det_scratch = offsets_host[L] * C 64-bit words of device memory (cleared by the
// call), or NULL = the plain atomics.  Two runs on the same inputs then give the same bits (tests; resumed training).
int sdn_grid_encode_backward_det(const void *grad, const float *inputs, const int32_t *offsets_host, void *grad_embeddings, uint32_t B,
                                 uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, const void *dy_dx, void *grad_inputs,
                                 uint32_t gridtype, int align_corners, uint32_t interp, int dtype, void *det_scratch, void *stream) {
    if (B == 0) return 0;
    if (!grad || !inputs || !offsets_host || !grad_embeddings) return SDN_E_BADARG;
    if (gridtype > 1 || interp > 1) return SDN_E_UNSUPPORTED;
    LevelParams lp;
    int rc = fill_levels(lp, offsets_host, L, S, H);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SDN_F32)
        rc = dispatch_bwd<float>(D, C, (const float *)grad, inputs, (float *)grad_embeddings, B, L, lp, (const float *)dy_dx,
                                 (float *)grad_inputs, gridtype, align_corners != 0, interp, st, (long long *)det_scratch);
    else if (dtype == SDN_F16)
        rc = dispatch_bwd<__half>(D, C, (const __half *)grad, inputs, (__half *)grad_embeddings, B, L, lp, (const __half *)dy_dx,
                                  (__half *)grad_inputs, gridtype, align_corners != 0, interp, st, (long long *)det_scratch);
    else
        return SDN_E_UNSUPPORTED;
    return rc ? rc : sdn_launch_status();
}

}  // extern "C"
