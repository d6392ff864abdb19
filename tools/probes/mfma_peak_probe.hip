// What does v_mfma_f32_32x32x16_f16 sustain on this chip when NOTHING else competes -- the practical matrix roof under the 2.5 PFLOP/s
// this repository prices against (MI355X_MICROARCH.md)?
//   A: 4 waves per SIMD, each a stream of independent MFMAs from registers (4 accumulators in rotation): pure issue / pipe rate
//   B: as A, but every MFMA's A operand comes from a ds_read_b128 of a 32-KiB LDS stage (the fused field kernel's weight path)
//   C: as B plus the accumulator -> fp16 operand conversion of a layer boundary (32 v_cvt_pk_f16_f32 + 32 v_pk_max_f16 per 32 MFMAs)
// Also prints the shader clock seen by s_memtime over the launch (a power-capped clock moves the roof, not the kernel).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak_probe mfma_peak_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
#include <stdint.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// pseudo-random value in [-1, 1): the matrix pipes' power (and with it the clock the chip grants) depends on the operands' bits -- constant or
// zero operands read 25 % faster than random ones -- so every variant multiplies random numbers
__device__ __forceinline__ float rnd(uint32_t x) {
    x = (x ^ 61u) ^ (x >> 16); x *= 9u; x ^= x >> 4; x *= 0x27d4eb2du; x ^= x >> 15;
    return (float)(x & 0xffffu) * (2.0f / 65536.0f) - 1.0f;
}

template <int MODE>
__global__ void __launch_bounds__(256, 2) k_mfma(float *out, unsigned long long *clk, int iters) {
    __shared__ __attribute__((aligned(16))) _Float16 s_w[16384];      // 32 KiB
    const uint32_t lane = threadIdx.x & 63u;
    for (int i = threadIdx.x; i < 16384; i += 256) s_w[i] = (_Float16)rnd((uint32_t)i);
    __syncthreads();
    half8 a, b;
    for (int k = 0; k < 8; k++) { a[k] = (_Float16)rnd(lane * 8 + k); b[k] = (_Float16)rnd(4096 + lane * 8 + k); }
    f32x16 acc[4];
    for (int m = 0; m < 4; m++) for (int v = 0; v < 16; v++) acc[m][v] = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
        #pragma unroll
        for (int ks = 0; ks < 8; ks++) {
            #pragma unroll
            for (int m = 0; m < 4; m++) {
                if (MODE >= 1) a = *reinterpret_cast<const half8 *>(s_w + (((ks * 4 + m) * 64 + lane) * 8 & 16383));
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
            }
        }
        if (MODE == 2) {      // a layer boundary: 64 accumulator registers -> 32 packed fp16 pairs with ReLU, folded back into b
            uint32_t x = 0;
            #pragma unroll
            for (int m = 0; m < 4; m++)
                #pragma unroll
                for (int v = 0; v < 16; v += 2) {
                    half2v p = {(_Float16)acc[m][v], (_Float16)acc[m][v + 1]};
                    const half2v z = {(_Float16)0.0f, (_Float16)0.0f};
                    p = __builtin_elementwise_max(p, z);
                    x ^= __builtin_bit_cast(uint32_t, p);
                }
            b[0] = __builtin_bit_cast(half2v, x)[0];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    float s = 0;
    for (int m = 0; m < 4; m++) for (int v = 0; v < 16; v++) s += acc[m][v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

// D: the fp32 matrix instruction of the fp32 fused field kernel (v_mfma_f32_32x32x2_f32: 64 cycles, 4 096 FLOP), from registers
__global__ void __launch_bounds__(256, 2) k_mfma_f32(float *out, int iters) {
    const uint32_t lane = threadIdx.x & 63u;
    float a = rnd(lane), b = rnd(4096 + lane);
    f32x16 acc[4];
    for (int m = 0; m < 4; m++) for (int v = 0; v < 16; v++) acc[m][v] = 0.0f;
    for (int it = 0; it < iters; it++) {
        #pragma unroll
        for (int ks = 0; ks < 8; ks++)
            #pragma unroll
            for (int m = 0; m < 4; m++) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
    }
    float s = 0;
    for (int m = 0; m < 4; m++) for (int v = 0; v < 16; v++) s += acc[m][v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// E: v_mfma_f32_16x16x32_f16 (16 384 FLOP, a quarter of the accumulator registers per tile), 8 accumulators in rotation, from registers
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256, 2) k_mfma_16(float *out, int iters) {
    const uint32_t lane = threadIdx.x & 63u;
    half8 a, b;
    for (int k = 0; k < 8; k++) { a[k] = (_Float16)rnd(lane * 8 + k); b[k] = (_Float16)rnd(4096 + lane * 8 + k); }
    f32x4 acc[8];
    for (int m = 0; m < 8; m++) for (int v = 0; v < 4; v++) acc[m][v] = 0.0f;
    for (int it = 0; it < iters; it++) {
        #pragma unroll
        for (int ks = 0; ks < 8; ks++)
            #pragma unroll
            for (int m = 0; m < 8; m++) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m], 0, 0, 0);
    }
    float s = 0;
    for (int m = 0; m < 8; m++) for (int v = 0; v < 4; v++) s += acc[m][v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// F: a wide layer of the field network on 16x16x32 tiles, data flow as it would be: 128 outputs x 32 points = 8 m-tiles x 2 n-tiles of
// accumulators (64 registers, as today), 4 k-steps, the A operand of every (m-tile, k-step) read once from LDS and used for both n-tiles
// (32 ds_read_b128 + 64 MFMAs per layer), then the layer boundary: lane (g, n) holds rows 16 t + 4 g + v of tile t, so tiles 2 s and 2 s + 1
// ARE the 8 k-values of k-step s for that lane group -- 32 v_cvt_pk_f16_f32 + 32 v_pk_max_f16, no permutation (k-order baked into the packing).
__global__ void __launch_bounds__(256, 2) k_layer_16(float *out, int layers) {
    __shared__ __attribute__((aligned(16))) _Float16 s_w[16384];      // one 128 x 128 fp16 layer
    const uint32_t lane = threadIdx.x & 63u;
    for (int i = threadIdx.x; i < 16384; i += 256) s_w[i] = (_Float16)(0.2165f * rnd((uint32_t)i));   // uniform, variance 2 / 128
    __syncthreads();
    u32x4 b[2][4];      // B operands as packed words (bit-cast to half8 at the MFMA)
    for (int nt = 0; nt < 2; nt++) for (int ks = 0; ks < 4; ks++) for (int k = 0; k < 4; k++) b[nt][ks][k] = __builtin_bit_cast(uint32_t, half2v{(_Float16)fabsf(rnd(lane * 64 + nt * 32 + ks * 8 + 2 * k)), (_Float16)fabsf(rnd(lane * 64 + nt * 32 + ks * 8 + 2 * k + 1))});
    f32x4 acc[8][2];
    for (int l = 0; l < layers; l++) {
        asm volatile("" ::: "memory");      // another layer's weights: the stage is read again, not kept in registers
        #pragma unroll
        for (int mt = 0; mt < 8; mt++) for (int nt = 0; nt < 2; nt++) for (int v = 0; v < 4; v++) acc[mt][nt][v] = 0.0f;
        #pragma unroll
        for (int ks = 0; ks < 4; ks++)
            #pragma unroll
            for (int mt = 0; mt < 8; mt++) {
                const half8 a = *reinterpret_cast<const half8 *>(s_w + ((ks * 8 + mt) * 64 + lane) * 8);
                acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(half8, b[0][ks]), acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(half8, b[1][ks]), acc[mt][1], 0, 0, 0);
            }
        const half2v z = {(_Float16)0.0f, (_Float16)0.0f};
        #pragma unroll
        for (int nt = 0; nt < 2; nt++)
            #pragma unroll
            for (int ks = 0; ks < 4; ks++)
                #pragma unroll
                for (int j = 0; j < 4; j++) {
                    const f32x4 &t = acc[2 * ks + j / 2][nt];
                    half2v p = {(_Float16)t[2 * (j % 2)], (_Float16)t[2 * (j % 2) + 1]};
                    p = __builtin_elementwise_max(p, z);
                    b[nt][ks][j] = __builtin_bit_cast(uint32_t, p);
                }
    }
    float s = 0;
    for (int nt = 0; nt < 2; nt++) for (int ks = 0; ks < 4; ks++) for (int k = 0; k < 4; k++) s += (float)b[nt][ks][k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// G: the same layer on today's 32x32x16 tiles: 4 m-tiles of 16 registers, 8 k-steps, 32 ds_read_b128 + 32 MFMAs, the same boundary
__global__ void __launch_bounds__(256, 2) k_layer_32(float *out, int layers) {
    __shared__ __attribute__((aligned(16))) _Float16 s_w[16384];
    const uint32_t lane = threadIdx.x & 63u;
    for (int i = threadIdx.x; i < 16384; i += 256) s_w[i] = (_Float16)(0.2165f * rnd((uint32_t)i));   // uniform, variance 2 / 128
    __syncthreads();
    u32x4 b[8];
    for (int ks = 0; ks < 8; ks++) for (int k = 0; k < 4; k++) b[ks][k] = __builtin_bit_cast(uint32_t, half2v{(_Float16)fabsf(rnd(lane * 64 + ks * 8 + 2 * k)), (_Float16)fabsf(rnd(lane * 64 + ks * 8 + 2 * k + 1))});
    f32x16 acc[4];
    for (int l = 0; l < layers; l++) {
        asm volatile("" ::: "memory");
        #pragma unroll
        for (int mt = 0; mt < 4; mt++) for (int v = 0; v < 16; v++) acc[mt][v] = 0.0f;
        #pragma unroll
        for (int ks = 0; ks < 8; ks++)
            #pragma unroll
            for (int mt = 0; mt < 4; mt++) {
                const half8 a = *reinterpret_cast<const half8 *>(s_w + ((ks * 4 + mt) * 64 + lane) * 8);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(half8, b[ks]), acc[mt], 0, 0, 0);
            }
        const half2v z = {(_Float16)0.0f, (_Float16)0.0f};
        #pragma unroll
        for (int ks = 0; ks < 8; ks++)
            #pragma unroll
            for (int j = 0; j < 4; j++) {
                half2v p = {(_Float16)acc[ks / 2][8 * (ks % 2) + 2 * j], (_Float16)acc[ks / 2][8 * (ks % 2) + 2 * j + 1]};
                p = __builtin_elementwise_max(p, z);
                b[ks][j] = __builtin_bit_cast(uint32_t, p);
            }
    }
    float s = 0;
    for (int ks = 0; ks < 8; ks++) for (int k = 0; k < 4; k++) s += (float)b[ks][k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename K>
static void run_layer(const char *name, K kern, int cus) {
    const int layers = 2000, wgs = cus * 16;
    float *out; (void)hipMalloc(&out, (size_t)wgs * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, out, layers);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flop = (double)wgs * 4 * layers * 2.0 * 128 * 128 * 32;
    printf("%-64s %8.3f ms  %7.1f TFLOP/s  = %.3f of 2500\n", name, best, flop / best / 1e9, flop / best / 1e9 / 2500.0);
    (void)hipFree(out);
}

static void run_16(int cus) {
    const int iters = 2000, wgs = cus * 16;
    float *out; (void)hipMalloc(&out, (size_t)wgs * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_mfma_16, dim3(wgs), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flop = (double)wgs * 4 * iters * 64 * 16384.0;
    printf("%-64s %8.3f ms  %7.1f TFLOP/s  = %.3f of 2500\n", "E  v_mfma_f32_16x16x32_f16 from registers, 8 accumulators", best, flop / best / 1e9, flop / best / 1e9 / 2500.0);
    (void)hipFree(out);
}

static void run_f32(int cus) {
    const int iters = 1000, wgs = cus * 16;
    float *out; (void)hipMalloc(&out, (size_t)wgs * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_mfma_f32, dim3(wgs), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flop = (double)wgs * 4 * iters * 32 * 4096.0;
    printf("%-64s %8.3f ms  %7.1f TFLOP/s  = %.3f of 157.3\n", "D  v_mfma_f32_32x32x2_f32 from registers", best, flop / best / 1e9, flop / best / 1e9 / 157.3);
    (void)hipFree(out);
}

template <int MODE>
static void run(const char *name, int cus) {
    const int iters = 2000, wgs = cus * 2 * 8;       // 2 workgroups of 4 waves per CU resident (8 waves = 2 per SIMD ... x 2), 8 rounds
    float *out; unsigned long long *clk;
    (void)hipMalloc(&out, (size_t)wgs * 256 * 4); (void)hipMalloc(&clk, 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f; unsigned long long c[2] = {0, 0};
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_mfma<MODE>, dim3(wgs), dim3(256), 0, 0, out, clk, iters);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) { best = ms; (void)hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost); }
    }
    const double flop = (double)wgs * 4 /*waves*/ * iters * 32 /*mfma*/ * 32768.0;
    // workgroup 0: s_memtime ticks over its life against the 100 MHz wall clock, and against the ticks its MFMAs need when 4 waves share a pipe
    const double mhz = c[1] ? (double)c[0] / (double)c[1] * 100.0 : 0.0;
    printf("%-64s %8.3f ms  %7.1f TFLOP/s  = %.3f of 2500 | wg0: %.2f M s_memtime ticks (%.0f MHz against the wall clock); 4 waves x %d MFMAs x 32 cycles = %.2f M\n",
           name, best, flop / best / 1e9, flop / best / 1e9 / 2500.0, c[0] / 1e6, mhz, iters * 32, 4.0 * iters * 32 * 32 / 1e6);
    (void)hipFree(out); (void)hipFree(clk);
}

int main() {
    int cus = 256; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    int khz = 0; (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    printf("CUs %d, reported peak shader clock %.0f MHz\n", cus, khz / 1000.0);
    run<0>("A  MFMA from registers, 4 accumulators in rotation", cus);
    run<1>("B  A operand from LDS (ds_read_b128 per MFMA)", cus);
    run<2>("C  B + accumulator -> fp16 + ReLU conversion every 32 MFMAs", cus);
    run_f32(cus);
    run_16(cus);
    run_layer("F  128x128 layer chain on 16x16x32 tiles (LDS A, boundary conv.)", k_layer_16, cus);
    run_layer("G  128x128 layer chain on 32x32x16 tiles (LDS A, boundary conv.)", k_layer_32, cus);
    return 0;
}
