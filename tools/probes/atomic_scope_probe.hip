// Where do no-return float atomics execute on MI355X (8 XCDs, one L2 each), and what does it cost?
//   A: 16 M atomic adds to random rows of ONE 2 MiB table (2^19 rows x 4 B), default (agent) scope        -- what k_grid_bwd does per level
//   B: the same adds, each XCD to its OWN copy of the table (8 copies; row + xcc_id * rows), default scope
//   C: as B with workgroup-scope atomics (__hip_atomic_fetch_add ... __HIP_MEMORY_SCOPE_WORKGROUP): the XCD's L2 is their coherence point
//   D: as A with workgroup scope (INCORRECT across XCDs; timing only, to separate scope from privatisation)
// Prints ms per launch and the sum check of B / C after reducing the copies.   hipcc --offload-arch=gfx950 -O3 -o atomic_scope_probe atomic_scope_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u; }   // HW_REG_XCC_ID bits 0..3

template <int MODE>
__global__ void __launch_bounds__(256) k_scatter(const uint32_t *__restrict__ idx, uint32_t n, float *table, uint32_t rows) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r = idx[i];
    if (MODE == 1 || MODE == 2) r += xcc_id() * rows;
    if (MODE == 0 || MODE == 1) unsafeAtomicAdd(table + r, 1.0f);
    else (void)__hip_atomic_fetch_add(table + r, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

int main() {
    const uint32_t rows = 1u << 19, n = 1u << 24;
    uint32_t *h = (uint32_t *)malloc(n * 4);
    srand(3);
    for (uint32_t i = 0; i < n; i++) h[i] = ((uint32_t)rand() * 2654435761u) >> 13;
    uint32_t *d_idx; float *d_tab;
    (void)hipMalloc(&d_idx, n * 4); (void)hipMalloc(&d_tab, (size_t)rows * 8 * 4);
    (void)hipMemcpy(d_idx, h, n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char *names[4] = {"A shared table, default scope", "B per-XCD copies, default scope", "C per-XCD copies, workgroup scope", "D shared table, workgroup scope (wrong sums)"};
    float *hs = (float *)malloc((size_t)rows * 8 * 4);
    for (int mode = 0; mode < 4; mode++) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            (void)hipMemset(d_tab, 0, (size_t)rows * 8 * 4);
            (void)hipEventRecord(e0, 0);
            switch (mode) {
                case 0: hipLaunchKernelGGL(k_scatter<0>, dim3(n / 256), dim3(256), 0, 0, d_idx, n, d_tab, rows); break;
                case 1: hipLaunchKernelGGL(k_scatter<1>, dim3(n / 256), dim3(256), 0, 0, d_idx, n, d_tab, rows); break;
                case 2: hipLaunchKernelGGL(k_scatter<2>, dim3(n / 256), dim3(256), 0, 0, d_idx, n, d_tab, rows); break;
                default: hipLaunchKernelGGL(k_scatter<3>, dim3(n / 256), dim3(256), 0, 0, d_idx, n, d_tab, rows); break;
            }
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        (void)hipMemcpy(hs, d_tab, (size_t)rows * 8 * 4, hipMemcpyDeviceToHost);
        double total = 0; uint32_t used = 0;
        for (size_t k = 0; k < (size_t)rows * 8; k++) { total += hs[k]; if (k % rows == 0 && hs[k] != 0) used++; }
        printf("%-45s %8.3f ms  (%.1f G adds/s)  sum %.0f of %u\n", names[mode], best, n / best / 1e6, total, n);
    }
    return 0;
}
