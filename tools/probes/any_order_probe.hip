// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950 (hip_ext.h says "not supported on GFX9xx")?
// A: one workgroup spinning ~T; B (any-order): one workgroup spinning ~T; C (ordinary): stamps the time it starts.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/any_order_probe.hip -o /tmp/any_order_probe && /tmp/any_order_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <chrono>

__global__ void spin(unsigned long long ticks, unsigned long long *out, int slot) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0) { out[2 * slot] = t0; out[2 * slot + 1] = wall_clock64(); }
}

int main() {
    unsigned long long *d, h[8];
    hipMalloc(&d, sizeof(h));
    hipStream_t st;
    hipStreamCreate(&st);
    int rate_khz = 0;
    hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    const unsigned long long ticks = (unsigned long long)rate_khz * 300 / 1000;   // 300 us
    for (int mode = 0; mode < 3; mode++) {
        hipMemsetAsync(d, 0, sizeof(h), st);
        hipStreamSynchronize(st);
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, ticks, d, 0);
        if (mode == 0) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, ticks, d, 1);
        else hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, nullptr, nullptr, mode == 1 ? hipExtAnyOrderLaunch : 0u, ticks, d, 1);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, 1ull, d, 2);
        hipStreamSynchronize(st);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        const double k = 1e3 / rate_khz;   // us per tick
        printf("%s: wall %.0f us;  A [%.0f, %.0f]  B [%.0f, %.0f]  C starts %.0f us after A started\n",
               mode == 0 ? "ordinary launches        " : (mode == 1 ? "B with hipExtAnyOrderLaunch" : "B with hipExt, flags 0     "), us, 0.0,
               (h[1] - h[0]) * k, (double)(long long)(h[2] - h[0]) * k, (double)(long long)(h[3] - h[0]) * k, (double)(long long)(h[4] - h[0]) * k);
    }
    return 0;
}
