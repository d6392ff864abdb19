// Measurement probe (not part of the library): (1) does a high-priority HIP stream's kernel get CU slots ahead of a chip-filling
// kernel on a normal-priority stream?  (2) what does a cross-stream dependency (event record + stream wait) cost per hop against
// back-to-back launches in one stream?      hipcc --offload-arch=gfx950 -O2 -o prio_probe prio_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// "field-like": 512 threads, 64 KiB of LDS, ~128 VGPRs worth of state is not needed: LDS alone limits it to 2 workgroups per CU
__global__ void __launch_bounds__(512) k_hog(float *out, int spin_us) {
    __shared__ float lds[16384];
    lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    const long long t0 = wall_clock64();           // 100 MHz
    float a = lds[(threadIdx.x * 7) & 16383];
    while (wall_clock64() - t0 < (long long)spin_us * 100) a = a * 1.0001f + 0.5f;
    if (a == 12345.f) out[0] = a;
}
// "marcher-like": 256 threads, 36 KiB of LDS, latency-bound for spin_us
__global__ void __launch_bounds__(256) k_probe(float *out, int spin_us) {
    __shared__ float lds[9216];
    lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    const long long t0 = wall_clock64();
    float a = lds[(threadIdx.x * 5) & 8191];
    while (wall_clock64() - t0 < (long long)spin_us * 100) a = a * 1.0001f + 0.25f;
    if (a == 12345.f) out[1] = a;
}
__global__ void k_tiny(float *out) { if (out[2] == 12345.f) out[3] = 1.f; }

static float probe_under_hog(hipStream_t hog_st, hipStream_t probe_st, float *buf, int hog_wgs, int probe_wgs, bool with_hog) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (with_hog) {
        for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_hog, dim3(hog_wgs), dim3(512), 0, hog_st, buf, 35);   // three chip-filling launches back to back
        // let the first one get going
        hipEvent_t w; CK(hipEventCreate(&w));
        struct timespec ts = {0, 60000}; nanosleep(&ts, nullptr);
        (void)w;
    }
    CK(hipEventRecord(e0, probe_st));
    hipLaunchKernelGGL(k_probe, dim3(probe_wgs), dim3(256), 0, probe_st, buf, 10);
    CK(hipEventRecord(e1, probe_st));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f;
}

int main() {
    int least = 0, greatest = 0;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    printf("priority range: least %d greatest %d\n", least, greatest);
    hipStream_t s_norm, s_norm2, s_high, s_low;
    CK(hipStreamCreateWithPriority(&s_norm, hipStreamNonBlocking, 0));
    CK(hipStreamCreateWithPriority(&s_norm2, hipStreamNonBlocking, 0));
    CK(hipStreamCreateWithPriority(&s_high, hipStreamNonBlocking, greatest));
    CK(hipStreamCreateWithPriority(&s_low, hipStreamNonBlocking, least));
    float *buf; CK(hipMalloc(&buf, 4096)); CK(hipMemset(buf, 0, 4096));
    // warm-up
    for (int i = 0; i < 3; i++) { probe_under_hog(s_norm, s_norm2, buf, 1200, 2500, true); }
    for (int rep = 0; rep < 3; rep++) {
        for (int pw : {2500, 300}) {
            printf("probe %4d WGs x 10 us | alone %.1f us | beside hog: normal/normal %.1f us | high/normal %.1f us | normal/low %.1f us | high/low %.1f us\n", pw,
                   probe_under_hog(s_norm, s_norm2, buf, 1200, pw, false), probe_under_hog(s_norm, s_norm2, buf, 1200, pw, true),
                   probe_under_hog(s_norm, s_high, buf, 1200, pw, true), probe_under_hog(s_low, s_norm2, buf, 1200, pw, true),
                   probe_under_hog(s_low, s_high, buf, 1200, pw, true));
        }
    }
    // (2) dependency hop: 200 tiny kernels back to back in one stream, against 200 alternating between two streams with events
    const int n = 200;
    std::vector<hipEvent_t> ev(n);
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(t0, s_norm));
        for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s_norm, buf);
        CK(hipEventRecord(t1, s_norm));
        CK(hipDeviceSynchronize());
        float same = 0; CK(hipEventElapsedTime(&same, t0, t1));
        CK(hipEventRecord(t0, s_norm));
        for (int i = 0; i < n; i++) {
            hipStream_t st = (i & 1) ? s_high : s_norm, other = (i & 1) ? s_norm : s_high;
            hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, buf);
            CK(hipEventRecord(ev[i], st));
            CK(hipStreamWaitEvent(other, ev[i], 0));
        }
        CK(hipEventRecord(t1, (n & 1) ? s_high : s_norm));
        CK(hipDeviceSynchronize());
        float cross = 0; CK(hipEventElapsedTime(&cross, t0, t1));
        printf("tiny kernels: same stream %.2f us per launch | alternating two streams behind events %.2f us per launch\n", same * 1e3f / n, cross * 1e3f / n);
    }
    return 0;
}
