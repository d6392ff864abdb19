// Accuracy of the frequency features of the fused field kernel after fp16 rounding, for three ways of producing them:
//   poly  : shared Cody-Waite reduction + two polynomials at the octaves 2^0 / 2^5, four angle doublings  (round 3's kernel)
//   hw    : v_sin_f32 / v_cos_f32 on revolutions at the octaves 2^0 / 2^5, four angle doublings
//   ref   : the reference kernel's own form, float sin(x 2^f) and sin(x 2^f + float(pi/2))   (freqencoder.cu:52-56, OCML sinf)
// against the exactly rounded fp16 of the double-precision value.  Prints per method: max |error| before rounding and the fraction of
// fp16 features that differ from the exactly rounded one.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o sincos_probe sincos_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

__device__ __forceinline__ void fast_sincos(float a, float &sn, float &cs) {
    const float k = rintf(a * 0.31830988618379067f);
    float r = __builtin_fmaf(-k, 3.140625f, a);
    r = __builtin_fmaf(-k, 9.67502593994140625e-4f, r);
    r = __builtin_fmaf(-k, 1.509957990978376e-7f, r);
    const float r2 = r * r;
    float p = __builtin_fmaf(r2, 2.6083159809786593e-6f, -1.9810690719168633e-4f);
    p = __builtin_fmaf(p, r2, 8.3330785855650902e-3f);
    p = __builtin_fmaf(p, r2, -1.6666659712791443e-1f);
    const float s = __builtin_fmaf(r * r2, p, r);
    float q = __builtin_fmaf(r2, -2.6051615e-07f, 2.4760495e-05f);
    q = __builtin_fmaf(q, r2, -1.3888378e-03f);
    q = __builtin_fmaf(q, r2, 4.1666638e-02f);
    q = __builtin_fmaf(q, r2, -0.5f);
    const float c = __builtin_fmaf(q, r2, 1.0f);
    const int sign = ((int)k & 1) << 31;
    sn = __int_as_float(__float_as_int(s) ^ sign);
    cs = __int_as_float(__float_as_int(c) ^ sign);
}

// out[method][0] = max abs error (as float bits, atomicMax on positive floats), out[method][1] = differing fp16 features, out[3][0] = features
__global__ void probe(const float *x, int n, unsigned int *maxerr, unsigned long long *diff) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float xv = x[i];
    float feat[3][20];
    for (int m = 0; m < 2; m++) {
        for (int hh = 0; hh < 2; hh++) {
            const float fs = hh ? 32.0f : 1.0f;
            float s, c;
            if (m == 0) fast_sincos(xv * fs, s, c);
            else { const float rev = xv * (fs * 0.15915494309189535f); s = __builtin_amdgcn_sinf(rev); c = __builtin_amdgcn_cosf(rev); }
            for (int f = 0; f < 5; f++) {
                feat[m][2 * (5 * hh + f)] = s; feat[m][2 * (5 * hh + f) + 1] = c;
                const float s2 = s + s;
                const float sn = s2 * c, cn = __builtin_fmaf(-s2, s, 1.0f);
                s = sn; c = cn;
            }
        }
    }
    for (int f = 0; f < 10; f++) {
        const float a = xv * exp2f((float)f);
        feat[2][2 * f] = sinf(a);
        feat[2][2 * f + 1] = sinf(a + 1.5707963267948966f);
    }
    for (int f = 0; f < 10; f++) {
        const double a = (double)xv * exp2((double)f);
        const double ex[2] = {sin(a), cos(a)};
        for (int k = 0; k < 2; k++) {
            const _Float16 want = (_Float16)ex[k];
            for (int m = 0; m < 3; m++) {
                const float got = feat[m][2 * f + k];
                const float e = fabsf((float)((double)got - ex[k]));
                atomicMax(&maxerr[m * 10 + f], __float_as_uint(e));
                if ((_Float16)got != want) atomicAdd(&diff[m * 10 + f], 1ull);
            }
        }
    }
}

int main() {
    const int n = 1 << 22;
    float *hx = (float *)malloc(n * sizeof(float));
    srand(1);
    for (int i = 0; i < n; i++) hx[i] = ((float)rand() / (float)RAND_MAX * 2.0f - 1.0f) * 1.5f;
    float *dx; unsigned int *dm; unsigned long long *dd;
    (void)hipMalloc(&dx, n * sizeof(float)); hipMalloc(&dm, 30 * sizeof(unsigned int)); hipMalloc(&dd, 30 * sizeof(unsigned long long));
    hipMemcpy(dx, hx, n * sizeof(float), hipMemcpyHostToDevice);
    hipMemset(dm, 0, 30 * sizeof(unsigned int)); hipMemset(dd, 0, 30 * sizeof(unsigned long long));
    hipLaunchKernelGGL(probe, dim3(n / 256), dim3(256), 0, 0, dx, n, dm, dd);
    unsigned int hm[30]; unsigned long long hd[30];
    hipMemcpy(hm, dm, sizeof(hm), hipMemcpyDeviceToHost); hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
    const char *names[3] = {"poly+doubling", "hw+doubling", "reference form"};
    for (int m = 0; m < 3; m++) {
        printf("%-15s", names[m]);
        double tot = 0;
        for (int f = 0; f < 10; f++) {
            float e; memcpy(&e, &hm[m * 10 + f], 4);
            printf(" 2^%d: %.1e/%.3f%%", f, e, 100.0 * hd[m * 10 + f] / (2.0 * n));
            tot += hd[m * 10 + f];
        }
        printf("  | all octaves %.3f%% of features differ from the exactly rounded fp16\n", 100.0 * tot / (20.0 * n));
    }
    return 0;
}
