// latbench.hip — dependent-issue latency of the instructions on the biquad's critical path (gfx950), one wave.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o latbench latbench.hip && ./latbench   (no contraction: like the kernels)
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAIN(NAME, BODY)                                                                   \
    __global__ void NAME(double *out, unsigned long long *cyc, double b1, double b2, int n) { \
        double y1 = out[threadIdx.x], y2 = out[threadIdx.x + 64], p = out[threadIdx.x + 128]; \
        float f = (float)y1;                                                                \
        unsigned long long t0 = __builtin_readcyclecounter();                               \
        for (int i = 0; i < n; ++i) {                                                       \
            _Pragma("unroll") for (int k = 0; k < 16; ++k) { BODY }                          \
        }                                                                                   \
        unsigned long long t1 = __builtin_readcyclecounter();                               \
        out[threadIdx.x] = y1 + y2 + (double)f;                                             \
        if (threadIdx.x == 0) *cyc = t1 - t0;                                               \
    }

CHAIN(k_mul, y1 = y1 * b1;)
CHAIN(k_add, y1 = y1 + b1;)
CHAIN(k_fma, y1 = __builtin_fma(y1, b1, b2);)
CHAIN(k_cvt, f = (float)y1; y1 = (double)f;)
CHAIN(k_cvt_only32, f = (float)y1; y1 = __hiloint2double(__float_as_int(f), __double2hiint(y1));)
CHAIN(k_mulf32, f = f * 1.0001f;)
CHAIN(k_full, { const float y = (float)((p - b1 * y1) - b2 * y2); y2 = y1; y1 = (double)y; })
CHAIN(k_full_or0, { const double u1 = (y1 != y1 || y1 == 0.0) ? 0.0 : y1; const double u2 = (y2 != y2 || y2 == 0.0) ? 0.0 : y2; const float y = (float)((p - b1 * u1) - b2 * u2); y2 = u1; y1 = (double)y; })

int main() {
    double *d; unsigned long long *c;
    hipMalloc(&d, 4096); hipMalloc(&c, 8);
    double h[512]; for (int i = 0; i < 512; ++i) h[i] = 0.001 * (i + 1);
    const int n = 4096;
#define RUN(NAME, OPS)                                                                       \
    for (int lanes : {64, 32, 1}) {                                                          \
        hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);                                        \
        hipLaunchKernelGGL(NAME, dim3(1), dim3(lanes), 0, 0, d, c, 0.999, -0.5, n);           \
        hipLaunchKernelGGL(NAME, dim3(1), dim3(lanes), 0, 0, d, c, 0.999, -0.5, n);           \
        unsigned long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);                  \
        printf("%-14s lanes=%2d  %.2f cycles per step (%d dependent op%s)\n", #NAME, lanes, (double)cy / (n * 16.0), OPS, OPS > 1 ? "s" : ""); \
    }
    RUN(k_mul, 1) RUN(k_add, 1) RUN(k_fma, 1) RUN(k_cvt, 2) RUN(k_cvt_only32, 1) RUN(k_mulf32, 1) RUN(k_full, 5) RUN(k_full_or0, 7)
    return 0;
}
