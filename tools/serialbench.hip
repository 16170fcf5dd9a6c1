// serialbench.hip — the Filter stage's recurrence loop by itself: one wave, 32 rows, tile in LDS; cycles per sample-step of variants.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) double lds_double;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
constexpr int SUB = 128, PITCH = SUB + 2, ROWS = 32;
__device__ __forceinline__ double or0(double v) { return (v != v || v == 0.0) ? 0.0 : v; }

template <int CHECK, int WRITE>
__device__ __forceinline__ void block8(const double (&pv)[8], double b1, double b2, double &u1, double &u2, lds_f32x4 *dst) {
    const double u1_in = u1, u2_in = u2;
    f32x4 y4[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float y = (float)((pv[i] - b1 * u1) - b2 * u2);
        y4[i >> 2][i & 3] = y;
        u2 = u1;
        u1 = (double)y;
    }
    if (CHECK) {
        if (!(u1 == u1 && u2 == u2)) {
            u1 = u1_in;
            u2 = u2_in;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float y = (float)((pv[i] - b1 * or0(u1)) - b2 * or0(u2));
                y4[i >> 2][i & 3] = y;
                u2 = or0(u1);
                u1 = (double)y;
            }
            u1 = or0(u1);
        }
    }
    if (WRITE) {
        dst[0] = y4[0];
        dst[1] = y4[1];
    } else {
        asm volatile("" ::"v"(y4[0]), "v"(y4[1]));
    }
}

// V: 0 = ping-pong, check per 8, writes; 1 = no check; 2 = no writes no check; 3 = P not reloaded (registers), no writes no check
template <int V>
__global__ void __launch_bounds__(1024) k_serial(double *out, unsigned long long *cyc, double b1, double b2, int n) {
    __shared__ __attribute__((aligned(16))) double tile[ROWS * PITCH + 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < ROWS * PITCH; i += blockDim.x) tile[i] = 0.001 * (i % 97);
    __syncthreads();
    double u1 = out[lane], u2 = out[lane + 64];
    unsigned long long total = 0;
    for (int it = 0; it < n; ++it) {
        if (wave == 0 && lane < ROWS) {
            const unsigned long long s0 = __builtin_readcyclecounter();
            const uint32_t row = ((uint32_t)(uintptr_t)(lds_double *)(tile + (size_t)lane * PITCH)) & 0x3ffffu;
            const lds_double *pr = (const lds_double *)(uintptr_t)row;
            lds_f32x4 *yr = (lds_f32x4 *)(uintptr_t)row;
            double pa[8], pb[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) pa[i] = pr[i];
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (V == 3) {
#pragma unroll
                for (int i = 0; i < 8; ++i) pb[i] = pr[8 + i];
            }
            for (int t0 = 0; t0 < SUB; t0 += 16) {
                if (V != 3) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) pb[i] = pr[8 + i];
                    __builtin_amdgcn_sched_barrier(0);
                }
                block8<V == 0, V <= 1>(pa, b1, b2, u1, u2, yr);
                if (V != 3) {
                    const int next = t0 + 16 < SUB ? 16 : 0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) pa[i] = pr[next + i];
                    __builtin_amdgcn_sched_barrier(0);
                }
                block8<V == 0, V <= 1>(pb, b1, b2, u1, u2, yr + 2);
                pr += 16;
                yr += 4;
            }
            total += __builtin_readcyclecounter() - s0;
        }
        __syncthreads();
        // refill (all waves), like park
        for (int i = threadIdx.x; i < ROWS * PITCH; i += blockDim.x) tile[i] = 0.001 * ((i + it) % 97);
        __syncthreads();
    }
    if (wave == 0 && lane < ROWS) out[lane] = u1 + u2;
    if (threadIdx.x == 0) *cyc = total;
}


// interleaved: the next block's reads and this block's / the previous block's y stores sit between the chain's dependent steps
template <int CHECK>
__device__ __forceinline__ void block8i(const double (&pv)[8], double (&pn)[8], const lds_double *src, double b1, double b2, double &u1, double &u2,
                                        lds_f32x4 *dst, f32x4 &pend, lds_f32x4 *pend_dst) {
    const double u1_in = u1, u2_in = u2;
    f32x4 y4[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float y = (float)((pv[i] - b1 * u1) - b2 * u2);
        y4[i >> 2][i & 3] = y;
        u2 = u1;
        u1 = (double)y;
        __builtin_amdgcn_sched_barrier(0);
        if ((i & 1) == 0) {
            pn[i] = src[i];
            pn[i + 1] = src[i + 1];
        }
        if (i == 1) *pend_dst = pend;
        if (i == 5) dst[0] = y4[0];
        __builtin_amdgcn_sched_barrier(0);
    }
    if (CHECK) {
        if (!(u1 == u1 && u2 == u2)) {
            u1 = u1_in;
            u2 = u2_in;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float y = (float)((pv[i] - b1 * or0(u1)) - b2 * or0(u2));
                y4[i >> 2][i & 3] = y;
                u2 = or0(u1);
                u1 = (double)y;
            }
            u1 = or0(u1);
            dst[0] = y4[0];
        }
    }
    pend = y4[1];
}

template <int CHECK>
__global__ void __launch_bounds__(1024) k_serial_i(double *out, unsigned long long *cyc, double b1, double b2, int n) {
    __shared__ __attribute__((aligned(16))) double tile[ROWS * PITCH + 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < ROWS * PITCH; i += blockDim.x) tile[i] = 0.001 * (i % 97);
    __syncthreads();
    double u1 = out[lane], u2 = out[lane + 64];
    unsigned long long total = 0;
    for (int it = 0; it < n; ++it) {
        if (wave == 0 && lane < ROWS) {
            const unsigned long long s0 = __builtin_readcyclecounter();
            const uint32_t row = ((uint32_t)(uintptr_t)(lds_double *)(tile + (size_t)lane * PITCH)) & 0x3ffffu;
            const lds_double *pr = (const lds_double *)(uintptr_t)row;
            lds_f32x4 *yr = (lds_f32x4 *)(uintptr_t)row;
            double pa[8], pb[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) pa[i] = pr[i];
            __builtin_amdgcn_s_waitcnt(0xc07f);
            f32x4 pend = {0.f, 0.f, 0.f, 0.f};
            lds_f32x4 *pend_dst = (lds_f32x4 *)(uintptr_t)(row + SUB * 8);  // (the row's two spare doubles: a harmless first store)
            for (int t0 = 0; t0 < SUB; t0 += 16) {
                block8i<CHECK>(pa, pb, pr + 8, b1, b2, u1, u2, yr, pend, pend_dst);
                const int next = t0 + 16 < SUB ? 16 : 0;
                block8i<CHECK>(pb, pa, pr + next, b1, b2, u1, u2, yr + 2, pend, yr + 1);
                pend_dst = yr + 3;
                pr += 16;
                yr += 4;
            }
            *pend_dst = pend;
            total += __builtin_readcyclecounter() - s0;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < ROWS * PITCH; i += blockDim.x) tile[i] = 0.001 * ((i + it) % 97);
        __syncthreads();
    }
    if (wave == 0 && lane < ROWS) out[lane] = u1 + u2;
    if (threadIdx.x == 0) *cyc = total;
}

int main() {
    double *d; unsigned long long *c;
    hipMalloc(&d, 4096); hipMalloc(&c, 8);
    double h[512]; for (int i = 0; i < 512; ++i) h[i] = 0.001 * (i + 1);
    const int n = 2000;
#define RUN(V, WHAT)                                                                          \
    for (int threads : {64, 1024}) {                                                          \
        hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);                                         \
        hipLaunchKernelGGL(k_serial<V>, dim3(1), dim3(threads), 0, 0, d, c, 0.999, -0.5, n);  \
        hipLaunchKernelGGL(k_serial<V>, dim3(1), dim3(threads), 0, 0, d, c, 0.999, -0.5, n);  \
        unsigned long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);                   \
        printf("%-44s threads=%4d  %.2f cycles per step\n", WHAT, threads, (double)cy / ((double)n * SUB)); \
    }
    RUN(0, "ping-pong, NaN check per 8, y written")
    RUN(1, "  no NaN check")
    RUN(2, "  no NaN check, y not written")
    RUN(3, "  P held in registers, nothing written")
#define k_serial k_serial_i
    RUN(1, "interleaved reads / stores, NaN check per 8")
    RUN(0, "interleaved reads / stores, no check")
    return 0;
}
