// A/B of the half-table fold (tools/ab_fold.hip): `min(i, N - i)` over a forward image against |i - M| (one v_sad_u32) over the image stored
// backwards from the middle (device_util.hpp Table<1>), on the sum chain's access pattern.  hipcc --offload-arch=gfx950 -O3 -o tools/ab_fold tools/ab_fold.hip
// MI355X: 1.86 ms / 1.74 ms / 2.02 ms (fold by select / v_sad_u32 / max - min as the compiler spells it).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cmath>
__device__ __forceinline__ uint32_t abs_diff(uint32_t a, uint32_t b) { uint32_t r; asm("v_sad_u32 %0, %1, %2, 0" : "=v"(r) : "v"(a), "s"(b)); return r; }
__device__ __forceinline__ uint32_t abs_diff2(uint32_t a, uint32_t b) { return max(a, b) - min(a, b); }
template <int MODE>
__global__ void __launch_bounds__(1024) k(float *out, const float *tab, uint32_t sr, int voices, int groups) {
    extern __shared__ float lds[];
    const uint32_t M = sr / 2, N = sr + 1, last = M + 1, n_words = last + (last >> 5) + 2;
    for (uint32_t q = threadIdx.x; q < n_words; q += 1024) {
        const uint32_t t = (q / 33) * 32 + (q % 33);
        lds[q] = MODE == 0 ? tab[min(t, last)] : tab[last - min(t, last)];
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    float acc[4][4] = {};
    for (int j = 0; j < voices; ++j) {
        const uint32_t f = 10u * (j + 1), s256 = (256u * f) % sr;
        uint32_t idx[4];
        idx[0] = (uint32_t)(((uint64_t)(blockIdx.x * 16 + (threadIdx.x >> 6)) * 1024 * f + (uint64_t)lane * 4 * f) % sr);
        for (int c = 1; c < 4; ++c) { idx[c] = idx[c - 1] + f; idx[c] = min(idx[c], idx[c] - sr); }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t i = idx[c];
                float v;
                if (MODE == 0) { const uint32_t kk = min(i, N - i); const float x = lds[kk + (kk >> 5)]; v = i > M ? -x : x; }
                else if (MODE == 1) { const bool up = i > M; const uint32_t t = abs_diff(i, M) + (up ? 0u : 1u); const float x = lds[t + (t >> 5)]; v = up ? -x : x; }
                else { const bool up = i > M; const uint32_t t = abs_diff2(i, M) + (up ? 0u : 1u); const float x = lds[t + (t >> 5)]; v = up ? -x : x; }
                acc[g][c] += v;
                idx[c] += s256; idx[c] = min(idx[c], idx[c] - sr);
            }
    }
    float *row = out + ((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 1024 + lane * 4;
    for (int g = 0; g < 4; ++g) *(float4 *)(row + g * 256) = make_float4(acc[g][0], acc[g][1], acc[g][2], acc[g][3]);
}
int main() {
    const uint32_t sr = 48000;
    std::vector<float> T(sr + 8);
    for (uint32_t t = 0; t <= sr; t++) T[t] = (float)sin(2.0 * M_PI * t / (sr + 1.0));
    float *d_tab, *d_out;
    hipMalloc(&d_tab, T.size() * 4); hipMemcpy(d_tab, T.data(), T.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&d_out, (size_t)256 * 16 * 1024 * 4 * 2);
    const size_t lds = (24001 + 750 + 2 + 8) * 4;
    auto run = [&](auto kern, const char *name) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        std::vector<float> h(256 * 16 * 1024);
        for (int r = 0; r < 3; r++) {
            hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(256), dim3(1024), lds, 0, d_out, d_tab, sr, 1024, 4); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost);
            double s = 0; for (float v : h) s += v;
            printf("%s %.3f ms  checksum %.6f\n", name, ms, s);
        }
    };
    run(k<0>, "old fold      ");
    run(k<1>, "sad asm       ");
    run(k<2>, "max-min       ");
    return 0;
}
