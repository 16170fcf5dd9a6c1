// Standalone write-bandwidth probe: which store pattern reaches the HBM write ceiling on MI355X?
// hipcc --offload-arch=gfx950 -O3 -o fillbench fillbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e_), #x); return 1; } } while (0)

template <int NT>
__global__ void __launch_bounds__(256) fill_gridstride(f32x4 *out, size_t n4, float v) {
    const f32x4 x = {v, v, v, v};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        if (NT) __builtin_nontemporal_store(x, out + i); else out[i] = x;
    }
}
// each WAVE owns a contiguous span of `span4` float4 (= the fused kernel's pattern: 1 KiB per wave-instruction)
template <int NT, int BLOCK>
__global__ void __launch_bounds__(BLOCK) fill_wavespan(f32x4 *out, size_t n4, size_t span4, float v) {
    const f32x4 x = {v, v, v, v};
    const size_t n_spans = n4 / span4;
    const size_t wave = (size_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    const size_t total = (size_t)gridDim.x * (BLOCK / 64);
    const unsigned lane = threadIdx.x & 63;
    for (size_t s = wave; s < n_spans; s += total) {
        f32x4 *p = out + s * span4 + lane;
        for (size_t k = 0; k < span4; k += 64) {
            if (NT) __builtin_nontemporal_store(x, p + k); else p[k] = x;
        }
    }
}
// each BLOCK owns a contiguous span
template <int NT>
__global__ void __launch_bounds__(256) fill_blockspan(f32x4 *out, size_t n4, size_t span4, float v) {
    const f32x4 x = {v, v, v, v};
    const size_t n_spans = n4 / span4;
    for (size_t s = blockIdx.x; s < n_spans; s += gridDim.x) {
        f32x4 *p = out + s * span4 + threadIdx.x;
        for (size_t k = 0; k < span4; k += 256) {
            if (NT) __builtin_nontemporal_store(x, p + k); else p[k] = x;
        }
    }
}

int main() {
    const size_t n_floats = (size_t)1024 * 2880000;
    const size_t n4 = n_floats / 4;
    float *d; CK(hipMalloc(&d, n_floats * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto timeit = [&](const char *name, auto launch) {
        std::vector<float> ts;
        for (int r = 0; r < 6; r++) {
            hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (r) ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        printf("%-40s median %7.3f ms  %7.1f GB/s\n", name, ts[ts.size()/2], n_floats * 4.0 / ts[ts.size()/2] / 1e6);
        fflush(stdout);
    };
    char name[128];
    for (int g : {256, 512, 1024, 2048, 4096, 8192, 16384}) {
        snprintf(name, sizeof name, "gridstride plain grid=%d", g);
        timeit(name, [&] { hipLaunchKernelGGL(fill_gridstride<0>, dim3(g), dim3(256), 0, 0, (f32x4*)d, n4, 1.f); });
        snprintf(name, sizeof name, "gridstride nt    grid=%d", g);
        timeit(name, [&] { hipLaunchKernelGGL(fill_gridstride<1>, dim3(g), dim3(256), 0, 0, (f32x4*)d, n4, 1.f); });
    }
    for (size_t spanKB : {16, 64, 256, 1024}) {
        const size_t span4 = spanKB * 1024 / 16;
        for (int g : {256 * 4, 256 * 8}) {
            snprintf(name, sizeof name, "wavespan plain %4zuKB grid=%d x256", spanKB, g);
            timeit(name, [&] { hipLaunchKernelGGL((fill_wavespan<0, 256>), dim3(g), dim3(256), 0, 0, (f32x4*)d, n4, span4, 1.f); });
            snprintf(name, sizeof name, "wavespan nt    %4zuKB grid=%d x256", spanKB, g);
            timeit(name, [&] { hipLaunchKernelGGL((fill_wavespan<1, 256>), dim3(g), dim3(256), 0, 0, (f32x4*)d, n4, span4, 1.f); });
        }
        snprintf(name, sizeof name, "wavespan nt    %4zuKB grid=256 x1024", spanKB);
        timeit(name, [&] { hipLaunchKernelGGL((fill_wavespan<1, 1024>), dim3(256), dim3(1024), 0, 0, (f32x4*)d, n4, span4, 1.f); });
        snprintf(name, sizeof name, "wavespan plain %4zuKB grid=256 x1024", spanKB);
        timeit(name, [&] { hipLaunchKernelGGL((fill_wavespan<0, 1024>), dim3(256), dim3(1024), 0, 0, (f32x4*)d, n4, span4, 1.f); });
        snprintf(name, sizeof name, "blockspan plain %4zuKB grid=2048", spanKB);
        timeit(name, [&] { hipLaunchKernelGGL(fill_blockspan<0>, dim3(2048), dim3(256), 0, 0, (f32x4*)d, n4, span4, 1.f); });
        snprintf(name, sizeof name, "blockspan nt    %4zuKB grid=2048", spanKB);
        timeit(name, [&] { hipLaunchKernelGGL(fill_blockspan<1>, dim3(2048), dim3(256), 0, 0, (f32x4*)d, n4, span4, 1.f); });
    }
    // hipMemsetAsync for reference
    timeit("hipMemsetD32Async", [&] { hipMemsetD32Async((hipDeviceptr_t)d, 0x3f800000, n_floats, 0); });
    hipFree(d);
    return 0;
}
