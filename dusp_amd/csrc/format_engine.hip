// format_engine.hip — PCM wire format: planar [instance][channel][sample] -> interleaved [instance][sample][channel].
//
// The render kernels write what renderChannelData returns (one array per channel, reference
// src/renderChannelData.js:35-45); streams and files want frames: RenderStream emits
// `buffer[t * numberOfChannels + c]` as 32-bit little-endian floats (reference src/RenderStream.js:28,54,63-68)
// and a WAV data chunk has the same layout.  Pure data movement, HBM-bound: 4 B read + 4 B written per sample.
//
// A workgroup moves a tile of kTile frames of one instance: every channel's kTile samples are read as one coalesced
// run into LDS, then the kTile * C interleaved floats leave as one contiguous run.  The LDS row pitch kTile + 1 keeps
// the transposing reads (stride = pitch) off a single bank.
#include <hip/hip_runtime.h>

#include <cstdint>

namespace dusp {

constexpr int kTile = 256;  // threads per workgroup = frames per tile (128 frames above 32 channels: the tile stays under 64 KB of LDS)

__global__ void __launch_bounds__(kTile) dusp_interleave_kernel(const float *__restrict__ in, float *__restrict__ out, uint32_t n_channels,
                                                                uint64_t n_samples, uint64_t tiles_per_instance, uint32_t tile_frames) {
    extern __shared__ float tile[];  // [n_channels][tile_frames + 1]
    const uint32_t pitch = tile_frames + 1;
    const uint64_t inst = blockIdx.x / tiles_per_instance;
    const uint64_t t0 = (blockIdx.x % tiles_per_instance) * tile_frames;
    const uint32_t frames = (uint32_t)(n_samples - t0 < (uint64_t)tile_frames ? n_samples - t0 : (uint64_t)tile_frames);
    const float *src = in + inst * n_channels * n_samples + t0;
    for (uint32_t j = threadIdx.x; j < n_channels * tile_frames; j += kTile) {  // one coalesced run per channel
        const uint32_t c = j / tile_frames, t = j - c * tile_frames;
        if (t < frames) tile[c * pitch + t] = src[(uint64_t)c * n_samples + t];
    }
    __syncthreads();
    float *dst = out + (inst * n_samples + t0) * n_channels;
    const uint32_t total = frames * n_channels;
    for (uint32_t j = threadIdx.x; j < total; j += kTile) {  // one contiguous run of frames
        const uint32_t t = j / n_channels, c = j - t * n_channels;
        dst[j] = tile[c * pitch + t];
    }
}

hipError_t launch_interleave(const float *d_planar, float *d_out, uint32_t n_instances, uint32_t n_channels, uint64_t n_samples, hipStream_t stream) {
    if (n_channels == 1)  // already frames
        return hipMemcpyAsync(d_out, d_planar, (size_t)n_instances * n_samples * sizeof(float), hipMemcpyDeviceToDevice, stream);
    const uint32_t tile_frames = n_channels <= 32 ? kTile : kTile / 2;
    const uint64_t tiles = (n_samples + tile_frames - 1) / tile_frames;
    hipLaunchKernelGGL(dusp_interleave_kernel, dim3((uint32_t)(tiles * n_instances)), dim3(kTile), n_channels * (tile_frames + 1) * sizeof(float),
                       stream, d_planar, d_out, n_channels, n_samples, tiles, tile_frames);
    return hipGetLastError();
}

}  // namespace dusp
