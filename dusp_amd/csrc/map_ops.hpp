// map_ops.hpp — the stateless elementwise units as one device function.
// JS computes each of them in f64 and rounds once when storing into the Float32Array chunk.
#pragma once
#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>
#endif

#include "device_types.hpp"

namespace dusp {
namespace {

// Math.pow: C pow except pow(+-1, +-Inf) and pow(x, NaN) are NaN (ECMA-262 Number::exponentiate)
__device__ __forceinline__ double js_pow(double x, double y) {
    if (y != y) return __builtin_nan("");
    if ((x == 1.0 || x == -1.0) && isinf(y)) return __builtin_nan("");
    return pow(x, y);
}

__device__ __forceinline__ float map_apply(int op, float x, float y, double d0) {
    switch (op) {
    case OP_SUBTRACT: return x - y;                                           // Subtract.js:23
    case OP_DIVIDE: return (float)((double)x / (double)y);                    // Divide.js:21
    case OP_POW: return (float)js_pow((double)x, (double)y);                  // Pow.js:27
    case OP_POLARITY_INVERT: return -x;                                       // PolarityInvert.js:15
    case OP_ABS: return fabsf(x);                                             // Abs.js:19
    case OP_DECIBEL_TO_SCALER: return (float)js_pow(10.0, (double)x / 20.0);  // DecibelToScaler.js:16
    case OP_SEMITONE_TO_RATIO: return (float)js_pow(2.0, (double)x / 12.0);   // SemitoneToRatio.js:16
    case OP_SECONDS_TO_SAMPLES:                                               // SecondsToSamples.js:19
    case OP_FIXED_MULTIPLY: return (float)((double)x * d0);                   // FixedMultiply.js:20
    case OP_CLIP: return fabsf(x) > fabsf(y) ? y : x;                         // Clip.js:19-21
    case OP_HARD_CLIP_ABOVE: return x > y ? y : x;                            // HardClipAbove.js:18-22
    case OP_HARD_CLIP_BELOW: return x < y ? y : x;                            // HardClipBelow.js:18-22
    case OP_GAIN: return (float)(js_pow(10.0, (double)y / 20.0) * (double)x); // Gain.js:21,24-26
    }
    return x;
}

// Stateless maps with more operands or an output-channel attribute: Pan, MidiToFrequency, Rescale, CrossFader,
// VectorMagnitude.  v[k] = operand k of this sample (unused operands are whatever the caller loaded).
// Pan.js:19-29 in two steps: the centre compensation is a pow() of the pan position alone, so a kernel generated for a circuit
// whose pan is not a signal takes it once per instance instead of once per sample and channel.
__device__ __forceinline__ double map_pan_compensation(float pan, double d0) { return js_pow(10.0, ((1.0 - fabs((double)pan)) * d0) / 20.0); }
__device__ __forceinline__ float map_pan(float in, float pan, int attr, double compensation) {
    return (float)((double)in * (attr ? 1.0 + (double)pan : 1.0 - (double)pan) / 2.0 * compensation);
}
__device__ __forceinline__ float map_wide(int op, int attr, int n_in, const float (&v)[kMaxIn], double d0) {
    switch (op) {
    case OP_PAN: return map_pan(v[0], v[1], attr, map_pan_compensation(v[1], d0));  // Pan.js:19-29
    case OP_MIDI_TO_FREQUENCY: return (float)(js_pow(2.0, ((double)v[0] - 69.0) / 12.0) * 440.0);  // MidiToFrequency.js:20
    case OP_RESCALE:                                                           // Rescale.js:34-35
        return (float)(((double)v[0] - (double)v[1]) / ((double)v[2] - (double)v[1]) * ((double)v[4] - (double)v[3]) + (double)v[3]);
    case OP_CROSS_FADER:                                                       // CrossFader.js:28
        return (float)((1.0 - (double)v[2]) * (double)v[0] + (double)v[2] * (double)v[1]);
    case OP_VECTOR_MAGNITUDE: {                                                // VectorMagnitude.js:20-26
        double square_sum = 0.0;
#pragma unroll
        for (int c = 0; c < kMaxIn; ++c)
            if (c < n_in) square_sum += (double)v[c] * (double)v[c];
        return (float)sqrt(square_sum);
    }
    }
    return v[0];
}

}  // namespace
}  // namespace dusp
