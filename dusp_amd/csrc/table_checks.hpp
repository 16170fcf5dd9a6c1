// table_checks.hpp — host-side properties of an uploaded wave table that the kernels' fast forms rely on (dusp_table_upload;
// tests/native/lerp_delta_check.cpp).  Plain C++, no device code.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdlib>

namespace dusp {

// Which delta form of the oscillator's lerp a table admits (device_util.hpp lerp_delta; checked over all its neighbours):
// 2 every T[t+1] - T[t] is an f32 of ordinary magnitude or zero, 1 every one is exact in f64, 0 neither (or an entry is not finite).
// (The reference's 48 kHz sine table is a 1: the two pairs next to its middle, sin(1.5 d) and sin(0.5 d), differ by 25 bits.)
inline int table_delta_class(const float *table, size_t n) {
    int cls = 2;
    for (size_t t = 0; t < n; t++)
        if (!(table[t] - table[t] == 0.f)) return 0;
    for (size_t t = 0; t + 1 < n; t++) {
        const double x = (double)table[t], y = (double)table[t + 1], d = y - x;
        // exact in f64: both are multiples of 2^(e_min - 23) and the difference is below 2^(e_max + 1), so it has at most
        // e_max - e_min + 25 bits (ilogb of an f32 subnormal converted to f64 is its true exponent: on the safe side)
        if (x != 0.0 && y != 0.0 && std::abs(std::ilogb(x) - std::ilogb(y)) > 28) return 0;
        const float df = (float)d;
        if (!((double)df == d && (df == 0.f || df >= 1.0e-30f || df <= -1.0e-30f))) cls = 1;
    }
    return cls;
}

}  // namespace dusp
