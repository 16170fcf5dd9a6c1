// filter_lamda.hpp — the one transcendental of the Filter unit, shared by every engine (host and device: tests/native/filter_lamda_check.cpp
// pins it on the CPU against the reference's expression in extended precision).
#pragma once
#if !defined(__HIPCC_RTC__)
#include <cmath>
#endif

#if !defined(DUSP_HOST_DEVICE)
#if defined(__HIPCC__)
#define DUSP_HOST_DEVICE __host__ __device__ __forceinline__
#else
#define DUSP_HOST_DEVICE inline
#endif
#endif

namespace dusp {

// ---- lamda of the Butterworth coefficients (Filter.js:66-84): 1 / tan(PI f / sr) for the low pass, tan(PI f / sr) for the high pass.
// A cutoff that is a SIGNAL needs one per sample, and the math library's tan() carries an argument reduction for any double
// (hundreds of instructions) that a cutoff never needs: inside 0 < PI f / sr < PI / 2 — every cutoff below Nyquist — sine and cosine come
// from the two classical minimax kernels on [0, PI/4] (the coefficients of fdlibm's __kernel_sin / __kernel_cos, Sun Microsystems
// 1993, each below one ulp; above PI/4 the complement, PI/2 in two parts) and lamda is ONE division cos / sin (low pass) or sin / cos
// (high pass): within 2.3 ulp of the reference's expression evaluated exactly (4.8 million cutoffs, tests/native/filter_lamda_check.cpp) — Filter graphs are graded by the north
// star's tolerance, and every engine uses this one function, so they still agree with each other bit for bit.  Anything else
// (zero, negative, at or above Nyquist, NaN) goes the math library's way.
DUSP_HOST_DEVICE double filter_lamda(int kind, double f, double sr) {
    const double x = 3.141592653589793 * f / sr;  // `Math.PI * f / this.sampleRate`
    if (!(x > 0.0 && x < 1.5707963267948966)) return kind == 0 ? 1.0 / tan(x) : tan(x);
    const bool upper = x > 0.7853981633974483;
    // above PI/4 the complement: PI/2 - x is exact in the high part of PI/2, the low part (6.1e-17) is a tail the kernels take in
    // first order — sin(y + t) = sin y + t cos y, cos(y + t) = cos y - t sin y — instead of a rounding of y
    const double y = upper ? 1.5707963267948966 - x : x, tail = upper ? 6.123233995736766e-17 : 0.0;
    const double z = y * y;
    const double sr_ = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    const double sn = y + ((z * y) * (-1.66666666666666324348e-01 + z * sr_) + tail * (1.0 - 0.5 * z));
    const double cr = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    // (1 - z/2 loses the last place above |y| = 0.3: there a quarter of y, cut to a float, comes off both sides exactly — fdlibm's qx)
    const double qx = y < 0.3 ? 0.0 : (double)(float)(0.25 * y);
    const double cs = (1.0 - qx) - ((0.5 * z - qx) - (z * cr - tail * y));
    const double s = upper ? cs : sn, c = upper ? sn : cs;  // sin x, cos x
    return kind == 0 ? c / s : s / c;
}

}  // namespace dusp
