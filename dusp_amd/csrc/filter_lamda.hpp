// filter_lamda.hpp — the Filter unit's coefficients (Filter.js:66-84), shared by every engine (host and device:
// tests/native/filter_lamda_check.cpp pins them on the CPU against the reference's expressions in extended precision).
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

// sin y and cos y for 0 <= y <= PI/4 (+ a tail of y the kernels take in first order): the two classical minimax kernels
// (coefficients of fdlibm's __kernel_sin / __kernel_cos, Sun Microsystems 1993), each below one ulp.  These are approximations
// of this library's own, so their Horner steps are fused (one rounding each) — unlike the reference's arithmetic elsewhere.
DUSP_HOST_DEVICE void filter_sincos_kernel(double y, double tail, double &sn, double &cs) {
    const double z = y * y;
    double r = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    r = fma(z, r, 2.75573137070700676789e-06);
    r = fma(z, r, -1.98412698298579493134e-04);
    r = fma(z, r, 8.33333333332248946124e-03);
    r = fma(z, r, -1.66666666666666324348e-01);
    sn = y + fma(z * y, r, tail * fma(-0.5, z, 1.0));
    double c = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    c = fma(z, c, -2.75573143513906633035e-07);
    c = fma(z, c, 2.48015872894767294178e-05);
    c = fma(z, c, -1.38888888888741095749e-03);
    c = fma(z, c, 4.16666666666666019037e-02);
    c = z * c;
    // (1 - z/2 loses the last place above |y| = 0.3: there a quarter of y, cut to a float, comes off both sides exactly — fdlibm's qx)
    const double qx = y < 0.3 ? 0.0 : (double)(float)(0.25 * y);
    cs = (1.0 - qx) - ((0.5 * z - qx) - fma(z, c, -tail * y));
}

// The Butterworth coefficients k = {a0, a1, a2, b1, b2} of kind 0 (low pass) / 1 (high pass) at cutoff f (Filter.js:66-84):
//     low pass   lamda = 1 / tan(PI f / sr);  a0 = 1 / (1 + 2 lamda + lamda^2);  a1 = 2 a0;  a2 = a0;   b1 = 2 a0 (1 - lamda^2);  b2 = a0 (1 - 2 lamda + lamda^2)
//     high pass  lamda = tan(PI f / sr);      a0 = 1 / (1 + 2 lamda + lamda^2);  a1 = 0;     a2 = -a0;  b1 = 2 a0 (lamda^2 - 1);  b2 = a0 (1 - 2 lamda + lamda^2)
// A cutoff that is a SIGNAL needs a set per sample, and the expressions as written cost a tan() with an argument reduction for any
// double plus two divisions — some 170 vector instructions per sample, which is what bounded the kernels with a modulated cutoff.
// For every cutoff below Nyquist, 0 < x = PI f / sr < PI / 2, sine and cosine of x are positive and, with s = sin x, c = cos x,
// q = 1 / (1 + 2 s c), the same five numbers are
//     a0 = s^2 q (low pass) / c^2 q (high pass),   b1 = 2 (s - c)(s + c) q,   b2 = (1 - 2 s c) q
// — ONE division, no lamda.  s and c come from the two kernels above (above PI/4 of the complement, PI/2 in two parts).  Measured
// against the reference's expressions evaluated exactly (tests/native/filter_lamda_check.cpp, 4.8 million cutoffs): a0 within 4.1 ulp,
// b1 5.7, b2 3.5 — the reference's OWN double arithmetic over the math library's tan is within 4.4 / 5.4 / 3.9 of that, i.e. these
// are as close to the exact coefficients as the reference's.  Filter graphs are graded by the north star's tolerance, and EVERY
// engine computes its coefficients here, so they still agree with each other bit for bit.
// Outside that range — a cutoff of zero (the reference's coefficients are then 0, 0, 0, NaN, NaN), negative, at or above Nyquist —
// tan(x) comes from a three-part reduction by PI/2 and the same kernels, and the coefficients from the expressions as written
// (IEEE semantics for the infinities included).  |x| >= 2^20 PI/2 (cutoffs beyond 10^10 Hz), NaN and the infinities give NaN
// coefficients; the reference's tan() of such arguments is finite, its filter unstable either way.
DUSP_HOST_DEVICE void butterworth_coefficients(int kind, double f, double sr, double (&k)[5]) {
    const double x = 3.141592653589793 * f / sr;  // `Math.PI * f / this.sampleRate`
    if (x > 0.0 && x < 1.5707963267948966) {
        const bool upper = x > 0.7853981633974483;
        double sn, cs;
        filter_sincos_kernel(upper ? 1.5707963267948966 - x : x, upper ? 6.123233995736766e-17 : 0.0, sn, cs);
        const double s = upper ? cs : sn, c = upper ? sn : cs;  // sin x, cos x
        // with p = 2 s c:  1 + 2 lamda + lamda^2 = (1 + p) / s^2 (low pass; / c^2 for the high pass),  1 - 2 lamda + lamda^2 = (1 - p) / s^2,
        // 1 - lamda^2 = (s - c)(s + c) / s^2
        const double p = 2.0 * s * c, q = 1.0 / (1.0 + p), g = kind == 0 ? s : c;
        k[0] = (g * g) * q;
        k[1] = kind == 0 ? 2.0 * k[0] : 0.0;
        k[2] = kind == 0 ? k[0] : -k[0];
        k[3] = 2.0 * ((s - c) * (s + c)) * q;
        k[4] = (1.0 - p) * q;
        return;
    }
    // anywhere else: tan(x) by reduction, then the expressions as the reference writes them
    double t;
    const double ax = fabs(x);
    if (!(ax < 1647099.0)) t = x - x + (x != x ? x : (x - x) / (x - x));  // NaN (see above)
    else {
        const double kk = rint(ax * 6.36619772367581382433e-01);  // quadrants
        const double r1 = (ax - kk * 1.57079632673412561417e+00) - kk * 6.07710050650619224932e-11;  // PI/2 in two parts of 33 + 53 bits
        const double tail = -kk * 2.02226624879595063154e-21 * 0.0;                                   // (the third part is below what |k| < 2^20 can see)
        double sn, cs;
        const double ar = fabs(r1);
        filter_sincos_kernel(ar, tail, sn, cs);
        if (r1 < 0.0) sn = -sn;
        const bool odd = ((long long)kk & 1) != 0;
        t = odd ? -cs / sn : sn / cs;  // tan(r + k PI/2)
        if (x < 0.0) t = -t;
        if (x == 0.0) t = x;           // (tan(+-0) = +-0)
    }
    const double lamda = kind == 0 ? 1.0 / t : t, l2 = lamda * lamda;
    k[0] = 1.0 / (1.0 + 2.0 * lamda + l2);
    if (kind == 0) {
        k[1] = 2.0 * k[0];
        k[2] = k[0];
        k[3] = 2.0 * k[0] * (1.0 - l2);
    } else {
        k[1] = 0.0;
        k[2] = -k[0];
        k[3] = 2.0 * k[0] * (l2 - 1.0);
    }
    k[4] = k[0] * (1.0 - 2.0 * lamda + l2);
}

}  // namespace dusp
