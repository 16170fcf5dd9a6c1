// repeat_add.hpp — the running sum `t += c` of the reference's Timer and Shape, n steps at once and bit for bit.
// Host and device: the device kernels use it (wave_engine.hip), the host planner decides with it whether a render can be
// split in time, and tests/native/repeat_add_check.cpp pins it against the plain loop on the CPU.
#pragma once
#if !defined(__HIPCC_RTC__)
#include <cmath>
#include <cstdint>
#endif

#if defined(__HIPCC__)
#define DUSP_HOST_DEVICE __host__ __device__ __forceinline__
#else
#define DUSP_HOST_DEVICE inline
#endif

namespace dusp {

// t after n repetitions of `t = fl(t + c)` (f64, round to nearest even), for c > 0 finite and t >= 0: the running sums of
// Timer (`t += samplePeriod`, Timer.js:38) and of a Shape with a constant duration (`t += 1 / duration`, Shape/index.js:31)
// in closed form.  While the sums stay inside one binade [2^K, 2^(K+1)) every addend is a multiple of u = 2^(K-52) and the
// rounding adds the same whole number of u each time: c/u rounded to nearest — on a tie (c/u = q + 1/2) to the neighbour
// that keeps t/u even, which after one step it always is.  So: one real addition to enter a binade, then all the steps
// that stay in it at once, in integers.  A render passes through a few dozen binades at most.
DUSP_HOST_DEVICE double repeat_add(double t, double c, uint64_t n) {
    while (n) {
        t = t + c;  // a real step: enters (or stays in) the binade of the result
        --n;
        if (t > 1.7e308) return t;                   // Infinity stays
        if (n == 0) break;
        const int K = ilogb(t);
        if (K < -900) continue;                      // (denormal territory: step by step)
        const double inv_u = ldexp(1.0, 52 - K);     // 1 / u, exact
        const double Cs = c * inv_u;                 // c in units of u (exact: a power-of-two scaling; c <= t, so Cs < 2^53)
        if (!(Cs < 9.0e15)) continue;
        const double qd = floor(Cs), fr = Cs - qd;   // exact
        const long long q = (long long)qd;
        long long T = (long long)(t * inv_u);        // t / u: an integer in [2^52, 2^53)
        long long ce;
        if (fr > 0.5) ce = q + 1;
        else if (fr < 0.5) ce = q;
        else {
            if (T & 1) continue;                     // one more real step makes it even
            ce = q + (q & 1);
        }
        if (ce == 0) return t;                       // c vanishes against t: the sum has stopped moving
        const long long limit = (1ll << 53) - q - 2; // sources up to here keep sum and result inside the binade
        if (T > limit) continue;
        uint64_t m = (uint64_t)((limit - T) / ce) + 1;
        if (m > n) m = n;
        T += (long long)m * ce;
        t = ldexp((double)T, K - 52);
        n -= m;
    }
    return t;
}


// The next `steps` values of t = fl(t + c) as integers, when they all stay in t's binade: t_j = (T + j ce) 2^(K-52)
// (the reasoning of repeat_add.hpp; t is itself such a sum, so it is a multiple of the binade's unit).
DUSP_HOST_DEVICE bool linear_run(double t, double c, long long steps, long long &T, long long &ce, int &K) {
    if (!(t > 0.0 && t < 1.0e300)) return false;
    K = ilogb(t);
    if (K < -900) return false;
    const double inv_u = ldexp(1.0, 52 - K);
    const double Cs = c * inv_u;
    if (!(Cs < 9.0e15)) return false;
    const double qd = floor(Cs), fr = Cs - qd;
    const long long q = (long long)qd;
    T = (long long)(t * inv_u);
    if (fr > 0.5) ce = q + 1;
    else if (fr < 0.5) ce = q;
    else {
        if (T & 1) return false;
        ce = q + (q & 1);
    }
    return T + (steps - 1) * ce <= (1ll << 53) - q - 2;  // the last source still keeps sum and result inside the binade
}

}  // namespace dusp
