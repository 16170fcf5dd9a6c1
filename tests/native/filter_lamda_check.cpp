// butterworth_coefficients (dusp_amd/csrc/filter_lamda.hpp) against the reference's expressions (Filter.js:66-84) evaluated in long double:
// ulp error of every coefficient over cutoffs between 0 and Nyquist (dense sweeps + random), and the way out of that range (zero,
// negative, above Nyquist) against the math library's tan() in the expressions as written.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include "../../dusp_amd/csrc/filter_lamda.hpp"

static double ulps(double got, long double want) {
    if (got == (double)want) return 0.0;
    const double w = (double)want;
    if (w == 0.0) return std::fabs(got) < 1e-300 ? 0.0 : 1e9;
    const double u = std::nextafter(std::fabs(w), INFINITY) - std::fabs(w);
    return (double)(std::fabs((long double)got - want) / u);
}
static void reference(int kind, long double x, long double (&k)[5]) {
    const long double t = tanl(x), lamda = kind == 0 ? 1.0L / t : t, l2 = lamda * lamda;
    k[0] = 1.0L / (1.0L + 2.0L * lamda + l2);
    k[1] = kind == 0 ? 2.0L * k[0] : 0.0L;
    k[2] = kind == 0 ? k[0] : -k[0];
    k[3] = kind == 0 ? 2.0L * k[0] * (1.0L - l2) : 2.0L * k[0] * (l2 - 1.0L);
    k[4] = k[0] * (1.0L - 2.0L * lamda + l2);
}
int main() {
    const double sr = 48000.0;
    double worst[2][5] = {{0}}, rel_out = 0;
    long n = 0, bad = 0;
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> uf(0.0, 24000.0), small(0.0, 1.0), wide(-2.0e6, 2.0e6);
    auto probe = [&](float ff) {
        const double f = (double)ff;
        const double x = 3.141592653589793 * f / sr;
        if (!(x > 0.0 && x < 1.5707963267948966)) return;
        for (int kind = 0; kind < 2; kind++) {
            long double want[5];
            double got[5];
            reference(kind, (long double)x, want);
            dusp::butterworth_coefficients(kind, f, sr, got);
            for (int i = 0; i < 5; i++) {
                // b1 and b2 pass through zero at a quarter of the sample rate: there an ulp of theirs is no measure; below 0.25 their error
                // is taken in ulps of 0.25 instead (what the recurrence feels next to a0-sized numbers)
                const double e = (i >= 3 && std::fabs((double)want[i]) < 0.25) ? (double)(fabsl((long double)got[i] - want[i]) / 5.6e-17L) : ulps(got[i], want[i]);
                worst[kind][i] = std::fmax(worst[kind][i], e);
            }
        }
        n++;
    };
    for (int i = 1; i < 2400000; i++) probe((float)(i * 0.01));
    for (int i = 0; i < 2000000; i++) probe((float)uf(rng));
    for (int i = 0; i < 200000; i++) probe((float)small(rng));
    for (int i = 0; i < 200000; i++) probe((float)(24000.0 - small(rng)));
    // outside the range: the expressions as written over the math library's tan (relative error; NaN / infinities must agree in kind)
    auto outside = [&](double f) {
        const double x = 3.141592653589793 * f / sr;
        if (x > 0.0 && x < 1.5707963267948966) return;
        for (int kind = 0; kind < 2; kind++) {
            double got[5];
            dusp::butterworth_coefficients(kind, f, sr, got);
            const double t = std::tan(x), lamda = kind == 0 ? 1.0 / t : t, l2 = lamda * lamda;
            double want[5];
            want[0] = 1.0 / (1.0 + 2.0 * lamda + l2);
            want[1] = kind == 0 ? 2.0 * want[0] : 0.0;
            want[2] = kind == 0 ? want[0] : -want[0];
            want[3] = kind == 0 ? 2.0 * want[0] * (1.0 - l2) : 2.0 * want[0] * (l2 - 1.0);
            want[4] = want[0] * (1.0 - 2.0 * lamda + l2);
            for (int i = 0; i < 5; i++) {
                if (want[i] != want[i] || got[i] != got[i]) { bad += (want[i] != want[i]) != (got[i] != got[i]); continue; }
                if (std::isinf(want[i]) || std::isinf(got[i])) { bad += want[i] != got[i]; continue; }
                const double scale = std::fmax(std::fabs(want[i]), 1e-300);
                // (near a pole of tan the coefficients swing through many orders of magnitude within an ulp of x: compare where they are tame)
                if (std::fabs(t) > 1e-6 && std::fabs(t) < 1e6) rel_out = std::fmax(rel_out, std::fabs(got[i] - want[i]) / scale);
            }
        }
    };
    for (double f : {0.0, -0.0, -5.0, 24000.0, 30000.0, 47999.0, 48000.0, 96000.5, -123456.75, 1e9}) outside(f);
    for (int i = 0; i < 400000; i++) outside((double)(float)wide(rng));
    {   // beyond the reduction's range and the non-finite cutoffs: NaN coefficients
        for (double f : {1e12, -1e15, (double)NAN, (double)INFINITY, -(double)INFINITY}) {
            double got[5];
            dusp::butterworth_coefficients(0, f, sr, got);
            bad += !(got[0] != got[0] && got[3] != got[3]);
        }
        double z[5];
        dusp::butterworth_coefficients(0, 0.0, sr, z);  // the reference at f = 0: 0, 0, 0, NaN, NaN
        bad += !(z[0] == 0.0 && z[1] == 0.0 && z[2] == 0.0 && z[3] != z[3] && z[4] != z[4]);
    }
    double w = 0;
    for (int kind = 0; kind < 2; kind++)
        for (int i = 0; i < 5; i++) w = std::fmax(w, worst[kind][i]);
    std::printf("{\"cases\": %ld, \"worst_ulp\": %.3f, \"worst_lp\": [%.2f, %.2f, %.2f, %.2f, %.2f], \"worst_hp\": [%.2f, %.2f, %.2f, %.2f, %.2f], \"outside_rel\": %.3g, \"bad\": %ld}\n",
                n, w, worst[0][0], worst[0][1], worst[0][2], worst[0][3], worst[0][4], worst[1][0], worst[1][1], worst[1][2], worst[1][3], worst[1][4], rel_out, bad);
    return 0;
}
