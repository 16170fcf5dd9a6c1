// filter_lamda (dusp_amd/csrc/device_util.hpp) against the reference's expression 1 / tan(PI f / sr) and tan(PI f / sr) in long double:
// ulp error over cutoffs between 0 and Nyquist (dense sweeps + random), and the fallback outside that range.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include "../../dusp_amd/csrc/filter_lamda.hpp"
static double ulps(double got, long double want) {
    if (got == (double)want) return 0.0;
    const double w = (double)want;
    const double u = std::nextafter(std::fabs(w), INFINITY) - std::fabs(w);
    return (double)(std::fabs((long double)got - want) / u);
}
int main() {
    const double sr = 48000.0;
    double worst[2] = {0, 0};
    long n = 0;
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> uf(0.0, 24000.0), small(0.0, 1.0);
    auto probe = [&](float ff) {
        const double f = (double)ff;
        const double x = 3.141592653589793 * f / sr;
        if (!(x > 0.0 && x < 1.5707963267948966)) return;
        const long double t = tanl((long double)x);
        worst[0] = std::fmax(worst[0], ulps(dusp::filter_lamda(0, f, sr), 1.0L / t));
        worst[1] = std::fmax(worst[1], ulps(dusp::filter_lamda(1, f, sr), t));
        n++;
    };
    for (int i = 1; i < 2400000; i++) probe((float)(i * 0.01));
    for (int i = 0; i < 2000000; i++) probe((float)uf(rng));
    for (int i = 0; i < 200000; i++) probe((float)small(rng));
    for (int i = 0; i < 200000; i++) probe((float)(24000.0 - small(rng)));
    // outside the range: the math library's values, bit for bit
    long bad = 0;
    for (double f : {0.0, -5.0, 24000.0, 30000.0, 1e9, (double)NAN, (double)INFINITY}) {
        const double x = 3.141592653589793 * f / sr;
        const double a = dusp::filter_lamda(0, f, sr), b = 1.0 / std::tan(x), c = dusp::filter_lamda(1, f, sr), d = std::tan(x);
        bad += std::memcmp(&a, &b, 8) != 0 && !(a != a && b != b);
        bad += std::memcmp(&c, &d, 8) != 0 && !(c != c && d != d);
    }
    std::printf("{\"cases\": %ld, \"worst_ulp_lp\": %.3f, \"worst_ulp_hp\": %.3f, \"bad_fallback\": %ld}\n", n, worst[0], worst[1], bad);
    return 0;
}
