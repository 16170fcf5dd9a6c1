// The arithmetic of jit_prelude.hpp JitFilterScan, restated on the CPU, against Filter.js:40-46 as written: per chunk of 256 samples the pairs
// (y[t], y[t-1]) in front of every lane's four samples come from the UNROUNDED recurrence started at the chunk's first pair (what the
// scan over the lanes computes), and each lane then runs its four samples with the reference's rounding of every y to f32.  Claim
// (DESIGN.md 6.2c): the result stays within 2^-24 * (sum|h| + 2) * max|y| of the reference's, h the impulse response of 1 / (1 + b1 z^-1 + b2 z^-2)
// (+ 2: the two results' own roundings to f32); jit_filter_scan_ok takes the form where sum|h| <= 30.  Checked here over white noise and over a sine, ten seconds each, for cutoffs
// across the admitted range and at its edges, both kinds.
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "../../dusp_amd/csrc/filter_lamda.hpp"

using namespace dusp;

int main() {
    const double sr = 48000.0;
    const int n = 480000;
    int bad = 0, cases = 0;
    double worst_ratio = 0.0, worst_of_scale = 0.0;
    std::mt19937_64 rng(7);
    std::vector<float> noise((size_t)n), sine((size_t)n);
    for (int t = 0; t < n; t++) {
        noise[(size_t)t] = (float)((double)(rng() >> 11) / 9007199254740992.0 * 2.0 - 1.0);
        sine[(size_t)t] = (float)std::sin(2.0 * 3.141592653589793 * 440.0 * t / sr);
    }
    for (int kind = 0; kind < 2; kind++)
        for (double f : {1500.0, 1560.0, 1600.0, 2000.0, 3000.5, 8000.0, 12000.0, 20000.0, 22000.0, 22400.0}) {
            double k[5];
            butterworth_coefficients(kind, f, sr, k);
            double h1 = 1.0, h2 = 0.0, sum = 1.0;
            for (int t = 0; t < 100000; t++) {
                const double h = -k[3] * h1 - k[4] * h2;
                h2 = h1, h1 = h, sum += std::fabs(h);
            }
            if (!(sum <= 30.0)) continue;  // (not a cutoff the generator admits)
            for (const std::vector<float> *x : {&noise, &sine}) {
                // the reference: every y rounded to f32 (Filter.js:40-46)
                std::vector<float> ref((size_t)n), got((size_t)n);
                {
                    double x1 = 0, x2 = 0, y1 = 0, y2 = 0;
                    for (int t = 0; t < n; t++) {
                        const double xin = (double)(*x)[(size_t)t];
                        const float y = (float)((((k[0] * xin + k[1] * x1) + k[2] * x2) - k[3] * y1) - k[4] * y2);
                        ref[(size_t)t] = y;
                        x2 = x1, x1 = xin, y2 = y1, y1 = (double)y;
                    }
                }
                // the scan form
                {
                    double x1 = 0, x2 = 0, s1 = 0, s2 = 0;  // the chunk's first pair: f32 values (lane 63's of the chunk before)
                    for (int t0 = 0; t0 < n; t0 += 256) {
                        double p[256], u[258];
                        for (int t = 0; t < 256 && t0 + t < n; t++) {
                            const double xin = (double)(*x)[(size_t)(t0 + t)];
                            p[t] = std::fma(k[2], x2, std::fma(k[1], x1, k[0] * xin));
                            x2 = x1, x1 = xin;
                        }
                        u[0] = s2, u[1] = s1;  // u[t + 2]: the unrounded recurrence from the chunk's first pair on
                        for (int t = 0; t < 256; t++) u[t + 2] = std::fma(-k[4], u[t], std::fma(-k[3], u[t + 1], p[t]));
                        for (int lane = 0; lane < 64; lane++) {
                            double e1 = u[4 * lane + 1], e2 = u[4 * lane];  // the pair in front of the lane's samples
                            for (int c = 0; c < 4 && t0 + 4 * lane + c < n; c++) {
                                const float y = (float)std::fma(-k[4], e2, std::fma(-k[3], e1, p[4 * lane + c]));
                                got[(size_t)(t0 + 4 * lane + c)] = y;
                                e2 = e1, e1 = (double)y;
                                if (lane == 63) s1 = e1, s2 = e2;
                            }
                        }
                    }
                }
                double scale = 0.0, err = 0.0;
                for (int t = 0; t < n; t++) {
                    scale = std::fmax(scale, std::fabs((double)ref[(size_t)t]));
                    err = std::fmax(err, std::fabs((double)got[(size_t)t] - (double)ref[(size_t)t]));
                }
                const double bound = std::ldexp(sum + 2.0, -24) * scale;
                cases++;
                worst_ratio = std::fmax(worst_ratio, err / bound);
                worst_of_scale = std::fmax(worst_of_scale, err / scale);
                if (!(err <= bound) || !(err <= 1.9e-6 * scale)) {
                    bad++;
                    std::printf("FAIL kind %d f %.1f: err %.3g, bound %.3g (sum|h| %.2f, scale %.3g)\n", kind, f, err, bound, sum, scale);
                }
            }
        }
    std::printf("{\"cases\": %d, \"bad\": %d, \"worst_of_bound\": %.3f, \"worst_of_scale\": %.3g}\n", cases, bad, worst_ratio, worst_of_scale);
    return bad ? 1 : 0;
}
