// The arithmetic of jit_prelude.hpp JitFilterScan, restated on the CPU, against Filter.js:40-46 as written: per chunk of 256 samples the pairs
// (y[t], y[t-1]) in front of every lane's four samples come from the UNROUNDED recurrence started at the chunk's first pair (what the
// scan over the lanes computes), and each lane then runs its four samples with the reference's rounding of every y to f32.  Claim
// (DESIGN.md 6.2c): the result stays within 2^-24 * (sum|h| + 2) * max|y| of the reference's, h the impulse response of 1 / (1 + b1 z^-1 + b2 z^-2)
// (+ 2: the two results' own roundings to f32); jit_filter_scan_ok takes the form where sum|h| <= 30.  Checked here over white noise and over a sine, ten seconds each, for cutoffs
// across the admitted range and at its edges, both kinds.  Then the same arithmetic inside configs[3]'s feedback loop, where a deviation comes back
// amplified by up to 1 / (1 - g): the bound jit_filter_scan_ok (2) works with, at gains 0.5 .. 0.99 (second half of main).
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
    // ---- the same arithmetic INSIDE a feedback loop: BASELINE configs[3]'s circuit, Osc -> Sum -> Delay(480) -> Filter -> x g -> (a chunk late)
    // Sum.  A deviation the scan injects comes back through the loop: the bound of jit_codegen.hpp jit_filter_scan_ok (2) is
    // eps / (1 - |g| sum|w|), eps = 2^-24 (sum|h| + 2), w the WHOLE Filter's impulse response (an integer delay and the Sum pass a deviation on
    // as it is).  Inputs on a resonance of the loop (latency 480 + 256 = 736 samples: f = m sr / 736) and off it, ten seconds each.
    int loop_cases = 0, loop_bad = 0;
    double loop_worst_ratio = 0.0, loop_worst_of_scale = 0.0, loop_worst_gain99 = 0.0;
    for (double f : {2000.0, 8000.0})
        for (double g : {0.5, 0.9, 0.95, 0.99})
            for (double fosc : {2.0 * sr / 736.0, 5.0 * sr / 736.0, 110.0}) {
                double k[5];
                butterworth_coefficients(0, f, sr, k);
                double h1 = 1.0, h2 = 0.0, sum_h = 1.0, w1 = k[0], w2 = 0.0, sum_w = std::fabs(k[0]);
                for (int t = 1; t < 100000; t++) {
                    const double h = -k[3] * h1 - k[4] * h2, w = (t == 1 ? k[1] : t == 2 ? k[2] : 0.0) - k[3] * w1 - k[4] * w2;
                    h2 = h1, h1 = h, w2 = w1, w1 = w, sum_h += std::fabs(h), sum_w += std::fabs(w);
                }
                std::vector<float> out[2];
                for (int form = 0; form < 2; form++) {  // 0: as written; 1: the scan
                    std::vector<float> &y = out[form];
                    y.assign((size_t)n, 0.f);
                    std::vector<float> line((size_t)n, 0.f);  // the Sum's output (what the Delay holds)
                    double x1 = 0, x2 = 0, s1 = 0, s2 = 0;
                    for (int t0 = 0; t0 < n; t0 += 256) {
                        double p[256], u[258];
                        for (int t = 0; t < 256; t++) {
                            const int a = t0 + t;
                            const float osc = (float)std::sin(2.0 * 3.141592653589793 * fosc * a / sr);
                            const float fb = a >= 256 ? (float)((double)y[(size_t)(a - 256)] * g) : 0.f;  // (Multiply: one f32 rounding; read a chunk late)
                            line[(size_t)a] = (float)((double)osc + (double)fb);
                            const double xin = a >= 480 ? (double)line[(size_t)(a - 480)] : 0.0;
                            p[t] = form ? std::fma(k[2], x2, std::fma(k[1], x1, k[0] * xin)) : (k[0] * xin + k[1] * x1) + k[2] * x2;
                            x2 = x1, x1 = xin;
                        }
                        if (!form) {
                            for (int t = 0; t < 256; t++) {
                                const float v = (float)((p[t] - k[3] * s1) - k[4] * s2);
                                y[(size_t)(t0 + t)] = v;
                                s2 = s1, s1 = (double)v;
                            }
                            continue;
                        }
                        u[0] = s2, u[1] = s1;
                        for (int t = 0; t < 256; t++) u[t + 2] = std::fma(-k[4], u[t], std::fma(-k[3], u[t + 1], p[t]));
                        for (int lane = 0; lane < 64; lane++) {
                            double e1 = u[4 * lane + 1], e2 = u[4 * lane];
                            for (int c = 0; c < 4; c++) {
                                const float v = (float)std::fma(-k[4], e2, std::fma(-k[3], e1, p[4 * lane + c]));
                                y[(size_t)(t0 + 4 * lane + c)] = v;
                                e2 = e1, e1 = (double)v;
                                if (lane == 63) s1 = e1, s2 = e2;
                            }
                        }
                    }
                }
                double scale = 0.0, err = 0.0;
                for (int t = 0; t < n; t++) {
                    scale = std::fmax(scale, std::fabs((double)out[0][(size_t)t]));
                    err = std::fmax(err, std::fabs((double)out[1][(size_t)t] - (double)out[0][(size_t)t]));
                }
                // (+ what the loop's own f32 roundings — the product, the sum — make of a deviated input: an ulp of the Sum's output each,
                // of the order of the Filter's scale here, passed through the Filter once more)
                const double bound = (std::ldexp(sum_h + 2.0, -24) + 2.0 * std::ldexp(sum_w, -24)) / (1.0 - g * sum_w) * scale;
                loop_cases++;
                loop_worst_ratio = std::fmax(loop_worst_ratio, err / bound);
                loop_worst_of_scale = std::fmax(loop_worst_of_scale, err / scale);
                if (g == 0.99) loop_worst_gain99 = std::fmax(loop_worst_gain99, err / scale);
                if (!(err <= bound) || !(err <= 1e-5 * scale)) {
                    loop_bad++;
                    std::printf("FAIL loop f %.1f g %.2f osc %.3f: err %.3g, bound %.3g (scale %.3g)\n", f, g, fosc, err, bound, scale);
                }
            }
    std::printf("{\"cases\": %d, \"bad\": %d, \"worst_of_bound\": %.3f, \"worst_of_scale\": %.3g, \"loop_cases\": %d, \"loop_bad\": %d, "
                "\"loop_worst_of_bound\": %.3f, \"loop_worst_of_scale\": %.3g, \"loop_worst_of_scale_at_gain_0.99\": %.3g}\n",
                cases, bad, worst_ratio, worst_of_scale, loop_cases, loop_bad, loop_worst_ratio, loop_worst_of_scale, loop_worst_gain99);
    return bad || loop_bad ? 1 : 0;
}
