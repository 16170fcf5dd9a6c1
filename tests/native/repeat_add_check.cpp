// CPU check of dusp_amd/csrc/repeat_add.hpp: n steps of `t = fl(t + c)` at once must equal the plain loop bit for bit —
// increments as Timer (1 / sampleRate) and Shape (1 / f32 duration) produce them, few-bit increments that tie exactly in
// some binade, arbitrary bit patterns; start values from zero, mid-run, huge; up to millions of steps.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../dusp_amd/csrc/repeat_add.hpp"

static double seq(double t, double c, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) t = t + c;
    return t;
}
static uint64_t rng = 88172645463325252ull;
static uint64_t xr() {
    rng ^= rng << 13;
    rng ^= rng >> 7;
    rng ^= rng << 17;
    return rng;
}

int main(int argc, char **argv) {
    const int iterations = argc > 1 ? atoi(argv[1]) : 40000;
    long bad = 0, cases = 0, linear = 0;
    for (int it = 0; it < iterations; it++) {
        double c, t;
        const int kind = it % 8;
        const uint64_t bits = xr();
        if (kind == 0) c = 1.0 / (double)(float)((xr() % 100000 + 1) / 997.0);
        else if (kind == 1) c = 1.0 / 48000.0;
        else if (kind == 2) c = 1.0 / 44100.0;
        else if (kind == 3) c = ldexp((double)(xr() % 7 + 1), (int)(xr() % 30) - 20);
        else if (kind == 4) c = ldexp(1.0 + (double)(xr() >> 12) * ldexp(1.0, -52), (int)(xr() % 60) - 40);
        else if (kind == 5) c = (double)(float)(0.001 + (xr() % 1000000) * 1e-3);
        else if (kind == 6) c = ldexp((double)((xr() % 1024) * 2 + 1), -(int)(xr() % 60));
        else {
            memcpy(&c, &bits, 8);
            c = fabs(c);
            if (!(c > 1e-300 && c < 1e300)) c = 0.37;
        }
        const int tk = xr() % 5;
        if (tk == 0) t = 0;
        else if (tk == 1) t = seq(0, c, xr() % 1000);
        else if (tk == 2) t = (double)(xr() % 100000);
        else if (tk == 3) t = ldexp((double)(xr() >> 11), (int)(xr() % 80) - 60);
        else t = c * (double)(xr() % 4096);
        const uint64_t n = (it % 16 == 0) ? xr() % 2000000 : xr() % 3000;
        const double want = seq(t, c, n), got = dusp::repeat_add(t, c, n);
        cases++;
        // linear_run: when it says the next 256 sums stay in t's binade, they are (T + j ce) 2^(K-52), one by one
        {
            long long T, ce;
            int K;
            if (t > 0 && dusp::linear_run(t, c, 256, T, ce, K)) {
                linear++;
                double x = t;
                for (long long j = 1; j <= 256; j++) {
                    x = x + c;
                    const double y = ldexp((double)(T + j * ce), K - 52);
                    if (memcmp(&x, &y, 8) != 0) {
                        if (bad++ < 10) printf("BAD linear t=%a c=%a j=%lld want=%a got=%a\n", t, c, j, x, y);
                        break;
                    }
                }
            }
        }
        if (memcmp(&want, &got, 8) != 0 && bad++ < 10)
            printf("BAD t=%a c=%a n=%llu want=%a got=%a\n", t, c, (unsigned long long)n, want, got);
    }
    printf("{\"cases\": %ld, \"bad\": %ld, \"linear_runs\": %ld}\n", cases, bad, linear);
    return bad != 0;
}
