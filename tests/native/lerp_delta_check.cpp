// CPU check of the oscillator lerp's delta form (dusp_amd/csrc/device_util.hpp lerp_delta / Table::pair_delta, restated here
// in host C++ operation for operation) against the reference's expression (src/components/Osc/Osc.js:43-46)
//     out = f32( T[floor p] * (1 - fraction) + T[ceil p] * fraction )          (f64 arithmetic, one rounding per operation)
// for the reference's sine table (waveTables.js:5-8) at several sample rates, read the way the kernels read it — the half image
// with mirrored pairs above the middle — at EVERY index, with fractions on the 2^-28 grid the form is admitted on: edge values and
// random ones.  Also the classification of tables (table_checks.hpp) the kernels' choice rests on.  Build with -ffp-contract=off.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../dusp_amd/csrc/table_checks.hpp"

static uint64_t rng = 0x9E3779B97F4A7C15ull;
static uint64_t xr() {
    rng ^= rng << 13;
    rng ^= rng >> 7;
    rng ^= rng << 17;
    return rng;
}
static bool same(float a, float b) { return std::memcmp(&a, &b, 4) == 0 || (a == 0.f && b == 0.f); }  // (-0 / +0: the copy-out maps both to +0)

int main() {
    long cases = 0, bad = 0;
    int classes[4] = {0, 0, 0, 0};
    const uint32_t rates[4] = {48000u, 44100u, 96000u, 22050u};
    for (int ri = 0; ri < 4; ri++) {
        const uint32_t sr = rates[ri], N = sr + 1, M = sr / 2;
        std::vector<float> T(N);
        for (uint32_t t = 0; t < N; t++) T[t] = (float)std::sin(2.0 * M_PI * (double)t / (double)N);
        const int cls = dusp::table_delta_class(T.data(), N);
        classes[ri] = cls;
        if (cls == 0) { bad++; continue; }
        for (uint32_t i = 0; i < sr; i++) {
            // what Table<1>::pair_delta reads: H[j], H[j + 1] of the half image H = T[0 .. M + 1]
            const uint32_t j = i < N - 1 - i ? i : N - 1 - i;
            const float x = T[j], y = T[j + 1];
            const float af = i > M ? -y : x;
            const double a = (double)af;
            double d;
            if (cls == 2) {
                const float df = y - x;
                d = (double)df;
            } else
                d = (double)y - (double)x;
            uint32_t Fs[12] = {0u, 16u, 0xFFFFFFF0u, 0x80000000u, 0x7FFFFFF0u, 0x80000010u, 0x10000000u, 0xF0000000u};
            for (int k = 8; k < 12; k++) Fs[k] = (uint32_t)xr() & 0xFFFFFFF0u;
            for (int k = 0; k < 12; k++) {
                const uint32_t F = Fs[k];
                const double fraction = (double)F * (1.0 / 4294967296.0);
                const float want = (float)((double)T[i] * (1.0 - fraction) + (double)T[i + 1] * fraction);
                const float got = (float)std::fma(d, (double)F * (1.0 / 4294967296.0), a);
                cases++;
                if (!same(want, got)) {
                    if (bad < 5) std::fprintf(stderr, "sr %u i %u F %08x: want %a got %a\n", sr, i, F, want, got);
                    bad++;
                }
            }
        }
    }
    // tables the delta form is not for: a NaN entry, neighbours 2^40 apart
    {
        std::vector<float> t(64, 0.25f);
        t[7] = NAN;
        if (dusp::table_delta_class(t.data(), t.size()) != 0) bad++;
        t[7] = 1.0f;
        t[8] = 1.0e-13f;
        if (dusp::table_delta_class(t.data(), t.size()) != 0) bad++;
        t[8] = 1.0f + 1.1920929e-07f;  // (a difference of one ulp: an f32)
        if (dusp::table_delta_class(t.data(), t.size()) != 2) bad++;
        t[8] = 3.0f * 0.33333334f;  // 1.0: fine
        t[9] = 0.37500003f;           // 25 bits apart from 1.0
        if (dusp::table_delta_class(t.data(), t.size()) != 1) bad++;
        cases += 4;
    }
    std::printf("{\"cases\": %ld, \"bad\": %ld, \"classes\": [%d, %d, %d, %d]}\n", cases, bad, classes[0], classes[1], classes[2], classes[3]);
    return bad ? 1 : 0;
}
