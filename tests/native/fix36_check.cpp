// CPU check of jit_prelude.hpp jit_fix36 (f32 increment -> 2^-36 fixed point through the f64 mantissa field; restated here in host
// C++ operation for operation) against the definition: the C cast (long long)(f * 2^36), truncated toward zero — every exponent a
// finite |f| < 2^16 can have, edge mantissas and random ones, both signs, zeros and subnormals.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

static long long fix36(float f) {
    const double x = std::trunc(std::ldexp((double)f, 36));
    const double y = std::fabs(x) + 4503599627370496.0;
    unsigned long long bits;
    std::memcpy(&bits, &y, 8);
    const unsigned long long mag = bits - 0x4330000000000000ull;
    return f < 0.f ? -(long long)mag : (long long)mag;
}
static uint64_t rng = 0x2545F4914F6CDD1Dull;
static uint64_t xr() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; }

int main() {
    long cases = 0, bad = 0;
    for (uint32_t e = 0; e < 143; e++) {  // biased exponent 142 = 2^15 .. 2^16
        uint32_t mants[40] = {0u, 1u, 2u, 0x7fffffu, 0x7ffffeu, 0x400000u, 0x400001u, 0x3fffffu};
        for (int k = 8; k < 40; k++) mants[k] = (uint32_t)xr() & 0x7fffffu;
        for (int k = 0; k < 40; k++)
            for (uint32_t sign = 0; sign < 2; sign++) {
                const uint32_t bits = (sign << 31) | (e << 23) | mants[k];
                float f;
                std::memcpy(&f, &bits, 4);
                const long long want = (long long)((double)f * 68719476736.0);
                cases++;
                if (fix36(f) != want) {
                    if (bad < 5) std::fprintf(stderr, "f %a: want %lld got %lld\n", f, want, fix36(f));
                    bad++;
                }
            }
    }
    std::printf("{\"cases\": %ld, \"bad\": %ld}\n", cases, bad);
    return bad ? 1 : 0;
}
