// device_util.hpp — device helpers shared by every engine: exact modular arithmetic on fixed-point phases, the wave-table
// image in LDS, PCM stores.  Plain device code with no host dependencies: hipcc compiles it into the library's kernels, and
// hiprtc compiles it again at run time into the per-circuit kernels (jit_engine.hip embeds this file's text).
#pragma once
#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>
#endif

#include "device_types.hpp"
#include "filter_lamda.hpp"

namespace dusp {
namespace {

__device__ __forceinline__ int lsb_exponent(double x) {  // x finite, != 0: exponent of its lowest set bit
    int ex;
    const double fr = frexp(fabs(x), &ex);
    const long long m = (long long)ldexp(fr, 53);
    return ex - 53 + __builtin_ctzll((unsigned long long)m);
}
__device__ __forceinline__ uint64_t addmod(uint64_t a, uint64_t b, uint64_t S) {  // a, b < S < 2^63
    const uint64_t s = a + b;
    return s >= S ? s - S : s;
}
__device__ __forceinline__ uint64_t mulmod(uint64_t a, uint64_t n, uint64_t S) {  // a < S
    uint64_t acc = 0;
    for (int bit = 63 - __builtin_clzll(n | 1); bit >= 0; --bit) {
        acc = addmod(acc, acc, S);
        if ((n >> bit) & 1) acc = addmod(acc, a, S);
    }
    return acc;
}

__device__ __forceinline__ float operand_value(const DevOperand &o, const float *params, uint32_t n_inst, uint32_t inst) {
    return o.kind == SRC_PARAM ? params[(size_t)o.idx * n_inst + inst] : o.cval;
}

// ---- Wave tables that need no table.  The reference's saw, square and triangle (src/components/Osc/waveTables.js:10-26) are
// functions of the index, and "8bit" (:28-31) is a function of the sine table's entry at the same index.  When an uploaded
// table IS such a function — the host checks all sr + 1 entries bit for bit at dusp_table_upload, with this very code — the
// kernels evaluate it instead of gathering from a 192 KB table that does not fit LDS:
//   saw       T[t] = f32(-1 + (t * 2) / N), t < sr; T[sr] = 0                              (N = sr + 1 = the table's length)
//   square    T[t] = 1 for t < sr/2, else -1
//   triangle  first quarter q = sr/4: T[t] = f32((t / sr) * 4); T[t+q] = f32(1 - T[t]); T[t+2q] = -T[t]; T[t+3q] = f32(-1 + T[t]); T[sr] = 0
//   8bit      T[t] = f32(Math.round(sin[t] * 128) / 128), sin = the f32 sine table (JS Math.round: half up, -0 for [-0.5, -0])
// The two divisions use the 2-FMA refined reciprocal; the host's check of every entry covers them.
enum : int { TABLE_FORM_DATA = 0, TABLE_FORM_SAW = 1, TABLE_FORM_SQUARE = 2, TABLE_FORM_TRIANGLE = 3, TABLE_FORM_8BIT = 4 };

#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
#define DUSP_HD __host__ __device__ __forceinline__
#else
#define DUSP_HD inline
#endif

struct TableForm {
    int32_t form;
    uint32_t sr, half, quarter;
    double N, rcp_N, srd, rcp_sr;
};

DUSP_HD TableForm make_table_form(int form, uint32_t sample_rate) {
    TableForm F;
    F.form = form;
    F.sr = sample_rate;
    F.half = sample_rate / 2;
    F.quarter = sample_rate / 4;
    F.N = (double)sample_rate + 1.0;
    F.rcp_N = 1.0 / F.N;
    F.srd = (double)sample_rate;
    F.rcp_sr = 1.0 / F.srd;
    return F;
}

DUSP_HD double refined_quotient(double x, double d, double rcp) {  // x / d wherever the host has checked it (see above)
    const double q = x * rcp;
    return fma(fma(-q, d, x), rcp, q);
}

// JS Math.round of a value whose half-way sum is exact (|x| <= 2^23 here): floor(x + 0.5), with -0 for x in [-0.5, -0]
DUSP_HD double js_round_small(double x) {
    const double r = floor(x + 0.5);
    return (r == 0.0 && (x < 0.0 || (x == 0.0 && 1.0 / x < 0.0))) ? -0.0 : r;
}

// entry i (0 <= i <= sr) of a saw / square / triangle table
DUSP_HD float closed_table_entry(const TableForm &F, uint32_t i) {
    if (F.form == TABLE_FORM_SQUARE) return i < F.half ? 1.f : -1.f;
    if (i >= F.sr) return 0.f;
    if (F.form == TABLE_FORM_SAW) return (float)(-1.0 + refined_quotient((double)(2u * i), F.N, F.rcp_N));
    const uint32_t seg = (i >= F.quarter ? 1u : 0u) + (i >= 2u * F.quarter ? 1u : 0u) + (i >= 3u * F.quarter ? 1u : 0u);
    const float base = (float)(refined_quotient((double)(i - seg * F.quarter), F.srd, F.rcp_sr) * 4.0);
    return seg == 0 ? base : seg == 1 ? (float)(1.0 - (double)base) : seg == 2 ? -base : (float)(-1.0 + (double)base);
}
DUSP_HD float eightbit_of_sine(float sine) { return (float)(js_round_small((double)sine * 128.0) / 128.0); }

// Table access.  TBL == 0: padded full table in global memory (served by L2).
// TBL == 1: half table in LDS (M = sr/2, N = sr+1 = 2M+1: T[i] = -T[N-i]), stored BACKWARDS from the middle:
//   E[t] = T[M+1-t], t = 0 .. M+1.  With a = |i - M| (one v_sad_u32 — no select between i and its mirror image):
//       i <= M:  T[i] =  E[a+1],  T[i+1] =  E[a]
//       i >  M:  T[i] = -E[a],    T[i+1] = -E[a+1]
//   so the lerp's pair is the two adjacent words E[a], E[a+1] either way, and T[i+1] - T[i] = E[a] - E[a+1] on both sides.
//   (i == M may be read either way: T[M+1] = -T[M].)
//   LDS image: blocks of 33 words, block b = E[32b .. 32b+32] (the 33rd word repeats the next
//   block's first), so word(k) = k + (k >> 5).  The odd pitch spreads the arithmetic progressions a
//   wave reads (lane l looks up phase0 + 4 l f) over the 32 banks — a linear image measured 9-way
//   conflicts on average over the 1024-voice sweep, this one 2.8 — and word(k)+1 always holds E[k+1].
// TBL == 2: no table at all — saw / square / triangle in closed form (`form`).
// TBL == 3: "8bit", evaluated from the SINE half-table image in LDS.
__device__ __forceinline__ uint32_t abs_diff(uint32_t a, uint32_t b) {  // |a - b| (the compiler spells max - min in three instructions)
    uint32_t r;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(r) : "v"(a), "s"(b));  // (b: wave-uniform wherever this is used)
    return r;
}

template <int TBL>
struct Table {
    const float *g;
    const float *h;
    uint32_t N, M;
    TableForm form;
    __device__ __forceinline__ const float *word(uint32_t k) const {
        return (const float *)((const char *)h + ((k + (k >> 5)) << 2));  // v_lshrrev + v_add_lshl
    }
    __device__ __forceinline__ float at(uint32_t i) const {
        if (TBL == 0) return g[i];
        if (TBL == 2) return closed_table_entry(form, i);
        const bool upper = i > M;
        const float v = *word(abs_diff(i, M) + (upper ? 0u : 1u));
        const float s = upper ? -v : v;
        return TBL == 3 ? eightbit_of_sine(s) : s;
    }
    __device__ __forceinline__ void pair(uint32_t i, float &a, float &b) const {  // (T[i], T[i+1]), i < N - 1
        if (TBL == 0) {
            a = g[i];
            b = g[i + 1];
            return;
        }
        if (TBL == 2) {
            a = closed_table_entry(form, i);
            b = closed_table_entry(form, i + 1 <= form.sr ? i + 1 : form.sr);  // (the pad entry repeats the last one)
            return;
        }
        if (TBL == 3) {
            a = at(i);
            b = at(i + 1 < N ? i + 1 : N - 1);
            return;
        }
        const bool upper = i > M;
        const float *p = word(abs_diff(i, M));
        const float p0 = p[0], p1 = p[1];
        a = upper ? -p0 : p1;
        b = upper ? -p1 : p0;
    }
    // T[i] and T[i+1] - T[i] for the lerp's delta form (lerp_delta below), given where the image holds them: a = |i - M| and
    // which side of the middle i is on.  D32: every difference of neighbours in the table is an f32 (one v_sub_f32); else it is
    // taken in f64, where it is exact for any table whose neighbours are within 2^28 of each other (dusp_table_upload checks
    // both: table_checks.hpp table_delta_class).
    template <bool D32>
    __device__ __forceinline__ void pair_delta_at(uint32_t a_img, bool upper, double &a, double &d) const {
        const float *p = word(a_img);
        const float p0 = p[0], p1 = p[1];
        a = (double)(upper ? -p0 : p1);
        delta<D32>(p0, p1, d);
    }
    template <bool D32>
    __device__ __forceinline__ void pair_delta(uint32_t i, double &a, double &d) const {  // i < N - 1
        if (TBL == 0) {
            const float x = g[i], y = g[i + 1];
            a = (double)x;
            delta<D32>(y, x, d);
        } else
            pair_delta_at<D32>(abs_diff(i, M), i > M, a, d);
    }
    template <bool D32>
    static __device__ __forceinline__ void delta(float y, float x, double &d) {  // y - x
        if (D32) {
            float df;  // (one v_sub_f32 the vectorizer cannot see: it would pair two of them up behind four register moves)
            asm("v_sub_f32 %0, %1, %2" : "=v"(df) : "v"(y), "v"(x));
            d = (double)df;
        } else
            d = (double)y - (double)x;
    }
};

// The oscillator's lerp (Osc.js:43-46) `T[i] (1 - fraction) + T[i+1] fraction` in its DELTA form, where it is the same arithmetic:
// with the fraction on a grid of 2^-28 or coarser both products are exact in f64 (a 24-bit entry times a weight of at most 29
// bits), so the reference's three roundings are ONE rounding of the exact sum a wa + b wb = a + (b - a) fraction — and with
// b - a exact that is fma(b - a, fraction, a).  F = the fraction in units of 2^-32.
__device__ __forceinline__ float lerp_delta(double a, double d, uint32_t F) {
    return (float)fma(d, (double)F * (1.0 / 4294967296.0), a);
}

// A phase on that grid as ONE 64-bit integer that is also a double: index.fraction in 32.32 fixed point plus 0x41300000 in the
// high word are the bits of 2^20 + phase (the double's last place is 2^-32 there).  Integer adds step it exactly, v_fract_f64
// of it IS the fraction (no conversion, no scaling), and the biased high word goes straight into the half image's address:
// with s = |h - (bias + sr)| (phases below 2 sr: one subtraction of sr pending or not) the image index is a = |s - M|, and the
// sample lies above the table's middle iff (s > M) != (h < bias + sr).
constexpr uint32_t kPhaseBias = 0x41300000u;
struct LeanPhase {
    // Pc: biased phase of one sample, below bias + 2 sr.  WRAPPED: known to be below bias + sr already.
    template <bool WRAPPED>
    static __device__ __forceinline__ void locate(unsigned long long Pc, uint32_t sr, uint32_t M, uint32_t &a_img, bool &upper, double &fraction) {
        const uint32_t h = (uint32_t)(Pc >> 32);
        if (WRAPPED) {
            a_img = abs_diff(h, kPhaseBias + M);
            upper = h > kPhaseBias + M;
        } else {
            const uint32_t s = abs_diff(h, kPhaseBias + sr);
            a_img = abs_diff(s, M);
            upper = (s > M) != (h < kPhaseBias + sr);
        }
        fraction = __builtin_amdgcn_fract(__longlong_as_double((long long)Pc));
    }
    static __device__ __forceinline__ uint32_t index(unsigned long long Pc, uint32_t sr) {  // the table index itself (gathers from L2)
        const uint32_t h = (uint32_t)(Pc >> 32);
        return min(h - kPhaseBias, h - (kPhaseBias + sr));  // (an underflow loses the min)
    }
    static __device__ __forceinline__ unsigned long long step(unsigned long long P, unsigned long long c, uint32_t sr) {  // (P + c) mod sr, both below sr
        const unsigned long long s = P + c;
        const uint32_t h = (uint32_t)(s >> 32);
        const uint32_t u = h - sr;  // (sub, compare, select: the constant goes into the compare, no register is loaded with sr)
        return ((unsigned long long)(u >= kPhaseBias ? u : h) << 32) | (uint32_t)s;
    }
};

__device__ __forceinline__ uint32_t mod_u32(uint32_t x, uint32_t m, double inv_m) {
    const uint32_t q = (uint32_t)((double)x * inv_m);
    uint32_t r = x - q * m;
    if ((int32_t)r < 0) r += m;
    if (r >= m) r -= m;
    return r;
}
__device__ __forceinline__ uint64_t mod_u64(uint64_t x, uint64_t m, double inv_m) {  // x < 2^64, m < 2^48
    const uint64_t q = (uint64_t)((double)x * inv_m);
    uint64_t r = x - q * m;
    if ((int64_t)r < 0) r += m;
    if ((int64_t)r < 0) r += m;
    if (r >= m) r -= m;
    if (r >= m) r -= m;
    return r;
}

__device__ __forceinline__ unsigned long long mod_u64_lifted(unsigned long long x, unsigned long long m, double inv_m) {
    return mod_u64(x, m, inv_m);  // x already lifted to a non-negative value below 2^63
}

template <bool FINITE>
__device__ __forceinline__ float fix_out(float v) {  // `x || 0` (renderChannelData.js:44): NaN, -0 -> +0
    if (FINITE) return v + 0.f;                       // operands verified finite on the host: only -0 can occur
    return __builtin_amdgcn_classf(v, 0x23) ? 0.f : v;  // v_cmp_class_f32 (signalling NaN | quiet NaN | -0) + one select (`..._class` without the f is the f64 test: a conversion more)
}

template <bool VEC>
__device__ __forceinline__ void store4(float *row, const float (&v)[4], uint64_t t, uint64_t n_samples) {
    if (VEC) {
#ifdef DUSP_NT_STORES
        __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, (f32x4 *)row);
#else
        *(f32x4 *)row = f32x4{v[0], v[1], v[2], v[3]};  // plain stores measured 1-2 % faster than `nt` here (tools/abench.py)
#endif
    } else
        for (int c = 0; c < 4; ++c)
            if (t + c < n_samples) row[c] = v[c];
}

// Cooperative fill of the LDS half-table image (33-word pitch, backwards from the middle: see Table<1>).
template <int BLOCK>
__device__ __forceinline__ void load_half_table(float *lds, const float *table, uint32_t sample_rate) {
    const uint32_t last = sample_rate / 2 + 1;
    const uint32_t n_words = last + (last >> 5) + 2;
    for (uint32_t q = threadIdx.x; q < n_words; q += BLOCK) {
        const uint32_t t = (q / 33) * 32 + (q % 33);  // E[t] = T[M + 1 - t]
        lds[q] = table[last - min(t, last)];
    }
    __syncthreads();
}

__host__ __device__ inline size_t half_table_lds_bytes(uint32_t sample_rate) { return (size_t)half_table_image_bytes(sample_rate); }

}  // namespace
}  // namespace dusp
