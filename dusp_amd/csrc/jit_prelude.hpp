// jit_prelude.hpp — device library of the per-circuit kernels.
//
// The circuit compiler (jit_codegen.hpp) turns ONE topologically sorted Circuit into ONE kernel: the chunk loop of
// Circuit.tick (src/Circuit.js:19-41) with every unit's `_tick` inlined in process order as straight-line code.  The kernel
// keeps the wave engine's mapping — one WAVEFRONT per circuit instance (or per time segment of one), one lane per 4 samples,
// 256-sample chunks — but nothing is interpreted and nothing that need not be is staged: a unit's output chunk is four
// floats in this lane's REGISTERS (a consumer reads exactly the samples its own lane produced), a feedback edge is the same
// registers carried over from the previous iteration (the reference's implicit one-chunk delay), unit state that is uniform
// over the wave (phase carries, ramp counters, filter coefficients) sits in scalar registers, and LDS holds only what lanes
// really share: the wave-table image, the Filter stage's hand-over tiles, a delay line's window.  The functions below are
// the units; the generated text only declares them, wires their operands and stores the outlet.
//
// hiprtc compiles this file (with device_types.hpp, device_util.hpp, map_ops.hpp, repeat_add.hpp, jit_args.hpp: their text
// is embedded in the library) together with the generated kernel, for gfx950, with -ffp-contract=off like every other
// kernel here: JS rounds each f64 sub-expression separately and so does this code.
#pragma once
#include "device_types.hpp"
#include "device_util.hpp"
#include "jit_args.hpp"
#include "map_ops.hpp"
#include "repeat_add.hpp"

namespace dusp {
namespace {

constexpr double kJ36 = 68719476736.0;  // oscillator phases are exact fixed point in units of 2^-36 (SURVEY.md §8a note ii)
constexpr int kJFrac = 36;
constexpr unsigned long long kJMask = (1ull << kJFrac) - 1ull;

__device__ __forceinline__ void jit_lds_barrier() {  // orders LDS traffic only: PCM / ring stores stay in flight across it
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ double jit_or0(double v) { return (v != v || v == 0.0) ? 0.0 : v; }  // JS `v || 0`

// What a wavefront knows about its place: instance, time segment, chunk range.  Everything but `lane` is wave-uniform.
struct JitCtx {
    uint32_t lane, wave, inst, seg, g_begin, g_end, sr, n_seg;  // g_begin .. g_end: the chunks the wave TICKS
    uint32_t g_own, g_stop;  // ... and the ones it stores (the same, but for a segment that warms up: JitArgs::warm)
    bool live;  // a short last workgroup keeps its surplus waves alive (they stand at the Filter stage's barriers): they shadow the last instance and never store
    double srd, inv_S;
    unsigned long long S, lift;
    Table<1> table;  // the half-table image in LDS (kernels generated with one)
    __device__ __forceinline__ uint64_t n0(uint32_t g) const { return (uint64_t)g * kChunk + lane * 4; }  // this lane's first sample of chunk g
};

// R: circuit instances per wavefront (unsplit renders of many instances; a time-split render has R == 1).  A wave's R instances
// are consecutive: workgroup b, wave w, slot r renders instance (b WAVES + w) R + r.  Their unit blocks stand next to each other
// in the generated text, so the independent work of R instances fills each other's latencies, and the Filter stage gets
// WAVES x R recurrences to run side by side.
template <int WAVES, int LDS_TABLE, int R>
__device__ __forceinline__ void jit_begin(const JitArgs &A, float *lds, JitCtx (&X)[R]) {
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (LDS_TABLE >= 0) load_half_table<WAVES * 64>(lds, A.tables + (size_t)LDS_TABLE * A.table_stride, A.sample_rate);
    const uint32_t n_virtual = A.n_inst * A.n_seg;  // a "virtual instance" is one time segment of an instance
#pragma unroll
    for (int r = 0; r < R; ++r) {
        JitCtx &x = X[r];
        x.sr = A.sample_rate;
        x.srd = (double)x.sr;
        x.S = (unsigned long long)x.sr << kJFrac;
        x.inv_S = 1.0 / (double)x.S;
        x.lift = x.S * ((1ull << 62) / x.S);  // multiple of S that makes any |x| < 2^62 non-negative before the modulo
        x.table.g = nullptr;
        x.table.h = lds;
        x.table.N = x.sr + 1;
        x.table.M = x.sr / 2;
        x.lane = lane;
        x.wave = wave;
        x.n_seg = A.n_seg;
        const uint32_t v = (blockIdx.x * WAVES + wave) * R + r;
        x.live = v < n_virtual;
        const uint32_t vinst = x.live ? v : n_virtual - 1;
        x.inst = vinst / A.n_seg;
        x.seg = vinst - x.inst * A.n_seg;
        x.g_begin = x.seg * A.seg_groups + A.g_first;
        x.g_end = A.n_seg == 1 ? A.n_groups : min(x.g_begin + A.seg_groups, A.n_groups);
        x.g_own = x.g_begin;
        x.g_stop = x.g_end;
        if (A.warm) {  // one segment of warm-up (none in front of the first), and the same trip count for every wave of the workgroup: they meet at the Filter stage's barriers
            x.g_begin = x.seg ? x.g_own - A.seg_groups : 0u;
            x.g_end = x.g_begin + 2u * A.seg_groups + 1u;  // (one past every segment's last chunk: what a Filter holds THERE is recorded at the top of an iteration)
        }
    }
}

// Values that are the same in every lane, made known as such: the compiler keeps them in scalar registers instead of one copy
// per lane (a load through a plain pointer lands in vector registers even when its address is wave-uniform).
__device__ __forceinline__ uint32_t jit_u(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane(v); }  // (the builtin returns int)
__device__ __forceinline__ float jit_u(float v) { return __uint_as_float(jit_u(__float_as_uint(v))); }
__device__ __forceinline__ unsigned long long jit_u(unsigned long long v) {
    return ((unsigned long long)jit_u((uint32_t)(v >> 32)) << 32) | (unsigned long long)jit_u((uint32_t)v);
}
__device__ __forceinline__ double jit_u(double v) { return __longlong_as_double((long long)jit_u((unsigned long long)__double_as_longlong(v))); }
__device__ __forceinline__ bool jit_u(bool v) { return jit_u(v ? 1u : 0u) != 0u; }
__device__ __forceinline__ bool jit_small(float k) { return jit_u(fabsf(k) <= 1073741824.f); }  // (a wave-uniform scalar; NaN fails)

// rows of 256 floats in a wave's LDS scratch: every lane its four samples
__device__ __forceinline__ void jit_row_put(float *row, uint32_t lane, const float (&v)[4]) { ((f32x4 *)row)[lane] = f32x4{v[0], v[1], v[2], v[3]}; }
__device__ __forceinline__ void jit_row_get(const float *row, uint32_t lane, float (&v)[4]) {
    const f32x4 x = ((const f32x4 *)row)[lane];
    v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
}
__device__ __forceinline__ void jit_wave_sync() {  // LDS written by some lanes of this wave, read by others
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float jit_param(const JitArgs &A, const JitCtx &X, uint32_t slot) { return jit_u(A.params[(size_t)slot * A.n_inst + X.inst]); }

// Continued programs (dusp_program_continue): an outlet's last chunk waits in saved_bufs [n_inst][n_bufs][256] between two launches —
// what a feedback (or late) edge reads first in the next one, and what the chunk engine takes over if the chain migrates.
__device__ __forceinline__ void jit_park(const JitArgs &A, const JitCtx &X, uint32_t buf, const float (&v)[4]) {
    if (X.live) ((f32x4 *)(A.saved_bufs + ((size_t)X.inst * A.n_bufs + buf) * kChunk))[X.lane] = f32x4{v[0], v[1], v[2], v[3]};
}
__device__ __forceinline__ void jit_unpark(const JitArgs &A, const JitCtx &X, uint32_t buf, float (&v)[4]) {
    const f32x4 x = ((const f32x4 *)(A.saved_bufs + ((size_t)X.inst * A.n_bufs + buf) * kChunk))[X.lane];
    v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
}

// (a * n) mod S for a < S < 2^53 and n < 2^32, exactly: the product is cut into pieces that mod_u64 (x < 2^64) can take
__device__ __forceinline__ unsigned long long jit_mulmod(unsigned long long a, unsigned long long n, unsigned long long S, double inv_S) {
    const unsigned long long ah = a >> 32, al = a & 0xffffffffull;  // ah < 2^21
    unsigned long long r = mod_u64(ah * n, S, inv_S);               // (ah n) < 2^53; now r 2^32 mod S in three shifts of <= 11 bits
    r = mod_u64(r << 11, S, inv_S);
    r = mod_u64(r << 11, S, inv_S);
    r = mod_u64(r << 10, S, inv_S);
    return addmod(r, mod_u64(al * n, S, inv_S), S);
}

// Inclusive prefix sum of a 32-bit value over the 64 lanes, in the VALU's own cross-lane network (DPP): four shifts inside
// each row of 16 lanes, then two row broadcasts — six adds, no LDS crossbar round trips (a ds_bpermute shuffle costs ~50
// cycles of latency, and a 64-bit scan by shuffles chains twelve of them).
__device__ __forceinline__ int jit_scan32(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);  // row_shr:1 (lanes without a source add 0)
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);  // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2 and 3
    return x;
}
// Inclusive prefix sum of signed 64-bit values with |x| < 2^56 over the 64 lanes: three limbs of 22 bits (the top one signed),
// scanned separately — a limb's 64-lane sum stays below 2^28 — and put together again.
__device__ __forceinline__ long long jit_wave_scan(long long x) {
    const int l0 = (int)((unsigned long long)x & 0x3fffffull), l1 = (int)(((unsigned long long)x >> 22) & 0x3fffffull), l2 = (int)(x >> 44);
    const long long s0 = jit_scan32(l0), s1 = jit_scan32(l1), s2 = jit_scan32(l2);
    return s0 + s1 * (1ll << 22) + s2 * (1ll << 44);
}

// f32 -> exact 2^-36 fixed point, truncated toward zero like the C cast of f * 2^36 (every f32 with |f| >= 2^-13 is a multiple
// of 2^-36).  For finite |f| < 2^16: f 2^36 is below 2^52, so |it| + 2^52 carries the magnitude in its mantissa field as an
// integer — four f64 instructions and a correction of the high word instead of unpacking the f32 with 64-bit shifts (ten
// instructions against eighteen).  Callers send larger magnitudes (which need an fmod by the sample rate first) and
// non-finite values down their slow path (a NaN or an out-of-range value gives some integer here, no trap).
__device__ __forceinline__ long long jit_fix36(float f) {
    const double x = __builtin_trunc(ldexp((double)f, 36));
    const unsigned long long mag = (unsigned long long)__double_as_longlong(fabs(x) + 4503599627370496.0) - 0x4330000000000000ull;
    return f < 0.f ? -(long long)mag : (long long)mag;
}

// One table lookup pair (T[idx], T[idx + 1]).  TF says where this oscillator's table comes from:
//   0  gathered from L2 (a table that is neither antisymmetric nor a closed form)
//   1  the LDS half-table image the kernel carries (antisymmetric tables: sine, 8bit)
//   2  no table: saw / square / triangle evaluated from the index (FORM = TABLE_FORM_*; the host has checked every entry)
//   3  8bit evaluated from the SINE image in LDS (circuits that mix sine and 8bit oscillators)
template <int TF, int FORM>
__device__ __forceinline__ void jit_pair(const JitCtx &X, const float *gtab, uint32_t idx, float &a, float &b) {
    if (TF == 1) X.table.pair(idx, a, b);
    else if (TF == 2) {
        const TableForm F = make_table_form(FORM, X.sr);
        a = closed_table_entry(F, idx);
        b = closed_table_entry(F, idx + 1 <= X.sr ? idx + 1 : X.sr);  // (the table's pad entry repeats its last one)
    } else if (TF == 3) {
        float sa, sb;
        X.table.pair(idx, sa, sb);
        a = eightbit_of_sine(sa);
        b = eightbit_of_sine(sb);
    } else {
        a = gtab[idx];
        b = gtab[idx + 1];
    }
}

// ---- Osc (src/components/Osc/Osc.js:35-47) with an unconnected f — a constant or a per-instance parameter.  Equal increments:
// phase(n) = (phase0 + (n + 1) q) mod S in exact 2^-36 fixed point, so the lane jumps to its own samples and thereafter
// advances by (256 q) mod S per chunk; nothing crosses lanes, nothing is carried but this lane's own phase.
// Two forms of the same arithmetic, chosen per KERNEL BODY (the generated kernel holds its chunk loop twice and takes the
// fast copy when every such oscillator of the wave qualifies, so that no branch stands inside the loop):
//   FX   every phase is a multiple of 2^-32 (any f32 f with |f| >= 2^-8, i.e. lsb(f) >= 2^-31, from a start phase on that
//        grid): index and fraction are two u32, advanced by add-with-carry; the fraction converts to f64 in one step
//   LEAN every phase is a multiple of 2^-28 (any f32 f with |f| >= 2^-4) and the table's neighbours differ by exact values: the
//        lerp in its delta form (one fma: device_util.hpp lerp_delta), index.fraction as ONE 64-bit integer that is also a
//        double (LeanPhase) — the lane's four phases are four v_lshl_add_u64 off its first (q, 2q, 3q and the chunk's step
//        in scalar registers), the fraction is v_fract_f64 of it, the wrap and the fold into the half image are two v_sad_u32
//   else the general u64 form
// All the lookups of a chunk are issued before the first lerp.  The FX and LEAN copies are only entered with a finite f, so
// they carry no NaN select.
struct JitOscK {
    uint32_t I, F;                 // FX: index and 2^-32 fraction of this lane's first sample of the next chunk
    uint32_t qI, qF, cI, cF;       // increment per sample / per chunk in the same form (wave-uniform)
    unsigned long long P;          // general form: phase of this lane's first sample of the next chunk
    unsigned long long q, q256;    // increment per sample / per chunk, mod S (wave-uniform)
    unsigned long long P_init;     // phase before the render's first sample (wave-uniform)
    unsigned long long PD;         // LEAN: this lane's first sample of the next chunk, index.fraction as a biased 64-bit integer (LeanPhase)
    unsigned long long Q1, Q2, Q3, C; // LEAN: q, 2q, 3q, 256q mod S as 32.32 integers (wave-uniform)
    bool bad, fx32, lean;          // bad: f is NaN / Inf (every sample NaN); fx32 / lean: qualifies for the FX / LEAN form

    __device__ __forceinline__ void begin(const JitArgs &A, const JitCtx &X, float f, int state_slot) {
        double fd = (double)f;
        bad = !(fabs(fd) <= 3.0e38);
        if (bad) fd = 0.0;
        if (fabs(fd) >= X.srd) fd = fmod(fd, X.srd);  // (a + b) % m == (a + b % m) % m
        const long long qs = (long long)(fd * kJ36);  // exact for |f| >= 2^-13 (or f == 0)
        q = qs >= 0 ? (unsigned long long)qs : X.S - (unsigned long long)(-qs);
        if (q >= X.S) q -= X.S;
        if (bad) q = 1ull;  // (every sample is NaN whatever the phase; an increment off the 2^-32 grid keeps such an oscillator out of the FX / LEAN copies, which carry no select for it)
        P_init = (unsigned long long)(A.init_state[state_slot] * kJ36);
        q256 = jit_mulmod(q, kChunk, X.S, X.inv_S);
        P = addmod(P_init, jit_mulmod(q, X.n0(X.g_begin) + 1, X.S, X.inv_S), X.S);
        q = jit_u(q);
        q256 = jit_u(q256);
        P_init = jit_u(P_init);
        bad = jit_u(bad);
        fx32 = ((q | P_init) & 15ull) == 0ull;  // (never with a NaN / Inf f: q is odd then)
        lean = ((q | P_init) & 255ull) == 0ull;
        const unsigned long long q2 = addmod(q, q, X.S);
        Q1 = q >> 4;
        Q2 = q2 >> 4;
        Q3 = addmod(q2, q, X.S) >> 4;
        C = q256 >> 4;
        I = (uint32_t)(P >> kJFrac);
        F = (uint32_t)((P & kJMask) >> 4);
        PD = ((unsigned long long)(I + kPhaseBias) << 32) | F;
        qI = (uint32_t)(q >> kJFrac);
        qF = (uint32_t)((q & kJMask) >> 4);
        cI = (uint32_t)(q256 >> kJFrac);
        cF = (uint32_t)((q256 & kJMask) >> 4);
    }
    static __device__ __forceinline__ void step32(uint32_t &i, uint32_t &f, uint32_t di, uint32_t df, uint32_t sr) {  // (i.f + di.df) mod sr
        const uint32_t f2 = f + df;
        uint32_t i2 = i + di + (f2 < f ? 1u : 0u);
        i2 = min(i2, i2 - sr);  // (i2 < 2 sr; an underflow loses the min)
        f = f2;
        i = i2;
    }
    template <int TF, int FORM, int MODE>  // MODE: 0 general, 1 FX, 2 LEAN (differences of neighbours in f64), 3 LEAN (in f32)
    __device__ __forceinline__ void tick(const JitCtx &X, const float *gtab, float (&out)[4]) {
        float ta[4], tb[4];
        if (MODE >= 2) {  // LEAN (device_util.hpp LeanPhase, lerp_delta)
            double da[4], dd[4], fr[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const unsigned long long Pc = c == 0 ? PD : PD + (c == 1 ? Q1 : c == 2 ? Q2 : Q3);
                uint32_t a_img;
                bool upper;
                if (c == 0) LeanPhase::locate<true>(Pc, X.sr, X.table.M, a_img, upper, fr[c]);
                else LeanPhase::locate<false>(Pc, X.sr, X.table.M, a_img, upper, fr[c]);
                if (TF == 1) X.table.pair_delta_at<MODE == 3>(a_img, upper, da[c], dd[c]);
                else {
                    Table<0> t;
                    t.g = gtab;
                    t.pair_delta<MODE == 3>(LeanPhase::index(Pc, X.sr), da[c], dd[c]);
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) out[c] = (float)fma(dd[c], fr[c], da[c]);
            PD = LeanPhase::step(PD, C, X.sr);
            return;
        }
        if (MODE == 1) {
            uint32_t iv[4], fv[4];
            iv[0] = I;
            fv[0] = F;
#pragma unroll
            for (int c = 1; c < 4; ++c) {
                iv[c] = iv[c - 1];
                fv[c] = fv[c - 1];
                step32(iv[c], fv[c], qI, qF, X.sr);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) jit_pair<TF, FORM>(X, gtab, iv[c], ta[c], tb[c]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double fraction = (double)fv[c] * (1.0 / 4294967296.0);
                out[c] = (float)((double)ta[c] * (1.0 - fraction) + (double)tb[c] * fraction);
            }
            step32(I, F, cI, cF, X.sr);
        } else {
            unsigned long long Pv[4];
            Pv[0] = P;
#pragma unroll
            for (int c = 1; c < 4; ++c) Pv[c] = addmod(Pv[c - 1], q, X.S);
#pragma unroll
            for (int c = 0; c < 4; ++c) jit_pair<TF, FORM>(X, gtab, (uint32_t)(Pv[c] >> kJFrac), ta[c], tb[c]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double fraction = (double)(Pv[c] & kJMask) * (1.0 / kJ36);
                out[c] = (float)((double)ta[c] * (1.0 - fraction) + (double)tb[c] * fraction);
            }
            P = addmod(P, q256, X.S);
        }
        if (MODE == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) out[c] = bad ? __builtin_nanf("") : out[c];
        }
    }
    // the phase after ceil(n_samples / 256) ticks (state write-back)
    __device__ __forceinline__ double end_phase(const JitArgs &A, const JitCtx &X) const {
        if (bad) return __builtin_nan("");
        return (double)addmod(P_init, jit_mulmod(q256, A.n_groups, X.S, X.inv_S), X.S) * (1.0 / kJ36);  // n_groups * 256 samples
    }
};

// What a constant-f oscillator carries from chunk to chunk, for circuits whose voices run in a loop (jit_codegen.hpp run_voices): the
// oscillators of voice j live in element j of a per-lane array, so only what tick() reads is kept — the LEAN form's phase and steps,
// the general form's, and the phase the unit ends the render with (a function of f and the render's length alone).
struct JitOscKV {
    unsigned long long PD, Q1, Q2, Q3, C;  // LEAN
    unsigned long long P, q, q256;         // general form
    double end;                            // end_phase()
    uint32_t bad;
    __device__ __forceinline__ void keep(const JitArgs &A, const JitCtx &X, const JitOscK &o) {
        PD = o.PD; Q1 = o.Q1; Q2 = o.Q2; Q3 = o.Q3; C = o.C;
        P = o.P; q = o.q; q256 = o.q256;
        end = o.end_phase(A, X);
        bad = o.bad ? 1u : 0u;
    }
    template <int MODE>
    __device__ __forceinline__ void lend(JitOscK &o) const {  // the fields tick<.., MODE> reads (wave-uniform ones made known as such)
        if (MODE >= 2) {
            o.PD = PD; o.Q1 = jit_u(Q1); o.Q2 = jit_u(Q2); o.Q3 = jit_u(Q3); o.C = jit_u(C);
        } else {
            o.P = P; o.q = jit_u(q); o.q256 = jit_u(q256);
        }
        o.bad = jit_u(bad) != 0u;
    }
    template <int MODE>
    __device__ __forceinline__ void take(const JitOscK &o) {
        if (MODE >= 2) PD = o.PD;
        else P = o.P;
    }
};

// ---- Osc with a connected f (FM): wavefront-wide phase accumulation.  Increments -> exact 2^-36 fixed point -> prefix of 4
// inside the lane -> 6-step integer scan over the wave (sums < 2^61, no modulo inside) -> one exact modulo for the lane's
// first sample, add-and-wrap for the next three.  Carried from chunk to chunk: the phase of the chunk's last sample and a
// poison flag (a NaN / Inf increment makes the reference's phase NaN for good) — both wave-uniform.
struct JitOscS {
    unsigned long long carry;
    uint32_t poison;

    // accumulating: a pass that only totals this oscillator's increments over the segment (time-split rendering)
    __device__ __forceinline__ void begin(const JitArgs &A, const JitCtx &X, int state_slot, int op_index, bool accumulating) {
        carry = 0ull;
        poison = 0u;
        if (X.n_seg == 1) carry = jit_u((unsigned long long)(A.init_state[state_slot] * kJ36));
        else if (!accumulating) {  // start phase known from the accumulate + prefix passes
            const uint32_t from = A.warm ? (X.seg ? X.seg - 1u : 0u) : X.seg;  // (a segment that warms up starts where the segment before it starts)
            const unsigned long long v = jit_u(A.seg_start[((size_t)op_index * A.n_inst + X.inst) * X.n_seg + from]);
            carry = v & ~(1ull << 63);
            poison = (uint32_t)(v >> 63);
        }
    }
    // Does this chunk need the careful path?  Wave-uniform: a poisoned oscillator, or an increment at or above the sample rate
    // (the reference's `phase %= sampleRate` folds those) or NaN / Inf.  The generated kernel asks all the instances of a wave
    // first and then runs ONE straight-line block for all of them, so that their independent work can interleave.
    __device__ __forceinline__ bool rare(const JitCtx &X, const float (&f)[4]) const {
        bool big = false;
#pragma unroll
        for (int c = 0; c < 4; ++c) big = big || !(fabsf(f[c]) < (float)X.sr);
        return poison != 0 || __any(big);
    }
    // RARE = false: every increment finite and below the sample rate in magnitude, the oscillator not poisoned (rare() said so).
    // S = sr << 36 has a zero low word, so every "+- S" and every comparison with S touches the HIGH word of a phase only.
    template <int TF, int FORM, bool LOOKUP, bool RARE>
    __device__ __forceinline__ void tick(const JitCtx &X, const float *gtab, const float (&f)[4], float (&out)[4]) {
        long long qv[4];
        bool bad = false;
#pragma unroll
        for (int c = 0; c < 4; ++c) qv[c] = jit_fix36(f[c]);
        if (RARE) {
            bool big = false;
#pragma unroll
            for (int c = 0; c < 4; ++c) big = big || !(fabsf(f[c]) < (float)X.sr);
            if (__any(big)) {  // (a + b) % m == (a + b % m) % m
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    double fd = (double)f[c];
                    const bool fin = fabs(fd) <= 3.0e38;
                    bad = bad || !fin;
                    if (!fin) fd = 0.0;
                    if (fabs(fd) >= X.srd) fd = fmod(fd, X.srd);
                    qv[c] = (long long)(fd * kJ36);
                }
            }
        }
        const long long total = qv[0] + qv[1] + qv[2] + qv[3];  // |total| < 4 * 2^53
        const long long incl = jit_wave_scan(total);
        const long long x0 = (long long)carry + (incl - total) + qv[0];  // |x0| < 256 S: a small quotient
        const uint32_t S_hi = X.sr << 4;
        // x0 mod S: quotient from the high words in f32 (off by one at most), remainder fixed up — all on the high word
        uint32_t hi[4], lo[4];
        {
            const int xh = (int)(x0 >> 32);
            const int k = (int)floorf((float)xh * (1.0f / (float)S_hi));
            int rh = xh - k * (int)S_hi;
            rh = rh < 0 ? rh + (int)S_hi : rh;
            rh = rh >= (int)S_hi ? rh - (int)S_hi : rh;
            rh = rh < 0 ? rh + (int)S_hi : rh;
            hi[0] = (uint32_t)rh;
            lo[0] = (uint32_t)x0;
        }
#pragma unroll
        for (int c = 1; c < 4; ++c) {  // + q (|q| < S): 64-bit add, then back into [0, S) by the high word
            const unsigned long long sum = (((unsigned long long)hi[c - 1] << 32) | lo[c - 1]) + (unsigned long long)qv[c];
            uint32_t t = (uint32_t)(sum >> 32) + S_hi;  // in (0, 3 S_hi)
            t = min(t, t - S_hi);
            t = min(t, t - S_hi);
            hi[c] = t;
            lo[c] = (uint32_t)sum;
        }
        if (LOOKUP) {  // all the lookups of the chunk first, then the lerps
            float ta[4], tb[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) jit_pair<TF, FORM>(X, gtab, hi[c] >> 4, ta[c], tb[c]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                // the 36 fraction bits as a double: mantissa of 2^16 + fraction (its last place is 2^-36), minus 2^16 — exact
                const double fraction = __hiloint2double((int)(0x40F00000u | (hi[c] & 15u)), (int)lo[c]) - 65536.0;
                out[c] = (float)((double)ta[c] * (1.0 - fraction) + (double)tb[c] * fraction);
            }
        }
        const unsigned long long bad_lanes = RARE ? __ballot(bad) : 0ull;
        if (RARE && LOOKUP && (poison != 0 || bad_lanes != 0)) {  // (uniform) from the first NaN / Inf increment on, every sample is NaN
            bool dead = poison != 0 || (bad_lanes & ((1ull << X.lane) - 1ull)) != 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                dead = dead || !(fabs((double)f[c]) <= 3.0e38);
                if (dead) out[c] = __builtin_nanf("");
            }
        }
        // the chunk's last phase, as a scalar (lane 63 holds it)
        carry = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(hi[3], 63) << 32) | (uint32_t)__builtin_amdgcn_readlane(lo[3], 63);
        if (bad_lanes) poison = 1u;
    }
    __device__ __forceinline__ double end_phase() const { return poison ? __builtin_nan("") : (double)carry * (1.0 / kJ36); }
    // ... of a render cut into segments that warm up (whose wavefronts tick past their own chunks): the last segment's start phase plus its total
    static __device__ __forceinline__ double warm_end_phase(const JitArgs &A, const JitCtx &X, int op_index) {
        const size_t at = ((size_t)op_index * A.n_inst + X.inst) * X.n_seg + (X.n_seg - 1);
        const unsigned long long v = A.seg_start[at], t = A.seg_sum[at];
        if ((v | t) >> 63) return __builtin_nan("");
        unsigned long long p = v + t;  // (both below S)
        if (p >= X.S) p -= X.S;
        return (double)p * (1.0 / kJ36);
    }
    __device__ __forceinline__ unsigned long long packed() const { return carry | ((unsigned long long)(poison != 0) << 63); }
};

// ---- Ramp (src/components/Ramp.js:25-40) in closed form: t(n) = min(t0 + n + 1, duration) while playing.
// FASTDIV: t / duration by a 2-FMA refined reciprocal that the host has verified against true division on every t this
// Ramp can take (fused_plan.hpp ramp_fastdiv_ok).
template <bool FASTDIV>
__device__ __forceinline__ void jit_ramp(const JitCtx &X, uint32_t g, double duration, double y0, double y1, double t0, bool playing, float (&out)[4]) {
    const double dy = y1 - y0, rcp = 1.0 / duration;
    const uint64_t n0 = X.n0(g);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const double tt = playing ? fmin(t0 + (double)(n0 + c + 1), duration) : t0;
        double q;
        if (FASTDIV) {
            q = tt * rcp;
            q = fma(fma(-q, duration, tt), rcp, q);
        } else
            q = tt / duration;
        out[c] = (float)(y0 + q * dy);
    }
}
__device__ __forceinline__ void jit_ramp_end(const JitArgs &A, const JitCtx &X, double duration, int state_slot) {
    const double t0 = A.init_state[state_slot], since = (double)((uint64_t)A.n_groups * kChunk);
    const bool playing = A.init_state[state_slot + 1] != 0.0;
    double *st = A.state + (size_t)state_slot * A.n_pad + X.inst;
    st[0] = playing ? fmin(t0 + since, duration) : t0;
    st[A.n_pad] = (playing && t0 + since <= duration) ? 1.0 : 0.0;
}

// ---- A Ramp that a Retriggerer of the same circuit restarts (Ramp.js:19-23 trigger(): t = 0, playing): the closed form counted from
// the chunk of its last restart.  (No refined reciprocal here: nobody has checked it on the t sequences a restart runs through.)
struct JitRampR {
    double t0;                  // t before the sample `origin`
    unsigned long long origin;  // ... a sample index of this launch
    bool playing;
    __device__ __forceinline__ void begin(const JitArgs &A, int state_slot) {
        t0 = jit_u(A.init_state[state_slot]);
        playing = jit_u(A.init_state[state_slot + 1] != 0.0);
        origin = 0ull;
    }
    __device__ __forceinline__ void fire(uint32_t g) {  // trigger() in front of chunk g's tick
        t0 = 0.0;
        origin = (unsigned long long)g * kChunk;
        playing = true;
    }
    __device__ __forceinline__ void tick(const JitCtx &X, uint32_t g, double duration, double y0, double y1, float (&out)[4]) const {
        const double dy = y1 - y0;
        const uint64_t n0 = X.n0(g) - origin;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double tt = playing ? fmin(t0 + (double)(n0 + c + 1), duration) : t0;
            out[c] = (float)(y0 + (tt / duration) * dy);
        }
    }
    __device__ __forceinline__ void end(const JitArgs &A, const JitCtx &X, double duration, int state_slot) const {
        const double since = (double)((uint64_t)A.n_groups * kChunk - origin);
        double *st = A.state + (size_t)state_slot * A.n_pad + X.inst;
        st[0] = playing ? fmin(t0 + since, duration) : t0;
        st[A.n_pad] = (playing && t0 + since <= duration) ? 1.0 : 0.0;
    }
};

// ---- Retriggerer (src/components/Retriggerer.js:13-24) of a Shape / an AHD / a Ramp of the same circuit: t += rate per sample; a crossing of the
// sample rate triggers the target and subtracts the rate's unit.  Wave-uniform.  Most chunks see no crossing: then the accumulator is a
// plain running f64 sum, which repeat_add() evaluates at once; a chunk with a crossing is walked (the subtraction breaks the closed form).
struct JitRetrig {
    double T;
    __device__ __forceinline__ void begin(const JitArgs &A, int state_slot) { T = jit_u(A.init_state[state_slot]); }
    __device__ __forceinline__ bool tick(const JitCtx &X, float rate_f) {  // -> fired in this chunk
        const double rate = (double)rate_f, srd = X.srd;
        bool fired = false;
        double t = T;
        const double quiet_end = (t >= 0.0 && rate > 0.0 && rate < 1.0e300) ? repeat_add(t, rate, kChunk) : srd;
        if (quiet_end < srd) t = quiet_end;
        else {
#pragma unroll 8
            for (int k = 0; k < kChunk; ++k) {
                t += rate;
                if (t >= srd) { fired = true; t -= srd; }
            }
        }
        T = jit_u(t);
        return jit_u(fired);
    }
};

// ---- Timer (src/components/Timer.js:36-41): t += samplePeriod, each sum rounded to f64 — in closed form (repeat_add.hpp).  The
// running value before the chunk is carried as a scalar; while a whole chunk stays inside one binade the 256 sums are
// t_j = (T + j ce) 2^(K-52) (linear_run), else every lane jumps to its own four.
struct JitTimer {
    double t;  // value before the next chunk's first sample (uniform)
    __device__ __forceinline__ void begin(const JitArgs &A, const JitCtx &X, double period, int state_slot) {
        t = jit_u(repeat_add(A.init_state[state_slot], period, (uint64_t)X.g_begin * kChunk));
    }
    __device__ __forceinline__ void tick(const JitCtx &X, double period, float (&out)[4]) {
        long long T, ce;
        int K;
        if (linear_run(t, period, kChunk, T, ce, K)) {
            long long Tl = T + (long long)(X.lane * 4) * ce;
#pragma unroll
            for (int c = 0; c < 4; ++c) out[c] = (float)ldexp((double)(Tl += ce), K - 52);
            t = jit_u(ldexp((double)(T + (long long)kChunk * ce), K - 52));
        } else {
            double tt = repeat_add(t, period, (uint64_t)X.lane * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) out[c] = (float)(tt = tt + period);
            t = jit_u(__shfl(tt, 63, 64));
        }
    }
};

// ---- Shape (src/components/Shape/index.js:28-59) with an unconnected duration: while playing, t += 1 / duration is a running
// f64 sum with a constant addend — closed form (repeat_add.hpp), like the Timer — then the 0..1 shape at t (a lerp in the
// Shape's table, or an edge value outside 0..sampleRate) scaled into [min, max].  min / max may be signals.
struct JitShape {
    double t;       // value before the next chunk's first sample (uniform)
    double c;       // 1 / duration (uniform)
    bool playing, finished;
    __device__ __forceinline__ void begin(const JitArgs &A, const JitCtx &X, float duration, int state_slot) {
        c = jit_u(1.0 / (double)duration);
        playing = jit_u(A.init_state[state_slot + 1] != 0.0);
        finished = jit_u(A.init_state[state_slot + 2] != 0.0);
        const double t0 = A.init_state[state_slot];
        // (a render is only split in time when the duration is a constant inside the closed form's regime: fused_plan.hpp)
        t = jit_u(playing && X.g_begin > 0 ? repeat_add(t0, c, (uint64_t)X.g_begin * kChunk) : t0);
    }
    // left / right: the edge values as numbers; attr bit 8 / 9: the edge is "shape" (= the table's first / last entry)
    __device__ __forceinline__ void tick(const JitCtx &X, const float *data, int attr, double left_k, double right_k, const float (&mn)[4],
                                         const float (&mx)[4], float (&out)[4]) {
        double tt[4];
        if (playing && !(c > 0.0 && c < 1.0e300 && t >= 0.0)) {
            // outside the closed form's regime (a per-instance duration that is zero, negative or NaN): the reference's additions
            // one by one, every lane up to its own samples
            double tl = t;
            for (uint32_t i = 0; i < X.lane * 4; ++i) tl += c;
#pragma unroll
            for (int k = 0; k < 4; ++k) tt[k] = tl = tl + c;
            t = jit_u(__shfl(tl, 63, 64));
        } else if (playing) {
            long long T, ce;
            int K;
            if (linear_run(t, c, kChunk, T, ce, K)) {  // the whole chunk inside one binade: t_j = (T + j ce) 2^(K-52)
                long long Tl = T + (long long)(X.lane * 4) * ce;
#pragma unroll
                for (int k = 0; k < 4; ++k) tt[k] = ldexp((double)(Tl += ce), K - 52);
                t = jit_u(ldexp((double)(T + (long long)kChunk * ce), K - 52));
            } else {
                double tl = repeat_add(t, c, (uint64_t)X.lane * 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) tt[k] = tl = tl + c;
                t = jit_u(__shfl(tl, 63, 64));
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) tt[k] = t;
        }
        values(X, data, attr, left_k, right_k, tt, mn, mx, out);
    }
    // the 0..1 shape at t (table lerp, or an edge value) scaled into [min, max] (Shape/index.js:33-58)
    __device__ __forceinline__ void values(const JitCtx &X, const float *data, int attr, double left_k, double right_k, const double (&tt)[4], const float (&mn)[4],
                                           const float (&mx)[4], float (&out)[4]) {
        const double left = (attr & 256) ? (double)data[0] : left_k, right = (attr & 512) ? (double)data[X.sr] : right_k;
        bool over = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double l = (double)mn[k], h = (double)mx[k], tk = tt[k];
            if (tk <= 0.0) out[k] = (float)(left * (h - l) + l);
            else if (tk > X.srd) {
                over = true;
                out[k] = (float)(right * (h - l) + l);
            } else if (tk == tk) {
                const double fl = floor(tk), frac = tk - fl;
                out[k] = (float)(l + (h - l) * ((double)data[(int)ceil(tk)] * frac + (double)data[(int)fl] * (1.0 - frac)));
            } else
                out[k] = __builtin_nanf("");
        }
        if (__ballot(over)) finished = true;  // finish() (UnitOrPatch.js:77-84)
    }
    // A connected duration: t += 1 / duration[t] is a running f64 sum with a different addend every sample — lane 0 adds them up out
    // of the wave's scratch (512 floats), the lookups and the scaling stay lane-parallel.
    __device__ __forceinline__ void tick_signal(const JitCtx &X, float *scr, const float *data, int attr, double left_k, double right_k, const float (&dur)[4],
                                                const float (&mn)[4], const float (&mx)[4], float (&out)[4]) {
        double tt[4];
        if (playing) {
            double *T = (double *)scr;
            jit_wave_sync();
#pragma unroll
            for (int k = 0; k < 4; ++k) T[X.lane * 4 + k] = 1.0 / (double)dur[k];
            jit_wave_sync();
            if (X.lane == 0) {
                double acc = t;
                for (int k = 0; k < kChunk; ++k) {
                    acc += T[k];
                    T[k] = acc;
                }
                t = acc;
            }
            t = jit_u(t);
            jit_wave_sync();
#pragma unroll
            for (int k = 0; k < 4; ++k) tt[k] = T[X.lane * 4 + k];
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) tt[k] = t;
        }
        values(X, data, attr, left_k, right_k, tt, mn, mx, out);
    }
    __device__ __forceinline__ void end(const JitArgs &A, const JitCtx &X, int state_slot) const {
        double *st = A.state + (size_t)state_slot * A.n_pad + X.inst;
        st[0] = t;  // (written by the wave of the LAST segment: its running sum has reached the end of the render)
        st[A.n_pad] = playing ? 1.0 : 0.0;
        st[(size_t)2 * A.n_pad] = finished ? 1.0 : 0.0;
    }
};

// ---- Units with a sequential stage.  Their state evolves sample by sample with its own roundings, so one lane walks the chunk's
// 256 samples out of a per-wave LDS scratch (`scr`: rows of 256 floats) while everything around it stays lane-parallel.
// FixedDelay / CombFilter / AllPass (FixedDelay.js:13-19, CombFilter.js:11-17, AllPass.js:8-15): a private ring of L slots read and
// rewritten one slot per sample, so sample t depends on sample t - L only: the min(L, 256) slots a chunk touches are staged in
// LDS and walked min(L, 64) independent samples at a time.  scr: row 0 input (output in place), row 1 the ring window, row 2 gain.
struct JitComb {
    uint32_t tb;  // tBuffer (uniform)
    __device__ __forceinline__ void begin(const JitArgs &A, int state_slot) { tb = jit_u((uint32_t)A.init_state[state_slot]); }
    template <int KIND /*OP_FIXED_DELAY / OP_COMB_FILTER / OP_ALL_PASS*/, bool GAIN_ROW>
    __device__ __forceinline__ void tick(const JitArgs &A, const JitCtx &X, float *scr, int64_t ring_base, uint32_t L, const float (&x)[4],
                                         const float (&gain)[4], float (&out)[4]) {
        float *Xr = scr, *R = scr + kChunk, *G = scr + 2 * kChunk;
        jit_wave_sync();  // (the scratch's previous user is done)
        jit_row_put(Xr, X.lane, x);
        if (GAIN_ROW) jit_row_put(G, X.lane, gain);
        const uint32_t first = (tb + 1u) % L, window = L < (uint32_t)kChunk ? L : (uint32_t)kChunk;
        float *ring = A.rings + (size_t)X.inst * (size_t)A.ring_samples + (size_t)ring_base;
        for (uint32_t p = X.lane; p < window; p += 64) {
            uint32_t s_ = first + p;
            if (s_ >= L) s_ -= L;
            R[p] = ring[s_];
        }
        jit_wave_sync();
        const uint32_t step = L < 64u ? L : 64u;
        uint32_t p = X.lane;  // (base + lane) mod L, kept incrementally
        for (uint32_t base = 0; base < (uint32_t)kChunk; base += step) {
            const uint32_t t = base + X.lane;
            if (X.lane < step && t < (uint32_t)kChunk) {
                const float xin = Xr[t], was = R[p];
                const double g = (double)(GAIN_ROW ? G[t] : gain[0]);
                float now, y;
                if (KIND == OP_FIXED_DELAY) { y = was; now = xin; }
                else if (KIND == OP_COMB_FILTER) { y = was; now = (float)((double)xin + (double)was * g); }
                else {
                    now = (float)((double)xin + (double)was * g);
                    y = (float)((double)was - (double)xin * g);
                }
                R[p] = now;
                Xr[t] = y;
            }
            p += step;
            if (p >= L) p -= L;
            jit_wave_sync();
        }
        if (X.live)
            for (uint32_t q = X.lane; q < window; q += 64) {
                uint32_t s_ = first + q;
                if (s_ >= L) s_ -= L;
                ring[s_] = R[q];
            }
        tb = (tb + (uint32_t)kChunk) % L;
        jit_row_get(Xr, X.lane, out);
    }
};

// AHD (AHD.js:35-76): a per-sample state machine over three slopes, on lane 0.  scr: row 0 output, rows 1-3 attack / hold / decay
// where they are signals.  A state outside 0..3 leaves the outlet's samples as the previous chunk had them.
struct JitAHD {
    int stage;
    bool playing;
    double t;
    float prev[4];  // this lane's samples of the previous chunk
    __device__ __forceinline__ void begin(const JitArgs &A, int state_slot) {
        stage = (int)jit_u((uint32_t)(int)A.init_state[state_slot]);
        playing = jit_u(A.init_state[state_slot + 1] != 0.0);
        t = jit_u(A.init_state[state_slot + 2]);
        prev[0] = prev[1] = prev[2] = prev[3] = 0.f;
    }
    template <bool ROW_A, bool ROW_H, bool ROW_D>
    __device__ __forceinline__ void tick(const JitCtx &X, float *scr, double period, const float (&att)[4], const float (&hold)[4], const float (&dec)[4],
                                         float (&out)[4]) {
        // Constant times (the usual envelope): the whole chunk in closed form unless a stage ends inside it.  `t += c` rounds at
        // every sample; repeat_add / linear_run take any number of those steps at once, bit for bit (repeat_add.hpp), and t only
        // grows, so "t after 256 steps is still below 1" says that no sample of the chunk leaves the stage.
        if (!ROW_A && !ROW_H && !ROW_D) {
            bool closed = stage == 0 || !playing;
            double c = 0.0, t_end = t;
            if (!closed) {
                c = period / (double)(stage == 1 ? att[0] : stage == 2 ? hold[0] : dec[0]);
                if (c > 0.0 && c < 1.0e300 && t >= 0.0) {
                    t_end = repeat_add(t, c, kChunk);
                    closed = t_end < 1.0;
                }
            }
            if (closed) {
                double tk[4];
                if (stage == 0 || !playing) tk[0] = tk[1] = tk[2] = tk[3] = t;
                else {
                    long long T, ce;
                    int K;
                    if (linear_run(t, c, kChunk, T, ce, K)) {
                        long long Tl = T + (long long)(X.lane * 4) * ce;
#pragma unroll
                        for (int i = 0; i < 4; ++i, Tl += ce) tk[i] = ldexp((double)Tl, K - 52);
                    } else {
                        double tt = repeat_add(t, c, (uint64_t)X.lane * 4);
#pragma unroll
                        for (int i = 0; i < 4; ++i, tt = tt + c) tk[i] = tt;
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) out[i] = stage == 1 ? (float)tk[i] : stage == 2 ? 1.f : stage == 3 ? (float)(1.0 - tk[i]) : 0.f;
                t = jit_u(t_end);
#pragma unroll
                for (int i = 0; i < 4; ++i) prev[i] = out[i];
                return;
            }
        }
        float *Y = scr;
        jit_wave_sync();
        jit_row_put(Y, X.lane, prev);
        if (ROW_A) jit_row_put(scr + kChunk, X.lane, att);
        if (ROW_H) jit_row_put(scr + 2 * kChunk, X.lane, hold);
        if (ROW_D) jit_row_put(scr + 3 * kChunk, X.lane, dec);
        jit_wave_sync();
        if (X.lane == 0) {
            int st = stage;
            bool pl = playing;
            double tt = t;
            for (int k = 0; k < kChunk; ++k) {
                if (st == 1) {
                    Y[k] = (float)tt;
                    if (pl) { tt += period / (double)(ROW_A ? scr[kChunk + k] : att[0]); if (tt >= 1.0) { ++st; tt -= 1.0; } }
                } else if (st == 2) {
                    Y[k] = 1.f;
                    if (pl) { tt += period / (double)(ROW_H ? scr[2 * kChunk + k] : hold[0]); if (tt >= 1.0) { ++st; tt -= 1.0; } }
                } else if (st == 3) {
                    Y[k] = (float)(1.0 - tt);
                    if (pl) { tt += period / (double)(ROW_D ? scr[3 * kChunk + k] : dec[0]); if (tt >= 1.0) { st = 0; pl = false; } }
                } else if (st == 0)
                    Y[k] = 0.f;
            }
            stage = st;
            playing = pl;
            t = tt;
        }
        stage = (int)jit_u((uint32_t)stage);
        playing = jit_u(playing);
        t = jit_u(t);
        jit_wave_sync();
        jit_row_get(Y, X.lane, out);
#pragma unroll
        for (int c = 0; c < 4; ++c) prev[c] = out[c];
    }
    __device__ __forceinline__ void end(const JitArgs &A, const JitCtx &X, int state_slot) const {
        double *st = A.state + (size_t)state_slot * A.n_pad + X.inst;
        st[0] = (double)stage;
        st[A.n_pad] = playing ? 1.0 : 0.0;
        st[(size_t)2 * A.n_pad] = t;
    }
};

// SampleRateRedux (SampleRateRedux.js:21-38): sample & hold with a counter, on lane 0.  scr: row 0 output, row 1 input, row 2 amount.
struct JitSRR {
    double since;
    float held;
    __device__ __forceinline__ void begin(const JitArgs &A, int state_slot) {
        since = jit_u(A.init_state[state_slot]);
        held = jit_u((float)A.init_state[state_slot + 1]);
    }
    template <bool ROW_IN, bool ROW_AMT>
    __device__ __forceinline__ void tick(const JitCtx &X, float *scr, const float (&in)[4], const float (&amt)[4], float (&out)[4]) {
        float *Y = scr;
        jit_wave_sync();
        if (ROW_IN) jit_row_put(scr + kChunk, X.lane, in);
        if (ROW_AMT) jit_row_put(scr + 2 * kChunk, X.lane, amt);
        jit_wave_sync();
        // A constant amount A: the counter runs 1, 2, .. and the input is taken whenever it exceeds A, i.e. every P = floor(A) + 1
        // samples — every lane finds the sample its four outputs hold, nobody walks.  (A counter that is not a whole number, or
        // a period beyond the integers f64 counts exactly, takes the walk below.)
        if (!ROW_AMT) {
            const double A = (double)amt[0], P = A >= 1.0 ? floor(A) + 1.0 : 1.0;
            if (since >= 0.0 && since == floor(since) && since < 4.0e15 && (A != A || P < 4.0e15)) {
                // first sample of the chunk that takes the input: the counter there is since + t + 1
                const double first = A != A ? 1.0e9 : fmax(0.0, floor(A) - since);  // (`x > NaN` never holds)
                const int t1 = first < (double)kChunk ? (int)first : kChunk, period = P < (double)kChunk ? (int)P : kChunk;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int t = (int)X.lane * 4 + c;
                    const int src = t < t1 ? -1 : t1 + ((t - t1) / period) * period;
                    out[c] = src < 0 ? held : (ROW_IN ? scr[kChunk + src] : in[0]);
                }
                if (t1 < kChunk) {
                    const int last = t1 + ((kChunk - 1 - t1) / period) * period;
                    held = jit_u(ROW_IN ? scr[kChunk + last] : in[0]);
                    since = jit_u((double)(kChunk - 1 - last));
                } else
                    since = jit_u(since + (double)kChunk);
                jit_wave_sync();  // (the next tick's rows may not overtake these reads)
                return;
            }
        }
        if (X.lane == 0) {
            double sn = since;
            float hd = held;
            for (int k = 0; k < kChunk; ++k) {
                sn += 1.0;
                if (sn > (double)(ROW_AMT ? scr[2 * kChunk + k] : amt[0])) { hd = ROW_IN ? scr[kChunk + k] : in[0]; sn = 0.0; }
                Y[k] = hd;
            }
            since = sn;
            held = hd;
        }
        since = jit_u(since);
        held = jit_u(held);
        jit_wave_sync();
        jit_row_get(Y, X.lane, out);
    }
    __device__ __forceinline__ void end(const JitArgs &A, const JitCtx &X, int state_slot) const {
        double *st = A.state + (size_t)state_slot * A.n_pad + X.inst;
        st[0] = since;
        st[A.n_pad] = (double)held;
    }
};

// MultiChannelOsc (MultiChannelOsc.js:21-38): `phase += f; phase %= sr` WITHOUT the Osc's `if (phase < 0) phase += sr`: the remainder
// keeps the dividend's sign, so the phase is not a modular sum in general — while nothing is negative it is (every increment of the
// chunk in [0, sr) on the 2^-36 grid, the start phase too: the Osc's exact fixed-point scan), else the 256 phases come from lane 0,
// in f64 exactly as the reference adds them.  Lookups (from L2, like the interpreter kernel) are lane-parallel.  scr: 512 floats.
struct JitMultiOsc {
    double phase;  // uniform
    __device__ __forceinline__ void begin(const JitArgs &A, int state_slot) { phase = jit_u(A.init_state[state_slot]); }
    template <int TF, int FORM>  // where the table comes from on the exact path (jit_pair)
    __device__ __forceinline__ void tick(const JitCtx &X, float *scr, const float *gtab, const float (&f)[4], float (&out)[4]) {
        double ph0 = phase;
        ph0 = (ph0 != ph0 || ph0 == 0.0) ? 0.0 : ph0;  // `this.phase[c] = this.phase[c] || 0`
        bool grid = ph0 >= 0.0 && ph0 < X.srd && ph0 * kJ36 == floor(ph0 * kJ36);
        long long q[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            double fd = (double)f[c];
            grid = grid && fd >= 0.0 && fd < X.srd;  // (below sampleRate: phase + f stays under 2^17 + 2^17, exact in f64 on this grid)
            if (!(fd >= 0.0 && fd < X.srd)) fd = 0.0;
            const double scaled = fd * kJ36;
            grid = grid && scaled == floor(scaled);
            q[c] = (long long)scaled;
        }
        if (__all(grid)) {
            const long long total = q[0] + q[1] + q[2] + q[3];
            const long long incl = jit_wave_scan(total);
            const long long before = (long long)(unsigned long long)(ph0 * kJ36) + (incl - total);
            unsigned long long P = mod_u64((unsigned long long)(before + q[0]) + X.lift, X.S, X.inv_S);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c > 0) {
                    P += (unsigned long long)q[c];
                    if (P >= X.S) P -= X.S;
                }
                const uint32_t idx = (uint32_t)(P >> kJFrac);
                const double fraction = (double)(P & kJMask) * (1.0 / kJ36);
                float ta, tb;
                jit_pair<TF, FORM>(X, gtab, idx, ta, tb);  // (idx < sampleRate: the entry after it exists; times a zero fraction it adds +0 like T[idx] would)
                out[c] = (float)((double)ta * (1.0 - fraction) + (double)tb * fraction);
            }
            const unsigned long long lastP = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(P >> 32), 63) << 32) |
                                             (uint32_t)__builtin_amdgcn_readlane((uint32_t)P, 63);
            phase = (double)lastP * (1.0 / kJ36);
            return;
        }
        double *T = (double *)scr;
        jit_wave_sync();
#pragma unroll
        for (int c = 0; c < 4; ++c) T[X.lane * 4 + c] = (double)f[c];
        jit_wave_sync();
        if (X.lane == 0) {
            double ph = ph0;
            for (int k = 0; k < kChunk; ++k) {
                double p = ph + T[k];
                if (fabs(p) >= X.srd) p = (p > 0.0 && p < 2.0 * X.srd) ? p - X.srd : (p < 0.0 && p > -2.0 * X.srd) ? p + X.srd : fmod(p, X.srd);
                T[k] = ph = p;
            }
            phase = ph;
        }
        phase = jit_u(phase);
        jit_wave_sync();
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double ph = T[X.lane * 4 + c];
            if (!(ph >= 0.0 && ph <= X.srd)) { out[c] = __builtin_nanf(""); continue; }  // typed-array[NaN / negative] is undefined
            const double lo = floor(ph), fraction = ph - lo;
            const int idx = (int)lo;
            out[c] = (float)((double)gtab[idx] * (1.0 - fraction) + (double)gtab[fraction != 0.0 ? idx + 1 : idx] * fraction);
        }
    }
};

// ---- CircleBuffer nodes with an unconnected offset (CircleBufferReader.js:12-25, CircleBufferWriter.js:12-25, CircleBuffer.js:15-34): the
// node's 256 accesses of a chunk are 256 consecutive slots of the ring — index = floor((T + t -+ sr offset) % len), negatives
// wrapped — one per sample, lane-parallel.  Several nodes share a ring and tick one after another, so each node first waits for
// the wave's earlier ring traffic.  T: the node's private sample counter.
struct JitCBNode {
    double T;  // uniform
    __device__ __forceinline__ void begin(const JitArgs &A, int state_slot) { T = jit_u(A.init_state[state_slot]); }
    __device__ __forceinline__ bool window(const JitCtx &X, bool reader, float off, int64_t len, int64_t &base) const {
        const double origin = reader ? T - X.srd * (double)off : T + X.srd * (double)off;
        const bool ok = fabs(origin) < 9.0e15;  // NaN / Inf offsets read `undefined` and write nowhere
        base = 0;
        if (ok) {
            base = (int64_t)fmod(floor(origin), (double)len);
            if (base < 0) base += len;
        }
        return ok;
    }
    template <bool WIPE>
    __device__ __forceinline__ void read(const JitArgs &A, const JitCtx &X, int64_t ring_base, int64_t len, float off, float (&out)[4]) {
        int64_t base;
        const bool ok = window(X, true, off, len, base);
        float *ring = A.rings + (size_t)X.inst * (size_t)A.ring_samples + (size_t)ring_base;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            int64_t idx = base + X.lane * 4 + c;
            if (idx >= len) idx -= len;
            out[c] = ok ? ring[idx] : __builtin_nanf("");
            if (ok && WIPE && X.live) ring[idx] = 0.f;  // postWipe
        }
        T += (double)kChunk;
    }
    template <bool WIPE, bool MIX>
    __device__ __forceinline__ void write(const JitArgs &A, const JitCtx &X, int64_t ring_base, int64_t len, float off, const float (&x)[4]) {
        int64_t base;
        const bool ok = window(X, false, off, len, base);
        float *ring = A.rings + (size_t)X.inst * (size_t)A.ring_samples + (size_t)ring_base;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        if (ok && X.live && (WIPE || MIX)) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                int64_t idx = base + X.lane * 4 + c;
                if (idx >= len) idx -= len;
                float v = WIPE ? 0.f : ring[idx];  // preWipe
                if (MIX) v = v + x[c];
                ring[idx] = v;
            }
        }
        T += (double)kChunk;
    }
};

// ---- Ordered slot operations: ring units whose accesses can land anywhere — a Delay with a signal-rate or sub-chunk delay
// (Delay.js:26-40), MonoDelay (MonoDelay.js:16-30), ReadBackDelay (ReadBackDelay.js:24-44), CircleBuffer nodes with a signal-rate
// offset or a ring shorter than a chunk.  The reference walks the chunk sample by sample — read (and clear) one slot, add a tap to
// each of two others, ... — in f32, so the ORDER of the operations on one slot matters, while operations on different slots
// commute.  Each lane owns four samples = up to twelve slot operations, keyed 3 t + j in the reference's order.  Rounds: every
// pending operation bids for its slot with its key (ds_min_u32 on a 1024-entry table in the wave's scratch, indexed by
// slot mod 1024: two slots sharing an entry only cost extra rounds), the lowest key of each entry performs its operation on the
// ring in HBM, and so on until nothing is pending — three rounds for a steady delay; whatever the modulation does, the result is
// the reference's.  KIND: the unit's opcode.  p0: the signal (delay lines) or the offset (CircleBuffer nodes); p1: the delay /
// the writer's input.  T: carried state — Delay: the previous chunk's last input; the others: the unit's running sample count.
struct JitRingOps {
    double T;  // uniform
    __device__ __forceinline__ void begin(const JitArgs &A, int state_slot) { T = jit_u(A.init_state[state_slot]); }
    // Delay: the carried input sample is engine-internal (no descriptor holds it): a continued launch takes it from where the last one left it
    __device__ __forceinline__ void begin_delay(const JitArgs &A, const JitCtx &X, int state_slot) {
        T = jit_u(A.resume ? A.state[(size_t)state_slot * A.n_pad + X.inst] : A.init_state[state_slot]);
    }
    enum : int { RO_NONE = 0, RO_READ, RO_READ_CLEAR, RO_ADD, RO_STORE };
    template <int KIND, int ATTR, int OWN = 1024>  // OWN: entries of the slot-ownership table (a power of two; fewer only cost extra rounds)
    __device__ __forceinline__ void tick(const JitArgs &A, const JitCtx &X, uint32_t g, float *scr, int64_t ring_base, uint32_t len, const float (&p0)[4],
                                         const float (&p1)[4], float (&out)[4]) {
        constexpr bool is_delay = KIND == OP_DELAY, is_mono = KIND == OP_MONO_DELAY, is_readback = KIND == OP_READBACK_DELAY;
        constexpr bool is_reader = KIND == OP_CB_READER, is_writer = KIND == OP_CB_WRITER;
        constexpr bool writer_mixes = is_writer && !(ATTR & 2);
        const double dlen = (double)len;
        const uint32_t lane = X.lane;
        float *ring = A.rings + (size_t)X.inst * (size_t)A.ring_samples + (size_t)ring_base;
        uint32_t *own = (uint32_t *)scr;
        constexpr uint32_t kOwnMask = (uint32_t)OWN - 1u, kFree = 0xffffffffu;
        jit_wave_sync();
#pragma unroll
        for (int k = 0; k < OWN / 256; ++k) ((uint4 *)own)[lane + 64 * k] = uint4{kFree, kFree, kFree, kFree};
        // what operation j of a sample does
        constexpr int kind0 = is_delay ? RO_READ_CLEAR : is_mono ? RO_ADD : is_readback ? RO_STORE
                              : is_reader ? ((ATTR & 1) ? RO_READ_CLEAR : RO_READ)
                              : (ATTR & 1) ? RO_STORE : writer_mixes ? RO_ADD : RO_NONE;  // writer: preWipe then mix = store; mix alone = add
        constexpr int kind1 = is_delay || is_mono ? RO_ADD : is_readback ? RO_READ : RO_NONE;
        constexpr int kind2 = is_delay ? RO_ADD : is_mono ? RO_READ_CLEAR : RO_NONE;
        const int kind[3] = {kind0, kind1, kind2};
        const double T0 = T;
        const uint32_t tb0 = is_readback ? (uint32_t)(int64_t)fmod(T0, dlen) : (uint32_t)((A.clock0 + (uint64_t)g * kChunk) % (uint64_t)len);
        int32_t slot[4][3];
        double val[4][3];
        uint32_t pending = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t t = lane * 4 + c;
            const uint32_t tb = (tb0 + t) % len;
            out[c] = 0.f;
            if (is_reader || is_writer) {  // CircleBuffer.js:16-18: floor(t % len), negatives wrapped; NaN / Inf go nowhere
                const double at = is_reader ? T0 + (double)t - X.srd * (double)p0[c] : T0 + (double)t + X.srd * (double)p0[c];
                double m = (at >= 0.0 && at < dlen) ? at : fmod(at, dlen);
                m = floor(m);
                if (m < 0.0) m += dlen;
                const bool valid = m >= 0.0 && m < dlen;
                slot[c][0] = valid ? (int32_t)m : -1; val[c][0] = writer_mixes ? (double)p1[c] : 0.0;
                slot[c][1] = slot[c][2] = -1; val[c][1] = val[c][2] = 0.0;
                if (is_reader && !valid) out[c] = __builtin_nanf("");
            } else if (is_readback) {
                double r = (T0 + (double)t) - (double)p1[c] + dlen;
                r = (r >= 0.0 && r < dlen) ? r : fmod(r, dlen);
                const bool valid = r >= 0.0 && r < dlen && r == floor(r);  // a fractional or negative index reads `undefined`
                slot[c][0] = (int32_t)tb; val[c][0] = (double)p0[c];
                slot[c][1] = valid ? (int32_t)r : -1; val[c][1] = 0.0;
                slot[c][2] = -1; val[c][2] = 0.0;
                if (!valid) out[c] = __builtin_nanf("");
            } else {
                const double xin = (double)p0[c];
                double tWrite = (double)tb + (double)p1[c];
                if (!(tWrite >= 0.0 && tWrite < dlen))
                    tWrite = (tWrite >= dlen && tWrite < 2.0 * dlen) ? tWrite - dlen : fmod(tWrite, dlen);
                const double lo = floor(tWrite), frac = tWrite - trunc(tWrite);
                double hi = ceil(tWrite);
                if (is_mono && hi >= dlen) hi -= dlen;  // MonoDelay wraps the ceil tap, Delay drops it at index len
                const int32_t slo = (lo >= 0.0 && lo < dlen) ? (int32_t)lo : -1, shi = (hi >= 0.0 && hi < dlen) ? (int32_t)hi : -1;
                const double vlo = xin * (1.0 - frac), vhi = xin * frac;
                slot[c][0] = is_delay ? (int32_t)tb : slo; val[c][0] = is_delay ? 0.0 : vlo;
                slot[c][1] = is_delay ? slo : shi;          val[c][1] = is_delay ? vlo : vhi;
                slot[c][2] = is_delay ? shi : (int32_t)tb;  val[c][2] = is_delay ? vhi : 0.0;
            }
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (slot[c][j] >= 0 && kind[j] != RO_NONE) pending |= 1u << (c * 3 + j);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): nodes of one CircleBuffer tick one after another
        __builtin_amdgcn_wave_barrier();
        while (__any(pending != 0)) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if (pending & (1u << (c * 3 + j)))
                        __hip_atomic_fetch_min(&own[(uint32_t)slot[c][j] & kOwnMask], (lane * 4 + c) * 3 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            uint32_t won = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if ((pending & (1u << (c * 3 + j))) && own[(uint32_t)slot[c][j] & kOwnMask] == (lane * 4 + c) * 3 + j) won |= 1u << (c * 3 + j);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // the winners' slots are distinct: all their loads first (up to twelve L2 round trips in flight per lane), then the stores
            float was[4][3];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    was[c][j] = 0.f;
                    if (!(won & (1u << (c * 3 + j)))) continue;
                    own[(uint32_t)slot[c][j] & kOwnMask] = kFree;
                    if (kind[j] != RO_STORE) was[c][j] = ring[slot[c][j]];
                }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (!(won & (1u << (c * 3 + j)))) continue;
                    float *p = ring + slot[c][j];
                    if (kind[j] == RO_READ || kind[j] == RO_READ_CLEAR) {
                        out[c] = was[c][j];
                        if (kind[j] == RO_READ_CLEAR && X.live) *p = 0.f;
                    } else if (kind[j] == RO_STORE) {
                        if (X.live) *p = (float)val[c][j];
                    } else if (X.live)
                        *p = (float)((double)was[c][j] + val[c][j]);
                }
            pending &= ~won;
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this round's ring traffic has landed before the next one starts
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (is_delay) T = (double)__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(p0[3]), 63));
        else if (is_readback || is_reader || is_writer) T = T0 + (double)kChunk;
    }
};

// ---- Delay with a signal-rate delay (an LFO on the delay time: SimpleDelay, chorus, flanger), Delay.js:26-40, WITHOUT slot rounds
// where the taps allow it.  What the reference does to one ring slot during a chunk is a sequence in sample order: the `+=` of every
// tap that lands there, each rounded to f32, and the read-and-clear by the sample whose slot it is (slot s0 + t belongs to sample t).
// When the tap positions t + delay[t] do not decrease along the chunk (a delay that falls by less than a sample per sample: any
// audio-rate modulation short of a jump), never put more than two floor taps into one slot, and every tap lands ahead of its own
// sample (a delay of at least a sample), a slot's sequence is short and known from NEIGHBOURING samples alone:
//     [ceil taps of the one or two samples whose floor tap fell one slot below]  [floor taps of the one or two samples that fall here]
// (a tap with fraction 0 brings its `+= in * 0` along).  So the first sample of every such group OWNS its slot: it takes the slot's
// value, replays that sequence out of its own and its neighbours' registers (the lane below and above by DPP) and stores the result —
// into the ring, or, for a slot of this chunk's read window, into the wave's row of read values, because everything that lands on such
// a slot lands BEFORE its sample reads it; the last sample in front of a gap owns the slot that only receives ceil taps.  Nothing is
// bid for, nothing waits on another lane, every touched slot is read and written once.  A chunk that breaks a condition (a jump,
// a delay below one sample, a negative / NaN delay, taps wrapping onto the read window) takes the ordered slot operations
// (JitRingOps) instead — whatever the modulation does, the result is the reference's.
// scr: 512 floats per wave — the chunk's 256 read values; the slot operations' ownership table (512 words) takes the same space when
// a chunk needs them.
// MONO: MonoDelay (MonoDelay.js:16-28) — the same taps, except that a ceil tap at index `length` wraps to slot 0 instead of being dropped,
// and a sample's read comes behind its own taps (which, with a delay of a sample at least, land elsewhere); no carried state.
template <bool MONO>
struct JitDelayGather {
    JitRingOps rounds;  // (holds T: the previous chunk's last input, what a Delay's state carries)
    __device__ __forceinline__ void begin(const JitArgs &A, const JitCtx &X, int state_slot) {
        if (MONO) rounds.T = 0.0;
        else rounds.begin_delay(A, X, state_slot);
    }
    static __device__ __forceinline__ float dpp_up(float mine, float edge) {  // the lane below's value; lane 0 gets `edge`
        return __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(edge), __float_as_uint(mine), 0x138, 0xf, 0xf, false));
    }
    static __device__ __forceinline__ float dpp_down(float mine, float edge) {  // the lane above's value; lane 63 gets `edge`
        return __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(edge), __float_as_uint(mine), 0x130, 0xf, 0xf, false));
    }
    static __device__ __forceinline__ double dpp_up(double mine) {  // (lane 0's value is never used: its neighbours below do not exist)
        const uint32_t lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)__double2loint(mine), 0x138, 0xf, 0xf, false);
        const uint32_t hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)__double2hiint(mine), 0x138, 0xf, 0xf, false);
        return __hiloint2double((int)hi, (int)lo);
    }
    static __device__ __forceinline__ double dpp_down(double mine) {
        const uint32_t lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)__double2loint(mine), 0x130, 0xf, 0xf, false);
        const uint32_t hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)__double2hiint(mine), 0x130, 0xf, 0xf, false);
        return __hiloint2double((int)hi, (int)lo);
    }
    __device__ __forceinline__ void tick(const JitArgs &A, const JitCtx &X, uint32_t g, float *scr, int64_t ring_base, uint32_t len, const float (&x)[4],
                                         const float (&dl)[4], float (&out)[4]) {
        const double dlen = (double)len;
        const uint32_t lane = X.lane;
        const uint32_t tb0 = (uint32_t)((A.clock0 + (uint64_t)g * kChunk) % (uint64_t)len);
        // where every sample's taps land, exactly as Delay.js:33-36 computes it.  Local index i = c + 2: samples 4 lane - 2 .. 4 lane + 5
        int32_t R[8];   // floor tap's slot, relative to the chunk's first slot (0 .. len - 1)
        double F[7];    // the tap's fraction
        float Xs[7];    // the input sample
        bool ok = len >= 2u * kChunk;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t t = lane * 4 + c;
            uint32_t tb = tb0 + t;
            if (tb >= len) tb -= len;
            double tW = (double)tb + (double)dl[c];
            if (!(tW >= 0.0 && tW < dlen)) tW = (tW >= dlen && tW < 2.0 * dlen) ? tW - dlen : fmod(tW, dlen);
            const double lo = floor(tW);
            F[c + 2] = tW - trunc(tW);
            const bool valid = lo >= 0.0 && lo < dlen;
            int32_t r = valid ? (int32_t)lo - (int32_t)tb0 : -1;
            if (valid && r < 0) r += (int32_t)len;
            R[c + 2] = r;
            Xs[c + 2] = x[c];
            ok = ok && valid && r > (int32_t)t;  // (ahead of its own sample: whatever lands in the read window lands before the slot is read)
        }
        // the neighbours: two samples of the lane below, two of the lane above (one for the inputs)
        R[0] = __builtin_amdgcn_update_dpp(-9, R[4], 0x138, 0xf, 0xf, false);
        R[1] = __builtin_amdgcn_update_dpp(-9, R[5], 0x138, 0xf, 0xf, false);
        R[6] = __builtin_amdgcn_update_dpp(0x7ffffff0, R[2], 0x130, 0xf, 0xf, false);
        R[7] = __builtin_amdgcn_update_dpp(0x7ffffff0, R[3], 0x130, 0xf, 0xf, false);
        Xs[0] = dpp_up(Xs[4], 0.f);
        Xs[1] = dpp_up(Xs[5], 0.f);
        Xs[6] = dpp_down(Xs[2], 0.f);
        F[0] = dpp_up(F[4]);
        F[1] = dpp_up(F[5]);
        F[6] = dpp_down(F[2]);
        // non-decreasing along the chunk, at most two floor taps per slot, every relative slot a ring slot of its own?
        ok = ok && R[2] >= R[1] && R[3] >= R[2] && R[4] >= R[3] && R[5] >= R[4] && R[4] > R[2] && R[5] > R[3] && R[6] > R[4] && R[7] > R[5];
        const int32_t top = __builtin_amdgcn_readlane(R[5], 63);
        ok = ok && (uint32_t)(top + 2) <= len;
        if (!__all(ok)) {
            rounds.tick<MONO ? OP_MONO_DELAY : OP_DELAY, 0, 512>(A, X, g, scr, ring_base, len, x, dl, out);
            return;
        }
        float *ring = A.rings + (size_t)X.inst * (size_t)A.ring_samples + (size_t)ring_base;
        float *OUT = scr;  // [256]: what the chunk's samples read
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (the last chunk's ring stores stay in front of these loads: a wave's accesses to one address keep their order)
        // the chunk's read window as the ring holds it; slots beyond it that this lane's samples own, fetched at once
        float old[4];
        uint32_t a0 = tb0 + lane * 4;
        if (a0 >= len) a0 -= len;
        const bool quad = a0 + 4u <= len && (a0 & 3u) == 0u && (((uintptr_t)ring) & 15u) == 0u;
        if (quad) {
            const f32x4 v = *(const f32x4 *)(ring + a0);
            old[0] = v[0]; old[1] = v[1]; old[2] = v[2]; old[3] = v[3];
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                uint32_t a = a0 + c;
                if (a >= len) a -= len;
                old[c] = ring[a];
            }
        }
        float far_floor[4], far_ceil[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = c + 2;
            uint32_t a = tb0 + (uint32_t)R[i];
            if (a >= len) a -= len;
            far_floor[c] = (R[i] != R[i - 1] && R[i] >= kChunk) ? ring[a] : 0.f;       // this sample owns its floor tap's slot, beyond the window
            uint32_t a1 = a + 1u;
            if (a1 >= len) a1 -= len;
            far_ceil[c] = (R[i + 1] > R[i] + 1 && R[i] + 1 >= kChunk) ? ring[a1] : 0.f;  // ... and the slot behind it, which only ceil taps reach
        }
        jit_wave_sync();  // (the scratch's previous user is done)
        jit_row_put(OUT, lane, old);
        jit_wave_sync();
        auto add = [](float acc, double v) __attribute__((always_inline)) { return (float)((double)acc + v); };
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = c + 2;
            const int32_t r = R[i];
            uint32_t a = tb0 + (uint32_t)r;
            if (a >= len) a -= len;
            if (r != R[i - 1]) {  // first of its group: the slot's whole sequence
                float acc = r < kChunk ? OUT[r] : far_floor[c];
                const bool below2 = R[i - 2] == r - 1, below1 = R[i - 1] == r - 1;  // ceil taps of the samples one slot below (at index `length` — slot 0 — they are dropped)
                if (below2 && F[i - 2] != 0.0 && (MONO || a != 0u)) acc = add(acc, (double)Xs[i - 2] * F[i - 2]);
                if (below1 && F[i - 1] != 0.0 && (MONO || a != 0u)) acc = add(acc, (double)Xs[i - 1] * F[i - 1]);
                acc = add(acc, (double)Xs[i] * (1.0 - F[i]));                          // floor tap
                if (F[i] == 0.0) acc = add(acc, (double)Xs[i] * F[i]);                // ... and a ceil tap that lands on the same slot
                if (R[i + 1] == r) {                                                   // the group's second sample
                    acc = add(acc, (double)Xs[i + 1] * (1.0 - F[i + 1]));
                    if (F[i + 1] == 0.0) acc = add(acc, (double)Xs[i + 1] * F[i + 1]);
                }
                if (r < kChunk) OUT[r] = acc;
                else if (X.live) ring[a] = acc;
            }
            if (R[i + 1] > r + 1) {  // last of its group in front of a gap: the slot behind, reached by this group's ceil taps only
                uint32_t a1 = a + 1u;
                if (a1 >= len) a1 -= len;
                if ((MONO || a1 != 0u) && (F[i] != 0.0 || (R[i - 1] == r && F[i - 1] != 0.0))) {
                    float acc = r + 1 < kChunk ? OUT[r + 1] : far_ceil[c];
                    if (R[i - 1] == r && F[i - 1] != 0.0) acc = add(acc, (double)Xs[i - 1] * F[i - 1]);
                    if (F[i] != 0.0) acc = add(acc, (double)Xs[i] * F[i]);
                    if (r + 1 < kChunk) OUT[r + 1] = acc;
                    else if (X.live) ring[a1] = acc;
                }
            }
        }
        jit_wave_sync();
        jit_row_get(OUT, lane, out);  // `out[t] = buf[tB]`
        if (X.live) {                 // `buf[tB] = 0` (Delay.js:28-29): the whole read window
            if (quad) *(f32x4 *)(ring + a0) = f32x4{0.f, 0.f, 0.f, 0.f};
            else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    uint32_t a = a0 + c;
                    if (a >= len) a -= len;
                    ring[a] = 0.f;
                }
            }
        }
        if (!MONO) rounds.T = (double)__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x[3]), 63));
    }
};

// ---- copy-out (src/renderChannelData.js:35-44): `x || 0`, then this lane's four samples of the outlet's channel
// FINITE: the generator has shown that this outlet cannot be NaN here (jit_codegen.hpp bound_of_buf): only -0 is left to fix
template <bool FINITE>
__device__ __forceinline__ void jit_store(const JitArgs &A, const JitCtx &X, uint32_t g, uint32_t oc, const float (&v)[4]) {
    if (!X.live || g < X.g_own || g >= X.g_stop) return;  // (a segment that warms up ticks chunks it does not own)
    const uint64_t n0 = X.n0(g);
    const float w[4] = {fix_out<FINITE>(v[0]), fix_out<FINITE>(v[1]), fix_out<FINITE>(v[2]), fix_out<FINITE>(v[3])};
    float *row = A.out + ((size_t)X.inst * A.n_out + oc) * A.n_samples + n0;
    // (whole chunks, all but a render's last, are known to be so without a look at the lane)
    if (A.vec4_ok && (uint64_t)(g + 1) * kChunk <= A.n_samples) store4<true>(row, w, n0, A.n_samples);
    else if (A.vec4_ok && n0 + 4 <= A.n_samples) store4<true>(row, w, n0, A.n_samples);
    else store4<false>(row, w, n0, A.n_samples);
}

// the same for an outlet a Filter scan writes: a chunk that took the scan (JitFilterScan::finite) holds no NaN, only -0 is left to fix
__device__ __forceinline__ void jit_store_scan(const JitArgs &A, const JitCtx &X, uint32_t g, uint32_t oc, const float (&v)[4], bool finite) {
    if (finite) jit_store<true>(A, X, g, oc, v);
    else jit_store<false>(A, X, g, oc, v);
}

// ---- host-computed signal (Noise): this lane's four samples of input stream `stream`
__device__ __forceinline__ void jit_input(const JitArgs &A, const JitCtx &X, uint32_t g, uint32_t stream, float (&out)[4]) {
    const float *src = A.inputs + ((size_t)stream * A.n_inst + X.inst) * A.n_samples;
    const uint64_t n0 = X.n0(g);
#pragma unroll
    for (int c = 0; c < 4; ++c) out[c] = n0 + c < A.n_samples ? src[n0 + c] : 0.f;
}

// Butterworth coefficients of Filter.js:66-84 (kind 0 = LP, 1 = HP): filter_lamda.hpp, shared by every engine
__device__ __forceinline__ void jit_filter_coefficients(int kind, double f, double sr, double (&k)[5]) { butterworth_coefficients(kind, f, sr, k); }

// ---- Filter (src/components/Filter.js:27-51) with an unconnected cutoff.  y = f32((P - b1 y1) - b2 y2) with
// P = (a0 x + a1 x1) + a2 x2 is a recurrence in y only.  P is lane-parallel: every lane computes its four samples' from its own
// inputs and the last two of the lane before it (one DPP shift each; in front of the chunk, the two inputs carried as
// scalars).  The recurrence is a dependent chain of five f64 operations per sample whatever the lane count, so the WAVES x R
// instances of a workgroup run theirs side by side on the lanes of ONE wave: in sub-blocks of SUB samples every wave parks
// its instances' P values (f64, out of the registers feed() left them in) in a shared LDS tile, ONE wave (`who`: the generated kernel lets waves 0 and 1 take turns) runs lane = instance
// over the rows — nothing but the chain, y written as f32 over the P values already consumed — and every wave picks its rows
// up again.  A row's coefficients live in "its" lane of EVERY wave (a function of the lane's instance alone), the two outputs before
// the sub-block in LDS behind the tile.
//   tile: rows of SUB + 2 doubles, row = wave R + r; then one word that says "given back"; then y1 / y2 of every row, stage by stage.
//   SUB = 256, 128 or 64 by what LDS holds next to the table image
typedef double f64x2 __attribute__((ext_vector_type(2)));
// MEM: byte offset from the tile's start of the word that says "given back" and, behind it, the stages' y1 / y2 (behind the rows of whichever
// kind of stage needs more of them: the stages of a circuit share the tile one after the other, their recurrence memories must not meet it)
template <int WAVES, int R, int SUB, int NA, int MEM>  // NA = R, or 1 when the cutoff is a constant of the circuit: one set of coefficients serves all
struct JitFilterK {
    static constexpr int kPitch = SUB + 2;  // doubles per row
    double a[NA][3];                        // this wave's instances: a0 a1 a2 (wave-uniform)
    float x1[R], x2[R];                     // the two inputs before the chunk: x1 as it was, x2 through `|| 0` (Filter.js:47-48)
    double k[5], lastF;                     // lane = row (the same in every wave): that instance's coefficients
    int stage;                              // which of the circuit's Filter stages this is (they share the tile, not the memory)
    // The recurrence's memory y1 / y2 lives in the row's two spare doubles of the tile, so that any wave can run a sub-block's
    // recurrences (the generated kernel alternates between waves 0 and 1: each does its other work while the other one serves).
#ifdef DUSP_JIT_PROFILE
    unsigned long long cyc_serial = 0;      // diagnostic build: cycles this wave spent inside serial()
#endif

    // fr: the cutoff of the instance THIS LANE serves when its wave runs the recurrences (a constant, or that instance's parameter: jit_row_param)
    // y1 / y2 of row `row` of this stage
    static __device__ __forceinline__ uint32_t flag_address(double *tile) { return row_address(tile, 0) + (uint32_t)MEM; }
    __device__ __forceinline__ uint32_t memory_address(double *tile, uint32_t row) const {
        return flag_address(tile) + 16u + ((uint32_t)stage * (WAVES * R) + row) * 16u;
    }
    __device__ __forceinline__ void begin(const JitArgs &A, const JitCtx &X, double *tile, int stage_, int kind, float fr, int state_slot) {
        stage = stage_;
        const double *is = A.init_state + state_slot;  // has_lastF lastF a0 a1 a2 b1 b2 x1 x2 y1 y2
        const double ft = (double)fr;
        if (is[0] == 0.0 || ft != is[1]) jit_filter_coefficients(kind, ft, X.srd, k);  // `if (this.f[t] != this.lastF)`
        else {
            k[0] = is[2]; k[1] = is[3]; k[2] = is[4]; k[3] = is[5]; k[4] = is[6];
        }
        lastF = ft;
        const uint32_t row_v = blockIdx.x * (WAVES * R) + X.lane, row_seg = row_v % A.n_seg;  // the (instance, segment) this lane serves
        if (X.wave == 0 && X.lane < WAVES * R) {  // (the first barrier of the chunk loop stands between this and the first reader)
            lds_double *mem = (lds_double *)(uintptr_t)memory_address(tile, X.lane);
            const bool rest = A.warm && row_seg != 0;  // (a segment that warms up starts its recurrence from rest)
            mem[0] = rest ? 0.0 : is[9];
            mem[1] = rest ? 0.0 : is[10];
        }
        // (what the render leaves of these is known now: written here, the feed-forward coefficients need no registers through the loop)
        const uint32_t inst = row_v / A.n_seg;
        if (X.wave == 0 && X.lane < WAVES * R && inst < A.n_inst && row_seg == 0) {
            double *st = A.state + (size_t)state_slot * A.n_pad + inst;
            st[0] = 1.0;
            st[A.n_pad] = lastF;
            for (int i = 0; i < 5; ++i) st[(size_t)(2 + i) * A.n_pad] = k[i];
        }
    }
    // f: the cutoff of the wave's instance in slot r
    __device__ __forceinline__ void begin_slot(const JitArgs &A, const JitCtx &X, int r, int kind, float f, int state_slot) {
        const double *is = A.init_state + state_slot;
        double kk[5];
        const double ft = (double)f;
        if (is[0] == 0.0 || ft != is[1]) jit_filter_coefficients(kind, ft, X.srd, kk);
        else {
            kk[0] = is[2]; kk[1] = is[3]; kk[2] = is[4];
        }
        if (r < NA) {
            a[r][0] = jit_u(kk[0]); a[r][1] = jit_u(kk[1]); a[r][2] = jit_u(kk[2]);
        }
        const bool rest = A.warm && X.seg != 0;
        x1[r] = rest ? 0.f : jit_u((float)is[7]);  // (inputs are f32 samples: nothing is lost)
        x2[r] = rest ? 0.f : jit_u((float)is[8]);
    }
    // Segments that warm up (JitArgs::warm), at the top of chunk g: what this stage holds for row `lane` when that row's own chunks begin and
    // when they end goes into the records the host checks; the render's last segment leaves the unit's state there as well.
    __device__ __forceinline__ void capture(const JitArgs &A, const JitCtx &X, double *tile, uint32_t g, int state_slot) const {
        if (!A.warm || X.wave != 0 || X.lane >= WAVES * R) return;
        const uint32_t v = blockIdx.x * (WAVES * R) + X.lane, n_virtual = A.n_inst * A.n_seg;
        if (v >= n_virtual) return;
        const uint32_t seg = v % A.n_seg, inst = v / A.n_seg, own = seg * A.seg_groups, stop = min(own + A.seg_groups, A.n_groups);
        // (the ROW's chunk, not the calling wave's: every wave of the workgroup is in the same iteration, each on the chunks of its own segment)
        const uint32_t row_g = (seg ? own - A.seg_groups : 0u) + (g - X.g_begin);
        const lds_double *mem = (const lds_double *)(uintptr_t)memory_address(tile, X.lane);
        double *rec = A.warm_records + ((size_t)stage * n_virtual + v) * 8;
        if (row_g == own) rec[0] = mem[0], rec[1] = mem[1];
        if (row_g == stop) {
            rec[2] = mem[0], rec[3] = mem[1];
            if (seg == A.n_seg - 1) {
                double *st = A.state + (size_t)state_slot * A.n_pad + inst;
                st[(size_t)9 * A.n_pad] = mem[0];
                st[(size_t)10 * A.n_pad] = mem[1];
            }
        }
    }
    __device__ __forceinline__ void capture_slot(const JitArgs &A, const JitCtx &X, int r, uint32_t g, int state_slot) const {
        if (!A.warm || !X.live || X.lane != 0 || g != X.g_stop) return;
        double *rec = A.warm_records + ((size_t)stage * (A.n_inst * A.n_seg) + (X.inst * A.n_seg + X.seg)) * 8;
        rec[4] = (double)x1[r], rec[5] = (double)x2[r];
        if (X.seg == A.n_seg - 1) {
            double *st = A.state + (size_t)state_slot * A.n_pad + X.inst;
            st[(size_t)7 * A.n_pad] = (double)x1[r];
            st[(size_t)8 * A.n_pad] = (double)x2[r];
        }
    }
    static __device__ __forceinline__ float or0f(float v) { return (v != v || v == 0.f) ? 0.f : v; }
    // The feed-forward half of slot r's chunk, P[t] = (a0 x[t] + a1 (x[t-1] || 0)) + a2 (x[t-2] || 0) in f64 with the reference's
    // order of roundings (Filter.js:40-42): lane l its samples 4l .. 4l+3.  Leaves the chunk's last two inputs for the next chunk.
    __device__ __forceinline__ void feed(const JitCtx &X, int r, const float (&x)[4], double (&p)[4]) {
        // the lane before this one's x[3] and x[2]; lane 0 gets the carried pair (DPP wave_shr:1, `old` stays where no lane shifts in)
        const float l1 = __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(x1[r]), __float_as_uint(x[3]), 0x138, 0xf, 0xf, false));
        const float l2 = __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(x2[r]), __float_as_uint(x[2]), 0x138, 0xf, 0xf, false));
        const double o[5] = {(double)or0f(l2), (double)or0f(l1), (double)or0f(x[0]), (double)or0f(x[1]), (double)or0f(x[2])};
#pragma unroll
        for (int c = 0; c < 4; ++c) p[c] = (a[r % NA][0] * (double)x[c] + a[r % NA][1] * o[c + 1]) + a[r % NA][2] * o[c];
        x1[r] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x[3]), 63));
        x2[r] = or0f(__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x[2]), 63)));
    }
    // sub-block s of slot r: the lanes whose samples it holds put their four P values into the wave's row
    static __device__ __forceinline__ void park(const JitCtx &X, double *tile, int r, int s, const double (&p)[4]) {
        const int first = s * (SUB / 4);
        if ((int)X.lane >= first && (int)X.lane < first + SUB / 4) {
            f64x2 *row = (f64x2 *)(tile + (size_t)(X.wave * R + r) * kPitch) + ((int)X.lane - first) * 2;
            row[0] = f64x2{p[0], p[1]};
            row[1] = f64x2{p[2], p[3]};
        }
    }
    static __device__ __forceinline__ void pick(const JitCtx &X, const double *tile, int r, int s, float (&out)[4]) {
        const int first = s * (SUB / 4);
        if ((int)X.lane >= first && (int)X.lane < first + SUB / 4) {
            const f32x4 y = ((const f32x4 *)(tile + (size_t)(X.wave * R + r) * kPitch))[(int)X.lane - first];
            out[0] = y[0]; out[1] = y[1]; out[2] = y[2]; out[3] = y[3];
        }
    }
    // PB (8 or 4) steps of the recurrence on P values held in registers, WITHOUT the `|| 0` selects of Filter.js:42-46
    template <int PB, typename DST>
    __device__ __forceinline__ void block(const double (&pv)[PB], double &u1, double &u2, DST *dst) const {
        const double b1 = k[3], b2 = k[4];
        f32x4 y4[PB / 4];
#pragma unroll
        for (int i = 0; i < PB; ++i) {
#ifdef DUSP_FILTER_FMA
            // EXPERIMENT (DUSP_FILTER_FMA=1, off by default): b2 y2 leaves P a step early, b1 y1 goes in by one fused multiply-add — the
            // dependent chain is fma, cvt, cvt instead of mul, sub, sub, cvt, cvt; the sum is rounded once where Filter.js:40-46 rounds twice
            const float y = (float)fma(-b1, u1, pv[i] - b2 * u2);
#else
            const float y = (float)((pv[i] - b1 * u1) - b2 * u2);
#endif
            y4[i >> 2][i & 3] = y;
            u2 = u1;
            u1 = (double)y;
        }
#pragma unroll
        for (int i = 0; i < PB / 4; ++i) dst[i] = y4[i];
    }
    typedef __attribute__((address_space(3))) double lds_double;
    typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    // a row's LDS address as a masked integer: known non-negative, so that element offsets fold into the ds instructions' immediates
    static __device__ __forceinline__ uint32_t row_address(double *tile, uint32_t row) {
        return ((uint32_t)(uintptr_t)(lds_double *)(tile + (size_t)row * kPitch)) & 0x3ffffu;
    }
    // Between two barriers: SUB samples of the recurrence of every row — one wave, up to 64 rows on its lanes.  A step's chain
    // is mul, sub, sub, cvt, cvt (26-30 cycles by itself) and the wave issues in order, so whatever else sits in the loop adds
    // its issue time (tools/serialbench.hip: an LDS read of two doubles 13 cycles, a 16-byte store 23, a test-and-branch per
    // block 45): P comes from LDS in blocks of 8 whose reads are issued a block ahead (two register sets, no copies), offsets
    // in the instructions' immediate fields; y overwrites P values that are in registers already; and the loop tests nothing.
    //
    // The `|| 0` selects of Filter.js:42-46 on y1 / y2 are speculated away.  Without them a NaN never leaves the recurrence (it
    // comes back through b1 y1 whatever b1 is), so the LAST output of the sub-block tells whether any of its outputs was one;
    // short of a NaN the selects only turn -0 into +0, which can only flip the sign of a later zero, and every consumer maps
    // that to +0 (see loop2_engine.hip).  A sub-block that met a NaN in some row is given back: the word after row 0 says so,
    // y1 / y2 stay as they were, every wave parks its rows again (failed()) and serial_exact() does the sub-block as written.
    template <int PB>  // 8, or 4 where the kernel is short of registers (two sets of PB doubles)
    __device__ __forceinline__ void serial(const JitCtx &X, double *tile, uint32_t who) {
        if (X.wave != who || X.lane >= WAVES * R) return;
#ifdef DUSP_JIT_PROFILE
        const unsigned long long stamp0 = __builtin_readcyclecounter();
#endif
        const uint32_t row = row_address(tile, X.lane);
        const lds_double *pr = (const lds_double *)(uintptr_t)row;
        lds_f32x4 *yr = (lds_f32x4 *)(uintptr_t)row;
        lds_double *mem = (lds_double *)(uintptr_t)memory_address(tile, X.lane);
        double u1 = jit_or0(mem[0]), u2 = jit_or0(mem[1]);
        double pa[PB], pb[PB];
#pragma unroll
        for (int i = 0; i < PB; ++i) pa[i] = pr[i];
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the loop is entered with its first block here, as every later iteration finds it
        for (int t0 = 0; t0 < SUB; t0 += 2 * PB) {
#pragma unroll
            for (int i = 0; i < PB; ++i) pb[i] = pr[PB + i];
            __builtin_amdgcn_sched_barrier(0);
            block<PB>(pa, u1, u2, yr);  // y of samples t0 .. t0+PB-1 goes where P of the samples from t0/2 on stood (P up to t0+2PB-1 is in registers)
            const int next = t0 + 2 * PB < SUB ? 2 * PB : 0;  // (the last block reads itself again: a branch around the reads would make every wait conservative)
#pragma unroll
            for (int i = 0; i < PB; ++i) pa[i] = pr[next + i];
            __builtin_amdgcn_sched_barrier(0);
            block<PB>(pb, u1, u2, yr + PB / 4);
            pr += 2 * PB;
            yr += PB / 2;
        }
        const bool met_nan = __builtin_amdgcn_ballot_w64(!(u1 == u1)) != 0;  // (some row's: the sub-block is given back whole)
        if (X.lane == 0) *(lds_u32 *)(uintptr_t)flag_address(tile) = met_nan ? 1u : 0u;
        if (!met_nan) {
            mem[0] = u1;
            mem[1] = u2;
        }
#ifdef DUSP_JIT_PROFILE
        cyc_serial += __builtin_readcyclecounter() - stamp0;
#endif
    }
    // after the barrier that ends serial(), every wave: was the sub-block given back?
    static __device__ __forceinline__ bool failed(double *tile) {
        return __builtin_amdgcn_readfirstlane((int)*(const lds_u32 *)(uintptr_t)flag_address(tile)) != 0;
    }
    // the sub-block as Filter.js:40-46 writes it, on freshly parked rows
    __device__ __forceinline__ void serial_exact(const JitCtx &X, double *tile, uint32_t who) {
        if (X.wave != who || X.lane >= WAVES * R) return;
        const uint32_t row = row_address(tile, X.lane);
        const lds_double *pr = (const lds_double *)(uintptr_t)row;
        lds_f32x4 *yr = (lds_f32x4 *)(uintptr_t)row;
        const double b1 = k[3], b2 = k[4];
        lds_double *mem = (lds_double *)(uintptr_t)memory_address(tile, X.lane);
        double y1 = mem[0], y2 = mem[1];
        for (int t0 = 0; t0 < SUB; t0 += 4) {
            double pv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) pv[i] = pr[t0 + i];
            f32x4 y4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float y = (float)((pv[i] - b1 * jit_or0(y1)) - b2 * jit_or0(y2));
                y4[i] = y;
                y2 = jit_or0(y1);
                y1 = (double)y;
            }
            yr[t0 >> 2] = y4;  // (over P of samples t0/2 and t0/2+1: read already)
        }
        mem[0] = y1;
        mem[1] = y2;
    }
    // state write-back: the input history by each wave (slot r), coefficients and recurrence memory by the lanes that hold them
    __device__ __forceinline__ void end_slot(const JitArgs &A, const JitCtx &X, int r, int state_slot) const {
        if (!X.live || X.lane != 0 || A.warm) return;  // (warm: capture_slot has written it)
        double *st = A.state + (size_t)state_slot * A.n_pad + X.inst;
        st[(size_t)7 * A.n_pad] = (double)x1[r];
        st[(size_t)8 * A.n_pad] = (double)x2[r];
    }
    __device__ __forceinline__ void end(const JitArgs &A, const JitCtx &X, const double *tile, int state_slot) const {
        const uint32_t inst = blockIdx.x * (WAVES * R) + X.lane;  // (unsplit: n_seg == 1; segments that warm up leave their state through capture())
        if (X.wave != 0 || X.lane >= WAVES * R || inst >= A.n_inst || A.warm) return;
        double *st = A.state + (size_t)state_slot * A.n_pad + inst;
        const lds_double *mem = (const lds_double *)(uintptr_t)const_cast<JitFilterK *>(this)->memory_address(const_cast<double *>(tile), X.lane);  // (behind the chunk loop's last barrier)
        st[(size_t)9 * A.n_pad] = mem[0];
        st[(size_t)10 * A.n_pad] = mem[1];
    }
};
// ---- Filter with a cutoff that is a constant of the circuit, as a SCAN over the chunk instead of the workgroup's serving wave.
// Filter.js:40-46 is y[t] = f32((P[t] - b1 y[t-1]) - b2 y[t-2]): a chain of 480 000 dependent steps for ten seconds, five
// instructions a step — what a circuit with a Filter stage takes however many instances share the chip.  Without the
// rounding to f32 the recurrence is LINEAR: the pair s = (y[t], y[t-1]) after a stretch of samples is  M^n s_before + (the stretch's
// response from rest),  M = [[-b1, -b2], [1, 0]], and stretches combine associatively.  So the wave renders its chunk at once: every
// lane runs its four samples from rest (lane 0: from the carried pair), a scan over the lanes — four steps inside the rows of 16
// (DPP row_shr 1 2 4 8 with M^4 M^8 M^16 M^32), three across them (the row's last lane broadcast, times M^(4 (j + 1)) held per
// lane) — gives every lane the pair in front of its samples, and the lane runs its four steps again from there, rounding each y to
// f32 as the reference does.  What differs from the reference is that the scan carries UNROUNDED pairs: the reference's rounding
// errors (at most 2^-24 |y| a step) never enter, and a rounding error e[t] reaches later samples through the all-pole part
// 1 / (1 + b1 z^-1 + b2 z^-2).  The deviation is therefore at most  2^-24 max|y| (sum|h| + 2)  with h that part's impulse response (+ 2:
// the two results' own roundings to f32) — the code generator takes this form only where that bound is 2^-19 (1.9e-6) of the signal's
// scale (jit_filter_scan_ok: sum|h| <= 30, cutoffs between about 1.5 and 22.5 kHz at 48 kHz), a fifth of the 1e-5 this path is held to;
// lower cutoffs keep the serving wave, bit for bit.
// A chunk that meets a NaN or anything beyond 1e30 (where the reference's `|| 0` or f32 overflow would act) is run again as written:
// serially, out of the lanes' registers.
struct JitFilterScanK {  // one per Filter: what the wave's instances share
    double k[5], lastF;  // a0 a1 a2 b1 b2 (uniform)
    double m[4][4];      // M^4, M^8, M^16, M^32: m11 m12 m21 m22 (the same in every lane)
    // what crosses the rows of 16, per lane (j = the lane's place in its row): w = M^(4 (j + 1)) in rows 1 and 3 (from lane 15 of the
    // row in front), w2 = the same in row 2 and times M^64 in row 3 (from lane 31) — and ZERO in the rows the broadcast does not
    // reach, so that whatever those lanes' registers hold there drops out (no register has to be cleared in front of the move)
    double w[4], w2[4];
    double nb1_0, nb2_0;  // -b1, -b2 in lane 0, zero elsewhere: only lane 0 starts from the carried pair
    static __device__ __forceinline__ void mul(const double (&a)[4], const double (&b)[4], double (&c)[4]) {
        const double c0 = fma(a[0], b[0], a[1] * b[2]), c1 = fma(a[0], b[1], a[1] * b[3]), c2 = fma(a[2], b[0], a[3] * b[2]), c3 = fma(a[2], b[1], a[3] * b[3]);
        c[0] = c0; c[1] = c1; c[2] = c2; c[3] = c3;
    }
    __device__ __forceinline__ void begin(const JitArgs &A, const JitCtx &X, int kind, float f, int state_slot) {
        const double *is = A.init_state + state_slot;  // has_lastF lastF a0 a1 a2 b1 b2 x1 x2 y1 y2
        const double ft = (double)f;
        if (is[0] == 0.0 || ft != is[1]) jit_filter_coefficients(kind, ft, X.srd, k);  // `if (this.f[t] != this.lastF)`
        else
            for (int i = 0; i < 5; ++i) k[i] = is[2 + i];
        for (int i = 0; i < 5; ++i) k[i] = jit_u(k[i]);
        lastF = jit_u(ft);
        const double M[4] = {-k[3], -k[4], 1.0, 0.0};
        double M2[4];
        mul(M, M, M2);
        mul(M2, M2, m[0]);
        for (int i = 1; i < 4; ++i) mul(m[i - 1], m[i - 1], m[i]);
        double wj[4];
        for (int j = 0; j < 4; ++j) wj[j] = m[0][j];
        for (uint32_t i = 0; i < 15u; ++i) {
            double n[4];
            mul(wj, m[0], n);
            if (i < (X.lane & 15u))
                for (int j = 0; j < 4; ++j) wj[j] = n[j];
        }
        double M64[4], far[4];
        mul(m[3], m[3], M64);
        mul(wj, M64, far);
        const uint32_t row = X.lane >> 4;
        for (int j = 0; j < 4; ++j) {
            w[j] = (row & 1u) ? wj[j] : 0.0;
            w2[j] = row == 3u ? far[j] : row == 2u ? wj[j] : 0.0;
        }
        nb1_0 = X.lane == 0 ? -k[3] : 0.0;
        nb2_0 = X.lane == 0 ? -k[4] : 0.0;
    }
};

struct JitFilterScan {  // one per Filter and instance
    float x1, x2;  // lane 0: the two inputs before the chunk, as they were (Filter.js:47-48's `|| 0`: where they are used as written, and at write-back);
                   // lane l: lane l - 1's last two of the chunk before (one rotation a chunk, and the shift that fetches the neighbours' writes over it)
    float y1, y2;  // every lane: its last two outputs of the chunk before (lane 63's are the pair the chunk starts from)
    bool finite;   // the last chunk took the scan: none of its outputs is a NaN or an infinity (uniform)
    __device__ __forceinline__ void begin(const JitArgs &A, int state_slot) {
        const double *is = A.init_state + state_slot;
        x1 = (float)is[7];
        x2 = (float)is[8];
        finite = false;
        y1 = (float)is[9];  // (outputs are f32 samples: nothing is lost)
        y2 = (float)is[10];
    }
    static __device__ __forceinline__ float or0f(float v) { return (v != v || v == 0.f) ? 0.f : v; }
    template <int CTRL, int ROWS, bool ZERO>
    static __device__ __forceinline__ double dpp(double old, double v) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)__double2loint(old), (int)(uint32_t)__double2loint(v), CTRL, ROWS, 0xf, ZERO);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)__double2hiint(old), (int)(uint32_t)__double2hiint(v), CTRL, ROWS, 0xf, ZERO);
        return __hiloint2double((int)hi, (int)lo);
    }
    static __device__ __forceinline__ double lane_of(double v, uint32_t l) {  // (l uniform)
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)__double2loint(v), (int)l);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)__double2hiint(v), (int)l);
        return __hiloint2double((int)hi, (int)lo);
    }
    template <int I>
    static __device__ __forceinline__ void step(const JitFilterScanK &K, double &c1, double &c2, double &t1, double &t2) {
        t1 = dpp<0x110 + (1 << I), 0xf, true>(0.0, c1);  // row_shr, zeros shifted in
        t2 = dpp<0x110 + (1 << I), 0xf, true>(0.0, c2);
        c1 = fma(K.m[I][0], t1, fma(K.m[I][1], t2, c1));
        c2 = fma(K.m[I][2], t1, fma(K.m[I][3], t2, c2));
    }
    __device__ __forceinline__ void tick(const JitCtx &X, const JitFilterScanK &K, const float (&x)[4], float (&out)[4]) {
        const double a0 = K.k[0], a1 = K.k[1], a2 = K.k[2], b1 = K.k[3], b2 = K.k[4];
        // the feed-forward half P[t] = a0 x[t] + a1 x[t-1] + a2 x[t-2] (a NaN among the inputs ends in the check below)
        const float l1 = __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(x1), __float_as_uint(x[3]), 0x138, 0xf, 0xf, false));
        const float l2 = __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(x2), __float_as_uint(x[2]), 0x138, 0xf, 0xf, false));
        const double o[6] = {(double)l2, (double)l1, (double)x[0], (double)x[1], (double)x[2], (double)x[3]};
        double p[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) p[c] = fma(a2, o[c], fma(a1, o[c + 1], a0 * o[c + 2]));
        // the lane's four samples from rest — lane 0: from the carried pair, lane 63's of the chunk before (wave_ror:1 brings every
        // lane its neighbour's; K.nb1_0 / K.nb2_0 are zero but in lane 0, and the other lanes' pairs are finite samples)
        float q1 = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(y1), 0x13c, 0xf, 0xf, true));
        float q2 = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(y2), 0x13c, 0xf, 0xf, true));
        asm("" : "+v"(q1), "+v"(q2));  // (the compiler would fold the move into the conversion: v_cvt_f64_f32 takes no wave_ror, and its own check says so)
        const double r1 = (double)q1, r2 = (double)q2;
        const double z0 = fma(K.nb2_0, r2, fma(K.nb1_0, r1, p[0]));
        const double z1 = fma(K.nb2_0, r1, fma(-b1, z0, p[1]));
        double c2 = fma(-b2, z0, fma(-b1, z1, p[2]));
        double c1 = fma(-b2, z1, fma(-b1, c2, p[3]));
        // inside the rows of 16; then lane 15 to the row behind it (rows 1 and 3), lane 31 to rows 2 and 3.  The two broadcasts write
        // over the step before's shifted pair: the rows they leave alone keep that (finite) pair, and K.w / K.w2 are zero there.
        double t1, t2;
        step<0>(K, c1, c2, t1, t2);
        step<1>(K, c1, c2, t1, t2);
        step<2>(K, c1, c2, t1, t2);
        step<3>(K, c1, c2, t1, t2);
        t1 = dpp<0x142, 0xa, false>(t1, c1);  // row_bcast:15
        t2 = dpp<0x142, 0xa, false>(t2, c2);
        c1 = fma(K.w[0], t1, fma(K.w[1], t2, c1));
        c2 = fma(K.w[2], t1, fma(K.w[3], t2, c2));
        t1 = dpp<0x143, 0xc, false>(t1, c1);  // row_bcast:31
        t2 = dpp<0x143, 0xc, false>(t2, c2);
        c1 = fma(K.w2[0], t1, fma(K.w2[1], t2, c1));
        c2 = fma(K.w2[2], t1, fma(K.w2[3], t2, c2));
        // the pair in front of this lane's samples, and the samples once more from there, every y rounded as the reference rounds it
        double u1 = dpp<0x138, 0xf, false>(r1, c1), u2 = dpp<0x138, 0xf, false>(r2, c2);  // wave_shr:1 (lane 0 keeps the carried pair)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float y = (float)fma(-b2, u2, fma(-b1, u1, p[c]));
            out[c] = y;
            u2 = u1;
            u1 = (double)y;
        }
        // A NaN or an infinity anywhere among the lane's outputs reaches its last (y[t+1] takes b1 y[t] whatever b1 is), and one in a
        // lane's unrounded pair reaches the next lane's outputs or is this test's own
        const bool odd = !(fabsf(out[3]) < 1e30f && fabs(c1) < 1e30);
        finite = __builtin_amdgcn_ballot_w64(odd) == 0;
        if (!finite) {
            // as written (Filter.js:40-46), sample by sample out of the lanes' registers
            const double e[5] = {(double)or0f(l2), (double)or0f(l1), (double)or0f(x[0]), (double)or0f(x[1]), (double)or0f(x[2])};
#pragma unroll
            for (int c = 0; c < 4; ++c) p[c] = (a0 * (double)x[c] + a1 * e[c + 1]) + a2 * e[c];
            double v1 = (double)__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(y1), 63)), v2 = (double)__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(y2), 63));
#pragma unroll 1
            for (uint32_t l = 0; l < 64u; ++l) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const double pv = lane_of(p[c], l);
                    const float y = (float)((pv - b1 * jit_or0(v1)) - b2 * jit_or0(v2));
                    if (X.lane == l) out[c] = y;
                    v2 = jit_or0(v1);
                    v1 = (double)y;
                }
            }
        }
        // (lane 63's last two are what the serial pass ended on too; `|| 0` is applied where the pair is used as written — in the
        // serial pass — and at write-back: on this path a NaN would have ended in the check, and -0 for +0 changes no sum but a zero's sign)
        y1 = out[3];
        y2 = out[2];
        x1 = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x[3]), 0x13c, 0xf, 0xf, true));  // wave_ror:1
        x2 = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x[2]), 0x13c, 0xf, 0xf, true));
    }
    __device__ __forceinline__ void end(const JitArgs &A, const JitCtx &X, const JitFilterScanK &K, int state_slot) const {  // (lane 0 of the instance's last segment)
        double *st = A.state + (size_t)state_slot * A.n_pad + X.inst;
        st[0] = 1.0;
        st[A.n_pad] = K.lastF;
        for (int i = 0; i < 5; ++i) st[(size_t)(2 + i) * A.n_pad] = K.k[i];
        st[(size_t)7 * A.n_pad] = (double)__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x1), 0));
        st[(size_t)8 * A.n_pad] = (double)or0f(__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x2), 0)));
        st[(size_t)9 * A.n_pad] = (double)__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(y1), 63));
        st[(size_t)10 * A.n_pad] = (double)or0f(__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(y2), 63)));
    }
};

// parameter `slot` of the instance lane `lane` serves in the Filter stage
template <int WAVES, int R>
__device__ __forceinline__ float jit_row_param(const JitArgs &A, const JitCtx &X, uint32_t slot) {
    const uint32_t inst = min((blockIdx.x * (WAVES * R) + X.lane) / A.n_seg, A.n_inst - 1);
    return A.params[(size_t)slot * A.n_inst + inst];
}

// ---- Filter with a CONNECTED cutoff (an LFO or an envelope on the cutoff): `if (this.f[t] != this.lastF)` recomputes the coefficients
// (Filter.js:34-37) — a pure function of f[t], so every lane computes its four samples' own (filter_lamda: two short polynomials and
// a division, no argument reduction) together with the feed-forward half P[t] = (a0 x + a1 x1) + a2 x2, and the recurrence
// y = f32((P - b1[t] y1) - b2[t] y2) runs like the constant-cutoff stage's: the WAVES x R instances of the workgroup side by side on
// the lanes of ONE wave, out of a shared tile whose rows hold P, b1 and b2 of a sub-block (three arrays of SUB doubles; y goes
// over the P values already consumed).  Same protocol as JitFilterK: park / serial / failed / serial_exact / pick, memory behind the tile.
template <int WAVES, int R, int SUB, int MEM>
struct JitFilterKM {
    static constexpr int kPitch = 3 * SUB + 2;  // doubles per row (even: rows stay 16-byte aligned)
    float x1[R], x2[R];                         // the two inputs before the chunk: x1 as it was, x2 through `|| 0` (Filter.js:47-48)
    float flast[R];                             // the cutoff of the last sample ticked (what `lastF` and the coefficients in the unit's state belong to)
    bool ticked[R];
    int stage;
#ifdef DUSP_JIT_PROFILE
    unsigned long long cyc_serial = 0;
#endif
    typedef __attribute__((address_space(3))) double lds_double;
    typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    static __device__ __forceinline__ uint32_t row_address(double *tile, uint32_t row) {
        return ((uint32_t)(uintptr_t)(lds_double *)(tile + (size_t)row * kPitch)) & 0x3ffffu;
    }
    static __device__ __forceinline__ uint32_t flag_address(double *tile) { return row_address(tile, 0) + (uint32_t)MEM; }
    __device__ __forceinline__ uint32_t memory_address(double *tile, uint32_t row) const {
        return flag_address(tile) + 16u + ((uint32_t)stage * (WAVES * R) + row) * 16u;
    }
    __device__ __forceinline__ void begin(const JitArgs &A, const JitCtx &X, double *tile, int stage_, int state_slot) {
        stage = stage_;
        const double *is = A.init_state + state_slot;  // has_lastF lastF a0 a1 a2 b1 b2 x1 x2 y1 y2
        if (X.wave == 0 && X.lane < WAVES * R) {  // (the first barrier of the chunk loop stands between this and the first reader)
            lds_double *mem = (lds_double *)(uintptr_t)memory_address(tile, X.lane);
            mem[0] = is[9];
            mem[1] = is[10];
        }
    }
    __device__ __forceinline__ void begin_slot(const JitArgs &A, const JitCtx &X, int r, int state_slot) {
        const double *is = A.init_state + state_slot;
        x1[r] = jit_u((float)is[7]);  // (inputs are f32 samples: nothing is lost)
        x2[r] = jit_u((float)is[8]);
        flast[r] = 0.f;
        ticked[r] = false;
    }
    static __device__ __forceinline__ float or0f(float v) { return (v != v || v == 0.f) ? 0.f : v; }
    // Sub-block s of the wave's instances r0 .. r0 + PER - 1 (PER = 64 / SUB of them side by side on the wave's lanes): every lane
    // takes ONE sample — lane l: instance r0 + l / SUB, sample s SUB + l % SUB — computes its coefficients and feed-forward half and
    // parks P, b1, b2 in the instance's tile row.  The samples come out of the owning lanes' registers (lane 4 j' holds samples
    // 4 j' .. 4 j' + 3) through the row itself: the owners lay the sub-block's inputs and cutoffs out there as floats, everybody reads
    // theirs (and the two inputs before it; in front of the sub-block, the carried pair), then the doubles go over them.
    // xa / fa: input and cutoff registers of instance r0, xb / fb of instance r0 + 1 (PER == 2).
    template <int PER>
    __device__ __forceinline__ void parkm(const JitCtx &X, double *tile, int r0, int s, int kind, const float (&xa)[4], const float (&fa)[4],
                                          const float (&xb)[4], const float (&fb)[4]) const {
        static_assert(PER * SUB <= 64 && (PER == 1 || PER == 2), "one or two instances per pass");
        const int mine = PER == 2 ? (int)(X.lane / SUB) : 0;          // which of the pass's instances this lane works for
        const int j = (int)(X.lane % SUB);
        const bool active = (int)X.lane < PER * SUB && r0 + mine < R;
        const int first = s * (SUB / 4);
        jit_wave_sync();  // (the rows' previous readers — pick — are done)
#pragma unroll
        for (int q = 0; q < PER; ++q) {  // the owners of the sub-block's samples: four inputs and four cutoffs each
            if (r0 + q >= R) break;
            if ((int)X.lane >= first && (int)X.lane < first + SUB / 4) {
                float *row = (float *)(tile + (size_t)(X.wave * R + r0 + q) * kPitch);
                const float (&x)[4] = q == 0 ? xa : xb;
                const float (&f)[4] = q == 0 ? fa : fb;
                ((f32x4 *)row)[(int)X.lane - first] = f32x4{x[0], x[1], x[2], x[3]};
                ((f32x4 *)(row + SUB))[(int)X.lane - first] = f32x4{f[0], f[1], f[2], f[3]};
            }
        }
        jit_wave_sync();
        double p = 0.0, b1 = 0.0, b2 = 0.0;
        double *drow = tile + (size_t)(X.wave * R + (active ? r0 + mine : r0)) * kPitch;
        if (active) {
            const float *row = (const float *)drow;
            const float xin = row[j], fc = row[SUB + j];
            // the two inputs before this one: earlier samples of the sub-block, else what the instance carries
            const int rm = mine == 0 ? r0 : (r0 + 1) % R;
            const float c1 = x1[rm], c2 = x2[rm];
            const float m1 = j >= 1 ? row[j - 1] : c1;
            const float m2 = j >= 2 ? row[j - 2] : (j == 1 ? c1 : c2);
            double k[5];
            jit_filter_coefficients(kind, (double)fc, X.srd, k);
            p = (k[0] * (double)xin + k[1] * (double)or0f(m1)) + k[2] * (double)or0f(m2);  // Filter.js:40-42, in its order of roundings
            b1 = k[3];
            b2 = k[4];
        }
        jit_wave_sync();  // (everybody has read the floats: the doubles go over them)
        if (active) {
            drow[j] = p;
            drow[SUB + j] = b1;
            drow[2 * SUB + j] = b2;
        }
    }
    // behind sub-block s of slot r: the instance's carried inputs and cutoff are now the sub-block's last (its owners' last lane holds them)
    __device__ __forceinline__ void carry(int r, int s, const float (&x)[4], const float (&f)[4]) {
        const int last = (s + 1) * (SUB / 4) - 1;
        x1[r] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x[3]), last));
        x2[r] = or0f(__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x[2]), last)));
        flast[r] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(f[3]), last));
        ticked[r] = true;
    }
    static __device__ __forceinline__ void pick(const JitCtx &X, const double *tile, int r, int s, float (&out)[4]) {
        const int first = s * (SUB / 4);
        if ((int)X.lane >= first && (int)X.lane < first + SUB / 4) {
            const f32x4 y = ((const f32x4 *)(tile + (size_t)(X.wave * R + r) * kPitch))[(int)X.lane - first];
            out[0] = y[0]; out[1] = y[1]; out[2] = y[2]; out[3] = y[3];
        }
    }
    // PB steps with the coefficients of each step, WITHOUT the `|| 0` selects of Filter.js:42-46 (speculated away like JitFilterK::serial does)
    template <int PB, typename DST>
    static __device__ __forceinline__ void block(const double (&pv)[PB], const double (&b1)[PB], const double (&b2)[PB], double &u1, double &u2, DST *dst) {
        f32x4 y4[PB / 4];
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const float y = (float)((pv[i] - b1[i] * u1) - b2[i] * u2);
            y4[i >> 2][i & 3] = y;
            u2 = u1;
            u1 = (double)y;
        }
#pragma unroll
        for (int i = 0; i < PB / 4; ++i) dst[i] = y4[i];
    }
    template <int PB_>  // (the generated text passes the constant-cutoff stage's block size; three values per step: always blocks of 4 here)
    __device__ __forceinline__ void serial(const JitCtx &X, double *tile, uint32_t who) {
        constexpr int PB = 4;
        if (X.wave != who || X.lane >= WAVES * R) return;
#ifdef DUSP_JIT_PROFILE
        const unsigned long long stamp0 = __builtin_readcyclecounter();
#endif
        const uint32_t row = row_address(tile, X.lane);
        const lds_double *pr = (const lds_double *)(uintptr_t)row;
        lds_f32x4 *yr = (lds_f32x4 *)(uintptr_t)row;
        lds_double *mem = (lds_double *)(uintptr_t)memory_address(tile, X.lane);
        double u1 = jit_or0(mem[0]), u2 = jit_or0(mem[1]);
        double pa[PB], ba[PB], ca[PB], pb[PB], bb[PB], cb[PB];
#pragma unroll
        for (int i = 0; i < PB; ++i) pa[i] = pr[i], ba[i] = pr[SUB + i], ca[i] = pr[2 * SUB + i];
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        for (int t0 = 0; t0 < SUB; t0 += 2 * PB) {
#pragma unroll
            for (int i = 0; i < PB; ++i) pb[i] = pr[PB + i], bb[i] = pr[SUB + PB + i], cb[i] = pr[2 * SUB + PB + i];
            __builtin_amdgcn_sched_barrier(0);
            block<PB>(pa, ba, ca, u1, u2, yr);
            const int next = t0 + 2 * PB < SUB ? 2 * PB : 0;  // (the last block reads itself again: no branch around the reads)
#pragma unroll
            for (int i = 0; i < PB; ++i) pa[i] = pr[next + i], ba[i] = pr[SUB + next + i], ca[i] = pr[2 * SUB + next + i];
            __builtin_amdgcn_sched_barrier(0);
            block<PB>(pb, bb, cb, u1, u2, yr + PB / 4);
            pr += 2 * PB;
            yr += PB / 2;
        }
        const bool met_nan = __builtin_amdgcn_ballot_w64(!(u1 == u1)) != 0;
        if (X.lane == 0) *(lds_u32 *)(uintptr_t)flag_address(tile) = met_nan ? 1u : 0u;
        if (!met_nan) {
            mem[0] = u1;
            mem[1] = u2;
        }
#ifdef DUSP_JIT_PROFILE
        cyc_serial += __builtin_readcyclecounter() - stamp0;
#endif
    }
    static __device__ __forceinline__ bool failed(double *tile) {
        return __builtin_amdgcn_readfirstlane((int)*(const lds_u32 *)(uintptr_t)flag_address(tile)) != 0;
    }
    __device__ __forceinline__ void serial_exact(const JitCtx &X, double *tile, uint32_t who) {
        if (X.wave != who || X.lane >= WAVES * R) return;
        const uint32_t row = row_address(tile, X.lane);
        const lds_double *pr = (const lds_double *)(uintptr_t)row;
        lds_f32x4 *yr = (lds_f32x4 *)(uintptr_t)row;
        lds_double *mem = (lds_double *)(uintptr_t)memory_address(tile, X.lane);
        double y1 = mem[0], y2 = mem[1];
        for (int t0 = 0; t0 < SUB; t0 += 4) {
            double pv[4], b1[4], b2[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) pv[i] = pr[t0 + i], b1[i] = pr[SUB + t0 + i], b2[i] = pr[2 * SUB + t0 + i];
            f32x4 y4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float y = (float)((pv[i] - b1[i] * jit_or0(y1)) - b2[i] * jit_or0(y2));
                y4[i] = y;
                y2 = jit_or0(y1);
                y1 = (double)y;
            }
            yr[t0 >> 2] = y4;
        }
        mem[0] = y1;
        mem[1] = y2;
    }
    __device__ __forceinline__ void end_slot(const JitArgs &A, const JitCtx &X, int r, int kind, int state_slot) const {
        if (!X.live || X.lane != 0) return;
        double *st = A.state + (size_t)state_slot * A.n_pad + X.inst;
        const double *is = A.init_state + state_slot;
        if (ticked[r]) {
            double k[5];
            jit_filter_coefficients(kind, (double)flast[r], X.srd, k);
            st[0] = 1.0;
            st[A.n_pad] = (double)flast[r];
            for (int i = 0; i < 5; ++i) st[(size_t)(2 + i) * A.n_pad] = k[i];
        } else
            for (int i = 0; i < 7; ++i) st[(size_t)i * A.n_pad] = is[i];
        st[(size_t)7 * A.n_pad] = (double)x1[r];
        st[(size_t)8 * A.n_pad] = (double)x2[r];
    }
    __device__ __forceinline__ void end(const JitArgs &A, const JitCtx &X, const double *tile, int state_slot) const {
        const uint32_t inst = blockIdx.x * (WAVES * R) + X.lane;  // (n_seg == 1 whenever a circuit has a Filter)
        if (X.wave != 0 || X.lane >= WAVES * R || inst >= A.n_inst) return;
        double *st = A.state + (size_t)state_slot * A.n_pad + inst;
        const lds_double *mem = (const lds_double *)(uintptr_t)const_cast<JitFilterKM *>(this)->memory_address(const_cast<double *>(tile), X.lane);
        st[(size_t)9 * A.n_pad] = mem[0];
        st[(size_t)10 * A.n_pad] = mem[1];
    }
};

// ---- Delay (src/components/Delay.js:20-41) with a constant delay D + phi, 256 <= D <= len - 256: the chunk's 256 reads are one
// coalesced load from the ring ([instance][slot] in HBM), reads and writes of one chunk never meet, and every slot's final
// value (ceil tap of sample n-1, then floor tap of sample n, with the reference's two `+=` roundings) is written once.
// Everything a chunk reads was written before the chunk began (D >= 256), so the NEXT chunk's reads are issued right after
// this chunk's writes and have the rest of the chunk — the Filter stage, usually — to arrive.
// MONO: MonoDelay — the same ring protocol (its taps and its read never meet inside a chunk either), no dropped ceil tap, no state.
// EXACT: the program will be continued (dusp_program_continue), so the ring is kept in exactly the state the reference's has at a launch
// boundary — slots zeroed once read (Delay.js:29), the launch's last ceil tap in place — so that the chain can move to the chunk
// engine's read-modify-write protocol whenever an event changes the delay; the carried input sample comes from the last launch.
template <bool MONO, bool EXACT = false>
struct JitDelayK {
    float carried;        // the input sample before the chunk (uniform)
    double phi;           // the delay's fraction (uniform)
    float *ring;          // this instance's ring
    uint32_t len, D, s0;  // ring length, whole delay, the chunk's first slot
    bool quad;            // ring length, delay and first slot are multiples of 4: a lane's four slots are ONE 16-byte access and wrap together
    float ahead[4];       // the coming chunk's reads
    __device__ __forceinline__ void fetch(const JitCtx &X) {
        if (quad) {
            uint32_t s_ = s0 + X.lane * 4;
            if (s_ >= len) s_ -= len;
            const f32x4 v = *(const f32x4 *)(ring + s_);
            ahead[0] = v[0]; ahead[1] = v[1]; ahead[2] = v[2]; ahead[3] = v[3];
            return;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint32_t s_ = s0 + X.lane * 4 + c;
            if (s_ >= len) s_ -= len;
            ahead[c] = ring[s_];
        }
    }
    __device__ __forceinline__ void begin(const JitArgs &A, const JitCtx &X, int state_slot, int64_t ring_base, int64_t ring_len, float delay) {
        carried = MONO ? 0.f : jit_u((float)A.init_state[state_slot]);  // (an f32 sample: nothing is lost)
        if (EXACT && !MONO && A.resume) carried = jit_u((float)A.state[(size_t)state_slot * A.n_pad + X.inst]);
        double dconst = (double)delay;
        if (dconst >= (double)ring_len) dconst = fmod(dconst, (double)ring_len);
        const double Dfl = floor(dconst);
        phi = jit_u(dconst - Dfl);
        D = jit_u((uint32_t)Dfl);
        len = (uint32_t)ring_len;
        ring = A.rings + (size_t)X.inst * (size_t)A.ring_samples + (size_t)ring_base;
        s0 = (uint32_t)((A.clock0 + (uint64_t)X.g_begin * kChunk) % (uint64_t)ring_len);
        quad = jit_u(((len | D | s0) & 3u) == 0u && ((uintptr_t)ring & 15u) == 0u);
        fetch(X);
    }
    __device__ __forceinline__ void tick(const JitCtx &X, uint32_t g, const float (&x)[4], float (&out)[4]) {
        read(X, out);
        write(X, g, x);
    }
    // The two halves of a tick.  What a chunk reads does not depend on what it writes, so a generated kernel may take the reads
    // early (the Filter stage's input) and the writes late (behind the recurrences), see jit_codegen.hpp `plan_overlap`.
    __device__ __forceinline__ void read(const JitCtx &X, float (&out)[4]) const {
#pragma unroll
        for (int c = 0; c < 4; ++c) out[c] = ahead[c];
        if (EXACT && X.live) {  // `buf[tB] = 0` (Delay.js:29): the slots this chunk has read
            if (quad) {
                uint32_t s_ = s0 + X.lane * 4;
                if (s_ >= len) s_ -= len;
                *(f32x4 *)(ring + s_) = f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    uint32_t s_ = s0 + X.lane * 4 + c;
                    if (s_ >= len) s_ -= len;
                    ring[s_] = 0.f;
                }
            }
        }
    }
    __device__ __forceinline__ void write(const JitCtx &X, uint32_t g, const float (&x)[4]) {
        float slot[4];
        if (phi != 0.0) {
            // the lane before this one's x[3]; lane 0 gets the carried sample (DPP wave_shr:1, `old` stays where no lane shifts in)
            const float x_left = __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(carried), __float_as_uint(x[3]), 0x138, 0xf, 0xf, false));
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                uint32_t lo = s0 + X.lane * 4 + c;  // (only slot 0 of the ring drops the ceil tap: the test needs the slot)
                if (lo >= len) lo -= len;
                lo += D;
                if (lo >= len) lo -= len;
                const double xin = (double)x[c], xprev = (double)(c == 0 ? x_left : x[c - 1]);
                const float tap = (MONO || lo != 0) ? (float)(0.0 + xprev * phi) : 0.f;  // ceil tap of sample n-1 (Delay: dropped at slot 0)
                slot[c] = (float)((double)tap + xin * (1.0 - phi));            // floor tap of sample n
            }
        } else {
            // (float)(0.0 + x * 1.0), then (float)(that + x * 0.0): x with -0 turned into +0, or NaN for a NaN / Inf — the same two
            // steps in f32 give the same bits, and the product x * 0 is exact, so the second is one fma
#pragma unroll
            for (int c = 0; c < 4; ++c) slot[c] = __builtin_fmaf(x[c], 0.f, x[c] + 0.f);
        }
        if (quad) {
            uint32_t lo = s0 + X.lane * 4;
            if (lo >= len) lo -= len;
            lo += D;
            if (lo >= len) lo -= len;
            if (X.live) *(f32x4 *)(ring + lo) = f32x4{slot[0], slot[1], slot[2], slot[3]};
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                uint32_t lo = s0 + X.lane * 4 + c;
                if (lo >= len) lo -= len;
                lo += D;
                if (lo >= len) lo -= len;
                if (X.live) ring[lo] = slot[c];
            }
        }
        if (EXACT && !MONO && phi != 0.0 && g + 1 == X.g_end && X.lane == 63 && X.live) {
            // the launch's last ceil tap, which the next chunk's first slot would fold in: in place, as the reference's ring has it
            uint32_t lo = s0 + 255u;
            if (lo >= len) lo -= len;
            lo += D;
            if (lo >= len) lo -= len;
            if (lo + 1u < len) ring[lo + 1u] = (float)(0.0 + (double)x[3] * phi);  // (at index `length` the reference's store is dropped)
        }
        if (phi != 0.0 || g + 1 == X.g_end) carried = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x[3]), 63));  // (a whole delay: only the unit's state asks for it)
        s0 += kChunk;
        if (s0 >= len) s0 -= len;  // (len >= 512 here)
        if (!EXACT || g + 1 != X.g_end) fetch(X);  // (after this chunk's stores, in program order: a wave's accesses to one address stay in order)
    }
};

// ---- Delay with a constant delay D + phi of LESS than a chunk, 1 <= D <= 255 (Delay.js:20-41).  Sample t reads its slot and clears
// it, then adds its floor tap D slots ahead and its ceil tap D + 1 ahead; so what sample t reads is what samples t-D-1 (ceil tap,
// first) and t-D (floor tap) left there, each `+=` rounded to f32 — a function of two INPUT samples, and no ring is needed at
// all: the wave keeps the chunk before this one in registers, lays both out in its scratch row and every lane picks the
// two samples each of its four needs.  (The tap that would land on slot 0 by way of index `length` is dropped by the
// reference's Float32Array: slot 0 gets no ceil tap.)  The ring in HBM stays untouched: nothing reads it after the render.
// MONO: MonoDelay (MonoDelay.js:16-28) — taps first, then the read; the ceil tap wraps instead of being dropped; a delay of 0 reads
// the sample's own floor tap; no state (fresh ring: the sample before the render left nothing).
template <bool MONO>
struct JitDelayShort {
    float carried;     // the input sample before the chunk (uniform; what the unit's state holds)
    float before[4];   // this lane's four samples of the chunk before
    double phi;        // the delay's fraction (uniform)
    uint32_t len, D, s0;
    __device__ __forceinline__ void begin(const JitArgs &A, const JitCtx &X, int state_slot, int64_t ring_len, float delay) {
        carried = MONO ? 0.f : jit_u((float)A.init_state[state_slot]);
        const double dconst = (double)delay, Dfl = floor(dconst);
        phi = dconst - Dfl;
        D = (uint32_t)Dfl;
        len = (uint32_t)ring_len;
        s0 = (uint32_t)((A.clock0 + (uint64_t)X.g_begin * kChunk) % (uint64_t)ring_len);
#pragma unroll
        for (int c = 0; c < 4; ++c) before[c] = 0.f;  // (a fresh ring: zeros — Delay.js:14)
        if (X.lane == 63) before[3] = carried;
    }
    __device__ __forceinline__ void tick(const JitCtx &X, float *scr, const float (&x)[4], float (&out)[4]) {
        jit_wave_sync();
        jit_row_put(scr, X.lane, before);
        jit_row_put(scr + kChunk, X.lane, x);
        jit_wave_sync();
        const int j0 = kChunk + (int)X.lane * 4 - (int)D;  // this lane's first sample, D ago: 1 .. 511 - 3
        float h[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) h[i] = scr[j0 - 1 + i];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint32_t slot = s0 + X.lane * 4 + c;
            if (slot >= len) slot -= len;
            const double xin = (double)h[c + 1], xprev = (double)h[c];
            if (phi != 0.0) {
                const float tap = (MONO || slot != 0) ? (float)(0.0 + xprev * phi) : 0.f;  // ceil tap of the sample before (Delay: dropped at slot 0)
                out[c] = (float)((double)tap + xin * (1.0 - phi));               // floor tap
            } else {
                const float tap = (float)(0.0 + xin * 1.0);  // floor(tWrite) == ceil(tWrite): both `+=` of one sample
                out[c] = (float)((double)tap + xin * 0.0);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) before[c] = x[c];
        carried = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x[3]), 63));
        s0 += kChunk;
        if (s0 >= len) s0 -= len;  // (len >= 512 here: jit_delay_short)
    }
};


// ---- Delay / MonoDelay with a constant delay D + phi of a chunk at least, as a line of INPUT samples in LDS instead of the ring in
// memory.  What sample t reads from its slot is what samples t-D-1 (ceil tap, first) and t-D (floor tap) left there, each `+=` rounded
// to f32 (JitDelayShort's observation; Delay.js:26-40) — a function of two input samples and of whether the slot is the ring's slot 0
// (whose ceil tap the reference drops).  So the wave keeps the last CH chunks of the unit's input in its own LDS rows (CH = the chunks
// that D + 1 samples back can reach, plus the current one), writes this chunk's four samples per lane (one 16-byte store), and every
// lane picks the five input samples its four outputs need out of two 16-byte loads (the offset inside the quad is the same for
// every lane: D + 1 is).  No ring traffic at all: a circuit like BASELINE configs[3] moved twice its PCM in ring reads and writes.
// The ring in memory stays untouched: nothing reads it after a render that is not continued (the code generator takes this form
// for those only, and only where the rows fit next to the table image at 16 wavefronts).
template <bool MONO>
struct JitDelayLine {
    float carried;       // the input sample before the chunk (uniform; what the unit's state holds)
    double phi;          // the delay's fraction (uniform)
    uint32_t len, D, s0; // ring length, whole delay, the chunk's first slot (for the slot-0 rule)
    uint32_t at, span;   // where this chunk's row starts in the line; the line's length (CH chunks)
    uint32_t sh;         // (span - D - 1) & 3: where in its quad a lane's first sample stands
    bool whole;          // a whole delay: no ceil tap, the fifth sample (in front) is not read
    uint32_t al;         // ... and where in its quad of the line a lane's first of the four stands: (span - D) & 3
    float ahead[5];      // the coming chunk's five input samples of this lane
    float *line;         // this wave's rows
    __device__ __forceinline__ void begin(const JitArgs &A, const JitCtx &X, float *rows, uint32_t chunks, int state_slot, int64_t ring_len, float delay) {
        carried = MONO ? 0.f : jit_u((float)A.init_state[state_slot]);
        double dconst = (double)delay;
        if (dconst >= (double)ring_len) dconst = fmod(dconst, (double)ring_len);
        const double Dfl = floor(dconst);
        phi = jit_u(dconst - Dfl);
        D = jit_u((uint32_t)Dfl);
        len = (uint32_t)ring_len;
        s0 = (uint32_t)((A.clock0 + (uint64_t)X.g_begin * kChunk) % (uint64_t)ring_len);
        line = rows;
        span = chunks * kChunk;
        at = 0;
        sh = jit_u((span - D - 1u) & 3u);
        whole = jit_u(phi == 0.0);
        al = jit_u((span - D) & 3u);
        for (uint32_t i = X.lane; i < span; i += 64u) rows[i] = 0.f;  // (a fresh ring: zeros — Delay.js:14)
        jit_wave_sync();
        if (X.lane == 0) rows[span - 1u] = carried;                  // the sample before the render
        jit_wave_sync();
        fetch(X, 0);
    }
    // the five samples from D + 1 before this lane's first one of the chunk whose row starts at `row`: two quads, the offset inside the first
    // the same in every lane and chunk
    __device__ __forceinline__ void fetch(const JitCtx &X, uint32_t row) {
        if (whole) {  // (uniform; so is lane 0's place in the line: scalar arithmetic, then one wrap a lane — 4 lane < 256 <= span)
            uint32_t base = row + span - D;
            if (base >= span) base -= span;
            if (base >= span) base -= span;
            uint32_t q0 = base + X.lane * 4u - al;
            q0 = min(q0, q0 - span);  // (an underflow loses the min)
            const f32x4 a = *(const f32x4 *)(line + q0);
            ahead[0] = 0.f;
            if (al == 0u) {
                ahead[1] = a[0]; ahead[2] = a[1]; ahead[3] = a[2]; ahead[4] = a[3];
                return;
            }
            uint32_t q1 = q0 + 4u;
            q1 = min(q1, q1 - span);
            const f32x4 b = *(const f32x4 *)(line + q1);
            // (branches on a scalar: which registers the four are is known in each, nothing is selected)
            if (al == 1u) {
                ahead[1] = a[1]; ahead[2] = a[2]; ahead[3] = a[3]; ahead[4] = b[0];
            } else if (al == 2u) {
                ahead[1] = a[2]; ahead[2] = a[3]; ahead[3] = b[0]; ahead[4] = b[1];
            } else {
                ahead[1] = a[3]; ahead[2] = b[0]; ahead[3] = b[1]; ahead[4] = b[2];
            }
            return;
        }
        uint32_t j = row + X.lane * 4u + span - D - 1u;  // (D + 1 <= span - 256: never negative)
        if (j >= span) j -= span;
        if (j >= span) j -= span;
        uint32_t q0 = j - sh, q1 = q0 + 4u;
        if (q1 >= span) q1 -= span;
        const f32x4 a = *(const f32x4 *)(line + q0), b = *(const f32x4 *)(line + q1);
        // (branches on a scalar: which registers the five are is known in each, nothing is selected or indexed)
        if (sh == 0u) {
            ahead[0] = a[0]; ahead[1] = a[1]; ahead[2] = a[2]; ahead[3] = a[3]; ahead[4] = b[0];
        } else if (sh == 1u) {
            ahead[0] = a[1]; ahead[1] = a[2]; ahead[2] = a[3]; ahead[3] = b[0]; ahead[4] = b[1];
        } else if (sh == 2u) {
            ahead[0] = a[2]; ahead[1] = a[3]; ahead[2] = b[0]; ahead[3] = b[1]; ahead[4] = b[2];
        } else {
            ahead[0] = a[3]; ahead[1] = b[0]; ahead[2] = b[1]; ahead[3] = b[2]; ahead[4] = b[3];
        }
    }
    __device__ __forceinline__ void tick(const JitCtx &X, const float (&x)[4], float (&out)[4]) {
        // what this chunk reads was written a chunk ago at least (D >= 256) and fetched then: the reads of a chunk do not wait for its input
        if (phi != 0.0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                uint32_t slot = s0 + X.lane * 4 + c;
                if (slot >= len) slot -= len;
                const double xin = (double)ahead[c + 1], xprev = (double)ahead[c];
                const float tap = (MONO || slot != 0) ? (float)(0.0 + xprev * phi) : 0.f;  // ceil tap of the sample before (Delay: dropped at slot 0)
                out[c] = (float)((double)tap + xin * (1.0 - phi));               // floor tap
            }
        } else {
            // floor(tWrite) == ceil(tWrite): both `+=` of one sample, (float)(0.0 + x * 1.0) and then (float)(that + x * 0.0) — x with -0
            // turned into +0, or NaN for a NaN / Inf; the same two steps in f32 give the same bits, the second (an exact product) as one fma
#pragma unroll
            for (int c = 0; c < 4; ++c) out[c] = __builtin_fmaf(ahead[c + 1], 0.f, ahead[c + 1] + 0.f);
        }
        // (no fence between the store and the loads: a wave's LDS accesses execute in the order they are issued, and the compiler keeps a
        // store and the loads behind it that may alias in that order — a fence here would also pin the Filter behind this unit's input)
        *(f32x4 *)(line + at + X.lane * 4u) = f32x4{x[0], x[1], x[2], x[3]};
        uint32_t next = at + kChunk;
        if (next >= span) next = 0;
        fetch(X, next);  // the coming chunk's
        carried = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x[3]), 63));
        s0 += kChunk;
        if (s0 >= len) s0 -= len;  // (len >= 512 here)
        at += kChunk;
        if (at >= span) at = 0;
    }
};

}  // namespace
}  // namespace dusp
