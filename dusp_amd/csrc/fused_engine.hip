// fused_engine.hip — fused, time-parallel render kernels for feed-forward voices.
//
// One LANE per SAMPLE: a wavefront renders 256 consecutive samples of a circuit
// instance per step (lane l owns samples 4l..4l+3 of the group), so the PCM
// leaves as one fully coalesced 1 KiB `global_store_dwordx4` per wave per voice
// per step and nothing is staged through HBM.  The whole unit tree of the
// circuit is evaluated in registers — this is the "one fused kernel per
// topologically sorted Circuit" of the north star for the shapes listed in
// fused_plan.hpp.
//
// Work decomposition.  A work item is (block of R voices, time segment); one
// wave takes one item at a time from a static stride over a persistent grid
// (one 1024-thread workgroup per CU).  Splitting TIME is legal because every
// stateful unit in these shapes can jump to any sample index exactly:
//   * Osc with a lane-constant f (Osc.js:38-46): phase(t) = (phase0 + (t+1) f) mod sr.
//     The reference accumulates in f64; while lsb(f) >= 2^-36 every partial sum
//     is an exact multiple of lsb(f) below 2^17, so no rounding ever happens
//     (SURVEY.md §8a note ii) and the closed form — evaluated in exact u64
//     fixed-point modular arithmetic — is bit-identical to the sequential loop.
//     Inside a segment each lane advances its own phase by (256 f) mod sr per
//     step.  Three representations, chosen per voice block (wave-uniform):
//       INT   f and phase0 integers: phase is a u32 table index, the lerp
//             degenerates to table[phase] (fraction == 0, Osc.js:43-45);
//       FX32  lsb(f) >= 2^-32: phase = u32 index + u32 fraction (32.32 fixed
//             point, add-with-carry), lerp weights are exact f64 integers;
//       F64   anything else: f64 phase, conditional subtract.
//   * Ramp (Ramp.js:25-40): t(n) = min(t0 + n + 1, duration) while playing.  It
//     depends on the sample index only, so a wave evaluates it ONCE per step and
//     reuses it for the R voices of its block.
// Wave-table lookups come from a half-table in LDS when the table is
// antisymmetric (T[N-t] == -T[t], true for the sine table; the full 192 KB
// table does not fit the 160 KB LDS), else from L2 (global gather, ~5x slower).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "device_types.hpp"
#include "fused_device.hpp"
#include "fused_plan.hpp"
#include "repeat_add.hpp"

namespace dusp {

namespace {

// y0 + (t/duration) * (y1-y0) with t = min(tn, duration)  (Ramp.js:27-38)
template <bool FASTDIV>
__device__ __forceinline__ float ramp_value(const FusedArgs &A, double tn, double dy) {
    const double tt = fmin(tn, A.r_d);
    double q;
    if (FASTDIV) {  // host-verified to equal tt / duration on this Ramp's whole t sequence (fused_plan.hpp)
        q = tt * A.r_rcp;
        const double rem = fma(-q, A.r_d, tt);
        q = fma(rem, A.r_rcp, q);
    } else
        q = tt / A.r_d;
    return (float)(A.r_y0 + q * dy);
}

// Shape/index.js:33-58 for one sample: the 0..1 shape at t (table lerp, or an edge) scaled into [min, max]
__device__ __forceinline__ float shape_value(const FusedArgs &A, double t, double srd) {
    const double l = (double)A.s_min, h = (double)A.s_max;
    if (t <= 0.0) return (float)((A.s_left_is_shape ? (double)A.s_table[0] : A.s_left) * (h - l) + l);
    if (t > srd) return (float)((A.s_right_is_shape ? (double)A.s_table[(uint32_t)srd] : A.s_right) * (h - l) + l);
    if (t != t) return __builtin_nanf("");
    const double fl = floor(t), frac = t - fl;
    return (float)(l + (h - l) * ((double)A.s_table[(int)ceil(t)] * frac + (double)A.s_table[(int)fl] * (1.0 - frac)));
}

// phase (in units of 2^E) of this lane's 4 samples at the start of segment `seg`:
// P(t) = P0 + (t+1) Fm  with  t = seg * seg_len + 4 lane + c
__device__ __forceinline__ void start_phase(const OscRec &rc, uint32_t seg, uint32_t lane, uint64_t (&P)[4]) {
    const uint64_t base = addmod(addmod(rc.P0, rc.Fm, rc.S), mulmod(rc.segstep, seg, rc.S), rc.S);
    P[0] = addmod(base, mulmod(rc.step4, lane, rc.S), rc.S);
    for (int c = 1; c < 4; ++c) P[c] = addmod(P[c - 1], rc.Fm, rc.S);
}

}  // namespace

// One thread per voice: decompose f into exact fixed point, precompute the strides the render
// kernel needs, and write the end-of-render state (state write-back).
__global__ void dusp_fused_prepare(FusedArgs A, OscRec *recs) {
    const uint32_t inst = blockIdx.x * blockDim.x + threadIdx.x;
    if (inst >= A.n_inst) return;
    const uint32_t sr = A.sample_rate;
    const double srd = (double)sr;
    OscRec rc;
    double fd = (double)operand_value(A.f, A.params, A.n_inst, inst);
    rc.bad = !(fabs(fd) <= 3.0e38);  // non-finite f: every sample is NaN -> 0 after `|| 0`
    if (rc.bad) fd = 0.0;
    if (fabs(fd) >= srd) fd = fmod(fd, srd);
    int E = 0;
    if (fd != 0.0) E = min(E, lsb_exponent(fd));
    if (A.phase0 != 0.0) E = min(E, lsb_exponent(A.phase0));
    const int Emin = -(62 - (32 - __builtin_clz(sr)));  // keep S = sr * 2^-E below 2^62
    if (E < Emin) E = Emin;                              // (only reachable for |f| < 2^-13: inexact regime)
    rc.E = E;
    rc.u = ldexp(1.0, E);
    rc.S = (uint64_t)sr << (-E);
    const long long F = (long long)rint(ldexp(fd, -E));
    rc.Fm = F >= 0 ? (uint64_t)F : rc.S - (uint64_t)(-F);
    if (rc.Fm >= rc.S) rc.Fm -= rc.S;
    rc.P0 = (uint64_t)rint(ldexp(A.phase0, -E));
    if (rc.P0 >= rc.S) rc.P0 %= rc.S;
    rc.step4 = mulmod(rc.Fm, 4, rc.S);
    rc.step256 = mulmod(rc.Fm, kChunk, rc.S);
    rc.segstep = mulmod(rc.Fm, (uint64_t)A.seg_groups * kChunk, rc.S);
    rc.gain = A.gain.kind == SRC_PARAM || A.gain.kind == SRC_CONST ? operand_value(A.gain, A.params, A.n_inst, inst) : 1.f;
    rc.pad = 0;
    recs[inst] = rc;

    // the state every unit holds after ceil(n_samples/256) ticks
    const uint64_t T_end = (uint64_t)A.n_chunks * kChunk;
    double phase_end = (double)addmod(rc.P0, mulmod(rc.Fm, T_end, rc.S), rc.S) * rc.u;
    if (rc.bad) phase_end = __builtin_nan("");
    A.end_state[(size_t)A.osc_state_word * A.n_inst + inst] = phase_end;
    if (A.ramp_state_word >= 0) {
        const double t_end = A.r_playing ? fmin(A.r_t0 + (double)T_end, A.r_d) : A.r_t0;
        const bool playing_end = A.r_playing && (A.r_t0 + (double)T_end <= A.r_d);
        A.end_state[(size_t)A.ramp_state_word * A.n_inst + inst] = t_end;
        A.end_state[(size_t)(A.ramp_state_word + 1) * A.n_inst + inst] = playing_end ? 1.0 : 0.0;
    }
    if (A.shape_state_word >= 0) {  // t keeps running while playing; finish() once a tick sees t > sampleRate (Shape/index.js:40-42)
        const double t_end = A.s_playing ? repeat_add(A.s_t0, A.s_c, T_end) : A.s_t0;
        A.end_state[(size_t)A.shape_state_word * A.n_inst + inst] = t_end;
        A.end_state[(size_t)(A.shape_state_word + 1) * A.n_inst + inst] = A.s_playing ? 1.0 : 0.0;
        A.end_state[(size_t)(A.shape_state_word + 2) * A.n_inst + inst] = (A.s_finished || (T_end > 0 && t_end > srd)) ? 1.0 : 0.0;
    }
}

// KIND: fused shape.  TBL: table placement.  R: voices per work item.  FASTDIV / FINITE:
// host-verified strength reductions (ramp_value, fix_out).
template <int KIND, int TBL, int R, bool FASTDIV, bool FINITE, int BLOCK>
__global__ void __launch_bounds__(BLOCK) dusp_fused_kernel(FusedArgs A, const OscRec *__restrict__ recs) {
    extern __shared__ __attribute__((aligned(16))) float lds_table[];
    Table<TBL> table;
    table.g = A.table;
    table.h = lds_table;
    table.N = A.sample_rate + 1;
    table.M = A.sample_rate / 2;
    table.form = make_table_form(A.table_form, A.sample_rate);
    if (TBL == 1) load_half_table<BLOCK>(lds_table, A.table, A.sample_rate);
    const uint32_t lane = threadIdx.x & 63u;
    constexpr uint32_t waves_per_block = BLOCK / 64;
    constexpr int RS = R < 4 ? R : 4;  // voices handled together by the FX32 / F64 paths (register budget)
    const uint32_t n_blk = (A.n_inst + R - 1) / R;
    const uint64_t n_items = (uint64_t)n_blk * A.n_seg;
    const uint64_t total_waves = (uint64_t)gridDim.x * waves_per_block;
    const double srd = (double)A.sample_rate;
    const uint32_t sr = A.sample_rate;
    const double dy = A.r_y1 - A.r_y0;
    // full 256-sample groups may use unguarded 16-byte stores; a trailing partial group may not
    const uint32_t n_full = A.vec4_ok ? (uint32_t)(A.n_samples / kChunk) : 0u;

    for (uint64_t item = (uint64_t)blockIdx.x * waves_per_block + (threadIdx.x >> 6); item < n_items; item += total_waves) {
        const uint32_t blk = A.seg_major ? (uint32_t)(item / A.n_seg) : (uint32_t)(item % n_blk);
        const uint32_t seg = A.seg_major ? (uint32_t)(item % A.n_seg) : (uint32_t)(item / n_blk);
        const uint32_t g0 = seg * A.seg_groups;
        const uint32_t g1 = min(g0 + A.seg_groups, A.n_groups);
        const uint32_t gm = max(g0, min(g1, n_full));
        const uint64_t t_start = (uint64_t)g0 * kChunk;

        // a short last block repeats its last voice: same data stored to the same address, no branches
        uint32_t inst[R];
        int e_min = 0;
        bool any_bad = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            inst[r] = min(blk * R + r, A.n_inst - 1);
            e_min = min(e_min, recs[inst[r]].E);
            any_bad = any_bad || recs[inst[r]].bad;
        }
        float *const row0 = A.out + t_start + lane * 4;
        const double tn0 = A.r_t0 + (A.r_playing ? (double)(t_start + lane * 4 + 1) : 0.0);
        const double tn_step = A.r_playing ? (double)kChunk : 0.0;
        const double tn_c = A.r_playing ? 1.0 : 0.0;
        double tn;  // Ramp: t0 + (n + 1) for this lane's first sample n of the current step; idle ramp: t0
        // Shape: t after the samples before this item (closed form), then step by step — as integers while a whole step
        // stays inside one binade, else every lane jumps to its own samples
        const double st0 = (KIND == FUSED_OSC_SHAPE && A.s_playing) ? repeat_add(A.s_t0, A.s_c, t_start) : A.s_t0;
        double st;
        auto ramp4 = [&](float (&rv)[4]) {
            if (KIND == FUSED_OSC_SHAPE) {
                if (!A.s_playing) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) rv[c] = shape_value(A, st, srd);
                } else {
                    long long T, ce;
                    int K;
                    if (linear_run(st, A.s_c, kChunk, T, ce, K)) {
                        T += (long long)(lane * 4) * ce;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            T += ce;
                            rv[c] = shape_value(A, ldexp((double)T, K - 52), srd);
                        }
                        st = ldexp((double)(T + (long long)(kChunk - 4 - lane * 4) * ce), K - 52);  // t after the step's 256 samples
                    } else {
                        double t = repeat_add(st, A.s_c, (uint64_t)lane * 4);
#pragma unroll
                        for (int c = 0; c < 4; ++c) rv[c] = shape_value(A, t = t + A.s_c, srd);
                        st = __shfl(t, 63, 64);
                    }
                }
            }
            if (KIND == FUSED_OSC_RAMP) {
                rv[0] = ramp_value<FASTDIV>(A, tn, dy);
#pragma unroll
                for (int c = 1; c < 4; ++c) rv[c] = ramp_value<FASTDIV>(A, tn + tn_c * c, dy);
                tn += tn_step;
            }
        };
        auto finish = [&](float v, float rv, float gain) {
            if (KIND == FUSED_OSC_RAMP || KIND == FUSED_OSC_SHAPE) v = v * rv;
            if (KIND == FUSED_OSC_GAIN) v = v * gain;
            return fix_out<FINITE>(v);
        };

        if (e_min == 0 && !any_bad) {
            // ---- INT: every phase is an integer, out = table[phase] exactly
            uint32_t idx[R][4], step[R];
            size_t roff[R];
            float gain[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const OscRec rc = recs[inst[r]];
                uint64_t P[4];
                start_phase(rc, seg, lane, P);
                for (int c = 0; c < 4; ++c) idx[r][c] = (uint32_t)P[c];
                step[r] = (uint32_t)rc.step256;
                roff[r] = (size_t)inst[r] * A.n_samples;
                gain[r] = rc.gain;
            }
            tn = tn0;
            st = st0;
            float *row = row0;
            auto run = [&](uint32_t ga, uint32_t gb, auto vec) {
                for (uint32_t g = ga; g < gb; ++g) {
                    float rv[4] = {1.f, 1.f, 1.f, 1.f};
                    ramp4(rv);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        float v[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            v[c] = finish(table.at(idx[r][c]), rv[c], gain[r]);
                            idx[r][c] += step[r];
                            idx[r][c] = min(idx[r][c], idx[r][c] - sr);  // wrap (idx < 2 sr; underflow loses the min)
                        }
                        store4<decltype(vec)::value>(row + roff[r], v, (uint64_t)g * kChunk + lane * 4, A.n_samples);
                    }
                    row += kChunk;
                }
            };
            run(g0, gm, std::true_type{});
            run(gm, g1, std::false_type{});
        } else if (e_min >= -32 && !any_bad && A.fx32_ok) {
            // ---- FX32: phase = I + F 2^-32.  Advance with add-with-carry; lerp on exact f64 integers:
            // table[I] (2^32 - F) + table[I+1] F, each product / the sum rounded exactly like the reference's
            // a*(1-fraction) + b*fraction scaled by 2^32 (a power of two commutes with rounding), then one
            // f32 rounding and an exact 2^-32 (results stay far above the f32 subnormal range: the host
            // checks min |table| >= 2^-20 before setting fx32_ok).
            for (int sub = 0; sub < R; sub += RS) {
                uint32_t I[RS][4], F[RS][4], dI[RS], dF[RS];
                size_t roff[RS];
                float gain[RS];
#pragma unroll
                for (int r = 0; r < RS; ++r) {
                    const OscRec rc = recs[inst[sub + r]];
                    const int sh = -rc.E;  // 0..32
                    uint64_t P[4];
                    start_phase(rc, seg, lane, P);
                    for (int c = 0; c < 4; ++c) {
                        I[r][c] = (uint32_t)(P[c] >> sh);
                        F[r][c] = sh ? (uint32_t)(P[c] << (32 - sh)) : 0u;
                    }
                    dI[r] = (uint32_t)(rc.step256 >> sh);
                    dF[r] = sh ? (uint32_t)(rc.step256 << (32 - sh)) : 0u;
                    roff[r] = (size_t)inst[sub + r] * A.n_samples;
                    gain[r] = rc.gain;
                }
                tn = tn0;
            st = st0;
                float *row = row0;
                // Lean form, where it is the same arithmetic: (1) a phase grid of 2^-28 or coarser (lsb(f) >= 2^-28: e.g. the k / 8 sweep
                // of BASELINE configs[4]) makes both products of the lerp EXACT in f64 — a 24-bit table entry times a weight of at most 29
                // bits — so the reference's three roundings are one, and fma(tb, wb, ta wa) is it; (2) the envelope takes the exact 2^-32
                // with it (a Ramp between values that are zero or of ordinary magnitude — A.r_scale_ok, checked on the host — has no f32
                // value that 2^-32 would push below the normal range), so a product is one rounding like the Multiply unit's; (3) products and
                // the `|| 0` additions go two at a time (v_pk_mul_f32 / v_pk_add_f32).
                // Delta form (device_util.hpp lerp_delta), where the phase grid is 2^-28 or coarser and the table's neighbours differ by exact
                // values (A.table_delta: 2 in f32, 1 in f64; the LDS image or the gathered table): T[i] + (T[i+1] - T[i]) fraction in ONE fma is
                // the reference's three roundings, whatever the shape; index.fraction is one 64-bit integer that is also a double (device_util.hpp
                // LeanPhase): the lane's four phases are four v_lshl_add_u64 off its first (q, 2q, 3q and the step's 256q as 32.32 integers,
                // wave-uniform), the fraction is v_fract_f64 of it, the wrap and the fold into the half image are two v_sad_u32.
                if ((TBL == 0 || TBL == 1) && e_min >= -28 && A.table_delta) {
                    unsigned long long PD[RS], Q1[RS], Q2[RS], Q3[RS], C[RS];
                    auto uni = [](unsigned long long v) {
                        return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
                    };
#pragma unroll
                    for (int r = 0; r < RS; ++r) {
                        const OscRec rc = recs[inst[sub + r]];
                        const int up = 32 + rc.E;  // phases are integers in units of 2^E, -28 <= E <= 0: as 32.32 integers, shifted up
                        PD[r] = (((unsigned long long)(I[r][0] + kPhaseBias)) << 32) | F[r][0];
                        const uint64_t q2 = addmod(rc.Fm, rc.Fm, rc.S);
                        Q1[r] = uni(rc.Fm << up);
                        Q2[r] = uni(q2 << up);
                        Q3[r] = uni(addmod(q2, rc.Fm, rc.S) << up);
                        C[r] = uni(rc.step256 << up);
                    }
                    auto run_delta = [&](uint32_t ga, uint32_t gb, auto vec, auto d32) {
                        for (uint32_t g = ga; g < gb; ++g) {
                            float rv[4] = {1.f, 1.f, 1.f, 1.f};
                            ramp4(rv);
#pragma unroll
                            for (int r = 0; r < RS; ++r) {
                                float v[4];
                                double ta[4], td[4], fr[4];
#pragma unroll
                                for (int c = 0; c < 4; ++c) {
                                    const unsigned long long Pc = c == 0 ? PD[r] : PD[r] + (c == 1 ? Q1[r] : c == 2 ? Q2[r] : Q3[r]);
                                    if (TBL == 1) {
                                        uint32_t a_img;
                                        bool upper;
                                        if (c == 0) LeanPhase::locate<true>(Pc, sr, table.M, a_img, upper, fr[c]);
                                        else LeanPhase::locate<false>(Pc, sr, table.M, a_img, upper, fr[c]);
                                        table.template pair_delta_at<decltype(d32)::value>(a_img, upper, ta[c], td[c]);
                                    } else {
                                        fr[c] = __builtin_amdgcn_fract(__longlong_as_double((long long)Pc));
                                        table.template pair_delta<decltype(d32)::value>(LeanPhase::index(Pc, sr), ta[c], td[c]);
                                    }
                                }
#pragma unroll
                                for (int c = 0; c < 4; ++c) v[c] = finish((float)fma(td[c], fr[c], ta[c]), rv[c], gain[r]);
                                PD[r] = LeanPhase::step(PD[r], C[r], sr);
                                store4<decltype(vec)::value>(row + roff[r], v, (uint64_t)g * kChunk + lane * 4, A.n_samples);
                            }
                            row += kChunk;
                        }
                    };
                    if (A.table_delta == 2) {
                        run_delta(g0, gm, std::true_type{}, std::true_type{});
                        run_delta(gm, g1, std::false_type{}, std::true_type{});
                    } else {
                        run_delta(g0, gm, std::true_type{}, std::false_type{});
                        run_delta(gm, g1, std::false_type{}, std::false_type{});
                    }
                    continue;
                }
                const bool lean = KIND == FUSED_OSC_RAMP && FINITE && e_min >= -28 && A.r_scale_ok;
                auto run_lean = [&](uint32_t ga, uint32_t gb, auto vec) {
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    for (uint32_t g = ga; g < gb; ++g) {
                        float rv[4] = {1.f, 1.f, 1.f, 1.f};
                        ramp4(rv);
                        const f32x2 e01 = f32x2{ldexpf(rv[0], -32), ldexpf(rv[1], -32)}, e23 = f32x2{ldexpf(rv[2], -32), ldexpf(rv[3], -32)};
#pragma unroll
                        for (int r = 0; r < RS; ++r) {
                            float x[4];
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                float ta, tb;
                                table.pair(I[r][c], ta, tb);
                                const double wb = (double)F[r][c];
                                const double wa = 4294967296.0 - wb;
                                x[c] = (float)fma((double)tb, wb, (double)ta * wa);
                                const uint32_t f2 = F[r][c] + dF[r];
                                uint32_t i2 = I[r][c] + dI[r] + (f2 < F[r][c] ? 1u : 0u);
                                i2 = min(i2, i2 - sr);
                                F[r][c] = f2;
                                I[r][c] = i2;
                            }
                            const f32x2 zero = f32x2{0.f, 0.f};
                            const f32x2 p01 = f32x2{x[0], x[1]} * e01 + zero, p23 = f32x2{x[2], x[3]} * e23 + zero;
                            const float v[4] = {p01[0], p01[1], p23[0], p23[1]};
                            store4<decltype(vec)::value>(row + roff[r], v, (uint64_t)g * kChunk + lane * 4, A.n_samples);
                        }
                        row += kChunk;
                    }
                };
                if (lean) {
                    run_lean(g0, gm, std::true_type{});
                    run_lean(gm, g1, std::false_type{});
                    continue;
                }
                auto run = [&](uint32_t ga, uint32_t gb, auto vec) {
                    for (uint32_t g = ga; g < gb; ++g) {
                        float rv[4] = {1.f, 1.f, 1.f, 1.f};
                        ramp4(rv);
#pragma unroll
                        for (int r = 0; r < RS; ++r) {
                            float v[4];
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                float ta, tb;
                                table.pair(I[r][c], ta, tb);
                                const double wb = (double)F[r][c];
                                const double wa = 4294967296.0 - wb;
                                const float x = (float)((double)ta * wa + (double)tb * wb);
                                v[c] = finish(ldexpf(x, -32), rv[c], gain[r]);
                                const uint32_t f2 = F[r][c] + dF[r];
                                uint32_t i2 = I[r][c] + dI[r] + (f2 < F[r][c] ? 1u : 0u);
                                i2 = min(i2, i2 - sr);
                                F[r][c] = f2;
                                I[r][c] = i2;
                            }
                            store4<decltype(vec)::value>(row + roff[r], v, (uint64_t)g * kChunk + lane * 4, A.n_samples);
                        }
                        row += kChunk;
                    }
                };
                run(g0, gm, std::true_type{});
                run(gm, g1, std::false_type{});
            }
        } else {
            // ---- F64: general path
            for (int sub = 0; sub < R; sub += RS) {
                double ph[RS][4], D[RS];
                size_t roff[RS];
                float gain[RS];
                bool bad[RS];
#pragma unroll
                for (int r = 0; r < RS; ++r) {
                    const OscRec rc = recs[inst[sub + r]];
                    uint64_t P[4];
                    start_phase(rc, seg, lane, P);
                    for (int c = 0; c < 4; ++c) ph[r][c] = (double)P[c] * rc.u;
                    D[r] = (double)rc.step256 * rc.u;
                    roff[r] = (size_t)inst[sub + r] * A.n_samples;
                    gain[r] = rc.gain;
                    bad[r] = rc.bad;
                }
                tn = tn0;
            st = st0;
                float *row = row0;
                auto run = [&](uint32_t ga, uint32_t gb, auto vec) {
                    for (uint32_t g = ga; g < gb; ++g) {
                        float rv[4] = {1.f, 1.f, 1.f, 1.f};
                        ramp4(rv);
#pragma unroll
                        for (int r = 0; r < RS; ++r) {
                            float v[4];
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const uint32_t idx = (uint32_t)(int)ph[r][c];
                                const double fraction = ph[r][c] - (double)(int)idx;
                                float ta, tb;
                                table.pair(idx, ta, tb);
                                float x = (float)((double)ta * (1.0 - fraction) + (double)tb * fraction);
                                if (bad[r]) x = 0.f;
                                v[c] = finish(x, rv[c], gain[r]);
                                ph[r][c] += D[r];
                                if (ph[r][c] >= srd) ph[r][c] -= srd;
                            }
                            store4<decltype(vec)::value>(row + roff[r], v, (uint64_t)g * kChunk + lane * 4, A.n_samples);
                        }
                        row += kChunk;
                    }
                };
                run(g0, gm, std::true_type{});
                run(gm, g1, std::false_type{});
            }
        }
    }
}

template <int KIND, int TBL, int R, bool FASTDIV, bool FINITE, int BLOCK>
static hipError_t launch_one(const FusedArgs &A, const OscRec *recs, int grid, size_t lds_bytes, hipStream_t stream) {
    auto kernel = dusp_fused_kernel<KIND, TBL, R, FASTDIV, FINITE, BLOCK>;
    if (lds_bytes > 65536) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(BLOCK), lds_bytes, stream, A, recs);
    return hipGetLastError();
}

hipError_t launch_fused(const FusedPlan &plan, const FusedLaunch &L, hipStream_t stream) {
    FusedArgs A{};
    A.params = L.params;
    A.table = L.tables + (size_t)plan.table_id * L.table_stride;
    A.out = L.out;
    A.end_state = L.end_state;
    A.n_samples = L.n_samples;
    A.n_inst = L.n_inst;
    A.n_groups = (uint32_t)((L.n_samples + kChunk - 1) / kChunk);
    A.sample_rate = L.sample_rate;
    A.n_chunks = L.n_chunks;
    A.f = plan.f;
    A.gain = plan.gain;
    if (plan.kind != FUSED_OSC_GAIN) A.gain.kind = -1;
    A.phase0 = plan.phase0;
    A.r_d = plan.r_d; A.r_y0 = plan.r_y0; A.r_y1 = plan.r_y1; A.r_t0 = plan.r_t0; A.r_rcp = plan.r_rcp;
    A.r_playing = plan.r_playing;
    A.r_fastdiv = plan.r_fastdiv;
    A.vec4_ok = (L.n_samples % 4 == 0) && (((uintptr_t)L.out & 15) == 0);
    A.osc_state_word = 0;
    A.ramp_state_word = plan.kind == FUSED_OSC_RAMP ? 1 : -1;
    A.shape_state_word = plan.kind == FUSED_OSC_SHAPE ? 1 : -1;
    if (plan.kind == FUSED_OSC_SHAPE) {
        A.s_table = L.tables + (size_t)plan.s_table_id * L.table_stride;
        A.s_t0 = plan.s_t0; A.s_c = plan.s_c; A.s_min = plan.s_min; A.s_max = plan.s_max;
        A.s_left = plan.s_left; A.s_right = plan.s_right;
        A.s_left_is_shape = plan.s_left_is_shape; A.s_right_is_shape = plan.s_right_is_shape;
        A.s_playing = plan.s_playing; A.s_finished = plan.s_finished;
    }
    // (the envelope may carry an exact 2^-32: every Ramp value is f32(y0 + q (y1 - y0)) with 0 <= q <= 1 — zero, or no smaller than the
    // rounding grain of the larger end, far above 2^-94)
    auto ordinary = [](double y) { return y == 0.0 || (std::fabs(y) >= 9.3e-10 && std::fabs(y) <= 1.1e18); };
    A.r_scale_ok = plan.kind == FUSED_OSC_RAMP && ordinary(plan.r_y0) && ordinary(plan.r_y1) && plan.r_d <= 1.1e12 && (plan.r_t0 == 0.0 || plan.r_t0 >= 1.0) &&
                           L.knobs.fused_fx32 != 2  // (DUSP_FUSED_FX32=2: A/B, the plain form)
                       ? 1 : 0;
    A.fx32_ok = L.table_fx32_ok && L.knobs.fused_fx32 ? 1 : 0;  // (knobs: device_types.hpp, read when the context was created)
    A.table_delta = L.knobs.fused_fx32 == 1 ? L.table_delta : 0;  // (DUSP_FUSED_FX32=2 / =3: A/B, the plain / the lean form)
    A.seg_major = L.knobs.fused_segmajor;

    int tbl = (L.table_antisym && L.sample_rate % 2 == 0) ? 1 : 0;
    if (L.table_form >= TABLE_FORM_SAW && L.table_form <= TABLE_FORM_TRIANGLE) tbl = 2;  // no table at all: a closed form of the index
    A.table_form = L.table_form;
    if (L.knobs.fused_table_global) tbl = 0;
    const size_t lds_bytes = tbl == 1 ? half_table_lds_bytes(L.sample_rate) : 0;
    if (lds_bytes > 160 * 1024) tbl = 0;

    // R voices per item share one Ramp evaluation per step; without a Ramp there is nothing to share
    int R = L.knobs.fused_R;
    if (R != 1 && R != 4 && R != 8) R = 4;
    // Items: (voice block, segment).  Aim at `per_wave` equal items for every resident wave so that all
    // waves finish together; segments of at least 16 steps keep the per-item jump-ahead negligible.
    const int block = tbl == 1 ? 1024 : 256;
    const int grid = tbl == 1 ? L.n_cus : L.n_cus * 8;
    const uint64_t total_waves = (uint64_t)grid * (block / 64);
    const uint64_t n_blk = (A.n_inst + R - 1) / R;
    // Four items per wave even the tail out on long renders; a short render (configs[4]'s shard: 8192 voices x 188 chunks) would
    // be cut into items of a couple of dozen chunks each, and what an item costs before its first sample (the jump-ahead, the
    // table image is per workgroup) shows: two items per wave there (0.398 -> 0.377 ms).
    auto cut = [&](uint64_t per_wave, uint64_t &n_seg) {
        n_seg = std::max<uint64_t>(1, (total_waves * per_wave + n_blk - 1) / n_blk);
        return (A.n_groups + n_seg - 1) / n_seg;
    };
    uint64_t n_seg = 1, seg = 0;
    if (L.knobs.fused_items > 0) seg = cut((uint64_t)L.knobs.fused_items, n_seg);
    else if ((seg = cut(4, n_seg)) < 64) seg = cut(2, n_seg);
    if (seg < 16) seg = 16;
    if (seg > 4096) seg = 4096;
    A.seg_groups = (uint32_t)seg;
    A.n_seg = (A.n_groups + A.seg_groups - 1) / A.seg_groups;
    const bool finite = L.table_finite && plan.kind != FUSED_OSC_GAIN && std::isfinite(plan.r_y0) && std::isfinite(plan.r_y1);

    hipLaunchKernelGGL(dusp_fused_prepare, dim3((A.n_inst + 255) / 256), dim3(256), 0, stream, A, L.recs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;

#define DUSP_L3(KIND, RR, FD, FIN) \
    return tbl == 1 ? launch_one<KIND, 1, RR, FD, FIN, 1024>(A, L.recs, grid, lds_bytes, stream) \
                    : launch_one<KIND, 0, RR, FD, FIN, 256>(A, L.recs, grid, 0, stream)
    // (the closed-form variants exist for 4 voices per item only)
#define DUSP_L2(KIND, FD, FIN) \
    do { if (tbl == 2) return launch_one<KIND, 2, 4, FD, FIN, 256>(A, L.recs, grid, 0, stream); \
         if (R == 8) { DUSP_L3(KIND, 8, FD, FIN); } if (R == 4) { DUSP_L3(KIND, 4, FD, FIN); } DUSP_L3(KIND, 1, FD, FIN); } while (0)
    switch (plan.kind) {
    case FUSED_OSC:
        if (finite) DUSP_L2(FUSED_OSC, false, true);
        DUSP_L2(FUSED_OSC, false, false);
    case FUSED_OSC_RAMP:
        if (plan.r_fastdiv && finite) DUSP_L2(FUSED_OSC_RAMP, true, true);
        if (plan.r_fastdiv) DUSP_L2(FUSED_OSC_RAMP, true, false);
        DUSP_L2(FUSED_OSC_RAMP, false, false);
    case FUSED_OSC_GAIN: DUSP_L2(FUSED_OSC_GAIN, false, false);
    case FUSED_OSC_SHAPE: DUSP_L2(FUSED_OSC_SHAPE, false, false);
    }
#undef DUSP_L2
#undef DUSP_L3
    return hipErrorInvalidValue;
}

}  // namespace dusp
