// fused_plan.hpp — host side of the fused (time-parallel) engine: matches a
// compiled program against the graph shapes fused_engine.hip instantiates and
// gathers the scalars the kernel needs.
//
// A graph is fusable when it is a feed-forward TREE that covers every unit of
// the circuit, every outlet is mono, and it has one of the signatures below
// (operands of Multiply may come in either order — f32 multiply commutes):
//     osc(k)               Osc with an unconnected (constant / per-instance) f
//     mul(osc(k),ramp)     the per-voice graph of BASELINE configs[2] / [4]
//     mul(osc(k),k)        Osc times a constant / per-instance gain
// Everything else runs on the chunk engine.
#pragma once
#include <cmath>
#include <string>
#include <vector>

#include "device_types.hpp"
#include "program.hpp"

namespace dusp {

enum : int { FUSED_OSC = 0, FUSED_OSC_RAMP = 1, FUSED_OSC_GAIN = 2 };

struct FusedPlan {
    std::string shape, why;
    int kind = FUSED_OSC;
    int table_id = 0;
    DevOperand f{}, gain{};
    double phase0 = 0;
    // Ramp (Ramp.js:3-14): duration, y0, y1, initial t / playing; rcp = RN(1/duration)
    double r_d = 1, r_y0 = 0, r_y1 = 0, r_t0 = 0, r_rcp = 1;
    int r_playing = 0, r_fastdiv = 0;
    // state write-back: words [first, first+count) of the end-state block belong to unit u
    int n_state_words = 0;
    std::vector<int> unit_state_first, unit_state_count;
};

// Per-voice oscillator record, written by the prepare kernel: f in exact fixed point (units of 2^E)
// plus the strides the render kernel steps by.
struct OscRec {
    uint64_t S, Fm, P0;               // modulus sr * 2^-E, increment f * 2^-E mod S, start phase
    uint64_t step4, step256, segstep; // (4 f), (256 f), (segment length * f) mod S
    double u;                         // 2^E
    int32_t E, bad;                   // bad: non-finite f (every sample NaN -> 0)
    float gain, pad;
};

// Launch-time arguments filled in by dusp_render_device.
struct FusedLaunch {
    const float *params;
    const float *tables;
    float *out;
    double *end_state;
    OscRec *recs;  // [n_inst] workspace
    uint64_t n_samples;
    uint32_t n_inst, n_chunks, sample_rate, table_stride;
    int n_cus;
    bool table_antisym, table_finite, table_fx32_ok;
};

// Kernel arguments (by value).
struct FusedArgs {
    const float *params;
    const float *table;  // row of the Osc's wave table (sample_rate + 2 entries)
    float *out;
    double *end_state;   // [n_state_words][n_inst]
    uint64_t n_samples;
    uint32_t n_inst, n_groups, seg_groups, n_seg, sample_rate, n_chunks;
    DevOperand f, gain;
    double phase0;
    double r_d, r_y0, r_y1, r_t0, r_rcp;
    int32_t r_playing, r_fastdiv, vec4_ok, osc_state_word, ramp_state_word, fx32_ok;
};

// q' of Markstein's division-by-reciprocal: q = t*r; rem = fma(-q, d, t); q' = fma(rem, r, q).
// It equals the correctly rounded t/d for almost every operand pair but not provably all, so
// the host CHECKS it over the exact sequence of t values a Ramp will produce before the
// kernel is allowed to use it (otherwise the kernel divides).
inline bool ramp_fastdiv_ok(double t0, double d, bool playing) {
    const double r = 1.0 / d;
    auto same = [&](double t) {
        const double q = t * r;
        const double rem = std::fma(-q, d, t);
        return std::fma(rem, r, q) == t / d;
    };
    if (!same(t0) || !same(d)) return false;
    if (!playing) return true;
    const double steps = std::ceil(d - t0) + 1;
    if (!(steps < 67108864.0)) return false;
    double t = t0;
    for (long k = 0; k < (long)steps; k++) {
        t += 1.0;
        if (t > d) break;
        if (!same(t)) return false;
    }
    return true;
}

inline bool plan_fused(const Program &P, FusedPlan &plan) {
    const Graph &g = P.g;
    auto no = [&](const std::string &why) {
        plan.why = why;
        return false;
    };
    if (!P.feed_forward) return no("feedback edge");
    if (!g.rings.empty() || P.ring_samples) return no("rings");
    if (P.out_bufs.size() != 1) return no("multichannel output");
    for (const auto &u : g.units)
        if (u.n_out != 1) return no("multichannel unit");

    std::vector<int> used(g.units.size(), 0);
    int osc_unit = -1, ramp_unit = -1;
    bool have_gain = false;
    auto leaf_k = [&](const InletDesc &in) { return in.kind != IN_CONNECT && in.vals.size() == 1; };
    auto take_osc = [&](int ui) {
        const UnitDesc &u = g.units[(size_t)ui];
        if (u.op != OP_OSC || !leaf_k(u.inlets[0]) || osc_unit >= 0) return false;
        osc_unit = ui;
        used[(size_t)ui]++;
        return true;
    };
    const UnitDesc &root = g.units[(size_t)g.out_unit];
    used[(size_t)g.out_unit]++;
    if (root.op == OP_OSC) {
        used[(size_t)g.out_unit]--;
        if (!take_osc(g.out_unit)) return no("Osc with a connected f");
        plan.kind = FUSED_OSC;
        plan.shape = "osc(k)";
    } else if (root.op == OP_MULTIPLY) {
        for (int k = 0; k < 2; k++) {
            const InletDesc &in = root.inlets[(size_t)k];
            if (in.kind == IN_CONNECT) {
                const UnitDesc &s = g.units[(size_t)in.src_unit];
                if (s.op == OP_OSC) {
                    if (!take_osc(in.src_unit)) return no("second Osc / connected f");
                } else if (s.op == OP_RAMP && ramp_unit < 0) {
                    ramp_unit = in.src_unit;
                    used[(size_t)in.src_unit]++;
                } else
                    return no("unsupported Multiply operand");
            } else if (leaf_k(in) && !have_gain) {
                have_gain = true;
                plan.gain = make_operand(g, in, 0);
            } else
                return no("unsupported Multiply operand");
        }
        if (osc_unit < 0) return no("no Osc under Multiply");
        if (ramp_unit >= 0) {
            plan.kind = FUSED_OSC_RAMP;
            plan.shape = "mul(osc(k),ramp)";
        } else if (have_gain) {
            plan.kind = FUSED_OSC_GAIN;
            plan.shape = "mul(osc(k),k)";
        } else
            return no("Multiply of two Oscs");
    } else
        return no("root is neither Osc nor Multiply");
    for (size_t i = 0; i < used.size(); i++)
        if (used[i] != 1) return no("circuit has units outside the fused tree");

    const UnitDesc &osc = g.units[(size_t)osc_unit];
    plan.table_id = (int)osc.attrs[0];
    plan.f = make_operand(g, osc.inlets[0], 0);
    plan.phase0 = osc.state[0];
    if (!(plan.phase0 >= 0 && plan.phase0 < g.sample_rate)) return no("Osc phase outside [0, sampleRate)");
    {   // the fixed-point jump-ahead needs the start phase on a 2^-36 grid (0 in practice)
        const double scaled = std::ldexp(plan.phase0, 36);
        if (scaled != std::floor(scaled)) return no("Osc start phase finer than 2^-36");
    }
    if (ramp_unit >= 0) {
        const UnitDesc &r = g.units[(size_t)ramp_unit];
        plan.r_d = r.attrs[0];
        plan.r_y0 = r.attrs[1];
        plan.r_y1 = r.attrs[2];
        plan.r_t0 = r.state[0];
        plan.r_playing = r.state[1] != 0;
        if (!(plan.r_d > 0) || !std::isfinite(plan.r_d) || !std::isfinite(plan.r_t0)) return no("degenerate Ramp duration");
        if (plan.r_playing && !(plan.r_t0 + 1 >= 0)) return no("Ramp starts below -1");
        if (plan.r_t0 > plan.r_d) return no("Ramp starts past its end");
        plan.r_rcp = 1.0 / plan.r_d;
        plan.r_fastdiv = ramp_fastdiv_ok(plan.r_t0, plan.r_d, plan.r_playing != 0) ? 1 : 0;
    }
    plan.unit_state_first.assign(g.units.size(), 0);
    plan.unit_state_count.assign(g.units.size(), 0);
    plan.n_state_words = 0;
    plan.unit_state_first[(size_t)osc_unit] = plan.n_state_words;
    plan.unit_state_count[(size_t)osc_unit] = 1;
    plan.n_state_words += 1;
    if (ramp_unit >= 0) {
        plan.unit_state_first[(size_t)ramp_unit] = plan.n_state_words;
        plan.unit_state_count[(size_t)ramp_unit] = 2;
        plan.n_state_words += 2;
    }
    return true;
}

}  // namespace dusp
