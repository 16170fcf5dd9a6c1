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
//     mul(osc(k),shape)    Osc under a Shape envelope with constant duration / min / max: the canonical Dusp voice "O440 * D1"
// Everything else runs on the chunk engine.
#pragma once
#include <cmath>
#include <string>
#include <vector>

#include "device_types.hpp"
#include "program.hpp"

namespace dusp {

enum : int { FUSED_OSC = 0, FUSED_OSC_RAMP = 1, FUSED_OSC_GAIN = 2, FUSED_SUMCHAIN = 3, FUSED_OSC_SHAPE = 4 };

struct FusedPlan {
    std::string shape, why;
    int kind = FUSED_OSC;
    int table_id = 0;
    DevOperand f{}, gain{};
    double phase0 = 0;
    // Ramp (Ramp.js:3-14): duration, y0, y1, initial t / playing; rcp = RN(1/duration)
    double r_d = 1, r_y0 = 0, r_y1 = 0, r_t0 = 0, r_rcp = 1;
    int r_playing = 0, r_fastdiv = 0;
    // Shape (Shape/index.js:7-59) with constant inlets: table, start t, addend 1/duration, range, edges, flags
    int s_table_id = 0, s_playing = 0, s_finished = 0, s_left_is_shape = 0, s_right_is_shape = 0;
    double s_t0 = 0, s_c = 0, s_left = 0, s_right = 0;
    float s_min = 0, s_max = 1;
    // FUSED_SUMCHAIN: the oscillators of a left-deep Sum.many chain, in chain order
    std::vector<double> sum_f, sum_phase0;
    std::vector<int> sum_units;
    bool sum_all_int = false;
    // ... whose voices may be enveloped: every voice Multiply(Osc, k) (sum_env 1: sum_gain per voice) or every voice
    // Multiply(Osc, Ramp) with Ramps that are equal in constants and state (sum_env 2: the r_* fields above describe them all)
    int sum_env = 0;
    std::vector<float> sum_gain;
    std::vector<int> sum_ramp_units;
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

// One oscillator of a Sum.many chain in exact 32.32 fixed point (units of 2^-32 table steps):
// phase of (block b, lane l, sample c) = (B0 + b*bs + l*step4 + c*Fm) mod (sampleRate << 32).
struct SumVoice {
    uint64_t B0, bs, step4, step256, Fm;
    float gain;  // enveloped chains of kind 1: the voice's Multiply constant
    uint32_t pad;
};

struct SumArgs {
    const SumVoice *voices;
    const float *table;
    float *out;
    uint64_t n_samples, S;
    double inv_S, inv_sr;
    uint32_t n_voices, n_inst, n_groups, n_blocks, sample_rate, vec4_ok;
    double r_d, r_y0, r_y1, r_t0;  // the voices' common Ramp (kernels instantiated with ENV == 2)
    int32_t r_playing, r_fastdiv;
    // A window of the render that continues a chain (dusp_render_chain_window: the voices of one `Sum.many` dealt over several
    // GPUs): blocks first_block .. of the timeline go to `out`, whose first sample is the window's; the running sums start as
    // `init` (same layout as `out`) instead of zeros; raw: no `x || 0` at the copy-out (a partial sum that another rank continues).
    const float *init;
    uint32_t first_block, raw;
};

// Arguments of the wave engine's kernel (wave_engine.hip).
struct WaveArgs {
    const DevOp *ops;
    const int32_t *out_bufs;
    const float *params;
    const float *tables;
    float *out;
    double *state;             // [n_slots][n_pad]  end-of-render state (chunk engine's slot layout)
    const double *init_state;  // [n_slots]
    uint64_t n_samples;
    uint32_t n_ops, n_out, n_inst, n_pad, n_bufs, n_groups, sample_rate, table_stride, vec4_ok;
    int32_t lds_table_id;
    uint32_t table_bytes, wave_bytes;
    float *rings;            // [n_inst][ring_samples]  Delay rings (wave-engine layout)
    uint64_t ring_samples, clock0;
    uint32_t has_filter, scratch_bytes;  // scratch_bytes: per-wave LDS scratch the program's units need (0: none)
    const float *inputs;   // [n_inputs][n_inst][n_samples] host-generated signals (OP_INPUT)
    uint32_t n_state_ops;  // ops that own a block of LDS state (DevOp::lds_slot)
    uint32_t ring_events;  // the program has a delay line that needs ordered slot operations: the kernel variant that carries them
    uint32_t ext_units;    // what the units ask of the kernel variant: bit 0 Filter / Delay, bit 1 anything beyond Osc/Ramp/Multiply/Sum/Repeater/maps
    // time-split rendering (few instances, long render): every instance is cut into n_seg segments of seg_groups
    // chunks, one wavefront each.  seg_sum / seg_start: [n_ops][n_inst][n_seg] oscillator phase totals / start phases
    // in 2^-36 units (bit 63 = poisoned by a NaN / Inf increment).
    uint32_t n_seg, seg_groups;
    uint32_t pass_mode;   // 0: render; 1: only accumulate the phase totals of the oscillators of level pass_level
    uint32_t pass_level;
    unsigned long long *seg_sum, *seg_start;
    int32_t max_osc_level, pad4;
    // continuation (dusp_program_continue): chunk buffers survive between launches in saved_bufs [n_inst][n_bufs][256]
    // (a feedback edge reads its producer's PREVIOUS chunk); resume = this launch continues one
    float *saved_bufs;
    uint32_t save_bufs, resume;
    uint32_t n_params, param_bytes;  // the instance's parameter column is copied into LDS once (param_bytes per wave; 0: read from HBM)
};

// LDS one wave of the wave engine needs: chunk buffers + 12 doubles of state per op + the Filter scratch (P, b1, b2)
// (the 6 KB scratch is only needed by Filters whose cutoff is connected: per-sample b1 / b2 and their own P)
// the instance's parameter column, copied into the wave's LDS once (too many parameters: read from HBM instead)
inline size_t wave_param_bytes(size_t n_params) { return n_params && n_params <= 2048 ? (n_params * 4 + 15) & ~(size_t)15 : 0; }
inline size_t wave_lds_bytes(size_t n_bufs, size_t n_state_ops, size_t scratch_bytes) {
    return (n_bufs * 1024 + n_state_ops * 96 + (scratch_bytes ? scratch_bytes + 16 : 0) + 15) & ~(size_t)15;
}

struct WavePlan {
    bool ok = false, has_filter = false;
    int scratch_bytes = 0;  // per-wave LDS scratch: the largest any unit of the program asks for
    // LDS economy: chunk buffers are handed out by liveness (a feed-forward graph without rings needs a buffer only from its
    // producer to its last reader within the chunk), and only stateful ops own a state block
    std::vector<int> order;      // execution order of the device ops (a permutation; see plan_wave)
    std::vector<int> buf_slot;   // chunk buffer -> LDS slot
    std::vector<int> op_state;   // device op -> state block (-1: stateless)
    int n_slots = 0, n_state_ops = 0;
    int ext_units = 0;         // bit 0: Filter / Delay present, bit 1: units beyond the lean set (dusp_wave_kernel's FILT / EXT)
    bool ring_events = false;  // a delay line that goes through ordered slot operations (short / signal-rate Delay, MonoDelay, ReadBackDelay)
    bool splittable = false;   // only Osc / Ramp / stateless units, feed-forward: time can be cut into segments
    bool splittable_but_for_filters = false;  // ... the same but for its Filters, whose memory fades: segments that warm up (jit_codegen.hpp jit_warm_chunks)
    int max_osc_level = 0;     // an Osc's level = number of oscillators stacked in its f input (FM depth)
    std::vector<int> osc_level;  // per device op (-1: not an Osc)
    std::vector<int> ramp_fastdiv;  // per device op: 1 = t / duration may be computed with the verified 2-FMA reciprocal
    // (duration, smallest verified start t) of Ramps already checked: a continued render's t sequence is a suffix
    struct RampChecked { double d, t0; bool ok; };
    std::vector<RampChecked> ramp_checked;
    int lds_table_id = -1;
    std::string why;
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
    const float *chain_init = nullptr;  // sum chain only (dusp_render_chain_window): the sums another GPU's voices left, laid out like `out`
    uint64_t chain_first = 0;           // ... the window's first sample (a whole number of the launch's blocks)
    bool chain_raw = false;             // ... no `x || 0` at the copy-out (a partial sum)
    int table_delta;  // the lerp's delta form on this table (table_checks.hpp table_delta_class): 0 no, 1 differences in f64, 2 in f32
    int table_form;  // TABLE_FORM_* (device_util.hpp): saw / square / triangle are evaluated from the index instead of gathered
    Knobs knobs;
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
    int32_t r_playing, r_fastdiv, vec4_ok, osc_state_word, ramp_state_word, fx32_ok, seg_major, r_scale_ok;
    // FUSED_OSC_SHAPE
    const float *s_table;  // row of the Shape's table
    double s_t0, s_c, s_left, s_right;  // edges as values of the 0..1 shape, unless they are "shape" (= the table's end values)
    float s_min, s_max;
    int32_t s_playing, s_finished, shape_state_word, s_left_is_shape, s_right_is_shape;
    int32_t table_form;  // TABLE_FORM_* of the Osc's table (kernels instantiated with TBL == 2 evaluate it)
    int32_t table_delta; // the lerp's delta form on the Osc's table: 0 no, 1 differences of neighbours in f64, 2 in f32
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
    if (!P.warm_ops.empty()) return no("channel counts grow during the first chunks");
    if (!P.feed_forward) return no("feedback edge");
    if (!g.rings.empty() || P.ring_samples) return no("rings");
    if (P.out_bufs.size() != 1) return no("multichannel output");
    for (const auto &u : g.units)
        if (u.n_out != 1) return no("multichannel unit");

    std::vector<int> used(g.units.size(), 0);
    int osc_unit = -1, ramp_unit = -1, shape_unit = -1;
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
                } else if (s.op == OP_RAMP && ramp_unit < 0 && shape_unit < 0) {
                    ramp_unit = in.src_unit;
                    used[(size_t)in.src_unit]++;
                } else if (s.op == OP_SHAPE && ramp_unit < 0 && shape_unit < 0) {
                    shape_unit = in.src_unit;
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
        } else if (shape_unit >= 0) {
            plan.kind = FUSED_OSC_SHAPE;
            plan.shape = "mul(osc(k),shape)";
        } else if (have_gain) {
            plan.kind = FUSED_OSC_GAIN;
            plan.shape = "mul(osc(k),k)";
        } else
            return no("Multiply of two Oscs");
    } else if (root.op == OP_SUM) {
        // left-deep chain of Sum units over constant-f oscillators (Sum.many, Sum.js:18-29)
        std::vector<int> rev;  // oscillators from the outermost Sum inwards
        int cur = g.out_unit;
        for (;;) {
            const UnitDesc &su = g.units[(size_t)cur];
            int next = -1, oscs[2] = {-1, -1};
            for (int k = 0; k < 2; k++) {
                const InletDesc &in = su.inlets[(size_t)k];
                if (in.kind != IN_CONNECT) return no("Sum with a constant operand");
                const UnitDesc &s = g.units[(size_t)in.src_unit];
                if (s.op == OP_SUM && next < 0) next = in.src_unit;
                else if (s.op == OP_OSC || s.op == OP_MULTIPLY) oscs[k] = in.src_unit;  // (a voice: a bare oscillator, or an enveloped one — looked into below)
                else return no("unsupported Sum operand");
            }
            if (next >= 0) {
                rev.push_back(oscs[0] >= 0 ? oscs[0] : oscs[1]);
                used[(size_t)next]++;
                cur = next;
            } else {  // innermost Sum(a, b): a is voice 0, b is voice 1
                rev.push_back(oscs[1]);
                rev.push_back(oscs[0]);
                break;
            }
            if (rev.size() > 70000) return no("chain too long");
        }
        plan.kind = FUSED_SUMCHAIN;
        plan.shape = "sumchain(osc(k) x " + std::to_string(rev.size()) + ")";
        plan.sum_all_int = true;
        plan.sum_env = -1;
        for (auto it = rev.rbegin(); it != rev.rend(); ++it) {
            int ui = *it;
            if (ui < 0) return no("malformed chain");
            int env = 0;
            if (g.units[(size_t)ui].op == OP_MULTIPLY) {  // Multiply(Osc, k) or Multiply(Osc, Ramp), operands in either order
                const UnitDesc &m = g.units[(size_t)ui];
                used[(size_t)ui]++;
                int osc = -1;
                for (int k = 0; k < 2; k++) {
                    const InletDesc &in = m.inlets[(size_t)k];
                    if (in.kind == IN_CONNECT && g.units[(size_t)in.src_unit].op == OP_OSC && osc < 0) osc = in.src_unit;
                    else if (in.kind == IN_CONNECT && g.units[(size_t)in.src_unit].op == OP_RAMP && env == 0) {
                        env = 2;
                        const UnitDesc &r = g.units[(size_t)in.src_unit];
                        used[(size_t)in.src_unit]++;
                        if (plan.sum_ramp_units.empty()) {
                            plan.r_d = r.attrs[0]; plan.r_y0 = r.attrs[1]; plan.r_y1 = r.attrs[2];
                            plan.r_t0 = r.state[0]; plan.r_playing = r.state[1] != 0;
                        } else if (plan.r_d != r.attrs[0] || plan.r_y0 != r.attrs[1] || plan.r_y1 != r.attrs[2] || plan.r_t0 != r.state[0] ||
                                   plan.r_playing != (r.state[1] != 0))
                            return no("chain voices with different Ramps");
                        plan.sum_ramp_units.push_back(in.src_unit);
                    } else if (in.kind == IN_CONST && in.vals.size() == 1 && env == 0) {
                        env = 1;
                        plan.sum_gain.push_back((float)in.vals[0]);
                    } else
                        return no("unsupported voice in the chain");
                }
                if (osc < 0 || env == 0) return no("unsupported voice in the chain");
                ui = osc;
            }
            if (plan.sum_env < 0) plan.sum_env = env;
            else if (plan.sum_env != env) return no("chain voices of different kinds");
            const UnitDesc &o = g.units[(size_t)ui];
            if (o.op != OP_OSC) return no("malformed chain");
            used[(size_t)ui]++;
            const InletDesc &fin = o.inlets[0];
            if (fin.kind != IN_CONST || fin.vals.size() != 1) return no("chain oscillator needs a constant f");
            const double f = (double)(float)fin.vals[0];
            if (!std::isfinite(f)) return no("non-finite f");
            const double fr = std::fmod(f, (double)g.sample_rate), p0 = o.state[0];
            if (std::ldexp(fr, 32) != std::floor(std::ldexp(fr, 32))) return no("f finer than 2^-32");
            if (!(p0 >= 0 && p0 < g.sample_rate) || std::ldexp(p0, 32) != std::floor(std::ldexp(p0, 32)))
                return no("start phase not on the 2^-32 grid");
            if (fr != std::floor(fr) || p0 != std::floor(p0)) plan.sum_all_int = false;
            if (plan.sum_units.empty()) plan.table_id = (int)o.attrs[0];
            else if (plan.table_id != (int)o.attrs[0]) return no("mixed waveforms");
            plan.sum_units.push_back(ui);
            plan.sum_f.push_back(fr);
            plan.sum_phase0.push_back(p0);
        }
        for (size_t i = 0; i < used.size(); i++)
            if (used[i] != 1) return no("circuit has units outside the fused tree");
        plan.unit_state_first.assign(g.units.size(), 0);
        plan.unit_state_count.assign(g.units.size(), 0);
        plan.n_state_words = 0;
        for (int ui : plan.sum_units) {
            plan.unit_state_first[(size_t)ui] = plan.n_state_words++;
            plan.unit_state_count[(size_t)ui] = 1;
        }
        if (plan.sum_env == 2) {
            if (!(plan.r_d > 0) || !std::isfinite(plan.r_d) || !std::isfinite(plan.r_t0) || plan.r_t0 > plan.r_d || (plan.r_playing && !(plan.r_t0 + 1 >= 0)))
                return no("chain Ramp outside the closed-form regime");
            plan.r_rcp = 1.0 / plan.r_d;
            plan.r_fastdiv = ramp_fastdiv_ok(plan.r_t0, plan.r_d, plan.r_playing != 0) ? 1 : 0;
            for (int ui : plan.sum_ramp_units) {  // [t, playing] per Ramp unit, behind the oscillators' phases
                plan.unit_state_first[(size_t)ui] = plan.n_state_words;
                plan.unit_state_count[(size_t)ui] = 2;
                plan.n_state_words += 2;
            }
        }
        if (plan.sum_env == 1) plan.shape = "sumchain(osc(k) * k x " + std::to_string(plan.sum_units.size()) + ")";
        if (plan.sum_env == 2) plan.shape = "sumchain(osc(k) * ramp x " + std::to_string(plan.sum_units.size()) + ")";
        return true;
    } else
        return no("root is not an Osc, a Multiply or a Sum chain");
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
    if (shape_unit >= 0) {  // every inlet a single constant: the envelope is the same for all instances of the program
        const UnitDesc &sh = g.units[(size_t)shape_unit];
        for (int k = 0; k < 3; k++)
            if (sh.inlets[(size_t)k].kind != IN_CONST || sh.inlets[(size_t)k].vals.size() != 1) return no("Shape with a connected / per-instance inlet");
        plan.s_table_id = (int)sh.attrs[0];
        plan.s_left_is_shape = sh.attrs[1] != 0;
        plan.s_left = sh.attrs[2];
        plan.s_right_is_shape = sh.attrs[3] != 0;
        plan.s_right = sh.attrs[4];
        plan.s_t0 = sh.state[0];
        plan.s_playing = sh.state[1] != 0;
        plan.s_finished = sh.state[2] != 0;
        plan.s_c = 1.0 / (double)(float)sh.inlets[0].vals[0];
        plan.s_min = (float)sh.inlets[1].vals[0];
        plan.s_max = (float)sh.inlets[2].vals[0];
        // the running sum t += 1/duration in closed form (repeat_add.hpp): needs a positive finite addend from t >= 0
        if (plan.s_playing && !(plan.s_t0 >= 0 && plan.s_c > 0 && plan.s_c < 1e300)) return no("Shape outside the closed-form regime");
        if (!std::isfinite(plan.s_t0)) return no("Shape with a non-finite t");
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
    if (shape_unit >= 0) {
        plan.unit_state_first[(size_t)shape_unit] = plan.n_state_words;
        plan.unit_state_count[(size_t)shape_unit] = 3;
        plan.n_state_words += 3;
    }
    return true;
}

// Exact 32.32 fixed-point records of a Sum.many chain's oscillators, and their phases after T_end samples.
inline void build_sum_voices(const FusedPlan &plan, uint32_t sample_rate, int gb, uint64_t T_end, std::vector<SumVoice> &voices,
                             std::vector<double> &end_phase) {
    typedef unsigned __int128 u128;
    const uint64_t S = (uint64_t)sample_rate << 32;
    voices.clear();
    end_phase.clear();
    for (size_t j = 0; j < plan.sum_f.size(); j++) {
        const double f = plan.sum_f[j];
        const long long F = (long long)std::ldexp(f, 32);  // |F| < S, exact by plan_fused's check
        const uint64_t Fm = F >= 0 ? (uint64_t)F % S : (S - (uint64_t)(-F) % S) % S;
        const uint64_t P0 = (uint64_t)std::ldexp(plan.sum_phase0[j], 32);
        SumVoice v;
        v.Fm = Fm;
        v.B0 = (uint64_t)(((u128)P0 + Fm) % S);
        v.bs = (uint64_t)(((u128)Fm * (uint64_t)(gb * kChunk)) % S);
        v.step4 = (uint64_t)(((u128)Fm * 4) % S);
        v.step256 = (uint64_t)(((u128)Fm * kChunk) % S);
        v.gain = plan.sum_env == 1 ? plan.sum_gain[j] : 1.f;
        v.pad = 0;
        voices.push_back(v);
        end_phase.push_back(std::ldexp((double)(uint64_t)(((u128)P0 + (u128)Fm * T_end) % S), -32));
    }
    if (plan.sum_env == 2)  // what every Ramp holds after T_end ticks (Ramp.js:27-36): behind the phases, [t, playing] per unit
        for (size_t j = 0; j < plan.sum_ramp_units.size(); j++) {
            const double since = (double)T_end;
            end_phase.push_back(plan.r_playing ? std::fmin(plan.r_t0 + since, plan.r_d) : plan.r_t0);
            end_phase.push_back((plan.r_playing && plan.r_t0 + since <= plan.r_d) ? 1.0 : 0.0);
        }
}

// Regime of a per-instance (parameter) delay, found by looking at the parameter column when a render starts (dusp_abi.hip
// classify_param_delays) and kept in the operand's spare word: every instance's delay of at least a chunk (and a chunk short of the
// ring's length), every instance's below a chunk, or anything else (mixed, negative, NaN, within a sample of the chunk size).
enum : int { DELAY_REGIME_UNKNOWN = 0, DELAY_REGIME_LONG = 1, DELAY_REGIME_SHORT = 2, DELAY_REGIME_OTHER = 4 };
// one delay value's regime, as the compiled kernels' units will treat it (jit_prelude.hpp JitDelayK / JitDelayShort)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int delay_value_regime(float delay, int64_t ring_len, bool mono) {
    double d = (double)delay;
    if (!(d >= 0.0) || !(d < 4.0e9)) return DELAY_REGIME_OTHER;  // negative, NaN, absurd
    const double len = (double)ring_len;
    if (d >= (mono ? 0.0 : 1.0) && d < (double)(kChunk - 1) && ring_len >= 2 * kChunk) return DELAY_REGIME_SHORT;
    if (d >= len) d = fmod(d, len);
    const double fl = floor(d);
    if (fl >= (double)kChunk && fl + (double)kChunk <= len) return DELAY_REGIME_LONG;
    return DELAY_REGIME_OTHER;
}

// Delay with a constant delay of at least a chunk, and a chunk short of the ring's length: the wave engine's write-once
// ring protocol (same predicate as delay_is_write_once in wave_engine.hip); other Delays go through ordered slot operations.
// A per-instance delay qualifies once the renderer has found every instance's in that regime.
inline bool delay_write_once(const DevOp &op) {
    if (op.in[1].kind == SRC_PARAM) return op.in[1].pad == DELAY_REGIME_LONG;
    if (op.in[1].kind != SRC_CONST) return false;
    const double len = (double)op.ring_len;
    double dconst = (double)op.in[1].cval;
    if (dconst >= len) dconst = std::fmod(dconst, len);
    return std::floor(dconst) >= kChunk && std::floor(dconst) + kChunk <= len;
}

// Can the wave engine (one wavefront per instance, chunk buffers in LDS) run this program?
// will_continue: the program is resumable, i.e. later launches pick up rings and parked chunk buffers in the reference's layout
// settled_only: plan the settled op list of a program whose first chunks run op lists of their own (Program::warm_ops) — those chunks
// are rendered by the chunk engine and the rest handed to a compiled kernel (dusp_abi.hip)
inline bool plan_wave(const Program &P, WavePlan &plan, bool will_continue = false, bool settled_only = false) {
    const Graph &g = P.g;
    auto no = [&](const std::string &why) {
        plan.why = why;
        plan.ok = false;
        return false;
    };
    if (!P.warm_ops.empty() && !settled_only) return no("channel counts grow during the first chunks");
    // CircleBuffer nodes: with a lane-constant offset a node touches 256 consecutive slots per chunk (lane-parallel); a
    // signal-rate offset can make two samples of one chunk meet in one slot, and a ring shorter than a chunk wraps onto
    // itself: those go through the ordered slot operations (below: ring_events)
    for (const DevOp &op : P.ops)
        if ((op.op == OP_CB_READER || op.op == OP_CB_WRITER) && (op.ring_len < 1 || op.ring_len >= (1ll << 31))) return no("CircleBuffer ring out of range");
    if (g.sample_rate > 131072) return no("sample rate above 2^17");
    plan.has_filter = plan.ring_events = false;
    plan.scratch_bytes = 0;
    plan.ext_units = 0;
    for (const DevOp &op : P.ops) {
        const int o = op.op;
        const int wants = (o == OP_OSC || o == OP_RAMP || o == OP_MULTIPLY || o == OP_SUM || o == OP_REPEATER || (o >= OP_MAP_FIRST && o <= OP_MAP_LAST)) ? 0
                          : (o == OP_FILTER && op.in[1].kind == SRC_BUF) ? 3  // a modulated cutoff: per-sample coefficients live in the EXT variants
                          : (o == OP_FILTER || o == OP_DELAY) ? 1 : 2;
        plan.ext_units |= wants;
        plan.ring_events = plan.ring_events || (op.op == OP_DELAY && !delay_write_once(op)) || op.op == OP_MONO_DELAY || op.op == OP_READBACK_DELAY ||
                           ((op.op == OP_CB_READER || op.op == OP_CB_WRITER) && (op.in[0].kind == SRC_BUF || op.ring_len < kChunk));
        plan.has_filter = plan.has_filter || op.op == OP_FILTER;
        // per-wave scratch: what the unit's serial stage (or its fallback) lays out there
        int scratch = 0;
        if (o == OP_FILTER && op.in[1].kind == SRC_BUF) scratch = 3 * 256 * 8;             // P, b1, b2 per sample (f64)
        else if (o == OP_SHAPE) {                                                          // the running sum's 256 addends — not for a
            const double c = op.in[0].kind == SRC_CONST ? 1.0 / (double)op.in[0].cval : 0.0;  // constant duration that repeat_add covers
            scratch = (c > 0 && c < 1e300 && P.init_state[(size_t)op.state_slot] >= 0) ? 0 : 256 * 8;
        } else if (o == OP_TIMER) scratch = (P.init_state[(size_t)op.state_slot] >= 0 && op.d[0] > 0 && op.d[0] < 1e300) ? 0 : 256 * 4;
        else if (o == OP_AHD || o == OP_SAMPLE_RATE_REDUX || o == OP_FIXED_DELAY || o == OP_COMB_FILTER || o == OP_ALL_PASS) scratch = 4 * 256 * 4;
        else if (o == OP_MULTI_OSC) scratch = 256 * 8;
        plan.scratch_bytes = std::max(plan.scratch_bytes, scratch);
    }
    if (plan.ring_events) plan.scratch_bytes = std::max(plan.scratch_bytes, 1024 * 4);  // slot-ownership table of the ordered ring operations
    // state blocks: only for the ops that keep something in LDS between chunks
    plan.op_state.assign(P.ops.size(), -1);
    plan.n_state_ops = 0;
    for (size_t k = 0; k < P.ops.size(); k++) {
        const int o = P.ops[k].op;
        const bool stateless = o == OP_RAMP || o == OP_MULTIPLY || o == OP_SUM || o == OP_REPEATER || o == OP_INPUT ||
                               (o >= OP_MAP_FIRST && o <= OP_MAP_LAST) || (o >= OP_WIDE_FIRST && o <= OP_WIDE_LAST);
        if (!stateless) plan.op_state[k] = plan.n_state_ops++;
    }
    for (const DevOp &op : P.ops)  // a Ramp that a Retriggerer restarts keeps (t at the restart, sample of the restart, playing) in LDS
        if (op.op == OP_RETRIGGER && op.pad >= 0 && (size_t)op.pad < P.ops.size() && P.ops[(size_t)op.pad].op == OP_RAMP &&
            plan.op_state[(size_t)op.pad] < 0)
            plan.op_state[(size_t)op.pad] = plan.n_state_ops++;
    // chunk buffers by liveness.  Only where no chunk survives the chunk boundary: a feedback (or late) edge reads its
    // producer's PREVIOUS chunk, and continued programs with rings park all buffers between launches — those keep one
    // buffer per outlet channel.  An op's output slot is taken before its inputs' slots are released, so an op never
    // writes over what it is still reading.
    plan.buf_slot.assign((size_t)std::max(1, P.n_bufs), 0);
    plan.order.resize(P.ops.size());
    for (size_t k = 0; k < P.ops.size(); k++) plan.order[k] = (int)k;
    // (a Delay's ring is private to it; CircleBuffers are shared between nodes whose order matters)
    bool has_retrigger = false;  // (acts on another op's state: it has to stay in front of its target, so the reference's order is kept)
    for (const DevOp &op : P.ops) has_retrigger = has_retrigger || op.op == OP_RETRIGGER;
    if (P.feed_forward && g.rings.empty() && (P.ring_samples == 0 || !will_continue) && !has_retrigger) {
        // Such a graph is pure dataflow (private state, no shared rings), so any order that respects the edges computes
        // the same thing.  The reference's order is level by level — every oscillator of a 200-voice mix before the first
        // Multiply — which keeps hundreds of chunks alive; depth-first from the output (inputs first, then the op) keeps a
        // mix-down chain at three or four.
        std::vector<int> producer((size_t)std::max(1, P.n_bufs), -1);
        for (size_t k = 0; k < P.ops.size(); k++)
            if (P.ops[k].out_buf >= 0) producer[(size_t)P.ops[k].out_buf] = (int)k;
        std::vector<char> done(P.ops.size(), 0);
        std::vector<int> order;
        std::vector<std::pair<int, int>> stack;  // (op, next operand to look at)
        auto visit = [&](int root) {
            if (done[(size_t)root]) return;
            stack.push_back({root, 0});
            done[(size_t)root] = 1;
            while (!stack.empty()) {
                auto &top = stack.back();
                const DevOp &op = P.ops[(size_t)top.first];
                bool descended = false;
                while (top.second < kMaxIn) {
                    const DevOperand &in = op.in[top.second++];
                    if (in.kind != SRC_BUF || in.idx < 0 || in.idx >= P.n_bufs) continue;
                    const int src = producer[(size_t)in.idx];
                    if (src < 0 || done[(size_t)src]) continue;
                    done[(size_t)src] = 1;
                    stack.push_back({src, 0});
                    descended = true;
                    break;
                }
                if (!descended && stack.back().second >= kMaxIn) {
                    order.push_back(stack.back().first);
                    stack.pop_back();
                }
            }
        };
        for (int b : P.out_bufs)
            if (producer[(size_t)b] >= 0) visit(producer[(size_t)b]);
        for (size_t k = 0; k < P.ops.size(); k++) visit((int)k);  // units nothing listens to still tick (their state is read back)
        plan.order = order;

        std::vector<int> last_use((size_t)std::max(1, P.n_bufs), -1);
        for (size_t at = 0; at < order.size(); at++)
            for (int j = 0; j < kMaxIn; j++) {
                const DevOperand &in = P.ops[(size_t)order[at]].in[j];
                if (in.kind == SRC_BUF && in.idx >= 0 && in.idx < P.n_bufs) last_use[(size_t)in.idx] = (int)at;
            }
        for (int b : P.out_bufs) last_use[(size_t)b] = (int)order.size();  // copied out at the end of the chunk
        std::vector<int> free_slots;
        int n_slots = 0;
        std::vector<char> have(plan.buf_slot.size(), 0);
        for (size_t at = 0; at < order.size(); at++) {
            const DevOp &op = P.ops[(size_t)order[at]];
            const int b = op.out_buf;
            // The lean units read their operands into registers before they store anything, so their output may take over
            // the slot of an operand that dies here; every other unit gets its output slot first.
            const bool registers_first = op.op == OP_OSC || op.op == OP_RAMP || op.op == OP_MULTIPLY || op.op == OP_SUM || op.op == OP_REPEATER ||
                                         (op.op >= OP_MAP_FIRST && op.op <= OP_MAP_LAST);
            auto release_dead_inputs = [&]() {
                for (int j = 0; j < kMaxIn; j++) {
                    const DevOperand &in = op.in[j];
                    if (in.kind == SRC_BUF && in.idx >= 0 && in.idx < P.n_bufs && last_use[(size_t)in.idx] == (int)at && have[(size_t)in.idx] == 1) {
                        free_slots.push_back(plan.buf_slot[(size_t)in.idx]);
                        have[(size_t)in.idx] = 2;  // released
                    }
                }
            };
            if (registers_first) release_dead_inputs();
            if (b >= 0 && !have[(size_t)b]) {
                if (free_slots.empty()) plan.buf_slot[(size_t)b] = n_slots++;
                else { plan.buf_slot[(size_t)b] = free_slots.back(); free_slots.pop_back(); }
                have[(size_t)b] = 1;
            }
            release_dead_inputs();
            if (b >= 0 && last_use[(size_t)b] < 0 && have[(size_t)b] == 1) {  // nobody reads it
                free_slots.push_back(plan.buf_slot[(size_t)b]);
                have[(size_t)b] = 2;
            }
        }
        plan.n_slots = std::max(1, n_slots);
    } else {
        for (size_t b = 0; b < plan.buf_slot.size(); b++) plan.buf_slot[b] = (int)b;
        plan.n_slots = std::max(1, P.n_bufs);
    }
    // (the same sum launch_wave_engine makes for one wavefront: buffers + state + scratch + parameter column + Filter tiles)
    if (wave_lds_bytes((size_t)plan.n_slots, (size_t)plan.n_state_ops, (size_t)plan.scratch_bytes) + wave_param_bytes((size_t)g.n_params) +
            (plan.has_filter ? 258 * 8 + 260 * 4 : 0) > 160 * 1024)
        return no("too many chunk buffers for LDS");
    for (const DevOp &op : P.ops) {
        switch (op.op) {
        case OP_OSC: {
            const double p0 = P.init_state[(size_t)op.state_slot];
            if (!(p0 >= 0 && p0 < g.sample_rate) || std::ldexp(p0, 36) != std::floor(std::ldexp(p0, 36)))
                return no("Osc start phase outside [0, sampleRate) or finer than 2^-36");
            if (plan.lds_table_id < 0) plan.lds_table_id = op.attr;
            break;
        }
        case OP_RAMP: {
            const double d = op.d[0], t0 = P.init_state[(size_t)op.state_slot];
            const bool playing = P.init_state[(size_t)op.state_slot + 1] != 0;
            if (!(d > 0) || !std::isfinite(d) || !std::isfinite(t0) || t0 > d || (playing && !(t0 + 1 >= 0)))
                return no("Ramp outside the closed-form regime");
            break;
        }
        case OP_DELAY: case OP_MONO_DELAY: case OP_READBACK_DELAY:  // write-once ring protocol or ordered slot operations (wave_engine.hip)
            if (op.ring_len < 1 || op.ring_len >= (1ll << 31)) return no("delay ring out of range");
            break;
        case OP_FILTER:
        case OP_MULTIPLY:
        case OP_SUM:
        case OP_REPEATER: break;
        case OP_SHAPE: case OP_AHD: case OP_TIMER: case OP_SAMPLE_RATE_REDUX: break;  // serial stage on one lane, rest lane-parallel
        case OP_CB_READER: case OP_CB_WRITER: break;                                     // checked above
        case OP_MULTI_OSC: break;                                                        // phases on the serial lane, lookups lane-parallel
        case OP_INPUT: break;                                                            // a float4 per lane from the host's stream
        case OP_RETRIGGER: break;                                                        // 256 additions on one lane; rewrites its target's state block
        case OP_FIXED_DELAY: case OP_COMB_FILTER: case OP_ALL_PASS:                     // lane-parallel in rounds of the ring length
            if (op.ring_len < 1 || op.ring_len >= (1ll << 31)) return no("comb ring out of range");
            break;
        default:
            if ((op.op >= OP_MAP_FIRST && op.op <= OP_MAP_LAST) || (op.op >= OP_WIDE_FIRST && op.op <= OP_WIDE_LAST)) break;  // stateless maps
            return no("unit the wave engine does not run");
        }
    }
    // Ramp: t / duration with a 2-FMA reciprocal refinement where it provably equals the division on every t this Ramp
    // can take (exhaustive host check, as for the fused kernels; the verdict is remembered across continuations, whose
    // t sequences are suffixes of the one already checked)
    plan.ramp_fastdiv.assign(P.ops.size(), 0);
    for (size_t k = 0; k < P.ops.size(); k++) {
        const DevOp &op = P.ops[k];
        if (op.op != OP_RAMP) continue;
        const double d = op.d[0], t0 = P.init_state[(size_t)op.state_slot];
        const bool playing = P.init_state[(size_t)op.state_slot + 1] != 0;
        bool known = false, ok = false;
        for (const auto &c : plan.ramp_checked)
            if (c.d == d && t0 >= c.t0 && t0 - c.t0 == std::floor(t0 - c.t0)) { known = true; ok = c.ok; break; }
        if (!known) {
            ok = d < 16777216.0 && ramp_fastdiv_ok(t0, d, true);  // (as if playing: covers an idle Ramp that is triggered later)
            plan.ramp_checked.push_back({d, t0, ok});
        }
        (void)playing;
        plan.ramp_fastdiv[k] = (ok && plan.op_state[k] < 0) ? 1 : 0;  // (a restarted Ramp runs through t sequences nobody checked)
    }
    // Time-split rendering: without Filters / Delays / feedback the only state that crosses a chunk boundary is each
    // oscillator's phase, and that is a modular SUM of its increments — segments can be rendered independently once
    // every segment's phase total is known.  An oscillator whose increments come from other oscillators (FM) needs
    // theirs resolved first: level = depth of oscillators stacked in the f input.
    plan.splittable = P.feed_forward && P.ring_samples == 0;  // (Filters: looked at behind the loop)
    plan.osc_level.assign(P.ops.size(), -1);
    std::vector<int> buf_depth((size_t)std::max(1, P.n_bufs), 0);
    for (size_t k = 0; k < P.ops.size(); k++) {
        const DevOp &op = P.ops[k];
        // Timer and a Shape with a constant duration run on sums that repeat_add() evaluates anywhere in closed form
        const bool closed_form_sum =
            (op.op == OP_TIMER && P.init_state[(size_t)op.state_slot] >= 0 && op.d[0] > 0 && op.d[0] < 1e300) ||
            (op.op == OP_SHAPE && (P.init_state[(size_t)op.state_slot + 1] == 0 ||  // idle: t does not move
                                   (op.in[0].kind == SRC_CONST && P.init_state[(size_t)op.state_slot] >= 0 && 1.0 / (double)op.in[0].cval > 0 &&
                                    1.0 / (double)op.in[0].cval < 1e300)));
        if (!closed_form_sum &&
            (op.op == OP_DELAY || op.op == OP_SHAPE || op.op == OP_AHD || op.op == OP_TIMER || op.op == OP_SAMPLE_RATE_REDUX ||
             op.op == OP_FIXED_DELAY || op.op == OP_COMB_FILTER || op.op == OP_ALL_PASS || op.op == OP_MULTI_OSC ||
             op.op == OP_MONO_DELAY || op.op == OP_READBACK_DELAY))
            plan.splittable = false;
        if (op.op == OP_RETRIGGER) plan.splittable = false;  // its accumulator wraps by subtraction: no closed form, and its target's state follows it  // state that is neither a modular sum nor a closed-form one
        int dep = 0;
        for (int j = 0; j < kMaxIn; j++)
            if (op.in[j].kind == SRC_BUF && op.in[j].idx >= 0 && op.in[j].idx < P.n_bufs) dep = std::max(dep, buf_depth[(size_t)op.in[j].idx]);
        if (op.op == OP_OSC) {
            plan.osc_level[k] = dep;
            plan.max_osc_level = std::max(plan.max_osc_level, dep);
            dep += 1;
        }
        if (op.out_buf >= 0) buf_depth[(size_t)op.out_buf] = dep;
    }
    plan.splittable_but_for_filters = plan.splittable && plan.has_filter;
    plan.splittable = plan.splittable && !plan.has_filter;
    plan.ok = true;
    return true;
}

}  // namespace dusp
