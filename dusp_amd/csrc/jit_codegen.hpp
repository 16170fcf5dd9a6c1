// jit_codegen.hpp — the circuit compiler's front half: a compiled Program (units in the reference's process order,
// channel-expanded) -> the TEXT of one HIP kernel that renders exactly that circuit.
//
// What the reference does per chunk — Circuit.tick walks `circuit.units` and calls every unit's `_tick`
// (src/Circuit.js:19-41, src/Unit.js:111-119) — becomes the body of one loop, one statement block per unit, in that order.
// Operands are named, not fetched: an inlet constant is `kN` (a scalar read once from the constants array), a per-instance
// parameter `pN`, a connection `vB` = the producing unit's four samples in this lane's registers — or `wB`, the same
// registers as the PREVIOUS iteration left them, when the producer ticks later than the consumer (a feedback edge or one
// of the edges the reference's process order gets "late": the reference reads the producer's previous chunk there).
//
// The text depends on the circuit's STRUCTURE only (unit kinds, wiring, table ids, which operands are constants): constants
// travel in the fk / dk arrays, state in init_state, so every circuit of the same shape — and every segment of an
// event-segmented render — reuses one compiled kernel (jit_engine.hip caches by the text's hash).
#pragma once
#include <algorithm>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <string>
#include <vector>

#include "device_types.hpp"
#include "filter_lamda.hpp"
#include "fused_plan.hpp"
#include "program.hpp"

namespace dusp {

struct JitSource {
    bool ok = false;
    std::string why;            // when not ok: what keeps this program on the interpreter
    std::string text;           // kernels: dusp_jit_render, dusp_jit_pass<L> for every L in pass_levels (per_wave == 1 only)
    std::vector<float> fk;      // values of the constants the text names k0, k1, ...
    std::vector<double> dk;     // ... and d0, d1, ...
    std::vector<int> pass_levels;                 // time-split rendering: FM levels that have scanned oscillators, ascending
    struct Scan { int op_pos, state_slot, level; };  // scanned oscillators: position in execution order, state slot, FM level
    std::vector<Scan> scans;
    bool has_filter = false;
    int n_filters = 0;
    bool voice_loop = false;    // the circuit's voices in a loop (VoicePlan): unsplit renders, one instance per wavefront
};

struct JitOptions {
    int waves = 16;          // wavefronts per workgroup (compile-time: launch bounds, the Filter stage's tile geometry)
    int per_wave = 1;        // R: circuit instances per wavefront (unsplit renders only)
    int lds_table = -1;      // table id whose half image sits in LDS, or -1
    size_t table_bytes = 0;  // size of that image (half_table_lds_bytes of the sample rate)
    int filter_sub = 256;    // samples per sub-block of the Filter stage (jit_filter_sub)
    int filter_stages = 0;   // Filter stages of the circuit (jit_filter_stages): each keeps its rows' y1 / y2 behind the tile
    bool filter_mod = false; // one of them has a connected cutoff (jit_filter_mod): the tile's rows hold P, b1 and b2 of a sub-block
    bool rotate_mod = false; // ... also for stages with a connected cutoff (24 more registers per instance and stage)
    bool rotate = true;      // ... and what feeds a Filter runs a chunk ahead there, where nothing else reads it (Emitter::plan_rotate)
    bool overlap = true;     // Filter circuits: units that neither feed a Filter nor hang on one run beside the recurrences (Emitter::plan_overlap)
    bool line_whole_only = false; // ... those with a whole number of samples of delay only (DUSP_DELAY_LINE=2)
    size_t line_floats = 0;    // per-wave LDS rows of the Delays kept as lines of input samples (jit_delay_lines: part of scratch_floats, behind the shared scratch); 0: none
    bool filter_scan = false;  // every Filter of the circuit as a scan over the chunk (jit_filter_scan_ok, JitFilterScan): no Filter stage, no tile
    bool warm = false;         // the kernel of a render cut into segments that warm up (JitArgs::warm): the Filter stages record what they hold at segment boundaries
    int filter_block = 8;    // P values per register set of the Filter stage's recurrence loop: 8, or 4 for a kernel short of registers
    int table_form[kNumTables] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // TABLE_FORM_* of every table (device_util.hpp), as the context found them at upload
    int table_delta[kNumTables] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // the lerp's delta form (device_util.hpp lerp_delta): 0 not for this table, 1 differences of neighbours in f64, 2 in f32
    int table_bound[kNumTables] = {1000, 1000, 1000, 1000, 1000, 1000, 1000, 1000, 1000};  // every |entry| <= 2^bound (finite tables; 1000: not known to be)
    size_t scratch_floats = 0;  // per-wave LDS scratch of the units with a sequential stage (jit_scratch_floats)
    bool voice_loop = false; // a sum of isomorphic voices above jit_loop_voices_from() units gets its voices in a loop (VoicePlan): unsplit renders only
    bool persistent = false; // a continued program with delay lines / feedback: outlets parked between launches, rings in the reference's state
    bool filter_fma = false; // EXPERIMENT (DUSP_FILTER_FMA=1): the Filter stage's recurrence with one fused multiply-add on its dependent chain
    bool nt_stores = false;  // the copy-out with non-temporal stores (circuits with delay rings in memory: DUSP_JIT_NT)
    bool profile = false;    // diagnostic build (DUSP_JIT_PROFILE=1): wave 0 of every workgroup stamps the cycle counter around the chunk loop and
                             // inside the Filter stage's serial part and leaves the sums in JitArgs::debug
};

// Where an oscillator's table comes from in a generated kernel (jit_prelude.hpp jit_pair): 1 the LDS half image, 2 a closed form
// of the index, 3 8bit derived from the sine image, 0 a gather from L2.
inline int jit_table_source(const JitOptions &opt, int table_id) {
    if (opt.lds_table == table_id) return 1;
    const int form = opt.table_form[table_id];
    if (form == 1 || form == 2 || form == 3) return 2;  // TABLE_FORM_SAW / SQUARE / TRIANGLE
    if (form == 4 && opt.lds_table == 0) return 3;       // TABLE_FORM_8BIT next to the sine image
    return 0;
}

// LDS of one workgroup of the generated kernel: the table image, then the Filter stage's tile, then the sequential-stage units'
// per-wave scratch.  Declared statically in the
// kernel text (a static declaration may take all 160 KiB; dynamic LDS would need a function attribute that module kernels
// do not have).
// Filter stage: one row of (sub-block + 2) doubles per instance of the workgroup; the sub-block (256, 128 or 64 samples) is the
// largest that fits the LDS left over.
// Behind the rows: one word that says "given back", then y1 / y2 of every row of every Filter stage of the circuit (the stages share
// the tile, one after the other, but each has its own recurrence memory).
// (mod: a stage with a connected cutoff shares the tile — rows of three arrays, P / b1 / b2 per sample: JitFilterKM)
inline size_t jit_filter_rows_bytes(int rows, int sub, bool mod) { return (size_t)rows * (size_t)((mod ? 3 * sub : sub) + 2) * 8; }
inline size_t jit_filter_tile_bytes(int rows, int sub, int stages, bool mod) { return jit_filter_rows_bytes(rows, sub, mod) + 16 + (size_t)stages * (size_t)rows * 16; }
inline int jit_filter_stages(const Program &P) {
    int n = 0;
    for (const DevOp &op : P.ops) n += op.op == OP_FILTER;
    return n;
}
// Can every Filter of the circuit run as a scan (jit_prelude.hpp JitFilterScan)?  The scan carries unrounded pairs where the reference
// rounds y to f32 every step: an error of at most 2^-24 |y| a step, which reaches later samples through the Filter's all-pole part
// 1 / (1 + b1 z^-1 + b2 z^-2) (impulse response h); the two results' own roundings to f32 add an ulp.  A Filter by itself therefore stays within
//     eps_F = 2^-24 (sum|h| + 2)
// of its own output's scale of the reference's (tests/native/filter_scan_bound_check.cpp).  Two conditions, both from the coefficients the
// render will use:
//   (1) every Filter: a cutoff that is a constant of the circuit and eps_F <= kFilterScanUnit = 2^-19 (1.9e-6; sum|h| <= 30: 48 kHz: cutoffs
//       between about 1.5 and 22.5 kHz, either kind);
//   (2) what the circuit makes of those deviations: every unit that hangs on a Filter's output must pass a deviation on LINEARLY, and with
//       the worst-case gain of each — a Filter's own sum|g| (g: its whole impulse response), |k| of a product with a constant k, the bound
//       of an untainted signal it is multiplied with, 1 for delay lines, the sum at a Sum — the deviation bound x at every buffer solves
//       x = G x + eps, AROUND FEEDBACK LOOPS TOO: a loop of gain g amplifies what is injected into it by up to 1 / (1 - g), and a loop whose
//       gain bound reaches 1 has no bound at all.  Taken only when the iteration converges and every outlet stays within
//       kFilterScanBound = 2.5e-6, a quarter of this path's 1e-5, of the LARGEST Filter output's scale.  (configs[3]: eps 1.22e-6, loop gain
//       0.5 x 1 x 1: 2.44e-6.  Round 3's gate had (1) and the structural half of (2) only: it bounded one pass and let the deviation go
//       round loops of any gain.)
// Where a Filter reaches an oscillator's frequency, a delay TIME, a division, a power, a clip or an envelope — units that integrate or bend
// their input —, a product of two deviating signals or one with a per-instance factor, the circuit keeps the Filter stage and its bits.
// mode 2 (DUSP_FILTER_SCAN=2; measurements and tests of the scan itself): (1) and the structure of (2), whatever the gains.
constexpr double kFilterScanUnit = 1.9073486328125e-6;  // 2^-19
constexpr int kFilterColumnKnown = 1;  // DevOperand::pad of a Filter's per-instance cutoff: the renderer has looked at the column, DevOp::d[0] / d[1] hold its smallest / largest value
constexpr double kFilterScanBound = 2.5e-6;
inline bool jit_filter_scan_ok(const Program &P, const int *table_bound = nullptr, int mode = 1, double *bound_out = nullptr) {
    if (bound_out) *bound_out = 0.0;
    const size_t n_bufs = (size_t)std::max(1, P.n_bufs);
    std::vector<double> eps(P.ops.size(), 0.0), gain(P.ops.size(), 0.0);  // per Filter: what it injects, its own worst-case gain
    bool any = false;
    for (size_t at = 0; at < P.ops.size(); at++) {
        const DevOp &op = P.ops[at];
        if (op.op != OP_FILTER) continue;
        any = true;
        if ((size_t)op.state_slot + 11 > P.init_state.size()) return false;
        const double *is = P.init_state.data() + op.state_slot;
        // the cutoffs to answer for: the circuit's constant, or — a per-instance parameter whose column the renderer has looked at
        // (kFilterColumnKnown: every instance's cutoff finite, d[0] <= f <= d[1] inside (0, Nyquist)) — a spread over that range:
        // sum|h| = 1 / (1 - |p|)^2 with the double pole p = (lamda - 1) / (lamda + 1) is largest at an end of the range, the whole
        // Filter's sum|g| is sampled along it
        double cutoffs[9];
        int n_cut = 0;
        if (op.in[1].kind == SRC_CONST) cutoffs[n_cut++] = (double)op.in[1].cval;
        else if (op.in[1].kind == SRC_PARAM && op.in[1].pad == kFilterColumnKnown && op.d[0] > 0.0 && op.d[0] <= op.d[1] && op.d[1] < 0.5 * (double)P.g.sample_rate)
            for (int i = 0; i < 9; i++) cutoffs[n_cut++] = op.d[0] + (op.d[1] - op.d[0]) * (double)i / 8.0;
        else return false;
        for (int c = 0; c < n_cut; c++) {
            const double f = cutoffs[c];
            double k[5];
            if (op.in[1].kind == SRC_CONST && is[0] != 0.0 && f == is[1])
                for (int i = 0; i < 5; i++) k[i] = is[2 + i];
            else butterworth_coefficients(op.attr, f, (double)P.g.sample_rate, k);
            for (double v : k)
                if (!std::isfinite(v)) return false;
            // h: the all-pole part's impulse response; g: the whole Filter's
            double h1 = 1.0, h2 = 0.0, sum_h = 1.0, g1 = k[0], g2 = 0.0, sum_g = std::fabs(k[0]);
            int quiet = 0;
            for (int t = 1; t < 100000 && quiet < 8; t++) {
                const double h = -k[3] * h1 - k[4] * h2;
                const double g = (t == 1 ? k[1] : t == 2 ? k[2] : 0.0) - k[3] * g1 - k[4] * g2;
                h2 = h1, h1 = h, g2 = g1, g1 = g;
                sum_h += std::fabs(h);
                sum_g += std::fabs(g);
                if (!(sum_h <= 30.0)) return false;
                quiet = std::fabs(h) < 1e-13 && t > 2 ? quiet + 1 : 0;
            }
            if (quiet < 8) return false;
            eps[at] = std::max(eps[at], std::ldexp(sum_h + 2.0, -24));
            gain[at] = std::max(gain[at], sum_g);
        }
        if (!(eps[at] <= kFilterScanUnit)) return false;
    }
    if (!any) return false;
    // taint: what a Filter's deviation reaches (to a fixed point: feedback edges run against the order), and that nothing on the way bends it
    std::vector<char> tainted(n_bufs, 0);
    std::vector<int> producer(n_bufs, -1);
    for (size_t at = 0; at < P.ops.size(); at++)
        if (P.ops[at].out_buf >= 0 && (size_t)P.ops[at].out_buf < n_bufs) producer[(size_t)P.ops[at].out_buf] = (int)at;
    auto deps = [&](const DevOp &op, bool (&dep)[kMaxIn]) {
        bool any_dep = false;
        for (int j = 0; j < kMaxIn; j++) dep[j] = false;
        for (int j = 0; j < kMaxIn && j < std::max(op.n_in, 2); j++)
            if (op.in[j].kind == SRC_BUF && op.in[j].idx >= 0 && op.in[j].idx < P.n_bufs && tainted[(size_t)op.in[j].idx]) dep[j] = any_dep = true;
        return any_dep;
    };
    for (int sweep = 0; sweep < 2 + (int)P.ops.size(); sweep++) {
        bool grew = false;
        for (const DevOp &op : P.ops) {
            bool dep[kMaxIn];
            const bool any_dep = deps(op, dep);
            if (op.op != OP_FILTER && !any_dep) continue;
            switch (op.op) {
            case OP_FILTER: case OP_SUM: case OP_SUBTRACT: case OP_REPEATER: case OP_POLARITY_INVERT: case OP_FIXED_MULTIPLY: case OP_FIXED_DELAY: break;
            case OP_MULTIPLY:
                if (dep[0] && dep[1]) return false;  // (a product of two deviating signals scales one deviation by the other signal)
                break;
            case OP_DELAY: case OP_MONO_DELAY:
                if (dep[1]) return false;  // (the delay time)
                break;
            default: return false;
            }
            if (op.out_buf >= 0 && op.out_buf < P.n_bufs && !tainted[(size_t)op.out_buf]) tainted[(size_t)op.out_buf] = 1, grew = true;
        }
        if (!grew) break;
    }
    if (mode == 2) return true;
    // a bound on the magnitude of an UNTAINTED signal (what a deviation is multiplied with); infinity: not known
    const double inf = std::numeric_limits<double>::infinity();
    std::vector<char> visiting(P.ops.size(), 0);
    std::function<double(const DevOperand &)> magnitude = [&](const DevOperand &o) -> double {
        if (o.kind == SRC_CONST) return std::fabs((double)o.cval);
        if (o.kind != SRC_BUF || o.idx < 0 || (size_t)o.idx >= n_bufs || producer[(size_t)o.idx] < 0) return inf;  // (a per-instance parameter: not in the text)
        const int k = producer[(size_t)o.idx];
        if (visiting[(size_t)k]) return inf;
        const DevOp &op = P.ops[(size_t)k];
        visiting[(size_t)k] = 1;
        double m = inf;
        switch (op.op) {
        case OP_OSC:
            if (table_bound && op.attr >= 0 && op.attr < kNumTables && table_bound[op.attr] < 64) m = std::ldexp(1.0, table_bound[op.attr]);
            break;
        case OP_RAMP: m = std::max(std::fabs(op.d[1]), std::fabs(op.d[2])); break;
        case OP_REPEATER: case OP_POLARITY_INVERT: case OP_FIXED_DELAY: m = magnitude(op.in[0]); break;
        case OP_MULTIPLY: m = magnitude(op.in[0]) * magnitude(op.in[1]); break;
        case OP_SUM: case OP_SUBTRACT: m = magnitude(op.in[0]) + magnitude(op.in[1]); break;
        default: break;
        }
        visiting[(size_t)k] = 0;
        return m == m ? m : inf;
    };
    // x = G x + eps by sweeps from zero: the iterates only grow, so one above the bound settles it; a loop whose gain bound reaches 1 never converges
    std::vector<double> x(n_bufs, 0.0);
    for (int sweep = 0; sweep < 4096; sweep++) {
        double moved = 0.0;
        for (size_t at = 0; at < P.ops.size(); at++) {
            const DevOp &op = P.ops[at];
            if (op.out_buf < 0 || op.out_buf >= P.n_bufs || !tainted[(size_t)op.out_buf]) continue;
            auto xin = [&](int j) { return op.in[j].kind == SRC_BUF && op.in[j].idx >= 0 && op.in[j].idx < P.n_bufs && tainted[(size_t)op.in[j].idx] ? x[(size_t)op.in[j].idx] : 0.0; };
            auto is_dep = [&](int j) { return op.in[j].kind == SRC_BUF && op.in[j].idx >= 0 && op.in[j].idx < P.n_bufs && tainted[(size_t)op.in[j].idx] != 0; };
            double v = 0.0;
            switch (op.op) {
            case OP_FILTER: v = gain[at] * xin(0) + eps[at]; break;
            case OP_SUM: case OP_SUBTRACT: v = xin(0) + xin(1); break;
            case OP_MULTIPLY: v = is_dep(0) ? xin(0) * magnitude(op.in[1]) : xin(1) * magnitude(op.in[0]); break;
            case OP_FIXED_MULTIPLY: v = xin(0) * std::fabs(op.d[0]); break;
            default: v = xin(0); break;  // Repeater, sign flip, delay lines: the deviation as it is
            }
            if (!(v < 1.0)) return false;  // (infinite, NaN, or beyond any use)
            moved = std::max(moved, v - x[(size_t)op.out_buf]);
            x[(size_t)op.out_buf] = std::max(x[(size_t)op.out_buf], v);
        }
        double worst = 0.0;
        for (int ob : P.out_bufs)
            if (ob >= 0 && ob < P.n_bufs && tainted[(size_t)ob]) worst = std::max(worst, x[(size_t)ob]);
        if (bound_out) *bound_out = worst;
        if (worst > kFilterScanBound) return false;
        if (moved <= 1e-12) return true;
    }
    return false;  // (not settled: a loop gain bound at or next to 1)
}
// Filter circuits cut in time.  A Filter's recurrence cannot be jumped into — but it forgets: two runs of it over the same input from different
// pairs differ by a fading transient, and once two consecutive outputs coincide they are the same run from there on, bit for bit.  So a
// circuit that is splittable but for its Filters (fused_plan.hpp: constant-f oscillators, closed forms of time, stateless units; no rings, no
// feedback) is cut into segments whose wavefronts start one segment EARLY, from rest, and store only their own chunks; the host then checks
// that what every Filter held when a segment's own chunks began is what the segment before ended with (JitArgs::warm_records) — by
// induction from the first segment the whole render is the sequential one's — and finishes a render whose check fails sequentially from the
// last good segment (dusp_abi.hip render_jit).  How long the warm-up must be is a property of the QUANTISED recurrence (every y rounded to
// f32): its all-pole part has a DC gain of sum|h|, and two trajectories an ulp apart stay an ulp apart with probability ~1 - 1/sum|h| a
// step.  Measured (tools/filter_merge_experiment.c --from-rest, sines and saws): merged within 20 sum|h| samples in every trial down to 400 Hz,
// effectively never at 200 Hz (sum|h| 1500: limit cycles).  Returns the chunks of warm-up to give — 32 sum|h| samples over the circuit's Filters
// in series, a chunk to spare — or 0 where the circuit is not one for it: a connected cutoff (or a per-instance one whose column nobody has looked
// at), sum|h| beyond 400 (below ~390 Hz at 48 kHz), a Filter that reaches an oscillator's frequency.
inline uint32_t jit_warm_chunks(const Program &P, const WavePlan &plan) {
    if (!plan.splittable_but_for_filters) return 0;
    double samples = 256.0;
    // (FM is fine IN FRONT of a Filter — the accumulate passes total the oscillators' increments per segment as in any time-split render — but a
    // Filter's output must not reach an oscillator's frequency: those totals would be the warm-up's, not the chain's)
    std::vector<char> filtered((size_t)std::max(1, P.n_bufs), 0);
    for (const DevOp &op : P.ops) {  // (feed-forward: producers stand in front of their readers)
        bool dep = op.op == OP_FILTER;
        for (int j = 0; j < kMaxIn; j++)
            if (op.in[j].kind == SRC_BUF && op.in[j].idx >= 0 && op.in[j].idx < P.n_bufs && filtered[(size_t)op.in[j].idx]) {
                dep = true;
                if (op.op == OP_OSC && j == 0) return 0;
            }
        if (dep && op.out_buf >= 0 && op.out_buf < P.n_bufs) filtered[(size_t)op.out_buf] = 1;
    }
    for (const DevOp &op : P.ops) {
        if (op.op != OP_FILTER) continue;
        if ((size_t)op.state_slot + 11 > P.init_state.size()) return 0;
        // the cutoffs to answer for: the circuit's constant, or both ends of a per-instance column the renderer has looked at (sum|h| = 1 / (1 - |p|)^2
        // is largest at an end of the range)
        double cutoffs[2];
        int n_cut = 0;
        if (op.in[1].kind == SRC_CONST) cutoffs[n_cut++] = (double)op.in[1].cval;
        else if (op.in[1].kind == SRC_PARAM && op.in[1].pad == kFilterColumnKnown && op.d[0] > 0.0 && op.d[0] <= op.d[1] && op.d[1] < 0.5 * (double)P.g.sample_rate)
            cutoffs[n_cut++] = op.d[0], cutoffs[n_cut++] = op.d[1];
        else return 0;
        double worst = 0.0;
        for (int c = 0; c < n_cut; c++) {
            const double *is = P.init_state.data() + op.state_slot, f = cutoffs[c];
            double k[5];
            if (op.in[1].kind == SRC_CONST && is[0] != 0.0 && f == is[1])
                for (int i = 0; i < 5; i++) k[i] = is[2 + i];
            else butterworth_coefficients(op.attr, f, (double)P.g.sample_rate, k);
            double h1 = 1.0, h2 = 0.0, sum = 1.0;
            int quiet = 0;
            for (int t = 1; t < 200000 && quiet < 8; t++) {
                const double h = -k[3] * h1 - k[4] * h2;
                if (!std::isfinite(h)) return 0;
                h2 = h1, h1 = h, sum += std::fabs(h);
                if (sum > 400.0) return 0;
                quiet = std::fabs(h) < 1e-13 ? quiet + 1 : 0;
            }
            if (quiet < 8) return 0;
            worst = std::max(worst, sum);
        }
        samples += 32.0 * worst;
    }
    return (uint32_t)std::ceil(samples / (double)kChunk);
}
inline bool jit_filter_mod(const Program &P) {
    for (const DevOp &op : P.ops)
        if (op.op == OP_FILTER && op.in[1].kind == SRC_BUF) return true;
    return false;
}
inline int jit_filter_sub(int waves, int per_wave, int stages, size_t lds_left, bool mod) {
    for (int sub : {256, 128, 64, 32})
        if ((!mod || sub <= 64) && jit_filter_tile_bytes(waves * per_wave, sub, stages, mod) <= lds_left) return sub;  // (a connected cutoff: one sample per lane and sub-block)
    return 0;
}
// A circuit whose whole chunk body is a few dozen instructions (constant-f oscillators, Ramp, Timer, the elementwise maps; two units
// at most): several instances per wavefront fill each other's latencies (osc(k): 4 % at four).  Anything heavier is bound by
// VALU issue and by registers, and a second instance only adds to both (tools/wave_ops.py at 1 / 2 / 4 instances per wave:
// Shape 1.67 / 1.84 / 1.95 ms, Multiply(Osc, Shape) 2.33 / 2.45 / 2.51, all-pass 1.93 / 1.88 / 2.04, FM pair 2.10 / 2.09 / 2.16).
inline bool jit_light(const Program &P) {
    int n = 0;
    for (const DevOp &op : P.ops) {
        if (op.op == OP_HOST_ONLY) continue;
        n++;
        const bool lean = (op.op == OP_OSC && op.in[0].kind != SRC_BUF) || op.op == OP_RAMP || op.op == OP_TIMER || op.op == OP_MULTIPLY || op.op == OP_SUM ||
                          (op.op >= OP_MAP_FIRST && op.op <= OP_MAP_LAST && op.op != OP_POW && op.op != OP_DECIBEL_TO_SCALER && op.op != OP_SEMITONE_TO_RATIO);
        if (!lean) return false;
    }
    return n <= 2;
}
// A Delay with a constant delay of less than a chunk reads what two known input samples left in its slot: no ring (JitDelayShort)
// DevOp::pad of a Delay / MonoDelay: the program will be continued (dusp_program_continue) and its rings must be in the reference's own
// state at every launch boundary: the ring-less form does not apply, and a MonoDelay stays on the slot operations
constexpr int kDelayExactRing = 1;
inline bool jit_delay_short(const DevOp &op) {
    if ((op.op == OP_DELAY || op.op == OP_MONO_DELAY) && op.pad == kDelayExactRing) return false;
    if ((op.op == OP_DELAY || op.op == OP_MONO_DELAY) && op.in[1].kind == SRC_PARAM)  // (every instance's delay looked at: fused_plan.hpp)
        return op.in[1].pad == DELAY_REGIME_SHORT && op.ring_len >= 2 * kChunk;
    if ((op.op != OP_DELAY && op.op != OP_MONO_DELAY) || op.in[1].kind != SRC_CONST) return false;
    const double d = (double)op.in[1].cval;  // (MonoDelay writes before it reads: a delay below one sample is a delay)
    return d >= (op.op == OP_MONO_DELAY ? 0.0 : 1.0) && d < (double)(kChunk - 1) && op.ring_len >= 2 * kChunk;  // (a chunk's slots wrap once at most)
}
// A MonoDelay with a constant delay of a chunk at least: the write-once ring of the Delay (JitDelayK<true>)
inline bool jit_mono_write_once(const DevOp &op) {
    if (op.op == OP_MONO_DELAY && op.pad == kDelayExactRing) return false;
    if (op.op == OP_MONO_DELAY && op.in[1].kind == SRC_PARAM) return op.in[1].pad == DELAY_REGIME_LONG;
    if (op.op != OP_MONO_DELAY || op.in[1].kind != SRC_CONST) return false;
    const double d = (double)op.in[1].cval, len = (double)op.ring_len;
    return d >= (double)kChunk && std::floor(d) + (double)kChunk <= len;
}
// units whose ring accesses can meet inside a chunk: ordered slot operations (the same rule as plan_wave's ring_events)
inline bool jit_ring_ops(const DevOp &op) {
    return (op.op == OP_DELAY && !delay_write_once(op) && !jit_delay_short(op)) ||
           (op.op == OP_MONO_DELAY && !jit_delay_short(op) && !jit_mono_write_once(op)) || op.op == OP_READBACK_DELAY ||
           ((op.op == OP_CB_READER || op.op == OP_CB_WRITER) && (op.in[0].kind == SRC_BUF || op.ring_len < kChunk));
}

// A Delay / MonoDelay on the write-once ring protocol whose delay is a constant of the circuit can keep its last chunks of INPUT in LDS
// instead (JitDelayLine): floats of line it needs — the chunks that D + 1 samples back can reach, plus the current one; 0: not such a Delay
// whole_only (DUSP_DELAY_LINE=2): ... and only a WHOLE number of samples
inline size_t jit_delay_line_floats(const DevOp &op, bool whole_only) {
    if (!((op.op == OP_DELAY && delay_write_once(op)) || jit_mono_write_once(op))) return 0;
    if (op.in[1].kind != SRC_CONST || op.pad == kDelayExactRing) return 0;
    double d = (double)op.in[1].cval;
    if (d >= (double)op.ring_len) d = std::fmod(d, (double)op.ring_len);
    if (whole_only && d != std::floor(d)) return 0;
    const size_t D = (size_t)std::floor(d);
    return ((D + 1 + kChunk - 1) / kChunk + 1) * kChunk;
}
inline size_t jit_delay_lines(const Program &P, bool whole_only) {
    size_t n = 0;
    for (const DevOp &op : P.ops) n += jit_delay_line_floats(op, whole_only);
    return n;
}
// Units with a sequential stage (comb family, AHD, SampleRateRedux, MultiChannelOsc) walk their chunk out of a per-wave LDS scratch:
// floats per wavefront the program's units ask for (rows of 256; the largest need).
inline size_t jit_scratch_floats(const Program &P) {
    size_t need = 0;
    for (const DevOp &op : P.ops) {
        size_t n = 0;
        if (op.op == OP_FIXED_DELAY || op.op == OP_COMB_FILTER || op.op == OP_ALL_PASS) n = 512 + (op.op != OP_FIXED_DELAY && op.in[1].kind == SRC_BUF ? 256 : 0);
        if (op.op == OP_AHD) n = (op.in[0].kind == SRC_BUF || op.in[1].kind == SRC_BUF || op.in[2].kind == SRC_BUF) ? 1024 : 256;
        if (op.op == OP_SAMPLE_RATE_REDUX) n = (op.in[0].kind == SRC_BUF || op.in[1].kind == SRC_BUF) ? 768 : 256;
        if (op.op == OP_MULTI_OSC) n = 512;
        if (op.op == OP_SHAPE && op.in[0].kind == SRC_BUF) n = 512;    // 256 doubles: the running sum's addends
        if (jit_ring_ops(op)) n = (op.op == OP_DELAY || op.op == OP_MONO_DELAY) ? 512 : 1024;       // the slot-ownership table; a Delay (JitDelayGather): the chunk's read values, or a table of 512
        if (jit_delay_short(op)) n = 512;                               // the chunk before and this one, side by side
        need = std::max(need, n);
    }
    return need;
}
inline size_t jit_lds_bytes(const JitOptions &opt, bool has_filter) {
    return std::max<size_t>(16, (opt.lds_table >= 0 ? opt.table_bytes : 0) +
                                    (has_filter ? jit_filter_tile_bytes(opt.waves * opt.per_wave, opt.filter_sub, opt.filter_stages, opt.filter_mod) : 0) +
                                    (size_t)opt.waves * opt.scratch_floats * 4);
}

inline bool jit_delay_write_once(const DevOp &op) { return delay_write_once(op); }


// Programs the compiler takes.  Everything else stays on the wave engine's interpreter (wave_engine.hip).
// ---- Voices in a loop.  `Sum.many(voices)` (Sum.js:18-29) of N isomorphic voices is N copies of one small circuit feeding a
// left-deep chain of Sums: as straight-line code its text, its compile time and its footprint in the instruction cache grow
// with N (jit_max_units below).  Where the voices are built from oscillators (constant f, a parameter, or a signal: FM), Multiply,
// Sum and Subtract, the generator emits the voice's units ONCE, inside a loop over the voices: what differs from voice to
// voice — constants, parameter slots, state slots — comes from a table behind the kernel's constants (fk), the oscillators'
// state lives in per-voice arrays, and the chain is the loop's running sum (f32, in voice order: the chain's own roundings).
struct VoicePlan {
    bool ok = false;
    int n_voices = 0, n_channels = 0;
    std::vector<std::vector<int>> ops;    // [voice][t]: the op at template position t (an evaluation order of the voice: producers first)
    std::vector<int> root_pos;            // [channel]: the template position of the voice's outlet that goes into that channel's chain
    std::vector<std::vector<int>> chain;  // [channel]: the chain's Sum ops, bottom up (chain[c][i] adds voice i + 1)
    std::vector<std::vector<int>> tail;   // [channel]: what hangs on the mix, in order: a master gain, an offset, a clip, .. (stateless units with one signal operand)
};
constexpr int kMaxLoopVoices = 256;     // (the oscillators' state arrays are per lane: 76 bytes a voice and constant-f oscillator)
constexpr int kVoiceOperands = 3;       // operands a voice's unit has at most (Shape: duration, min, max)
// units a voice may be made of: oscillators, Ramps (closed form), Multiply / Sum and the stateless maps of at most two operands
inline bool jit_voice_unit(const DevOp &op) {
    return op.op == OP_OSC || op.op == OP_RAMP || op.op == OP_MULTIPLY || op.op == OP_SUM || (op.op >= OP_MAP_FIRST && op.op <= OP_MAP_LAST) ||
           (op.op == OP_SHAPE && op.in[0].kind != SRC_BUF) ||  // (a Shape whose duration is not a signal: closed form)
           (op.op == OP_AHD && op.in[0].kind != SRC_BUF && op.in[1].kind != SRC_BUF && op.in[2].kind != SRC_BUF) ||  // (an AHD with constant times)
           (op.op == OP_PAN && op.in[1].kind != SRC_BUF);  // (a Pan whose position is not a signal: its compensation gain once per voice)
}
inline int jit_voice_operands(const DevOp &op) {
    switch (op.op) {
    case OP_RAMP: return 0;
    case OP_SHAPE: case OP_AHD: return 3;
    case OP_OSC: case OP_POLARITY_INVERT: case OP_ABS: case OP_DECIBEL_TO_SCALER: case OP_SEMITONE_TO_RATIO: case OP_SECONDS_TO_SAMPLES: case OP_FIXED_MULTIPLY: return 1;
    default: return 2;
    }
}

inline bool jit_find_voices(const Program &P, const WavePlan &plan, VoicePlan &V) {
    V = VoicePlan();
    if (!plan.ok || P.out_bufs.empty() || P.out_bufs.size() > 16 || P.ring_samples != 0 || !P.feed_forward || plan.order.size() != P.ops.size()) return false;
    const int n_ops = (int)P.ops.size();
    std::vector<int> producer((size_t)std::max(1, P.n_bufs), -1), pos((size_t)n_ops, 0), owner((size_t)n_ops, -1);
    for (int k = 0; k < n_ops; k++)
        if (P.ops[(size_t)k].out_buf >= 0 && P.ops[(size_t)k].out_buf < P.n_bufs) producer[(size_t)P.ops[(size_t)k].out_buf] = k;
    for (size_t at = 0; at < plan.order.size(); at++) pos[(size_t)plan.order[at]] = (int)at;
    auto src = [&](const DevOperand &o) { return o.kind == SRC_BUF && o.idx >= 0 && o.idx < P.n_bufs ? producer[(size_t)o.idx] : -1; };
    auto n_operands = [](const DevOp &op) { return jit_voice_operands(op); };
    // a voice: everything its root reaches, producers first (operand 0's side, then operand 1's, then the unit)
    auto collect = [&](int root, std::vector<int> &list) -> bool {
        list.clear();
        std::vector<std::pair<int, int>> stack{{root, 0}};
        std::vector<char> seen((size_t)n_ops, 0);
        while (!stack.empty()) {
            auto &[k, j] = stack.back();
            if (k < 0) return false;
            const DevOp &op = P.ops[(size_t)k];
            if (!jit_voice_unit(op) || (op.op == OP_RAMP && k < (int)plan.op_state.size() && plan.op_state[(size_t)k] >= 0)) return false;  // (a Ramp a Retriggerer restarts: not here)
            if ((int)list.size() > 64) return false;
            if (j < n_operands(op)) {
                const DevOperand &o = op.in[j++];
                if (o.kind == SRC_BUF) {
                    const int p = src(o);
                    if (p < 0 || pos[(size_t)p] >= pos[(size_t)stack.back().first]) return false;  // (an edge read a chunk late: not here)
                    if (!seen[(size_t)p]) { seen[(size_t)p] = 1; stack.push_back({p, 0}); }
                } else if (o.kind != SRC_CONST && o.kind != SRC_PARAM) return false;
                continue;
            }
            list.push_back(k);
            stack.pop_back();
        }
        return true;
    };
    // a voice with several outlets (a Pan: one per output channel): the union of what they reach, each unit once
    auto collect_all = [&](const std::vector<int> &roots, std::vector<int> &list, std::vector<int> &root_pos) -> bool {
        list.clear();
        root_pos.clear();
        for (int r : roots) {
            std::vector<int> part;
            if (!collect(r, part)) return false;
            for (int k : part)
                if (std::find(list.begin(), list.end(), k) == list.end()) list.push_back(k);
            root_pos.push_back((int)(std::find(list.begin(), list.end(), r) - list.begin()));
        }
        return (int)list.size() <= 64;
    };
    const int C = (int)P.out_bufs.size();
    // per output channel, from the outlet down: stateless units on the mix (one signal operand each), then the chain's top
    std::vector<int> top((size_t)C, -1);
    std::vector<std::vector<int>> tail_rev((size_t)C);
    for (int c = 0; c < C; c++) {
        int root = producer[(size_t)P.out_bufs[(size_t)c]];
        while (root >= 0 && (int)tail_rev[(size_t)c].size() < 16) {
            const DevOp &op = P.ops[(size_t)root];
            const bool unary = op.op == OP_REPEATER || (op.op >= OP_MAP_FIRST && op.op <= OP_MAP_LAST && jit_voice_operands(op) == 1);
            const bool binary = op.op == OP_MULTIPLY || op.op == OP_SUM || (op.op >= OP_MAP_FIRST && op.op <= OP_MAP_LAST && jit_voice_operands(op) == 2);
            int signal = -1;
            if (unary && op.in[0].kind == SRC_BUF) signal = 0;
            else if (binary && (op.in[0].kind == SRC_BUF) != (op.in[1].kind == SRC_BUF) && (op.in[0].kind == SRC_BUF || op.in[0].kind == SRC_CONST || op.in[0].kind == SRC_PARAM) &&
                     (op.in[1].kind == SRC_BUF || op.in[1].kind == SRC_CONST || op.in[1].kind == SRC_PARAM))
                signal = op.in[0].kind == SRC_BUF ? 0 : 1;
            if (signal < 0) break;
            const int below = src(op.in[signal]);
            if (below < 0 || pos[(size_t)below] >= pos[(size_t)root]) return false;
            tail_rev[(size_t)c].push_back(root);
            root = below;
        }
        if (root < 0 || P.ops[(size_t)root].op != OP_SUM || P.ops[(size_t)root].in[0].kind != SRC_BUF || P.ops[(size_t)root].in[1].kind != SRC_BUF) return false;
        top[(size_t)c] = root;
    }
    std::vector<int> tmpl, tmpl_roots;
    auto same_shape = [&](const std::vector<int> &a, const std::vector<int> &a_roots) {
        if (a.size() != tmpl.size() || a_roots != tmpl_roots) return false;
        for (size_t t = 0; t < a.size(); t++) {
            const DevOp &x = P.ops[(size_t)a[t]], &y = P.ops[(size_t)tmpl[t]];
            if (x.op != y.op || x.attr != y.attr) return false;
            if (((x.op >= OP_MAP_FIRST && x.op <= OP_MAP_LAST) || x.op == OP_PAN || x.op == OP_AHD) && std::memcmp(&x.d[0], &y.d[0], sizeof(double)) != 0) return false;  // (FixedMultiply's factor, Pan's compensation dB, the AHD's sample period: in the text's constants)
            if (x.op == OP_SHAPE && std::memcmp(&x.d[0], &y.d[0], 2 * sizeof(double)) != 0) return false;  // (a Shape's edge values)
            for (int j = 0; j < n_operands(x); j++) {
                if (x.in[j].kind != y.in[j].kind) return false;
                if (x.in[j].kind == SRC_BUF) {
                    const size_t px = (size_t)(std::find(a.begin(), a.end(), src(x.in[j])) - a.begin()), py = (size_t)(std::find(tmpl.begin(), tmpl.end(), src(y.in[j])) - tmpl.begin());
                    if (px != py || px >= a.size()) return false;
                }
            }
        }
        return true;
    };
    // The chains (one per output channel, link for link side by side): `Sum.many` is left-deep (Sum.js:18-29: the chain in operand A,
    // a voice in B); a string's `a + b + c + ..` builds the mirror image (the chain in B).  f32 addition commutes, so either way a
    // channel's mix is the running sum over the voices from the chain's bottom up.
    std::vector<std::vector<int>> rev;
    std::vector<std::vector<int>> chain_rev((size_t)C);
    auto walk = [&](int vi, int ci) -> bool {
        rev.clear();
        for (auto &ch : chain_rev) ch.clear();
        std::vector<int> cur = top, roots((size_t)C), below((size_t)C);
        for (int c = 0; c < C; c++) roots[(size_t)c] = src(P.ops[(size_t)cur[(size_t)c]].in[vi]);
        if (!collect_all(roots, tmpl, tmpl_roots)) return false;
        for (;;) {
            for (int c = 0; c < C; c++) {
                const DevOp &sum = P.ops[(size_t)cur[(size_t)c]];
                if (sum.op != OP_SUM || sum.in[0].kind != SRC_BUF || sum.in[1].kind != SRC_BUF) return false;
                roots[(size_t)c] = src(sum.in[vi]);
                below[(size_t)c] = src(sum.in[ci]);
                if (roots[(size_t)c] < 0 || below[(size_t)c] < 0 || pos[(size_t)roots[(size_t)c]] >= pos[(size_t)cur[(size_t)c]] || pos[(size_t)below[(size_t)c]] >= pos[(size_t)cur[(size_t)c]]) return false;
            }
            std::vector<int> v, v_roots;
            if (!collect_all(roots, v, v_roots) || !same_shape(v, v_roots)) return false;
            rev.push_back(v);
            for (int c = 0; c < C; c++) chain_rev[(size_t)c].push_back(cur[(size_t)c]);
            if ((int)rev.size() > kMaxLoopVoices) return false;
            std::vector<int> first, first_roots;
            if (collect_all(below, first, first_roots) && same_shape(first, first_roots)) {  // the chains' bottom: a voice
                rev.push_back(first);
                return true;
            }
            for (int c = 0; c < C; c++)
                if (P.ops[(size_t)below[(size_t)c]].op != OP_SUM) return false;
            cur = below;
        }
    };
    if (!walk(1, 0) && !walk(0, 1)) return false;
    V.n_voices = (int)rev.size();
    V.n_channels = C;
    if (V.n_voices < 4 || V.n_voices > kMaxLoopVoices) return false;
    V.ops.assign(rev.rbegin(), rev.rend());
    V.root_pos = tmpl_roots;
    V.chain.resize((size_t)C);
    V.tail.resize((size_t)C);
    for (int c = 0; c < C; c++) {
        V.chain[(size_t)c].assign(chain_rev[(size_t)c].rbegin(), chain_rev[(size_t)c].rend());
        V.tail[(size_t)c].assign(tail_rev[(size_t)c].rbegin(), tail_rev[(size_t)c].rend());
    }
    // every unit of the circuit belongs to exactly one voice, one chain or one tail; a voice's outlets are read inside it (or by its Sums) only
    int counted = 0;
    for (int v = 0; v < V.n_voices; v++)
        for (int k : V.ops[(size_t)v]) {
            if (owner[(size_t)k] >= 0) return false;
            owner[(size_t)k] = v;
            counted++;
        }
    for (int c = 0; c < C; c++) {
        for (int k : V.chain[(size_t)c]) {
            if (owner[(size_t)k] >= 0) return false;
            owner[(size_t)k] = V.n_voices;
            counted++;
        }
        for (int k : V.tail[(size_t)c]) {
            if (owner[(size_t)k] >= 0) return false;
            owner[(size_t)k] = V.n_voices + 1;
            counted++;
        }
    }
    if (counted != n_ops) return false;
    for (int k = 0; k < n_ops; k++) {
        if (owner[(size_t)k] > V.n_voices) continue;  // (a tail: its one signal operand is the unit below it, by the way it was found)
        for (int j = 0; j < n_operands(P.ops[(size_t)k]); j++) {
            const int p = src(P.ops[(size_t)k].in[j]);
            if (p < 0 || owner[(size_t)p] == owner[(size_t)k]) continue;
            if (!(owner[(size_t)k] == V.n_voices && owner[(size_t)p] < V.n_voices)) return false;  // (only a chain reads a voice)
            const std::vector<int> &vo = V.ops[(size_t)owner[(size_t)p]];
            bool is_root = false;
            for (int rp : V.root_pos) is_root = is_root || vo[(size_t)rp] == p;
            if (!is_root) return false;
        }
    }
    V.ok = true;
    return true;
}

// Largest circuit (channel-expanded units) the generator takes as straight-line code (DUSP_JIT_MAX_UNITS, read once).  Such code
// outgrows the 64 KB instruction cache from about a hundred units on, but every instruction still serves 64 lanes and the
// waves of a workgroup follow each other through it: measured (tools/big_circuits.py, 256 instances x 1 s, Sum.many of FM pairs)
// 119 units 21.8 -> 3.8 ms, 239 units 44.4 -> 9.4 ms, 479 units 90 -> 21.9 ms against the interpreter.  What bounds it is the
// compile: 6 s / 22 s / 80 s for those three (once per machine with the code-object cache, and in the background while the
// interpreter renders under the default knob) — hence 256.
inline size_t jit_max_units() {
    static const size_t n = [] {
        const char *e = getenv("DUSP_JIT_MAX_UNITS");
        const long v = e ? atol(e) : 0;
        return (size_t)(v >= 1 && v <= 4096 ? v : 256);
    }();
    return n;
}

// From how many units on a circuit of isomorphic voices (VoicePlan) gets the loop instead of straight-line code (DUSP_JIT_LOOP_VOICES)
inline size_t jit_loop_voices_from() {
    static const size_t n = [] {
        const char *e = getenv("DUSP_JIT_LOOP_VOICES");
        const long v = e ? atol(e) : -1;
        return (size_t)(v >= 0 && v <= 4096 ? v : 96);
    }();
    return n;
}

inline bool jit_eligible(const Program &P, const WavePlan &plan, std::string &why) {
    auto no = [&](const char *w) {
        why = w;
        return false;
    };
    if (!plan.ok) return no("not a wave-engine program");
    if (P.ops.size() > jit_max_units()) {
        VoicePlan voices;
        if (!jit_find_voices(P, plan, voices)) return no("more channel-expanded units than the generator takes as straight-line code (DUSP_JIT_MAX_UNITS, 256: compile time)");
    }
    if (P.out_bufs.size() > 16) return no("more than 16 output channels");
    for (size_t k = 0; k < P.ops.size(); k++) {
        const DevOp &op = P.ops[k];
        switch (op.op) {
        case OP_OSC: case OP_MULTIPLY: case OP_SUM: case OP_REPEATER: case OP_INPUT: break;
        case OP_RAMP: break;  // (one a Retriggerer restarts: the closed form counted from its last restart, JitRampR)
        case OP_FILTER: break;  // (a connected cutoff: per-sample coefficients next to P in the stage's tile)
        case OP_DELAY: case OP_MONO_DELAY: case OP_READBACK_DELAY:  // write-once ring protocol, or ordered slot operations
            if (op.ring_len < 1 || op.ring_len >= (1ll << 31)) return no("delay ring out of range");
            break;
        case OP_TIMER:
            if (!(P.init_state[(size_t)op.state_slot] >= 0 && op.d[0] > 0 && op.d[0] < 1e300)) return no("a Timer outside the closed-form regime");
            break;
        case OP_SHAPE: break;  // the running sum t += 1 / duration in closed form where it applies (the kernel checks, per instance), else on one lane
        case OP_FIXED_DELAY: case OP_COMB_FILTER: case OP_ALL_PASS:  // a private ring walked in rounds of its length
            if (op.ring_len < 1 || op.ring_len >= (1ll << 31)) return no("comb ring out of range");
            break;
        case OP_AHD: case OP_SAMPLE_RATE_REDUX: case OP_MULTI_OSC: break;  // serial stage on one lane out of the wave's LDS scratch
        case OP_CB_READER: case OP_CB_WRITER: break;  // (an unconnected offset and a ring of at least a chunk: else plan.ring_events, above)
        case OP_RETRIGGER:  // a Retriggerer of a Shape / an AHD / a Ramp of this circuit: a wave-uniform accumulator; a firing rewrites the target's registers before it ticks
            if (((int)op.d[0] != OP_SHAPE && (int)op.d[0] != OP_AHD && (int)op.d[0] != OP_RAMP) || op.in[0].kind == SRC_BUF) return no("a Retriggerer with a signal-rate `rate`");
            break;
        default:
            if ((op.op >= OP_MAP_FIRST && op.op <= OP_MAP_LAST) || (op.op >= OP_WIDE_FIRST && op.op <= OP_WIDE_LAST)) break;
            return no("a unit the circuit compiler does not emit yet");
        }
    }
    return true;
}

namespace jitgen {

struct Emitter {
    const Program &P;
    const WavePlan &plan;
    JitOptions opt;
    JitSource &out;
    int R = 1;
    std::vector<int> pos_of_op;    // execution position of op k
    std::vector<int> producer;     // buffer -> producing op (-1: none)
    std::vector<char> late;        // buffer is read by an op that ticks before its producer: keep last iteration's registers
    std::vector<char> shared;      // op computes the same chunk for every instance (constants and time only): emitted once per wave
    std::vector<int> fconst_of;    // per (op, operand): index into fk, or -1
    std::vector<long long> dconst_of;  // per op: index of its first f64 constant (Ramp: duration, y0, y1; Delay: ring base, length; maps / Timer: d[0])
    std::string s;
    // The overlapped form of a Filter circuit's chunk (plan_overlap): which units go where, and what unit() is emitting right now
    std::vector<char> grp_early, grp_side, grp_post, split_delay;
    std::vector<char> chained;     // a Filter stage whose input hangs on another stage (a 4-pole filter): fed in place, behind that stage
    std::vector<char> in_chain;    // post units on the way from one stage to the next: emitted between the two
    bool any_side = false;         // plan_overlap: there are side units
    int phase = 0;                 // diagnostic build: the next barrier-to-barrier stamp of the chunk body
    bool in_early = false;         // unit(): emitting the block that works a chunk ahead
    std::map<std::pair<int, int>, std::string> mod_x, mod_f;  // (Filter stage with a connected cutoff, instance slot) -> its input / cutoff registers
    std::vector<char> dbl;         // plan_rotate: buffers an early unit produces and something outside that block reads: written a chunk
                                   // ahead as `vn`, copied into `v` at the top of the chunk they belong to
    bool predeclared = false;      // unit(): the outlets' register arrays are declared at the top of the loop body already
    int delay_half = 0;            // unit(): 0 a write-once Delay's whole tick, 1 its reads only, 2 its writes only

    Emitter(const Program &P_, const WavePlan &plan_, JitOptions o, JitSource &out_) : P(P_), plan(plan_), opt(o), out(out_), R(std::max(1, o.per_wave)) {}

    void line(const std::string &t) { s += t; s += '\n'; }
    static std::string num(long long v) { return std::to_string(v); }

    int add_fk(float v) { out.fk.push_back(v); return (int)out.fk.size() - 1; }
    int add_dk(double v) { out.dk.push_back(v); return (int)out.dk.size() - 1; }

    static bool operand_live(const DevOp &op, int j) {
        switch (op.op) {
        case OP_OSC: case OP_REPEATER: case OP_POLARITY_INVERT: case OP_ABS: case OP_DECIBEL_TO_SCALER: case OP_SEMITONE_TO_RATIO:
        case OP_SECONDS_TO_SAMPLES: case OP_FIXED_MULTIPLY: return j < 1;
        case OP_RAMP: case OP_TIMER: case OP_INPUT: return false;
        case OP_RETRIGGER: return j < 1;
        case OP_SHAPE: case OP_AHD: return j < 3;
        case OP_FIXED_DELAY: case OP_MULTI_OSC: case OP_CB_READER: return j < 1;
        case OP_CB_WRITER: return j < ((op.attr & 2) ? 1 : 2);  // (attr bit 1: nothing to mix on this channel)
        default:
            if (op.op >= OP_WIDE_FIRST && op.op <= OP_WIDE_LAST) return j < op.n_in;
            return j < 2;
        }
    }
    bool reads_late(int consumer_pos, int buf) const {
        const int pr = producer[(size_t)buf];
        return pr < 0 || pos_of_op[(size_t)pr] >= consumer_pos;
    }
    // slot suffix of op k's variables for instance slot r: a shared op has one copy
    std::string sfx(int k, int r) const { return "_" + num(shared[(size_t)k] ? 0 : r); }
    std::string vout(int buf, int r) const { return std::string(in_early && !dbl.empty() && dbl[(size_t)buf] ? "vn" : "v") + num(buf) + "_" + num(r); }
    std::string buf_name(int consumer_pos, int buf, int r) const {
        if (reads_late(consumer_pos, buf)) return "w" + num(buf) + sfx(producer[(size_t)buf], r);
        return vout(buf, shared[(size_t)producer[(size_t)buf]] ? 0 : r);
    }
    // text of operand j of op k at sample c ("c" may be a literal digit), for instance slot r
    std::string opnd(int k, int j, const std::string &c, int r) const {
        const DevOperand &o = P.ops[(size_t)k].in[j];
        if (o.kind == SRC_BUF) return buf_name(pos_of_op[(size_t)k], o.idx, r) + "[" + c + "]";
        if (o.kind == SRC_PARAM) return "p" + num(o.idx) + "_" + num(r);
        return "k" + num(fconst_of[(size_t)k * kMaxIn + j]);
    }
    // operand as a float[4] the unit functions can take by reference
    std::string opnd_array(int k, int j, const std::string &tmp, int r) {
        const DevOperand &o = P.ops[(size_t)k].in[j];
        if (o.kind == SRC_BUF) return buf_name(pos_of_op[(size_t)k], o.idx, r);
        const std::string sc = opnd(k, j, "0", r);
        line("        const float " + tmp + "[4] = {" + sc + ", " + sc + ", " + sc + ", " + sc + "};");
        return tmp;
    }
    std::string in_lds(int table_id) const { return num(jit_table_source(opt, table_id)) + ", " + num(opt.table_form[table_id]); }
    // the constant-f oscillator's fast form on this table: LEAN (delta lerp out of the LDS image or a gather: 2 with the differences
    // in f64, 3 in f32), else 1 FX
    int osc_fast_mode(int table_id) const {
        const int tf = jit_table_source(opt, table_id);
        return (tf == 0 || tf == 1) && opt.table_delta[table_id] ? 1 + opt.table_delta[table_id] : 1;
    }
    bool restarted(int k) const { return P.ops[(size_t)k].op == OP_RAMP && (size_t)k < plan.op_state.size() && plan.op_state[(size_t)k] >= 0; }  // a Ramp some Retriggerer restarts
    std::string table_row(int table_id) const { return "A.tables + (size_t)" + num(table_id) + " * A.table_stride"; }
    std::string ctx(int r) const { return "X[" + num(r) + "]"; }
    int copies(int k) const { return shared[(size_t)k] ? 1 : R; }
    // a scan Filter's coefficients and matrix powers: one set for the wave where the cutoff is a constant of the circuit, one per instance slot where it is a parameter
    std::string scan_k(int k, int r) const { return "fk" + num(k) + (P.ops[(size_t)k].in[1].kind == SRC_PARAM ? "_" + num(r) : std::string()); }
    bool is_filter_stage(int k) const { return P.ops[(size_t)k].op == OP_FILTER && !opt.filter_scan; }  // (a scan Filter is a unit like any other)
    bool is_mod_stage(int k) const { return P.ops[(size_t)k].op == OP_FILTER && P.ops[(size_t)k].in[1].kind == SRC_BUF; }  // a connected cutoff

    // The Filter stage keeps ONE wave busy with the recurrences while the others wait; whatever the chunk holds that neither feeds a
    // Filter nor hangs on one can run on those others meanwhile.  Units are sorted into
    //   early: what a Filter's input needs of this chunk (its same-chunk ancestors).  A write-once Delay on the way contributes its
    //          READS only — they were fetched a chunk ago and need nothing of this chunk — and the search stops there;
    //   post:  what hangs on a Filter's output in this chunk;
    //   side:  the rest, and the WRITES of those Delays: emitted as a block that waves 1.. run between the barriers of a sub-block
    //          (next to wave 0's recurrences) and wave 0 runs after the last sub-block.
    // Every consumer still follows its same-chunk producers, and edges the reference reads late (the `w` registers, copied at the end
    // of the body) do not care about the order inside the body.  Not for circuits whose units share rings (CircleBuffer nodes, slot
    // operations: their mutual order is their meaning) nor for a Filter that hangs on another one.
    bool plan_overlap(const std::vector<char> &used) {
        const size_t n = P.ops.size();
        grp_early.assign(n, 0); grp_side.assign(n, 0); grp_post.assign(n, 0); split_delay.assign(n, 0);
        chained.assign(n, 0); in_chain.assign(n, 0);
        std::vector<int> filters;
        for (size_t at = 0; at < plan.order.size(); at++) {
            const int k = plan.order[at];
            if (!used[(size_t)k]) continue;
            const DevOp &op = P.ops[(size_t)k];
            if (op.op == OP_CB_READER || op.op == OP_CB_WRITER || jit_ring_ops(op)) return false;
            if (is_mod_stage(k)) return false;  // (a connected cutoff: its sub-blocks run in a loop of their own, unit())
            if (is_filter_stage(k)) filters.push_back(k);
        }
        if (filters.empty()) return false;
        auto same_chunk_producer = [&](int k, int j) -> int {
            const DevOp &op = P.ops[(size_t)k];
            const DevOperand &o = op.in[j];
            if (!operand_live(op, j) || o.kind != SRC_BUF || reads_late(pos_of_op[(size_t)k], o.idx)) return -1;
            return producer[(size_t)o.idx];
        };
        for (size_t at = 0; at < plan.order.size(); at++) {  // post: descendants of a Filter stage, in execution order
            const int k = plan.order[at];
            if (!used[(size_t)k]) continue;
            for (int j = 0; j < kMaxIn; j++) {
                const int pr = same_chunk_producer(k, j);
                if (pr >= 0 && (is_filter_stage(pr) || grp_post[(size_t)pr])) grp_post[(size_t)k] = 1;
            }
            if (grp_post[(size_t)k] && is_filter_stage(k)) chained[(size_t)k] = 1;
        }
        std::vector<int> stack;
        for (int f : filters) {  // a chained stage's input may pass through post units only (they are emitted right in front of it)
            if (!chained[(size_t)f]) continue;
            for (int j = 0; j < 2; j++) {  // (the input, and a connected cutoff)
                const int first = same_chunk_producer(f, j);
                if (first >= 0) stack.push_back(first);
            }
            while (!stack.empty()) {
                const int k = stack.back();
                stack.pop_back();
                if (is_filter_stage(k) || in_chain[(size_t)k]) continue;
                if (!grp_post[(size_t)k]) return false;
                in_chain[(size_t)k] = 1;
                for (int j = 0; j < kMaxIn; j++) {
                    const int pr = same_chunk_producer(k, j);
                    if (pr >= 0) stack.push_back(pr);
                }
            }
        }
        for (int f : filters) {
            if (chained[(size_t)f]) continue;
            for (int j = 0; j < 2; j++) {
                const int pr = same_chunk_producer(f, j);
                if (pr >= 0) stack.push_back(pr);
            }
        }
        while (!stack.empty()) {
            const int k = stack.back();
            stack.pop_back();
            if (grp_early[(size_t)k]) continue;
            grp_early[(size_t)k] = 1;
            if ((P.ops[(size_t)k].op == OP_DELAY || P.ops[(size_t)k].op == OP_MONO_DELAY) && !jit_delay_short(P.ops[(size_t)k])) {  // (write-once: the slot-operation kind returned above)
                split_delay[(size_t)k] = 1;
                continue;
            }
            for (int j = 0; j < kMaxIn; j++) {
                const int pr = same_chunk_producer(k, j);
                if (pr >= 0) stack.push_back(pr);
            }
        }
        any_side = false;
        for (size_t k = 0; k < n; k++) {
            if (!used[k] || is_filter_stage((int)k)) continue;
            grp_side[k] = split_delay[k] || (!grp_early[k] && !grp_post[k]);
            any_side = any_side || grp_side[k];
        }
        return true;
    }

    // ops in the input cone of `roots`, for the accumulate passes
    std::vector<char> cone(const std::vector<int> &roots) const {
        std::vector<char> in(P.ops.size(), 0);
        std::vector<int> stack = roots;
        while (!stack.empty()) {
            const int k = stack.back();
            stack.pop_back();
            if (in[(size_t)k]) continue;
            in[(size_t)k] = 1;
            for (int j = 0; j < kMaxIn; j++) {
                const DevOperand &o = P.ops[(size_t)k].in[j];
                if (operand_live(P.ops[(size_t)k], j) && o.kind == SRC_BUF && producer[(size_t)o.idx] >= 0) stack.push_back(producer[(size_t)o.idx]);
            }
        }
        return in;
    }

    // Outlets that cannot be NaN in the fast copy of the chunk loop — their copy-out is one addition (-0 -> +0) instead of a
    // test, a select and the addition (`x || 0`, renderChannelData.js:44).  A bound 2^E on every sample's magnitude, by
    // structure: a constant-f oscillator (finite f in the fast copy) is bounded by its table, a product by the product of
    // its operands' bounds, a sum by twice the larger; constants and parameters on the way are checked once per wave
    // (|k| <= 2^30, which a NaN fails too: `small`, part of the fast copy's condition).  Everything below 2^120 is finite, and
    // finite operands make no NaN in a product or a sum.  Anything else: not known (1000).
    std::vector<std::string> small;  // scalars the fast copy requires to be small
    std::vector<char> out_finite;    // per outlet channel: cannot be NaN in the fast copy
    std::vector<char> visiting;
    int bound_of_operand(int k, int j, std::vector<std::string> &checks) {
        const DevOp &op = P.ops[(size_t)k];
        const DevOperand &o = op.in[j];
        if (o.kind == SRC_BUF) return bound_of_buf(o.idx, checks);
        for (int r = 0; r < (o.kind == SRC_PARAM ? R : 1); r++) {
            const std::string name = opnd(k, j, "0", r);
            if (std::find(checks.begin(), checks.end(), name) == checks.end()) checks.push_back(name);
        }
        return 30;
    }
    int bound_of_buf(int buf, std::vector<std::string> &checks) {
        const int k = buf >= 0 && buf < (int)producer.size() ? producer[(size_t)buf] : -1;
        if (k < 0 || visiting[(size_t)k]) return 1000;
        const DevOp &op = P.ops[(size_t)k];
        int e = 1000;
        visiting[(size_t)k] = 1;
        if (op.op == OP_OSC && op.in[0].kind != SRC_BUF) e = opt.table_bound[op.attr];
        else if (op.op == OP_REPEATER) e = bound_of_operand(k, 0, checks);
        else if (op.op == OP_MULTIPLY || op.op == OP_SUM || op.op == OP_SUBTRACT) {
            const int a = bound_of_operand(k, 0, checks), b = bound_of_operand(k, 1, checks);
            e = a >= 1000 || b >= 1000 ? 1000 : op.op == OP_MULTIPLY ? a + b : std::max(a, b) + 1;
        }
        visiting[(size_t)k] = 0;
        return e > 120 ? 1000 : e;
    }

    // One kernel.  pass_level < 0: the render kernel; else the accumulate pass of that FM level (only the cone of its scanned
    // oscillators, no lookups for them, no stores; ends by writing their phase totals).
    void kernel(int pass_level) {
        const bool render = pass_level < 0;
        std::vector<char> used(P.ops.size(), 1);
        if (!render) {
            std::vector<int> roots;
            for (const auto &sc : out.scans)
                if (sc.level == pass_level) roots.push_back(plan.order[(size_t)sc.op_pos]);
            used = cone(roots);
        }
        const std::string W = num(opt.waves), RR = num(R);
        line("extern \"C\" __global__ void __launch_bounds__(" + W + " * 64) " + (render ? std::string("dusp_jit_render") : "dusp_jit_pass" + num(pass_level)) + "(JitArgs A) {");
        line("    __shared__ __attribute__((aligned(16))) float lds[" + num((long long)(jit_lds_bytes(opt, out.has_filter) / 4)) + "];");
        line("    JitCtx X[" + RR + "];");
        line("    jit_begin<" + W + ", " + num(opt.lds_table) + ", " + RR + ">(A, lds, X);");
        if (out.has_filter) {
            const long long at = (long long)((opt.lds_table >= 0 ? opt.table_bytes : 0) / 4);
            line("    double *tile = (double *)(lds + " + num(at) + ");");
        }
        if (opt.scratch_floats) {
            const long long at = (long long)((opt.lds_table >= 0 ? opt.table_bytes : 0) / 4) +
                                 (out.has_filter ? (long long)(jit_filter_tile_bytes(opt.waves * opt.per_wave, opt.filter_sub, opt.filter_stages, opt.filter_mod) / 4) : 0);
            line("    float *scr = lds + " + num(at) + " + X[0].wave * " + num((long long)opt.scratch_floats) + ";");
        }
        // constants and parameters the used ops name
        std::vector<char> fk_used(out.fk.size(), 0);
        std::vector<int> params_used;
        for (size_t k = 0; k < P.ops.size(); k++) {
            if (!used[k]) continue;
            for (int j = 0; j < kMaxIn; j++) {
                const DevOperand &o = P.ops[k].in[j];
                if (fconst_of[k * kMaxIn + j] >= 0) fk_used[(size_t)fconst_of[k * kMaxIn + j]] = 1;
                if (o.kind == SRC_PARAM && operand_live(P.ops[k], j) && std::find(params_used.begin(), params_used.end(), o.idx) == params_used.end())
                    params_used.push_back(o.idx);
            }
        }
        for (size_t i = 0; i < out.fk.size(); i++)
            if (fk_used[i]) line("    const float k" + num((long long)i) + " = jit_u(A.fk[" + num((long long)i) + "]);");
        for (int p : params_used)
            for (int r = 0; r < R; r++) line("    const float p" + num(p) + "_" + num(r) + " = jit_param(A, " + ctx(r) + ", " + num(p) + ");");
        // state of the units
        std::string fast = "true";  // every constant-f oscillator of the wave qualifies for the 32.32 form
        int filter_ordinal = 0;
        size_t line_at = 0;  // (Delays as LDS lines: the next one's rows)
        for (size_t at = 0; at < plan.order.size(); at++) {
            const int k = plan.order[at];
            if (!used[(size_t)k]) continue;
            const DevOp &op = P.ops[(size_t)k];
            for (int r = 0; r < copies(k); r++) {
                const std::string id = num(k) + "_" + num(r);
                switch (op.op) {
                case OP_OSC:
                    if (op.in[0].kind != SRC_BUF) {
                        line("    JitOscK o" + id + ";");
                        line("    o" + id + ".begin(A, " + ctx(r) + ", " + opnd(k, 0, "0", r) + ", " + num(op.state_slot) + ");");
                        fast += " && o" + id + (osc_fast_mode(op.attr) >= 2 ? ".lean" : ".fx32");
                    } else {
                        int scan_id = -1;
                        for (size_t i = 0; i < out.scans.size(); i++)
                            if (out.scans[i].op_pos == (int)at) scan_id = (int)i;
                        const bool accumulating = !render && plan.osc_level[(size_t)k] >= pass_level;
                        line("    JitOscS o" + id + ";");
                        line("    o" + id + ".begin(A, " + ctx(r) + ", " + num(op.state_slot) + ", " + num(scan_id) + ", " + (accumulating ? "true" : "false") + ");");
                    }
                    break;
                case OP_DELAY: case OP_MONO_DELAY: case OP_READBACK_DELAY:
                    if (jit_ring_ops(op)) {
                        if (op.op == OP_DELAY || op.op == OP_MONO_DELAY) {  // slots replayed lane-parallel where the taps do not decrease, slot rounds where they do
                            line(std::string("    JitDelayGather<") + (op.op == OP_MONO_DELAY ? "true" : "false") + "> q" + id + ";");
                            line("    q" + id + ".begin(A, " + ctx(r) + ", " + num(op.op == OP_MONO_DELAY ? 0 : op.state_slot) + ");");  // (a continued launch: a Delay's carried sample from the last one)
                        } else {
                            line("    JitRingOps q" + id + ";");
                            line("    q" + id + ".begin(A, " + num(op.state_slot) + ");");
                        }
                    } else if (jit_delay_short(op)) {
                        line(std::string("    JitDelayShort<") + (op.op == OP_MONO_DELAY ? "true" : "false") + "> z" + id + ";");
                        line("    z" + id + ".begin(A, " + ctx(r) + ", " + num(op.op == OP_MONO_DELAY ? 0 : op.state_slot) + ", (int64_t)d" + num(dconst_of[(size_t)k] + 1) + ", " + opnd(k, 1, "0", r) + ");");
                    } else if (opt.line_floats && jit_delay_line_floats(op, opt.line_whole_only)) {  // the unit's input in LDS rows of this wave: no ring traffic
                        const size_t n = jit_delay_line_floats(op, opt.line_whole_only);
                        line(std::string("    JitDelayLine<") + (op.op == OP_MONO_DELAY ? "true" : "false") + "> y" + id + ";");
                        line("    y" + id + ".begin(A, " + ctx(r) + ", scr + " + num((long long)(opt.scratch_floats - opt.line_floats + line_at)) + ", " + num((long long)(n / kChunk)) + ", " +
                             num(op.op == OP_MONO_DELAY ? 0 : op.state_slot) + ", (int64_t)d" + num(dconst_of[(size_t)k] + 1) + ", " + opnd(k, 1, "0", r) + ");");
                        line_at += n;
                    } else {
                        line(std::string("    JitDelayK<") + (op.op == OP_MONO_DELAY ? "true" : "false") + ", " + (opt.persistent ? "true" : "false") + "> y" + id + ";");
                        line("    y" + id + ".begin(A, " + ctx(r) + ", " + num(op.op == OP_MONO_DELAY ? 0 : op.state_slot) + ", (int64_t)d" + num(dconst_of[(size_t)k]) + ", (int64_t)d" + num(dconst_of[(size_t)k] + 1) + ", " + opnd(k, 1, "0", r) + ");");
                    }
                    break;
                case OP_PAN:  // (a pan position that is not a signal: its pow() once, here)
                    if (op.in[1].kind != SRC_BUF)
                        line("    const double g" + id + " = map_pan_compensation(" + opnd(k, 1, "0", r) + ", " + (dconst_of[(size_t)k] >= 0 ? "d" + num(dconst_of[(size_t)k]) : std::string("0.0")) + ");");
                    break;
                case OP_TIMER:
                    line("    JitTimer c" + id + ";");
                    line("    c" + id + ".begin(A, " + ctx(r) + ", d" + num(dconst_of[(size_t)k]) + ", " + num(op.state_slot) + ");");
                    break;
                case OP_SHAPE:
                    line("    JitShape s" + id + ";");
                    line("    s" + id + ".begin(A, " + ctx(r) + ", " + (op.in[0].kind == SRC_BUF ? std::string("1.f") : opnd(k, 0, "0", r)) + ", " + num(op.state_slot) + ");");
                    break;
                case OP_FIXED_DELAY: case OP_COMB_FILTER: case OP_ALL_PASS:
                    line("    JitComb b" + id + ";");
                    line("    b" + id + ".begin(A, " + num(op.state_slot) + ");");
                    break;
                case OP_AHD:
                    line("    JitAHD e" + id + ";");
                    line("    e" + id + ".begin(A, " + num(op.state_slot) + ");");
                    if (opt.persistent) line("    if (A.resume) jit_unpark(A, " + ctx(r) + ", " + num(op.out_buf) + ", e" + id + ".prev);");
                    break;
                case OP_RAMP:
                    if (restarted(k)) {
                        line("    JitRampR a" + id + ";");
                        line("    a" + id + ".begin(A, " + num(op.state_slot) + ");");
                    }
                    break;
                case OP_RETRIGGER:
                    line("    JitRetrig r" + id + ";");
                    line("    r" + id + ".begin(A, " + num(op.state_slot) + ");");
                    break;
                case OP_SAMPLE_RATE_REDUX:
                    line("    JitSRR h" + id + ";");
                    line("    h" + id + ".begin(A, " + num(op.state_slot) + ");");
                    break;
                case OP_MULTI_OSC:
                    line("    JitMultiOsc m" + id + ";");
                    line("    m" + id + ".begin(A, " + num(op.state_slot) + ");");
                    break;
                case OP_CB_READER: case OP_CB_WRITER:
                    line(std::string(jit_ring_ops(op) ? "    JitRingOps q" : "    JitCBNode n") + id + ";");
                    line(std::string(jit_ring_ops(op) ? "    q" : "    n") + id + ".begin(A, " + num(op.state_slot) + ");");
                    break;
                default: break;
                }
            }
            if (op.op == OP_FILTER && opt.filter_scan) {  // a scan over the chunk: coefficients and matrix powers once (per instance where the cutoff is a parameter), the memory per instance
                for (int r = 0; r < (op.in[1].kind == SRC_PARAM ? copies(k) : 1); r++) {
                    line("    JitFilterScanK " + scan_k(k, r) + ";");
                    line("    " + scan_k(k, r) + ".begin(A, " + ctx(r) + ", " + num(op.attr) + ", " + opnd(k, 1, "0", r) + ", " + num(op.state_slot) + ");");
                }
                for (int r = 0; r < copies(k); r++) {
                    line("    JitFilterScan f" + num(k) + "_" + num(r) + ";");
                    line("    f" + num(k) + "_" + num(r) + ".begin(A, " + num(op.state_slot) + ");");
                }
            } else if (op.op == OP_FILTER && op.in[1].kind == SRC_BUF) {  // a connected cutoff: coefficients per sample, in the tile next to P
                line("    JitFilterKM<" + W + ", " + RR + ", " + num(opt.filter_sub) + ", " + num((long long)jit_filter_rows_bytes(opt.waves * opt.per_wave, opt.filter_sub, opt.filter_mod)) + "> f" + num(k) + ";");
                line("    f" + num(k) + ".begin(A, X[0], tile, " + num(filter_ordinal++) + ", " + num(op.state_slot) + ");");
                for (int r = 0; r < R; r++) line("    f" + num(k) + ".begin_slot(A, " + ctx(r) + ", " + num(r) + ", " + num(op.state_slot) + ");");
            } else if (op.op == OP_FILTER) {  // one object per Filter: the recurrence's state lives in the lanes of wave 0 (lane = instance of the workgroup)
                const std::string f = op.in[1].kind == SRC_PARAM ? "jit_row_param<" + W + ", " + RR + ">(A, X[0], " + num(op.in[1].idx) + ")" : opnd(k, 1, "0", 0);
                line("    JitFilterK<" + W + ", " + RR + ", " + num(opt.filter_sub) + ", " + (op.in[1].kind == SRC_PARAM ? RR : std::string("1")) + ", " +
                     num((long long)jit_filter_rows_bytes(opt.waves * opt.per_wave, opt.filter_sub, opt.filter_mod)) + "> f" + num(k) + ";");
                line("    f" + num(k) + ".begin(A, X[0], tile, " + num(filter_ordinal++) + ", " + num(op.attr) + ", " + f + ", " + num(op.state_slot) + ");");
                for (int r = 0; r < R; r++)
                    line("    f" + num(k) + ".begin_slot(A, " + ctx(r) + ", " + num(r) + ", " + num(op.attr) + ", " + opnd(k, 1, "0", r) + ", " + num(op.state_slot) + ");");
            }
        }
        // registers of late edges: the producer's previous chunk (outlets start as zeros, SignalChunk.js:7)
        for (int b = 0; b < P.n_bufs; b++)
            if (late[(size_t)b] && producer[(size_t)b] >= 0 && used[(size_t)producer[(size_t)b]])
                for (int r = 0; r < copies(producer[(size_t)b]); r++) {
                    line("    float w" + num(b) + "_" + num(r) + "[4] = {0.f, 0.f, 0.f, 0.f};");
                    if (opt.persistent && render) line("    if (A.resume) jit_unpark(A, " + ctx(r) + ", " + num(b) + ", w" + num(b) + "_" + num(r) + ");");
                }
        // the chunk loop, twice: with the constant-f oscillators in 32.32 fixed point, and in the general form
        if (opt.profile) line("    const unsigned long long stamp_loop = __builtin_readcyclecounter();");
        if (opt.profile) line("    unsigned long long ph[12] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull}, ph_t = 0ull;  // (cycles from barrier to barrier, as wave 0 sees them)");
        out_finite.assign(P.out_bufs.size(), 0);
        if (render) {
            visiting.assign(P.ops.size(), 0);
            for (size_t oc = 0; oc < P.out_bufs.size(); oc++) {
                std::vector<std::string> checks;
                if (bound_of_buf(P.out_bufs[oc], checks) >= 1000) continue;
                out_finite[oc] = 1;
                for (const std::string &c : checks)
                    if (std::find(small.begin(), small.end(), c) == small.end()) small.push_back(c), fast += " && jit_small(" + c + ")";
            }
        }
        line("    if (" + fast + ") {");
        loop(render, pass_level, used, true);
        line("    } else {");
        loop(render, pass_level, used, false);
        line("    }");
        if (opt.profile && render) {
            line("    if (A.debug && X[0].wave == 0 && X[0].lane == 0) {");
            line("        A.debug[(size_t)blockIdx.x * 16 + 0] = __builtin_readcyclecounter() - stamp_loop;");
            std::string ser = "0ull";
            for (size_t k = 0; k < P.ops.size(); k++)
                if (P.ops[k].op == OP_FILTER) ser += " + f" + num((long long)k) + ".cyc_serial";
            line("        A.debug[(size_t)blockIdx.x * 16 + 1] = " + ser + ";");
            line("        A.debug[(size_t)blockIdx.x * 16 + 2] = X[0].g_end - X[0].g_begin;");
            line("    }");
            line("    if (A.debug && X[0].wave == 0 && X[0].lane == 0) for (int i = 0; i < 12; ++i) A.debug[(size_t)blockIdx.x * 16 + 4 + i] = ph[i];");
            line("    if (A.debug && X[0].wave == 1 && X[0].lane == 0) A.debug[(size_t)blockIdx.x * 16 + 3] = " + ser + ";  // (the sub-blocks wave 1 served)");
        }
        if (render) {
            // state write-back: what every unit holds after ceil(n_samples / 256) ticks, in the chunk engine's slot layout
            for (size_t k = 0; k < P.ops.size(); k++)
                if (P.ops[k].op == OP_FILTER && opt.filter_scan) continue;  // (with the other units' state, below)
                else if (P.ops[k].op == OP_FILTER && P.ops[k].in[1].kind == SRC_BUF) {
                    line("    f" + num((long long)k) + ".end(A, X[0], tile, " + num(P.ops[k].state_slot) + ");");
                    for (int r = 0; r < R; r++)
                        line("    f" + num((long long)k) + ".end_slot(A, " + ctx(r) + ", " + num(r) + ", " + num(P.ops[k].attr) + ", " + num(P.ops[k].state_slot) + ");");
                } else if (P.ops[k].op == OP_FILTER) {
                    line("    f" + num((long long)k) + ".end(A, X[0], tile, " + num(P.ops[k].state_slot) + ");");
                    for (int r = 0; r < R; r++) line("    f" + num((long long)k) + ".end_slot(A, " + ctx(r) + ", " + num(r) + ", " + num(P.ops[k].state_slot) + ");");
                }
            for (int r = 0; r < R; r++) {
                line("    if (" + ctx(r) + ".live && " + ctx(r) + ".seg == " + ctx(r) + ".n_seg - 1 && " + ctx(r) + ".lane == 0) {");
                for (size_t k = 0; k < P.ops.size(); k++) {
                    const DevOp &op = P.ops[k];
                    const std::string slot = "A.state[(size_t)" + num(op.state_slot) + " * A.n_pad + " + ctx(r) + ".inst]", id = num((long long)k) + sfx((int)k, r);
                    if (op.op == OP_OSC && op.in[0].kind != SRC_BUF) line("        " + slot + " = o" + id + ".end_phase(A, " + ctx(r) + ");");
                    if (op.op == OP_OSC && op.in[0].kind == SRC_BUF) {
                        int scan_id = 0;
                        for (size_t i = 0; i < out.scans.size(); i++)
                            if (plan.order[(size_t)out.scans[i].op_pos] == (int)k) scan_id = (int)i;
                        line("        " + slot + " = " + (opt.warm ? "A.warm ? JitOscS::warm_end_phase(A, " + ctx(r) + ", " + num(scan_id) + ") : " : std::string()) + "o" + id + ".end_phase();");
                    }
                    if (op.op == OP_RAMP && restarted((int)k)) line("        a" + id + ".end(A, " + ctx(r) + ", d" + num(dconst_of[k]) + ", " + num(op.state_slot) + ");");
                    else if (op.op == OP_RAMP) line("        jit_ramp_end(A, " + ctx(r) + ", d" + num(dconst_of[k]) + ", " + num(op.state_slot) + ");");
                    if (op.op == OP_TIMER)
                        line("        " + slot + " = repeat_add(A.init_state[" + num(op.state_slot) + "], d" + num(dconst_of[k]) + ", (uint64_t)A.n_groups * kChunk);");
                    if (jit_ring_ops(op)) {
                        if (op.op != OP_MONO_DELAY) line("        " + slot + " = q" + id + (op.op == OP_DELAY ? ".rounds.T;" : ".T;"));  // (MonoDelay: no state)
                        continue;
                    }
                    if (op.op == OP_DELAY) line("        " + slot + " = " + (jit_delay_short(op) ? "z" : "y") + id + ".carried;");
                    if (op.op == OP_SHAPE) line("        s" + id + ".end(A, " + ctx(r) + ", " + num(op.state_slot) + ");");
                    if (op.op == OP_FIXED_DELAY || op.op == OP_COMB_FILTER || op.op == OP_ALL_PASS) line("        " + slot + " = (double)b" + id + ".tb;");
                    if (op.op == OP_AHD) line("        e" + id + ".end(A, " + ctx(r) + ", " + num(op.state_slot) + ");");
                    if (op.op == OP_RETRIGGER) line("        " + slot + " = r" + id + ".T;");
                    if (op.op == OP_SAMPLE_RATE_REDUX) line("        h" + id + ".end(A, " + ctx(r) + ", " + num(op.state_slot) + ");");
                    if (op.op == OP_MULTI_OSC) line("        " + slot + " = m" + id + ".phase;");
                    if (op.op == OP_FILTER && opt.filter_scan) line("        f" + id + ".end(A, " + ctx(r) + ", " + scan_k((int)k, r) + ", " + num(op.state_slot) + ");");
                    if (op.op == OP_CB_READER || op.op == OP_CB_WRITER) line("        " + slot + " = n" + id + ".T;");
                }
                line("    }");
            }
        } else {
            line("    if (X[0].live && X[0].lane == 0) {");
            for (size_t i = 0; i < out.scans.size(); i++)
                if (out.scans[i].level == pass_level)
                    line("        A.seg_sum[((size_t)" + num((long long)i) + " * A.n_inst + X[0].inst) * X[0].n_seg + X[0].seg] = o" + num(plan.order[(size_t)out.scans[i].op_pos]) + "_0.packed();");
            line("    }");
        }
        line("}");
        line("");
    }

    // the Filter stage of unit k: the feed-forward halves into registers ...
    void filter_feed(int k, const char *into = "q", bool declare = true) {
        const DevOp &op = P.ops[(size_t)k];
        const bool mod = is_mod_stage(k);
        for (int r = 0; r < R; r++) {
            const std::string x = opnd_array(k, 0, "t" + num(k) + "_" + num(r), r);
            if (!predeclared) line("        float v" + num(op.out_buf) + "_" + num(r) + "[4];");
            if (mod) {  // a connected cutoff: nothing is computed ahead — every sub-block's coefficients and feed-forward halves are made when it is parked (JitFilterKM::parkm)
                mod_x[{k, r}] = x;
                mod_f[{k, r}] = opnd_array(k, 1, "tf" + num(k) + "_" + num(r), r);
                continue;
            }
            if (declare) line("        double " + std::string(into) + num(k) + "_" + num(r) + "[4];");
            line("        f" + num(k) + ".feed(" + ctx(r) + ", " + num(r) + ", " + x + ", " + into + num(k) + "_" + num(r) + ");");
        }
    }
    // May the early units and the feed-forward halves of chunk g+1 run inside chunk g (beside its recurrences)?  They may when they
    // read nothing the reference reads late (last chunk's registers become this chunk's only when it ends).  What they produce and
    // something outside their block reads as well (a unit behind the Filter, a side unit, the PCM store, a late reader) is
    // double-buffered: `dbl`.
    bool plan_rotate(const std::vector<char> &used) {
        dbl.assign((size_t)std::max(1, P.n_bufs), 0);
        for (size_t k = 0; k < P.ops.size(); k++)  // (a connected cutoff: twelve doubles per instance and stage would have to be held twice)
            if (used[k] && is_mod_stage((int)k) && !opt.rotate_mod) return false;
        for (size_t k = 0; k < P.ops.size(); k++) {
            if (used[k] && is_filter_stage((int)k) && !chained[k] && P.ops[k].in[0].kind == SRC_BUF && reads_late(pos_of_op[k], P.ops[k].in[0].idx)) return false;
            if (!used[k] || !grp_early[k]) continue;
            const DevOp &op = P.ops[k];
            if (!split_delay[k])
                for (int j = 0; j < kMaxIn; j++)
                    if (operand_live(op, j) && op.in[j].kind == SRC_BUF && reads_late(pos_of_op[k], op.in[j].idx)) return false;
            const int b = op.out_buf;
            if (b < 0) continue;
            if (late[(size_t)b]) dbl[(size_t)b] = 1;
            for (int ob : P.out_bufs)
                if (ob == b) dbl[(size_t)b] = 1;
            for (size_t c = 0; c < P.ops.size(); c++) {
                if (!used[c]) continue;
                for (int j = 0; j < kMaxIn; j++) {
                    const DevOperand &o = P.ops[c].in[j];
                    if (!operand_live(P.ops[c], j) || o.kind != SRC_BUF || o.idx != b) continue;
                    const bool inside = (grp_early[c] && !split_delay[c]) || (is_filter_stage((int)c) && j == 0);
                    if (!inside) dbl[(size_t)b] = 1;
                }
            }
        }
        return true;
    }
    // ... then, sub-block by sub-block: park P, all recurrences on wave 0 (`beside`: what the other waves do meanwhile), pick y up
    void filter_sub_block(int k, int sb, const std::string &beside, int who = 0) {
        const DevOp &op = P.ops[(size_t)k];
        const std::string f = "f" + num(k);
        auto park = [&](const char *indent) {
            if (is_mod_stage(k)) {  // one sample per lane: 64 / sub-block instances side by side per pass
                const int per = std::max(1, std::min(R, 64 / opt.filter_sub));
                for (int r0 = 0; r0 < R; r0 += per) {
                    const int r1 = std::min(R - 1, r0 + 1);
                    line(std::string(indent) + f + ".parkm<" + num(per) + ">(X[0], tile, " + num(r0) + ", " + num(sb) + ", " + num(op.attr) + ", " + mod_x[{k, r0}] + ", " +
                         mod_f[{k, r0}] + ", " + mod_x[{k, r1}] + ", " + mod_f[{k, r1}] + ");");
                }
                return;
            }
            for (int r = 0; r < R; r++) line(std::string(indent) + f + ".park(" + ctx(r) + ", tile, " + num(r) + ", " + num(sb) + ", q" + num(k) + "_" + num(r) + ");");
        };
        auto stamp = [&]() {
            if (opt.profile && phase < 11) line("        { const unsigned long long now_ = __builtin_readcyclecounter(); ph[" + num(phase++) + "] += now_ - ph_t; ph_t = now_; }");
        };
        park("        ");
        line("        jit_lds_barrier();");
        stamp();
        line("        " + f + ".serial<" + num(opt.filter_block) + ">(X[0], tile, " + num(who) + ");");
        if (!beside.empty()) line("        " + beside);
        line("        jit_lds_barrier();");
        stamp();
        line("        if (" + f + ".failed(tile)) {  // a NaN in some row's recurrence: the sub-block once more, as written");
        park("            ");
        line("            jit_lds_barrier();");
        line("            " + f + ".serial_exact(X[0], tile, " + num(who) + ");");
        line("            jit_lds_barrier();");
        line("        }");
        for (int r = 0; r < R; r++) line("        " + f + ".pick(" + ctx(r) + ", tile, " + num(r) + ", " + num(sb) + ", v" + num(op.out_buf) + "_" + num(r) + ");");
        if (is_mod_stage(k))
            for (int r = 0; r < R; r++) line("        " + f + ".carry(" + num(r) + ", " + num(sb) + ", " + mod_x[{k, r}] + ", " + mod_f[{k, r}] + ");");
        // (rows are per wave: a wave parks into and picks from its own rows only; wave 0 touches the others' between the barriers)
    }

    void loop(bool render, int pass_level, const std::vector<char> &used, bool fx) {
        const bool sorted = render && opt.overlap && plan_overlap(used);
        const bool rotate = sorted && opt.rotate && plan_rotate(used);
        const bool overlapped = sorted && (any_side || rotate);
        if (!rotate) dbl.clear();
        if (rotate)  // what is computed a chunk ahead: the feed-forward halves, and the early units' outlets that others read too
            for (size_t at = 0; at < plan.order.size(); at++) {
                const int k = plan.order[at];
                if (used[(size_t)k] && is_filter_stage(k) && !chained[(size_t)k])
                    for (int r = 0; r < R; r++) line("    double qn" + num(k) + "_" + num(r) + (is_mod_stage(k) ? "[12];" : "[4];"));
                if (used[(size_t)k] && grp_early[(size_t)k] && P.ops[(size_t)k].out_buf >= 0 && dbl[(size_t)P.ops[(size_t)k].out_buf])
                    for (int r = 0; r < copies(k); r++) line("    float vn" + num(P.ops[(size_t)k].out_buf) + "_" + num(r) + "[4];");
            }
        // Scan kernels are bound by vector-ALU issue, and what a chunk hands the next (the Delay's reads fetched a chunk ahead, feedback
        // registers, the scan's carried inputs) costs a register move each at the loop's back edge: with the chunks in PAIRS the second
        // writes where the first reads (configs[3]: 216 -> 201 instructions a chunk, profiles/r04_cfg4_isa.md)
        const bool pairs = render && opt.filter_scan && !opt.warm && !opt.profile && !opt.persistent;
        if (pairs) line("    auto chunk = [&](uint32_t g) __attribute__((always_inline)) {");
        else line("    for (uint32_t g = X[0].g_begin; g < X[0].g_end; ++g) {");
        if (render && opt.warm)  // segments that warm up (JitArgs::warm): what every Filter stage holds where a segment's own chunks begin and end
            for (size_t at = 0; at < plan.order.size(); at++) {
                const int k = plan.order[at];
                if (!used[(size_t)k] || !is_filter_stage(k) || is_mod_stage(k)) continue;
                line("        f" + num(k) + ".capture(A, X[0], tile, g, " + num(P.ops[(size_t)k].state_slot) + ");");
                for (int r = 0; r < R; r++) line("        f" + num(k) + ".capture_slot(A, " + ctx(r) + ", " + num(r) + ", g, " + num(P.ops[(size_t)k].state_slot) + ");");
            }
        phase = 0;
        if (opt.profile && render) line("        ph_t = __builtin_readcyclecounter();");
        if (overlapped) {
            const int subs = kChunk / opt.filter_sub;
            int windows = 0;
            for (size_t at = 0; at < plan.order.size(); at++)
                if (used[(size_t)plan.order[at]] && is_filter_stage(plan.order[at])) windows += subs;
            for (size_t at = 0; at < plan.order.size(); at++) {  // every outlet's registers, visible to all the blocks below
                const int k = plan.order[at];
                if (used[(size_t)k] && P.ops[(size_t)k].out_buf >= 0)
                    for (int r = 0; r < copies(k); r++) line("        float v" + num(P.ops[(size_t)k].out_buf) + "_" + num(r) + "[4];");
            }
            predeclared = true;
            line("        auto side0 = [&]() __attribute__((always_inline)) {  // (a wave runs it whole, in one window it does not serve)");
            for (size_t at = 0; at < plan.order.size(); at++) {
                const int k = plan.order[at];
                if (!used[(size_t)k] || !grp_side[(size_t)k]) continue;
                delay_half = split_delay[(size_t)k] ? 2 : 0;
                unit(k, render, pass_level, fx);
            }
            line("        };");
            if (rotate) line("        auto early = [&](uint32_t g) __attribute__((always_inline)) {  // (its own g: it works for the chunk after this one)");
            in_early = rotate;
            for (size_t at = 0; at < plan.order.size(); at++) {
                const int k = plan.order[at];
                if (!used[(size_t)k] || !grp_early[(size_t)k]) continue;
                delay_half = split_delay[(size_t)k] ? 1 : 0;
                unit(k, render, pass_level, fx);
            }
            delay_half = 0;
            for (size_t at = 0; at < plan.order.size(); at++)
                if (used[(size_t)plan.order[at]] && is_filter_stage(plan.order[at]) && !chained[(size_t)plan.order[at]])
                    filter_feed(plan.order[at], rotate ? "qn" : "q", !rotate);
            in_early = false;
            if (rotate) {
                line("        };");
                line("        if (g == X[0].g_begin) early(g);");
                for (size_t at = 0; at < plan.order.size(); at++) {
                    const int k = plan.order[at];
                    if (used[(size_t)k] && grp_early[(size_t)k] && P.ops[(size_t)k].out_buf >= 0 && dbl[(size_t)P.ops[(size_t)k].out_buf])
                        for (int r = 0; r < copies(k); r++) {
                            const std::string id = num(P.ops[(size_t)k].out_buf) + "_" + num(r);
                            line("        for (int c = 0; c < 4; ++c) v" + id + "[c] = vn" + id + "[c];");
                        }
                }
                for (size_t at = 0; at < plan.order.size(); at++)
                    if (used[(size_t)plan.order[at]] && is_filter_stage(plan.order[at]) && !chained[(size_t)plan.order[at]])
                        for (int r = 0; r < R; r++) {
                            const std::string id = num(plan.order[at]) + "_" + num(r), n_q = is_mod_stage(plan.order[at]) ? "12" : "4";
                            line("        double q" + id + "[" + n_q + "];");
                            line("        for (int c = 0; c < " + n_q + "; ++c) q" + id + "[c] = qn" + id + "[c];");
                        }
            }
            // Waves 0 and 1 take turns at the recurrences (windows 0, 2, .. and 1, 3, ..); wave i does all its other work — its side
            // block, then the next chunk's early units — in window (i + 1) mod windows, which is never one it serves: the work is dealt
            // evenly over the windows and nobody is left with any behind the last sub-block.  (One wavefront per workgroup, or one
            // window per chunk: wave 0 serves, the others work beside it, wave 0 catches up after.)
            const bool turns = opt.waves >= 2 && windows >= 2;
            const std::string work = " side0();" + std::string(rotate ? " if (g + 1 < X[0].g_end) early(g + 1);" : "");
            // Which window a wave works in (turns): never one it serves, and — wavefront i of a workgroup sits on SIMD i mod 4 — as
            // few as possible on the SIMD of the window's serving wave (it leaves a quarter of the issue slots) and no more than
            // two on any other: waves of SIMD 0 in odd windows, of SIMD 1 in even ones, the rest dealt round.
            std::vector<unsigned> workers((size_t)std::max(1, windows), 0u);
            for (int i = 0; turns && i < opt.waves; i++) {
                const int simd = i & 3, group = i >> 2, half = std::max(1, windows / 2);
                const int w = simd == 0 ? (2 * (group % half) + 1) % windows : simd == 1 ? 2 * (group % half) : (group + simd) % windows;
                workers[(size_t)w] |= 1u << i;
            }
            int window = 0;
            for (size_t at = 0; at < plan.order.size(); at++) {
                const int k = plan.order[at];
                if (!used[(size_t)k]) continue;
                if (in_chain[(size_t)k]) unit(k, render, pass_level, fx);  // (on the way from one stage to the next)
                if (!is_filter_stage(k)) continue;
                if (chained[(size_t)k]) filter_feed(k);
                for (int sb = 0; sb < subs; sb++, window++) {
                    std::string beside;
                    if (turns && workers[(size_t)window]) beside = "if ((" + num((long long)workers[(size_t)window]) + "u >> X[0].wave) & 1u) {" + work + " }";
                    else if (window == 0 && opt.waves >= 2) beside = "if (X[0].wave != 0) {" + work + " }";
                    filter_sub_block(k, sb, beside.empty() ? beside : beside + "  // (beside the recurrences)", turns ? (window & 1) : 0);
                }
            }
            if (!turns) line("        if (X[0].wave == 0) {" + work + " }  // (wave 0's own instances)");
            for (size_t at = 0; at < plan.order.size(); at++) {
                const int k = plan.order[at];
                if (used[(size_t)k] && grp_post[(size_t)k] && !in_chain[(size_t)k] && !is_filter_stage(k)) unit(k, render, pass_level, fx);
            }
            predeclared = false;
        } else
        for (size_t at = 0; at < plan.order.size(); at++) {
            const int k = plan.order[at];
            if (!used[(size_t)k]) continue;
            unit(k, render, pass_level, fx);
        }
        if (render)
            for (size_t oc = 0; oc < P.out_bufs.size(); oc++)
                for (int r = 0; r < R; r++)
                {
                    const int from = producer[(size_t)P.out_bufs[oc]];
                    const std::string what = "(A, " + ctx(r) + ", g, " + num((long long)oc) + ", v" + num(P.out_bufs[oc]) + sfx(from, r);
                    if (opt.filter_scan && from >= 0 && P.ops[(size_t)from].op == OP_FILTER && !(fx && out_finite[oc]))  // (what the scan knows about the chunk it wrote)
                        line("        jit_store_scan" + what + ", f" + num(from) + "_" + num(r) + ".finite);");
                    else
                        line("        jit_store<" + std::string(fx && out_finite[oc] ? "true" : "false") + ">" + what + ");");
                }
        for (int b = 0; b < P.n_bufs; b++)
            if (late[(size_t)b] && producer[(size_t)b] >= 0 && used[(size_t)producer[(size_t)b]])
                for (int r = 0; r < copies(producer[(size_t)b]); r++)
                    line("        for (int c = 0; c < 4; ++c) w" + num(b) + "_" + num(r) + "[c] = v" + num(b) + "_" + num(r) + "[c];");
        if (opt.persistent && render) {  // the launch's last chunk: every outlet's samples are parked for the launch that continues it
            line("        if (A.save_bufs && g + 1 == X[0].g_end) {");
            for (int b = 0; b < P.n_bufs; b++)
                if (producer[(size_t)b] >= 0 && used[(size_t)producer[(size_t)b]])
                    for (int r = 0; r < R; r++) line("            jit_park(A, " + ctx(r) + ", " + num(b) + ", v" + num(b) + sfx(producer[(size_t)b], r) + ");");
            line("        }");
        }
        if (opt.profile && render) line("        ph[11] += __builtin_readcyclecounter() - ph_t;  // (behind the last barrier)");
        if (pairs) {
            line("    };");
            line("    {");
            line("        uint32_t g = X[0].g_begin;");
            line("        for (; g + 1 < X[0].g_end; g += 2) {");
            line("            chunk(g);");
            line("            chunk(g + 1);");
            line("        }");
            line("        if (g < X[0].g_end) chunk(g);");
            line("    }");
        } else
            line("    }");
    }

    void unit(int k, bool render, int pass_level, bool fx) {
        const DevOp &op = P.ops[(size_t)k];
        const std::string dref = dconst_of[(size_t)k] >= 0 ? "d" + num(dconst_of[(size_t)k]) : std::string("0.0");
        if (op.op == OP_FILTER && !opt.filter_scan && is_mod_stage(k)) {
            // A connected cutoff: the coefficient code (a division, two polynomials, the math library's tan() for cutoffs outside 0 .. Nyquist)
            // is bulky, so it stands ONCE per pass in a loop over the sub-blocks — with the redo of a sub-block that met a NaN as a second
            // trip through the same code — instead of once per sub-block and case (a kernel of 280 KB otherwise: the instruction cache holds 64).
            filter_feed(k);
            const std::string f = "f" + num(k), N = num(kChunk / opt.filter_sub), kind = num(op.attr);
            const int per = std::max(1, std::min(R, 64 / opt.filter_sub));
            auto parks = [&](const std::string &indent, const std::string &sb) {
                for (int r0 = 0; r0 < R; r0 += per) {
                    const int r1 = std::min(R - 1, r0 + 1);
                    line(indent + f + ".parkm<" + num(per) + ">(X[0], tile, " + num(r0) + ", " + sb + ", " + kind + ", " + mod_x[{k, r0}] + ", " +
                         mod_f[{k, r0}] + ", " + mod_x[{k, r1}] + ", " + mod_f[{k, r1}] + ");");
                }
            };
            // one set of rows: park, serve, pick up — sub-block by sub-block (a second set of rows, sub-block s + 1 parked while s is served,
            // measured slower: the parks then compete with the serving wave's own)
            line("#pragma unroll 1");
            line("        for (int sb = 0; sb < " + N + "; ++sb) {");
            line("            for (bool exact = false;; exact = true) {  // (once; twice when some row's recurrence met a NaN: then as written)");
            parks("                ", "sb");
            line("                jit_lds_barrier();");
            line("                if (exact) " + f + ".serial_exact(X[0], tile, 0); else " + f + ".serial<4>(X[0], tile, 0);");
            line("                jit_lds_barrier();");
            line("                if (exact || !" + f + ".failed(tile)) break;");
            line("            }");
            for (int r = 0; r < R; r++) line("            " + f + ".pick(" + ctx(r) + ", tile, " + num(r) + ", sb, v" + num(op.out_buf) + "_" + num(r) + ");");
            for (int r = 0; r < R; r++) line("            " + f + ".carry(" + num(r) + ", sb, " + mod_x[{k, r}] + ", " + mod_f[{k, r}] + ");");
            line("        }");
            return;
        }
        if (op.op == OP_FILTER && !opt.filter_scan) {
            filter_feed(k);
            for (int sb = 0; sb < kChunk / opt.filter_sub; sb++) filter_sub_block(k, sb, std::string());
            return;
        }
        if (op.op == OP_OSC && op.in[0].kind == SRC_BUF) {
            // A scanned oscillator: all the wave's instances are asked first whether this chunk needs the careful path (increments at
            // or above the sample rate, NaN / Inf, a poisoned phase); then ONE straight-line block serves them all.
            const bool lookup = render || plan.osc_level[(size_t)k] < pass_level;
            std::vector<std::string> fs((size_t)copies(k));
            std::string rare = "false";
            for (int r = 0; r < copies(k); r++) {
                fs[(size_t)r] = opnd_array(k, 0, "t" + num(k) + "_" + num(r), r);
                if (!predeclared) line("        float v" + num(op.out_buf) + "_" + num(r) + "[4];");
                rare += " | o" + num(k) + "_" + num(r) + ".rare(" + ctx(r) + ", " + fs[(size_t)r] + ")";
            }
            for (int careful = 0; careful < 2; careful++) {
                line(careful ? "        } else {" : "        if (!(" + rare + ")) {");
                for (int r = 0; r < copies(k); r++)
                    line("            o" + num(k) + "_" + num(r) + ".tick<" + in_lds(op.attr) + ", " + (lookup ? "true" : "false") + ", " + (careful ? "true" : "false") + ">(" + ctx(r) + ", " +
                             table_row(op.attr) + ", " + fs[(size_t)r] + ", " + vout(op.out_buf, r) + ");");
            }
            line("        }");
            if (!lookup)
                for (int r = 0; r < copies(k); r++)
                    line("        for (int c = 0; c < 4; ++c) " + vout(op.out_buf, r) + "[c] = 0.f;");
            return;
        }
        for (int r = 0; r < copies(k); r++) {
            const std::string id = num(k) + "_" + num(r), v = op.out_buf >= 0 ? vout(op.out_buf, r) : std::string("v_none"), X_ = ctx(r);
            auto decl = [&]() { if (!predeclared) line("        float " + v + "[4];"); };
            auto each = [&](const std::string &expr) {
                decl();
                line("        for (int c = 0; c < 4; ++c) " + v + "[c] = " + expr + ";");
            };
            if (jit_ring_ops(op)) {  // ordered slot operations (JitRingOps): p0 the signal / the offset, p1 the delay / the writer's input
                const bool has_out = op.out_buf >= 0;
                if (has_out) decl();
                else line("        float u" + id + "[4];");
                const std::string p0 = opnd_array(k, 0, "t" + id, r);
                std::string p1 = p0;
                if (op.op != OP_CB_READER && !(op.op == OP_CB_WRITER && (op.attr & 2))) p1 = opnd_array(k, 1, "tz" + id, r);
                line("        q" + id + ".tick" + (op.op == OP_DELAY || op.op == OP_MONO_DELAY ? std::string() : "<" + num(op.op) + ", " + num(op.attr) + ">") + "(A, " + X_ + ", g, scr, (int64_t)" + dref +
                     ", (uint32_t)d" + num(dconst_of[(size_t)k] + 1) + ", " + p0 + ", " + p1 + ", " + (has_out ? v : "u" + id) + ");");
                continue;
            }
            switch (op.op) {
            case OP_OSC:
                decl();
                if (op.in[0].kind != SRC_BUF)
                    line("        o" + id + ".tick<" + in_lds(op.attr) + ", " + num(fx ? osc_fast_mode(op.attr) : 0) + ">(" + X_ + ", " + table_row(op.attr) + ", " + v + ");");
                else {
                    const bool lookup = render || plan.osc_level[(size_t)k] < pass_level;
                    const std::string f = opnd_array(k, 0, "t" + id, r);
                    line("        o" + id + ".tick<" + in_lds(op.attr) + ", " + (lookup ? "true" : "false") + ">(" + X_ + ", " + table_row(op.attr) + ", " + f + ", " + v + ");");
                    if (!lookup) line("        for (int c = 0; c < 4; ++c) " + v + "[c] = 0.f;");
                }
                break;
            case OP_RAMP:
                decl();
                if (restarted(k)) {
                    line("        a" + id + ".tick(" + X_ + ", g, " + dref + ", d" + num(dconst_of[(size_t)k] + 1) + ", d" + num(dconst_of[(size_t)k] + 2) + ", " + v + ");");
                    break;
                }
                line("        jit_ramp<" + std::string(plan.ramp_fastdiv[(size_t)k] ? "true" : "false") + ">(" + X_ + ", g, " + dref + ", d" + num(dconst_of[(size_t)k] + 1) + ", d" +
                     num(dconst_of[(size_t)k] + 2) + ", A.init_state[" + num(op.state_slot) + "], A.init_state[" + num(op.state_slot + 1) + "] != 0.0, " + v + ");");
                break;
            case OP_RETRIGGER: {  // Retriggerer.js:13-24; the target (chained behind it) ticks later in this chunk, from the rewritten state
                int target = -1;
                for (size_t kk = 0; kk < P.ops.size(); kk++)
                    if ((P.ops[kk].op == OP_SHAPE || P.ops[kk].op == OP_AHD || P.ops[kk].op == OP_RAMP) && P.ops[kk].state_slot == op.attr) target = (int)kk;
                const std::string tid = num(target) + sfx(target, r);
                line("        if (r" + id + ".tick(" + X_ + ", " + opnd(k, 0, "0", r) + ")) { " +
                     ((int)op.d[0] == OP_SHAPE ? "s" + tid + ".t = 0.0; s" + tid + ".playing = true;" : (int)op.d[0] == OP_RAMP ? "a" + tid + ".fire(g);" : "e" + tid + ".stage = 1; e" + tid + ".playing = true;") +
                     " }  // trigger()");
                break;
            }
            case OP_MULTIPLY: each(opnd(k, 0, "c", r) + " * " + opnd(k, 1, "c", r)); break;  // Multiply.js:23-34
            case OP_SUM: each(opnd(k, 0, "c", r) + " + " + opnd(k, 1, "c", r)); break;       // Sum.js:33-44
            case OP_REPEATER: each(opnd(k, 0, "c", r)); break;                                // Repeater.js:23-30
            case OP_TIMER:
                decl();
                line("        c" + id + ".tick(" + X_ + ", " + dref + ", " + v + ");");
                break;
            case OP_INPUT:
                decl();
                line("        jit_input(A, " + X_ + ", g, " + num(op.attr) + ", " + v + ");");
                break;
            case OP_FIXED_DELAY: case OP_COMB_FILTER: case OP_ALL_PASS: {
                decl();
                const std::string x = opnd_array(k, 0, "t" + id, r);
                const bool row = op.op != OP_FIXED_DELAY && op.in[1].kind == SRC_BUF;
                const std::string gain = op.op == OP_FIXED_DELAY ? x : opnd_array(k, 1, "tg" + id, r);
                line("        b" + id + ".tick<" + num(op.op) + ", " + (row ? "true" : "false") + ">(A, " + X_ + ", scr, (int64_t)" + dref + ", (uint32_t)d" + num(dconst_of[(size_t)k] + 1) +
                     ", " + x + ", " + gain + ", " + v + ");");
                break;
            }
            case OP_AHD: {
                decl();
                const std::string a0 = opnd_array(k, 0, "ta" + id, r), a1 = opnd_array(k, 1, "th" + id, r), a2 = opnd_array(k, 2, "td" + id, r);
                auto is_row = [&](int j) { return std::string(op.in[j].kind == SRC_BUF ? "true" : "false"); };
                line("        e" + id + ".tick<" + is_row(0) + ", " + is_row(1) + ", " + is_row(2) + ">(" + X_ + ", scr, " + dref + ", " + a0 + ", " + a1 + ", " + a2 + ", " + v + ");");
                break;
            }
            case OP_SAMPLE_RATE_REDUX: {
                decl();
                const std::string a0 = opnd_array(k, 0, "ti" + id, r), a1 = opnd_array(k, 1, "tm" + id, r);
                line("        h" + id + ".tick<" + (op.in[0].kind == SRC_BUF ? "true" : "false") + ", " + (op.in[1].kind == SRC_BUF ? "true" : "false") + ">(" + X_ + ", scr, " + a0 + ", " + a1 + ", " + v + ");");
                break;
            }
            case OP_CB_READER:
                decl();
                line("        n" + id + ".read<" + ((op.attr & 1) ? "true" : "false") + ">(A, " + X_ + ", (int64_t)" + dref + ", (int64_t)d" + num(dconst_of[(size_t)k] + 1) + ", " + opnd(k, 0, "0", r) + ", " + v + ");");
                break;
            case OP_CB_WRITER: {  // no outlet
                const bool mix = !(op.attr & 2);
                const std::string x = mix ? opnd_array(k, 1, "t" + id, r) : std::string();
                if (!mix) line("        const float t" + id + "[4] = {0.f, 0.f, 0.f, 0.f};");
                line("        n" + id + ".write<" + ((op.attr & 1) ? "true" : "false") + ", " + (mix ? "true" : "false") + ">(A, " + X_ + ", (int64_t)" + dref + ", (int64_t)d" + num(dconst_of[(size_t)k] + 1) + ", " +
                     opnd(k, 0, "0", r) + ", " + (mix ? x : "t" + id) + ");");
                break;
            }
            case OP_FILTER: {  // (opt.filter_scan: the others left through the Filter stage above)
                decl();
                const std::string x = opnd_array(k, 0, "t" + id, r);
                line("        f" + id + ".tick(" + X_ + ", " + scan_k(k, r) + ", " + x + ", " + v + ");");
                break;
            }
            case OP_MULTI_OSC: {
                decl();
                const std::string f = opnd_array(k, 0, "t" + id, r);
                line("        m" + id + ".tick<" + in_lds(op.attr) + ">(" + X_ + ", scr, " + table_row(op.attr) + ", " + f + ", " + v + ");");
                break;
            }
            case OP_SHAPE: {
                decl();
                const std::string mn = opnd_array(k, 1, "tn" + id, r), mx = opnd_array(k, 2, "tx" + id, r);
                if (op.in[0].kind == SRC_BUF) {
                    const std::string du = opnd_array(k, 0, "tu" + id, r);
                    line("        s" + id + ".tick_signal(" + X_ + ", scr, " + table_row(op.attr & 255) + ", " + num(op.attr) + ", " + dref + ", d" + num(dconst_of[(size_t)k] + 1) + ", " + du + ", " + mn + ", " + mx + ", " + v + ");");
                    break;
                }
                line("        s" + id + ".tick(" + X_ + ", " + table_row(op.attr & 255) + ", " + num(op.attr) + ", " + dref + ", d" + num(dconst_of[(size_t)k] + 1) + ", " + mn + ", " + mx + ", " + v + ");");
                break;
            }
            case OP_DELAY: case OP_MONO_DELAY: {  // (a MonoDelay gets here with a constant delay below a chunk only: else the slot operations above)
                decl();
                if (jit_delay_short(op)) {
                    const std::string x = opnd_array(k, 0, "t" + id, r);
                    line("        z" + id + ".tick(" + X_ + ", scr, " + x + ", " + v + ");");
                    break;
                }
                if (opt.line_floats && jit_delay_line_floats(op, opt.line_whole_only)) {
                    const std::string x = opnd_array(k, 0, "t" + id, r);
                    line("        y" + id + ".tick(" + X_ + ", " + x + ", " + v + ");");
                    break;
                }
                if (delay_half == 1) {
                    line("        y" + id + ".read(" + X_ + ", " + v + ");");
                    break;
                }
                const std::string x = opnd_array(k, 0, "t" + id, r);
                if (delay_half == 2) line("        y" + id + ".write(" + X_ + ", g, " + x + ");");
                else line("        y" + id + ".tick(" + X_ + ", g, " + x + ", " + v + ");");
                break;
            }
            default:
                if (op.op >= OP_MAP_FIRST && op.op <= OP_MAP_LAST) {  // stateless maps of at most two operands (map_ops.hpp)
                    const std::string y = operand_live(op, 1) ? opnd(k, 1, "c", r) : std::string("0.f");
                    each("map_apply(" + num(op.op) + ", " + opnd(k, 0, "c", r) + ", " + y + ", " + dref + ")");
                } else if (op.op == OP_PAN && op.in[1].kind != SRC_BUF) {
                    each("map_pan(" + opnd(k, 0, "c", r) + ", " + opnd(k, 1, "c", r) + ", " + num(op.attr) + ", g" + id + ")");
                } else {  // Pan, MidiToFrequency, Rescale, CrossFader, VectorMagnitude
                    decl();
                    line("        for (int c = 0; c < 4; ++c) {");
                    std::string arr = "            const float in_[kMaxIn] = {";
                    for (int j = 0; j < kMaxIn; j++) arr += (j ? ", " : "") + (j < op.n_in ? opnd(k, j, "c", r) : opnd(k, 0, "c", r));
                    line(arr + "};");
                    line("            " + v + "[c] = map_wide(" + num(op.op) + ", " + num(op.attr) + ", " + num(op.n_in) + ", in_, " + dref + ");");
                    line("        }");
                }
                break;
            }
        }
    }

    // The kernel of a circuit whose voices run in a loop (VoicePlan above).  fk gets a table behind the ordinary constants:
    // per voice NS floats — for every template operand that is a constant its value, for every one that is a parameter its
    // slot, for every oscillator its state slot (small integers travel as floats).
    bool run_voices(const VoicePlan &V) {
        const std::vector<int> &T = V.ops[0];
        const int NV = V.n_voices, n_t = (int)T.size();
        auto n_operands = [](const DevOp &op) { return jit_voice_operands(op); };
        // slots of the per-voice table
        std::vector<int> slot_of((size_t)n_t * kVoiceOperands, -1), state_slot_of((size_t)n_t, -1);
        int NS = 0;
        for (int t = 0; t < n_t; t++) {
            const DevOp &op = P.ops[(size_t)T[(size_t)t]];
            for (int j = 0; j < n_operands(op); j++)
                if (op.in[j].kind != SRC_BUF) slot_of[(size_t)t * kVoiceOperands + j] = NS++;
            if (op.op == OP_OSC || op.op == OP_RAMP || op.op == OP_SHAPE || op.op == OP_AHD) state_slot_of[(size_t)t] = NS++;
        }
        // f64 constants: the maps' (one per template unit, the same in every voice); then per voice the Ramps' duration, y0, y1
        std::vector<int> dmap((size_t)n_t, -1), dramp((size_t)n_t, -1);
        int ND = 0;
        bool fastdiv = true;
        for (int t = 0; t < n_t; t++) {
            const DevOp &op = P.ops[(size_t)T[(size_t)t]];
            if (op.op >= OP_MAP_FIRST && op.op <= OP_MAP_LAST) dmap[(size_t)t] = add_dk(op.d[0]);
            if (op.op == OP_SHAPE) dmap[(size_t)t] = add_dk(op.d[0]), add_dk(op.d[1]);
            if (op.op == OP_AHD || op.op == OP_PAN) dmap[(size_t)t] = add_dk(op.d[0]);
        }
        for (int t = 0; t < n_t; t++)
            if (P.ops[(size_t)T[(size_t)t]].op == OP_RAMP) dramp[(size_t)t] = ND, ND += 3;
        const int DB = (int)out.dk.size();
        for (int v = 0; v < NV; v++)
            for (int t = 0; t < n_t; t++) {
                const int k = V.ops[(size_t)v][(size_t)t];
                const DevOp &op = P.ops[(size_t)k];
                if (op.op != OP_RAMP) continue;
                add_dk(op.d[0]); add_dk(op.d[1]); add_dk(op.d[2]);
                fastdiv = fastdiv && k < (int)plan.ramp_fastdiv.size() && plan.ramp_fastdiv[(size_t)k];  // (the refined reciprocal only where the host has checked it on every voice's Ramp)
            }
        const int VB = (int)out.fk.size();
        for (int v = 0; v < NV; v++)
            for (int t = 0; t < n_t; t++) {
                const DevOp &op = P.ops[(size_t)V.ops[(size_t)v][(size_t)t]];
                for (int j = 0; j < n_operands(op); j++)
                    if (op.in[j].kind == SRC_CONST) out.fk.push_back(op.in[j].cval);
                    else if (op.in[j].kind == SRC_PARAM) out.fk.push_back((float)op.in[j].idx);
                if (op.op == OP_OSC || op.op == OP_RAMP || op.op == OP_SHAPE || op.op == OP_AHD) out.fk.push_back((float)op.state_slot);
            }
        // the tails' constants: ordinary entries of fk / dk (one unit each, not per voice)
        const int C = V.n_channels;
        std::vector<std::vector<int>> tail_fk((size_t)C), tail_dk((size_t)C);
        for (int c = 0; c < C; c++)
            for (size_t i = 0; i < V.tail[(size_t)c].size(); i++) {
                const DevOp &op = P.ops[(size_t)V.tail[(size_t)c][i]];
                tail_fk[(size_t)c].push_back(-1);
                tail_dk[(size_t)c].push_back(-1);
                for (int j = 0; j < 2 && j < (op.op == OP_REPEATER ? 1 : jit_voice_operands(op)); j++)
                    if (op.in[j].kind == SRC_CONST) tail_fk[(size_t)c][i] = add_fk(op.in[j].cval);
                if (op.op >= OP_MAP_FIRST && op.op <= OP_MAP_LAST) tail_dk[(size_t)c][i] = add_dk(op.d[0]);
            }
        auto pos_in = [&](int k) { return (int)(std::find(T.begin(), T.end(), k) - T.begin()); };
        auto tname = [&](int t) { return "t" + num(t); };
        // operand j of template op t: an array of this chunk's samples, or a wave-uniform scalar out of the voice's table row
        auto scalar = [&](int t, int j) {
            const DevOperand &o = P.ops[(size_t)T[(size_t)t]].in[j];
            const std::string at = "vt[" + num(slot_of[(size_t)t * kVoiceOperands + j]) + "]";
            return o.kind == SRC_PARAM ? "jit_param(A, X[0], (uint32_t)jit_u(" + at + "))" : "jit_u(" + at + ")";
        };
        auto operand = [&](int t, int j, const std::string &c) {
            const DevOperand &o = P.ops[(size_t)T[(size_t)t]].in[j];
            if (o.kind == SRC_BUF) return tname(pos_in(producer[(size_t)o.idx])) + "[" + c + "]";
            return "s" + num(t) + "_" + num(j);
        };
        const std::string W = num(opt.waves), NVs = num(NV), row = "A.fk + (" + num(VB) + " + j * " + num(NS) + ")";
        line("// generated by dusp_amd/csrc/jit_codegen.hpp — a circuit of " + NVs + " isomorphic voices summed by a chain of Sums: the voice's units once, in a loop");
        line("#include \"jit_prelude.hpp\"");
        line("using namespace dusp;");
        line("");
        // scanned oscillators (time-split renders: one accumulate pass + prefix per FM level gives every segment its start phases): scan
        // i = (the template oscillator's ordinal) * NV + voice, as the host's prefix kernel and JitOscS::begin index them
        std::vector<int> scan_base((size_t)n_t, -1), level_of((size_t)n_t, -1);
        {
            int n_scanned = 0;
            for (int t = 0; t < n_t; t++) {
                const int k = T[(size_t)t];
                const DevOp &op = P.ops[(size_t)k];
                if (op.op != OP_OSC || op.in[0].kind != SRC_BUF) continue;
                scan_base[(size_t)t] = n_scanned * NV;
                level_of[(size_t)t] = plan.osc_level[(size_t)k];
                for (int v = 0; v < NV; v++) {
                    const int kv = V.ops[(size_t)v][(size_t)t];
                    out.scans.push_back({pos_of_op[(size_t)kv], P.ops[(size_t)kv].state_slot, plan.osc_level[(size_t)kv]});
                    if (plan.osc_level[(size_t)kv] != level_of[(size_t)t]) { out.why = "voices whose oscillators stack differently"; return false; }
                }
                if (std::find(out.pass_levels.begin(), out.pass_levels.end(), level_of[(size_t)t]) == out.pass_levels.end()) out.pass_levels.push_back(level_of[(size_t)t]);
                n_scanned++;
            }
            std::sort(out.pass_levels.begin(), out.pass_levels.end());
        }
        // pass_level < 0: the render kernel; else the accumulate pass of that FM level (oscillators of that level and above only total their
        // increments over the segment, the ones below render; nothing is stored but the totals)
        auto kernel_text = [&](int pass_level) {
        const bool render = pass_level < 0;
        line("extern \"C\" __global__ void __launch_bounds__(" + W + " * 64) " + (render ? std::string("dusp_jit_render") : "dusp_jit_pass" + num(pass_level)) + "(JitArgs A) {");
        line("    __shared__ __attribute__((aligned(16))) float lds[" + num((long long)(jit_lds_bytes(opt, false) / 4)) + "];");
        line("    JitCtx X[1];");
        line("    jit_begin<" + W + ", " + num(opt.lds_table) + ", 1>(A, lds, X);");
        if (opt.scratch_floats)
            line("    float *scr = lds + " + num((long long)((opt.lds_table >= 0 ? opt.table_bytes : 0) / 4)) + " + X[0].wave * " + num((long long)opt.scratch_floats) + ";");
        for (int t = 0; t < n_t; t++) {
            const DevOp &op = P.ops[(size_t)T[(size_t)t]];
            if (op.op == OP_SHAPE) line("    JitShape e" + num(t) + "[" + NVs + "];");
            if (op.op == OP_AHD) line("    JitAHD a" + num(t) + "[" + NVs + "];");
            if (op.op == OP_PAN) line("    double g" + num(t) + "[" + NVs + "];  // Pan.js:19-29: the centre compensation, a pow() of the pan position alone");
            if (op.op != OP_OSC) continue;
            line(std::string("    ") + (op.in[0].kind != SRC_BUF ? "JitOscKV" : "JitOscS") + " o" + num(t) + "[" + NVs + "];");
        }
        line("    bool fast = true;");
        line("    for (int j = 0; j < " + NVs + "; ++j) {");
        line("        const float *vt = " + row + ";");
        for (int t = 0; t < n_t; t++) {
            const DevOp &op = P.ops[(size_t)T[(size_t)t]];
            if (op.op == OP_SHAPE) line("        e" + num(t) + "[j].begin(A, X[0], " + scalar(t, 0) + ", (int)jit_u(vt[" + num(state_slot_of[(size_t)t]) + "]));");
            if (op.op == OP_AHD) line("        a" + num(t) + "[j].begin(A, (int)jit_u(vt[" + num(state_slot_of[(size_t)t]) + "]));");
            if (op.op == OP_PAN) line("        g" + num(t) + "[j] = map_pan_compensation(" + scalar(t, 1) + ", jit_u(A.dk[" + num(dmap[(size_t)t]) + "]));");
            if (op.op != OP_OSC) continue;
            const std::string slot = "(int)jit_u(vt[" + num(state_slot_of[(size_t)t]) + "])";
            if (op.in[0].kind != SRC_BUF) {
                line("        {");
                line("            JitOscK o;");
                line("            o.begin(A, X[0], " + scalar(t, 0) + ", " + slot + ");");
                line("            o" + num(t) + "[j].keep(A, X[0], o);");
                if (osc_fast_mode(op.attr) >= 2) line("            fast = fast && o.lean;");
                line("        }");
            } else
                line("        o" + num(t) + "[j].begin(A, X[0], " + slot + ", " + num(scan_base[(size_t)t]) + " + j, " + (!render && level_of[(size_t)t] >= pass_level ? "true" : "false") + ");");
        }
        line("    }");
        for (int fx = 1; fx >= 0; fx--) {
            line(fx ? "    if (fast) {" : "    } else {");
            line("    for (uint32_t g = X[0].g_begin; g < X[0].g_end; ++g) {");
            if (render)
                for (int c = 0; c < C; c++)
                    line("        float acc" + num(c) + "[4] = {0.f, 0.f, 0.f, 0.f};" + (c ? "" : "  // (0 + v0: the chain's first voice as it stands, but for the sign of a zero the copy-out drops anyway)"));
            line("#pragma unroll 1");
            line("        for (int j = 0; j < " + NVs + "; ++j) {");
            line("            const float *vt = " + row + ";");
            for (int t = 0; t < n_t; t++) {
                const DevOp &op = P.ops[(size_t)T[(size_t)t]];
                for (int j = 0; j < n_operands(op); j++)
                    if (op.in[j].kind != SRC_BUF && !(op.op == OP_OSC) && !(op.op == OP_SHAPE && j == 0)) line("            const float s" + num(t) + "_" + num(j) + " = " + scalar(t, j) + ";");
            }
            for (int t = 0; t < n_t; t++) {
                const DevOp &op = P.ops[(size_t)T[(size_t)t]];
                line("            float " + tname(t) + "[4];");
                switch (op.op) {
                case OP_OSC:
                    if (op.in[0].kind != SRC_BUF) {
                        const int mode = fx && osc_fast_mode(op.attr) >= 2 ? osc_fast_mode(op.attr) : 0;
                        line("            {");
                        line("                JitOscK o;");
                        line("                o" + num(t) + "[j].lend<" + num(mode) + ">(o);");
                        line("                o.tick<" + in_lds(op.attr) + ", " + num(mode) + ">(X[0], " + table_row(op.attr) + ", " + tname(t) + ");");
                        line("                o" + num(t) + "[j].take<" + num(mode) + ">(o);");
                        line("            }");
                    } else {
                        const std::string f = tname(pos_in(producer[(size_t)op.in[0].idx])), lookup = render || level_of[(size_t)t] < pass_level ? "true" : "false";
                        line("            if (!o" + num(t) + "[j].rare(X[0], " + f + ")) o" + num(t) + "[j].tick<" + in_lds(op.attr) + ", " + lookup + ", false>(X[0], " + table_row(op.attr) + ", " + f + ", " + tname(t) + ");");
                        line("            else o" + num(t) + "[j].tick<" + in_lds(op.attr) + ", " + lookup + ", true>(X[0], " + table_row(op.attr) + ", " + f + ", " + tname(t) + ");");
                        if (lookup == "false") line("            for (int c = 0; c < 4; ++c) " + tname(t) + "[c] = 0.f;");
                    }
                    break;
                case OP_RAMP: {
                    const std::string dd = "A.dk[" + num(DB) + " + j * " + num(ND) + " + " + num(dramp[(size_t)t]), ss = "(int)jit_u(vt[" + num(state_slot_of[(size_t)t]) + "])";
                    line("            jit_ramp<" + std::string(fastdiv ? "true" : "false") + ">(X[0], g, jit_u(" + dd + "]), jit_u(" + dd + " + 1]), jit_u(" + dd + " + 2]), A.init_state[" + ss + "], A.init_state[" + ss +
                         " + 1] != 0.0, " + tname(t) + ");");
                    break;
                }
                case OP_PAN:
                    line("            for (int c = 0; c < 4; ++c) " + tname(t) + "[c] = map_pan(" + operand(t, 0, "c") + ", " + operand(t, 1, "c") + ", " + num(op.attr) + ", jit_u(g" + num(t) + "[j]));");
                    break;
                case OP_AHD: {  // (constant times: closed form unless a stage ends inside the chunk — then lane 0 walks it out of the wave's scratch)
                    std::string tm[3];
                    for (int j = 0; j < 3; j++) {
                        tm[j] = "m" + num(t) + "_" + num(j);
                        line("            const float " + tm[j] + "[4] = {" + operand(t, j, "0") + ", " + operand(t, j, "0") + ", " + operand(t, j, "0") + ", " + operand(t, j, "0") + "};");
                    }
                    line("            a" + num(t) + "[j].tick<false, false, false>(X[0], scr, jit_u(A.dk[" + num(dmap[(size_t)t]) + "]), " + tm[0] + ", " + tm[1] + ", " + tm[2] + ", " + tname(t) + ");");
                    break;
                }
                case OP_SHAPE: {  // (min / max as arrays of this lane's four samples)
                    std::string lim[2];
                    for (int j = 1; j <= 2; j++) {
                        if (op.in[j].kind == SRC_BUF) lim[j - 1] = tname(pos_in(producer[(size_t)op.in[j].idx]));
                        else {
                            lim[j - 1] = "m" + num(t) + "_" + num(j);
                            line("            const float " + lim[j - 1] + "[4] = {" + operand(t, j, "0") + ", " + operand(t, j, "0") + ", " + operand(t, j, "0") + ", " + operand(t, j, "0") + "};");
                        }
                    }
                    line("            e" + num(t) + "[j].tick(X[0], " + table_row(op.attr & 255) + ", " + num(op.attr) + ", jit_u(A.dk[" + num(dmap[(size_t)t]) + "]), jit_u(A.dk[" + num(dmap[(size_t)t] + 1) + "]), " + lim[0] + ", " + lim[1] +
                         ", " + tname(t) + ");");
                    break;
                }
                case OP_MULTIPLY: line("            for (int c = 0; c < 4; ++c) " + tname(t) + "[c] = " + operand(t, 0, "c") + " * " + operand(t, 1, "c") + ";"); break;
                case OP_SUM: line("            for (int c = 0; c < 4; ++c) " + tname(t) + "[c] = " + operand(t, 0, "c") + " + " + operand(t, 1, "c") + ";"); break;
                default:  // the stateless maps (map_ops.hpp)
                    line("            for (int c = 0; c < 4; ++c) " + tname(t) + "[c] = map_apply(" + num(op.op) + ", " + operand(t, 0, "c") + ", " + (n_operands(op) > 1 ? operand(t, 1, "c") : std::string("0.f")) + ", jit_u(A.dk[" +
                         num(dmap[(size_t)t]) + "]));");
                    break;
                }
            }
            if (render)
                for (int ch = 0; ch < C; ch++)
                    line("            for (int c = 0; c < 4; ++c) acc" + num(ch) + "[c] = acc" + num(ch) + "[c] + " + tname(V.root_pos[(size_t)ch]) + "[c];" + (ch ? "" : "  // (Sum.js:33-44: one f32 rounding per link of the chain)"));
            line("        }");
            if (render)
                for (int ch = 0; ch < C; ch++) {
                    std::string mix = "acc" + num(ch);
                    for (size_t i = 0; i < V.tail[(size_t)ch].size(); i++) {  // what hangs on the mix
                        const DevOp &op = P.ops[(size_t)V.tail[(size_t)ch][i]];
                        const std::string u = "u" + num(ch) + "_" + num((long long)i);
                        auto side = [&](int j) -> std::string {
                            const DevOperand &o = op.in[j];
                            if (o.kind == SRC_BUF) return mix + "[c]";
                            return o.kind == SRC_PARAM ? "jit_param(A, X[0], " + num(o.idx) + ")" : "jit_u(A.fk[" + num(tail_fk[(size_t)ch][i]) + "])";
                        };
                        line("        float " + u + "[4];");
                        if (op.op == OP_REPEATER) line("        for (int c = 0; c < 4; ++c) " + u + "[c] = " + side(0) + ";");
                        else if (op.op == OP_MULTIPLY) line("        for (int c = 0; c < 4; ++c) " + u + "[c] = " + side(0) + " * " + side(1) + ";");
                        else if (op.op == OP_SUM) line("        for (int c = 0; c < 4; ++c) " + u + "[c] = " + side(0) + " + " + side(1) + ";");
                        else
                            line("        for (int c = 0; c < 4; ++c) " + u + "[c] = map_apply(" + num(op.op) + ", " + side(0) + ", " + (jit_voice_operands(op) > 1 ? side(1) : std::string("0.f")) + ", jit_u(A.dk[" +
                                 num(tail_dk[(size_t)ch][i]) + "]));");
                        mix = u;
                    }
                    line("        jit_store<false>(A, X[0], g, " + num(ch) + ", " + mix + ");");
                }
            line("    }");
        }
        line("    }");
        if (render) {
            // state write-back: every oscillator's phase after ceil(n_samples / 256) ticks, in the chunk engine's slot layout
            line("    if (X[0].live && X[0].seg == X[0].n_seg - 1 && X[0].lane == 0)");
            line("        for (int j = 0; j < " + NVs + "; ++j) {");
            line("            const float *vt = " + row + ";");
            for (int t = 0; t < n_t; t++) {
                const DevOp &op = P.ops[(size_t)T[(size_t)t]];
                if (op.op == OP_RAMP) line("            jit_ramp_end(A, X[0], jit_u(A.dk[" + num(DB) + " + j * " + num(ND) + " + " + num(dramp[(size_t)t]) + "]), (int)vt[" + num(state_slot_of[(size_t)t]) + "]);");
                if (op.op == OP_SHAPE) line("            e" + num(t) + "[j].end(A, X[0], (int)vt[" + num(state_slot_of[(size_t)t]) + "]);");
                if (op.op == OP_AHD) line("            a" + num(t) + "[j].end(A, X[0], (int)vt[" + num(state_slot_of[(size_t)t]) + "]);");
                if (op.op != OP_OSC) continue;
                line("            A.state[(size_t)(int)vt[" + num(state_slot_of[(size_t)t]) + "] * A.n_pad + X[0].inst] = o" + num(t) + "[j]." + (op.in[0].kind != SRC_BUF ? "end" : "end_phase()") + ";");
            }
            line("        }");
        } else {
            line("    if (X[0].live && X[0].lane == 0)");
            line("        for (int j = 0; j < " + NVs + "; ++j) {");
            for (int t = 0; t < n_t; t++)
                if (scan_base[(size_t)t] >= 0 && level_of[(size_t)t] == pass_level)
                    line("            A.seg_sum[((size_t)(" + num(scan_base[(size_t)t]) + " + j) * A.n_inst + X[0].inst) * X[0].n_seg + X[0].seg] = o" + num(t) + "[j].packed();");
            line("        }");
        }
        line("}");
        line("");
        };
        kernel_text(-1);
        if (plan.splittable)
            for (int L : out.pass_levels) kernel_text(L);
        out.voice_loop = true;
        out.text = s;
        out.ok = true;
        return true;
    }

    bool run() {
        // execution order and who produces what
        pos_of_op.assign(P.ops.size(), 0);
        for (size_t at = 0; at < plan.order.size(); at++) pos_of_op[(size_t)plan.order[at]] = (int)at;
        producer.assign((size_t)std::max(1, P.n_bufs), -1);
        for (size_t k = 0; k < P.ops.size(); k++)
            if (P.ops[k].out_buf >= 0) producer[(size_t)P.ops[k].out_buf] = (int)k;
        late.assign((size_t)std::max(1, P.n_bufs), 0);
        fconst_of.assign(P.ops.size() * kMaxIn, -1);
        dconst_of.assign(P.ops.size(), -1);
        for (size_t k = 0; k < P.ops.size(); k++) {
            const DevOp &op = P.ops[k];
            for (int j = 0; j < kMaxIn; j++) {
                if (!operand_live(op, j)) continue;
                const DevOperand &o = op.in[j];
                if (o.kind == SRC_BUF) {
                    if (o.idx < 0 || o.idx >= P.n_bufs || producer[(size_t)o.idx] < 0) { out.why = "operand without a producer"; return false; }
                    if (reads_late(pos_of_op[k], o.idx)) late[(size_t)o.idx] = 1;
                } else if (o.kind == SRC_CONST)
                    fconst_of[k * kMaxIn + j] = add_fk(o.cval);
            }
            if (op.op == OP_RAMP) {
                dconst_of[k] = add_dk(op.d[0]);
                add_dk(op.d[1]);
                add_dk(op.d[2]);
            } else if (op.op == OP_DELAY || op.op == OP_MONO_DELAY || op.op == OP_READBACK_DELAY) {
                dconst_of[k] = add_dk((double)op.ring_base);
                add_dk((double)op.ring_len);
            } else if (op.op == OP_SHAPE) {
                dconst_of[k] = add_dk(op.d[0]);
                add_dk(op.d[1]);
            } else if (op.op == OP_FIXED_DELAY || op.op == OP_COMB_FILTER || op.op == OP_ALL_PASS || op.op == OP_CB_READER || op.op == OP_CB_WRITER) {
                dconst_of[k] = add_dk((double)op.ring_base);
                add_dk((double)op.ring_len);
            } else if (op.op == OP_AHD)
                dconst_of[k] = add_dk(op.d[0]); else if (op.op == OP_TIMER || (op.op >= OP_MAP_FIRST && op.op <= OP_MAP_LAST) || (op.op >= OP_WIDE_FIRST && op.op <= OP_WIDE_LAST))
                dconst_of[k] = add_dk(op.d[0]);
            if (op.op == OP_FILTER && !opt.filter_scan) { out.has_filter = true; out.n_filters++; }  // (the workgroup-wide stage)
        }
        for (int b : P.out_bufs)
            if (producer[(size_t)b] < 0) { out.why = "the rendered outlet has no producer"; return false; }
        if (R == 1 && opt.voice_loop && !opt.persistent && P.ops.size() > jit_loop_voices_from()) {  // voices in a loop (VoicePlan)
            VoicePlan voices;
            if (jit_find_voices(P, plan, voices)) {
                out.fk.clear();  // (only the voices' table)
                out.dk.clear();
                return run_voices(voices);
            }
        }
        // Units that compute the same chunk for every instance — constants, closed forms of time, and whatever is built from
        // those alone — are emitted once per wave and shared by its R instances (the fused kernels' "one Ramp evaluation per
        // step for the block's voices", in general).  In execution order; an operand read late is not known yet: not shared.
        shared.assign(P.ops.size(), 0);
        for (size_t at = 0; at < plan.order.size() && R > 1; at++) {
            const int k = plan.order[at];
            const DevOp &op = P.ops[(size_t)k];
            bool retriggered = false;  // (a Retriggerer's target follows ITS instance's rate: never shared)
            for (const DevOp &rt : P.ops) retriggered = retriggered || (rt.op == OP_RETRIGGER && rt.attr == op.state_slot && (op.op == OP_SHAPE || op.op == OP_AHD || op.op == OP_RAMP));
            bool ok = !retriggered && (op.op == OP_OSC || (op.op == OP_RAMP && !restarted(k)) || op.op == OP_TIMER || op.op == OP_SHAPE || op.op == OP_MULTIPLY || op.op == OP_SUM || op.op == OP_REPEATER ||
                      (op.op >= OP_MAP_FIRST && op.op <= OP_MAP_LAST) || (op.op >= OP_WIDE_FIRST && op.op <= OP_WIDE_LAST));
            for (int j = 0; ok && j < kMaxIn; j++) {
                if (!operand_live(op, j)) continue;
                const DevOperand &o = op.in[j];
                if (o.kind == SRC_PARAM) ok = false;
                if (o.kind == SRC_BUF) ok = !reads_late((int)at, o.idx) && shared[(size_t)producer[(size_t)o.idx]];
            }
            shared[(size_t)k] = ok ? 1 : 0;
        }
        // scanned oscillators and the accumulate passes a time-split render needs
        for (size_t at = 0; at < plan.order.size(); at++) {
            const DevOp &op = P.ops[(size_t)plan.order[at]];
            if (op.op == OP_OSC && op.in[0].kind == SRC_BUF) {
                const int level = plan.osc_level[(size_t)plan.order[at]];
                out.scans.push_back({(int)at, op.state_slot, level});
                if (std::find(out.pass_levels.begin(), out.pass_levels.end(), level) == out.pass_levels.end()) out.pass_levels.push_back(level);
            }
        }
        std::sort(out.pass_levels.begin(), out.pass_levels.end());
        line("// generated by dusp_amd/csrc/jit_codegen.hpp — one kernel per topologically sorted Circuit");
        if (opt.profile) line("#define DUSP_JIT_PROFILE 1");
        if (opt.filter_fma) line("#define DUSP_FILTER_FMA 1  // (experiment: not the reference's roundings)");
        if (opt.nt_stores) line("#define DUSP_NT_STORES 1  // (PCM leaves past the caches: the circuit's delay rings are what they are for)");
        line("#include \"jit_prelude.hpp\"");
        line("using namespace dusp;");
        // f64 constants are read where they are used (loop-invariant scalar loads)
        for (size_t i = 0; i < out.dk.size(); i++) line("#define d" + num((long long)i) + " jit_u(A.dk[" + num((long long)i) + "])");
        line("");
        kernel(-1);
        if ((plan.splittable || (opt.warm && plan.splittable_but_for_filters)) && R == 1)
            for (int L : out.pass_levels) kernel(L);
        out.text = s;
        out.ok = true;
        return true;
    }
};

}  // namespace jitgen

inline bool jit_generate(const Program &P, const WavePlan &plan, const JitOptions &opt, JitSource &out) {
    out = JitSource();
    jitgen::Emitter e(P, plan, opt, out);
    return e.run();
}

// Descriptor words -> kernel text for a given workgroup geometry, WITHOUT a device: what dusp_circuit_kernel_source (dusp_abi.hip) does in
// front of the run-time compiler, as one host-only function — parse and expand the descriptor (program.hpp), plan it for the wave engine
// (fused_plan.hpp), choose the options a context would choose for the reference's own tables, generate.  Everything the untrusted input
// reaches on the way lives in these headers, so tests/native/hostcheck.cpp drives exactly this under AddressSanitizer / UBSan with
// truncated and corrupted descriptors.  Returns 0 (src.text holds the kernel), 1 (a malformed descriptor or a geometry that does not
// fit: err says why), 2 (a well-formed circuit this path does not take).
struct JitSourceRequest {
    int waves = 4, per_wave = 1;
    bool lds_table = true;        // the sine / 8bit half-table image in LDS and the closed forms, as a context finds them for the reference's tables
    bool continued = false;       // the program will be continued (dusp_program_continue): persistent outlets, rings in the reference's state
    bool lean_recurrence = false; // the Filter stage's recurrence loop with 4 P values per register set
    int scan_knob = 1;            // DUSP_FILTER_SCAN
    bool lean = true;             // DUSP_JIT_LEAN
    int delay_line = 1;           // DUSP_DELAY_LINE (0 never, 1 every constant delay of a chunk at least, 2 the whole ones)
    double cutoff_lo = 0.0, cutoff_hi = 0.0;  // per-instance Filter cutoffs: the range a renderer would have found in their columns (0, 0: not looked at)
};
inline int jit_source_from_descriptor(const double *desc, size_t n_words, const JitSourceRequest &rq, JitSource &src, std::string &err) {
    Program P;
    if (rq.waves < 1 || rq.waves > 16 || rq.per_wave < 1 || rq.per_wave > 4) {
        err = "waves must be 1 .. 16 and per_wave 1 .. 4";
        return 1;
    }
    if (!compile(desc, n_words, P, err, /*continuation=*/true)) return 1;
    const bool continued = rq.continued && (P.ring_samples != 0 || !P.feed_forward);
    if (continued)
        for (DevOp &op : P.ops)
            if (op.op == OP_DELAY || op.op == OP_MONO_DELAY) op.pad = kDelayExactRing;
    if (rq.cutoff_hi > 0.0)
        for (DevOp &op : P.ops)
            if (op.op == OP_FILTER && op.in[1].kind == SRC_PARAM) op.in[1].pad = kFilterColumnKnown, op.d[0] = rq.cutoff_lo, op.d[1] = rq.cutoff_hi;
    WavePlan plan;
    if (!plan_wave(P, plan, continued)) {
        err = "the wave engine cannot run this graph (" + plan.why + ")";
        return 2;
    }
    for (size_t k = 0; k < plan.osc_level.size() && k < P.ops.size(); k++)
        if (plan.osc_level[k] >= 0) P.ops[k].d[0] = (double)plan.osc_level[k];
    std::string why;
    if (!jit_eligible(P, plan, why)) {
        err = "not a circuit the compiler takes (" + why + ")";
        return 2;
    }
    JitOptions opt;
    opt.waves = rq.waves;
    opt.per_wave = rq.per_wave;
    opt.persistent = continued;
    opt.voice_loop = !continued;  // (where the circuit is a sum of isomorphic voices: the form an unsplit render gets)
    if (rq.lean_recurrence) opt.filter_block = 4;
    opt.scratch_floats = jit_scratch_floats(P);
    if (rq.lds_table && P.g.sample_rate % 2 == 0) {  // what a context finds for the reference's tables: sine and 8bit antisymmetric, the rest closed forms
        opt.table_form[1] = 1;  // TABLE_FORM_SAW
        opt.table_form[2] = 2;  // TABLE_FORM_SQUARE
        if (P.g.sample_rate % 4 == 0) opt.table_form[3] = 3;  // TABLE_FORM_TRIANGLE
        opt.table_form[4] = 4;  // TABLE_FORM_8BIT
        for (int k = 0; k < 5; k++) opt.table_bound[k] = 1;  // (the oscillators' tables stay within [-1, 1])
        opt.table_delta[0] = rq.lean ? 1 : 0;  // (the sine table: differences of neighbours exact in f64 at any sample rate, in f32 at some — 44.1 kHz, not 48)
        for (const DevOp &op : P.ops)
            if ((op.op == OP_OSC || op.op == OP_MULTI_OSC) && opt.lds_table < 0 && (op.attr == 0 || op.attr == 4)) {
                opt.lds_table = 0;
                opt.table_bytes = (size_t)half_table_image_bytes((uint32_t)P.g.sample_rate);
            }
    }
    opt.filter_scan = !continued && rq.scan_knob != 0 && jit_filter_scan_ok(P, opt.table_bound, rq.scan_knob == 2 ? 2 : 1);
    opt.filter_stages = opt.filter_scan ? 0 : jit_filter_stages(P);
    opt.filter_mod = !opt.filter_scan && jit_filter_mod(P);
    if (!continued && opt.filter_stages == 0 && rq.per_wave == 1 && rq.delay_line) {
        opt.line_whole_only = rq.delay_line == 2;
        const size_t lines = jit_delay_lines(P, opt.line_whole_only);
        if (lines && opt.table_bytes + 16 * (opt.scratch_floats + lines) * 4 <= 160 * 1024) opt.line_floats = lines, opt.scratch_floats += lines;
    }
    if (plan.has_filter && !opt.filter_scan) {
        const size_t used = opt.table_bytes + (size_t)rq.waves * opt.scratch_floats * 4;
        opt.filter_sub = used < 160 * 1024 ? jit_filter_sub(rq.waves, rq.per_wave, opt.filter_stages, 160 * 1024 - used, opt.filter_mod) : 0;
        if (!opt.filter_sub) {
            err = "waves x per_wave Filter rows do not fit LDS";
            return 1;
        }
    }
    if (!jit_generate(P, plan, opt, src)) {
        err = src.why;
        return 2;
    }
    return 0;
}

}  // namespace dusp
