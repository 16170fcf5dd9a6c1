// program.hpp — host-side compiler: descriptor words -> device program.
//
// Stage 1 (parse)      words -> Graph (units in the reference's process order,
//                      src/Circuit.js:125-131; the order is taken as given).
// Stage 2 (channels)   static channel-count inference.  The reference grows
//                      channel lists lazily at tick time (Multiply.js:29,
//                      Filter.js:31-32, Delay.js:22-24); for the graphs accepted
//                      here the count every outlet ends up with is a function of
//                      the graph alone, so it is fixed before launch.
// Stage 3 (expand)     every (unit, output channel) becomes one mono DevOp that
//                      reads / writes mono chunk buffers, so kernels never see
//                      channel lists.
// Stage 4 (shape)      feed-forward Osc/Ramp/Multiply/Sum trees are matched
//                      against the fused kernels' signatures (fused_engine.hip).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "device_types.hpp"

namespace dusp {

constexpr double kMagic = 1146442576.0;
constexpr size_t kHeaderWords = 12;

struct InletDesc {
    int kind = IN_CONST;       // IN_CONST / IN_CONNECT / IN_PARAM
    std::vector<double> vals;  // CONST: channel values; PARAM: slot per channel
    int src_unit = -1;
    int n_channels(const std::vector<struct UnitDesc> &units) const;
};

struct UnitDesc {
    int op = 0;
    std::vector<InletDesc> inlets;
    std::vector<double> attrs, state;
    int n_out = 1;       // output channels (0: no data outlet)
    int first_buf = -1;  // chunk buffer of output channel 0
    int first_op = -1;   // first DevOp of this unit
    int first_slot = -1; // first state slot of channel 0
    int slots_per_ch = 0;
};

struct RingDesc {
    int nch = 1;
    int64_t len = 0;
    int first_dev_ring = -1;
};

struct DevRing {
    int64_t base = 0, len = 0;  // in samples; arena index = (base + idx) * n_pad + instance
};

struct Graph {
    int sample_rate = 0, chunk = 0, n_params = 0, out_unit = 0;
    int n_inputs = 0;  // host-generated input streams the INPUT units read (largest index + 1)
    int64_t clock0 = 0;
    std::vector<RingDesc> rings;
    std::vector<UnitDesc> units;
    // Channel counts of the first chunks, while they still differ from the settled ones: warm_counts[p][u] = channels of unit
    // u's outlet after chunk p (see infer_channels).  Empty for almost every graph.
    std::vector<std::vector<int>> warm_counts;
};

struct Program {
    Graph g;
    std::vector<DevOp> ops;
    std::vector<std::vector<DevOp>> warm_ops;  // op lists of chunks 0 .. warm_ops.size()-1 when channel counts grow at first
    std::vector<int32_t> out_bufs;
    std::vector<double> init_state;  // one value per state slot, broadcast to every instance
    std::vector<DevRing> dev_rings;
    int n_bufs = 0;
    int64_t ring_samples = 0;  // per instance
    bool feed_forward = true;  // every connection goes from an earlier unit to a later one
};

inline int InletDesc::n_channels(const std::vector<UnitDesc> &units) const {
    if (kind == IN_CONNECT) return units[(size_t)src_unit].n_out;
    return (int)vals.size();
}

inline bool fail(std::string &err, const std::string &msg) {
    err = msg;
    return false;
}

inline bool parse(const double *d, size_t nw, Graph &g, std::string &err) {
    if (!d || nw < kHeaderWords) return fail(err, "descriptor too short");
    if (d[0] != kMagic) return fail(err, "descriptor magic mismatch");
    if (d[1] != 1) return fail(err, "unsupported descriptor version");
    auto as_count = [&](double v, double max, int64_t &out) {
        if (!(v >= 0 && v <= max) || v != std::floor(v)) return false;
        out = (int64_t)v;
        return true;
    };
    int64_t sr, chunk, n_units, n_rings, n_params, out_unit, clock0;
    if (!as_count(d[2], 1 << 22, sr) || sr < 8) return fail(err, "bad sample rate");
    if (!as_count(d[3], 4096, chunk) || chunk != 256)
        return fail(err, "only the standard chunk size 256 is supported (reference src/config.js:6)");
    if (!as_count(d[4], 1 << 20, n_units) || n_units < 1) return fail(err, "bad unit count");
    if (!as_count(d[5], 1 << 16, n_rings)) return fail(err, "bad ring count");
    // counts are bounded by what the descriptor can actually hold (a unit record is at least 4 words, a ring 2) BEFORE
    // anything is sized by them
    if ((size_t)n_units > (nw - kHeaderWords) / 4 || (size_t)n_rings > (nw - kHeaderWords) / 2) return fail(err, "descriptor too short for its unit / ring counts");
    if (!as_count(d[6], 1 << 20, n_params)) return fail(err, "bad parameter count");
    if (!as_count(d[7], (double)n_units - 1, out_unit)) return fail(err, "output unit out of range");
    if (d[8] != 0) return fail(err, "only outlet 0 (\"out\") can be rendered");
    if (!as_count(d[9], 9e15, clock0)) return fail(err, "bad clock");
    g.sample_rate = (int)sr;
    g.chunk = (int)chunk;
    g.n_params = (int)n_params;
    g.out_unit = (int)out_unit;
    g.clock0 = clock0;
    size_t p = kHeaderWords;
    g.rings.resize((size_t)n_rings);
    for (auto &r : g.rings) {
        if (p + 2 > nw) return fail(err, "truncated ring table");
        int64_t nch, len;
        if (!as_count(d[p], 64, nch) || nch < 1) return fail(err, "bad ring channel count");
        if (!as_count(d[p + 1], 1e9, len) || len < 1) return fail(err, "bad ring length");
        r.nch = (int)nch;
        r.len = len;
        p += 2;
    }
    g.units.resize((size_t)n_units);
    for (size_t i = 0; i < g.units.size(); i++) {
        UnitDesc &u = g.units[i];
        const std::string where = "unit " + std::to_string(i) + ": ";
        if (p + 4 > nw) return fail(err, where + "truncated record");
        int64_t op, n_in, n_attr, n_state;
        if (!as_count(d[p], 64, op) || !as_count(d[p + 1], kMaxIn, n_in) || !as_count(d[p + 2], 64, n_attr) ||
            !as_count(d[p + 3], 1 << 12, n_state))
            return fail(err, where + "bad record header");
        p += 4;
        u.op = (int)op;
        u.inlets.resize((size_t)n_in);
        for (auto &in : u.inlets) {
            if (p + 2 > nw) return fail(err, where + "truncated inlet");
            int64_t kind, n;
            if (!as_count(d[p], 2, kind) || !as_count(d[p + 1], 64, n)) return fail(err, where + "bad inlet header");
            p += 2;
            if (p + (size_t)n > nw) return fail(err, where + "truncated inlet values");
            in.kind = (int)kind;
            if (kind == IN_CONNECT) {
                int64_t src;
                if (n != 3 || !as_count(d[p], (double)n_units - 1, src)) return fail(err, where + "bad connection");
                if (d[p + 1] != 0) return fail(err, where + "only outlet 0 carries data");
                in.src_unit = (int)src;
            } else {
                if (n < 1) return fail(err, where + "empty constant");
                in.vals.assign(d + p, d + p + n);
                if (kind == IN_PARAM)
                    for (double v : in.vals) {
                        int64_t slot;
                        if (!as_count(v, (double)n_params - 1, slot)) return fail(err, where + "parameter slot out of range");
                    }
            }
            p += (size_t)n;
        }
        if (p + (size_t)(n_attr + n_state) > nw) return fail(err, where + "truncated attributes/state");
        u.attrs.assign(d + p, d + p + n_attr);
        u.state.assign(d + p + n_attr, d + p + n_attr + n_state);
        p += (size_t)(n_attr + n_state);
        auto need = [&](size_t inl, size_t at, size_t st) {
            return u.inlets.size() == inl && u.attrs.size() == at && (st == SIZE_MAX || u.state.size() == st);
        };
        switch (u.op) {
        case OP_OSC:
            if (!need(1, 1, 1)) return fail(err, where + "bad Osc record");
            if (!(u.attrs[0] >= 0 && u.attrs[0] < kNumTables && u.attrs[0] == std::floor(u.attrs[0])))
                return fail(err, where + "waveform doesn't exist");
            break;
        case OP_RAMP:
            if (!need(0, 3, 2)) return fail(err, where + "bad Ramp record");
            break;
        case OP_MULTIPLY:
        case OP_SUM:
            if (!need(2, 0, 0)) return fail(err, where + "bad Multiply/Sum record");
            break;
        case OP_FILTER: {
            if (!need(2, 1, SIZE_MAX) || u.state.size() < 8) return fail(err, where + "bad Filter record");
            if (u.attrs[0] != 0 && u.attrs[0] != 1) return fail(err, where + "filter kind not supported (LP/HP only)");
            int64_t nch;
            if (!as_count(u.state[7], 64, nch) || u.state.size() != (size_t)(8 + 4 * nch))
                return fail(err, where + "bad Filter state");
            break;
        }
        case OP_DELAY: {
            int64_t md;
            if (!need(2, 1, 0) || !as_count(u.attrs[0], 1e9, md) || md < 1) return fail(err, where + "bad Delay record");
            break;
        }
        case OP_CB_READER:
        case OP_CB_WRITER: {
            int64_t ring;
            if (!need(u.op == OP_CB_READER ? 1 : 2, 2, 1) || !as_count(u.attrs[0], (double)n_rings - 1, ring))
                return fail(err, where + "bad CircleBuffer node record");
            break;
        }
        case OP_REPEATER:
            if (!need(1, 0, 0)) return fail(err, where + "bad Repeater record");
            break;
        case OP_SUBTRACT: case OP_DIVIDE: case OP_POW: case OP_CLIP: case OP_HARD_CLIP_ABOVE: case OP_HARD_CLIP_BELOW: case OP_GAIN:
            if (!need(2, 0, 0)) return fail(err, where + "bad binary map record");
            break;
        case OP_POLARITY_INVERT: case OP_ABS: case OP_DECIBEL_TO_SCALER: case OP_SEMITONE_TO_RATIO: case OP_SECONDS_TO_SAMPLES:
            if (!need(1, 0, 0)) return fail(err, where + "bad unary map record");
            break;
        case OP_FIXED_MULTIPLY:
            if (!need(1, 1, 0)) return fail(err, where + "bad FixedMultiply record");
            break;
        case OP_FIXED_DELAY: case OP_COMB_FILTER: case OP_ALL_PASS: case OP_READBACK_DELAY: case OP_MONO_DELAY: {
            int64_t len, tb = 0;
            const size_t n_in = u.op == OP_FIXED_DELAY ? 1 : 2, n_st = u.op == OP_MONO_DELAY ? 0 : 1;
            if (!need(n_in, 1, n_st) || !as_count(u.attrs[0], 1e9, len) || len < 1) return fail(err, where + "bad delay-family record");
            if (n_st && u.op != OP_READBACK_DELAY && (!as_count(u.state[0], (double)len - 1, tb))) return fail(err, where + "bad ring position");
            if (u.op == OP_READBACK_DELAY && !(u.state[0] >= 0 && u.state[0] < 9e15 && u.state[0] == std::floor(u.state[0])))
                return fail(err, where + "bad ring position");
            break;
        }
        case OP_MULTI_OSC: {
            int64_t nph;
            if (!need(1, 1, SIZE_MAX) || u.state.empty() || !as_count(u.state[0], 64, nph) || u.state.size() != (size_t)(1 + nph))
                return fail(err, where + "bad MultiChannelOsc record");
            if (!(u.attrs[0] >= 0 && u.attrs[0] < kNumTables && u.attrs[0] == std::floor(u.attrs[0])))
                return fail(err, where + "waveform doesn't exist");
            break;
        }
        case OP_PAN:
            if (!need(2, 1, 0)) return fail(err, where + "bad Pan record");
            break;
        case OP_MIDI_TO_FREQUENCY: case OP_VECTOR_MAGNITUDE:
            if (!need(1, 0, 0)) return fail(err, where + "bad unary map record");
            break;
        case OP_RESCALE:
            if (!need(5, 0, 0)) return fail(err, where + "bad Rescale record");
            break;
        case OP_CROSS_FADER:
            if (!need(3, 0, 0)) return fail(err, where + "bad CrossFader record");
            break;
        case OP_TIMER:
            if (!need(0, 1, 1)) return fail(err, where + "bad Timer record");
            break;
        case OP_SAMPLE_RATE_REDUX: {
            int64_t nval;
            if (!need(2, 0, SIZE_MAX) || u.state.size() < 2 || !as_count(u.state[1], 64, nval) || u.state.size() != (size_t)(2 + nval))
                return fail(err, where + "bad SampleRateRedux record");
            break;
        }
        case OP_SHAPE:
            if (!need(3, 5, 3)) return fail(err, where + "bad Shape record");
            if (!(u.attrs[0] >= kFirstShapeTable && u.attrs[0] < kNumTables && u.attrs[0] == std::floor(u.attrs[0])))
                return fail(err, where + "shape table doesn't exist");
            break;
        case OP_AHD:
            if (!need(3, 1, 3)) return fail(err, where + "bad AHD record");
            break;
        case OP_HOST_ONLY:
            if (!need(0, 0, 0)) return fail(err, where + "bad host-only record");
            break;
        case OP_RETRIGGER: {  // inlets: rate (a constant / parameter); attrs: target unit; state: t
            if (!need(1, 1, 1)) return fail(err, where + "bad Retriggerer record");
            if (u.inlets[0].kind == IN_CONNECT || u.inlets[0].vals.size() != 1)
                return fail(err, where + "Retriggerer with a signal-rate rate is not supported on the GPU path");
            if (!(u.attrs[0] >= 0 && u.attrs[0] < (double)n_units) || u.attrs[0] != std::floor(u.attrs[0])) return fail(err, where + "Retriggerer target out of range");
            break;
        }
        case OP_INPUT:
            if (!need(0, 1, 0)) return fail(err, where + "bad input record");
            if (!(u.attrs[0] >= 0 && u.attrs[0] < 4096) || u.attrs[0] != std::floor(u.attrs[0])) return fail(err, where + "input index out of range");
            g.n_inputs = std::max(g.n_inputs, (int)u.attrs[0] + 1);
            break;
        case OP_CONCAT_CHANNELS:
            if (!need(2, 0, 0)) return fail(err, where + "bad ConcatChannels record");
            break;
        case OP_PICK_CHANNEL:
            if (!need(2, 0, 0)) return fail(err, where + "bad PickChannel record");
            // `this.in[this.c[t] % this.in.length][t]` (PickChannel.js:20): the index has to be known when the program
            // is built, and has to be a channel number — anything else makes the reference throw mid-render
            if (u.inlets[1].kind != IN_CONST)
                return fail(err, where + "PickChannel with a signal-rate or per-instance channel index is not supported on the GPU path");
            if (!(u.inlets[1].vals[0] >= 0 && u.inlets[1].vals[0] < 9e15 && u.inlets[1].vals[0] == std::floor(u.inlets[1].vals[0])))
                return fail(err, where + "PickChannel index is not a channel number (the reference would throw a TypeError)");
            break;
        default:
            return fail(err, where + "unknown opcode " + std::to_string(u.op));
        }
    }
    if (p != nw) return fail(err, "trailing words after the last unit");
    return true;
}

inline int unit_channels(const Graph &g, const UnitDesc &u) {
    auto nin = [&](size_t k) { return u.inlets[k].n_channels(g.units); };
    switch (u.op) {
    case OP_OSC:
    case OP_RAMP: return 1;                                               // mono outlets (Osc.js:12, Ramp.js:6)
    case OP_MULTIPLY:
    case OP_SUM: return std::max(nin(0), nin(1));                         // Multiply.js:26, Sum.js:35-37
    case OP_FILTER: return std::max(1, nin(0));                           // Filter.js:31-32
    case OP_DELAY: return std::max(1, std::max(nin(0), nin(1)));          // Delay.js:21
    case OP_CB_READER: return std::max(1, g.rings[(size_t)u.attrs[0]].nch);  // CircleBufferNode.js:19-22
    case OP_CB_WRITER: case OP_HOST_ONLY: case OP_RETRIGGER: return 0;    // no data outlet
    case OP_REPEATER: return std::max(1, nin(0));                         // Repeater.js:24-25
    case OP_SUBTRACT: case OP_DIVIDE: case OP_POW: return std::max(nin(0), nin(1));
    case OP_FIXED_MULTIPLY: return 1;                                     // mono in / mono out
    case OP_FIXED_DELAY: case OP_COMB_FILTER: case OP_ALL_PASS: case OP_MONO_DELAY: return 1;  // mono units
    case OP_READBACK_DELAY: return std::max(1, std::max(nin(0), nin(1)));  // ReadBackDelay.js:27
    case OP_MULTI_OSC: return std::max(1, nin(0));                         // MultiChannelOsc.js:22
    case OP_PAN: return 2;                                                 // Pan.js:8
    case OP_MIDI_TO_FREQUENCY: return 1;                                   // only channel 0 is ever stored (MidiToFrequency.js:18)
    case OP_RESCALE: return std::max(1, nin(0));                           // Rescale.js:26
    case OP_CROSS_FADER: return std::max(1, std::max(nin(0), nin(1)));     // CrossFader.js:22
    case OP_VECTOR_MAGNITUDE: case OP_TIMER: case OP_PICK_CHANNEL: case OP_SHAPE: case OP_AHD: case OP_INPUT: return 1;  // mono outlets
    case OP_SAMPLE_RATE_REDUX: return std::max(1, nin(0));                 // SampleRateRedux.js:23-24
    case OP_CONCAT_CHANNELS: return nin(0) + nin(1);                       // ConcatChannels.js:18
    }
    if (u.op >= OP_MAP_FIRST && u.op <= OP_MAP_LAST) return std::max(1, nin(0));  // loop `c < this.in.length`
    return 1;
}

// The reference grows channel lists lazily while it ticks.  One pass over the units in process order, with every outlet
// starting at one channel (Piglet.js:13), is exactly what its first chunk does: a unit that is scheduled BEFORE one of
// its inputs (a feedback edge, or one of the DAG edges the reference's process order gets "wrong") still sees that
// input's previous channel count.  Pass p therefore yields the counts of chunk p; they only grow, and once a pass
// changes nothing they are settled.  Graphs whose first pass is already settled need nothing special; otherwise the
// counts of the first chunks are kept and those chunks run their own op lists (Program::warm_ops).

inline bool infer_channels(Graph &g, std::string &err) {
    for (auto &u : g.units) u.n_out = (u.op == OP_CB_WRITER || u.op == OP_HOST_ONLY || u.op == OP_RETRIGGER) ? 0 : 1;
    std::vector<std::vector<int>> hist;
    for (int pass = 0;; pass++) {
        bool changed = false;
        for (auto &u : g.units) {
            for (auto &in : u.inlets)
                if (in.kind == IN_CONNECT && g.units[(size_t)in.src_unit].n_out == 0)
                    return fail(err, "a unit without a data outlet (CircleBufferWriter, Retriggerer) feeds an inlet");
            int n = unit_channels(g, u);
            if (n > 64) return fail(err, "more than 64 channels on one outlet are not supported on the GPU path");
            if (n != u.n_out) { u.n_out = n; changed = true; }
        }
        if (pass > 0 && !changed) break;
        std::vector<int> now;
        for (auto &u : g.units) now.push_back(u.n_out);
        hist.push_back(now);
        if (pass > 70) return fail(err, "channel counts do not settle");
    }
    // hist[p] = counts after chunk p; the last entry equals the settled counts: drop it and every trailing equal
    while (!hist.empty()) {
        bool same = true;
        for (size_t i = 0; i < g.units.size(); i++) same = same && hist.back()[i] == g.units[i].n_out;
        if (!same) break;
        hist.pop_back();
    }
    for (auto &h : hist)
        for (size_t i = 0; i < g.units.size(); i++) {
            const int op = g.units[i].op;
            // a channel that appears late must start from the same state as in the reference: true for stateless units,
            // Filter (per-channel memory, coefficients a pure function of f) and MultiChannelOsc (phase[c] || 0), but
            // not for units whose channels share a running counter
            if (h[i] != g.units[i].n_out && (op == OP_DELAY || op == OP_READBACK_DELAY || op == OP_SAMPLE_RATE_REDUX))
                return fail(err, "channel count of a Delay / ReadBackDelay / SampleRateRedux grows after the first chunk "
                                 "(not supported on the GPU path)");
        }
    if (hist.size() > (size_t)kMaxWarmChunks)
        return fail(err, "channel counts take more than " + std::to_string(kMaxWarmChunks) + " chunks to settle (not supported on the GPU path)");
    g.warm_counts = hist;
    for (auto &u : g.units) {
        if (u.op == OP_DELAY && u.inlets[1].n_channels(g.units) > u.inlets[0].n_channels(g.units))
            return fail(err, "Delay with more delay channels than input channels is not supported "
                             "(the reference aliases its input chunk, Delay.js:23)");
        if (u.op == OP_FILTER && u.inlets[1].n_channels(g.units) < 1) return fail(err, "Filter without f");
        if (u.op == OP_VECTOR_MAGNITUDE && u.inlets[0].n_channels(g.units) > kMaxIn)
            return fail(err, "VectorMagnitude of more than " + std::to_string(kMaxIn) + " channels is not supported on the GPU path");
        if (u.op == OP_CONCAT_CHANNELS && u.n_out > 64) return fail(err, "ConcatChannels: too many channels");
    }
    if (g.units[(size_t)g.out_unit].n_out < 1) return fail(err, "the rendered unit has no data outlet");
    return true;
}

inline DevOperand make_operand(const Graph &g, const InletDesc &in, int ch) {
    DevOperand o{};
    const int n = in.n_channels(g.units);
    const int c = ch % n;
    if (in.kind == IN_CONNECT) {
        o.kind = SRC_BUF;
        o.idx = g.units[(size_t)in.src_unit].first_buf + c;
    } else if (in.kind == IN_PARAM) {
        o.kind = SRC_PARAM;
        o.idx = (int)in.vals[(size_t)c];
    } else {
        o.kind = SRC_CONST;
        o.cval = (float)in.vals[(size_t)c];  // setConstant stores into a Float32Array (Inlet.js:88-91)
    }
    return o;
}

inline bool expand(Program &P, std::string &err) {
    Graph &g = P.g;
    int nb = 0;
    for (auto &u : g.units) {
        u.first_buf = nb;
        nb += u.n_out;
    }
    P.n_bufs = nb;
    int64_t ring_pos = 0;
    for (auto &r : g.rings) {
        r.first_dev_ring = (int)P.dev_rings.size();
        for (int c = 0; c < r.nch; c++) {
            P.dev_rings.push_back({ring_pos, r.len});
            ring_pos += r.len;
        }
    }
    // One DevOp of unit `ui`, output channel `c`.  alloc: the settled program — state slots and rings are handed out as
    // the ops are made.  !alloc: a warm-up chunk — `op` starts as a copy of the settled op (same slots, same ring) and
    // only its operands are bound again, against the channel counts g.units[..].n_out holds at that moment.
    bool alloc = true;
    auto slot = [&](double init) {
        if (alloc) P.init_state.push_back(init);
        return 0;
    };
    auto own_ring = [&](DevOp &op, int64_t len) {
        if (!alloc) return;
        op.ring_len = len;
        op.ring_base = ring_pos;
        P.dev_rings.push_back({ring_pos, len});
        ring_pos += len;
    };
    auto make_op = [&](size_t ui, int c, DevOp &op) {
        UnitDesc &u = g.units[ui];
        switch (u.op) {
        case OP_OSC:
            op.attr = (int)u.attrs[0];
            op.in[0] = make_operand(g, u.inlets[0], 0);  // mono inlet: channel 0 (Piglet.js:56-61)
            slot(u.state[0]);
            break;
        case OP_RAMP:
            op.d[0] = u.attrs[0];
            op.d[1] = u.attrs[1];
            op.d[2] = u.attrs[2];
            slot(u.state[0]);
            slot(u.state[1] != 0 ? 1.0 : 0.0);
            break;
        case OP_MULTIPLY:
        case OP_SUM:
            op.in[0] = make_operand(g, u.inlets[0], c);
            op.in[1] = make_operand(g, u.inlets[1], c);
            break;
        case OP_FILTER: {
            op.attr = (int)u.attrs[0];
            op.in[0] = make_operand(g, u.inlets[0], c);
            op.in[1] = make_operand(g, u.inlets[1], 0);  // f is a mono inlet (Filter.js:9)
            const int have = (int)u.state[7];
            for (int k = 0; k < 7; k++) slot(u.state[(size_t)k]);  // has_lastF, lastF, a0, a1, a2, b1, b2
            for (int k = 0; k < 4; k++) slot(c < have ? u.state[(size_t)(8 + 4 * c + k)] : 0.0);  // x1 x2 y1 y2
            break;
        }
        case OP_DELAY: {
            op.in[0] = make_operand(g, u.inlets[0], c);
            op.in[1] = make_operand(g, u.inlets[1], c);
            own_ring(op, (int64_t)u.attrs[0]);
            slot(0.0);  // previous input sample (chunk engine's constant-delay path)
            break;
        }
        case OP_CB_READER:
        case OP_CB_WRITER: {
            const RingDesc &r = g.rings[(size_t)u.attrs[0]];
            const DevRing &dr = P.dev_rings[(size_t)(r.first_dev_ring + c)];
            op.ring_base = dr.base;
            op.ring_len = dr.len;
            op.attr = u.attrs[1] != 0 ? 1 : 0;  // postWipe / preWipe
            op.in[0] = make_operand(g, u.inlets[0], c);
            if (u.op == OP_CB_WRITER) {
                const int nin = u.inlets[1].n_channels(g.units);
                if (c < nin) op.in[1] = make_operand(g, u.inlets[1], c);  // `if(this.in[c])`: no modulo (Writer.js:19)
                else op.attr |= 2;                                          // nothing to mix on this channel
            }
            slot(u.state[0]);
            break;
        }
        case OP_REPEATER:
            op.in[0] = make_operand(g, u.inlets[0], c);
            break;
        case OP_SUBTRACT:  // `this.a[c] || zeroChunk`: a missing channel is silence, no modulo (Subtract.js:20-21)
            for (int k = 0; k < 2; k++) {
                if (c < u.inlets[(size_t)k].n_channels(g.units)) op.in[k] = make_operand(g, u.inlets[(size_t)k], c);
                else op.in[k] = DevOperand{SRC_CONST, 0, 0.f, 0};
            }
            break;
        case OP_DIVIDE: case OP_POW: case OP_CLIP: case OP_HARD_CLIP_ABOVE: case OP_HARD_CLIP_BELOW:
            op.in[0] = make_operand(g, u.inlets[0], c);
            op.in[1] = make_operand(g, u.inlets[1], c);  // modulo broadcast
            break;
        case OP_GAIN:
            op.in[0] = make_operand(g, u.inlets[0], c);
            op.in[1] = make_operand(g, u.inlets[1], 0);  // gain is a mono inlet (Gain.js:6)
            break;
        case OP_FIXED_MULTIPLY:
            op.in[0] = make_operand(g, u.inlets[0], 0);
            op.d[0] = u.attrs[0];
            break;
        case OP_SECONDS_TO_SAMPLES:
            op.in[0] = make_operand(g, u.inlets[0], c);
            op.d[0] = (double)g.sample_rate;
            break;
        case OP_POLARITY_INVERT: case OP_ABS: case OP_DECIBEL_TO_SCALER: case OP_SEMITONE_TO_RATIO:
            op.in[0] = make_operand(g, u.inlets[0], c);
            break;
        case OP_FIXED_DELAY: case OP_COMB_FILTER: case OP_ALL_PASS: case OP_MONO_DELAY: case OP_READBACK_DELAY: {
            const bool mono = u.op != OP_READBACK_DELAY;
            op.in[0] = make_operand(g, u.inlets[0], mono ? 0 : c);
            if (u.inlets.size() > 1) op.in[1] = make_operand(g, u.inlets[1], mono ? 0 : c);
            own_ring(op, (int64_t)u.attrs[0]);
            if (u.op != OP_MONO_DELAY) slot(u.state[0]);  // tBuffer (MonoDelay indexes with the circuit clock)
            break;
        }
        case OP_MULTI_OSC:
            op.attr = (int)u.attrs[0];
            op.in[0] = make_operand(g, u.inlets[0], c);
            slot(c < (int)u.state[0] ? u.state[(size_t)(1 + c)] : 0.0);  // `this.phase[c] = this.phase[c] || 0`
            break;
        case OP_PAN:  // mono inlets (Pan.js:6-7); one device op per output channel
            op.attr = c;
            op.n_in = 2;
            op.in[0] = make_operand(g, u.inlets[0], 0);
            op.in[1] = make_operand(g, u.inlets[1], 0);
            op.d[0] = u.attrs[0];
            break;
        case OP_MIDI_TO_FREQUENCY:
            op.n_in = 1;
            op.in[0] = make_operand(g, u.inlets[0], 0);
            break;
        case OP_RESCALE:  // every inlet but `in` broadcasts by modulo (Rescale.js:29-32)
            op.n_in = 5;
            for (int k = 0; k < 5; k++) op.in[k] = make_operand(g, u.inlets[(size_t)k], c);
            break;
        case OP_CROSS_FADER:  // `this.a[c] || zeroChannel` (CrossFader.js:23-24); dial is mono
            op.n_in = 3;
            for (int k = 0; k < 2; k++) {
                if (c < u.inlets[(size_t)k].n_channels(g.units)) op.in[k] = make_operand(g, u.inlets[(size_t)k], c);
                else op.in[k] = DevOperand{SRC_CONST, 0, 0.f, 0};
            }
            op.in[2] = make_operand(g, u.inlets[2], 0);
            break;
        case OP_VECTOR_MAGNITUDE:
            op.n_in = u.inlets[0].n_channels(g.units);
            for (int k = 0; k < op.n_in; k++) op.in[k] = make_operand(g, u.inlets[0], k);
            break;
        case OP_TIMER:
            op.d[0] = u.attrs[0];
            slot(u.state[0]);
            break;
        case OP_INPUT:
            op.attr = (int)u.attrs[0];
            break;
        case OP_RETRIGGER:  // (attr / pad / d[0] — the target's state slot, device op and kind — are filled in once every unit has its ops)
            op.n_in = 1;
            op.in[0] = make_operand(g, u.inlets[0], 0);
            slot(u.state[0]);
            break;
        case OP_SHAPE:  // mono inlets duration / min / max; attr: table id | left-is-shape << 8 | right-is-shape << 9
            op.n_in = 3;
            for (int k = 0; k < 3; k++) op.in[k] = make_operand(g, u.inlets[(size_t)k], 0);
            op.attr = (int)u.attrs[0] | (u.attrs[1] != 0 ? 256 : 0) | (u.attrs[3] != 0 ? 512 : 0);
            op.d[0] = u.attrs[2];  // leftEdge as a number
            op.d[1] = u.attrs[4];  // rightEdge as a number
            slot(u.state[0]);                    // t
            slot(u.state[1] != 0 ? 1.0 : 0.0);   // playing
            slot(u.state[2] != 0 ? 1.0 : 0.0);   // finished
            break;
        case OP_AHD:
            op.n_in = 3;
            for (int k = 0; k < 3; k++) op.in[k] = make_operand(g, u.inlets[(size_t)k], 0);
            op.d[0] = u.attrs[0];  // samplePeriod
            slot(u.state[0]);                    // state (0 off, 1 attack, 2 hold, 3 decay)
            slot(u.state[1] != 0 ? 1.0 : 0.0);   // playing
            slot(u.state[2]);                    // t
            break;
        case OP_SAMPLE_RATE_REDUX:  // every channel keeps its own copy of the (shared) counter next to its held value
            op.in[0] = make_operand(g, u.inlets[0], c);
            op.in[1] = make_operand(g, u.inlets[1], 0);  // ammount is mono (SampleRateRedux.js:6)
            slot(u.state[0]);
            slot(c < (int)u.state[1] ? u.state[(size_t)(2 + c)] : 0.0);  // unwritten output channels read 0
            break;
        case OP_CONCAT_CHANNELS: {  // a plain copy of one input channel (ConcatChannels.js:19-30)
            const int na = u.inlets[0].n_channels(g.units);
            op.op = OP_REPEATER;
            op.in[0] = c < na ? make_operand(g, u.inlets[0], c) : make_operand(g, u.inlets[1], c - na);
            break;
        }
        case OP_PICK_CHANNEL: {  // constant index (checked by parse): a copy of channel c % n (PickChannel.js:20)
            const int n = u.inlets[0].n_channels(g.units);
            op.op = OP_REPEATER;
            op.in[0] = make_operand(g, u.inlets[0], (int)std::fmod(u.inlets[1].vals[0], (double)n));
            break;
        }
        }
    };
    auto n_dev_ops = [&](const UnitDesc &u, int n_out) { return u.op == OP_CB_WRITER ? g.rings[(size_t)u.attrs[0]].nch : u.op == OP_RETRIGGER ? 1 : n_out; };
    for (size_t ui = 0; ui < g.units.size(); ui++) {
        UnitDesc &u = g.units[ui];
        u.first_op = (int)P.ops.size();
        u.first_slot = (int)P.init_state.size();
        for (auto &in : u.inlets)
            if (in.kind == IN_CONNECT && (size_t)in.src_unit >= ui) P.feed_forward = false;
        for (int c = 0; c < n_dev_ops(u, u.n_out); c++) {
            DevOp op{};
            op.op = u.op;
            op.unit = (int)ui;
            op.out_buf = u.n_out ? u.first_buf + c : -1;
            op.state_slot = (int)P.init_state.size();
            op.ring_base = op.ring_len = 0;
            make_op(ui, c, op);
            if (c == 0) u.slots_per_ch = (int)P.init_state.size() - u.first_slot;
            P.ops.push_back(op);
        }
    }
    P.ring_samples = ring_pos;
    // every ring is at most 1e9 samples (parse); all rings of one instance together: 2^31 samples = 8 GiB of f32
    if (ring_pos > ((int64_t)1 << 31)) return fail(err, "delay lines / CircleBuffers of one circuit instance add up to more than 2^31 samples (not supported on the GPU path)");
    if (P.ops.size() > ((size_t)1 << 22)) return fail(err, "more than 2^22 channel-expanded ops (not supported on the GPU path)");
    for (auto &op : P.ops)
        if (op.op == OP_RETRIGGER) {  // trigger(): Shape, Ramp -> t = 0, playing; AHD -> state = 1, playing (Shape/index.js:107-111, Ramp.js:19-23, AHD.js:24-28)
            const UnitDesc &target = g.units[(size_t)g.units[(size_t)op.unit].attrs[0]];
            if (target.op != OP_SHAPE && target.op != OP_AHD && target.op != OP_RAMP)
                return fail(err, "Retriggerer target is not a Shape / AHD / Ramp (not supported on the GPU path)");
            op.attr = target.first_slot;
            op.pad = target.first_op;
            op.d[0] = (double)target.op;
            op.d[1] = (double)g.sample_rate;
        }
    // Warm-up chunks: replay the channel growth of infer_channels, binding every op's operands against the counts the
    // reference's unit would see at that point of that chunk (units before it: this chunk's, units after it: last chunk's).
    if (!g.warm_counts.empty()) {
        alloc = false;
        std::vector<int> settled;
        for (auto &u : g.units) settled.push_back(u.n_out);
        for (auto &u : g.units) u.n_out = (u.op == OP_CB_WRITER || u.op == OP_HOST_ONLY || u.op == OP_RETRIGGER) ? 0 : 1;
        for (const auto &counts : g.warm_counts) {
            std::vector<DevOp> ops;
            for (size_t ui = 0; ui < g.units.size(); ui++) {
                UnitDesc &u = g.units[ui];
                for (int c = 0; c < n_dev_ops(u, counts[ui]); c++) {
                    DevOp op = P.ops[(size_t)(u.first_op + c)];
                    for (auto &o : op.in) o = DevOperand{};
                    // (no unit asks for its OWN count while its operands are bound: only its inputs' matter)
                    make_op(ui, c, op);
                    ops.push_back(op);
                }
                u.n_out = counts[ui];
            }
            P.warm_ops.push_back(std::move(ops));
        }
        for (size_t i = 0; i < g.units.size(); i++) g.units[i].n_out = settled[i];
    }
    const UnitDesc &ou = g.units[(size_t)g.out_unit];
    for (int c = 0; c < ou.n_out; c++) P.out_bufs.push_back(ou.first_buf + c);
    (void)err;
    return true;
}

inline bool compile(const double *d, size_t nw, Program &P, std::string &err, bool continuation = false) {
    if (!(parse(d, nw, P.g, err) && infer_channels(P.g, err) && expand(P, err))) return false;
    // A non-zero start clock means the circuit has been ticked before.  Unit state travels in the descriptor;
    // ring contents and the previous chunk behind a feedback edge do not, so a FRESH program can only start
    // such a circuit at clock 0 (dusp_program_continue carries them over on the device instead).
    if (!continuation && P.g.clock0 != 0 && (P.ring_samples != 0 || !P.feed_forward))
        return fail(err, "a circuit with delay lines or feedback that has already been ticked is not supported by "
                         "dusp_program_build (continue the program it was rendered with: dusp_program_continue)");
    return true;
}

}  // namespace dusp
