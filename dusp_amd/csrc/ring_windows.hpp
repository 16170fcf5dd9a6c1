// ring_windows.hpp — which slots of its delay rings can a render touch?  (host side; dusp_abi.hip zero_rings fills exactly these with zeros)
//
// Rings start as zeros (Delay.js:14, CircleBuffer.js:12).  A render that nothing continues only ever touches part of a long ring — a
// Delay of constant length d reads [clock, clock + n) and writes up to d + 1 further on; a CircleBuffer node with an unconnected
// offset walks n slots from where its own clock starts — and a default Delay line is five seconds per instance: zeroing all of it
// was most of such a render's time (reference patch "multitap" x 256: 236 GB of fill for 0.6 ms of kernel).  So only what can be
// touched is filled, with two chunks to spare for the kernels' look-ahead; anything else (a modulated or per-instance delay, a
// ring shorter than its window, the comb family's short rings) is its whole ring.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

#include "program.hpp"

namespace dusp {

struct RingWindow {
    int64_t at, count;  // slots [at, at + count) of the per-instance ring area (no window wraps: a wrapped one is two)
};

// n_chunks chunks from P.g.clock0 on.  covered: the windows' slots, counted with overlaps (the caller compares it with P.ring_samples).
inline void ring_windows(const Program &P, uint32_t n_chunks, std::vector<RingWindow> &wins, size_t &covered) {
    wins.clear();
    covered = 0;
    const double N = (double)n_chunks * kChunk, spare = 2.0 * kChunk + 4.0;
    auto add = [&](const DevOp &op, double from, double upto) {  // [from, upto) in the op's ring, wrapped
        const double len = (double)op.ring_len;
        from -= spare;
        upto += spare;
        if (!(upto - from < len)) {
            wins.push_back({op.ring_base, op.ring_len});
            covered += (size_t)op.ring_len;
            return;
        }
        double a = std::fmod(std::floor(from), len);
        if (a < 0) a += len;
        const int64_t at = (int64_t)a, count = (int64_t)(std::ceil(upto) - std::floor(from));
        const int64_t head = std::min(count, op.ring_len - at);
        wins.push_back({op.ring_base + at, head});
        if (count > head) wins.push_back({op.ring_base, count - head});
        covered += (size_t)count;
    };
    for (const DevOp &op : P.ops) {
        if (op.ring_len <= 0) continue;
        const double len = (double)op.ring_len, sr = (double)P.g.sample_rate;
        const double c0 = (double)(P.g.clock0 % op.ring_len);
        const bool k0 = op.in[0].kind == SRC_CONST, k1 = op.in[1].kind == SRC_CONST;
        const double T0 = (size_t)op.state_slot < P.init_state.size() ? P.init_state[(size_t)op.state_slot] : NAN;  // (the nodes' own clock)
        switch (op.op) {
        case OP_DELAY:
        case OP_MONO_DELAY: {
            const double d = (double)op.in[1].cval;
            if (k1 && d >= 0.0 && d < len) add(op, c0, c0 + N + d + 2.0);
            else add(op, 0.0, len);
            break;
        }
        case OP_CB_READER:
        case OP_CB_WRITER: {
            const double o = sr * (double)op.in[0].cval * (op.op == OP_CB_READER ? -1.0 : 1.0);
            if (k0 && std::isfinite(T0) && std::isfinite(o) && std::fabs(T0 + o) < 1e15) add(op, T0 + o, T0 + o + N + 1.0);
            else add(op, 0.0, len);
            break;
        }
        case OP_READBACK_DELAY: {
            const double d = (double)op.in[1].cval;
            if (k1 && std::isfinite(T0) && d >= 0.0 && d < len && std::fabs(T0) < 1e15) add(op, T0 - d, T0 + N + 1.0);
            else add(op, 0.0, len);
            break;
        }
        default: add(op, 0.0, len);
        }
    }
}

}  // namespace dusp
