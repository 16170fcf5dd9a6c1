// chunk_engine.hip — the universal ("chunk") engine.
//
// One lane per circuit INSTANCE, one wavefront per 64 instances, one persistent
// launch per render: every wave loops over all chunks, and inside a chunk over
// the device ops in the circuit's own process order, each op running its 256
// samples sequentially — i.e. the schedule of reference src/Circuit.js:19-41
// executed 64 instances at a time.  Because the schedule is the reference's,
// feedback edges (a unit reading a buffer whose producer ticks later gets the
// PREVIOUS chunk), Delay rings and shared CircleBuffers
// need no special handling.
//
// Memory layout (all instance-interleaved so that the 64 lanes of a wave touch
// 64 consecutive floats — one 256-byte coalesced access per wave instruction):
//   scratch [buffer][t][instance]   chunk buffers = SignalChunk channels (SignalChunk.js:5-8)
//   state   [slot][instance]  f64   Osc.phase, Ramp.t/playing, Filter x1..y2 + coefficients, node t
//   rings   [ring sample][instance] Delay / CircleBuffer rings
// PCM leaves through a 64x64 LDS transpose so that each store instruction
// writes 256 contiguous bytes of one instance's channel.
//
// Arithmetic follows the JS semantics exactly: f64 for every expression JS
// evaluates in Number, one f32 rounding per store into a chunk
// (compile with -ffp-contract=off).
#include <hip/hip_runtime.h>

#include "device_types.hpp"
#include "fused_device.hpp"
#include "map_ops.hpp"

namespace dusp {

__device__ __forceinline__ double or0(double v) { return (v != v || v == 0.0) ? 0.0 : v; }  // JS `v || 0`

struct Src {
    const float *ptr;
    size_t stride;
    float c;
    bool is_buf;
    __device__ __forceinline__ float at(int t) const { return is_buf ? ptr[(size_t)t * stride] : c; }
    // kBatch independent loads in flight: a lane's 256 samples are a serial recurrence for the ALU, but
    // their chunk-buffer reads need not wait for each other (one L2 round trip per batch, not per sample)
    __device__ __forceinline__ void load(int t0, float (&v)[16]) const {
        if (is_buf) {
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = ptr[(size_t)(t0 + k) * stride];
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = c;
        }
    }
};

constexpr int kBatch = 16;

__device__ __forceinline__ void store_batch(float *outp, size_t NP, int t0, const float (&r)[kBatch]) {
#pragma unroll
    for (int k = 0; k < kBatch; ++k) outp[(size_t)(t0 + k) * NP] = r[k];
}

__device__ __forceinline__ Src make_src(const DevOperand &o, const ChunkArgs &a, uint32_t i) {
    Src s;
    s.is_buf = o.kind == SRC_BUF;
    s.stride = a.n_pad;
    s.ptr = a.scratch + (size_t)(s.is_buf ? o.idx : 0) * kChunk * a.n_pad + i;
    s.c = o.cval;
    if (o.kind == SRC_PARAM) s.c = i < a.n_inst ? a.params[(size_t)o.idx * a.n_inst + i] : 0.f;
    return s;
}

// `phase += f; phase %= sr; if(phase < 0) phase += sr` (Osc.js:39-42).  fmod is exact, and for
// sr <= p < 2sr it equals p - sr (exact by Sterbenz), so only far-out values pay for the libcall.
__device__ __forceinline__ double osc_advance(double phase, double f, double sr) {
    double p = phase + f;
    if (fabs(p) >= sr) p = (p < 2.0 * sr && p > 0.0) ? p - sr : fmod(p, sr);
    if (p < 0.0) p += sr;
    return p;
}

// waveTable[floor(phase)]*(1-fraction) + waveTable[ceil(phase)]*fraction  (Osc.js:43-45).
// Table row has sr+2 entries (one pad) so idx+1 is always in range.
__device__ __forceinline__ float osc_lookup(const float *tbl, double phase, double sr) {
    if (!(phase >= 0.0 && phase <= sr)) return __builtin_nanf("");  // typed-array[NaN / OOB] is undefined
    const double lo = floor(phase);
    const double fraction = phase - lo;  // == phase % 1 for phase >= 0
    const int idx = (int)lo;
    const double a = (double)tbl[idx];
    const double b = (double)tbl[fraction != 0.0 ? idx + 1 : idx];  // ceil(phase)
    return (float)(a * (1.0 - fraction) + b * fraction);
}

// CircleBuffer.read/write/mix index (CircleBuffer.js:16-18): floor(t % len), then wrap negatives.
__device__ __forceinline__ int64_t ring_index(double t, double len) {
    double m = (t >= 0.0 && t < len) ? t : fmod(t, len);
    m = floor(m);
    if (m < 0.0) m += len;
    return (m >= 0.0 && m < len) ? (int64_t)m : -1;  // NaN -> dropped / undefined
}

__global__ void __launch_bounds__(64) dusp_chunk_kernel(ChunkArgs a) {
    __shared__ float tile[64 * 65];
    const uint32_t lane = threadIdx.x;
    const uint32_t i = blockIdx.x * 64 + lane;  // < n_pad by construction of the grid
    const size_t NP = a.n_pad;
    const double sr = (double)a.sample_rate;

    for (uint32_t ck = 0; ck < a.n_chunks; ++ck) {
        const int64_t clock = a.clock0 + (int64_t)ck * kChunk;
        // the first chunks of a circuit whose channel counts are still growing have op lists of their own
        const uint64_t chunk_no = (uint64_t)(clock / kChunk);
        const bool warm = chunk_no < a.n_warm;
        const DevOp *ops = warm ? a.ops + a.warm_first[chunk_no] : a.ops;
        const uint32_t n_ops = warm ? a.warm_n[chunk_no] : a.n_ops;
        for (uint32_t u = 0; u < n_ops; ++u) {
            const DevOp &op = ops[u];
            float *outp = a.scratch + (size_t)(op.out_buf >= 0 ? op.out_buf : 0) * kChunk * NP + i;
            double *st = a.state + (size_t)op.state_slot * NP + i;
            switch (op.op) {
            case OP_OSC: {  // Osc.js:35-47
                const Src f = make_src(op.in[0], a, i);
                const float *tbl = a.tables + (size_t)op.attr * a.table_stride;
                double phase = st[0];
                for (int t0 = 0; t0 < kChunk; t0 += kBatch) {
                    float fv[kBatch], r[kBatch];
                    double ph[kBatch];
                    f.load(t0, fv);
#pragma unroll
                    for (int k = 0; k < kBatch; ++k) ph[k] = phase = osc_advance(phase, (double)fv[k], sr);
#pragma unroll
                    for (int k = 0; k < kBatch; ++k) r[k] = osc_lookup(tbl, ph[k], sr);  // 2 x kBatch gathers in flight
                    store_batch(outp, NP, t0, r);
                }
                st[0] = phase;
                break;
            }
            case OP_RAMP: {  // Ramp.js:25-40
                const double duration = op.d[0], y0 = op.d[1], y1 = op.d[2];
                double tt = st[0];
                bool playing = st[NP] != 0.0;
                for (int t = 0; t < kChunk; ++t) {
                    if (playing) {
                        tt += 1.0;
                        if (tt > duration) { playing = false; tt = duration; }
                        if (tt < 0.0) { playing = false; tt = 0.0; }
                    }
                    outp[(size_t)t * NP] = (float)(y0 + (tt / duration) * (y1 - y0));
                }
                st[0] = tt;
                st[NP] = playing ? 1.0 : 0.0;
                break;
            }
            case OP_MULTIPLY:   // Multiply.js:23-34
            case OP_SUM:        // Sum.js:33-44
            case OP_REPEATER: { // Repeater.js:23-30
                const Src x = make_src(op.in[0], a, i), y = make_src(op.in[1], a, i);
                for (int t0 = 0; t0 < kChunk; t0 += kBatch) {
                    float xv[kBatch], yv[kBatch], r[kBatch];
                    x.load(t0, xv);
                    if (op.op != OP_REPEATER) y.load(t0, yv);
#pragma unroll
                    for (int k = 0; k < kBatch; ++k)
                        r[k] = op.op == OP_MULTIPLY ? xv[k] * yv[k] : op.op == OP_SUM ? xv[k] + yv[k] : xv[k];
                    store_batch(outp, NP, t0, r);
                }
                break;
            }
            case OP_FILTER: {  // Filter.js:27-51, coefficients :66-84
                const Src x = make_src(op.in[0], a, i), f = make_src(op.in[1], a, i);
                bool has_lastF = st[0] != 0.0;
                double lastF = st[NP], a0 = st[2 * NP], a1 = st[3 * NP], a2 = st[4 * NP], b1 = st[5 * NP],
                       b2 = st[6 * NP];
                double x1 = st[7 * NP], x2 = st[8 * NP], y1 = st[9 * NP], y2 = st[10 * NP];
                for (int t0 = 0; t0 < kChunk; t0 += kBatch) {
                    float xv[kBatch], fv[kBatch], r[kBatch];
                    x.load(t0, xv);
                    f.load(t0, fv);
#pragma unroll
                    for (int k = 0; k < kBatch; ++k) {
                        const double ft = (double)fv[k];
                        if (!has_lastF || ft != lastF) {
                            has_lastF = true;
                            lastF = ft;
                            double kf[5];
                            butterworth_coefficients(op.attr == 0 ? 0 : 1, ft, sr, kf);  // Filter.js:66-84 (filter_lamda.hpp: shared by every engine)
                            a0 = kf[0]; a1 = kf[1]; a2 = kf[2]; b1 = kf[3]; b2 = kf[4];
                        }
                        const double xin = (double)xv[k];
                        const float y = (float)(a0 * xin + a1 * or0(x1) + a2 * or0(x2) - b1 * or0(y1) - b2 * or0(y2));
                        r[k] = y;
                        y2 = or0(y1);
                        y1 = (double)y;
                        x2 = or0(x1);
                        x1 = xin;
                    }
                    store_batch(outp, NP, t0, r);
                }
                st[0] = has_lastF ? 1.0 : 0.0;
                st[NP] = lastF; st[2 * NP] = a0; st[3 * NP] = a1; st[4 * NP] = a2; st[5 * NP] = b1; st[6 * NP] = b2;
                st[7 * NP] = x1; st[8 * NP] = x2; st[9 * NP] = y1; st[10 * NP] = y2;
                break;
            }
            case OP_DELAY: {  // Delay.js:20-41
                const Src x = make_src(op.in[0], a, i), dl = make_src(op.in[1], a, i);
                float *ring = a.rings + (size_t)op.ring_base * NP + i;
                const int64_t len = op.ring_len;
                const double dlen = (double)len;
                int64_t tBuffer = clock % len;
                // Lane-constant delay (a constant or a per-instance parameter): every ring slot then receives
                // exactly the ceil tap of sample n-1 followed by the floor tap of sample n (slot 0: floor tap
                // only — the reference drops the ceil tap at index len), both onto the zero left by the read.
                // So the slot's final f32 value is computed in registers from x[n-1], x[n] with the reference's
                // two `+=` roundings and written ONCE: no read-modify-write chains through L2, and the ring
                // reads of a batch are independent of its writes when kBatch <= D <= len - kBatch.
                double dconst = (double)dl.c;
                if (dconst >= dlen) dconst = fmod(dconst, dlen);
                const double Dfl = floor(dconst);
                const bool fast_lane = !dl.is_buf && dconst >= (double)kBatch && Dfl <= dlen - (double)kBatch &&
                                       !(a.flags & kChunkFlagResumable);  // (leaves stale slots and a pending tap behind)
                if (__all(fast_lane)) {
                    const int64_t D = (int64_t)Dfl;
                    const double phi = dconst - Dfl;
                    double xprev = st[0];
                    for (int t0 = 0; t0 < kChunk; t0 += kBatch) {
                        float xv[kBatch], r[kBatch], w[kBatch];
                        x.load(t0, xv);
#pragma unroll
                        for (int k = 0; k < kBatch; ++k) {
                            int64_t s_ = tBuffer + k;
                            if (s_ >= len) s_ -= len;
                            r[k] = ring[(size_t)s_ * NP];
                        }
#pragma unroll
                        for (int k = 0; k < kBatch; ++k) {
                            int64_t s_ = tBuffer + k;
                            if (s_ >= len) s_ -= len;
                            int64_t lo = s_ + D;
                            if (lo >= len) lo -= len;
                            const double xin = (double)xv[k];
                            float slot;
                            if (phi != 0.0) {
                                slot = lo != 0 ? (float)(0.0 + xprev * phi) : 0.f;   // ceil tap of sample n-1 (dropped at slot 0)
                                slot = (float)((double)slot + xin * (1.0 - phi));    // floor tap of sample n
                            } else {
                                slot = (float)(0.0 + xin * 1.0);                     // floor(tWrite) == ceil(tWrite): both taps hit it
                                slot = (float)((double)slot + xin * 0.0);
                            }
                            w[k] = slot;
                            xprev = xin;
                            ring[(size_t)lo * NP] = w[k];
                        }
                        store_batch(outp, NP, t0, r);
                        tBuffer += kBatch;
                        if (tBuffer >= len) tBuffer -= len;
                    }
                    st[0] = xprev;
                    break;
                }
                for (int t = 0; t < kChunk; ++t) {
                    outp[(size_t)t * NP] = ring[(size_t)tBuffer * NP];
                    ring[(size_t)tBuffer * NP] = 0.f;
                    double tWrite = (double)tBuffer + (double)dl.at(t);
                    if (!(tWrite >= 0.0 && tWrite < dlen))
                        tWrite = (tWrite >= dlen && tWrite < 2.0 * dlen) ? tWrite - dlen : fmod(tWrite, dlen);
                    const double lo = floor(tWrite), hi = ceil(tWrite);
                    const double frac = tWrite - trunc(tWrite);  // tWrite % 1
                    const double xin = (double)x.at(t);
                    // indices outside [0, len) (and NaN) are silently dropped; hi == len does NOT wrap
                    if (lo >= 0.0 && lo < dlen) {
                        float *p = ring + (size_t)(int64_t)lo * NP;
                        *p = (float)((double)*p + xin * (1.0 - frac));
                    }
                    if (hi >= 0.0 && hi < dlen) {
                        float *p = ring + (size_t)(int64_t)hi * NP;
                        *p = (float)((double)*p + xin * frac);
                    }
                    if (++tBuffer == len) tBuffer = 0;
                }
                st[0] = (double)x.at(kChunk - 1);  // (the last input: what a kernel that takes the render over — write-once ring — starts from)
                break;
            }
            case OP_CB_READER: {  // CircleBufferReader.js:12-25
                const Src off = make_src(op.in[0], a, i);
                float *ring = a.rings + (size_t)op.ring_base * NP + i;
                const double dlen = (double)op.ring_len;
                const double T = st[0];
                for (int t = 0; t < kChunk; ++t) {
                    const double tRead = T + (double)t - sr * (double)off.at(t);
                    const int64_t idx = ring_index(tRead, dlen);
                    outp[(size_t)t * NP] = idx >= 0 ? ring[(size_t)idx * NP] : __builtin_nanf("");
                    if ((op.attr & 1) && idx >= 0) ring[(size_t)idx * NP] = 0.f;  // postWipe
                }
                st[0] = T + (double)kChunk;
                break;
            }
            case OP_CB_WRITER: {  // CircleBufferWriter.js:12-25
                const Src off = make_src(op.in[0], a, i), x = make_src(op.in[1], a, i);
                float *ring = a.rings + (size_t)op.ring_base * NP + i;
                const double dlen = (double)op.ring_len;
                const double T = st[0];
                for (int t = 0; t < kChunk; ++t) {
                    const double tWrite = T + (double)t + sr * (double)off.at(t);
                    const int64_t idx = ring_index(tWrite, dlen);
                    if (idx < 0) continue;
                    float *p = ring + (size_t)idx * NP;
                    if (op.attr & 1) *p = 0.f;              // preWipe
                    if (!(op.attr & 2)) *p = *p + x.at(t);  // mix
                }
                st[0] = T + (double)kChunk;
                break;
            }
            case OP_FIXED_DELAY:   // FixedDelay.js:13-19
            case OP_COMB_FILTER:   // CombFilter.js:11-17
            case OP_ALL_PASS: {    // AllPass.js:8-15
                const Src x = make_src(op.in[0], a, i), fb = make_src(op.in[1], a, i);
                float *ring = a.rings + (size_t)op.ring_base * NP + i;
                const int64_t len = op.ring_len;
                int64_t tB = (int64_t)st[0];
                for (int t = 0; t < kChunk; ++t) {
                    if (++tB >= len) tB = 0;  // (tBuffer + 1) % length
                    float *slot = ring + (size_t)tB * NP;
                    const float xin = x.at(t), delayOut = *slot;
                    if (op.op == OP_FIXED_DELAY) {
                        outp[(size_t)t * NP] = delayOut;
                        *slot = xin;
                    } else if (op.op == OP_COMB_FILTER) {
                        outp[(size_t)t * NP] = delayOut;
                        *slot = (float)((double)xin + (double)delayOut * (double)fb.at(t));
                    } else {
                        const double g = (double)fb.at(t);
                        *slot = (float)((double)xin + (double)delayOut * g);
                        outp[(size_t)t * NP] = (float)((double)delayOut - (double)xin * g);
                    }
                }
                st[0] = (double)tB;
                break;
            }
            case OP_MONO_DELAY: {  // MonoDelay.js:16-30: write first (ceil tap wraps), then read and clear
                const Src x = make_src(op.in[0], a, i), dl = make_src(op.in[1], a, i);
                float *ring = a.rings + (size_t)op.ring_base * NP + i;
                const int64_t len = op.ring_len;
                const double dlen = (double)len;
                int64_t tBuffer = clock % len;
                for (int t = 0; t < kChunk; ++t) {
                    double tWrite = (double)tBuffer + (double)dl.at(t);
                    if (!(tWrite >= 0.0 && tWrite < dlen))
                        tWrite = (tWrite >= dlen && tWrite < 2.0 * dlen) ? tWrite - dlen : fmod(tWrite, dlen);
                    const double lo = floor(tWrite), frac = tWrite - trunc(tWrite);
                    double hi = ceil(tWrite);
                    if (hi >= dlen) hi -= dlen;  // Math.ceil(tWrite) % length
                    const double xin = (double)x.at(t);
                    if (lo >= 0.0 && lo < dlen) {
                        float *p = ring + (size_t)(int64_t)lo * NP;
                        *p = (float)((double)*p + xin * (1.0 - frac));
                    }
                    if (hi >= 0.0 && hi < dlen) {
                        float *p = ring + (size_t)(int64_t)hi * NP;
                        *p = (float)((double)*p + xin * frac);
                    }
                    outp[(size_t)t * NP] = ring[(size_t)tBuffer * NP];
                    ring[(size_t)tBuffer * NP] = 0.f;
                    if (++tBuffer == len) tBuffer = 0;
                }
                break;
            }
            case OP_READBACK_DELAY: {  // ReadBackDelay.js:24-44
                const Src x = make_src(op.in[0], a, i), dl = make_src(op.in[1], a, i);
                float *ring = a.rings + (size_t)op.ring_base * NP + i;
                const double dlen = (double)op.ring_len;
                const double T0 = st[0];
                int64_t w = (int64_t)fmod(T0, dlen);
                for (int t = 0; t < kChunk; ++t) {
                    ring[(size_t)w * NP] = x.at(t);
                    double r = (T0 + (double)t) - (double)dl.at(t) + dlen;
                    r = (r >= 0.0 && r < dlen) ? r : fmod(r, dlen);
                    // a fractional or negative index reads `undefined` -> NaN in the reference
                    outp[(size_t)t * NP] = (r >= 0.0 && r < dlen && r == floor(r)) ? ring[(size_t)(int64_t)r * NP] : __builtin_nanf("");
                    if (++w >= op.ring_len) w = 0;
                }
                st[0] = T0 + (double)kChunk;
                break;
            }
            case OP_MULTI_OSC: {  // MultiChannelOsc.js:21-38: no `phase < 0` fix-up, so negative phases read NaN
                const Src f = make_src(op.in[0], a, i);
                const float *tbl = a.tables + (size_t)op.attr * a.table_stride;
                double phase = or0(st[0]);
                for (int t0 = 0; t0 < kChunk; t0 += kBatch) {
                    float fv[kBatch], r[kBatch];
                    double ph[kBatch];
                    f.load(t0, fv);
#pragma unroll
                    for (int k = 0; k < kBatch; ++k) {
                        double p = phase + (double)fv[k];
                        if (fabs(p) >= sr) p = fmod(p, sr);
                        ph[k] = phase = p;
                    }
#pragma unroll
                    for (int k = 0; k < kBatch; ++k) r[k] = osc_lookup(tbl, ph[k], sr);
                    store_batch(outp, NP, t0, r);
                }
                st[0] = phase;
                break;
            }
            case OP_PAN: case OP_MIDI_TO_FREQUENCY: case OP_RESCALE: case OP_CROSS_FADER: case OP_VECTOR_MAGNITUDE: {
                Src src[kMaxIn];
#pragma unroll
                for (int k = 0; k < kMaxIn; ++k) src[k] = make_src(op.in[k < op.n_in ? k : 0], a, i);
                for (int t0 = 0; t0 < kChunk; t0 += kBatch) {
                    float v[kMaxIn][kBatch], r[kBatch];
#pragma unroll
                    for (int k = 0; k < kMaxIn; ++k)
                        if (k < op.n_in) src[k].load(t0, v[k]);
#pragma unroll
                    for (int k = 0; k < kBatch; ++k) {
                        const float w[kMaxIn] = {v[0][k], v[1][k], v[2][k], v[3][k], v[4][k]};
                        r[k] = map_wide(op.op, op.attr, op.n_in, w, op.d[0]);
                    }
                    store_batch(outp, NP, t0, r);
                }
                break;
            }
            case OP_SHAPE: {  // Shape/index.js:28-59
                const Src duration = make_src(op.in[0], a, i), mn = make_src(op.in[1], a, i), mx = make_src(op.in[2], a, i);
                const float *data = a.tables + (size_t)(op.attr & 255) * a.table_stride;
                const bool left_shape = (op.attr & 256) != 0, right_shape = (op.attr & 512) != 0;
                const double left = left_shape ? (double)data[0] : op.d[0];
                const double right = right_shape ? (double)data[a.sample_rate] : op.d[1];
                double tt = st[0];
                const bool playing = st[NP] != 0.0;
                bool finished = st[2 * NP] != 0.0;
                for (int t0 = 0; t0 < kChunk; t0 += kBatch) {
                    float dv[kBatch], lo[kBatch], hi[kBatch], r[kBatch];
                    duration.load(t0, dv);
                    mn.load(t0, lo);
                    mx.load(t0, hi);
#pragma unroll
                    for (int k = 0; k < kBatch; ++k) {
                        const double l = (double)lo[k], h = (double)hi[k];
                        if (playing) tt += 1.0 / (double)dv[k];
                        if (tt <= 0.0) {
                            r[k] = (float)(left * (h - l) + l);
                        } else if (tt > sr) {
                            finished = true;
                            r[k] = (float)(right * (h - l) + l);
                        } else if (tt == tt) {
                            const double fl = floor(tt), frac = tt - fl;  // t % 1 for t > 0
                            const double hi_tap = (double)data[(int)ceil(tt)], lo_tap = (double)data[(int)fl];
                            r[k] = (float)(l + (h - l) * (hi_tap * frac + lo_tap * (1.0 - frac)));
                        } else
                            r[k] = __builtin_nanf("");  // t is NaN: table[NaN] is undefined
                    }
                    store_batch(outp, NP, t0, r);
                }
                st[0] = tt;
                st[2 * NP] = finished ? 1.0 : 0.0;
                break;
            }
            case OP_AHD: {  // AHD.js:35-76
                const Src attack = make_src(op.in[0], a, i), hold = make_src(op.in[1], a, i), decay = make_src(op.in[2], a, i);
                const double period = op.d[0];
                int stage = (int)st[0];
                bool playing = st[NP] != 0.0;
                double tt = st[2 * NP];
                for (int t = 0; t < kChunk; ++t) {
                    float y = outp[(size_t)t * NP];  // an unknown `state` leaves the sample as it was
                    if (stage == 1) {
                        y = (float)tt;
                        if (playing) {
                            tt += period / (double)attack.at(t);
                            if (tt >= 1.0) { ++stage; tt -= 1.0; }
                        }
                    } else if (stage == 2) {
                        y = 1.f;
                        if (playing) {
                            tt += period / (double)hold.at(t);
                            if (tt >= 1.0) { ++stage; tt -= 1.0; }
                        }
                    } else if (stage == 3) {
                        y = (float)(1.0 - tt);
                        if (playing) {
                            tt += period / (double)decay.at(t);
                            if (tt >= 1.0) { stage = 0; playing = false; }  // stop()
                        }
                    } else if (stage == 0)
                        y = 0.f;
                    outp[(size_t)t * NP] = y;
                }
                st[0] = (double)stage;
                st[NP] = playing ? 1.0 : 0.0;
                st[2 * NP] = tt;
                break;
            }
            case OP_RETRIGGER: {  // Retriggerer.js:13-24: an accumulator of `rate`; every crossing of sampleRate triggers the target
                const Src rate = make_src(op.in[0], a, i);
                double T = st[0];
                bool fired = false;
                for (int t = 0; t < kChunk; ++t) {
                    T += (double)rate.at(t);
                    if (T >= sr) { fired = true; T -= sr; }
                }
                st[0] = T;
                if (fired) {  // the target ticks later in this chunk (the Retriggerer is chained before it)
                    double *ts = a.state + (size_t)op.attr * NP + i;
                    if ((int)op.d[0] == OP_AHD) { ts[0] = 1.0; ts[NP] = 1.0; }   // state = attack, playing
                    else { ts[0] = 0.0; ts[NP] = 1.0; }                          // Shape, Ramp: t = 0, playing
                }
                break;
            }
            case OP_INPUT: {  // a signal the host computed (Noise.js:16-27 draws Math.random() per sample): stream op.attr
                const float *src = a.inputs + ((size_t)op.attr * a.n_inst + (i < a.n_inst ? i : a.n_inst - 1)) * a.n_samples;
                const uint64_t t0 = (uint64_t)ck * kChunk;
                for (int t = 0; t < kChunk; ++t) outp[(size_t)t * NP] = t0 + t < a.n_samples ? src[t0 + t] : 0.f;  // (past the end: never copied out)
                break;
            }
            case OP_TIMER: {  // Timer.js:36-41: a running f64 sum of 1/sampleRate, rounded to f32 per sample
                double tt = st[0];
                const double period = op.d[0];
                for (int t = 0; t < kChunk; ++t) {
                    tt += period;
                    outp[(size_t)t * NP] = (float)tt;
                }
                st[0] = tt;
                break;
            }
            case OP_SAMPLE_RATE_REDUX: {  // SampleRateRedux.js:21-38
                const Src x = make_src(op.in[0], a, i), amount = make_src(op.in[1], a, i);
                double since = st[0];
                float held = (float)st[NP];
                for (int t0 = 0; t0 < kChunk; t0 += kBatch) {
                    float xv[kBatch], av[kBatch], r[kBatch];
                    x.load(t0, xv);
                    amount.load(t0, av);
#pragma unroll
                    for (int k = 0; k < kBatch; ++k) {
                        since += 1.0;
                        if (since > (double)av[k]) {
                            held = xv[k];
                            since = 0.0;
                        }
                        r[k] = held;
                    }
                    store_batch(outp, NP, t0, r);
                }
                st[0] = since;
                st[NP] = (double)held;
                break;
            }
            default: {  // stateless elementwise maps (map_ops.hpp)
                const Src x = make_src(op.in[0], a, i), y = make_src(op.in[1], a, i);
                for (int t0 = 0; t0 < kChunk; t0 += kBatch) {
                    float xv[kBatch], yv[kBatch], r[kBatch];
                    x.load(t0, xv);
                    y.load(t0, yv);
#pragma unroll
                    for (int k = 0; k < kBatch; ++k) r[k] = map_apply(op.op, xv[k], yv[k], op.d[0]);
                    store_batch(outp, NP, t0, r);
                }
                break;
            }
            }
        }

        // copy-out (renderChannelData.js:35-44): `|| 0` maps NaN and -0 to +0; samples past
        // n_samples in the last chunk are dropped.
        for (uint32_t oc = 0; oc < a.n_out; ++oc) {
            const float *src = a.scratch + (size_t)a.out_bufs[oc] * kChunk * NP + i;
            for (int tb = 0; tb < kChunk / 64; ++tb) {
                for (int tt = 0; tt < 64; ++tt) {
                    float v = src[(size_t)(tb * 64 + tt) * NP];
                    v = (v != v) ? 0.f : v + 0.f;
                    tile[tt * 65 + lane] = v;
                }
                __syncthreads();
                const uint64_t t = (uint64_t)ck * kChunk + (uint64_t)tb * 64 + lane;
                if (t < a.n_samples) {
                    for (int r = 0; r < 64; ++r) {
                        const uint32_t inst = blockIdx.x * 64 + r;
                        if (inst >= a.n_inst) break;
                        a.out[((size_t)inst * a.n_out + oc) * a.n_samples + t] = tile[lane * 65 + r];
                    }
                }
                __syncthreads();
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
hipError_t launch_chunk_engine(const ChunkArgs &a, hipStream_t stream) {
    const uint32_t blocks = a.n_pad / 64;
    hipLaunchKernelGGL(dusp_chunk_kernel, dim3(blocks), dim3(64), 0, stream, a);
    return hipGetLastError();
}

// Broadcast the per-slot initial state to every instance column.
__global__ void dusp_state_init_kernel(double *state, const double *init, uint32_t n_slots, uint32_t n_pad) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (size_t)n_slots * n_pad) state[idx] = init[idx / n_pad];
}

hipError_t launch_state_init(double *state, const double *init, uint32_t n_slots, uint32_t n_pad, hipStream_t stream) {
    const size_t n = (size_t)n_slots * n_pad;
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(dusp_state_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, state, init,
                       n_slots, n_pad);
    return hipGetLastError();
}

// Fill kernel: the measured HBM write ceiling (16 B per lane, fully coalesced, nothing else).
// Plain stores on a 16384-block grid were the fastest of the patterns tried (tools/fillbench.hip).
__global__ void __launch_bounds__(256) dusp_fill_kernel(f32x4 *out, size_t n4, float value) {
    const f32x4 v = {value, value, value, value};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        out[i] = v;
}

hipError_t launch_fill(float *out, size_t n_floats, float value, hipStream_t stream) {
    const size_t n4 = n_floats / 4;
    if (n4) hipLaunchKernelGGL(dusp_fill_kernel, dim3(16384), dim3(256), 0, stream, (f32x4 *)out, n4, value);
    return hipGetLastError();
}

}  // namespace dusp
