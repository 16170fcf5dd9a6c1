// wave_engine.hip — the "wave" engine: circuits of Osc / Ramp / Multiply / Sum / Repeater / the elementwise
// maps (arbitrary fan-out, connected oscillator frequencies = FM) and, since v2, Filter and constant-delay
// Delay units, with or without feedback edges: one WAVEFRONT per circuit instance, one LANE per SAMPLE.
//
// This is the north star's kernel shape in its general form: the wave walks the circuit's units in
// the reference's process order (Circuit.js:34-37) once per 256-sample chunk, each unit reading and
// writing its per-unit chunk buffers (SignalChunk.js:5-8) in LDS as one float4 per lane (lane l owns
// samples 4l..4l+3), and the rendered outlet leaves as one coalesced 1 KiB store per chunk.
//
// Oscillators use WAVEFRONT-WIDE PHASE ACCUMULATION: the per-sample increments f[t] (a constant, a
// per-instance parameter or another unit's chunk) are converted to exact 2^-36 fixed point — every f32
// with |f| >= 2^-13 is a multiple of 2^-36, the same bound below which the reference's own f64
// accumulation starts to round (SURVEY.md §8a note ii) — prefix-summed inside the lane, scanned across
// the 64 lanes with integer adds (6 shuffle steps; sums stay below 2^61 so no step needs a modulo) and
// reduced mod sampleRate once per sample with exact integer arithmetic.  That reproduces
// `phase += f[t]; phase %= sr; if (phase < 0) phase += sr` (Osc.js:39-42) bit for bit in the exact
// regime, including negative and mixed-sign FM increments.  The only state an oscillator carries from
// chunk to chunk is the phase of its last sample (one u64 per wave in LDS) — plus a poison flag:
// once f is NaN/Inf the reference's phase is NaN for good.
// Ramp is evaluated in closed form from the sample index (Ramp.js:25-40).
//
// Filter (Filter.js:27-51) splits like loop2_engine.hip: the feed-forward half (a0 x + a1 x1) + a2 x2 is
// computed per lane (neighbour samples by DPP shuffle, coefficients per sample when the cutoff is modulated),
// the output recurrence y = f32((P - b1 y1) - b2 y2) runs serially over the chunk's 256 samples out of LDS.
// Delay (Delay.js:20-41) with a constant delay of at least one chunk reads and writes its ring
// ([instance][slot] in HBM) fully in parallel.  Chunk buffers persist in LDS from chunk to chunk, so a unit
// that reads a buffer whose producer ticks later sees the previous chunk — the reference's feedback semantics.
//
// Time is sequential per instance (the scan carry), so parallelism = instances: thousands of voices
// fill the chip; a single circuit (BASELINE configs[1]) runs on one wave and is latency-bound
// (~0.7 us per chunk) — still two orders of magnitude faster than ticking it on the host.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "device_types.hpp"
#include "fused_device.hpp"
#include "repeat_add.hpp"
#include "map_ops.hpp"

namespace dusp {

namespace {

constexpr double kTwo36 = 68719476736.0;
constexpr int kFracBits = 36;
constexpr int kOpState = 12;  // doubles of LDS state per op per wave

// Barrier that orders LDS traffic only: __syncthreads() also drains the vector-memory queue, i.e. waits for every PCM /
// ring store in flight to be acknowledged.  Inside the chunk loop no wave reads GLOBAL data another wave wrote (an
// instance — its ring rows, its PCM rows — belongs to one wave), so only the LDS tiles need the fence.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ double or0w(double v) { return (v != v || v == 0.0) ? 0.0 : v; }  // JS `v || 0`

// Butterworth coefficients of Filter.js:66-84 (kind 0 = LP, 1 = HP): filter_lamda.hpp, shared by every engine
__device__ __forceinline__ void filter_coefficients(int kind, double f, double sr, double (&k)[5]) { butterworth_coefficients(kind, f, sr, k); }

struct V4 {
    float v[4];
};

// Delay with a constant delay of at least a chunk (and a chunk short of the ring's length): reads and writes of one chunk
// never meet, every slot receives its two taps from neighbouring samples — computed in registers, written once.
__device__ __forceinline__ bool delay_is_write_once(const DevOp &op) {
    if (op.in[1].kind != SRC_CONST) return false;
    const double len = (double)op.ring_len;
    double dconst = (double)op.in[1].cval;
    if (dconst >= len) dconst = fmod(dconst, len);
    return floor(dconst) >= (double)kChunk && floor(dconst) + (double)kChunk <= len;
}

__device__ __forceinline__ V4 load_operand(const DevOperand &o, const f32x4 *bufs, uint32_t lane, const float *params,
                                           uint32_t n_inst, uint32_t inst) {
    V4 r;
    if (o.kind == SRC_BUF) {
        const f32x4 x = bufs[(size_t)o.idx * 64 + lane];
        r.v[0] = x[0]; r.v[1] = x[1]; r.v[2] = x[2]; r.v[3] = x[3];
    } else {
        const float c = o.kind == SRC_PARAM ? params[(size_t)o.idx * n_inst + inst] : o.cval;
        r.v[0] = r.v[1] = r.v[2] = r.v[3] = c;
    }
    return r;
}

__device__ __forceinline__ long long wave_inclusive_scan(long long x, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const long long y = __shfl_up(x, d, 64);
        if ((int)lane >= d) x += y;
    }
    return x;
}

}  // namespace

// TBL: half wave table in LDS.  WAVES: wavefronts (instances) per workgroup.  RING: carries the ordered slot operations of
// short / signal-rate delay lines — a separate variant, so that programs without them keep the leaner kernel.
// Which units a variant carries — always: Osc, Ramp, Multiply, Sum, Repeater, the two-operand maps; FILT: Filter and the
// write-once Delay; EXT (2): everything else — so that the common graphs keep kernels whose register allocation the other
// units' code does not disturb.
template <int TBL, int WAVES, int RING, int EXT, int FILT>
__global__ void __launch_bounds__(WAVES * 64) dusp_wave_kernel(WaveArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int BLOCK = WAVES * 64;
    Table<1> lds_table;
    lds_table.h = lds;
    lds_table.N = A.sample_rate + 1;
    lds_table.M = A.sample_rate / 2;
    if (TBL == 1) load_half_table<BLOCK>(lds, A.tables + (size_t)A.lds_table_id * A.table_stride, A.sample_rate);

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    // a short last workgroup keeps its surplus waves alive (they take part in the Filter stage's barriers):
    // they shadow the last real instance but never store to HBM
    // time-split mode: a "virtual instance" is one segment of an instance (n_seg == 1: the instance itself)
    const uint32_t n_virtual = A.n_inst * A.n_seg;
    const bool live = blockIdx.x * WAVES + wave < n_virtual;
    const uint32_t vinst = live ? blockIdx.x * WAVES + wave : n_virtual - 1;
    const uint32_t inst = vinst / A.n_seg, seg = vinst - inst * A.n_seg;
    const uint32_t g_begin = seg * A.seg_groups;
    const uint32_t g_end = A.n_seg == 1 ? A.n_groups : (g_begin + A.seg_groups < A.n_groups ? g_begin + A.seg_groups : A.n_groups);
    const bool rendering = A.pass_mode == 0;
    char *mine = (char *)lds + A.table_bytes + (size_t)wave * A.wave_bytes;
    // workgroup-shared hand-over tiles of the cooperative Filter stage: Pt[WAVES][258] f64, Yt[WAVES][260] f32
    double *Pt = (double *)((char *)lds + A.table_bytes + (size_t)WAVES * A.wave_bytes);
    float *Yt = (float *)(Pt + WAVES * 258);
    f32x4 *bufs = (f32x4 *)mine;                                            // [n_bufs][64] float4 = chunk buffers
    double *opstate = (double *)(mine + (size_t)A.n_bufs * 1024);           // [n_ops][kOpState] per-unit state
    double *scratch = opstate + (size_t)A.n_state_ops * kOpState;                 // Filter: P[256], b1[256], b2[256]
    // This instance's parameter column, fetched once: an operand that is a per-instance parameter would otherwise cost an
    // HBM round trip per unit and chunk, which a wavefront has nothing to hide behind.
    const float *pvals = A.params ? A.params + inst : nullptr;
    uint32_t pstride = A.n_inst;
    if (A.param_bytes) {
        float *mine_p = (float *)(mine + A.wave_bytes - A.param_bytes);
        for (uint32_t k = lane; k < A.n_params; k += 64) mine_p[k] = A.params[(size_t)k * A.n_inst + inst];
        pvals = mine_p;
        pstride = 1;
    }
    // Osc: [0] phase carry (u64 bits, 2^-36 units), [1] poison flag.  Delay: [0] previous input sample.
    // Filter: [0] has_lastF [1] lastF [2..6] a0 a1 a2 b1 b2 [7..10] x1 x2 y1 y2

    const uint32_t sr = A.sample_rate;
    const double srd = (double)sr;
    const unsigned long long S = (unsigned long long)sr << kFracBits;
    const double inv_S = 1.0 / (double)S;
    // multiple of S that makes any |x| < 2^62 non-negative before the modulo
    const unsigned long long lift = S * ((1ull << 62) / S);

    if (A.resume)  // a continued render: every outlet's last chunk comes back (feedback edges read it)
        for (uint32_t b = 0; b < A.n_bufs; ++b)
            bufs[(size_t)b * 64 + lane] = ((const f32x4 *)(A.saved_bufs + ((size_t)inst * A.n_bufs + b) * kChunk))[lane];
    else
        for (uint32_t b = 0; b < A.n_bufs; ++b) bufs[(size_t)b * 64 + lane] = f32x4{0.f, 0.f, 0.f, 0.f};  // outlets start as zeros
    if (lane == 0)
        for (uint32_t u = 0; u < A.n_ops; ++u) {
            const DevOp &op = A.ops[u];
            if (op.lds_slot < 0) continue;  // stateless
            double *os = opstate + (size_t)op.lds_slot * kOpState;
            for (int k = 0; k < kOpState; ++k) os[k] = 0.0;
            if (op.op == OP_OSC) {
                if (A.n_seg == 1) *(unsigned long long *)os = (unsigned long long)(A.init_state[op.state_slot] * kTwo36);
                else if (rendering || (uint32_t)op.d[0] < A.pass_level) {  // start phase known from the earlier passes
                    const unsigned long long v = A.seg_start[((size_t)u * A.n_inst + inst) * A.n_seg + seg];
                    *(unsigned long long *)os = v & ~(1ull << 63);
                    ((uint32_t *)(os + 1))[0] = (uint32_t)(v >> 63);
                }  // else: accumulate from 0 — the segment's phase total
            }
            if (op.op == OP_DELAY || op.op == OP_TIMER || op.op == OP_FIXED_DELAY || op.op == OP_COMB_FILTER || op.op == OP_ALL_PASS ||
                op.op == OP_CB_READER || op.op == OP_CB_WRITER || op.op == OP_MULTI_OSC || op.op == OP_READBACK_DELAY || op.op == OP_RETRIGGER)
                os[0] = A.init_state[op.state_slot];
            // Delay's carried input sample is engine-internal (no descriptor carries it): a continued render takes it
            // from where the previous launch left it
            if (op.op == OP_DELAY && A.resume) os[0] = A.state[(size_t)op.state_slot * A.n_pad + inst];
            if (op.op == OP_SHAPE || op.op == OP_AHD)
                for (int k = 0; k < 3; ++k) os[k] = A.init_state[op.state_slot + k];
            if (op.op == OP_RAMP) { os[0] = A.init_state[op.state_slot]; os[1] = 0.0; os[2] = A.init_state[op.state_slot + 1]; }
            // time-split: a later segment starts from the running sum's value at its first sample (plan_wave admits only
            // Timers and constant-duration Shapes whose sums repeat_add covers)
            if (A.n_seg > 1 && g_begin > 0) {
                if (op.op == OP_TIMER) os[0] = repeat_add(os[0], op.d[0], (uint64_t)g_begin * kChunk);
                if (op.op == OP_SHAPE && os[1] != 0.0) os[0] = repeat_add(os[0], 1.0 / (double)op.in[0].cval, (uint64_t)g_begin * kChunk);
            }
            if (op.op == OP_SAMPLE_RATE_REDUX)
                for (int k = 0; k < 2; ++k) os[k] = A.init_state[op.state_slot + k];
            if (op.op == OP_FILTER)
                for (int k = 0; k < 11; ++k) os[k] = A.init_state[op.state_slot + k];
        }
    __syncthreads();


    // ---- Ordered slot operations (RING variant): ring units whose accesses can land anywhere — a Delay with a signal-rate
    // or sub-chunk delay (Delay.js:26-40), MonoDelay (MonoDelay.js:16-30), ReadBackDelay (ReadBackDelay.js:24-44),
    // CircleBuffer nodes with a signal-rate offset or a ring shorter than a chunk (CircleBufferReader.js:12-25,
    // CircleBufferWriter.js:12-25).  The reference walks the chunk sample by sample — read (and clear) one slot, add a tap
    // to each of two others, ... — in f32, so the ORDER of the operations on one slot matters, while operations on
    // different slots commute.  Each lane owns four samples = up to twelve slot operations, keyed 3 t + j in the
    // reference's order.  Rounds: every pending operation bids for its slot with its key (ds_min_u32 on a 1024-entry
    // table indexed by slot mod 1024: two slots sharing an entry only cost extra rounds), the lowest key of each entry
    // performs its operation on the ring in HBM, and so on until nothing is pending — three rounds for a steady delay;
    // whatever the modulation does, the result is the reference's.
    enum : int { RO_NONE = 0, RO_READ, RO_READ_CLEAR, RO_ADD, RO_STORE };
    auto ordered_ring_ops = [&](const DevOp &op, uint32_t u, uint32_t g, V4 &out) {
        double *ss = opstate + (size_t)op.lds_slot * kOpState;
        const bool is_delay = op.op == OP_DELAY, is_mono = op.op == OP_MONO_DELAY, is_readback = op.op == OP_READBACK_DELAY;
        const bool is_reader = op.op == OP_CB_READER, is_writer = op.op == OP_CB_WRITER;
        const bool writer_mixes = is_writer && !(op.attr & 2);
        // operand 0: the signal (delay lines) or the offset (CircleBuffer nodes); operand 1: the delay / the writer's input
        const V4 p0 = load_operand(op.in[0], bufs, lane, pvals, pstride, 0);
        V4 p1 = {{0.f, 0.f, 0.f, 0.f}};
        if (!is_reader && !(is_writer && !writer_mixes)) p1 = load_operand(op.in[1], bufs, lane, pvals, pstride, 0);
        const uint32_t len = (uint32_t)op.ring_len;
        const double dlen = (double)len;
        float *ring = A.rings + (size_t)inst * (size_t)A.ring_samples + (size_t)op.ring_base;
        uint32_t *own = (uint32_t *)scratch;
        constexpr uint32_t kOwnMask = 1023u, kFree = 0xffffffffu;
#pragma unroll
        for (int k = 0; k < 4; ++k) ((uint4 *)own)[lane + 64 * k] = uint4{kFree, kFree, kFree, kFree};
        // what operation j of a sample does
        int kind[3];
        kind[0] = is_delay ? RO_READ_CLEAR : is_mono ? RO_ADD : is_readback ? RO_STORE
                  : is_reader ? ((op.attr & 1) ? RO_READ_CLEAR : RO_READ)
                  : (op.attr & 1) ? RO_STORE : writer_mixes ? RO_ADD : RO_NONE;  // writer: preWipe then mix = store; mix alone = add
        kind[1] = is_delay || is_mono ? RO_ADD : is_readback ? RO_READ : RO_NONE;
        kind[2] = is_delay ? RO_ADD : is_mono ? RO_READ_CLEAR : RO_NONE;
        const double T0 = ss[0];  // ReadBackDelay / CircleBuffer nodes: the unit's running sample count
        const uint32_t tb0 = is_readback ? (uint32_t)(int64_t)fmod(T0, dlen) : (uint32_t)((A.clock0 + (uint64_t)g * kChunk) % (uint64_t)len);
        int32_t slot[4][3];
        double val[4][3];
        uint32_t pending = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t t = lane * 4 + c;
            const uint32_t tb = (tb0 + t) % len;
            out.v[c] = 0.f;
            // (selects, not `slot[c][j_read] = ..`: a runtime index would put the arrays into scratch memory)
            if (is_reader || is_writer) {  // CircleBuffer.js:16-18: floor(t % len), negatives wrapped; NaN / Inf go nowhere
                const double at = is_reader ? T0 + (double)t - srd * (double)p0.v[c] : T0 + (double)t + srd * (double)p0.v[c];
                double m = (at >= 0.0 && at < dlen) ? at : fmod(at, dlen);
                m = floor(m);
                if (m < 0.0) m += dlen;
                const bool valid = m >= 0.0 && m < dlen;
                slot[c][0] = valid ? (int32_t)m : -1; val[c][0] = writer_mixes ? (double)p1.v[c] : 0.0;
                slot[c][1] = slot[c][2] = -1; val[c][1] = val[c][2] = 0.0;
                if (is_reader && !valid) out.v[c] = __builtin_nanf("");
            } else if (is_readback) {
                double r = (T0 + (double)t) - (double)p1.v[c] + dlen;
                r = (r >= 0.0 && r < dlen) ? r : fmod(r, dlen);
                const bool valid = r >= 0.0 && r < dlen && r == floor(r);  // a fractional or negative index reads `undefined`
                slot[c][0] = (int32_t)tb; val[c][0] = (double)p0.v[c];
                slot[c][1] = valid ? (int32_t)r : -1; val[c][1] = 0.0;
                slot[c][2] = -1; val[c][2] = 0.0;
                if (!valid) out.v[c] = __builtin_nanf("");
            } else {
                const double xin = (double)p0.v[c];
                double tWrite = (double)tb + (double)p1.v[c];
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
            // (plain accesses: the wavefront's ring traffic goes through one L1 in order, and each round ends with vmcnt(0))
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
                        out.v[c] = was[c][j];
                        if (kind[j] == RO_READ_CLEAR && live) *p = 0.f;
                    } else if (kind[j] == RO_STORE) {
                        if (live) *p = (float)val[c][j];
                    } else if (live)
                        *p = (float)((double)was[c][j] + val[c][j]);
                }
            pending &= ~won;
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this round's ring traffic has landed before the next one starts
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 63 && is_delay) ss[0] = (double)p0.v[3];
        if (lane == 0 && (is_readback || is_reader || is_writer)) ss[0] = T0 + (double)kChunk;
        __builtin_amdgcn_wave_barrier();
    };
    (void)ordered_ring_ops;

    for (uint32_t g = g_begin; g < g_end; ++g) {
        const uint64_t n0 = (uint64_t)g * kChunk + lane * 4;  // index of this lane's first sample
        for (uint32_t u = 0; u < A.n_ops; ++u) {
            const DevOp &op = A.ops[u];
            V4 out;
            switch (op.op) {
            case OP_OSC: {  // Osc.js:35-47
                unsigned long long *carry = (unsigned long long *)(opstate + (size_t)op.lds_slot * kOpState);
                uint32_t *poison = (uint32_t *)(carry + 1);
                const V4 f = load_operand(op.in[0], bufs, lane, pvals, pstride, 0);
                long long q[4];
                bool bad = false;
                const bool lane_constant = op.in[0].kind != SRC_BUF;  // wave-uniform: f is a constant / parameter
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (lane_constant && c > 0) { q[c] = q[0]; continue; }
                    double fd = (double)f.v[c];
                    const bool fin = fabs(fd) <= 3.0e38;
                    bad = bad || !fin;
                    if (!fin) fd = 0.0;
                    if (fabs(fd) >= srd) fd = fmod(fd, srd);  // (a + b) % m == (a + b % m) % m
                    q[c] = (long long)(fd * kTwo36);         // exact for |f| >= 2^-13 (or f == 0)
                }
                long long s0, before;
                if (lane_constant) {  // equal increments: the prefix is a product, no scan needed
                    before = (long long)*carry + q[0] * (long long)(lane * 4);
                    s0 = q[0];
                } else {
                    const long long total = q[0] + q[1] + q[2] + q[3];
                    const long long incl = wave_inclusive_scan(total, lane);
                    before = (long long)*carry + (incl - total);
                    s0 = q[0];
                }
                // poison: NaN/Inf increments make the reference's phase NaN from that sample on
                const unsigned long long bad_lanes = __ballot(bad);
                const bool poisoned_before = *poison != 0 || (bad_lanes & ((1ull << lane) - 1ull)) != 0;
                const float *gtab = A.tables + (size_t)op.attr * A.table_stride;
                const bool in_lds = TBL == 1 && op.attr == A.lds_table_id;
                // phase of this lane's first sample by one exact modulo; the next three by add + single wrap (|q| < S)
                unsigned long long P = mod_u64_lifted((unsigned long long)(before + s0) + lift, S, inv_S);
                bool dead = poisoned_before;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c > 0) {
                        long long Pn = (long long)P + q[c];
                        if (Pn < 0) Pn += (long long)S;
                        if (Pn >= (long long)S) Pn -= (long long)S;
                        P = (unsigned long long)Pn;
                    }
                    dead = dead || (lane_constant ? bad : !(fabs((double)f.v[c]) <= 3.0e38));
                    const uint32_t idx = (uint32_t)(P >> kFracBits);
                    const double fraction = (double)(P & ((1ull << kFracBits) - 1ull)) * (1.0 / kTwo36);
                    float ta, tb;
                    if (in_lds) lds_table.pair(idx, ta, tb);
                    else { ta = gtab[idx]; tb = gtab[idx + 1]; }
                    out.v[c] = dead ? __builtin_nanf("") : (float)((double)ta * (1.0 - fraction) + (double)tb * fraction);
                }
                const unsigned long long lastP = __shfl(P, 63, 64);  // phase of the chunk's last sample
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) {
                    *carry = lastP;
                    if (bad_lanes) *poison = 1;
                }
                break;
            }
            case OP_RAMP: {  // Ramp.js:25-40 in closed form: t(n) = min(t0 + n + 1, duration) while playing
                const double duration = op.d[0], y0 = op.d[1], dy = op.d[2] - op.d[1];
                double t0 = A.init_state[op.state_slot];
                bool playing = A.init_state[op.state_slot + 1] != 0.0;
                if (EXT >= 2 && op.lds_slot >= 0) {  // restarted by a Retriggerer: t was rs[0] at sample rs[1] of this launch (Ramp.js:19-23)
                    const double *rs = opstate + (size_t)op.lds_slot * kOpState;
                    playing = rs[2] != 0.0;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const double tt = playing ? fmin(rs[0] + ((double)(n0 + c + 1) - rs[1]), duration) : rs[0];
                        out.v[c] = (float)(y0 + (tt / duration) * dy);
                    }
                    break;
                }
                if (op.attr & 1) {  // host-verified: the refined reciprocal equals tt / duration on this Ramp's whole t sequence
                    const double rcp = 1.0 / duration;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const double tt = playing ? fmin(t0 + (double)(n0 + c + 1), duration) : t0;
                        double q = tt * rcp;
                        q = fma(fma(-q, duration, tt), rcp, q);
                        out.v[c] = (float)(y0 + q * dy);
                    }
                    break;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const double tt = playing ? fmin(t0 + (double)(n0 + c + 1), duration) : t0;
                    out.v[c] = (float)(y0 + (tt / duration) * dy);
                }
                break;
            }
            case OP_FILTER: if constexpr (FILT != 0) {  // Filter.js:27-51
                double *fs = opstate + (size_t)op.lds_slot * kOpState;
                const V4 x = load_operand(op.in[0], bufs, lane, pvals, pstride, 0);
                const V4 fv = load_operand(op.in[1], bufs, lane, pvals, pstride, 0);
                const bool f_const = op.in[1].kind != SRC_BUF;  // wave-uniform
                double k0[5];
                if (f_const) {
                    // `if(this.f[t] != this.lastF)`: with a constant f the coefficients change at most once
                    const double ft = (double)fv.v[0];
                    if (fs[0] == 0.0 || ft != fs[1]) filter_coefficients(op.attr, ft, srd, k0);
                    else { k0[0] = fs[2]; k0[1] = fs[3]; k0[2] = fs[4]; k0[3] = fs[5]; k0[4] = fs[6]; }
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) { fs[0] = 1.0; fs[1] = ft; fs[2] = k0[0]; fs[3] = k0[1]; fs[4] = k0[2]; fs[5] = k0[3]; fs[6] = k0[4]; }
                }
                // feed-forward half, per lane: ((a0 x + a1 (x1||0)) + a2 (x2||0)) with x1, x2 = the two previous inputs
                const float xl1 = __shfl_up(x.v[3], 1, 64), xl2 = __shfl_up(x.v[2], 1, 64);
                double xm1 = lane == 0 ? fs[7] : (double)xl1, xm2 = lane == 0 ? fs[8] : (double)xl2;
                double klast[5] = {k0[0], k0[1], k0[2], k0[3], k0[4]};
                if (f_const) {
                    // ---- cooperative form: every wave of the workgroup (= WAVES instances) publishes its chunk's
                    // feed-forward half, then ONE wave runs the output recurrence with lane = instance, so the
                    // serial instruction stream is shared by WAVES instances instead of being repeated per wave.
                    double *prow = Pt + wave * 258;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const double xin = (double)x.v[c];
                        prow[lane * 4 + c] = (k0[0] * xin + k0[1] * or0w(xm1)) + k0[2] * or0w(xm2);
                        xm2 = or0w(xm1);
                        xm1 = xin;
                    }
                    if (lane == 63) { fs[7] = xm1; fs[8] = xm2; }
                    lds_barrier();
                    if (wave == 0 && lane < WAVES) {
                        double *os = (double *)((char *)lds + A.table_bytes + (size_t)lane * A.wave_bytes + (size_t)A.n_bufs * 1024) +
                                     (size_t)op.lds_slot * kOpState;  // instance `lane`'s state of this Filter
                        const double b1 = os[5], b2 = os[6];
                        double y1 = os[9], y2 = os[10];
                        const double *pr = Pt + lane * 258;
                        f32x4 *yr = (f32x4 *)(Yt + lane * 260);
                        // A wave issues one instruction per ~4-6 cycles whatever its lane count, so the serial stage costs
                        // (instructions per sample) x 256: the P values of a 16-sample block are pulled into registers
                        // first (an LDS read left on the chain costs ~100 cycles), and the `|| 0` selects of Filter.js:42-46
                        // are speculated away — without them a NaN never leaves the recurrence, so testing the block's
                        // last outputs finds one anywhere in it (the block is then redone exactly), and a -0 in place of
                        // +0 can only flip the sign of a later zero (see loop2_engine.hip).
                        constexpr int PB = WAVES >= 16 ? 8 : 16;  // P values per block held in registers (the 16-wave variants have 128 VGPRs)
                        for (int t0 = 0; t0 < kChunk; t0 += PB) {
                            double pv[PB];
#pragma unroll
                            for (int k = 0; k < PB; ++k) pv[k] = pr[t0 + k];
                            __builtin_amdgcn_sched_barrier(0);
                            const double y1_in = y1, y2_in = y2;
                            double u1 = or0w(y1), u2 = or0w(y2);
                            f32x4 y4[PB / 4];
#pragma unroll
                            for (int k = 0; k < PB; ++k) {
                                const float y = (float)((pv[k] - b1 * u1) - b2 * u2);
                                y4[k >> 2][k & 3] = y;
                                u2 = u1;
                                u1 = (double)y;
                            }
                            if (u1 == u1 && u2 == u2) {
                                y1 = u1;
                                y2 = u2;  // = y2 || 0 up to the sign of a zero
                            } else {
                                y1 = y1_in;
                                y2 = y2_in;
#pragma unroll
                                for (int k = 0; k < PB; ++k) {
                                    const float y = (float)((pv[k] - b1 * or0w(y1)) - b2 * or0w(y2));  // Filter.js:40-46
                                    y4[k >> 2][k & 3] = y;
                                    y2 = or0w(y1);
                                    y1 = (double)y;
                                }
                            }
#pragma unroll
                            for (int k = 0; k < PB / 4; ++k) yr[(t0 >> 2) + k] = y4[k];
                        }
                        os[9] = y1;
                        os[10] = y2;
                    }
                    lds_barrier();
                    const f32x4 yv = *(const f32x4 *)(Yt + wave * 260 + lane * 4);
                    out.v[0] = yv[0]; out.v[1] = yv[1]; out.v[2] = yv[2]; out.v[3] = yv[3];
                    break;
                }
                // ---- modulated cutoff: coefficients per sample, recurrence per wave out of its own scratch (EXT variants only:
                // a tan() per sample is the bulkiest code of the kernel)
                if constexpr (EXT < 2) break;
                else {
                double *P = scratch, *B1 = scratch + kChunk, *B2 = scratch + 2 * kChunk;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    filter_coefficients(op.attr, (double)fv.v[c], srd, klast);  // a function of f[t] alone
                    B1[lane * 4 + c] = klast[3];
                    B2[lane * 4 + c] = klast[4];
                    const double xin = (double)x.v[c];
                    P[lane * 4 + c] = (klast[0] * xin + klast[1] * or0w(xm1)) + klast[2] * or0w(xm2);
                    xm2 = or0w(xm1);
                    xm1 = xin;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                double y1 = fs[9], y2 = fs[10];
                float *outf = (float *)(bufs + (size_t)op.out_buf * 64);
#pragma unroll 8
                for (int t = 0; t < kChunk; ++t) {
                    const float y = (float)((P[t] - B1[t] * or0w(y1)) - B2[t] * or0w(y2));
                    if (lane == 0) outf[t] = y;
                    y2 = or0w(y1);
                    y1 = (double)y;
                }
                __builtin_amdgcn_wave_barrier();
                if (lane == 63) {
                    fs[7] = xm1;
                    fs[8] = xm2;
                    fs[0] = 1.0; fs[1] = (double)fv.v[3]; fs[2] = klast[0]; fs[3] = klast[1]; fs[4] = klast[2]; fs[5] = klast[3]; fs[6] = klast[4];
                }
                if (lane == 0) { fs[9] = y1; fs[10] = y2; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                continue;  // the output chunk is already in LDS
                }
            } else break;
            case OP_DELAY:  // Delay.js:20-41
              if (FILT != 0 && (!RING || delay_is_write_once(op))) {  // constant delay D + phi with 256 <= D <= len - 256: every slot is written once
                double *ds = opstate + (size_t)op.lds_slot * kOpState;
                const V4 x = load_operand(op.in[0], bufs, lane, pvals, pstride, 0);
                const int64_t len = op.ring_len;
                double dconst = (double)op.in[1].cval;
                if (dconst >= (double)len) dconst = fmod(dconst, (double)len);
                const double Dfl = floor(dconst), phi = dconst - Dfl;
                const int64_t D = (int64_t)Dfl;
                float *ring = A.rings + (size_t)inst * (size_t)A.ring_samples + (size_t)op.ring_base;
                const int64_t s0 = (int64_t)((A.clock0 + (uint64_t)g * kChunk) % (uint64_t)len);
                float x_left = __shfl_up(x.v[3], 1, 64);
                const double carried = ds[0];
                // In a chain of continued renders the ring is kept in exactly the state the reference's ring has at a
                // chunk boundary — slots zeroed once read (Delay.js:29), the last sample's ceil tap in place — so that the
                // chain can move to the chunk engine's read-modify-write protocol whenever an event changes the delay.
                const bool exact_ring = A.save_bufs != 0;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    int64_t s_ = s0 + lane * 4 + c;
                    if (s_ >= len) s_ -= len;
                    out.v[c] = ring[s_];
                    if (exact_ring && live) ring[s_] = 0.f;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    int64_t s_ = s0 + lane * 4 + c;
                    if (s_ >= len) s_ -= len;
                    int64_t lo = s_ + D;
                    if (lo >= len) lo -= len;
                    const double xin = (double)x.v[c];
                    const double xprev = c == 0 ? (lane == 0 ? carried : (double)x_left) : (double)x.v[c - 1];
                    float slot;
                    if (phi != 0.0) {
                        slot = lo != 0 ? (float)(0.0 + xprev * phi) : 0.f;  // ceil tap of sample n-1 (dropped at slot 0)
                        slot = (float)((double)slot + xin * (1.0 - phi));   // floor tap of sample n
                    } else {
                        slot = (float)(0.0 + xin * 1.0);
                        slot = (float)((double)slot + xin * 0.0);
                    }
                    if (live) ring[lo] = slot;
                    if (exact_ring && live && c == 3 && lane == 63 && g + 1 == g_end && phi != 0.0 && lo + 1 < len)
                        ring[lo + 1] = (float)(0.0 + xin * phi);  // the launch's last ceil tap, which the next launch would fold in
                }
                __builtin_amdgcn_wave_barrier();
                if (lane == 63) ds[0] = (double)x.v[3];
                break;
              }
              [[fallthrough]];
            case OP_MONO_DELAY: case OP_READBACK_DELAY:
                if constexpr (RING != 0) ordered_ring_ops(op, u, g, out);  // (launch_wave_engine picks the RING variant whenever the plan has such a unit)
                break;
            case OP_RETRIGGER: if constexpr (EXT >= 2) {  // Retriggerer.js:13-24 on one lane; a firing rewrites the target's state block before the target ticks
                double *ss = opstate + (size_t)op.lds_slot * kOpState;  // [0] t
                if (lane == 0) {
                    const double rate = (double)(op.in[0].kind == SRC_PARAM ? pvals[(size_t)op.in[0].idx * pstride] : op.in[0].cval);
                    double T = ss[0];
                    bool fired = false;
                    // most chunks see no crossing: then the accumulator is a plain running sum, which repeat_add() evaluates at once
                    const double quiet_end = (T >= 0.0 && rate > 0.0 && rate < 1.0e300) ? repeat_add(T, rate, kChunk) : srd;
                    if (quiet_end < srd) T = quiet_end;
                    else {
#pragma unroll 8
                        for (int k = 0; k < kChunk; ++k) {
                            T += rate;
                            if (T >= srd) { fired = true; T -= srd; }
                        }
                    }
                    ss[0] = T;
                    if (fired) {
                        double *ts = opstate + (size_t)op.pad * kOpState;  // (pad: the target op's state block)
                        if ((int)op.d[0] == OP_SHAPE) { ts[0] = 0.0; ts[1] = 1.0; }
                        else if ((int)op.d[0] == OP_RAMP) {
                            // restart: t = 0 before this chunk's first sample (rs[1] = that sample's index in this launch)
                            ts[0] = 0.0; ts[1] = (double)((uint64_t)g * kChunk); ts[2] = 1.0;
                        } else { ts[0] = 1.0; ts[1] = 1.0; }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                continue;  // no outlet
            } else break;
            case OP_INPUT: if constexpr (EXT >= 2) {  // a signal the host computed (Noise): this lane's four samples of stream op.attr
                const float *src = A.inputs + ((size_t)op.attr * A.n_inst + inst) * A.n_samples;
#pragma unroll
                for (int c = 0; c < 4; ++c) out.v[c] = n0 + c < A.n_samples ? src[n0 + c] : 0.f;
                break;
            } else break;
            case OP_MULTIPLY: {  // Multiply.js:23-34
                const V4 x = load_operand(op.in[0], bufs, lane, pvals, pstride, 0);
                const V4 y = load_operand(op.in[1], bufs, lane, pvals, pstride, 0);
                for (int c = 0; c < 4; ++c) out.v[c] = x.v[c] * y.v[c];
                break;
            }
            case OP_SUM: {  // Sum.js:33-44
                const V4 x = load_operand(op.in[0], bufs, lane, pvals, pstride, 0);
                const V4 y = load_operand(op.in[1], bufs, lane, pvals, pstride, 0);
                for (int c = 0; c < 4; ++c) out.v[c] = x.v[c] + y.v[c];
                break;
            }
            // ---- units whose state evolves sample by sample with its own roundings: the sequential part runs on lane 0
            // out of the wave's LDS scratch (a few instructions per sample), everything else stays lane-parallel
            case OP_SHAPE: if constexpr (EXT >= 2) {  // Shape/index.js:28-59
                double *ss = opstate + (size_t)op.lds_slot * kOpState;  // [0] t [1] playing [2] finished
                const V4 dur = load_operand(op.in[0], bufs, lane, pvals, pstride, 0);
                const V4 mn = load_operand(op.in[1], bufs, lane, pvals, pstride, 0);
                const V4 mx = load_operand(op.in[2], bufs, lane, pvals, pstride, 0);
                const float *data = A.tables + (size_t)(op.attr & 255) * A.table_stride;
                const double left = (op.attr & 256) ? (double)data[0] : op.d[0];
                const double right = (op.attr & 512) ? (double)data[sr] : op.d[1];
                double tt[4];
                const double c_lane = 1.0 / (double)dur.v[0];
                if (ss[1] != 0.0 && op.in[0].kind != SRC_BUF && ss[0] >= 0.0 && c_lane > 0.0 && c_lane < 1.0e300) {
                    // playing with a constant duration: the running sum in closed form (repeat_add), every lane its own four samples
                    const double t0 = ss[0];
                    long long T, ce;
                    int K;
                    double t_end;
                    if (linear_run(t0, c_lane, kChunk, T, ce, K)) {  // the whole chunk inside one binade: t_j = (T + j ce) 2^(K-52)
                        long long Tl = T + (long long)(lane * 4) * ce;
#pragma unroll
                        for (int c = 0; c < 4; ++c) tt[c] = ldexp((double)(Tl += ce), K - 52);
                        t_end = ldexp((double)(T + (long long)kChunk * ce), K - 52);
                    } else {
                        double t = repeat_add(t0, c_lane, (uint64_t)lane * 4);
#pragma unroll
                        for (int c = 0; c < 4; ++c) tt[c] = t = t + c_lane;
                        t_end = __shfl(t, 63, 64);
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) ss[0] = t_end;
                } else if (ss[1] != 0.0) {  // playing: t += 1 / duration[t], a running f64 sum
                    double *T = scratch;
#pragma unroll
                    for (int c = 0; c < 4; ++c) T[lane * 4 + c] = 1.0 / (double)dur.v[c];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) {
                        double t = ss[0];
#pragma unroll 8
                        for (int k = 0; k < kChunk; ++k) {
                            t += T[k];
                            T[k] = t;
                        }
                        ss[0] = t;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int c = 0; c < 4; ++c) tt[c] = T[lane * 4 + c];
                    __builtin_amdgcn_wave_barrier();
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) tt[c] = ss[0];
                }
                bool over = false;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const double l = (double)mn.v[c], h = (double)mx.v[c], t = tt[c];
                    if (t <= 0.0) out.v[c] = (float)(left * (h - l) + l);
                    else if (t > srd) { over = true; out.v[c] = (float)(right * (h - l) + l); }
                    else if (t == t) {
                        const double fl = floor(t), frac = t - fl;
                        out.v[c] = (float)(l + (h - l) * ((double)data[(int)ceil(t)] * frac + (double)data[(int)fl] * (1.0 - frac)));
                    } else out.v[c] = __builtin_nanf("");
                }
                if (__ballot(over) && lane == 0) ss[2] = 1.0;  // finish() (UnitOrPatch.js:77-84)
                break;
            } else break;
            case OP_TIMER: if constexpr (EXT >= 2) {  // Timer.js:36-41: t += samplePeriod, rounded to f32 per sample
                double *ss = opstate + (size_t)op.lds_slot * kOpState;
                const double period = op.d[0], t0 = ss[0];
                if (t0 >= 0.0 && period > 0.0 && period < 1.0e300) {  // the running sum in closed form, every lane its own four samples
                    long long T, ce;
                    int K;
                    double t_end;
                    if (linear_run(t0, period, kChunk, T, ce, K)) {
                        long long Tl = T + (long long)(lane * 4) * ce;
#pragma unroll
                        for (int c = 0; c < 4; ++c) out.v[c] = (float)ldexp((double)(Tl += ce), K - 52);
                        t_end = ldexp((double)(T + (long long)kChunk * ce), K - 52);
                    } else {
                        double t = repeat_add(t0, period, (uint64_t)lane * 4);
#pragma unroll
                        for (int c = 0; c < 4; ++c) out.v[c] = (float)(t = t + period);
                        t_end = __shfl(t, 63, 64);
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) ss[0] = t_end;
                    break;
                }
                float *Y = (float *)scratch;
                if (lane == 0) {
                    double t = t0;
#pragma unroll 8
                    for (int k = 0; k < kChunk; ++k) {
                        t += period;
                        Y[k] = (float)t;
                    }
                    ss[0] = t;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const f32x4 y = ((const f32x4 *)Y)[lane];
                out.v[0] = y[0]; out.v[1] = y[1]; out.v[2] = y[2]; out.v[3] = y[3];
                __builtin_amdgcn_wave_barrier();
                break;
            } else break;
            case OP_AHD: case OP_SAMPLE_RATE_REDUX: if constexpr (EXT >= 2) {  // AHD.js:35-76, SampleRateRedux.js:21-38
                double *ss = opstate + (size_t)op.lds_slot * kOpState;
                float *Y = (float *)scratch;  // [256] output, then up to three operand rows
                // operands as plain rows: a connected inlet is the producer's chunk buffer, a constant fills a scratch row
                const float *rows[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const DevOperand &o = op.in[k];
                    if (o.kind == SRC_BUF) rows[k] = (const float *)(bufs + (size_t)o.idx * 64);
                    else {
                        float *r = Y + (k + 1) * kChunk;
                        const float cst = o.kind == SRC_PARAM ? pvals[(size_t)o.idx * pstride] : o.cval;
                        ((f32x4 *)r)[lane] = f32x4{cst, cst, cst, cst};
                        rows[k] = r;
                    }
                }
                ((f32x4 *)Y)[lane] = bufs[(size_t)op.out_buf * 64 + lane];  // AHD: an unknown `state` leaves samples as they were
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) {
                    if (op.op == OP_AHD) {
                        int stage = (int)ss[0];
                        bool playing = ss[1] != 0.0;
                        double t = ss[2];
                        const double period = op.d[0];
                        for (int k = 0; k < kChunk; ++k) {
                            if (stage == 1) {
                                Y[k] = (float)t;
                                if (playing) { t += period / (double)rows[0][k]; if (t >= 1.0) { ++stage; t -= 1.0; } }
                            } else if (stage == 2) {
                                Y[k] = 1.f;
                                if (playing) { t += period / (double)rows[1][k]; if (t >= 1.0) { ++stage; t -= 1.0; } }
                            } else if (stage == 3) {
                                Y[k] = (float)(1.0 - t);
                                if (playing) { t += period / (double)rows[2][k]; if (t >= 1.0) { stage = 0; playing = false; } }
                            } else if (stage == 0)
                                Y[k] = 0.f;
                        }
                        ss[0] = (double)stage;
                        ss[1] = playing ? 1.0 : 0.0;
                        ss[2] = t;
                    } else {
                        double since = ss[0];
                        float held = (float)ss[1];
                        for (int k = 0; k < kChunk; ++k) {
                            since += 1.0;
                            if (since > (double)rows[1][k]) { held = rows[0][k]; since = 0.0; }
                            Y[k] = held;
                        }
                        ss[0] = since;
                        ss[1] = (double)held;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const f32x4 y = ((const f32x4 *)Y)[lane];
                out.v[0] = y[0]; out.v[1] = y[1]; out.v[2] = y[2]; out.v[3] = y[3];
                __builtin_amdgcn_wave_barrier();
                break;
            } else break;
            case OP_FIXED_DELAY: case OP_COMB_FILTER: case OP_ALL_PASS: if constexpr (EXT >= 2) {  // FixedDelay.js:13-19, CombFilter.js:11-17, AllPass.js:8-15
                // A private ring of L slots read and rewritten one slot per sample: sample t depends on sample t - L only,
                // so a chunk is L (at most 64) independent samples at a time.  The slots the chunk touches — min(L, 256)
                // of them — are staged in LDS, walked in rounds, and written back.
                double *ss = opstate + (size_t)op.lds_slot * kOpState;  // [0] tBuffer
                float *Y = (float *)scratch, *R = Y + kChunk;
                const float *rows[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const DevOperand &o = op.in[k];
                    if (o.kind == SRC_BUF) rows[k] = (const float *)(bufs + (size_t)o.idx * 64);
                    else {
                        float *r = Y + (k + 2) * kChunk;
                        const float cst = o.kind == SRC_PARAM ? pvals[(size_t)o.idx * pstride] : o.cval;
                        ((f32x4 *)r)[lane] = f32x4{cst, cst, cst, cst};
                        rows[k] = r;
                    }
                }
                const uint32_t L = (uint32_t)op.ring_len;
                const uint32_t first = ((uint32_t)ss[0] + 1u) % L;  // slot of the chunk's first sample: (tBuffer + 1) % length
                const uint32_t window = L < (uint32_t)kChunk ? L : (uint32_t)kChunk;
                float *ring = A.rings + (size_t)inst * (size_t)A.ring_samples + (size_t)op.ring_base;
                for (uint32_t p = lane; p < window; p += 64) {
                    uint32_t s_ = first + p;
                    if (s_ >= L) s_ -= L;
                    R[p] = ring[s_];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const uint32_t step = L < 64u ? L : 64u;
                uint32_t p = lane;  // (base + lane) mod L, kept incrementally
                for (uint32_t base = 0; base < (uint32_t)kChunk; base += step) {
                    const uint32_t t = base + lane;
                    if (lane < step && t < (uint32_t)kChunk) {
                        const float xin = rows[0][t], was = R[p];
                        float now, y;
                        if (op.op == OP_FIXED_DELAY) { y = was; now = xin; }
                        else if (op.op == OP_COMB_FILTER) { y = was; now = (float)((double)xin + (double)was * (double)rows[1][t]); }
                        else {
                            const double g = (double)rows[1][t];
                            now = (float)((double)xin + (double)was * g);
                            y = (float)((double)was - (double)xin * g);
                        }
                        R[p] = now;
                        Y[t] = y;
                    }
                    p += step;
                    if (p >= L) p -= L;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                if (live)
                    for (uint32_t q = lane; q < window; q += 64) {
                        uint32_t s_ = first + q;
                        if (s_ >= L) s_ -= L;
                        ring[s_] = R[q];
                    }
                if (lane == 0) ss[0] = (double)(((uint32_t)ss[0] + (uint32_t)kChunk) % L);
                const f32x4 y4 = ((const f32x4 *)Y)[lane];
                out.v[0] = y4[0]; out.v[1] = y4[1]; out.v[2] = y4[2]; out.v[3] = y4[3];
                __builtin_amdgcn_wave_barrier();
                break;
            } else break;
            case OP_CB_READER: case OP_CB_WRITER: if constexpr (EXT >= 2) {  // CircleBufferReader.js:12-25, CircleBufferWriter.js:12-25, CircleBuffer.js:15-34
                if constexpr (RING != 0)
                    if (op.in[0].kind == SRC_BUF || op.ring_len < kChunk) {  // accesses that can meet inside the chunk
                        ordered_ring_ops(op, u, g, out);
                        if (op.op == OP_CB_WRITER) continue;  // no outlet
                        break;
                    }
                // Lane-constant offset: the node's 256 accesses of a chunk are 256 consecutive slots of the ring
                // (index = floor((T + t -+ sr*offset) % len), negatives wrapped), one per sample, lane-parallel.  Several
                // nodes share a ring and tick one after another, so each node waits for the wave's earlier ring traffic.
                double *ss = opstate + (size_t)op.lds_slot * kOpState;  // [0] the node's private sample counter
                const float off = op.in[0].kind == SRC_PARAM ? pvals[(size_t)op.in[0].idx * pstride] : op.in[0].cval;
                const double origin = op.op == OP_CB_READER ? ss[0] - srd * (double)off : ss[0] + srd * (double)off;
                const bool ok = fabs(origin) < 9.0e15;  // NaN / Inf offsets read `undefined` and write nowhere
                const int64_t len = op.ring_len;
                int64_t base = 0;
                if (ok) {
                    base = (int64_t)fmod(floor(origin), (double)len);
                    if (base < 0) base += len;
                }
                float *ring = A.rings + (size_t)inst * (size_t)A.ring_samples + (size_t)op.ring_base;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
                V4 x;
                if (op.op == OP_CB_WRITER && !(op.attr & 2)) x = load_operand(op.in[1], bufs, lane, pvals, pstride, 0);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    int64_t idx = base + lane * 4 + c;
                    if (idx >= len) idx -= len;
                    if (op.op == OP_CB_READER) {
                        out.v[c] = ok ? ring[idx] : __builtin_nanf("");
                        if (ok && (op.attr & 1) && live) ring[idx] = 0.f;  // postWipe
                    } else if (ok && live) {
                        float v = (op.attr & 1) ? 0.f : ring[idx];           // preWipe
                        if (!(op.attr & 2)) v = v + x.v[c];                  // mix
                        if ((op.attr & 1) || !(op.attr & 2)) ring[idx] = v;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) ss[0] += (double)kChunk;
                if (op.op == OP_CB_WRITER) continue;  // no outlet
                break;
            } else break;
            case OP_MULTI_OSC: if constexpr (EXT >= 2) {  // MultiChannelOsc.js:21-38: `phase += f; phase %= sr` WITHOUT the Osc's `if (phase < 0) phase += sr`
                // The remainder keeps the dividend's sign, so the phase is not a modular sum (a negative excursion reads
                // `undefined` -> NaN until the sum comes back): the 256 phases come from the serial lane, in f64 exactly
                // as the reference adds them; the table lookups and the lerp are lane-parallel.
                double *ss = opstate + (size_t)op.lds_slot * kOpState;  // [0] phase
                const V4 f = load_operand(op.in[0], bufs, lane, pvals, pstride, 0);
                const float *gtab = A.tables + (size_t)op.attr * A.table_stride;
                {   // While nothing is negative the missing fix-up cannot matter and the phase IS the Osc's modular sum: every
                    // increment of the chunk finite, >= 0 and on the 2^-36 grid, the start phase too -> the Osc's exact
                    // fixed-point wave scan (FM voices built on MultiChannelOsc: the FMOsc / StereoDetune patches).
                    double ph0 = ss[0];
                    ph0 = (ph0 != ph0 || ph0 == 0.0) ? 0.0 : ph0;  // `this.phase[c] = this.phase[c] || 0`
                    bool grid = ph0 >= 0.0 && ph0 < srd && ph0 * kTwo36 == floor(ph0 * kTwo36);
                    long long q[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        double fd = (double)f.v[c];
                        grid = grid && fd >= 0.0 && fd < srd;  // (below sampleRate: phase + f stays under 2^17 + 2^17, exact in f64 on this grid)
                        if (!(fd >= 0.0 && fd < srd)) fd = 0.0;
                        const double scaled = fd * kTwo36;
                        grid = grid && scaled == floor(scaled);
                        q[c] = (long long)scaled;
                    }
                    if (__all(grid)) {
                        const long long total = q[0] + q[1] + q[2] + q[3];
                        const long long incl = wave_inclusive_scan(total, lane);
                        const long long before = (long long)(unsigned long long)(ph0 * kTwo36) + (incl - total);
                        unsigned long long P = mod_u64_lifted((unsigned long long)(before + q[0]) + lift, S, inv_S);
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            if (c > 0) {
                                P += (unsigned long long)q[c];
                                if (P >= S) P -= S;
                            }
                            const uint32_t idx = (uint32_t)(P >> kFracBits);
                            const double fraction = (double)(P & ((1ull << kFracBits) - 1ull)) * (1.0 / kTwo36);
                            out.v[c] = (float)((double)gtab[idx] * (1.0 - fraction) + (double)gtab[fraction != 0.0 ? idx + 1 : idx] * fraction);
                        }
                        const unsigned long long lastP = __shfl(P, 63, 64);
                        __builtin_amdgcn_wave_barrier();
                        if (lane == 0) ss[0] = (double)lastP * (1.0 / kTwo36);
                        break;
                    }
                }
                double *T = scratch;
#pragma unroll
                for (int c = 0; c < 4; ++c) T[lane * 4 + c] = (double)f.v[c];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) {
                    double ph = ss[0];
                    ph = (ph != ph || ph == 0.0) ? 0.0 : ph;  // `this.phase[c] = this.phase[c] || 0`
#pragma unroll 4
                    for (int k = 0; k < kChunk; ++k) {
                        double p = ph + T[k];
                        if (fabs(p) >= srd) p = (p > 0.0 && p < 2.0 * srd) ? p - srd : (p < 0.0 && p > -2.0 * srd) ? p + srd : fmod(p, srd);
                        T[k] = ph = p;
                    }
                    ss[0] = ph;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const double phase = T[lane * 4 + c];
                    if (!(phase >= 0.0 && phase <= srd)) { out.v[c] = __builtin_nanf(""); continue; }  // typed-array[NaN / negative] is undefined
                    const double lo = floor(phase), fraction = phase - lo;
                    const int idx = (int)lo;
                    out.v[c] = (float)((double)gtab[idx] * (1.0 - fraction) + (double)gtab[fraction != 0.0 ? idx + 1 : idx] * fraction);
                }
                __builtin_amdgcn_wave_barrier();
                break;
            } else break;
            case OP_REPEATER: {  // Repeater.js:23-30
                out = load_operand(op.in[0], bufs, lane, pvals, pstride, 0);
                break;
            }
            case OP_PAN: case OP_MIDI_TO_FREQUENCY: case OP_RESCALE: case OP_CROSS_FADER: case OP_VECTOR_MAGNITUDE: if constexpr (EXT >= 2) {
                V4 w[kMaxIn];
#pragma unroll
                for (int k = 0; k < kMaxIn; ++k)
                    if (k < op.n_in) w[k] = load_operand(op.in[k], bufs, lane, pvals, pstride, 0);
                    else w[k] = w[0];
                for (int c = 0; c < 4; ++c) {
                    const float v[kMaxIn] = {w[0].v[c], w[1].v[c], w[2].v[c], w[3].v[c], w[4].v[c]};
                    out.v[c] = map_wide(op.op, op.attr, op.n_in, v, op.d[0]);
                }
                break;
            } else break;
            default: {  // stateless elementwise maps (map_ops.hpp)
                const V4 x = load_operand(op.in[0], bufs, lane, pvals, pstride, 0);
                const V4 y = load_operand(op.in[1], bufs, lane, pvals, pstride, 0);
                for (int c = 0; c < 4; ++c) out.v[c] = map_apply(op.op, x.v[c], y.v[c], op.d[0]);
                break;
            }
            }
            bufs[(size_t)op.out_buf * 64 + lane] = f32x4{out.v[0], out.v[1], out.v[2], out.v[3]};
        }
        // copy-out (renderChannelData.js:35-44)
        for (uint32_t oc = 0; oc < A.n_out; ++oc) {
            const f32x4 x = bufs[(size_t)A.out_bufs[oc] * 64 + lane];
            float v[4] = {fix_out<false>(x[0]), fix_out<false>(x[1]), fix_out<false>(x[2]), fix_out<false>(x[3])};
            float *row = A.out + ((size_t)inst * A.n_out + oc) * A.n_samples + n0;
            if (!live || !rendering) continue;
            if (A.vec4_ok && n0 + 4 <= A.n_samples) store4<true>(row, v, n0, A.n_samples);
            else store4<false>(row, v, n0, A.n_samples);
        }
    }

    // state write-back: what every unit holds after ceil(n_samples/256) ticks, in the chunk engine's slot layout
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (A.save_bufs && live && rendering)
        for (uint32_t b = 0; b < A.n_bufs; ++b)
            ((f32x4 *)(A.saved_bufs + ((size_t)inst * A.n_bufs + b) * kChunk))[lane] = bufs[(size_t)b * 64 + lane];
    if (lane == 0 && live && !rendering) {  // phase totals of this pass's oscillators over this segment
        for (uint32_t u = 0; u < A.n_ops; ++u) {
            const DevOp &op = A.ops[u];
            if (op.op != OP_OSC || (uint32_t)op.d[0] != A.pass_level) continue;
            const double *os = opstate + (size_t)op.lds_slot * kOpState;
            A.seg_sum[((size_t)u * A.n_inst + inst) * A.n_seg + seg] =
                *(const unsigned long long *)os | ((unsigned long long)(((const uint32_t *)(os + 1))[0] != 0) << 63);
        }
    }
    if (lane == 0 && live && rendering && seg == A.n_seg - 1) {
        const uint64_t T_end = (uint64_t)A.n_groups * kChunk;
        for (uint32_t u = 0; u < A.n_ops; ++u) {
            const DevOp &op = A.ops[u];
            double *st = A.state + (size_t)op.state_slot * A.n_pad + inst;
            const double *os = opstate + (size_t)op.lds_slot * kOpState;
            if (op.op == OP_OSC) st[0] = ((const uint32_t *)(os + 1))[0] ? __builtin_nan("") : (double)*(const unsigned long long *)os * (1.0 / kTwo36);
            if (op.op == OP_DELAY || op.op == OP_TIMER || op.op == OP_FIXED_DELAY || op.op == OP_COMB_FILTER || op.op == OP_ALL_PASS ||
                op.op == OP_CB_READER || op.op == OP_CB_WRITER || op.op == OP_MULTI_OSC || op.op == OP_READBACK_DELAY || op.op == OP_RETRIGGER)
                st[0] = os[0];
            if (op.op == OP_FILTER)
                for (int k = 0; k < 11; ++k) st[(size_t)k * A.n_pad] = os[k];
            if (op.op == OP_SHAPE || op.op == OP_AHD)
                for (int k = 0; k < 3; ++k) st[(size_t)k * A.n_pad] = os[k];
            if (op.op == OP_SAMPLE_RATE_REDUX)
                for (int k = 0; k < 2; ++k) st[(size_t)k * A.n_pad] = os[k];
            if (op.op == OP_RAMP) {
                const double duration = op.d[0];
                double t0 = A.init_state[op.state_slot], since = (double)T_end;
                bool playing = A.init_state[op.state_slot + 1] != 0.0;
                if (op.lds_slot >= 0) { t0 = os[0]; since = (double)T_end - os[1]; playing = os[2] != 0.0; }  // since its last restart
                st[0] = playing ? fmin(t0 + since, duration) : t0;
                st[A.n_pad] = (playing && t0 + since <= duration) ? 1.0 : 0.0;
            }
        }
    }
}

// Start phase of every segment from the segments' phase totals: a serial modular prefix per (oscillator, instance) —
// n_seg additions, one thread each.  A poisoned segment poisons everything after it.
__global__ void dusp_wave_prefix_kernel(WaveArgs A) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= A.n_ops * A.n_inst) return;
    const uint32_t u = k / A.n_inst, inst = k - u * A.n_inst;
    const DevOp &op = A.ops[u];
    if (op.op != OP_OSC || (uint32_t)op.d[0] != A.pass_level) return;
    const unsigned long long S = (unsigned long long)A.sample_rate << kFracBits;
    unsigned long long phase = (unsigned long long)(A.init_state[op.state_slot] * kTwo36), poison = 0;
    const size_t base = ((size_t)u * A.n_inst + inst) * A.n_seg;
    for (uint32_t s = 0; s < A.n_seg; ++s) {
        A.seg_start[base + s] = phase | poison;
        const unsigned long long v = A.seg_sum[base + s];
        phase += v & ~(1ull << 63);  // both below S
        if (phase >= S) phase -= S;
        poison |= v & (1ull << 63);
    }
}

// A continued program that can no longer run on this engine (an event changed a constant, e.g. a Delay below one
// chunk) moves to the chunk engine: rings [instance][slot] -> [slot][instance], saved chunk buffers
// [instance][buffer][t] -> [buffer][t][instance].  Unit state already lives in the shared slot layout.
__global__ void dusp_wave_to_chunk_kernel(const float *wave_rings, float *chunk_rings, uint64_t ring_samples, const float *saved_bufs,
                                          float *chunk_scratch, uint32_t n_bufs, uint32_t n_inst, uint32_t n_pad) {
    const uint64_t total_r = ring_samples * n_inst, total_b = (uint64_t)n_bufs * kChunk * n_inst;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < total_r + total_b; k += (uint64_t)gridDim.x * blockDim.x) {
        if (k < total_r) {
            const uint64_t inst = k / ring_samples, slot = k - inst * ring_samples;
            chunk_rings[slot * n_pad + inst] = wave_rings[k];
        } else {
            const uint64_t j = k - total_r, inst = j / ((uint64_t)n_bufs * kChunk), bt = j - inst * (uint64_t)n_bufs * kChunk;
            chunk_scratch[bt * n_pad + inst] = saved_bufs[j];
        }
    }
}

// The other way: a program whose first chunks ran on the chunk engine (channel counts that grow at first: Program::warm_ops) goes on
// on a compiled kernel — rings [slot][instance] -> [instance][slot], every outlet's last chunk [buffer][t][instance] ->
// [instance][buffer][t], and the unit state of instance 0 becomes the state the kernel starts from (the compiled kernels read ONE
// start state for all instances: the hand-off is taken for single circuits).
__global__ void dusp_chunk_to_wave_kernel(const float *chunk_rings, float *wave_rings, uint64_t ring_samples, const float *chunk_scratch, float *saved_bufs,
                                          uint32_t n_bufs, uint32_t n_inst, uint32_t n_pad, const double *state, double *init_state, uint32_t n_slots) {
    const uint64_t total_r = ring_samples * n_inst, total_b = (uint64_t)n_bufs * kChunk * n_inst;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < total_r + total_b + n_slots; k += (uint64_t)gridDim.x * blockDim.x) {
        if (k < total_r) {
            const uint64_t inst = k / ring_samples, slot = k - inst * ring_samples;
            wave_rings[k] = chunk_rings[slot * n_pad + inst];
        } else if (k < total_r + total_b) {
            const uint64_t j = k - total_r, inst = j / ((uint64_t)n_bufs * kChunk), bt = j - inst * (uint64_t)n_bufs * kChunk;
            saved_bufs[j] = chunk_scratch[bt * n_pad + inst];
        } else {
            const uint64_t slot = k - total_r - total_b;
            init_state[slot] = state[slot * n_pad];
        }
    }
}
hipError_t launch_chunk_to_wave(const float *chunk_rings, float *wave_rings, uint64_t ring_samples, const float *chunk_scratch, float *saved_bufs, uint32_t n_bufs,
                                uint32_t n_inst, uint32_t n_pad, const double *state, double *init_state, uint32_t n_slots, hipStream_t stream) {
    hipLaunchKernelGGL(dusp_chunk_to_wave_kernel, dim3(1024), dim3(256), 0, stream, chunk_rings, wave_rings, ring_samples, chunk_scratch, saved_bufs, n_bufs,
                       n_inst, n_pad, state, init_state, n_slots);
    return hipGetLastError();
}

hipError_t launch_wave_to_chunk(const float *wave_rings, float *chunk_rings, uint64_t ring_samples, const float *saved_bufs, float *chunk_scratch,
                                uint32_t n_bufs, uint32_t n_inst, uint32_t n_pad, hipStream_t stream) {
    hipLaunchKernelGGL(dusp_wave_to_chunk_kernel, dim3(1024), dim3(256), 0, stream, wave_rings, chunk_rings, ring_samples, saved_bufs,
                       chunk_scratch, n_bufs, n_inst, n_pad);
    return hipGetLastError();
}

template <int TBL, int WAVES, int RING, int EXT, int FILT>
static hipError_t launch_wave_one(const WaveArgs &A, size_t lds_bytes, hipStream_t stream) {
    auto kernel = dusp_wave_kernel<TBL, WAVES, RING, EXT, FILT>;
    if (lds_bytes > 65536) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    const unsigned grid = (A.n_inst * A.n_seg + WAVES - 1) / WAVES;
    if (A.n_seg > 1) {  // one accumulate pass + prefix per FM level, then the render pass
        WaveArgs pass = A;
        pass.pass_mode = 1;
        for (int level = 0; level <= A.max_osc_level; ++level) {
            pass.pass_level = (uint32_t)level;
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(WAVES * 64), lds_bytes, stream, pass);
            hipLaunchKernelGGL(dusp_wave_prefix_kernel, dim3((A.n_ops * A.n_inst + 63) / 64), dim3(64), 0, stream, pass);
        }
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(WAVES * 64), lds_bytes, stream, A);
    return hipGetLastError();
}

// Picks the LDS geometry: half table (when the plan has an antisymmetric one) + per-wave chunk buffers.
hipError_t launch_wave_engine(WaveArgs A, bool lds_table_ok, int max_waves_cap, hipStream_t stream) {
    A.param_bytes = (uint32_t)wave_param_bytes(A.n_params);
    A.wave_bytes = (uint32_t)wave_lds_bytes(A.n_bufs, A.n_state_ops, A.scratch_bytes) + A.param_bytes;
    const size_t budget = 160 * 1024;
    size_t table_bytes = lds_table_ok && A.lds_table_id >= 0 ? half_table_lds_bytes(A.sample_rate) : 0;
    const size_t one_wave = A.wave_bytes + (A.has_filter ? 258 * 8 + 260 * 4 : 0);
    if (table_bytes && table_bytes + one_wave > budget) table_bytes = 0;  // buffers first; lookups fall back to L2
    if (one_wave > budget) return hipErrorInvalidValue;                   // plan_wave() guards this
    A.table_bytes = (uint32_t)table_bytes;
    if (!table_bytes) A.lds_table_id = -1;
    // waves per workgroup: as many as LDS holds next to the table image (they share it and hide each other's
    // scan / lookup latency), but no more than needed to give every CU a workgroup
    const size_t shared_per_wave = A.has_filter ? 258 * 8 + 260 * 4 : 0;  // Pt / Yt rows of the cooperative Filter stage
    const int fit = (int)((budget - table_bytes) / (A.wave_bytes + shared_per_wave));
    const unsigned want = (A.n_inst * A.n_seg + 255) / 256;  // (virtual) instances per CU on a 256-CU part
    int waves = 1;
    int most = A.ring_events ? 8 : 16;  // (the RING variant wants its 256 VGPRs: two waves per SIMD)
    if (max_waves_cap > 0) most = std::max(1, std::min(most, max_waves_cap));  // A/B knob (Knobs::wave_max_waves)
    while (waves < most && waves * 2 <= fit && (unsigned)waves < want) waves *= 2;
    const size_t lds_bytes = table_bytes + (size_t)waves * (A.wave_bytes + shared_per_wave);
#define DUSP_W(T, W, R, E, F) launch_wave_one<T, W, R, E, F>(A, lds_bytes, stream)
    // classes: 0 lean; 1 + Filter / Delay; 2 + the other units, no Filter / Delay; 3 everything; 4 everything + ordered slot operations
    const int cls = A.ring_events ? 4 : (A.ext_units & 2 ? 2 : 0) + (A.ext_units & 1);
    switch ((cls * 2 + (table_bytes ? 1 : 0)) * 32 + waves) {
    case 0 * 32 + 16: return DUSP_W(0, 16, 0, 0, 0);
    case 0 * 32 + 8: return DUSP_W(0, 8, 0, 0, 0);
    case 0 * 32 + 4: return DUSP_W(0, 4, 0, 0, 0);
    case 0 * 32 + 2: return DUSP_W(0, 2, 0, 0, 0);
    case 0 * 32 + 1: return DUSP_W(0, 1, 0, 0, 0);
    case 1 * 32 + 16: return DUSP_W(1, 16, 0, 0, 0);
    case 1 * 32 + 8: return DUSP_W(1, 8, 0, 0, 0);
    case 1 * 32 + 4: return DUSP_W(1, 4, 0, 0, 0);
    case 1 * 32 + 2: return DUSP_W(1, 2, 0, 0, 0);
    case 1 * 32 + 1: return DUSP_W(1, 1, 0, 0, 0);
    case 2 * 32 + 16: return DUSP_W(0, 16, 0, 0, 1);
    case 2 * 32 + 8: return DUSP_W(0, 8, 0, 0, 1);
    case 2 * 32 + 4: return DUSP_W(0, 4, 0, 0, 1);
    case 2 * 32 + 2: return DUSP_W(0, 2, 0, 0, 1);
    case 2 * 32 + 1: return DUSP_W(0, 1, 0, 0, 1);
    case 3 * 32 + 16: return DUSP_W(1, 16, 0, 0, 1);
    case 3 * 32 + 8: return DUSP_W(1, 8, 0, 0, 1);
    case 3 * 32 + 4: return DUSP_W(1, 4, 0, 0, 1);
    case 3 * 32 + 2: return DUSP_W(1, 2, 0, 0, 1);
    case 3 * 32 + 1: return DUSP_W(1, 1, 0, 0, 1);
    case 4 * 32 + 16: return DUSP_W(0, 16, 0, 2, 0);
    case 4 * 32 + 8: return DUSP_W(0, 8, 0, 2, 0);
    case 4 * 32 + 4: return DUSP_W(0, 4, 0, 2, 0);
    case 4 * 32 + 2: return DUSP_W(0, 2, 0, 2, 0);
    case 4 * 32 + 1: return DUSP_W(0, 1, 0, 2, 0);
    case 5 * 32 + 16: return DUSP_W(1, 16, 0, 2, 0);
    case 5 * 32 + 8: return DUSP_W(1, 8, 0, 2, 0);
    case 5 * 32 + 4: return DUSP_W(1, 4, 0, 2, 0);
    case 5 * 32 + 2: return DUSP_W(1, 2, 0, 2, 0);
    case 5 * 32 + 1: return DUSP_W(1, 1, 0, 2, 0);
    case 6 * 32 + 16: return DUSP_W(0, 16, 0, 2, 1);
    case 6 * 32 + 8: return DUSP_W(0, 8, 0, 2, 1);
    case 6 * 32 + 4: return DUSP_W(0, 4, 0, 2, 1);
    case 6 * 32 + 2: return DUSP_W(0, 2, 0, 2, 1);
    case 6 * 32 + 1: return DUSP_W(0, 1, 0, 2, 1);
    case 7 * 32 + 16: return DUSP_W(1, 16, 0, 2, 1);
    case 7 * 32 + 8: return DUSP_W(1, 8, 0, 2, 1);
    case 7 * 32 + 4: return DUSP_W(1, 4, 0, 2, 1);
    case 7 * 32 + 2: return DUSP_W(1, 2, 0, 2, 1);
    case 7 * 32 + 1: return DUSP_W(1, 1, 0, 2, 1);
    case 8 * 32 + 8: return DUSP_W(0, 8, 1, 2, 1);
    case 8 * 32 + 4: return DUSP_W(0, 4, 1, 2, 1);
    case 8 * 32 + 2: return DUSP_W(0, 2, 1, 2, 1);
    case 8 * 32 + 1: return DUSP_W(0, 1, 1, 2, 1);
    case 9 * 32 + 8: return DUSP_W(1, 8, 1, 2, 1);
    case 9 * 32 + 4: return DUSP_W(1, 4, 1, 2, 1);
    case 9 * 32 + 2: return DUSP_W(1, 2, 1, 2, 1);
    case 9 * 32 + 1: return DUSP_W(1, 1, 1, 2, 1);
    }
    return hipErrorInvalidValue;
#undef DUSP_W
}

}  // namespace dusp
