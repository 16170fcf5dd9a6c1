// wave_engine.hip — the "wave" engine: any FEED-FORWARD circuit of Osc / Ramp / Multiply / Sum /
// Repeater units (arbitrary fan-out, connected oscillator frequencies = FM), one WAVEFRONT per
// circuit instance, one LANE per SAMPLE.
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
// Time is sequential per instance (the scan carry), so parallelism = instances: thousands of voices
// fill the chip; a single circuit (BASELINE configs[1]) runs on one wave and is latency-bound
// (~0.7 us per chunk) — still two orders of magnitude faster than ticking it on the host.
#include <hip/hip_runtime.h>

#include "device_types.hpp"
#include "fused_device.hpp"
#include "map_ops.hpp"

namespace dusp {

namespace {

constexpr double kTwo36 = 68719476736.0;
constexpr int kFracBits = 36;

struct V4 {
    float v[4];
};

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

template <int TBL, int WAVES>
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
    const uint32_t inst = blockIdx.x * WAVES + wave;
    if (inst >= A.n_inst) return;  // whole waves only; no workgroup barrier follows
    char *mine = (char *)lds + A.table_bytes + (size_t)wave * A.wave_bytes;
    f32x4 *bufs = (f32x4 *)mine;                                            // [n_bufs][64] float4 = chunk buffers
    unsigned long long *carry = (unsigned long long *)(mine + (size_t)A.n_bufs * 1024);  // [n_ops] phase carry (2^-36 units)
    uint32_t *poison = (uint32_t *)(carry + A.n_ops);                        // [n_ops]

    const uint32_t sr = A.sample_rate;
    const double srd = (double)sr;
    const unsigned long long S = (unsigned long long)sr << kFracBits;
    const double inv_S = 1.0 / (double)S;
    // multiple of S that makes any |x| < 2^62 non-negative before the modulo
    const unsigned long long lift = S * ((1ull << 62) / S);

    if (lane == 0)
        for (uint32_t u = 0; u < A.n_ops; ++u) {
            const DevOp &op = A.ops[u];
            carry[u] = op.op == OP_OSC ? (unsigned long long)(A.init_state[op.state_slot] * kTwo36) : 0ull;
            poison[u] = 0;
        }
    // (single wave: LDS accesses of one wave are issued in order; the fence keeps the compiler honest)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    for (uint32_t g = 0; g < A.n_groups; ++g) {
        const uint64_t n0 = (uint64_t)g * kChunk + lane * 4;  // index of this lane's first sample
        for (uint32_t u = 0; u < A.n_ops; ++u) {
            const DevOp &op = A.ops[u];
            V4 out;
            switch (op.op) {
            case OP_OSC: {  // Osc.js:35-47
                const V4 f = load_operand(op.in[0], bufs, lane, A.params, A.n_inst, inst);
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
                    before = (long long)carry[u] + q[0] * (long long)(lane * 4);
                    s0 = q[0];
                } else {
                    const long long total = q[0] + q[1] + q[2] + q[3];
                    const long long incl = wave_inclusive_scan(total, lane);
                    before = (long long)carry[u] + (incl - total);
                    s0 = q[0];
                }
                // poison: NaN/Inf increments make the reference's phase NaN from that sample on
                const unsigned long long bad_lanes = __ballot(bad);
                const bool poisoned_before = poison[u] != 0 || (bad_lanes & ((1ull << lane) - 1ull)) != 0;
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
                    carry[u] = lastP;
                    if (bad_lanes) poison[u] = 1;
                }
                break;
            }
            case OP_RAMP: {  // Ramp.js:25-40 in closed form: t(n) = min(t0 + n + 1, duration) while playing
                const double duration = op.d[0], y0 = op.d[1], dy = op.d[2] - op.d[1];
                const double t0 = A.init_state[op.state_slot];
                const bool playing = A.init_state[op.state_slot + 1] != 0.0;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const double tt = playing ? fmin(t0 + (double)(n0 + c + 1), duration) : t0;
                    out.v[c] = (float)(y0 + (tt / duration) * dy);
                }
                break;
            }
            case OP_MULTIPLY: {  // Multiply.js:23-34
                const V4 x = load_operand(op.in[0], bufs, lane, A.params, A.n_inst, inst);
                const V4 y = load_operand(op.in[1], bufs, lane, A.params, A.n_inst, inst);
                for (int c = 0; c < 4; ++c) out.v[c] = x.v[c] * y.v[c];
                break;
            }
            case OP_SUM: {  // Sum.js:33-44
                const V4 x = load_operand(op.in[0], bufs, lane, A.params, A.n_inst, inst);
                const V4 y = load_operand(op.in[1], bufs, lane, A.params, A.n_inst, inst);
                for (int c = 0; c < 4; ++c) out.v[c] = x.v[c] + y.v[c];
                break;
            }
            case OP_REPEATER: {  // Repeater.js:23-30
                out = load_operand(op.in[0], bufs, lane, A.params, A.n_inst, inst);
                break;
            }
            default: {  // stateless elementwise maps (map_ops.hpp)
                const V4 x = load_operand(op.in[0], bufs, lane, A.params, A.n_inst, inst);
                const V4 y = load_operand(op.in[1], bufs, lane, A.params, A.n_inst, inst);
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
            if (A.vec4_ok && n0 + 4 <= A.n_samples) store4<true>(row, v, n0, A.n_samples);
            else store4<false>(row, v, n0, A.n_samples);
        }
    }

    // state write-back: what every unit holds after ceil(n_samples/256) ticks, in the chunk engine's slot layout
    if (lane == 0) {
        const uint64_t T_end = (uint64_t)A.n_groups * kChunk;
        for (uint32_t u = 0; u < A.n_ops; ++u) {
            const DevOp &op = A.ops[u];
            double *st = A.state + (size_t)op.state_slot * A.n_pad + inst;
            if (op.op == OP_OSC) st[0] = poison[u] ? __builtin_nan("") : (double)carry[u] * (1.0 / kTwo36);
            if (op.op == OP_RAMP) {
                const double duration = op.d[0], t0 = A.init_state[op.state_slot];
                const bool playing = A.init_state[op.state_slot + 1] != 0.0;
                st[0] = playing ? fmin(t0 + (double)T_end, duration) : t0;
                st[A.n_pad] = (playing && t0 + (double)T_end <= duration) ? 1.0 : 0.0;
            }
        }
    }
}

template <int TBL, int WAVES>
static hipError_t launch_wave_one(const WaveArgs &A, size_t lds_bytes, hipStream_t stream) {
    auto kernel = dusp_wave_kernel<TBL, WAVES>;
    if (lds_bytes > 65536) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    const unsigned grid = (A.n_inst + WAVES - 1) / WAVES;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(WAVES * 64), lds_bytes, stream, A);
    return hipGetLastError();
}

// Picks the LDS geometry: half table (when the plan has an antisymmetric one) + per-wave chunk buffers.
hipError_t launch_wave_engine(WaveArgs A, bool lds_table_ok, hipStream_t stream) {
    A.wave_bytes = (uint32_t)(((size_t)A.n_bufs * 1024 + (size_t)A.n_ops * 12 + 15) & ~(size_t)15);
    const size_t budget = 160 * 1024;
    size_t table_bytes = lds_table_ok && A.lds_table_id >= 0 ? half_table_lds_bytes(A.sample_rate) : 0;
    if (table_bytes && table_bytes + A.wave_bytes > budget) table_bytes = 0;  // buffers first; lookups fall back to L2
    if (A.wave_bytes > budget) return hipErrorInvalidValue;                   // plan_wave() guards this
    A.table_bytes = (uint32_t)table_bytes;
    if (!table_bytes) A.lds_table_id = -1;
    // waves per workgroup: as many as LDS holds next to the table image (they share it and hide each other's
    // scan / lookup latency), but no more than needed to give every CU a workgroup
    const int fit = (int)((budget - table_bytes) / A.wave_bytes);
    const unsigned want = (A.n_inst + 255) / 256;  // instances per CU on a 256-CU part
    int waves = 1;
    while (waves < 16 && waves * 2 <= fit && (unsigned)waves < want) waves *= 2;
    const size_t lds_bytes = table_bytes + (size_t)waves * A.wave_bytes;
#define DUSP_W(T, W) launch_wave_one<T, W>(A, lds_bytes, stream)
    if (table_bytes) {
        switch (waves) {
        case 16: return DUSP_W(1, 16);
        case 8: return DUSP_W(1, 8);
        case 4: return DUSP_W(1, 4);
        case 2: return DUSP_W(1, 2);
        default: return DUSP_W(1, 1);
        }
    }
    switch (waves) {
    case 16: return DUSP_W(0, 16);
    case 8: return DUSP_W(0, 8);
    case 4: return DUSP_W(0, 4);
    case 2: return DUSP_W(0, 2);
    default: return DUSP_W(0, 1);
    }
#undef DUSP_W
}

}  // namespace dusp
