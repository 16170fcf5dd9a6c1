// loop2_engine.hip — two-stage kernel for the feedback voice of BASELINE configs[3]
//     sum = Sum(Osc(k), fb);  d = Delay(sum, D, maxDelay);  f = Filter(d, k);  fb = Multiply(f, k);  sum.B = fb
// when the delay is at least one chunk (D >= 256, D + 256 <= maxDelay).
//
// The loop engine (chunk_engine.hip) evaluates this voice lane-per-INSTANCE and is bound by one wave's
// instruction issue: ~100 dependent-issue instructions per sample, of which only the Filter's output
// recurrence (y depends on y1, y2) is inherently serial.  Here a workgroup owns 16 instances and splits
// each 256-sample chunk in two stages that hand the chunk over in LDS:
//
//   stage A — lane per SAMPLE (a wave takes one instance at a time, lane l = samples 4l..4l+3): Osc by exact
//     fixed-point phase (as in wave_engine.hip), feedback = Filter's PREVIOUS chunk x gain (the implicit
//     256-sample delay of the reference's process order), Sum, the Delay ring read and — because the delay
//     is >= a chunk, writes never land on slots this chunk still reads — the ring write of each slot's final
//     value (ceil tap of sample n-1, floor tap of sample n, with the reference's two `+=` roundings), and the
//     feed-forward half of the biquad, P[t] = (a0 x[t] + a1 x[t-1]) + a2 x[t-2] in f64 with the reference's
//     rounding order.  x[t-1], x[t-2] come from the neighbouring lane (DPP shuffles) or the chunk carry.
//   stage B — lane per INSTANCE (16 lanes of wave 0): y[t] = f32((P[t] - b1 y1) - b2 y2): 2 multiplies,
//     2 subtracts and the f32 round trip per sample — ~8 serial instructions instead of ~100.
//   stage C — lane per sample again: PCM rows leave as coalesced 1 KiB stores.
//
// Ring layout is [instance][slot] (a wave reads/writes 256 consecutive slots of one instance).
// Results are bit-identical to the chunk / loop engines in the oscillator's exact regime (|f| >= 2^-13).
#include <hip/hip_runtime.h>

#include "device_types.hpp"
#include "fused_device.hpp"
#include "fused_plan.hpp"

namespace dusp {

namespace {

constexpr int kIW = 16;            // instances per workgroup
constexpr int kPStride = 257;      // doubles per instance row of P (odd pitch: 16 lanes -> 32 distinct banks)
constexpr int kFStride = 260;      // floats per instance row of F (16-byte aligned rows)
constexpr double kTwo36L = 68719476736.0;

__device__ __forceinline__ double or0d(double v) { return (v != v || v == 0.0) ? 0.0 : v; }

struct InstCarry {                 // per-instance state that stage A carries from chunk to chunk (LDS)
    unsigned long long phase;      // Osc phase of the previous chunk's last sample, 2^-36 units
    long long q;                   // Osc increment f, 2^-36 units
    double xprev;                  // Sum output at the previous chunk's last sample (Delay's ceil tap)
    double xd1, xd2;               // delayed signal at t-1, t-2 (Filter's x1, x2)
    double a0, a1, a2;             // feed-forward biquad coefficients (set by stage B's lane at start)
    float gain, delay_f;
    int bad, pad;
};

}  // namespace

__global__ void __launch_bounds__(256) dusp_loop2_kernel(ChunkArgs a, LoopShape L) {
    __shared__ __attribute__((aligned(16))) double Pt[kIW * kPStride];
    __shared__ __attribute__((aligned(16))) float Ft[kIW * kFStride];
    __shared__ InstCarry carry[kIW];

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t inst0 = blockIdx.x * kIW;
    const size_t NP = a.n_pad;
    const uint32_t sr = a.sample_rate;
    const double srd = (double)sr;
    const unsigned long long S = (unsigned long long)sr << 36;
    const double inv_S = 1.0 / (double)S;
    const unsigned long long lift = S * ((1ull << 62) / S);
    const float *gtab = a.tables + (size_t)L.osc.attr * a.table_stride;
    const int64_t len = L.delay.ring_len;

    auto lane_const = [&](const DevOperand &o, uint32_t i) {
        return o.kind == SRC_PARAM ? (i < a.n_inst ? a.params[(size_t)o.idx * a.n_inst + i] : 0.f) : o.cval;
    };

    // ---- start-up: stage B's lanes own the filter state; they also publish the feed-forward coefficients
    double b1 = 0, b2 = 0, y1 = 0, y2 = 0, lastF = 0;
    const bool b_lane = wave == 0 && lane < kIW;
    if (b_lane) {
        const uint32_t i = min(inst0 + lane, a.n_inst - 1);
        const double *st = a.state + (size_t)L.filter.state_slot * NP + i;
        InstCarry c;
        c.phase = (unsigned long long)(a.state[(size_t)L.osc.state_slot * NP + i] * kTwo36L);
        double fd = (double)lane_const(L.osc.in[0], i);
        c.bad = !(fabs(fd) <= 3.0e38);
        if (c.bad) fd = 0.0;
        if (fabs(fd) >= srd) fd = fmod(fd, srd);
        c.q = (long long)(fd * kTwo36L);
        c.xprev = a.state[(size_t)L.delay.state_slot * NP + i];
        c.gain = lane_const(L.mul.in[L.mul_gain_operand], i);
        c.delay_f = lane_const(L.delay.in[1], i);
        // Filter.js:34-37 with a constant f: the coefficients are (re)computed at the first sample iff
        // f != lastF (or lastF is still undefined) and never again
        const bool has_lastF = st[0] != 0.0;
        const double ft = (double)lane_const(L.filter.in[1], i);
        double a0 = st[2 * NP], a1 = st[3 * NP], a2 = st[4 * NP];
        b1 = st[5 * NP];
        b2 = st[6 * NP];
        lastF = st[NP];
        if (!has_lastF || ft != lastF) {
            const double PI = 3.141592653589793;
            lastF = ft;
            if (L.filter.attr == 0) {  // LP (Filter.js:67-75)
                const double lamda = 1.0 / tan(PI * ft / srd);
                const double l2 = lamda * lamda;
                a0 = 1.0 / (1.0 + 2.0 * lamda + l2);
                a1 = 2.0 * a0;
                a2 = a0;
                b1 = 2.0 * a0 * (1.0 - l2);
                b2 = a0 * (1.0 - 2.0 * lamda + l2);
            } else {  // HP (Filter.js:76-84)
                const double lamda = tan(PI * ft / srd);
                const double l2 = lamda * lamda;
                a0 = 1.0 / (1.0 + 2.0 * lamda + l2);
                a1 = 0.0;
                a2 = -a0;
                b1 = 2.0 * a0 * (l2 - 1.0);
                b2 = a0 * (1.0 - 2.0 * lamda + l2);
            }
        }
        c.a0 = a0; c.a1 = a1; c.a2 = a2;
        c.xd1 = st[7 * NP];
        c.xd2 = st[8 * NP];
        y1 = st[9 * NP];
        y2 = st[10 * NP];
        c.pad = 0;
        carry[lane] = c;
    }
    for (int k = threadIdx.x; k < kIW * kFStride; k += 256) Ft[k] = 0.f;  // Filter.out starts as zeros (SignalChunk.js:7)
    __syncthreads();

    int64_t s0 = a.clock0 % len;  // ring slot of the chunk's first sample
    for (uint32_t ck = 0; ck < a.n_chunks; ++ck) {
        // ------------------------------------------------------------------ stage A
        for (uint32_t j = wave; j < kIW; j += 4) {
            const uint32_t inst = inst0 + j;
            if (inst >= a.n_inst) break;  // wave-uniform
            const InstCarry c = carry[j];
            float *ring = a.rings + (size_t)inst * (size_t)len;
            // Osc (Osc.js:35-47): exact fixed-point phase of this lane's 4 samples
            unsigned long long P = mod_u64_lifted((unsigned long long)((long long)c.phase + c.q * (long long)(lane * 4 + 1)) + lift, S, inv_S);
            float x[4];  // Sum output
            const f32x4 fbv = *(const f32x4 *)&Ft[j * kFStride + lane * 4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                if (cc > 0) {
                    long long Pn = (long long)P + c.q;
                    if (Pn < 0) Pn += (long long)S;
                    if (Pn >= (long long)S) Pn -= (long long)S;
                    P = (unsigned long long)Pn;
                }
                const uint32_t idx = (uint32_t)(P >> 36);
                const double fraction = (double)(P & ((1ull << 36) - 1ull)) * (1.0 / kTwo36L);
                const float ta = gtab[idx], tb = gtab[idx + 1];
                const float osc = c.bad ? __builtin_nanf("") : (float)((double)ta * (1.0 - fraction) + (double)tb * fraction);
                const float fb = L.mul_gain_operand ? fbv[cc] * c.gain : c.gain * fbv[cc];  // Multiply.js:31
                x[cc] = L.sum_osc_operand ? fb + osc : osc + fb;                              // Sum.js:42
            }
            const unsigned long long lastP = __shfl(P, 63, 64);
            // Delay (Delay.js:27-39), constant delay D + phi with D >= 256
            double dconst = (double)c.delay_f;
            if (dconst >= (double)len) dconst = fmod(dconst, (double)len);
            const double Dfl = floor(dconst), phi = dconst - Dfl;
            const int64_t D = (int64_t)Dfl;
            float delayed[4];
            float x_left = __shfl_up(x[3], 1, 64);  // Sum output of sample 4l-1
            if (lane == 0) x_left = (float)c.xprev;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                int64_t s_ = s0 + lane * 4 + cc;
                if (s_ >= len) s_ -= len;
                delayed[cc] = ring[s_];
            }
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                int64_t s_ = s0 + lane * 4 + cc;
                if (s_ >= len) s_ -= len;
                int64_t lo = s_ + D;
                if (lo >= len) lo -= len;
                const double xin = (double)x[cc];
                const double xprev = cc == 0 ? (lane == 0 ? c.xprev : (double)x_left) : (double)x[cc - 1];
                float slot;
                if (phi != 0.0) {
                    slot = lo != 0 ? (float)(0.0 + xprev * phi) : 0.f;  // ceil tap of sample n-1 (dropped at slot 0)
                    slot = (float)((double)slot + xin * (1.0 - phi));   // floor tap of sample n
                } else {
                    slot = (float)(0.0 + xin * 1.0);
                    slot = (float)((double)slot + xin * 0.0);
                }
                ring[lo] = slot;
            }
            // Filter, feed-forward half (Filter.js:40-42): ((a0 x + a1 (x1||0)) + a2 (x2||0)) in f64
            float d_l1 = __shfl_up(delayed[3], 1, 64), d_l2 = __shfl_up(delayed[2], 1, 64);
            double xm1 = lane == 0 ? c.xd1 : (double)d_l1, xm2 = lane == 0 ? c.xd2 : (double)d_l2;
            double p[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const double xin = (double)delayed[cc];
                p[cc] = (c.a0 * xin + c.a1 * or0d(xm1)) + c.a2 * or0d(xm2);
                xm2 = or0d(xm1);  // the reference stores x2 = x1 || 0, x1 = in (Filter.js:47-48)
                xm1 = xin;
            }
            double *prow = &Pt[j * kPStride + lane * 4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) prow[cc] = p[cc];
            if (lane == 63) {
                carry[j].phase = lastP;
                carry[j].xprev = (double)x[3];
                carry[j].xd1 = xm1;
                carry[j].xd2 = xm2;
            }
        }
        __syncthreads();
        // ------------------------------------------------------------------ stage B
        if (b_lane && inst0 + lane < a.n_inst) {
            const double *prow = &Pt[lane * kPStride];
            float *frow = &Ft[lane * kFStride];
#pragma unroll 8
            for (int t = 0; t < kChunk; ++t) {
                const float y = (float)((prow[t] - b1 * or0d(y1)) - b2 * or0d(y2));  // Filter.js:40-44
                frow[t] = y;
                y2 = or0d(y1);  // :45
                y1 = (double)y; // :46 (reads the f32-rounded sample back)
            }
        }
        __syncthreads();
        // ------------------------------------------------------------------ stage C: copy-out (renderChannelData.js:35-44)
        for (uint32_t j = wave; j < kIW; j += 4) {
            const uint32_t inst = inst0 + j;
            if (inst >= a.n_inst) break;
            const f32x4 yv = *(const f32x4 *)&Ft[j * kFStride + lane * 4];
            float v[4] = {fix_out<false>(yv[0]), fix_out<false>(yv[1]), fix_out<false>(yv[2]), fix_out<false>(yv[3])};
            const uint64_t n0 = (uint64_t)ck * kChunk + lane * 4;
            float *row = a.out + (size_t)inst * a.n_samples + n0;
            if ((a.n_samples & 3) == 0 && n0 + 4 <= a.n_samples) store4<true>(row, v, n0, a.n_samples);
            else store4<false>(row, v, n0, a.n_samples);
        }
        s0 += kChunk;
        if (s0 >= len) s0 -= len;
        // (no barrier here: the next write to Ft is stage B of the next chunk, behind the next A->B barrier)
    }
    __syncthreads();
    // ---- state write-back, in the chunk engine's slot layout
    if (b_lane && inst0 + lane < a.n_inst) {
        const uint32_t i = inst0 + lane;
        const InstCarry c = carry[lane];
        a.state[(size_t)L.osc.state_slot * NP + i] = c.bad ? __builtin_nan("") : (double)c.phase * (1.0 / kTwo36L);
        a.state[(size_t)L.delay.state_slot * NP + i] = c.xprev;
        double *st = a.state + (size_t)L.filter.state_slot * NP + i;
        st[0] = 1.0;
        st[NP] = lastF; st[2 * NP] = c.a0; st[3 * NP] = c.a1; st[4 * NP] = c.a2; st[5 * NP] = b1; st[6 * NP] = b2;
        st[7 * NP] = c.xd1; st[8 * NP] = c.xd2; st[9 * NP] = y1; st[10 * NP] = y2;
    }
}

hipError_t launch_loop2_engine(const ChunkArgs &a, const LoopShape &L, hipStream_t stream) {
    const uint32_t blocks = (a.n_inst + kIW - 1) / kIW;
    hipLaunchKernelGGL(dusp_loop2_kernel, dim3(blocks), dim3(256), 0, stream, a, L);
    return hipGetLastError();
}

}  // namespace dusp
