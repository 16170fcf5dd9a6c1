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

#include <cstdlib>

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
            lastF = ft;
            double kf[5];
            butterworth_coefficients(L.filter.attr == 0 ? 0 : 1, ft, srd, kf);  // Filter.js:66-84 (filter_lamda.hpp: shared by every engine)
            a0 = kf[0]; a1 = kf[1]; a2 = kf[2]; b1 = kf[3]; b2 = kf[4];
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

// ---------------------------------------------------------------------------------------------------------
// Wide variant (the one configs[3] runs on): ONE workgroup of 8 waves per CU owns kWI = 32 instances and keeps
//   * the sine half-table in LDS (Table<1>, 99 KB) — a global gather of 2 taps per sample is TA-bound and was
//     ~40 % of the narrow kernel's time;
//   * one f32 tile row per instance that changes meaning through the chunk: Filter's previous output (read by
//     stage A as the feedback, and copied out) -> delayed Sum signal (written by stage A) -> Filter's output.
// Stage B is the serial part: 32 recurrences run side by side on the lanes of wave 0, but a wave issues one
// instruction at a time whatever its lane count (~4-6 cycles each on gfx950, tools/latbench.hip), so what matters
// is the number of instructions per SAMPLE on that wave.  It therefore executes nothing but the recurrence —
// multiply, multiply, two subtracts and the f64->f32->f64 round trip (the `|| 0` selects are speculated away, see
// below) — while six other waves compute the feed-forward half P[t] = (a0 x[t] + a1 x[t-1]) + a2 x[t-2] lane per
// sample, two sub-blocks of 32 samples AHEAD, into a ring of three f64 blocks; wave 0 pulls the next sub-block's
// P values into registers while it works through the current one (an LDS read costs ~100 cycles and there is
// nothing else on that wave to hide it behind).  All barriers in the chunk loop order LDS traffic only, so ring /
// PCM stores and the next chunk's ring prefetch stay in flight across them.
// Stage A keeps its per-instance constants in scalar registers (readfirstlane), indexes the ring in 32 bits with
// 16-byte accesses when the geometry allows, and short-cuts the taps of an integer delay.
// Measured on configs[3] (8192 loops x 10 s): 20.2 ms — roughly half stage A's instruction issue (~390 instructions
// per instance-chunk per wave: Osc lerp and Delay taps in f64, 64-bit phase arithmetic) and half the recurrence
// (~35-48 cycles per sample on one wave); the narrow kernel above takes 33 ms.
// Tried and dropped: pipelining the recurrence against stage A of the NEXT chunk at sub-block granularity (stage A of
// sub-block j of chunk c+1 needs only sub-block j of chunk c).  It was bit-exact but no faster (21.7 ms): the CU executes
// ~17 500 vector instructions per chunk (stage A ~12 500, feed-forward ~3 000, recurrence ~2 000) at one instruction per
// ~4.5 cycles per SIMD, i.e. ~20 000 cycles however they are arranged — the kernel is bound by total instruction
// issue, not by the recurrence's latency.  What would help is a leaner stage A (~98 instructions per sample now).
namespace {
constexpr int kWI = 32;        // instances per workgroup
constexpr int kWWaves = 8;     // waves per workgroup (2 per SIMD: 256 VGPRs each, so stage B keeps a sub-block's 32 P values in registers)
constexpr int kRow = 260;      // floats per instance row (16-byte aligned rows)
constexpr int kSub = 32;       // samples per stage-B sub-block
constexpr int kPPitch = 33;    // doubles per instance row of a P block (odd: lanes = instances hit distinct banks)

// Barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory queue (vmcnt(0)), i.e. it waits
// for every ring / PCM store of stage A to be acknowledged and for prefetches to land — microseconds that belong behind
// stage B.  Nothing in the chunk loop hands GLOBAL data from one wave to another (a ring row is read and written by
// the same wave, PCM is write-only), so only the LDS tiles need the fence.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

struct WideCarry {             // per-instance state stage A carries from chunk to chunk (LDS)
    unsigned long long phase;  // Osc phase of the previous chunk's last sample, 2^-36 units
    long long q;               // Osc increment f, 2^-36 units
    unsigned long long qm;     // q mod S (non-negative): sample-to-sample steps need one wrap test only
    double xprev;              // Sum output at the previous chunk's last sample (Delay's ceil tap)
    double phi;                // fractional part of the delay
    uint32_t D, pad2;          // integer part of the delay (256 <= D, D + 256 <= len)
    double a0, a1, a2;         // feed-forward biquad coefficients (published by stage B's lane at start)
    float xd[2][2];            // [chunk parity][0: x[-1], 1: x[-2]]: the delayed signal's last two samples of the chunk before
    float gain, delay_f;
    int bad, pad;
};
}  // namespace

__global__ void __launch_bounds__(kWWaves * 64) dusp_loop3_kernel(ChunkArgs a, LoopShape L) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float *half = (float *)lds_raw;
    float *tile = (float *)(lds_raw + half_table_lds_bytes(a.sample_rate));  // [kWI][kRow]
    double *pblk = (double *)(tile + kWI * kRow);                             // [3][kWI][kPPitch]
    WideCarry *carry = (WideCarry *)(pblk + 3 * kWI * kPPitch);

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t inst0 = blockIdx.x * kWI;
    const size_t NP = a.n_pad;
    const uint32_t sr = a.sample_rate;
    const double srd = (double)sr;
    const unsigned long long S = (unsigned long long)sr << 36;
    const double inv_S = 1.0 / (double)S;
    const unsigned long long lift = S * ((1ull << 62) / S);
    const int64_t len = L.delay.ring_len;
    load_half_table<kWWaves * 64>(half, a.tables + (size_t)L.osc.attr * a.table_stride, sr);
    Table<1> T;
    T.g = nullptr;
    T.h = half;
    T.N = sr + 1;
    T.M = sr / 2;

    auto lane_const = [&](const DevOperand &o, uint32_t i) {
        return o.kind == SRC_PARAM ? (i < a.n_inst ? a.params[(size_t)o.idx * a.n_inst + i] : 0.f) : o.cval;
    };

    // ---- start-up: stage B's lanes own the recurrence state and publish the feed-forward coefficients
    double b1 = 0, b2 = 0, y1 = 0, y2 = 0, lastF = 0;
    const bool b_lane = wave == 0 && lane < kWI;
    if (b_lane) {
        const uint32_t i = min(inst0 + lane, a.n_inst - 1);
        const double *st = a.state + (size_t)L.filter.state_slot * NP + i;
        WideCarry c;
        c.phase = (unsigned long long)(a.state[(size_t)L.osc.state_slot * NP + i] * kTwo36L);
        double fd = (double)lane_const(L.osc.in[0], i);
        c.bad = !(fabs(fd) <= 3.0e38);
        if (c.bad) fd = 0.0;
        if (fabs(fd) >= srd) fd = fmod(fd, srd);
        c.q = (long long)(fd * kTwo36L);
        c.xprev = a.state[(size_t)L.delay.state_slot * NP + i];
        c.qm = c.q < 0 ? (unsigned long long)(c.q + (long long)S) : (unsigned long long)c.q;
        c.gain = lane_const(L.mul.in[L.mul_gain_operand], i);
        c.delay_f = lane_const(L.delay.in[1], i);
        {
            double dconst = (double)c.delay_f;  // Delay.js:30: tWrite = (tBuffer + delay) % len
            if (dconst >= (double)len) dconst = fmod(dconst, (double)len);
            const double Dfl = floor(dconst);
            c.phi = dconst - Dfl;
            c.D = (uint32_t)Dfl;
        }
        c.pad = c.pad2 = 0;
        // Filter.js:34-37 with a constant f: coefficients are (re)computed at the first sample iff f != lastF
        const bool has_lastF = st[0] != 0.0;
        const double ft = (double)lane_const(L.filter.in[1], i);
        double a0 = st[2 * NP], a1 = st[3 * NP], a2 = st[4 * NP];
        b1 = st[5 * NP];
        b2 = st[6 * NP];
        lastF = st[NP];
        if (!has_lastF || ft != lastF) {
            lastF = ft;
            double kf[5];
            butterworth_coefficients(L.filter.attr == 0 ? 0 : 1, ft, srd, kf);  // Filter.js:66-84 (filter_lamda.hpp: shared by every engine)
            a0 = kf[0]; a1 = kf[1]; a2 = kf[2]; b1 = kf[3]; b2 = kf[4];
        }
        c.a0 = a0; c.a1 = a1; c.a2 = a2;
        // x1 / x2 are f32-valued (they were read out of Float32Arrays, Filter.js:47-48)
        c.xd[1][0] = (float)st[7 * NP];
        c.xd[1][1] = (float)st[8 * NP];
        c.xd[0][0] = c.xd[0][1] = 0.f;
        y1 = st[9 * NP];
        y2 = st[10 * NP];
        carry[lane] = c;
    }
    for (int k = threadIdx.x; k < kWI * kRow; k += kWWaves * 64) tile[k] = 0.f;  // Filter.out starts as zeros (SignalChunk.js:7)
    __syncthreads();

    // feed-forward half of the biquad for the 32 samples of sub-block `sb`, all instances: lane per (instance, sample)
    // (n_threads == 0: called by the helper waves of stage B, wave & 3 != 0; otherwise by the first n_threads threads)
    auto feed_forward = [&](uint32_t ck, int sb, uint32_t first_thread, uint32_t n_threads) {
        double *dst = pblk + (size_t)(sb % 3) * kWI * kPPitch;
        uint32_t k0 = threadIdx.x - first_thread;
        if (n_threads == 0) {
            k0 = ((wave >> 2) * 3 + (wave & 3u) - 1) * 64 + lane;  // helper rank
            n_threads = (kWWaves / 4) * 3 * 64;
        }
        for (uint32_t k = k0; k < (uint32_t)(kWI * kSub); k += n_threads) {
            const uint32_t i = k >> 5, s = k & 31u;
            const int t = sb * kSub + (int)s;
            const WideCarry &c = carry[i];
            const float *row = tile + i * kRow;
            const float *old = c.xd[(ck + 1) & 1];  // what the previous chunk left
            const double x0 = (double)row[t];
            const double xm1 = (double)(t >= 1 ? row[t - 1] : old[0]);
            const double xm2 = (double)(t >= 2 ? row[t - 2] : old[1 - t]);
            dst[i * kPPitch + s] = (c.a0 * x0 + c.a1 * or0d(xm1)) + c.a2 * or0d(xm2);  // Filter.js:40-42
        }
    };

    uint32_t s0 = (uint32_t)(a.clock0 % len);  // ring slot of the chunk's first sample (ring lengths are below 2^31)
    const uint32_t len32 = (uint32_t)len;
    auto uniform32 = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    auto uniform64 = [&](unsigned long long v) { return ((unsigned long long)uniform32((uint32_t)(v >> 32)) << 32) | uniform32((uint32_t)v); };
    // The Delay's ring reads of a chunk are issued one chunk ahead (right after the previous stage A, whose ring
    // writes — the latest this chunk can depend on, because D >= 256 — precede them), so their HBM / MALL latency
    // hides behind stage B instead of sitting at the head of stage A.  A lane's four slots are one 16-byte access
    // when the ring geometry keeps them contiguous and aligned (configs[3]: len 4096, D 480).
    f32x4 ahead[kWI / kWWaves];
    auto fetch_ring = [&](uint32_t first_slot) {
#pragma unroll
        for (int jj = 0; jj < kWI / kWWaves; ++jj) {
            const uint32_t inst = min(inst0 + wave + jj * kWWaves, a.n_inst - 1);
            const float *ring = a.rings + (size_t)inst * (size_t)len;
            uint32_t s_ = first_slot + lane * 4;
            if ((len32 & 3u) == 0 && (first_slot & 3u) == 0) {
                if (s_ >= len32) s_ -= len32;
                ahead[jj] = *(const f32x4 *)(ring + s_);
            } else {
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    uint32_t sc = s_ + cc;
                    if (sc >= len32) sc -= len32;
                    ahead[jj][cc] = ring[sc];
                }
            }
        }
    };
    fetch_ring(s0);
    for (uint32_t ck = 0; ck < a.n_chunks; ++ck) {
        // ------------------------------------------------------------------ stage A (and the previous chunk's copy-out)
        // Per-instance constants are wave-uniform: pulled into scalar registers so that the address and wrap
        // arithmetic built on them runs on the scalar unit and the branches on them do not touch the exec mask.
#pragma unroll
        for (int jj = 0; jj < kWI / kWWaves; ++jj) {
            const uint32_t j = wave + jj * kWWaves;
            const uint32_t inst = inst0 + j;
            if (inst >= a.n_inst) break;  // wave-uniform
            f32x4 *cell = (f32x4 *)&tile[j * kRow + lane * 4];
            const f32x4 fbv = *cell;  // Filter's output of the previous chunk
            if (ck > 0) {             // stage C of chunk ck-1: copy-out (renderChannelData.js:35-44)
                float v[4] = {fix_out<false>(fbv[0]), fix_out<false>(fbv[1]), fix_out<false>(fbv[2]), fix_out<false>(fbv[3])};
                const uint64_t n0 = (uint64_t)(ck - 1) * kChunk + lane * 4;
                float *row = a.out + (size_t)inst * a.n_samples + n0;
                if ((a.n_samples & 3) == 0 && n0 + 4 <= a.n_samples) store4<true>(row, v, n0, a.n_samples);
                else store4<false>(row, v, n0, a.n_samples);
            }
            const WideCarry &cr = carry[j];
            const unsigned long long phase0 = uniform64(cr.phase), qm = uniform64(cr.qm);
            const long long q = (long long)uniform64((unsigned long long)cr.q);
            const float gain = __uint_as_float(uniform32(__float_as_uint(cr.gain)));
            const uint32_t D = uniform32(cr.D);
            const double phi = __longlong_as_double((long long)uniform64((unsigned long long)__double_as_longlong(cr.phi)));
            const double xprev_chunk = __longlong_as_double((long long)uniform64((unsigned long long)__double_as_longlong(cr.xprev)));
            const bool bad = uniform32((uint32_t)cr.bad) != 0;
            float *ring = a.rings + (size_t)inst * (size_t)len;
            // Osc (Osc.js:35-47): exact fixed-point phase of this lane's 4 samples
            unsigned long long P = mod_u64_lifted((unsigned long long)((long long)phase0 + q * (long long)(lane * 4 + 1)) + lift, S, inv_S);
            float x[4];  // Sum output
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                if (cc > 0) {
                    P += qm;  // P, qm < S: one wrap test
                    if (P >= S) P -= S;
                }
                const uint32_t idx = (uint32_t)(P >> 36);
                // fraction = (P mod 2^36) / 2^36, exactly: 4 high bits and 32 low bits are converted separately
                const double fraction =
                    ldexp(fma((double)((uint32_t)(P >> 32) & 15u), 4294967296.0, (double)(uint32_t)P), -36);
                float ta, tb;
                T.pair(idx, ta, tb);
                const float osc = (float)((double)ta * (1.0 - fraction) + (double)tb * fraction);
                const float fb = L.mul_gain_operand ? fbv[cc] * gain : gain * fbv[cc];  // Multiply.js:31
                x[cc] = L.sum_osc_operand ? fb + osc : osc + fb;                          // Sum.js:42
            }
            if (bad)  // f is NaN / Inf: the reference's phase is NaN for good and every table read undefined
                for (int cc = 0; cc < 4; ++cc) x[cc] = __builtin_nanf("");
            const unsigned long long lastP = __shfl(P, 63, 64);
            // Delay (Delay.js:27-39), constant delay D + phi with D >= 256: every slot gets the ceil tap of sample
            // n-1 and the floor tap of sample n, with the reference's two `+=` roundings, and is written once
            const f32x4 delayed = ahead[jj];
            float x_left = __shfl_up(x[3], 1, 64);  // Sum output of sample 4l-1
            if (lane == 0) x_left = (float)xprev_chunk;
            uint32_t lo0 = s0 + lane * 4 + D;  // slot of this lane's first write (< 3 len)
            if (lo0 >= len32) lo0 -= len32;
            if (lo0 >= len32) lo0 -= len32;
            float slot[4];
            if (phi == 0.0) {  // floor(tWrite) == ceil(tWrite): (0 + x*1) + x*0 — x itself, with -0 -> +0 and Inf -> NaN
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) slot[cc] = fabsf(x[cc]) < __builtin_inff() ? x[cc] + 0.f : __builtin_nanf("");
            } else {
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    uint32_t lo = lo0 + cc;
                    if (lo >= len32) lo -= len32;
                    const double xin = (double)x[cc];
                    const double xprev = (double)(cc == 0 ? x_left : x[cc - 1]);
                    const float ceil_tap = (float)(0.0 + xprev * phi);  // of sample n-1; dropped at slot 0 (no wrap, Delay.js:38)
                    slot[cc] = (float)((double)(lo != 0 ? ceil_tap : 0.f) + xin * (1.0 - phi));  // floor tap of sample n
                }
            }
            if ((len32 & 3u) == 0 && ((s0 + D) & 3u) == 0) {
                *(f32x4 *)(ring + lo0) = f32x4{slot[0], slot[1], slot[2], slot[3]};
            } else {
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    uint32_t lo = lo0 + cc;
                    if (lo >= len32) lo -= len32;
                    ring[lo] = slot[cc];
                }
            }
            *cell = delayed;  // the Filter's input takes the row over
            if (lane == 63) {
                carry[j].phase = lastP;
                carry[j].xprev = (double)x[3];
                carry[j].xd[ck & 1][0] = delayed[3];  // for the next chunk's first two samples
                carry[j].xd[ck & 1][1] = delayed[2];
            }
        }
        lds_barrier();
        // ------------------------------------------------------------------ stage B: the biquad, software-pipelined
        // Helpers run TWO sub-blocks ahead of the recurrence (three P blocks in LDS), so that wave 0 can pull the next
        // sub-block's 32 P values into registers while it works through the current one: an LDS read costs ~100
        // cycles, and there is nothing on this wave to hide it behind except the chain itself.
        feed_forward(ck, 0, 0, kWWaves * 64);
        feed_forward(ck, 1, 0, kWWaves * 64);
        lds_barrier();
        double pa[kSub], pb[kSub];
        const bool b_live = b_lane && inst0 + lane < a.n_inst;
        auto load_p = [&](int sb, double (&dst)[kSub]) {
            const double *prow = pblk + (size_t)(sb % 3) * kWI * kPPitch + lane * kPPitch;
#pragma unroll
            for (int k = 0; k < kSub; ++k) dst[k] = prow[k];
        };
        auto recur = [&](int sb, const double (&pv)[kSub]) {
            f32x4 *row = (f32x4 *)&tile[lane * kRow + sb * kSub];
            // Fast pass without the `|| 0` selects (Filter.js:42-46 reads `y1 || 0`, `y2 || 0`): they only matter when
            // an output is NaN — a -0 instead of +0 can change nothing but the sign of a later zero, and every consumer
            // of this signal maps -0 to +0 (copy-out `|| 0`, the ring's `0 + ...`, the next `|| 0`).  Without the
            // selects a NaN never leaves the recurrence again, so testing the sub-block's LAST output detects one
            // anywhere in it; such a sub-block is redone exactly.
            const double y1_in = y1, y2_in = y2;
            double u1 = or0d(y1), u2 = or0d(y2);
#pragma unroll
            for (int k = 0; k < kSub / 4; ++k) {
                f32x4 y4;
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const float y = (float)((pv[k * 4 + cc] - b1 * u1) - b2 * u2);
                    y4[cc] = y;
                    u2 = u1;
                    u1 = (double)y;
                }
                row[k] = y4;
            }
            if (u1 == u1 && u2 == u2) {
                y2 = u2;  // = y2 || 0 up to the sign of a zero
                y1 = u1;
            } else {
                y1 = y1_in;
                y2 = y2_in;
#pragma unroll
                for (int k = 0; k < kSub / 4; ++k) {
                    f32x4 y4;
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const float y = (float)((pv[k * 4 + cc] - b1 * or0d(y1)) - b2 * or0d(y2));  // Filter.js:42-44
                        y4[cc] = y;
                        y2 = or0d(y1);   // :45
                        y1 = (double)y;  // :46 (reads the f32-rounded sample back)
                    }
                    row[k] = y4;
                }
            }
        };
        if (b_live) load_p(0, pa);
        for (int sb = 0; sb < kChunk / kSub; sb += 2) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int cur = sb + half;
                if (cur == 2 && ck + 1 < a.n_chunks) {  // next chunk's ring reads: stage A's stores have long landed by now
                    uint32_t nxt = s0 + kChunk;
                    if (nxt >= len32) nxt -= len32;
                    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's ring stores precede its ring loads
                    fetch_ring(nxt);
                }
                if (wave == 0) {
                    if (b_live) {
                        __builtin_amdgcn_s_setprio(3);  // the chunk's critical path
                        if (cur + 1 < kChunk / kSub) {  // (computed a step ago)
                            if (half == 0) load_p(cur + 1, pb);
                            else load_p(cur + 1, pa);
                        }
                        if (half == 0) recur(cur, pa);
                        else recur(cur, pb);
                        __builtin_amdgcn_s_setprio(0);
                    }
                } else if ((wave & 3u) != 0 && cur + 2 < kChunk / kSub) {
                    // helpers: the waves that do not share wave 0's SIMD (waves are dealt round-robin over the 4 SIMDs)
                    feed_forward(ck, cur + 2, 0, 0);
                }
                lds_barrier();
            }
        }
        s0 += kChunk;
        if (s0 >= len32) s0 -= len32;
    }
    // ---- copy-out of the last chunk
    for (uint32_t j = wave; j < kWI && a.n_chunks > 0; j += kWWaves) {
        const uint32_t inst = inst0 + j;
        if (inst >= a.n_inst) break;
        const f32x4 yl = *(const f32x4 *)&tile[j * kRow + lane * 4];
        float v[4] = {fix_out<false>(yl[0]), fix_out<false>(yl[1]), fix_out<false>(yl[2]), fix_out<false>(yl[3])};
        const uint64_t n0 = (uint64_t)(a.n_chunks - 1) * kChunk + lane * 4;
        float *row = a.out + (size_t)inst * a.n_samples + n0;
        if ((a.n_samples & 3) == 0 && n0 + 4 <= a.n_samples) store4<true>(row, v, n0, a.n_samples);
        else store4<false>(row, v, n0, a.n_samples);
    }
    // ---- state write-back, in the chunk engine's slot layout
    if (b_lane && inst0 + lane < a.n_inst) {
        const uint32_t i = inst0 + lane;
        const WideCarry c = carry[lane];
        a.state[(size_t)L.osc.state_slot * NP + i] = c.bad ? __builtin_nan("") : (double)c.phase * (1.0 / kTwo36L);
        a.state[(size_t)L.delay.state_slot * NP + i] = c.xprev;
        double *st = a.state + (size_t)L.filter.state_slot * NP + i;
        const float *xl = c.xd[(a.n_chunks + 1) & 1];  // what the last chunk left (the initial state if nothing was rendered)
        st[0] = 1.0;
        st[NP] = lastF; st[2 * NP] = c.a0; st[3 * NP] = c.a1; st[4 * NP] = c.a2; st[5 * NP] = b1; st[6 * NP] = b2;
        st[7 * NP] = (double)xl[0]; st[8 * NP] = or0d((double)xl[1]); st[9 * NP] = y1; st[10 * NP] = y2;
    }
}

hipError_t launch_loop2_engine(const ChunkArgs &a, const LoopShape &L, bool lds_table_ok, bool wide, hipStream_t stream) {
    if (lds_table_ok && wide) {  // (wide: Knobs::loop_wide)
        const size_t lds = half_table_lds_bytes(a.sample_rate) + (size_t)kWI * kRow * sizeof(float) + (size_t)3 * kWI * kPPitch * sizeof(double) +
                           (size_t)kWI * sizeof(WideCarry);
        hipError_t e = hipFuncSetAttribute((const void *)dusp_loop3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(dusp_loop3_kernel, dim3((a.n_inst + kWI - 1) / kWI), dim3(kWWaves * 64), lds, stream, a, L);
        return hipGetLastError();
    }
    const uint32_t blocks = (a.n_inst + kIW - 1) / kIW;
    hipLaunchKernelGGL(dusp_loop2_kernel, dim3(blocks), dim3(256), 0, stream, a, L);
    return hipGetLastError();
}

}  // namespace dusp
