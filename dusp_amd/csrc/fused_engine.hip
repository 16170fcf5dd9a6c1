// fused_engine.hip — fused, time-parallel render kernels for feed-forward voices.
//
// One LANE per SAMPLE: a wavefront renders 256 consecutive samples of one
// circuit instance per step (lane l owns samples 4l..4l+3 of the group), so the
// PCM leaves as one fully coalesced 1 KiB `global_store_dwordx4` per wave per
// step and nothing is staged through HBM.  The whole unit tree of the circuit
// is evaluated in registers — this is the "one fused kernel per topologically
// sorted Circuit" of the north star for the shapes listed in fused_plan.hpp.
//
// Time is split into segments so that a 1024-voice x 60 s render exposes ~10^5
// independent (instance, segment) work items instead of 1024 serial voices.
// That is legal because every stateful unit in these shapes can JUMP to any
// sample index exactly:
//   * Osc with a lane-constant f (Osc.js:38-46): phase(t) = (phase0 + (t+1) f) mod sr.
//     The reference accumulates in f64; for |f| >= 2^-13 every partial sum is an
//     exact multiple of lsb(f) below 2^17, so no rounding ever happens
//     (SURVEY.md §8a note ii) and the closed form — evaluated in exact u64
//     fixed-point modular arithmetic — is bit-identical to the sequential loop.
//     Inside a segment each lane then advances its own f64 phase by
//     (256 f) mod sr per step with one conditional subtract: also exact.
//     When f and phase0 are integers every phase is an integer, the lerp
//     degenerates to table[phase] and a pure u32 path is taken (wave-uniform).
//   * Ramp (Ramp.js:25-40): t(n) = min(t0 + n + 1, duration) while playing.
// Wave-table lookups come either from L2 (global gather) or, when the table is
// antisymmetric (T[N-t] == -T[t], true for the sine table), from a 96 KB
// half-table held in LDS — the full 192 KB table does not fit the 160 KB LDS.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "device_types.hpp"
#include "fused_plan.hpp"

namespace dusp {

namespace {

__device__ __forceinline__ int lsb_exponent(double x) {  // x finite, != 0: exponent of its lowest set bit
    int ex;
    const double fr = frexp(fabs(x), &ex);
    const long long m = (long long)ldexp(fr, 53);
    return ex - 53 + __builtin_ctzll((unsigned long long)m);
}
__device__ __forceinline__ uint64_t addmod(uint64_t a, uint64_t b, uint64_t S) {  // a, b < S < 2^63
    const uint64_t s = a + b;
    return s >= S ? s - S : s;
}
__device__ __forceinline__ uint64_t mulmod(uint64_t a, uint64_t n, uint64_t S) {  // a < S
    uint64_t acc = 0;
    for (int bit = 63 - __builtin_clzll(n | 1); bit >= 0; --bit) {
        acc = addmod(acc, acc, S);
        if ((n >> bit) & 1) acc = addmod(acc, a, S);
    }
    return acc;
}

struct OscFix {
    uint64_t S, Fm, P0;  // modulus, per-sample increment and start phase in units of u = 2^E
    double u;
    int E;
    bool bad;            // non-finite f: every sample is NaN -> 0 after `|| 0`
};

__device__ __forceinline__ OscFix osc_setup(float f, double phase0, uint32_t sr) {
    OscFix o;
    double fd = (double)f;
    const double srd = (double)sr;
    o.bad = !(fabs(fd) <= 3.0e38);
    if (o.bad) fd = 0.0;
    if (fabs(fd) >= srd) fd = fmod(fd, srd);
    int E = 0;
    if (fd != 0.0) E = min(E, lsb_exponent(fd));
    if (phase0 != 0.0) E = min(E, lsb_exponent(phase0));
    const int Emin = -(62 - (32 - __builtin_clz(sr)));  // keep S = sr * 2^-E below 2^62
    if (E < Emin) E = Emin;                              // (only reachable for |f| < 2^-13: inexact regime)
    o.E = E;
    o.u = ldexp(1.0, E);
    o.S = (uint64_t)sr << (-E);
    const long long F = (long long)rint(ldexp(fd, -E));
    o.Fm = F >= 0 ? (uint64_t)F : o.S - (uint64_t)(-F);
    if (o.Fm >= o.S) o.Fm -= o.S;
    o.P0 = (uint64_t)rint(ldexp(phase0, -E));
    if (o.P0 >= o.S) o.P0 %= o.S;
    return o;
}

__device__ __forceinline__ float operand_value(const DevOperand &o, const float *params, uint32_t n_inst, uint32_t inst) {
    return o.kind == SRC_PARAM ? params[(size_t)o.idx * n_inst + inst] : o.cval;
}

// Table access.  TBL == 0: padded full table in global memory (served by L2).
// TBL == 1: half table H[0..M+1] = T[0..M+1] in LDS (M = sr/2, N = sr+1); T[i] = -H[N-i] above M.
//   LDS layout: blocks of 33 words, block b = H[32b .. 32b+32] (the 33rd word repeats the next
//   block's first).  position(h) = h + (h >> 5).  The odd block pitch spreads the arithmetic
//   progressions a wave reads (lane l looks up phase0 + 4 l f) over the 32 banks — a linear layout
//   measured 9-way conflicts on average for the 1024-voice sweep, this one 2.8 — and position+1
//   always holds H[h+1], so the lerp's pair is one ds_read2_b32.
template <int TBL>
struct Table {
    const float *g;
    const float *h;
    uint32_t N, M;
    __device__ __forceinline__ float lds_at(uint32_t k) const {  // word k of the 33-pitch image
        return *(const float *)((const char *)h + ((k << 2) + ((k >> 5) << 2)));
    }
    __device__ __forceinline__ float at(uint32_t i) const {
        if (TBL == 0) return g[i];
        const float v = lds_at(min(i, N - i));
        return i > M ? -v : v;
    }
    // (T[i], T[i+1]) for the lerp
    __device__ __forceinline__ void pair(uint32_t i, float &a, float &b) const {
        if (TBL == 0) {
            a = g[i];
            b = g[i + 1];
            return;
        }
        const bool upper = i > M;
        const uint32_t k = upper ? N - i - 1 : i;
        const float *p = (const float *)((const char *)h + ((k << 2) + ((k >> 5) << 2)));
        const float x = p[0], y = p[1];
        a = upper ? -y : x;
        b = upper ? -x : y;
    }
};

// y0 + (t/duration) * (y1-y0) with t = min(tn, duration)  (Ramp.js:27-38)
template <bool FASTDIV>
__device__ __forceinline__ float ramp_value(const FusedArgs &A, double tn, double dy) {
    const double tt = fmin(tn, A.r_d);
    double q;
    if (FASTDIV) {  // host-verified to equal tt / duration on this Ramp's whole t sequence (fused_plan.hpp)
        q = tt * A.r_rcp;
        const double rem = fma(-q, A.r_d, tt);
        q = fma(rem, A.r_rcp, q);
    } else
        q = tt / A.r_d;
    return (float)(A.r_y0 + q * dy);
}

template <bool FINITE>
__device__ __forceinline__ float fix_out(float v) {  // `x || 0` (renderChannelData.js:44): NaN, -0 -> +0
    if (FINITE) return v + 0.f;                       // operands verified finite on the host: only -0 can occur
    return (v != v) ? 0.f : v + 0.f;
}

template <bool VEC>
__device__ __forceinline__ void store4(float *row, const float (&v)[4], uint64_t t, uint64_t n_samples) {
    if (VEC)
        __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, (f32x4 *)row);
    else
        for (int c = 0; c < 4; ++c)
            if (t + c < n_samples) row[c] = v[c];
}

}  // namespace

// KIND: fused shape.  TBL: table placement.  R: instances per wave (the Ramp, which depends only on
// the sample index, is evaluated once per step and reused for all R voices).  FASTDIV / FINITE: host-
// verified strength reductions (see ramp_value / fix_out).
template <int KIND, int TBL, int R, bool FASTDIV, bool FINITE, int BLOCK>
__global__ void __launch_bounds__(BLOCK) dusp_fused_kernel(FusedArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds_table[];
    Table<TBL> table;
    table.g = A.table;
    table.h = lds_table;
    table.N = A.sample_rate + 1;
    table.M = A.sample_rate / 2;
    if (TBL == 1) {
        const uint32_t last = table.M + 1;
        const uint32_t n_words = last + (last >> 5) + 2;
        for (uint32_t q = threadIdx.x; q < n_words; q += BLOCK) {
            const uint32_t src = (q / 33) * 32 + (q % 33);
            lds_table[q] = A.table[min(src, last)];
        }
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63u;
    constexpr uint32_t waves_per_block = BLOCK / 64;
    const uint32_t n_blk = (A.n_inst + R - 1) / R;
    const uint64_t n_items = (uint64_t)n_blk * A.n_seg;
    const uint64_t total_waves = (uint64_t)gridDim.x * waves_per_block;
    const double srd = (double)A.sample_rate;
    const uint32_t sr = A.sample_rate;
    const double dy = A.r_y1 - A.r_y0;
    // full 256-sample groups may use unguarded 16-byte stores; a trailing partial group may not
    const uint32_t n_full = A.vec4_ok ? (uint32_t)(A.n_samples / kChunk) : 0u;

    for (uint64_t item = (uint64_t)blockIdx.x * waves_per_block + (threadIdx.x >> 6); item < n_items; item += total_waves) {
        const uint32_t blk = (uint32_t)(item % n_blk);
        const uint32_t seg = (uint32_t)(item / n_blk);
        const uint32_t g0 = seg * A.seg_groups;
        const uint32_t g1 = min(g0 + A.seg_groups, A.n_groups);
        const uint64_t t_start = (uint64_t)g0 * kChunk;

        uint64_t P[R][4], step256[R];
        double u[R];
        float gain[R];
        bool bad[R];
        size_t roff[R];  // row offset of instance r; a short last block repeats its last instance (same data, same address)
        bool all_integer = true;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t inst = min(blk * R + r, A.n_inst - 1);
            const bool live = blk * R + r < A.n_inst;
            roff[r] = (size_t)inst * A.n_samples;
            const float f = operand_value(A.f, A.params, A.n_inst, inst);
            const OscFix o = osc_setup(f, A.phase0, sr);
            const uint64_t step4 = mulmod(o.Fm, 4, o.S);
            step256[r] = mulmod(o.Fm, 256, o.S);
            P[r][0] = addmod(addmod(o.P0, mulmod(o.Fm, t_start + 1, o.S), o.S), mulmod(step4, lane, o.S), o.S);
            for (int c = 1; c < 4; ++c) P[r][c] = addmod(P[r][c - 1], o.Fm, o.S);
            u[r] = o.u;
            bad[r] = o.bad;
            all_integer = all_integer && o.E == 0 && !o.bad;
            gain[r] = KIND == FUSED_OSC_GAIN ? operand_value(A.gain, A.params, A.n_inst, inst) : 1.f;

            if (seg == 0 && lane == 0 && live) {  // state write-back: state after ceil(n_samples/256) ticks
                const uint64_t T_end = (uint64_t)A.n_chunks * kChunk;
                double phase_end = (double)addmod(o.P0, mulmod(o.Fm, T_end, o.S), o.S) * o.u;
                if (o.bad) phase_end = __builtin_nan("");
                A.end_state[(size_t)A.osc_state_word * A.n_inst + inst] = phase_end;
                if (KIND == FUSED_OSC_RAMP) {
                    const double t_end = A.r_playing ? fmin(A.r_t0 + (double)T_end, A.r_d) : A.r_t0;
                    const bool playing_end = A.r_playing && (A.r_t0 + (double)T_end <= A.r_d);
                    A.end_state[(size_t)A.ramp_state_word * A.n_inst + inst] = t_end;
                    A.end_state[(size_t)(A.ramp_state_word + 1) * A.n_inst + inst] = playing_end ? 1.0 : 0.0;
                }
            }
        }
        // Ramp: tn = t0 + (n + 1) for this lane's first sample n; +256 per step.  Idle ramp: t stays t0.
        double tn = A.r_t0 + (A.r_playing ? (double)(t_start + lane * 4 + 1) : 0.0);
        const double tn_step = A.r_playing ? (double)kChunk : 0.0;
        const double tn_c = A.r_playing ? 1.0 : 0.0;
        float *row = A.out + t_start + lane * 4;

        auto ramp4 = [&](float (&rv)[4]) {
            if (KIND == FUSED_OSC_RAMP) {
                rv[0] = ramp_value<FASTDIV>(A, tn, dy);
#pragma unroll
                for (int c = 1; c < 4; ++c) rv[c] = ramp_value<FASTDIV>(A, tn + tn_c * c, dy);
                tn += tn_step;
            }
        };

        if (all_integer) {
            // every phase is an integer: fraction == 0 and out = table[phase] exactly (Osc.js:43-45)
            uint32_t idx[R][4], step[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                step[r] = (uint32_t)step256[r];
                for (int c = 0; c < 4; ++c) idx[r][c] = (uint32_t)P[r][c];
            }
            auto run = [&](uint32_t ga, uint32_t gb, auto vec) {
                for (uint32_t g = ga; g < gb; ++g) {
                    float rv[4];
                    ramp4(rv);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        float v[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            v[c] = table.at(idx[r][c]);
                            idx[r][c] += step[r];
                            idx[r][c] = min(idx[r][c], idx[r][c] - sr);  // wrap (idx < 2 sr; underflow loses the min)
                            if (KIND == FUSED_OSC_RAMP) v[c] = v[c] * rv[c];
                            if (KIND == FUSED_OSC_GAIN) v[c] = v[c] * gain[r];
                            v[c] = fix_out<FINITE>(v[c]);
                        }
                        store4<decltype(vec)::value>(row + roff[r], v, (uint64_t)g * kChunk + lane * 4, A.n_samples);
                    }
                    row += kChunk;
                }
            };
            const uint32_t gm = max(g0, min(g1, n_full));
            run(g0, gm, std::true_type{});
            run(gm, g1, std::false_type{});
        } else {
            double ph[R][4], D[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                D[r] = (double)step256[r] * u[r];
                for (int c = 0; c < 4; ++c) ph[r][c] = (double)P[r][c] * u[r];
            }
            auto run = [&](uint32_t ga, uint32_t gb, auto vec) {
                for (uint32_t g = ga; g < gb; ++g) {
                    float rv[4];
                    ramp4(rv);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        float v[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const uint32_t idx = (uint32_t)(int)ph[r][c];
                            const double fraction = ph[r][c] - (double)(int)idx;
                            float ta, tb;
                            table.pair(idx, ta, tb);
                            v[c] = (float)((double)ta * (1.0 - fraction) + (double)tb * fraction);
                            if (bad[r]) v[c] = 0.f;
                            ph[r][c] += D[r];
                            if (ph[r][c] >= srd) ph[r][c] -= srd;
                            if (KIND == FUSED_OSC_RAMP) v[c] = v[c] * rv[c];
                            if (KIND == FUSED_OSC_GAIN) v[c] = v[c] * gain[r];
                            v[c] = fix_out<FINITE>(v[c]);
                        }
                        store4<decltype(vec)::value>(row + roff[r], v, (uint64_t)g * kChunk + lane * 4, A.n_samples);
                    }
                    row += kChunk;
                }
            };
            const uint32_t gm = max(g0, min(g1, n_full));
            run(g0, gm, std::true_type{});
            run(gm, g1, std::false_type{});
        }
    }
}

template <int KIND, int TBL, int R, bool FASTDIV, bool FINITE, int BLOCK>
static hipError_t launch_one(const FusedArgs &A, int grid, size_t lds_bytes, hipStream_t stream) {
    auto kernel = dusp_fused_kernel<KIND, TBL, R, FASTDIV, FINITE, BLOCK>;
    if (lds_bytes > 65536) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(BLOCK), lds_bytes, stream, A);
    return hipGetLastError();
}

// DUSP_FUSED_TABLE=global|lds overrides the table placement (used by the A/B benchmarks).
static int table_mode_override() {
    const char *e = getenv("DUSP_FUSED_TABLE");
    if (!e) return -1;
    return e[0] == 'l' ? 1 : 0;
}

hipError_t launch_fused(const FusedPlan &plan, const FusedLaunch &L, hipStream_t stream) {
    FusedArgs A{};
    A.params = L.params;
    A.table = L.tables + (size_t)plan.table_id * L.table_stride;
    A.out = L.out;
    A.end_state = L.end_state;
    A.n_samples = L.n_samples;
    A.n_inst = L.n_inst;
    A.n_groups = (uint32_t)((L.n_samples + kChunk - 1) / kChunk);
    A.sample_rate = L.sample_rate;
    A.n_chunks = L.n_chunks;
    A.f = plan.f;
    A.gain = plan.gain;
    A.phase0 = plan.phase0;
    A.r_d = plan.r_d; A.r_y0 = plan.r_y0; A.r_y1 = plan.r_y1; A.r_t0 = plan.r_t0; A.r_rcp = plan.r_rcp;
    A.r_playing = plan.r_playing;
    A.r_fastdiv = plan.r_fastdiv;
    A.vec4_ok = (L.n_samples % 4 == 0) && (((uintptr_t)L.out & 15) == 0);
    A.osc_state_word = 0;
    A.ramp_state_word = 1;

    int tbl = (L.table_antisym && L.sample_rate % 2 == 0) ? 1 : 0;
    if (table_mode_override() == 0) tbl = 0;
    const uint32_t last = L.sample_rate / 2 + 1;
    const size_t lds_bytes = tbl ? ((size_t)(last + (last >> 5) + 2) * sizeof(float) + 15) & ~(size_t)15 : 0;
    if (lds_bytes > 160 * 1024) tbl = 0;

    // R voices per wave share one Ramp evaluation per step; without a Ramp there is nothing to share
    constexpr int kR = 4;
    const int R = plan.kind == FUSED_OSC_RAMP ? kR : 1;
    // segment length: enough (instance block, segment) items to keep every wave slot busy several times over
    const int block = tbl ? 1024 : 256;
    const int grid = tbl ? L.n_cus : L.n_cus * 8;
    const uint64_t total_waves = (uint64_t)grid * (block / 64);
    const uint64_t n_blk = (A.n_inst + R - 1) / R;
    const uint64_t all_groups = n_blk * A.n_groups;
    uint64_t seg = all_groups / (total_waves * 8);
    if (seg < 32) seg = 32;
    if (seg > 1024) seg = 1024;
    A.seg_groups = (uint32_t)seg;
    A.n_seg = (A.n_groups + A.seg_groups - 1) / A.seg_groups;
    const bool finite = L.table_finite && plan.kind != FUSED_OSC_GAIN && std::isfinite(plan.r_y0) && std::isfinite(plan.r_y1);

#define DUSP_LAUNCH2(KIND, RR, FD, FIN) \
    return tbl ? launch_one<KIND, 1, RR, FD, FIN, 1024>(A, grid, lds_bytes, stream) : launch_one<KIND, 0, RR, FD, FIN, 256>(A, grid, 0, stream)
    switch (plan.kind) {
    case FUSED_OSC:
        if (finite) { DUSP_LAUNCH2(FUSED_OSC, 1, false, true); }
        DUSP_LAUNCH2(FUSED_OSC, 1, false, false);
    case FUSED_OSC_RAMP:
        if (plan.r_fastdiv && finite) { DUSP_LAUNCH2(FUSED_OSC_RAMP, kR, true, true); }
        if (plan.r_fastdiv) { DUSP_LAUNCH2(FUSED_OSC_RAMP, kR, true, false); }
        DUSP_LAUNCH2(FUSED_OSC_RAMP, kR, false, false);
    case FUSED_OSC_GAIN: DUSP_LAUNCH2(FUSED_OSC_GAIN, 1, false, false);
    }
#undef DUSP_LAUNCH2
    return hipErrorInvalidValue;
}

}  // namespace dusp
