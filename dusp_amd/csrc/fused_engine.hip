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

__device__ __forceinline__ float fix_out(float v) { return (v != v) ? 0.f : v + 0.f; }  // `x || 0`

// Table access.  TBL == 0: padded full table in global memory (served by L2).
// TBL == 1: half table H[0..M+1] = T[0..M+1] in LDS, M = sr/2, N = sr+1;
//           T[i] = H[i] for i <= M+1 and -H[N-i] above.
template <int TBL>
struct Table {
    const float *g;
    const float *h;
    int N, M;
    __device__ __forceinline__ float at(int i) const {
        if (TBL == 0) return g[i];
        const bool upper = i > M + 1;
        const float v = h[upper ? N - i : i];
        return upper ? -v : v;
    }
    // (T[i], T[i+1]) for the lerp
    __device__ __forceinline__ void pair(int i, float &a, float &b) const {
        if (TBL == 0) {
            a = g[i];
            b = g[i + 1];
            return;
        }
        const bool upper = i > M;
        const int hidx = upper ? N - i - 1 : i;
        const float x = h[hidx], y = h[hidx + 1];
        a = upper ? -y : x;
        b = upper ? -x : y;
    }
};

template <int KIND>
__device__ __forceinline__ float ramp_value(const FusedArgs &A, double tn, double dy) {
    (void)A; (void)tn; (void)dy;
    return 1.f;
}
template <>
__device__ __forceinline__ float ramp_value<FUSED_OSC_RAMP>(const FusedArgs &A, double tn, double dy) {
    const double tt = fmin(tn, A.r_d);  // t++ then clamp to duration (Ramp.js:27-32)
    double q;
    if (A.r_fastdiv) {                  // host-verified to equal tt / duration on this Ramp's whole t sequence
        q = tt * A.r_rcp;
        const double rem = fma(-q, A.r_d, tt);
        q = fma(rem, A.r_rcp, q);
    } else
        q = tt / A.r_d;
    return (float)(A.r_y0 + q * dy);    // y0 + (t/duration) * (y1-y0)  (Ramp.js:38)
}

}  // namespace

template <int KIND, int TBL, int BLOCK>
__global__ void __launch_bounds__(BLOCK) dusp_fused_kernel(FusedArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds_table[];
    Table<TBL> table;
    table.g = A.table;
    table.h = lds_table;
    table.N = (int)A.sample_rate + 1;
    table.M = (int)A.sample_rate / 2;
    if (TBL == 1) {
        const int n_half = table.M + 2;
        for (int k = threadIdx.x; k < n_half; k += BLOCK) lds_table[k] = A.table[k];
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t waves_per_block = BLOCK / 64;
    const uint64_t n_items = (uint64_t)A.n_inst * A.n_seg;
    const uint64_t total_waves = (uint64_t)gridDim.x * waves_per_block;
    const double srd = (double)A.sample_rate;
    const uint32_t sr = A.sample_rate;

    for (uint64_t item = (uint64_t)blockIdx.x * waves_per_block + (threadIdx.x >> 6); item < n_items; item += total_waves) {
        const uint32_t inst = (uint32_t)(item % A.n_inst);
        const uint32_t seg = (uint32_t)(item / A.n_inst);
        const uint32_t g0 = seg * A.seg_groups;
        const uint32_t g1 = min(g0 + A.seg_groups, A.n_groups);
        const uint64_t t_start = (uint64_t)g0 * kChunk;

        const float f = operand_value(A.f, A.params, A.n_inst, inst);
        const OscFix o = osc_setup(f, A.phase0, sr);
        const uint64_t step4 = mulmod(o.Fm, 4, o.S);
        const uint64_t step256 = mulmod(o.Fm, 256, o.S);
        uint64_t P[4];
        P[0] = addmod(addmod(o.P0, mulmod(o.Fm, t_start + 1, o.S), o.S), mulmod(step4, lane, o.S), o.S);
        for (int c = 1; c < 4; ++c) P[c] = addmod(P[c - 1], o.Fm, o.S);

        float gain = 1.f;
        if (KIND == FUSED_OSC_GAIN) gain = operand_value(A.gain, A.params, A.n_inst, inst);
        const double dy = A.r_y1 - A.r_y0;
        // Ramp: tn = t0 + (n + 1) for this lane's first sample n; +256 per step.  Idle ramp: t stays t0.
        double tn = A.r_t0 + (A.r_playing ? (double)(t_start + lane * 4 + 1) : 0.0);
        const double tn_step = A.r_playing ? (double)kChunk : 0.0;
        const double tn_c = A.r_playing ? 1.0 : 0.0;

        float *row = A.out + (size_t)inst * A.n_samples + t_start + lane * 4;
        const bool integer_phase = (o.E == 0) && !o.bad;

        if (integer_phase) {
            // every phase is an integer: fraction == 0, out = table[phase] exactly (Osc.js:43-45)
            uint32_t idx[4];
            for (int c = 0; c < 4; ++c) idx[c] = (uint32_t)P[c];
            const uint32_t step = (uint32_t)step256;
            for (uint32_t g = g0; g < g1; ++g) {
                float v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    v[c] = table.at((int)idx[c]);
                    idx[c] += step;
                    idx[c] = min(idx[c], idx[c] - sr);  // wrap: idx < 2 sr, unsigned underflow loses the min
                    if (KIND == FUSED_OSC_RAMP) v[c] = v[c] * ramp_value<KIND>(A, tn + tn_c * c, dy);
                    if (KIND == FUSED_OSC_GAIN) v[c] = v[c] * gain;
                    v[c] = fix_out(v[c]);
                }
                tn += tn_step;
                const uint64_t t = (uint64_t)g * kChunk + lane * 4;
                if (A.vec4_ok && t + 4 <= A.n_samples)
                    __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, (f32x4 *)row);
                else
                    for (int c = 0; c < 4; ++c)
                        if (t + c < A.n_samples) row[c] = v[c];
                row += kChunk;
            }
        } else {
            double ph[4];
            for (int c = 0; c < 4; ++c) ph[c] = (double)P[c] * o.u;
            const double D = (double)step256 * o.u;
            for (uint32_t g = g0; g < g1; ++g) {
                float v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int idx = (int)ph[c];
                    const double fraction = ph[c] - (double)idx;
                    float ta, tb;
                    table.pair(idx, ta, tb);
                    v[c] = (float)((double)ta * (1.0 - fraction) + (double)tb * fraction);
                    if (o.bad) v[c] = 0.f;
                    ph[c] += D;
                    if (ph[c] >= srd) ph[c] -= srd;
                    if (KIND == FUSED_OSC_RAMP) v[c] = v[c] * ramp_value<KIND>(A, tn + tn_c * c, dy);
                    if (KIND == FUSED_OSC_GAIN) v[c] = v[c] * gain;
                    v[c] = fix_out(v[c]);
                }
                tn += tn_step;
                const uint64_t t = (uint64_t)g * kChunk + lane * 4;
                if (A.vec4_ok && t + 4 <= A.n_samples)
                    __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, (f32x4 *)row);
                else
                    for (int c = 0; c < 4; ++c)
                        if (t + c < A.n_samples) row[c] = v[c];
                row += kChunk;
            }
        }

        // state write-back: the state every unit holds after ceil(n_samples/256) ticks
        if (seg == 0 && lane == 0) {
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
}

template <int KIND, int TBL, int BLOCK>
static hipError_t launch_one(const FusedArgs &A, int grid, size_t lds_bytes, hipStream_t stream) {
    auto kernel = dusp_fused_kernel<KIND, TBL, BLOCK>;
    if (lds_bytes > 65536) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(BLOCK), lds_bytes, stream, A);
    return hipGetLastError();
}

// DUSP_FUSED_TABLE=global|lds overrides the table placement (used by the A/B benchmarks).
static int table_mode_override() {
    static int mode = [] {
        const char *e = getenv("DUSP_FUSED_TABLE");
        if (!e) return -1;
        return e[0] == 'l' ? 1 : 0;
    }();
    return mode;
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
    if (table_mode_override() == 1 && !(L.table_antisym && L.sample_rate % 2 == 0)) tbl = 0;
    const size_t lds_bytes = tbl ? ((size_t)(L.sample_rate / 2 + 2) * sizeof(float) + 15) & ~(size_t)15 : 0;
    if (lds_bytes > 160 * 1024) tbl = 0;

    // segment length: enough (instance, segment) items to keep every wave slot busy several times over
    const int block = tbl ? 1024 : 256;
    const int grid = tbl ? L.n_cus : L.n_cus * 8;
    const uint64_t total_waves = (uint64_t)grid * (block / 64);
    const uint64_t all_groups = (uint64_t)A.n_inst * A.n_groups;
    uint64_t seg = all_groups / (total_waves * 8);
    if (seg < 16) seg = 16;
    if (seg > 1024) seg = 1024;
    A.seg_groups = (uint32_t)seg;
    A.n_seg = (A.n_groups + A.seg_groups - 1) / A.seg_groups;

#define DUSP_LAUNCH(KIND)                                                                  \
    return tbl ? launch_one<KIND, 1, 1024>(A, grid, lds_bytes, stream) : launch_one<KIND, 0, 256>(A, grid, 0, stream)
    switch (plan.kind) {
    case FUSED_OSC: DUSP_LAUNCH(FUSED_OSC);
    case FUSED_OSC_RAMP: DUSP_LAUNCH(FUSED_OSC_RAMP);
    case FUSED_OSC_GAIN: DUSP_LAUNCH(FUSED_OSC_GAIN);
    }
#undef DUSP_LAUNCH
    return hipErrorInvalidValue;
}

}  // namespace dusp
