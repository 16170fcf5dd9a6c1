// sumchain_engine.hip — fused kernel for the reference's `Sum.many` mix-down (Sum.js:18-29):
// a left-deep chain  ((osc0 + osc1) + osc2) + ...  of N constant-frequency oscillators rendered
// into ONE channel (BASELINE configs[2], "voices summed" reading).
//
// The reference ticks 2N-1 units per chunk and rounds to f32 after every add; the f32 order of
// the adds is part of the result, so the voices cannot be split across waves.  Time can: a work
// item is a block of GB consecutive 256-sample groups of one instance; its wave walks the N voices
// IN CHAIN ORDER, each lane holding the running f32 sums of its 4 x GB samples in registers, and
// writes the block once.  Nothing is read from HBM but the voice records; 4 B per OUTPUT sample are
// written — by construction this config is ALU/LDS-bound, not HBM-bound (N table lookups per 4 B).
//
// Every voice's phase is exact 32.32 fixed point (host-checked: lsb(f) >= 2^-32, f constant): the
// phase of (block b, lane l, sample c) is (B0 + b*bs + l*step4 + c*Fm) mod S evaluated with exact
// integer arithmetic (quotients estimated in f64 and corrected), then advanced by step256 per group
// — the same INT / FX32 paths as fused_engine.hip.
//
// Enveloped voices (ENV): a chain of Multiply(Osc, k) voices multiplies every lookup by the voice's constant (ENV 1); a chain of
// Multiply(Osc, Ramp) voices whose Ramps are equal — the usual mix of enveloped voices — evaluates the Ramp ONCE per work item
// (a function of the sample index alone: 4 x GB values per lane) and multiplies every voice's lookups by it (ENV 2): each
// product is one f32 rounding like the Multiply unit's (Multiply.js:23-34), then the Sum's.
#include <hip/hip_runtime.h>

#include "device_types.hpp"
#include "fused_device.hpp"
#include "fused_plan.hpp"

namespace dusp {

template <int TBL, bool INT, int GB, bool FINITE, int BLOCK, int ENV>
__global__ void __launch_bounds__(BLOCK) dusp_sumchain_kernel(SumArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds_table[];
    Table<TBL> table;
    table.g = A.table;
    table.h = lds_table;
    table.N = A.sample_rate + 1;
    table.M = A.sample_rate / 2;
    if (TBL == 1) load_half_table<BLOCK>(lds_table, A.table, A.sample_rate);

    const uint32_t lane = threadIdx.x & 63u;
    constexpr uint32_t waves_per_block = BLOCK / 64;
    const uint64_t n_items = (uint64_t)A.n_inst * A.n_blocks;
    const uint64_t total_waves = (uint64_t)gridDim.x * waves_per_block;
    const uint32_t sr = A.sample_rate;
    const uint32_t n_full = A.first_block * GB + (A.vec4_ok ? (uint32_t)(A.n_samples / kChunk) : 0u);  // (groups counted like g0: from the render's first sample)

    for (uint64_t item = (uint64_t)blockIdx.x * waves_per_block + (threadIdx.x >> 6); item < n_items; item += total_waves) {
        const uint32_t blk = (uint32_t)(item % A.n_blocks) + A.first_block;  // (a window: blocks counted from the render's first sample)
        const uint32_t inst = (uint32_t)(item / A.n_blocks);
        const uint32_t g0 = blk * GB;
        const size_t row_at = (size_t)inst * A.n_samples + (size_t)(g0 - A.first_block * GB) * kChunk + lane * 4;
        float acc[GB][4];
#pragma unroll
        for (int g = 0; g < GB; ++g)
            for (int c = 0; c < 4; ++c) acc[g][c] = 0.f;  // 0 + v0 == v0: the chain starts at the first voice
        if (A.init) {  // ... or continues the sums another GPU's voices left (whole groups, 16-byte rows: the host checks)
#pragma unroll
            for (int g = 0; g < GB; ++g)
                if (g0 + g < n_full) {
                    const f32x4 v = *(const f32x4 *)(A.init + row_at + (size_t)g * kChunk);
                    acc[g][0] = v[0]; acc[g][1] = v[1]; acc[g][2] = v[2]; acc[g][3] = v[3];
                } else if (g0 + g < A.n_groups) {  // (the window's last, partial group)
                    const uint64_t t = (uint64_t)(g0 + g - A.first_block * GB) * kChunk + lane * 4;
                    for (int c = 0; c < 4; ++c)
                        if (t + c < A.n_samples) acc[g][c] = A.init[row_at + (size_t)g * kChunk + c];
                }
        }
        float env[ENV == 2 ? GB : 1][4];
        if (ENV == 2) {  // Ramp.js:25-40 in closed form: t(n) = min(t0 + n + 1, duration) while playing
            const double dy = A.r_y1 - A.r_y0, rcp = 1.0 / A.r_d;
#pragma unroll
            for (int g = 0; g < (ENV == 2 ? GB : 1); ++g)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint64_t n = (uint64_t)(g0 + g) * kChunk + lane * 4 + c;
                    const double tt = A.r_playing ? fmin(A.r_t0 + (double)(n + 1), A.r_d) : A.r_t0;
                    double q;
                    if (A.r_fastdiv) {  // (the host has checked the refined reciprocal against the division on every t of this Ramp)
                        q = tt * rcp;
                        q = fma(fma(-q, A.r_d, tt), rcp, q);
                    } else
                        q = tt / A.r_d;
                    env[g][c] = (float)(A.r_y0 + q * dy);
                }
        }

        for (uint32_t j = 0; j < A.n_voices; ++j) {
            const SumVoice rc = A.voices[j];
            if (INT) {
                // every phase is an integer: out = table[phase] exactly (Osc.js:43-45 with fraction 0)
                const uint32_t f = (uint32_t)(rc.Fm >> 32), s4 = (uint32_t)(rc.step4 >> 32), s256 = (uint32_t)(rc.step256 >> 32);
                const uint32_t base = mod_u32((uint32_t)(rc.B0 >> 32) + blk * (uint32_t)(rc.bs >> 32), sr, A.inv_sr);
                uint32_t idx[4];
                idx[0] = mod_u32(base + lane * s4, sr, A.inv_sr);
#pragma unroll
                for (int c = 1; c < 4; ++c) {
                    idx[c] = idx[c - 1] + f;
                    idx[c] = min(idx[c], idx[c] - sr);
                }
#pragma unroll
                for (int g = 0; g < GB; ++g)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float v = table.at(idx[c]);
                        if (ENV == 1) v = v * rc.gain;          // the voice's Multiply: one f32 rounding
                        if (ENV == 2) v = v * env[g][c];
                        acc[g][c] = acc[g][c] + v;  // f32 add, rounded per voice like the Sum units
                        idx[c] += s256;
                        idx[c] = min(idx[c], idx[c] - sr);
                    }
            } else {
                const uint64_t base = mod_u64(rc.B0 + (uint64_t)blk * rc.bs, A.S, A.inv_S);
                uint64_t P = mod_u64(base + (uint64_t)lane * rc.step4, A.S, A.inv_S);
                uint32_t I[4], F[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    I[c] = (uint32_t)(P >> 32);
                    F[c] = (uint32_t)P;
                    P += rc.Fm;
                    if (P >= A.S) P -= A.S;
                }
                const uint32_t dI = (uint32_t)(rc.step256 >> 32), dF = (uint32_t)rc.step256;
#pragma unroll
                for (int g = 0; g < GB; ++g)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float ta, tb;
                        table.pair(I[c], ta, tb);
                        const double wb = (double)F[c];
                        const double wa = 4294967296.0 - wb;
                        const float x = (float)((double)ta * wa + (double)tb * wb);  // see fused_engine.hip FX32
                        float v = ldexpf(x, -32);
                        if (ENV == 1) v = v * rc.gain;
                        if (ENV == 2) v = v * env[g][c];
                        acc[g][c] = acc[g][c] + v;
                        const uint32_t f2 = F[c] + dF;
                        uint32_t i2 = I[c] + dI + (f2 < F[c] ? 1u : 0u);
                        i2 = min(i2, i2 - sr);
                        F[c] = f2;
                        I[c] = i2;
                    }
            }
        }
        float *row = A.out + row_at;
#pragma unroll
        for (int g = 0; g < GB; ++g) {
            const uint32_t gg = g0 + g;
            if (gg >= A.n_groups) break;
            float v[4];
            for (int c = 0; c < 4; ++c) v[c] = A.raw ? acc[g][c] : fix_out<FINITE>(acc[g][c]);
            if (gg < n_full)
                store4<true>(row, v, 0, A.n_samples);
            else
                store4<false>(row, v, (uint64_t)(gg - A.first_block * GB) * kChunk + lane * 4, A.n_samples);
            row += kChunk;
        }
    }
}

template <int TBL, bool INT, int GB, bool FINITE, int BLOCK, int ENV>
static hipError_t launch_sum(const SumArgs &A, int grid, size_t lds_bytes, hipStream_t stream) {
    auto kernel = dusp_sumchain_kernel<TBL, INT, GB, FINITE, BLOCK, ENV>;
    if (lds_bytes > 65536) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(BLOCK), lds_bytes, stream, A);
    return hipGetLastError();
}

hipError_t launch_sumchain(const FusedPlan &plan, const FusedLaunch &L, const SumVoice *d_voices, int gb, hipStream_t stream) {
    SumArgs A{};
    A.voices = d_voices;
    A.table = L.tables + (size_t)plan.table_id * L.table_stride;
    A.out = L.out;
    A.n_samples = L.n_samples;
    A.S = (uint64_t)L.sample_rate << 32;
    A.inv_S = 1.0 / (double)A.S;
    A.inv_sr = 1.0 / (double)L.sample_rate;
    A.n_voices = (uint32_t)plan.sum_f.size();
    A.n_inst = L.n_inst;
    // (a window of a chain: L.chain_first is a whole number of blocks — the caller checks; groups and blocks count from the render's first sample)
    A.first_block = (uint32_t)(L.chain_first / ((uint64_t)gb * kChunk));
    A.init = L.chain_init;
    A.raw = L.chain_raw ? 1u : 0u;
    const uint32_t window_groups = (uint32_t)((L.n_samples + kChunk - 1) / kChunk);
    A.n_groups = A.first_block * (uint32_t)gb + window_groups;
    A.n_blocks = (window_groups + gb - 1) / gb;
    A.sample_rate = L.sample_rate;
    A.vec4_ok = (L.n_samples % 4 == 0) && (((uintptr_t)L.out & 15) == 0);
    A.r_d = plan.r_d;
    A.r_y0 = plan.r_y0;
    A.r_y1 = plan.r_y1;
    A.r_t0 = plan.r_t0;
    A.r_playing = plan.r_playing;
    A.r_fastdiv = plan.r_fastdiv;

    const bool tbl = L.table_antisym && L.sample_rate % 2 == 0 && half_table_lds_bytes(L.sample_rate) <= 160 * 1024;
    const size_t lds_bytes = tbl ? half_table_lds_bytes(L.sample_rate) : 0;
    const int grid = tbl ? L.n_cus : L.n_cus * 8;
    const bool finite = L.table_finite && plan.sum_env == 0;  // (enveloped voices: products may overflow, so the copy-out keeps the full `x || 0`)
    // INT needs blk * (bs >> 32) < 2^32
    const bool use_int = plan.sum_all_int && ((uint64_t)A.first_block + A.n_blocks) * L.sample_rate < (1ull << 32);

#define DUSP_S5(TB, IN, G, FIN, EN) \
    return TB ? launch_sum<1, IN, G, FIN, 1024, EN>(A, grid, lds_bytes, stream) : launch_sum<0, IN, G, FIN, 256, EN>(A, grid, 0, stream)
#define DUSP_S4(TB, IN, G, FIN) \
    do { if (plan.sum_env == 2) { DUSP_S5(TB, IN, G, FIN, 2); } if (plan.sum_env == 1) { DUSP_S5(TB, IN, G, FIN, 1); } DUSP_S5(TB, IN, G, FIN, 0); } while (0)
#define DUSP_S3(IN, G) \
    do { if (finite) { DUSP_S4(tbl, IN, G, true); } DUSP_S4(tbl, IN, G, false); } while (0)
    if (gb == 8) {
        if (use_int) DUSP_S3(true, 8);
        DUSP_S3(false, 8);
    }
    if (use_int) DUSP_S3(true, 4);
    DUSP_S3(false, 4);
#undef DUSP_S3
#undef DUSP_S4
#undef DUSP_S5
}

}  // namespace dusp
