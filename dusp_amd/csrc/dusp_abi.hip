// dusp_abi.hip — implementation of the C ABI declared in include/dusp_hip.h.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "../../include/dusp_hip.h"
#include "device_types.hpp"
#include "fused_plan.hpp"
#include "program.hpp"

namespace dusp {
hipError_t launch_chunk_engine(const ChunkArgs &a, hipStream_t stream);
hipError_t launch_state_init(double *state, const double *init, uint32_t n_slots, uint32_t n_pad, hipStream_t stream);
hipError_t launch_fill(float *out, size_t n_floats, float value, hipStream_t stream);
hipError_t launch_wave_to_chunk(const float *wave_rings, float *chunk_rings, uint64_t ring_samples, const float *saved_bufs, float *chunk_scratch,
                                uint32_t n_bufs, uint32_t n_inst, uint32_t n_pad, hipStream_t stream);
hipError_t launch_interleave(const float *d_planar, float *d_out, uint32_t n_instances, uint32_t n_channels, uint64_t n_samples, hipStream_t stream);
hipError_t launch_fused(const FusedPlan &plan, const FusedLaunch &L, hipStream_t stream);
hipError_t launch_loop2_engine(const ChunkArgs &a, const LoopShape &L, bool lds_table_ok, hipStream_t stream);
hipError_t launch_loop_engine(const ChunkArgs &a, const LoopShape &L, bool lds_table_ok, int n_cus, hipStream_t stream);
hipError_t launch_wave_engine(WaveArgs A, bool lds_table_ok, hipStream_t stream);
hipError_t launch_sumchain(const FusedPlan &plan, const FusedLaunch &L, const SumVoice *d_voices, int gb, hipStream_t stream);
}  // namespace dusp

static thread_local std::string g_error;

struct dusp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    float *d_tables = nullptr;  // [kNumTables][table_stride]
    uint32_t table_len = 0;     // entries per uploaded table (= sample_rate + 1), 0 until the first upload
    uint32_t table_stride = 0;
    bool table_set[dusp::kNumTables] = {false, false, false, false, false};
    bool table_antisym[dusp::kNumTables] = {false, false, false, false, false};
    bool table_finite[dusp::kNumTables] = {false, false, false, false, false};
    bool table_fx32_ok[dusp::kNumTables] = {false, false, false, false, false};  // min nonzero |T| >= 2^-20
    int n_cus = 256;
};

template <class T>
struct DevBuf {  // grow-only device allocation
    T *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc((void **)&p, n * sizeof(T));
        if (e == hipSuccess) cap = n;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct dusp_program {
    dusp_ctx *ctx = nullptr;
    dusp::Program P;
    int engine = DUSP_ENGINE_CHUNK;
    dusp::FusedPlan fused;
    dusp::WavePlan wave;
    dusp::LoopShape loop;
    bool loop_two_stage = false;  // constant delay of at least one chunk: loop2_engine.hip
    // program constants on the device
    DevBuf<dusp::DevOp> d_ops;
    DevBuf<int32_t> d_out_bufs;
    DevBuf<double> d_init;
    // per-render workspaces (grown on demand)
    DevBuf<float> d_scratch, d_rings;
    DevBuf<double> d_state;
    DevBuf<unsigned long long> d_seg;  // WAVE, time-split: segment phase totals + start phases
    DevBuf<double> d_fused_state;  // FUSED: [n_state_words][n_inst] end-of-render state
    DevBuf<dusp::OscRec> d_recs;   // FUSED: per-voice oscillator records
    DevBuf<dusp::SumVoice> d_sum_voices;  // FUSED sum chain: per-oscillator records
    std::vector<dusp::SumVoice> h_sum_voices;
    std::vector<double> h_sum_end;
    uint32_t last_n_inst = 0, last_n_pad = 0;
    bool rendered = false;
    // event-segmented rendering (dusp_program_continue)
    // host-side conveniences for segment-by-segment rendering (many short renders + state downloads per second)
    std::vector<double> h_state;   // copy of d_state / d_fused_state, fetched once per render on the first state download
    bool h_state_valid = false;
    DevBuf<float> d_host_out, d_host_par, d_host_frames, d_host_in;  // dusp_render_host staging, grown on demand
    int requested_engine = DUSP_ENGINE_AUTO;
    bool resumable = false;      // built with DUSP_ENGINE_RESUMABLE
    bool persistent = false;     // rings / feedback edges: device memory carries over between segments (CHUNK engine only)
    bool keep_memory = false;    // the next render continues: do not clear chunk buffers and rings
    bool delay_changed = false;     // a continuation changed a Delay's constant: the wave engine's write-once ring protocol no longer applies
    bool migrate_to_chunk = false;  // ... and first moves the wave engine's rings / saved buffers into the chunk layout
    DevBuf<float> d_saved_bufs, d_rings_wave;  // wave engine, resumable: outlets' last chunk; rings parked during a migration
    int64_t next_clock = 0;      // circuit clock the last render stopped at
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

#define CTX_FAIL(ctx, code, msg)  \
    do {                          \
        (ctx)->err = (msg);       \
        return (code);            \
    } while (0)

#define HIP_TRY(ctx, expr)                                                                        \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (ctx)->err = std::string("HIP error: ") + hipGetErrorString(e_) + " in " + #expr;     \
            return DUSP_ERR_HIP;                                                                  \
        }                                                                                         \
    } while (0)

extern "C" {

const char *dusp_version(void) { return "dusp-hip 0.1.0 (gfx950)"; }
int dusp_abi_version(void) { return DUSP_ABI_VERSION; }

const char *dusp_last_error(const dusp_ctx *ctx) { return ctx ? ctx->err.c_str() : g_error.c_str(); }

int dusp_ctx_create(int device, dusp_ctx **out) {
    if (!out) {
        g_error = "dusp_ctx_create: out is NULL";
        return DUSP_ERR_ARG;
    }
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count < 1) {
        g_error = std::string("no usable HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0") +
                  " (this library has no CPU fallback)";
        return DUSP_ERR_HIP;
    }
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) device = 0;
    }
    if (device >= count) {
        g_error = "device index out of range";
        return DUSP_ERR_ARG;
    }
    std::unique_ptr<dusp_ctx> ctx(new (std::nothrow) dusp_ctx);
    if (!ctx) {
        g_error = "out of memory";
        return DUSP_ERR_ARG;
    }
    ctx->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreate(&ctx->stream)) != hipSuccess) {
        g_error = std::string("HIP error: ") + hipGetErrorString(e);
        return DUSP_ERR_HIP;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        ctx->n_cus = prop.multiProcessorCount;
    *out = ctx.release();
    return DUSP_OK;
}

void dusp_ctx_destroy(dusp_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamDestroy(ctx->stream);
    }
    if (ctx->d_tables) (void)hipFree(ctx->d_tables);
    delete ctx;
}

int dusp_table_upload(dusp_ctx *ctx, int table_id, const float *table, size_t n) {
    if (!ctx) return DUSP_ERR_ARG;
    if (table_id < 0 || table_id >= dusp::kNumTables || !table || n < 9 || n > (1u << 22) + 1)
        CTX_FAIL(ctx, DUSP_ERR_ARG, "dusp_table_upload: bad table id, pointer or length");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->d_tables) {
        ctx->table_len = (uint32_t)n;
        ctx->table_stride = (uint32_t)((n + 1 + 3) & ~(size_t)3);  // >= n+1 entries (one pad for idx+1), 16-byte rows
        HIP_TRY(ctx, hipMalloc((void **)&ctx->d_tables, sizeof(float) * dusp::kNumTables * ctx->table_stride));
        HIP_TRY(ctx, hipMemset(ctx->d_tables, 0, sizeof(float) * dusp::kNumTables * ctx->table_stride));
    } else if (n != ctx->table_len) {
        CTX_FAIL(ctx, DUSP_ERR_ARG, "dusp_table_upload: all tables of a context must have the same length");
    }
    std::vector<float> row(ctx->table_stride, 0.f);
    std::memcpy(row.data(), table, n * sizeof(float));
    for (size_t k = n; k < row.size(); k++) row[k] = table[n - 1];  // pad: keeps an idx+1 read finite and in range
    HIP_TRY(ctx, hipMemcpy(ctx->d_tables + (size_t)table_id * ctx->table_stride, row.data(),
                           row.size() * sizeof(float), hipMemcpyHostToDevice));
    // T[N-t] == -T[t] for t = 1..N-1 lets a fused kernel keep half the table in LDS
    bool antisym = (n % 2) == 1;
    for (size_t t = 1; antisym && t < n; t++) antisym = table[n - t] == -table[t];
    ctx->table_antisym[table_id] = antisym;
    bool finite = true;
    for (size_t t = 0; finite && t < n; t++) finite = std::isfinite(table[t]);
    ctx->table_finite[table_id] = finite;
    bool big = finite;
    for (size_t t = 0; big && t < n; t++) big = table[t] == 0.f || std::fabs(table[t]) >= 9.5367431640625e-07f;
    ctx->table_fx32_ok[table_id] = big;
    ctx->table_set[table_id] = true;
    return DUSP_OK;
}

// Engine selection + upload of the program constants; shared by build and continue.
static int finish_build(dusp_program *prog) {
    dusp_ctx *ctx = prog->ctx;
    int engine = prog->requested_engine;
    prog->fused = dusp::FusedPlan();
    {
        auto checked = std::move(prog->wave.ramp_checked);  // verdicts survive re-planning
        prog->wave = dusp::WavePlan();
        prog->wave.ramp_checked = std::move(checked);
    }
    const bool fusable = dusp::plan_fused(prog->P, prog->fused);
    const bool wavable = dusp::plan_wave(prog->P, prog->wave, prog->resumable);
    for (size_t k = 0; k < prog->wave.osc_level.size() && k < prog->P.ops.size(); k++)  // FM depth, for time-split rendering
        if (prog->wave.osc_level[k] >= 0) prog->P.ops[k].d[0] = (double)prog->wave.osc_level[k];
    for (size_t k = 0; k < prog->wave.ramp_fastdiv.size() && k < prog->P.ops.size(); k++)
        if (prog->P.ops[k].op == dusp::OP_RAMP) prog->P.ops[k].attr = prog->wave.ramp_fastdiv[k];
    // A continuation of a resumable program never fails over the engine: the circuit's state may have left the regime of the
    // engine the program was built with (an oscillator phase gone NaN, ...); the choice then falls to AUTO's rules below.
    if (prog->rendered && prog->resumable &&
        ((engine == DUSP_ENGINE_FUSED && !fusable) || (engine == DUSP_ENGINE_WAVE && !wavable)))
        engine = DUSP_ENGINE_AUTO;
    if (engine == DUSP_ENGINE_FUSED && !fusable)
        CTX_FAIL(ctx, DUSP_ERR_UNSUPPORTED, "dusp_program_build: no fused kernel for this graph shape (" + prog->fused.why + ")");
    if (engine == DUSP_ENGINE_WAVE && !wavable)
        CTX_FAIL(ctx, DUSP_ERR_UNSUPPORTED, "dusp_program_build: the wave engine cannot run this graph (" + prog->wave.why + ")");
    std::string loop_why;
    const bool loopable = dusp::plan_loop(prog->P, prog->loop, loop_why);
    prog->loop_two_stage = false;
    if (loopable) {
        const dusp::DevOperand &dl = prog->loop.delay.in[1];
        const double len = (double)prog->loop.delay.ring_len;
        double dconst = (double)dl.cval;
        if (dconst >= len) dconst = std::fmod(dconst, len);
        const char *knob = getenv("DUSP_LOOP2");
        prog->loop_two_stage = dl.kind == dusp::SRC_CONST && std::floor(dconst) >= dusp::kChunk &&
                               std::floor(dconst) + dusp::kChunk <= len && prog->P.g.sample_rate <= 131072 &&
                               !(knob && knob[0] == '0');
    }
    if (engine == DUSP_ENGINE_LOOP && !loopable)
        CTX_FAIL(ctx, DUSP_ERR_UNSUPPORTED, "dusp_program_build: not the feedback-voice shape of the loop engine (" + loop_why + ")");
    // A circuit with rings or a feedback edge carries device memory from one segment to the next; only the chunk
    // engine keeps all of it (rings, every outlet's previous chunk) in HBM in a layout a later launch can pick up.
    // A circuit with rings or a feedback edge carries device memory from one segment to the next.  The chunk engine keeps
    // all of it in HBM; the wave engine parks its LDS chunk buffers in HBM between launches.  An engine, once chosen, is
    // kept for the whole chain (ring layouts differ) — except that a wave program that stops being plannable migrates to
    // the chunk engine.
    prog->persistent = prog->P.ring_samples != 0 || !prog->P.feed_forward;
    if (prog->resumable && prog->persistent) {
        if (engine != DUSP_ENGINE_AUTO && engine != DUSP_ENGINE_CHUNK && engine != DUSP_ENGINE_WAVE)
            CTX_FAIL(ctx, DUSP_ERR_UNSUPPORTED, "dusp_program_build: a resumable program with delay lines / feedback runs on DUSP_ENGINE_WAVE or DUSP_ENGINE_CHUNK");
        if (prog->rendered)  // continuing: stay, or fall back to the chunk engine
            engine = (prog->engine == DUSP_ENGINE_WAVE && wavable && !prog->delay_changed) ? DUSP_ENGINE_WAVE : DUSP_ENGINE_CHUNK;
        else if (engine == DUSP_ENGINE_AUTO)
            engine = wavable ? DUSP_ENGINE_WAVE : DUSP_ENGINE_CHUNK;
    }
    if (engine == DUSP_ENGINE_AUTO)
        engine = fusable ? DUSP_ENGINE_FUSED
                 : (loopable && (prog->loop_two_stage || prog->wave.ring_events)) ? DUSP_ENGINE_LOOP  // (one lane per voice beats slot rounds)
                 : wavable ? DUSP_ENGINE_WAVE
                 : loopable ? DUSP_ENGINE_LOOP
                           : DUSP_ENGINE_CHUNK;
    prog->engine = engine;

    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const dusp::Program &P = prog->P;
    size_t n_all_ops = P.ops.size();  // the settled op list, then the lists of the warm-up chunks (if any)
    for (const auto &w : P.warm_ops) n_all_ops += w.size();
    HIP_TRY(ctx, prog->d_ops.ensure(n_all_ops));
    std::vector<int32_t> out_bufs = P.out_bufs;
    if (engine == DUSP_ENGINE_WAVE) {  // the wave engine's view: chunk buffers renamed to their LDS slots, state blocks numbered
        std::vector<dusp::DevOp> ops = P.ops;
        const auto &slot = prog->wave.buf_slot;
        for (size_t k = 0; k < ops.size(); k++) {
            if (ops[k].out_buf >= 0) ops[k].out_buf = slot[(size_t)ops[k].out_buf];
            for (auto &in : ops[k].in)
                if (in.kind == dusp::SRC_BUF && in.idx >= 0 && in.idx < P.n_bufs) in.idx = slot[(size_t)in.idx];
            ops[k].lds_slot = prog->wave.op_state[k];
        }
        for (auto &op : ops)
            if (op.op == dusp::OP_RETRIGGER) op.pad = prog->wave.op_state[(size_t)op.pad];  // target op -> its state block
        for (auto &b : out_bufs) b = slot[(size_t)b];
        std::vector<dusp::DevOp> ordered(ops.size());
        for (size_t at = 0; at < ops.size(); at++) ordered[at] = ops[(size_t)prog->wave.order[at]];
        HIP_TRY(ctx, hipMemcpy(prog->d_ops.p, ordered.data(), ordered.size() * sizeof(dusp::DevOp), hipMemcpyHostToDevice));
    } else
    HIP_TRY(ctx, hipMemcpy(prog->d_ops.p, P.ops.data(), P.ops.size() * sizeof(dusp::DevOp), hipMemcpyHostToDevice));
    n_all_ops = P.ops.size();
    for (const auto &w : P.warm_ops) {
        if (!w.empty())
            HIP_TRY(ctx, hipMemcpy(prog->d_ops.p + n_all_ops, w.data(), w.size() * sizeof(dusp::DevOp), hipMemcpyHostToDevice));
        n_all_ops += w.size();
    }
    HIP_TRY(ctx, prog->d_out_bufs.ensure(out_bufs.size()));
    HIP_TRY(ctx, hipMemcpy(prog->d_out_bufs.p, out_bufs.data(), out_bufs.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    if (!P.init_state.empty()) {
        HIP_TRY(ctx, prog->d_init.ensure(P.init_state.size()));
        HIP_TRY(ctx, hipMemcpy(prog->d_init.p, P.init_state.data(), P.init_state.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    return DUSP_OK;
}

static int compile_status(dusp_ctx *ctx, const char *who, const std::string &err) {
    const bool unsupported = err.find("not supported") != std::string::npos || err.find("only ") != std::string::npos;
    CTX_FAIL(ctx, unsupported ? DUSP_ERR_UNSUPPORTED : DUSP_ERR_ARG, std::string(who) + ": " + err);
}

int dusp_program_build(dusp_ctx *ctx, const double *desc, size_t n_words, int engine, dusp_program **out) {
    if (!ctx) return DUSP_ERR_ARG;
    if (!out) CTX_FAIL(ctx, DUSP_ERR_ARG, "dusp_program_build: out is NULL");
    *out = nullptr;
    const bool resumable = (engine & DUSP_ENGINE_RESUMABLE) != 0;
    engine &= ~DUSP_ENGINE_RESUMABLE;
    if (engine != DUSP_ENGINE_AUTO && engine != DUSP_ENGINE_CHUNK && engine != DUSP_ENGINE_FUSED && engine != DUSP_ENGINE_WAVE && engine != DUSP_ENGINE_LOOP)
        CTX_FAIL(ctx, DUSP_ERR_ARG, "dusp_program_build: bad engine");
    std::unique_ptr<dusp_program> prog(new (std::nothrow) dusp_program);
    if (!prog) CTX_FAIL(ctx, DUSP_ERR_ARG, "out of memory");
    prog->ctx = ctx;
    prog->requested_engine = engine;
    prog->resumable = resumable;
    std::string err;
    if (!dusp::compile(desc, n_words, prog->P, err, /*continuation=*/false)) return compile_status(ctx, "dusp_program_build", err);
    if (ctx->table_len && ctx->table_len != (uint32_t)prog->P.g.sample_rate + 1)
        CTX_FAIL(ctx, DUSP_ERR_STATE, "dusp_program_build: uploaded wave tables do not match the program's sample rate");
    if (int rc = finish_build(prog.get())) return rc;
    HIP_TRY(ctx, hipEventCreate(&prog->ev0));
    HIP_TRY(ctx, hipEventCreate(&prog->ev1));
    *out = prog.release();
    return DUSP_OK;
}

int dusp_program_continue(dusp_program *prog, const double *desc, size_t n_words) {
    if (!prog) return DUSP_ERR_ARG;
    dusp_ctx *ctx = prog->ctx;
    if (!prog->rendered) CTX_FAIL(ctx, DUSP_ERR_STATE, "dusp_program_continue: nothing has been rendered yet");
    dusp::Program next;
    std::string err;
    if (!dusp::compile(desc, n_words, next, err, /*continuation=*/true)) return compile_status(ctx, "dusp_program_continue", err);
    const dusp::Program &P = prog->P;
    // same circuit: same units, wiring, channel counts, buffers, state slots and rings — only constants and state may differ
    bool same = next.g.units.size() == P.g.units.size() && next.ops.size() == P.ops.size() && next.n_bufs == P.n_bufs &&
                next.ring_samples == P.ring_samples && next.out_bufs == P.out_bufs && next.g.n_params == P.g.n_params &&
                next.g.sample_rate == P.g.sample_rate && next.init_state.size() == P.init_state.size() &&
                next.dev_rings.size() == P.dev_rings.size();
    for (size_t k = 0; same && k < P.g.units.size(); k++) same = next.g.units[k].op == P.g.units[k].op && next.g.units[k].n_out == P.g.units[k].n_out;
    for (size_t k = 0; same && k < P.ops.size(); k++) {
        const dusp::DevOp &a = P.ops[k], &b = next.ops[k];
        same = a.op == b.op && a.unit == b.unit && a.out_buf == b.out_buf && a.state_slot == b.state_slot && a.ring_base == b.ring_base &&
               a.ring_len == b.ring_len && a.n_in == b.n_in;
        for (int j = 0; same && j < dusp::kMaxIn; j++)  // connections must stay connections to the same buffer
            same = (a.in[j].kind == dusp::SRC_BUF) == (b.in[j].kind == dusp::SRC_BUF) && (a.in[j].kind != dusp::SRC_BUF || a.in[j].idx == b.in[j].idx);
    }
    if (!same) CTX_FAIL(ctx, DUSP_ERR_ARG, "dusp_program_continue: the descriptor does not describe the circuit this program was built from");
    if (next.g.clock0 != prog->next_clock)
        CTX_FAIL(ctx, DUSP_ERR_STATE, "dusp_program_continue: descriptor clock " + std::to_string(next.g.clock0) + " does not follow the rendered clock " +
                                          std::to_string(prog->next_clock));
    const bool persistent = next.ring_samples != 0 || !next.feed_forward;
    if (persistent && !(prog->resumable && (prog->engine == DUSP_ENGINE_CHUNK || prog->engine == DUSP_ENGINE_WAVE)))
        CTX_FAIL(ctx, DUSP_ERR_STATE, "dusp_program_continue: a circuit with delay lines / feedback has to be built with DUSP_ENGINE_RESUMABLE");
    const int engine_before = prog->engine;
    // The wave engine writes every Delay slot once, with its final value; after a change of the delay new taps can land on
    // slots that already hold data, which only the chunk engine's read-modify-write protocol accumulates like the reference.
    for (size_t k = 0; k < P.ops.size(); k++)
        if (P.ops[k].op == dusp::OP_DELAY) {
            const dusp::DevOperand &a = P.ops[k].in[1], &b = next.ops[k].in[1];
            if (a.kind != b.kind || a.idx != b.idx || std::memcmp(&a.cval, &b.cval, sizeof(float)) != 0) prog->delay_changed = true;
        }
    prog->P = std::move(next);
    if (int rc = finish_build(prog)) return rc;
    prog->keep_memory = persistent;
    if (persistent && engine_before == DUSP_ENGINE_WAVE && prog->engine == DUSP_ENGINE_CHUNK) prog->migrate_to_chunk = true;
    return DUSP_OK;
}

void dusp_program_destroy(dusp_program *prog) {
    if (!prog) return;
    (void)hipSetDevice(prog->ctx->device);
    (void)hipStreamSynchronize(prog->ctx->stream);
    prog->d_ops.release();
    prog->d_out_bufs.release();
    prog->d_init.release();
    prog->d_scratch.release();
    prog->d_host_out.release();
    prog->d_saved_bufs.release();
    prog->d_rings_wave.release();
    prog->d_host_par.release();
    prog->d_host_frames.release();
    prog->d_host_in.release();
    prog->d_rings.release();
    prog->d_state.release();
    prog->d_fused_state.release();
    prog->d_recs.release();
    prog->d_sum_voices.release();
    if (prog->ev0) (void)hipEventDestroy(prog->ev0);
    if (prog->ev1) (void)hipEventDestroy(prog->ev1);
    delete prog;
}

int dusp_program_info_get(const dusp_program *prog, dusp_program_info *info) {
    if (!prog || !info) return DUSP_ERR_ARG;
    std::memset(info, 0, sizeof *info);
    const dusp::Graph &g = prog->P.g;
    info->sample_rate = (uint32_t)g.sample_rate;
    info->chunk_size = (uint32_t)g.chunk;
    info->n_units = (uint32_t)g.units.size();
    info->n_out_channels = (uint32_t)prog->P.out_bufs.size();
    info->n_params = (uint32_t)g.n_params;
    info->engine = (uint32_t)prog->engine;
    info->n_device_ops = (uint32_t)prog->P.ops.size();
    info->n_inputs = (uint32_t)g.n_inputs;
    if (prog->engine == DUSP_ENGINE_FUSED) std::snprintf(info->shape, sizeof info->shape, "%s", prog->fused.shape.c_str());
    if (prog->engine == DUSP_ENGINE_LOOP)
        std::snprintf(info->shape, sizeof info->shape, prog->loop_two_stage ? "loop(osc,sum,delay,filter,gain) two-stage" : "loop(osc,sum,delay,filter,gain)");
    if (prog->engine == DUSP_ENGINE_WAVE)
        std::snprintf(info->shape, sizeof info->shape, "%s, %d chunk buffers in LDS", prog->P.feed_forward ? "feed-forward" : "feedback", prog->wave.n_slots);
    return DUSP_OK;
}

static int check_tables(dusp_program *prog) {
    dusp_ctx *ctx = prog->ctx;
    for (const auto &u : prog->P.g.units)
        if (u.op == dusp::OP_OSC || u.op == dusp::OP_MULTI_OSC || u.op == dusp::OP_SHAPE) {
            const int w = (int)u.attrs[0];
            if (!ctx->d_tables || !ctx->table_set[w])
                CTX_FAIL(ctx, DUSP_ERR_STATE, "render: wave table " + std::to_string(w) + " has not been uploaded (dusp_table_upload)");
            if (ctx->table_len != (uint32_t)prog->P.g.sample_rate + 1)
                CTX_FAIL(ctx, DUSP_ERR_STATE, "render: wave table length != sample_rate + 1");
        }
    return DUSP_OK;
}

static int render_device(dusp_program *prog, size_t n_instances, size_t n_samples, const float *d_params, const float *d_inputs,
                         float *d_out, void *stream_);

int dusp_render_device(dusp_program *prog, size_t n_instances, size_t n_samples, const float *d_params, float *d_out,
                       void *stream_) {
    if (!prog) return DUSP_ERR_ARG;
    if (prog->P.g.n_inputs > 0)
        CTX_FAIL(prog->ctx, DUSP_ERR_ARG, "render: the program reads host-generated input streams; use dusp_render_device_inputs / dusp_render_host_inputs");
    return render_device(prog, n_instances, n_samples, d_params, nullptr, d_out, stream_);
}

int dusp_render_device_inputs(dusp_program *prog, size_t n_instances, size_t n_samples, const float *d_params, const float *d_inputs,
                              float *d_out, void *stream_) {
    if (!prog) return DUSP_ERR_ARG;
    if (prog->P.g.n_inputs > 0 && !d_inputs) CTX_FAIL(prog->ctx, DUSP_ERR_ARG, "render: the program has input streams but d_inputs is NULL");
    return render_device(prog, n_instances, n_samples, d_params, d_inputs, d_out, stream_);
}

static int render_device(dusp_program *prog, size_t n_instances, size_t n_samples, const float *d_params, const float *d_inputs,
                         float *d_out, void *stream_) {
    dusp_ctx *ctx = prog->ctx;
    const dusp::Program &P = prog->P;
    if (n_instances < 1 || n_instances > (1u << 24)) CTX_FAIL(ctx, DUSP_ERR_ARG, "render: n_instances must be in [1, 2^24]");
    if (n_samples < 1 || n_samples > (1ull << 31)) CTX_FAIL(ctx, DUSP_ERR_ARG, "render: n_samples must be in [1, 2^31]");
    if (!d_out) CTX_FAIL(ctx, DUSP_ERR_ARG, "render: d_out is NULL");
    if (P.g.n_params > 0 && !d_params) CTX_FAIL(ctx, DUSP_ERR_ARG, "render: program has parameters but d_params is NULL");
    if (int rc = check_tables(prog)) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = stream_ ? (hipStream_t)stream_ : ctx->stream;
    const uint32_t n_inst = (uint32_t)n_instances;
    const uint32_t n_chunks = (uint32_t)((n_samples + dusp::kChunk - 1) / dusp::kChunk);

    if (prog->engine == DUSP_ENGINE_FUSED) {
        dusp::FusedLaunch L{};
        L.params = d_params;
        L.tables = ctx->d_tables;
        L.table_stride = ctx->table_stride;
        L.out = d_out;
        L.n_inst = n_inst;
        L.n_samples = n_samples;
        L.n_chunks = n_chunks;
        L.sample_rate = (uint32_t)P.g.sample_rate;
        L.n_cus = ctx->n_cus;
        L.table_antisym = ctx->table_antisym[prog->fused.table_id];
        L.table_finite = ctx->table_finite[prog->fused.table_id];
        L.table_fx32_ok = ctx->table_fx32_ok[prog->fused.table_id];
        HIP_TRY(ctx, prog->d_recs.ensure(n_inst));
        L.recs = prog->d_recs.p;
        HIP_TRY(ctx, prog->d_fused_state.ensure((size_t)std::max(1, prog->fused.n_state_words) * n_inst));
        L.end_state = prog->d_fused_state.p;
        if (prog->fused.kind == dusp::FUSED_SUMCHAIN) {
            if (!L.table_fx32_ok) CTX_FAIL(ctx, DUSP_ERR_UNSUPPORTED, "render: wave table has entries below 2^-20; build this program with DUSP_ENGINE_CHUNK");
            // blocks of 8 groups when that still leaves every wave slot an item, else 4
            const uint64_t groups = (n_samples + dusp::kChunk - 1) / dusp::kChunk;
            const uint64_t slots = (uint64_t)ctx->n_cus * 16;
            int gb = (uint64_t)n_inst * ((groups + 7) / 8) >= slots ? 8 : 4;
            if ((groups + gb - 1) / gb > 65535) gb = 8;
            if ((groups + gb - 1) / gb > 65535) CTX_FAIL(ctx, DUSP_ERR_UNSUPPORTED, "render: too many samples for the fused sum chain; use DUSP_ENGINE_CHUNK");
            dusp::build_sum_voices(prog->fused, (uint32_t)P.g.sample_rate, gb, (uint64_t)n_chunks * dusp::kChunk, prog->h_sum_voices, prog->h_sum_end);
            HIP_TRY(ctx, prog->d_sum_voices.ensure(prog->h_sum_voices.size()));
            HIP_TRY(ctx, hipMemcpyAsync(prog->d_sum_voices.p, prog->h_sum_voices.data(), prog->h_sum_voices.size() * sizeof(dusp::SumVoice), hipMemcpyHostToDevice, stream));
            std::vector<double> end((size_t)prog->fused.n_state_words * n_inst);
            for (int w = 0; w < prog->fused.n_state_words; w++)
                for (uint32_t i = 0; i < n_inst; i++) end[(size_t)w * n_inst + i] = prog->h_sum_end[(size_t)w];
            HIP_TRY(ctx, hipMemcpyAsync(prog->d_fused_state.p, end.data(), end.size() * sizeof(double), hipMemcpyHostToDevice, stream));
            HIP_TRY(ctx, hipStreamSynchronize(stream));  // the staging vectors above are reused by the next call
            HIP_TRY(ctx, hipEventRecord(prog->ev0, stream));
            HIP_TRY(ctx, dusp::launch_sumchain(prog->fused, L, prog->d_sum_voices.p, gb, stream));
            HIP_TRY(ctx, hipEventRecord(prog->ev1, stream));
        } else {
            HIP_TRY(ctx, hipEventRecord(prog->ev0, stream));
            HIP_TRY(ctx, dusp::launch_fused(prog->fused, L, stream));
            HIP_TRY(ctx, hipEventRecord(prog->ev1, stream));
        }
        prog->last_n_inst = n_inst;
        prog->rendered = true;
        prog->h_state_valid = false;
        prog->next_clock = P.g.clock0 + (int64_t)n_chunks * dusp::kChunk;
        return DUSP_OK;
    }

    const uint32_t n_pad = (n_inst + 63u) & ~63u;
    const size_t n_slots = P.init_state.size();
    if (prog->engine == DUSP_ENGINE_WAVE) {
        HIP_TRY(ctx, prog->d_state.ensure(std::max<size_t>(1, n_slots) * n_pad));
        dusp::WaveArgs w{};
        w.ops = prog->d_ops.p;
        w.out_bufs = prog->d_out_bufs.p;
        w.params = d_params;
        w.tables = ctx->d_tables;
        w.inputs = d_inputs;
        w.out = d_out;
        w.state = prog->d_state.p;
        w.init_state = prog->d_init.p;
        w.n_samples = n_samples;
        w.n_ops = (uint32_t)P.ops.size();
        w.n_out = (uint32_t)P.out_bufs.size();
        w.n_inst = n_inst;
        w.n_pad = n_pad;
        w.n_bufs = (uint32_t)prog->wave.n_slots;
        w.n_state_ops = (uint32_t)prog->wave.n_state_ops;
        w.n_groups = n_chunks;
        w.sample_rate = (uint32_t)P.g.sample_rate;
        w.table_stride = ctx->table_stride;
        w.vec4_ok = (n_samples % 4 == 0) && (((uintptr_t)d_out & 15) == 0);
        w.lds_table_id = prog->wave.lds_table_id;
        w.clock0 = (uint64_t)P.g.clock0;
        w.has_filter = prog->wave.has_filter ? 1u : 0u;
        w.scratch_bytes = (uint32_t)prog->wave.scratch_bytes;
        w.ring_events = prog->wave.ring_events ? 1u : 0u;
        w.ext_units = (uint32_t)prog->wave.ext_units;
        w.n_params = (uint32_t)P.g.n_params;
        w.ring_samples = (uint64_t)P.ring_samples;
        const bool resume = prog->keep_memory;
        if (resume && n_inst != prog->last_n_inst) CTX_FAIL(ctx, DUSP_ERR_STATE, "render: the instance count cannot change while a program is being continued");
        if (P.ring_samples && !resume) {  // Delay rings start as zeros (Delay.js:14); wave-engine layout [instance][slot]
            HIP_TRY(ctx, prog->d_rings.ensure((size_t)P.ring_samples * n_pad));
            HIP_TRY(ctx, hipMemsetAsync(prog->d_rings.p, 0, (size_t)P.ring_samples * n_pad * sizeof(float), stream));
        }
        w.rings = prog->d_rings.p;
        w.resume = resume ? 1u : 0u;
        w.save_bufs = (prog->resumable && prog->persistent) ? 1u : 0u;
        if (w.save_bufs) {
            HIP_TRY(ctx, prog->d_saved_bufs.ensure((size_t)std::max(1, P.n_bufs) * dusp::kChunk * n_inst));
            w.saved_bufs = prog->d_saved_bufs.p;
        }
        prog->keep_memory = false;
        // Few instances, long render: cut time into segments so that the whole chip works on it (wave_engine.hip).
        w.n_seg = 1;
        w.seg_groups = n_chunks;
        w.max_osc_level = prog->wave.max_osc_level;
        if (prog->wave.splittable) {
            const char *knob = getenv("DUSP_WAVE_SEGMENTS");  // 0 / 1: off; n: force n segments
            const uint64_t target = (uint64_t)ctx->n_cus * 8;  // wavefronts that fill the chip
            uint64_t n_seg = n_inst >= target ? 1 : std::min<uint64_t>(target / n_inst, n_chunks / 8);
            if (knob) n_seg = (uint64_t)std::max(0, atoi(knob));
            n_seg = std::max<uint64_t>(1, std::min<uint64_t>(n_seg, n_chunks));
            if (n_seg > 1) {
                w.seg_groups = (uint32_t)((n_chunks + n_seg - 1) / n_seg);
                w.n_seg = (uint32_t)((n_chunks + w.seg_groups - 1) / w.seg_groups);  // no empty segments
            }
            if (w.n_seg > 1) {
                const size_t per = (size_t)w.n_ops * n_inst * w.n_seg;
                HIP_TRY(ctx, prog->d_seg.ensure(2 * per));
                w.seg_sum = prog->d_seg.p;
                w.seg_start = prog->d_seg.p + per;
            }
        }
        const bool lds_ok = w.lds_table_id >= 0 && ctx->table_antisym[w.lds_table_id] && P.g.sample_rate % 2 == 0;
        HIP_TRY(ctx, hipEventRecord(prog->ev0, stream));
        HIP_TRY(ctx, dusp::launch_wave_engine(w, lds_ok, stream));
        HIP_TRY(ctx, hipEventRecord(prog->ev1, stream));
        prog->last_n_inst = n_inst;
        prog->last_n_pad = n_pad;
        prog->rendered = true;
        prog->h_state_valid = false;
        prog->next_clock = P.g.clock0 + (int64_t)n_chunks * dusp::kChunk;
        return DUSP_OK;
    }
    if (prog->keep_memory) {  // continuing: chunk buffers and rings hold what the previous segment left
        if (n_inst != prog->last_n_inst) CTX_FAIL(ctx, DUSP_ERR_STATE, "render: the instance count cannot change while a program is being continued");
        if (prog->migrate_to_chunk) {  // the chain ran on the wave engine so far: move its memory into this engine's layout
            std::swap(prog->d_rings_wave.p, prog->d_rings.p);
            std::swap(prog->d_rings_wave.cap, prog->d_rings.cap);
            HIP_TRY(ctx, prog->d_scratch.ensure((size_t)std::max(1, P.n_bufs) * dusp::kChunk * n_pad));
            HIP_TRY(ctx, prog->d_rings.ensure(std::max<size_t>(1, (size_t)P.ring_samples) * n_pad));
            HIP_TRY(ctx, hipMemsetAsync(prog->d_scratch.p, 0, (size_t)std::max(1, P.n_bufs) * dusp::kChunk * n_pad * sizeof(float), stream));
            if (P.ring_samples) HIP_TRY(ctx, hipMemsetAsync(prog->d_rings.p, 0, (size_t)P.ring_samples * n_pad * sizeof(float), stream));
            HIP_TRY(ctx, dusp::launch_wave_to_chunk(prog->d_rings_wave.p, prog->d_rings.p, (uint64_t)P.ring_samples, prog->d_saved_bufs.p, prog->d_scratch.p,
                                                    (uint32_t)P.n_bufs, n_inst, n_pad, stream));
            prog->migrate_to_chunk = false;
        }
    } else {
        HIP_TRY(ctx, prog->d_scratch.ensure((size_t)std::max(1, P.n_bufs) * dusp::kChunk * n_pad));
        HIP_TRY(ctx, prog->d_state.ensure(std::max<size_t>(1, n_slots) * n_pad));
        HIP_TRY(ctx, prog->d_rings.ensure(std::max<size_t>(1, (size_t)P.ring_samples) * n_pad));
        // outlets' chunks and all rings start as zeros (SignalChunk.js:7, Delay.js:14, CircleBuffer.js:12)
        HIP_TRY(ctx, hipMemsetAsync(prog->d_scratch.p, 0, (size_t)std::max(1, P.n_bufs) * dusp::kChunk * n_pad * sizeof(float), stream));
        if (P.ring_samples) HIP_TRY(ctx, hipMemsetAsync(prog->d_rings.p, 0, (size_t)P.ring_samples * n_pad * sizeof(float), stream));
    }
    prog->keep_memory = false;
    HIP_TRY(ctx, dusp::launch_state_init(prog->d_state.p, prog->d_init.p, (uint32_t)n_slots, n_pad, stream));

    dusp::ChunkArgs a{};
    a.ops = prog->d_ops.p;
    a.out_bufs = prog->d_out_bufs.p;
    a.scratch = prog->d_scratch.p;
    a.state = prog->d_state.p;
    a.rings = prog->d_rings.p;
    a.params = d_params;
    a.tables = ctx->d_tables;
    a.inputs = d_inputs;
    a.out = d_out;
    a.n_samples = n_samples;
    a.clock0 = P.g.clock0;
    a.n_ops = (uint32_t)P.ops.size();
    a.n_out = (uint32_t)P.out_bufs.size();
    a.n_inst = n_inst;
    a.n_pad = n_pad;
    a.n_chunks = n_chunks;
    a.sample_rate = (uint32_t)P.g.sample_rate;
    a.table_stride = ctx->table_stride;
    a.flags = (prog->resumable && prog->persistent) ? dusp::kChunkFlagResumable : 0u;
    a.n_warm = (uint32_t)P.warm_ops.size();
    for (uint32_t k = 0, at = a.n_ops; k < a.n_warm; k++) {
        a.warm_first[k] = at;
        a.warm_n[k] = (uint32_t)P.warm_ops[k].size();
        at += a.warm_n[k];
    }
    HIP_TRY(ctx, hipEventRecord(prog->ev0, stream));
    if (prog->engine == DUSP_ENGINE_LOOP) {
        const int w = prog->loop.osc.attr;
        if (prog->loop_two_stage)
            HIP_TRY(ctx, dusp::launch_loop2_engine(a, prog->loop, ctx->table_antisym[w] && P.g.sample_rate % 2 == 0, stream));
        else
            HIP_TRY(ctx, dusp::launch_loop_engine(a, prog->loop, ctx->table_antisym[w] && P.g.sample_rate % 2 == 0, ctx->n_cus, stream));
    } else
        HIP_TRY(ctx, dusp::launch_chunk_engine(a, stream));
    HIP_TRY(ctx, hipEventRecord(prog->ev1, stream));
    prog->last_n_inst = n_inst;
    prog->last_n_pad = n_pad;
    prog->rendered = true;
    prog->h_state_valid = false;
    prog->next_clock = P.g.clock0 + (int64_t)n_chunks * dusp::kChunk;
    return DUSP_OK;
}

static int render_host(dusp_program *prog, size_t n_instances, size_t n_samples, const float *h_params, const float *h_inputs, float *h_out,
                       bool interleaved);

int dusp_render_host(dusp_program *prog, size_t n_instances, size_t n_samples, const float *h_params, float *h_out) {
    return render_host(prog, n_instances, n_samples, h_params, nullptr, h_out, false);
}

int dusp_render_host_interleaved(dusp_program *prog, size_t n_instances, size_t n_samples, const float *h_params, float *h_out) {
    return render_host(prog, n_instances, n_samples, h_params, nullptr, h_out, true);
}

int dusp_render_host_inputs(dusp_program *prog, size_t n_instances, size_t n_samples, const float *h_params, const float *h_inputs,
                            float *h_out, int interleaved) {
    if (!prog) return DUSP_ERR_ARG;
    if (prog->P.g.n_inputs > 0 && !h_inputs) CTX_FAIL(prog->ctx, DUSP_ERR_ARG, "render: the program has input streams but h_inputs is NULL");
    return render_host(prog, n_instances, n_samples, h_params, h_inputs, h_out, interleaved != 0);
}

int dusp_interleave_device(dusp_ctx *ctx, const float *d_planar, size_t n_instances, size_t n_channels, size_t n_samples, float *d_interleaved,
                           void *stream_) {
    if (!ctx) return DUSP_ERR_ARG;
    if (!d_planar || !d_interleaved) CTX_FAIL(ctx, DUSP_ERR_ARG, "dusp_interleave_device: NULL buffer");
    if (n_channels < 1 || n_channels > 64 || n_instances < 1 || n_samples < 1 || n_samples > (1ull << 31) ||
        (n_samples + 127) / 128 * n_instances > 0x7fffffffull)
        CTX_FAIL(ctx, DUSP_ERR_ARG, "dusp_interleave_device: need 1..64 channels and at most 2^31 tiles of 128 frames");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, dusp::launch_interleave(d_planar, d_interleaved, (uint32_t)n_instances, (uint32_t)n_channels, n_samples,
                                         stream_ ? (hipStream_t)stream_ : ctx->stream));
    return DUSP_OK;
}

static int render_host(dusp_program *prog, size_t n_instances, size_t n_samples, const float *h_params, const float *h_inputs, float *h_out,
                       bool interleaved) {
    if (!prog) return DUSP_ERR_ARG;
    dusp_ctx *ctx = prog->ctx;
    if (!h_out) CTX_FAIL(ctx, DUSP_ERR_ARG, "render: h_out is NULL");
    const size_t n_par = (size_t)prog->P.g.n_params * n_instances;
    if (n_par && !h_params) CTX_FAIL(ctx, DUSP_ERR_ARG, "render: program has parameters but h_params is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t n_out = n_instances * prog->P.out_bufs.size() * n_samples;
    // staging buffers live with the program (grown on demand): a segmented render calls this hundreds of times a second
    HIP_TRY(ctx, prog->d_host_out.ensure(std::max<size_t>(1, n_out)));
    float *d_out = prog->d_host_out.p, *d_par = nullptr, *d_frames = nullptr;
    int rc = DUSP_OK;
    hipError_t e = hipSuccess;
    if (n_par) {
        HIP_TRY(ctx, prog->d_host_par.ensure(n_par));
        d_par = prog->d_host_par.p;
        e = hipMemcpyAsync(d_par, h_params, n_par * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    }
    const size_t n_in = (size_t)prog->P.g.n_inputs * n_instances * n_samples;
    float *d_in = nullptr;
    if (n_in && !h_inputs) CTX_FAIL(ctx, DUSP_ERR_ARG, "render: the program reads host-generated input streams; use dusp_render_host_inputs");
    if (n_in && e == hipSuccess) {
        HIP_TRY(ctx, prog->d_host_in.ensure(n_in));
        d_in = prog->d_host_in.p;
        e = hipMemcpyAsync(d_in, h_inputs, n_in * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    }
    if (e == hipSuccess) {
        rc = render_device(prog, n_instances, n_samples, d_par, d_in, d_out, ctx->stream);
        const size_t n_ch = prog->P.out_bufs.size();
        if (rc == DUSP_OK && interleaved && n_ch > 1) {  // frames: transpose on the device, then download those
            HIP_TRY(ctx, prog->d_host_frames.ensure(n_out));
            d_frames = prog->d_host_frames.p;
            rc = dusp_interleave_device(ctx, d_out, n_instances, n_ch, n_samples, d_frames, ctx->stream);
        }
        if (rc == DUSP_OK) {
            e = hipMemcpyAsync(h_out, d_frames ? d_frames : d_out, n_out * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        }
    }
    if (rc != DUSP_OK) return rc;
    if (e != hipSuccess) CTX_FAIL(ctx, DUSP_ERR_HIP, std::string("HIP error: ") + hipGetErrorString(e));
    return DUSP_OK;
}

int dusp_state_download(dusp_program *prog, size_t instance, size_t unit, double *out, size_t cap) {
    if (!prog) return DUSP_ERR_ARG;
    dusp_ctx *ctx = prog->ctx;
    if (!prog->rendered) CTX_FAIL(ctx, DUSP_ERR_STATE, "dusp_state_download: nothing has been rendered yet");
    const dusp::Graph &g = prog->P.g;
    if (unit >= g.units.size() || instance >= prog->last_n_inst || !out)
        CTX_FAIL(ctx, DUSP_ERR_ARG, "dusp_state_download: unit / instance out of range");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const dusp::UnitDesc &u = g.units[unit];
    std::vector<double> words;
    auto or0 = [](double v) { return (v != v || v == 0) ? 0.0 : v; };
    // one device-to-host copy of the whole state array per render, however many units are read back afterwards
    const bool fused = prog->engine == DUSP_ENGINE_FUSED;
    const size_t stride = fused ? prog->last_n_inst : prog->last_n_pad;
    if (!prog->h_state_valid) {
        const size_t rows = fused ? (size_t)std::max(1, prog->fused.n_state_words) : std::max<size_t>(1, prog->P.init_state.size());
        prog->h_state.resize(rows * stride);
        HIP_TRY(ctx, hipMemcpy(prog->h_state.data(), fused ? prog->d_fused_state.p : prog->d_state.p, rows * stride * sizeof(double),
                               hipMemcpyDeviceToHost));
        prog->h_state_valid = true;
    }
    if (fused) {
        const int first = prog->fused.unit_state_first[unit], n = prog->fused.unit_state_count[unit];
        for (int k = 0; k < n; k++) words.push_back(prog->h_state[(size_t)(first + k) * stride + instance]);
    } else {
        auto rd = [&](int slot, double &v) {
            v = prog->h_state[(size_t)slot * stride + instance];
            return hipSuccess;
        };
        const int n_ch = (u.op == dusp::OP_FILTER) ? u.n_out : 1;
        const int per = u.op == dusp::OP_DELAY ? 0 : u.slots_per_ch;  // Delay's slot is engine-internal, not unit state
        // (FixedDelay / CombFilter / AllPass / ReadBackDelay: one word, the ring position; MonoDelay: none)
        if (u.op == dusp::OP_SAMPLE_RATE_REDUX) {  // [timeSinceLastUpdate, n, held value per channel]; `val` is `[0]` until the first update
            double since;
            HIP_TRY(ctx, rd(u.first_slot, since));
            const int n_val = std::isinf(since) ? 1 : u.n_out;
            words.push_back(since);
            words.push_back((double)n_val);
            for (int c = 0; c < n_val; c++) {
                double v;
                HIP_TRY(ctx, rd(u.first_slot + c * per + 1, v));
                words.push_back(v);
            }
        } else if (u.op == dusp::OP_MULTI_OSC) {  // [n, phase per channel]
            words.push_back((double)u.n_out);
            for (int c = 0; c < u.n_out; c++) {
                double v;
                HIP_TRY(ctx, rd(u.first_slot + c * per, v));
                words.push_back(v);
            }
        } else if (u.op == dusp::OP_FILTER) {
            for (int k = 0; k < 7; k++) {
                double v;
                HIP_TRY(ctx, rd(u.first_slot + k, v));
                words.push_back(v);
            }
            words.push_back((double)n_ch);
            for (int c = 0; c < n_ch; c++)
                for (int k = 7; k < 11; k++) {
                    double v;
                    HIP_TRY(ctx, rd(u.first_slot + c * per + k, v));
                    words.push_back(or0(v));
                }
        } else {
            for (int k = 0; k < per; k++) {
                double v;
                HIP_TRY(ctx, rd(u.first_slot + k, v));
                words.push_back(v);
            }
        }
    }
    for (size_t k = 0; k < words.size() && k < cap; k++) out[k] = words[k];
    return (int)words.size();
}

int dusp_last_kernel_ms(dusp_program *prog, float *ms) {
    if (!prog || !ms) return DUSP_ERR_ARG;
    dusp_ctx *ctx = prog->ctx;
    if (!prog->rendered) CTX_FAIL(ctx, DUSP_ERR_STATE, "dusp_last_kernel_ms: nothing has been rendered yet");
    HIP_TRY(ctx, hipEventSynchronize(prog->ev1));
    HIP_TRY(ctx, hipEventElapsedTime(ms, prog->ev0, prog->ev1));
    return DUSP_OK;
}

int dusp_fill_device(dusp_ctx *ctx, float *d_out, size_t n_floats, float value, void *stream_) {
    if (!ctx) return DUSP_ERR_ARG;
    if (!d_out || (n_floats & 3) || ((uintptr_t)d_out & 15)) CTX_FAIL(ctx, DUSP_ERR_ARG, "dusp_fill_device: need a 16-byte aligned buffer of 4k floats");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, dusp::launch_fill(d_out, n_floats, value, stream_ ? (hipStream_t)stream_ : ctx->stream));
    return DUSP_OK;
}

}  // extern "C"
