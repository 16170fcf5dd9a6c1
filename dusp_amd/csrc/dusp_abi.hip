// dusp_abi.hip — implementation of the C ABI declared in include/dusp_hip.h.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dusp_hip.h"
#include "device_types.hpp"
#include "device_util.hpp"
#include "fused_plan.hpp"
#include "jit_codegen.hpp"
#include "jit_engine.hpp"
#include "program.hpp"
#include "ring_windows.hpp"
#include "table_checks.hpp"

namespace dusp {
hipError_t launch_chunk_engine(const ChunkArgs &a, hipStream_t stream);
hipError_t launch_state_init(double *state, const double *init, uint32_t n_slots, uint32_t n_pad, hipStream_t stream);
hipError_t launch_fill(float *out, size_t n_floats, float value, hipStream_t stream);
hipError_t launch_wave_to_chunk(const float *wave_rings, float *chunk_rings, uint64_t ring_samples, const float *saved_bufs, float *chunk_scratch,
                                uint32_t n_bufs, uint32_t n_inst, uint32_t n_pad, hipStream_t stream);
hipError_t launch_chunk_to_wave(const float *chunk_rings, float *wave_rings, uint64_t ring_samples, const float *chunk_scratch, float *saved_bufs, uint32_t n_bufs,
                                uint32_t n_inst, uint32_t n_pad, const double *state, double *init_state, uint32_t n_slots, hipStream_t stream);
hipError_t launch_interleave(const float *d_planar, float *d_out, uint32_t n_instances, uint32_t n_channels, uint64_t n_samples, hipStream_t stream);
hipError_t launch_fused(const FusedPlan &plan, const FusedLaunch &L, hipStream_t stream);
hipError_t launch_wave_engine(WaveArgs A, bool lds_table_ok, int max_waves_cap, hipStream_t stream);
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
    int table_bound[dusp::kNumTables] = {1000, 1000, 1000, 1000, 1000, 1000, 1000, 1000, 1000};  // every |entry| <= 2^bound (1000: not finite / not set)
    int table_delta[dusp::kNumTables] = {0, 0, 0, 0, 0};  // the lerp's delta form (device_util.hpp lerp_delta): 2 every T[t+1] - T[t] is an f32, 1 exact in f64, 0 neither (or not finite)
    // TABLE_FORM_*: the uploaded table equals a closed form of the index (saw, square, triangle) or of the sine table's entry
    // (8bit) on EVERY entry, bit for bit — checked at upload — so kernels may evaluate it instead of gathering (device_util.hpp)
    int table_form[dusp::kNumTables] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<float> h_tables[2];  // host copies of table 0 (sine) and table 4 (8bit) for that check
    int n_cus = 256;
    dusp::Knobs knobs;  // A/B switches, read from the environment once (dusp_ctx_create)
    // pinned host buffers handed out by dusp_host_alloc (in_use) or waiting for reuse
    struct HostBuf { void *p; size_t bytes; bool in_use; };
    std::vector<HostBuf> host_pool;
    std::mutex host_pool_mutex;  // dusp_host_free may come from a finalizer thread (a garbage collector) while the owner allocates
    // bumped by every dusp_table_upload: compiled kernels are generated against what the context knows about its tables
    // (closed forms, antisymmetry, the LDS image's table), so a program's generated text is dropped when this has moved on
    uint64_t table_generation = 0;
    bool tables_guarded = false;
};

// DUSP_GUARD=1 (tests): every device allocation of the library carries guard bytes behind its end, filled with a pattern and
// checked after each render — a kernel that writes past the end of a workspace (state, rings, parked chunks, staging PCM, the
// tables) fails THAT render with a message instead of corrupting whatever the allocator placed next to it.
static size_t g_guard_bytes = 0;
constexpr unsigned char kGuardPattern = 0xA5;

static bool guard_intact(const void *end_of_payload) {
    unsigned char tail[4096];
    const size_t n = std::min(g_guard_bytes, sizeof tail);
    if (hipMemcpy(tail, end_of_payload, n, hipMemcpyDeviceToHost) != hipSuccess) return false;
    for (size_t i = 0; i < n; i++)
        if (tail[i] != kGuardPattern) return false;
    return true;
}

template <class T>
struct DevBuf {  // grow-only device allocation
    T *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc((void **)&p, n * sizeof(T) + g_guard_bytes);
        if (e == hipSuccess && g_guard_bytes) e = hipMemset((char *)p + n * sizeof(T), kGuardPattern, g_guard_bytes);
        if (e == hipSuccess) cap = n;
        return e;
    }
    bool intact() const { return !p || !g_guard_bytes || guard_intact((const char *)p + cap * sizeof(T)); }
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
    hipStream_t last_stream = nullptr;  // stream of the most recent render (workspaces and state are ordered on it)
    // WAVE programs the circuit compiler takes (jit_codegen.hpp): generated text per workgroup geometry, constants on the device
    bool jit_ok = false;
    std::string jit_why;
    std::map<std::pair<int, int>, dusp::JitSource> jit_src;  // (wavefronts per workgroup, 8 x instances per wavefront + Filter block) -> kernel text (+ constants, scan list)
    bool jit_consts_uploaded = false;
    int voice_loop = -1;  // the circuit's voices run in a loop on its compiled kernel (jit_codegen.hpp VoicePlan): -1 not looked at yet
    // dusp_render_chain_window: the window the next render of this (sum chain) program is (set for the duration of that call)
    bool chain_on = false, chain_raw = false;
    const float *chain_init = nullptr;
    uint64_t chain_first = 0;
    uint64_t jit_table_generation = 0;  // ctx->table_generation the texts in jit_src were generated against
    int jit_waves = 0, jit_per_wave = 0;  // geometry of the last compiled launch (shown in dusp_program_info.shape)
    unsigned jit_segments = 1;            // ... the time segments it was cut into
    bool jit_voices = false;              // ... its voices ran in a loop (jit_codegen.hpp VoicePlan)
    bool jit_scan = false;                // ... its Filters ran as scans over the chunk (jit_filter_scan_ok)
    DevBuf<float> d_jit_fk;
    DevBuf<double> d_jit_dk;
    DevBuf<int> d_jit_scan;  // [2][n_scans]: state slot, FM level of every scanned oscillator
    // channel counts that grow during the first chunks (Program::warm_ops): those chunks on the chunk engine, the rest on a compiled kernel
    bool handoff_ok = false;
    std::string handoff_why;  // when not: what keeps the settled circuit on the chunk engine
    DevBuf<double> d_handoff_init;   // the unit state the chunk engine left (instance 0), as the compiled kernel's start state
    DevBuf<double> d_warm_records;   // segments that warm up (JitArgs::warm): [Filter stage][segment][8] what the stage held where the segment's own chunks began / ended
    DevBuf<double> d_warm_init;      // ... and the start state of the launch that finishes a render whose check failed
    unsigned warm_redo_from = 0;     // the last render's check failed at this segment: finished sequentially from there (0: it held)
    DevBuf<float> d_handoff_out;     // the two parts' PCM before they are put side by side
    DevBuf<int64_t> d_jit_regime;  // per-instance delays: [slot, ring length, mono] per unit, then the verdicts (render_jit)

    dusp_program() = default;
    dusp_program(const dusp_program &) = delete;
    dusp_program &operator=(const dusp_program &) = delete;
    ~dusp_program() {  // every exit path — a failed build included — gives the device memory back
        if (ctx) (void)hipSetDevice(ctx->device);
        for (DevBuf<float> *b : {&d_scratch, &d_rings, &d_host_out, &d_host_par, &d_host_frames, &d_host_in, &d_saved_bufs, &d_rings_wave, &d_jit_fk}) b->release();
        for (DevBuf<double> *b : {&d_init, &d_state, &d_fused_state, &d_jit_dk}) b->release();
        d_jit_scan.release();
        d_jit_regime.release();
        d_handoff_init.release();
        d_warm_records.release();
        d_warm_init.release();
        d_handoff_out.release();
        d_ops.release();
        d_out_bufs.release();
        d_seg.release();
        d_recs.release();
        d_sum_voices.release();
        if (pin[0]) (void)hipHostFree(pin[0]);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
    }
    // dusp_render_host* into pageable memory: pinned staging tiles, two per copy worker (download_staged)
    float *pin[1] = {nullptr};
    size_t pin_floats = 0;
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

// Exception firewall: nothing thrown inside the library (std::bad_alloc / std::length_error from a container sized by a
// descriptor, ...) may cross the C boundary — it would be std::terminate for the caller.  Every entry point that can
// allocate runs its body through guarded(): the exception becomes a status + message like any other failure.
template <class F>
static int guarded(std::string &err, const char *who, F &&body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        try { err = std::string(who) + ": out of host memory"; } catch (...) {}
        return DUSP_ERR_NOMEM;
    } catch (const std::length_error &e) {
        try { err = std::string(who) + ": size out of range (" + e.what() + ")"; } catch (...) {}
        return DUSP_ERR_ARG;
    } catch (const std::exception &e) {
        try { err = std::string(who) + ": " + e.what(); } catch (...) {}
        return DUSP_ERR_ARG;
    } catch (...) {
        try { err = std::string(who) + ": unknown internal error"; } catch (...) {}
        return DUSP_ERR_ARG;
    }
}

static dusp::Knobs read_knobs() {
    dusp::Knobs k;
    auto num = [](const char *name, int fallback) {
        const char *e = getenv(name);
        return e && *e ? atoi(e) : fallback;
    };
    if (const char *t = getenv("DUSP_FUSED_TABLE")) k.fused_table_global = t[0] == 'g';
    k.fused_R = num("DUSP_FUSED_R", k.fused_R);
    k.fused_items = num("DUSP_FUSED_ITEMS", k.fused_items);
    k.fused_fx32 = num("DUSP_FUSED_FX32", k.fused_fx32);
    k.fused_segmajor = num("DUSP_FUSED_SEGMAJOR", k.fused_segmajor);
    k.wave_segments = num("DUSP_WAVE_SEGMENTS", k.wave_segments);
    k.wave_max_waves = num("DUSP_WAVE_MAX_WAVES", k.wave_max_waves);
    k.wave_jit = num("DUSP_WAVE_JIT", k.wave_jit);
    k.wave_per_wave = num("DUSP_WAVE_PER_WAVE", k.wave_per_wave);
    k.jit_profile = num("DUSP_JIT_PROFILE", k.jit_profile);
    k.filter_fma = num("DUSP_FILTER_FMA", k.filter_fma);
    k.jit_spill_bytes = num("DUSP_JIT_SPILL", k.jit_spill_bytes);
    k.jit_lds_table = num("DUSP_JIT_LDS_TABLE", k.jit_lds_table);
    k.jit_lean = num("DUSP_JIT_LEAN", k.jit_lean);
    k.jit_log = num("DUSP_JIT_LOG", k.jit_log);
    k.ring_window = num("DUSP_RING_WINDOW", k.ring_window);
    k.filter_scan = num("DUSP_FILTER_SCAN", k.filter_scan);
    k.filter_warm = num("DUSP_FILTER_WARM", k.filter_warm);
    k.jit_nt = num("DUSP_JIT_NT", k.jit_nt);
    k.delay_line = num("DUSP_DELAY_LINE", k.delay_line);
    k.jit_rotate = num("DUSP_JIT_ROTATE", k.jit_rotate);
    k.ring_poison = num("DUSP_RING_POISON", k.ring_poison);
    if (const char *f = getenv("DUSP_JIT_FORCE")) {
        int w = 0, r = 0;
        if (std::sscanf(f, "%dx%d", &w, &r) == 2 && w >= 1 && w <= 16 && r >= 1 && r <= 4) k.jit_force_waves = w, k.jit_force_per_wave = r;
    }
    return k;
}

extern "C" {

const char *dusp_version(void) { return "dusp-hip 0.1.0 (gfx950)"; }
int dusp_abi_version(void) { return DUSP_ABI_VERSION; }

int dusp_device_count(void) {
    int count = 0;
    const hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess) {
        try {
            g_error = std::string("no usable HIP device: ") + hipGetErrorString(e) + " (this library has no CPU fallback)";
        } catch (...) {
        }
        return DUSP_ERR_HIP;
    }
    return count;
}

const char *dusp_last_error(const dusp_ctx *ctx) { return ctx ? ctx->err.c_str() : g_error.c_str(); }

int dusp_ctx_create(int device, dusp_ctx **out) {
    return guarded(g_error, "dusp_ctx_create", [&]() -> int {
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
    ctx->knobs = read_knobs();
    if (const char *g = getenv("DUSP_GUARD"))  // (process-wide, set by the first context: allocations made before keep their size)
        if (atoi(g) > 0 && !g_guard_bytes) g_guard_bytes = 4096;
    dusp::jit_configure();  // (the code-object cache directory: read once per process)
    *out = ctx.release();
    return DUSP_OK;
    });
}

void dusp_ctx_destroy(dusp_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamDestroy(ctx->stream);
    }
    if (ctx->d_tables) (void)hipFree(ctx->d_tables);
    for (auto &b : ctx->host_pool) (void)hipHostFree(b.p);
    delete ctx;
}

int dusp_table_upload(dusp_ctx *ctx, int table_id, const float *table, size_t n) {
    if (!ctx) return DUSP_ERR_ARG;
    return guarded(ctx->err, "dusp_table_upload", [&]() -> int {
    if (table_id < 0 || table_id >= dusp::kNumTables || !table || n < 9 || n > (1u << 22) + 1)
        CTX_FAIL(ctx, DUSP_ERR_ARG, "dusp_table_upload: bad table id, pointer or length");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->d_tables) {
        ctx->table_len = (uint32_t)n;
        ctx->table_stride = (uint32_t)((n + 1 + 3) & ~(size_t)3);  // >= n+1 entries (one pad for idx+1), 16-byte rows
        HIP_TRY(ctx, hipMalloc((void **)&ctx->d_tables, sizeof(float) * dusp::kNumTables * ctx->table_stride + g_guard_bytes));
        HIP_TRY(ctx, hipMemset(ctx->d_tables, 0, sizeof(float) * dusp::kNumTables * ctx->table_stride));
        if (g_guard_bytes) HIP_TRY(ctx, hipMemset((char *)ctx->d_tables + sizeof(float) * dusp::kNumTables * ctx->table_stride, kGuardPattern, g_guard_bytes));
        ctx->tables_guarded = g_guard_bytes != 0;
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
    ctx->table_delta[table_id] = dusp::table_delta_class(table, n);
    {
        float top = 0.f;
        for (size_t t = 0; t < n; t++) top = std::max(top, std::fabs(table[t]));
        ctx->table_bound[table_id] = !finite ? 1000 : top == 0.f ? 0 : std::ilogb(top) + 1;
    }
    ctx->table_set[table_id] = true;
    ctx->table_generation++;
    // closed forms (device_util.hpp): every entry has to match, sign of zero included
    auto same_bits = [](float a, float b) { return std::memcmp(&a, &b, sizeof a) == 0; };
    const uint32_t sr = (uint32_t)n - 1;
    ctx->table_form[table_id] = dusp::TABLE_FORM_DATA;
    if (table_id >= 1 && table_id <= 3 && n <= 131073) {
        const int forms[3] = {dusp::TABLE_FORM_SAW, dusp::TABLE_FORM_SQUARE, dusp::TABLE_FORM_TRIANGLE};
        const dusp::TableForm F = dusp::make_table_form(forms[table_id - 1], sr);
        bool ok = F.form != dusp::TABLE_FORM_TRIANGLE || sr % 4 == 0;
        for (uint32_t i = 0; ok && i <= sr; i++) ok = same_bits(dusp::closed_table_entry(F, i), table[i]);
        if (ok) ctx->table_form[table_id] = F.form;
    }
    if (table_id == 0 || table_id == 4) {
        ctx->h_tables[table_id ? 1 : 0].assign(table, table + n);
        ctx->table_form[4] = dusp::TABLE_FORM_DATA;
        const std::vector<float> &sine = ctx->h_tables[0], &bit8 = ctx->h_tables[1];
        if (sine.size() == n && bit8.size() == n && ctx->table_antisym[0] && sr % 2 == 0) {
            // as the kernels see the sine table: the half image in LDS, mirrored with a sign above the middle
            bool ok = true;
            for (uint32_t i = 0; ok && i <= sr; i++) ok = same_bits(dusp::eightbit_of_sine(i > sr / 2 ? -sine[n - i] : sine[i]), bit8[i]);
            if (ok) ctx->table_form[4] = dusp::TABLE_FORM_8BIT;
        }
    }
    return DUSP_OK;
    });
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
    // A circuit with rings or a feedback edge carries device memory from one segment to the next; only the chunk
    // engine keeps all of it (rings, every outlet's previous chunk) in HBM in a layout a later launch can pick up.
    // A circuit with rings or a feedback edge carries device memory from one segment to the next.  The chunk engine keeps
    // all of it in HBM; the wave engine parks its LDS chunk buffers in HBM between launches.  An engine, once chosen, is
    // kept for the whole chain (ring layouts differ) — except that a wave program that stops being plannable migrates to
    // the chunk engine.
    prog->persistent = prog->P.ring_samples != 0 || !prog->P.feed_forward;
    for (dusp::DevOp &op : prog->P.ops)  // continued programs keep their rings in the reference's own state (jit_codegen.hpp kDelayExactRing)
        if (op.op == dusp::OP_DELAY || op.op == dusp::OP_MONO_DELAY) op.pad = (prog->resumable && prog->persistent) ? dusp::kDelayExactRing : 0;
    if (prog->resumable && prog->persistent) {
        if (engine != DUSP_ENGINE_AUTO && engine != DUSP_ENGINE_CHUNK && engine != DUSP_ENGINE_WAVE)
            CTX_FAIL(ctx, DUSP_ERR_UNSUPPORTED, "dusp_program_build: a resumable program with delay lines / feedback runs on DUSP_ENGINE_WAVE or DUSP_ENGINE_CHUNK");
        if (prog->rendered)  // continuing: stay, or fall back to the chunk engine
            engine = (prog->engine == DUSP_ENGINE_WAVE && wavable && !prog->delay_changed) ? DUSP_ENGINE_WAVE : DUSP_ENGINE_CHUNK;
        else if (engine == DUSP_ENGINE_AUTO)
            engine = wavable ? DUSP_ENGINE_WAVE : DUSP_ENGINE_CHUNK;
    }
    // (the hand-written kernels for the feedback voice of BASELINE configs[3] — a one-stage and two two-stage forms, rounds 1 and 2 — are gone since
    // round 4: the kernel compiled for the circuit renders it in 8.0 ms (scan) / 12.0 ms (stage, bit-equal) against their 20.4, and whatever delay the voice has)
    if (engine == DUSP_ENGINE_AUTO) engine = fusable ? DUSP_ENGINE_FUSED : wavable ? DUSP_ENGINE_WAVE : DUSP_ENGINE_CHUNK;
    prog->engine = engine;
    prog->jit_src.clear();
    prog->jit_consts_uploaded = false;
    prog->jit_ok = engine == DUSP_ENGINE_WAVE && ctx->knobs.wave_jit != 0 &&
                   dusp::jit_eligible(prog->P, prog->wave, prog->jit_why);
    // Channel counts that grow during the first chunks keep a program on the chunk engine — for those chunks.  When the SETTLED op list is
    // one the circuit compiler takes, a single circuit's render hands over behind them (render_device: rings, outlets' last chunk and
    // unit state move into the compiled kernel's layout).
    prog->handoff_ok = false;
    prog->handoff_why.clear();
    if (engine == DUSP_ENGINE_CHUNK && !prog->P.warm_ops.empty() && !prog->resumable && prog->requested_engine == DUSP_ENGINE_AUTO && ctx->knobs.wave_jit != 0 &&
        prog->P.g.n_inputs == 0) {
        auto checked = std::move(prog->wave.ramp_checked);
        prog->wave = dusp::WavePlan();
        prog->wave.ramp_checked = std::move(checked);
        std::string why;
        const bool planned = dusp::plan_wave(prog->P, prog->wave, /*will_continue=*/true, /*settled_only=*/true);
        if (!planned) prog->handoff_why = prog->wave.why;
        else if (!dusp::jit_eligible(prog->P, prog->wave, why)) prog->handoff_why = why;
        if (planned && prog->handoff_why.empty()) {
            prog->handoff_ok = true;
            for (size_t k = 0; k < prog->wave.osc_level.size() && k < prog->P.ops.size(); k++)
                if (prog->wave.osc_level[k] >= 0) prog->P.ops[k].d[0] = (double)prog->wave.osc_level[k];
            for (size_t k = 0; k < prog->wave.ramp_fastdiv.size() && k < prog->P.ops.size(); k++)
                if (prog->P.ops[k].op == dusp::OP_RAMP) prog->P.ops[k].attr = prog->wave.ramp_fastdiv[k];
            for (dusp::DevOp &op : prog->P.ops)  // (the kernel continues rings the chunk engine kept in the reference's state)
                if (op.op == dusp::OP_DELAY || op.op == dusp::OP_MONO_DELAY) op.pad = dusp::kDelayExactRing;
        }
    }

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
    return guarded(ctx->err, "dusp_program_build", [&]() -> int {
    if (!out) CTX_FAIL(ctx, DUSP_ERR_ARG, "dusp_program_build: out is NULL");
    *out = nullptr;
    const bool resumable = (engine & DUSP_ENGINE_RESUMABLE) != 0;
    engine &= ~DUSP_ENGINE_RESUMABLE;
    if (engine == DUSP_ENGINE_LOOP) engine = DUSP_ENGINE_AUTO;  // (ABI v7: the loop kernels are gone; a caller that still names them gets what AUTO picks for its circuit)
    if (engine != DUSP_ENGINE_AUTO && engine != DUSP_ENGINE_CHUNK && engine != DUSP_ENGINE_FUSED && engine != DUSP_ENGINE_WAVE)
        CTX_FAIL(ctx, DUSP_ERR_ARG, "dusp_program_build: bad engine");
    std::unique_ptr<dusp_program> prog(new dusp_program);  // (its destructor frees whatever a failing step below has allocated)
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
    });
}

int dusp_program_continue(dusp_program *prog, const double *desc, size_t n_words) {
    if (!prog) return DUSP_ERR_ARG;
    dusp_ctx *ctx = prog->ctx;
    return guarded(ctx->err, "dusp_program_continue", [&]() -> int {
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
    bool delay_changed = prog->delay_changed;
    for (size_t k = 0; k < P.ops.size(); k++)
        if (P.ops[k].op == dusp::OP_DELAY) {
            const dusp::DevOperand &a = P.ops[k].in[1], &b = next.ops[k].in[1];
            if (a.kind != b.kind || a.idx != b.idx || std::memcmp(&a.cval, &b.cval, sizeof(float)) != 0) delay_changed = true;
        }
    // All or nothing: if planning or the upload fails, the program goes back to the circuit it was rendering (host plans AND
    // the device copies of its constants), so a later render never runs a mixture of the two.
    struct Before {
        dusp::Program P; bool delay_changed; int engine;
    } before{std::move(prog->P), prog->delay_changed, prog->engine};
    prog->P = std::move(next);
    prog->delay_changed = delay_changed;
    if (int rc = finish_build(prog)) {
        const std::string why = ctx->err;
        prog->P = std::move(before.P);
        prog->delay_changed = before.delay_changed;
        const int requested = prog->requested_engine;
        prog->requested_engine = before.engine;  // re-plan the old circuit onto the engine it was on
        const int back = finish_build(prog);
        prog->requested_engine = requested;
        ctx->err = back == DUSP_OK ? why : why + " (and the previous program could not be restored: " + ctx->err + ")";
        return rc;
    }
    prog->keep_memory = persistent;
    if (persistent && engine_before == DUSP_ENGINE_WAVE && prog->engine == DUSP_ENGINE_CHUNK) prog->migrate_to_chunk = true;
    return DUSP_OK;
    });
}

void dusp_program_destroy(dusp_program *prog) {
    if (!prog) return;
    (void)hipSetDevice(prog->ctx->device);
    // renders may have gone to a caller's stream: wait for what the last one recorded, not only for the context's own stream
    if (prog->rendered && prog->ev1) (void)hipEventSynchronize(prog->ev1);
    (void)hipStreamSynchronize(prog->ctx->stream);
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
    if (prog->engine == DUSP_ENGINE_WAVE && prog->jit_ok && (prog->jit_waves || !prog->rendered))
    {
        const int at = std::snprintf(info->shape, sizeof info->shape, "%s, compiled kernel: %d units, %dx%d", prog->P.feed_forward ? "feed-forward" : "feedback",
                                     (int)prog->P.ops.size(), prog->jit_waves, prog->jit_per_wave);
        if (at > 0 && (size_t)at < sizeof info->shape && (prog->jit_voices || prog->jit_segments > 1)) {
            if (prog->jit_voices && prog->jit_segments > 1) std::snprintf(info->shape + at, sizeof info->shape - (size_t)at, ", loop, %u seg", prog->jit_segments);
            else if (prog->jit_voices) std::snprintf(info->shape + at, sizeof info->shape - (size_t)at, ", voice loop");
            else if (prog->warm_redo_from) std::snprintf(info->shape + at, sizeof info->shape - (size_t)at, ", %u seg, redo@%u", prog->jit_segments, prog->warm_redo_from);
            else std::snprintf(info->shape + at, sizeof info->shape - (size_t)at, ", %u seg", prog->jit_segments);
        } else if (at > 0 && (size_t)at < sizeof info->shape && prog->jit_scan)
            std::snprintf(info->shape + at, sizeof info->shape - (size_t)at, ", scan");  // (the Filters as scans over the chunk: within the gate's bound, not bit for bit)
    }
    else if (prog->engine == DUSP_ENGINE_WAVE && prog->jit_ok)
        std::snprintf(info->shape, sizeof info->shape, "%s, %d chunk buffers in LDS (kernel compiling)", prog->P.feed_forward ? "feed-forward" : "feedback", prog->wave.n_slots);
    else if (prog->engine == DUSP_ENGINE_WAVE)
        std::snprintf(info->shape, sizeof info->shape, "%s, %d chunk buffers in LDS", prog->P.feed_forward ? "feed-forward" : "feedback", prog->wave.n_slots);
    if (prog->engine == DUSP_ENGINE_CHUNK && !prog->P.warm_ops.empty() && !prog->handoff_ok && !prog->handoff_why.empty())
        std::snprintf(info->shape, sizeof info->shape, "warm-up, then not compiled: %.34s", prog->handoff_why.c_str());
    if (prog->engine == DUSP_ENGINE_CHUNK && prog->handoff_ok && prog->jit_waves)
        std::snprintf(info->shape, sizeof info->shape, "%d warm-up chunks here, then compiled kernel: %d units, %dx%d", (int)prog->P.warm_ops.size(), (int)prog->P.ops.size(),
                      prog->jit_waves, prog->jit_per_wave);
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

int dusp_render_chain_window(dusp_program *prog, uint64_t first_sample, size_t n_samples, const float *d_init, int raw, float *d_out, void *stream_) {
    if (!prog) return DUSP_ERR_ARG;
    if (prog->engine != DUSP_ENGINE_FUSED || prog->fused.kind != dusp::FUSED_SUMCHAIN)
        CTX_FAIL(prog->ctx, DUSP_ERR_UNSUPPORTED, "dusp_render_chain_window: the program is not on the fused sum chain (a Sum.many of constant-f oscillators)");
    if (prog->P.g.n_params > 0) CTX_FAIL(prog->ctx, DUSP_ERR_UNSUPPORTED, "dusp_render_chain_window: programs with per-instance parameters are not chained");
    if (first_sample % 2048 != 0) CTX_FAIL(prog->ctx, DUSP_ERR_ARG, "dusp_render_chain_window: first_sample must be a multiple of 2048");
    if (first_sample > (1ull << 40)) CTX_FAIL(prog->ctx, DUSP_ERR_ARG, "dusp_render_chain_window: first_sample out of range");
    if ((((uintptr_t)d_init | (uintptr_t)d_out) & 15) != 0) CTX_FAIL(prog->ctx, DUSP_ERR_ARG, "dusp_render_chain_window: d_init and d_out must be 16-byte aligned");
    prog->chain_on = true;
    prog->chain_first = first_sample;
    prog->chain_init = d_init;
    prog->chain_raw = raw != 0;
    const int rc = render_device(prog, 1, n_samples, nullptr, nullptr, d_out, stream_);
    prog->chain_on = false;
    prog->chain_init = nullptr;
    return rc;
}

int dusp_render_device_inputs(dusp_program *prog, size_t n_instances, size_t n_samples, const float *d_params, const float *d_inputs,
                              float *d_out, void *stream_) {
    if (!prog) return DUSP_ERR_ARG;
    if (prog->P.g.n_inputs > 0 && !d_inputs) CTX_FAIL(prog->ctx, DUSP_ERR_ARG, "render: the program has input streams but d_inputs is NULL");
    return render_device(prog, n_instances, n_samples, d_params, d_inputs, d_out, stream_);
}

constexpr int kJitLater = 1;  // render_jit: the kernel is being compiled in the background; render this one on the interpreter

// WAVE programs the circuit compiler takes: ONE kernel generated for this circuit's structure (jit_codegen.hpp), compiled for
// gfx950 in process the first time the structure is seen (jit_engine.hip), cached from then on.
// Rings start as zeros (Delay.js:14, CircleBuffer.js:12).  A render nothing continues fills only the slots it can reach (ring_windows.hpp);
// a program that will be continued, one whose channel counts still grow, and DUSP_RING_WINDOW=0 fill the whole rings as before.
// instance_major: rings laid out [instance][slot] (wave engine, compiled kernels), else [slot][n_pad] (chunk engine).
static hipError_t zero_rings(dusp_program *prog, uint32_t n_pad, uint32_t n_chunks, bool instance_major, hipStream_t stream) {
    const dusp::Program &P = prog->P;
    float *rings = prog->d_rings.p;
    const size_t total = (size_t)P.ring_samples;
    if (prog->ctx->knobs.ring_poison) {  // (tests: what the fill leaves out holds NaN patterns, so a window cut too short shows)
        if (hipError_t e = hipMemsetAsync(rings, 0xff, total * n_pad * sizeof(float), stream)) return e;
    }
    const bool windowed = !prog->resumable && prog->ctx->knobs.ring_window != 0 && P.warm_ops.empty();
    std::vector<dusp::RingWindow> wins;
    size_t covered = 0;
    if (windowed) dusp::ring_windows(P, n_chunks, wins, covered);
    if (!windowed || wins.empty() || covered * 2 > total || wins.size() > 64)
        return hipMemsetAsync(rings, 0, total * n_pad * sizeof(float), stream);
    for (const dusp::RingWindow &w : wins) {
        hipError_t e = instance_major ? hipMemset2DAsync(rings + w.at, total * sizeof(float), 0, (size_t)w.count * sizeof(float), n_pad, stream)
                                      : hipMemsetAsync(rings + (size_t)w.at * n_pad, 0, (size_t)w.count * n_pad * sizeof(float), stream);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// handoff_chunks > 0: this launch continues a render whose first handoff_chunks chunks the chunk engine has just rendered (Program::warm_ops):
// start state in d_handoff_init, rings and outlets' last chunk already in this kernel's layout.
// probe: only find out whether the kernel is at hand (DUSP_OK) or being compiled in the background (kJitLater); nothing is launched.
static int render_jit(dusp_program *prog, uint32_t n_inst, size_t n_samples, uint32_t n_chunks, const float *d_params, const float *d_inputs, float *d_out,
                      hipStream_t stream, uint32_t handoff_chunks = 0, bool probe = false) {
    dusp_ctx *ctx = prog->ctx;
    const dusp::Program &P = prog->P;
    const uint32_t n_pad = (n_inst + 63u) & ~63u;
    const size_t n_slots = P.init_state.size();
    const bool persistent = (prog->resumable && prog->persistent) || handoff_chunks > 0;
    const bool resume = prog->keep_memory || handoff_chunks > 0;
    if (resume && !handoff_chunks && n_inst != prog->last_n_inst) CTX_FAIL(ctx, DUSP_ERR_STATE, "render: the instance count cannot change while a program is being continued");
    // Per-instance (parameter) delays: the kernel a Delay gets depends on where its instances' values lie — all of at least a chunk
    // (write-once ring), all below a chunk (no ring), or neither (ordered slot operations) — so the column is looked at first
    // (one small launch + a few bytes back; only programs with such a unit pay it).  The verdict lives in the operand's spare word.
    // Per-instance CUTOFFS of Filters likewise: whether the Filter may run as a scan (jit_filter_scan_ok) depends on the range of the
    // column — its smallest and largest value travel back with the Delays' verdicts, behind the same synchronisation.
    {
        std::vector<int64_t> entries;
        std::vector<size_t> which, filters;
        std::vector<int> slots;
        for (size_t k = 0; k < prog->P.ops.size(); k++) {
            const dusp::DevOp &op = prog->P.ops[k];
            if ((op.op == dusp::OP_DELAY || op.op == dusp::OP_MONO_DELAY) && op.in[1].kind == dusp::SRC_PARAM) {
                entries.insert(entries.end(), {(int64_t)op.in[1].idx, op.ring_len, (int64_t)(op.op == dusp::OP_MONO_DELAY)});
                which.push_back(k);
            }
            if (op.op == dusp::OP_FILTER && op.in[1].kind == dusp::SRC_PARAM && ctx->knobs.filter_scan != 0 && !persistent) {
                slots.push_back(op.in[1].idx);
                filters.push_back(k);
            }
        }
        if (!which.empty() || !filters.empty()) {
            const size_t n = which.size(), nf = filters.size();
            HIP_TRY(ctx, prog->d_jit_regime.ensure(4 * n + 2 * nf + 2));  // [3 n] entries as int64, n verdicts (one int64 slot each), then nf slots (int) and 3 nf range words (unsigned)
            int64_t *d_entries = prog->d_jit_regime.p;
            int *d_bits = (int *)(prog->d_jit_regime.p + 3 * n);
            int *d_slots = (int *)(prog->d_jit_regime.p + 4 * n);
            unsigned *d_range = (unsigned *)(d_slots + nf);
            std::vector<int> bits(n, 0);
            std::vector<unsigned> range(3 * nf, 0u);
            if (n) {
                HIP_TRY(ctx, hipMemcpyAsync(d_entries, entries.data(), 3 * n * sizeof(int64_t), hipMemcpyHostToDevice, stream));
                HIP_TRY(ctx, hipMemsetAsync(d_bits, 0, n * sizeof(int64_t), stream));
                HIP_TRY(ctx, dusp::jit_launch_classify_delays(d_params, n_inst, d_entries, (int)n, d_bits, stream));
                HIP_TRY(ctx, hipMemcpyAsync(bits.data(), d_bits, n * sizeof(int), hipMemcpyDeviceToHost, stream));
            }
            if (nf) {
                HIP_TRY(ctx, hipMemcpyAsync(d_slots, slots.data(), nf * sizeof(int), hipMemcpyHostToDevice, stream));
                HIP_TRY(ctx, hipMemsetAsync(d_range, 0, 3 * nf * sizeof(unsigned), stream));
                HIP_TRY(ctx, dusp::jit_launch_column_range(d_params, n_inst, d_slots, (int)nf, d_range, stream));
                HIP_TRY(ctx, hipMemcpyAsync(range.data(), d_range, 3 * nf * sizeof(unsigned), hipMemcpyDeviceToHost, stream));
            }
            HIP_TRY(ctx, hipStreamSynchronize(stream));
            bool changed = false;
            for (size_t i = 0; i < n; i++) {
                const int regime = bits[i] == dusp::DELAY_REGIME_LONG || bits[i] == dusp::DELAY_REGIME_SHORT ? bits[i] : dusp::DELAY_REGIME_OTHER;
                int32_t &pad = prog->P.ops[which[i]].in[1].pad;
                changed = changed || pad != regime;
                pad = regime;
            }
            for (size_t i = 0; i < nf; i++) {  // (the range itself is no part of the text: only whether the circuit's Filters scan — the key of jit_src — is)
                dusp::DevOp &op = prog->P.ops[filters[i]];
                const bool known = range[3 * i + 2] == 0u && range[3 * i] != 0u && range[3 * i + 1] != 0u;
                op.in[1].pad = known ? dusp::kFilterColumnKnown : 0;
                if (known) {
                    float lo, hi;
                    const unsigned lo_bits = 0x7fffffffu - range[3 * i], hi_bits = range[3 * i + 1];
                    std::memcpy(&lo, &lo_bits, 4);
                    std::memcpy(&hi, &hi_bits, 4);
                    op.d[0] = (double)lo;
                    op.d[1] = (double)hi;
                }
            }
            if (changed) {  // (another kernel text: the generated units differ)
                prog->jit_src.clear();
                prog->jit_consts_uploaded = false;
            }
        }
    }
    if (prog->jit_table_generation != ctx->table_generation) {  // a table was uploaded since: forms / the LDS image may have changed
        prog->jit_src.clear();
        prog->jit_consts_uploaded = false;
        prog->jit_table_generation = ctx->table_generation;
    }

    dusp::JitArgs a{};
    a.params = d_params;
    a.tables = ctx->d_tables;
    a.inputs = d_inputs;
    a.out = d_out;
    a.state = prog->d_state.p;
    a.init_state = handoff_chunks ? prog->d_handoff_init.p : prog->d_init.p;
    a.n_samples = n_samples;
    a.ring_samples = (uint64_t)P.ring_samples;
    a.clock0 = (uint64_t)P.g.clock0 + (uint64_t)handoff_chunks * dusp::kChunk;
    a.n_inst = n_inst;
    a.n_pad = n_pad;
    a.n_groups = n_chunks;
    a.sample_rate = (uint32_t)P.g.sample_rate;
    a.table_stride = ctx->table_stride;
    a.vec4_ok = (n_samples % 4 == 0) && (((uintptr_t)d_out & 15) == 0);
    a.n_out = (uint32_t)P.out_bufs.size();
    // Few instances, long render: cut time into segments so that the whole chip works on it
    a.n_seg = 1;
    a.seg_groups = n_chunks;
    if (prog->voice_loop < 0) {
        dusp::VoicePlan voices;
        prog->voice_loop = !persistent && P.ops.size() > dusp::jit_loop_voices_from() && dusp::jit_find_voices(P, prog->wave, voices) ? 1 : 0;
    }
    if (prog->wave.splittable && !ctx->knobs.jit_force_waves) {
        const uint64_t target = (uint64_t)ctx->n_cus * 8;  // wavefronts that fill the chip
        uint64_t n_seg = n_inst >= target ? 1 : std::min<uint64_t>(target / n_inst, n_chunks / 8);
        if (ctx->knobs.wave_segments >= 0) n_seg = (uint64_t)ctx->knobs.wave_segments;
        n_seg = std::max<uint64_t>(1, std::min<uint64_t>(n_seg, n_chunks));
        if (n_seg > 1) {
            a.seg_groups = (uint32_t)((n_chunks + n_seg - 1) / n_seg);
            a.n_seg = (uint32_t)((n_chunks + a.seg_groups - 1) / a.seg_groups);  // no empty segments
        }
    }
    // A few circuits with Filters, long: segments that warm up (jit_codegen.hpp jit_warm_chunks) — every segment starts a segment early, from rest,
    // its Filters merge with the sequential trajectory on the way (checked below, after the launch), and only its own chunks are stored
    // (a Filter stage's serving wave runs 32 recurrences side by side at the price of one: the chip is full at 32 rows a CU, so up to a quarter of that many
    // instances are still worth cutting)
    if ((uint64_t)n_inst * 4 <= (uint64_t)ctx->n_cus * 32 && !persistent && !resume && !handoff_chunks && !d_inputs && !ctx->knobs.jit_force_waves && ctx->knobs.filter_warm != 0 && ctx->knobs.wave_segments != 0 &&
        ctx->knobs.wave_segments != 1) {
        const uint32_t warm_chunks = dusp::jit_warm_chunks(P, prog->wave);
        if (warm_chunks) {
            // (DUSP_FILTER_WARM=n > 1, tests: segments of n chunks whatever the Filters need — too short a warm-up shows in the check, and the render is finished sequentially)
            const uint64_t target = (uint64_t)ctx->n_cus * 32, per = ctx->knobs.filter_warm > 1 ? (uint64_t)ctx->knobs.filter_warm : std::max<uint64_t>(8, warm_chunks);
            uint64_t n_seg = std::min<uint64_t>(target / n_inst, n_chunks / per);
            if (ctx->knobs.wave_segments > 1) n_seg = std::min<uint64_t>((uint64_t)ctx->knobs.wave_segments, n_chunks / per);
            if (n_seg >= (ctx->knobs.filter_warm > 1 ? 2u : 4u)) {  // (below that the warm-up costs what the split gains)
                a.seg_groups = (uint32_t)((n_chunks + n_seg - 1) / n_seg);
                a.n_seg = (uint32_t)((n_chunks + a.seg_groups - 1) / a.seg_groups);
                a.warm = 1u;
            }
        }
    }
    // Voices in a loop (jit_codegen.hpp VoicePlan): where the circuit is a sum of isomorphic voices above jit_loop_voices_from() units
    const bool voice_loop = prog->voice_loop != 0;
    // workgroup geometry: as many wavefronts as LDS holds next to the table image, no more than gives every CU a workgroup
    dusp::JitOptions opt;
    opt.persistent = persistent;
    opt.voice_loop = voice_loop;
    opt.profile = ctx->knobs.jit_profile != 0;
    opt.filter_fma = ctx->knobs.filter_fma != 0;
    opt.nt_stores = ctx->knobs.jit_nt == 1 || (ctx->knobs.jit_nt == 2 && P.ring_samples != 0);
    for (int k = 0; k < dusp::kNumTables; k++) opt.table_form[k] = ctx->table_form[k], opt.table_delta[k] = ctx->knobs.jit_lean ? ctx->table_delta[k] : 0, opt.table_bound[k] = ctx->table_bound[k];
    // the LDS image goes to the first oscillator table that needs one (saw / square / triangle are evaluated, not looked up)
    for (const dusp::DevOp &op : P.ops)
        if ((op.op == dusp::OP_OSC || op.op == dusp::OP_MULTI_OSC) && opt.lds_table < 0 && ctx->knobs.jit_lds_table != 0 && ctx->table_antisym[op.attr] && P.g.sample_rate % 2 == 0 &&
            !(ctx->table_form[op.attr] >= dusp::TABLE_FORM_SAW && ctx->table_form[op.attr] <= dusp::TABLE_FORM_TRIANGLE)) {
            opt.lds_table = ctx->table_form[op.attr] == dusp::TABLE_FORM_8BIT && ctx->table_antisym[0] ? 0 : op.attr;  // (the sine image serves 8bit too)
            opt.table_bytes = dusp::half_table_lds_bytes((uint32_t)P.g.sample_rate);
        }
    opt.scratch_floats = dusp::jit_scratch_floats(P);
    // Filters whose cutoff is a constant of the circuit, high enough for the bound of jit_filter_scan_ok: a scan over the chunk, the circuit
    // an ordinary one (no Filter stage).  Not for programs that are continued (the stage's y1 / y2 are what the other engines hand over).
    opt.filter_scan = ctx->knobs.filter_scan != 0 && !persistent && !a.warm && dusp::jit_filter_scan_ok(P, ctx->table_bound, ctx->knobs.filter_scan == 2 ? 2 : 1);
    if (!ctx->knobs.jit_rotate) opt.rotate = false;
    if (a.warm) opt.warm = true, opt.rotate = false;  // (what a stage holds at the top of a chunk must be the chunk before's: nothing of the next one computed ahead)
    opt.filter_stages = opt.filter_scan ? 0 : dusp::jit_filter_stages(P);
    opt.filter_mod = !opt.filter_scan && dusp::jit_filter_mod(P);
    // Constant delays of a chunk at least as lines of input samples in LDS (JitDelayLine) instead of rings in memory: where the circuit
    // has no Filter stage (whose overlap splits a Delay's tick), the render is not continued, and all the lines fit at 16 wavefronts next
    // to the table image and the shared scratch — one instance per wavefront.
    if (ctx->knobs.delay_line != 0 && !persistent && opt.filter_stages == 0 && a.n_seg == 1 && !opt.voice_loop && !(ctx->knobs.jit_force_waves && ctx->knobs.jit_force_per_wave > 1)) {
        opt.line_whole_only = ctx->knobs.delay_line == 2;
        const size_t lines = dusp::jit_delay_lines(P, opt.line_whole_only);
        if (lines && opt.table_bytes + 16 * (opt.scratch_floats + lines) * 4 <= 160 * 1024) {
            opt.line_floats = lines;
            opt.scratch_floats += lines;
        }
    }
    const uint64_t n_virtual = (uint64_t)n_inst * a.n_seg;
    const unsigned want = (unsigned)((n_virtual + 255) / 256);
    int most = 16;
    if (ctx->knobs.wave_max_waves > 0) most = std::max(1, std::min(most, ctx->knobs.wave_max_waves));
    // the sequential-stage units' per-wave scratch comes out of the same 160 KiB: as many wavefronts as fit next to the table image
    // (a power of two; the table image goes only when not even one wave's scratch fits beside it)
    if (opt.scratch_floats) {
        const size_t scratch = opt.scratch_floats * 4;
        if (opt.table_bytes + scratch > 160 * 1024) opt.lds_table = -1, opt.table_bytes = 0;
        while (most > 1 && opt.table_bytes + (size_t)most * scratch > 160 * 1024) most /= 2;
    }
    const size_t budget = 160 * 1024 - (size_t)most * opt.scratch_floats * 4;
    int waves = 1, per_wave = 1;
    // (the lines are per wavefront; a wavefront walks ONE segment's chunks: the chunk loop has one counter)
    const int per_wave_cap = opt.line_floats || a.n_seg > 1 ? 1 : ctx->knobs.wave_per_wave >= 1 ? std::min(4, ctx->knobs.wave_per_wave) : 4;
    const bool filter_stage = opt.filter_stages > 0;
    if (filter_stage) {
        // The Filter stage runs one recurrence per lane of ONE wave: a workgroup wants as many instances (rows) as that wave has
        // lanes, and every CU the same number of rounds — rows = instances per CU / rounds, spread over up to 16 wavefronts.
        const uint64_t per_cu = (n_virtual + (uint64_t)ctx->n_cus - 1) / (uint64_t)ctx->n_cus;  // (instances, or the segments of one)
        const uint64_t rounds = (per_cu + 63) / 64;
        const int rows = (int)std::max<uint64_t>(1, (per_cu + rounds - 1) / rounds);
        waves = std::min(most, rows);
        per_wave = std::min(per_wave_cap, (rows + waves - 1) / waves);
        // (16 wavefronts of FOUR instances — 128 registers a lane — spill 170-470 bytes in every Filter circuit measured, filter(osc) included, and end at 16 x 2 two
        // compiles later: start there.  A first render of such a structure: 2.1-3.3 s -> one compile)
        if (waves == 16 && per_wave == 4 && ctx->knobs.wave_per_wave < 4) per_wave = 2;
        opt.filter_sub = dusp::jit_filter_sub(waves, per_wave, opt.filter_stages, budget - opt.table_bytes, opt.filter_mod);
        // (a connected cutoff parks three values per sample in two sets of rows: fewer rows per workgroup before the table image goes)
        while (opt.filter_mod && !opt.filter_sub && (per_wave > 1 || waves > 1)) {
            if (per_wave > 1) per_wave /= 2;
            else waves /= 2;
            opt.filter_sub = dusp::jit_filter_sub(waves, per_wave, opt.filter_stages, budget - opt.table_bytes, opt.filter_mod);
        }
        if (!opt.filter_sub) {  // (cannot happen with a 99 KB image: 64 rows of 64 samples take 33 KB)
            opt.lds_table = -1, opt.table_bytes = 0;
            opt.filter_sub = dusp::jit_filter_sub(waves, per_wave, opt.filter_stages, budget, opt.filter_mod);
        }
    } else {
        while (waves < most && (unsigned)waves < want) waves *= 2;
        // instances per wavefront (unsplit renders of light circuits, jit_light): 4 or 2 while that leaves every CU a workgroup —
        // their independent unit blocks fill each other's latencies
        // (not next to the table image: since the oscillators' delta form — 17 instructions a sample instead of 26 — one instance per wave is
        // the faster: osc(k) 0.66 / 0.70 / 0.69 ms at 1 / 2 / 4, mul(osc, k) 0.68 / 0.66 / 0.70; ramp and timer graphs 0.79 / 0.66 / 0.63)
        if (a.n_seg == 1 && ((dusp::jit_light(P) && opt.lds_table < 0) || ctx->knobs.wave_per_wave > 1))
            for (int r : {4, 2})
                if (r <= per_wave_cap && (uint64_t)ctx->n_cus * waves * r <= n_inst) {
                    per_wave = r;
                    break;
                }
    }

    if (ctx->knobs.jit_force_waves) {  // tests: this geometry, whatever the batch
        waves = std::min(most, ctx->knobs.jit_force_waves);
        per_wave = ctx->knobs.jit_force_per_wave;
        if (filter_stage) {
            opt.filter_sub = dusp::jit_filter_sub(waves, per_wave, opt.filter_stages, budget - opt.table_bytes, opt.filter_mod);
            if (!opt.filter_sub) CTX_FAIL(ctx, DUSP_ERR_ARG, "render: DUSP_JIT_FORCE: the Filter stage's rows do not fit LDS at this geometry");
        }
    }

    hipFunction_t render = nullptr;
    dusp::JitSource *src = nullptr;
    int jit_scratch = 0;
    for (;;) {  // a kernel that spills (128 registers per lane at 16 wavefronts) is rebuilt for fewer instances per wave, then fewer waves
        opt.waves = waves;
        opt.per_wave = per_wave;
        auto it = prog->jit_src.find({waves, per_wave * 8 + opt.filter_block % 8 + (opt.voice_loop ? 64 : 0) + (opt.filter_scan ? 128 : 0) + (opt.rotate ? 0 : 256)});
        if (it == prog->jit_src.end()) {
            dusp::JitSource gen;
            if (!dusp::jit_generate(P, prog->wave, opt, gen)) CTX_FAIL(ctx, DUSP_ERR_UNSUPPORTED, "render: circuit compiler: " + gen.why);
            it = prog->jit_src.emplace(std::make_pair(waves, per_wave * 8 + opt.filter_block % 8 + (opt.voice_loop ? 64 : 0) + (opt.filter_scan ? 128 : 0) + (opt.rotate ? 0 : 256)), std::move(gen)).first;
        }
        src = &it->second;
        // A structure seen for the first time costs a compile of 0.3-0.8 s.  A render the interpreter kernel finishes sooner
        // than that does not wait for it: the compile starts in a background thread, THIS render runs on the interpreter (same
        // PCM, same state), and the next render of the structure — in this process, or in any with DUSP_JIT_CACHE set — finds
        // its kernel.  DUSP_WAVE_JIT=2 always waits (tests, benchmarks).
        if (ctx->knobs.wave_jit == 1 && (!handoff_chunks || probe) && !dusp::jit_code_ready(src->text)) {
            // interpreter: ~0.7 ns per unit and chunk with the chip full, ~1 us per unit and chunk along one wavefront's serial path
            const double units = (double)P.ops.size();
            const double est_ms = std::max(units * (double)n_inst * n_chunks * 0.7e-6, units * (double)a.seg_groups * 1.0e-3);
            if (est_ms < 400.0) {
                dusp::jit_compile_in_background(src->text);
                if (filter_stage && opt.filter_block == 8) {
                    // (a Filter stage's kernel at 16 wavefronts usually ends on the recurrence loop's narrower form: that text joins the
                    // queue now, so the geometry search does not cost a further render on the interpreter per step)
                    dusp::JitOptions narrow = opt;
                    narrow.filter_block = 4;
                    const auto key = std::make_pair(waves, per_wave * 8 + narrow.filter_block % 8 + (narrow.voice_loop ? 64 : 0) + (narrow.filter_scan ? 128 : 0) + (narrow.rotate ? 0 : 256));
                    auto alt = prog->jit_src.find(key);
                    if (alt == prog->jit_src.end()) {
                        dusp::JitSource gen;
                        if (dusp::jit_generate(P, prog->wave, narrow, gen)) alt = prog->jit_src.emplace(key, std::move(gen)).first;
                    }
                    if (alt != prog->jit_src.end() && !dusp::jit_code_ready(alt->second.text)) dusp::jit_compile_in_background(alt->second.text);
                }
                return kJitLater;
            }
        }
        std::string err;
        int scratch = 0;
        if (!dusp::jit_get_kernel(ctx->device, src->text, "dusp_jit_render", &render, &scratch, err))
            CTX_FAIL(ctx, DUSP_ERR_HIP, "render: circuit compiler: " + err);
        jit_scratch = scratch;
        if (ctx->knobs.jit_log) fprintf(stderr, "[dusp jit] %d waves x %d instances, filter block %d: %d bytes of scratch per lane\n", waves, per_wave, opt.filter_block, scratch);
        if (scratch <= ctx->knobs.jit_spill_bytes || ctx->knobs.jit_force_waves) break;  // (a few registers spilled outside the hot path is cheaper than halving the instances in flight)
        // The kernel spills at this geometry (16 wavefronts: 128 registers per lane).  A Filter circuit keeps its rows if it can —
        // half the wavefronts with twice the instances each have twice the registers — else instances per wave, then waves, go down.
        if (filter_stage && opt.filter_block == 8) {
            opt.filter_block = 4;  // (first: the recurrence loop with half the P values in flight, 16 registers less)
            continue;
        }
        opt.filter_block = 8;
        if (filter_stage && waves > 4 && waves % 2 == 0 && per_wave * 2 <= per_wave_cap) {
            waves /= 2;
            per_wave *= 2;
        } else if (filter_stage && per_wave > 1) per_wave /= 2;  // (rows stay a power of two: whole rounds on every CU)
        else if (per_wave > 1) per_wave /= 2;  // (4, 2, 1: an odd count leaves the last round of workgroups a third full at the usual batch sizes)
        else if (waves > 4) waves /= 2;
        else break;
        if (filter_stage) opt.filter_sub = dusp::jit_filter_sub(waves, per_wave, opt.filter_stages, budget - opt.table_bytes, opt.filter_mod);
    }
    if (probe) return DUSP_OK;
    // (from here on the render happens on the compiled kernel: workspaces)
    HIP_TRY(ctx, prog->d_state.ensure(std::max<size_t>(1, n_slots) * n_pad));
    a.state = prog->d_state.p;
    if (P.ring_samples && !resume) {  // Delay rings start as zeros (Delay.js:14); layout [instance][slot]
        HIP_TRY(ctx, prog->d_rings.ensure((size_t)P.ring_samples * n_pad));
        HIP_TRY(ctx, zero_rings(prog, n_pad, n_chunks, true, stream));
    }
    a.rings = prog->d_rings.p;
    a.resume = resume ? 1u : 0u;
    a.save_bufs = persistent ? 1u : 0u;
    a.n_bufs = (uint32_t)std::max(1, P.n_bufs);
    if (persistent) {  // every outlet's last chunk, parked between launches (the interpreter kernel's layout: either may continue the other)
        HIP_TRY(ctx, prog->d_saved_bufs.ensure((size_t)a.n_bufs * dusp::kChunk * n_inst));
        a.saved_bufs = prog->d_saved_bufs.p;
    }
    prog->keep_memory = false;
    if (!prog->jit_consts_uploaded) {
        HIP_TRY(ctx, prog->d_jit_fk.ensure(std::max<size_t>(1, src->fk.size())));
        HIP_TRY(ctx, prog->d_jit_dk.ensure(std::max<size_t>(1, src->dk.size())));
        if (!src->fk.empty()) HIP_TRY(ctx, hipMemcpyAsync(prog->d_jit_fk.p, src->fk.data(), src->fk.size() * sizeof(float), hipMemcpyHostToDevice, stream));
        if (!src->dk.empty()) HIP_TRY(ctx, hipMemcpyAsync(prog->d_jit_dk.p, src->dk.data(), src->dk.size() * sizeof(double), hipMemcpyHostToDevice, stream));
        std::vector<int> scan(2 * src->scans.size() + 2, 0);
        for (size_t i = 0; i < src->scans.size(); i++) {
            scan[i] = src->scans[i].state_slot;
            scan[src->scans.size() + i] = src->scans[i].level;
        }
        HIP_TRY(ctx, prog->d_jit_scan.ensure(scan.size()));
        HIP_TRY(ctx, hipMemcpyAsync(prog->d_jit_scan.p, scan.data(), scan.size() * sizeof(int), hipMemcpyHostToDevice, stream));
        HIP_TRY(ctx, hipStreamSynchronize(stream));  // (the host vectors above are temporaries / may be regenerated)
        prog->jit_consts_uploaded = true;
    }
    a.fk = prog->d_jit_fk.p;
    a.dk = prog->d_jit_dk.p;
    const unsigned per_block = (unsigned)(waves * per_wave);
    const unsigned grid = (unsigned)((n_virtual + per_block - 1) / per_block);
    if (!handoff_chunks) HIP_TRY(ctx, hipEventRecord(prog->ev0, stream));  // (a hand-off's clock started in front of the chunk engine's part)
    if (a.n_seg > 1 && !src->scans.empty()) {  // one accumulate pass + prefix per FM level that has scanned oscillators, then the render pass
        const size_t per = src->scans.size() * (size_t)n_inst * a.n_seg;
        HIP_TRY(ctx, prog->d_seg.ensure(2 * per));
        a.seg_sum = prog->d_seg.p;
        a.seg_start = prog->d_seg.p + per;
        for (int level : src->pass_levels) {
            hipFunction_t pass = nullptr;
            std::string err;
            if (!dusp::jit_get_kernel(ctx->device, src->text, "dusp_jit_pass" + std::to_string(level), &pass, nullptr, err))
                CTX_FAIL(ctx, DUSP_ERR_HIP, "render: circuit compiler: " + err);
            dusp::JitArgs own = a;
            own.warm = 0u;  // (a pass totals every segment's OWN chunks, whatever the render kernel does in front of them)
            HIP_TRY(ctx, dusp::jit_launch(pass, own, grid, (unsigned)waves * 64, stream));
            HIP_TRY(ctx, dusp::jit_launch_prefix(a.seg_sum, a.seg_start, prog->d_init.p, prog->d_jit_scan.p, prog->d_jit_scan.p + src->scans.size(),
                                                 (int)src->scans.size(), level, n_inst, a.n_seg, a.sample_rate, stream));
        }
    }
    DevBuf<unsigned long long> d_debug;
    if (opt.profile) {
        HIP_TRY(ctx, d_debug.ensure((size_t)grid * 16));
        HIP_TRY(ctx, hipMemsetAsync(d_debug.p, 0, (size_t)grid * 16 * sizeof(unsigned long long), stream));
        a.debug = d_debug.p;
    }
    if (a.warm) {
        HIP_TRY(ctx, prog->d_warm_records.ensure((size_t)std::max(1, opt.filter_stages) * n_virtual * 8));
        a.warm_records = prog->d_warm_records.p;
    }
    HIP_TRY(ctx, dusp::jit_launch(render, a, grid, (unsigned)waves * 64, stream));
    if (a.warm) {
        // Segments that warmed up: does every Filter stage hold, where a segment's own chunks begin, what the segment before ended with?  Then —
        // by induction from the first segment, which started from the render's true state — every stored sample is the sequential render's.
        // Otherwise the render is finished sequentially from the last segment that is known to be right: one wavefront from that
        // segment's first chunk on, its Filters started from the state recorded there (x1 x2 y1 y2 of every stage into a copy of the start state).
        const size_t n_rec = (size_t)opt.filter_stages * n_virtual * 8;
        std::vector<double> rec(n_rec);
        HIP_TRY(ctx, hipMemcpyAsync(rec.data(), prog->d_warm_records.p, n_rec * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_TRY(ctx, hipStreamSynchronize(stream));
        uint32_t bad = 0;  // first segment (of any instance) whose start differs from its predecessor's end (0: none)
        for (uint32_t i = 0; i < n_inst; i++)
            for (uint32_t s = 1; s < a.n_seg && (!bad || s < bad); s++)
                for (int st = 0; st < opt.filter_stages; st++) {
                    const double *now = rec.data() + ((size_t)st * n_virtual + (size_t)i * a.n_seg + s) * 8, *before = now - 8;
                    if (!(now[0] == before[2] && now[1] == before[3])) bad = s;  // (a NaN never equals: such a render is finished as written)
                }
        prog->warm_redo_from = bad;
        if (bad && (n_inst > 1 || !src->scans.empty())) {
            // several instances (each with a state of its own by then), or scanned oscillators (their phases are the passes' business): the whole
            // render once more, every instance as one chain from its first chunk
            dusp::JitArgs whole = a;
            whole.warm = 0u;
            whole.n_seg = 1u;
            whole.seg_groups = n_chunks;
            const unsigned blocks = (unsigned)((n_inst + per_block - 1) / per_block);
            HIP_TRY(ctx, dusp::jit_launch(render, whole, blocks, (unsigned)waves * 64, stream));
            prog->warm_redo_from = 1;
        } else if (bad) {
            std::vector<double> init(P.init_state);
            int st = 0;
            for (int k : prog->wave.order) {  // (stage ordinals are dealt in the plan's execution order: jit_codegen.hpp filter_ordinal)
                const dusp::DevOp &op = P.ops[(size_t)k];
                if (op.op == dusp::OP_FILTER) {
                    const double *before = rec.data() + ((size_t)st * n_virtual + (bad - 1)) * 8;
                    const size_t at = (size_t)op.state_slot;
                    if (at + 11 <= init.size()) init[at + 7] = before[4], init[at + 8] = before[5], init[at + 9] = before[2], init[at + 10] = before[3];
                    st++;
                }
            }
            HIP_TRY(ctx, prog->d_warm_init.ensure(init.size()));
            HIP_TRY(ctx, hipMemcpyAsync(prog->d_warm_init.p, init.data(), init.size() * sizeof(double), hipMemcpyHostToDevice, stream));
            dusp::JitArgs rest = a;
            rest.warm = 0u;
            rest.n_seg = 1u;
            rest.g_first = bad * a.seg_groups;
            rest.seg_groups = n_chunks;
            rest.init_state = prog->d_warm_init.p;
            HIP_TRY(ctx, dusp::jit_launch(render, rest, 1u, (unsigned)waves * 64, stream));
            HIP_TRY(ctx, hipStreamSynchronize(stream));  // (`init` is a temporary)
        }
    }
    HIP_TRY(ctx, hipEventRecord(prog->ev1, stream));
    if (opt.profile) {  // diagnostic build: what wave 0 of the workgroups measured (mean over workgroups), to stderr
        std::vector<unsigned long long> h((size_t)grid * 16);
        HIP_TRY(ctx, hipStreamSynchronize(stream));
        HIP_TRY(ctx, hipMemcpy(h.data(), d_debug.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double loop = 0, serial = 0, chunks = 0;
        double ph[12] = {0};
        for (unsigned b = 0; b < grid; b++) {
            loop += (double)h[b * 16], serial += (double)h[b * 16 + 1] + (double)h[b * 16 + 3], chunks += (double)h[b * 16 + 2];
            for (int i = 0; i < 12; i++) ph[i] += (double)h[b * 16 + 4 + i];
        }
        std::fprintf(stderr, "[dusp jit profile] %ux%d waves x instances (%d B of scratch per lane), %u workgroups: chunk loop %.0f cycles per chunk, of which Filter recurrences %.0f (%.1f per sample-step)\n",
                     (unsigned)waves, per_wave, jit_scratch, grid, loop / std::max(1.0, chunks), serial / std::max(1.0, chunks), serial / std::max(1.0, chunks) / 256.0);
        std::fprintf(stderr, "[dusp jit profile]   barrier to barrier, as wave 0 sees them:");
        for (int i = 0; i < 12; i++)
            if (ph[i] > 0) std::fprintf(stderr, " %s%.0f", i == 11 ? "| tail " : "", ph[i] / std::max(1.0, chunks));
        std::fprintf(stderr, "\n");
        d_debug.release();
    }
    prog->jit_waves = waves;
    prog->jit_per_wave = per_wave;
    prog->jit_segments = a.n_seg;
    prog->jit_voices = src->voice_loop;
    prog->jit_scan = opt.filter_scan;
    if (!a.warm) prog->warm_redo_from = 0;
    prog->last_n_inst = n_inst;
    prog->last_n_pad = n_pad;
    prog->rendered = true;
    prog->h_state_valid = false;
    prog->next_clock = P.g.clock0 + (int64_t)(n_chunks + handoff_chunks) * dusp::kChunk;
    return DUSP_OK;
}

static int render_device_unguarded(dusp_program *prog, size_t n_instances, size_t n_samples, const float *d_params, const float *d_inputs,
                                   float *d_out, void *stream_);

// DUSP_GUARD=1: wait for the render and look at the guard bytes behind every workspace it could have touched
static int check_guards(dusp_program *prog, hipStream_t stream) {
    dusp_ctx *ctx = prog->ctx;
    if (!g_guard_bytes) return DUSP_OK;
    HIP_TRY(ctx, hipStreamSynchronize(stream));
    const char *hit = nullptr;
    if (!prog->d_scratch.intact()) hit = "chunk buffers";
    else if (!prog->d_rings.intact()) hit = "rings";
    else if (!prog->d_state.intact()) hit = "unit state";
    else if (!prog->d_seg.intact()) hit = "segment phases";
    else if (!prog->d_fused_state.intact()) hit = "fused end state";
    else if (!prog->d_recs.intact()) hit = "voice records";
    else if (!prog->d_sum_voices.intact()) hit = "sum-chain records";
    else if (!prog->d_saved_bufs.intact()) hit = "parked chunks";
    else if (!prog->d_rings_wave.intact()) hit = "parked rings";
    else if (!prog->d_host_out.intact()) hit = "staging PCM";
    else if (!prog->d_host_frames.intact()) hit = "staging frames";
    else if (!prog->d_host_par.intact()) hit = "staging parameters";
    else if (!prog->d_host_in.intact()) hit = "staging inputs";
    else if (!prog->d_handoff_init.intact() || !prog->d_handoff_out.intact()) hit = "hand-off buffers";
    else if (!prog->d_ops.intact() || !prog->d_out_bufs.intact() || !prog->d_init.intact()) hit = "program constants";
    else if (!prog->d_jit_fk.intact() || !prog->d_jit_dk.intact() || !prog->d_jit_scan.intact()) hit = "compiled kernel's constants";
    else if (ctx->tables_guarded && ctx->d_tables && !guard_intact((const char *)ctx->d_tables + sizeof(float) * dusp::kNumTables * ctx->table_stride)) hit = "lookup tables";
    if (hit) CTX_FAIL(ctx, DUSP_ERR_HIP, std::string("render: a kernel wrote past the end of a device buffer (") + hit + "): guard bytes overwritten");
    return DUSP_OK;
}

static int render_device(dusp_program *prog, size_t n_instances, size_t n_samples, const float *d_params, const float *d_inputs,
                         float *d_out, void *stream_) {
    return guarded(prog->ctx->err, "render", [&]() -> int {
        if (int rc = render_device_unguarded(prog, n_instances, n_samples, d_params, d_inputs, d_out, stream_)) return rc;
        return check_guards(prog, stream_ ? (hipStream_t)stream_ : prog->ctx->stream);
    });
}

static int render_device_unguarded(dusp_program *prog, size_t n_instances, size_t n_samples, const float *d_params, const float *d_inputs,
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
    // A program's workspaces (state, rings, scratch, voice records) are reused from render to render: a render on ANOTHER
    // stream than the previous one first waits for what that one recorded.
    if (prog->rendered && prog->last_stream != stream) HIP_TRY(ctx, hipStreamWaitEvent(stream, prog->ev1, 0));
    prog->last_stream = stream;
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
        L.table_delta = ctx->table_delta[prog->fused.table_id];
        L.table_form = ctx->table_form[prog->fused.table_id];
        L.knobs = ctx->knobs;
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
            if (prog->chain_on) {  // (dusp_render_chain_window: this launch is a window of the timeline and continues another rank's sums)
                L.chain_first = prog->chain_first;
                L.chain_init = prog->chain_init;
                L.chain_raw = prog->chain_raw;
            }
            dusp::build_sum_voices(prog->fused, (uint32_t)P.g.sample_rate, gb, L.chain_first + (uint64_t)n_chunks * dusp::kChunk, prog->h_sum_voices, prog->h_sum_end);
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
    prog->jit_waves = prog->jit_per_wave = 0;
    if (prog->engine == DUSP_ENGINE_WAVE && prog->jit_ok) {
        const int rc = render_jit(prog, n_inst, n_samples, n_chunks, d_params, d_inputs, d_out, stream);
        if (rc != kJitLater) return rc;
    }
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
            HIP_TRY(ctx, zero_rings(prog, n_pad, n_chunks, true, stream));
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
            const uint64_t target = (uint64_t)ctx->n_cus * 8;  // wavefronts that fill the chip
            uint64_t n_seg = n_inst >= target ? 1 : std::min<uint64_t>(target / n_inst, n_chunks / 8);
            if (ctx->knobs.wave_segments >= 0) n_seg = (uint64_t)ctx->knobs.wave_segments;  // 0 / 1: off; n: force n segments
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
        HIP_TRY(ctx, dusp::launch_wave_engine(w, lds_ok, ctx->knobs.wave_max_waves, stream));
        HIP_TRY(ctx, hipEventRecord(prog->ev1, stream));
        prog->last_n_inst = n_inst;
        prog->last_n_pad = n_pad;
        prog->rendered = true;
        prog->h_state_valid = false;
        prog->next_clock = P.g.clock0 + (int64_t)n_chunks * dusp::kChunk;
        return DUSP_OK;
    }
    // Channel counts that grow during the first chunks (Program::warm_ops): a single circuit renders those chunks here, on the chunk engine
    // (with the reference's own ring protocol), and the rest on the kernel compiled for its settled op list — rings, every outlet's last
    // chunk and the unit state move into that kernel's layout in between (dusp_chunk_to_wave_kernel).
    if (prog->engine == DUSP_ENGINE_CHUNK && prog->handoff_ok && n_inst == 1 && !prog->keep_memory && P.g.clock0 % dusp::kChunk == 0) {
        const uint64_t first_chunk = (uint64_t)P.g.clock0 / dusp::kChunk;
        const uint32_t n_warm = (uint32_t)P.warm_ops.size();
        const uint32_t W = first_chunk < n_warm ? (uint32_t)std::min<uint64_t>(n_chunks, n_warm - first_chunk) : 0u;
        // (under the default knob a structure seen for the first time renders on the chunk engine alone while its kernel compiles in the background)
        int ready = kJitLater;
        if (W > 0 && W < n_chunks) {
            ready = render_jit(prog, n_inst, n_samples - (size_t)W * dusp::kChunk, n_chunks - W, d_params, d_inputs, d_out, stream, W, /*probe=*/true);
            if (ready != DUSP_OK && ready != kJitLater) return ready;  // (a kernel that does not compile is a failure, not a reason to render elsewhere)
        }
        if (ready == DUSP_OK) {
            const size_t n_ch = P.out_bufs.size(), n_head = (size_t)W * dusp::kChunk, n_rest = n_samples - n_head;
            const uint32_t n_bufs = (uint32_t)std::max(1, P.n_bufs);
            HIP_TRY(ctx, prog->d_scratch.ensure((size_t)n_bufs * dusp::kChunk * n_pad));
            HIP_TRY(ctx, prog->d_state.ensure(std::max<size_t>(1, n_slots) * n_pad));
            HIP_TRY(ctx, prog->d_rings.ensure(std::max<size_t>(1, (size_t)P.ring_samples) * n_pad));
            HIP_TRY(ctx, prog->d_rings_wave.ensure(std::max<size_t>(1, (size_t)P.ring_samples) * n_pad));
            HIP_TRY(ctx, prog->d_saved_bufs.ensure((size_t)n_bufs * dusp::kChunk * n_inst));
            HIP_TRY(ctx, prog->d_handoff_init.ensure(std::max<size_t>(1, n_slots)));
            HIP_TRY(ctx, prog->d_handoff_out.ensure(n_ch * std::max(n_head, n_rest)));
            HIP_TRY(ctx, hipMemsetAsync(prog->d_scratch.p, 0, (size_t)n_bufs * dusp::kChunk * n_pad * sizeof(float), stream));
            if (P.ring_samples) HIP_TRY(ctx, hipMemsetAsync(prog->d_rings.p, 0, (size_t)P.ring_samples * n_pad * sizeof(float), stream));
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
            a.out = prog->d_handoff_out.p;
            a.n_samples = n_head;
            a.clock0 = P.g.clock0;
            a.n_ops = (uint32_t)P.ops.size();
            a.n_out = (uint32_t)n_ch;
            a.n_inst = n_inst;
            a.n_pad = n_pad;
            a.n_chunks = W;
            a.sample_rate = (uint32_t)P.g.sample_rate;
            a.table_stride = ctx->table_stride;
            a.flags = dusp::kChunkFlagResumable;  // (rings in the reference's own state: a compiled kernel continues them)
            a.n_warm = n_warm;
            for (uint32_t k = 0, at = a.n_ops; k < a.n_warm; k++) {
                a.warm_first[k] = at;
                a.warm_n[k] = (uint32_t)P.warm_ops[k].size();
                at += a.warm_n[k];
            }
            HIP_TRY(ctx, hipEventRecord(prog->ev0, stream));
            HIP_TRY(ctx, dusp::launch_chunk_engine(a, stream));
            HIP_TRY(ctx, hipMemcpy2DAsync(d_out, n_samples * sizeof(float), prog->d_handoff_out.p, n_head * sizeof(float), n_head * sizeof(float), n_ch,
                                          hipMemcpyDeviceToDevice, stream));
            HIP_TRY(ctx, dusp::launch_chunk_to_wave(prog->d_rings.p, prog->d_rings_wave.p, (uint64_t)P.ring_samples, prog->d_scratch.p, prog->d_saved_bufs.p, n_bufs,
                                                    n_inst, n_pad, prog->d_state.p, prog->d_handoff_init.p, (uint32_t)n_slots, stream));
            std::swap(prog->d_rings_wave.p, prog->d_rings.p);  // (same size; the compiled kernel's rings are the program's rings from here on)
            std::swap(prog->d_rings_wave.cap, prog->d_rings.cap);
            prog->last_n_inst = n_inst;
            const int rc = render_jit(prog, n_inst, n_rest, n_chunks - W, d_params, d_inputs, prog->d_handoff_out.p, stream, W);
            if (rc == kJitLater) CTX_FAIL(ctx, DUSP_ERR_STATE, "render: internal error: a hand-off that does not wait for its kernel");
            if (rc != DUSP_OK) return rc;
            HIP_TRY(ctx, hipMemcpy2DAsync(d_out + n_head, n_samples * sizeof(float), prog->d_handoff_out.p, n_rest * sizeof(float), n_rest * sizeof(float), n_ch,
                                          hipMemcpyDeviceToDevice, stream));
            HIP_TRY(ctx, hipEventRecord(prog->ev1, stream));
            return DUSP_OK;
        }
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
        if (P.ring_samples) HIP_TRY(ctx, zero_rings(prog, n_pad, n_chunks, false, stream));
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

// Device -> host delivery of the rendered PCM (what renderChannelData's caller finally holds).
//   * h_out is pinned memory (dusp_host_alloc, or registered by the caller): one asynchronous DMA straight into it.
//   * h_out is pageable and large: kCopyWorkers worker threads, each with its own stream and a pair of pinned staging
//     tiles, walk disjoint ranges of the output — DMA of tile i+1 into one tile while the CPU copies tile i out of the
//     other.  (A plain hipMemcpy to pageable memory stages through ONE pinned buffer with ONE copying thread: 10-13 GB/s.)
//   * small outputs (event-segmented rendering: hundreds of short renders a second): plain asynchronous copy.
constexpr size_t kCopyTileFloats = (size_t)2 << 20;   // 8 MiB staging tiles
constexpr int kCopyWorkers = 4;
constexpr size_t kStagedMinFloats = (size_t)8 << 20;  // 32 MiB: below this the staging pipeline is not worth its threads

static bool is_pinned_host(const void *p) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();  // an unregistered pointer is reported as an error: not one of ours
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

static hipError_t download_staged(dusp_program *prog, float *h_out, const float *d_src, size_t n_floats) {
    dusp_ctx *ctx = prog->ctx;
    const size_t need = (size_t)kCopyWorkers * 2 * kCopyTileFloats;
    if (prog->pin_floats < need) {
        if (prog->pin[0]) (void)hipHostFree(prog->pin[0]);
        prog->pin[0] = nullptr;
        prog->pin_floats = 0;
        hipError_t e = hipHostMalloc((void **)&prog->pin[0], need * sizeof(float), hipHostMallocDefault);
        if (e != hipSuccess) return e;
        prog->pin_floats = need;
    }
    hipError_t e = hipStreamSynchronize(ctx->stream);  // the render (and the interleave) have finished: the workers only copy
    if (e != hipSuccess) return e;
    hipError_t results[kCopyWorkers];
    std::thread workers[kCopyWorkers];
    const size_t n_tiles = (n_floats + kCopyTileFloats - 1) / kCopyTileFloats;
    for (int w = 0; w < kCopyWorkers; w++) {
        results[w] = hipSuccess;
        workers[w] = std::thread([&, w]() {
            hipError_t &r = results[w];
            hipStream_t st = nullptr;
            if ((r = hipSetDevice(ctx->device)) != hipSuccess || (r = hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) != hipSuccess) return;
            float *tile[2] = {prog->pin[0] + (size_t)(2 * w) * kCopyTileFloats, prog->pin[0] + (size_t)(2 * w + 1) * kCopyTileFloats};
            // this worker's tiles: a contiguous range (neighbouring pages of h_out are faulted in by one thread)
            const size_t t0 = n_tiles * (size_t)w / kCopyWorkers, t1 = n_tiles * (size_t)(w + 1) / kCopyWorkers;
            auto span = [&](size_t t, size_t &at, size_t &n) {
                at = t * kCopyTileFloats;
                n = std::min(kCopyTileFloats, n_floats - at);
            };
            hipEvent_t done[2] = {nullptr, nullptr};
            if ((r = hipEventCreateWithFlags(&done[0], hipEventDisableTiming)) == hipSuccess) r = hipEventCreateWithFlags(&done[1], hipEventDisableTiming);
            size_t at, n;
            for (size_t t = t0; r == hipSuccess && t < std::min(t0 + 2, t1); t++) {  // prime both tiles
                span(t, at, n);
                if ((r = hipMemcpyAsync(tile[(t - t0) & 1], d_src + at, n * sizeof(float), hipMemcpyDeviceToHost, st)) == hipSuccess)
                    r = hipEventRecord(done[(t - t0) & 1], st);
            }
            for (size_t t = t0; r == hipSuccess && t < t1; t++) {
                const int k = (int)((t - t0) & 1);
                if ((r = hipEventSynchronize(done[k])) != hipSuccess) break;
                span(t, at, n);
                std::memcpy(h_out + at, tile[k], n * sizeof(float));
                if (t + 2 < t1) {
                    span(t + 2, at, n);
                    if ((r = hipMemcpyAsync(tile[k], d_src + at, n * sizeof(float), hipMemcpyDeviceToHost, st)) == hipSuccess)
                        r = hipEventRecord(done[k], st);
                }
            }
            (void)hipStreamSynchronize(st);
            if (done[0]) (void)hipEventDestroy(done[0]);
            if (done[1]) (void)hipEventDestroy(done[1]);
            (void)hipStreamDestroy(st);
        });
    }
    for (auto &t : workers) t.join();
    for (hipError_t r : results)
        if (r != hipSuccess) return r;
    return hipSuccess;
}

static int render_host(dusp_program *prog, size_t n_instances, size_t n_samples, const float *h_params, const float *h_inputs, float *h_out,
                       bool interleaved) {
    if (!prog) return DUSP_ERR_ARG;
    dusp_ctx *ctx = prog->ctx;
    return guarded(ctx->err, "render", [&]() -> int {
    if (!h_out) CTX_FAIL(ctx, DUSP_ERR_ARG, "render: h_out is NULL");
    if (n_instances < 1 || n_instances > (1u << 24) || n_samples < 1 || n_samples > (1ull << 31))
        CTX_FAIL(ctx, DUSP_ERR_ARG, "render: n_instances must be in [1, 2^24] and n_samples in [1, 2^31]");
    const size_t n_par = (size_t)prog->P.g.n_params * n_instances;
    if (n_par && !h_params) CTX_FAIL(ctx, DUSP_ERR_ARG, "render: program has parameters but h_params is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const auto t_start = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_start).count(); };
    const size_t n_out = n_instances * prog->P.out_bufs.size() * n_samples;
    // staging buffers live with the program (grown on demand): a segmented render calls this hundreds of times a second
    HIP_TRY(ctx, prog->d_host_out.ensure(std::max<size_t>(1, n_out)));
    const double us_alloc = since();
    float *d_out = prog->d_host_out.p, *d_par = nullptr, *d_frames = nullptr;
    if (n_par) {
        HIP_TRY(ctx, prog->d_host_par.ensure(n_par));
        d_par = prog->d_host_par.p;
        HIP_TRY(ctx, hipMemcpyAsync(d_par, h_params, n_par * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    }
    const size_t n_in = (size_t)prog->P.g.n_inputs * n_instances * n_samples;
    float *d_in = nullptr;
    if (n_in && !h_inputs) CTX_FAIL(ctx, DUSP_ERR_ARG, "render: the program reads host-generated input streams; use dusp_render_host_inputs");
    if (n_in) {
        HIP_TRY(ctx, prog->d_host_in.ensure(n_in));
        d_in = prog->d_host_in.p;
        HIP_TRY(ctx, hipMemcpyAsync(d_in, h_inputs, n_in * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    }
    if (int rc = render_device_unguarded(prog, n_instances, n_samples, d_par, d_in, d_out, ctx->stream)) return rc;
    const double us_enqueued = since();
    if (int rc = check_guards(prog, ctx->stream)) return rc;
    const size_t n_ch = prog->P.out_bufs.size();
    if (interleaved && n_ch > 1) {  // frames: transpose on the device, then download those
        HIP_TRY(ctx, prog->d_host_frames.ensure(n_out));
        d_frames = prog->d_host_frames.p;
        if (int rc = dusp_interleave_device(ctx, d_out, n_instances, n_ch, n_samples, d_frames, ctx->stream)) return rc;
    }
    const float *d_src = d_frames ? d_frames : d_out;
    if (n_out >= kStagedMinFloats && !is_pinned_host(h_out)) {
        HIP_TRY(ctx, download_staged(prog, h_out, d_src, n_out));
    } else {  // pinned destination: one DMA at link speed; small output: not worth more
        HIP_TRY(ctx, hipMemcpyAsync(h_out, d_src, n_out * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (ctx->knobs.jit_log >= 2)  // (DUSP_JIT_LOG=2: where a host render's time goes)
        fprintf(stderr, "[dusp host render] output buffer %.0f us, render enqueued (workspaces, constants, launches) %.0f us, download + wait %.0f us\n", us_alloc, us_enqueued - us_alloc,
                since() - us_enqueued);
    return DUSP_OK;
    });
}

int dusp_host_alloc(dusp_ctx *ctx, size_t n_bytes, void **out) {
    if (!ctx) return DUSP_ERR_ARG;
    return guarded(ctx->err, "dusp_host_alloc", [&]() -> int {
    if (!out) CTX_FAIL(ctx, DUSP_ERR_ARG, "dusp_host_alloc: out is NULL");
    *out = nullptr;
    if (n_bytes < 1 || n_bytes > ((size_t)1 << 40)) CTX_FAIL(ctx, DUSP_ERR_ARG, "dusp_host_alloc: size out of range");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> pool_lock(ctx->host_pool_mutex);
    // smallest free buffer that fits and is not wastefully large; pinning fresh pages is the slow part, so buffers are kept
    dusp_ctx::HostBuf *best = nullptr;
    for (auto &b : ctx->host_pool)
        if (!b.in_use && b.bytes >= n_bytes && b.bytes <= n_bytes + n_bytes / 4 + 65536 && (!best || b.bytes < best->bytes)) best = &b;
    if (best) {
        best->in_use = true;
        *out = best->p;
        return DUSP_OK;
    }
    size_t idle = 0;  // keep the pool's idle part bounded: free idle buffers first when the new one would push it past 4 GiB
    for (auto &b : ctx->host_pool) idle += b.in_use ? 0 : b.bytes;
    for (size_t k = ctx->host_pool.size(); k-- > 0 && idle + n_bytes > ((size_t)4 << 30);)
        if (!ctx->host_pool[k].in_use) {
            idle -= ctx->host_pool[k].bytes;
            (void)hipHostFree(ctx->host_pool[k].p);
            ctx->host_pool.erase(ctx->host_pool.begin() + (long)k);
        }
    ctx->host_pool.reserve(ctx->host_pool.size() + 1);
    void *p = nullptr;
    HIP_TRY(ctx, hipHostMalloc(&p, n_bytes, hipHostMallocDefault));
    ctx->host_pool.push_back({p, n_bytes, true});
    *out = p;
    return DUSP_OK;
    });
}

int dusp_host_free(dusp_ctx *ctx, void *p) {
    if (!ctx) return DUSP_ERR_ARG;
    if (!p) return DUSP_OK;
    {
        std::lock_guard<std::mutex> pool_lock(ctx->host_pool_mutex);  // (nothing else of the context is touched on this path: a finalizer thread may call it)
        for (auto &b : ctx->host_pool)
            if (b.p == p && b.in_use) {
                b.in_use = false;  // stays pinned for the next render of that size (dusp_ctx_destroy releases the pool)
                return DUSP_OK;
            }
    }
    CTX_FAIL(ctx, DUSP_ERR_ARG, "dusp_host_free: not a live buffer of this context");
}

int dusp_state_download(dusp_program *prog, size_t instance, size_t unit, double *out, size_t cap) {
    if (!prog) return DUSP_ERR_ARG;
    dusp_ctx *ctx = prog->ctx;
    return guarded(ctx->err, "dusp_state_download", [&]() -> int {
    if (!prog->rendered) CTX_FAIL(ctx, DUSP_ERR_STATE, "dusp_state_download: nothing has been rendered yet");
    const dusp::Graph &g = prog->P.g;
    if (unit >= g.units.size() || instance >= prog->last_n_inst || !out)
        CTX_FAIL(ctx, DUSP_ERR_ARG, "dusp_state_download: unit / instance out of range");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventSynchronize(prog->ev1));  // the render may have gone to a caller's stream: wait for what IT recorded
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
    });
}

int dusp_last_kernel_ms(dusp_program *prog, float *ms) {
    if (!prog || !ms) return DUSP_ERR_ARG;
    dusp_ctx *ctx = prog->ctx;
    if (!prog->rendered) CTX_FAIL(ctx, DUSP_ERR_STATE, "dusp_last_kernel_ms: nothing has been rendered yet");
    HIP_TRY(ctx, hipEventSynchronize(prog->ev1));
    HIP_TRY(ctx, hipEventElapsedTime(ms, prog->ev0, prog->ev1));
    return DUSP_OK;
}

int dusp_circuit_kernel_source(const double *desc, size_t n_words, int waves, int per_wave, int lds_table, int compile, char *text, size_t cap) {
    return guarded(g_error, "dusp_circuit_kernel_source", [&]() -> int {
    if (!desc || (cap && !text)) {
        g_error = "dusp_circuit_kernel_source: NULL argument";
        return DUSP_ERR_ARG;
    }
    if (waves < 1 || waves > 16 || per_wave < 1 || per_wave > 4) {
        g_error = "dusp_circuit_kernel_source: waves must be 1 .. 16 and per_wave 1 .. 4";
        return DUSP_ERR_ARG;
    }
    // (descriptor -> program -> plan -> options -> text: host code only, jit_codegen.hpp jit_source_from_descriptor — the same function the
    // sanitizer build of tests/native/hostcheck.cpp drives with malformed descriptors)
    dusp::JitSourceRequest rq;
    rq.waves = waves;
    rq.per_wave = per_wave;
    rq.continued = (lds_table & 2) != 0;
    rq.lean_recurrence = (lds_table & 4) != 0;  // (the Filter stage's recurrence loop with 4 P values per register set: what a render falls back to when the kernel spills)
    rq.lds_table = (lds_table & 1) != 0;
    rq.scan_knob = getenv("DUSP_FILTER_SCAN") ? atoi(getenv("DUSP_FILTER_SCAN")) : 1;
    rq.lean = !(getenv("DUSP_JIT_LEAN") && atoi(getenv("DUSP_JIT_LEAN")) == 0);
    rq.delay_line = getenv("DUSP_DELAY_LINE") ? atoi(getenv("DUSP_DELAY_LINE")) : 1;
    if (const char *range = getenv("DUSP_CUTOFF_RANGE"))  // (tests of the generator: "lo,hi" = what a renderer would have found in the Filters' cutoff columns)
        if (sscanf(range, "%lf,%lf", &rq.cutoff_lo, &rq.cutoff_hi) != 2) rq.cutoff_lo = rq.cutoff_hi = 0.0;
    dusp::JitSource src;
    std::string err;
    const int verdict = dusp::jit_source_from_descriptor(desc, n_words, rq, src, err);
    if (verdict != 0) {
        g_error = "dusp_circuit_kernel_source: " + err;
        return verdict == 1 ? DUSP_ERR_ARG : DUSP_ERR_UNSUPPORTED;
    }
    if (compile && !dusp::jit_compile_only(src.text, nullptr, err)) {
        g_error = "dusp_circuit_kernel_source: " + err;
        return DUSP_ERR_HIP;
    }
    if (cap) {
        const size_t n = std::min(cap - 1, src.text.size());
        std::memcpy(text, src.text.data(), n);
        text[n] = 0;
    }
    return (int)std::min<size_t>(src.text.size(), 0x7fffffff);
    });
}

const char *dusp_jit_cache_dir(void) { return dusp::jit_cache_directory(); }

int dusp_fill_device(dusp_ctx *ctx, float *d_out, size_t n_floats, float value, void *stream_) {
    if (!ctx) return DUSP_ERR_ARG;
    if (!d_out || (n_floats & 3) || ((uintptr_t)d_out & 15)) CTX_FAIL(ctx, DUSP_ERR_ARG, "dusp_fill_device: need a 16-byte aligned buffer of 4k floats");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, dusp::launch_fill(d_out, n_floats, value, stream_ ? (hipStream_t)stream_ : ctx->stream));
    return DUSP_OK;
}

}  // extern "C"
