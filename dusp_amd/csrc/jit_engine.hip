// jit_engine.hip — the circuit compiler's back half: generated kernel text -> gfx950 code object (hiprtc, in process) ->
// loaded module -> launches.  One fused kernel per topologically sorted Circuit (the north star's kernel shape), compiled
// the first time a circuit STRUCTURE is rendered and reused for every circuit of that structure afterwards.
//
// Caches.  Code objects are kept process-wide, keyed by the generated text (constants are not part of it, jit_codegen.hpp);
// modules are loaded per device (a hipModule_t belongs to the device it was loaded on).  A render that finds its kernel
// costs a map lookup; a new structure costs one hiprtc compile (0.3-0.8 s) — in the foreground when the render is worth
// waiting for, else on the compile worker while the interpreter kernel renders this once (dusp_abi.hip decides).
// Code objects are also kept ON DISK across processes, by default under $XDG_CACHE_HOME/dusp-hip (~/.cache/dusp-hip);
// DUSP_JIT_CACHE=<directory> moves the cache, DUSP_JIT_CACHE=0 turns it off (read once per process, with the first
// context).  A cache file is keyed by everything the code object depends on — the text, the embedded device library's
// text, the compile options, the target and the hiprtc version — and carries its length and a hash of its payload: a
// truncated or damaged file is deleted and the kernel compiled again.
//
// One compile at a time per text: whoever needs a text that somebody is compiling waits for that compile (foreground) or
// leaves it to the worker (background).  Compiles run OUTSIDE the lock, so lookups of finished kernels never wait for one.
//
// No fallback hides a failure here: if hiprtc or the module load fails the render call fails with the compiler's log — a
// failed background compile is remembered with its log and fails the next render of that structure.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <dirent.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>

#include <algorithm>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "fused_plan.hpp"
#include "jit_args.hpp"
#include "jit_engine.hpp"

// the device library's text, embedded by jit_embed.S (.incbin of the very files hipcc compiles into the AOT kernels)
extern "C" const char dusp_src_device_types[], dusp_src_device_util[], dusp_src_filter_lamda[], dusp_src_map_ops[], dusp_src_repeat_add[],
    dusp_src_jit_args[], dusp_src_jit_prelude[];

namespace dusp {

namespace {

std::mutex g_mutex;
std::condition_variable g_cv;                     // a compile has finished (either outcome), or the worker has work / must leave
std::map<std::string, std::vector<char>> g_code;  // generated text -> code object
std::map<std::string, std::string> g_failed;      // generated text -> the compiler's log of a compile that failed
std::set<std::string> g_compiling;                // texts somebody is compiling right now (worker or a foreground caller)
std::deque<std::string> g_queue;                  // texts waiting for the worker
std::thread g_worker;
bool g_worker_started = false, g_leaving = false;
std::string g_cache_dir;                          // "" = no disk cache
uint64_t g_cache_cap_bytes = (uint64_t)256 << 20; // the cache directory is kept below this (DUSP_JIT_CACHE_MAX_MB): least recently used files go first
std::once_flag g_configured;

// the numerics contract of every kernel in this library: no contraction, no fast-math, correctly rounded division
const char *const kCompileOptions[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                                       "-fhip-fp32-correctly-rounded-divide-sqrt"};
constexpr int kNumCompileOptions = (int)(sizeof kCompileOptions / sizeof kCompileOptions[0]);

uint64_t fnv1a(const char *p, size_t n, uint64_t h = 1469598103934665603ull) {
    for (size_t i = 0; i < n; i++) h = (h ^ (unsigned char)p[i]) * 1099511628211ull;
    return h;
}

bool make_dirs(const std::string &path) {
    for (size_t i = 1; i <= path.size(); i++)
        if (i == path.size() || path[i] == '/') {
            const std::string part = path.substr(0, i);
            if (::mkdir(part.c_str(), 0700) != 0 && errno != EEXIST) return false;
        }
    struct stat st;
    return ::stat(path.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}

// Where code objects are kept across processes.  Read ONCE (std::call_once from jit_configure / first use).
void configure_once() {
    const char *e = getenv("DUSP_JIT_CACHE");
    std::string dir;
    if (e && *e) {
        if (!std::strcmp(e, "0") || !std::strcmp(e, "off")) return;
        dir = e;
    } else {
        const char *xdg = getenv("XDG_CACHE_HOME"), *home = getenv("HOME");
        if (xdg && *xdg) dir = std::string(xdg) + "/dusp-hip";
        else if (home && *home) dir = std::string(home) + "/.cache/dusp-hip";
        else return;
    }
    if (!make_dirs(dir)) return;  // (a directory that cannot be made: no disk cache, nothing fails)
    // code objects are loaded and RUN: only a directory of this user's that nobody else may write to is trusted with them
    struct stat st;
    if (::stat(dir.c_str(), &st) != 0 || st.st_uid != ::geteuid() || (st.st_mode & (S_IWGRP | S_IWOTH))) return;
    g_cache_dir = dir;
    if (const char *cap = getenv("DUSP_JIT_CACHE_MAX_MB")) {
        const long mb = std::atol(cap);
        if (mb > 0) g_cache_cap_bytes = (uint64_t)mb << 20;
    }
}

// key of the disk cache: everything a code object depends on
uint64_t cache_key(const std::string &text) {
    uint64_t h = fnv1a(text.data(), text.size());
    for (const char *src : {dusp_src_device_types, dusp_src_device_util, dusp_src_filter_lamda, dusp_src_map_ops, dusp_src_repeat_add, dusp_src_jit_args,
                            dusp_src_jit_prelude})
        h = fnv1a(src, std::strlen(src) + 1, h);
    for (const char *o : kCompileOptions) h = fnv1a(o, std::strlen(o) + 1, h);
    int major = 0, minor = 0;
    (void)hiprtcVersion(&major, &minor);
    const int version[3] = {major, minor, HIP_VERSION_PATCH};
    return fnv1a((const char *)version, sizeof version, h);
}
std::string disk_path(const std::string &text) {
    if (g_cache_dir.empty()) return std::string();
    char name[64];
    std::snprintf(name, sizeof name, "/dusp_%016llx_%zu.hsaco", (unsigned long long)cache_key(text), text.size());
    return g_cache_dir + name;
}
struct DiskHeader {
    char magic[8];  // "DUSPHSA2"
    uint64_t bytes, hash;  // of the code object behind the header
    uint64_t text_hash;    // a second, independent hash of the kernel text: with the key in the file's name (and the text's length) 128 bits say whose kernel this is
};
uint64_t second_hash(const std::string &text) {  // (xorshift-multiply over 8-byte words: nothing in common with FNV-1a's byte walk)
    uint64_t h = 0x9e3779b97f4a7c15ull ^ (uint64_t)text.size();
    for (size_t i = 0; i < text.size(); i += 8) {
        uint64_t w = 0;
        std::memcpy(&w, text.data() + i, std::min<size_t>(8, text.size() - i));
        h ^= w;
        h *= 0xff51afd7ed558ccdull;
        h ^= h >> 32;
    }
    return h;
}
// keep the directory below its cap: the files this library wrote, least recently used (loaded or stored) first
void disk_evict() {
    DIR *d = ::opendir(g_cache_dir.c_str());
    if (!d) return;
    struct Entry { std::string path; uint64_t bytes; time_t used; };
    std::vector<Entry> files;
    uint64_t total = 0;
    while (struct dirent *e = ::readdir(d)) {
        const std::string name = e->d_name;
        if (name.rfind("dusp_", 0) != 0 || name.size() < 6 || name.substr(name.size() - 6) != ".hsaco") continue;
        struct stat st;
        const std::string path = g_cache_dir + "/" + name;
        if (::stat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) continue;
        files.push_back({path, (uint64_t)st.st_size, st.st_mtime});
        total += (uint64_t)st.st_size;
    }
    ::closedir(d);
    if (total <= g_cache_cap_bytes) return;
    std::sort(files.begin(), files.end(), [](const Entry &a, const Entry &b) { return a.used < b.used; });
    for (const Entry &f : files) {
        if (total <= g_cache_cap_bytes / 4 * 3) break;
        if (std::remove(f.path.c_str()) == 0) total -= f.bytes;
    }
}
bool disk_load(const std::string &text, std::vector<char> &code) {
    const std::string path = disk_path(text);
    if (path.empty()) return false;
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    DiskHeader h{};
    bool ok = std::fread(&h, sizeof h, 1, f) == 1 && !std::memcmp(h.magic, "DUSPHSA2", 8) && h.bytes > 0 && h.bytes < ((uint64_t)64 << 20) && h.text_hash == second_hash(text);
    if (ok) {
        code.resize((size_t)h.bytes);
        ok = std::fread(code.data(), 1, code.size(), f) == code.size() && std::fgetc(f) == EOF && fnv1a(code.data(), code.size()) == h.hash;
    }
    std::fclose(f);
    if (!ok) {  // truncated, damaged, from another format or — the second hash — another text's: gone, the kernel is compiled again
        code.clear();
        std::remove(path.c_str());
    } else
        (void)::utimes(path.c_str(), nullptr);  // (used now: the eviction goes by this)
    return ok;
}
void disk_store(const std::string &text, const std::vector<char> &code) {
    const std::string path = disk_path(text);
    if (path.empty()) return;
    const std::string tmp = path + ".tmp" + std::to_string((long long)::getpid()) + "_" + std::to_string((unsigned long long)fnv1a((const char *)&code, sizeof(void *)));
    FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) return;
    DiskHeader h{};
    std::memcpy(h.magic, "DUSPHSA2", 8);
    h.bytes = code.size();
    h.hash = fnv1a(code.data(), code.size());
    h.text_hash = second_hash(text);
    const bool ok = std::fwrite(&h, sizeof h, 1, f) == 1 && std::fwrite(code.data(), 1, code.size(), f) == code.size();
    if (std::fclose(f) != 0 || !ok || std::rename(tmp.c_str(), path.c_str()) != 0) std::remove(tmp.c_str());
    else disk_evict();
}
void disk_forget(const std::string &text) {
    const std::string path = disk_path(text);
    if (!path.empty()) std::remove(path.c_str());
}

struct Loaded {
    hipModule_t module = nullptr;
    std::map<std::string, hipFunction_t> kernels;
    std::map<std::string, int> scratch_bytes;
};
std::map<std::pair<int, std::string>, Loaded> g_loaded;  // (device, text) -> module on that device

bool compile_text(const std::string &text, std::vector<char> &code, std::string &err) {
    const char *headers[] = {dusp_src_device_types, dusp_src_device_util, dusp_src_filter_lamda, dusp_src_map_ops, dusp_src_repeat_add, dusp_src_jit_args,
                             dusp_src_jit_prelude};
    const char *names[] = {"device_types.hpp", "device_util.hpp", "filter_lamda.hpp", "map_ops.hpp", "repeat_add.hpp", "jit_args.hpp", "jit_prelude.hpp"};
    hiprtcProgram prog = nullptr;
    hiprtcResult r = hiprtcCreateProgram(&prog, text.c_str(), "dusp_circuit.hip", 7, headers, names);
    if (r != HIPRTC_SUCCESS) {
        err = std::string("hiprtcCreateProgram: ") + hiprtcGetErrorString(r);
        return false;
    }
    r = hiprtcCompileProgram(prog, kNumCompileOptions, const_cast<const char **>(kCompileOptions));
    if (r != HIPRTC_SUCCESS) {
        size_t n = 0;
        hiprtcGetProgramLogSize(prog, &n);
        std::string log(n, '\0');
        if (n) hiprtcGetProgramLog(prog, &log[0]);
        err = std::string("hiprtc: ") + hiprtcGetErrorString(r) + "\n" + log;
        hiprtcDestroyProgram(&prog);
        return false;
    }
    size_t n = 0;
    hiprtcGetCodeSize(prog, &n);
    code.resize(n);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    if (n == 0) err = "hiprtc: empty code object";
    return n > 0;
}
// compile_text behind a firewall: nothing thrown in there (bad_alloc ...) may leave a thread of this library
bool compile_guarded(const std::string &text, std::vector<char> &code, std::string &err) noexcept {
    try {
        return compile_text(text, code, err);
    } catch (const std::exception &e) {
        try { err = std::string("circuit compiler: ") + e.what(); } catch (...) {}
    } catch (...) {
        try { err = "circuit compiler: unknown internal error"; } catch (...) {}
    }
    return false;
}

// The code object of `text`, whoever gets it first: memory, the disk cache, a compile by this thread, or a compile somebody
// else is running (waited for).  Called with `lock` held; the compile itself runs with it released.  use_disk = false: skip
// the disk cache (its file has just failed to load as a module).
bool obtain_code(std::unique_lock<std::mutex> &lock, const std::string &text, bool use_disk, std::string &err) {
    for (;;) {
        if (g_code.count(text)) return true;
        auto bad = g_failed.find(text);
        if (bad != g_failed.end()) {
            err = bad->second;
            return false;
        }
        if (!g_compiling.count(text)) break;
        g_cv.wait(lock);  // the worker (or another caller) is at it
    }
    g_compiling.insert(text);
    for (auto q = g_queue.begin(); q != g_queue.end(); ++q)  // (the worker need not pick it up any more)
        if (*q == text) { g_queue.erase(q); break; }
    lock.unlock();
    std::vector<char> code;
    std::string log;
    bool ok = false;
    try {
        ok = use_disk && disk_load(text, code);
        if (!ok) {
            ok = compile_guarded(text, code, log);
            if (ok) disk_store(text, code);
        }
    } catch (...) {
        ok = false;
        if (log.empty()) log = "circuit compiler: out of memory";
    }
    lock.lock();
    g_compiling.erase(text);
    try {
        if (ok) g_code.emplace(text, std::move(code));
        else g_failed[text] = log;
    } catch (...) {
        ok = false;
    }
    g_cv.notify_all();
    if (!ok) err = log;
    return ok;
}

void worker_main() noexcept {
    std::unique_lock<std::mutex> lock(g_mutex);
    for (;;) {
        while (g_queue.empty() && !g_leaving) g_cv.wait(lock);
        if (g_leaving) return;
        std::string text;
        try {
            text = g_queue.front();
        } catch (...) {
            g_queue.pop_front();
            continue;
        }
        g_queue.pop_front();
        if (g_code.count(text) || g_failed.count(text) || g_compiling.count(text)) continue;
        std::string err;
        try {
            (void)obtain_code(lock, text, true, err);  // (a failure stays in g_failed: the next render of that structure reports it)
        } catch (...) {
            if (!lock.owns_lock()) lock.lock();
            g_compiling.erase(text);
            g_cv.notify_all();
        }
    }
}

}  // namespace

void jit_configure() { std::call_once(g_configured, configure_once); }

const char *jit_cache_directory() {
    jit_configure();
    return g_cache_dir.c_str();
}

bool jit_compile_only(const std::string &text, size_t *code_bytes, std::string &err) {
    jit_configure();
    std::unique_lock<std::mutex> lock(g_mutex);
    if (!obtain_code(lock, text, true, err)) return false;
    if (code_bytes) *code_bytes = g_code[text].size();
    return true;
}

bool jit_code_ready(const std::string &text) {
    jit_configure();
    std::unique_lock<std::mutex> lock(g_mutex);
    if (g_code.count(text) || g_failed.count(text)) return true;  // (a failed text is "ready" to fail the render with its log)
    if (g_compiling.count(text)) return false;
    lock.unlock();
    std::vector<char> code;
    if (!disk_load(text, code)) return false;
    lock.lock();
    g_code.emplace(text, std::move(code));
    return true;
}

void jit_compile_in_background(const std::string &text) {
    jit_configure();
    std::lock_guard<std::mutex> lock(g_mutex);
    if (g_code.count(text) || g_failed.count(text) || g_compiling.count(text)) return;
    for (const std::string &q : g_queue)
        if (q == text) return;
    if (g_queue.size() >= 4096) g_queue.pop_front();  // (a bound that recovers: the oldest waiting structure is simply asked for again later)
    g_queue.push_back(text);
    if (!g_worker_started) {  // ONE worker: compiles are -O3 hiprtc runs of 0.3-0.8 s each, side by side they only slow each other
        g_worker_started = true;
        g_worker = std::thread(worker_main);
        std::atexit([] {  // the process does not leave while the worker is inside a compile
            {
                std::lock_guard<std::mutex> l(g_mutex);
                g_leaving = true;
                g_queue.clear();
            }
            g_cv.notify_all();
            if (g_worker.joinable()) g_worker.join();
        });
    }
    g_cv.notify_all();
}

bool jit_get_kernel(int device, const std::string &text, const std::string &name, hipFunction_t *fn, int *scratch_bytes, std::string &err) {
    jit_configure();
    std::unique_lock<std::mutex> lock(g_mutex);
    for (int attempt = 0;; attempt++) {
        {
            auto it = g_loaded.find({device, text});
            if (it != g_loaded.end() && it->second.module) break;
        }
        if (!obtain_code(lock, text, attempt == 0, err)) return false;
        hipModule_t module = nullptr;
        const hipError_t e = hipModuleLoadData(&module, g_code[text].data());
        if (e == hipSuccess) {
            g_loaded[{device, text}].module = module;
            break;
        }
        (void)hipGetLastError();
        // A code object the device does not take: if it came from the disk cache it may be stale in a way the key does not see
        // (or damaged in a way the hash cannot, i.e. written wrong): forget file and blob, compile once, then give up.
        g_code.erase(text);
        disk_forget(text);
        if (attempt >= 1) {
            err = std::string("hipModuleLoadData: ") + hipGetErrorString(e);
            return false;
        }
    }
    Loaded &L = g_loaded[{device, text}];
    auto k = L.kernels.find(name);
    if (k == L.kernels.end()) {
        hipFunction_t f = nullptr;
        hipError_t e = hipModuleGetFunction(&f, L.module, name.c_str());
        if (e != hipSuccess) {
            err = "hipModuleGetFunction(" + name + "): " + hipGetErrorString(e);
            return false;
        }
        int scratch = 0;
        (void)hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, f);
        L.scratch_bytes[name] = scratch;
        k = L.kernels.emplace(name, f).first;
    }
    *fn = k->second;
    if (scratch_bytes) *scratch_bytes = L.scratch_bytes[name];
    return true;
}

hipError_t jit_launch(hipFunction_t fn, const JitArgs &A, unsigned grid, unsigned block, hipStream_t stream) {
    JitArgs args = A;
    size_t size = sizeof args;
    void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    return hipModuleLaunchKernel(fn, grid, 1, 1, block, 1, 1, 0, stream, nullptr, config);
}

// Per-instance (parameter) delays: which regime does every instance's value fall into?  One workgroup walks the parameter columns;
// out[k] = OR of delay_value_regime over the instances, for entry k = (parameter slot, ring length, MonoDelay?).
__global__ void dusp_classify_delays_kernel(const float *params, uint32_t n_inst, const int64_t *entries, int n_entries, int *out) {
    for (int k = 0; k < n_entries; ++k) {
        const int64_t slot = entries[3 * k], ring_len = entries[3 * k + 1];
        const bool mono = entries[3 * k + 2] != 0;
        int bits = 0;
        for (uint32_t i = threadIdx.x; i < n_inst; i += blockDim.x) bits |= delay_value_regime(params[(size_t)slot * n_inst + i], ring_len, mono);
        if (bits) atomicOr(&out[k], bits);
    }
}
// Per-instance (parameter) cutoffs of Filters: the range every instance's value lies in.  out[3 k .. 3 k + 2] (zeroed by the caller) = for
// entry k (a parameter slot): 0x7fffffff - bits of the smallest value, bits of the largest, 1 if some value is NaN / Inf / not positive.
__global__ void dusp_column_range_kernel(const float *params, uint32_t n_inst, const int *slots, int n_entries, unsigned *out) {
    for (int k = 0; k < n_entries; ++k) {
        unsigned lo = 0u, hi = 0u, bad = 0u;
        for (uint32_t i = threadIdx.x; i < n_inst; i += blockDim.x) {
            const float v = params[(size_t)slots[k] * n_inst + i];
            if (!(v > 0.f && v <= 3.0e38f)) bad = 1u;
            else {
                const unsigned b = __float_as_uint(v);  // (positive floats order like their bits)
                lo = max(lo, 0x7fffffffu - b);
                hi = max(hi, b);
            }
        }
        if (lo) atomicMax(&out[3 * k], lo);
        if (hi) atomicMax(&out[3 * k + 1], hi);
        if (bad) atomicOr(&out[3 * k + 2], 1u);
    }
}
hipError_t jit_launch_column_range(const float *params, uint32_t n_inst, const int *d_slots, int n_entries, unsigned *d_out, hipStream_t stream) {
    hipLaunchKernelGGL(dusp_column_range_kernel, dim3(1), dim3(1024), 0, stream, params, n_inst, d_slots, n_entries, d_out);
    return hipGetLastError();
}
hipError_t jit_launch_classify_delays(const float *params, uint32_t n_inst, const int64_t *d_entries, int n_entries, int *d_out, hipStream_t stream) {
    hipLaunchKernelGGL(dusp_classify_delays_kernel, dim3(1), dim3(1024), 0, stream, params, n_inst, d_entries, n_entries, d_out);
    return hipGetLastError();
}

// Start phase of every segment from the segments' phase totals: a serial modular prefix per (scanned oscillator, instance) —
// n_seg additions, one thread each.  A poisoned segment poisons everything after it.
__global__ void dusp_jit_prefix_kernel(const unsigned long long *seg_sum, unsigned long long *seg_start, const double *init_state, const int *scan_slot,
                                       const int *scan_level, int n_scans, int level, uint32_t n_inst, uint32_t n_seg, uint32_t sample_rate) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint32_t)n_scans * n_inst) return;
    const uint32_t s = t / n_inst, inst = t - s * n_inst;
    if (scan_level[s] != level) return;
    const unsigned long long S = (unsigned long long)sample_rate << 36;
    unsigned long long phase = (unsigned long long)(init_state[scan_slot[s]] * 68719476736.0), poison = 0;
    const size_t base = ((size_t)s * n_inst + inst) * n_seg;
    for (uint32_t k = 0; k < n_seg; ++k) {
        seg_start[base + k] = phase | poison;
        const unsigned long long v = seg_sum[base + k];
        phase += v & ~(1ull << 63);  // both below S
        if (phase >= S) phase -= S;
        poison |= v & (1ull << 63);
    }
}

hipError_t jit_launch_prefix(const unsigned long long *seg_sum, unsigned long long *seg_start, const double *init_state, const int *d_scan_slot,
                             const int *d_scan_level, int n_scans, int level, uint32_t n_inst, uint32_t n_seg, uint32_t sample_rate, hipStream_t stream) {
    const uint32_t n = (uint32_t)n_scans * n_inst;
    hipLaunchKernelGGL(dusp_jit_prefix_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, seg_sum, seg_start, init_state, d_scan_slot, d_scan_level, n_scans,
                       level, n_inst, n_seg, sample_rate);
    return hipGetLastError();
}

}  // namespace dusp
