// jit_engine.hip — the circuit compiler's back half: generated kernel text -> gfx950 code object (hiprtc, in process) ->
// loaded module -> launches.  One fused kernel per topologically sorted Circuit (the north star's kernel shape), compiled
// the first time a circuit STRUCTURE is rendered and reused for every circuit of that structure afterwards.
//
// Caches.  Code objects are kept process-wide, keyed by the generated text (constants are not part of it, jit_codegen.hpp);
// modules are loaded per device (a hipModule_t belongs to the device it was loaded on).  A render that finds its kernel
// costs a map lookup; a new structure costs one hiprtc compile (0.3-0.8 s) — in the foreground when the render is worth
// waiting for, else in a background thread while the interpreter kernel renders this once (dusp_abi.hip decides).
// DUSP_JIT_CACHE=<directory> additionally keeps code objects on disk across processes (off unless set).
//
// No fallback hides a failure here: if hiprtc or the module load fails the render call fails with the compiler's log.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "jit_args.hpp"
#include "jit_engine.hpp"

// the device library's text, embedded by jit_embed.S (.incbin of the very files hipcc compiles into the AOT kernels)
extern "C" const char dusp_src_device_types[], dusp_src_device_util[], dusp_src_map_ops[], dusp_src_repeat_add[], dusp_src_jit_args[],
    dusp_src_jit_prelude[];

namespace dusp {

namespace {

std::mutex g_mutex;
std::map<std::string, std::vector<char>> g_code;  // generated text -> code object
std::set<std::string> g_compiling;                // texts a background thread is compiling right now
std::vector<std::thread> g_threads;
std::atomic<bool> g_exit_hook{false};

uint64_t fnv1a(const char *p, size_t n, uint64_t h = 1469598103934665603ull) {
    for (size_t i = 0; i < n; i++) h = (h ^ (unsigned char)p[i]) * 1099511628211ull;
    return h;
}
// file of the disk cache for a text: the key covers the device library's text too (a rebuilt library invalidates the cache)
std::string disk_path(const std::string &text) {
    const char *dir = getenv("DUSP_JIT_CACHE");
    if (!dir || !*dir) return std::string();
    uint64_t h = fnv1a(text.data(), text.size());
    for (const char *src : {dusp_src_device_types, dusp_src_device_util, dusp_src_map_ops, dusp_src_repeat_add, dusp_src_jit_args, dusp_src_jit_prelude})
        h = fnv1a(src, std::strlen(src), h);
    char name[64];
    std::snprintf(name, sizeof name, "/dusp_%016llx_%zu.hsaco", (unsigned long long)h, text.size());
    return std::string(dir) + name;
}
bool disk_load(const std::string &text, std::vector<char> &code) {
    const std::string path = disk_path(text);
    if (path.empty()) return false;
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    bool ok = n > 0 && n < (64 << 20);
    if (ok) {
        code.resize((size_t)n);
        ok = std::fread(code.data(), 1, (size_t)n, f) == (size_t)n;
    }
    std::fclose(f);
    return ok;
}
void disk_store(const std::string &text, const std::vector<char> &code) {
    const std::string path = disk_path(text);
    if (path.empty()) return;
    const std::string tmp = path + ".tmp" + std::to_string((unsigned long long)fnv1a((const char *)&code, sizeof(void *)));
    FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) return;
    const bool ok = std::fwrite(code.data(), 1, code.size(), f) == code.size();
    std::fclose(f);
    if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) std::remove(tmp.c_str());
}

struct Loaded {
    hipModule_t module = nullptr;
    std::map<std::string, hipFunction_t> kernels;
    std::map<std::string, int> scratch_bytes;
};
std::map<std::pair<int, std::string>, Loaded> g_loaded;  // (device, text) -> module on that device

bool compile_text(const std::string &text, std::vector<char> &code, std::string &err) {
    const char *headers[] = {dusp_src_device_types, dusp_src_device_util, dusp_src_map_ops, dusp_src_repeat_add, dusp_src_jit_args, dusp_src_jit_prelude};
    const char *names[] = {"device_types.hpp", "device_util.hpp", "map_ops.hpp", "repeat_add.hpp", "jit_args.hpp", "jit_prelude.hpp"};
    hiprtcProgram prog = nullptr;
    hiprtcResult r = hiprtcCreateProgram(&prog, text.c_str(), "dusp_circuit.hip", 6, headers, names);
    if (r != HIPRTC_SUCCESS) {
        err = std::string("hiprtcCreateProgram: ") + hiprtcGetErrorString(r);
        return false;
    }
    // the numerics contract of every kernel in this library: no contraction, no fast-math, correctly rounded division
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt"};
    r = hiprtcCompileProgram(prog, (int)(sizeof opts / sizeof opts[0]), opts);
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
    return n > 0;
}

}  // namespace

// the code object of `text`: from memory, from the disk cache, or compiled now (g_mutex held by the caller)
static bool code_for(const std::string &text, const std::vector<char> **out, std::string &err) {
    auto it = g_code.find(text);
    if (it == g_code.end()) {
        std::vector<char> code;
        if (!disk_load(text, code)) {
            if (!compile_text(text, code, err)) return false;
            disk_store(text, code);
        }
        it = g_code.emplace(text, std::move(code)).first;
    }
    *out = &it->second;
    return true;
}

bool jit_compile_only(const std::string &text, size_t *code_bytes, std::string &err) {
    std::lock_guard<std::mutex> lock(g_mutex);
    const std::vector<char> *code = nullptr;
    if (!code_for(text, &code, err)) return false;
    if (code_bytes) *code_bytes = code->size();
    return true;
}

bool jit_code_ready(const std::string &text) {
    std::lock_guard<std::mutex> lock(g_mutex);
    if (g_code.count(text)) return true;
    std::vector<char> code;
    if (!disk_load(text, code)) return false;
    g_code.emplace(text, std::move(code));
    return true;
}

void jit_compile_in_background(const std::string &text) {
    std::lock_guard<std::mutex> lock(g_mutex);
    if (g_code.count(text) || g_compiling.count(text) || g_threads.size() >= 256) return;
    if (!g_exit_hook.exchange(true))  // the process does not leave while a compile is running in one of these threads
        std::atexit([] {
            std::vector<std::thread> threads;
            {
                std::lock_guard<std::mutex> l(g_mutex);
                threads.swap(g_threads);
            }
            for (auto &t : threads)
                if (t.joinable()) t.join();
        });
    g_compiling.insert(text);
    g_threads.emplace_back([text] {
        std::vector<char> code;
        std::string err;
        const bool ok = compile_text(text, code, err);  // (outside the lock: renders go on meanwhile)
        std::lock_guard<std::mutex> l(g_mutex);
        if (ok) {
            disk_store(text, code);
            g_code.emplace(text, std::move(code));
        }
        g_compiling.erase(text);  // (a failed compile is retried, and reported, by the next render that waits for it)
    });
}

bool jit_get_kernel(int device, const std::string &text, const std::string &name, hipFunction_t *fn, int *scratch_bytes, std::string &err) {
    std::lock_guard<std::mutex> lock(g_mutex);
    Loaded &L = g_loaded[{device, text}];
    if (!L.module) {
        const std::vector<char> *code = nullptr;
        if (!code_for(text, &code, err)) return false;
        hipError_t e = hipModuleLoadData(&L.module, code->data());
        if (e != hipSuccess) {
            L.module = nullptr;
            err = std::string("hipModuleLoadData: ") + hipGetErrorString(e);
            return false;
        }
    }
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
