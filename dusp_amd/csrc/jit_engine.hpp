// jit_engine.hpp — interface of the circuit compiler's back half (jit_engine.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "jit_args.hpp"

namespace dusp {

// Reads DUSP_JIT_CACHE / XDG_CACHE_HOME once per process (called when a context is created) and makes the cache directory.
void jit_configure();
const char *jit_cache_directory();  // "" when there is no disk cache
// Compile generated kernel text for gfx950 (no device needed: used by the CPU test of the generator too).  Cached by text.
bool jit_compile_only(const std::string &text, size_t *code_bytes, std::string &err);
// Is the code object of `text` at hand (memory, or the disk cache) — or known to fail?  If not, a render may leave its compile
// to the background worker and use the interpreter kernel meanwhile.
bool jit_code_ready(const std::string &text);
void jit_compile_in_background(const std::string &text);
// The kernel `name` of that text, loaded on `device` (compiles / loads on first use).  scratch_bytes: private memory the
// compiler had to spill into (0 when the kernel fits its registers).
bool jit_get_kernel(int device, const std::string &text, const std::string &name, hipFunction_t *fn, int *scratch_bytes, std::string &err);
hipError_t jit_launch(hipFunction_t fn, const JitArgs &A, unsigned grid, unsigned block, hipStream_t stream);
// regimes of per-instance delays (fused_plan.hpp delay_value_regime): d_entries = [slot, ring_len, mono] x n_entries, d_out zeroed by the caller
hipError_t jit_launch_classify_delays(const float *params, uint32_t n_inst, const int64_t *d_entries, int n_entries, int *d_out, hipStream_t stream);
// the range of per-instance Filter cutoffs (jit_codegen.hpp jit_filter_scan_ok): d_slots = parameter slots, d_out [3 n_entries] zeroed by the caller
hipError_t jit_launch_column_range(const float *params, uint32_t n_inst, const int *d_slots, int n_entries, unsigned *d_out, hipStream_t stream);
hipError_t jit_launch_prefix(const unsigned long long *seg_sum, unsigned long long *seg_start, const double *init_state, const int *d_scan_slot,
                             const int *d_scan_level, int n_scans, int level, uint32_t n_inst, uint32_t n_seg, uint32_t sample_rate, hipStream_t stream);

}  // namespace dusp
