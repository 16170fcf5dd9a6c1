// jit_args.hpp — kernel arguments of the per-circuit kernels (by value), shared by the host launcher (jit_engine.hip) and
// the device code hiprtc compiles at run time (jit_prelude.hpp + the text jit_codegen.hpp emits for one circuit).
#pragma once
#include "device_types.hpp"

namespace dusp {

struct JitArgs {
    const float *fk;           // the circuit's f32 constants (inlet constants), in the order the generated code numbers them
    const double *dk;          // its f64 constants (Ramp duration / y0 / y1, FixedMultiply factor, ...)
    const float *params;       // [n_params][n_inst]
    const float *tables;       // [kNumTables][table_stride]
    const float *inputs;       // [n_inputs][n_inst][n_samples]  host-generated signals (OP_INPUT)
    float *out;                // [n_inst][n_out][n_samples]
    double *state;             // [n_slots][n_pad]  end-of-render state (chunk engine's slot layout)
    const double *init_state;  // [n_slots]
    float *rings;              // [n_inst][ring_samples]
    // time-split rendering: [n_ops][n_inst][n_seg] phase totals / start phases of the scanned oscillators (2^-36 units, bit 63 = poisoned)
    unsigned long long *seg_sum, *seg_start;
    unsigned long long *debug;  // diagnostic builds (DUSP_JIT_PROFILE=1): [workgroup][4] cycle counts of wave 0; NULL otherwise
    uint64_t n_samples, ring_samples, clock0;
    uint32_t n_inst, n_pad, n_groups, sample_rate, table_stride, vec4_ok, n_out, pad0;
    // continued programs with delay lines / feedback (dusp_program_continue): every outlet's last chunk is parked in
    // saved_bufs [n_inst][n_bufs][256] when a launch ends (save_bufs) and picked up by the next one (resume); Delay rings are kept in
    // exactly the reference's state at launch boundaries
    float *saved_bufs;
    uint32_t resume, save_bufs, n_bufs, pad1;
    uint32_t n_seg, seg_groups;  // every instance is cut into n_seg segments of seg_groups chunks, one wavefront each (1: no split)
    // Filter circuits cut in time (jit_codegen.hpp jit_warm_chunks): warm = 1: every segment but the first starts one segment EARLY, from rest,
    // and stores nothing until its own chunks begin — by then its Filters' recurrences have, almost always, merged with the trajectory the
    // segment before computes; each Filter stage leaves what it held when its own chunks began and when they ended in
    // warm_records [stage][instance x segment][8] (y1 y2 at the start; y1 y2 x1 x2 at the end), and the host checks that every start equals
    // the end before it — then, by induction from the first segment, every sample is the sequential render's.  A launch that finishes a
    // render whose check failed starts at chunk g_first (n_seg = 1, warm = 0) from the state the last good segment left.
    double *warm_records;
    uint32_t warm, g_first;
};

}  // namespace dusp
