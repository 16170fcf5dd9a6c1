// device_types.hpp — PODs shared by the host compiler and the HIP kernels.
#pragma once
#if defined(__HIPCC_RTC__)  // compiled at run time by hiprtc (jit_engine.hip): no host headers there, the basic types come from here
typedef unsigned int uint32_t;
typedef int int32_t;
typedef unsigned long long uint64_t;
typedef long long int64_t;
typedef unsigned long uintptr_t;
#else
#include <cstdint>
#endif

namespace dusp {

enum : int {
    OP_OSC = 1, OP_RAMP, OP_MULTIPLY, OP_SUM, OP_FILTER, OP_DELAY, OP_CB_READER, OP_CB_WRITER, OP_REPEATER,
    // elementwise maps
    OP_SUBTRACT, OP_DIVIDE, OP_POLARITY_INVERT, OP_ABS, OP_CLIP, OP_HARD_CLIP_ABOVE, OP_HARD_CLIP_BELOW,
    OP_SECONDS_TO_SAMPLES, OP_FIXED_MULTIPLY, OP_GAIN, OP_DECIBEL_TO_SCALER, OP_SEMITONE_TO_RATIO, OP_POW,
    // delay / filter family, per-channel oscillator
    OP_FIXED_DELAY, OP_COMB_FILTER, OP_ALL_PASS, OP_MONO_DELAY, OP_READBACK_DELAY, OP_MULTI_OSC,
    // rest of the elementwise sweep.  ConcatChannels / PickChannel exist in descriptors only: the
    // host compiler turns them into per-channel copies (OP_REPEATER device ops).
    OP_PAN, OP_MIDI_TO_FREQUENCY, OP_RESCALE, OP_CROSS_FADER, OP_VECTOR_MAGNITUDE, OP_TIMER, OP_SAMPLE_RATE_REDUX,
    OP_CONCAT_CHANNELS, OP_PICK_CHANNEL,
    OP_SHAPE, OP_AHD,  // envelopes
    OP_HOST_ONLY,      // a unit that produces no signal and acts through host callbacks (Retriggerer): keeps its place in the unit list
    OP_INPUT,          // a unit whose signal the HOST computes (Noise: Math.random() per sample): reads input stream `attr`
    OP_RETRIGGER,      // Retriggerer whose target is a Shape / AHD of the same circuit: ticks on the device and triggers it in place
    OP_MAP_FIRST = OP_SUBTRACT, OP_MAP_LAST = OP_POW,    // stateless maps of at most two operands (map_apply)
    OP_WIDE_FIRST = OP_PAN, OP_WIDE_LAST = OP_VECTOR_MAGNITUDE  // stateless maps of up to kMaxIn operands (map_wide)
};
constexpr int kMaxIn = 5;  // Rescale has five inlets
enum : int { IN_CONST = 0, IN_CONNECT = 1, IN_PARAM = 2 };   // descriptor inlet kinds
enum : int { SRC_CONST = 0, SRC_BUF = 1, SRC_PARAM = 2 };    // device operand kinds
constexpr int kNumTables = 9;      // 0-4 oscillator wave tables, 5-8 Shape tables (decay, attack, semiSine, decaySquared)
constexpr int kFirstShapeTable = 5;
constexpr int kChunk = 256;
// bytes of the oscillators' half-table image in LDS (device_util.hpp Table<1>: entries M + 1 .. 0 in blocks of 33 words); here, without
// any HIP attribute, so that host-only code (the circuit compiler's option logic, its sanitizer build) can size a workgroup's LDS too
constexpr unsigned long half_table_image_bytes(uint32_t sample_rate) {
    return ((unsigned long)((sample_rate / 2 + 1) + ((sample_rate / 2 + 1) >> 5) + 2) * 4ul + 15ul) & ~15ul;
}
constexpr int kMaxWarmChunks = 8;  // chunks at the start of a render that may run their own op list (program.hpp infer_channels)

// A/B switches for tools/ and tests.  They are read from the environment ONCE, when a context is created
// (dusp_ctx_create), never on the launch path; the defaults are the product.
struct Knobs {
    int fused_table_global = 0;  // DUSP_FUSED_TABLE=global: wave table from L2 even when the LDS half-table applies
    int fused_R = 4;             // DUSP_FUSED_R: voices per work item (1, 4, 8)
    int fused_items = 0;         // DUSP_FUSED_ITEMS: work items per resident wave (0: 4, or 2 when that would cut the render into items of under 64 chunks)
    int fused_fx32 = 1;          // DUSP_FUSED_FX32=0: no 32.32 fixed-point phase path
    int fused_segmajor = 0;      // DUSP_FUSED_SEGMAJOR=1: item order
    int wave_segments = -1;      // DUSP_WAVE_SEGMENTS=n: force n time segments (0 / 1: off; -1: automatic)
    int wave_max_waves = 0;      // DUSP_WAVE_MAX_WAVES=n: cap the wavefronts per workgroup (0: no cap)
    int wave_jit = 1;            // DUSP_WAVE_JIT=0: keep wave-engine programs on the interpreter; 2: always wait for a circuit's kernel
                                 // (1: a render the interpreter finishes sooner than a compile runs there while the kernel compiles in the background)
    int wave_per_wave = 0;       // DUSP_WAVE_PER_WAVE=n: circuit instances per wavefront in compiled kernels (0: automatic, up to 4)
    int jit_lean = 1;            // DUSP_JIT_LEAN=0: compiled kernels' constant-f oscillators keep the 32.32 form (no delta lerp: A/B)
    int jit_log = 0;             // DUSP_JIT_LOG=1: the geometry search's steps on stderr (wavefronts x instances, Filter block, scratch bytes per lane)
    int jit_lds_table = 1;       // DUSP_JIT_LDS_TABLE=0: compiled kernels look wave tables up in HBM / L2 (no LDS image)
    int jit_spill_bytes = 96;    // DUSP_JIT_SPILL: scratch bytes per lane a compiled kernel may use before it is rebuilt for a smaller geometry (round 4: 96 — `delay(osc, lfo)` spills 92
                                 // bytes at 16 wavefronts and is still 16 % faster there than at 8: 4.12 against 4.90 ms; the 16 x 4 Filter geometries, 170-470 bytes, stay out)
    int jit_force_waves = 0;     // DUSP_JIT_FORCE="WxR" (tests): compiled kernels with exactly W wavefronts per workgroup and R instances per wavefront,
    int jit_force_per_wave = 0;  //   whatever the batch size (renders are then not split in time)
    int ring_window = 1;         // DUSP_RING_WINDOW=0: every render zero-fills whole rings (else only what a render nothing continues can touch)
    int ring_poison = 0;         // DUSP_RING_POISON=1 (tests): rings filled with NaN patterns before the zero-fill
    int filter_scan = 1;         // DUSP_FILTER_SCAN=0: constant-cutoff Filters always through the Filter stage's serving wave (else: a scan over the chunk where its error bound,
                                 // feedback loops' gains included, allows: jit_filter_scan_ok); 2 (measurements, tests): the scan wherever its structure allows, whatever the loop gains
    int filter_warm = 1;         // DUSP_FILTER_WARM=0: ONE long circuit with Filters is not cut into segments that warm up (jit_codegen.hpp jit_warm_chunks): one wavefront walks the whole render;
                                 // n > 1 (tests): segments of n chunks whatever warm-up the Filters need
    int jit_nt = 0;              // DUSP_JIT_NT=1: compiled kernels copy PCM out with non-temporal stores; 2: only circuits with delay rings; 0: plain stores
    int jit_rotate = 1;          // DUSP_JIT_ROTATE=0 (measurements): Filter circuits compute nothing of the next chunk beside the recurrences (fewer registers an instance)
    int delay_line = 1;          // DUSP_DELAY_LINE: constant delays of a chunk at least as lines of input samples in LDS where they fit (JitDelayLine) instead of rings in memory: no ring
                                 // traffic (configs[3] 7.4 -> 6.6 ms, delay(osc, 300) 1.13 -> 0.84 ms, delay(osc, 300.5) 1.4-1.55 -> 1.15 ms).  2: only those of a whole number of samples; 0: none
    int filter_fma = 0;          // DUSP_FILTER_FMA=1 (EXPERIMENT, off): the Filter stage's recurrence as fma(-b1, y1, P - b2 y2) — three dependent instructions a step
                                 // instead of five, but ANOTHER rounding than Filter.js:40-46's (tolerance-level; compiled kernels only)
    int jit_profile = 0;         // DUSP_JIT_PROFILE=1: diagnostic kernels that stamp the cycle counter; sums printed to stderr after each render
};

#if defined(__HIPCC__)
typedef float f32x4 __attribute__((ext_vector_type(4)));  // native vector: what __builtin_nontemporal_store accepts
#endif

struct DevOperand {
    int32_t kind;  // SRC_*
    int32_t idx;   // SRC_BUF: chunk buffer; SRC_PARAM: parameter slot
    float cval;    // SRC_CONST: the f32-rounded constant
    int32_t pad;
};

// One mono operation: a (unit, output channel) pair of the circuit.
struct DevOp {
    int32_t op, unit, out_buf, state_slot;
    int32_t attr, n_in;      // Osc: table id; Filter: kind; CircleBuffer node: bit0 wipe, bit1 no-input; Pan: output channel
    DevOperand in[kMaxIn];   // n_in operands are meaningful (VectorMagnitude: one per input channel)
    double d[3];             // Ramp: duration, y0, y1; FixedMultiply: sf; SecondsToSamples: sample rate; Pan: compensation dB; Timer: period
    int64_t ring_base, ring_len;  // Delay / CircleBuffer nodes: ring location (samples, per instance)
    int32_t lds_slot, pad;        // wave engine: which block of per-wave LDS state the op owns (-1: a stateless op)
};

// Arguments of the chunk engine's kernel (chunk_engine.hip).
struct ChunkArgs {
    const DevOp *ops;
    const int32_t *out_bufs;
    float *scratch;         // [n_bufs][256][n_pad]   chunk buffers, instance-interleaved
    double *state;          // [n_slots][n_pad]
    float *rings;           // [ring_samples][n_pad]
    const float *params;    // [n_params][n_inst]
    const float *tables;    // [kNumTables][table_stride]
    const float *inputs;    // [n_inputs][n_inst][n_samples]   host-generated signals (OP_INPUT); NULL without any
    float *out;             // [n_inst][n_out][n_samples]
    uint64_t n_samples;
    int64_t clock0;
    uint32_t n_ops, n_out, n_inst, n_pad, n_chunks, sample_rate, table_stride, flags;
    // chunk k < n_warm (counted from clock 0) runs ops[warm_first[k] .. +warm_n[k]) instead of ops[0 .. n_ops)
    uint32_t n_warm, warm_first[kMaxWarmChunks], warm_n[kMaxWarmChunks];
};
constexpr uint32_t kChunkFlagResumable = 1;  // segments will follow: Delay keeps to the reference's read-modify-write ring protocol

}  // namespace dusp
