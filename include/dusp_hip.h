/* dusp_hip.h — C ABI of the MI355X (gfx950) offline render path for Dusp.
 *
 * This library replaces ONE path of the reference: the chunk loop of
 *   src/renderChannelData.js:5-49  ->  src/Circuit.js:19-47 (tick/tickUntil)
 *   ->  src/Unit.js:111-119 (tick)  ->  per-unit `_tick` loops in src/components/
 * The reference has no FFI boundary on this path (it is pure in-process JS), so
 * the entry points below are the ones a Node N-API addon / ctypes stub binds in
 * order to stand in for `renderChannelData(outlet, duration)`; INTEGRATION.md
 * shows that binding.  Plain pointers and sizes only, no C++ or torch types.
 *
 * Hand-off format: a host-side extractor (dusp_amd/js/lib/extract.js, or
 * dusp_amd/descriptor.py) flattens the live Unit/Inlet/Outlet graph — in the
 * order `circuit.units` already holds (src/Circuit.js:125-131) — into an array of
 * little-endian f64 "descriptor words" (layout: DESIGN.md §3).  Many
 * structurally identical circuits (voices, a parameter sweep) are rendered by
 * ONE program plus a per-instance f32 parameter table.
 *
 * Error convention mirrors the reference's (thrown strings that surface as
 * Promise rejections, src/renderChannelData.js:12-17): every call returns 0 on
 * success or a negative dusp_status, and dusp_last_error() returns the message.
 * No C++ exception ever crosses this boundary and nothing calls abort(): sizes
 * taken from a descriptor are bounded before anything is allocated, and an
 * allocation failure inside the library comes back as a status like any other.
 * Nothing here ever falls back to a CPU implementation: without a usable HIP
 * device the calls fail with DUSP_ERR_HIP.
 *
 * Numerics: PCM and unit state are the reference's bit for bit wherever the device executes the IEEE operations JS
 * does — everything but the units that call Math.tan / Math.pow (Filter coefficients; Gain, DecibelToScaler,
 * SemitoneToRatio, Pow, Pan, MidiToFrequency), which are within 1e-5 of full scale.  One form trades
 * bits for speed inside that tolerance: a Filter whose cutoff is a constant between about 1.5 and 22.5 kHz (at 48 kHz) and
 * whose output only feeds sums, products with constants or bounded signals, delay lines and outlets is evaluated as a
 * scan over each chunk: each such Filter deviates from the reference's recurrence by at most 2^-24 (sum|h| + 2) <= 1.9e-6
 * of its own output's scale, and the form is taken only where what the circuit makes of those deviations — every unit's
 * worst-case gain, feedback loops' 1 / (1 - loop gain) included — stays within 2.5e-6 of the Filters' output scale at every
 * outlet (jit_codegen.hpp jit_filter_scan_ok).  A loop of gain 0.6 and above, a per-instance gain, a
 * product of two filtered signals keep the Filter stage.  DUSP_FILTER_SCAN=0 in the environment of dusp_ctx_create keeps
 * EVERY Filter on the stage, whose results are the reference recurrence's, operation for operation.
 *
 * Threading: a dusp_ctx is bound to one HIP device and is not thread-safe;
 * different contexts are independent (one context per GPU for multi-GPU use).
 * Streams: a program's workspaces are reused from render to render; renders of
 * one program are ordered among themselves whatever streams they are given (a
 * render on another stream than the previous one waits for it), and
 * dusp_state_download / dusp_program_destroy wait for the last render wherever
 * it ran.
 */
#ifndef DUSP_HIP_H
#define DUSP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DUSP_ABI_VERSION 7

typedef struct dusp_ctx dusp_ctx;
typedef struct dusp_program dusp_program;

typedef enum {
    DUSP_OK = 0,
    DUSP_ERR_ARG = -1,         /* bad argument / malformed descriptor */
    DUSP_ERR_UNSUPPORTED = -2, /* graph uses something the GPU path does not implement */
    DUSP_ERR_HIP = -3,         /* HIP runtime error (no device, out of memory, launch failure) */
    DUSP_ERR_STATE = -4,       /* call sequence error (e.g. wave table not uploaded) */
    DUSP_ERR_NOMEM = -5        /* a host allocation failed while the call was being prepared */
} dusp_status;

/* Engine that executes a program (chosen at build time from the graph's shape):
 *   CHUNK — universal engine: one lane per instance, units ticked chunk by chunk
 *           exactly in circuit order (every opcode: feedback, filters, delay lines, CircleBuffers,
 *           envelopes).
 *   FUSED — time-parallel fused kernels for recognised voice shapes (Osc, Osc x Ramp,
 *           Osc x gain, Sum.many chains): one lane per sample, time split across waves,
 *           16-byte coalesced PCM stores.
 *   WAVE  — one wavefront per instance (or a few), wavefront-wide phase accumulation.  A circuit gets a kernel COMPILED FOR IT
 *           (units inlined in process order, operands in registers, the Filters' recurrences of a workgroup side by side on
 *           one wave with the other units beside them; dusp_circuit_kernel_source) — every unit of the path: FM, Filters,
 *           feedback edges, envelopes, the comb family, delay lines and CircleBuffer nodes lane-parallel or, where their
 *           accesses can meet inside a chunk, through ordered slot operations.  What stays on the interpreter kernel
 *           (chunk buffers in LDS) is decided by regime: circuits of more
 *           than 256 units (DUSP_JIT_MAX_UNITS), and a structure's FIRST render while its kernel compiles in the background.  Few instances and
 *           a long render are split in time when the graph allows it: oscillators, closed forms of time and stateless units by exact jumps,
 *           and — ABI v7 — circuits with constant-cutoff Filters too, in segments that warm up a segment early from rest and are checked
 *           against each other on the host (bit for bit the one long recurrence).  Refuses by regime, not by unit: channel
 *           counts that grow during the first chunks, more chunk buffers than LDS holds, an oscillator phase outside [0, sampleRate).
 * AUTO picks FUSED, else WAVE, else CHUNK.  (DUSP_ENGINE_LOOP — hand-written kernels for the feedback voice Osc -> Sum -> Delay -> Filter -> gain
 * of rounds 1 and 2 — is accepted and means AUTO since ABI v7: the kernel compiled for that circuit renders it 1.7 to 2.5 times faster.)
 *
 * DUSP_ENGINE_RESUMABLE may be OR-ed into the engine argument of dusp_program_build: the program will be
 * continued with dusp_program_continue (event-segmented rendering, src/Circuit.js:23,57-65).  Programs whose
 * circuit owns delay lines / CircleBuffers or has a feedback edge then run on the WAVE or the CHUNK engine and
 * keep their rings and chunk buffers resident between segments; a continuation never fails over the engine
 * (a WAVE program whose state leaves that engine's regime migrates to CHUNK). */
typedef enum {
    DUSP_ENGINE_AUTO = 0, DUSP_ENGINE_CHUNK = 1, DUSP_ENGINE_FUSED = 2, DUSP_ENGINE_WAVE = 3, DUSP_ENGINE_LOOP = 4,
    DUSP_ENGINE_RESUMABLE = 0x100
} dusp_engine;

typedef struct {
    uint32_t sample_rate;
    uint32_t chunk_size;
    uint32_t n_units;        /* units in the circuit */
    uint32_t n_out_channels; /* channels of the rendered outlet == result.length of renderChannelData */
    uint32_t n_params;       /* per-instance parameter slots the descriptor references */
    uint32_t engine;         /* dusp_engine actually selected */
    uint32_t n_device_ops;   /* channel-expanded ops the kernel executes per chunk */
    uint32_t n_inputs;       /* host-generated input streams the program reads (dusp_render_*_inputs) */
    char shape[64];          /* FUSED: signature of the fused kernel, e.g. "mul(osc(k),ramp)" */
} dusp_program_info;

/* Library / ABI identification. */
const char *dusp_version(void);
int dusp_abi_version(void);

/* Message of the last failing call on this context (or, with ctx == NULL, of
 * the last failing dusp_ctx_create on this thread).  Never NULL. */
const char *dusp_last_error(const dusp_ctx *ctx);

/* ABI v7.  HIP devices this process sees (what `device` of dusp_ctx_create ranges over), or a negative dusp_status
 * (message: dusp_last_error(NULL)).  A host that shards independent circuit instances over a node's GPUs creates one
 * context per device (dusp_amd/js/lib/renderChannelData.js renderMany, dusp_amd/shard.py). */
int dusp_device_count(void);

/* Create a context on HIP device `device` (-1 = current device). */
int dusp_ctx_create(int device, dusp_ctx **out);
void dusp_ctx_destroy(dusp_ctx *ctx);

/* Upload one lookup table (replaces src/components/Osc/waveTables.js:5-40 and
 * src/components/Shape/shapeTables.js:3-38: the host computes the tables exactly as the
 * reference does and hands them over as data).  table_id: 0 sin, 1 saw, 2 square,
 * 3 triangle, 4 8bit (oscillators); 5 decay, 6 attack, 7 semiSine, 8 decaySquared (Shape).
 * n must equal sample_rate + 1 of every program later built on this context. */
int dusp_table_upload(dusp_ctx *ctx, int table_id, const float *table, size_t n);

/* Compile a descriptor into a device program (replaces `new Circuit(unit)` +
 * computeOrders as the thing that fixes the schedule, src/Circuit.js:67-148 —
 * the ORDER itself comes from the descriptor).  `engine` = DUSP_ENGINE_AUTO
 * normally; tests force CHUNK to cross-check the engines. */
int dusp_program_build(dusp_ctx *ctx, const double *desc, size_t n_words, int engine, dusp_program **out);
void dusp_program_destroy(dusp_program *prog);

/* Continue a rendered program from a LATER extraction of the same circuit (replaces what
 * src/Circuit.js:23,57-65 does between two ticks: host callbacks — scheduled events — have run
 * on the unit objects, changing state fields or inlet constants).  `desc` must describe the same
 * structure (units, connections, channel counts, rings) and carry the clock the previous render
 * stopped at; unit state and constants are taken from it, ring contents and (for feedback graphs)
 * the previous chunk of every outlet stay as the device left them.  The next dusp_render_* call
 * renders the next segment.  Needs a program built with DUSP_ENGINE_RESUMABLE unless the circuit is
 * feed-forward and owns no rings; the instance count of the following render must not change. */
int dusp_program_continue(dusp_program *prog, const double *desc, size_t n_words);

int dusp_program_info_get(const dusp_program *prog, dusp_program_info *info);

/* Render n_instances independent instances of the program for n_samples samples
 * each, starting from the descriptor's initial state (replaces the loop at
 * src/renderChannelData.js:29-45; like it, ticks ceil(n_samples/chunk) chunks
 * and maps NaN / -0 to +0 on copy-out).
 *
 *   d_params  device pointer, f32 [n_params][n_instances] (slot-major); may be
 *             NULL when the program has no parameters.
 *   d_out     device pointer, f32 [n_instances][n_out_channels][n_samples].
 *   stream    hipStream_t to launch on (NULL = the context's own stream).
 *
 * Asynchronous: returns once the work is enqueued.  Inputs and outputs stay
 * resident in HBM; nothing crosses PCIe.  Three kinds of program wait for the stream inside the call, because what they launch
 * depends on a few bytes the device has to hand back first: a Delay or a Filter whose delay / cutoff is a per-instance parameter
 * (the column is looked at: one small launch, its verdict back), and a few long circuits with Filters cut into segments that warm
 * up (the segments' hand-overs are checked when the launch is done).  dusp_render_host waits anyway. */
int dusp_render_device(dusp_program *prog, size_t n_instances, size_t n_samples,
                       const float *d_params, float *d_out, void *stream);

/* One `Sum.many` dealt over several GPUs, bit for bit (replaces the single process's left-deep chain of
 * src/components/Sum.js:18-29 — ((v0 + v1) + v2) + ... with an f32 rounding per add — when the voices of the mix are sharded):
 * the program holds a CONTIGUOUS run of the chain's voices (built on the fused sum chain: every voice a constant-f oscillator,
 * bare or under the supported envelopes), and this call renders the window [first_sample, first_sample + n_samples) of the
 * render's timeline, one instance, CONTINUING the chain from d_init — the running sums the ranks before this one left for the
 * same window — instead of from zero (d_init NULL: this rank holds the chain's first voices).  raw != 0 writes the sums as
 * they stand (a partial sum another rank continues: no `x || 0`, so a NaN travels on as the reference's would); the rank with
 * the chain's last voices passes raw = 0 and gets the mix as dusp_render_device would have written it.
 *   first_sample  a multiple of 2048 (whole blocks of the kernel)
 *   d_init, d_out device pointers, f32 [n_out_channels = 1][n_samples], 16-byte aligned; they may be the same buffer
 * Asynchronous like dusp_render_device.  DUSP_ERR_UNSUPPORTED for a program that is not on the fused sum chain. */
int dusp_render_chain_window(dusp_program *prog, uint64_t first_sample, size_t n_samples,
                             const float *d_init, int raw, float *d_out, void *stream);

/* Host callers (the N-API addon): uploads h_params, renders, downloads into h_out
 * (same layouts as above) and synchronises.  The download is what the caller
 * waits for — PCIe, hundreds of times slower than the render — so:
 *   h_out from dusp_host_alloc (pinned): one DMA at link speed straight into it;
 *   h_out pageable and large: several worker threads, each double-buffering pinned
 *   staging tiles on its own stream, copy disjoint ranges concurrently;
 *   small outputs: a plain asynchronous copy. */
int dusp_render_host(dusp_program *prog, size_t n_instances, size_t n_samples,
                     const float *h_params, float *h_out);

/* Result buffers (replaces `new TypedArray(lengthInSamples)` of src/renderChannelData.js:39 as the thing that owns the
 * returned samples): n_bytes of PINNED host memory from the context's pool.  dusp_render_host* into such a buffer is a
 * direct device-to-host DMA.  dusp_host_free hands the buffer back to the pool (it stays pinned for the next render of
 * that size; dusp_ctx_destroy releases everything).  The N-API addon wraps these in external ArrayBuffers whose
 * finalizer calls dusp_host_free. */
int dusp_host_alloc(dusp_ctx *ctx, size_t n_bytes, void **out);
int dusp_host_free(dusp_ctx *ctx, void *p);

/* Host-generated signals.  A descriptor may hold INPUT units (opcode 41, attribute = stream index): units whose
 * output the HOST computes while the rest of the circuit runs on the device — the reference's Noise
 * (src/components/Noise.js:16-27: a Math.random() per sample, which only the caller's JavaScript engine can draw in
 * the reference's order), or any source the caller ticks itself.  dusp_program_info.n_inputs says how many streams the
 * program reads; bind them at every render:
 *   inputs   f32 [n_inputs][n_instances][n_samples]  (the samples of THIS render call, i.e. of this segment)
 * The plain render calls refuse a program with inputs. */
int dusp_render_device_inputs(dusp_program *prog, size_t n_instances, size_t n_samples,
                              const float *d_params, const float *d_inputs, float *d_out, void *stream);
int dusp_render_host_inputs(dusp_program *prog, size_t n_instances, size_t n_samples,
                            const float *h_params, const float *h_inputs, float *h_out, int interleaved);

/* Wire format (replaces the per-chunk loop of src/RenderStream.js:36-57 as the thing that produces frames):
 * planar PCM f32 [n_instances][n_channels][n_samples], as the render calls write it, to interleaved frames
 * f32 [n_instances][n_samples][n_channels] — `buffer[t * numberOfChannels + c]`, 32-bit little-endian floats
 * (src/RenderStream.js:54,63-68; also the layout of a WAV data chunk).  Device pointers, asynchronous; the two
 * buffers must not overlap. */
int dusp_interleave_device(dusp_ctx *ctx, const float *d_planar, size_t n_instances, size_t n_channels, size_t n_samples,
                           float *d_interleaved, void *stream);

/* dusp_render_host, delivering interleaved frames: h_out is f32 [n_instances][n_samples][n_out_channels]. */
int dusp_render_host_interleaved(dusp_program *prog, size_t n_instances, size_t n_samples, const float *h_params, float *h_out);

/* State write-back (SURVEY.md §5 "checkpoint/resume"): after a render, copy the
 * state of `unit` for `instance` into out[] in the layout of the descriptor's
 * state words for that unit's opcode (Osc: phase; Ramp: t, playing; Filter:
 * has_lastF,lastF,a0,a1,a2,b1,b2,nch,(x1,x2,y1,y2)*nch; CircleBuffer nodes: t; Timer: t;
 * Shape: t,playing,finished; AHD: state,playing,t; SampleRateRedux: timeSinceLastUpdate,n,val*n).
 * Returns the number of words (>= 0) or a negative dusp_status. */
int dusp_state_download(dusp_program *prog, size_t instance, size_t unit, double *out, size_t cap);

/* The circuit compiler, on its own (needs no device).  Programs on the WAVE engine whose units it knows are rendered by ONE
 * kernel generated for that circuit — the chunk loop of src/Circuit.js:19-41 with every unit's `_tick` inlined in process
 * order, operands in registers — compiled for gfx950 in process (hiprtc) the first time a circuit structure is rendered and
 * cached afterwards (in the process and on disk).  A circuit that is a `Sum.many` of isomorphic voices gets the voice's units
 * ONCE, in a loop over the voices, from 96 units on; other circuits are straight-line code up to 256 units.  This call returns that kernel's HIP text for a descriptor (for inspection, and so that the generator
 * and the run-time compiler can be tested without a GPU):
 *   waves      wavefronts per workgroup the text is generated for (1 .. 16)
 *   per_wave   circuit instances per wavefront (1 .. 4; renders that are split in time use 1)
 *   lds_table  bit 0: assume the oscillators' first wave table is antisymmetric (half image in LDS), as a context would find;
 *              bit 1: the text of a program built with DUSP_ENGINE_RESUMABLE (a circuit with delay lines / feedback that will be
 *              continued: outlets parked between launches, rings kept in the reference's own state);
 *              bit 2: the Filter stage's recurrence loop with 4 P values per register set (what a render falls back to when the
 *              kernel spills at 8)
 *   compile    non-zero: also compile the text for gfx950
 *   text, cap  receives at most cap - 1 characters, NUL-terminated (cap 0: nothing is copied)
 * Returns the length of the text, DUSP_ERR_UNSUPPORTED when the circuit stays on the interpreter (dusp_last_error(NULL) says
 * why), or another negative dusp_status. */
int dusp_circuit_kernel_source(const double *desc, size_t n_words, int waves, int per_wave, int lds_table, int compile, char *text, size_t cap);

/* ABI v5.  Where compiled circuit kernels (code objects) are kept across processes: $DUSP_JIT_CACHE if set ("0" / "off": no
 * disk cache), else $XDG_CACHE_HOME/dusp-hip, else $HOME/.cache/dusp-hip — read once per process.  Files are keyed by the
 * kernel text, the device library's text, the compile options, the target and the hiprtc version, and carry their length and
 * a hash of their payload; a damaged file is deleted and the kernel compiled again.  Returns "" when there is no disk cache.
 * The string lives as long as the library. */
const char *dusp_jit_cache_dir(void);

/* Duration in milliseconds of the most recent render's kernel(s) on this
 * program, measured with HIP events on the launch stream (synchronises). */
int dusp_last_kernel_ms(dusp_program *prog, float *ms);

/* Fill-kernel ceiling: writes n_floats f32 to d_out with 16-byte coalesced
 * stores and nothing else — the measured HBM write roofline the render kernels
 * are compared with. */
int dusp_fill_device(dusp_ctx *ctx, float *d_out, size_t n_floats, float value, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DUSP_HIP_H */
