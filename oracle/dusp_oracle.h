/* TEST INFRASTRUCTURE — CPU oracle for the Dusp offline render path.
 *
 * A scalar, single-threaded restatement in plain C of the reference's
 * renderChannelData + Circuit/Unit tick loop (see dusp_oracle.c for the
 * reference file:line each function follows).  Only tests/, the smoke check in
 * __graft_entry__.py and bench.py's cpu_baseline leg may link or load this; the
 * product (dusp_amd/) never does.
 *
 * Parity pin: every golden vector under tests/golden/ was produced by the JS
 * reference itself (oracle/js/gen_golden.js); tests/test_oracle_golden.py
 * checks this restatement against all of them.
 */
#ifndef DUSP_ORACLE_H
#define DUSP_ORACLE_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dusp_oracle dusp_oracle;

/* Build the circuit for ONE instance of a descriptor.  `params` is the
 * slot-major [n_params][n_instances] f32 table (may be NULL when the
 * descriptor has no PARAM inlets).  Returns NULL and fills `err` on failure. */
dusp_oracle *dusp_oracle_create(const double *desc, size_t n_words,
                                const float *params, size_t n_instances, size_t instance,
                                char *err, size_t errlen);
void dusp_oracle_destroy(dusp_oracle *o);

/* renderChannelData(outlet, n_samples / sampleRate): ticks ceil(n_samples/chunk)
 * chunks and writes channel c to out[c * n_samples ...].  Channels beyond
 * max_channels are dropped.  Returns the channel count of the result. */
int dusp_oracle_render(dusp_oracle *o, size_t n_samples, float *out, int max_channels);

/* Unit state after rendering, in the layout of the descriptor's state words for
 * that unit's opcode.  Returns the number of words (written up to cap). */
size_t dusp_oracle_unit_state(const dusp_oracle *o, size_t unit, double *out, size_t cap);
size_t dusp_oracle_n_units(const dusp_oracle *o);

/* Wave tables of reference src/components/Osc/waveTables.js:5-40.
 * id: 0 sin, 1 saw, 2 square, 3 triangle, 4 8bit.  out has sample_rate+1 entries. */
int dusp_oracle_wavetable(int id, int sample_rate, float *out);

#ifdef __cplusplus
}
#endif
#endif
