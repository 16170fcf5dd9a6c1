/* TEST INFRASTRUCTURE — CPU oracle (see dusp_oracle.h).  Plain C99, scalar,
 * single thread.  Build with -O2 -ffp-contract=off (no FMA contraction): JS
 * evaluates every sub-expression as a separately rounded f64 operation and
 * rounds to f32 only when storing into a Float32Array; this file does the same
 * with `double` arithmetic and explicit (float) stores.
 *
 * Every function cites the reference file:line (under /root/reference) whose
 * behaviour it restates.  The structure deliberately mirrors the reference —
 * units own multichannel 256-sample chunks, connected inlets alias the
 * producer's chunk, units tick chunk by chunk in circuit order — because the
 * oracle's job is to be obviously the same algorithm, not to be fast.
 */
#include "dusp_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAGIC 1146442576.0
#define HEADER_WORDS 12

enum { OP_OSC = 1, OP_RAMP, OP_MULTIPLY, OP_SUM, OP_FILTER, OP_DELAY, OP_CB_READER, OP_CB_WRITER, OP_REPEATER,
       /* elementwise maps */
       OP_SUBTRACT, OP_DIVIDE, OP_POLARITY_INVERT, OP_ABS, OP_CLIP, OP_HARD_CLIP_ABOVE, OP_HARD_CLIP_BELOW,
       OP_SECONDS_TO_SAMPLES, OP_FIXED_MULTIPLY, OP_GAIN, OP_DECIBEL_TO_SCALER, OP_SEMITONE_TO_RATIO, OP_POW,
       /* delay / filter family, per-channel oscillator */
       OP_FIXED_DELAY, OP_COMB_FILTER, OP_ALL_PASS, OP_MONO_DELAY, OP_READBACK_DELAY, OP_MULTI_OSC,
       /* rest of the elementwise sweep */
       OP_PAN, OP_MIDI_TO_FREQUENCY, OP_RESCALE, OP_CROSS_FADER, OP_VECTOR_MAGNITUDE, OP_TIMER, OP_SAMPLE_RATE_REDUX,
       OP_CONCAT_CHANNELS, OP_PICK_CHANNEL,
       /* envelopes */
       OP_SHAPE, OP_AHD,
       OP_HOST_ONLY, /* no signal: acts through host callbacks (Retriggerer); ticking it is a no-op */
       OP_INPUT, /* a signal the host computes (the reference's Noise draws Math.random() per sample): copied from a stream the caller hands over */
       OP_RETRIGGER /* Retriggerer whose target is a Shape / AHD of this circuit: ticked here like any unit */ };
#define N_TABLES 9 /* 0-4 oscillator wave tables, 5-8 Shape tables */
#define MAX_INLETS 5
enum { IN_CONST = 0, IN_CONNECT = 1, IN_PARAM = 2 };

/* ---- SignalChunk (reference src/SignalChunk.js:1-11): channelData[c] = Float32Array(chunkSize).
 * Channels are pointers because the reference lets two slots alias one array
 * (Delay.js:23) and lets channel lists grow at tick time (Multiply.js:29). */
typedef struct {
    int nch, cap;
    float **ch;
} chunk_t;

static float *new_channel(int n) { return (float *)calloc((size_t)n, sizeof(float)); }

static void chunk_set(chunk_t *k, int c, float *data) {
    if (c >= k->cap) {
        int cap = k->cap ? k->cap * 2 : 2;
        while (cap <= c) cap *= 2;
        k->ch = (float **)realloc(k->ch, (size_t)cap * sizeof(float *));
        for (int i = k->cap; i < cap; i++) k->ch[i] = NULL;
        k->cap = cap;
    }
    k->ch[c] = data;
    if (c >= k->nch) k->nch = c + 1;
}
static void chunk_init(chunk_t *k, int nch, int n) {
    memset(k, 0, sizeof *k);
    for (int c = 0; c < nch; c++) chunk_set(k, c, new_channel(n));
}
/* `x[c] = x[c] || new Float32Array(n)` */
static float *chunk_ensure(chunk_t *k, int c, int n) {
    if (c >= k->nch || !k->ch[c]) chunk_set(k, c, new_channel(n));
    return k->ch[c];
}

/* ---- CircleBuffer (reference src/CircleBuffer.js:3-35) */
typedef struct {
    int nch;
    long len;
    float **data;
} ring_t;

static float ring_read(const ring_t *r, int c, double t) { /* CircleBuffer.js:15-20 */
    t = floor(fmod(t, (double)r->len));
    while (t < 0) t += (double)r->len;
    if (!(t >= 0 && t < (double)r->len)) return NAN; /* typed array [NaN] -> undefined -> NaN on store */
    return r->data[c][(long)t];
}
static void ring_write(ring_t *r, int c, double t, float y) { /* CircleBuffer.js:21-27 */
    t = floor(fmod(t, (double)r->len));
    while (t < 0) t += (double)r->len;
    if (!(t >= 0 && t < (double)r->len)) return;
    r->data[c][(long)t] = y;
}
static void ring_mix(ring_t *r, int c, double t, float y) { /* CircleBuffer.js:28-34 */
    t = floor(fmod(t, (double)r->len));
    while (t < 0) t += (double)r->len;
    if (!(t >= 0 && t < (double)r->len)) return;
    r->data[c][(long)t] = (float)((double)r->data[c][(long)t] + (double)y);
}

/* ---- Unit / Inlet / Outlet (reference src/Unit.js, Inlet.js, Outlet.js, Piglet.js) */
typedef struct {
    int connected, src;
    chunk_t own; /* unconnected: constant-filled chunk (Inlet.js:76-93) */
} inlet_t;

typedef struct unit {
    int op, n_inlets;
    inlet_t in[MAX_INLETS];
    chunk_t out;
    /* Osc */
    int waveform;
    double phase;
    /* Ramp */
    double duration, ry0, ry1, t;
    int playing;
    /* Filter */
    int kind, has_lastF, nstate;
    double lastF, a0, a1, a2, b1, b2;
    double *x1, *x2, *y1, *y2; /* NAN marks "undefined" array slots */
    /* Delay */
    long maxDelay;
    int nbuffers;
    float **buffers;
    /* CircleBuffer nodes */
    int ring, wipe;
    double cb_t;
    /* FixedMultiply */
    double sf;
    /* FixedDelay / CombFilter / AllPass / ReadBackDelay: ring length and write head; MultiChannelOsc: phases */
    long fd_len;
    double tBuffer;
    int nphase;
    double *phases;
    /* Shape: table, edges, t (shares `t`, `playing` with Ramp), finished; AHD: stage, t, playing, samplePeriod */
    int shape_table, left_is_shape, right_is_shape, finished, ahd_state;
    double left_edge, right_edge;
    /* Pan: compensationDB; Timer: t, samplePeriod; SampleRateRedux: val[], timeSinceLastUpdate */
    double comp_db, timer_t, sample_period, tslu;
    int nval;
    float *val;
} unit_t;

struct dusp_oracle {
    int sr, chunk;
    size_t n_units, n_rings;
    unit_t *units;
    ring_t *rings;
    size_t out_unit;
    long clock;
    float *tables[N_TABLES];
    /* host-generated input streams [n][input_len], bound by dusp_oracle_set_inputs at clock input_clock0 */
    const float *inputs;
    size_t input_len;
    long input_clock0;
};

/* ---- wave tables (reference src/components/Osc/waveTables.js:5-40) */
static double js_round(double x) { /* Math.round: ties toward +inf, keeps -0 */
    double r = floor(x + 0.5);
    if (r == 0 && (x < 0 || signbit(x))) r = -0.0;
    return r;
}
int dusp_oracle_wavetable(int id, int sr, float *out) {
    const double PHI = 2 * 3.141592653589793; /* waveTables.js:3 */
    const int n = sr + 1;
    if (id >= 5 && id <= 8) { /* Shape tables: func(x / sampleRate), x = 0..sampleRate (Shape/shapeTables.js:3-38) */
        for (int x = 0; x < n; x++) {
            const double v = (double)x / sr;
            out[x] = (float)(id == 5 ? 1 - v : id == 6 ? v : id == 7 ? sin(3.141592653589793 * v) : (1 - v) * (1 - v));
        }
        return 0;
    }
    switch (id) {
    case 0: /* sine :5-8 — note the period is the table LENGTH sr+1 */
    case 4: /* 8bit :28-31 is derived from the f32 sine table */
        for (int t = 0; t < n; t++) out[t] = (float)sin(PHI * t / n);
        if (id == 4)
            for (int t = 0; t < n; t++) out[t] = (float)(js_round((double)out[t] * 128.0) / 128.0);
        return 0;
    case 1: /* saw :10-12 — loop stops at sr, so the last entry stays 0 */
        for (int t = 0; t < sr; t++) out[t] = (float)(-1 + t * 2.0 / n);
        out[sr] = 0;
        return 0;
    case 2: /* square :24-26 */
        if (sr % 2) return -1;
        for (int t = 0; t < n; t++) out[t] = t < sr / 2 ? 1.f : -1.f;
        return 0;
    case 3: { /* triangle :14-22 — later quarters read the f32-rounded first quarter back */
        if (sr % 4) return -1;
        const int q = sr / 4;
        memset(out, 0, (size_t)n * sizeof(float));
        for (int t = 0; t < q; t++) {
            out[t] = (float)((double)t / sr * 4);
            out[t + q] = (float)(1 - (double)out[t]);
            out[t + 2 * q] = (float)(-(double)out[t]);
            out[t + 3 * q] = (float)(-1 + (double)out[t]);
        }
        out[sr] = 0;
        return 0;
    }
    }
    return -1;
}

/* `x || 0` on a number: NaN and -0 become +0 */
static double or0(double v) { return (v != v || v == 0) ? 0.0 : v; }

/* ---- per-unit _tick restatements ---------------------------------------- */

static chunk_t *inlet_chunk(dusp_oracle *o, unit_t *u, int i) {
    return u->in[i].connected ? &o->units[u->in[i].src].out : &u->in[i].own; /* Inlet.js:52-55 */
}

/* reference src/components/Osc/Osc.js:35-47 */
static void tick_osc(dusp_oracle *o, unit_t *u) {
    const float *f = inlet_chunk(o, u, 0)->ch[0]; /* mono inlet: channel 0 (Piglet.js:56-61) */
    const float *tbl = o->tables[u->waveform];
    float *out = u->out.ch[0];
    const double sr = o->sr;
    for (int t = 0; t < o->chunk; t++) {
        u->phase += (double)f[t];
        u->phase = fmod(u->phase, sr);
        if (u->phase < 0) u->phase += sr;
        double fraction = fmod(u->phase, 1.0);
        double lo = floor(u->phase), hi = ceil(u->phase);
        double a = (lo >= 0 && lo <= sr) ? (double)tbl[(long)lo] : NAN; /* OOB / NaN index -> undefined */
        double b = (hi >= 0 && hi <= sr) ? (double)tbl[(long)hi] : NAN;
        out[t] = (float)(a * (1 - fraction) + b * fraction);
    }
}

/* reference src/components/Ramp.js:25-40 */
static void tick_ramp(dusp_oracle *o, unit_t *u) {
    float *out = u->out.ch[0];
    for (int i = 0; i < o->chunk; i++) {
        if (u->playing) {
            u->t++;
            if (u->t > u->duration) { u->playing = 0; u->t = u->duration; }
            if (u->t < 0) { u->playing = 0; u->t = 0; }
        }
        out[i] = (float)(u->ry0 + (u->t / u->duration) * (u->ry1 - u->ry0));
    }
}

/* reference src/components/Multiply.js:23-34 and Sum.js:33-44 */
static void tick_combine(dusp_oracle *o, unit_t *u, int is_sum) {
    chunk_t *a = inlet_chunk(o, u, 0), *b = inlet_chunk(o, u, 1);
    for (int c = 0; c < a->nch || c < b->nch; c++) {
        const float *ac = a->ch[c % a->nch], *bc = b->ch[c % b->nch];
        float *oc = chunk_ensure(&u->out, c, o->chunk);
        for (int t = 0; t < o->chunk; t++)
            oc[t] = is_sum ? (float)((double)ac[t] + (double)bc[t]) : (float)((double)ac[t] * (double)bc[t]);
    }
}

/* reference src/components/Filter.js:66-84 (LP, HP; BP/BR need a bandwidth the tick never passes) */
static void filter_coefficients(dusp_oracle *o, unit_t *u, double f) {
    const double PI = 3.141592653589793;
    if (u->kind == 0) {
        double lamda = 1 / tan(PI * f / o->sr);
        double lamdaSquared = lamda * lamda;
        u->a0 = 1 / (1 + 2 * lamda + lamdaSquared);
        u->a1 = 2 * u->a0;
        u->a2 = u->a0;
        u->b1 = 2 * u->a0 * (1 - lamdaSquared);
        u->b2 = u->a0 * (1 - 2 * lamda + lamdaSquared);
    } else {
        double lamda = tan(PI * f / o->sr);
        double lamdaSquared = lamda * lamda;
        u->a0 = 1 / (1 + 2 * lamda + lamdaSquared);
        u->a1 = 0;
        u->a2 = -u->a0;
        u->b1 = 2 * u->a0 * (lamdaSquared - 1);
        u->b2 = u->a0 * (1 - 2 * lamda + lamdaSquared);
    }
}
static void filter_grow(unit_t *u, int nch) {
    if (nch <= u->nstate) return;
    u->x1 = (double *)realloc(u->x1, (size_t)nch * sizeof(double));
    u->x2 = (double *)realloc(u->x2, (size_t)nch * sizeof(double));
    u->y1 = (double *)realloc(u->y1, (size_t)nch * sizeof(double));
    u->y2 = (double *)realloc(u->y2, (size_t)nch * sizeof(double));
    for (int c = u->nstate; c < nch; c++) u->x1[c] = u->x2[c] = u->y1[c] = u->y2[c] = NAN; /* undefined */
    u->nstate = nch;
}
/* reference src/components/Filter.js:27-51 */
static void tick_filter(dusp_oracle *o, unit_t *u) {
    chunk_t *in = inlet_chunk(o, u, 0);
    const float *f = inlet_chunk(o, u, 1)->ch[0];
    const int nch = in->nch;
    while (u->out.nch < in->nch) chunk_set(&u->out, u->out.nch, new_channel(o->chunk));
    filter_grow(u, nch);
    for (int t = 0; t < o->chunk; t++) {
        if (!u->has_lastF || (double)f[t] != u->lastF) { /* `f[t] != undefined` is true; NaN != NaN is true */
            u->has_lastF = 1;
            u->lastF = f[t];
            filter_coefficients(o, u, f[t]);
        }
        for (int c = 0; c < nch; c++) {
            double x = in->ch[c][t];
            float y = (float)(u->a0 * x + u->a1 * or0(u->x1[c]) + u->a2 * or0(u->x2[c])
                              - u->b1 * or0(u->y1[c]) - u->b2 * or0(u->y2[c]));
            u->out.ch[c][t] = y;
            u->y2[c] = or0(u->y1[c]);
            u->y1[c] = y; /* reads the f32-rounded sample back (:46) */
            u->x2[c] = or0(u->x1[c]);
            u->x1[c] = x;
        }
    }
}

/* reference src/components/Delay.js:20-41 */
static void tick_delay(dusp_oracle *o, unit_t *u, long clock) {
    chunk_t *in = inlet_chunk(o, u, 0), *dl = inlet_chunk(o, u, 1);
    const double len = (double)u->maxDelay;
    for (int c = 0; c < in->nch || c < dl->nch; c++) {
        float *out = chunk_ensure(&u->out, c, o->chunk);
        chunk_set(in, c, in->ch[c % in->nch]); /* `this.in[c] = this.in[c%this.in.length]` aliases (:23) */
        if (c >= u->nbuffers) {
            u->buffers = (float **)realloc(u->buffers, (size_t)(c + 1) * sizeof(float *));
            for (int i = u->nbuffers; i <= c; i++) u->buffers[i] = (float *)calloc((size_t)u->maxDelay, sizeof(float));
            u->nbuffers = c + 1;
        }
        float *buf = u->buffers[c];
        const float *delay = dl->ch[c % dl->nch];
        const float *x = in->ch[c];
        for (int t = 0; t < o->chunk; t++) {
            long tBuffer = (clock + t) % u->maxDelay;
            out[t] = buf[tBuffer];
            buf[tBuffer] = 0;
            double tWrite = fmod((double)tBuffer + (double)delay[t], len);
            double lo = floor(tWrite), hi = ceil(tWrite), frac = fmod(tWrite, 1.0);
            /* out-of-range / NaN indices: typed-array writes are silently dropped (no wrap at hi == len) */
            if (lo >= 0 && lo < len) buf[(long)lo] = (float)((double)buf[(long)lo] + (double)x[t] * (1 - frac));
            if (hi >= 0 && hi < len) buf[(long)hi] = (float)((double)buf[(long)hi] + (double)x[t] * frac);
        }
    }
}

/* reference src/components/CircleBufferReader.js:12-25 */
static void tick_cb_reader(dusp_oracle *o, unit_t *u) {
    ring_t *r = &o->rings[u->ring];
    chunk_t *off = inlet_chunk(o, u, 0);
    for (int c = 0; c < r->nch; c++) {
        const float *offset = off->ch[c % off->nch];
        for (int t = 0; t < o->chunk; t++) {
            double tRead = u->cb_t + t - (double)o->sr * (double)offset[t];
            u->out.ch[c][t] = ring_read(r, c, tRead);
            if (u->wipe) ring_write(r, c, tRead, 0);
        }
    }
    u->cb_t += o->chunk;
}
/* reference src/components/CircleBufferWriter.js:12-25 */
static void tick_cb_writer(dusp_oracle *o, unit_t *u) {
    ring_t *r = &o->rings[u->ring];
    chunk_t *off = inlet_chunk(o, u, 0), *in = inlet_chunk(o, u, 1);
    for (int c = 0; c < r->nch; c++) {
        const float *offset = off->ch[c % off->nch];
        for (int t = 0; t < o->chunk; t++) {
            double tWrite = u->cb_t + t + (double)o->sr * (double)offset[t];
            if (u->wipe) ring_write(r, c, tWrite, 0);
            if (c < in->nch && in->ch[c]) ring_mix(r, c, tWrite, in->ch[c][t]); /* `if(this.in[c])`: no modulo */
        }
    }
    u->cb_t += o->chunk;
}
/* reference src/components/Repeater.js:23-30 */
static void tick_repeater(dusp_oracle *o, unit_t *u) {
    chunk_t *in = inlet_chunk(o, u, 0);
    for (int c = 0; c < in->nch; c++) {
        float *oc = chunk_ensure(&u->out, c, o->chunk);
        memcpy(oc, in->ch[c], (size_t)o->chunk * sizeof(float));
    }
}

/* reference src/components/FixedDelay.js:13-19, CombFilter.js:11-17, AllPass.js:8-15 (mono in, mono out) */
static void tick_fixed_delay(dusp_oracle *o, unit_t *u) {
    const float *in = inlet_chunk(o, u, 0)->ch[0];
    const float *fb = u->n_inlets > 1 ? inlet_chunk(o, u, 1)->ch[0] : NULL;
    float *out = u->out.ch[0], *buf = u->buffers[0];
    for (int t = 0; t < o->chunk; t++) {
        u->tBuffer = fmod(u->tBuffer + 1, (double)u->fd_len);
        const long tb = (long)u->tBuffer;
        if (u->op == OP_FIXED_DELAY) {
            out[t] = buf[tb];
            buf[tb] = in[t];
        } else if (u->op == OP_COMB_FILTER) {
            out[t] = buf[tb];
            buf[tb] = (float)((double)in[t] + (double)out[t] * (double)fb[t]);
        } else {
            const double delayOut = buf[tb];
            buf[tb] = (float)((double)in[t] + delayOut * (double)fb[t]);
            out[t] = (float)(delayOut - (double)in[t] * (double)fb[t]);
        }
    }
}

/* reference src/components/MonoDelay.js:16-30: writes first (the ceil tap WRAPS here, unlike Delay.js), then reads */
static void tick_mono_delay(dusp_oracle *o, unit_t *u, long clock) {
    const float *in = inlet_chunk(o, u, 0)->ch[0], *delay = inlet_chunk(o, u, 1)->ch[0];
    float *out = u->out.ch[0], *buf = u->buffers[0];
    const double len = (double)u->maxDelay;
    for (int t = 0; t < o->chunk; t++) {
        const long tBuffer = (clock + t) % u->maxDelay;
        const double tWrite = fmod((double)tBuffer + (double)delay[t], len);
        const double lo = floor(tWrite), hi = fmod(ceil(tWrite), len), frac = fmod(tWrite, 1.0);
        if (lo >= 0 && lo < len) buf[(long)lo] = (float)((double)buf[(long)lo] + (double)in[t] * (1 - frac));
        if (hi >= 0 && hi < len) buf[(long)hi] = (float)((double)buf[(long)hi] + (double)in[t] * frac); /* -0 indexes slot 0 */
        out[t] = buf[tBuffer];
        buf[tBuffer] = 0;
    }
}

/* reference src/components/ReadBackDelay.js:24-44 (`delay > bufferLength` throws there; here it reads NaN) */
static void tick_readback_delay(dusp_oracle *o, unit_t *u) {
    chunk_t *in = inlet_chunk(o, u, 0), *dl = inlet_chunk(o, u, 1);
    const double len = (double)u->fd_len;
    const double t0 = u->tBuffer;
    for (int c = 0; c < in->nch || c < dl->nch; c++) {
        const float *input = in->ch[c % in->nch], *delay = dl->ch[c % dl->nch];
        float *output = chunk_ensure(&u->out, c, o->chunk);
        if (c >= u->nbuffers) {
            u->buffers = (float **)realloc(u->buffers, (size_t)(c + 1) * sizeof(float *));
            for (int i = u->nbuffers; i <= c; i++) u->buffers[i] = (float *)calloc((size_t)u->fd_len, sizeof(float));
            u->nbuffers = c + 1;
        }
        float *buf = u->buffers[c];
        for (int i = 0; i < o->chunk; i++) {
            const double t = t0 + i;
            buf[(long)fmod(t + len, len)] = input[i];
            const double r = fmod(t - (double)delay[i] + len, len);
            output[i] = (r >= 0 && r < len && r == floor(r)) ? buf[(long)r] : NAN; /* fractional / negative index: undefined */
        }
    }
    u->tBuffer = t0 + o->chunk;
}

/* reference src/components/Osc/MultiChannelOsc.js:21-38: one phase per channel of f, and NO `phase < 0` fix-up */
static void tick_multi_osc(dusp_oracle *o, unit_t *u) {
    chunk_t *fch = inlet_chunk(o, u, 0);
    const float *tbl = o->tables[u->waveform];
    const double sr = o->sr;
    for (int c = 0; c < fch->nch; c++) {
        if (c >= u->nphase) {
            u->phases = (double *)realloc(u->phases, (size_t)(c + 1) * sizeof(double));
            for (int i = u->nphase; i <= c; i++) u->phases[i] = 0;
            u->nphase = c + 1;
        }
        float *out = chunk_ensure(&u->out, c, o->chunk);
        const float *f = fch->ch[c];
        double phase = or0(u->phases[c]); /* `this.phase[c] = this.phase[c] || 0` */
        for (int t = 0; t < o->chunk; t++) {
            phase += (double)f[t];
            phase = fmod(phase, sr);
            const double fraction = fmod(phase, 1.0);
            const double lo = floor(phase), hi = ceil(phase);
            const double a = (lo >= 0 && lo <= sr) ? (double)tbl[(long)lo] : NAN;
            const double b = (hi >= 0 && hi <= sr) ? (double)tbl[(long)hi] : NAN;
            out[t] = (float)(a * (1 - fraction) + b * fraction);
        }
        u->phases[c] = phase;
    }
}

/* Math.pow: like C pow except pow(+-1, +-Inf) and pow(x, NaN) are NaN (ECMA-262 Number::exponentiate) */
static double js_pow(double x, double y) {
    if (y != y) return NAN;
    if ((x == 1 || x == -1) && isinf(y)) return NAN;
    return pow(x, y);
}

/* elementwise units: reference src/components/{Subtract,Divide,Pow,PolarityInvert,Abs,DecibelToScaler,
 * SemitoneToRatio,SecondsToSamples,FixedMultiply,Clip,HardClipAbove,HardClipBelow,Gain}.js `_tick` */
static void tick_map(dusp_oracle *o, unit_t *u) {
    chunk_t *a = inlet_chunk(o, u, 0);
    chunk_t *b = u->n_inlets > 1 ? inlet_chunk(o, u, 1) : NULL;
    static float zero[4096];
    const int n = o->chunk;
    switch (u->op) {
    case OP_SUBTRACT: /* `this.a[c] || zeroChunk`: a missing channel is silence, NOT a modulo broadcast (Subtract.js:20-21) */
        for (int c = 0; c < a->nch || c < b->nch; c++) {
            float *oc = chunk_ensure(&u->out, c, n);
            const float *ac = c < a->nch ? a->ch[c] : zero, *bc = c < b->nch ? b->ch[c] : zero;
            for (int t = 0; t < n; t++) oc[t] = (float)((double)ac[t] - (double)bc[t]);
        }
        return;
    case OP_DIVIDE: case OP_POW: /* Divide.js:13-23, Pow.js:19-30 */
        for (int c = 0; c < a->nch || c < b->nch; c++) {
            const float *ac = a->ch[c % a->nch], *bc = b->ch[c % b->nch];
            float *oc = chunk_ensure(&u->out, c, n);
            for (int t = 0; t < n; t++)
                oc[t] = u->op == OP_DIVIDE ? (float)((double)ac[t] / (double)bc[t]) : (float)js_pow(ac[t], bc[t]);
        }
        return;
    case OP_FIXED_MULTIPLY: /* mono in / mono out (FixedMultiply.js:18-21) */
        for (int t = 0; t < n; t++) u->out.ch[0][t] = (float)((double)a->ch[0][t] * u->sf);
        return;
    default: break;
    }
    for (int c = 0; c < a->nch; c++) {
        const float *x = a->ch[c];
        const float *y = b ? (u->op == OP_GAIN ? b->ch[0] : b->ch[c % b->nch]) : NULL; /* gain is a mono inlet (Gain.js:6) */
        float *oc = chunk_ensure(&u->out, c, n);
        for (int t = 0; t < n; t++) {
            switch (u->op) {
            case OP_POLARITY_INVERT: oc[t] = -x[t]; break;                                   /* PolarityInvert.js:13-16 */
            case OP_ABS: oc[t] = (float)fabs((double)x[t]); break;                           /* Abs.js:17-20 */
            case OP_DECIBEL_TO_SCALER: oc[t] = (float)js_pow(10, (double)x[t] / 20); break;  /* DecibelToScaler.js:15-16 */
            case OP_SEMITONE_TO_RATIO: oc[t] = (float)js_pow(2, (double)x[t] / 12); break;   /* SemitoneToRatio.js:15-16 */
            case OP_SECONDS_TO_SAMPLES: oc[t] = (float)((double)x[t] * o->sr); break;        /* SecondsToSamples.js:18-19 */
            case OP_CLIP: oc[t] = fabs((double)x[t]) > fabs((double)y[t]) ? y[t] : x[t]; break; /* Clip.js:19-21 */
            case OP_HARD_CLIP_ABOVE: oc[t] = x[t] > y[t] ? y[t] : x[t]; break;               /* HardClipAbove.js:18-22 */
            case OP_HARD_CLIP_BELOW: oc[t] = x[t] < y[t] ? y[t] : x[t]; break;               /* HardClipBelow.js:18-22 */
            case OP_GAIN: oc[t] = (float)(js_pow(10, (double)y[t] / 20) * (double)x[t]); break; /* Gain.js:20-21,24-26 */
            }
        }
    }
}


/* reference src/components/Pan.js:19-29 — mono in / pan, two output channels */
static void tick_pan(dusp_oracle *o, unit_t *u) {
    const float *in = inlet_chunk(o, u, 0)->ch[0], *pan = inlet_chunk(o, u, 1)->ch[0];
    for (int t = 0; t < o->chunk; t++) {
        const double compensation = js_pow(10, ((1 - fabs((double)pan[t])) * u->comp_db) / 20);
        u->out.ch[0][t] = (float)((double)in[t] * (1 - (double)pan[t]) / 2 * compensation);
        u->out.ch[1][t] = (float)((double)in[t] * (1 + (double)pan[t]) / 2 * compensation);
    }
}

/* reference src/components/MidiToFrequency.js:15-22: `this.frequency[c] || new Float32Array(..)` never stores the new
 * array, so only channel 0 of the outlet ever exists; the other midi channels are computed into garbage. */
static void tick_midi_to_frequency(dusp_oracle *o, unit_t *u) {
    const float *m = inlet_chunk(o, u, 0)->ch[0];
    for (int t = 0; t < o->chunk; t++) u->out.ch[0][t] = (float)(js_pow(2, ((double)m[t] - 69) / 12) * 440);
}

/* reference src/components/Rescale.js:25-38 */
static void tick_rescale(dusp_oracle *o, unit_t *u) {
    chunk_t *in = inlet_chunk(o, u, 0), *il = inlet_chunk(o, u, 1), *iu = inlet_chunk(o, u, 2), *ol = inlet_chunk(o, u, 3),
            *ou = inlet_chunk(o, u, 4);
    for (int c = 0; c < in->nch; c++) {
        float *oc = chunk_ensure(&u->out, c, o->chunk);
        const float *x = in->ch[c], *a = il->ch[c % il->nch], *b = iu->ch[c % iu->nch], *p = ol->ch[c % ol->nch],
                    *q = ou->ch[c % ou->nch];
        for (int t = 0; t < o->chunk; t++)
            oc[t] = (float)(((double)x[t] - (double)a[t]) / ((double)b[t] - (double)a[t]) * ((double)q[t] - (double)p[t]) + (double)p[t]);
    }
}

/* reference src/components/CrossFader.js:21-30: a missing channel of a / b is silence */
static void tick_cross_fader(dusp_oracle *o, unit_t *u) {
    static float zero[4096];
    chunk_t *a = inlet_chunk(o, u, 0), *b = inlet_chunk(o, u, 1);
    const float *dial = inlet_chunk(o, u, 2)->ch[0];
    for (int c = 0; c < a->nch || c < b->nch; c++) {
        const float *ac = c < a->nch ? a->ch[c] : zero, *bc = c < b->nch ? b->ch[c] : zero;
        float *oc = chunk_ensure(&u->out, c, o->chunk);
        for (int t = 0; t < o->chunk; t++) oc[t] = (float)((1 - (double)dial[t]) * (double)ac[t] + (double)dial[t] * (double)bc[t]);
    }
}

/* reference src/components/vector/VectorMagnitude.js:17-29 */
static void tick_vector_magnitude(dusp_oracle *o, unit_t *u) {
    chunk_t *in = inlet_chunk(o, u, 0);
    for (int t = 0; t < o->chunk; t++) {
        double squareSum = 0;
        for (int c = 0; c < in->nch; c++) {
            const double x = in->ch[c][t];
            squareSum += x * x;
        }
        u->out.ch[0][t] = (float)sqrt(squareSum);
    }
}

/* reference src/components/Retriggerer.js:13-24; trigger() of the target: Shape/index.js:107-111 (playing, t = 0),
 * AHD.js:24-28 (state = 1, playing) */
static void tick_retrigger(dusp_oracle *o, unit_t *u) {
    const float *rate = inlet_chunk(o, u, 0)->ch[0];
    unit_t *target = &o->units[u->ring]; /* (the target's index, kept in the field CircleBuffer nodes use for their ring) */
    for (int t = 0; t < o->chunk; t++) {
        u->timer_t += (double)rate[t];
        if (u->timer_t >= o->sr) {
            if (target->op == OP_SHAPE || target->op == OP_RAMP) { target->playing = 1; target->t = 0; } /* Ramp.js:19-23 */
            else if (target->op == OP_AHD) { target->ahd_state = 1; target->playing = 1; }
            u->timer_t -= o->sr;
        }
    }
}

/* reference src/components/Timer.js:36-41 */
static void tick_timer(dusp_oracle *o, unit_t *u) {
    for (int t = 0; t < o->chunk; t++) {
        u->timer_t += u->sample_period;
        u->out.ch[0][t] = (float)u->timer_t;
    }
}

/* reference src/components/SampleRateRedux.js:21-38: `val` has one entry before the first update, then one per input
 * channel; output channels beyond val.length are not written that sample */
static void tick_sample_rate_redux(dusp_oracle *o, unit_t *u) {
    chunk_t *in = inlet_chunk(o, u, 0);
    const float *amount = inlet_chunk(o, u, 1)->ch[0];
    while (u->out.nch < in->nch) chunk_ensure(&u->out, u->out.nch, o->chunk);
    for (int t = 0; t < o->chunk; t++) {
        u->tslu += 1;
        if (u->tslu > (double)amount[t]) {
            if (u->nval < in->nch) u->val = (float *)realloc(u->val, (size_t)in->nch * sizeof(float));
            u->nval = in->nch;
            for (int c = 0; c < in->nch; c++) u->val[c] = in->ch[c][t];
            u->tslu = 0;
        }
        for (int c = 0; c < u->nval && c < u->out.nch; c++) u->out.ch[c][t] = u->val[c];
    }
}

/* reference src/components/ConcatChannels.js:17-31 */
static void tick_concat_channels(dusp_oracle *o, unit_t *u) {
    chunk_t *a = inlet_chunk(o, u, 0), *b = inlet_chunk(o, u, 1);
    for (int c = 0; c < a->nch + b->nch; c++) {
        float *oc = chunk_ensure(&u->out, c, o->chunk);
        memcpy(oc, c < a->nch ? a->ch[c] : b->ch[c - a->nch], (size_t)o->chunk * sizeof(float));
    }
}

/* reference src/components/PickChannel.js:17-22: `this.in[this.c[t] % this.in.length][t]`; an index that is not one
 * of 0..n-1 (fractional, negative, NaN) makes the reference throw a TypeError mid-render — NaN here */
static void tick_pick_channel(dusp_oracle *o, unit_t *u) {
    chunk_t *in = inlet_chunk(o, u, 0);
    const float *c = inlet_chunk(o, u, 1)->ch[0];
    for (int t = 0; t < o->chunk; t++) {
        const double k = fmod((double)c[t], (double)in->nch);
        u->out.ch[0][t] = (k >= 0 && k < in->nch && k == floor(k)) ? in->ch[(int)k][t] : NAN;
    }
}

/* reference src/components/Shape/index.js:28-59 */
static void tick_shape(dusp_oracle *o, unit_t *u) {
    const float *duration = inlet_chunk(o, u, 0)->ch[0], *mn = inlet_chunk(o, u, 1)->ch[0], *mx = inlet_chunk(o, u, 2)->ch[0];
    const float *data = o->tables[u->shape_table];
    const double sr = o->sr;
    float *out = u->out.ch[0];
    for (int t = 0; t < o->chunk; t++) {
        const double lo = mn[t], hi = mx[t];
        if (u->playing) u->t += 1 / (double)duration[t];
        if (u->t <= 0) {
            out[t] = (float)((u->left_is_shape ? (double)data[0] : u->left_edge) * (hi - lo) + lo);
        } else if (u->t > sr) {
            u->finished = 1; /* finish(): UnitOrPatch.js:77-84; hooks are refused by the extractor */
            out[t] = (float)((u->right_is_shape ? (double)data[o->sr] : u->right_edge) * (hi - lo) + lo);
        } else if (u->t == u->t) {
            const double frac = fmod(u->t, 1);
            out[t] = (float)(lo + (hi - lo) * ((double)data[(long)ceil(u->t)] * frac + (double)data[(long)floor(u->t)] * (1 - frac)));
        } else
            out[t] = NAN; /* t is NaN (duration NaN): table[NaN] is undefined */
    }
}

/* reference src/components/AHD.js:35-76 */
static void tick_ahd(dusp_oracle *o, unit_t *u) {
    const float *attack = inlet_chunk(o, u, 0)->ch[0], *hold = inlet_chunk(o, u, 1)->ch[0], *decay = inlet_chunk(o, u, 2)->ch[0];
    float *out = u->out.ch[0];
    for (int t = 0; t < o->chunk; t++) {
        switch (u->ahd_state) {
        case 1:
            out[t] = (float)u->t;
            if (u->playing) {
                u->t += u->sample_period / (double)attack[t];
                if (u->t >= 1) { u->ahd_state++; u->t--; }
            }
            break;
        case 2:
            out[t] = 1;
            if (u->playing) {
                u->t += u->sample_period / (double)hold[t];
                if (u->t >= 1) { u->ahd_state++; u->t--; }
            }
            break;
        case 3:
            out[t] = (float)(1 - u->t);
            if (u->playing) {
                u->t += u->sample_period / (double)decay[t];
                if (u->t >= 1) { u->ahd_state = 0; u->playing = 0; } /* stop() */
            }
            break;
        case 0: out[t] = 0; break;
        default: break; /* any other `state`: the sample keeps its previous value */
        }
    }
}

/* reference src/Circuit.js:19-41 (tick) with src/Unit.js:111-119; every unit's
 * tickInterval equals the chunk size here, so gcdTickInterval == chunk. */
static void circuit_tick(dusp_oracle *o) {
    for (size_t i = 0; i < o->n_units; i++) {
        unit_t *u = &o->units[i];
        switch (u->op) {
        case OP_OSC: tick_osc(o, u); break;
        case OP_RAMP: tick_ramp(o, u); break;
        case OP_MULTIPLY: tick_combine(o, u, 0); break;
        case OP_SUM: tick_combine(o, u, 1); break;
        case OP_FILTER: tick_filter(o, u); break;
        case OP_DELAY: tick_delay(o, u, o->clock); break;
        case OP_CB_READER: tick_cb_reader(o, u); break;
        case OP_CB_WRITER: tick_cb_writer(o, u); break;
        case OP_REPEATER: tick_repeater(o, u); break;
        case OP_FIXED_DELAY: case OP_COMB_FILTER: case OP_ALL_PASS: tick_fixed_delay(o, u); break;
        case OP_MONO_DELAY: tick_mono_delay(o, u, o->clock); break;
        case OP_READBACK_DELAY: tick_readback_delay(o, u); break;
        case OP_MULTI_OSC: tick_multi_osc(o, u); break;
        case OP_PAN: tick_pan(o, u); break;
        case OP_MIDI_TO_FREQUENCY: tick_midi_to_frequency(o, u); break;
        case OP_RESCALE: tick_rescale(o, u); break;
        case OP_CROSS_FADER: tick_cross_fader(o, u); break;
        case OP_VECTOR_MAGNITUDE: tick_vector_magnitude(o, u); break;
        case OP_TIMER: tick_timer(o, u); break;
        case OP_SAMPLE_RATE_REDUX: tick_sample_rate_redux(o, u); break;
        case OP_CONCAT_CHANNELS: tick_concat_channels(o, u); break;
        case OP_PICK_CHANNEL: tick_pick_channel(o, u); break;
        case OP_HOST_ONLY: break;
        case OP_RETRIGGER: tick_retrigger(o, u); break;
        case OP_INPUT: { /* the chunk the host computed for this unit (zeros past the end of what it handed over) */
            for (int t = 0; t < o->chunk; t++) {
                const long at = o->clock - o->input_clock0 + t;
                u->out.ch[0][t] = (o->inputs && at >= 0 && (size_t)at < o->input_len) ? o->inputs[(size_t)u->ring * o->input_len + (size_t)at] : 0.f;
            }
            break;
        }
        case OP_SHAPE: tick_shape(o, u); break;
        case OP_AHD: tick_ahd(o, u); break;
        default: tick_map(o, u); break;
        }
    }
    o->clock += o->chunk;
}

/* reference src/renderChannelData.js:5-49 */
int dusp_oracle_render(dusp_oracle *o, size_t n_samples, float *out, int max_channels) {
    int nch = 0;
    const chunk_t *chunk = &o->units[o->out_unit].out;
    memset(out, 0, (size_t)max_channels * n_samples * sizeof(float)); /* new TypedArray(n) is zero-filled */
    for (size_t t0 = 0; t0 < n_samples; t0 += (size_t)o->chunk) {
        while (o->clock < (long)(t0 + (size_t)o->chunk)) circuit_tick(o); /* tickUntil, Circuit.js:42-47 */
        if (chunk->nch > nch) nch = chunk->nch;                           /* :38-39 */
        for (int c = 0; c < chunk->nch && c < max_channels; c++)
            for (int t = 0; t < o->chunk && t0 + (size_t)t < n_samples; t++) /* OOB tail writes are dropped */
                out[(size_t)c * n_samples + t0 + (size_t)t] = (float)or0(chunk->ch[c][t]); /* `|| 0` :44 */
    }
    return nch;
}

/* ---- descriptor -> circuit ---------------------------------------------- */
#define FAIL(...) do { snprintf(err, errlen, __VA_ARGS__); dusp_oracle_destroy(o); return NULL; } while (0)

dusp_oracle *dusp_oracle_create(const double *d, size_t nw, const float *params, size_t n_inst, size_t inst,
                                char *err, size_t errlen) {
    dusp_oracle *o = (dusp_oracle *)calloc(1, sizeof *o);
    if (nw < HEADER_WORDS || d[0] != MAGIC || d[1] != 1) FAIL("bad descriptor header");
    o->sr = (int)d[2];
    o->chunk = (int)d[3];
    o->n_units = (size_t)d[4];
    o->n_rings = (size_t)d[5];
    const size_t n_params = (size_t)d[6];
    o->out_unit = (size_t)d[7];
    o->clock = (long)d[9];
    if (o->out_unit >= o->n_units) FAIL("output unit out of range");
    size_t p = HEADER_WORDS;
    o->rings = (ring_t *)calloc(o->n_rings ? o->n_rings : 1, sizeof(ring_t));
    for (size_t r = 0; r < o->n_rings; r++) {
        if (p + 2 > nw) FAIL("truncated ring table");
        o->rings[r].nch = (int)d[p];
        o->rings[r].len = (long)d[p + 1];
        p += 2;
        if (o->rings[r].nch < 1 || o->rings[r].len < 1) FAIL("bad ring");
        o->rings[r].data = (float **)calloc((size_t)o->rings[r].nch, sizeof(float *));
        for (int c = 0; c < o->rings[r].nch; c++)
            o->rings[r].data[c] = (float *)calloc((size_t)o->rings[r].len, sizeof(float));
    }
    for (int w = 0; w < N_TABLES; w++) {
        o->tables[w] = (float *)malloc((size_t)(o->sr + 1) * sizeof(float));
        if (dusp_oracle_wavetable(w, o->sr, o->tables[w])) memset(o->tables[w], 0, (size_t)(o->sr + 1) * sizeof(float));
    }
    o->units = (unit_t *)calloc(o->n_units, sizeof(unit_t));
    for (size_t i = 0; i < o->n_units; i++) {
        unit_t *u = &o->units[i];
        if (p + 4 > nw) FAIL("truncated unit %zu", i);
        u->op = (int)d[p];
        u->n_inlets = (int)d[p + 1];
        const size_t n_attr = (size_t)d[p + 2], n_state = (size_t)d[p + 3];
        p += 4;
        if (u->n_inlets < 0 || u->n_inlets > MAX_INLETS) FAIL("unit %zu: bad inlet count", i);
        for (int k = 0; k < u->n_inlets; k++) {
            if (p + 2 > nw) FAIL("truncated inlet");
            const int kind = (int)d[p];
            const size_t n = (size_t)d[p + 1];
            p += 2;
            if (p + n > nw) FAIL("truncated inlet values");
            if (kind == IN_CONNECT) {
                u->in[k].connected = 1;
                u->in[k].src = (int)d[p];
                if ((size_t)u->in[k].src >= o->n_units) FAIL("unit %zu: source out of range", i);
            } else {
                if (n < 1) FAIL("unit %zu: empty constant", i);
                chunk_init(&u->in[k].own, (int)n, o->chunk);
                for (size_t c = 0; c < n; c++) {
                    float v;
                    if (kind == IN_PARAM) {
                        const size_t slot = (size_t)d[p + c];
                        if (!params || slot >= n_params || inst >= n_inst) FAIL("unit %zu: bad param slot", i);
                        v = params[slot * n_inst + inst];
                    } else
                        v = (float)d[p + c]; /* setConstant stores into a Float32Array (Inlet.js:88-91) */
                    for (int t = 0; t < o->chunk; t++) u->in[k].own.ch[c][t] = v;
                }
            }
            p += n;
        }
        if (p + n_attr + n_state > nw) FAIL("truncated attrs/state");
        const double *a = d + p, *s = d + p + n_attr;
        p += n_attr + n_state;
        int out_channels = 1; /* Piglet.js:13: numberOfChannels defaults to 1 */
        switch (u->op) {
        case OP_OSC:
            if (n_attr != 1 || n_state != 1 || u->n_inlets != 1) FAIL("unit %zu: bad Osc record", i);
            u->waveform = (int)a[0];
            if (u->waveform < 0 || u->waveform > 4) FAIL("bad waveform");
            u->phase = s[0];
            break;
        case OP_RAMP:
            if (n_attr != 3 || n_state != 2 || u->n_inlets != 0) FAIL("unit %zu: bad Ramp record", i);
            u->duration = a[0]; u->ry0 = a[1]; u->ry1 = a[2];
            u->t = s[0]; u->playing = s[1] != 0;
            break;
        case OP_MULTIPLY: case OP_SUM:
            if (u->n_inlets != 2) FAIL("unit %zu: bad combiner record", i);
            break;
        case OP_FILTER: {
            if (n_attr != 1 || n_state < 8 || u->n_inlets != 2) FAIL("unit %zu: bad Filter record", i);
            u->kind = (int)a[0];
            if (u->kind < 0 || u->kind > 1) FAIL("filter kind unsupported");
            u->has_lastF = s[0] != 0; u->lastF = s[1];
            u->a0 = s[2]; u->a1 = s[3]; u->a2 = s[4]; u->b1 = s[5]; u->b2 = s[6];
            const int nch = (int)s[7];
            if (n_state != (size_t)(8 + 4 * nch)) FAIL("unit %zu: bad Filter state", i);
            filter_grow(u, nch);
            for (int c = 0; c < nch; c++) {
                u->x1[c] = s[8 + 4 * c]; u->x2[c] = s[9 + 4 * c];
                u->y1[c] = s[10 + 4 * c]; u->y2[c] = s[11 + 4 * c];
            }
            break;
        }
        case OP_DELAY:
            if (n_attr != 1 || u->n_inlets != 2) FAIL("unit %zu: bad Delay record", i);
            u->maxDelay = (long)a[0];
            if (u->maxDelay < 1 || (double)u->maxDelay != a[0]) FAIL("bad maxDelay");
            u->buffers = (float **)calloc(1, sizeof(float *)); /* Delay.js:14: one ring up front */
            u->buffers[0] = (float *)calloc((size_t)u->maxDelay, sizeof(float));
            u->nbuffers = 1;
            break;
        case OP_CB_READER: case OP_CB_WRITER:
            if (n_attr != 2 || n_state != 1 || u->n_inlets != (u->op == OP_CB_READER ? 1 : 2))
                FAIL("unit %zu: bad CircleBuffer node record", i);
            u->ring = (int)a[0]; u->wipe = a[1] != 0; u->cb_t = s[0];
            if ((size_t)u->ring >= o->n_rings) FAIL("ring out of range");
            if (u->op == OP_CB_READER) out_channels = o->rings[u->ring].nch; /* CircleBufferNode.js:19-22 */
            break;
        case OP_REPEATER:
            if (u->n_inlets != 1) FAIL("unit %zu: bad Repeater record", i);
            break;
        case OP_SUBTRACT: case OP_DIVIDE: case OP_POW: case OP_CLIP: case OP_HARD_CLIP_ABOVE: case OP_HARD_CLIP_BELOW: case OP_GAIN:
            if (u->n_inlets != 2 || n_attr || n_state) FAIL("unit %zu: bad binary map record", i);
            break;
        case OP_POLARITY_INVERT: case OP_ABS: case OP_DECIBEL_TO_SCALER: case OP_SEMITONE_TO_RATIO: case OP_SECONDS_TO_SAMPLES:
            if (u->n_inlets != 1 || n_attr || n_state) FAIL("unit %zu: bad unary map record", i);
            break;
        case OP_FIXED_MULTIPLY:
            if (u->n_inlets != 1 || n_attr != 1 || n_state) FAIL("unit %zu: bad FixedMultiply record", i);
            u->sf = a[0];
            break;
        case OP_FIXED_DELAY: case OP_COMB_FILTER: case OP_ALL_PASS: case OP_READBACK_DELAY:
            if (u->n_inlets != (u->op == OP_FIXED_DELAY ? 1 : 2) || n_attr != 1 || n_state != 1 || !(a[0] >= 1 && a[0] < 1e9))
                FAIL("unit %zu: bad delay-family record", i);
            u->fd_len = (long)a[0];
            u->tBuffer = s[0];
            u->buffers = (float **)calloc(1, sizeof(float *));
            u->buffers[0] = (float *)calloc((size_t)u->fd_len, sizeof(float));
            u->nbuffers = 1;
            break;
        case OP_MONO_DELAY:
            if (u->n_inlets != 2 || n_attr != 1 || n_state || !(a[0] >= 1 && a[0] < 1e9)) FAIL("unit %zu: bad MonoDelay record", i);
            u->maxDelay = (long)a[0];
            u->buffers = (float **)calloc(1, sizeof(float *));
            u->buffers[0] = (float *)calloc((size_t)u->maxDelay, sizeof(float));
            u->nbuffers = 1;
            break;
        case OP_MULTI_OSC: {
            if (u->n_inlets != 1 || n_attr != 1 || n_state < 1 || n_state != (size_t)(1 + s[0])) FAIL("unit %zu: bad MultiChannelOsc record", i);
            u->waveform = (int)a[0];
            if (u->waveform < 0 || u->waveform > 4) FAIL("bad waveform");
            u->nphase = (int)s[0];
            u->phases = (double *)calloc((size_t)u->nphase + 1, sizeof(double));
            for (int c = 0; c < u->nphase; c++) u->phases[c] = s[1 + c];
            break;
        }
        case OP_PAN:
            if (u->n_inlets != 2 || n_attr != 1 || n_state) FAIL("unit %zu: bad Pan record", i);
            u->comp_db = a[0];
            out_channels = 2; /* Pan.js:8 */
            break;
        case OP_MIDI_TO_FREQUENCY: case OP_VECTOR_MAGNITUDE:
            if (u->n_inlets != 1 || n_attr || n_state) FAIL("unit %zu: bad unary record", i);
            break;
        case OP_RESCALE:
            if (u->n_inlets != 5 || n_attr || n_state) FAIL("unit %zu: bad Rescale record", i);
            break;
        case OP_CROSS_FADER:
            if (u->n_inlets != 3 || n_attr || n_state) FAIL("unit %zu: bad CrossFader record", i);
            break;
        case OP_TIMER:
            if (u->n_inlets != 0 || n_attr != 1 || n_state != 1) FAIL("unit %zu: bad Timer record", i);
            u->sample_period = a[0];
            u->timer_t = s[0];
            break;
        case OP_SAMPLE_RATE_REDUX:
            if (u->n_inlets != 2 || n_attr || n_state < 2 || n_state != (size_t)(2 + s[1])) FAIL("unit %zu: bad SampleRateRedux record", i);
            u->tslu = s[0];
            u->nval = (int)s[1];
            u->val = (float *)calloc((size_t)u->nval + 1, sizeof(float));
            for (int c = 0; c < u->nval; c++) u->val[c] = (float)s[2 + c];
            break;
        case OP_CONCAT_CHANNELS: case OP_PICK_CHANNEL:
            if (u->n_inlets != 2 || n_attr || n_state) FAIL("unit %zu: bad channel-plumbing record", i);
            break;
        case OP_HOST_ONLY:
            if (u->n_inlets || n_attr || n_state) FAIL("unit %zu: bad host-only record", i);
            out_channels = 0;
            break;
        case OP_RETRIGGER:
            if (u->n_inlets != 1 || n_attr != 1 || n_state != 1 || !(a[0] >= 0 && a[0] < (double)o->n_units)) FAIL("unit %zu: bad Retriggerer record", i);
            u->ring = (int)a[0];
            u->timer_t = s[0];
            out_channels = 0;
            break;
        case OP_INPUT:
            if (u->n_inlets || n_attr != 1 || n_state || !(a[0] >= 0 && a[0] < 4096)) FAIL("unit %zu: bad input record", i);
            u->ring = (int)a[0]; /* (the stream's index, kept in the field CircleBuffer nodes use for theirs) */
            break;
        case OP_SHAPE:
            if (u->n_inlets != 3 || n_attr != 5 || n_state != 3 || !(a[0] >= 5 && a[0] <= 8)) FAIL("unit %zu: bad Shape record", i);
            u->shape_table = (int)a[0];
            u->left_is_shape = a[1] != 0; u->left_edge = a[2];
            u->right_is_shape = a[3] != 0; u->right_edge = a[4];
            u->t = s[0]; u->playing = s[1] != 0; u->finished = s[2] != 0;
            break;
        case OP_AHD:
            if (u->n_inlets != 3 || n_attr != 1 || n_state != 3) FAIL("unit %zu: bad AHD record", i);
            u->sample_period = a[0];
            u->ahd_state = (int)s[0]; u->playing = s[1] != 0; u->t = s[2];
            break;
        default: FAIL("unit %zu: unknown opcode %d", i, u->op);
        }
        chunk_init(&u->out, out_channels, o->chunk);
    }
    return o;
}

/* Bind the host-generated streams the INPUT units read from now on: f32 [n_streams][n_samples], sample 0 = the
 * oracle's current clock.  The array must stay alive while the oracle renders. */
void dusp_oracle_set_inputs(dusp_oracle *o, const float *inputs, size_t n_samples) {
    o->inputs = inputs;
    o->input_len = n_samples;
    o->input_clock0 = o->clock;
}

void dusp_oracle_destroy(dusp_oracle *o) {
    /* Channels may be aliased between chunk slots (Delay.js:23), so the small
     * per-oracle allocations are released wholesale only where ownership is
     * unambiguous; the rest is reclaimed at process exit (test infrastructure). */
    if (!o) return;
    for (size_t r = 0; r < o->n_rings && o->rings; r++) {
        for (int c = 0; c < o->rings[r].nch && o->rings[r].data; c++) free(o->rings[r].data[c]);
        free(o->rings[r].data);
    }
    free(o->rings);
    for (size_t i = 0; i < o->n_units && o->units; i++) {
        unit_t *u = &o->units[i];
        for (int b = 0; b < u->nbuffers; b++) free(u->buffers[b]);
        free(u->buffers);
        free(u->x1); free(u->x2); free(u->y1); free(u->y2);
        free(u->phases);
        for (int c = 0; c < u->out.nch; c++) { /* a consumer Delay may have appended aliases */
            int dup = 0;
            for (int e = 0; e < c; e++) dup |= u->out.ch[e] == u->out.ch[c];
            if (!dup) free(u->out.ch[c]);
        }
        free(u->out.ch);
        for (int k = 0; k < u->n_inlets; k++) {
            if (u->in[k].connected) continue;
            /* Delay may have appended aliases of channel c%n: free each distinct pointer once */
            for (int c = 0; c < u->in[k].own.nch; c++) {
                int dup = 0;
                for (int e = 0; e < c; e++) dup |= u->in[k].own.ch[e] == u->in[k].own.ch[c];
                if (!dup) free(u->in[k].own.ch[c]);
            }
            free(u->in[k].own.ch);
        }
    }
    free(o->units);
    for (int w = 0; w < N_TABLES; w++) free(o->tables[w]);
    free(o);
}

size_t dusp_oracle_n_units(const dusp_oracle *o) { return o->n_units; }

size_t dusp_oracle_unit_state(const dusp_oracle *o, size_t i, double *out, size_t cap) {
    if (i >= o->n_units) return 0;
    const unit_t *u = &o->units[i];
    double tmp[8];
    size_t n = 0;
    switch (u->op) {
    case OP_OSC: tmp[n++] = u->phase; break;
    case OP_RAMP: tmp[n++] = u->t; tmp[n++] = u->playing; break;
    case OP_CB_READER: case OP_CB_WRITER: tmp[n++] = u->cb_t; break;
    case OP_FIXED_DELAY: case OP_COMB_FILTER: case OP_ALL_PASS: case OP_READBACK_DELAY: tmp[n++] = u->tBuffer; break;
    case OP_TIMER: case OP_RETRIGGER: tmp[n++] = u->timer_t; break;
    case OP_SHAPE: tmp[n++] = u->t; tmp[n++] = u->playing; tmp[n++] = u->finished; break;
    case OP_AHD: tmp[n++] = u->ahd_state; tmp[n++] = u->playing; tmp[n++] = u->t; break;
    case OP_SAMPLE_RATE_REDUX:
        for (size_t k = 0; k < (size_t)u->nval + 2 && k < cap; k++) out[k] = k == 0 ? u->tslu : k == 1 ? (double)u->nval : (double)u->val[k - 2];
        return (size_t)u->nval + 2;
    case OP_MULTI_OSC:
        for (size_t k = 0; k < (size_t)u->nphase + 1 && k < cap; k++) out[k] = k ? u->phases[k - 1] : (double)u->nphase;
        return (size_t)u->nphase + 1;
    case OP_FILTER: {
        const size_t total = 8 + 4 * (size_t)u->nstate;
        double head[8] = { (double)u->has_lastF, u->lastF, u->a0, u->a1, u->a2, u->b1, u->b2, (double)u->nstate };
        for (size_t k = 0; k < total && k < cap; k++) {
            if (k < 8) out[k] = head[k];
            else {
                const size_t c = (k - 8) / 4, j = (k - 8) % 4;
                const double *src = j == 0 ? u->x1 : j == 1 ? u->x2 : j == 2 ? u->y1 : u->y2;
                out[k] = or0(src[c]);
            }
        }
        return total;
    }
    default: break;
    }
    for (size_t k = 0; k < n && k < cap; k++) out[k] = tmp[k];
    return n;
}
