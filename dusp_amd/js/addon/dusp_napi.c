/* dusp_napi.c — thin N-API addon binding the C ABI of include/dusp_hip.h for Node.js.
 *
 * Raw C against <node_api.h> (N-API v4+, present in Node 12): no node-gyp, no node-addon-api.
 *   deviceCount() -> number of HIP devices this process sees
 *   ctxCreate(device) -> ctx            tableUpload(ctx, id, Float32Array)
 *   programBuild(ctx, Float64Array words, engine) -> prog
 *   programInfo(prog) -> { sampleRate, nUnits, nOutChannels, nParams, nInputs, engine, shape, nDeviceOps }
 *   render(prog, nInstances, nSamples, Float32Array params | null[, interleaved[, Float32Array inputs]]) -> Promise<Float32Array>
 *         (runs dusp_render_host on the libuv pool so the event loop stays live)
 *   stateDownload(prog, instance, unit) -> Float64Array
 *   programDestroy(prog), ctxDestroy(ctx), version(), abiVersion()
 * Failures surface the way the reference's do: synchronous calls THROW A STRING, render() REJECTS
 * WITH A STRING (reference src/renderChannelData.js:12-17 throws strings inside an async function).
 */
#include <node_api.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../../include/dusp_hip.h"

#define PINNED_MIN_BYTES ((size_t)1 << 20) /* results of at least 1 MiB live in pinned memory (dusp_host_alloc) */

/* A dusp_ctx is not thread-safe and async renders run on pool threads: calls on ONE context are serialised by that context's
 * lock (ctx_box.lock).  Different contexts are independent (include/dusp_hip.h "Threading"), so renders on different contexts —
 * one per GPU when renderMany shards its instances over the node's devices — run side by side on the pool.  g_create_lock only
 * orders context creation and destruction (device enumeration, the process-wide settings the first context reads). */
static pthread_mutex_t g_create_lock = PTHREAD_MUTEX_INITIALIZER;

#define NAPI_OK(call)                                                          \
    do {                                                                       \
        if ((call) != napi_ok) {                                               \
            throw_string(env, "dusp-hip: N-API call failed: " #call);         \
            return NULL;                                                       \
        }                                                                      \
    } while (0)

static void throw_string(napi_env env, const char *msg) {
    napi_value s;
    if (napi_create_string_utf8(env, msg, NAPI_AUTO_LENGTH, &s) == napi_ok) napi_throw(env, s);
}

/* Lifetimes.  A program and a pinned PCM buffer both point into their context, and an async render points into its
 * program, so the boxes count what still depends on them (all counting happens on the JS thread):
 *   ctx_box.refs        the context external + every program box + every PCM ArrayBuffer backed by the context's pinned pool;
 *                       the box itself is freed when the last of them is finalized
 *   ctx_box.n_programs / n_buffers   live programs / pinned PCM buffers: ctxDestroy throws while either is non-zero
 *   prog_box.in_flight  renders queued or running: programDestroy during one is deferred to its completion */
typedef struct {
    dusp_ctx *ctx;
    pthread_mutex_t lock; /* every library call on this context or on one of its programs */
    int refs, n_programs, n_buffers;
} ctx_box;
typedef struct {
    dusp_program *prog;
    ctx_box *cb;
    int in_flight, destroy_deferred;
} prog_box;

static void ctx_unref(ctx_box *b) {
    if (--b->refs > 0) return;
    if (b->ctx) { /* never destroyed explicitly: the last dependant is gone, release the device side too */
        pthread_mutex_lock(&g_create_lock);
        dusp_ctx_destroy(b->ctx);
        pthread_mutex_unlock(&g_create_lock);
    }
    pthread_mutex_destroy(&b->lock);
    free(b);
}
static void ctx_finalize(napi_env env, void *data, void *hint) {
    (void)env; (void)hint;
    ctx_unref((ctx_box *)data);
}
static void prog_destroy_now(prog_box *b) {
    if (!b->prog) return;
    pthread_mutex_lock(&b->cb->lock);
    dusp_program_destroy(b->prog);
    pthread_mutex_unlock(&b->cb->lock);
    b->prog = NULL;
    b->cb->n_programs--;
}
static void prog_finalize(napi_env env, void *data, void *hint) {
    (void)env; (void)hint;
    prog_box *b = (prog_box *)data; /* (no render can be in flight: a job holds a reference on this external) */
    prog_destroy_now(b);
    ctx_unref(b->cb);
    free(b);
}

static int get_args(napi_env env, napi_callback_info info, size_t want, napi_value *argv) {
    size_t argc = want;
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < want) {
        throw_string(env, "dusp-hip: wrong number of arguments");
        return 0;
    }
    return 1;
}
static ctx_box *as_ctx(napi_env env, napi_value v) {
    void *p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p || !((ctx_box *)p)->ctx) {
        throw_string(env, "dusp-hip: not a live context");
        return NULL;
    }
    return (ctx_box *)p;
}
static prog_box *as_prog(napi_env env, napi_value v) {
    void *p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p || !((prog_box *)p)->prog || ((prog_box *)p)->destroy_deferred) {
        throw_string(env, "dusp-hip: not a live program");
        return NULL;
    }
    return (prog_box *)p;
}
static int typed_array(napi_env env, napi_value v, napi_typedarray_type want, void **data, size_t *len) {
    bool is = false;
    napi_typedarray_type type;
    napi_value ab;
    size_t off;
    if (napi_is_typedarray(env, v, &is) != napi_ok || !is) return 0;
    if (napi_get_typedarray_info(env, v, &type, len, data, &ab, &off) != napi_ok || type != want) return 0;
    return 1;
}

static napi_value fn_version(napi_env env, napi_callback_info info) {
    (void)info;
    napi_value s;
    NAPI_OK(napi_create_string_utf8(env, dusp_version(), NAPI_AUTO_LENGTH, &s));
    return s;
}
static napi_value fn_abi_version(napi_env env, napi_callback_info info) {
    (void)info;
    napi_value v;
    NAPI_OK(napi_create_int32(env, dusp_abi_version(), &v));
    return v;
}

/* deviceCount() -> HIP devices this process sees (dusp_device_count); throws the library's message when there is no usable device */
static napi_value fn_device_count(napi_env env, napi_callback_info info) {
    (void)info;
    pthread_mutex_lock(&g_create_lock);
    int n = dusp_device_count();
    char msg[512];
    if (n < 0) snprintf(msg, sizeof msg, "dusp-hip: %s", dusp_last_error(NULL));
    pthread_mutex_unlock(&g_create_lock);
    if (n < 0) {
        throw_string(env, msg);
        return NULL;
    }
    napi_value v;
    NAPI_OK(napi_create_int32(env, n, &v));
    return v;
}

static napi_value fn_ctx_create(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    int32_t device = -1;
    napi_get_value_int32(env, argv[0], &device);
    dusp_ctx *ctx = NULL;
    pthread_mutex_lock(&g_create_lock);
    int rc = dusp_ctx_create(device, &ctx);
    char msg[512];
    if (rc != DUSP_OK) snprintf(msg, sizeof msg, "dusp-hip: %s", dusp_last_error(NULL));
    pthread_mutex_unlock(&g_create_lock);
    if (rc != DUSP_OK) {
        throw_string(env, msg);
        return NULL;
    }
    ctx_box *b = (ctx_box *)calloc(1, sizeof *b);
    if (!b) {
        dusp_ctx_destroy(ctx);
        throw_string(env, "dusp-hip: out of host memory");
        return NULL;
    }
    b->ctx = ctx;
    b->refs = 1;
    pthread_mutex_init(&b->lock, NULL);
    napi_value ext;
    if (napi_create_external(env, b, ctx_finalize, NULL, &ext) != napi_ok) {
        dusp_ctx_destroy(ctx);
        pthread_mutex_destroy(&b->lock);
        free(b);
        throw_string(env, "dusp-hip: could not wrap the context");
        return NULL;
    }
    return ext;
}

static napi_value fn_ctx_destroy(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    ctx_box *b = as_ctx(env, argv[0]);
    if (!b) return NULL;
    if (b->n_programs > 0 || b->n_buffers > 0) { /* they point into the context (device, pinned pool) */
        char msg[160];
        snprintf(msg, sizeof msg, "dusp-hip: ctxDestroy: %d program(s) and %d rendered PCM buffer(s) of this context are still alive",
                 b->n_programs, b->n_buffers);
        throw_string(env, msg);
        return NULL;
    }
    pthread_mutex_lock(&g_create_lock);
    dusp_ctx_destroy(b->ctx);
    pthread_mutex_unlock(&g_create_lock);
    b->ctx = NULL;
    return NULL;
}

static napi_value fn_table_upload(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return NULL;
    ctx_box *b = as_ctx(env, argv[0]);
    if (!b) return NULL;
    int32_t id = -1;
    napi_get_value_int32(env, argv[1], &id);
    void *data;
    size_t len;
    if (!typed_array(env, argv[2], napi_float32_array, &data, &len)) {
        throw_string(env, "dusp-hip: tableUpload expects a Float32Array");
        return NULL;
    }
    pthread_mutex_lock(&b->lock);
    int rc = dusp_table_upload(b->ctx, id, (const float *)data, len);
    char msg[512];
    if (rc != DUSP_OK) snprintf(msg, sizeof msg, "dusp-hip: %s", dusp_last_error(b->ctx));
    pthread_mutex_unlock(&b->lock);
    if (rc != DUSP_OK) throw_string(env, msg);
    return NULL;
}

static napi_value fn_program_build(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return NULL;
    ctx_box *b = as_ctx(env, argv[0]);
    if (!b) return NULL;
    void *data;
    size_t len;
    if (!typed_array(env, argv[1], napi_float64_array, &data, &len)) {
        throw_string(env, "dusp-hip: programBuild expects a Float64Array of descriptor words");
        return NULL;
    }
    int32_t engine = 0;
    napi_get_value_int32(env, argv[2], &engine);
    dusp_program *prog = NULL;
    pthread_mutex_lock(&b->lock);
    int rc = dusp_program_build(b->ctx, (const double *)data, len, engine, &prog);
    char msg[512];
    if (rc != DUSP_OK) snprintf(msg, sizeof msg, "dusp-hip: %s", dusp_last_error(b->ctx));
    pthread_mutex_unlock(&b->lock);
    if (rc != DUSP_OK) {
        throw_string(env, msg);
        return NULL;
    }
    prog_box *pb = (prog_box *)calloc(1, sizeof *pb);
    napi_value ext;
    if (!pb || napi_create_external(env, pb, prog_finalize, NULL, &ext) != napi_ok) {
        pthread_mutex_lock(&b->lock);
        dusp_program_destroy(prog);
        pthread_mutex_unlock(&b->lock);
        free(pb);
        throw_string(env, "dusp-hip: could not wrap the program");
        return NULL;
    }
    pb->prog = prog;
    pb->cb = b;
    b->refs++;
    b->n_programs++;
    return ext;
}

/* programContinue(prog, words): dusp_program_continue — re-arm a rendered program from a later extraction of
 * the same circuit (event-segmented rendering). */
static napi_value fn_program_continue(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    prog_box *pb = as_prog(env, argv[0]);
    if (!pb) return NULL;
    void *data;
    size_t len;
    if (!typed_array(env, argv[1], napi_float64_array, &data, &len)) {
        throw_string(env, "dusp-hip: programContinue expects a Float64Array of descriptor words");
        return NULL;
    }
    pthread_mutex_lock(&pb->cb->lock);
    int rc = dusp_program_continue(pb->prog, (const double *)data, len);
    char msg[512];
    if (rc != DUSP_OK) snprintf(msg, sizeof msg, "dusp-hip: %s", dusp_last_error(pb->cb->ctx));
    pthread_mutex_unlock(&pb->cb->lock);
    if (rc != DUSP_OK) {
        throw_string(env, msg);
        return NULL;
    }
    napi_value undef;
    NAPI_OK(napi_get_undefined(env, &undef));
    return undef;
}

static napi_value fn_program_destroy(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    prog_box *pb = as_prog(env, argv[0]);
    if (!pb) return NULL;
    if (pb->in_flight > 0) pb->destroy_deferred = 1; /* a render is using it on a pool thread: destroyed when that completes */
    else prog_destroy_now(pb);
    return NULL;
}

static napi_value fn_program_info(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    prog_box *pb = as_prog(env, argv[0]);
    if (!pb) return NULL;
    dusp_program_info pi;
    dusp_program_info_get(pb->prog, &pi);
    napi_value obj, v;
    NAPI_OK(napi_create_object(env, &obj));
#define SET_U32(name, val)                                   \
    NAPI_OK(napi_create_uint32(env, (val), &v));             \
    NAPI_OK(napi_set_named_property(env, obj, name, v))
    SET_U32("sampleRate", pi.sample_rate);
    SET_U32("chunkSize", pi.chunk_size);
    SET_U32("nUnits", pi.n_units);
    SET_U32("nOutChannels", pi.n_out_channels);
    SET_U32("nParams", pi.n_params);
    SET_U32("nInputs", pi.n_inputs);
    SET_U32("nDeviceOps", pi.n_device_ops);
#undef SET_U32
    NAPI_OK(napi_create_string_utf8(env, pi.engine == DUSP_ENGINE_FUSED ? "fused" : "chunk", NAPI_AUTO_LENGTH, &v));
    NAPI_OK(napi_set_named_property(env, obj, "engine", v));
    NAPI_OK(napi_create_string_utf8(env, pi.shape, NAPI_AUTO_LENGTH, &v));
    NAPI_OK(napi_set_named_property(env, obj, "shape", v));
    return obj;
}

static napi_value fn_state_download(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return NULL;
    prog_box *pb = as_prog(env, argv[0]);
    if (!pb) return NULL;
    uint32_t instance = 0, unit = 0;
    napi_get_value_uint32(env, argv[1], &instance);
    napi_get_value_uint32(env, argv[2], &unit);
    double words[512];
    pthread_mutex_lock(&pb->cb->lock);
    int n = dusp_state_download(pb->prog, instance, unit, words, 512);
    char msg[512];
    if (n < 0) snprintf(msg, sizeof msg, "dusp-hip: %s", dusp_last_error(pb->cb->ctx));
    pthread_mutex_unlock(&pb->cb->lock);
    if (n < 0) {
        throw_string(env, msg);
        return NULL;
    }
    if (n > 512) n = 512;
    napi_value ab, arr;
    void *mem;
    NAPI_OK(napi_create_arraybuffer(env, (size_t)n * sizeof(double), &mem, &ab));
    memcpy(mem, words, (size_t)n * sizeof(double));
    NAPI_OK(napi_create_typedarray(env, napi_float64_array, (size_t)n, ab, 0, &arr));
    return arr;
}

/* ---- async render ---- */
typedef struct {
    napi_async_work work;
    napi_deferred deferred;
    napi_ref prog_ref; /* keeps the program external alive while the render is in flight */
    prog_box *pb;
    int out_pinned; /* out comes from dusp_host_alloc (the context's pinned pool) rather than malloc */
    size_t n_instances, n_samples, n_floats;
    float *params;
    float *inputs; /* host-generated streams [nInputs][nInstances][nSamples] (copied: the caller may reuse its array) */
    float *out;
    int interleaved; /* frames [instance][sample][channel] instead of planar [instance][channel][sample] */
    int rc;
    char err[512];
} render_job;

static void render_execute(napi_env env, void *data) {
    (void)env;
    render_job *j = (render_job *)data;
    pthread_mutex_lock(&j->pb->cb->lock); /* (this context's: renders on other contexts — other GPUs — run beside this one) */
    dusp_program *prog = j->pb->prog; /* (in_flight > 0 keeps it alive: programDestroy defers) */
    if (!prog) {
        j->rc = DUSP_ERR_STATE;
        snprintf(j->err, sizeof j->err, "dusp-hip: render: the program has been destroyed");
    } else {
        if (j->inputs) j->rc = dusp_render_host_inputs(prog, j->n_instances, j->n_samples, j->params, j->inputs, j->out, j->interleaved);
        else
            j->rc = j->interleaved ? dusp_render_host_interleaved(prog, j->n_instances, j->n_samples, j->params, j->out)
                                   : dusp_render_host(prog, j->n_instances, j->n_samples, j->params, j->out);
        if (j->rc != DUSP_OK) snprintf(j->err, sizeof j->err, "dusp-hip: %s", dusp_last_error(j->pb->cb->ctx));
    }
    pthread_mutex_unlock(&j->pb->cb->lock);
}
static void free_pcm(napi_env env, void *data, void *hint) {
    (void)env; (void)hint;
    free(data);
}
/* a PCM buffer from the context's pinned pool goes back to it when its ArrayBuffer is collected */
static void free_pinned_pcm(napi_env env, void *data, void *hint) {
    (void)env;
    ctx_box *cb = (ctx_box *)hint;
    if (cb->ctx) {
        pthread_mutex_lock(&cb->lock);
        dusp_host_free(cb->ctx, data);
        pthread_mutex_unlock(&cb->lock);
    }
    cb->n_buffers--;
    ctx_unref(cb);
}
static void release_out(render_job *j) { /* an output buffer that never reached JavaScript */
    if (!j->out) return;
    if (j->out_pinned) {
        ctx_box *cb = j->pb->cb;
        if (cb->ctx) {
            pthread_mutex_lock(&cb->lock);
            dusp_host_free(cb->ctx, j->out);
            pthread_mutex_unlock(&cb->lock);
        }
        cb->n_buffers--;
        ctx_unref(cb);
    } else
        free(j->out);
    j->out = NULL;
}
static void render_complete(napi_env env, napi_status status, void *data) {
    render_job *j = (render_job *)data;
    napi_value result;
    if (status != napi_ok && j->rc == DUSP_OK) {
        j->rc = DUSP_ERR_STATE;
        snprintf(j->err, sizeof j->err, "dusp-hip: render was cancelled");
    }
    if (j->rc == DUSP_OK) {
        napi_value ab;
        if (napi_create_external_arraybuffer(env, j->out, j->n_floats * sizeof(float), j->out_pinned ? free_pinned_pcm : free_pcm,
                                             j->out_pinned ? (void *)j->pb->cb : NULL, &ab) == napi_ok &&
            napi_create_typedarray(env, napi_float32_array, j->n_floats, ab, 0, &result) == napi_ok) {
            j->out = NULL; /* owned by the ArrayBuffer now */
            napi_resolve_deferred(env, j->deferred, result);
        } else {
            napi_create_string_utf8(env, "dusp-hip: could not wrap the PCM buffer", NAPI_AUTO_LENGTH, &result);
            napi_reject_deferred(env, j->deferred, result);
        }
    } else {
        napi_create_string_utf8(env, j->err, NAPI_AUTO_LENGTH, &result);
        napi_reject_deferred(env, j->deferred, result); /* a string, like the reference's rejections */
    }
    release_out(j);
    if (--j->pb->in_flight == 0 && j->pb->destroy_deferred) prog_destroy_now(j->pb); /* programDestroy came while this render ran */
    napi_delete_reference(env, j->prog_ref);
    napi_delete_async_work(env, j->work);
    free(j->params);
    free(j->inputs);
    free(j);
}

/* render(prog, nInstances, nSamples, params | null [, interleaved]) -> Promise<Float32Array> */
static napi_value fn_render(napi_env env, napi_callback_info info) {
    napi_value argv[6];
    size_t argc = 6;
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < 4) {
        throw_string(env, "dusp-hip: wrong number of arguments");
        return NULL;
    }
    bool interleaved = false;
    if (argc >= 5) napi_get_value_bool(env, argv[4], &interleaved);
    prog_box *pb = as_prog(env, argv[0]);
    if (!pb) return NULL;
    double n_inst = 0, n_samples = 0;
    napi_get_value_double(env, argv[1], &n_inst);
    napi_get_value_double(env, argv[2], &n_samples);
    if (!(n_inst >= 1 && n_inst <= 16777216.0 && n_samples >= 1 && n_samples <= 2147483648.0)) {
        throw_string(env, "dusp-hip: render: nInstances / nSamples out of range");
        return NULL;
    }
    dusp_program_info pi;
    dusp_program_info_get(pb->prog, &pi);
    render_job *j = (render_job *)calloc(1, sizeof *j);
    if (!j) {
        throw_string(env, "dusp-hip: render: out of host memory");
        return NULL;
    }
    j->pb = pb;
    j->n_instances = (size_t)n_inst;
    j->n_samples = (size_t)n_samples;
    j->n_floats = j->n_instances * pi.n_out_channels * j->n_samples;
    j->interleaved = interleaved ? 1 : 0;
    napi_valuetype vt;
    napi_typeof(env, argv[3], &vt);
    if (vt != napi_null && vt != napi_undefined) {
        void *data;
        size_t len;
        if (!typed_array(env, argv[3], napi_float32_array, &data, &len) || len != (size_t)pi.n_params * j->n_instances) {
            free(j);
            throw_string(env, "dusp-hip: render: params must be a Float32Array of nParams * nInstances values");
            return NULL;
        }
        j->params = (float *)malloc(len * sizeof(float) + 1);
        if (!j->params) {
            free(j);
            throw_string(env, "dusp-hip: render: out of host memory for the parameter table");
            return NULL;
        }
        memcpy(j->params, data, len * sizeof(float));
    } else if (pi.n_params) {
        free(j);
        throw_string(env, "dusp-hip: render: this program needs a parameter table");
        return NULL;
    }
    if (argc >= 6) napi_typeof(env, argv[5], &vt);
    if (argc >= 6 && vt != napi_null && vt != napi_undefined) { /* inputs: Float32Array of nInputs * nInstances * nSamples values */
        void *data;
        size_t len;
        if (!typed_array(env, argv[5], napi_float32_array, &data, &len) || len != (size_t)pi.n_inputs * j->n_instances * j->n_samples || !len) {
            free(j->params);
            free(j);
            throw_string(env, "dusp-hip: render: inputs must be a Float32Array of nInputs * nInstances * nSamples values");
            return NULL;
        }
        j->inputs = (float *)malloc(len * sizeof(float) + 1);
        if (!j->inputs) {
            free(j->params);
            free(j);
            throw_string(env, "dusp-hip: render: out of host memory for the input streams");
            return NULL;
        }
        memcpy(j->inputs, data, len * sizeof(float));
    } else if (pi.n_inputs) {
        free(j->params);
        free(j);
        throw_string(env, "dusp-hip: render: this program reads host-generated input streams");
        return NULL;
    }
    /* The result buffer IS the ArrayBuffer JavaScript gets (renderChannelData.js:39 allocates per call; no copy here).  Large
     * results come from the context's pinned pool so that the download is one DMA at link speed; small ones (event-segmented
     * rendering makes hundreds a second) from malloc. */
    if (j->n_floats * sizeof(float) >= PINNED_MIN_BYTES) {
        void *p = NULL;
        pthread_mutex_lock(&pb->cb->lock);
        int rc = dusp_host_alloc(pb->cb->ctx, j->n_floats * sizeof(float), &p);
        pthread_mutex_unlock(&pb->cb->lock);
        if (rc == DUSP_OK) {
            j->out = (float *)p;
            j->out_pinned = 1;
            pb->cb->n_buffers++;
            pb->cb->refs++;
        }
    }
    if (!j->out) j->out = (float *)malloc(j->n_floats * sizeof(float) + 1);
    if (!j->out) {
        free(j->inputs);
        free(j->params);
        free(j);
        throw_string(env, "dusp-hip: render: out of host memory for the PCM buffer");
        return NULL;
    }
    napi_value promise, name;
    if (napi_create_promise(env, &j->deferred, &promise) != napi_ok || napi_create_reference(env, argv[0], 1, &j->prog_ref) != napi_ok ||
        napi_create_string_utf8(env, "dusp-hip render", NAPI_AUTO_LENGTH, &name) != napi_ok ||
        napi_create_async_work(env, NULL, name, render_execute, render_complete, j, &j->work) != napi_ok) {
        release_out(j);
        free(j->inputs);
        free(j->params);
        free(j);
        throw_string(env, "dusp-hip: render: could not queue the render");
        return NULL;
    }
    pb->in_flight++;
    if (napi_queue_async_work(env, j->work) != napi_ok) {
        pb->in_flight--;
        release_out(j);
        free(j->inputs);
        free(j->params);
        free(j);
        throw_string(env, "dusp-hip: render: could not queue the render");
        return NULL;
    }
    return promise;
}

static napi_value init(napi_env env, napi_value exports) {
    static const struct {
        const char *name;
        napi_callback fn;
    } fns[] = {
        {"version", fn_version},          {"abiVersion", fn_abi_version},     {"ctxCreate", fn_ctx_create},
        {"ctxDestroy", fn_ctx_destroy},   {"tableUpload", fn_table_upload},   {"programBuild", fn_program_build},
        {"programDestroy", fn_program_destroy}, {"programInfo", fn_program_info}, {"stateDownload", fn_state_download},
        {"render", fn_render},            {"programContinue", fn_program_continue}, {"deviceCount", fn_device_count},
    };
    for (size_t i = 0; i < sizeof fns / sizeof fns[0]; i++) {
        napi_value f;
        if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok ||
            napi_set_named_property(env, exports, fns[i].name, f) != napi_ok)
            return NULL;
    }
    return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
