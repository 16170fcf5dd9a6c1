"""ctypes binding of the C ABI in include/dusp_hip.h (dusp_amd/libdusp_hip.so).

There is deliberately no CPU fallback here: if the HIP library is missing or no
GPU is usable, every entry point raises.
"""
import ctypes
import threading
import os

import numpy as np

from .wavetables import N_TABLES, make_table

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DUSP_HIP_LIB") or os.path.join(_HERE, "libdusp_hip.so")  # (DUSP_HIP_LIB: another build of the library, for A/B runs)

ENGINE_AUTO, ENGINE_CHUNK, ENGINE_FUSED, ENGINE_WAVE, ENGINE_LOOP = 0, 1, 2, 3, 4
ENGINE_RESUMABLE = 0x100  # OR into the engine: the program will be continued (Program.continue_with)
ENGINE_NAMES = {ENGINE_CHUNK: "chunk", ENGINE_FUSED: "fused", ENGINE_WAVE: "wave"}  # (ENGINE_LOOP: accepted by the library, means AUTO since ABI v7)

EXPORTS = [
    "dusp_version", "dusp_abi_version", "dusp_last_error", "dusp_ctx_create", "dusp_ctx_destroy",
    "dusp_table_upload", "dusp_program_build", "dusp_program_destroy", "dusp_program_continue", "dusp_program_info_get",
    "dusp_render_device", "dusp_render_host", "dusp_render_host_interleaved", "dusp_interleave_device", "dusp_state_download",
    "dusp_last_kernel_ms", "dusp_fill_device", "dusp_render_device_inputs", "dusp_render_host_inputs",
    "dusp_host_alloc", "dusp_host_free", "dusp_circuit_kernel_source", "dusp_jit_cache_dir", "dusp_render_chain_window", "dusp_device_count",
]


class DuspHipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("%s (status %d)" % (message, status))
        self.status = status
        self.message = message


class ProgramInfo(ctypes.Structure):
    _fields_ = [("sample_rate", ctypes.c_uint32), ("chunk_size", ctypes.c_uint32), ("n_units", ctypes.c_uint32),
                ("n_out_channels", ctypes.c_uint32), ("n_params", ctypes.c_uint32), ("engine", ctypes.c_uint32),
                ("n_device_ops", ctypes.c_uint32), ("n_inputs", ctypes.c_uint32), ("shape", ctypes.c_char * 64)]


_lib = None


def load():
    """Load libdusp_hip.so (built by dusp_amd/csrc/Makefile).  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DuspHipError(-3, "HIP extension %s is missing: build it with `make -C dusp_amd/csrc` "
                               "(there is no CPU fallback)" % LIB_PATH)
    try:
        # PyTorch-ROCm ships its own libamdhip64; whichever HIP runtime initialises first owns the GPU
        # in this process, so let torch's copy load first and have this library bind to the same one.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    L.dusp_version.restype = ctypes.c_char_p
    L.dusp_abi_version.restype = ci
    L.dusp_device_count.restype = ci
    L.dusp_last_error.restype = ctypes.c_char_p
    L.dusp_last_error.argtypes = [vp]
    L.dusp_ctx_create.argtypes = [ci, ctypes.POINTER(vp)]
    L.dusp_ctx_destroy.argtypes = [vp]
    L.dusp_ctx_destroy.restype = None
    L.dusp_table_upload.argtypes = [vp, ci, vp, sz]
    L.dusp_program_build.argtypes = [vp, vp, sz, ci, ctypes.POINTER(vp)]
    L.dusp_program_destroy.argtypes = [vp]
    L.dusp_program_destroy.restype = None
    L.dusp_program_continue.argtypes = [vp, vp, sz]
    L.dusp_program_info_get.argtypes = [vp, ctypes.POINTER(ProgramInfo)]
    L.dusp_render_device.argtypes = [vp, sz, sz, vp, vp, vp]
    L.dusp_render_host.argtypes = [vp, sz, sz, vp, vp]
    L.dusp_render_host_interleaved.argtypes = [vp, sz, sz, vp, vp]
    L.dusp_interleave_device.argtypes = [vp, vp, sz, sz, sz, vp, vp]
    L.dusp_state_download.argtypes = [vp, sz, sz, vp, sz]
    L.dusp_last_kernel_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    L.dusp_fill_device.argtypes = [vp, vp, sz, ctypes.c_float, vp]
    L.dusp_render_device_inputs.argtypes = [vp, sz, sz, vp, vp, vp, vp]
    L.dusp_render_chain_window.argtypes = [vp, ctypes.c_uint64, sz, vp, ci, vp, vp]
    L.dusp_render_host_inputs.argtypes = [vp, sz, sz, vp, vp, vp, ci]
    L.dusp_host_alloc.argtypes = [vp, sz, ctypes.POINTER(vp)]
    L.dusp_host_free.argtypes = [vp, vp]
    L.dusp_jit_cache_dir.restype = ctypes.c_char_p
    L.dusp_jit_cache_dir.argtypes = []
    L.dusp_circuit_kernel_source.argtypes = [vp, sz, ci, ci, ci, ci, ctypes.c_char_p, sz]
    _lib = L
    return L


PINNED_MIN_BYTES = 1 << 20  # results of at least 1 MiB are delivered in pinned memory (dusp_host_alloc): one DMA, no staging
# Page-locked memory is a machine-wide resource: results a caller keeps alive stay pinned, so beyond this many bytes in use per
# context new results are ordinary (pageable) numpy arrays again (the staged download).  `render(..., pinned=False)` opts out per call.
PINNED_MAX_LIVE_BYTES = 8 << 30


class _PinnedBlock:
    """Owner of one dusp_host_alloc buffer: handed back to the context's pool when the last array over it goes."""

    def __init__(self, ctx, ptr, nbytes):
        self.ctx, self.ptr, self.nbytes = ctx, ptr, nbytes

    def __del__(self):  # (whatever thread the garbage collector runs on: _host_release takes the context's lock)
        try:
            self.ctx._host_release(self.ptr, self.nbytes)
        except Exception:
            pass


def circuit_kernel_source(words, waves=16, per_wave=1, lds_table=True, compile=False, continued=False, lean_recurrence=False):
    """HIP text of the kernel the circuit compiler generates for a descriptor (dusp_circuit_kernel_source; needs no GPU).
    Raises DuspHipError(-2) for circuits that stay on the interpreter."""
    L = load()
    words = np.ascontiguousarray(words, dtype=np.float64)
    buf = ctypes.create_string_buffer(1 << 20)
    n = L.dusp_circuit_kernel_source(words.ctypes.data, words.size, waves, per_wave, int(bool(lds_table)) | (2 if continued else 0) | (4 if lean_recurrence else 0), int(compile), buf, len(buf))
    if n < 0:
        raise DuspHipError(n, L.dusp_last_error(None).decode())
    return buf.value.decode()


class Context:
    """One HIP device + stream + the uploaded wave tables."""

    def __init__(self, device=-1, sample_rate=None):
        self._L = load()
        h = ctypes.c_void_p()
        rc = self._L.dusp_ctx_create(device, ctypes.byref(h))
        if rc != 0:
            raise DuspHipError(rc, self._L.dusp_last_error(None).decode())
        self._h = h
        self._host_live = 0       # pinned result buffers some numpy array still looks at
        self._host_live_bytes = 0
        # pool calls may come from a finalizer on another thread — or on THIS one: the cyclic collector can run while the lock is held
        # (allocating Python objects inside host_empty) and collect a cycle that owns a pinned block, whose __del__ takes the lock again.
        # dusp_host_alloc / _free take only the library's own pool mutex, which no Python code runs under, so re-entry is safe.
        self._host_lock = threading.RLock()
        self._close_pending = False
        self.sample_rate = None
        if sample_rate is not None:
            self.upload_tables(sample_rate)

    def host_empty(self, shape, pinned=None):
        """float32 array for a render result (replaces `new TypedArray(lengthInSamples)`, renderChannelData.js:39).  Large
        results live in the context's pinned pool so that the download is a direct DMA; the block returns to the pool when
        the array (and every view of it) has been collected."""
        n = int(np.prod(shape))
        if pinned is None:
            with self._host_lock:
                pinned = n * 4 >= PINNED_MIN_BYTES and self._host_live_bytes + n * 4 <= PINNED_MAX_LIVE_BYTES
        if not pinned or n == 0:
            return np.empty(shape, dtype=np.float32)
        p = ctypes.c_void_p()
        with self._host_lock:
            self._check(self._L.dusp_host_alloc(self._h, n * 4, ctypes.byref(p)))
            self._host_live += 1
            self._host_live_bytes += n * 4
        buf = (ctypes.c_float * n).from_address(p.value)
        buf._dusp_owner = _PinnedBlock(self, p.value, n * 4)
        return np.frombuffer(buf, dtype=np.float32).reshape(shape)

    def _host_release(self, ptr, nbytes=0):
        with self._host_lock:
            self._host_live -= 1
            self._host_live_bytes -= nbytes
            close_now = False
            if self._h:
                self._L.dusp_host_free(self._h, ptr)
                close_now = self._close_pending and self._host_live == 0
        if close_now:
            self.close()

    def _check(self, rc):
        if rc < 0:
            raise DuspHipError(rc, self._L.dusp_last_error(self._h).decode())
        return rc

    def upload_tables(self, sample_rate, tables=None):
        """Compute (or take) the wave tables (ids 0-4) and Shape tables (ids 5-8) for this sample rate and hand
        them to the device."""
        for tid in range(N_TABLES):
            try:
                t = tables[tid] if tables is not None else make_table(tid, sample_rate)
            except ValueError:
                continue  # e.g. no triangle table at a sample rate not divisible by 4
            t = np.ascontiguousarray(t, dtype=np.float32)
            self._check(self._L.dusp_table_upload(self._h, tid, t.ctypes.data, t.size))
        self.sample_rate = sample_rate

    def build(self, words, engine=ENGINE_AUTO):
        return Program(self, words, engine)

    def interleave(self, d_planar, n_instances, n_channels, n_samples, d_out, stream=None):
        """Device pointers: planar f32 [instance][channel][sample] -> frames f32 [instance][sample][channel]."""
        self._check(self._L.dusp_interleave_device(self._h, d_planar, n_instances, n_channels, n_samples, d_out, stream))

    def fill(self, d_ptr, n_floats, value=0.0, stream=None):
        self._check(self._L.dusp_fill_device(self._h, d_ptr, n_floats, value, stream))

    def close(self):
        if self._h and self._host_live > 0:  # arrays over the pinned pool are still alive: the last one closes
            self._close_pending = True
            return
        if self._h:
            self._L.dusp_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Program:
    def __init__(self, ctx, words, engine=ENGINE_AUTO):
        self.ctx = ctx
        self._L = ctx._L
        words = np.ascontiguousarray(words, dtype=np.float64)
        h = ctypes.c_void_p()
        ctx._check(self._L.dusp_program_build(ctx._h, words.ctypes.data, words.size, engine, ctypes.byref(h)))
        self._h = h
        self._read_info()

    def _read_info(self):
        info = ProgramInfo()
        self.ctx._check(self._L.dusp_program_info_get(self._h, ctypes.byref(info)))
        self.sample_rate = info.sample_rate
        self.n_units = info.n_units
        self.n_out_channels = info.n_out_channels
        self.n_params = info.n_params
        self.n_inputs = info.n_inputs
        self.engine = ENGINE_NAMES.get(info.engine, str(info.engine))
        self.shape = info.shape.decode()
        self.n_device_ops = info.n_device_ops

    def read_shape(self):
        """The program's shape string as of now (after a render it names the kernel that ran: "compiled kernel: ..." or not)."""
        self._read_info()
        return self.shape

    def continue_with(self, words):
        """dusp_program_continue: re-arm this rendered program from a later extraction of the same circuit
        (unit state and constants from `words`, delay lines / feedback chunks stay on the device)."""
        words = np.ascontiguousarray(words, dtype=np.float64)
        self.ctx._check(self._L.dusp_program_continue(self._h, words.ctypes.data, words.size))
        self._read_info()

    def render(self, n_samples, n_instances=1, params=None, interleaved=False, inputs=None, pinned=None):
        """Host round trip: float32 [n_instances, n_out_channels, n_samples], or — interleaved — frames
        [n_instances, n_samples, n_out_channels] (the RenderStream / WAV layout, transposed on the device).
        inputs: the host-generated streams of this call, float32 [n_inputs, n_instances, n_samples] (programs with INPUT units)."""
        shape = (n_instances, n_samples, self.n_out_channels) if interleaved else (n_instances, self.n_out_channels, n_samples)
        out = self.ctx.host_empty(shape, pinned)  # pinned=None: by size; False: pageable memory (the staged download)
        pp = None
        if self.n_params:
            params = np.ascontiguousarray(params, dtype=np.float32)
            if params.shape != (self.n_params, n_instances):
                raise ValueError("params must have shape (n_params=%d, n_instances=%d)" % (self.n_params, n_instances))
            pp = params.ctypes.data
        if self.n_inputs:
            inputs = np.ascontiguousarray(inputs, dtype=np.float32)
            if inputs.shape != (self.n_inputs, n_instances, n_samples):
                raise ValueError("inputs must have shape (n_inputs=%d, n_instances=%d, n_samples=%d)" % (self.n_inputs, n_instances, n_samples))
            self.ctx._check(self._L.dusp_render_host_inputs(self._h, n_instances, n_samples, pp, inputs.ctypes.data, out.ctypes.data, int(interleaved)))
            return out
        call = self._L.dusp_render_host_interleaved if interleaved else self._L.dusp_render_host
        self.ctx._check(call(self._h, n_instances, n_samples, pp, out.ctypes.data))
        return out

    def render_device(self, n_samples, n_instances, d_params, d_out, stream=None, d_inputs=None):
        """Asynchronous render between device pointers (ints), e.g. torch tensors' data_ptr()."""
        if d_inputs is not None or self.n_inputs:
            self.ctx._check(self._L.dusp_render_device_inputs(self._h, n_instances, n_samples, d_params, d_inputs, d_out, stream))
        else:
            self.ctx._check(self._L.dusp_render_device(self._h, n_instances, n_samples, d_params, d_out, stream))

    def render_chain_window(self, first_sample, n_samples, d_init, raw, d_out, stream=None):
        """dusp_render_chain_window: the window [first_sample, first_sample + n_samples) of a fused sum-chain program's timeline, its sums
        continued from d_init (None: the chain's first voices are this program's); raw: a partial sum for the next rank (shard.chain_mixdown)."""
        self.ctx._check(self._L.dusp_render_chain_window(self._h, first_sample, n_samples, d_init, int(bool(raw)), d_out, stream))

    def state(self, unit, instance=0):
        buf = np.zeros(128, dtype=np.float64)
        n = self.ctx._check(self._L.dusp_state_download(self._h, instance, unit, buf.ctypes.data, buf.size))
        return buf[:n].copy()

    def last_kernel_ms(self):
        ms = ctypes.c_float()
        self.ctx._check(self._L.dusp_last_kernel_ms(self._h, ctypes.byref(ms)))
        return ms.value

    def close(self):
        if self._h:
            self._L.dusp_program_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
