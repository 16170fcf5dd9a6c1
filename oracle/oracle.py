"""TEST INFRASTRUCTURE — ctypes wrapper around the CPU oracle (oracle/dusp_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module; the product package (dusp_amd/) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("DUSP_ORACLE_LIB") or os.path.join(_HERE, "libdusp_oracle.so")  # (DUSP_ORACLE_LIB: the sanitizer build, tests/test_sanitizers.py)
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "dusp_oracle.c")
    if os.environ.get("DUSP_ORACLE_LIB"):
        return _LIB_PATH  # (built by whoever named it)
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libdusp_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.dusp_oracle_create.restype = ctypes.c_void_p
        L.dusp_oracle_create.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                         ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
        L.dusp_oracle_destroy.argtypes = [ctypes.c_void_p]
        L.dusp_oracle_render.restype = ctypes.c_int
        L.dusp_oracle_render.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
        L.dusp_oracle_unit_state.restype = ctypes.c_size_t
        L.dusp_oracle_unit_state.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
        L.dusp_oracle_n_units.restype = ctypes.c_size_t
        L.dusp_oracle_n_units.argtypes = [ctypes.c_void_p]
        L.dusp_oracle_set_inputs.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        L.dusp_oracle_set_inputs.restype = None
        L.dusp_oracle_wavetable.restype = ctypes.c_int
        L.dusp_oracle_wavetable.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


def render(desc, n_samples, params=None, n_instances=1, instance=0, max_channels=8, return_state=False, inputs=None):
    """Render ONE instance of a descriptor on the CPU oracle -> float32 [n_channels, n_samples].
    inputs: this instance's host-generated streams, float32 [n_streams, n_samples] (descriptors with INPUT units)."""
    L = lib()
    desc = np.ascontiguousarray(desc, dtype=np.float64)
    pp = None
    if params is not None:
        params = np.ascontiguousarray(params, dtype=np.float32)
        pp = params.ctypes.data
    err = ctypes.create_string_buffer(256)
    h = L.dusp_oracle_create(desc.ctypes.data, desc.size, pp, n_instances, instance, err, len(err))
    if not h:
        raise OracleError(err.value.decode())
    try:
        if inputs is not None:
            inputs = np.ascontiguousarray(inputs, dtype=np.float32)
            L.dusp_oracle_set_inputs(h, inputs.ctypes.data, inputs.shape[1])
        out = np.zeros((max_channels, int(n_samples)), dtype=np.float32)
        nch = L.dusp_oracle_render(h, int(n_samples), out.ctypes.data, max_channels)
        if nch > max_channels:
            raise OracleError("output has %d channels > max_channels=%d" % (nch, max_channels))
        result = out[:nch].copy()
        if not return_state:
            return result
        states = []
        for u in range(L.dusp_oracle_n_units(h)):
            buf = np.zeros(64, dtype=np.float64)
            n = L.dusp_oracle_unit_state(h, u, buf.ctypes.data, buf.size)
            states.append(buf[:n].copy())
        return result, states
    finally:
        L.dusp_oracle_destroy(h)


def wavetable(table_id, sample_rate):
    out = np.zeros(sample_rate + 1, dtype=np.float32)
    if lib().dusp_oracle_wavetable(table_id, sample_rate, out.ctypes.data) != 0:
        raise OracleError("wavetable %d not defined at sample rate %d" % (table_id, sample_rate))
    return out
