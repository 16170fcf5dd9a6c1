import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# Contexts of the test processes (and of the node processes they start) WAIT for a circuit's compiled kernel, so that every GPU
# test that can runs on one; the product default lets a short first render run on the interpreter kernel while the circuit
# compiles in the background (tests/test_gpu_batch.py::test_first_render_does_not_wait_for_the_compiler covers that).
os.environ.setdefault("DUSP_WAVE_JIT", "2")
os.environ.setdefault("DUSP_GUARD", "1")  # guard bytes behind every device workspace, checked after each render (dusp_abi.hip)
# The code-object cache of this test session: a fresh private directory (the library reads DUSP_JIT_CACHE once per process, before the
# first context; node and python child processes inherit it).  Nothing lands in $HOME, and what a test finds compiled does not depend on
# what the box rendered before: the tests of the product default DUSP_WAVE_JIT=1 (first render on the interpreter while the kernel
# compiles) always start from a structure nobody has compiled.
if "DUSP_JIT_CACHE" not in os.environ:
    import atexit
    import shutil
    import tempfile
    _session_cache = tempfile.mkdtemp(prefix="dusp_jit_cache_")
    os.environ["DUSP_JIT_CACHE"] = _session_cache
    atexit.register(shutil.rmtree, _session_cache, True)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names(sample_rate=48000):
    name = "index.json" if sample_rate == 48000 else "index_sr%d.json" % sample_rate
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


class Golden:
    """One reference-generated vector: descriptor (input) + PCM windows (expected output)."""

    def __init__(self, name):
        self.name = name
        with open(os.path.join(GOLDEN, name + ".json")) as f:
            self.meta = json.load(f)
        self.desc = np.fromfile(os.path.join(GOLDEN, name + ".desc.f64"), dtype=np.float64)
        self.pcm = np.fromfile(os.path.join(GOLDEN, name + ".pcm.f32"), dtype=np.float32)
        self.n_samples = self.meta["n_samples"]
        self.n_channels = self.meta["n_channels"]
        self.windows = [tuple(w) for w in self.meta["windows"]]
        self.sample_rate = self.meta["sample_rate"]

    def windowed(self, pcm):
        """Cut [n_channels, n_samples] PCM down to the stored windows (same order as the file)."""
        parts = []
        for c in range(self.n_channels):
            for a, n in self.windows:
                parts.append(pcm[c, a:a + n])
        return np.concatenate(parts)


ALL_GOLDEN = golden_names(48000) + golden_names(44100)


class _Oracle:
    """The oracle module, with the last few plain full-length renders kept: the engine columns of one golden case (auto, chunk, wave,
    loop, interp run one after the other) compare against the SAME oracle render instead of five."""

    def __init__(self, module):
        self._m = module
        self._kept = []  # [(key, pcm)], newest last

    def __getattr__(self, name):
        return getattr(self._m, name)

    def render(self, desc, n_samples, **kw):
        if kw:
            return self._m.render(desc, n_samples, **kw)
        words = np.ascontiguousarray(desc, dtype=np.float64)
        key = (words.tobytes(), int(n_samples))
        for k, pcm in self._kept:
            if k == key:
                return pcm
        pcm = self._m.render(words, n_samples)
        pcm.setflags(write=False)
        self._kept = self._kept[-3:] + [(key, pcm)]
        return pcm

    def render_instances(self, desc, n_samples, params, n_instances, instances, threads=None, **kw):
        """[oracle.render(..., instance=i) for i in instances], spread over the host cores this process may use (instances are independent; the
        C call drops the GIL)."""
        from concurrent.futures import ThreadPoolExecutor
        try:
            usable = len(os.sched_getaffinity(0))
        except AttributeError:
            usable = os.cpu_count() or 1
        threads = threads or max(1, min(16, usable))
        one = lambda i: self._m.render(desc, n_samples, params=params, n_instances=n_instances, instance=int(i), **kw)
        with ThreadPoolExecutor(threads) as pool:
            return list(pool.map(one, instances))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return _Oracle(o)


_knob_contexts = {}


def knob_context(sample_rate, **knobs):
    """A context created under the given DUSP_* A/B knobs (the library reads them ONCE, in dusp_ctx_create); cached per setting."""
    from dusp_amd import runtime
    key = (sample_rate,) + tuple(sorted(knobs.items()))
    if key not in _knob_contexts:
        saved = {k: os.environ.get(k) for k in knobs}
        os.environ.update({k: str(v) for k, v in knobs.items()})
        try:
            _knob_contexts[key] = runtime.Context(-1, sample_rate)
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    return _knob_contexts[key]
