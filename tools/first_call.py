#!/usr/bin/env python3
"""First render of a circuit in a NEW process, with and without its kernel in the code-object cache (jit_engine.hip).
The circuit is the README's `[Osc f:[Osc 5] * 100 + 440] * D2` (FM voice under a decay envelope), 10 s at 48 kHz, through the host
surface (upload + render + download).  Process 1 starts from an empty cache directory and waits for the compile (DUSP_WAVE_JIT=2);
process 2 finds the code object on disk.  Prints one JSON line: first-call and second-call milliseconds of both processes.
  python tools/first_call.py [cache_dir]"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT)
    import numpy as np
    import dusp_amd as d
    from dusp_amd import descriptor, runtime
    sr = 48000
    d.configure(sr)
    ctx = runtime.Context(0, sr)
    warm = ctx.build(descriptor.extract(d.Osc(1)).words)   # (the context's own first launch: not what is measured)
    warm.render(256)
    voice = d.Multiply(d.Osc(d.Sum(d.Multiply(d.Osc(5), 100), 440)), d.Shape("decay", 2).trigger())
    ex = descriptor.extract(voice)
    t0 = time.perf_counter()
    prog = ctx.build(ex.words)
    a = prog.render(10 * sr)
    t1 = time.perf_counter()
    b = prog.render(10 * sr)
    t2 = time.perf_counter()
    assert np.array_equal(a, b) and float(np.abs(a).max()) > 0.1
    print(json.dumps({"first_ms": round((t1 - t0) * 1e3, 2), "again_ms": round((t2 - t1) * 1e3, 2), "shape": prog.read_shape(),
                      "cache": runtime.load().dusp_jit_cache_dir().decode()}))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child()
    cache = sys.argv[1] if len(sys.argv) > 1 else tempfile.mkdtemp(prefix="dusp_cache_")
    out = {}
    for tag, jit in (("empty_cache_waiting_for_the_compile", "2"), ("second_process_default_knob", "1")):
        env = dict(os.environ, DUSP_JIT_CACHE=cache, DUSP_WAVE_JIT=jit)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], capture_output=True, text=True, env=env, timeout=600)
        if r.returncode != 0:
            raise SystemExit(r.stderr[-3000:])
        out[tag] = json.loads(r.stdout.strip().splitlines()[-1])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
