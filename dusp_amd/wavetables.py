"""Oscillator wave tables, computed on the host exactly as the reference does
(src/components/Osc/waveTables.js:5-40) and uploaded to the device as data.

`math.sin` (libm) reproduces V8's Math.sin on every entry of the 44.1 k and
48 k tables after the f32 store — tests/test_wavetables.py checks the sha256
of all five tables against hashes captured from the reference.
"""
import math

import numpy as np

WAVEFORMS = {"sin": 0, "sine": 0, "saw": 1, "square": 2, "triangle": 3, "8bit": 4}
TABLE_NAMES = ["sin", "saw", "square", "triangle", "8bit", "decay", "attack", "semiSine", "decaySquared"]
SHAPES = {"decay": 5, "attack": 6, "semiSine": 7, "decaySquared": 8}  # Shape's tables share the table space (ids 5-8)
N_TABLES = 9


def _js_round(x):
    r = math.floor(x + 0.5)
    return -0.0 if (r == 0 and (x < 0 or math.copysign(1.0, x) < 0)) else float(r)


def make_table(table_id, sample_rate):
    n = sample_rate + 1
    phi = 2 * math.pi
    out = np.zeros(n, dtype=np.float32)
    if table_id in (0, 4):  # sine; period is the table LENGTH (sr+1), not sr
        out[:] = [math.sin(phi * t / n) for t in range(n)]
        if table_id == 4:   # "8bit": round(sin * 128) / 128 on the f32 sine
            out[:] = [_js_round(float(v) * 128.0) / 128.0 for v in out]
    elif table_id == 1:     # saw; the loop stops at sr so the last entry stays 0
        out[:sample_rate] = [-1 + t * 2 / n for t in range(sample_rate)]
    elif table_id == 2:     # square
        if sample_rate % 2:
            raise ValueError("square table needs an even sample rate")
        out[: sample_rate // 2] = 1
        out[sample_rate // 2:] = -1
    elif table_id == 3:     # triangle; later quarters re-read the f32-rounded first quarter
        if sample_rate % 4:
            raise ValueError("triangle table needs a sample rate divisible by 4")
        q = sample_rate // 4
        first = np.array([t / sample_rate * 4 for t in range(q)], dtype=np.float32)
        out[:q] = first
        out[q:2 * q] = (1 - first.astype(np.float64)).astype(np.float32)
        out[2 * q:3 * q] = -first
        out[3 * q:4 * q] = (-1 + first.astype(np.float64)).astype(np.float32)
        out[sample_rate] = 0
    elif 5 <= table_id <= 8:  # Shape tables: func(x / sampleRate), x = 0..sampleRate (Shape/shapeTables.js:3-38)
        func = {5: lambda x: 1 - x, 6: lambda x: x, 7: lambda x: math.sin(math.pi * x), 8: lambda x: (1 - x) * (1 - x)}[table_id]
        out[:] = [func(x / sample_rate) for x in range(n)]
    else:
        raise ValueError("no such wave table: %r" % (table_id,))
    return out
