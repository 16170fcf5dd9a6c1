"""Compiled circuit kernels at every workgroup geometry x ragged batch sizes (regression test asked for after an unexplained
abort in round 2's records): instances per wavefront R in {1, 2, 4}, batches that leave the last wavefront and
the last workgroup partly empty (1, 17, 63, 65, 16383 instances), a light circuit and a Filter circuit, EVERY instance
against the oracle.  The library runs with DUSP_GUARD=1 here (conftest): a kernel that writes past the end of a workspace
fails its render instead of corrupting a neighbour."""
import numpy as np
import pytest

import dusp_amd as d
from conftest import knob_context
from dusp_amd import descriptor, runtime

pytestmark = pytest.mark.gpu


def _light(k):      # Multiply(Osc(f_k), gain_k): two lean units (jit_light) — per-instance f and gain
    return d.Multiply(d.Osc(100.0 + 0.37 * k), 0.25 + (k % 7) / 16.0)


def _filtered(k):   # Filter(Osc(f_k) + Osc(3), cutoff_k) * gain: the workgroup-wide Filter stage with per-instance rows
    return d.Multiply(d.Filter(d.Sum(d.Osc(110.0 + k / 4.0), d.Osc(3.0)), 500.0 + 11.0 * (k % 90)), 0.5)


@pytest.mark.parametrize("per_wave", [1, 2, 4])
@pytest.mark.parametrize("n_inst", [1, 17, 63, 65, 16383])
@pytest.mark.parametrize("kind", ["light", "filter"])
def test_every_geometry_and_ragged_batch_against_the_oracle(kind, n_inst, per_wave, oracle):
    d.configure(48000)
    make = _light if kind == "light" else _filtered
    # structure from three circuits whose parameters all differ; the columns are then written directly (every parameter of these
    # voices is a known function of k)
    uni = descriptor.unify([descriptor.extract(make(k)) for k in (0, 1, 9)])
    k = np.arange(n_inst, dtype=np.float64)
    cols = []
    for p in range(uni.n_params):
        v0, v1, v9 = (float(uni.params[p, j]) for j in range(3))
        if kind == "light":
            cols.append((100.0 + 0.37 * k) if v1 - v0 > 0.3 else (0.25 + (k % 7) / 16.0))
        else:
            cols.append((110.0 + k / 4.0) if abs((v1 - v0) - 0.25) < 1e-6 else (500.0 + 11.0 * (k % 90)))
    params = np.ascontiguousarray(np.stack(cols).astype(np.float32))
    if n_inst > 9:  # the columns written above are the ones the circuits themselves carry
        assert np.array_equal(params[:, [0, 1, 9]], uni.params), (params[:, [0, 1, 9]], uni.params)
    waves = 16 if n_inst > 64 else 4
    ctx = knob_context(48000, DUSP_JIT_FORCE="%dx%d" % (waves, per_wave))
    prog = ctx.build(uni.words, runtime.ENGINE_WAVE)
    n = 256 * 3 + 40 if n_inst > 1000 else 256 * 9 + 40
    pcm = prog.render(n, n_inst, params)
    shape = prog.read_shape()
    assert "compiled kernel" in shape and shape.endswith("%dx%d" % (waves, per_wave)), shape
    worst = 0.0
    wants = oracle.render_instances(uni.words, n, params, n_inst, range(n_inst), max_channels=1)  # (every instance, over the host's cores)
    for i, want in enumerate(wants):
        if kind == "light":
            assert np.array_equal(pcm[i], want), "instance %d of %d at %dx%d" % (i, n_inst, waves, per_wave)
        else:
            worst = max(worst, float(np.max(np.abs(pcm[i].astype(np.float64) - want))))
    assert worst <= 1e-5, worst  # (Filter: device tan(), north star's tolerance; full scale here is ~1)
    prog.close()
