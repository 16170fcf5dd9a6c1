"""GPU parity: the HIP path (through the C ABI) against the reference's golden vectors and the oracle.

Bar: bit-exact wherever the device executes the same IEEE operations as JS (everything except the
Filter's tan(), which comes from the device math library): there the north star's tolerance applies,
1e-5 relative to full scale (SURVEY.md §7 "tolerance definition").
"""
import numpy as np
import pytest

from conftest import ALL_GOLDEN, Golden
from dusp_amd import render, runtime

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5  # north_star: "within 1e-5 relative float tolerance" (of full scale)
USES_DEVICE_TAN = ("loop_", "filter_")


def check(name, got, ref):
    assert got.shape == ref.shape
    if name.startswith(USES_DEVICE_TAN):
        scale = max(1e-30, float(np.max(np.abs(ref))))
        err = float(np.max(np.abs(got.astype(np.float64) - ref.astype(np.float64))))
        assert err <= REL_TOL * scale, "max abs err %.3g vs full scale %.3g" % (err, scale)
    else:
        assert np.array_equal(got, ref), "first mismatch at %d" % int(np.argmax(got != ref))


@pytest.mark.parametrize("engine", ["auto", "chunk"])
@pytest.mark.parametrize("name", ALL_GOLDEN)
def test_render_matches_reference_golden(name, engine, oracle):
    g = Golden(name)
    ctx = render.context(g.sample_rate)
    prog = ctx.build(g.desc, runtime.ENGINE_AUTO if engine == "auto" else runtime.ENGINE_CHUNK)
    assert prog.n_out_channels == g.n_channels
    pcm = prog.render(g.n_samples)[0]
    check(name, g.windowed(pcm), g.pcm)            # vs the JS reference's own output
    check(name, pcm, oracle.render(g.desc, g.n_samples))  # vs the oracle, full length
    prog.close()


@pytest.mark.parametrize("name", ["osc_f_440p5", "voice3_k7", "ramp_300", "loop_220", "circlebuffer_taps", "cfg2_sweep"])
@pytest.mark.parametrize("engine", ["auto", "chunk"])
def test_state_write_back_matches_oracle(name, engine, oracle):
    g = Golden(name)
    n = min(g.n_samples, 5000)
    prog = render.context(g.sample_rate).build(g.desc, runtime.ENGINE_AUTO if engine == "auto" else runtime.ENGINE_CHUNK)
    prog.render(n)
    _, states = oracle.render(g.desc, n, return_state=True)
    for u, want in enumerate(states):
        got = prog.state(u)
        assert got.size == want.size
        if name.startswith(USES_DEVICE_TAN):
            np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-9, equal_nan=True)
        else:
            assert np.array_equal(got, want, equal_nan=True), (u, got, want)
    prog.close()
