"""Host-generated input streams (descriptor opcode INPUT, dusp_render_*_inputs): a unit whose signal the HOST computes —
the JS host's Noise draws its Math.random() numbers this way — read by the rest of the circuit on the device.  GPU
against the oracle, which copies the same streams (oracle/dusp_oracle.c OP_INPUT), on both general engines."""
import numpy as np
import pytest

import dusp_amd as d
from dusp_amd import descriptor, render, runtime

pytestmark = pytest.mark.gpu

SR = 48000


def graphs(x):
    src = lambda: d.HostSource(x)  # noqa: E731

    def feedback():
        s = d.Sum(src(), 0)
        dl = d.Delay(s, 300.25, 2048)
        s.B = d.Multiply(d.Filter(dl, 3000), 0.5)
        return dl

    def short_delay():
        return d.Delay(src(), d.Sum(d.Multiply(d.Osc(3), 20), 40), 1024)

    return {
        "bare": src,
        "times_osc": lambda: d.Multiply(src(), d.Osc(440)),
        "fm": lambda: d.Osc(d.Sum(d.Multiply(src(), 300), 440)),
        "filter": lambda: d.Filter(src(), 1200),
        "feedback_delay_filter": feedback,
        "modulated_delay": short_delay,
        "two_sources": lambda: d.Sum(d.Multiply(src(), 0.5), d.Delay(d.HostSource(x[::-1].copy()), 256, 1024)),
    }


@pytest.mark.parametrize("engine", ["chunk", "wave", "auto"])
@pytest.mark.parametrize("name", list(graphs(np.zeros(1, dtype=np.float32))))
def test_render_with_host_inputs_matches_oracle(name, engine, oracle):
    d.configure(SR)
    n = 256 * 9 + 100
    x = np.random.RandomState(7).uniform(-1, 1, n).astype(np.float32)
    ex = descriptor.extract(graphs(x)[name]())
    assert ex.sources
    streams = np.stack([s.samples[:n] for s in ex.sources])
    want = oracle.render(ex.words, n, inputs=streams)
    prog = render.context(SR).build(ex.words, {"chunk": runtime.ENGINE_CHUNK, "wave": runtime.ENGINE_WAVE, "auto": runtime.ENGINE_AUTO}[engine])
    assert prog.n_inputs == len(ex.sources)
    got = prog.render(n, 1, inputs=streams[:, None, :])[0]
    prog.close()
    if "filter" in name:  # device tan() in the coefficients
        assert float(np.max(np.abs(got - want))) <= 1e-5 * max(1e-30, float(np.max(np.abs(want))))
    else:
        assert np.array_equal(got, want)


def test_batch_every_instance_reads_its_own_streams(oracle):
    d.configure(SR)
    n, V = 256 * 4 + 33, 70
    rng = np.random.RandomState(3)
    x = rng.uniform(-1, 1, (2, V, n)).astype(np.float32)
    exs = [descriptor.extract(d.Sum(d.Multiply(d.HostSource(x[0, 0]), d.Osc(100 + k)), d.Delay(d.HostSource(x[1, 0]), 300, 1024))) for k in range(3)]
    uni = descriptor.unify(exs)
    base, step = uni.params[:, 0].astype(np.float64), uni.params[:, 1].astype(np.float64) - uni.params[:, 0].astype(np.float64)
    params = (base[:, None] + step[:, None] * np.arange(V)[None, :]).astype(np.float32)
    outs = []
    for engine in (runtime.ENGINE_CHUNK, runtime.ENGINE_WAVE):
        prog = render.context(SR).build(uni.words, engine)
        outs.append(prog.render(n, V, params, inputs=x))
        prog.close()
    assert np.array_equal(outs[0], outs[1])
    for v in (0, 1, 63, 64, V - 1):
        want = oracle.render(uni.words, n, params=params, n_instances=V, instance=v, inputs=x[:, v, :])
        assert np.array_equal(outs[0][v], want), v


def test_render_channel_data_segments_carry_the_streams():
    """An event forces segments: the source's samples are sliced per segment; delay lines survive the boundary."""
    d.configure(SR)
    n = 256 * 8
    x = np.random.RandomState(5).uniform(-1, 1, n).astype(np.float32)

    def build(with_event):
        src = d.HostSource(x)
        gain = d.Multiply(src, 1)
        out = d.Delay(gain, 300, 2048)
        if with_event:
            gain.schedule(0.011, lambda unit: setattr(unit, "B", 0.5))
        return out

    seg = np.asarray(d.renderChannelData(build(True), n / SR)[0])
    whole = np.asarray(d.renderChannelData(build(False), n / SR)[0])
    cut = 512 + 300  # the event lands in the chunk that starts at sample 512 (0.011 s = sample 528); the delay moves it by 300
    assert np.array_equal(seg[:cut], whole[:cut])
    assert np.array_equal(seg[cut:], (whole[cut:].astype(np.float64) * 0.5).astype(np.float32))


def test_plain_render_calls_refuse_a_program_with_inputs():
    d.configure(SR)
    ex = descriptor.extract(d.Multiply(d.HostSource(np.zeros(8, dtype=np.float32)), 2))
    prog = render.context(SR).build(ex.words)
    out = np.empty((1, 1, 256), dtype=np.float32)
    rc = prog._L.dusp_render_host(prog._h, 1, 256, None, out.ctypes.data)
    assert rc == -1 and b"input streams" in prog._L.dusp_last_error(prog.ctx._h)
    with pytest.raises(ValueError):
        prog.render(256, 1, inputs=np.zeros((1, 1, 255), dtype=np.float32))
    prog.close()
