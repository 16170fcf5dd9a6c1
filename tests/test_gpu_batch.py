"""GPU parity for batched renders (many instances of one program) and full-size properties."""
import numpy as np
import pytest

import dusp_amd as d
from conftest import knob_context
from dusp_amd import descriptor, render, runtime

pytestmark = pytest.mark.gpu


def voices(fs, ramp_len, sr=48000):
    d.configure(sr)
    return descriptor.unify([descriptor.extract(d.Multiply(d.Osc(float(f)), d.Ramp(ramp_len, 1, 0).trigger())) for f in fs])


@pytest.mark.parametrize("engine", [runtime.ENGINE_FUSED, runtime.ENGINE_CHUNK])
@pytest.mark.parametrize("fs", [
    [10.0 * k for k in range(1, 71)],                                   # integer f: INT path, 70 voices = ragged last block
    [20 + k / 8 for k in range(0, 4000, 61)],                           # configs[4] sweep: FX32 path
    [3e-5, 0.1, -0.37, 47999.5, 1e5, -7e4, 440.5, 12345.678, 2.0 ** -20, 0.0 + 1e-30],  # tiny / negative / > sr: F64 path
    [261.6255653, 1e-3, 5.0, 5.5, 24000.0, 23999.999],                  # mixed blocks
])
def test_voice_batches_match_oracle(engine, fs, oracle):
    n = 3000  # not a multiple of 256: exercises the guarded tail
    uni = voices(fs, 2500)
    prog = render.context(48000).build(uni.words, engine)
    pcm = prog.render(n, uni.n_instances, uni.params)
    exact_regime = all(f == 0 or abs(np.float32(f)) >= 2.0 ** -13 for f in fs)
    for i in range(uni.n_instances):
        want = oracle.render(uni.words, n, params=uni.params, n_instances=uni.n_instances, instance=i)
        if engine == runtime.ENGINE_CHUNK or exact_regime:
            assert np.array_equal(pcm[i], want), "voice %d (f=%r)" % (i, fs[i])
        else:  # |f| < 2^-13: the reference itself rounds while accumulating; closed form is within tolerance
            assert np.max(np.abs(pcm[i].astype(np.float64) - want)) <= 1e-5
    prog.close()


def test_nonfinite_parameter_renders_zeros(oracle):
    uni = voices([440.0, 441.0], 1000)
    uni.params[0, 1] = np.inf
    prog = render.context(48000).build(uni.words)
    pcm = prog.render(1024, 2, uni.params)
    want = oracle.render(uni.words, 1024, params=uni.params, n_instances=2, instance=1)
    assert np.array_equal(pcm[1], want) and not pcm[1].any()  # NaN phase -> NaN sample -> `|| 0`
    prog.close()


@pytest.mark.parametrize("engine", [runtime.ENGINE_AUTO, runtime.ENGINE_CHUNK])
def test_feedback_loop_batch(engine, oracle):
    """BASELINE configs[3] at small size: instances of the Osc->Sum->Delay->Filter->Multiply loop."""
    d.configure(48000)
    def loop(k):
        s = d.Sum(d.Osc(110 + k / 64), 0)
        f = d.Filter(d.Delay(s, 480, 4096), 2000)
        s.B = d.Multiply(f, 0.5)
        return f
    uni = descriptor.unify([descriptor.extract(loop(k)) for k in range(0, 8192, 97)])
    n = 2000
    prog = render.context(48000).build(uni.words, engine)
    assert prog.engine == ("wave" if engine == runtime.ENGINE_AUTO else "chunk")  # (AUTO: a kernel compiled for the circuit)
    pcm = prog.render(n, uni.n_instances, uni.params)
    for i in range(0, uni.n_instances, 7):
        want = oracle.render(uni.words, n, params=uni.params, n_instances=uni.n_instances, instance=i)
        assert np.max(np.abs(pcm[i].astype(np.float64) - want)) <= 1e-5 * np.max(np.abs(want))
    prog.close()


@pytest.mark.parametrize("regime,delay_of,expect", [
    ("long", lambda k: 300 + k % 400, "compiled kernel"),                   # every instance >= one chunk: the write-once ring, per instance
    ("long_frac", lambda k: 256 + (k % 97) * 3.25, "compiled kernel"),      # ... with fractions, some rings not 16-byte friendly
    ("short", lambda k: 1 + (k % 250) + 0.5 * (k % 2), "compiled kernel"),  # every instance below a chunk: no ring
    ("mixed", lambda k: 40 + 7 * (k % 90), "compiled kernel"),              # both sides of a chunk: ordered slot operations
    ("edge", lambda k: 255.25 if k % 2 else 700, "compiled kernel"),        # within a sample of the chunk size: slot operations
])
def test_per_instance_delays_on_compiled_kernels(regime, delay_of, expect, oracle):
    """A delay-time sweep of the configs[3] voice (Delay.js:20-41 with a different constant per instance): the compiled kernel looks at
    the parameter column and takes the write-once ring / the ring-less form / the ordered slot operations; bit-equal to the chunk
    engine on every instance, within the Filter tolerance of the oracle on a few."""
    d.configure(48000)
    def loop(k):
        s = d.Sum(d.Osc(110 + k / 64), 0)
        f = d.Filter(d.Delay(s, delay_of(k), 4096), 2000)
        s.B = d.Multiply(f, 0.5)
        return f
    ks = list(range(0, 1200, 5)) + [1199]
    uni = descriptor.unify([descriptor.extract(loop(k)) for k in ks])
    assert uni.n_params == 2
    n = 256 * 14 + 100
    ctx = knob_context(48000, DUSP_FILTER_SCAN=0)  # (the Filter through its stage: the chunk engine's bits; the Delay is what this test is about)
    prog = ctx.build(uni.words)  # AUTO
    assert prog.engine == "wave", prog.engine
    pcm = prog.render(n, uni.n_instances, uni.params)
    assert expect in prog.read_shape(), prog.read_shape()
    ref = ctx.build(uni.words, runtime.ENGINE_CHUNK)
    want = ref.render(n, uni.n_instances, uni.params)
    assert np.array_equal(pcm, want), "instance %d differs from the chunk engine" % int(np.argmax((pcm != want).any(axis=(1, 2))))
    scan = render.context(48000).build(uni.words)  # the default: the 2 kHz Filter as a scan over the chunk, next to every kind of Delay
    got = scan.render(n, uni.n_instances, uni.params)
    assert expect in scan.read_shape() and float(np.max(np.abs(got.astype(np.float64) - want))) <= 4e-6 * float(np.max(np.abs(want)))
    scan.close()
    for i in (0, 1, 17, uni.n_instances - 1):
        w = oracle.render(uni.words, n, params=uni.params, n_instances=uni.n_instances, instance=i)
        assert np.max(np.abs(pcm[i].astype(np.float64) - w)) <= 1e-5 * max(1.0, float(np.max(np.abs(w)))), (regime, i)
    # the same program again with another column: the regime is looked at per render
    other = uni.params.copy()
    other[:] = uni.params[:, ::-1]
    pcm2 = prog.render(n, uni.n_instances, other)
    assert np.array_equal(pcm2, ref.render(n, uni.n_instances, other))
    prog.close()
    ref.close()


def test_per_instance_mono_delays_on_compiled_kernels(oracle):
    """MonoDelay with a per-instance delay (the Space patch's distance -> delay at batch scale): long, short and mixed columns."""
    d.configure(48000)
    for delays in ([300 + 3.5 * k for k in range(70)], [0.25 + 3.5 * k for k in range(70)], [100 + 9.75 * k for k in range(70)]):
        uni = descriptor.unify([descriptor.extract(d.MonoDelay(d.Osc(200 + 3 * k), dl)) for k, dl in enumerate(delays)])
        n = 256 * 6 + 17
        ctx = render.context(48000)
        prog = ctx.build(uni.words, runtime.ENGINE_WAVE)
        pcm = prog.render(n, uni.n_instances, uni.params)
        assert "compiled kernel" in prog.read_shape()
        for i in range(0, uni.n_instances, 3):
            want = oracle.render(uni.words, n, params=uni.params, n_instances=uni.n_instances, instance=i)
            assert np.array_equal(pcm[i], want), (delays[i], i)
        prog.close()


def _filter_voice(kind, k):
    o = d.Osc(110 + k / 4)
    if kind == "env_after":      # a unit hanging on the Filter, a per-instance cutoff
        return d.Multiply(d.Filter(o, 600 + k), d.Ramp(2500, 1, 0).trigger())
    if kind == "dry_wet":        # the oscillator is read by the Filter AND behind it: it cannot run a chunk ahead
        return d.Sum(d.Filter(o, 900), d.Multiply(o, 0.3))
    if kind == "two_channels":   # two Filter stages sharing the tile, one after the other
        return d.Filter(d.Multiply(o, [0.5, 0.25]), 1200)
    if kind == "beside":         # a second oscillator that neither feeds the Filter nor hangs on it: a side block
        return d.Sum(d.Filter(o, 1500, "HP"), d.Osc(330 + k / 8, "triangle"))
    if kind == "late_input":     # the Filter ticks BEFORE the Delay it reads: last chunk's registers
        s = d.Sum(o, 0)
        dl = d.Delay(s, 300.25, 2048)
        s.B = d.Multiply(d.Filter(dl, 3000), 0.5)
        return dl
    raise KeyError(kind)


@pytest.mark.parametrize("scan", [0, 1])
@pytest.mark.parametrize("kind", ["env_after", "dry_wet", "two_channels", "beside", "late_input"])
def test_filter_circuits_in_full_workgroups(kind, scan, oracle):
    """Filter circuits at a batch size that fills workgroups of 16 wavefronts (several instances per wavefront, waves 0 and 1 taking
    turns at the recurrences, the other units beside them or a chunk ahead where the circuit allows): sampled instances against the
    oracle, within the Filter tolerance (device tan()).  scan = 1, the default: the circuits whose cutoffs are constants above the
    scan's bound ("late_input": 3 kHz) have no Filter stage — each wavefront scans its own chunk (JitFilterScan)."""
    d.configure(48000)
    V, n = 4096 + 17, 256 * 11 + 40
    ks = np.arange(V)
    uni = descriptor.unify([descriptor.extract(_filter_voice(kind, int(k))) for k in (0, 1, 2)])
    base = uni.params[:, 0].astype(np.float64)
    params = (base[:, None] + (uni.params[:, 1].astype(np.float64) - base)[:, None] * ks[None, :]).astype(np.float32)
    prog = knob_context(48000, DUSP_FILTER_SCAN=scan).build(uni.words, runtime.ENGINE_WAVE)
    pcm = prog.render(n, V, params)
    prog._read_info()
    import re
    waves, per_wave = (int(t) for t in re.search(r", (\d+)x(\d+)", prog.shape).groups())
    staged = not scan or kind != "late_input"  # (the only cutoff of these above the scan's bound: 3 kHz; 1.2 and 1.5 kHz have sum|h| = 47 and 31)
    assert prog.shape.endswith(", scan") == (not staged), prog.shape
    assert "compiled kernel" in prog.shape and waves >= 8 and (waves * per_wave >= 17 or not staged), prog.shape  # (several wavefronts: the turns are taken)
    for i in (0, 1, 31, 32, 63, 64, 1000, 1023, 1024, 2047, 2048, 4095, 4096, V - 1):
        want = oracle.render(uni.words, n, params=params, n_instances=V, instance=i)
        err = np.max(np.abs(pcm[i].astype(np.float64) - want))
        assert err <= 1e-5 * max(1.0, float(np.max(np.abs(want)))), (kind, i, err)
    prog.close()


@pytest.mark.parametrize("delay,ring", [(480, 4096), (257, 2048), (301.5, 2047), (1023, 1501), (700.25, 1000), (300, 1026)])
def test_write_once_delay_geometries(delay, ring, oracle):
    """Delay lines of at least a chunk on the compiled kernel: whole and fractional delays, ring lengths / delays / positions that are
    and are not multiples of 4 (16-byte and 4-byte ring accesses), the ring wrapping inside a chunk; rendered past several ring
    lengths, several instances per wavefront's worth of instances, against the oracle bit for bit (Delay.js:20-41)."""
    d.configure(48000)
    uni = descriptor.unify([descriptor.extract(d.Delay(d.Multiply(d.Osc(300 + 7 * k), 0.5 + k / 64), delay, ring)) for k in range(0, 40)])
    n = 4 * 4096 + 300
    prog = render.context(48000).build(uni.words, runtime.ENGINE_WAVE)
    assert "compiled kernel" in prog.shape
    pcm = prog.render(n, uni.n_instances, uni.params)
    for i in range(0, uni.n_instances, 3):
        want = oracle.render(uni.words, n, params=uni.params, n_instances=uni.n_instances, instance=i)
        assert np.array_equal(pcm[i], want), (delay, ring, i)
    # (the default keeps such a Delay's INPUT samples as a line in LDS where that fits) the same Delay on its write-once ring in memory
    line = knob_context(48000, DUSP_DELAY_LINE=0).build(uni.words, runtime.ENGINE_WAVE)
    assert np.array_equal(line.render(n, uni.n_instances, uni.params), pcm)
    for u in range(line.n_units):
        assert np.array_equal(line.state(u, 7), prog.state(u, 7), equal_nan=True)
    line.close()
    prog.close()


@pytest.mark.parametrize("delay", [0, 0.25, 1, 30.5, 200, 254.5, 255.5, 256, 300.5, 4410])
def test_mono_delay_constant_delays(delay, oracle):
    """MonoDelay (taps first, then the read: a delay of 0 reads the sample's own tap; MonoDelay.js:16-28) with a constant delay: below a
    chunk on the ring-less path of the compiled kernel, from a chunk on on the write-once ring, in between (255.5) on the ordered slot
    operations.  Bit for bit against the oracle."""
    d.configure(48000)
    uni = descriptor.unify([descriptor.extract(d.MonoDelay(d.Multiply(d.Osc(300 + 7 * k), 0.5 + k / 64), delay)) for k in range(0, 24)])
    n = 256 * 24 + 77
    prog = render.context(48000).build(uni.words, runtime.ENGINE_WAVE)
    pcm = prog.render(n, uni.n_instances, uni.params)
    for i in range(0, uni.n_instances, 3):
        want = oracle.render(uni.words, n, params=uni.params, n_instances=uni.n_instances, instance=i)
        assert np.array_equal(pcm[i], want), (delay, i)
    line = knob_context(48000, DUSP_DELAY_LINE=0).build(uni.words, runtime.ENGINE_WAVE)  # (from a chunk on the default keeps the input in LDS, JitDelayLine<true>: here the ring in memory)
    assert np.array_equal(line.render(n, uni.n_instances, uni.params), pcm)
    line.close()
    prog.close()


def test_sample_rate_redux_periods(oracle):
    """SampleRateRedux with a constant amount takes its input every floor(amount) + 1 samples (SampleRateRedux.js:24-35): the compiled
    kernel finds each output's source sample in closed form.  Amounts below one, whole, fractional, around and beyond a chunk, per
    instance; against the oracle bit for bit."""
    d.configure(48000)
    amounts = [0.0, 0.5, 1.0, 2.5, 5.0, 17.25, 255.0, 255.5, 256.0, 300.7, 1000.0, 47.0]
    uni = descriptor.unify([descriptor.extract(d.SampleRateRedux(d.Osc(300 + 11 * i), a)) for i, a in enumerate(amounts)])
    n = 256 * 14 + 33
    prog = render.context(48000).build(uni.words, runtime.ENGINE_WAVE)
    pcm = prog.render(n, uni.n_instances, uni.params)
    prog._read_info()
    assert "compiled kernel" in prog.shape
    for i in range(uni.n_instances):
        want = oracle.render(uni.words, n, params=uni.params, n_instances=uni.n_instances, instance=i)
        assert np.array_equal(pcm[i], want), (amounts[i], i)
    prog.close()


@pytest.mark.parametrize("delay,ring", [(1, 512), (1.5, 700), (30.5, 2048), (64, 1000), (255, 512), (255.75, 513), (100.25, 4410), (30.5, 300)])
def test_delays_shorter_than_a_chunk(delay, ring, oracle):
    """A constant delay of less than a chunk on the compiled kernel: what a sample reads is what two known input samples left in its
    slot, so the kernel needs no ring (JitDelayShort) — whole and fractional delays, the slot-0 rule of the reference's ceil tap,
    rings that wrap inside a chunk, several ring lengths into the render; the last case (a ring shorter than two chunks) stays on
    the ordered slot operations.  Against the oracle bit for bit (Delay.js:20-41)."""
    d.configure(48000)
    uni = descriptor.unify([descriptor.extract(d.Delay(d.Multiply(d.Osc(300 + 7 * k), 0.5 + k / 64), delay, ring)) for k in range(0, 40)])
    n = 3 * 4410 + 300
    prog = render.context(48000).build(uni.words, runtime.ENGINE_WAVE)
    pcm = prog.render(n, uni.n_instances, uni.params)
    prog._read_info()
    assert "compiled kernel" in prog.shape
    for i in range(0, uni.n_instances, 3):
        want = oracle.render(uni.words, n, params=uni.params, n_instances=uni.n_instances, instance=i)
        assert np.array_equal(pcm[i], want), (delay, ring, i)
    prog.close()


@pytest.mark.parametrize("engine", [runtime.ENGINE_WAVE, runtime.ENGINE_CHUNK])
def test_filter_takes_nan_inputs_as_the_reference_does(engine, oracle):
    """0/0 at the input of a Filter every 480 samples (some instances only): the recurrence reads `this.y1 || 0` (Filter.js:42-46), so it
    recovers one sample later.  On the compiled kernel the recurrence runs without those selects and the sub-block that met the NaN is
    given back and done over as written — the instances that share the workgroup must not notice."""
    d.configure(48000)
    def voice(k):
        return d.Filter(d.Divide(d.Osc(100), d.Osc(100 + k)), 700 + k)
    uni = descriptor.unify([descriptor.extract(voice(k)) for k in (0, 1, 0, 3, 0, 0, 7, 0, 2)])
    n = 3000
    prog = render.context(48000).build(uni.words, engine)
    pcm = prog.render(n, uni.n_instances, uni.params)
    if engine == runtime.ENGINE_WAVE:
        assert "compiled kernel" in prog.shape
    met = 0
    for i in range(uni.n_instances):
        want = oracle.render(uni.words, n, params=uni.params, n_instances=uni.n_instances, instance=i)
        # a NaN output sample reaches the PCM as 0 (renderChannelData.js:44): an exact zero between two samples that are far from it
        w = want[0]
        met += int(np.any((w[1:-1] == 0.0) & (np.abs(w[:-2]) > 0.1) & (np.abs(w[2:]) > 0.1)))
        assert np.max(np.abs(pcm[i].astype(np.float64) - want)) <= 1e-5 * max(1.0, float(np.max(np.abs(want)))), i
    assert met >= 5  # (the k = 0 instances divide 0 by 0 at every period's start)
    prog.close()


def _scan_case(case, k):
    o = d.Sum(d.Osc(220 + k / 4, "saw"), d.Multiply(d.Osc(3001 + k), 0.25))  # (a bright input: energy where the filters work)
    if case == "lp1600":
        return d.Filter(o, 1600)
    if case == "lp20000":
        return d.Filter(o, 20000)
    if case == "hp5000":
        return d.Filter(o, 5000, "HP")
    if case == "four_poles":
        return d.Filter(d.Filter(o, 3000), 9000, "HP")
    if case == "loop":      # BASELINE configs[3]'s circuit
        s = d.Sum(d.Osc(110 + k / 64), 0)
        f = d.Filter(d.Delay(s, 480, 4096), 2000)
        s.B = d.Multiply(f, 0.5)
        return f
    if case == "nan":       # 0 / 0 once a third of a second at the Filter's input: that chunk runs as written, the next ones scan on
        return d.Filter(d.Sum(d.Osc(440 + k), d.Sum(d.Divide(d.Osc(3), d.Osc(3)), -1)), 2500)
    if case == "inf":       # 1 / 0 there: y = Infinity, then NaN || 0
        return d.Filter(d.Sum(d.Osc(440 + k), d.Multiply(d.Divide(1, d.Osc(3)), 1e-3)), 2500)
    if case == "huge":      # beyond 1e30: every chunk as written
        return d.Filter(d.Multiply(d.Osc(440 + k), 1e35), 2500)
    raise KeyError(case)


@pytest.mark.parametrize("case", ["lp1600", "lp20000", "hp5000", "four_poles", "loop", "nan", "inf", "huge"])
def test_filters_as_a_scan_stay_within_their_bound(case, oracle):
    """Constant cutoffs above the bound of jit_filter_scan_ok run as a scan over the chunk (jit_prelude.hpp JitFilterScan): the pairs
    (y[t], y[t-1]) travel between lanes unrounded where the reference rounds every y to f32, which costs at most 2^-24 (sum|h| + 2) of the
    Filter's own output scale (sum|h| <= 30: 1.9e-6), times what the circuit makes of it — 1 / (1 - 0.5) behind the loop's feedback: the gate
    (jit_filter_scan_ok) admits what stays within 2.5e-6 at every outlet.  Two seconds of a bright input per case, every
    sample of several instances against the oracle — and the unit's state, which the next render would start from; chunks that meet
    a NaN, an infinity or a value beyond 1e30 are run as the reference writes them (Filter.js:40-46)."""
    d.configure(48000)
    uni = descriptor.unify([descriptor.extract(_scan_case(case, k)) for k in (0, 1, 2)])
    V, n = 70, 96000 + 100
    base = uni.params[:, 0].astype(np.float64)
    params = (base[:, None] + (uni.params[:, 1].astype(np.float64) - base)[:, None] * np.arange(V)[None, :]).astype(np.float32)
    prog = render.context(48000).build(uni.words, runtime.ENGINE_WAVE)
    staged = knob_context(48000, DUSP_FILTER_SCAN=0).build(uni.words, runtime.ENGINE_WAVE)
    pcm = prog.render(n, V, params)
    ref = staged.render(n, V, params)
    assert "compiled kernel" in prog.read_shape()
    assert "JitFilterScan" in runtime.circuit_kernel_source(uni.words, 16, 1)
    for i in (0, 1, 63, 64, 69):
        want, states = oracle.render(uni.words, n, params=params, n_instances=V, instance=i, return_state=True)
        fin = np.isfinite(want)
        scale = max(1.0, float(np.max(np.abs(want[fin]))))
        got = pcm[i].astype(np.float64)
        assert np.array_equal(np.isfinite(got), fin), (case, i)
        err = float(np.max(np.abs(got[fin] - want[fin])))
        assert err <= 2.5e-6 * scale + 1e-5 * scale * (case in ("nan", "inf", "huge")), (case, i, err, scale)
        assert float(np.max(np.abs(ref[i].astype(np.float64)[fin] - want[fin]))) <= 1e-5 * scale
        for u, st in enumerate(states):
            have = prog.state(u, i)
            assert have.size == st.size
            np.testing.assert_allclose(have, st, rtol=2e-5, atol=2e-6 * scale, equal_nan=True)
    prog.close()
    staged.close()


@pytest.mark.parametrize("kind", ["saw_lp800", "four_poles", "two_channels_enveloped", "lp400_forced_short", "three_voices", "three_voices_forced_short", "three_hundred_voices",
                                  "fm_into_lp", "fm_forced_short", "five_cutoffs"])
def test_long_filter_circuits_are_cut_into_segments_that_warm_up(kind, oracle):
    """`renderChannelData(unit, 10)` of ONE filtered circuit — the reference's everyday call — or of a few.  A Filter's recurrence is a chain of
    480 000 dependent steps, so one wavefront used to walk the whole render (DUSP_FILTER_WARM=0: ~10 ms).  Cut into segments that start a
    segment early from rest and store only their own chunks, with the host's check that every stage held at a segment's start what the
    segment before ended with, it is a hundred short chains side by side — and the SAME samples and unit state, bit for bit, as the one long
    chain; within 1e-5 of the oracle (whose coefficients come through libm's tan).  The forced cases use segments far shorter than a 400 Hz
    Filter needs: the check fails somewhere, the render is finished sequentially (one circuit: from the last good segment; several: every
    instance once more as one chain), and is still the one long chain's."""
    d.configure(48000)
    n = 480000
    knobs, V, params = {}, 1, None
    if kind == "saw_lp800":                 # the dusp string `Z110 -> LP800`
        words = descriptor.extract(d.Filter(d.Osc(110, "saw"), 800)).words
    elif kind == "four_poles":
        words = descriptor.extract(d.Filter(d.Filter(d.Sum(d.Osc(220.5), d.Multiply(d.Osc(3301, "triangle"), 0.25)), 3000), 900, "HP")).words
    elif kind == "two_channels_enveloped":
        words = descriptor.extract(d.Multiply(d.Filter(d.Multiply(d.Osc(330), [0.5, 0.25]), 1200), d.Ramp(400000, 1, 0).trigger())).words
    elif kind == "lp400_forced_short":
        words = descriptor.extract(d.Filter(d.Osc(82.4, "saw"), 400)).words
        knobs = {"DUSP_FILTER_WARM": 8}
    elif kind.startswith("fm_"):            # an FM pair (and an LFO on the carrier's modulation depth: two FM levels) in front of the Filter: accumulate passes + warm-up
        car = d.Osc(d.Sum(d.Multiply(d.Osc(d.Sum(d.Multiply(d.Osc(0.7), 3), 110.5)), 180), 440.25))
        words = descriptor.extract(d.Multiply(d.Filter(car, 400 if kind == "fm_forced_short" else 1500), d.Ramp(420000, 1, 0.1).trigger())).words
        knobs = {"DUSP_FILTER_WARM": 8} if kind == "fm_forced_short" else {}
    elif kind == "five_cutoffs":            # a cutoff per instance: the warm-up answers for the lowest of the column
        uni = descriptor.unify([descriptor.extract(d.Filter(d.Osc(82.4 + 27.5 * k, "saw"), 650 + 700 * k, "LP" if True else "HP")) for k in range(5)])
        words, V, params = uni.words, 5, uni.params
    elif kind == "three_hundred_voices":    # a mid-size batch: too few instances to fill the Filter stages' rows, each cut into a few segments
        uni = descriptor.unify([descriptor.extract(d.Multiply(d.Filter(d.Osc(82.4 + 1.25 * k, "saw"), 1000), 0.5 + k / 1024)) for k in range(300)])
        words, V, params, n = uni.words, 300, uni.params, 96000
    else:                                   # three voices of one structure: per-instance f and gain, one constant cutoff
        cutoff = 400 if kind.endswith("forced_short") else 1000
        uni = descriptor.unify([descriptor.extract(d.Multiply(d.Filter(d.Osc(82.4 + 27.5 * k, "saw"), cutoff), 0.5 + k / 8)) for k in range(3)])
        words, V, params = uni.words, 3, uni.params
        if kind.endswith("forced_short"):
            knobs = {"DUSP_FILTER_WARM": 8}
    one = knob_context(48000, DUSP_FILTER_WARM=0).build(words, runtime.ENGINE_WAVE)
    want = one.render(n, V, params)
    assert "compiled kernel" in one.read_shape() and " seg" not in one.read_shape()
    t_one = one.last_kernel_ms()
    prog = (knob_context(48000, **knobs) if knobs else render.context(48000)).build(words)
    got = prog.render(n, V, params)
    shape = prog.read_shape()
    assert " seg" in shape and ("redo@" in shape) == bool(knobs), shape
    assert np.array_equal(got, want), (shape, int(np.argmax((got != want).any(axis=(0, 1)))))
    for i in (range(V) if V <= 5 else (0, 1, 149, 298, 299)):
        for u in range(prog.n_units):
            assert np.array_equal(prog.state(u, i), one.state(u, i), equal_nan=True), (u, i, shape)
        ref = oracle.render(words, n, params=params, n_instances=V, instance=i)
        assert float(np.max(np.abs(got[i].astype(np.float64) - ref))) <= 1e-5 * float(np.max(np.abs(ref)))
    if not knobs:
        again = prog.render(n, V, params)  # (kernels at hand: what the split buys)
        assert np.array_equal(again, want)
        assert prog.last_kernel_ms() <= (0.5 if V > 5 else 0.25) * t_one, (prog.last_kernel_ms(), t_one, shape)
    print("%s: %s; one chain %.2f ms, segments %.2f ms" % (kind, shape, t_one, prog.last_kernel_ms()))
    prog.close()
    one.close()


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("DUSP_WARM_SEEDS", "10"))))  # (a soak: DUSP_WARM_SEEDS=300)
def test_random_filter_circuits_in_warming_segments_equal_the_one_chain(seed):
    """Random feed-forward circuits around Filters — sums of oscillators of every wave table into one to three low / high passes in series or side by
    side, cutoffs between 700 Hz and 15 kHz, gains, an envelope — three seconds each: cut into warming segments (the default) they are the one
    long chain (DUSP_FILTER_WARM=0), sample for sample and state for state; a render whose check fails says so and is the chain's all the same."""
    rng = np.random.RandomState(1000 + seed)
    d.configure(48000)
    def source():
        s = d.Osc(float(np.round(rng.uniform(40, 2000), 2)), ["sin", "saw", "triangle", "square", "8bit"][rng.randint(5)])
        for _ in range(rng.randint(3)):
            s = d.Sum(s, d.Multiply(d.Osc(float(np.round(rng.uniform(40, 6000), 3)), ["sin", "saw", "triangle"][rng.randint(3)]), float(np.round(rng.uniform(0.1, 0.9), 3))))
        return s
    def filt(x):
        return d.Filter(x, float(np.round(np.exp(rng.uniform(np.log(700), np.log(15000))), 1)), ["LP", "HP"][rng.randint(2)])
    g = filt(source())
    for _ in range(rng.randint(3)):
        g = filt(d.Multiply(g, float(np.round(rng.uniform(0.3, 1.5), 3)))) if rng.randint(2) else d.Sum(g, filt(source()))
    if rng.randint(2):
        g = d.Multiply(g, d.Ramp(int(rng.uniform(20000, 140000)), 1, float(np.round(rng.uniform(0, 0.5), 2))).trigger())
    ex = descriptor.extract(g)
    n = 144000 + int(rng.randint(0, 300))
    one = knob_context(48000, DUSP_FILTER_WARM=0, DUSP_FILTER_SCAN=0).build(ex.words, runtime.ENGINE_WAVE)
    want = one.render(n)[0]
    prog = render.context(48000).build(ex.words)
    got = prog.render(n)[0]
    shape = prog.read_shape()
    assert "compiled kernel" in shape and " seg" in shape, shape
    assert np.array_equal(got, want), (shape, int(np.argmax((got != want).any(axis=0))))
    for u in range(prog.n_units):
        assert np.array_equal(prog.state(u), one.state(u), equal_nan=True), (u, shape)
    prog.close()
    one.close()


def test_cutoff_sweeps_scan_where_their_column_allows(oracle):
    """A per-instance cutoff — the most natural parameter sweep of a filtered voice — takes the scan too: the renderer looks at the
    column (smallest and largest value, next to the Delays' regimes) and the gate answers for the whole range; every wavefront computes
    its own coefficients and matrix powers.  BASELINE configs[3]'s circuit with cutoff = 2000 + k % 4000 per instance: on the scan
    (the shape says so), within the gate's 2.5e-6 of the oracle on a spread of instances, PCM and unit state; a column that reaches
    below the bound (1500 + ...: twice 1.85e-6 behind the loop) renders on the Filter stage, bit-equal to DUSP_FILTER_SCAN=0."""
    d.configure(48000)
    def loop(k, base):
        s = d.Sum(d.Osc(110 + k / 64), 0)
        flt = d.Filter(d.Delay(s, 480, 4096), base + k % 4000)
        s.B = d.Multiply(flt, 0.5)
        return flt
    V, n = 4200, 256 * 40 + 33
    for base, scans in ((2000, True), (1500, False)):
        uni = descriptor.unify([descriptor.extract(loop(k, base)) for k in (0, 64)])
        assert uni.n_params == 2
        k = np.arange(V, dtype=np.float64)
        cols = {110.0: 110.0 + k / 64.0, float(base): base + k % 4000}
        params = np.ascontiguousarray(np.stack([cols[float(uni.params[p, 0])] for p in range(2)]).astype(np.float32))
        prog = render.context(48000).build(uni.words, runtime.ENGINE_WAVE)
        pcm = prog.render(n, V, params)
        shape = prog.read_shape()
        assert "compiled kernel" in shape and shape.endswith(", scan") == scans, shape
        staged = knob_context(48000, DUSP_FILTER_SCAN=0).build(uni.words, runtime.ENGINE_WAVE)
        ref = staged.render(n, V, params)
        assert not staged.read_shape().endswith(", scan")
        if not scans:
            assert np.array_equal(pcm, ref)
        for i in (0, 1, 63, 64, 2047, 3999, 4000, V - 1):
            want, states = oracle.render(uni.words, n, params=params, n_instances=V, instance=i, return_state=True)
            scale = float(np.max(np.abs(want)))
            assert float(np.max(np.abs(pcm[i].astype(np.float64) - want))) <= 2.5e-6 * scale, (base, i)
            assert float(np.max(np.abs(ref[i].astype(np.float64) - want))) <= 1e-5 * scale, (base, i)
            for u, st in enumerate(states):
                np.testing.assert_allclose(prog.state(u, i), st, rtol=2e-5, atol=2.5e-6 * scale, equal_nan=True)
        prog.close()
        staged.close()


@pytest.mark.parametrize("gain", [0.9, 0.95, 0.99])
def test_feedback_loops_of_high_gain_keep_the_reference_bits(gain, oracle):
    """BASELINE configs[3]'s circuit with a feedback gain next to 1 (a plucked string, a comb) and an input ON the loop's resonances
    (the loop's latency is 480 + 256 = 736 samples: f = m sr / 736), ten seconds: a deviation injected into such a loop is amplified by
    up to 1 / (1 - g), so jit_filter_scan_ok (2) refuses the scan and the default knobs render on the Filter stage — held to the oracle
    at 1e-5 of scale here, and bit-equal to the stage of DUSP_FILTER_SCAN=0 by construction (the same kernel text).  The scan FORCED onto
    the same circuits (DUSP_FILTER_SCAN=2: what round 3's gate did by default) is measured next to it: it must stay within the path's
    1e-5 too, but nothing promises that — which is why the gate no longer takes it."""
    d.configure(48000)
    def loop(f):
        s = d.Sum(d.Osc(f), 0)
        flt = d.Filter(d.Delay(s, 480, 4096), 2000)
        s.B = d.Multiply(flt, gain)
        return flt
    base = 48000.0 / 736.0
    fs = np.array([2 * base, 3 * base, 5 * base, 110.0, 7 * base + 0.01, 20 * base], dtype=np.float32)
    uni = descriptor.unify([descriptor.extract(loop(float(f))) for f in fs[:2]])
    params = fs.reshape(1, -1)
    V, n = fs.size, 480000
    assert "JitFilterScan" not in runtime.circuit_kernel_source(uni.words, 16, 1)
    prog = render.context(48000).build(uni.words, runtime.ENGINE_WAVE)
    forced = knob_context(48000, DUSP_FILTER_SCAN=2).build(uni.words, runtime.ENGINE_WAVE)
    pcm, scan = prog.render(n, V, params), forced.render(n, V, params)
    assert "compiled kernel" in prog.read_shape() and "compiled kernel" in forced.read_shape()
    worst = 0.0
    for i in range(V):
        want = oracle.render(uni.words, n, params=params, n_instances=V, instance=i)[0].astype(np.float64)
        scale = float(np.max(np.abs(want)))
        assert scale > 1.0 / (1.0 - gain) * 0.05 or i >= 3, (i, scale)  # (on its first resonances the loop really rings up)
        assert float(np.max(np.abs(pcm[i, 0].astype(np.float64) - want))) <= 1e-5 * scale, (gain, i)
        dev = float(np.max(np.abs(scan[i, 0].astype(np.float64) - want))) / scale
        worst = max(worst, dev)
        assert dev <= 1e-5, (gain, i, dev)
    print("forced scan, feedback gain %g: largest deviation %.3g of scale" % (gain, worst))
    prog.close()
    forced.close()


def test_headline_config_full_size(oracle):
    """BASELINE configs[2] at FULL size (1024 voices x 60 s @ 48 kHz, 11.8 GB of PCM resident in HBM):
    a spread of voices is compared with the oracle sample for sample over the whole minute, and a
    checksum of per-voice checksums guards the rest (every voice must differ from its neighbours)."""
    import torch
    sr, n, V = 48000, 2880000, 1024
    d.configure(sr)
    uni = descriptor.unify([descriptor.extract(d.Multiply(d.Osc(10.0 * k), d.Ramp(n, 1, 0).trigger())) for k in (1, 2)])
    params = (10.0 * np.arange(1, V + 1)).astype(np.float32).reshape(1, V)
    ctx = render.context(sr)
    prog = ctx.build(uni.words)
    assert prog.engine == "fused" and prog.shape == "mul(osc(k),ramp)"
    out = torch.empty((V, 1, n), dtype=torch.float32, device="cuda")
    dp = torch.from_numpy(params).cuda()
    prog.render_device(n, V, dp.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for i in (0, 1, 6, 255, 511, 512, 1000, 1023):
        want = oracle.render(uni.words, n, params=params, n_instances=V, instance=i)[0]
        got = out[i, 0].cpu().numpy()
        assert np.array_equal(got, want), "voice %d" % i
    sums = out.double().abs().sum(dim=2).flatten().cpu().numpy()
    assert np.isfinite(sums).all() and (sums > 0).all() and len(np.unique(sums)) > V // 2
    # envelope property: |sample| <= ramp(t) everywhere (sine table is bounded by 1)
    t = torch.arange(n, device="cuda", dtype=torch.float64)
    env = (1.0 - (t + 1) / n).float()
    assert bool((out[::37, 0].abs() <= env + 1e-7).all())
    prog.close()


@pytest.mark.parametrize("fs", [
    [10.0 * k for k in range(1, 41)],                         # integer f: INT path of the fused sum chain
    [20 + k / 8 for k in range(33)] + [-3.25, 47999.5, 0.5],  # 32.32 fixed-point path, negative / near-sr f
])
def test_sum_many_chain_fused_matches_oracle(fs, oracle):
    """Sum.many left-deep chain (configs[2] mix-down): f32 rounding after every add, in chain order."""
    d.configure(48000)
    ex = descriptor.extract(d.Sum.many([d.Osc(f) for f in fs]))
    n = 256 * 37 + 10
    want, states = oracle.render(ex.words, n, return_state=True)
    for engine in (runtime.ENGINE_AUTO, runtime.ENGINE_CHUNK):
        prog = render.context(48000).build(ex.words, engine)
        if engine == runtime.ENGINE_AUTO:
            assert prog.engine == "fused" and prog.shape.startswith("sumchain(")
        got = prog.render(n)[0]
        assert np.array_equal(got, want)
        for u, st in enumerate(states):
            assert np.array_equal(prog.state(u), st), u
        prog.close()


@pytest.mark.parametrize("kind", ["ramp_int", "ramp_frac", "ramp_idle", "gain", "gain_first_operand"])
def test_sum_many_chain_of_enveloped_voices_matches_oracle(kind, oracle):
    """Sum.many over Multiply(Osc, Ramp) / Multiply(Osc, k) voices (Sum.js:18-29, Multiply.js:23-34): the mix of enveloped voices on the
    fused sum chain — one f32 rounding per product, one per add, in chain order; PCM and every unit's state against the oracle."""
    d.configure(48000)
    n = 256 * 29 + 77
    if kind == "ramp_int":
        voices = [d.Multiply(d.Osc(10.0 * k), d.Ramp(5000, 1, 0).trigger()) for k in range(1, 38)]
    elif kind == "ramp_frac":
        voices = [d.Multiply(d.Ramp(7000.5, 0.25, 1.5).trigger(), d.Osc(20 + k / 8)) for k in range(33)]
    elif kind == "ramp_idle":
        voices = [d.Multiply(d.Osc(100.5 * k), d.Ramp(300, 0.5, 2)) for k in range(1, 9)]  # never triggered: a constant y0
    elif kind == "gain":
        voices = [d.Multiply(d.Osc(30.0 * k), 1.0 / k) for k in range(1, 50)]
    else:
        voices = [d.Multiply(0.5 + k / 7, d.Osc(55.25 * k)) for k in range(1, 20)]
    ex = descriptor.extract(d.Sum.many(voices))
    want, states = oracle.render(ex.words, n, return_state=True)
    for engine in (runtime.ENGINE_AUTO, runtime.ENGINE_CHUNK):
        prog = render.context(48000).build(ex.words, engine)
        if engine == runtime.ENGINE_AUTO:
            assert prog.engine == "fused" and prog.shape.startswith("sumchain(osc(k) * "), (prog.engine, prog.shape)
        got = prog.render(n)[0]
        assert np.array_equal(got, want), "first mismatch at %d" % int(np.argmax(got != want))
        for u, st in enumerate(states):
            assert np.array_equal(prog.state(u), st), u
        prog.close()


def test_config2_enveloped_mixdown_full_size(oracle):
    """1024 x Multiply(Osc(10k), Ramp(T, 1, 0) triggered) through Sum.many, 60 s: the head against the oracle, the rest by properties —
    the un-enveloped mix repeats every 4800 samples, so sample n of this one is that mix's sample (n mod 4800) scaled voice by voice by
    the same envelope: bounded by 1024 x env(n), and zero wherever the plain mix's period has all voices at a zero crossing (n = 0 mod 4800)."""
    d.configure(48000)
    n = 2880000
    ex = descriptor.extract(d.Sum.many([d.Multiply(d.Osc(10 * k), d.Ramp(n, 1, 0).trigger()) for k in range(1, 1025)]))
    prog = render.context(48000).build(ex.words)
    assert prog.engine == "fused" and prog.shape == "sumchain(osc(k) * ramp x 1024)", (prog.engine, prog.shape)
    got = prog.render(n)[0, 0]
    head = oracle.render(ex.words, 2048)[0]
    assert np.array_equal(got[:2048], head)
    env = 1.0 - (np.arange(n, dtype=np.float64) + 1) / n
    assert np.all(np.abs(got.astype(np.float64)) <= 1024 * env + 1e-3)
    assert got[4799::4800].tolist() == [0.0] * (n // 4800)  # phase(n) = 10k (n + 1) mod 48000 = 0 for every voice there
    assert float(np.abs(got[:48000]).max()) > 10.0
    ms = prog.last_kernel_ms()
    assert ms < 30.0, ms  # (a guard against falling off the fused path, not a benchmark)
    prog.close()


def test_sum_chain_shapes_that_must_not_fuse():
    d.configure(48000)
    ctx = render.context(48000)
    for g in (d.Sum(d.Osc(100), 0.5),                                    # constant operand
              d.Sum(d.Sum(d.Osc(1), d.Osc(2)), d.Sum(d.Osc(3), d.Osc(4))),  # a tree, not a chain
              d.Sum.many([d.Osc(100), d.Osc(200, "saw"), d.Osc(300)]),      # mixed waveforms
              d.Sum.many([d.Osc(100), d.Osc(2.0 ** -40)]),                  # f finer than 2^-32
              d.Sum.many([d.Multiply(d.Osc(100), d.Ramp(500, 1, 0).trigger()), d.Multiply(d.Osc(200), d.Ramp(600, 1, 0).trigger())]),  # different Ramps
              d.Sum.many([d.Multiply(d.Osc(100), 0.5), d.Osc(200)])):       # voices of different kinds
        prog = ctx.build(descriptor.extract(g).words)
        assert prog.engine == "wave"  # feed-forward, but not a Sum.many chain of constant oscillators
        prog.close()


# ---- the remaining BASELINE configs at their FULL sizes (parity through spot checks + size-independent properties)

def test_config1_three_unit_dag_full_size(oracle):
    """configs[1]: [Multiply A:[Osc f:[Ramp ...]] B:[Osc 3]], 10 s @ 48 kHz, both readings of the Ramp arguments."""
    d.configure(48000)
    for ramp in (lambda: d.Ramp(200, 100, 2), lambda: d.Ramp(2 * 48000, 200, 100).trigger()):
        ex = descriptor.extract(d.Multiply(d.Osc(ramp()), d.Osc(3)))
        prog = render.context(48000).build(ex.words)
        assert prog.engine == "wave"
        got = prog.render(480000)[0]
        assert np.array_equal(got, oracle.render(ex.words, 480000))
        prog.close()


def test_config2_mixdown_full_size(oracle):
    """configs[2], 'voices summed' reading: Sum.many of 1024 Osc(10k), 60 s -> one channel of 2 880 000 samples."""
    d.configure(48000)
    ex = descriptor.extract(d.Sum.many([d.Osc(10 * k) for k in range(1, 1025)]))
    prog = render.context(48000).build(ex.words)
    assert prog.engine == "fused"
    n = 2880000
    got = prog.render(n)[0, 0]
    # the oracle needs ~1 s per 4000 samples of this 2047-unit circuit: check the first chunks exactly ...
    head = oracle.render(ex.words, 4096)[0]
    assert np.array_equal(got[:4096], head)
    # ... and the rest through periodicity: every f is a multiple of 10 Hz, so the mix repeats every 4800 samples
    assert np.array_equal(got[4800:9600], got[:4800]) and np.array_equal(got[-4800:], got[:4800])
    prog.close()


def test_config3_feedback_loops_full_size(oracle):
    """configs[3]: 8192 instances of Osc->Sum->Delay->Filter->Multiply->(Sum), 10 s @ 48 kHz (15.7 GB of PCM) on the kernel compiled
    for the circuit, and — an old caller's DUSP_ENGINE_LOOP — the same program under the name of the loop kernels it replaced."""
    import torch
    d.configure(48000)
    def loop(k):
        s = d.Sum(d.Osc(110 + k / 64), 0)
        f = d.Filter(d.Delay(s, 480, 4096), 2000)
        s.B = d.Multiply(f, 0.5)
        return f
    uni = descriptor.unify([descriptor.extract(loop(k)) for k in (0, 64)])
    V, n = 8192, 480000
    params = (110 + np.arange(V) / 64.0).astype(np.float32).reshape(1, V)
    prog = render.context(48000).build(uni.words)
    assert prog.engine == "wave"
    named = render.context(48000).build(uni.words, runtime.ENGINE_LOOP)
    assert named.engine == "wave"
    named.close()
    out = torch.empty((V, 1, n), dtype=torch.float32, device="cuda")
    dp = torch.from_numpy(params).cuda()
    prog.render_device(n, V, dp.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert "compiled kernel" in prog.read_shape()
    for i in (0, 4095, 8191):
        want = oracle.render(uni.words, n, params=params, n_instances=V, instance=i)[0]
        got = out[i, 0].cpu().numpy()
        assert np.max(np.abs(got.astype(np.float64) - want)) <= 1e-5 * np.max(np.abs(want)), i
    assert bool(torch.isfinite(out[::64]).all())
    assert bool((out[:, 0, :480] == 0).all())  # nothing leaves the 480-sample delay line before t = 480
    prog.close()


def test_config4_sweep_shard_full_size(oracle):
    """configs[4]: one GPU's shard of the 65536-voice sweep: 8192 x Multiply(Osc(20 + k/8), Ramp(T,1,0) triggered), 1 s."""
    import torch
    d.configure(48000)
    V, n = 8192, 48000
    uni = descriptor.unify([descriptor.extract(d.Multiply(d.Osc(20 + k / 8), d.Ramp(n, 1, 0).trigger())) for k in (0, 1)])
    for shard in (0, 7):
        params = (20 + (shard * V + np.arange(V)) / 8.0).astype(np.float32).reshape(1, V)
        prog = render.context(48000).build(uni.words)
        assert prog.engine == "fused"
        out = torch.empty((V, 1, n), dtype=torch.float32, device="cuda")
        dp = torch.from_numpy(params).cuda()
        prog.render_device(n, V, dp.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        for i in (0, 1, 3, 4097, 8191):
            want = oracle.render(uni.words, n, params=params, n_instances=V, instance=i)[0]
            assert np.array_equal(out[i, 0].cpu().numpy(), want), (shard, i)
        prog.close()


def test_feedback_voices_on_compiled_kernels_equal_the_chunk_engine():
    """The feedback voice at constant delays of every regime (a chunk at least, fractional, near the ring's length) and per-instance cutoffs:
    the kernel compiled for the circuit (Filter stage) and the generic chunk engine execute the same operations in the same order per sample —
    identical PCM and state.  (Rounds 1-3 kept three hand-written kernels for this voice to the same test; they are gone.)"""
    d.configure(48000)
    def loop(k, delay):
        s = d.Sum(d.Osc(110 + k / 64), 0)
        f = d.Filter(d.Delay(s, delay, 4096), 2000 + k)
        s.B = d.Multiply(f, 0.5)
        return f
    for delay in (480, 300.25, 3840):
        uni = descriptor.unify([descriptor.extract(loop(k, delay)) for k in range(0, 1700, 41)])
        n = 256 * 9 + 17
        results = []
        for engine in (runtime.ENGINE_AUTO, runtime.ENGINE_CHUNK):
            prog = knob_context(48000, DUSP_FILTER_SCAN=0).build(uni.words, engine)
            pcm = prog.render(n, uni.n_instances, uni.params)
            assert ("compiled kernel" in prog.read_shape()) == (engine == runtime.ENGINE_AUTO), prog.read_shape()
            states = [prog.state(u, instance=3) for u in range(5)]
            results.append((pcm, states))
            prog.close()
        for pcm, states in results[1:]:
            assert np.array_equal(pcm, results[0][0])
            for s_a, s_b in zip(states, results[0][1]):
                assert np.array_equal(s_a, s_b, equal_nan=True)


@pytest.mark.parametrize("n_inst,n_ch,n", [(1, 2, 2400), (3, 3, 1000), (5, 8, 257), (2, 33, 513), (1, 64, 4096), (7, 1, 300), (1, 2, 2_880_000)])
def test_interleave_device_is_a_transpose(n_inst, n_ch, n):
    """dusp_interleave_device: planar [instance][channel][sample] -> frames [instance][sample][channel], bit for bit,
    for channel counts on both sides of the tile-size switch, ragged tails and a full-length stereo render."""
    import torch
    ctx = render.context(48000)
    g = torch.Generator(device="cpu").manual_seed(n_ch * 1000 + n)
    planar = torch.randn((n_inst, n_ch, n), generator=g, dtype=torch.float32).cuda()
    frames = torch.full((n_inst, n, n_ch), float("nan"), dtype=torch.float32, device="cuda")
    ctx.interleave(planar.data_ptr(), n_inst, n_ch, n, frames.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(frames, planar.transpose(1, 2).contiguous())


def test_render_host_interleaved_matches_planar():
    from conftest import Golden
    g = Golden("rest_pan")
    prog = render.context(48000).build(g.desc)
    planar = prog.render(g.n_samples, 1)
    frames = prog.render(g.n_samples, 1, interleaved=True)
    assert frames.shape == (1, g.n_samples, 2) and np.array_equal(frames[0].T, planar[0])
    prog.close()


def test_time_split_with_many_instances_and_parameters():
    """Time-split wave rendering of a BATCH: per-instance parameters feed both FM levels; every (instance, segment)
    pair gets its own wavefront and the per-instance prefix of phase totals.  Must equal the unsplit render."""
    d.configure(48000)
    def voice(k):
        return d.Multiply(d.Osc(d.Sum(d.Multiply(d.Osc(3.5 + k), 40 + 3 * k), 220.25 + 10 * k)), d.Ramp(5000, 1, 0.25).trigger())
    uni = descriptor.unify([descriptor.extract(voice(k)) for k in range(7)])
    assert uni.n_params >= 3
    n = 256 * 37 + 100
    prog = knob_context(48000, DUSP_WAVE_SEGMENTS=1).build(uni.words, runtime.ENGINE_WAVE)
    want = prog.render(n, uni.n_instances, uni.params)
    want_state = [prog.state(u, instance=5) for u in range(prog.n_units)]
    prog.close()
    for segs in ("3", "11", "38"):
        prog = knob_context(48000, DUSP_WAVE_SEGMENTS=segs).build(uni.words, runtime.ENGINE_WAVE)
        got = prog.render(n, uni.n_instances, uni.params)
        assert np.array_equal(got, want), segs
        for a, b in zip([prog.state(u, instance=5) for u in range(prog.n_units)], want_state):
            assert np.array_equal(a, b, equal_nan=True)
        prog.close()


def test_circlebuffer_batch_wave_equals_chunk():
    """CircleBuffer taps with PER-INSTANCE offsets and a feedback writer, 70 instances (ragged last workgroup):
    the wave engine (lane-parallel ring windows) against the chunk engine (the reference's schedule)."""
    d.configure(48000)
    def taps(k):  # the MultiTapDelay topology of the `circlebuffer_taps` golden, offsets and pitch per instance
        buf = d.CircleBuffer(1, 0.05)
        w = d.CircleBufferWriter(buf)
        w.preWipe = True
        w.IN = d.Osc(200 + 3.5 * k)
        tap = d.CircleBufferReader(buf, 0.01 + 0.0001 * k)
        tap.chain(w)
        fb_tap = d.CircleBufferReader(buf, 0.02)
        fb_tap.chain(w)
        fb = d.CircleBufferWriter(buf, 0.005 + 0.00005 * k)
        fb.IN = d.quick.multiply(fb_tap, 0.5)
        fb.chain(w)
        return d.Sum(tap, fb_tap)
    uni = descriptor.unify([descriptor.extract(taps(k)) for k in range(70)])
    assert uni.n_params >= 3
    n = 256 * 30 + 77
    ctx = render.context(48000)
    out = []
    for engine in (runtime.ENGINE_WAVE, runtime.ENGINE_CHUNK):
        prog = ctx.build(uni.words, engine)
        pcm = prog.render(n, uni.n_instances, uni.params)
        out.append((pcm, [prog.state(u, instance=69) for u in range(prog.n_units)]))
        prog.close()
    assert np.abs(out[0][0]).max() > 0.5
    assert np.array_equal(out[0][0], out[1][0])
    for a, b in zip(out[0][1], out[1][1]):
        assert np.array_equal(a, b, equal_nan=True)


def test_running_sums_in_closed_form_equal_the_sequential_engine():
    """Timer's and Shape's running f64 sums (`t += c`): the wave engine evaluates them with repeat_add() — lane-parallel inside
    a chunk, and from the render's start for every time segment — the chunk engine adds sample by sample like the reference.
    300 voices with random durations (per-instance parameters), then one long circuit split in time."""
    from dusp_amd import descriptor
    d.configure(48000)
    rng = np.random.RandomState(11)

    def voice(dur, f):
        return d.Multiply(d.Osc(d.Sum(d.Multiply(d.Timer(), 50), f)), d.Shape("decaySquared", dur, -0.25, 1).trigger())

    uni = descriptor.unify([descriptor.extract(voice(0.1 + 0.01 * k, 100 + k)) for k in range(3)])
    V, n = 300, 48000 * 2 + 77
    params = np.empty((uni.n_params, V), dtype=np.float32)
    for p in range(uni.n_params):
        lo, hi = float(uni.params[p].min()), float(uni.params[p].max())
        params[p] = rng.uniform(0.001, 5.0, V) if hi < 1 else rng.uniform(lo, hi * 3, V)  # durations from 1 ms to 5 s
    ctx = render.context(48000)
    outs, states = [], []
    for engine in (runtime.ENGINE_CHUNK, runtime.ENGINE_WAVE):
        prog = ctx.build(uni.words, engine)
        outs.append(prog.render(n, V, params))
        states.append([[prog.state(u, i) for u in range(prog.n_units)] for i in (0, 7, V - 1)])
        prog.close()
    assert np.array_equal(outs[0], outs[1])
    for a, b in zip(states[0], states[1]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y, equal_nan=True)
    # one circuit, 10 s: AUTO -> wave engine, split in time; the chunk engine is the sequential reference
    ex = descriptor.extract(voice(1.37, 330.5))
    n = 480000
    whole = ctx.build(ex.words, runtime.ENGINE_CHUNK)
    want = whole.render(n)
    want_state = [whole.state(u) for u in range(whole.n_units)]
    whole.close()
    split = ctx.build(ex.words)
    assert split.engine == "wave"
    got = split.render(n)
    assert split.last_kernel_ms() < 5.0, "not split in time?"
    got_state = [split.state(u) for u in range(split.n_units)]
    split.close()
    assert np.array_equal(got, want)
    for x, y in zip(got_state, want_state):
        assert np.array_equal(x, y, equal_nan=True)


@pytest.mark.parametrize("case", ["decay_1s", "attack_short", "semisine_range", "decaysq_numbers", "idle", "midway", "finished_before", "tiny_duration"])
def test_fused_shape_voice_equals_the_sequential_engine(case):
    """mul(osc(k), shape): the canonical Dusp voice "O440 * D1" on the fused, time-parallel kernel (Shape's running sum in
    closed form) against the chunk engine, which adds sample by sample — PCM and state, per-instance frequencies."""
    from dusp_amd import descriptor
    d.configure(48000)

    def shape():
        if case == "decay_1s":
            return d.Shape("decay", 1).trigger()
        if case == "attack_short":
            return d.Shape("attack", 0.0123).trigger()
        if case == "semisine_range":
            return d.Shape("semiSine", 0.31, -0.5, 2).trigger()
        if case == "decaysq_numbers":
            s = d.Shape("decaySquared", 0.07, 0.25, 0.75).trigger()
            s.leftEdge, s.rightEdge = 0.5, -1
            return s
        if case == "idle":
            return d.Shape("decay", 0.5, 0.1, 0.9)
        if case == "midway":
            s = d.Shape("decay", 0.2).trigger()
            s.t = 12345.678
            return s
        if case == "finished_before":
            s = d.Shape("attack", 0.1).trigger()
            s.t, s.finished = 50000.5, True
            return s
        return d.Shape("decay", 3e-5).trigger()  # over after two samples

    uni = descriptor.unify([descriptor.extract(d.Multiply(d.Osc(110.5 + 7 * k), shape())) for k in range(3)])
    V, n = 37, 48000 + 300
    params = (uni.params[:, :1] + (uni.params[:, 1:2] - uni.params[:, :1]) * np.arange(V)[None, :]).astype(np.float32)
    ctx = render.context(48000)
    progs = {e: ctx.build(uni.words, e) for e in (runtime.ENGINE_CHUNK, runtime.ENGINE_AUTO)}
    assert progs[runtime.ENGINE_AUTO].engine == "fused" and progs[runtime.ENGINE_AUTO].shape == "mul(osc(k),shape)"
    outs = {e: p.render(n, V, params) for e, p in progs.items()}
    assert np.array_equal(outs[runtime.ENGINE_CHUNK], outs[runtime.ENGINE_AUTO])
    for i in (0, V - 1):
        for u in range(progs[runtime.ENGINE_CHUNK].n_units):
            assert np.array_equal(progs[runtime.ENGINE_CHUNK].state(u, i), progs[runtime.ENGINE_AUTO].state(u, i), equal_nan=True)
    for p in progs.values():
        p.close()


def test_big_mix_down_fits_the_wave_engine_by_buffer_liveness():
    """200 enveloped voices summed into one output: 800 units, 800 chunk buffers if every outlet kept its own — LDS holds
    them because a feed-forward graph needs a buffer only from its producer to its last reader — 3 slots here, once the ops run
    depth-first from the output instead of level by level —, and only the
    stateful ops own LDS state.  Bit-identical to the chunk engine; split in time, since every state is a closed-form sum."""
    from dusp_amd import descriptor
    d.configure(48000)
    voices = [d.Multiply(d.Osc(55.0 * (k + 1) + 0.25 * (k % 4), ["sin", "saw", "triangle"][k % 3]),
                         d.Shape(["decay", "decaySquared", "semiSine"][k % 3], 0.05 + 0.01 * k, 0, 1.0 / (k + 1)).trigger()) for k in range(200)]
    ex = descriptor.extract(d.Sum.many(voices))
    n = 48000 * 2 + 5
    ctx = render.context(48000)
    ref = ctx.build(ex.words, runtime.ENGINE_CHUNK)
    want = ref.render(n)
    want_state = [ref.state(u) for u in range(0, ref.n_units, 37)]
    ref.close()
    prog = ctx.build(ex.words)
    assert prog.engine == "wave" and "3 chunk buffers" in prog.shape, (prog.engine, prog.shape)
    got = prog.render(n)
    kernel_ms = prog.last_kernel_ms()
    got_state = [prog.state(u) for u in range(0, prog.n_units, 37)]
    prog.close()
    assert np.array_equal(got, want)
    for a, b in zip(got_state, want_state):
        assert np.array_equal(a, b, equal_nan=True)
    assert kernel_ms < 50, kernel_ms


def test_first_render_does_not_wait_for_the_compiler(oracle):
    """Product default (DUSP_WAVE_JIT=1): the first render of a circuit structure this process has not seen is not held up by
    the 0.3-0.8 s compile when the interpreter kernel finishes it sooner — the kernel compiles in a background thread and a
    later render of the same structure runs on it.  Same PCM and state either way; and the interpreter (DUSP_WAVE_JIT=0) agrees."""
    import time
    d.configure(48000)
    # (a structure no other test builds: 7 oscillators in a chain of FM pairs)
    g = d.Osc(330.5)
    for k in range(3):
        g = d.Sum(d.Multiply(d.Osc(d.Sum(d.Multiply(g, 17.0 + k), 201.25 * (k + 1))), 0.5), d.Osc(77.0 + k))
    ex = descriptor.extract(g)
    n = 256 * 5 + 11
    want = oracle.render(ex.words, n)
    prog = knob_context(48000, DUSP_WAVE_JIT=1).build(ex.words, runtime.ENGINE_WAVE)
    t0 = time.perf_counter()
    first = prog.render(n)[0]
    waited = time.perf_counter() - t0
    prog._read_info()
    # (the session's code-object cache starts empty — conftest.py — and no other test builds this structure: there is nothing to find)
    assert "kernel compiling" in prog.shape and waited < 0.25, (prog.shape, waited)
    assert np.array_equal(first, want)
    state_first = [prog.state(u) for u in range(prog.n_units)]
    prog.close()
    deadline = time.perf_counter() + 60
    while True:  # a fresh program of the same structure finds the kernel once the background compile is done
        prog = knob_context(48000, DUSP_WAVE_JIT=1).build(ex.words, runtime.ENGINE_WAVE)
        again = prog.render(n)[0]
        prog._read_info()
        shape = prog.shape
        state_again = [prog.state(u) for u in range(prog.n_units)]
        prog.close()
        assert np.array_equal(again, want)
        if "compiled kernel" in shape or time.perf_counter() > deadline:
            break
        time.sleep(0.1)
    assert "compiled kernel" in shape, shape
    for a, b in zip(state_first, state_again):
        assert np.array_equal(a, b, equal_nan=True)
    off = knob_context(48000, DUSP_WAVE_JIT=0).build(ex.words, runtime.ENGINE_WAVE)
    assert np.array_equal(off.render(n)[0], want) and "chunk buffers in LDS" in off.shape
    off.close()


def test_a_filter_stage_circuit_reaches_its_kernel_under_the_default_knob():
    """DUSP_WAVE_JIT=1 and a circuit with a Filter stage at a batch that fills 16-wavefront workgroups: the kernel's geometry search may need
    a second text (the recurrence loop's narrower form) — it is compiled alongside the first, so a few renders later the circuit runs on its
    compiled kernel, with the interpreter's PCM (the stage is the same arithmetic on both)."""
    import time
    d.configure(48000)
    voice = lambda k: d.Sum(d.Filter(d.Sum(d.Osc(97.5 + k), d.Osc(301.25, "triangle")), 640.5), d.Multiply(d.Osc(3.5), 0.125))  # (no other test's structure)
    uni = descriptor.unify([descriptor.extract(voice(k)) for k in (0, 1, 2)])
    V, n = 4096, 256 * 6
    base = uni.params[:, 0].astype(np.float64)
    params = (base[:, None] + (uni.params[:, 1].astype(np.float64) - base)[:, None] * np.arange(V)[None, :]).astype(np.float32)
    prog = knob_context(48000, DUSP_WAVE_JIT=1).build(uni.words, runtime.ENGINE_WAVE)
    first = prog.render(n, V, params)
    assert "kernel compiling" in prog.read_shape()  # (an empty cache per test session, a structure of this test's own: conftest.py)
    deadline, renders = time.perf_counter() + 90, 1
    while "compiled kernel" not in prog.read_shape() and time.perf_counter() < deadline:
        time.sleep(0.3)
        again = prog.render(n, V, params)
        renders += 1
        assert np.array_equal(again, first)
    assert "compiled kernel" in prog.read_shape(), (prog.read_shape(), renders)
    prog.close()


@pytest.mark.parametrize("name", ["patch_scary"])  # (the grow_* vectors settle within their first chunk and are wave-engine programs as they stand)
def test_growing_channel_counts_hand_over_to_a_compiled_kernel(name, oracle):
    """Circuits whose channel counts grow during the first chunks (a feedback edge sees one channel at first, more later: Program::warm_ops)
    render those chunks on the chunk engine and the rest on the kernel compiled for the settled circuit: same PCM and state as the chunk
    engine alone, the reference's windows, and 10 s in well under half the chunk engine's time."""
    from conftest import Golden
    g = Golden(name)
    ctx = render.context(g.sample_rate)
    n = 480000
    prog = ctx.build(g.desc)  # AUTO
    assert prog.engine == "chunk"
    pcm = prog.render(n)[0]
    assert "then compiled kernel" in prog.read_shape(), prog.read_shape()
    ms = prog.last_kernel_ms()
    ref = ctx.build(g.desc, runtime.ENGINE_CHUNK)
    want = ref.render(n)[0]
    assert "compiled" not in ref.read_shape()
    assert np.array_equal(pcm, want), "first mismatch at sample %d" % int(np.argmax((pcm != want).any(axis=0)))  # (Filters too: the engines share their coefficients)
    for u in range(prog.n_units):
        assert np.array_equal(prog.state(u), ref.state(u), equal_nan=True), u
    head = oracle.render(g.desc, 256 * 12)
    assert np.max(np.abs(pcm[:, :256 * 12].astype(np.float64) - head)) <= 1e-5 * max(1.0, float(np.max(np.abs(head))))
    # (this patch's delay times move by several samples per sample: its MonoDelays keep the ordered slot operations, ~50 us a chunk for one
    # circuit; what the hand-off removes is the chunk engine's ~150 us a chunk)
    assert ms <= 0.6 * ref.last_kernel_ms() and ms <= 160.0, (ms, ref.last_kernel_ms())
    prog.close()
    ref.close()


def test_a_second_process_finds_its_kernel_on_disk():
    """The code-object cache (on by default): the first render of a circuit in a NEW process does not pay the compile again — nor a render on
    the interpreter — when an earlier process left the kernel on disk (tools/first_call.py: FM voice under an envelope, 10 s)."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "first_call.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    first, second = rep["empty_cache_waiting_for_the_compile"], rep["second_process_default_knob"]
    assert "compiled kernel" in first["shape"] and "compiled kernel" in second["shape"], rep
    assert first["first_ms"] > 100.0            # (a real hiprtc compile)
    assert second["first_ms"] <= 30.0, rep      # (generate the text, read the file, load the module, render, download: milliseconds)


def test_circuits_above_a_hundred_units_run_on_compiled_kernels(oracle):
    """A Sum.many of FM pairs — voices the fused sum chain does not take — is 5 N - 1 channel-expanded units.  From 96 units on the
    generator emits the voice's units ONCE, in a loop over the voices (jit_codegen.hpp VoicePlan: constants, parameter and state slots out
    of a per-voice table, oscillator state in per-voice arrays, the chain as the loop's running f32 sum), whatever N up to 256; circuits
    that are no such sum run as straight-line code up to DUSP_JIT_MAX_UNITS (256) and on the interpreter beyond.  Same PCM and unit state
    as the chunk engine bit for bit, and the oracle's PCM."""
    import dusp_amd as d
    from dusp_amd import descriptor
    d.configure(48000)
    voice = lambda k, j: d.Osc(d.Sum(d.Multiply(d.Osc(3.0 + j / 7 + k / 100), 40), 220 + 11.5 * j + k / 4))
    mix = lambda k, nv: d.Sum.many([voice(k, j) for j in range(nv)])
    n = 256 * 9 + 100
    for nv, knobs in ((30, {}), (30, {"DUSP_JIT_LOOP_VOICES": 4096}), (100, {})):
        uni = descriptor.unify([descriptor.extract(mix(k, nv)) for k in (0, 8, 16)])
        ctx = knob_context(48000, **knobs) if knobs else render.context(48000)
        prog = ctx.build(uni.words, runtime.ENGINE_WAVE)
        assert prog.n_units == 5 * nv - 1
        pcm = prog.render(n, 3, uni.params)
        assert "compiled kernel: %d units" % (5 * nv - 1) in prog.read_shape(), prog.read_shape()
        ref = ctx.build(uni.words, runtime.ENGINE_CHUNK)
        want = ref.render(n, 3, uni.params)
        assert np.array_equal(pcm, want), (nv, knobs)
        for u in range(prog.n_units):
            for i in (0, 2):
                assert np.array_equal(prog.state(u, i), ref.state(u, i), equal_nan=True), (nv, u, i)
        if nv == 30:
            for i in range(3):
                assert np.array_equal(pcm[i], oracle.render(uni.words, n, params=uni.params, n_instances=3, instance=i)), i
        prog.close()
        ref.close()
    # voices with their own envelopes and a few maps: FM pair x Ramp (a duration per voice), rectified and scaled — 9 units a voice
    env = lambda k, j: d.Multiply(d.Abs(voice(k, j)), d.FixedMultiply(0.5, d.Multiply(d.Ramp(1500 + 37 * j, 1, 0.125).trigger(), 1.0 + k / 64)))
    uni = descriptor.unify([descriptor.extract(d.Sum.many([env(k, j) for j in range(33)])) for k in (0, 8, 16)])
    prog = render.context(48000).build(uni.words, runtime.ENGINE_WAVE)
    pcm = prog.render(n, 3, uni.params)
    assert "compiled kernel: %d units" % prog.n_units in prog.read_shape() and prog.n_units > 256, prog.read_shape()
    ref = render.context(48000).build(uni.words, runtime.ENGINE_CHUNK)
    assert np.array_equal(pcm, ref.render(n, 3, uni.params))
    for u in range(prog.n_units):
        assert np.array_equal(prog.state(u, 1), ref.state(u, 1), equal_nan=True), u
    assert np.array_equal(pcm[2], oracle.render(uni.words, n, params=uni.params, n_instances=3, instance=2))
    prog.close()
    ref.close()
    # enveloped oscillators, the dusp strings' `O440 * D0.5` voices: Shape envelopes with a duration per voice (and per instance)
    shaped = lambda k, j: d.Multiply(d.Osc(110 + 7.25 * j + k / 8), d.Shape("decay" if j % 2 else "attack", 0.02 + j / 500 + k / 6400).trigger())
    uni = descriptor.unify([descriptor.extract(d.Sum.many([shaped(k, j) for j in range(30)])) for k in (0, 8, 16)])
    prog = render.context(48000).build(uni.words, runtime.ENGINE_WAVE)
    pcm = prog.render(n, 3, uni.params)
    ref = render.context(48000).build(uni.words, runtime.ENGINE_CHUNK)
    if "loop" in prog.read_shape():  # (two tables by turns: not isomorphic — straight-line code; see below for the loop)
        raise AssertionError(prog.read_shape())
    assert np.array_equal(pcm, ref.render(n, 3, uni.params))
    prog.close()
    ref.close()
    shaped = lambda k, j: d.Multiply(d.Osc(110 + 7.25 * j + k / 8), d.Shape("decay", 0.02 + j / 500 + k / 6400).trigger())
    uni = descriptor.unify([descriptor.extract(d.Sum.many([shaped(k, j) for j in range(70)])) for k in (0, 8, 16)])
    prog = render.context(48000).build(uni.words, runtime.ENGINE_WAVE)
    pcm = prog.render(n, 3, uni.params)
    assert "compiled kernel: 279 units" in prog.read_shape() and "loop" in prog.read_shape(), prog.read_shape()
    ref = render.context(48000).build(uni.words, runtime.ENGINE_CHUNK)
    assert np.array_equal(pcm, ref.render(n, 3, uni.params))
    for u in range(prog.n_units):
        assert np.array_equal(prog.state(u, 2), ref.state(u, 2), equal_nan=True), u
    prog.close()
    ref.close()
    # notes under AHD envelopes (constant times, some stages ending inside a chunk: closed form and lane walk by turns)
    note = lambda k, j: d.Multiply(d.Osc(110 + 7.25 * j + k / 8), d.AHD(0.002 + j / 4000, 0.004 + k / 8000, 0.02 + j / 900).trigger())
    uni = descriptor.unify([descriptor.extract(d.Sum.many([note(k, j) for j in range(40)])) for k in (0, 8, 16)])
    prog = render.context(48000).build(uni.words, runtime.ENGINE_WAVE)
    pcm = prog.render(n, 3, uni.params)
    assert "compiled kernel: 159 units" in prog.read_shape() and "loop" in prog.read_shape(), prog.read_shape()
    ref = render.context(48000).build(uni.words, runtime.ENGINE_CHUNK)
    assert np.array_equal(pcm, ref.render(n, 3, uni.params))
    for u in range(prog.n_units):
        assert np.array_equal(prog.state(u, 1), ref.state(u, 1), equal_nan=True), u
    prog.close()
    ref.close()
    # ... and what hangs on the mix: a master gain per instance, an offset, a clip
    uni = descriptor.unify([descriptor.extract(d.HardClipAbove(d.Sum(d.Multiply(d.Sum.many([note(k, j) for j in range(40)]), 0.2 + k / 100), -0.05), 0.6)) for k in (0, 8, 16)])
    prog = render.context(48000).build(uni.words, runtime.ENGINE_WAVE)
    pcm = prog.render(n, 3, uni.params)
    assert "compiled kernel: 162 units" in prog.read_shape() and "loop" in prog.read_shape(), prog.read_shape()
    ref = render.context(48000).build(uni.words, runtime.ENGINE_CHUNK)
    assert np.array_equal(pcm, ref.render(n, 3, uni.params))
    assert np.array_equal(pcm[1], oracle.render(uni.words, n, params=uni.params, n_instances=3, instance=1))
    prog.close()
    ref.close()
    # panned voices: two output channels, a chain each, the voice's units (and one compensation pow() per voice) once
    uni = descriptor.unify([descriptor.extract(d.Multiply(d.Sum.many([d.Pan(note(k, j), -1 + j / 18 + k / 1000) for j in range(36)]), 0.5)) for k in (0, 8, 16)])
    prog = render.context(48000).build(uni.words, runtime.ENGINE_WAVE)
    pcm = prog.render(n, 3, uni.params)
    assert prog.n_out_channels == 2 and "loop" in prog.read_shape(), prog.read_shape()
    ref = render.context(48000).build(uni.words, runtime.ENGINE_CHUNK)
    want = ref.render(n, 3, uni.params)
    scale = float(np.max(np.abs(want)))
    assert np.max(np.abs(pcm.astype(np.float64) - want)) <= 1e-6 * scale   # (Pan's compensation is a pow(): device math, the north star's tolerance)
    assert np.max(np.abs(pcm[1].astype(np.float64) - oracle.render(uni.words, n, params=uni.params, n_instances=3, instance=1))) <= 1e-5 * scale
    prog.close()
    ref.close()
    # one circuit, a long render: time is cut into segments (every FM carrier's start phases from the loop's own accumulate pass + prefix)
    one = descriptor.extract(d.Sum.many([env(3, j) for j in range(40)]))
    prog = render.context(48000).build(one.words, runtime.ENGINE_WAVE)
    long_n = 48000 + 77
    pcm = prog.render(long_n)[0]
    assert "compiled kernel: %d units" % prog.n_units in prog.read_shape() and "loop" in prog.read_shape() and "seg" in prog.read_shape(), prog.read_shape()
    want, states = oracle.render(one.words, long_n, return_state=True)
    assert np.array_equal(pcm, want)
    for u, st in enumerate(states):
        assert np.array_equal(prog.state(u), np.asarray(st, dtype=np.float64), equal_nan=True), u
    prog.close()
    # a few instances, each cut in time (parameters per instance AND start phases per segment)
    uni = descriptor.unify([descriptor.extract(d.Multiply(d.Sum.many([env(k, j) for j in range(20)]), 0.5)) for k in (0, 8, 16, 24, 32)])
    prog = render.context(48000).build(uni.words, runtime.ENGINE_WAVE)
    pcm = prog.render(long_n, 5, uni.params)
    assert "loop" in prog.read_shape() and "seg" in prog.read_shape(), prog.read_shape()
    for i in (0, 3, 4):
        assert np.array_equal(pcm[i], oracle.render(uni.words, long_n, params=uni.params, n_instances=5, instance=i)), i
    prog.close()
    # voices of two kinds by turns: no loop, and 4 x 40 + 2 x 40 + 79 = 319 units are more than straight-line code takes
    other = lambda j: d.Multiply(d.Osc(50.5 + j), 0.25)
    mixed = d.Sum.many([voice(0, j) if j % 2 else other(j) for j in range(80)])
    big = render.context(48000).build(descriptor.extract(mixed).words, runtime.ENGINE_WAVE)
    big.render(512)
    assert "compiled" not in big.read_shape()
    big.close()
