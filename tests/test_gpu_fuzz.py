"""Differential fuzzing of the engines: random circuits built from every unit the wave engine runs — oscillators (FM),
ramps, envelopes, filters with constant and modulated cutoffs, delay lines (constant, sub-chunk, signal-rate; MonoDelay,
ReadBackDelay), the comb family, CircleBuffer taps,
feedback edges, multi-channel signals — rendered by the wave engine and by the chunk engine (the reference's schedule,
pinned by the golden vectors) must agree bit for bit, PCM and written-back state, one-shot and in continued segments."""
import random

import numpy as np
import pytest

import dusp_amd as d
from conftest import knob_context
from dusp_amd import descriptor, render, runtime

pytestmark = pytest.mark.gpu


def random_circuit(rng, crng=None):
    """A random feed-forward-or-feedback circuit; returns the unit to render.  `rng` decides the STRUCTURE, `crng` (default:
    the same stream) the inlet constants — so one structure seed with several constant seeds gives a batch of voices."""
    structure = rng
    rng = _Split(structure, crng if crng is not None else structure)
    pool = [d.Osc(rng.cchoice([110, 220.5, 3.25, 1000.125])), d.Ramp(rng.choice([600, 3000]), 1, rng.choice([0, 0.25])).trigger()]

    def pick():
        return rng.choice(pool)

    def const_or_signal(lo, hi):
        return pick() if rng.random() < 0.35 else round(rng.cuniform(lo, hi), 3)

    feedback_sum = None
    if rng.random() < 0.4:  # a loop: Sum(x, <closed later>)
        feedback_sum = d.Sum(pick(), 0)
        pool.append(feedback_sum)
    for _ in range(rng.randint(3, 9)):
        kind = rng.randrange(20)
        if kind == 0:
            u = d.Osc(d.Sum(d.Multiply(pick(), rng.cchoice([20, 200])), rng.cchoice([220, 440.5])), rng.choice(["sin", "saw", "triangle", "square"]))
        elif kind == 1:
            u = d.Multiply(pick(), const_or_signal(-1, 1))
        elif kind == 2:
            u = d.Sum(pick(), const_or_signal(-1, 1))
        elif kind == 3:
            u = d.Filter(pick(), rng.cchoice([500, 2000.5]) if rng.random() < 0.6 else d.Sum(d.Multiply(pick(), 300), 1500), rng.choice(["LP", "HP"]))
        elif kind == 4:
            u = d.Delay(pick(), rng.choice([256, 300.25, 700, 1000.5]), 2048)
        elif kind == 17:  # delay lines whose taps meet inside a chunk: sub-chunk, signal-rate, rings shorter than a chunk
            how = rng.randrange(4)
            if how == 0:
                u = d.Delay(pick(), rng.cchoice([0, 1, 30.5, 100, 255.75]), rng.choice([2048, 300]))
            elif how == 1:
                u = d.Delay(pick(), d.Sum(d.Multiply(pick(), rng.cchoice([40, 400])), rng.cchoice([200, 500.5])), 2048)
            elif how == 2:
                u = d.Delay(pick(), rng.cchoice([10, 77.25, 150]), rng.choice([100, 200]))
            else:
                u = d.Delay(d.Multiply(pick(), [1, -0.5]), [rng.cchoice([20.5, 300]), 700.25], 1024)
        elif kind == 18:
            u = d.MonoDelay(pick(), rng.cchoice([0, 12.5, 123.5, 4000]) if rng.random() < 0.5 else d.Sum(d.Multiply(pick(), rng.cchoice([30, 300])), 350))
        elif kind == 19:
            u = d.ReadBackDelay(pick(), rng.cchoice([0, 100, 2000]) if rng.random() < 0.6 else d.Multiply(d.Timer(), 48000 * 8), rng.choice([4096, 300]))
        elif kind == 5:
            u = d.CombFilter(rng.choice([0.0007, 0.004, 0.02]), const_or_signal(-0.8, 0.8))
            u.IN = pick()
        elif kind == 6:
            u = d.AllPass(rng.choice([0.0005, 0.0021, 0.012]), rng.uniform(-0.7, 0.7))
            u.IN = pick()
        elif kind == 7:
            u = d.FixedDelay(rng.choice([0.0002, 0.003, 0.01]))
            u.IN = pick()
        elif kind == 8:
            u = d.Shape(rng.choice(["decay", "attack", "semiSine", "decaySquared"]), const_or_signal(0.004, 0.05) if rng.random() < 0.5 else 0.01,
                        rng.uniform(-1, 0), rng.uniform(0.5, 2)).trigger()
        elif kind == 9:
            u = d.AHD(rng.cuniform(0.001, 0.01), rng.cuniform(0, 0.01), rng.cuniform(0.002, 0.02)).trigger()
        elif kind == 10:
            u = d.Multiply(d.Timer(), rng.cchoice([2, 100]))
        elif kind == 11:
            u = d.SampleRateRedux(pick(), const_or_signal(2, 40))
        elif kind == 12:
            u = d.Pan(pick(), const_or_signal(-1, 1))
        elif kind == 13:
            u = d.CrossFader(pick(), pick(), const_or_signal(0, 1))
        elif kind == 14:
            u = d.MultiChannelOsc(d.Sum(d.Multiply(pick(), [30, -60]), [200, 300.5]))
        elif kind == 15:  # a CircleBuffer with two taps, one of them fed back
            buf = d.CircleBuffer(1, rng.choice([0.02, 0.05, 0.003]))  # (0.003 s: a ring shorter than a chunk)
            w = d.CircleBufferWriter(buf)
            w.preWipe = rng.random() < 0.7
            w.IN = pick()
            if rng.random() < 0.3:  # a moving tap: a signal-rate offset
                tap = d.CircleBufferReader(buf, d.Sum(d.Multiply(pick(), rng.cchoice([0.002, 0.0004])), rng.cchoice([0.006, 0.011])))
                tap.postWipe = rng.random() < 0.3
            else:
                tap = d.CircleBufferReader(buf, rng.cchoice([0.006, 0.011]))
            tap.chain(w)
            if rng.random() < 0.5:
                fb = d.CircleBufferWriter(buf, 0.004)
                fb.IN = d.Multiply(tap, 0.4)
                fb.chain(w)
            u = tap
        else:
            u = d.Subtract(pick(), d.Abs(pick()))
        pool.append(u)
    out = pool[-1]
    if feedback_sum is not None:
        feedback_sum.B = d.Multiply(out, rng.cuniform(-0.5, 0.5))
    return out


class _Split:
    """Structure decisions from one stream, inlet constants from another."""

    def __init__(self, structure, constants):
        self.s, self.c = structure, constants

    def choice(self, xs): return self.s.choice(xs)
    def random(self): return self.s.random()
    def randint(self, a, b): return self.s.randint(a, b)
    def randrange(self, n): return self.s.randrange(n)
    def uniform(self, a, b): return self.s.uniform(a, b)
    def cchoice(self, xs): return self.c.choice(xs)
    def cuniform(self, a, b): return self.c.uniform(a, b)


@pytest.mark.parametrize("seed", range(40))
def test_random_batch_wave_equals_chunk(seed):
    """The same, as a batch: one random structure, 5 voices that differ in their inlet constants (-> per-instance
    parameters), rendered as ONE program by both engines."""
    d.configure(48000)
    try:
        exs = [descriptor.extract(random_circuit(random.Random(seed), random.Random(1000 * seed + k))) for k in range(5)]
        uni = descriptor.unify(exs)
    except (RecursionError, descriptor.DuspError):
        pytest.skip("degenerate graph")
    ctx = render.context(48000)
    progs = []
    for engine in (runtime.ENGINE_CHUNK, runtime.ENGINE_WAVE):
        try:
            progs.append(ctx.build(uni.words, engine))
        except runtime.DuspHipError as e:
            assert e.status == -2
            for p in progs:
                p.close()
            pytest.skip(e.message)
    n = 256 * 6 + 77
    want = progs[0].render(n, uni.n_instances, uni.params)
    got = progs[1].render(n, uni.n_instances, uni.params)
    assert np.array_equal(got, want, equal_nan=True)
    for p in progs:
        p.close()


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("DUSP_FUZZ_SEEDS", "100"))))
def test_random_circuit_wave_equals_chunk(seed):
    rng = random.Random(seed)
    d.configure(48000)
    try:
        ex = descriptor.extract(random_circuit(rng))
    except RecursionError:
        pytest.skip("degenerate graph")
    # (DUSP_FILTER_SCAN=0: Filters through the Filter stage, whose bits are the chunk engine's; the scan form of high constant cutoffs
    # is tolerance-level by design — checked against this render below)
    ctx = knob_context(48000, DUSP_FILTER_SCAN=0)
    try:
        chunk = ctx.build(ex.words, runtime.ENGINE_CHUNK)
    except runtime.DuspHipError as e:
        assert e.status == -2  # e.g. channel counts growing through the feedback edge
        pytest.skip(e.message)
    try:
        wave = ctx.build(ex.words, runtime.ENGINE_WAVE)
    except runtime.DuspHipError as e:
        assert e.status == -2
        chunk.close()
        pytest.skip("not a wave-engine graph: " + e.message)
    n = 256 * rng.randint(3, 12) + rng.choice([0, 1, 130])
    want = chunk.render(n)
    got = wave.render(n)
    assert np.array_equal(got, want, equal_nan=True), "PCM differs first at sample %d" % int(np.argmax((got != want).any(axis=(0, 1))))
    for u in range(chunk.n_units):
        a, b = wave.state(u), chunk.state(u)
        if a.size == b.size and chunk.n_units:  # (Delay's engine-internal slot is not unit state)
            assert np.array_equal(a, b, equal_nan=True), (u, a, b)
    # the same circuit as a continued chain on the wave engine: two segments == one shot
    chain = ctx.build(ex.words, runtime.ENGINE_WAVE | runtime.ENGINE_RESUMABLE)
    cut = 256 * max(1, (n // 256) // 2)
    first = chain.render(cut)
    chain.continue_with(descriptor.continued(ex.words, cut, [chain.state(u) for u in range(chain.n_units)]))
    second = chain.render(n - cut)
    assert np.array_equal(np.concatenate([first, second], axis=2), want, equal_nan=True)
    # the default knobs: a Filter whose deviation only passes through sums, products and delay lines may run as a scan
    scan = render.context(48000).build(ex.words, runtime.ENGINE_WAVE)
    got = scan.render(n).astype(np.float64)
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin)
    if fin.any():
        assert float(np.max(np.abs(got[fin] - want[fin]))) <= 1e-5 * max(1.0, float(np.max(np.abs(want[fin]))))
    for p in (chunk, wave, chain, scan):
        p.close()


LARGE = [(17, 1100), (23, 2500), (29, 700), (31, 4200), (41, 300), (43, 900), (47, 600), (53, 1300), (59, 2100), (61, 4500), (67, 8192), (71, 5000)]
# DUSP_FUZZ_BATCHES=n: n more structures at batch sizes that fill 16-wavefront workgroups (a soak run, not the default)
LARGE += [(100 + s, 4096 + 37 * s) for s in range(int(__import__("os").environ.get("DUSP_FUZZ_BATCHES", "0")))]


@pytest.mark.parametrize("seed,n_inst", LARGE)
def test_random_large_batch_wave_equals_chunk(seed, n_inst):
    """Hundreds to thousands of instances: the wave engine packs 2-16 wavefronts per workgroup (shared table image,
    cooperative Filter stage, surplus waves in the last workgroup); still bit-identical to the chunk engine."""
    d.configure(48000)
    # two real extractions fix the structure and show which constants vary; the parameter table is then drawn directly
    exs = [descriptor.extract(random_circuit(random.Random(seed), random.Random(7000 * seed + k))) for k in range(2)]
    try:
        uni = descriptor.unify(exs)
    except descriptor.DuspError:
        pytest.skip("degenerate graph")
    if not uni.n_params:
        pytest.skip("no varying constant")
    rng = np.random.RandomState(seed)
    lo, hi = uni.params.min(axis=1, keepdims=True), uni.params.max(axis=1, keepdims=True)
    params = (lo + (hi - lo) * rng.rand(uni.n_params, n_inst)).astype(np.float32)
    ctx = knob_context(48000, DUSP_FILTER_SCAN=0)  # (Filters through their stage: the chunk engine's bits; the scan form is checked below)
    progs = []
    for engine in (runtime.ENGINE_CHUNK, runtime.ENGINE_WAVE):
        try:
            progs.append(ctx.build(uni.words, engine))
        except runtime.DuspHipError as e:
            assert e.status == -2
            for p in progs:
                p.close()
            pytest.skip(e.message)
    n = 256 * 4 + 9
    want = progs[0].render(n, n_inst, params)
    got = progs[1].render(n, n_inst, params)
    assert np.array_equal(got, want, equal_nan=True)
    progs.append(render.context(48000).build(uni.words, runtime.ENGINE_WAVE))  # the default knobs: a Filter may run as a scan (tolerance-level)
    got = progs[2].render(n, n_inst, params).astype(np.float64)
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin)
    if fin.any():
        assert float(np.max(np.abs(got[fin] - want[fin]))) <= 1e-5 * max(1.0, float(np.max(np.abs(want[fin]))))
    for p in progs:
        p.close()
