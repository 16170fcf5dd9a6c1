"""GPU parity: the HIP path (through the C ABI) against the reference's golden vectors and the oracle.

Bar: bit-exact wherever the device executes the same IEEE operations as JS (everything except the
Filter's tan() and the pow() of the dB / semitone / Pow units, which come from the device math library):
there the north star's tolerance applies,
1e-5 relative to full scale (SURVEY.md §7 "tolerance definition").
"""
import numpy as np
import pytest

from conftest import ALL_GOLDEN, Golden, knob_context
from dusp_amd import render, runtime

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5  # north_star: "within 1e-5 relative float tolerance" (of full scale)
# device tan() in the Filter coefficients, device pow() in Gain / DecibelToScaler / SemitoneToRatio / Pow
# ... and in Pan's compensation gain / MidiToFrequency
USES_DEVICE_TAN = ("loop_", "filter_", "map_gain", "map_db_semitone", "map_pow", "map_fm_semitone", "rest_pan", "rest_midi",
                   "grow_feedback_filter")  # (... whose 2 kHz Filter runs as a scan on the compiled kernel: tolerance-level by design, JitFilterScan)


# |f| < 2^-13: the reference's own f64 phase accumulation rounds there (SURVEY.md §8a note ii), so the wave
# engine's 2^-36 fixed-point scan is only within tolerance for it (the fused and chunk engines stay exact)
BELOW_EXACT_REGIME = {("osc_f_tiny", "wave"), ("osc_f_tiny", "interp")}


def check(name, got, ref, engine=None):
    assert got.shape == ref.shape
    if name.startswith(USES_DEVICE_TAN) or (name, engine) in BELOW_EXACT_REGIME:
        scale = max(1e-30, float(np.max(np.abs(ref))))
        err = float(np.max(np.abs(got.astype(np.float64) - ref.astype(np.float64))))
        assert err <= REL_TOL * scale, "max abs err %.3g vs full scale %.3g" % (err, scale)
    else:
        assert np.array_equal(got, ref), "first mismatch at %d" % int(np.argmax(got != ref))


# "interp": the wave engine's interpreter kernel (DUSP_WAVE_JIT=0) — what renders a new structure's short first renders under the
# default knob and whatever the circuit compiler refuses; "wave" (with the suite's DUSP_WAVE_JIT=2) is the compiled kernels
ENGINES = {"auto": runtime.ENGINE_AUTO, "chunk": runtime.ENGINE_CHUNK, "wave": runtime.ENGINE_WAVE, "interp": runtime.ENGINE_WAVE}


def engine_context(sample_rate, engine):
    return knob_context(sample_rate, DUSP_WAVE_JIT=0) if engine == "interp" else render.context(sample_rate)
# what AUTO must pick for a few cases (fused voice shapes / feed-forward wave engine / universal chunk engine)
EXPECTED_ENGINE = {"osc440_1s": "fused", "voice3_k7": "fused", "summany_1024": "fused", "cfg2_sweep": "wave",
                   "cfg2_literal": "wave", "fm_mixed": "wave", "fm_sum": "wave", "mult_2ch": "wave", "ramp_300": "wave",
                   "loop_220": "wave", "loop_110p5_short": "wave", "loop_frac_delay": "wave", "loop_220_sr44100": "wave",
                   "delay_mod": "wave", "delay_2ch": "wave", "circlebuffer_taps": "wave", "circlebuffer_2ch": "wave", "circlebuffer_moving_tap": "wave", "circlebuffer_moving_writer": "wave", "circlebuffer_short_ring": "wave",
                   "rest_crossfader": "wave", "rest_rescale_2ch": "wave", "rest_vecmag": "wave", "rest_concat": "wave", "rest_pick": "wave",
                   "rest_timer_fm": "wave", "rest_srr_mod": "wave", "env_shape_mod": "wave", "env_ahd_mod": "wave", "fam_comb": "wave", "fam_allpass_loop": "wave", "fam_multiosc_fm": "wave", "fam_multiosc_negative": "wave", "fam_monodelay": "wave", "fam_readback": "wave",
                   "filter_2ch": "wave", "filter_lp_mod": "wave", "filter_hp": "wave", "delay_default": "wave", "delay_wrap": "wave"}


@pytest.mark.parametrize("engine", ["auto", "chunk", "wave", "interp"])
@pytest.mark.parametrize("name", ALL_GOLDEN)
def test_render_matches_reference_golden(name, engine, oracle):
    g = Golden(name)
    ctx = engine_context(g.sample_rate, engine)
    try:
        prog = ctx.build(g.desc, ENGINES[engine])
    except runtime.DuspHipError as e:
        assert engine in ("wave", "interp") and e.status == -2, e  # the wave engine refuses by regime
        pytest.skip("graph shape not handled by this engine")
    if engine == "auto" and name in EXPECTED_ENGINE:
        assert prog.engine == EXPECTED_ENGINE[name]
    assert prog.n_out_channels == g.n_channels
    pcm = prog.render(g.n_samples)[0]
    if engine == "interp":
        assert "compiled kernel" not in prog.read_shape()
    check(name, g.windowed(pcm), g.pcm, engine)            # vs the JS reference's own output
    check(name, pcm, oracle.render(g.desc, g.n_samples), engine)  # vs the oracle, full length
    prog.close()


@pytest.mark.parametrize("name", ["osc_f_440p5", "voice3_k7", "ramp_300", "loop_220", "circlebuffer_taps", "cfg2_sweep", "fm_sum",
                                  "rest_timer_fm", "rest_srr_mod", "rest_srr_nan", "rest_srr", "rest_vecmag_2d",
                                  "env_shape_mod", "env_shape_edges", "env_ahd_mod", "env_ahd_zero_hold", "env_ahd"])
@pytest.mark.parametrize("engine", ["auto", "chunk", "wave", "interp"])
def test_state_write_back_matches_oracle(name, engine, oracle):
    g = Golden(name)
    n = min(g.n_samples, 5000)
    try:
        prog = engine_context(g.sample_rate, engine).build(g.desc, ENGINES[engine])
    except runtime.DuspHipError:
        pytest.skip("not a feed-forward graph")
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


@pytest.mark.parametrize("name", ALL_GOLDEN)
def test_segmented_render_equals_one_shot(name, oracle):
    """dusp_program_continue: render every golden case in three segments — state written back into the descriptor
    between them, delay lines / CircleBuffers / feedback chunks staying on the device — and require the PCM of
    the one-shot render, bit for bit (same engine family: segment boundaries must not be observable)."""
    from dusp_amd import descriptor
    g = Golden(name)
    n = g.n_samples
    if n < 3 * 256 + 1:
        pytest.skip("shorter than three segments")
    ctx = render.context(g.sample_rate)
    whole = ctx.build(g.desc, runtime.ENGINE_RESUMABLE)  # same engine choice as the segmented chain (wave where it can, else chunk)
    want = whole.render(n)[0]
    engine = whole.engine
    whole.close()
    cuts = [0, 256, 256 * (1 + (n // 256) // 2), n]
    prog = ctx.build(g.desc, runtime.ENGINE_RESUMABLE)
    parts, words, shapes = [], g.desc, []
    for a, b in zip(cuts[:-1], cuts[1:]):
        if a:
            words = descriptor.continued(words, a, [prog.state(u) for u in range(prog.n_units)])
            prog.continue_with(words)
        parts.append(prog.render(b - a)[0])
        shapes.append(prog.read_shape())
    got = np.concatenate(parts, axis=1)
    prog.close()
    # circuits with delay lines / feedback are continued on their compiled kernels too (outlets parked between launches, exact rings)
    if engine == "wave" and name.startswith(("loop_", "delay_", "fam_", "circlebuffer_")):
        assert all("compiled kernel" in sh for sh in shapes), shapes
    check(name, want, oracle.render(g.desc, n), engine)  # ... and the chain's engine renders this case correctly in the first place
    assert np.array_equal(got, want), "first mismatch at sample %d" % int(np.argmax((got != want).any(axis=0)))


@pytest.mark.parametrize("n_seg", [2, 5, 64])
@pytest.mark.parametrize("name", ALL_GOLDEN)
def test_time_split_wave_render_equals_unsplit(name, n_seg):
    """Wave engine, time-split mode (few instances, long render): every instance is cut into segments rendered by
    separate wavefronts after one accumulate + prefix pass per FM level.  Oscillator phases are exact modular sums,
    so PCM and written-back state must equal the unsplit render bit for bit, whatever the number of segments."""
    g = Golden(name)
    try:
        prog = knob_context(g.sample_rate, DUSP_WAVE_SEGMENTS=1).build(g.desc, runtime.ENGINE_WAVE)
    except runtime.DuspHipError:
        pytest.skip("not a wave-engine graph")
    want = prog.render(g.n_samples)[0]
    want_state = [prog.state(u) for u in range(prog.n_units)]
    prog.close()
    prog = knob_context(g.sample_rate, DUSP_WAVE_SEGMENTS=n_seg).build(g.desc, runtime.ENGINE_WAVE)
    got = prog.render(g.n_samples)[0]
    got_state = [prog.state(u) for u in range(prog.n_units)]
    prog.close()
    assert np.array_equal(got, want), "first mismatch at sample %d" % int(np.argmax((got != want).any(axis=0)))
    for a, b in zip(got_state, want_state):
        assert np.array_equal(a, b, equal_nan=True)


def test_filter_circuits_in_segments_that_warm_up_equal_the_unsplit_render():
    """ONE circuit with Filters, cut in time (jit_codegen.hpp jit_warm_chunks): every segment starts a segment early, from rest, stores only its
    own chunks, and the host checks that every Filter stage held at a segment's start what the segment before ended with — else the render
    is finished sequentially from the last good segment.  Either way the PCM and the written-back state are the unsplit render's, bit for
    bit.  Every golden with a Filter that the form applies to, with segments of only two chunks (DUSP_FILTER_WARM=2: far less warm-up
    than the Filters need, so both outcomes of the check occur)."""
    split, redone = [], []
    for name in ALL_GOLDEN:
        g = Golden(name)
        if not any(u.startswith("Filter") for u in g.meta.get("reference_unit_order", [])) or g.n_samples < 256 * 4:
            continue
        try:
            prog = knob_context(g.sample_rate, DUSP_FILTER_WARM=0, DUSP_FILTER_SCAN=0).build(g.desc, runtime.ENGINE_WAVE)  # (the one long chain, on the Filter stage)
        except runtime.DuspHipError:
            continue
        if prog.n_params or prog.n_inputs:
            prog.close()
            continue
        want = prog.render(g.n_samples)[0]
        want_state = [prog.state(u) for u in range(prog.n_units)]
        assert " seg" not in prog.read_shape()
        prog.close()
        prog = knob_context(g.sample_rate, DUSP_FILTER_WARM=2, DUSP_FILTER_SCAN=0).build(g.desc, runtime.ENGINE_WAVE)  # (circuits the form does not apply to: the same chain)
        got = prog.render(g.n_samples)[0]
        shape = prog.read_shape()
        got_state = [prog.state(u) for u in range(prog.n_units)]
        prog.close()
        assert np.array_equal(got, want), (name, shape, int(np.argmax((got != want).any(axis=0))))
        for a, b in zip(got_state, want_state):
            assert np.array_equal(a, b, equal_nan=True), (name, shape)
        if " seg" in shape:
            split.append(name)
            if "redo@" in shape:
                redone.append(name)
    print("cut into warming segments:", split, "finished sequentially:", redone)
    assert len(split) >= 2, split


def test_continue_refuses_a_different_circuit_and_a_wrong_clock():
    from dusp_amd import descriptor
    ctx = render.context(48000)
    g = Golden("loop_220")
    prog = ctx.build(g.desc, runtime.ENGINE_RESUMABLE)
    with pytest.raises(runtime.DuspHipError, match="nothing has been rendered"):
        prog.continue_with(g.desc)
    prog.render(512)
    states = [prog.state(u) for u in range(prog.n_units)]
    with pytest.raises(runtime.DuspHipError, match="does not follow the rendered clock"):
        prog.continue_with(descriptor.continued(g.desc, 256, states))
    with pytest.raises(runtime.DuspHipError, match="does not describe the circuit"):
        prog.continue_with(Golden("fam_comb").desc)
    prog.continue_with(descriptor.continued(g.desc, 512, states))
    prog.close()
    plain = ctx.build(g.desc)  # not resumable: a circuit with a delay line cannot be continued
    plain.render(512)
    with pytest.raises(runtime.DuspHipError, match="DUSP_ENGINE_RESUMABLE"):
        plain.continue_with(descriptor.continued(g.desc, 512, states))
    plain.close()
    with pytest.raises(runtime.DuspHipError, match="already been ticked"):
        ctx.build(descriptor.continued(g.desc, 512, states))  # a fresh program cannot know the ring contents


def test_malformed_descriptors_are_rejected_not_crashed():
    """Truncations and single-word corruptions of real descriptors: program_build must either accept the
    program or fail with a status + message — never crash, hang or read out of bounds."""
    rng = np.random.RandomState(7)
    ctx = render.context(48000)
    poison = [np.nan, np.inf, -1.0, 0.5, 1e18, -1e18, 3.0, 65536.0, 2.0 ** 40]
    built = rejected = 0
    for name in ("cfg2_sweep", "loop_220", "circlebuffer_2ch", "filter_2ch", "summany_8", "map_db_semitone", "rest_rescale_2ch",
                 "rest_srr_mod", "rest_pick", "rest_vecmag", "env_shape_mod", "env_ahd_amp"):
        g = Golden(name)
        trials = [g.desc[:k] for k in range(0, g.desc.size, 3)]
        for _ in range(150):
            d2 = g.desc.copy()
            d2[rng.randint(d2.size)] = poison[rng.randint(len(poison))]
            trials.append(d2)
        for words in trials:
            try:
                prog = ctx.build(words)
                prog.close()
                built += 1
            except runtime.DuspHipError as e:
                assert e.status in (-1, -2, -4) and e.message
                rejected += 1
    assert rejected > 300 and built > 50


OSC_GOLDEN = [n for n in ALL_GOLDEN if n.startswith(("osc", "voice", "cfg2", "fm_", "summany_8", "mult_", "map_invert", "rest_timer_fm", "fam_multiosc"))]


@pytest.mark.parametrize("knobs", [{"DUSP_JIT_LDS_TABLE": 0}, {"DUSP_JIT_LEAN": 0}, {"DUSP_FUSED_FX32": 3}, {"DUSP_FUSED_FX32": 2}, {"DUSP_FUSED_TABLE": "global"}],
                         ids=["jit-gather", "jit-fx", "fused-lean", "fused-plain", "fused-gather"])
def test_the_oscillators_other_forms_render_the_same_pcm(knobs, oracle):
    """The oscillator's lerp has several forms of the same arithmetic, chosen by table, phase grid and knob: the delta form
    over the LDS image (the default) or over gathers from L2 (no image: DUSP_JIT_LDS_TABLE=0, DUSP_FUSED_TABLE=global), the 32.32 form
    (DUSP_JIT_LEAN=0), the fused engine's earlier lean and plain forms (DUSP_FUSED_FX32=3 / 2).  Every oscillator golden, bit for bit."""
    seen = 0
    for name in OSC_GOLDEN:
        g = Golden(name)
        ctx = knob_context(g.sample_rate, **knobs)
        for engine in (runtime.ENGINE_AUTO, runtime.ENGINE_WAVE):
            try:
                prog = ctx.build(g.desc, engine)
            except runtime.DuspHipError as e:
                assert e.status == -2, e
                continue
            pcm = prog.render(g.n_samples)[0]
            check(name, g.windowed(pcm), g.pcm, "wave" if engine == runtime.ENGINE_WAVE else None)
            prog.close()
            seen += 1
    assert seen >= 2 * len(OSC_GOLDEN) - 8


RING_GOLDEN = [n for n in ALL_GOLDEN if n.startswith(("delay_", "loop_", "circlebuffer_", "fam_", "patch_"))]


@pytest.mark.parametrize("engine", ["auto", "chunk", "wave", "interp"])
def test_rings_are_zeroed_wherever_a_render_can_touch_them(engine):
    """A render nothing continues zero-fills only the part of each delay ring it can touch (dusp_abi.hip zero_rings: a five-second
    default Delay line is mostly out of a short render's reach).  DUSP_RING_POISON=1 fills the rings with NaN patterns first, so a window
    cut too short shows in the PCM: every golden with a ring, alone and as a batch of 70 (two rows of instances), bit for bit against
    the reference's own output, and the same with the windows switched off."""
    seen = 0
    for name in RING_GOLDEN:
        g = Golden(name)
        knobs = {"DUSP_RING_POISON": 1}
        if engine == "interp":
            knobs["DUSP_WAVE_JIT"] = 0
        for window in ((1, 0) if engine == "auto" else (1,)):  # (windows off — every ring zero-filled whole — is one code path for all engines: once)
            ctx = knob_context(g.sample_rate, DUSP_RING_WINDOW=window, **knobs)
            try:
                prog = ctx.build(g.desc, ENGINES[engine])
            except runtime.DuspHipError as e:
                assert engine in ("wave", "interp") and e.status == -2, e
                continue
            if prog.n_params or prog.n_inputs:
                prog.close()
                continue
            pcm = prog.render(g.n_samples, 70)
            for i in (0, 63, 64, 69):
                check(name, g.windowed(pcm[i]), g.pcm, engine)
            prog.close()
            seen += 1
    assert seen >= 10 * (2 if engine == "auto" else 1)
