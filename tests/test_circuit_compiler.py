"""The circuit compiler without a GPU (dusp_amd/csrc/jit_codegen.hpp + jit_engine.hip through dusp_circuit_kernel_source):
one kernel text per circuit STRUCTURE, constants outside the text, hiprtc compiles it for gfx950 in process.
What the kernels compute is the GPU tests' business (tests/test_gpu_parity.py renders every golden on them)."""
import re

import numpy as np
import pytest

import dusp_amd as d
from conftest import ALL_GOLDEN, Golden
from dusp_amd import descriptor, runtime

# a spread of what the compiler takes: plain voices, FM (scanned oscillators + accumulate passes), feedback edges, Filter, Delay, maps
COMPILED = ["osc440_1s", "voice3_k7", "cfg2_sweep", "fm_mixed", "fm_sum", "mult_2ch", "loop_220", "filter_2ch", "filter_hp", "delay_default",
            "map_gain", "rest_crossfader", "rest_timer_fm", "osc_triangle",
            # units with a sequential stage (one lane walks the chunk out of the wave's LDS scratch)
            "fam_allpass_series", "fam_comb_mod", "env_ahd_mod", "rest_srr_mod", "fam_multiosc_negative", "env_shape_semisine_amp",
            "env_shape_mod", "filter_lp_mod", "circlebuffer_taps", "circlebuffer_2ch",
            # ordered slot operations: delay lines and CircleBuffer nodes whose accesses can meet inside a chunk
            "delay_mod", "fam_monodelay_mod", "fam_readback_frac", "circlebuffer_moving_tap", "circlebuffer_short_ring", "loop_110p5_short"]


def source(words, **kw):
    return runtime.circuit_kernel_source(words, **kw)


@pytest.mark.parametrize("name", COMPILED)
def test_golden_circuits_compile_for_gfx950(name):
    g = Golden(name)
    text = source(g.desc, waves=4, compile=True)
    assert "dusp_jit_render" in text and '#include "jit_prelude.hpp"' in text
    # one statement block per channel-expanded unit, in process order: every outlet buffer is declared exactly once per kernel
    render = text.split('extern "C"')[1]
    # (the chunk loop stands twice in a kernel: constant-f oscillators in 32.32 fixed point / in the general form)
    bufs = re.findall(r"float (v\d+_\d+)\[4\];", render)
    assert len(bufs) == 2 * len(set(bufs)) >= 2


def test_what_the_compiler_takes_and_what_stays_on_the_interpreter():
    taken, refused = 0, {}
    for name in ALL_GOLDEN:
        try:
            source(Golden(name).desc, waves=1)
            taken += 1
        except runtime.DuspHipError as e:
            assert e.status == -2, e
            refused[name] = e.message
    assert taken >= 157 and len(refused) <= 2
    assert "channel counts grow" in refused["patch_scary"]
    assert "DUSP_JIT_MAX_UNITS" in refused["summany_1024"]


def test_kernel_text_depends_on_structure_not_on_constants():
    """Constants travel in arrays next to the kernel, so every circuit of one shape — and every segment of an
    event-segmented render — reuses one compiled kernel."""
    d.configure(48000)

    def voice(f, mod, depth, gain, length):
        return d.Multiply(d.Osc(d.Sum(d.Multiply(d.Osc(mod), depth), f)), d.Multiply(d.Ramp(length, 1, 0).trigger(), gain))

    a = source(descriptor.extract(voice(220, 5, 40, 0.5, 48000)).words)
    b = source(descriptor.extract(voice(331.25, 0.75, 3, 0.125, 9000)).words)
    assert a == b
    assert "220" not in a and "0.5" not in a.replace("0.5)", "")  # (no literal constants of the circuit in the text)
    c = source(descriptor.extract(d.Multiply(d.Osc(d.Sum(d.Multiply(d.Osc(5, "saw"), 40), 220)), d.Multiply(d.Ramp(48000, 1, 0).trigger(), 0.5))).words)
    assert c != a  # another wave table is another structure
    # a per-instance parameter instead of a constant is another structure too (pN instead of kN)
    uni = descriptor.unify([descriptor.extract(voice(f, 5, 40, 0.5, 48000)) for f in (220, 330)])
    p = source(uni.words)
    assert p != a and "jit_param(A, X[0], 0)" in p


def test_fm_levels_get_their_accumulate_passes():
    """Time-split rendering needs every scanned (FM) oscillator's phase total per segment: one pass kernel per FM level,
    holding only the cone of units that feeds that level's oscillators."""
    d.configure(48000)
    inner = d.Osc(d.Sum(d.Multiply(d.Osc(3), 20), 110))        # level 1
    outer = d.Osc(d.Sum(d.Multiply(inner, 50), 440))            # level 2
    text = source(descriptor.extract(d.Multiply(outer, d.Ramp(48000, 1, 0).trigger())).words)
    assert "dusp_jit_pass1" in text and "dusp_jit_pass2" in text and "dusp_jit_pass0" not in text
    pass1 = text.split("dusp_jit_pass1")[1].split('extern "C"')[0]
    assert "jit_ramp" not in pass1 and "jit_store" not in pass1   # neither the envelope nor the outlet is in the cone
    assert pass1.count("JitOscS") == 1                            # the level-2 oscillator is not needed to total level 1
    # a circuit with a Filter cannot be cut in time: no pass kernels
    assert "dusp_jit_pass" not in source(descriptor.extract(d.Filter(inner, 800)).words)


def test_feedback_edges_read_last_iterations_registers(monkeypatch):
    monkeypatch.setenv("DUSP_FILTER_SCAN", "0")  # (the Filter stage's form; the scan form: test_filters_with_a_constant_cutoff_run_as_a_scan)
    g = Golden("loop_220")
    text = source(g.desc, waves=8)
    late = set(re.findall(r"float (w\d+_0)\[4\] = \{0\.f", text))
    assert len(late) == 1                                        # the loop's one back edge: the Filter's previous chunk
    w = late.pop()
    assert re.search(r"%s\[c\] = v%s\[c\]" % (w, w[1:]), text)   # carried over at the end of the iteration
    assert "JitFilterK<8, 1, 256, 1, " in text and "JitDelayK" in text


def test_work_that_does_not_hang_on_a_filter_runs_beside_its_recurrences(monkeypatch):
    """(DUSP_FILTER_SCAN=0: the form of cutoffs below the scan's bound, per-instance cutoffs and continued programs.)  The Filter stage keeps one wave busy while the others would wait.  The generated chunk body therefore sorts its units: what the
    Filter's input needs runs a chunk AHEAD (`early`, where nothing else reads it), what neither feeds a Filter nor hangs on one runs
    in a `side` block, both between the barriers of a sub-block on the waves that do not serve it; waves 0 and 1 take turns serving."""
    monkeypatch.setenv("DUSP_FILTER_SCAN", "0")
    d.configure(48000)
    def loop(k):   # BASELINE configs[3]: the Delay's reads feed the Filter, everything else only reaches the ring
        s = d.Sum(d.Osc(110 + k / 64), 0)
        f = d.Filter(d.Delay(s, 480, 4096), 2000)
        s.B = d.Multiply(f, 0.5)
        return f
    uni = descriptor.unify([descriptor.extract(loop(k)) for k in (0, 64)])
    fast = source(uni.words, waves=16, per_wave=2, compile=True).split("} else {")[0]
    side = fast.split("auto side0 = [&]()")[1].split("};")[0]
    assert ".tick<1, 0, 2>" in side and side.count(".write(") == 2 and ".read(" not in side      # oscillators and the ring writes of both instances
    early = fast.split("auto early = [&](uint32_t g)")[1].split("};")[0]
    assert early.count(".read(") == 2 and early.count("f4.feed(") == 2 and ".write(" not in early     # ring reads and feed-forward halves, a chunk ahead
    assert "if (g == X[0].g_begin) early(g);" in fast and fast.count("if (g + 1 < X[0].g_end) early(g + 1);") == 2
    assert "f4.serial<8>(X[0], tile, 0);" in fast and "f4.serial<8>(X[0], tile, 1);" in fast         # two sub-blocks, two serving waves
    assert fast.count("u >> X[0].wave) & 1u) { side0();") == 2  # (each wave works in one window it does not serve)
    # a voice whose oscillator also reaches the output: it still runs ahead, into registers of its own that become the chunk's at its top
    dry = descriptor.unify([descriptor.extract((lambda o: d.Sum(d.Filter(o, 900), o))(d.Osc(200 + k))) for k in (0, 8)])
    fast = source(dry.words, waves=16, per_wave=2, compile=True).split("} else {")[0]
    early = fast.split("auto early = [&](uint32_t g)")[1].split("};")[0]
    assert early.count(", vn0_") == 4 and "v0_0" not in early.replace("vn0_0", "")                   # written and fed to the Filter as vn
    assert "for (int c = 0; c < 4; ++c) v0_0[c] = vn0_0[c];" in fast and "+ v0_0[c]" in fast                # ... and read by the sum as v
    # Filters in series (a 4-pole filter): the second stage is fed in place, behind the first one's last sub-block, with whatever stands
    # between the two; the oscillator and the first stage's feed-forward half still run a chunk ahead, dealt over all four windows
    fast = source(descriptor.unify([descriptor.extract(d.Filter(d.Multiply(d.Filter(d.Osc(200 + k), 900), 0.8), 1200)) for k in (0, 8)]).words,
                  waves=16, per_wave=2, compile=True).split("} else {")[0]
    early = fast.split("auto early = [&](uint32_t g)")[1].split("};")[0]
    assert early.count("f1.feed(") == 2 and "f3.feed(" not in early
    tail = fast.split("};")[-1]
    assert tail.index("f1.serial<8>(X[0], tile, 1)") < tail.index("* k1") < tail.index("f3.feed(") < tail.index("f3.serial<8>(X[0], tile, 0)")
    assert fast.count("u >> X[0].wave) & 1u) { side0();") == 4


def test_filters_with_a_constant_cutoff_run_as_a_scan():
    """A Filter whose cutoff is a constant of the circuit, high enough for the bound of jit_codegen.hpp jit_filter_scan_ok (the all-pole
    part's impulse response sums to at most 30: between about 1.5 and 22.5 kHz at 48 kHz), is a unit like any other — JitFilterScan, a scan over the
    chunk's lanes — and the circuit has no Filter stage: no tile, no barriers, no serving wave.  Lower cutoffs, per-instance cutoffs,
    connected cutoffs and continued programs keep the stage."""
    d.configure(48000)
    def loop(k, cutoff=2000):
        s = d.Sum(d.Osc(110 + k / 64), 0)
        f = d.Filter(d.Delay(s, 480, 4096), cutoff)
        s.B = d.Multiply(f, 0.5)
        return f
    uni = descriptor.unify([descriptor.extract(loop(k)) for k in (0, 64)])
    text = source(uni.words, waves=16, per_wave=1, compile=True)
    assert "JitFilterScanK fk4;" in text and "JitFilterScan f4_0;" in text and "f4_0.tick(X[0], fk4, v3_0, v4_0);" in text
    assert "JitFilterK<" not in text and "tile" not in text.split('extern "C"')[1] and "jit_lds_barrier" not in text
    assert "f4_0.end(A, X[0], fk4, 2);" in text                                             # state write-back in the chunk engine's slots
    for build in (lambda k: loop(k, 600),                                                    # below the bound: bit for bit through the stage
                  lambda k: loop(k, 2000 + k),                                               # a per-instance cutoff
                  lambda k: d.Filter(d.Osc(200 + k), d.Sum(d.Multiply(d.Osc(5), 500), 3000)),  # a connected one
                  lambda k: d.Filter(d.Filter(d.Osc(200 + k), 3000), 500)):                  # one of two below the bound: both stay
        text = source(descriptor.unify([descriptor.extract(build(k)) for k in (0, 64)]).words, waves=16, per_wave=1)
        assert "JitFilterScan" not in text and "JitFilterK" in text
    assert "JitFilterScan" not in source(uni.words, waves=16, per_wave=1, continued=True)
    # what hangs on the Filter must pass its deviation on as it is (sums, products, delay lines' signal inlets): a Filter that reaches
    # an oscillator's frequency, a delay time or a division keeps its stage
    for build in (lambda k: d.Osc(d.Sum(d.Multiply(d.Filter(d.Osc(3 + k), 3000), 40), 440)),
                  lambda k: d.Delay(d.Osc(200 + k), d.Sum(d.Multiply(d.Filter(d.Osc(3), 3000), 100), 300), 4096),
                  lambda k: d.Divide(d.Osc(200 + k), d.Sum(d.Filter(d.Osc(3), 3000), 2))):
        assert "JitFilterScan" not in source(descriptor.unify([descriptor.extract(build(k)) for k in (0, 64)]).words, waves=16, per_wave=1)
    wet = lambda k: d.Sum(d.Multiply(d.Delay(d.Filter(d.Osc(200 + k), 3000), 700.5, 4096), 0.4), d.Osc(330))
    assert "JitFilterScan" in source(descriptor.unify([descriptor.extract(wet(k)) for k in (0, 64)]).words, waves=16, per_wave=1)
    hp = source(descriptor.extract(d.Filter(d.Filter(d.Osc(300), 3000), 5000, "HP")).words, waves=4, compile=True)  # 4 poles: two scans
    assert hp.count("JitFilterScanK fk") == 2


def test_per_instance_cutoffs_scan_when_their_column_allows(monkeypatch):
    """A cutoff SWEEP — the Filter's cutoff a per-instance parameter — runs as a scan too when the renderer has looked at the column
    (one small launch next to the Delays' regimes: smallest and largest value) and the whole range passes the gate: coefficients and
    matrix powers per instance slot (fk<unit>_<slot>), from that instance's parameter.  A range that reaches below the bound, or one
    nobody has looked at, keeps the Filter stage."""
    d.configure(48000)
    def loop(k):
        s = d.Sum(d.Osc(110 + k / 64), 0)
        f = d.Filter(d.Delay(s, 480, 4096), 2000 + 10 * k)
        s.B = d.Multiply(f, 0.5)
        return f
    uni = descriptor.unify([descriptor.extract(loop(k)) for k in (0, 64)])
    assert "JitFilterScan" not in source(uni.words, waves=16, per_wave=1)          # (no device has looked at the column)
    monkeypatch.setenv("DUSP_CUTOFF_RANGE", "2000,6000")
    text = source(uni.words, waves=16, per_wave=2, compile=True)
    assert "JitFilterScanK fk4_0;" in text and "JitFilterScanK fk4_1;" in text
    assert re.search(r"fk4_1\.begin\(A, X\[1\], 0, p\d+_1, 2\);", text) and "f4_1.tick(X[1], fk4_1, v3_1, v4_1);" in text and "f4_0.end(A, X[0], fk4_0, 2);" in text
    assert "JitFilterK<" not in text
    monkeypatch.setenv("DUSP_CUTOFF_RANGE", "1500,6000")   # inside the loop (gain 0.5) the low end's 2 x 1.85e-6 is beyond the outlets' bound
    assert "JitFilterScan" not in source(uni.words, waves=16, per_wave=1)
    monkeypatch.setenv("DUSP_CUTOFF_RANGE", "900,6000")
    assert "JitFilterScan" not in source(uni.words, waves=16, per_wave=1)
    sweep = descriptor.unify([descriptor.extract(d.Filter(d.Osc(200 + k), 1600 + 40 * k)) for k in (0, 64)])
    monkeypatch.setenv("DUSP_CUTOFF_RANGE", "1600,4160")
    assert "JitFilterScanK fk1_0;" in source(sweep.words, waves=16, per_wave=1, compile=True)
    monkeypatch.setenv("DUSP_CUTOFF_RANGE", "1400,4160")   # below the Filter's own bound (sum|h| <= 30)
    assert "JitFilterScan" not in source(sweep.words, waves=16, per_wave=1)


def test_the_scan_gate_bounds_the_gain_of_feedback_loops(monkeypatch):
    """jit_filter_scan_ok (2): a deviation the scan injects goes round feedback loops, and a loop of gain g amplifies it by up to 1 / (1 - g).
    The gate solves x = G x + eps over the circuit with every unit's worst-case gain and takes the scan only when every outlet stays
    within 2.5e-6 of the Filters' output scale: configs[3] (gain 0.5: 2 x 1.22e-6) keeps the scan, the same voice with a feedback gain of
    0.6 and above — plucked strings, combs — renders on the Filter stage, bit for bit; so does a product with another signal whose size
    the generator does not know, and a product of two deviating signals.  DUSP_FILTER_SCAN=2 (measurements, tests) lifts the gain rule only."""
    d.configure(48000)
    def loop(k, gain, cutoff=2000, kind="LP"):
        s = d.Sum(d.Osc(110 + k / 64), 0)
        f = d.Filter(d.Delay(s, 480, 4096), cutoff, kind)
        s.B = d.Multiply(f, gain)
        return f
    def text(build, **kw):
        return source(descriptor.unify([descriptor.extract(build(k)) for k in (0, 64)]).words, waves=16, per_wave=1, **kw)
    assert "JitFilterScan" in text(lambda k: loop(k, 0.5))
    assert "JitFilterScan" in text(lambda k: loop(k, -0.5))
    for g in (0.6, 0.9, 0.95, 0.99, 1.0, 1.5):
        t = text(lambda k: loop(k, g))
        assert "JitFilterScan" not in t and "JitFilterK<" in t, g
    assert "JitFilterScan" in text(lambda k: loop(k, 0.9, 8000))            # eps = 2^-24 (2.7 + 2): ten times that is still within the bound
    assert "JitFilterScan" not in text(lambda k: loop(k, 0.99, 8000))       # ... a hundred times is not
    # two loops through one Filter add up (0.3 + 0.3: as good as one of 0.6)
    def two(k):
        s = d.Sum(d.Osc(110 + k / 64), 0)
        f = d.Filter(d.Delay(s, 480, 4096), 2000)
        s.B = d.Sum(d.Multiply(f, 0.3), d.Multiply(d.Delay(f, 700, 4096), 0.3))
        return f
    assert "JitFilterScan" not in text(two)
    # a gain that is a per-instance parameter is not in the text: refused; a known oscillator as the other factor is bounded by its table (|sin| <= 2^1)
    assert "JitFilterScan" not in text(lambda k: loop(k, 0.25 + k / 1024))
    assert "JitFilterScan" in text(lambda k: d.Multiply(d.Filter(d.Osc(200 + k), 3000), d.Osc(3)))
    assert "JitFilterScan" not in text(lambda k: d.Multiply(d.Filter(d.Osc(200 + k), 3000), d.Filter(d.Osc(3), 3000)))  # two deviating signals
    assert "JitFilterScan" not in text(lambda k: d.Multiply(d.Filter(d.Osc(200 + k), 3000), 40))  # 40 x 8e-7 of the Filter's scale
    monkeypatch.setenv("DUSP_FILTER_SCAN", "2")
    assert "JitFilterScan" in text(lambda k: loop(k, 0.99))
    assert "JitFilterScan" not in text(lambda k: loop(k, 0.5, 600))         # (the Filter's own bound stays)


def test_delays_as_lines_of_input_samples_in_lds(monkeypatch):
    """A Delay with a constant delay of a chunk at least keeps the last chunks of its INPUT in LDS rows of its wavefront (JitDelayLine) where they
    fit at 16 wavefronts next to the table image — no ring in memory (DUSP_DELAY_LINE=0: the write-once ring; =2: lines for whole delays only)."""
    d.configure(48000)
    voice = lambda k, delay: d.Delay(d.Multiply(d.Osc(300 + k), 0.5), delay, 8192)
    words = lambda delay: descriptor.unify([descriptor.extract(voice(k, delay)) for k in (0, 64)]).words
    monkeypatch.setenv("DUSP_DELAY_LINE", "0")
    assert "JitDelayLine" not in source(words(480.5), waves=16, per_wave=1) and "JitDelayK" in source(words(480.5), waves=16, per_wave=1)
    monkeypatch.setenv("DUSP_DELAY_LINE", "2")
    assert "JitDelayLine" not in source(words(480.5), waves=16, per_wave=1) and "JitDelayLine" in source(words(480), waves=16, per_wave=1)
    monkeypatch.delenv("DUSP_DELAY_LINE")
    text = source(words(480.5), waves=16, per_wave=1, compile=True)
    assert "JitDelayLine<false> y" in text and ", 3, " in text.split("JitDelayLine<false> y")[1].split("\n")[1]   # 481 samples back: two chunks and the current one
    assert "JitDelayLine" not in source(words(480.5), waves=16, per_wave=2)                                   # the rows are per wavefront
    assert "JitDelayLine" not in source(words(2000), waves=16, per_wave=1)                                    # 9 chunks x 16 wavefronts do not fit next to the image
    assert "JitDelayLine" in source(words(2000), waves=16, per_wave=1, lds_table=False)
    assert "JitDelayLine" not in source(words(480.5), waves=16, per_wave=1, continued=True)


def test_constant_delays_need_no_slot_operations():
    d.configure(48000)
    short = source(descriptor.extract(d.Delay(d.Osc(500), 30.5, 2048)).words)
    assert "JitDelayShort" in short and "JitRingOps" not in short and "JitDelayK" not in short   # two input samples per output sample: no ring
    long_ = source(descriptor.extract(d.Delay(d.Osc(500), 300.5, 2048)).words)
    assert "JitDelayLine" in long_ and "JitRingOps" not in long_ and "JitDelayK" not in long_     # its input as a line in LDS (a write-once ring in memory where that does not fit)
    moving = source(descriptor.extract(d.Delay(d.Osc(500), d.Sum(d.Multiply(d.Osc(2), 40), 200), 1024)).words)
    assert "JitDelayGather" in moving                                                               # a moving tap: slots replayed lane-parallel (slot rounds where taps decrease)
    mono = source(descriptor.extract(d.MonoDelay(d.Osc(500), d.Sum(d.Multiply(d.Osc(2), 40), 200))).words)
    assert "JitDelayGather<true>" in mono                                                           # MonoDelay likewise (its ceil tap wraps)
    readback = source(descriptor.extract(d.ReadBackDelay(d.Osc(500), d.Sum(d.Multiply(d.Osc(2), 40), 200))).words)
    assert "JitRingOps" in readback and "JitDelayGather" not in readback                            # ReadBackDelay: ordered slot operations


def test_instances_of_a_wave_share_what_does_not_depend_on_the_instance():
    """per_wave instances per wavefront: every unit block stands per_wave times — except units that compute the same chunk for
    every instance (here the envelope: constants and time only), which are emitted once and read by all."""
    d.configure(48000)
    uni = descriptor.unify([descriptor.extract(d.Filter(d.Multiply(d.Osc(110 + k), d.Ramp(48000, 1, 0).trigger()), 800)) for k in (0, 8)])
    text = source(uni.words, waves=16, per_wave=3, compile=True)
    fast = text.split("} else {")[0]
    assert fast.count("jit_ramp<") == 1 and fast.count(".tick<1, 0, 2>") == 3   # one envelope, three oscillators
    # 48 recurrences side by side on wave 0, in two sub-blocks of 128 samples (what LDS holds next to the table image): the feed-forward
    # half once per instance and chunk, into registers; each sub-block parked from there, followed by its give-back branch (parked again
    # and done as written if a recurrence met a NaN)
    assert fast.count("f3.feed(") == 3 and fast.count("f3.park(") == 12 and fast.count("f3.serial<8>(") == 2 and fast.count("f3.pick(") == 6
    assert fast.count("if (f3.failed(tile))") == 2 and fast.count("f3.serial_exact(") == 2
    assert "JitFilterK<16, 3, 128, 1, " in text and "dusp_jit_pass" not in text
    with pytest.raises(runtime.DuspHipError, match="per_wave"):
        source(uni.words, waves=16, per_wave=5)


def test_bad_arguments_come_back_as_statuses():
    g = Golden("osc440_1s")
    with pytest.raises(runtime.DuspHipError, match="waves must be"):
        source(g.desc, waves=17)
    with pytest.raises(runtime.DuspHipError, match="magic"):
        source(np.zeros(40))


def test_saw_square_triangle_are_evaluated_not_looked_up():
    """The reference's saw, square and triangle tables (waveTables.js:10-26) are functions of the index: kernels evaluate them
    (table source 2) instead of gathering from a 192 KB table that does not fit LDS; sine and 8bit come from the LDS half image."""
    d.configure(48000)
    mix = d.Sum(d.Sum(d.Osc(110, "saw"), d.Osc(220, "square")), d.Sum(d.Osc(330, "triangle"), d.Sum(d.Osc(440), d.Osc(550, "8bit"))))
    text = source(descriptor.extract(mix).words, waves=4, compile=True)
    fast = text.split("} else {")[0]
    for args in ("tick<2, 1, 1>", "tick<2, 2, 1>", "tick<2, 3, 1>", "tick<1, 0, 2>", "tick<3, 4, 1>"):
        assert fast.count(args) == 1, args
    # without a context's verdict on the tables nothing is assumed: everything is gathered
    plain = source(descriptor.extract(mix).words, waves=4, lds_table=False)
    assert "tick<0, 0, 1>" in plain and "tick<2," not in plain


def _compile_in_a_fresh_process(cache, name="fm_sum", extra_env=None):
    """One process: compile the kernel of a golden circuit, report seconds and the cache directory the library uses."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = ("import sys, time, json; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests');\n"
            "from conftest import Golden; from dusp_amd import runtime; runtime.load()\n"
            "t0 = time.perf_counter(); runtime.circuit_kernel_source(Golden(%r).desc, waves=4, compile=True)\n"
            "print(json.dumps({'s': time.perf_counter() - t0, 'dir': runtime.load().dusp_jit_cache_dir().decode()}))" % (ROOT, ROOT, name))
    env = dict(os.environ)
    env.pop("XDG_CACHE_HOME", None)
    if cache is not None:
        env["DUSP_JIT_CACHE"] = cache
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_code_objects_are_cached_on_disk_and_a_damaged_file_is_compiled_again(tmp_path):
    """The code-object cache (jit_engine.hip): a second PROCESS finds the kernel on disk instead of compiling; a truncated or
    altered cache file is noticed (length + hash in its header), deleted and replaced by a fresh compile; "0" turns the cache off."""
    import os
    cache = str(tmp_path / "cache")
    first = _compile_in_a_fresh_process(cache)
    assert first["dir"] == cache
    files = [f for f in os.listdir(cache) if f.endswith(".hsaco")]
    assert len(files) == 1
    path = os.path.join(cache, files[0])
    good = open(path, "rb").read()
    assert good[:8] == b"DUSPHSA2" and len(good) > 4096
    again = _compile_in_a_fresh_process(cache)
    assert again["s"] < 0.5 * first["s"], (first, again)  # (no hiprtc run: generate the text, read the file)
    assert open(path, "rb").read() == good
    # truncated
    open(path, "wb").write(good[: len(good) // 2])
    _compile_in_a_fresh_process(cache)
    assert open(path, "rb").read() == good
    # same length, one byte of the payload changed
    bad = bytearray(good)
    bad[len(bad) // 2] ^= 0x40
    open(path, "wb").write(bytes(bad))
    _compile_in_a_fresh_process(cache)
    assert open(path, "rb").read() == good
    # the header's second hash of the kernel TEXT: a file that holds another text's kernel under this one's name is not run
    bad = bytearray(good)
    bad[24] ^= 0x01
    open(path, "wb").write(bytes(bad))
    _compile_in_a_fresh_process(cache)
    assert open(path, "rb").read() == good
    # a directory others may write to is not trusted with code objects: no disk cache then
    os.chmod(cache, 0o777)
    assert _compile_in_a_fresh_process(cache)["dir"] == ""
    os.chmod(cache, 0o700)
    # a cap on the directory's size (DUSP_JIT_CACHE_MAX_MB): the least recently used files go when a store takes it over
    for k in range(5):
        open(os.path.join(cache, "dusp_%016x_1.hsaco" % k), "wb").write(b"x" * (300 << 10))
        os.utime(os.path.join(cache, "dusp_%016x_1.hsaco" % k), (1000 + k, 1000 + k))
    os.remove(path)
    _compile_in_a_fresh_process(cache, extra_env={"DUSP_JIT_CACHE_MAX_MB": "1"})
    left = sorted(os.listdir(cache))
    assert files[0] in left and "dusp_%016x_1.hsaco" % 0 not in left and "dusp_%016x_1.hsaco" % 1 not in left and len(left) < 6, left
    # off
    off = _compile_in_a_fresh_process("0")
    assert off["dir"] == ""


def test_voices_of_a_sum_run_in_a_loop():
    """A Sum.many of N isomorphic voices above 96 units: the voice's units ONCE, inside a loop over the voices (jit_codegen.hpp VoicePlan /
    run_voices) — the text does not grow with N, names no per-voice constant, and carries one accumulate pass per FM level for time-split
    renders; a sum of voices of two kinds, or a small one, is straight-line code; above 256 voices or 256 units of anything else: refused."""
    d.configure(48000)
    fm = lambda j: d.Osc(d.Sum(d.Multiply(d.Osc(3.0 + j / 7), 40), 220 + 11.5 * j))
    env = lambda j: d.Multiply(fm(j), d.Multiply(d.Shape("decay", 0.3 + j / 50).trigger(), d.Ramp(24000 + 100 * j, 1, 0).trigger()))
    text = {n: source(descriptor.extract(d.Sum.many([env(j) for j in range(n)])).words) for n in (30, 120)}
    assert "in a loop" in text[30].splitlines()[0] and abs(len(text[30]) - len(text[120])) < 64
    assert "for (int j = 0; j < 120; ++j)" in text[120] and "JitOscS o3[120];" in text[120] and "JitShape e" in text[120] and "jit_ramp<" in text[120]
    assert "dusp_jit_pass1" in text[120] and "A.seg_sum[" in text[120]                      # FM carriers: start phases of time segments
    assert not re.search(r"\bk\d+\b = jit_u\(A\.fk", text[120])                              # constants come out of the voice's table row
    source(descriptor.extract(d.Sum.many([env(j) for j in range(40)])).words, compile=True)  # (and it compiles)
    mastered = source(descriptor.extract(d.HardClipAbove(d.Multiply(d.Sum.many([env(j) for j in range(30)]), 0.25), 0.9)).words)  # a master gain and a clip on the mix
    assert "in a loop" in mastered.splitlines()[0] and "u0_0[c] = acc0[c] * jit_u(A.fk[" in mastered and "u0_1[c] = map_apply(" in mastered and "jit_store<false>(A, X[0], g, 0, u0_1);" in mastered
    stereo = source(descriptor.extract(d.Sum.many([d.Pan(env(j), -1 + j / 15) for j in range(30)])).words)  # panned voices: a chain per output channel, the voice's units once
    assert "in a loop" in stereo.splitlines()[0] and stereo.count("map_pan(") >= 4 and "map_pan_compensation(" in stereo and "jit_store<false>(A, X[0], g, 1, acc1);" in stereo
    small = source(descriptor.extract(d.Sum.many([fm(j) for j in range(12)])).words)         # 59 units: as before
    assert "in a loop" not in small.splitlines()[0] and "o3_0.tick<" in small
    mixed = d.Sum.many([fm(j) if j % 2 else d.Multiply(d.Osc(50.5 + j), 0.25) for j in range(40)])  # 20 x 4 + 20 x 2 + 39 = 159 units, two kinds of voice
    assert "in a loop" not in source(descriptor.extract(mixed).words).splitlines()[0]
    for too_big in (d.Sum.many([fm(j) for j in range(257)]),
                    d.Sum.many([fm(j) if j % 2 else d.Multiply(d.Osc(50.5 + j), 0.25) for j in range(80)])):
        with pytest.raises(runtime.DuspHipError, match="DUSP_JIT_MAX_UNITS"):
            source(descriptor.extract(too_big).words)
