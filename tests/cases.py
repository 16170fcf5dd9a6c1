"""The golden cases of oracle/js/gen_golden.js, rebuilt with this package's own
graph classes (dusp_amd/graph.py).  Used to check that the host mirror produces
the same descriptor — constants, state AND unit order — as the reference objects
did, and as ready-made graphs for the GPU parity tests."""
import dusp_amd as d
from dusp_amd import (Abs, AllPass, CircleBuffer, CircleBufferReader, CircleBufferWriter, Clip, CombFilter, DecibelToScaler, Delay, Divide, Filter, FixedDelay,
                      FixedMultiply, Gain, HardClipAbove, HardClipBelow, MonoDelay, MultiChannelOsc, Multiply, Osc, Pow, Ramp, ReadBackDelay, Repeater,
                      SecondsToSamples, SemitoneToRatio, Subtract, Sum, quick,
                      AHD, ConcatChannels, CrossFader, MidiToFrequency, Pan, PickChannel, Rescale, SampleRateRedux, Shape, Timer, VectorMagnitude)


def _loop(f_osc, delay, max_delay, cutoff, gain):
    s = Sum(Osc(f_osc), 0)
    dl = Delay(s, delay, max_delay)
    f = Filter(dl, cutoff)
    fb = Multiply(f, gain)
    s.B = fb
    return f


def _taps():
    buffer = CircleBuffer(1, 0.05)
    writer = CircleBufferWriter(buffer)
    writer.preWipe = True
    writer.IN = Osc(330)
    tap = CircleBufferReader(buffer, 0.01)
    tap.chain(writer)
    fb_tap = CircleBufferReader(buffer, 0.02)
    fb_tap.chain(writer)
    fb_writer = CircleBufferWriter(buffer, 0.005)
    fb_writer.IN = quick.multiply(fb_tap, 0.5)
    fb_writer.chain(writer)
    return Sum(tap, fb_tap)


def _cb2():
    buffer = CircleBuffer(2, 0.011)
    writer = CircleBufferWriter(buffer, 0.001)
    writer.IN = Multiply(Osc(700), [1, 0.5])
    reader = CircleBufferReader(buffer, [0.002, 0.0035])
    reader.postWipe = True
    reader.chain(writer)
    return reader


def _cb_moving_tap():
    buffer = CircleBuffer(1, 0.05)
    writer = CircleBufferWriter(buffer)
    writer.preWipe = True
    writer.IN = Osc(330, "saw")
    tap = CircleBufferReader(buffer, Sum(Multiply(Osc(3), 0.004), 0.01))
    tap.chain(writer)
    fast = CircleBufferReader(buffer, Sum(Multiply(Osc(700), 0.002), 0.02))
    fast.chain(writer)
    return Sum(tap, fast)


def _cb_moving_writer():
    buffer = CircleBuffer(1, 0.02)
    writer = CircleBufferWriter(buffer, Sum(Multiply(Osc(900), 0.003), 0.004))
    writer.IN = Osc(440)
    reader = CircleBufferReader(buffer, 0.001)
    reader.postWipe = True
    reader.chain(writer)
    return reader


def _cb_short_ring():
    buffer = CircleBuffer(1, 0.003)
    writer = CircleBufferWriter(buffer)
    writer.IN = Multiply(Osc(500), Ramp(3000, 1, 0).trigger())
    reader = CircleBufferReader(buffer, 0.002)
    reader.chain(writer)
    return reader


def _mult_inlet_zero():
    m = Multiply(Osc(440), 2)
    m.B = 0
    return m


def _clip():
    c = Clip(Multiply(Osc(2), 0.8))
    c.IN = Osc(333)
    return c


def _seconds():
    s = SecondsToSamples()
    s.IN = Multiply(Osc(10), 0.001)
    return s


def _gain():
    g = Gain(Multiply(Osc(3), 12))
    g.IN = Osc(220)
    return g


def _with_in(unit, source):
    unit.IN = source
    return unit


def _allpass_series():
    a = _with_in(AllPass(0.0021, 0.6), Osc(220, "square"))
    return _with_in(AllPass(0.0013, -0.45), a)


def _allpass_loop():
    s = Sum(Osc(180), 0)
    a = _with_in(AllPass(0.0052, 0.5), s)
    s.B = Multiply(a, 0.4)
    return a


def _shape_edges():
    s = Shape("attack", 0.004, 0.5, 2)
    s.leftEdge = "shape"
    s.rightEdge = 0.5
    return s.trigger()


def _shape_left_number():
    s = Shape("decay", 0.01, -1, 1)
    s.leftEdge = 0.25
    return s


def _srr_nan():
    r = SampleRateRedux(Osc(300), 5)
    r.AMMOUNT = float("nan")
    return r


def _grow_stereo():
    s = Sum(Osc(440), 0)
    s.B = Multiply(s, [0.5, -0.25])
    return s


def _grow_filter():
    s = Sum(Osc(300, "saw"), 0)
    f = Filter(s, 1500)
    s.B = Multiply(f, [0.4, 0.2, -0.3])
    return f


def _grow_multiosc():
    f = Sum(200, 0)
    osc = MultiChannelOsc(f)
    f.B = Multiply(osc, [30, 50])
    return osc


def _grow_two_loops():
    a, b = Sum(Osc(220), 0), Sum(Osc(331, "triangle"), 0)
    a.B = Multiply(b, 0.5)
    b.B = Multiply(a, [0.25, -0.5])
    return Sum(a, b)


def builders(sr):
    """name -> zero-argument builder; call d.configure(sr) first (done by build())."""
    voices = lambda n: [Osc(k * 10) for k in range(1, n + 1)]
    b = {
        "osc440_1s": lambda: Osc(440),
        "grow_feedback_stereo": _grow_stereo, "grow_feedback_filter": _grow_filter,
        "grow_feedback_multiosc": _grow_multiosc, "grow_two_loops": _grow_two_loops,
        "cfg2_literal": lambda: Multiply(Osc(Ramp(200, 100, 2)), Osc(3)),
        "cfg2_sweep": lambda: Multiply(Osc(Ramp(2 * sr, 200, 100).trigger()), Osc(3)),
        "ramp_300": lambda: Ramp(300, 0.25, 2).trigger(),
        "ramp_default_idle": lambda: Ramp(),
        "ramp_1": lambda: Ramp(1, 5, -5).trigger(),
        "ramp_frac": lambda: Ramp(1000.5, -1, 1).trigger(),
        "summany_8": lambda: Sum.many(voices(8)),
        "summany_1024": lambda: Sum.many(voices(1024)),
        "loop_220": lambda: _loop(220, 480, 4096, 2000, 0.5),
        "loop_110p5_short": lambda: _loop(110.5, 100, 4096, 2000, 0.5),
        "loop_frac_delay": lambda: _loop(330, 300.25, 2048, 1500, 0.7),
        "delay_default": lambda: Delay(Osc(100), 0, 0),
        "delay_wrap": lambda: Delay(Osc(1000), 700.5, 1000),
        "delay_mod": lambda: Delay(Osc(500), Sum(Multiply(Osc(2), 40), 200), 1024),
        "delay_2ch": lambda: Delay(Multiply(Osc(300), [1, -0.5]), [64, 333.75], 2048),
        "filter_lp_mod": lambda: Filter(Osc(150, "saw"), Sum(Multiply(Osc(5), 800), 1000)),
        "filter_hp": lambda: Filter(Osc(150, "square"), 3000, "HP"),
        "filter_2ch": lambda: Filter(Multiply(Osc(440), [0.5, 0.25]), 1200),
        "fm_mixed": lambda: Osc(Multiply(Osc(70), 9000)),
        "fm_sum": lambda: Osc(Sum(Multiply(Osc(3.5, "triangle"), 300), 220.25)),
        "circlebuffer_taps": _taps,
        "circlebuffer_2ch": _cb2,
        "circlebuffer_moving_tap": _cb_moving_tap, "circlebuffer_moving_writer": _cb_moving_writer, "circlebuffer_short_ring": _cb_short_ring,
        "osc440_480": lambda: Osc(440),
        "mult_2ch": lambda: Multiply(Osc(440), [0.5, 0.25]),
        "repeater": lambda: Repeater(quick.mult(Osc(123.4), 1)),
        "sum_const": lambda: quick.add(Osc(50), 0.75),
        "mult_inlet_zero": _mult_inlet_zero,
        # elementwise maps
        "map_subtract": lambda: Subtract(Osc(300), Multiply(Osc(7), [0.5, 0.25])),
        "map_subtract_quick": lambda: quick.subtract(Osc(300, "saw"), 0.3),
        "map_divide": lambda: Divide(Osc(200), Sum(Osc(3), 1.5)),
        "map_divide_by_zero": lambda: Divide(Osc(100), Osc(50, "square")),
        "map_invert_abs": lambda: Abs(quick.invert(Osc(440.5))),
        "map_clip": _clip,
        "map_hardclip": lambda: HardClipBelow(HardClipAbove(Osc(150), 0.5), [-0.25, -0.75]),
        "map_seconds": _seconds,
        "map_fixedmultiply": lambda: FixedMultiply(0.1, Osc(441)),
        "map_gain": _gain,
        "map_db_semitone": lambda: Multiply(DecibelToScaler(Multiply(Osc(5), 20)), SemitoneToRatio(Multiply(Osc(2), 7))),
        "map_pow": lambda: Pow(Sum(Osc(100), 1.5), Multiply(Osc(1.5), 2)),
        "map_pow_negative_base": lambda: quick.pow(Osc(100), 0.5),
        "map_fm_semitone": lambda: Osc(Multiply(SemitoneToRatio(Multiply(Osc(4), 12)), 220)),
        # delay / filter family, per-channel oscillator
        "fam_fixeddelay": lambda: _with_in(FixedDelay(0.0031), Osc(441)),
        "fam_comb": lambda: _with_in(CombFilter(0.004, 0.7), Osc(150, "saw")),
        "fam_comb_mod": lambda: _with_in(CombFilter(0.0007, Multiply(Osc(3), 0.9)), Osc(333.3)),
        "fam_allpass_series": _allpass_series,
        "fam_allpass_loop": _allpass_loop,
        "fam_monodelay": lambda: MonoDelay(Osc(500), 123.5),
        "fam_monodelay_mod": lambda: MonoDelay(Osc(300), Sum(Multiply(Osc(1.5), 30), 100)),
        "fam_readback": lambda: ReadBackDelay(Multiply(Osc(400), [1, -1]), [100, 2000], 4096),
        "fam_readback_frac": lambda: ReadBackDelay(Osc(400), 10.5, 1024),
        "fam_multiosc": lambda: MultiChannelOsc([220, 330.5, 441.25]),
        "fam_multiosc_fm": lambda: MultiChannelOsc(Sum(Multiply(Osc(5), [20, 40]), 300), "triangle"),
        "fam_multiosc_negative": lambda: MultiChannelOsc(-100),
        # SURVEY.md 8f-1, the rest of the sweep
        "rest_pan": lambda: Pan(Osc(440), Osc(2)),
        "rest_pan_const": lambda: Pan(Osc(300.5, "saw"), -0.3),
        "rest_pan_filter": lambda: Filter(Pan(Osc(200, "square"), 0.25), 1500),
        "rest_midi_fm": lambda: Osc(MidiToFrequency(Sum(Multiply(Osc(3), 12), 60))),
        "rest_midi_multi": lambda: MidiToFrequency([60, 72]),
        "rest_rescale_fm": lambda: Osc(_with_in(Rescale(-1, 1, 200, 800), Osc(5))),
        "rest_rescale_2ch": lambda: _with_in(Rescale(Multiply(Osc(1), 0.5), 2, [0, 10], [1, 20]), Multiply(Osc(100), [1, 0.5])),
        "rest_rescale_default": lambda: _with_in(Rescale(), Osc(441, "triangle")),
        "rest_crossfader": lambda: CrossFader(Osc(220), Multiply(Osc(330, "square"), [1, 0.5]), Sum(Multiply(Osc(4), 0.5), 0.5)),
        "rest_crossfader_const": lambda: CrossFader(Osc(100), 0.5, 0.25),
        "rest_vecmag": lambda: _with_in(VectorMagnitude(), Multiply(Osc(50), [1, 0.5, -2])),
        "rest_vecmag_2d": lambda: _with_in(VectorMagnitude(), ConcatChannels(Osc(100), Osc(100.5, "triangle"))),
        "rest_vecmag_default": lambda: VectorMagnitude(),
        "rest_timer": lambda: Timer(),
        "rest_timer_fm": lambda: Osc(Multiply(Timer(), 4000)),
        "rest_srr": lambda: SampleRateRedux(Osc(440), 10),
        "rest_srr_mod": lambda: SampleRateRedux(Multiply(Osc(300), [1, -1]), Sum(Multiply(Osc(2), 20), 20.5)),
        "rest_srr_nan": _srr_nan,
        "rest_srr_nan_later": lambda: SampleRateRedux(Osc(300), Divide(Osc(100), Osc(100))),
        "rest_concat": lambda: ConcatChannels(Multiply(Osc(100), [1, 0.5]), Osc(200)),
        "rest_concat_quick": lambda: quick.concat(Osc(50), 0.25),
        "rest_pick": lambda: PickChannel(Multiply(Osc(100), [1, 0.5, 0.25]), 4),
        "rest_pick_default": lambda: PickChannel(Multiply(Osc(60), [0.75, 0.5])),
        # envelopes
        "env_shape_decay": lambda: Shape("decay", 0.02).trigger(),
        "env_shape_idle": lambda: Shape("attack", 0.01, 0.25, 0.75),
        "env_shape_semisine_amp": lambda: Multiply(Osc(440), Shape("semiSine", 0.03).trigger()),
        "env_shape_decaysq_range": lambda: Shape("decaySquared", 0.013, -1, 2).trigger(),
        "env_shape_edges": _shape_edges,
        "env_shape_left_number": _shape_left_number,
        "env_shape_mod": lambda: Shape("decay", Sum(Multiply(Osc(20), 0.01), 0.02), Multiply(Osc(3), 0.5), 1).trigger(),
        "env_ahd": lambda: AHD(0.01, 0.02, 0.03).trigger(),
        "env_ahd_idle": lambda: AHD(0.01, 0.02, 0.03),
        "env_ahd_zero_hold": lambda: AHD(0.005, 0, 0.005).trigger(),
        "env_ahd_amp": lambda: Multiply(Osc(330), AHD(0.002, 0.01, 0.02).trigger()),
        "env_ahd_mod": lambda: AHD(Sum(Multiply(Osc(50), 0.002), 0.004), 0.002, Sum(Multiply(Osc(30), 0.01), 0.02)).trigger(),
    }
    for tag, f in [("440p5", 440.5), ("0p1", 0.1), ("neg3", -3), ("47999p5", 47999.5), ("neg0p37", -0.37),
                   ("12345p678", 12345.678), ("tiny", 3e-5)]:
        b["osc_f_" + tag] = (lambda f=f: Osc(f))
    for w in ["saw", "square", "triangle", "8bit"]:
        b["osc_" + w] = (lambda w=w: Osc(441.3, w))
    for k in [1, 7, 512, 1024]:
        b["voice3_k%d" % k] = (lambda k=k: Multiply(Osc(10 * k), Ramp(sr, 1, 0).trigger()))
    for k in [0, 3, 4097, 65535]:
        b["voice5_k%d" % k] = (lambda k=k: Multiply(Osc(20 + k / 8), Ramp(sr, 1, 0).trigger()))
    return b


def event_builders(sr):
    """The `ev_` cases of tests/js/cases.js: graphs with scheduled host callbacks (callbacks take the unit)."""
    def retrigger():
        r = Ramp(2400, 1, 0)
        r.scheduleTrigger(0.01)
        r.scheduleTrigger(0.08)
        return Multiply(Osc(440), r)

    def freq_steps():
        osc = Osc(220)
        osc.schedule([0.02, 0.05, 0.09], lambda u: setattr(u, "F", u.F.constant * 1.5))
        return osc

    def repeating():
        r = Ramp(600, 1, 0.25)

        def again(u):
            u.trigger()
            return 0.03  # a positive return value reschedules (Event.js:20-27)
        r.schedule(0.004, again)
        return Multiply(Osc(330.5, "triangle"), r)

    def filter_sweep():
        filt = Filter(Osc(100, "saw"), 500)
        filt.schedule(0.03, lambda u: setattr(u, "F", 2500))
        filt.schedule(0.06, lambda u: setattr(u, "F", 800))
        return filt

    def delay_retrigger():
        r = Ramp(1200, 1, 0)
        r.scheduleTrigger([0.004, 0.03, 0.055])
        return Delay(Multiply(Osc(660), r), 300.5, 2048)

    def delay_time_change():
        dl = Delay(Osc(500), 100, 4096)
        dl.schedule(0.02, lambda u: setattr(u, "DELAY", 1000.25))
        dl.schedule(0.05, lambda u: setattr(u, "DELAY", 17))
        return dl

    def delay_long_to_short():
        dl = Delay(Osc(500), 1000.25, 4096)
        dl.schedule(0.02, lambda u: setattr(u, "DELAY", 300))
        dl.schedule(0.05, lambda u: setattr(u, "DELAY", 17.5))
        return dl

    def loop_delay_change():
        s = Sum(Osc(220), 0)
        dl = Delay(s, 480, 4096)
        f = Filter(dl, 2000)
        s.B = Multiply(f, 0.6)
        dl.schedule(0.04, lambda u: setattr(u, "DELAY", 700.5))
        return f

    def loop_gain():
        s = Sum(Osc(220), 0)
        f = Filter(Delay(s, 480, 4096), 2000)
        fb = Multiply(f, 0.5)
        s.B = fb
        fb.schedule(0.03, lambda u: setattr(u, "B", 0.9))
        fb.schedule(0.07, lambda u: setattr(u, "B", 0.1))
        return f

    def feedback_no_delay():
        s = Sum(Osc(330), 0)
        m = Multiply(s, 0.5)
        s.B = m
        m.schedule([0.011, 0.033], lambda u: setattr(u, "B", u.B.constant * -1.5))
        return s

    def circlebuffer():
        buffer = CircleBuffer(1, 0.05)
        writer = CircleBufferWriter(buffer)
        writer.preWipe = True
        osc = Osc(330)
        writer.IN = osc
        tap = CircleBufferReader(buffer, 0.01)
        tap.chain(writer)
        osc.schedule(0.02, lambda u: setattr(u, "F", 495))
        tap.schedule(0.04, lambda u: setattr(u, "OFFSET", 0.003))
        return tap

    def comb():
        c = CombFilter(0.004, 0.7)
        r = Ramp(800, 1, 0)
        c.IN = Multiply(Osc(150, "saw"), r)
        r.scheduleTrigger([0.0, 0.05])
        c.schedule(0.03, lambda u: setattr(u, "FEEDBACKGAIN", -0.5))
        return c

    def shape_retrigger():
        sh = Shape("decaySquared", 0.02)
        sh.scheduleTrigger([0.01, 0.06])
        return Multiply(Osc(440), sh)

    def ahd_retrigger():
        e = AHD(0.004, 0.003, 0.006)
        e.scheduleTrigger([0.005, 0.04])
        return e

    def timer_trigger():
        tm = Timer()
        tm.scheduleTrigger(0.02)
        return Osc(Multiply(tm, 20000))

    return {"ev_retrigger": retrigger, "ev_freq_steps": freq_steps, "ev_repeating": repeating, "ev_filter_sweep": filter_sweep,
            "ev_delay_retrigger": delay_retrigger, "ev_delay_long_to_short": delay_long_to_short, "ev_loop_delay_change": loop_delay_change, "ev_delay_time_change": delay_time_change, "ev_loop_gain": loop_gain,
            "ev_feedback_no_delay": feedback_no_delay, "ev_circlebuffer": circlebuffer, "ev_comb": comb,
            "ev_shape_retrigger": shape_retrigger, "ev_ahd_retrigger": ahd_retrigger, "ev_timer_trigger": timer_trigger}


def build_event_case(name, sr):
    d.configure(sr)
    return event_builders(sr)[name]()


def build(name, sr):
    """Build the graph of golden case `name` (48 kHz names carry no suffix)."""
    d.configure(sr)
    base = name[: -len("_sr%d" % sr)] if name.endswith("_sr%d" % sr) else name
    return builders(sr)[base]()
