'use strict'
/* The golden cases, written ONCE against an abstract unit library `L` so that the very same
 * construction code runs
 *   - on the reference's constructors (oracle/js/gen_golden.js, build container only) to produce
 *     tests/golden/*, and
 *   - on this package's own graph classes (tests/js/check_*.js) to verify that the host mirror
 *     extracts to the same descriptor and that the GPU renders the same PCM.
 * L = { Osc, Ramp, Multiply, Sum, Filter, Delay, Repeater, CircleBuffer, CircleBufferReader,
 *       CircleBufferWriter, quick };  SR = sample rate the library was configured with. */
module.exports = function goldenCases(L, SR) {
  const { Osc, Ramp, Multiply, Sum, Filter, Delay, Repeater, CircleBuffer, CircleBufferReader, CircleBufferWriter, quick,
    Subtract, Divide, PolarityInvert, Abs, Clip, HardClipAbove, HardClipBelow, SecondsToSamples, FixedMultiply, Gain,
    DecibelToScaler, SemitoneToRatio, Pow, FixedDelay, CombFilter, AllPass, MonoDelay, ReadBackDelay, MultiChannelOsc,
    Pan, MidiToFrequency, Rescale, CrossFader, VectorMagnitude, Timer, SampleRateRedux, ConcatChannels, PickChannel,
    Shape, AHD, Retriggerer } = L // + L.patches (optional): the patch builders
  const S = (name) => (SR === 48000 ? name : name + '_sr' + SR)
  const cases = []
  // opts.seed: build AND render under a seeded Math.random (units that draw random numbers while they tick)
  const add = (name, build, duration, windows, opts) => cases.push({ name: S(name), build, duration, windows, ...(opts || {}) })

  // G1: configs[0] — single [Osc 440], 1 s
  add('osc440_1s', () => new Osc(440), 1)

  // G2: wrap / negative / fractional / tiny increments
  for (const [tag, f] of [['440p5', 440.5], ['0p1', 0.1], ['neg3', -3], ['47999p5', 47999.5],
    ['neg0p37', -0.37], ['12345p678', 12345.678], ['tiny', 3e-5]])
    add('osc_f_' + tag, () => new Osc(f), 0.1)

  // G3: every wave table, fractional phase so the lerp is exercised
  for (const w of ['saw', 'square', 'triangle', '8bit'])
    add('osc_' + w, () => new Osc(441.3, w), 0.05)

  // G4: configs[1] — [Multiply A:[Osc f:[Ramp 200 100 2]] B:[Osc 3]]; literal reading (idle ramp = 100 Hz)
  add('cfg2_literal', () => new Multiply(new Osc(new Ramp(200, 100, 2)), new Osc(3)), 10,
    [[0, 4096], [5 * SR - 128, 512], [10 * SR - 256, 256]])
  // ... and the intended 200->100 Hz sweep over 2 s
  add('cfg2_sweep', () => new Multiply(new Osc(new Ramp(2 * SR, 200, 100).trigger()), new Osc(3)), 10,
    [[0, 4096], [2 * SR - 2048, 4096], [10 * SR - 256, 256]])

  // ramp edge cases: duration not a multiple of the chunk, rising ramp, 1-sample ramp, idle default ramp
  add('ramp_300', () => new Ramp(300, 0.25, 2).trigger(), 0.02)
  add('ramp_default_idle', () => new Ramp(), 0.01)
  add('ramp_1', () => new Ramp(1, 5, -5).trigger(), 0.01)
  add('ramp_frac', () => new Ramp(1000.5, -1, 1).trigger(), 0.03)

  // G5: Sum.many left-deep chain (configs[2] mix-down mode), 8 and 1024 voices
  const voices = (n) => { const v = []; for (let k = 1; k <= n; k++) v.push(new Osc(k * 10)); return v }
  add('summany_8', () => Sum.many(voices(8)), 2048 / SR)
  add('summany_1024', () => Sum.many(voices(1024)), 2048 / SR)

  // configs[2]/[4] per-voice mode: Multiply(Osc(f), Ramp(T,1,0) triggered); a few voices of each sweep
  for (const k of [1, 7, 512, 1024])
    add('voice3_k' + k, () => new Multiply(new Osc(10 * k), new Ramp(SR, 1, 0).trigger()), 1,
      [[0, 2048], [SR - 1024, 1024]])
  for (const k of [0, 3, 4097, 65535])
    add('voice5_k' + k, () => new Multiply(new Osc(20 + k / 8), new Ramp(SR, 1, 0).trigger()), 1,
      [[0, 2048], [SR - 1024, 1024]])

  // G6: configs[3] — feedback loop Osc -> Sum -> Delay -> Filter -> Multiply -> (back into Sum)
  const loop = (fOsc, delay, maxDelay, cutoff, gain) => {
    const sum = new Sum(new Osc(fOsc), 0)
    const d = new Delay(sum, delay, maxDelay)
    const f = new Filter(d, cutoff)
    const fb = new Multiply(f, gain)
    sum.B = fb
    return f
  }
  add('loop_220', () => loop(220, 480, 4096, 2000, 0.5), 0.1)
  add('loop_110p5_short', () => loop(110.5, 100, 4096, 2000, 0.5), 0.05) // delay < chunk
  add('loop_frac_delay', () => loop(330, 300.25, 2048, 1500, 0.7), 0.05)

  // delay on its own: default 5 s ring, integer / fractional / wrapping write position
  add('delay_default', () => new Delay(new Osc(100), 0, 0), 0.12) // literal 0s fall back to 4410 / 5*sr
  add('delay_wrap', () => new Delay(new Osc(1000), 700.5, 1000), 0.1)
  add('delay_mod', () => new Delay(new Osc(500), new Sum(new Multiply(new Osc(2), 40), 200), 1024), 0.1)
  add('delay_2ch', () => new Delay(new Multiply(new Osc(300), [1, -0.5]), [64, 333.75], 2048), 0.05)

  // filter: LP with modulated cutoff (per-sample coefficient refresh), HP, 2 channels, no cutoff given
  add('filter_lp_mod', () => new Filter(new Osc(150, 'saw'), new Sum(new Multiply(new Osc(5), 800), 1000)), 0.1)
  add('filter_hp', () => new Filter(new Osc(150, 'square'), 3000, 'HP'), 0.05)
  add('filter_2ch', () => new Filter(new Multiply(new Osc(440), [0.5, 0.25]), 1200), 0.03)

  // FM: connected f with mixed-sign increments (exercises the modular phase scan)
  add('fm_mixed', () => new Osc(new Multiply(new Osc(70), 9000)), 0.1)
  add('fm_sum', () => new Osc(new Sum(new Multiply(new Osc(3.5, 'triangle'), 300), 220.25)), 0.1)

  // G7: CircleBuffer writer / reader / feedback writer, MultiTapDelay topology
  //     (reference src/patches/MultiTapDelay.js:27-43), built by hand from its parts
  add('circlebuffer_taps', () => {
    const buffer = new CircleBuffer(1, 0.05)
    const writer = new CircleBufferWriter(buffer)
    writer.preWipe = true
    writer.IN = new Osc(330)
    const tap = new CircleBufferReader(buffer, 0.01)
    tap.chain(writer)
    const fbTap = new CircleBufferReader(buffer, 0.02)
    fbTap.chain(writer)
    const fbWriter = new CircleBufferWriter(buffer, 0.005)
    fbWriter.IN = quick.multiply(fbTap, 0.5)
    fbWriter.chain(writer)
    return new Sum(tap, fbTap)
  }, 0.15)
  add('circlebuffer_2ch', () => {
    const buffer = new CircleBuffer(2, 0.011)
    const writer = new CircleBufferWriter(buffer, 0.001)
    writer.IN = new Multiply(new Osc(700), [1, 0.5])
    const reader = new CircleBufferReader(buffer, [0.002, 0.0035])
    reader.postWipe = true
    reader.chain(writer)
    return reader
  }, 0.05)

  // CircleBuffer nodes whose accesses can meet inside a chunk: a moving tap (chorus), a moving writer, a ring shorter than a chunk
  add('circlebuffer_moving_tap', () => {
    const buffer = new CircleBuffer(1, 0.05)
    const writer = new CircleBufferWriter(buffer)
    writer.preWipe = true
    writer.IN = new Osc(330, 'saw')
    const tap = new CircleBufferReader(buffer, new Sum(new Multiply(new Osc(3), 0.004), 0.01)) // up to 0.58 samples per sample
    tap.chain(writer)
    const fast = new CircleBufferReader(buffer, new Sum(new Multiply(new Osc(700), 0.002), 0.02)) // the tap moves faster than time
    fast.chain(writer)
    return new Sum(tap, fast)
  }, 0.1)
  add('circlebuffer_moving_writer', () => {
    const buffer = new CircleBuffer(1, 0.02)
    const writer = new CircleBufferWriter(buffer, new Sum(new Multiply(new Osc(900), 0.003), 0.004)) // writes pile up on slots (mix)
    writer.IN = new Osc(440)
    const reader = new CircleBufferReader(buffer, 0.001)
    reader.postWipe = true
    reader.chain(writer)
    return reader
  }, 0.06)
  add('circlebuffer_short_ring', () => { // 144 slots: a chunk laps the ring
    const buffer = new CircleBuffer(1, 0.003)
    const writer = new CircleBufferWriter(buffer)
    writer.IN = new Multiply(new Osc(500), new Ramp(3000, 1, 0).trigger())
    const reader = new CircleBufferReader(buffer, 0.002)
    reader.chain(writer)
    return reader
  }, 0.05)

  // G8: length not a multiple of the chunk; 2-channel output; Repeater; quick.mult(x, 1) elision
  add('osc440_480', () => new Osc(440), 0.01)
  add('mult_2ch', () => new Multiply(new Osc(440), [0.5, 0.25]), 0.02)
  add('repeater', () => new Repeater(quick.mult(new Osc(123.4), 1)), 0.01)
  add('sum_const', () => quick.add(new Osc(50), 0.75), 0.01)
  add('mult_inlet_zero', () => { const m = new Multiply(new Osc(440), 2); m.B = 0; return m }, 0.01)

  // SURVEY.md §8f-1: elementwise maps (only when the library provides them)
  if (Subtract) {
    add('map_subtract', () => new Subtract(new Osc(300), new Multiply(new Osc(7), [0.5, 0.25])), 0.02) // 1 ch - 2 ch: no modulo broadcast
    add('map_subtract_quick', () => quick.subtract(new Osc(300, 'saw'), 0.3), 0.01)
    add('map_divide', () => new Divide(new Osc(200), new Sum(new Osc(3), 1.5)), 0.02)
    add('map_divide_by_zero', () => new Divide(new Osc(100), new Osc(50, 'square')), 0.01) // +-1 and a 0/x, x/0 mix -> Inf/NaN -> `|| 0`
    add('map_invert_abs', () => new Abs(quick.invert(new Osc(440.5))), 0.01)
    add('map_clip', () => { const c = new Clip(new Multiply(new Osc(2), 0.8)); c.IN = new Osc(333); return c }, 0.05)
    add('map_hardclip', () => new HardClipBelow(new HardClipAbove(new Osc(150), 0.5), [-0.25, -0.75]), 0.02)
    add('map_seconds', () => { const s = new SecondsToSamples(); s.IN = new Multiply(new Osc(10), 0.001); return s }, 0.01)
    add('map_fixedmultiply', () => new FixedMultiply(0.1, new Osc(441)), 0.01)
    add('map_gain', () => { const g = new Gain(new Multiply(new Osc(3), 12)); g.IN = new Osc(220); return g }, 0.05)
    add('map_db_semitone', () => new Multiply(new DecibelToScaler(new Multiply(new Osc(5), 20)), new SemitoneToRatio(new Multiply(new Osc(2), 7))), 0.05)
    add('map_pow', () => new Pow(new Sum(new Osc(100), 1.5), new Multiply(new Osc(1.5), 2)), 0.05)
    add('map_pow_negative_base', () => quick.pow(new Osc(100), 0.5), 0.01) // NaN for negative bases -> `|| 0`
    add('map_fm_semitone', () => new Osc(new Multiply(new SemitoneToRatio(new Multiply(new Osc(4), 12)), 220)), 0.05) // vibrato in semitones
  }

  // SURVEY.md §8f-2: delay / filter family and the per-channel oscillator
  if (FixedDelay) {
    add('fam_fixeddelay', () => { const d = new FixedDelay(0.0031); d.IN = new Osc(441); return d }, 0.05)
    add('fam_comb', () => { const c = new CombFilter(0.004, 0.7); c.IN = new Osc(150, 'saw'); return c }, 0.1)
    add('fam_comb_mod', () => { const c = new CombFilter(0.0007, new Multiply(new Osc(3), 0.9)); c.IN = new Osc(333.3); return c }, 0.1)
    add('fam_allpass_series', () => {
      const a = new AllPass(0.0021, 0.6); a.IN = new Osc(220, 'square')
      const b = new AllPass(0.0013, -0.45); b.IN = a
      return b
    }, 0.1)
    add('fam_allpass_loop', () => { // all-pass inside a feedback loop
      const s = new Sum(new Osc(180), 0)
      const a = new AllPass(0.0052, 0.5); a.IN = s
      s.B = new Multiply(a, 0.4)
      return a
    }, 0.1)
    add('fam_monodelay', () => new MonoDelay(new Osc(500), 123.5), 0.05)
    add('fam_monodelay_mod', () => new MonoDelay(new Osc(300), new Sum(new Multiply(new Osc(1.5), 30), 100)), 0.1)
    add('fam_readback', () => new ReadBackDelay(new Multiply(new Osc(400), [1, -1]), [100, 2000], 4096), 0.1)
    add('fam_readback_frac', () => new ReadBackDelay(new Osc(400), 10.5, 1024), 0.01) // fractional index -> undefined -> NaN -> `|| 0`
    add('fam_multiosc', () => new MultiChannelOsc([220, 330.5, 441.25]), 0.05)
    add('fam_multiosc_fm', () => new MultiChannelOsc(new Sum(new Multiply(new Osc(5), [20, 40]), 300), 'triangle'), 0.05)
    add('fam_multiosc_negative', () => new MultiChannelOsc(-100), 0.01) // no `phase < 0` fix-up in this unit: negative index -> NaN
  }

  // SURVEY.md §8f-1, the rest of the sweep: multi-inlet maps, channel plumbing, Timer, SampleRateRedux
  if (Pan) {
    const withIn = (u, x) => { u.IN = x; return u }
    add('rest_pan', () => new Pan(new Osc(440), new Osc(2)), 0.05) // stereo out, pan sweeping -1..1
    add('rest_pan_const', () => new Pan(new Osc(300.5, 'saw'), -0.3), 0.01)
    add('rest_pan_filter', () => new Filter(new Pan(new Osc(200, 'square'), 0.25), 1500), 0.02) // 2 channels through a Filter
    add('rest_midi_fm', () => new Osc(new MidiToFrequency(new Sum(new Multiply(new Osc(3), 12), 60))), 0.05) // "frequency" outlet feeding an inlet
    add('rest_midi_multi', () => new MidiToFrequency([60, 72]), 0.01) // only channel 0 is ever stored (MidiToFrequency.js:18)
    add('rest_rescale_fm', () => new Osc(withIn(new Rescale(-1, 1, 200, 800), new Osc(5))), 0.05)
    add('rest_rescale_2ch', () => withIn(new Rescale(new Multiply(new Osc(1), 0.5), 2, [0, 10], [1, 20]), new Multiply(new Osc(100), [1, 0.5])), 0.02)
    add('rest_rescale_default', () => withIn(new Rescale(), new Osc(441, 'triangle')), 0.01)
    add('rest_crossfader', () => new CrossFader(new Osc(220), new Multiply(new Osc(330, 'square'), [1, 0.5]),
      new Sum(new Multiply(new Osc(4), 0.5), 0.5)), 0.05) // b has a channel a lacks: silence, not a modulo broadcast
    add('rest_crossfader_const', () => new CrossFader(new Osc(100), 0.5, 0.25), 0.01)
    add('rest_vecmag', () => withIn(new VectorMagnitude(), new Multiply(new Osc(50), [1, 0.5, -2])), 0.02)
    add('rest_vecmag_2d', () => withIn(new VectorMagnitude(), new ConcatChannels(new Osc(100), new Osc(100.5, 'triangle'))), 0.02)
    add('rest_vecmag_default', () => new VectorMagnitude(), 0.01)
    add('rest_timer', () => new Timer(), 0.05)
    add('rest_timer_fm', () => new Osc(new Multiply(new Timer(), 4000)), 0.05)
    add('rest_srr', () => new SampleRateRedux(new Osc(440), 10), 0.02)
    add('rest_srr_mod', () => new SampleRateRedux(new Multiply(new Osc(300), [1, -1]), new Sum(new Multiply(new Osc(2), 20), 20.5)), 0.1)
    add('rest_srr_nan', () => { const r = new SampleRateRedux(new Osc(300), 5); r.AMMOUNT = NaN; return r }, 0.01) // `x > NaN` is false: nothing is ever sampled
    add('rest_srr_nan_later', () => new SampleRateRedux(new Osc(300), new Divide(new Osc(100), new Osc(100))), 0.02) // 0/0 once per 480 samples
    add('rest_concat', () => new ConcatChannels(new Multiply(new Osc(100), [1, 0.5]), new Osc(200)), 0.01) // 2 + 1 channels
    add('rest_concat_quick', () => quick.concat(new Osc(50), 0.25), 0.01)
    add('rest_pick', () => new PickChannel(new Multiply(new Osc(100), [1, 0.5, 0.25]), 4), 0.01) // 4 % 3 = channel 1
    add('rest_pick_default', () => new PickChannel(new Multiply(new Osc(60), [0.75, 0.5])), 0.01)
  }

  // SURVEY.md §8f-3: envelopes (idle until trigger(); the ev_ cases below trigger them from scheduled events)
  if (Shape) {
    add('env_shape_decay', () => new Shape('decay', 0.02).trigger(), 0.05)
    add('env_shape_idle', () => new Shape('attack', 0.01, 0.25, 0.75), 0.01) // never triggered: leftEdge 0 -> min
    add('env_shape_semisine_amp', () => new Multiply(new Osc(440), new Shape('semiSine', 0.03).trigger()), 0.05)
    add('env_shape_decaysq_range', () => new Shape('decaySquared', 0.013, -1, 2).trigger(), 0.03)
    add('env_shape_edges', () => { const s = new Shape('attack', 0.004, 0.5, 2); s.leftEdge = 'shape'; s.rightEdge = 0.5; return s.trigger() }, 0.01)
    add('env_shape_left_number', () => { const s = new Shape('decay', 0.01, -1, 1); s.leftEdge = 0.25; return s }, 0.01)
    add('env_shape_mod', () => new Shape('decay', new Sum(new Multiply(new Osc(20), 0.01), 0.02), new Multiply(new Osc(3), 0.5), 1).trigger(), 0.1)
    add('env_ahd', () => new AHD(0.01, 0.02, 0.03).trigger(), 0.08)
    add('env_ahd_idle', () => new AHD(0.01, 0.02, 0.03), 0.01)
    add('env_ahd_zero_hold', () => new AHD(0.005, 0, 0.005).trigger(), 0.02) // `hold || 0` = 0 s: samplePeriod / 0 = Infinity
    add('env_ahd_amp', () => new Multiply(new Osc(330), new AHD(0.002, 0.01, 0.02).trigger()), 0.05)
    add('env_ahd_mod', () => new AHD(new Sum(new Multiply(new Osc(50), 0.002), 0.004), 0.002, new Sum(new Multiply(new Osc(30), 0.01), 0.02)).trigger(), 0.08)
  }

  // SURVEY.md §8f-3: scheduled events (host callbacks at chunk boundaries).  `ev_` cases are rendered only
  // through the JS surface (tests/js/check_render.js): their descriptor alone does not carry the callbacks.
  add('ev_retrigger', () => {
    const r = new Ramp(2400, 1, 0)
    r.scheduleTrigger(0.01)
    r.scheduleTrigger(0.08)
    return new Multiply(new Osc(440), r)
  }, 0.15)
  add('ev_freq_steps', () => {
    const osc = new Osc(220)
    osc.schedule([0.02, 0.05, 0.09], function () { this.F = this.F.constant * 1.5 })
    return osc
  }, 0.12)
  add('ev_repeating', () => {
    const r = new Ramp(600, 1, 0.25)
    r.schedule(0.004, function () { this.trigger(); return 0.03 }) // a positive return value reschedules (Event.js:20-27)
    return new Multiply(new Osc(330.5, 'triangle'), r)
  }, 0.13)
  add('ev_filter_sweep', () => {
    const filt = new Filter(new Osc(100, 'saw'), 500)
    filt.schedule(0.03, function () { this.F = 2500 })
    filt.schedule(0.06, function () { this.F = 800 })
    return filt
  }, 0.1)
  // events on circuits that carry device memory from segment to segment: delay lines, CircleBuffers, feedback chunks
  add('ev_delay_retrigger', () => {
    const r = new Ramp(1200, 1, 0)
    r.scheduleTrigger([0.004, 0.03, 0.055])
    return new Delay(new Multiply(new Osc(660), r), 300.5, 2048)
  }, 0.08)
  add('ev_delay_time_change', () => {
    const d = new Delay(new Osc(500), 100, 4096)
    d.schedule(0.02, function () { this.DELAY = 1000.25 })
    d.schedule(0.05, function () { this.DELAY = 17 })
    return d
  }, 0.08)
  add('ev_delay_long_to_short', () => { // starts on the wave engine's write-once ring protocol, then the delay changes: the chain
    const d = new Delay(new Osc(500), 1000.25, 4096) // has to move to the read-modify-write protocol with the ring intact
    d.schedule(0.02, function () { this.DELAY = 300 })
    d.schedule(0.05, function () { this.DELAY = 17.5 })
    return d
  }, 0.08)
  add('ev_loop_delay_change', () => { // the same inside a feedback loop (the outlets' previous chunks move along)
    const sum = new Sum(new Osc(220), 0)
    const dl = new Delay(sum, 480, 4096)
    const f = new Filter(dl, 2000)
    sum.B = new Multiply(f, 0.6)
    dl.schedule(0.04, function () { this.DELAY = 700.5 })
    return f
  }, 0.1)
  add('ev_loop_gain', () => { // configs[3]'s feedback voice with the feedback gain changed on the fly
    const sum = new Sum(new Osc(220), 0)
    const f = new Filter(new Delay(sum, 480, 4096), 2000)
    const fb = new Multiply(f, 0.5)
    sum.B = fb
    fb.schedule(0.03, function () { this.B = 0.9 })
    fb.schedule(0.07, function () { this.B = 0.1 })
    return f
  }, 0.1)
  add('ev_feedback_no_delay', () => { // a bare feedback edge: only the previous CHUNK of the loop carries over
    const sum = new Sum(new Osc(330), 0)
    const m = new Multiply(sum, 0.5)
    sum.B = m
    m.schedule([0.011, 0.033], function () { this.B = this.B.constant * -1.5 })
    return sum
  }, 0.05)
  add('ev_circlebuffer', () => {
    const buffer = new CircleBuffer(1, 0.05)
    const writer = new CircleBufferWriter(buffer)
    writer.preWipe = true
    const osc = new Osc(330)
    writer.IN = osc
    const tap = new CircleBufferReader(buffer, 0.01)
    tap.chain(writer)
    osc.schedule(0.02, function () { this.F = 495 })
    tap.schedule(0.04, function () { this.OFFSET = 0.003 })
    return tap
  }, 0.07)
  if (CombFilter)
    add('ev_comb', () => {
      const c = new CombFilter(0.004, 0.7)
      const r = new Ramp(800, 1, 0)
      c.IN = new Multiply(new Osc(150, 'saw'), r)
      r.scheduleTrigger([0.0, 0.05])
      c.schedule(0.03, function () { this.FEEDBACKGAIN = -0.5 })
      return c
    }, 0.1)
  if (Shape) {
    add('ev_shape_retrigger', () => {
      const s = new Shape('decaySquared', 0.02)
      s.scheduleTrigger([0.01, 0.06])
      return new Multiply(new Osc(440), s)
    }, 0.1)
    add('ev_ahd_retrigger', () => { // trigger() does not reset t: the second attack starts from where stop() left it (AHD.js:23-27)
      const e = new AHD(0.004, 0.003, 0.006)
      e.scheduleTrigger([0.005, 0.04])
      return e
    }, 0.07)
  }
  // Retriggerer: a unit without a signal that calls target.trigger() every sampleRate / rate samples; ticked on the host
  // (`rt_` cases, like `ev_` ones, are rendered through the JS surface only)
  if (Retriggerer) {
    add('rt_ramp', () => {
      const r = new Ramp(2400, 1, 0).trigger()
      new Retriggerer(r, 8)
      return new Multiply(new Osc(440), r)
    }, 0.5)
    add('rt_shape_fast', () => { // fires every 960 samples: segments of three or four chunks
      const s = new Shape('decay', 0.01).trigger()
      new Retriggerer(s, 50)
      return new Multiply(new Osc(330.5, 'saw'), s)
    }, 0.2)
    add('rt_faster_than_chunks', () => { // 400 Hz: more than one crossing per chunk, still one trigger per chunk boundary
      const s = new Shape('decaySquared', 0.002).trigger()
      new Retriggerer(s, 400)
      return s
    }, 0.05)
    add('rt_two_targets_and_event', () => {
      const a = new Ramp(3000, 1, 0).trigger(), b = new AHD(0.002, 0.004, 0.01).trigger()
      const ra = new Retriggerer(a, 10)
      new Retriggerer(b, 23.5)
      ra.schedule(0.15, function () { this.RATE = 30 }) // the rate changes on the fly
      return new Sum(new Multiply(new Osc(220), a), new Multiply(new Osc(331), b))
    }, 0.3)
    // Retriggerers whose target is a Shape / AHD run on the DEVICE (descriptor opcode RETRIGGER): whole renders in one launch
    add('rt_dev_ahd', () => {
      const e = new AHD(0.002, 0.003, 0.008)
      new Retriggerer(e, 31.5)
      return new Multiply(new Osc(440, 'triangle'), e)
    }, 0.3)
    add('rt_dev_rhythm', () => { // two envelopes retriggered at different rates, mixed into a feedback delay line
      const a = new Shape('decay', 0.012).trigger(), b = new Shape('semiSine', 0.03)
      new Retriggerer(a, 8)
      new Retriggerer(b, 5.5)
      const sum = new Sum(new Sum(new Multiply(new Osc(330), a), new Multiply(new Osc(220.5, 'saw'), b)), 0)
      const d = new Delay(sum, 2400.5, 8192)
      sum.B = new Multiply(d, 0.4)
      return new Sum(sum, d)
    }, 0.6)
    add('rt_delay_line', () => { // retriggered bursts into a delay line: the ring has to survive the segment boundaries
      const r = new Ramp(600, 1, 0).trigger()
      new Retriggerer(r, 12)
      return new Delay(new Multiply(new Osc(700), r), 1000.5, 4096)
    }, 0.3)
  }
  // SporadicRetriggerer: one Math.random() per chunk decides whether the target is triggered; reproducible under a seeded
  // Math.random, in the reference and here alike (`seed` makes the generator and the checkers install one)
  if (L.SporadicRetriggerer) {
    const { SporadicRetriggerer } = L
    add('rt_sporadic', () => {
      const s = new Shape('decay', 0.01).trigger()
      new SporadicRetriggerer(s, 40)
      return new Multiply(new Osc(440), s)
    }, 0.5, undefined, { seed: 101 })
    add('rt_sporadic_two_and_regular', () => { // two random units and a regular one: the draws interleave chunk by chunk
      const a = new Shape('decay', 0.008).trigger(), b = new Ramp(1500, 1, 0).trigger(), c = new AHD(0.001, 0.002, 0.01)
      new SporadicRetriggerer(a, 60)
      new SporadicRetriggerer(b, 25)
      new Retriggerer(c, 17)
      return Sum.many([new Multiply(new Osc(330), a), new Multiply(new Osc(550, 'saw'), b), new Multiply(new Osc(110), c)])
    }, 0.5, undefined, { seed: 202 })
    add('rt_sporadic_with_events', () => { // a scheduled event that itself draws a number: the order of the draws is the reference's
      const s = new Shape('decaySquared', 0.02).trigger()
      const r = new SporadicRetriggerer(s, 30)
      const osc = new Osc(300)
      osc.schedule([0.1, 0.2, 0.3], function () { this.F = 200 + 400 * Math.random() })
      r.schedule(0.25, function () { this.RATE = 120 })
      return new Multiply(osc, s)
    }, 0.4, undefined, { seed: 303 })
    add('rt_sporadic_certain', () => { // rate * chunk / sampleRate >= 1: fires every chunk
      const s = new Shape('decay', 0.004).trigger()
      new SporadicRetriggerer(s, 400)
      return s
    }, 0.03, undefined, { seed: 404 })
  }
  if (Timer)
    add('ev_timer_trigger', () => {
      const tm = new Timer()
      tm.scheduleTrigger(0.02)
      return new Osc(new Multiply(tm, 20000))
    }, 0.05)

  /* Channel counts that grow AFTER the first chunk: a unit ticked before one of its inputs (feedback edge) sees that
   * input's previous chunk — one channel at first, more later — so its own outlet gains channels on the second chunk or
   * later (the reference grows channel lists lazily; the device runs those first chunks with their own op lists). */
  add('grow_feedback_stereo', () => {
    const sum = new Sum(new Osc(440), 0)
    sum.B = new Multiply(sum, [0.5, -0.25])
    return sum
  }, 0.03)
  add('grow_feedback_filter', () => {
    const sum = new Sum(new Osc(300, 'saw'), 0)
    const f = new Filter(sum, 1500)
    sum.B = new Multiply(f, [0.4, 0.2, -0.3])
    return f
  }, 0.04)
  if (MultiChannelOsc)
    add('grow_feedback_multiosc', () => {
      const f = new Sum(200, 0)
      const osc = new MultiChannelOsc(f)
      f.B = new Multiply(osc, [30, 50])
      return osc
    }, 0.04)
  add('grow_two_loops', () => { // the second loop only learns about the first one's channels a chunk later
    const a = new Sum(new Osc(220), 0), b = new Sum(new Osc(331, 'triangle'), 0)
    a.B = new Multiply(b, 0.5)
    b.B = new Multiply(a, [0.25, -0.5])
    return new Sum(a, b)
  }, 0.04)

  /* Patches (reference src/patches/): host-side builders over the units above.  Builders that draw random numbers run
   * under a seeded Math.random so that both libraries build the same graph. */
  const P = L.patches
  const seeded = (seed, fn) => () => {
    const original = Math.random, log = console.log
    let x = seed >>> 0
    Math.random = () => { x = (Math.imul(x, 1664525) + 1013904223) >>> 0; return x / 4294967296 }
    console.log = () => {} // a few reference patches print while they build
    try { return fn() } finally { Math.random = original; console.log = log }
  }
  if (P) {
    const quiet = (fn) => seeded(1, fn)
    add('patch_mixer', () => {
      const m = new P.Mixer(new Osc(220), new Osc(330.5, 'saw'))
      m.addInput(new Osc(441)).addMultiplied(new Osc(55, 'square'), 0.25).addAttenuated(new Osc(880), -12).addAttenuated(new Osc(3), 0)
      return m
    }, 0.02)
    add('patch_mixer_one', () => new P.Mixer(new Osc(220)), 0.01)
    add('patch_simple_delay', () =>
      new P.SimpleDelay(new Multiply(new Osc(300), new Ramp(2000, 1, 0).trigger()), 0.0123, 0.5, 0.3), 0.1)
    add('patch_simple_delay_defaults', () => new P.SimpleDelay(new Osc(100)), 0.02)
    add('patch_stereo_osc', () => { const o = new P.StereoOsc(62, -6, 0.3); o.waveform = 'saw'; return o }, 0.02)
    add('patch_stereo_osc_vibrato', () => {
      const o = new P.StereoOsc(57, 0, -0.5)
      o.PCONTROL = new P.LFO(6, 0.5, 0)
      return o
    }, 0.25)
    add('patch_lfo', () => new Osc(new P.LFO(3, 50, 400, 'triangle')), 0.5, [[0, 4096], [SR / 4, 2048]])
    add('patch_lfo_defaults', () => new P.LFO(), 0.02)
    add('patch_midi_osc', () => new P.MidiOsc(57), 0.02)
    add('patch_band_filter', quiet(() => new P.BandFilter(new Osc(110, 'saw'), 300, 2000)), 0.05)
    add('patch_multitap', () => {
      const burst = new Multiply(new Osc(500), new Ramp(700, 1, 0).trigger())
      const d = new P.MultiTapDelay(1, 4800, burst)
      const a = d.addTap(1000), b = d.addTap(2345.5), c = d.addFeedback(3000, 0.6, 0)
      return Sum.many([a, b, c])
    }, 0.2)
    add('patch_delay_mixer', () => {
      const d = new P.DelayMixer(1, 4800)
      d.addInput(new Multiply(new Osc(500), new Ramp(700, 1, 0).trigger()), 300)
      d.addInput(new Osc(3), 1000.5, 0.25)
      d.addInput(new Osc(220, 'saw'), 0, 0.1)
      return d
    }, 0.1)
    add('patch_space', () => new P.Space(new Osc(440), [0.5, 1]), 0.05)
    add('patch_space_moving', () =>
      new P.Space(new Osc(330, 'saw'), new ConcatChannels(new P.LFO(2, 3, 0), new P.LFO(3, 2, 1, 'triangle'))), 0.25)
    add('patch_space_stereo4', () => P.Space.stereo(new Osc(440), [-0.25, 2]), 0.03)
    add('patch_space_channel', () => {
      const c = new P.SpaceChannel([1, 1])
      c.IN = new Osc(200)
      c.PLACEMENT = [4, 5]
      return c
    }, 0.05)
    add('patch_scary', () => new P.ScaryPatch(new ConcatChannels(new Osc(200), new Osc(0.7)), 2), 0.1)
    add('patch_boop', () => new P.Boop(440, 0.02), 0.05)
    add('patch_sine_boop', () => new P.SineBoop(72, 0.03).trigger(), 0.05)
    add('patch_sine_boop_idle', quiet(() => new P.SineBoop()), 0.01)
    add('patch_space_boop', () => { const b = new P.SpaceBoop(64, 'saw', 0.04, 'decaySquared', [1, 0.5]); b.trigger(); return b }, 0.06)
    add('patch_space_boop_retuned', () => { const b = new P.SpaceBoop(); b.trigger(70, 0.02); return b }, 0.04)
    add('patch_fm_osc', seeded(7, () => {
      const o = new P.FMOsc(220)
      o.addModulatorOsc(110, 7)
      o.addModulator(new Osc(3), 0.5)
      return o
    }), 0.1)
    add('patch_fm_osc_cleared', seeded(8, () => { const o = new P.FMOsc(); o.addModulatorOsc(50, 12); o.clearModulation(); return o }), 0.01)
    add('patch_many_osc', () => P.ManyOsc.ofFrequencies(110, [1, 1.5, 2.01, 3.997]), 0.05)
    add('patch_many_osc_random', seeded(11, () => P.ManyOsc.random(5, 100, 2000)), 0.02)
    add('patch_many_osc_shared_f', () => P.ManyOsc.ofFrequencies(new P.LFO(2, 20, 200), [1, 2, 3]), 0.25, [[0, 2048], [SR / 8, 2048]])
    add('patch_stereo_detune', () => new MultiChannelOsc(new P.StereoDetune(220, 0.3)), 0.05)
    add('patch_stereo_detune_random', seeded(5, () => new MultiChannelOsc(P.StereoDetune.random(new P.LFO(1, 10, 300), 0.5))), 0.05)
    add('patch_frequency_group', () => {
      const fg = new P.FrequencyGroup(110)
      const h = fg.addHarmonic(1.5)
      return new Sum(new Osc(fg.fOuts[0]), new Osc(h))
    }, 0.03)
    add('patch_frequency_group_random', seeded(3, () => {
      const fg = new P.FrequencyGroup()
      return Sum.many(fg.addRandomHarmonics(4).map((f) => new Osc(f)))
    }), 0.03)
    add('patch_ap_stack', seeded(21, () => { const st = new P.APStack(4, 0.01, 0.5); st.IN = new Osc(220, 'saw'); return st }), 0.1)
    add('patch_ap_web', seeded(22, () => { const w = new P.APWeb(4, 0.01, 0.3); w.IN = new Osc(150, 'square'); return w }), 0.1)
    add('patch_attenuation_matrix', seeded(23, () => {
      const nodes = [new FixedDelay(0.004), new FixedDelay(0.0071), new Filter(0, 900)]
      const m = new P.AttenuationMatrix({ nodes, allowFeedback: false, pConnection: 0.8, pMix: 0.9, minAmmount: -12, maxAmmount: -3 })
      m.IN = new Multiply(new Osc(400), new Ramp(600, 1, 0).trigger())
      return m
    }), 0.1)
    add('patch_lfo_random', seeded(31, () => new Osc(P.LFO.randomInRange(8, 100, 900, 'saw'))), 0.25, [[0, 4096]])
    add('patch_shape_random', seeded(32, () => Shape.randomInRange(0.03, -1, 1).trigger()), 0.05)
    add('patch_all_pass_series', seeded(33, () => { const s = AllPass.manyRandomInSeries(3, 0.01, 0.6); s.IN.set(new Osc(300, 'saw')); return s.OUT }), 0.05)

  }
  // Noise: Math.random() per sample (or per 1/f), drawn on the host in tick order and handed to the device as an input stream
  if (L.Noise) {
    const { Noise } = L
    add('rt_noise', () => new Noise(), 0.02, undefined, { seed: 11 })
    add('rt_noise_held', () => new Multiply(new Noise(1000.5), new Osc(3)), 0.1, undefined, { seed: 12 }) // sample & hold at 1000.5 Hz
    add('rt_noise_into_delay', () => { // bursts of noise into a feedback delay line: inputs, rings and segments together
      const env = new Shape('decay', 0.01).trigger()
      new Retriggerer(env, 15)
      const sum = new Sum(new Multiply(new Noise(), env), 0)
      const d = new Delay(sum, 700.5, 4096)
      sum.B = new Multiply(d, 0.6)
      return d
    }, 0.3, undefined, { seed: 13 })
    add('rt_noise_and_sporadic', () => { // every kind of draw in one circuit: noise samples, sporadic triggers, an event
      const env = new Shape('decaySquared', 0.015).trigger()
      new L.SporadicRetriggerer(env, 50)
      const n1 = new Noise(), n2 = new Noise(4000)
      n2.schedule(0.12, function () { this.F = 500 + 1000 * Math.random() })
      return new Sum(new Multiply(n1, env), new Multiply(n2, 0.25))
    }, 0.25, undefined, { seed: 14 })
  }
  return cases
}

/* run fn (sync or async) under a seeded Math.random, then put the original back */
module.exports.withSeed = async function withSeed(seed, fn) {
  if (seed === undefined) return fn()
  const original = Math.random
  let x = seed >>> 0
  Math.random = () => { x = (Math.imul(x, 1664525) + 1013904223) >>> 0; return x / 4294967296 }
  try { return await fn() } finally { Math.random = original }
}
