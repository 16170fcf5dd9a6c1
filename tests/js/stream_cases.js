'use strict'
/* RenderStream cases, written once against an abstract unit library (see tests/js/cases.js): the reference's
 * RenderStream produces the golden frames (oracle/js/gen_golden_stream.js), this package's must reproduce them. */
module.exports = function streamCases(L, SR) {
  const { Osc, Ramp, Multiply, Sum, Delay, Filter, Pan } = L
  return [
    { name: 'stream_mono', channels: 1, chunks: 40, build: () => new Osc(440) },
    { name: 'stream_clipping', channels: 1, chunks: 40, build: () => new Multiply(new Osc(330.5), 2.5) }, // auto-normalised
    { name: 'stream_growing_stereo', channels: 2, chunks: 60, // the gain keeps shrinking while the ramp rises
      build: () => new Pan(new Multiply(new Osc(220), new Ramp(8000, 0.5, 3).trigger()), new Osc(3)) },
    { name: 'stream_events_delay', channels: 1, chunks: 48, build: () => { // events + a delay line across blocks
      const r = new Ramp(1200, 2, 0)
      r.scheduleTrigger([0.004, 0.03, 0.055, 0.2])
      return new Delay(new Multiply(new Osc(660), r), 300.5, 2048)
    } },
    { name: 'stream_feedback', channels: 1, chunks: 40, build: () => {
      const sum = new Sum(new Osc(220), 0)
      const f = new Filter(new Delay(sum, 480, 4096), 2000)
      sum.B = new Multiply(f, 0.9)
      return f
    } },
  ]
}
