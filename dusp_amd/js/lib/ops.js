'use strict'
/* Opcode table of the flat render-program descriptor ("DUSP words", see
 * include/dusp_hip.h and DESIGN.md §3).  The same numbers live in
 * dusp_amd/descriptor.py, dusp_amd/csrc/program.hpp and oracle/dusp_oracle.c;
 * tests/test_descriptor.py keeps the four in step.
 *
 * The descriptor is a Float64Array so that JS, Python and C can all produce /
 * parse it with no framing code: every field is one little-endian f64 word.
 */

const MAGIC = 1146442576 // 'DUSP' big-endian as an integer
const VERSION = 1
const HEADER_WORDS = 12

const OP = Object.freeze({
  OSC: 1, RAMP: 2, MULTIPLY: 3, SUM: 4, FILTER: 5, DELAY: 6,
  CB_READER: 7, CB_WRITER: 8, REPEATER: 9,
  // elementwise maps (SURVEY.md §8f-1)
  SUBTRACT: 10, DIVIDE: 11, POLARITY_INVERT: 12, ABS: 13, CLIP: 14, HARD_CLIP_ABOVE: 15, HARD_CLIP_BELOW: 16,
  SECONDS_TO_SAMPLES: 17, FIXED_MULTIPLY: 18, GAIN: 19, DECIBEL_TO_SCALER: 20, SEMITONE_TO_RATIO: 21, POW: 22,
  // delay / filter family and the per-channel oscillator (SURVEY.md §8f-2)
  FIXED_DELAY: 23, COMB_FILTER: 24, ALL_PASS: 25, MONO_DELAY: 26, READBACK_DELAY: 27, MULTI_OSC: 28,
  // rest of the elementwise sweep (SURVEY.md §8f-1): multi-inlet maps, channel plumbing, two small stateful units
  PAN: 29, MIDI_TO_FREQUENCY: 30, RESCALE: 31, CROSS_FADER: 32, VECTOR_MAGNITUDE: 33, TIMER: 34, SAMPLE_RATE_REDUX: 35,
  CONCAT_CHANNELS: 36, PICK_CHANNEL: 37,
  // envelopes driven by trigger() events (SURVEY.md §8f-3)
  SHAPE: 38, AHD: 39,
  HOST_ONLY: 40, // no signal: the unit acts through host callbacks between segments (Retriggerer)
  RETRIGGER: 42, // a Retriggerer whose target is a Shape / AHD of the same circuit: runs on the device (attribute = target unit)
  INPUT: 41, // a signal the HOST computes chunk by chunk (Noise: Math.random() per sample); attribute = stream index
})

const INLET = Object.freeze({ CONST: 0, CONNECT: 1, PARAM: 2 })

const WAVEFORMS = Object.freeze({ sin: 0, sine: 0, saw: 1, square: 2, triangle: 3, '8bit': 4 })
const WAVEFORM_NAMES = Object.freeze(['sin', 'saw', 'square', 'triangle', '8bit'])
const FILTER_KINDS = Object.freeze({ LP: 0, HP: 1 })
/* Shape's lookup tables (reference src/components/Shape/shapeTables.js:21-38) share the table space of the wave tables */
const SHAPES = Object.freeze({ decay: 5, attack: 6, semiSine: 7, decaySquared: 8 })

/* constructor name -> { op, inlets (data inlets, in descriptor order), outlet (name of the data outlet, default "out") } */
const UNITS = Object.freeze({
  Osc: { op: OP.OSC, inlets: ['f'] },
  Ramp: { op: OP.RAMP, inlets: [] },
  Multiply: { op: OP.MULTIPLY, inlets: ['a', 'b'] },
  Sum: { op: OP.SUM, inlets: ['a', 'b'] },
  Filter: { op: OP.FILTER, inlets: ['in', 'f'] },
  Delay: { op: OP.DELAY, inlets: ['in', 'delay'] },
  CircleBufferReader: { op: OP.CB_READER, inlets: ['offset'] },
  CircleBufferWriter: { op: OP.CB_WRITER, inlets: ['offset', 'in'] },
  Repeater: { op: OP.REPEATER, inlets: ['in'] },
  Subtract: { op: OP.SUBTRACT, inlets: ['a', 'b'] },
  Divide: { op: OP.DIVIDE, inlets: ['a', 'b'] },
  PolarityInvert: { op: OP.POLARITY_INVERT, inlets: ['in'] },
  Abs: { op: OP.ABS, inlets: ['in'] },
  Clip: { op: OP.CLIP, inlets: ['in', 'threshold'] },
  HardClipAbove: { op: OP.HARD_CLIP_ABOVE, inlets: ['in', 'threshold'] },
  HardClipBelow: { op: OP.HARD_CLIP_BELOW, inlets: ['in', 'threshold'] },
  SecondsToSamples: { op: OP.SECONDS_TO_SAMPLES, inlets: ['in'] },
  FixedMultiply: { op: OP.FIXED_MULTIPLY, inlets: ['in'] },
  Gain: { op: OP.GAIN, inlets: ['in', 'gain'] },
  DecibelToScaler: { op: OP.DECIBEL_TO_SCALER, inlets: ['in'] },
  SemitoneToRatio: { op: OP.SEMITONE_TO_RATIO, inlets: ['in'] },
  Pow: { op: OP.POW, inlets: ['a', 'b'] },
  FixedDelay: { op: OP.FIXED_DELAY, inlets: ['in'] },
  CombFilter: { op: OP.COMB_FILTER, inlets: ['in', 'feedbackGain'] },
  AllPass: { op: OP.ALL_PASS, inlets: ['in', 'feedbackGain'] },
  MonoDelay: { op: OP.MONO_DELAY, inlets: ['in', 'delay'] },
  ReadBackDelay: { op: OP.READBACK_DELAY, inlets: ['in', 'delay'] },
  MultiChannelOsc: { op: OP.MULTI_OSC, inlets: ['f'] },
  Pan: { op: OP.PAN, inlets: ['in', 'pan'] },
  MidiToFrequency: { op: OP.MIDI_TO_FREQUENCY, inlets: ['midi'], outlet: 'frequency' },
  Rescale: { op: OP.RESCALE, inlets: ['in', 'inLower', 'inUpper', 'outLower', 'outUpper'] },
  CrossFader: { op: OP.CROSS_FADER, inlets: ['a', 'b', 'dial'] },
  VectorMagnitude: { op: OP.VECTOR_MAGNITUDE, inlets: ['in'] },
  Timer: { op: OP.TIMER, inlets: [] },
  SampleRateRedux: { op: OP.SAMPLE_RATE_REDUX, inlets: ['in', 'ammount'] },
  ConcatChannels: { op: OP.CONCAT_CHANNELS, inlets: ['a', 'b'] },
  PickChannel: { op: OP.PICK_CHANNEL, inlets: ['in', 'c'] },
  Shape: { op: OP.SHAPE, inlets: ['duration', 'min', 'max'] },
  AHD: { op: OP.AHD, inlets: ['attack', 'hold', 'decay'] },
  Retriggerer: { op: OP.HOST_ONLY, inlets: [], hostTick: true }, // (unless lib/extract.js deviceRetrigger() says it can run on the device)
  DeviceRetriggerer: { op: OP.RETRIGGER, inlets: ['rate'] },
  SporadicRetriggerer: { op: OP.HOST_ONLY, inlets: [], hostTick: true },
  Noise: { op: OP.INPUT, inlets: [], hostTick: true },
  HostSignal: { op: OP.INPUT, inlets: [], hostTick: true }, // (and every subclass: lib/graph.js)
})

module.exports = { MAGIC, VERSION, HEADER_WORDS, OP, INLET, WAVEFORMS, WAVEFORM_NAMES, FILTER_KINDS, SHAPES, UNITS }
