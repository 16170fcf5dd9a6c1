'use strict'
/* Graph-construction classes for the GPU render path: the constructor surface of the
 * reference units this package executes, so graphs can be written exactly as with `dusp`:
 *
 *   const { Osc, Ramp, Multiply } = require('dusp-hip')
 *   renderChannelData(new Multiply(new Osc(440), new Ramp(48000, 1, 0).trigger()), 1)
 *
 * Nothing here ticks — there is no JS `_tick`: the classes only record structure and state
 * for lib/extract.js.  Behaviour restated from the reference (file:line under src/):
 *   ports and constants   Piglet.js:5-23, Inlet.js:40-93, Outlet.js:4-8
 *   accessor sugar        Unit.js:48-86  (unit.F = 440 | unit.F = otherUnit | unit.OUT)
 *   chain edges           Unit.js:88-105
 *   process index         Unit.js:171-209
 *   circuit flood fill    Circuit.js:67-107, ordering Circuit.js:125-148
 * The same objects also work with the real `dusp` package's graphs: extract.js is duck-typed.
 */
const config = require('./config')
const { WAVEFORMS, SHAPES } = require('./ops')

class Port {
  constructor(unit, name, opts = {}) {
    this.unit = unit
    this.name = name
    this.mono = !!opts.mono
    this.chunkSize = config.standardChunkSize
    this.sampleRate = config.sampleRate
  }
  get label() { return this.unit.label + '.' + this.name.toUpperCase() }
  get circuit() { return this.unit.circuit }
}

class Outlet extends Port {
  constructor(unit, name, opts) {
    super(unit, name, opts)
    this.connections = []
  }
  get isOutlet() { return true }
}

class Inlet extends Port {
  constructor(unit, name, opts) {
    super(unit, name, opts)
    this.connected = false
    this.outlet = null
    this.constant = 0
    // stands in for the reference's own constant-filled chunk: one f32 per channel
    this.signalChunk = { channelData: [new Float32Array(1)] }
  }
  get isInlet() { return true }
  set(val) { // Inlet.js:26-31
    if (val && (val.isUnit || val.isOutlet || val.isPatch)) this.connect(val)
    else this.setConstant(val)
  }
  get() { return this.connected ? this.outlet : this.constant }

  disconnect() {
    if (!this.outlet) return
    this.outlet.connections.splice(this.outlet.connections.indexOf(this), 1)
    this.outlet = null
    this.connected = false
    this.signalChunk = { channelData: this.signalChunk.channelData.map(() => new Float32Array(1)) }
  }

  setConstant(value) {
    if (this.outlet) this.disconnect()
    this.constant = value
    const vals = Array.isArray(value) ? value : [value]
    const chans = this.signalChunk.channelData
    for (let c = 0; c < chans.length || c < vals.length; c++) {
      chans[c] = chans[c] || new Float32Array(1)
      chans[c][0] = vals[c % vals.length]
    }
  }

  connect(target) {
    let outlet = target
    if (outlet.isUnit || outlet.isPatch) outlet = outlet.defaultOutlet
    if (this.connected) this.disconnect()
    this.connected = true
    this.outlet = outlet
    outlet.connections.push(this)
    const mine = this.unit.circuit, theirs = outlet.unit.circuit
    if (mine && theirs && mine !== theirs) throw 'SHIT: Circuit conflict' // the reference's message (Inlet.js:58)
    let touched = null
    if (mine) { mine.add(outlet.unit); touched = mine } else if (theirs) { theirs.add(this.unit); touched = theirs }
    if (touched) {
      this.unit.computeProcessIndex()
      outlet.unit.computeProcessIndex()
      touched.computeOrders()
    }
  }
}

/* Time-ordered host callbacks, run at the start of the chunk that contains them
 * (reference src/Event.js:3-30, src/Circuit.js:57-65, src/UnitOrPatch.js:9-33). */
class Event {
  constructor(time, func, unit, circuit) {
    this.time = time // seconds; `t` is the same instant in samples
    this.function = func
    this.unit = unit
    this.circuit = circuit
  }
  get time() { return this.t / config.sampleRate }
  set time(seconds) { this.t = seconds * config.sampleRate }
  run() {
    const again = this.function.call(this.unit || this.circuit || null)
    return again > 0 ? new Event(this.time + again, this.function, this.unit, this.circuit) : null
  }
}

function insertByTime(list, event) {
  for (let i = 0; i < list.length; i++)
    if (event.t < list[i].t) { list.splice(i, 0, event); return }
  list.push(event)
}

const timesUsed = new Map()

class Unit {
  constructor() {
    this.inlets = {}
    this.inletsOrdered = []
    this.outlets = {}
    this.outletsOrdered = []
    this.events = []
    this.promises = []
    this.circuit = undefined
    this.clock = 0
    this.tickInterval = config.standardChunkSize
    this.processIndex = undefined
    this.finished = false
    this.nChains = 0
    this.sampleRate = config.sampleRate
    const kind = this.constructor.name
    timesUsed.set(kind, (timesUsed.get(kind) || 0) + 1)
    this.label = kind + timesUsed.get(kind)
  }
  get isUnit() { return true }
  get isUnitOrPatch() { return true }

  addInlet(name, opts) {
    const inlet = new Inlet(this, name, opts)
    this.inlets[name] = inlet
    this.inletsOrdered.push(inlet)
    Object.defineProperty(this, name.toUpperCase(), {
      configurable: true,
      get: () => inlet,
      set: (val) => {
        if (val === undefined || val === null) throw 'Passed bad value to ' + inlet.label
        if (typeof val === 'number' || Array.isArray(val)) inlet.setConstant(val)
        else if (val.isOutlet || val.isUnit || val.isPatch) inlet.connect(val)
      },
    })
    return inlet
  }

  addOutlet(name, opts) {
    const outlet = new Outlet(this, name, opts)
    this.outlets[name] = outlet
    this.outletsOrdered.push(outlet)
    Object.defineProperty(this, name.toUpperCase(), { configurable: true, value: outlet })
    return outlet
  }

  chainAfter(unit) { // data-less ordering edge
    if (!unit || !unit.isUnit) throw 'chainAfter expects a Unit'
    this.addInlet('chain' + this.nChains++).connect(unit.addOutlet('chain' + unit.nChains++))
  }
  chain(unit) { return this.chainAfter(unit) }
  chainBefore(unit) { return unit.chainAfter(this) }

  get defaultInlet() { return this.inletsOrdered[0] }
  get defaultOutlet() { return this.outletsOrdered[0] }

  get inputUnits() {
    const found = []
    for (const name of Object.keys(this.inlets)) {
      const inlet = this.inlets[name]
      if (inlet.connected && !found.includes(inlet.outlet.unit)) found.push(inlet.outlet.unit)
    }
    return found
  }
  get outputUnits() {
    const found = []
    for (const name of Object.keys(this.outlets))
      for (const inlet of this.outlets[name].connections)
        if (!found.includes(inlet.unit)) found.push(inlet.unit)
    return found
  }

  /* 1 + the largest index among inputs not already on this walk, then push every dependent whose
   * index is not above ours.  `history` cuts feedback loops, and WHERE it cuts decides which edge of
   * a loop carries the implicit one-chunk delay. */
  computeProcessIndex(history) {
    const walk = (history || []).concat([this])
    let best = -1
    for (const unit of this.inputUnits) {
      if (walk.includes(unit)) continue
      if (unit.processIndex === undefined) unit.computeProcessIndex(walk)
      if (unit.processIndex > best) best = unit.processIndex
    }
    this.processIndex = best + 1
    for (const unit of this.outputUnits) {
      if (walk.includes(unit)) continue
      if (unit.processIndex === undefined || unit.processIndex <= this.processIndex) unit.computeProcessIndex(walk)
    }
    return this.processIndex
  }

  getOrBuildCircuit() { return this.circuit || new Circuit(this) }

  addEvent(event) {
    if (this.circuit) this.circuit.addEvent(event)
    else insertByTime(this.events, event)
  }
  schedule(time /* seconds */, func) {
    if (Array.isArray(time)) { for (const t of time) this.schedule(t, func); return }
    this.addEvent(new Event(time, func, this))
    return this
  }
  scheduleTrigger(t, val) {
    if (!this.trigger) throw this.label + ': cannot call scheduleTrigger because trigger is undefined'
    this.schedule(t, function () { this.trigger(val) })
  }
  finish() { // UnitOrPatch.js:77-84
    this.finished = true
    if (this._finish) this._finish()
    if (this.onFinish) this.onFinish()
  }
  scheduleFinish(t) { this.schedule(t, () => { this.finish() }) } // UnitOrPatch.js:85-90

  trigger() {
    for (const unit of this.inputUnits) unit.trigger()
    return this
  }
}

class Circuit {
  constructor(...units) {
    this.units = []
    this.tickIntervals = []
    this.clock = 0
    this.events = []
    this.promises = []
    for (const unit of units) this.add(unit)
  }

  add(unit) {
    if (unit.circuit && unit.circuit !== this) throw 'circuit clash, oh god ' + unit.label
    if (this.units.includes(unit)) return null
    this.units.push(unit)
    unit.circuit = this
    if (!this.tickIntervals.includes(unit.tickInterval)) {
      this.tickIntervals.push(unit.tickInterval)
      this.tickIntervals.sort((a, b) => a - b)
    }
    if (unit.events) {
      for (const e of unit.events) this.addEvent(e)
      unit.events = null // from now on the unit's events go straight to the circuit
    }
    if (unit.promises) {
      for (const p of unit.promises) this.promises.push(p)
      unit.promises = null
    }
    for (const other of unit.inputUnits) this.add(other)
    for (const other of unit.outputUnits) this.add(other)
    unit.computeProcessIndex()
    this.computeOrders()
    return true
  }

  addEvent(event) {
    event.circuit = this
    insertByTime(this.events, event)
  }

  /* run every event due before `beforeT` (samples); callbacks may return a delay in seconds to be run again */
  runEvents(beforeT) {
    while (this.events[0] && this.events[0].t < beforeT) {
      const followUp = this.events.shift().run()
      if (followUp) this.addEvent(followUp)
    }
  }

  computeOrders() {
    // same comparator as the reference: undefined indices give NaN, which the engine's stable sort
    // treats as "equal" while a flood fill is still in flight
    this.units.sort((a, b) => a.processIndex - b.processIndex)
    const gcd = (a, b) => { while (b) { [a, b] = [b, a % b] } return a }
    this.gcdTickInterval = this.tickIntervals.reduce(gcd)
  }
}

/* ------------------------------------------------------------------ units */

class Osc extends Unit { // Osc.js:7-17
  constructor(f, waveform) {
    super()
    this.addInlet('f', { mono: true })
    this.addOutlet('out', { mono: true })
    this.F = f || 440
    this.phase = 0
    this.waveform = waveform || 'sin'
  }
  get waveform() { return this._waveform }
  set waveform(w) {
    if (WAVEFORMS[w] === undefined) throw "waveform doesn't exist: " + w
    this._waveform = w
  }
}

class Ramp extends Unit { // Ramp.js:3-23 — (duration in SAMPLES, y0, y1); idle until trigger()
  constructor(duration, y0, y1) {
    super()
    this.addOutlet('out', { mono: true })
    this.duration = duration || this.sampleRate
    this.y0 = y0 || 1
    this.y1 = y1 || 0
    this.t = 0
    this.playing = false
  }
  trigger() {
    this.playing = true
    this.t = 0
    return this
  }
}

class Multiply extends Unit { // Multiply.js:4-12
  constructor(a, b) {
    super()
    this.addInlet('a')
    this.addInlet('b')
    this.addOutlet('out')
    this.A = a || 1
    this.B = b || 1
  }
}

class Sum extends Unit { // Sum.js + SignalCombiner.js:3-12
  constructor(a, b) {
    super()
    this.addInlet('a')
    this.addInlet('b')
    this.addOutlet('out')
    this.A = a || 0
    this.B = b || 0
  }
  static many(inputs) { // left-deep chain (Sum.js:18-29)
    if (inputs.length === 1) return inputs[0]
    let acc = new Sum(inputs[0], inputs[1])
    for (let i = 2; i < inputs.length; i++) acc = new Sum(acc, inputs[i])
    return acc
  }
}

class Filter extends Unit { // Filter.js:5-22
  constructor(input, f, kind) {
    super()
    this.addInlet('in')
    this.addInlet('f', { mono: true })
    this.addOutlet('out')
    if (input) this.IN = input
    if (f) this.F = f
    this.kind = kind || 'LP'
    if (this.kind !== 'LP' && this.kind !== 'HP') throw 'invalid filter type: ' + this.kind
    this.lastF = undefined
    // the reference's kind setter runs the coefficient function with f undefined (Filter.js:63)
    this.a0 = this.a2 = this.b1 = this.b2 = NaN
    this.a1 = this.kind === 'HP' ? 0 : NaN
    this.x1 = []; this.x2 = []; this.y1 = []; this.y2 = []
  }
}

class Delay extends Unit { // Delay.js:6-18 — delay in SAMPLES, ring of maxDelay samples
  constructor(input, delay, maxDelay) {
    super()
    this.addInlet('in')
    this.addInlet('delay')
    this.addOutlet('out')
    this.maxDelay = maxDelay || this.sampleRate * 5
    this.IN = input || 0
    this.DELAY = delay || 4410
  }
}

class CircleBuffer { // CircleBuffer.js:3-13
  constructor(numberOfChannels, lengthInSeconds) {
    this.numberOfChannels = numberOfChannels || 1
    this.lengthInSeconds = lengthInSeconds
    this.sampleRate = config.sampleRate
    this.lengthInSamples = Math.ceil(this.lengthInSeconds * this.sampleRate)
  }
}

class CircleBufferNode extends Unit { // CircleBufferNode.js:7-31
  constructor(buffer, offset) {
    super()
    this.t = 0
    this.buffer = buffer
    this.addInlet('offset')
    this.OFFSET = offset || 0
  }
}

class CircleBufferReader extends CircleBufferNode { // CircleBufferReader.js:4-10
  constructor(buffer, offset) {
    super(buffer, offset)
    this.addOutlet('out')
    this.postWipe = false
  }
}

class CircleBufferWriter extends CircleBufferNode { // CircleBufferWriter.js:4-10
  constructor(buffer, offset) {
    super(buffer, offset)
    this.addInlet('in')
    this.preWipe = false
  }
}

class Repeater extends Unit { // Repeater.js:3-11
  constructor(val, measuredIn) {
    super()
    this.addInlet('in')
    this.addOutlet('out')
    this.measuredIn = measuredIn
    this.IN = val || 0
  }
}

/* ---- elementwise maps (SURVEY.md §8f-1); constructor defaults as in src/components/<Name>.js */
const binary = (defA, defB) => class extends Unit {
  constructor(a, b) {
    super()
    this.addInlet('a')
    this.addInlet('b')
    this.addOutlet('out')
    this.A = a || defA
    this.B = b || defB
  }
}
const Subtract = class Subtract extends binary(0, 0) {} // Subtract.js:4-11
const Divide = class Divide extends binary(1, 1) {} // Divide.js:3-10
class Pow extends Unit { // Pow.js:4-11 — no defaults: undefined operands throw, as in the reference
  constructor(a, b) {
    super()
    this.addInlet('a')
    this.addInlet('b')
    this.addOutlet('out')
    this.A = a
    this.B = b
  }
}
const unary = (def) => class extends Unit {
  constructor(input) {
    super()
    this.addInlet('in')
    this.addOutlet('out')
    this.IN = input || def
  }
}
const PolarityInvert = class PolarityInvert extends unary(0) {} // PolarityInvert.js:4-9
const Abs = class Abs extends unary(0) {} // Abs.js:3-9
const DecibelToScaler = class DecibelToScaler extends unary(0) {} // DecibelToScaler.js:3-8
const SemitoneToRatio = class SemitoneToRatio extends unary(69) {} // SemitoneToRatio.js:3-8
class SecondsToSamples extends Unit { // SecondsToSamples.js:4-8 — no constructor argument
  constructor() {
    super()
    this.addInlet('in')
    this.addOutlet('out')
  }
}
class FixedMultiply extends Unit { // FixedMultiply.js:3-9 — (sf, input); sf is a plain number
  constructor(sf, input) {
    super()
    this.addInlet('in', { mono: true })
    this.addOutlet('out', { mono: true })
    this.sf = sf
    this.IN = input || 0
  }
}
class Clip extends Unit { // Clip.js:4-11 — (threshold) only; `in` stays 0 until set
  constructor(threshold) {
    super()
    this.addInlet('in')
    this.addInlet('threshold')
    this.addOutlet('out')
    this.THRESHOLD = threshold
  }
}
const hardClip = () => class extends Unit { // HardClipAbove.js:4-12 / HardClipBelow.js:4-12
  constructor(input, threshold) {
    super()
    this.addInlet('in')
    this.addInlet('threshold')
    this.addOutlet('out')
    this.IN = input || 0
    this.THRESHOLD = threshold || 0
  }
}
const HardClipAbove = class HardClipAbove extends hardClip() {}
const HardClipBelow = class HardClipBelow extends hardClip() {}
class Gain extends Unit { // Gain.js:3-10 — (gain in dB); `in` stays 0 until set
  constructor(gain) {
    super()
    this.addInlet('in')
    this.addInlet('gain', { mono: true })
    this.addOutlet('out')
    this.GAIN = gain || 0
  }
}

/* ---- delay / filter family (SURVEY.md §8f-2) */
class FixedDelay extends Unit { // FixedDelay.js:4-33 — (delayTime in seconds); `in` stays 0 until set
  constructor(delayTime) {
    super()
    this.addInlet('in', { mono: true })
    this.addOutlet('out', { mono: true })
    this.setSeconds(delayTime)
    this.tBuffer = 0
  }
  setDelayTime(tSamples) {
    if (!tSamples || tSamples < 0.5) throw 'Cannot have fixed delay of length 0 samples' // the reference's message
    this.delayTimeInSamples = Math.round(tSamples)
    this.delayTimeInSeconds = tSamples / this.sampleRate
  }
  setSeconds(duration) { this.setDelayTime(duration * this.sampleRate) }
  setFrequency(f) { this.setSeconds(1 / f) }
}
class CombFilter extends FixedDelay { // CombFilter.js:4-9
  constructor(delayTime, feedbackGain) {
    super(delayTime)
    this.addInlet('feedbackGain', { mono: true })
    this.FEEDBACKGAIN = feedbackGain || 0
  }
  set totalReverbTime(RVT) { this.FEEDBACKGAIN = Math.pow(0.001, this.delayTimeInSeconds / RVT) } // CombFilter.js:23-25
}
class AllPass extends CombFilter { // AllPass.js:4-7, random builders :19-51
  static random(maxDelayTime, maxFeedbackGain) {
    return new AllPass((maxDelayTime || 1) * Math.random(), (maxFeedbackGain || 1) * Math.random())
  }
  static manyRandom(n, maxDelay, maxFeedback) {
    const list = []
    for (let i = 0; i < n; i++) {
      Math.random() // the reference draws a third number here and never uses it (AllPass.js:29)
      list.push(new AllPass(Math.random() * maxDelay, Math.random() * maxFeedback))
    }
    return list
  }
  static manyRandomInSeries(n, maxDelayTime, maxFeedbackGain) {
    const list = []
    for (let i = 0; i < n; i++) {
      list[i] = AllPass.random(maxDelayTime, maxFeedbackGain)
      if (i !== 0) list[i].IN = list[i - 1].OUT
    }
    return { list, IN: list[0].IN, OUT: list[n - 1].OUT }
  }
}
class MonoDelay extends Unit { // MonoDelay.js:3-14 — delay in SAMPLES, fixed 5 s ring
  constructor(input, delay) {
    super()
    this.addInlet('in', { mono: true })
    this.addInlet('delay', { mono: true })
    this.addOutlet('out', { mono: true })
    this.maxDelay = this.sampleRate * 5
    this.IN = input || 0
    this.DELAY = delay || 4410
  }
}
class ReadBackDelay extends Unit { // ReadBackDelay.js:4-17
  constructor(input, delay, bufferLength) {
    super()
    this.addInlet('in')
    this.addInlet('delay')
    this.addOutlet('out')
    this.bufferLength = bufferLength || config.sampleRate
    this.tBuffer = 0
    this.IN = input || 0
    this.DELAY = delay || 0
  }
}
class MultiChannelOsc extends Unit { // Osc/MultiChannelOsc.js:7-17 — one phase per channel of f
  constructor(f, waveform) {
    super()
    this.addInlet('f')
    this.addOutlet('out')
    this.F = f || 440
    this.phase = []
    this.waveform = waveform || 'sin'
  }
  get waveform() { return this._waveform }
  set waveform(w) {
    if (WAVEFORMS[w] === undefined) throw "waveform doesn't exist: " + w
    this._waveform = w
  }
  resetPhase() { for (const i in this.phase) this.phase[i] = 0 }
  randomPhaseFlip() { // MultiChannelOsc.js:58-62 (a no-op on a fresh unit: no channel has a phase yet)
    if (Math.random() < 0.5) for (const i in this.phase) this.phase[i] += config.sampleRate / 2
  }
}

/* ---- rest of the elementwise sweep (SURVEY.md §8f-1) */
class Pan extends Unit { // Pan.js:3-13 — mono in, stereo out, constant-power-ish compensation
  constructor(input, pan) {
    super()
    this.addInlet('in', { mono: true })
    this.addInlet('pan', { mono: true })
    this.addOutlet('out', { numberOfChannels: 2 })
    this.PAN = pan || 0
    this.IN = input || 0
    this.compensationDB = 1.5
  }
}
class MidiToFrequency extends Unit { // MidiToFrequency.js:3-9 — the data outlet is called "frequency"
  constructor(midi) {
    super()
    this.addInlet('midi')
    this.addOutlet('frequency')
    this.MIDI = midi || 69
  }
}
class Rescale extends Unit { // Rescale.js:3-17 — `in` is not a constructor argument
  constructor(inLower, inUpper, outLower, outUpper) {
    super()
    for (const n of ['in', 'inLower', 'inUpper', 'outLower', 'outUpper']) this.addInlet(n)
    this.addOutlet('out')
    this.IN = 0
    this.INLOWER = inLower || -1
    this.INUPPER = inUpper || 1
    this.OUTLOWER = outLower || 0
    this.OUTUPPER = outUpper || 1
  }
  get isRescale() { return true }
}
class CrossFader extends Unit { // CrossFader.js:3-14 — dial 0: all A, 1: all B
  constructor(a, b, dial) {
    super()
    this.addInlet('a')
    this.addInlet('b')
    this.addInlet('dial', { mono: true })
    this.addOutlet('out')
    this.A = a || 0
    this.B = b || 0
    this.DIAL = dial || 0
  }
}
class VectorMagnitude extends Unit { // vector/VectorMagnitude.js:5-11 — no constructor argument
  constructor() {
    super()
    this.addInlet('in')
    this.addOutlet('out', { mono: true })
    this.IN = [0, 0]
  }
}
class Timer extends Unit { // Timer.js:26-32,43-45
  constructor() {
    super()
    this.addOutlet('out', { mono: true })
    this.t = 0
    this.samplePeriod = 1 / this.sampleRate
  }
  trigger() { this.t = 0 }
}
class SampleRateRedux extends Unit { // SampleRateRedux.js:3-16 — sample & hold every `ammount` samples
  constructor(input, ammount) {
    super()
    this.addInlet('in')
    this.addInlet('ammount', { mono: true })
    this.addOutlet('out')
    this.val = [0]
    this.timeSinceLastUpdate = Infinity
    this.IN = input || 0
    this.AMMOUNT = ammount || 0
  }
}
class ConcatChannels extends Unit { // ConcatChannels.js:3-11
  constructor(a, b) {
    super()
    this.addInlet('a')
    this.addInlet('b')
    this.addOutlet('out')
    this.A = a || 0
    this.B = b || 0
  }
}
class PickChannel extends Unit { // PickChannel.js:3-11
  constructor(input, c) {
    super()
    this.addInlet('in')
    this.addInlet('c', { mono: true })
    this.addOutlet('out', { mono: true })
    this.IN = input || 0
    this.C = c || 0
  }
}

/* ---- envelopes (SURVEY.md §8f-3) */
class Shape extends Unit { // Shape/index.js:7-23,107-122 — a table read once over `duration` seconds after trigger()
  constructor(shape, durationInSeconds, min, max) {
    super()
    this.addInlet('duration', { mono: true })
    this.addInlet('min', { mono: true })
    this.addInlet('max', { mono: true })
    this.addOutlet('out', { mono: true })
    this.t = 0
    this.playing = false
    this.finished = false
    this.leftEdge = 0
    this.rightEdge = 'shape'
    this.shape = shape || 'decay'
    this.DURATION = durationInSeconds || 1
    this.MIN = min || 0
    this.MAX = max || 1
  }
  get shape() { return this._shape }
  set shape(shape) {
    if (SHAPES[shape] === undefined) throw this.label + ':\n\tinvalid shape function: ' + shape
    this._shape = shape
  }
  trigger() { this.playing = true; this.t = 0; return this }
  stop() { this.playing = false }
  randomDecay(maxDuration) { // Shape/index.js:157-162
    this.shape = 'decay'
    this.DURATION = Math.random() * (maxDuration || 5)
    this.MIN = 0
    this.MAX = 1
  }
  static randomShapeStr() { // Shape/index.js:145-148 — in the key order of the reference's shape tables
    const keys = Object.keys(SHAPES)
    return keys[Math.floor(Math.random() * keys.length)]
  }
  static randomDecay(maxDuration) { return new Shape('decaySquared', Math.random() * (maxDuration || 5)) }
  static randomInRange(maxDuration, minMin, maxMax) { // Shape/index.js:124-143
    const a = minMin + Math.random() * (maxMax - minMin)
    const b = minMin + Math.random() * (maxMax - minMin)
    return new Shape(Shape.randomShapeStr(), Math.random() * (maxDuration || 1), Math.min(a, b), Math.max(a, b))
  }
}
class AHD extends Unit { // AHD.js:6-34 — attack / hold / decay times in seconds
  constructor(attack, hold, decay) {
    super()
    this.addInlet('attack', { mono: true })
    this.addInlet('hold', { mono: true })
    this.addInlet('decay', { mono: true })
    this.addOutlet('out', { mono: true })
    this.ATTACK = attack || 0
    this.HOLD = hold || 0
    this.DECAY = decay || 0
    this.state = 0
    this.playing = false
    this.t = 0
  }
  trigger() { this.state = 1; this.playing = true; return this }
  stop() { this.state = 0; this.playing = false; return this }
}

/* ---- host-ticked units */
class Retriggerer extends Unit { // Retriggerer.js:3-43 — calls target.trigger() every sampleRate / rate samples
  constructor(target, rate) {
    super()
    this.addInlet('rate', { mono: true })
    if (target) this.target = target
    this.t = 0
    this.RATE = rate || 1
  }
  get target() { return this._target }
  set target(target) {
    if (target) {
      this._target = target
      this.chainBefore(target) // ticks before its target, so a trigger takes effect in the same chunk
    }
  }
  /* The unit produces no signal: its _tick only runs an accumulator and calls trigger() (Retriggerer.js:13-24), and
   * because it is ordered before its target the trigger acts at the START of the chunk in which the accumulator
   * crosses the sample rate.  The GPU path therefore ticks it on the host, chunk by chunk, and treats a firing chunk
   * as a segment boundary.  Needs a constant rate (the signal would have to be read back from the device). */
  get isHostTicked() { return true }
  rateConstant() {
    if (this.RATE.connected) throw 'dusp-hip: Retriggerer with a signal-rate `rate` is not supported on the GPU path (' + this.label + ')'
    return this.RATE.signalChunk.channelData[0][0]
  }
  /* Host-tick protocol (lib/renderChannelData.js SegmentRenderer): hostTick() runs the unit's tick for the chunk that is
   * about to be rendered; peekBegin() / peekNext() then look ahead one chunk at a time WITHOUT touching the unit's state
   * (would it fire in that chunk?), and skipQuiet(n) commits n looked-at chunks that fired nothing. */
  hostTick(chunkSize) {
    const rate = this.rateConstant()
    let fired = false
    for (let k = 0; k < chunkSize; k++) {
      this.t += rate
      if (this.t >= this.sampleRate) { fired = true; if (this._target && this._target.trigger) this._target.trigger(); this.t -= this.sampleRate }
    }
    return fired
  }
  peekBegin() { this._peekT = this.t }
  peekNext(chunkSize) {
    const rate = this.rateConstant()
    let fired = false
    for (let k = 0; k < chunkSize; k++) {
      this._peekT += rate
      if (this._peekT >= this.sampleRate) { fired = true; this._peekT -= this.sampleRate }
    }
    return fired
  }
  skipQuiet(chunkSize, chunks) { const rate = this.rateConstant(); for (let k = 0; k < chunks * chunkSize; k++) this.t += rate }
}

/* SporadicRetrigger.js:3-31 — every chunk, with probability rate * chunkSize / sampleRate, calls target.trigger().  One
 * Math.random() per chunk and unit, in tick order: the host ticks it exactly so (looking ahead draws the numbers of the
 * coming chunks in the same chunk-major order and keeps them on a tape), which makes a render reproducible under a
 * seeded Math.random just like the reference's. */
class SporadicRetriggerer extends Unit {
  constructor(target, rate) {
    super()
    this.addInlet('rate', { mono: true })
    if (target) this.target = target
    this.RATE = rate || 1
    this._tape = [] // numbers already drawn for chunks not yet rendered
  }
  get target() { return this._target }
  set target(target) {
    if (target) {
      this._target = target
      this.chainBefore(target)
    }
  }
  get isHostTicked() { return true }
  rateConstant() {
    if (this.RATE.connected) throw 'dusp-hip: SporadicRetriggerer with a signal-rate `rate` is not supported on the GPU path (' + this.label + ')'
    return this.RATE.signalChunk.channelData[0][0]
  }
  fires(r, chunkSize) { return r < this.rateConstant() * chunkSize / this.sampleRate }
  hostTick(chunkSize) {
    if (!(this._target && this._target.trigger)) return false // (the reference draws nothing without a target)
    const r = this._tape.length ? this._tape.shift() : Math.random()
    if (!this.fires(r, chunkSize)) return false
    this._target.trigger()
    return true
  }
  peekBegin() { this._peekAt = 0 }
  peekNext(chunkSize) {
    if (!(this._target && this._target.trigger)) return false
    if (this._peekAt >= this._tape.length) this._tape.push(Math.random())
    return this.fires(this._tape[this._peekAt++], chunkSize)
  }
  skipQuiet(chunkSize, chunks) { if (this._target && this._target.trigger) this._tape.splice(0, chunks) }
}

/* A unit whose signal the HOST computes (descriptor opcode INPUT): hostChunk(chunkSize) returns the unit's next chunk,
 * called once per chunk in the circuit's tick order, exactly like a `_tick` would be.  The segment loop of
 * lib/renderChannelData.js collects the chunks of a segment and hands them to the device as an input stream, where the
 * rest of the circuit reads them like any other outlet.  Looking ahead (peekNext) computes coming chunks early, in the
 * same chunk-major order as everything else that is ticked on the host, and keeps them on a tape. */
class HostSignal extends Unit {
  constructor() {
    super()
    this.addOutlet('out', { mono: true })
    this._tape = []    // chunks computed ahead of time
    this._segment = [] // chunks of the segment being assembled
  }
  get isHostSignal() { return true }
  get isHostTicked() { return true }
  hostChunk(chunkSize) { return new Float32Array(chunkSize) } // silence; subclasses compute their signal here
  hostTick(chunkSize) { this._segment.push(this._tape.length ? this._tape.shift() : this.hostChunk(chunkSize)); return false }
  peekBegin() { this._peekAt = 0 }
  peekNext(chunkSize) {
    if (this._peekAt >= this._tape.length) this._tape.push(this.hostChunk(chunkSize))
    this._peekAt++
    return false
  }
  skipQuiet(chunkSize, chunks) { for (let k = 0; k < chunks; k++) this._segment.push(this._tape.shift()) }
  takeSegment(nSamples, into, at) { // this segment's samples (the last chunk may be cut short) -> into[at ..]
    let done = 0
    for (const chunk of this._segment) {
      const n = Math.min(chunk.length, nSamples - done)
      if (n > 0) into.set(n === chunk.length ? chunk : chunk.subarray(0, n), at + done)
      done += chunk.length
    }
    this._segment = []
  }
}

/* Noise.js:3-27 — a sample-and-hold of Math.random() at rate f (default: a fresh number every sample).  The numbers can
 * only be drawn by the JavaScript engine, in tick order, so the unit is a HostSignal; with a seeded Math.random a render
 * is reproducible, here as in the reference. */
class Noise extends HostSignal {
  constructor(f) {
    super()
    this.addInlet('f')
    this.F = f || this.sampleRate
    this.phase = 0
    this.y = Math.random() * 2 - 1
  }
  hostChunk(chunkSize) {
    if (this.F.connected) throw 'dusp-hip: Noise with a signal-rate `f` is not supported on the GPU path (' + this.label + ')'
    const f = this.F.signalChunk.channelData[0][0], out = new Float32Array(chunkSize)
    for (let t = 0; t < chunkSize; t++) {
      this.phase += f
      if (this.phase >= this.sampleRate) { this.phase = 0; this.y = 2 * Math.random() - 1 }
      out[t] = this.y
    }
    return out
  }
}

module.exports = { Retriggerer, SporadicRetriggerer, HostSignal, Noise, Shape, AHD, Pan, MidiToFrequency, Rescale, CrossFader, VectorMagnitude, Timer, SampleRateRedux, ConcatChannels, PickChannel,
  FixedDelay, CombFilter, AllPass, MonoDelay, ReadBackDelay, MultiChannelOsc, Event, Subtract, Divide, Pow, PolarityInvert, Abs, DecibelToScaler, SemitoneToRatio, SecondsToSamples,
  FixedMultiply, Clip, HardClipAbove, HardClipBelow, Gain,
  Unit, Inlet, Outlet, Circuit, Osc, Ramp, Multiply, Sum, Filter, Delay,
  CircleBuffer, CircleBufferNode, CircleBufferReader, CircleBufferWriter, Repeater }
