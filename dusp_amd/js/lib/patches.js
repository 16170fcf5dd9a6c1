'use strict'
/* Patches: host-side builders of sub-circuits with aliased inlets / outlets (reference src/Patch.js and src/patches/).
 * A Patch is not a unit and never reaches the device: it wires ordinary units together, and renderChannelData / an
 * inlet follow `defaultOutlet` to the unit behind it.  Only deterministic patches made of units the GPU path runs are
 * provided (FMOsc, FMSynth, Worm, ... draw random numbers or use spectral / noise units). */
const g = require('./graph')
const quick = require('./quick')
const config = require('./config')

const timesUsed = new Map()

class Patch { // Patch.js:6-119
  constructor() {
    this.inlets = {}
    this.outlets = {}
    this.inletsOrdered = []
    this.outletsOrdered = []
    this.units = []
    this.finished = false
    const kind = this.constructor.name
    timesUsed.set(kind, (timesUsed.get(kind) || 0) + 1)
    this.label = kind + timesUsed.get(kind)
  }
  get isPatch() { return true }
  get isUnitOrPatch() { return true }
  get defaultInlet() { return this.inletsOrdered[0] }
  get defaultOutlet() { return this.outletsOrdered[0] }

  _alias(port, name, table, ordered) {
    if (name === undefined) { // first free name: in, in1, in2, ...
      name = port.name
      for (let n = 1; table[name]; n++) name = port.name + n
    }
    table[name] = port
    ordered.push(port)
    const upper = port.name.toUpperCase()
    Object.defineProperty(this, name.toUpperCase(), {
      configurable: true,
      get: () => port.unit[upper],
      set: (val) => { port.unit[upper] = val },
    })
  }
  aliasInlet(inlet, name) {
    if (inlet.isUnit || inlet.isPatch) inlet = inlet.inletsOrdered[0]
    this._alias(inlet, name, this.inlets, this.inletsOrdered)
  }
  aliasOutlet(outlet, name) {
    if (outlet.isUnit || outlet.isPatch) outlet = outlet.defaultOutlet
    this._alias(outlet, name, this.outlets, this.outletsOrdered)
  }
  alias(port, name) {
    if (port.isInlet) this.aliasInlet(port, name)
    else if (port.isOutlet) this.aliasOutlet(port, name)
  }
  addUnit(unit) { if (unit.isUnit || unit.isPatch) { this.units.push(unit); unit.ownerPatch = this } }
  addUnits(...units) { for (const u of units) if (Array.isArray(u)) u.forEach((x) => this.addUnit(x)); else this.addUnit(u) }

  addEvent(event) {
    if (!this.units[0]) throw 'Could not add event as Patch posseses no units: ' + this.label
    this.units[0].addEvent(event)
  }
  schedule(time, func) { // UnitOrPatch.js:9-23: the callback's `this` is the patch
    if (Array.isArray(time)) { for (const t of time) this.schedule(t, func); return }
    this.addEvent(new g.Event(time, func, this))
    return this
  }
  scheduleTrigger(t, val) {
    if (!this.trigger) throw this.label + ': cannot call scheduleTrigger because trigger is undefined'
    this.schedule(t, function () { this.trigger(val) })
  }
  finish() { this.finished = true; if (this._finish) this._finish(); if (this.onFinish) this.onFinish() }
  scheduleFinish(t) { this.schedule(t, () => { this.finish() }) }
  trigger() {
    for (const u of this.units) if (u.trigger) u.trigger()
    return this
  }
}

class Mixer extends Patch { // patches/Mixer.js:7-58 — a right-deep chain of Sums behind a Repeater
  constructor(...inputs) {
    super()
    this.sums = []
    this.inputs = []
    this.addUnits(this.addRepeater = new g.Repeater(0))
    this.aliasOutlet(this.addRepeater.OUT)
    for (const x of inputs) this.addInput(x)
  }
  get numberOfInputs() { return this.inputs.length }
  addInput(outlet) {
    if (!outlet.isOutlet && outlet.defaultOutlet) outlet = outlet.defaultOutlet
    if (this.inputs.length === 0) this.addRepeater.IN = outlet
    else if (this.inputs.length === 1) {
      const sum = new g.Sum(this.addRepeater.IN.outlet, outlet)
      this.addRepeater.IN = sum
      this.sums.push(sum)
    } else {
      const last = this.sums[this.sums.length - 1]
      const sum = new g.Sum(last.B.outlet, outlet)
      last.B = sum
      this.sums.push(sum)
    }
    this.inputs.push(outlet)
    return this
  }
  addMultiplied(outlet, sf) { return sf ? this.addInput(new g.Multiply(outlet, sf)) : this.addInput(outlet) }
  addAttenuated(outlet, gain) {
    if (!gain) return this.addInput(outlet)
    const unit = new g.Gain()
    unit.IN = outlet
    unit.GAIN = gain
    return this.addInput(unit)
  }
  addInputs(...xs) { for (const x of xs) if (Array.isArray(x)) x.forEach((y) => this.addInput(y)); else this.addInput(x); return this }
}

class SimpleDelay extends Patch { // patches/SimpleDelay.js:9-41 — delay in SECONDS, feedback, dry/wet cross-fade
  constructor(input, delay, feedback, dryWet) {
    super()
    this.addUnits(
      this.inputRepeater = new g.Repeater(),
      this.feedbackInputSum = new g.Sum(),
      this.delayer = new g.Delay(),
      this.mixDryWet = new g.CrossFader(),
      this.feedbackScaler = new g.Multiply(),
      this.delayScaler = new g.SecondsToSamples())
    this.feedbackInputSum.A = this.inputRepeater.OUT
    this.feedbackInputSum.B = this.feedbackScaler.OUT
    this.feedbackScaler.A = this.delayer.OUT
    this.mixDryWet.B = this.delayer.OUT
    this.mixDryWet.A = this.inputRepeater.OUT
    this.delayer.IN = this.feedbackInputSum.OUT
    this.delayer.DELAY = this.delayScaler.OUT
    this.aliasInlet(this.inputRepeater.IN)
    this.aliasInlet(this.delayScaler.IN, 'delay')
    this.aliasInlet(this.feedbackScaler.B, 'feedback')
    this.aliasInlet(this.mixDryWet.DIAL, 'dryWet')
    this.aliasOutlet(this.mixDryWet.OUT)
    this.IN = input || 0
    this.DELAY = delay || 4410
    this.FEEDBACK = feedback || 0
    this.DRYWET = dryWet || 0.4
  }
}

class StereoOsc extends Patch { // patches/StereoOsc.js:9-41 — midi pitch (+ control) -> Osc -> Gain (dB) -> Pan
  constructor(p, gain, pan) {
    super()
    const sum = new g.Sum()
    this.alias(sum.A, 'p')
    this.alias(sum.B, 'pControl')
    const mToF = new g.MidiToFrequency(sum)
    const osc = new g.Osc()
    osc.F = mToF.FREQUENCY
    this.osc = osc
    const gainUnit = new g.Gain()
    gainUnit.IN = osc
    this.alias(gainUnit.GAIN)
    const panUnit = new g.Pan()
    panUnit.IN = gainUnit.OUT
    this.alias(panUnit.PAN)
    this.alias(panUnit.OUT)
    this.addUnit(sum) // sic: addUnit takes one argument, so only the Sum is registered (StereoOsc.js:32)
    this.GAIN = gain || 0
    this.PAN = pan || 0
    this.P = p || 60
    this.PCONTROL = 0
  }
  trigger() { this.osc.phase = 0 }
  get waveform() { return this.osc.waveform }
  set waveform(w) { this.osc.waveform = w }
}

class LFO extends Patch { // patches/LFO.js:6-27 — origin + amplitude * osc
  constructor(frequency, amplitude, origin, waveform) {
    super()
    const osc = new g.Osc()
    this.alias(osc.F)
    this.osc = osc
    const mult = new g.Multiply(osc.OUT)
    this.alias(mult.B, 'a')
    const location = new g.Sum(mult.OUT)
    this.alias(location.B, 'o')
    this.alias(location.OUT)
    this.addUnits(osc, mult, location)
    this.F = frequency || 1
    this.A = amplitude || 1 / 2
    this.O = origin || 1 / 2
    this.waveform = waveform || 'sine'
  }
  get waveform() { return this.osc.waveform }
  set waveform(w) { this.osc.waveform = w }
}

class MidiOsc extends Patch { // patches/MidiOsc.js:5-19
  constructor(p) {
    super()
    this.addUnits(this.mToF = new g.MidiToFrequency(), this.osc = new g.Osc(this.mToF.FREQUENCY))
    this.aliasInlet(this.mToF.MIDI, 'P')
    this.aliasOutlet(this.osc.OUT)
    this.P = p || 69
  }
}

class BandFilter extends Patch { // patches/BandFilter.js:4-23 — low-pass at fHigh into a high-pass at fLow
  constructor(input, fLow, fHigh) {
    super()
    this.addUnits(this.lowPass = new g.Filter(input, fHigh, 'LP'), this.highPass = new g.Filter(this.lowPass.OUT, fLow, 'HP'))
    this.highPass.kind = 'HP'
    this.aliasInlet(this.lowPass.IN)
    this.aliasInlet(this.lowPass.F, 'fHigh')
    this.aliasInlet(this.highPass.F, 'fLow')
    this.aliasOutlet(this.highPass.OUT)
  }
}

class MultiTapDelay extends Patch { // patches/MultiTapDelay.js:8-43 — one CircleBuffer, a pre-wiping writer, any number of taps
  constructor(nChannels, maxDelay, input) {
    super()
    if (!nChannels || !maxDelay) throw 'MultiTapDelay requires constructor args (nChannels, maxDelay[, input])'
    this.buffer = new g.CircleBuffer(nChannels, maxDelay)
    this.writer = new g.CircleBufferWriter(this.buffer)
    this.addUnits(this.writer) // (the reference also "adds" the buffer, which addUnit ignores)
    this.writer.preWipe = true
    this.aliasInlet(this.writer.IN)
    this.IN = input || 0
  }
  addTap(delay) {
    const reader = new g.CircleBufferReader(this.buffer, delay)
    reader.t = this.writer.t
    this.addUnits(reader)
    reader.chain(this.writer)
    return reader
  }
  addFeedback(delay, feedbackGain, feedbackDelay) {
    const reader = this.addTap(delay)
    const writer = new g.CircleBufferWriter(this.buffer, feedbackDelay || 0)
    writer.IN = quick.multiply(reader, feedbackGain)
    writer.t = this.writer.t
    writer.chain(this.writer)
    this.addUnits(writer)
    return reader
  }
}

class DelayMixer extends Patch { // patches/DelayMixer.js:7-38 — inputs written at their own delays, one post-wiping reader
  constructor(nChannels, maxDelay) {
    super()
    if (!nChannels || !maxDelay) throw 'DelayMixer requires constructor arguments: (nChannels, maxDelay)'
    this.buffer = new g.CircleBuffer(nChannels, maxDelay)
    this.addUnits(this.outReader = new g.CircleBufferReader(this.buffer))
    this.outReader.postWipe = true
    this.aliasOutlet(this.outReader.OUT)
  }
  addInput(input, delay, attenuation) {
    const writer = new g.CircleBufferWriter(this.buffer, delay)
    writer.t = this.outReader.t
    this.outReader.chain(writer)
    this.addUnits(writer)
    writer.IN = attenuation ? quick.multiply(input, attenuation) : input
  }
}

class TriggerGroup extends Patch { // patches/TriggerGroup.js:4-40 — a Mixer of named triggerable things
  constructor() {
    super()
    this.addUnits(this.mixer = new Mixer())
    this.triggers = {}
    this.aliasOutlet(this.mixer.defaultOutlet)
  }
  addTrigger(trigger, name) {
    if (name === undefined) for (name = 0; this.triggers[name] !== undefined;) name++
    this.triggers[name] = trigger
    this.mixer.addInput(trigger)
  }
  trigger(which) {
    if (this.triggers[which]) this.triggers[which].trigger()
    else if (this.handleUnknownTrigger) this.handleUnknownTrigger(which)
  }
}

class Synth extends Patch { // patches/Synth.js:3-27 — base class: trigger() fires the registered envelopes
  constructor() {
    super()
    this.triggerList = []
  }
  trigger(p, note) {
    if (this._trigger) this._trigger(p, note)
    for (const env of this.triggerList) env.trigger()
    return this
  }
  addEnvelope(env) {
    if (env.isOutlet) env = env.unit
    this.triggerList.push(env)
    return env
  }
}

class SpaceChannel extends Patch { // patches/SpaceChannel.js:10-44 — distance to one speaker -> attenuation (dB) + propagation delay
  constructor(speakerPosition) {
    super()
    this.addUnits(
      this.speakerPositionSubtracter = new g.Subtract(),
      this.distanceCalculator = new g.VectorMagnitude(),
      this.attenuationScaler = new g.Multiply(),
      this.delayScaler = new g.Multiply(),
      this.delayer = new g.MonoDelay(),
      this.attenuator = new g.Gain())
    this.distanceCalculator.IN = this.speakerPositionSubtracter.OUT
    this.attenuationScaler.A = this.distanceCalculator.OUT
    this.delayScaler.A = this.distanceCalculator.OUT
    this.attenuator.GAIN = this.attenuationScaler.OUT
    this.delayer.DELAY = this.delayScaler.OUT
    this.delayer.IN = this.attenuator.OUT
    this.aliasInlet(this.attenuator.IN)
    this.aliasInlet(this.speakerPositionSubtracter.A, 'placement')
    this.aliasInlet(this.speakerPositionSubtracter.B, 'speakerPosition')
    this.aliasInlet(this.attenuationScaler.B, 'decibelsPerMeter')
    this.aliasInlet(this.delayScaler.B, 'sampleDelayPerMeter')
    this.aliasOutlet(this.delayer.OUT)
    this.IN = 0
    this.PLACEMENT = [0, 0]
    this.SPEAKERPOSITION = speakerPosition || [0, 0]
    this.DECIBELSPERMETER = -3
    this.SAMPLEDELAYPERMETER = config.sampleRate / 343
  }
}

class Space extends Patch { // patches/Space.js:8-63 — one SpaceChannel per speaker, concatenated into the output channels
  constructor(input, place) {
    super()
    this.addUnits(this.signalIn = new g.Repeater(), this.placementIn = new g.Repeater(), this.outRepeater = new g.Repeater())
    this.spaceChannels = []
    this.alias(this.signalIn.IN)
    this.alias(this.placementIn.IN, 'placement')
    this.alias(this.outRepeater.OUT)
    this.IN = input || 0
    this.PLACEMENT = place || [0, 0]
    if (config.channelFormat === 'stereo') {
      this.addSpeaker([-1, 0]); this.addSpeaker([1, 0])
    } else if (config.channelFormat === 'surround') {
      for (const at of [[-1, 1], [1, 1], [0, Math.sqrt(2)], [0, 0], [-1, -1], [1, -1]]) this.addSpeaker(at)
    }
  }
  static stereo(input, place) { // (sic: on top of the two the constructor already added)
    const space = new Space(input, place)
    space.addSpeaker([-1, 0]); space.addSpeaker([1, 0])
    return space
  }
  addSpeaker(speakerPosition) {
    const chan = new SpaceChannel()
    chan.SPEAKERPOSITION = speakerPosition
    chan.PLACEMENT = this.placementIn.OUT
    chan.IN = this.signalIn
    if (this.outRepeater.IN.connected) this.outRepeater.IN = new g.ConcatChannels(this.outRepeater.IN.outlet, chan)
    else this.outRepeater.IN = chan
    this.spaceChannels.push(chan)
    this.addUnit(chan)
  }
}

class ScaryPatch extends Patch { // patches/ScaryPatch.js:6-26 — a signal placed in Space by (a multiple of) itself
  constructor(input, ammount) {
    super()
    this.addUnits(
      this.inRepeater = new g.Repeater(),
      this.ammountScaler = new g.Multiply(this.inRepeater, 1),
      this.space = new Space(this.inRepeater, this.ammountScaler))
    this.alias(this.inRepeater.IN)
    this.aliasInlet(this.ammountScaler.B, 'ammount')
    this.alias(this.space.OUT)
    this.IN = input || [0, 0]
    this.AMMOUNT = ammount || 1
  }
}

class Boop extends Patch { // patches/Boop.js:6-28 — an Osc under a triggered decay
  constructor(f, duration) {
    super()
    this.addUnits(
      this.osc = new g.Osc(f),
      this.envelope = new g.Shape('decay', duration).trigger(),
      this.mult = new g.Multiply(this.osc, this.envelope))
    /* The reference runs this hook inside the tick in which the envelope ends; it only sets flags on the patch, so here
     * it runs when the envelope's state comes back from the device (marked hostOnly for the descriptor extractor). */
    const hook = () => { this.finish() }
    Object.defineProperty(hook, 'hostOnly', { get: () => !this.onFinish && !this._finish }) // (not once the patch itself carries a finish hook)
    this.envelope.onFinish = hook
    this.aliasOutlet(this.mult.OUT)
  }
  trigger() { this.envelope.trigger() }
  stop() { this.envelope.stop() }
}

class SineBoop extends Patch { // patches/SineBoop.js:9-42
  constructor(p, duration) {
    super()
    this.addUnits(
      this.osc = new MidiOsc(p),
      this.ramp = new g.Shape('decay', duration),
      this.multiply = new g.Multiply(this.ramp, this.osc.OUT))
    this.alias(this.osc.P, 'p')
    this.alias(this.ramp.DURATION)
    this.alias(this.multiply.OUT)
    this.P = p || 60
    this.DURATION = duration || 1
  }
  static randomTwinkle(maxDuration) {
    const boop = new SineBoop()
    boop.P = 100 + Math.random() * 37
    boop.ramp.randomDecay(maxDuration || 1)
    return boop
  }
  trigger() {
    this.ramp.trigger()
    this.osc.phase = 0 // sic: a property of the MidiOsc patch, not of its Osc
    return this
  }
}

class SpaceBoop extends Patch { // patches/SpaceBoop.js:11-58
  constructor(p, waveform, d, decayForm, place) {
    super()
    this.addUnits(
      this.mToF = new g.MidiToFrequency(),
      this.osc = new g.Osc(this.mToF),
      this.durationToRate = new g.Divide(1 / config.sampleRate),
      this.envelope = new g.Shape('decay', this.durationToRate),
      this.envelopeAttenuator = new g.Multiply(this.osc, this.envelope),
      this.space = new Space(this.envelopeAttenuator.OUT))
    this.aliasInlet(this.mToF.MIDI, 'p')
    this.aliasInlet(this.space.PLACEMENT, 'placement')
    this.aliasInlet(this.durationToRate.B, 'duration')
    this.aliasOutlet(this.space.OUT)
    this.P = p || 60
    this.PLACEMENT = place || [0, 0]
    this.DURATION = d || 1
    this.waveform = waveform || 'sin'
    this.decayForm = decayForm || 'decay'
  }
  trigger(pitch, duration) {
    if (pitch) this.P = pitch
    if (duration) this.DURATION = duration
    this.osc.phase = 0
    this.envelope.trigger()
  }
  get waveform() { return this.osc.waveform }
  set waveform(w) { this.osc.waveform = w }
  get decayForm() { return this.envelope.shape }
  set decayForm(shape) { this.envelope.shape = shape }
}

class FMOsc extends Patch { // patches/FMOsc.js:8-58 — a MultiChannelOsc whose f is scaled by 2^(modulator * ammount / 12)
  constructor(f) {
    super()
    this.addUnits(this.fRepeater = new g.Repeater(), this.osc = new g.MultiChannelOsc(this.fRepeater))
    this.osc.randomPhaseFlip()
    this.aliasInlet(this.fRepeater.IN, 'f')
    this.aliasOutlet(this.osc.OUT)
    this.F = f || 440
  }
  get isFMOsc() { return true }
  addModulator(modulator, ammount) {
    const scaled = new g.Multiply(modulator, ammount || 1)
    const ratio = new g.SemitoneToRatio(scaled)
    const product = new g.Multiply(ratio, this.osc.F.outlet)
    this.addUnits(scaled, product, ratio)
    this.osc.F = product
  }
  addModulatorOsc(f, ammount) { this.addModulator(new FMOsc(f), ammount) }
  clearModulation() { this.osc.F = this.fRepeater }
  resetPhase() { this.osc.resetPhase() }
}

class ManyOsc extends Patch { // patches/ManyOsc.js:8-45
  constructor(oscs) {
    super()
    const mix = g.Sum.many(oscs)
    this.addUnits(mix, oscs)
    this.alias(mix.OUT, 'OUT')
  }
  get isManyOsc() { return true }
  static ofFrequencies(fundamental, ratios) {
    const oscs = []
    for (const i in ratios) {
      const osc = new g.Osc()
      osc.F = new g.Multiply(fundamental, ratios[i])
      oscs[i] = osc
    }
    return new ManyOsc(oscs)
  }
  static random(n, min, max) {
    n = n || 3; min = min || 20; max = max || 1000
    const freqs = []
    for (let i = 0; i < n; i++) freqs[i] = min + Math.random() * (max - min)
    return ManyOsc.ofFrequencies(1, freqs)
  }
}

class StereoDetune extends Patch { // patches/StereoDetune.js:5-29 — (ratio, 1/ratio) on the two channels
  constructor(input, ammount) {
    super()
    ammount = ammount || 0.1 * Math.random()
    const ratioL = quick.semitoneToRatio(ammount)
    const ratioR = quick.divide(1, ratioL)
    this.addUnits(this.mult = new g.Multiply(input, quick.concat(ratioL, ratioR)))
    this.alias(this.mult.A, 'in')
    this.alias(this.mult.OUT)
  }
  static random(input, maxAmmount) { return new StereoDetune(input, quick.multiply(maxAmmount || 0.1, Math.random())) }
}

class FrequencyGroup extends Patch { // patches/FrequencyGroup.js:5-42 — a fundamental and multiples of it
  constructor(f) {
    super()
    this.addUnits(this.fundamentalRepeater = new g.Repeater(f || 440, 'Hz'))
    this.fOuts = [this.fundamentalRepeater.OUT]
    this.alias(this.fundamentalRepeater.IN, 'f')
  }
  addHarmonic(ratio) {
    const harmonic = quick.mult(this.fOuts[0], ratio)
    this.fOuts.push(harmonic)
    return harmonic
  }
  addRandomHarmonic(maxNum, maxDenom) {
    const numerator = Math.ceil(Math.random() * (maxNum || 8))
    const denominator = Math.ceil(Math.random() * (maxDenom || 8))
    return this.addHarmonic(numerator / denominator)
  }
  addRandomHarmonics(n, maxNum, maxDenom) {
    const added = []
    for (let i = 0; i < (n || 1); i++) added[i] = this.addRandomHarmonic(maxNum, maxDenom)
    return added
  }
}

class AttenuationMatrix extends Patch { // patches/AttenuationMatrix.js:4-41 — random gains between a list of nodes
  constructor({ nodes, pConnection = 0.5, pMix = 0.5, maxAmmount = 1, minAmmount = 0, maxMixAmmount = 1, minMixAmmount = 0,
    allowFeedback = true }) {
    super()
    const outMixer = new Mixer()
    for (let i = 0; i < nodes.length; i++) {
      const mixer = new Mixer()
      for (let j = 0; j < nodes.length; j++) {
        if (j < i && !allowFeedback) continue
        if (Math.random() < pConnection) mixer.addAttenuated(nodes[j].OUT, Math.random() * (maxAmmount - minAmmount) + minAmmount)
      }
      if (mixer.numberOfInputs) {
        this.addUnits(mixer)
        nodes[i].IN = mixer
      }
      if (Math.random() < pMix) // sic: + minAmmount, not minMixAmmount
        outMixer.addAttenuated(nodes[i].OUT, Math.random() * (maxMixAmmount - minMixAmmount) + minAmmount)
    }
    this.aliasInlet(nodes[0].IN, 'in')
    this.aliasOutlet(outMixer.OUT, 'out')
  }
}

class APStack extends Patch { // patches/APStack.js:4-31 — random all-passes in series
  constructor(n = 4, maxDelay = 0.1, maxFeedback = 0.5) {
    super()
    const stack = g.AllPass.manyRandom(n, maxDelay, maxFeedback)
    for (let i = 1; i < stack.length; i++) stack[i].IN = stack[i - 1]
    this.addUnits(stack)
    this.aliasInlet(stack[0].IN, 'in')
    this.aliasOutlet(stack[stack.length - 1].OUT, 'out')
  }
}

class APWeb extends Patch { // patches/APWeb.js:5-23 — random all-passes behind a feed-forward AttenuationMatrix
  constructor(n = 4, maxDelay = 0.01, maxFeedback = 0.1) {
    super()
    const matrix = new AttenuationMatrix({ nodes: g.AllPass.manyRandom(n, maxDelay, maxFeedback), allowFeedback: false, pMix: 1 })
    this.addUnits(matrix)
    this.aliasInlet(matrix.IN, 'in')
    this.aliasOutlet(matrix.OUT, 'out')
  }
}

class Worm extends Patch { // patches/Worm.js:7-26 — low-passed noise: a slowly wandering control signal
  constructor(f = 1) {
    super()
    this.addUnits(this.noise = new g.Noise(), this.filter = new g.Filter(this.noise, f))
    this.aliasInlet(this.filter.F)
    this.aliasOutlet(this.filter.OUT)
    this.F = f
  }
  static random(fMax = 5) { return new Worm(quick.multiply(fMax, Math.random())) }
}

LFO.randomInRange = function (maxF, minMin, maxMax, waveform) { // patches/LFO.js:30-49
  const a = minMin + (maxMax - minMin) * Math.random()
  const b = minMin + (maxMax - minMin) * Math.random()
  const min = Math.min(a, b), max = Math.max(a, b)
  return new LFO(Math.random() * maxF, (min + max) / 2, Math.random() * (max - min), waveform)
}

module.exports = { Patch, Mixer, SimpleDelay, StereoOsc, LFO, MidiOsc, BandFilter, MultiTapDelay, DelayMixer, TriggerGroup, Synth,
  SpaceChannel, Space, ScaryPatch, Boop, SineBoop, SpaceBoop, FMOsc, ManyOsc, StereoDetune, FrequencyGroup, AttenuationMatrix,
  APStack, APWeb, Worm }
