'use strict'
/* Graph extractor: live Dusp object graph -> flat descriptor words.
 *
 * Works on duck-typed objects, so it accepts graphs built with the real `dusp`
 * package as well as with this package's own graph classes (lib/graph.js).
 * What it reads is the object contract listed in SURVEY.md §8b:
 *   outlet.unit / .isOutlet / .name / .chunkSize / .sampleRate   (reference src/Piglet.js:9-22)
 *   unit.circuit / .getOrBuildCircuit() / .inlets / .tickInterval (reference src/Unit.js:10-28,288-293)
 *   inlet.connected / .outlet / .signalChunk / .constant          (reference src/Inlet.js:7-8,47-55,80)
 *   circuit.units (already in process order) / .clock / .events / .promises
 *                                                                 (reference src/Circuit.js:4-17,125-131)
 * plus the per-unit state fields named in SURVEY.md §8a.
 *
 * The unit order is NEVER re-derived here: it is whatever the circuit object
 * holds, i.e. the reference's own computeProcessIndex + stable sort result.
 */

const { MAGIC, VERSION, HEADER_WORDS, OP, INLET, WAVEFORMS, FILTER_KINDS, SHAPES, UNITS } = require('./ops')

function toOutlet(x) {
  // same checks and messages as reference src/renderChannelData.js:12-17
  if (!x) throw 'renderAudioBuffer expects an outlet'
  if (x.isUnit || x.isPatch) x = x.defaultOutlet
  if (!x || !x.isOutlet) throw 'renderAudioBuffer expects an outlet'
  return x
}

function inletConstants(inlet) {
  // The chunk is the truth: setConstant() writes the f32-rounded value into
  // every sample of every channel (reference src/Inlet.js:76-93).
  const chunk = inlet.signalChunk && inlet.signalChunk.channelData
  if (chunk && chunk.length) {
    const vals = []
    for (let c = 0; c < chunk.length; c++) vals.push(chunk[c][0])
    return vals
  }
  const k = inlet.constant
  return Array.isArray(k) ? k.map(Math.fround) : [Math.fround(k || 0)]
}

function dataOutletName(unit) { // "out" everywhere except MidiToFrequency's "frequency" (MidiToFrequency.js:6)
  const spec = UNITS[unit.constructor && unit.constructor.name]
  return (spec && spec.outlet) || 'out'
}

/* A Retriggerer (Retriggerer.js:3-43) runs on the device when all it does can be done there: a constant rate, and a target
 * whose trigger() the device knows — a Shape, an AHD or a Ramp of the same circuit.  Otherwise the host ticks it between segments. */
function deviceRetrigger(unit, units) {
  if (!unit || !unit.constructor || unit.constructor.name !== 'Retriggerer') return false
  const target = unit._target
  if (!target || !target.constructor || !['Shape', 'AHD', 'Ramp'].includes(target.constructor.name)) return false
  if (units && !units.includes(target)) return false
  const rate = unit.inlets && unit.inlets.rate
  return !!rate && !rate.connected
}

function extract(target, opts = {}) {
  const outlet = toOutlet(target)
  const outUnit = outlet.unit
  const circuit = outUnit.circuit || outUnit.getOrBuildCircuit()
  if (circuit.events && circuit.events.length && !opts.allowEvents)
    throw 'dusp-hip: this circuit has scheduled events; extract() flattens a single segment (renderChannelData segments them)'
  if (circuit.promises && circuit.promises.length)
    throw 'dusp-hip: circuits with pending promises are not supported on the GPU path'
  if (circuit.clock && !opts.allowClock)
    throw 'dusp-hip: circuit has already been ticked (clock=' + circuit.clock + '); render a fresh circuit'

  const units = circuit.units
  const chunkSize = outlet.chunkSize
  const sampleRate = outlet.sampleRate
  const rings = [] // CircleBuffer objects, by identity
  const ringId = (buf) => {
    let i = rings.indexOf(buf)
    if (i < 0) { i = rings.length; rings.push(buf) }
    return i
  }

  const body = []
  const constSites = [] // {kindPos, valPos, n} relative to body start
  const labels = []
  const sources = [] // units whose signal the host computes, in stream order

  for (const unit of units) {
    const kind = unit.constructor && unit.constructor.name
    const spec = deviceRetrigger(unit, units) ? UNITS.DeviceRetriggerer : UNITS[kind] || (unit.isHostSignal ? UNITS.HostSignal : undefined)
    if (!spec) throw 'dusp-hip: unit type not supported on the GPU path: ' + kind + ' (' + unit.label + ')'
    if (unit.tickInterval !== chunkSize)
      throw 'dusp-hip: unit ' + unit.label + ' has tickInterval ' + unit.tickInterval + ' != chunk size ' + chunkSize
    labels.push(unit.label || kind)

    let attrs = [], state = []
    switch (spec.op) {
      case OP.OSC: {
        const w = WAVEFORMS[unit.waveform]
        if (w === undefined) throw "waveform doesn't exist: " + unit.waveform
        attrs = [w]; state = [unit.phase]
        break
      }
      case OP.RAMP:
        attrs = [unit.duration, unit.y0, unit.y1]; state = [unit.t, unit.playing ? 1 : 0]
        break
      case OP.FILTER: {
        const k = FILTER_KINDS[unit.kind]
        if (k === undefined) throw 'dusp-hip: filter kind not supported on the GPU path: ' + unit.kind
        attrs = [k]
        const nch = Math.max(unit.x1.length, unit.x2.length, unit.y1.length, unit.y2.length)
        const has = unit.lastF !== undefined
        state = [has ? 1 : 0, has ? unit.lastF : 0, unit.a0, unit.a1, unit.a2, unit.b1, unit.b2, nch]
        for (let c = 0; c < nch; c++)
          state.push(unit.x1[c] || 0, unit.x2[c] || 0, unit.y1[c] || 0, unit.y2[c] || 0)
        break
      }
      case OP.DELAY:
        attrs = [unit.maxDelay]
        break
      case OP.CB_READER:
        attrs = [ringId(unit.buffer), unit.postWipe ? 1 : 0]; state = [unit.t]
        break
      case OP.CB_WRITER:
        attrs = [ringId(unit.buffer), unit.preWipe ? 1 : 0]; state = [unit.t]
        break
      case OP.FIXED_MULTIPLY:
        attrs = [Number(unit.sf)] // a plain JS number on the unit, not an inlet (FixedMultiply.js:8,20)
        break
      case OP.FIXED_DELAY: case OP.COMB_FILTER: case OP.ALL_PASS: // ring = round(delayTime * sr) samples (FixedDelay.js:24-26)
        attrs = [unit.delayTimeInSamples]; state = [unit.tBuffer]
        break
      case OP.MONO_DELAY:
        attrs = [unit.maxDelay]
        break
      case OP.READBACK_DELAY:
        attrs = [unit.bufferLength]; state = [unit.tBuffer]
        break
      case OP.MULTI_OSC: {
        const w = WAVEFORMS[unit.waveform]
        if (w === undefined) throw "waveform doesn't exist: " + unit.waveform
        attrs = [w]
        state = [unit.phase.length].concat(Array.from(unit.phase, (p) => p || 0))
        break
      }
      case OP.PAN:
        attrs = [Number(unit.compensationDB)] // a plain property (Pan.js:12)
        break
      case OP.TIMER:
        attrs = [unit.samplePeriod]; state = [unit.t]
        break
      case OP.RETRIGGER:
        attrs = [units.indexOf(unit._target)]; state = [unit.t]
        break
      case OP.INPUT: // Noise.js:16-27 reads `this.f[0][t]`: a connected f would have to come back from the device
        if (unit.inlets.f && unit.inlets.f.connected)
          throw 'dusp-hip: Noise with a signal-rate `f` is not supported on the GPU path (' + unit.label + ')'
        attrs = [sources.length]
        sources.push(unit)
        break
      case OP.HOST_ONLY:
        if (unit.rateConstant) unit.rateConstant() // throws for a connected rate
        else if (unit.inlets.rate && unit.inlets.rate.connected)
          throw 'dusp-hip: Retriggerer with a signal-rate `rate` is not supported on the GPU path (' + unit.label + ')'
        break
      case OP.SHAPE: { // edges: the string "shape" (= the table's end value) or a plain number (Shape/index.js:35-49)
        const table = SHAPES[unit.shape]
        if (table === undefined) throw 'dusp-hip: shape function not supported on the GPU path: ' + unit.shape
        const edge = (e) => {
          if (e === 'shape') return [1, 0]
          if (typeof e === 'number') return [0, e]
          throw 'dusp-hip: Shape edge must be "shape" or a number (' + unit.label + ')'
        }
        // a hook marked hostOnly (lib/patches.js Boop) only keeps host-side books: write-back runs it once the state shows `finished`
        if (unit._finish || (unit.onFinish && !unit.onFinish.hostOnly && !opts.allowFinishHooks))
          throw 'dusp-hip: finish callbacks cannot run on the GPU path (' + unit.label + ')'
        attrs = [table, ...edge(unit.leftEdge), ...edge(unit.rightEdge)]
        state = [unit.t, unit.playing ? 1 : 0, unit.finished ? 1 : 0]
        break
      }
      case OP.AHD: // samplePeriod is a module constant of the reference (AHD.js:4)
        attrs = [1 / sampleRate]; state = [unit.state, unit.playing ? 1 : 0, unit.t]
        break
      case OP.SAMPLE_RATE_REDUX: // val = the held sample per channel, `[0]` before the first update (SampleRateRedux.js:9-10)
        state = [unit.timeSinceLastUpdate, unit.val.length].concat(Array.from(unit.val))
        break
    }

    body.push(spec.op, spec.inlets.length, attrs.length, state.length)
    for (const name of spec.inlets) {
      const inlet = unit.inlets[name]
      if (!inlet) throw 'dusp-hip: unit ' + unit.label + ' has no inlet ' + name
      if (inlet.connected) {
        const src = units.indexOf(inlet.outlet.unit)
        if (src < 0) throw 'dusp-hip: inlet ' + inlet.label + ' is fed from outside the circuit'
        if (inlet.outlet.name !== dataOutletName(inlet.outlet.unit))
          throw 'dusp-hip: only the data outlet ("out") of a unit can feed an inlet on the GPU path (' + inlet.outlet.label + ')'
        body.push(INLET.CONNECT, 3, src, 0, 0)
      } else {
        const vals = inletConstants(inlet)
        constSites.push({ kindPos: body.length, valPos: body.length + 2, n: vals.length })
        body.push(INLET.CONST, vals.length, ...vals)
      }
    }
    body.push(...attrs, ...state)
  }

  const outIndex = units.indexOf(outUnit)
  if (outlet.name !== dataOutletName(outUnit)) throw 'dusp-hip: only "out" outlets can be rendered on the GPU path'

  const ringWords = []
  for (const r of rings) ringWords.push(r.numberOfChannels, r.lengthInSamples)

  const head = [MAGIC, VERSION, sampleRate, chunkSize, units.length, rings.length,
    0 /* n_params */, outIndex, 0 /* out outlet */, circuit.clock || 0, 0, 0]
  const base = HEADER_WORDS + ringWords.length
  const words = Float64Array.from(head.concat(ringWords, body))
  for (const s of constSites) { s.kindPos += base; s.valPos += base }
  return { words, constSites, labels, sampleRate, chunkSize, circuit, sources }
}

/* Turn N structurally identical circuits ("voices" / a parameter sweep) into ONE
 * program plus a per-instance parameter table: every unconnected inlet whose
 * constant differs between instances becomes a PARAM inlet.  Returns
 * { words, params: Float32Array (slot-major [n_params][n_instances]), nParams, nInstances }.
 */
function unify(extractions) {
  const n = extractions.length
  if (!n) throw 'dusp-hip: no instances'
  const first = extractions[0]
  const words = Float64Array.from(first.words)
  const isConstVal = new Uint8Array(words.length)
  for (const s of first.constSites) for (let k = 0; k < s.n; k++) isConstVal[s.valPos + k] = 1
  for (let i = 1; i < n; i++) {
    const w = extractions[i].words
    if (w.length !== words.length) throw 'dusp-hip: instance ' + i + ' differs in structure from instance 0'
    for (let p = 0; p < w.length; p++)
      if (!isConstVal[p] && !Object.is(w[p], words[p]) && !(w[p] === words[p]))
        throw 'dusp-hip: instance ' + i + ' differs from instance 0 outside inlet constants (word ' + p + ')'
  }
  const columns = []
  for (const s of first.constSites) {
    let varies = false
    for (let i = 1; i < n && !varies; i++)
      for (let k = 0; k < s.n; k++)
        if (!Object.is(extractions[i].words[s.valPos + k], words[s.valPos + k])) { varies = true; break }
    if (!varies) continue
    words[s.kindPos] = INLET.PARAM
    for (let k = 0; k < s.n; k++) {
      const col = new Float32Array(n)
      for (let i = 0; i < n; i++) col[i] = extractions[i].words[s.valPos + k]
      words[s.valPos + k] = columns.length
      columns.push(col)
    }
  }
  words[6] = columns.length
  const params = new Float32Array(columns.length * n)
  columns.forEach((col, j) => params.set(col, j * n))
  return { words, params, nParams: columns.length, nInstances: n, labels: first.labels,
    sampleRate: first.sampleRate, chunkSize: first.chunkSize }
}

module.exports = {
  deviceRetrigger, extract, unify, toOutlet }
