'use strict'
/* renderChannelData(outlet | unit | patch, duration = 1, { TypedArray = Float32Array }) ->
 *   Promise<Array<TypedArray>> with `.sampleRate`
 * Drop-in for reference src/renderChannelData.js:5-49, executed on the MI355X through the N-API addon.
 * Misuse rejects with the reference's own strings; anything the GPU path cannot run rejects with a
 * "dusp-hip: ..." string (callers that hold the real `dusp` package can then fall back to it).
 *
 * After the render the circuit is consumed exactly as the reference leaves it: circuit.clock has
 * advanced by the ticked chunks and every unit's state fields (Osc.phase, Ramp.t/playing, Filter
 * history and coefficients, CircleBuffer node t) hold the post-render values (state write-back).
 */
const native = require('./native')
const { extract, unify } = require('./extract')
const { makeTables } = require('./wavetables')
const { OP, UNITS } = require('./ops')

const contexts = new Map() // sampleRate -> native context with that rate's wave tables

function contextFor(sampleRate) {
  if (!contexts.has(sampleRate)) {
    const n = native()
    const ctx = n.ctxCreate(-1)
    makeTables(sampleRate).forEach((t, id) => n.tableUpload(ctx, id, t))
    contexts.set(sampleRate, ctx)
  }
  return contexts.get(sampleRate)
}

function sampleCount(duration, sampleRate) {
  const n = Math.trunc(duration * sampleRate) // `new TypedArray(lengthInSamples)` truncates (:24,39)
  if (!(n >= 0)) throw 'dusp-hip: bad duration ' + duration
  return n
}

function writeBack(n, prog, circuit, chunkSize, nSamples) {
  circuit.units.forEach((unit, u) => {
    const spec = UNITS[unit.constructor.name]
    if (!spec) return
    if (spec.op === OP.OSC) unit.phase = n.stateDownload(prog, 0, u)[0]
    else if (spec.op === OP.RAMP) { const s = n.stateDownload(prog, 0, u); unit.t = s[0]; unit.playing = s[1] !== 0 }
    else if (spec.op === OP.CB_READER || spec.op === OP.CB_WRITER) unit.t = n.stateDownload(prog, 0, u)[0]
    else if (spec.op === OP.FILTER) {
      const s = n.stateDownload(prog, 0, u)
      if (s[0]) unit.lastF = s[1]
      unit.a0 = s[2]; unit.a1 = s[3]; unit.a2 = s[4]; unit.b1 = s[5]; unit.b2 = s[6]
      for (let c = 0; c < s[7]; c++) {
        unit.x1[c] = s[8 + 4 * c]; unit.x2[c] = s[9 + 4 * c]; unit.y1[c] = s[10 + 4 * c]; unit.y2[c] = s[11 + 4 * c]
      }
    }
  })
  circuit.clock += Math.ceil(nSamples / chunkSize) * chunkSize
}

async function renderChannelData(outlet, duration = 1, { TypedArray = Float32Array, engine = 0, stateWriteBack = true } = {}) {
  const ex = extract(outlet)
  const nSamples = sampleCount(duration, ex.sampleRate)
  const channelData = []
  channelData.sampleRate = ex.sampleRate
  if (nSamples === 0) return channelData
  const n = native()
  const prog = n.programBuild(contextFor(ex.sampleRate), ex.words, engine)
  try {
    const info = n.programInfo(prog)
    const pcm = await n.render(prog, 1, nSamples, null) // Float32Array [channel][sample]
    for (let c = 0; c < info.nOutChannels; c++) {
      const chan = pcm.subarray(c * nSamples, (c + 1) * nSamples)
      channelData.push(TypedArray === Float32Array ? chan : TypedArray.from(chan))
    }
    if (stateWriteBack) writeBack(n, prog, ex.circuit, ex.chunkSize, nSamples)
  } finally {
    n.programDestroy(prog)
  }
  return channelData
}

/* N structurally identical circuits (voices, a parameter sweep) as ONE GPU program:
 * resolves to result[instance][channel] = Float32Array(duration * sampleRate). */
async function renderMany(outlets, duration = 1, { engine = 0 } = {}) {
  const uni = unify(outlets.map(extract))
  const nSamples = sampleCount(duration, uni.sampleRate)
  if (nSamples === 0) return outlets.map(() => [])
  const n = native()
  const prog = n.programBuild(contextFor(uni.sampleRate), uni.words, engine)
  try {
    const info = n.programInfo(prog)
    const pcm = await n.render(prog, uni.nInstances, nSamples, uni.nParams ? uni.params : null)
    const result = []
    for (let i = 0; i < uni.nInstances; i++) {
      const chans = []
      for (let c = 0; c < info.nOutChannels; c++) {
        const at = (i * info.nOutChannels + c) * nSamples
        chans.push(pcm.subarray(at, at + nSamples))
      }
      chans.sampleRate = uni.sampleRate
      result.push(chans)
    }
    return result
  } finally {
    n.programDestroy(prog)
  }
}

module.exports = renderChannelData
module.exports.renderChannelData = renderChannelData
module.exports.renderMany = renderMany
