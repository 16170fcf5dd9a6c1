'use strict'
/* renderChannelData(outlet | unit | patch, duration = 1, { TypedArray = Float32Array }) ->
 *   Promise<Array<TypedArray>> with `.sampleRate`
 * Drop-in for reference src/renderChannelData.js:5-49, executed on the MI355X through the N-API addon.
 * Misuse rejects with the reference's own strings; anything the GPU path cannot run rejects with a
 * "dusp-hip: ..." string (callers that hold the real `dusp` package can then fall back to it).
 *
 * After the render the circuit is consumed exactly as the reference leaves it: circuit.clock has
 * advanced by the ticked chunks and every unit's state fields (Osc.phase, Ramp.t/playing, Filter
 * history and coefficients, CircleBuffer node t, Timer.t, SampleRateRedux.val) hold the post-render values
 * (state write-back).
 */
const native = require('./native')
const { extract, unify, deviceRetrigger } = require('./extract')
const { makeTables } = require('./wavetables')
const { OP, UNITS } = require('./ops')

const contexts = new Map() // "sampleRate|device|slot" -> native context with that rate's wave tables

/* One context per (sample rate, device): a dusp_ctx is bound to one HIP device (include/dusp_hip.h "Threading").  device -1 = the
 * process's current device (single renders).  `slot` tells contexts of one device apart: renderMany's device list may name a GPU
 * twice (two contexts, two shards side by side on it). */
function contextFor(sampleRate, device = -1, slot = 0) {
  const key = sampleRate + '|' + device + '|' + slot
  if (!contexts.has(key)) {
    const n = native()
    const ctx = n.ctxCreate(device)
    makeTables(sampleRate).forEach((t, id) => n.tableUpload(ctx, id, t))
    contexts.set(key, ctx)
  }
  return contexts.get(key)
}

/* Contiguous, balanced split of n instances over `world` shards: the first n % world get one more (dusp_amd/shard.py
 * instance_range, SURVEY.md 8e: rank r renders [r N / G, (r + 1) N / G)). */
function instanceRange(n, rank, world) {
  const base = Math.floor(n / world), extra = n % world
  const lo = rank * base + Math.min(rank, extra)
  return [lo, lo + base + (rank < extra ? 1 : 0)]
}

/* The device list of a sharded render: undefined = every HIP device this process sees, a number n = devices 0 .. n-1, or the ids
 * themselves (one shard per entry, in order; an id may repeat). */
function deviceList(devices) {
  const n = native()
  const have = n.deviceCount()
  let list
  if (devices === undefined || devices === null) list = Array.from({ length: have }, (_, i) => i)
  else if (typeof devices === 'number') list = Array.from({ length: devices }, (_, i) => i)
  else list = Array.from(devices)
  if (!list.length) throw 'dusp-hip: renderMany: empty device list'
  for (const d of list)
    if (!Number.isInteger(d) || d < 0 || d >= have) throw 'dusp-hip: renderMany: device ' + d + ' is not one of the ' + have + ' this process sees'
  return list
}

function sampleCount(duration, sampleRate) {
  const n = Math.trunc(duration * sampleRate) // `new TypedArray(lengthInSamples)` truncates (:24,39)
  if (!(n >= 0)) throw 'dusp-hip: bad duration ' + duration
  return n
}

function writeBack(n, prog, circuit, chunkSize, nSamples) {
  circuit.units.forEach((unit, u) => {
    if (deviceRetrigger(unit, circuit.units)) { unit.t = n.stateDownload(prog, 0, u)[0]; return }
    const spec = UNITS[unit.constructor.name]
    if (!spec) return
    if (spec.op === OP.OSC) unit.phase = n.stateDownload(prog, 0, u)[0]
    else if (spec.op === OP.RAMP) { const s = n.stateDownload(prog, 0, u); unit.t = s[0]; unit.playing = s[1] !== 0 }
    else if (spec.op === OP.CB_READER || spec.op === OP.CB_WRITER) unit.t = n.stateDownload(prog, 0, u)[0]
    else if (spec.op === OP.FIXED_DELAY || spec.op === OP.COMB_FILTER || spec.op === OP.ALL_PASS || spec.op === OP.READBACK_DELAY)
      unit.tBuffer = n.stateDownload(prog, 0, u)[0]
    else if (spec.op === OP.TIMER) unit.t = n.stateDownload(prog, 0, u)[0]
    else if (spec.op === OP.SHAPE) {
      const s = n.stateDownload(prog, 0, u)
      const finishedNow = s[2] !== 0 && !unit.finished
      unit.t = s[0]; unit.playing = s[1] !== 0; unit.finished = s[2] !== 0
      if (finishedNow && unit.onFinish) unit.onFinish() // hostOnly hooks only (the extractor refuses the rest)
    } else if (spec.op === OP.AHD) {
      const s = n.stateDownload(prog, 0, u)
      unit.state = s[0]; unit.playing = s[1] !== 0; unit.t = s[2]
    }
    else if (spec.op === OP.SAMPLE_RATE_REDUX) {
      const s = n.stateDownload(prog, 0, u)
      unit.timeSinceLastUpdate = s[0]
      unit.val = Array.from(s.subarray ? s.subarray(2, 2 + s[1]) : s.slice(2, 2 + s[1]))
    } else if (spec.op === OP.MULTI_OSC) { const s = n.stateDownload(prog, 0, u); for (let c = 0; c < s[0]; c++) unit.phase[c] = s[1 + c] }
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

const RESUMABLE = 0x100 // DUSP_ENGINE_RESUMABLE (include/dusp_hip.h)

/* A circuit's wiring by unit identity, in process order: which units, and what feeds every inlet.  A host callback that
 * rewires the graph (`a then b`, constructOperation.js:45-61: the finish hook of `a` points a Repeater at `b`, whose units join
 * the running circuit) shows up as another text; the device program is then built again for the new circuit, every unit's
 * state coming from the objects (the state write-back of the segment before). */
const unitIds = new WeakMap()
let nextUnitId = 1
const idOf = (u) => { if (!unitIds.has(u)) unitIds.set(u, nextUnitId++); return unitIds.get(u) }
const RING_OPS = new Set([OP.DELAY, OP.MONO_DELAY, OP.READBACK_DELAY, OP.FIXED_DELAY, OP.COMB_FILTER, OP.ALL_PASS, OP.CB_READER, OP.CB_WRITER])
function wiringOf(circuit) {
  const units = circuit.units
  const at = new Map(units.map((u, i) => [u, i]))
  const ids = new Set(), late = [] // late: edges read a chunk late (the source ticks after its reader), as [source id, reader id]
  let text = '', rings = false
  units.forEach((u, i) => {
    ids.add(idOf(u))
    const spec = UNITS[u.constructor.name]
    if (spec && RING_OPS.has(spec.op)) rings = true
    text += idOf(u) + '('
    for (const name of Object.keys(u.inlets || {})) {
      const inlet = u.inlets[name]
      if (inlet.connected) {
        const src = inlet.outlet.unit
        text += name + ':' + idOf(src) + ','
        if (!(at.get(src) < i)) late.push([idOf(src), idOf(u)])
      }
    }
    text += ')'
  })
  return { text, ids, late, rings }
}
/* Why a rewired circuit cannot simply start on a new device program (null: it can).  What a program keeps on the device only
 * — delay lines, CircleBuffers, the previous chunk of an outlet that something reads a chunk late — would have to move into the
 * new program's layout; not built, so such circuits are refused rather than rendered from zeros. */
function cannotRebuild(before, after) {
  if (before.rings) return 'the circuit holds delay lines or CircleBuffers on the device'
  if (before.late.length) return 'an edge of the circuit is read a chunk late, its last chunk lives on the device'
  for (const [src] of after.late) if (before.ids.has(src)) return 'the new process order reads a unit of the old circuit a chunk late'
  return null
}

/* Renders a circuit piecewise, in time order.  Every call continues where the previous one stopped:
 *
 * Event-segmented rendering (SURVEY.md 8f-3).  The reference runs every event with t < clock + chunk at the start
 * of the tick at `clock` (Circuit.js:23,57-65), i.e. events take effect on chunk boundaries.  So: run the due
 * callbacks on the host objects, render up to the chunk in which the next event falls due, write the unit state
 * back, re-extract and CONTINUE the same device program (dusp_program_continue: unit state and constants come from
 * the objects, delay lines / CircleBuffers / feedback chunks stay resident on the device). */
class SegmentRenderer {
  constructor(outlet, { engine = 0, resumable = false } = {}) {
    this.outlet = outlet
    this.first = extract(outlet, { allowEvents: true })
    this.circuit = this.first.circuit
    this.chunk = this.first.chunkSize
    this.sampleRate = this.first.sampleRate
    this.retick()
    this.hasEvents = !!(this.circuit.events && this.circuit.events.length) || this.tickers.length > 0
    this.engine = resumable || this.hasEvents ? engine | RESUMABLE : engine
    this.native = native()
    this.prog = null
    this.wiring = null
    this.clock = 0
  }

  // units that act through host callbacks between chunks (Retriggerer, SporadicRetriggerer; host-computed signals: Noise): ticked here, firing = segment boundary
  retick() {
    this.tickers = this.circuit.units.filter((u) => ((UNITS[u.constructor.name] && UNITS[u.constructor.name].hostTick) || u.isHostSignal) &&
      !deviceRetrigger(u, this.circuit.units)) // (a Retriggerer of a Shape / AHD runs on the device)
    for (const u of this.tickers)
      if (!u.hostTick) throw 'dusp-hip: ' + u.label + ' needs host-side ticking, which only this package\'s own unit classes provide'
  }

  /* the next nSamples samples (a whole number of chunks, except in the last call of a render) ->
   * { pcm: Float32Array [channel][nSamples], nChannels } */
  async next(nSamples) {
    const n = this.native, chunk = this.chunk
    const start = this.clock
    const end = start + Math.ceil(nSamples / chunk) * chunk
    const pieces = []
    while (this.clock < end) {
      let next = end
      if (this.circuit.events && this.circuit.events.length) {
        this.circuit.runEvents(this.clock + chunk)
        if (this.circuit.events.length) {
          const due = Math.floor(this.circuit.events[0].t / chunk) * chunk
          next = Math.min(end, Math.max(this.clock + chunk, due))
        }
      }
      if (this.tickers.length) { // this chunk's host ticks (after the events, as in Circuit.tick), then run up to the next firing
        for (const u of this.tickers) u.hostTick(chunk)
        // look ahead chunk by chunk, every ticker in tick order (the random ones draw their numbers in the reference's
        // order), never past the next scheduled event; the first chunk in which anything fires starts the next segment
        const room = (next - this.clock) / chunk - 1
        let quiet = 0
        for (const u of this.tickers) u.peekBegin()
        while (quiet < room) {
          let fires = false
          for (const u of this.tickers) fires = u.peekNext(chunk) || fires
          if (fires) break
          quiet++
        }
        for (const u of this.tickers) u.skipQuiet(chunk, quiet)
        next = this.clock + (1 + quiet) * chunk
      }
      const ex = !this.prog && !this.hasEvents ? this.first : extract(this.outlet, { allowEvents: true, allowClock: true })
      const wiring = wiringOf(ex.circuit)
      if (this.prog && wiring.text !== this.wiring.text) { // a callback rewired the circuit (`then`: unDusp.js): another device program from here on
        const why = cannotRebuild(this.wiring, wiring)
        if (why) throw 'dusp-hip: the circuit was rewired during the render (' + why + '): not supported on the GPU path'
        n.programDestroy(this.prog)
        this.prog = null
        this.circuit = ex.circuit
        const before = this.tickers
        this.retick()
        if (this.tickers.length !== before.length || this.tickers.some((u, i) => u !== before[i]))
          throw 'dusp-hip: the circuit was rewired during the render (units that tick on the host joined or left): not supported on the GPU path'
      }
      this.wiring = wiring
      if (!this.prog) this.prog = n.programBuild(contextFor(ex.sampleRate), ex.words, this.engine)
      else n.programContinue(this.prog, ex.words)
      const len = Math.min(next, start + nSamples) - this.clock // the last segment may end inside a chunk
      let inputs = null // the host-computed signals of this segment: [source][len]
      if (ex.sources.length) {
        inputs = new Float32Array(ex.sources.length * len)
        ex.sources.forEach((u, k) => u.takeSegment(len, inputs, k * len))
      }
      const pcm = await n.render(this.prog, 1, len, null, false, inputs) // Float32Array [channel][len]
      writeBack(n, this.prog, this.circuit, chunk, len) // advances circuit.clock to `next`
      pieces.push({ pcm, len, nChannels: n.programInfo(this.prog).nOutChannels })
      this.clock = next
    }
    const nChannels = Math.max(...pieces.map((p) => p.nChannels))
    if (pieces.length === 1) return { pcm: pieces[0].pcm, nChannels }
    const pcm = new Float32Array(nChannels * nSamples) // late channels start as zeros (renderChannelData.js:38-39)
    let at = 0
    for (const p of pieces) {
      for (let c = 0; c < p.nChannels; c++) pcm.set(p.pcm.subarray(c * p.len, (c + 1) * p.len), c * nSamples + at)
      at += p.len
    }
    return { pcm, nChannels }
  }

  close() {
    if (this.prog) this.native.programDestroy(this.prog)
    this.prog = null
  }
}

async function renderChannelData(outlet, duration = 1, { TypedArray = Float32Array, engine = 0 } = {}) {
  const renderer = new SegmentRenderer(outlet, { engine })
  try {
    const nSamples = sampleCount(duration, renderer.sampleRate)
    const channelData = []
    channelData.sampleRate = renderer.sampleRate
    if (nSamples === 0) return channelData
    const { pcm, nChannels } = await renderer.next(nSamples)
    for (let c = 0; c < nChannels; c++) {
      const channel = pcm.subarray(c * nSamples, (c + 1) * nSamples) // Float32Array: handed over without a copy
      channelData.push(TypedArray === Float32Array ? channel : TypedArray.from(channel))
    }
    return channelData
  } finally {
    renderer.close()
  }
}

/* N structurally identical circuits (voices, a parameter sweep) as ONE GPU program per device:
 * resolves to result[instance][channel] = Float32Array(duration * sampleRate).
 *
 * The instances share nothing, so they shard over the GPUs of the node with no exchange at all (SURVEY.md 8e): `devices` (default:
 * every HIP device the process sees; a count; or a list of ids) names one shard per entry, shard r renders the contiguous instance
 * range instanceRange(N, r, shards) on a context of ITS device — the same program text, its rows of the parameter table — and all
 * shards are in flight at once (one async render each on the libuv pool; the addon serialises calls per context, not across them).
 * The results come back in instance order whatever order the devices finish in.  A render rejects as a whole with the first
 * failing shard's string. */
async function renderMany(outlets, duration = 1, { engine = 0, devices } = {}) {
  const extractions = outlets.map((o) => extract(o))
  // one launch for all circuits: nothing ticks on the host in between, so units that need that are refused, not ignored
  for (const ex of extractions)
    for (const u of ex.circuit.units)
      if ((u.isHostSignal || (UNITS[u.constructor.name] && UNITS[u.constructor.name].hostTick)) && !deviceRetrigger(u, ex.circuit.units))
        throw 'dusp-hip: renderMany does not take circuits with host-ticked units (' + u.label + '): render them one by one'
  const uni = unify(extractions)
  const nSamples = sampleCount(duration, uni.sampleRate)
  if (nSamples === 0) return outlets.map(() => [])
  const n = native()
  const list = deviceList(devices)
  const seen = new Map() // device id -> contexts of it handed out so far
  const shards = []
  for (let r = 0; r < list.length; r++) {
    const [lo, hi] = instanceRange(uni.nInstances, r, list.length)
    if (hi === lo) continue
    const slot = seen.get(list[r]) || 0
    seen.set(list[r], slot + 1)
    shards.push({ lo, hi, device: list[r], slot, prog: null })
  }
  try {
    for (const sh of shards) sh.prog = n.programBuild(contextFor(uni.sampleRate, list.length === 1 && devices === undefined ? -1 : sh.device, sh.slot), uni.words, engine)
    const jobs = shards.map((sh) => {
      let params = null
      if (uni.nParams) { // slot-major [nParams][nInstances]: this shard's columns of every row
        const count = sh.hi - sh.lo
        params = new Float32Array(uni.nParams * count)
        for (let p = 0; p < uni.nParams; p++) params.set(uni.params.subarray(p * uni.nInstances + sh.lo, p * uni.nInstances + sh.hi), p * count)
      }
      return n.render(sh.prog, sh.hi - sh.lo, nSamples, params)
    })
    const pcms = await Promise.all(jobs.map((j) => j.then((v) => ({ ok: v }), (e) => ({ err: e })))) // (every shard ends before anything is torn down)
    const failed = pcms.find((p) => p.err !== undefined)
    if (failed) throw failed.err
    const result = new Array(uni.nInstances)
    shards.forEach((sh, k) => {
      const nCh = n.programInfo(sh.prog).nOutChannels, pcm = pcms[k].ok
      for (let i = sh.lo; i < sh.hi; i++) {
        const chans = []
        for (let c = 0; c < nCh; c++) {
          const at = ((i - sh.lo) * nCh + c) * nSamples
          chans.push(pcm.subarray(at, at + nSamples))
        }
        chans.sampleRate = uni.sampleRate
        result[i] = chans
      }
    })
    return result
  } finally {
    for (const sh of shards) if (sh.prog) n.programDestroy(sh.prog)
  }
}

/* A flat descriptor (what lib/extract.js produces — from this package's graph classes or from the reference's own objects,
 * patches included: their units reach the extractor as they are) rendered as it stands: no host objects, hence no events,
 * no host-ticked units, no state write-back.  Resolves to channelData like renderChannelData. */
async function renderDescriptor(words, nSamples, { engine = 0 } = {}) {
  const sampleRate = words[2]
  const n = native()
  const prog = n.programBuild(contextFor(sampleRate), words, engine)
  try {
    const info = n.programInfo(prog)
    const pcm = await n.render(prog, 1, nSamples, null)
    const channelData = []
    for (let c = 0; c < info.nOutChannels; c++) channelData.push(pcm.subarray(c * nSamples, (c + 1) * nSamples))
    channelData.sampleRate = sampleRate
    return channelData
  } finally {
    n.programDestroy(prog)
  }
}

module.exports = renderChannelData
module.exports.renderChannelData = renderChannelData
module.exports.renderDescriptor = renderDescriptor
module.exports.renderMany = renderMany
module.exports.instanceRange = instanceRange
module.exports.deviceCount = () => native().deviceCount()
module.exports.SegmentRenderer = SegmentRenderer
