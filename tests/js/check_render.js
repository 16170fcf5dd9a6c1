'use strict'
/* GPU parity through the JS surface: renderChannelData(unit, duration) via the N-API addon against
 * the reference's golden PCM.  usage: node check_render.js --sampleRate=48000 */
const fs = require('fs')
const path = require('path')
const lib = require('../../dusp_amd/js')
const SR = lib.config.sampleRate
const GOLDEN = path.join(__dirname, '..', 'golden')
const goldenCases = require('./cases')
const cases = goldenCases(lib, SR)
const USES_DEVICE_TAN = /^(loop_|filter_|map_gain|map_db_semitone|map_pow|map_fm_semitone|rest_pan|rest_midi|ev_filter|ev_loop|rt_|str_|grow_feedback_filter)/ // device tan() / pow()

async function main() {
  const report = { sampleRate: SR, checked: 0, exact: 0, withinTol: 0, failed: [], fromDescriptor: 0 }
  // Vectors whose graphs the reference built with its patches (src/patches/*: host-side builders this package does not mirror):
  // the descriptor extracted from the reference's live objects is the input, rendered as it stands through the addon.
  const built = new Set(cases.map((c) => c.name))
  const index = JSON.parse(fs.readFileSync(path.join(GOLDEN, SR === 48000 ? 'index.json' : 'index_sr' + SR + '.json')))
  const todo = cases.concat(index.filter((name) => !built.has(name)).map((name) => ({ name, descriptor: true })))
  for (const c of todo) {
    const metaFile = path.join(GOLDEN, c.name + '.json')
    if (!fs.existsSync(metaFile)) continue
    const meta = JSON.parse(fs.readFileSync(metaFile))
    const buf = fs.readFileSync(path.join(GOLDEN, c.name + '.pcm.f32'))
    const want = new Float32Array(buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.byteLength))
    let cd
    if (c.descriptor) {
      const db = fs.readFileSync(path.join(GOLDEN, c.name + '.desc.f64'))
      cd = await lib.renderDescriptor(new Float64Array(db.buffer.slice(db.byteOffset, db.byteOffset + db.byteLength)), meta.n_samples)
      report.fromDescriptor++
    } else cd = await goldenCases.withSeed(c.seed, () => lib.renderChannelData(c.build(), c.duration))
    let ok = cd.length === meta.n_channels && cd.sampleRate === SR && cd[0].length === meta.n_samples
    let at = 0, maxErr = 0, scale = 0, exact = true
    for (let ch = 0; ok && ch < meta.n_channels; ch++)
      for (const [a, len] of meta.windows)
        for (let t = 0; t < len; t++, at++) {
          const g = cd[ch][a + t], w = want[at]
          if (g !== w) exact = false
          maxErr = Math.max(maxErr, Math.abs(g - w)); scale = Math.max(scale, Math.abs(w))
        }
    report.checked++
    if (ok && exact) report.exact++
    else if (ok && USES_DEVICE_TAN.test(c.name) && maxErr <= 1e-5 * scale) report.withinTol++
    else report.failed.push({ name: c.name, maxErr, ok })
  }
  // state write-back into the JS objects, as the reference leaves them after a render
  const osc = new lib.Osc(440.5)
  await lib.renderChannelData(osc, 1000 / SR)
  report.oscPhaseAfter = osc.phase           // 1024 ticks of 440.5 -> (1024 * 440.5) mod sr
  report.oscPhaseExpected = (1024 * 440.5) % SR
  report.clockAfter = osc.circuit.clock
  const again = await lib.renderChannelData(osc, 0.01).then(() => 'resolved', (e) => e)
  report.secondRenderRejects = again
  // batched voices
  const voices = [1, 2, 3, 4, 5].map((k) => new lib.Multiply(new lib.Osc(110 * k + 0.5), new lib.Ramp(2000, 1, 0).trigger()))
  const many = await lib.renderMany(voices, 2048 / SR)
  const solo = await lib.renderChannelData(new lib.Multiply(new lib.Osc(110 * 3 + 0.5), new lib.Ramp(2000, 1, 0).trigger()), 2048 / SR)
  report.manyMatchesSolo = many.length === 5 && many[2][0].every((v, i) => v === solo[0][i])
  // ... but a Retriggerer of a Shape / AHD runs on the device, so a batch of retriggered voices is one launch too
  const beat = (rate, f) => { const e = new lib.Shape('decay', 0.01).trigger(); new lib.Retriggerer(e, rate); return new lib.Multiply(new lib.Osc(f), e) }
  const beats = [[50, 330.5], [20, 220], [33.25, 110], [400, 55.5]]
  const manyBeats = await lib.renderMany(beats.map(([r, f]) => beat(r, f)), 0.2)
  report.manyRetriggered = true
  for (let k = 0; k < beats.length; k++) {
    const one = await lib.renderChannelData(beat(...beats[k]), 0.2)
    report.manyRetriggered = report.manyRetriggered && manyBeats[k][0].length === one[0].length && manyBeats[k][0].every((v, i) => v === one[0][i])
  }
  // renderMany over the node's GPUs: one shard per entry of `devices`, each on a context of its own, all in flight at once; the
  // results in instance order.  On a one-GPU box the same device listed two / three times exercises the split, the parameter rows,
  // the ordering and the concurrency (two contexts side by side on one card): bit for bit the one-context render.
  const sweep = (k) => new lib.Multiply(new lib.Multiply(new lib.Osc(55 + 13.25 * k), new lib.Ramp(3000, 1, 0.25).trigger()), 0.5 + k / 32) // (two parameter rows: f and the gain)
  const eleven = () => Array.from({ length: 11 }, (_, k) => sweep(k))
  const nSweep = 256 * 40 + 17
  const oneCtx = await lib.renderMany(eleven(), nSweep / SR, { devices: [0] })
  const same = (a, b) => a.length === b.length && a.every((chans, i) => chans.length === b[i].length && chans.every((ch, c) => ch.length === b[i][c].length && ch.every((v, t) => v === b[i][c][t])))
  report.devices = lib.deviceCount()
  report.shardedTwice = same(oneCtx, await lib.renderMany(eleven(), nSweep / SR, { devices: [0, 0] }))
  report.shardedThrice = same(oneCtx, await lib.renderMany(eleven(), nSweep / SR, { devices: [0, 0, 0] }))
  report.shardedDefault = same(oneCtx, await lib.renderMany(eleven(), nSweep / SR)) // every visible device
  report.shardedMoreDevicesThanVoices = same(oneCtx.slice(0, 2), await lib.renderMany(eleven().slice(0, 2), nSweep / SR, { devices: [0, 0, 0, 0, 0] }))
  const loopVoice = (k) => { const sum = new lib.Sum(new lib.Osc(110 + k / 4), 0); const f = new lib.Filter(new lib.Delay(sum, 480, 4096), 2000); sum.B = new lib.Multiply(f, 0.5); return f }
  const loops = () => Array.from({ length: 7 }, (_, k) => loopVoice(k))
  report.shardedLoops = same(await lib.renderMany(loops(), 0.25, { devices: [0] }), await lib.renderMany(loops(), 0.25, { devices: [0, 0, 0] }))
  report.badDevice = await lib.renderMany(eleven(), 0.01, { devices: [0, 99] }).then(() => 'resolved', (e) => String(e))
  report.instanceRange = [[10, 0, 3], [10, 1, 3], [10, 2, 3], [2, 4, 5], [65536, 7, 8]].map((a) => lib.renderChannelData.instanceRange(...a))
  // renderMany runs one launch for all voices: a circuit whose unit needs host ticks in between is refused, not mis-rendered
  const ticking = () => { const e = new lib.Shape('decay', 0.01).trigger(); new lib.SporadicRetriggerer(e, 50); return new lib.Multiply(new lib.Osc(200), e) } // (random: ticked on the host)
  report.manyRefusesHostTicked = await lib.renderMany([ticking(), ticking()], 0.01).then(() => 'resolved', (e) => String(e))
  // unsupported graphs reject with a string
  class Crackle extends lib.Unit { constructor() { super(); this.addOutlet('out') } } // a unit kind this package does not know
  report.unsupported = await lib.renderChannelData(new lib.Multiply(new Crackle(), 0.5), 0.01).then(() => 'resolved', (e) => e)
  // `then` rewires the circuit from a finish hook: a new device program — unless the old circuit holds device-only memory (refused, not rendered from zeros)
  report.thenWithDeviceMemory = await lib.renderChannelData(lib.unDusp('(O330 -> [Delay delay:100]) for 0.02 then O220'), 0.03).then(() => 'resolved', (e) => String(e))
  console.log(JSON.stringify(report))
  process.exit(report.failed.length ? 1 : 0)
}
main().catch((e) => { console.log(JSON.stringify({ fatal: String(e && e.stack || e) })); process.exit(2) })
