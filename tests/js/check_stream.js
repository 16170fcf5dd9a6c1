'use strict'
/* RenderStream + WAV (GPU): this package's RenderStream must emit the reference's frames, chunk for chunk —
 * including the auto-normalising gain — while rendering in blocks that continue one device program.
 *   node check_stream.js --sampleRate=48000 */
const fs = require('fs')
const path = require('path')
const lib = require('../../dusp_amd/js')
const SR = lib.config.sampleRate
const GOLDEN = path.join(__dirname, '..', 'golden')
const cases = require('./stream_cases')(lib, SR)
const USES_DEVICE_MATH = /^stream_(growing_stereo|feedback)/ // Pan: device pow(); Filter: device tan()

function take(stream, chunks) {
  return new Promise((resolve, reject) => {
    const got = []
    stream.on('data', (buf) => {
      if (got.length < chunks) got.push(buf)
      if (got.length === chunks) { stream.stop(); resolve(got) }
    })
    stream.on('error', reject)
  })
}

async function main() {
  const report = { checked: 0, failed: [], blocks: [] }
  for (const c of cases) {
    const meta = JSON.parse(fs.readFileSync(path.join(GOLDEN, c.name + '.json')))
    const buf = fs.readFileSync(path.join(GOLDEN, c.name + '.frames.f32'))
    const want = new Float32Array(buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.byteLength))
    for (const blockChunks of [7, 64]) { // block size must not be observable
      const stream = new lib.RenderStream(c.build(), c.channels, { blockChunks })
      const bufs = await take(stream, c.chunks)
      let ok = bufs.every((b) => b instanceof Float32Array && b.length === 256 * c.channels) &&
        JSON.stringify(stream.format) === JSON.stringify(meta.format)
      let exact = true, maxErr = 0
      for (let k = 0; ok && k < bufs.length; k++)
        for (let j = 0; j < bufs[k].length; j++) {
          const a = bufs[k][j], b = want[k * 256 * c.channels + j]
          if (a !== b) exact = false
          maxErr = Math.max(maxErr, Math.abs(a - b))
        }
      const gainOk = exact ? stream.normaliseFactor === meta.normalise_factor_after
        : Math.abs(stream.normaliseFactor - meta.normalise_factor_after) <= 1e-5 * meta.normalise_factor_after
      report.checked++
      if (!(ok && gainOk && (exact || (USES_DEVICE_MATH.test(c.name) && maxErr <= 1e-5))))
        report.failed.push({ name: c.name, blockChunks, ok, exact, maxErr, gain: stream.normaliseFactor, want: meta.normalise_factor_after })
    }
  }
  // WAV: a stereo render as 32-bit float frames and as 16-bit PCM, decoded back
  const cd = await lib.renderChannelData(new lib.Pan(new lib.Osc(440), 0.25), 0.05)
  const f32 = lib.decodeWav(lib.encodeWav(cd))
  const s16 = lib.decodeWav(lib.encodeWav(cd, { bitDepth: 16 }))
  report.wavFloatExact = f32.format === 3 && f32.sampleRate === SR && f32.numberOfChannels === 2 &&
    f32.channelData.every((ch, c) => ch.length === cd[c].length && ch.every((v, t) => v === cd[c][t]))
  report.wavPcm16Close = s16.format === 1 && s16.bitDepth === 16 &&
    s16.channelData.every((ch, c) => ch.every((v, t) => Math.abs(v - cd[c][t]) <= 0.5 / 32767 + 1e-7))
  // interleaved frames straight from the device == planar render, transposed
  const native = require('../../dusp_amd/js/lib/native')()
  const ex = lib.extract(new lib.Pan(new lib.Osc(440), 0.25))
  const ctx = native.ctxCreate(-1)
  require('../../dusp_amd/js/lib/wavetables').makeTables(SR).forEach((t, id) => native.tableUpload(ctx, id, t))
  const prog = native.programBuild(ctx, ex.words, 0)
  const frames = await native.render(prog, 1, 2400, null, true)
  report.deviceFramesMatch = frames.length === 4800 && cd[0].every((v, t) => frames[2 * t] === v && frames[2 * t + 1] === cd[1][t])
  native.programDestroy(prog)
  console.log(JSON.stringify(report))
  process.exit(report.failed.length || !report.wavFloatExact || !report.wavPcm16Close || !report.deviceFramesMatch ? 1 : 0)
}
main().catch((e) => { console.log(JSON.stringify({ fatal: String(e && e.stack || e) })); process.exit(2) })
