'use strict'
/* TEST INFRASTRUCTURE — golden frames of the reference's RenderStream (src/RenderStream.js).  Build container only.
 *   node oracle/js/gen_golden_stream.js --sampleRate=48000
 * For every case of tests/js/stream_cases.js: read `chunks` buffers from `new RenderStream(outlet, channels)` and
 * write them back to back as <name>.frames.f32 (interleaved f32 LE, exactly the stream's wire format) + <name>.json. */
const fs = require('fs')
const path = require('path')
const crypto = require('crypto')
const { ref } = require('./refload')
const argv = require('minimist')(process.argv.slice(2))
const OUT = path.resolve(argv.out || path.join(__dirname, '../../tests/golden'))
const SR = ref('config.js').sampleRate
const log = console.log
console.log = console.warn = () => {} // the reference's RenderStream prints its format and every gain change
const RenderStream = ref('RenderStream.js')
const L = {}
for (const n of ['Ramp', 'Multiply', 'Sum', 'Delay', 'Filter', 'Pan']) L[n] = ref('components/' + n + '.js')
L.Osc = ref('components/Osc')
const cases = require('../../tests/js/stream_cases')(L, SR)

function take(stream, chunks) {
  return new Promise((resolve, reject) => {
    const got = []
    stream.on('data', (buf) => {
      if (got.length < chunks) got.push(Float32Array.from(buf))
      if (got.length === chunks) { // stop the producer for good: a flowing stream without listeners would tick forever
        stream._read = () => {}
        stream.circuit.stopTicking()
        stream.removeAllListeners('data')
        stream.pause()
        resolve(got)
      }
    })
    stream.on('error', reject)
  })
}

async function main() {
  const index = []
  for (const c of cases) {
    const stream = new RenderStream(c.build(), c.channels)
    const bufs = await take(stream, c.chunks)
    const all = Buffer.concat(bufs.map((b) => Buffer.from(b.buffer, b.byteOffset, b.byteLength)))
    fs.writeFileSync(path.join(OUT, c.name + '.frames.f32'), all)
    const meta = { name: c.name, sample_rate: SR, channels: c.channels, chunks: c.chunks, format: stream.format,
      normalise_factor_after: stream.normaliseFactor, sha256: crypto.createHash('sha256').update(all).digest('hex') }
    fs.writeFileSync(path.join(OUT, c.name + '.json'), JSON.stringify(meta, null, 1) + '\n')
    index.push(c.name)
    log(c.name, 'chunks=' + c.chunks, 'normaliseFactor=' + stream.normaliseFactor)
  }
  fs.writeFileSync(path.join(OUT, 'index_streams.json'), JSON.stringify(index, null, 1) + '\n')
  process.exit(0)
}
main().catch((e) => { console.error('gen_golden_stream failed:', e); process.exit(1) })
