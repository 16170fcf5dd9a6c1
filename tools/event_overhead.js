'use strict'
/* Host-side overhead of event-segmented rendering: a Shape retriggered `rate` times per second for `seconds` seconds
 * through the JS surface.   node tools/event_overhead.js --sampleRate=48000 [--rate=50] [--seconds=10] */
const lib = require('../dusp_amd/js')
const argv = require('minimist')(process.argv.slice(2))
const rate = argv.rate || 50, seconds = argv.seconds || 10

async function main() {
  await lib.renderChannelData(new lib.Osc(440), 0.1) // context + tables
  for (const r of [0, rate]) {
    const s = new lib.Shape('decay', 0.01).trigger()
    if (r) new lib.Retriggerer(s, r)
    const g = new lib.Multiply(new lib.Osc(330.5, 'saw'), s)
    const t0 = process.hrtime.bigint()
    const cd = await lib.renderChannelData(g, seconds)
    const ms = Number(process.hrtime.bigint() - t0) / 1e6
    const segments = r ? Math.round(seconds * r) : 1
    console.log(`retrigger rate ${r} Hz: ${seconds} s rendered in ${ms.toFixed(1)} ms (${segments} segments, ${(ms / segments).toFixed(3)} ms per segment), ${cd[0].length} samples`)
  }
}
main().catch((e) => { console.error(e); process.exit(1) })
