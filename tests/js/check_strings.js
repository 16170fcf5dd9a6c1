'use strict'
/* The Dusp string front-end (dusp_amd/js/lib/parse.js + unDusp.js) against the reference's own parser and
 * constructors, whose results were captured from the reference's bundle by oracle/js/gen_golden_strings.js.
 *   node check_strings.js --sampleRate=48000 [--render]     (--render needs the GPU)
 * 1. syntax trees: node for node, including lengths, for ~100 strings (also rejected input: null)
 * 2. graphs: unDusp(text) extracts to the reference's descriptor; strings the reference rejects are rejected
 * 3. --render: renderChannelData(unDusp(text), duration) equals the reference's PCM */
const fs = require('fs')
const path = require('path')
const lib = require('../../dusp_amd/js')
const argv = require('minimist')(process.argv.slice(2))
const GOLDEN = path.join(__dirname, '..', 'golden')
const plain = (node) => JSON.parse(JSON.stringify(node, (k, v) => (typeof v === 'number' && !Number.isFinite(v) ? String(v) : v)))
const USES_DEVICE_MATH = /^str_(saw_lowpass|square_highpass|pan|pow_clip|semitone)/ // device tan() / pow()

function canonical(x) { // key order must not matter
  if (Array.isArray(x)) return x.map(canonical)
  if (x && typeof x === 'object') { const o = {}; for (const k of Object.keys(x).sort()) o[k] = canonical(x[k]); return o }
  return x
}

async function main() {
  const report = { trees: 0, treeMismatches: [], graphs: 0, graphMismatches: [], rejected: 0, rendered: 0, renderFailures: [] }
  for (const { text, tree } of JSON.parse(fs.readFileSync(path.join(GOLDEN, 'str_ast.json')))) {
    let mine
    try { mine = lib.parse.parseExpression(text) } catch (e) { mine = { threw: String(e) } }
    report.trees++
    if (JSON.stringify(canonical(mine === null ? null : plain(mine))) !== JSON.stringify(canonical(tree)))
      report.treeMismatches.push({ text, mine: mine === null ? null : plain(mine), reference: tree })
  }
  for (const g of JSON.parse(fs.readFileSync(path.join(GOLDEN, 'index_strings.json')))) {
    report.graphs++
    let target, threw = null
    try { target = lib.unDusp(g.text) } catch (e) { threw = e }
    if (g.throws !== undefined) { // the reference rejects this string: so must we, with a string
      if (threw === null || typeof threw !== 'string') report.graphMismatches.push({ name: g.name, expected: 'throws ' + g.throws, threw: String(threw) })
      else report.rejected++
      continue
    }
    if (threw !== null) { report.graphMismatches.push({ name: g.name, threw: String(threw) }); continue }
    if (g.value !== undefined) { // not a graph: a folded number (or nothing); renderChannelData refuses it like the reference
      const rejects = await lib.renderChannelData(target, 0.01).then(() => null, (e) => String(e))
      if (!(Object.is(target, g.value) || (target === undefined && g.value === null)) || rejects !== g.render_rejects)
        report.graphMismatches.push({ name: g.name, value: target, rejects })
      else report.rejected++
      continue
    }
    const buf = fs.readFileSync(path.join(GOLDEN, g.name + '.desc.f64'))
    const want = new Float64Array(buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.byteLength))
    const got = lib.extract(target, { allowEvents: true }).words
    let same = got.length === want.length
    for (let i = 0; same && i < got.length; i++) same = Object.is(got[i], want[i]) || got[i] === want[i]
    if (!same) { report.graphMismatches.push({ name: g.name, descriptor: 'differs' }); continue }
    if (!argv.render) continue
    const pbuf = fs.readFileSync(path.join(GOLDEN, g.name + '.pcm.f32'))
    const pcm = new Float32Array(pbuf.buffer.slice(pbuf.byteOffset, pbuf.byteOffset + pbuf.byteLength))
    const cd = await lib.renderChannelData(lib.unDusp(g.text), g.duration)
    const n = cd[0].length
    let exact = cd.length * n === pcm.length, maxErr = 0, scale = 0
    for (let c = 0; c < cd.length && cd.length * n === pcm.length; c++)
      for (let t = 0; t < n; t++) {
        const a = cd[c][t], b = pcm[c * n + t]
        if (a !== b) exact = false
        maxErr = Math.max(maxErr, Math.abs(a - b)); scale = Math.max(scale, Math.abs(b))
      }
    if (exact || (cd.length * n === pcm.length && USES_DEVICE_MATH.test(g.name) && maxErr <= 1e-5 * scale)) report.rendered++
    else report.renderFailures.push({ name: g.name, maxErr, channels: cd.length, n })
  }
  console.log(JSON.stringify(report))
  process.exit(report.treeMismatches.length || report.graphMismatches.length || report.renderFailures.length ? 1 : 0)
}
main().catch((e) => { console.log(JSON.stringify({ fatal: String(e && e.stack || e) })); process.exit(2) })
