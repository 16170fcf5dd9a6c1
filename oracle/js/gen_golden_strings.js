'use strict'
/* TEST INFRASTRUCTURE — golden vectors for the Dusp STRING front-end.  Runs ONLY in the build container.
 *
 *   node oracle/js/gen_golden_strings.js [--sampleRate=48000] [--out=tests/golden]
 *
 * src/unDusp.js cannot be imported from /root/reference/src (its parser, src/parseDSP, is an empty git submodule),
 * but the reference's own browserify bundle (demos/browser-dusp-demo-2.bundle.js) still contains the complete
 * library INCLUDING that parser.  This script evaluates the bundle (oracle/js/refload.js), takes ITS unDusp and
 * renderChannelData, and for every string in tests/js/string_cases.js writes
 *   tests/golden/str_ast.json         the reference parser's syntax tree for every string (also the ones it rejects)
 *   str_<name>.desc.f64 / .pcm.f32 / .json   as oracle/js/gen_golden.js does, for the strings that build a graph
 * Only data leaves this script.
 */
const fs = require('fs')
const path = require('path')
const crypto = require('crypto')
const argv = require('minimist')(process.argv.slice(2))
const { bundle } = require('./refload')
const { extract } = require('../../dusp_amd/js/lib/extract')

const SR = argv.sampleRate || 48000
const OUT = path.resolve(argv.out || path.join(__dirname, '../../tests/golden'))
const quiet = (f) => { const log = console.log, warn = console.warn; console.log = console.warn = () => {}; try { return f() } finally { console.log = log; console.warn = warn } }

const config = bundle(99)
config.sampleRate = SR // before any unit / table module of the bundle is loaded
config.sampleInterval = 1 / SR
const refParse = quiet(() => bundle(116)) // src/parseDSP/getExpression.js
const unDusp = quiet(() => bundle(171))
const renderChannelData = quiet(() => bundle(170))
const { graphs, syntax } = require('../../tests/js/string_cases')

const plain = (node) => JSON.parse(JSON.stringify(node, (k, v) => (typeof v === 'number' && !Number.isFinite(v) ? String(v) : v)))

async function main() {
  const asts = []
  for (const text of syntax.concat(graphs.map((g) => g.text))) {
    let tree
    try { tree = quiet(() => refParse(text)) } catch (e) { tree = { threw: String(e) } }
    asts.push({ text, tree: tree === null ? null : plain(tree) })
  }
  fs.writeFileSync(path.join(OUT, 'str_ast.json'), '[\n' + asts.map((a) => JSON.stringify(a)).join(',\n') + '\n]\n') // one string per line

  const index = []
  for (const g of graphs) {
    const name = 'str_' + g.name
    let target
    try { target = quiet(() => unDusp(g.text)) } catch (e) {
      index.push({ name, text: g.text, throws: String(e) })
      console.log(name, 'THROWS', String(e))
      continue
    }
    if (!target || !(target.isUnit || target.isOutlet || target.isPatch)) { // a folded number, a string, nothing at all
      const rejects = await renderChannelData(target, g.duration).then(() => null, (e) => String(e))
      index.push({ name, text: g.text, value: target === undefined ? null : target, render_rejects: rejects })
      console.log(name, 'VALUE', target, 'render rejects:', rejects)
      continue
    }
    const ex = extract(target, { allowEvents: true })
    const cd = await quiet(() => renderChannelData(target, g.duration))
    const n = cd[0].length
    const h = crypto.createHash('sha256')
    for (const ch of cd) h.update(Buffer.from(ch.buffer, ch.byteOffset, ch.byteLength))
    fs.writeFileSync(path.join(OUT, name + '.pcm.f32'), Buffer.concat(cd.map((ch) => Buffer.from(ch.buffer, ch.byteOffset, ch.byteLength))))
    fs.writeFileSync(path.join(OUT, name + '.desc.f64'), Buffer.from(ex.words.buffer, ex.words.byteOffset, ex.words.byteLength))
    const meta = { name, text: g.text, sample_rate: SR, chunk_size: ex.chunkSize, duration: g.duration, n_samples: n,
      n_channels: cd.length, windows: [[0, n]], sha256_full: h.digest('hex'), has_events: ex.circuit.events.length > 0 || !!g.events }
    fs.writeFileSync(path.join(OUT, name + '.json'), JSON.stringify(meta, null, 1) + '\n')
    index.push({ name, text: g.text, duration: g.duration, events: meta.has_events })
    console.log(name, 'n=' + n, 'ch=' + cd.length, 'units=' + ex.circuit.units.length)
  }
  fs.writeFileSync(path.join(OUT, 'index_strings.json'), JSON.stringify(index, null, 1) + '\n')
}
main().catch((e) => { console.error('gen_golden_strings failed:', e); process.exit(1) })
