'use strict'
/* Where one renderChannelData(unit, 10) spends its wall time (HOST side), per stage: extract, programBuild, render (upload + kernel + download),
 * write-back (stateDownload per unit), programDestroy.   node tools/surface_breakdown.js --sampleRate=48000 */
const lib = require('../dusp_amd/js')
const native = require('../dusp_amd/js/lib/native')()
const { makeTables } = require('../dusp_amd/js/lib/wavetables')
const ms = (t0) => Number(process.hrtime.bigint() - t0) / 1e6
async function main() {
  const ctx = native.ctxCreate(-1)
  makeTables(48000).forEach((t, id) => native.tableUpload(ctx, id, t))
  for (const text of ['O440', 'O440 * D1', 'Z110 -> LP800', '[Delay in:O500 delay:300.5]']) {
    for (let rep = 0; rep < 4; rep++) {
      let t = process.hrtime.bigint()
      const outlet = lib.unDusp(text)
      const tParse = ms(t); t = process.hrtime.bigint()
      const ex = lib.extract(outlet)
      const tExtract = ms(t); t = process.hrtime.bigint()
      const prog = native.programBuild(ctx, ex.words, 0)
      const tBuild = ms(t); t = process.hrtime.bigint()
      const pcm = await native.render(prog, 1, 480000, null)
      const tRender = ms(t); t = process.hrtime.bigint()
      for (let u = 0; u < ex.circuit.units.length; u++) native.stateDownload(prog, 0, u)
      const tState = ms(t); t = process.hrtime.bigint()
      native.programDestroy(prog)
      const tDestroy = ms(t)
      if (rep >= 2) console.log(`${text.padEnd(30)} parse ${tParse.toFixed(2)} extract ${tExtract.toFixed(2)} build ${tBuild.toFixed(2)} render ${tRender.toFixed(2)} state ${tState.toFixed(2)} destroy ${tDestroy.toFixed(2)} ms  (${pcm.length} samples)`)
    }
  }
}
main().catch((e) => { console.error(e); process.exit(1) })
