'use strict'
/* Host mirror check (no GPU): build every golden case with dusp_amd/js graph classes and compare the
 * extracted descriptor — constants, state, ring table AND unit order — with the one the reference's
 * objects produced.  usage: node check_descriptors.js --sampleRate=48000 */
const fs = require('fs')
const path = require('path')
const lib = require('../../dusp_amd/js')
const SR = lib.config.sampleRate
const GOLDEN = path.join(__dirname, '..', 'golden')
const cases = require('./cases')(lib, SR)

let checked = 0, bad = 0
for (const c of cases) {
  const file = path.join(GOLDEN, c.name + '.desc.f64')
  if (!fs.existsSync(file)) continue
  const buf = fs.readFileSync(file)
  const want = new Float64Array(buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.byteLength))
  const got = lib.extract(c.build(), { allowEvents: true }).words
  let same = got.length === want.length
  for (let i = 0; same && i < got.length; i++) same = Object.is(got[i], want[i]) || got[i] === want[i]
  checked++
  if (!same) { bad++; console.log('MISMATCH', c.name) }
}
// unify: differing constants become parameters
const uni = lib.unify([1, 2, 3].map((k) => lib.extract(new lib.Multiply(new lib.Osc(10 * k), new lib.Ramp(SR, 1, 0).trigger()))))
const unifyOk = uni.nParams === 1 && uni.nInstances === 3 && Array.from(uni.params).join() === '10,20,30' && uni.words[6] === 1
console.log(JSON.stringify({ sampleRate: SR, checked, bad, unifyOk }))
process.exit(bad || !unifyOk || !checked ? 1 : 0)
