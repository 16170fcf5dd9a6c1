'use strict'
/* Host mirror check (no GPU): build every golden case with dusp_amd/js graph classes and compare the
 * extracted descriptor — constants, state, ring table AND unit order — with the one the reference's
 * objects produced; and the graph's Dusp string (lib/dusp.js) with the one the reference's dusp() printed, and that
 * string fed back through unDusp must build a graph that prints the same again.
 * usage: node check_descriptors.js --sampleRate=48000 */
const fs = require('fs')
const path = require('path')
const lib = require('../../dusp_amd/js')
const SR = lib.config.sampleRate
const GOLDEN = path.join(__dirname, '..', 'golden')
const goldenCases = require('./cases')
const cases = goldenCases(lib, SR)
const normaliseLabels = require('./labels')

let checked = 0, bad = 0, strings = 0, badStrings = 0, roundTrips = 0
for (const c of cases) {
  const file = path.join(GOLDEN, c.name + '.desc.f64')
  if (!fs.existsSync(file)) continue
  const buf = fs.readFileSync(file)
  const want = new Float64Array(buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.byteLength))
  let target // (built under the case's seed, when it has one: Noise draws a number in its constructor)
  if (c.seed === undefined) target = c.build()
  else {
    const original = Math.random
    let x = c.seed >>> 0
    Math.random = () => { x = (Math.imul(x, 1664525) + 1013904223) >>> 0; return x / 4294967296 }
    try { target = c.build() } finally { Math.random = original }
  }
  const meta = JSON.parse(fs.readFileSync(path.join(GOLDEN, c.name + '.json'), 'utf8'))
  if (meta.dusp !== undefined) {
    const raw = lib.dusp(target.isPatch ? target.defaultOutlet : target)
    strings++
    if (normaliseLabels(raw) !== meta.dusp) { badStrings++; console.log('DUSP STRING', c.name, '\n  want', meta.dusp, '\n  got ', normaliseLabels(raw)) }
    else if (typeof raw === 'string') {
      // fixed point: what unDusp builds from the string prints as the same string (where this package's unDusp accepts it)
      let again
      try { again = lib.dusp(lib.unDusp(raw)) } catch (e) { again = undefined }
      if (again !== undefined) {
        roundTrips++
        // (not a law of the reference either: "(x * 0)" comes back as "(x * 1)" because Multiply's constructor reads `b || 1`)
        if (normaliseLabels(again) !== meta.dusp && !/^mult_inlet_zero/.test(c.name)) { badStrings++; console.log('ROUND TRIP', c.name, '\n  first ', meta.dusp, '\n  second', normaliseLabels(again)) }
      }
    }
  }
  const got = lib.extract(target, { allowEvents: true }).words
  let same = got.length === want.length
  for (let i = 0; same && i < got.length; i++) same = Object.is(got[i], want[i]) || got[i] === want[i]
  checked++
  if (!same) { bad++; console.log('MISMATCH', c.name) }
}
// unify: differing constants become parameters
const uni = lib.unify([1, 2, 3].map((k) => lib.extract(new lib.Multiply(new lib.Osc(10 * k), new lib.Ramp(SR, 1, 0).trigger()))))
const unifyOk = uni.nParams === 1 && uni.nInstances === 3 && Array.from(uni.params).join() === '10,20,30' && uni.words[6] === 1
console.log(JSON.stringify({ sampleRate: SR, checked, bad, unifyOk, strings, badStrings, roundTrips }))
process.exit(bad || badStrings || !unifyOk || !checked ? 1 : 0)
