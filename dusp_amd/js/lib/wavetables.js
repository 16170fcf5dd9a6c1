'use strict'
/* Oscillator wave tables (ids 0-4) and Shape tables (ids 5-8: decay, attack, semiSine, decaySquared), computed on the host exactly as the reference computes them
 * (src/components/Osc/waveTables.js:5-40) and handed to the device as data.  Under Node these are
 * V8's own Math.sin values, i.e. bit-identical to what `dusp` itself would hold. */
const cache = new Map()

function makeTables(sampleRate) {
  if (cache.has(sampleRate)) return cache.get(sampleRate)
  const n = sampleRate + 1
  const TWO_PI = 2 * Math.PI
  const sin = new Float32Array(n)
  for (let t = 0; t < n; t++) sin[t] = Math.sin(TWO_PI * t / n) // period = table LENGTH, not sampleRate
  const saw = new Float32Array(n)
  for (let t = 0; t < sampleRate; t++) saw[t] = -1 + t * 2 / n // last entry stays 0
  const square = new Float32Array(n)
  square.fill(1, 0, sampleRate / 2)
  square.fill(-1, sampleRate / 2, n)
  const triangle = new Float32Array(n)
  const q = sampleRate / 4
  for (let t = 0; t < q; t++) { // later quarters re-read the f32-rounded first quarter
    triangle[t] = t / sampleRate * 4
    triangle[t + q] = 1 - triangle[t]
    triangle[t + 2 * q] = -triangle[t]
    triangle[t + 3 * q] = -1 + triangle[t]
  }
  triangle[sampleRate] = 0
  const eightBit = sin.map((s) => Math.round(s * 128) / 128)
  // Shape's tables (src/components/Shape/shapeTables.js:3-38): func(x / sampleRate) for x = 0..sampleRate
  const shape = (func) => { const t = new Float32Array(n); for (let x = 0; x < n; x++) t[x] = func(x / sampleRate); return t }
  const tables = [sin, saw, square, triangle, eightBit,
    shape((x) => 1 - x), shape((x) => x), shape((x) => Math.sin(Math.PI * x)), shape((x) => (1 - x) * (1 - x))]
  cache.set(sampleRate, tables)
  return tables
}

module.exports = { makeTables }
