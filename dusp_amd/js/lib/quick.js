'use strict'
/* Operator helpers with the reference's folding rules (src/quick.js:15-110): numbers fold, signals
 * build a unit. */
const { ConcatChannels, Sum, Multiply, Subtract, Divide, PolarityInvert, SemitoneToRatio, Pow, HardClipAbove, HardClipBelow } = require('./graph')

const isNum = (x) => typeof x === 'number'
const isSignal = (x) => x && (x.isUnitOrPatch || x.isOutlet)

exports.add = (a, b) => (isNum(a) && isNum(b) ? a + b : new Sum(a, b))

exports.mult = function (a, b) {
  if (a === undefined || a === null || a === 1) return b
  if (b === undefined || b === null || b === 1) return a
  return isNum(a) && isNum(b) ? a * b : new Multiply(a, b)
}
exports.multiply = exports.mult

exports.subtract = (a, b) => (isNum(a) && isNum(b) ? a - b : new Subtract(a, b))
exports.divide = (a, b) => (isNum(a) && isNum(b) ? a / b : new Divide(a, b))
exports.invert = (a) => (isNum(a) ? -a : new PolarityInvert(a))
exports.semitoneToRatio = (p) => (isNum(p) ? Math.pow(2, p / 12) : new SemitoneToRatio(p))
exports.pToF = function (p) {
  if (isNum(p)) return Math.pow(2, (p - 69) / 12) * 440
  throw 'quick.pToF(non number) has not been implemented' // the reference's own message (quick.js:55)
}
exports.pow = (a, b) => (isSignal(a) || isSignal(b) ? new Pow(a, b) : Math.pow(a, b))
exports.clipAbove = (input, th) => (isSignal(input) || isSignal(th) ? new HardClipAbove(input, th) : (input > th ? th : input))
exports.clipBelow = (input, th) => (isSignal(input) || isSignal(th) ? new HardClipBelow(input, th) : (input < th ? th : input))
exports.clip = function (input, th) {
  // the reference's signal branch calls `new Clip(input, threshold)` without importing Clip: it throws (quick.js:102-104)
  if (isSignal(input) || isSignal(th)) throw 'dusp-hip: quick.clip on signals is broken in the reference (Clip is not imported); build a Clip unit directly'
  return Math.abs(input) < Math.abs(th) ? th : input // sic: the reference's number branch (quick.js:106-109)
}
exports.concat = (a, b) => (isSignal(a) || isSignal(b) ? new ConcatChannels(a, b) : [].concat(a, b)) // quick.js:68-73
