'use strict'
/* Operator helpers with the reference's folding rules (src/quick.js:15-110): numbers fold, signals
 * build a unit.  Helpers whose unit the GPU path does not execute yet fold numbers and refuse signals. */
const { Sum, Multiply } = require('./graph')

const isNum = (x) => typeof x === 'number'
const isSignal = (x) => x && (x.isUnitOrPatch || x.isOutlet)

exports.add = (a, b) => (isNum(a) && isNum(b) ? a + b : new Sum(a, b))

exports.mult = function (a, b) {
  if (a === undefined || a === null || a === 1) return b
  if (b === undefined || b === null || b === 1) return a
  return isNum(a) && isNum(b) ? a * b : new Multiply(a, b)
}
exports.multiply = exports.mult

function numbersOnly(name, fold) {
  return function (...args) {
    if (args.some(isSignal)) throw 'dusp-hip: quick.' + name + ' on signals needs a unit the GPU path does not execute yet'
    return fold(...args)
  }
}
exports.subtract = numbersOnly('subtract', (a, b) => a - b)
exports.divide = numbersOnly('divide', (a, b) => a / b)
exports.invert = numbersOnly('invert', (a) => -a)
exports.semitoneToRatio = numbersOnly('semitoneToRatio', (p) => Math.pow(2, p / 12))
exports.pToF = numbersOnly('pToF', (p) => Math.pow(2, (p - 69) / 12) * 440)
exports.pow = numbersOnly('pow', (a, b) => Math.pow(a, b))
exports.concat = numbersOnly('concat', (a, b) => [].concat(a, b))
exports.clipAbove = numbersOnly('clipAbove', (x, th) => (x > th ? th : x))
exports.clipBelow = numbersOnly('clipBelow', (x, th) => (x < th ? th : x))
