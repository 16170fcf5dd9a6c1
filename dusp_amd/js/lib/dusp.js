'use strict'
/* dusp(thing) -> its Dusp string: the inverse of unDusp (reference src/dusp.js:4-130 and the `dusp` descriptors on
 * Osc.js:22-33, Multiply.js:17-21, Sum.js:12-16, Repeater.js:15-22, Shape/index.js:61-76).
 *
 * A unit prints as "[Kind #label INLET:value ... extra]"; a unit met a second time prints as "#label".  Shorthands
 * ("O440", "(a * b)", "(a + b)") are used only for a unit NOTHING listens to: the reference's count of outgoing
 * connections adds ARRAYS to a number (Unit.js:144-149), which yields the string "0" for a unit without listeners
 * ("0" <= 1) and "0[object Object]" otherwise (NaN) — so in practice only the root of the expression is ever shortened.
 * Constants print through String(): an array constant inside a unit prints as "1,2", a bare inlet as "(1,2)". */
const config = require('./config')

function shorthandAllowed(unit) { // the reference's arithmetic, spelled out (see above)
  let n = 0
  for (const name of Object.keys(unit.outlets)) n += unit.outlets[name].connections
  return n <= 1
}

function dusp(o, index) {
  index = index || {}
  if (o === undefined) return undefined
  if (o === null) return null
  if (o === 0 || (o && o.constructor === Number)) return o
  if (o && o.constructor === String) return '"' + o + '"'
  if (o.isUnit) {
    if (index[o.label]) return '#' + o.label
    index[o.label] = o
    const useShorthand = config.useDuspShorthands ? shorthandAllowed(o) : false
    const how = o.dusp
    if (useShorthand && how && how.shorthand) {
      const short = how.shorthand.call(o, index)
      if (short) return short
    }
    const args = [o.constructor.name]
    if (!useShorthand) args.push('#' + o.label)
    for (const name of Object.keys(o.inlets)) {
      const inlet = o.inlets[name]
      args.push(name.toUpperCase() + ':' + (inlet.outlet ? duspOutlet(inlet.outlet, index) : inlet.constant))
    }
    if (how) {
      const extra = how.extraProperties
      if (Array.isArray(extra)) for (const prop of extra) args.push(prop + ':' + dusp(o[prop]))
      else if (extra) { for (const prop of Object.keys(extra)) if (o[prop] != extra[prop]) args.push(prop + ':' + dusp(o[prop])) }
      if (how.extraArgs) {
        const more = how.extraArgs.call(o)
        if (more) args.push(...more)
      }
    }
    return '[' + args.join(' ') + ']'
  }
  if (o.isOutlet) return duspOutlet(o, index)
  if (o.isInlet) return duspInlet(o, index)
  return null // (patches included: the reference warns "unable to turn object to dusp")
}

function duspOutlet(outlet, index) {
  if (outlet === outlet.unit.defaultOutlet) return dusp(outlet.unit, index)
  return dusp(outlet.unit, index) + '.' + outlet.name.toUpperCase()
}

function duspInlet(inlet, index) {
  if (inlet.connected) return dusp(inlet.outlet, index)
  if (inlet.constant.constructor === Number) return inlet.constant
  if (Array.isArray(inlet.constant)) return '(' + inlet.constant.join(',') + ')'
  throw 'strange constant: ' + inlet.constant
}

/* the per-kind descriptors, attached to this package's classes (dusp() reads `unit.dusp` like the reference does) */
function describe(graph) {
  graph.Osc.prototype.dusp = {
    extraProperties: { waveform: 'sin' },
    shorthand() { if (this.waveform == 'sin' && !this.F.connected) return 'O' + this.F.constant },
  }
  graph.Multiply.prototype.dusp = { shorthand(index) { return '(' + dusp(this.A, index) + ' * ' + dusp(this.B, index) + ')' } }
  graph.Sum.prototype.dusp = { shorthand(index) { return '(' + dusp(this.A, index) + ' + ' + dusp(this.B, index) + ')' } }
  graph.Repeater.prototype.dusp = { extraArgs() { return this.measuredIn ? ['"' + this.measuredIn + '"'] : null } }
  graph.Shape.prototype.dusp = {
    flagFunctions: { trigger() { this.trigger() } },
    extraArgs() { return this.playing ? ['trigger'] : [] },
    extraProperties: ['shape'],
  }
}

describe(require('./graph'))

module.exports = dusp
module.exports.usingShorthands = config.useDuspShorthands
