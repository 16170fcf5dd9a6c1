'use strict'
/* unDusp(string | number | unit | outlet) -> the thing itself, or the graph a Dusp string describes
 * (reference src/unDusp.js:4-16 + src/construct/*.js), built from this package's unit classes.
 *
 * The syntax tree comes from lib/parse.js; construction follows the reference's rules node by node, including the
 * way it threads (and loses) the `#id` index: only an object's named attributes share the index of the object they
 * belong to — positional arguments and the two sides of an operator each start from scratch, so "#id" references
 * resolve inside "[Multiply a:[Osc #x 3] b:#x]" but not in "[Osc #x 3] * #x" (the reference throws there, and so
 * does this).  `!` / `~!` build a Retriggerer / SporadicRetriggerer, which this package ticks on the host
 * between segments.  `then` rewires the graph from a finish callback: renderChannelData rebuilds the device program there.
 */
const graph = require('./graph')
const quick = require('./quick')
const { parseExpression } = require('./parse')
require('./dusp') // attaches the `dusp` descriptors (flag functions) to the unit classes

/* constructor name -> class: everything lib/graph.js can execute (reference src/patchesAndComponents.js; the reference's
 * patches are host-side builders that wire these units — they are not mirrored here: with the real `dusp` package a
 * patch's units reach the extractor as they are, INTEGRATION.md) */
const COMPONENTS = {}
for (const name of Object.keys(graph)) {
  const C = graph[name]
  if (typeof C === 'function' && C.prototype instanceof graph.Unit) COMPONENTS[name] = C
}

const SHORTHAND = { // reference src/construct/shorthandConstructors.js:3-46
  O: (f) => new graph.Osc(f),
  Z: (f) => { const o = new graph.Osc(f); o.waveform = 'saw'; return o },
  Sq: (f) => { const o = new graph.Osc(f); o.waveform = 'square'; return o },
  A: (time) => new graph.Shape('attack', time).trigger(),
  D: (time) => new graph.Shape('decay', time).trigger(),
  t: () => new graph.Timer(),
  LP: (freq) => new graph.Filter(null, freq),
  HP: (freq) => new graph.Filter(null, freq, 'HP'),
  AP: (delayTime, feedback) => new graph.AllPass(delayTime, feedback),
  random: () => Math.random(),
}

const isSignal = (x) => x && (x.isUnitOrPatch || x.isOutlet)

/* `index` is what the reference passes along: an object (ids -> units), undefined (start a new one), or — when the call
 * came through Array.prototype.map — the position of the argument, on which ids can be neither stored nor found. */
function construct(node, index) {
  if (typeof node === 'string') {
    const text = node
    node = parseExpression(text, typeof index === 'number' ? index : 0) // sic: the index doubles as start offset
    if (!node) throw "Can't construct expression: " + text
  }
  switch (node.type) {
    case 'object': return constructObject(node, index)
    case 'number': return node.n
    case 'id': {
      const found = index && typeof index === 'object' ? index[node.id] : undefined
      if (found) return found
      throw 'Error: Referencing an object which has not been declared: #' + node.id
    }
    case 'operation': return constructOperation(node, index)
    case 'objectProperty': return construct(node.object, index)[node.property]
    case 'shorthand': return constructShorthand(node)
    case 'unnamedArgument': return construct(node.value, index)
    case 'string': return node.string
    case 'json': return node.o
    default: throw 'Unknown expression type: ' + node.type
  }
}

function constructObject(node, index) {
  if (!index) index = {}
  const C = COMPONENTS[node.constructor]
  if (!C) {
    if (require('./parse').REFERENCE_NAMES.includes(node.constructor))
      throw 'dusp-hip: unit type not supported on the GPU path: ' + node.constructor
    throw 'Unknown object constructor: ' + node.constructor
  }
  const args = node.arguments.map((arg, position) => construct(arg, position))
  const obj = new C(...args)
  if (node.id) obj.label = node.id
  if (typeof index === 'object') {
    if (index[obj.label]) { if (index[obj.label] !== obj) throw obj.label } else index[obj.label] = obj
  }
  for (const attr of node.attributes) {
    const upper = attr.property.toUpperCase()
    const asInlet = obj[upper] && obj[upper].isInlet // also true for a patch's aliased inlets (constructObject.js:38-40)
    const value = construct(attr.value, index)
    if (asInlet) obj[upper] = value
    else obj[attr.property] = value
  }
  if (obj.dusp && obj.dusp.flagFunctions) // (only Shape has one: `trigger`, Shape/index.js:63-67)
    for (const f of node.flags) {
      const run = obj.dusp.flagFunctions[f.flag]
      if (run) run.call(obj)
    }
  return obj
}

function constructShorthand(node) {
  const args = node.arguments.map((n) => n.n)
  const make = SHORTHAND[node.constructorAlias]
  if (make) return make(...args)
  const C = COMPONENTS[node.constructorAlias]
  if (C) return new C(...args)
  throw 'dusp-hip: unit type not supported on the GPU path: ' + node.constructorAlias
}

function constructOperation(node, index) {
  if (!node.a || !node.b || !node.operator) throw 'could not construct operation'
  const a = construct(node.a, index), b = construct(node.b, index)
  switch (node.operator) {
    case '*': return quick.multiply(a, b)
    case '/': return quick.divide(a, b)
    case '+': return quick.add(a, b)
    case '-': return quick.subtract(a, b)
    case ',': return quick.concat(a, b)
    case '@': return new graph.Pan(a, b)
    case '^': return quick.pow(a, b)
    case '->':
      if (!b.isUnitOrPatch) throw 'unknown use of -> operator'
      if (isSignal(a)) b.defaultInlet.connect(a)
      else b.defaultInlet.setConstant(a)
      return b
    case '|<': return quick.clipBelow(b, a)
    case '>|': return quick.clipAbove(a, b)
    case 'at':
      if (!a.stop || !a.trigger) throw "invalid use of 'at' operator"
      a.stop()
      a.scheduleTrigger(b)
      return a
    case 'for': {
      const unit = typeof a === 'number' ? new graph.Repeater(a) : a
      if (!unit.scheduleFinish) throw "invalid use of 'for' operator. First operand has no scheduleFinish function"
      unit.scheduleFinish(b)
      return unit
    }
    case '!': // regular retrigger (constructOperation.js:69-75)
      if (!a.stop || !a.trigger) throw "invalid use of '!' operator"
      a.trigger()
      new graph.Retriggerer(a, b)
      return a
    case '~!': // sporadic retrigger (constructOperation.js:78-82; unlike `!` it does not trigger first)
      if (!a.stop || !a.trigger) throw "invalide use of '!~' operator"
      new graph.SporadicRetriggerer(a, b)
      return a
    case 'then': { // (constructOperation.js:45-61; `destinations` is never passed there, so always this form) a Repeater that
      // plays `a` until it finishes, then `b`: the finish hook REWIRES the circuit.  The hook runs on the host where the
      // finish is a scheduled event (`a for 2 then b`); renderChannelData's SegmentRenderer then builds a new device program
      // for the rewired circuit (a circuit that holds device-only memory — delay lines, late edges — is refused there).
      const out = new graph.Repeater()
      out.IN = a
      if (a !== null && typeof a === 'object') a.onFinish = () => { out.IN = b }
      else throw 'dusp-hip: `then` needs a unit or patch to finish on its left'
      return out
    }
    default: throw 'Unknown operator: ' + node.operator
  }
}

function unDusp(o) {
  if (o === null) return null
  if (o === undefined) return undefined
  if (typeof o === 'string' || o instanceof String) return construct(String(o))
  if (typeof o === 'number') return o
  if (o.isUnit || o.isOutlet || o.isPatch) return o
  return undefined
}

module.exports = unDusp
module.exports.construct = construct
module.exports.COMPONENTS = COMPONENTS
