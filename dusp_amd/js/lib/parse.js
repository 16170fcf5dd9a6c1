'use strict'
/* Parser for the Dusp string language:  "[Osc f:440] * D0.5",  "O220 -> LP800",  "[Multiply a:[Osc #lfo 3] b:#lfo]" ...
 *
 * The reference's own parser is an un-vendored git submodule (src/parseDSP/, empty in the repository;
 * src/unDusp.js:1, src/construct/*.js require it), so this is an independent implementation of the grammar —
 * pinned against the reference's browserify bundle, which still carries the original modules
 * (demos/browser-dusp-demo-2.bundle.js:6449-7264; tests/golden/str_*.json are produced by running THAT parser).
 * It yields the same syntax tree (node `type`s as src/construct/constructExpression.js:7-35 dispatches on them),
 * including the grammar's oddities:
 *   - a number is the longest run of [0-9.-] fed to parseFloat, so "3-2" is the number 3 and subtraction needs spaces;
 *   - every operator is RIGHT-associative and its rank is its position in OPERATORS: "10 - 3 - 2" is 10-(3-2) and
 *     `+` binds tighter than `-`;
 *   - inside [...] arguments must be separated by whitespace; "word:" / "word=" starts a named attribute, "#id"
 *     names the object, a bare word that is not a shorthand is a flag;
 *   - a shorthand is a constructor alias glued to comma-separated numbers: "O440", "AP0.01,0.5", "Osc220".
 * A node's `length` is the number of characters it spans from where it started.
 */

// rank = index (first occurrence): lower binds tighter.  "->" is listed twice in the reference's table; the
// first position is the one indexOf finds.
const OPERATORS = ['->', 'at', '^', '*', '/', '@', '+', '-', '~!', '!', ',', '->', '>|', '|<', 'for', 'then']

const SHORTHAND_ALIASES = ['O', 'Z', 'Sq', 'A', 'D', 't', 'random', 'LP', 'AP', 'HP']
/* every name the reference would also accept as a shorthand / object constructor (src/components/index.js,
 * src/patches/index.js): names only — whether a name can be BUILT here is decided by lib/unDusp.js */
const REFERENCE_NAMES = ['AHD', 'Abs', 'AllPass', 'CircleBufferNode', 'CircleBufferReader', 'CircleBufferWriter', 'Clip',
  'CombFilter', 'ConcatChannels', 'CrossFader', 'DecibelToScaler', 'Delay', 'Divide', 'Filter', 'FixedDelay',
  'FixedMultiply', 'Gain', 'GreaterThan', 'HardClipAbove', 'HardClipBelow', 'LessThan', 'MidiToFrequency', 'Monitor',
  'MonoDelay', 'Multiply', 'Noise', 'MultiChannelOsc', 'Osc', 'Pan', 'PickChannel', 'PolarityInvert', 'Pow', 'Ramp',
  'ReadBackDelay', 'Repeater', 'Rescale', 'Retriggerer', 'SampleRateRedux', 'SecondsToSamples', 'SemitoneToRatio',
  'Shape', 'SignalCombiner', 'SporadicRetriggerer', 'Subtract', 'Sum', 'Timer', 'VectorMagnitude', 'Augment', 'BinShift',
  'FFT', 'HardHighPass', 'HardLowPass', 'Hopper', 'IFFT', 'ReChunk', 'SpectralGate', 'SpectralSum', 'SpectralUnit',
  'UnHopper', 'Windower', 'CircularMotion', 'LinearMotion',
  'APStack', 'APWeb', 'AttenuationMatrix', 'BandFilter', 'Boop', 'ComplexOrbit', 'DelayMixer', 'FMOsc', 'FMSynth',
  'FrequencyGroup', 'HardBandPass', 'LFO', 'ManyOsc', 'MidiOsc', 'Mixer', 'MultiTapDelay', 'OrbittySine', 'ScaryPatch',
  'SimpleDelay', 'SineBoop', 'SineCloud', 'Space', 'SpaceBoop', 'SpaceChannel', 'StereoDetune', 'StereoOsc', 'Synth',
  'TriggerGroup', 'Worm']
const SHORTHANDS = new Set(SHORTHAND_ALIASES.concat(REFERENCE_NAMES))

const isSpace = (c) => c !== undefined && /\s/.test(c)
const isLetter = (c) => c !== undefined && /[a-zA-Z_]/.test(c)
const isWordChar = (c) => c !== undefined && /[a-zA-Z0-9_]/.test(c)
const isNumberChar = (c) => c !== undefined && /[0-9.\-]/.test(c)

function skipSpace(s, i) { while (i < s.length && isSpace(s[i])) i++; return i }
function run(s, i, test) { let j = i; while (j < s.length && test(s[j])) j++; return s.slice(i, j) }
const word = (s, i) => run(s, i, isLetter) || null

function parseNumber(s, i) {
  const text = run(s, i, isNumberChar)
  return text ? { type: 'number', n: parseFloat(text), length: text.length } : null
}

function parseReference(s, i) {
  if (s[i] !== '#') return null
  const id = run(s, i + 1, isWordChar)
  // an id that runs up to the very end of the input is NOT recognised: the reference's scanner tests the character
  // past the end (`undefined` -> "undefined", which its [a-zA-Z0-9_] accepts) and falls off its loop without a result
  if (!id || i + 1 + id.length >= s.length) return null
  return { type: 'id', id, length: id.length + 1 }
}

function parseString(s, i) {
  const quote = s[i]
  if (quote !== '"' && quote !== "'") return null
  const end = s.indexOf(quote, i + 1)
  if (end < 0) return null
  if (s[end - 1] === '\\') throw 'dusp-hip: an escaped quote inside a Dusp string makes the reference parser loop forever'
  return { type: 'string', string: s.slice(i + 1, end), length: end - i + 1 }
}

/* ---- the JSON-ish literal behind "{": strings, numbers, [a, b], {key: value, flag,} */
function parseJsonValue(s, i) {
  const str = parseString(s, i)
  if (str) return { type: 'json', o: str.string, length: str.length }
  const num = parseNumber(s, i)
  if (num) return { type: 'json', o: num.n, length: num.length }
  return parseJsonArray(s, i) || parseJsonObject(s, i)
}
function parseJsonArray(s, i0) {
  if (s[i0] !== '[') return null
  const items = []
  let i = skipSpace(s, i0 + 1)
  while (i < s.length) {
    if (s[i] === ']') { i++; break }
    const item = parseJsonValue(s, i)
    if (!item) return null
    items.push(item.o)
    i = skipSpace(s, i + item.length)
    if (s[i] === ',') i = skipSpace(s, i + 1)
    else if (s[i] === ']') { i++; break } else return null
  }
  return { type: 'json', o: items, length: i - i0 }
}
function parseJsonObject(s, i0) {
  if (s[i0] !== '{') return null
  const o = {}
  let i = skipSpace(s, i0 + 1)
  while (i < s.length) {
    if (s[i] === '}') { i++; break }
    // key: a bare word, a quoted string or a number
    const w = word(s, i), q = w ? null : parseString(s, i), n = w || q ? null : parseNumber(s, i)
    if (!w && !q && !n) return null
    const keyLength = w ? w.length : (q || n).length
    let j = skipSpace(s, i + keyLength), used
    if (s[j] === ',') { // "{flag, ...}": a key on its own is true
      o[w || (q ? q.string : (n.n || n))] = true
      used = keyLength
    } else {
      if (s[j] !== ':') return null
      j = skipSpace(s, j + 1)
      const value = parseJsonValue(s, j)
      if (!value) return null
      o[w || (q ? q.string : n)] = value.o
      used = j + value.length - i
    }
    i = skipSpace(s, i + used)
    if (s[i] === ',') i = skipSpace(s, i + 1)
    else if (s[i] === '}') { i++; break } else return null
  }
  return { type: 'json', o, length: i - i0 }
}

function parseShorthand(s, i0) {
  const alias = word(s, i0)
  if (!alias || !SHORTHANDS.has(alias)) return null
  const args = []
  let i = i0 + alias.length
  let n = parseNumber(s, i)
  if (n) {
    args.push(n); i += n.length
    while (s[i] === ',') { // a comma glued to the number MUST be followed by another number
      n = parseNumber(s, i + 1)
      if (!n) return null
      args.push(n); i += 1 + n.length
    }
  }
  return { type: 'shorthand', constructorAlias: alias, arguments: args, length: i - i0 }
}

/* "[Constructor arg arg name:value #id flag]" */
function parseObject(s, i0) {
  if (s[i0] !== '[') return null
  const at = skipSpace(s, i0 + 1)
  const ctor = word(s, at)
  if (!ctor) return null
  // NB `constructor` is the field name the reference's tree uses (constructObject.js:17)
  const node = { type: 'object', constructor: ctor, arguments: [], flags: [], attributes: [] }
  let i = at + ctor.length
  while (i < s.length) {
    if (s[i] === ']') { node.length = i - i0 + 1; return node }
    if (!isSpace(s[i])) return null // arguments are separated by whitespace
    i = skipSpace(s, i)
    if (i >= s.length) return null
    if (s[i] === ']') { node.length = i - i0 + 1; return node }
    const arg = parseArgument(s, i)
    if (!arg) return null
    if (arg.type === 'id') node.id = arg.id
    else if (arg.type === 'attribute') node.attributes.push(arg)
    else if (arg.type === 'unnamedArgument') node.arguments.push(arg)
    else node.flags.push(arg)
    i += arg.length
  }
  return null
}

function parseArgument(s, i) {
  const ref = parseReference(s, i)
  if (ref) return ref
  // name: value   /   name = value
  const name = word(s, i)
  if (name) {
    const sep = skipSpace(s, i + name.length)
    if (s[sep] === '=' || s[sep] === ':') {
      const from = skipSpace(s, sep + 1)
      const value = parseExpression(s, from)
      if (value) return { type: 'attribute', property: name, value, length: from - i + value.length }
    }
  }
  const value = parseExpression(s, i)
  if (value) return { type: 'unnamedArgument', value, length: value.length }
  return name ? { type: 'flag', flag: name, length: name.length } : null
}

/* an object, a #reference or a shorthand, optionally followed by ".property" */
function parseObjectOrProperty(s, i0) {
  const object = parseObject(s, i0) || parseReference(s, i0) || parseShorthand(s, i0)
  if (!object) return null
  const dot = skipSpace(s, i0 + object.length)
  if (s[dot] === '.') {
    const at = skipSpace(s, dot + 1)
    const property = word(s, at)
    if (property) return { type: 'objectProperty', property, object, length: at - i0 + property.length }
  }
  return object
}

function parseSimple(s, i0) {
  if (s[i0] === '{') return parseJsonValue(s, i0)
  if (s[i0] === '(') {
    const from = skipSpace(s, i0 + 1)
    const inner = parseExpression(s, from)
    if (!inner) return null
    const close = skipSpace(s, from + inner.length)
    if (s[close] !== ')') return null
    inner.length = close + 1 - i0
    inner.bracketed = true
    return inner
  }
  return parseReference(s, i0) || parseNumber(s, i0) || parseObjectOrProperty(s, i0) || parseString(s, i0)
}

function operatorAt(s, i) { // the longest operator spelled at position i
  let best = null
  for (const op of OPERATORS) if (s.startsWith(op, i) && (!best || op.length > best.length)) best = op
  return best
}

/* simple (operator simple)*, folded by rank; every operator associates to the right */
function parseExpression(s, i0 = 0) {
  const first = parseSimple(s, i0)
  if (!first) return null
  const operands = [first], ops = []
  let end = i0 + first.length
  for (;;) {
    const at = skipSpace(s, end)
    const op = operatorAt(s, at)
    if (!op) break
    const from = skipSpace(s, at + op.length)
    const operand = parseSimple(s, from)
    if (!operand) break
    ops.push(op); operands.push(operand)
    end = from + operand.length
  }
  delete first.length // (the operands after an operator keep theirs, as in the reference's tree)
  let k = 0 // next unread operator
  const fold = (maxRank) => {
    let left = operands[k]
    while (k < ops.length && OPERATORS.indexOf(ops[k]) <= maxRank) {
      const operator = ops[k++]
      const right = fold(OPERATORS.indexOf(operator)) // same rank continues on the right: right-associative
      left = { type: 'operation', operator, a: left, b: right, bindingOrder: OPERATORS.indexOf(operator) }
    }
    return left
  }
  const tree = fold(Infinity)
  tree.length = end - i0
  return tree
}

module.exports = { parseExpression, parseObject, parseNumber, parseShorthand, parseReference, parseString, OPERATORS, SHORTHAND_ALIASES,
  REFERENCE_NAMES }
