'use strict'
/* Strings of the Dusp language used to pin the string front-end (dusp_amd/js/lib/parse.js, unDusp.js) against the
 * reference's own parser and constructors (oracle/js/gen_golden_strings.js).
 *   syntax  parsed only: the tree must equal the reference parser's, node for node (also for rejected input: null)
 *   graphs  parsed, built, extracted and rendered: descriptor and PCM must equal the reference's */
const syntax = [
  '440', '-3.5', '3-2', '10 - 3 - 2', '10 - 3 + 2', '2 * 3 + 4', '2 + 3 * 4', '2 * 3 * 4 + 5', '1 + 2 * 3 - 4', '1 - 2 * 3 ^ 4 / 5',
  '(1 + 2) * 3', '( 1+2 )', '1 , 2 , 3', '1 -> 2', '#abc', '#a1_b', '# a', '[Osc]', '[Osc 440]', '[ Osc   f:440 ]', '[Osc f=440 #myosc]',
  '[Osc440]', '[Osc 440', '[Osc f: [Osc 2] * 10 + 440]', '[Multiply [Osc 440] [Ramp 48000 1 0 trigger]]', '[Shape "decay" 0.5 trigger]',
  "[Osc waveform:'saw' f:110]", 'O440', 'O440.5', 'Z110 * D1', 'Sq55', 'AP0.01,0.5', 'AP0.01, 0.5', 'O440,', 'Osc220', 'LP500', 't', 'random',
  'O440 .phase', '[Osc 1].OUT', '#x.OUT', 'O1 then O2', 'D1 at 0.5', 'O1 for 2', 'O1 ! 2', 'O1 ~! 2', 'O1 @ 0.5', 'O1 >| 0.5', 'O1 |< 0.5',
  '{a: 1, b: "two", c: [1, 2, 3], d}', '{flag, x: 1}', '{}', '[Osc {f: 1}]', '"hello"', "'single'", 'Hz', 'attack', '', '   ', ')', '*', '1 +',
  '1 * (2', '[Osc f:]', '[Osc f:440]]', 'O1 attack', 'O1 -> LP200 -> HP100', '- 1', '1 - -1', '.5 * .', 'O-440', '[Sum a:1 b:2].out',
]

const graphs = [
  { name: 'osc', text: '[Osc 440]', duration: 0.05 },
  { name: 'ramp_literal', text: '[Multiply A:[Osc f:[Ramp 200 100 2]] B:[Osc 3]]', duration: 0.1 }, // configs[1] as BASELINE spells it
  { name: 'shorthand_voice', text: 'O440 * D0.05', duration: 0.06 },
  { name: 'saw_lowpass', text: 'Z110 -> LP800', duration: 0.05 },
  { name: 'square_highpass', text: 'Sq55 -> HP2000', duration: 0.05 },
  { name: 'fm', text: '[Osc f:[Osc 5] * 100 + 440]', duration: 0.05 },
  { name: 'precedence', text: 'O100 * 0.5 + O200 * 0.25 - O300 * 0.125', duration: 0.02 },
  { name: 'right_assoc', text: 'O100 - O200 - O300', duration: 0.02 },
  { name: 'brackets', text: '(O100 + O150) * (O2 + 1.5)', duration: 0.05 },
  { name: 'reference', text: '[Multiply a:[Osc #lfo 3] b:[Sum a:#lfo b:1]]', duration: 0.1 },
  { name: 'reference_undeclared', text: '[Osc #x 3] * #x', duration: 0.01 }, // the reference throws: operands do not share an index
  { name: 'positional_reference', text: '[Multiply [Osc #x 3] #x]', duration: 0.01 }, // ... nor do positional arguments
  { name: 'concat_pan', text: '[Osc f:(220 , 330.5)] * 0.5', duration: 0.02 },
  { name: 'pan', text: 'O440 @ O1', duration: 0.05 },
  { name: 'pow_clip', text: '(O100 ^ 2) >| 0.5', duration: 0.02 },
  { name: 'clip_below', text: '-0.25 |< Z200', duration: 0.02 },
  { name: 'divide', text: 'O300 / (O2 + 1.5)', duration: 0.02 },
  { name: 'waveform_attribute', text: "[Osc waveform:'triangle' f:330.5]", duration: 0.02 },
  { name: 'shape_flag', text: '[Multiply a:O220 b:[Shape "semiSine" 0.04 trigger]]', duration: 0.05 },
  { name: 'allpass', text: 'Sq100 -> AP0.0021,0.6', duration: 0.05 },
  { name: 'comb_object', text: 'Z150 -> [CombFilter 0.004 0.7]', duration: 0.05 },
  { name: 'timer', text: '[Osc f:t * 4000]', duration: 0.05 },
  { name: 'component_shorthand', text: 'Osc220 * Ramp4800', duration: 0.02 }, // any component name works as a shorthand
  { name: 'at_trigger', text: 'O330 * (D0.02 at 0.03)', duration: 0.08, events: true },
  { name: 'for_finish', text: 'O330 for 0.02', duration: 0.03, events: true },
  { name: 'then_two_tones', text: 'O330 for 0.02 then Z220.5 * 0.5', duration: 0.05, events: true }, // the finish hook rewires the circuit: `then`
  { name: 'then_thrice', text: '(O330 * D0.01) for 0.011 then (Sq110 -> LP900) for 0.03 then O55.5', duration: 0.06, events: true },
  { name: 'delay_attribute', text: '[Delay in:O500 delay:300.5]', duration: 0.03 },
  { name: 'semitone', text: '[Osc f:[SemitoneToRatio in:O4 * 12] * 220]', duration: 0.05 },
  { name: 'retrigger', text: '(D0.02 ! 20) * O440', duration: 0.2, events: true }, // `!`: Retriggerer, ticked on the host
  // 25 FM voices added up by `+` (a right-deep chain of Sums, 124 units): on the GPU the voices run in a loop (jit_codegen.hpp VoicePlan)
  { name: 'additive_fm', text: Array.from({ length: 25 }, (_, j) => '[Osc f:[Osc ' + (3 + j) + '] * 40 + ' + (220 + 11.5 * j) + ']').join(' + '), duration: 0.05 },
  { name: 'number_only', text: '2 * 3 + 4', duration: 0.01 }, // not a graph: unDusp returns 10 and renderChannelData rejects it
  { name: 'unknown', text: '[Foo 1]', duration: 0.01 },
  { name: 'garbage', text: ']] nothing [[', duration: 0.01 },
]

/* token soup: deterministic pseudo-random strings over the language's alphabet (no backslash: an escaped quote sends the
 * reference's string scanner into an endless loop), so that odd juxtapositions are compared with the reference too */
const TOKENS = ['[', ']', '(', ')', '{', '}', ' ', ' ', ' ', '#', ':', '=', ',', '.', '-', '->', '*', '/', '+', '^', '@', '!', '~!', '>|', '|<',
  'at', 'for', 'then', 'O', 'Z', 'Sq', 'D', 'A', 't', 'LP', 'AP', 'Osc', 'Sum', 'Multiply', 'Ramp', 'Shape', 'f', 'a', 'b', 'x1', 'trigger',
  '440', '0.5', '-1', '3', '.', '"', "'", 'decay', '_']
let seed = 12345
const rnd = (n) => { seed = (seed * 1103515245 + 12345) % 2147483648; return Math.floor(seed / 2147483648 * n) }
for (let k = 0; k < 400; k++) {
  let text = ''
  const len = 1 + rnd(12)
  for (let j = 0; j < len; j++) text += TOKENS[rnd(TOKENS.length)]
  syntax.push(text)
}

/* grammar-shaped random expressions (mostly valid): nesting, attributes, ids, flags, every operator, odd spacing */
const pick = (xs) => xs[rnd(xs.length)]
const sp = () => pick(['', ' ', ' ', '  '])
function genSimple(depth) {
  const k = rnd(depth > 2 ? 4 : 8)
  if (k === 0) return pick(['440', '0.5', '-2', '3.25', '.5', '100'])
  if (k === 1) return pick(['O', 'Z', 'Sq', 'D', 'A', 'LP', 'HP', 'Osc', 'Ramp']) + pick(['', '440', '0.5', '2,3', '1,2,3'])
  if (k === 2) return pick(['#a', '#lfo', '#x1'])
  if (k === 3) return pick(['"saw"', "'decay'", 't', 'random'])
  if (k === 4) return '(' + sp() + genExpr(depth + 1) + sp() + ')'
  if (k === 5) return genSimple(depth + 1) + pick(['.OUT', ' . phase', '.f'])
  let o = '[' + sp() + pick(['Osc', 'Sum', 'Multiply', 'Ramp', 'Shape', 'Filter', 'Delay', 'Foo'])
  const n = rnd(4)
  for (let j = 0; j < n; j++) {
    const a = rnd(5)
    o += pick([' ', ' ', '  ', '']) // a missing separator makes the whole object unparsable
    if (a === 0) o += pick(['#a', '#lfo', '#x1'])
    else if (a === 1) o += pick(['f', 'a', 'b', 'waveform', 'in']) + sp() + pick([':', '=']) + sp() + genExpr(depth + 1)
    else if (a === 2) o += pick(['trigger', 'loud', 'x'])
    else o += genExpr(depth + 1)
  }
  return o + sp() + ']'
}
function genExpr(depth) {
  let e = genSimple(depth)
  const n = depth > 2 ? rnd(2) : rnd(4)
  for (let j = 0; j < n; j++)
    e += sp() + pick(['*', '/', '+', '-', '^', ',', '@', '->', '>|', '|<', 'at', 'for', 'then', '!', '~!']) + sp() + genSimple(depth + 1)
  return e
}
for (let k = 0; k < 300; k++) syntax.push(genExpr(0))

module.exports = { syntax, graphs }
