'use strict'
/* TEST INFRASTRUCTURE — golden-vector generator.  Runs ONLY in the build
 * container (it imports the reference from /root/reference, which does not
 * exist on the GPU box).  Usage:
 *
 *   node oracle/js/gen_golden.js --sampleRate=48000 [--only=name,...] [--out=tests/golden]
 *
 * `--sampleRate` is consumed by the reference's own src/config.js:1,17 (minimist
 * over process.argv) — that is how the reference selects 48 kHz.
 *
 * For every case it builds the graph PROGRAMMATICALLY with the reference's
 * constructors (the string front-end is an un-vendored submodule, SURVEY.md
 * §8c), renders it with the reference's own renderChannelData, and writes
 *   <name>.desc.f64   descriptor words produced by dusp_amd/js/lib/extract.js
 *                     from the live reference objects (little-endian f64)
 *   <name>.pcm.f32    the reference's PCM: the listed windows of every channel
 *   <name>.json       sizes, windows, sha256 of the FULL pcm, reference unit order
 * Only data leaves this script: inputs (descriptor) and expected outputs (PCM).
 */
const fs = require('fs')
const path = require('path')
const crypto = require('crypto')
const { ref } = require('./refload')
const { extract } = require('../../dusp_amd/js/lib/extract')

const argv = require('minimist')(process.argv.slice(2))
const OUT = path.resolve(argv.out || path.join(__dirname, '../../tests/golden'))
const ONLY = argv.only ? String(argv.only).split(',') : null

const config = ref('config.js')
const SR = config.sampleRate
const renderChannelData = ref('renderChannelData.js')
const Osc = ref('components/Osc')
const Ramp = ref('components/Ramp.js')
const Multiply = ref('components/Multiply.js')
const Sum = ref('components/Sum.js')
const Filter = ref('components/Filter.js')
const Delay = ref('components/Delay.js')
const Repeater = ref('components/Repeater.js')
const CircleBuffer = ref('CircleBuffer.js')
const CircleBufferReader = ref('components/CircleBufferReader.js')
const CircleBufferWriter = ref('components/CircleBufferWriter.js')
const quick = ref('quick.js')
const refDusp = ref('dusp.js')
const normaliseLabels = require('../../tests/js/labels')
const more = {}
for (const n of ['Subtract', 'Divide', 'PolarityInvert', 'Abs', 'Clip', 'HardClipAbove', 'HardClipBelow', 'SecondsToSamples',
  'FixedMultiply', 'Gain', 'DecibelToScaler', 'SemitoneToRatio', 'Pow', 'FixedDelay', 'CombFilter', 'AllPass', 'MonoDelay',
  'ReadBackDelay', 'Pan', 'MidiToFrequency', 'Rescale', 'CrossFader', 'VectorMagnitude', 'Timer', 'SampleRateRedux',
  'ConcatChannels', 'PickChannel', 'Retriggerer']) more[n] = ref('components/' + n + '.js')
more.MultiChannelOsc = ref('components/Osc/MultiChannelOsc.js')
more.SporadicRetriggerer = ref('components/SporadicRetrigger.js')
more.Noise = ref('components/Noise.js')
more.Shape = ref('components/Shape')
more.AHD = ref('components/AHD.js')
const patches = {}
for (const n of ['Mixer', 'SimpleDelay', 'StereoOsc', 'LFO', 'MidiOsc', 'BandFilter', 'MultiTapDelay', 'DelayMixer', 'TriggerGroup',
  'Synth', 'SpaceChannel', 'Space', 'ScaryPatch', 'Boop', 'SineBoop', 'SpaceBoop', 'FMOsc', 'ManyOsc', 'StereoDetune',
  'FrequencyGroup', 'AttenuationMatrix', 'APStack', 'APWeb', 'Worm']) patches[n] = ref('patches/' + n + '.js')
const shapeTables = ref('components/Shape/shapeTables.js')
const waveTables = ref('components/Osc/waveTables.js')

const goldenCases = require('../../tests/js/cases')
const cases = goldenCases({ Osc, Ramp, Multiply, Sum, Filter, Delay, Repeater, CircleBuffer,
  CircleBufferReader, CircleBufferWriter, quick, ...more, patches }, SR)
const S = (name) => (SR === 48000 ? name : name + '_sr' + SR)

async function main() {
  fs.mkdirSync(OUT, { recursive: true })
  const index = [], eventIndex = [], hostIndex = []
  for (const c of cases) {
    if (ONLY && !ONLY.includes(c.name)) continue
    const [target, ex, order, text, cd] = await goldenCases.withSeed(c.seed, async () => {
      const target = c.build()
      const ex = extract(target, { allowEvents: true, allowFinishHooks: true }) // before rendering: captures the initial state
      const order = ex.circuit.units.map((u) => u.label + ':' + u.processIndex)
      const text = normaliseLabels(refDusp(target.isPatch ? target.defaultOutlet : target)) // the reference's own stringifier, before any tick
      return [target, ex, order, text, await renderChannelData(target, c.duration)]
    })
    const n = cd[0].length
    const windows = (c.windows || [[0, n]]).map(([a, len]) => [a, Math.min(len, n - a)])
    const h = crypto.createHash('sha256')
    for (const ch of cd) h.update(Buffer.from(ch.buffer, ch.byteOffset, ch.byteLength))
    const parts = []
    for (const ch of cd)
      for (const [a, len] of windows) parts.push(Buffer.from(ch.buffer, ch.byteOffset + 4 * a, 4 * len))
    fs.writeFileSync(path.join(OUT, c.name + '.pcm.f32'), Buffer.concat(parts))
    fs.writeFileSync(path.join(OUT, c.name + '.desc.f64'),
      Buffer.from(ex.words.buffer, ex.words.byteOffset, ex.words.byteLength))
    const meta = { name: c.name, sample_rate: SR, chunk_size: ex.chunkSize, duration: c.duration,
      n_samples: n, n_channels: cd.length, windows, sha256_full: h.digest('hex'), dusp: text,
      reference_unit_order: order.length <= 64 ? order : order.slice(0, 8).concat(['...' + order.length + ' units']) }
    fs.writeFileSync(path.join(OUT, c.name + '.json'), JSON.stringify(meta, null, 1) + '\n')
    // ev_: scheduled events (both hosts); rt_ and ev_patch_: features only the JS host has (host-ticked units, patches)
    ;(c.name.startsWith('rt_') || c.name.startsWith('ev_patch_') ? hostIndex : c.name.startsWith('ev_') ? eventIndex : index).push(c.name)
    console.log(c.name, 'n=' + n, 'ch=' + cd.length, 'units=' + order.length)
  }
  // wave tables: hashes of all five, plus the few entries the docs quote (SURVEY.md §8c)
  if (!ONLY || SR !== 48000) {
    const tables = {}
    for (const w of ['sin', 'saw', 'square', 'triangle', '8bit']) {
      const t = waveTables[w]
      tables[w] = { length: t.length,
        sha256: crypto.createHash('sha256').update(Buffer.from(t.buffer, t.byteOffset, t.byteLength)).digest('hex'),
        head: Array.from(t.slice(0, 4)), tail: Array.from(t.slice(t.length - 2)) }
    }
    for (const w of ['decay', 'attack', 'semiSine', 'decaySquared']) { // Shape's tables (shapeTables.js:21-38)
      const t = shapeTables[w].data
      tables[w] = { length: t.length,
        sha256: crypto.createHash('sha256').update(Buffer.from(t.buffer, t.byteOffset, t.byteLength)).digest('hex'),
        head: Array.from(t.slice(0, 4)), tail: Array.from(t.slice(t.length - 2)) }
    }
    fs.writeFileSync(path.join(OUT, S('wavetables') + '.json'), JSON.stringify({ sample_rate: SR, tables }, null, 1) + '\n')
    fs.writeFileSync(path.join(OUT, S('index') + '.json'), JSON.stringify(index, null, 1) + '\n')
    if (eventIndex.length) fs.writeFileSync(path.join(OUT, S('index_events') + '.json'), JSON.stringify(eventIndex, null, 1) + '\n')
    if (hostIndex.length) fs.writeFileSync(path.join(OUT, S('index_host') + '.json'), JSON.stringify(hostIndex, null, 1) + '\n')
  }
}
main().catch((e) => { console.error('gen_golden failed:', e); process.exit(1) })
