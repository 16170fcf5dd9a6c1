'use strict'
/* Wall-clock of single-circuit renders through the JS surface (what one renderChannelData(unDusp(text), T) costs): the first
 * render of each circuit (a structure this process has not seen: by default it does not wait for the circuit compiler and runs
 * on the interpreter kernel) and a second one (the compiled kernel, once the background compile is done).
 *   node tools/single_circuit.js --sampleRate=48000 [--seconds=10]        (DUSP_WAVE_JIT=2: always wait for the compiler) */
const lib = require('../dusp_amd/js')
const argv = require('minimist')(process.argv.slice(2))
const seconds = argv.seconds || 10
const texts = ['O440', 'O440 * D1', '[Osc f:[Osc 5] * 100 + 440] * D2', 'Z110 -> LP800', 'Z150 -> AP0.0021,0.6', 'Z150 -> [CombFilter 0.004 0.7]',
  '[Delay in:O500 delay:300.5]', '[Delay in:O500 delay:30.5]', 'O440 @ O1', '(D0.02 ! 20) * O440', '[MultiChannelOsc f:(220 , 330)]',
  'Sq100 -> AP0.0021,0.6 -> AP0.0013,0.45 -> AP0.0007,0.3', '[Osc f:t * 400 + 100]', '[Multiply A:[Osc f:[Ramp 96000 200 100 trigger]] B:O3]']
async function main() {
  await lib.renderChannelData(new lib.Osc(440), 0.1)
  const time = async (text) => {
    const t0 = process.hrtime.bigint()
    const cd = await lib.renderChannelData(lib.unDusp(text), seconds)
    return [Number(process.hrtime.bigint() - t0) / 1e6, cd.length]
  }
  const first = []
  for (const text of texts) first.push(await time(text))
  await new Promise((resolve) => setTimeout(resolve, 3000)) // (background compiles finish)
  for (let k = 0; k < texts.length; k++) {
    const [ms2] = await time(texts[k])
    const [ms, ch] = first[k]
    console.log(`first ${ms.toFixed(1).padStart(8)} ms  again ${ms2.toFixed(1).padStart(8)} ms  ${(seconds * 1000 / ms2).toFixed(0).padStart(6)}x realtime  ${ch} ch  ${texts[k]}`)
  }
}
main().catch((e) => { console.error(e); process.exit(1) })
