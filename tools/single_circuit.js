'use strict'
/* Wall-clock of single-circuit renders through the JS surface (what one renderChannelData(unDusp(text), T) costs).
 *   node tools/single_circuit.js --sampleRate=48000 [--seconds=10] */
const lib = require('../dusp_amd/js')
const argv = require('minimist')(process.argv.slice(2))
const seconds = argv.seconds || 10
const texts = ['O440', 'O440 * D1', '[Osc f:[Osc 5] * 100 + 440] * D2', 'Z110 -> LP800', 'Z150 -> AP0.0021,0.6', 'Z150 -> [CombFilter 0.004 0.7]',
  '[Delay in:O500 delay:300.5]', '[Delay in:O500 delay:30.5]', 'O440 @ O1', '(D0.02 ! 20) * O440', '[MultiChannelOsc f:(220 , 330)]',
  'Sq100 -> AP0.0021,0.6 -> AP0.0013,0.45 -> AP0.0007,0.3', '[Osc f:t * 400 + 100]',
  // patches (lib/patches.js)
  '[Mixer O220 Z330 Sq441]', '[StereoOsc p:60 pan:0.3]', '[SimpleDelay in:Z110 delay:0.25 feedback:0.5]', '[Space in:O440 placement:(O0.5 , 1)]',
  '[SpaceBoop p:64 duration:2 trigger]', 'Z110 -> [BandFilter fLow:300 fHigh:2000]', 'Z200 -> [APStack 4 0.01 0.5]']
async function main() {
  await lib.renderChannelData(new lib.Osc(440), 0.1)
  for (const text of texts) {
    const t0 = process.hrtime.bigint()
    const cd = await lib.renderChannelData(lib.unDusp(text), seconds)
    const ms = Number(process.hrtime.bigint() - t0) / 1e6
    console.log(`${ms.toFixed(1).padStart(8)} ms  ${(seconds * 1000 / ms).toFixed(0).padStart(6)}x realtime  ${cd.length} ch  ${text}`)
  }
}
main().catch((e) => { console.error(e); process.exit(1) })
