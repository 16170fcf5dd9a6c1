'use strict'
/* RenderStream(outlet, numberOfChannels = 1): a Readable (object mode) that emits one Float32Array of interleaved
 * frames per 256-sample chunk — `buffer[t * numberOfChannels + c]`, i.e. 32-bit little-endian float PCM — for as
 * long as it is read (reference src/RenderStream.js:6-80; `format` is what the reference hands to a speaker / file
 * sink).  Like the reference it auto-normalises: a running gain that shrinks whenever a scaled sample would exceed
 * +-1 (RenderStream.js:38-49, visiting channel by channel within each chunk), and it refuses NaN.
 *
 * The circuit is rendered on the GPU in blocks of `blockChunks` chunks by ONE resumable device program that is
 * continued from block to block (lib/renderChannelData.js SegmentRenderer), so an endless stream costs one kernel
 * launch per block, keeps delay lines on the device and honours scheduled events.  The normalising pass is the
 * reference's sequential arithmetic, done on the host while the frames are laid out. */
const { Readable } = require('stream')
const { SegmentRenderer } = require('./renderChannelData')
const { toOutlet } = require('./extract')

class RenderStream extends Readable {
  constructor(outlet, numberOfChannels = 1, { blockChunks = 64, engine = 0 } = {}) {
    super({ objectMode: true })
    if (!outlet) throw 'RenderStream requires an outlet argument'
    if (outlet.isUnitOrPatch) outlet = outlet.defaultOutlet
    if (!outlet || !outlet.isOutlet) throw 'RenderStream expects an outlet'
    this.numberOfChannels = numberOfChannels
    this.outlet = toOutlet(outlet)
    this.sampleRate = outlet.sampleRate
    this.normaliseFactor = 1
    this.blockChunks = blockChunks
    this.renderer = new SegmentRenderer(outlet, { engine, resumable: true })
    this.circuit = this.renderer.circuit
    this.format = { channels: this.numberOfChannels, bitDepth: 32, sampleRate: this.sampleRate, endianness: 'LE' }
    this.busy = false
    this.stopped = false
  }

  _read() {
    if (this.busy || this.stopped) return
    this.busy = true
    const chunk = this.renderer.chunk, nSamples = this.blockChunks * chunk
    this.renderer.next(nSamples).then(({ pcm, nChannels }) => {
      let wantMore = true
      for (let k = 0; k < this.blockChunks && !this.stopped; k++) {
        const buffer = new Float32Array(this.numberOfChannels * chunk)
        for (let c = 0; c < this.numberOfChannels; c++) {
          if (c >= nChannels) throw new TypeError("Cannot read property '0' of undefined") // the reference indexes channelData[c] blindly
          const at = c * nSamples + k * chunk
          for (let t = 0; t < chunk; t++) {
            let val = pcm[at + t] * this.normaliseFactor
            if (Math.abs(val) > 1) { // digital clipping: shrink the gain for good
              const sf = Math.abs(1 / val)
              val *= sf
              this.normaliseFactor *= sf
            }
            if (isNaN(val)) throw "can't record NaN"
            buffer[t * this.numberOfChannels + c] = val
          }
        }
        wantMore = this.push(buffer)
      }
      this.busy = false
      if (wantMore && !this.stopped) this._read()
    }).catch((e) => { this.busy = false; this.destroy(typeof e === 'string' ? new Error(e) : e) })
  }

  stop() {
    this.stopped = true
    this.push(null)
    this.renderer.close()
  }
}

module.exports = RenderStream
