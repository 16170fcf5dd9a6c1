'use strict'
/* RIFF/WAVE encoding of rendered PCM (SURVEY.md §8f-4).  The reference itself has no file writer: its sinks take
 * RenderStream's frames (`format`: 32-bit LE floats, src/RenderStream.js:63-68).  encodeWav keeps exactly those
 * frames — WAVE_FORMAT_IEEE_FLOAT (3) — or, with bitDepth 16, rounds them to signed PCM (1) the usual way
 * (clamp to [-1, 1], scale by 32767, round half away from zero).
 *
 *   encodeWav(channelData [, { bitDepth: 32 | 16, sampleRate }]) -> Buffer
 *   channelData: what renderChannelData resolves to (array of per-channel typed arrays with .sampleRate), or
 *                { frames: Float32Array, numberOfChannels, sampleRate } for already interleaved frames. */
function encodeWav(channelData, { bitDepth = 32, sampleRate } = {}) {
  if (bitDepth !== 32 && bitDepth !== 16) throw 'dusp-hip: WAV bitDepth must be 32 (float) or 16 (PCM)'
  let frames, nChannels
  if (channelData && channelData.frames) {
    frames = channelData.frames; nChannels = channelData.numberOfChannels; sampleRate = sampleRate || channelData.sampleRate
  } else {
    nChannels = channelData.length
    if (!nChannels) throw 'dusp-hip: nothing to encode (no channels)'
    sampleRate = sampleRate || channelData.sampleRate
    const n = channelData[0].length
    frames = new Float32Array(n * nChannels)
    for (let c = 0; c < nChannels; c++) { const ch = channelData[c]; for (let t = 0; t < n; t++) frames[t * nChannels + c] = ch[t] }
  }
  if (!(sampleRate > 0)) throw 'dusp-hip: WAV needs a sample rate'
  const bytes = bitDepth / 8, dataBytes = frames.length * bytes
  const float = bitDepth === 32
  const fmtBytes = float ? 18 : 16 // non-PCM formats carry cbSize and need a fact chunk
  const header = 12 + 8 + fmtBytes + (float ? 12 : 0) + 8
  const out = Buffer.alloc(header + dataBytes + (dataBytes & 1))
  let p = 0
  const tag = (s) => { out.write(s, p, 'ascii'); p += 4 }
  const u32 = (v) => { out.writeUInt32LE(v >>> 0, p); p += 4 }
  const u16 = (v) => { out.writeUInt16LE(v, p); p += 2 }
  tag('RIFF'); u32(out.length - 8); tag('WAVE')
  tag('fmt '); u32(fmtBytes); u16(float ? 3 : 1); u16(nChannels); u32(sampleRate); u32(sampleRate * nChannels * bytes); u16(nChannels * bytes); u16(bitDepth)
  if (float) { u16(0); tag('fact'); u32(4); u32(frames.length / nChannels) }
  tag('data'); u32(dataBytes)
  if (float) for (let i = 0; i < frames.length; i++, p += 4) out.writeFloatLE(frames[i], p)
  else for (let i = 0; i < frames.length; i++, p += 2) {
    const v = Math.max(-1, Math.min(1, frames[i])) * 32767
    out.writeInt16LE(v < 0 ? -Math.round(-v) : Math.round(v) || 0, p)
  }
  return out
}

/* the inverse, for tests and round trips: -> { sampleRate, numberOfChannels, bitDepth, format, channelData: Float32Array[] } */
function decodeWav(buf) {
  if (buf.toString('ascii', 0, 4) !== 'RIFF' || buf.toString('ascii', 8, 12) !== 'WAVE') throw 'dusp-hip: not a RIFF/WAVE file'
  let p = 12, fmt = null, data = null
  while (p + 8 <= buf.length) {
    const id = buf.toString('ascii', p, p + 4), size = buf.readUInt32LE(p + 4)
    if (id === 'fmt ') fmt = { format: buf.readUInt16LE(p + 8), numberOfChannels: buf.readUInt16LE(p + 10), sampleRate: buf.readUInt32LE(p + 12), bitDepth: buf.readUInt16LE(p + 22) }
    else if (id === 'data') data = buf.slice(p + 8, p + 8 + size)
    p += 8 + size + (size & 1)
  }
  if (!fmt || !data) throw 'dusp-hip: WAV without fmt / data chunk'
  const bytes = fmt.bitDepth / 8, nFrames = data.length / bytes / fmt.numberOfChannels
  const channelData = []
  for (let c = 0; c < fmt.numberOfChannels; c++) channelData.push(new Float32Array(nFrames))
  for (let t = 0; t < nFrames; t++)
    for (let c = 0; c < fmt.numberOfChannels; c++) {
      const at = (t * fmt.numberOfChannels + c) * bytes
      channelData[c][t] = fmt.format === 3 ? data.readFloatLE(at) : data.readInt16LE(at) / 32767
    }
  channelData.sampleRate = fmt.sampleRate
  return Object.assign(fmt, { channelData })
}

module.exports = { encodeWav, decodeWav }
