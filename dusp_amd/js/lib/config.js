'use strict'
/* Process-wide configuration.  Like the reference (src/config.js:1-17) the defaults can be overridden
 * from the command line of the hosting script: `node app.js --sampleRate=48000`.  They can also be set
 * programmatically with configure() BEFORE any graph is built. */
const config = { standardChunkSize: 256, sampleRate: 44100, channelFormat: 'stereo', useDuspShorthands: true }

for (const arg of process.argv.slice(2)) {
  const m = /^--(sampleRate|standardChunkSize)(?:=(.*))?$/.exec(arg)
  if (m && m[2] !== undefined && m[2] !== '' && !isNaN(Number(m[2]))) config[m[1]] = Number(m[2])
}

config.configure = function (opts) {
  if (opts && opts.sampleRate !== undefined) config.sampleRate = Number(opts.sampleRate)
  return config
}

module.exports = config
