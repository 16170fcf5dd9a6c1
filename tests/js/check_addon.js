'use strict'
/* Addon smoke (no GPU needed): it loads, reports the ABI, and without a device every entry point
 * fails with a STRING (the reference's error convention) instead of falling back to a CPU path. */
const lib = require('../../dusp_amd/js')
const native = require('../../dusp_amd/js/lib/native')()
const out = { version: native.version(), abi: native.abiVersion(), exports: Object.keys(native).sort() }
let threw = null
try { native.ctxCreate(-1); out.gpu = true } catch (e) { threw = e; out.gpu = false }
out.ctxErrorIsString = threw === null ? null : typeof threw === 'string'
lib.renderChannelData(null).then(() => { out.nullRejects = false }, (e) => { out.nullRejects = e }).then(() => {
  return lib.renderChannelData(new lib.Osc(440), 0.01).then((cd) => { out.rendered = cd.length }, (e) => { out.renderError = e })
}).then(() => { console.log(JSON.stringify(out)) })
