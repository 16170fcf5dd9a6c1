'use strict'
/* Loader for the N-API addon (addon/dusp_napi.c), which binds the C ABI of include/dusp_hip.h.
 * There is no JavaScript fallback renderer: if the addon or the GPU is missing, calls throw / reject. */
const path = require('path')

let addon = null
function native() {
  if (addon) return addon
  const file = path.join(__dirname, '..', 'addon', 'dusp_napi.node')
  try {
    addon = require(file)
  } catch (e) {
    throw 'dusp-hip: native addon not available (' + file + '): ' + (e && e.message ? e.message : e) +
      ' — build it with `make -C dusp_amd/js/addon`; there is no CPU fallback'
  }
  return addon
}

module.exports = native
