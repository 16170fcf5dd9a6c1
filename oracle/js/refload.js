'use strict'
/* TEST INFRASTRUCTURE — runs only in the build container, never on the GPU box.
 *
 * Loads the reference's own `src/` modules from /root/reference by absolute
 * path.  Three third-party packages the reference `require`s (compute-gcd,
 * promise, audio-buffer; package.json:22-31) are not installed anywhere in this
 * image.  Instead of writing stand-ins, this loader serves them from the
 * reference's OWN browserify bundle, demos/browser-dusp-demo-2.bundle.js, which
 * embeds the genuine sources of all three (SURVEY.md §8c): the bundle text is
 * evaluated with its entry list emptied (so the browser demo itself never runs)
 * and the bundle's internal require() hands out the modules by id.
 * Nothing from /root/reference is copied into this repository.
 */
const fs = require('fs')
const Module = require('module')

const REF = process.env.DUSP_REFERENCE || '/root/reference'
const BUNDLE = REF + '/demos/browser-dusp-demo-2.bundle.js'
const BARE = ['compute-gcd', 'promise', 'audio-buffer']

let installed = false
let bundleRequireFn = null
function install() {
  if (installed) return
  let text = fs.readFileSync(BUNDLE, 'utf8')
  const tail = text.match(/\},\{\},\[(\d+(?:,\d+)*)\]\)\s*;?\s*$/)
  if (!tail) throw new Error('unrecognised browserify bundle tail in ' + BUNDLE)
  text = text.slice(0, tail.index) + '},{},[])'
  const bundleRequire = (0, eval)(text) // browserify prelude returns its require-by-id
  bundleRequireFn = bundleRequire
  const ids = {}
  for (const name of BARE) {
    const m = text.match(new RegExp('"' + name + '":(\\d+)'))
    if (!m) throw new Error('module ' + name + ' not found in bundle')
    ids[name] = +m[1]
  }
  const load = Module._load
  Module._load = function (request, parent) {
    if (ids[request] !== undefined && parent && parent.filename && parent.filename.startsWith(REF))
      return bundleRequire(ids[request])
    return load.apply(this, arguments)
  }
  installed = true
}

function ref(path) { install(); return require(REF + '/src/' + path) }

/* A module of the reference's bundle by its browserify id.  The bundle is the only place the reference's string
 * front-end still exists (src/parseDSP is an empty submodule): ids 110-137 are the parser, 171 is unDusp, 170
 * renderChannelData, 99 config.  The bundle is a SEPARATE instance of the library (its own config, classes and
 * wave tables), so set bundle(99).sampleRate before touching anything else in it. */
function bundle(id) { install(); return bundleRequireFn(id) }

module.exports = { ref, bundle, REF }
