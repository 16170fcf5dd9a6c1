'use strict'
/* Unit labels are "<Kind><running count>" in both libraries, but the counts depend on everything built before in the
 * process: renumber the `#Kind12` references of a Dusp string by first appearance before comparing. */
module.exports = function normaliseLabels(text) {
  const seen = new Map()
  return String(text).replace(/#([A-Za-z_]+)(\d+)/g, (whole, kind) => {
    if (!seen.has(whole)) seen.set(whole, seen.size + 1)
    return '#' + kind + seen.get(whole)
  })
}
