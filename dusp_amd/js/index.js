'use strict'
/* dusp-hip: the `renderChannelData` / `quick` surface of Dusp (reference src/index.js:6,11) plus the
 * constructors of the units the MI355X render path executes. */
// renderMany keeps one render per GPU in flight on the libuv pool, whose size (default 4) is read when the pool is first used: on a node
// with more GPUs ask for as many threads, unless the host application has chosen a size itself
if (!process.env.UV_THREADPOOL_SIZE) process.env.UV_THREADPOOL_SIZE = '16'
const graph = require('./lib/graph')
const renderChannelData = require('./lib/renderChannelData')
const dusp = require('./lib/dusp')

module.exports = {
  renderChannelData,
  renderMany: renderChannelData.renderMany,
  deviceCount: renderChannelData.deviceCount,
  renderDescriptor: renderChannelData.renderDescriptor,
  quick: require('./lib/quick'),
  unDusp: require('./lib/unDusp'),
  dusp,
  RenderStream: require('./lib/RenderStream'),
  SegmentRenderer: renderChannelData.SegmentRenderer,
  encodeWav: require('./lib/wav').encodeWav,
  decodeWav: require('./lib/wav').decodeWav,
  parse: require('./lib/parse'),
  config: require('./lib/config'),
  extract: require('./lib/extract').extract,
  unify: require('./lib/extract').unify,
  Unit: graph.Unit,
  Circuit: graph.Circuit,
  CircleBuffer: graph.CircleBuffer,
  components: {
    Osc: graph.Osc, Ramp: graph.Ramp, Multiply: graph.Multiply, Sum: graph.Sum, Filter: graph.Filter,
    Delay: graph.Delay, CircleBufferReader: graph.CircleBufferReader, CircleBufferWriter: graph.CircleBufferWriter,
    Repeater: graph.Repeater,
  },
  ...graph,
}
