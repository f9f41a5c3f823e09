// Node entry point: same exports as the reference's bindings/node/index.js (DasContextJs + the size constants), served by the
// MI355X library through the N-API addon next to this file (make -C bindings/node).
'use strict'
const native = require('./eth_kzg_amd.node')
module.exports = native
module.exports.DASContextJs = native.DasContextJs
