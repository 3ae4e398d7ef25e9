// Test driver for the N-API shim (node >= 12, CommonJS).  Reads a JSON job from stdin:
//   {matrix: [...], width, height, options: {precision, maxPivots, checkCycles}, viewOffset: k}
// builds the Tableau object the reference passes to simplex() (src/tableau.ts:9-15) -- the typed
// arrays are subarray() VIEWS at element offset `viewOffset` into larger buffers, like branch and
// cut's buffers (src/branchAndCut.ts:55-59) -- calls the addon and prints the mutated tableau.
"use strict"
const path = require("path")
const addon = require(path.join(__dirname, "yalps_napi.node"))
const job = JSON.parse(require("fs").readFileSync(0, "utf-8"))
const { width, height } = job
const off = job.viewOffset || 0
const n = width + height
const matBuf = new Float64Array(off + width * height + 5).fill(-777)
const posBuf = new Int32Array(off + n + 3).fill(-7)
const varBuf = new Int32Array(off + n + 3).fill(-7)
const matrix = matBuf.subarray(off, off + width * height)
const positionOfVariable = posBuf.subarray(off, off + n)
const variableAtPosition = varBuf.subarray(off, off + n)
matrix.set(job.matrix)
for (let i = 0; i < n; i++) { positionOfVariable[i] = i; variableAtPosition[i] = i }
const opt = job.options || {}
const options = { precision: opt.precision == null ? 1e-8 : opt.precision,
                  maxPivots: opt.maxPivots === "Infinity" ? Infinity : (opt.maxPivots == null ? 8192 : opt.maxPivots),
                  checkCycles: !!opt.checkCycles }
let out
try {
  const [status, result] = addon.simplex({ matrix, width, height, positionOfVariable, variableAtPosition }, options)
  const guardsIntact = matBuf.slice(0, off).every(x => x === -777) && matBuf.slice(off + width * height).every(x => x === -777) &&
                       posBuf.slice(0, off).every(x => x === -7) && varBuf.slice(off + n).every(x => x === -7)
  out = { status, result: Number.isFinite(result) ? result : String(result), matrix: Array.from(matrix),
          positionOfVariable: Array.from(positionOfVariable), variableAtPosition: Array.from(variableAtPosition), guardsIntact }
} catch (e) {
  out = { error: String(e && e.message ? e.message : e) }
}
console.log(JSON.stringify(out))
