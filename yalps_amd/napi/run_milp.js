// Test driver for solveInteger (node >= 12, CommonJS).  Reads a JSON job from stdin:
//   {matrix: [...], width, height, integers: [...], sign, options: {...}, nodeBatch}
// and prints status / result / height and, of the best tableau, column 0 (hex of the bytes) and both permutations.
"use strict"
const path = require("path")
const addon = require(path.join(__dirname, "yalps_napi.node"))
const job = JSON.parse(require("fs").readFileSync(0, "utf-8"))
const { width, height } = job
const integers = Int32Array.from(job.integers)
const hmax = height + 2 * integers.length
const matrix = new Float64Array(width * hmax)
matrix.set(job.matrix)
const positionOfVariable = new Int32Array(width + hmax), variableAtPosition = new Int32Array(width + hmax)
for (let i = 0; i < width + height; i++) { positionOfVariable[i] = i; variableAtPosition[i] = i }
const opt = job.options || {}
const inf = x => (x === "Infinity" ? Infinity : x)
const options = { precision: opt.precision == null ? 1e-8 : opt.precision, maxPivots: opt.maxPivots == null ? 8192 : inf(opt.maxPivots),
                  checkCycles: !!opt.checkCycles, tolerance: opt.tolerance || 0, timeout: opt.timeout == null ? Infinity : inf(opt.timeout),
                  maxIterations: opt.maxIterations == null ? 32768 : inf(opt.maxIterations) }
let out
try {
  const [status, result, h] = addon.solveInteger({ matrix, width, height, positionOfVariable, variableAtPosition }, integers,
                                                 job.sign, options, job.nodeBatch || 0)
  const col0 = new Float64Array(h)
  for (let r = 0; r < h; r++) col0[r] = matrix[r * width]
  out = { status, result: Number.isFinite(result) ? result : String(result), height: h,
          col0: Buffer.from(col0.buffer).toString("hex"),
          positionOfVariable: Array.from(positionOfVariable.subarray(0, width + h)),
          variableAtPosition: Array.from(variableAtPosition.subarray(0, width + h)) }
} catch (e) {
  out = { error: String(e && e.message ? e.message : e) }
}
console.log(JSON.stringify(out))
