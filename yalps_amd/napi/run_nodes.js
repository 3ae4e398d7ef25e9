// Test driver for the branch-and-cut exports of the N-API shim (node >= 12, CommonJS).  Reads a JSON job from stdin:
//   {matrix: [...], width, height, options: {...}, maxCuts, nodes: [[[sign, variable, value], ...], ...]}
// rootSolve()s the tableau (the reference's root simplex() call, src/YALPS.ts:79, left resident in HBM), then
// nodeSolve()s every cut list (applyCuts + simplex of src/branchAndCut.ts:126-127 on the device) and prints
// what came back: the root's status / result / column 0 / permutations and, per node, the same.
"use strict"
const path = require("path")
const addon = require(path.join(__dirname, "yalps_napi.node"))
const job = JSON.parse(require("fs").readFileSync(0, "utf-8"))
const { width, height, maxCuts } = job
const n = width + height
const matrix = Float64Array.from(job.matrix)
const positionOfVariable = new Int32Array(n), variableAtPosition = new Int32Array(n)
for (let i = 0; i < n; i++) { positionOfVariable[i] = i; variableAtPosition[i] = i }
const opt = job.options || {}
const options = { precision: opt.precision == null ? 1e-8 : opt.precision,
                  maxPivots: opt.maxPivots === "Infinity" ? Infinity : (opt.maxPivots == null ? 8192 : opt.maxPivots),
                  checkCycles: !!opt.checkCycles }
const num = x => (Number.isFinite(x) ? x : String(x))
// doubles as hex of their bytes: JSON would turn -0 into 0
const hex = f64 => Buffer.from(f64.buffer, f64.byteOffset, f64.byteLength).toString("hex")
let out
try {
  const [status, result, handle] = addon.rootSolve({ matrix, width, height, positionOfVariable, variableAtPosition }, options, maxCuts)
  const col0 = new Float64Array(height)
  for (let r = 0; r < height; r++) col0[r] = matrix[r * width]
  const nodes = []
  const res = { col0: new Float64Array(height + maxCuts), positionOfVariable: new Int32Array(n + maxCuts),
                variableAtPosition: new Int32Array(n + maxCuts) }
  for (const cuts of job.nodes) {
    const [st, value, h] = addon.nodeSolve(handle, cuts, options, res)
    nodes.push({ status: st, result: num(value), height: h, col0: hex(res.col0.subarray(0, h)),
                 positionOfVariable: Array.from(res.positionOfVariable.subarray(0, width + h)),
                 variableAtPosition: Array.from(res.variableAtPosition.subarray(0, width + h)) })
  }
  addon.rootFree(handle)
  out = { status, result: num(result), col0: hex(col0), positionOfVariable: Array.from(positionOfVariable),
          variableAtPosition: Array.from(variableAtPosition), nodes }
} catch (e) {
  out = { error: String(e && e.message ? e.message : e) }
}
console.log(JSON.stringify(out))
