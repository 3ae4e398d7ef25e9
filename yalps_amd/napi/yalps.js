// yalps.js -- `solve(model, options) -> Solution` for node on top of the N-API addon (CommonJS, node >= 12).
//
// The whole public entry point of the reference (src/YALPS.ts:73-92) with the simplex on the MI355X: the model is turned
// into the reference's tableau layout here on the host (what src/tableau.ts:47-137 does; written against that behaviour,
// pinned by the reference's own test models through tests/test_napi.py), the pivot loop -- and for models with integers
// the whole branch and cut -- runs in the addon (simplex / solveInteger = yalps_simplex_f64 / yalps_milp_f64), and the
// Solution object is marshalled from column 0 and the permutations exactly as src/YALPS.ts:8-50 does.
// An integration that keeps the TypeScript host only needs the addon's `simplex` (INTEGRATION.md section 2); this file is
// for callers who want the reference's `solve` itself.
"use strict"
const path = require("path")
const addon = require(path.join(__dirname, "yalps_napi.node"))

// src/YALPS.ts:52-60
const defaultOptionValues = { precision: 1e-8, checkCycles: false, maxPivots: 8192, tolerance: 0, timeout: Infinity,
                              maxIterations: 32768, includeZeroVariables: false }
const defaultOptions = Object.assign({}, defaultOptionValues) // (a copy, like the reference's export, :65)

// iterable of [key, value] | plain object -> array of pairs (an object's own enumerable properties in property order)
const pairs = seq => (seq == null ? [] : typeof seq[Symbol.iterator] === "function" ? Array.from(seq) : Object.entries(seq))
// true | iterable of keys -> true | Set
const keySet = s => (s === true ? true : s === false || s == null ? new Set() : s instanceof Set ? s : new Set(s))

// Model -> { matrix, width, height, positionOfVariable, variableAtPosition, sign, variables, integers }.  `room` = rows
// to reserve behind the tableau (solveInteger appends up to 2 * integers.length cut rows in place).
function tableauModel(model, extraRowsPerInteger) {
  const sign = model.direction === "minimize" ? -1.0 : 1.0
  const variables = pairs(model.variables)
  const binaryColumns = [], integers = []
  if (model.integers != null || model.binaries != null) {
    const bin = keySet(model.binaries)
    const int = bin === true ? true : keySet(model.integers)
    variables.forEach(([key], i) => {
      if (bin === true || bin.has(key)) { binaryColumns.push(i + 1); integers.push(i + 1) }
      else if (int === true || int.has(key)) integers.push(i + 1)
    })
  }
  // constraints with the same key merge to the tightest [lower, upper]; `equal` overrides min / max
  const bounds = new Map()
  for (const [key, c] of pairs(model.constraints)) {
    let b = bounds.get(key)
    if (b === undefined) { b = { row: -1, lower: -Infinity, upper: Infinity }; bounds.set(key, b) }
    const lo = c.equal != null ? c.equal : c.min != null ? c.min : -Infinity
    const hi = c.equal != null ? c.equal : c.max != null ? c.max : Infinity
    b.lower = Math.max(b.lower, lo)
    b.upper = Math.min(b.upper, hi)
  }
  let rows = 1 // row 0 = objective; per key: the upper-bound row first, then the lower-bound row
  for (const b of bounds.values()) { b.row = rows; rows += (Number.isFinite(b.lower) ? 1 : 0) + (Number.isFinite(b.upper) ? 1 : 0) }
  const width = variables.length + 1, height = rows + binaryColumns.length
  const room = height + extraRowsPerInteger * integers.length
  const matrix = new Float64Array(width * room)
  const positionOfVariable = new Int32Array(width + room), variableAtPosition = new Int32Array(width + room)
  for (let i = 0; i < width + height; i++) { positionOfVariable[i] = i; variableAtPosition[i] = i }
  variables.forEach(([, coefficients], i) => {
    const col = i + 1
    for (const [key, coef] of pairs(coefficients)) { // (a later coefficient on the same key overwrites an earlier one)
      if (model.objective !== undefined && key === model.objective) matrix[col] = sign * coef
      const b = bounds.get(key)
      if (b === undefined) continue
      if (Number.isFinite(b.upper)) {
        matrix[b.row * width + col] = coef
        if (Number.isFinite(b.lower)) matrix[(b.row + 1) * width + col] = -coef
      } else if (Number.isFinite(b.lower)) matrix[b.row * width + col] = -coef
    }
  })
  for (const b of bounds.values()) {
    if (Number.isFinite(b.upper)) {
      matrix[b.row * width] = b.upper
      if (Number.isFinite(b.lower)) matrix[(b.row + 1) * width] = -b.lower
    } else if (Number.isFinite(b.lower)) matrix[b.row * width] = -b.lower
  }
  binaryColumns.forEach((col, i) => { // x <= 1 for every binary, behind the constraint rows
    matrix[(rows + i) * width] = 1.0
    matrix[(rows + i) * width + col] = 1.0
  })
  return { matrix, width, height, positionOfVariable, variableAtPosition, sign, variables, integers }
}

// src/util.ts:1-4 (Math.round: halves toward +infinity)
const roundToPrecision = (num, precision) => {
  const rounding = Math.round(1.0 / precision)
  return Math.round((num + Number.EPSILON) * rounding) / rounding
}

// src/YALPS.ts:8-50 on column 0 and the permutations of a tableau of `height` rows
function solution(tm, height, status, result, options) {
  if (status === "optimal" || (status === "timedout" && !Number.isNaN(result))) {
    const variables = []
    tm.variables.forEach(([key], i) => {
      const row = tm.positionOfVariable[i + 1] - tm.width
      const value = row >= 0 ? tm.matrix[row * tm.width] : 0.0
      if (value > options.precision) variables.push([key, roundToPrecision(value, options.precision)])
      else if (options.includeZeroVariables) variables.push([key, 0.0])
    })
    return { status, result: -tm.sign * result, variables }
  }
  if (status === "unbounded") {
    const variable = tm.variableAtPosition[result] - 1
    return { status: "unbounded", result: tm.sign * Infinity,
             variables: variable >= 0 && variable < tm.variables.length ? [[tm.variables[variable][0], Infinity]] : [] }
  }
  return { status, result: NaN, variables: [] } // infeasible | cycled | timedout without a result
}

// src/YALPS.ts:73-92.  nodeBatch: frontier nodes per GPU batch of the native branch and cut (0 = one node at a time).
function solve(model, options, nodeBatch) {
  if (model == null) throw new Error("model was null or undefined.")
  const opt = Object.assign({}, defaultOptionValues, options)
  const tm = tableauModel(model, 2)
  if (tm.integers.length === 0) {
    const view = { matrix: tm.matrix.subarray(0, tm.width * tm.height), width: tm.width, height: tm.height,
                   positionOfVariable: tm.positionOfVariable.subarray(0, tm.width + tm.height),
                   variableAtPosition: tm.variableAtPosition.subarray(0, tm.width + tm.height) }
    const [status, result] = addon.simplex(view, opt)
    return solution(tm, tm.height, status, result, opt)
  }
  const [status, result, height] = addon.solveInteger(tm, Int32Array.from(tm.integers), tm.sign, opt, nodeBatch == null ? 32 : nodeBatch)
  return solution(tm, height, status, result, opt)
}

module.exports = { solve, defaultOptions, tableauModel, addon,
                   lessEq: value => ({ max: value }), greaterEq: value => ({ min: value }), equalTo: value => ({ equal: value }),
                   inRange: (lower, upper) => ({ min: lower, max: upper }) }
