// Test driver for yalps.js (node >= 12, CommonJS).  Reads {model, options, nodeBatch} as JSON from stdin ("Infinity" /
// "-Infinity" / "NaN" strings stand for those numbers in options) and prints the Solution the same way.
"use strict"
const path = require("path")
const job = JSON.parse(require("fs").readFileSync(0, "utf-8"))
const dec = x => (x === "Infinity" ? Infinity : x === "-Infinity" ? -Infinity : x === "NaN" ? NaN : x)
const enc = x => (typeof x === "number" && !Number.isFinite(x) ? String(x) : x)
const options = {}
for (const k of Object.keys(job.options || {})) options[k] = dec(job.options[k])
let out
try {
  const { solve } = require(path.join(__dirname, "yalps.js"))
  const s = solve(job.model, options, job.nodeBatch)
  out = { status: s.status, result: enc(s.result), variables: s.variables.map(([k, v]) => [k, enc(v)]) }
} catch (e) {
  out = { error: String(e && e.message ? e.message : e) }
}
console.log(JSON.stringify(out))
