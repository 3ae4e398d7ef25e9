#!/usr/bin/env python3
"""Generate tests/golden/* by RUNNING THE REFERENCE's own hot path.

TEST-FIXTURE TOOLING, build container only (needs /root/reference and node).
  1. erase_types.py strips the type syntax of /root/reference/src/{simplex,
     tableau,util}.ts into /tmp/yalps_erased (outside the repo, never shipped);
  2. a small driver (written below, our code) feeds the reference's
     tableauModel()+simplex() with (a) every tests/cases/*.json model of the
     reference's own test-suite, (b) dense-LP(M,N,seed) tableaux (SURVEY.md
     section 8d) and (c) sparse mixed-sign tableaux with near-1e-16 entries,
     and records for each run: status, result, the pivot sequence, the final
     permutations, the final RHS column and SHA-256 digests of the initial and
     final Float64Array bytes;
  3. the records are written as gzip'd JSON DATA under tests/golden/, and the
     reference's test-case data files are copied to tests/golden/cases/.

Only data travels: no reference source text is stored anywhere in the repo.
Usage:  python oracle/tools/gen_golden.py [--max-dense 2048]
"""
import argparse
import gzip
import json
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
from erase_types import erase  # noqa: E402

DRIVER = r"""
import { simplex } from "./simplex.mjs"
import { tableauModel } from "./tableau.mjs"
import * as fs from "fs"
import * as crypto from "crypto"

const b64 = (ta) => Buffer.from(ta.buffer, ta.byteOffset, ta.byteLength).toString("base64")
const sha = (ta) => crypto.createHash("sha256").update(Buffer.from(ta.buffer, ta.byteOffset, ta.byteLength)).digest("hex")
const num = (x) => (Number.isFinite(x) ? x : String(x))

// the reference test-suite's PRNG (tests/helpers/util.ts:20-41), restated
const hash32 = (n) => { let x = n; x ^= x >>> 16; x = Math.imul(x, 0x21f0aaad); x ^= x >>> 15; x = Math.imul(x, 0xd35a2d97); x ^= x >>> 15; return x }
const newRand = (seed) => () => { seed += 0x9e3779b9; return (hash32(seed) >>> 0) / 4294967296 }

const identityPerms = (n) => { const p = new Int32Array(n), v = new Int32Array(n); for (let i = 0; i < n; i++) { p[i] = i; v[i] = i } return [p, v] }

const coo = (matrix) => {
  const idx = [], vals = []
  for (let i = 0; i < matrix.length; i++) if (matrix[i] !== 0 || Object.is(matrix[i], -0)) { idx.push(i); vals.push(matrix[i]) }
  return { idx: b64(Int32Array.from(idx)), val: b64(Float64Array.from(vals)) }
}

const run = (rec, tableau, options, storeInit) => {
  rec.width = tableau.width; rec.height = tableau.height
  rec.init_sha256 = sha(tableau.matrix)
  if (storeInit) rec.init_coo = coo(tableau.matrix)
  rec.options = { precision: options.precision, maxPivots: num(options.maxPivots), checkCycles: !!options.checkCycles }
  globalThis.__yalps_trace = []
  const t0 = Date.now()
  const [status, result] = simplex(tableau, options)
  rec.wall_ms = Date.now() - t0
  rec.status = status; rec.result = num(result)
  rec.n_pivots = globalThis.__yalps_trace.length / 2
  rec.pivots = b64(Int32Array.from(globalThis.__yalps_trace))
  rec.pos = b64(tableau.positionOfVariable); rec.var = b64(tableau.variableAtPosition)
  const col0 = new Float64Array(tableau.height)
  for (let r = 0; r < tableau.height; r++) col0[r] = tableau.matrix[r * tableau.width]
  rec.col0 = b64(col0)
  rec.final_sha256 = sha(tableau.matrix)
  console.log(JSON.stringify(rec))
}

const defaults = { precision: 1e-8, checkCycles: false, maxPivots: 8192 }
const mode = process.argv[2]

if (mode === "cases") {
  const dir = "/root/reference/tests/cases"
  for (const file of fs.readdirSync(dir).sort()) {
    const data = JSON.parse(fs.readFileSync(dir + "/" + file, "utf-8"))
    const tm = tableauModel(data.model)
    const rec = { kind: "case", name: file.replace(/\.json$/, ""), sign: tm.sign, integers: tm.integers }
    run(rec, tm.tableau, Object.assign({}, defaults, data.options || {}), true)
  }
} else if (mode === "dense") {
  const maxDense = Number(process.argv[3])
  const shapes = [[2, 2, 42], [5, 7, 42], [16, 16, 42], [16, 16, 7], [33, 20, 42], [64, 64, 42], [64, 64, 7], [100, 37, 3],
                  [128, 128, 42], [200, 300, 42], [256, 256, 42], [512, 512, 42], [1024, 1024, 42], [2048, 2048, 42]]
  for (const [M, N, seed] of shapes) {
    if (Math.max(M, N) > maxDense) continue
    const w = N + 1, h = M + 1, rand = newRand(seed)
    const matrix = new Float64Array(w * h)
    for (let j = 1; j < w; j++) matrix[j] = rand()
    for (let r = 1; r < h; r++) { matrix[r * w] = N * 0.25 * (1 + rand()); for (let j = 1; j < w; j++) matrix[r * w + j] = rand() }
    const [pos, vr] = identityPerms(w + h)
    run({ kind: "dense", M, N, seed }, { matrix, width: w, height: h, positionOfVariable: pos, variableAtPosition: vr },
        Object.assign({}, defaults, { maxPivots: Infinity }), false)
  }
} else if (mode === "mixed") {
  // sparse mixed-sign tableaux: negative RHS rows (phase 1), exact zeros and
  // entries straddling the 1e-16 flush/skip threshold of pivot()
  const shapes = [[3, 3], [5, 4], [12, 9], [9, 14], [30, 40], [64, 33], [100, 80], [150, 200]]
  const variants = [
    {}, { checkCycles: true }, {}, { maxPivots: 3 }, { precision: 1e-6 }, {}, { precision: 1e-11, checkCycles: true },
    { maxPivots: 2.5 }, {}, { maxPivots: 0 }, { checkCycles: true, maxPivots: 700 }, { precision: 0 }, {},
  ]
  let id = 0
  for (const [M, N] of shapes) for (let s = 0; s < 6; s++) {
    const seed = 1000 * M + 10 * N + s, rand = newRand(seed)
    const density = 0.15 + 0.7 * rand(), negFrac = s % 3 === 0 ? 0 : 0.4 * rand()
    const w = N + 1, h = M + 1
    const init = new Float64Array(w * h)
    const entry = () => { const u = rand(); if (u >= density) return 0; const t = rand(); return t < 0.06 ? (rand() - 0.5) * 4e-16 : rand() * 2 - 1 }
    for (let j = 1; j < w; j++) init[j] = entry()
    for (let r = 1; r < h; r++) {
      init[r * w] = rand() < negFrac ? -rand() * N * 0.1 : (rand() < 0.15 ? 0 : rand() * N * 0.25)
      for (let j = 1; j < w; j++) init[r * w + j] = entry()
    }
    const opts = Object.assign({}, defaults, variants[(id + s) % variants.length])
    const [pos, vr] = identityPerms(w + h)
    run({ kind: "mixed", id: id++, M, N, seed }, { matrix: Float64Array.from(init), width: w, height: h, positionOfVariable: pos, variableAtPosition: vr }, opts, true)
  }
}
"""


def run_driver(erased, mode, *args):
    out = subprocess.run(["node", os.path.join(erased, "golden_driver.mjs"), mode, *map(str, args)],
                         check=True, capture_output=True, text=True).stdout
    return [json.loads(line) for line in out.splitlines() if line.startswith("{")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-dense", type=int, default=2048)
    ap.add_argument("--erased-dir", default="/tmp/yalps_erased")
    args = ap.parse_args()

    erased = erase(args.erased_dir)
    with open(os.path.join(erased, "golden_driver.mjs"), "w") as f:
        f.write(DRIVER)
    golden = os.path.join(REPO, "tests", "golden")
    os.makedirs(os.path.join(golden, "cases"), exist_ok=True)

    for mode, extra in (("cases", ()), ("mixed", ()), ("dense", (args.max_dense,))):
        recs = run_driver(erased, mode, *extra)
        path = os.path.join(golden, f"simplex_{mode}.json.gz")
        with gzip.GzipFile(path, "wb", mtime=0) as gz:
            gz.write(json.dumps({"generator": "oracle/tools/gen_golden.py", "reference": "Ivordir/YALPS src/simplex.ts "
                                 "(type-erased, node %s)" % subprocess.run(["node", "--version"], capture_output=True,
                                                                           text=True).stdout.strip(),
                                 "records": recs}).encode())
        print(f"{mode}: {len(recs)} records -> {path} ({os.path.getsize(path)} bytes)")

    # the reference test-suite's own data files (model + expected), as data
    src = "/root/reference/tests/cases"
    for name in sorted(os.listdir(src)):
        shutil.copyfile(os.path.join(src, name), os.path.join(golden, "cases", name))
    print("copied", len(os.listdir(src)), "case data files")


if __name__ == "__main__":
    main()
