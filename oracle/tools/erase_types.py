#!/usr/bin/env python3
"""Type-erase the reference's hot-path TypeScript into runnable ES modules.

TEST-FIXTURE TOOLING (build container only).  The reference (Ivordir/YALPS) is
TypeScript and this image has node 12 but no TypeScript compiler, so the three
source files the hot path lives in are read from /root/reference/src, stripped
of their type syntax and written to a scratch directory OUTSIDE the repository
(default /tmp/yalps_erased).  Nothing produced here is committed or shipped:
gen_golden.py runs the erased modules once to emit tests/golden/*.json.gz.

Every edit is a literal, asserted replacement, so a changed reference fails
loudly instead of silently producing different code.  No behaviour is changed:
the only non-type edits are `??` -> explicit null checks (node 12 has no `??`)
and one appended trace hook in pivot() that records (row, col).
"""
import os
import re
import sys

REF_SRC = "/root/reference/src"


def _apply(text, edits, fname):
    for old, new in edits:
        if old not in text:
            raise SystemExit(f"erase_types: pattern not found in {fname}: {old!r}")
        text = text.replace(old, new)
    return text


def _drop_type_decls(text):
    """Remove `type X = ...` / `export type X = ...` statements (brace balanced)."""
    out, lines, i = [], text.split("\n"), 0
    while i < len(lines):
        ln = lines[i]
        if re.match(r"^(export )?type \w+", ln):
            depth = ln.count("{") - ln.count("}")
            while depth > 0:
                i += 1
                depth += lines[i].count("{") - lines[i].count("}")
            i += 1
            continue
        out.append(ln)
        i += 1
    return "\n".join(out)


SIMPLEX_EDITS = [
    ('import { Options, SolutionStatus } from "./types.js"\n', ""),
    ('import { index, Tableau, update } from "./tableau.js"', 'import { index, update } from "./tableau.mjs"'),
    ('from "./util.js"', 'from "./util.mjs"'),
    ("(tableau: Tableau, row: number, col: number) => {",
     "(tableau, row, col) => {\n  if (globalThis.__yalps_trace) globalThis.__yalps_trace.push(row, col)"),
    ("const nonZeroColumns: number[] = []", "const nonZeroColumns = []"),
    ("(history: PivotHistory, tableau: Tableau, row: number, col: number)", "(history, tableau, row, col)"),
    ("const phase2 = (tableau: Tableau, options: Required<Options>): [SolutionStatus, number] =>",
     "const phase2 = (tableau, options) =>"),
    ("const phase1 = (tableau: Tableau, options: Required<Options>): [SolutionStatus, number] =>",
     "const phase1 = (tableau, options) =>"),
    ("const pivotHistory: PivotHistory = []", "const pivotHistory = []"),
]

TABLEAU_EDITS = [
    ('import { Coefficients, Model } from "./types.js"\n', ""),
    ("(tableau: Tableau, row: number, col: number) =>", "(tableau, row, col) =>"),
    ("(tableau: Tableau, row: number, col: number, value: number) =>", "(tableau, row, col, value) =>"),
    ("const convertToIterable = <K, V>(\n"
     "  seq: Iterable<readonly [K, V]> | ([K] extends [string] ? { readonly [key in K]?: V } : never),\n"
     ") =>", "const convertToIterable = (seq) =>"),
    ("(seq as any)[Symbol.iterator]", "(seq)[Symbol.iterator]"),
    ("(seq as Iterable<readonly [K, V]>)", "(seq)"),
    ("(Object.entries(seq) as Iterable<readonly [K, V]>)", "(Object.entries(seq))"),
    ("const convertToSet = <T>(set: boolean | Iterable<T> | undefined): true | Set<T> =>",
     "const convertToSet = (set) =>"),
    ("export const tableauModel = <VarKey = string, ConKey = string>(\n"
     "  model: Model<VarKey, ConKey>,\n"
     "): TableauModel<VarKey, ConKey> => {", "export const tableauModel = (model) => {"),
    ("const variables: Variables<VarKey, ConKey> =", "const variables ="),
    ("const binaryConstraintCol: number[] = []", "const binaryConstraintCol = []"),
    ("const ints: number[] = []", "const ints = []"),
    ("new Map<ConKey, { row: number; lower: number; upper: number }>()", "new Map()"),
    # node 12 has no `??`: a ?? b  ==  (a != null ? a : b)
    ("constraints.get(key) ?? { row: NaN, lower: -Infinity, upper: Infinity }",
     "(constraints.get(key) != null ? constraints.get(key) : { row: NaN, lower: -Infinity, upper: Infinity })"),
    ("constraint.equal ?? constraint.min ?? -Infinity",
     "(constraint.equal != null ? constraint.equal : constraint.min != null ? constraint.min : -Infinity)"),
    ("constraint.equal ?? constraint.max ?? Infinity",
     "(constraint.equal != null ? constraint.equal : constraint.max != null ? constraint.max : Infinity)"),
]

UTIL_EDITS = [
    ("(num: number, precision: number) =>", "(num, precision) =>"),
]


def erase(out_dir="/tmp/yalps_erased"):
    os.makedirs(out_dir, exist_ok=True)
    jobs = [("simplex.ts", SIMPLEX_EDITS), ("tableau.ts", TABLEAU_EDITS), ("util.ts", UTIL_EDITS)]
    for fname, edits in jobs:
        with open(os.path.join(REF_SRC, fname)) as f:
            text = f.read()
        text = _drop_type_decls(_apply(text, edits, fname))
        code = re.sub(r"export \{[^}]*\}", "", text)  # `export { a as b }` is plain JS
        code = re.sub(r"//[^\n]*", "", code)
        leftover = re.findall(r":\s*(?:number|Tableau|string)\b|\bas\s+\w+|<\w+(?:,\s*\w+)*>\(", code)
        if leftover:
            raise SystemExit(f"erase_types: type syntax left in {fname}: {leftover[:5]}")
        with open(os.path.join(out_dir, fname.replace(".ts", ".mjs")), "w") as f:
            f.write(text)
    return out_dir


if __name__ == "__main__":
    print(erase(*sys.argv[1:]))
