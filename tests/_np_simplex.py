"""Vectorised numpy restatement of the hot path (TEST INFRASTRUCTURE, like oracle/): the same
simplex() as oracle/simplex_oracle.c -- src/simplex.ts:5-39 (pivot), :66-103 (phase 2), :106-142
(phase 1), src/util.ts:1-4 -- but with whole-row numpy operations, so that tableaux of BASELINE's
full sizes (16385 x 16385: 2.1 GB) can be checked bit for bit in seconds.  numpy never fuses a
multiply with a subtract, so `a - c * p` rounds twice exactly like V8 and like the kernels.
It is pinned itself: tests/test_oracle_golden.py runs it over the golden records of the reference.
No checkCycles (the full-size workloads do not use it)."""
import math

import numpy as np

STATUS = ("optimal", "infeasible", "unbounded", "cycled")


def _js_round(x):
    if x != x or math.isinf(x):
        return x
    f = math.floor(x)
    return f + 1.0 if x - f >= 0.5 else float(f)


def round_to_precision(num, precision):  # src/util.ts:1-4
    rounding = _js_round(1.0 / precision)
    return _js_round((num + 2.220446049250313e-16) * rounding) / rounding


def pivot(M, pos, var, row, col, block=1024):
    """src/simplex.ts:5-39 on the 2-D view M (h, w), in place."""
    h, w = M.shape
    q = M[row, col]
    leaving, entering = var[w + row], var[col]  # :7-12
    var[w + row], var[col] = entering, leaving
    pos[leaving], pos[entering] = col, w + row
    prow = M[row]
    nz = np.abs(prow) > 1e-16  # :14-23 nonZeroColumns
    prow[:] = np.where(nz, prow / q, 0.0)
    prow[col] = 1.0 / q  # :25
    all_nz = bool(nz.all())
    nzi = None if all_nz else np.flatnonzero(nz)
    for r0 in range(0, h, block):  # :27-38
        blk = M[r0:r0 + block]
        coef = blk[:, col].copy()
        act = np.abs(coef) > 1e-16
        if r0 <= row < r0 + block:
            act[row - r0] = False
        if not act.any():
            continue
        ai = np.flatnonzero(act)
        if all_nz:
            blk[ai] = blk[ai] - coef[ai, None] * prow[None, :]
        else:
            sub = blk[np.ix_(ai, nzi)]
            blk[np.ix_(ai, nzi)] = sub - coef[ai, None] * prow[None, nzi]
        blk[ai, col] = -coef[ai] / q  # :36


def simplex(matrix, width, height, pos, var, precision=1e-8, max_pivots=8192.0):
    """Returns (status, result, n_pivots); matrix (flat, row-major) and the permutations are
    updated in place like the reference does."""
    M = matrix.reshape(height, width)
    npiv = 0
    it = 0.0
    phase = 1
    while True:
        if not it < max_pivots:  # :69,109 -> :102,141
            return "cycled", math.nan, npiv
        if phase == 1:
            rhs = M[1:, 0]
            if rhs.size == 0 or not (rhs.min() < -precision):  # :111-120
                phase, it = 2, 0.0
                continue
            row = int(np.argmin(rhs)) + 1  # first minimum
            coef = M[row, 1:]
            elig = np.flatnonzero(coef < -precision)  # :123-134
            ratio = -M[0, 1:][elig] / coef[elig]
            ok = ratio > -math.inf
            if not ok.any():
                return "infeasible", math.nan, npiv
            best = ratio[ok].max()
            col = int(elig[ok][np.argmax(ratio[ok] == best)]) + 1  # first maximum
        else:
            obj = M[0, 1:]
            elig = np.flatnonzero(obj > precision)  # :71-79
            if elig.size == 0:
                return "optimal", round_to_precision(float(M[0, 0]), precision), npiv
            col = int(elig[np.argmax(obj[elig])]) + 1
            value = M[1:, col]
            rows = np.flatnonzero(value > precision)  # :83-95
            with np.errstate(all="ignore"):
                ratio = M[1:, 0][rows] / value[rows]
            ok = ratio < math.inf
            rows, ratio = rows[ok], ratio[ok]
            if rows.size == 0:
                return "unbounded", float(col), npiv
            early = np.flatnonzero(ratio <= precision)  # the `break` at :93
            row = int(rows[early[0]] if early.size else rows[np.argmin(ratio)]) + 1
        pivot(M, pos, var, row, col)
        it += 1.0
        npiv += 1
