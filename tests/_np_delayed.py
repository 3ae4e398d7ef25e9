"""numpy model of the DELAYED ROW UPDATES of stream2_kernel / stream3_kernel / dshard_kernel -- TEST
INFRASTRUCTURE, like tests/_np_simplex.py, whose selection rules it shares.  Not the reference's algorithm restated but the
kernels' data flow restated on the CPU, so that the claim they rest on -- "bit for bit what one sweep per pivot leaves" -- is
checked without a GPU (tests/test_delayed_model.py compares it with tests/_np_simplex.py, which the reference's golden
records pin):
  * a pivot's elimination stays pending: its normalised row, the rows' entries of its column as they were, what replaces them;
  * the RHS column and the objective row are updated at once;
  * a column "as it is now" = memory run through the pending pivots, one scalar chain per row (`column_now`);
  * a candidate row "as it is now" = memory run through the pending pivots in registers (`row_now`);
  * every `depth` pivots, and at the end, every touched row gets all pending eliminations, in order (`flush`).
numpy never fuses a multiply with a subtract: every product and difference is rounded on its own, as in the kernels."""
import math

import numpy as np

from tests._np_simplex import round_to_precision


class _Pending:
    __slots__ = ("row", "col", "pn", "nz", "colv", "nq")


def simplex_delayed(matrix, width, height, pos, var, depth, precision=1e-8, max_pivots=8192.0):
    """Same contract as tests/_np_simplex.simplex: (status, result, n_pivots); matrix and permutations updated in place."""
    M = matrix.reshape(height, width)  # memory: rows as of the last flush (column 0 too: the RHS is written back on the way out)
    w = width
    rhs = M[:, 0].copy()  # current
    obj = M[0].copy()     # current objective row, columns 1.. (index 0 unused)
    pend = []

    def column_now(c):
        v = M[:, c].copy()
        for p in pend:
            act = np.abs(p.colv) > 1e-16
            act[p.row] = False
            if c == p.col:
                v[act] = p.nq[act]
                v[p.row] = p.nq[p.row]
            else:
                if p.nz[c]:
                    v[act] = v[act] - p.colv[act] * p.pn[c]
                v[p.row] = p.pn[c] if p.nz[c] else 0.0
        return v

    def apply_to_row(x, i, p):  # src/simplex.ts:14-25 for the pivot row itself, :31-36 for another row
        if i == p.row:
            x[1:] = np.where(p.nz[1:], p.pn[1:], 0.0)
            x[p.col] = p.nq[i]
        elif abs(p.colv[i]) > 1e-16:
            nzi = np.flatnonzero(p.nz[1:]) + 1
            x[nzi] = x[nzi] - p.colv[i] * p.pn[nzi]
            x[p.col] = p.nq[i]

    def row_now(i):
        x = M[i].copy()
        for p in pend:
            apply_to_row(x, i, p)
        return x

    def flush():
        for p in pend:  # (the kernels take a row through all pending pivots at once; the order per element is this one)
            for i in range(height):
                apply_to_row(M[i], i, p)
        if pend:
            assert np.array_equal(M[0, 1:].view(np.int64), obj[1:].view(np.int64)), "objective replica diverged from the flushed row"
        pend.clear()

    def leave(status, result, npiv):
        flush()
        M[:, 0] = rhs
        return status, result, npiv

    npiv, it, phase = 0, 0.0, 1
    while True:
        if not it < max_pivots:
            return leave("cycled", math.nan, npiv)
        if phase == 1:
            r = rhs[1:]
            if r.size == 0 or not (r.min() < -precision):
                phase, it = 2, 0.0
                continue
            row = int(np.argmin(r)) + 1
            x = row_now(row)
            coef = x[1:]
            elig = np.flatnonzero(coef < -precision)
            ratio = -obj[1:][elig] / coef[elig]
            ok = ratio > -math.inf
            if not ok.any():
                return leave("infeasible", math.nan, npiv)
            best = ratio[ok].max()
            col = int(elig[ok][np.argmax(ratio[ok] == best)]) + 1
            colv = column_now(col)
        else:
            o = obj[1:]
            elig = np.flatnonzero(o > precision)
            if elig.size == 0:
                return leave("optimal", round_to_precision(float(rhs[0]), precision), npiv)
            col = int(elig[np.argmax(o[elig])]) + 1
            colv = column_now(col)
            value = colv[1:]
            rows = np.flatnonzero(value > precision)
            with np.errstate(all="ignore"):
                ratio = rhs[1:][rows] / value[rows]
            ok = ratio < math.inf
            rows, ratio = rows[ok], ratio[ok]
            if rows.size == 0:
                return leave("unbounded", float(col), npiv)
            early = np.flatnonzero(ratio <= precision)
            row = int(rows[early[0]] if early.size else rows[np.argmin(ratio)]) + 1
            x = row_now(row)
        assert colv[0] == obj[col] or (colv[0] != colv[0] and obj[col] != obj[col]), "scalar chain of the objective row diverged"
        # ---- the pivot: pending from here on ----
        leaving, entering = var[w + row], var[col]
        var[w + row], var[col] = entering, leaving
        pos[leaving], pos[entering] = col, w + row
        q = x[col]
        p = _Pending()
        p.row, p.col = row, col
        p.nz = np.abs(x) > 1e-16
        p.nz[0] = False
        p.pn = np.where(p.nz, x / q, 0.0)
        p.colv = colv
        p.nq = -colv / q
        p.nq[row] = 1.0 / q
        act = np.abs(colv) > 1e-16
        act[row] = False
        rhs_row = rhs[row]
        if abs(rhs_row) > 1e-16:  # column 0 of :14-23 / :33
            pn_rhs = rhs_row / q
            rhs[act] = rhs[act] - colv[act] * pn_rhs
            rhs[row] = pn_rhs
        else:
            rhs[row] = 0.0
        if act[0]:  # the objective row, :27-38 for row 0
            nzi = np.flatnonzero(p.nz)
            obj[nzi] = obj[nzi] - colv[0] * p.pn[nzi]
            obj[col] = p.nq[0]
        pend.append(p)
        it += 1.0
        npiv += 1
        if len(pend) == depth:
            flush()
