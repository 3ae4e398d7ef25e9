"""CPU stand-in for the per-rank steps of the row-sharded solve (TEST INFRASTRUCTURE): the same
select / apply protocol as yalps_shard_select / yalps_shard_apply (slot layout included), in plain
numpy, so that yalps_amd.sharded.sharded_simplex + torch.distributed can be exercised with
world_size > 1 on CPU (gloo).  Arithmetic follows src/simplex.ts exactly (numpy never fuses)."""
import math

import numpy as np
import torch

INF, NONE = math.inf, 2147483647
HDR = 8


def _better(ka, ia, kb, ib):
    return ka < kb or (ka == kb and ia < ib)


def _round_to_precision(num, precision):
    from yalps_amd.solve import round_to_precision
    return round_to_precision(num, precision)


class NumpyShardOps:
    def __init__(self, local_matrix, width, bounds, rank, global_height, pos, var):
        self.w, self.rank, self.bounds = width, rank, list(bounds)
        self.nranks = len(bounds) - 1
        self.m = local_matrix.reshape(-1, width).copy()
        self.h = self.m.shape[0]
        self.base = bounds[rank] - 1  # global = local + base for local rows >= 1
        self.pos, self.var = pos.copy(), var.copy()
        self.pitch = (width - 1 + 15) // 16 * 16
        self.slot = HDR + 2 * self.pitch
        self.send = torch.zeros(self.slot, dtype=torch.float64)
        self.recv = torch.zeros(self.nranks * self.slot, dtype=torch.float64)
        self.status, self.result = -1, math.nan

    # -- candidates of my rows for the next decision (global indices) + look-ahead column
    def _scan(self):
        p = self.precision
        obj = self.m[0, 1:]
        cand = np.where(obj > p)[0]
        self.la = int(cand[np.argmax(obj[cand])]) + 1 if cand.size else 0  # first max wins
        self.c_ratio, self.c_rhs = (INF, NONE), (INF, NONE)
        for r in range(1, self.h):
            g = r + self.base
            rhs = self.m[r, 0]
            if rhs < -p and _better(rhs, g, *self.c_rhs):
                self.c_rhs = (rhs, g)
            if self.la:
                v = self.m[r, self.la]
                if v > p:
                    ratio = rhs / v
                    if ratio < INF:
                        key = -INF if ratio <= p else ratio
                        if _better(key, g, *self.c_ratio):
                            self.c_ratio = (key, g)

    def begin(self, precision, max_pivots, check_cycles=False):
        self.precision, self.max_pivots = precision, max_pivots
        self.phase, self.iter, self.pivots = 1, 0.0, 0
        self.check_cycles, self.history = bool(check_cycles), []  # (the history is per phase: src/simplex.ts:67,107)
        self._scan()

    def _has_cycle(self, row, col):
        """src/simplex.ts:44-63 on the replicated basis: every rank runs it on the same pivot, no communication."""
        if not self.check_cycles:
            return False
        h = self.history
        h.append((int(self.var[self.w + row]), int(self.var[col])))
        for length in range(6, len(h) // 2 + 1):
            if all(h[len(h) - 1 - i] == h[len(h) - 1 - i - length] for i in range(length)):
                return True
        return False

    def select(self):
        s = self.send.numpy()
        s[:] = 0.0
        (kr, ir), (kn, inn) = self.c_ratio, self.c_rhs
        s[0:4] = (kr, float(ir), kn, float(inn))
        n = self.w - 1
        if self.status < 0:
            if ir != NONE:
                s[4] = self.m[ir - self.base, 0]
                s[HDR:HDR + n] = self.m[ir - self.base, 1:]
            if inn != NONE:
                s[5] = self.m[inn - self.base, 0]
                s[HDR + self.pitch:HDR + self.pitch + n] = self.m[inn - self.base, 1:]

    def _owner(self, grow):
        g = 0
        for k in range(1, self.nranks):
            if grow >= self.bounds[k]:
                g = k
        return g

    def apply(self):
        if self.status >= 0:
            return
        R = self.recv.numpy().reshape(self.nranks, self.slot)
        p, n = self.precision, self.w - 1
        while True:  # src/simplex.ts:106-142 / 66-103
            if not (self.iter < self.max_pivots):
                self.status, self.result = 3, math.nan
                return
            if self.phase == 1:
                best = (INF, NONE)
                for g in range(self.nranks):
                    if _better(R[g, 2], int(R[g, 3]), *best):
                        best = (R[g, 2], int(R[g, 3]))
                if best[1] == NONE:
                    self.phase, self.iter, self.history = 2, 0.0, []
                    continue
                row = best[1]
                slot = R[self._owner(row)]
                prow, rhs_row = slot[HDR + self.pitch:HDR + self.pitch + n].copy(), slot[5]
                col, mx = 0, -INF
                for c in range(1, self.w):
                    coefficient = prow[c - 1]
                    if coefficient < -p:
                        ratio = -self.m[0, c] / coefficient
                        if ratio > mx:
                            mx, col = ratio, c
                if col == 0:
                    self.status, self.result = 1, math.nan
                    return
                break
            col = self.la
            if col == 0:
                self.status, self.result = 0, _round_to_precision(self.m[0, 0], p)
                return
            best = (INF, NONE)
            for g in range(self.nranks):
                if _better(R[g, 0], int(R[g, 1]), *best):
                    best = (R[g, 0], int(R[g, 1]))
            if best[1] == NONE:
                self.status, self.result = 2, float(col)
                return
            row = best[1]
            slot = R[self._owner(row)]
            prow, rhs_row = slot[HDR:HDR + n].copy(), slot[4]
            break
        if self._has_cycle(row, col):  # :98,137
            self.status, self.result = 3, math.nan
            return
        # pivot (src/simplex.ts:5-39) on my rows, with the raw pivot row (rhs_row, prow)
        full = np.concatenate(([rhs_row], prow))
        q = full[col]
        leaving, entering = self.var[self.w + row], self.var[col]
        self.var[self.w + row], self.var[col] = entering, leaving
        self.pos[leaving], self.pos[entering] = col, self.w + row
        nz = np.abs(full) > 1e-16
        norm = np.where(nz, full / q, 0.0)
        lrow = row - self.base if self.bounds[self.rank] <= row < self.bounds[self.rank + 1] else -1
        for r in range(self.h):
            if r == lrow:
                self.m[r] = norm
                self.m[r, col] = 1.0 / q
                continue
            coef = self.m[r, col]
            if abs(coef) > 1e-16:
                prod = coef * norm
                self.m[r] = np.where(nz, self.m[r] - prod, self.m[r])
                self.m[r, col] = -coef / q
        self.iter += 1.0
        self.pivots += 1
        self._scan()

    def poll(self):
        return self.status, self.result, self.pivots

    def download(self):
        return self.m.reshape(-1).copy(), self.pos, self.var

    def close(self):
        pass


class NumpyDelayedShardOps(NumpyShardOps):
    """The same protocol with the DELAYED ROW UPDATES of dshard_kernel / dshard_select_kernel: a pivot stays
    pending on this rank's rows (its normalised row, my rows' entries of its column as they were, what replaces them), the RHS
    column and the objective row are updated at once, candidates come from scalar chains, the candidate rows are sent with
    the pending pivots applied, and every `depth` pivots -- and when the solve ends -- all pending eliminations are carried
    out in order.  Slot layout and collective unchanged: a rank with delayed updates and one without produce the same bytes."""

    def __init__(self, *a, depth=4):
        super().__init__(*a)
        self.depth = depth

    def begin(self, precision, max_pivots, check_cycles=False):
        self.rhs = self.m[:, 0].copy()  # current
        self.obj = self.m[0].copy()     # current (columns 1..)
        self.pend = []
        super().begin(precision, max_pivots, check_cycles)

    def _column_now(self, c):
        v = self.m[:, c].copy()
        for (prow, pcol, pn, nz, colv, nq) in self.pend:
            act = np.abs(colv) > 1e-16
            if prow >= 0:
                act[prow] = False
            if c == pcol:
                v[act] = nq[act]
            elif nz[c]:
                v[act] = v[act] - colv[act] * pn[c]
            if prow >= 0:
                v[prow] = nq[prow] if c == pcol else (pn[c] if nz[c] else 0.0)
        return v

    def _apply_to_row(self, x, i, p):
        prow, pcol, pn, nz, colv, nq = p
        if i == prow:
            x[1:] = np.where(nz[1:], pn[1:], 0.0)
            x[pcol] = nq[i]
        elif abs(colv[i]) > 1e-16:
            nzi = np.flatnonzero(nz[1:]) + 1
            x[nzi] = x[nzi] - colv[i] * pn[nzi]
            x[pcol] = nq[i]

    def _row_now(self, i):
        x = self.m[i].copy()
        for p in self.pend:
            self._apply_to_row(x, i, p)
        return x

    def _flush(self):
        for p in self.pend:
            for i in range(self.h):
                self._apply_to_row(self.m[i], i, p)
        if self.pend:
            assert np.array_equal(self.m[0, 1:].view(np.int64), self.obj[1:].view(np.int64))
        self.pend = []

    def _finish(self):
        self._flush()
        self.m[:, 0] = self.rhs

    def _scan(self):
        p = self.precision
        obj = self.obj[1:]
        cand = np.where(obj > p)[0]
        self.la = int(cand[np.argmax(obj[cand])]) + 1 if cand.size else 0
        self.lav = self._column_now(self.la) if self.la else None
        self.c_ratio, self.c_rhs = (INF, NONE), (INF, NONE)
        for r in range(1, self.h):
            g = r + self.base
            rhs = self.rhs[r]
            if rhs < -p and _better(rhs, g, *self.c_rhs):
                self.c_rhs = (rhs, g)
            if self.la:
                v = self.lav[r]
                if v > p:
                    ratio = rhs / v
                    if ratio < INF:
                        key = -INF if ratio <= p else ratio
                        if _better(key, g, *self.c_ratio):
                            self.c_ratio = (key, g)

    def select(self):
        s = self.send.numpy()
        s[:] = 0.0
        (kr, ir), (kn, inn) = self.c_ratio, self.c_rhs
        s[0:4] = (kr, float(ir), kn, float(inn))
        n = self.w - 1
        if self.status < 0:
            if ir != NONE:
                s[4] = self.rhs[ir - self.base]
                s[HDR:HDR + n] = self._row_now(ir - self.base)[1:]
            if inn != NONE:
                s[5] = self.rhs[inn - self.base]
                s[HDR + self.pitch:HDR + self.pitch + n] = self._row_now(inn - self.base)[1:]

    def apply(self):
        if self.status >= 0:
            return
        R = self.recv.numpy().reshape(self.nranks, self.slot)
        p, n = self.precision, self.w - 1

        def stop(status, result):
            self.status, self.result = status, result
            self._finish()

        while True:
            if not (self.iter < self.max_pivots):
                return stop(3, math.nan)
            if self.phase == 1:
                best = (INF, NONE)
                for g in range(self.nranks):
                    if _better(R[g, 2], int(R[g, 3]), *best):
                        best = (R[g, 2], int(R[g, 3]))
                if best[1] == NONE:
                    self.phase, self.iter, self.history = 2, 0.0, []
                    continue
                row = best[1]
                slot = R[self._owner(row)]
                prow, rhs_row = slot[HDR + self.pitch:HDR + self.pitch + n].copy(), slot[5]
                col, mx = 0, -INF
                for c in range(1, self.w):
                    coefficient = prow[c - 1]
                    if coefficient < -p:
                        ratio = -self.obj[c] / coefficient
                        if ratio > mx:
                            mx, col = ratio, c
                if col == 0:
                    return stop(1, math.nan)
                colv = self._column_now(col)
                break
            col = self.la
            if col == 0:
                return stop(0, _round_to_precision(self.rhs[0], p))
            best = (INF, NONE)
            for g in range(self.nranks):
                if _better(R[g, 0], int(R[g, 1]), *best):
                    best = (R[g, 0], int(R[g, 1]))
            if best[1] == NONE:
                return stop(2, float(col))
            row = best[1]
            slot = R[self._owner(row)]
            prow, rhs_row = slot[HDR:HDR + n].copy(), slot[4]
            colv = self.lav  # (the look-ahead priced exactly this column)
            break
        if self._has_cycle(row, col):  # :98,137: this pivot is not carried out, the pending ones are
            return stop(3, math.nan)
        full = np.concatenate(([0.0], prow))  # (column 0, the RHS, is handled on its own below)
        q = full[col]
        leaving, entering = self.var[self.w + row], self.var[col]
        self.var[self.w + row], self.var[col] = entering, leaving
        self.pos[leaving], self.pos[entering] = col, self.w + row
        nz = np.abs(full) > 1e-16
        pn = np.where(nz, full / q, 0.0)
        lrow = row - self.base if self.bounds[self.rank] <= row < self.bounds[self.rank + 1] else -1
        nq = -colv / q
        act = np.abs(colv) > 1e-16
        if lrow >= 0:
            nq[lrow] = 1.0 / q
            act[lrow] = False
        if abs(rhs_row) > 1e-16:
            pn_rhs = rhs_row / q
            self.rhs[act] = self.rhs[act] - colv[act] * pn_rhs
            if lrow >= 0:
                self.rhs[lrow] = pn_rhs
        elif lrow >= 0:
            self.rhs[lrow] = 0.0
        if act[0]:
            nzi = np.flatnonzero(nz)
            self.obj[nzi] = self.obj[nzi] - colv[0] * pn[nzi]
            self.obj[col] = nq[0]
        self.pend.append((lrow, col, pn, nz, colv.copy(), nq))
        self.iter += 1.0
        self.pivots += 1
        if len(self.pend) == self.depth:
            self._flush()
        self._scan()
