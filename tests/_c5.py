"""BASELINE config 5 (16385 x 16385 dense LP, 2.1 GB) as a parity case: inputs, the oracle run and the digests that the
test process and its worker processes compare.  Test-side only.

The checker is oracle/liboracle_omp.so -- oracle/simplex_oracle.c (pinned by the reference's golden records,
tests/test_oracle_golden.py) with the elimination's row loop (src/simplex.ts:27-38) split over threads: every thread
owns whole rows, the arithmetic per element is the scalar build's.
"""
import hashlib
import os

import numpy as np

M = N = 16384
W, H = N + 1, M + 1
SEED = 42
# Pivot budget per phase (src/simplex.ts:69,109): no multiple of the delay depth (16 since round 3: 20 full flushes, then
# thirteen pivots pending when the loop stops; 41 + 5 at depth 8); with 100 pivots per persistent launch
# (YALPS_HIP_RESIDENT_CHUNK) three launch boundaries, each of them with pivots pending (100 = 6 * 16 + 4).
BUDGET = 333
CHUNK = 100
BLOCK = 512  # rows per SHA-256 block


def make_input(gen, variant):
    """dense-LP(16384,16384,42) from `gen(M, N, seed)` (the product's or the oracle's generator: same stream).
    "phase2": as it is -- feasible at the start, every pivot a phase-2 pivot (src/simplex.ts:66-103).
    "phase1": one row turned into "-a x <= -b" -- the start is infeasible and the budget is spent in phase 1
    (src/simplex.ts:106-142) -- and a lattice of exact zeros: rows whose pivot-column entry is 0 are skipped
    (:31), pivot-row entries that are 0 drop out of nonZeroColumns (:17-24)."""
    m = gen(M, N, SEED)
    if variant == "phase1":
        A = m.reshape(H, W)
        A[H // 3] *= -1.0
        A[5::7, 3::5] = 0.0
    else:
        assert variant == "phase2"
    return m


def threads():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(n, int(os.environ.get("YALPS_TEST_ORACLE_THREADS", "16"))))


def digest_rows(rows2d, first_row):
    """[(global first row, global end row, sha256 of the rows' bytes)] in blocks of BLOCK rows."""
    out = []
    for lo in range(0, rows2d.shape[0], BLOCK):
        blk = np.ascontiguousarray(rows2d[lo:lo + BLOCK])
        out.append((first_row + lo, first_row + lo + blk.shape[0], hashlib.sha256(blk.tobytes()).hexdigest()))
    return out


def check_digests(ref2d, digests):
    """Every (lo, hi, sha) against the same rows of the oracle's tableau; returns the row ranges that differ."""
    bad = []
    for lo, hi, sha in digests:
        if hashlib.sha256(np.ascontiguousarray(ref2d[lo:hi]).tobytes()).hexdigest() != sha:
            bad.append((int(lo), int(hi)))
    return bad


_cache = {}


def reference(variant, budget=BUDGET):
    """The oracle's run of `budget` pivots per phase on all host cores: (status, result, pivots, tableau, pos, var)."""
    key = (variant, budget)
    if key not in _cache:
        from tests import _oracle
        orc = _oracle.load(omp=True)
        orc.set_threads(threads())
        m = make_input(orc.dense_lp, variant)
        pos = np.arange(W + H, dtype=np.int32)
        var = pos.copy()
        status, result, pivots, trace = orc.simplex(m, W, H, pos, var, max_pivots=float(budget), trace_cap=4096)
        _cache[key] = dict(status=status, result=result, pivots=pivots, ref=m.reshape(H, W), pos=pos, var=var, trace=trace)
    return _cache[key]
