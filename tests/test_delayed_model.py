"""The delayed row updates of stream2_kernel / stream3_kernel / dshard_kernel, as a numpy model of their data flow
(tests/_np_delayed.py), against the pinned one-sweep-per-pivot restatement (tests/_np_simplex.py): every bit of the tableau,
the basis, status, result and pivot count, for every depth the kernels use -- no GPU involved.  What the GPU tests then
show is that the kernels are this model; what this test shows is that the model is the reference."""
import numpy as np
import pytest

from tests import _golden as G
from tests import _np_delayed as D
from tests import _np_simplex as NP

CASES = [
    # M, N, seed, style, budget
    (30, 40, 1, "plain", 8192.0),
    (25, 60, 2, "phase1", 200.0),
    (60, 25, 3, "phase1", 57.0),
    (17, 90, 4, "zeros", 41.0),
    (48, 48, 5, "degenerate", 23.0),
    (12, 200, 6, "zeros", 8192.0),
    (90, 14, 7, "phase1", 8192.0),
]


def _lp(oracle, M, N, seed, style):
    w, h = N + 1, M + 1
    m = oracle.dense_lp(M, N, seed)
    A = m.reshape(h, w)
    if style in ("phase1", "zeros", "degenerate"):
        A[h // 3] *= -1.0  # "-a x <= -b": phase 1 first
    if style in ("zeros", "degenerate"):
        A[5::7, 3::5] = 0.0  # exact zeros: untouched rows, flushed pivot-row entries
    if style == "degenerate":
        A[2::9, 0] = 0.0  # ratios <= precision
    return m, w, h


@pytest.mark.parametrize("depth", [1, 2, 3, 4, 6, 8])
@pytest.mark.parametrize("M,N,seed,style,budget", CASES)
def test_delayed_model_equals_one_sweep_per_pivot(oracle, M, N, seed, style, budget, depth):
    m, w, h = _lp(oracle, M, N, seed, style)
    ident = np.arange(w + h, dtype=np.int32)
    ref, rpos, rvar = m.copy(), ident.copy(), ident.copy()
    est, eres, epiv = NP.simplex(ref, w, h, rpos, rvar, max_pivots=budget)
    got, gpos, gvar = m.copy(), ident.copy(), ident.copy()
    st, res, npiv = D.simplex_delayed(got, w, h, gpos, gvar, depth, max_pivots=budget)
    assert (st, npiv) == (est, epiv) and G.same_number(res, eres)
    assert np.array_equal(gpos, rpos) and np.array_equal(gvar, rvar)
    assert np.array_equal(got.view(np.int64), ref.view(np.int64))


def test_delayed_model_against_the_c_oracle(oracle):
    """... and against oracle/simplex_oracle.c itself (the restatement the golden records pin), whole solves."""
    for M, N, seed, style in ((40, 55, 11, "plain"), (33, 70, 12, "phase1"), (70, 20, 13, "zeros")):
        m, w, h = _lp(oracle, M, N, seed, style)
        ident = np.arange(w + h, dtype=np.int32)
        ref, rpos, rvar = m.copy(), ident.copy(), ident.copy()
        est, eres, epiv, _ = oracle.simplex(ref, w, h, rpos, rvar, max_pivots=3000.0)
        got, gpos, gvar = m.copy(), ident.copy(), ident.copy()
        st, res, npiv = D.simplex_delayed(got, w, h, gpos, gvar, 8, max_pivots=3000.0)
        assert (st, npiv) == (est, epiv) and G.same_number(res, eres)
        assert np.array_equal(gpos, rpos) and np.array_equal(gvar, rvar)
        assert np.array_equal(got.view(np.int64), ref.view(np.int64))


def test_delayed_model_random_small_tableaux_all_statuses():
    """Random sign patterns (unbounded and infeasible outcomes among them), random sparsity, every depth."""
    rng = np.random.default_rng(20261004)
    seen = set()
    for case in range(300):
        h, w = int(rng.integers(3, 14)), int(rng.integers(3, 16))
        m = rng.uniform(-1, 1, (h, w))
        m[rng.random((h, w)) < 0.2] = 0.0
        if rng.random() < 0.6:
            m[1:, 0] = np.abs(m[1:, 0])
        m[0, 0] = 0.0
        m = m.reshape(-1)
        ident = np.arange(w + h, dtype=np.int32)
        ref, rpos, rvar = m.copy(), ident.copy(), ident.copy()
        est, eres, epiv = NP.simplex(ref, w, h, rpos, rvar, max_pivots=60.0)
        got, gpos, gvar = m.copy(), ident.copy(), ident.copy()
        st, res, npiv = D.simplex_delayed(got, w, h, gpos, gvar, int(rng.integers(1, 9)), max_pivots=60.0)
        seen.add(est)
        assert (st, npiv) == (est, epiv) and G.same_number(res, eres), case
        assert np.array_equal(gpos, rpos) and np.array_equal(gvar, rvar), case
        assert np.array_equal(got.view(np.int64), ref.view(np.int64)), case
    assert {"optimal", "infeasible", "unbounded"} <= seen, seen
