"""Pins oracle/simplex_oracle.c to the reference: every golden record produced by
the reference's src/simplex.ts must be reproduced bit for bit (status, result,
pivot sequence, permutations, RHS column, whole-matrix digest)."""
import numpy as np
import pytest

from tests import _golden as G

ALL = [pytest.param(r, id=G.label(r)) for kind in ("cases", "mixed", "dense") for r in G.records(kind)
       if not (r["kind"] == "dense" and r["M"] > 1024)]
BIG = [pytest.param(r, id=G.label(r)) for r in G.records("dense") if r["M"] > 1024]


def _check(oracle, rec):
    m = G.initial_matrix(rec, oracle)
    pos, var = G.identity_perms(rec)
    exp = G.expected(rec)
    status, result, npiv, trace = oracle.simplex(m, rec["width"], rec["height"], pos, var,
                                                 trace_cap=exp["n_pivots"] + 8, **G.options(rec))
    assert status == exp["status"]
    assert G.same_number(result, exp["result"])
    assert npiv == exp["n_pivots"]
    assert np.array_equal(trace, exp["pivots"])
    assert np.array_equal(pos, exp["pos"]) and np.array_equal(var, exp["var"])
    assert G.sha256(m) == exp["final_sha256"]
    assert np.array_equal(m.reshape(rec["height"], rec["width"])[:, 0].view(np.int64), exp["col0"].view(np.int64))


@pytest.mark.parametrize("rec", ALL)
def test_oracle_reproduces_reference(oracle, rec):
    _check(oracle, rec)


@pytest.mark.parametrize("rec", BIG)
def test_oracle_reproduces_reference_c2(oracle, rec):
    """BASELINE config 2 (2048x2048 dense): 3923 pivots, -1022.09813705."""
    _check(oracle, rec)
    assert rec["n_pivots"] == 3923 and rec["result"] == -1022.09813705


OMP_RECORDS = [pytest.param(r, id=G.label(r)) for r in G.records("dense") if 64 <= r["M"] <= 512] + \
              [pytest.param(r, id=G.label(r)) for r in G.records("cases") if r["height"] >= 64][:6]


@pytest.mark.parametrize("rec", OMP_RECORDS)
def test_row_parallel_oracle_reproduces_reference(rec):
    """liboracle_omp.so (the same source with -fopenmp: the elimination's row loop over threads; the all-core CPU
    baseline of bench.py) is pinned by the same golden records."""
    from tests import _oracle
    omp = _oracle.load(omp=True)
    assert omp.set_threads(4) == 4
    _check(omp, rec)


def test_round_to_precision_js_semantics(oracle):
    # Math.round rounds halves toward +inf (src/util.ts:1-4)
    assert oracle.round_to_precision(-14666.666666666668, 1e-8) == -14666.66666667
    assert oracle.round_to_precision(2.5, 1.0) == 3.0
    assert oracle.round_to_precision(-2.5, 1.0) == -2.0
    assert oracle.round_to_precision(0.0, 1e-8) == 0.0
    assert np.isnan(oracle.round_to_precision(1.0, 0.0))


NP_RECORDS = [pytest.param(r, id=G.label(r)) for kind in ("cases", "mixed", "dense") for r in G.records(kind)
              if not r["options"]["checkCycles"] and not (r["kind"] == "dense" and r["M"] > 512)]


@pytest.mark.parametrize("rec", NP_RECORDS)
def test_numpy_restatement_reproduces_reference(oracle, rec):
    """tests/_np_simplex.py (the vectorised restatement used for the full-size GPU checks) against
    the reference's golden records: status, result, pivot count, permutations, whole matrix."""
    from tests import _np_simplex as NP
    m = G.initial_matrix(rec, oracle)
    pos, var = G.identity_perms(rec)
    exp = G.expected(rec)
    o = G.options(rec)
    status, result, npiv = NP.simplex(m, rec["width"], rec["height"], pos, var, o["precision"], o["max_pivots"])
    assert (status, npiv) == (exp["status"], exp["n_pivots"]) and G.same_number(result, exp["result"])
    assert np.array_equal(pos, exp["pos"]) and np.array_equal(var, exp["var"])
    assert G.sha256(m) == exp["final_sha256"]
