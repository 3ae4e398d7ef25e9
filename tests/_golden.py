"""Decoders for tests/golden/*.json.gz (data emitted by oracle/tools/gen_golden.py)."""
import base64
import functools
import gzip
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _dec(s, dtype):
    return np.frombuffer(base64.b64decode(s), dtype=dtype).copy()


def _num(x):
    return float(x) if isinstance(x, str) else x


@functools.lru_cache(None)
def records(kind):
    with gzip.open(os.path.join(GOLDEN, f"simplex_{kind}.json.gz")) as f:
        return json.load(f)["records"]


def label(rec):
    if rec["kind"] == "case":
        return rec["name"]
    return "%s-%dx%d-s%d" % (rec["kind"], rec["M"], rec["N"], rec["seed"])


def sha256(arr):
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()


def initial_matrix(rec, oracle=None, dense_gen=None):
    """Initial row-major tableau of a record (COO-decoded, or regenerated for dense records)."""
    w, h = rec["width"], rec["height"]
    if "init_coo" in rec:
        m = np.zeros(w * h, np.float64)
        m[_dec(rec["init_coo"]["idx"], np.int32)] = _dec(rec["init_coo"]["val"], np.float64)
    else:
        gen = dense_gen if dense_gen is not None else oracle.dense_lp
        m = gen(rec["M"], rec["N"], rec["seed"])
    assert sha256(m) == rec["init_sha256"], "initial tableau differs from the reference's"
    return m


def expected(rec):
    return dict(status=rec["status"], result=_num(rec["result"]), n_pivots=rec["n_pivots"],
                pivots=_dec(rec["pivots"], np.int32).reshape(-1, 2), pos=_dec(rec["pos"], np.int32),
                var=_dec(rec["var"], np.int32), col0=_dec(rec["col0"], np.float64), final_sha256=rec["final_sha256"])


def options(rec):
    o = rec["options"]
    return dict(precision=o["precision"], max_pivots=_num(o["maxPivots"]), check_cycles=o["checkCycles"])


def identity_perms(rec):
    n = rec["width"] + rec["height"]
    return np.arange(n, dtype=np.int32), np.arange(n, dtype=np.int32)


def same_number(a, b):
    return (a != a and b != b) or a == b
