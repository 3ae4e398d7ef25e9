"""Persistent launches need their whole grid on the chip at once.  Between processes of this library that is
arranged by a per-device lock file (yalps_hip.hip DeviceLock), inside a launch every wait is bounded by the
real-time counter, and a launch that gives up is reported and retried a few solves later -- not switched off for the
life of the context.  (the reference has nothing comparable: one JS thread, src/simplex.ts.)"""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from tests import _golden as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CHILD = r"""
import sys, time, json, numpy as np
sys.path.insert(0, %(root)r)
from tests import _golden as G, _oracle
from yalps_amd import _native as n
rec = next(r for r in G.records('dense') if r['M'] == 1024)
exp = G.expected(rec)
m = G.initial_matrix(rec, _oracle.load(), dense_gen=n.dense_lp); pos, var = G.identity_perms(rec)
ctx = n.Context(0); t = n.DeviceTableau(ctx, rec['width'], rec['height'])
open(%(ready)r %% sys.argv[1], 'w').write('x')
while not all(__import__('os').path.exists(%(ready)r %% k) for k in ('0', '1')):
    time.sleep(0.01)
paths, worst = [], 0.0
for _ in range(%(solves)d):
    t.upload(m, rec['height'], pos, var)
    t0 = time.perf_counter()
    st, res, piv, _ = t.solve(max_pivots=float('inf'))
    worst = max(worst, time.perf_counter() - t0)
    info = t.info()
    paths.append(info['last_path'])
    gm, gp, gv = t.download()
    assert (st, res, piv) == (exp['status'], exp['result'], exp['n_pivots']), (st, res, piv)
    assert G.sha256(gm) == exp['final_sha256'] and np.array_equal(gp, exp['pos']) and np.array_equal(gv, exp['var'])
print(json.dumps({'paths': sorted(set(paths)), 'giveups': int(info['giveups']), 'worst_s': worst}))
"""


def test_two_processes_share_the_gpu_without_stalling(tmp_path):
    """Two processes each solve the 1025 x 1025 golden LP 12 times on the same GPU, started together: every solve takes
    the resident kernel (last_path == "resident", no give-up), every result is the reference's bit for bit, and no
    solve takes anywhere near a second (599 pivots are ~4 ms; a grid waiting for CUs it cannot get would spin)."""
    import json
    ready = str(tmp_path / "ready%s")
    code = CHILD % {"root": ROOT, "ready": ready, "solves": 12}
    procs = [subprocess.Popen([sys.executable, "-c", code, str(k)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for k in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so + se
        rec = json.loads(so.strip().splitlines()[-1])
        assert rec["paths"] == ["resident"] and rec["giveups"] == 0, (rec, se)
        assert rec["worst_s"] < 0.5, rec
        assert "gave up" not in se, se


def test_give_up_is_reported_and_the_path_comes_back(tmp_path):
    """A persistent launch that reports a failed hand-off (forced: YALPS_HIP_RESIDENT_FAULT declares the 2nd persistent
    launch of the context failed) falls back within that solve, says so on stderr and in yalps_tableau_info, leaves the
    resident path alone for the next 8 solves of the context and then uses it again."""
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "from tests import _golden as G, _oracle\n"
        "from yalps_amd import _native as n\n"
        "rec = next(r for r in G.records('dense') if r['M'] == 256)\n"
        "m = G.initial_matrix(rec, _oracle.load(), dense_gen=n.dense_lp); pos, var = G.identity_perms(rec)\n"
        "exp = G.expected(rec)\n"
        "ctx = n.Context(0); t = n.DeviceTableau(ctx, rec['width'], rec['height'])\n"
        "paths = []\n"
        "for k in range(11):\n"
        "    t.upload(m, rec['height'], pos, var)\n"
        "    st, res, piv, _ = t.solve(max_pivots=float('inf')); info = t.info(); gm, gp, gv = t.download()\n"
        "    assert (st, res, piv) == (exp['status'], exp['result'], exp['n_pivots']), (k, st, res, piv)\n"
        "    assert G.sha256(gm) == exp['final_sha256'] and np.array_equal(gp, exp['pos'])\n"
        "    paths.append(info['last_path'])\n"
        "print(paths, info['giveups'])\n"
        "assert paths[0] == 'resident+inplace' and all(p == 'inplace' for p in paths[1:9]) and paths[9:] == ['resident'] * 2, paths\n"
        "assert int(info['giveups']) == 1, info\n"
        "print('ok')\n" % ROOT)
    env = dict(os.environ, YALPS_HIP_RESIDENT_CHUNK="40", YALPS_HIP_RESIDENT_FAULT="2")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr
    assert out.stderr.count("gave up waiting for its grid") == 1, out.stderr


def test_batch_solve_rejects_bad_cut_lists():
    """yalps_batch_solve checks everything the kernel will index with before it enqueues anything: cut_offsets[0] == 0,
    non-decreasing offsets, at most max_cuts per node, cut variables inside [0, width + root height)."""
    import ctypes as C
    from yalps_amd import _native as nat
    w, h0 = 5, 4
    ctx = nat.Context(0)
    b = nat.NodeBatch(ctx, w, h0, 3, 4)
    try:
        m = nat.dense_lp(h0 - 1, w - 1, 3)
        ident = np.arange(w + h0, dtype=np.int32)
        b.set_root(m, ident, ident.copy())

        def call(off, sign, var, val):
            off, sign, var = (np.asarray(a, np.int32) for a in (off, sign, var))
            val = np.asarray(val, np.float64)
            st, res = np.zeros(4, np.int32), np.zeros(4, np.float64)
            return nat.lib().yalps_batch_solve(b.handle, len(off) - 1, off.ctypes.data, sign.ctypes.data, var.ctypes.data,
                                               val.ctypes.data, 1e-8, 100.0, st.ctypes.data, res.ctypes.data, None, None)
        assert call([0, 1], [1], [1], [0.5]) == 0
        assert call([1, 2], [1, 1], [1, 1], [0.5, 0.5]) == -1           # offsets must start at 0
        assert call([0, 2, 1], [1, 1], [1, 1], [0.5, 0.5]) == -1        # decreasing
        assert call([0, 4], [1] * 4, [1] * 4, [0.5] * 4) == -1          # more than max_cuts on one node
        assert call([0, 1], [1], [w + h0], [0.5]) == -1                 # unknown variable
        assert call([0, 1], [1], [-1], [0.5]) == -1
        assert b"unknown variable" in nat.lib().yalps_last_error()
    finally:
        b.close()
        ctx.close()


def test_lock_wait_is_bounded_and_the_lock_file_is_not_a_trap(tmp_path):
    """(a) the per-device lock is taken with LOCK_NB and a deadline -- while a stranger holds the file, a solve
    falls back to the launch-per-pivot kernels (bit-exact), says so and counts a give-up instead of blocking for ever;
    (b) a lock path that was pre-planted as a symbolic link is refused (O_NOFOLLOW, regular single-link file of this user),
    the link's target is left alone, solves go on without the inter-process lock."""
    import fcntl
    lock_dir = tmp_path / "locks"
    lock_dir.mkdir(mode=0o700)
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "from tests import _golden as G, _oracle\n"
        "from yalps_amd import _native as n\n"
        "rec = next(r for r in G.records('dense') if r['M'] == 256)\n"
        "m = G.initial_matrix(rec, _oracle.load(), dense_gen=n.dense_lp); pos, var = G.identity_perms(rec)\n"
        "exp = G.expected(rec)\n"
        "ctx = n.Context(0); t = n.DeviceTableau(ctx, rec['width'], rec['height'])\n"
        "t.upload(m, rec['height'], pos, var)\n"
        "st, res, piv, _ = t.solve(max_pivots=float('inf')); info = t.info(); gm, gp, gv = t.download()\n"
        "assert (st, res, piv) == (exp['status'], exp['result'], exp['n_pivots']), (st, res, piv)\n"
        "assert G.sha256(gm) == exp['final_sha256'] and np.array_equal(gp, exp['pos'])\n"
        "print('path', info['last_path'], 'giveups', info['giveups'], 'lock_giveups', info['lock_giveups'])\n" % ROOT)
    env = dict(os.environ, YALPS_HIP_LOCK_DIR=str(lock_dir), YALPS_HIP_LOCK_WAIT_MS="150")

    def run():
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        return out

    out = run()  # creates the lock file; nobody holds it
    assert "path resident giveups 0 lock_giveups 0" in out.stdout, out.stdout + out.stderr
    files = list(lock_dir.iterdir())
    assert len(files) == 1 and files[0].name.startswith("yalps_hip_") and (files[0].stat().st_mode & 0o777) == 0o600, files
    with open(files[0], "r+") as held:  # (a) a stranger sits on the lock
        fcntl.flock(held, fcntl.LOCK_EX)
        t0 = time.perf_counter()
        out = run()
        assert time.perf_counter() - t0 < 120
        fcntl.flock(held, fcntl.LOCK_UN)
    assert "path streaming giveups 1 lock_giveups 1" in out.stdout, out.stdout + out.stderr
    assert "did not get the device's lock file in time" in out.stderr, out.stderr
    victim = tmp_path / "victim.txt"  # (b) the lock path is a link to a file of ours
    victim.write_text("precious")
    victim.chmod(0o600)
    files[0].unlink()
    files[0].symlink_to(victim)
    out = run()
    assert "cannot use" in out.stderr and "as a lock file" in out.stderr, out.stderr
    assert "path resident giveups 0 lock_giveups 0" in out.stdout, out.stdout + out.stderr
    assert victim.read_text() == "precious" and (victim.stat().st_mode & 0o777) == 0o600
