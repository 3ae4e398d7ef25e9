"""Worker of the multi-process row-sharded tests: one rank of a gloo (or nccl) group.
usage: python -m tests._shard_worker <numpy|numpy-delayed<depth>|hip|hip-native|hip-rccl> <M> <N> <seed> <out.npz> [max_pivots [digest|phase1]]
(hip-native: the library's own loop, yalps_shard_run, with the host transport carried by gloo; hip-rccl: the same loop
over the library's RCCL communicator, batches of 64 pivots captured into a hipGraph)
(RANK/WORLD_SIZE/MASTER_* in env).  `digest`: instead of the assembled tableau, rank 0 saves the SHA-256 of the
objective row and of every rank's block of rows (full-size runs: the tableau is 2.1 GB).  `phase1`: the input of
tests/test_hip_parity.py's sweep cases (one row "-a x <= -b", exact zeros) instead of the seed's parity.
`c5:<variant>`: the input of tests/_c5.py (BASELINE config 5); rank 0 saves SHA-256 digests per block of 512 rows with
their global row numbers, every rank's copy of the objective row, column 0 and the basis.
`npy:<file>[:check]`: the (M+1) x (N+1) tableau in <file> instead of a generated one; `:check` = options.checkCycles."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import time
    t_start = time.perf_counter()

    def lap(what):  # (timings of the full-size runs: stderr, shown by the tests on failure or under YALPS_TEST_LOG_DIR)
        print("[shard_worker rank %s] %6.1f s %s" % (os.environ.get("RANK"), time.perf_counter() - t_start, what), file=sys.stderr, flush=True)

    kind, M, N, seed, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    max_pivots = float(sys.argv[6]) if len(sys.argv) > 6 else float("inf")
    digest = len(sys.argv) > 7 and sys.argv[7] == "digest"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import _oracle
    from yalps_amd import sharded
    w, h = N + 1, M + 1
    c5 = sys.argv[7][3:] if len(sys.argv) > 7 and sys.argv[7].startswith("c5:") else None
    npy = sys.argv[7][4:] if len(sys.argv) > 7 and sys.argv[7].startswith("npy:") else None
    check_cycles = bool(npy) and npy.endswith(":check")
    if npy:
        m = np.load(npy[:-6] if check_cycles else npy).astype(np.float64).reshape(-1)
        assert m.size == w * h
    elif c5:
        from tests import _c5
        from yalps_amd import _native
        assert (M, N, seed) == (_c5.M, _c5.N, _c5.SEED)
        m = _c5.make_input(_native.dense_lp, c5)
    else:
        m = _oracle.load().dense_lp(M, N, seed)
    if c5 or npy:
        pass
    elif digest:  # (the full-size test's input: one row "-a x <= -b", so that the first pivot is a phase-1 pivot)
        m.reshape(h, w)[h // 3] *= -1.0
    elif len(sys.argv) > 7 and sys.argv[7] == "phase1":
        A = m.reshape(h, w)
        A[h // 3] *= -1.0
        A[5::7, 3::5] = 0.0  # exact zeros: untouched rows, flushed pivot-row entries
    elif seed % 2:  # make some right-hand sides negative so that phase 1 runs too
        m.reshape(h, w)[1::3, 0] *= -0.05
    lap("input generated")
    bounds = sharded.partition(h, world)
    ident = np.arange(w + h, dtype=np.int32)
    local = sharded.local_rows(m, w, h, bounds, rank)
    if kind == "numpy":
        from tests._shard_numpy import NumpyShardOps
        ops = NumpyShardOps(local, w, bounds, rank, h, ident, ident.copy())
    elif kind.startswith("numpy-delayed"):  # "numpy-delayed<depth>": the delayed row updates of dshard_kernel, on the CPU
        from tests._shard_numpy import NumpyDelayedShardOps
        ops = NumpyDelayedShardOps(local, w, bounds, rank, h, ident, ident.copy(), depth=int(kind[len("numpy-delayed"):] or 4))
    else:
        ops = sharded.HipShardOps(local, w, bounds, rank, h, ident, ident.copy(), device=0, private_stream=kind in ("hip-native", "hip-rccl"))
    del m
    lap("shard uploaded")
    if kind in ("hip-native", "hip-rccl"):  # (hip-rccl: one rank per GPU -- RCCL refuses ranks that share a device)
        transport = "host" if kind == "hip-native" else "rccl"
        ncomm = sharded.native_comm(ops.ctx, rank, world, transport=transport)
        status, result, pivots = sharded.sharded_simplex_native(ops, ncomm, max_pivots=max_pivots, check_every=8 if kind == "hip-native" else 64,
                                                                check_cycles=check_cycles)
        assert ncomm.info()["transport"] == transport and int(ncomm.info()["collectives"]) >= pivots
        ncomm.close()
    else:
        comm = sharded.TorchComm()
        status, result, pivots = sharded.sharded_simplex(ops, comm, max_pivots=max_pivots, check_every=8 if max_pivots > 8 else 1,
                                                         check_cycles=check_cycles)
    lap("solve done: %s, %d pivots" % (status, pivots))
    kernel = kind if kind.startswith("numpy") else ops.tab.info()["streaming"]
    lm, pos, var = ops.download()
    lap("downloaded")
    if c5:
        lm = lm.reshape(-1, w)
        mine = (_c5.digest_rows(lm[0:1], 0), _c5.digest_rows(lm[1:], bounds[rank]), lm[1:, 0].copy())
        parts = [None] * world
        lap("digests")
        dist.all_gather_object(parts, mine)
        if rank == 0:
            dig = [d for p in parts for d in p[0] + p[1]]
            np.savez(out, lo=[d[0] for d in dig], hi=[d[1] for d in dig], sha=[d[2] for d in dig],
                     col0=np.concatenate([lm[0:1, 0]] + [p[2] for p in parts]), pos=pos, var=var, status=status, result=result,
                     pivots=pivots, kernel=kernel)
        ops.close()
        dist.barrier()
        dist.destroy_process_group()
        return
    if digest:
        import hashlib
        lm = lm.reshape(-1, w)
        mine = (hashlib.sha256(lm[0].tobytes()).hexdigest(), hashlib.sha256(lm[1:].tobytes()).hexdigest())
        parts = [None] * world
        dist.all_gather_object(parts, mine)
        if rank == 0:
            np.savez(out, row0=[p[0] for p in parts], blocks=[p[1] for p in parts], bounds=bounds, pos=pos, var=var,
                     status=status, result=result, pivots=pivots, kernel=kernel)
        ops.close()
        dist.barrier()
        dist.destroy_process_group()
        return
    # assemble the global tableau on rank 0
    parts = [None] * world
    dist.all_gather_object(parts, (lm, bounds[rank], bounds[rank + 1]))
    if rank == 0:
        full = np.zeros((h, w))
        for r, (pm, lo, hi) in enumerate(parts):
            pm = pm.reshape(-1, w)
            if r == 0:
                full[0] = pm[0]
            full[lo:hi] = pm[1:]
        np.savez(out, matrix=full.reshape(-1), pos=pos, var=var, status=status, result=result, pivots=pivots, kernel=kernel)
    ops.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
