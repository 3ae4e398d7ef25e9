#!/usr/bin/env python3
"""bench.py -- fp64 pivots/sec on a dense tableau (BASELINE.json metric), MI355X.

A "step" = one complete two-phase simplex solve of dense-LP(2048,2048,seed) (BASELINE config 2:
tableau 2049 x 2049 fp64, 3923 pivots for seed 42) by the HIP path, starting from a pristine
copy of the tableau that is already resident in HBM.  value = pivots executed / wall time.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--size 2048] [--workload replicas|sharded]
  N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
         one process per GPU; every rank solves its own LP (seed 42 + rank) -- independent
         sub-problems shard with no data-path collective (weak scaling); torch.distributed
         (RCCL) only provides the barriers and the max-over-ranks reduction of the time.

  --workload sharded (optional, BASELINE config 5): ONE dense-LP(size,size,42) row-sharded over the
         N ranks (yalps_amd/sharded.py: one RCCL all-gather of candidates + candidate rows per pivot);
         a step = --pivots-per-step pivots of that solve; strong scaling.

  N > 1, default workload: after the replicas measurement every rank also starts a child process that joins a second
         process group and runs `--workload sharded --size 16384` (BASELINE config 5: ONE 16385 x 16385 tableau,
         rows sharded over the N GPUs); rank 0 adds that line's figures as "sharded_c5" to the same JSON line.  The
         children are separate processes with a time limit, so that nothing that goes wrong there -- RCCL init, graph
         capture, memory, a hang -- can change this run's return code or headline: it becomes "sharded_c5": {"error": ...}
         (--no-sharded-c5 skips it).

Extra objects on the JSON line (N == 1 / rank 0 only):
  roofline      the dominant (only) kernel of the timed region.  At 2049^2 the tableau fits on chip
                and that is resident_kernel: ONE launch = up to 4096 complete pivots (selection,
                pivot-row normalise, rank-1 elimination) on a register-resident tableau.  achieved =
                algorithmic bytes per launch (per pivot 16*(h-1)*w + 16*w + 8*(h-1) +
                8*(w-1) + 16*(h-1), times the pivots of the launch) / average launch duration from
                HIP events on the library's stream over the timed region.
  streaming_apply_only  the general HBM-streaming kernel (pivot_kernel) in APPLY mode, fixed pivot,
                back-to-back launches, HIP events (what a tableau that does not fit on chip gets).
  cpu_baseline  the CPU oracle (scalar C restatement of the reference, 1 thread) timed on the
                same LP on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_COPY_GBPS = 6290.0  # what a float4 copy reaches on this chip (same guide: 79 % of the spec peak)


def onchip_floor_us(T, J, R, clock_ghz=2.4):
    """(The round-2 MODEL of the floor, kept beside the measured one for comparison; `roofline.frac` no longer uses it.)
    What bounds resident_kernel<T,J,R> per pivot once the tableau sits in the register files: not HBM, but
    (a) the one exchange through the fabric that every pivot needs, priced with the guide's own list
        (MI355X_MICROARCH.md, "Persistent kernels: synchronisation and hand-off price list"):
        handoff-flag, drained sc1 payload + 16-byte flag, idle chip .................. 1.3 us
        handoff-payload, one dependent read of the winner's row (16-64 KB at the latency-bound
        12-20 GB/s per block: row bytes / 16 GB/s, never under one Infinity-Cache round trip 0.23 us)
    (b) the fp64 vector issue time of what cannot overlap the exchange: pivot-row normalisation, objective replica and
        the candidate row (3 row passes of 2*J doubles per lane: mul + sub, 4 cycles per wave64 fp64 instruction,
        T/256 waves per SIMD), and of the remaining R-1 rows where that exceeds the time the flags travel (0.5 us).
    floor = (a) + (b); frac = floor / measured.  A model, stated so that it can be checked -- not a measurement."""
    row_bytes = 16 * T * J
    exchange = 1.3 + max(row_bytes / 16e3, 0.23)
    waves_per_simd = T / 256.0
    per_row = 2 * J * 2 * 4 * waves_per_simd / (clock_ghz * 1e3)  # us: 2J doubles x (mul + sub) x 4 cycles
    critical = 3 * per_row
    rest = max(0.0, (R - 1) * per_row - 0.5)
    return {"us": exchange + critical + rest, "exchange_us": exchange, "valu_critical_us": critical, "valu_rest_us": rest,
            "valu_all_rows_us": (R + 2) * per_row, "model": "handoff-flag 1.3 us + winner's row %d B / 16 GB/s + fp64 issue of 3 row "
            "passes + what of the other %d rows exceeds 0.5 us of flag travel (4 cycles per wave64 fp64 op, %g waves/SIMD, %.1f GHz)"
            % (row_bytes, R - 1, waves_per_simd, clock_ghz)}


def algorithmic_bytes_per_pivot(h, w):
    # SURVEY.md section 8(d)
    return 16 * (h - 1) * w + 16 * w + 8 * (h - 1) + 8 * (w - 1) + 16 * (h - 1)


def measured_traffic(size, resident, pivots_per_launch):
    """(HBM bytes per launch, the profiles/ file they come from): PMC counters (FETCH_SIZE x2 + WRITE_SIZE, corrected as
    MI355X_MICROARCH.md prescribes), collected in separate rocprofv3 --pmc passes of this same
    workload and committed under profiles/ -- bench.py cannot run the profiler on itself."""
    name = "r02_pmc_traffic_resident.json" if resident else "r01_pmc_traffic.json"
    path = os.path.join(ROOT, "profiles", name)
    if size in (16384, 8192):  # the delayed in-place kernel at BASELINE config 5 and at 8193^2: round 3's PMC passes (per pivot of a bounded launch)
        name = "r03_pmc_traffic_delayed.json"
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            with open(path) as f:
                for case in json.load(f)["cases"]:
                    if case["tableau"] == "%dx%d" % (size + 1, size + 1):
                        return case["traffic_bytes_per_pivot"] * pivots_per_launch, "profiles/" + name
        return None, None
    if size != 2048 or not os.path.exists(path):
        return None, None
    with open(path) as f:
        rec = json.load(f)
    if resident:
        return rec["traffic_bytes_per_pivot"] * pivots_per_launch, "profiles/" + name
    return rec["traffic_bytes_per_launch"], "profiles/" + name


def measured_exchange_floor(native, ctx, nb, T, J, epochs=4000):
    """The resident kernels' per-pivot exchange and nothing else, measured on this chip in this run
    (yalps_ctx_exchange_floor, persistent_floor.hip: `nb` workgroups of T lanes, rows of 2 J T doubles): us per round of
    flag only | flag + dependent fetch of the winner's row | row published write-through + drained, flag, fetch."""
    units = J if (T, J) in ((512, 2), (512, 3), (256, 1), (256, 2)) else (2 if T == 512 else 1)
    out = {}
    for name, variant in (("flags_only_us", 0), ("flags_and_fetch_us", 2), ("publish_flags_fetch_us", 3)):
        native.exchange_floor(ctx, nb, T, units, 200, variant)  # warm-up
        out[name] = min(native.exchange_floor(ctx, nb, T, units, epochs, variant) for _ in range(3))
    out["lanes"], out["units"], out["workgroups"], out["row_bytes"] = T, units, nb, 16 * T * units
    return out


def cpu_baseline(M, N, seed, budget_pivots, threads=1):
    """Oracle (kind "port") on a bounded sample: the first `budget_pivots` pivots of the same LP
    (all of them by default: 3923 pivots of the 2049x2049 tableau take ~7-10 s on one core).
    threads > 1: the row-parallel build of the same source (liboracle_omp.so)."""
    from tests import _oracle
    orc = _oracle.load(omp=threads > 1)
    cores = orc.set_threads(threads) if threads > 1 else 1
    w, h = N + 1, M + 1
    m = orc.dense_lp(M, N, seed)
    pos = np.arange(w + h, dtype=np.int32)
    var = pos.copy()
    t0 = time.perf_counter()
    _, _, npiv, _ = orc.simplex(m, w, h, pos, var, max_pivots=budget_pivots)
    dt = time.perf_counter() - t0
    return {"value": npiv / dt, "unit": "pivots/s", "cores": cores, "kind": "port",
            "sample": "%d pivots of dense-LP(%d,%d,seed=%d), oracle/simplex_oracle.c -O2 -ffp-contract=off%s, %.1f s, host has %d cores"
                      % (npiv, M, N, seed, " -fopenmp (row loop of the elimination over %d threads)" % cores if threads > 1 else "",
                         dt, os.cpu_count())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=2048, help="M = N of dense-LP(M,N,seed)")
    ap.add_argument("--cpu-pivots", type=float, default=float("inf"),
                    help="oracle sample: first N pivots of the same LP (default: the whole solve, ~7 s; 0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="threads of the second, row-parallel oracle run (a 1-GPU box's CPU share; 0 = skip)")
    ap.add_argument("--sweep-launches", type=int, default=400)
    ap.add_argument("--workload", choices=("replicas", "sharded"), default="replicas")
    ap.add_argument("--pivots-per-step", type=int, default=256, help="sharded workload: pivots per step")
    ap.add_argument("--shard-rows", type=int, default=0,
                    help="sharded workload: M of dense-LP(M,size,seed) instead of size (one rank's share of a larger world, "
                         "measured on one GPU: e.g. 2048 rows of 16384 columns = a rank of 8)")
    ap.add_argument("--verify-pivots", type=int, default=48,
                    help="sharded workload: after the timed run, this many pivots from the fresh input are run again, every rank's rows "
                         "are hashed and rank 0 compares them with the CPU oracle's (the same run is the workload's cpu_baseline); 0 = skip")
    ap.add_argument("--no-sharded-c5", action="store_true", help="N > 1, default workload: skip the row-sharded config-5 measurement")
    ap.add_argument("--sharded-c5-size", type=int, default=16384, help="M = N of the row-sharded LP measured beside the replicas when N > 1")
    ap.add_argument("--sharded-c5-timeout", type=float, default=420.0, help="seconds the children of that measurement get")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # (rehearsal on a box with fewer GPUs than ranks: the ranks share GPU 0 and gloo carries the barriers;
    #  that is only for checking the multi-rank control flow, never for numbers)
    rehearsal = world > torch.cuda.device_count()
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    if args.workload == "sharded":
        return bench_sharded(args, torch, dist, rank, local_rank, world)

    from yalps_amd import _native
    M = N = args.size
    w, h = N + 1, M + 1
    seed = 42 + rank
    ctx = _native.Context(local_rank)
    pristine = _native.DeviceTableau(ctx, w, h)
    work = _native.DeviceTableau(ctx, w, h)
    m = _native.dense_lp(M, N, seed)
    ident = np.arange(w + h, dtype=np.int32)
    pristine.upload(m, h, ident, ident.copy())  # input resident in HBM before the timed region

    def step():
        work.copy_from(pristine)
        return work.solve(max_pivots=float("inf"))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        status, result, npiv, _ = step()
    barrier()
    t0 = time.perf_counter()
    pivots = 0
    gpu_ms = 0.0
    for _ in range(args.steps):
        status, result, npiv, ms = step()
        pivots += npiv
        gpu_ms += ms
    barrier()
    dt = time.perf_counter() - t0
    assert status == "optimal", status
    if seed == 42 and args.size == 2048:  # known answer of the reference
        assert (npiv, result) == (3923, -1022.09813705), (npiv, result)

    tot = torch.tensor([dt, float(pivots)], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
    if dist is not None:
        tmax = tot.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt_max, pivots_all = tmax[0].item(), tot[1].item()
    else:
        dt_max, pivots_all = dt, float(pivots)

    out = None
    if rank == 0:
        bpp = algorithmic_bytes_per_pivot(h, w)
        out = {
            "metric": "fp64 pivots/sec on dense m x n tableau", "value": pivots_all / dt_max, "unit": "pivots/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt_max / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "dense-LP(%d,%d,seed=42+rank): tableau %dx%d fp64, full two-phase simplex solve "
                                   "per step (%d pivots on rank 0), one independent LP per GPU" % (M, N, h, w, npiv),
                       "pivots_per_step": npiv, "objective_cell": result},
        }
        if rehearsal:
            out["rehearsal"] = "ranks share GPU 0 (fewer GPUs than ranks): control-flow check only"
        if world == 1:
            info = work.info()
            resident = info["last_path"] == "resident"
            inplace = info["last_path"] == "inplace"
            launches = int(info["last_resident_launches"]) if (resident or inplace) else npiv  # per step
            us_pivot = 1e3 * gpu_ms / pivots
            us_launch = 1e3 * gpu_ms / (launches * args.steps)
            bytes_launch = bpp * npiv / launches
            ach = bytes_launch / (us_launch * 1e-6) / 1e9
            traffic, traffic_source = measured_traffic(args.size, resident, bytes_launch / bpp) if (not inplace or args.size in (16384, 8192)) else (None, None)
            out["roofline"] = {
                "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                "frac_hbm": ach / HBM_PEAK_GBPS, "frac_of_copy_rate": ach / HBM_COPY_GBPS,
                "traffic": traffic, "traffic_source": traffic_source,
                "kernel": info["resident"].split(" ")[0] if resident else info["inplace"] if inplace else info["streaming"],
                "launches_per_step": launches, "avg_us": us_launch, "bytes_per_launch": bytes_launch,
                "us_per_pivot": us_pivot,
                "note": ("persistent kernel: one launch = up to %s pivots with the tableau resident in registers; achieved = "
                         "algorithmic bytes (SURVEY 8d, 16*h*w per pivot) / HIP-event time = frac_hbm, kept as algorithmic_equiv; real HBM "
                         "traffic is `traffic`; the bound that binds this kernel is the exchange through the fabric every pivot needs: "
                         "bound/achieved/peak/unit/frac are stated against onchip_floor, MEASURED in this run on this chip."
                         % info.get("chunk", "?")) if resident else
                        ("persistent in-place kernel: one launch = many pivots, rows streamed from HBM / Infinity Cache; "
                         "algorithmic bytes (SURVEY 8d) / HIP-event time") if inplace else
                        "one launch = one pivot; HIP events over the timed pivot loops / pivots"}
            if resident:
                # HBM does not bind a kernel that keeps the tableau in registers (measured traffic: a few per cent of the
                # algorithmic bytes): `frac` is the fraction of the bound that does bind it, the algorithmic figure stays
                # beside it under its own name
                T, J, R = (int(x) for x in info["resident"].split("<")[1].split(">")[0].split(",")[:3])
                model = onchip_floor_us(T, J, R)
                nb = int(info["workgroups"])
                meas = measured_exchange_floor(_native, ctx, nb, T, J)
                floor_us = meas["publish_flags_fetch_us"] + model["valu_critical_us"]
                rf = out["roofline"]
                rf["algorithmic_equiv"] = {"achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                                           "frac_of_copy_rate": ach / HBM_COPY_GBPS,
                                           "note": "what a streaming implementation would have to move / time; not an efficiency"}
                rf["onchip_floor"] = {
                    "us": floor_us, "measured_exchange": meas, "valu_critical_us": model["valu_critical_us"], "model_us_round2": model["us"],
                    "note": "floor = the exchange every pivot needs, measured in this run by exchange_floor_kernel (%d workgroups x %d lanes: each "
                            "publishes a %d-byte row write-through + drained, raises a 16-byte record, polls everybody's, fetches the winner's "
                            "row -- no tableau, no arithmetic) + the fp64 issue time of the three row passes that cannot overlap it"
                            % (nb, T, meas["row_bytes"])}
                # the block states ONE bound consistently: pivots/s against the pivots/s of the measured floor
                rf["bound"], rf["unit"] = "onchip", "pivots/s"
                rf["achieved"], rf["peak"] = 1e6 / us_pivot, 1e6 / floor_us
                rf["frac"] = floor_us / us_pivot
                rf["bound_detail"] = "on-chip: the measured exchange through the fabric + fp64 vector issue (onchip_floor); the HBM figure of SURVEY 8(d) is frac_hbm / algorithmic_equiv"
            if inplace and "delay_depth" in info and info["inplace"].startswith(("stream2_kernel", "stream3_kernel")):
                # delayed row updates: a row is streamed once per `delay_depth` pivots, so the algorithmic bytes / time
                # may exceed the HBM peak without being an efficiency; the roofline that binds is the traffic actually moved
                depth = int(info["delay_depth"])
                rf = out["roofline"]
                rf["algorithmic_equiv"] = {"achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                                           "frac_of_copy_rate": ach / HBM_COPY_GBPS,
                                           "note": "16*h*w bytes per pivot (SURVEY 8d) / time: what a one-sweep-per-pivot implementation would have to move; not an efficiency"}
                moved = ach / depth
                rf["achieved"], rf["frac"], rf["frac_of_copy_rate"] = moved, moved / HBM_PEAK_GBPS, moved / HBM_COPY_GBPS
                rf["delay_depth"] = depth
                rf["note"] = ("persistent in-place kernel with delayed row updates: every touched row is read and written once per %d "
                              "pivots (stream2_kernel / stream3_kernel); achieved = row traffic actually moved = algorithmic bytes / "
                              "depth / time (a lower bound of the HBM traffic: the per-pivot exchange adds a few per cent); the "
                              "per-pivot algorithmic figure is kept as algorithmic_equiv" % depth)
            work.copy_from(pristine)
            us_apply = work.bench_sweep(h // 2, w // 2, args.sweep_launches)
            out["streaming_apply_only"] = {
                "kernel": info["streaming"], "avg_us": us_apply, "achieved_GBps": bpp / (us_apply * 1e-6) / 1e9,
                "note": "the general (HBM-streaming) kernel in APPLY mode, fixed pivot, back-to-back launches"}
            if args.cpu_pivots > 0:
                out["cpu_baseline"] = cpu_baseline(M, N, seed, args.cpu_pivots)
                threads = min(args.cpu_threads, os.cpu_count() or 1)
                if threads > 1:
                    out["cpu_baseline_all_cores"] = cpu_baseline(M, N, seed, args.cpu_pivots, threads)
    work.close()
    pristine.close()
    ctx.close()
    if dist is not None and not args.no_sharded_c5:
        c5 = sharded_c5_in_children(args, dist, rank, local_rank, world)
        if out is not None:
            out["sharded_c5"] = c5
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out))


def sharded_c5_in_children(args, dist, rank, local_rank, world):
    """BASELINE config 5 beside the replicas (N > 1): every rank starts `bench.py --workload sharded --size 16384` as a
    child process; the children form a process group of their own (fresh port from rank 0) and run the row-sharded solve
    over RCCL.  Returns rank 0's summary of the child's JSON line, or {"error": ...}; never raises."""
    import socket
    import subprocess
    try:
        box = [None]
        if rank == 0:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                box[0] = sk.getsockname()[1]
        dist.broadcast_object_list(box, src=0)
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(box[0]), HSA_ENABLE_IPC_MODE_LEGACY="0")
        for k in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_USE_AGENT_STORE", "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE"):
            env.pop(k, None)  # (the children rendezvous through MASTER_ADDR / MASTER_PORT, not through the parent's agent)
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--workload", "sharded", "--size", str(args.sharded_c5_size),
               "--steps", str(max(1, min(args.steps, 10))), "--warmup", "1", "--pivots-per-step", str(args.pivots_per_step)]
        t0 = time.perf_counter()
        child = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        try:
            so, se = child.communicate(timeout=args.sharded_c5_timeout)
        except subprocess.TimeoutExpired:
            child.kill()
            so, se = child.communicate()
            return {"error": "timed out after %.0f s" % args.sharded_c5_timeout, "stderr_tail": se[-600:]}
        if child.returncode != 0:
            return {"error": "child exited with %d" % child.returncode, "stderr_tail": se[-600:]}
        if rank != 0:
            return None
        line = [ln for ln in so.splitlines() if ln.startswith("{")]
        if not line:
            return {"error": "no JSON line from the child", "stderr_tail": se[-600:]}
        rec = json.loads(line[-1])
        return {"value": rec["value"], "unit": rec["unit"], "us_per_pivot": rec["roofline"]["us_per_pivot"], "n_gpus": rec["n_gpus"],
                "scaling": rec["scaling"], "workload": rec["config"]["workload"], "exchange": rec["config"]["exchange"],
                "roofline": rec["roofline"], "parity": rec.get("parity"), "cpu_baseline": rec.get("cpu_baseline"), "rehearsal": rec.get("rehearsal"),
                "wall_s": time.perf_counter() - t0}
    except Exception as e:  # noqa: BLE001 -- by contract nothing here may change the headline or the return code
        return {"error": "%s: %s" % (type(e).__name__, e)}


def bench_sharded(args, torch, dist, rank, local_rank, world):
    """ONE tableau row-sharded over the ranks; every rank generates its own block of rows.  The pivot loop is the
    library's (yalps_shard_run): select kernel, ncclAllGather (RCCL over xGMI) on the same stream, apply kernel, a batch
    of pivots per hipGraph replay -- Python is entered once per timed region."""
    from yalps_amd import _native, sharded
    N = args.size
    M = args.shard_rows if args.shard_rows > 0 else N
    w, h = N + 1, M + 1
    bounds = sharded.partition(h, world)
    ident = np.arange(w + h, dtype=np.int32)
    # only this rank's rows are generated: the objective row + rows bounds[rank] .. bounds[rank + 1] of the stream
    local = np.concatenate([_native.dense_lp_rows(M, N, 42, 0, 1), _native.dense_lp_rows(M, N, 42, bounds[rank], bounds[rank + 1])])
    rehearsal = world > torch.cuda.device_count()  # (ranks share GPU 0: RCCL refuses; the host transport over gloo instead)

    def run(pivots):
        ops = sharded.HipShardOps(local, w, bounds, rank, h, ident, ident.copy(), device=local_rank, private_stream=True)
        comm = sharded.native_comm(ops.ctx, rank, world, transport="host" if rehearsal else "rccl")
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        status, result, npiv, gpu_ms = ops.run_native(comm, max_pivots=float(pivots), check_every=64)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        info = comm.info()
        info["kernel"] = ops.tab.info()["streaming"]
        comm.close()
        ops.close()
        return dt, npiv, status, gpu_ms, info

    run(args.pivots_per_step * max(args.warmup, 1))
    dt, npiv, status, gpu_ms, info = run(args.pivots_per_step * args.steps)
    verdict = verify_sharded(args, torch, dist, sharded, local, w, h, M, N, bounds, ident, rank, local_rank, world, rehearsal) if args.verify_pivots > 0 else None
    t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = t.item()
    if rank == 0:
        bpp = algorithmic_bytes_per_pivot(h, w)
        ach = bpp * npiv / dt / 1e9 / world
        out = {
            "metric": "fp64 pivots/sec on dense m x n tableau", "value": npiv / dt, "unit": "pivots/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "ONE dense-LP(%d,%d,seed=42), tableau %dx%d fp64, rows sharded over %d GPU(s), "
                                   "one all-gather of candidates+rows per pivot; %d pivots per step (status %s)"
                                   % (M, N, h, w, world, args.pivots_per_step, status),
                       "exchange": info, "python_calls_per_pivot": 0},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                         "frac_of_copy_rate": ach / HBM_COPY_GBPS, "traffic": None, "us_per_pivot": 1e6 * dt / max(npiv, 1),
                         "gpu_ms_rank0": gpu_ms,
                         "note": "per GPU: algorithmic bytes of the whole tableau per pivot / n_gpus / wall time of yalps_shard_run"}}
        if info["kernel"].startswith("dshard_kernel"):
            # delayed row updates: a rank's rows are streamed once per `depth` pivots (dshard_kernel.cuh), so the algorithmic
            # bytes / time is no efficiency; the row traffic actually moved is 1/depth of it
            depth = int(info["kernel"].rsplit("delay_depth:", 1)[1])
            rf = out["roofline"]
            rf["algorithmic_equiv"] = {"achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                                       "frac_of_copy_rate": ach / HBM_COPY_GBPS,
                                       "note": "16*h*w bytes per pivot (SURVEY 8d) / n_gpus / time: what one sweep per pivot would have to move; not an efficiency"}
            moved = ach / depth
            rf["achieved"], rf["frac"], rf["frac_of_copy_rate"], rf["delay_depth"] = moved, moved / HBM_PEAK_GBPS, moved / HBM_COPY_GBPS, depth
            rf["note"] = ("per GPU, delayed row updates: every touched row of the shard is read and written once per %d pivots; achieved = "
                          "algorithmic bytes / n_gpus / depth / wall time of yalps_shard_run (the row traffic actually moved); the per-pivot "
                          "figure is kept as algorithmic_equiv" % depth)
        rf_kernel = info["kernel"]
        out["roofline"]["kernel"] = rf_kernel
        if verdict is not None:
            out["cpu_baseline"], out["parity"] = verdict
        if rehearsal:
            out["rehearsal"] = "ranks share GPU 0 (fewer GPUs than ranks): host transport over gloo, control-flow check only"
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def verify_sharded(args, torch, dist, sharded, local, w, h, M, N, bounds, ident, rank, local_rank, world, rehearsal):
    """The first pivots of the same LP once more, every rank's rows hashed (SHA-256 per 512 rows) and compared on rank 0 with the
    CPU oracle's tableau after the same pivots -- the oracle run is timed and doubles as this workload's cpu_baseline (row loop of
    the elimination over --cpu-threads threads).  Returns (cpu_baseline, parity) on rank 0, None elsewhere; never raises."""
    import hashlib
    try:
        K = args.verify_pivots
        ops = sharded.HipShardOps(local, w, bounds, rank, h, ident, ident.copy(), device=local_rank, private_stream=True)
        comm = sharded.native_comm(ops.ctx, rank, world, transport="host" if rehearsal else "rccl")
        status, result, npiv, _ = ops.run_native(comm, max_pivots=float(K), check_every=16)
        lm, pos, var = ops.download()
        comm.close()
        ops.close()
        lm = lm.reshape(-1, w)
        mine = [(0, 1, hashlib.sha256(lm[0].tobytes()).hexdigest())]
        for lo in range(0, lm.shape[0] - 1, 512):
            blk = np.ascontiguousarray(lm[1 + lo:1 + lo + 512])
            mine.append((bounds[rank] + lo, bounds[rank] + lo + blk.shape[0], hashlib.sha256(blk.tobytes()).hexdigest()))
        del lm
        parts = [mine]
        if dist is not None:
            parts = [None] * world
            dist.all_gather_object(parts, mine)
        if rank != 0:
            return None
        from tests import _oracle  # (the checker: CPU restatement of the reference, pinned by its golden records)
        threads = max(1, min(args.cpu_threads if args.cpu_threads > 0 else 1, os.cpu_count() or 1))
        orc = _oracle.load(omp=threads > 1)
        cores = orc.set_threads(threads) if threads > 1 else 1
        ref = orc.dense_lp(M, N, 42)
        rpos, rvar = ident.copy(), ident.copy()
        t0 = time.perf_counter()
        est, eres, epiv, _ = orc.simplex(ref, w, h, rpos, rvar, max_pivots=float(K))
        dt = time.perf_counter() - t0
        ref = ref.reshape(h, w)
        bad = [(lo, hi) for p in parts for lo, hi, sha in p if hashlib.sha256(np.ascontiguousarray(ref[lo:hi]).tobytes()).hexdigest() != sha]
        ok = not bad and (status, npiv) == (est, epiv) and np.array_equal(pos, rpos) and np.array_equal(var, rvar)
        cpu = {"value": epiv / dt, "unit": "pivots/s", "cores": cores, "kind": "port",
               "sample": "%d pivots of the same dense-LP(%d,%d,seed=42), oracle/simplex_oracle.c%s, %.1f s" % (epiv, M, N, " -fopenmp, %d threads" % cores if threads > 1 else "", dt)}
        parity = {"ok": bool(ok), "pivots": int(npiv), "blocks_checked": sum(len(p) for p in parts), "blocks_wrong": bad[:8], "status": [status, est],
                  "what": "every rank's rows after %d pivots (SHA-256 per 512 rows), both permutations, status and pivot count against the CPU oracle" % K}
        return cpu, parity
    except Exception as e:  # noqa: BLE001 -- a failed check is reported, the measurement above stands
        return ({"value": None, "unit": "pivots/s", "cores": 0, "kind": "port", "sample": "failed"}, {"ok": False, "error": "%s: %s" % (type(e).__name__, e)}) if rank == 0 else None


if __name__ == "__main__":
    main()
