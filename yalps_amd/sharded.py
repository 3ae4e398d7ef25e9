"""Row-sharded simplex of ONE tableau over several GPUs (BASELINE config 5).

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI).  Rows 1..h-1 are split
into contiguous blocks, the objective row is replicated.  Per pivot there is exactly one
collective: an all-gather in which every rank contributes its two candidates (min-ratio row /
most-negative-RHS row of its block) TOGETHER WITH those rows' data (8 + 2*pitch doubles, 262 KB at
w = 16385), so selection and pivot-row broadcast are one exchange and every rank then takes the
same decision from the same bytes.  xGMI is point-to-point; a 2 MB all-gather per pivot is
latency-bound and far below the local sweep (537 MB of traffic per GPU per pivot at C5).

Two drivers:
  * `sharded_simplex_native`: the whole loop inside the library (`yalps_shard_run`): select kernel, ncclAllGather on the
    context's stream, apply kernel, a batch of pivots captured once into a hipGraph and replayed; Python is entered once
    per solve.  The communicator is the library's own (`_native.Comm.rccl`; the 128-byte RCCL id travels through
    torch.distributed or any other channel the host has), or a host callback (`_native.Comm.host`) where RCCL cannot
    run -- ranks that share one GPU in the tests.
  * `sharded_simplex`: the same steps sequenced from Python around a torch.distributed collective (`TorchComm`); the
    `ops` / `comm` split lets the multi-process CPU tests drive it with a numpy stand-in + gloo.
"""
import numpy as np

from . import _native

STATUS = _native.STATUS


def partition(height, nranks):
    """bounds[r]..bounds[r+1] = global rows of rank r; rows 1..height-1, sizes differ by <= 1."""
    rows = height - 1
    base, extra = divmod(rows, nranks)
    bounds = [1]
    for r in range(nranks):
        bounds.append(bounds[-1] + base + (1 if r < extra else 0))
    return bounds


def local_rows(matrix, width, height, bounds, rank):
    """objective row + this rank's block, as a flat row-major local tableau."""
    m = matrix.reshape(height, width)
    return np.ascontiguousarray(np.vstack([m[0:1], m[bounds[rank]:bounds[rank + 1]]])).reshape(-1)


class HipShardOps:
    """The per-rank steps on the GPU, enqueued on torch's current stream (the per-pivot torch collectives of
    sharded_simplex are ordered with them there) or, private_stream=True, on a stream of the context's own: what the
    native loop (run_native) wants -- torch's current stream is the null stream, which cannot be captured into a hipGraph."""

    def __init__(self, local_matrix, width, bounds, rank, global_height, pos, var, device=0, private_stream=False):
        import torch
        self.torch = torch
        nranks = len(bounds) - 1
        local_h = 1 + bounds[rank + 1] - bounds[rank]
        self.ctx = _native.Context(device) if private_stream else _native.Context(device, stream=torch.cuda.current_stream(device).cuda_stream)
        self.tab = _native.DeviceTableau(self.ctx, width, local_h)
        ident = np.arange(width + local_h, dtype=np.int32)
        self.tab.upload(local_matrix, local_h, ident, ident.copy())
        self.tab.set_shard(rank, nranks, bounds, global_height, pos, var)
        self.perm_len = width + global_height
        self.slot = self.tab.shard_slot_doubles()
        self.send = torch.zeros(self.slot, dtype=torch.float64, device=f"cuda:{device}")
        self.recv = torch.zeros(nranks * self.slot, dtype=torch.float64, device=f"cuda:{device}")
        if private_stream:
            torch.cuda.synchronize(device)  # (the buffers were zeroed on torch's stream)

    def begin(self, precision, max_pivots, check_cycles=False):
        self.tab.shard_begin(precision, max_pivots, check_cycles)

    def select(self):
        self.tab.shard_select(self.send.data_ptr())

    def apply(self):
        self.tab.shard_apply(self.recv.data_ptr())

    def poll(self):
        return self.tab.shard_poll()

    def download(self):
        m, pos, var = self.tab.download(perm_len=self.perm_len)
        return m, pos, var

    def run_native(self, comm, precision=1e-8, max_pivots=8192.0, check_every=64, check_cycles=False):
        """yalps_shard_run: returns (status name, result, n_pivots, gpu_ms)."""
        st, result, pivots, ms = self.tab.shard_run(comm, precision, max_pivots, check_every, check_cycles)
        return STATUS[st], result, pivots, ms

    def close(self):
        self.tab.close()
        self.ctx.close()


def native_comm(ctx, rank, world, transport="rccl", group=None):
    """This rank's yalps_comm.  "rccl": ncclCommInitRank inside the library, the unique id broadcast from rank 0 through
    torch.distributed (any backend).  "host": all-gathers staged through host memory and carried by torch.distributed
    (gloo) -- for ranks that share a GPU, where RCCL refuses to run."""
    import torch
    import torch.distributed as dist
    if transport == "rccl":
        box = [_native.Comm.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        return _native.Comm.rccl(ctx, box[0], rank, world)

    def allgather(send):
        if world == 1:
            return send
        out = torch.empty(world * send.size, dtype=torch.float64)
        dist.all_gather_into_tensor(out, torch.from_numpy(send), group=group)
        return out.numpy()
    return _native.Comm.host(ctx, allgather, rank, world)


def sharded_simplex_native(ops, comm, precision=1e-8, max_pivots=8192.0, check_every=64, check_cycles=False):
    """One row-sharded solve with no Python between two pivots; returns (status, result, n_pivots)."""
    status, result, pivots, _ = ops.run_native(comm, precision, max_pivots, check_every, check_cycles)
    return status, result, pivots


class TorchComm:
    """All-gather of the candidate slots through torch.distributed.  With a backend that cannot
    move device tensors (gloo on this build) the slots are staged through host memory."""

    def __init__(self, group=None, stage_on_host=None, always_collective=False):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if stage_on_host is None:
            stage_on_host = dist.is_initialized() and dist.get_backend(group) != "nccl"
        self.stage = stage_on_host
        self.always = always_collective and dist.is_initialized()  # (a group of one still goes through the backend)

    def all_gather(self, recv, send):
        if self.world == 1 and not self.always:
            recv.copy_(send)
        elif self.stage and recv.is_cuda:
            s = send.cpu()
            r = recv.new_empty(recv.shape, device="cpu")
            self.dist.all_gather_into_tensor(r, s, group=self.group)
            recv.copy_(r)
        else:
            self.dist.all_gather_into_tensor(recv, send, group=self.group)


def sharded_simplex(ops, comm, precision=1e-8, max_pivots=8192.0, check_every=32, check_cycles=False):
    """Drives one row-sharded solve; returns (status, result, n_pivots), identical on every rank.
    Mirrors the return protocol of the reference's simplex() (src/simplex.ts:66-142); check_cycles = options.checkCycles
    (:98,137): the permutations and the pivot history are replicated, every rank runs hasCycle on the same pivot."""
    ops.begin(precision, max_pivots, check_cycles)
    while True:
        for _ in range(check_every):
            ops.select()
            comm.all_gather(ops.recv, ops.send)
            ops.apply()
        status, result, pivots = ops.poll()
        if status >= 0:
            return STATUS[status], result, pivots
