"""Multi-GPU plumbing: reads are independent, so a job shards them across one
process per GPU, started by ``torch.distributed.run``. There is no data-path collective. The only exchanges are

  * gather_bytes     a rank's rows (its part of the compressed output frame, its ``.errors`` lines) -> rank 0
                     (BASELINE.json config 4),
  * allreduce_sum    pooled sufficient statistics (w, s1, s2)[numKmers] and the per-batch sums (config 5).

ONE exchange stack: on GPUs the payloads travel through the library's own RCCL path, ``dyn_comm_*``
(dynamont_amd/csrc/rccl_comm.cpp: count all-gather + grouped ncclSend / ncclRecv, ncclAllReduce) -- what
``bench.py --gpus N`` times and ``tests/test_gpu_comm_ranks.py`` exercises at 2 and 4 ranks. ``torch.distributed``
only starts the ranks, hands the 128-byte communicator id round, broadcasts a few control strings and holds the final
barrier. CPU rehearsals (backend "gloo", no GPU: ``tests/test_parallel_gloo.py``) carry the payloads over torch instead
(``DYN_DIST_EXCHANGE=torch``); ``Comm.implementation`` says which one a job used.

This module never touches the DP itself; payloads are plain byte / float64 buffers.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n_items: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous, balanced [lo, hi) of ``n_items`` for ``rank`` (first n%world ranks get one more)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_by_cost(costs, world: int) -> list[list[int]]:
    """Greedy longest-first assignment of items to ranks by cost (signal length): keeps the
    per-GPU lattice work even when read lengths differ (config 3)."""
    order = np.argsort(-np.asarray(costs, dtype=np.float64), kind="stable")
    load = np.zeros(world)
    out = [[] for _ in range(world)]
    for i in order:
        r = int(np.argmin(load))
        out[r].append(int(i))
        load[r] += float(costs[i])
    return out


class Comm:
    """Thin wrapper so callers do not depend on torch when running single-process."""

    def __init__(self, device=None, xchg=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.active = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank() if self.active else 0
        self.world = dist.get_world_size() if self.active else 1
        self.device = device if device is not None else "cpu"
        # dynamont_amd._dynamont.RcclComm over all ranks (dyn_comm_*): carries every payload when present
        self.xchg = xchg

    @property
    def implementation(self) -> str:
        if self.xchg is not None:
            return "dyn_comm_* (dynamont_amd/csrc/rccl_comm.cpp): dyn_comm_gather_bytes / dyn_comm_allreduce_f64 over RCCL"
        return "torch.distributed (%s): CPU rehearsal of the exchange" % (self.dist.get_backend() if self.active else "single process")

    def close(self):
        if self.xchg is not None:
            self.xchg.close()
            self.xchg = None

    def allreduce_sum(self, x: np.ndarray) -> np.ndarray:
        """Sum of a float64 vector over ranks (sufficient statistics are linear-domain sums, so a
        plain sum all-reduce is exact up to fp64 association)."""
        if not self.active:
            return x
        if self.xchg is not None:
            return self.xchg.allreduce(x, "sum")
        t = self.torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def allreduce_sum_tensor(self, t):
        if self.active:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t

    def gather_rows(self, rows_u8, dst: int = 0):
        """Variable-length gather of a uint8 tensor to ``dst``: sizes are exchanged first
        (all_gather), payloads are padded to the maximum for the gather. Returns the list of
        per-rank tensors on ``dst`` and None elsewhere."""
        torch, dist = self.torch, self.dist
        if not self.active:
            return [rows_u8]
        n = torch.tensor([rows_u8.numel()], dtype=torch.int64, device=rows_u8.device)
        sizes = [torch.zeros_like(n) for _ in range(self.world)]
        dist.all_gather(sizes, n)
        sizes = [int(s.item()) for s in sizes]
        cap = max(sizes)
        pad = rows_u8 if rows_u8.numel() == cap else torch.cat(
            [rows_u8, torch.zeros(cap - rows_u8.numel(), dtype=torch.uint8, device=rows_u8.device)])
        bufs = [torch.empty(cap, dtype=torch.uint8, device=rows_u8.device) for _ in range(self.world)] if self.rank == dst else None
        dist.gather(pad, bufs, dst=dst)
        if self.rank != dst:
            return None
        return [b[:s] for b, s in zip(bufs, sizes)]


def rows_from_bytes(buf: np.ndarray) -> np.ndarray:
    """View gathered bytes as the C ABI's dyn_segment_row records."""
    dt = np.dtype([("signal_pos", "<u4"), ("sequence_pos", "<u4"), ("probability", "<f8")])
    return np.frombuffer(np.ascontiguousarray(buf).tobytes(), dtype=dt)


def init_from_env():
    """When launched by ``torch.distributed.run`` (WORLD_SIZE > 1) join the process group and return
    (Comm, local_rank); otherwise (None, 0). Backend: ``DYN_DIST_BACKEND`` (default "nccl" = RCCL;
    "gloo" for CPU rehearsals). ``DYN_DIST_ONE_DEVICE=1`` maps every rank to device 0 (rehearsal)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return None, int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    backend = os.environ.get("DYN_DIST_BACKEND", "nccl")
    one_device = bool(os.environ.get("DYN_DIST_ONE_DEVICE"))
    local_rank = 0 if one_device else int(os.environ.get("LOCAL_RANK", "0"))
    exchange = os.environ.get("DYN_DIST_EXCHANGE", "dyn_comm" if backend == "nccl" else "torch")
    if one_device and exchange == "dyn_comm":
        one_device_rccl_env(int(os.environ.get("RANK", "0")))
    if not dist.is_initialized():
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    comm = Comm(device=f"cuda:{local_rank}" if backend == "nccl" else "cpu")
    if exchange == "dyn_comm":
        from dynamont_amd._dynamont import RcclComm
        uid = broadcast_bytes(comm, RcclComm.unique_id() if comm.rank == 0 else b"", 128)
        comm.xchg = RcclComm(uid, comm.rank, comm.world, local_rank)
    return comm, local_rank


def one_device_rccl_env(rank: int) -> None:
    """Rehearsals with every rank on ONE device (a 1-GPU box): RCCL refuses two ranks of a communicator on the same device
    ("Duplicate GPU detected": same host hash, same bus id), so every rank gets its own NCCL_HOSTID -- the ranks then look
    like one-GPU nodes and RCCL connects them by its socket transport over loopback. Everything above the transport is the
    real thing. Must run before the process' first RCCL call."""
    import os
    os.environ.setdefault("NCCL_HOSTID", "dyn-one-device-rank-%d" % rank)
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.environ.setdefault("NCCL_IB_DISABLE", "1")
    os.environ.setdefault("NCCL_NET", "Socket")


def broadcast_bytes(comm: "Comm", data: bytes, n: int, src: int = 0) -> bytes:
    """``n`` bytes of rank ``src`` on every rank (control plane: torch.distributed) -- the communicator id."""
    import torch
    buf = torch.zeros(n, dtype=torch.uint8, device=comm.device)
    if comm.rank == src:
        buf.copy_(torch.frombuffer(bytearray(data), dtype=torch.uint8))
    comm.dist.broadcast(buf, src=src)
    return bytes(buf.cpu().numpy().tobytes())


def gather_bytes(comm: "Comm", payload: bytes, dst: int = 0):
    """Variable-length gather of one bytes object per rank (CSV rows, error lines) to ``dst``."""
    if comm.xchg is not None:
        return comm.xchg.gather_bytes(payload, root=dst)
    import torch
    t = torch.frombuffer(bytearray(payload), dtype=torch.uint8) if payload else torch.zeros(0, dtype=torch.uint8)
    parts = comm.gather_rows(t.to(comm.device), dst=dst)
    if parts is None:
        return None
    return [bytes(p.cpu().numpy().tobytes()) for p in parts]


def any_rank(comm: "Comm", flag: bool) -> bool:
    """Logical OR of a flag over ranks."""
    if comm.xchg is not None:
        return bool(comm.xchg.allreduce(np.array([1.0 if flag else 0.0]), "max")[0] > 0.5)
    import torch
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=comm.device)
    comm.dist.all_reduce(t, op=comm.dist.ReduceOp.MAX)
    return bool(t.item())


def broadcast_str(comm: "Comm", text: str, src: int = 0) -> str:
    """Rank ``src``'s string on every rank (e.g. a time-stamped output directory name)."""
    import torch
    data = text.encode() if comm.rank == src else b""
    n = torch.tensor([len(data)], dtype=torch.int64, device=comm.device)
    comm.dist.broadcast(n, src=src)
    buf = torch.zeros(int(n.item()), dtype=torch.uint8, device=comm.device)
    if comm.rank == src and data:
        buf.copy_(torch.frombuffer(bytearray(data), dtype=torch.uint8))
    comm.dist.broadcast(buf, src=src)
    return bytes(buf.cpu().numpy().tobytes()).decode()


_SCRATCH: list[str] = []


def register_scratch(*paths: str) -> None:
    """Files of this rank that must not outlive the job (a rank's part of the output frame in local scratch space can
    be gigabytes): removed by ``abort`` -- which leaves through ``os._exit``, past every ``finally`` and ``atexit`` --
    and by ``remove_scratch`` on the ordinary way out."""
    _SCRATCH.extend(p for p in paths if p)


def remove_scratch() -> None:
    import os
    while _SCRATCH:
        try:
            os.remove(_SCRATCH.pop())
        except OSError:
            pass


def abort(comm: "Comm | None", exc: BaseException, grace_s: float = 5.0) -> None:
    """A rank that fails must not leave the others waiting in a collective until the process-group timeout: print
    the error, tear the process group down (closing its connections makes peers blocked in a gloo collective fail
    at once; under ``torch.distributed.run`` the agent kills the remaining ranks as soon as this one has exited
    non-zero) and leave with exit code 1. Never re-executes anything. A watchdog bounds the teardown itself."""
    import os
    import sys
    import threading
    import traceback
    traceback.print_exception(type(exc), exc, exc.__traceback__)
    print(f"rank {comm.rank if comm else 0}: aborting the job", file=sys.stderr, flush=True)
    threading.Timer(grace_s, lambda: os._exit(1)).start()
    remove_scratch()
    # (the dyn_comm communicator is NOT destroyed here -- ncclCommDestroy is collective; this process leaves through
    # os._exit, its peers' exchanges then fail by RCCL's asynchronous error or at DYN_COMM_TIMEOUT_S, and under
    # torch.distributed.run the agent kills them as soon as this rank has exited non-zero)
    try:
        if comm is not None and comm.dist.is_initialized():
            comm.dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass
    os._exit(1)


class abort_on_error:
    """``with abort_on_error(comm): ...`` -- multi-rank jobs: an exception inside ends the whole job (see abort);
    single-process runs (comm None): the exception propagates as usual."""

    def __init__(self, comm):
        self.comm = comm

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if ev is not None and self.comm is not None and not isinstance(ev, SystemExit):
            abort(self.comm, ev)
        return False

