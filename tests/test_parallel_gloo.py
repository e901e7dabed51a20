"""CPU, world_size 2 and 8 on gloo: the multi-GPU plumbing of SURVEY.md §8e -- shard the reads, run the
(CPU oracle as stand-in for the per-rank hot path), gather per-read segment rows to rank 0
(config 4) and all-reduce pooled sufficient statistics (config 5); results must equal the serial
run. The DP itself is not under test here (tests/test_gpu_parity.py)."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT
from dynamont_amd import parallel, synth


def test_shard_helpers():
    spans = [parallel.shard_bounds(10, 4, r) for r in range(4)]
    assert spans == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert parallel.shard_bounds(3, 8, 7) == (3, 3)
    parts = parallel.shard_by_cost([5, 1, 9, 3, 7, 2], 2)
    assert sorted(sum(parts, [])) == list(range(6))
    loads = [sum([5, 1, 9, 3, 7, 2][i] for i in p) for p in parts]
    assert abs(loads[0] - loads[1]) <= 3
    dt = parallel.rows_from_bytes(np.zeros(32, dtype=np.uint8))
    assert dt.dtype.itemsize == 16 and len(dt) == 2


def _worker(rank, world, port, model, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from oracle.pyoracle import Oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = parallel.Comm()
        _, mean, sd = synth.read_model_file(model)
        reads = synth.make_reads(8, 7, "rna002", mean, sd, (30, 80))
        lo, hi = parallel.shard_bounds(len(reads), world, rank)
        orc = Oracle(model, 0)
        K = orc.num_kmers
        rows, pooled = [], np.zeros(3 * K)
        for r in reads[lo:hi]:
            a = orc.align(r.signal, r.sequence, True)
            rec = np.zeros(len(a["signal_positions"]), dtype=[("signal_pos", "<u4"), ("sequence_pos", "<u4"), ("probability", "<f8")])
            rec["signal_pos"], rec["sequence_pos"], rec["probability"] = a["signal_positions"], a["sequence_positions"], a["probabilities"]
            rows.append(rec)
            t = orc.train(r.signal, r.sequence, dense=False)
            pooled += np.concatenate([t["weight"], t["sum"], t["sumsq"]])
        row_dt = np.dtype([("signal_pos", "<u4"), ("sequence_pos", "<u4"), ("probability", "<f8")])
        mine = np.concatenate(rows) if rows else np.zeros(0, dtype=row_dt)
        got = comm.gather_rows(torch.from_numpy(np.frombuffer(mine.tobytes(), dtype=np.uint8).copy()), dst=0)
        total = comm.allreduce_sum(pooled)
        np.save(os.path.join(outdir, f"pooled_{rank}.npy"), total)
        # helpers of the multi-rank CLIs: rank 0's string everywhere, OR of a flag, ragged byte gather
        assert parallel.broadcast_str(comm, f"out_{rank}_stamp") == "out_0_stamp"
        assert parallel.broadcast_str(comm, "") == ""
        assert parallel.any_rank(comm, rank == 1) and not parallel.any_rank(comm, False)
        parts = parallel.gather_bytes(comm, b"x" * (3 * rank))
        assert parts == [b"x" * (3 * r) for r in range(world)] if rank == 0 else parts is None
        if rank == 0:
            allrows = np.concatenate([parallel.rows_from_bytes(g.numpy()) for g in got])
            np.save(os.path.join(outdir, "rows.npy"), allrows)
        else:
            assert got is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_gather_and_allreduce(models, tmp_path, oracle_built, world):
    """world 2, and world 8 = the rank count of BASELINE configs 4 and 5 (7 reads over 8 ranks: one rank has
    nothing to contribute and still takes part in every collective)."""
    from oracle.pyoracle import Oracle
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, models["syn5"], str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    # serial reference
    _, mean, sd = synth.read_model_file(models["syn5"])
    reads = synth.make_reads(8, 7, "rna002", mean, sd, (30, 80))
    orc = Oracle(models["syn5"], 0)
    sig, prob, pooled = [], [], np.zeros(3 * orc.num_kmers)
    for r in reads:
        a = orc.align(r.signal, r.sequence, True)
        sig.append(a["signal_positions"])
        prob.append(a["probabilities"])
        t = orc.train(r.signal, r.sequence, dense=False)
        pooled += np.concatenate([t["weight"], t["sum"], t["sumsq"]])
    rows = np.load(tmp_path / "rows.npy")
    assert np.array_equal(rows["signal_pos"], np.concatenate(sig).astype(np.uint32))
    assert np.array_equal(rows["probability"], np.concatenate(prob))
    for r in range(world):
        assert np.allclose(np.load(tmp_path / f"pooled_{r}.npy"), pooled, rtol=1e-12, atol=1e-12)


def _failing_worker(rank, world, port):
    """rounds of the CLIs' collectives; rank 2 fails in round 3 while the others are inside the next gather"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = parallel.Comm()
    with parallel.abort_on_error(comm):
        for rnd in range(1000):
            if rank == 2 and rnd == 3:
                raise RuntimeError("rank 2 lost its GPU")
            parallel.gather_bytes(comm, b"rows" * (rank + 1))
            parallel.any_rank(comm, True)
    dist.destroy_process_group()


def test_one_failing_rank_ends_the_job_within_seconds():
    """the CLIs ran dist.barrier() in a `finally`, so a rank that raised left the others waiting in their
    next collective until the process-group timeout (30 min). Now: every rank of a world-4 job exits non-zero within
    seconds -- the failing one through parallel.abort, the others because their collective fails when its peer is gone."""
    import time
    world = 4
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_failing_worker, args=(r, world, port)) for r in range(world)]
    t0 = time.time()
    for p in procs:
        p.start()
    for p in procs:
        p.join(90)
    took = time.time() - t0
    codes = [p.exitcode for p in procs]
    for p in procs:
        if p.is_alive():
            p.kill()
    assert all(c not in (0, None) for c in codes), codes
    assert took < 60, took

