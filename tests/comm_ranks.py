"""One rank of a multi-rank dyn_comm_* job on ONE device (a child process of tests/test_gpu_comm_ranks.py; also runnable by
hand: `python tests/comm_ranks.py launch gather 2 /tmp/w <model>`).

A gpurun box has one GPU and RCCL refuses two ranks of one communicator on the same device ("Duplicate GPU detected":
same host hash, same bus id). Every rank therefore gets its own NCCL_HOSTID: the ranks then look like one-GPU NODES to
RCCL, which connects them through its socket transport over the loopback interface. Everything above the transport is
the real thing: ncclCommInitRank with n_ranks > 1, the 8-byte all-gather of counts over N entries, the grouped
ncclSend / ncclRecv of the rows (root's receive loop over its peers, the non-root send), ncclAllReduce, ncclCommAbort.
What this does not exercise is the xGMI transport itself.

The children import no torch: librccl is the system one, bound by the library's own dlopen."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rank_env(rank, extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(NCCL_HOSTID="dyn-one-device-rank-%d" % rank, NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", NCCL_NET="Socket",
               HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT, NCCL_DEBUG=os.environ.get("NCCL_DEBUG", "WARN"))
    env.update(extra or {})
    return env


def launch(scenario, n, workdir, model, timeout=420, extra=None):
    """start the n ranks (fresh processes), wait for them, return [(returncode, RESULT dict or None, stderr tail)]"""
    os.makedirs(workdir, exist_ok=True)
    for f in os.listdir(workdir):
        os.remove(os.path.join(workdir, f))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "rank", scenario, str(r), str(n), workdir, model],
                              env=rank_env(r, extra), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(n)]
    deadline = time.time() + timeout
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
            o, e = p.communicate()
            e += "\n[comm_ranks] killed at the test's timeout"
        line = [l for l in o.splitlines() if l.startswith("RESULT ")]
        outs.append((p.returncode, json.loads(line[-1][7:]) if line else None, e[-3000:]))
    return outs


_LOOPBACK = None


def rccl_loopback_ok(workdir):
    """(cached) True when two ranks on this box's one device get a communicator and an all-reduce through RCCL's socket
    transport over loopback; else the reason. Boxes of the pool do; a box without a usable `lo` interface would not."""
    global _LOOPBACK
    if _LOOPBACK is None:
        try:
            outs = launch("probe", 2, workdir, "-", timeout=120, extra={"DYN_COMM_TIMEOUT_S": "60"})
            ok = all(rc == 0 and res is not None and res["sum"] == 3.0 for rc, res, _ in outs)
            _LOOPBACK = True if ok else "RCCL over loopback sockets unavailable on this box: " + " | ".join(e[-300:] for _, _, e in outs)
        except Exception as e:   # noqa: BLE001
            _LOOPBACK = "RCCL loopback probe failed: %s" % e
    return _LOOPBACK


def exchange_id(workdir, rank):
    from dynamont_amd._dynamont import RcclComm
    path = os.path.join(workdir, "id.bin")
    if rank == 0:
        uid = RcclComm.unique_id()
        with open(path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(path + ".tmp", path)
        return uid
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > 120:
            raise RuntimeError("rank 0 never wrote the communicator id")
        time.sleep(0.02)
    return open(path, "rb").read()


def rows_of(res, total):
    """what dyn_comm_gather_rows must deliver for this batch: the device rows in read order, as fetch returns them"""
    import numpy as np
    from dynamont_amd._dynamont import RcclComm
    out = np.zeros(total, dtype=RcclComm.ROW)
    out["signal_pos"] = res.signal_positions[:total].astype(np.uint32)
    out["sequence_pos"] = res.sequence_positions[:total].astype(np.uint32)
    out["probability"] = res.probabilities[:total]
    return out


def run_rank(scenario, rank, n, workdir, model):
    import ctypes as C

    import numpy as np
    sys.path.insert(0, ROOT)
    from dynamont_amd import Aligner, synth
    from dynamont_amd import _native as N
    from dynamont_amd._dynamont import RcclComm
    assert "torch" not in sys.modules
    L = N.lib()
    if scenario == "probe":   # can RCCL connect two ranks on this box at all? (communicator + one all-reduce, no aligner)
        comm = RcclComm(exchange_id(workdir, rank), rank, n, 0)
        got = comm.allreduce(np.array([1.0 + rank]), "sum")
        comm.close()
        print("RESULT " + json.dumps({"rank": rank, "sum": float(got[0])}), flush=True)
        return
    _, mean, sd = synth.read_model_file(model)
    reads = synth.make_reads(7000 + rank, 12 + 5 * rank, "rna004", mean, sd, (60, 400))
    al = Aligner(model, "rna004", device=0)
    comm = RcclComm(exchange_id(workdir, rank), rank, n, 0)
    sig, so, sq, qo = synth.pack_reads(reads)
    out = {"rank": rank, "n": n}

    def save(name, **arrays):
        np.savez(os.path.join(workdir, "%s_rank%d.npz" % (name, rank)), **arrays)

    # A: a synchronous batch, rows to root 0
    with al.batch_packed(sig, so, sq, qo) as b:
        b.align(True)
        res = b.fetch()
        mine = rows_of(res, int(b.capacity))
        out["capacity"] = int(b.capacity)
        rows, counts = comm.gather_rows(b, root=0)
        save("A", mine=mine, counts=counts, **({"rows": rows} if rank == 0 else {}))
        out["A_counts"] = counts.tolist()
        if scenario in ("die", "stall"):
            # the peers are connected now. Counts once more -- then rank 1 is gone before the rows
            cnt = np.zeros(n, dtype=np.uint64)
            out["die_counts_rc"] = int(L.dyn_comm_gather_counts(comm._h, b._h, cnt.ctypes.data_as(N.c_u64_p)))
            if rank == 1:
                print("RESULT " + json.dumps(out), flush=True)
                if scenario == "stall":
                    time.sleep(9.0)    # alive, its sockets open, but never in the exchange: only the root's clock can tell
                os._exit(0)   # no destructors, no ncclCommDestroy: a crashed rank
            t0 = time.time()
            buf = np.empty(int(cnt.sum()) + 1, dtype=RcclComm.ROW)
            rc = int(L.dyn_comm_gather_rows(comm._h, b._h, 0, buf.ctypes.data if rank == 0 else None, buf.size if rank == 0 else 0, None))
            out["die_rows_rc"] = rc
            out["die_rows_s"] = time.time() - t0
            out["die_msg"] = (L.dyn_comm_last_error(comm._h) or b"").decode()
            # an aborted communicator answers every later call at once
            t0 = time.time()
            out["die_after_rc"] = int(L.dyn_comm_gather_counts(comm._h, b._h, cnt.ctypes.data_as(N.c_u64_p)))
            out["die_after_s"] = time.time() - t0
            out["die_after_msg"] = (L.dyn_comm_last_error(comm._h) or b"").decode()
            print("RESULT " + json.dumps(out), flush=True)
            os._exit(0)       # (the communicator of a job that lost a rank is not destroyed collectively)

    # B: an asynchronous ticket, rows to the LAST rank
    t = al.align_async(sig, so, sq, qo, True)
    rows, counts = comm.gather_rows(t, root=n - 1)
    save("B", counts=counts, **({"rows": rows} if rank == n - 1 else {}))
    t.close()

    # C: rank 1's batch has not been aligned: it announces 0 rows, takes part in both collectives, THEN reports its error
    with al.batch_packed(sig, so, sq, qo) as b:
        if rank != 1:
            b.align(True)
        try:
            rows, counts = comm.gather_rows(b, root=0)
            out["C_error"] = None
            save("C", counts=counts, **({"rows": rows} if rank == 0 else {}))
        except Exception as e:   # noqa: BLE001
            out["C_error"] = str(e)

    # D: the two-call form with a root buffer that is too small: the exchange completes on every rank, root is told, the
    # communicator is good for the next gather
    with al.batch_packed(sig, so, sq, qo) as b:
        b.align(True)
        cnt = np.zeros(n, dtype=np.uint64)
        out["D_counts_rc"] = int(L.dyn_comm_gather_counts(comm._h, b._h, cnt.ctypes.data_as(N.c_u64_p)))
        small = np.empty(10, dtype=RcclComm.ROW)
        out["D_rows_rc"] = int(L.dyn_comm_gather_rows(comm._h, b._h, 0, small.ctypes.data if rank == 0 else None, 10 if rank == 0 else 0, None))
        out["D_msg"] = (L.dyn_comm_last_error(comm._h) or b"").decode() if rank == 0 else ""
        rows, counts = comm.gather_rows(b, root=0)
        save("D", counts=counts, **({"rows": rows} if rank == 0 else {}))

    # E: config 5's exchange: the pooled statistics of this rank's training batch, summed over the ranks in place
    with al.batch_packed(sig, so, sq, qo) as b:
        b.train()
        own = b.fetch_train(pooled=True).pooled
        summed = comm.allreduce_pooled(b, al.num_kmers)
        save("E", own=own, summed=summed)
    # F: the CLIs' payloads (host memory) through the same exchange: a ragged byte gather to rank 0 and to the last rank (rank 1
    # sends nothing), sum and max all-reduce of a small float64 vector
    payload = (b"rank%d|" % rank) * (1 + 50000 * rank) if rank != 1 else b""
    for root in (0, n - 1):
        parts = comm.gather_bytes(payload, root=root)
        if rank == root:
            out["F_ok_root%d" % root] = bool(parts == [((b"rank%d|" % r) * (1 + 50000 * r) if r != 1 else b"") for r in range(n)])
        else:
            out["F_none_root%d" % root] = parts is None
    v = np.array([1.0 + rank, -float(rank), 0.5])
    out["F_sum"] = comm.allreduce(v, "sum").tolist()
    out["F_max"] = comm.allreduce(v, "max").tolist()
    comm.close()
    al.close()
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "rank":
        run_rank(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6])
    else:   # launch <scenario> <n> <workdir> <model>
        sys.path.insert(0, ROOT)
        for rc, res, err in launch(sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5], extra={"DYN_COMM_TIMEOUT_S": "8"} if sys.argv[2] == "die" else None):
            print(rc, json.dumps(res))
            if rc != 0 or res is None:
                print(err)
