"""GPU (-m gpu): dyn_comm_* -- the RCCL gather of segment rows (BASELINE config 4) and the RCCL all-reduce of the pooled
statistics (config 5) below Python, from the device buffers of a batch. One GPU is all a gpurun box has, so the
communicator has ONE rank here: ncclCommInitRank, the count all-gather, the root's own rows through the receive
buffer, ncclAllReduce and the D2H copies all run; what a one-rank job cannot exercise is ncclSend/ncclRecv between
peers (the 8-GPU legs of configs 4 and 5 are the driver's to run). The child process imports no torch: librccl is the
system one, bound by the library's own dlopen."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = [pytest.mark.gpu]

CHILD = r'''
import json, sys
import numpy as np
sys.path.insert(0, %(root)r)
from dynamont_amd import Aligner, synth
from dynamont_amd._dynamont import RcclComm
assert "torch" not in sys.modules
model = %(model)r
_, mean, sd = synth.read_model_file(model)
reads = synth.make_reads(4242, 24, "rna004", mean, sd, (60, 400))
al = Aligner(model, "rna004", device=0)
comm = RcclComm(RcclComm.unique_id(), 0, 1, 0)
sig, so, sq, qo = synth.pack_reads(reads)
out = {}
with al.batch_packed(sig, so, sq, qo) as b:
    b.align(True)
    res = b.fetch()
    rows, counts = comm.gather_rows(b, root=0)
    out["counts"] = counts.tolist()
    out["capacity"] = int(b.capacity)
    ok = True
    for i in range(len(reads)):
        a, n = int(res.seg_offsets[i]), int(res.n_segments[i])
        ok &= np.array_equal(rows["signal_pos"][a:a + n], res.signal_positions[a:a + n].astype(np.uint32))
        ok &= np.array_equal(rows["sequence_pos"][a:a + n], res.sequence_positions[a:a + n].astype(np.uint32))
        ok &= np.array_equal(rows["probability"][a:a + n], res.probabilities[a:a + n])
    out["rows_equal_fetch"] = bool(ok)
    # the two-call form below Python: counts first, then a root buffer that is too small -- the exchange completes (nobody
    # would hang), root is told, and the communicator is still good for the next gather
    from dynamont_amd import _native as N
    import ctypes as C
    L = N.lib()
    cnt = np.zeros(1, dtype=np.uint64)
    out["counts_rc"] = int(L.dyn_comm_gather_counts(comm._h, b._h, cnt.ctypes.data_as(N.c_u64_p)))
    out["counts_two_call"] = cnt.tolist()
    small = np.empty(10, dtype=RcclComm.ROW)
    out["too_small_rc"] = int(L.dyn_comm_gather_rows(comm._h, b._h, 0, small.ctypes.data, 10, None))
    out["too_small_msg"] = (L.dyn_comm_last_error(comm._h) or b"").decode()
    rows3, _ = comm.gather_rows(b, root=0)
    out["gather_after_too_small"] = bool(np.array_equal(rows3, rows))
t = al.align_async(sig, so, sq, qo, True)      # a ticket works too (the call waits for it)
rows2, _ = comm.gather_rows(t, root=0)
out["ticket_rows_equal"] = bool(np.array_equal(rows2, rows))
t.close()
with al.batch_packed(sig, so, sq, qo) as b:
    b.train()
    want = b.fetch_train(pooled=True).pooled
    got = comm.allreduce_pooled(b, al.num_kmers)
    out["pooled_close"] = bool(np.allclose(got, want, rtol=1e-12, atol=1e-12))
    out["pooled_weight"] = float(got[:al.num_kmers].sum())
    # the device-resident pooled statistics are summed in a fixed order (pool_stats.hip): the host's sum in read order,
    # bit for bit, and the same bits on every run (a one-rank all-reduce is the identity)
    out["pooled_bits_equal_host"] = bool(np.array_equal(got.view(np.uint64), want.view(np.uint64)))
same = True
for _ in range(10):
    with al.batch_packed(sig, so, sq, qo) as b:
        b.train()
        again = comm.allreduce_pooled(b, al.num_kmers)
        same &= bool(np.array_equal(again.view(np.uint64), got.view(np.uint64)))
out["pooled_identical_over_10_runs"] = same
# a batch in which one read fails on the host (an N) and one on the device (a NaN sample): both stay out of the sums
seqs2 = [r.sequence for r in reads]
seqs2[3] = seqs2[3][:20] + "N" + seqs2[3][21:]
sigs2 = [r.signal.copy() for r in reads]
sigs2[7][50] = np.nan
with al.batch(sigs2, seqs2) as b:
    b.train()
    r2 = b.fetch_train(pooled=True)
    got2 = comm.allreduce_pooled(b, al.num_kmers)
    out["failed_reads"] = [int(x) for x in np.flatnonzero(r2.status != 0)]
    out["pooled_bits_equal_host_with_failed_reads"] = bool(np.array_equal(got2.view(np.uint64), r2.pooled.view(np.uint64)))
comm.close()
al.close()
print("RESULT " + json.dumps(out))
'''


def test_one_rank_rccl_gather_and_allreduce(models):
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "model": models["syn9"]}], capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert r.returncode == 0 and line, r.stderr[-3000:]
    out = json.loads(line[-1][7:])
    assert out["counts"] == [out["capacity"]] and out["capacity"] > 1000
    assert out["rows_equal_fetch"] and out["ticket_rows_equal"] and out["pooled_close"]
    assert out["counts_rc"] == 0 and out["counts_two_call"] == out["counts"]
    assert out["too_small_rc"] == 1 and "rows_cap" in out["too_small_msg"] and out["gather_after_too_small"]
    assert out["pooled_weight"] > 100.0
    assert out["pooled_bits_equal_host"] and out["pooled_identical_over_10_runs"]
    assert out["failed_reads"] == [3, 7] and out["pooled_bits_equal_host_with_failed_reads"]


def test_bench_exchange_through_dyn_comm_beside_the_resident_queue():
    """`bench.py` with its exchange forced on for ONE rank (DYN_BENCH_FORCE_DIST=1): the timed steps of BASELINE's cfg2 run in
    the resident read queue with 8 compute units reserved (dyn_aligner_set_session_mode) while every step's segment rows go
    through dyn_comm_gather_counts / dyn_comm_gather_rows -- the code path of `bench.py --gpus N`, on the one rank a 1-GPU box
    can host. The line must name the exchange implementation and count the rows it gathered."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    import dynamont_amd
    # the pools this process' earlier tests parked would leave the child too little memory for an arena per resident wave
    # (it would then run one launch per batch: session_plan's page-starved exit)
    dynamont_amd.release_cached_memory()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(DYN_BENCH_FORCE_DIST="1", PYTHONPATH=ROOT)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--batches", "2", "--no-plain", "--no-cpu-baseline",
           "--no-e2e", "--no-resident"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["rccl_ranks"] == 1 and d["value"] > 0
    ex = d["exchange"]
    assert ex["implementation"].startswith("dyn_comm_") and ex["resident_queue"] and ex["reserved_cus"] == 8
    assert ex["rows_gathered_rank0"] > 4 * 1024 * 1500   # ~2 000 segments per read, 1 024 reads, 4 timed steps (+ the warm-up's)
    rq = d["roofline"]["resident_queue"]
    assert rq and rq["aborted"] == 0 and rq["waves"] == (256 - 8) * 4 * rq["sessions"]   # 8 CUs left to RCCL
