"""GPU (-m gpu): dyn_comm_* with MORE THAN ONE RANK -- the code BASELINE's configs 4 and 5 add to the single-GPU path
(rccl_comm.cpp: the N-entry all-gather of counts, root's ncclRecv loop over its peers, the non-root ncclSend, the
all-reduce across ranks, the abort path) -- executed on the one device a gpurun box has, through REAL RCCL: every rank is
a fresh process with its own NCCL_HOSTID, so RCCL takes the ranks for one-GPU nodes and connects them by its socket
transport over loopback (tests/comm_ranks.py). 2 and 4 ranks (a box admits 6 processes on its card). Results must equal
the single-process rows / the host's sum bit for bit."""
import os

import numpy as np
import pytest

import comm_ranks

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("native_lib")]

DYN_ERR_INVALID_ARGUMENT, DYN_ERR_DEVICE = 1, 3


def _load(workdir, name, rank):
    return np.load(os.path.join(workdir, "%s_rank%d.npz" % (name, rank)))


@pytest.fixture(scope="module", autouse=True)
def _rccl_over_loopback(tmp_path_factory):
    ok = comm_ranks.rccl_loopback_ok(str(tmp_path_factory.mktemp("rccl_probe")))
    if ok is not True:
        pytest.skip(ok)


def _release_parked():
    import dynamont_amd
    dynamont_amd.release_cached_memory()   # pools parked by earlier tests of this process: leave the children the memory


@pytest.mark.parametrize("n", [2, 4])
def test_gather_and_allreduce_across_ranks(models, tmp_path, n):
    _release_parked()
    w = str(tmp_path / "w")
    outs = comm_ranks.launch("gather", n, w, models["syn9"])
    for rc, res, err in outs:
        assert rc == 0 and res is not None, err
    res = [o[1] for o in outs]
    mine = [_load(w, "A", r)["mine"] for r in range(n)]
    sizes = [len(m) for m in mine]
    assert all(s > 1000 for s in sizes) and len(set(sizes)) == n   # every rank brings another number of rows
    everything = np.concatenate(mine)
    # A: counts on every rank, rows on root 0 in rank order
    for r in range(n):
        assert _load(w, "A", r)["counts"].tolist() == sizes and res[r]["A_counts"] == sizes
    assert np.array_equal(_load(w, "A", 0)["rows"], everything)
    # B: asynchronous tickets, root = the last rank
    assert np.array_equal(_load(w, "B", n - 1)["rows"], everything)
    assert _load(w, "B", 0)["counts"].tolist() == sizes
    # C: rank 1 announced 0 rows, nobody hung, root has the others' rows, rank 1 (alone) got its error afterwards
    assert "has not been aligned" in res[1]["C_error"]
    assert all(res[r]["C_error"] is None for r in range(n) if r != 1)
    c0 = _load(w, "C", 0)
    assert c0["counts"].tolist() == [0 if r == 1 else sizes[r] for r in range(n)]
    assert np.array_equal(c0["rows"], np.concatenate([mine[r] for r in range(n) if r != 1]))
    # D: a root buffer that is too small: every rank's exchange completed, root was told, the next gather works
    assert all(res[r]["D_counts_rc"] == 0 for r in range(n))
    assert res[0]["D_rows_rc"] == DYN_ERR_INVALID_ARGUMENT and "rows_cap" in res[0]["D_msg"]
    assert all(res[r]["D_rows_rc"] == 0 for r in range(1, n))
    assert np.array_equal(_load(w, "D", 0)["rows"], everything)
    # E: the pooled statistics: every rank holds the same bits, and they are the sum of the ranks' own statistics
    own = [_load(w, "E", r)["own"] for r in range(n)]
    summed = [_load(w, "E", r)["summed"] for r in range(n)]
    for r in range(1, n):
        assert np.array_equal(summed[r].view(np.uint64), summed[0].view(np.uint64))
    host = own[0].copy()
    for r in range(1, n):
        host += own[r]
    assert host[: len(host) // 3].sum() > 1000.0   # total weight = the samples of all ranks' reads
    # F: the CLIs' host payloads through the same code path (dyn_comm_gather_bytes / dyn_comm_allreduce_f64)
    assert res[0]["F_ok_root0"] and res[n - 1]["F_ok_root%d" % (n - 1)]
    assert all(res[r].get("F_none_root0", True) and res[r].get("F_none_root%d" % (n - 1), True) for r in range(n))
    for r in range(n):
        assert res[r]["F_sum"] == [n * (n + 1) / 2, -n * (n - 1) / 2, 0.5 * n] and res[r]["F_max"] == [float(n), 0.0, 0.5]
    if n == 2:
        assert np.array_equal(summed[0].view(np.uint64), host.view(np.uint64))   # two addends: no association to differ in
    else:
        assert np.allclose(summed[0], host, rtol=1e-13, atol=1e-13)   # RCCL's order of adding four addends is its own


@pytest.mark.parametrize("n", [2, 4])
def test_a_rank_that_dies_between_counts_and_rows_does_not_hang_the_root(models, tmp_path, n):
    """Rank 1 leaves (os._exit: no ncclCommDestroy) after dyn_comm_gather_counts. Root's receive from it can never
    complete: dyn_comm_gather_rows must come back -- an asynchronous RCCL error or DYN_COMM_TIMEOUT_S, then ncclCommAbort --
    with an error, and every later call on the handle must fail at once."""
    _release_parked()
    w = str(tmp_path / "w")
    outs = comm_ranks.launch("die", n, w, models["syn9"], timeout=240, extra={"DYN_COMM_TIMEOUT_S": "8"})
    for rc, res, err in outs:
        assert res is not None, err       # nobody was killed at the test's timeout
    root = outs[0][1]
    assert root["die_counts_rc"] == 0
    assert root["die_rows_rc"] == DYN_ERR_DEVICE and "aborted" in root["die_msg"], root
    assert root["die_rows_s"] < 60.0
    assert root["die_after_rc"] == DYN_ERR_DEVICE and root["die_after_s"] < 1.0 and "aborted" in root["die_after_msg"]


def test_a_rank_that_never_arrives_costs_the_root_its_timeout_not_a_hang(models, tmp_path):
    """Rank 1 stays ALIVE (connections open: RCCL has no asynchronous error to report -- the situation of a peer over xGMI,
    where no socket closes) but does not join the gather of rows for 9 s. Root gives up at DYN_COMM_TIMEOUT_S = 3 s."""
    _release_parked()
    outs = comm_ranks.launch("stall", 2, str(tmp_path / "w"), models["syn9"], timeout=240, extra={"DYN_COMM_TIMEOUT_S": "3"})
    for rc, res, err in outs:
        assert res is not None, err
    root = outs[0][1]
    assert root["die_rows_rc"] == DYN_ERR_DEVICE and "did not complete within" in root["die_msg"] and "aborted" in root["die_msg"], root
    assert 2.5 < root["die_rows_s"] < 8.5
    assert root["die_after_rc"] == DYN_ERR_DEVICE and root["die_after_s"] < 1.0
