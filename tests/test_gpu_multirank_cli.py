"""GPU (-m gpu): the CLIs under torch.distributed.run with two and four ranks, all on cuda:0. The payloads travel through
the product's ONE exchange stack, dyn_comm_* over real RCCL (every rank gets its own NCCL_HOSTID, so RCCL takes the ranks
for one-GPU nodes and connects them by sockets over loopback: dynamont_amd/parallel.py one_device_rccl_env); gloo only
carries torch.distributed's control plane (the communicator id, control strings, the final barrier). Reads are sharded,
rank 0 gathers the ranks' parts of the output frame (BASELINE config 4) / pooled statistics are all-reduced (config 5).
Results must equal the single-process run up to row order."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, model_for
from dynamont_amd import synth, zstd_io
from dynamont_amd.segmentation import segment as seg
from dynamont_amd.segmentation import train as trn
from dynamont_amd.segmentation import utils as U

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("native_lib")]


def _exchange(tmp_path_factory=None):
    """dyn_comm over real RCCL where two ranks can be connected on this box (every box of the pool), else torch / gloo"""
    import tempfile

    import comm_ranks
    return "dyn_comm" if comm_ranks.rccl_loopback_ok(tempfile.mkdtemp(prefix="rccl_probe_")) is True else "torch"


def _torchrun(module, args, port, ranks=2):
    exchange = _exchange()
    env = dict(os.environ, DYN_DIST_BACKEND="gloo", DYN_DIST_ONE_DEVICE="1", DYN_DIST_EXCHANGE=exchange, PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr",
           "127.0.0.1", "--master-port", str(port), "-m", module] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    # the same exchange implementation `bench.py --gpus N` names (test_bench_gpus_n_launches_its_own_ranks below)
    assert ("exchange: dyn_comm_* (dynamont_amd/csrc/rccl_comm.cpp)" if exchange == "dyn_comm" else "exchange: torch.distributed (gloo)") in r.stderr, r.stderr[-3000:]


@pytest.mark.parametrize("ranks", [2, 4])   # (a GPU box lets one job put 6 processes on its card: pytest itself + 4 ranks stay inside)
def test_resquiggle_ranks_gather_rows(models, tmp_path, ranks):
    pore = "rna004"
    model = model_for(models, pore)
    _, mean, sd = synth.read_model_file(model)
    reads = synth.make_reads(91, 11, pore, mean, sd, (60, 200))
    raw, bam, _ = synth.write_dataset(str(tmp_path / "in"), "ds", reads, pore, seed=2)
    lines = open(bam).read().splitlines()
    f = lines[5].split("\t"); f[1] = f[1][:30] + "N" + f[1][31:]; lines[5] = "\t".join(f)
    open(bam, "w").write("\n".join(lines) + "\n")
    base = ["-r", str(tmp_path / "in"), "-b", bam, "--mode", "basic", "-p", pore, "--model_path", model, "--batch-reads", "3"]
    seg.main(base + ["-o", str(tmp_path / "single.csv")])
    _torchrun("dynamont_amd.segmentation.segment", base + ["-o", str(tmp_path / "multi.csv")], 29621 + ranks, ranks)
    one = zstd_io.decompress(open(tmp_path / "single.csv.zst", "rb").read()).decode().splitlines()
    two = zstd_io.decompress(open(tmp_path / "multi.csv.zst", "rb").read()).decode().splitlines()
    assert one[0] == two[0] and sorted(one[1:]) == sorted(two[1:]) and len(one) > 500
    # every rank compressed its own rows into a part of the frame; rank 0 appended the parts and closed it: ONE frame
    assert zstd_io.count_frames(open(tmp_path / "multi.csv.zst", "rb").read()) == 1
    assert not [f for f in os.listdir(tmp_path) if "part" in f]
    assert sorted(open(tmp_path / "single.errors").read().splitlines()) == sorted(open(tmp_path / "multi.errors").read().splitlines())


def test_train_two_ranks_allreduce(models, tmp_path):
    pore = "rna002"
    kmers, mean, sd = synth.read_model_file(model_for(models, pore))
    mean[kmers.index("AAAAA")] = 1.2
    model = str(tmp_path / "polyA_ok.model")
    U.write_kmer_model(model, {k_: (float(m), float(s)) for k_, m, s in zip(kmers, mean, sd)})
    reads = synth.make_reads(92, 8, pore, mean, sd, (80, 200))
    raw, bam, _ = synth.write_dataset(str(tmp_path / "in"), "tr", reads, pore, seed=5)
    lines = open(bam).read().splitlines()
    for i in range(1, len(lines)):
        f = lines[i].split("\t"); f[2] = "15.0"; lines[i] = "\t".join(f)
    open(bam, "w").write("\n".join(lines) + "\n")
    common = ["-r", str(tmp_path / "in"), "-b", bam, "-p", pore, "--model_path", model, "--aggregate", "pooled", "--no-timestamp"]
    # one process, batches of 8  ==  two ranks, batches of 4 reads per rank
    trn.main(common + ["-o", str(tmp_path / "single"), "--batch_size", "8", "--max_batches", "1"])
    _torchrun("dynamont_amd.segmentation.train", common + ["-o", str(tmp_path / "multi"), "--batch_size", "4", "--max_batches", "1"], 29622)
    a, b = U.read_kmer_model(str(tmp_path / "single" / "trained_0_1.model")), U.read_kmer_model(str(tmp_path / "multi" / "trained_0_1.model"))
    assert list(a) == list(b)
    am, bm = np.array([v for v in a.values()]), np.array([v for v in b.values()])
    assert np.abs(am - bm).max() <= 1e-9   # same statistics, different summation order
    ra, rb = open(tmp_path / "single" / "params.csv").read().splitlines(), open(tmp_path / "multi" / "params.csv").read().splitlines()
    assert ra[0] == rb[0] and rb[1].startswith("0,1,8,")
    assert abs(float(ra[1].split(",")[-1]) - float(rb[1].split(",")[-1])) <= 1e-6
    # pooled statistics are summed in a fixed order on every rank (pool_stats.hip): the same job again writes the SAME
    # model file, byte for byte (rounds 1-3: fp64 atomics, last digits differed run to run)
    _torchrun("dynamont_amd.segmentation.train", common + ["-o", str(tmp_path / "multi2"), "--batch_size", "4", "--max_batches", "1"], 29623)
    assert open(tmp_path / "multi" / "trained_0_1.model", "rb").read() == open(tmp_path / "multi2" / "trained_0_1.model", "rb").read()
    trn.main(common + ["-o", str(tmp_path / "single2"), "--batch_size", "8", "--max_batches", "1"])
    assert open(tmp_path / "single" / "trained_0_1.model", "rb").read() == open(tmp_path / "single2" / "trained_0_1.model", "rb").read()


@pytest.mark.parametrize("ranks", [2, 4])
def test_bench_gpus_n_launches_its_own_ranks(tmp_path, ranks):
    """`python bench.py --gpus N` WITHOUT a launcher (what a driver may type): bench.py starts its N ranks itself as a
    child torch.distributed.run and prints ONE line that says n_gpus N. Rehearsal hooks: every rank on cuda:0, gloo for torch's
    bracket collectives; the exchange of every step is the REAL one -- dyn_comm_gather_counts / _rows over RCCL, a NCCL_HOSTID
    per rank (sockets over loopback) -- beside resident sessions that share the card's CUs between the ranks; the line must SAY
    that it is a one-device rehearsal and name dyn_comm_*. Without the hooks the same command must refuse
    a box with one device instead of measuring one GPU and calling it N. (4 ranks: a GPU box lets one job put 6 processes on its
    card, this test runner included; gloo at world 8 runs on the CPU, tests/test_parallel_gloo.py.)"""
    import json
    if _exchange() != "dyn_comm":
        pytest.skip("RCCL over loopback sockets unavailable on this box: bench.py --gpus N has no other exchange to rehearse with")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(DYN_BENCH_ONE_DEVICE="1", DYN_BENCH_BACKEND="gloo", PYTHONPATH=ROOT)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--workload", "cfg2_small",
           "--no-plain"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["rccl_ranks"] == ranks and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["strict_mode"] == "ties" and len(d["per_rank_ms"]) == ranks
    assert "ONE-DEVICE REHEARSAL" in d["exchange"]["implementation"] and "dyn_comm_* (dynamont_amd/csrc/rccl_comm.cpp)" in d["exchange"]["implementation"]
    assert d["exchange"]["rows_gathered_rank0"] > 0 and d["collective_backend"].startswith("rccl (dyn_comm_*)")
    if ranks != 2:
        return
    env.pop("DYN_BENCH_ONE_DEVICE")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode != 0 and "refusing to measure fewer GPUs" in r.stderr, r.stderr[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
