#!/usr/bin/env python
"""GPU box: the dynamont-resquiggle counterpart on the bench's e2e dataset (32 768 reads, .pod5 + BAM) as ONE process and
as TWO ranks under torch.distributed.run (both on cuda:0, gloo: a 1-GPU box cannot host two RCCL ranks). Cold processes:
interpreter start, imports, model load and pool allocation are inside the wall times. What it shows on one GPU: the
two-rank job is no slower than the single process (each rank compresses its own rows; rank 0 only appends bytes), the
output is one zstd frame with the same rows."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, "/root/repo")
from dynamont_amd import synth, zstd_io
d = tempfile.mkdtemp(prefix="dyn_mr_")
model = synth.write_model(os.path.join(d, "m9.model"), 9, seed=7, stdev=0.15)
_, mean, sd = synth.read_model_file(model)
reads = synth.make_reads(5, 4096, "rna004", mean, sd, 2000)
raw, bam, _ = synth.write_dataset(os.path.join(d, "in"), "ds", reads, "rna004", seed=1, container="pod5", replicate=8, basecalls="bam")
samples = sum(len(r.signal) for r in reads) * 8
del reads
args = ["-r", os.path.join(d, "in"), "-b", bam, "--mode", "basic", "-p", "rna004", "--model_path", model]
env = dict(os.environ, PYTHONPATH="/root/repo")
def run(cmd, out, extra_env=None):
    t0 = time.perf_counter()
    r = subprocess.run(cmd + args + ["-o", out], env=dict(env, **(extra_env or {})), capture_output=True, text=True)
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr[-2000:]
    return dt
one = run([sys.executable, "-m", "dynamont_amd.segmentation.segment"], os.path.join(d, "one.csv"))
two = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29655",
           "-m", "dynamont_amd.segmentation.segment"], os.path.join(d, "two.csv"), {"DYN_DIST_BACKEND": "gloo", "DYN_DIST_ONE_DEVICE": "1"})
print(f"one process : {one:.2f} s wall (cold) = {samples / one / 1e6:.0f} Msamp/s")
print(f"two ranks   : {two:.2f} s wall (cold, incl. torch.distributed.run) = {samples / two / 1e6:.0f} Msamp/s")
a, b = open(os.path.join(d, "one.csv.zst"), "rb").read(), open(os.path.join(d, "two.csv.zst"), "rb").read()
print("bytes:", len(a), len(b), "frames:", zstd_io.count_frames(a), zstd_io.count_frames(b))
import hashlib
def digest(blob):  # order-independent digest of the rows
    import zlib
    acc = 0; n = 0
    for line in zstd_io.decompress(blob).split(b"\n"):
        acc ^= int.from_bytes(hashlib.blake2b(line, digest_size=8).digest(), "little"); n += 1
    return n, acc
print("rows, xor of row hashes:", digest(a[:]), digest(b[:]))
