#!/usr/bin/env python
"""GPU box: the dynamont-resquiggle counterpart over and over in ONE process on small random datasets -- reads of very
different length, sequences with an invalid nucleotide, ids no raw file holds, slices cut too short, two raw files whose
reads interleave -- with random --batch-reads and --depth, handles created and destroyed every run (parked pools and
buffers change hands, tickets merge or not). Every run's rows and error lines must equal those of a plain run of the same
dataset (batches of 64, one in flight, DYN_NO_MERGE=1). Recycled buffers, merged launches and failed reads meet here the
way they do in a long production run; `pytest -m gpu` meets each of them once.

    python tools/soak_cli.py [iterations] [seed]
"""
import hashlib, os, sys, tempfile, time, uuid
sys.path.insert(0, "/root/repo")
import numpy as np
from dynamont_amd import bam_io, synth, zstd_io
from dynamont_amd.pod5_io import iter_basecalls
from dynamont_amd.segmentation import segment as seg

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
root = tempfile.mkdtemp(prefix="dyn_soak_")
model = synth.write_model(os.path.join(root, "m9.model"), 9, seed=7, stdev=0.15)
_, mean, sd = synth.read_model_file(model)


def digest(path):
    rows = zstd_io.decompress(open(path, "rb").read()).split(b"\n")
    acc = 0
    for r in rows[1:]:
        acc ^= int.from_bytes(hashlib.blake2b(r, digest_size=8).digest(), "little")
    return rows[0], len(rows), acc


def errors(path):
    return sorted(open(path).read().splitlines()) if os.path.exists(path) else []


t0 = time.time()
for it in range(iters):
    d = os.path.join(root, f"it{it}")
    n_a, n_b = int(rng.integers(20, 400)), int(rng.integers(5, 200))
    hi = int(rng.choice([300, 1200, 4000]))
    _, bam_a, _ = synth.write_dataset(os.path.join(d, "in"), "a", synth.make_reads(1000 + it, n_a, "rna004", mean, sd, (40, hi)), "rna004",
                                      seed=it, container="pod5", basecalls="bam", pod5_chunk_samples=int(rng.integers(500, 20000)))
    _, bam_b, _ = synth.write_dataset(os.path.join(d, "in"), "b", synth.make_reads(2000 + it, n_b, "rna004", mean, sd, (40, hi), polya=(20, 150)), "rna004",
                                      seed=100 + it, container="pod5", basecalls="bam", pod5_chunk_samples=int(rng.integers(500, 20000)))
    os.environ["DYN_PY_BAM"] = "1"
    recs = [(r.query_name, r.query_sequence, dict(r._tags)) for f in (bam_a, bam_b) for r in iter_basecalls(f)]
    del os.environ["DYN_PY_BAM"]
    order = rng.permutation(len(recs))
    recs = [recs[i] for i in order]
    for i in rng.choice(len(recs), size=max(1, len(recs) // 15), replace=False):   # damage some
        name, seq, tags = recs[i]
        kind = int(rng.integers(0, 5))
        if kind == 0 and len(seq) > 20:
            p = int(rng.integers(0, len(seq)))
            seq = seq[:p] + "N" + seq[p + 1:]
        elif kind == 1:
            name = str(uuid.uuid4())
        elif kind == 2:
            tags = dict(tags, ns=tags["ts"] + int(rng.integers(0, 40)))
        elif kind == 3:
            seq = seq[:int(rng.integers(1, 9))]
        else:
            tags = dict(tags, sm=500.0, sd=80.0)
        recs[i] = (name, seq, tags)
    bam = os.path.join(d, "mixed.bam")
    bam_io.write_bam(bam, recs)
    base = ["-r", os.path.join(d, "in"), "-b", bam, "--mode", "basic", "-p", "rna004", "--model_path", model]
    os.environ["DYN_NO_MERGE"] = "1"
    seg.main(base + ["-o", os.path.join(d, "ref.csv"), "--batch-reads", "64", "--depth", "1"])
    del os.environ["DYN_NO_MERGE"]
    want = digest(os.path.join(d, "ref.csv.zst")), errors(os.path.join(d, "ref.errors"))
    for rep in range(3):
        br, dp = int(rng.choice([1, 7, 33, 128, 500, 1500])), int(rng.integers(1, 13))
        strict = str(rng.choice(["ties", "ties", "all", "off"]))
        out = os.path.join(d, f"run{rep}.csv")
        seg.main(base + ["-o", out, "--batch-reads", str(br), "--depth", str(dp), "--strict-ties", strict])
        got = digest(out + ".zst"), errors(os.path.join(d, f"run{rep}.errors"))
        same = got == want if strict != "off" else (got[0][:2] == want[0][:2] and got[1] == want[1])  # plain arithmetic: same rows up to the rare tie
        print(f"iteration {it} run {rep}: {len(recs)} records, batch {br}, depth {dp}, strict {strict}: rows {got[0][1]}, errors {len(got[1])} -> {'ok' if same else 'MISMATCH'}", flush=True)
        assert same, (want[0][1:], got[0][1:], len(want[1]), len(got[1]))
    seg.close_raw_cache()
print(f"soak done: {iters} datasets x 3 runs in {time.time() - t0:.0f} s, all equal to the plain run")
