#!/usr/bin/env python
"""End-to-end timing of the dynamont-train counterpart on a synthetic dataset (cProfile of the main thread).

    python tools/cli_train_bench.py [n_reads] [batch_size] [extra flags, e.g. --aggregate pooled]
"""
import cProfile, os, pstats, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamont_amd import synth
from dynamont_amd.segmentation import train as tr

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 24
extra = sys.argv[3:]
d = tempfile.mkdtemp(prefix="dyn_train_e2e_")
model = synth.write_model(os.path.join(d, "m9.model"), 9)
_, mean, sd = synth.read_model_file(model)
reads = synth.make_reads(5, n, "rna004", mean, sd, 2000)
raw, bam, _ = synth.write_dataset(os.path.join(d, "in"), "ds", reads, "rna004", seed=1)
samples = sum(len(r.signal) for r in reads)
del reads
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
tr.main(["-r", os.path.join(d, "in"), "-b", bam, "-o", os.path.join(d, "out"), "-p", "rna004", "--model_path", model,
         "--batch_size", str(bs), "-q", "0", "--no-timestamp"] + extra)
pr.disable()
dt = time.time() - t0
print(f"{n} reads, batch_size {bs}, flags {extra}: {dt:.2f} s -> {samples/dt/1e6:.2f} Msamp/s, {n/dt:.1f} reads/s, {dt/max(1, n // bs):.2f} s per batch", flush=True)
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
