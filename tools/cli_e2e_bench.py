#!/usr/bin/env python
"""End-to-end timing of the dynamont-resquiggle counterpart on a synthetic dataset."""
import cProfile, os, pstats, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from dynamont_amd import synth
from dynamont_amd.segmentation import segment as seg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
extra = sys.argv[2:]  # e.g. --parallel-zstd-frames
container = "npz"
if "--container" in extra:  # --container pod5: the raw signal in a .pod5 file (VBZ chunks, vendor-free reader)
    k = extra.index("--container")
    container = extra[k + 1]
    del extra[k:k + 2]
d = tempfile.mkdtemp(prefix="dyn_e2e_")
model = synth.write_model(os.path.join(d, "m9.model"), 9)
_, mean, sd = synth.read_model_file(model)
t0 = time.time()
reads = synth.make_reads(5, n, "rna004", mean, sd, 2000)
raw, bam, _ = synth.write_dataset(os.path.join(d, "in"), "ds", reads, "rna004", seed=1, container=container)
samples = sum(len(r.signal) for r in reads)
print(f"dataset: {n} reads, {samples/1e6:.1f} Msamples, container {container} ({os.path.getsize(raw)/1e6:.0f} MB), generated in {time.time()-t0:.1f} s", flush=True)
del reads
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
seg.main(["-r", os.path.join(d, "in"), "-b", bam, "-o", os.path.join(d, "out.csv"), "--mode", "basic", "-p", "rna004",
          "--model_path", model, "--batch-reads", "1024"] + extra)
pr.disable()
dt = time.time() - t0
print(f"flags {extra}: end to end: {dt:.2f} s -> {samples/dt/1e6:.1f} Msamp/s, {n/dt:.0f} reads/s, output {os.path.getsize(os.path.join(d,'out.csv.zst'))/1e6:.1f} MB")
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
