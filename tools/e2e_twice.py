#!/usr/bin/env python
"""GPU box: the dynamont-resquiggle counterpart twice in ONE process on a 32 768-read .pod5 + BAM dataset, the second run
(lattice pool parked by the first: what bench.py's e2e_cli record times) under cProfile. Round 4: 1.92 s end to end, of
which the 32 launches take 1.63 s; the main thread is busy for ~1.2 s of it (BAM parse 30 us per read, slicing 15 us) --
the run is GPU-bound, the Python front end would become the limit at ~25 000 reads/s."""
import cProfile, os, pstats, sys, tempfile, time
sys.path.insert(0, "/root/repo")
import numpy as np
from dynamont_amd import synth
from dynamont_amd.segmentation import segment as seg
d = tempfile.mkdtemp(prefix="dyn_e2e_")
model = synth.write_model(os.path.join(d, "m9.model"), 9)
_, mean, sd = synth.read_model_file(model)
reads = synth.make_reads(5, 4096, "rna004", mean, sd, 2000)
raw, bam, _ = synth.write_dataset(os.path.join(d, "in"), "ds", reads, "rna004", seed=1, container="pod5", replicate=8, basecalls="bam")
samples = sum(len(r.signal) for r in reads) * 8
del reads
args = ["-r", os.path.join(d, "in"), "-b", bam, "--mode", "basic", "-p", "rna004", "--model_path", model]
seg.main(args + ["-o", os.path.join(d, "out0.csv")])
os.environ["DYN_TRACE_HOST"] = "0"
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
seg.main(args + ["-o", os.path.join(d, "out1.csv")])
pr.disable()
dt = time.time() - t0
print(f"second run: {dt:.3f} s -> {samples/dt/1e6:.1f} Msamp/s")
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
