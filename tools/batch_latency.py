#!/usr/bin/env python
"""Wall time of Batch.align() for consecutive DIFFERENT batches (what the CLI sees), next to the
kernel time from HIP events: shows workspace (re)allocation and host staging costs."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamont_amd import synth, Aligner

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
d = tempfile.mkdtemp(prefix="dyn_lat_")
model = synth.write_model(os.path.join(d, "m9.model"), 9)
_, mean, sd = synth.read_model_file(model)
al = Aligner(model, "rna004", device=0)
res = None
for rep in range(6):
    reads = synth.make_reads(100 + rep, n, "rna004", mean, sd, 2000)
    sig, seq = [r.signal for r in reads], [r.sequence for r in reads]
    t0 = time.time()
    b = al.batch(sig, seq)
    t1 = time.time()
    b.align(True)
    t2 = time.time()
    res = b.fetch(res)  # refill the previous result object (what the CLI does)
    t3 = time.time()
    tm = b.timing()
    b.close() if hasattr(b, "close") else None
    print(f"batch {rep}: rows {sum(len(s) for s in sig)}  create {t1-t0:.3f}s  align {t2-t1:.3f}s  fetch {t3-t2:.3f}s  kernels {tm}", flush=True)
