#!/usr/bin/env python
"""Does overlapping k_backward of one half-batch with k_forward of another pay? Two handles (two
streams) on one GPU, 512 reads each, aligned concurrently from two threads with a phase offset,
against one handle with the 1 024 reads in one launch per kernel."""
import os, sys, time, tempfile, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamont_amd import synth, Aligner

d = tempfile.mkdtemp(prefix="dyn_ovl_")
model = synth.write_model(os.path.join(d, "m9.model"), 9, seed=7, stdev=0.15)
_, mean, sd = synth.read_model_file(model)
reads = synth.make_reads(2, 1024, "rna004", mean, sd, 2000)
sig, seq = [r.signal for r in reads], [r.sequence for r in reads]
samples = sum(len(s) for s in sig)
STEPS = 6

one = Aligner(model, "rna004", device=0)
b = one.batch(sig, seq)
b.align(True)
t0 = time.time()
for _ in range(STEPS):
    b.align(True)
t1 = time.time() - t0
print(f"one handle, 1024 reads/launch: {t1/STEPS*1e3:.2f} ms/step  {samples*STEPS/t1/1e6:.1f} Msamp/s", flush=True)
del b

halves = [(sig[:512], seq[:512]), (sig[512:], seq[512:])]
handles = [Aligner(model, "rna004", device=0) for _ in range(2)]
batches = [h.batch(*hv) for h, hv in zip(handles, halves)]
for bb in batches:
    bb.align(True)

def run(bb, delay):
    time.sleep(delay)
    for _ in range(STEPS):
        bb.align(True)

for delay in (0.0, 0.012):
    th = [threading.Thread(target=run, args=(batches[0], 0.0)), threading.Thread(target=run, args=(batches[1], delay))]
    t0 = time.time()
    for x in th: x.start()
    for x in th: x.join()
    t2 = time.time() - t0 - delay
    print(f"two handles x 512 reads, phase offset {delay*1e3:.0f} ms: {t2/STEPS*1e3:.2f} ms per 1024 reads  {samples*STEPS/t2/1e6:.1f} Msamp/s", flush=True)
