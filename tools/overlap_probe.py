#!/usr/bin/env python
"""GPU box: do the read-queue launches of CONSECUTIVE batches overlap when they come from two compute streams?

cfg2 batches (1 024 reads = one read per wave) in strict mode "ties": a launch lasts as long as its slowest (tie) read,
the other three quarters of the waves idle for the last quarter of it. Two handles (each has its own stream and lattice
pool) fed alternately vs one handle -- the gain is what a second compute lane inside ONE handle can bring.

    python tools/overlap_probe.py [steps] [strict]
"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from dynamont_amd import Aligner, synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
strict = sys.argv[2] if len(sys.argv) > 2 else "ties"
d = tempfile.mkdtemp(prefix="ovl_")
model = synth.write_model(os.path.join(d, "m9.model"), 9)
_, mean, sd = synth.read_model_file(model)
batches = []
for j in range(4):
    reads = synth.make_reads(2 + 100003 * j, 1024, "rna004", mean, sd, 2000)
    batches.append(synth.pack_reads(reads))
samples = [int(b[1][-1]) for b in batches]


def run(handles, depth):
    import collections
    q = collections.deque()
    free = {id(h): [] for h in handles}
    t0 = time.perf_counter()
    n = 0
    for s in range(steps):
        h = handles[s % len(handles)]
        if len(q) >= depth:
            hh, t = q.popleft()
            res = t.wait()
            t.close()
            free[id(hh)].append(res)
        sig, so, sq, qo = batches[s % len(batches)]
        out = free[id(h)].pop() if free[id(h)] else None
        q.append((h, h.align_async(sig, so, sq, qo, True, out=out)))
        n += samples[s % len(batches)]
    while q:
        hh, t = q.popleft()
        t.wait()
        t.close()
    dt = time.perf_counter() - t0
    return n / dt / 1e6, dt / steps * 1e3


a = Aligner(model, "rna004", device=0)
a.set_strict(strict)
run([a], 3)
one = run([a], 3)
print(f"one handle, 3 in flight: {one[0]:.1f} Msamp/s, {one[1]:.2f} ms per batch", flush=True)
b = Aligner(model, "rna004", device=0)
b.set_strict(strict)
run([a, b], 4)
two = run([a, b], 4)
print(f"two handles alternating, 4 in flight: {two[0]:.1f} Msamp/s, {two[1]:.2f} ms per batch  (x{two[0] / one[0]:.3f})", flush=True)
