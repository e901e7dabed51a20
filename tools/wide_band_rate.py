#!/usr/bin/env python
"""GPU box: what the generic wide-band kernel (wide_band.hip) delivers -- 256 rna004 reads of ~20 k samples (2 000 bases) at band
1 000 (half band 500: every read wide) and, for scale, the same reads at band 400 through the tuned sweeps. Not a bench line:
no caller of the reference uses a band above 400."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamont_amd import Aligner, synth
d = tempfile.mkdtemp()
model = synth.write_model(os.path.join(d, "m9.model"), 9, seed=7, stdev=0.15)
_, mean, sd = synth.read_model_file(model)
reads = synth.make_reads(77, 256, "rna004", mean, sd, 2000)
packed = synth.pack_reads(reads)
samples = int(packed[1][-1])
for band in (400, 1000, 2000):
    al = Aligner(model, "rna004", band=band, device=0)
    with al.batch_packed(*packed) as b:
        b.align(True)                      # warm
        t0 = time.perf_counter(); b.align(True); dt = time.perf_counter() - t0
        tm = b.timing()
        t0 = time.perf_counter(); b.train(); dtt = time.perf_counter() - t0
    print("band %4d: align(calc=true) %.3f s = %.1f Msamp/s, %.2f G in-band cells/s; train %.3f s = %.1f Msamp/s" % (band, dt, samples / dt / 1e6, tm["cells"] / dt / 1e9, dtt, samples / dtt / 1e6), flush=True)
    al.close()
