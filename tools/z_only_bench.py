#!/usr/bin/env python
"""GPU box: time of align(calc_probabilities=False) -- Z only -- on cfg2's 1 024 reads (kernel time per launch)."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamont_amd import Aligner, synth
d = tempfile.mkdtemp()
model = synth.write_model(os.path.join(d, "syn9.model"), 9, seed=7, stdev=0.15)
_, mean, sd = synth.read_model_file(model)
cfg = synth.CONFIGS["cfg2"]
reads = synth.make_reads(cfg["seed"], 1024, cfg["pore"], mean, sd, cfg["n_bases"])
packed = synth.pack_reads(reads)
al = Aligner(model, cfg["pore"], device=0)
ms = []
for rep in range(6):
    t = al.align_async(*packed, False)
    t.wait()
    tm = t.timing()
    t.close()
    ms.append(tm["ms_dp"])
print("align(calc=false), 1024 x 20 k: kernel ms per launch %s -> %.0f Msamp/s" % ([round(x, 2) for x in ms], tm["samples"] / min(ms[1:]) / 1e3))
