#!/usr/bin/env python
"""GPU box: how much fp64 range the linear-domain training sweeps need. DYN_LIN_PARK moves the exponent at which the
row maximum is parked (the range below it is 2^(1074 + park)); reads that lose posterior mass are redone in the log
domain and counted (dyn_timing.reads_log_redo). Prints the share of cfg5 reads redone per parking exponent.

    python tools/lin_range_curve.py [n_reads]
"""
import math
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from dynamont_amd import Aligner, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
d = tempfile.mkdtemp()
model = synth.write_model(os.path.join(d, "syn9.model"), 9, seed=7, stdev=0.15)
_, mean, sd = synth.read_model_file(model)
cfg = synth.CONFIGS["cfg5"]
reads = synth.make_reads(cfg["seed"], n, cfg["pore"], mean, sd, cfg["n_bases"])
sig, so, sq, qo = synth.pack_reads(reads)
al = Aligner(model, cfg["pore"], device=0)
for park in (-600, -500, -400, -300, -200, -100, 0, 300, 950):
    os.environ["DYN_LIN_PARK"] = str(park)
    t = al.train_async(sig, so, sq, qo, pooled=False, emissions=False)
    res = t.wait()
    tm = t.timing()
    t.close()
    print("park 2^%5d  range below the row maximum e^%4.0f  reads redone %5d of %d  (all ok: %s)" % (
        park, (1074 + park) * math.log(2), tm["reads_log_redo"], n, bool((res.status == 0).all())), flush=True)
