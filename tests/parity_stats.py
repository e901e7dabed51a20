#!/usr/bin/env python
"""Test infrastructure (run by hand on a GPU box: `python tests/parity_stats.py`; not collected by pytest).
How close is the HIP path to the CPU oracle (itself bit-identical to the compiled reference)?
Random reads of every pore type + dense reads: integer columns, max |d posterior|, max relative |d Z|."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from dynamont_amd import synth, Aligner
from oracle.pyoracle import Oracle

d = tempfile.mkdtemp(prefix="dyn_par_")
m5 = synth.write_model(os.path.join(d, "m5.model"), 5, seed=7, stdev=0.25)
m9 = synth.write_model(os.path.join(d, "m9.model"), 9, seed=7, stdev=0.15)
tot = dict(reads=0, segs=0, seg_mismatch=0, z_equal=0, dz=0.0, dp=0.0, dp_m=0.0)
for pore, nb, dwell in (("rna002", (30, 400), None), ("rna004", (100, 900), None), ("dna_r9", (20, 300), None),
                        ("dna_r10_260bps", (300, 1200), None), ("dna_r10_400bps", (450, 700), None),
                        ("rna004", (430, 900), 0.5), ("dna_r9", (430, 700), 2.5), ("rna004", 2000, None)):
    pid, rna, k = synth.PORES[pore]
    model = m5 if k == 5 else m9
    _, mean, sd = synth.read_model_file(model)
    reads = synth.make_reads(hash((pore, str(nb), dwell)) % 10000, 3 if nb == 2000 else 16, pore, mean, sd, nb, dwell=dwell)
    al, orc = Aligner(model, pore, device=0), Oracle(model, pid)
    res = al.align_batch([r.signal for r in reads], [r.sequence for r in reads], True)
    for i, r in enumerate(reads):
        try:
            want = orc.align(r.signal, r.sequence, True)
        except RuntimeError:
            continue
        got = res.read(i)
        tot["reads"] += 1
        tot["segs"] += len(want["signal_positions"])
        same = np.array_equal(got["signal_positions"], want["signal_positions"]) and np.array_equal(got["sequence_positions"], want["sequence_positions"])
        tot["seg_mismatch"] += 0 if same else 1
        tot["z_equal"] += int(got["Z"] == want["Z"])
        tot["dz"] = max(tot["dz"], abs(got["Z"] - want["Z"]) / max(1.0, abs(want["Z"])))
        if same:
            tot["dp"] = max(tot["dp"], float(np.abs(got["probabilities"] - want["probabilities"]).max()))
print(tot)
