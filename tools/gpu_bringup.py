#!/usr/bin/env python
"""GPU bring-up: run the HIP path on the golden vectors and print diagnostics."""
import os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamont_amd import Aligner, synth

G = os.path.join(ROOT, "tests", "golden")
tmp = tempfile.mkdtemp()
m5 = synth.write_model(os.path.join(tmp, "syn5.model"), 5, seed=7, stdev=0.25)
m9 = synth.write_model(os.path.join(tmp, "syn9.model"), 9, seed=7, stdev=0.15)

def cmp(tag, res, g, p):
    ok_int = np.array_equal(res["sequence_positions"], g[p + "seqpos"]) and np.array_equal(res["signal_positions"], g[p + "sigpos"])
    n = min(len(res["probabilities"]), len(g[p + "prob"]))
    dp = np.abs(res["probabilities"][:n] - g[p + "prob"][:n]).max() if n else 0
    dz = abs(res["Z"] - float(g[p + "Z"]))
    nd = int((res["signal_positions"][:n] != g[p + "sigpos"][:n]).sum())
    print(f"{tag}: ints {'OK' if ok_int else 'MISMATCH'} nseg {len(res['probabilities'])}/{len(g[p+'prob'])} sigpos diffs {nd} max|dprob| {dp:.3e} |dZ| {dz:.3e} Z {res['Z']:.6f}")
    return ok_int

g = np.load(os.path.join(G, "g1_cfg1.npz"))
a5 = Aligner(m5, "rna002")
r = a5.align(g["syn_signal"], str(g["syn_sequence"]), True)
cmp("G1 syn", r, g, "syn_")
r0 = a5.align(g["syn_signal"], str(g["syn_sequence"]), False)
print("calc=false Z", r0["Z"], len(r0["probabilities"]))

g3 = np.load(os.path.join(G, "g3_short.npz"))
bad = 0
als = {}
for i in range(int(g3["n_cases"])):
    p = f"c{i}_"
    pore = str(g3[p + "pore"])
    band = int(g3[p + "band"]) if p + "band" in g3 else 400
    key = (pore, band)
    if key not in als:
        als[key] = Aligner(m5 if pore == "rna002" else m9, pore, band=band)
    try:
        r = als[key].align(g3[p + "signal"], str(g3[p + "sequence"]), True)
        if not cmp(f"G3 case {i} {pore} band {band} N={len(str(g3[p+'sequence']))-als[key].kmer_size+2}", r, g3, p): bad += 1
    except Exception as e:
        print("G3 case", i, "EXC", e); bad += 1
print("G3 bad:", bad)

g2 = np.load(os.path.join(G, "g2_rna004.npz"))
_, mean, sd = synth.read_model_file(m9)
reads = synth.make_reads(22, 16, "rna004", mean, sd, (200, 2000))
a9 = Aligner(m9, "rna004")
t0 = time.time()
res = a9.align_batch([r.signal for r in reads], [r.sequence for r in reads], True)
print("G2 batch time", time.time() - t0)
for i in range(16):
    if res.status[i] != 0:
        print("G2", i, "status", res.status[i], res.error(i)); continue
    cmp(f"G2 r{i}", res.read(i), g2, f"r{i}_")
