#!/usr/bin/env python
"""TEST INFRASTRUCTURE (uses the CPU oracle; run by hand on a GPU box, not collected by pytest):

    python tests/parity_full.py --workload cfg4_share --out gpurun_out/parity_cfg4_share.json [--procs 16]
    python tests/parity_full.py --workload cfg5_share --train --out gpurun_out/parity_cfg5_train.json

EVERY read of a full-size workload against the oracle (itself bit-identical to the compiled reference): integer
columns, posteriors, Z. The GPU takes a fraction of a second; the oracle takes ~2 s per 20 k-sample read per core,
so 4 096 reads are ~9 minutes on the box's 16 host cores. Progress is printed every 256 reads."""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_O = None
_READS = None


def _init(model, pore_id):
    global _O
    from oracle.pyoracle import Oracle
    _O = Oracle(model, pore_id)


def _work(i):
    r = _READS[i]
    a = _O.align(r.signal, r.sequence, True)
    return i, a["Z"], a["signal_positions"].astype(np.uint32), a["sequence_positions"].astype(np.uint32), a["probabilities"]


def _work_train(i):
    r = _READS[i]
    t = _O.train(r.signal, r.sequence, dense=False)
    return i, t["Z"], t["m1"], t["e2"], t["weight"], t["sum"]


def main_train(a):
    """Every read of cfg5's per-GPU share through train(): Z, transitions, per-k-mer weights and means."""
    global _READS
    from dynamont_amd import Aligner, synth
    cfg = synth.CONFIGS["cfg5"]
    n = a.reads or 1024
    pore = cfg["pore"]
    pid, _, k = synth.PORES[pore]
    tmp = tempfile.mkdtemp(prefix="parity_")
    model = synth.write_model(os.path.join(tmp, f"syn{k}.model"), k, seed=7, stdev=0.15)
    _, mean, sd = synth.read_model_file(model)
    _READS = synth.make_reads(cfg["seed"], n, pore, mean, sd, cfg["n_bases"], polya=cfg.get("polya"))
    al = Aligner(model, pore, device=0)
    sig, so, sq, qo = synth.pack_reads(_READS)
    t = al.train_async(sig, so, sq, qo, pooled=False)
    res = t.wait()
    t.close()
    tot = dict(reads=0, status_not_ok=int((res.status != 0).sum()), kmer_sets_differ=0, max_rel_dZ=0.0, max_abs_dtransition=0.0,
               max_rel_dweight=0.0, max_abs_dmean=0.0, max_rel_weight_sum_minus_samples=0.0)
    t0 = time.time()
    with mp.get_context("fork").Pool(a.procs, initializer=_init, initargs=(model, pid)) as pool:
        for done, (i, Z, m1, e2, w, s1) in enumerate(pool.imap_unordered(_work_train, range(n), chunksize=2), 1):
            code, m, _ = res.sparse(i)
            off = int(res.em_offsets[i])
            gw = res.em_weight[off:off + len(code)]
            touched = np.nonzero(w > 0)[0]
            tot["reads"] += 1
            tot["max_rel_dZ"] = max(tot["max_rel_dZ"], abs(res.Z[i] - Z) / max(1.0, abs(Z)))
            tot["max_abs_dtransition"] = max(tot["max_abs_dtransition"], abs(res.transitions[3 * i] - m1), abs(res.transitions[3 * i + 2] - e2))
            tot["max_rel_weight_sum_minus_samples"] = max(tot["max_rel_weight_sum_minus_samples"], abs(gw.sum() / len(_READS[i].signal) - 1.0))
            if not np.array_equal(code, touched):
                tot["kmer_sets_differ"] += 1
                continue
            tot["max_rel_dweight"] = max(tot["max_rel_dweight"], float(np.max(np.abs(gw - w[touched]) / w[touched])))
            heavy = w[touched] > 1e-3
            tot["max_abs_dmean"] = max(tot["max_abs_dmean"], float(np.max(np.abs(m[heavy] - (s1[touched] / w[touched])[heavy]))))
            if done % 64 == 0:
                msg = f"{done}/{n} reads checked, k-mer sets differing so far: {tot['kmer_sets_differ']}"
                print(msg, flush=True)
                with open(a.out + ".progress", "a") as wf:
                    wf.write(msg + "\n")
    out = {"workload": f"cfg5_share: {n} reads (seed {cfg['seed']}), {pore}, synthetic {k}-mer model, ONE batch through dyn_batch_train_async, every read against the oracle's train()",
           "oracle": {"procs": a.procs, "wall_s": round(time.time() - t0, 1)}, **tot,
           "note": "relative weight differences are the ORACLE's log-space rounding noise (tests/extended_precision_train.py)"}
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out))


def main():
    global _READS
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg4_share", choices=["cfg2", "cfg2_polya", "cfg4_share", "cfg3", "cfg5_share"])
    ap.add_argument("--train", action="store_true")
    ap.add_argument("--reads", type=int, default=0)
    ap.add_argument("--procs", type=int, default=min(16, os.cpu_count() or 1))
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    if a.train:
        return main_train(a)
    from dynamont_amd import Aligner, synth
    cfgname, n = {"cfg2": ("cfg2", 1024), "cfg2_polya": ("cfg2_polya", 1024), "cfg4_share": ("cfg4", 4096), "cfg3": ("cfg3", 4096)}[a.workload]
    n = a.reads or n
    cfg = synth.CONFIGS[cfgname]
    pore = cfg["pore"]
    pid, _, k = synth.PORES[pore]
    tmp = tempfile.mkdtemp(prefix="parity_")
    model = synth.write_model(os.path.join(tmp, f"syn{k}.model"), k, seed=7, stdev=0.15)
    _, mean, sd = synth.read_model_file(model)
    _READS = synth.make_reads(cfg["seed"], n, pore, mean, sd, cfg["n_bases"], polya=cfg.get("polya"))
    al = Aligner(model, pore, device=0)
    t0 = time.time()
    sig, so, sq, qo = synth.pack_reads(_READS)
    t = al.align_async(sig, so, sq, qo, True)
    res = t.wait()
    tm = t.timing()
    t.close()
    gpu_s = time.time() - t0
    tot = dict(reads=0, segments=0, reads_with_integer_mismatch=0, z_bit_equal=0, max_rel_dZ=0.0, max_abs_dprob=0.0, status_not_ok=int((res.status != 0).sum()))
    t0 = time.time()
    with mp.get_context("fork").Pool(a.procs, initializer=_init, initargs=(model, pid)) as pool:
        for done, (i, Z, sp, qp, pr) in enumerate(pool.imap_unordered(_work, range(n), chunksize=2), 1):
            got = res.read(i)
            same = np.array_equal(got["signal_positions"], sp) and np.array_equal(got["sequence_positions"], qp)
            tot["reads"] += 1
            tot["segments"] += len(sp)
            tot["reads_with_integer_mismatch"] += 0 if same else 1
            tot["z_bit_equal"] += int(got["Z"] == Z)
            tot["max_rel_dZ"] = max(tot["max_rel_dZ"], abs(got["Z"] - Z) / max(1.0, abs(Z)))
            if same and len(pr):
                tot["max_abs_dprob"] = max(tot["max_abs_dprob"], float(np.abs(got["probabilities"] - pr).max()))
            if done % 64 == 0:  # progress on stdout AND in a file next to the report (a silent run is taken to be hung)
                msg = f"{done}/{n} reads checked, mismatching so far: {tot['reads_with_integer_mismatch']}"
                print(msg, flush=True)
                with open(a.out + ".progress", "a") as w:
                    w.write(msg + "\n")
    out = {"workload": f"{a.workload}: {n} reads (seed {cfg['seed']}), {pore}, synthetic {k}-mer model, ONE batch through dyn_batch_align_async",
           "gpu": {"wall_s": round(gpu_s, 3), "launches": tm["launches"], "lp_inplace": tm["lp_inplace"], "n_static": tm["n_static"]},
           "oracle": {"procs": a.procs, "wall_s": round(time.time() - t0, 1)}, **tot}
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
