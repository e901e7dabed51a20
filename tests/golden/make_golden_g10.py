#!/usr/bin/env python
"""Generate golden fixture G10 -- structural-tie reads with the COMPILED REFERENCE's answers -- and the stability
record profiles/r03/tie_stability.json.

Runs only in the authoring container (needs /root/reference compiled into oracle/_ref by `make -C oracle ref`).

    python tests/golden/make_golden_g10.py

What it pins. Where two neighbouring lattice columns carry the same k-mer the traceback comparison
(NT_aligner_api.cpp:445-448) is a tie in exact arithmetic and the reference's decision rests on the last bits of its
logPlus (aligner.cpp:276-285). G10 holds 3 400 such reads in nine families (tests/tie_parity.py, G10_FAMILIES: read-start
ties for 5-mer and 9-mer, RNA and DNA pores; internal homopolymers of 20-120 bases) -- inputs are regenerated from
seeds, the fixture stores the reference's segment borders, Z and probabilities -- plus, per family,
  * `<fam>_product_deviates`: ids of the reads on which the oracle's control flow REPLAYED with the product's default
    arithmetic (dp_math.hpp, <= 1 ulp table softplus) takes a different border than the reference, and
  * `<fam>_strict_deviates`: the same with the strict arithmetic (dp_math_strict.hpp); must be empty.
The stability record answers "does the reference's own answer depend on its libm variant?": the same compiled
reference is run again in a child process under GLIBC_TUNABLES=glibc.cpu.hwcaps=-AVX2,-FMA (glibc's non-FMA exp, which
differs from the FMA variant in ~0.07 % of calls) and the borders / Z are compared read by read.
"""
from __future__ import annotations

import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dynamont_amd import synth  # noqa: E402
from oracle.pyoracle import Reference  # noqa: E402
import tie_parity  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
NOFMA = "glibc.cpu.hwcaps=-AVX2,-FMA"


def models(d):
    return tie_parity.g10_model_paths(d)


def run_reference(d):
    """family -> list of reference results (None for reads the reference rejects)"""
    mp = models(d)
    out = {}
    for fam, (pore, mkey, gen) in tie_parity.G10_FAMILIES.items():
        enum = synth.PORES[pore][0]
        _, mean, sd = synth.read_model_file(mp[mkey])
        reads = gen(mean, sd)
        out[fam] = (reads, tie_parity.reference_results(Reference(mp[mkey], enum, 400), reads))
    return out


def dump(path):
    d = tempfile.mkdtemp(prefix="g10w_")
    store = {}
    for fam, (_, res) in run_reference(d).items():
        store[fam + "_Z"] = np.array([np.nan if r is None else r["Z"] for r in res])
        store[fam + "_sig"] = np.concatenate([np.zeros(0, np.uint32)] + [r["signal_positions"].astype(np.uint32) for r in res if r is not None])
    np.savez(path, **store)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--dump":
        dump(sys.argv[2])
        return
    d = tempfile.mkdtemp(prefix="g10_")
    mp = models(d)
    so = tie_parity.build_replay(d)
    fams = run_reference(d)
    store, record = {}, {"glibc": os.confstr("CS_GNU_LIBC_VERSION"), "tunables_second_run": NOFMA, "families": {}}
    for fam, (reads, res) in fams.items():
        pore, mkey, _ = tie_parity.G10_FAMILIES[fam]
        enum = synth.PORES[pore][0]
        ok = [r is not None for r in res]
        seg_off = np.zeros(len(res) + 1, dtype=np.int64)
        for i, r in enumerate(res):
            seg_off[i + 1] = seg_off[i] + (0 if r is None else len(r["signal_positions"]))
        cat = lambda key, dt: np.concatenate([np.zeros(0, dt)] + [r[key].astype(dt) for r in res if r is not None])  # noqa: E731
        store[fam + "_ok"] = np.array(ok)
        store[fam + "_S"] = np.array([len(r.signal) for r in reads], dtype=np.int32)  # guards the regenerated inputs
        store[fam + "_seg_off"] = seg_off
        store[fam + "_sigpos"] = cat("signal_positions", np.uint32)
        store[fam + "_seqpos"] = cat("sequence_positions", np.uint32)
        store[fam + "_prob"] = cat("probabilities", np.float32)
        store[fam + "_Z"] = np.array([np.nan if r is None else r["Z"] for r in res])
        rp = tie_parity.Replay(so, mp[mkey], enum, 400, mode=1)
        dev_product = tie_parity.differing_reads(rp, reads, res)
        rp.set_mode(6)
        dev_strict = tie_parity.differing_reads(rp, reads, res)
        store[fam + "_product_deviates"] = np.array(dev_product, dtype=np.int32)
        store[fam + "_strict_deviates"] = np.array(dev_strict, dtype=np.int32)
        record["families"][fam] = dict(pore=pore, model=mkey, reads=len(reads), reads_ok=int(sum(ok)), segments=int(seg_off[-1]),
                                       replay_product_arithmetic_differs=dev_product, replay_strict_arithmetic_differs=dev_strict)
        print(fam, record["families"][fam], flush=True)
    # the reference again, under the non-FMA libm variants
    other = os.path.join(d, "nofma.npz")
    env = dict(os.environ, GLIBC_TUNABLES=NOFMA)
    subprocess.run([sys.executable, os.path.abspath(__file__), "--dump", other], check=True, env=env)
    o = np.load(other)
    for fam, (reads, res) in fams.items():
        z = np.array([np.nan if r is None else r["Z"] for r in res])
        sig = store[fam + "_sigpos"]
        seg_off = store[fam + "_seg_off"]
        same_len = len(o[fam + "_sig"]) == len(sig)
        diff_reads = []
        if same_len:
            neq = o[fam + "_sig"] != sig
            diff_reads = [i for i in range(len(res)) if neq[seg_off[i]:seg_off[i + 1]].any()]
        zdiff = int(np.sum(~((o[fam + "_Z"] == z) | (np.isnan(z) & np.isnan(o[fam + "_Z"])))))
        record["families"][fam].update(reference_nofma_border_differs=diff_reads if same_len else "layout differs",
                                       reference_nofma_Z_bits_differ=zdiff)
        store[fam + "_reference_unstable"] = np.array(diff_reads, dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, "g10_ties.npz"), **store)
    os.makedirs(os.path.join(ROOT, "profiles", "r03"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "r03", "tie_stability.json"), "w") as f:
        json.dump(record, f, indent=1)
    print(json.dumps(record, indent=1))


if __name__ == "__main__":
    main()
