#!/usr/bin/env python
"""Generate golden fixture G12 -- the WIDENED structural-tie evidence of round 4 -- and profiles/r04/tie_stability.json.

Runs only in the authoring container (needs /root/reference compiled into oracle/_ref by `make -C oracle ref`).

    python tests/golden/make_golden_g12.py [workers]

Question: strict mode "ties" -- a handle's default -- runs a read bit for bit iff it carries two
neighbouring columns with the same emission parameters (dyn_tie_rows), and then only up to the forward row in which the
last such pair has left the band. Is that rule enough? Fourteen families (tests/tie_parity.py, G12_FAMILIES), ~28 000
reads: internal homopolymers for four pores (3 000 reads each), ties next to but not AT the read start (k-mers j, j+1
equal for j = 1..3 / right behind the RNA pad), ties beyond lattice row 1 024, read-start ties in reads long enough for
the certified rows to be a PREFIX of the sweep, a model whose table holds distinct k-mers with identical entries, and
random reads that carry no tie at all. For every read, against the COMPILED REFERENCE's segment borders:
  * the oracle's control flow replayed with the PLAIN arithmetic (mode 1): how many reads deviate, and whether any of
    them is a read the rule does NOT flag (must be none -- such a read would be a hole in the rule);
  * the replay of strict mode "ties" row by row (mode 8: certified backward sweep, certified forward blocks, plain
    elsewhere, exactly as nt_kernels.hip switches): must deviate on no read, and Z must be bit-identical on flagged reads.
The fixture keeps, per family, the reference's borders and Z of the first 48 reads (what `pytest -m gpu` runs in the
default configuration) and the ids of the reads the plain arithmetic gets wrong.
"""
from __future__ import annotations

import json
import multiprocessing as mp
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dynamont_amd import synth, Aligner  # noqa: E402
from oracle.pyoracle import Reference  # noqa: E402
import tie_parity  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
KEEP = 48

_W = {}


def _init(so, model, pore):
    enum = synth.PORES[pore][0]
    _W["ref"] = Reference(model, enum, 400)
    _W["rp"] = tie_parity.Replay(so, model, enum, 400, mode=1)
    _W["al"] = Aligner(model, pore, device="host")


def _one(job):
    sig, seq = job
    ref, rp, al = _W["ref"], _W["rp"], _W["al"]
    try:
        want = ref.align(sig, seq, True)
    except RuntimeError:
        return None
    _, _, kms = al.validate([len(sig)], [seq])
    rows = al.tie_rows(kms[0], len(sig))
    rp.set_mode(1)
    plain = rp.align(sig, seq, True)
    rp.set_mode(8)
    rp.set_strict_rows(rows)
    rp.counts()
    ties = rp.align(sig, seq, True)
    calls, amb = rp.counts()
    return dict(rows=rows, sig=want["signal_positions"].astype(np.uint32), seq=want["sequence_positions"].astype(np.uint32),
                Z=want["Z"], plain_ok=tie_parity.borders_equal(plain, want), ties_ok=tie_parity.borders_equal(ties, want),
                ties_z_ok=ties["Z"] == want["Z"], plain_z_ok=plain["Z"] == want["Z"], calls=calls, amb=amb,
                T=len(sig) + 1)


def main():
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    d = tempfile.mkdtemp(prefix="g12_")
    mpaths = tie_parity.g12_model_paths(d)
    so = tie_parity.build_replay(d)
    store, record = {}, {"glibc": os.confstr("CS_GNU_LIBC_VERSION"), "families": {}}
    for fam, (pore, mkey, gen) in tie_parity.G12_FAMILIES.items():
        _, mean, sd = synth.read_model_file(mpaths[mkey])
        reads = gen(mean, sd)
        with mp.get_context("fork").Pool(workers, initializer=_init, initargs=(so, mpaths[mkey], pore)) as pool:
            res = pool.map(_one, [(r.signal, r.sequence) for r in reads], chunksize=8)
        ok = [r is not None for r in res]
        good = [r for r in res if r is not None]
        flagged = [i for i, r in enumerate(res) if r is not None and r["rows"] != 0]
        prefix = [i for i in flagged if res[i]["rows"] != 0xffffffff and res[i]["rows"] < res[i]["T"] - 64]
        plain_dev = [i for i, r in enumerate(res) if r is not None and not r["plain_ok"]]
        ties_dev = [i for i, r in enumerate(res) if r is not None and not r["ties_ok"]]
        hole = [i for i in plain_dev if res[i]["rows"] == 0]
        z_bad = [i for i in flagged if not res[i]["ties_z_ok"]]
        rec = dict(pore=pore, model=mkey, reads=len(reads), reads_ok=int(sum(ok)), segments=int(sum(len(r["sig"]) for r in good)),
                   flagged_by_the_rule=len(flagged), flagged_with_a_certified_prefix_only=len(prefix),
                   plain_arithmetic_deviates=plain_dev, plain_deviates_on_unflagged_reads=hole,
                   plain_Z_bits_differ=int(sum(not r["plain_z_ok"] for r in good)),
                   mode_ties_deviates=ties_dev, mode_ties_Z_bits_differ_on_flagged_reads=z_bad,
                   certified_log_plus_calls=int(sum(r["calls"] for r in good)), ambiguous=int(sum(r["amb"] for r in good)))
        record["families"][fam] = rec
        print(fam, {k: v for k, v in rec.items()}, flush=True)
        keep = list(range(min(KEEP, len(reads)))) + [i for i in plain_dev if i >= KEEP]
        seg_off = np.zeros(len(keep) + 1, dtype=np.int64)
        for j, i in enumerate(keep):
            seg_off[j + 1] = seg_off[j] + (0 if res[i] is None else len(res[i]["sig"]))
        store[fam + "_ids"] = np.array(keep, dtype=np.int32)
        store[fam + "_S"] = np.array([len(reads[i].signal) for i in keep], dtype=np.int32)  # guards the regenerated inputs
        store[fam + "_n_reads"] = np.array([len(reads)], dtype=np.int32)
        store[fam + "_seg_off"] = seg_off
        store[fam + "_sigpos"] = np.concatenate([np.zeros(0, np.uint32)] + [res[i]["sig"] for i in keep if res[i] is not None])
        store[fam + "_seqpos"] = np.concatenate([np.zeros(0, np.uint32)] + [res[i]["seq"] for i in keep if res[i] is not None])
        store[fam + "_Z"] = np.array([np.nan if res[i] is None else res[i]["Z"] for i in keep])
        store[fam + "_rows"] = np.array([0 if res[i] is None else res[i]["rows"] for i in keep], dtype=np.uint32)
        store[fam + "_plain_deviates"] = np.array(plain_dev, dtype=np.int32)
    tot = record["families"].values()
    record["total"] = dict(reads=sum(r["reads"] for r in tot), flagged=sum(r["flagged_by_the_rule"] for r in tot),
                           plain_arithmetic_deviates=sum(len(r["plain_arithmetic_deviates"]) for r in tot),
                           plain_deviates_on_unflagged_reads=sum(len(r["plain_deviates_on_unflagged_reads"]) for r in tot),
                           mode_ties_deviates=sum(len(r["mode_ties_deviates"]) for r in tot),
                           mode_ties_Z_bits_differ_on_flagged_reads=sum(len(r["mode_ties_Z_bits_differ_on_flagged_reads"]) for r in tot))
    print("TOTAL", record["total"], flush=True)
    np.savez_compressed(os.path.join(OUT, "g12_ties_wide.npz"), **store)
    os.makedirs(os.path.join(ROOT, "profiles", "r04"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "r04", "tie_stability.json"), "w") as f:
        json.dump(record, f, indent=1)


if __name__ == "__main__":
    main()
