#!/usr/bin/env python
"""GPU box: EVERY read of fixture G12's fourteen families (25 300 reads; `pytest -m gpu` runs 48 per family) on the device
in the default configuration -- a handle as created, strict mode `ties` -- against the compiled reference's borders and Z
(tests/golden/_full/g12_all.npz, written by tests/golden/make_g12_full.py in the authoring container; not committed).

    python tests/tie_device_full.py [--out profiles/r04/tie_device_all_g12.json]

Per family: reads whose segment borders differ from the reference's (must be none), reads the rule flags
(dyn_tie_rows != 0, = the launch's reads_strict), flagged reads whose Z differs in any bit (must be none), unflagged reads
with a different Z (allowed: the plain arithmetic is not bit-exact; informational)."""
import argparse, json, os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dynamont_amd import Aligner, synth  # noqa: E402
import tie_parity  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--out", default=None)
ap.add_argument("--batch", type=int, default=1000)
ap.add_argument("--strict", default=None, help="off | ties | all: set the handle's strict mode instead of leaving the default")
ap.add_argument("--families", default=None, help="comma-separated subset")
args = ap.parse_args()
g = np.load(os.path.join(ROOT, "tests", "golden", "_full", "g12_all.npz"))
d = tempfile.mkdtemp(prefix="g12dev_")
mpaths = tie_parity.g12_model_paths(d)
record, t0 = {"families": {}}, time.time()
for fam, (pore, mkey, gen) in {**tie_parity.G12_FAMILIES, **tie_parity.EXTRA_FAMILIES}.items():
    if args.families and fam not in args.families.split(","):
        continue
    _, mean, sd = synth.read_model_file(mpaths[mkey])
    reads = gen(mean, sd)
    assert np.array_equal(g[fam + "_S"], [len(r.signal) for r in reads]), fam   # the regenerated inputs are the generator's
    seg_off, sigpos, seqpos, Z = g[fam + "_seg_off"], g[fam + "_sigpos"], g[fam + "_seqpos"], g[fam + "_Z"]
    al = Aligner(mpaths[mkey], pore, band=400, device=0)
    if args.strict:
        al.set_strict(args.strict)
    border_dev, z_dev_flagged, z_dev_plain, flagged, failed, n_strict = [], [], 0, 0, [], 0
    for lo in range(0, len(reads), args.batch):
        part = reads[lo:lo + args.batch]
        with al.batch([r.signal for r in part], [r.sequence for r in part]) as b:
            b.align(True)
            res = b.fetch()
            n_strict += int(b.timing()["reads_strict"])
        _, _, kms = al.validate([len(r.signal) for r in part], [r.sequence for r in part])
        for j, r in enumerate(part):
            i = lo + j
            if np.isnan(Z[i]):      # the reference refused the read
                continue
            if res.status[j] != 0:
                failed.append(i)
                continue
            got = res.read(j)
            a, e = int(seg_off[i]), int(seg_off[i + 1])
            if not (np.array_equal(got["signal_positions"], sigpos[a:e]) and np.array_equal(got["sequence_positions"], seqpos[a:e])):
                border_dev.append(i)
            rows = al.tie_rows(kms[j], len(r.signal))
            flagged += rows != 0
            if got["Z"] != Z[i]:
                if rows:
                    z_dev_flagged.append(i)
                else:
                    z_dev_plain += 1
    al.close()
    rec = dict(pore=pore, model=mkey, reads=len(reads), reference_refused=int(np.isnan(Z).sum()), failed_on_device=failed,
               borders_differ=border_dev, flagged_by_the_rule=int(flagged), launch_reads_strict=n_strict,
               flagged_reads_with_a_different_Z=z_dev_flagged, unflagged_reads_with_a_different_Z=int(z_dev_plain))
    record["families"][fam] = rec
    print(fam, rec, flush=True)
tot = list(record["families"].values())
record["total"] = dict(reads=sum(r["reads"] for r in tot), borders_differ=sum(len(r["borders_differ"]) for r in tot),
                       failed_on_device=sum(len(r["failed_on_device"]) for r in tot), flagged=sum(r["flagged_by_the_rule"] for r in tot),
                       flagged_reads_with_a_different_Z=sum(len(r["flagged_reads_with_a_different_Z"]) for r in tot),
                       unflagged_reads_with_a_different_Z=sum(r["unflagged_reads_with_a_different_Z"] for r in tot),
                       wall_s=round(time.time() - t0, 1))
record["strict_mode"] = args.strict or "default (ties)"
record["how"] = "python tests/tie_device_full.py: handle as created (strict mode ties) unless --strict says otherwise, batches of %d reads through dyn_batch_create/align/fetch" % args.batch
print("TOTAL", record["total"], flush=True)
if args.out:
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump(record, open(args.out, "w"), indent=1)
