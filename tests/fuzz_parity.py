"""TEST INFRASTRUCTURE (uses the CPU oracle; run by hand on a GPU box, not collected by pytest):

    python tests/fuzz_parity.py [seed = 20261004] [reads per pore and band = 150] > gpurun_out/fuzz.txt

3 000 random short reads (k .. 400 bases, dwell 0.5 .. 10) over the five pore types and four random band widths each
against the oracle: integer columns identical, posteriors within 1e-6, Z within 1e-9 relative, failures with the
reference's message; round 3 also train() (Z, transitions, per-k-mer weights at 1e-7 relative, weights summing to the
sample count) and align(calc=false) on the same reads. Round 2: 0 mismatches (profiles/r02/fuzz_parity_3000_reads.txt);
round 3, final build: profiles/r03/fuzz_parity_3000_reads.txt."""
import os, sys, tempfile, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamont_amd import Aligner, synth
from oracle.pyoracle import Oracle
d = tempfile.mkdtemp()
SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 20261004
PER = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rng = np.random.default_rng(SEED)
tot = bad = err = bad_train = bad_z = 0
for pore in ("rna002", "rna004", "dna_r9", "dna_r10_260bps", "dna_r10_400bps"):
    k = synth.PORES[pore][2]
    path = synth.write_model(os.path.join(d, f"{pore}.model"), k)
    _, mean, sd = synth.read_model_file(path)
    for band in (int(x) for x in rng.choice([6, 16, 50, 100, 200, 300, 400, 446], 4, replace=False)):
        reads = []
        for i in range(PER):
            nb = int(rng.integers(k, 400))
            reads += synth.make_reads(int(rng.integers(1 << 30)), 1, pore, mean, sd, nb, dwell=float(rng.choice([0.5, 2.0, 3.5, 7.0, 10.0])))
        al = Aligner(path, pore, band=band, device=0)
        orc = Oracle(path, synth.PORES[pore][0], band)
        res = al.align_batch([r.signal for r in reads], [r.sequence for r in reads], True)
        zs = al.align_batch([r.signal for r in reads], [r.sequence for r in reads], False)
        tr = al.train_batch([r.signal for r in reads], [r.sequence for r in reads])
        for i, r in enumerate(reads):
            try:
                want = orc.align(r.signal, r.sequence, True)
            except RuntimeError as e:
                err += 1
                assert res.error(i) == str(e), (pore, band, i, res.error(i), str(e))
                continue
            tot += 1
            got = res.read(i)
            ok = res.status[i] == 0 and np.array_equal(got["sequence_positions"], want["sequence_positions"]) and np.array_equal(got["signal_positions"], want["signal_positions"])
            if ok:
                ok = np.abs(got["probabilities"] - want["probabilities"]).max() <= 1e-6 and abs(got["Z"] - want["Z"]) <= 1e-9 * max(1.0, abs(want["Z"]))
            if not ok:
                bad += 1
                print("MISMATCH", pore, band, i, len(r.sequence), len(r.signal), flush=True)
            if not (zs.status[i] == 0 and abs(zs.Z[i] - want["Z"]) <= 1e-9 * max(1.0, abs(want["Z"]))):
                bad_z += 1
                print("Z-ONLY MISMATCH", pore, band, i, flush=True)
            wt = orc.train(r.signal, r.sequence, dense=False)
            code, m, _ = tr.sparse(i)
            a = int(tr.em_offsets[i])
            touched = np.nonzero(wt["weight"] > 0)[0]
            okt = tr.status[i] == 0 and abs(tr.Z[i] - wt["Z"]) <= 1e-9 * max(1.0, abs(wt["Z"])) and np.array_equal(code, touched)
            if okt:
                gw = tr.em_weight[a:a + len(code)]
                okt = np.allclose(gw, wt["weight"][touched], rtol=1e-7, atol=1e-12) and abs(gw.sum() - len(r.signal)) <= 1e-9 * len(r.signal)
                okt = okt and abs(tr.transitions[3 * i] - wt["m1"]) <= 1e-8 and abs(tr.transitions[3 * i + 2] - wt["e2"]) <= 1e-8
            if not okt:
                bad_train += 1
                print("TRAIN MISMATCH", pore, band, i, len(r.sequence), len(r.signal), flush=True)
        print(pore, band, "done", tot, bad, bad_z, bad_train, err, flush=True)
        al.close()
print("TOTAL reads compared", tot, "align mismatching", bad, "Z-only mismatching", bad_z, "train mismatching", bad_train, "expected errors reproduced", err)
