"""TEST INFRASTRUCTURE (uses the CPU oracle; run by hand on a GPU box, not collected by pytest):

    python tests/fuzz_wide_band.py > gpurun_out/fuzz_wide.txt

Random reads under bands the register sweeps cannot hold (round 6, wide_band.hip): five pore types x two random bands out of
{448 .. 3000}, 40 reads each of k .. 1 600 bases with dwells 2 .. 10 -- so that a batch MIXES reads the tuned sweeps compute at
that band (half band <= 223 because the read is short) with reads the generic kernel computes --, a homopolymer stretch in every
fourth read (structural ties), one read with an N, one with a NaN sample. Against the oracle at the same band: integer columns
identical; Z of the wide reads BIT-identical (the generic kernel runs the reference's own arithmetic), of the others within 1e-9
relative; posteriors within 1e-6; the Z-only call; train() (Z, transitions, per-k-mer weights at 1e-7 relative); failures with
the reference's message."""
import os, sys, tempfile, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamont_amd import Aligner, synth
from oracle.pyoracle import Oracle
d = tempfile.mkdtemp()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 20261005)  # python tests/fuzz_wide_band.py [seed]
tot = wide = bad = bad_zbits = bad_z = bad_train = err = 0
for pore in ("rna002", "rna004", "dna_r9", "dna_r10_260bps", "dna_r10_400bps"):
    k = synth.PORES[pore][2]
    path = synth.write_model(os.path.join(d, f"{pore}.model"), k)
    _, mean, sd = synth.read_model_file(path)
    for band in (int(x) for x in rng.choice([448, 500, 640, 900, 1300, 2000, 3000], 2, replace=False)):
        reads = []
        for i in range(40):
            nb = int(rng.integers(k + 2, 1600))
            r = synth.make_reads(int(rng.integers(1 << 30)), 1, pore, mean, sd, nb, dwell=float(rng.choice([2.0, 3.5, 7.0, 10.0])))[0]
            if i % 4 == 1 and nb > 60:   # a homopolymer of 2k bases somewhere: tied column pairs
                p = int(rng.integers(10, nb - 2 * k - 10))
                r = synth.SynthRead(r.signal, r.sequence[:p] + "A" * (2 * k) + r.sequence[p + 2 * k:])
            if i == 7:
                r = synth.SynthRead(r.signal, r.sequence[:5] + "N" + r.sequence[6:])
            if i == 11:
                s = r.signal.copy(); s[len(s) // 2] = np.nan
                r = synth.SynthRead(s, r.sequence)
            reads.append(r)
        al = Aligner(path, pore, band=band, device=0)
        orc = Oracle(path, synth.PORES[pore][0], band)
        sigs, seqs = [r.signal for r in reads], [r.sequence for r in reads]
        res = al.align_batch(sigs, seqs, True)
        zs = al.align_batch(sigs, seqs, False)
        tr = al.train_batch(sigs, seqs)
        for i, r in enumerate(reads):
            n_cols = len(r.sequence) - k + 2
            is_wide = min(band // 2, n_cols // 2) > 223
            if i == 11:   # a NaN sample: the reference's |Zf - Zb| test lets NaN through (NT_aligner_api.cpp:288) and returns garbage;
                err += 1  # this build fails the read (documented deviation) -- whichever kernel takes it
                assert res.status[i] != 0 and zs.status[i] != 0 and tr.status[i] != 0, (pore, band, i)
                continue
            try:
                want = orc.align(r.signal, r.sequence, True)
            except RuntimeError as e:
                err += 1
                assert res.error(i) == str(e), (pore, band, i, res.error(i), str(e))
                continue
            tot += 1
            wide += is_wide
            if res.status[i] != 0:
                bad += 1
                print("FAILED WHERE THE ORACLE SUCCEEDS", pore, band, i, len(r.sequence), len(r.signal), is_wide, res.error(i), "Z want", want["Z"], flush=True)
                continue
            got = res.read(i)
            ok = res.status[i] == 0 and np.array_equal(got["sequence_positions"], want["sequence_positions"]) and np.array_equal(got["signal_positions"], want["signal_positions"])
            if ok:
                ok = np.abs(got["probabilities"] - want["probabilities"]).max() <= 1e-6 and abs(got["Z"] - want["Z"]) <= 1e-9 * max(1.0, abs(want["Z"]))
            if not ok:
                bad += 1
                print("MISMATCH", pore, band, i, len(r.sequence), len(r.signal), is_wide, flush=True)
            if is_wide and got["Z"] != want["Z"]:
                bad_zbits += 1
                print("Z BITS DIFFER on a wide read", pore, band, i, got["Z"], want["Z"], flush=True)
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
                # (the weights are the REFERENCE's to ~1e-15 here -- its own log-domain noise included: they sum to the sample count to ~1e-9)
                okt = np.allclose(gw, wt["weight"][touched], rtol=1e-7, atol=1e-12) and abs(gw.sum() - len(r.signal)) <= 1e-8 * len(r.signal)
                okt = okt and abs(tr.transitions[3 * i] - wt["m1"]) <= 1e-8 and abs(tr.transitions[3 * i + 2] - wt["e2"]) <= 1e-8
            if not okt:
                bad_train += 1
                gw = tr.em_weight[a:a + len(code)]
                same_codes = np.array_equal(code, touched)
                rel = (np.abs(gw - wt["weight"][touched]) / np.maximum(wt["weight"][touched], 1e-300)).max() if same_codes and len(code) else -1
                print("TRAIN MISMATCH", pore, band, i, len(r.sequence), len(r.signal), is_wide, "status", tr.status[i], "Z", tr.Z[i], wt["Z"], "codes equal", same_codes,
                      len(code), len(touched), "max rel weight err", rel, "sum", float(gw.sum()), "m1", tr.transitions[3 * i], wt["m1"], "e2", tr.transitions[3 * i + 2], wt["e2"], flush=True)
                if not same_codes:
                    only_ours, only_theirs = np.setdiff1d(code, touched), np.setdiff1d(touched, code)
                    print("   codes only ours", only_ours[:8], "only the oracle's", only_theirs[:8], "their weights", wt["weight"][only_theirs[:8]], flush=True)
        print(pore, band, "done: compared", tot, "of which wide", wide, "| mismatches: align", bad, "Z bits", bad_zbits, "Z-only", bad_z, "train", bad_train, "| errors reproduced", err, flush=True)
        al.close()
print("TOTAL reads compared", tot, "of which by the generic kernel", wide, "| align mismatching", bad, "| wide reads whose Z bits differ", bad_zbits,
      "| Z-only mismatching", bad_z, "| train mismatching", bad_train, "| expected errors reproduced", err)
