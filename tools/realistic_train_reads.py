#!/usr/bin/env python
"""GPU box: train() on reads less ideal than BASELINE's synthetic ones -- heavy-tailed dwell times with occasional
stalls (the path strays from the band's diagonal) and basecalls that disagree with the signal (substitutions,
insertions, deletions). Prints, per scenario, the reads that trained, how far their weights are from summing to the
number of samples, and the throughput of the launch. (A forward-backward product in the linear domain, tried before the
posterior chain, lost 79 % of such reads at 2 % substitutions: profiles/r03/linear_domain_on_imperfect_reads.txt.)

    python tools/realistic_train_reads.py [n_reads] [n_bases]
"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from dynamont_amd import Aligner, synth  # noqa: E402


def make(rng, mean_code, sd_code, k, n_bases, heavy, p_sub, p_indel):
    digits = rng.integers(0, 4, size=n_bases)
    digits[:9] = 0
    codes = synth._seq_codes(digits, k)
    if heavy:  # gamma dwell (mean 10, shape 1.2) and one k-mer in a hundred stalls 10-40 times as long
        dw = 2 + np.floor(rng.gamma(1.2, 8.0 / 1.2, size=len(codes))).astype(np.int64)
        stall = rng.random(len(codes)) < 0.01
        dw[stall] *= rng.integers(10, 40, size=int(stall.sum()))
    else:
        dw = np.maximum(2, rng.poisson(10.0, size=len(codes)))
    c = rng.uniform(0.8, 2.0)
    idx = np.repeat(codes, dw)
    sig = mean_code[idx] + c * sd_code[idx] * rng.standard_normal(len(idx))
    # what the basecaller reports: the true bases with errors (the polyA pad stays)
    called = []
    for i, d in enumerate(digits):
        if i < 9:
            called.append(d)
            continue
        u = rng.random()
        if u < p_indel / 2:
            continue                                  # deletion
        if u < p_indel:
            called.append(int(rng.integers(0, 4)))    # insertion before the base
        called.append(int(rng.integers(0, 4)) if rng.random() < p_sub else int(d))
    return synth.SynthRead(np.ascontiguousarray(sig), "".join(synth.BASES[d] for d in called))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    d = tempfile.mkdtemp()
    model = synth.write_model(os.path.join(d, "syn9.model"), 9, seed=7, stdev=0.15)
    _, mean, sd = synth.read_model_file(model)
    mean_code, sd_code = synth.code_order_table(mean, sd, 9, True)
    al = Aligner(model, "rna004", device=0)
    for heavy in (False, True):
        for p_sub, p_indel in ((0.0, 0.0), (0.02, 0.01), (0.05, 0.03), (0.10, 0.06)):
            rng = np.random.default_rng(31337)
            reads = [make(rng, mean_code, sd_code, 9, nb, heavy, p_sub, p_indel) for _ in range(n)]
            sig, so, sq, qo = synth.pack_reads(reads)
            for rep in range(2):  # the second launch is timed (the first may grow the pool)
                t = al.train_async(sig, so, sq, qo, pooled=False)
                res = t.wait()
                tm = t.timing()
                t.close()
            ok = res.status == 0
            worst = 0.0
            for i in np.nonzero(ok)[0]:
                a, c = int(res.em_offsets[i]), int(res.em_count[i])
                worst = max(worst, abs(res.em_weight[a:a + c].sum() / len(reads[i].signal) - 1.0))
            print("dwell %-7s substitutions %4.0f %%  indels %4.0f %%: trained %4d of %d, |sum of weights / samples - 1| <= %.1e, "
                  "launch %.1f ms = %.0f Msamp/s" % ("heavy" if heavy else "poisson", 100 * p_sub, 100 * p_indel, int(ok.sum()), n,
                                                      worst, tm["ms_dp"], tm["samples"] / tm["ms_dp"] / 1e3), flush=True)


if __name__ == "__main__":
    main()
