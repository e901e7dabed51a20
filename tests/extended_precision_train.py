#!/usr/bin/env python
"""TEST INFRASTRUCTURE (CPU): train()'s per-k-mer weights in 80-bit extended precision, as an adjudicator.

The oracle (= the reference's arithmetic, NT_aligner_api.cpp:110-207 and :462-561) works in log space: for a 20 k-sample
read its forward/backward values are ~ -4e4, whose fp64 spacing is 7e-12, so every one of the T additions along a path
rounds the PROBABILITY by ~7e-12 relative -- the posteriors it sums carry 1e-9 .. 1e-8 of rounding noise. The product's
forward sweep is the posterior chain (one exponential of a difference formed in registers per cell; mass conserved by
construction). When the two disagree at 5e-9 this restatement, 2 000 times finer than either, says which one moved.

    python tests/extended_precision_train.py            # cfg5 reads 0 and 100: oracle vs extended precision
    python tests/extended_precision_train.py --gpu FILE  # + the product's weights saved by --save on the GPU box

Same recursions as oracle/nt_oracle.c forward()/backward()/nto_train(), vectorised per lattice row over the band.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LD = np.longdouble
NEG = LD(-np.inf)


def train_weights_extended(signal, kmers, mean, stdev, m1, e1, e2, bw=200):
    """(Z, weight per lattice column 1..N-1) in long double. kmers: codes of columns 1..N-1; m1/e1/e2: probabilities."""
    sig = np.asarray(signal, dtype=LD)
    T, N = len(sig) + 1, len(kmers) + 1
    mu = np.asarray(mean, dtype=LD)[kmers]
    sd = np.asarray(stdev, dtype=LD)[kmers]
    lognorm = -np.log(sd) - LD(0.5) * np.log(LD(2) * LD(np.pi))
    # (pi to double precision, like the reference's M_PI: the constant cancels in every posterior anyway)
    lm1, le1, le2 = np.log(LD(m1)), np.log(LD(e1)), np.log(LD(e2))
    ratio = np.float64(N) / np.float64(T)
    mid = (np.arange(T, dtype=np.float64) * ratio).astype(np.int64)
    n_start = np.maximum(mid - bw, 0)
    n_end = np.minimum(mid + bw + 1, N)

    def score(t_sig, lo, hi):  # log pdf of sample t_sig for columns lo..hi-1 (column n scores k-mer n-1; lo >= 1)
        z = (sig[t_sig] - mu[lo - 1:hi - 1]) / sd[lo - 1:hi - 1]
        return LD(-0.5) * z * z + lognorm[lo - 1:hi - 1]

    # backward, stored per row over [n_start, n_end)
    bM = [None] * T
    bE = [None] * T
    Mn = np.full(N + 1, NEG, dtype=LD)  # row t+1, full width (+1 so that n+1 may be read)
    En = np.full(N + 1, NEG, dtype=LD)
    En[N - 1] = 0
    bM[T - 1] = Mn[n_start[T - 1]:n_end[T - 1]].copy()
    bE[T - 1] = En[n_start[T - 1]:n_end[T - 1]].copy()
    with np.errstate(invalid="ignore"):
        for t in range(T - 2, -1, -1):
            lo, hi = int(n_start[t]), int(n_end[t])
            Mc = np.full(N + 1, NEG, dtype=LD)
            Ec = np.full(N + 1, NEG, dtype=LD)
            ext = np.full(hi - lo, NEG, dtype=LD)
            # n + 1 < N: M(t+1, n+1) + score(sig[t], kmers[n]) + m1  -- column n+1's k-mer
            hi1 = min(hi, N - 1)
            if hi1 > lo:
                ext[:hi1 - lo] = Mn[lo + 1:hi1 + 1] + score(t, lo + 1, hi1 + 1) + lm1
            lo1 = max(lo, 1)
            if hi > lo1:
                sc = score(t, lo1, hi)
                Mc[lo1:hi] = En[lo1:hi] + sc
                ext[lo1 - lo:] = np.logaddexp(ext[lo1 - lo:], En[lo1:hi] + sc + le2)
            Ec[lo:hi] = ext
            bM[t], bE[t] = Mc[lo:hi].copy(), Ec[lo:hi].copy()
            Mn, En = Mc, Ec
        Zb = bE[0][0 - int(n_start[0])]
        # forward with the statistics on the fly
        w = np.zeros(N, dtype=LD)
        Mp = np.full(N + 1, NEG, dtype=LD)
        Ep = np.full(N + 1, NEG, dtype=LD)
        Ep[0] = 0
        for t in range(1, T):
            lo, hi = max(int(n_start[t]), 1), int(n_end[t])
            Mc = np.full(N + 1, NEG, dtype=LD)
            Ec = np.full(N + 1, NEG, dtype=LD)
            sc = score(t - 1, lo, hi)
            Mc[lo:hi] = Ep[lo - 1:hi - 1] + sc + lm1
            Ec[lo:hi] = np.logaddexp(Mp[lo:hi] + sc + le1, Ep[lo:hi] + sc + le2)
            o = lo - int(n_start[t])
            w[lo:hi] += np.exp(Mc[lo:hi] + bM[t][o:o + hi - lo] - Zb) + np.exp(Ec[lo:hi] + bE[t][o:o + hi - lo] - Zb)
            Mp, Ep = Mc, Ec
    return Zb, w[1:]


def chain_weights_quantised(signal, kmers, mean, stdev, m1, e1, e2, store_dtype, bw=200):
    """The product's forward sweep (posterior chain) emulated on the CPU with the backward sweep's ln P(stay) rounded to
    ``store_dtype`` (np.float64 = what the product stores, np.float32 = the half-size lattice DESIGN.md section 10
    mentions): backward values in long double, the chain itself in fp64. Returns the weight per lattice column."""
    sig = np.asarray(signal, dtype=LD)
    T, N = len(sig) + 1, len(kmers) + 1
    mu = np.asarray(mean, dtype=LD)[kmers]
    sd = np.asarray(stdev, dtype=LD)[kmers]
    lognorm = -np.log(sd) - LD(0.5) * np.log(LD(2) * LD(np.pi))
    lm1, le2 = np.log(LD(m1)), np.log(LD(e2))
    ratio = np.float64(N) / np.float64(T)
    mid = (np.arange(T, dtype=np.float64) * ratio).astype(np.int64)
    n_start = np.maximum(mid - bw, 0)
    n_end = np.minimum(mid + bw + 1, N)

    def score(t_sig, lo, hi):
        z = (sig[t_sig] - mu[lo - 1:hi - 1]) / sd[lo - 1:hi - 1]
        return LD(-0.5) * z * z + lognorm[lo - 1:hi - 1]

    stay = [None] * T  # ln P(stay | E(t, n)) over the full width, as stored
    Mn = np.full(N + 1, NEG, dtype=LD)
    En = np.full(N + 1, NEG, dtype=LD)
    En[N - 1] = 0
    with np.errstate(invalid="ignore", over="ignore"):
        for t in range(T - 2, -1, -1):
            lo, hi = int(n_start[t]), int(n_end[t])
            Mc = np.full(N + 1, NEG, dtype=LD)
            Ec = np.full(N + 1, NEG, dtype=LD)
            ext = np.full(hi - lo, NEG, dtype=LD)
            C = np.full(hi - lo, NEG, dtype=LD)
            hi1 = min(hi, N - 1)
            if hi1 > lo:
                ext[:hi1 - lo] = Mn[lo + 1:hi1 + 1] + score(t, lo + 1, hi1 + 1) + lm1
            lo1 = max(lo, 1)
            if hi > lo1:
                sc = score(t, lo1, hi)
                Mc[lo1:hi] = En[lo1:hi] + sc
                C[lo1 - lo:] = En[lo1:hi] + sc + le2
                ext[lo1 - lo:] = np.logaddexp(ext[lo1 - lo:], C[lo1 - lo:])
            Ec[lo:hi] = ext
            row = np.full(N + 1, -np.inf)
            d = (C - ext).astype(np.float64)
            d[np.isnan(d)] = -np.inf
            row[lo:hi] = d.astype(store_dtype).astype(np.float64)
            stay[t] = row
            Mn, En = Mc, Ec
        gE = np.zeros(N + 1)
        gM = np.zeros(N + 1)
        gE[0] = 1.0
        w = np.zeros(N + 1)
        for t in range(1, T):
            sp = np.minimum(np.exp(stay[t - 1]), 1.0)
            st = gE * sp
            mv = gE - st
            gE = gM + st
            gM = np.zeros(N + 1)
            gM[1:] = mv[:-1]
            w += gE + gM
    return w[1:N]


def per_kmer(w_cols, kmers, K):
    out = np.zeros(K, dtype=LD)
    np.add.at(out, kmers, w_cols)
    return out


def main():
    from dynamont_amd import synth
    from oracle.pyoracle import Oracle
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", default="0,100")
    ap.add_argument("--gpu", help="npz with w<i> = the product's per-k-mer weights of read i (written by --save)")
    ap.add_argument("--save", help="ON THE GPU BOX: run the product's train() on the reads and save their weights here")
    ap.add_argument("--chain-storage", action="store_true",
                    help="CPU emulation of the posterior chain with ln P(stay) stored as float64 and as float32")
    args = ap.parse_args()
    import tempfile
    d = tempfile.mkdtemp()
    model = synth.write_model(os.path.join(d, "syn9.model"), 9, seed=7, stdev=0.15)
    _, mean, sd = synth.read_model_file(model)
    cfg = synth.CONFIGS["cfg5"]
    picks = [int(x) for x in args.reads.split(",")]
    reads = synth.make_reads(cfg["seed"], max(picks) + 1, cfg["pore"], mean, sd, cfg["n_bases"])
    if args.save:
        from dynamont_amd import Aligner
        al = Aligner(model, cfg["pore"], device=0)
        res = al.train_batch([reads[i].signal for i in picks], [reads[i].sequence for i in picks])
        out = {}
        for j, i in enumerate(picks):
            code, _, _ = res.sparse(j)
            a = int(res.em_offsets[j])
            w = np.zeros(al.num_kmers)
            w[code] = res.em_weight[a:a + len(code)]
            out["w%d" % i] = w
        np.savez_compressed(args.save, **out)
        return
    orc = Oracle(model, 1)
    mean, sd = orc.table()  # indexed by k-mer code
    gpu = np.load(args.gpu) if args.gpu else None
    for i in picks:
        r = reads[i]
        km = orc.kmers(r.sequence)
        Z, wc = train_weights_extended(r.signal, km, mean, sd, 0.031111753637096777, 1.0, 0.9688882463622581)
        truth = per_kmer(wc, km, len(mean))
        t = orc.train(r.signal, r.sequence, dense=False)
        nz = truth > 0
        rel = lambda a: float(np.max(np.abs(a[nz].astype(LD) - truth[nz]) / truth[nz]))
        line = "read %d (S %d): Z %.12f; oracle Z off by %.3g; oracle weights off by %.3g relative" % (
            i, len(r.signal), float(Z), float(t["Z"] - Z), rel(t["weight"]))
        if args.chain_storage:
            for dt in (np.float64, np.float32):
                wq = per_kmer(chain_weights_quantised(r.signal, km, mean, sd, 0.031111753637096777, 1.0, 0.9688882463622581, dt), km, len(mean))
                line += "; chain with %s rows off by %.3g" % (np.dtype(dt).name, rel(wq))
        if gpu is not None and ("w%d" % i) in gpu:
            line += "; product off by %.3g, product vs oracle %.3g" % (
                rel(gpu["w%d" % i]), float(np.max(np.abs(gpu["w%d" % i][nz] - t["weight"][nz]) / t["weight"][nz])))
        print(line, flush=True)


if __name__ == "__main__":
    main()
