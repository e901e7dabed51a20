#!/usr/bin/env python
"""``dynamont-train`` counterpart (reference: src/dynamont/segmentation/train.py).

Batch Baum-Welch driver: per batch of reads one ``train_batch`` on the GPU, then a parameter
update, a model file ``trained_{epoch}_{batch}.model``, a ``params.csv`` row and a re-evaluation
of Z with the new model (train.py:178-249).

Aggregation (``--aggregate``):
  window-mean  the reference's estimator: every parameter becomes the mean of a 100-deep sliding
               window of PER-READ estimates (ManagedList, train.py:19-46,106,213,220);
  pooled       one M-step from the batch-pooled sufficient statistics (w, s1, s2) -- the quantity
               BASELINE.json config 5 all-reduces across GPUs (dynamont_amd.parallel).

Deliberate differences from the reference (not silently "fixed"):
  * k-mer <-> parameter pairing is by k-mer code; the reference zips file order with code order
    (utils.py:171-175), which mislabels RNA models;
  * as in the reference, updated TRANSITIONS are logged but never reach the native aligner
    (no setter exists, aligner_bindings.cpp:191-216): every read trains with the pore defaults.
"""
from __future__ import annotations

import sys
from argparse import ArgumentDefaultsHelpFormatter, ArgumentParser, Namespace
from collections import deque
from datetime import datetime
from os import makedirs
from os.path import basename, dirname, exists, join

import numpy as np

from dynamont_amd import Aligner, __version__
from dynamont_amd.pod5_io import get_signal, iter_basecalls, open_pod5
from dynamont_amd.segmentation.utils import (cnt_nts_ratios, get_model, hampel, kmer_of_code, read_kmer_model,
                                             write_kmer_model, write_kmer_model_arrays)

WINDOW = 100


class ManagedList:
    """Bounded list of recent values (train.py:19-46)."""

    def __init__(self, values, max_size=WINDOW):
        self.values = deque(values, maxlen=max_size)

    def add(self, value):
        self.values.append(value)

    def get_list(self):
        return list(self.values)

    def __repr__(self):
        return f"ManagedList({list(self.values)})"

    def mean(self):
        return None if len(self.values) == 0 else np.mean(self.values)

    def median(self):
        return None if len(self.values) == 0 else np.median(self.values)


class ManagedTable:
    """ManagedList for every k-mer at once: a ring of dense (mean, stdev) vectors. ``add`` appends
    one per-read estimate for EVERY k-mer (the reference appends the full dense model of each
    read, train.py:202-205); ``mean`` is the column mean over the filled part of the window."""

    def __init__(self, mean0: np.ndarray, sd0: np.ndarray, max_size=WINDOW):
        self.buf = np.empty((max_size, 2, mean0.size))
        self.buf[0, 0] = mean0
        self.buf[0, 1] = sd0
        self.count = 1
        self.head = 1
        self.max_size = max_size

    def add(self, mean: np.ndarray, sd: np.ndarray):
        self.buf[self.head % self.max_size, 0] = mean
        self.buf[self.head % self.max_size, 1] = sd
        self.head += 1
        self.count = min(self.count + 1, self.max_size)

    def mean(self):
        m = self.buf[:self.count].mean(axis=0)
        return m[0], m[1]


def parse(argv=None) -> Namespace:
    """Flags of train.py:48-66 plus the build-only ones."""
    p = ArgumentParser(formatter_class=ArgumentDefaultsHelpFormatter, prog="dynamont-train")
    p.add_argument("-r", "--raw", type=str, required=True, metavar="PATH", help="Path to raw ONT data. [POD5|FAST5]")
    p.add_argument("-b", "--basecalls", type=str, required=True, metavar="BAM", help="Basecalls of ONT training data as .bam file")
    p.add_argument("-o", "--outdir", type=str, required=True, metavar="PATH", help="Outpath to write files")
    p.add_argument("-p", "--pore", type=str, required=True, choices=["rna002", "rna004", "dna_r10_260bps", "dna_r10_400bps"], help="Pore generation used to sequence the data")
    p.add_argument("--model_path", type=str, help="Which initial kmer models to use for training")
    p.add_argument("--batch_size", type=int, default=24, help="Number of reads to train before updating")
    p.add_argument("--max_batches", type=int, default=None, help="Numbers of batches to train each epoch")
    p.add_argument("-e", "--epochs", type=int, default=1, help="Number of training epochs")
    p.add_argument("-q", "--qscore", type=float, default=10.0, help="Minimal allowed quality score")
    p.add_argument("--version", action="version", version=f"%(prog)s {__version__}")
    p.add_argument("--device", type=int, default=0, help="HIP device ordinal")
    p.add_argument("--aggregate", choices=["window-mean", "pooled"], default="window-mean",
                   help="parameter update: the reference's sliding-window mean of per-read estimates, or a pooled M-step")
    p.add_argument("--no-timestamp", action="store_true", help="write into OUTDIR itself (the reference appends a timestamp)")
    p.add_argument("--host-preprocess", action="store_true", help="normalise + Hampel-filter with NumPy on the host instead of on the GPU (same values)")
    p.add_argument("--reference-zcheck", action="store_true",
                   help="also refuse the reads the reference's |Zf - Zb| / size > 1e-8 rule refuses (NT_aligner_api.cpp:619-625): "
                        "one more Z-only forward sweep per read; default: the posterior chain's own mass check")
    return p.parse_args(argv)


def _code_order(model: dict, k: int, rna: bool):
    n = 4 ** k
    names = [kmer_of_code(c, k, rna) for c in range(n)]
    mean = np.array([model[s][0] for s in names])
    sd = np.array([model[s][1] for s in names])
    return names, mean, sd


def read_items(data_path, basecalls, pore, minq, raw: bool = False):
    """Read filters + preprocessing of train.py:125-176, one item per accepted read:
    (signal, sequence, readid). With ``raw=True`` the arithmetic is left to the device and the item
    is ((raw float32 slice, shift, scale), sequence, readid) for ``Aligner.batch_raw(..., window=7,
    n_sigmas=5.0, f32=True)``."""
    old_file, r5 = None, None
    for rec in iter_basecalls(basecalls):
        if minq and rec.get_tag("qs") < minq:
            yield "qskip"
            continue
        seq = rec.query_sequence
        counts = cnt_nts_ratios(seq)
        if any(counts[b] >= 0.6 for b in counts):  # homopolymer-like artefacts (train.py:136-142)
            continue
        readid = rec.get_tag("pi") if rec.has_tag("pi") else rec.query_name
        sp = rec.get_tag("sp") if rec.has_tag("sp") else 0
        start, end = sp + rec.get_tag("ts"), sp + rec.get_tag("ns")
        shift, scale = rec.get_tag("sm"), rec.get_tag("sd")
        ont_file = join(data_path, rec.get_tag("fn"))
        if old_file != ont_file:
            old_file, r5 = ont_file, open_pod5(ont_file)
        try:
            signal = get_signal(r5, readid, calibrated=shift <= 400)[start:end]
        except Exception:  # noqa: BLE001
            yield "mismatch"
            continue
        signal = np.array(signal, dtype=np.float32 if signal.dtype.kind != "f" else signal.dtype, copy=True)
        if "rna" in pore:
            seq = seq[::-1]
            if not seq.startswith("AAAAAAAAA"):
                seq = "AAAAAAAAA" + seq
        if raw:
            yield ((signal, float(shift), float(scale)), seq, readid)
            continue
        signal -= shift   # float32 arithmetic when the reader returns float32, as in train.py:168-169
        signal /= scale
        hampel(signal, 7, 5.0)
        yield (signal, seq, readid)


def _train_items(al: Aligner, items, pooled: bool, raw: bool):
    if not raw:
        return al.train_batch([x[0] for x in items], [x[1] for x in items], pooled=pooled)
    f32 = items[0][0][0].dtype == np.float32
    with al.batch_raw([x[0][0] for x in items], [x[1] for x in items], [x[0][1] for x in items],
                      [x[0][2] for x in items], window=7, n_sigmas=5.0, f32=f32) as b:
        b.train()
        return b.fetch_train(pooled)


def _z_items(al: Aligner, items, raw: bool):
    if not raw:
        return al.align_batch([x[0] for x in items], [x[1] for x in items], calc_probabilities=False)
    f32 = items[0][0][0].dtype == np.float32
    with al.batch_raw([x[0][0] for x in items], [x[1] for x in items], [x[0][1] for x in items],
                      [x[0][2] for x in items], window=7, n_sigmas=5.0, f32=f32) as b:
        b.align(False)
        return b.fetch()


def train(data_path: str, basecalls: str, batch_size: int, epochs: int, param_file: str, mode: str, model_path: str,
          max_batches, pore: str, minq=None, device: int = 0, aggregate: str = "window-mean", comm=None,
          host_preprocess: bool = False, reference_zcheck: bool = False) -> None:
    """Counterpart of train.py:68-253. ``comm`` (optional, dynamont_amd.parallel.Comm): multi-GPU
    run -- accepted reads are dealt round-robin to the ranks (each rank fills ``batch_size`` reads of
    its own per batch, so all ranks reach a batch boundary on the same read), the pooled sufficient
    statistics, transition counts and Z changes are summed over ranks, every rank computes the same
    model, and rank 0 owns params.csv and the model files in ``outdir``."""
    if mode != "basic":
        print(f"Mode {mode} not implemented", file=sys.stderr)
        sys.exit(1)
    rank, world = (comm.rank, comm.world) if comm is not None else (0, 1)
    if world > 1 and aggregate != "pooled":
        raise ValueError("multi-GPU training needs --aggregate pooled (per-read windows are rank-local)")
    model = read_kmer_model(model_path)
    outdir = dirname(param_file)
    if rank != 0:  # private copies of the per-batch model files; params.csv is rank 0's
        outdir = join(outdir, f".rank{rank}")
        makedirs(outdir, exist_ok=True)
        param_file = join(outdir, "params.csv")
    trained_model = join(outdir, "trained_0_0.model")
    write_kmer_model(trained_model, model)
    transition_params = {"e1": 1.0, "m1": 0.03, "e2": 0.97}  # train.py:76-82 (logged only)
    probe = Aligner(trained_model, pore, device="host")
    k, rna, K = probe.kmer_size, probe.rna, probe.num_kmers
    names, mean0, sd0 = _code_order(model, k, rna)
    # the model file keeps the row order of the file it started from: row -> k-mer code, once
    code_of_name = {name: c for c, name in enumerate(names)}
    code_of_row = np.fromiter((code_of_name[name] for name in model), dtype=np.int64, count=len(model))
    file_kmers = "".join(model).encode()
    table = ManagedTable(mean0, sd0)
    trans = {p: ManagedList([v]) for p, v in transition_params.items()}
    any_seen = False
    al = None
    i = qskips = mismatches = 0
    with open(param_file, "w") as pw:
        pw.write("epoch,batch,read," + "".join(p + "," for p in transition_params) + "Zchange\n")
        for e in range(epochs):
            items, cbatch, accepted = [], 0, 0
            for it in read_items(data_path, basecalls, pore, minq, raw=not host_preprocess):
                if it == "qskip":
                    qskips += 1
                    continue
                if it == "mismatch":
                    mismatches += 1
                    continue
                accepted += 1
                if (accepted - 1) % world == rank:
                    items.append(it)
                if accepted % (batch_size * world):  # a batch = batch_size reads on EVERY rank
                    continue
                print("============================", file=sys.stderr)
                print(f"{datetime.now().strftime('%Y-%m-%d_%H-%M-%S')}: Training epoch: {e}, reads: {i}, batch: {cbatch}\n{transition_params}", file=sys.stderr)
                cbatch += 1
                # train.py:179 builds an Aligner from the model file of the previous batch; the one kept here holds exactly
                # those values (set_model below; a float64 survives the file's shortest decimal representation unchanged)
                if al is None:
                    al = Aligner(trained_model, pore, mode="basic", threads=4, band=400, device=device)
                    al.set_train_zcheck(reference_zcheck)
                cur_mean, cur_sd = al.model_table()
                res = _train_items(al, items, aggregate == "pooled", raw=not host_preprocess)
                preZ = {}
                for j, (_, _, readid) in enumerate(items):
                    if res.status[j] != 0:
                        print(f"error: native, {res.error(j)} T: {len(items[j][0][0] if not host_preprocess else items[j][0])} N: {len(items[j][1])} Sid: {readid}", file=sys.stderr)
                        print(f"No segmentation calculated for {readid} in {e}: {trained_model}.", file=sys.stderr)
                        continue
                    i += 1
                    preZ[j] = float(res.Z[j])
                    t = res.transitions[3 * j:3 * j + 3]
                    trans["m1"].add(float(t[0]))
                    trans["e1"].add(float(t[1]))
                    trans["e2"].add(float(t[2]))
                    code, m, s = res.sparse(j)
                    # the read's dense model: the current one with the touched k-mers replaced (train.py:190-197); only the
                    # window mean needs all of it, the "weird training" test below needs k-mer 0
                    at0 = np.nonzero(code == 0)[0]
                    mean0_read = float(m[at0[0]]) if len(at0) else float(cur_mean[0])
                    if mean0_read < 0.5:  # "skip weird trainings": polyA k-mer mean collapsed (train.py:198-199)
                        continue
                    any_seen = True
                    if aggregate == "window-mean":
                        dm, ds = cur_mean.copy(), cur_sd.copy()
                        dm[code] = m
                        ds[code] = s
                        table.add(dm, ds)
                print(f"Zs: {list(preZ.values())}", file=sys.stderr)
                if comm is not None:  # reads trained so far and pooled transition estimate over all ranks
                    ok = [j for j in preZ]
                    tot = comm.allreduce_sum(np.array([len(ok), res.trans_counts[0::2][ok].sum(), res.trans_counts[1::2][ok].sum()]))
                    i_global = getattr(train, "_i_global", 0) + int(tot[0])
                    train._i_global = i_global
                    m1 = tot[1] / (tot[1] + tot[2]) if tot[1] + tot[2] > 0 else 0.0
                    pooled_trans = {"e1": 1.0, "m1": m1, "e2": 1.0 - m1 if tot[1] + tot[2] > 0 else 0.0}
                pw.write(f"{e},{cbatch},{i if comm is None else i_global},")
                for p in transition_params:
                    transition_params[p] = trans[p].mean() if comm is None else pooled_trans[p]
                    pw.write(f"{transition_params[p]},")
                if aggregate == "pooled":
                    pooled = res.pooled
                    if comm is not None:
                        pooled = comm.allreduce_sum(pooled)
                    w, s1, s2 = pooled[:K], pooled[K:2 * K], pooled[2 * K:]
                    hit = w > 0
                    new_mean, new_sd = cur_mean.copy(), cur_sd.copy()
                    new_mean[hit] = s1[hit] / w[hit]
                    var = np.maximum(s2[hit] / w[hit] - new_mean[hit] ** 2, 1e-12)  # NT_aligner_api.cpp:523-529
                    new_sd[hit] = np.sqrt(var)
                elif any_seen:
                    new_mean, new_sd = table.mean()
                else:
                    new_mean, new_sd = cur_mean, cur_sd
                trained_model = join(outdir, f"trained_{e}_{cbatch}.model")
                # write_kmer_model(trained_model, {name: [new_mean[c], new_sd[c]] ...}) (train.py:221-224), same bytes
                write_kmer_model_arrays(trained_model, file_kmers, k, np.asarray(new_mean)[code_of_row], np.asarray(new_sd)[code_of_row])
                pw.flush()
                # rerun with the new model to compare Zs (train.py:226-242)
                al.set_model(new_mean, new_sd)  # = Aligner(trained_model, ...) of train.py:227
                post = _z_items(al, items, raw=not host_preprocess)
                dZ = np.array([float(post.Z[j]) - z for j, z in preZ.items() if post.status[j] == 0])
                print(f"Z changes: {dZ}", file=sys.stderr)
                if comm is not None:
                    sz = comm.allreduce_sum(np.array([dZ.sum(), float(len(dZ))]))
                    deltaZ = sz[0] / sz[1] if sz[1] else 0.0
                else:
                    deltaZ = np.mean(dZ) if len(dZ) else 0.0
                pw.write(f"{deltaZ}\n")
                pw.flush()
                items = []
                if max_batches is not None and cbatch >= max_batches:
                    break
    train._i_global = 0
    print("Done training", file=sys.stderr)
    print(f"Skipped reads due to low quality: {qskips}", file=sys.stderr)


def main(argv=None) -> None:
    args = parse(argv)
    # under torch.distributed.run: every rank trains its own shard of each batch on its own GPU and the
    # pooled sufficient statistics are summed over ranks (BASELINE.json config 5)
    from dynamont_amd import parallel
    comm, local_rank = parallel.init_from_env()
    outdir = args.outdir if args.no_timestamp else args.outdir + f'_{datetime.now().strftime("%Y-%m-%d_%H-%M-%S")}'
    if comm is not None:
        outdir = parallel.broadcast_str(comm, outdir)  # one time stamp for the whole job
    makedirs(outdir, exist_ok=True)  # every rank gets here; the first one wins
    param_file = join(outdir, "params.csv")
    if args.model_path:
        model_path = args.model_path
        assert exists(model_path), "Model path does not exist"
    else:
        model_path = get_model(args.pore)
        assert exists(model_path), f"Default model not found for pore: {args.pore}, {model_path}"
    print(f"Loaded model: {basename(model_path)}", file=sys.stderr)
    if comm is not None and comm.rank == 0:
        print(f"exchange: {comm.implementation}", file=sys.stderr, flush=True)
    # a failing rank ends the whole job instead of leaving the others in their next all-reduce (parallel.abort)
    with parallel.abort_on_error(comm):
        train(args.raw, args.basecalls, args.batch_size, args.epochs, param_file, "basic", model_path, args.max_batches,
              args.pore, args.qscore, device=local_rank if comm else args.device, aggregate=args.aggregate, comm=comm,
              host_preprocess=args.host_preprocess, reference_zcheck=args.reference_zcheck)
        if comm is not None:
            comm.close()  # (collective: every rank has finished its exchanges)


if __name__ == "__main__":
    main()
