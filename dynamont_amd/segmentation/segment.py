#!/usr/bin/env python
"""``dynamont-resquiggle`` counterpart (reference: src/dynamont/segmentation/segment.py).

Same command line, same CSV bytes, same ``.errors`` lines. The reference forks one worker per
CPU core, each aligning one read at a time; here one process drives one MI355X and hands the
aligner batches of reads (``--batch-reads``), while a writer thread streams the zstd CSV.

  P1 preprocessing   segment.py:141-158  -> prepare_job()
  P3 job generator   segment.py:189-258  -> generate_jobs()
  P4 writer          segment.py:69-107   -> listener()
"""
from __future__ import annotations

import queue as queue_mod
import sys
import threading
from argparse import ArgumentDefaultsHelpFormatter, ArgumentParser, Namespace
from collections import OrderedDict
from os import makedirs
from os.path import basename, dirname, exists, isdir, join, splitext

import numpy as np

from dynamont_amd import Aligner, __version__
from dynamont_amd.pod5_io import get_signal, iter_basecalls, open_pod5
from dynamont_amd.segmentation.utils import get_model, hampel, segmentation_to_string
from dynamont_amd.zstd_io import open_writer

CSV_HEADER = b"readid,signalid,start,end,basepos,base,motif,state,posterior_probability,polish\n"
POLYA = "AAAAAAAAA"

RAW_CACHE: OrderedDict | None = None
RAW_CACHE_SIZE = 3  # the pod5 files should more or less be ordered (segment.py:44)
ZSTD_WORKERS = 0    # compression threads of the writer (0: up to 8 host cores; level 3 as the reference)
ZSTD_PARALLEL_FRAMES = False  # --parallel-zstd-frames: consecutive independent frames instead of the reference's one


def parse(argv=None) -> Namespace:
    """Flags of segment.py:47-67 plus the build-only GPU flags."""
    p = ArgumentParser(formatter_class=ArgumentDefaultsHelpFormatter, prog="dynamont-resquiggle")
    p.add_argument("-r", "--raw", type=str, required=True, metavar="PATH", help="Path to raw ONT data. [POD5]")
    p.add_argument("-b", "--basecalls", type=str, required=True, metavar="BAM", help="Basecalls of ONT training data as .bam file")
    p.add_argument("-o", "--outfile", type=str, required=True, help="Path to output file. Will be zstd level 3 compressed. If directory is given, will write to dynamont.csv in that directory.")
    p.add_argument("--mode", type=str, required=True, choices=["basic", "resquiggle"], help="Segmentation algorithm used for segmentation")
    p.add_argument("--processes", type=int, default=1, help="Kept for command-line compatibility; the GPU build runs one process per device")
    p.add_argument("-p", "--pore", type=str, required=True, choices=["rna002", "rna004", "dna_r10_260bps", "dna_r10_400bps"], help="Pore generation used to sequence the data")
    p.add_argument("--model_path", type=str, help="Which kmer model to use for segmentation")
    p.add_argument("-q", "--qscore", type=float, default=0.0, help="Minimal allowed quality score")
    p.add_argument("--version", action="version", version=f"%(prog)s {__version__}")
    # build-only
    p.add_argument("--device", type=int, default=0, help="HIP device ordinal")
    p.add_argument("--batch-reads", type=int, default=1024, help="Reads per GPU batch")
    p.add_argument("--mem-budget", type=float, default=0.0, help="HBM budget for lattice workspaces in GiB (0 = 90%% of free)")
    p.add_argument("--host-preprocess", action="store_true", help="normalise + Hampel-filter with NumPy on the host instead of on the GPU (same bytes)")
    p.add_argument("--parallel-zstd-frames", action="store_true",
                   help="write the CSV as consecutive independent zstd frames (readers must read across frames: "
                        "python-zstandard's defaults stop after the first). Default: one frame, like the reference, "
                        "compressed on several threads either way")
    return p.parse_args(argv)


def listener(q, outfile: str) -> None:
    """Writer (segment.py:69-107): header, then bytes -> CSV stream, str -> ``<out>.errors``,
    "kill" terminates."""
    errfile = splitext(splitext(outfile)[0])[0] + ".errors"
    num_err = 0
    with open(outfile, "wb") as raw:
        with open_writer(raw, level=3, threads=ZSTD_WORKERS, parallel_frames=ZSTD_PARALLEL_FRAMES) as output:
            output.write(CSV_HEADER)
            while True:
                result = q.get()
                if isinstance(result, str) and result == "kill":
                    break
                if isinstance(result, str):
                    with open(errfile, "a") as err:
                        err.write(result + "\n")
                    num_err += 1
                else:
                    output.write(result)
    print("Done segmenting reads.", file=sys.stderr)


def close_raw_cache():
    """segment.py:109-120"""
    global RAW_CACHE
    if RAW_CACHE is None:
        return
    while RAW_CACHE:
        _, reader = RAW_CACHE.popitem(last=False)
        try:
            reader.close()
        except Exception:  # noqa: BLE001
            pass


def get_raw(path):
    """LRU of RAW_CACHE_SIZE open readers (segment.py:123-139)."""
    global RAW_CACHE
    if RAW_CACHE is None:
        RAW_CACHE = OrderedDict()
    if path in RAW_CACHE:
        RAW_CACHE.move_to_end(path)
        return RAW_CACHE[path]
    if len(RAW_CACHE) >= RAW_CACHE_SIZE:
        _, old = RAW_CACHE.popitem(last=False)
        try:
            old.close()
        except Exception:  # noqa: BLE001
            pass
    RAW_CACHE[path] = open_pod5(path)
    return RAW_CACHE[path]


def generate_jobs(dataPath: str, basecalls: str, minQual: float = 0):
    """(rawFile, shift, scale, start, end, sequence, readid, signalid) per basecalled read
    (segment.py:189-258): ``qs`` filter, ``pi`` parent id, ``start = sp+ts``, ``end = sp+ns``,
    file ``fn`` or ``f5``, normalisation tags ``sm``/``sd``."""
    skipped = 0
    for rec in iter_basecalls(basecalls):
        qs = rec.get_tag("qs")
        if minQual and qs < minQual:
            skipped += 1
            continue
        readid = rec.query_name
        signalid = rec.get_tag("pi") if rec.has_tag("pi") else readid
        ns = rec.get_tag("ns")
        ts = rec.get_tag("ts")
        sp = rec.get_tag("sp") if rec.has_tag("sp") else 0
        raw_file = join(dataPath, rec.get_tag("fn")) if rec.has_tag("fn") else join(dataPath, rec.get_tag("f5"))
        yield (raw_file, rec.get_tag("sm"), rec.get_tag("sd"), sp + ts, sp + ns, rec.query_sequence, readid, signalid)
    print(f"Skipped reads due to low quality: {skipped}", file=sys.stderr)


def prepare_job(job, is_rna: bool):
    """P1 (segment.py:141-158): slice, float64, ``-= shift``, ``/= scale``, Hampel(3, 3 sigma);
    RNA: reverse the basecall and prepend the polyA pad unless present."""
    raw_file, shift, scale, start, end, read, readid, signalid = job
    r5 = get_raw(raw_file)
    signal = np.array(get_signal(r5, signalid, calibrated=shift <= 400)[start:end], dtype=np.float64, copy=True)
    signal -= shift
    signal /= scale
    hampel(signal)
    if is_rna:
        read = read[::-1]
        if not read.startswith(POLYA):
            read = POLYA + read
    return signal, read


def prepare_job_raw(job, is_rna: bool):
    """P1 without the arithmetic: the raw [start:end) slice (float32 picoampere when ``shift <= 400``,
    int16 ADC otherwise, segment.py:147) and the aligner-orientation read; normalisation and the
    Hampel filter then run on the device (dyn_batch_create_raw), bit-identically."""
    raw_file, shift, scale, start, end, read, readid, signalid = job
    raw = np.ascontiguousarray(get_signal(get_raw(raw_file), signalid, calibrated=shift <= 400)[start:end])
    if raw.dtype not in (np.float32, np.int16):
        raw = raw.astype(np.float64)
    if is_rna:
        read = read[::-1]
        if not read.startswith(POLYA):
            read = POLYA + read
    return raw, read


def _flush(aligner: Aligner, pending, q, is_rna: bool, kmer_size: int, threads: int = 8, raw: bool = False) -> None:
    """Align one batch and queue CSV bytes / error lines exactly as segment.py:160-176. The rows are
    formatted by the native dyn_format_csv (same bytes as utils.segmentation_to_string; Python row
    formatting would be ~50x slower than the GPU)."""
    if not pending:
        return
    from dynamont_amd._dynamont import format_csv
    signals, reads = [p[0] for p in pending], [p[1] for p in pending]
    if raw:  # one launch group per raw dtype (float32 pA vs int16 ADC)
        res = None
        kinds = sorted({s.dtype.str for s in signals})
        if len(kinds) > 1:
            for kd in kinds:
                _flush(aligner, [p for p in pending if p[0].dtype.str == kd], q, is_rna, kmer_size, threads, raw=True)
            return
        with aligner.batch_raw(signals, reads, [p[2][1] for p in pending], [p[2][2] for p in pending]) as b:
            b.align(True)
            res = b.fetch(getattr(aligner, "_fetch_cache", None))  # batches run one at a time: refill the arrays
            aligner._fetch_cache = res
    else:
        res = aligner.align_batch(signals, reads, calc_probabilities=True)
    starts = [p[2][3] for p in pending]
    buf, begin, end = format_csv(aligner, res, reads, [p[2][6] for p in pending], [p[2][7] for p in pending], starts,
                                 [len(sig) + st for sig, st in zip(signals, starts)], threads=threads)
    for i, (signal, read, job) in enumerate(pending):
        if res.status[i] != 0:
            _, _, _, _, _, _, readid, signalid = job
            q.put(f"error: native, {res.error(i)}\tT: {len(signal)}\tN: {len(read)}\tRid: {readid}\tSid: {signalid}")
        else:
            q.put(buf[int(begin[i]):int(end[i])].tobytes())


class _Collector:
    """Queue-like sink used by the non-root ranks of a multi-GPU run: rows and error lines of one
    round are collected and shipped to rank 0 (BASELINE.json config 4: gather of per-read CSV rows)."""

    def __init__(self):
        self.rows, self.errors = [], []

    def put(self, item):
        (self.errors if isinstance(item, str) else self.rows).append(item)

    def drain(self):
        r, e = b"".join(self.rows), "\n".join(self.errors).encode()
        self.rows, self.errors = [], []
        return r, e


def segment(data_path: str, basecalls: str, processes: int, outfile: str, model_path: str, pore: str, mode: str,
            minq: float = 0, device: int = 0, batch_reads: int = 1024, mem_budget_gib: float = 0.0,
            host_preprocess: bool = False) -> None:
    """Counterpart of segment.py:261-371. Under ``torch.distributed.run`` every rank drives one GPU
    on the reads ``index % world == rank`` and the formatted rows are gathered to rank 0, which owns
    the writer (reads are independent; the gather is the only exchange)."""
    from dynamont_amd import parallel
    comm, local_rank = parallel.init_from_env()
    rank, world = (comm.rank, comm.world) if comm else (0, 1)
    if comm:
        device = local_rank
    q = queue_mod.Queue() if rank == 0 else None
    writer = None
    if rank == 0:
        writer = threading.Thread(target=listener, args=(q, outfile), daemon=True)
        writer.start()
    sink = q if comm is None else _Collector()
    is_rna = "rna" in pore
    kmer_size = 5 if pore in ("dna_r9", "rna002") else 9

    def ship():  # one collective round: everything collected since the last round goes to rank 0
        rows, errs = sink.drain()
        all_rows = parallel.gather_bytes(comm, rows)
        all_errs = parallel.gather_bytes(comm, errs)
        if rank == 0:
            for blob in all_rows:
                if blob:
                    q.put(blob)
            for blob in all_errs:
                for line in blob.decode().split("\n") if blob else []:
                    q.put(line)

    try:
        aligner = Aligner(model_path, pore, mode=mode, threads=1, band=400, device=device)
        if mem_budget_gib:
            aligner.set_mem_budget(int(mem_budget_gib * (1 << 30)))
        job_iter = enumerate(generate_jobs(data_path, basecalls, minq))
        exhausted = False
        # Two-stage pipeline: this thread reads and slices the next batch (pod5 / BAM, GIL-bound Python)
        # while one worker thread runs the previous batch through the GPU and the native formatter
        # (ctypes calls, GIL released). One batch in flight bounds the memory.
        from concurrent.futures import ThreadPoolExecutor
        gpu_worker = ThreadPoolExecutor(max_workers=1)
        in_flight = None
        while True:
            pending = []
            while len(pending) < batch_reads:
                nxt = next(job_iter, None)
                if nxt is None:
                    exhausted = True
                    break
                idx, job = nxt
                if idx % world != rank:  # every rank walks the basecalls and keeps its share
                    continue
                try:
                    signal, read = prepare_job(job, is_rna) if host_preprocess else prepare_job_raw(job, is_rna)
                except Exception as error:  # noqa: BLE001  (segment.py:178-187)
                    _, _, _, _, _, read, readid, signalid = job
                    sink.put(f"error: worker, {error}\tN: {len(read)}\tRid: {readid}\tSid: {signalid}")
                    continue
                pending.append((signal, read, job))
            if in_flight is not None:
                in_flight.result()  # re-raises what the batch raised
            in_flight = gpu_worker.submit(_flush, aligner, pending, sink, is_rna, kmer_size, 8, not host_preprocess)
            if comm is None:
                if exhausted:
                    in_flight.result()
                    break
                continue
            in_flight.result()  # multi-rank: the round's rows must be complete before they are shipped
            in_flight = None
            ship()  # collective: every rank takes part in every round, with empty payloads once it is done
            if not parallel.any_rank(comm, not exhausted):
                break
        print("Done with segmentation.", file=sys.stderr, flush=True)
    finally:
        if "gpu_worker" in locals():
            gpu_worker.shutdown(wait=True)
        if rank == 0:
            q.put("kill")
            writer.join()
        close_raw_cache()
        if comm is not None and comm.dist.is_initialized():
            comm.dist.barrier()


def main(argv=None) -> None:
    args = parse(argv)
    outfile = args.outfile
    if isdir(outfile):
        outfile = join(outfile, "dynamont.csv.zst")
    elif not outfile.endswith(".zst"):
        outfile += ".zst"
    parent = dirname(outfile)
    if parent and not exists(parent):
        makedirs(parent, exist_ok=True)
    if args.model_path:
        model_path = args.model_path
        assert exists(model_path), "Model path does not exist"
    else:
        model_path = get_model(args.pore)
        assert exists(model_path), f"Default model not found for pore: {args.pore}, {model_path}"
    print(f"Loaded model: {basename(model_path)}", file=sys.stderr)
    global ZSTD_PARALLEL_FRAMES
    ZSTD_PARALLEL_FRAMES = bool(args.parallel_zstd_frames)
    segment(args.raw, args.basecalls, args.processes, outfile, model_path, args.pore, args.mode, args.qscore,
            device=args.device, batch_reads=args.batch_reads, mem_budget_gib=args.mem_budget,
            host_preprocess=args.host_preprocess)


if __name__ == "__main__":
    main()
