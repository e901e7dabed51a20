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
from dynamont_amd.pod5_io import VbzSlice, get_signal, get_signal_adc, get_signal_chunks, iter_basecalls, open_pod5
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
    p.add_argument("--depth", type=int, default=6, help="batches inside the asynchronous engine at once (batches that wait "
                   "while the GPU is busy are merged into one launch: a few more than the 2 that overlap copies with kernels "
                   "let the launches balance)")
    p.add_argument("--strict-ties", type=str, default="ties", choices=["off", "ties", "start", "all"],
                   help="reproduce the reference's sums bit for bit (dyn_aligner_set_strict): 'ties' (default; 'start' is its "
                        "old name) for reads with a structural tie -- two neighbouring columns with the same emission "
                        "parameters, e.g. polyA pad + A --, 'all' for every read (1.3-1.4x), 'off' for none")
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
    """P1 without the arithmetic: the raw [start:end) slice and the aligner-orientation read. The slice is int16 ADC
    counts; ``cal`` = (offset, scale) when the reference would take the calibrated picoampere signal (``shift <= 400``,
    segment.py:147), None when it takes the ADC counts themselves. Calibration, normalisation and the Hampel filter
    then run on the device (dyn_batch_align_raw_async), bit-identically. Returns (raw, read, cal)."""
    raw_file, shift, scale, start, end, read, readid, signalid = job
    reader = get_raw(raw_file)
    chunks = get_signal_chunks(reader, signalid)
    if chunks is not None:  # a .pod5 file with VBZ chunks: they go to the library as they are (decoded on its helper threads)
        ptrs, nbytes, samples, cal_offset, cal_scale = chunks
        raw = VbzSlice(ptrs, nbytes, samples, start, end, owner=reader)
    else:
        adc, cal_offset, cal_scale = get_signal_adc(reader, signalid)
        raw = adc[start:end]
        if raw.dtype != np.int16:  # a reader that hands out something else: through the generic float64 path
            raw = raw.astype(np.float64)
    if is_rna:
        read = read[::-1]
        if not read.startswith(POLYA):
            read = POLYA + read
    return raw, read, ((cal_offset, cal_scale) if shift <= 400 and raw.dtype == np.int16 else None)


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


def _pack_jobs(pending, scattered: bool = False):
    """(signal slices, reads, jobs) of one batch -> the packed arrays of the C ABI. ``scattered``: the slices stay
    where they are (a list of arrays; the library gathers them into its pinned staging buffer on its own threads)."""
    n = len(pending)
    sig_off = np.zeros(n + 1, dtype=np.uint64)
    seq_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum([len(p[0]) for p in pending], out=sig_off[1:])
    np.cumsum([len(p[1]) for p in pending], out=seq_off[1:])
    if n and getattr(pending[0][0], "vbz", False):  # compressed POD5 chunks: flattened chunk tables
        counts = [len(p[0].ptrs) for p in pending]
        read_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(counts, out=read_off[1:])
        sig = (np.concatenate([p[0].ptrs for p in pending]), np.concatenate([p[0].nbytes for p in pending]),
               np.concatenate([p[0].samples for p in pending]), read_off, np.array([p[0].start for p in pending], dtype=np.uint64))
    elif scattered:
        sig = [p[0] for p in pending]
    else:
        sig = np.concatenate([p[0] for p in pending]) if n else np.zeros(0, dtype=np.int16)
    seqs = "".join(p[1] for p in pending).encode("latin-1")
    return sig, sig_off, seqs, seq_off


class _Pipeline:
    """The GPU side of dynamont-resquiggle as a stream of batches (the reference keeps a pool of worker processes
    permanently fed, segment.py:296-325): the caller's thread reads and slices batch k+1 while up to ``depth``
    batches are inside the asynchronous engine (dyn_batch_align_raw_async: staging, H2D, normalise + Hampel, the
    read queue, D2H, unpacking -- all overlapped between neighbouring batches), and ONE consumer thread waits for
    the oldest batch, formats its rows natively (dyn_format_csv, bytes == utils.segmentation_to_string) and hands
    them to the sink, under the kernels of the batches behind it. Python only moves references; every heavy call
    releases the GIL."""

    def __init__(self, aligner: Aligner, sink, raw: bool, depth: int = 3, threads: int = 8):
        self.aligner, self.sink, self.raw, self.threads = aligner, sink, raw, threads
        self.inflight = queue_mod.Queue(maxsize=max(1, depth))
        self.free = []            # result objects of completed batches, refilled instead of reallocated
        self.error = None
        self.rounds_done = queue_mod.Queue()
        self.consumer = threading.Thread(target=self._drain, daemon=True)
        self.consumer.start()

    def submit(self, pending, end_of_round: bool = False) -> None:
        """Queue one batch (blocks while ``depth`` batches are in flight). Batches whose raw slices have different
        dtypes (float32 pA vs int16 ADC, segment.py:147) go up as one submission per dtype."""
        self.check()
        groups = [pending]
        if self.raw and pending:  # one submission per kind of raw data: calibrated ADC, plain ADC, anything else
            kind = lambda p: (p[0].dtype.str, p[3] is not None, getattr(p[0], "vbz", False))  # noqa: E731
            kinds = sorted({kind(p) for p in pending})
            if len(kinds) > 1:
                groups = [[p for p in pending if kind(p) == kd] for kd in kinds]
        for g in groups:
            if not g:
                continue
            sig, sig_off, seqs, seq_off = _pack_jobs(g, scattered=self.raw)
            out = self.free.pop() if self.free else None
            if self.raw:
                cal = ([p[3][0] for p in g], [p[3][1] for p in g]) if g[0][3] is not None else None
                submit_raw = self.aligner.align_vbz_async if isinstance(sig, tuple) else self.aligner.align_raw_async
                t = submit_raw(sig, sig_off, [p[2][1] for p in g], [p[2][2] for p in g], seqs, seq_off,
                               window=3, n_sigmas=3.0, f32=False, calc_probabilities=True, out=out, calibration=cal)
            else:
                t = self.aligner.align_async(sig, sig_off, seqs, seq_off, True, out=out)
            self.inflight.put((t, g, seqs, seq_off))
        if end_of_round:
            self.inflight.put("round")

    def check(self) -> None:
        if self.error is not None:
            raise self.error

    def wait_round(self) -> None:
        self.rounds_done.get()
        self.check()

    def close(self) -> None:
        """Everything submitted has been formatted and handed to the sink when this returns."""
        self.inflight.put(None)
        self.consumer.join()
        self.check()

    def _finish(self, t, g, seqs, seq_off) -> None:
        from dynamont_amd._dynamont import format_csv
        res = t.wait()
        starts = [p[2][3] for p in g]
        buf, begin, end = format_csv(self.aligner, res, None, [p[2][6] for p in g], [p[2][7] for p in g], starts,
                                     [len(p[0]) + st for p, st in zip(g, starts)], threads=self.threads, compact=True,
                                     seqs_packed=(seqs, seq_off))
        total = int(end[-1]) if len(end) else 0
        if total:
            self.sink.put(buf[:total].tobytes())
        for i in np.flatnonzero(res.status[:len(g)] != 0):
            signal, read, job = g[i][:3]
            if int(res.status[i]) == 10:  # DYN_READ_BAD_SIGNAL: the reference's worker fails before the aligner (segment.py:178-187)
                self.sink.put(f"error: worker, {res.error(i)}\tN: {len(read)}\tRid: {job[6]}\tSid: {job[7]}")
                continue
            self.sink.put(f"error: native, {res.error(i)}\tT: {len(signal)}\tN: {len(read)}\tRid: {job[6]}\tSid: {job[7]}")
        t.close()
        self.free.append(res)

    def _drain(self) -> None:
        while True:
            item = self.inflight.get()
            if item is None:
                return
            if isinstance(item, str):
                self.rounds_done.put(True)
                continue
            try:
                if self.error is None:
                    self._finish(*item)
                else:
                    item[0].close()  # after a failure: only release what is still queued
            except BaseException as e:  # noqa: BLE001  (re-raised on the submitting thread)
                self.error = e
                try:
                    item[0].close()
                except Exception:  # noqa: BLE001
                    pass


class _NativePipeline:
    """Single-process form of :class:`_Pipeline` with the whole back half native (csv_sink.cpp): a batch is handed to
    dyn_batch_align_raw_async and its ticket straight on to the library's CSV sink, whose threads wait for it, format,
    compress into the one zstd frame and write -- Python only builds the next batch. Same interface as _Pipeline."""

    def __init__(self, aligner: Aligner, outfile: str, raw: bool, depth: int = 3, threads: int = 8):
        import ctypes as C
        from dynamont_amd import _native as N
        self.C, self.N, self.L = C, N, N.lib()
        self.aligner, self.raw, self.depth = aligner, raw, max(1, depth)
        errfile = splitext(splitext(outfile)[0])[0] + ".errors"
        h = C.c_void_p()
        err = C.create_string_buffer(1024)
        rc = self.L.dyn_csv_sink_open(outfile.encode(), errfile.encode(), 3, int(threads), C.byref(h), err, 1024)
        if rc != N.DYN_OK:
            raise OSError(err.value.decode())
        self.h = h
        self.submitted = 0
        self.keep = {}   # batch number -> (ticket, arrays the sink still reads)
        self.free = []   # result objects of consumed batches

    def put(self, line: str) -> None:
        """an error line from the producer (reads that failed before the aligner, segment.py:178-187)"""
        if self.L.dyn_csv_sink_error_line(self.h, line.encode()) != self.N.DYN_OK:
            raise OSError("cannot append to the .errors file")

    def _reap(self, block_until: int | None = None) -> None:
        import time
        while True:
            done = int(self.L.dyn_csv_sink_completed(self.h))
            for k in [k for k in self.keep if k < done]:
                t, res = self.keep.pop(k)[:2]
                t.close()
                self.free.append(res)
            if block_until is None or self.submitted - done <= block_until:
                return
            self.check()
            time.sleep(0.0005)

    def submit(self, pending, end_of_round: bool = False) -> None:
        C, N = self.C, self.N
        groups = [pending]
        if self.raw and pending:
            kind = lambda p: (p[0].dtype.str, p[3] is not None, getattr(p[0], "vbz", False))  # noqa: E731
            kinds = sorted({kind(p) for p in pending})
            if len(kinds) > 1:
                groups = [[p for p in pending if kind(p) == kd] for kd in kinds]
        for g in groups:
            if not g:
                continue
            self.check()
            self._reap(block_until=self.depth - 1)
            n = len(g)
            sig, sig_off, seqs, seq_off = _pack_jobs(g, scattered=self.raw)
            out = self.free.pop() if self.free else None
            if self.raw:
                cal = ([p[3][0] for p in g], [p[3][1] for p in g]) if g[0][3] is not None else None
                submit_raw = self.aligner.align_vbz_async if isinstance(sig, tuple) else self.aligner.align_raw_async
                t = submit_raw(sig, sig_off, [p[2][1] for p in g], [p[2][2] for p in g], seqs, seq_off,
                               window=3, n_sigmas=3.0, f32=False, calc_probabilities=True, out=out, calibration=cal)
            else:
                t = self.aligner.align_async(sig, sig_off, seqs, seq_off, True, out=out)
            res = t.result
            rid = (C.c_char_p * n)(*[str(p[2][6]).encode() for p in g])
            sid = (C.c_char_p * n)(*[str(p[2][7]).encode() for p in g])
            starts = np.array([p[2][3] for p in g], dtype=np.int64)
            lengths = np.diff(sig_off).astype(np.uint64)
            rc = self.L.dyn_csv_sink_submit(self.h, self.aligner._h, t._h, C.byref(res._c), n, seqs,
                                            seq_off.ctypes.data_as(N.c_u64_p), rid, sid,
                                            starts.ctypes.data_as(C.POINTER(C.c_int64)), lengths.ctypes.data_as(N.c_u64_p))
            if rc != N.DYN_OK:
                t.close()
                self.check()  # the sink's own failure, with its message
                raise RuntimeError("dyn_csv_sink_submit failed")
            self.keep[self.submitted] = (t, res, seqs, seq_off, rid, sid, starts, lengths, g)  # g: the slices (and their readers) stay alive
            self.submitted += 1

    def check(self) -> None:
        """raise as soon as the sink has failed (a batch error, zstd, the output file) instead of parsing and aligning the
        rest of the input first; close() delivers the message"""
        if self.h is not None and self.L.dyn_csv_sink_failed(self.h):
            self.close()

    def close(self) -> None:
        C = self.C
        if self.h is None:
            return
        csv, zst, nerr = C.c_uint64(), C.c_uint64(), C.c_uint64()
        err = C.create_string_buffer(1024)
        rc = self.L.dyn_csv_sink_close(self.h, C.byref(csv), C.byref(zst), C.byref(nerr), err, 1024)
        self.h = None
        for t, *_ in self.keep.values():
            t.close()
        self.keep = {}
        print("Done segmenting reads.", file=sys.stderr)
        if rc != self.N.DYN_OK:
            raise RuntimeError(err.value.decode())


def segment(data_path: str, basecalls: str, processes: int, outfile: str, model_path: str, pore: str, mode: str,
            minq: float = 0, device: int = 0, batch_reads: int = 1024, mem_budget_gib: float = 0.0,
            host_preprocess: bool = False, depth: int = 6, strict_ties: str = "ties") -> None:
    """Counterpart of segment.py:261-371. Under ``torch.distributed.run`` every rank drives one GPU
    on the reads ``index % world == rank`` and the formatted rows are gathered to rank 0, which owns
    the writer (reads are independent; the gather is the only exchange)."""
    from dynamont_amd import parallel
    comm, local_rank = parallel.init_from_env()
    rank, world = (comm.rank, comm.world) if comm else (0, 1)
    if comm:
        device = local_rank
    # single process: the library's native sink owns the output file; multi-rank: rank 0 runs the Python listener
    # and receives the other ranks' rows through the gather
    native = comm is None and not ZSTD_PARALLEL_FRAMES
    q = queue_mod.Queue() if (rank == 0 and not native) else None
    writer = None
    if q is not None:
        writer = threading.Thread(target=listener, args=(q, outfile), daemon=True)
        writer.start()
    sink = None if native else (q if comm is None else _Collector())
    is_rna = "rna" in pore

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

    # A failing rank ends the whole job (parallel.abort): a barrier in a `finally` would leave the other ranks in
    # their next collective until the process-group timeout.
    with parallel.abort_on_error(comm):
        pipe = None
        try:
            aligner = Aligner(model_path, pore, mode=mode, threads=1, band=400, device=device)
            if mem_budget_gib:
                aligner.set_mem_budget(int(mem_budget_gib * (1 << 30)))
            aligner.set_strict(strict_ties)
            if native:
                pipe = sink = _NativePipeline(aligner, outfile, raw=not host_preprocess, depth=depth)
            else:
                pipe = _Pipeline(aligner, sink, raw=not host_preprocess, depth=depth)
            job_iter = enumerate(generate_jobs(data_path, basecalls, minq))
            exhausted = False
            rounds = 0
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
                        if host_preprocess:
                            signal, read = prepare_job(job, is_rna)
                            cal = None
                        else:
                            signal, read, cal = prepare_job_raw(job, is_rna)
                    except Exception as error:  # noqa: BLE001  (segment.py:178-187)
                        _, _, _, _, _, read, readid, signalid = job
                        sink.put(f"error: worker, {error}\tN: {len(read)}\tRid: {readid}\tSid: {signalid}")
                        continue
                    pending.append((signal, read, job, cal))
                pipe.submit(pending, end_of_round=comm is not None)
                if comm is None:
                    if exhausted:
                        break
                    continue
                # multi-rank: round k is on the GPU while round k-1 (complete on every rank) is shipped. Every rank
                # takes part in every round, with empty payloads once it has run out of reads.
                if rounds >= 1:
                    pipe.wait_round()
                    ship()
                rounds += 1
                if not parallel.any_rank(comm, not exhausted):
                    pipe.wait_round()
                    ship()
                    break
            pipe.close()
            pipe = None
            print("Done with segmentation.", file=sys.stderr, flush=True)
        finally:
            if pipe is not None:  # an exception is on its way: release what is queued, keep the first error
                try:
                    pipe.close()
                except BaseException:  # noqa: BLE001
                    pass
            if q is not None:
                q.put("kill")
                writer.join()
            close_raw_cache()
    if comm is not None and comm.dist.is_initialized():
        comm.dist.barrier()  # normal completion only


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
            host_preprocess=args.host_preprocess, depth=args.depth, strict_ties=args.strict_ties)


if __name__ == "__main__":
    main()
