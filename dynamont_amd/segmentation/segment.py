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


def _stamp(name: str) -> None:
    """DYN_CLI_TRACE=1: a timeline of the run on stderr, seconds since the process was started (DYN_CLI_T0 = the parent's
    time.time() at Popen, else since this module was imported) -- where a cold start goes (tools/cold_start_trace.py)"""
    import os
    import time
    if os.environ.get("DYN_CLI_TRACE"):
        t0 = float(os.environ.get("DYN_CLI_T0") or _IMPORTED_AT)
        print("[cli %8.3f s] %s" % (time.time() - t0, name), file=sys.stderr, flush=True)


import time as _time  # noqa: E402

_IMPORTED_AT = _time.time()

CSV_HEADER = b"readid,signalid,start,end,basepos,base,motif,state,posterior_probability,polish\n"
POLYA = "AAAAAAAAA"

MAX_SAMPLES_IN_FLIGHT = 512 << 20  # ~1 GB of pinned int16 staging + ~4 GB of float64 on the device, whatever --depth says
LAST_RUN: dict = {}  # what the native sink of the last single-process run reported at close (bench.py's e2e_cli record)
RAW_CACHE: OrderedDict | None = None
RAW_CACHE_SIZE = 32  # open readers kept (the reference keeps 3, segment.py:44,123-139: "the pod5 files should more or less be
                     # ordered"); a basecall file over a directory of pod5 files interleaves them, a batch of 1 024 reads then touches
                     # many, and re-opening one costs its index -- a memory-mapped reader that is merely kept costs nothing
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
    p.add_argument("--depth", type=int, default=12, help="batches in flight at once, from submission to the written rows "
                   "(batches that wait while the GPU is busy are merged into one launch, whose queue balances reads of unequal "
                   "length: with 12 in flight launches of 3 batches follow each other)")
    p.add_argument("--zstd-level", type=int, default=3, help="compression level of the output frame (3: the reference's, "
                   "segment.py:60,74; on these rows level 1 is 1.8x faster AND 6 %% smaller)")
    p.add_argument("--host-threads", type=int, default=0, help="threads that compress the output (0: the CPUs this process "
                   "may use minus 4, at most 16 -- zstd level 3 of 230 MB of rows per 1 024-read batch is the largest host cost)")
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


def generate_jobs(dataPath: str, basecalls: str, minQual: float = 0, rank: int = 0, world: int = 1):
    """(rawFile, shift, scale, start, end, sequence, readid, signalid) per basecalled read
    (segment.py:189-258): ``qs`` filter, ``pi`` parent id, ``start = sp+ts``, ``end = sp+ns``,
    file ``fn`` or ``f5``, normalisation tags ``sm``/``sd``. ``rank`` / ``world``: of the jobs that pass the filter only
    those with index % world == rank (every rank of a multi-GPU run walks the basecalls and keeps its share)."""
    skipped = 0
    if _native_bam(basecalls):  # the same walk in native code (csrc/bam_reader.cpp), re-yielded read by read
        for jb in job_batches(basecalls, minQual, 4096, False, rank, world):
            for i in range(jb.n):
                yield (join(dataPath, jb.files[jb.file_id[i]]), float(jb.shift[i]), float(jb.scale[i]), int(jb.start[i]), int(jb.end[i]),
                       jb.read(i), jb.name(i), jb.sid(i))
        return
    index = -1
    for rec in iter_basecalls(basecalls):
        qs = rec.get_tag("qs")
        if minQual and qs < minQual:
            skipped += 1
            continue
        index += 1
        if index % world != rank:
            continue
        readid = rec.query_name
        signalid = rec.get_tag("pi") if rec.has_tag("pi") else readid
        ns = rec.get_tag("ns")
        ts = rec.get_tag("ts")
        sp = rec.get_tag("sp") if rec.has_tag("sp") else 0
        raw_file = join(dataPath, rec.get_tag("fn")) if rec.has_tag("fn") else join(dataPath, rec.get_tag("f5"))
        yield (raw_file, rec.get_tag("sm"), rec.get_tag("sd"), sp + ts, sp + ns, rec.query_sequence, readid, signalid)
    print(f"Skipped reads due to low quality: {skipped}", file=sys.stderr)


def available_cpus() -> int:
    """CPUs this process may use: the affinity mask, cut down to the cgroup's CPU quota when there is one"""
    import os
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def _native_bam(basecalls: str) -> bool:
    """BAM basecalls go through the native reader unless DYN_PY_BAM=1 asks for the Python parsers (pysam when present,
    bam_io.iter_bam otherwise; tests compare the two)"""
    import os
    return basecalls.endswith(".bam") and os.environ.get("DYN_PY_BAM", "0") != "1"


def job_batches(basecalls: str, minQual: float, batch_reads: int, is_rna: bool, rank: int = 0, world: int = 1):
    """generate_jobs a batch at a time, as columns (bam_io.JobBatch): the quality filter, the rank's share
    (index % world == rank) and -- ``is_rna`` -- the workers' sequence orientation (segment.py:149-153) are applied by the
    native reader."""
    from dynamont_amd.bam_io import NativeBamJobs
    reader = NativeBamJobs(basecalls, rna=is_rna, pad=POLYA, min_qual=minQual or 0.0, rank=rank, world=world)
    try:
        while True:
            jb = reader.next(batch_reads)
            if jb is None:
                break
            yield jb
        print(f"Skipped reads due to low quality: {reader.skipped}", file=sys.stderr)
    finally:
        reader.close()


def _ragged(starts: np.ndarray, counts: np.ndarray) -> np.ndarray:
    """[starts[0], starts[0]+1, .. starts[0]+counts[0]-1, starts[1], ..] as one int64 array"""
    counts = counts.astype(np.int64)
    total = int(counts.sum())
    first = np.cumsum(counts) - counts
    return np.arange(total, dtype=np.int64) - np.repeat(first, counts) + np.repeat(starts.astype(np.int64), counts)


def prepare_job_columns(jb, data_path: str, put_error):
    """prepare_job_raw for a whole JobBatch whose raw files can point at their compressed chunks
    (pod5_native.Pod5File.signal_chunks_batch). Returns (jb, chunks, raw_off, cal, owners): ``jb`` without the reads that
    failed (each reported through ``put_error`` with the worker's line, segment.py:178-187), ``chunks`` = (ptrs, nbytes,
    samples, read_off, slice_start) for dyn_batch_align_vbz_async, ``raw_off`` the prefix sums of the slice lengths,
    ``cal`` = (offset, scale, calibrated mask); or None when a reader of the batch cannot do that (the caller then takes
    the per-read path)."""
    import uuid as uuid_mod
    n = jb.n
    for i in np.flatnonzero(jb.uuid_ok == 0):  # ids the native parser did not take: uuid.UUID() accepts a few more spellings
        try:
            jb.uuid[i] = np.frombuffer(uuid_mod.UUID(jb.sid(i)).bytes, dtype=np.uint8)
            jb.uuid_ok[i] = 1
        except ValueError:
            pass
    found = np.zeros(n, dtype=bool)
    cnt = np.zeros(n, dtype=np.int64)
    cal_o = np.zeros(n, dtype=np.float32)
    cal_s = np.zeros(n, dtype=np.float32)
    parts, owners, failed = [], [], {}
    for fid, fname in enumerate(jb.files):
        sel = np.flatnonzero(jb.file_id == fid)
        if not len(sel):
            continue
        try:
            reader = get_raw(join(data_path, fname))
            fn = getattr(reader, "signal_chunks_batch", None)
            res = fn(jb.uuid[sel]) if fn is not None else None
        except Exception as error:  # noqa: BLE001  the file cannot be opened or read: every read of it fails like in the worker
            for i in sel:
                failed[int(i)] = error
            continue
        if res is None:
            return None
        ok, ptrs, nbytes, samples, roff, co, cs = res
        ok = ok & (jb.uuid_ok[sel] != 0)
        if not ok.all():  # (chunks of reads that were found under a damaged id cannot exist: ok False implies no chunks, or drop them)
            keep_chunks = np.repeat(ok, np.diff(roff.astype(np.int64)))
            ptrs, nbytes, samples = ptrs[keep_chunks], nbytes[keep_chunks], samples[keep_chunks]
        c = np.where(ok, np.diff(roff.astype(np.int64)), 0)
        found[sel], cnt[sel], cal_o[sel], cal_s[sel] = ok, c, co, cs
        parts.append((sel, c, ptrs, nbytes, samples))
        owners.append(reader)
    for i in np.flatnonzero(~found):
        i = int(i)
        error = failed.get(i, KeyError(jb.sid(i)))
        put_error(f"error: worker, {error}\tN: {int(jb.bases[i])}\tRid: {jb.name(i)}\tSid: {jb.sid(i)}")
    read_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(cnt, out=read_off[1:])
    total = int(read_off[-1])
    if len(parts) == 1 and found.all():
        _, _, ptrs, nbytes, samples = parts[0]
    else:  # several raw files in one batch: each file's chunks go to its reads' places
        ptrs, nbytes, samples = np.zeros(total, dtype=np.uint64), np.zeros(total, dtype=np.uint64), np.zeros(total, dtype=np.uint32)
        for sel, c, p_, b_, s_ in parts:
            pos = _ragged(read_off[:-1][sel], c)
            ptrs[pos], nbytes[pos], samples[pos] = p_, b_, s_
    if not found.all():
        keep = np.flatnonzero(found)
        jb = jb.take(keep)
        cnt, cal_o, cal_s = cnt[keep], cal_o[keep], cal_s[keep]
        read_off = np.zeros(jb.n + 1, dtype=np.uint64)
        np.cumsum(cnt, out=read_off[1:])
    # the [start:end) slice, clipped to the read like VbzSlice clips
    csum = np.zeros(total + 1, dtype=np.int64)
    np.cumsum(samples, out=csum[1:])
    ro = read_off.astype(np.int64)
    have = csum[ro[1:]] - csum[ro[:-1]]
    s0 = np.minimum(np.maximum(jb.start, 0), have)
    e0 = np.minimum(np.maximum(jb.end, s0), have)
    raw_off = np.zeros(jb.n + 1, dtype=np.uint64)
    np.cumsum(e0 - s0, out=raw_off[1:])
    calibrated = jb.shift <= 400  # segment.py:147: the picoampere signal for these reads, the ADC counts otherwise
    return jb, (ptrs, nbytes, samples, read_off, s0.astype(np.uint64)), raw_off, (cal_o, cal_s, calibrated), owners


def jobs_from_columns(jb, data_path: str):
    """the per-read job tuples of a JobBatch (sequences as the batch holds them: already oriented when it was read so)"""
    return [(join(data_path, jb.files[jb.file_id[i]]), float(jb.shift[i]), float(jb.scale[i]), int(jb.start[i]), int(jb.end[i]),
             jb.read(i), jb.name(i), jb.sid(i)) for i in range(jb.n)]


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


def _stored_bases(p) -> int:
    """bases of the basecall as the BAM stores it -- what the reference's worker line reports as N (segment.py:178-187).
    ``p`` = (signal, read, job, cal[, stored bases]): the fifth element where job[5] is already in aligner orientation"""
    return int(p[4]) if len(p) > 4 else len(p[2][5])


def prepare_job_raw(job, is_rna: bool, oriented: bool = False):
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
    if is_rna and not oriented:  # (``oriented``: the job comes from a JobBatch that was read in aligner orientation)
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
                self.sink.put(f"error: worker, {res.error(i)}\tN: {_stored_bases(g[i])}\tRid: {job[6]}\tSid: {job[7]}")
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

    def __init__(self, aligner: Aligner, outfile: str, raw: bool, depth: int = 3, threads: int = 8, first: bool = True,
                 last: bool = True, errfile: str | None = None, level: int = 3):
        """``first`` / ``last``: this process writes a PART of the frame (dyn_csv_sink_open_part: one process per GPU, every
        rank compresses its own rows) -- with the header line and the frame header / with the frame's closing block."""
        import ctypes as C
        from dynamont_amd import _native as N
        self.C, self.N, self.L = C, N, N.lib()
        self.aligner, self.raw, self.depth = aligner, raw, max(1, depth)
        errfile = errfile or splitext(splitext(outfile)[0])[0] + ".errors"
        h = C.c_void_p()
        err = C.create_string_buffer(1024)
        import os
        threads = int(os.environ.get("DYN_SINK_THREADS", threads))
        self.threads = threads
        rc = self.L.dyn_csv_sink_open_part(outfile.encode(), errfile.encode(), int(os.environ.get("DYN_SINK_LEVEL", level)), int(threads),
                                           int(first), int(last), C.byref(h), err, 1024)
        if rc != N.DYN_OK:
            raise OSError(err.value.decode())
        self.h = h
        self.submitted = 0
        self.samples = {}  # batch number -> signal samples (in flight: bounded by MAX_SAMPLES_IN_FLIGHT as well as by depth)
        self.keep = {}   # batch number -> (ticket, arrays the sink still reads)
        self.free = []   # result objects of consumed batches

    def put(self, line: str) -> None:
        """an error line from the producer (reads that failed before the aligner, segment.py:178-187)"""
        if self.L.dyn_csv_sink_error_line(self.h, line.encode()) != self.N.DYN_OK:
            raise OSError("cannot append to the .errors file")

    def _reap(self, block_until: int | None = None) -> None:
        """release what the sink has consumed; ``block_until``: wait (in the library, not polling) until at most that many
        batches are in flight"""
        while True:
            done = int(self.L.dyn_csv_sink_completed(self.h))
            for k in [k for k in self.keep if k < done]:
                t, res = self.keep.pop(k)[:2]
                self.samples.pop(k, None)
                t.close()
                self.free.append(res)
            if block_until is None or self.submitted - done <= block_until:
                return
            self.check()
            self.L.dyn_csv_sink_wait(self.h, self.submitted - block_until, 200)

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
            n = len(g)
            sig, sig_off, seqs, seq_off = _pack_jobs(g, scattered=self.raw)
            self._room_for(int(sig_off[-1]))
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
            bases = np.array([_stored_bases(p) for p in g], dtype=np.uint32)
            rc = self.L.dyn_csv_sink_submit_bases(self.h, self.aligner._h, t._h, C.byref(res._c), n, seqs,
                                                  seq_off.ctypes.data_as(N.c_u64_p), rid, sid,
                                                  starts.ctypes.data_as(C.POINTER(C.c_int64)), lengths.ctypes.data_as(N.c_u64_p),
                                                  bases.ctypes.data_as(C.POINTER(C.c_uint32)))
            if rc != N.DYN_OK:
                t.close()
                self.check()  # the sink's own failure, with its message
                raise RuntimeError("dyn_csv_sink_submit failed")
            self.keep[self.submitted] = (t, res, seqs, seq_off, rid, sid, starts, lengths, g, bases)  # g: the slices (and their readers) stay alive
            self.samples[self.submitted] = int(sig_off[-1])
            self.submitted += 1

    def _room_for(self, samples: int) -> None:
        """``depth`` batches in flight, fewer when they are large (every batch in flight holds its samples as int16 in
        pinned host memory and as float64 on the device)"""
        self._reap(block_until=self.depth - 1)
        while self.samples and sum(self.samples.values()) + samples > MAX_SAMPLES_IN_FLIGHT:
            self._reap(block_until=len(self.samples) - 1)

    def submit_columns(self, jb, chunks, raw_off, cal, owners) -> None:
        """One batch prepared by prepare_job_columns: compressed POD5 chunks and column arrays straight into
        dyn_batch_align_vbz_async and the sink -- no Python object per read. Reads with and without calibration
        (segment.py:147) go up as separate submissions, like in submit()."""
        C, N = self.C, self.N
        cal_o, cal_s, calibrated = cal
        if jb.n == 0:
            return
        if calibrated.any() and not calibrated.all():
            ptrs, nbytes, samples, read_off, skip = chunks
            cnt = np.diff(read_off.astype(np.int64))
            lens = np.diff(raw_off.astype(np.int64))
            for mask in (~calibrated, calibrated):  # (the order submit() sends its groups in)
                keep = np.flatnonzero(mask)
                pos = _ragged(read_off[:-1][keep], cnt[keep])
                ro = np.zeros(len(keep) + 1, dtype=np.uint64)
                np.cumsum(cnt[keep], out=ro[1:])
                so = np.zeros(len(keep) + 1, dtype=np.uint64)
                np.cumsum(lens[keep], out=so[1:])
                self.submit_columns(jb.take(keep), (ptrs[pos], nbytes[pos], samples[pos], ro, skip[keep]), so,
                                    (cal_o[keep], cal_s[keep], calibrated[keep]), owners)
            return
        self.check()
        self._room_for(int(raw_off[-1]))
        n = jb.n
        out = self.free.pop() if self.free else None
        t = self.aligner.align_vbz_async(chunks, raw_off, jb.shift, jb.scale, jb.seqs, jb.seq_off, window=3, n_sigmas=3.0, f32=False,
                                         calc_probabilities=True, out=out, calibration=(cal_o, cal_s) if calibrated.all() else None)
        res = t.result
        rid = (jb.names.ctypes.data + jb.name_off[:-1]).astype(np.uint64)  # char* of every NUL-terminated name
        sid = (jb.sids.ctypes.data + jb.sid_off[:-1]).astype(np.uint64)
        starts = np.ascontiguousarray(jb.start, dtype=np.int64)
        lengths = np.diff(raw_off).astype(np.uint64)
        seq_off = np.ascontiguousarray(jb.seq_off, dtype=np.uint64)
        bases = np.ascontiguousarray(jb.bases, dtype=np.uint32)
        rc = self.L.dyn_csv_sink_submit_bases(self.h, self.aligner._h, t._h, C.byref(res._c), n, jb.seqs, seq_off.ctypes.data_as(N.c_u64_p),
                                              C.cast(rid.ctypes.data, C.POINTER(C.c_char_p)), C.cast(sid.ctypes.data, C.POINTER(C.c_char_p)),
                                              starts.ctypes.data_as(C.POINTER(C.c_int64)), lengths.ctypes.data_as(N.c_u64_p),
                                              bases.ctypes.data_as(C.POINTER(C.c_uint32)))
        if rc != N.DYN_OK:
            t.close()
            self.check()
            raise RuntimeError("dyn_csv_sink_submit failed")
        self.keep[self.submitted] = (t, res, jb, seq_off, rid, sid, starts, lengths, chunks, raw_off, cal, owners, bases)
        self.samples[self.submitted] = int(raw_off[-1])
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
        _stamp("closing the sink (every batch submitted; the last tickets are on the GPU)")
        rc = self.L.dyn_csv_sink_close(self.h, C.byref(csv), C.byref(zst), C.byref(nerr), err, 1024)
        _stamp("sink closed: the file is complete")
        self.h = None
        LAST_RUN.update(csv_bytes=int(csv.value), compressed_bytes=int(zst.value), error_lines=int(nerr.value), batches=self.submitted,
                        depth=self.depth, compress_threads=self.threads)
        for t, *_ in self.keep.values():
            t.close()
        self.keep = {}
        _stamp("tickets released")
        print("Done segmenting reads.", file=sys.stderr)
        if rc != self.N.DYN_OK:
            raise RuntimeError(err.value.decode())


def _gather_parts(comm, parallel, outfile: str, part: str, part_err: str | None) -> None:
    """One process per GPU, all ranks done: the parts of the frame (complete zstd blocks, compressed where their rows
    were computed) travel to rank 0 rank by rank -- a part must stay in one piece -- and are appended to its own part,
    followed by the frame's closing block; the ranks' error lines likewise. The only exchange of the job."""
    import os
    rank, world = comm.rank, comm.world
    chunk_bytes = 64 << 20
    out = open(outfile, "ab") if rank == 0 else None
    for src in range(1, world):
        f = open(part, "rb") if rank == src else None
        while True:
            chunk = f.read(chunk_bytes) if f else b""
            blobs = parallel.gather_bytes(comm, chunk)
            if rank == 0 and blobs[src]:
                out.write(blobs[src])
            if not parallel.any_rank(comm, rank == src and len(chunk) == chunk_bytes):
                break
        if f:
            f.close()
    if out:
        out.write(b"\x01\x00\x00")  # DYN_ZSTD_FRAME_END: an empty last block closes the frame
        out.close()
    text = b""
    if rank != 0 and part_err and os.path.exists(part_err):
        text = open(part_err, "rb").read()
    blobs = parallel.gather_bytes(comm, text)
    if rank == 0:
        extra = b"".join(b for b in blobs[1:] if b)
        if extra:
            with open(splitext(splitext(outfile)[0])[0] + ".errors", "ab") as f:
                f.write(extra)
    else:
        for path in (part, part_err):
            if path and os.path.exists(path):
                os.remove(path)


def segment(data_path: str, basecalls: str, processes: int, outfile: str, model_path: str, pore: str, mode: str,
            minq: float = 0, device: int = 0, batch_reads: int = 1024, mem_budget_gib: float = 0.0,
            host_preprocess: bool = False, depth: int = 12, strict_ties: str = "ties", host_threads: int = 0,
            zstd_level: int = 3) -> None:
    """Counterpart of segment.py:261-371. Under ``torch.distributed.run`` every rank drives one GPU
    on the reads ``index % world == rank``, formats and compresses its rows into a part of the output
    frame, and the parts' bytes are gathered to rank 0, which owns the file (reads are independent;
    that gather at the end is the only exchange)."""
    from dynamont_amd import parallel
    comm, local_rank = parallel.init_from_env()
    rank, world = (comm.rank, comm.world) if comm else (0, 1)
    if comm:
        device = local_rank
        if rank == 0:
            print(f"exchange: {comm.implementation}", file=sys.stderr, flush=True)
        LAST_RUN["exchange"] = comm.implementation
    # Every process owns a native sink (csv_sink.cpp). One process: it writes the output file. One process per GPU: every
    # rank formats AND compresses its own rows into a part of the frame (compression is the largest host cost of the
    # output: 11 core-seconds per 32 768 reads -- left to rank 0 it would cap the job at one GPU's speed); at the end the
    # parts' bytes are gathered to rank 0, which appends them to its own and closes the frame. --parallel-zstd-frames keeps
    # the Python listener (rank 0) and the per-round gather of the formatted rows.
    native = not ZSTD_PARALLEL_FRAMES
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
            _stamp("segment(): creating the aligner")
            aligner = Aligner(model_path, pore, mode=mode, threads=1, band=400, device=device)
            _stamp("aligner ready (model parsed, device initialised, tables uploaded)")
            if mem_budget_gib:
                aligner.set_mem_budget(int(mem_budget_gib * (1 << 30)))
            aligner.set_strict(strict_ties)
            if native:
                import os
                import tempfile
                part, part_err = outfile, None
                if rank != 0:  # a part of the frame, in local scratch space; its bytes travel to rank 0 at the end
                    fd, part = tempfile.mkstemp(prefix=f"dynamont_part{rank}_", suffix=".zst")
                    os.close(fd)
                    part_err = part + ".errors"
                    parallel.register_scratch(part, part_err)  # gone with the job however it ends (parallel.abort included)
                # ONE frame out of every rank's blocks: all of them must be compressed for the window rank 0's frame header
                # declares, i.e. at rank 0's effective level -- a per-process DYN_SINK_LEVEL must not differ between ranks
                if comm is not None:
                    zstd_level = int(parallel.broadcast_str(comm, str(int(os.environ.get("DYN_SINK_LEVEL", zstd_level)))))
                    os.environ.pop("DYN_SINK_LEVEL", None)
                local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
                threads = host_threads or max(2, min(16, available_cpus() // local_world - (4 if local_world == 1 else 2)))
                pipe = sink = _NativePipeline(aligner, part, raw=not host_preprocess, depth=depth, threads=threads,
                                              first=rank == 0, last=world == 1, errfile=part_err, level=zstd_level)
                _stamp("sink open")
                if not host_preprocess and _native_bam(basecalls):
                    # BAM basecalls: the jobs arrive as columns (this rank's share of them); raw files that can point at their
                    # compressed chunks (.pod5, VBZ) are served without a Python object per read, any other reader read by read
                    for jb in job_batches(basecalls, minq, batch_reads, is_rna, rank, world):
                        prepared = prepare_job_columns(jb, data_path, sink.put)
                        if prepared is not None:
                            pipe.submit_columns(*prepared)
                            if pipe.submitted <= 2:
                                _stamp("batch %d submitted" % (pipe.submitted - 1))
                            continue
                        pending = []
                        for i, job in enumerate(jobs_from_columns(jb, data_path)):
                            try:
                                signal, read, cal = prepare_job_raw(job, is_rna, oriented=True)
                            except Exception as error:  # noqa: BLE001  (segment.py:178-187)
                                sink.put(f"error: worker, {error}\tN: {int(jb.bases[i])}\tRid: {job[6]}\tSid: {job[7]}")
                                continue
                            pending.append((signal, read, job, cal, int(jb.bases[i])))
                        pipe.submit(pending)
                else:
                    pending = []
                    for job in generate_jobs(data_path, basecalls, minq, rank, world):
                        try:
                            if host_preprocess:
                                signal, read = prepare_job(job, is_rna)
                                cal = None
                            else:
                                signal, read, cal = prepare_job_raw(job, is_rna)
                        except Exception as error:  # noqa: BLE001  (segment.py:178-187)
                            sink.put(f"error: worker, {error}\tN: {len(job[5])}\tRid: {job[6]}\tSid: {job[7]}")
                            continue
                        pending.append((signal, read, job, cal))
                        if len(pending) >= batch_reads:
                            pipe.submit(pending)
                            pending = []
                    if pending:
                        pipe.submit(pending)
                _stamp("every batch submitted")
                pipe.close()
                pipe = None
                _stamp("pipeline closed: output complete")
                if comm is not None:
                    _gather_parts(comm, parallel, outfile, part, part_err)
                parallel.remove_scratch()
                print("Done with segmentation.", file=sys.stderr, flush=True)
                return
            pipe = _Pipeline(aligner, sink, raw=not host_preprocess, depth=depth)
            job_iter = generate_jobs(data_path, basecalls, minq, rank, world)
            exhausted = False
            rounds = 0
            while True:
                pending = []
                while len(pending) < batch_reads:
                    job = next(job_iter, None)
                    if job is None:
                        exhausted = True
                        break
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
    if comm is not None:
        comm.close()  # (collective: every rank has finished its exchanges)
    if comm is not None and comm.dist.is_initialized():
        comm.dist.barrier()  # normal completion only


def main(argv=None) -> None:
    _stamp("main() entered (interpreter started, modules imported)")
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
            host_preprocess=args.host_preprocess, depth=args.depth, strict_ties=args.strict_ties, host_threads=args.host_threads,
            zstd_level=args.zstd_level)
    _stamp("segment() returned (aligner closed)")
    if argv is None and not int(__import__("os").environ.get("WORLD_SIZE", "1") or 1) > 1:
        # Invoked as the command (console script / python -m), single process, everything written and closed: leave without
        # the interpreter's and the HIP runtime's orderly teardown (0.3 s in which ~130 GB of device memory are handed back
        # allocation by allocation -- the driver reclaims them with the process anyway). Callers of main([...]) are not affected.
        import os
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
