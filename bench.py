#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on BASELINE.json's config, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2] [--mode align|train]

A "step" is one pass of the hot path over one batch of synthetic reads THROUGH THE DROP-IN BOUNDARY:
the batch starts in host arrays the caller owns, goes through dyn_batch_align_async (validateInput +
sequenceToKmers, H2D, backward, fused forward+posterior+posterior-Viterbi, traceback, medians, D2H,
unpacking) and ends with the segment columns in caller-owned host arrays -- SURVEY.md §8(d)'s
"wall seconds of align_batch incl. H2D/D2H". Steps cycle through `--batches` DISTINCT batches and up
to `--depth` of them are in flight, so neighbouring batches overlap host work and copies with
kernels exactly as a stream of batches does in production; the timed region starts with an idle
pipeline and ends when the last batch's results are in host memory (and, for N > 1, the RCCL
collective of the last step is complete). `value` = signal samples of all ranks / max-over-ranks
wall time. `roofline` prices the dominant kernel as it ran: since round 5 the batches of the timed region are published into
ONE launch of resident waves (k_session: the resident read queue, include/dynamont_mi.h), so `roofline.launches` counts
SESSIONS, `kernel_ms_total` is the sum of their durations (HIP events on the session stream = what a rocprofv3 kernel trace of
the same command shows) and `wave_occupancy` is busy / lifetime wave-cycles; with `--no-sessions` the engine merges waiting
tickets into launches of two or three batches as in round 4 and the same fields describe those (each counted once through the
tickets' `launch_share`). `kernel_resident_Msamp_s` is ONE batch alone in one launch with inputs resident in HBM (the round-1
headline), kept as a reference point only.

N = 1 workload: BASELINE.json configs[1] -- 1 024 synthetic RNA004 reads x ~20 k samples per batch,
synthetic 9-mer model, --mode basic, band 400. N > 1: configs[3]'s per-GPU share, 4 096 reads per
rank and batch, RCCL gather of the segment rows to rank 0 (weak scaling). --mode train: configs[4]'s
per-GPU share (1 024 reads per rank and batch), RCCL all-reduce of the pooled statistics.

N > 1, what the timed region contains per step and rank: the batch through the drop-in boundary as above, then the
library's OWN exchange on the waited ticket -- dyn_comm_gather_counts + dyn_comm_gather_rows (rccl_comm.cpp: an 8-byte count
all-gather, then one ncclSend per peer / ncclRecv per peer on rank 0 in one group, straight from the batch's device rows;
rank 0 copies them to page-locked host memory: "gather_lands_in": "rank0_pinned_host"), or dyn_comm_allreduce_pooled for
--mode train. torch.distributed only hands the 128-byte communicator id round, brackets the clock with barriers and
reduces the ranks' clocks. RCCL's kernels do not fit beside resident waves: --reserve-cus (8) compute units stay free of
the resident read queue (`exchange` in the line reports what was observed).

N = 1 also carries: `plain_arithmetic` (strict mode off), `cfg2_polya` (cfg2 with a polyA tail behind the pad: EVERY read runs
the certified sweeps, as on real direct-RNA data), `scale_ref` (configs[3]'s per-GPU share on this one GPU: the N = 1 point of
the scaling curve, the workload every rank runs at N > 1) -- each with its own warm-up, timed region and roofline fraction --
and `e2e_cli`: the dynamont-resquiggle counterpart itself on a synthetic .pod5 + BAM dataset of 32 768 reads (configs[3]'s
read count), and `e2e_cli.large` on 131 072 -- file in, compressed CSV out, in this process (run_e2e_cli).

Import order (asserted below): `torch` is imported BEFORE the first dynamont_amd.Aligner is created. PyTorch's wheel
bundles its own libamdhip64 and refuses to initialise once another copy is mapped; libdynamont_mi.so links the
system one and is loaded lazily, at the first Aligner (dynamont_amd/_native.py does the same import itself in any
process that carries WORLD_SIZE > 1).
"""
from __future__ import annotations

import argparse
import collections
import json
import os
import subprocess
import sys
import shutil
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec
# ALGORITHMIC bytes per in-band lattice cell of k_read_queue, the one kernel that runs a read's whole pipeline:
KBWD_BYTES_PER_CELL = 8.0        # backward sweep: write bE
KFWD_BYTES_PER_CELL = 12.125     # forward sweep: read bE 8 B + write float LPE 4 B + 1 decision bit
KFWD_INPLACE_BYTES_PER_CELL = 16.125  # page-starved layout: read bE 8 B + write (float LPM, float LPE) 8 B + 1 bit
KTRAIN_BYTES_PER_CELL = 8.0      # forward sweep of train(): read bE
VALU_F64_PEAK_LANE_OPS = 256 * 4 * 16 * 2.4e9   # fp64 VALU issue peak: 256 CUs x 4 SIMDs x 16 lanes per cycle at 2.4 GHz

WORKLOADS = {
    # name -> (synth config, reads per batch, default number of distinct batches)
    "cfg1": ("cfg1", None, 8),
    "cfg2": ("cfg2", None, 8),
    "cfg2_small": ("cfg2", 64, 8),
    "cfg2_polya": ("cfg2_polya", None, 8),
    "cfg3": ("cfg3", None, 2),
    "cfg4_share": ("cfg4", 4096, 4),
    "cfg5_share": ("cfg5", 1024, 8),
    "short_reads": ("short_reads", None, 2),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--batches", type=int, default=0, help="distinct batches the steps cycle through (0 = workload default)")
    ap.add_argument("--depth", type=int, default=12,
                    help="batches in flight (the CLI's default too). The engine merges tickets that wait while the GPU is busy into "
                         "one launch (a launch of 1 024 reads on 1 024 waves cannot balance): 3 overlap copies with kernels, 12 "
                         "keep launches of three batches following each other")
    ap.add_argument("--pinned-inputs", action="store_true",
                    help="experiment: caller arrays in page-locked memory (dyn_host_alloc); default is ordinary NumPy memory")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-reads", type=int, default=0, help="reads in the CPU sample (0 = 4 per core)")
    ap.add_argument("--reads", type=int, default=0, help="experiment only: override reads per batch (not a bench line)")
    ap.add_argument("--strict", default="ties", choices=["off", "ties", "start", "all"],
                    help="dyn_aligner_set_strict; 'ties' is what a new handle does (reads with a structural tie run bit for "
                         "bit) and what the bench line is quoted in; 'off' / 'all' are experiments, not bench lines")
    ap.add_argument("--no-e2e", action="store_true", help="skip the `e2e_cli` record (N = 1: the whole CLI on a synthetic .pod5 + basecall file)")
    ap.add_argument("--e2e-reads", type=int, default=32768, help="reads of the e2e_cli dataset (a multiple of 4 096: that many distinct reads, repeated)")
    ap.add_argument("--e2e-large-reads", type=int, default=131072,
                    help="a second e2e_cli run on that many reads (`e2e_cli.large`: start-up and drain weigh a quarter as much); 0 = skip")
    ap.add_argument("--e2e-batch-reads", type=int, default=0, help="--batch-reads of the e2e_cli run (0 = the CLI's default)")
    ap.add_argument("--no-resident", action="store_true", help="skip the kernel_resident leg (profiling runs: every launch of the process then belongs to the timed region)")
    ap.add_argument("--no-plain", action="store_true", help="skip the `plain_arithmetic` record (the same workload with strict mode off)")
    ap.add_argument("--no-polya", action="store_true", help="skip the `cfg2_polya` record (N = 1: cfg2 with every read flagged)")
    ap.add_argument("--no-scale-ref", action="store_true", help="skip the `scale_ref` record (N = 1: configs[3]'s per-GPU share on one GPU)")
    ap.add_argument("--no-cfg3", action="store_true", help="skip the `cfg3` record (N = 1: BASELINE configs[2], 4 096 DNA reads of 10 k-100 k samples)")
    ap.add_argument("--no-train", action="store_true", help="skip the `train` record (N = 1: configs[4]'s per-GPU share, the Baum-Welch statistics pass)")
    ap.add_argument("--no-short", action="store_true", help="skip the `short_reads` record (N = 1: 16 384 reads of 150-400 bases, narrower than the band)")
    ap.add_argument("--no-cold", action="store_true", help="skip `e2e_cli.cold` (the CLI on the same dataset in a fresh child process, process start to exit)")
    ap.add_argument("--no-sessions", action="store_true", help="experiment: one launch per batch (no resident read queue)")
    ap.add_argument("--exchange-stall-ms", type=float, default=60.0,
                    help="N > 1: untimed probe steps compare the exchange beside resident sessions with the exchange beside one launch "
                         "per batch; a median above twice the baseline's + this = RCCL's kernels are not served beside the resident "
                         "read queue on this node; the ranks then run one launch per batch")
    ap.add_argument("--reserve-cus", type=int, default=8,
                    help="N > 1: compute units kept free of the resident read queue for RCCL's kernels (dyn_aligner_set_session_mode)")
    ap.add_argument("--mode", default="align", choices=["align", "train"],
                    help="align = the headline metric; train = Baum-Welch statistics pass (config 5 shape, secondary)")
    return ap.parse_args()


def start_cpu_baseline(args, workload, model_path, workdir):
    """Launch the CPU baseline as a child process BEFORE this process touches the GPU."""
    cores = min(os.cpu_count() or 1, 16)
    n = args.cpu_reads or 4 * cores
    out = os.path.join(workdir, "cpu_baseline.json")
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--model", model_path,
           "--workload", workload, "--mode", args.mode, "--reads", str(n), "--procs", str(cores), "--out", out]
    return subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE), out


def load_traffic():
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/traffic.json), or None."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(p):
        try:
            return json.load(open(p))
        except Exception:
            return None
    return None


_E2E_DATASETS: dict = {}


def make_e2e_dataset(n_reads: int, workdir: str, sub: str) -> dict:
    """the synthetic .pod5 + BAM dataset of an e2e_cli record (written once per `sub`, CPU only: no GPU call)"""
    if sub in _E2E_DATASETS:
        return _E2E_DATASETS[sub]
    from dynamont_amd import synth
    d = os.path.join(workdir, sub)
    os.makedirs(d, exist_ok=True)
    model = synth.write_model(os.path.join(d, "syn9.model"), 9, seed=7, stdev=0.15)
    _, mean, sd = synth.read_model_file(model)
    distinct = max(1, min(n_reads, 4096))
    rep = max(1, n_reads // distinct)
    t0 = time.perf_counter()
    reads = synth.make_reads(5, distinct, "rna004", mean, sd, 2000)
    raw, bam, _ = synth.write_dataset(os.path.join(d, "in"), "ds", reads, "rna004", seed=1, container="pod5", replicate=rep, basecalls="bam")
    samples = sum(len(r.signal) for r in reads) * rep
    del reads
    ds = {"dir": d, "model": model, "raw": raw, "bam": bam, "samples": samples, "reads": distinct * rep, "distinct": distinct, "rep": rep,
          "t_gen": time.perf_counter() - t0, "out": os.path.join(d, "out.csv")}
    _E2E_DATASETS[sub] = ds
    return ds


def e2e_cli_args(ds: dict, strict: str, batch_reads: int = 0) -> list:
    return ["-r", os.path.join(ds["dir"], "in"), "-b", ds["bam"], "-o", ds["out"], "--mode", "basic", "-p", "rna004", "--model_path", ds["model"],
            "--strict-ties", strict] + (["--batch-reads", str(batch_reads)] if batch_reads else [])


def run_e2e_cold(ds: dict, strict: str, batch_reads: int, vram: str) -> dict:
    """The dataset through `python -m dynamont_amd.segmentation.segment ...` in a FRESH CHILD PROCESS, wall clock from Popen
    to exit -- what a user who types the command waits for: interpreter start-up, imports, library load, model parse,
    allocation of the lattice pool and of every buffer, then everything the warm record times. `vram` says in what state
    the device was: the driver scrubs memory a process has freed before it hands it out again (tools/ubench/vmm_probe.hip:
    112 GiB come in 0.000 s when clean and in 3-7 s right after another process has freed them)."""
    for f in (ds["out"] + ".zst", ds["out"] + ".errors"):
        if os.path.exists(f):
            os.remove(f)
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, "-m", "dynamont_amd.segmentation.segment"] + e2e_cli_args(ds, strict, batch_reads), env=env,
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    dt = time.perf_counter() - t0
    rec = {"value": round(ds["samples"] / dt / 1e6, 3), "unit": "Msamp/s", "wall_s": round(dt, 3), "returncode": r.returncode, "vram": vram,
           "output_bytes": os.path.getsize(ds["out"] + ".zst") if os.path.exists(ds["out"] + ".zst") else None}
    if r.returncode != 0:
        rec["stderr_tail"] = r.stderr.decode(errors="replace")[-500:]
    return rec


def run_e2e_cli(n_reads: int, workdir: str, strict: str, batch_reads: int = 0, sub: str = "e2e") -> dict:
    """north_star's "throughput on synthetic pod5+bam": the dynamont-resquiggle counterpart end to end, in this process.
    A .pod5 file (VBZ-compressed int16 chunks) and an unaligned BAM (dorado's tags) written by synth.write_dataset ->
    dynamont_amd.segmentation.segment.main -> out.csv.zst (reference: src/dynamont/segmentation/segment.py:261-371).
    Timed: model load, BAM parse, pod5 open + VBZ decode, pA calibration + normalisation + Hampel on the device, the
    DP, CSV formatting, zstd, file write. Not timed: interpreter start-up, dataset generation; the lattice pool and the
    batch buffers are the ones the bench's own handle has just parked -- the `cold` records time all of it."""
    from dynamont_amd.segmentation import segment as seg
    ds = make_e2e_dataset(n_reads, workdir, sub)
    out, raw, bam = ds["out"], ds["raw"], ds["bam"]
    for f in (out + ".zst", out + ".errors"):
        if os.path.exists(f):
            os.remove(f)
    t0 = time.perf_counter()
    seg.main(e2e_cli_args(ds, strict, batch_reads))
    dt = time.perf_counter() - t0
    err = out + ".errors" if os.path.exists(out + ".errors") else None
    rec = {"value": round(ds["samples"] / dt / 1e6, 3), "unit": "Msamp/s", "wall_s": round(dt, 3), "reads": ds["reads"],
           "reads_per_s": round(ds["reads"] / dt, 1), "samples": ds["samples"],
           "input": f"{os.path.basename(raw)} ({os.path.getsize(raw) / 1e6:.0f} MB, VBZ) + {os.path.basename(bam)} ({os.path.getsize(bam) / 1e6:.1f} MB); "
                    f"{ds['distinct']} distinct synthetic rna004 reads x {ds['rep']}, written by synth.write_dataset in {ds['t_gen']:.1f} s (not timed)",
           "output": (f"out.csv.zst, {os.path.getsize(out + '.zst') / 1e6:.1f} MB (one zstd frame, level 3) holding "
                      f"{seg.LAST_RUN.get('csv_bytes', 0) / 1e6:.0f} MB of CSV rows") if os.path.exists(out + ".zst") else None,
           "output_bytes": os.path.getsize(out + ".zst") if os.path.exists(out + ".zst") else None,
           "batches_in_flight": seg.LAST_RUN.get("depth"), "compress_threads": seg.LAST_RUN.get("compress_threads"),
           "error_lines": sum(1 for _ in open(err)) if err else 0, "strict_mode": strict,
           "batch_reads": batch_reads or "CLI default",
           "timed": "segment.main: model load, BAM parse, pod5 VBZ decode, device preprocessing, DP, CSV format, zstd, write",
           "not_timed": "interpreter start-up; allocation of the lattice pool and of the batch buffers (those the bench's own handle has just parked are taken over) -- `cold` times all of it"}
    return rec


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD `python -m torch.distributed.run`
    (never exec: see the GPU box's rules; nothing in this process has touched the GPU or imported torch yet), hand its
    output through and return its exit code."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this host driver
    # stdout carries the ONE JSON line and nothing else (the launcher and gloo print banners there): the rest goes to stderr
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        (sys.stdout if line.startswith("{") else sys.stderr).write(line)
        sys.stdout.flush()
    return proc.wait()


class Workload:
    """the stream of distinct batches of one workload, in caller-owned host arrays (cached per name)"""
    _cache: dict = {}

    def __init__(self, name: str, rank: int, args, model_cache: dict, workdir: str, n_distinct: int = 0):
        from dynamont_amd import synth
        import numpy as np
        cfgname, per_batch, n_batches = WORKLOADS[name]
        cfg = dict(synth.CONFIGS[cfgname])
        if per_batch:
            cfg["n_reads"] = per_batch
        if args.reads:
            cfg["n_reads"] = args.reads
        self.name, self.cfg = name, cfg
        self.n_batches = max(1, n_distinct or args.batches or n_batches)
        self.pore = cfg["pore"]
        _, _rna, self.k = synth.PORES[self.pore]
        if self.k not in model_cache:
            model_cache[self.k] = synth.write_model(os.path.join(workdir, f"syn{self.k}.model"), self.k, seed=7, stdev=0.25 if self.k == 5 else 0.15)
        self.model_path = model_cache[self.k]
        _, mean, sd = synth.read_model_file(self.model_path)
        self.batches = []
        for j in range(self.n_batches):
            reads = synth.make_reads(cfg["seed"] + 1000 * rank + 100003 * j, cfg["n_reads"], self.pore, mean, sd, cfg["n_bases"], polya=cfg.get("polya"))
            sig, sig_off, seqs, seq_off = synth.pack_reads(reads)
            if args.pinned_inputs:
                from dynamont_amd._dynamont import pinned_empty
                ps = pinned_empty(sig.size, np.float64)
                ps[:] = sig
                sig = ps
            self.batches.append((sig, sig_off, seqs, seq_off, len(reads)))
            del reads
        self.samples_of = [int(b[1][-1]) for b in self.batches]

    @classmethod
    def get(cls, name, rank, args, model_cache, workdir, n_distinct=0):
        key = (name, rank)
        if key not in cls._cache:
            cls._cache[key] = cls(name, rank, args, model_cache, workdir, n_distinct)
        return cls._cache[key]

    @classmethod
    def drop(cls, name, rank):
        cls._cache.pop((name, rank), None)   # (a side record's batches: gigabytes of host memory the next record wants)


def measure(al, wl: Workload, args, steps: int, warmup: int, step0: int, mode: str, exch, sync):
    """`warmup` untimed steps, then exactly `steps` timed ones (barrier + device synchronisation on both sides). Returns the
    raw sums; `exch` (N > 1) is the exchange of every step: the library's own RCCL path (dyn_comm_*)."""
    depth = max(1, args.depth)
    if max(b[4] for b in wl.batches) > 1536:
        # each ticket in flight holds its samples twice (pinned staging, device): 4 batches of 4 096 reads overlap everything
        depth = min(depth, 4)
    free_results: list = []  # result objects are reused: fresh 50 MB arrays per batch would be page-faulted in every time
    kern = collections.Counter()
    launches = collections.Counter()
    done_steps = [0]
    exch_ms: list = []

    def submit(j):
        sig, sig_off, seqs, seq_off, _n = wl.batches[j % wl.n_batches]
        if mode == "train":
            return al.train_async(sig, sig_off, seqs, seq_off, pooled=False, emissions=False)
        out = free_results.pop() if free_results else None
        return al.align_async(sig, sig_off, seqs, seq_off, True, out=out)

    def finish(t, timed):
        res = t.wait()  # results are in host arrays from here on
        if exch is not None:
            te = time.perf_counter()
            exch.step(t, mode)
            exch_ms.append((time.perf_counter() - te) * 1e3)
        if timed:
            tm = t.timing()
            share = tm["launch_share"]
            if tm["launches"] == 0 and tm["reads_ok"]:
                # a ticket of the RESIDENT read queue: no launch of its own -- ms_* are its reads' wave time / waves, which add up
                # over tickets; the session kernels' own durations come from al.session_stats() around the timed region
                share = 1.0
                launches["resident_tickets"] += 1
            else:
                # tickets that waited together were merged into ONE launch: each reports that launch's timing and its share of it
                launches["classic_ms_dp"] += tm["ms_dp"] * share
            for key in ("ms_dp", "ms_backward", "ms_forward", "ms_trace", "ms_total", "wave_wait_share", "wave_occupancy",
                        "ms_backward_strict", "ms_forward_strict"):
                kern[key] += tm[key] * share
            for key in ("launches", "lp_inplace", "cert_fallbacks", "cert_rows"):
                launches[key] += tm[key] * share
            for key in ("cells", "reads_strict"):
                launches[key] += tm[key]
            launches["pool_pages"], launches["page_rows"] = tm["pool_pages"], tm["page_rows"]
            launches["n_static"], launches["n_waves"] = tm["n_static"], tm["n_waves"]
            done_steps[0] += 1
        t.close()
        if mode == "align":
            free_results.append(res)
        return res

    def run(first, count, timed):
        q = collections.deque()
        last = None
        for s_ in range(first, first + count):
            if len(q) >= depth:
                last = finish(q.popleft(), timed)
            q.append(submit(s_))
        while q:
            last = finish(q.popleft(), timed)
        return last

    run(step0, warmup, False)
    sync()
    warm_exch = list(exch_ms)
    exch_ms.clear()
    sess0 = al.session_stats()  # (closes the warm-up's session: the timed region starts with an idle pipeline and no resident wave)
    t0 = time.perf_counter()
    last = run(step0 + warmup, steps, True)
    sync()
    elapsed = time.perf_counter() - t0
    sess1 = al.session_stats()
    sess = {k_: sess1[k_] - sess0[k_] for k_ in sess1 if k_ != "wave_occupancy"}
    n_samples = sum(wl.samples_of[s_ % wl.n_batches] for s_ in range(step0 + warmup, step0 + warmup + steps))
    n_reads = sum(wl.batches[s_ % wl.n_batches][4] for s_ in range(step0 + warmup, step0 + warmup + steps))
    return {"elapsed": elapsed, "kern": kern, "launches": launches, "sess": sess, "steps_done": done_steps[0], "last": last, "depth": depth,
            "samples": n_samples, "reads": n_reads, "exchange_ms": exch_ms, "warmup_exchange_ms": warm_exch}


def roofline_of(m: dict, workload: str, mode: str, args, profile: dict) -> dict:
    """the dominant kernel's ALGORITHMIC bytes over its duration (HIP events on the kernel's own stream: one
    interval per launch -- with the resident read queue one per SESSION, the waves stay on the chip across batches -- and
    launches on one stream never overlap, so the sum is the union of the intervals)."""
    kern, launches, sess = m["kern"], m["launches"], m["sess"]
    steps = max(1, m["steps_done"])
    cells_total = launches["cells"]
    n_launch = max(1e-9, launches["launches"] + sess["sessions"])
    kernel_ms_total = launches["classic_ms_dp"] + sess["ms"]
    ms_dp = kernel_ms_total / n_launch
    cells_per_launch = cells_total / n_launch
    inplace = bool(launches["lp_inplace"])
    resident = bool(sess["sessions"]) and not launches["launches"]
    if mode == "train":
        kname = "k_read_queue<JOB_TRAIN> (per read: backward sweep, forward sweep + Baum-Welch statistics)"
        bpc_f, tkey, survey_bpc = KTRAIN_BYTES_PER_CELL, "train", 32.0
    else:
        kname = ("k_session (resident waves; per read: backward, forward + posterior + posterior-Viterbi, traceback)" if resident
                 else "k_read_queue<JOB_ALIGN%s> (per read: backward, forward + posterior + posterior-Viterbi, traceback)" % ("_INPLACE" if inplace else ""))
        bpc_f, tkey, survey_bpc = (KFWD_INPLACE_BYTES_PER_CELL if inplace else KFWD_BYTES_PER_CELL), "align", 64.125
    bpc = KBWD_BYTES_PER_CELL + bpc_f
    secs = kernel_ms_total * 1e-3
    achieved = cells_total * bpc / secs / 1e9 if secs else 0.0
    # counters are quoted from the committed rocprofv3 --pmc passes over this command, per lattice cell, and only for the
    # workload and layout they were measured on (profiles/traffic.json: what was run, the units, the gfx950 correction)
    tinfo = (profile or {}).get(tkey) or {}
    per_cell_ok = workload == tinfo.get("workload") and not args.reads and not inplace and tinfo.get("cells_per_launch")
    t_per_cell = tinfo["bytes_per_launch"] / tinfo["cells_per_launch"] if per_cell_ok else None
    valu_per_cell = tinfo.get("valu_wave_instructions_per_cell") if per_cell_ok else None
    share = lambda key: kern[key] / kern["ms_dp"] if kern["ms_dp"] else 0.0
    occ = (kern["wave_occupancy"] + (sess["wave_cycles_busy"] / sess["wave_cycles_life"] * sess["sessions"] if sess["wave_cycles_life"] else 0.0)) / n_launch
    r = {
        # the roofline `frac` is taken against (the contract's "hbm" | "mfma"); what actually binds the kernel is named beside it
        "bound": "hbm",
        "binding_limit": "fp64 VALU issue under the package power cap (DESIGN.md section 7: the access pattern alone tops out at 5.4-7.0 TB/s, "
                         "the kernel issues at valu_issue_frac of the fp64 rate while the shader clock sits at ~2.1 of 2.4 GHz): neither HBM nor MFMA",
        "kernel": kname,
        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
        "traffic": int(round(t_per_cell * cells_per_launch)) if t_per_cell else None,
        "traffic_source": ("profiles/traffic.json (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE per lattice cell, round %s) x cells_per_launch" % (profile or {}).get("round")) if t_per_cell else None,
        "traffic_over_algorithmic": round(t_per_cell / bpc, 3) if t_per_cell else None,
        # the same launches priced at the HBM bytes the counters saw, and at SURVEY 8(d)'s own three-pass accounting
        "frac_counter_traffic": round(cells_total * t_per_cell / secs / 1e9 / HBM_PEAK_GBPS, 4) if t_per_cell and secs else None,
        "survey_accounting": {"bytes_per_cell": survey_bpc, "frac": round(cells_total * survey_bpc / secs / 1e9 / HBM_PEAK_GBPS, 4) if secs else None,
                              "note": "SURVEY.md 8(d) prices the reference's formulation (forward, backward + posterior, Viterbi as separate passes "
                                      "over stored matrices); this design runs backward first and fuses the rest into one sweep: %.3g B per cell "
                                      "instead of %.3g (the counters confirm it), so priced at the survey's bytes the same launches read above "
                                      "1 of peak -- not a fraction, the stated deviation" % (bpc, survey_bpc)},
        # VALU wave-instructions per cell (committed SQ counter pass) x 64 lanes over the fp64 issue peak: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz
        "valu_issue_frac": round(cells_total * valu_per_cell * 64 / secs / VALU_F64_PEAK_LANE_OPS, 4) if valu_per_cell and secs else None,
        "bytes_per_cell": bpc, "cells_per_launch": round(cells_per_launch), "launches": round(n_launch, 3),
        "cells_total": int(cells_total), "kernel_ms_total": round(kernel_ms_total, 3), "avg_launch_ms": round(ms_dp, 3),
        # share of wave time per phase (device cycle counters)
        "wave_time_share": {"backward": round(share("ms_backward"), 4), "forward": round(share("ms_forward"), 4),
                            # one launch per batch: the launches' own counters; resident queue: wave-cycles of paged sessions
                            # spent getting pages / lifetime wave-cycles (dyn_aligner_session_page_wait)
                            "waiting_for_pages": round(kern["wave_wait_share"] / n_launch + (sess.get("wave_cycles_pages", 0) / sess["wave_cycles_life"] if sess["wave_cycles_life"] else 0.0), 4),
                            # resident queue: where the idle share sat, counted by the waves (dyn_aligner_session_idle_split):
                            # before a wave's first read, and in its LAST turn (no read left to claim: what a finite run pays
                            # once per session and a stream does not)
                            "before_first_read": round(sess.get("wave_cycles_before_first_read", 0) / sess["wave_cycles_life"], 4) if sess["wave_cycles_life"] else None,
                            "last_turn": round(sess.get("wave_cycles_last_turn", 0) / sess["wave_cycles_life"], 4) if sess["wave_cycles_life"] else None},
        "phases": {"backward_sweep": {"bytes_per_cell": KBWD_BYTES_PER_CELL, "ms": round(kernel_ms_total * share("ms_backward"), 3),
                                      "frac": round(cells_total * KBWD_BYTES_PER_CELL / (kernel_ms_total * share("ms_backward") * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if share("ms_backward") else None},
                   "forward_sweep": {"bytes_per_cell": bpc_f, "ms": round(kernel_ms_total * share("ms_forward"), 3),
                                     "frac": round(cells_total * bpc_f / (kernel_ms_total * share("ms_forward") * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if share("ms_forward") else None}},
        # classic launches: sum of wave lifetimes / (waves x longest lifetime), averaged over launches; resident queue: busy /
        # lifetime wave-cycles of the sessions (a wave is busy from claiming a read to releasing its results)
        "wave_occupancy": round(occ, 4),
        "batches_per_launch": round(steps / n_launch, 3),
        "resident_queue": ({"sessions": sess["sessions"], "tickets": sess["tickets"], "reads": sess["reads"], "ms": round(sess["ms"], 3),
                            "wave_cycles_busy": sess["wave_cycles_busy"], "wave_cycles_idle": sess["wave_cycles_idle"],
                            "wave_cycles_life": sess["wave_cycles_life"], "waves": sess["waves"], "aborted": sess["aborted"]} if sess["sessions"] else None),
        "page_pool": {"pages": launches["pool_pages"], "rows_per_page": launches["page_rows"],
                      "reads_with_reserved_pages": launches["n_static"], "waves": launches["n_waves"]},
    }
    return r


class Exchange:
    """N > 1: the exchange BASELINE.json names, through the library's OWN RCCL path (dyn_comm_*, rccl_comm.cpp) -- not
    torch.distributed, which only hands the 128-byte communicator id round and brackets the timed region.
    align: dyn_comm_gather_counts + dyn_comm_gather_rows (8-byte count all-gather, then ONE ncclSend per peer / ncclRecv per
    peer on the root inside a group: each peer's rows cross its own xGMI link once, straight from the batch's device rows; on
    rank 0 they land in page-locked host memory). train: dyn_comm_allreduce_pooled (ncclAllReduce, sum, in place on the
    device-resident (w, s1, s2)[4^k])."""

    def __init__(self, dist, torch, rank, world, device, coll_dev, cap_rows_per_rank, num_kmers):
        from dynamont_amd._dynamont import RcclComm, pinned_empty
        import numpy as np
        uid = torch.zeros(128, dtype=torch.uint8, device=coll_dev)
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(RcclComm.unique_id()), dtype=torch.uint8))
        dist.broadcast(uid, src=0)
        self.comm = RcclComm(bytes(uid.cpu().numpy().tobytes()), rank, world, device)
        self.rank, self.world, self.num_kmers = rank, world, num_kmers
        self.rows = pinned_empty(max(1, cap_rows_per_rank * world), RcclComm.ROW) if rank == 0 else None
        self.rows_gathered = 0

    def step(self, ticket, mode):
        if mode == "train":
            self.comm.allreduce_pooled(ticket, self.num_kmers)
        else:
            _rows, counts = self.comm.gather_rows(ticket, root=0, out=self.rows)
            self.rows_gathered += int(counts.sum())

    def close(self):
        self.comm.close()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s): the line would not be an N = {args.gpus} measurement")
    n_gpus = world
    workload = args.workload or ("cfg5_share" if args.mode == "train" else ("cfg2" if n_gpus == 1 else "cfg4_share"))
    workdir = tempfile.mkdtemp(prefix=f"dyn_bench_r{rank}_")
    model_cache: dict = {}
    wl = Workload.get(workload, rank, args, model_cache, workdir)

    cpu_proc = cpu_out = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        cpu_proc, cpu_out = start_cpu_baseline(args, workload, wl.model_path, workdir)
        # the CPU sample uses every host core: let it finish before timing the GPU
        _, err = cpu_proc.communicate()
        if cpu_proc.returncode != 0:
            print("cpu baseline failed:\n" + err.decode(errors="replace")[-2000:], file=sys.stderr)

    # e2e_cli.cold, first leg: the CLI in a fresh child process BEFORE this process touches the GPU -- the device as the bench
    # found it (whatever ran before has long exited: its memory is scrubbed)
    cold_first = None
    want_e2e = rank == 0 and n_gpus == 1 and args.mode == "align" and not args.no_e2e and "WORLD_SIZE" not in os.environ \
        and not os.environ.get("DYN_BENCH_FORCE_DIST")
    if want_e2e and not args.no_cold:
        try:
            strict0 = {"start": "ties"}.get(args.strict, args.strict)
            cold_first = run_e2e_cold(make_e2e_dataset(args.e2e_reads, workdir, "e2e"), strict0, args.e2e_batch_reads,
                                      "as the bench found it (before this process touched the GPU)")
        except Exception as e:  # the headline stands on its own
            cold_first = {"error": f"{type(e).__name__}: {e}"}

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: dynamont_amd has no CPU compute path")
    # Rehearsal hooks (control-flow checks on a 1-GPU box; never set by the driver):
    #   DYN_BENCH_BACKEND=gloo   torch's bracket collectives over gloo instead of nccl/RCCL (the exchange itself is always dyn_comm_*)
    #   DYN_BENCH_ONE_DEVICE=1   every rank uses cuda:0
    #   DYN_BENCH_FORCE_DIST=1   run the exchange even with a single rank (exercises dyn_comm_* on one GPU)
    backend = os.environ.get("DYN_BENCH_BACKEND", "nccl")
    one_device = bool(os.environ.get("DYN_BENCH_ONE_DEVICE")) and n_gpus > 1
    if os.environ.get("DYN_BENCH_ONE_DEVICE"):
        local_rank = 0
        if one_device:
            # RCCL refuses two ranks of a communicator on one device; with a NCCL_HOSTID of its own every rank looks like a
            # one-GPU node and RCCL connects them by sockets over loopback: the exchange below is then the REAL dyn_comm_* code
            # (count all-gather, grouped ncclSend / ncclRecv, ncclAllReduce) on a transport that is not xGMI
            from dynamont_amd import parallel
            parallel.one_device_rccl_env(rank)
    elif torch.cuda.device_count() < n_gpus:
        raise SystemExit(f"bench.py: --gpus {n_gpus} but only {torch.cuda.device_count()} device(s) visible; refusing to measure fewer GPUs than asked for")
    torch.cuda.set_device(local_rank)
    use_dist = n_gpus > 1 or bool(os.environ.get("DYN_BENCH_FORCE_DIST"))
    rccl_channel_cap = None
    if use_dist and not args.no_sessions and args.reserve_cus > 0:
        # Measured (tools/ubench/resident_probe.hip big): beside a resident session a kernel whose
        # workgroups need CUs of their own starts at once iff it has NO MORE workgroups than there are free CUs -- one more and
        # it waits for the session's end. RCCL launches one workgroup per channel: cap the channels at the CUs the sessions
        # leave free, BEFORE the first communicator of the process reads its parameters (torch's included). Whether this RCCL
        # build honours the caps is what the two untimed probe steps below find out; if not, every rank falls back together.
        rccl_channel_cap = int(args.reserve_cus)
        for key in ("NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS", "NCCL_MAX_P2P_NCHANNELS"):  # (MIN: an arch default above the cap would lift it)
            os.environ.setdefault(key, str(rccl_channel_cap))
    if use_dist:
        if n_gpus == 1:  # DYN_BENCH_FORCE_DIST without a launcher: a one-rank group on the loopback
            for key, val in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29561")):
                os.environ.setdefault(key, val)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = f"cuda:{local_rank}"
    coll_dev = dev if backend == "nccl" else "cpu"

    from dynamont_amd import Aligner, _native
    assert "torch" in sys.modules and _native._lib is None, "torch must be imported before libdynamont_mi.so is loaded"

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    profile = load_traffic()
    al = Aligner(wl.model_path, wl.pore, mode="basic", band=400, device=local_rank)
    al.set_strict(args.strict)
    exch = None
    reserved_cus = 0
    if use_dist:
        # RCCL's kernels (37 KB of LDS, 248-256 registers) do not fit beside a resident session: a few CUs stay free for them
        reserved_cus = args.reserve_cus
        if one_device:
            # N ranks share the card: every rank's sessions take an N-th of the CUs the reserve leaves (a session is one
            # workgroup per CU it uses and two sessions cannot share a CU: 150 KB of LDS each)
            n_cus = torch.cuda.get_device_properties(local_rank).multi_processor_count
            reserved_cus = n_cus - max(1, (n_cus - args.reserve_cus) // n_gpus)
            # ... and an N-th of the memory (a handle plans its lattice pool with 90 % of what is free)
            _free_b, tot_b = torch.cuda.mem_get_info(local_rank)
            al.set_mem_budget(int(0.6 * tot_b / n_gpus))
        al.set_session_mode(not args.no_sessions, reserved_cus)
        cap_local = max(al.segment_capacity(b[3]) for b in wl.batches) if args.mode == "align" else 0
        t = torch.tensor([cap_local], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        exch = Exchange(dist, torch, rank, n_gpus, local_rank, coll_dev, int(t.item()), al.num_kmers)
    elif args.no_sessions:
        al.set_session_mode(False)

    sessions_with_exchange = None
    resident_with_exchange = use_dist and not args.no_sessions
    if resident_with_exchange:
        # Are RCCL's kernels served beside the resident waves on this node? They need whole free CUs (37 KB of LDS, 248-256
        # registers per lane); the reserved CUs are meant for them. UNTIMED probe steps decide, and they compare like with like:
        # three steps with one launch per batch first (the first exchange of a communicator also connects the peers: hundreds of
        # milliseconds that say nothing about sessions), then three steps with the resident queue. An exchange that has to wait
        # for a session to END takes as long as the pipeline needs to run dry -- every one of them, so the MEDIAN tells: above
        # twice the baseline's + --exchange-stall-ms every rank falls back to one launch per batch, together, and the line says so.
        def med(v):
            return sorted(v)[len(v) // 2] if v else 0.0
        al.set_session_mode(False)
        base = measure(al, wl, args, 3, 0, 0, args.mode, exch, sync)
        al.set_session_mode(True, reserved_cus)
        probe = measure(al, wl, args, 3, 0, 0, args.mode, exch, sync)
        t = torch.tensor([med(base["exchange_ms"][1:]), med(probe["exchange_ms"]), max(probe["exchange_ms"] or [0.0])], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        base_ms, probe_ms, probe_worst = (float(x) for x in t.tolist())
        served = probe_ms < 2.0 * base_ms + args.exchange_stall_ms
        sessions_with_exchange = {"exchange_ms_one_launch_per_batch": round(base_ms, 3), "exchange_ms_beside_sessions": round(probe_ms, 3),
                                  "worst_beside_sessions": round(probe_worst, 3), "rule": "median beside sessions < 2 x baseline + %.0f ms" % args.exchange_stall_ms,
                                  "exchanges_served_beside_sessions": served}
        if not served:
            al.set_session_mode(False)
            resident_with_exchange = False
    m = measure(al, wl, args, args.steps, args.warmup, 0, args.mode, exch, sync)
    elapsed = m["elapsed"]
    per_rank_ms = [elapsed * 1e3]
    if use_dist:
        t = torch.tensor([elapsed], device=coll_dev, dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(n_gpus)]
        dist.all_gather(allt, t)
        per_rank_ms = [float(x.item()) * 1e3 for x in allt]
        elapsed = max(per_rank_ms) / 1e3
        tot = torch.tensor([m["samples"], m["reads"]], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_samples, total_reads = float(tot[0].item()), float(tot[1].item())
    else:
        total_samples, total_reads = float(m["samples"]), float(m["reads"])
    last = m["last"]
    ok = int((last.status == 0).sum()) if last is not None else 0

    # ---- secondary: kernels only, inputs resident in HBM, ONE batch in one launch (the round-1 headline) -----------------
    resident = None
    if (rank == 0 or use_dist) and not args.no_resident:
        sig, sig_off, seqs, seq_off, _n = wl.batches[0]
        with al.batch_packed(sig, sig_off, seqs, seq_off) as b0:
            best = None
            for _ in range(3):
                b0.train() if args.mode == "train" else b0.align(True)
                tm0 = b0.timing()
                best = tm0 if best is None or tm0["ms_total"] < best["ms_total"] else best
            resident = best

    def side_record(al_, wl_, steps_, strict_, note, mode_=None):
        """a secondary workload / mode on the same terms as the headline: own warm-up, own timed region, own roofline"""
        mode_ = mode_ or args.mode
        al_.set_strict(strict_)
        mm = measure(al_, wl_, args, steps_, 1, 0, mode_, exch, sync)
        el, smp, rds = mm["elapsed"], mm["samples"], mm["reads"]
        if use_dist:
            t_ = torch.tensor([el, -float(smp)], device=coll_dev, dtype=torch.float64)
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)   # slowest rank; (samples are equal per rank: weak scaling)
            el = float(t_[0].item())
            smp, rds = smp * n_gpus, rds * n_gpus
        rf = roofline_of(mm, wl_.name, mode_, args, profile)
        return {"workload": wl_.name, "mode": mode_, "strict_mode": {"start": "ties"}.get(strict_, strict_), "steps": steps_, "value": round(smp / el / 1e6, 3), "unit": "Msamp/s",
                "ms_per_step": round(el * 1e3 / steps_, 3), "reads_per_s": round(rds / el, 1),
                "reads_per_batch": wl_.batches[0][4], "samples_per_batch": wl_.samples_of[0], "distinct_batches": wl_.n_batches,
                "strict_reads_per_step": mm["launches"]["reads_strict"] / max(1, mm["steps_done"]),
                "roofline_frac": rf["frac"], "bytes_per_cell": rf["bytes_per_cell"], "kernel": rf["kernel"].split(" (")[0],
                "wave_occupancy": rf["wave_occupancy"], "waiting_for_pages": rf["wave_time_share"]["waiting_for_pages"],
                "before_first_read": rf["wave_time_share"]["before_first_read"], "last_turn": rf["wave_time_share"]["last_turn"],
                "kernel_ms_total": rf["kernel_ms_total"], "cells_total": rf["cells_total"],
                "launches": rf["launches"], "batches_per_launch": rf["batches_per_launch"], "batches_in_flight": mm["depth"], "note": note}

    # ---- what bit-exactness costs: the same steps on the plain arithmetic (strict mode off), outside the headline region
    plain = None
    if args.mode == "align" and args.strict != "off" and not args.no_plain:
        plain = side_record(al, wl, min(args.steps, max(6, args.steps // 2)), "off",
                            "the table softplus alone: equal to the reference on every read without a structural tie, and on "
                            "3 397 of the 3 400 tie-bearing reads of tests/golden/g10_ties.npz")
        al.set_strict(args.strict)

    line = None
    if rank == 0:
        steps = max(1, m["steps_done"])
        kern, launches = m["kern"], m["launches"]
        ms_per_step = elapsed * 1e3 / max(1, args.steps)
        value = total_samples / elapsed / 1e6
        roofline = roofline_of(m, workload, args.mode, args, profile)
        what = "calc_probabilities=true" if args.mode == "align" else "train()"
        line = {
            "metric": "signal samples resquiggled/sec" if args.mode == "align" else "signal samples trained/sec (Baum-Welch statistics)",
            "value": round(value, 3), "unit": "Msamp/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{workload}: {wl.cfg['n_reads']} synthetic {wl.pore} reads x ~{wl.samples_of[0] // wl.batches[0][4]} samples per batch and GPU, "
                                   f"synthetic {wl.k}-mer model, --mode basic, band 400, {what}; host arrays -> H2D -> kernels -> D2H -> host arrays",
                       "reads_per_batch": wl.cfg["n_reads"], "samples_per_batch": wl.samples_of[0], "distinct_batches": wl.n_batches,
                       "batches_in_flight": m["depth"], "caller_memory": "pinned" if args.pinned_inputs else "pageable",
                       "strict_mode": {"start": "ties"}.get(args.strict, args.strict),
                       "parallelism": f"reads sharded x{n_gpus}" + ((", RCCL gather of segment rows to rank 0" if args.mode == "align" else ", RCCL all-reduce of pooled statistics") if use_dist else "")},
            "reads_per_s": round(total_reads / elapsed, 1),
            "reads_ok_last_batch": ok,
            "strict_reads_per_step": launches["reads_strict"] / steps,
            # certified arithmetic: lattice rows run in it per step, and how often a register of 64 sums fell back to the
            # restated glibc (7 registers per row)
            "certified_rows_per_step": launches["cert_rows"] / steps,
            "certificate_fallbacks_per_row": round(launches["cert_fallbacks"] / launches["cert_rows"], 5) if launches["cert_rows"] else None,
            # per step: the tickets' wave time / waves (resident queue) or launch time x share (one launch per batch)
            "kernel_ms_per_step": {k_: round(v / steps, 3) for k_, v in kern.items() if k_.startswith("ms_")},
            # ONE batch alone in one launch, inputs resident in HBM (the round-1 headline; no ratio is formed with it: a launch of
            # 1 024 reads on 1 024 waves lasts as long as its slowest read, the resident queue has no such tail)
            "kernel_resident_Msamp_s": round(resident["samples"] / resident["ms_total"] / 1e3, 3) if resident and resident["ms_total"] else None,
            "roofline": roofline,
        }
        if use_dist:
            line["rccl_ranks"] = exch.comm.n_ranks
            assert line["rccl_ranks"] == n_gpus, "the exchange must span every rank of the measurement"
            line["exchange"] = {
                "implementation": ("ONE-DEVICE REHEARSAL (DYN_BENCH_ONE_DEVICE: every rank on cuda:0, a NCCL_HOSTID per rank, RCCL's socket "
                                   "transport over loopback -- the exchange code is the real one, the transport is not xGMI: not a scaling "
                                   "measurement) of " if one_device else "") + "dyn_comm_* (dynamont_amd/csrc/rccl_comm.cpp): " +
                                  ("dyn_comm_gather_counts + dyn_comm_gather_rows -- 8-byte count all-gather, then one ncclSend per peer / ncclRecv per peer "
                                   "on rank 0 in one group, from the batch's device rows; rank 0 copies the gathered rows to page-locked host memory"
                                   if args.mode == "align" else "dyn_comm_allreduce_pooled -- ncclAllReduce(sum, double) in place on the device-resident (w, s1, s2)[4^k]"),
                "bracket_collectives": f"torch.distributed ({backend}): communicator id broadcast, barriers, the reduction of the ranks' clocks",
                "per_step_ms_rank0": {"median": round(sorted(m["exchange_ms"])[len(m["exchange_ms"]) // 2], 3), "max": round(max(m["exchange_ms"]), 3),
                                      "in_step_order": [round(x, 1) for x in m["exchange_ms"][:64]]} if m["exchange_ms"] else None,
                "rows_gathered_rank0": exch.rows_gathered if args.mode == "align" else None,
                "resident_queue": resident_with_exchange, "reserved_cus": reserved_cus if resident_with_exchange else 0, "observed": sessions_with_exchange,
                "rccl_channel_cap": rccl_channel_cap and {k: os.environ.get(k) for k in ("NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS", "NCCL_MAX_P2P_NCHANNELS")},
            }
            line["collective_backend"] = "rccl (dyn_comm_*)" + (", one-device rehearsal over loopback sockets" if one_device else "")
            if args.mode == "align":
                line["gather_lands_in"] = "rank0_pinned_host"
            line["per_rank_ms"] = [round(x, 2) for x in per_rank_ms]
        if cpu_out and os.path.exists(cpu_out):
            line["cpu_baseline"] = json.load(open(cpu_out))
        elif n_gpus == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = None
        if plain is not None:
            line["plain_arithmetic"] = plain
        assert line["n_gpus"] == args.gpus

    # ---- N = 1: the workloads the headline does not show (same clock discipline, each with its own roofline) -------------
    if n_gpus == 1 and not use_dist and args.mode == "align" and workload == "cfg2" and not args.reads:
        if not args.no_polya:
            # real direct-RNA reads start with their polyA tail: EVERY read then carries a structural tie and runs the certified
            # sweeps (segment.py:155-158 pads every RNA read; the tail follows the pad) -- the headline's synthetic reads: 26 %
            wl_p = Workload.get("cfg2_polya", rank, args, model_cache, workdir)
            line["cfg2_polya"] = side_record(al, wl_p, args.steps, args.strict,
                                             "cfg2 with 20-150 A's behind the pad: every read flagged, the certified sweeps' full price")
        if not args.no_scale_ref:
            # what `bench.py --gpus N` runs per rank (configs[3]'s per-GPU share), on this one GPU: the N = 1 point of the
            # scaling curve, like for like
            wl_s = Workload.get("cfg4_share", rank, args, model_cache, workdir)
            line["scale_ref"] = side_record(al, wl_s, max(4, min(8, args.steps)), args.strict,
                                            "configs[3]'s per-GPU share (4 096 reads per batch), the workload of every rank at N > 1, without the exchange")
        if not args.no_short:
            # reads narrower than the band (N < 400 columns: bw = N / 2): every row still occupies 448 lane slots, so the
            # cells-based fraction shows what a narrow band costs
            wl_n = Workload.get("short_reads", rank, args, model_cache, workdir)
            line["short_reads"] = side_record(al, wl_n, max(4, min(8, args.steps)), args.strict,
                                              "16 384 reads of 150-400 bases per batch: fewer lattice columns than the band is wide "
                                              "(half band = N / 2, NT_aligner_api.cpp:243); roofline_frac counts IN-BAND cells only")
            Workload.drop("short_reads", rank)
        if not args.no_train:
            # BASELINE configs[4]'s per-GPU share: the Baum-Welch statistics pass (log-domain backward sweep + posterior chain)
            wl_t = Workload.get("cfg5_share", rank, args, model_cache, workdir, n_distinct=4)
            line["train"] = side_record(al, wl_t, max(8, min(16, args.steps)), args.strict,
                                        "configs[4]'s per-GPU share: train() on 1 024 rna004 reads x ~20 k per batch, one launch per batch; "
                                        "16 algorithmic bytes per cell (write + read of one fp64 lattice row)", mode_="train")
            Workload.drop("cfg5_share", rank)
    al.close()  # (parks the lattice pool: the next handle takes it over instead of allocating its own)
    if exch is not None:
        exch.close()
    if rank == 0 and n_gpus == 1 and args.mode == "align" and not args.no_e2e and not use_dist:
        try:
            strict = {"start": "ties"}.get(args.strict, args.strict)
            big = None
            if args.e2e_large_reads > args.e2e_reads:
                big = run_e2e_cli(args.e2e_large_reads, workdir, strict, args.e2e_batch_reads, sub="e2e_large")
                shutil.rmtree(os.path.join(workdir, "e2e_large"), ignore_errors=True)
            line["e2e_cli"] = run_e2e_cli(args.e2e_reads, workdir, strict, args.e2e_batch_reads)
            if big is not None:
                line["e2e_cli"]["large"] = {k_: big[k_] for k_ in ("value", "unit", "wall_s", "reads", "reads_per_s", "samples", "input", "output", "error_lines")}
        except Exception as e:  # the headline stands on its own
            line.setdefault("e2e_cli", {})["error"] = f"{type(e).__name__}: {e}"
    if rank == 0 and n_gpus == 1 and not use_dist and args.mode == "align" and workload == "cfg2" and not args.reads and not args.no_cfg3:
        # BASELINE configs[2]: another pore, hence another handle. 4 096 DNA reads of 10 k-100 k samples per batch do not get an
        # arena each (1 024 arenas of the longest read: 550 GB): a PAGED session, posteriors in place (24.125 B per cell)
        try:
            Workload.drop("cfg2_polya", rank), Workload.drop("cfg4_share", rank)
            wl_3 = Workload.get("cfg3", rank, args, model_cache, workdir, n_distinct=1)
            al3 = Aligner(wl_3.model_path, wl_3.pore, mode="basic", band=400, device=local_rank)
            line["cfg3"] = side_record(al3, wl_3, max(4, min(8, args.steps)), args.strict,
                                       "BASELINE configs[2]: 4 096 dna_r10_400bps reads, 800-8 000 bases (10 k-100 k samples), page-starved: "
                                       "a paged session, posteriors in place")
            al3.close()
            Workload.drop("cfg3", rank)
        except Exception as e:  # the headline stands on its own
            line["cfg3"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and n_gpus == 1 and args.mode == "align" and not args.no_e2e and not use_dist and not args.no_cold and line.get("e2e_cli", {}).get("wall_s"):
        try:
            strict = {"start": "ties"}.get(args.strict, args.strict)
            if True:  # (second leg of e2e_cli.cold)
                import dynamont_amd
                warm = line["e2e_cli"]
                dynamont_amd.release_cached_memory()  # second leg: every byte this process held has JUST been freed
                after = run_e2e_cold(make_e2e_dataset(args.e2e_reads, workdir, "e2e"), strict, args.e2e_batch_reads,
                                     "right after this process freed its lattice pools (~130-250 GB): the child's allocations wait for the driver's scrubbing")
                cold = dict(cold_first or {})
                cold["timed"] = ("a fresh child process, Popen to exit: interpreter start-up, imports, library load, model parse, allocation of the "
                                 "lattice pool and every buffer, then everything the warm record times")
                for rec_ in (cold, after):
                    if rec_.get("wall_s"):
                        rec_["of_warm"] = round(warm["wall_s"] / rec_["wall_s"], 3)
                        rec_["output_bytes_equal_warm"] = rec_.get("output_bytes") == warm.get("output_bytes")
                cold["after_release"] = after
                warm["cold"] = cold
        except Exception as e:  # the headline stands on its own
            line["e2e_cli"]["cold"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        print(json.dumps(line))
    shutil.rmtree(workdir, ignore_errors=True)  # models, the cpu_baseline's files, the e2e datasets and their output
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
