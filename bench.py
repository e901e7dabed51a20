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
wall time. The kernel-only rate with inputs resident in HBM is reported as the secondary
`kernel_resident_Msamp_s` -- of ONE batch in one launch: the engine merges tickets that wait while the GPU is busy into one
launch (include/dynamont_mi.h), and a launch of two or three 1 024-read batches balances better than one of a single batch,
so `pipeline_efficiency` can exceed 1. `roofline` prices the launches as they ran: `cells_per_launch`, `avg_launch_ms` and
`batches_per_launch` describe the merged launches (HIP events per launch, each counted once through the tickets'
`launch_share`), which is what a rocprofv3 kernel trace of the same command shows.

N = 1 workload: BASELINE.json configs[1] -- 1 024 synthetic RNA004 reads x ~20 k samples per batch,
synthetic 9-mer model, --mode basic, band 400. N > 1: configs[3]'s per-GPU share, 4 096 reads per
rank and batch, RCCL gather of the segment rows to rank 0 (weak scaling). --mode train: configs[4]'s
per-GPU share (1 024 reads per rank and batch), RCCL all-reduce of the pooled statistics.

N > 1, what the timed region contains per step and rank: the batch through the drop-in boundary as above, then
the rank's segment rows (still resident in HBM) into a send buffer, an ASYNCHRONOUS gather to rank 0 over RCCL that
overlaps the next batch, and on rank 0 the copy of all ranks' rows into pinned host memory -- the line says
"gather_lands_in": "rank0_pinned_host". The clock stops after the last gather and its host copy have completed.

N = 1 also carries `e2e_cli`: the dynamont-resquiggle counterpart itself on a synthetic .pod5 + BAM dataset of 32 768 reads
(configs[3]'s read count), and `e2e_cli.large` on 131 072 -- file in, compressed CSV out, in this process (run_e2e_cli).

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

WORKLOADS = {
    # name -> (synth config, reads per batch, default number of distinct batches)
    "cfg1": ("cfg1", None, 8),
    "cfg2": ("cfg2", None, 8),
    "cfg2_small": ("cfg2", 64, 8),
    "cfg2_polya": ("cfg2_polya", None, 8),
    "cfg3": ("cfg3", None, 2),
    "cfg4_share": ("cfg4", 4096, 4),
    "cfg5_share": ("cfg5", 1024, 8),
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


def run_e2e_cli(n_reads: int, workdir: str, strict: str, batch_reads: int = 0) -> dict:
    """north_star's "throughput on synthetic pod5+bam": the dynamont-resquiggle counterpart end to end, in this process.
    A .pod5 file (VBZ-compressed int16 chunks) and an unaligned BAM (dorado's tags) written by synth.write_dataset ->
    dynamont_amd.segmentation.segment.main -> out.csv.zst (reference: src/dynamont/segmentation/segment.py:261-371).
    Timed: model load, BAM parse, pod5 open + VBZ decode, pA calibration + normalisation + Hampel on the device, the
    DP, CSV formatting, zstd, file write. Not timed: interpreter start-up, dataset generation; the lattice pool and the
    batch buffers are the ones the bench's own handle has just parked (a fresh process allocates them: ~1 s on clean VRAM)."""
    from dynamont_amd import synth
    from dynamont_amd.segmentation import segment as seg
    d = os.path.join(workdir, "e2e")
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
    t_gen = time.perf_counter() - t0
    out = os.path.join(d, "out.csv")
    t0 = time.perf_counter()
    seg.main(["-r", os.path.join(d, "in"), "-b", bam, "-o", out, "--mode", "basic", "-p", "rna004", "--model_path", model,
              "--strict-ties", strict] + (["--batch-reads", str(batch_reads)] if batch_reads else []))
    dt = time.perf_counter() - t0
    err = out + ".errors" if os.path.exists(out + ".errors") else None
    rec = {"value": round(samples / dt / 1e6, 3), "unit": "Msamp/s", "wall_s": round(dt, 3), "reads": distinct * rep,
           "reads_per_s": round(distinct * rep / dt, 1), "samples": samples,
           "input": f"{os.path.basename(raw)} ({os.path.getsize(raw) / 1e6:.0f} MB, VBZ) + {os.path.basename(bam)} ({os.path.getsize(bam) / 1e6:.1f} MB); "
                    f"{distinct} distinct synthetic rna004 reads x {rep}, written by synth.write_dataset in {t_gen:.1f} s (not timed)",
           "output": (f"out.csv.zst, {os.path.getsize(out + '.zst') / 1e6:.1f} MB (one zstd frame, level 3) holding "
                      f"{seg.LAST_RUN.get('csv_bytes', 0) / 1e6:.0f} MB of CSV rows") if os.path.exists(out + ".zst") else None,
           "batches_in_flight": seg.LAST_RUN.get("depth"), "compress_threads": seg.LAST_RUN.get("compress_threads"),
           "error_lines": sum(1 for _ in open(err)) if err else 0, "strict_mode": strict,
           "batch_reads": batch_reads or "CLI default",
           "timed": "segment.main: model load, BAM parse, pod5 VBZ decode, device preprocessing, DP, CSV format, zstd, write",
           "not_timed": "interpreter start-up; allocation of the lattice pool and of the batch buffers (those the bench's own handle has just parked are taken over)"}
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

    from dynamont_amd import synth
    cfgname, per_batch, n_batches = WORKLOADS[workload]
    cfg = dict(synth.CONFIGS[cfgname])
    if per_batch:
        cfg["n_reads"] = per_batch
    if args.reads:
        cfg["n_reads"] = args.reads
    n_batches = max(1, args.batches or n_batches)
    pore = cfg["pore"]
    _, rna, k = synth.PORES[pore]
    workdir = tempfile.mkdtemp(prefix=f"dyn_bench_r{rank}_")
    model_path = synth.write_model(os.path.join(workdir, f"syn{k}.model"), k, seed=7, stdev=0.25 if k == 5 else 0.15)

    cpu_proc = cpu_out = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        cpu_proc, cpu_out = start_cpu_baseline(args, workload, model_path, workdir)
        # the CPU sample uses every host core: let it finish before timing the GPU
        _, err = cpu_proc.communicate()
        if cpu_proc.returncode != 0:
            print("cpu baseline failed:\n" + err.decode(errors="replace")[-2000:], file=sys.stderr)

    import numpy as np
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: dynamont_amd has no CPU compute path")
    # Rehearsal hooks (control-flow checks on a 1-GPU box; never set by the driver):
    #   DYN_BENCH_BACKEND=gloo   use gloo instead of nccl/RCCL
    #   DYN_BENCH_ONE_DEVICE=1   every rank uses cuda:0
    #   DYN_BENCH_FORCE_DIST=1   run the collectives even with a single rank (exercises RCCL on one GPU)
    backend = os.environ.get("DYN_BENCH_BACKEND", "nccl")
    if os.environ.get("DYN_BENCH_ONE_DEVICE"):
        local_rank = 0
    elif torch.cuda.device_count() < n_gpus:
        raise SystemExit(f"bench.py: --gpus {n_gpus} but only {torch.cuda.device_count()} device(s) visible; refusing to measure fewer GPUs than asked for")
    torch.cuda.set_device(local_rank)
    use_dist = n_gpus > 1 or bool(os.environ.get("DYN_BENCH_FORCE_DIST"))
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
    from dynamont_amd._dynamont import AlignBatchResult, pinned_empty
    assert "torch" in sys.modules and _native._lib is None, "torch must be imported before libdynamont_mi.so is loaded"

    # ---- the stream of distinct batches, in caller-owned host arrays ----------------------------
    _, mean, sd = synth.read_model_file(model_path)
    batches = []
    for j in range(n_batches):
        reads = synth.make_reads(cfg["seed"] + 1000 * rank + 100003 * j, cfg["n_reads"], pore, mean, sd, cfg["n_bases"], polya=cfg.get("polya"))
        sig, sig_off, seqs, seq_off = synth.pack_reads(reads)
        if args.pinned_inputs:
            ps = pinned_empty(sig.size, np.float64)
            ps[:] = sig
            sig = ps
        batches.append((sig, sig_off, seqs, seq_off, len(reads)))
        del reads
    samples_of = [int(b[1][-1]) for b in batches]

    al = Aligner(model_path, pore, mode="basic", band=400, device=local_rank)
    al.set_strict(args.strict)
    depth = max(1, args.depth)
    if max(b[4] for b in batches) > 1536:
        # batches of more reads than the engine merges (its cap: three reads per wave slot, two tickets at least) gain nothing
        # from waiting tickets; each ticket in flight holds its samples twice (pinned staging, device): 4 overlap everything
        depth = min(depth, 4)
    free_results: list = []  # result objects are reused: fresh 50 MB arrays per batch would be page-faulted in every time

    # ---- N > 1: fixed-size buffers for the gather (sizes differ per rank and batch), two sets: the gather of step k
    # is in flight while step k+1 fills the other set
    NBUF = 2
    send_bufs = gather_sets = host_sets = None
    gather_pending = [None] * NBUF
    if use_dist and args.mode == "align":
        cap_local = max(al.segment_capacity(b[3]) for b in batches)
        t = torch.tensor([cap_local], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        cap_max = int(t.item())
        # [16-byte header: valid bytes][rows ...]
        send_bufs = [torch.zeros(16 + cap_max * 16, dtype=torch.uint8, device=dev) for _ in range(NBUF)]
        if rank == 0:
            gdev = dev if backend == "nccl" else "cpu"
            gather_sets = [[torch.empty(16 + cap_max * 16, dtype=torch.uint8, device=gdev) for _ in range(n_gpus)] for _ in range(NBUF)]
            if backend == "nccl":  # where the gathered rows end up: page-locked host memory of rank 0
                host_sets = [[torch.empty(16 + cap_max * 16, dtype=torch.uint8).pin_memory() for _ in range(n_gpus)] for _ in range(NBUF)]

    def drain_gather(slot):
        """complete the gather that last used this buffer set; rank 0 then brings the rows to host memory"""
        work = gather_pending[slot]
        if work is None:
            return
        gather_pending[slot] = None
        work.wait()  # NCCL: the current stream waits for the collective
        if rank == 0 and host_sets is not None:
            for r in range(n_gpus):
                host_sets[slot][r].copy_(gather_sets[slot][r], non_blocking=True)

    def wrap(ptr, nbytes, typestr="|u1", count=None):
        class _A:
            __cuda_array_interface__ = {"shape": (count if count is not None else nbytes,), "typestr": typestr,
                                        "data": (ptr, False), "version": 2}
        return torch.as_tensor(_A(), device=dev)

    kern = collections.Counter()
    launches = collections.Counter()
    done_steps = [0]
    gather_step = [0]

    def submit(j):
        sig, sig_off, seqs, seq_off, _n = batches[j % n_batches]
        if args.mode == "train":
            return al.train_async(sig, sig_off, seqs, seq_off, pooled=False, emissions=False)
        out = free_results.pop() if free_results else None
        return al.align_async(sig, sig_off, seqs, seq_off, True, out=out)

    def finish(t, timed):
        res = t.wait()  # results are in host arrays from here on
        if use_dist:
            if args.mode == "train":  # config 5: sum all-reduce of the pooled sufficient statistics (3 * 4^k doubles)
                ptr, cnt = t.device_pooled()
                pooled_t = wrap(ptr, cnt * 8, "<f8", cnt)
                if backend == "nccl":
                    dist.all_reduce(pooled_t, op=dist.ReduceOp.SUM)
                    torch.cuda.current_stream().synchronize()  # the batch's buffers are recycled after close()
                else:
                    h = pooled_t.cpu()
                    dist.all_reduce(h, op=dist.ReduceOp.SUM)
            else:  # config 4: gather of the segment rows to rank 0, device to device over xGMI, asynchronous
                slot = gather_step[0] % NBUF
                gather_step[0] += 1
                drain_gather(slot)
                ptr, cap, _ = t.device_results()
                nbytes = cap * 16
                sb = send_bufs[slot]
                sb[:8].copy_(torch.tensor([nbytes], dtype=torch.int64).view(torch.uint8))
                if nbytes:
                    sb[16:16 + nbytes].copy_(wrap(ptr, nbytes))
                torch.cuda.current_stream().synchronize()  # the batch's device buffers are recycled after close()
                if backend == "nccl":
                    gather_pending[slot] = dist.gather(sb, gather_sets[slot] if rank == 0 else None, dst=0, async_op=True)
                else:  # gloo rehearsal: host hop
                    gather_pending[slot] = dist.gather(sb.cpu(), gather_sets[slot] if rank == 0 else None, dst=0, async_op=True)
        if timed:
            # tickets that waited together are merged into ONE launch by the engine: each reports that launch's timing and
            # its share of it (dyn_timing.launch_share), so sums over tickets count every launch once
            tm = t.timing()
            share = tm["launch_share"]
            if tm["launches"] == 0 and tm["reads_ok"]:
                # a ticket of the RESIDENT read queue: no launch of its own -- ms_* are its reads' wave time / waves, which add up
                # over tickets; the session kernels' own durations come from al.session_stats() around the timed region
                share = 1.0
                launches["resident_tickets"] += 1
            else:
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
        if args.mode == "align":
            free_results.append(res)
        return res

    def run(first, count, timed):
        q = collections.deque()
        last = None
        for s in range(first, first + count):
            if len(q) >= depth:
                last = finish(q.popleft(), timed)
            q.append(submit(s))
        while q:
            last = finish(q.popleft(), timed)
        for slot in range(NBUF):  # the gathers still in flight (and rank 0's host copies) belong to these steps
            drain_gather(slot)
        return last

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    run(0, args.warmup, False)
    sync()
    sess0 = al.session_stats()  # (closes the warm-up's session: the timed region starts with an idle pipeline and no resident wave)
    t0 = time.perf_counter()
    last = run(args.warmup, args.steps, True)
    sync()
    elapsed = time.perf_counter() - t0
    sess1 = al.session_stats()
    sess = {k_: sess1[k_] - sess0[k_] for k_ in sess1 if k_ != "wave_occupancy"}

    n_samples = sum(samples_of[s % n_batches] for s in range(args.warmup, args.warmup + args.steps))
    n_reads_done = sum(batches[s % n_batches][4] for s in range(args.warmup, args.warmup + args.steps))
    per_rank_ms = [elapsed * 1e3]
    if use_dist:
        t = torch.tensor([elapsed], device=coll_dev, dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(n_gpus)]
        dist.all_gather(allt, t)
        per_rank_ms = [float(x.item()) * 1e3 for x in allt]
        elapsed = max(per_rank_ms) / 1e3
        tot = torch.tensor([n_samples, n_reads_done], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_samples, total_reads = float(tot[0].item()), float(tot[1].item())
    else:
        total_samples, total_reads = float(n_samples), float(n_reads_done)
    ok = int((last.status == 0).sum()) if last is not None else 0

    # ---- secondary: kernels only, inputs resident in HBM (the round-1 headline) -----------------
    resident = None
    if (rank == 0 or use_dist) and not args.no_resident:
        sig, sig_off, seqs, seq_off, _n = batches[0]
        with al.batch_packed(sig, sig_off, seqs, seq_off) as b0:
            best = None
            for _ in range(3):
                b0.train() if args.mode == "train" else b0.align(True)
                tm0 = b0.timing()
                best = tm0 if best is None or tm0["ms_total"] < best["ms_total"] else best
            resident = best

    # ---- what bit-exactness costs: the same steps on the plain arithmetic (strict mode off), outside the headline region
    plain = None
    if args.mode == "align" and args.strict != "off" and not args.no_plain:
        keep = (collections.Counter(kern), collections.Counter(launches), done_steps[0])
        al.set_strict("off")
        p_steps = min(6, args.steps)
        run(args.warmup + args.steps, 1, False)
        sync()
        sess1 = al.session_stats()  # baseline of the plain leg (its warm-up step's session is closed and counted here)
        tp = time.perf_counter()
        run(args.warmup + args.steps + 1, p_steps, True)
        sync()
        p_el = time.perf_counter() - tp
        p_samples = sum(samples_of[s % n_batches] for s in range(args.warmup + args.steps + 1, args.warmup + args.steps + 1 + p_steps))
        if use_dist:
            t = torch.tensor([p_el, -float(p_samples)], device=coll_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)   # slowest rank; (samples are equal per rank: weak scaling)
            p_el = float(t[0].item())
            p_samples *= n_gpus
        p_sess = al.session_stats()
        p_launch = max(1e-9, launches["launches"] - keep[1]["launches"] + (p_sess["sessions"] - sess1["sessions"]))
        p_cells = launches["cells"] - keep[1]["cells"]
        p_ms = launches["classic_ms_dp"] - keep[1]["classic_ms_dp"] + (p_sess["ms"] - sess1["ms"])
        plain = {"strict_mode": "off", "steps": p_steps, "value": round(p_samples / p_el / 1e6, 3), "unit": "Msamp/s",
                 "ms_per_step": round(p_el * 1e3 / p_steps, 3),
                 "avg_launch_ms": round(p_ms / p_launch, 3), "batches_per_launch": round(p_steps / p_launch, 3),
                 "cells": p_cells, "kernel_ms": round(p_ms, 3),
                 "note": "the table softplus alone: equal to the reference on every read without a structural tie, and on "
                         "3 397 of the 3 400 tie-bearing reads of tests/golden/g10_ties.npz"}
        al.set_strict(args.strict)
        kern.clear(); kern.update(keep[0])
        launches.clear(); launches.update(keep[1])
        done_steps[0] = keep[2]

    line = None
    if rank == 0:
        steps = max(1, done_steps[0])
        ms_per_step = elapsed * 1e3 / max(1, args.steps)
        value = total_samples / elapsed / 1e6
        cells_total = launches["cells"]
        # launches of the dominant kernel in the timed region: one per (merged) batch -- or, with the resident read queue, one
        # per SESSION: the waves stay on the chip across batches. Duration: HIP events on the kernel's own stream.
        n_launch = max(1e-9, launches["launches"] + sess["sessions"])
        kernel_ms_total = launches["classic_ms_dp"] + sess["ms"]
        ms_dp = kernel_ms_total / n_launch
        cells_per_launch = cells_total / n_launch
        inplace = bool(launches["lp_inplace"])
        traffic = load_traffic() or {}
        if args.mode == "train":
            kname = "k_read_queue<JOB_TRAIN> (per read: backward sweep, forward sweep + Baum-Welch statistics)"
            bpc_f, tkey = KTRAIN_BYTES_PER_CELL, "train"
        else:
            kname = ("k_session (resident waves; per read: backward, forward + posterior + posterior-Viterbi, traceback)" if sess["sessions"] and not launches["launches"]
                     else "k_read_queue<JOB_ALIGN%s> (per read: backward, forward + posterior + posterior-Viterbi, traceback)" % ("_INPLACE" if inplace else ""))
            bpc_f, tkey = (KFWD_INPLACE_BYTES_PER_CELL if inplace else KFWD_BYTES_PER_CELL), "align"
        bpc = KBWD_BYTES_PER_CELL + bpc_f
        # PMC traffic is only quoted for the workload (and layout) it was measured on
        tinfo = traffic.get(tkey) or {}
        tbytes = tinfo.get("bytes_per_launch") if (workload == tinfo.get("workload") and not args.reads and not inplace) else None
        if tbytes and tinfo.get("cells_per_launch"):
            # the PMC passes measured launches of ONE batch; a merged launch moves the same bytes per cell
            tbytes = int(round(tbytes * cells_per_launch / tinfo["cells_per_launch"]))
        achieved = cells_per_launch * bpc / (ms_dp * 1e-3) / 1e9 if ms_dp else 0.0
        share = lambda key: kern[key] / kern["ms_dp"] if kern["ms_dp"] else 0.0
        roofline = {
            "bound": "hbm", "kernel": kname,
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "traffic": tbytes,
            # NOT a counter of this run: the committed rocprofv3 --pmc passes over the same command (plain arithmetic: the
            # certified rows move the same bytes), quoted only for the workload and layout they were measured on
            "traffic_source": ("profiles/traffic.json (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE per lattice cell, round %s) x cells_per_launch" % traffic.get("round")) if tbytes else None,
            "traffic_over_algorithmic": round(tbytes / (cells_per_launch * bpc), 3) if tbytes else None,
            "bytes_per_cell": bpc, "cells_per_launch": round(cells_per_launch), "launches": round(n_launch, 3),
            # totals of the timed region (what a rocprofv3 kernel trace of the same tickets adds up to: launches vary in size)
            "cells_total": int(cells_total), "kernel_ms_total": round(kernel_ms_total, 3),
            "avg_launch_ms": round(ms_dp, 3),
            # share of wave time per phase (device cycle counters) and how well the persistent waves were kept busy
            "wave_time_share": {"backward": round(share("ms_backward"), 4), "forward": round(share("ms_forward"), 4),
                                "waiting_for_pages": round(kern["wave_wait_share"] / n_launch, 4)},
            # the two sweeps apart: their algorithmic bytes over their share of the launch (they overlap in time only
            # when waves are out of phase, i.e. in batches of more reads than waves)
            "phases": {"backward_sweep": {"bytes_per_cell": KBWD_BYTES_PER_CELL, "ms": round(ms_dp * share("ms_backward"), 3),
                                          "frac": round(cells_per_launch * KBWD_BYTES_PER_CELL / (ms_dp * share("ms_backward") * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if share("ms_backward") else None},
                       "forward_sweep": {"bytes_per_cell": bpc_f, "ms": round(ms_dp * share("ms_forward"), 3),
                                         "frac": round(cells_per_launch * bpc_f / (ms_dp * share("ms_forward") * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if share("ms_forward") else None}},
            # classic launches: sum of wave lifetimes / (waves x longest lifetime), averaged over launches; resident queue: busy /
            # lifetime wave-cycles of the sessions (a wave is busy from claiming a read to releasing its results)
            "wave_occupancy": round((kern["wave_occupancy"] + (sess["wave_cycles_busy"] / sess["wave_cycles_life"] * sess["sessions"] if sess["wave_cycles_life"] else 0.0)) / n_launch, 4),
            "resident_queue": ({"sessions": sess["sessions"], "tickets": sess["tickets"], "reads": sess["reads"], "ms": round(sess["ms"], 3),
                                "wave_cycles_busy": sess["wave_cycles_busy"], "wave_cycles_idle": sess["wave_cycles_idle"],
                                "wave_cycles_life": sess["wave_cycles_life"], "aborted": sess["aborted"]} if sess["sessions"] else None),
            # tickets per launch: batches that were waiting while the GPU was busy share one launch (one queue balances
            # what one read per wave cannot)
            "batches_per_launch": round(steps / n_launch, 3),
            "page_pool": {"pages": launches["pool_pages"], "rows_per_page": launches["page_rows"],
                          "reads_with_reserved_pages": launches["n_static"], "waves": launches["n_waves"]},
        }
        what = "calc_probabilities=true" if args.mode == "align" else "train()"
        line = {
            "metric": "signal samples resquiggled/sec" if args.mode == "align" else "signal samples trained/sec (Baum-Welch statistics)",
            "value": round(value, 3), "unit": "Msamp/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{workload}: {cfg['n_reads']} synthetic {pore} reads x ~{samples_of[0] // batches[0][4]} samples per batch and GPU, "
                                   f"synthetic {k}-mer model, --mode basic, band 400, {what}; host arrays -> H2D -> kernels -> D2H -> host arrays",
                       "reads_per_batch": cfg["n_reads"], "samples_per_batch": samples_of[0], "distinct_batches": n_batches,
                       "batches_in_flight": depth, "caller_memory": "pinned" if args.pinned_inputs else "pageable",
                       "strict_mode": {"start": "ties"}.get(args.strict, args.strict),
                       "parallelism": f"reads sharded x{n_gpus}" + ((", RCCL gather of segment rows to rank 0" if args.mode == "align" else ", RCCL all-reduce of pooled statistics") if use_dist else "")},
            "reads_per_s": round(total_reads / elapsed, 1),
            "reads_ok_last_batch": ok,
            "strict_reads_per_step": launches["reads_strict"] / steps,
            # certified arithmetic: lattice rows run in it per step, and how often a register of 64 sums fell back to the
            # restated glibc (7 registers per row)
            "certified_rows_per_step": launches["cert_rows"] / steps,
            "certificate_fallbacks_per_row": round(launches["cert_fallbacks"] / launches["cert_rows"], 5) if launches["cert_rows"] else None,
            "kernel_ms_per_step": {k_: round(v / steps, 3) for k_, v in kern.items() if k_.startswith("ms_")},
            "kernel_resident_Msamp_s": round(resident["samples"] / resident["ms_total"] / 1e3, 3) if resident and resident["ms_total"] else None,
            "pipeline_efficiency": round((total_samples / n_gpus / elapsed / 1e6) / (resident["samples"] / resident["ms_total"] / 1e3), 4) if resident and resident["ms_total"] else None,
            "roofline": roofline,
        }
        if use_dist:
            line["rccl_ranks"] = dist.get_world_size()
            line["collective_backend"] = backend
            if args.mode == "align":
                line["gather_lands_in"] = "rank0_pinned_host" if backend == "nccl" else "rank0_host"
            line["per_rank_ms"] = [round(x, 2) for x in per_rank_ms]
        if cpu_out and os.path.exists(cpu_out):
            line["cpu_baseline"] = json.load(open(cpu_out))
        elif n_gpus == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = None
        if plain is not None:
            # the same algorithmic bytes over the plain kernel's launch time: what `roofline.frac` was before bit-exact
            # borders became the default (in cfg2 every wave holds ONE read, so the default launch lasts as long as its
            # slowest, certified read)
            if plain.get("kernel_ms"):
                plain["roofline_frac"] = round(plain.pop("cells") * bpc / (plain.pop("kernel_ms") * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
            line["plain_arithmetic"] = plain
        assert line["n_gpus"] == args.gpus
    al.close()  # (parks the lattice pool: the CLI's handle below takes it over instead of allocating its own)
    if rank == 0 and n_gpus == 1 and args.mode == "align" and not args.no_e2e and not use_dist:
        try:
            strict = {"start": "ties"}.get(args.strict, args.strict)
            line["e2e_cli"] = run_e2e_cli(args.e2e_reads, workdir, strict, args.e2e_batch_reads)
            if args.e2e_large_reads > args.e2e_reads:
                shutil.rmtree(os.path.join(workdir, "e2e"), ignore_errors=True)
                big = run_e2e_cli(args.e2e_large_reads, workdir, strict, args.e2e_batch_reads)
                line["e2e_cli"]["large"] = {k_: big[k_] for k_ in ("value", "unit", "wall_s", "reads", "reads_per_s", "samples", "input", "output", "error_lines")}
        except Exception as e:  # the headline stands on its own
            line.setdefault("e2e_cli", {})["error"] = f"{type(e).__name__}: {e}"
    if rank == 0:
        print(json.dumps(line))
    shutil.rmtree(workdir, ignore_errors=True)  # models, the cpu_baseline's files, the e2e datasets and their output
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
