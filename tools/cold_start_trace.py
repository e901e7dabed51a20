#!/usr/bin/env python
"""GPU box: where does a COLD run of the dynamont-resquiggle counterpart go? The bench's e2e dataset (32 768 reads, .pod5 +
BAM) through `python -m dynamont_amd.segmentation.segment` in a fresh child process, with the CLI's timeline
(DYN_CLI_TRACE) and the library's host trace (DYN_TRACE_HOST) on stderr; run twice (the second start finds the files in the
page cache and the VRAM already scrubbed by the first). Usage: python tools/cold_start_trace.py [n_reads] [extra CLI args]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamont_amd import synth

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
d = tempfile.mkdtemp(prefix="dyn_cold_")
model = synth.write_model(os.path.join(d, "m9.model"), 9, seed=7, stdev=0.15)
_, mean, sd = synth.read_model_file(model)
distinct = min(n_reads, 4096)
reads = synth.make_reads(5, distinct, "rna004", mean, sd, 2000)
raw, bam, _ = synth.write_dataset(os.path.join(d, "in"), "ds", reads, "rna004", seed=1, container="pod5", replicate=max(1, n_reads // distinct), basecalls="bam")
samples = sum(len(r.signal) for r in reads) * max(1, n_reads // distinct)
del reads
cli = ["-r", os.path.join(d, "in"), "-b", bam, "--mode", "basic", "-p", "rna004", "--model_path", model] + sys.argv[2:]
RUNS = int(os.environ.get("COLD_RUNS", "3"))          # COLD_RUNS=1 COLD_TRACE_ALL=1: one run on the VRAM as found, every trace line
TRACE_ALL = bool(os.environ.get("COLD_TRACE_ALL"))
for run in range(RUNS):
    env = dict(os.environ, PYTHONPATH=ROOT, DYN_CLI_TRACE="1", DYN_CLI_T0=repr(time.time()))
    if run == 2 or TRACE_ALL:
        env["DYN_TRACE_HOST"] = "1"
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, "-m", "dynamont_amd.segmentation.segment"] + cli + ["-o", os.path.join(d, "out%d.csv" % run)], env=env,
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    dt = time.perf_counter() - t0
    print("==== run %d: rc %d, %.3f s wall = %.1f Msamp/s" % (run, r.returncode, dt, samples / dt / 1e6))
    lines = r.stderr.splitlines()
    keep = lines if TRACE_ALL else [l for l in lines if l.startswith("[cli")] if run < 2 else [l for l in lines if l.startswith("[cli") or "create:" in l or "destroy" in l or "publish 1" in l or "publish 2" in l or "publish 3" in l][:60]
    print("\n".join(keep))
