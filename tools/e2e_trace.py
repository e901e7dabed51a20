#!/usr/bin/env python
"""GPU box: a timeline of the dynamont-resquiggle counterpart on the bench's e2e dataset (32 768 reads, .pod5 + BAM), second
run of the process (lattice pool parked by the first): when the model was loaded, when each batch was submitted, how long
the producer waited for a free slot, when the sink had consumed each batch, how long the drain took. Where is the wall
time when neither the front end nor the kernels account for it?"""
import os, sys, tempfile, time
sys.path.insert(0, "/root/repo")
from dynamont_amd import synth
from dynamont_amd.segmentation import segment as seg
import dynamont_amd._dynamont as dm

d = tempfile.mkdtemp(prefix="dyn_e2e_")
model = synth.write_model(os.path.join(d, "m9.model"), 9, seed=7, stdev=0.15)
_, mean, sd = synth.read_model_file(model)
reads = synth.make_reads(5, 4096, "rna004", mean, sd, 2000)
raw, bam, _ = synth.write_dataset(os.path.join(d, "in"), "ds", reads, "rna004", seed=1, container="pod5", replicate=8, basecalls="bam")
samples = sum(len(r.signal) for r in reads) * 8
del reads
args = ["-r", os.path.join(d, "in"), "-b", bam, "--mode", "basic", "-p", "rna004", "--model_path", model] + sys.argv[1:]
seg.main(args + ["-o", os.path.join(d, "out0.csv")])

T0 = [0.0]
ev = []
def stamp(name):
    ev.append((time.perf_counter() - T0[0], name))

NP = seg._NativePipeline
orig_init, orig_submit, orig_reap, orig_close = NP.__init__, NP.submit_columns, NP._reap, NP.close
wait = [0.0]
def init(self, *a, **k):
    stamp("aligner ready (model loaded)")
    orig_init(self, *a, **k)
    stamp("sink open")
def submit(self, *a, **k):
    t = time.perf_counter()
    orig_submit(self, *a, **k)
    ev.append((time.perf_counter() - T0[0], "submitted batch %d (call %.1f ms, of which waiting for a slot %.1f ms)" % (self.submitted - 1, (time.perf_counter() - t) * 1e3, wait[0] * 1e3)))
    wait[0] = 0.0
gpu = []
def reap(self, block_until=None):
    t = time.perf_counter()
    orig_reap(self, block_until)
    wait[0] += time.perf_counter() - t
orig_tclose = dm.AsyncBatch.close
def tclose(self):
    if getattr(self, "_h", None) and T0[0]:
        tm = self.timing()
        gpu.append((len(gpu), time.perf_counter() - T0[0], tm["ms_total"], tm["ms_dp"], tm["launch_share"], tm["wave_occupancy"]))
    orig_tclose(self)
dm.AsyncBatch.close = tclose
def close(self):
    stamp("close() called: all batches submitted")
    orig_close(self)
    stamp("close() returned: output complete")
NP.__init__, NP.submit_columns, NP._reap, NP.close = init, submit, reap, close

# a canary: one Python thread compressing rows of the FIRST run in a loop (ctypes releases the GIL) while the second run
# goes on -- does CPU work of this process slow down while the pipeline runs?
import ctypes as C, threading
from dynamont_amd import zstd_io
canary_data = zstd_io.decompress(open(os.path.join(d, "out0.csv.zst"), "rb").read())[:4 << 20]
zc = C.CDLL("libzstd.so.1")
zc.ZSTD_compressBound.restype = C.c_size_t; zc.ZSTD_compressBound.argtypes = [C.c_size_t]
zc.ZSTD_compress.restype = C.c_size_t; zc.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
canary_log, canary_stop = [], [False]
def canary():
    buf = C.create_string_buffer(zc.ZSTD_compressBound(len(canary_data)))
    while not canary_stop[0]:
        t = time.perf_counter()
        zc.ZSTD_compress(buf, len(buf), canary_data, len(canary_data), 3)
        canary_log.append((time.perf_counter() - T0[0], len(canary_data) / 1e6 / (time.perf_counter() - t)))
def thread_cpu():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{tid}/stat").read()
            comm = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            out[int(tid)] = (comm, (int(rest[11]) + int(rest[12])) / os.sysconf("SC_CLK_TCK"))
        except (OSError, ValueError):
            pass
    return out
tc0 = thread_cpu()
T0[0] = time.perf_counter()
can = threading.Thread(target=canary); can.start()
c0 = os.times()
seg.main(args + ["-o", os.path.join(d, "out1.csv")])
dt = time.perf_counter() - T0[0]
c1 = os.times()
tc1 = thread_cpu()
busy = sorted(((v[1] - tc0.get(t, (v[0], 0.0))[1], t, v[0]) for t, v in tc1.items()), reverse=True)
print("CPU seconds per thread during the run (top 40 of %d):" % len(busy), " ".join(f"{c:.2f}" for c, _, _ in busy[:40]))
time.sleep(0.5); canary_stop[0] = True; can.join()
print(f"process CPU time during the run: user {c1.user - c0.user:.2f} s + system {c1.system - c0.system:.2f} s = {(c1.user - c0.user + c1.system - c0.system) / dt:.1f} cores busy on average")
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip(), "| cpu.stat:", " ".join(open("/sys/fs/cgroup/cpu.stat").read().split()))
except OSError as e:
    print("no cgroup v2 cpu files:", e)
print("sched_getaffinity:", len(os.sched_getaffinity(0)))
print(f"second run: {dt:.3f} s -> {samples/dt/1e6:.1f} Msamp/s")
for t, name in ev:
    print(f"{t*1e3:9.1f} ms  {name}")
print("batch  seen-consumed-at   ms_total  ms_dp  launch_share  occupancy")
for k, t, a, b, c, o in gpu:
    print(f"{k:4d} {t*1e3:12.1f} {a:10.2f} {b:8.2f} {c:8.3f} {o:8.3f}")
print("GPU busy (sum ms_total x share): %.1f ms over %d batches seen; launches %.1f" % (sum(a * c for _, _, a, _, c, _ in gpu), len(gpu), sum(c for *_, c, _ in gpu)))

print("canary (one thread, zstd level 3 on 4 MB of rows, MB/s) during the run and for 0.5 s after it:")
print(" ".join(f"{t*1e3:.0f}ms:{v:.0f}" for t, v in canary_log))
# the same rows through the sink's zstd call sequence from plain Python threads of THIS process, pipeline idle
import ctypes as C, threading
from dynamont_amd import zstd_io
data = zstd_io.decompress(open(os.path.join(d, "out1.csv.zst"), "rb").read())[:256 << 20]
z = C.CDLL("libzstd.so.1")
for name, res, a in (("ZSTD_createCCtx", C.c_void_p, []), ("ZSTD_compressBegin", C.c_size_t, [C.c_void_p, C.c_int]),
                     ("ZSTD_compressContinue", C.c_size_t, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]),
                     ("ZSTD_invalidateRepCodes", None, [C.c_void_p]), ("ZSTD_compressBound", C.c_size_t, [C.c_size_t])):
    getattr(z, name).restype = res
    getattr(z, name).argtypes = a
pieces = [data[o:o + (4 << 20)] for o in range(0, len(data), 4 << 20)]
def sink_way(my):
    ctx = z.ZSTD_createCCtx()
    buf = C.create_string_buffer(z.ZSTD_compressBound(4 << 20) + 64)
    for k, piece in enumerate(my):
        z.ZSTD_compressBegin(ctx, 3)
        if k:
            z.ZSTD_compressContinue(ctx, buf, len(buf), None, 0)
            z.ZSTD_invalidateRepCodes(ctx)
        z.ZSTD_compressContinue(ctx, buf, len(buf), piece, len(piece))
for nt in (1, 8):
    th = [threading.Thread(target=sink_way, args=(pieces[i::nt],)) for i in range(nt)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; t1 = time.perf_counter()
    print(f"zstd level 3, the sink's call sequence on {len(data) >> 20} MB of this run's rows, {nt} Python threads, pipeline idle: {len(data) / 1e6 / (t1 - t0):.0f} MB/s in sum")
