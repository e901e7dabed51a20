#!/usr/bin/env python
"""GPU box: how fast is libzstd on the CSV this package writes? Runs the CLI on a small synthetic dataset, then times
ZSTD_compress (one thread) on 64 MB of the real rows at levels -1, 1, 2, 3, and in 4 MB pieces (what the sink's jobs are)."""
import ctypes as C, os, sys, tempfile, time
sys.path.insert(0, "/root/repo")
from dynamont_amd import synth, zstd_io
from dynamont_amd.segmentation import segment as seg
d = tempfile.mkdtemp(prefix="dyn_z_")
model = synth.write_model(os.path.join(d, "m9.model"), 9, seed=7, stdev=0.15)
_, mean, sd = synth.read_model_file(model)
reads = synth.make_reads(5, 2048, "rna004", mean, sd, 2000)
raw, bam, _ = synth.write_dataset(os.path.join(d, "in"), "ds", reads, "rna004", seed=1, container="pod5", basecalls="bam")
seg.main(["-r", os.path.join(d, "in"), "-b", bam, "--mode", "basic", "-p", "rna004", "--model_path", model, "-o", os.path.join(d, "out.csv")])
data = zstd_io.decompress(open(os.path.join(d, "out.csv.zst"), "rb").read())[:64 << 20]
print(len(data) / 1e6, "MB of rows; first rows:")
print(data[:400].decode())
z = C.CDLL("libzstd.so.1")
z.ZSTD_versionString.restype = C.c_char_p
print("libzstd", z.ZSTD_versionString().decode())
z.ZSTD_compressBound.restype = C.c_size_t; z.ZSTD_compressBound.argtypes = [C.c_size_t]
z.ZSTD_compress.restype = C.c_size_t; z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
cap = z.ZSTD_compressBound(len(data)); out = C.create_string_buffer(cap)
for lvl in (-1, 1, 2, 3):
    t0 = time.perf_counter(); n = z.ZSTD_compress(out, cap, data, len(data), lvl); t1 = time.perf_counter()
    print("level", lvl, "ratio", round(n / len(data), 3), round(len(data) / 1e6 / (t1 - t0)), "MB/s (one call)")
    t0 = time.perf_counter(); tot = 0
    for o in range(0, len(data), 4 << 20):
        piece = data[o:o + (4 << 20)]
        tot += z.ZSTD_compress(out, cap, piece, len(piece), lvl)
    t1 = time.perf_counter()
    print("        in 4 MB pieces: ratio", round(tot / len(data), 3), round(len(data) / 1e6 / (t1 - t0)), "MB/s")

# the sink's own call sequence per 4 MB job (csv_sink.cpp compress_loop), one thread, then 8 threads at once
import threading
for name, res, args in (("ZSTD_createCCtx", C.c_void_p, []), ("ZSTD_compressBegin", C.c_size_t, [C.c_void_p, C.c_int]),
                        ("ZSTD_compressContinue", C.c_size_t, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]),
                        ("ZSTD_invalidateRepCodes", None, [C.c_void_p])):
    getattr(z, name).restype = res
    getattr(z, name).argtypes = args
pieces = [data[o:o + (4 << 20)] for o in range(0, len(data), 4 << 20)]
def sink_way(my):
    ctx = z.ZSTD_createCCtx()
    buf = C.create_string_buffer(z.ZSTD_compressBound(4 << 20) + 64)
    tot = 0
    for k, piece in enumerate(my):
        z.ZSTD_compressBegin(ctx, 3)
        if k:
            z.ZSTD_compressContinue(ctx, buf, len(buf), None, 0)
            z.ZSTD_invalidateRepCodes(ctx)
        tot += z.ZSTD_compressContinue(ctx, buf, len(buf), piece, len(piece))
    return tot
t0 = time.perf_counter(); tot = sink_way(pieces); t1 = time.perf_counter()
print("sink's sequence, one thread: ratio", round(tot / len(data), 3), round(len(data) / 1e6 / (t1 - t0)), "MB/s")
for nt in (8, 16):
    th = [threading.Thread(target=sink_way, args=(pieces,)) for _ in range(nt)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; t1 = time.perf_counter()
    print(f"sink's sequence, {nt} threads each doing all pieces: {nt * len(data) / 1e6 / (t1 - t0):.0f} MB/s in sum, {len(data) / 1e6 / (t1 - t0):.0f} per thread")
