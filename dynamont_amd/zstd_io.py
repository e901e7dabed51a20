"""zstd level-3 writers for the CSV output (reference: ``zstd.ZstdCompressor(level=3)
.stream_writer``, segment.py:74-79): :class:`ParallelZstdWriter` -- chunks compressed concurrently and
written either as ONE frame (default of the CLIs: the job scheme of zstd's own multi-threaded mode, over
the system ``libzstd.so.1`` through ctypes) or, opt-in, as consecutive independent frames -- and
:class:`ZstdWriter` (one frame, one thread; the ``zstandard`` package when it is installed).

Why one frame is the default: python-zstandard's default readers (``stream_reader``, ``zstd.open``,
``decompressobj()``, one-shot ``decompress``) stop at the end of the FIRST frame, so a consumer such as
``pandas.read_csv("x.csv.zst")`` would silently see only the first chunk of a multi-frame file."""
from __future__ import annotations

import ctypes as C
import ctypes.util

try:  # pragma: no cover - not installed in the build image
    import zstandard as _zstandard
except Exception:  # noqa: BLE001
    _zstandard = None


class _InBuf(C.Structure):
    _fields_ = [("src", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]


class _OutBuf(C.Structure):
    _fields_ = [("dst", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]


_ZSTD_c_compressionLevel = 100
_ZSTD_c_nbWorkers = 400
_ZSTD_e_continue, _ZSTD_e_flush, _ZSTD_e_end = 0, 1, 2
_lib = None


def _libzstd():
    global _lib
    if _lib is None:
        name = ctypes.util.find_library("zstd") or "libzstd.so.1"
        L = C.CDLL(name)
        L.ZSTD_createCCtx.restype = C.c_void_p
        L.ZSTD_freeCCtx.argtypes = [C.c_void_p]
        L.ZSTD_CCtx_setParameter.restype = C.c_size_t
        L.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ZSTD_compressStream2.restype = C.c_size_t
        L.ZSTD_compressStream2.argtypes = [C.c_void_p, C.POINTER(_OutBuf), C.POINTER(_InBuf), C.c_int]
        L.ZSTD_isError.restype = C.c_uint
        L.ZSTD_isError.argtypes = [C.c_size_t]
        L.ZSTD_getErrorName.restype = C.c_char_p
        L.ZSTD_getErrorName.argtypes = [C.c_size_t]
        L.ZSTD_CStreamOutSize.restype = C.c_size_t
        L.ZSTD_createDCtx.restype = C.c_void_p
        L.ZSTD_freeDCtx.argtypes = [C.c_void_p]
        L.ZSTD_decompressStream.restype = C.c_size_t
        L.ZSTD_decompressStream.argtypes = [C.c_void_p, C.POINTER(_OutBuf), C.POINTER(_InBuf)]
        L.ZSTD_DStreamOutSize.restype = C.c_size_t
        _lib = L
    return _lib


class ZstdWriter:
    """File-like ``write(bytes)`` / ``close()`` producing one zstd frame at the given level."""

    def __init__(self, raw, level: int = 3, workers: int = 0):
        self._raw = raw
        self._backend = None
        if _zstandard is not None:
            self._backend = _zstandard.ZstdCompressor(level=level, threads=workers).stream_writer(raw, closefd=False)
            return
        L = self._L = _libzstd()
        self._ctx = L.ZSTD_createCCtx()
        rc = L.ZSTD_CCtx_setParameter(self._ctx, _ZSTD_c_compressionLevel, level)
        self._check(rc)
        if workers > 0:  # one frame, compressed by `workers` threads (no-op if libzstd lacks MT support)
            rc = L.ZSTD_CCtx_setParameter(self._ctx, _ZSTD_c_nbWorkers, int(workers))
            if L.ZSTD_isError(rc):
                workers = 0
        # a large output buffer: one ZSTD_compressStream2 call (GIL released) then consumes megabytes of input
        # instead of returning to Python every 128 KB (that loop alone held the CLI at 24 Msamp/s)
        self._cap = max(int(L.ZSTD_CStreamOutSize()), 8 << 20)
        self._out = C.create_string_buffer(self._cap)
        self._view = memoryview(self._out)

    def _check(self, rc):
        if self._L.ZSTD_isError(rc):
            raise OSError("zstd: " + self._L.ZSTD_getErrorName(rc).decode())

    def _pump(self, data: bytes, mode: int):
        # `data` (bytes) stays referenced for the duration of the call: no copy of a 300 MB CSV blob
        ib = _InBuf(C.cast(C.c_char_p(data), C.c_void_p) if data else None, len(data), 0)
        while True:
            ob = _OutBuf(C.cast(self._out, C.c_void_p), self._cap, 0)
            remaining = self._L.ZSTD_compressStream2(self._ctx, C.byref(ob), C.byref(ib), mode)
            self._check(remaining)
            if ob.pos:
                self._raw.write(self._view[:ob.pos])
            done = (ib.pos == ib.size) if mode == _ZSTD_e_continue else (remaining == 0)
            if done:
                break

    def write(self, data: bytes) -> int:
        if self._backend is not None:
            return self._backend.write(data)
        if data:
            self._pump(bytes(data), _ZSTD_e_continue)
        return len(data)

    def flush(self):
        if self._backend is not None:
            self._backend.flush()
        else:
            self._pump(b"", _ZSTD_e_flush)

    def close(self):
        if self._backend is not None:
            self._backend.close()
            self._backend = None
            return
        if getattr(self, "_ctx", None):
            self._pump(b"", _ZSTD_e_end)
            self._L.ZSTD_freeCCtx(self._ctx)
            self._ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class ParallelZstdWriter:
    """``write(bytes)`` / ``close()`` like :class:`ZstdWriter`, but the stream is cut into chunks
    that are compressed concurrently (``ZSTD_compress`` releases the GIL) and written in order as
    independent frames. A sequence of frames is a valid ``.zst`` stream: ``zstd -d``, ``zstandard``
    and :func:`decompress` return the concatenation. One level-3 thread manages ~400 MB/s of CSV,
    which made the writer the slowest stage of ``dynamont-resquiggle`` on an MI355X."""

    def __init__(self, raw, level: int = 3, threads: int = 4, chunk_bytes: int = 4 << 20, single_frame: bool = False):
        """``single_frame=True``: the chunks become the jobs of ONE zstd frame, the way zstd's own multi-threaded
        mode (zstdmt, not compiled into the image's libzstd 1.4.8) builds it: every job is compressed by a context
        of its own with the same parameters; job 0 keeps its frame header, every later job flushes its header into
        the void (``ZSTD_compressContinue`` with 0 bytes) and calls ``ZSTD_invalidateRepCodes`` so that its first
        block does not lean on repeat offsets the decoder will not have at that point (a fresh context has no
        previous entropy tables either); the frame is closed by an empty last block. Any zstd decoder sees one
        ordinary frame."""
        from concurrent.futures import ThreadPoolExecutor
        self._raw = raw
        self._single = bool(single_frame)
        import threading
        self._tls, self._ctxs, self._ctx_lock = threading.local(), [], threading.Lock()
        self._level = int(level)
        self._chunk = int(chunk_bytes)
        self._threads = max(1, int(threads))
        self._pool = ThreadPoolExecutor(max_workers=self._threads)
        self._pending = []      # futures in stream order
        self._buf = []          # small writes waiting for a full chunk
        self._buffered = 0
        self._frames = 0
        L = self._L = _libzstd()
        L.ZSTD_compress.restype = C.c_size_t
        L.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        L.ZSTD_compressBound.restype = C.c_size_t
        L.ZSTD_compressBound.argtypes = [C.c_size_t]
        if self._single:
            for name in ("ZSTD_compressBegin", "ZSTD_compressContinue", "ZSTD_compressEnd", "ZSTD_invalidateRepCodes"):
                if not hasattr(L, name):
                    raise OSError(f"libzstd lacks {name}")
            L.ZSTD_compressBegin.restype = C.c_size_t
            L.ZSTD_compressBegin.argtypes = [C.c_void_p, C.c_int]
            for fn in (L.ZSTD_compressContinue, L.ZSTD_compressEnd):
                fn.restype = C.c_size_t
                fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
            L.ZSTD_invalidateRepCodes.restype = None
            L.ZSTD_invalidateRepCodes.argtypes = [C.c_void_p]

    def _err(self, rc):
        if self._L.ZSTD_isError(rc):
            raise OSError("zstd: " + self._L.ZSTD_getErrorName(rc).decode())
        return rc

    def _job(self, owner, addr: int, n: int, first: bool, last: bool) -> bytes:
        """One job of the single frame (see __init__). `last`: the terminating job (no data, closes the frame)."""
        L = self._L
        ctx = getattr(self._tls, "ctx", None)  # one context per worker thread, reset by ZSTD_compressBegin
        if ctx is None:
            ctx = self._tls.ctx = L.ZSTD_createCCtx()
            with self._ctx_lock:
                self._ctxs.append(ctx)
        self._err(L.ZSTD_compressBegin(ctx, self._level))
        cap = int(L.ZSTD_compressBound(n)) + 64
        dst = C.create_string_buffer(cap)
        if not first:
            self._err(L.ZSTD_compressContinue(ctx, dst, cap, None, 0))  # this context's frame header: dropped
            L.ZSTD_invalidateRepCodes(ctx)
        fn = L.ZSTD_compressEnd if last else L.ZSTD_compressContinue
        rc = self._err(fn(ctx, dst, cap, addr if n else None, n))
        return dst.raw[:rc]

    def _compress(self, owner, addr: int, n: int, first: bool = True) -> bytes:
        # `owner` keeps the source bytes alive while the worker reads [addr, addr + n)
        if self._single:
            return self._job(owner, addr, n, first, False)
        L = self._L
        cap = int(L.ZSTD_compressBound(n))
        dst = C.create_string_buffer(cap)
        rc = L.ZSTD_compress(dst, cap, addr, n, self._level)
        if L.ZSTD_isError(rc):
            raise OSError("zstd: " + L.ZSTD_getErrorName(rc).decode())
        return dst.raw[:rc]

    def _submit(self, data: bytes, off: int, n: int):
        base = C.cast(C.c_char_p(data), C.c_void_p).value or 0
        self._pending.append(self._pool.submit(self._compress, data, base + off, n, self._frames == 0))
        self._frames += 1
        while len(self._pending) > 2 * self._threads:  # back-pressure: bounded memory
            self._raw.write(self._pending.pop(0).result())
        while self._pending and self._pending[0].done():
            self._raw.write(self._pending.pop(0).result())

    def write(self, data) -> int:
        data = bytes(data)
        n = len(data)
        if n == 0:
            return 0
        if self._buffered + n < self._chunk:
            self._buf.append(data)
            self._buffered += n
            return n
        if self._buf:
            head = b"".join(self._buf)
            self._buf, self._buffered = [], 0
            self._submit(head, 0, len(head))
        for off in range(0, n, self._chunk):
            self._submit(data, off, min(self._chunk, n - off))
        return n

    def flush(self):
        if self._buf:
            head = b"".join(self._buf)
            self._buf, self._buffered = [], 0
            self._submit(head, 0, len(head))
        while self._pending:
            self._raw.write(self._pending.pop(0).result())

    def close(self):
        if self._pool is None:
            return
        self.flush()
        if self._single:  # the empty last block that closes the frame (with the header, if nothing was written)
            self._raw.write(self._job(b"", 0, 0, self._frames == 0, True))
        elif self._frames == 0:  # an empty stream is still one (empty) frame
            self._raw.write(self._compress(b"", 0, 0))
        self._pool.shutdown()
        self._pool = None
        for ctx in self._ctxs:
            self._L.ZSTD_freeCCtx(ctx)
        self._ctxs = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def open_writer(raw, level: int = 3, threads: int = 0, parallel_frames: bool = False):
    """The CSV writer of the CLIs. Default: ONE zstd frame, as the reference writes it, its 4 MB jobs compressed
    on up to 8 threads (:class:`ParallelZstdWriter` with ``single_frame=True``; the one-thread :class:`ZstdWriter`
    if the library lacks the block-level entry points). ``parallel_frames=True`` (``--parallel-zstd-frames``):
    the same chunks written as consecutive independent frames -- for consumers that read across frames
    (``zstd -d``, ``zstandard`` with ``read_across_frames=True``, :func:`decompress`); no faster than the default
    any more, kept for files whose chunks are to be decompressed independently."""
    import os
    if threads <= 0:
        threads = max(1, min(8, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    try:
        return ParallelZstdWriter(raw, level, threads, single_frame=not parallel_frames)
    except OSError:
        return ZstdWriter(raw, level, threads)


def count_frames(data: bytes) -> int:
    """Number of zstd frames in ``data`` (walks the frames with ZSTD_findFrameCompressedSize)."""
    L = _libzstd()
    L.ZSTD_findFrameCompressedSize.restype = C.c_size_t
    L.ZSTD_findFrameCompressedSize.argtypes = [C.c_void_p, C.c_size_t]
    src = C.create_string_buffer(data, len(data))
    base = C.cast(src, C.c_void_p).value or 0
    pos = frames = 0
    while pos < len(data):
        n = L.ZSTD_findFrameCompressedSize(base + pos, len(data) - pos)
        if L.ZSTD_isError(n):
            raise OSError("zstd: " + L.ZSTD_getErrorName(n).decode())
        pos += n
        frames += 1
    return frames


def decompress(data: bytes) -> bytes:
    """Whole-buffer decompression of (possibly multi-frame) zstd data; used by tests/tools."""
    if _zstandard is not None:
        return _zstandard.ZstdDecompressor().decompressobj(read_across_frames=True).decompress(data)
    L = _libzstd()
    ctx = L.ZSTD_createDCtx()
    cap = int(L.ZSTD_DStreamOutSize())
    out = C.create_string_buffer(cap)
    src = C.create_string_buffer(data, len(data))
    ib = _InBuf(C.cast(src, C.c_void_p), len(data), 0)
    chunks = []
    try:
        while ib.pos < ib.size:
            ob = _OutBuf(C.cast(out, C.c_void_p), cap, 0)
            rc = L.ZSTD_decompressStream(ctx, C.byref(ob), C.byref(ib))
            if L.ZSTD_isError(rc):
                raise OSError("zstd: " + L.ZSTD_getErrorName(rc).decode())
            chunks.append(out.raw[:ob.pos])
    finally:
        L.ZSTD_freeDCtx(ctx)
    return b"".join(chunks)
