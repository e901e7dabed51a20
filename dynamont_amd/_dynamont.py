"""Host-side mirror of the reference's pybind11 module ``dynamont._dynamont``
(src/cpp/aligner_bindings.cpp:180-219), backed by the C ABI in include/dynamont_mi.h.

Same names, argument meaning, return dicts and exception types/messages:

    Aligner(model_file, pore, mode="basic", threads=1, band=400)
    Aligner.align(signal, sequence, calc_probabilities=False) -> dict
    Aligner.train(signal, sequence) -> dict
    PoreType, pore_type(str)

Additions for the GPU (a single read cannot fill an MI355X): ``align_batch``/``train_batch``
and the staged ``Batch`` whose inputs are HBM-resident before kernels run.
"""
from __future__ import annotations

import ctypes as C
import enum
from typing import Sequence

import numpy as np

from . import _native as N

ERRCAP = 4096


class PoreType(enum.IntEnum):
    """aligner_bindings.cpp:184-189 / include/dynamont/aligner.hpp:26-33"""
    RNA002 = 0
    RNA004 = 1
    DNA_R9 = 2
    DNA_R10_260 = 3
    DNA_R10_400 = 4


def _raise(code: int, msg: str):
    if code == N.DYN_ERR_INVALID_ARGUMENT:
        raise ValueError(msg)          # std::invalid_argument under pybind11
    if code == N.DYN_ERR_OUT_OF_MEMORY:
        raise MemoryError(msg)
    raise RuntimeError(msg)            # std::runtime_error under pybind11 / device failures


def pore_type(pore: str) -> PoreType:
    """aligner_bindings.cpp:18-32,218"""
    out = C.c_int()
    err = C.create_string_buffer(ERRCAP)
    rc = N.lib().dyn_pore_from_string(str(pore).encode(), C.byref(out), err, ERRCAP)
    if rc != N.DYN_OK:
        _raise(rc, err.value.decode())
    return PoreType(out.value)


def read_error_message(status: int, bad_char: bytes | int = 0) -> str:
    buf = C.create_string_buffer(256)
    if isinstance(bad_char, int):
        bad_char = bytes([bad_char & 0xFF])
    N.lib().dyn_read_strerror(int(status), bad_char[:1] or b"\0", buf, 256)
    return buf.value.decode(errors="replace")


def _ptr(a: np.ndarray, t):
    return a.ctypes.data_as(t)


def _pack(signals: Sequence, sequences: Sequence[str]):
    n = len(signals)
    if n != len(sequences):
        raise ValueError("signals and sequences differ in length")
    sig_off = np.zeros(n + 1, dtype=np.uint64)
    seq_off = np.zeros(n + 1, dtype=np.uint64)
    arrs = []
    for i, s in enumerate(signals):
        a = np.ascontiguousarray(s, dtype=np.float64)  # py::array::c_style | forcecast
        if a.ndim != 1:
            raise ValueError("Signal must be a one-dimensional array")  # aligner_bindings.cpp:137-138
        arrs.append(a)
        sig_off[i + 1] = sig_off[i] + a.size
        seq_off[i + 1] = seq_off[i] + len(sequences[i])
    sig = np.concatenate(arrs) if arrs else np.zeros(0, dtype=np.float64)
    if sig.size == 0:
        sig = np.zeros(1, dtype=np.float64)[:0]
    seqs = "".join(sequences).encode("latin-1")
    return np.ascontiguousarray(sig), sig_off, seqs, seq_off


class AlignBatchResult:
    """Columnar results of one batch; ``read(i)`` gives the reference's per-read dict."""

    def __init__(self, n, cap):
        self.n = n
        self.cap = cap
        self.Z = np.zeros(n)
        self.status = np.zeros(n, dtype=np.int32)
        self.bad_char = np.zeros(n, dtype=np.uint8)
        self.seg_offsets = np.zeros(n + 1, dtype=np.uint64)
        self.n_segments = np.zeros(n, dtype=np.uint64)
        self.sequence_positions = np.zeros(cap, dtype=np.uint64)
        self.signal_positions = np.zeros(cap, dtype=np.uint64)
        self.probabilities = np.zeros(cap)
        self.states = np.zeros(cap, dtype=np.uint8)
        self._c = N.DynAlignOut(_ptr(self.Z, N.c_double_p), _ptr(self.status, N.c_i32_p),
                                self.bad_char.ctypes.data, _ptr(self.seg_offsets, N.c_u64_p),
                                _ptr(self.n_segments, N.c_u64_p), _ptr(self.sequence_positions, N.c_u64_p),
                                _ptr(self.signal_positions, N.c_u64_p), _ptr(self.probabilities, N.c_double_p),
                                _ptr(self.states, N.c_u8_p), cap)

    def error(self, i: int) -> str | None:
        if self.status[i] == 0:
            return None
        return read_error_message(int(self.status[i]), int(self.bad_char[i]))

    def read(self, i: int) -> dict:
        """Per-read dict exactly as resultToPython builds it (aligner_bindings.cpp:53-84);
        raises RuntimeError(message) for a failed read like the reference's align()."""
        if self.status[i] != 0:
            raise RuntimeError(self.error(i))
        a = int(self.seg_offsets[i])
        b = a + int(self.n_segments[i])
        return {
            "Z": float(self.Z[i]),
            "sequence_positions": self.sequence_positions[a:b].copy(),
            "signal_positions": self.signal_positions[a:b].copy(),
            "probabilities": self.probabilities[a:b].copy(),
            "states": [chr(c) for c in self.states[a:b]],
            "polishes": [""] * (b - a),
        }


def format_csv(aligner: "Aligner", res: AlignBatchResult, sequences: Sequence[str], readids: Sequence[str],
               signalids: Sequence[str], sig_offsets: Sequence[int], last_index: Sequence[int],
               threads: int = 8, compact: bool = False, seqs_packed=None):
    """Native segmentation_to_string for a batch (dyn_format_csv). Returns (buffer, begin[n], end[n]):
    read i's CSV rows are ``buffer[begin[i]:end[i]]`` (empty for failed reads). ``compact=True`` closes the gaps
    between the reads' ranges (dyn_csv_compact): ``buffer[:end[-1]]`` is then the whole batch in read order.
    ``seqs_packed`` = (bytes, uint64 offsets) of the sequences if the caller already holds them packed."""
    n = res.n
    if seqs_packed is not None:
        seqs, seq_off = seqs_packed
    else:
        seq_off = np.zeros(n + 1, dtype=np.uint64)
        seq_off[1:] = np.cumsum(np.array([len(s) for s in sequences], dtype=np.uint64))
        seqs = "".join(sequences).encode("latin-1")
    rid = (C.c_char_p * n)(*[str(x).encode() for x in readids])
    sid = (C.c_char_p * n)(*[str(x).encode() for x in signalids])
    so = np.ascontiguousarray(sig_offsets, dtype=np.int64)
    li = np.ascontiguousarray(last_index, dtype=np.int64)
    begin = np.zeros(n, dtype=np.uint64)
    end = np.zeros(n, dtype=np.uint64)
    L = N.lib()
    cap = int(L.dyn_format_csv_bound(aligner._h, n, C.byref(res._c), rid, sid))
    # one grow-only buffer per handle: first-touch page faults of a fresh 350 MB buffer cost ~100x
    # the formatting itself (15 ms per 1 024-read batch with warm pages)
    buf = getattr(aligner, "_csv_buf", None)
    if buf is None or buf.size < cap:
        buf = aligner._csv_buf = np.empty(max(cap, 1), dtype=np.uint8)
    rc = L.dyn_format_csv(aligner._h, n, C.byref(res._c), seqs, _ptr(seq_off, N.c_u64_p), rid, sid,
                          so.ctypes.data_as(C.POINTER(C.c_int64)), li.ctypes.data_as(C.POINTER(C.c_int64)),
                          int(threads), buf.ctypes.data, cap, _ptr(begin, N.c_u64_p), _ptr(end, N.c_u64_p))
    if rc != N.DYN_OK:
        _raise(rc, "dyn_format_csv failed")
    if compact:
        L.dyn_csv_compact(buf.ctypes.data, n, _ptr(begin, N.c_u64_p), _ptr(end, N.c_u64_p))
    return buf, begin, end


class TrainBatchResult:
    def __init__(self, n, cap, num_kmers, pooled: bool, emissions: bool = True):
        """``emissions=False``: no per-read sparse emission updates (their arrays stay empty and the library
        skips the per-column D2H and grouping) -- for jobs that only consume Z, transitions and the pooled
        statistics."""
        if not emissions:
            cap = 0
        self.n = n
        self.num_kmers = num_kmers
        self.Z = np.zeros(n)
        self.status = np.zeros(n, dtype=np.int32)
        self.bad_char = np.zeros(n, dtype=np.uint8)
        self.transitions = np.zeros(3 * n)
        self.em_offsets = np.zeros(n + 1, dtype=np.uint64)
        self.em_count = np.zeros(n, dtype=np.uint64)
        self.em_code = np.zeros(cap, dtype=np.int32)
        self.em_mean = np.zeros(cap)
        self.em_stdev = np.zeros(cap)
        self.em_weight = np.zeros(cap)
        self.em_sum = np.zeros(cap)
        self.em_sumsq = np.zeros(cap)
        self.trans_counts = np.zeros(2 * n)
        self.pooled = np.zeros(3 * num_kmers) if pooled else None
        opt = (lambda a, t: _ptr(a, t)) if emissions else (lambda a, t: None)
        self._c = N.DynTrainOut(_ptr(self.Z, N.c_double_p), _ptr(self.status, N.c_i32_p), self.bad_char.ctypes.data,
                                _ptr(self.transitions, N.c_double_p), _ptr(self.em_offsets, N.c_u64_p),
                                _ptr(self.em_count, N.c_u64_p), opt(self.em_code, N.c_i32_p),
                                opt(self.em_mean, N.c_double_p), opt(self.em_stdev, N.c_double_p),
                                opt(self.em_weight, N.c_double_p), opt(self.em_sum, N.c_double_p),
                                opt(self.em_sumsq, N.c_double_p), _ptr(self.trans_counts, N.c_double_p), cap)

    def error(self, i: int) -> str | None:
        if self.status[i] == 0:
            return None
        return read_error_message(int(self.status[i]), int(self.bad_char[i]))

    def sparse(self, i: int):
        a = int(self.em_offsets[i])
        b = a + int(self.em_count[i])
        return self.em_code[a:b], self.em_mean[a:b], self.em_stdev[a:b]

    def read(self, i: int, model_mean: np.ndarray, model_stdev: np.ndarray) -> dict:
        """Dense dict of trainingResultToPython (aligner_bindings.cpp:86-107), rebuilt on demand."""
        if self.status[i] != 0:
            raise RuntimeError(self.error(i))
        mean = model_mean.copy()
        sd = model_stdev.copy()
        code, m, s = self.sparse(i)
        mean[code] = m
        sd[code] = s
        t = self.transitions[3 * i:3 * i + 3]
        return {
            "Z": float(self.Z[i]),
            "transition_params": {"m1": float(t[0]), "e1": float(t[1]), "e2": float(t[2])},
            "emission_model": [{"mean": float(a), "stdev": float(b)} for a, b in zip(mean, sd)],
        }


class Batch:
    """Staged form (dyn_batch_*): inputs are validated, k-mer coded and resident in HBM after
    construction; align()/train() only launch kernels; fetch() copies results back."""

    def __init__(self, aligner: "Aligner", signals, sig_offsets, seqs: bytes, seq_offsets, raw=None):
        """raw = None: ``signals`` are normalised float64 samples. raw = dict(shift, scale, window,
        n_sigmas, f32): ``signals`` are RAW float32 / int16 / float64 samples and the preprocessing of
        segment.py:146-153 runs on the device (dyn_batch_create_raw)."""
        self._al = aligner
        self._L = N.lib()
        self.n = len(sig_offsets) - 1
        self._sig_off = np.ascontiguousarray(sig_offsets, dtype=np.uint64)
        self._seq_off = np.ascontiguousarray(seq_offsets, dtype=np.uint64)
        self.capacity = int(self._L.dyn_segment_capacity(aligner._h, self.n, _ptr(self._seq_off, N.c_u64_p)))
        h = C.c_void_p()
        if raw is None:
            sig = np.ascontiguousarray(signals, dtype=np.float64)
            rc = self._L.dyn_batch_create(aligner._h, self.n, _ptr(sig, N.c_double_p), _ptr(self._sig_off, N.c_u64_p),
                                          seqs, _ptr(self._seq_off, N.c_u64_p), C.byref(h))
        else:
            sig = np.ascontiguousarray(signals)
            code = {np.dtype(np.float32): 0, np.dtype(np.int16): 1, np.dtype(np.float64): 2}.get(sig.dtype)
            if code is None:
                raise ValueError(f"raw signal dtype {sig.dtype} is not float32, int16 or float64")
            shift = np.ascontiguousarray(raw["shift"], dtype=np.float64)
            scale = np.ascontiguousarray(raw["scale"], dtype=np.float64)
            rc = self._L.dyn_batch_create_raw(aligner._h, self.n, sig.ctypes.data, code, _ptr(self._sig_off, N.c_u64_p),
                                              _ptr(shift, N.c_double_p), _ptr(scale, N.c_double_p),
                                              int(raw.get("window", 3)), float(raw.get("n_sigmas", 3.0)),
                                              int(bool(raw.get("f32", False))), seqs, _ptr(self._seq_off, N.c_u64_p),
                                              C.byref(h))
        if rc != N.DYN_OK:
            _raise(rc, aligner.last_error())
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.dyn_batch_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def align(self, calc_probabilities: bool = True) -> None:
        rc = self._L.dyn_batch_align(self._h, int(bool(calc_probabilities)))
        if rc != N.DYN_OK:
            _raise(rc, self._al.last_error())

    def train(self) -> None:
        rc = self._L.dyn_batch_train(self._h)
        if rc != N.DYN_OK:
            _raise(rc, self._al.last_error())

    def fetch(self, out: AlignBatchResult | None = None) -> AlignBatchResult:
        """Results of the last align(). ``out``: a result object of an earlier batch to refill (same
        number of reads, enough segment capacity) instead of allocating and page-faulting in ~50 MB of
        fresh arrays per 1 024-read batch; whatever it held is overwritten."""
        if out is None or out.n != self.n or out.cap < self.capacity:
            out = AlignBatchResult(self.n, self.capacity + self.capacity // 8)
        rc = self._L.dyn_batch_fetch(self._h, C.byref(out._c))
        if rc != N.DYN_OK:
            _raise(rc, self._al.last_error())
        return out

    def fetch_train(self, pooled: bool = False) -> TrainBatchResult:
        out = TrainBatchResult(self.n, self.capacity, self._al.num_kmers, pooled)
        rc = self._L.dyn_batch_fetch_train(self._h, C.byref(out._c),
                                           _ptr(out.pooled, N.c_double_p) if pooled else None)
        if rc != N.DYN_OK:
            _raise(rc, self._al.last_error())
        return out

    def signals(self) -> np.ndarray:
        """The device-resident (normalised, Hampel-filtered) samples, concatenated."""
        out = np.empty(int(self._sig_off[-1] - self._sig_off[0]))
        rc = self._L.dyn_batch_signals(self._h, _ptr(out, N.c_double_p), out.size)
        if rc != N.DYN_OK:
            _raise(rc, self._al.last_error())
        return out

    def timing(self) -> dict:
        t = N.DynTiming()
        self._L.dyn_batch_timing(self._h, C.byref(t))
        return {k: getattr(t, k) for k, _ in N.DynTiming._fields_ if k != "reserved"}

    def device_results(self):
        """(rows_ptr, capacity, state_ptr): device addresses for a gather without a host hop."""
        rows = C.c_void_p()
        st = C.c_void_p()
        cap = C.c_uint64()
        rc = self._L.dyn_batch_device_results(self._h, C.byref(rows), C.byref(cap), C.byref(st))
        if rc != N.DYN_OK:
            _raise(rc, self._al.last_error())
        return rows.value, cap.value, st.value

    def device_pooled(self):
        p = C.c_void_p()
        cnt = C.c_uint64()
        rc = self._L.dyn_batch_device_pooled(self._h, C.byref(p), C.byref(cnt))
        if rc != N.DYN_OK:
            _raise(rc, self._al.last_error())
        return p.value, cnt.value


class AsyncBatch:
    """Ticket of ``Aligner.align_async`` / ``train_async`` (dyn_batch_align_async): the batch runs on the
    handle's pipeline; ``wait()`` returns the filled result object. The input arrays are kept alive here
    (the library reads them until the batch is complete)."""

    def __init__(self, aligner: "Aligner", handle, result, keep):
        self._al = aligner
        self._L = N.lib()
        self._h = handle
        self.result = result
        self._keep = keep
        self._waited = False

    def wait(self):
        if not self._waited:
            rc = self._L.dyn_batch_wait(self._h)
            self._waited = True
            if rc != N.DYN_OK:
                _raise(rc, self._al.last_error())
        return self.result

    def timing(self) -> dict:
        self.wait()
        t = N.DynTiming()
        self._L.dyn_batch_timing(self._h, C.byref(t))
        return {k: getattr(t, k) for k, _ in N.DynTiming._fields_}

    device_results = Batch.device_results
    device_pooled = Batch.device_pooled

    def close(self):
        if getattr(self, "_h", None):
            self._L.dyn_batch_destroy(self._h)  # waits first if the batch is still in flight
            self._h = None
            self._keep = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def release_cached_memory() -> None:
    """Free the lattice pools that destroyed aligners have parked for their successors (dyn_release_cached_memory)."""
    N.lib().dyn_release_cached_memory()


def pinned_empty(n: int, dtype) -> np.ndarray:
    """An uninitialised 1-D array in page-locked host memory (dyn_host_alloc): H2D/D2H of the asynchronous
    calls then run as plain DMA. The memory is released when the array (and every view of it) is gone."""
    dt = np.dtype(dtype)
    nbytes = max(1, int(n) * dt.itemsize)
    L = N.lib()
    p = L.dyn_host_alloc(nbytes)
    if not p:
        raise MemoryError(f"dyn_host_alloc({nbytes}) failed")

    class _Owner:
        def __init__(self, ptr):
            self.ptr = ptr

        def __del__(self):
            L.dyn_host_free(self.ptr)

    buf = (C.c_char * nbytes).from_address(p)
    buf._owner = _Owner(p)  # ctypes array -> numpy keeps `buf` alive as its base
    return np.frombuffer(buf, dtype=dt, count=int(n))


class _Scattered:
    """A batch's raw slices as a pointer table (DYN_RAW_SCATTERED); keeps the slices alive."""

    def __init__(self, table: np.ndarray, slices, dtype):
        self.table, self.slices, self.dtype = table, slices, np.dtype(dtype)
        self.ctypes = table.ctypes  # .ctypes.data -> the table


class Aligner:
    """Mirror of ``_dynamont.Aligner`` (aligner_bindings.cpp:191-216).

    ``device``: HIP device ordinal (default: current device). ``device=None`` with no GPU raises;
    ``device="host"`` creates a handle for the host-side contract only (model/validation), every
    compute call on it raises RuntimeError -- there is no CPU compute path.
    """

    def __init__(self, model_file: str, pore, mode: str = "basic", threads: int = 1, band: int = 400,
                 device=None):
        if isinstance(pore, str):
            pore = pore_type(pore)
        elif not isinstance(pore, (PoreType, int, np.integer)):
            raise TypeError("pore must be a PoreType or str")
        self._L = N.lib()
        if device == "host":
            dev = N.DYN_DEVICE_HOST_ONLY
        elif device is None:
            dev = -1
        else:
            dev = int(device)
        h = C.c_void_p()
        err = C.create_string_buffer(ERRCAP)
        rc = self._L.dyn_aligner_create(str(model_file).encode(), int(pore), str(mode).encode(), int(threads),
                                        int(band), dev, C.byref(h), err, ERRCAP)
        if rc != N.DYN_OK:
            _raise(rc, err.value.decode())
        self._h = h
        info = N.DynInfo()
        self._L.dyn_aligner_info(h, C.byref(info))
        self.info = info
        self.pore = PoreType(info.pore)
        self.rna = bool(info.rna)
        self.kmer_size = int(info.kmer_size)
        self.num_kmers = int(info.num_kmers)
        self._model = None

    def close(self):
        if getattr(self, "_h", None):
            self._L.dyn_aligner_destroy(self._h)
            self._h = None

    __del__ = close

    def last_error(self) -> str:
        return (self._L.dyn_aligner_last_error(self._h) or b"").decode(errors="replace")

    def set_mem_budget(self, nbytes: int) -> None:
        self._L.dyn_aligner_set_mem_budget(self._h, int(nbytes))

    STRICT_MODES = {"off": 0, "ties": 1, "start": 1, "all": 2}  # "start": round 3's name of mode 1

    def set_model(self, mean, stdev) -> None:
        """Replace the model table (k-mer-code order, as model_table() returns it): the aligner then equals one built
        from a model file with these values (dyn_aligner_set_model)."""
        t = np.empty(2 * self.num_kmers)
        t[0::2] = np.asarray(mean, dtype=np.float64)
        t[1::2] = np.asarray(stdev, dtype=np.float64)
        rc = self._L.dyn_aligner_set_model(self._h, _ptr(t, N.c_double_p))
        if rc != N.DYN_OK:
            _raise(rc, self.last_error())
        self._model = None  # model_table() reads it back

    def set_strict(self, mode) -> None:
        """dyn_aligner_set_strict: 0/"off" plain arithmetic; 1/"ties" (the default of a new handle) reads that carry a
        structural tie -- two neighbouring columns with the same emission parameters -- go through the kernels that
        reproduce the reference's sums bit for bit; 2/"all" every read."""
        m = self.STRICT_MODES[mode] if isinstance(mode, str) else int(mode)
        rc = self._L.dyn_aligner_set_strict(self._h, m)
        if rc != N.DYN_OK:
            raise ValueError(self.last_error() or "strict mode must be 0 (off), 1 (ties) or 2 (all)")

    def set_session_mode(self, enabled: bool, reserved_cus: int = 0) -> None:
        """dyn_aligner_set_session_mode: the resident read queue on / off, and compute units kept free of it (for RCCL's
        kernels, which do not fit beside a resident session)."""
        rc = self._L.dyn_aligner_set_session_mode(self._h, 1 if enabled else 0, int(reserved_cus))
        if rc != N.DYN_OK:
            _raise(rc, self.last_error())

    def session_stats(self) -> dict:
        """dyn_aligner_session_stats: totals over the closed sessions of the resident read queue (closes an open one and
        waits for its waves first). ``wave_occupancy`` = busy / lifetime wave-cycles."""
        t = N.DynSessionStats()
        rc = self._L.dyn_aligner_session_stats(self._h, C.byref(t))
        if rc != N.DYN_OK:
            _raise(rc, self.last_error())
        d = {k: getattr(t, k) for k, _ in N.DynSessionStats._fields_}
        d["wave_occupancy"] = d["wave_cycles_busy"] / d["wave_cycles_life"] if d["wave_cycles_life"] else 0.0
        pw = C.c_uint64(0)
        self._L.dyn_aligner_session_page_wait(self._h, C.byref(pw))   # paged sessions: the part of `idle` spent getting pages
        d["wave_cycles_pages"] = int(pw.value)
        sp = (C.c_uint64 * 4)()
        self._L.dyn_aligner_session_idle_split(self._h, sp)   # where the idle share sat: first read, its pages, last turn, longest
        d["wave_cycles_before_first_read"], d["wave_cycles_before_first_read_pages"] = int(sp[0]), int(sp[1])
        d["wave_cycles_last_turn"], d["wave_cycles_longest_last_turn"] = int(sp[2]), int(sp[3])
        return d

    def set_train_zcheck(self, on: bool) -> None:
        """dyn_aligner_set_train_zcheck: also refuse the reads the reference's |Zf - Zb| rule refuses (one more Z-only
        forward sweep per read)."""
        self._L.dyn_aligner_set_train_zcheck(self._h, 1 if on else 0)

    def tie_rows(self, kmers, signal_len: int) -> int:
        """dyn_tie_rows: 0 = the read carries no structural tie; else the forward rows mode "ties" runs bit for bit
        (0xffffffff = all)."""
        km = np.ascontiguousarray(kmers, dtype=np.int32)
        return int(self._L.dyn_tie_rows(self._h, _ptr(km, N.c_i32_p), len(km), int(signal_len)))

    def model_table(self):
        """(mean, stdev) in k-mer-code order."""
        if self._model is None:
            out = np.zeros(2 * self.num_kmers)
            self._L.dyn_aligner_model(self._h, _ptr(out, N.c_double_p))
            self._model = (out[0::2].copy(), out[1::2].copy())
        return self._model

    # -- host-side contract -------------------------------------------------------------------
    def validate(self, signal_lengths: Sequence[int], sequences: Sequence[str]):
        """validateInput + sequenceToKmers per read: (status[n], messages[n], kmers list)."""
        n = len(sequences)
        sig_off = np.zeros(n + 1, dtype=np.uint64)
        sig_off[1:] = np.cumsum(np.asarray(signal_lengths, dtype=np.uint64))
        seq_off = np.zeros(n + 1, dtype=np.uint64)
        seq_off[1:] = np.cumsum(np.array([len(s) for s in sequences], dtype=np.uint64))
        seqs = "".join(sequences).encode("latin-1")
        status = np.zeros(n, dtype=np.int32)
        bad = np.zeros(n, dtype=np.uint8)
        cap = int(self._L.dyn_segment_capacity(self._h, n, _ptr(seq_off, N.c_u64_p)))
        kmers = np.zeros(max(cap, 1), dtype=np.int32)
        rc = self._L.dyn_validate_batch(self._h, n, _ptr(sig_off, N.c_u64_p), seqs, _ptr(seq_off, N.c_u64_p),
                                        _ptr(status, N.c_i32_p), bad.ctypes.data, _ptr(kmers, N.c_i32_p), cap)
        if rc != N.DYN_OK:
            _raise(rc, self.last_error())
        msgs = [None if s == 0 else read_error_message(int(s), int(b)) for s, b in zip(status, bad)]
        out, off = [], 0
        for i, s in enumerate(sequences):
            kc = max(0, len(s) - self.kmer_size + 1)
            out.append(kmers[off:off + kc].copy() if status[i] == 0 else None)
            off += kc
        return status, msgs, out

    # -- batch API ------------------------------------------------------------------------------
    def batch(self, signals: Sequence, sequences: Sequence[str]) -> Batch:
        sig, sig_off, seqs, seq_off = _pack(signals, sequences)
        return Batch(self, sig, sig_off, seqs, seq_off)

    def batch_packed(self, signals, sig_offsets, seqs: bytes, seq_offsets) -> Batch:
        return Batch(self, signals, sig_offsets, seqs, seq_offsets)

    def batch_raw(self, raw_slices: Sequence, sequences: Sequence[str], shift, scale, window: int = 3,
                  n_sigmas: float = 3.0, f32: bool = False) -> Batch:
        """Batch from RAW sample slices (all float32, all int16 or all float64): normalisation and the
        Hampel filter of the reference's workers run on the device, bit-identically."""
        n = len(raw_slices)
        sig_off = np.zeros(n + 1, dtype=np.uint64)
        seq_off = np.zeros(n + 1, dtype=np.uint64)
        for i, s in enumerate(raw_slices):
            sig_off[i + 1] = sig_off[i] + len(s)
            seq_off[i + 1] = seq_off[i] + len(sequences[i])
        raw = np.concatenate(raw_slices) if n else np.zeros(0, dtype=np.float32)
        seqs = "".join(sequences).encode("latin-1")
        return Batch(self, raw, sig_off, seqs, seq_off,
                     raw=dict(shift=shift, scale=scale, window=window, n_sigmas=n_sigmas, f32=f32))

    def align_async(self, signals, sig_offsets, seqs: bytes, seq_offsets, calc_probabilities: bool = True,
                    out: AlignBatchResult | None = None) -> AsyncBatch:
        """dyn_batch_align_async on packed inputs (``synth.pack_reads`` layout): returns at once; the batch's
        validation, H2D, kernels, D2H and unpacking overlap with those of the batches submitted around it.
        ``out``: result object of an EARLIER, completed batch to refill."""
        sig = np.ascontiguousarray(signals, dtype=np.float64)
        sig_off = np.ascontiguousarray(sig_offsets, dtype=np.uint64)
        seq_off = np.ascontiguousarray(seq_offsets, dtype=np.uint64)
        n = len(sig_off) - 1
        cap = int(self._L.dyn_segment_capacity(self._h, n, _ptr(seq_off, N.c_u64_p)))
        if out is None or out.n != n or out.cap < cap:
            out = AlignBatchResult(n, cap + cap // 8)
        h = C.c_void_p()
        rc = self._L.dyn_batch_align_async(self._h, n, _ptr(sig, N.c_double_p), _ptr(sig_off, N.c_u64_p), seqs,
                                           _ptr(seq_off, N.c_u64_p), int(bool(calc_probabilities)), C.byref(out._c),
                                           C.byref(h))
        if rc != N.DYN_OK:
            _raise(rc, self.last_error())
        return AsyncBatch(self, h, out, (sig, sig_off, seqs, seq_off))

    def _raw_args(self, raw, shift, scale, calibration=None):
        scattered = isinstance(raw, (list, tuple))
        if scattered:  # one array per read: the library gathers them (DYN_RAW_SCATTERED), no concatenation here
            slices = [np.ascontiguousarray(x) for x in raw]
            dt = slices[0].dtype if slices else np.dtype(np.int16)
            if any(x.dtype != dt for x in slices):
                raise ValueError("raw slices of one batch must share a dtype")
            table = np.fromiter((x.ctypes.data for x in slices), dtype=np.uint64, count=len(slices))
            raw = _Scattered(table, slices, dt)
        else:
            raw = np.ascontiguousarray(raw)
        code = {np.dtype(np.float32): 0, np.dtype(np.int16): 1, np.dtype(np.float64): 2}.get(raw.dtype)
        if code is None:
            raise ValueError(f"raw signal dtype {raw.dtype} is not float32, int16 or float64")
        if scattered:
            code |= 0x100
        cal = (None, None)
        if calibration is not None:  # int16 ADC + pod5 calibration: picoampere formed on the device
            if code & 0xff != 1:
                raise ValueError("calibration arrays go with int16 ADC samples")
            code = 3 | (code & 0x100)
            cal = (np.ascontiguousarray(calibration[0], dtype=np.float32), np.ascontiguousarray(calibration[1], dtype=np.float32))
        return raw, code, np.ascontiguousarray(shift, dtype=np.float64), np.ascontiguousarray(scale, dtype=np.float64), cal

    def align_raw_async(self, raw, raw_offsets, shift, scale, seqs: bytes, seq_offsets, window: int = 3,
                        n_sigmas: float = 3.0, f32: bool = False, calc_probabilities: bool = True,
                        out: AlignBatchResult | None = None, calibration=None) -> AsyncBatch:
        """dyn_batch_align_raw_async: ``raw`` = the concatenated RAW [start:end) slices of the batch (float32 pA,
        int16 ADC or float64), or a LIST of per-read arrays (gathered by the library's helper threads, no copy here); normalisation + Hampel filter (segment.py:146-153) run on the device as the first
        stage of the asynchronous pipeline. ``calibration`` = (offset[n], scale[n]) with int16 ``raw``: the samples are
        ADC counts and picoampere = (float32(adc) + offset) * scale is formed on the device too (what pod5's
        ``signal_pa`` computes on the host). Returns at once."""
        raw, code, shift, scale, cal = self._raw_args(raw, shift, scale, calibration)
        raw_off = np.ascontiguousarray(raw_offsets, dtype=np.uint64)
        seq_off = np.ascontiguousarray(seq_offsets, dtype=np.uint64)
        n = len(raw_off) - 1
        cap = int(self._L.dyn_segment_capacity(self._h, n, _ptr(seq_off, N.c_u64_p)))
        if out is None or out.n != n or out.cap < cap:
            out = AlignBatchResult(n, cap + cap // 8)
        h = C.c_void_p()
        rc = self._L.dyn_batch_align_raw_async(self._h, n, raw.ctypes.data, code, _ptr(raw_off, N.c_u64_p),
                                               _ptr(cal[0], N.c_float_p) if code & 0xff == 3 else None,
                                               _ptr(cal[1], N.c_float_p) if code & 0xff == 3 else None, _ptr(shift, N.c_double_p), _ptr(scale, N.c_double_p), int(window),
                                               float(n_sigmas), int(bool(f32)), seqs, _ptr(seq_off, N.c_u64_p),
                                               int(bool(calc_probabilities)), C.byref(out._c), C.byref(h))
        if rc != N.DYN_OK:
            _raise(rc, self.last_error())
        return AsyncBatch(self, h, out, (raw, raw_off, shift, scale, seqs, seq_off, cal))

    def align_vbz_async(self, chunks, raw_offsets, shift, scale, seqs: bytes, seq_offsets, window: int = 3,
                        n_sigmas: float = 3.0, f32: bool = False, calc_probabilities: bool = True,
                        out: AlignBatchResult | None = None, calibration=None) -> AsyncBatch:
        """dyn_batch_align_vbz_async: the batch's signal as POD5 chunks that are still VBZ-compressed. ``chunks`` =
        (ptrs uint64[c], nbytes uint64[c], samples uint32[c], read_chunk_offsets uint64[n+1], slice_start uint64[n]);
        ``raw_offsets`` = prefix sums of the slice lengths. The library's helper threads decode straight into the pinned
        staging buffer."""
        ptrs, nbytes, samples, read_off, skip = (np.ascontiguousarray(x, dtype=d) for x, d in
                                                 zip(chunks, (np.uint64, np.uint64, np.uint32, np.uint64, np.uint64)))
        shift = np.ascontiguousarray(shift, dtype=np.float64)
        scale = np.ascontiguousarray(scale, dtype=np.float64)
        cal = (None, None)
        if calibration is not None:
            cal = (np.ascontiguousarray(calibration[0], dtype=np.float32), np.ascontiguousarray(calibration[1], dtype=np.float32))
        raw_off = np.ascontiguousarray(raw_offsets, dtype=np.uint64)
        seq_off = np.ascontiguousarray(seq_offsets, dtype=np.uint64)
        n = len(raw_off) - 1
        cap = int(self._L.dyn_segment_capacity(self._h, n, _ptr(seq_off, N.c_u64_p)))
        if out is None or out.n != n or out.cap < cap:
            out = AlignBatchResult(n, cap + cap // 8)
        h = C.c_void_p()
        rc = self._L.dyn_batch_align_vbz_async(self._h, n, ptrs.ctypes.data, _ptr(nbytes, N.c_u64_p),
                                               samples.ctypes.data_as(C.POINTER(C.c_uint32)), _ptr(read_off, N.c_u64_p),
                                               _ptr(skip, N.c_u64_p), _ptr(raw_off, N.c_u64_p),
                                               _ptr(cal[0], N.c_float_p) if calibration is not None else None,
                                               _ptr(cal[1], N.c_float_p) if calibration is not None else None,
                                               _ptr(shift, N.c_double_p), _ptr(scale, N.c_double_p), int(window), float(n_sigmas),
                                               int(bool(f32)), seqs, _ptr(seq_off, N.c_u64_p), int(bool(calc_probabilities)),
                                               C.byref(out._c), C.byref(h))
        if rc != N.DYN_OK:
            _raise(rc, self.last_error())
        return AsyncBatch(self, h, out, (ptrs, nbytes, samples, read_off, skip, raw_off, shift, scale, seqs, seq_off, cal))

    def train_raw_async(self, raw, raw_offsets, shift, scale, seqs: bytes, seq_offsets, window: int = 7,
                        n_sigmas: float = 5.0, f32: bool = True, pooled: bool = False,
                        emissions: bool = True, calibration=None) -> AsyncBatch:
        """dyn_batch_train_raw_async (train.py:163-170: float32 arithmetic, Hampel(7, 5 sigma) by default)."""
        raw, code, shift, scale, cal = self._raw_args(raw, shift, scale, calibration)
        raw_off = np.ascontiguousarray(raw_offsets, dtype=np.uint64)
        seq_off = np.ascontiguousarray(seq_offsets, dtype=np.uint64)
        n = len(raw_off) - 1
        cap = int(self._L.dyn_segment_capacity(self._h, n, _ptr(seq_off, N.c_u64_p)))
        out = TrainBatchResult(n, cap, self.num_kmers, pooled, emissions)
        h = C.c_void_p()
        rc = self._L.dyn_batch_train_raw_async(self._h, n, raw.ctypes.data, code, _ptr(raw_off, N.c_u64_p),
                                               _ptr(cal[0], N.c_float_p) if code & 0xff == 3 else None,
                                               _ptr(cal[1], N.c_float_p) if code & 0xff == 3 else None, _ptr(shift, N.c_double_p), _ptr(scale, N.c_double_p), int(window),
                                               float(n_sigmas), int(bool(f32)), seqs, _ptr(seq_off, N.c_u64_p),
                                               C.byref(out._c), _ptr(out.pooled, N.c_double_p) if pooled else None,
                                               C.byref(h))
        if rc != N.DYN_OK:
            _raise(rc, self.last_error())
        return AsyncBatch(self, h, out, (raw, raw_off, shift, scale, seqs, seq_off, cal))

    def segment_capacity(self, seq_offsets) -> int:
        """dyn_segment_capacity: rows to allocate for a batch with these sequence offsets."""
        so = np.ascontiguousarray(seq_offsets, dtype=np.uint64)
        return int(self._L.dyn_segment_capacity(self._h, len(so) - 1, _ptr(so, N.c_u64_p)))

    def train_async(self, signals, sig_offsets, seqs: bytes, seq_offsets, pooled: bool = False,
                    emissions: bool = True) -> AsyncBatch:
        """dyn_batch_train_async on packed inputs."""
        sig = np.ascontiguousarray(signals, dtype=np.float64)
        sig_off = np.ascontiguousarray(sig_offsets, dtype=np.uint64)
        seq_off = np.ascontiguousarray(seq_offsets, dtype=np.uint64)
        n = len(sig_off) - 1
        cap = int(self._L.dyn_segment_capacity(self._h, n, _ptr(seq_off, N.c_u64_p)))
        out = TrainBatchResult(n, cap, self.num_kmers, pooled, emissions)
        h = C.c_void_p()
        rc = self._L.dyn_batch_train_async(self._h, n, _ptr(sig, N.c_double_p), _ptr(sig_off, N.c_u64_p), seqs,
                                           _ptr(seq_off, N.c_u64_p), C.byref(out._c),
                                           _ptr(out.pooled, N.c_double_p) if pooled else None, C.byref(h))
        if rc != N.DYN_OK:
            _raise(rc, self.last_error())
        return AsyncBatch(self, h, out, (sig, sig_off, seqs, seq_off))

    def align_batch(self, signals: Sequence, sequences: Sequence[str], calc_probabilities: bool = True) -> AlignBatchResult:
        with self.batch(signals, sequences) as b:
            b.align(calc_probabilities)
            return b.fetch()

    def train_batch(self, signals: Sequence, sequences: Sequence[str], pooled: bool = False) -> TrainBatchResult:
        with self.batch(signals, sequences) as b:
            b.train()
            return b.fetch_train(pooled)

    # -- the reference's single-read surface ----------------------------------------------------
    def align(self, signal, sequence: str, calc_probabilities: bool = False) -> dict:
        """aligner_bindings.cpp:132-147,169-176 (single read = batch of one)."""
        return self.align_batch([signal], [sequence], calc_probabilities).read(0)

    def train(self, signal, sequence: str) -> dict:
        """aligner_bindings.cpp:149-163"""
        mean, sd = self.model_table()
        return self.train_batch([signal], [sequence]).read(0, mean, sd)


class RcclComm:
    """dyn_comm_*: the RCCL gather / all-reduce of a one-process-per-GPU job, below Python (no torch needed).
    ``RcclComm.unique_id()`` on rank 0, hand the 128 bytes to the other ranks, then ``RcclComm(id, rank, n_ranks, device)``
    on every rank (collective)."""

    ROW = np.dtype([("signal_pos", "<u4"), ("sequence_pos", "<u4"), ("probability", "<f8")])

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        err = C.create_string_buffer(ERRCAP)
        rc = N.lib().dyn_comm_unique_id(buf, err, ERRCAP)
        if rc != N.DYN_OK:
            _raise(rc, err.value.decode())
        return bytes(buf)

    def __init__(self, unique_id: bytes, rank: int, n_ranks: int, device: int = 0):
        self._L = N.lib()
        self.rank, self.n_ranks = int(rank), int(n_ranks)
        h = C.c_void_p()
        err = C.create_string_buffer(ERRCAP)
        idbuf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        rc = self._L.dyn_comm_create(idbuf, self.rank, self.n_ranks, int(device), C.byref(h), err, ERRCAP)
        if rc != N.DYN_OK:
            _raise(rc, err.value.decode())
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.dyn_comm_destroy(self._h)
            self._h = None

    __del__ = close

    def gather_rows(self, batch, root: int = 0, out=None):
        """(rows, counts): on ``root`` a structured array of every rank's segment rows in rank order; elsewhere rows is
        None. ``batch``: an aligned Batch or an AsyncBatch ticket. Two collectives: the counts (dyn_comm_gather_counts),
        then -- root's buffer allocated for exactly their sum -- the rows. A rank whose batch failed still takes part in
        both (with 0 rows) before its error is raised here, so its peers never block on it. ``out`` (root): a ROW array to
        receive into (e.g. ``pinned_empty(n, RcclComm.ROW)``: the copy off the device is then a plain DMA); one that is too
        small is replaced by a fresh array."""
        counts = np.zeros(self.n_ranks, dtype=np.uint64)
        rc1 = self._L.dyn_comm_gather_counts(self._h, batch._h, _ptr(counts, N.c_u64_p))
        err1 = (self._L.dyn_comm_last_error(self._h) or b"").decode()
        total = int(counts.sum())
        rows = None
        if self.rank == root:
            rows = out if (out is not None and out.dtype == self.ROW and out.size >= max(1, total)) else np.empty(max(1, total), dtype=self.ROW)
        rc = self._L.dyn_comm_gather_rows(self._h, batch._h, int(root), rows.ctypes.data if rows is not None else None,
                                          total if rows is not None else 0, None)
        if rc1 != N.DYN_OK:
            _raise(rc1, err1)
        if rc != N.DYN_OK:
            _raise(rc, (self._L.dyn_comm_last_error(self._h) or b"").decode())
        return (rows[:total] if rows is not None else None), counts

    def _check(self, rc):
        if rc != N.DYN_OK:
            _raise(rc, (self._L.dyn_comm_last_error(self._h) or b"").decode())

    def gather_bytes(self, payload: bytes, root: int = 0):
        """dyn_comm_gather_bytes: one bytes object per rank to ``root`` through the code path of ``gather_rows`` (count
        all-gather, grouped ncclSend / ncclRecv). Returns the list of per-rank payloads on ``root``, None elsewhere."""
        buf = np.frombuffer(payload, dtype=np.uint8) if payload else np.zeros(0, dtype=np.uint8)
        counts = np.zeros(self.n_ranks, dtype=np.uint64)
        self._check(self._L.dyn_comm_gather_bytes(self._h, buf.ctypes.data if buf.size else None, buf.size, int(root), _ptr(counts, N.c_u64_p)))
        if self.rank != root:
            return None
        total = int(counts.sum())
        out = np.empty(max(1, total), dtype=np.uint8)
        self._check(self._L.dyn_comm_gathered_bytes(self._h, out.ctypes.data, total))
        offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        return [out[offs[r]:offs[r + 1]].tobytes() for r in range(self.n_ranks)]

    def allreduce(self, x: np.ndarray, op: str = "sum") -> np.ndarray:
        """dyn_comm_allreduce_f64: elementwise sum / max over the ranks of a float64 vector (a copy is returned)."""
        out = np.ascontiguousarray(x, dtype=np.float64).copy()
        self._check(self._L.dyn_comm_allreduce_f64(self._h, _ptr(out, N.c_double_p), out.size, {"sum": 0, "max": 1}[op]))
        return out

    def allreduce_pooled(self, batch, num_kmers: int) -> np.ndarray:
        out = np.empty(3 * int(num_kmers))
        rc = self._L.dyn_comm_allreduce_pooled(self._h, batch._h, _ptr(out, N.c_double_p))
        if rc != N.DYN_OK:
            _raise(rc, (self._L.dyn_comm_last_error(self._h) or b"").decode())
        return out


class MultiAligner:
    """dyn_multi_*: one handle driving several GPUs of a node from ONE process. Reads of a batch are cut into
    contiguous ranges of equal lattice work, one per device; results come back in the layout of a single-device
    batch. ``devices``: HIP ordinals (an ordinal may repeat, e.g. ``[0, 0]`` = two pipelines on one GPU)."""

    def __init__(self, model_file: str, pore, devices: Sequence[int], mode: str = "basic", threads: int = 1, band: int = 400):
        if isinstance(pore, str):
            pore = pore_type(pore)
        self._L = N.lib()
        ids = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        err = C.create_string_buffer(ERRCAP)
        rc = self._L.dyn_multi_create(str(model_file).encode(), int(pore), str(mode).encode(), int(threads), int(band), ids,
                                      len(devices), C.byref(h), err, ERRCAP)
        if rc != N.DYN_OK:
            _raise(rc, err.value.decode())
        self._h = h
        info = N.DynInfo()
        self._L.dyn_aligner_info(self._L.dyn_multi_handle(h, 0), C.byref(info))
        self.kmer_size, self.num_kmers, self.n_devices = int(info.kmer_size), int(info.num_kmers), len(devices)

    def close(self):
        if getattr(self, "_h", None):
            self._L.dyn_multi_destroy(self._h)
            self._h = None

    __del__ = close

    def last_error(self) -> str:
        return (self._L.dyn_multi_last_error(self._h) or b"").decode(errors="replace")

    def _cap(self, seq_off) -> int:
        return int(self._L.dyn_segment_capacity(self._L.dyn_multi_handle(self._h, 0), len(seq_off) - 1, _ptr(seq_off, N.c_u64_p)))

    def align_batch(self, signals: Sequence, sequences: Sequence[str], calc_probabilities: bool = True) -> AlignBatchResult:
        sig, sig_off, seqs, seq_off = _pack(signals, sequences)
        out = AlignBatchResult(len(sequences), self._cap(seq_off))
        rc = self._L.dyn_multi_align_batch(self._h, out.n, _ptr(sig, N.c_double_p), _ptr(sig_off, N.c_u64_p), seqs,
                                           _ptr(seq_off, N.c_u64_p), int(bool(calc_probabilities)), C.byref(out._c))
        if rc != N.DYN_OK:
            _raise(rc, self.last_error())
        return out

    def train_batch(self, signals: Sequence, sequences: Sequence[str], pooled: bool = False) -> TrainBatchResult:
        sig, sig_off, seqs, seq_off = _pack(signals, sequences)
        out = TrainBatchResult(len(sequences), self._cap(seq_off), self.num_kmers, pooled)
        rc = self._L.dyn_multi_train_batch(self._h, out.n, _ptr(sig, N.c_double_p), _ptr(sig_off, N.c_u64_p), seqs,
                                           _ptr(seq_off, N.c_u64_p), C.byref(out._c),
                                           _ptr(out.pooled, N.c_double_p) if pooled else None)
        if rc != N.DYN_OK:
            _raise(rc, self.last_error())
        return out
