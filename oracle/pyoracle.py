"""TEST INFRASTRUCTURE ONLY -- ctypes front-ends for the two CPU checkers.

* ``Oracle``    -> oracle/_build/libnt_oracle.so  (our plain-C restatement, nt_oracle.c)
* ``Reference`` -> oracle/_ref/libdynamont_ref.so (the reference's own sources compiled by
                   oracle/Makefile + ref_shim.cpp). Exists only where it was built in the
                   authoring container; it travels to the GPU box as a binary.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Both raise RuntimeError(message) with the reference's message text on failure, like the
pybind11 module does (aligner_bindings.cpp:132-165).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "_build", "libnt_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libdynamont_ref.so")

_dp = C.POINTER(C.c_double)
_u64p = C.POINTER(C.c_uint64)


def build(target: str = "all") -> None:
    subprocess.run(["make", "-s", "-C", HERE, target], check=True)


def _ptr(a, t):
    return a.ctypes.data_as(t)


class _Base:
    ERRCAP = 4096

    def _err(self):
        return C.create_string_buffer(self.ERRCAP)


class Oracle(_Base):
    def __init__(self, model_path: str, pore: int, band: int = 400):
        if not os.path.exists(ORACLE_SO):
            build("oracle")
        L = self.lib = C.CDLL(ORACLE_SO)
        L.nto_model_load.restype = C.c_void_p
        L.nto_model_load.argtypes = [C.c_char_p, C.c_int, C.c_uint64, C.c_char_p, C.c_uint64]
        L.nto_model_free.argtypes = [C.c_void_p]
        L.nto_model_num_kmers.restype = C.c_uint64
        L.nto_model_num_kmers.argtypes = [C.c_void_p]
        L.nto_model_kmer_size.argtypes = [C.c_void_p]
        L.nto_model_table.argtypes = [C.c_void_p, _dp]
        L.nto_sequence_to_kmers.restype = C.c_int64
        L.nto_sequence_to_kmers.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int32), C.c_char_p, C.c_uint64]
        L.nto_compute_bounds.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_int64), _u64p, _u64p]
        L.nto_align.restype = C.c_int64
        L.nto_align.argtypes = [C.c_void_p, _dp, C.c_uint64, C.c_char_p, C.c_int, _dp, _u64p, _u64p, _dp,
                                C.c_char_p, C.c_char_p, C.c_uint64]
        L.nto_train.restype = C.c_int64
        L.nto_train.argtypes = [C.c_void_p, _dp, C.c_uint64, C.c_char_p, _dp, _dp, _dp, _dp, _dp,
                                C.c_char_p, C.c_uint64]
        L.nto_debug_fb.restype = C.c_int64
        L.nto_debug_fb.argtypes = [C.c_void_p, _dp, C.c_uint64, C.c_char_p, _dp, _dp, _dp, _dp, C.c_uint64,
                                   C.c_char_p, C.c_uint64]
        L.nto_log_normal_pdf.restype = C.c_double
        L.nto_log_normal_pdf.argtypes = [C.c_double] * 3
        L.nto_log_plus.restype = C.c_double
        L.nto_log_plus.argtypes = [C.c_double] * 2
        L.nto_last_decision_margin.restype = C.c_double
        L.nto_last_decision_margin_distinct.restype = C.c_double
        L.nto_last_decision_margin_at.argtypes = [_u64p, _u64p, _dp, _dp]
        err = self._err()
        self.h = L.nto_model_load(model_path.encode(), pore, band, err, self.ERRCAP)
        if not self.h:
            raise RuntimeError(err.value.decode())
        self.num_kmers = int(L.nto_model_num_kmers(self.h))
        self.k = int(L.nto_model_kmer_size(self.h))

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.nto_model_free(self.h)
            self.h = None

    def table(self):
        out = np.empty(2 * self.num_kmers)
        self.lib.nto_model_table(self.h, _ptr(out, _dp))
        return out[0::2].copy(), out[1::2].copy()

    def kmers(self, seq: str) -> np.ndarray:
        out = np.empty(max(1, len(seq)), dtype=np.int32)
        err = self._err()
        n = self.lib.nto_sequence_to_kmers(self.h, seq.encode(), _ptr(out, C.POINTER(C.c_int32)), err, self.ERRCAP)
        if n < 0:
            raise RuntimeError(err.value.decode())
        return out[:n].copy()

    def bounds(self, T: int, N: int, bw: int):
        s = np.empty(T, dtype=np.int64)
        a = np.empty(T, dtype=np.uint64)
        b = np.empty(T, dtype=np.uint64)
        self.lib.nto_compute_bounds(T, N, bw, _ptr(s, C.POINTER(C.c_int64)), _ptr(a, _u64p), _ptr(b, _u64p))
        return s, a, b

    def align(self, signal, seq: str, calc: bool = False) -> dict:
        sig = np.ascontiguousarray(signal, dtype=np.float64)
        cap = max(1, len(seq))
        sp = np.empty(cap, dtype=np.uint64)
        gp = np.empty(cap, dtype=np.uint64)
        pr = np.empty(cap)
        st = C.create_string_buffer(cap)
        Z = C.c_double()
        err = self._err()
        n = self.lib.nto_align(self.h, _ptr(sig, _dp), len(sig), seq.encode(), int(calc), C.byref(Z),
                               _ptr(sp, _u64p), _ptr(gp, _u64p), _ptr(pr, _dp), st, err, self.ERRCAP)
        if n < 0:
            raise RuntimeError(err.value.decode())
        return dict(Z=Z.value, sequence_positions=sp[:n].copy(), signal_positions=gp[:n].copy(),
                    probabilities=pr[:n].copy(), states=[chr(c) for c in st.raw[:n]])

    def last_decision_margin(self) -> float:
        """Smallest |vM - vE| over the on-path traceback decisions of the last align(calc=True)."""
        return float(self.lib.nto_last_decision_margin())

    def last_decision_margin_distinct(self) -> float:
        """The same over decisions between columns with different k-mers (structural ties excluded)."""
        return float(self.lib.nto_last_decision_margin_distinct())

    def last_decision_margin_at(self):
        """(row, column, vM, vE) of the smallest margin of the last align(calc=True)."""
        t, n, vm, ve = C.c_uint64(), C.c_uint64(), C.c_double(), C.c_double()
        self.lib.nto_last_decision_margin_at(C.byref(t), C.byref(n), C.byref(vm), C.byref(ve))
        return t.value, n.value, vm.value, ve.value

    def train(self, signal, seq: str, dense: bool = True) -> dict:
        sig = np.ascontiguousarray(signal, dtype=np.float64)
        K = self.num_kmers
        em = np.empty(2 * K) if dense else None
        stats = np.empty(3 * K)
        ls = np.empty(2)
        tr = np.empty(3)
        Z = C.c_double()
        err = self._err()
        n = self.lib.nto_train(self.h, _ptr(sig, _dp), len(sig), seq.encode(), C.byref(Z), _ptr(tr, _dp),
                               _ptr(em, _dp) if dense else None, _ptr(stats, _dp), _ptr(ls, _dp), err,
                               self.ERRCAP)
        if n < 0:
            raise RuntimeError(err.value.decode())
        out = dict(Z=Z.value, m1=tr[0], e1=tr[1], e2=tr[2], weight=stats[:K].copy(), sum=stats[K:2 * K].copy(),
                   sumsq=stats[2 * K:].copy(), log_m1=ls[0], log_e2=ls[1])
        if dense:
            out["mean"] = em[0::2].copy()
            out["stdev"] = em[1::2].copy()
        return out

    def debug_fb(self, signal, seq: str, W: int):
        sig = np.ascontiguousarray(signal, dtype=np.float64)
        T = len(sig) + 1
        arrs = [np.empty(T * W) for _ in range(4)]
        err = self._err()
        n = self.lib.nto_debug_fb(self.h, _ptr(sig, _dp), len(sig), seq.encode(), *[_ptr(a, _dp) for a in arrs],
                                  T * W, err, self.ERRCAP)
        if n < 0:
            raise RuntimeError(err.value.decode())
        assert n == W, (n, W)
        return dict(zip(("fE", "bE", "fM", "bM"), [a.reshape(T, W) for a in arrs]))


def reference_available() -> bool:
    return os.path.exists(REF_SO)


class Reference(_Base):
    """The compiled reference (NTAligner) through oracle/ref_shim.cpp."""

    def __init__(self, model_path: str, pore: int, band: int = 400):
        if not reference_available():
            raise FileNotFoundError(REF_SO)
        L = self.lib = C.CDLL(REF_SO)
        L.ref_create.restype = C.c_void_p
        L.ref_create.argtypes = [C.c_char_p, C.c_int, C.c_uint64, C.c_char_p, C.c_uint64]
        L.ref_destroy.argtypes = [C.c_void_p]
        L.ref_align.restype = C.c_int64
        L.ref_align.argtypes = [C.c_void_p, _dp, C.c_uint64, C.c_char_p, C.c_int, _dp, _u64p, _u64p, _dp,
                                C.c_char_p, C.c_char_p, C.c_uint64]
        L.ref_train.restype = C.c_int64
        L.ref_train.argtypes = [C.c_void_p, _dp, C.c_uint64, C.c_char_p, _dp, _dp, _dp, C.c_uint64,
                                C.c_char_p, C.c_uint64]
        err = self._err()
        self.h = L.ref_create(model_path.encode(), pore, band, err, self.ERRCAP)
        if not self.h:
            raise RuntimeError(err.value.decode())

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ref_destroy(self.h)
            self.h = None

    def align(self, signal, seq: str, calc: bool = False) -> dict:
        sig = np.ascontiguousarray(signal, dtype=np.float64)
        cap = max(1, len(seq))
        sp = np.empty(cap, dtype=np.uint64)
        gp = np.empty(cap, dtype=np.uint64)
        pr = np.empty(cap)
        st = C.create_string_buffer(cap)
        Z = C.c_double()
        err = self._err()
        n = self.lib.ref_align(self.h, _ptr(sig, _dp), len(sig), seq.encode(), int(calc), C.byref(Z),
                               _ptr(sp, _u64p), _ptr(gp, _u64p), _ptr(pr, _dp), st, err, self.ERRCAP)
        if n < 0:
            raise RuntimeError(err.value.decode())
        return dict(Z=Z.value, sequence_positions=sp[:n].copy(), signal_positions=gp[:n].copy(),
                    probabilities=pr[:n].copy(), states=[chr(c) for c in st.raw[:n]])

    def train(self, signal, seq: str, num_kmers: int) -> dict:
        sig = np.ascontiguousarray(signal, dtype=np.float64)
        em = np.empty(2 * num_kmers)
        tr = np.empty(3)
        Z = C.c_double()
        err = self._err()
        n = self.lib.ref_train(self.h, _ptr(sig, _dp), len(sig), seq.encode(), C.byref(Z), _ptr(tr, _dp),
                               _ptr(em, _dp), 2 * num_kmers, err, self.ERRCAP)
        if n < 0:
            raise RuntimeError(err.value.decode())
        return dict(Z=Z.value, m1=tr[0], e1=tr[1], e2=tr[2], mean=em[0::2].copy(), stdev=em[1::2].copy())
