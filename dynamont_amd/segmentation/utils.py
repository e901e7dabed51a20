"""Counterpart of the reference's ``src/dynamont/segmentation/utils.py`` (names, arguments and
outputs kept; plotting helpers are out of scope).

Parity pins: tests/test_harness.py against tests/golden/g6_harness.npz (bytes produced by the
reference's own functions) and the reference tests' known-answer vectors
(tests/test_segment.py:179-200, tests/test_utils.py:7-114 of the reference).
"""
from __future__ import annotations

import sys
from os.path import dirname, join

import numpy as np

from dynamont_amd import Aligner

MAD_TO_SIGMA = 1.4826


def hampel(signal: np.ndarray, WINDOW: int = 3, n_sigmas: float = 3.0) -> None:
    """In-place Hampel filter with the reference's exact footprint (utils.py:16-43):
    windows are taken from the UNFILTERED signal; centre i = WINDOW//2 + w for window w;
    ``len - WINDOW - (WINDOW even)`` windows are examined, so the first WINDOW//2 and the last
    WINDOW//2 + 1 samples are never touched; a centre is replaced by its window median when
    |x - median| > n_sigmas * 1.4826 * MAD. No-op when size <= WINDOW."""
    size = signal.size
    if size <= WINDOW:
        return
    n_windows = size - WINDOW - (1 if WINDOW % 2 == 0 else 0)
    if n_windows <= 0:
        return
    src = signal.copy()
    half = WINDOW // 2
    win = np.lib.stride_tricks.sliding_window_view(src, WINDOW)[:n_windows]

    def row_median(a):
        s = np.sort(a, axis=1)
        if WINDOW % 2:
            return s[:, half].copy()
        return (s[:, half - 1] + s[:, half]) / 2.0

    med = row_median(win)
    mad = row_median(np.abs(win - med[:, None]))
    centre = src[half:half + n_windows]
    outlier = np.abs(centre - med) > n_sigmas * (MAD_TO_SIGMA * mad)
    view = signal[half:half + n_windows]
    view[outlier] = med[outlier]


def cnt_nts(sequence):
    """utils.py:45-61"""
    return {b: sequence.count(b) for b in "ACGT"}


def cnt_nts_ratios(sequence):
    """utils.py:63-79"""
    n = len(sequence)
    return {b: sequence.count(b) / n for b in "ACGT"}


class SegmentationError(Exception):
    """Raised when no segmentation was calculated for a read (utils.py:81-86)."""

    def __init__(self, read) -> None:
        self.read = read
        self.message = f"No segmentation calculated for {read}"
        super().__init__(self.message)


def get_model(pore: str) -> str:
    """Default model path per pore (utils.py:96-117); unknown keys are returned as a path."""
    models = {
        "rna002": "models/rna/rna002/rna002_5mer.model",
        "rna004": "models/rna/rna004/rna004_9mer.model",
        "dna_r10_260bps": "models/dna/r10.4.1/dna_r10.4.1_e8.2_260bps.model",
        "dna_r10_400bps": "models/dna/r10.4.1/dna_r10.4.1_e8.2_400bps.model",
    }
    base_dir = dirname(dirname(dirname(__file__)))
    return join(base_dir, models.get(pore, pore))


def read_kmer_model(file: str) -> dict:
    """{kmer: (mean, stdev)} from the TSV columns ``kmer, level_mean, level_stdv`` (utils.py:119-134)."""
    out = {}
    with open(file) as f:
        header = f.readline().rstrip("\n").split("\t")
        ik, im, isd = header.index("kmer"), header.index("level_mean"), header.index("level_stdv")
        for line in f:
            if not line.strip():
                continue
            p = line.rstrip("\n").split("\t")
            out[p[ik]] = (float(p[im]), float(p[isd]))
    return out


def write_kmer_model(file: str, kmer_model: dict) -> None:
    """utils.py:136-152: header + f'{kmer}\\t{mean}\\t{stdev}\\n' rows (repr-style floats)."""
    with open(file, "w") as w:
        w.write("kmer\tlevel_mean\tlevel_stdv\n")
        for kmer, (mean, stdev) in ((k, (v[0], v[1])) for k, v in kmer_model.items()):
            w.write(f"{kmer}\t{mean}\t{stdev}\n")


def write_kmer_model_arrays(file: str, kmers: bytes, k: int, mean, stdev) -> None:
    """write_kmer_model for values that already are arrays in file order (``kmers``: the rows' k-mer strings back to
    back): the same bytes, formatted by the library (dyn_format_model) -- 262 144 rows take 25 ms instead of 0.25 s."""
    import ctypes as C
    from .. import _native as N
    mean = np.ascontiguousarray(mean, dtype=np.float64)
    stdev = np.ascontiguousarray(stdev, dtype=np.float64)
    n = len(mean)
    assert len(stdev) == n and len(kmers) == n * k
    L = N.lib()
    cap = int(L.dyn_format_model(kmers, k, mean.ctypes.data_as(N.c_double_p), stdev.ctypes.data_as(N.c_double_p), n, None, 0))
    buf = C.create_string_buffer(cap)
    used = int(L.dyn_format_model(kmers, k, mean.ctypes.data_as(N.c_double_p), stdev.ctypes.data_as(N.c_double_p), n, buf, cap))
    with open(file, "wb") as w:
        w.write(memoryview(buf)[:used])


def _make_native_aligner(model: str, params: dict, mode: str, device=None) -> Aligner:
    """utils.py:154-161"""
    pore = params.get("r") or params.get("pore")
    if pore is None:
        raise ValueError("Missing pore type in segmentation parameters")
    return Aligner(model, pore, mode=mode, threads=int(params.get("t", 1)), band=int(params.get("band", 400)),
                   device=device)


def _state_char(state) -> str:
    if isinstance(state, bytes):
        return state.decode("ascii")
    if isinstance(state, str):
        return state
    if isinstance(state, (int, np.integer)):
        return chr(int(state))
    return str(state)


def segmentation_to_string(result: dict, readid: str, signalid: str, sigOffset: int, lastIndex: int,
                           read: str, kmerSize: int, rna: bool) -> bytes:
    """CSV rows of one read (utils.py:193-232): ``readid,signalid,start,end,basepos,base,motif,
    state,posterior_probability,polish``. ``read`` is in aligner orientation; for RNA the motif is
    reversed and basepos flipped AFTER ``base`` was taken."""
    seq_pos = result["sequence_positions"]
    sig_pos = result["signal_positions"]
    probs = result["probabilities"]
    states = result["states"]
    polishes = result.get("polishes")
    n = len(seq_pos)
    half = kmerSize // 2
    L = len(read)
    rows = []
    for i in range(n):
        bp = int(seq_pos[i])
        start = int(sig_pos[i]) + sigOffset
        end = int(sig_pos[i + 1]) + sigOffset if i + 1 < n else lastIndex
        motif = read[max(0, bp - half):min(L, bp + half + 1)]
        base = read[bp]
        polish = str(polishes[i]) if polishes is not None and polishes[i] else "NA"
        if rna:
            motif = motif[::-1]
            bp = L - bp - 1
        rows.append(f"{readid},{signalid},{start},{end},{bp},{base},{motif},{_state_char(states[i])},"
                    f"{float(probs[i]):.6f},{polish}\n")
    return "".join(rows).encode("utf-8")


def train_transition_emission(signal, read, params, script, model, signalid, aligner: Aligner | None = None):
    """Single-read form of utils.py:163-182, kept for API parity. The k-mer <-> parameter pairing
    is by k-mer CODE (``Aligner.kmer_of_code``); the reference zips file order with code order,
    which mislabels RNA models."""
    try:
        al = aligner or _make_native_aligner(model, params, script)
        if script == "basic":
            res = al.train_batch([signal], [read])
            if res.status[0]:
                raise RuntimeError(res.error(0))
            t = res.transitions[:3]
            code, mean, sd = res.sparse(0)
            new_models = dict(read_kmer_model(model))
            for c, m, s in zip(code, mean, sd):
                new_models[kmer_of_code(int(c), al.kmer_size, al.rna)] = (float(m), float(s))
            return {"m1": float(t[0]), "e1": float(t[1]), "e2": float(t[2])}, new_models, float(res.Z[0])
        result = al.align(signal, read, calc_probabilities=False)
        trans = {k: float(v) for k, v in params.items() if k not in {"r", "t", "band"}}
        return trans, read_kmer_model(model), float(result["Z"])
    except Exception as error:
        print(f"error: native, {error} T: {len(signal)} N: {len(read)} Sid: {signalid}", file=sys.stderr)
        return signalid


def calcZ(signal, read, params, script, model, signalid, aligner: Aligner | None = None):
    """utils.py:184-191"""
    try:
        al = aligner or _make_native_aligner(model, params, script)
        return float(al.align(signal, read, calc_probabilities=False)["Z"])
    except Exception as error:
        print(f"error: native, {error} T: {len(signal)} N: {len(read)} Sid: {signalid}", file=sys.stderr)
        return signalid


def kmer_of_code(code: int, k: int, rna: bool) -> str:
    """intToKmer (aligner.cpp:222-239): base-4 digits, reversed back to 5'->3' for RNA."""
    s = []
    for _ in range(k):
        s.append("ACGT"[code % 4])
        code //= 4
    kmer = "".join(reversed(s))
    return kmer[::-1] if rna else kmer
