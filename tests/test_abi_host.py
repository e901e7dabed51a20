"""CPU: the C-ABI library loads, exports every declared symbol, and its host-side contract
(model loader, validation, k-mer coding, error texts) matches the oracle and the goldens.
No compute call is possible without a GPU -- and that must fail loudly, never fall back."""
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from dynamont_amd import Aligner, PoreType, pore_type, synth
from dynamont_amd import _native as N
from oracle.pyoracle import Oracle

pytestmark = pytest.mark.usefixtures("native_lib", "oracle_built")


def test_every_declared_symbol_is_exported(native_lib):
    hdr = open(os.path.join(ROOT, "include", "dynamont_mi.h")).read()
    declared = set(re.findall(r"\b(dyn_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "header parse failed"
    assert declared == set(N.SIGNATURES), declared ^ set(N.SIGNATURES)
    for name in declared:
        assert getattr(native_lib, name) is not None


def test_header_cites_the_reference_interfaces():
    hdr = open(os.path.join(ROOT, "include", "dynamont_mi.h")).read()
    for needle in ("aligner_bindings.cpp", "NT_aligner_api.cpp", "aligner.cpp", "aligner.hpp"):
        assert needle in hdr


def test_pore_type_surface():
    assert [p.name for p in PoreType] == ["RNA002", "RNA004", "DNA_R9", "DNA_R10_260", "DNA_R10_400"]
    for s, v in (("rna002", 0), ("rna004", 1), ("dna_r9", 2), ("dna_r10_260bps", 3), ("dna_r10_400bps", 4)):
        assert pore_type(s) == PoreType(v)
    with pytest.raises(ValueError, match="Unknown pore type: foo"):
        pore_type("foo")


def test_constructor_errors_match_reference(models):
    d = json.load(open(os.path.join(GOLDEN, "g5_failures.json")))
    for c in d["ctor"]:
        path = {"syn5": models["syn5"], "syn9": models["syn9"]}.get(c["model"], c["model"])
        with pytest.raises(RuntimeError) as e:
            Aligner(path, PoreType(c["pore"]), device="host")
        assert str(e.value) == c["message"]
    with pytest.raises(ValueError, match="Unknown aligner mode: fancy"):
        Aligner(models["syn5"], "rna002", mode="fancy", device="host")
    with pytest.raises(ValueError, match="Unknown pore type: nope"):
        Aligner(models["syn5"], "nope", device="host")
    # modes "resquiggle" / "ntk" construct like the reference's do (aligner_bindings.cpp:46-49); what such a handle
    # answers is pinned by G11 on the GPU box (tests/test_gpu_parity.py)
    for mode in ("resquiggle", "ntk", "basic", "nt"):
        Aligner(models["syn5"], "rna002", mode=mode, device="host").close()
    # band: any value constructs, as in the reference (aligner.cpp:21). The register sweeps hold 448 band slots per lattice row;
    # reads whose half band min(band / 2, columns / 2) exceeds their 223 take the generic kernel (4 096 band columns per row:
    # half bands up to 2 046), and only beyond THAT a read gets a per-read status
    for band in (447, 448, 1000, 4093, 4094, 10 ** 6):
        al = Aligner(models["syn5"], "rna002", band=band, device="host")
        assert al.info.half_band == band // 2 and al.info.max_half_band == 2046
        seqs = ["ACGTA" * 20, "ACGTA" * 90, "ACGTA" * 91, "ACGTA" * 400, "ACGTA" * 819, "ACGTA" * 820]   # .. 1 996, 4 091, 4 096 k-mers
        status, _, _ = al.validate([10000] * 6, seqs)
        wide = [0, 0, 0, 0, 0, 11] if band // 2 > 2046 else [0] * 6       # 4 092 columns: half band 2 046; 4 097: 2 048
        assert list(status) == wide, (band, list(status))
        al.close()
    from dynamont_amd._dynamont import read_error_message
    assert read_error_message(11) == "Band wider than this build's 4096 band columns for a read of this length"


@pytest.mark.parametrize("pore,key", [("rna002", "syn5"), ("rna004", "syn9"), ("dna_r9", "syn5"),
                                      ("dna_r10_260bps", "syn9"), ("dna_r10_400bps", "syn9")])
def test_model_loader_matches_oracle(models, pore, key):
    al = Aligner(models[key], pore, device="host")
    orc = Oracle(models[key], synth.PORES[pore][0])
    m, s = al.model_table()
    om, os_ = orc.table()
    assert np.array_equal(m, om) and np.array_equal(s, os_)
    assert al.kmer_size == orc.k and al.num_kmers == orc.num_kmers
    assert al.rna == synth.PORES[pore][1]
    # and the generator's code-order view agrees with both (RNA k-mers are reversed on load)
    _, fm, fs = synth.read_model_file(models[key])
    gm, gs = synth.code_order_table(fm, fs, al.kmer_size, al.rna)
    assert np.array_equal(gm, m) and np.array_equal(gs, s)
    # default log transitions (NT_aligner_api.cpp:36-86)
    want = {"rna002": (0.019889650396799997, 0.9801103496029998), "dna_r9": (1.0, 1.0)}.get(
        pore, (0.031111753637096777, 0.9688882463622581))
    assert al.info.log_m1 == np.log(want[0]) and al.info.log_e2 == np.log(want[1]) and al.info.log_e1 == 0.0
    assert al.info.half_band == 200


def test_validation_and_messages_match_goldens(models):
    d = json.load(open(os.path.join(GOLDEN, "g5_failures.json")))
    al = Aligner(models["syn5"], "rna002", device="host")
    cases = d["align"]
    status, msgs, kmers = al.validate([len(c["signal"]) for c in cases], [c["sequence"] for c in cases])
    orc = Oracle(models["syn5"], 0)
    for c, st, msg, km in zip(cases, status, msgs, kmers):
        if c["ok"]:
            assert st == 0 and msg is None
            assert np.array_equal(km, orc.kmers(c["sequence"]))
        else:
            assert st != 0 and msg == c["message"], (c["name"], msg)


def test_kmer_coding_random(models):
    rng = np.random.default_rng(5)
    for pore, key in (("rna004", "syn9"), ("dna_r9", "syn5")):
        al = Aligner(models[key], pore, device="host")
        orc = Oracle(models[key], synth.PORES[pore][0])
        seqs = ["".join(rng.choice(list("ACGTacgtUu"), size=int(n))) for n in rng.integers(al.kmer_size, 300, 20)]
        st, _, km = al.validate([10 ** 6] * len(seqs), seqs)
        assert (st == 0).all()
        for s, k in zip(seqs, km):
            assert np.array_equal(k, orc.kmers(s))


def test_compute_without_gpu_fails_loudly(models):
    al = Aligner(models["syn5"], "rna002", device="host")
    with pytest.raises(RuntimeError, match="no CPU compute path"):
        al.align(np.zeros(100), "ACGTACGTAC", True)
    with pytest.raises(RuntimeError, match="no CPU compute path"):
        al.train(np.zeros(100), "ACGTACGTAC")
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU compute path"):
            Aligner(models["syn5"], "rna002")


def test_signal_must_be_one_dimensional(models):
    al = Aligner(models["syn5"], "rna002", device="host")
    with pytest.raises(ValueError, match="Signal must be a one-dimensional array"):
        al.align(np.zeros((4, 4)), "ACGTACGTAC")


def test_queue_planner_for_page_starved_launches(native_lib):
    """dyn_plan_queue (no GPU): for a config-3-like batch (4 096 reads of 10 k - 100 k rows) and a pool that holds
    ~80 % of the 1 024 longest lattices, the planned queue is a permutation of the reads, never worse than
    longest-first in the host replay, and clearly better here; with enough pages nothing is reordered."""
    import ctypes as C
    rng = np.random.default_rng(3)
    rows = np.sort((rng.integers(800, 8001, size=4096) * 12.55).astype(np.uint64))[::-1].copy()
    pages = ((rows + 1 + 255) // 256).astype(np.uint32)
    order = np.zeros(4096, dtype=np.uint32)
    lpt, planned = C.c_uint64(), C.c_uint64()

    def plan(pool):
        rc = native_lib.dyn_plan_queue(4096, pages.ctypes.data_as(C.POINTER(C.c_uint32)), rows.ctypes.data_as(N.c_u64_p), 1024,
                                       pool, order.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(lpt), C.byref(planned))
        assert rc == N.DYN_OK
        return lpt.value, planned.value

    a, b = plan(294713)
    ideal = rows.sum() / 1024
    assert sorted(order.tolist()) == list(range(4096))
    assert b <= a and a / ideal > 1.10 and b / ideal < 1.08, (a / ideal, b / ideal)
    assert not np.array_equal(order, np.arange(4096))
    a, b = plan(int(pages[:1024].sum()) + 10)          # every wave gets its lattice: longest-first stays
    assert a == b and np.array_equal(order, np.arange(4096))
    assert a / ideal < 1.03


def test_queue_planner_properties_on_random_batches(native_lib):
    """Whatever the length distribution and the pool size (as long as the longest read fits): the planned queue is a
    permutation, every read gets served in the host replay (finite makespan) and the plan is never worse than
    longest-first."""
    import ctypes as C
    rng = np.random.default_rng(11)
    for trial in range(40):
        n = int(rng.integers(1, 3000))
        kind = trial % 4
        if kind == 0:
            rows = rng.integers(2000, 120000, size=n)
        elif kind == 1:
            rows = (rng.lognormal(10.0, 0.8, size=n)).astype(np.int64) + 50
        elif kind == 2:
            rows = np.full(n, int(rng.integers(100, 50000)))
        else:
            rows = np.concatenate([rng.integers(90000, 100000, size=n // 10 + 1), rng.integers(500, 3000, size=n)])[:n]
        rows = np.sort(rows.astype(np.uint64))[::-1].copy()
        pages = ((rows + 1 + 255) // 256).astype(np.uint32)
        n_slots = int(rng.choice([4, 64, 1024]))
        top = int(pages[:n_slots].sum())
        pool = int(max(int(pages[0]), rng.uniform(0.15, 1.3) * top))
        order = np.zeros(n, dtype=np.uint32)
        lpt, planned = C.c_uint64(), C.c_uint64()
        rc = native_lib.dyn_plan_queue(n, pages.ctypes.data_as(C.POINTER(C.c_uint32)), rows.ctypes.data_as(N.c_u64_p), n_slots,
                                       pool, order.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(lpt), C.byref(planned))
        assert rc == N.DYN_OK
        assert sorted(order.tolist()) == list(range(n)), trial
        assert lpt.value != 2 ** 64 - 1 and planned.value <= lpt.value, (trial, lpt.value, planned.value)
        assert planned.value >= int(rows[0])


def test_model_file_text_matches_the_python_writer(tmp_path):
    """dyn_format_model: the bytes of write_kmer_model (f"{kmer}\\t{mean}\\t{stdev}\\n" with numpy float64 values, i.e.
    Python's repr of a float) for 5-mers and 9-mers -- magnitudes from 1e-12 to 1e22, zero, negative zero, the
    fixed / scientific switch points, a denormal, the largest double, inf, nan."""
    import numpy as np
    from dynamont_amd import synth
    from dynamont_amd.segmentation.utils import write_kmer_model, write_kmer_model_arrays
    rng = np.random.default_rng(1)
    for k in (5, 9):
        names = synth.kmer_strings(k)
        n = len(names)
        mean = rng.standard_normal(n) * 10.0 ** rng.integers(-12, 22, n)
        sd = np.abs(rng.standard_normal(n)) * 10.0 ** rng.integers(-6, 3, n)
        edge = [0.0, -0.0, 1e16, 1e-5, 0.0001, 123456789012345678.0, 1e15, 9999999999999998.0, 0.001, 1.0, 100.0, 1e22,
                5e-324, 1.7976931348623157e308, np.inf, -np.inf, np.nan, 0.1, 1 / 3, 2.5e-5, 12345.678, 1e-4, 9.999e-5]
        mean[:len(edge)] = edge
        write_kmer_model(str(tmp_path / "py.model"), {names[i]: [mean[i], sd[i]] for i in range(n)})
        write_kmer_model_arrays(str(tmp_path / "native.model"), "".join(names).encode(), k, mean, sd)
        assert (tmp_path / "py.model").read_bytes() == (tmp_path / "native.model").read_bytes(), k


def test_session_order_spreads_the_demand_for_pages(native_lib):
    """dyn_session_order (no GPU): the order in which a paged session of the resident read queue takes a page-starved ticket's
    reads. A permutation; the shortest eighth last, longest first; and over the rest ANY window of 1 024 consecutive reads --
    the reads the resident waves hold at one time -- asks for about the average number of lattice pages of a config-3-like
    batch (10 k - 100 k rows), where the 1 024 longest would ask for 1.6x that."""
    import ctypes as C
    n = 4096
    rng = np.random.default_rng(3)
    rows = np.sort(rng.integers(10_000, 100_000, size=n))[::-1]          # longest first: rank -> rows
    order = np.zeros(n, dtype=np.uint32)
    assert native_lib.dyn_session_order(n, order.ctypes.data_as(C.POINTER(C.c_uint32))) == 0
    assert sorted(order.tolist()) == list(range(n))
    m = n - n // 8
    assert np.array_equal(order[m:], np.arange(m, n, dtype=np.uint32))    # the shortest eighth: in rank order
    assert set(order[:m].tolist()) == set(range(m))
    demand = rows[order[:m]].astype(np.float64)
    mean = demand.mean()
    win = np.convolve(demand, np.ones(1024), mode="valid") / 1024.0
    assert win.max() < 1.03 * mean and win.min() > 0.97 * mean, (win.min() / mean, win.max() / mean)
    assert rows[:1024].mean() > 1.4 * mean                                # what longest-first would ask for at once
    # degenerate sizes
    for k in (0, 1, 2, 3, 9):
        o = np.zeros(max(1, k), dtype=np.uint32)
        assert native_lib.dyn_session_order(k, o.ctypes.data_as(C.POINTER(C.c_uint32))) == 0
        assert sorted(o[:k].tolist()) == list(range(k))
