"""The reference-side binding of INTEGRATION.md (integration/aligner_bindings_mi.cpp: a pybind11 module `_dynamont`
over the C ABI) compiles, links against libdynamont_mi.so and exposes the reference's surface
(src/cpp/aligner_bindings.cpp:180-219). CPU part: a host-only handle (DYNAMONT_MI_DEVICE=host) -- surface,
enum, error translation. GPU part (-m gpu): align()/train() through the stub against the goldens."""
import importlib.util
import os
import subprocess
import sys
import sysconfig

import numpy as np
import pytest

from conftest import ROOT, golden


@pytest.fixture(scope="module")
def stub(tmp_path_factory, native_lib):
    pybind11 = pytest.importorskip("pybind11")
    d = tmp_path_factory.mktemp("stub")
    so = d / ("_dynamont" + sysconfig.get_config_var("EXT_SUFFIX"))
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"],
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "integration", "aligner_bindings_mi.cpp"),
           "-L" + os.path.join(ROOT, "dynamont_amd"), "-ldynamont_mi", "-Wl,-rpath," + os.path.join(ROOT, "dynamont_amd"), "-o", str(so)]
    subprocess.run(cmd, check=True)
    spec = importlib.util.spec_from_file_location("_dynamont", str(so))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_stub_surface_and_error_translation(stub, models, monkeypatch):
    monkeypatch.setenv("DYNAMONT_MI_DEVICE", "host")
    assert [p for p in stub.PoreType.__members__] == ["RNA002", "RNA004", "DNA_R9", "DNA_R10_260", "DNA_R10_400"]
    assert int(stub.PoreType.DNA_R10_400) == 4 and stub.pore_type("rna004") == stub.PoreType.RNA004
    with pytest.raises(ValueError, match="Unknown pore type: foo"):
        stub.pore_type("foo")
    al = stub.Aligner(models["syn5"], "rna002")                       # pore as str, defaults mode="basic", threads=1, band=400
    al2 = stub.Aligner(model_file=models["syn9"], pore=stub.PoreType.RNA004, mode="nt", threads=4, band=400)
    for name in ("align", "train"):
        assert callable(getattr(al, name)) and callable(getattr(al2, name))
    with pytest.raises(ValueError, match="Unknown pore type: bar"):
        stub.Aligner(models["syn5"], "bar")
    with pytest.raises(ValueError, match="Unknown aligner mode: fast"):
        stub.Aligner(models["syn5"], "rna002", mode="fast")
    with pytest.raises(RuntimeError, match="Could not open model file"):
        stub.Aligner("/nonexistent.model", "rna002")
    with pytest.raises(RuntimeError, match="Inconsistent kmer size in model"):
        stub.Aligner(models["syn9"], "rna002")
    with pytest.raises(ValueError, match="Signal must be a one-dimensional array"):
        al.align(np.zeros((3, 3)), "ACGTACGT", True)
    with pytest.raises(ValueError, match="Signal must be a one-dimensional array"):
        al.train(np.zeros((3, 3)), "ACGTACGT")
    # no CPU compute path: a host-only handle refuses to compute, loudly
    with pytest.raises(RuntimeError, match="no CPU compute path"):
        al.align(np.zeros(100), "ACGTACGTACGT", False)
    with pytest.raises(RuntimeError, match="no CPU compute path"):
        al.train(np.zeros(100), "ACGTACGTACGT")


@pytest.mark.gpu
def test_stub_align_and_train_on_the_gpu(stub, models, monkeypatch):
    from conftest import assert_matches_golden
    monkeypatch.setenv("DYNAMONT_MI_DEVICE", "0")
    g = golden("g1_cfg1.npz")
    al = stub.Aligner(models["syn5"], "rna002")
    res = al.align(g["syn_signal"].astype(np.float32).astype(np.float64), str(g["syn_sequence"]), True)  # forcecast path below
    res = al.align(g["syn_signal"], str(g["syn_sequence"]), calc_probabilities=True)
    assert_matches_golden(res, g, "syn_", 1e-6)
    assert res["sequence_positions"].dtype == np.uint64 and res["polishes"] == [""] * len(res["states"])
    z = al.align(list(g["syn_signal"]), str(g["syn_sequence"]))             # any array-like, calc_probabilities defaults to False
    assert len(z["probabilities"]) == 0 and abs(z["Z"] - float(g["syn_Z"])) <= 1e-9 * abs(float(g["syn_Z"]))
    with pytest.raises(RuntimeError, match="Invalid nucleotide: N"):
        al.align(g["syn_signal"], "N" + str(g["syn_sequence"])[1:], True)
    g7 = golden("g7_train.npz")
    for i in range(int(g7["n_cases"])):
        p = f"t{i}_"
        if str(g7[p + "pore"]) != "rna002":
            continue
        tr = al.train(g7[p + "signal"], str(g7[p + "sequence"]))
        assert abs(tr["Z"] - float(g7[p + "Z"])) <= 1e-9 * abs(float(g7[p + "Z"]))
        t = tr["transition_params"]
        assert np.abs(np.array([t["m1"], t["e1"], t["e2"]]) - g7[p + "trans"]).max() <= 1e-9
        assert len(tr["emission_model"]) == 4 ** 5
        codes = g7[p + "codes"]
        got = np.array([tr["emission_model"][int(c)]["mean"] for c in codes])
        assert np.abs(got - g7[p + "mean"]).max() <= 1e-9
    batch = al.align_batch([g["syn_signal"], np.zeros(0), g["syn_signal"]], [str(g["syn_sequence"]), "ACGTACGT", str(g["syn_sequence"])], True)
    assert batch[1] == {"error": "Signal is empty"} and np.array_equal(batch[0]["signal_positions"], batch[2]["signal_positions"])


@pytest.fixture(scope="module")
def c_example(tmp_path_factory, native_lib):
    """integration/multi_gpu_example.c: a plain C consumer of the ABI (header is C, not only C++)."""
    exe = tmp_path_factory.mktemp("cex") / "multi_gpu_example"
    subprocess.run(["gcc", "-std=c11", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "integration", "multi_gpu_example.c"), "-L" + os.path.join(ROOT, "dynamont_amd"),
                    "-ldynamont_mi", "-Wl,-rpath," + os.path.join(ROOT, "dynamont_amd"), "-lm", "-o", str(exe)], check=True)
    return str(exe)


def test_c_example_builds_and_fails_loudly_without_a_gpu(c_example, models):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([c_example, models["syn9"], "rna004", "0"], capture_output=True, text=True)
    assert r.returncode == 1 and "no CPU compute path" in r.stderr


@pytest.mark.gpu
def test_c_example_multi_device_on_the_gpu(c_example, models):
    r = subprocess.run([c_example, models["syn9"], "rna004", "0", "0", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "3 device handle(s): identical to the single-device result" in r.stdout
