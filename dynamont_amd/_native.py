"""ctypes binding of libdynamont_mi.so (include/dynamont_mi.h) and its in-tree build recipe.

The shared library is the product boundary; this module only declares signatures. There is
deliberately no fallback: if the library is missing or a GPU call fails, the error propagates.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# DYN_LIB_PATH: another build of the SAME sources (tools/sanitize: the host side under ASan / UBSan / TSan on the CPU)
LIB_PATH = os.environ.get("DYN_LIB_PATH") or os.path.join(HERE, "libdynamont_mi.so")
SOURCES = ["dynamont_mi.cpp", "buffers.cpp", "launch.cpp", "session.cpp", "async_engine.cpp", "pore_model.cpp", "csv_format.cpp", "csv_sink.cpp", "vbz_decode.cpp", "bam_reader.cpp", "rccl_comm.cpp", "model_format.cpp",
           "nt_kernels.hip", "pool_stats.hip", "wide_band.hip"]
HEADERS = ["engine.hpp", "engine_internal.hpp", "zstd_dl.hpp", "vbz_decode.hpp", "nt_kernels.hpp", "pore_model.hpp", "dp_math.hpp", "dp_math_strict.hpp", "strict_exp_table.inc", os.path.join("..", "..", "include", "dynamont_mi.h")]

DYN_DEVICE_HOST_ONLY = -2
DYN_OK, DYN_ERR_INVALID_ARGUMENT, DYN_ERR_RUNTIME, DYN_ERR_DEVICE, DYN_ERR_OUT_OF_MEMORY = range(5)

c_double_p = C.POINTER(C.c_double)
c_float_p = C.POINTER(C.c_float)
c_u64_p = C.POINTER(C.c_uint64)
c_i32_p = C.POINTER(C.c_int32)
c_u8_p = C.POINTER(C.c_uint8)


class DynInfo(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("pore", C.c_int32), ("rna", C.c_int32), ("kmer_size", C.c_int32),
                ("alphabet_size", C.c_int32), ("device", C.c_int32), ("num_kmers", C.c_uint64),
                ("half_band", C.c_uint64), ("log_m1", C.c_double), ("log_e1", C.c_double), ("log_e2", C.c_double),
                ("max_half_band", C.c_uint64)]


class DynAlignOut(C.Structure):
    _fields_ = [("Z", c_double_p), ("status", c_i32_p), ("bad_char", C.c_void_p), ("seg_offsets", c_u64_p),
                ("n_segments", c_u64_p), ("sequence_positions", c_u64_p), ("signal_positions", c_u64_p),
                ("probabilities", c_double_p), ("states", c_u8_p), ("capacity", C.c_uint64)]


class DynTrainOut(C.Structure):
    _fields_ = [("Z", c_double_p), ("status", c_i32_p), ("bad_char", C.c_void_p), ("transitions", c_double_p),
                ("em_offsets", c_u64_p), ("em_count", c_u64_p), ("em_code", c_i32_p), ("em_mean", c_double_p),
                ("em_stdev", c_double_p), ("em_weight", c_double_p), ("em_sum", c_double_p),
                ("em_sumsq", c_double_p), ("trans_counts", c_double_p), ("capacity", C.c_uint64)]


class DynJobBatch(C.Structure):
    """dyn_job_batch: the columns dyn_bam_next returns (owned by the reader, valid until its next call)"""
    _fields_ = [("n", C.c_uint64),
                ("names", C.c_void_p), ("name_off", C.POINTER(C.c_uint64)), ("names_bytes", C.c_uint64),
                ("signal_ids", C.c_void_p), ("signal_id_off", C.POINTER(C.c_uint64)), ("signal_ids_bytes", C.c_uint64),
                ("signal_uuid", C.c_void_p), ("signal_uuid_ok", C.c_void_p),
                ("seqs", C.c_void_p), ("seq_off", C.POINTER(C.c_uint64)), ("seqs_bytes", C.c_uint64),
                ("shift", C.POINTER(C.c_double)), ("scale", C.POINTER(C.c_double)),
                ("start", C.POINTER(C.c_int64)), ("end", C.POINTER(C.c_int64)),
                ("n_files", C.c_uint64), ("files", C.c_void_p), ("file_off", C.POINTER(C.c_uint64)), ("files_bytes", C.c_uint64),
                ("file_id", C.POINTER(C.c_uint32)), ("bases", C.POINTER(C.c_uint32))]


class DynTiming(C.Structure):
    _fields_ = [("ms_total", C.c_double), ("ms_dp", C.c_double), ("ms_backward", C.c_double), ("ms_forward", C.c_double),
                ("ms_trace", C.c_double), ("wave_wait_share", C.c_double), ("wave_occupancy", C.c_double),
                ("cells", C.c_uint64), ("samples", C.c_uint64), ("reads_ok", C.c_uint64),
                ("launches", C.c_uint32), ("lp_inplace", C.c_uint32), ("pool_pages", C.c_uint32),
                ("page_rows", C.c_uint32), ("n_static", C.c_uint32), ("n_waves", C.c_uint32),
                ("reads_strict", C.c_uint32), ("reserved", C.c_uint32),
                ("ms_backward_strict", C.c_double), ("ms_forward_strict", C.c_double), ("cert_fallbacks", C.c_uint64),
                ("cert_rows", C.c_uint64), ("launch_share", C.c_double)]


class DynSessionStats(C.Structure):
    _fields_ = [("sessions", C.c_uint64), ("tickets", C.c_uint64), ("reads", C.c_uint64), ("cells", C.c_uint64), ("ms", C.c_double),
                ("wave_cycles_busy", C.c_uint64), ("wave_cycles_idle", C.c_uint64), ("wave_cycles_life", C.c_uint64),
                ("waves", C.c_uint64), ("aborted", C.c_uint64), ("republished", C.c_uint64)]


# every symbol include/dynamont_mi.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "dyn_pore_from_string": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.c_char_p, C.c_uint64]),
    "dyn_aligner_create": (C.c_int, [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_uint64, C.c_int,
                                     C.POINTER(C.c_void_p), C.c_char_p, C.c_uint64]),
    "dyn_aligner_destroy": (None, [C.c_void_p]),
    "dyn_release_cached_memory": (None, []),
    "dyn_aligner_set_model": (C.c_int, [C.c_void_p, c_double_p]),
    "dyn_format_model": (C.c_uint64, [C.c_char_p, C.c_int, c_double_p, c_double_p, C.c_uint64, C.c_char_p, C.c_uint64]),
    "dyn_aligner_info": (C.c_int, [C.c_void_p, C.POINTER(DynInfo)]),
    "dyn_aligner_model": (C.c_int, [C.c_void_p, c_double_p]),
    "dyn_aligner_session_stats": (C.c_int, [C.c_void_p, C.POINTER(DynSessionStats)]),
    "dyn_aligner_session_page_wait": (C.c_int, [C.c_void_p, c_u64_p]),
    "dyn_aligner_session_idle_split": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "dyn_aligner_set_session_mode": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "dyn_aligner_set_mem_budget": (C.c_int, [C.c_void_p, C.c_uint64]),
    "dyn_aligner_set_strict": (C.c_int, [C.c_void_p, C.c_int]),
    "dyn_aligner_set_train_zcheck": (C.c_int, [C.c_void_p, C.c_int]),
    "dyn_tie_rows": (C.c_uint32, [C.c_void_p, c_i32_p, C.c_uint64, C.c_uint64]),
    "dyn_aligner_last_error": (C.c_char_p, [C.c_void_p]),
    "dyn_read_strerror": (C.c_int, [C.c_int, C.c_char, C.c_char_p, C.c_uint64]),
    "dyn_segment_capacity": (C.c_uint64, [C.c_void_p, C.c_uint64, c_u64_p]),
    "dyn_validate_batch": (C.c_int, [C.c_void_p, C.c_uint64, c_u64_p, C.c_char_p, c_u64_p, c_i32_p, C.c_void_p,
                                     c_i32_p, C.c_uint64]),
    "dyn_align_batch": (C.c_int, [C.c_void_p, C.c_uint64, c_double_p, c_u64_p, C.c_char_p, c_u64_p, C.c_int,
                                  C.POINTER(DynAlignOut)]),
    "dyn_train_batch": (C.c_int, [C.c_void_p, C.c_uint64, c_double_p, c_u64_p, C.c_char_p, c_u64_p,
                                  C.POINTER(DynTrainOut), c_double_p]),
    "dyn_format_csv_bound": (C.c_uint64, [C.c_void_p, C.c_uint64, C.POINTER(DynAlignOut), C.POINTER(C.c_char_p),
                                          C.POINTER(C.c_char_p)]),
    "dyn_format_csv": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(DynAlignOut), C.c_char_p, c_u64_p,
                                 C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_int64),
                                 C.POINTER(C.c_int64), C.c_int, C.c_void_p, C.c_uint64, c_u64_p, c_u64_p]),
    "dyn_csv_compact": (C.c_uint64, [C.c_void_p, C.c_uint64, c_u64_p, c_u64_p]),
    "dyn_csv_sink_open": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_char_p, C.c_uint64]),
    "dyn_csv_sink_open_part": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_char_p, C.c_uint64]),
    "dyn_csv_sink_submit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(DynAlignOut), C.c_uint64, C.c_char_p, c_u64_p,
                                      C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_int64), c_u64_p]),
    "dyn_csv_sink_submit_bases": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(DynAlignOut), C.c_uint64, C.c_char_p, c_u64_p,
                                            C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_int64), c_u64_p, C.POINTER(C.c_uint32)]),
    "dyn_csv_sink_error_line": (C.c_int, [C.c_void_p, C.c_char_p]),
    "dyn_csv_sink_completed": (C.c_uint64, [C.c_void_p]),
    "dyn_csv_sink_wait": (C.c_uint64, [C.c_void_p, C.c_uint64, C.c_int]),
    "dyn_csv_sink_failed": (C.c_int, [C.c_void_p]),
    "dyn_csv_sink_close": (C.c_int, [C.c_void_p, c_u64_p, c_u64_p, c_u64_p, C.c_char_p, C.c_uint64]),
    "dyn_bam_open": (C.c_int, [C.c_char_p, C.c_int, C.c_char_p, C.POINTER(C.c_void_p), C.c_char_p, C.c_uint64]),
    "dyn_bam_next": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_double, C.c_uint32, C.c_uint32, C.POINTER(DynJobBatch),
                               C.c_char_p, C.c_uint64]),
    "dyn_bam_skipped": (C.c_uint64, [C.c_void_p]),
    "dyn_bam_close": (None, [C.c_void_p]),
    "dyn_batch_create": (C.c_int, [C.c_void_p, C.c_uint64, c_double_p, c_u64_p, C.c_char_p, c_u64_p,
                                   C.POINTER(C.c_void_p)]),
    "dyn_batch_create_raw": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, c_u64_p, c_double_p, c_double_p,
                                       C.c_int, C.c_double, C.c_int, C.c_char_p, c_u64_p, C.POINTER(C.c_void_p)]),
    "dyn_batch_signals": (C.c_int, [C.c_void_p, c_double_p, C.c_uint64]),
    "dyn_batch_destroy": (None, [C.c_void_p]),
    "dyn_batch_align": (C.c_int, [C.c_void_p, C.c_int]),
    "dyn_batch_train": (C.c_int, [C.c_void_p]),
    "dyn_batch_fetch": (C.c_int, [C.c_void_p, C.POINTER(DynAlignOut)]),
    "dyn_batch_fetch_train": (C.c_int, [C.c_void_p, C.POINTER(DynTrainOut), c_double_p]),
    "dyn_batch_device_results": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), c_u64_p, C.POINTER(C.c_void_p)]),
    "dyn_batch_device_pooled": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), c_u64_p]),
    "dyn_batch_timing": (C.c_int, [C.c_void_p, C.POINTER(DynTiming)]),
    "dyn_plan_queue": (C.c_int, [C.c_uint64, C.POINTER(C.c_uint32), c_u64_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32),
                                 c_u64_p, c_u64_p]),
    "dyn_session_order": (C.c_int, [C.c_uint64, C.POINTER(C.c_uint32)]),
    "dyn_batch_align_async": (C.c_int, [C.c_void_p, C.c_uint64, c_double_p, c_u64_p, C.c_char_p, c_u64_p, C.c_int,
                                        C.POINTER(DynAlignOut), C.POINTER(C.c_void_p)]),
    "dyn_batch_train_async": (C.c_int, [C.c_void_p, C.c_uint64, c_double_p, c_u64_p, C.c_char_p, c_u64_p,
                                        C.POINTER(DynTrainOut), c_double_p, C.POINTER(C.c_void_p)]),
    "dyn_batch_align_raw_async": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, c_u64_p, c_float_p, c_float_p, c_double_p, c_double_p,
                                            C.c_int, C.c_double, C.c_int, C.c_char_p, c_u64_p, C.c_int,
                                            C.POINTER(DynAlignOut), C.POINTER(C.c_void_p)]),
    "dyn_batch_train_raw_async": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, c_u64_p, c_float_p, c_float_p, c_double_p, c_double_p,
                                            C.c_int, C.c_double, C.c_int, C.c_char_p, c_u64_p,
                                            C.POINTER(DynTrainOut), c_double_p, C.POINTER(C.c_void_p)]),
    "dyn_batch_align_vbz_async": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, c_u64_p, C.POINTER(C.c_uint32), c_u64_p, c_u64_p, c_u64_p,
                                            c_float_p, c_float_p, c_double_p, c_double_p, C.c_int, C.c_double, C.c_int, C.c_char_p,
                                            c_u64_p, C.c_int, C.POINTER(DynAlignOut), C.POINTER(C.c_void_p)]),
    "dyn_vbz_decode": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_char_p, C.c_uint64]),
    "dyn_batch_wait": (C.c_int, [C.c_void_p]),
    "dyn_host_alloc": (C.c_void_p, [C.c_uint64]),
    "dyn_host_free": (None, [C.c_void_p]),
    "dyn_comm_unique_id": (C.c_int, [c_u8_p, C.c_char_p, C.c_uint64]),
    "dyn_comm_create": (C.c_int, [c_u8_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_char_p, C.c_uint64]),
    "dyn_comm_destroy": (None, [C.c_void_p]),
    "dyn_comm_last_error": (C.c_char_p, [C.c_void_p]),
    "dyn_comm_gather_counts": (C.c_int, [C.c_void_p, C.c_void_p, c_u64_p]),
    "dyn_comm_gather_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, c_u64_p]),
    "dyn_comm_allreduce_pooled": (C.c_int, [C.c_void_p, C.c_void_p, c_double_p]),
    "dyn_comm_gather_bytes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, c_u64_p]),
    "dyn_comm_gathered_bytes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "dyn_comm_allreduce_f64": (C.c_int, [C.c_void_p, c_double_p, C.c_uint64, C.c_int]),
    "dyn_multi_create": (C.c_int, [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_uint64, C.POINTER(C.c_int), C.c_int,
                                   C.POINTER(C.c_void_p), C.c_char_p, C.c_uint64]),
    "dyn_multi_destroy": (None, [C.c_void_p]),
    "dyn_multi_device_count": (C.c_int, [C.c_void_p]),
    "dyn_multi_handle": (C.c_void_p, [C.c_void_p, C.c_int]),
    "dyn_multi_last_error": (C.c_char_p, [C.c_void_p]),
    "dyn_multi_align_batch": (C.c_int, [C.c_void_p, C.c_uint64, c_double_p, c_u64_p, C.c_char_p, c_u64_p, C.c_int,
                                        C.POINTER(DynAlignOut)]),
    "dyn_multi_train_batch": (C.c_int, [C.c_void_p, C.c_uint64, c_double_p, c_u64_p, C.c_char_p, c_u64_p,
                                        C.POINTER(DynTrainOut), c_double_p]),
}


def hipcc_path() -> str:
    for p in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if p and (os.path.sep not in p or os.path.exists(p)):
            return p
    return "hipcc"


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> dynamont_amd/libdynamont_mi.so (in-tree, so it travels). One object per source under
    build/obj/<hash of the flags>/ (compiled in parallel, recompiled only when the source or any header is newer), then
    one link: the same flags and the same device code as a single hipcc command over all sources."""
    if os.environ.get("DYN_LIB_PATH") or (not force and not needs_build()):
        return LIB_PATH  # (a library named by DYN_LIB_PATH is built by whoever named it)
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result", "-pthread"]
    flags += os.environ.get("DYN_HIPCC_EXTRA", "").split()  # kernel experiments: -DDYN_EXP_...
    objdir = os.path.join(os.path.dirname(HERE), "build", "obj", hashlib.sha1(" ".join(flags).encode()).hexdigest()[:12])
    os.makedirs(objdir, exist_ok=True)
    newest_header = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS if os.path.exists(os.path.join(CSRC, h)))

    def compile_one(src):
        obj = os.path.join(objdir, src + ".o")
        path = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), newest_header):
            return obj
        cmd = [hipcc_path()] + flags + ["-x", "hip", "-c", path, "-o", obj + ".tmp"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        os.replace(obj + ".tmp", obj)
        return obj

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", LIB_PATH + ".tmp"] + objs + ["-ldl", "-lz"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    """Load the C-ABI library. Raises if it is absent -- there is no other implementation."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). dynamont_amd has no CPU or PyTorch fallback.")
        import sys
        if "torch" not in sys.modules and int(os.environ.get("WORLD_SIZE", "1") or 1) > 1:
            # A multi-rank job will need torch.distributed. PyTorch's wheel bundles its own libamdhip64 and refuses to
            # initialise once another copy is mapped, so in such a process torch goes first.
            import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
