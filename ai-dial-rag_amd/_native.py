"""ctypes binding of libmiretr.so (C ABI: include/miretr.h).

Fails loudly: a missing library is an ImportError at import of this module, a
missing GPU is a RuntimeError at the first compute call.  Nothing here falls
back to numpy.
"""

import ctypes as C
import os

import numpy as np

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "libmiretr.so")

MIR_OK, MIR_ERR_INVALID, MIR_ERR_HIP, MIR_ERR_NO_DEVICE, MIR_ERR_EMPTY, MIR_ERR_UNSUPPORTED = range(6)
METRIC_CODES = {"cosine_sim": 0, "euclidean_dist": 1, "sqeuclidean_dist": 2, "inner_product": 3}
DTYPE_F32, DTYPE_F16 = 0, 1
FLAG_UNCERTAIN = 1  # never returned since ABI 2
FLAG_EXACT_PASS = 2  # the query was answered by the exact pass (exact_topk_kernel)
ABI_VERSION = 4


class NativeLibraryMissing(ImportError):
    pass


def _share_torch_hip_runtime():
    """One process can hold only ONE HIP/HSA runtime (a second copy cannot open
    /dev/kfd again and reports "no GPUs").  PyTorch-ROCm wheels bundle their own
    libamdhip64.so.7; libmiretr.so needs the same soname.  Map torch's copy first
    (without importing torch) so that both resolve to it, whichever is imported
    first.  Without torch installed the system runtime under /opt/rocm is used."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def _load():
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C {os.path.join(_PKG_DIR, 'csrc')}`. There is no CPU fallback."
        )
    _share_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    sig = {
        "mir_abi_version": ([], i32),
        "mir_last_error": ([], C.c_char_p),
        "mir_device_count": ([vp], i32),
        "mir_index_create": ([vp, i64, i32, i32, vp, vp, i32, i64, vp], i32),
        "mir_index_create_from_device": ([vp, i64, i32, i32, vp, vp, i32, i64, vp, vp], i32),
        "mir_rows_create": ([vp, i64, i32, i32, vp, i32, vp], i32),
        "mir_rows_info": ([vp, vp, vp, vp, vp, vp], i32),
        "mir_rows_destroy": ([vp], i32),
        "mir_index_create_from_rows": ([vp, vp, i32, i32, i64, vp], i32),
        "mir_index_destroy": ([vp], i32),
        "mir_index_info": ([vp, vp, vp, vp, vp, vp], i32),
        "mir_index_search": ([vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp], i32),
        "mir_index_search_device": ([vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp], i32),
        "mir_index_profile": ([vp, i32], i32),
        "mir_index_profile_read": ([vp, i32, vp, vp], i32),
        "mir_index_scan_stats": ([vp, i32, vp], i32),
        "mir_index_metric_eval": ([vp, vp, i32, vp], i32),
        "mir_metric_eval": ([vp, i64, i32, i32, vp, i32, i32, vp], i32),
        "mir_topk_merge_device": ([vp, vp, vp, i32, i64, i32, i32, i32, vp, vp, vp, i32, vp], i32),
        "mir_topk_merge_host": ([vp, vp, vp, i32, i64, i32, i32, i32, vp, vp, vp], i32),
        "mir_bm25_create": ([vp, vp, i64, i32, C.c_double, C.c_double, C.c_double, vp, C.c_double, i32, i64, vp], i32),
        "mir_compact_term_ids": ([vp, i64, i32, vp, vp, vp], i32),
        "mir_stem_english": ([vp, i64, C.c_char, vp, vp], i32),
        "mir_keywords_preprocess": ([vp, vp, i32, i32, i32, vp], i32),
        "mir_kwp_result_data": ([vp, vp, vp, vp, vp, vp], i32),
        "mir_kwp_result_dedupe": ([vp, i32], i32),
        "mir_kwp_result_unique": ([vp, vp, vp, vp, vp], i32),
        "mir_kwp_result_free": ([vp], i32),
        "mir_bm25_destroy": ([vp], i32),
        "mir_bm25_tune": ([vp, i32], i32),
        "mir_bm25_corpus_stats": ([vp, vp, vp, vp, vp], i32),
        "mir_bm25_idf_from_stats": ([vp, vp, i32, i64, C.c_double, vp, vp], i32),
        "mir_bm25_set_global_stats": ([vp, vp, C.c_double, C.c_double], i32),
        "mir_bm25_info": ([vp, vp, vp, vp, vp, vp, vp], i32),
        "mir_bm25_idf": ([vp, vp], i32),
        "mir_bm25_scores": ([vp, vp, i32, vp], i32),
        "mir_bm25_search": ([vp, vp, vp, i32, i32, vp, vp, vp], i32),
        "mir_bm25_workspace_bytes": ([vp, i32, i32], i64),
        "mir_bm25_search_device": ([vp, vp, vp, i32, i32, vp, vp, vp, vp, vp], i32),
        "mir_encoder_create": ([i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp], i32),
        "mir_encoder_destroy": ([vp], i32),
        "mir_encoder_info": ([vp, vp, vp, vp], i32),
        "mir_encoder_encode": ([vp, vp, vp, i32, i32, vp], i32),
        "mir_encoder_encode_to_device": ([vp, vp, vp, i32, i32, vp, vp], i32),
        "mir_encoder_debug_hidden": ([vp, vp, vp, i32, i32, vp, vp, i64], i32),
        "mir_wordpiece_create": ([vp, i64, vp, vp, vp, vp, i32, vp], i32),
        "mir_wordpiece_destroy": ([vp], i32),
        "mir_wordpiece_encode": ([vp, vp, vp, i32, i32, i32, vp, vp, vp], i32),
        "mir_rrf_fuse": ([vp, vp, vp, i32, i32, vp, vp, vp], i32),
        "mir_rrf_fuse_batch": ([vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp], i32),
    }
    for name, (args, res) in sig.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.argtypes = args
        fn.restype = res
    if lib.mir_abi_version() != ABI_VERSION:
        raise ImportError(f"libmiretr ABI {lib.mir_abi_version()} != binding ABI {ABI_VERSION}; rebuild")
    return lib, sorted(sig)


lib, DECLARED_SYMBOLS = _load()


def last_error() -> str:
    msg = lib.mir_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int) -> None:
    """Map a status to the exception type the reference raises in that case."""
    if rc == MIR_OK:
        return
    msg = last_error()
    if rc == MIR_ERR_INVALID:
        raise ValueError(msg)
    if rc == MIR_ERR_EMPTY:
        raise ValueError("Text index is empty.")  # bm25_retriever.py:75-76
    if rc == MIR_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == MIR_ERR_NO_DEVICE:
        raise RuntimeError(f"libmiretr: {msg}")
    raise RuntimeError(f"libmiretr: HIP failure: {msg}")


def ptr(a):
    """Host pointer of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def device_count() -> int:
    n = C.c_int32(0)
    check(lib.mir_device_count(C.byref(n)))
    return int(n.value)


def as_f64_queries(q, d: int) -> np.ndarray:
    q = np.ascontiguousarray(q, dtype=np.float64)
    if q.ndim == 1:
        q = q[None, :]
    if q.ndim != 2 or q.shape[1] != d:
        raise ValueError(f"query shape {q.shape} does not match index dimension {d}")
    return q
