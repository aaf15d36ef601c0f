"""ctypes binding of libfspann_hip.so (include/fspann.h).

Product code.  There is no CPU fallback: if the HIP library is missing, import
fails; if no GPU is present, creating a context raises FspannDeviceError.
Nothing here imports oracle/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfspann_hip.so")
_SRC = os.path.join(_HERE, "csrc")

HIPCC_FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared",
               "-Wall", "-Wno-unused-function", "-pthread", "-ldl"]

OK, E_STATE, E_ARG, E_NULL, E_DEVICE, E_NOMEM, E_RANGE = 0, -1, -2, -3, -4, -5, -6
F32, F64 = 0, 1
INT32_MAX = 2**31 - 1


class FspannError(RuntimeError):
    code = None


class FspannStateError(FspannError):        # java.lang.IllegalStateException
    code = E_STATE


class FspannArgumentError(FspannError, ValueError):  # java.lang.IllegalArgumentException
    code = E_ARG


class FspannNullError(FspannError, TypeError):       # java.lang.NullPointerException
    code = E_NULL


class FspannDeviceError(FspannError):
    code = E_DEVICE


class FspannMemoryError(FspannError, MemoryError):
    code = E_NOMEM


class FspannRangeError(FspannError):
    code = E_RANGE


_ERR = {E_STATE: FspannStateError, E_ARG: FspannArgumentError, E_NULL: FspannNullError,
        E_DEVICE: FspannDeviceError, E_NOMEM: FspannMemoryError, E_RANGE: FspannRangeError}


class Cfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "tables", "divisions", "m", "lambda_", "dim", "block_size", "default_probes", "probe_override",
        "max_global_candidates", "refinement_limit", "hamming_prefilter_threshold", "reserved")]


class Tick(C.Structure):
    """fspann_tick of include/fspann.h (device pointers as integers; 0 / None = absent)."""
    _fields_ = [
        ("nq_encode", C.c_int64), ("enc_q_dev", C.c_void_p), ("enc_dtype", C.c_int32), ("pad0", C.c_int32),
        ("enc_codes_dev", C.c_void_p), ("enc_bad_dev", C.c_void_p),
        ("nq_route", C.c_int64), ("route_codes_dev", C.c_void_p), ("route_probe_override", C.c_int32), ("route_limit", C.c_int32),
        ("route_ids_dev", C.c_void_p), ("route_count_dev", C.c_void_p), ("route_handover_dev", C.c_void_p),
        ("nq_refine", C.c_int64), ("ref_q_dev", C.c_void_p), ("ref_q_dtype", C.c_int32), ("ref_cand_dtype", C.c_int32),
        ("ref_cand_dev", C.c_void_p), ("ref_B", C.c_int64), ("ref_ids_dev", C.c_void_p), ("ref_count_dev", C.c_void_p),
        ("ref_codes_dev", C.c_void_p), ("ref_handover_dev", C.c_void_p), ("ref_probe_override", C.c_int32), ("k", C.c_int32),
        ("out_ids_dev", C.c_void_p), ("out_dist_dev", C.c_void_p), ("out_count_dev", C.c_void_p), ("scored_dev", C.c_void_p)]


def sources():
    return [os.path.join(_SRC, f) for f in sorted(os.listdir(_SRC))]


def needs_build() -> bool:
    if not os.path.exists(_SO):
        return True
    t = os.path.getmtime(_SO)
    inc = os.path.join(_HERE, "..", "include", "fspann.h")
    return any(os.path.getmtime(p) > t for p in sources() + [inc])


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc cross-compiles for gfx950 without a GPU present.  FSPANN_BUILD_DEBUG=1: with the per-phase clock stamps and
    the fspann_debug_route_stamps export the tools/ scripts use (never part of a release build)."""
    if force or needs_build():
        dbg = ["-DFSPANN_DEBUG_STAMPS"] if os.environ.get("FSPANN_BUILD_DEBUG") == "1" else []
        cmd = ["hipcc"] + HIPCC_FLAGS + dbg + ["-o", _SO, os.path.join(_SRC, "fspann_api.hip")]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return _SO


_LIB = None

_vp, _i, _i32, _i64, _sz = C.c_void_p, C.c_int, C.c_int32, C.c_int64, C.c_size_t
_SIGS = {
    "fspann_ctx_create": (_i, [_i, C.POINTER(Cfg), C.POINTER(_vp)]),
    "fspann_ctx_destroy": (None, [_vp]),
    "fspann_ctx_clone": (_i, [_vp, C.POINTER(_vp)]),
    "fspann_last_error": (C.c_char_p, []),
    "fspann_version": (C.c_char_p, []),
    "fspann_ctx_stream": (_vp, [_vp]),
    "fspann_sync": (_i, [_vp]),
    "fspann_set_gfunctions": (_i, [_vp, _vp, _vp, _vp]),
    "fspann_registry_initialize": (_i, [_vp, _vp, _i64, _i64]),
    "fspann_get_gfunctions": (_i, [_vp, _vp, _vp, _vp]),
    "fspann_host_buffer": (_vp, [_vp, _sz]),
    "fspann_set_index": (_i, [_vp, _i, _i64, _vp, _vp, _vp, _vp, _vp]),
    "fspann_set_id_meta": (_i, [_vp, _i64, _vp, _vp]),
    "fspann_finalize": (_i, [_vp]),
    "fspann_build_index": (_i, [_vp, _i64, _vp, _i, _vp]),
    "fspann_build_begin": (_i, [_vp, _i64]),
    "fspann_build_append": (_i, [_vp, _i64, _vp, _i]),
    "fspann_build_finish": (_i, [_vp, _vp]),
    "fspann_set_deleted": (_i, [_vp, _vp, _i64, _i]),
    "fspann_route_resolve_dev": (_i, [_vp, _i64, _vp, _i, _i32, _i64, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i64)]),
    "fspann_search_store_finish_dev": (_i, [_vp, _i64, _vp, _i, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i64)]),
    "fspann_index_save": (_i, [_vp, C.c_char_p]),
    "fspann_index_load": (_i, [_vp, C.c_char_p]),
    "fspann_index_dims": (_i, [_vp, _i, C.POINTER(_i64), C.POINTER(_i64)]),
    "fspann_get_index": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "fspann_encode": (_i, [_vp, _i64, _vp, _i, _vp, _vp]),
    "fspann_encode_dev": (_i, [_vp, _i64, _vp, _i, _vp, _vp, _vp]),
    "fspann_set_encode_mode": (_i, [_vp, _i]),
    "fspann_last_encode_rechecked": (_i64, [_vp]),
    "fspann_route": (_i, [_vp, _i64, _vp, _i, _i32, _i64, _vp, _vp, _vp, _vp, _vp]),
    "fspann_route_dev": (_i, [_vp, _i64, _vp, _i, _i32, _i64, _vp, _vp, _vp, _vp, _vp]),
    "fspann_route_max_candidates": (_i64, [_vp, _i]),
    "fspann_effective_probes": (_i, [_vp, _i]),
    "fspann_set_route_mode": (_i, [_vp, _i]),
    "fspann_last_route_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "fspann_unmodelled_queries": (_i, [_vp, C.POINTER(_i64), _i]),
    "fspann_refine": (_i, [_vp, _i64, _vp, _vp, _i, _i64, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "fspann_refine_dev": (_i, [_vp, _i64, _vp, _i, _vp, _i, _i64, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "fspann_refine_store": (_i, [_vp, _i64, _vp, _i, _i64, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "fspann_refine_store_dev": (_i, [_vp, _i64, _vp, _i, _i64, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "fspann_refine_timing_begin": (_i, [_vp, _i, _i]),
    "fspann_refine_timing_end": (_i, [_vp, C.POINTER(_i), C.POINTER(C.c_double)]),
    "fspann_search_store_dev": (_i, [_vp, _i64, _vp, _i, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fspann_store_set": (_i, [_vp, _i64, _vp, _i]),
    "fspann_store_attach_dev": (_i, [_vp, _i64, _vp, _i]),
    "fspann_store_gather_dev": (_i, [_vp, _i64, _vp, _vp, _i64, _vp]),
    "fspann_store_dev_ptr": (_vp, [_vp, C.POINTER(_i)]),
    "fspann_route_handover_bytes": (_sz, [_vp, _i64, _i]),
    "fspann_tick_dev": (_i, [_vp, C.POINTER(Tick)]),
    "fspann_last_tick_fused": (_i, [_vp]),
    "fspann_groundtruth_dev": (_i, [_vp, _i64, _vp, _i64, _vp, _i, _i, _vp, _vp]),
    "fspann_eval_metrics_dev": (_i, [_vp, _i64, _vp, _i64, _vp, _i, _i, _vp, _i64, _vp, _vp, _i64, _vp, _vp]),
    "fspann_pointstore_create": (_i, [_i64, _i, C.POINTER(_vp)]),
    "fspann_pointstore_destroy": (None, [_vp]),
    "fspann_pointstore_set_master_key": (_i, [_vp, _vp]),
    "fspann_pointstore_current_version": (_i, [_vp]),
    "fspann_pointstore_rotate": (_i, [_vp, C.POINTER(_i)]),
    "fspann_pointstore_retire": (_i, [_vp, _i]),
    "fspann_pointstore_encrypt": (_i, [_vp, _i64, _i64, _vp, _i, _i]),
    "fspann_pointstore_delete": (_i, [_vp, _i64]),
    "fspann_pointstore_reencrypt": (_i, [_vp, _vp, _i64, _i, C.POINTER(_i64)]),
    "fspann_pointstore_open_batch": (_i, [_vp, _i64, _i64, _vp, _vp, _vp, _i, _vp, _vp, _i]),
    "fspann_pointstore_get_record": (_i, [_vp, _i64, C.POINTER(_i32), _vp, _vp]),
    "fspann_pointstore_put_record": (_i, [_vp, _i64, _i32, _vp, _vp]),
    "fspann_pointstore_stats": (_i, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "fspann_pipeline_create": (_i, [_vp, _vp, _i64, _i64, _i, _i, C.POINTER(_vp)]),
    "fspann_pipeline_submit": (_i, [_vp, _i64, _vp, C.POINTER(C.c_uint64)]),
    "fspann_pipeline_collect": (_i, [_vp, C.POINTER(C.c_uint64), C.POINTER(_i64), _vp, _vp, _vp]),
    "fspann_pipeline_stats": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i64)]),
    "fspann_pipeline_destroy": (None, [_vp]),
    "fspann_topk_bytes": (_sz, [_i64, _i]),
    "fspann_topk_dist_offset": (_sz, [_i64, _i]),
    "fspann_comm_available": (_i, []),
    "fspann_comm_unique_id": (_i, [_vp]),
    "fspann_comm_create": (_i, [_vp, _vp, _i, _i, C.POINTER(_vp)]),
    "fspann_comm_destroy": (_i, [_vp]),
    "fspann_comm_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(C.c_char_p)]),
    "fspann_allgather_topk_dev": (_i, [_vp, _i64, _i, _vp, _vp]),
    "fspann_hbm_read_peak": (_i, [_vp, _sz, _i, C.POINTER(C.c_double)]),
    "fspann_hbm_read_window": (_i, [_vp, _sz, _sz, _i, C.POINTER(C.c_double)]),
    "fspann_dev_alloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "fspann_dev_free": (_i, [_vp, _vp]),
    "fspann_h2d": (_i, [_vp, _vp, _vp, _sz]),
    "fspann_d2h": (_i, [_vp, _vp, _vp, _sz]),
}


def lib() -> C.CDLL:
    """Load libfspann_hip.so; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(_SO):
            raise ImportError(
                f"{_SO} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  fspann has no CPU fallback.")
        L = C.CDLL(_SO)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)  # AttributeError if the ABI and the header drift apart
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc: int):
    if rc == OK:
        return
    msg = lib().fspann_last_error().decode("utf-8", "replace")
    raise _ERR.get(rc, FspannError)(msg)


def exported_symbols():
    return sorted(_SIGS)
