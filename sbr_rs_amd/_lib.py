"""Loader of libsbr_hip.so (the gfx950 engine).  There is no CPU fallback: if the library is
missing or no HIP device is visible, the engine raises."""
from __future__ import annotations

import ctypes as C
import os

from ._abi import ABI_VERSION, SbrHparams

_HERE = os.path.dirname(os.path.abspath(__file__))
# SBR_HIP_LIB: another build of the same engine (kernel A/B experiments: tools/, profiles/); never a different implementation
LIB_PATH = os.environ.get("SBR_HIP_LIB") or os.path.join(_HERE, "libsbr_hip.so")
_lib = None


class EngineUnavailable(RuntimeError):
    """libsbr_hip.so cannot be loaded (not built) — the product path never falls back to CPU."""


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineUnavailable(
            f"{LIB_PATH} is missing: build it with `python -m sbr_rs_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, u64p, u32p, fp = C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_float)
    sig = {
        "sbr_model_create": [C.POINTER(SbrHparams), C.POINTER(vp)],
        "sbr_model_fit": [vp, vp, vp, C.c_uint64, fp],
        "sbr_fit_begin": [vp, vp, vp, C.c_uint64, C.POINTER(vp)],
        "sbr_fit_epoch_prepare": [vp, u64p],
        "sbr_fit_epoch_prefetch": [vp],
        "sbr_fit_step": [vp, C.c_uint64],
        "sbr_fit_steps": [vp, C.c_uint64, C.c_uint64],
        "sbr_model_set_step_fusion": [vp, C.c_int32],
        "sbr_model_set_reference_order": [vp, C.c_int32],
        "sbr_comm_unique_id": [vp],
        "sbr_comm_create": [vp, C.c_uint32, C.c_uint32, C.POINTER(vp)],
        "sbr_fit_step_exchange": [vp, C.c_uint64, vp],
        "sbr_model_fit_comm": [vp, vp, vp, vp, C.c_uint64, fp],
        "sbr_fit_block_bytes": [vp, u64p],
        "sbr_fit_step_apply_blocks_in_order": [vp, C.c_uint64, vp],
        "sbr_fit_debug_phase_clocks": [vp, u64p],
        "sbr_fit_minibatch_rows": [vp, C.c_uint64, u64p],
        "sbr_fit_end": [vp, fp, u64p],
        "sbr_fit_counters": [vp, u64p, u64p],
        "sbr_fit_end_lagged": [vp, fp],
        "sbr_model_last_fit_lagged_loss": [vp, fp],
        "sbr_model_get_rng": [vp, vp],
        "sbr_model_set_rng": [vp, vp],
        "sbr_fit_sparse_stats": [vp, u64p, u64p],
        "sbr_fit_step_local": [vp, C.c_uint64],
        "sbr_fit_step_apply": [vp, C.c_uint64],
        "sbr_fit_chunk_bytes": [vp, u64p],
        "sbr_fit_dense_bytes": [vp, u64p],
        "sbr_fit_step_scatter": [vp, C.c_uint64, vp],
        "sbr_fit_step_dense": [vp, vp],
        "sbr_fit_step_owner_reduce": [vp, vp, vp],
        "sbr_fit_step_owner_reduce_on": [vp, vp, vp, vp],
        "sbr_fit_step_apply_table": [vp, vp, vp],
        "sbr_fit_step_apply_rows": [vp, vp],
        "sbr_fit_step_apply_dense": [vp, vp],
        "sbr_fit_step_owner_update": [vp, vp],
        "sbr_model_table_slice": [vp, C.c_int32, C.POINTER(vp), u64p],
        "sbr_model_optimizer_state_gathered": [vp],
        "sbr_model_optimizer_state_is_partial": [vp, C.POINTER(C.c_int32)],
        "sbr_group_plan_set_exchange": [vp, C.c_int32],
        "sbr_group_gather_optimizer_state": [vp],
        "sbr_comm_gather_optimizer_state": [vp, vp],
        "sbr_model_set_stream": [vp, vp],
        "sbr_model_synchronize": [vp],
        "sbr_fit_debug_fetch": [vp, C.c_int32, vp, C.c_uint64],
        "sbr_user_representation": [vp, vp, C.c_uint64, vp],
        "sbr_predict": [vp, vp, vp, C.c_uint64, vp],
        "sbr_mrr_score": [vp, vp, vp, C.c_uint64, fp, vp, u64p],
        "sbr_model_param_count": [vp, C.c_int32, u64p],
        "sbr_model_get_param": [vp, C.c_int32, vp, C.c_uint64],
        "sbr_model_set_param": [vp, C.c_int32, vp, C.c_uint64],
        "sbr_model_get_param_rows": [vp, C.c_int32, vp, C.c_uint64, vp],
        "sbr_model_get_epoch": [vp, u64p],
        "sbr_model_get_counters": [vp, u64p, u64p],
        "sbr_model_set_counters": [vp, C.c_uint64, C.c_uint64],
        "sbr_device_info": [C.c_char_p, C.c_uint64, u32p, u64p],
        "sbr_model_timing_enable": [vp, C.c_int32],
        "sbr_model_timing_select": [vp, C.c_uint32],
        "sbr_model_set_overlap": [vp, C.c_int32],
        "sbr_model_timing_read": [vp, C.POINTER(C.c_double), u64p],
        "sbr_set_device": [C.c_int32],
        "sbr_group_fit": [C.POINTER(vp), C.c_uint32, vp, vp, C.c_uint64, fp],
        "sbr_device_count": [C.POINTER(C.c_int32)],
        "sbr_group_fit_begin": [C.POINTER(vp), C.c_uint32, vp, vp, C.c_uint64, C.POINTER(vp)],
        "sbr_group_epoch_prepare": [vp, u64p, C.c_int32],
        "sbr_group_step": [vp, C.c_uint64],
        "sbr_group_step_local": [vp, C.c_uint64],
        "sbr_group_member_plan": [vp, C.c_uint32, C.POINTER(vp)],
        "sbr_group_synchronize": [vp],
        "sbr_group_plan_set_host_threads": [vp, C.c_int32],
        "sbr_group_plan_stats": [vp, C.POINTER(C.c_double), u64p, C.POINTER(C.c_int32)],
        "sbr_group_fit_end": [vp, fp],
        "sbr_group_create": [C.POINTER(SbrHparams), C.c_uint32, C.c_uint32, C.POINTER(vp)],
        "sbr_model_is_partitioned": [vp, C.POINTER(C.c_int32)],
        "sbr_fit_exchange_export": [vp, C.POINTER(C.c_int32), u64p],
        "sbr_fit_exchange_import": [vp, C.c_uint32, C.POINTER(C.c_int32), u64p],
        "sbr_fit_step_scatter_shared": [vp, C.c_uint64],
        "sbr_fit_step_owner_reduce_peers": [vp],
        "sbr_fit_step_apply_table_peers": [vp, vp],
        "sbr_model_create_partitioned": [C.POINTER(SbrHparams), C.POINTER(vp)],
        "sbr_partition_num_parts": [vp, u32p],
        "sbr_partition_part_info": [vp, C.c_uint32, u32p, u64p],
        "sbr_partition_export_part": [vp, C.c_uint32, C.POINTER(C.c_int32)],
        "sbr_partition_import_part": [vp, C.c_uint32, C.c_int32],
        "sbr_partition_finalize": [vp],
        "sbr_fit_lists_export": [vp, C.POINTER(C.c_int32), u64p],
        "sbr_fit_lists_import": [vp, C.c_uint32, C.POINTER(C.c_int32), u64p],
        "sbr_fit_step_reduce_own": [vp, C.c_uint64, u32p, vp],
        "sbr_fit_step_owner_apply": [vp, u32p, vp],
        "sbr_fit_step_reduce_own_queued": [vp, C.c_uint64, C.POINTER(vp), vp],
        "sbr_fit_step_owner_apply_queued": [vp, vp, vp],
        "sbr_selftest_math": [vp, C.c_uint64, vp, vp, vp],
        "sbr_selftest_dot_tree": [vp, vp, C.c_uint32, C.c_uint64, vp],
        "sbr_selftest_mfma": [vp, vp, vp, C.c_uint32, vp, vp, vp, vp],
        "sbr_selftest_sort": [vp, C.c_uint64, C.c_uint32, vp, vp, vp],
    }
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = C.c_int
    L.sbr_model_destroy.argtypes = [vp]
    L.sbr_model_destroy.restype = None
    L.sbr_fit_plan_destroy.argtypes = [vp]
    L.sbr_fit_plan_destroy.restype = None
    L.sbr_comm_destroy.argtypes = [vp]
    L.sbr_comm_destroy.restype = None
    L.sbr_group_plan_destroy.argtypes = [vp]
    L.sbr_group_plan_destroy.restype = None
    L.sbr_status_string.argtypes = [C.c_int]
    L.sbr_status_string.restype = C.c_char_p
    L.sbr_abi_version.argtypes = []
    L.sbr_abi_version.restype = C.c_uint32
    L.sbr_release_cached_memory.argtypes = []
    L.sbr_release_cached_memory.restype = None
    if L.sbr_abi_version() != ABI_VERSION:
        raise EngineUnavailable("libsbr_hip.so ABI version mismatch; rebuild with `python -m sbr_rs_amd.build`")
    _lib = L
    return L


# every symbol include/sbr_hip.h declares (tests/test_abi.py checks the export table)
DECLARED_SYMBOLS = [
    "sbr_model_create", "sbr_model_destroy", "sbr_model_fit", "sbr_fit_begin", "sbr_fit_epoch_prepare",
    "sbr_fit_epoch_prefetch",
    "sbr_fit_step", "sbr_fit_minibatch_rows", "sbr_fit_end", "sbr_fit_end_lagged", "sbr_model_last_fit_lagged_loss", "sbr_fit_counters", "sbr_model_get_rng", "sbr_model_set_rng", "sbr_fit_sparse_stats", "sbr_fit_plan_destroy", "sbr_fit_chunk_bytes", "sbr_fit_dense_bytes",
    "sbr_fit_step_local", "sbr_fit_step_apply", "sbr_fit_step_scatter", "sbr_fit_step_dense", "sbr_fit_step_owner_reduce", "sbr_fit_step_owner_reduce_on", "sbr_fit_step_apply_table", "sbr_fit_step_apply_rows", "sbr_fit_step_apply_dense", "sbr_model_set_stream", "sbr_model_synchronize",
    "sbr_fit_debug_fetch", "sbr_user_representation", "sbr_predict", "sbr_mrr_score", "sbr_model_param_count",
    "sbr_model_get_param", "sbr_model_set_param", "sbr_model_get_param_rows", "sbr_model_get_epoch", "sbr_model_get_counters", "sbr_model_set_counters", "sbr_device_info", "sbr_status_string",
    "sbr_abi_version", "sbr_model_timing_enable", "sbr_model_timing_select", "sbr_model_set_overlap", "sbr_model_timing_read", "sbr_set_device", "sbr_group_fit", "sbr_device_count", "sbr_group_create", "sbr_model_is_partitioned", "sbr_model_create_partitioned", "sbr_partition_num_parts", "sbr_fit_exchange_export", "sbr_fit_exchange_import",
    "sbr_fit_step_scatter_shared", "sbr_fit_step_owner_reduce_peers", "sbr_fit_step_apply_table_peers",
    "sbr_partition_part_info", "sbr_partition_export_part", "sbr_partition_import_part", "sbr_partition_finalize",
    "sbr_fit_lists_export", "sbr_fit_lists_import", "sbr_fit_step_reduce_own", "sbr_fit_step_owner_apply", "sbr_selftest_math",
    "sbr_selftest_dot_tree", "sbr_selftest_mfma", "sbr_selftest_sort", "sbr_release_cached_memory",
    "sbr_group_fit_begin", "sbr_group_epoch_prepare", "sbr_group_step", "sbr_group_step_local", "sbr_group_member_plan",
    "sbr_fit_steps", "sbr_comm_unique_id", "sbr_comm_create", "sbr_comm_destroy", "sbr_fit_step_exchange", "sbr_model_fit_comm", "sbr_model_set_reference_order", "sbr_fit_block_bytes", "sbr_fit_step_apply_blocks_in_order", "sbr_model_set_step_fusion", "sbr_fit_debug_phase_clocks", "sbr_group_synchronize", "sbr_group_plan_set_host_threads", "sbr_group_plan_stats", "sbr_group_fit_end", "sbr_group_plan_destroy",
    "sbr_fit_step_owner_update", "sbr_model_table_slice", "sbr_model_optimizer_state_gathered", "sbr_model_optimizer_state_is_partial",
    "sbr_group_plan_set_exchange", "sbr_group_gather_optimizer_state", "sbr_comm_gather_optimizer_state",
    "sbr_fit_step_reduce_own_queued", "sbr_fit_step_owner_apply_queued",
]
