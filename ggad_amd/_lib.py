"""ctypes binding of libggad_hip.so (the C-ABI declared in include/ggad_hip.h).

The product path has NO fallback: if the shared library is missing this module raises
(``GgadLibraryError``) instead of routing to a CPU / PyTorch implementation.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int32, c_int64, c_uint32, c_uint64, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GGAD_LIB_PATH") or os.path.join(HERE, "libggad_hip.so")      # (GGAD_LIB_PATH: experiment builds)


class GgadLibraryError(RuntimeError):
    pass


# status codes of include/ggad_hip.h
GGAD_OK, GGAD_E_INVALID, GGAD_E_LAUNCH, GGAD_E_CAPACITY, GGAD_E_UNSUPPORTED = 0, -1, -2, -3, -4


class GgadKernelError(RuntimeError):
    pass


ABI_VERSION = 10   # what this binding was written against (include/ggad_hip.h, runtime.cpp); `load` refuses any other library
_P = c_void_p      # device (or host) pointer
EXCHANGE_CB = ctypes.CFUNCTYPE(c_int32, c_void_p)      # int exchange(void *user): the data-parallel all-reduce hook
_I = c_int32
_L = c_int64
_F = c_float

# name -> (restype, argtypes); MUST mirror include/ggad_hip.h (tests/test_abi.py checks the symbol list)
SIGNATURES = {
    "ggad_abi_version": (c_int32, []),
    "ggad_last_error": (c_char_p, []),
    "ggad_max_embed_dim": (c_int32, []),
    "ggad_max_feat_dim": (c_int32, []),
    "ggad_scan_workspace_elems": (c_int64, [_L]),
    "ggad_exclusive_scan_i32": (c_int32, [_P, _P, _L, _P, _P]),
    "ggad_mb_chunk_len": (c_int32, []),
    "ggad_mb_slice_len": (c_int32, []),
    "ggad_mb_group_words": (c_int32, []),
    "ggad_mb_item_words": (c_int32, []),
    "ggad_mb_set_gather_options": (c_int32, [c_int32, c_int32]),
    "ggad_mb_plan_counter_elems": (c_int32, []),
    "ggad_mb_plan_build": (c_int32, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _P]),
    "ggad_event_create": (c_int32, [_I, _P]),
    "ggad_event_destroy": (c_int32, [_P]),
    "ggad_event_record": (c_int32, [_P, _P]),
    "ggad_event_synchronize": (c_int32, [_P]),
    "ggad_event_elapsed_ms": (c_int32, [_P, _P, _P]),
    "ggad_mb_count2": (c_int32, [_P, _P, _P, _P, _P, _L, _L, _P, _P, _P]),
    "ggad_mb_gather2": (c_int32, [_P, _P, _P, _I, _I, _P, _P, _P, _P, _L, _L, _P, _P, _P]),
    "ggad_mb_plan_reset": (c_int32, [_P, _P, _P, _P, _P, _P, _L, _L, _P, _P, _I, _P]),
    "ggad_mb_tile_offsets_elems": (c_int64, [_L, _I]),
    "ggad_mb_tile_offsets": (c_int32, [_P, _P, _L, _I, _P, _P]),
    "ggad_mb_tile_major_workspace_elems": (c_int64, [_L, _I]),
    "ggad_mb_tile_major": (c_int32, [_P, _P, _L, _I, _P, _P, _P, _P, _P]),
    "ggad_mb_ldsw_tile_shift": (c_int32, []),
    "ggad_mb_ldsw_max_owners": (c_int32, []),
    "ggad_mb_ldsw_seg_elems": (c_int64, [_L, _L]),
    "ggad_mb_dw_part_elems": (c_int64, [_I, _I, _I]),
    "ggad_mb_train_chunk": (c_int32, [_P, _I, _P, _P, _P, _P, _I, _I, _P]),
    "ggad_mb_train_chunk_dp": (c_int32, [_P, _I, _P, _P, _P, _P, _I, c_float, EXCHANGE_CB, _P, _P]),
    "ggad_stream_create_cu_mask": (c_int32, [_P, _I, _P]),
    "ggad_stream_destroy": (c_int32, [_P]),
    "ggad_device_cu_count": (c_int32, [_I, _P]),
    "ggad_xchg_create": (c_int32, [_I, _I, _L, _P]),
    "ggad_xchg_handle_bytes": (c_int32, []),
    "ggad_xchg_handle": (c_int32, [_P, _P]),
    "ggad_xchg_connect": (c_int32, [_P, _P]),
    "ggad_xchg_error": (c_int32, [_P, _P]),
    "ggad_xchg_destroy": (c_int32, [_P]),
    "ggad_xchg_adam": (c_int32, [_P, _P, _P, _P, _I, _I, _F, _F, _F, _P, _P, _P]),
    "ggad_mb_train_chunk_xchg": (c_int32, [_P, _I, _P, _P, _P, _P, _I, c_float, _P, _P]),
    "ggad_mb_xcd_grid": (c_int32, []),
    "ggad_xcd_first_of_stream": (c_int32, [_P, _P]),
    "ggad_mb_xcd_workspace_elems": (c_int64, [_I, _I, _I, _L, _L]),
    "ggad_mb_train_chunk_xcd": (c_int32, [_P, _I, _P, _I, _I, _I, _I, _L, _L, _P, _I, _P, c_float, _P, _I, _P, _P]),
    "ggad_mb_xcd_record_elems": (c_int64, [_L, _L]),
    "ggad_mb_xcd_prepare": (c_int32, [_P, _I, _P, _I, _I, _I, _L, _L, _P, _P]),
    "ggad_mb_xcd_status": (c_int32, [_P, _P, _P]),
    "ggad_mb_xcd_clear_error": (c_int32, [_P, _I, _P]),
    "ggad_mb_param_count": (c_int64, [_I, _I]),
    "ggad_mb_param_block_elems": (c_int64, [_I, _I]),
    "ggad_mb_params_sync": (c_int32, [_P, _I, _I, _P]),
    "ggad_mb_project": (c_int32, [_P, _I, _I, _P, _P, _I, _I, _P, _P]),
    "ggad_mb_fwd_rows": (c_int32, [_P, _I, _I, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "ggad_mb_loss_workspace_elems": (c_int64, [_I]),
    "ggad_mb_loss": (c_int32, [_P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ggad_mb_row_coefs": (c_int32, [_P, _I, _I, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ggad_mb_bwd_parts": (c_int32, []),
    "ggad_mb_bwd_flat": (c_int32, [_I, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "ggad_mb_grad_reduce": (c_int32, [_I, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "ggad_mb_adam": (c_int32, [_P, _P, _P, _P, _I, _I, _F, _F, _F, _P, _P]),
    "ggad_mb_train_step": (c_int32, [_P, _I, _P]),
    "ggad_mb_encode": (c_int32, [_P, _I, _I, _P, _I, _P, _P]),
    "ggad_seg_mean": (c_int32, [_P, _I, _P, _P, _I, _P, _P]),
    "ggad_seg_wsum": (c_int32, [_P, _I, _P, _P, _P, _I, _P, _P]),
    "ggad_recon_cols_f32": (c_int32, [_P, _P, _I, _I, _F, _F, _P, _P, _P, _P]),
    "ggad_recon_rows_f32": (c_int32, [_P, _P, _L, _I, _P, _P]),
    "ggad_ocgnn_loss_f32": (c_int32, [_P, _P, _L, _I, _P, _F, _F, _P, _P, _P, _P]),
    "ggad_mb_score": (c_int32, [_P, _I, _I, _P, _I, _P, _P]),
    "ggad_gemm_workspace_elems": (c_int64, [_I, _I, _I]),
    "ggad_gemm_f32": (c_int32, [_P, _P, _P, _I, _I, _I, _L, _L, _L, _L, _L, _P, _I, _P, _P]),
    "ggad_spmm_seg_len": (c_int32, []),
    "ggad_spmm_csr_f32": (c_int32, [_P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _P, _L, _I, _P, _P, _P, _L, _P, _P, _P]),
    "ggad_spmm_rowslice_group": (c_int32, []),
    "ggad_spmm_rowslice_short": (c_int32, []),
    "ggad_spmm_rowslice_long": (c_int32, []),
    "ggad_spmm_rowslice_f32": (c_int32, [_P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P, _I, _P, _L, _I, _P, _P, _P, _L, _P, _P]),
    "ggad_spmm_rowline_supported": (c_int32, [_P, _L, _I, _L]),
    "ggad_spmm_rowline_f32": (c_int32, [_P, _P, _I, _P, _I, _P, _I, _P, _L, _I, _L, _P, _P, _P, _L, _P, _P]),
    "ggad_prelu_bwd_ld_f32": (c_int32, [_P, _P, _P, _I, _I, _P, _L, _P, _P, _P, _P]),
    "ggad_prelu_bwd_one_f32": (c_int32, [_P, _P, _P, _I, _I, _P, _L, _P, _P, _P, _P, _P]),
    "ggad_prelu_bwd_one_workspace_elems": (c_int64, [_I, _I]),
    "ggad_prelu_bwd_one_tickets": (c_int32, []),
    "ggad_spmm_sliced_workspace_elems": (c_int64, [_L, _I]),
    "ggad_spmm_sliced_seg_len": (c_int32, []),
    "ggad_spmm_panel_available": (c_int32, []),
    "ggad_spmm_panel_rows": (c_int32, []),
    "ggad_spmm_panel_waves": (c_int32, []),
    "ggad_spmm_panel_rounds": (c_int32, []),
    "ggad_spmm_panel_values_factor": (c_int32, [_P, _P, _P, _P, _I, ctypes.c_double, _I]),
    "ggad_spmm_panel_count": (c_int32, [_P, _P, _I, _P, _P, _I, _I, _I, _P, _I]),
    "ggad_spmm_panel_fill": (c_int32, [_P, _P, _I, _P, _P, _I, _I, _I, _P, _P, _P, _L, _I, _I]),
    "ggad_spmm_panel_f32": (c_int32, [_P, _I, _P, _P, _P, _I, _P, _P, _P, _P, _L, _I, _L, _P, _P, _P, _P, _L, _P, _P]),
    "ggad_spmm_sliced_f32": (c_int32, [_P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _P, _L, _I, _L, _P, _P, _P, _P, _L, _P, _P, _P]),
    "ggad_spmm_ring_available": (c_int32, []),
    "ggad_spmm_ring_slot_rows": (c_int32, []),
    "ggad_spmm_ring_slots": (c_int32, []),
    "ggad_spmm_ring_window": (c_int32, []),
    "ggad_spmm_ring_walkers": (c_int32, []),
    "ggad_spmm_ring_rounds": (c_int32, []),
    "ggad_spmm_ring_count": (c_int32, [_P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P, _I]),
    "ggad_spmm_ring_fill": (c_int32, [_P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _L, _I]),
    "ggad_spmm_ring_walkers_subset": (c_int32, []),
    "ggad_spmm_ring_f32": (c_int32, [_P, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _L, _I, _L, _P, _P, _P, _P, _L, _P, _P]),
    "ggad_mlp_score_supported": (c_int32, [_I, _I, _I]),
    "ggad_mlp_score_fwd_f32": (c_int32, [_P, _L, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "ggad_mlp_score_wgrad_workspace_elems": (c_int64, [_I, _I, _I, _I]),
    "ggad_mlp_score_wgrad_f32": (c_int32, [_P, _L, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "ggad_mlp_score_dgrad_f32": (c_int32, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _L, _P]),
    "ggad_prelu_bwd_splits": (c_int32, [_I]),
    "ggad_prelu_bwd_f32": (c_int32, [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P]),
    "ggad_relu_bwd_f32": (c_int32, [_P, _P, _L, _P, _P]),
    "ggad_prelu_fwd_f32": (c_int32, [_P, _P, _L, _P, _P]),
    "ggad_linear_prelu_f32": (c_int32, [_P, _L, _P, _L, _P, _P, _I, _I, _I, _P, _L, _P, _L, _P]),
    "ggad_rownorm_f32": (c_int32, [_P, _I, _I, _P, _P, _P]),
    "ggad_rownorm_bwd_f32": (c_int32, [_P, _P, _P, _I, _I, _P, _P]),
    "ggad_rowdot_f32": (c_int32, [_P, _P, _P, _I, _I, _P, _P, _P]),
    "ggad_edge_dist_f32": (c_int32, [_P, _P, _P, _I, _I, _P, _P]),
    "ggad_rows_scale_f32": (c_int32, [_P, _P, _P, _I, _I, _I, _P, _P]),
    "ggad_full_loss_bwd_scale_f32": (c_int32, [_P, _P, _P, _P, _P, _I, _L, _P, _P, _P, _P, _P]),
    "ggad_head_gather_f32": (c_int32, [_P, _P, _P, _I, _I, _P, _P]),
    "ggad_head_combine_f32": (c_int32, [_P, _P, _I, _P, _I, _I, _P, _P]),
    "ggad_head_rows_f32": (c_int32, [_P, _P, _P, _I, _P, _I, _I, _P, _P, _P]),
    "ggad_head_emb_put_f32": (c_int32, [_P, _P, _I, _I, _P, _P]),
    "ggad_head_emb_out_f32": (c_int32, [_P, _P, _P, _I, _I, _P, _P]),
    "ggad_head_con_grad_f32": (c_int32, [_P, _P, _P, _P, _P, _I, _I, _P, _P]),
    "ggad_head_emb_grad_f32": (c_int32, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P]),
    "ggad_full_loss_workspace_elems": (c_int64, [_I, _I]),
    "ggad_full_loss_fused_workspace_elems": (c_int64, [_I, _I]),
    "ggad_full_loss_fused_f32": (c_int32, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ggad_full_loss_bwd_fused_f32": (c_int32, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P]),
    "ggad_rownorm_bwd_add_f32": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P]),
    "ggad_full_loss_f32": (c_int32, [_P, _P, _I, _I, _P, _P, _I, _F, _P, _P, _P, _P, _P, _P]),
    "ggad_adam_f32": (c_int32, [_P, _P, _P, _P, _L, _F, _F, _P, _I, _P]),
    "ggad_adam_multi_max": (c_int32, []),
    "ggad_adam_multi_f32": (c_int32, [_I, _P, _P, _P, _P, _P, _P, _F, _F, _P, _P]),
    "ggad_mt_new": (c_void_p, []),
    "ggad_mt_free": (None, [c_void_p]),
    "ggad_mt_seed_u64": (c_int32, [c_void_p, c_uint64]),
    "ggad_mt_set_state": (c_int32, [c_void_p, POINTER(c_uint32), c_int32]),
    "ggad_mt_get_state": (c_int32, [c_void_p, POINTER(c_uint32), POINTER(c_int32)]),
    "ggad_mt_shuffle_i64": (c_int32, [c_void_p, POINTER(c_int64), c_int64]),
    "ggad_mt_shuffle_targets": (c_int32, [c_void_p, c_int64, c_void_p]),
    "ggad_apply_swaps_i64": (c_int32, [c_void_p, c_int64, c_void_p]),
    "ggad_mt_getrandbits32": (c_uint32, [c_void_p]),
    "ggad_sched_batches": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_int32,
                                     c_void_p, c_void_p]),
}



class MbStep(ctypes.Structure):
    """Mirror of `ggad_mb_step` (include/ggad_hip.h)."""
    _fields_ = ([(n, c_void_p) for n in ("params", "exp_avg", "exp_avg_sq", "grads", "step_counter", "x1", "x2",
                                         "ent_ptr", "ent_own", "ent_row", "labels", "pos_meta", "row_pos", "h1", "nbar",
                                         "gen", "dz", "coef_a", "coef_g", "h2", "dw_part", "loss_ws", "losses8")]
                + [(n, c_int32) for n in ("D", "F", "row0", "n_rows", "ent0", "n_ents")]
                + [("lr", c_float), ("weight_decay", c_float), ("chain", c_int32), ("max_row_entries", c_int32)]
                + [(n, c_void_p) for n in ("row_ck_ptr", "ck_rc", "ck_e0", "chunk_part")])


class MbPlan(ctypes.Structure):
    """Mirror of `ggad_mb_plan` (include/ggad_hip.h)."""
    _fields_ = ([(n, c_void_p) for n in ("rowptr", "col", "feat", "tile_off", "closed_deg_host", "pair_bound_host", "stage_host",
                                         "stage", "stage_event", "cnt1", "own1", "cnt2", "ent_col", "ent_slot", "ent_row",
                                         "ent_own", "ent_c1", "x1", "x2", "ck_part", "own_deg", "own_rp", "pw_base", "seg_t",
                                         "node_head", "own_next", "grp", "items", "counters", "pc", "part2", "ev_gather0",
                                         "ev_gather1")]
                + [(n, c_int64) for n in ("n_nodes", "ent_cap", "ck_cap", "pair_cap", "item_cap", "part2_cap", "stage_cap",
                                          "seg_cap")]
                + [(n, c_int32) for n in ("feat_dim", "feat_stride", "max_batches", "rows_cap", "ck_part_stride", "train", "hop2",
                                          "node_major")]
                + [("mean_nbr_deg", c_float), ("xcd_skip", c_int32)]
                + [(n, c_void_p) for n in ("ev_tile0", "ev_tile1", "node_pack_host", "tile_start", "col_t")])


class MbPlanInfo(ctypes.Structure):
    """Mirror of `ggad_mb_plan_info`."""
    _fields_ = ([(n, c_int64) for n in ("pair_bound", "off_batch_ptr", "off_batch_ent_ptr", "off_nodes", "off_labels",
                                        "off_pos_meta", "off_row_pos", "off_row_slot", "off_ent_ptr", "off_row_ck_ptr",
                                        "off_ck_rc", "off_ck_e0", "need_rows", "need_ents", "need_chunks", "need_pairs",
                                        "need_items", "need_part2", "need_stage", "need_seg")]
                + [(n, c_int32) for n in ("n_batches", "n_rows", "n_ents", "n_chunks", "mode", "need_cnt2")])


_lib = None


def load(path: str = LIB_PATH) -> ctypes.CDLL:
    """Load the library (once) and attach argtypes.  Raises GgadLibraryError if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise GgadLibraryError(
            f"{path} not found: build it with `python -m ggad_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback for the GGAD hot path.")
    # torch ships its own libamdhip64: it must be in the process BEFORE this library is loaded, otherwise the loader binds
    # libggad_hip.so to /opt/rocm's copy and its launches run in a second HIP runtime that knows nothing of torch's
    # device pointers and streams ("no ROCm-capable device is detected" on the first launch)
    import torch  # noqa: F401
    try:
        lib = ctypes.CDLL(path)
    except OSError as exc:  # missing ROCm runtime etc.
        raise GgadLibraryError(f"cannot load {path}: {exc}") from exc
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise GgadLibraryError(f"{path} does not export {name}; rebuild it") from exc
        fn.restype = res
        fn.argtypes = args
    if int(lib.ggad_abi_version()) != ABI_VERSION:
        raise GgadLibraryError(f"{path} has ABI {int(lib.ggad_abi_version())}, this binding needs {ABI_VERSION}; rebuild it "
                               "(`python -m ggad_amd.build --force`)")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().ggad_last_error()
        raise GgadKernelError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")


def ptr(t) -> int:
    """Device pointer of a torch tensor (must be contiguous)."""
    if t is None:
        return 0
    if not t.is_contiguous():
        raise ValueError("tensor handed to the C-ABI must be contiguous")
    return t.data_ptr()


def ptr_rows(t) -> int:
    """Device pointer of a 2-D tensor whose rows are contiguous (unit column stride; the row stride is passed beside it)."""
    if t.dim() != 2 or t.stride(1) != 1 or t.stride(0) < t.shape[1]:
        raise ValueError("tensor handed to the C-ABI must have contiguous rows")
    return t.data_ptr()


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def call(name: str, *args) -> None:
    """Call an int-returning kernel entry point on torch's current stream and raise on error."""
    lib = load()
    rc = getattr(lib, name)(*args, current_stream())
    check(rc, name)
