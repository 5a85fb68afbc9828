"""ctypes binding of libctrlsim_hip.so (the C ABI of include/ctrlsim.h).

The product path has NO fallback: if the HIP library is missing or cannot be loaded this module raises — callers
must build it first (`python ctrl-sim_amd/csrc/build.py`, done by `__graft_entry__.build()`).
`import torch` happens before the CDLL so that the library binds to the HIP runtime torch already loaded
(both carry soname libamdhip64.so.7).
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL: shares the already-loaded HIP runtime)

HERE = os.path.dirname(os.path.abspath(__file__))
# CTRLSIM_LIB: another build of the SAME library (tools/: ablation variants of one kernel source); never a fallback — a missing
# file raises in lib() either way
LIB_PATH = os.environ.get("CTRLSIM_LIB") or os.path.join(HERE, "csrc", "libctrlsim_hip.so")


class Dims(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("A", "T", "P", "NP", "D", "H", "F", "V", "R", "C", "NE", "ND", "MAXT", "variant", "flags")]


class DtRewardCfg(C.Structure):
    """ctrlsim_dt_reward_cfg (include/ctrlsim.h)."""
    _fields_ = [(k, C.c_double) for k in ("pos_tol", "shaped_unit", "goal_mult", "shaped_min", "shaped_max", "veh_mult",
                                          "max_veh_dist", "edge_mult", "edge_scale")] + \
               [("rtg_lo", C.c_double * 3), ("rtg_hi", C.c_double * 3)] + \
               [(k, C.c_int) for k in ("remove_shaped_goal", "remove_shaped_veh", "remove_shaped_edge", "pad_")]


class Ctx(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("st12", "exist", "goal5", "act_tok", "rtg_bin", "tstep", "slot_gid",
                                           "road_pts", "road_types")]


P, I, L, D, F = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_float
U64 = C.c_uint64

SIGNATURES = {
    "ctrlsim_version": (C.c_char_p, []),
    "ctrlsim_set_option": (I, [I, I]),
    "ctrlsim_split_scheme": (I, []),
    "ctrlsim_nonfinite_count": (I, [I]),
    "ctrlsim_bind": (I, [I, P]),
    "ctrlsim_unbind": (I, [P]),
    "ctrlsim_bound_guard": (P, []),
    "ctrlsim_option_count": (I, []),
    "ctrlsim_bind_options": (I, [P]),
    "ctrlsim_get_option": (I, [I]),
    "ctrlsim_prof_classes": (I, []),
    "ctrlsim_prof_enable": (None, [I]),
    "ctrlsim_prof_collect": (I, [P, P, P]),
    "ctrlsim_prof_bytes": (I, [P]),
    "ctrlsim_prof_collect_stream": (I, [P, I, P, P, P, P]),
    "ctrlsim_prof_subclasses": (I, []),
    "ctrlsim_prof_collect_sub": (I, [P, I, P, P, P, P]),
    "ctrlsim_metrics_size": (I, []),
    "ctrlsim_dt_ledger_step": (I, [I, I, I, I, I, I, P, P, P, P, P, C.POINTER(DtRewardCfg), P, P, P, P]),
    "ctrlsim_metrics_pack": (I, [I, I, I, I, I, D, P, P, P, P, P, P, P, P, P, P]),
    "ctrlsim_gemm_nt": (I, [P, I, P, I, P, P, I, P, I, I, I, I, I, P]),
    "ctrlsim_gemm_nt_bf16x6": (I, [P, I, P, I, I, P, P, I, P, I, I, I, I, I, P, P, P]),
    "ctrlsim_gemm256_rows": (I, [P, I, P, I, I, P, P, I, P, I, P]),
    "ctrlsim_gemm_nt_kv": (I, [P, I, P, I, I, P, P, I, I, I, I, P, I, I, I, P]),
    "ctrlsim_gemm_kv_blocks": (I, [P, I, P, P, P, I, I, I, P, I, I, I, P]),
    "ctrlsim_ffn_fused": (I, [P, I, P, P, P, P, P, P, P, I, I, I, P]),
    "ctrlsim_ffn_fused_pre": (I, [P, I, P, I, P, P, P, P, P, P, P, P, P, P, P, I, I, I, P]),
    "ctrlsim_outproj_ln_q": (I, [P, I, P, I, P, P, P, P, P, P, P, I, P, I, I, P]),
    "ctrlsim_layernorm256": (I, [P, I, P, I, P, P, P, I, I, I, P]),
    "ctrlsim_kv_split": (I, [P, P, I, L, P, I, I, I, P, P]),
    "ctrlsim_attention_presplit": (I, [I, P, I, L, P, I, P, I, L, P, P, I, I, I, I, P]),
    "ctrlsim_attention": (I, [I, P, I, L, P, P, I, L, P, I, L, P, P, I, I, I, I, P]),
    "ctrlsim_sim_init": (I, [I, I, I, P, P, P, P, P, P, P, I, P, P]),
    "ctrlsim_sim_contact_floats": (L, [I]),
    "ctrlsim_sim_step": (I, [I, I, I, P, P, P, P, P, P, P, P, P, P, I, I, F, I, P, P]),
    "ctrlsim_sim_step_expert": (I, [I, I, I, P, P, P, P, P, P, P, P, P, P, I, I, F, P, P, P]),
    "ctrlsim_sim_set_position": (I, [I, I, P, P, P]),
    "ctrlsim_group_build": (I, [I, I, I, I, I, I, D, P, P, I, P, P, P, P, P, P, P, P, P]),
    "ctrlsim_groups_changed": (I, [I, I, P, P, P, P, P, P, P, P]),
    "ctrlsim_group_size_hist": (I, [I, I, P, P, I, P, P, P]),
    "ctrlsim_ctx_index_classes": (I, [I, I, I, I, P, P, P, P, I, P, P, P, P, P, P, P, P, P, P]),
    "ctrlsim_ctx_index": (I, [I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "ctrlsim_build_context": (I, [I] * 12 + [P] * 12 + [C.POINTER(Ctx), P]),
    "ctrlsim_build_context_c": (I, [I, P, P, C.POINTER(Ctx)] + [I] * 10 + [P] * 12 + [P]),
    "ctrlsim_model_create": (I, [C.POINTER(Dims), P, I, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(P)]),
    "ctrlsim_model_destroy": (None, [P]),
    "ctrlsim_forward_workspace_bytes": (L, [C.POINTER(Dims), I, I]),
    "ctrlsim_forward_workspace_bytes_a": (L, [C.POINTER(Dims), I, I, I]),
    "ctrlsim_forward_workspace_bytes_c": (L, [C.POINTER(Dims), I, P, P, I]),
    "ctrlsim_dt_forward_pass1_c": (I, [P, I, P, P, P, I, P, P, P, P]),
    "ctrlsim_dt_forward_pass1_c2": (I, [P, I, P, P, P, I, P, P, P, P]),
    "ctrlsim_dt_forward_pass2_c": (I, [P, I, P, P, P, I, I, I, I, P, P, P, P, I, P]),
    "ctrlsim_dt_forward_pass1_cached_c": (I, [P, I, P, P, P, I, P, P, P]),
    "ctrlsim_dt_forward_pass1_a": (I, [P, I, I, I, C.POINTER(Ctx), P, P, P, P]),
    "ctrlsim_dt_forward_pass2_a": (I, [P, I, I, I, I, I, I, C.POINTER(Ctx), P, P, P, P, I, P]),
    "ctrlsim_dt_forward_pass1_cached_a": (I, [P, I, I, I, C.POINTER(Ctx), P, P, P]),
    "ctrlsim_attention_compact": (I, [P, I, L, P, I, P, I, L, P, I, I, I, I, I, I, I, P]),
    "ctrlsim_attn_class_prof": (I, [I, P]),
    "ctrlsim_attention_mask_table_bytes": (L, [I, I]),
    "ctrlsim_attention_mask_table": (I, [I, I, I, I, I, I, P, P]),
    "ctrlsim_attention_tbl": (I, [P, I, L, P, I, P, I, L, I, I, I, I, I, I, P, P]),
    "ctrlsim_sample_rtg_rows": (I, [P, P, I, P, P, P, P, P, P, U64, P, I, P, I, I, I, P]),
    "ctrlsim_sample_action_rows": (I, [P, P, I, P, P, F, D, P, U64, P, I, P, P, I, I, I, I, P]),
    "ctrlsim_dt_forward_pass1": (I, [P, I, I, C.POINTER(Ctx), P, P, P, P]),
    "ctrlsim_dt_forward_actions": (I, [P, I, I, C.POINTER(Ctx), P, P, P]),
    "ctrlsim_map_pool": (I, [P, I, P, P, P, P]),
    "ctrlsim_forward_all": (I, [P, I, I, C.POINTER(Ctx), P, P, P, P, P]),
    "ctrlsim_dt_forward_pass2": (I, [P, I, I, I, I, I, C.POINTER(Ctx), P, P, P, P, I, P]),
    "ctrlsim_dt_forward_pass1_cached": (I, [P, I, I, C.POINTER(Ctx), P, P, P]),
    "ctrlsim_sample_rtg": (I, [P, I, I, P, P, P, P, P, P, U64, P, I, P, I, I, I, P]),
    "ctrlsim_sample_action": (I, [P, I, I, P, P, F, D, P, U64, P, I, P, P, I, I, I, I, P]),
}

_lib = None


def lib():
    """Load (once) and return the CDLL with argtypes set.  Raises if the HIP extension is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build the HIP extension first "
                               "(python ctrl-sim_amd/csrc/build.py); there is no CPU fallback")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)      # AttributeError if an ABI symbol is missing
            fn.restype = res
            fn.argtypes = args
        for kv in filter(None, os.environ.get("CTRLSIM_OPTIONS", "").split(",")):   # "<option>=<value>,...": kernel A/B runs only
            k, v = kv.split("=")
            l.ctrlsim_set_option(int(k), int(v))
        _lib = l
    return _lib


def check(code, what=""):
    if code != 0:
        raise RuntimeError(f"ctrlsim call failed ({what}): status {code}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "ctrlsim kernels take contiguous tensors"
    return t.data_ptr()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream
