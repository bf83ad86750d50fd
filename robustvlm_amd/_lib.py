"""ctypes binding of librvlm.so (the C ABI declared in include/rvlm.h).

The library is hand-written HIP for gfx950, built in-tree by ``robustvlm_amd/csrc/Makefile``
(``__graft_entry__.build()``).  There is NO fallback: if the shared object is missing or a call
fails, this module raises - the product path never routes through a CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RVLM_LIB_PATH: development knob for A/B runs of two builds on one box (scripts/trip_ab.sh); never a fallback
LIB_PATH = os.path.abspath(os.environ["RVLM_LIB_PATH"]) if os.environ.get("RVLM_LIB_PATH") else os.path.join(_HERE, "librvlm.so")

RVLM_OK, RVLM_ERR_ARG, RVLM_ERR_HIP, RVLM_ERR_STATE, RVLM_ERR_UNSUPPORTED = 0, 1, 2, 3, 4
PREC_F32, PREC_BF16, PREC_F32X3 = 0, 1, 2
ACT_QUICK_GELU, ACT_GELU = 0, 1
LOSS_L2, LOSS_CE, LOSS_DLR, LOSS_DLR_TARGETED = 0, 1, 2, 3
RED_MEAN, RED_NONE = 0, 1
FLAG_INPUT_RANGE, FLAG_NAN_GRAD, FLAG_NAN_DELTA, FLAG_ADV_RANGE = 1, 2, 4, 8

c_f32p = C.c_void_p   # device pointers travel as integers (tensor.data_ptr())
c_stream = C.c_void_p


class VitConfigC(C.Structure):
    _fields_ = [("image_size", C.c_int32), ("patch", C.c_int32), ("width", C.c_int32),
                ("layers", C.c_int32), ("heads", C.c_int32), ("out_dim", C.c_int32),
                ("act", C.c_int32), ("precision", C.c_int32), ("max_batch", C.c_int32),
                ("mean", C.c_float * 3), ("std", C.c_float * 3), ("trainable", C.c_int32)]


BLOCK_FIELDS = ["ln_1_weight", "ln_1_bias", "attn_in_proj_weight", "attn_in_proj_bias",
                "attn_out_proj_weight", "attn_out_proj_bias", "ln_2_weight", "ln_2_bias",
                "mlp_c_fc_weight", "mlp_c_fc_bias", "mlp_c_proj_weight", "mlp_c_proj_bias"]
# C field -> open_clip state_dict suffix under transformer.resblocks.{i}.
BLOCK_KEYS = {"ln_1_weight": "ln_1.weight", "ln_1_bias": "ln_1.bias",
              "attn_in_proj_weight": "attn.in_proj_weight", "attn_in_proj_bias": "attn.in_proj_bias",
              "attn_out_proj_weight": "attn.out_proj.weight", "attn_out_proj_bias": "attn.out_proj.bias",
              "ln_2_weight": "ln_2.weight", "ln_2_bias": "ln_2.bias",
              "mlp_c_fc_weight": "mlp.c_fc.weight", "mlp_c_fc_bias": "mlp.c_fc.bias",
              "mlp_c_proj_weight": "mlp.c_proj.weight", "mlp_c_proj_bias": "mlp.c_proj.bias"}


class BlockWeightsC(C.Structure):
    _fields_ = [(f, C.c_void_p) for f in BLOCK_FIELDS]


TOP_FIELDS = ["class_embedding", "positional_embedding", "proj", "conv1_weight", "ln_pre_weight",
              "ln_pre_bias", "ln_post_weight", "ln_post_bias"]
TOP_KEYS = {"class_embedding": "class_embedding", "positional_embedding": "positional_embedding",
            "proj": "proj", "conv1_weight": "conv1.weight", "ln_pre_weight": "ln_pre.weight",
            "ln_pre_bias": "ln_pre.bias", "ln_post_weight": "ln_post.weight",
            "ln_post_bias": "ln_post.bias"}


class VitWeightsC(C.Structure):
    _fields_ = [(f, C.c_void_p) for f in TOP_FIELDS] + [("blocks_host", C.POINTER(BlockWeightsC))]


class LossSpecC(C.Structure):
    _fields_ = [("loss_kind", C.c_int32), ("reduction", C.c_int32), ("output_normalize", C.c_int32),
                ("n_classes", C.c_int32), ("logit_scale", C.c_float), ("ref", C.c_void_p),
                ("targets", C.c_void_p), ("y_target", C.c_void_p)]


class ProfileEntryC(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("total_ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double), ("launches", C.c_int64)]


class RvlmError(RuntimeError):
    pass


_lib = None

_SIGS = {
    # name: (restype, argtypes)
    "rvlm_version": (C.c_int, []),
    "rvlm_last_error": (C.c_char_p, []),
    "rvlm_vit_create": (C.c_int, [C.POINTER(VitConfigC), C.POINTER(VitWeightsC), c_stream,
                                  C.POINTER(C.c_void_p)]),
    "rvlm_vit_destroy": (C.c_int, [C.c_void_p]),
    "rvlm_vit_load_weights": (C.c_int, [C.c_void_p, C.POINTER(VitWeightsC), c_stream]),
    "rvlm_vit_workspace_bytes": (C.c_size_t, [C.c_void_p]),
    "rvlm_vit_forward": (C.c_int, [C.c_void_p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, c_f32p,
                                   c_stream]),
    "rvlm_vit_backward_input": (C.c_int, [C.c_void_p, c_f32p, C.c_int, c_f32p, c_stream]),
    "rvlm_vit_backward_params": (C.c_int, [C.c_void_p, c_f32p, C.c_int, C.POINTER(VitWeightsC), C.c_int, c_stream]),
    "rvlm_vit_backward_params_stages": (C.c_int, [C.c_void_p, c_f32p, C.c_int, C.POINTER(VitWeightsC), C.c_int, C.c_int,
                                                  C.c_int, c_stream]),
    "rvlm_vit_fwd_inputgrad": (C.c_int, [C.c_void_p, c_f32p, c_f32p, C.c_int, C.POINTER(LossSpecC), c_f32p, c_f32p,
                                         c_f32p, c_f32p, c_stream]),
    "rvlm_ce_logits": (C.c_int, [c_f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, c_f32p, C.c_void_p,
                                 c_stream]),
    "rvlm_head_logits": (C.c_int, [c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_float, c_f32p, c_stream]),
    "rvlm_head_logits_bwd": (C.c_int, [c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_float, c_f32p, c_stream]),
    "rvlm_cosine_rows": (C.c_int, [c_f32p, c_f32p, C.c_int, C.c_int, c_f32p, c_f32p, c_stream]),
    "rvlm_l2_normalize_rows": (C.c_int, [c_f32p, C.c_int, C.c_int, c_f32p, c_f32p, c_stream]),
    "rvlm_adamw_step": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_size_t, C.c_double, C.c_double, C.c_double,
                                  C.c_double, C.c_double, C.c_int, C.c_float, c_stream]),
    "rvlm_loss_grad": (C.c_int, [C.c_int, C.c_int, c_f32p, c_f32p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                 C.c_int, C.c_float, c_f32p, c_f32p, c_f32p, C.c_void_p, c_f32p,
                                 c_stream]),
    "rvlm_argmax_eq": (C.c_int, [c_f32p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, c_stream]),
    "rvlm_check_image_range": (C.c_int, [c_f32p, C.c_size_t, C.c_void_p, c_stream]),
    "rvlm_pgd_linf_update": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_size_t, C.c_float,
                                       C.c_float, C.c_float, C.c_int, c_f32p, C.c_void_p, c_stream]),
    "rvlm_project_perturbation": (C.c_int, [c_f32p, C.c_size_t, C.c_int, C.c_int, C.c_float, c_f32p, c_stream]),
    "rvlm_normalize_grad": (C.c_int, [c_f32p, C.c_size_t, C.c_int, C.c_int, c_f32p, c_stream]),
    "rvlm_pgd_l2_update": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_size_t, C.c_int, C.c_float, C.c_float,
                                     C.c_float, C.c_int, c_f32p, C.c_void_p, c_stream]),
    "rvlm_pgd_run_norm": (C.c_int, [C.c_void_p, c_f32p, c_f32p, C.c_int, C.POINTER(LossSpecC), C.c_int, C.c_float,
                                    C.c_int, C.c_float, C.c_float, C.c_int, c_f32p, c_f32p, C.c_void_p, c_stream]),
    "rvlm_apgd_linf_step": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_float, C.c_float,
                                      C.c_size_t, C.c_int, c_stream]),
    "rvlm_apgd_l2_step": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_float, C.c_float, C.c_size_t, C.c_int,
                                    c_stream]),
    "rvlm_apgd_run_norm": (C.c_int, [C.c_void_p, c_f32p, c_f32p, C.c_int, C.POINTER(LossSpecC), C.c_int, C.c_float,
                                     C.c_int, C.c_float, C.c_int, C.c_int, c_f32p, c_f32p, c_f32p, C.c_void_p, c_stream]),
    "rvlm_apgd_controller": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p,
                                       C.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, c_stream]),
    "rvlm_apgd_controller_rho": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, c_f32p,
                                           C.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, c_stream]),
    "rvlm_vit_set_apgd_rho": (C.c_int, [C.c_void_p, C.c_double]),
    "rvlm_apgd_select": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_size_t, C.c_int, c_stream]),
    "rvlm_linf_random_start": (C.c_int, [c_f32p, c_f32p, C.c_float, C.c_size_t, C.c_int, c_f32p,
                                         c_stream]),
    "rvlm_l2_random_start": (C.c_int, [c_f32p, c_f32p, C.c_float, C.c_size_t, C.c_int, c_f32p, c_stream]),
    "rvlm_pgd_run": (C.c_int, [C.c_void_p, c_f32p, c_f32p, C.c_int, C.POINTER(LossSpecC), C.c_float,
                               C.c_int, C.c_float, C.c_float, C.c_int, c_f32p, c_f32p, C.c_void_p,
                               c_stream]),
    "rvlm_pgd_run_mixed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, c_f32p, c_f32p, C.c_int, C.POINTER(LossSpecC), C.c_int,
                                     C.c_float, C.c_int, C.c_float, C.c_float, C.c_int, c_f32p, c_f32p, C.c_void_p,
                                     c_stream]),
    "rvlm_pgd_run_mixed_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, c_f32p, c_f32p, C.c_int, C.POINTER(LossSpecC), C.c_int,
                                         C.c_float, C.c_int, C.c_float, C.c_float, C.c_int, c_f32p, c_f32p, C.c_void_p,
                                         c_stream]),
    "rvlm_vit_backward_input_from": (C.c_int, [C.c_void_p, C.c_void_p, c_f32p, C.c_int, c_f32p, c_stream]),
    "rvlm_vit_set_flash_inference": (C.c_int, [C.c_void_p, C.c_int]),
    "rvlm_vit_forward_for": (C.c_int, [C.c_void_p, C.c_void_p, c_f32p, c_f32p, C.c_int, C.c_int, c_f32p, c_stream]),
    "rvlm_apgd_run": (C.c_int, [C.c_void_p, c_f32p, c_f32p, C.c_int, C.POINTER(LossSpecC), C.c_float,
                                C.c_int, C.c_float, C.c_int, C.c_int, c_f32p, c_f32p, c_f32p,
                                C.c_void_p, c_stream]),
    "rvlm_preproc_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "rvlm_preproc_destroy": (C.c_int, [C.c_void_p]),
    "rvlm_preproc_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, c_f32p, c_stream]),
    "rvlm_preproc_run_batch": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int,
                                         c_f32p, c_stream]),
    "rvlm_vit_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "rvlm_vit_get_profile": (C.c_int, [C.c_void_p, C.POINTER(ProfileEntryC), C.POINTER(C.c_int)]),
    "rvlm_vit_reset_profile": (C.c_int, [C.c_void_p]),
    "rvlm_square_linf_propose": (C.c_int, [c_f32p, c_f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_float, c_f32p, c_f32p, c_stream]),
    "rvlm_square_accept": (C.c_int, [c_f32p, c_f32p, C.c_void_p, c_f32p, C.c_int, C.c_size_t, c_stream]),
    # kernel-level test surface (include/rvlm_kernels.h)
    "rvlm_k_gemm_bf16_nt": (C.c_int, [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int, c_f32p, C.c_void_p, C.c_long,
                                      C.c_void_p, C.c_void_p, c_f32p, C.c_int, c_stream]),
    "rvlm_k_gemm_f32": (C.c_int, [c_f32p, C.c_long, C.c_long, c_f32p, C.c_long, C.c_long, c_f32p,
                                  C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, C.c_float, c_f32p,
                                  c_stream]),
    "rvlm_k_gemm_f32_ex": (C.c_int, [c_f32p, C.c_long, C.c_long, C.c_long, c_f32p, C.c_long, C.c_long, C.c_long, c_f32p,
                                     C.c_long, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, c_f32p,
                                     C.c_int, c_f32p, c_f32p, c_f32p, c_stream]),
    "rvlm_k_gemm_f32_set_valu": (C.c_int, [C.c_int]),
    "rvlm_k_softmax_rows": (C.c_int, [c_f32p, c_f32p, C.c_long, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "rvlm_k_attn_fwd_f32_flash": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "rvlm_k_attn_bwd_f32_flash": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "rvlm_k_attn_fwd_bf16": (C.c_int, [C.c_void_p, C.c_void_p, c_f32p, C.c_int, C.c_int, C.c_int,
                                       c_stream]),
    "rvlm_k_attn_bwd_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_f32p, c_f32p,
                                       C.c_void_p, C.c_int, C.c_int, C.c_int, c_stream]),
    "rvlm_k_attn_set_use_tr": (C.c_int, [C.c_int]),
    "rvlm_k_gemm_set_variant": (C.c_int, [C.c_int]),
    "rvlm_k_gemm_last_kernels": (C.c_int, []),
    "rvlm_k_gemm_set_trace": (C.c_int, [C.c_void_p]),
    "rvlm_k_gemm_set_pingpong": (C.c_int, [C.c_int, C.c_int]),
    "rvlm_k_gemm_set_m16": (C.c_int, [C.c_int]),
    "rvlm_k_gemm_x_set_trace": (C.c_int, [C.c_void_p]),
    "rvlm_k_gemm_set_ablate": (C.c_int, [C.c_int]),
    "rvlm_k_probe_operand_stream": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_void_p, c_stream]),
    "rvlm_k_attn_occupancy": (C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    "rvlm_k_layernorm_fwd_f32": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int,
                                           C.c_int, c_stream]),
    "rvlm_k_layernorm_bwd_f32": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int,
                                           C.c_int, C.c_int, c_stream]),
    "rvlm_k_layernorm_bwd_bf16": (C.c_int, [C.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                            C.c_void_p]),
    "rvlm_k_probe_tr16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_stream]),
    "rvlm_k_wgrad_work_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "rvlm_k_wgrad_set_transposed": (None, [C.c_int]),
    "rvlm_k_wgrad_bf16": (C.c_int, [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, c_f32p,
                                    C.c_long, C.c_int, c_f32p, C.c_void_p, C.c_size_t, c_stream]),
}

_SIGS.update({
    # data-parallel trainer without torch.distributed: RCCL behind the C ABI (include/rvlm.h, csrc/comm.hip)
    "rvlm_comm_unique_id": (C.c_int, [C.c_void_p]),
    "rvlm_comm_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "rvlm_comm_destroy": (C.c_int, [C.c_void_p]),
    "rvlm_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rvlm_allreduce_grads": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, c_stream]),
})
COMM_ID_BYTES, DTYPE_F32, DTYPE_BF16 = 128, 0, 1
EXPORTED_SYMBOLS = tuple(_SIGS)


def load():
    """Load librvlm.so (raises if it has not been built - no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RvlmError(
                f"{LIB_PATH} not found: build the HIP library first "
                f"(python -c 'import __graft_entry__ as g; g.build()' or make -C robustvlm_amd/csrc)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int, what: str = ""):
    """Map an rvlm_status to the Python exception the reference would raise."""
    if rc == RVLM_OK:
        return
    msg = load().rvlm_last_error().decode(errors="replace")
    full = f"{what}: {msg}" if what else msg
    if rc == RVLM_ERR_ARG:
        raise ValueError(full)
    if rc == RVLM_ERR_UNSUPPORTED:
        raise NotImplementedError(full)
    raise RvlmError(f"[rvlm status {rc}] {full}")


def ptr(t):
    """Device pointer of a tensor (or None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
