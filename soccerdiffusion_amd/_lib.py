"""ctypes binding of include/soccerdiffusion_hip.h.

The product path has NO CPU fallback: if the HIP library is missing or a call fails,
this module raises.  (The CPU oracle lives in oracle/ and is test infrastructure.)
"""

from __future__ import annotations

import ctypes as C
import os

# torch MUST be imported before the HIP library is dlopen'ed: the torch wheel bundles its own
# libamdhip64.so.7 / libhsa-runtime64.so.1, and the dynamic loader binds our NEEDED entry to
# whichever copy with that SONAME is already mapped.  Loading ours first would map the
# system ROCm runtime instead and leave the process with a runtime torch did not initialise
# (launches then fail with hipErrorNoDevice).  One process, one HIP runtime: torch's.
import torch  # noqa: F401

_PKG = os.path.dirname(os.path.abspath(__file__))
# SD_HIP_LIB: developer hook to load another build of the same library (kernel A/B runs, tools/ab_build.sh)
LIB_PATH = os.environ.get("SD_HIP_LIB") or os.path.join(_PKG, "lib", "libsoccerdiffusion_hip.so")

c_float_p = C.POINTER(C.c_float)


class LayerWeights(C.Structure):
    """sd_layer_weights (field order = header order)."""

    FIELDS = (
        "sa_in_w", "sa_in_b", "sa_out_w", "sa_out_b",
        "ca_in_w", "ca_in_b", "ca_out_w", "ca_out_b",
        "lin1_w", "lin1_b", "lin2_w", "lin2_b",
        "n1_w", "n1_b", "n2_w", "n2_b", "n3_w", "n3_b",
    )
    _fields_ = [(n, C.c_void_p) for n in FIELDS]


class DenoiserWeights(C.Structure):
    _fields_ = [
        ("d", C.c_int32), ("J", C.c_int32), ("L", C.c_int32), ("heads", C.c_int32),
        ("emb_w", C.c_void_p), ("emb_b", C.c_void_p), ("out_w", C.c_void_p), ("out_b", C.c_void_p),
        ("pe", C.c_void_p), ("T_max", C.c_int32), ("_pad", C.c_int32),
        ("layers", C.POINTER(LayerWeights)),
    ]


class EncoderWeights(C.Structure):
    _fields_ = [
        ("d", C.c_int32), ("C", C.c_int32), ("p", C.c_int32), ("L", C.c_int32),
        ("heads", C.c_int32), ("S_max", C.c_int32),
        ("emb_w", C.c_void_p), ("emb_b", C.c_void_p), ("pe", C.c_void_p),
        ("layers", C.POINTER(LayerWeights)),
    ]


class TrainFwdChainArgs(C.Structure):
    """sd_train_fwd_chain_args (field order = header order)."""

    _fields_ = ([("R", C.c_int64), ("d", C.c_int32), ("n_next", C.c_int32)]
                + [(n, C.c_void_p) for n in ("a", "wo", "bo", "h_in", "h_out", "ln_w", "ln_b", "n_out", "w1", "b1", "pre", "u", "w2", "b2",
                                             "h2_out", "nln_w", "nln_b", "nn_out", "wn", "bn", "y_out")]
                + [("p", C.c_float), ("seed", C.c_uint64), ("site_out", C.c_uint64), ("site_act", C.c_uint64), ("site_ffn", C.c_uint64)]
                + [(n, C.c_void_p) for n in ("amax_a", "amax_n", "amax_u", "amax_nn", "amax_h2")])


class TrainLayerFwdArgs(C.Structure):
    """sd_train_layer_fwd_args (field order = header order)."""

    _fields_ = ([(n, C.c_int32) for n in ("B", "T", "M", "d", "heads")]
                + [(n, C.c_void_p) for n in ("h", "qkv", "a_sa", "lse_sa", "h1", "n2", "q", "kv", "a_ca", "lse_ca", "h2", "nf", "pre", "u", "h3", "nn1",
                                             "qkv2", "w_o", "w_q", "w_oc", "w_1", "w_2", "w_n", "b_o", "b_q", "b_oc", "b_1", "b_2", "b_n", "n2_w", "n2_b",
                                             "n3_w", "n3_b", "nn_w", "nn_b")]
                + [("p", C.c_float)] + [(n, C.c_uint64) for n in ("seed", "site_sa_probs", "site_sa_out", "site_ca_probs", "site_ca_out", "site_act",
                                                                  "site_ffn")]
                + [(n, C.c_void_p) for n in ("amax_a_sa", "amax_n2", "amax_a_ca", "amax_nf", "amax_u", "amax_nn", "amax_out")])


class TrainBwdChainArgs(C.Structure):
    """sd_train_bwd_chain_args (field order = header order)."""

    _fields_ = ([("R", C.c_int64), ("d", C.c_int32), ("passes", C.c_int32), ("ldy", C.c_int32)]
                + [(n, C.c_void_p) for n in ("dy", "dym", "wt", "pre", "dpre", "wt1", "x", "ln_w", "dres", "dg", "db", "dx")]
                + [("p", C.c_float), ("seed", C.c_uint64), ("site_in", C.c_uint64), ("site_act", C.c_uint64)]
                + [(n, C.c_void_p) for n in ("amax_dy", "amax_dpre", "amax_dx")])


class GemmTnProblem(C.Structure):
    """sd_gemm_tn_problem (field order = header order)."""

    _fields_ = ([(n, C.c_void_p) for n in ("dY", "X", "dW", "db", "amax_dy", "amax_x")]
                + [("R", C.c_int64), ("N", C.c_int32), ("K", C.c_int32), ("ldy", C.c_int32), ("ldx", C.c_int32), ("ldw", C.c_int32)])


# name -> (restype, argtypes); mirrors the header one to one (tests check the export list)
SIGNATURES = {
    "sd_abi_version": (C.c_int, []),
    "sd_last_error": (C.c_char_p, []),
    "sd_workspace_floats": (C.c_size_t, [C.c_int] * 6),
    "sd_sampler_mode": (C.c_int, [C.c_int] * 5),
    "sd_step_token": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p]),
    "sd_denoiser_forward": (C.c_int, [C.POINTER(DenoiserWeights), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "sd_encoder_forward": (C.c_int, [C.POINTER(EncoderWeights), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "sd_game_state_embed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "sd_ddim_add_noise": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "sd_ddim_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_long, C.c_void_p]),
    "sd_normalize": (C.c_int, [C.c_void_p] * 4 + [C.c_long, C.c_int, C.c_int, C.c_void_p]),
    "sd_ddim_sample": (C.c_int, [C.POINTER(DenoiserWeights), C.c_void_p, C.c_void_p, c_float_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "sd_ddim_sample_ex": (C.c_int, [C.POINTER(DenoiserWeights), C.c_void_p, C.c_void_p, c_float_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "sd_ddim_sample_eps": (C.c_int, [C.POINTER(DenoiserWeights), C.c_void_p, C.c_void_p, c_float_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "sd_bn_scratch_floats": (C.c_size_t, [C.c_int64, C.c_int]),
    "sd_bn_train_fwd": (C.c_int, [C.c_void_p] * 12 + [C.c_int64, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p]),
    "sd_convt3x3_s2": (C.c_int, [C.c_void_p] * 9 + [C.c_int] * 5 + [C.c_void_p]),
    "sd_convt1x1_s2": (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 5 + [C.c_void_p]),
    "sd_bn_relu_pool_fwd": (C.c_int, [C.c_void_p] * 12 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    "sd_bn_relu_pool_bwd": (C.c_int, [C.c_void_p] * 12 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "sd_bn_train_bwd": (C.c_int, [C.c_void_p] * 14 + [C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "sd_conv_wgrad": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 7 + [C.c_void_p]),
    "sd_conv_wgrad_scratch_floats": (C.c_size_t, [C.c_int] * 7),
    "sd_stem_conv_raw": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 3 + [C.c_void_p]),
    "sd_stem_wgrad_scratch_floats": (C.c_size_t, [C.c_int] * 3),
    "sd_stem_wgrad": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 3 + [C.c_void_p]),
    "sd_sampler_prepare": (C.c_int, [C.POINTER(DenoiserWeights), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "sd_sampler_eps": (C.c_int, [C.POINTER(DenoiserWeights), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "sd_op_linear": (C.c_int, [C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "sd_op_attention": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "sd_op_patch_embed": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 5 + [C.c_void_p]),
    "sd_op_fc_out": (C.c_int, [C.c_void_p] * 5 + [c_float_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "sd_op_linear_strided": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_void_p]),
    "sd_op_attention_lse": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]),
    "sd_op_attention_bwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_int] * 5 + [C.c_void_p]),
    "sd_op_dropout": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_float, C.c_uint64, C.c_uint64, C.c_void_p]),
    "sd_op_dropout_mask": (C.c_int, [C.c_void_p, C.c_long, C.c_int, C.c_float, C.c_uint64, C.c_uint64, C.c_void_p]),
    "sd_op_gelu_dropout_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_float, C.c_uint64, C.c_uint64, C.c_void_p]),
    "sd_op_gelu_dropout_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_float, C.c_uint64, C.c_uint64, C.c_void_p]),
    "sd_op_linear_dropout": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                       C.c_float, C.c_uint64, C.c_uint64, C.c_void_p]),
    "sd_train_fwd_chain": (C.c_int, [C.POINTER(TrainFwdChainArgs), C.c_void_p]),
    "sd_train_bwd_chain": (C.c_int, [C.POINTER(TrainBwdChainArgs), C.c_void_p]),
    "sd_gemm_tn_grouped": (C.c_int, [C.POINTER(GemmTnProblem), C.c_int, C.c_void_p]),
    "sd_gemm_tn_grouped_plan": (C.c_long, [C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_long), C.c_int, C.POINTER(C.c_long)]),
    "sd_op_absmax": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "sd_pack_weight_blocks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "sd_op_linear_packed": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_uint64, C.c_void_p]),
    "sd_op_attention_lse_dropout": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 5 +
                                    [C.c_float, C.c_uint64, C.c_uint64, C.c_void_p]),
    "sd_op_attention_bwd_dropout": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                              C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_int] * 5 +
                                    [C.c_float, C.c_uint64, C.c_uint64, C.c_void_p]),
    "sd_op_gemm_tn": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p]),
    "sd_op_layernorm_fwd": (C.c_int, [C.c_void_p] * 6 + [C.c_long, C.c_int, C.c_void_p]),
    "sd_op_layernorm_bwd": (C.c_int, [C.c_void_p] * 9 + [C.c_long, C.c_int, C.c_void_p]),
    "sd_op_gelu_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]),
    "sd_op_gelu_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]),
    "sd_op_colsum": (C.c_int, [C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_void_p, C.c_void_p]),
    "sd_op_small_k_matmul": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p]),
    "sd_mse_loss": (C.c_int, [C.c_void_p] * 5 + [C.c_long, C.c_void_p]),
    "sd_adamw_step": (C.c_int, [C.c_void_p] * 4 + [C.c_long] + [C.c_double] * 5 + [C.c_long, C.c_void_p]),
    "sd_adamw_step_dev": (C.c_int, [C.c_void_p] * 4 + [C.c_long, C.c_void_p, C.c_void_p]),
    "sd_adamw_hyper": (C.c_int, [C.c_double] * 5 + [C.c_long, c_float_p]),
    "sd_set_dropout_epoch": (C.c_int, [C.c_void_p]),
    "sd_train_layer_fwd_ok": (C.c_int, [C.c_int] * 4),
    "sd_train_layer_fwd": (C.c_int, [C.POINTER(TrainLayerFwdArgs), C.c_void_p]),
    "sd_train_head_fwd": (C.c_int, [C.c_void_p] * 12 + [C.c_int] * 3 + [C.c_void_p]),
    "sd_pack_weight_traj_halfs": (C.c_size_t, [C.c_int, C.c_int]),
    "sd_pack_weight_traj": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "sd_pack_weight_traj_multi": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "sd_conv3x3_packed_halfs": (C.c_size_t, [C.c_int, C.c_int]),
    "sd_conv3x3_pack": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sd_conv3x3_bn_act": (C.c_int, [C.c_void_p] * 9 + [C.c_int] * 6 + [C.c_void_p]),
    "sd_conv1x1_bn_act": (C.c_int, [C.c_void_p] * 9 + [C.c_int] * 6 + [C.c_void_p]),
    "sd_absmax_word": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "sd_conv_packed_halfs": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "sd_conv_pack": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sd_conv_s2_bn_act": (C.c_int, [C.c_void_p] * 8 + [C.c_int] * 7 + [C.c_void_p]),
    "sd_stem_packed_halfs": (C.c_size_t, []),
    "sd_stem_pack": (C.c_int, [C.c_void_p] * 5),
    "sd_stem_conv_bn_relu_pool": (C.c_int, [C.c_void_p] * 8 + [C.c_int] * 3 + [C.c_void_p]),
    "sd_profile_enable": (C.c_int, [C.c_int]),
    "sd_profile_collect": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_long), C.c_int]),
}

KERNEL_CLASSES = ("panel_gemm_kernel", "attention_kernel", "patch_embed_kernel", "fc_out_kernel", "decoder_layer_kernel", "decoder_head_kernel",
                  "traj_step_kernel")

_lib = None


def load() -> C.CDLL:
    """Loads the HIP library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m soccerdiffusion_amd.build` "
            "(hipcc, gfx950).  soccerdiffusion_amd has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.sd_abi_version() != 1:
        raise RuntimeError("libsoccerdiffusion_hip.so ABI version mismatch: rebuild")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().sd_last_error().decode()
        kind = "invalid argument" if rc < 0 else "HIP error"
        raise RuntimeError(f"{what}: {kind} {rc}: {msg}")
