"""ctypes binding of libsnuffy_hip.so (the C ABI of include/snuffy_hip.h).

There is NO fallback: if the library is missing or a call fails, this raises.  (The CPU oracle under oracle/ is test
infrastructure and is never imported from here.)
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# SNUFFY_HIP_LIB: load another build of the same library (A/B timing of kernel variants, tools/ab.sh)
LIB_PATH = os.environ.get("SNUFFY_HIP_LIB") or os.path.join(_HERE, "lib", "libsnuffy_hip.so")

SNF_OK = 0
SNF_EINVAL = -1
SNF_ELAUNCH = -2
SNF_EUNSUPPORTED = -3
SNF_EWORKSPACE = -4

ACT_CODES = {"relu": 0, "gelu": 1, "leakyrelu": 2, "selu": 3, "none": 4}
DT_F32 = 0
DT_BF16 = 1
DT_BF16_SPLIT3 = 2
DT_BF16_HL = 3
TOPK_MAX_K = 2048

# name -> (restype, argtypes); mirrors include/snuffy_hip.h one to one
SIGNATURES = {
    "snf_version": (c_char_p, []),
    "snf_last_error": (c_char_p, []),
    "snf_device_cu_count": (c_int, []),
    "snf_critic_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                               c_void_p]),
    "snf_critic_ln_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_float, c_void_p,
                                  c_void_p]),
    "snf_topk_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "snf_topk_f32": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "snf_topk_gather_f32": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                    c_size_t, c_void_p]),
    "snf_gather_rows_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "snf_selector_state_bytes": (c_size_t, []),
    "snf_critic_select_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                      c_void_p]),
    "snf_topk_select_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "snf_topk_hist_select_f32": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "snf_scatter_rows_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "snf_scatter_add_rows_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "snf_slot_map_i32": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_void_p]),
    "snf_gather_slot_map_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "snf_layernorm_rows_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "snf_layernorm_bwd_blocks": (c_int, [c_int64]),
    "snf_layernorm_rows_bwd_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_int64, c_void_p, c_float, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_void_p]),
    "snf_fold_blocks": (c_int, [c_int]),
    "snf_fold_linear_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                    c_void_p]),
    "snf_unfold_linear_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p]),
    "snf_split3_f32": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "snf_mil_loss_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "snf_mil_loss_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "snf_split3_weight_f32": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "snf_split3_colsum_f32": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "snf_colsum_blocks": (c_int, [c_int64]),
    "snf_colsum_fused": (c_int, [c_void_p, c_int, c_int64, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "snf_layernorm_rows_split3_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                              c_void_p, c_void_p]),
    "snf_linear_rows_x3_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_void_p, c_int64, c_int,
                                       c_void_p]),
    "snf_critic_ln_hl_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_float,
                                     c_void_p, c_void_p, c_void_p]),
    "snf_bias_act": (c_int, [c_void_p, c_int, c_int64, c_int, c_void_p, c_int, c_void_p]),
    "snf_ln_mean_head_workspace_bytes": (c_size_t, [c_int]),
    "snf_ln_mean_head_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                     c_void_p, c_size_t, c_void_p]),
    "snf_sparse_attn_fwd_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int, c_int]),
    "snf_sparse_attn_fwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_float, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "snf_sparse_attn_fwd_x3u_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_float, c_void_p,
                                            c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "snf_sparse_attn_bwd_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int]),
    "snf_sparse_attn_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                        c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "snf_sparse_attn_bwd_ld_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                           c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "snf_sparse_attn_bwd_dropout_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_float, c_uint64, c_uint64, c_void_p,
                                                c_int64, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "snf_sparse_attn_bwd_mfma": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_int64, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "snf_sparse_attn_dkp_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t,
                                        c_void_p]),
    "snf_sparse_attn_fwd_mfma": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int, c_int64, c_int, c_int,
                                         c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "snf_sparse_attn_fwd_x3_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int]),
    "snf_sparse_attn_fwd_x3": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_float,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "snf_sparse_attn_fwd_x3_dropout": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_float,
                                               c_float, c_uint64, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "snf_sparse_attn_fwd_x3_hl_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int]),
    "snf_sparse_attn_x3_hl_kpfrag_bytes": (c_size_t, [c_int, c_int, c_int]),
    "snf_linear_rows_x3_kpfrag_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                              c_void_p, c_size_t, c_void_p]),
    "snf_gather_linear_rows_x3_kpfrag_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_int,
                                                     c_int, c_float, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "snf_sparse_attn_fwd_x3_hl_kpfrag": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int,
                                                 c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "snf_sparse_attn_fwd_x3_hl": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_float,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "snf_sparse_attn_fwd_mfma_dropout": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int, c_int64, c_int, c_int,
                                                 c_int, c_float, c_void_p, c_void_p, c_void_p, c_float, c_uint64, c_uint64,
                                                 c_void_p, c_size_t, c_void_p]),
    "snf_sparse_attn_bwd_mfma_dropout": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                                 c_float, c_uint64, c_uint64, c_int64, c_int, c_int, c_int, c_float, c_void_p,
                                                 c_void_p, c_void_p, c_int, c_void_p]),
    "snf_sparse_attn_bwd_mfma_ex": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_float, c_uint64, c_uint64, c_int64, c_int, c_int, c_int, c_float, c_void_p,
                                            c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p]),
    "snf_dropout_mask_f32": (c_int, [c_float, c_uint64, c_uint64, c_int, c_int64, c_int, c_void_p, c_void_p]),
    "snf_tile_preprocess_u8": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                       c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "snf_gemm_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_int64,
                              c_int, c_int, c_void_p]),
    "snf_gemm_bf16_resid_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int,
                                        c_void_p, c_int64, c_int, c_void_p]),
    "snf_linear_rows_x3_resid_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                             c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "snf_gemm_bf16_lnfold": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                     c_void_p, c_int64, c_void_p]),
    "snf_gemm_bf16_resid": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_void_p, c_int64, c_void_p,
                                    c_int64, c_void_p, c_void_p]),
    "snf_vit_row_stats": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int64, c_int, c_float, c_void_p, c_void_p, c_int64, c_void_p]),
    "snf_split_hl_f32": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "snf_layernorm_rows_hl_patch_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p,
                                                c_void_p]),
    "snf_layernorm_rows_hl_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                          c_void_p, c_void_p]),
    "snf_gemm_hl_resid_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int,
                                       c_void_p, c_int64, c_int, c_void_p]),
    "snf_gemm_hl_ws_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "snf_gemm_hl_gated_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_int64,
                                       c_void_p]),
    "snf_gemm_tn_ws_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "snf_gemm_tn_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_int64, c_int, c_int, c_int, c_int64, c_int, c_int, c_void_p,
                                c_int64, c_void_p, c_size_t, c_void_p]),
    "snf_split_hl_colsum_f32": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    "snf_gemm_hl_ws_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int,
                                    c_void_p, c_int64, c_int, c_void_p, c_size_t, c_void_p]),
    "snf_gemm_hl_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_int64,
                                 c_int, c_void_p]),
    "snf_topk_segmented_f32": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p]),
    "snf_sparse_attn_varlen_plan": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "snf_sparse_attn_fwd_mfma_varlen": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int, c_int, c_int,
                                                c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                                c_void_p]),
    "snf_sparse_attn_x3_varlen_plan": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "snf_sparse_attn_fwd_x3_varlen": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                              c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "snf_sparse_attn_fwd_ragged_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int64, c_int, c_int,
                                               c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "snf_ln_mean_head_varlen_plan": (c_int, [c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "snf_ln_mean_head_varlen_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_size_t, c_void_p]),
    "snf_vit_patchify": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "snf_vit_assemble_tokens": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "snf_vit_residual_ln": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_float,
                                    c_void_p, c_void_p, c_void_p]),
    "snf_vit_attention_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "snf_sampler_advance": (c_int, [c_void_p, c_void_p]),
    "snf_random_share_keys_f32": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int, c_void_p, c_void_p]),
    "snf_debug_attn_trace": (None, [c_void_p]),
    "snf_debug_attn_trace_wg": (None, [c_int]),
    "snf_debug_x3p_kbw": (None, [c_int]),
    "snf_debug_exact_attn_mfma": (None, [c_int]),
    "snf_vit_attention_mfma": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "snf_vit_attention_x3_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p]),
}

_lib = None


class SnuffyHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once). Raises SnuffyHipError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SnuffyHipError(
            "libsnuffy_hip.so not found at %s -- build it with `python -m snuffy_amd.build` "
            "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
    # torch owns device memory and streams, so the kernels must run on the SAME HIP runtime instance torch uses:
    # import torch first so that its bundled libamdhip64 is the one already mapped when our NEEDED entry resolves.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == header / library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != SNF_OK:
        msg = load().snf_last_error()
        raise SnuffyHipError("%s failed (code %d): %s" % (what, rc, msg.decode() if msg else "?"))


def version():
    return load().snf_version().decode()
