"""ctypes binding of csrc/libmi355rec.so (the C ABI declared in include/mi355rec.h).

The HIP library is the product: there is NO fallback.  If the shared object is missing or a symbol the
header declares cannot be resolved, importing this module raises -- build it with
``python -c "import __graft_entry__ as g; g.build()"`` (or ``csrc/build.sh``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmi355rec.so")

p, i32, i64, f32, sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t

# symbol -> (restype, argtypes); must list every function of include/mi355rec.h
SIGNATURES = {
    "rec_version": (i32, []),
    "rec_index_pack_i64": (i32, [p, i32, i64, p, i64, i64, p]),
    "rec_emb_gather_f32": (i32, [p, i64, i32, i64, p, i64, p, p, p]),
    "rec_emb_fm_fwd_f32": (i32, [p, i64, p, i64, p, i64, i32, p, i64, i32, p, p, p, p, p, p]),
    "rec_emb_fm_bwd_vals_f32": (i32, [p, i64, i64, i32, p, i64, i32, p, p, p, p, p, p]),
    "rec_dedup_workspace_bytes": (sz, [i64]),
    "rec_dedup_plan_i64": (i32, [p, i64, i64, p, p, p, p, p, sz, p]),
    "rec_dedup_plan_sorted_lists_i64": (i32, [p, i64, p, i32, i64, p, p, p, p, p, sz, p]),
    "rec_segment_sum_workspace_bytes": (sz, [i64, i32]),
    "rec_segment_sum_f32": (i32, [p, i32, p, p, i64, i32, p, p, p]),
    "rec_gemm_f32": (i32, [i32, i32, i64, i64, i64, p, i64, p, i64, p, i64, i32, p, p, i64, p, i64, i32, p, p, p]),
    "rec_act_fwd_f32": (i32, [i32, p, p, p, i64, p]),
    "rec_crossnet_mat_bwd_elem_f32": (i32, [p, p, p, p, p, i32, i64, p]),
    "rec_act_bwd_f32": (i32, [i32, p, p, p, i64, p]),
    "rec_colsum_workspace_bytes": (sz, [i64, i64]),
    "rec_colsum_fused_f32": (i32, [p, i64, i64, i64, p, p, p, p]),
    "rec_colsum_f32": (i32, [p, i64, i64, i64, p, p, p]),
    "rec_axpby_f32": (i32, [f32, p, f32, p, i64, p]),
    "rec_copy_cols_f32": (i32, [p, i64, p, i64, i64, i64, p]),
    "rec_crossnet_vec_fwd_f32": (i32, [p, i64, i32, i32, p, p, p, p, p]),
    "rec_crossnet_vec_bwd_workspace_bytes": (sz, [i64, i32, i32]),
    "rec_crossnet_vec_bwd_f32": (i32, [p, i64, i32, i32, p, p, p, p, p, p, p, p]),
    "rec_cosine_fwd_f32": (i32, [p, p, i64, i32, p, p]),
    "rec_cosine_bwd_f32": (i32, [p, p, i64, i32, p, p, p, p]),
    "rec_bce_fwd_bwd_f32": (i32, [p, p, i64, p, p, p, p]),
    "rec_adam_lr_t_f32": (f32, [f32, f32, f32, i64]),
    "rec_adam_advance_f32": (i32, [p, p, i64, p, p]),
    "rec_adam_dense_multi_f32": (i32, [i32, p, p, p, p, p, p, f32, f32, f32, p]),
    "rec_deepfm_fused_post_direct_adam_dev_f32": (i32, [i32, i64] + [p] * 19 + [p, i64, i64, p, p, p, p, i64, i64, p, f32,
                                                                           f32, f32, p, p, p]),
    "rec_adam_keras_catchup_f32": (i32, [p, p, i64, i32, p, i64, i64, p, p, i64, p, p, i64, p, p, p, i64, f32, f32, f32, p]),
    "rec_adam_keras_flush_f32": (i32, [p, i64, i64, p, p, i64, p, p, i64, p, p, p, i64, f32, f32, f32, p]),
    "rec_adam_dense_f32": (i32, [p, p, p, p, i64, i64, f32, f32, f32, f32, p]),
    "rec_adam_sparse_keras_f32": (i32, [p, i64, p, p, i64, i32, p, p, p, i64, p, i64, f32, f32, f32, f32, p]),
    "rec_adam_sparse_keras_pair_f32": (i32, [p, i64, p, p, p, p, i64, i32, p, p, p, p, i64, p, p, i64, f32, f32, f32, f32, p]),
    "rec_adam_rows_f32": (i32, [p, i64, p, p, i64, i32, p, p, p, i64, i64, f32, f32, f32, f32, p]),
    "rec_l2_rows_workspace_bytes": (sz, [i64, i32]),
    "rec_l2_rows_f32": (i32, [p, i64, i64, i32, p, p, i64, f32, p, p, p, p]),
    "rec_l2_normalize_rows_f32": (i32, [p, i64, i32, i64, p, i64, p]),
    "rec_topk_l2_workspace_bytes": (sz, [i64, i64, i32]),
    "rec_topk_l2_f32": (i32, [p, i64, i32, i64, p, i64, i64, i32, p, p, p, sz, p]),
    "rec_emb_ipn_fwd_f32": (i32, [p, i64, i32, i64, p, i64, i32, p, i64, p, p]),
    "rec_emb_ipn_bwd_vals_f32": (i32, [p, i64, p, i64, i64, i32, i32, p, p]),
    "rec_emb_bi_fwd_f32": (i32, [p, i64, i32, i64, p, i64, i32, p, i64, p, p, p]),
    "rec_emb_bi_bwd_vals_f32": (i32, [p, i64, i32, i64, p, i64, i32, p, i64, p, p, p]),
    "rec_ip_attn_fwd_f32": (i32, [p, i64, i64, i32, i32, p, i64, i32, p, i64, i64, p, p, i64, p, p]),
    "rec_ip_attn_bwd_f32": (i32, [p, i64, i64, i32, i32, p, i64, i32, p, i64, i64, p, p, i64, p, p, p]),
    "rec_ffm_fwd_f32": (i32, [p, i64, p, i64, p, i64, i32, p, i64, i32, p, p, p, p]),
    "rec_ffm_bwd_rows_f32": (i32, [p, i64, i64, i32, p, i64, i32, p, p, p, p, p, p]),
    "rec_batchnorm_workspace_bytes": (sz, [i64, i32]),
    "rec_batchnorm_fwd_f32": (i32, [p, i64, i64, i32, p, p, f32, f32, i32, p, p, p, p, p, p, p]),
    "rec_batchnorm_bwd_f32": (i32, [p, p, p, i64, i32, p, i32, p, p, p, p, p]),
    "rec_shard_bucketize_workspace_bytes": (sz, [i64, i32]),
    "rec_shard_bucketize_i64": (i32, [p, i64, i64, i32, p, p, p, p, p, sz, p]),
    "rec_colsort_shard_map_i64": (i32, [p, p, p, p, i64, i32, i64, i32, p, p, p, p, p, p]),
    "rec_colsort_shard_map_fixed_i64": (i32, [p, p, p, p, i64, i32, i64, i32, i64, p, p, p, p, p, p]),
    "rec_emb_gather_lists_f32": (i32, [p, i64, i32, i64, p, i32, i64, p, p, p]),
    "rec_block_copy": (i32, [p, i32, i64, p, p]),
    "rec_auc_hist_update_f32": (i32, [p, p, i64, p, i32, p, p, i32, p, p]),
    "rec_shard_slab_map_i64": (i32, [p, p, p, p, i64, i64, i32, i64, p, p, p, p]),
    "rec_shard_slab_map_uslot_i64": (i32, [p, p, p, p, i64, i64, i32, i64, p, p, p, p, p]),
    "rec_dedup_plan_sorted_slabs_i64": (i32, [p, i32, i64, i64, p, p, p, p, p, sz, p]),
    "rec_deepfm_fused_post_slots_f32": (i32, [i32, i64] + [p] * 17 + [p]),
    "rec_permute_rows_f32": (i32, [p, p, i64, i32, i32, p, p]),
    "rec_deepfm_fused_workspace_bytes": (sz, [i64, i32]),
    "rec_deepfm_fused_fwd_bwd_f32": (i32, [p, i64, i64, p, i32, i64, p, p, p, p, p, p, p, p, p, p, p, p, p, p, p, p, p,
                                           p, p, p, p, p]),
    "rec_deepfm_fused_step_f32": (i32, [p, i64, i64, p, i32, i64] + [p] * 21 + [p] * 8 + [i32, p]),
    "rec_deepfm_fused_main_f32": (i32, [p, i64, i64, p, i32, i64] + [p] * 13 + [p]),
    "rec_deepfm_fused_post_f32": (i32, [i32, i64] + [p] * 19 + [i32, p]),
    "rec_colsort_workspace_bytes": (sz, [i64, i32]),
    "rec_colsort_plan_i64": (i32, [p, i32, i64, i64, p, i64, p, p, p, p, p, p, p]),
    "rec_colsort_plan_dest_i64": (i32, [p, i32, i64, i64, p, i64, p, p, p, p, p, p, p, p]),
    "rec_deepfm_fused_main_direct_f32": (i32, [p, i64, i64, p, i32, i64] + [p] * 13 + [p, p, p] + [p]),
    "rec_deepfm_fused_post_direct_f32": (i32, [i32, i64] + [p] * 19 + [p]),
    "rec_deepfm_k0t_f32": (i32, [p, i32, p, p]),
    "rec_deepfm_fused3_main_f32": (i32, [p, i64, i64, p, i32, i64] + [p] * 14 + [p]),
    "rec_deepfm_fused3_main_direct_f32": (i32, [p, i64, i64, p, i32, i64] + [p] * 14 + [p, p, p] + [p]),
    "rec_deepfm_fused3_main_direct_adv_f32": (i32, [p, i64, i64, p, i32, i64] + [p] * 14 + [p, p, p] + [p, p, i64, p] + [p]),
    "rec_deepfm_fused_post_direct_adam_f32": (i32, [i32, i64] + [p] * 19 + [p, i64, i64, p, p, p, p, i64, f32, f32, f32,
                                                                          f32, p]),
    "rec_colseg_sum_f32": (i32, [p, p, p, p, p, p, i64, i32, p, p, p, p, p]),
    "rec_colseg_sum_packed_f32": (i32, [p, p, p, p, p, p, i64, i32, p, p, p, p]),
    "rec_din_prepare_f32": (i32, [p, p, i32, i32, p, p, p, p]),
    "rec_din_prepare_bwd_f32": (i32, [p, p, i32, i32, p, p]),
    "rec_din_attn_fwd_f32": (i32, [p, i64, i64, i32, i32, p, i64, i32, p, p, i32, i32, p, p, p, p, p, i64, i32, p, p, p,
                                   p]),
    "rec_din_attn_bwd_f32": (i32, [p, i64, i64, i32, i32, p, i64, i32, p, p, i32, i32, p, p, p, p, p, i64, i32, p, p, p,
                                   p, p, p, p, p]),
    "rec_feat_act_fwd_f32": (i32, [i32, p, p, p, p, p, i64, i32, p]),
    "rec_feat_act_bwd_f32": (i32, [i32, p, p, p, p, p, p, p, i64, i32, p]),
    "rec_layernorm_fwd_f32": (i32, [p, p, p, i64, i32, p, p, p, p]),
    "rec_layernorm_bwd_f32": (i32, [p, p, p, p, i64, i32, p, p, p]),
    "rec_softmax_fwd_f32": (i32, [p, i64, i32, p, p]),
    "rec_softmax_bwd_f32": (i32, [p, p, i64, i32, p, p]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: the HIP extension is required (no CPU fallback exists). "
            "Build it with __graft_entry__.build() or csrc/build.sh" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # header / library out of sync
            raise ImportError("libmi355rec.so does not export %s: rebuild it" % name) from e
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()

# The hot operators are also registered with the PyTorch dispatcher -- TORCH_LIBRARY(mi355rec, ...) in csrc/torch_ops.cpp,
# a host-only wrapper over the SAME C ABI -- and ops.py calls them as torch.ops.mi355rec.<op>: one dispatcher hop per
# operator instead of a ctypes call with ~20 marshalled arguments.  Part of the product: missing -> ImportError.
TORCH_LIB_PATH = os.path.join(_HERE, "csrc", "libmi355rec_torch.so")


def _load_torch_ops():
    import torch
    if not os.path.exists(TORCH_LIB_PATH):
        raise ImportError("%s not found: build it with __graft_entry__.build() or csrc/build.sh" % TORCH_LIB_PATH)
    torch.ops.load_library(TORCH_LIB_PATH)
    return torch.ops.mi355rec


tops = _load_torch_ops()


class RecError(RuntimeError):
    pass


def check(status, what):
    if status == 0:
        return
    if status == -1:
        raise ValueError("%s: invalid argument" % what)
    if status == -2:
        raise NotImplementedError("%s: unsupported configuration" % what)
    if status == -3:
        raise RecError("%s: workspace too small" % what)
    raise RecError("%s: hipError_t %d" % (what, status))
