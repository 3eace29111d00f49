"""Tensor-level wrappers over the C ABI (include/mi355rec.h).

torch is plumbing here: it owns device memory and the current HIP stream; every function below hands raw
device pointers and the stream handle to libmi355rec.so and returns torch tensors that view the results.
Nothing in this file computes on the CPU or through torch ops -- a non-CUDA tensor is an error.
"""
import ctypes as C

import torch

from ._lib import lib, check, tops

EPI_NONE, EPI_BIAS, EPI_BIAS_RELU, EPI_BIAS_SIGMOID, EPI_BIAS_TANH, EPI_CROSS, EPI_ADD = range(7)
ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH = range(4)
ACT_CODE = {None: ACT_NONE, "linear": ACT_NONE, "relu": ACT_RELU, "sigmoid": ACT_SIGMOID, "tanh": ACT_TANH}
EPI_OF_ACT = {ACT_NONE: EPI_BIAS, ACT_RELU: EPI_BIAS_RELU, ACT_SIGMOID: EPI_BIAS_SIGMOID, ACT_TANH: EPI_BIAS_TANH}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _req(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("%s must be a tensor on the MI355X (got %r): the HIP path has no CPU fallback"
                           % (name, getattr(t, "device", type(t))))
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t


def _f32(t, name):
    return _req(t, torch.float32, name)


def _i64(t, name):
    return _req(t, torch.int64, name)


def _table(t, name):
    """A table may be a strided row view (fused layout): 2-D fp32 CUDA with unit inner stride."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("%s must be a tensor on the MI355X: the HIP path has no CPU fallback" % name)
    if t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1 or t.stride(0) < t.shape[1]:
        raise ValueError("%s must be a 2-D fp32 table with unit inner stride" % name)
    return t


def fused_row_stride(E):
    """Row stride (floats) of the fused [embed(E) | w | pad] layout: next power of two >= E+1, at least 16."""
    ld = 16
    while ld < E + 1:
        ld *= 2
    return ld


# ---------------------------------------------------------------------------------------------------
# K1 / K2 / K3
# ---------------------------------------------------------------------------------------------------

def index_pack(cols, out=None, col0=0):
    """cols: list of int64 tensors with the same number of elements ([B], [B,1] or [B,T]).
    Returns X [rows, F] int64 (expand_dims + concat(axis=1), 2.FM/CustomLayers.py:138-144)."""
    F = len(cols)
    rows = cols[0].numel()
    for c in cols:
        _i64(c, "index column")
        if c.numel() != rows:
            raise ValueError("index columns differ in length")
    return tops.index_pack(list(cols), out, col0)


def new_flag(device):
    return torch.zeros(1, dtype=torch.int32, device=device)


def emb_gather(table, idx, oob=None):
    _table(table, "table"); _i64(idx, "idx")
    return tops.emb_gather(table, idx, oob)


def emb_fm_fwd(embed, w, bias, X, want_prob=False, want_rows=False, want_sum=True, oob=None):
    """Fused w(X), embed(X) and the FM sum-square trick.  Returns z [B], prob [B]|None, rows [B,F,E]|None,
    sumvec [B,E]|None."""
    _table(embed, "embed"); _table(w, "w"); _f32(bias, "bias"); _i64(X, "X")
    return tops.emb_fm_fwd(embed, w, bias, X, want_prob, want_rows, want_sum, oob)


def emb_fm_bwd_vals(embed, X, gz, sumvec, rows=None, extra=None):
    """IndexedSlices values of the FM part: [B*F, E]."""
    return tops.emb_fm_bwd_vals(embed, X, _f32(gz, "gz"), sumvec, rows, extra)


# ---------------------------------------------------------------------------------------------------
# K4 de-duplication
# ---------------------------------------------------------------------------------------------------

class DedupPlan:
    """Sorted-unique plan of a flat id list; reusable for every table indexed by the same ids.

    ``list_counts`` (int64 device tensor [P]): the ids are P ascending duplicate-free lists laid end to end (what P
    requesters send to a shard owner after de-duplicating their own batches); the plan is then a rank merge instead
    of a radix sort (rec_dedup_plan_sorted_lists_i64)."""

    def __init__(self, ids, V, list_counts=None):
        ids = _i64(ids.reshape(-1), "ids")
        n = ids.numel()
        dev = ids.device
        self.n = n
        if list_counts is None:
            self.uniq_ids, self.seg_start, self.perm, self.n_uniq = tops.dedup_plan(ids, V)
        else:
            self.uniq_ids = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
            self.seg_start = torch.empty(n + 1, dtype=torch.int32, device=dev)
            self.perm = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
            self.n_uniq = torch.empty(1, dtype=torch.int64, device=dev)
            nbytes = lib.rec_dedup_workspace_bytes(n)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            lc = _i64(list_counts.reshape(-1), "list_counts")
            check(lib.rec_dedup_plan_sorted_lists_i64(_ptr(ids), n, _ptr(lc), lc.numel(), V, _ptr(self.uniq_ids),
                                                      _ptr(self.seg_start), _ptr(self.perm), _ptr(self.n_uniq),
                                                      _ptr(ws), nbytes, _stream()),
                  "rec_dedup_plan_sorted_lists_i64")

    def segment_sum(self, vals, E, row_div=1):
        """vals [n/row_div, E] -> [n, E]; rows >= n_uniq are zero."""
        if E > 256:      # wide rows (FFM's F*E): the kernel takes up to 256 columns at a time
            out = torch.empty((max(self.n, 1), E), dtype=torch.float32, device=vals.device)
            for c0 in range(0, E, 256):
                c1 = min(E, c0 + 256)
                out[:, c0:c1] = self.segment_sum(vals[:, c0:c1].contiguous(), c1 - c0, row_div)
            return out
        return tops.segment_sum(_f32(vals, "vals"), E, self.perm, self.seg_start, self.n, row_div)


def l2_used_rows(table, plan, factor):
    """factor * l2_loss(table[unique ids of the plan]) (5.DIN/ModelManager.py:188-190) -> (loss [1], its gradient as
    rows [n,E] aligned with plan.uniq_ids; zero beyond n_uniq)."""
    _table(table, "table")
    V, E = table.shape
    n = plan.n
    dev = table.device
    rows = torch.empty((max(n, 1), E), dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    ws = torch.empty(lib.rec_l2_rows_workspace_bytes(n, E) // 4 + 1, dtype=torch.float32, device=dev)
    check(lib.rec_l2_rows_f32(_ptr(table), table.stride(0), V, E, _ptr(plan.uniq_ids), _ptr(plan.n_uniq), n,
                              float(factor), _ptr(rows), _ptr(loss), _ptr(ws), _stream()), "rec_l2_rows_f32")
    return loss, rows


# ---------------------------------------------------------------------------------------------------
# dense
# ---------------------------------------------------------------------------------------------------

def gemm(A, B, transA=False, transB=False, epi=EPI_NONE, bias=None, e0=None, e1=None, split_k=None, out=None,
         aux=None):
    """C = epi(op(A) @ op(B)) on the fp32 matrix cores.  A, B are 2-D row-major (leading dim = stride(0)).
    ``split_k=None``: chosen by split_k_for (a deep reduction over few output tiles is cut into slices that are added
    in slice order -- e.g. DIN's gq = gMext . Wcat^T, [4096,3492] x [3492,96], ran on 32 workgroups)."""
    for t, nm in ((A, "A"), (B, "B")):
        if t.dtype != torch.float32 or not t.is_cuda or t.dim() != 2 or t.stride(1) != 1:
            raise ValueError("%s must be a 2-D fp32 CUDA tensor with unit inner stride" % nm)
    M, K = (A.shape[1], A.shape[0]) if transA else (A.shape[0], A.shape[1])
    K2, N = (B.shape[1], B.shape[0]) if transB else (B.shape[0], B.shape[1])
    if K != K2:
        raise ValueError("gemm inner dimensions differ: %d vs %d" % (K, K2))
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    if out.dim() != 2 or tuple(out.shape) != (M, N) or out.stride(1) != 1 or out.dtype != torch.float32:
        raise ValueError("out must be a 2-D fp32 [%d,%d] tensor with unit inner stride" % (M, N))
    if aux is not None:
        # the kernel writes U with C's leading dimension: a contiguous aux beside a column-slice `out` would be
        # written out of bounds
        if (tuple(aux.shape) != (M, N) or aux.dtype != torch.float32 or aux.stride(1) != 1
                or aux.stride(0) != out.stride(0)):
            raise ValueError("aux must be fp32 [%d,%d] with the same row stride as out (%d), got stride %s"
                             % (M, N, out.stride(0), tuple(aux.stride())))
    if split_k is None:
        skinny = (not transA) and (not transB) and N <= 64 and M >= 256      # the no-LDS kernel splits K over its waves
        split_k = 1 if skinny else split_k_for(K, M, N, transA, transB)
    return tops.gemm(A, B, bool(transA), bool(transB), epi, bias, e0, e1, int(split_k), out, aux)


def act_fwd(act, x, x2=None):
    """y = act(x + x2)."""
    return tops.act_fwd(act, _f32(x, "x"), x2)


def crossnet_mat_bwd_elem(g, x0, u, gx0, accumulate):
    """h = g*x0 (returned); gx0 (+)= g*u in place."""
    return tops.crossnet_mat_bwd_elem(_f32(g, "g"), x0, u, gx0, bool(accumulate))


def act_bwd(act, post, dpost):
    return tops.act_bwd(act, _f32(post, "post"), _f32(dpost, "dpost"))


def colsum(X, out=None):
    return tops.colsum(X, out)


def axpby(a, x, b, y):
    return tops.axpby(float(a), _f32(x, "x"), float(b), _f32(y, "y"))


def copy_cols(src, dst_view):
    """dst_view[:, :] = src, both 2-D with unit inner stride (dst may be a column block of a wider buffer)."""
    rows, w = src.shape
    check(lib.rec_copy_cols_f32(_ptr(src), src.stride(0), _ptr(dst_view), dst_view.stride(0), rows, w, _stream()),
          "rec_copy_cols_f32")
    return dst_view


def split_k_for(K, M, N, transA=False, transB=False):
    """Heuristic split of a reduction over the batch.  The tile kernels keep up to three workgroups per CU, and the time of
    a few-tile weight gradient [M,N] = X^T dY with K = batch is a staircase in tiles x slices: it is best just below a
    multiple of the 256 CUs (measured, M = N = 835, K = 16384, 49 tiles: 8 slices 324 us, 10: 265, 12: 313, 15: 260,
    21: 300; M = N = 323, 9 tiles: 24 slices 79 us, 56: 64, 57: 77).  So: the largest number of slices that keeps
    tiles x slices <= 768 (512 for very few tiles, whose partials are cheap to add either way), at least 64 k per slice
    and at most 64 MB of partials; these are added in slice order by the split-K reduce."""
    t = 128 if (M > 64 and N > 64) else 64          # tile edge the kernel will use
    tiles = ((M + t - 1) // t) * ((N + t - 1) // t)
    if tiles >= 512 or K < 1024:
        return 1
    slots = 512 if tiles < 16 else 768
    split = max(1, min(256, slots // max(tiles, 1), K // 64))
    cap = max(8, (64 << 20) // max(1, 4 * M * N))       # the partials are written and read again
    return int(min(split, cap))


# ---------------------------------------------------------------------------------------------------
# CrossNet vector mode, cosine, BCE, Adam
# ---------------------------------------------------------------------------------------------------

def crossnet_vec_fwd(x0, w, b, save=True):
    B, D = x0.shape
    L = w.shape[0]
    y = torch.empty_like(x0)
    xs = torch.empty((L, B, D), dtype=torch.float32, device=x0.device) if save else None
    check(lib.rec_crossnet_vec_fwd_f32(_ptr(_f32(x0, "x0")), B, D, L, _ptr(_f32(w, "w")), _ptr(_f32(b, "b")),
                                       _ptr(y), _ptr(xs), _stream()), "rec_crossnet_vec_fwd_f32")
    return y, xs


def crossnet_vec_bwd(x0, w, xs, gy):
    B, D = x0.shape
    L = w.shape[0]
    gx0 = torch.empty_like(x0)
    dw = torch.empty_like(w)
    db = torch.empty_like(w)
    nbytes = lib.rec_crossnet_vec_bwd_workspace_bytes(B, D, L)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x0.device)
    check(lib.rec_crossnet_vec_bwd_f32(_ptr(x0), B, D, L, _ptr(w), _ptr(xs), _ptr(_f32(gy, "gy")), _ptr(gx0),
                                       _ptr(dw), _ptr(db), _ptr(ws), _stream()), "rec_crossnet_vec_bwd_f32")
    return gx0, dw, db


def cosine_fwd(u, i):
    return tops.cosine_fwd(_f32(u, "u"), _f32(i, "i"))


def cosine_bwd(u, i, gout):
    return tops.cosine_bwd(u, i, _f32(gout, "gout"))


def bce_fwd_bwd(y, p, want_dp=True, want_dz=False):
    y = _f32(y.reshape(-1), "y")
    p = _f32(p.reshape(-1), "p")
    return tops.bce_fwd_bwd(y, p, bool(want_dp), bool(want_dz))


def adam_dense(var, m, v, g, t, lr, b1=0.9, b2=0.999, eps=1e-7):
    check(lib.rec_adam_dense_f32(_ptr(_f32(var, "var")), _ptr(m), _ptr(v), _ptr(_f32(g, "g")), var.numel(), t, lr,
                                 b1, b2, eps, _stream()), "rec_adam_dense_f32")


def adam_sparse_keras(var, m, v, uniq_ids, g_rows, n_uniq, t, lr, b1=0.9, b2=0.999, eps=1e-7):
    V, E = var.shape
    cap = g_rows.shape[0]
    side = torch.empty((cap, 3, E), dtype=torch.float32, device=var.device)
    check(lib.rec_adam_sparse_keras_f32(_ptr(_table(var, "var")), var.stride(0), _ptr(m), _ptr(v), V, E, _ptr(uniq_ids),
                                        _ptr(g_rows), _ptr(n_uniq), cap, _ptr(side), t, lr, b1, b2, eps, _stream()),
          "rec_adam_sparse_keras_f32")


def adam_rows(var, m, v, uniq_ids, g_rows, n_uniq, t, lr, b1=0.9, b2=0.999, eps=1e-7):
    V, E = var.shape
    check(lib.rec_adam_rows_f32(_ptr(_table(var, "var")), var.stride(0), _ptr(m), _ptr(v), V, E, _ptr(uniq_ids),
                                _ptr(g_rows), _ptr(n_uniq), g_rows.shape[0], t, lr, b1, b2, eps, _stream()),
          "rec_adam_rows_f32")


# ---------------------------------------------------------------------------------------------------
# sharding
# ---------------------------------------------------------------------------------------------------

def shard_bucketize(ids, rows_per_shard, n_shard, oob=None):
    ids = _i64(ids.reshape(-1), "ids")
    n = ids.numel()
    dev = ids.device
    perm = torch.empty(n, dtype=torch.int64, device=dev)
    counts = torch.empty(n_shard, dtype=torch.int64, device=dev)
    local = torch.empty(n, dtype=torch.int64, device=dev)
    nbytes = lib.rec_shard_bucketize_workspace_bytes(n, n_shard)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    check(lib.rec_shard_bucketize_i64(_ptr(ids), n, rows_per_shard, n_shard, _ptr(perm), _ptr(counts), _ptr(local),
                                      _ptr(oob), _ptr(ws), nbytes, _stream()), "rec_shard_bucketize_i64")
    return perm, counts, local


def permute_rows(x, perm, scatter):
    """scatter=True: out[perm[i]] = x[i];  scatter=False: out[i] = x[perm[i]]."""
    n, E = x.shape
    out = torch.empty_like(x)
    check(lib.rec_permute_rows_f32(_ptr(_f32(x, "x")), _ptr(_i64(perm, "perm")), n, E, int(scatter), _ptr(out),
                                   _stream()), "rec_permute_rows_f32")
    return out


# ---------------------------------------------------------------------------------------------------
# retrieval
# ---------------------------------------------------------------------------------------------------
def l2_normalize_rows(x):
    """x / ||x||_2 per row (2.FM/OfflineLoader.py:140)."""
    _table(x, "x")                                       # 2-D fp32 CUDA, unit inner stride, any row stride
    y = torch.empty((x.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
    check(lib.rec_l2_normalize_rows_f32(_ptr(x), x.shape[0], x.shape[1], x.stride(0), _ptr(y), y.stride(0), _stream()),
          "rec_l2_normalize_rows_f32")
    return y


def topk_l2(queries, items, k):
    """BallTree(items).query(queries, k) of the reference, brute force: (dist [nq,k] ascending, ind [nq,k] int64)."""
    _table(queries, "queries"); _table(items, "items")   # row views of wider buffers are fine
    if queries.shape[1] != items.shape[1]:
        raise ValueError("queries [nq,d] and items [n,d] must share d")
    nq, d = queries.shape
    n = items.shape[0]
    dev = queries.device
    ind = torch.empty((nq, k), dtype=torch.int64, device=dev)
    dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
    nbytes = lib.rec_topk_l2_workspace_bytes(nq, n, k)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    check(lib.rec_topk_l2_f32(_ptr(queries), nq, d, queries.stride(0), _ptr(items), n, items.stride(0), k, _ptr(ind),
                              _ptr(dist), _ptr(ws), nbytes, _stream()), "rec_topk_l2_f32")
    return dist, ind


# ---------------------------------------------------------------------------------------------------
# DIN
# ---------------------------------------------------------------------------------------------------
DACT_NONE, DACT_RELU, DACT_SIGMOID, DACT_TANH, DACT_DICE, DACT_PRELU = range(6)
DACT_CODE = {None: DACT_NONE, "linear": DACT_NONE, "relu": DACT_RELU, "sigmoid": DACT_SIGMOID, "tanh": DACT_TANH,
             "dice": DACT_DICE, "prelu": DACT_PRELU}


def din_prepare(W1, b1, D, H):
    """W1 [3D+D*D, H], b1 [H] -> Wcat [D, D*H+H], Wkd [D,H], bext [D*H+H]."""
    dev = W1.device
    N = D * H + H
    Wcat = torch.empty((D, N), dtype=torch.float32, device=dev)
    Wkd = torch.empty((D, H), dtype=torch.float32, device=dev)
    bext = torch.empty(N, dtype=torch.float32, device=dev)
    check(lib.rec_din_prepare_f32(_ptr(_f32(W1, "W1")), _ptr(_f32(b1, "b1")), D, H, _ptr(Wcat), _ptr(Wkd), _ptr(bext),
                                  _stream()), "rec_din_prepare_f32")
    return Wcat, Wkd, bext


def din_prepare_bwd(gWcat, gWkd, D, H):
    gW1 = torch.empty((3 * D + D * D, H), dtype=torch.float32, device=gWcat.device)
    check(lib.rec_din_prepare_bwd_f32(_ptr(_f32(gWcat, "gWcat")), _ptr(_f32(gWkd, "gWkd")), D, H, _ptr(gW1), _stream()),
          "rec_din_prepare_bwd_f32")
    return gW1


def _attn_common(embed, series, Mext, Wkd, act, alpha, mean, var, w2, b2, padding_index, mask_valid):
    _table(embed, "embed"); _i64(series, "series")
    V, E = embed.shape
    B, T, Cn = series.shape
    H = Wkd.shape[1]
    return [_ptr(embed), embed.stride(0), V, E, Cn, _ptr(series), B, T, _ptr(_f32(Mext, "Mext")), _ptr(_f32(Wkd, "Wkd")),
            H, act, _ptr(alpha), _ptr(mean), _ptr(var), _ptr(_f32(w2, "w2")), _ptr(_f32(b2, "b2")), int(padding_index),
            int(bool(mask_valid))], (B, T, Cn * E, H)


def din_attn_fwd(embed, series, Mext, Wkd, act, alpha, mean, var, w2, b2, padding_index, mask_valid, oob=None):
    args, (B, T, D, H) = _attn_common(embed, series, Mext, Wkd, act, alpha, mean, var, w2, b2, padding_index,
                                      mask_valid)
    scores = torch.empty((B, T), dtype=torch.float32, device=embed.device)
    pooled = torch.empty((B, D), dtype=torch.float32, device=embed.device)
    check(lib.rec_din_attn_fwd_f32(*args, _ptr(scores), _ptr(pooled), _ptr(oob), _stream()), "rec_din_attn_fwd_f32")
    return scores, pooled


def din_attn_bwd(embed, series, Mext, Wkd, act, alpha, mean, var, w2, b2, padding_index, mask_valid, scores, gpooled,
                 gkeys=None):
    """``gkeys``: optional preallocated contiguous [B,T,D] destination (the tail of a shared value buffer)."""
    args, (B, T, D, H) = _attn_common(embed, series, Mext, Wkd, act, alpha, mean, var, w2, b2, padding_index,
                                      mask_valid)
    dev = embed.device
    f32 = dict(dtype=torch.float32, device=dev)
    if gkeys is None:
        gkeys = torch.empty((B, T, D), **f32)
    _f32(gkeys, "gkeys")
    gMext = torch.empty((B, D * H + H), **f32)
    gw2p = torch.empty((B, H), **f32)
    galphap = torch.empty((B, H), **f32)
    gb2p = torch.empty((B, 1), **f32)
    check(lib.rec_din_attn_bwd_f32(*args, _ptr(_f32(scores, "scores")), _ptr(_f32(gpooled, "gpooled")), _ptr(gkeys),
                                   _ptr(gMext), _ptr(gw2p), _ptr(galphap), _ptr(gb2p), _stream()),
          "rec_din_attn_bwd_f32")
    return gkeys, gMext, gw2p, galphap, gb2p


def feat_act_fwd(kind, x, alpha=None, mean=None, var=None):
    M, N = x.shape
    y = torch.empty_like(x)
    check(lib.rec_feat_act_fwd_f32(kind, _ptr(_f32(x, "x")), _ptr(alpha), _ptr(mean), _ptr(var), _ptr(y), M, N,
                                   _stream()), "rec_feat_act_fwd_f32")
    return y


def feat_act_bwd(kind, x, gy, alpha=None, mean=None, var=None, want_alpha=False):
    M, N = x.shape
    gx = torch.empty_like(x)
    ga = torch.empty_like(x) if want_alpha else None
    check(lib.rec_feat_act_bwd_f32(kind, _ptr(x), _ptr(_f32(gy, "gy")), _ptr(alpha), _ptr(mean), _ptr(var), _ptr(gx),
                                   _ptr(ga), M, N, _stream()), "rec_feat_act_bwd_f32")
    return gx, ga


def layernorm_fwd(x, gamma, beta):
    M, N = x.shape
    y, xhat = torch.empty_like(x), torch.empty_like(x)
    rstd = torch.empty(M, dtype=torch.float32, device=x.device)
    check(lib.rec_layernorm_fwd_f32(_ptr(_f32(x, "x")), _ptr(_f32(gamma, "gamma")), _ptr(_f32(beta, "beta")), M, N,
                                    _ptr(y), _ptr(xhat), _ptr(rstd), _stream()), "rec_layernorm_fwd_f32")
    return y, xhat, rstd


def layernorm_bwd(gy, xhat, rstd, gamma):
    M, N = gy.shape
    gx, gg = torch.empty_like(gy), torch.empty_like(gy)
    check(lib.rec_layernorm_bwd_f32(_ptr(_f32(gy, "gy")), _ptr(xhat), _ptr(rstd), _ptr(gamma), M, N, _ptr(gx), _ptr(gg),
                                    _stream()), "rec_layernorm_bwd_f32")
    return gx, gg


def softmax_fwd(x):
    M, N = x.shape
    y = torch.empty_like(x)
    check(lib.rec_softmax_fwd_f32(_ptr(_f32(x, "x")), M, N, _ptr(y), _stream()), "rec_softmax_fwd_f32")
    return y


def softmax_bwd(y, gy):
    M, N = y.shape
    gx = torch.empty_like(y)
    check(lib.rec_softmax_bwd_f32(_ptr(y), _ptr(_f32(gy, "gy")), M, N, _ptr(gx), _stream()), "rec_softmax_bwd_f32")
    return gx


# ---------------------------------------------------------------------------------------------------
# f4: sibling interaction layers on the same gather (PNN inner product, NFM bi-interaction, SIM GSU attention)
# ---------------------------------------------------------------------------------------------------

def emb_ipn_fwd(table, X, oob=None):
    """[Flatten(embed(X)) | <e_i,e_j> for i<j]  ->  [B, F*E + F(F-1)/2]  (2.FM/CustomLayers.py:737-745,755-792)."""
    _table(table, "table"); _i64(X, "X")
    V, E = table.shape
    B, F = X.shape
    W = F * E + F * (F - 1) // 2
    out = torch.empty((B, W), dtype=torch.float32, device=table.device)
    check(lib.rec_emb_ipn_fwd_f32(_ptr(table), V, E, table.stride(0), _ptr(X), B, F, _ptr(out), W, _ptr(oob), _stream()),
          "rec_emb_ipn_fwd_f32")
    return out


def emb_ipn_bwd_vals(out, g, F, E):
    _f32(out, "out"); _f32(g, "g")
    B, W = out.shape
    vals = torch.empty((B * F, E), dtype=torch.float32, device=out.device)
    check(lib.rec_emb_ipn_bwd_vals_f32(_ptr(out), W, _ptr(g), g.shape[1], B, F, E, _ptr(vals), _stream()),
          "rec_emb_ipn_bwd_vals_f32")
    return vals


def emb_bi_fwd(table, X, out=None, oob=None):
    """NFM bi-interaction 0.5*((sum e)^2 - sum e^2) -> the leading E columns of `out` (a [B, >=E] fp32 buffer, fresh
    [B,E] when None) and sumvec [B,E]."""
    _table(table, "table"); _i64(X, "X")
    V, E = table.shape
    B, F = X.shape
    if out is None:
        out = torch.empty((B, E), dtype=torch.float32, device=table.device)
    _f32(out, "out")
    S = torch.empty((B, E), dtype=torch.float32, device=table.device)
    check(lib.rec_emb_bi_fwd_f32(_ptr(table), V, E, table.stride(0), _ptr(X), B, F, _ptr(out), out.shape[1], _ptr(S),
                                 _ptr(oob), _stream()), "rec_emb_bi_fwd_f32")
    return out, S


def emb_bi_bwd_vals(table, X, g, S):
    """g: [B, >=E] (only the leading E columns are read)."""
    _table(table, "table"); _i64(X, "X"); _f32(g, "g"); _f32(S, "S")
    V, E = table.shape
    B, F = X.shape
    vals = torch.empty((B * F, E), dtype=torch.float32, device=table.device)
    check(lib.rec_emb_bi_bwd_vals_f32(_ptr(table), V, E, table.stride(0), _ptr(X), B, F, _ptr(g), g.shape[1], _ptr(S),
                                      _ptr(vals), _stream()), "rec_emb_bi_bwd_vals_f32")
    return vals


def ip_attn_fwd(embed, series, q, padding_index, oob=None):
    """series [B,T,C] int64, q [B,C*E] -> masked scores [B,T], pooled [B,C*E]  (7.SIM/CustomLayers.py:88-96)."""
    _table(embed, "embed"); _i64(series, "series"); _f32(q, "q")
    V, E = embed.shape
    B, T, C = series.shape
    D = C * E
    scores = torch.empty((B, T), dtype=torch.float32, device=embed.device)
    pooled = torch.empty((B, D), dtype=torch.float32, device=embed.device)
    check(lib.rec_ip_attn_fwd_f32(_ptr(embed), embed.stride(0), V, E, C, _ptr(series), B, T, _ptr(q), q.shape[1],
                                  int(padding_index), _ptr(scores), _ptr(pooled), D, _ptr(oob), _stream()),
          "rec_ip_attn_fwd_f32")
    return scores, pooled


def ip_attn_bwd(embed, series, q, padding_index, scores, gpooled, gkeys=None):
    _table(embed, "embed"); _i64(series, "series"); _f32(q, "q"); _f32(scores, "scores"); _f32(gpooled, "gpooled")
    V, E = embed.shape
    B, T, C = series.shape
    D = C * E
    if gkeys is None:
        gkeys = torch.empty((B, T, D), dtype=torch.float32, device=embed.device)
    _f32(gkeys, "gkeys")
    gq = torch.empty((B, D), dtype=torch.float32, device=embed.device)
    check(lib.rec_ip_attn_bwd_f32(_ptr(embed), embed.stride(0), V, E, C, _ptr(series), B, T, _ptr(q), q.shape[1],
                                  int(padding_index), _ptr(scores), _ptr(gpooled), gpooled.shape[1], _ptr(gkeys),
                                  _ptr(gq), _stream()), "rec_ip_attn_bwd_f32")
    return gkeys, gq


def batchnorm_fwd(x, gamma, beta, moving_mean, moving_var, training, eps=1e-3, momentum=0.99, save=True):
    """Keras BatchNormalization on [B,N]; moving statistics are updated in place when training."""
    _f32(x, "x")
    B, N = x.shape
    y = torch.empty_like(x)
    xhat = torch.empty_like(x) if save else None
    rstd = torch.empty(N, dtype=torch.float32, device=x.device) if save else None
    ws = torch.empty(lib.rec_batchnorm_workspace_bytes(B, N) // 4, dtype=torch.float32, device=x.device)
    check(lib.rec_batchnorm_fwd_f32(_ptr(x), N, B, N, _ptr(gamma), _ptr(beta), float(eps), float(momentum),
                                    1 if training else 0, _ptr(moving_mean), _ptr(moving_var), _ptr(y), _ptr(xhat),
                                    _ptr(rstd), _ptr(ws), _stream()), "rec_batchnorm_fwd_f32")
    return y, xhat, rstd


def batchnorm_bwd(g, xhat, rstd, gamma, training):
    _f32(g, "g")
    B, N = g.shape
    gx = torch.empty_like(g)
    ggamma = torch.empty(N, dtype=torch.float32, device=g.device)
    gbeta = torch.empty(N, dtype=torch.float32, device=g.device)
    ws = torch.empty(lib.rec_batchnorm_workspace_bytes(B, N) // 4, dtype=torch.float32, device=g.device)
    check(lib.rec_batchnorm_bwd_f32(_ptr(g), _ptr(xhat), _ptr(rstd), B, N, _ptr(gamma), 1 if training else 0, _ptr(gx),
                                    _ptr(ggamma), _ptr(gbeta), _ptr(ws), _stream()), "rec_batchnorm_bwd_f32")
    return gx, ggamma, gbeta


def ffm_fwd(v, w, bias, X, want_prob=False, oob=None):
    """v [V,F,E] field-aware table, w [V,1], bias [1], X [B,F] -> z [B] (and prob [B])."""
    _f32(v, "v"); _table(w, "w"); _f32(bias, "bias"); _i64(X, "X")
    V, F, E = v.shape
    B = X.shape[0]
    if X.shape[1] != F:
        raise ValueError("X has %d fields, the table %d" % (X.shape[1], F))
    z = torch.empty(B, dtype=torch.float32, device=v.device)
    prob = torch.empty(B, dtype=torch.float32, device=v.device) if want_prob else None
    check(lib.rec_ffm_fwd_f32(_ptr(v), F * E, _ptr(w), w.stride(0), _ptr(bias), V, E, _ptr(X), B, F, _ptr(z), _ptr(prob),
                              _ptr(oob), _stream()), "rec_ffm_fwd_f32")
    return z, prob


def ffm_bwd_rows(v, X, gz, plan):
    """-> g_rows [B*F, F, E] aligned with plan.uniq_ids (zero beyond n_uniq)."""
    _f32(v, "v"); _i64(X, "X"); _f32(gz, "gz")
    V, F, E = v.shape
    B = X.shape[0]
    rows = torch.empty((max(B * F, 1), F, E), dtype=torch.float32, device=v.device)
    check(lib.rec_ffm_bwd_rows_f32(_ptr(v), F * E, V, E, _ptr(X), B, F, _ptr(gz), _ptr(plan.perm), _ptr(plan.seg_start),
                                   _ptr(plan.n_uniq), _ptr(rows), _stream()), "rec_ffm_bwd_rows_f32")
    return rows
