"""torch.autograd.Function wrappers: forward and backward of each op are HIP kernels called through the C ABI
(ops.py).  Embedding tables receive their gradient the way TF's GradientTape delivers it -- as an
IndexedSlices-like sparse gradient -- here already de-duplicated (sorted unique ids + summed rows) and wrapped
as an uncoalesced torch sparse COO tensor whose padded tail points at a valid id with zero rows, so no
host synchronisation is needed to learn the unique count.
"""
import torch

from . import ops


class SparseRowGrad:
    """(uniq_ids [cap], rows [cap,E], n_uniq [1]) as produced by ops.DedupPlan; what the fused optimizers eat."""

    def __init__(self, uniq_ids, rows, n_uniq, shape):
        self.uniq_ids, self.rows, self.n_uniq, self.shape = uniq_ids, rows, n_uniq, shape

    def to_sparse(self):
        return torch.sparse_coo_tensor(self.uniq_ids[: self.rows.shape[0]].unsqueeze(0), self.rows, self.shape)


def _sparse_grad(plan, vals, E, shape, row_div=1):
    rows = plan.segment_sum(vals, E, row_div)
    return SparseRowGrad(plan.uniq_ids, rows, plan.n_uniq, shape).to_sparse()


class GradSink:
    """One de-duplication per table per step for a layer with several lookups into the SAME table (DIN, SIM GSU:
    profile / target-item rows through Gather, the behaviour series through the attention kernel).  The attention's
    backward -- which autograd runs first, its query comes out of the Gather -- leaves its ids and its IndexedSlices
    values here, written behind ``n_head`` free rows of one shared buffer; the Gather's backward then fills the head
    with its own values and builds ONE plan over both id lists.  Without it torch concatenates the two sparse gradients
    (a 157-MB copy at DIN config E) and each lookup sorts on its own."""

    def __init__(self, n_head, n_tail=0):
        self.n_head = int(n_head)
        self.n_tail = int(n_tail)                            # zero rows behind the series' values (sharded.ShardedEmbedding)
        self.ids = None
        self.buf = None

    def new_buf(self, n_series, E, device):
        buf = torch.empty((self.n_head + n_series + self.n_tail, E), dtype=torch.float32, device=device)
        if self.n_tail:
            buf[self.n_head + n_series:].zero_()
        return buf

    def take(self):
        ids, buf = self.ids, self.buf
        self.ids = self.buf = None
        return ids, buf


class Gather(torch.autograd.Function):
    """Embedding(V,E)(X): K2.  Backward: sparse row gradient (K4)."""

    @staticmethod
    def forward(ctx, table, X, oob, sink=None):
        out = ops.emb_gather(table, X, oob)
        ctx.save_for_backward(X)
        ctx.shape = tuple(table.shape)
        ctx.sink = sink
        return out

    @staticmethod
    def backward(ctx, g):
        (X,) = ctx.saved_tensors
        V, E = ctx.shape
        g = g.contiguous().reshape(-1, E)
        ids, buf = ctx.sink.take() if ctx.sink is not None else (None, None)
        if buf is not None:                                  # the series lookups of the same table ride along
            buf[: g.shape[0]].copy_(g)
            plan = ops.DedupPlan(torch.cat([X.reshape(-1), ids.reshape(-1)]), V)
            return _sparse_grad(plan, buf, E, (V, E)), None, None, None
        plan = ops.DedupPlan(X, V)
        return _sparse_grad(plan, g, E, (V, E)), None, None, None


class UsedRowsL2(torch.autograd.Function):
    """factor * tf.nn.l2_loss(tf.gather(table, tf.unique(ids).y))  (5.DIN/ModelManager.py:176-190): every row the
    batch touched is penalised ONCE, however often it was looked up.  Backward: sparse rows factor * table[u]."""

    @staticmethod
    def forward(ctx, table, ids, factor):
        V, E = table.shape
        plan = ops.DedupPlan(ids.reshape(-1).contiguous(), V)
        loss, rows = ops.l2_used_rows(table, plan, factor)
        ctx.save_for_backward(plan.uniq_ids, rows, plan.n_uniq)
        ctx.shape = (V, E)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        uniq_ids, rows, n_uniq = ctx.saved_tensors
        return SparseRowGrad(uniq_ids, rows * g, n_uniq, ctx.shape).to_sparse(), None, None


class EmbFM(torch.autograd.Function):
    """Fused w(X), embed(X), FM first + second order (K2+K3).  Returns z [B] and, for DeepFM, the gathered
    rows [B,F,E] that feed the DNN part.  Backward builds the IndexedSlices values (FM term + whatever came
    back through the rows) and de-duplicates once for both tables."""

    @staticmethod
    def forward(ctx, embed, w, bias, X, want_rows, oob):
        z, _, rows, S = ops.emb_fm_fwd(embed, w, bias, X, want_rows=want_rows, oob=oob)
        ctx.save_for_backward(embed, X, S, rows if want_rows else None)
        if not want_rows:
            rows = torch.empty(0, dtype=torch.float32, device=embed.device)
            ctx.mark_non_differentiable(rows)
        return z, rows

    @staticmethod
    def backward(ctx, gz, grows):
        embed, X, S, rows = ctx.saved_tensors
        V, E = embed.shape
        B, F = X.shape
        if gz is None:
            gz = torch.zeros(B, dtype=torch.float32, device=embed.device)
        gz = gz.contiguous()
        extra = grows.contiguous() if (grows is not None and rows is not None) else None
        vals = ops.emb_fm_bwd_vals(embed, X, gz, S, rows, extra)
        plan = ops.DedupPlan(X, V)
        g_embed = _sparse_grad(plan, vals, E, (V, E))
        g_w = _sparse_grad(plan, gz.reshape(B, 1), 1, (V, 1), row_div=F)
        g_bias = ops.colsum(gz.reshape(B, 1))
        return g_embed, g_w, g_bias, None, None, None


class LinearAct(torch.autograd.Function):
    """act(x @ K + b): MatMul + BiasAdd + activation of MLPLayer / Dense (K5) on the fp32 matrix cores."""

    @staticmethod
    def forward(ctx, x, K, b, act):
        x = x.contiguous()
        if b is not None:
            y = ops.gemm(x, K, epi=ops.EPI_OF_ACT[act], bias=b)
        else:
            y = ops.gemm(x, K)
            if act != ops.ACT_NONE:
                raise NotImplementedError("activation without bias")
        ctx.save_for_backward(x, K, y)
        ctx.act = act
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, K, y = ctx.saved_tensors
        gy = gy.contiguous()
        dpre = ops.act_bwd(ctx.act, y, gy) if ctx.act != ops.ACT_NONE else gy
        M, Kd = x.shape
        N = K.shape[1]
        gx = ops.gemm(dpre, K, transB=True) if ctx.needs_input_grad[0] else None
        gK = ops.gemm(x, dpre, transA=True, split_k=ops.split_k_for(M, Kd, N, True, False))
        gb = ops.colsum(dpre) if ctx.has_bias else None
        return gx, gK, gb, None


class CrossVec(torch.autograd.Function):
    """CrossLayer (3.DCN/CustomLayers.py:195-203).  w, b: [L, D]."""

    @staticmethod
    def forward(ctx, x0, w, b):
        x0 = x0.contiguous()
        y, xs = ops.crossnet_vec_fwd(x0, w.contiguous(), b.contiguous(), save=True)
        ctx.save_for_backward(x0, w, xs)
        return y

    @staticmethod
    def backward(ctx, gy):
        x0, w, xs = ctx.saved_tensors
        gx0, dw, db = ops.crossnet_vec_bwd(x0, w.contiguous(), xs, gy.contiguous())
        return gx0, dw, db


class CrossMat(torch.autograd.Function):
    """MatrixCrossLayer (3.DCN/CustomLayers.py:297-305): x_{l+1} = x0 * (x_l W_l^T + b_l) + x_l, one MFMA GEMM
    per layer with the elementwise part fused into its epilogue.  W: [L, D, D], b: [L, D]."""

    @staticmethod
    def forward(ctx, x0, W, b):
        x0 = x0.contiguous()
        L = W.shape[0]
        xs, us = [x0], []
        for l in range(L):
            u = torch.empty_like(x0)                     # U_l = x_l W_l^T + b_l, written by the GEMM epilogue
            xs.append(ops.gemm(xs[-1], W[l], transB=True, epi=ops.EPI_CROSS, bias=b[l], e0=x0, e1=xs[-1], aux=u))
            us.append(u)
        ctx.L = L
        ctx.save_for_backward(x0, W, b, *xs[:-1], *us)
        return xs[-1]

    @staticmethod
    def backward(ctx, gy):
        x0, W, b = ctx.saved_tensors[:3]
        L = ctx.L
        xs = ctx.saved_tensors[3:3 + L]
        us = ctx.saved_tensors[3 + L:]
        B, D = x0.shape
        g = gy.contiguous()
        gx0 = torch.empty_like(x0)
        dW = torch.empty_like(W)
        db = torch.empty_like(b)
        for l in range(L - 1, -1, -1):
            xl = xs[l]
            u = us[l]
            h = ops.crossnet_mat_bwd_elem(g, x0, u, gx0, accumulate=(l != L - 1))   # H = G(.)X0 ; dX0 += G(.)U
            ops.gemm(h, xl, transA=True, split_k=ops.split_k_for(B, D, D, True, False), out=dW[l])
            ops.colsum(h, out=db[l])
            g = ops.gemm(h, W[l], epi=ops.EPI_ADD, e1=g)                             # dX_l = G + H W
        if L == 0:
            return g, dW, db
        ops.axpby(1.0, g, 1.0, gx0)                                                  # x_0 is also layer 0's input
        return gx0, dW, db


class Cosine(torch.autograd.Function):
    """(1 + keras cosine_similarity(u, i))/2 = (1 - cos)/2  (2.FM/CustomLayers.py:233-234)."""

    @staticmethod
    def forward(ctx, u, i):
        u, i = u.contiguous(), i.contiguous()
        ctx.save_for_backward(u, i)
        return ops.cosine_fwd(u, i)

    @staticmethod
    def backward(ctx, g):
        u, i = ctx.saved_tensors
        return ops.cosine_bwd(u, i, g.contiguous())


class KerasBCE(torch.autograd.Function):
    """reduce_sum(BinaryCrossentropy()(y, p)) on probabilities (2.FM/ModelManager.py:100,175)."""

    @staticmethod
    def forward(ctx, p, y):
        loss, dp, _ = ops.bce_fwd_bwd(y, p.contiguous(), want_dp=True)
        ctx.save_for_backward(dp)
        ctx.shape = p.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dp,) = ctx.saved_tensors
        return (dp * g).reshape(ctx.shape), None


class Sigmoid(torch.autograd.Function):
    """tf.nn.sigmoid head (2.FM/CustomLayers.py:155,305)."""

    @staticmethod
    def forward(ctx, z, z2=None):
        """sigmoid(z + z2): z2 is the optional second addend (fm_part + dnn_part)."""
        y = ops.act_fwd(ops.ACT_SIGMOID, z.contiguous(), z2.contiguous().reshape(z.shape) if z2 is not None else None)
        ctx.save_for_backward(y)
        ctx.shape2 = z2.shape if z2 is not None else None
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        gz = ops.act_bwd(ops.ACT_SIGMOID, y, g.contiguous())
        return gz, (gz.reshape(ctx.shape2) if ctx.shape2 is not None else None)


# ---------------------------------------------------------------------------------------------------
# DIN (5.DIN/CustomLayers.py:142-289)
# ---------------------------------------------------------------------------------------------------

class FeatAct(torch.autograd.Function):
    """Per-feature activation on [M,N]: Dice (inference-mode BN statistics), PReLU or a named activation."""

    @staticmethod
    def forward(ctx, x, kind, alpha, mean, var):
        x = x.contiguous()
        ctx.kind = kind
        ctx.save_for_backward(x, alpha, mean, var)
        return ops.feat_act_fwd(kind, x, alpha, mean, var)

    @staticmethod
    def backward(ctx, gy):
        x, alpha, mean, var = ctx.saved_tensors
        want_a = alpha is not None and ctx.needs_input_grad[2]
        gx, ga = ops.feat_act_bwd(ctx.kind, x, gy.contiguous(), alpha, mean, var, want_alpha=want_a)
        return gx, None, (ops.colsum(ga) if want_a else None), None, None


class LayerNorm(torch.autograd.Function):
    """tf.keras.layers.LayerNormalization() over the last axis, epsilon 1e-3."""

    @staticmethod
    def forward(ctx, x, gamma, beta):
        y, xhat, rstd = ops.layernorm_fwd(x.contiguous(), gamma, beta)
        ctx.save_for_backward(xhat, rstd, gamma)
        return y

    @staticmethod
    def backward(ctx, gy):
        xhat, rstd, gamma = ctx.saved_tensors
        gy = gy.contiguous()
        gx, gg = ops.layernorm_bwd(gy, xhat, rstd, gamma)
        return gx, ops.colsum(gg), ops.colsum(gy)


class Softmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = ops.softmax_fwd(x.contiguous())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        return ops.softmax_bwd(y, gy.contiguous())


class DinAttention(torch.autograd.Function):
    """DinActivationLayer over every time step + mask + weighted sum pooling (5.DIN/CustomLayers.py:173-180,
    256-282), bilinear-factorised.  Inputs: table, q = flattened candidate-item embedding [B,D], series ids
    [B,T,C], the Dense(H) kernel W1 [3D+D*D,H] / bias b1, the activation's parameters, the Dense(1) kernel W2 [H,1]
    / bias b2.  Returns pooled [B,D] and the raw scores [B,T] (not differentiable)."""

    @staticmethod
    def forward(ctx, embed, q, series, W1, b1, kind, alpha, mean, var, W2, b2, padding_index, mask_valid, oob,
                sink=None):
        q = q.contiguous()
        B, D = q.shape
        H = W1.shape[1]
        Wcat, Wkd, bext = ops.din_prepare(W1.contiguous(), b1.contiguous(), D, H)
        Mext = ops.gemm(q, Wcat, epi=ops.EPI_BIAS, bias=bext)                 # [B, D*H + H] on the matrix cores
        w2 = W2.contiguous().reshape(-1)
        scores, pooled = ops.din_attn_fwd(embed, series, Mext, Wkd, kind, alpha, mean, var, w2, b2.contiguous(),
                                          padding_index, mask_valid, oob)
        ctx.save_for_backward(embed, q, series, Wcat, Wkd, Mext, alpha, mean, var, w2, b2, scores)
        ctx.cfg = (kind, padding_index, mask_valid, D, H)
        ctx.sink = sink
        ctx.mark_non_differentiable(scores)
        return pooled, scores

    @staticmethod
    def backward(ctx, gpooled, _gscores):
        embed, q, series, Wcat, Wkd, Mext, alpha, mean, var, w2, b2, scores = ctx.saved_tensors
        kind, padding_index, mask_valid, D, H = ctx.cfg
        V, E = embed.shape
        B = q.shape[0]
        sink, buf, dst = ctx.sink, None, None
        if sink is not None:                                 # values land behind the Gather's rows in one shared buffer
            n_ser = series.numel()
            buf = sink.new_buf(n_ser, E, embed.device)
            dst = buf[sink.n_head:sink.n_head + n_ser].view(series.shape[0], series.shape[1], D)
        gkeys, gMext, gw2p, galphap, gb2p = ops.din_attn_bwd(embed, series, Mext, Wkd, kind, alpha, mean, var, w2, b2,
                                                             padding_index, mask_valid, scores, gpooled.contiguous(),
                                                             gkeys=dst)
        gq = ops.gemm(gMext, Wcat, transB=True)                                          # [B,D]
        gWcat = ops.gemm(q, gMext, transA=True, split_k=ops.split_k_for(B, D, D * H + H, True, False))  # [D, D*H+H]
        gbext = ops.colsum(gMext)
        gWkd = gbext[:D * H].reshape(D, H)                    # Eff_b = Wkd + M_b  =>  dWkd = sum_b dEff_b: the leading
                                                              # D*H column sums of gMext, already in gbext
        gW1 = ops.din_prepare_bwd(gWcat, gWkd, D, H)
        gb1 = gbext[D * H:].clone()
        gW2 = ops.colsum(gw2p).reshape(H, 1)
        galpha = ops.colsum(galphap) if alpha is not None else None
        gb2 = ops.colsum(gb2p)
        if sink is not None:
            sink.ids, sink.buf = series, buf
            gembed = None
        else:
            plan = ops.DedupPlan(series, V)
            gembed = _sparse_grad(plan, gkeys.reshape(-1, E), E, (V, E))
        return gembed, gq, None, gW1, gb1, None, galpha, None, None, gW2, gb2, None, None, None, None


class EmbIpn(torch.autograd.Function):
    """PNN inner-product front end, fused with the lookup: X [B,F] -> [Flatten(embed(X)) | <e_i,e_j>, i<j]
    (2.FM/CustomLayers.py:737-745 with IpnLayer :773-792).  Backward: IndexedSlices values from the saved output."""

    @staticmethod
    def forward(ctx, table, X, oob):
        out = ops.emb_ipn_fwd(table, X, oob)
        ctx.save_for_backward(X, out)
        ctx.shape = tuple(table.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        X, out = ctx.saved_tensors
        V, E = ctx.shape
        vals = ops.emb_ipn_bwd_vals(out, g.contiguous(), X.shape[1], E)
        plan = ops.DedupPlan(X, V)
        return _sparse_grad(plan, vals, E, (V, E)), None, None


class EmbBiInteraction(torch.autograd.Function):
    """NFM bi-interaction pooling fused with the lookup, written into the leading columns of
    [second_order | X_cont] (3.DCN/CustomLayers.py:493-503): returns that combined matrix."""

    @staticmethod
    def forward(ctx, table, X, cont, oob):
        V, E = table.shape
        B = X.shape[0]
        nc = 0 if cont is None else cont.shape[1]
        comb = torch.empty((B, E + nc), dtype=torch.float32, device=table.device)
        _, S = ops.emb_bi_fwd(table, X, comb, oob)
        if nc:
            comb[:, E:] = cont
        ctx.save_for_backward(table, X, S)
        ctx.nc = nc
        return comb

    @staticmethod
    def backward(ctx, g):
        table, X, S = ctx.saved_tensors
        V, E = table.shape
        g = g.contiguous()
        vals = ops.emb_bi_bwd_vals(table, X, g, S)
        plan = ops.DedupPlan(X, V)
        gcont = g[:, E:].contiguous() if ctx.nc and ctx.needs_input_grad[2] else None
        return _sparse_grad(plan, vals, E, (V, E)), None, gcont, None


class IpAttention(torch.autograd.Function):
    """GSU inner-product attention + sum pooling over the embedded behaviour series, fused with the series lookup
    (7.SIM/CustomLayers.py:88-96,107-118).  Returns pooled [B,D] and the masked scores [B,T]."""

    @staticmethod
    def forward(ctx, embed, q, series, padding_index, oob, sink=None):
        q = q.contiguous()
        scores, pooled = ops.ip_attn_fwd(embed, series, q, padding_index, oob)
        ctx.save_for_backward(embed, q, series, scores)
        ctx.padding_index = padding_index
        ctx.sink = sink
        ctx.mark_non_differentiable(scores)
        return pooled, scores

    @staticmethod
    def backward(ctx, gpooled, _gscores):
        embed, q, series, scores = ctx.saved_tensors
        V, E = embed.shape
        sink, dst = ctx.sink, None
        if sink is not None:
            buf = sink.new_buf(series.numel(), E, embed.device)
            dst = buf[sink.n_head:sink.n_head + series.numel()].view(series.shape[0], series.shape[1], -1)
        gkeys, gq = ops.ip_attn_bwd(embed, series, q, ctx.padding_index, scores, gpooled.contiguous(), gkeys=dst)
        if sink is not None:
            sink.ids, sink.buf = series, buf
            return None, gq, None, None, None, None
        plan = ops.DedupPlan(series, V)
        return _sparse_grad(plan, gkeys.reshape(-1, E), E, (V, E)), gq, None, None, None, None


class BatchNorm(torch.autograd.Function):
    """tf.keras.layers.BatchNormalization on [B,N] (training: batch statistics, moving averages updated in place)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, moving_mean, moving_var, training, eps, momentum):
        y, xhat, rstd = ops.batchnorm_fwd(x.contiguous(), gamma, beta, moving_mean, moving_var, training, eps, momentum)
        ctx.save_for_backward(xhat, rstd, gamma)
        ctx.training = training
        return y

    @staticmethod
    def backward(ctx, g):
        xhat, rstd, gamma = ctx.saved_tensors
        gx, ggamma, gbeta = ops.batchnorm_bwd(g.contiguous(), xhat, rstd, gamma, ctx.training)
        return gx, ggamma, gbeta, None, None, None, None, None


class FFM(torch.autograd.Function):
    """FFM logit: bias + first order + field-aware second order, fused with the lookups
    (2.FM/CustomLayers.py:398-425 / 428-462, 480-494).  v [V,F,E], w [V,1], bias [1], X [B,F] -> z [B]."""

    @staticmethod
    def forward(ctx, v, w, bias, X, oob):
        z, _ = ops.ffm_fwd(v, w, bias, X, oob=oob)
        ctx.save_for_backward(v, X)
        return z

    @staticmethod
    def backward(ctx, gz):
        v, X = ctx.saved_tensors
        V, F, E = v.shape
        B = X.shape[0]
        gz = gz.contiguous()
        plan = ops.DedupPlan(X, V)
        rows = ops.ffm_bwd_rows(v, X, gz, plan)
        g_v = SparseRowGrad(plan.uniq_ids, rows, plan.n_uniq, (V, F, E)).to_sparse()
        g_w = _sparse_grad(plan, gz.reshape(B, 1), 1, (V, 1), row_div=F)
        return g_v, g_w, ops.colsum(gz.reshape(B, 1)), None, None
