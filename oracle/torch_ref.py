"""TEST INFRASTRUCTURE ONLY -- op-for-op torch-CPU restatement of the reference hot path.

Second, independent restatement (the first is ``layers_np``): the same op ORDER as the TF source
(gather -> square -> reduce_sum(axis=1) ...), eager torch ops on the CPU, gradients by autograd.  It is
(1) the cross-check for ``layers_np``'s hand-derived backward, (2) the gradient oracle for the HIP
kernels and (3) the "TF2-CPU stand-in" that ``bench.py`` times as ``cpu_baseline`` (kind "port";
TensorFlow is not installed here or on the GPU box).  Never imported by the product package.

All functions take a dict ``p`` of tensors (same keys as ``layers_np``) and int64 index tensors.
"""
import math

import torch
import torch.nn.functional as F

KERAS_EPS = 1e-7
BN_EPS = 1e-3
LN_EPS = 1e-3


def index_assemble(inputs, feature_names):
    """2.FM/CustomLayers.py:138-144."""
    cols = []
    for name in feature_names:
        t = inputs[name]
        if t.dim() == 1:
            t = t.unsqueeze(1)
        cols.append(t)
    return torch.cat(cols, dim=1)


def lookup(table, X, sparse=False):
    """Keras Embedding -> gather (2.FM/CustomLayers.py:146-147)."""
    if X.numel() and (int(X.min()) < 0 or int(X.max()) >= table.shape[0]):
        raise IndexError("embedding id out of range")
    return F.embedding(X, table, sparse=sparse)


def _act(name, x):
    if name is None:
        return x
    return {"relu": torch.relu, "sigmoid": torch.sigmoid, "tanh": torch.tanh}[name](x)


def mlp(x, kernels, biases, activation):
    """MLPLayer.call, 2.FM/CustomLayers.py:72-84."""
    for K, b in zip(kernels, biases):
        x = x @ K
        if b is not None:
            x = x + b
        x = _act(activation, x)
    return x


def fm_logit(p, X, sparse=False):
    """2.FM/CustomLayers.py:146-155 without the final sigmoid."""
    w_output = lookup(p["w"], X, sparse)
    emb_output = lookup(p["embed"], X, sparse)
    first_order = torch.sum(w_output, dim=1)
    sum_of_square = torch.sum(torch.square(emb_output), dim=1)
    square_of_sum = torch.square(torch.sum(emb_output, dim=1))
    second_order = 0.5 * torch.sum(square_of_sum - sum_of_square, dim=1, keepdim=True)
    return p["bias"] + first_order + second_order


def fm_forward(p, X, sparse=False):
    return torch.sigmoid(fm_logit(p, X, sparse))


def deepfm_logit(p, X, sparse=False):
    """2.FM/CustomLayers.py:289-305 without the final sigmoid."""
    w_output = lookup(p["w"], X, sparse)
    emb_output = lookup(p["embed"], X, sparse)
    first_order = torch.sum(w_output, dim=1) + p["bias"]
    sum_of_square = torch.sum(torch.square(emb_output), dim=1)
    square_of_sum = torch.square(torch.sum(emb_output, dim=1))
    second_order = 0.5 * torch.sum(square_of_sum - sum_of_square, dim=1, keepdim=True)
    fm_part = first_order + second_order
    dense_embedding = emb_output.flatten(1)
    dnn_part = mlp(mlp(dense_embedding, p["k1"], p["b1"], "relu"), p["k2"], p["b2"], None)
    return fm_part + dnn_part


def deepfm_forward(p, X, sparse=False):
    return torch.sigmoid(deepfm_logit(p, X, sparse))


def dssm_tower(p, X, sparse=False):
    """2.FM/CustomLayers.py:196-201."""
    x = lookup(p["embed"], X, sparse).flatten(1)
    x = mlp(x, p["mlp_k"], p["mlp_b"], "relu")
    return mlp(x, p["final_k"], p["final_b"], None)


def two_tower_score(u, i):
    """2.FM/CustomLayers.py:233-234; keras cosine_similarity = -sum(l2norm(u)*l2norm(i))."""
    un = u * torch.rsqrt(torch.clamp(torch.sum(u * u, dim=1, keepdim=True), min=1e-12))
    vn = i * torch.rsqrt(torch.clamp(torch.sum(i * i, dim=1, keepdim=True), min=1e-12))
    similarity = -torch.sum(un * vn, dim=1)
    return (1 + similarity) / 2


def cross_vec(x0, ws, bs):
    """CrossLayer.call, 3.DCN/CustomLayers.py:195-203 (batched-matmul form kept)."""
    x0c = x0.unsqueeze(2)
    xl = x0c
    for w, b in zip(ws, bs):
        xl_w = torch.matmul(xl.transpose(1, 2), w)          # [B,1,1]
        xl = torch.matmul(x0c, xl_w) + b + xl
    return xl.squeeze(2)


def cross_mat(x0, Ws, bs, row_form=False):
    """MatrixCrossLayer.call, 3.DCN/CustomLayers.py:297-305.  ``row_form``: the same numbers as U = X W^T on [B,D] rows
     -- for config-size batches, where the literal broadcast matmul would expand W to [B,D,D]
    (13.7 GB in fp64 at B = 16384, D = 323); tests/test_oracle.py holds the two forms equal."""
    if row_form:
        xl = x0
        for W, b in zip(Ws, bs):
            xl = x0 * (xl @ W.t() + b.reshape(1, -1)) + xl
        return xl
    x0c = x0.unsqueeze(2)
    xl = x0c
    for W, b in zip(Ws, bs):
        xl_w = torch.matmul(W, xl)                          # [B,D,1]
        xl = x0c * (xl_w + b) + xl
    return xl.squeeze(2)


def dcn_forward(p, X, X_cont, kind="vec", sparse=False, row_form=False):
    """DeepCrossNetworkLayer.call, 3.DCN/CustomLayers.py:239-269."""
    x_flat = lookup(p["embed"], X, sparse).flatten(1)
    _input = torch.cat([X_cont, x_flat], dim=1)
    cross = cross_vec(_input, p["cross_w"], p["cross_b"]) if kind == "vec" else \
        cross_mat(_input, p["cross_w"], p["cross_b"], row_form)
    dnn = mlp(_input, p["dnn_k"], p["dnn_b"], "relu")
    comb = torch.cat([cross, dnn], dim=1)
    return torch.sigmoid(comb @ p["out_k"] + p["out_b"])


def _din_act(act, x):
    kind = act["kind"]
    if kind == "dice":                                      # 5.DIN/CustomLayers.py:193-196
        xn = (x - act["mean"]) / torch.sqrt(act["var"] + BN_EPS)
        xp = torch.sigmoid(xn)
        return act["alpha"] * (1.0 - xp) * x + xp * x
    if kind == "prelu":
        return torch.relu(x) + act["alpha"] * (-torch.relu(-x))
    return _act(kind, x)


def din_activation_unit(q, k, a):
    """DinActivationLayer.call literally (5.DIN/CustomLayers.py:173-180): materialises the
    [B, 3D + D*D] concat.  q, k: [B,D] -> [B,1]."""
    diff = q - k
    outer = q.unsqueeze(1) * k.unsqueeze(2)
    x = torch.cat([q, diff, k, outer.reshape(q.shape[0], -1)], dim=1)
    h = _din_act(a["act"], x @ a["W1"] + a["b1"])
    return h @ a["W2"] + a["b2"]


def din_forward(p, profile_ids, item_ids, series_ids, padding_index=0, mask_mode="reference",
                sparse=False):
    """DINLayer.call, 5.DIN/CustomLayers.py:229-289 (vectorized_map over T written as a loop)."""
    B, T, C = series_ids.shape
    E = p["embed"].shape[1]
    profile = lookup(p["embed"], profile_ids, sparse).flatten(1)
    q = lookup(p["embed"], item_ids, sparse).flatten(1)
    pad = series_ids[:, :, 0] == padding_index
    mask = (pad if mask_mode == "reference" else ~pad).to(q.dtype)
    keys = lookup(p["embed"], series_ids.reshape(B, T * C), sparse).reshape(B, T, C * E)
    scores = torch.stack([din_activation_unit(q, keys[:, t, :], p["att"]) for t in range(T)], dim=1)
    scores_masked = scores * mask.unsqueeze(-1)             # [B,T,1]
    pooled = torch.sum(keys * scores_masked, dim=1)
    x = torch.cat([profile, pooled], dim=1)
    for lyr in p["mlp"]:
        x = x @ lyr["K"] + lyr["b"]
        x = F.layer_norm(x, (x.shape[-1],), lyr["gamma"], lyr["beta"], eps=LN_EPS)
        x = _din_act(lyr["act"], x)
    return torch.softmax(x @ p["out_k"] + p["out_b"], dim=-1), scores.squeeze(-1), pooled


def l2_used_rows(table, ids, factor):
    """5.DIN/ModelManager.py:176-190: tf.unique over every id of the batch, tf.gather, tf.nn.l2_loss * factor."""
    uniq = torch.unique(ids.reshape(-1))
    used = table[uniq]
    return factor * 0.5 * torch.sum(used * used)


def keras_bce(y, prob):
    """reduce_sum(BinaryCrossentropy()(y, prob)), 2.FM/ModelManager.py:100,175."""
    if prob.dim() == 1 and y.dim() == 2 and y.shape[1] == 1:
        y = y[:, 0]
    pc = torch.clamp(prob, KERAS_EPS, 1.0 - KERAS_EPS)
    bce = -(y * torch.log(pc + KERAS_EPS) + (1.0 - y) * torch.log(1.0 - pc + KERAS_EPS))
    if bce.dim() == 1:
        return bce.mean()
    return bce.mean(dim=-1).mean()


class KerasAdam:
    """Keras Adam (2.FM/ModelManager.py:104): dense apply for dense grads; for sparse grads the
    Keras path decays m and v on ALL rows and updates ALL rows (dense sweep)."""

    def __init__(self, params, lr=1e-3, b1=0.9, b2=0.999, eps=KERAS_EPS):
        self.params = list(params)
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.m = [torch.zeros_like(q) for q in self.params]
        self.v = [torch.zeros_like(q) for q in self.params]
        self.t = 0

    @torch.no_grad()
    def step(self):
        self.t += 1
        lr_t = self.lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        for q, m, v in zip(self.params, self.m, self.v):
            g = q.grad
            if g is None:
                continue
            if g.is_sparse:
                g = g.coalesce()
                ids, vals = g.indices()[0], g.values()
                m.mul_(self.b1)
                m.index_add_(0, ids, vals * (1 - self.b1))
                v.mul_(self.b2)
                v.index_add_(0, ids, vals * vals * (1 - self.b2))
            else:
                m.add_((g - m) * (1 - self.b1))
                v.add_((g * g - v) * (1 - self.b2))
            q.sub_(lr_t * m / (torch.sqrt(v) + self.eps))
            q.grad = None


# ---------------------------------------------------------------------------------------------------
# sibling interaction layers, op for op as the TF source writes them
# ---------------------------------------------------------------------------------------------------

def shared_fields_interaction(x):
    """SharedFieldsInteraction (2.FM/CustomLayers.py:755-771): x1 * x2 broadcast to [B,F,F,E], strict upper
    triangle kept by a boolean mask in row-major order -> [B, F(F-1)/2, E]."""
    B, Fn, E = x.shape
    inter = x.unsqueeze(1) * x.unsqueeze(2)
    mask = torch.triu(torch.ones(Fn, Fn), diagonal=1) > 0
    return inter[:, mask, :]


def ipn(x):
    """IpnLayer.call (:788-792): reduce_sum over the embedding axis."""
    return shared_fields_interaction(x).sum(dim=2)


def pnn_combined(table, X, sparse=False):
    """PNNLayer.call (:737-745): emb -> [Flatten(emb) | IpnLayer(emb)]."""
    emb = lookup(table, X, sparse)
    return torch.cat([emb.reshape(emb.shape[0], -1), ipn(emb)], dim=1)


def pnn_forward(p, X, sparse=False):
    comb = pnn_combined(p["embed"], X, sparse)
    h = mlp(comb, p["k1"], p["b1"], "relu")
    return mlp(h, p["k2"], p["b2"], "sigmoid")


def bi_interaction(table, X, sparse=False):
    """3.DCN/CustomLayers.py:499-501."""
    emb = lookup(table, X, sparse)
    sum_of_square = torch.sum(torch.square(emb), dim=1)
    square_of_sum = torch.square(torch.sum(emb, dim=1))
    return 0.5 * (square_of_sum - sum_of_square)


def batchnorm(x, gamma, beta, mov_mean, mov_var, training, eps=BN_EPS):
    """Keras BatchNormalization (non-fused, [B,N]); the moving-average update is not modelled here (see layers_np)."""
    if training:
        mean = x.mean(dim=0)
        var = torch.square(x - mean).mean(dim=0)
    else:
        mean, var = mov_mean, mov_var
    return (x - mean) * torch.rsqrt(var + eps) * gamma + beta


def nfm_forward(p, X, X_cont, training=True, sparse=False):
    """NeuralFactorizationMachineLayer.call (3.DCN/CustomLayers.py:476-509)."""
    comb = torch.cat([bi_interaction(p["embed"], X, sparse), X_cont], dim=1)
    comb = batchnorm(comb, p["bn_gamma"], p["bn_beta"], p["bn_mean"], p["bn_var"], training)
    h = mlp(comb, p["k1"], p["b1"], p.get("activation", "relu"))
    return mlp(h, p["k2"], p["b2"], "sigmoid")


def ip_attention(table, q, series, padding_index=0, sparse=False):
    """GSULayer (7.SIM/CustomLayers.py:88-96,107-118): embedded series [B,T,C*E], einsum scores, valid mask,
    einsum pooling.  -> (masked scores, pooled)."""
    B, T, C = series.shape
    k = lookup(table, series.reshape(B, T * C), sparse).reshape(B, T, -1)
    valid = (series[:, :, 0] != padding_index).to(torch.float32)
    scores = torch.einsum("be,ble->bl", q, k) * valid
    pooled = torch.einsum("bl,ble->be", scores, k)
    return scores, pooled


def gsu_combined(p, item_ids, series, padding_index=0, sparse=False):
    """GSULayer.call up to X_combined = [Flatten(embed(item ids)) | pooled] (7.SIM/CustomLayers.py:98-122); the MLP
    that follows is the same make_mlp_layer stack din_forward restates."""
    q = lookup(p["embed"], item_ids, sparse)
    q = q.reshape(q.shape[0], -1)
    _, pooled = ip_attention(p["embed"], q, series, padding_index, sparse)
    return torch.cat([q, pooled], dim=1)


def field_aware_interaction(v, X, sparse=False):
    """FieldAwareInteractionLayer.call (2.FM/CustomLayers.py:436-462): embedding_lookup -> [B,F,F,E], multiply with
    its transpose over the two field axes, keep the strict upper triangle -> [B, F(F-1)/2, E]."""
    B, Fn = X.shape
    V, _, E = v.shape
    emb = lookup(v.reshape(V, Fn * E), X, sparse).reshape(B, Fn, Fn, E)
    inter = emb * emb.transpose(1, 2)
    mask = torch.triu(torch.ones(Fn, Fn), diagonal=1) > 0
    return inter[:, mask, :]


def ffm_logit(p, X, sparse=False):
    """FFMLayer.call (:480-494).  p: v [V,F,E], w [V,1], bias [1]."""
    linear = lookup(p["w"], X, sparse).sum(dim=1)
    inter = field_aware_interaction(p["v"], X, sparse).sum(dim=1).sum(dim=1, keepdim=True)
    return p["bias"] + linear + inter


def ffm_forward(p, X, sparse=False):
    return torch.sigmoid(ffm_logit(p, X, sparse))
