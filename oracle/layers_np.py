"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference hot path.

Every function cites the reference lines it follows (paths relative to /root/reference).  ``dt``
selects the arithmetic type: ``np.float32`` restates the reference op-for-op in its own precision,
``np.float64`` is the high-precision checker used for tolerance tests.  Backward formulas are derived
by hand and cross-checked against ``torch_ref`` autograd in tests/test_oracle.py.

TF/Keras semantics not visible in the reference source are restated from their public definitions:
Keras ``cosine_similarity`` is the *negative* cosine; Keras ``BinaryCrossentropy`` clips probabilities
to [1e-7, 1-1e-7] and adds 1e-7 inside the logs; Keras ``Adam`` (epsilon 1e-7) applies sparse
gradients with a dense decay of m and v; LayerNormalization / BatchNormalization epsilon = 1e-3.
"""
import numpy as np

# --------------------------------------------------------------------------------------------
# a1  index assembly            2.FM/CustomLayers.py:138-144 (same idiom 3.DCN:240-246, 5.DIN:236-242)
# --------------------------------------------------------------------------------------------

def index_assemble(inputs, feature_names):
    """Per feature: rank-1 -> expand_dims(axis=1); concat(axis=1).  int64 [B,F], bit exact."""
    cols = []
    for name in feature_names:
        t = np.asarray(inputs[name])
        if t.ndim == 1:
            t = t[:, None]
        cols.append(t.astype(np.int64))
    return np.concatenate(cols, axis=1)


# --------------------------------------------------------------------------------------------
# a2  embedding lookup          2.FM/CustomLayers.py:129-134,146-147 (Keras Embedding -> gather)
# --------------------------------------------------------------------------------------------

def embedding_lookup(table, X):
    """out[..., :] = table[X[...], :].  Out-of-range ids raise (TF CPU: InvalidArgumentError)."""
    X = np.asarray(X)
    if X.size and (X.min() < 0 or X.max() >= table.shape[0]):
        raise IndexError("embedding id out of range [0, %d)" % table.shape[0])
    return table[X]


# --------------------------------------------------------------------------------------------
# activations
# --------------------------------------------------------------------------------------------

def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def _act(name, x):
    if name is None or name == "linear":
        return x
    if name == "relu":
        return np.maximum(x, 0)
    if name == "sigmoid":
        return sigmoid(x)
    if name == "tanh":
        return np.tanh(x)
    raise ValueError("unknown activation %r" % (name,))


def _act_grad(name, pre, post, g):
    if name is None or name == "linear":
        return g
    if name == "relu":
        return g * (pre > 0)
    if name == "sigmoid":
        return g * post * (1 - post)
    if name == "tanh":
        return g * (1 - post * post)
    raise ValueError(name)


# --------------------------------------------------------------------------------------------
# a4  MLPLayer                  2.FM/CustomLayers.py:72-84  (activation on EVERY layer, :80-81)
# --------------------------------------------------------------------------------------------

def mlp_forward(x, kernels, biases, activation, dt=np.float32, keep=False):
    """x @ K_i + b_i -> activation, for every layer i.  Returns y (and the saved tensors)."""
    saved = []
    h = x.astype(dt)
    for K, b in zip(kernels, biases):
        pre = h @ K.astype(dt)
        if b is not None:
            pre = pre + b.astype(dt)
        post = _act(activation, pre)
        saved.append((h, pre, post))
        h = post
    return (h, saved) if keep else h


def mlp_backward(saved, kernels, activation, gy, dt=np.float32):
    """Returns (gx, [dK_i], [db_i])."""
    dKs, dbs = [], []
    g = gy.astype(dt)
    for (h, pre, post), K in zip(reversed(saved), reversed(kernels)):
        g = _act_grad(activation, pre, post, g)
        dKs.append(h.T @ g)
        dbs.append(g.sum(axis=0))
        g = g @ K.astype(dt).T
    return g, dKs[::-1], dbs[::-1]


# --------------------------------------------------------------------------------------------
# a3  FMRankingLayer            2.FM/CustomLayers.py:137-157
# --------------------------------------------------------------------------------------------

def fm_terms(embed, w, X, dt=np.float32):
    """first_order [B,1], S=sum_f e [B,E], second_order [B,1], e [B,F,E]  (:146-153)."""
    e = embedding_lookup(embed, X).astype(dt)                  # [B,F,E]
    wv = embedding_lookup(w, X).astype(dt)                     # [B,F,1]
    first = wv.sum(axis=1)                                     # reduce_sum(w_output, axis=1)
    sum_of_square = np.square(e).sum(axis=1)                   # [B,E]
    S = e.sum(axis=1)
    square_of_sum = np.square(S)
    second = dt(0.5) * (square_of_sum - sum_of_square).sum(axis=1, keepdims=True)
    return first, S, second, e


def fm_forward(embed, w, bias, X, dt=np.float32):
    """sigmoid(bias + first + second) -> [B,1]  (:155)."""
    first, S, second, _ = fm_terms(embed, w, X, dt)
    z = bias.astype(dt) + first + second
    return sigmoid(z), z


def fm_backward(embed, X, gz, dt=np.float32):
    """Given g_z [B,1]: IndexedSlices values for embed ([B*F,E]), for w ([B*F,1]) and dbias [1].

    d z/d e[b,f,:] = S[b,:] - e[b,f,:];  d z/d w = 1;  d z/d bias = 1   (SURVEY.md section 9).
    """
    e = embedding_lookup(embed, X).astype(dt)
    S = e.sum(axis=1, keepdims=True)
    g = gz.astype(dt).reshape(-1, 1, 1)
    demb = g * (S - e)
    dw = np.broadcast_to(g, e.shape[:2] + (1,))
    return demb.reshape(-1, e.shape[2]), dw.reshape(-1, 1).copy(), gz.astype(dt).sum().reshape(1)


# --------------------------------------------------------------------------------------------
# DeepFMRankingLayer            2.FM/CustomLayers.py:279-308
# --------------------------------------------------------------------------------------------

def deepfm_forward(p, X, dt=np.float32, keep=False):
    """p: dict embed,w,bias,k1 (list),b1 (list),k2 (list),b2 (list).  Returns prob [B,1], z."""
    first, S, second, e = fm_terms(p["embed"], p["w"], X, dt)
    fm_part = (first + p["bias"].astype(dt)) + second                        # :293,:297
    dense = e.reshape(e.shape[0], -1)                                        # Flatten :300
    h1, s1 = mlp_forward(dense, p["k1"], p["b1"], "relu", dt, keep=True)     # MLP_layer1 relu
    h2, s2 = mlp_forward(h1, p["k2"], p["b2"], None, dt, keep=True)          # MLP_layer2 linear
    z = fm_part + h2
    out = sigmoid(z)
    if keep:
        return out, z, (e, S, s1, s2)
    return out, z


def deepfm_backward(p, X, gz, dt=np.float32):
    """Gradients of sum(gz*z) w.r.t. every parameter.  Embedding grads as IndexedSlices values."""
    _, _, (e, S, s1, s2) = deepfm_forward(p, X, dt, keep=True)
    B, F, E = e.shape
    g = gz.astype(dt).reshape(B, 1)
    g1, dk2, db2 = mlp_backward(s2, p["k2"], None, g, dt)
    gx, dk1, db1 = mlp_backward(s1, p["k1"], "relu", g1, dt)
    demb = g.reshape(B, 1, 1) * (S[:, None, :] - e) + gx.reshape(B, F, E)
    dw = np.broadcast_to(g.reshape(B, 1, 1), (B, F, 1)).reshape(-1, 1).copy()
    return {"embed_values": demb.reshape(-1, E), "w_values": dw, "bias": g.sum().reshape(1),
            "k1": dk1, "b1": db1, "k2": dk2, "b2": db2}


# --------------------------------------------------------------------------------------------
# a5/a6  DSSM towers            2.FM/CustomLayers.py:183-206, 230-239
# --------------------------------------------------------------------------------------------

def dssm_tower_forward(p, X, dt=np.float32):
    """gather -> Flatten (field-major, dim-minor) -> MLP(relu) -> MLP(final, linear)  (:196-201)."""
    e = embedding_lookup(p["embed"], X).astype(dt)
    x = e.reshape(e.shape[0], -1)
    x = mlp_forward(x, p["mlp_k"], p["mlp_b"], "relu", dt)
    return mlp_forward(x, p["final_k"], p["final_b"], None, dt)


def l2_normalize(x, axis, dt=np.float32):
    """tf.math.l2_normalize: x * rsqrt(max(sum(x^2), 1e-12))."""
    sq = np.square(x).sum(axis=axis, keepdims=True)
    return x / np.sqrt(np.maximum(sq, dt(1e-12)))


def two_tower_score(u, i, dt=np.float32):
    """(1 + keras.losses.cosine_similarity(u, i, axis=1)) / 2 = (1 - cos)/2, shape [B]  (:233-234)."""
    sim = -(l2_normalize(u.astype(dt), 1, dt) * l2_normalize(i.astype(dt), 1, dt)).sum(axis=1)
    return (dt(1) + sim) / dt(2)


def two_tower_score_backward(u, i, gout, dt=np.float32):
    """d out/d u, d out/d i for out=(1-cos)/2: dc/du = (i_hat - c*u_hat)/|u| (|u|^2 > 1e-12)."""
    u = u.astype(dt)
    i = i.astype(dt)
    nu = np.sqrt(np.maximum(np.square(u).sum(1, keepdims=True), dt(1e-12)))
    ni = np.sqrt(np.maximum(np.square(i).sum(1, keepdims=True), dt(1e-12)))
    uh, ih = u / nu, i / ni
    c = (uh * ih).sum(1, keepdims=True)
    gc = -dt(0.5) * gout.astype(dt).reshape(-1, 1)
    return gc * (ih - c * uh) / nu, gc * (uh - c * ih) / ni


# --------------------------------------------------------------------------------------------
# a7  CrossLayer (vector)       3.DCN/CustomLayers.py:195-203
# --------------------------------------------------------------------------------------------

def cross_vec_forward(x0, ws, bs, dt=np.float32, keep=False):
    """x_{l+1} = x0 * (x_l . w_l) + b_l + x_l.   ws[l], bs[l]: [D,1] as in the reference."""
    x0 = x0.astype(dt)
    xl = x0
    xs = []
    for w, b in zip(ws, bs):
        xs.append(xl)
        s = xl @ w.astype(dt)                       # [B,1]   (matmul(xl^T, w) :199)
        xl = x0 * s + b.astype(dt).reshape(1, -1) + xl   # :200-201
    return (xl, xs) if keep else xl


def cross_vec_backward(x0, ws, bs, gy, dt=np.float32):
    """Returns (gx0, [dw_l], [db_l])."""
    _, xs = cross_vec_forward(x0, ws, bs, dt, keep=True)
    x0 = x0.astype(dt)
    g = gy.astype(dt)
    gx0 = np.zeros_like(x0)
    dws, dbs = [], []
    for xl, w in zip(reversed(xs), reversed(ws)):
        w = w.astype(dt)
        s = xl @ w                                  # [B,1]
        t = (g * x0).sum(axis=1, keepdims=True)     # [B,1]
        dws.append(xl.T @ t)                        # [D,1]
        dbs.append(g.sum(axis=0).reshape(-1, 1))
        gx0 = gx0 + g * s
        g = g + t * w.reshape(1, -1)
    return gx0 + g, dws[::-1], dbs[::-1]


# --------------------------------------------------------------------------------------------
# a8  MatrixCrossLayer          3.DCN/CustomLayers.py:297-305
# --------------------------------------------------------------------------------------------

def cross_mat_forward(x0, Ws, bs, dt=np.float32, keep=False):
    """x_{l+1} = x0 (.) (W_l x_l + b_l) + x_l;  row form U = X_l W_l^T + b_l^T."""
    x0 = x0.astype(dt)
    xl = x0
    saved = []
    for W, b in zip(Ws, bs):
        u = xl @ W.astype(dt).T + b.astype(dt).reshape(1, -1)   # :301-302
        saved.append((xl, u))
        xl = x0 * u + xl                                         # :302-303
    return (xl, saved) if keep else xl


def cross_mat_backward(x0, Ws, bs, gy, dt=np.float32):
    """Returns (gx0, [dW_l], [db_l]).  H = G (.) X0; dW = H^T X_l; dX_l = G + H W; dX0 += G (.) U."""
    _, saved = cross_mat_forward(x0, Ws, bs, dt, keep=True)
    x0 = x0.astype(dt)
    g = gy.astype(dt)
    gx0 = np.zeros_like(x0)
    dWs, dbs = [], []
    for (xl, u), W in zip(reversed(saved), reversed(Ws)):
        h = g * x0
        dWs.append(h.T @ xl)
        dbs.append(h.sum(axis=0).reshape(-1, 1))
        gx0 = gx0 + g * u
        g = g + h @ W.astype(dt)
    return gx0 + g, dWs[::-1], dbs[::-1]


# --------------------------------------------------------------------------------------------
# a9  DeepCrossNetworkLayer     3.DCN/CustomLayers.py:239-269
# --------------------------------------------------------------------------------------------

def dcn_forward(p, X, X_cont, kind="vec", dt=np.float32):
    """_input=[cont | flatten(embed(X))] (:259); cross; Dense stack relu (:230,:263);
    Dense(1, sigmoid) over [cross | dnn] (:265-267).  p: embed, cross_w, cross_b, dnn_k, dnn_b, out_k, out_b."""
    e = embedding_lookup(p["embed"], X).astype(dt)
    x0 = np.concatenate([X_cont.astype(dt), e.reshape(e.shape[0], -1)], axis=1)
    if kind == "vec":
        cross = cross_vec_forward(x0, p["cross_w"], p["cross_b"], dt)
    else:
        cross = cross_mat_forward(x0, p["cross_w"], p["cross_b"], dt)
    dnn = mlp_forward(x0, p["dnn_k"], p["dnn_b"], "relu", dt)
    comb = np.concatenate([cross, dnn], axis=1)
    z = comb @ p["out_k"].astype(dt) + p["out_b"].astype(dt)
    return sigmoid(z), z


# --------------------------------------------------------------------------------------------
# a10/a11  DIN                  5.DIN/CustomLayers.py:142-289
# --------------------------------------------------------------------------------------------

BN_EPS = 1e-3      # keras BatchNormalization default epsilon
LN_EPS = 1e-3      # keras LayerNormalization default epsilon


def dice(x, alpha, mov_mean, mov_var, dt=np.float32):
    """Dice (:183-196): p = sigmoid(BN(x)) with BN(center=False, scale=False) in inference mode
    (moving statistics; the oracle is pinned to inference-mode BN);
    out = alpha*(1-p)*x + p*x."""
    xn = (x - mov_mean.astype(dt)) / np.sqrt(mov_var.astype(dt) + dt(BN_EPS))
    p = sigmoid(xn)
    return alpha.astype(dt) * (dt(1) - p) * x + p * x


def prelu(x, alpha, dt=np.float32):
    """keras PReLU: max(0,x) + alpha*min(0,x)."""
    return np.maximum(x, 0) + alpha.astype(dt) * np.minimum(x, 0)


def _din_act(act, x, dt):
    kind = act["kind"]
    if kind == "dice":
        return dice(x, act["alpha"], act["mean"], act["var"], dt)
    if kind == "prelu":
        return prelu(x, act["alpha"], dt)
    return _act(kind, x)


def din_activation_unit_naive(q, k, W1, b1, act, W2, b2, dt=np.float32):
    """DinActivationLayer.call (:173-180), literally: concat([q, q-k, k, vec(k q^T)]) -> Dense(36)
    -> act -> Dense(1).  q,k: [B,D]; outer[b,i,j] = k[b,i]*q[b,j] flattened row-major (:176-177)."""
    q = q.astype(dt)
    k = k.astype(dt)
    outer = q[:, None, :] * k[:, :, None]            # expand_dims(vec1,1)*expand_dims(vec2,2)
    x = np.concatenate([q, q - k, k, outer.reshape(q.shape[0], -1)], axis=1)
    h = _din_act(act, x @ W1.astype(dt) + b1.astype(dt), dt)
    return h @ W2.astype(dt) + b2.astype(dt)         # [B,1]


def din_activation_unit_factorised(q, keys, W1, b1, act, W2, b2, dt=np.float32):
    """Same function, bilinear-factorised: never materialise the outer
    product.  q [B,D], keys [B,T,D] -> scores [B,T]."""
    q = q.astype(dt)
    keys = keys.astype(dt)
    D = q.shape[1]
    H = W1.shape[1]
    W1 = W1.astype(dt)
    Wq, Wd, Wk = W1[:D], W1[D:2 * D], W1[2 * D:3 * D]
    Wo = W1[3 * D:].reshape(D, D, H)                 # [i (k index), j (q index), o]
    c = q @ (Wq + Wd) + b1.astype(dt)                # [B,H]
    M = np.einsum("bj,ijo->bio", q, Wo)              # [B,D,H]
    eff = (Wk - Wd)[None] + M                        # [B,D,H]
    pre = np.einsum("btd,bdo->bto", keys, eff) + c[:, None, :]
    h = _din_act(act, pre, dt)
    return (h @ W2.astype(dt) + b2.astype(dt))[..., 0]


def layer_norm(x, gamma, beta, dt=np.float32):
    mu = x.mean(axis=-1, keepdims=True)
    var = np.square(x - mu).mean(axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + dt(LN_EPS)) * gamma.astype(dt) + beta.astype(dt)


def softmax(x):
    x = x - x.max(axis=-1, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis=-1, keepdims=True)


def din_forward(p, profile_ids, item_ids, series_ids, padding_index=0, mask_mode="reference",
                dt=np.float32):
    """DINLayer.call (:229-289).

    profile_ids [B, n_user+n_item]  (user/context features then item features, :237)
    item_ids    [B, n_item]
    series_ids  [B, T, n_item]      (stack(axis=2) of the behaviour series, :258)
    mask quirk (:256,:277-278): mask = (series_0 == padding_index) is 1 on PADDED positions and is
    multiplied in un-negated -- ``mask_mode='reference'`` reproduces that; 'valid' is the intended form.
    p: embed, att (W1,b1,act,W2,b2), mlp: list of (K,b,gamma,beta,act) + out_k,out_b.
    """
    B, T, C = series_ids.shape
    E = p["embed"].shape[1]
    profile = embedding_lookup(p["embed"], profile_ids).astype(dt).reshape(B, -1)       # :243
    q = embedding_lookup(p["embed"], item_ids).astype(dt).reshape(B, -1)                # :253
    pad = (series_ids[:, :, 0] == padding_index)                                       # :256
    mask = pad if mask_mode == "reference" else ~pad
    keys = embedding_lookup(p["embed"], series_ids.reshape(B, T * C)).astype(dt)        # :261-262
    keys = keys.reshape(B, T, C * E)                                                    # :263
    a = p["att"]
    scores = din_activation_unit_factorised(q, keys, a["W1"], a["b1"], a["act"], a["W2"], a["b2"], dt)
    pooled = (keys * (scores * mask.astype(dt))[..., None]).sum(axis=1)                 # :277-282
    x = np.concatenate([profile, pooled], axis=1)                                       # :285
    for lyr in p["mlp"]:
        x = x @ lyr["K"].astype(dt) + lyr["b"].astype(dt)
        x = layer_norm(x, lyr["gamma"], lyr["beta"], dt)
        x = _din_act(lyr["act"], x, dt)
    logits = x @ p["out_k"].astype(dt) + p["out_b"].astype(dt)
    return softmax(logits), scores, pooled


# --------------------------------------------------------------------------------------------
# a12  loss + optimizer         2.FM/ModelManager.py:100,104,171-181
# --------------------------------------------------------------------------------------------

KERAS_EPS = 1e-7


def bce_forward(y, p, dt=np.float32):
    """reduce_sum(keras BinaryCrossentropy()(y, p)): clip p to [eps,1-eps]; -(y log(p+eps) +
    (1-y) log(1-p+eps)); mean over the last axis, then mean over the batch."""
    y = y.astype(dt)
    p = p.astype(dt)
    if p.ndim == 1 and y.ndim == 2 and y.shape[1] == 1:       # keras squeezes y for [B] predictions
        y = y[:, 0]
    eps = dt(KERAS_EPS)
    pc = np.clip(p, eps, dt(1) - eps)
    bce = -(y * np.log(pc + eps) + (dt(1) - y) * np.log(dt(1) - pc + eps))
    if bce.ndim == 1:
        return bce.mean()
    return bce.mean(axis=-1).mean()


def bce_backward(y, p, dt=np.float32):
    """dL/dp following the chain exactly (clip has zero gradient outside (eps, 1-eps))."""
    y = y.astype(dt)
    p = p.astype(dt)
    if p.ndim == 1 and y.ndim == 2 and y.shape[1] == 1:
        y = y[:, 0]
    eps = dt(KERAS_EPS)
    pc = np.clip(p, eps, dt(1) - eps)
    inside = ((p >= eps) & (p <= dt(1) - eps)).astype(dt)
    g = -(y / (pc + eps) - (dt(1) - y) / (dt(1) - pc + eps)) * inside
    return g / dt(p.size)


def l2_used_rows(table, ids, factor, dt=np.float32):
    """5.DIN/ModelManager.py:176-190: all ids of the batch in one vector, tf.unique, gather, factor * tf.nn.l2_loss
    (= sum(x^2)/2).  Returns (loss, uniq ids ascending, d loss / d table[uniq] = factor * table[uniq])."""
    uniq = np.unique(np.asarray(ids).reshape(-1))
    rows = table[uniq].astype(dt)
    return dt(factor) * dt(0.5) * np.square(rows).sum(dtype=dt), uniq, dt(factor) * rows


def dedup_indexed_slices(indices, values, order="sorted"):
    """Sum rows of ``values`` that share an id.

    order='sorted'  -> ids ascending (the build's canonical, deterministic form);
    order='first'   -> ids in order of first occurrence (tf.unique, as Keras' optimizer does).
    Rows of one id are added in ascending position order (as unsorted_segment_sum on CPU does)."""
    indices = np.asarray(indices).reshape(-1)
    if order == "sorted":
        uniq, inv = np.unique(indices, return_inverse=True)
    else:
        uniq_sorted, first, inv_sorted = np.unique(indices, return_index=True, return_inverse=True)
        rank = np.argsort(np.argsort(first, kind="stable"), kind="stable")
        uniq = uniq_sorted[np.argsort(first, kind="stable")]
        inv = rank[inv_sorted]
    out = np.zeros((uniq.shape[0],) + values.shape[1:], dtype=values.dtype)
    np.add.at(out, inv, values)                     # sequential, position order
    return uniq, out


def adam_lr_t(lr, b1, b2, t, dt=np.float32):
    return dt(lr) * np.sqrt(dt(1) - dt(b2) ** dt(t)) / (dt(1) - dt(b1) ** dt(t))


def adam_dense_step(var, m, v, g, t, lr=1e-3, b1=0.9, b2=0.999, eps=KERAS_EPS, dt=np.float32):
    """Keras Adam dense apply (ResourceApplyAdam): m += (g-m)(1-b1); v += (g^2-v)(1-b2);
    var -= lr_t*m/(sqrt(v)+eps).  t is the 1-based step."""
    lr_t = adam_lr_t(lr, b1, b2, t, dt)
    m = m + (g - m) * (dt(1) - dt(b1))
    v = v + (np.square(g) - v) * (dt(1) - dt(b2))
    var = var - lr_t * m / (np.sqrt(v) + dt(eps))
    return var, m, v


def adam_sparse_keras_step(var, m, v, indices, values, t, lr=1e-3, b1=0.9, b2=0.999,
                           eps=KERAS_EPS, dt=np.float32):
    """Keras Adam ``_resource_apply_sparse`` after de-duplication: m <- b1*m on ALL rows,
    m[ids] += (1-b1) g; same for v; var <- var - lr_t*m/(sqrt(v)+eps) on ALL rows (dense sweep)."""
    ids, g = dedup_indexed_slices(indices, values, order="first")
    lr_t = adam_lr_t(lr, b1, b2, t, dt)
    m = m * dt(b1)
    m[ids] += g * (dt(1) - dt(b1))
    v = v * dt(b2)
    v[ids] += np.square(g) * (dt(1) - dt(b2))
    var = var - lr_t * m / (np.sqrt(v) + dt(eps))
    return var, m, v


def adam_rows_step(var, m, v, ids, g, t, lr=1e-3, b1=0.9, b2=0.999, eps=KERAS_EPS, dt=np.float32):
    """'Lazy' variant (NOT the reference's semantics): only touched rows decay/update."""
    lr_t = adam_lr_t(lr, b1, b2, t, dt)
    var, m, v = var.copy(), m.copy(), v.copy()
    m[ids] = m[ids] * dt(b1) + g * (dt(1) - dt(b1))
    v[ids] = v[ids] * dt(b2) + np.square(g) * (dt(1) - dt(b2))
    var[ids] = var[ids] - lr_t * m[ids] / (np.sqrt(v[ids]) + dt(eps))
    return var, m, v


# --------------------------------------------------------------------------------------------
# (e)  row-wise block sharding of a table (no reference counterpart)
# --------------------------------------------------------------------------------------------

def shard_bucketize(ids, rows_per_shard, n_shard):
    """owner = id // rows_per_shard; stable partition of positions by owner.

    Returns perm (positions grouped by owner, ascending position inside a group), send_counts
    [n_shard], local_ids (ids[perm] - owner*rows_per_shard)."""
    ids = np.asarray(ids, dtype=np.int64).reshape(-1)
    owner = ids // rows_per_shard
    if ids.size and (ids.min() < 0 or owner.max() >= n_shard):
        raise IndexError("id outside the sharded table")
    perm = np.argsort(owner, kind="stable").astype(np.int64)
    counts = np.bincount(owner, minlength=n_shard).astype(np.int64)
    local = ids[perm] - owner[perm] * rows_per_shard
    return perm, counts, local


# ---------------------------------------------------------------------------------------------------
# sibling interaction layers (closed forms + hand-derived backward)
# ---------------------------------------------------------------------------------------------------

def pair_list(F):
    """(i, j), i < j, in the row-major order of the upper triangle -- the order tf.boolean_mask keeps
    (2.FM/CustomLayers.py:764-771) and the order of the explicit double loop (:662-665)."""
    return [(i, j) for i in range(F - 1) for j in range(i + 1, F)]


def ipn_forward(table, X, dt=np.float32):
    """PNNLayer.call front end with method='inner' (2.FM/CustomLayers.py:737-745; IpnLayer :773-792):
    combined_vector = [Flatten(embed(X)) | reduce_sum_e(e_i * e_j) for i<j]."""
    e = embedding_lookup(table, X).astype(dt)                    # [B,F,E]
    B, F, E = e.shape
    pairs = pair_list(F)
    prod = np.zeros((B, len(pairs)), dt)
    for p, (i, j) in enumerate(pairs):
        prod[:, p] = (e[:, i, :] * e[:, j, :]).sum(axis=1)
    return np.concatenate([e.reshape(B, F * E), prod], axis=1)


def ipn_backward_vals(table, X, g, dt=np.float32):
    """Per-lookup gradient rows [B*F, E] of ipn_forward for an upstream gradient g [B, F*E + P]."""
    e = embedding_lookup(table, X).astype(dt)
    B, F, E = e.shape
    ge = g[:, :F * E].reshape(B, F, E).astype(dt).copy()
    for p, (i, j) in enumerate(pair_list(F)):
        gp = g[:, F * E + p].astype(dt)[:, None]
        ge[:, i, :] += gp * e[:, j, :]
        ge[:, j, :] += gp * e[:, i, :]
    return ge.reshape(B * F, E)


def bi_interaction_forward(table, X, dt=np.float32):
    """3.DCN/CustomLayers.py:499-501: 0.5 * (square(reduce_sum(e, 1)) - reduce_sum(square(e), 1))  -> [B,E]."""
    e = embedding_lookup(table, X).astype(dt)
    return (0.5 * (np.square(e.sum(axis=1)) - np.square(e).sum(axis=1))).astype(dt)


def bi_interaction_backward_vals(table, X, g, dt=np.float32):
    e = embedding_lookup(table, X).astype(dt)
    B, F, E = e.shape
    S = e.sum(axis=1, keepdims=True)
    return (g.astype(dt)[:, None, :] * (S - e)).reshape(B * F, E)


def batchnorm_forward(x, gamma, beta, mov_mean, mov_var, training, eps=1e-3, momentum=0.99, dt=np.float32):
    """tf.keras.layers.BatchNormalization on [B,N], non-fused path: tf.nn.moments (biased variance) when training,
    moving statistics otherwise; returns (y, new_moving_mean, new_moving_var)."""
    x = x.astype(dt)
    gamma, beta = np.asarray(gamma, dt), np.asarray(beta, dt)
    mov_mean, mov_var = np.asarray(mov_mean, dt), np.asarray(mov_var, dt)
    if training:
        mean = x.mean(axis=0)
        var = np.square(x - mean).mean(axis=0)
        new_mean = mov_mean * momentum + mean * (1 - momentum)
        new_var = mov_var * momentum + var * (1 - momentum)
    else:
        mean, var, new_mean, new_var = mov_mean, mov_var, mov_mean, mov_var
    y = (x - mean) / np.sqrt(var + eps) * gamma + beta
    return y.astype(dt), new_mean.astype(dt), new_var.astype(dt)


def batchnorm_backward(x, gamma, g, eps=1e-3, dt=np.float32):
    """Training-mode backward (batch statistics are functions of x)."""
    x = x.astype(dt); g = g.astype(dt); gamma = np.asarray(gamma, dt)
    B = x.shape[0]
    mean = x.mean(axis=0)
    var = np.square(x - mean).mean(axis=0)
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (x - mean) * rstd
    gbeta = g.sum(axis=0)
    ggamma = (g * xhat).sum(axis=0)
    gx = gamma * rstd * (g - gbeta / B - xhat * ggamma / B)
    return gx.astype(dt), ggamma.astype(dt), gbeta.astype(dt)


def nfm_forward(p, X, X_cont, training=True, dt=np.float32):
    """NeuralFactorizationMachineLayer.call (3.DCN/CustomLayers.py:476-509).  p: embed, bn_gamma, bn_beta,
    bn_mean, bn_var, k1/b1 lists (MLP_layer1, activation on every layer), k2/b2 (MLP_layer2, sigmoid)."""
    second = bi_interaction_forward(p["embed"], X, dt)
    comb = np.concatenate([second, X_cont.astype(dt)], axis=1)
    comb, _, _ = batchnorm_forward(comb, p["bn_gamma"], p["bn_beta"], p["bn_mean"], p["bn_var"], training, dt=dt)
    h = mlp_forward(comb, p["k1"], p["b1"], p.get("activation", "relu"), dt)
    return mlp_forward(h, p["k2"], p["b2"], "sigmoid", dt)


def pnn_forward(p, X, dt=np.float32):
    """PNNLayer.call, method='inner' (2.FM/CustomLayers.py:729-752)."""
    comb = ipn_forward(p["embed"], X, dt)
    h = mlp_forward(comb, p["k1"], p["b1"], "relu", dt)
    return mlp_forward(h, p["k2"], p["b2"], "sigmoid", dt)


def ip_attention_forward(table, q, series, padding_index=0, dt=np.float32):
    """GSULayer.inner_product_attention over the embedded series (7.SIM/CustomLayers.py:88-96,107-118).
    series [B,T,C] -> (masked scores [B,T], pooled [B,C*E])."""
    B, T, C = series.shape
    k = embedding_lookup(table, series.reshape(B, T * C)).astype(dt).reshape(B, T, -1)     # [B,T,D]
    valid = (series[:, :, 0] != padding_index).astype(dt)
    scores = np.einsum("be,ble->bl", q.astype(dt), k) * valid
    pooled = np.einsum("bl,ble->be", scores, k)
    return scores.astype(dt), pooled.astype(dt)


def ip_attention_backward(table, q, series, gpooled, padding_index=0, dt=np.float32):
    """-> (gkeys [B,T,D], gq [B,D])."""
    B, T, C = series.shape
    k = embedding_lookup(table, series.reshape(B, T * C)).astype(dt).reshape(B, T, -1)
    valid = (series[:, :, 0] != padding_index).astype(dt)
    q = q.astype(dt); gpooled = gpooled.astype(dt)
    scores = np.einsum("be,ble->bl", q, k) * valid
    gs = np.einsum("be,ble->bl", gpooled, k) * valid             # d pooled / d score, masked
    gq = np.einsum("bl,ble->be", gs, k)
    gkeys = scores[:, :, None] * gpooled[:, None, :] + gs[:, :, None] * q[:, None, :]
    return gkeys.astype(dt), gq.astype(dt)


def ffm_forward(v, w, bias, X, dt=np.float32):
    """FFMRankingLayer.call written as its double loop (2.FM/CustomLayers.py:398-425) over F tables
    table_i = v[:, i, :]:  ebd_out[i][:, j] * ebd_out[j][:, i] = table_i[x_j] * table_j[x_i].  -> (prob, z) [B,1]."""
    V, F, E = v.shape
    v = v.astype(dt)
    first = w[X, 0].astype(dt).sum(axis=1, keepdims=True)
    second = np.zeros((X.shape[0], 1), dt)
    for i in range(F):
        for j in range(i + 1, F):
            second[:, 0] += (v[X[:, j], i, :] * v[X[:, i], j, :]).sum(axis=1)
    z = bias.astype(dt) + first + second
    return sigmoid(z), z


def ffm_backward(v, X, gz, dt=np.float32):
    """Per-lookup gradient rows of v for an upstream d loss / d z = gz [B,1]: rows [B*F, F, E] where
    rows[b*F + a, c, :] = gz[b] * v[X[b,c], a, :] (c != a); d w rows [B*F,1] = gz; d bias = sum gz."""
    V, F, E = v.shape
    B = X.shape[0]
    v = v.astype(dt)
    rows = np.zeros((B, F, F, E), dt)
    for a in range(F):
        for c in range(F):
            if a != c:
                rows[:, a, c, :] = gz.astype(dt) * v[X[:, c], a, :]
    return rows.reshape(B * F, F, E), np.repeat(gz.astype(dt), F, axis=0), gz.astype(dt).sum(axis=0)
