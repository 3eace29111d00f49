"""GPU parity tests (through the C ABI) of the f4 sibling layers -- PNN inner product, NFM bi-interaction +
BatchNormalization, SIM GSU inner-product attention -- against the oracle (oracle/layers_np.py, oracle/torch_ref.py).
Tolerances: forward 1e-5 relative to the largest entry (north_star), gradients 2e-5.
"""
import numpy as np
import pytest
import torch

from oracle import layers_np as L
from oracle import torch_ref as T
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    assert torch.cuda.is_available()
    import explicit_tf2_recommendation_amd as pkg
    from explicit_tf2_recommendation_amd import layers  # noqa: F401
    return pkg


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def close(a, b, tol=2e-5):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())


def set_params(module, mapping):
    sd = dict(module.named_parameters())
    with torch.no_grad():
        for name, arr in mapping.items():
            assert name in sd, (name, list(sd))
            assert tuple(sd[name].shape) == tuple(arr.shape), (name, sd[name].shape, arr.shape)
            sd[name].copy_(torch.from_numpy(arr))
    assert set(mapping) == set(sd), set(sd) ^ set(mapping)


def fused_view(tab):
    """The table as a strided view of a [V, ld] array (fused FM layout), to exercise ld != E."""
    V, E = tab.shape
    ld = 16
    while ld < E + 1:
        ld *= 2
    st = torch.zeros((V, ld), dtype=torch.float32, device="cuda")
    st[:, :E] = dev(tab)
    return st[:, :E]


# ------------------------------------------------------------------------------------------------
# PNN inner product
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,F,E,strided", [(1, 1, 4, False), (3, 2, 3, False), (4, 5, 8, False), (257, 5, 16, True),
                                           (1030, 26, 16, True), (8192, 26, 16, False), (65, 40, 32, False),
                                           (9, 7, 6, False)])
def test_ipn_fwd_bwd(R, B, F, E, strided):
    ops = R.ops
    V = 5000
    r = H.rng(B + F)
    tab = (r.normal(size=(V, E)) * 0.3).astype(np.float32)
    X = r.integers(0, V, size=(B, F)).astype(np.int64)
    if B > 2:
        X[1] = X[0]                                    # duplicate ids across examples
        X[2, :] = X[2, 0]                              # and inside one example
    t = fused_view(tab) if strided else dev(tab)
    out = ops.emb_ipn_fwd(t, dev(X))
    ref = L.ipn_forward(tab, X, np.float64)
    assert tuple(out.shape) == ref.shape
    assert np.array_equal(out[:, :F * E].cpu().numpy(), tab[X].reshape(B, F * E))      # the flatten part is a copy
    assert close(out.cpu().numpy(), ref, 1e-5)
    g = r.normal(size=ref.shape).astype(np.float32)
    vals = ops.emb_ipn_bwd_vals(out, dev(g), F, E)
    refv = L.ipn_backward_vals(tab, X, g, np.float64)
    assert close(vals.cpu().numpy(), refv)


def test_ipn_micro_case_and_errors(R):
    """PNNLayer docstring inputs (2.FM/CustomLayers.py:699-703) on the hand-checkable table."""
    ops = R.ops
    names = ["user_tag0", "user_tag1", "item_tag1", "item_tag2", "item_tag3"]
    ins = {"item_tag1": np.array([0, 1, 2, 3]), "item_tag2": np.array([4, 5, 6, 7]),
           "item_tag3": np.array([8, 9, 10, 11]), "user_tag0": np.array([12, 13, 14, 15]),
           "user_tag1": np.array([16, 17, 18, 19])}
    X = L.index_assemble(ins, names)
    tab = H.det_table(20, 8)
    out = ops.emb_ipn_fwd(dev(tab), dev(X)).cpu().numpy()
    assert close(out, L.ipn_forward(tab, X, np.float64), 1e-6)
    flag = ops.new_flag(torch.device("cuda"))
    Xb = X.copy()
    Xb[2, 3] = 20
    ops.emb_ipn_fwd(dev(tab), dev(Xb), flag)
    assert int(flag.item()) == 1
    with pytest.raises(RuntimeError):
        ops.emb_ipn_fwd(torch.from_numpy(tab), dev(X))             # CPU tensor: no fallback
    assert tuple(ops.emb_ipn_fwd(dev(tab), dev(X[:0])).shape) == (0, 50)


@pytest.mark.parametrize("two_d", [True, False])
def test_pnn_layer(R, two_d):
    names = ["user_tag0", "user_tag1", "item_tag1", "item_tag2", "item_tag3"]
    V, E, B = 400, 16, 96
    pr = H.pnn_params(3, V, len(names), E)
    layer = R.layers.PNNLayer(feature_names=names, feature_dims=V, embedding_dims=E, mlp_dims=[32, 8]).cuda()
    set_params(layer, {"embed.embeddings": pr["embed"], "MLP_layer1.kernel_0": pr["k1"][0],
                       "MLP_layer1.bias_0": pr["b1"][0], "MLP_layer1.kernel_1": pr["k1"][1],
                       "MLP_layer1.bias_1": pr["b1"][1], "MLP_layer2.kernel_0": pr["k2"][0],
                       "MLP_layer2.bias_0": pr["b2"][0]})
    r = H.rng(4)
    ins = {n: r.integers(0, V, size=(B, 1) if two_d else (B,)).astype(np.int64) for n in names}
    out = layer({k: dev(v) for k, v in ins.items()})["output"]
    X = L.index_assemble(ins, names)
    tp = H.to_torch(pr, torch.float64, True)
    o64 = T.pnn_forward(tp, torch.from_numpy(X))
    assert tuple(out.shape) == (B, 1)
    assert np.abs(out.detach().cpu().numpy() - o64.detach().numpy()).max() <= 1e-5
    assert close(out.detach().cpu().numpy(), L.pnn_forward(pr, X, np.float64), 1e-5)
    y = (r.uniform(size=(B, 1)) < 0.3).astype(np.float32)
    loss = R.functional.KerasBCE.apply(out, dev(y))
    loss.backward()
    lt = T.keras_bce(torch.from_numpy(y).double(), o64)
    lt.backward()
    assert abs(loss.item() - lt.item()) <= 1e-5
    assert close(layer.embed.embeddings.grad.to_dense().cpu().numpy(), tp["embed"].grad.numpy())
    assert close(layer.MLP_layer1.kernel_0.grad.cpu().numpy(), tp["k1"][0].grad.numpy())
    assert close(layer.MLP_layer2.bias_0.grad.cpu().numpy(), tp["b2"][0].grad.numpy())
    with pytest.raises(IndexError):
        bad = dict(ins)
        bad[names[0]] = np.full_like(ins[names[0]], V)
        layer({k: dev(v) for k, v in bad.items()})


# ------------------------------------------------------------------------------------------------
# NFM
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,F,E,strided", [(1, 1, 4, False), (5, 3, 6, False), (300, 10, 16, True), (16384, 10, 32, False),
                                           (77, 26, 16, True), (33, 9, 5, False)])
def test_bi_interaction_fwd_bwd(R, B, F, E, strided):
    ops = R.ops
    V = 3000
    r = H.rng(B * 7 + F)
    tab = (r.normal(size=(V, E)) * 0.3).astype(np.float32)
    X = r.integers(0, V, size=(B, F)).astype(np.int64)
    t = fused_view(tab) if strided else dev(tab)
    comb = torch.full((B, E + 3), 7.0, dtype=torch.float32, device="cuda")
    out, S = ops.emb_bi_fwd(t, dev(X), comb)
    assert out.data_ptr() == comb.data_ptr()
    assert torch.all(comb[:, E:] == 7.0)                           # only the leading E columns are written
    assert close(comb[:, :E].cpu().numpy(), L.bi_interaction_forward(tab, X, np.float64), 1e-5)
    assert close(S.cpu().numpy(), tab[X].astype(np.float64).sum(1), 1e-5)
    g = r.normal(size=(B, E + 3)).astype(np.float32)
    vals = ops.emb_bi_bwd_vals(t, dev(X), dev(g), S)
    assert close(vals.cpu().numpy(), L.bi_interaction_backward_vals(tab, X, g[:, :E], np.float64))


@pytest.mark.parametrize("B,N", [(1, 1), (64, 19), (8192, 19), (1000, 67), (3000, 200)])
def test_batchnorm(R, B, N):
    ops = R.ops
    r = H.rng(B + N)
    x = (r.normal(size=(B, N)) * 3 + 1).astype(np.float32)
    gamma, beta = r.uniform(0.5, 1.5, N).astype(np.float32), r.normal(size=N).astype(np.float32)
    mm, mv = r.normal(size=N).astype(np.float32), r.uniform(0.5, 2, N).astype(np.float32)
    for training in (True, False):
        dm, dv = dev(mm), dev(mv)
        y, xhat, rstd = ops.batchnorm_fwd(dev(x), dev(gamma), dev(beta), dm, dv, training)
        yn, nm, nv = L.batchnorm_forward(x, gamma, beta, mm, mv, training, dt=np.float64)
        assert close(y.cpu().numpy(), yn, 1e-5)
        assert close(dm.cpu().numpy(), nm, 1e-6) and close(dv.cpu().numpy(), nv, 1e-6)
        g = r.normal(size=(B, N)).astype(np.float32)
        gx, gg, gb = ops.batchnorm_bwd(dev(g), xhat, rstd, dev(gamma), training)
        if training:
            rx, rg, rb = L.batchnorm_backward(x, gamma, g, dt=np.float64)
            assert close(gx.cpu().numpy(), rx) and close(gg.cpu().numpy(), rg) and close(gb.cpu().numpy(), rb)
        else:
            assert close(gx.cpu().numpy(), g.astype(np.float64) * gamma / np.sqrt(mv.astype(np.float64) + 1e-3))
    # run-to-run bit identical (tree sums, no atomics)
    a = ops.batchnorm_fwd(dev(x), dev(gamma), dev(beta), dev(mm), dev(mv), True)[0]
    b = ops.batchnorm_fwd(dev(x), dev(gamma), dev(beta), dev(mm), dev(mv), True)[0]
    assert torch.equal(a, b)


@pytest.mark.parametrize("training", [True, False])
def test_nfm_layer(R, training):
    cat = ["uid", "iid", "utag1", "utag2", "utag3", "utag4", "itag1", "itag2", "itag3", "itag4"]
    cont = ["itag4_origin", "itag4_square", "itag4_cube"]
    V, E, B = 1000, 16, 128
    pr = H.nfm_params(5, V, E, len(cont))
    layer = R.layers.NeuralFactorizationMachineLayer(categorical_features=cat, continuous_features=cont, feature_dims=V,
                                                     embedding_dims=E, units=[64, 8], activation="relu").cuda()
    set_params(layer, {"embed.embeddings": pr["embed"], "bn_layer.gamma": pr["bn_gamma"], "bn_layer.beta": pr["bn_beta"],
                       "MLP_layer1.kernel_0": pr["k1"][0], "MLP_layer1.bias_0": pr["b1"][0],
                       "MLP_layer1.kernel_1": pr["k1"][1], "MLP_layer1.bias_1": pr["b1"][1],
                       "MLP_layer2.kernel_0": pr["k2"][0], "MLP_layer2.bias_0": pr["b2"][0]})
    with torch.no_grad():
        layer.bn_layer.moving_mean.copy_(torch.from_numpy(pr["bn_mean"]))
        layer.bn_layer.moving_variance.copy_(torch.from_numpy(pr["bn_var"]))
    layer.train(training)
    r = H.rng(6)
    ins = {n: r.integers(0, V, size=(B, 1)).astype(np.int64) for n in cat}
    xc = {n: r.normal(size=(B, 1)).astype(np.float32) for n in cont}
    feed = {k: dev(v) for k, v in {**ins, **xc}.items()}
    out = layer(feed)["output"]
    X = L.index_assemble(ins, cat)
    XC = np.concatenate([xc[n] for n in cont], axis=1)
    tp = H.to_torch(pr, torch.float64, True)
    o64 = T.nfm_forward(tp, torch.from_numpy(X), torch.from_numpy(XC).double(), training)
    assert tuple(out.shape) == (B, 1)
    assert np.abs(out.detach().cpu().numpy() - o64.detach().numpy()).max() <= 1e-5
    y = (r.uniform(size=(B, 1)) < 0.3).astype(np.float32)
    loss = R.functional.KerasBCE.apply(out, dev(y))
    loss.backward()
    lt = T.keras_bce(torch.from_numpy(y).double(), o64)
    lt.backward()
    assert abs(loss.item() - lt.item()) <= 1e-5
    assert close(layer.embed.embeddings.grad.to_dense().cpu().numpy(), tp["embed"].grad.numpy())
    assert close(layer.bn_layer.gamma.grad.cpu().numpy(), tp["bn_gamma"].grad.numpy())
    assert close(layer.bn_layer.beta.grad.cpu().numpy(), tp["bn_beta"].grad.numpy())
    assert close(layer.MLP_layer1.kernel_0.grad.cpu().numpy(), tp["k1"][0].grad.numpy())
    if training:      # moving statistics moved towards the batch statistics of [second_order | X_cont]
        comb = np.concatenate([L.bi_interaction_forward(pr["embed"], X, np.float64), XC], axis=1)
        _, nm, nv = L.batchnorm_forward(comb, pr["bn_gamma"], pr["bn_beta"], pr["bn_mean"], pr["bn_var"], True,
                                        dt=np.float64)
        assert close(layer.bn_layer.moving_mean.cpu().numpy(), nm, 1e-6)
        assert close(layer.bn_layer.moving_variance.cpu().numpy(), nv, 1e-6)


def test_mlp_layer_batch_norm(R):
    """MLPLayer(is_batch_norm=True): MatMul, BiasAdd, BatchNormalization, activation (2.FM/CustomLayers.py:74-81)."""
    r = H.rng(12)
    x = r.normal(size=(200, 24)).astype(np.float32)
    layer = R.layers.MLPLayer(units=[16, 4], activation="relu", is_batch_norm=True, input_dim=24).cuda()
    layer.train(True)
    out = layer(dev(x))
    h = torch.from_numpy(x).double()
    for i in range(2):
        K = getattr(layer, "kernel_%d" % i).detach().cpu().double()
        b = getattr(layer, "bias_%d" % i).detach().cpu().double()
        h = h @ K + b
        h = torch.relu(T.batchnorm(h, torch.ones(h.shape[1]).double(), torch.zeros(h.shape[1]).double(), None, None,
                                   True))
    assert close(out.detach().cpu().numpy(), h.numpy(), 1e-5)


# ------------------------------------------------------------------------------------------------
# SIM GSU
# ------------------------------------------------------------------------------------------------
def _series(r, B, T_, C, V, all_pad_row=True):
    s = r.integers(1, V, size=(B, T_, C)).astype(np.int64)
    lens = r.integers(1, T_ + 1, size=B)
    if all_pad_row and B > 1:
        lens[1] = 0
    for b in range(B):
        s[b, lens[b]:, :] = 0
    return s


@pytest.mark.parametrize("B,T_,C,E,strided", [(1, 1, 1, 4, False), (5, 7, 3, 8, False), (130, 100, 3, 16, True),
                                              (64, 33, 3, 32, False), (9, 20, 5, 50, False), (4096, 100, 3, 32, False)])
def test_ip_attention_fwd_bwd(R, B, T_, C, E, strided):
    ops = R.ops
    V = 20000
    r = H.rng(B + T_)
    tab = (r.normal(size=(V, E)) * 0.3).astype(np.float32)
    series = _series(r, B, T_, C, V)
    q = r.normal(size=(B, C * E)).astype(np.float32)
    t = fused_view(tab) if strided else dev(tab)
    scores, pooled = ops.ip_attn_fwd(t, dev(series), dev(q), 0)
    if B <= 512:
        s_n, p_n = L.ip_attention_forward(tab, q, series, 0, np.float64)
        assert close(scores.cpu().numpy(), s_n, 1e-5)
        assert close(pooled.cpu().numpy(), p_n, 1e-5)
    else:                                            # full-size case: the torch twin is faster than numpy einsum
        s_t, p_t = T.ip_attention(torch.from_numpy(tab), torch.from_numpy(q), torch.from_numpy(series), 0)
        assert close(scores.cpu().numpy(), s_t.numpy(), 2e-5)
        assert close(pooled.cpu().numpy(), p_t.numpy(), 2e-5)
    assert torch.all(scores[dev(series[:, :, 0] == 0)] == 0)
    g = r.normal(size=(B, C * E)).astype(np.float32)
    gk, gq = ops.ip_attn_bwd(t, dev(series), dev(q), 0, scores, dev(g))
    if B <= 512:
        gk_n, gq_n = L.ip_attention_backward(tab, q, series, g, 0, np.float64)
        assert close(gk.cpu().numpy(), gk_n)
        assert close(gq.cpu().numpy(), gq_n)
    assert torch.all(gk[dev(series[:, :, 0] == 0)] == 0)


def test_gsu_layer(R):
    item = ["i_goods_id", "i_shop_id", "i_cate_id"]
    ser = ["visited_goods_ids", "visited_shop_ids", "visited_cate_ids"]
    V, E, B, T_ = 600, 8, 24, 9
    r = H.rng(31)
    layer = R.layers.GSULayer(item_categorical_features=item, behavior_series_features=ser, feature_dims=V,
                              embedding_dims=E, activation="Dice", padding_index=0, return_series=True).cuda()
    tab = (r.normal(size=(V, E)) * 0.3).astype(np.float32)
    with torch.no_grad():
        layer.embed.embeddings.copy_(torch.from_numpy(tab))
    ins = {n: r.integers(1, V, size=(B, 1)).astype(np.int64) for n in item}
    series = _series(r, B, T_, 3, V)
    for j, n in enumerate(ser):
        ins[n] = series[:, :, j].copy()
    res = layer({k: dev(v) for k, v in ins.items()})
    out = res["output"]
    assert tuple(out.shape) == (B, 2)
    assert np.array_equal(res["valid_mask"].cpu().numpy(), series[:, :, 0] != 0)
    assert np.array_equal(res["X_series"].detach().cpu().numpy(), tab[series].reshape(B, T_, 3 * E))
    # X_combined from the op-for-op twin, then the SAME mlp modules on it: isolates the fused front end
    tt = torch.from_numpy(tab).double().requires_grad_(True)
    itm = L.index_assemble(ins, item)
    xc = T.gsu_combined({"embed": tt}, torch.from_numpy(itm), torch.from_numpy(series), 0)
    ref_out = layer.mlp(xc.detach().float().cuda())
    assert np.abs(out.detach().cpu().numpy() - ref_out.detach().cpu().numpy()).max() <= 1e-5
    # gradient of the table through [q | pooled]: feed the same upstream gradient to both
    layer.zero_grad()
    X_item = R.layers.assemble_index({k: dev(v) for k, v in ins.items()}, item)
    qd = layer.embed(X_item).reshape(B, -1)
    pooled, _ = R.functional.IpAttention.apply(layer.embed.embeddings, qd, dev(series), 0, None)
    g = r.normal(size=(B, 6 * E)).astype(np.float32)
    (torch.cat([qd, pooled], dim=1) * dev(g)).sum().backward()
    (xc * torch.from_numpy(g).double()).sum().backward()
    assert close(layer.embed.embeddings.grad.to_dense().cpu().numpy(), tt.grad.numpy())


# ------------------------------------------------------------------------------------------------
# FFM
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,F,E,zipf", [(1, 1, 4, False), (4, 3, 4, False), (33, 5, 16, False), (700, 26, 16, False),
                                        (65, 7, 6, False), (500, 10, 8, True)])
def test_ffm_fwd_bwd(R, B, F, E, zipf):
    ops = R.ops
    V = 900
    r = H.rng(B + F * 3)
    v = (r.normal(size=(V, F, E)) * 0.2).astype(np.float32)
    w = r.normal(size=(V, 1)).astype(np.float32)
    bias = np.array([0.3], np.float32)
    X = (np.minimum(r.zipf(1.3, size=(B, F)) - 1, V - 1) if zipf else r.integers(0, V, size=(B, F))).astype(np.int64)
    z, prob = ops.ffm_fwd(dev(v), dev(w), dev(bias), dev(X), want_prob=True)
    pn, zn = L.ffm_forward(v, w, bias, X, np.float64)
    assert close(z.cpu().numpy(), zn[:, 0], 1e-5)
    assert np.abs(prob.cpu().numpy() - pn[:, 0]).max() <= 1e-5
    gz = r.normal(size=(B, 1)).astype(np.float32)
    plan = ops.DedupPlan(dev(X), V)
    rows = ops.ffm_bwd_rows(dev(v), dev(X), dev(gz[:, 0].copy()), plan)
    nu = int(plan.n_uniq.item())
    ref_rows, _, _ = L.ffm_backward(v, X, gz, np.float64)
    uid, ref = L.dedup_indexed_slices(X.reshape(-1), ref_rows.reshape(B * F, F * E), "sorted")
    assert nu == len(uid) and np.array_equal(plan.uniq_ids[:nu].cpu().numpy(), uid)
    assert close(rows[:nu].reshape(nu, F * E).cpu().numpy(), ref)
    assert torch.all(rows[nu:] == 0)


def test_ffm_layers(R):
    """FFMRankingLayer docstring inputs (2.FM/CustomLayers.py:372-374) and FFMLayer's default field list."""
    names3 = ["item_tag1", "item_tag2", "item_tag3"]
    ins = {"item_tag1": np.array([0, 1, 2, 3]), "item_tag2": np.array([4, 5, 6, 7]), "item_tag3": np.array([8, 9, 10, 11])}
    layer = R.layers.FFMRankingLayer(feature_names=names3, feature_dims=20, embedding_dims=16).cuda()
    out = layer({k: dev(v) for k, v in ins.items()})["output"]
    p = {k: q.detach().cpu().numpy() for k, q in layer.named_parameters()}
    assert set(p) == {"bias", "w", "fa_interaction_layer.v"}
    assert len(layer.embedding_list) == 3 and tuple(layer.embedding_list[1].shape) == (20, 16)
    X = L.index_assemble(ins, names3)
    pn, _ = L.ffm_forward(p["fa_interaction_layer.v"], p["w"], p["bias"], X, np.float64)
    assert tuple(out.shape) == (4, 1) and np.abs(out.detach().cpu().numpy() - pn).max() <= 1e-6
    # a training step's gradients against the torch twin (vectorised FieldAwareInteractionLayer form)
    names5 = ["item_tag1", "item_tag2", "item_tag3", "user_tag0", "user_tag1"]
    V, E, B = 300, 8, 64
    layer = R.layers.FFMLayer(feature_names=names5, feature_dims=V, embedding_dims=E).cuda()
    r = H.rng(77)
    with torch.no_grad():
        layer.fa_interaction_layer.v.mul_(6.0)
    ins = {n: r.integers(0, V, size=(B, 1)).astype(np.int64) for n in names5}
    out = layer({k: dev(v) for k, v in ins.items()})["output"]
    tp = {"v": layer.fa_interaction_layer.v.detach().cpu().double().requires_grad_(True),
          "w": layer.w.detach().cpu().double().requires_grad_(True),
          "bias": layer.bias.detach().cpu().double().requires_grad_(True)}
    X = L.index_assemble(ins, names5)
    o64 = T.ffm_forward(tp, torch.from_numpy(X))
    assert np.abs(out.detach().cpu().numpy() - o64.detach().numpy()).max() <= 1e-5
    y = (r.uniform(size=(B, 1)) < 0.3).astype(np.float32)
    loss = R.functional.KerasBCE.apply(out, dev(y))
    loss.backward()
    lt = T.keras_bce(torch.from_numpy(y).double(), o64)
    lt.backward()
    assert abs(loss.item() - lt.item()) <= 1e-5
    assert close(layer.fa_interaction_layer.v.grad.to_dense().cpu().numpy(), tp["v"].grad.numpy())
    assert close(layer.w.grad.to_dense().cpu().numpy(), tp["w"].grad.numpy())
    assert close(layer.bias.grad.cpu().numpy(), tp["bias"].grad.numpy())


def test_f4_layers_train_through_manager(R):
    """make_layer_choice strings of the reference (2.FM/ModelManager.py:76-82; 3.DCN/ModelManager.py:78-79): a few Adam
    steps on one batch lower its loss."""
    from explicit_tf2_recommendation_amd import data, layers
    from explicit_tf2_recommendation_amd.model_manager import ModelManager
    names = ["user_tag1", "user_tag2", "item_tag1", "item_tag2", "item_tag3"]
    B, V = 128, 3000
    for name, cls in (("ffm_ranking", layers.FFMRankingLayer), ("pnn_ranking", layers.PNNRankingLayer)):
        layers.set_init_seed(5)
        mm = ModelManager(feature_names=names, data_info=data.data_info(V, 5), embedding_dims=8, lr=0.01, batch=B,
                          layer=name)
        assert isinstance(mm.layer, cls)
        batch = data.SyntheticGenerator(names, V, dist="zipf", seed=1).batch(B)
        losses = [mm.train_loop(dict(batch)).item() for _ in range(8)]
        assert np.all(np.isfinite(losses)) and losses[-1] < losses[0], (name, losses)
    cat = ["uid", "iid", "utag1", "utag2", "utag3", "utag4", "itag1", "itag2", "itag3", "itag4"]
    cont = ["itag4_origin", "itag4_square", "itag4_cube"]
    layers.set_init_seed(6)
    mm = ModelManager(feature_names=cat, continuous_features=cont, data_info=data.data_info(4000, 10), embedding_dims=8,
                      lr=0.01, batch=B, layer="NFM")
    assert isinstance(mm.layer, layers.NeuralFactorizationMachineLayer)
    batch = data.SyntheticGenerator(cat, 4000, continuous=cont, seed=2).batch(B)
    losses = [mm.train_loop(dict(batch)).item() for _ in range(8)]
    assert np.all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    assert not torch.equal(mm.layer.bn_layer.moving_mean, torch.zeros_like(mm.layer.bn_layer.moving_mean))
