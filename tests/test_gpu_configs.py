"""GPU parity tests AT the shapes of BASELINE.json's configs C, D and E, which select other
kernels / code paths than the small cases of the per-op tests:

  C  DCN matrix-mode CrossNet, B = 16384, D = 3 + F*32 in {323, 835}: the large-tile MFMA GEMM with the cross
     epilogue (+ aux), its EPI_ADD form and the split-K weight gradient (3.DCN/CustomLayers.py:297-305);
  D  DSSM towers at E = 64 (192 -> 64 -> 32 -> 8 and 128 -> 64 -> 32 -> 8) over row-sharded tables
     (2.FM/CustomLayers.py:183-206,230-239);
  E  DIN ActivationUnit at T in {64, 65, 100, 128}, E = 32 (D = 96): the multi-chunk time loop of the attention
     kernels (5.DIN/CustomLayers.py:163-180,256-282);
  +  the de-duplication plan captured in a hipGraph and REPLAYED at n around and above 2^20 keys.

Oracle: fp64 numpy / torch-CPU restatements (oracle/), tolerances in the asserts.  Everything goes through the C ABI.
"""
import os
import socket

import numpy as np
import pytest
import torch

from oracle import layers_np as L
from oracle import torch_ref as T
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    import explicit_tf2_recommendation_amd as pkg
    return pkg


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def close(a, b, tol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())


def grad_np(p):
    g = p.grad
    return (g.to_dense() if g.is_sparse else g).cpu().numpy()


# ---------------------------------------------------------------------------------------------------------------
# E: the short-K, wide-N product in front of the DIN attention (q.Wcat + bext: row panels x several column panels)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K,tb", [(4096, 3492, 96, 0), (4096, 3492, 96, 1), (3000, 1001, 200, 0), (2048, 5000, 33, 1)])
def test_short_k_wide_n_gemm_in_column_panels(R, M, N, K, tb):
    ops = R.ops
    r = H.rng(M + N + K)
    A = r.normal(size=(M, K)).astype(np.float32)
    Bm = r.normal(0, 0.1, size=(N, K) if tb else (K, N)).astype(np.float32)
    b = r.normal(size=(N,)).astype(np.float32)
    E1 = r.normal(size=(M, N)).astype(np.float32)
    ref = A.astype(np.float64) @ (Bm.astype(np.float64).T if tb else Bm.astype(np.float64))
    tol = 2e-6 * np.sqrt(K) * max(1.0, np.abs(ref).max())
    Ad, Bd, bd, E1d = dev(A), dev(Bm), dev(b), dev(E1)
    C = ops.gemm(Ad, Bd, transB=bool(tb)).cpu().numpy()
    assert np.abs(C - ref).max() <= tol
    C = ops.gemm(Ad, Bd, transB=bool(tb), epi=ops.EPI_BIAS, bias=bd).cpu().numpy()
    assert np.abs(C - (ref + b)).max() <= tol
    C = ops.gemm(Ad, Bd, transB=bool(tb), epi=ops.EPI_BIAS_RELU, bias=bd).cpu().numpy()
    assert np.abs(C - np.maximum(ref + b, 0)).max() <= tol
    C = ops.gemm(Ad, Bd, transB=bool(tb), epi=ops.EPI_ADD, e1=E1d).cpu().numpy()
    assert np.abs(C - (ref + E1)).max() <= tol
    wide = torch.full((M, N + 5), 7.0, device="cuda")                 # strided output: columns beyond N stay untouched
    ops.gemm(Ad, Bd, transB=bool(tb), epi=ops.EPI_BIAS, bias=bd, out=wide[:, :N])
    assert np.abs(wide[:, :N].cpu().numpy() - (ref + b)).max() <= tol and torch.all(wide[:, N:] == 7.0)


# ---------------------------------------------------------------------------------------------------------------
# C: the GEMMs of MatrixCrossLayer at config size
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D", [323, 835])
def test_crossnet_gemms_at_config_c(R, D):
    """M = B = 16384, N = K = D: forward X.W^T with EPI_CROSS (+ the U side output), backward H.W with EPI_ADD and the
    split-K weight gradient H^T.X, each against fp64 numpy; the remaining (transA, transB) form for completeness."""
    ops = R.ops
    r = H.rng(D)
    M = 16384
    X = r.normal(size=(M, D)).astype(np.float32)
    X0 = r.normal(size=(M, D)).astype(np.float32)
    W = r.normal(0, 0.05, size=(D, D)).astype(np.float32)
    b = r.normal(0, 0.05, size=(D,)).astype(np.float32)
    Hm = r.normal(size=(M, D)).astype(np.float32)
    G = r.normal(size=(M, D)).astype(np.float32)
    Xd, X0d, Wd, bd, Hd, Gd = (dev(a) for a in (X, X0, W, b, Hm, G))
    X64, W64 = X.astype(np.float64), W.astype(np.float64)

    # forward: x_{l+1} = x0 * (x_l W^T + b) + x_l ; U kept for the backward
    U64 = X64 @ W64.T + b
    aux = torch.empty((M, D), device="cuda")
    Y = ops.gemm(Xd, Wd, transB=True, epi=ops.EPI_CROSS, bias=bd, e0=X0d, e1=Xd, aux=aux).cpu().numpy()
    tol_u = 2e-6 * np.sqrt(D) * max(1.0, np.abs(U64).max())
    assert np.abs(aux.cpu().numpy() - U64).max() <= tol_u
    Yref = X0 * U64 + X64
    assert np.abs(Y - Yref).max() <= 4 * tol_u * max(1.0, np.abs(X0).max())
    # a contiguous aux beside a strided out would be written with the wrong leading dimension: refused
    wide = torch.empty((M, D + 3), device="cuda")
    with pytest.raises(ValueError):
        ops.gemm(Xd, Wd, transB=True, epi=ops.EPI_CROSS, bias=bd, e0=X0d, e1=Xd, aux=aux, out=wide[:, :D])

    # backward dX_l = G + H W
    dX = ops.gemm(Hd, Wd, epi=ops.EPI_ADD, e1=Gd).cpu().numpy()
    ref = Hm.astype(np.float64) @ W64 + G
    assert np.abs(dX - ref).max() <= 2e-6 * np.sqrt(D) * max(1.0, np.abs(ref).max())

    # backward dW = H^T X (K = 16384 split over slices added in slice order: deterministic)
    ref = Hm.astype(np.float64).T @ X64
    tol = 2e-6 * np.sqrt(M) * np.abs(ref).max()
    for split in (None, 8, 40):
        dW = torch.empty((D, D), device="cuda")
        dW1 = ops.gemm(Hd, Xd, transA=True, split_k=split, out=dW).cpu().numpy()
        dW2 = ops.gemm(Hd, Xd, transA=True, split_k=split).cpu().numpy()
        assert np.array_equal(dW1, dW2)
        assert np.abs(dW1 - ref).max() <= tol

    # (transA, transB) = (1, 1) and plain (0, 0) on the same big shapes
    C = ops.gemm(dev(X.T.copy()), dev(W.T.copy()), transA=True, transB=True).cpu().numpy()
    ref = X64 @ W64
    assert np.abs(C - ref).max() <= 2e-6 * np.sqrt(D) * max(1.0, np.abs(ref).max())
    C = ops.gemm(Xd, Wd).cpu().numpy()
    assert np.abs(C - ref).max() <= 2e-6 * np.sqrt(D) * max(1.0, np.abs(ref).max())


CONT = ["itag4_origin", "itag4_square", "itag4_cube"]


@pytest.mark.parametrize("F", [10, 26])
def test_dcn_matrix_layer_at_config_c(R, F):
    """DeepCrossNetworkLayer(type='matrix') at B = 16384, E = 32, 3 continuous + F categorical features (D = 323 / 835)
    against the oracle in fp64 (row form of the same contraction): probabilities 1e-5, every gradient 3e-5 relative."""
    B, E, V = 16384, 32, 200_000
    cat = ["c%d" % i for i in range(F)]
    pr = H.dcn_params(31 + F, V, F, E, kind="mat")
    layer = R.layers.DeepCrossNetworkLayer(categorical_features=cat, continuous_features=CONT, feature_dims=V,
                                           embedding_dims=E, type="matrix").cuda()
    sd = dict(layer.named_parameters())
    mp = {"embedding_layer.embeddings": pr["embed"], "output_layer.kernel": pr["out_k"], "output_layer.bias": pr["out_b"]}
    for i in range(3):
        mp["cross_layer.w%d" % i] = pr["cross_w"][i]
        mp["cross_layer.b%d" % i] = pr["cross_b"][i]
    for i in range(2):
        mp["dense_layer.hidden_layer.%d.kernel" % i] = pr["dnn_k"][i]
        mp["dense_layer.hidden_layer.%d.bias" % i] = pr["dnn_b"][i]
    assert set(mp) == set(sd)
    with torch.no_grad():
        for k, a in mp.items():
            sd[k].copy_(torch.from_numpy(a))
    r = H.rng(32)
    dims = [V // F] * F
    dims[-1] += V - sum(dims)
    off = np.concatenate([[0], np.cumsum(dims)[:-1]])
    ins = {n: (off[f] + np.minimum(r.zipf(1.1, size=B) - 1, dims[f] - 1)).astype(np.int64)[:, None]
           for f, n in enumerate(cat)}
    cont = r.normal(size=(B, 3)).astype(np.float32)
    feed = {k: dev(v) for k, v in ins.items()}
    for j, n in enumerate(CONT):
        feed[n] = dev(cont[:, j:j + 1])
    out = layer(feed)["output"]
    X = L.index_assemble(ins, cat)
    tp = H.to_torch(pr, torch.float64, True)
    o64 = T.dcn_forward(tp, torch.from_numpy(X), torch.from_numpy(cont).double(), "mat", row_form=True)
    assert tuple(out.shape) == (B, 1)
    assert np.abs(out.detach().cpu().numpy() - o64.detach().numpy()).max() <= 1e-5
    y = (r.uniform(size=(B, 1)) < 0.4).astype(np.float32)
    loss = R.functional.KerasBCE.apply(out, dev(y))
    loss.backward()
    lt = T.keras_bce(torch.from_numpy(y).double(), o64)
    lt.backward()
    assert abs(loss.item() - lt.item()) <= 1e-5
    assert close(grad_np(layer.embedding_layer.embeddings), tp["embed"].grad.numpy(), 3e-5)
    for i in range(3):
        assert close(grad_np(getattr(layer.cross_layer, "w%d" % i)), tp["cross_w"][i].grad.numpy(), 3e-5)
        assert close(grad_np(getattr(layer.cross_layer, "b%d" % i)), tp["cross_b"][i].grad.numpy(), 3e-5)
    assert close(grad_np(layer.dense_layer.hidden_layer[0].kernel), tp["dnn_k"][0].grad.numpy(), 3e-5)
    assert close(grad_np(layer.output_layer.kernel), tp["out_k"].grad.numpy(), 3e-5)


# ---------------------------------------------------------------------------------------------------------------
# D: DSSM at E = 64 over row-sharded tables (world size 1: the same kernels and collectives as N > 1)
# ---------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_dssm_two_tower_e64_sharded_tables_at_config_d(R):
    """u_tower: 2 fields x 64d -> 64 -> 32 -> 8, i_tower: 3 fields x 64d -> 64 -> 32 -> 8, (1 - cos)/2, Keras BCE; both
    tables behind sharded.ShardedEmbedding (bucketize, RCCL all-to-all of ids / rows / row gradients, owner-side
    de-duplication).  Against the oracle on the full tables: outputs 1e-5, gradients 2e-5 relative."""
    import torch.distributed as dist
    from explicit_tf2_recommendation_amd import sharded
    un, inn = ["user_tag1", "user_tag2"], ["item_tag1", "item_tag2", "item_tag3"]
    Vu, Vi, E, B = 20_011, 300_007, 64, 4096
    pu, pi = H.tower_params(41, Vu, 2, E), H.tower_params(42, Vi, 3, E)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        # the product path: the reference's constructor plus sharded=True (layers.make_embedding -> sharded.ShardedEmbedding)
        layer = R.layers.DSSMTwoTowerRetrievalLayer(u_feature_names=un, i_feature_names=inn, u_feature_dims=Vu,
                                                    i_feature_dims=Vi, u_embedding_dims=E, i_embedding_dims=E,
                                                    sharded=True).cuda()
        tabs = {}
        for tower, p, V in (("u_tower", pu, Vu), ("i_tower", pi, Vi)):
            emb = getattr(layer, tower).embed
            assert isinstance(emb, sharded.ShardedEmbedding)
            emb.load_global_rows(dev(p["embed"]))
            tabs[tower] = emb
            t = getattr(layer, tower)
            with torch.no_grad():
                t.mlp.kernel_0.copy_(dev(p["mlp_k"][0])); t.mlp.bias_0.copy_(dev(p["mlp_b"][0]))
                t.mlp.kernel_1.copy_(dev(p["mlp_k"][1])); t.mlp.bias_1.copy_(dev(p["mlp_b"][1]))
                t.final.kernel_0.copy_(dev(p["final_k"][0])); t.final.bias_0.copy_(dev(p["final_b"][0]))
        r = H.rng(43)
        ins = {}
        for names, V in ((un, Vu), (inn, Vi)):
            for n in names:
                ins[n] = np.minimum(r.zipf(1.1, size=(B, 1)) - 1, V - 1).astype(np.int64)
        ins[inn[0]][:64] = r.integers(0, Vi, size=(64, 1))           # some uniformly drawn ids too
        res = layer({k: dev(v) for k, v in ins.items()})
        Xu, Xi = L.index_assemble(ins, un), L.index_assemble(ins, inn)
        tu, ti = H.to_torch(pu, torch.float64, True), H.to_torch(pi, torch.float64, True)
        u64, i64 = T.dssm_tower(tu, torch.from_numpy(Xu)), T.dssm_tower(ti, torch.from_numpy(Xi))
        s64 = T.two_tower_score(u64, i64)
        assert tuple(res["output"].shape) == (B,)
        assert np.abs(res["user_embedding"].detach().cpu().numpy() - u64.detach().numpy()).max() <= 1e-5
        assert np.abs(res["item_embedding"].detach().cpu().numpy() - i64.detach().numpy()).max() <= 1e-5
        assert np.abs(res["output"].detach().cpu().numpy() - s64.detach().numpy()).max() <= 1e-5
        y = (r.uniform(size=(B, 1)) < 0.25).astype(np.float32)
        loss = R.functional.KerasBCE.apply(res["output"], dev(y))
        loss.backward()
        lt = T.keras_bce(torch.from_numpy(y).double(), s64)
        lt.backward()
        assert abs(loss.item() - lt.item()) <= 1e-5
        for tower, tp, V in (("u_tower", tu, Vu), ("i_tower", ti, Vi)):
            got = tabs[tower].embeddings_shard.grad.to_dense().cpu().numpy()[:V]
            assert close(got, tp["embed"].grad.numpy(), 2e-5), tower
            t = getattr(layer, tower)
            assert close(grad_np(t.mlp.kernel_0), tp["mlp_k"][0].grad.numpy(), 2e-5)
            assert close(grad_np(t.mlp.kernel_1), tp["mlp_k"][1].grad.numpy(), 2e-5)
            assert close(grad_np(t.final.kernel_0), tp["final_k"][0].grad.numpy(), 2e-5)
            assert close(grad_np(t.final.bias_0), tp["final_b"][0].grad.numpy(), 2e-5)
        for emb in tabs.values():
            emb.check_flags()
    finally:
        dist.destroy_process_group()


def test_din_layer_over_a_sharded_table_at_config_e_shape(R):
    """BASELINE config E's "sharded embeddings" half: DINLayer(sharded=True) at T = 100, E = 32 (5 user + 3 item + 3
    series features), B = 64, against the oracle on the full table -- ONE de-duplicated exchange serves the profile ids,
    the 100-step behaviour series and the padding id; the attention kernel then runs on the rows that came back with
    slots in place of ids (both mask conventions)."""
    import torch.distributed as dist
    from explicit_tf2_recommendation_amd import sharded
    from tests.test_gpu_din import _load_din
    user = ["uid", "utag1", "utag2", "utag3", "utag4"]
    item = ["i_goods_id", "i_shop_id", "i_cate_id"]
    ser = ["visited_goods_ids", "visited_shop_ids", "visited_cate_ids"]
    V, E, B, T_ = 5000, 32, 64, 100
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for mode in ("valid", "reference"):
            pr = H.din_params(71, V, E, act="dice")
            ref = R.layers.DINLayer(user_and_context_categorical_features=user, item_categorical_features=item,
                                    behavior_series_features=ser, feature_dims=V, embedding_dims=E, activation="Dice",
                                    padding_index=0, mask_mode=mode).cuda()
            _load_din(ref, pr, "Dice")
            layer = R.layers.DINLayer(user_and_context_categorical_features=user, item_categorical_features=item,
                                      behavior_series_features=ser, feature_dims=V, embedding_dims=E, activation="Dice",
                                      padding_index=0, mask_mode=mode, sharded=True).cuda()
            assert isinstance(layer.embed, sharded.ShardedEmbedding)
            sd = dict(ref.named_parameters()); sd.update(dict(ref.named_buffers()))
            with torch.no_grad():
                layer.embed.load_global_rows(ref.embed.embeddings.detach())
                for n, q in list(layer.named_parameters()) + list(layer.named_buffers()):
                    if not n.endswith("embeddings_shard"):
                        q.copy_(sd[n])
            r = H.rng(72)
            ins = {n: r.integers(1, V, size=(B, 1)).astype(np.int64) for n in user + item}
            series, _ = _series(r, B, T_, 3, V)
            for j, n in enumerate(ser):
                ins[n] = series[:, :, j].copy()
            out = layer({k: dev(v) for k, v in ins.items()})["output"]
            layer.embed.check_flags()
            profile = L.index_assemble(ins, user + item)
            itm = L.index_assemble(ins, item)
            tp = H.to_torch(pr, torch.float64, True)
            o64, _, _ = T.din_forward(tp, torch.from_numpy(profile), torch.from_numpy(itm), torch.from_numpy(series), 0,
                                      mode)
            assert np.abs(out.detach().cpu().numpy() - o64.detach().numpy()).max() <= 1e-5, mode
            y = (r.uniform(size=(B, 1)) < 0.4).astype(np.float32)
            loss = R.functional.KerasBCE.apply(out, dev(np.repeat(y, 2, axis=1)))
            loss.backward()
            lt = T.keras_bce(torch.from_numpy(y).double(), o64)
            lt.backward()
            assert abs(loss.item() - lt.item()) <= 1e-5
            assert close(layer.embed.embeddings_shard.grad.to_dense().cpu().numpy()[:V], tp["embed"].grad.numpy(), 3e-5), mode
            base = layer.din_activation_layer
            assert close(base.mlp_layer.layers[0].kernel.grad.cpu().numpy(), tp["att"]["W1"].grad.numpy(), 3e-5)
            assert close(base.output_layer.kernel.grad.cpu().numpy(), tp["att"]["W2"].grad.numpy(), 3e-5)
            assert close(layer.mlp.layers[0].kernel.grad.cpu().numpy(), tp["mlp"][0]["K"].grad.numpy(), 3e-5)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("F,E", [(3, 64), (2, 64)])
def test_dssm_tower_e64(R, F, E):
    """DSSMSingleTowerLayer at config D's widths, plain (unsharded) table, B = 8192."""
    names = ["t%d" % i for i in range(F)]
    V, B = 250_000, 8192
    p = H.tower_params(50 + F, V, F, E)
    layer = R.layers.DSSMSingleTowerLayer(feature_names=names, feature_dims=V, embedding_dims=E, mlp_dims=[64, 32],
                                          final_dim=8).cuda()
    sd = dict(layer.named_parameters())
    mp = {"embed.embeddings": p["embed"], "mlp.kernel_0": p["mlp_k"][0], "mlp.bias_0": p["mlp_b"][0],
          "mlp.kernel_1": p["mlp_k"][1], "mlp.bias_1": p["mlp_b"][1], "final.kernel_0": p["final_k"][0],
          "final.bias_0": p["final_b"][0]}
    assert set(mp) == set(sd)
    with torch.no_grad():
        for k, a in mp.items():
            sd[k].copy_(torch.from_numpy(a))
    r = H.rng(51)
    ins = {n: r.integers(0, V, size=(B, 1)).astype(np.int64) for n in names}
    out = layer({k: dev(v) for k, v in ins.items()})["output"]
    X = L.index_assemble(ins, names)
    tp = H.to_torch(p, torch.float64, True)
    o64 = T.dssm_tower(tp, torch.from_numpy(X))
    assert tuple(out.shape) == (B, 8)
    assert np.abs(out.detach().cpu().numpy() - o64.detach().numpy()).max() <= 1e-5
    g = r.normal(size=(B, 8)).astype(np.float32)
    (out * dev(g)).sum().backward()
    (o64 * torch.from_numpy(g).double()).sum().backward()
    assert close(grad_np(layer.embed.embeddings), tp["embed"].grad.numpy(), 2e-5)
    assert close(grad_np(layer.mlp.kernel_0), tp["mlp_k"][0].grad.numpy(), 2e-5)
    assert close(grad_np(layer.final.kernel_0), tp["final_k"][0].grad.numpy(), 2e-5)


# ---------------------------------------------------------------------------------------------------------------
# E: DIN attention over long series (several 64-step chunks per example)
# ---------------------------------------------------------------------------------------------------------------
def _series(r, B, T_, C, V, full=(0,), empty=()):
    series = r.integers(1, V, size=(B, T_, C)).astype(np.int64)
    lens = r.integers(1, T_ + 1, size=B)
    for b in full:
        lens[b] = T_
    for b in empty:
        lens[b] = 0                                  # an all-padding series
    for b in range(B):
        series[b, lens[b]:, :] = 0
    return series, lens


@pytest.mark.parametrize("act", ["dice", "prelu"])
@pytest.mark.parametrize("T_,mask_valid", [(64, 0), (64, 1), (65, 0), (65, 1), (100, 0), (100, 1), (128, 0), (128, 1),
                                           (200, 1)])
def test_din_attention_long_series_vs_literal_activation_unit(R, act, T_, mask_valid):
    """E = 32, C = 3 (D = 96, the literal unit is 9504 wide), B = 16: scores, pooled and every gradient (keys -> table,
    q, W1, b1, W2, b2, alpha) against the LITERAL ActivationUnit evaluated per time step in fp64."""
    ops, Fn = R.ops, R.functional
    r = H.rng(1000 + T_ + mask_valid)
    V, B, C, E = 3000, 16, 3, 32
    D = C * E
    pr = H.din_params(60 + T_, V, E, act=act)
    a = pr["att"]
    series, lens = _series(r, B, T_, C, V, full=(0, 5), empty=(3,))
    lens_long = r.integers(max(1, T_ - 3), T_ + 1, size=4)          # lengths that end inside the last chunk
    for j, b in enumerate((7, 8, 9, 10)):
        series[b] = r.integers(1, V, size=(T_, C))
        series[b, lens_long[j]:, :] = 0
    q = r.normal(size=(B, D)).astype(np.float32)
    ta = H.to_torch(a, torch.float64, True)
    emb_t = torch.from_numpy(pr["embed"]).double().requires_grad_()
    qt = torch.from_numpy(q).double().requires_grad_()
    keys = T.lookup(emb_t, torch.from_numpy(series.reshape(B, T_ * C))).reshape(B, T_, D)
    sc = torch.stack([T.din_activation_unit(qt, keys[:, t, :], ta) for t in range(T_)], dim=1).squeeze(-1)
    pad = torch.from_numpy(series[:, :, 0] == 0)
    m = (~pad if mask_valid else pad).double()
    pooled_t = (keys * (sc * m).unsqueeze(-1)).sum(1)
    g = r.normal(size=(B, D)).astype(np.float32)
    (pooled_t * torch.from_numpy(g).double()).sum().backward()
    kind = ops.DACT_CODE[act]
    alpha = dev(a["act"]["alpha"]).requires_grad_()
    mean = dev(a["act"]["mean"]) if act == "dice" else None
    var = dev(a["act"]["var"]) if act == "dice" else None
    emb_d = dev(pr["embed"]).requires_grad_()
    qd = dev(q).requires_grad_()
    W1, b1, W2, b2 = [dev(a[k]).requires_grad_() for k in ("W1", "b1", "W2", "b2")]
    pooled, scores = Fn.DinAttention.apply(emb_d, qd, dev(series), W1, b1, kind, alpha, mean, var, W2, b2, 0,
                                           mask_valid, None)
    (pooled * dev(g)).sum().backward()
    assert close(scores.cpu().numpy(), sc.detach().numpy(), 1e-5)
    assert close(pooled.detach().cpu().numpy(), pooled_t.detach().numpy(), 1e-5)
    assert close(qd.grad.cpu().numpy(), qt.grad.numpy(), 3e-5)
    assert close(emb_d.grad.to_dense().cpu().numpy(), emb_t.grad.numpy(), 3e-5)
    assert close(W1.grad.cpu().numpy(), ta["W1"].grad.numpy(), 3e-5)
    assert close(b1.grad.cpu().numpy(), ta["b1"].grad.numpy(), 3e-5)
    assert close(W2.grad.cpu().numpy(), ta["W2"].grad.numpy(), 3e-5)
    assert close(b2.grad.cpu().numpy(), ta["b2"].grad.numpy(), 3e-5)
    assert close(alpha.grad.cpu().numpy(), ta["act"]["alpha"].grad.numpy(), 3e-5)
    if not mask_valid:                       # reference quirk: a series without padding pools to exactly zero
        assert np.all(pooled.detach().cpu().numpy()[0] == 0.0)
    else:                                    # ... and with the valid mask an all-padding series does
        assert np.all(pooled.detach().cpu().numpy()[3] == 0.0)


def test_din_layer_at_config_e_shape(R):
    """DINLayer end to end at T = 100, E = 32 (5 user + 3 item + 3 series features), B = 64: output and gradients."""
    from tests.test_gpu_din import _load_din
    user = ["uid", "utag1", "utag2", "utag3", "utag4"]
    item = ["i_goods_id", "i_shop_id", "i_cate_id"]
    ser = ["visited_goods_ids", "visited_shop_ids", "visited_cate_ids"]
    V, E, B, T_ = 5000, 32, 64, 100
    pr = H.din_params(71, V, E, act="dice")
    layer = R.layers.DINLayer(user_and_context_categorical_features=user, item_categorical_features=item,
                              behavior_series_features=ser, feature_dims=V, embedding_dims=E, activation="Dice",
                              padding_index=0, mask_mode="valid").cuda()
    _load_din(layer, pr, "Dice")
    r = H.rng(72)
    ins = {n: r.integers(1, V, size=(B, 1)).astype(np.int64) for n in user + item}
    series, _ = _series(r, B, T_, 3, V)
    for j, n in enumerate(ser):
        ins[n] = series[:, :, j].copy()
    out = layer({k: dev(v) for k, v in ins.items()})["output"]
    profile = L.index_assemble(ins, user + item)
    itm = L.index_assemble(ins, item)
    tp = H.to_torch(pr, torch.float64, True)
    o64, _, _ = T.din_forward(tp, torch.from_numpy(profile), torch.from_numpy(itm), torch.from_numpy(series), 0, "valid")
    assert np.abs(out.detach().cpu().numpy() - o64.detach().numpy()).max() <= 1e-5
    y = (r.uniform(size=(B, 1)) < 0.4).astype(np.float32)
    loss = R.functional.KerasBCE.apply(out, dev(np.repeat(y, 2, axis=1)))
    loss.backward()
    lt = T.keras_bce(torch.from_numpy(y).double(), o64)
    lt.backward()
    assert abs(loss.item() - lt.item()) <= 1e-5
    assert close(layer.embed.embeddings.grad.to_dense().cpu().numpy(), tp["embed"].grad.numpy(), 3e-5)
    base = layer.din_activation_layer
    assert close(base.mlp_layer.layers[0].kernel.grad.cpu().numpy(), tp["att"]["W1"].grad.numpy(), 3e-5)
    assert close(base.output_layer.kernel.grad.cpu().numpy(), tp["att"]["W2"].grad.numpy(), 3e-5)
    assert close(layer.mlp.layers[0].kernel.grad.cpu().numpy(), tp["mlp"][0]["K"].grad.numpy(), 3e-5)


# ---------------------------------------------------------------------------------------------------------------
# the de-duplication plan inside a captured hipGraph, replayed
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,V", [(1 << 19, 50_000_000), ((1 << 19) + 777, 10_000_000), ((1 << 20) + 4097, 50_000_000),
                                 (1_273_856, 50_000_000), (2_500_000, 3_000_000_000), (5000, 97)])
def test_dedup_plan_captured_and_replayed_bit_exact(R, n, V):
    """rec_dedup_plan_i64 + the segment sum captured once, replayed for three different id lists (one with a 40 % run of
    one id, as DIN's padding id): unique ids, run starts and the stable permutation bit-exact against numpy; the sums
    against fp64.  n = 1,273,856 is the table gradient of DIN config E (4096 x (8 + 3 + 300))."""
    ops = R.ops
    E = 4
    r = H.rng(n % 1000 + 7)
    ids_buf = torch.zeros(n, dtype=torch.int64, device="cuda")
    vals_buf = torch.zeros((n, E), dtype=torch.float32, device="cuda")

    def make(kind):
        if kind == "uniform":
            ids = r.integers(0, V, size=n)
        elif kind == "zipf":
            ids = np.minimum(r.zipf(1.05, size=n) - 1, V - 1)
        else:
            ids = r.integers(0, V, size=n)
            ids[r.uniform(size=n) < 0.4] = 0
        return ids.astype(np.int64), r.normal(size=(n, E)).astype(np.float32)

    ids0, vals0 = make("uniform")
    ids_buf.copy_(dev(ids0)); vals_buf.copy_(dev(vals0))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):                      # warm-up outside the capture (allocator, module load)
        ops.DedupPlan(ids_buf, V).segment_sum(vals_buf, E)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        plan = ops.DedupPlan(ids_buf, V)
        out = plan.segment_sum(vals_buf, E)
    for kind in ("uniform", "zipf", "padded"):
        ids, vals = make(kind)
        ids_buf.copy_(dev(ids)); vals_buf.copy_(dev(vals))
        g.replay()
        g.replay()                                   # a second replay over the same buffers must not disturb anything
        torch.cuda.synchronize()
        uid = np.unique(ids)
        nu = int(plan.n_uniq.item())
        assert nu == uid.size
        assert np.array_equal(plan.uniq_ids.cpu().numpy()[:nu], uid)
        order = np.argsort(ids, kind="stable")
        assert np.array_equal(plan.perm.cpu().numpy(), order.astype(np.int32))
        starts = np.flatnonzero(np.concatenate([[True], ids[order][1:] != ids[order][:-1]]))
        seg = plan.seg_start.cpu().numpy()
        assert np.array_equal(seg[:nu], starts.astype(np.int32)) and np.all(seg[nu:] == n)
        ref = np.add.reduceat(vals[order].astype(np.float64), starts, axis=0)
        got = out.cpu().numpy()
        assert np.abs(got[:nu] - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())
        assert np.all(got[nu:] == 0)


# ---------------------------------------------------------------------------------------------------------------
# tables larger than 4 GiB (configs D: 100M x 64d = 25.6 GB, E: 50M x 32d = 6.4 GB): row addressing
# ---------------------------------------------------------------------------------------------------------------
def _hash_rows(ids, E):
    """Row r of the synthetic big table: element c = ((r * 2654435761 + c * 40503 + 12345) mod 1000003) / 1000003 - 0.5,
    exactly representable steps in fp32 arithmetic done in int64 first (the same bits wherever it is evaluated)."""
    r = ids.reshape(-1, 1).to(torch.int64)
    c = torch.arange(E, device=ids.device, dtype=torch.int64).reshape(1, -1)
    return (((r * 2654435761 + c * 40503 + 12345) % 1000003).to(torch.float32) / 1000003.0) - 0.5


def _big_table(V, E, chunk=5_000_000):
    tab = torch.empty((V, E), dtype=torch.float32, device="cuda")
    for lo in range(0, V, chunk):
        hi = min(V, lo + chunk)
        tab[lo:hi] = _hash_rows(torch.arange(lo, hi, device="cuda"), E)
    return tab


@pytest.mark.parametrize("V,E", [(100_000_000, 64), (50_000_000, 32)])
def test_gather_addresses_rows_beyond_4gib(R, V, E):
    """rec_emb_gather_f32 on the tables of configs D (25.6 GB) and E (6.4 GB): every returned row is recomputed on the
    device from its id (element = a hash of (row, column)) -- bit exact, including ids whose byte offset needs more than
    32 bits, the last row, and ids drawn around the 4 GiB / 8 GiB / 16 GiB boundaries."""
    tab = _big_table(V, E)
    r = H.rng(V % 1000 + E)
    ids = r.integers(0, V, size=200_000)
    row_bytes = 4 * E
    edges = []
    for gib in (4, 8, 16, 24):
        b = (gib << 30) // row_bytes
        edges += [b - 2, b - 1, b, b + 1]
    edges = np.array([e for e in edges if 0 <= e < V] + [0, V - 1], dtype=np.int64)
    ids = np.concatenate([ids, edges])
    assert (ids.astype(np.int64) * row_bytes).max() > (4 << 30)
    idx = torch.from_numpy(ids).cuda()
    got = R.ops.emb_gather(tab, idx)
    assert torch.equal(got, _hash_rows(idx, E))
    # the 2-D form the layers use ([B, F] ids -> [B, F, E]) and the strided form (a row stride larger than E)
    X = idx[:199_998].reshape(-1, 3)
    assert torch.equal(R.ops.emb_gather(tab, X), _hash_rows(X, E).reshape(X.shape[0], 3, E))
    del tab, got
    torch.cuda.empty_cache()


def test_din_attention_addresses_rows_beyond_4gib(R):
    """rec_din_attn_fwd_f32 / _bwd_f32 gather their keys inside the kernel: on the 50M x 32d table of config E (6.4 GB) the
    scores, the pooled vectors and the key gradients must equal, bit for bit, those of the same call on a COMPACT table
    that holds only the touched rows (ids remapped) -- i.e. every key was read from the right row of the big table."""
    ops = R.ops
    V, E, C, B, T_ = 50_000_000, 32, 3, 64, 100
    D, Hh = C * E, 36
    tab = _big_table(V, E)
    r = H.rng(77)
    series = r.integers(1, V, size=(B, T_, C))
    series[:, :, 0] = np.where(r.random((B, T_)) < 0.3, 0, series[:, :, 0])      # padding positions (id 0)
    series[0, 0] = [V - 1, (4 << 30) // (4 * E), (4 << 30) // (4 * E) + 1]       # the last row, the 4-GiB boundary
    uniq, inv = np.unique(np.concatenate([[0], series.reshape(-1)]), return_inverse=True)
    assert uniq[0] == 0
    small = _hash_rows(torch.from_numpy(uniq).cuda(), E).contiguous()
    ser_small = torch.from_numpy(inv[1:].reshape(B, T_, C).astype(np.int64)).cuda()
    ser_big = torch.from_numpy(series.astype(np.int64)).cuda()
    g = torch.Generator(device="cuda").manual_seed(5)
    Mext = torch.randn((B, D * Hh + Hh), device="cuda", generator=g) * 0.1
    Wkd = torch.randn((D, Hh), device="cuda", generator=g) * 0.1
    alpha = torch.rand(Hh, device="cuda", generator=g)
    w2 = torch.randn(Hh, device="cuda", generator=g)
    b2 = torch.zeros(1, device="cuda")
    gp = torch.randn((B, D), device="cuda", generator=g)
    kind = ops.DACT_CODE["prelu"]
    for mask_valid in (0, 1):
        outs = []
        for table, ser in ((tab, ser_big), (small, ser_small)):
            sc, po = ops.din_attn_fwd(table, ser, Mext, Wkd, kind, alpha, None, None, w2, b2, 0, mask_valid)
            bw = ops.din_attn_bwd(table, ser, Mext, Wkd, kind, alpha, None, None, w2, b2, 0, mask_valid, sc, gp)
            outs.append((sc, po) + tuple(bw))
        for a, b in zip(*outs):
            assert torch.equal(a, b)
    del tab
    torch.cuda.empty_cache()
