"""Oracle self-checks for the f4 sibling layers (PNN inner product, NFM bi-interaction + BatchNormalization, SIM GSU
inner-product attention).  The reference holds no outputs for these layers (parity unpinned), so -- as for FM/DCN/DIN --
the two independent restatements (numpy closed form with hand-derived backward vs torch-CPU op-for-op with autograd)
must agree, plus hand-checkable micro cases on the reference's docstring inputs (2.FM/CustomLayers.py:699-703).
"""
import numpy as np
import pytest
import torch

from oracle import layers_np as L
from oracle import torch_ref as T
from tests import helpers as H


def _ids(seed, B, F, V):
    return H.rng(seed).integers(0, V, size=(B, F)).astype(np.int64)


def test_pair_order_is_boolean_mask_order():
    """SharedFieldsInteraction keeps the strict upper triangle in row-major order (2.FM/CustomLayers.py:764-771);
    the explicit double loop of the older OpnLayer walks the same order (:662-665)."""
    for F in (2, 3, 5, 26):
        mask = np.triu(np.ones((F, F)), 1) > 0
        assert [tuple(x) for x in np.argwhere(mask)] == L.pair_list(F)
    assert L.pair_list(1) == []


def test_pnn_docstring_micro_case():
    """PNNLayer docstring inputs (2.FM/CustomLayers.py:699-703): 5 fields, ids 0..19, the hand-checkable table."""
    names = ["user_tag0", "user_tag1", "item_tag1", "item_tag2", "item_tag3"]
    ins = {"item_tag1": np.array([0, 1, 2, 3]), "item_tag2": np.array([4, 5, 6, 7]),
           "item_tag3": np.array([8, 9, 10, 11]), "user_tag0": np.array([12, 13, 14, 15]),
           "user_tag1": np.array([16, 17, 18, 19])}
    X = L.index_assemble(ins, names)
    E = 8
    tab = H.det_table(20, E)
    out = L.ipn_forward(tab, X, np.float64)
    assert out.shape == (4, 5 * E + 10)
    for b in range(4):
        rows = [tab[X[b, f]].astype(np.float64) for f in range(5)]
        assert np.array_equal(out[b, :5 * E], np.concatenate(rows))
        k = 5 * E
        for i in range(5):
            for j in range(i + 1, 5):
                assert abs(out[b, k] - float(rows[i] @ rows[j])) < 1e-14
                k += 1
    t = T.pnn_combined(torch.from_numpy(tab).double(), torch.from_numpy(X))
    assert np.abs(t.numpy() - out).max() < 1e-14


@pytest.mark.parametrize("F,E", [(1, 4), (2, 3), (5, 16), (26, 16)])
def test_ipn_np_vs_torch(F, E):
    V, B = 200, 17
    tab = H.rng(1).normal(size=(V, E)).astype(np.float32) * 0.3
    X = _ids(2, B, F, V)
    tt = torch.from_numpy(tab).double().requires_grad_(True)
    out_t = T.pnn_combined(tt, torch.from_numpy(X))
    out_n = L.ipn_forward(tab, X, np.float64)
    assert np.abs(out_n - out_t.detach().numpy()).max() < 1e-13
    g = H.rng(3).normal(size=out_n.shape)
    (out_t * torch.from_numpy(g)).sum().backward()
    vals = L.ipn_backward_vals(tab, X, g, np.float64)
    dense = np.zeros((V, E))
    np.add.at(dense, X.reshape(-1), vals)
    assert np.abs(dense - tt.grad.numpy()).max() < 1e-12


def test_bi_interaction_np_vs_torch_and_identity():
    V, B, F, E = 150, 33, 10, 16
    tab = H.rng(4).normal(size=(V, E)).astype(np.float32) * 0.2
    X = _ids(5, B, F, V)
    tt = torch.from_numpy(tab).double().requires_grad_(True)
    out_t = T.bi_interaction(tt, torch.from_numpy(X))
    out_n = L.bi_interaction_forward(tab, X, np.float64)
    assert np.abs(out_n - out_t.detach().numpy()).max() < 1e-13
    # the pooling is the sum over pairs of the element-wise products (He & Chua, NFM eq. 4)
    e = tab[X].astype(np.float64)
    brute = sum(e[:, i] * e[:, j] for i in range(F) for j in range(i + 1, F))
    assert np.abs(out_n - brute).max() < 1e-13
    g = H.rng(6).normal(size=out_n.shape)
    (out_t * torch.from_numpy(g)).sum().backward()
    vals = L.bi_interaction_backward_vals(tab, X, g, np.float64)
    dense = np.zeros((V, E))
    np.add.at(dense, X.reshape(-1), vals)
    assert np.abs(dense - tt.grad.numpy()).max() < 1e-12


def test_batchnorm_np_vs_torch():
    B, N = 64, 19
    r = H.rng(7)
    x = r.normal(size=(B, N)) * 3 + 1
    gamma, beta = r.uniform(0.5, 1.5, N), r.normal(size=N)
    mm, mv = r.normal(size=N), r.uniform(0.5, 2, N)
    for training in (True, False):
        xt = torch.from_numpy(x).requires_grad_(True)
        gt = torch.from_numpy(gamma).requires_grad_(True)
        bt = torch.from_numpy(beta).requires_grad_(True)
        yt = T.batchnorm(xt, gt, bt, torch.from_numpy(mm), torch.from_numpy(mv), training)
        yn, nm, nv = L.batchnorm_forward(x, gamma, beta, mm, mv, training, dt=np.float64)
        assert np.abs(yn - yt.detach().numpy()).max() < 1e-12
        # torch's own batch_norm (unbiased=False statistics for normalisation) agrees
        yb = torch.nn.functional.batch_norm(torch.from_numpy(x), torch.from_numpy(mm.copy()), torch.from_numpy(mv.copy()),
                                            torch.from_numpy(gamma), torch.from_numpy(beta), training, 0.01, 1e-3)
        assert np.abs(yn - yb.numpy()).max() < 1e-12
        if training:
            assert np.allclose(nm, 0.99 * mm + 0.01 * x.mean(0), atol=1e-14)
            assert np.allclose(nv, 0.99 * mv + 0.01 * x.var(0), atol=1e-14)      # biased variance (non-fused Keras path)
            g = r.normal(size=(B, N))
            (yt * torch.from_numpy(g)).sum().backward()
            gx, gg, gb = L.batchnorm_backward(x, gamma, g, dt=np.float64)
            assert np.abs(gx - xt.grad.numpy()).max() < 1e-12
            assert np.abs(gg - gt.grad.numpy()).max() < 1e-12
            assert np.abs(gb - bt.grad.numpy()).max() < 1e-12
        else:
            assert np.array_equal(nm, mm) and np.array_equal(nv, mv)


def test_nfm_np_vs_torch():
    V, B, F, E, NC = 120, 40, 10, 16, 3
    pr = H.nfm_params(8, V, E, NC)
    X = _ids(9, B, F, V)
    xc = H.rng(10).normal(size=(B, NC)).astype(np.float32)
    tp = H.to_torch(pr, torch.float64)
    for training in (True, False):
        o_t = T.nfm_forward(tp, torch.from_numpy(X), torch.from_numpy(xc).double(), training)
        o_n = L.nfm_forward(pr, X, xc, training, np.float64)
        assert o_n.shape == (B, 1)
        assert np.abs(o_n - o_t.numpy()).max() < 1e-12


def test_ip_attention_np_vs_torch():
    V, B, T_, C, E = 90, 9, 11, 3, 8
    r = H.rng(11)
    tab = r.normal(size=(V, E)).astype(np.float32) * 0.3
    series = r.integers(1, V, size=(B, T_, C)).astype(np.int64)
    lens = r.integers(0, T_ + 1, size=B)
    for b in range(B):
        series[b, lens[b]:, :] = 0                                 # padded_batch pads every series with 0
    q = r.normal(size=(B, C * E))
    tt = torch.from_numpy(tab).double().requires_grad_(True)
    qt = torch.from_numpy(q).requires_grad_(True)
    s_t, p_t = T.ip_attention(tt, qt, torch.from_numpy(series), 0)
    s_n, p_n = L.ip_attention_forward(tab, q, series, 0, np.float64)
    assert np.abs(s_n - s_t.detach().numpy()).max() < 1e-13
    assert np.abs(p_n - p_t.detach().numpy()).max() < 1e-13
    assert np.all(s_n[np.arange(T_)[None, :] >= lens[:, None]] == 0)
    g = r.normal(size=p_n.shape)
    (p_t * torch.from_numpy(g)).sum().backward()
    gk, gq = L.ip_attention_backward(tab, q, series, g, 0, np.float64)
    assert np.abs(gq - qt.grad.numpy()).max() < 1e-12
    dense = np.zeros((V, E))
    np.add.at(dense, series.reshape(-1), gk.reshape(-1, E))
    assert np.abs(dense - tt.grad.numpy()).max() < 1e-12
    assert np.all(gk[np.arange(T_)[None, :] >= lens[:, None]] == 0)


def test_ip_attention_hand_case():
    """One example, two valid steps and one padded, E=2, C=1: by hand."""
    tab = np.array([[9., 9.], [1., 2.], [3., -1.]])
    series = np.array([[[1], [2], [0]]])
    q = np.array([[2., 1.]])
    s, p = L.ip_attention_forward(tab, q, series, 0, np.float64)
    assert np.array_equal(s, [[4., 5., 0.]])                      # <q,k1> = 4, <q,k2> = 5, padded -> 0
    assert np.array_equal(p, [[4 * 1 + 5 * 3, 4 * 2 - 5]])


def test_inner_product_network_docstring_case():
    """InnerProductNetwork docstring input (2.FM/CustomLayers.py:603-606): arange(24).reshape(2,3,4); by hand."""
    x = torch.arange(24, dtype=torch.float64).reshape(2, 3, 4)
    out = T.ipn(x).numpy()
    assert np.array_equal(out[0], [38., 62., 214.])              # <r0,r1>, <r0,r2>, <r1,r2>
    assert np.array_equal(out[1], [12 * 16 + 13 * 17 + 14 * 18 + 15 * 19, 12 * 20 + 13 * 21 + 14 * 22 + 15 * 23,
                                   16 * 20 + 17 * 21 + 18 * 22 + 19 * 23])


@pytest.mark.parametrize("F,E", [(1, 4), (2, 3), (3, 16), (5, 8)])
def test_ffm_np_loop_form_vs_torch_vectorised_form(F, E):
    """The two forms the reference holds -- FFMRankingLayer's double loop over F tables and FFMLayer's vectorised
    FieldAwareInteractionLayer -- give the same numbers; the numpy oracle follows the first, the torch one the second."""
    V, B = 60, 21
    r = H.rng(F * 10 + E)
    v = r.normal(size=(V, F, E)).astype(np.float32) * 0.3
    w = r.normal(size=(V, 1)).astype(np.float32)
    bias = np.array([0.3], np.float32)
    X = _ids(3, B, F, V)
    tp = H.to_torch({"v": v, "w": w, "bias": bias}, torch.float64, True)
    z_t = T.ffm_logit(tp, torch.from_numpy(X))
    prob, z = L.ffm_forward(v, w, bias, X, np.float64)
    assert np.abs(z - z_t.detach().numpy()).max() < 1e-12
    assert np.abs(prob - T.ffm_forward(tp, torch.from_numpy(X)).detach().numpy()).max() < 1e-12
    gz = r.normal(size=(B, 1))
    (z_t * torch.from_numpy(gz)).sum().backward()
    rows, wrows, gb = L.ffm_backward(v, X, gz, np.float64)
    dense = np.zeros((V, F, E))
    np.add.at(dense, X.reshape(-1), rows)
    assert np.abs(dense - tp["v"].grad.numpy()).max() < 1e-12
    dw = np.zeros((V, 1))
    np.add.at(dw, X.reshape(-1), wrows)
    assert np.abs(dw - tp["w"].grad.numpy()).max() < 1e-12
    assert np.abs(gb - tp["bias"].grad.numpy()).max() < 1e-12


def test_ffm_docstring_micro_case():
    """FFMRankingLayer docstring inputs (2.FM/CustomLayers.py:372-374), hand-checkable tables, expected by hand."""
    ins = {"item_tag1": np.array([0, 1, 2, 3]), "item_tag2": np.array([4, 5, 6, 7]), "item_tag3": np.array([8, 9, 10, 11])}
    X = L.index_assemble(ins, ["item_tag1", "item_tag2", "item_tag3"])
    V, F, E = 20, 3, 4
    v = H.det_table(V, F * E).reshape(V, F, E)
    w = H.det_table(V, 1)
    bias = np.array([0.25], np.float32)
    prob, z = L.ffm_forward(v, w, bias, X, np.float64)
    for b in range(4):
        x = X[b]
        zz = 0.25 + sum(float(w[x[f], 0]) for f in range(3))
        zz += float(v[x[0], 1].astype(np.float64) @ v[x[1], 0]) + float(v[x[0], 2].astype(np.float64) @ v[x[2], 0]) \
            + float(v[x[1], 2].astype(np.float64) @ v[x[2], 1])
        assert abs(z[b, 0] - zz) < 1e-12
