"""KAT-3: the reference's TRAINED DeepFM weights (2.FM/ranking_model/checkpoint/ckpt-2, extracted by
scripts/make_golden_fm.py into tests/golden/fm_ckpt2_weights.npz) as the value range of the FM / DeepFM parity tests.

The checkpoint holds weights, not outputs, so nothing here is a reference-pinned answer ("parity unpinned" for FM /
DeepFM stays as DESIGN.md says); what the fixture adds is that oracle and HIP path agree on REAL table rows (|x| up to
1.7, first-order weights up to 1.9, trained 80->32->8 kernels) and on the reference's own field layout, instead of only
on U(-0.05, 0.05) initialisations.  CPU part: the two independent restatements of the oracle agree on them.  GPU part:
FMRankingLayer / DeepFMRankingLayer and the fused train step against the fp64 oracle, outputs 1e-5, gradients 2e-5.
"""
import os

import numpy as np
import pytest
import torch

from oracle import layers_np as L
from oracle import torch_ref as T
from tests import helpers as H

NAMES = ["user_tag1", "user_tag2", "item_tag1", "item_tag2", "item_tag3"]       # 2.FM/ModelManager.py:13


def _kat(golden_dir):
    return np.load(os.path.join(golden_dir, "fm_ckpt2_weights.npz"))


def _params(k, seed=5):
    r = H.rng(seed)
    return {"embed": k["embed"], "w": k["w"], "bias": k["bias"], "k1": [k["k0"], k["k1"]], "b1": [k["b0"], k["b1"]],
            "k2": [H.glorot(r, 8, 1)], "b2": [r.uniform(-0.1, 0.1, size=(1,)).astype(np.float32)]}


def _batch(k, B, seed, model="deepfm"):
    """ids inside each field's own range (DataGenerator contract); labels drawn from the trained model's own
    probabilities.  (Independent coin-flip labels against a TRAINED model put examples at p ~ 1 with y = 0: there Keras'
    clip constant 1 - 1e-7 is 1 - 1.19e-7 in fp32, the term log(1 - p + 1e-7) differs by 0.09 between fp32 -- what TF
    computes -- and the fp64 oracle, and a single such example moves the mean loss by 1e-4; tests/test_gpu_ops.py
    test_bce_saturated_* pins that corner against the fp32 restatement instead.)"""
    r = H.rng(seed)
    ins = {n: (k["field_offsets"][f] + r.integers(0, k["field_dims"][f], size=(B, 1))).astype(np.int64)
           for f, n in enumerate(NAMES)}
    p = _params(k)
    X = L.index_assemble(ins, NAMES)
    prob = (L.fm_forward(p["embed"], p["w"], p["bias"], X, np.float64)[0] if model == "fm" else
            L.deepfm_forward(p, X, np.float64)[0])
    ins["label"] = (r.uniform(size=(B, 1)) < prob).astype(np.float32)
    return ins


def test_fixture_is_the_checkpoints_shape(golden_dir):
    k = _kat(golden_dir)
    assert k["embed"].shape == (5547, 16) and k["w"].shape == (5547, 1) and k["bias"].shape == (1,)
    assert k["k0"].shape == (80, 32) and k["k1"].shape == (32, 8)
    assert int(k["field_dims"].sum()) == 5547 and np.array_equal(np.cumsum(k["field_dims"])[:-1], k["field_offsets"][1:])
    assert np.abs(k["embed"]).max() > 0.5            # trained: far outside the U(-0.05, 0.05) initialisation


@pytest.mark.parametrize("model", ["fm", "deepfm"])
def test_oracle_restatements_agree_on_trained_weights(golden_dir, model):
    k = _kat(golden_dir)
    p = _params(k)
    b = _batch(k, 300, 1, model)
    X = L.index_assemble(b, NAMES)
    tp = H.to_torch(p, torch.float64, True)
    if model == "fm":
        p_np, z_np = L.fm_forward(p["embed"], p["w"], p["bias"], X, np.float64)
        out = T.fm_forward(tp, torch.from_numpy(X))
    else:
        p_np, z_np = L.deepfm_forward(p, X, np.float64)
        out = T.deepfm_forward(tp, torch.from_numpy(X))
    assert np.abs(p_np - out.detach().numpy()).max() < 1e-12
    assert np.abs(z_np).max() < 15                   # the parity data keeps |z| < 15
    loss = T.keras_bce(torch.from_numpy(b["label"]).double(), out)
    assert abs(loss.item() - L.bce_forward(b["label"], p_np, np.float64)) < 1e-12


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _close(a, b, tol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() <= tol * max(1e-3, np.abs(b).max())


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["fm", "deepfm"])
def test_layers_on_trained_weights(golden_dir, model):
    import explicit_tf2_recommendation_amd as R
    k = _kat(golden_dir)
    p = _params(k)
    B = 1024
    b = _batch(k, B, 2, model)
    X = L.index_assemble(b, NAMES)
    if model == "fm":
        layer = R.layers.FMRankingLayer(feature_names=NAMES, feature_dims=5547, embedding_dims=16).cuda()
        mp = {"bias": p["bias"], "embed.embeddings": p["embed"], "w.embeddings": p["w"]}
    else:
        layer = R.layers.DeepFMRankingLayer(feature_names=NAMES, feature_dims=5547, embedding_dims=16).cuda()
        mp = {"bias": p["bias"], "embed.embeddings": p["embed"], "w.embeddings": p["w"],
              "MLP_layer1.kernel_0": p["k1"][0], "MLP_layer1.bias_0": p["b1"][0], "MLP_layer1.kernel_1": p["k1"][1],
              "MLP_layer1.bias_1": p["b1"][1], "MLP_layer2.kernel_0": p["k2"][0], "MLP_layer2.bias_0": p["b2"][0]}
    sd = dict(layer.named_parameters())
    assert set(sd) == set(mp)
    with torch.no_grad():
        for n, a in mp.items():
            sd[n].copy_(torch.from_numpy(a))
    out = layer({n: dev(b[n]) for n in NAMES})["output"]
    tp = H.to_torch(p, torch.float64, True)
    o64 = (T.fm_forward if model == "fm" else T.deepfm_forward)(tp, torch.from_numpy(X))
    assert np.abs(out.detach().cpu().numpy() - o64.detach().numpy()).max() <= 1e-5
    loss = R.functional.KerasBCE.apply(out, dev(b["label"]))
    loss.backward()
    lt = T.keras_bce(torch.from_numpy(b["label"]).double(), o64)
    lt.backward()
    assert abs(loss.item() - lt.item()) <= 1e-5
    g = lambda q: (q.grad.to_dense() if q.grad.is_sparse else q.grad).cpu().numpy()
    assert _close(g(layer.embed.embeddings), tp["embed"].grad.numpy(), 2e-5)
    assert _close(g(layer.w.embeddings), tp["w"].grad.numpy(), 2e-5)
    assert _close(g(layer.bias), tp["bias"].grad.numpy(), 2e-5)
    if model == "deepfm":
        assert _close(g(layer.MLP_layer1.kernel_0), tp["k1"][0].grad.numpy(), 2e-5)
        assert _close(g(layer.MLP_layer1.kernel_1), tp["k1"][1].grad.numpy(), 2e-5)


@pytest.mark.gpu
def test_fused_step_on_trained_weights(golden_dir):
    """The headline kernel (engine.DeepFMFusedStep: fused forward+backward, direct-mode value rows) on the reference's
    trained tables and its own field layout (5 fields, 5547 ids)."""
    from explicit_tf2_recommendation_amd import engine, layers
    k = _kat(golden_dir)
    p = _params(k)
    B = 2048
    b = _batch(k, B, 3)
    layer = layers.DeepFMRankingLayer(feature_names=NAMES, feature_dims=5547, embedding_dims=16).cuda()
    sd = dict(layer.named_parameters())
    mp = {"bias": p["bias"], "embed.embeddings": p["embed"], "w.embeddings": p["w"],
          "MLP_layer1.kernel_0": p["k1"][0], "MLP_layer1.bias_0": p["b1"][0], "MLP_layer1.kernel_1": p["k1"][1],
          "MLP_layer1.bias_1": p["b1"][1], "MLP_layer2.kernel_0": p["k2"][0], "MLP_layer2.bias_0": p["b2"][0]}
    with torch.no_grad():
        for n, a in mp.items():
            sd[n].copy_(torch.from_numpy(a))
    step = engine.DeepFMFusedStep(layer, B, [int(x) for x in k["field_dims"]], [int(x) for x in k["field_offsets"]],
                                  use_graph=False)
    loss = step({n: dev(v) for n, v in b.items()}).item()
    step.check_flags()
    X = L.index_assemble(b, NAMES)
    tp = H.to_torch(p, torch.float64, True)
    lt = T.keras_bce(torch.from_numpy(b["label"]).double(), T.deepfm_forward(tp, torch.from_numpy(X)))
    lt.backward()
    assert abs(loss - lt.item()) <= 1e-5
    gr = step.gradients()
    ids, rows, nu = gr["embed.embeddings"]
    nu = int(nu.item())
    touched = np.unique(X)
    assert np.array_equal(ids.cpu().numpy()[:nu], touched)                 # bit exact, ascending
    assert _close(rows.cpu().numpy()[:nu], tp["embed"].grad.numpy()[touched], 2e-5)
    assert _close(gr["w.embeddings"][1].cpu().numpy()[:nu], tp["w"].grad.numpy()[touched], 2e-5)
    assert _close(gr["MLP_layer1.kernel_0"].cpu().numpy(), tp["k1"][0].grad.numpy(), 2e-5)
    assert _close(gr["bias"].cpu().numpy(), tp["bias"].grad.numpy(), 2e-5)
