"""GPU tests of the train-step engine (explicit-tf2-recommendation_amd/engine.py): the fixed C-ABI call sequence
must produce the same loss and gradients as the oracle's train_loop restatement (2.FM/ModelManager.py:171-181),
eagerly and when replayed from a captured hipGraph, and the optimizer variants must match the Keras-Adam oracle.
"""
import numpy as np
import pytest
import torch

from oracle import layers_np as L
from oracle import torch_ref as T
from tests import helpers as H

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def make(B, F, E, V, seed, dist):
    from explicit_tf2_recommendation_amd import layers, data
    names = ["f%d" % i for i in range(F)]
    layers.set_init_seed(seed)
    layer = layers.DeepFMRankingLayer(feature_names=names, feature_dims=V, embedding_dims=E, mlp_dims=[32, 8]).cuda()
    with torch.no_grad():   # non-zero biases so that every gradient path is exercised
        for n, p in layer.named_parameters():
            if "bias_" in n:
                p.uniform_(-0.1, 0.1)
        layer.embed.embeddings.mul_(6.0)
    gen = data.SyntheticGenerator(names, V, dist=dist, seed=seed)
    return layer, names, gen


def oracle_grads(layer, names, batch):
    pr = {k: v.detach().cpu().double().requires_grad_() for k, v in layer.named_parameters()}
    p = {"embed": pr["embed.embeddings"], "w": pr["w.embeddings"], "bias": pr["bias"],
         "k1": [pr["MLP_layer1.kernel_0"], pr["MLP_layer1.kernel_1"]],
         "b1": [pr["MLP_layer1.bias_0"], pr["MLP_layer1.bias_1"]],
         "k2": [pr["MLP_layer2.kernel_0"]], "b2": [pr["MLP_layer2.bias_0"]]}
    X = torch.from_numpy(L.index_assemble(batch, names))
    loss = T.keras_bce(torch.from_numpy(batch["label"]).double(), T.deepfm_forward(p, X))
    loss.backward()
    return loss.item(), {k: v.grad.numpy() for k, v in pr.items()}


def close(a, b, tol=2e-5):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() <= tol * max(1e-30, np.abs(b).max())


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("B,F,E,V,dist", [(256, 5, 16, 5547, "zipf"), (2048, 26, 16, 1000000, "uniform")])
def test_deepfm_step_gradients(use_graph, B, F, E, V, dist):
    from explicit_tf2_recommendation_amd import engine, data
    layer, names, gen = make(B, F, E, V, 3, dist)
    step = engine.DeepFMTrainStep(layer, B, optimizer=None, use_graph=use_graph)
    for it in range(3):                                    # replayed graphs must track new batches
        batch = gen.batch(B)
        dbatch = data.to_device(batch)
        loss = step(dbatch)
        if use_graph:
            loss = step(dbatch)                            # second call on the same tensors = pure replay
        ref_loss, ref = oracle_grads(layer, names, batch)
        assert abs(loss.item() - ref_loss) <= 1e-5 * max(1, abs(ref_loss))
        g = step.gradients()
        for name in ("MLP_layer1.kernel_0", "MLP_layer1.bias_0", "MLP_layer1.kernel_1", "MLP_layer1.bias_1",
                     "MLP_layer2.kernel_0", "MLP_layer2.bias_0", "bias"):
            assert close(g[name].cpu().numpy(), ref[name]), name
        for name in ("embed.embeddings", "w.embeddings"):
            ids, rows, nu = g[name]
            nu = int(nu.item())
            ids = ids.cpu().numpy()[:nu]
            touched = np.unique(L.index_assemble(batch, names))
            assert np.array_equal(ids, touched)            # bit exact, ascending
            assert close(rows.cpu().numpy()[:nu], ref[name][touched]), name
        assert step.oob.item() == 0


def test_engine_matches_autograd_path():
    from explicit_tf2_recommendation_amd import engine, data, functional
    B, F, E, V = 512, 7, 8, 3000
    layer, names, gen = make(B, F, E, V, 5, "zipf")
    batch = gen.batch(B)
    dbatch = data.to_device(batch)
    step = engine.DeepFMTrainStep(layer, B, use_graph=False)
    loss_e = step(dbatch).item()
    out = layer({k: dbatch[k] for k in names})["output"]
    loss_a = functional.KerasBCE.apply(out, dbatch["label"])
    loss_a.backward()
    assert abs(loss_e - loss_a.item()) <= 1e-6
    g = step.gradients()
    for name, p in layer.named_parameters():
        if name in ("embed.embeddings", "w.embeddings"):
            ids, rows, nu = g[name]
            nu = int(nu.item())
            dense = p.grad.to_dense()
            assert torch.equal(rows[:nu], dense[ids[:nu]])          # same kernels, same order: bitwise
        else:
            assert close(g[name].cpu().numpy(), p.grad.cpu().numpy(), 1e-6), name


@pytest.mark.parametrize("opt", ["keras_adam", "lazy_adam"])
def test_deepfm_train_steps_with_adam(opt):
    """Three optimizer steps against the numpy restatement of Keras Adam (dense sweep for the tables)."""
    from explicit_tf2_recommendation_amd import engine, data
    B, F, E, V = 256, 5, 16, 2000
    layer, names, gen = make(B, F, E, V, 7, "zipf")
    lr = 0.01
    params = {k: v.detach().cpu().numpy().copy() for k, v in layer.named_parameters()}
    m = {k: np.zeros_like(v) for k, v in params.items()}
    v_ = {k: np.zeros_like(v) for k, v in params.items()}
    step = engine.DeepFMTrainStep(layer, B, optimizer=opt, lr=lr, use_graph=False)
    for t in range(1, 4):
        batch = gen.batch(B)
        # oracle step on the oracle's own copy of the parameters (fp32 restatement, fp64 gradients)
        tp = {k: torch.from_numpy(val).double().requires_grad_() for k, val in params.items()}
        p = {"embed": tp["embed.embeddings"], "w": tp["w.embeddings"], "bias": tp["bias"],
             "k1": [tp["MLP_layer1.kernel_0"], tp["MLP_layer1.kernel_1"]],
             "b1": [tp["MLP_layer1.bias_0"], tp["MLP_layer1.bias_1"]],
             "k2": [tp["MLP_layer2.kernel_0"]], "b2": [tp["MLP_layer2.bias_0"]]}
        X = L.index_assemble(batch, names)
        T.keras_bce(torch.from_numpy(batch["label"]).double(), T.deepfm_forward(p, torch.from_numpy(X))).backward()
        for k in params:
            g = tp[k].grad.numpy().astype(np.float32)
            if k in ("embed.embeddings", "w.embeddings"):
                ids = np.unique(X)
                if opt == "keras_adam":
                    params[k], m[k], v_[k] = L.adam_sparse_keras_step(params[k], m[k], v_[k], ids, g[ids], t, lr=lr)
                else:
                    params[k], m[k], v_[k] = L.adam_rows_step(params[k], m[k], v_[k], ids, g[ids], t, lr=lr)
            else:
                params[k], m[k], v_[k] = L.adam_dense_step(params[k], m[k], v_[k], g, t, lr=lr)
        step(data.to_device(batch))
        for k, q in layer.named_parameters():
            assert np.abs(q.detach().cpu().numpy() - params[k]).max() <= 5e-5, (k, t)
