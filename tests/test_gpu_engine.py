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
    torch.manual_seed(1000 + seed)          # the random biases below must not depend on the order tests run in
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


def close(a, b, tol=2e-5, floor=1e-3):
    """max-abs error <= tol * max|reference|.  A one-element gradient (the FM bias, the last bias) is a sum of B
    signed terms of size ~1/B that can cancel to ~0, so the scale never drops below `floor` (i.e. an absolute 2e-8:
    a few fp32 ulps of the terms that were added)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err, scale = np.abs(a - b).max(), max(floor, np.abs(b).max())
    if err > tol * scale:
        print("close(): max-abs error %.3e vs scale %.3e (ratio %.2e > tol %.1e)" % (err, scale, err / scale, tol))
    return err <= tol * scale


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


# ------------------------------------------------------------------------------------------------
# the 4-launch fused step (csrc/deepfm_fused.hip)
# ------------------------------------------------------------------------------------------------
def make16(B, F, V, seed, dist):
    from explicit_tf2_recommendation_amd import layers, data
    names = ["f%d" % i for i in range(F)]
    layers.set_init_seed(seed)
    layer = layers.DeepFMRankingLayer(feature_names=names, feature_dims=V, embedding_dims=16, mlp_dims=[32, 8]).cuda()
    torch.manual_seed(2000 + seed)
    with torch.no_grad():
        for n, p in layer.named_parameters():
            if "bias_" in n:
                p.uniform_(-0.1, 0.1)
        # keep |z| moderate (test data keeps |z| < 15): with 26 fields a 6x table saturates the
        # sigmoid and fp32 p*(1-p) -- in TF as much as here -- drifts from the fp64 oracle by more than the tolerance
        layer.embed.embeddings.mul_(6.0 if F <= 8 else 2.0)
    gen = data.SyntheticGenerator(names, V, dist=dist, seed=seed)
    return layer, names, gen


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("B,F,V,dist", [(256, 5, 5547, "zipf"), (8192, 26, 1000000, "uniform"), (1000, 3, 300, "zipf"),
                                         (2048, 26, 200000, "zipf"), (33, 7, 4000, "uniform"), (64, 1, 500, "zipf"),
                                         (96, 2, 900, "uniform"), (160, 28, 60000, "zipf"), (70, 27, 60000, "uniform"),
                                         (16384, 26, 2000000, "uniform"),
                                         (8192, 26, 10000000, "uniform"),      # BASELINE.json metric config, full size
                                         (8192, 26, 10000000, "zipf")])
def test_fused_step_matches_oracle(use_graph, B, F, V, dist):
    from explicit_tf2_recommendation_amd import engine, data
    layer, names, gen = make16(B, F, V, 11, dist)
    step = engine.DeepFMFusedStep(layer, B, gen.dims, gen.offsets, use_graph=use_graph)
    for it in range(2):
        batch = gen.batch(B)
        dbatch = data.to_device(batch)
        loss = step(dbatch)
        if use_graph:                                        # eager, eager + capture, replay
            loss = step(dbatch)
            loss = step(dbatch)
            assert len(step._graphs) == it + 1
        step.check_flags()
        ref_loss, ref = oracle_grads(layer, names, batch)
        assert abs(loss.item() - ref_loss) <= 1e-5 * max(1, abs(ref_loss))
        g = step.gradients()
        for name in ("MLP_layer1.kernel_0", "MLP_layer1.bias_0", "MLP_layer1.kernel_1", "MLP_layer1.bias_1",
                     "MLP_layer2.kernel_0", "MLP_layer2.bias_0", "bias"):
            assert close(g[name].cpu().numpy(), ref[name]), name
        touched = np.unique(L.index_assemble(batch, names))
        for name in ("embed.embeddings", "w.embeddings"):
            ids, rows, nu = g[name]
            nu = int(nu.item())
            assert np.array_equal(ids.cpu().numpy()[:nu], touched)            # bit exact, ascending
            assert close(rows.cpu().numpy()[:nu], ref[name][touched]), name
            assert np.all(rows.cpu().numpy()[nu:] == 0) and np.all(ids.cpu().numpy()[nu:] == touched[0])


@pytest.mark.parametrize("use_graph", [False, True])
def test_fused_step_pipelined_plan(use_graph):
    """``next_inputs``: the plan of batch k+1 is built (second stream) while batch k is differentiated; every step
    must still deliver batch k's own gradients -- also when an announced batch is NOT the one that follows."""
    from explicit_tf2_recommendation_amd import engine, data
    B, F, V = 2048, 26, 300000
    layer, names, gen = make16(B, F, V, 23, "zipf")
    step = engine.DeepFMFusedStep(layer, B, gen.dims, gen.offsets, use_graph=use_graph)
    host = [gen.batch(B) for _ in range(3)]
    devb = [data.to_device(b) for b in host]
    refs = [oracle_grads(layer, names, b) for b in host]
    order = [(0, 1), (1, 2), (2, 0), (0, 1), (1, 2), (0, 2), (2, None), (1, 0)]   # (batch, announced next batch)
    for cur, nxt in order * (3 if use_graph else 1):         # graphs: first sighting eager, second captured, then replays
        loss = step(devb[cur], next_inputs=devb[nxt] if nxt is not None else None)
        step.check_flags()
        ref_loss, ref = refs[cur]
        assert abs(loss.item() - ref_loss) <= 1e-5 * max(1, abs(ref_loss)), (cur, nxt)
        g = step.gradients()
        touched = np.unique(L.index_assemble(host[cur], names))
        ids, rows, nu = g["embed.embeddings"]
        nu = int(nu.item())
        assert np.array_equal(ids.cpu().numpy()[:nu], touched), (cur, nxt)
        assert close(rows.cpu().numpy()[:nu], ref["embed.embeddings"][touched]), (cur, nxt)
        assert close(g["w.embeddings"][1].cpu().numpy()[:nu], ref["w.embeddings"][touched]), (cur, nxt)
        assert close(g["MLP_layer1.kernel_0"].cpu().numpy(), ref["MLP_layer1.kernel_0"]), (cur, nxt)


def test_fused_step_multi_step_graph_equals_single_steps():
    """``many()``: several consecutive iterations captured as ONE hipGraph.  What is left in the buffers afterwards
    must be bit-identical to running the same batches one call at a time -- for every prefix length, with and without
    a ``then`` batch, entered with or without a prefetched plan."""
    from explicit_tf2_recommendation_amd import engine, data
    B, F, V = 1024, 26, 200000
    layer, names, gen = make16(B, F, V, 29, "zipf")
    devb = [data.to_device(gen.batch(B)) for _ in range(4)]
    one = engine.DeepFMFusedStep(layer, B, gen.dims, gen.offsets, use_graph=False)
    multi = engine.DeepFMFusedStep(layer, B, gen.dims, gen.offsets, use_graph=True)

    def snapshot(step):
        g = step.gradients()
        nu = int(g["embed.embeddings"][2].item())
        out = {k: v.clone() for k, v in g.items() if torch.is_tensor(v)}
        out["ids"] = g["embed.embeddings"][0][:nu].clone()
        out["rows"] = g["embed.embeddings"][1][:nu].clone()
        out["wrows"] = g["w.embeddings"][1][:nu].clone()
        out["loss"] = step.loss.clone()
        return out

    # `then` = None, one batch, or the list of batches of the next call (all of their plans are built beside this call)
    cases = (([0, 1, 2, 3], 0), ([0, 1, 2, 3], 0), ([2, 1], None), ([3], 1), ([1, 0, 2], None),
             ([0, 1], [2, 3]), ([2, 3], [0, 1]), ([0, 1], [3, 2, 1]), ([3, 1, 2], [0]), ([0, 0], [0, 0]), ([0, 0], None))
    for seq, then in cases * 3:                              # first sighting eager, second captured, third replayed
        for i in seq:
            one(devb[i])
        want = snapshot(one)
        ann = None if then is None else (devb[then] if isinstance(then, int) else [devb[i] for i in then])
        multi.many([devb[i] for i in seq], then=ann)
        multi.check_flags()
        got = snapshot(multi)
        for k in want:
            assert torch.equal(want[k], got[k]), (seq, then, k)
    assert len(multi._graphs) >= 4                           # the replay path was exercised


def test_fused_step_fresh_batches_neither_recapture_nor_grow():
    """An input pipeline that hands over NEW tensors every batch: with a train step inside the graph (lazy Adam) every
    unseen set of addresses runs eagerly -- no device synchronisation, no capture, nothing retained -- and the graph
    cache is a bounded LRU; a pipeline that cycles a ring of staging buffers is captured once per slot and replayed."""
    from explicit_tf2_recommendation_amd import engine, data
    B, F, V = 256, 5, 5547
    layer, names, gen = make16(B, F, V, 31, "zipf")
    step = engine.DeepFMFusedStep(layer, B, gen.dims, gen.offsets, optimizer="lazy_adam", lr=0.01, use_graph=True)
    step.MAX_GRAPHS = 4
    keep = []                                                # keeps every batch alive: all addresses are distinct
    for i in range(40):
        b = data.to_device(gen.batch(B))
        keep.append(b)
        step(b)
    assert len(step._graphs) == 0 and len(step._seen) <= 8 * step.MAX_GRAPHS and step.t == 40
    torch.cuda.synchronize()
    m0 = torch.cuda.memory_allocated()
    ring = keep[:8]                                          # a ring of 8 staging buffers, refilled in place
    fresh = [gen.batch(B) for _ in range(8)]
    for rnd in range(6):
        for slot, b in enumerate(ring):
            src = fresh[(slot + rnd) % 8]
            for k in b:
                b[k].copy_(torch.from_numpy(src[k]))
            step(b)
        assert len(step._graphs) <= step.MAX_GRAPHS
    assert len(step._graphs) == step.MAX_GRAPHS              # 8 slots, 4 graphs: the least recently used were dropped
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() - m0 < (8 << 20)    # nothing accumulates beyond the bounded cache
    step.check_flags()
    # one lazy-optimizer step per layer: the table padding holds its state
    with pytest.raises(ValueError):
        engine.DeepFMFusedStep(layer, B, gen.dims, gen.offsets, optimizer="lazy_adam", use_graph=False)
    step.release()
    engine.DeepFMFusedStep(layer, B, gen.dims, gen.offsets, optimizer="lazy_adam", use_graph=False)
    with pytest.raises(ValueError):
        engine.DeepFMFusedStep(layer, B, gen.dims, gen.offsets, optimizer="adamw")
    with pytest.raises(ValueError):
        engine.DeepFMFusedStep(layer, B, gen.dims, gen.offsets, optimizer="keras_adam_lazy", direct=False)


def test_fused_step_equals_generic_step_and_is_deterministic():
    from explicit_tf2_recommendation_amd import engine, data
    B, F, V = 4096, 26, 500000
    layer, names, gen = make16(B, F, V, 13, "zipf")
    batch = data.to_device(gen.batch(B))
    fused = engine.DeepFMFusedStep(layer, B, gen.dims, gen.offsets, use_graph=False)
    generic = engine.DeepFMTrainStep(layer, B, use_graph=False)
    l1, l2 = fused(batch).item(), generic(batch).item()
    assert abs(l1 - l2) <= 1e-6
    g1, g2 = fused.gradients(), generic.gradients()
    nu = int(g1["embed.embeddings"][2].item())
    assert nu == int(g2["embed.embeddings"][2].item())
    assert torch.equal(g1["embed.embeddings"][0][:nu], g2["embed.embeddings"][0][:nu])
    assert close(g1["embed.embeddings"][1][:nu].cpu().numpy(), g2["embed.embeddings"][1][:nu].cpu().numpy(), 1e-5)
    assert close(g1["MLP_layer1.kernel_0"].cpu().numpy(), g2["MLP_layer1.kernel_0"].cpu().numpy(), 1e-5)
    snap = {k: (v[1].clone() if isinstance(v, tuple) else v.clone()) for k, v in g1.items()}
    fused(batch)
    for k, v in fused.gradients().items():
        assert torch.equal(v[1] if isinstance(v, tuple) else v, snap[k]), k       # run-to-run bit identical


def test_fused_step_notices_a_replaced_column():
    """The per-batch checks are cached per batch dict: replacing a tensor inside a dict that was seen before must be
    noticed -- a wrong dtype is refused, a new id column is used."""
    from explicit_tf2_recommendation_amd import engine, data
    B, F, V = 512, 5, 5547
    layer, names, gen = make16(B, F, V, 29, "zipf")
    step = engine.DeepFMFusedStep(layer, B, gen.dims, gen.offsets, use_graph=True)
    h0, h1 = gen.batch(B), gen.batch(B)
    d0 = data.to_device(h0)
    step(d0)
    step(d0)                                                 # second call: the cached path
    d0[names[2]] = d0[names[2]].to(torch.int32)
    with pytest.raises(ValueError):
        step(d0)
    h0[names[2]] = h1[names[2]]                              # another column of valid ids of the same field
    d0[names[2]] = data.to_device(h1)[names[2]]
    loss = step(d0)
    ref_loss, ref = oracle_grads(layer, names, h0)
    assert abs(loss.item() - ref_loss) <= 1e-5 * max(1, abs(ref_loss))
    ids, rows, nu = step.gradients()["embed.embeddings"]
    nu = int(nu.item())
    touched = np.unique(L.index_assemble(h0, names))
    assert np.array_equal(ids.cpu().numpy()[:nu], touched)
    assert close(rows.cpu().numpy()[:nu], ref["embed.embeddings"][touched])


def test_fused_step_flags_contract_violation():
    from explicit_tf2_recommendation_amd import engine, data
    B, F, V = 256, 5, 5547
    layer, names, gen = make16(B, F, V, 17, "uniform")
    step = engine.DeepFMFusedStep(layer, B, gen.dims, gen.offsets, use_graph=False)
    batch = gen.batch(B)
    batch[names[3]][7, 0] = gen.offsets[1]              # an id of field 1 in the column of field 3
    step(data.to_device(batch))
    with pytest.raises(ValueError):
        step.check_flags()


def test_fused_step_with_keras_adam_matches_generic():
    from explicit_tf2_recommendation_amd import engine, data
    B, F, V = 512, 6, 3000
    la, names, gen = make16(B, F, V, 19, "zipf")
    lb, _, _ = make16(B, F, V, 19, "zipf")
    lb.load_state_dict(la.state_dict())                  # the random biases of make16 differ between the two
    a = engine.DeepFMFusedStep(la, B, gen.dims, gen.offsets, optimizer="keras_adam", lr=0.01, use_graph=False)
    b = engine.DeepFMTrainStep(lb, B, optimizer="keras_adam", lr=0.01, use_graph=False)
    for _ in range(3):
        batch = data.to_device(gen.batch(B))
        a(batch)
        b(batch)
    for (k, p), (_, q) in zip(la.named_parameters(), lb.named_parameters()):
        assert np.abs(p.detach().cpu().numpy() - q.detach().cpu().numpy()).max() <= 2e-5, k


@pytest.mark.parametrize("dist", ["uniform", "zipf"])
def test_fused_step_lazy_adam_in_post_launch(dist):
    """the touched-rows Adam of both tables applied INSIDE the post launch (direct mode) against (a) the
    same update as separate launches over the finished row gradients (rec_adam_rows_f32: same arithmetic) and (b) the oracle's lazy Adam (L.adam_rows_step) on the oracle's own gradients, three steps."""
    from explicit_tf2_recommendation_amd import engine, data
    B, F, V = 2048, 26, 40000
    la, names, gen = make16(B, F, V, 31, dist)
    lb, _, _ = make16(B, F, V, 31, dist)
    lb.load_state_dict(la.state_dict())
    a = engine.DeepFMFusedStep(la, B, gen.dims, gen.offsets, optimizer="lazy_adam", lr=0.01, use_graph=False)
    b = engine.DeepFMFusedStep(lb, B, gen.dims, gen.offsets, optimizer="lazy_adam", lr=0.01, use_graph=False,
                               direct=False)              # classic path: rows updated by rec_adam_rows_f32 afterwards
    assert a._fused_lazy() and not b._fused_lazy()
    emb = la.embed.embeddings.detach().cpu().numpy().copy()
    w = la.w.embeddings.detach().cpu().numpy().copy()
    st = {"e": (emb.copy(), np.zeros_like(emb), np.zeros_like(emb)), "w": (w.copy(), np.zeros_like(w), np.zeros_like(w))}
    for t in range(1, 4):
        host = gen.batch(B)
        # oracle gradients at the CURRENT parameters of `a`, then the oracle's lazy Adam
        _, ref = oracle_grads(la, names, host)
        touched = np.unique(L.index_assemble(host, names))
        st["e"] = L.adam_rows_step(st["e"][0], st["e"][1], st["e"][2], touched, ref["embed.embeddings"][touched]
                                   .astype(np.float32), t, lr=0.01, dt=np.float32)
        st["w"] = L.adam_rows_step(st["w"][0], st["w"][1], st["w"][2], touched, ref["w.embeddings"][touched]
                                   .astype(np.float32), t, lr=0.01, dt=np.float32)
        batch = data.to_device(host)
        a(batch)
        b(batch)
        a.check_flags()
        # same formula in two kernels: the compiler contracts m*b1 + g*(1-b1) into an fma its own way in each, and
        # runs of more than 8 lookups are summed in another (fixed) order -- equal to rounding, not bit for bit
        for (k, p), (_, q) in zip(la.named_parameters(), lb.named_parameters()):
            assert (p - q).abs().max().item() <= 1e-6, (t, k)
        assert np.abs(la.embed.embeddings.detach().cpu().numpy() - st["e"][0]).max() <= 2e-5, t
        assert np.abs(la.w.embeddings.detach().cpu().numpy() - st["w"][0]).max() <= 2e-5, t
    assert (a.state["embed.embeddings"][0] - b.state["embed.embeddings"][0]).abs().max().item() <= 1e-6
    assert (a.state["w.embeddings"][1] - b.state["w.embeddings"][1]).abs().max().item() <= 1e-6


@pytest.mark.parametrize("dist,use_graph", [("uniform", False), ("zipf", False), ("zipf", True)])
def test_keras_adam_evaluated_lazily_equals_the_dense_sweep_bit_for_bit(dist, use_graph):
    """Keras' sparse apply sweeps EVERY row every step (2.FM/ModelManager.py:178-179).  'keras_adam_lazy' lets rows skip
    the sweeps and replays them -- with the sweep's own arithmetic -- when a batch is about to read the rows (or at
    flush()): after 9 steps over 5 different batches (rows touched once, repeatedly, with gaps, never) the tables, both
    moment arrays and the dense parameters must equal those of the dense-sweep implementation bit for bit; and a row the
    lazy path has NOT brought up to date must differ from it before flush() (the test would be vacuous otherwise)."""
    from explicit_tf2_recommendation_amd import engine, data
    B, F, V = 512, 6, 6000
    la, names, gen = make16(B, F, V, 41, dist)
    lb, _, _ = make16(B, F, V, 41, dist)
    lb.load_state_dict(la.state_dict())
    a = engine.DeepFMFusedStep(la, B, gen.dims, gen.offsets, optimizer="keras_adam", lr=0.01, use_graph=False)
    b = engine.DeepFMFusedStep(lb, B, gen.dims, gen.offsets, optimizer="keras_adam_lazy", lr=0.01, use_graph=use_graph)
    batches = [data.to_device(gen.batch(B)) for _ in range(5)]
    order = [0, 1, 2, 0, 3, 3, 4, 1, 0]
    for i in order:
        la_loss = a(batches[i]).item()
        lb_loss = b(batches[i]).item()
        assert la_loss == lb_loss, i
    ea, eb = la.embed.embeddings.detach(), lb.embed.embeddings.detach()
    stale = (b._last.cpu().numpy() < len(order)) & (b._last.cpu().numpy() > 0)
    assert stale.any() and not torch.equal(ea, eb)           # some touched rows are behind the sweep
    b.flush()
    assert int(b._last.min().item()) == len(order)
    assert torch.equal(ea, eb) and torch.equal(la.w.embeddings.detach(), lb.w.embeddings.detach())
    for k in ("embed.embeddings", "w.embeddings"):
        assert torch.equal(a.state[k][0], b.state[k][0].contiguous()), k
        assert torch.equal(a.state[k][1], b.state[k][1].contiguous()), k
    for (k, p), (_, q) in zip(la.named_parameters(), lb.named_parameters()):
        assert torch.equal(p, q), k
    # and the steps go on from the flushed state
    assert a(batches[2]).item() == b(batches[2]).item()


def test_fused_lazy_adam_train_steps_replayed_from_graphs_equal_eager_ones():
    """The whole train step with the lazy Adam inside the post launch holds no per-step host scalar (the step counter and
    the bias-corrected step size live on the device), so cycles of steps are captured and replayed: parameters and
    optimizer state after 12 steps -- first calls enqueued eagerly and captured, later ones replayed -- must equal the
    eagerly enqueued steps bit for bit."""
    from explicit_tf2_recommendation_amd import engine, data
    B, F, V = 1024, 26, 30000
    la, names, gen = make16(B, F, V, 37, "zipf")
    lb, _, _ = make16(B, F, V, 37, "zipf")
    lb.load_state_dict(la.state_dict())
    a = engine.DeepFMFusedStep(la, B, gen.dims, gen.offsets, optimizer="lazy_adam", lr=0.01, use_graph=False)
    b = engine.DeepFMFusedStep(lb, B, gen.dims, gen.offsets, optimizer="lazy_adam", lr=0.01, use_graph=True)
    batches = [data.to_device(gen.batch(B)) for _ in range(4)]
    for rep in range(3):                                     # the same two calls three times: capture, then replays
        for lo in (0, 2):
            cur, nxt = batches[lo:lo + 2], batches[(lo + 2) % 4:(lo + 2) % 4 + 2]
            la_loss = a.many(cur, then=nxt).item()
            lb_loss = b.many(cur, then=nxt).item()
            assert la_loss == lb_loss, (rep, lo)
    assert a.t == b.t == 12 and int(b._step_dev.item()) == 12
    for (k, p), (_, q) in zip(la.named_parameters(), lb.named_parameters()):
        assert torch.equal(p, q), k
    for k in a.state:
        assert torch.equal(a.state[k][0], b.state[k][0]) and torch.equal(a.state[k][1], b.state[k][1]), k
    assert len(b._graphs) >= 2 and len(a._graphs) == 0


@pytest.mark.parametrize("family", ["dssm", "dcn_matrix", "dcn_vec", "din"])
def test_graphed_train_step_equals_eager_autograd(family):
    """engine.GraphedTrainStep: forward + KerasBCE + autograd backward replayed from one hipGraph must give the eager
    path's loss and gradients on every batch it is fed (static input buffers, gradients in the graph's pool)."""
    import copy
    from explicit_tf2_recommendation_amd import engine, data, layers, functional as Fn
    B, V = 192, 4000
    layers.set_init_seed(41)
    if family == "dssm":
        un, inn = ["user_tag1", "user_tag2"], ["item_tag1", "item_tag2", "item_tag3"]
        layer = layers.DSSMTwoTowerRetrievalLayer(u_feature_names=un, i_feature_names=inn, u_feature_dims=V,
                                                  i_feature_dims=V).cuda()
        gen = data.SyntheticGenerator(un + inn, V, dist="zipf", seed=1)
    elif family.startswith("dcn"):
        cat, cont = ["c%d" % i for i in range(6)], ["x0", "x1"]
        layer = layers.DeepCrossNetworkLayer(categorical_features=cat, continuous_features=cont, feature_dims=V,
                                             embedding_dims=8, layer_num=2,
                                             type="matrix" if family == "dcn_matrix" else "vec").cuda()
        gen = data.SyntheticGenerator(cat, V, continuous=cont, dist="zipf", seed=2)
    else:
        user, item = ["uid", "utag1"], ["i_goods_id", "i_shop_id", "i_cate_id"]
        ser = ["visited_goods_ids", "visited_shop_ids", "visited_cate_ids"]
        layer = layers.DINLayer(user_and_context_categorical_features=user, item_categorical_features=item,
                                behavior_series_features=ser, feature_dims=V, embedding_dims=8).cuda()
        gen = data.SyntheticGenerator(user + item, V, series=ser, seq_len=9, seed=3)
    batches = [data.to_device(gen.batch(B)) for _ in range(3)]
    eager = copy.deepcopy(layer)
    step = engine.GraphedTrainStep(layer, batches[0])

    def eager_grads(b):
        for p in eager.parameters():
            p.grad = None
        out = eager({k: v for k, v in b.items() if k != "label"})["output"]
        y = b["label"]
        if out.dim() == 2 and out.shape[1] > 1:
            y = y.expand(-1, out.shape[1]).contiguous()
        loss = Fn.KerasBCE.apply(out, y)
        loss.backward()
        return loss.item(), [p.grad.to_dense().clone() if p.grad.is_sparse else p.grad.clone() for p in eager.parameters()]

    for b in (batches[1], batches[2], batches[0], batches[1]):
        loss = step(b).item()
        want_loss, want = eager_grads(b)
        assert abs(loss - want_loss) <= 1e-6 * max(1.0, abs(want_loss))
        for (name, p), w in zip(layer.named_parameters(), want):
            g = p.grad.to_dense() if p.grad.is_sparse else p.grad
            assert torch.equal(g, w), name                  # same kernels in the same order: bit identical
