"""GPU tests of the ModelManager mirror (2.FM/ModelManager.py:61-119,156-241): BASELINE config A -- FM through the
manager on DataGenerator-contract synthetic batches, batch 256 -- trained for a few steps and compared, parameter by
parameter, with the oracle's train_loop restatement (Keras BCE + Keras Adam with the dense sweep on the tables);
plus the layer factory's strings and error, and a smoke pass over every layer family.
"""
import numpy as np
import pytest
import torch

from oracle import layers_np as L
from oracle import torch_ref as T

pytestmark = pytest.mark.gpu

NAMES = ["user_tag1", "user_tag2", "item_tag1", "item_tag2", "item_tag3"]     # 2.FM/ModelManager.py:13


def test_config_a_fm_through_manager_matches_oracle_training():
    from explicit_tf2_recommendation_amd import data
    from explicit_tf2_recommendation_amd.model_manager import ModelManager
    V, E, B, lr = 5547, 16, 256, 0.01
    mm = ModelManager(feature_names=NAMES, data_info=data.data_info(V, len(NAMES)), embedding_dims=E, lr=lr, batch=B,
                      layer="fm_ranking")
    assert mm.feature_dims == V
    with torch.no_grad():
        mm.layer.embed.embeddings.mul_(8.0)           # larger than the U(-0.05,0.05) init: exercises the 2nd order
    params = {k: v.detach().cpu().numpy().copy() for k, v in mm.layer.named_parameters()}
    m = {k: np.zeros_like(v) for k, v in params.items()}
    v_ = {k: np.zeros_like(v) for k, v in params.items()}
    gen = data.SyntheticGenerator(NAMES, V, dist="zipf", seed=3)
    for t in range(1, 4):
        batch = gen.batch(B)
        X = L.index_assemble(batch, NAMES)
        tp = {k: torch.from_numpy(val).double().requires_grad_() for k, val in params.items()}
        p = {"embed": tp["embed.embeddings"], "w": tp["w.embeddings"], "bias": tp["bias"]}
        loss_ref = T.keras_bce(torch.from_numpy(batch["label"]).double(), T.fm_forward(p, torch.from_numpy(X)))
        loss_ref.backward()
        ids = np.unique(X)
        for k in params:
            g = tp[k].grad.numpy().astype(np.float32)
            if k == "bias":
                params[k], m[k], v_[k] = L.adam_dense_step(params[k], m[k], v_[k], g, t, lr=lr)
            else:
                params[k], m[k], v_[k] = L.adam_sparse_keras_step(params[k], m[k], v_[k], ids, g[ids], t, lr=lr)
        loss = mm.train_loop(dict(batch))
        assert abs(loss.item() - loss_ref.item()) <= 2e-5
        for k, q in mm.layer.named_parameters():
            assert np.abs(q.detach().cpu().numpy() - params[k]).max() <= 5e-5, (k, t)
    res = mm._metric_result()
    assert 0.0 <= res["auc"] <= 1.0 and np.isfinite(res["loss"])


def test_make_layer_choice_strings_and_error():
    from explicit_tf2_recommendation_amd import data, layers
    from explicit_tf2_recommendation_amd.model_manager import ModelManager
    info = data.data_info(2000, 5)
    kinds = {"fm_ranking": layers.FMRankingLayer, "deepfm_ranking": layers.DeepFMRankingLayer,
             "dssm_double_tower": layers.DSSMTwoTowerRetrievalLayer}
    for name, cls in kinds.items():
        assert isinstance(ModelManager(feature_names=NAMES, data_info=info, layer=name).layer, cls)
    with pytest.raises(ValueError):
        ModelManager(feature_names=NAMES, data_info=info, layer="no_such_layer")


def test_every_layer_family_trains():
    from explicit_tf2_recommendation_amd import data
    from explicit_tf2_recommendation_amd.model_manager import ModelManager
    B = 128
    # DeepFM, two-tower
    for layer in ("deepfm_ranking", "dssm_double_tower"):
        V = 3000
        mm = ModelManager(feature_names=NAMES, data_info=data.data_info(V, 5), embedding_dims=8, lr=0.01, batch=B,
                          layer=layer, adam_sparse_mode="lazy")
        gen = data.SyntheticGenerator(NAMES, V, dist="zipf", seed=1)
        r = mm.train_step([gen.batch(B) for _ in range(4)])
        assert np.isfinite(r["loss"]) and np.isfinite(mm.eval_step([gen.batch(B)])["loss"])
    # DCN, both cross modes (3.DCN/ModelManager.py:69-71)
    cat = ["uid", "iid", "utag1", "utag2", "utag3", "utag4", "itag1", "itag2", "itag3", "itag4"]
    cont = ["itag4_origin", "itag4_square", "itag4_cube"]
    for kind in ("vec", "matrix"):
        V = 4000
        mm = ModelManager(feature_names=cat, continuous_features=cont, data_info=data.data_info(V, 10),
                          embedding_dims=8, lr=0.01, batch=B, layer="dcn_ranking", model_params={"type": kind})
        gen = data.SyntheticGenerator(cat, V, continuous=cont, seed=2)
        assert np.isfinite(mm.train_step([gen.batch(B) for _ in range(3)])["loss"])
    # DIN (5.DIN/ModelManager.py:72-73)
    user = ["uid", "utag1", "utag2", "utag3", "utag4"]
    item = ["i_goods_id", "i_shop_id", "i_cate_id"]
    ser = ["visited_goods_ids", "visited_shop_ids", "visited_cate_ids"]
    V = 5000
    mm = ModelManager(feature_names=user + item, behavior_series_features=ser, data_info=data.data_info(V, 11),
                      embedding_dims=8, lr=0.01, batch=B, layer="din_layer",
                      model_params={"user_and_context_categorical_features": user, "item_categorical_features": item,
                                    "behavior_series_features": ser})
    gen = data.SyntheticGenerator(user + item, V, series=ser, seq_len=12, seed=4)
    r1 = mm.train_step([gen.batch(B) for _ in range(3)])
    assert np.isfinite(r1["loss"])


def test_din_train_loop_adds_l2_on_used_rows():
    """5.DIN/ModelManager.py:176-192: loss = BCE + regularization_factor * l2_loss(embed[unique ids of the batch]);
    the table gradient gains regularization_factor * embed[u] once per used row."""
    from explicit_tf2_recommendation_amd import data, functional as Fn
    from explicit_tf2_recommendation_amd.model_manager import ModelManager
    from oracle import layers_np as L
    user = ["uid", "utag1", "utag2", "utag3", "utag4"]
    item = ["i_goods_id", "i_shop_id", "i_cate_id"]
    ser = ["visited_goods_ids", "visited_shop_ids", "visited_cate_ids"]
    V, B, E = 5000, 96, 8
    kw = dict(feature_names=user + item, behavior_series_features=ser, data_info=data.data_info(V, 11),
              embedding_dims=E, lr=0.01, batch=B, layer="din_layer",
              model_params={"user_and_context_categorical_features": user, "item_categorical_features": item,
                            "behavior_series_features": ser})
    mm = ModelManager(regularization_factor=0.01, **kw)
    with torch.no_grad():
        mm.layer.embed.embeddings.mul_(10.0)
    batch = data.SyntheticGenerator(user + item, V, series=ser, seq_len=12, seed=4).batch(B)
    inputs = mm._to_device(dict(batch))
    target = inputs.pop("label")
    table = mm.layer.embed.embeddings.detach().cpu().numpy()
    all_ids = np.concatenate([batch[f].reshape(-1) for f in user + item + ser])
    l2_ref, uniq_ref, rows_ref = L.l2_used_rows(table.astype(np.float64), all_ids, 0.01, np.float64)
    # the term alone: value and gradient through the C ABI
    term = mm.used_rows_l2(inputs)
    assert abs(term.item() - l2_ref) <= 1e-6 * max(1.0, abs(l2_ref))
    term.backward()
    g = mm.layer.embed.embeddings.grad.coalesce()
    gi, gv = g.indices()[0].cpu().numpy(), g.values().cpu().numpy()
    keep = np.abs(gv).sum(1) > 0
    assert np.array_equal(gi[keep], uniq_ref[np.abs(rows_ref).sum(1) > 0])
    dense = np.zeros_like(table, dtype=np.float64)
    dense[gi] = gv
    assert np.abs(dense[uniq_ref] - rows_ref).max() <= 1e-7
    mm.layer.embed.embeddings.grad = None
    # inside the loop: total loss = BCE + term
    logits = mm.model(inputs)
    bce = mm.loss(target, logits["output"]).item()
    mm2 = ModelManager(regularization_factor=0.01, **kw)
    mm2.model.load_state_dict(mm.model.state_dict())
    total = mm2.train_loop(dict(batch)).item()
    assert abs(total - (bce + l2_ref)) <= 2e-5 * max(1.0, abs(total))
    mm3 = ModelManager(regularization_factor=0.0, **kw)
    mm3.model.load_state_dict(mm.model.state_dict())
    assert abs(mm3.train_loop(dict(batch)).item() - bce) <= 2e-5


def test_model_manager_trains_from_tfrecord_files(tmp_path):
    """The reference's loop end to end on its own file format: DataGenerator-style label encoding -> TFRecord files +
    data_info.json -> ModelManager.init_dataset -> train_step / eval_step (2.FM/ModelManager.py:122-153, 156-241)."""
    import json
    from explicit_tf2_recommendation_amd import tfrecord as TR
    from explicit_tf2_recommendation_amd.model_manager import ModelManager
    r = np.random.default_rng(5)
    n = 900
    raw = {k: r.integers(0, m, n) for k, m in zip(NAMES, (3, 27, 40, 300, 60))}
    enc, rec, info = TR.label_encode_columns(raw)
    score = (enc["user_tag1"] % 2 + enc["item_tag1"] % 3) / 3.0
    labels = (r.random(n) < 0.15 + 0.5 * score / score.max()).astype(np.float32)
    dtype = np.where(np.arange(n) < 700, "train", "test")
    out = str(tmp_path / "generated")
    TR.write_dataset(out, "fm", enc, labels, dtype, NAMES)
    json.dump(info, open(out + "/data_info.json", "w"))
    from explicit_tf2_recommendation_amd import layers
    layers.set_init_seed(17)
    torch.manual_seed(17)
    mm = ModelManager(feature_names=NAMES, json_path=out + "/data_info.json", embedding_dims=8, lr=0.01, batch=100,
                      layer="fm_ranking", epochs=1)
    assert mm.feature_dims == info[2]
    assert sum(len(b["label"]) for b in mm.init_dataset("train", out)) == 700
    assert sum(len(b["label"]) for b in mm.init_dataset("test", out)) == 200
    first = mm.eval_step(mm.init_dataset("train", out))["loss"]
    for _ in range(6):
        mm.train_step(list(mm.init_dataset("train", out)))
    last = mm.eval_step(mm.init_dataset("train", out))["loss"]
    assert np.isfinite(last) and last < first                  # the loss on the training files goes down
    assert np.isfinite(mm.eval_step(mm.init_dataset("test", out))["loss"])


def test_compiled_train_loop_fresh_batches_neither_recapture_nor_grow():
    """2.FM/ModelManager.py:171-199 through the drop-in: ModelManager(layer='deepfm_ranking') compiles the train loop
    (engine.DeepFMFusedStep, Keras Adam evaluated lazily and exactly, steps replayed from hipGraphs) and feeds it through a
    fixed ring of staging buffers -- 1000 FRESH host batches must not capture more graphs than the ring has forms, must
    not grow device memory, and must train (loss falls); metrics are read back once, at the end."""
    from explicit_tf2_recommendation_amd import data
    from explicit_tf2_recommendation_amd.model_manager import ModelManager
    V, B = 20000, 512
    names = ["C%d" % i for i in range(8)]
    mm = ModelManager(feature_names=names, data_info=data.data_info(V, len(names)), embedding_dims=16, lr=0.01, batch=B,
                      layer="deepfm_ranking")
    gen = data.SyntheticGenerator(names, V, dist="zipf", seed=5)
    w_true = np.random.default_rng(0).normal(size=V).astype(np.float32)

    def make():
        b = gen.batch(B)
        X = L.index_assemble(b, names)
        b["label"] = (w_true[X].sum(1, keepdims=True) > 0).astype(np.float32)     # a learnable target
        return b

    first = [mm.train_loop(make()) for _ in range(40)]
    assert mm._eng[0] == "fused"
    torch.cuda.synchronize()
    n_graphs, m0 = len(mm._eng[1]._graphs), torch.cuda.memory_allocated()
    assert 1 <= n_graphs <= 4                                # (slot pairs) x (plan-ring halves)
    ds = [make() for _ in range(1000)]
    res = mm.train_step(ds)
    torch.cuda.synchronize()
    assert len(mm._eng[1]._graphs) <= 4 and len(mm._eng[1]._seen) <= 8
    assert torch.cuda.memory_allocated() - m0 < (48 << 20)   # the metric history of 1000 steps is ~4 MB; nothing else grows
    assert np.isfinite(res["loss"]) and res["loss"] < float(first[0].item()) and res["auc"] > 0.6
    mm._eng[1].check_flags()
    # the same iterations through the eager autograd path give the same parameters (Keras Adam, dense sweep)
    mm2 = ModelManager(feature_names=names, data_info=data.data_info(V, len(names)), embedding_dims=16, lr=0.01, batch=B,
                       layer="deepfm_ranking", engine="eager")
    mm3 = ModelManager(feature_names=names, data_info=data.data_info(V, len(names)), embedding_dims=16, lr=0.01, batch=B,
                       layer="deepfm_ranking")
    mm3.model.load_state_dict(mm2.model.state_dict())
    few = [make() for _ in range(6)]
    for b in few:
        l2 = mm2.train_loop(dict(b))
        l3 = mm3.train_loop(dict(b))
        assert abs(l2.item() - l3.item()) <= 2e-5
    mm3.sync_parameters()
    for (k, p2), (_, p3) in zip(mm2.model.named_parameters(), mm3.model.named_parameters()):
        assert (p2 - p3).abs().max().item() <= 5e-5, k


@pytest.mark.parametrize("layer", ["fm_ranking", "dssm_double_tower"])
def test_compiled_train_loop_other_layers_replay_one_graph(layer):
    """Layers without a hand-fused step: forward + Keras BCE + autograd backward replay from ONE hipGraph
    (engine.GraphedTrainStep) behind the same train_loop; Keras Adam follows eagerly.  Same parameters as the eager path."""
    from explicit_tf2_recommendation_amd import data
    from explicit_tf2_recommendation_amd.model_manager import ModelManager
    V, B = 5547, 256
    kw = dict(feature_names=NAMES, data_info=data.data_info(V, len(NAMES)), embedding_dims=16, lr=0.01, batch=B, layer=layer)
    a, b = ModelManager(engine="eager", **kw), ModelManager(**kw)
    b.model.load_state_dict(a.model.state_dict())
    gen = data.SyntheticGenerator(NAMES, V, dist="zipf", seed=9)
    for _ in range(5):
        batch = gen.batch(B)
        la, lb = a.train_loop(dict(batch)), b.train_loop(dict(batch))
        assert abs(la.item() - lb.item()) <= 1e-6
    assert b._eng[0] == "graphed"
    for (k, p), (_, q) in zip(a.model.named_parameters(), b.model.named_parameters()):
        assert torch.equal(p, q), k
    ra, rb = a._metric_result(), b._metric_result()
    assert abs(ra["loss"] - rb["loss"]) <= 1e-6 and abs(ra["auc"] - rb["auc"]) <= 1e-9


@pytest.mark.gpu
def test_block_copy_and_auc_histogram_kernels():
    """rec_block_copy (n device arrays -> consecutive slices of one block; 16-byte and 4-byte paths) and
    rec_auc_hist_update_f32 (Keras streaming-AUC buckets + loss sum) against numpy."""
    import ctypes as C
    from explicit_tf2_recommendation_amd import ops
    from explicit_tf2_recommendation_amd.model_manager import ModelManager
    r = np.random.default_rng(3)
    st = ops._stream()
    for n_el, dt, off in ((8192, torch.int64, 0), (1000, torch.float32, 0), (1001, torch.float32, 1), (300, torch.int32, 3)):
        srcs = []
        for i in range(27 if dt == torch.int64 else 300):
            base = torch.from_numpy(r.integers(0, 1 << 30, size=n_el + 8)).to(dt).cuda()
            srcs.append(base[off:off + n_el])               # off != 0: not 16-byte aligned
        out = torch.empty((len(srcs), n_el), dtype=dt, device="cuda")
        arr = (C.c_void_p * len(srcs))(*[s.data_ptr() for s in srcs])
        ops.check(ops.lib.rec_block_copy(arr, len(srcs), n_el * out.element_size(), ops._ptr(out), st), "rec_block_copy")
        assert torch.equal(out, torch.stack(srcs))
    T = ModelManager.AUC_THRESHOLDS
    thr = ModelManager._auc_thresholds().astype(np.float32)
    n = 70001
    p = r.random(n).astype(np.float32)
    p[:50] = thr[r.integers(0, T, size=50)]                 # predictions that sit exactly on a threshold
    p[50:60] = 0.0
    p[60:70] = 1.0
    y = (r.random(n) < 0.3).astype(np.float32)
    losses = r.random(5).astype(np.float32)
    hist = torch.zeros(2 * (T + 1), dtype=torch.int64, device="cuda")
    acc = torch.zeros(1, dtype=torch.float64, device="cuda")
    dp, dy, dl, dthr = (torch.from_numpy(x).cuda() for x in (p, y, losses, thr))
    for _ in range(2):                                       # accumulates
        ops.check(ops.lib.rec_auc_hist_update_f32(ops._ptr(dp), ops._ptr(dy), n, ops._ptr(dthr), T, ops._ptr(hist),
                                                  ops._ptr(dl), 5, ops._ptr(acc), st), "rec_auc_hist_update_f32")
    k = np.searchsorted(thr, p, side="left")                 # thresholds strictly below p
    want = np.zeros((2, T + 1), np.int64)
    np.add.at(want, ((y > 0.5).astype(np.int64), k), 1)
    assert np.array_equal(hist.cpu().numpy().reshape(2, -1), 2 * want)
    assert abs(acc.item() - 2 * float(losses.astype(np.float64).sum())) <= 1e-12
