#!/usr/bin/env python3
"""Secondary measurements for the other BASELINE.json configs (parity-test cases, NOT the bench.py headline):
forward + backward through the Layer mirror (autograd over the HIP kernels) on synthetic DataGenerator-contract
batches, one MI355X.  Prints one JSON line per config.  Usage:  python scripts/bench_configs.py [--graphed] [A B C C26 D E R P N FF G]

  A   FM, 5 fields, V=5547, E=16, B=256                       (BASELINE configs[0], via ModelManager)
  B   DeepFM, 26 fields, V=1M, E=16, B=8192                   (configs[1]; the 10M variant is bench.py)
  C   DCN matrix CrossNet, 10 cat + 3 cont, V=10M, E=32, L=3, B=16384 (D=323); C26 = 26 cat fields (D=835)
  D   DSSM two-tower, item V=100M x 64d on ONE GPU (25.6 GB table; the 8-way sharded form is sharded.py), B=8192
  E   DIN, T=100, V=50M, E=32, B=4096
  R   retrieval after the towers: 10M items x 8d (L2-normalised), 1024 user vectors, top-20
  P   PNN inner product (f4), 26 fields, V=10M, E=16, B=8192      N   NFM (f4), 10 cat + 3 cont, V=10M, E=16, B=16384
  FF  FFM (f4), 26 fields, V=10M rows of 26x16 floats (16.6 GB), B=8192
  G   SIM GSU inner-product attention (f4), T=100, V=50M, E=32, B=4096
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from explicit_tf2_recommendation_amd import layers, data, functional as Fn  # noqa: E402
from explicit_tf2_recommendation_amd.model_manager import ModelManager  # noqa: E402


def timed(fn, warmup, iters):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


GRAPHED = False      # --graphed: engine.GraphedTrainStep (forward + loss + autograd backward as one replayed hipGraph)


def fwd_bwd(layer, batch, names_all):
    if GRAPHED:
        from explicit_tf2_recommendation_amd import engine
        b = {k: batch[k] for k in list(names_all) + ["label"]}
        gstep = engine.GraphedTrainStep(layer, b)
        return lambda: gstep(b)
    ins = {k: batch[k] for k in names_all}

    def step():
        for p in layer.parameters():
            p.grad = None
        out = layer(ins)["output"]
        y = batch["label"]
        if out.dim() == 2 and out.shape[1] > 1:
            y = y.expand(-1, out.shape[1]).contiguous()
        Fn.KerasBCE.apply(out, y).backward()
    return step


def big_table_(emb):
    """Fill a very large table on the device (a host-side torch.rand of 25.6 GB would take minutes)."""
    with torch.no_grad():
        emb.uniform_(-0.05, 0.05)


def run(name):
    torch.cuda.empty_cache()
    layers.Layer.check_ids = False          # no per-call device->host read of the bounds flag while timing
    if name == "A":
        names = ["user_tag1", "user_tag2", "item_tag1", "item_tag2", "item_tag3"]
        V, B = 5547, 256
        mm = ModelManager(feature_names=names, data_info=data.data_info(V, 5), embedding_dims=16, lr=1e-3, batch=B,
                          layer="fm_ranking")
        gen = data.SyntheticGenerator(names, V, dist="zipf", seed=0)
        batches = [gen.batch(B) for _ in range(64)]                     # host batches, as the reference's pipeline delivers
        for _ in range(2):
            mm.train_step(batches)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mm.train_step(batches)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / len(batches)
        return {"config": "A FM via ModelManager.train_step (compiled loop: one hipGraph for fwd+bwd, Keras Adam incl. the "
                          "dense sweep, host batches staged)", "B": B, "V": V, "ms_per_step": dt * 1e3,
                "examples_per_s": B / dt}
    if name == "BM":
        # the drop-in against the engine called by hand: DeepFM (26 fields, 1M x 16d, B = 8192) trained through
        # ModelManager.train_step (2.FM/ModelManager.py:171-199: compiled loop, staging ring, Keras Adam evaluated lazily
        # and exactly, metrics read back once) vs engine.DeepFMFusedStep.many on resident batches, same optimizer
        from explicit_tf2_recommendation_amd import engine
        names = ["C%d" % i for i in range(26)]
        V, B = 1_000_000, 8192
        gen = data.SyntheticGenerator(names, V, seed=0)
        ds = [data.to_device(gen.batch(B)) for _ in range(64)]
        mm = ModelManager(feature_names=names, data_info=data.data_info(V, 26), embedding_dims=16, lr=1e-3, batch=B,
                          layer="deepfm_ranking")
        for _ in range(6):       # (every form of a chunk is captured at its second sighting: as many warm-up epochs as
            mm.train_step(ds)    # the engine called by hand gets below)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mm.train_step(ds)
        torch.cuda.synchronize()
        dt_mm = (time.perf_counter() - t0) / len(ds)
        layer = layers.DeepFMRankingLayer(feature_names=names, feature_dims=V, embedding_dims=16).cuda()
        st = engine.DeepFMFusedStep(layer, B, gen.dims, gen.offsets, optimizer="keras_adam_lazy", lr=1e-3)

        def direct():
            for i in range(0, 64, 4):
                st.many(ds[i:i + 4], then=ds[(i + 4) % 64:(i + 4) % 64 + 4])
        for _ in range(6):
            direct()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        direct()
        torch.cuda.synchronize()
        dt_en = (time.perf_counter() - t0) / 64
        return {"config": "BM DeepFM 1M x 16d train step (exact lazily evaluated Keras Adam): ModelManager.train_step vs "
                          "engine.DeepFMFusedStep.many", "B": B, "V": V, "ms_per_step": dt_mm * 1e3,
                "examples_per_s": B / dt_mm, "engine_direct_ms_per_step": dt_en * 1e3, "ratio": dt_mm / dt_en}
    if name == "B":
        names = ["C%d" % i for i in range(26)]
        V, B = 1_000_000, 8192
        layer = layers.DeepFMRankingLayer(feature_names=names, feature_dims=V, embedding_dims=16).cuda()
        batch = data.to_device(data.SyntheticGenerator(names, V, seed=0).batch(B))
        dt = timed(fwd_bwd(layer, batch, names), 5, 50)
        return {"config": "B DeepFM 1M x 16d (autograd path; the fused engine is bench.py)", "B": B, "V": V,
                "ms_per_step": dt * 1e3, "examples_per_s": B / dt}
    if name in ("C", "C26"):
        ncat = 10 if name == "C" else 26
        cat = ["c%d" % i for i in range(ncat)]
        cont = ["x0", "x1", "x2"]
        V, B, E, Lyr = 10_000_000, 16384, 32, 3
        layer = layers.DeepCrossNetworkLayer(categorical_features=cat, continuous_features=cont, feature_dims=V,
                                             embedding_dims=E, layer_num=Lyr, type="matrix")
        layer = layer.cuda()
        batch = data.to_device(data.SyntheticGenerator(cat, V, continuous=cont, seed=0).batch(B))
        D = 3 + ncat * E
        dt = timed(fwd_bwd(layer, batch, cat + cont), 3, 20)
        flops = 2.0 * B * D * D * Lyr * (1 + 3)          # fwd + (recompute U, dW, dX) in bwd
        return {"config": "%s DCN matrix CrossNet D=%d L=3" % (name, D), "B": B, "V": V, "ms_per_step": dt * 1e3,
                "examples_per_s": B / dt, "crossnet_gemm_flops_per_step": flops,
                "crossnet_tflops_if_all_time_were_gemm": flops / dt / 1e12}
    if name == "D":
        un, inn = ["user_tag1", "user_tag2"], ["item_tag1", "item_tag2", "item_tag3"]
        V, B, E = 100_000_000, 8192, 64
        layer = layers.DSSMTwoTowerRetrievalLayer(u_feature_names=un, i_feature_names=inn, u_feature_dims=1000,
                                                  i_feature_dims=1000, u_embedding_dims=E, i_embedding_dims=E)
        layer = layer.cuda()
        # swap in the big item table directly on the device
        layer.i_tower.embed.embeddings = torch.nn.Parameter(torch.empty((V, E), device="cuda"))
        layer.u_tower.embed.embeddings = torch.nn.Parameter(torch.empty((V // 10, E), device="cuda"))
        big_table_(layer.i_tower.embed.embeddings)
        big_table_(layer.u_tower.embed.embeddings)
        gi = data.SyntheticGenerator(inn, V, seed=0).batch(B)
        gu = data.SyntheticGenerator(un, V // 10, seed=1).batch(B)
        batch = data.to_device({**{k: gu[k] for k in un}, **{k: gi[k] for k in inn}, "label": gi["label"]})
        dt = timed(fwd_bwd(layer, batch, un + inn), 3, 20)
        return {"config": "D DSSM two-tower, item table 100M x 64d (25.6 GB) on one GPU", "B": B, "V": V,
                "ms_per_step": dt * 1e3, "examples_per_s": B / dt}
    if name in ("DS", "ES"):
        # configs D / E with their tables ROW-SHARDED (layers.*(sharded=True): de-duplicated fixed-capacity exchange) at
        # world size 1 -- the exchange code of N > 1 with the rank's own slab kept out of RCCL; target <= 1.3 x D / E
        import torch.distributed as dist
        from explicit_tf2_recommendation_amd import sharded
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29591")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        if name == "DS":
            un, inn = ["user_tag1", "user_tag2"], ["item_tag1", "item_tag2", "item_tag3"]
            V, B, E = 100_000_000, 8192, 64
            layer = layers.DSSMTwoTowerRetrievalLayer(u_feature_names=un, i_feature_names=inn, u_feature_dims=1000,
                                                      i_feature_dims=1000, u_embedding_dims=E, i_embedding_dims=E).cuda()
            layer.i_tower.embed = sharded.ShardedEmbedding(V, E, device="cuda")
            layer.u_tower.embed = sharded.ShardedEmbedding(V // 10, E, device="cuda")
            gi = data.SyntheticGenerator(inn, V, seed=0).batch(B)
            gu = data.SyntheticGenerator(un, V // 10, seed=1).batch(B)
            batch = data.to_device({**{k: gu[k] for k in un}, **{k: gi[k] for k in inn}, "label": gi["label"]})
            dt = timed(fwd_bwd(layer, batch, un + inn), 3, 20)
            return {"config": "DS DSSM two-tower, item table 100M x 64d row-sharded (world size 1)", "B": B, "V": V,
                    "ms_per_step": dt * 1e3, "examples_per_s": B / dt}
        user = ["uid", "utag1", "utag2", "utag3", "utag4"]
        item = ["i_goods_id", "i_shop_id", "i_cate_id"]
        ser = ["visited_goods_ids", "visited_shop_ids", "visited_cate_ids"]
        V, B, E, T = 50_000_000, 4096, 32, 100
        layer = layers.DINLayer(user_and_context_categorical_features=user, item_categorical_features=item,
                                behavior_series_features=ser, feature_dims=1000, embedding_dims=E).cuda()
        layer.embed = sharded.ShardedEmbedding(V, E, device="cuda")
        layer.feature_dims = V
        batch = data.to_device(data.SyntheticGenerator(user + item, V, series=ser, seq_len=T, seed=0).batch(B))
        dt = timed(fwd_bwd(layer, batch, user + item + ser), 2, 10)
        return {"config": "ES DIN T=100, 50M x 32d row-sharded (world size 1)", "B": B, "V": V, "ms_per_step": dt * 1e3,
                "examples_per_s": B / dt}
    if name == "E":
        user = ["uid", "utag1", "utag2", "utag3", "utag4"]
        item = ["i_goods_id", "i_shop_id", "i_cate_id"]
        ser = ["visited_goods_ids", "visited_shop_ids", "visited_cate_ids"]
        V, B, E, T = 50_000_000, 4096, 32, 100
        layer = layers.DINLayer(user_and_context_categorical_features=user, item_categorical_features=item,
                                behavior_series_features=ser, feature_dims=1000, embedding_dims=E)
        layer = layer.cuda()
        layer.embed.embeddings = torch.nn.Parameter(torch.empty((V, E), device="cuda"))
        big_table_(layer.embed.embeddings)
        layer.feature_dims = V
        batch = data.to_device(data.SyntheticGenerator(user + item, V, series=ser, seq_len=T, seed=0).batch(B))
        dt = timed(fwd_bwd(layer, batch, user + item + ser), 2, 10)
        return {"config": "E DIN T=100, 50M x 32d", "B": B, "V": V, "ms_per_step": dt * 1e3, "examples_per_s": B / dt,
                "attention_flops_factorised_fwd": 2.0 * B * 96 * 96 * 36 + 2.0 * B * T * 96 * 36}
    if name == "P":
        names = ["C%d" % i for i in range(26)]
        V, B = 10_000_000, 8192
        layer = layers.PNNLayer(feature_names=names, feature_dims=1000, embedding_dims=16).cuda()
        layer.embed.embeddings = torch.nn.Parameter(torch.empty((V, 16), device="cuda"))
        big_table_(layer.embed.embeddings)
        batch = data.to_device(data.SyntheticGenerator(names, V, seed=0).batch(B))
        dt = timed(fwd_bwd(layer, batch, names), 5, 50)
        return {"config": "P PNN inner product, 26 fields, 10M x 16d", "B": B, "V": V, "ms_per_step": dt * 1e3,
                "examples_per_s": B / dt}
    if name == "N":
        cat = ["c%d" % i for i in range(10)]
        cont = ["x0", "x1", "x2"]
        V, B = 10_000_000, 16384
        layer = layers.NeuralFactorizationMachineLayer(categorical_features=cat, continuous_features=cont,
                                                       feature_dims=1000, embedding_dims=16).cuda()
        layer.embed.embeddings = torch.nn.Parameter(torch.empty((V, 16), device="cuda"))
        big_table_(layer.embed.embeddings)
        batch = data.to_device(data.SyntheticGenerator(cat, V, continuous=cont, seed=0).batch(B))
        dt = timed(fwd_bwd(layer, batch, cat + cont), 5, 50)
        return {"config": "N NFM bi-interaction + BatchNormalization, 10 cat + 3 cont, 10M x 16d", "B": B, "V": V,
                "ms_per_step": dt * 1e3, "examples_per_s": B / dt}
    if name == "FF":
        names = ["C%d" % i for i in range(26)]
        V, B, E = 10_000_000, 8192, 16
        layer = layers.FFMLayer(feature_names=names, feature_dims=1000, embedding_dims=E).cuda()
        layer.fa_interaction_layer.v = torch.nn.Parameter(torch.empty((V, 26, E), device="cuda"))
        layer.w = torch.nn.Parameter(torch.empty((V, 1), device="cuda"))
        big_table_(layer.fa_interaction_layer.v)
        big_table_(layer.w)
        batch = data.to_device(data.SyntheticGenerator(names, V, seed=0).batch(B))
        dt = timed(fwd_bwd(layer, batch, names), 3, 20)
        return {"config": "FF FFM, 26 fields, 10M rows x (26 x 16d)", "B": B, "V": V, "ms_per_step": dt * 1e3,
                "examples_per_s": B / dt, "row_bytes_gathered_fwd": B * 26 * 25 * E * 4}
    if name == "G":
        item = ["i_goods_id", "i_shop_id", "i_cate_id"]
        ser = ["visited_goods_ids", "visited_shop_ids", "visited_cate_ids"]
        V, B, E, T = 50_000_000, 4096, 32, 100
        layer = layers.GSULayer(item_categorical_features=item, behavior_series_features=ser, feature_dims=1000,
                                embedding_dims=E).cuda()
        layer.embed.embeddings = torch.nn.Parameter(torch.empty((V, E), device="cuda"))
        big_table_(layer.embed.embeddings)
        layer.feature_dims = V
        batch = data.to_device(data.SyntheticGenerator(item, V, series=ser, seq_len=T, seed=0).batch(B))
        dt = timed(fwd_bwd(layer, batch, item + ser), 3, 20)
        return {"config": "G SIM GSU inner-product attention T=100, 50M x 32d", "B": B, "V": V, "ms_per_step": dt * 1e3,
                "examples_per_s": B / dt}
    if name == "R":
        from explicit_tf2_recommendation_amd import ops
        n, d, nq, k = 10_000_000, 8, 1024, 20
        g = torch.Generator(device="cuda").manual_seed(0)
        items = ops.l2_normalize_rows(torch.randn((n, d), device="cuda", generator=g))
        q = torch.randn((nq, d), device="cuda", generator=g) * 0.5
        dt = timed(lambda: ops.topk_l2(q, items, k), 2, 10)
        return {"config": "R retrieval top-%d, %d items x %dd, %d queries" % (k, n, d, nq), "B": nq, "V": n,
                "ms_per_step": dt * 1e3, "examples_per_s": nq / dt, "pairs_per_s": nq * n / dt,
                "valu_tflops (3*d flop per pair)": 3.0 * d * nq * n / dt / 1e12,
                "hbm_GBps (items read once per 256 queries)": n * d * 4 * ((nq + 255) // 256) / dt / 1e9}
    raise SystemExit("unknown config %r" % name)


if __name__ == "__main__":
    argv = [a for a in sys.argv[1:] if a != "--graphed"]
    GRAPHED = "--graphed" in sys.argv[1:]
    for n in (argv or ["A", "B", "C", "C26", "D", "E", "R", "P", "N", "FF", "G"]):
        r = run(n)
        if GRAPHED and n in ("B", "C", "C26", "D", "E", "DS", "ES", "P", "N", "FF", "G"):
            r["config"] += " [GraphedTrainStep]"
        r["n_gpus"] = 1
        print(json.dumps(r), flush=True)
