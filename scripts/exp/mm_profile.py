#!/usr/bin/env python3
"""Diagnostic: where a ModelManager.train_step epoch of the compiled DeepFM loop spends its time (host vs device)."""
import os, sys, time, cProfile, pstats
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from explicit_tf2_recommendation_amd import data, layers  # noqa: E402
from explicit_tf2_recommendation_amd.model_manager import ModelManager  # noqa: E402
layers.Layer.check_ids = False
names = ["C%d" % i for i in range(26)]
V, B = 1_000_000, 8192
gen = data.SyntheticGenerator(names, V, seed=0)
ds = [data.to_device(gen.batch(B)) for _ in range(64)]
mm = ModelManager(feature_names=names, data_info=data.data_info(V, 26), embedding_dims=16, lr=1e-3, batch=B,
                  layer="deepfm_ranking")
for _ in range(4):
    mm.train_step(ds)
torch.cuda.synchronize()
t0 = time.perf_counter(); mm.train_step(ds); t_host = time.perf_counter() - t0
torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print("epoch of 64 steps: host returned after %.2f ms, device done after %.2f ms; graphs %d seen %d" %
      (t_host * 1e3, t_all * 1e3, len(mm._eng[1]._graphs), len(mm._eng[1]._seen)))
pr = cProfile.Profile(); pr.enable(); mm.train_step(ds); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
