#!/usr/bin/env python3
"""EXPERIMENT: host time to launch the captured cycle graphs of DeepFMFusedStep vs the GPU time they take."""
import os
import sys
import time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from explicit_tf2_recommendation_amd import engine, data, layers  # noqa: E402

Cy = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B, F, V = 8192, 26, 10_000_000
names = ["C%d" % (i + 1) for i in range(F)]
layers.set_init_seed(1234)
layer = layers.DeepFMRankingLayer(feature_names=names, feature_dims=V, embedding_dims=16, mlp_dims=[32, 8]).cuda()
gen = data.SyntheticGenerator(names, V, seed=0)
NB = 16
batches = [data.to_device(gen.batch(B)) for _ in range(NB)]
step = engine.DeepFMFusedStep(layer, B, gen.dims, gen.offsets, optimizer=None, use_graph=True)


def run(n):
    i = 0
    while i < n:
        b0 = i % NB
        step.many(batches[b0:b0 + Cy], then=[batches[(b0 + Cy + j) % NB] for j in range(Cy)])
        i += Cy


run(64)
torch.cuda.synchronize()
import gc
gc.collect(); gc.disable()
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(400)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("cycle %d: host issue %.2f us/step, total %.2f us/step" % (Cy, (t1 - t0) / 400 * 1e6, (t2 - t0) / 400 * 1e6), flush=True)
# GPU-only time of one graph: launch one, sync, repeat
g = list(step._graphs.values())[0] if hasattr(step, "_graphs") else None
print("graphs cached:", len(getattr(step, "_graphs", {})))


def region(n, idle=0.0, prespin=0):
    if idle:
        time.sleep(idle)
    if prespin:
        run(prespin)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for idle, pre in ((0, 0), (0, 0), (0.05, 0), (0.05, 0), (0.05, 16), (0.05, 16), (0.05, 64), (0.05, 64), (0.0, 16)):
    print("K=20 idle %.2f s prespin %2d: %.2f us/step" % (idle, pre, region(20, idle, pre)), flush=True)
for idle, pre in ((0, 0), (0.05, 0), (0.05, 64)):
    print("K=200 idle %.2f s prespin %2d: %.2f us/step" % (idle, pre, region(200, idle, pre)), flush=True)

import cProfile, pstats
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
run(400)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
