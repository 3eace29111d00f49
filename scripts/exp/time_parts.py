#!/usr/bin/env python3
"""Diagnostic: the launches of a fused DeepFM step timed one kind at a time (no other stream active), each as a graph of
back-to-back launches over the 16 resident batches:  fused kernel (direct mode), post launch, sort chain of 1 / 2 batches."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from explicit_tf2_recommendation_amd import layers, data, engine  # noqa: E402

V, F, E, B = 10_000_000, 26, 16, 8192
dist = sys.argv[1] if len(sys.argv) > 1 else "uniform"
direct = (sys.argv[2] != "classic") if len(sys.argv) > 2 else True
names = ["C%d" % (i + 1) for i in range(F)]
layers.set_init_seed(1234)
L = layers.DeepFMRankingLayer(feature_names=names, feature_dims=V, embedding_dims=E, mlp_dims=[32, 8]).cuda()
gen = data.SyntheticGenerator(names, V, dist=dist, seed=0)
NB = 16
batches = [data.to_device(gen.batch(B)) for _ in range(NB)]
fs = engine.DeepFMFusedStep(L, B, gen.dims, gen.offsets, optimizer=None, use_graph=False, direct=direct)
cols = [fs._cols(b) for b in batches]
cur = torch.cuda.current_stream()
for i in range(NB):
    fs._sort(cols[i], i, cur)
torch.cuda.synchronize()


def timed(launch, reps):
    launch(4)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        launch(reps)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay()
    torch.cuda.synchronize()
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def l_main(n):
    for i in range(n):
        fs._launch_main(cols[i % NB], batches[i % NB]["label"], st(), i % NB)


def l_post(n):
    for i in range(n):
        fs._launch_post(i % NB, st())


def l_both(n):
    for i in range(n):
        fs._launch_main(cols[i % NB], batches[i % NB]["label"], st(), i % NB)
        fs._launch_post(i % NB, st())


def l_sort1(n):
    for i in range(n):
        fs._sort(cols[i % NB], i % NB, torch.cuda.current_stream())


def l_sort4(n):
    for i in range(n):
        j = (4 * i) % NB
        fs._sort_group([cols[j], cols[j + 1], cols[j + 2], cols[j + 3]], j, torch.cuda.current_stream())


def l_sort2(n):
    for i in range(n):
        j = (2 * i) % NB
        fs._sort_group([cols[j], cols[j + 1]], j, torch.cuda.current_stream())


print("id distribution: %s, %s mode" % (dist, "direct" if direct else "classic"))
print("fused kernel         %7.2f us" % timed(l_main, 48))
print("post launch          %7.2f us" % timed(l_post, 48))
print("fused + post back to back    %7.2f us per step" % timed(l_both, 48))
print("sort chain, one batch        %7.2f us" % timed(l_sort1, 32))
print("sort chain, two batches      %7.2f us (per batch %.2f)" % ((lambda t: (t, t / 2))(timed(l_sort2, 16))))
if fs.GROUP >= 4:
    print("sort chain, four batches     %7.2f us (per batch %.2f)" % ((lambda t: (t, t / 4))(timed(l_sort4, 8))))
