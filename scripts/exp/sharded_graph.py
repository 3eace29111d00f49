#!/usr/bin/env python3
"""EXPERIMENT (world size 1, RCCL): the fixed-capacity sharded DeepFM step eager vs captured in one hipGraph per cycle
of 4 batches.  Usage: sharded_graph.py [eager|graph] [exit-mode]"""
import os
import sys
import time
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from explicit_tf2_recommendation_amd import engine, data, layers  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29612")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
B, F, V = 8192, 26, 10_000_000
names = ["f%d" % i for i in range(F)]
layers.set_init_seed(1)
layer = layers.DeepFMRankingLayer(feature_names=names, feature_dims=V, embedding_dims=16).cuda()
gen = data.SyntheticGenerator(names, V, seed=0)
batches = [data.to_device(gen.batch(B)) for _ in range(16)]
step = engine.ShardedDeepFMStep(layer, B, gen.dims, gen.offsets)
print("capacity per owner", step.cap, flush=True)


def eager(n):
    for i in range(n):
        step(batches[i % 16], next_inputs=batches[(i + 1) % 16])


def graphed(n):
    for i in range(0, n, 4):
        b0 = i % 16
        step.many(batches[b0:b0 + 4])


run = eager if mode == "eager" else graphed
run(32)
torch.cuda.synchronize()
l_mode = step.loss.item()
step._next = None
step(batches[15])
print("loss", l_mode, "eager last batch", step.loss.item(), flush=True)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(400)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print("%s: %.4f ms/step (host issue %.4f)" % (mode, t / 400 * 1e3, t_issue / 400 * 1e3), flush=True)
step.check_flags()
step.release_graphs()
torch.cuda.synchronize()
print("destroying", flush=True)
dist.destroy_process_group()
print("done", flush=True)
