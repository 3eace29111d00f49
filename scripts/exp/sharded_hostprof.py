#!/usr/bin/env python3
"""EXPERIMENT: where the host time of the eager sharded step goes (world size 1, RCCL)."""
import cProfile
import os
import pstats
import sys
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from explicit_tf2_recommendation_amd import engine, data, layers  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29613")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
B, F, V = 8192, 26, 10_000_000
names = ["f%d" % i for i in range(F)]
layers.set_init_seed(1)
layer = layers.DeepFMRankingLayer(feature_names=names, feature_dims=V, embedding_dims=16).cuda()
gen = data.SyntheticGenerator(names, V, seed=0)
batches = [data.to_device(gen.batch(B)) for _ in range(16)]
step = engine.ShardedDeepFMStep(layer, B, gen.dims, gen.offsets)


def eager(n):
    for i in range(n):
        step(batches[i % 16], next_inputs=batches[(i + 1) % 16])


eager(64)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
eager(400)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
dist.destroy_process_group()
