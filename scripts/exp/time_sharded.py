#!/usr/bin/env python3
"""Host-side profile of ShardedDeepFMStep at world_size 1 (RCCL): where the Python/driver time of a step goes."""
import cProfile
import os
import pstats
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from explicit_tf2_recommendation_amd import layers, engine, data  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
V, F, B = 10_000_000, 26, 8192
names = ["C%d" % i for i in range(F)]
layers.set_init_seed(1)
layer = layers.DeepFMRankingLayer(feature_names=names, feature_dims=V, embedding_dims=16).cuda()
gen = data.SyntheticGenerator(names, V, seed=0)
batches = [data.to_device(gen.batch(B)) for _ in range(4)]
step = engine.ShardedDeepFMStep(layer, B, gen.dims, gen.offsets)
for i in range(40):
    step(batches[i % 4], next_inputs=batches[(i + 1) % 4])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200):
    step(batches[i % 4], next_inputs=batches[(i + 1) % 4])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host issue %.1f us/step, with drain %.1f us/step" % ((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
# wall-clock split of the host side (monkeypatched timers; no profiler overhead)
acc = {}


def timed(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name

    def w(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        acc[label] = acc.get(label, 0.0) + time.perf_counter() - t
        return r
    setattr(obj, name, w)


for hc in step.be.host_counts:
    timed(hc, "get", "host_counts.get (event wait)")
timed(step.comm, "all_to_all")
timed(step.comm, "exchange_counts")
for nm in ("plan", "gather", "rows_step", "local_grad", "owner_reduce", "counts_to_host", "fork", "join", "begin"):
    timed(step.be, nm)
timed(step, "_cols")
t0 = time.perf_counter()
for i in range(200):
    step(batches[i % 4], next_inputs=batches[(i + 1) % 4])
tt = time.perf_counter() - t0
torch.cuda.synchronize()
print("instrumented: %.1f us/step" % (tt / 200 * 1e6))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-34s %7.1f us/step" % (k, v / 200 * 1e6))
dist.destroy_process_group()
sys.exit(0)
pr = cProfile.Profile()
pr.enable()
for i in range(200):
    step(batches[i % 4], next_inputs=batches[(i + 1) % 4])
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
dist.destroy_process_group()
