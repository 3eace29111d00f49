#!/usr/bin/env python3
"""BASELINE config D as the north star states it: DSSM two-tower with the 100M x 64d item table ROW-SHARDED over the
ranks (sharded.ShardedEmbedding: HIP bucketize -> RCCL all-to-all of ids -> owner gather -> all-to-all of rows -> inverse
permutation; backward all-to-all of row gradients + owner de-duplication), dense parameters data-parallel (one flat
all-reduce).  One process per GPU:

    python scripts/bench_sharded_dssm.py                                   # world size 1 (RCCL self-copies)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/bench_sharded_dssm.py

Prints one JSON line on rank 0: whole-job examples/s (weak scaling: B examples per rank), ms/step.
"""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from explicit_tf2_recommendation_amd import layers, data, sharded, functional as Fn  # noqa: E402


class ShardedLookup(torch.nn.Module):
    """Drop-in for layers.Embedding inside a tower: the table lives row-sharded over the process group."""

    def __init__(self, emb):
        super().__init__()
        self.emb = emb

    def forward(self, X, oob=None):
        return self.emb(X)


def main():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    V, E, B = int(os.environ.get("V", 100_000_000)), 64, 8192
    steps, warm = int(os.environ.get("STEPS", 50)), 5
    un, inn = ["user_tag1", "user_tag2"], ["item_tag1", "item_tag2", "item_tag3"]
    layers.set_init_seed(7)
    layers.Layer.check_ids = False
    layer = layers.DSSMTwoTowerRetrievalLayer(u_feature_names=un, i_feature_names=inn, u_feature_dims=1000,
                                              i_feature_dims=1000, u_embedding_dims=E, i_embedding_dims=E).cuda()
    item_tab = sharded.ShardedEmbedding(V, E).cuda()                    # rows [rank*ceil(V/P), ...) on this rank
    user_tab = sharded.ShardedEmbedding(V // 10, E).cuda()
    layer.i_tower.embed = ShardedLookup(item_tab)
    layer.u_tower.embed = ShardedLookup(user_tab)
    dense = [p for n, p in layer.named_parameters() if "embeddings_shard" not in n]
    gi = data.SyntheticGenerator(inn, V, seed=rank).batch(B)
    gu = data.SyntheticGenerator(un, V // 10, seed=100 + rank).batch(B)
    batch = data.to_device({**{k: gu[k] for k in un}, **{k: gi[k] for k in inn}, "label": gi["label"]})
    ins = {k: batch[k] for k in un + inn}

    def step():
        for p in layer.parameters():
            p.grad = None
        out = layer(ins)["output"]
        Fn.KerasBCE.apply(out, batch["label"]).backward()
        sharded.allreduce_dense_grads(dense)                              # C4

    for _ in range(warm):
        step()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dist.barrier()
    torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - t0], device="cuda", dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    if rank == 0:
        dt = float(el.item()) / steps
        sys.stdout.flush()
        print(json.dumps({"config": "D DSSM two-tower, item table %d x %dd row-sharded over %d rank(s)" % (V, E, world),
                          "n_gpus": world, "batch_per_gpu": B, "ms_per_step": dt * 1e3,
                          "examples_per_s": world * B / dt, "scaling": "weak"}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
