#!/usr/bin/env python3
"""EXPERIMENT: can an RCCL all-to-all with constant split sizes be captured in a hipGraph on this stack (world size 1)?"""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29611")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.arange(1 << 20, device="cuda", dtype=torch.float32)
y = torch.empty_like(x)
dist.all_to_all_single(y, x)                       # warm-up: communicator creation must not be captured
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        dist.all_to_all_single(y, x, output_split_sizes=[x.numel()], input_split_sizes=[x.numel()])
        z = y * 2
    x.add_(1)
    g.replay()
    torch.cuda.synchronize()
    print("captured all_to_all_single: OK", bool(torch.equal(z, x * 2)))
except Exception as e:                              # noqa: BLE001
    print("capture failed:", type(e).__name__, str(e)[:300])
dist.destroy_process_group()
