#!/usr/bin/env python3
"""H^T.X (weight gradient of the CrossNet matrix layer): time against the number of K slices."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from explicit_tf2_recommendation_amd import ops


def t(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


M = 16384
for D in (835, 323):
    Hm = torch.randn(M, D, device="cuda")
    X = torch.randn(M, D, device="cuda")
    fl = 2.0 * M * D * D
    for split in ((5, 10, 15, 14, 20, 21) if D == 835 else (24, 28, 42, 56, 57, 84, 85)):
        us = t(lambda: ops.gemm(Hm, X, transA=True, split_k=split))
        print("D=%d split %2d: %7.1f us  %5.1f TF" % (D, split, us, fl / us / 1e6), flush=True)
