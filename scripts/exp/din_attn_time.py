#!/usr/bin/env python3
"""DIN attention kernels at config E (B=4096, T=100, C=3, E=32, H=36) with a 50M-row table (HBM-random rows) and with a
1000-row table (rows stay in L2): tells memory latency / bandwidth apart from everything else in the kernel."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from explicit_tf2_recommendation_amd import ops  # noqa: E402

B, T, C, E, H = 4096, 100, 3, 32, 36
D = C * E


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for V in (50_000_000, 1000):
    g = torch.Generator(device="cuda").manual_seed(0)
    embed = torch.empty((V, E), device="cuda").uniform_(-0.05, 0.05)
    series = torch.randint(1, V, (B, T, C), device="cuda", generator=g)
    lens = torch.randint(1, T + 1, (B,), device="cuda", generator=g)
    series[torch.arange(T, device="cuda")[None, :] >= lens[:, None]] = 0
    q = torch.randn((B, D), device="cuda", generator=g) * 0.1
    W1 = torch.randn((3 * D + D * D, H), device="cuda", generator=g) * 0.05
    b1 = torch.zeros(H, device="cuda")
    Wcat, Wkd, bext = ops.din_prepare(W1, b1, D, H)
    Mext = ops.gemm(q, Wcat, epi=ops.EPI_BIAS, bias=bext)
    alpha = torch.zeros(H, device="cuda"); mean = torch.zeros(H, device="cuda"); var = torch.ones(H, device="cuda")
    w2 = torch.randn(H, device="cuda", generator=g) * 0.1
    b2 = torch.zeros(1, device="cuda")
    args = (embed, series, Mext, Wkd, ops.DACT_DICE, alpha, mean, var, w2, b2, 0, 0)
    scores, pooled = ops.din_attn_fwd(*args)
    gp = torch.randn((B, D), device="cuda", generator=g)
    print("V=%9d  fwd %7.1f us   bwd %7.1f us" % (V, t(lambda: ops.din_attn_fwd(*args)),
                                                   t(lambda: ops.din_attn_bwd(*args, scores, gp))))
    del embed
