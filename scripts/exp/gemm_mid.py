#!/usr/bin/env python3
"""Mid-size GEMMs of the MLP layers (64x64-tile kernel): DIN [4096,352]x[352,200], [4096,200]x[200,80]; DSSM towers."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from explicit_tf2_recommendation_amd import ops


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for M, K, N in ((4096, 352, 200), (4096, 200, 80), (8192, 192, 64), (8192, 128, 64), (4096, 200, 352), (16384, 323, 128)):
    A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda"); b = torch.randn(N, device="cuda")
    G = torch.randn(M, N, device="cuda")
    fl = 2.0 * M * K * N
    u1 = t(lambda: ops.gemm(A, B, epi=ops.EPI_BIAS_RELU, bias=b))
    u2 = t(lambda: ops.gemm(G, B, transB=True))
    u3 = t(lambda: ops.gemm(A, G, transA=True, split_k=ops.split_k_for(M, K, N, True, False)))
    u4 = t(lambda: torch.addmm(b, A, B))
    print("M=%5d K=%4d N=%4d  fwd %6.1f us (%5.1f TF)  dX %6.1f us  dW %6.1f us   torch fwd %6.1f us" % (M, K, N, u1, fl / u1 / 1e6, u2, u3, u4))
