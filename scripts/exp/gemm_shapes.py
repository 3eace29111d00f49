#!/usr/bin/env python3
"""Time the fp32 GEMM on the mid-size shapes of the DIN / GSU / PNN steps (one MI355X)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from explicit_tf2_recommendation_amd import ops  # noqa: E402

SHAPES = [  # M, N, K, transA, transB
    (4096, 96, 3492, False, True), (4096, 3492, 96, False, False), (96, 3492, 4096, True, False),
    (4096, 200, 352, False, False), (4096, 352, 200, False, True), (4096, 80, 200, False, False),
    (4096, 200, 80, False, True), (4096, 200, 192, False, False), (8192, 741, 32, False, True),
    (8192, 64, 192, False, False),
    (16384, 323, 323, False, True), (16384, 323, 323, False, False), (323, 323, 16384, True, False),
    (16384, 835, 835, False, True), (16384, 835, 835, False, False), (835, 835, 16384, True, False),
]
if len(sys.argv) > 1 and sys.argv[1] == "crossnet":
    SHAPES = SHAPES[-6:]


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for M, N, K, tA, tB in SHAPES:
    A = torch.randn((K, M) if tA else (M, K), device="cuda")
    B = torch.randn((N, K) if tB else (K, N), device="cuda")
    row = []
    for sk in (None, 1):
        row.append(t(lambda: ops.gemm(A, B, tA, tB, split_k=sk)))
    ref = t(lambda: (A.t() if tA else A) @ (B.t() if tB else B))
    print("M=%5d N=%5d K=%5d tA=%d tB=%d  auto %7.1f us  split1 %7.1f us  torch %7.1f us  (%.1f TF auto)"
          % (M, N, K, tA, tB, row[0], row[1], ref, 2.0 * M * N * K / row[0] / 1e6))
