#!/usr/bin/env python3
"""The three CrossNet GEMM shapes at D=835, B=16384 (X.W^T, H.W, H^T.X), 10 launches each -- the workload of pmc_gemm.sh."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from explicit_tf2_recommendation_amd import ops  # noqa: E402

B, D = 16384, 835
X = torch.randn((B, D), device="cuda")
W = torch.randn((D, D), device="cuda")
for _ in range(10):
    ops.gemm(X, W, transB=True)
    ops.gemm(X, W)
    ops.gemm(X, X, transA=True)
torch.cuda.synchronize()
