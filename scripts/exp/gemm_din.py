#!/usr/bin/env python3
"""The three GEMMs around the DIN attention at config E (B=4096, D=96, H=36: N = D*H + H = 3492)."""
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


B, D, N = 4096, 96, 3492
q = torch.randn(B, D, device="cuda")
Wcat = torch.randn(D, N, device="cuda") * 0.05
bext = torch.randn(N, device="cuda")
gM = torch.randn(B, N, device="cuda")
out = torch.empty(B, N, device="cuda")
fl = 2.0 * B * D * N
for name, fn in (("Mext = q.Wcat + b", lambda: ops.gemm(q, Wcat, epi=ops.EPI_BIAS, bias=bext, out=out)),
                 ("gq = gMext.Wcat^T", lambda: ops.gemm(gM, Wcat, transB=True)),
                 ("gWcat = q^T.gMext", lambda: ops.gemm(q, gM, transA=True, split_k=ops.split_k_for(B, D, N, True, False))),
                 ("torch q@Wcat", lambda: torch.addmm(bext, q, Wcat)),
                 ("torch gM@Wcat^T", lambda: gM @ Wcat.t()),
                 ("torch q^T@gM", lambda: q.t() @ gM)):
    us = t(fn)
    print("%-22s %7.1f us  %5.1f TF" % (name, us, fl / us / 1e6))
