#!/usr/bin/env python3
"""B sweep of the two gather-bound kernels of the headline path ("sweep B to 64 k to show the
asymptote"): rec_emb_fm_fwd_f32 (gather + FM forward) and the fused forward+backward kernel (plan-after form, which has
no batch limit), B = 8k .. 128k, fresh ids every launch, V = 10M x 16d fused rows.
    python scripts/exp/b_sweep.py > gpurun_out/b_sweep.json
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from explicit_tf2_recommendation_amd import layers, engine, ops  # noqa: E402
from explicit_tf2_recommendation_amd._lib import lib, check  # noqa: E402

V, F, E = 10_000_000, 26, 16
names = ["C%d" % (i + 1) for i in range(F)]
layers.set_init_seed(1234)
L = layers.DeepFMRankingLayer(feature_names=names, feature_dims=V, embedding_dims=E, mlp_dims=[32, 8]).cuda()
emb, w, bias = L.embed.embeddings, L.w.embeddings, L.bias
vp = lambda t: C.c_void_p(t.data_ptr())
dims = [V // F] * F
dims[-1] += V - sum(dims)
offs = np.concatenate([[0], np.cumsum(dims[:-1])])


def timed(launch, reps):
    launch(reps)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode=engine.CAPTURE_MODE):
        launch(reps)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


out = []
rng = np.random.Generator(np.random.PCG64(5))
for B in (8192, 16384, 32768, 65536, 131072):
    nset = max(4, (16 * 8192) // B)                       # >= 436 MB of distinct lines between two uses of a set
    cols = [[torch.from_numpy(rng.integers(0, dims[f], size=B) + offs[f]).cuda() for f in range(F)] for _ in range(nset)]
    Xs = [ops.index_pack(c) for c in cols]
    y = (torch.rand(B, device="cuda") < 0.25).float()
    z = torch.empty(B, dtype=torch.float32, device="cuda")
    gz = torch.empty(B, dtype=torch.float32, device="cuda")
    vals = torch.empty((B * F, 16), dtype=torch.float32, device="cuda")
    ws = torch.empty(lib.rec_deepfm_fused_workspace_bytes(B, F), dtype=torch.uint8, device="cuda")
    oob = torch.zeros(1, dtype=torch.int32, device="cuda")
    arrs = [(C.c_void_p * F)(*[c.data_ptr() for c in cs]) for cs in cols]

    def l_gather(n):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for i in range(n):
            check(lib.rec_emb_fm_fwd_f32(vp(emb), emb.stride(0), vp(w), w.stride(0), vp(bias), V, E, vp(Xs[i % nset]), B,
                                         F, vp(z), None, None, None, None, st), "fm_fwd")

    def l_fused(n):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for i in range(n):
            check(lib.rec_deepfm_fused_main_f32(vp(emb), emb.stride(0), V, arrs[i % nset], F, B, vp(L.bias),
                                                vp(L.MLP_layer1.kernel_0), vp(L.MLP_layer1.bias_0),
                                                vp(L.MLP_layer1.kernel_1), vp(L.MLP_layer1.bias_1),
                                                vp(L.MLP_layer2.kernel_0), vp(L.MLP_layer2.bias_0), vp(y), vp(gz),
                                                vp(vals), None, vp(oob), vp(ws), st), "fused")

    reps = 2 * nset
    ug, uf = timed(l_gather, reps), timed(l_fused, reps)
    n = B * F
    gb = n * (8 + 4 * E + 4) + 4 * B
    fb = gb + 8 * B + n * E * 4
    out.append({"B": B, "n_lookups": n, "gather_fm_us": ug, "gather_fm_G_lookups_per_s": n / ug / 1e3,
                "gather_fm_frac_algorithmic": gb / (ug * 1e-6) / 8e12, "gather_fm_frac_lines": n * 128 / (ug * 1e-6) / 8e12,
                "fused_us": uf, "fused_frac_algorithmic": fb / (uf * 1e-6) / 8e12,
                "fused_us_per_8192": uf * 8192 / B})
    sys.stderr.write(json.dumps(out[-1]) + "\n")
    del cols, Xs, vals, ws
    torch.cuda.empty_cache()
print(json.dumps({"sweep": out}))
