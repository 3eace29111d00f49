#!/usr/bin/env python3
"""Diagnostic: duration of the per-column LDS sort (rec_colsort_plan_dest_i64) alone, by number of columns per launch and
by key width (radix passes)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from explicit_tf2_recommendation_amd import layers, data, engine  # noqa: E402
B, F = 8192, 26
for V in (26 * 100, 26 * 16000, 10_000_000):
    names = ["C%d" % i for i in range(F)]
    layers.set_init_seed(1)
    L = layers.DeepFMRankingLayer(feature_names=names, feature_dims=V, embedding_dims=16).cuda()
    gen = data.SyntheticGenerator(names, V, dist="uniform", seed=0)
    bs = [data.to_device(gen.batch(B)) for _ in range(8)]
    fs = engine.DeepFMFusedStep(L, B, gen.dims, gen.offsets, use_graph=False)
    cols = [fs._cols(b) for b in bs]
    for k in (1, 2, 4):
        def launch(n):
            st = torch.cuda.current_stream()
            for i in range(n):
                fs._sort_group(cols[(i * k) % 8:(i * k) % 8 + k] if (i * k) % 8 + k <= 8 else cols[:k], 0, st)
        launch(4); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode=engine.CAPTURE_MODE):
            launch(16)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay(); torch.cuda.synchronize()
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        print("V=%9d key_bits=%2d  %d batch(es) = %3d columns per launch: %6.2f us per launch" %
              (V, int(np.ceil(np.log2(max(gen.dims)))), k, k * F, e0.elapsed_time(e1) * 1e3 / 16))
    del L, fs
    torch.cuda.empty_cache()
