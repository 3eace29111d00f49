#!/usr/bin/env python3
"""Diagnostic: v3 against v2 of the fused kernel on small shapes (partial tiles, F = 27 / 28: the form without K0 in LDS)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from explicit_tf2_recommendation_amd import layers, data, engine  # noqa: E402
for (B, F, V) in [(70, 27, 60000), (64, 27, 60000), (96, 27, 60000), (70, 28, 60000), (70, 26, 60000), (160, 28, 60000),
                  (33, 7, 4000), (6, 27, 60000), (38, 27, 60000), (22, 27, 60000)]:
    names = ["f%d" % i for i in range(F)]
    layers.set_init_seed(11)
    L = layers.DeepFMRankingLayer(feature_names=names, feature_dims=V, embedding_dims=16).cuda()
    with torch.no_grad():
        L.embed.embeddings.mul_(2.0)
    gen = data.SyntheticGenerator(names, V, dist="uniform", seed=11)
    b = data.to_device(gen.batch(B))
    out = {}
    for ver in (2, 3):
        for direct in (True, False):
            st = engine.DeepFMFusedStep(L, B, gen.dims, gen.offsets, use_graph=False, kernel=ver, direct=direct)
            losses = [st(b).item() for _ in range(3)]
            g = st.gradients()
            out[(ver, direct)] = (losses, g["MLP_layer1.kernel_0"].clone(), g["embed.embeddings"][1].clone(), st.gz.clone())
    ref = out[(2, True)]
    msg = []
    for k, v in out.items():
        msg.append("%s loss %s dK0 %.1e rows %.1e gz %.1e" % (k, ["%.7f" % x for x in v[0]], (v[1] - ref[1]).abs().max().item(),
                                                             (v[2] - ref[2]).abs().max().item(), (v[3] - ref[3]).abs().max().item()))
    print((B, F), "\n   " + "\n   ".join(msg))
