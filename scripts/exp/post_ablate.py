#!/usr/bin/env python3
"""Diagnostic: the post launch (deepfm_post_direct_kernel) alone over Zipf / uniform batches, with parts of its run
handling compiled out (-DABL_NOSHORT / -DABL_NOLONG / -DABL_NOHUGE; results are then wrong -- timing only).
    DIST=zipf python scripts/exp/post_ablate.py"""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from explicit_tf2_recommendation_amd import layers, data, engine  # noqa: E402
from explicit_tf2_recommendation_amd._lib import lib  # noqa: E402

V, F, B = 10_000_000, 26, 8192
names = ["C%d" % (i + 1) for i in range(F)]
layers.set_init_seed(1234)
L = layers.DeepFMRankingLayer(feature_names=names, feature_dims=V, embedding_dims=16, mlp_dims=[32, 8]).cuda()
gen = data.SyntheticGenerator(names, V, dist=os.environ.get("DIST", "zipf"), seed=0)
NB = 16
batches = [data.to_device(gen.batch(B)) for _ in range(NB)]
fs = engine.DeepFMFusedStep(L, B, gen.dims, gen.offsets, optimizer=None, use_graph=False)
colss = [fs._cols(b) for b in batches]
for i in range(NB):
    fs._sort(colss[i], i, torch.cuda.current_stream())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
for i in range(NB):
    fs._launch_main(colss[i], batches[i]["label"], st(), i)
torch.cuda.synchronize()
CS = os.path.join(ROOT, "explicit-tf2-recommendation_amd", "csrc")
NAME = "rec_deepfm_fused_post_direct_f32"
VARIANTS = ([], ["-DABL_NOSHORT"], ["-DABL_NOLONG"], ["-DABL_NOHUGE"], ["-DABL_NOSHORT", "-DABL_NOLONG", "-DABL_NOHUGE"])
if os.environ.get("VARIANTS"):
    VARIANTS = [v.split() for v in os.environ["VARIANTS"].split(";")]
for flags in VARIANTS:
    out = os.path.join(ROOT, "gpurun_out", "libpost_abl.so")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared"] + flags +
                          ["-I" + os.path.join(ROOT, "include"), os.path.join(CS, "deepfm_fused.hip"), "-o", out + ".%d" % len(flags)])
    os.replace(out + ".%d" % len(flags), out + "".join(flags))
    dbg = C.CDLL(out + "".join(flags))
    getattr(dbg, NAME).argtypes = getattr(lib, NAME).argtypes
    getattr(dbg, NAME).restype = C.c_int

    class Proxy:
        def __getattr__(self, n):
            return getattr(dbg, n) if n == NAME else getattr(lib, n)
    engine.lib = Proxy()

    def run(n):
        for i in range(n):
            fs._launch_post(i % NB, st())
    run(NB)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode=engine.CAPTURE_MODE):
        run(3 * NB)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / (3 * NB))
    print("%-50s %.2f us per post launch" % (" ".join(flags) or "(shipped)", min(ts)))
