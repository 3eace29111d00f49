#!/usr/bin/env python3
"""Diagnostic: phase stamps of the standalone post3 kernel (csrc/deepfm_fused3.hip, -DREC_FUSED_STAMPS build: every stamp
is preceded by s_waitcnt vmcnt(0), so a phase's time includes its loads landing) on the results of a real fused launch.
    [DIST=zipf] python scripts/exp/post3_stamps.py
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from explicit_tf2_recommendation_amd import layers, data, engine  # noqa: E402

V, F, E, B = 10_000_000, 26, 16, int(os.environ.get("B", "8192"))
names = ["C%d" % (i + 1) for i in range(F)]
layers.set_init_seed(1234)
L = layers.DeepFMRankingLayer(feature_names=names, feature_dims=V, embedding_dims=E, mlp_dims=[32, 8]).cuda()
gen = data.SyntheticGenerator(names, V, dist=os.environ.get("DIST", "uniform"), seed=0)
batches = [data.to_device(gen.batch(B)) for _ in range(4)]
fs = engine.DeepFMFusedStep(L, B, gen.dims, gen.offsets, optimizer=None, use_graph=False)
assert fs._pipelined
for b in batches:
    fs(b)
torch.cuda.synchronize()
ABL = [x for x in os.environ.get("ABL", "").split() if x]
CS = os.path.join(ROOT, "explicit-tf2-recommendation_amd", "csrc")
OUT = os.path.join(ROOT, "gpurun_out", "libpost3_stamps.so")
os.makedirs(os.path.dirname(OUT), exist_ok=True)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-DREC_FUSED_STAMPS"] + ABL +
                      ["-I" + os.path.join(ROOT, "include"), os.path.join(CS, "deepfm_fused3.hip"), "-o", OUT])
dbg = C.CDLL(OUT)
fn = dbg.rec_deepfm_fused3_post_f32
fn.restype = C.c_int
# the last step of the loop above left its results in parity 0 with the plan in some buffer: redo one by hand
cols = fs._cols(batches[0])
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
fs._sort(cols, 0, torch.cuda.current_stream())
fs._row = 0
fs._launch_main(cols, batches[0]["label"], st, 0, par=0)
torch.cuda.synchronize()
args = fs._post_args(0, 0, 0)
nwg = (B + 31) // 32
acc = []
for rep in range(6):
    assert fn(C.c_int(F), C.c_int64(B), *args, st) == 0
    torch.cuda.synchronize()
    host = np.zeros(nwg * 4 * 12, dtype=np.uint64)
    assert dbg.rec_debug_post3_stamps(host.ctypes.data_as(C.POINTER(C.c_ulonglong)), nwg) == 0
    if rep >= 2:
        acc.append(host.reshape(nwg, 4, 12).astype(np.int64))
acc = np.stack(acc)
rel = (acc - acc[:, :, :, 0].min(axis=(1, 2))[:, None, None, None]) * 0.01
labels = {0: "start", 1: "issue: partials + plan words in", 2: "mid: gz(head), second members in",
          3: "mid2: first round of short runs in", 4: "reduce done (stores drained)", 5: "slots done (stores drained)"}
for k, n in labels.items():
    x = rel[:, :, :, k].reshape(-1)
    print("   %-40s median %6.2f us   p10 %6.2f   p90 %6.2f   max %6.2f" % (n, np.median(x), np.percentile(x, 10),
                                                                          np.percentile(x, 90), x.max()))
d = rel[:, :, :, 5] - rel[:, :, :, 4]
print("slots phase per wave: median %.2f  p90 %.2f  max %.2f us" % (np.median(d), np.percentile(d, 90), d.max()))
print("n_uniq", int(fs._pb[0]["n_uniq"].item()), "of", F * B)
