#!/usr/bin/env python3
"""Diagnostic: phase stamps (-DREC_FUSED_STAMPS build of csrc/deepfm_fused3.hip) of the fused kernel that carries the post
step of the previous iteration, in the steady state of a graph of back-to-back launches over fresh batches.
    [DIST=zipf] python scripts/exp/post3_merged_stamps.py
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
from explicit_tf2_recommendation_amd._lib import lib  # noqa: E402

V, F, E, B = 10_000_000, 26, 16, int(os.environ.get("B", "8192"))
names = ["C%d" % (i + 1) for i in range(F)]
layers.set_init_seed(1234)
L = layers.DeepFMRankingLayer(feature_names=names, feature_dims=V, embedding_dims=E, mlp_dims=[32, 8]).cuda()
gen = data.SyntheticGenerator(names, V, dist=os.environ.get("DIST", "uniform"), seed=0)
NB = 16
batches = [data.to_device(gen.batch(B)) for _ in range(NB)]
fs = engine.DeepFMFusedStep(L, B, gen.dims, gen.offsets, optimizer=None, use_graph=False)
assert fs._pipelined
colss = [fs._cols(b) for b in batches]
for i in range(NB):
    fs._sort(colss[i], i, torch.cuda.current_stream())
torch.cuda.synchronize()

ABL = [x for x in os.environ.get("ABL", "").split() if x]
CS = os.path.join(ROOT, "explicit-tf2-recommendation_amd", "csrc")
OUT = os.path.join(ROOT, "gpurun_out", "libpost3m_stamps.so")
os.makedirs(os.path.dirname(OUT), exist_ok=True)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-DREC_FUSED_STAMPS"] + ABL +
                      ["-I" + os.path.join(ROOT, "include"), os.path.join(CS, "deepfm_fused3.hip"), "-o", OUT])
dbg = C.CDLL(OUT)
SWAP = ("rec_deepfm_fused3_main_direct_post_f32", "rec_deepfm_fused3_main_direct_f32", "rec_deepfm_fused3_post_f32")
for n in SWAP:
    getattr(dbg, n).argtypes = getattr(lib, n).argtypes
    getattr(dbg, n).restype = getattr(lib, n).restype


class Proxy:
    def __getattr__(self, n):
        return getattr(dbg, n) if n in SWAP else getattr(lib, n)


engine.lib = Proxy()


def st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(n):
    fs._row = 0
    fs._launch_main(colss[0], batches[0]["label"], st(), 0, par=0)
    for i in range(1, n):
        fs._launch_main(colss[i % NB], batches[i % NB]["label"], st(), i % NB, par=i & 1, prev=((i - 1) & 1, (i - 1) % NB, 0))
    # (no trailing post launch: the stamps of the LAST fused launch are what is read)


nwg = (B + 31) // 32
run(3)
torch.cuda.synchronize()
acc = []
for nl in (17, 18, 19, 20, 21, 22, 23, 24):
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, capture_error_mode=engine.CAPTURE_MODE):
        run(nl)
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record()
    torch.cuda.synchronize()
    host = np.zeros(nwg * 8 * 12, dtype=np.uint64)
    assert dbg.rec_debug_fused3_stamps(host.ctypes.data_as(C.POINTER(C.c_ulonglong)), nwg) == 0
    acc.append(host.reshape(nwg, 8, 12).astype(np.int64))
    print("stamps build: %.2f us per launch over %d back-to-back launches" % (e0.elapsed_time(e1) * 1e3 / nl, nl))
acc = np.stack(acc)
rel = (acc - acc[:, :, :, 0].min(axis=(1, 2))[:, None, None, None]) * 0.01
labels = [(0, "start"), (8, "ids arrived (B: barrier 0 passed)"), (1, "row loads issued"),
          (9, "barrier 0 passed (A)"), (2, "layer 1 done"), (3, "half sync after layer 1 passed"), (11, "head done (B)"),
          (5, "barrier after head passed"), (6, "dX (+dK0 of A) done"), (7, "end")]
for hname, sl in (("half A (waves 0-3)", slice(0, 4)), ("half B (waves 4-7)", slice(4, 8))):
    print(hname)
    for k, n in labels:
        x = rel[:, :, sl, k].reshape(-1)
        print("   %-36s median %6.2f us   p10 %6.2f   p90 %6.2f   max %6.2f" % (n, np.median(x), np.percentile(x, 10),
                                                                              np.percentile(x, 90), x.max()))
