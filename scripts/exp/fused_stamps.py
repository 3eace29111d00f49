#!/usr/bin/env python3
"""Diagnostic (never part of the product): where the waves of deepfm_fwd_bwd_kernel spend their time.

Builds csrc/deepfm_fused.hip with -DREC_FUSED_STAMPS into its own library (gpurun_out/libfused_stamps.so), launches the
fused kernel on resident batches and reads the s_memrealtime stamps (10 ns ticks) every wave left at the phase seams.
Prints, per seam, the median / p90 over waves of the time since the EARLIEST stamp of the launch.
    python scripts/exp/fused_stamps.py
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CS = os.path.join(ROOT, "explicit-tf2-recommendation_amd", "csrc")
OUT = os.path.join(ROOT, "gpurun_out", "libfused_stamps.so")
os.makedirs(os.path.dirname(OUT), exist_ok=True)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-DREC_FUSED_STAMPS",
                       "-I" + os.path.join(ROOT, "include"), os.path.join(CS, "deepfm_fused.hip"), "-o", OUT])
from explicit_tf2_recommendation_amd import layers, data, engine  # noqa: E402

dbg = C.CDLL(OUT)
V, F, E, B = 10_000_000, 26, 16, 8192
names = ["C%d" % (i + 1) for i in range(F)]
layers.set_init_seed(1234)
L = layers.DeepFMRankingLayer(feature_names=names, feature_dims=V, embedding_dims=E, mlp_dims=[32, 8]).cuda()
gen = data.SyntheticGenerator(names, V, dist="uniform", seed=0)
batches = [data.to_device(gen.batch(B)) for _ in range(16)]
fs = engine.DeepFMFusedStep(L, B, gen.dims, gen.offsets, optimizer=None, use_graph=False)
vp = lambda t: C.c_void_p(t.data_ptr())
emb = L.embed.embeddings
fn = dbg.rec_deepfm_fused_main_f32
fn.restype = C.c_int
nwg = (B + 31) // 32
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
res = []
for it in range(12):
    b = batches[it % 16]
    arr = (C.c_void_p * F)(*[b[k].data_ptr() for k in names])
    rc = fn(vp(emb), C.c_int64(emb.stride(0)), C.c_int64(V), arr, C.c_int(F), C.c_int64(B), vp(L.bias),
            vp(L.MLP_layer1.kernel_0), vp(L.MLP_layer1.bias_0), vp(L.MLP_layer1.kernel_1), vp(L.MLP_layer1.bias_1),
            vp(L.MLP_layer2.kernel_0), vp(L.MLP_layer2.bias_0), vp(b["label"]), vp(fs.gz), vp(fs.vals), None, vp(fs.oob),
            vp(fs.ws), st)
    assert rc == 0, rc
    torch.cuda.synchronize()
    host = np.zeros(nwg * 8 * 12, dtype=np.uint64)
    assert dbg.rec_debug_fused_stamps(host.ctypes.data_as(C.POINTER(C.c_ulonglong)), nwg) == 0
    if it >= 4:
        res.append(host.reshape(nwg, 8, 12).astype(np.int64))
names_ = ["start", "ids in, row loads issued", "rows consumed (phase A end)", "barrier 1 passed", "(unused)",
          "barrier 2 passed (phase B end)", "fields done (phase C)", "end"]
acc = np.stack(res)                               # [it, wg, wave, stamp]
rel = (acc - acc[:, :, :, 0].min(axis=(1, 2))[:, None, None, None]) * 0.01     # us since the launch's first wave start
for k, n in enumerate(names_):
    x = rel[..., k].reshape(-1)
    print("%-30s median %6.2f us   p10 %6.2f   p90 %6.2f   max %6.2f" % (n, np.median(x), np.percentile(x, 10),
                                                                        np.percentile(x, 90), x.max()))
print("start by blockIdx %% 8 (blocks that share an XCD), median us: " +
      " ".join("%.2f" % np.median(rel[:, x::8, :, 0]) for x in range(8)))
print("start by blockIdx // 32, median us: " + " ".join("%.2f" % np.median(rel[:, 32 * x:32 * x + 32, :, 0]) for x in range(nwg // 32)))
x = (acc[..., 8] - acc[..., 0]).reshape(-1) * 0.01
print("start -> ids back (drained):   median %6.2f us   p10 %6.2f   p90 %6.2f   max %6.2f" % (np.median(x), np.percentile(x, 10), np.percentile(x, 90), x.max()))
x = (acc[..., 1] - acc[..., 8]).reshape(-1) * 0.01
print("ids back -> row loads + K0 fragments issued: median %6.2f us   p10 %6.2f   p90 %6.2f" % (np.median(x), np.percentile(x, 10), np.percentile(x, 90)))
seams = [0, 1, 2, 3, 5, 6, 7]                # stamp 4 (a second barrier inside phase B) no longer exists
for k0, k1 in zip(seams[:-1], seams[1:]):
    x = ((acc[..., k1] - acc[..., k0]) * 0.01).reshape(-1)
    print("segment %d -> %d: median %6.2f us   p10 %6.2f   p90 %6.2f   max %6.2f" % (k0, k1, np.median(x),
          np.percentile(x, 10), np.percentile(x, 90), x.max()))
c = (acc[..., 6] - acc[..., 5]) * 0.01
print("phase C (5 -> 6) waves with 4 fields: median %.2f us; with 3 fields: %.2f us" % (np.median(c[:, :, :2]), np.median(c[:, :, 2:])))
