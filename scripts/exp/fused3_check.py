#!/usr/bin/env python3
"""Diagnostic: the third form of the fused kernel (csrc/deepfm_fused3.hip) against the second (csrc/deepfm_fused.hip) on
the same batches -- outputs (gz, value rows, reduced partials), time per launch (graph-replayed, fresh batch per launch),
and the phase stamps of a -DREC_FUSED_STAMPS build.
    python scripts/exp/fused3_check.py [--no-stamps]
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
from explicit_tf2_recommendation_amd._lib import lib, check  # noqa: E402

V, F, E, B = 10_000_000, 26, 16, int(os.environ.get("B", "8192"))
names = ["C%d" % (i + 1) for i in range(F)]
layers.set_init_seed(1234)
L = layers.DeepFMRankingLayer(feature_names=names, feature_dims=V, embedding_dims=E, mlp_dims=[32, 8]).cuda()
with torch.no_grad():
    L.MLP_layer1.bias_0.uniform_(-0.1, 0.1)
    L.MLP_layer1.bias_1.uniform_(-0.1, 0.1)
gen = data.SyntheticGenerator(names, V, dist=os.environ.get("DIST", "uniform"), seed=0)
NB = 16
batches = [data.to_device(gen.batch(B)) for _ in range(NB)]
vp = lambda t: C.c_void_p(t.data_ptr())

res = {}
for ver in (() if "--stamps-only" in sys.argv else (2, 3)):
    fs = engine.DeepFMFusedStep(L, B, gen.dims, gen.offsets, optimizer=None, use_graph=False, kernel=ver)
    loss = fs(batches[0]).item()
    torch.cuda.synchronize()
    g = fs.gradients()
    res[ver] = dict(loss=loss, gz=fs.gz.clone(), K0=g["MLP_layer1.kernel_0"].clone(), K1=g["MLP_layer1.kernel_1"].clone(),
                    b0=g["MLP_layer1.bias_0"].clone(), b1=g["MLP_layer1.bias_1"].clone(),
                    K2=g["MLP_layer2.kernel_0"].clone(), bias=g["bias"].clone(), uid=fs.uniq_ids.clone(),
                    rows=fs.g_embed_rows.clone(), gw=fs.g_w_rows.clone(), nu=int(fs.n_uniq.item()))
    colss = [fs._cols(b) for b in batches]
    for i in range(NB):
        fs._sort(colss[i], i, torch.cuda.current_stream())
    torch.cuda.synchronize()

    def launch(n, fs=fs, colss=colss):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for i in range(n):
            fs._launch_main(colss[i % NB], batches[i % NB]["label"], st, i % NB)

    launch(NB)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, capture_error_mode=engine.CAPTURE_MODE):
        launch(3 * NB)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gr.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / (3 * NB))
    res[ver]["us"] = min(ts)
    print("kernel v%d: loss %.7f  %.2f us per launch (direct mode, fresh batch per launch)" % (ver, loss, min(ts)))

if res:
    a, b = res[2], res[3]
    print("n_uniq", a["nu"], b["nu"], "uniq ids equal:", bool(torch.equal(a["uid"], b["uid"])))
    for k in ("gz", "K0", "K1", "b0", "b1", "K2", "bias", "rows", "gw"):
        x, y = a[k].double(), b[k].double()
        print("%-5s max|v2| %.3e   max|v3 - v2| %.3e   rel %.2e" % (k, x.abs().max().item(), (x - y).abs().max().item(),
                                                                    (x - y).abs().max().item() / max(1e-30, x.abs().max().item())))

if "--no-stamps" in sys.argv:
    sys.exit(0)
ABL = [x for x in os.environ.get("ABL", "").split() if x]
print("stamps build, extra flags:", ABL)
CS = os.path.join(ROOT, "explicit-tf2-recommendation_amd", "csrc")
OUT = os.path.join(ROOT, "gpurun_out", "libfused3_stamps.so")
os.makedirs(os.path.dirname(OUT), exist_ok=True)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-DREC_FUSED_STAMPS"] + ABL +
                      ["-I" + os.path.join(ROOT, "include"), os.path.join(CS, "deepfm_fused3.hip"), "-o", OUT])
dbg = C.CDLL(OUT)
fs = engine.DeepFMFusedStep(L, B, gen.dims, gen.offsets, optimizer=None, use_graph=False, kernel=3)
fs._ensure_k0t()
fn = dbg.rec_deepfm_fused3_main_f32
fn.restype = C.c_int
nwg = (B + 31) // 32
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
emb = L.embed.embeddings
acc = []


def dbg_launch(n):
    for it in range(n):
        bt = batches[it % NB]
        arr = (C.c_void_p * F)(*[bt[k].data_ptr() for k in names])
        rc = fn(vp(emb), C.c_int64(emb.stride(0)), C.c_int64(V), arr, C.c_int(F), C.c_int64(B), vp(L.bias),
                vp(L.MLP_layer1.kernel_0), vp(fs._k0t), vp(L.MLP_layer1.bias_0), vp(L.MLP_layer1.kernel_1),
                vp(L.MLP_layer1.bias_1), vp(L.MLP_layer2.kernel_0), vp(L.MLP_layer2.bias_0), vp(bt["label"]), vp(fs.gz),
                vp(fs.vals), None, vp(fs.oob), vp(fs.ws), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc


# stamps of the LAST launch of a graph of back-to-back launches (the steady state the bench times: warm instruction
# caches, working clocks), replayed with a different number of launches per graph so that the last batch varies
dbg_launch(2)
torch.cuda.synchronize()
for nl in (17, 18, 19, 20, 21, 22, 23, 24):
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, capture_error_mode=engine.CAPTURE_MODE):
        dbg_launch(nl)
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
labels = {0: "start", 8: "ids arrived (B: barrier 0 passed)", 1: "row loads issued", 9: "barrier 0 passed (A)", 2: "layer 1 done",
          3: "half sync after layer 1 passed",
          5: "barrier after head passed", 6: "dX (+dK0 of A) done", 7: "end"}
for hname, sl in (("half A (waves 0-3)", slice(0, 4)), ("half B (waves 4-7)", slice(4, 8))):
    print(hname)
    for k, n in labels.items():
        x = rel[:, :, sl, k].reshape(-1)
        print("   %-30s median %6.2f us   p10 %6.2f   p90 %6.2f   max %6.2f" % (n, np.median(x), np.percentile(x, 10),
                                                                              np.percentile(x, 90), x.max()))
print("start by blockIdx % 8, median us: " + " ".join("%.2f" % np.median(rel[:, x::8, :, 0]) for x in range(8)))
