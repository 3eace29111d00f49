#!/usr/bin/env python3
"""Diagnostic: phase stamps of colsort_onewg_kernel (-DREC_SORT_STAMPS build in its own library)."""
import ctypes as C, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CS = os.path.join(ROOT, "explicit-tf2-recommendation_amd", "csrc")
OUT = os.path.join(ROOT, "gpurun_out", "libsort_stamps.so")
os.makedirs(os.path.dirname(OUT), exist_ok=True)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-DREC_SORT_STAMPS",
                       "-I" + os.path.join(ROOT, "include"), os.path.join(CS, "deepfm_fused.hip"), "-o", OUT])
from explicit_tf2_recommendation_amd import layers, data, engine  # noqa: E402
dbg = C.CDLL(OUT)
B, F, V = 8192, 26, 10_000_000
names = ["C%d" % i for i in range(F)]
layers.set_init_seed(1)
L = layers.DeepFMRankingLayer(feature_names=names, feature_dims=V, embedding_dims=16).cuda()
gen = data.SyntheticGenerator(names, V, dist="uniform", seed=0)
bs = [data.to_device(gen.batch(B)) for _ in range(8)]
fs = engine.DeepFMFusedStep(L, B, gen.dims, gen.offsets, use_graph=False)
fn = dbg.rec_colsort_plan_dest_i64
fn.restype = C.c_int
vp = lambda t: C.c_void_p(t.data_ptr())
res = []
for it in range(10):
    k = 4
    cl = [fs._cols(b) for b in bs[(it % 2) * 4:(it % 2) * 4 + 4]]
    arr = (C.c_void_p * (k * F))(*[c.data_ptr() for cols in cl for c in cols])
    pl = fs.plans[0]
    for rep in range(3):     # back to back: the last one is measured warm
        rc = fn(arr, C.c_int(k * F), C.c_int64(B), C.c_int64(V), vp(fs.col_lo_rep), C.c_int64(fs.max_key), vp(pl["perm"]),
                vp(pl["col_uid"]), vp(pl["col_seg"]), vp(pl["col_nu"]), vp(pl["dloc"]), vp(fs.bad_ids), vp(fs.sort_ws),
                C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
    torch.cuda.synchronize()
    host = np.zeros(256 * 16, dtype=np.uint64)
    assert dbg.rec_debug_sort_stamps(host.ctypes.data_as(C.POINTER(C.c_ulonglong))) == 0
    if it >= 2:
        res.append(host.reshape(256, 16)[:k * F].astype(np.int64))
acc = np.stack(res)
rel = (acc - acc[:, :, 0].min(axis=1)[:, None, None]) * 0.01
for k, n in ((0, "start"), (1, "ids loaded, words in LDS"), (2, "pass 1 done"), (3, "pass 2 done"), (4, "pass 3 done"),
             (8, "heads + stores issued"), (9, "stores drained"), (10, "thread 1023 end"), (11, "thread 512 end")):
    x = rel[:, :, k].reshape(-1)
    print("%-28s median %6.2f  p10 %6.2f  p90 %6.2f  max %6.2f" % (n, np.median(x), np.percentile(x, 10), np.percentile(x, 90), x.max()))
