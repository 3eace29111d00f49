"""Standalone timing of the four launches of the fused DeepFM step (no concurrency between them): each piece is
captured `reps` times into a hipGraph and replayed between HIP events.  Run on the GPU box from the repo root."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
from explicit_tf2_recommendation_amd import layers, engine, data  # noqa: E402
from explicit_tf2_recommendation_amd._lib import lib, check  # noqa: E402

V, F, E, B = 10_000_000, 26, 16, 8192
names = ["C%d" % (i + 1) for i in range(F)]
layer = layers.DeepFMRankingLayer(feature_names=names, feature_dims=V, embedding_dims=E, mlp_dims=[32, 8]).cuda()
gen = data.SyntheticGenerator(names, V, dist=sys.argv[1] if len(sys.argv) > 1 else "uniform", seed=0)
batch = data.to_device(gen.batch(B))
st = engine.DeepFMFusedStep(layer, B, gen.dims, gen.offsets, use_graph=False)
st(batch)
torch.cuda.synchronize()
st.check_flags()
p = engine._p
cols = [batch[n] for n in names]
arr = (C.c_void_p * F)(*[c.data_ptr() for c in cols])
L = layer
g = st.g
emb = L.embed.embeddings


def sort_chain(s):
    check(lib.rec_colsort_plan_i64(arr, F, B, V, p(st.col_lo), st.max_key, p(st.perm), p(st.col_uid), p(st.col_seg),
                                   p(st.col_nu), p(st.bad_ids), p(st.sort_ws), s), "sort")


def fwd_bwd(s):
    check(lib.rec_deepfm_fused_fwd_bwd_f32(
        p(emb), emb.stride(0), V, arr, F, B, p(L.bias), p(L.MLP_layer1.kernel_0), p(L.MLP_layer1.bias_0),
        p(L.MLP_layer1.kernel_1), p(L.MLP_layer1.bias_1), p(L.MLP_layer2.kernel_0), p(L.MLP_layer2.bias_0),
        p(batch["label"]), p(st.gz), p(st.vals), None, p(g["MLP_layer1.kernel_0"]), p(g["MLP_layer1.bias_0"]),
        p(g["MLP_layer1.kernel_1"]), p(g["MLP_layer1.bias_1"]), p(g["MLP_layer2.kernel_0"]),
        p(g["MLP_layer2.bias_0"]), p(g["bias"]), p(st.loss), p(st.oob), p(st.ws), s), "fwd_bwd")


def colseg(s):
    check(lib.rec_colseg_sum_f32(p(st.vals), p(st.gz), p(st.perm), p(st.col_uid), p(st.col_seg), p(st.col_nu), B, F,
                                 p(st.uniq_ids), p(st.g_embed_rows), p(st.g_w_rows), p(st.n_uniq), s), "colseg")


def timeit(fn, reps=50):
    fn(C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(reps):
            fn(s)
    gr.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


print("fwd_bwd + reduce : %7.2f us" % timeit(fwd_bwd))
print("sort chain (3)   : %7.2f us" % timeit(sort_chain))
print("colseg_sum       : %7.2f us" % timeit(colseg))
