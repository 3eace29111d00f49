"""Standalone timing of deepfm_fwd_bwd_kernel (rec_deepfm_fused_main_f32): `reps` launches captured into a hipGraph and
replayed between HIP events.  With REC_FUSED_STOP=N (diagnostics switch of the library) the kernel leaves after phase N,
which gives the cumulative cost of the phases (scripts/exp/time_phases.sh).  Run on the GPU box from the repo root."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
from explicit_tf2_recommendation_amd import layers, engine, data  # noqa: E402
from explicit_tf2_recommendation_amd._lib import lib, check  # noqa: E402

V, F, E, B = 10_000_000, 26, 16, int(os.environ.get("B", 8192))
names = ["C%d" % (i + 1) for i in range(F)]
layer = layers.DeepFMRankingLayer(feature_names=names, feature_dims=V, embedding_dims=E, mlp_dims=[32, 8]).cuda()
gen = data.SyntheticGenerator(names, V, dist=sys.argv[1] if len(sys.argv) > 1 else "uniform", seed=0)
batch = data.to_device(gen.batch(B))
st = engine.DeepFMFusedStep(layer, B, gen.dims, gen.offsets, use_graph=False)
p = engine._p
cols = [batch[n] for n in names]
arr = (C.c_void_p * F)(*[c.data_ptr() for c in cols])
L = layer
emb = L.embed.embeddings


def fwd_bwd(s):
    check(lib.rec_deepfm_fused_main_f32(
        p(emb), emb.stride(0), V, arr, F, B, p(L.bias), p(L.MLP_layer1.kernel_0), p(L.MLP_layer1.bias_0),
        p(L.MLP_layer1.kernel_1), p(L.MLP_layer1.bias_1), p(L.MLP_layer2.kernel_0), p(L.MLP_layer2.bias_0),
        p(batch["label"]), p(st.gz), p(st.vals), None, p(st.oob), p(st.ws), s), "fwd_bwd")


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


print("fwd_bwd %.2f us" % timeit(fwd_bwd))
