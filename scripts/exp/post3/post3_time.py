#!/usr/bin/env python3
"""Diagnostic: gradient-only iterations of the fused step at the BASELINE config, three ways, graph-replayed over fresh
batches (plans prebuilt): (a) fused launch + post launch of its own (deepfm_post_direct), (b) fused launch + post3 launch,
(c) ONE launch per iteration (the post step of iteration k-1 inside launch k).
    python scripts/exp/post3_time.py
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from explicit_tf2_recommendation_amd import layers, data, engine  # noqa: E402

V, F, E, B = 10_000_000, 26, 16, int(os.environ.get("B", "8192"))
names = ["C%d" % (i + 1) for i in range(F)]
layers.set_init_seed(1234)
L = layers.DeepFMRankingLayer(feature_names=names, feature_dims=V, embedding_dims=E, mlp_dims=[32, 8]).cuda()
gen = data.SyntheticGenerator(names, V, dist=os.environ.get("DIST", "uniform"), seed=0)
NB = 16
batches = [data.to_device(gen.batch(B)) for _ in range(NB)]
fs = engine.DeepFMFusedStep(L, B, gen.dims, gen.offsets, optimizer=None, use_graph=False)
assert fs._pipelined, "post3 form not available for this shape"
colss = [fs._cols(b) for b in batches]
for i in range(NB):
    fs._sort(colss[i], i, torch.cuda.current_stream())
torch.cuda.synchronize()
N = 3 * NB


def st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def form_a():
    for i in range(N):
        fs._row = 0
        fs._launch_main(colss[i % NB], batches[i % NB]["label"], st(), i % NB, par=0)
        fs._launch_post(i % NB, st())


def form_b():
    for i in range(N):
        fs._row = 0
        fs._launch_main(colss[i % NB], batches[i % NB]["label"], st(), i % NB, par=0)
        fs._launch_post3(0, i % NB, 0, st())


def form_c():
    fs._row = 0
    fs._launch_main(colss[0], batches[0]["label"], st(), 0, par=0)
    for i in range(1, N):
        fs._launch_main(colss[i % NB], batches[i % NB]["label"], st(), i % NB, par=i & 1,
                        prev=((i - 1) & 1, (i - 1) % NB, 0))
    fs._launch_post3((N - 1) & 1, (N - 1) % NB, 0, st())


side = torch.cuda.Stream()


def form_d():
    """fused launches back to back on the main stream; the post step of every iteration on a second stream (needs the
    iteration's fused launch; the fused launch two iterations later reuses its result buffers)"""
    main = torch.cuda.current_stream()
    done = [None, None]
    for i in range(N):
        par = i & 1
        if done[par] is not None:
            main.wait_event(done[par])
        fs._row = 0
        fs._launch_main(colss[i % NB], batches[i % NB]["label"], C.c_void_p(main.cuda_stream), i % NB, par=par)
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        fs._launch_post3(par, i % NB, 0, C.c_void_p(side.cuda_stream))
        done[par] = torch.cuda.Event()
        done[par].record(side)
    main.wait_stream(side)


def main_only():
    for i in range(N):
        fs._launch_main(colss[i % NB], batches[i % NB]["label"], st(), i % NB, par=0)


def post3_only():
    for i in range(N):
        fs._launch_post3(0, i % NB, 0, st())


for name, fn in (("fused launch alone", main_only), ("post3 launch alone", post3_only),
                 ("(a) fused + post_direct launch", form_a), ("(b) fused + post3 launch", form_b),
                 ("(c) one launch per iteration", form_c), ("(d) post3 on a second stream", form_d)):
    fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, capture_error_mode=engine.CAPTURE_MODE):
        fn()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(7):
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / N)
    print("%-36s %.2f us per iteration (min of 7; median %.2f)" % (name, min(ts), sorted(ts)[3]))
fs.check_flags()
