#!/usr/bin/env python3
"""Diagnostic: rows of the de-duplicated table gradient, post3 form against the post_direct form, and what kind of run the
rows that differ belong to."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from explicit_tf2_recommendation_amd import layers, data, engine  # noqa: E402

V, F, E, B = int(os.environ.get("V", "1000000")), 26, 16, int(os.environ.get("B", "8192"))
names = ["C%d" % (i + 1) for i in range(F)]
layers.set_init_seed(1234)
L = layers.DeepFMRankingLayer(feature_names=names, feature_dims=V, embedding_dims=E, mlp_dims=[32, 8]).cuda()
gen = data.SyntheticGenerator(names, V, dist=os.environ.get("DIST", "uniform"), seed=0)
batch = data.to_device(gen.batch(B))
old = engine.DeepFMFusedStep(L, B, gen.dims, gen.offsets, optimizer=None, use_graph=False, kernel=2)
new = engine.DeepFMFusedStep(L, B, gen.dims, gen.offsets, optimizer=None, use_graph=False)
assert new._pipelined and not old._pipelined
old(batch); new(batch)
torch.cuda.synchronize()
nu = int(old.n_uniq.item())
print("n_uniq", nu, int(new.n_uniq.item()), "ids equal", bool(torch.equal(old.uniq_ids[:nu], new.uniq_ids[:nu])))
a, b = old.g_embed_rows[:nu], new.g_embed_rows[:nu]
bad = ((a - b).abs() > 1e-6 * a.abs().max()).any(dim=1).nonzero().flatten().cpu().numpy()
badw = ((old.g_w_rows[:nu] - new.g_w_rows[:nu]).abs() > 1e-6 * old.g_w_rows[:nu].abs().max()).any(dim=1).nonzero().flatten().cpu().numpy()
print("rows that differ:", len(bad), " w rows that differ:", len(badw))
pl = new.plans[[k for k in range(new.NBUF)][0]]
# the plan the last call used: find by col_nu sum
for k in range(new.NBUF):
    if int(new.plans[k]["col_nu"].sum().item()) == nu:
        pl = new.plans[k]
        break
col_nu = pl["col_nu"].cpu().numpy()
seg = pl["col_seg"].cpu().numpy()
before = np.concatenate([[0], np.cumsum(col_nu)])
per_wg = -(-F * B // ((B + 31) // 32))
for d in bad[:40]:
    f = int(np.searchsorted(before, d, side="right") - 1)
    u = int(d - before[f])
    ln = int(seg[f, u + 1] - seg[f, u])
    t = f * B + u
    wg, rem = divmod(t, per_wg)
    it, th = divmod(rem, 256)
    # rank among the two-member runs of the same wave and iteration
    lane0 = t - (th & 63)
    rank = sum(1 for tt in range(lane0, t) if (lambda ff, uu: uu < col_nu[ff] and seg[ff, uu + 1] - seg[ff, uu] == 2)(tt // B, tt % B))
    print("slot %7d  col %2d run %5d len %d | wg %3d it %d wave %d lane %2d | rank among 2-runs %d | err %.3e of %.3e"
          % (d, f, u, ln, wg, it, th >> 6, th & 63, rank, (a[d] - b[d]).abs().max().item(), a[d].abs().max().item()))
lens = []
for d in bad:
    f = int(np.searchsorted(before, d, side="right") - 1)
    u = int(d - before[f])
    lens.append(int(seg[f, u + 1] - seg[f, u]))
print("lengths of the runs that differ:", np.bincount(np.array(lens, dtype=np.int64)) if lens else [])
