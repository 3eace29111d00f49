#!/usr/bin/env python3
"""DEBUG: dense sweep k times vs one catch-up over k pending steps, same rows: bit equality."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from explicit_tf2_recommendation_amd._lib import lib, check
vp = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
V, K = 1000, 5
g = torch.Generator(device="cuda").manual_seed(0)
fused = torch.randn(V, 32, device="cuda", generator=g) * 0.1
me = torch.randn(V, 16, device="cuda", generator=g) * 0.01
ve = torch.rand(V, 16, device="cuda", generator=g) * 1e-4
mw = torch.randn(V, device="cuda", generator=g) * 0.01
vw = torch.rand(V, device="cuda", generator=g) * 1e-4
lr, b1, b2, eps = 0.01, 0.9, 0.999, 1e-7
# A: K dense sweeps with no touched rows
fa, mea, vea, mwa, vwa = fused.clone(), me.clone(), ve.clone(), mw.clone(), vw.clone()
uniq = torch.zeros(1, dtype=torch.int64, device="cuda"); ge = torch.zeros(1, 16, device="cuda"); gw = torch.zeros(1, 1, device="cuda")
nu = torch.zeros(1, dtype=torch.int64, device="cuda"); se = torch.zeros(1, 3, 16, device="cuda"); sw = torch.zeros(1, 3, 1, device="cuda")
for t in range(1, K + 1):
    check(lib.rec_adam_sparse_keras_pair_f32(vp(fa), 32, vp(mea), vp(vea), vp(mwa), vp(vwa), V, 16, vp(uniq), vp(ge), vp(gw),
                                             vp(nu), 1, vp(se), vp(sw), t, lr, b1, b2, eps, st), "pair")
# B: flush with last = 0, step = K
tab = torch.tensor([lib.rec_adam_lr_t_f32(lr, b1, b2, t) for t in range(1, 65)], device="cuda")
fb, meb, veb, mwb, vwb = fused.clone(), me.clone(), ve.clone(), mw.clone(), vw.clone()
last = torch.zeros(V, dtype=torch.int32, device="cuda"); step = torch.full((1,), K, dtype=torch.int64, device="cuda")
check(lib.rec_adam_keras_flush_f32(vp(fb), 32, V, vp(meb), vp(veb), 16, vp(mwb), vp(vwb), 1, vp(last), vp(step), vp(tab), 64,
                                   b1, b2, eps, st), "flush")
torch.cuda.synchronize()
print("table[:, :17] equal:", torch.equal(fa[:, :17], fb[:, :17]), " m_e:", torch.equal(mea, meb), " v_e:", torch.equal(vea, veb),
      " m_w:", torch.equal(mwa, mwb), " v_w:", torch.equal(vwa, vwb), " last:", int(last.min()), int(last.max()))
d = (fa[:, :17] - fb[:, :17]).abs()
print("max diff", d.max().item(), "n diff", int((d > 0).sum()), "cols with diff", torch.nonzero(d.sum(0) > 0).flatten().tolist()[:20])
