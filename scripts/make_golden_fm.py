#!/usr/bin/env python3
"""Generate tests/golden/fm_ckpt2_weights.npz  (KAT-3 of SURVEY.md 8c: trained weights for the FM / DeepFM value range).

Runs ONLY in the build container (needs /root/reference).  It reads a *data* artifact the reference ships -- no reference
code is imported or executed: 2.FM/ranking_model/checkpoint/ckpt-2.{index,data-00000-of-00001}, the DeepFM
(`deepfm_ranking`) checkpoint of the reference's best epoch (eval AUC 0.9271), through oracle/tensorbundle.py (a pure
reader of the TensorBundle format; nothing in the file is executed).

The checkpoint holds no outputs, so this fixture pins no result: it supplies the TRAINED value range of
  embed/embeddings [5547,16], w/embeddings [5547,1], bias [1]            (2.FM/CustomLayers.py:125-134, 262-274)
  MLP_layer1/kernel_0 [80,32], bias_0 [32], kernel_1 [32,8], bias_1 [8]  (:275, 5 fields x 16d -> 32 -> 8)
for the FM / DeepFM parity tests (the random-initialised tables of the other tests sit inside U(-0.05, 0.05); trained rows
reach |x| ~ 1 and the first-order weights ~ 3).  MLP_layer2/kernel_0 is [10,1] in this checkpoint -- an older revision of
the head (current code: [8,1], 2.FM/CustomLayers.py:276,301) -- and is NOT taken; tests draw that layer themselves.
Field layout of the 5547 ids (recovered for the DSSM fixture, scripts/make_golden_dssm.py): user_tag1 [0,3),
user_tag2 [3,30), item_tag1 [30,215), item_tag2 [215,5200), item_tag3 [5200,5547).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.tensorbundle import read_index, read_tensor  # noqa: E402

CKPT = "/root/reference/2.FM/ranking_model/checkpoint/ckpt-2"
PRE, SUF = "model/layer_with_weights-0/", "/.ATTRIBUTES/VARIABLE_VALUE"


def main():
    index = read_index(CKPT + ".index")
    g = lambda n: read_tensor(CKPT, PRE + n + SUF, index).astype(np.float32)
    out = {"embed": g("embed/embeddings"), "w": g("w/embeddings"), "bias": g("bias"),
           "k0": g("MLP_layer1/kernel_0"), "b0": g("MLP_layer1/bias_0"),
           "k1": g("MLP_layer1/kernel_1"), "b1": g("MLP_layer1/bias_1"),
           "field_offsets": np.array([0, 3, 30, 215, 5200], np.int64),
           "field_dims": np.array([3, 27, 185, 4985, 347], np.int64)}
    assert out["embed"].shape == (5547, 16) and out["w"].shape == (5547, 1) and out["k0"].shape == (80, 32)
    assert index[PRE + "MLP_layer2/kernel_0" + SUF]["shape"] == [10, 1]        # the older head: not taken
    path = os.path.join(ROOT, "tests", "golden", "fm_ckpt2_weights.npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%d bytes): max|embed| %.3f, max|w| %.3f, bias %.4f" %
          (path, os.path.getsize(path), np.abs(out["embed"]).max(), np.abs(out["w"]).max(), out["bias"][0]))


if __name__ == "__main__":
    main()
