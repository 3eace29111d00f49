"""CPU checks of the drop-in boundary: the C-ABI shared library loads without a GPU, exports every symbol that
include/mi355rec.h declares, the ctypes table binds exactly that set, and the host-side mirror keeps the reference's
constructor keywords and error behaviour (no compute calls here: there is no GPU in this container)."""
import ctypes
import inspect
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mi355rec.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rec_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from explicit_tf2_recommendation_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 40
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(_lib.SIGNATURES) == syms           # the binding table and the header agree, symbol for symbol
    assert _lib.lib.rec_version() >= 100


def test_argument_errors_do_not_need_a_gpu():
    """Status codes of the ABI: invalid arguments are rejected on the host before anything is enqueued."""
    from explicit_tf2_recommendation_amd._lib import lib, check
    assert lib.rec_emb_gather_f32(None, 10, 4, 4, None, 5, None, None, None) == -1          # null table, n > 0
    assert lib.rec_emb_gather_f32(None, 10, 4, 2, None, 0, None, None, None) == -1          # ld < E
    assert lib.rec_emb_gather_f32(None, 10, 4, 4, None, 0, None, None, None) == 0           # empty batch is a no-op
    assert lib.rec_gemm_f32(0, 0, 4, 4, 4, None, 4, None, 4, None, 4, 0, None, None, 0, None, 0, 1, None, None,
                            None) == -1
    assert lib.rec_dedup_workspace_bytes(0) > 0
    with pytest.raises(ValueError):
        check(-1, "x")
    with pytest.raises(NotImplementedError):
        check(-2, "x")


def test_layer_signatures_match_the_reference():
    """Constructor keywords of 2.FM/CustomLayers.py:117,167,220-222,255-256; 3.DCN/CustomLayers.py:171,220-224,273;
    5.DIN/CustomLayers.py:164,200-205 (reference spelling kept, e.g. `is_dropput`)."""
    from explicit_tf2_recommendation_amd import layers as CL
    want = {
        CL.MLPLayer: ["units", "activation", "use_bias", "is_batch_norm", "is_dropput", "kernel_initializer",
                      "bias_initializer"],
        CL.FMRankingLayer: ["feature_names", "feature_dims", "embedding_dims"],
        CL.DeepFMRankingLayer: ["feature_names", "feature_dims", "embedding_dims", "mlp_dims"],
        CL.DSSMSingleTowerLayer: ["feature_names", "feature_dims", "embedding_dims", "mlp_dims", "final_dim"],
        CL.DSSMTwoTowerRetrievalLayer: ["u_feature_names", "i_feature_names", "u_feature_dims", "i_feature_dims",
                                        "u_embedding_dims", "i_embedding_dims", "u_mlp_dims", "i_mlp_dims", "final_dim"],
        CL.CrossLayer: ["layer_num", "reg_w", "reg_b"],
        CL.MatrixCrossLayer: ["layer_num", "reg_w", "reg_b"],
        CL.DeepCrossNetworkLayer: ["categorical_features", "continuous_features", "feature_dims", "embedding_dims",
                                   "units", "activation", "layer_num", "reg_w", "reg_b", "type"],
        CL.DinActivationLayer: ["activation"],
        CL.DINLayer: ["user_and_context_categorical_features", "item_categorical_features",
                      "behavior_series_features", "continuous_features", "feature_dims", "embedding_dims", "activation",
                      "padding_index"],
    }
    for cls, names in want.items():
        params = list(inspect.signature(cls.__init__).parameters)[1:]
        assert params[: len(names)] == names, (cls.__name__, params)
    # reference defaults
    assert inspect.signature(CL.DSSMSingleTowerLayer.__init__).parameters["embedding_dims"].default == 8
    assert inspect.signature(CL.DeepFMRankingLayer.__init__).parameters["mlp_dims"].default == [32, 8]
    assert inspect.signature(CL.DeepCrossNetworkLayer.__init__).parameters["type"].default == "vec"
    assert inspect.signature(CL.DINLayer.__init__).parameters["activation"].default == "Dice"


def test_host_side_errors_match_the_reference():
    from explicit_tf2_recommendation_amd import layers as CL
    with pytest.raises(ValueError):                      # 2.FM/CustomLayers.py:27-30
        CL.MLPLayer(units=[])
    with pytest.raises(AssertionError):                  # 5.DIN/CustomLayers.py:209-210
        CL.DINLayer(item_categorical_features=["a", "b"], behavior_series_features=["x"], feature_dims=10)
    layer = CL.FMRankingLayer(feature_names=["a"], feature_dims=10)
    import torch
    with pytest.raises(RuntimeError):                    # the HIP path has no CPU fallback: CPU tensors are refused
        from explicit_tf2_recommendation_amd import ops
        ops.emb_gather(torch.zeros(4, 4), torch.zeros(2, dtype=torch.int64))
    assert [n for n, _ in layer.named_parameters()] == ["bias", "embed.embeddings", "w.embeddings"]


def test_synthetic_generator_honours_the_datagenerator_contract():
    """2.FM/DataGenerator.py:76-88,126-134: one global id space, field f owns [offset_f, offset_f + dim_f)."""
    import numpy as np
    from explicit_tf2_recommendation_amd import data
    V, names = 1003, ["a", "b", "c"]
    info = data.data_info(V, 3)
    assert info[-1] == V and sum(info[0]) == V and info[1] == [0, info[0][0], info[0][0] + info[0][1]]
    for dist in ("uniform", "zipf"):
        g = data.SyntheticGenerator(names, V, dist=dist, seed=1)
        b = g.batch(500)
        assert b["label"].dtype == np.float32 and b["label"].shape == (500, 1)
        for f, n in enumerate(names):
            x = b[n]
            assert x.dtype == np.int64 and x.shape == (500, 1)
            assert x.min() >= info[1][f] and x.max() < info[1][f] + info[0][f]
    g = data.SyntheticGenerator(["u"], 900, series=["s1", "s2"], seq_len=7, seed=2)
    b = g.batch(64)
    assert b["s1"].shape == (64, 7)
    pad = b["s1"] == 0
    assert np.array_equal(pad, b["s2"] == 0)            # right padding with padding_index on every series feature
    assert np.all(pad[:, 1:] >= pad[:, :-1])            # once padded, padded to the end
    a = data.SyntheticGenerator(names, V, seed=5).batch(10)
    c = data.SyntheticGenerator(names, V, seed=5).batch(10)
    assert all(np.array_equal(a[k], c[k]) for k in a)   # seeded


def test_torch_library_registration():
    """the hot operators are registered with the PyTorch dispatcher as TORCH_LIBRARY(mi355rec, ...)
    (csrc/torch_ops.cpp, a host-only wrapper over the same C ABI); ops.py calls them as torch.ops.mi355rec.<op>."""
    import torch
    from explicit_tf2_recommendation_amd import _lib, ops  # noqa: F401  (importing the package loads the library)
    assert os.path.exists(_lib.TORCH_LIB_PATH)
    names = ["index_pack", "emb_gather", "emb_fm_fwd", "emb_fm_bwd_vals", "gemm", "act_fwd", "act_bwd", "colsum",
             "bce_fwd_bwd", "dedup_plan", "segment_sum", "cosine_fwd", "cosine_bwd", "crossnet_mat_bwd_elem", "axpby"]
    for n in names:
        op = getattr(torch.ops.mi355rec, n)
        assert "mi355rec::" + n in str(op.default._schema)
    assert "Tensor A, Tensor B, bool transA, bool transB" in str(torch.ops.mi355rec.gemm.default._schema)
    # no CPU kernel is registered: the dispatcher refuses CPU tensors (the HIP path has no fallback)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.mi355rec.cosine_fwd(torch.zeros(2, 4), torch.zeros(2, 4))

