"""GPU parity tests of the C-ABI entry points against the oracle (run with -m gpu on the MI355X).

Bar: bit-exact for integer / index work; fp32 within the stated tolerance of the fp64 oracle.
Every call goes through ctypes into csrc/libmi355rec.so.
"""
import numpy as np
import pytest
import torch

from oracle import layers_np as L
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    from explicit_tf2_recommendation_amd import ops as _ops
    return _ops


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def field_ids(seed, B, F, V, zipf=None):
    r = H.rng(seed)
    dims = [V // F] * F
    dims[-1] += V - sum(dims)
    off = np.concatenate([[0], np.cumsum(dims)[:-1]])
    cols = []
    for f in range(F):
        if zipf:
            x = np.minimum(r.zipf(zipf, size=B) - 1, dims[f] - 1)
        else:
            x = r.integers(0, dims[f], size=B)
        cols.append(off[f] + x)
    return np.stack(cols, axis=1).astype(np.int64)


# ---------------------------------------------------------------------------------------------
# K1 index pack (bit exact)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,F", [(1, 1), (257, 5), (8192, 26), (100, 70)])
def test_index_pack_bit_exact(ops, B, F):
    r = H.rng(B + F)
    cols = [r.integers(0, 2 ** 40, size=(B, 1)) if f % 2 else r.integers(0, 2 ** 40, size=(B,)) for f in range(F)]
    X = ops.index_pack([dev(c) for c in cols]).cpu().numpy()
    ref = L.index_assemble({str(f): c for f, c in enumerate(cols)}, [str(f) for f in range(F)])
    assert X.dtype == np.int64 and np.array_equal(X, ref)


def test_index_pack_series_stack(ops):
    """tf.stack(axis=2) of [B,T] series (5.DIN/CustomLayers.py:258)."""
    r = H.rng(3)
    B, T = 37, 11
    series = [r.integers(0, 1000, size=(B, T)) for _ in range(3)]
    X = ops.index_pack([dev(s) for s in series]).cpu().numpy().reshape(B, T, 3)
    assert np.array_equal(X, np.stack(series, axis=2))


# ---------------------------------------------------------------------------------------------
# K2 gather (bit exact) + out-of-range flag
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("E", [1, 3, 8, 16, 32, 64, 20])
def test_gather_bit_exact(ops, E):
    r = H.rng(E)
    V = 1000
    tab = r.normal(size=(V, E)).astype(np.float32)
    idx = r.integers(0, V, size=(300, 7))
    out = ops.emb_gather(dev(tab), dev(idx)).cpu().numpy()
    assert np.array_equal(out, tab[idx])


def test_gather_empty_and_oob(ops):
    tab = dev(H.det_table(10, 4))
    out = ops.emb_gather(tab, torch.empty((0,), dtype=torch.int64, device="cuda"))
    assert out.shape == (0, 4)
    flag = ops.new_flag("cuda")
    out = ops.emb_gather(tab, dev(np.array([3, 10, -1, 9])), oob=flag).cpu().numpy()
    assert flag.item() == 1
    assert np.array_equal(out[0], H.det_table(10, 4)[3]) and np.all(out[1] == 0) and np.all(out[2] == 0)
    flag.zero_()
    ops.emb_gather(tab, dev(np.array([0, 9])), oob=flag)
    assert flag.item() == 0


# ---------------------------------------------------------------------------------------------
# K2+K3 fused FM forward:  |z - z64| <= 1e-5 (north_star tolerance), rows bit exact
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,F,E,V", [(256, 5, 16, 5547), (1000, 26, 16, 100000), (77, 3, 8, 50), (64, 10, 32, 3000),
                                      (50, 3, 64, 500), (33, 8, 3, 200), (19, 4, 20, 100), (9, 2, 1, 40),
                                      (5, 3, 100, 64)])
def test_emb_fm_fwd(ops, B, F, E, V):
    pr = H.deepfm_params(B + E, V, F, E, scale=0.5 if E <= 16 else 0.2)
    X = field_ids(B, B, F, V)
    z, prob, rows, S = ops.emb_fm_fwd(dev(pr["embed"]), dev(pr["w"]), dev(pr["bias"]), dev(X), want_prob=True,
                                      want_rows=True)
    p64, z64 = L.fm_forward(pr["embed"], pr["w"], pr["bias"], X, np.float64)
    scale = max(1.0, np.abs(z64).max())
    assert np.abs(z.cpu().numpy() - z64[:, 0]).max() <= 1e-5 * scale
    assert np.abs(prob.cpu().numpy() - p64[:, 0]).max() <= 1e-5
    assert np.array_equal(rows.cpu().numpy(), pr["embed"][X])
    assert np.abs(S.cpu().numpy() - pr["embed"][X].astype(np.float64).sum(1)).max() <= 1e-5


def test_emb_fm_fwd_oob_flag(ops):
    pr = H.deepfm_params(1, 100, 3, 16)
    X = field_ids(2, 8, 3, 100)
    X[5, 1] = 100
    flag = ops.new_flag("cuda")
    ops.emb_fm_fwd(dev(pr["embed"]), dev(pr["w"]), dev(pr["bias"]), dev(X), oob=flag)
    assert flag.item() == 1


# ---------------------------------------------------------------------------------------------
# K4: values + de-duplication
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("E,zipf", [(16, None), (16, 1.05), (8, 1.2), (3, 1.05), (1, None)])
def test_fm_bwd_vals_and_dedup(ops, E, zipf):
    B, F, V = 512, 6, 4000
    pr = H.deepfm_params(E, V, F, E, scale=0.5)
    X = field_ids(7, B, F, V, zipf)
    gz = H.rng(8).normal(size=(B, 1)).astype(np.float32)
    extra = H.rng(9).normal(size=(B, F, E)).astype(np.float32)
    emb, Xd = dev(pr["embed"]), dev(X)
    _, _, rows, S = ops.emb_fm_fwd(emb, dev(pr["w"]), dev(pr["bias"]), Xd, want_rows=True)
    for use_rows in (True, False):
        vals = ops.emb_fm_bwd_vals(emb, Xd, dev(gz[:, 0]), S, rows if use_rows else None, dev(extra))
        ref, dw_ref, _ = L.fm_backward(pr["embed"], X, gz, np.float64)
        ref = ref + extra.reshape(-1, E)
        assert np.abs(vals.cpu().numpy() - ref).max() <= 2e-6 * max(1, np.abs(ref).max())
    plan = ops.DedupPlan(Xd, V)
    nu = int(plan.n_uniq.item())
    uniq_ref, sum_ref = L.dedup_indexed_slices(X.reshape(-1), vals.cpu().numpy().astype(np.float64), "sorted")
    assert nu == uniq_ref.shape[0]
    uid = plan.uniq_ids.cpu().numpy()
    assert np.array_equal(uid[:nu], uniq_ref)                       # ids: bit exact, ascending
    assert np.all(uid[nu:] == uniq_ref[0])                          # padded tail = a valid id ...
    seg = plan.seg_start.cpu().numpy()
    assert seg[nu] == X.size and np.all(seg[nu:] == X.size)         # ... with empty runs
    perm = plan.perm.cpu().numpy()
    assert sorted(perm.tolist()) == list(range(X.size))
    assert np.array_equal(X.reshape(-1)[perm], np.sort(X.reshape(-1), kind="stable"))
    for u in range(nu):                                             # stable: positions ascending in a run
        run = perm[seg[u]:seg[u + 1]]
        assert np.all(np.diff(run) > 0)
    out = plan.segment_sum(vals, E).cpu().numpy()
    assert np.abs(out[:nu] - sum_ref).max() <= 1e-5 * max(1, np.abs(sum_ref).max())
    assert np.all(out[nu:] == 0)
    # w table: per-lookup gradient is gz[b] (row_div = F)
    outw = plan.segment_sum(dev(gz), 1, row_div=F).cpu().numpy()
    uw, sw = L.dedup_indexed_slices(X.reshape(-1), np.repeat(gz.astype(np.float64), F, axis=0), "sorted")
    assert np.abs(outw[:nu] - sw).max() <= 1e-5 * max(1, np.abs(sw).max())
    # run-to-run bit identical
    out2 = ops.DedupPlan(Xd, V).segment_sum(vals, E).cpu().numpy()
    assert np.array_equal(out, out2)


def test_dedup_all_equal_and_single(ops):
    ids = dev(np.full(5000, 7, dtype=np.int64))
    plan = ops.DedupPlan(ids, 10)
    assert plan.n_uniq.item() == 1 and plan.uniq_ids[0].item() == 7
    vals = dev(np.ones((5000, 4), np.float32))
    assert np.all(plan.segment_sum(vals, 4).cpu().numpy()[0] == 5000.0)
    plan = ops.DedupPlan(dev(np.array([3], dtype=np.int64)), 10)
    assert plan.n_uniq.item() == 1 and plan.seg_start.cpu().tolist() == [0, 1]


# ---------------------------------------------------------------------------------------------
# GEMM
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(64, 64, 16), (8192, 32, 416), (100, 8, 32), (77, 1, 8), (323, 323, 323),
                                    (130, 70, 1), (5, 3, 2)])
@pytest.mark.parametrize("tA,tB", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_layouts(ops, M, N, K, tA, tB):
    r = H.rng(M * 7 + N * 3 + K)
    A = r.normal(size=(M, K)).astype(np.float32)
    B = r.normal(size=(K, N)).astype(np.float32)
    Ad = dev(A.T.copy() if tA else A)
    Bd = dev(B.T.copy() if tB else B)
    C = ops.gemm(Ad, Bd, tA, tB).cpu().numpy()
    ref = A.astype(np.float64) @ B.astype(np.float64)
    tol = 2e-6 * np.sqrt(K) * max(1.0, np.abs(ref).max())
    assert np.abs(C - ref).max() <= tol


def test_gemm_asymmetric_identity(ops):
    """A = I with an asymmetric B catches a transposed C write (cdna guide, section 3)."""
    n = 96
    Bm = (np.arange(n * n, dtype=np.float32).reshape(n, n) % 251) / 7.0
    C = ops.gemm(dev(np.eye(n, dtype=np.float32)), dev(Bm)).cpu().numpy()
    assert np.array_equal(C, Bm)


@pytest.mark.parametrize("epi", ["bias", "relu", "sigmoid", "tanh", "cross", "add"])
def test_gemm_epilogues(ops, epi):
    r = H.rng(11)
    M, N, K = 150, 45, 37
    A = r.normal(size=(M, K)).astype(np.float32)
    B = r.normal(size=(K, N)).astype(np.float32)
    bias = r.normal(size=(N,)).astype(np.float32)
    e0 = r.normal(size=(M, N)).astype(np.float32)
    e1 = r.normal(size=(M, N)).astype(np.float32)
    acc = A.astype(np.float64) @ B + bias
    code = {"bias": ops.EPI_BIAS, "relu": ops.EPI_BIAS_RELU, "sigmoid": ops.EPI_BIAS_SIGMOID,
            "tanh": ops.EPI_BIAS_TANH, "cross": ops.EPI_CROSS, "add": ops.EPI_ADD}[epi]
    ref = {"bias": acc, "relu": np.maximum(acc, 0), "sigmoid": 1 / (1 + np.exp(-acc)), "tanh": np.tanh(acc),
           "cross": e0 * acc + e1, "add": acc - bias + e1}[epi]
    C = ops.gemm(dev(A), dev(B), epi=code, bias=dev(bias), e0=dev(e0), e1=dev(e1)).cpu().numpy()
    assert np.abs(C - ref).max() <= 2e-5 * max(1, np.abs(ref).max())


@pytest.mark.parametrize("split", [2, 7, 64])
def test_gemm_split_k_deterministic(ops, split):
    r = H.rng(split)
    K, M, N = 4096, 48, 32
    X = r.normal(size=(K, M)).astype(np.float32)
    dY = r.normal(size=(K, N)).astype(np.float32)
    C1 = ops.gemm(dev(X), dev(dY), transA=True, split_k=split).cpu().numpy()
    C2 = ops.gemm(dev(X), dev(dY), transA=True, split_k=split).cpu().numpy()
    ref = X.astype(np.float64).T @ dY
    assert np.array_equal(C1, C2)
    assert np.abs(C1 - ref).max() <= 2e-6 * np.sqrt(K) * np.abs(ref).max()


@pytest.mark.parametrize("M,N,K", [(256, 32, 741), (8192, 32, 741), (257, 1, 8), (1000, 8, 32), (4096, 64, 192),
                                    (300, 33, 17), (511, 2, 80), (2048, 64, 1), (290, 20, 129)])
def test_gemm_tall_skinny(ops, M, N, K):
    """N <= 64, row-major operands, M >= 256: the no-LDS kernel whose 4 waves split K (every MLP forward layer)."""
    r = H.rng(M + N + K)
    A = r.normal(size=(M, K)).astype(np.float32)
    B = r.normal(size=(K, N)).astype(np.float32)
    bias = r.normal(size=(N,)).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    tol = 2e-6 * np.sqrt(K) * max(1.0, np.abs(ref).max())
    C = ops.gemm(dev(A), dev(B)).cpu().numpy()
    assert np.abs(C - ref).max() <= tol
    assert np.array_equal(C, ops.gemm(dev(A), dev(B)).cpu().numpy())                 # deterministic
    Cr = ops.gemm(dev(A), dev(B), epi=ops.EPI_BIAS_RELU, bias=dev(bias)).cpu().numpy()
    assert np.abs(Cr - np.maximum(ref + bias, 0)).max() <= tol
    # A as a column slice of a wider matrix (leading dimension != K), C into a column slice of a wider buffer
    wide = torch.zeros((M, K + 5), device="cuda")
    wide[:, 3:3 + K] = dev(A)
    outw = torch.full((M, N + 7), -1.0, device="cuda")
    ops.gemm(wide[:, 3:3 + K], dev(B), out=outw[:, 2:2 + N])
    assert np.abs(outw[:, 2:2 + N].cpu().numpy() - ref).max() <= tol
    assert torch.all(outw[:, :2] == -1) and torch.all(outw[:, 2 + N:] == -1)
    e0 = r.normal(size=(M, N)).astype(np.float32)
    e1 = r.normal(size=(M, N)).astype(np.float32)
    aux = torch.empty((M, N), device="cuda")
    Cc = ops.gemm(dev(A), dev(B), epi=ops.EPI_CROSS, bias=dev(bias), e0=dev(e0), e1=dev(e1), aux=aux).cpu().numpy()
    assert np.abs(Cc - (e0 * (ref + bias) + e1)).max() <= 4 * tol
    assert np.abs(aux.cpu().numpy() - (ref + bias)).max() <= tol


@pytest.mark.parametrize("M,N,strided", [(4096, 36, False), (4096, 1, False), (1024, 256, True), (1025, 40, False),
                                          (16384, 33, True), (300, 3492, False), (1, 7, False), (5000, 200, True)])
def test_colsum_one_pass_and_two_stage(ops, M, N, strided):
    """Column sums: the one-launch kernel (M <= 1024) and the two-stage one on both sides of the switch, contiguous and
    as a column slice of a wider matrix; fixed summation order -> run-to-run bit-identical."""
    r = H.rng(M + N)
    X = r.normal(size=(M, N + (5 if strided else 0))).astype(np.float32)
    Xd = dev(X)
    view = Xd[:, 2:2 + N] if strided else Xd
    ref = (X[:, 2:2 + N] if strided else X).astype(np.float64).sum(0)
    a = ops.colsum(view)
    b = ops.colsum(view)
    assert torch.equal(a, b)
    assert np.abs(a.cpu().numpy() - ref).max() <= 2e-6 * np.sqrt(M) * max(1.0, np.abs(ref).max())
    # the one-launch form (arrival counters; what torch.ops.mi355rec.colsum uses above 1024 rows) against the two-stage
    # form through the C ABI: the same summation order, bit for bit; the counters are zero again afterwards
    import ctypes as C
    from explicit_tf2_recommendation_amd._lib import lib, check
    vp = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ws = torch.empty(lib.rec_colsum_workspace_bytes(M, N) // 4 + 1, device="cuda")
    two, one = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    cnt = torch.zeros((N + 31) // 32, dtype=torch.int32, device="cuda")
    check(lib.rec_colsum_f32(vp(view), M, N, view.stride(0), vp(two), vp(ws), st), "rec_colsum_f32")
    for _ in range(3):
        check(lib.rec_colsum_fused_f32(vp(view), M, N, view.stride(0), vp(one), vp(ws), vp(cnt), st), "rec_colsum_fused_f32")
        if M > 1024:
            assert torch.equal(one, two)
        else:
            assert np.abs(one.cpu().numpy() - ref).max() <= 2e-6 * np.sqrt(M) * max(1.0, np.abs(ref).max())
        assert int(cnt.abs().sum().item()) == 0


def test_dense_helpers(ops):
    r = H.rng(12)
    post = r.normal(size=(100, 33)).astype(np.float32)
    g = r.normal(size=(100, 33)).astype(np.float32)
    relu_post = np.maximum(post, 0)
    assert np.array_equal(ops.act_bwd(ops.ACT_RELU, dev(relu_post), dev(g)).cpu().numpy(), g * (relu_post > 0))
    sg = 1 / (1 + np.exp(-post))
    assert np.allclose(ops.act_bwd(ops.ACT_SIGMOID, dev(sg), dev(g)).cpu().numpy(), g * sg * (1 - sg), atol=1e-6)
    assert np.abs(ops.colsum(dev(post)).cpu().numpy() - post.astype(np.float64).sum(0)).max() < 1e-4
    buf = torch.zeros((100, 50), device="cuda")
    ops.copy_cols(dev(post), buf[:, 10:43])
    assert np.array_equal(buf.cpu().numpy()[:, 10:43], post) and buf[:, :10].abs().sum().item() == 0


# ---------------------------------------------------------------------------------------------
# CrossNet vector mode
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,D,Lyr", [(64, 323, 3), (37, 835, 3), (5, 7, 1), (130, 64, 4)])
def test_crossnet_vec(ops, B, D, Lyr):
    r = H.rng(D)
    x0 = r.normal(size=(B, D)).astype(np.float32)
    ws = [r.normal(0, 0.05, size=(D, 1)).astype(np.float32) for _ in range(Lyr)]
    bs = [r.normal(0, 0.05, size=(D, 1)).astype(np.float32) for _ in range(Lyr)]
    w = dev(np.stack([x[:, 0] for x in ws]))
    b = dev(np.stack([x[:, 0] for x in bs]))
    y, xs = ops.crossnet_vec_fwd(dev(x0), w, b)
    ref = L.cross_vec_forward(x0, ws, bs, np.float64)
    assert np.abs(y.cpu().numpy() - ref).max() <= 1e-5 * max(1, np.abs(ref).max())
    gy = r.normal(size=(B, D)).astype(np.float32)
    gx0, dw, db = ops.crossnet_vec_bwd(dev(x0), w, xs, dev(gy))
    rgx0, rdw, rdb = L.cross_vec_backward(x0, ws, bs, gy, np.float64)
    assert np.abs(gx0.cpu().numpy() - rgx0).max() <= 2e-5 * max(1, np.abs(rgx0).max())
    rdw = np.stack([x[:, 0] for x in rdw])
    rdb = np.stack([x[:, 0] for x in rdb])
    assert np.abs(dw.cpu().numpy() - rdw).max() <= 2e-5 * max(1, np.abs(rdw).max())
    assert np.abs(db.cpu().numpy() - rdb).max() <= 2e-5 * max(1, np.abs(rdb).max())


# ---------------------------------------------------------------------------------------------
# cosine / BCE / Adam
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d", [8, 5, 64, 100])
def test_cosine(ops, d):
    r = H.rng(d)
    u = r.normal(size=(333, d)).astype(np.float32)
    i = r.normal(size=(333, d)).astype(np.float32)
    out = ops.cosine_fwd(dev(u), dev(i)).cpu().numpy()
    assert np.abs(out - L.two_tower_score(u, i, np.float64)).max() <= 1e-6
    g = r.normal(size=(333,)).astype(np.float32)
    gu, gi = ops.cosine_bwd(dev(u), dev(i), dev(g))
    rgu, rgi = L.two_tower_score_backward(u, i, g, np.float64)
    assert np.abs(gu.cpu().numpy() - rgu).max() <= 1e-5
    assert np.abs(gi.cpu().numpy() - rgi).max() <= 1e-5


@pytest.mark.parametrize("n", [1, 100, 8192, 70001])
def test_bce(ops, n):
    r = H.rng(n)
    y = (r.uniform(size=(n, 1)) < 0.25).astype(np.float32)
    p = r.uniform(0.005, 0.995, size=(n, 1)).astype(np.float32)
    loss, dp, dz = ops.bce_fwd_bwd(dev(y), dev(p), want_dp=True, want_dz=True)
    ref = L.bce_forward(y, p, np.float64)
    assert abs(loss.item() - ref) <= 1e-5 * max(1, abs(ref))
    rdp = L.bce_backward(y, p, np.float64)[:, 0]
    assert np.abs(dp.cpu().numpy() - rdp).max() <= 1e-5 * np.abs(rdp).max()
    assert np.abs(dz.cpu().numpy() - rdp * p[:, 0] * (1 - p[:, 0])).max() <= 1e-5 * np.abs(rdp).max()


def test_bce_saturated_matches_fp32_restatement(ops):
    """p = 0 / 1 exercise the clip; 1 - (1 - 1e-7) is not 1e-7 in fp32, so the fp32 op-for-op
    restatement (what TF computes) is the reference here, not fp64."""
    y = np.array([[1], [0], [0], [1], [1]], np.float32)
    p = np.array([[0.0], [1.0], [0.0], [1.0], [0.5]], np.float32)
    loss, dp, _ = ops.bce_fwd_bwd(dev(y), dev(p), want_dp=True)
    ref = L.bce_forward(y, p, np.float32)
    assert abs(loss.item() - ref) <= 1e-5 * abs(ref)
    rdp = L.bce_backward(y, p, np.float32)[:, 0]
    assert np.abs(dp.cpu().numpy() - rdp).max() <= 1e-5 * np.abs(rdp).max()


def test_adam_dense_and_sparse(ops):
    r = H.rng(20)
    V, E = 3000, 16
    var0 = r.normal(size=(V, E)).astype(np.float32)
    var, m, v = var0.copy(), np.zeros((V, E), np.float32), np.zeros((V, E), np.float32)
    dvar, dm, dv = dev(var0), torch.zeros((V, E), device="cuda"), torch.zeros((V, E), device="cuda")
    lvar, lm, lv = dev(var0), torch.zeros((V, E), device="cuda"), torch.zeros((V, E), device="cuda")
    rl = (var0.copy(), np.zeros((V, E), np.float32), np.zeros((V, E), np.float32))
    for t in range(1, 4):
        ids = r.integers(0, 500, size=800)
        vals = r.normal(size=(800, E)).astype(np.float32)
        var, m, v = L.adam_sparse_keras_step(var, m, v, ids, vals, t, lr=0.01, dt=np.float32)
        plan = ops.DedupPlan(dev(ids), V)
        rows = plan.segment_sum(dev(vals), E)
        ops.adam_sparse_keras(dvar, dm, dv, plan.uniq_ids, rows, plan.n_uniq, t, 0.01)
        assert np.abs(dvar.cpu().numpy() - var).max() <= 2e-5
        assert np.abs(dm.cpu().numpy() - m).max() <= 1e-6
        assert np.abs(dv.cpu().numpy() - v).max() <= 1e-6
        uid, g = L.dedup_indexed_slices(ids, vals, "sorted")
        rl = L.adam_rows_step(rl[0], rl[1], rl[2], uid, g, t, lr=0.01, dt=np.float32)
        ops.adam_rows(lvar, lm, lv, plan.uniq_ids, rows, plan.n_uniq, t, 0.01)
        assert np.abs(lvar.cpu().numpy() - rl[0]).max() <= 2e-5
    assert np.array_equal(dvar.cpu().numpy()[500:], var0[500:])     # never-touched rows do not move
    # dense
    g = r.normal(size=(V, E)).astype(np.float32)
    a = L.adam_dense_step(var, m, v, g, 4, lr=0.01, dt=np.float32)
    ops.adam_dense(dvar, dm, dv, dev(g), 4, 0.01)
    assert np.abs(dvar.cpu().numpy() - a[0]).max() <= 2e-5


def test_adam_sparse_keras_pair_equals_two_sweeps(ops):
    """One sweep over the fused [embed | w | pad] rows == the two per-table sweeps, bit for bit (var, m, v)."""
    from explicit_tf2_recommendation_amd._lib import lib, check
    import ctypes as C
    r = H.rng(21)
    V, E, ld = 5000, 16, 32

    def state():
        fused = torch.zeros((V, ld), device="cuda")
        fused[:, :E + 1] = dev(r0[:, :E + 1])
        return [fused] + [dev(x.copy()) for x in (me0, ve0, mw0, vw0)]

    r0 = r.normal(size=(V, ld)).astype(np.float32)
    me0, ve0 = r.normal(size=(V, E)).astype(np.float32) * 0.1, r.uniform(0, 0.1, size=(V, E)).astype(np.float32)
    mw0, vw0 = r.normal(size=(V, 1)).astype(np.float32) * 0.1, r.uniform(0, 0.1, size=(V, 1)).astype(np.float32)
    ids = r.integers(0, 700, size=900)
    plan = ops.DedupPlan(dev(ids), V)
    ge = plan.segment_sum(dev(r.normal(size=(900, E)).astype(np.float32)), E)
    gw = plan.segment_sum(dev(r.normal(size=(900, 1)).astype(np.float32)), 1)
    a = state()
    ops.adam_sparse_keras(a[0][:, :E], a[1], a[2], plan.uniq_ids, ge, plan.n_uniq, 3, 0.01)
    ops.adam_sparse_keras(a[0][:, E:E + 1], a[3], a[4], plan.uniq_ids, gw, plan.n_uniq, 3, 0.01)
    b = state()
    cap = ge.shape[0]
    side_e = torch.empty((cap, 3, E), device="cuda")
    side_w = torch.empty((cap, 3, 1), device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    check(lib.rec_adam_sparse_keras_pair_f32(p(b[0]), ld, p(b[1]), p(b[2]), p(b[3]), p(b[4]), V, E, p(plan.uniq_ids),
                                             p(ge), p(gw), p(plan.n_uniq), cap, p(side_e), p(side_w), 3, 0.01, 0.9,
                                             0.999, 1e-7, C.c_void_p(torch.cuda.current_stream().cuda_stream)),
          "rec_adam_sparse_keras_pair_f32")
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert torch.all(b[0][:, E + 1:] == 0)                            # the padding floats of a row are never written


# ---------------------------------------------------------------------------------------------
# sharding (bit exact)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("P", [1, 2, 4, 8])
@pytest.mark.parametrize("zipf", [None, 1.05])
def test_shard_bucketize_bit_exact(ops, P, zipf):
    V, n, E = 100003, 20000, 8
    r = H.rng(P)
    ids = np.minimum(r.zipf(zipf, size=n) - 1, V - 1) if zipf else r.integers(0, V, size=n)
    rows_per = -(-V // P)
    perm, counts, local = ops.shard_bucketize(dev(ids), rows_per, P)
    rp, rc, rl = L.shard_bucketize(ids, rows_per, P)
    assert np.array_equal(perm.cpu().numpy(), rp)
    assert np.array_equal(counts.cpu().numpy(), rc)
    assert np.array_equal(local.cpu().numpy(), rl)
    # simulated exchange: gather per shard, inverse permutation, compare with the unsharded lookup bitwise
    tab = r.normal(size=(V, E)).astype(np.float32)
    got = []
    start = 0
    for s in range(P):
        shard = dev(tab[s * rows_per:(s + 1) * rows_per])
        c = int(rc[s])
        got.append(ops.emb_gather(shard, local[start:start + c].contiguous()))
        start += c
    back = ops.permute_rows(torch.cat(got), perm, scatter=True).cpu().numpy()
    assert np.array_equal(back, tab[ids])


def test_shard_bucketize_many_tiles(ops):
    """> 1024 tiles of 1024 ids: the single-workgroup scan runs several passes with a carry."""
    V, n, P = 10_000_019, 1_300_000, 8
    ids = H.rng(77).integers(0, V, size=n)
    rows_per = -(-V // P)
    perm, counts, local = ops.shard_bucketize(dev(ids), rows_per, P)
    rp, rc, rl = L.shard_bucketize(ids, rows_per, P)
    assert np.array_equal(counts.cpu().numpy(), rc)
    assert np.array_equal(perm.cpu().numpy(), rp) and np.array_equal(local.cpu().numpy(), rl)


@pytest.mark.parametrize("P", [1, 2, 8, 64])
@pytest.mark.parametrize("overlap", ["heavy", "none", "ragged"])
def test_dedup_plan_sorted_lists(ops, P, overlap):
    """Owner-side union of P ascending duplicate-free lists (rank merge): ids bit exact, sums in list order."""
    r = H.rng(1000 + P)
    V, E = 50_000, 16
    lists = []
    for q in range(P):
        if overlap == "heavy":
            k = 3000
            pool = V // 10
        elif overlap == "none":
            k = 500
            pool = V
        else:
            k = [0, 1, 4000, 17][q % 4]
            pool = V // 4
        ids = np.sort(r.choice(pool, size=k, replace=False)).astype(np.int64)
        if overlap == "none":
            ids = ids // P * P + q if P <= 8 else ids          # disjoint residues for small P
            ids = np.unique(ids)
        lists.append(ids)
    counts = np.array([len(x) for x in lists], np.int64)
    ids = np.concatenate(lists) if counts.sum() else np.zeros(0, np.int64)
    n = ids.size
    if n == 0:
        pytest.skip("empty union")
    vals = r.normal(size=(n, E)).astype(np.float32)
    plan = ops.DedupPlan(dev(ids), V, list_counts=dev(counts))
    nu = int(plan.n_uniq.item())
    uniq_ref, sum_ref = L.dedup_indexed_slices(ids, vals.astype(np.float64), "sorted")
    assert nu == uniq_ref.size and np.array_equal(plan.uniq_ids.cpu().numpy()[:nu], uniq_ref)
    perm, seg = plan.perm.cpu().numpy(), plan.seg_start.cpu().numpy()
    assert sorted(perm.tolist()) == list(range(n)) and np.all(np.diff(ids[perm]) >= 0)
    for u in range(0, nu, max(1, nu // 200)):                       # members of a run in list (= position) order
        assert np.all(np.diff(perm[seg[u]:seg[u + 1]]) > 0)
    out = plan.segment_sum(dev(vals), E).cpu().numpy()
    assert np.abs(out[:nu] - sum_ref).max() <= 1e-5 * max(1, np.abs(sum_ref).max())
    assert np.all(out[nu:] == 0)
    # the radix-sort plan of the same ids gives the same unique list and the same sums bit for bit (same order)
    plan2 = ops.DedupPlan(dev(ids), V)
    assert np.array_equal(plan2.uniq_ids.cpu().numpy()[:nu], uniq_ref)
    assert np.array_equal(plan2.segment_sum(dev(vals), E).cpu().numpy(), out)


@pytest.mark.parametrize("P", [1, 2, 8])
@pytest.mark.parametrize("B,F,zipf", [(8192, 26, None), (1000, 26, 1.05), (33, 3, 1.2), (4096, 5, 1.05)])
def test_colsort_shard_map(ops, P, B, F, zipf):
    """De-duplicate-first exchange map over the per-column sort plan: compact indices, local ids and per-owner counts
    are integer work -- bit exact against numpy.unique."""
    import ctypes as C
    from explicit_tf2_recommendation_amd._lib import lib, check
    V = 1_000_003
    X = field_ids(5 + P, B, F, V, zipf)
    dims = [V // F] * F
    dims[-1] += V - sum(dims)
    off = np.concatenate([[0], np.cumsum(dims)[:-1]]).astype(np.int64)
    cols = [dev(X[:, f]) for f in range(F)]
    i32 = dict(dtype=torch.int32, device="cuda")
    i64 = dict(dtype=torch.int64, device="cuda")
    perm, col_uid = torch.empty((F, B), **i32), torch.empty((F, B), **i64)
    col_seg, col_nu = torch.empty((F, B + 1), **i32), torch.zeros(F, **i32)
    bad, oob = torch.zeros(1, **i32), torch.zeros(1, **i32)
    ws = torch.empty(lib.rec_colsort_workspace_bytes(B, F), dtype=torch.uint8, device="cuda")
    vp = lambda t: C.c_void_p(t.data_ptr())
    arr = (C.c_void_p * F)(*[c.data_ptr() for c in cols])
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(lib.rec_colsort_plan_i64(arr, F, B, V, vp(dev(off)), max(dims) - 1, vp(perm), vp(col_uid), vp(col_seg),
                                   vp(col_nu), vp(bad), vp(ws), st), "rec_colsort_plan_i64")
    rps = -(-V // P)
    uid_local, uidx = torch.full((B * F,), -1, **i64), torch.full((F, B), -1, **i64)
    counts, n_uniq = torch.full((P,), 99, **i64), torch.zeros(1, **i64)
    check(lib.rec_colsort_shard_map_i64(vp(perm), vp(col_uid), vp(col_seg), vp(col_nu), B, F, rps, P, vp(uid_local),
                                        vp(uidx), vp(counts), vp(n_uniq), vp(oob), st), "rec_colsort_shard_map_i64")
    uid, inv = np.unique(X, return_inverse=True)
    nu = int(n_uniq.item())
    assert nu == uid.size and bad.item() == 0 and oob.item() == 0
    owner = uid // rps
    assert np.array_equal(uid_local.cpu().numpy()[:nu], uid - owner * rps)
    assert np.array_equal(uidx.cpu().numpy(), inv.reshape(B, F).T)
    assert np.array_equal(counts.cpu().numpy(), np.bincount(owner, minlength=P))


@pytest.mark.parametrize("P", [1, 2, 8])
@pytest.mark.parametrize("B,F,zipf", [(8192, 26, None), (1000, 26, 1.05), (33, 3, 1.2), (4096, 5, 1.05)])
def test_colsort_shard_map_fixed_and_owner_side(ops, P, B, F, zipf):
    """Fixed-capacity exchange plan (constant split sizes): id message, slots of lookups and of unique ids -- integer
    work, bit exact against numpy.unique; then the owner side on a message assembled from P such plans: gather of the
    used slots only, union of the slabs (rank merge, device-side counts) and the sums over it."""
    import ctypes as C
    from explicit_tf2_recommendation_amd import engine
    from explicit_tf2_recommendation_amd._lib import lib, check
    V = 1_000_003
    dims = [V // F] * F
    dims[-1] += V - sum(dims)
    off = np.concatenate([[0], np.cumsum(dims)[:-1]]).astype(np.int64)
    rps = -(-V // P)
    cap = engine.exchange_capacity(dims, off, B, rps, P)
    i32 = dict(dtype=torch.int32, device="cuda")
    i64 = dict(dtype=torch.int64, device="cuda")
    vp = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def plan(seed):
        X = field_ids(seed, B, F, V, zipf)
        cols = [dev(X[:, f]) for f in range(F)]
        perm, col_uid = torch.empty((F, B), **i32), torch.empty((F, B), **i64)
        col_seg, col_nu = torch.empty((F, B + 1), **i32), torch.zeros(F, **i32)
        bad, oob = torch.zeros(1, **i32), torch.zeros(1, **i32)
        ws = torch.empty(lib.rec_colsort_workspace_bytes(B, F), dtype=torch.uint8, device="cuda")
        arr = (C.c_void_p * F)(*[c.data_ptr() for c in cols])
        check(lib.rec_colsort_plan_i64(arr, F, B, V, vp(dev(off)), max(dims) - 1, vp(perm), vp(col_uid), vp(col_seg),
                                       vp(col_nu), vp(bad), vp(ws), st), "rec_colsort_plan_i64")
        msg, uidx = torch.full((P, cap + 2), -7, **i64), torch.full((F, B), -1, **i64)
        slot_map, n_uniq = torch.full((B * F,), -1, **i32), torch.zeros(1, **i64)
        check(lib.rec_colsort_shard_map_fixed_i64(vp(perm), vp(col_uid), vp(col_seg), vp(col_nu), B, F, rps, P, cap,
                                                  vp(msg), vp(uidx), vp(slot_map), vp(n_uniq), vp(oob), st),
              "rec_colsort_shard_map_fixed_i64")
        uid, inv = np.unique(X, return_inverse=True)
        owner = uid // rps
        counts = np.bincount(owner, minlength=P)
        start = np.concatenate([[0], np.cumsum(counts)])
        slot = owner * cap + np.arange(uid.size) - start[owner]
        assert int(n_uniq.item()) == uid.size and bad.item() == 0 and oob.item() == 0 and counts.max() <= cap
        m = msg.cpu().numpy()
        assert np.array_equal(m[:, 0], counts) and np.all(m[:, 1] == 0)
        for o in range(P):
            assert np.array_equal(m[o, 2:2 + counts[o]], uid[owner == o] - o * rps)
        assert np.array_equal(uidx.cpu().numpy(), slot[inv.reshape(B, F)].T)
        assert np.array_equal(slot_map.cpu().numpy()[:uid.size], slot)
        return m

    # what owner 0 receives: its slab of the messages of P different batches (requester q = seed q)
    theirs = np.stack([plan(300 + q)[0] for q in range(P)])
    if P > 1:
        theirs[P - 1, 0] = 0                             # a requester with nothing for this owner
    counts = theirs[:, 0]
    r = H.rng(P * 7 + B)
    table = torch.from_numpy(r.normal(size=(rps, 32)).astype(np.float32)).cuda()
    out = torch.full((P * cap, 32), 123.0, device="cuda")
    oob = torch.zeros(1, **i32)
    mt = dev(theirs)
    check(lib.rec_emb_gather_lists_f32(vp(table), rps, 32, 32, vp(mt), P, cap, vp(out), vp(oob), st),
          "rec_emb_gather_lists_f32")
    o = out.cpu().numpy().reshape(P, cap, 32)
    for q in range(P):
        assert np.array_equal(o[q, :counts[q]], table.cpu().numpy()[theirs[q, 2:2 + counts[q]]])
        assert np.all(o[q, counts[q]:] == 123.0)         # unused slots are not touched
    assert oob.item() == 0
    out20 = torch.full((P * cap, 20), 123.0, device="cuda")          # only the leading 80 bytes of every row
    check(lib.rec_emb_gather_lists_f32(vp(table), rps, 20, 32, vp(mt), P, cap, vp(out20), vp(oob), st),
          "rec_emb_gather_lists_f32")
    o20 = out20.cpu().numpy().reshape(P, cap, 20)
    for q in range(P):
        assert np.array_equal(o20[q, :counts[q]], table.cpu().numpy()[theirs[q, 2:2 + counts[q]], :20])
        assert np.all(o20[q, counts[q]:] == 123.0)
    n = P * cap
    uniq, seg, perm = torch.empty(n, **i64), torch.empty(n + 1, **i32), torch.empty(n, **i32)
    nu_d = torch.zeros(1, **i64)
    wsb = lib.rec_dedup_workspace_bytes(n)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    check(lib.rec_dedup_plan_sorted_slabs_i64(vp(mt), P, cap, rps, vp(uniq), vp(seg), vp(perm), vp(nu_d), vp(ws), wsb,
                                              st), "rec_dedup_plan_sorted_slabs_i64")
    rows = r.normal(size=(n, 20)).astype(np.float32)
    ids_flat = np.concatenate([theirs[q, 2:2 + counts[q]] for q in range(P)])
    pos_flat = np.concatenate([q * cap + np.arange(counts[q]) for q in range(P)])
    uniq_ref, sum_ref = L.dedup_indexed_slices(ids_flat, rows[pos_flat].astype(np.float64), "sorted")
    nu = int(nu_d.item())
    assert nu == uniq_ref.size and np.array_equal(uniq.cpu().numpy()[:nu], uniq_ref)
    sg, pm = seg.cpu().numpy(), perm.cpu().numpy()
    assert np.all(sg[nu:] == ids_flat.size) and sorted(pm[:ids_flat.size].tolist()) == sorted(pos_flat.tolist())
    assert np.all(uniq.cpu().numpy()[nu:] == uniq_ref[0])
    sums = torch.empty((n, 20), device="cuda")
    sws = torch.empty(lib.rec_segment_sum_workspace_bytes(n, 20), dtype=torch.uint8, device="cuda")
    check(lib.rec_segment_sum_f32(vp(dev(rows)), 20, vp(perm), vp(seg), n, 1, vp(sums), vp(sws), st),
          "rec_segment_sum_f32")
    sums = sums.cpu().numpy()
    assert np.abs(sums[:nu] - sum_ref).max() <= 1e-5 * max(1, np.abs(sum_ref).max()) and np.all(sums[nu:] == 0)


def test_sorted_slabs_all_empty(ops):
    """An owner that receives no id at all: n_uniq = 0, every run empty."""
    import ctypes as C
    from explicit_tf2_recommendation_amd._lib import lib, check
    P, cap = 4, 64
    vp = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    msg = torch.zeros((P, cap + 2), dtype=torch.int64, device="cuda")
    n = P * cap
    uniq = torch.full((n,), -5, dtype=torch.int64, device="cuda")
    seg = torch.full((n + 1,), -5, dtype=torch.int32, device="cuda")
    perm = torch.empty(n, dtype=torch.int32, device="cuda")
    nu = torch.full((1,), 9, dtype=torch.int64, device="cuda")
    wsb = lib.rec_dedup_workspace_bytes(n)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    check(lib.rec_dedup_plan_sorted_slabs_i64(vp(msg), P, cap, 1000, vp(uniq), vp(seg), vp(perm), vp(nu), vp(ws), wsb, st),
          "rec_dedup_plan_sorted_slabs_i64")
    assert nu.item() == 0 and torch.all(seg == 0) and torch.all(uniq == 0)


# ---------------------------------------------------------------------------------------------
# strided tables / fused [embed | w | pad] rows
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("E", [4, 8, 16, 32])
@pytest.mark.parametrize("B,F", [(300, 26), (8192, 26), (33, 1), (100, 40)])
def test_fused_table_layout_matches_separate_tables(ops, E, B, F):
    V = 5000
    pr = H.deepfm_params(E + F, V, F, E, scale=0.4)
    X = field_ids(B + E, B, F, V, 1.1)
    ld = ops.fused_row_stride(E)
    assert ld >= E + 1 and ld & (ld - 1) == 0
    storage = torch.zeros((V, ld), device="cuda")
    storage[:, :E] = dev(pr["embed"])
    storage[:, E:E + 1] = dev(pr["w"])
    emb_v, w_v = storage[:, :E], storage[:, E:E + 1]
    z1, p1, r1, s1 = ops.emb_fm_fwd(emb_v, w_v, dev(pr["bias"]), dev(X), want_prob=True, want_rows=True)
    z2, p2, r2, s2 = ops.emb_fm_fwd(dev(pr["embed"]), dev(pr["w"]), dev(pr["bias"]), dev(X), want_prob=True,
                                    want_rows=True)
    p64, z64 = L.fm_forward(pr["embed"], pr["w"], pr["bias"], X, np.float64)
    for z in (z1, z2):
        assert np.abs(z.cpu().numpy() - z64[:, 0]).max() <= 1e-5 * max(1.0, np.abs(z64).max())
    assert torch.equal(r1, r2) and np.array_equal(r1.cpu().numpy(), pr["embed"][X])
    assert np.abs(s1.cpu().numpy() - s2.cpu().numpy()).max() <= 1e-6
    # strided gather and strided re-gather in the backward values
    assert np.array_equal(ops.emb_gather(emb_v, dev(X)).cpu().numpy(), pr["embed"][X])
    gz = dev(H.rng(1).normal(size=(B,)).astype(np.float32))
    v1 = ops.emb_fm_bwd_vals(emb_v, dev(X), gz, s1, None, None)
    v2 = ops.emb_fm_bwd_vals(dev(pr["embed"]), dev(X), gz, s1, r1, None)
    assert torch.equal(v1, v2)
