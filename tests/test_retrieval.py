"""Retrieval after the DSSM towers (2.FM/OfflineLoader.py:129-162, 2.FM/OnlineServer.py:53-75).

tests/golden/retrieval_ckpt7_top20.npz holds what the reference's own call -- sklearn BallTree over the L2-normalised
item vectors, query(k=20) with the raw user vector -- returns on the reference's own ebd_result/*.json
(scripts/make_golden_retrieval.py).  CPU: the brute-force restatement reproduces it exactly.  GPU: the HIP scan
reproduces it (indices exact wherever neighbours are further apart than fp32 can confuse; distances to 1e-6 relative)
and, on synthetic data up to 4M items, the oracle and the domain's size-independent properties."""
import os

import numpy as np
import pytest
import torch

from oracle import retrieval_np as R
from tests import helpers as H

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "retrieval_ckpt7_top20.npz")


def test_oracle_reproduces_the_reference_balltree_results():
    g = np.load(GOLD)
    items_hat = R.normalize_items(g["item_vec"])
    assert np.allclose(np.linalg.norm(items_hat, axis=1), 1.0, atol=1e-12)
    dist, ind = R.topk_l2(g["user_vec"], items_hat, 20)
    assert np.array_equal(ind, g["ind"])                       # the same 20 items in the same order, all 54 users
    assert np.abs(dist - g["dist"]).max() <= 1e-12
    assert np.all(np.diff(dist, axis=1) >= 0)


def test_oracle_tie_break_and_short_tables():
    items = np.array([[1.0, 0.0], [0.0, 1.0], [1.0, 0.0], [-1.0, 0.0]])
    dist, ind = R.topk_l2(np.array([[1.0, 0.0]]), items, 3)
    assert ind.tolist() == [[0, 2, 1]] and dist[0, 0] == 0 and dist[0, 1] == 0     # equal distances: lower index first
    dist, ind = R.topk_l2(np.array([[0.0, 0.0]]), items[:2], 5)
    assert ind.shape == (1, 2)


def _check(dist, ind, q, items_hat, k, tol=2e-6):
    """HIP result against the float64 brute force: distances everywhere; indices wherever the neighbour gap is clear."""
    rd, ri = R.topk_l2(q, items_hat, k)
    dist, ind = dist.cpu().numpy().astype(np.float64), ind.cpu().numpy()
    scale = max(1.0, np.abs(rd).max())
    assert np.abs(dist - rd).max() <= tol * scale
    assert np.all(np.diff(dist, axis=1) >= 0)
    same = ind == ri
    if not same.all():                                          # only fp32-indistinguishable neighbours may swap
        bad_q, bad_j = np.nonzero(~same)
        for qq, jj in zip(bad_q, bad_j):
            d_true = np.sqrt(np.square(items_hat[ind[qq, jj]] - q[qq]).sum())
            assert abs(d_true - rd[qq, jj]) <= tol * scale, (qq, jj)
    for row in ind:
        assert len(set(row.tolist())) == len(row)               # no item twice
    return same.mean()


@pytest.mark.gpu
def test_gpu_topk_reproduces_the_reference_balltree_results():
    from explicit_tf2_recommendation_amd import ops, retrieval
    g = np.load(GOLD)
    item = torch.from_numpy(g["item_vec"].astype(np.float32)).cuda()
    user = torch.from_numpy(g["user_vec"].astype(np.float32)).cuda()
    index = retrieval.RetrievalIndex([str(s) for s in g["item_ids"]], item)
    hat = index.items_hat.cpu().numpy().astype(np.float64)
    assert np.abs(hat - R.normalize_items(g["item_vec"])).max() <= 2e-7
    items, dist, ind = index.query(user, fetch_num=20)
    d, i = dist.cpu().numpy(), ind.cpu().numpy()
    assert np.abs(d - g["dist"]).max() <= 2e-6 * g["dist"].max()
    d21, _ = R.topk_l2(g["user_vec"], R.normalize_items(g["item_vec"]), 21)      # the 21st neighbour bounds the last place
    gaps = np.diff(d21, axis=1)                                 # [54, 20]: gap after every one of the 20 places
    clear = gaps > 1e-5                                         # places whose neighbours are > 1e-5 away in distance
    clear[:, 1:] &= gaps[:, :-1] > 1e-5
    assert np.array_equal(i[clear], g["ind"][clear])            # exact wherever fp32 can tell the neighbours apart
    assert (i == g["ind"]).mean() >= 0.97                       # (the reference data has neighbours 7e-8 apart)
    _check(dist, ind, g["user_vec"], R.normalize_items(g["item_vec"]), 20)
    assert items[0][0] == str(g["item_ids"][i[0, 0]])
    # one user at a time (retrieve_online) gives the same row
    it1, d1, i1 = index.query(user[7], fetch_num=20)
    assert torch.equal(i1[0], ind[7]) and (d1[0] - dist[7]).abs().max().item() <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("n,d,nq,k", [(1, 8, 3, 1), (5, 3, 2, 5), (300, 8, 257, 20), (70000, 8, 100, 20), (5000, 64, 33, 64),
                                       (100000, 16, 600, 33), (1000, 33, 10, 16)])
def test_gpu_topk_matches_bruteforce(n, d, nq, k):
    from explicit_tf2_recommendation_amd import ops
    r = H.rng(n + d)
    items = R.normalize_items(r.normal(size=(n, d)))
    q = r.normal(size=(nq, d)) * 0.7
    kk = min(k, n)
    dist, ind = ops.topk_l2(torch.from_numpy(q.astype(np.float32)).cuda(),
                            torch.from_numpy(items.astype(np.float32)).cuda(), kk)
    _check(dist, ind, q.astype(np.float32).astype(np.float64), items.astype(np.float32).astype(np.float64), kk)


@pytest.mark.gpu
def test_gpu_topk_ties_padding_and_strides():
    from explicit_tf2_recommendation_amd import ops
    items = torch.tensor([[1.0, 0.0], [0.0, 1.0], [1.0, 0.0], [-1.0, 0.0]]).cuda()
    dist, ind = ops.topk_l2(torch.tensor([[1.0, 0.0]]).cuda(), items, 3)
    assert ind.cpu().tolist() == [[0, 2, 1]]                    # equal distances: the lower index first
    dist, ind = ops.topk_l2(torch.tensor([[0.0, 0.0]]).cuda(), items[:2], 4)      # fewer items than k
    assert ind.cpu().tolist()[0][2:] == [-1, -1] and torch.isinf(dist[0, 2:]).all()
    # strided views (a column block of a wider buffer)
    big = torch.randn(1000, 40).cuda()
    a, i1 = ops.topk_l2(big[:50, 8:16], big[:, 24:32], 7)
    b, i2 = ops.topk_l2(big[:50, 8:16].contiguous(), big[:, 24:32].contiguous(), 7)
    assert torch.equal(i1, i2) and (a - b).abs().max().item() <= 1e-6
    with pytest.raises(NotImplementedError):
        ops.topk_l2(big[:5, :8], big[:, :8], 65)


@pytest.mark.gpu
def test_gpu_topk_full_size_properties():
    """4M items x 8d (config-D-like item count scaled to a test), 1024 queries: sortedness, the k-th distance bounds
    every other item of a sampled slice, and the results do not depend on how the scan was split."""
    from explicit_tf2_recommendation_amd import ops
    n, d, nq, k = 4_000_000, 8, 1024, 20
    g = torch.Generator(device="cuda").manual_seed(3)
    items = ops.l2_normalize_rows(torch.randn((n, d), device="cuda", generator=g))
    q = torch.randn((nq, d), device="cuda", generator=g) * 0.5
    dist, ind = ops.topk_l2(q, items, k)
    assert (dist[:, 1:] >= dist[:, :-1]).all() and int(ind.min()) >= 0 and int(ind.max()) < n
    true = (items[ind.reshape(-1)].reshape(nq, k, d) - q[:, None, :]).norm(dim=2)
    assert (true - dist).abs().max().item() <= 2e-6
    sl = items[1_000_000:1_200_000]
    dmin = torch.cdist(q[:64], sl).topk(k, dim=1, largest=False).values            # torch as an independent check
    assert (dist[:64, k - 1:k] <= dmin[:, k - 1:k] + 1e-6).all()
    d2, i2 = ops.topk_l2(q[:300], items, k)                     # another query blocking -> another slab split
    assert (d2 - dist[:300]).abs().max().item() <= 1e-5 and (i2 == ind[:300]).float().mean().item() >= 0.999
