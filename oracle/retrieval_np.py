"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the retrieval step that follows the DSSM towers
in the reference -- SURVEY.md section 8(f3).

    2.FM/OfflineLoader.py:129-146   build_ball_tree: item vectors are L2-normalised (info[1]/np.linalg.norm(info[1])) and
                                    put into sklearn.neighbors.BallTree (Euclidean metric)
    2.FM/OfflineLoader.py:148-162   load_to_redis:   dist, ind = ball_tree.query([user_emb], k=fetch_num) for every user,
                                    with the RAW (not normalised) user vector; results ascending by distance
    2.FM/OnlineServer.py:53-75      retrieve_online: the same query for one user at serving time

BallTree.query is an exact k-nearest-neighbour search, so its result is the brute-force one: the k smallest
||u - i_hat||_2, ascending.  Third-party dependency: scikit-learn (version unpinned in the reference; any version's exact
BallTree gives the same neighbours up to ties).  Pinned by tests/golden/retrieval_ckpt7_top20.npz, which
scripts/make_golden_retrieval.py produced by running sklearn's BallTree -- the reference's own call -- on the reference's
own ebd_result/{user,item}_embedding.json.
"""
import numpy as np


def normalize_items(items, dt=np.float64):
    """info[1] / np.linalg.norm(info[1]) per item (OfflineLoader.py:140)."""
    items = np.asarray(items, dt)
    return items / np.linalg.norm(items, axis=1, keepdims=True)


def topk_l2(queries, items_hat, k, dt=np.float64):
    """k nearest rows of items_hat for every query (Euclidean), ascending; ties: the lower index first.
    Returns (dist [nq,k], ind [nq,k] int64)."""
    q = np.asarray(queries, dt)
    x = np.asarray(items_hat, dt)
    nq, n = q.shape[0], x.shape[0]
    k = min(k, n)
    dist = np.empty((nq, k), dt)
    ind = np.empty((nq, k), np.int64)
    for i in range(nq):
        d2 = np.square(x - q[i]).sum(axis=1)
        order = np.lexsort((np.arange(n), d2))[:k]           # by distance, then by index
        ind[i] = order
        dist[i] = np.sqrt(d2[order])
    return dist, ind
