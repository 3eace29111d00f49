"""Retrieval after the DSSM towers -- the slice of 2.FM/OfflineLoader.py (generate_*_embedding :97-127, build_ball_tree
:129-146, load_to_redis :148-162) and 2.FM/OnlineServer.py (retrieve_online :53-75) that computes: tower outputs for
users and items, the L2-normalised item matrix, and the k nearest items of a user vector.  The BallTree, its pickle and
Redis are storage/serving plumbing of the reference and out of scope; BallTree.query is an exact search, so the same
neighbours come out of a brute-force scan on the GPU (csrc/retrieval.hip) with the item matrix resident in HBM.

    index = RetrievalIndex(item_ids, item_embeddings)          # build_ball_tree: normalises the item vectors
    items, dist = index.query(user_embeddings, fetch_num=20)   # ball_tree.query(user_emb, k=fetch_num)
"""
import torch

from . import ops


@torch.no_grad()
def tower_embeddings(tower, batches, id_name):
    """generate_user_embedding / generate_item_embedding (2.FM/OfflineLoader.py:97-127): run a tower over batches of
    feature dicts; returns (ids list, embeddings [n, final_dim] on the GPU).  ``id_name`` = 'user_id' | 'item_id'."""
    ids, outs = [], []
    for batch in batches:
        res = tower(batch)
        got = res[id_name] if id_name in res else batch.get(id_name)
        if got is not None:
            ids.extend(got.tolist() if hasattr(got, "tolist") else list(got))
        outs.append(res["output"])
    return ids, torch.cat(outs, 0)


class RetrievalIndex:
    """The item side of the two-tower retrieval: ids + L2-normalised vectors (OfflineLoader.py:138-141)."""

    def __init__(self, item_ids, item_embeddings):
        if not isinstance(item_embeddings, torch.Tensor) or not item_embeddings.is_cuda:
            raise RuntimeError("item_embeddings must be a tensor on the MI355X: the HIP path has no CPU fallback")
        self.item_list = list(item_ids)
        if len(self.item_list) != item_embeddings.shape[0]:
            raise ValueError("one id per item vector")
        self.items_hat = ops.l2_normalize_rows(item_embeddings.to(torch.float32).contiguous())

    def query(self, user_embeddings, fetch_num=20):
        """dist, ind = ball_tree.query(user_emb, k=fetch_num); items = [item_list[i] for i in ind]
        (OfflineLoader.py:157-160, OnlineServer.py:69-71).  Returns (items [nq][k], dist [nq,k], ind [nq,k])."""
        q = user_embeddings.to(torch.float32)
        if q.dim() == 1:
            q = q.unsqueeze(0)
        dist, ind = ops.topk_l2(q.contiguous(), self.items_hat, int(fetch_num))
        rows = ind.cpu().tolist()
        items = [[self.item_list[i] for i in row if i >= 0] for row in rows]
        return items, dist, ind
