#!/usr/bin/env python3
"""tests/golden/retrieval_ckpt7_top20.npz: the retrieval step of the reference run on the reference's own artifacts.

Inputs (data files, read as JSON): 2.FM/retrieval_model/ebd_result/{user,item}_embedding.json -- the tower outputs the
reference saved (9 990 users, of which 54 distinct vectors; 5 040 items).  Computation: exactly what
2.FM/OfflineLoader.py:129-162 does -- items L2-normalised, sklearn.neighbors.BallTree(leaf_size=10), query(k=20) with
the raw user vector -- executed with scikit-learn in this container (the reference's dependency itself, not a
restatement).  Stored: the 54 distinct user vectors, all item vectors (raw, float64 as in the JSON), and BallTree's
(dist, ind) for k = 20.  Run once:  python scripts/make_golden_retrieval.py
"""
import json
import os

import numpy as np
from sklearn.neighbors import BallTree

REF = "/root/reference/2.FM/retrieval_model/ebd_result"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    users = json.load(open(os.path.join(REF, "user_embedding.json")))
    items = json.load(open(os.path.join(REF, "item_embedding.json")))
    item_ids = list(items.keys())
    item_vec = np.array([items[i] for i in item_ids], np.float64)
    uvec = np.unique(np.array(list(users.values()), np.float64), axis=0)
    tree = BallTree(np.array([v / np.linalg.norm(v) for v in item_vec]), leaf_size=10)
    dist, ind = tree.query(uvec, k=20)
    out = os.path.join(ROOT, "tests", "golden", "retrieval_ckpt7_top20.npz")
    np.savez_compressed(out, user_vec=uvec, item_vec=item_vec, dist=dist, ind=ind.astype(np.int64),
                        item_ids=np.array(item_ids))
    print("wrote", out, uvec.shape, item_vec.shape, dist.shape, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
