"""Row-wise sharded embedding (explicit-tf2-recommendation_amd/sharded.py).

CPU part (runs in the build container, no GPU): a 2-rank gloo group exercises the real exchange logic -- id counts,
ids, rows back, gradients out -- with the oracle standing in for the device kernels (tests may do that; the product
never does).  GPU part: P logical shards on one device with the HIP kernels, bitwise against the unsharded lookup,
and the world_size-1 RCCL path end to end.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import layers_np as L
from tests import helpers as H


class OracleBackend:
    """CPU stand-in for sharded.HipBackend built on the oracle (TEST ONLY)."""

    @staticmethod
    def bucketize(ids, rows_per_shard, n_shard):
        perm, counts, local = L.shard_bucketize(ids.numpy(), rows_per_shard, n_shard)
        return torch.from_numpy(perm), torch.from_numpy(counts), torch.from_numpy(local), None

    @staticmethod
    def gather(table, ids):
        return torch.from_numpy(L.embedding_lookup(table.detach().numpy(), ids.numpy()).copy())

    @staticmethod
    def permute_rows(x, perm, scatter):
        out = torch.empty_like(x)
        if scatter:
            out[perm] = x
        else:
            out = x[perm].clone()
        return out

    @staticmethod
    def dedup_sum(ids, vals, V):
        uniq, rows = L.dedup_indexed_slices(ids.numpy(), vals.numpy(), "sorted")
        n = ids.numel()
        u = np.full(n, uniq[0], np.int64)
        u[: len(uniq)] = uniq
        r = np.zeros((n, vals.shape[1]), np.float32)
        r[: len(uniq)] = rows
        return torch.from_numpy(u), torch.from_numpy(r), torch.tensor([len(uniq)])

    # ---- the de-duplicate-first, fixed-capacity exchange (numpy restatement of the kernels' contracts)
    @staticmethod
    def plan(ids, V):
        a = ids.numpy().reshape(-1)
        n = a.size
        order = np.argsort(a, kind="stable")
        srt = a[order]
        head = np.ones(n, bool)
        head[1:] = srt[1:] != srt[:-1]
        uniq = srt[head]
        nu = len(uniq)
        seg = np.full(n + 1, n, np.int32)
        seg[:nu] = np.nonzero(head)[0]
        u = np.full(max(n, 1), uniq[0] if nu else 0, np.int64)
        u[:nu] = uniq

        class P_:
            pass
        pl = P_()
        pl.uniq_ids, pl.seg_start = torch.from_numpy(u), torch.from_numpy(seg)
        pl.perm, pl.n_uniq, pl.n = torch.from_numpy(order.astype(np.int32)), torch.tensor([nu]), n
        return pl

    @staticmethod
    def slab_map(plan, n, rows_per_shard, n_shard, cap, flag):
        nu = int(plan.n_uniq.item())
        uniq = plan.uniq_ids.numpy()[:nu]
        msg = np.zeros((n_shard, cap + 2), np.int64)
        first = np.searchsorted(uniq, np.arange(n_shard + 1) * rows_per_shard)
        first[-1] = nu
        uslot = np.zeros(nu, np.int64)
        for o in range(n_shard):
            c = first[o + 1] - first[o]
            assert c <= cap, "exchange capacity exceeded"
            msg[o, 0] = c
            msg[o, 2:2 + c] = uniq[first[o]:first[o + 1]] - o * rows_per_shard
            uslot[first[o]:first[o + 1]] = o * cap + np.arange(c)
        seg = plan.seg_start.numpy()[:nu]
        slot = np.empty(n, np.int64)
        pos_to_u = np.searchsorted(seg, np.arange(n), side="right") - 1
        slot[plan.perm.numpy()] = uslot[pos_to_u]
        us = np.full(n, n_shard * cap, np.int64)
        us[:nu] = uslot
        return torch.from_numpy(msg), torch.from_numpy(slot), torch.from_numpy(us)

    @staticmethod
    def gather_lists(table, msg, n_shard, cap, flag):
        t = table.detach().numpy()
        out = np.zeros((n_shard * cap, t.shape[1]), np.float32)
        m = msg.numpy()
        for q in range(n_shard):
            c = int(m[q, 0])
            out[q * cap:q * cap + c] = L.embedding_lookup(t, m[q, 2:2 + c])
        return torch.from_numpy(out)

    @staticmethod
    def take_rows(rows, slots, sink, xplan=None):
        return rows[slots]                                      # torch indexing: dense gradient of the rows buffer

    @staticmethod
    def owner_reduce(msg, g_rows, n_shard, cap, rows_per_shard):
        m_, g = msg.numpy(), g_rows.numpy()
        ids = np.concatenate([m_[q, 2:2 + int(m_[q, 0])] for q in range(n_shard)])
        vals = np.concatenate([g[q * cap:q * cap + int(m_[q, 0])] for q in range(n_shard)])
        tot = n_shard * cap
        u = np.zeros(tot, np.int64)
        r = np.zeros((tot, g.shape[1]), np.float32)
        nu = 0
        if len(ids):
            uniq, rows = L.dedup_indexed_slices(ids, vals, "sorted")
            nu = len(uniq)
            u[:] = uniq[0]
            u[:nu] = uniq
            r[:nu] = rows
        return torch.from_numpy(u), torch.from_numpy(r), torch.tensor([nu])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, V, E, result):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from explicit_tf2_recommendation_amd import sharded
        table = torch.from_numpy(H.rng(0).normal(size=(V, E)).astype(np.float32))
        # ranks with different batch sizes agree on the slab size through an explicit capacity (the default,
        # min(lookups, rows per shard), needs the same number of lookups on every rank)
        emb = sharded.ShardedEmbedding(V, E, backend=OracleBackend, capacity=100)
        emb.load_global_rows(table)
        r = H.rng(100 + rank)
        ids = torch.from_numpy(np.minimum(r.zipf(1.2, size=(37 + 5 * rank, 3)) - 1, V - 1).astype(np.int64))
        out = emb(ids)
        ok_fwd = torch.equal(out, table[ids])                              # bitwise
        g = torch.from_numpy(r.normal(size=tuple(out.shape)).astype(np.float32))
        (out * g).sum().backward()
        # reference: dense gradient of the FULL table summed over both ranks' batches
        dense_local = torch.zeros((V, E))
        dense_local.index_add_(0, ids.reshape(-1), g.reshape(-1, E))
        dist.all_reduce(dense_local)
        lo, hi = emb.row_range
        mine = emb.embeddings_shard.grad.to_dense()[: hi - lo]
        ok_bwd = torch.allclose(mine, dense_local[lo:hi], atol=1e-5)
        # C4: dense data-parallel gradients
        p = torch.nn.Parameter(torch.zeros(5))
        p.grad = torch.full((5,), float(rank + 1))
        sharded.allreduce_dense_grads([p])
        ok_dp = torch.equal(p.grad, torch.full((5,), 3.0))
        result[rank] = (bool(ok_fwd), bool(ok_bwd), bool(ok_dp))
    finally:
        dist.destroy_process_group()


def test_sharded_embedding_two_ranks_gloo():
    world, V, E = 2, 1001, 8
    mgr = mp.Manager()
    result = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), V, E, result), nprocs=world, join=True)
    assert dict(result) == {0: (True, True, True), 1: (True, True, True)}


def test_oracle_backend_matches_bucketize_contract():
    ids = torch.from_numpy(H.rng(3).integers(0, 100, size=50))
    perm, counts, local, _ = OracleBackend.bucketize(ids, 25, 4)
    assert counts.sum().item() == 50 and sorted(perm.tolist()) == list(range(50))
    owner = ids[perm] // 25
    assert torch.all(owner[1:] >= owner[:-1])
    assert torch.equal(local, ids[perm] - owner * 25)


class OracleStepBackend:
    """CPU stand-in for engine.HipStepBackend (TEST ONLY): the oracle plays the device kernels so that the 2-rank
    exchange logic of ShardedDeepFMStep (unique-first fixed-capacity slabs, 1/P scaling, flat all-reduce) runs under
    gloo."""

    def __init__(self, step, field_dims, field_offsets):
        self.step = step
        m = step.P * step.cap
        self.rows_local = torch.zeros((m, 20))
        self.rows_theirs = torch.zeros((m, 20))

    def begin(self):
        pass

    def fork(self):
        pass

    def side_context(self):
        import contextlib
        return contextlib.nullcontext()

    def join(self):
        pass

    def plan(self, cols, buf, on_side=False):
        st = self.step
        P, cap = st.P, st.cap
        X = np.concatenate([c.numpy().reshape(-1, 1) for c in cols], 1)                 # [B,F]
        uid, inv = np.unique(X, return_inverse=True)
        owner = uid // st.rows_per_shard
        counts = np.bincount(owner, minlength=P)
        assert counts.max() <= cap                       # exchange_capacity's bound
        start = np.concatenate([[0], np.cumsum(counts)])
        slot = owner * cap + (np.arange(len(uid)) - start[owner])
        msg = np.zeros((P, cap + 2), np.int64)
        msg[:, 0] = counts
        msg[owner, 2 + slot - owner * cap] = uid - owner * st.rows_per_shard
        return {"msg": torch.from_numpy(msg), "msg_theirs": torch.zeros((P, cap + 2), dtype=torch.int64),
                "uidx": torch.from_numpy(slot[inv.reshape(X.shape)].T.copy()),       # [F,B]
                "slot_map": torch.from_numpy(slot.astype(np.int32)), "n_uniq": torch.tensor([len(uid)])}

    def owner_plan(self, pl, buf, on_side=False):
        pass

    def gather(self, table, pl):
        st = self.step
        msg = pl["msg_theirs"]
        out = torch.zeros((st.P * st.cap, 20))
        for q in range(st.P):
            c = int(msg[q, 0])
            out[q * st.cap:q * st.cap + c] = table[msg[q, 2:2 + c], :20]
        return out

    def rows_step(self, pl, rows_local, y):
        from oracle import torch_ref as T
        st = self.step
        lay = st.layer
        rows = rows_local.double().requires_grad_()
        pr = {k: v.detach().double().requires_grad_() for k, v in lay.named_parameters()
              if k not in ("embed.embeddings", "w.embeddings")}
        p = {"embed": rows[:, :16], "w": rows[:, 16:17], "bias": pr["bias"],
             "k1": [pr["MLP_layer1.kernel_0"], pr["MLP_layer1.kernel_1"]],
             "b1": [pr["MLP_layer1.bias_0"], pr["MLP_layer1.bias_1"]],
             "k2": [pr["MLP_layer2.kernel_0"]], "b2": [pr["MLP_layer2.bias_0"]]}
        loss = T.keras_bce(y.double(), T.deepfm_forward(p, pl["uidx"].T.contiguous()))
        loss.backward()
        st.loss.copy_(loss.detach().float().reshape(1))
        for k in st.g:
            st.g[k].copy_(pr[k].grad.float().reshape(st.g[k].shape))
        self._row_grad = rows.grad                       # already summed per slot by autograd
        return None, None

    def local_grad(self, pl, vals, gz):
        return self._row_grad[:, :20].float().contiguous()         # [embed 16 | w | pad]: autograd leaves pad = 0

    def owner_reduce(self, pl, rows_theirs, scale):
        st = self.step
        msg = pl["msg_theirs"]
        ids, rows = [], []
        for q in range(st.P):                            # contract of the merge: every list ascending and unique
            c = int(msg[q, 0])
            part = msg[q, 2:2 + c].numpy()
            assert np.all(part[1:] > part[:-1])
            ids.append(msg[q, 2:2 + c])
            rows.append(rows_theirs[q * st.cap:q * st.cap + c])
        u, rows, nu = OracleBackend.dedup_sum(torch.cat(ids), torch.cat(rows), 0)
        rows = rows * scale
        return u, rows[:, :16], rows[:, 16:17], nu

    def check_flags(self):
        pass


def _step_worker(rank, world, port, V, B, names, result):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from explicit_tf2_recommendation_amd import engine, data, layers
        from oracle import torch_ref as T
        layers.set_init_seed(3)
        layer = layers.DeepFMRankingLayer(feature_names=names, feature_dims=V, embedding_dims=16)   # CPU parameters
        with torch.no_grad():
            layer.embed.embeddings.mul_(6.0)
        batches = [data.SyntheticGenerator(names, V, dist="zipf", seed=50 + r).batch(B) for r in range(world)]
        mine = {k: torch.from_numpy(v) for k, v in batches[rank].items()}
        gen0 = data.SyntheticGenerator(names, V, seed=0)
        step = engine.ShardedDeepFMStep(layer, B, gen0.dims, gen0.offsets, backend=OracleStepBackend)
        nxt = {k: torch.from_numpy(v) for k, v in data.SyntheticGenerator(names, V, seed=90 + rank).batch(B).items()}
        loss = step(mine, next_inputs=nxt).item()                  # announces a next batch: the pipelined branch
        # oracle on the concatenated global batch with the full table
        pr = {k: v.detach().double().requires_grad_() for k, v in layer.named_parameters()}
        p = {"embed": pr["embed.embeddings"], "w": pr["w.embeddings"], "bias": pr["bias"],
             "k1": [pr["MLP_layer1.kernel_0"], pr["MLP_layer1.kernel_1"]],
             "b1": [pr["MLP_layer1.bias_0"], pr["MLP_layer1.bias_1"]],
             "k2": [pr["MLP_layer2.kernel_0"]], "b2": [pr["MLP_layer2.bias_0"]]}
        Xg = np.concatenate([L.index_assemble(b, names) for b in batches])
        yg = np.concatenate([b["label"] for b in batches])
        ref = T.keras_bce(torch.from_numpy(yg).double(), T.deepfm_forward(p, torch.from_numpy(Xg)))
        ref.backward()
        ok = [abs(loss - ref.item()) <= 1e-6]
        for k, v in step.g.items():
            want = pr[k].grad.reshape(v.shape)
            ok.append(bool((v.double() - want).abs().max() <= 1e-6 + 1e-5 * want.abs().max()))
        lo, hi = step.row_range
        ids, rows_e, rows_w, nu = step.table_grad
        nu = int(nu.item())
        touched = np.unique(Xg)
        touched = touched[(touched >= lo) & (touched < hi)]
        ok.append(np.array_equal(ids.numpy()[:nu] + lo, touched))
        we, ww = pr["embed.embeddings"].grad[touched], pr["w.embeddings"].grad[touched]
        ok.append(bool((rows_e[:nu].double() - we).abs().max() <= 1e-6 + 1e-5 * we.abs().max()))
        ok.append(bool((rows_w[:nu].double() - ww).abs().max() <= 1e-6 + 1e-5 * ww.abs().max()))
        result[rank] = ok
    finally:
        dist.destroy_process_group()


def test_sharded_deepfm_step_two_ranks_gloo_matches_global_batch_oracle():
    """world_size 2: per-rank batches, row-sharded table; loss, dense gradients and each owner's row gradients must
    equal the oracle's on the concatenated batch of 2*B examples (mean loss over the global batch)."""
    world, V, B = 2, 3001, 48
    names = ["f%d" % i for i in range(6)]
    mgr = mp.Manager()
    result = mgr.dict()
    mp.spawn(_step_worker, args=(world, _free_port(), V, B, names, result), nprocs=world, join=True)
    assert all(all(v) for v in dict(result).values()) and len(result) == world, dict(result)


# ------------------------------------------------------------------------------------------------
# GPU
# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("P", [1, 2, 4, 8])
@pytest.mark.parametrize("dist_kind", ["uniform", "zipf"])
def test_logical_shards_bitwise_on_device(P, dist_kind):
    from explicit_tf2_recommendation_amd import sharded
    V, E = 100003, 64                                   # config D's row width
    r = H.rng(P)
    table = torch.from_numpy(r.normal(size=(V, E)).astype(np.float32)).cuda()
    n = (4096, 3)
    ids = r.integers(0, V, size=n) if dist_kind == "uniform" else np.minimum(r.zipf(1.05, size=n) - 1, V - 1)
    ids = torch.from_numpy(ids.astype(np.int64)).cuda()
    out = sharded.LocalShards(table, P).lookup(ids)
    assert torch.equal(out, table[ids])


@pytest.mark.gpu
def test_sharded_embedding_world1_rccl():
    """The real collective path (RCCL all-to-all) with a single rank: same kernels, same code as N > 1."""
    from explicit_tf2_recommendation_amd import sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        V, E = 5000, 16
        r = H.rng(9)
        table = torch.from_numpy(r.normal(size=(V, E)).astype(np.float32)).cuda()
        emb = sharded.ShardedEmbedding(V, E).cuda()
        emb.load_global_rows(table)
        ids = torch.from_numpy(np.minimum(r.zipf(1.1, size=(300, 4)) - 1, V - 1).astype(np.int64)).cuda()
        out = emb(ids)
        assert torch.equal(out, table[ids])
        g = torch.from_numpy(r.normal(size=(300, 4, E)).astype(np.float32)).cuda()
        (out * g).sum().backward()
        ref = torch.zeros((V, E), device="cuda", dtype=torch.float64)
        ref.index_add_(0, ids.reshape(-1), g.reshape(-1, E).double())
        got = emb.embeddings_shard.grad.to_dense()
        assert (got.double() - ref).abs().max().item() <= 1e-5
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_deepfm_step_world1_matches_fused_step():
    """The sharded train step (bucketize -> RCCL all-to-all -> owner gather -> fused kernel on the returned rows ->
    gradients back -> owner de-duplication) at world_size 1 against the single-GPU fused step on the same batch."""
    from explicit_tf2_recommendation_amd import engine, data, layers
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        B, F, V = 1024, 26, 200000
        names = ["f%d" % i for i in range(F)]
        layers.set_init_seed(5)
        layer = layers.DeepFMRankingLayer(feature_names=names, feature_dims=V, embedding_dims=16).cuda()
        with torch.no_grad():
            layer.embed.embeddings.mul_(2.0)
        gen = data.SyntheticGenerator(names, V, dist="zipf", seed=5)
        batch = data.to_device(gen.batch(B))
        ref = engine.DeepFMFusedStep(layer, B, gen.dims, gen.offsets, use_graph=False)
        sh = engine.ShardedDeepFMStep(layer, B, gen.dims, gen.offsets)
        batch2 = data.to_device(gen.batch(B))
        l_ref = ref(batch).item()
        l_sh = sh(batch, next_inputs=batch2).item()
        g_sh = {k: v.clone() for k, v in sh.g.items()}
        tg = tuple(t.clone() for t in sh.table_grad)
        # second step consumes the plan that was built beside the first one
        assert abs(sh(batch2).item() - ref(batch2).item()) <= 1e-6
        g_ref2 = ref.gradients()
        for k, v in sh.g.items():
            assert torch.equal(v, g_ref2[k]), k
        sh.check_flags()
        l_ref = ref(batch).item()
        assert abs(l_ref - l_sh) <= 1e-6
        g_ref = ref.gradients()
        for k, v in g_sh.items():
            assert torch.equal(v, g_ref[k]), k                       # same kernel on the same rows: bitwise
        ids, rows_e, rows_w, nu = tg
        nu = int(nu.item())
        rid, re_, rnu = g_ref["embed.embeddings"]
        assert nu == int(rnu.item()) and torch.equal(ids[:nu], rid[:nu])      # local id = global id at world 1
        assert (rows_e[:nu] - re_[:nu]).abs().max().item() <= 1e-6
        assert (rows_w[:nu] - g_ref["w.embeddings"][1][:nu]).abs().max().item() <= 1e-6
        # a loader that refills a batch dict with NEW tensors: the cached column list must not be used (the old ids would
        # be trained on silently); wrong label / column types are refused before any pointer reaches a kernel
        batch3 = data.to_device(gen.batch(B))
        for k in names:
            batch[k] = batch3[k]
        batch["label"] = batch3["label"]
        assert abs(sh(batch).item() - ref(batch3).item()) <= 1e-6
        g_ref3 = ref.gradients()
        for k, v in sh.g.items():
            assert torch.equal(v, g_ref3[k]), k
        for wrong in (batch3["label"].double(), batch3["label"].cpu(), batch3["label"][: B // 2]):
            with pytest.raises(ValueError):
                sh(dict(batch3, label=wrong))
        with pytest.raises(ValueError):
            sh(dict(batch3, **{names[0]: batch3[names[0]].cpu()}))
        # many(): the graph key covers every column, so two cycles that differ in one inner column are two graphs
        other = dict(batch2)
        other[names[7]] = batch3[names[7]]
        la = sh.many([batch3, batch2]).item()
        lb = sh.many([batch3, other]).item()
        assert len(sh._graphs) == 2
        assert abs(sh.many([batch3, batch2]).item() - la) == 0 and la != lb
        sh.release_graphs()
    finally:
        dist.destroy_process_group()


def _gpu_step_worker(rank, world, port, V, B, F, result):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from explicit_tf2_recommendation_amd import engine, data, layers, sharded

        HostStagedComm = sharded.HostStagedComm

        torch.cuda.set_device(0)
        names = ["f%d" % i for i in range(F)]
        layers.set_init_seed(11)
        layer = layers.DeepFMRankingLayer(feature_names=names, feature_dims=V, embedding_dims=16).cuda()
        with torch.no_grad():
            layer.embed.embeddings.mul_(2.0)
        gens = [data.SyntheticGenerator(names, V, dist="zipf", seed=70 + r) for r in range(world)]
        batches = [g.batch(B) for g in gens]
        step = engine.ShardedDeepFMStep(layer, B, gens[0].dims, gens[0].offsets, comm=HostStagedComm())
        warm = data.to_device(gens[rank].batch(B))
        step(warm, next_inputs=data.to_device(batches[rank]))        # the measured step runs on a pipelined plan
        loss = step(data.to_device(batches[rank])).item()
        # single-GPU fused step on the concatenated global batch (same device, full table)
        cat = {k: np.concatenate([b[k] for b in batches]) for k in batches[0]}
        ref = engine.DeepFMFusedStep(layer, world * B, gens[0].dims, gens[0].offsets, use_graph=False)
        l_ref = ref(data.to_device(cat)).item()
        g_ref = ref.gradients()
        ok = [abs(loss - l_ref) <= 2e-6]
        for k, v in step.g.items():
            ok.append(bool((v - g_ref[k]).abs().max() <= 1e-7 + 2e-5 * g_ref[k].abs().max()))
        lo, hi = step.row_range
        ids, rows_e, rows_w, nu = step.table_grad
        nu = int(nu.item())
        rid, re_, rnu = g_ref["embed.embeddings"]
        rnu = int(rnu.item())
        rid, re_, rw_ = rid[:rnu], re_[:rnu], g_ref["w.embeddings"][1][:rnu]
        m = (rid >= lo) & (rid < hi)
        ok.append(bool(nu == int(m.sum().item()) and torch.equal(ids[:nu] + lo, rid[m])))
        ok.append(bool((rows_e[:nu] - re_[m]).abs().max() <= 1e-7 + 2e-5 * re_.abs().max()))
        ok.append(bool((rows_w[:nu] - rw_[m]).abs().max() <= 1e-7 + 2e-5 * rw_.abs().max()))
        ok.append(int(step.oob.item()) == 0)
        result[rank] = ok
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_deepfm_step_two_ranks_hip_kernels_match_single_gpu_global_batch():
    """Two ranks (two processes on the one GPU, gloo through host memory) with the HIP kernels: loss, dense
    gradients and the owners' row gradients against the single-GPU fused step on the concatenated batch."""
    world, V, B, F = 2, 200001, 512, 26
    mgr = mp.Manager()
    result = mgr.dict()
    mp.spawn(_gpu_step_worker, args=(world, _free_port(), V, B, F, result), nprocs=world, join=True)
    assert len(result) == world and all(all(v) for v in dict(result).values()), dict(result)


# ---------------------------------------------------------------------------------------------------
# Row-sharded tables behind the layers of BASELINE configs D (DSSM two-tower) and E (DIN): layers.*(sharded=True)
# ---------------------------------------------------------------------------------------------------
def _family(family, V, sharded, comm=None, E=8, T=9):
    from explicit_tf2_recommendation_amd import data, layers
    kw = dict(sharded=sharded, comm=comm) if sharded else {}
    if family == "dssm":
        un, inn = ["user_tag1", "user_tag2"], ["item_tag1", "item_tag2", "item_tag3"]
        layer = layers.DSSMTwoTowerRetrievalLayer(u_feature_names=un, i_feature_names=inn, u_feature_dims=V,
                                                  i_feature_dims=V, u_embedding_dims=E, i_embedding_dims=E, **kw)
        mk = lambda seed: data.SyntheticGenerator(un + inn, V, dist="zipf", seed=seed)
    elif family == "dcn":
        cat, cont = ["c%d" % i for i in range(6)], ["x0", "x1"]
        layer = layers.DeepCrossNetworkLayer(categorical_features=cat, continuous_features=cont, feature_dims=V,
                                             embedding_dims=E, layer_num=2, type="matrix", **kw)
        mk = lambda seed: data.SyntheticGenerator(cat, V, continuous=cont, dist="zipf", seed=seed)
    else:
        user, item = ["uid", "utag1"], ["i_goods_id", "i_shop_id", "i_cate_id"]
        ser = ["visited_goods_ids", "visited_shop_ids", "visited_cate_ids"]
        layer = layers.DINLayer(user_and_context_categorical_features=user, item_categorical_features=item,
                                behavior_series_features=ser, feature_dims=V, embedding_dims=E,
                                mask_mode="reference" if family == "din" else "valid", **kw)
        mk = lambda seed: data.SyntheticGenerator(user + item, V, series=ser, seq_len=T, seed=seed)
    return layer.cuda(), mk


def _tables(layer):
    """(name, module) of every embedding table of a layer"""
    return [(n, m) for n, m in layer.named_modules() if hasattr(m, "embeddings") or hasattr(m, "embeddings_shard")]


def _copy_into_sharded(ref, sh):
    """Parameters of the unsharded layer -> the sharded twin (tables: this rank's block of rows)."""
    rt, st = dict(_tables(ref)), dict(_tables(sh))
    with torch.no_grad():
        for n, m in st.items():
            if hasattr(m, "embeddings_shard"):
                m.load_global_rows(rt[n].embeddings.detach())
        rp = dict(ref.named_parameters())
        for n, p in sh.named_parameters():
            if not n.endswith("embeddings_shard"):
                p.copy_(rp[n])


def _loss_and_grads(layer, batch):
    from explicit_tf2_recommendation_amd import functional as Fn
    for p in layer.parameters():
        p.grad = None
    out = layer({k: v for k, v in batch.items() if k != "label"})["output"]
    y = batch["label"]
    if out.dim() == 2 and out.shape[1] > 1:
        y = y.expand(-1, out.shape[1]).contiguous()
    loss = Fn.KerasBCE.apply(out, y)
    loss.backward()
    return out.detach().clone(), loss.item(), {n: (p.grad.to_dense() if p.grad.is_sparse else p.grad).clone()
                                               for n, p in layer.named_parameters()}


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["dssm", "dcn", "din", "din_valid"])
def test_sharded_layers_world1_match_unsharded_and_replay_from_a_graph(family):
    """layers.*(sharded=True) at world size 1 (the exchange code of N > 1, collectives skipped for the rank's own slab)
    against the unsharded layer with the same parameters: outputs, loss and every gradient; then the same step captured by
    engine.GraphedTrainStep -- the fixed-capacity exchange reads nothing back, so forward + backward replay from ONE
    hipGraph bit-identically to the eager run."""
    from explicit_tf2_recommendation_amd import data, engine, layers
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        B, V = 160, 3000
        layers.set_init_seed(7)
        ref, mk = _family(family, V, False)
        sh, _ = _family(family, V, True)
        _copy_into_sharded(ref, sh)
        gen = mk(3)
        batches = [data.to_device(gen.batch(B)) for _ in range(3)]
        for b in batches[:2]:
            o_ref, l_ref, g_ref = _loss_and_grads(ref, b)
            o_sh, l_sh, g_sh = _loss_and_grads(sh, b)
            assert torch.equal(o_ref, o_sh) and l_ref == l_sh           # same rows into the same kernels: bitwise
            for n, g in g_sh.items():
                want = g_ref[n.replace("embeddings_shard", "embeddings")]
                if n.endswith("embeddings_shard"):
                    g = g[: want.shape[0]]
                assert (g - want).abs().max().item() <= 1e-6 * max(1.0, want.abs().max().item()), n
        for _, m in _tables(sh):
            if hasattr(m, "check_flags"):
                m.check_flags()
        step = engine.GraphedTrainStep(sh, batches[0])
        for b in (batches[2], batches[1], batches[2]):
            loss = step(b).item()
            _, want_loss, want = _loss_and_grads(ref, b)
            assert abs(loss - want_loss) <= 1e-6 * max(1.0, abs(want_loss))
            for n, p in sh.named_parameters():
                g = p.grad.to_dense() if p.grad.is_sparse else p.grad
                w = want[n.replace("embeddings_shard", "embeddings")]
                if n.endswith("embeddings_shard"):
                    g = g[: w.shape[0]]
                assert (g - w).abs().max().item() <= 1e-6 * max(1.0, w.abs().max().item()), n
    finally:
        dist.destroy_process_group()


def _sharded_layer_worker(rank, world, port, family, V, B, result):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from explicit_tf2_recommendation_amd import data, layers, sharded
        torch.cuda.set_device(0)
        layers.set_init_seed(13)
        ref, mk = _family(family, V, False)                 # every rank builds the same full layer (same seed)
        layers.set_init_seed(13)
        sh, _ = _family(family, V, True, comm=sharded.HostStagedComm(None, True))
        _copy_into_sharded(ref, sh)
        mine = data.to_device(mk(50 + rank).batch(B))
        o_ref, l_ref, g_ref = _loss_and_grads(ref, mine)
        o_sh, l_sh, g_sh = _loss_and_grads(sh, mine)
        ok = [bool(torch.equal(o_ref, o_sh)), l_ref == l_sh]
        # table gradients: an owner receives BOTH ranks' contributions -> compare with the sum of the unsharded dense
        # gradients of both ranks' batches, restricted to this rank's block of rows
        for n, g in g_sh.items():
            want = g_ref[n.replace("embeddings_shard", "embeddings")]
            if n.endswith("embeddings_shard"):
                tot = want.cpu().clone()
                dist.all_reduce(tot)
                mod = dict(_tables(sh))[n.rsplit(".", 1)[0]]
                lo, hi = mod.row_range
                ok.append(bool((g[: hi - lo].cpu() - tot[lo:hi]).abs().max() <= 1e-6 * max(1.0, tot.abs().max().item())))
            else:
                ok.append(bool((g - want).abs().max() <= 1e-6 * max(1.0, want.abs().max().item())))
        for _, m in _tables(sh):
            if hasattr(m, "check_flags"):
                m.check_flags()
        result[rank] = ok
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["dssm", "din"])
def test_sharded_layers_two_ranks_hip_kernels(family):
    """Two ranks (two processes on the one GPU, exchanges through host memory: RCCL refuses two ranks on one device) with
    the HIP kernels: each rank's outputs against the unsharded layer on its own batch, and every owner's table gradient
    against the sum of both ranks' unsharded gradients."""
    world, V, B = 2, 3001, 96
    mgr = mp.Manager()
    result = mgr.dict()
    mp.spawn(_sharded_layer_worker, args=(world, _free_port(), family, V, B, result), nprocs=world, join=True)
    assert len(result) == world and all(all(v) for v in dict(result).values()), dict(result)


@pytest.mark.gpu
@pytest.mark.parametrize("P,cap,dist_kind", [(1, 700, "uniform"), (2, 900, "zipf"), (8, 120, "uniform"), (4, 2000, "zipf")])
def test_slab_map_kernel_equals_its_numpy_restatement(P, cap, dist_kind):
    """rec_shard_slab_map_uslot_i64 (message slabs, slot of every lookup, slot of every unique id) against
    OracleBackend.slab_map on the same plan -- bit exact, incl. the tail of `uslot` (one row past the buffer)."""
    from explicit_tf2_recommendation_amd import sharded
    r = H.rng(7 + P)
    V = 6000
    rps = -(-V // P)
    n = 900
    a = r.integers(0, V, size=n) if dist_kind == "uniform" else np.minimum(r.zipf(1.2, size=n) - 1, V - 1)
    # every owner within capacity: clip the id set if needed (the overflow flag is covered by the DeepFM step's tests)
    ids = torch.from_numpy(a.astype(np.int64))
    ref_plan = OracleBackend.plan(ids, V)
    nu = int(ref_plan.n_uniq.item())
    uniq = ref_plan.uniq_ids.numpy()[:nu]
    per_owner = np.bincount(uniq // rps, minlength=P)
    if per_owner.max() > cap:
        pytest.skip("id sample exceeds the capacity of this case")
    msg0, slot0, us0 = OracleBackend.slab_map(ref_plan, n, rps, P, cap, None)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    plan = sharded.HipBackend.plan(ids.cuda(), V)
    msg, slot, us = sharded.HipBackend.slab_map(plan, n, rps, P, cap, flag)
    assert int(flag.item()) == 0
    assert torch.equal(slot.cpu(), slot0) and torch.equal(us.cpu(), us0)
    m, m0 = msg.cpu().numpy(), msg0.numpy()
    for o in range(P):
        c = int(m0[o, 0])
        assert m[o, 0] == c and np.array_equal(m[o, 2:2 + c], m0[o, 2:2 + c])
    # the plan of the ids is the plan of the slots: slot[perm[seg_start[u]]] == uslot[u]
    perm, seg = plan.perm.cpu().numpy(), plan.seg_start.cpu().numpy()
    assert np.array_equal(slot0.numpy()[perm[seg[:nu]]], us0.numpy()[:nu])
