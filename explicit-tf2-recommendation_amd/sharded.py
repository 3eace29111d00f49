"""Row-wise sharded embedding tables across the GPUs of a node.

The reference is single-device; this is the build's own model-parallel step for tables that do not fit (or should
not be replicated on) one GPU: 100M x 64d = 25.6 GB per tower at BASELINE config D.  One process per GPU;
``torch.distributed`` backend "nccl" is RCCL on ROCm, and on an MI355X node every GPU pair has its own xGMI link,
so all-to-all is the natural collective (one hop, all 7 links busy).

    block partition:  rows_per_shard = ceil(V / P),  owner = id // rows_per_shard,  local = id - owner*rows_per_shard
    forward   C1  all-to-all of per-owner id counts, then of the ids themselves (int64)
              --  local gather on the owner (HIP gather kernel)
              C2  all-to-all of the gathered rows back to the requesters, inverse permutation
    backward  C3  all-to-all of the row gradients to the owners, who de-duplicate and segment-sum them

The integer side (bucketize / permutation / counts) is bit exact and independent of P; tests/test_sharded.py holds
``lookup == table[ids]`` bitwise for P in {1,2,4,8} logical shards on one device and for a 2-rank gloo group.
"""
import torch
import torch.distributed as dist

from . import ops


class HipBackend:
    """The device-side pieces, all HIP kernels (ops.py).  Tests on CPU inject an oracle-backed stand-in."""

    @staticmethod
    def bucketize(ids, rows_per_shard, n_shard):
        flag = ops.new_flag(ids.device)
        perm, counts, local = ops.shard_bucketize(ids, rows_per_shard, n_shard, flag)
        return perm, counts, local, flag

    @staticmethod
    def gather(table, ids):
        return ops.emb_gather(table, ids)

    @staticmethod
    def permute_rows(x, perm, scatter):
        return ops.permute_rows(x, perm, scatter)

    @staticmethod
    def dedup_sum(ids, vals, V):
        """(uniq_ids [n], rows [n,E], n_uniq [1]) with the padded-tail convention of ops.DedupPlan."""
        plan = ops.DedupPlan(ids, V)
        return plan.uniq_ids, plan.segment_sum(vals, vals.shape[1]), plan.n_uniq


class DistComm:
    """all-to-all over a torch.distributed process group (RCCL on the GPUs, gloo in the CPU tests).

    ``separate_count_channel=True`` creates a second communicator over the same ranks for the tiny split-size exchange
    (C0): it then has its own RCCL stream and does not queue behind the payload collectives of the step in flight,
    which is what lets a pipelined step learn the next batch's split sizes a whole step early."""

    def __init__(self, group=None, separate_count_channel=False):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.count_group = group
        if separate_count_channel:
            ranks = dist.get_process_group_ranks(group) if group is not None else list(range(self.world))
            self.count_group = dist.new_group(ranks=ranks)           # collective: every rank constructs its DistComm

    def exchange_counts(self, counts, out=None):
        if out is None:
            out = torch.empty_like(counts)
        dist.all_to_all_single(out, counts, group=self.count_group)
        return out

    def all_to_all(self, x, in_splits, out_splits):
        out = x.new_empty((sum(out_splits),) + tuple(x.shape[1:]))
        dist.all_to_all_single(out, x, output_split_sizes=out_splits, input_split_sizes=in_splits, group=self.group)
        return out

    def exchange(self, x, out):
        """Equal-split all-to-all of a [world * k, ...] buffer into a preallocated one (fixed-capacity exchanges)."""
        dist.all_to_all_single(out, x, group=self.group)
        return out

    def exchange_ids(self, x, out):
        """The same on the second communicator (the id messages of the NEXT batch, beside the current step)."""
        dist.all_to_all_single(out, x, group=self.count_group)
        return out

    def all_reduce_sum(self, x):
        dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group)
        return x


class HostStagedComm(DistComm):
    """REHEARSAL ONLY: the collectives of a gloo process group with every payload hopping through host memory, for
    several ranks that share ONE GPU (RCCL refuses two ranks on one device).  Every kernel of the step is the product
    path; only the exchanges differ.  Used by tests/test_sharded.py and by `bench.py --dist-backend gloo`, which rehearses
    the multi-rank control flow of the benchmark on a single-GPU box."""

    def exchange(self, x, out):
        torch.cuda.current_stream().synchronize()
        h = torch.empty(x.shape, dtype=x.dtype)
        dist.all_to_all_single(h, x.cpu(), group=self.group)
        return out.copy_(h)

    def exchange_ids(self, x, out):
        torch.cuda.current_stream().synchronize()
        h = torch.empty(x.shape, dtype=x.dtype)
        dist.all_to_all_single(h, x.cpu(), group=self.count_group)
        return out.copy_(h)

    def all_reduce_sum(self, x):
        h = x.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
        x.copy_(h)
        return x


class _Lookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, local_table, ids, emb):
        be, comm = emb.backend, emb.comm
        flat = ids.reshape(-1).contiguous()
        perm, counts, local_ids, flag = be.bucketize(flat, emb.rows_per_shard, comm.world)
        send = counts.tolist()                                   # host sync: RCCL needs the split sizes
        if flag is not None and int(flag.item()) != 0:
            raise IndexError("embedding id out of range [0, %d)" % emb.num_embeddings)
        recv = comm.exchange_counts(counts).tolist()             # C1 (counts)
        their_ids = comm.all_to_all(local_ids, send, recv)       # C1 (ids I must serve)
        rows = be.gather(local_table, their_ids)                 # local HIP gather
        back = comm.all_to_all(rows, recv, send)                 # C2
        out = be.permute_rows(back, perm, True)                  # out[perm[i]] = back[i]
        ctx.save_for_backward(perm, their_ids)
        ctx.meta = (emb, send, recv, tuple(local_table.shape))
        return out.reshape(tuple(ids.shape) + (local_table.shape[1],))

    @staticmethod
    def backward(ctx, g):
        perm, their_ids = ctx.saved_tensors
        emb, send, recv, shape = ctx.meta
        be, comm = emb.backend, emb.comm
        E = shape[1]
        g = g.contiguous().reshape(-1, E)
        g_sorted = be.permute_rows(g, perm, False)               # g_sorted[i] = g[perm[i]]
        g_theirs = comm.all_to_all(g_sorted, send, recv)         # C3
        if their_ids.numel() == 0:
            return torch.sparse_coo_tensor(torch.zeros((1, 0), dtype=torch.int64, device=g.device),
                                           g.new_zeros((0, E)), shape), None, None
        uniq, rows, _ = be.dedup_sum(their_ids, g_theirs, shape[0])
        grad = torch.sparse_coo_tensor(uniq[: rows.shape[0]].unsqueeze(0), rows, shape)
        return grad, None, None


class ShardedEmbedding(torch.nn.Module):
    """Embedding(V, E) whose rows are block-partitioned over the ranks of ``group``; this rank holds
    ``embeddings_shard`` [rows_per_shard, E] = global rows [rank*rows_per_shard, ...)."""

    def __init__(self, num_embeddings, embedding_dim, group=None, comm=None, backend=None, init_scale=0.05,
                 seed=1234):
        super().__init__()
        self.comm = comm if comm is not None else DistComm(group)
        self.backend = backend if backend is not None else HipBackend
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        P, r = self.comm.world, self.comm.rank
        self.rows_per_shard = -(-num_embeddings // P)
        lo = min(num_embeddings, r * self.rows_per_shard)
        hi = min(num_embeddings, lo + self.rows_per_shard)
        self.row_range = (lo, hi)
        g = torch.Generator().manual_seed(seed + r)
        shard = (torch.rand((self.rows_per_shard, embedding_dim), generator=g) * 2 - 1) * init_scale
        self.embeddings_shard = torch.nn.Parameter(shard)

    def load_global_rows(self, table):
        """Copy this rank's block out of a full [V,E] table (tests / checkpoint import)."""
        lo, hi = self.row_range
        with torch.no_grad():
            self.embeddings_shard.zero_()
            self.embeddings_shard[: hi - lo].copy_(table[lo:hi])

    def forward(self, ids):
        return _Lookup.apply(self.embeddings_shard, ids, self)


def allreduce_dense_grads(params, group=None):
    """C4: data-parallel SUM of the dense parameters' gradients as ONE flat buffer (a few MB at most here, so the
    ring is latency-bound: one collective, not one per tensor)."""
    grads = [p.grad for p in params if p.grad is not None and not p.grad.is_sparse]
    if not grads or dist.get_world_size(group) == 1:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].reshape(g.shape))
        off += n


class LocalShards:
    """P logical shards inside ONE process (memcpy "collective"): proves the exchange exact on a single GPU and
    is the world_size = 1 form of the same data path."""

    def __init__(self, table, n_shard, backend=None):
        self.backend = backend if backend is not None else HipBackend
        self.P = n_shard
        self.V, self.E = table.shape
        self.rows_per_shard = -(-self.V // n_shard)
        self.shards = [table[s * self.rows_per_shard:(s + 1) * self.rows_per_shard].contiguous()
                       for s in range(n_shard)]

    def lookup(self, ids):
        be = self.backend
        flat = ids.reshape(-1).contiguous()
        perm, counts, local_ids, _ = be.bucketize(flat, self.rows_per_shard, self.P)
        counts = counts.tolist()
        parts, start = [], 0
        for s in range(self.P):
            parts.append(be.gather(self.shards[s], local_ids[start:start + counts[s]].contiguous()))
            start += counts[s]
        back = torch.cat(parts) if parts else flat.new_zeros((0, self.E), dtype=torch.float32)
        return be.permute_rows(back, perm, True).reshape(tuple(ids.shape) + (self.E,))
