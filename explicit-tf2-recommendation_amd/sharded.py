"""Row-wise sharded embedding tables across the GPUs of a node.

The reference is single-device; this is the build's own model-parallel step for tables that do not fit (or should
not be replicated on) one GPU: 100M x 64d = 25.6 GB per tower at BASELINE config D.  One process per GPU;
``torch.distributed`` backend "nccl" is RCCL on ROCm, and on an MI355X node every GPU pair has its own xGMI link,
so all-to-all is the natural collective (one hop, all 7 links busy).

    block partition:  rows_per_shard = ceil(V / P),  owner = id // rows_per_shard,  local = id - owner*rows_per_shard
    forward   --  de-duplicate the batch's ids (sorted-unique plan; ascending = grouped by owner)
              C1  all-to-all of fixed-capacity id slabs (no count exchange, nothing read back by the host)
              --  local gather on the owner (HIP gather kernel)
              C2  all-to-all of the gathered rows back to the requesters (row = owner*cap + rank: no permutation)
    backward  --  ONE segment sum of the consumers' gradient rows in the order of the forward's plan (the slot is monotone
                  in the id: the plan of the ids is the plan of the slots), laid out dense by slot
              C3  all-to-all of the per-unique-id row gradients to the owners, who add the P ascending lists
                  (world size 1: no collective, no merge)

The integer side (bucketize / permutation / counts) is bit exact and independent of P; tests/test_sharded.py holds
``lookup == table[ids]`` bitwise for P in {1,2,4,8} logical shards on one device and for a 2-rank gloo group.
"""
import torch
import torch.distributed as dist

from . import ops


class HipBackend:
    """The device-side pieces, all HIP kernels on the current stream (ops.py / the C ABI); nothing here reads a value back
    to the host.  Tests on CPU inject an oracle-backed stand-in with the same methods; the product never does."""

    @staticmethod
    def bucketize(ids, rows_per_shard, n_shard):                 # (LocalShards: P logical shards in one process)
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
    def plan(ids, V):
        """Sorted-unique plan of the flat id list (own LSD radix sort, graph-safe): .uniq_ids / .seg_start / .perm /
        .n_uniq, padded tails."""
        return ops.DedupPlan(ids, V)

    @staticmethod
    def slab_map(plan, n, rows_per_shard, n_shard, cap, flag):
        """msg [P, 2+cap] (count, 0, owner-local ids ascending), slot [n] (row of every lookup in the [P*cap, E] buffer the
        rows come back in) and uslot [n] (the same per UNIQUE id, in the plan's order; P*cap -- one row past the buffer --
        beyond the last unique): rec_shard_slab_map_uslot_i64."""
        dev = plan.uniq_ids.device
        msg = torch.zeros((n_shard, cap + 2), dtype=torch.int64, device=dev)
        slot = torch.empty(n, dtype=torch.int64, device=dev)
        uslot = torch.empty(n, dtype=torch.int64, device=dev)
        ops.check(ops.lib.rec_shard_slab_map_uslot_i64(ops._ptr(plan.uniq_ids), ops._ptr(plan.n_uniq),
                                                       ops._ptr(plan.seg_start), ops._ptr(plan.perm), n, rows_per_shard,
                                                       n_shard, cap, ops._ptr(msg), ops._ptr(slot), ops._ptr(uslot),
                                                       ops._ptr(flag), ops._stream()), "rec_shard_slab_map_uslot_i64")
        return msg, slot, uslot

    @staticmethod
    def gather_lists(table, msg, n_shard, cap, flag):
        """Owner side: rows of the ids every rank asked for, [P*cap, E]; slots beyond a list's count are not touched."""
        V, E = table.shape
        out = torch.zeros((n_shard * cap, E), dtype=torch.float32, device=table.device)
        ops.check(ops.lib.rec_emb_gather_lists_f32(ops._ptr(table), V, E, table.stride(0), ops._ptr(msg), n_shard, cap,
                                                   ops._ptr(out), ops._ptr(flag), ops._stream()), "rec_emb_gather_lists_f32")
        return out

    @staticmethod
    def take_rows(rows, slots, sink, xplan=None):
        """out[i] = rows[slots[i]].  With the exchange's plan (``xplan``, covering exactly the lookups of `slots` followed by
        those the sink collects) the gradient of `rows` is dense and costs ONE segment sum -- the forward's de-duplication
        is reused; without it: the ordinary gather of layers.py (sparse gradient from a de-duplication of its own)."""
        if xplan is not None:
            return _TakeRows.apply(rows, slots, sink, xplan)
        from . import functional as Fn
        return Fn.Gather.apply(rows, slots, None, sink)

    @staticmethod
    def owner_reduce(msg, g_rows, n_shard, cap, rows_per_shard):
        """Union of the P ascending id lists that arrived (rank merge, no sort) and the row sums in its order:
        (uniq local ids [P*cap], rows [P*cap, E], n_uniq [1]); tails padded (valid id, zero rows)."""
        m, E = g_rows.shape
        dev = g_rows.device
        uniq = torch.empty(m, dtype=torch.int64, device=dev)
        seg = torch.empty(m + 1, dtype=torch.int32, device=dev)
        perm = torch.empty(m, dtype=torch.int32, device=dev)
        nu = torch.zeros(1, dtype=torch.int64, device=dev)
        nbytes = ops.lib.rec_dedup_workspace_bytes(m)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        ops.check(ops.lib.rec_dedup_plan_sorted_slabs_i64(ops._ptr(msg), n_shard, cap, rows_per_shard, ops._ptr(uniq),
                                                          ops._ptr(seg), ops._ptr(perm), ops._ptr(nu), ops._ptr(ws), nbytes,
                                                          ops._stream()), "rec_dedup_plan_sorted_slabs_i64")
        rows = ops.tops.segment_sum(g_rows.contiguous(), E, perm, seg, m, 1)
        return uniq, rows, nu


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
        """Equal-split all-to-all of a [world * k, ...] buffer into a preallocated one (fixed-capacity exchanges).  At world
        size 1 the only slab is the rank's own: it stays out of RCCL altogether (the buffer itself is handed back -- two
        17-MB self-copies through RCCL kernels were 40 us of the sharded DeepFM step)."""
        if self.world == 1:
            return x
        dist.all_to_all_single(out, x, group=self.group)
        return out

    def exchange_ids(self, x, out):
        """The same on the second communicator (the id messages of the NEXT batch, beside the current step)."""
        if self.world == 1:
            return x
        dist.all_to_all_single(out, x, group=self.count_group)
        return out

    def all_reduce_sum(self, x):
        if self.world > 1:
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


class _SlabPlan:
    """What the forward's de-duplication leaves for the backward: the sorted-unique plan of the exchanged ids -- which is
    also the plan of their slots (slot is monotone in the id) -- and the slot of every unique id."""
    __slots__ = ("plan", "uslot", "n", "world", "cap")

    def __init__(self, plan, uslot, n, world, cap):
        self.plan, self.uslot, self.n, self.world, self.cap = plan, uslot, n, world, cap


class _TakeRows(torch.autograd.Function):
    """out[i] = rows[slots[i]] on the local [P*cap, E] buffer of an exchange; backward = segment sums of the consumers'
    gradient in the order of the exchange's plan (no second sort), laid out dense by slot."""

    @staticmethod
    def forward(ctx, rows, slots, sink, xplan):
        out = ops.emb_gather(rows, slots, None)
        ctx.sink, ctx.xplan, ctx.n_rows = sink, xplan, rows.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        xp, E = ctx.xplan, g.shape[-1]
        g = g.contiguous().reshape(-1, E)
        _ids, buf = ctx.sink.take() if ctx.sink is not None else (None, None)
        if buf is not None:                                  # lookups of the same exchange that went elsewhere (DIN's
            buf[: g.shape[0]].copy_(g)                       # series through the attention kernel) wrote behind the head
        elif g.shape[0] == xp.n:
            buf = g
        else:                                                # lookups of the exchange nobody differentiated: zero rows
            buf = torch.zeros((xp.n, E), dtype=torch.float32, device=g.device)
            buf[: g.shape[0]].copy_(g)
        if buf.shape[0] != xp.n:
            raise ValueError("the exchange covered %d lookups, their gradients %d rows" % (xp.n, buf.shape[0]))
        sums = xp.plan.segment_sum(buf, E)                   # [n, E] in the plan's order; rows >= n_uniq are zero
        if xp.world == 1:
            return sums[: xp.cap], None, None, None          # one owner: slot = rank of the unique id (cap <= n)
        dense = torch.zeros((xp.world * xp.cap + 1, E), dtype=torch.float32, device=g.device)
        dense.index_copy_(0, xp.uslot, sums)                 # unique slots; the padded tail lands on the extra row
        return dense[: xp.world * xp.cap], None, None, None


class _Exchange(torch.autograd.Function):
    """ids -> (rows [P*cap, E], slot [n]): de-duplicate first, then exchange in FIXED-CAPACITY slabs -- constant split
    sizes, so no count exchange, nothing read back by the host, every shape known before the call: the whole lookup (and
    its backward) is a fixed program that engine.GraphedTrainStep captures in a hipGraph.

        plan     sorted-unique plan of the flat ids; ascending = already grouped by owner (block partition)
        map      rec_shard_slab_map_i64: owner o's slab of the id message, every lookup's slot o*cap + rank
        C1       all-to-all of the id messages [P, 2+cap] (own communicator)          -- skipped at world size 1
                 owner-side gather of the requested rows -> [P*cap, E]
        C2       all-to-all of the rows back: row o*cap + j = the j-th id this rank asked owner o for
      backward   the consumers' (sparse, de-duplicated) gradient of the rows buffer is laid out dense [P*cap, E],
        C3       travels to the owners, who add the P lists in the order of their union (rank merge, no sort)
    """

    @staticmethod
    def forward(ctx, shard, ids, emb, flag):
        be, comm = emb.backend, emb.comm
        P, rps = comm.world, emb.rows_per_shard
        n = ids.numel()
        cap = emb.capacity_for(n)
        plan = be.plan(ids, P * rps)
        msg, slot, uslot = be.slab_map(plan, n, rps, P, cap, flag)
        msg_theirs = comm.exchange_ids(msg, torch.empty_like(msg)) if P > 1 else msg          # C1
        rows_out = be.gather_lists(shard, msg_theirs, P, cap, flag)
        rows = comm.exchange(rows_out, torch.empty_like(rows_out)) if P > 1 else rows_out     # C2
        emb._xplan = _SlabPlan(plan, uslot, n, P, cap)       # for the consumers' backward (ShardedEmbedding.take)
        # one owner: the unique ids it was asked for ARE the plan's (ascending, tail = the first id, as DedupPlan pads)
        ctx.save_for_backward(msg_theirs, plan.uniq_ids[:cap] if P == 1 else None)
        ctx.meta = (emb, cap, tuple(shard.shape))
        ctx.mark_non_differentiable(slot)
        return rows, slot

    @staticmethod
    def backward(ctx, g, _gslot):
        msg_theirs, uniq_own = ctx.saved_tensors
        emb, cap, shape = ctx.meta
        be, comm = emb.backend, emb.comm
        P = comm.world
        if g.is_sparse:
            # (uniq slots, row sums) of the consumers' de-duplication.  Its padded tail repeats the first slot with zero rows:
            # exact to add, but hundreds of thousands of atomic adds onto ONE row (DIN: half of 1.2 M lookups are the padding
            # id) took 20 ms -- zero rows that repeat the first slot are dealt out over 8192 scratch rows behind the buffer
            idx, vals = g._indices()[0], g._values()
            tail = (idx == idx[:1]) & (vals.abs().amax(dim=1) == 0)
            tail[:1].zero_()                                        # (in place on the device: capturable)
            spread = P * cap + (torch.arange(idx.numel(), device=idx.device) & 8191)
            idx = torch.where(tail, spread, idx)
            dense = torch.zeros((P * cap + 8192, shape[1]), dtype=torch.float32, device=g.device)
            dense.index_add_(0, idx, vals)
            g = dense[: P * cap]
        g = g.contiguous()
        if P == 1 and uniq_own is not None:
            # one owner, one list: nothing to merge -- the rows are already the sums per unique id, in ascending id order
            return torch.sparse_coo_tensor(uniq_own.unsqueeze(0), g, shape), None, None, None
        g_theirs = comm.exchange(g, torch.empty_like(g)) if P > 1 else g                       # C3
        uniq, rows, _ = be.owner_reduce(msg_theirs, g_theirs, P, cap, emb.rows_per_shard)
        return torch.sparse_coo_tensor(uniq[: rows.shape[0]].unsqueeze(0), rows, shape), None, None, None


class ShardedEmbedding(torch.nn.Module):
    """Embedding(V, E) whose rows are block-partitioned over the ranks of ``group``: this rank holds
    ``embeddings_shard`` [rows_per_shard, E] = global rows [rank*rows_per_shard, ...).  Plugs in where the reference
    builds ``tf.keras.layers.Embedding`` (2.FM/CustomLayers.py:176-178, 5.DIN/CustomLayers.py:216-217): ``forward(X, oob,
    sink)`` has the call signature of layers.Embedding.

    ``capacity``: slots per owner of the fixed-capacity exchange = the most UNIQUE ids one call can hold for one owner.
    Default min(lookups, rows_per_shard) -- always enough; a caller that knows the field layout passes the tighter
    engine.exchange_capacity(...) bound (P x capacity rows travel per exchange).  Too small a capacity sets the flag
    that ``check_flags()`` turns into an IndexError (as an out-of-range id does)."""

    def __init__(self, num_embeddings, embedding_dim, group=None, comm=None, backend=None, init_scale=0.05,
                 seed=1234, capacity=None, device=None):
        super().__init__()
        self.comm = comm if comm is not None else DistComm(group, separate_count_channel=True)
        self.backend = backend if backend is not None else HipBackend
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        P, r = self.comm.world, self.comm.rank
        self.rows_per_shard = -(-num_embeddings // P)
        lo = min(num_embeddings, r * self.rows_per_shard)
        hi = min(num_embeddings, lo + self.rows_per_shard)
        self.row_range = (lo, hi)
        self.capacity = capacity
        if device is not None and torch.device(device).type == "cuda":
            # a large shard is drawn on the device (a host-side torch.rand of 25.6 GB takes minutes)
            g = torch.Generator(device=device).manual_seed(seed + r)
            shard = torch.empty((self.rows_per_shard, embedding_dim), dtype=torch.float32, device=device)
            shard.uniform_(-init_scale, init_scale, generator=g)
        else:
            g = torch.Generator().manual_seed(seed + r)
            shard = (torch.rand((self.rows_per_shard, embedding_dim), generator=g) * 2 - 1) * init_scale
        self.embeddings_shard = torch.nn.Parameter(shard)
        self.flag = None                                         # device int32: overflow / out-of-range, set by the kernels
        self._xplan = None                                       # the last exchange's plan (take())

    def capacity_for(self, n):
        cap = min(int(n), self.rows_per_shard)
        if self.capacity is not None:
            cap = min(cap, int(self.capacity))
        return max(cap, 1)

    def load_global_rows(self, table):
        """Copy this rank's block out of a full [V,E] table (tests / checkpoint import)."""
        lo, hi = self.row_range
        with torch.no_grad():
            self.embeddings_shard.zero_()
            self.embeddings_shard[: hi - lo].copy_(table[lo:hi])

    def grad_sink(self, X, n_tail=0):
        """``n_tail``: lookups of the exchange BEHIND those the sink collects that nobody differentiates (DIN's padding
        id): the shared gradient buffer gets as many zero rows, so that it lines up with the exchange's plan."""
        from . import functional as Fn
        if torch.is_grad_enabled() and self.embeddings_shard.requires_grad:
            return Fn.GradSink(X.numel(), n_tail)
        return None

    def take(self, rows, slots, sink=None):
        """rows[slots] for the lookups at the HEAD of the last exchange's id list (the rest, if any, reach their consumers
        through ``sink``): the gradient of the rows buffer comes out of the exchange's own plan."""
        return self.backend.take_rows(rows, slots, sink, self._xplan)

    def exchange(self, ids, oob=None):
        """(rows [P*cap, E] -- differentiable --, slot [n]) for the flat id list: rows[slot[i]] = table[ids[i]].  For a
        layer with several lookups into this table (DIN: profile + behaviour series) ONE exchange serves them all."""
        ids = ids.reshape(-1).contiguous()
        if oob is None and ids.is_cuda:
            if self.flag is None:
                self.flag = torch.zeros(1, dtype=torch.int32, device=ids.device)
            oob = self.flag
        return _Exchange.apply(self.embeddings_shard, ids, self, oob)

    def forward(self, X, oob=None, sink=None):
        rows, slot = self.exchange(X, oob)
        out = self.take(rows, slot.reshape(X.shape), sink)
        return out.reshape(tuple(X.shape) + (self.embedding_dim,))

    def check_flags(self):
        if self.flag is not None and int(self.flag.item()) != 0:
            raise IndexError("embedding id out of range [0, %d), or more unique ids for one owner than the exchange "
                             "capacity" % self.num_embeddings)


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
