"""Train-step engines: the reference's ``train_loop`` (2.FM/ModelManager.py:171-181) for one layer family as a
fixed sequence of C-ABI calls over preallocated device buffers.

    GradientTape -> model(inputs) -> BinaryCrossentropy -> tape.gradient -> (optionally) Adam.apply_gradients

Nothing is allocated and nothing synchronises inside a step, so a step can be captured once into a hipGraph
(``torch.cuda.CUDAGraph``) and replayed: the launch-bound chain of small kernels then costs one graph launch on
the host.  Gradients come out exactly as the autograd path of layers.py produces them (dense tensors for dense
parameters; (uniq_ids, rows, n_uniq) for the tables) -- tests/test_gpu_engine.py holds the two paths equal.
"""
import ctypes as C

import torch

from . import ops
from ._lib import lib, check


# Stream captures use the thread-local error mode: with a process group alive, ProcessGroupNCCL's watchdog thread polls
# the events of finished collectives (hipEventQuery) whenever it likes, and in the default global mode such a call from
# ANOTHER thread during a capture invalidates the capture and terminates the process ("operation not permitted when
# stream is capturing") -- seen in bench.py --sharded, where graphs are captured after collectives have run.
CAPTURE_MODE = "thread_local"


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class _Program:
    """A recorded list of (name, fn, args) C-ABI calls; ``run(stream)`` enqueues them in order."""

    def __init__(self):
        self.calls = []

    def add(self, name, *args):
        self.calls.append((name, getattr(lib, name), args))

    def run(self, stream):
        for name, fn, args in self.calls:
            st = fn(*args, stream)
            if st != 0:
                check(st, name)


def _tables_share_rows(embed, w):
    """embed [V,E] and w [V,1] are the two strided views of one fused [V,ld] array (layers._FMTables.fuse_tables)."""
    E = embed.shape[1]
    return (embed.is_cuda and E % 4 == 0 and embed.stride(0) == w.stride(0) and embed.stride(0) % 4 == 0
            and embed.stride(0) >= E + 1 and w.data_ptr() == embed.data_ptr() + 4 * E and embed.data_ptr() % 16 == 0)


class DeepFMTrainStep:
    """fwd + bwd (+ optimizer) of DeepFMRankingLayer (2.FM/CustomLayers.py:279-308) under the reference's loss.

    optimizer: None (gradients only -- the 'fwd+bwd' of the headline metric), 'keras_adam' (the reference's semantics:
    dense sweep over the tables) or 'lazy_adam' (touched rows only; NOT the reference's semantics).
    """

    def __init__(self, layer, batch_size, optimizer=None, lr=1e-3, use_graph=True):
        self.layer = layer
        self.B = B = int(batch_size)
        self.F = F = len(layer.feature_names)
        self.V, self.E = layer.embed.embeddings.shape
        E = self.E
        dev = layer.embed.embeddings.device
        self.dev = dev
        self.optimizer = optimizer
        self.lr = lr
        self.use_graph = use_graph
        self.t = 0
        f32 = dict(dtype=torch.float32, device=dev)
        n = B * F
        u1, u2 = layer.mlp_dims
        D = F * E
        self.X = torch.empty((B, F), dtype=torch.int64, device=dev)
        self.z_fm = torch.empty(B, **f32)
        self.rows = torch.empty((B, F, E), **f32)
        self.S = torch.empty((B, E), **f32)
        self.h1 = torch.empty((B, u1), **f32)
        self.h2 = torch.empty((B, u2), **f32)
        self.dnn = torch.empty((B, 1), **f32)
        self.prob = torch.empty((B, 1), **f32)
        self.loss = torch.empty(1, **f32)
        self.dz = torch.empty(B, **f32)
        self.dh2 = torch.empty((B, u2), **f32)
        self.dh1 = torch.empty((B, u1), **f32)
        self.drows = torch.empty((B, D), **f32)
        self.vals = torch.empty((n, E), **f32)
        self.oob = torch.zeros(1, dtype=torch.int32, device=dev)
        # gradients
        self.g = {
            "MLP_layer1.kernel_0": torch.empty((D, u1), **f32), "MLP_layer1.bias_0": torch.empty(u1, **f32),
            "MLP_layer1.kernel_1": torch.empty((u1, u2), **f32), "MLP_layer1.bias_1": torch.empty(u2, **f32),
            "MLP_layer2.kernel_0": torch.empty((u2, 1), **f32), "MLP_layer2.bias_0": torch.empty(1, **f32),
            "bias": torch.empty(1, **f32),
        }
        self.uniq_ids = torch.empty(n, dtype=torch.int64, device=dev)
        self.seg_start = torch.empty(n + 1, dtype=torch.int32, device=dev)
        self.perm = torch.empty(n, dtype=torch.int32, device=dev)
        self.n_uniq = torch.zeros(1, dtype=torch.int64, device=dev)
        self.g_embed_rows = torch.empty((n, E), **f32)
        self.g_w_rows = torch.empty((n, 1), **f32)
        self.dedup_bytes = lib.rec_dedup_workspace_bytes(n)
        self.dedup_ws = torch.empty(self.dedup_bytes, dtype=torch.uint8, device=dev)
        # split-K partials for the weight-gradient GEMMs (reduction over the batch)
        self.sk0 = ops.split_k_for(B, D, u1)
        self.sk1 = ops.split_k_for(B, u1, u2)
        self.sk2 = ops.split_k_for(B, u2, 1)
        self.ws0 = torch.empty((self.sk0, D, u1), **f32) if self.sk0 > 1 else None
        self.ws1 = torch.empty((self.sk1, u1, u2), **f32) if self.sk1 > 1 else None
        self.ws2 = torch.empty((self.sk2, u2, 1), **f32) if self.sk2 > 1 else None
        if optimizer is not None:
            self.state = {}
            for name, p in layer.named_parameters():
                self.state[name] = (torch.zeros(p.shape, **f32), torch.zeros(p.shape, **f32))   # dense m, v
            self.side_e = torch.empty((n, 3, E), **f32)
            self.side_w = torch.empty((n, 3, 1), **f32)
        self.colsum_ws = torch.empty(lib.rec_colsum_workspace_bytes(B, max(u1, u2)) // 4 + 1, **f32)
        self.segsum_ws = torch.empty(lib.rec_segment_sum_workspace_bytes(n, E) // 4, **f32)
        self._graphs = {}
        self._static_prog = self._build_static()

    # -- program construction ---------------------------------------------------------------------
    def _build_static(self):
        """Everything after index assembly; independent of where the input tensors live."""
        L = self.layer
        B, F, E, V = self.B, self.F, self.E, self.V
        u1, u2 = L.mlp_dims
        D = F * E
        emb, w, bias = L.embed.embeddings, L.w.embeddings, L.bias
        K0, b0 = L.MLP_layer1.kernel_0, L.MLP_layer1.bias_0
        K1, b1 = L.MLP_layer1.kernel_1, L.MLP_layer1.bias_1
        K2, b2 = L.MLP_layer2.kernel_0, L.MLP_layer2.bias_0
        P = _Program()
        # ---- forward
        P.add("rec_emb_fm_fwd_f32", _p(emb), emb.stride(0), _p(w), w.stride(0), _p(bias), V, E, _p(self.X), B, F,
              _p(self.z_fm), None, _p(self.rows), _p(self.S), _p(self.oob))
        P.add("rec_gemm_f32", 0, 0, B, u1, D, _p(self.rows), D, _p(K0), u1, _p(self.h1), u1, ops.EPI_BIAS_RELU, _p(b0),
              None, 0, None, 0, 1, None, None)
        P.add("rec_gemm_f32", 0, 0, B, u2, u1, _p(self.h1), u1, _p(K1), u2, _p(self.h2), u2, ops.EPI_BIAS_RELU, _p(b1),
              None, 0, None, 0, 1, None, None)
        P.add("rec_gemm_f32", 0, 0, B, 1, u2, _p(self.h2), u2, _p(K2), 1, _p(self.dnn), 1, ops.EPI_BIAS, _p(b2),
              None, 0, None, 0, 1, None, None)
        P.add("rec_act_fwd_f32", ops.ACT_SIGMOID, _p(self.dnn), _p(self.z_fm), _p(self.prob), B)
        self._loss_call_index = len(P.calls)
        P.add("rec_bce_fwd_bwd_f32", None, _p(self.prob), B, _p(self.loss), None, _p(self.dz))   # y bound per batch
        # ---- backward: MLP_layer2 (linear)
        g = self.g
        P.add("rec_gemm_f32", 1, 0, u2, 1, B, _p(self.h2), u2, _p(self.dz), 1, _p(g["MLP_layer2.kernel_0"]), 1,
              ops.EPI_NONE, None, None, 0, None, 0, self.sk2, _p(self.ws2), None)
        P.add("rec_colsum_f32", _p(self.dz), B, 1, 1, _p(g["MLP_layer2.bias_0"]), _p(self.colsum_ws))
        P.add("rec_gemm_f32", 0, 1, B, u2, 1, _p(self.dz), 1, _p(K2), 1, _p(self.dh2), u2, ops.EPI_NONE, None, None, 0,
              None, 0, 1, None, None)
        # ---- MLP_layer1 layer 1 (relu)
        P.add("rec_act_bwd_f32", ops.ACT_RELU, _p(self.h2), _p(self.dh2), _p(self.dh2), B * u2)
        P.add("rec_gemm_f32", 1, 0, u1, u2, B, _p(self.h1), u1, _p(self.dh2), u2, _p(g["MLP_layer1.kernel_1"]), u2,
              ops.EPI_NONE, None, None, 0, None, 0, self.sk1, _p(self.ws1), None)
        P.add("rec_colsum_f32", _p(self.dh2), B, u2, u2, _p(g["MLP_layer1.bias_1"]), _p(self.colsum_ws))
        P.add("rec_gemm_f32", 0, 1, B, u1, u2, _p(self.dh2), u2, _p(K1), u2, _p(self.dh1), u1, ops.EPI_NONE, None, None,
              0, None, 0, 1, None, None)
        # ---- MLP_layer1 layer 0 (relu)
        P.add("rec_act_bwd_f32", ops.ACT_RELU, _p(self.h1), _p(self.dh1), _p(self.dh1), B * u1)
        P.add("rec_gemm_f32", 1, 0, D, u1, B, _p(self.rows), D, _p(self.dh1), u1, _p(g["MLP_layer1.kernel_0"]), u1,
              ops.EPI_NONE, None, None, 0, None, 0, self.sk0, _p(self.ws0), None)
        P.add("rec_colsum_f32", _p(self.dh1), B, u1, u1, _p(g["MLP_layer1.bias_0"]), _p(self.colsum_ws))
        P.add("rec_gemm_f32", 0, 1, B, D, u1, _p(self.dh1), u1, _p(K0), u1, _p(self.drows), D, ops.EPI_NONE, None, None,
              0, None, 0, 1, None, None)
        # ---- tables: IndexedSlices values, de-duplication, segment sums
        P.add("rec_emb_fm_bwd_vals_f32", _p(emb), emb.stride(0), V, E, _p(self.X), B, F, _p(self.dz), _p(self.S),
              _p(self.rows), _p(self.drows), _p(self.vals))
        P.add("rec_dedup_plan_i64", _p(self.X), B * F, V, _p(self.uniq_ids), _p(self.seg_start), _p(self.perm),
              _p(self.n_uniq), _p(self.dedup_ws), self.dedup_bytes)
        P.add("rec_segment_sum_f32", _p(self.vals), E, _p(self.perm), _p(self.seg_start), B * F, 1,
              _p(self.g_embed_rows), _p(self.segsum_ws))
        P.add("rec_segment_sum_f32", _p(self.dz), 1, _p(self.perm), _p(self.seg_start), B * F, F, _p(self.g_w_rows),
              _p(self.segsum_ws))
        P.add("rec_colsum_f32", _p(self.dz), B, 1, 1, _p(g["bias"]), _p(self.colsum_ws))
        return P

    def _optimizer_program(self, t):
        L = self.layer
        P = _Program()
        lr, b1, b2, eps = self.lr, 0.9, 0.999, 1e-7
        params = dict(L.named_parameters())
        for name, grad in self.g.items():
            m, v = self.state[name]
            P.add("rec_adam_dense_f32", _p(params[name]), _p(m), _p(v), _p(grad), grad.numel(), t, lr, b1, b2, eps)
        n = self.B * self.F
        pe, pw = params["embed.embeddings"], params["w.embeddings"]
        if self.optimizer == "keras_adam" and _tables_share_rows(pe, pw):
            (me, ve), (mw, vw) = self.state["embed.embeddings"], self.state["w.embeddings"]
            P.add("rec_adam_sparse_keras_pair_f32", _p(pe), pe.stride(0), _p(me), _p(ve), _p(mw), _p(vw), self.V, self.E,
                  _p(self.uniq_ids), _p(self.g_embed_rows), _p(self.g_w_rows), _p(self.n_uniq), n, _p(self.side_e),
                  _p(self.side_w), t, lr, b1, b2, eps)
            return P
        for name, rows, side, E in (("embed.embeddings", self.g_embed_rows, self.side_e, self.E),
                                    ("w.embeddings", self.g_w_rows, self.side_w, 1)):
            m, v = self.state[name]
            if self.optimizer == "keras_adam":
                P.add("rec_adam_sparse_keras_f32", _p(params[name]), params[name].stride(0), _p(m), _p(v), self.V, E,
                      _p(self.uniq_ids), _p(rows), _p(self.n_uniq), n, _p(side), t, lr, b1, b2, eps)
            else:
                P.add("rec_adam_rows_f32", _p(params[name]), params[name].stride(0), _p(m), _p(v), self.V, E,
                      _p(self.uniq_ids), _p(rows), _p(self.n_uniq), n, t, lr, b1, b2, eps)
        return P

    # -- execution ----------------------------------------------------------------------------------
    def _enqueue(self, cols, label, stream, t):
        F = self.F
        arr = (C.c_void_p * F)(*[c.data_ptr() for c in cols])
        check(lib.rec_index_pack_i64(arr, F, self.B, _p(self.X), F, 0, stream), "rec_index_pack_i64")
        calls = self._static_prog.calls
        name, fn, args = calls[self._loss_call_index]
        calls[self._loss_call_index] = (name, fn, (_p(label),) + args[1:])
        self._static_prog.run(stream)
        if self.optimizer is not None:
            self._optimizer_program(t).run(stream)

    def _check_inputs(self, inputs, label_name):
        cols = []
        for name in self.layer.feature_names:
            t = inputs[name]
            if t.dtype != torch.int64 or not t.is_cuda or t.numel() != self.B or not t.is_contiguous():
                raise ValueError("feature %r must be a contiguous int64 CUDA tensor with %d ids" % (name, self.B))
            cols.append(t)
        y = inputs[label_name]
        if y.dtype != torch.float32 or not y.is_cuda or y.numel() != self.B:
            raise ValueError("label must be a float32 CUDA tensor with %d entries" % self.B)
        return cols, y

    def __call__(self, inputs, label_name="label"):
        """One train_loop iteration.  Returns the device scalar loss (no synchronisation)."""
        cols, y = self._check_inputs(inputs, label_name)
        self.t += 1
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if not self.use_graph or self.optimizer is not None:
            # the optimizer's bias correction depends on t: enqueue eagerly
            self._enqueue(cols, y, stream, self.t)
            return self.loss
        key = tuple(c.data_ptr() for c in cols) + (y.data_ptr(),)
        g = self._graphs.get(key)
        if g is None:
            # warm-up outside capture, then capture the same sequence
            self._enqueue(cols, y, stream, self.t)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                cs = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                self._enqueue(cols, y, cs, self.t)
            self._graphs[key] = (g, cols, y)      # keep the inputs alive: the graph holds their addresses
            g = self._graphs[key]
        g[0].replay()
        return self.loss

    def gradients(self):
        """Dense grads by parameter name + the two tables' (uniq_ids, rows, n_uniq)."""
        out = dict(self.g)
        out["embed.embeddings"] = (self.uniq_ids, self.g_embed_rows, self.n_uniq)
        out["w.embeddings"] = (self.uniq_ids, self.g_w_rows, self.n_uniq)
        return out


def _bits(n):
    b = 1
    while (1 << b) < n:
        b += 1
    return b


class DeepFMFusedStep:
    """The same train_loop iteration as DeepFMTrainStep in two launches on the main stream (csrc/deepfm_fused.hip):
    the fused forward+backward kernel, then ONE launch for the fixed-order reduction of its partials and the segment
    sums -- plus, behind the steps of a call, the per-column LDS sort of the de-duplication plans of the batches announced
    for the next call (they depend only on the ids).  ``many()`` runs several iterations as one captured hipGraph.

    Requirements (checked; otherwise use DeepFMTrainStep): embedding_dims 16, mlp_dims [32,8], fused table layout,
    F <= 28, B <= 16384, and the DataGenerator id-space contract -- ``field_offsets[f]``/``field_dims[f]`` =
    ``data_info.json``'s offsets and dims (2.FM/DataGenerator.py:126-134); an id outside its field's range sets
    ``self.bad_ids`` (checked by ``check_flags()``).
    """

    NBUF = 64           # plan buffers (see __init__): two halves of 32, many() alternates between them (a hipGraph launch
                        # leaves the GPU idle for ~30 us, so a call may hold up to 32 steps: ~1 us per step at that length)
    MAX_GRAPHS = 64     # captured hipGraphs kept (least recently used beyond that are dropped with the inputs they hold)

    def __init__(self, layer, batch_size, field_dims, field_offsets, optimizer=None, lr=1e-3, use_graph=True,
                 direct=True, kernel=None, want_prob=False):
        self.layer = layer
        self.direct = bool(direct)
        self.B = B = int(batch_size)
        self.F = F = len(layer.feature_names)
        emb, w = layer.embed.embeddings, layer.w.embeddings
        self.V, self.E = emb.shape
        if self.E != 16 or list(layer.mlp_dims) != [32, 8]:
            raise NotImplementedError("the fused step covers embedding_dims=16, mlp_dims=[32,8]")
        if emb.stride(0) != 32 or w.stride(0) != 32 or w.data_ptr() != emb.data_ptr() + 64:
            raise NotImplementedError("the fused step needs the fused [embed|w|pad] table layout (layer.cuda())")
        if F > 28 or B > 16384 or len(field_dims) != F or len(field_offsets) != F:
            raise NotImplementedError("fused step: F <= 28, B <= 16384, one (dim, offset) per feature")
        if any(field_offsets[i] >= field_offsets[i + 1] for i in range(F - 1)):
            raise ValueError("field offsets must be ascending (DataGenerator contract)")
        self.max_key = max(int(d) for d in field_dims) - 1
        if _bits(self.max_key + 1) + _bits(B) > 32 or ((self.max_key << _bits(B)) | (B - 1)) >= 0xFFFFFFFF:
            raise NotImplementedError("field too wide for the 32-bit sort words at this batch size")
        dev = emb.device
        self.dev = dev
        if optimizer not in (None, "keras_adam", "lazy_adam", "keras_adam_lazy"):
            raise ValueError("optimizer must be None, 'keras_adam', 'lazy_adam' or 'keras_adam_lazy', not %r" % (optimizer,))
        if optimizer == "keras_adam_lazy" and not self.direct:
            # ('lazy_adam' has a plan-after form: rec_adam_rows_f32 over the finished row sums -- the same touched-rows
            # arithmetic; the lazily EVALUATED Keras Adam only exists inside the direct-mode post launch)
            raise ValueError("optimizer 'keras_adam_lazy' applies its update inside the direct-mode post launch: direct=True")
        self.optimizer, self.lr, self.use_graph, self.t = optimizer, lr, use_graph, 0
        f32 = dict(dtype=torch.float32, device=dev)
        n, D = B * F, F * 16
        self.col_lo = torch.tensor([int(o) for o in field_offsets], dtype=torch.int64, device=dev)
        self.gz = torch.empty(B, **f32)
        self.vals = torch.empty((n, 16), **f32)
        # per-step results of a many() call (a caller that keeps metrics reads them after the call): loss_steps[i] and --
        # want_prob -- prob_steps[i] of the i-th batch; `loss` / `prob` are views of the LAST step's entries
        self.loss_steps = torch.zeros(self.NBUF // 2, **f32)
        self.prob_steps = torch.empty((self.NBUF // 2, B), **f32) if want_prob else None
        self.loss = self.loss_steps[:1]
        self.prob = self.prob_steps[0] if want_prob else None
        self._row = 0                                                # row of the step being enqueued
        self.oob = torch.zeros(1, dtype=torch.int32, device=dev)
        self.bad_ids = torch.zeros(1, dtype=torch.int32, device=dev)
        self.g = {
            "MLP_layer1.kernel_0": torch.empty((D, 32), **f32), "MLP_layer1.bias_0": torch.empty(32, **f32),
            "MLP_layer1.kernel_1": torch.empty((32, 8), **f32), "MLP_layer1.bias_1": torch.empty(8, **f32),
            "MLP_layer2.kernel_0": torch.empty((8, 1), **f32), "MLP_layer2.bias_0": torch.empty(1, **f32),
            "bias": torch.empty(1, **f32),
        }
        self.ws = torch.empty(lib.rec_deepfm_fused_workspace_bytes(B, F), dtype=torch.uint8, device=dev)
        # NBUF plan buffers: the de-duplication plan depends on the ids only, so the plans of the NEXT call's batches are
        # built behind the steps of this call.  Every batch of a call has a buffer of its own (two halves of NBUF / 2 used
        # alternately: one is read by this call's steps while the other is filled for the next call)
        NB = self.NBUF
        self._perm = torch.empty((NB, F, B), dtype=torch.int32, device=dev)
        self._col_uid = torch.empty((NB, F, B), dtype=torch.int64, device=dev)
        self._col_seg = torch.empty((NB, F, B + 1), dtype=torch.int32, device=dev)
        self._col_nu = torch.zeros((NB, F), dtype=torch.int32, device=dev)
        self._dloc = torch.empty((NB, F, B), dtype=torch.int32, device=dev)
        self.plans = [dict(perm=self._perm[b], col_uid=self._col_uid[b], col_seg=self._col_seg[b],
                           col_nu=self._col_nu[b], dloc=self._dloc[b]) for b in range(NB)]
        # consecutive buffers are contiguous, so ONE sort call can build the plans of GROUP upcoming batches as
        # GROUP*F columns (the sort kernels are latency-bound at < 1 wave per SIMD: two batches cost ~1.2x one)
        # (256 column pointers per sort launch: one workgroup per column, and a CU holds ONE sort workgroup (86 KB of LDS) --
        # 17 batches per launch, 442 workgroups, were measured slower than 9)
        self.GROUP = max(1, 256 // F)
        self.col_lo_rep = self.col_lo.repeat(self.GROUP).contiguous()
        self._prefetched, self._half = {}, 0     # plans announced by the previous call: id-tensor addresses -> buffer
        self._col_cache = {}
        import operator
        names_ = list(layer.feature_names)
        self._getter = (operator.itemgetter(*names_) if len(names_) > 1 else (lambda d, n=names_[0]: (d[n],)))
        self.uniq_ids = torch.empty(n, dtype=torch.int64, device=dev)
        self.g_embed_rows = torch.empty((n, 16), **f32)
        self.g_w_rows = torch.empty((n, 1), **f32)
        self.n_uniq = torch.zeros(1, dtype=torch.int64, device=dev)
        self.sort_ws = torch.empty(lib.rec_colsort_workspace_bytes(B, F * self.GROUP), dtype=torch.uint8, device=dev)
        # (the plan sorts run on the caller's stream behind the steps of a call.  A second stream was tried at every
        # priority the device offers -- the range is (0, -1), the default 0 is the lowest, and a step enqueued on a
        # priority -1 stream ran 2.5x SLOWER from its graphs: a sort workgroup cannot share a CU with a fused-kernel
        # workgroup, so concurrency only moved the wait into one fused launch in eight)
        self._advanced = False                               # the fused launch of the step in flight advanced the step counter
        if optimizer is not None:
            self.state = {name: (torch.zeros(p.shape, **f32), torch.zeros(p.shape, **f32))
                          for name, p in layer.named_parameters()}
            self.side_e = torch.empty((n, 3, 16), **f32)
            self.side_w = torch.empty((n, 3, 1), **f32)
        if self._fused_lazy():
            # optimizer state packed beside the rows: [m 16 | v 16] of a row is ONE 128-byte line, m_w / v_w live in
            # floats 17 / 18 of the fused table row (its padding) -- a touched row costs two line requests instead of
            # five or six (table row, m_e, v_e, m_w, v_w).  state[...] stays a pair of (strided) views
            fused = getattr(layer, "_fused_storage", None)
            if fused is not None and fused.shape[1] == 32:
                # floats 17 / 18 of every fused table row are RESERVED for this step's m_w / v_w: the storage must still be
                # what embed.embeddings views, and only one step may own it (a second one would wipe the first one's state)
                if fused.data_ptr() != emb.data_ptr():
                    raise ValueError("layer._fused_storage no longer aliases embed.embeddings")
                owner = getattr(layer, "_fused_state_owner", None)
                if owner is not None and owner() is not None and owner() is not self:
                    raise ValueError("another DeepFMFusedStep already keeps its optimizer state in this layer's table "
                                     "padding (floats 17/18 of the fused rows): one lazy-optimizer step per layer")
                import weakref
                layer._fused_state_owner = weakref.ref(self)
                self._mv = torch.zeros((self.V, 32), **f32)
                fused[:, 17:19].zero_()
                self.state["embed.embeddings"] = (self._mv[:, :16], self._mv[:, 16:])
                self.state["w.embeddings"] = (fused[:, 17:18], fused[:, 18:19])
            self._last = (torch.zeros(self.V, dtype=torch.int32, device=dev)
                          if optimizer == "keras_adam_lazy" else None)    # the step every row holds
            # the step counter and the bias-corrected step size live on the device (rec_adam_advance_f32): the train step
            # holds no per-step host scalar, so it is captured and replayed like the gradient-only step.  The table holds
            # lr_t of steps 1..N exactly as the host-side entry points compute it; beyond it the corrections are 1.0f
            N = 32768
            tab = [lib.rec_adam_lr_t_f32(lr, 0.9, 0.999, t) for t in range(1, N + 1)]
            self._lr_tab = torch.tensor(tab, **f32)
            self._step_dev = torch.zeros(1, dtype=torch.int64, device=dev)
            self._lr_t_dev = torch.zeros(1, **f32)
            params = dict(layer.named_parameters())
            names = list(self.g)
            k = len(names)
            self._multi = (k, (C.c_void_p * k)(*[params[nm].data_ptr() for nm in names]),
                           (C.c_void_p * k)(*[self.state[nm][0].data_ptr() for nm in names]),
                           (C.c_void_p * k)(*[self.state[nm][1].data_ptr() for nm in names]),
                           (C.c_void_p * k)(*[self.g[nm].data_ptr() for nm in names]),
                           (C.c_int64 * k)(*[self.g[nm].numel() for nm in names]))
        import collections
        self._graphs = collections.OrderedDict()            # gkey -> (graph, inputs kept alive), LRU order
        self._seen = collections.OrderedDict()              # gkeys enqueued eagerly once (addresses only: nothing is held)
        # transposed copy of layer 1's kernel for the fused kernel (csrc/deepfm_fused3.hip reads its K0 operand as 16-byte
        # pieces of K0^T); refreshed whenever the parameter changed -- by torch (its version counter) or by this step's
        # own optimizer launches (which re-transpose in the same stream)
        self.kernel_version = 3 if kernel is None else int(kernel)
        self._k0t = torch.empty((32, D), **f32)
        self._k0_ver = None

    def _ensure_k0t(self, st=None):
        if self.F <= 26:
            return                                           # K0 is staged in LDS by the kernel itself: K0T is not read
        K0 = self.layer.MLP_layer1.kernel_0
        ver = (K0._version, K0.data_ptr())
        if ver != self._k0_ver:
            if st is None:
                st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            check(lib.rec_deepfm_k0t_f32(_p(K0), self.F, _p(self._k0t), st), "rec_deepfm_k0t_f32")
            self._k0_ver = ver

    def _retranspose(self, st):
        """after an in-stream update of the dense parameters through the C ABI (torch's version counter does not see it)"""
        if self.F <= 26:
            return
        check(lib.rec_deepfm_k0t_f32(_p(self.layer.MLP_layer1.kernel_0), self.F, _p(self._k0t), st), "rec_deepfm_k0t_f32")

    def _sort(self, cols, buf, stream):
        self._sort_group([cols], buf, stream)

    def _sort_group(self, cols_list, first_buf, stream):
        """Plans of len(cols_list) <= GROUP batches into the consecutive buffers first_buf, first_buf+1, ... as one
        rec_colsort_plan_i64 call over len*F columns."""
        k, F = len(cols_list), self.F
        assert 1 <= k <= self.GROUP and first_buf + k <= self.NBUF
        arr = (C.c_void_p * (k * F))(*[c.data_ptr() for cols in cols_list for c in cols])
        pl = self.plans[first_buf]
        check(lib.rec_colsort_plan_dest_i64(arr, k * F, self.B, self.V, _p(self.col_lo_rep), self.max_key,
                                            _p(pl["perm"]), _p(pl["col_uid"]), _p(pl["col_seg"]), _p(pl["col_nu"]),
                                            _p(pl["dloc"]), _p(self.bad_ids), _p(self.sort_ws),
                                            C.c_void_p(stream.cuda_stream)), "rec_colsort_plan_dest_i64")

    def _ploss(self):
        return C.c_void_p(self.loss_steps.data_ptr() + 4 * self._row)

    def _pprob(self):
        return C.c_void_p(self.prob_steps.data_ptr() + 4 * self.B * self._row) if self.prob_steps is not None else None

    def _launch_main(self, cols, label, st, buf):
        """The fused kernel.  Direct mode: the plan in buffer ``buf`` is complete, so the value row of every run's first
        member goes straight to its slot of g_embed_rows (with gz and the id)."""
        L, F = self.layer, self.F
        arr = (C.c_void_p * F)(*[c.data_ptr() for c in cols])
        emb = L.embed.embeddings
        pl = self.plans[buf]
        if self._fused_lazy() and self._last is not None:
            # Keras Adam, lazily: the rows this batch reads replay the sweeps they skipped (steps last+1 .. now)
            (me, ve), (mw, vw) = self.state["embed.embeddings"], self.state["w.embeddings"]
            check(lib.rec_adam_keras_catchup_f32(_p(pl["col_uid"]), _p(pl["col_nu"]), self.B, F, _p(emb), emb.stride(0),
                                                 self.V, _p(me), _p(ve), me.stride(0), _p(mw), _p(vw), mw.stride(0),
                                                 _p(self._last), _p(self._step_dev), _p(self._lr_tab),
                                                 self._lr_tab.numel(), 0.9, 0.999, 1e-7, st),
                  "rec_adam_keras_catchup_f32")
        if self.kernel_version >= 3:
            self._ensure_k0t(st)
            if not self.direct:
                check(lib.rec_deepfm_fused3_main_f32(
                    _p(emb), emb.stride(0), self.V, arr, F, self.B, _p(L.bias), _p(L.MLP_layer1.kernel_0), _p(self._k0t),
                    _p(L.MLP_layer1.bias_0), _p(L.MLP_layer1.kernel_1), _p(L.MLP_layer1.bias_1),
                    _p(L.MLP_layer2.kernel_0), _p(L.MLP_layer2.bias_0), _p(label), _p(self.gz), _p(self.vals),
                    self._pprob(), _p(self.oob), _p(self.ws), st), "rec_deepfm_fused3_main_f32")
                return
            head = (_p(emb), emb.stride(0), self.V, arr, F, self.B, _p(L.bias), _p(L.MLP_layer1.kernel_0), _p(self._k0t),
                    _p(L.MLP_layer1.bias_0), _p(L.MLP_layer1.kernel_1), _p(L.MLP_layer1.bias_1), _p(L.MLP_layer2.kernel_0),
                    _p(L.MLP_layer2.bias_0), _p(label), _p(self.gz), _p(self.vals), self._pprob(), _p(self.oob), _p(self.ws),
                    _p(pl["dloc"]), _p(pl["col_nu"]), _p(self.g_embed_rows))
            if self._fused_lazy():
                # the optimizer's device-side step counter advances inside this launch (the kernel reads neither word):
                # the catch-up above saw the old step, the post launch and the dense update below see the new one
                check(lib.rec_deepfm_fused3_main_direct_adv_f32(*head, _p(self._step_dev), _p(self._lr_tab),
                                                                self._lr_tab.numel(), _p(self._lr_t_dev), st),
                      "rec_deepfm_fused3_main_direct_adv_f32")
                self._advanced = True
            else:
                check(lib.rec_deepfm_fused3_main_direct_f32(*head, st), "rec_deepfm_fused3_main_direct_f32")
            return
        if not self.direct:
            check(lib.rec_deepfm_fused_main_f32(
                _p(emb), emb.stride(0), self.V, arr, F, self.B, _p(L.bias), _p(L.MLP_layer1.kernel_0),
                _p(L.MLP_layer1.bias_0), _p(L.MLP_layer1.kernel_1), _p(L.MLP_layer1.bias_1),
                _p(L.MLP_layer2.kernel_0), _p(L.MLP_layer2.bias_0), _p(label), _p(self.gz), _p(self.vals), None,
                _p(self.oob), _p(self.ws), st), "rec_deepfm_fused_main_f32")
            return
        check(lib.rec_deepfm_fused_main_direct_f32(
            _p(emb), emb.stride(0), self.V, arr, F, self.B, _p(L.bias), _p(L.MLP_layer1.kernel_0),
            _p(L.MLP_layer1.bias_0), _p(L.MLP_layer1.kernel_1), _p(L.MLP_layer1.bias_1), _p(L.MLP_layer2.kernel_0),
            _p(L.MLP_layer2.bias_0), _p(label), _p(self.gz), _p(self.vals), None, _p(self.oob), _p(self.ws),
            _p(pl["dloc"]), _p(pl["col_nu"]), _p(self.g_embed_rows), st), "rec_deepfm_fused_main_direct_f32")

    def _fused_lazy(self):
        """optimizer 'lazy_adam' / 'keras_adam_lazy' in direct mode: the touched-rows update of both tables rides in the
        post launch ('keras_adam_lazy': and the rows a batch is about to read first replay the dense sweeps they
        skipped -- Keras' Adam, bit for bit, without sweeping the table every step; flush() before reading parameters)"""
        return self.optimizer in ("lazy_adam", "keras_adam_lazy") and self.direct

    def _launch_post(self, buf, st, t=0):
        """reduction of the workgroup partials side by side with the segment sums (direct mode: with what is left of
        them -- runs of more than one lookup and the padded tail) in ONE launch"""
        g, pl = self.g, self.plans[buf]
        if self._fused_lazy():
            params = dict(self.layer.named_parameters())
            pe = params["embed.embeddings"]
            (me, ve), (mw, vw) = self.state["embed.embeddings"], self.state["w.embeddings"]
            if not self._advanced:                           # (the v3 fused launch has already advanced the counter)
                check(lib.rec_adam_advance_f32(_p(self._step_dev), _p(self._lr_tab), self._lr_tab.numel(),
                                               _p(self._lr_t_dev), st), "rec_adam_advance_f32")
            self._advanced = False
            check(lib.rec_deepfm_fused_post_direct_adam_dev_f32(
                self.F, self.B, _p(self.gz), _p(self.vals), _p(g["MLP_layer1.kernel_0"]), _p(g["MLP_layer1.bias_0"]),
                _p(g["MLP_layer1.kernel_1"]), _p(g["MLP_layer1.bias_1"]), _p(g["MLP_layer2.kernel_0"]),
                _p(g["MLP_layer2.bias_0"]), _p(g["bias"]), self._ploss(), _p(self.ws), _p(pl["perm"]),
                _p(pl["col_uid"]), _p(pl["col_seg"]), _p(pl["col_nu"]), _p(self.uniq_ids), _p(self.g_embed_rows),
                _p(self.g_w_rows), _p(self.n_uniq), _p(pe), pe.stride(0), self.V, _p(me), _p(ve), _p(mw), _p(vw),
                me.stride(0), mw.stride(0), _p(self._lr_t_dev), 0.9, 0.999, 1e-7,
                _p(self._last) if self._last is not None else None, _p(self._step_dev), st),
                "rec_deepfm_fused_post_direct_adam_dev_f32")
            return
        if not self.direct:
            check(lib.rec_deepfm_fused_post_f32(
                self.F, self.B, _p(self.gz), _p(self.vals), _p(g["MLP_layer1.kernel_0"]), _p(g["MLP_layer1.bias_0"]),
                _p(g["MLP_layer1.kernel_1"]), _p(g["MLP_layer1.bias_1"]), _p(g["MLP_layer2.kernel_0"]),
                _p(g["MLP_layer2.bias_0"]), _p(g["bias"]), self._ploss(), _p(self.ws), _p(pl["perm"]),
                _p(pl["col_uid"]), _p(pl["col_seg"]), _p(pl["col_nu"]), _p(self.uniq_ids), _p(self.g_embed_rows),
                _p(self.g_w_rows), _p(self.n_uniq), 0, st), "rec_deepfm_fused_post_f32")
            return
        check(lib.rec_deepfm_fused_post_direct_f32(
            self.F, self.B, _p(self.gz), _p(self.vals), _p(g["MLP_layer1.kernel_0"]), _p(g["MLP_layer1.bias_0"]),
            _p(g["MLP_layer1.kernel_1"]), _p(g["MLP_layer1.bias_1"]), _p(g["MLP_layer2.kernel_0"]),
            _p(g["MLP_layer2.bias_0"]), _p(g["bias"]), self._ploss(), _p(self.ws), _p(pl["perm"]), _p(pl["col_uid"]),
            _p(pl["col_seg"]), _p(pl["col_nu"]), _p(self.uniq_ids), _p(self.g_embed_rows), _p(self.g_w_rows),
            _p(self.n_uniq), st), "rec_deepfm_fused_post_direct_f32")

    def _optimizer(self, t, st):
        lr, b1, b2, eps = self.lr, 0.9, 0.999, 1e-7
        if self._fused_lazy():
            # the tables were updated inside the post launch; the dense parameters follow in ONE launch, with the
            # step size the post launch used (device memory)
            k, var, m, v, g, numel = self._multi
            check(lib.rec_adam_dense_multi_f32(k, var, m, v, g, numel, _p(self._lr_t_dev), b1, b2, eps, st),
                  "rec_adam_dense_multi_f32")
            self._retranspose(st)
            return
        params = dict(self.layer.named_parameters())
        for name, grad in self.g.items():
            m, v = self.state[name]
            check(lib.rec_adam_dense_f32(_p(params[name]), _p(m), _p(v), _p(grad), grad.numel(), t, lr, b1, b2, eps, st),
                  "rec_adam_dense_f32")
        self._retranspose(st)
        n = self.B * self.F
        pe, pw = params["embed.embeddings"], params["w.embeddings"]
        if self.optimizer == "keras_adam" and _tables_share_rows(pe, pw):
            # one sweep over the fused [embed | w | pad] rows instead of one per table
            (me, ve), (mw, vw) = self.state["embed.embeddings"], self.state["w.embeddings"]
            check(lib.rec_adam_sparse_keras_pair_f32(_p(pe), pe.stride(0), _p(me), _p(ve), _p(mw), _p(vw), self.V, 16,
                                                     _p(self.uniq_ids), _p(self.g_embed_rows), _p(self.g_w_rows),
                                                     _p(self.n_uniq), n, _p(self.side_e), _p(self.side_w), t, lr, b1,
                                                     b2, eps, st), "rec_adam_sparse_keras_pair_f32")
            return
        for name, rows, side, E in (("embed.embeddings", self.g_embed_rows, self.side_e, 16),
                                    ("w.embeddings", self.g_w_rows, self.side_w, 1)):
            m, v = self.state[name]
            p = params[name]
            if self.optimizer == "keras_adam":
                check(lib.rec_adam_sparse_keras_f32(_p(p), p.stride(0), _p(m), _p(v), self.V, E, _p(self.uniq_ids),
                                                    _p(rows), _p(self.n_uniq), n, _p(side), t, lr, b1, b2, eps, st),
                      "rec_adam_sparse_keras_f32")
            else:
                check(lib.rec_adam_rows_f32(_p(p), p.stride(0), _p(m), _p(v), self.V, E, _p(self.uniq_ids), _p(rows),
                                            _p(self.n_uniq), n, t, lr, b1, b2, eps, st), "rec_adam_rows_f32")

    def _cols(self, inputs):
        cols = []
        for name in self.layer.feature_names:
            c = inputs[name]
            if c.dtype != torch.int64 or not c.is_cuda or c.numel() != self.B or not c.is_contiguous():
                raise ValueError("feature %r must be a contiguous int64 CUDA tensor with %d ids" % (name, self.B))
            cols.append(c)
        return cols

    def __call__(self, inputs, label_name="label", next_inputs=None):
        """One train_loop iteration on `inputs`.  ``next_inputs`` (optional) = the batch of the NEXT call: its
        de-duplication plan is built behind this call's step (input-pipeline style prefetch: the plan depends on ids
        only).  Without it, or when the previous call did not announce this batch,
        the plan is built inside this call, in front of the fused kernel."""
        return self.many([inputs], label_name, then=next_inputs)

    def _key(self, cols):
        return tuple(c.data_ptr() for c in cols)

    def _cols_key(self, inputs):
        """(columns, their addresses) of a batch.  The 26 dtype / device / size / stride checks and address reads of a
        batch were half of the host time of a call (which the GPU waits for whenever a call starts from an empty
        queue): done once per batch dict, and again only when a tensor of the dict was replaced (identity of the 26
        tensor objects, compared in C: itemgetter + map(id))."""
        ent = self._col_cache.get(id(inputs))
        if ent is not None and ent[2] is inputs and tuple(map(id, self._getter(inputs))) == ent[3]:
            return ent[0], ent[1]
        cols = self._cols(inputs)
        if len(self._col_cache) > 256:
            self._col_cache.clear()
        key = self._key(cols)
        self._col_cache[id(inputs)] = (cols, key, inputs, tuple(map(id, cols)))
        return cols, key

    def many(self, batches, label_name="label", then=None):
        """len(batches) consecutive train_loop iterations; with ``use_graph`` as ONE hipGraph replay (a launch-bound
        inner loop: one graph launch costs ~20 us of idle GPU).  ``then``: the batch, or the list of batches, of the NEXT
        call -- their de-duplication plans are built behind this call's steps, several batches per sort launch, so that
        no fused kernel of the next call waits for a sort (the plan has to be complete before the kernel starts: direct
        mode).
        A batch of this call that no earlier call announced is sorted in line.  Results left in the buffers are the last
        step's; gradients are to be consumed by an optimizer in the same call or after single-step calls."""
        half = self.NBUF // 2
        if then is None:
            then_list = []
        elif isinstance(then, dict):
            then_list = [then]
        else:
            then_list = list(then)
        if len(batches) > half or len(then_list) > half:
            raise ValueError("many(): at most %d batches per call (and per announcement)" % half)
        seq, keys = [], []
        for b in batches:
            cols, key = self._cols_key(b)
            y = b[label_name]
            if y.dtype != torch.float32 or not y.is_cuda or y.numel() != self.B or not y.is_contiguous():
                raise ValueError("label must be a contiguous float32 CUDA tensor with %d entries" % self.B)
            seq.append((cols, y))
            keys.append(key)
        then_cols, then_keys = [], []
        for b in then_list:
            cols, key = self._cols_key(b)
            then_cols.append(cols)
            then_keys.append(key)
        n = len(seq)
        # plan buffers: the ring has two halves.  Plans announced by the previous call live in half `cur_half`; batches
        # of this call that were not announced take the free slots of the same half; the announced batches of the next
        # call go to the other half
        cur_half = self._half
        pre = self._prefetched
        used = set(pre[k] for k in keys if k in pre)
        free = [cur_half * half + j for j in range(half) if cur_half * half + j not in used]
        bufs, inline = [], []
        for i, k in enumerate(keys):
            if k in pre and pre[k] not in bufs:
                bufs.append(pre[k])
            else:
                bufs.append(free.pop(0))
                inline.append(i)
        other = (1 - cur_half) * half
        then_bufs = [other + j for j in range(len(then_cols))]
        graphed = self.use_graph and (self.optimizer is None or self._fused_lazy())
        gkey = (tuple(keys), tuple(y.data_ptr() for _, y in seq), tuple(then_keys), tuple(bufs), tuple(inline))

        def enqueue_all():
            main = torch.cuda.current_stream()
            st = C.c_void_p(main.cuda_stream)
            for i in inline:                                     # not announced: sorted in line
                self._sort(seq[i][0], bufs[i], main)
            self._row = 0
            self._launch_main(seq[0][0], seq[0][1], st, bufs[0])
            for i in range(n):
                self._row = i
                if i > 0:
                    self._launch_main(seq[i][0], seq[i][1], st, bufs[i])
                t = t_base + i + 1                               # 1-based step (host scalar of the non-graphed optimizers)
                self._launch_post(bufs[i], st, t)
                if self.optimizer is not None:
                    self._optimizer(t, st)
            # the plans of the batches announced for the NEXT call: GROUP batches (consecutive buffers) per sort launch, on the
            # main stream BEHIND this call's steps.  (They used to run on a second stream beside the steps -- but a sort
            # workgroup cannot share a CU with a fused-kernel workgroup (LDS), so "beside" meant that one fused launch in
            # eight waited for the sort: serial on one stream is 0.7 us per step faster at K = 200, the same at K = 20, and
            # the graph has no fork / join.)
            m_ = len(then_cols)
            if m_:
                nl = -(-m_ // self.GROUP)                        # as few launches as the column limit allows, evenly filled
                per = -(-m_ // nl)
                for j in range(0, m_, per):
                    self._sort_group(then_cols[j:j + per], then_bufs[j], main)

        t_base = self.t
        if self.kernel_version >= 3:
            self._ensure_k0t()                                   # outside any capture: a replayed graph reads K0T
        if not graphed:
            enqueue_all()
        else:
            ent = self._graphs.get(gkey)
            if ent is not None:
                self._graphs.move_to_end(gkey)
                ent[0].replay()
            elif gkey not in self._seen:
                # first sighting of these addresses: plain eager enqueue, no device synchronisation, nothing retained.  An
                # input pipeline that hands over fresh tensors every batch never gets past this branch (its steps run
                # eagerly, ~0.15 ms of host time each); one that cycles staging buffers is captured on the next round
                enqueue_all()
                self._seen[gkey] = True
                if len(self._seen) > 8 * self.MAX_GRAPHS:
                    self._seen.popitem(last=False)
            else:
                # second sighting: enqueued eagerly -- that IS this call's work -- and then captured, without running, for
                # the calls to come.  (With an optimizer in the step a warm-up followed by a replay would apply the update
                # twice.)
                enqueue_all()
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                    enqueue_all()
                self._graphs[gkey] = (g, seq, then_cols)         # the inputs stay alive as long as the graph does
                del self._seen[gkey]
                while len(self._graphs) > self.MAX_GRAPHS:
                    self._graphs.popitem(last=False)             # least recently used: graph and retained inputs go
        self.t = t_base + n
        self.loss = self.loss_steps[n - 1:n]                     # the last step's
        if self.prob_steps is not None:
            self.prob = self.prob_steps[n - 1]
        self._prefetched = dict(zip(then_keys, then_bufs))
        self._half = 1 - cur_half if then_cols else cur_half
        return self.loss

    def flush(self):
        """optimizer 'keras_adam_lazy': bring EVERY row of the tables up to the current step (the dense sweeps the rows
        skipped) -- before the parameters are read from outside the step (evaluation, checkpoint)."""
        if not (self._fused_lazy() and self._last is not None):
            return
        emb = self.layer.embed.embeddings
        (me, ve), (mw, vw) = self.state["embed.embeddings"], self.state["w.embeddings"]
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.rec_adam_keras_flush_f32(_p(emb), emb.stride(0), self.V, _p(me), _p(ve), me.stride(0), _p(mw), _p(vw),
                                           mw.stride(0), _p(self._last), _p(self._step_dev), _p(self._lr_tab),
                                           self._lr_tab.numel(), 0.9, 0.999, 1e-7, st), "rec_adam_keras_flush_f32")

    def release(self):
        """Give up the optimizer state kept in the layer's table padding (lazy optimizers) and the captured graphs, so
        that another step may be built on the same layer."""
        owner = getattr(self.layer, "_fused_state_owner", None)
        if owner is not None and owner() is self:
            self.layer._fused_state_owner = None
        self._graphs.clear()
        self._seen.clear()

    def check_flags(self):
        if int(self.oob.item()) != 0:
            raise IndexError("embedding id out of range [0, feature_dims)")
        if int(self.bad_ids.item()) != 0:
            raise ValueError("an id lies outside its field's [offset, offset+dim) range (DataGenerator contract)")

    def gradients(self):
        out = dict(self.g)
        out["embed.embeddings"] = (self.uniq_ids, self.g_embed_rows, self.n_uniq)
        out["w.embeddings"] = (self.uniq_ids, self.g_w_rows, self.n_uniq)
        return out


def exchange_capacity(field_dims, field_offsets, batch_size, rows_per_shard, n_shard):
    """Slots per owner of a fixed-capacity exchange: the most unique ids ONE batch can hold for one owner =
    max over owners of the sum over the fields that intersect the owner's block of min(B, overlap) (a field contributes
    at most one id per example).  Rounded up to a multiple of 8."""
    cap = 1
    for o in range(n_shard):
        lo, hi = o * rows_per_shard, (o + 1) * rows_per_shard
        c = 0
        for d, off in zip(field_dims, field_offsets):
            ov = min(hi, int(off) + int(d)) - max(lo, int(off))
            if ov > 0:
                c += min(int(batch_size), ov)
        cap = max(cap, c)
    return (cap + 7) // 8 * 8


class HipStepBackend:
    """Device-side pieces of ShardedDeepFMStep, all HIP kernels on preallocated buffers of constant size.  The CPU gloo
    test (tests/test_sharded.py) injects an oracle-backed stand-in with the same methods to exercise the exchange logic
    without a GPU; the product never does.

    The host issues ~10 C-ABI calls and 4 collectives per step and must stay ahead of the GPU, so nothing here looks up
    torch's current stream or builds pointer arrays more than once: `begin()` caches the stream handles of a step, the
    per-plan pointer arrays are built at construction, nothing is allocated per step and nothing is read back."""

    def __init__(self, step, field_dims, field_offsets):
        self.step = step
        B, F, P, cap = step.B, step.F, step.P, step.cap
        dev = step.dev
        n = B * F
        self.max_key = max(int(d) for d in field_dims) - 1
        if _bits(self.max_key + 1) + _bits(B) > 32 or ((self.max_key << _bits(B)) | (B - 1)) >= 0xFFFFFFFF:
            raise NotImplementedError("field too wide for the 32-bit sort words at this batch size")
        if any(field_offsets[i] >= field_offsets[i + 1] for i in range(F - 1)):
            raise ValueError("field offsets must be ascending (DataGenerator contract)")
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        i64 = dict(dtype=torch.int64, device=dev)
        m = P * cap                                        # rows of every exchange buffer
        self.col_lo = torch.tensor([int(o) for o in field_offsets], **i64)
        self.bad_ids = torch.zeros(1, **i32)
        self.o_ws_bytes = lib.rec_dedup_workspace_bytes(m)
        # NPL plan buffers.  The eager step alternates between the first two (the plan of batch k+1 is built beside step
        # k); many() gives every batch of its cycle a buffer of its own and sorts GROUP batches per launch -- the sort
        # arrays of consecutive buffers are contiguous, so ONE rec_colsort_plan_i64 call over GROUP*F columns fills them
        # (a sort workgroup cannot share a CU with a fused-kernel workgroup -- LDS --, so a 32-us sort per step was 32 us
        # on the step's critical path; eight batches per launch cost ~36)
        self.NPL = NPL = 16
        self.GROUP = max(1, min(8, 256 // F))
        self.col_lo_rep = self.col_lo.repeat(self.GROUP).contiguous()
        self._perm = torch.empty((NPL, F, B), **i32)
        self._col_uid = torch.empty((NPL, F, B), **i64)
        self._col_seg = torch.empty((NPL, F, B + 1), **i32)
        self._col_nu = torch.zeros((NPL, F), **i32)
        self.plans = [dict(perm=self._perm[b], col_uid=self._col_uid[b], col_seg=self._col_seg[b], col_nu=self._col_nu[b],
                           msg=torch.zeros((P, cap + 2), **i64), msg_theirs=torch.zeros((P, cap + 2), **i64),
                           uidx=torch.empty((F, B), **i64), slot_map=torch.zeros(n, **i32), n_uniq=torch.zeros(1, **i64),
                           # owner side: union of the P lists that arrive (depends on ids only: built with the plan)
                           o_uniq=torch.empty(m, **i64), o_seg=torch.empty(m + 1, **i32), o_perm=torch.empty(m, **i32),
                           o_nu=torch.zeros(1, **i64)) for b in range(NPL)]
        for pl in self.plans:
            pl["uidx_arr"] = (C.c_void_p * F)(*[pl["uidx"][f].data_ptr() for f in range(F)])
        self.sort_ws = [torch.empty(lib.rec_colsort_workspace_bytes(B, F * self.GROUP), dtype=torch.uint8, device=dev)
                        for _ in range(NPL)]
        self.o_ws = [torch.empty(self.o_ws_bytes, dtype=torch.uint8, device=dev) for _ in range(NPL)]
        # every pointer below is fixed for the life of the step: the ctypes argument tuples are built once
        for buf, pl in enumerate(self.plans):
            pl["a_sort"] = (_p(self.col_lo), self.max_key, _p(pl["perm"]), _p(pl["col_uid"]), _p(pl["col_seg"]),
                            _p(pl["col_nu"]), _p(self.bad_ids), _p(self.sort_ws[buf]))
            pl["a_map"] = (_p(pl["perm"]), _p(pl["col_uid"]), _p(pl["col_seg"]), _p(pl["col_nu"]), B, F,
                           step.rows_per_shard, P, cap, _p(pl["msg"]), _p(pl["uidx"]), _p(pl["slot_map"]),
                           _p(pl["n_uniq"]), _p(step.oob))
            pl["a_owner"] = (P, cap, step.rows_per_shard, _p(pl["o_uniq"]), _p(pl["o_seg"]), _p(pl["o_perm"]),
                             _p(pl["o_nu"]), _p(self.o_ws[buf]), self.o_ws_bytes)
            pl["a_plan4"] = (_p(pl["perm"]), _p(pl["col_uid"]), _p(pl["col_seg"]), _p(pl["col_nu"]), _p(pl["slot_map"]))
        self.gz = torch.empty(B, **f32)
        self.vals = torch.empty((n, 16), **f32)
        # rows travel as [embed 16 | w | pad 3] = 80 bytes, not as the 128-byte lines they are stored in (C2 -37 %)
        self.rows_out = torch.zeros((m, 20), **f32)        # owner-side gather result           -> C2
        self.rows_local = torch.zeros((m, 20), **f32)      # C2 ->  the batch's rows, [owner, slot]
        self.grows = torch.zeros((m, 20), **f32)           # [embed 16 | w | pad 3] per unique id -> C3
        self.rows_theirs = torch.zeros((m, 20), **f32)     # C3 ->
        self.o_rows = torch.empty((m, 20), **f32)
        self.o_sws = torch.empty(lib.rec_segment_sum_workspace_bytes(m, 20), dtype=torch.uint8, device=dev)
        self.ws = torch.empty(lib.rec_deepfm_fused_workspace_bytes(B, F), dtype=torch.uint8, device=dev)
        self.side = torch.cuda.Stream(device=dev)
        self.side_ctx = torch.cuda.stream(self.side)
        self.side_st = C.c_void_p(self.side.cuda_stream)
        self.main, self.st = None, None
        self._arr_cache = {}

    # -- streams: the plan of the next batch depends on ids only and is built beside the current step
    def begin(self):
        self.main = torch.cuda.current_stream()
        self.st = C.c_void_p(self.main.cuda_stream)

    def fork(self):
        self.side.wait_stream(self.main)

    def side_context(self):
        return self.side_ctx

    def join(self):
        self.main.wait_stream(self.side)

    def _col_arr(self, cols):
        key = tuple(c.data_ptr() for c in cols)
        arr = self._arr_cache.get(key)
        if arr is None:
            if len(self._arr_cache) > 64:
                self._arr_cache.clear()
            arr = self._arr_cache[key] = (C.c_void_p * len(cols))(*key)
        return arr

    def plan(self, cols, buf, on_side=False):
        """Per-column sort + unique (rec_colsort_plan_i64), then the fixed-capacity exchange map
        (rec_colsort_shard_map_fixed_i64): the id message, the slot of every lookup and of every unique id."""
        st_ = self.step
        B, F = st_.B, st_.F
        pl = self.plans[buf]
        st = self.side_st if on_side else self.st
        check(lib.rec_colsort_plan_i64(self._col_arr(cols), F, B, st_.V, *pl["a_sort"], st), "rec_colsort_plan_i64")
        check(lib.rec_colsort_shard_map_fixed_i64(*pl["a_map"], st), "rec_colsort_shard_map_fixed_i64")
        return pl

    def plan_group(self, cols_list, first_buf, on_side=False):
        """The plans of len(cols_list) <= GROUP batches into the consecutive buffers first_buf, first_buf + 1, ...: ONE
        per-column sort over all their columns, then the exchange map of each."""
        st_ = self.step
        B, F, k = st_.B, st_.F, len(cols_list)
        assert 1 <= k <= self.GROUP and first_buf + k <= self.NPL
        st = self.side_st if on_side else self.st
        arr = (C.c_void_p * (k * F))(*[c.data_ptr() for cols in cols_list for c in cols])
        pl = self.plans[first_buf]
        check(lib.rec_colsort_plan_i64(arr, k * F, B, st_.V, _p(self.col_lo_rep), self.max_key, _p(pl["perm"]),
                                       _p(pl["col_uid"]), _p(pl["col_seg"]), _p(pl["col_nu"]), _p(self.bad_ids),
                                       _p(self.sort_ws[first_buf]), st), "rec_colsort_plan_i64")
        for j in range(k):
            check(lib.rec_colsort_shard_map_fixed_i64(*self.plans[first_buf + j]["a_map"], st),
                  "rec_colsort_shard_map_fixed_i64")

    def owner_plan(self, pl, buf, on_side=False):
        """Union of the P ascending id lists that arrived (rank merge) as a segment plan over the payload rows."""
        st_ = self.step
        st = self.side_st if on_side else self.st
        check(lib.rec_dedup_plan_sorted_slabs_i64(_p(pl["msg_theirs"]), *pl["a_owner"], st),
              "rec_dedup_plan_sorted_slabs_i64")

    def gather(self, table, pl):
        """Owner-side gather of the 128-byte fused rows for the ids every rank asked for."""
        st_ = self.step
        check(lib.rec_emb_gather_lists_f32(_p(table), table.shape[0], 20, 32, _p(pl["msg_theirs"]), st_.P, st_.cap,
                                           _p(self.rows_out), _p(st_.oob), self.st), "rec_emb_gather_lists_f32")
        return self.rows_out

    def rows_step(self, pl, rows_local, y):
        """The fused forward+backward kernel on the exchanged rows: the local [P*cap,32] buffer is the "table" and
        the ids are the slots uidx.  Returns (vals [n,16], gz [B]); the dense gradients follow in local_grad (the
        reduction shares its launch with the segment sums)."""
        st_ = self.step
        w = self.__dict__.get("_a_weights")
        L = st_.layer
        if w is None:                                        # parameters are updated in place: their addresses stay
            self._k0t = torch.empty((32, st_.F * 16), dtype=torch.float32, device=st_.dev)
            self._k0_ver = None
            w = self._a_weights = (_p(L.bias), _p(L.MLP_layer1.kernel_0), _p(self._k0t), _p(L.MLP_layer1.bias_0),
                                   _p(L.MLP_layer1.kernel_1), _p(L.MLP_layer1.bias_1), _p(L.MLP_layer2.kernel_0),
                                   _p(L.MLP_layer2.bias_0))
            self._a_tail = (_p(self.gz), _p(self.vals), None, _p(st_.oob), _p(self.ws))
        K0 = L.MLP_layer1.kernel_0
        ver = (K0._version, K0.data_ptr())
        if ver != self._k0_ver and st_.F > 26:               # layer 1's kernel changed: refresh its transposed copy (only
            check(lib.rec_deepfm_k0t_f32(_p(K0), st_.F, _p(self._k0t), self.st), "rec_deepfm_k0t_f32")   # read when K0 does
            self._k0_ver = ver                               # not fit in LDS beside the rows)
        check(lib.rec_deepfm_fused3_main_f32(_p(rows_local), 20, rows_local.shape[0], pl["uidx_arr"], st_.F, st_.B, *w,
                                             _p(y), *self._a_tail, self.st), "rec_deepfm_fused3_main_f32")
        return self.vals, self.gz

    def local_grad(self, pl, vals, gz):
        """Reduction of the workgroup partials (fills step.g / step.loss) and, in the same launch, this batch's
        gradient per unique id as rows [embed 16 | w | 0 0 0] in the id's slot of the [P*cap,20] send buffer."""
        st_ = self.step
        a = self.__dict__.get("_a_post")
        if a is None:
            g = st_.g
            a = self._a_post = (_p(g["MLP_layer1.kernel_0"]), _p(g["MLP_layer1.bias_0"]), _p(g["MLP_layer1.kernel_1"]),
                                _p(g["MLP_layer1.bias_1"]), _p(g["MLP_layer2.kernel_0"]), _p(g["MLP_layer2.bias_0"]),
                                _p(g["bias"]), _p(st_.loss), _p(self.ws))
        check(lib.rec_deepfm_fused_post_slots_f32(st_.F, st_.B, _p(gz), _p(vals), *a, *pl["a_plan4"], _p(self.grows),
                                                  self.st), "rec_deepfm_fused_post_slots_f32")
        return self.grows

    def owner_reduce(self, pl, rows_theirs, scale):
        """Row sums in the order of the owner plan, times ``scale``.  Returns (uniq local ids, embed rows [.,16], w rows
        [.,1] -- views of one [P*cap,20] buffer that the next step overwrites --, n_uniq); entries past n_uniq are
        padding (a valid id, zero rows)."""
        m = rows_theirs.shape[0]
        st = self.st
        check(lib.rec_segment_sum_f32(_p(rows_theirs), 20, _p(pl["o_perm"]), _p(pl["o_seg"]), m, 1, _p(self.o_rows),
                                      _p(self.o_sws), st), "rec_segment_sum_f32")
        rows = self.o_rows
        if scale != 1.0:
            check(lib.rec_axpby_f32(scale, _p(rows), 0.0, _p(rows), rows.numel(), st), "rec_axpby_f32")
        return pl["o_uniq"], rows[:, :16], rows[:, 16:17], pl["o_nu"]

    def check_flags(self):
        if int(self.bad_ids.item()) != 0:
            raise ValueError("an id lies outside its field's [offset, offset+dim) range (DataGenerator contract)")


class ShardedDeepFMStep:
    """DeepFM train_loop iteration with the fused [embed|w|pad] table ROW-SHARDED over the ranks of a process group
    (SURVEY.md section 8e): data-parallel batch (every rank its own B examples), block partition
    ``owner = id // ceil(V/P)``.  De-duplicate first, then exchange -- in FIXED-CAPACITY slabs: the field layout bounds
    the unique ids one batch can hold for one owner (exchange_capacity), so every exchange has constant split sizes:
    no count exchange, nothing read back by the host, every buffer allocated once.

        plan (ids only; built for batch k+1 on a second stream while batch k is differentiated)
            per-column sort + unique  ->  the batch's unique ids, ascending = already grouped by owner; every owner's
            ids go to its slab of the id message [P, 2+cap] (word 0 = how many)
            C1  all-to-all of the id messages on its OWN communicator (own RCCL stream: it does not queue behind
                the payload collectives), then the owner's union of the P lists that arrived (rank merge)
        --  owner-side gather of the fused rows' 80 used bytes into [P, cap, 20] (HIP, ids ascending per list)
        C2  all-to-all of the rows back -> a local [P*cap, 20] table, row = owner*cap + slot: no permutation anywhere
        --  the fused forward+backward kernel on those rows (it gathers by the slot of each lookup), then the
            per-unique-id segment sums of the row gradients (embed 16 + w 1) written to the ids' slots
        C3  all-to-all of the summed row gradients ([P*cap,20]) to the owners, who add them in the order of the union
        C4  one flat all-reduce (SUM) of the dense gradients and the loss, divided by P

    The loss is the mean over the GLOBAL batch of P*B examples (2.FM/ModelManager.py:171-177 on the concatenated
    batch): every rank's kernel scales by 1/B, so dense gradients are averaged over ranks and owner-side row sums
    are multiplied by 1/P.  Same kernels as the single-GPU path; at world_size 1 it reproduces DeepFMFusedStep and at
    world_size 2 (gloo) the global-batch result (tests/test_sharded.py).
    ``table_shard`` [rows_per_shard, 32] holds global rows [rank*rows_per_shard, ...).
    """

    def __init__(self, layer, batch_size, field_dims, field_offsets, group=None, backend=None, comm=None):
        from . import sharded
        self.comm = comm if comm is not None else sharded.DistComm(group, separate_count_channel=True)
        self.P, self.rank = self.comm.world, self.comm.rank
        self.layer = layer
        self.B = B = int(batch_size)
        self.F = F = len(layer.feature_names)
        emb, w = layer.embed.embeddings, layer.w.embeddings
        V, E = emb.shape
        if E != 16 or list(layer.mlp_dims) != [32, 8]:
            raise NotImplementedError("sharded step: embedding_dims=16, mlp_dims=[32,8]")
        if F > 28 or B > 16384 or len(field_dims) != F or len(field_offsets) != F:
            raise NotImplementedError("sharded step: F <= 28, B <= 16384, one (dim, offset) per feature")
        fused = getattr(layer, "_fused_storage", None)      # [V,32] rows = [embed | w | pad] (layers._FMTables)
        if fused is None:                                   # layer not on the GPU (CPU exchange-logic test)
            fused = torch.zeros((V, 32), dtype=torch.float32, device=emb.device)
            fused[:, :16].copy_(emb.data)
            fused[:, 16:17].copy_(w.data)
        self.V = V
        self.rows_per_shard = -(-V // self.P)
        self.cap = exchange_capacity(field_dims, field_offsets, B, self.rows_per_shard, self.P)
        lo = min(V, self.rank * self.rows_per_shard)
        hi = min(V, lo + self.rows_per_shard)
        self.row_range = (lo, hi)
        self.dev = dev = emb.device
        f32 = dict(dtype=torch.float32, device=dev)
        # this rank's block of the fused table (a real deployment would never hold the full table anywhere)
        self.table_shard = torch.zeros((self.rows_per_shard, 32), **f32)
        self.table_shard[: hi - lo].copy_(fused[lo:hi])
        self.n = B * F
        self.oob = torch.zeros(1, dtype=torch.int32, device=dev)
        D = F * 16
        # the dense gradients and the loss are views of ONE flat buffer: C4 is a single in-place all-reduce, no
        # concatenation before it and no copies after it (every view starts on a 16-byte boundary)
        shapes = [("MLP_layer1.kernel_0", (D, 32)), ("MLP_layer1.bias_0", (32,)), ("MLP_layer1.kernel_1", (32, 8)),
                  ("MLP_layer1.bias_1", (8,)), ("MLP_layer2.kernel_0", (8, 1)), ("MLP_layer2.bias_0", (1,)),
                  ("bias", (1,)), ("loss", (1,))]
        offs, total = [], 0
        for _, shp in shapes:
            offs.append(total)
            k = 1
            for d in shp:
                k *= d
            total += (k + 3) // 4 * 4
        self.flat = torch.zeros(total, **f32)
        views = {}
        for (name, shp), off in zip(shapes, offs):
            k = 1
            for d in shp:
                k *= d
            views[name] = self.flat[off:off + k].view(shp)
        self.loss = views.pop("loss")
        self.g = views
        self.be = (backend or HipStepBackend)(self, field_dims, field_offsets)
        self._next = None               # (key, buffer, plan) announced by the previous call
        self.table_grad = None          # (local uniq ids, embed rows [.,16], w rows [.,1], n_uniq) after a step

    def _cols(self, inputs):
        cols = []
        on_gpu = isinstance(self.be, HipStepBackend)          # the kernels take raw device pointers
        for name in self.layer.feature_names:
            c = inputs[name]
            if c.dtype != torch.int64 or c.numel() != self.B or not c.is_contiguous() or (on_gpu and not c.is_cuda):
                raise ValueError("feature %r must be a contiguous int64 %stensor with %d ids"
                                 % (name, "CUDA " if on_gpu else "", self.B))
            cols.append(c)
        return cols

    def _label(self, inputs, label_name):
        y = inputs[label_name]
        on_gpu = isinstance(self.be, HipStepBackend)
        if y.dtype != torch.float32 or y.numel() != self.B or not y.is_contiguous() or (on_gpu and not y.is_cuda):
            raise ValueError("label must be a contiguous float32 %stensor with %d entries" % ("CUDA " if on_gpu else "", self.B))
        return y

    def _cols_key(self, inputs):
        """(columns, their addresses) of a batch; validated once per batch dict (the host is the bottleneck of the eager
        step: 26 dtype / size / stride checks per call were a tenth of it) and again whenever a tensor of the dict was
        replaced (identity of the tensor objects, as in DeepFMFusedStep._cols_key)."""
        cache = self.__dict__.setdefault("_col_cache", {})
        getter = self.__dict__.get("_getter")
        if getter is None:
            import operator
            names_ = list(self.layer.feature_names)
            getter = self._getter = (operator.itemgetter(*names_) if len(names_) > 1
                                     else (lambda d, n=names_[0]: (d[n],)))
        ent = cache.get(id(inputs))
        if ent is None or ent[2] is not inputs or tuple(map(id, getter(inputs))) != ent[3]:
            cols = self._cols(inputs)
            if len(cache) > 256:
                cache.clear()
            ent = cache[id(inputs)] = (cols, tuple(c.data_ptr() for c in cols), inputs, tuple(map(id, cols)))
        return ent[0], ent[1]

    def _plan(self, cols, buf, on_side):
        """Plan + C1 + the owner's union of what arrived: everything that depends on the ids alone."""
        pl = self.be.plan(cols, buf, on_side=on_side)
        pl["msg_theirs"] = self.comm.exchange_ids(pl["msg"], pl["msg_theirs"])
        self.be.owner_plan(pl, buf, on_side=on_side)
        return pl

    def __call__(self, inputs, label_name="label", next_inputs=None):
        be, comm = self.be, self.comm
        be.begin()
        cols, key = self._cols_key(inputs)
        y = self._label(inputs, label_name)
        if self._next is not None and self._next[0] == key:
            _, buf, pl = self._next
            be.join()                                        # the plan was built on the second stream
        else:
            buf = 0
            pl = self._plan(cols, buf, False)
        self._next = None
        if next_inputs is not None:
            next_cols, next_key = self._cols_key(next_inputs)
            be.fork()
            with be.side_context():                          # C1 of the next batch: own communicator, second stream
                nxt = self._plan(next_cols, 1 - buf, True)
            self._next = (next_key, 1 - buf, nxt)
        return self._body(pl, y)

    def _body(self, pl, y):
        """What of a step needs the batch's plan to be complete: owner gather, C2, fused kernel, local sums, C3, owner sums,
        C4."""
        be, comm = self.be, self.comm
        rows_out = be.gather(self.table_shard, pl)                             # owner-side gather of 128-B rows
        rows_local = comm.exchange(rows_out, be.rows_local)                    # C2: [P*cap, 32], row = owner*cap + slot
        vals, gz = be.rows_step(pl, rows_local, y)
        grows = be.local_grad(pl, vals, gz)
        rows_theirs = comm.exchange(grows, be.rows_theirs)                     # C3
        self.table_grad = be.owner_reduce(pl, rows_theirs, 1.0 / self.P)
        # C4: dense gradients and the loss, one flat all-reduce (mean over ranks = the global-batch gradient)
        if self.P > 1:
            comm.all_reduce_sum(self.flat)
            self.flat /= self.P
        return self.loss

    def many(self, batches, label_name="label"):
        """A cycle of steps over resident batches as ONE captured hipGraph (RCCL collectives included: every exchange
        has constant sizes and nothing is read back, so the sequence is a fixed program).  The plan of the first batch
        is built inside the graph on the main stream, the plans of the following ones on the second stream beside the
        step before.  Every rank must call it with the same number of batches.  Returns the loss of the last step."""
        key = (tuple(self._cols_key(b)[1] for b in batches) +
               tuple(self._label(b, label_name).data_ptr() for b in batches))     # every column and label address
        graphs = self.__dict__.setdefault("_graphs", {})
        g = graphs.get(key)
        if g is None:
            grouped = isinstance(self.be, HipStepBackend) and len(batches) <= self.be.NPL

            def run():
                self._next = None
                if not grouped:
                    for i, b in enumerate(batches):
                        self(b, label_name, next_inputs=batches[i + 1] if i + 1 < len(batches) else None)
                    return
                # every plan of the cycle on the second stream, GROUP batches per sort launch (ids only: nothing of the
                # steps is needed), each followed by its C1 and the owner's merge; step i waits for plan i alone
                be, comm = self.be, self.comm
                be.begin()
                colss = [self._cols_key(b)[0] for b in batches]
                ys = [self._label(b, label_name) for b in batches]
                be.fork()
                evs = []
                with be.side_context():
                    for j0 in range(0, len(batches), be.GROUP):
                        be.plan_group(colss[j0:j0 + be.GROUP], j0, on_side=True)
                    for i in range(len(batches)):
                        pl = be.plans[i]
                        pl["msg_theirs"] = comm.exchange_ids(pl["msg"], pl["msg_theirs"])
                        be.owner_plan(pl, i, on_side=True)
                        ev = torch.cuda.Event()
                        ev.record(be.side)
                        evs.append(ev)
                for i in range(len(batches)):
                    be.main.wait_event(evs[i])
                    self._body(be.plans[i], ys[i])
                be.join()
            run()                                            # communicators and lazy initialisation: not captured
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                run()
            graphs[key] = g
        g.replay()
        return self.loss

    def release_graphs(self):
        """Drop the captured graphs (before the process group is destroyed: they hold RCCL kernels)."""
        self.__dict__.pop("_graphs", None)

    def check_flags(self):
        if int(self.oob.item()) != 0:
            raise IndexError("embedding id out of range [0, feature_dims), or more unique ids for one owner than the "
                             "exchange capacity")
        self.be.check_flags()


class GraphedTrainStep:
    """Any layer of layers.py under the reference's loss as ONE replayed hipGraph: forward, KerasBCE and the whole
    autograd backward (every kernel goes through the C ABI on torch's current stream, nothing synchronises, all
    intermediates come from the graph's private pool), for the families that have no hand-fused step (DSSM towers, DCN,
    DIN).  The eager path costs ~0.5-1 ms of Python and launch overhead per iteration whatever the batch; a replay
    costs one graph launch.

        step = GraphedTrainStep(layer, example_batch, label_name="label")
        loss = step(batch)            # copies the batch into the static input buffers, replays; .grad of every parameter
                                      # is a tensor of the graph (dense, or sparse COO rows for tables) valid until the
                                      # next replay

    The debug-mode bounds check of the layers (one host read per call) is switched off on this layer's modules: a host
    read cannot be captured.  Out-of-range ids then read as zero rows, as the kernels guarantee (no fault).
    """

    def __init__(self, layer, example_batch, label_name="label", loss_fn=None, warmup=3, extra_loss_fn=None):
        from . import functional as Fn
        from . import layers as CL
        self.layer = layer
        self.label_name = label_name
        self.extra_loss_fn = extra_loss_fn                   # (inputs dict) -> scalar added to the loss (DIN's L2 term)
        self.out = None                                      # the layer's output of the last replay (graph pool tensor)
        self.static = {k: v.clone() for k, v in example_batch.items() if isinstance(v, torch.Tensor)}
        self.loss_fn = loss_fn or (lambda out, y: Fn.KerasBCE.apply(out, y))
        for m in layer.modules():
            if isinstance(m, CL.Layer):
                m.check_ids = False                          # a host read of the bounds flag cannot be captured
        self.params = [p for p in layer.parameters() if p.requires_grad]
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._fwd_bwd()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        for p in self.params:
            p.grad = None
        with torch.cuda.graph(self.graph, capture_error_mode=CAPTURE_MODE):
            self.loss = self._fwd_bwd()
        self.grads = [p.grad for p in self.params]           # tensors of the graph's pool: refreshed by every replay

    def _target(self, out):
        y = self.static[self.label_name].to(torch.float32)
        if out.dim() == 2 and out.shape[1] > 1 and y.reshape(-1).numel() == out.shape[0]:
            y = y.reshape(-1, 1).expand(-1, out.shape[1])
        return y.contiguous()

    def _fwd_bwd(self):
        for p in self.params:
            p.grad = None
        ins = {k: v for k, v in self.static.items() if k != self.label_name}
        out = self.layer(ins)["output"]
        loss = self.loss_fn(out, self._target(out))
        if self.extra_loss_fn is not None:
            loss = loss + self.extra_loss_fn(ins)
        loss.backward()
        self.out = out.detach()
        return loss.detach()

    def __call__(self, batch):
        # one multi-tensor copy per dtype instead of one tiny copy per feature column (26 + label for DeepFM)
        groups = {}
        for k, v in self.static.items():
            src = batch[k]
            if src.dtype == v.dtype and src.device == v.device and src.shape == v.shape:
                groups.setdefault(v.dtype, ([], []))
                groups[v.dtype][0].append(v)
                groups[v.dtype][1].append(src)
            else:
                v.copy_(src, non_blocking=True)
        for dsts, srcs in groups.values():
            torch._foreach_copy_(dsts, srcs)
        self.graph.replay()
        for p, g in zip(self.params, self.grads):
            p.grad = g
        return self.loss
