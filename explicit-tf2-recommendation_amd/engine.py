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


class DeepFMTrainStep:
    """fwd + bwd (+ optimizer) of DeepFMRankingLayer (2.FM/CustomLayers.py:279-308) under the reference's loss.

    optimizer: None (gradients only -- the 'fwd+bwd' of the headline metric), 'keras_adam' (reference-exact:
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
            with torch.cuda.graph(g):
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
    """The same train_loop iteration as DeepFMTrainStep in FOUR launches (csrc/deepfm_fused.hip): the fused
    forward+backward kernel and its fixed-order reduction on one stream, the per-column LDS sort of the
    de-duplication plan on a second stream (it depends only on the ids), then the segment sums.

    Requirements (checked; otherwise use DeepFMTrainStep): embedding_dims 16, mlp_dims [32,8], fused table layout,
    F <= 28, B <= 16384, and the DataGenerator id-space contract -- ``field_offsets[f]``/``field_dims[f]`` =
    ``data_info.json``'s offsets and dims (2.FM/DataGenerator.py:126-134); an id outside its field's range sets
    ``self.bad_ids`` (checked by ``check_flags()``).
    """

    def __init__(self, layer, batch_size, field_dims, field_offsets, optimizer=None, lr=1e-3, use_graph=True):
        self.layer = layer
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
        self.optimizer, self.lr, self.use_graph, self.t = optimizer, lr, use_graph, 0
        f32 = dict(dtype=torch.float32, device=dev)
        n, D = B * F, F * 16
        self.col_lo = torch.tensor([int(o) for o in field_offsets], dtype=torch.int64, device=dev)
        self.gz = torch.empty(B, **f32)
        self.vals = torch.empty((n, 16), **f32)
        self.loss = torch.empty(1, **f32)
        self.oob = torch.zeros(1, dtype=torch.int32, device=dev)
        self.bad_ids = torch.zeros(1, dtype=torch.int32, device=dev)
        self.g = {
            "MLP_layer1.kernel_0": torch.empty((D, 32), **f32), "MLP_layer1.bias_0": torch.empty(32, **f32),
            "MLP_layer1.kernel_1": torch.empty((32, 8), **f32), "MLP_layer1.bias_1": torch.empty(8, **f32),
            "MLP_layer2.kernel_0": torch.empty((8, 1), **f32), "MLP_layer2.bias_0": torch.empty(1, **f32),
            "bias": torch.empty(1, **f32),
        }
        self.ws = torch.empty(lib.rec_deepfm_fused_workspace_bytes(B, F), dtype=torch.uint8, device=dev)
        # two plan buffers: the de-duplication plan depends on the ids only, so the plan of batch k+1 can be built
        # (second stream) while batch k is being differentiated
        self.plans = [dict(perm=torch.empty((F, B), dtype=torch.int32, device=dev),
                           col_uid=torch.empty((F, B), dtype=torch.int64, device=dev),
                           col_seg=torch.empty((F, B + 1), dtype=torch.int32, device=dev),
                           col_nu=torch.zeros(F, dtype=torch.int32, device=dev)) for _ in range(2)]
        self._plan_key, self._plan_buf = None, 0
        self.uniq_ids = torch.empty(n, dtype=torch.int64, device=dev)
        self.g_embed_rows = torch.empty((n, 16), **f32)
        self.g_w_rows = torch.empty((n, 1), **f32)
        self.n_uniq = torch.zeros(1, dtype=torch.int64, device=dev)
        self.sort_ws = torch.empty(lib.rec_colsort_workspace_bytes(B, F), dtype=torch.uint8, device=dev)
        self.side_stream = torch.cuda.Stream(device=dev)
        if optimizer is not None:
            self.state = {name: (torch.zeros(p.shape, **f32), torch.zeros(p.shape, **f32))
                          for name, p in layer.named_parameters()}
            self.side_e = torch.empty((n, 3, 16), **f32)
            self.side_w = torch.empty((n, 3, 1), **f32)
        self._graphs = {}

    def _sort(self, cols, buf, stream):
        F = self.F
        arr = (C.c_void_p * F)(*[c.data_ptr() for c in cols])
        pl = self.plans[buf]
        check(lib.rec_colsort_plan_i64(arr, F, self.B, self.V, _p(self.col_lo), self.max_key, _p(pl["perm"]),
                                       _p(pl["col_uid"]), _p(pl["col_seg"]), _p(pl["col_nu"]), _p(self.bad_ids),
                                       _p(self.sort_ws), C.c_void_p(stream.cuda_stream)), "rec_colsort_plan_i64")

    def _enqueue(self, cols, label, t, cur, have_plan, next_cols):
        """cur: plan buffer of this batch; have_plan: it was filled by the previous step; next_cols: columns of
        the next batch, whose plan is built into the other buffer concurrently (second stream)."""
        L = self.layer
        F, B, V = self.F, self.B, self.V
        arr = (C.c_void_p * F)(*[c.data_ptr() for c in cols])
        main = torch.cuda.current_stream()
        side = self.side_stream
        forked = (not have_plan) or (next_cols is not None)
        if forked:
            side.wait_stream(main)                               # fork: a sort only needs ids
            with torch.cuda.stream(side):
                if not have_plan:
                    self._sort(cols, cur, side)                  # this batch's own plan (non-pipelined call)
                if next_cols is not None:
                    self._sort(next_cols, 1 - cur, side)         # the next batch's plan
        st = C.c_void_p(main.cuda_stream)
        g = self.g
        emb = L.embed.embeddings
        check(lib.rec_deepfm_fused_fwd_bwd_f32(
            _p(emb), emb.stride(0), V, arr, F, B, _p(L.bias), _p(L.MLP_layer1.kernel_0), _p(L.MLP_layer1.bias_0),
            _p(L.MLP_layer1.kernel_1), _p(L.MLP_layer1.bias_1), _p(L.MLP_layer2.kernel_0), _p(L.MLP_layer2.bias_0),
            _p(label), _p(self.gz), _p(self.vals), None, _p(g["MLP_layer1.kernel_0"]), _p(g["MLP_layer1.bias_0"]),
            _p(g["MLP_layer1.kernel_1"]), _p(g["MLP_layer1.bias_1"]), _p(g["MLP_layer2.kernel_0"]),
            _p(g["MLP_layer2.bias_0"]), _p(g["bias"]), _p(self.loss), _p(self.oob), _p(self.ws), st),
            "rec_deepfm_fused_fwd_bwd_f32")
        if forked and not have_plan:
            main.wait_stream(side)                               # this batch's plan is needed now
        pl = self.plans[cur]
        check(lib.rec_colseg_sum_f32(_p(self.vals), _p(self.gz), _p(pl["perm"]), _p(pl["col_uid"]), _p(pl["col_seg"]),
                                     _p(pl["col_nu"]), B, F, _p(self.uniq_ids), _p(self.g_embed_rows),
                                     _p(self.g_w_rows), _p(self.n_uniq), st), "rec_colseg_sum_f32")
        if self.optimizer is not None:
            self._optimizer(t, st)
        if forked and have_plan:
            main.wait_stream(side)                               # join: the next step may rely on the other buffer

    def _optimizer(self, t, st):
        lr, b1, b2, eps = self.lr, 0.9, 0.999, 1e-7
        params = dict(self.layer.named_parameters())
        for name, grad in self.g.items():
            m, v = self.state[name]
            check(lib.rec_adam_dense_f32(_p(params[name]), _p(m), _p(v), _p(grad), grad.numel(), t, lr, b1, b2, eps, st),
                  "rec_adam_dense_f32")
        n = self.B * self.F
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
        de-duplication plan is built on the second stream while this batch is differentiated (input-pipeline style
        prefetch: the plan depends on ids only).  Without it, or when the previous call did not announce this batch,
        the plan is built inside this call."""
        cols = self._cols(inputs)
        y = inputs[label_name]
        if y.dtype != torch.float32 or not y.is_cuda or y.numel() != self.B or not y.is_contiguous():
            raise ValueError("label must be a contiguous float32 CUDA tensor with %d entries" % self.B)
        next_cols = self._cols(next_inputs) if next_inputs is not None else None
        key = tuple(c.data_ptr() for c in cols)
        next_key = tuple(c.data_ptr() for c in next_cols) if next_cols is not None else None
        have_plan = self._plan_key is not None and self._plan_key == key
        cur = self._plan_buf if have_plan else 0
        self.t += 1
        if not self.use_graph or self.optimizer is not None:
            self._enqueue(cols, y, self.t, cur, have_plan, next_cols)
        else:
            gkey = (key, y.data_ptr(), next_key, cur, have_plan)
            ent = self._graphs.get(gkey)
            if ent is None:
                self._enqueue(cols, y, self.t, cur, have_plan, next_cols)   # warm-up (sets the kernel attributes)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._enqueue(cols, y, self.t, cur, have_plan, next_cols)
                ent = (g, cols, y, next_cols)                    # keep the inputs alive: the graph holds addresses
                self._graphs[gkey] = ent
            ent[0].replay()
        if next_cols is not None:
            self._plan_key, self._plan_buf = next_key, 1 - cur
        else:
            self._plan_key = None
        return self.loss

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
