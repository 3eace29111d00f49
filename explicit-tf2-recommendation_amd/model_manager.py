"""Mirror of the slice of the reference's ModelManager.py that drives the hot path: the layer factory
(``make_layer_choice``), the model wrap (``init_model``), loss / optimizer construction and the custom train and
eval loops (2.FM/ModelManager.py:13-48, 61-119, 156-241; 3.DCN/ModelManager.py:64-112; 5.DIN/ModelManager.py:69-92,
170-197).  Data files, TFRecord parsing, checkpoints, TensorBoard and SavedModel export are harness plumbing of the
reference and out of scope; batches are dicts of tensors (the DataGenerator contract, see
data.py).

    mm = ModelManager(layer='deepfm_ranking', feature_names=[...], data_info=data.data_info(V, F), batch=8192)
    result = mm.train_step(batches)          # -> {'auc': ..., 'loss': ...}
"""
import json
import random

import numpy as np
import torch

from . import functional as Fn
from . import layers as CL
from . import ops


class KerasAdam:
    """tf.keras.optimizers.Adam(learning_rate) (2.FM/ModelManager.py:104) on the HIP kernels: dense parameters get
    the dense apply; tables get the Keras SPARSE apply, which decays m, v and moves var on ALL rows every step
    (``sparse_mode='keras'``, the reference's semantics; fp32 within an ulp of x per step of a correctly rounded evaluation, see csrc/common.h) or the touched rows only (``'lazy'``, not the reference)."""

    def __init__(self, params, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, sparse_mode="keras"):
        self.params = list(params)
        self.lr, self.b1, self.b2, self.eps = learning_rate, beta_1, beta_2, epsilon
        if sparse_mode not in ("keras", "lazy"):
            raise ValueError("sparse_mode must be 'keras' or 'lazy'")
        self.sparse_mode = sparse_mode
        self.iterations = 0
        self.m = [torch.zeros(p.shape, dtype=torch.float32, device=p.device) for p in self.params]
        self.v = [torch.zeros(p.shape, dtype=torch.float32, device=p.device) for p in self.params]

    @torch.no_grad()
    def apply_gradients(self, grads_and_vars=None):
        self.iterations += 1
        t = self.iterations
        for p, m, v in zip(self.params, self.m, self.v):
            g = p.grad
            if g is None:
                continue
            if g.is_sparse:
                # torch hands back an uncoalesced COO gradient (several lookups of one table add up): re-run the
                # deterministic de-duplication so that every id is applied once
                idx = g._indices()[0].contiguous()
                vals = g._values().contiguous().reshape(idx.numel(), -1)
                plan = ops.DedupPlan(idx, p.shape[0])
                rows = plan.segment_sum(vals, vals.shape[1])
                fn = ops.adam_sparse_keras if self.sparse_mode == "keras" else ops.adam_rows
                V0 = p.shape[0]        # tables with more than two axes (FFM's [V,F,E]) are rows of F*E floats
                fn(p.data.reshape(V0, -1), m.reshape(V0, -1), v.reshape(V0, -1), plan.uniq_ids, rows, plan.n_uniq, t,
                   self.lr, self.b1, self.b2, self.eps)
            else:
                ops.adam_dense(p.data, m, v, g.contiguous(), t, self.lr, self.b1, self.b2, self.eps)
            p.grad = None


def auc_score(labels, scores):
    """Area under the ROC curve (rank statistic); metric plumbing, evaluated on the host."""
    labels = np.asarray(labels).reshape(-1)
    scores = np.asarray(scores).reshape(-1)
    pos = labels > 0.5
    n_pos, n_neg = int(pos.sum()), int((~pos).sum())
    if n_pos == 0 or n_neg == 0:
        return float("nan")
    order = np.argsort(scores, kind="stable")
    s_sorted = scores[order]
    n = len(scores)
    change = np.ones(n, bool)                      # first element of every group of tied scores
    change[1:] = s_sorted[1:] != s_sorted[:-1]
    starts = np.nonzero(change)[0]
    ends = np.append(starts[1:], n)
    avg = 0.5 * (starts + ends - 1) + 1.0          # average rank of a tie group (1-based)
    ranks = np.empty(n, np.float64)
    ranks[order] = avg[np.cumsum(change) - 1]
    return float((ranks[pos].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


class ModelManager:
    def __init__(self, feature_names=["user_tag1", "user_tag2", "item_tag1", "item_tag2", "item_tag3"],
                 json_path=None, data_info=None, embedding_dims=16, lr=0.00003, label_name="label", batch=100,
                 epochs=30, layer="fm_ranking", model_params={}, continuous_features=None,
                 behavior_series_features=None, adam_sparse_mode="keras", device="cuda", regularization_factor=0.01,
                 engine="auto", staging_slots=2, steps_per_call=8):
        # engine: 'auto' -- the train loop of a batch shape is compiled once: engine.DeepFMFusedStep for 'deepfm_ranking'
        # with the reference's default head (two launches per iteration, optimizer inside, steps replayed from hipGraphs),
        # engine.GraphedTrainStep (forward + loss + autograd backward as ONE replayed hipGraph) for every other layer;
        # 'eager' -- every iteration through the autograd path, launch by launch (what rounds 1-2 did).
        if engine not in ("auto", "eager"):
            raise ValueError("engine must be 'auto' or 'eager'")
        self.engine = engine
        self.staging_slots = max(2, int(staging_slots))
        self.steps_per_call = max(1, int(steps_per_call))    # train_step: iterations per hipGraph of the compiled DeepFM loop
        self._eng = None                                     # built at the first train_loop call (needs the batch shapes)
        self.embedding_dims = embedding_dims
        self.lr = lr
        self.label_name = label_name
        self.batch = batch
        self.epochs = epochs
        self.model_params = dict(model_params)
        self.device = device
        self.adam_sparse_mode = adam_sparse_mode
        self.regularization_factor = regularization_factor   # 5.DIN/ModelManager.py:20,38 (used by the DIN loop only)
        self.continuous_features = list(continuous_features or [])
        self.behavior_series_features = list(behavior_series_features or [])
        self.set_feature_names(feature_names, label_name)
        self.load_json_info(json_path, data_info)
        self.feature_dims = self.feature_info[-1]          # total vocabulary (2.FM/ModelManager.py:38)
        self.make_layer_choice(layer_name=layer, model_params=self.model_params)
        self.init_model()
        self.init_loss()
        self.init_opt()
        self.init_metric()

    def load_json_info(self, json_path=None, data_info=None):
        if data_info is not None:
            self.feature_info = data_info
        else:
            with open(json_path, "rb") as f:
                self.feature_info = json.load(f)

    def set_feature_names(self, feature_names=None, label_name=None):
        if feature_names:
            self.feature_names = feature_names
        if label_name:
            self.label_name = label_name

    def make_layer_choice(self, layer_name="fm_ranking", model_params={}):
        kw = dict(feature_names=self.feature_names, feature_dims=self.feature_dims,
                  embedding_dims=self.embedding_dims)
        if layer_name == "fm_ranking":
            self.layer = CL.FMRankingLayer(**kw, **model_params)
        elif layer_name == "deepfm_ranking":
            self.layer = CL.DeepFMRankingLayer(**kw, **model_params)
        elif layer_name == "dssm_single_tower":
            self.layer = CL.DSSMSingleTowerLayer(**model_params)
        elif layer_name == "dssm_double_tower":
            if len(model_params) == 0:
                self.layer = CL.DSSMTwoTowerRetrievalLayer(
                    u_feature_names=["user_tag1", "user_tag2"], i_feature_names=["item_tag1", "item_tag2", "item_tag3"],
                    u_feature_dims=self.feature_dims, i_feature_dims=self.feature_dims)
            else:
                self.layer = CL.DSSMTwoTowerRetrievalLayer(**model_params)
        elif layer_name == "ffm_ranking":                  # 2.FM/ModelManager.py:76-77
            self.layer = CL.FFMRankingLayer(**kw, **model_params)
        elif layer_name == "pnn_ranking":                  # 2.FM/ModelManager.py:78-82 (method 'inner' is accelerated)
            self.layer = CL.PNNRankingLayer(**kw, method=model_params.get("method", "inner"),
                                            kernel_type=model_params.get("kernel_type", "mat"))
        elif layer_name == "NFM":                          # 3.DCN/ModelManager.py:78-79
            self.layer = CL.NeuralFactorizationMachineLayer(
                categorical_features=self.feature_names, continuous_features=self.continuous_features,
                feature_dims=self.feature_dims, embedding_dims=self.embedding_dims, **model_params)
        elif layer_name == "dcn_ranking":                  # 3.DCN/ModelManager.py:69-71 (type: 'vec' | 'matrix')
            p = dict(model_params)
            p.setdefault("categorical_features", self.feature_names)
            p.setdefault("continuous_features", self.continuous_features)
            self.layer = CL.DeepCrossNetworkLayer(feature_dims=self.feature_dims, embedding_dims=self.embedding_dims,
                                                  **p)
        elif layer_name == "din_layer":                    # 5.DIN/ModelManager.py:72-73
            p = dict(model_params)
            p.setdefault("feature_dims", self.feature_dims)
            p.setdefault("embedding_dims", self.embedding_dims)
            self.layer = CL.DINLayer(**p)
        else:
            raise ValueError("不在可用的模型范围内")

    def init_model(self, layer=None):
        layer = layer if layer else self.layer
        self.model = layer.to(self.device)                 # dict in, dict out: the Keras functional wrap adds nothing

    def init_loss(self):
        self.loss = lambda target, output: Fn.KerasBCE.apply(output, self._match_target(target, output))

    @staticmethod
    def _match_target(target, output):
        """Keras squeezes y [B,1] for [B] predictions and broadcasts it for [B,k] ones."""
        t = target.to(torch.float32)
        if output.dim() == 2 and output.shape[1] > 1 and t.reshape(-1).numel() == output.shape[0]:
            t = t.reshape(-1, 1).expand(-1, output.shape[1])
        return t.contiguous()

    def init_opt(self):
        self.opt = KerasAdam(self.model.trainable_variables, learning_rate=self.lr, sparse_mode=self.adam_sparse_mode)

    # ---- metrics: tf.keras.metrics.AUC(name="auc") + Mean loss (2.FM/ModelManager.py:106-108) ---------------------------
    # Keras' AUC is a STREAMING approximation: 200 thresholds ((i+1)/199 for i < 198, framed by -1e-7 and 1+1e-7), per
    # threshold the counts of true / false positives `prediction > threshold`, ROC area by the trapezoid rule
    # (summation_method='interpolation').  Restated as a histogram: a prediction falls into bucket k = number of thresholds
    # below it; TP[i] = positives in buckets > i.  Two tiny device kernels per iteration, nothing read back until the
    # result is asked for (the reference logs every 500 steps, 2.FM/ModelManager.py:194-199; a per-step .item() would
    # stall the GPU once per iteration, and an exact rank-AUC of an epoch sorts millions of scores on the host).
    AUC_THRESHOLDS = 200

    def init_metric(self):
        self._loss_sum, self._loss_n = 0.0, 0
        self._auc_hist = np.zeros((2, self.AUC_THRESHOLDS + 1), np.float64)      # [negatives | positives] per bucket (host)
        self._dev_metric = None                                                 # device accumulators of the compiled loop

    def _metric_reset(self):
        self.init_metric()

    @classmethod
    def _auc_thresholds(cls):
        n = cls.AUC_THRESHOLDS
        return np.array([-1e-7] + [(i + 1) / (n - 1) for i in range(n - 2)] + [1.0 + 1e-7], np.float64)

    def _metric_update_dev(self, loss, target, prob, n_steps=1):
        """One launch (rec_auc_hist_update_f32): the predictions' buckets counted with integer atomics, the per-step losses
        added in double -- nothing is read back until _metric_result()."""
        d = self._dev_metric
        if d is None:
            dev = loss.device
            d = self._dev_metric = {"thr": torch.from_numpy(self._auc_thresholds().astype(np.float32)).to(dev),
                                    "hist": torch.zeros(2 * (self.AUC_THRESHOLDS + 1), dtype=torch.int64, device=dev),
                                    "loss": torch.zeros(1, dtype=torch.float64, device=dev), "n": 0}
        y, p, ls = target.reshape(-1), prob.reshape(-1), loss.reshape(-1)
        if y.dtype != torch.float32 or not y.is_contiguous():
            y = y.to(torch.float32).contiguous()
        if p.dtype != torch.float32 or not p.is_contiguous():
            p = p.to(torch.float32).contiguous()
        if ls.dtype != torch.float32 or not ls.is_contiguous():
            ls = ls.to(torch.float32).contiguous()
        assert y.numel() == p.numel() and ls.numel() >= n_steps
        ops.check(ops.lib.rec_auc_hist_update_f32(ops._ptr(p), ops._ptr(y), p.numel(), ops._ptr(d["thr"]),
                                                  self.AUC_THRESHOLDS, ops._ptr(d["hist"]), ops._ptr(ls), int(n_steps),
                                                  ops._ptr(d["loss"]), ops._stream()), "rec_auc_hist_update_f32")
        d["n"] += n_steps

    def _metric_update(self, loss, target, output):
        self._loss_sum += float(loss)
        self._loss_n += 1
        y = target.detach().reshape(target.shape[0], -1)[:, 0].cpu().numpy()
        out = output.detach()
        out = out[:, -1] if (out.dim() == 2 and out.shape[1] > 1) else out.reshape(-1)
        p = out.cpu().numpy().astype(np.float32)
        k = np.searchsorted(self._auc_thresholds().astype(np.float32), p, side="left")
        np.add.at(self._auc_hist, ((y > 0.5).astype(np.int64), k), 1.0)

    def _metric_result(self):
        d = self._dev_metric
        if d is not None and d["n"] > 0:                     # ONE device -> host read
            self._auc_hist += d["hist"].cpu().numpy().astype(np.float64).reshape(2, -1)
            self._loss_sum += float(d["loss"].item())
            self._loss_n += d["n"]
            d["hist"].zero_(); d["loss"].zero_(); d["n"] = 0
        if self._loss_n == 0:
            return {"auc": float("nan"), "loss": float("nan")}
        neg, pos = self._auc_hist
        # counts above threshold i = buckets i+1 ..: reverse cumulative sums
        tp = np.cumsum(pos[::-1])[::-1][1:]
        fp = np.cumsum(neg[::-1])[::-1][1:]
        n_pos, n_neg = pos.sum(), neg.sum()
        if n_pos == 0 or n_neg == 0:
            auc = float("nan")
        else:
            tpr, fpr = tp / n_pos, fp / n_neg
            auc = float(np.sum((fpr[:-1] - fpr[1:]) * (tpr[:-1] + tpr[1:]) / 2.0))
        return {"auc": auc, "loss": self._loss_sum / max(1, self._loss_n)}

    def _to_device(self, inputs):
        out = {}
        for k, v in inputs.items():
            t = v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))
            out[k] = t.to(self.device)
        return out

    def used_rows_l2(self, inputs):
        """5.DIN/ModelManager.py:176-190: ids of the user/context, item and (flattened) behaviour-series features in
        one vector, tf.unique, then regularization_factor * l2_loss of those embedding rows."""
        lay = self.model
        parts = [inputs[f].reshape(-1) for f in list(lay.user_and_context_categorical_features) +
                 list(lay.item_categorical_features) + list(lay.behavior_series_features)]
        all_ids = torch.cat([p.to(torch.int64) for p in parts])
        return Fn.UsedRowsL2.apply(lay.embed.embeddings, all_ids, self.regularization_factor)

    # ---- the compiled train loop -------------------------------------------------------------------------------
    def _fill(self, slot, inputs):
        """Copy a batch into a staging slot (dict of device tensors of fixed shape; None: allocate one)."""
        src = {}
        for name, v in inputs.items():
            t = v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))
            if t.dtype in (torch.float64, torch.float16, torch.bfloat16):
                t = t.to(torch.float32)                      # labels / continuous features: float32 (2.FM/ModelManager.py:92)
            elif t.dtype in (torch.int32, torch.int16, torch.int8, torch.uint8):
                t = t.to(torch.int64)
            src[name] = t
        if slot is None:
            # tensors of one (dtype, shape) -- the 26 id columns of a DeepFM batch -- live in ONE block, so that a batch is
            # staged by one stack / one host-to-device copy per group instead of one copy per feature (27 launches, ~0.1 ms
            # of host time per batch)
            groups = {}
            for n, t in src.items():
                groups.setdefault((t.dtype, tuple(t.shape)), []).append(n)
            slot = {"__groups__": []}
            for (dt, shp), members in groups.items():
                block = torch.empty((len(members),) + shp, dtype=dt, device=self.device)
                slot["__groups__"].append((members, block))
                for g, n in enumerate(members):
                    slot[n] = block[g]
        for n, t in src.items():
            if n not in slot or tuple(slot[n].shape) != tuple(t.shape) or slot[n].dtype != t.dtype:
                raise ValueError("the compiled train loop takes batches of one shape (%r changed): build another "
                                 "ModelManager, or engine='eager'" % n)
        for members, block in slot["__groups__"]:
            srcs = [src[n] for n in members]
            if all(t.is_cuda for t in srcs):
                if len(srcs) == 1:
                    block[0].copy_(srcs[0])
                else:
                    torch.stack(srcs, out=block)             # one launch
            elif not any(t.is_cuda for t in srcs):
                block.copy_(torch.stack(srcs) if len(srcs) > 1 else srcs[0].unsqueeze(0), non_blocking=True)   # one copy
            else:
                for g, t in enumerate(srcs):
                    block[g].copy_(t, non_blocking=True)
        return slot

    def _stage(self, inputs):
        """Copy a batch into the next slot of a fixed ring of device staging buffers: the compiled step sees the same
        addresses again and again, so its hipGraphs are captured once per slot and replayed -- whatever the input
        pipeline hands over (host arrays, fresh device tensors)."""
        ring = self.__dict__.setdefault("_ring", [])
        k = self.__dict__.get("_ring_pos", 0)
        self._ring_pos = (k + 1) % self.staging_slots
        if len(ring) <= k:
            ring.append(None)
        ring[k] = self._fill(ring[k], inputs)
        return ring[k]

    def _build_engine(self, slot):
        from . import engine as EN
        lay = self.model
        info = self.feature_info
        B = slot[self.label_name].numel()
        fused_ok = (isinstance(lay, CL.DeepFMRankingLayer) and lay.embedding_dims == 16 and
                    list(lay.mlp_dims) == [32, 8] and len(lay.feature_names) <= 28 and B <= 16384 and
                    isinstance(info, (list, tuple)) and len(info) >= 2 and
                    len(info[0]) == len(lay.feature_names) == len(info[1]) and str(self.device).startswith("cuda"))
        if fused_ok:
            try:
                opt = "keras_adam_lazy" if self.adam_sparse_mode == "keras" else "lazy_adam"
                step = EN.DeepFMFusedStep(lay, B, list(info[0]), list(info[1]), optimizer=opt, lr=self.lr, want_prob=True)
                return ("fused", step)
            except NotImplementedError:
                pass
        extra = None
        if isinstance(lay, CL.DINLayer) and self.regularization_factor and hasattr(lay.embed, "embeddings"):
            extra = lambda ins: self.used_rows_l2(ins)
        step = EN.GraphedTrainStep(lay, slot, label_name=self.label_name,
                                   loss_fn=lambda out, y: Fn.KerasBCE.apply(out, y), extra_loss_fn=extra)
        return ("graphed", step)

    def sync_parameters(self):
        """Bring every parameter up to date before it is read from outside the train loop (evaluation, export): the
        compiled DeepFM step evaluates Keras' dense Adam sweep LAZILY -- rows skip the sweeps and replay them when a
        batch reads them -- and this replays what is still pending for every row (bit-identical to sweeping each step)."""
        if self._eng is not None and self._eng[0] == "fused":
            self._eng[1].flush()

    def train_loop(self, inputs, next_inputs=None):
        """One iteration of 2.FM/ModelManager.py:171-181 (DIN: 5.DIN/ModelManager.py:170-197, which adds the L2 term
        on the embedding rows the batch used).  Returns the loss as a DEVICE scalar (no synchronisation).
        ``next_inputs``: the batch of the next call, if the caller already has it -- its de-duplication plan is then built
        beside this iteration (compiled DeepFM step)."""
        if self.engine == "auto" and str(self.device).startswith("cuda"):
            pend = self.__dict__.get("_pending")
            if pend is not None and pend[0] is inputs:       # staged (and its plan prefetched) by the previous call
                slot = pend[1]
            else:
                slot = self._stage(inputs)
            self._pending = None
            if self._eng is None:
                self._eng = self._build_engine(slot)
            kind, step = self._eng
            target = slot[self.label_name]
            if kind == "fused":
                nxt = None
                if next_inputs is not None:
                    nxt = self._stage(next_inputs)
                    self._pending = (next_inputs, nxt)       # the next call finds its batch staged (same dict object)
                loss = step(slot, self.label_name, next_inputs=nxt).clone()   # the step's buffer is overwritten by the next call
                self._metric_update_dev(loss, target, step.prob)
                return loss
            loss = step(slot).clone()
            self.opt.apply_gradients()
            out = step.out
            prob = out[:, -1] if (out.dim() == 2 and out.shape[1] > 1) else out.reshape(-1)
            self._metric_update_dev(loss, target, prob)
            return loss
        inputs = self._to_device(inputs)
        target = inputs.pop(self.label_name)
        logits = self.model(inputs)
        scaled_loss = self.loss(target, logits["output"])
        if isinstance(self.model, CL.DINLayer) and self.regularization_factor:
            scaled_loss = scaled_loss + self.used_rows_l2(inputs)
        scaled_loss.backward()
        self.opt.apply_gradients()
        self._metric_update(scaled_loss.item(), target, logits["output"])
        return scaled_loss

    def init_dataset(self, mode="train", data_dir=None, feature_names=None, label_name=None):
        """2.FM/ModelManager.py:122-153: the TFRecord files of ``data_dir`` whose name contains ``mode``, parsed as
        FixedLenFeature([1]) and batched (tfrecord.TFRecordDataset; no TensorFlow involved)."""
        from . import tfrecord
        assert mode in ("train", "test")
        self.set_feature_names(feature_names, label_name)
        return tfrecord.TFRecordDataset(data_dir, mode, self.feature_names, self.label_name, self.batch)

    def _train_chunks(self, order):
        """The epoch of the compiled DeepFM loop, `steps_per_call` iterations per call: a hipGraph launch leaves the GPU
        idle for ~30 us and every call costs ~0.1 ms of Python, so the chunk of batches is staged into one half of a ring
        of 2 x steps_per_call buffers, the next chunk into the other half (its de-duplication plans are built beside this
        chunk's iterations), and the chunk runs as ONE replayed graph; the per-iteration losses / predictions come out of
        the step's own per-step buffers, for the metrics."""
        R = self.steps_per_call
        step = self._eng[1]
        ring = self.__dict__.setdefault("_chunk_ring", [None, None])

        def half_of(h, example):
            """R staging slots whose tensors of one (dtype, shape) are slices of ONE block [R, members, ...]: a chunk of
            device-resident batches is staged by one rec_block_copy launch per block, and the labels of the chunk are a
            view of their block"""
            if ring[h] is None:
                proto = self._fill(None, example)            # (normalises dtypes; its blocks give the groups)
                blocks, slots = [], [{"__groups__": []} for _ in range(R)]
                for members, blk in proto["__groups__"]:
                    big = torch.empty((R,) + tuple(blk.shape), dtype=blk.dtype, device=blk.device)
                    blocks.append((members, big))
                    for j in range(R):
                        slots[j]["__groups__"].append((members, big[j]))
                        for g, nme in enumerate(members):
                            slots[j][nme] = big[j][g]
                ring[h] = (blocks, slots)
            return ring[h]

        def stage(chunk, h):
            blocks, slots = half_of(h, chunk[0])
            n_names = len(slots[0]) - 1
            fast, todo = True, []
            for members, big in blocks:                      # device-resident batches of exactly the staged layout?
                dt, shp, ptrs = big.dtype, big.shape[2:], []
                for b in chunk:
                    if len(b) != n_names:
                        fast = False
                        break
                    for nme in members:
                        t = b.get(nme)
                        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype is dt and t.shape == shp and
                                t.is_contiguous()):
                            fast = False
                            break
                        ptrs.append(t.data_ptr())
                    if not fast:
                        break
                if not fast:
                    break
                todo.append((big, ptrs))
            if fast:
                for big, ptrs in todo:
                    arr = (ops.C.c_void_p * len(ptrs))(*ptrs)
                    ops.check(ops.lib.rec_block_copy(arr, len(ptrs), big[0][0].numel() * big.element_size(), ops._ptr(big),
                                                     ops._stream()), "rec_block_copy")
            else:
                for j, b in enumerate(chunk):
                    self._fill(slots[j], b)
            return slots[:len(chunk)]

        def labels_of(h, n):
            for members, big in ring[h][0]:
                if self.label_name in members:
                    return big[:n, members.index(self.label_name)]
            raise KeyError(self.label_name)

        chunks = [order[i:i + R] for i in range(0, len(order), R)]
        cur = stage(chunks[0], 0)
        for c, chunk in enumerate(chunks):
            nxt = stage(chunks[c + 1], (c + 1) % 2) if c + 1 < len(chunks) else None
            step.many(cur, self.label_name, then=nxt)
            n = len(cur)
            self._metric_update_dev(step.loss_steps[:n], labels_of(c % 2, n), step.prob_steps[:n], n)
            cur = nxt
        self._pending = None

    def train_step(self, ds, epoch=None, summary_writer=None):
        self._metric_reset()
        order = [dict(b) for b in sorted(list(ds), key=lambda x: random.random())]    # batch-order shuffle only (:185)
        if self.engine == "auto" and str(self.device).startswith("cuda") and order:
            if self._eng is None:
                self.train_loop(order[0])                    # builds the engine for this batch shape (and trains on it)
                order = order[1:]
            if self._eng[0] == "fused" and order:
                self._train_chunks(order)
                return self._metric_result()
        for i, batch_data in enumerate(order):
            nxt = order[i + 1] if i + 1 < len(order) else None                # the input pipeline knows what comes next
            self.train_loop(batch_data, next_inputs=nxt)
        return self._metric_result()

    @torch.no_grad()
    def eval_step(self, ds):
        self.sync_parameters()
        self._metric_reset()
        for batch_data in ds:
            inputs = self._to_device(dict(batch_data))
            target = inputs.pop(self.label_name)
            logits = self.model(inputs)
            loss, _, _ = ops.bce_fwd_bwd(self._match_target(target, logits["output"]), logits["output"].contiguous(),
                                         want_dp=False)
            self._metric_update(loss.item(), target, logits["output"])
        return self._metric_result()

    def run(self, train_ds=None, test_ds=None, mode="train_and_eval"):
        results = []
        if mode == "train_and_eval":
            for epoch in range(self.epochs):
                results.append((self.train_step(train_ds, epoch), self.eval_step(test_ds)))
        elif mode == "train":
            for epoch in range(self.epochs):
                results.append(self.train_step(train_ds, epoch))
        elif mode == "eval":
            results.append(self.eval_step(test_ds))
        return results
