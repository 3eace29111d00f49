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
    (``sparse_mode='keras'``, reference-exact) or the touched rows only (``'lazy'``, not the reference)."""

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
    ranks = np.empty(len(scores), np.float64)
    s_sorted = scores[order]
    i = 0
    while i < len(scores):                       # average ranks over ties
        j = i
        while j + 1 < len(scores) and s_sorted[j + 1] == s_sorted[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    return float((ranks[pos].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


class ModelManager:
    def __init__(self, feature_names=["user_tag1", "user_tag2", "item_tag1", "item_tag2", "item_tag3"],
                 json_path=None, data_info=None, embedding_dims=16, lr=0.00003, label_name="label", batch=100,
                 epochs=30, layer="fm_ranking", model_params={}, continuous_features=None,
                 behavior_series_features=None, adam_sparse_mode="keras", device="cuda", regularization_factor=0.01):
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

    def init_metric(self):
        self._loss_sum, self._loss_n, self._y, self._p = 0.0, 0, [], []

    def _metric_reset(self):
        self.init_metric()

    def _metric_update(self, loss, target, output):
        self._loss_sum += float(loss)
        self._loss_n += 1
        self._y.append(target.detach().reshape(target.shape[0], -1)[:, 0].cpu().numpy())
        out = output.detach()
        out = out[:, -1] if (out.dim() == 2 and out.shape[1] > 1) else out.reshape(-1)
        self._p.append(out.cpu().numpy())

    def _metric_result(self):
        if not self._y:
            return {"auc": float("nan"), "loss": float("nan")}
        return {"auc": auc_score(np.concatenate(self._y), np.concatenate(self._p)),
                "loss": self._loss_sum / max(1, self._loss_n)}

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

    def train_loop(self, inputs):
        """One iteration of 2.FM/ModelManager.py:171-181 (DIN: 5.DIN/ModelManager.py:170-197, which adds the L2 term
        on the embedding rows the batch used)."""
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

    def train_step(self, ds, epoch=None, summary_writer=None):
        self._metric_reset()
        step = 0
        for batch_data in sorted(list(ds), key=lambda x: random.random()):   # batch-order shuffle only (:185)
            self.train_loop(dict(batch_data))
            step += 1
        return self._metric_result()

    @torch.no_grad()
    def eval_step(self, ds):
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
