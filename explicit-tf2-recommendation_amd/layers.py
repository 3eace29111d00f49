"""Mirror of the reference's CustomLayers.py for the hot path -- same class names, constructor keywords,
dict-in / dict-out ``__call__`` and error behaviour -- with every op executed by the HIP kernels of
libmi355rec.so (through functional.py / ops.py).  Reference classes and the lines they follow:

  MLPLayer                      2.FM/CustomLayers.py:15-84   (default activation None; 'relu' in the 3.DCN/5.DIN copies)
  FMRankingLayer                2.FM/CustomLayers.py:87-157
  DSSMSingleTowerLayer          2.FM/CustomLayers.py:159-206
  DSSMTwoTowerRetrievalLayer    2.FM/CustomLayers.py:208-239
  DeepFMRankingLayer            2.FM/CustomLayers.py:241-308
  DenseLayer                    3.DCN/CustomLayers.py:153-167
  CrossLayer                    3.DCN/CustomLayers.py:170-203
  DeepCrossNetworkLayer         3.DCN/CustomLayers.py:206-269
  MatrixCrossLayer              3.DCN/CustomLayers.py:272-305

Parameters are named after the TF checkpoint keys (``embed.embeddings``, ``w.embeddings``, ``bias``,
``MLP_layer1.kernel_0`` ...), so a TensorBundle checkpoint maps onto ``state_dict()`` by name.
Index tensors may be int64 ``[B,1]`` (ModelManager path) or ``[B]`` (direct layer call); out-of-range ids raise
IndexError when ``check_ids`` is on (the reference raises InvalidArgumentError on CPU).
"""
import math

import torch

from . import functional as Fn
from . import ops


# ---------------------------------------------------------------------------------------------------
# initialisers (Keras defaults)
# ---------------------------------------------------------------------------------------------------

_init_gen = torch.Generator(device="cpu")
_init_gen.manual_seed(1234)


def set_init_seed(seed):
    _init_gen.manual_seed(seed)


def _uniform(shape, lim):
    return (torch.rand(shape, generator=_init_gen) * 2 - 1) * lim


def glorot_uniform(shape):
    fan_in, fan_out = (shape[0], shape[1]) if len(shape) == 2 else (shape[0], shape[0])
    return _uniform(shape, math.sqrt(6.0 / (fan_in + fan_out)))


def _initializer(name):
    if callable(name):
        return name
    table = {"glorot_uniform": glorot_uniform, "zeros": lambda s: torch.zeros(s),
             "random_normal": lambda s: torch.randn(s, generator=_init_gen) * 0.05,
             "uniform": lambda s: _uniform(s, 0.05)}
    if name not in table:
        raise ValueError("Unknown initializer: %r" % (name,))
    return table[name]


def _activation_code(activation):
    if activation not in ops.ACT_CODE:
        raise ValueError("Unknown activation function: %r" % (activation,))
    return ops.ACT_CODE[activation]


class Layer(torch.nn.Module):
    """Keras-Layer conveniences on top of torch.nn.Module."""

    check_ids = True            # debug-mode bounds check: one 4-byte device->host read per call

    @property
    def trainable_variables(self):
        return [p for p in self.parameters() if p.requires_grad]

    def _device(self):
        for p in self.parameters():
            return p.device
        return torch.device("cuda")

    def _raise_if_oob(self, flag):
        if flag is not None and int(flag.item()) != 0:
            raise IndexError("embedding id out of range [0, feature_dims)")

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        post = getattr(self, "_post_apply", None)
        if post is not None:
            post()                  # e.g. re-establish the fused table layout after .cuda()/.to()
        return out


class _FMTables:
    """Mixin of the FM-family layers: keep ``embed.embeddings`` [V,E] and ``w.embeddings`` [V,1] as two strided
    views of ONE device array [V, ld] (row = [embed | w | pad], ld = next_pow2(E+1) >= 16) so that both values
    of an id arrive in the same 128-byte line.  The two Parameters keep their reference names and shapes
    (``state_dict``, ``copy_``, sparse gradients all work); only their strides change."""

    def fuse_tables(self):
        e, w = self.embed.embeddings, self.w.embeddings
        V, E = e.shape
        if not e.is_cuda or E % 4 != 0:
            return False
        ld = ops.fused_row_stride(E)
        if e.stride(0) == ld and w.stride(0) == ld and w.data_ptr() == e.data_ptr() + 4 * E:
            return True
        storage = torch.zeros((V, ld), dtype=torch.float32, device=e.device)
        storage[:, :E].copy_(e.data)
        storage[:, E:E + 1].copy_(w.data)
        self.embed.embeddings = torch.nn.Parameter(storage[:, :E], requires_grad=e.requires_grad)
        self.w.embeddings = torch.nn.Parameter(storage[:, E:E + 1], requires_grad=w.requires_grad)
        self._fused_storage = storage
        return True

    def _post_apply(self):
        self.fuse_tables()


def assemble_index(inputs, feature_names):
    """expand_dims(rank-1) + concat(axis=1)  (2.FM/CustomLayers.py:138-144) -> X [B,F] int64, on the GPU."""
    cols = []
    for name in feature_names:
        t = inputs[name]
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(t)
        if t.dtype != torch.int64:
            t = t.to(torch.int64)
        if t.dim() not in (1, 2) or (t.dim() == 2 and t.shape[1] != 1):
            raise ValueError("feature %r must have shape [B] or [B,1], got %s" % (name, tuple(t.shape)))
        if not t.is_cuda:
            t = t.cuda()
        cols.append(t.contiguous())
    return ops.index_pack(cols)


class Embedding(Layer):
    """tf.keras.layers.Embedding(V, E): table ``embeddings`` ~ U(-0.05, 0.05)."""

    def __init__(self, input_dim, output_dim, embeddings_regularizer=None):
        super().__init__()
        # the "l2" regulariser of the reference lands in layer.losses, which its train loop never adds
        # (2.FM/ModelManager.py:175): kept for signature compatibility, no effect on gradients.
        self.embeddings_regularizer = embeddings_regularizer
        self.embeddings = torch.nn.Parameter(_uniform((input_dim, output_dim), 0.05))

    def forward(self, X, oob=None, sink=None):
        return Fn.Gather.apply(self.embeddings, X, oob, sink)

    def grad_sink(self, X):
        """A functional.GradSink for a second lookup of this table in the same step (None when no gradient is
        recorded)."""
        if torch.is_grad_enabled() and self.embeddings.requires_grad:
            return Fn.GradSink(X.numel())
        return None


def make_embedding(feature_dims, embedding_dims, sharded=False, group=None, comm=None, capacity=None,
                   embeddings_regularizer=None, device=None):
    """The table of a layer: layers.Embedding, or -- ``sharded=True`` -- sharded.ShardedEmbedding, its rows
    block-partitioned over the ranks of ``group`` (where the reference builds tf.keras.layers.Embedding,
    2.FM/CustomLayers.py:176-178, 3.DCN/CustomLayers.py:231, 5.DIN/CustomLayers.py:216-217).  Same call signature;
    the parameter is ``embeddings_shard`` [ceil(V/P), E] instead of ``embeddings`` [V, E]."""
    if not sharded:
        return Embedding(feature_dims, embedding_dims, embeddings_regularizer=embeddings_regularizer)
    from . import sharded as _sh
    return _sh.ShardedEmbedding(feature_dims, embedding_dims, group=group, comm=comm, capacity=capacity, device=device)


def _is_sharded(embed):
    return hasattr(embed, "embeddings_shard")


class MLPLayer(Layer):
    """MatMul + BiasAdd + activation on EVERY layer (2.FM/CustomLayers.py:72-84)."""

    def __init__(self, units, activation=None, use_bias=True, is_batch_norm=False, is_dropput=0,
                 kernel_initializer="glorot_uniform", bias_initializer="zeros", input_dim=None, **kwargs):
        super().__init__()
        self.units = [units] if not isinstance(units, list) else units
        if len(self.units) <= 0:
            raise ValueError("Received an invalid value for `units`, expected a positive integer, got %r." % (units,))
        self.is_batch_norm = bool(is_batch_norm)
        self.use_bias = use_bias
        self.is_dropout = is_dropput          # can never fire in the reference (is_train is never passed)
        self.activation = activation
        self._act = _activation_code(activation)
        self._kinit = _initializer(kernel_initializer)
        self._binit = _initializer(bias_initializer)
        self.built = False
        if input_dim is not None:
            self.build(input_dim)

    def build(self, last_dim):
        dims = [int(last_dim)] + list(self.units)
        for i in range(len(dims) - 1):
            self.register_parameter("kernel_%d" % i, torch.nn.Parameter(self._kinit((dims[i], dims[i + 1]))))
            if self.use_bias:
                self.register_parameter("bias_%d" % i, torch.nn.Parameter(self._binit((dims[i + 1],))))
            if self.is_batch_norm:
                self.add_module("bn_%d" % i, BatchNormalization(input_dim=dims[i + 1]))
        self.built = True

    def forward(self, inputs, is_train=False):
        if not self.built:
            if inputs.shape[-1] is None:
                raise ValueError("The last dimension of the inputs to `Dense` should be defined. Found `None`.")
            self.build(inputs.shape[-1])
            self.to(inputs.device)
        x = inputs
        for i in range(len(self.units)):
            b = getattr(self, "bias_%d" % i) if self.use_bias else None
            if self.is_batch_norm:      # MatMul, BiasAdd, BatchNormalization, activation (2.FM/CustomLayers.py:74-81)
                x = Fn.LinearAct.apply(x, getattr(self, "kernel_%d" % i), b, ops.ACT_CODE[None])
                x = getattr(self, "bn_%d" % i)(x)
                if self.activation is not None:
                    x = Activation(self.activation)(x)
            else:
                x = Fn.LinearAct.apply(x, getattr(self, "kernel_%d" % i), b, self._act)
        return x


class FMRankingLayer(_FMTables, Layer):
    def __init__(self, feature_names=["item_tag1", "item_tag2", "item_tag3"], feature_dims=20, embedding_dims=16,
                 **kwargs):
        super().__init__()
        self.feature_names = feature_names
        self.feature_dims = feature_dims
        self.embedding_dims = embedding_dims
        self.bias = torch.nn.Parameter(glorot_uniform((1,)))          # Keras default for a float weight
        self.embed = Embedding(feature_dims, embedding_dims, embeddings_regularizer="l2")
        self.w = Embedding(feature_dims, 1, embeddings_regularizer="l2")

    def forward(self, inputs):
        X = assemble_index(inputs, self.feature_names)
        flag = ops.new_flag(X.device) if self.check_ids else None
        z, _ = Fn.EmbFM.apply(self.embed.embeddings, self.w.embeddings, self.bias, X, False, flag)
        self._raise_if_oob(flag)
        output = Fn.Sigmoid.apply(z).reshape(-1, 1)
        return {"output": output}


class DeepFMRankingLayer(_FMTables, Layer):
    def __init__(self, feature_names=["user_tag0", "user_tag1", "item_tag1", "item_tag2", "item_tag3"],
                 feature_dims=20, embedding_dims=16, mlp_dims=[32, 8], **kwargs):
        super().__init__()
        self.feature_names = feature_names
        self.feature_dims = feature_dims
        self.embedding_dims = embedding_dims
        self.mlp_dims = mlp_dims
        self.bias = torch.nn.Parameter(glorot_uniform((1,)))
        self.embed = Embedding(feature_dims, embedding_dims, embeddings_regularizer="l2")
        self.w = Embedding(feature_dims, 1, embeddings_regularizer="l2")
        self.MLP_layer1 = MLPLayer(units=list(mlp_dims), activation="relu",
                                   input_dim=len(feature_names) * embedding_dims)
        self.MLP_layer2 = MLPLayer(units=[1], input_dim=list(mlp_dims)[-1])

    def forward(self, inputs):
        X = assemble_index(inputs, self.feature_names)
        flag = ops.new_flag(X.device) if self.check_ids else None
        fm_part, rows = Fn.EmbFM.apply(self.embed.embeddings, self.w.embeddings, self.bias, X, True, flag)
        self._raise_if_oob(flag)
        dense_embedding = rows.reshape(rows.shape[0], -1)             # Flatten(): field-major, dim-minor
        dnn_part = self.MLP_layer2(self.MLP_layer1(dense_embedding))  # [B,1]
        return {"output": Fn.Sigmoid.apply(dnn_part, fm_part)}         # sigmoid(fm_part + dnn_part), [B,1]


class DSSMSingleTowerLayer(Layer):
    def __init__(self, feature_names=["item_tag1", "item_tag2", "item_tag3"], feature_dims=20, embedding_dims=8,
                 mlp_dims=[64, 32], final_dim=8, sharded=False, group=None, comm=None, capacity=None, **kwargs):
        super().__init__()
        self.feature_names = feature_names
        self.feature_dims = feature_dims
        self.embedding_dims = embedding_dims
        self.mlp_dims = mlp_dims
        self.final_dim = final_dim
        # sharded=True: the table is row-sharded over the process group (BASELINE config D: 100M x 64d over 8 GPUs)
        self.embed = make_embedding(feature_dims, embedding_dims, sharded, group, comm, capacity, "l2")
        self.mlp = MLPLayer(units=list(mlp_dims), activation="relu", input_dim=len(feature_names) * embedding_dims)
        self.final = MLPLayer(units=[final_dim], activation=None, input_dim=list(mlp_dims)[-1])

    def forward(self, inputs):
        # the reference accepts [B,1] (model path) and, through an exception-driven reshape, [B] (direct call,
        # 2.FM/CustomLayers.py:188-194): both are handled explicitly here
        X = assemble_index(inputs, self.feature_names)
        flag = ops.new_flag(X.device) if self.check_ids else None
        e = self.embed(X, flag)
        self._raise_if_oob(flag)
        x = e.reshape(e.shape[0], -1)
        x = self.final(self.mlp(x))
        return {"user_id": inputs.get("user_id", None), "item_id": inputs.get("item_id", None), "output": x}


class DSSMTwoTowerRetrievalLayer(Layer):
    def __init__(self, u_feature_names=["user_tag1", "user_tag2"],
                 i_feature_names=["item_tag1", "item_tag2", "item_tag3"], u_feature_dims=20, i_feature_dims=20,
                 u_embedding_dims=8, i_embedding_dims=8, u_mlp_dims=[64, 32], i_mlp_dims=[64, 32], final_dim=8,
                 sharded=False, group=None, comm=None, u_capacity=None, i_capacity=None, **kwargs):
        super().__init__()
        # sharded: True (both towers' tables), or "u" / "i" / ("u", "i")
        which = ("u", "i") if sharded is True else ((sharded,) if isinstance(sharded, str) else tuple(sharded or ()))
        self.u_tower = DSSMSingleTowerLayer(feature_names=u_feature_names, feature_dims=u_feature_dims,
                                            embedding_dims=u_embedding_dims, mlp_dims=u_mlp_dims, final_dim=final_dim,
                                            sharded="u" in which, group=group, comm=comm, capacity=u_capacity)
        self.i_tower = DSSMSingleTowerLayer(feature_names=i_feature_names, feature_dims=i_feature_dims,
                                            embedding_dims=i_embedding_dims, mlp_dims=i_mlp_dims, final_dim=final_dim,
                                            sharded="i" in which, group=group, comm=comm, capacity=i_capacity)

    def forward(self, inputs):
        u_embedding = self.u_tower(inputs)["output"]
        i_embedding = self.i_tower(inputs)["output"]
        similarity = Fn.Cosine.apply(u_embedding, i_embedding)        # (1 - cos)/2, shape [B]
        return {"user_embedding": u_embedding, "item_embedding": i_embedding, "output": similarity}


# ---------------------------------------------------------------------------------------------------
# 3.DCN
# ---------------------------------------------------------------------------------------------------

class Dense(Layer):
    """tf.keras.layers.Dense(units, activation): ``kernel`` glorot-uniform, ``bias`` zeros."""

    def __init__(self, units, activation=None, input_dim=None):
        super().__init__()
        self.units = units
        self._act = _activation_code(activation)
        self.built = False
        if input_dim is not None:
            self.build(input_dim)

    def build(self, last_dim):
        self.kernel = torch.nn.Parameter(glorot_uniform((int(last_dim), self.units)))
        self.bias = torch.nn.Parameter(torch.zeros(self.units))
        self.built = True

    def forward(self, x):
        if not self.built:
            self.build(x.shape[-1])
            self.to(x.device)
        return Fn.LinearAct.apply(x, self.kernel, self.bias, self._act)


class DenseLayer(Layer):
    def __init__(self, units, activation, input_dim=None):
        super().__init__()
        dims = [input_dim] + list(units)
        self.hidden_layer = torch.nn.ModuleList(
            [Dense(u, activation=activation, input_dim=dims[i]) for i, u in enumerate(units)])

    def forward(self, inputs, **kwargs):
        x = inputs
        for layer in self.hidden_layer:
            x = layer(x)
        return x


class _CrossBase(Layer):
    def __init__(self, layer_num, reg_w=1e-4, reg_b=1e-4, input_dim=None):
        super().__init__()
        self.layer_num = layer_num
        self.reg_w = reg_w          # l2 regularisers never reach the loss in the reference's loop
        self.reg_b = reg_b
        self.built = False
        if input_dim is not None:
            self.build(input_dim)

    def _maybe_build(self, x):
        if not self.built:
            self.build(x.shape[1])
            self.to(x.device)


class CrossLayer(_CrossBase):
    """x_{l+1} = x0 * (x_l^T w_l) + b_l + x_l; w_l, b_l: [D,1] (w ~ N(0,0.05), b = 0)."""

    def build(self, D):
        for i in range(self.layer_num):
            self.register_parameter("w%d" % i, torch.nn.Parameter(torch.randn((D, 1), generator=_init_gen) * 0.05))
            self.register_parameter("b%d" % i, torch.nn.Parameter(torch.zeros((D, 1))))
        self.built = True

    def forward(self, inputs, **kwargs):
        self._maybe_build(inputs)
        w = torch.cat([getattr(self, "w%d" % i).reshape(1, -1) for i in range(self.layer_num)], dim=0)
        b = torch.cat([getattr(self, "b%d" % i).reshape(1, -1) for i in range(self.layer_num)], dim=0)
        return Fn.CrossVec.apply(inputs, w, b)


class MatrixCrossLayer(_CrossBase):
    """x_{l+1} = x0 (.) (W_l x_l + b_l) + x_l; W_l: [D,D] ~ N(0,0.05), b_l: [D,1] = 0."""

    def build(self, D):
        for i in range(self.layer_num):
            self.register_parameter("w%d" % i, torch.nn.Parameter(torch.randn((D, D), generator=_init_gen) * 0.05))
            self.register_parameter("b%d" % i, torch.nn.Parameter(torch.zeros((D, 1))))
        self.built = True

    def forward(self, inputs, **kwargs):
        self._maybe_build(inputs)
        W = torch.stack([getattr(self, "w%d" % i) for i in range(self.layer_num)], dim=0)
        b = torch.cat([getattr(self, "b%d" % i).reshape(1, -1) for i in range(self.layer_num)], dim=0)
        return Fn.CrossMat.apply(inputs, W, b)


class ConcatCols(torch.autograd.Function):
    """tf.concat(axis=1) of 2-D fp32 blocks, by column-block copies."""

    @staticmethod
    def forward(ctx, *blocks):
        widths = [b.shape[1] for b in blocks]
        out = torch.empty((blocks[0].shape[0], sum(widths)), dtype=torch.float32, device=blocks[0].device)
        c = 0
        for b, w in zip(blocks, widths):
            ops.copy_cols(b.contiguous(), out[:, c:c + w])
            c += w
        ctx.widths = widths
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        outs = []
        c = 0
        for i, w in enumerate(ctx.widths):
            if ctx.needs_input_grad[i]:                  # input features (the continuous columns) need no gradient
                blk = torch.empty((g.shape[0], w), dtype=torch.float32, device=g.device)
                ops.copy_cols(g[:, c:c + w], blk)
                outs.append(blk)
            else:
                outs.append(None)
            c += w
        return tuple(outs)


def _cont_block(inputs, names, device):
    cols = []
    for n in names:
        t = inputs[n]
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(t)
        t = t.to(device=device, dtype=torch.float32)
        if t.dim() == 1:
            t = t.unsqueeze(1)
        cols.append(t.contiguous())
    if len(cols) > 1:                                     # one [B, n_cont] block: one copy into the concatenation
        return [torch.cat(cols, dim=1)]
    return cols


class DeepCrossNetworkLayer(Layer):
    def __init__(self, categorical_features=["uid", "iid", "utag1", "utag2", "utag3", "utag4", "itag1", "itag2",
                                             "itag3", "itag4"],
                 continuous_features=["itag4_origin", "itag4_square", "itag4_cube"], feature_dims=160000,
                 embedding_dims=16, units=[64, 8], activation="relu", layer_num=3, reg_w=1e-4, reg_b=1e-4,
                 type="vec", sharded=False, group=None, comm=None, capacity=None):
        super().__init__()
        D = len(continuous_features) + len(categorical_features) * embedding_dims
        if type == "vec":
            self.cross_layer = CrossLayer(layer_num, reg_w, reg_b, input_dim=D)
        else:
            self.cross_layer = MatrixCrossLayer(layer_num, reg_w, reg_b, input_dim=D)
        self.dense_layer = DenseLayer(units, activation, input_dim=D)
        self.embedding_layer = make_embedding(feature_dims, embedding_dims, sharded, group, comm, capacity)
        self.categorical_features = categorical_features
        self.continuous_features = continuous_features
        self.output_layer = Dense(1, activation="sigmoid", input_dim=D + list(units)[-1])

    def forward(self, inputs):
        X = assemble_index(inputs, self.categorical_features)
        flag = ops.new_flag(X.device) if self.check_ids else None
        X_emb = self.embedding_layer(X, flag)
        self._raise_if_oob(flag)
        X_flatten = X_emb.reshape(X_emb.shape[0], -1)
        cont = _cont_block(inputs, self.continuous_features, X.device)
        _input = ConcatCols.apply(*cont, X_flatten)                   # continuous FIRST (:259)
        cross_output = self.cross_layer(_input)
        dnn_output = self.dense_layer(_input)
        combine_output = ConcatCols.apply(cross_output, dnn_output)
        return {"output": self.output_layer(combine_output)}


# ---------------------------------------------------------------------------------------------------
# 5.DIN
# ---------------------------------------------------------------------------------------------------

class Dice(Layer):
    """Dice (5.DIN/CustomLayers.py:183-196): p = sigmoid(BN(x)) with BatchNormalization(center=False, scale=False),
    out = alpha*(1-p)*x + p*x, alpha init 0.  The BN runs on its moving statistics (inference mode: mean 0,
    variance 1 at init) -- the oracle is pinned to that mode; batch statistics inside
    tf.vectorized_map are not reproduced."""

    def __init__(self, axis=-1, epsilon=1e-9, input_dim=None):
        super().__init__()
        self.axis, self.epsilon = axis, epsilon
        self.built = False
        if input_dim is not None:
            self.build(input_dim)

    def build(self, n):
        self.alpha = torch.nn.Parameter(torch.zeros(int(n)))
        self.register_buffer("moving_mean", torch.zeros(int(n)))
        self.register_buffer("moving_variance", torch.ones(int(n)))
        self.built = True

    def forward(self, x):
        if not self.built:
            self.build(x.shape[-1])
            self.to(x.device)
        return Fn.FeatAct.apply(x, ops.DACT_DICE, self.alpha, self.moving_mean, self.moving_variance)


class PReLU(Layer):
    """tf.keras.layers.PReLU(): max(0,x) + alpha*min(0,x), alpha per feature, init 0."""

    def __init__(self, input_dim=None):
        super().__init__()
        self.built = False
        if input_dim is not None:
            self.build(input_dim)

    def build(self, n):
        self.alpha = torch.nn.Parameter(torch.zeros(int(n)))
        self.built = True

    def forward(self, x):
        if not self.built:
            self.build(x.shape[-1])
            self.to(x.device)
        return Fn.FeatAct.apply(x, ops.DACT_PRELU, self.alpha, None, None)


class Activation(Layer):
    """tf.keras.layers.Activation(name or callable layer)."""

    def __init__(self, activation):
        super().__init__()
        if isinstance(activation, torch.nn.Module):
            self.inner, self.kind = activation, None
        else:
            if activation not in ops.DACT_CODE:
                raise ValueError("Unknown activation function: %r" % (activation,))
            self.inner, self.kind = None, ops.DACT_CODE[activation]

    def forward(self, x):
        if self.inner is not None:
            return self.inner(x)
        return Fn.FeatAct.apply(x, self.kind, None, None, None)


class LayerNormalization(Layer):
    """tf.keras.layers.LayerNormalization(): last axis, epsilon 1e-3, gamma 1, beta 0."""

    def __init__(self, input_dim):
        super().__init__()
        self.gamma = torch.nn.Parameter(torch.ones(int(input_dim)))
        self.beta = torch.nn.Parameter(torch.zeros(int(input_dim)))

    def forward(self, x):
        return Fn.LayerNorm.apply(x, self.gamma, self.beta)


class SoftmaxDense(Layer):
    """Dense(units, activation='softmax')."""

    def __init__(self, units, input_dim):
        super().__init__()
        self.dense = Dense(units, activation=None, input_dim=input_dim)

    def forward(self, x):
        return Fn.Softmax.apply(self.dense(x))


class Sequential(Layer):
    def __init__(self, layers_):
        super().__init__()
        self.layers = torch.nn.ModuleList(layers_)

    def forward(self, x):
        for l in self.layers:
            x = l(x)
        return x


def make_mlp_layer(units, activation="PReLU", normalization="layernorm", softmax_units=-1, sigmoid_units=False,
                   input_dim=None):
    """5.DIN/CustomLayers.py:142-160.  ``input_dim`` (extension) is needed because layers are built eagerly."""
    if input_dim is None:
        raise ValueError("make_mlp_layer needs input_dim")
    seq = []
    d = input_dim
    for unit in units:
        seq.append(Dense(unit, input_dim=d))
        d = unit
        if normalization == "batchnorm":
            raise NotImplementedError("normalization='batchnorm' is not on the hot path")
        elif normalization == "layernorm":
            seq.append(LayerNormalization(d))
        if activation == "PReLU":
            seq.append(PReLU(d))
        elif activation == "Dice":
            seq.append(Dice(input_dim=d))
        elif isinstance(activation, torch.nn.Module):
            # the reference wraps a Dice() *instance* in Activation(...) (5.DIN/CustomLayers.py:154-155,219)
            if isinstance(activation, (Dice, PReLU)) and not activation.built:
                activation.build(d)
            seq.append(Activation(activation))
        else:
            seq.append(Activation(activation))
    if softmax_units > 0:
        seq.append(SoftmaxDense(softmax_units, d))
    elif sigmoid_units:
        seq.append(Dense(1, activation="sigmoid", input_dim=d))
    return Sequential(seq)


class DinActivationLayer(Layer):
    """5.DIN/CustomLayers.py:163-180: Dense(36) -> activation -> Dense(1) over [q, q-k, k, vec(k q^T)].
    ``call((vec1, vec2))`` scores ONE key per example like the reference; DINLayer uses the batched, fused
    attention kernel with the same parameters."""

    def __init__(self, activation="PReLU", input_dim=None, **kwargs):
        super().__init__()
        self._activation_spec = (activation,)     # in a tuple: a Dice() instance must register under mlp_layer only
        self.hidden = 36
        self.built = False
        if input_dim is not None:
            self.build(input_dim)

    def build(self, D):
        D = int(D)
        self.D = D
        self.mlp_layer = make_mlp_layer([self.hidden], activation=self._activation_spec[0], normalization="none",
                                        input_dim=3 * D + D * D)
        self.output_layer = Dense(1, input_dim=self.hidden)
        self.built = True

    def act_params(self):
        act = self.mlp_layer.layers[1]
        inner = act.inner if isinstance(act, Activation) and act.inner is not None else act
        if isinstance(inner, Dice):
            return ops.DACT_DICE, inner.alpha, inner.moving_mean, inner.moving_variance
        if isinstance(inner, PReLU):
            return ops.DACT_PRELU, inner.alpha, None, None
        return inner.kind, None, None, None

    def attend(self, embed, q, series, padding_index, mask_valid, oob=None, sink=None):
        """All T keys of every example at once: pooled [B,D], raw scores [B,T]."""
        dense = self.mlp_layer.layers[0]
        kind, alpha, mean, var = self.act_params()
        return Fn.DinAttention.apply(embed, q, series, dense.kernel, dense.bias, kind, alpha, mean, var,
                                     self.output_layer.kernel, self.output_layer.bias, padding_index, mask_valid, oob,
                                     sink)

    def forward(self, inputs):
        vec1, vec2 = inputs                       # q [B,D], k [B,D]
        if not self.built:
            self.build(vec1.shape[1])
            self.to(vec1.device)
        # a single key per example = a length-1 "series" of already-gathered rows: gather from an identity table
        B, D = vec2.shape
        ids = torch.arange(B, device=vec2.device, dtype=torch.int64).reshape(B, 1, 1)
        pooled, scores = self.attend_rows(vec1, vec2, ids)
        return scores.reshape(B, 1)

    def attend_rows(self, q, keys2d, ids):
        dense = self.mlp_layer.layers[0]
        kind, alpha, mean, var = self.act_params()
        # mask_valid=1 with padding_index=-1: no position is masked
        return Fn.DinAttention.apply(keys2d.contiguous(), q, ids, dense.kernel, dense.bias, kind, alpha, mean, var,
                                     self.output_layer.kernel, self.output_layer.bias, -1, 1, None)


class DINLayer(Layer):
    """5.DIN/CustomLayers.py:199-289.  ``mask_mode='reference'`` reproduces the reference's mask (only PADDED
    positions contribute, :256,277-278); ``'valid'`` is the intended convention."""

    def __init__(self, user_and_context_categorical_features=["uid", "utag1", "utag2", "utag3", "utag4"],
                 item_categorical_features=["i_goods_id", "i_shop_id", "i_cate_id"],
                 behavior_series_features=["visited_goods_ids", "visited_shop_ids", "visited_cate_ids"],
                 continuous_features=["itag4_origin", "itag4_square", "itag4_cube"], feature_dims=160000,
                 embedding_dims=16, activation="Dice", padding_index=0, mask_mode="reference", sharded=False, group=None,
                 comm=None, capacity=None):
        super().__init__()
        self.user_and_context_categorical_features = user_and_context_categorical_features
        self.item_categorical_features = item_categorical_features
        assert len(item_categorical_features) == len(behavior_series_features), \
            "Features to be interacted should match in item and behavior series"
        self.behavior_series_features = behavior_series_features
        self.continuous_features = continuous_features
        self.feature_dims = feature_dims
        self.embedding_dims = embedding_dims
        if mask_mode not in ("reference", "valid"):
            raise ValueError("mask_mode must be 'reference' or 'valid'")
        self.mask_mode = mask_mode
        if sharded and padding_index != 0:
            raise NotImplementedError("sharded DINLayer: padding_index must be 0 (the smallest id: its row then sits in "
                                      "slot 0 of the exchanged rows, which is what the attention kernel's mask compares)")
        # sharded=True: the table is row-sharded over the process group (BASELINE config E: 50M x 32d over 8 GPUs)
        self.embed = make_embedding(feature_dims, embedding_dims, sharded, group, comm, capacity)
        D = len(item_categorical_features) * embedding_dims
        self.din_activation_layer = DinActivationLayer(
            activation=Dice() if activation == "Dice" else activation, input_dim=D)
        n_profile = len(user_and_context_categorical_features) + len(item_categorical_features)
        self.mlp = make_mlp_layer([200, 80], activation=activation, softmax_units=2,
                                  input_dim=n_profile * embedding_dims + D)
        self.padding_index = padding_index

    def _series(self, inputs, device):
        series_cols = []
        for name in self.behavior_series_features:
            t = inputs[name]
            if not isinstance(t, torch.Tensor):
                t = torch.as_tensor(t)
            t = t.to(device=device, dtype=torch.int64).contiguous()
            if t.dim() != 2:
                raise ValueError("behaviour series %r must have shape [B,T]" % name)
            series_cols.append(t)
        B, T = series_cols[0].shape
        return ops.index_pack(series_cols).reshape(B, T, len(series_cols))        # tf.stack(axis=2)

    def forward(self, inputs):
        prof_names = self.user_and_context_categorical_features + self.item_categorical_features
        X_cate = assemble_index(inputs, prof_names)
        flag = ops.new_flag(X_cate.device) if self.check_ids else None
        n_item = len(self.item_categorical_features)
        mask_valid = 1 if self.mask_mode == "valid" else 0
        if _is_sharded(self.embed):
            # row-sharded table: ONE de-duplicated exchange for the profile ids, the behaviour series and the padding id
            # (always part of the lookup: as the smallest id it is unique 0 of owner 0 = slot 0 of the rows that come
            # back, so the attention kernel's mask `first series id == padding_index` becomes `slot == 0`); every
            # consumer then works on the local [P*cap, E] rows buffer with slots in place of ids
            series = self._series(inputs, X_cate.device)
            pad = torch.full((1,), self.padding_index, dtype=torch.int64, device=X_cate.device)
            n1, n2 = X_cate.numel(), series.numel()
            rows, slot = self.embed.exchange(torch.cat([X_cate.reshape(-1), series.reshape(-1), pad]), flag)
            s_cate = slot[:n1].reshape(X_cate.shape).contiguous()
            s_series = slot[n1:n1 + n2].reshape(series.shape).contiguous()
            # (the consumers' gradients -- profile rows, series values, a zero row for the padding lookup -- line up with
            # the exchange's id list, whose de-duplication plan the backward reuses)
            sink = self.embed.grad_sink(X_cate, n_tail=1)
            profile = self.embed.take(rows, s_cate, sink)
            profile_output = profile.reshape(profile.shape[0], -1)
            q = profile[:, profile.shape[1] - n_item:, :].reshape(profile.shape[0], -1)
            pooled, _ = self.din_activation_layer.attend(rows, q, s_series, 0, mask_valid, None, sink)
            self._raise_if_oob(flag)
            X_combined = ConcatCols.apply(profile_output, pooled)
            return {"output": self.mlp(X_combined)}
        # one lookup serves both: the candidate item's rows are the tail of the profile rows (the reference looks the
        # item ids up a second time, 5.DIN/CustomLayers.py:244-247 -- same values), and the series lookups of the
        # attention share the profile lookup's de-duplication (functional.GradSink)
        sink = self.embed.grad_sink(X_cate)
        profile = self.embed(X_cate, flag, sink)
        profile_output = profile.reshape(profile.shape[0], -1)
        q = profile[:, profile.shape[1] - n_item:, :].reshape(profile.shape[0], -1)
        series = self._series(inputs, X_cate.device)
        pooled, _ = self.din_activation_layer.attend(self.embed.embeddings, q, series, self.padding_index,
                                                     mask_valid, flag, sink)
        self._raise_if_oob(flag)
        X_combined = ConcatCols.apply(profile_output, pooled)
        return {"output": self.mlp(X_combined)}


# ---------------------------------------------------------------------------------------------------
# sibling layers on the same gather
# ---------------------------------------------------------------------------------------------------

class BatchNormalization(Layer):
    """tf.keras.layers.BatchNormalization() on [B,N]: axis=-1, momentum=0.99, epsilon=1e-3, gamma ones, beta zeros,
    moving mean 0 / variance 1.  Like Keras it normalises with BATCH statistics (and updates the moving averages)
    when the module is in training mode -- the reference's train loops call ``model(inputs, training=True)``
    (3.DCN/ModelManager.py:192) -- and with the moving statistics in eval mode."""

    def __init__(self, momentum=0.99, epsilon=1e-3, center=True, scale=True, input_dim=None):
        super().__init__()
        self.momentum, self.epsilon, self.center, self.scale = momentum, epsilon, center, scale
        self.built = False
        if input_dim is not None:
            self.build(input_dim)

    def build(self, n):
        n = int(n)
        self.gamma = torch.nn.Parameter(torch.ones(n)) if self.scale else None
        self.beta = torch.nn.Parameter(torch.zeros(n)) if self.center else None
        self.register_buffer("moving_mean", torch.zeros(n))
        self.register_buffer("moving_variance", torch.ones(n))
        self.built = True

    def forward(self, x):
        if not self.built:
            self.build(x.shape[-1])
            self.to(x.device)
        return Fn.BatchNorm.apply(x, self.gamma, self.beta, self.moving_mean, self.moving_variance, self.training,
                                  self.epsilon, self.momentum)


class PNNLayer(Layer):
    """2.FM/CustomLayers.py:696-745 with method='inner' (IpnLayer, :773-792): embed -> [Flatten | pairwise inner
    products] -> MLP(relu) -> MLP([1], sigmoid).  The lookup, the flatten and the F(F-1)/2 inner products are one
    kernel; its output IS ``combined_vector``.  method='outer' (OpnLayer) is not on the hot path."""

    def __init__(self, feature_names=["user_tag0", "user_tag1", "item_tag1", "item_tag2", "item_tag3"], feature_dims=20,
                 embedding_dims=16, mlp_dims=[32, 8], dropout=0, method="inner", kernel_type=None, **kwargs):
        super().__init__()
        assert method in ("inner", "outer")
        if method == "outer":
            raise NotImplementedError("PNNLayer(method='outer') is outside the accelerated path")
        self.feature_names = feature_names
        self.feature_dims = feature_dims
        self.fields_cnt = len(feature_names)
        self.embedding_dims = embedding_dims
        self.mlp_dims = mlp_dims
        self.method, self.dropout, self.kernel_type = method, dropout, kernel_type
        F = self.fields_cnt
        self.embed = Embedding(feature_dims, embedding_dims, embeddings_regularizer="l2")
        self.MLP_layer1 = MLPLayer(units=mlp_dims, activation="relu", is_dropput=dropout,
                                   input_dim=F * embedding_dims + F * (F - 1) // 2)
        self.MLP_layer2 = MLPLayer(units=[1], activation="sigmoid", input_dim=list(mlp_dims)[-1])

    def forward(self, inputs):
        X = assemble_index(inputs, self.feature_names)
        flag = ops.new_flag(X.device) if self.check_ids else None
        combined_vector = Fn.EmbIpn.apply(self.embed.embeddings, X, flag)
        self._raise_if_oob(flag)
        output = self.MLP_layer2(self.MLP_layer1(combined_vector))
        return {"output": output}


class NeuralFactorizationMachineLayer(Layer):
    """3.DCN/CustomLayers.py:451-509: bi-interaction pooling of the categorical embeddings ++ continuous features ->
    BatchNormalization -> MLP(units, activation) -> MLP([1], sigmoid).  (The first-order table ``w`` is created by
    the reference but never used, :491-495; it is not allocated here.)"""

    def __init__(self, categorical_features=["uid", "iid", "utag1", "utag2", "utag3", "utag4", "itag1", "itag2",
                                             "itag3", "itag4"],
                 continuous_features=["itag4_origin", "itag4_square", "itag4_cube"], feature_dims=160000,
                 embedding_dims=16, units=[64, 8], activation="relu"):
        super().__init__()
        self.categorical_features = categorical_features
        self.continuous_features = continuous_features
        self.feature_dims = feature_dims
        self.embedding_dims = embedding_dims
        n = embedding_dims + len(continuous_features)
        self.embed = Embedding(feature_dims, embedding_dims, embeddings_regularizer="l2")
        self.bn_layer = BatchNormalization(input_dim=n)
        self.MLP_layer1 = MLPLayer(units=units, activation=activation, input_dim=n)
        self.MLP_layer2 = MLPLayer(units=[1], activation="sigmoid", input_dim=list(units)[-1])

    def forward(self, inputs):
        X = assemble_index(inputs, self.categorical_features)
        flag = ops.new_flag(X.device) if self.check_ids else None
        cont = _cont_block(inputs, self.continuous_features, X.device)
        X_cont = torch.cat(cont, dim=1) if cont else None
        combined_vector = Fn.EmbBiInteraction.apply(self.embed.embeddings, X, X_cont, flag)
        self._raise_if_oob(flag)
        combined_vector = self.bn_layer(combined_vector)
        output = self.MLP_layer2(self.MLP_layer1(combined_vector))
        return {"output": output}


class GSULayer(Layer):
    """7.SIM/CustomLayers.py:62-127 (general search unit): target item embedding, inner-product attention over the
    embedded behaviour series with the valid mask, sum pooling, MLP [200,80] + softmax(2).  The series lookup,
    scores and pooling are one kernel, so ``X_series`` (the [B,T,D] embedded series the reference also returns) is
    only materialised when ``return_series=True``."""

    def __init__(self, item_categorical_features=["i_goods_id", "i_shop_id", "i_cate_id"],
                 behavior_series_features=["visited_goods_ids", "visited_shop_ids", "visited_cate_ids"],
                 feature_dims=1000, embedding_dims=16, activation="Dice", padding_index=0, embedding_layer=None,
                 l2_reg=0.01, return_series=False):
        super().__init__()
        self.item_categorical_features = item_categorical_features
        assert len(item_categorical_features) == len(behavior_series_features), \
            "Features to be interacted should match in item and behavior series"
        self.behavior_series_features = behavior_series_features
        self.feature_dims = feature_dims
        self.embedding_dims = embedding_dims
        self.l2_reg = l2_reg
        self.embed = embedding_layer if embedding_layer is not None else Embedding(feature_dims, embedding_dims)
        D = len(item_categorical_features) * embedding_dims
        self.mlp = make_mlp_layer([200, 80], activation=activation, softmax_units=2, input_dim=2 * D)
        self.padding_index = padding_index
        self.return_series = return_series

    def forward(self, inputs):
        X_item = assemble_index(inputs, self.item_categorical_features)
        flag = ops.new_flag(X_item.device) if self.check_ids else None
        sink = self.embed.grad_sink(X_item)        # the series lookups share the item lookup's de-duplication
        q = self.embed(X_item, flag, sink)
        q = q.reshape(q.shape[0], -1)
        series_cols = []
        for name in self.behavior_series_features:
            t = inputs[name]
            if not isinstance(t, torch.Tensor):
                t = torch.as_tensor(t)
            t = t.to(device=X_item.device, dtype=torch.int64).contiguous()
            if t.dim() != 2:
                raise ValueError("behaviour series %r must have shape [B,T]" % name)
            series_cols.append(t)
        B, T = series_cols[0].shape
        series = ops.index_pack(series_cols).reshape(B, T, len(series_cols))      # tf.stack(axis=2)
        pooled, _ = Fn.IpAttention.apply(self.embed.embeddings, q, series, self.padding_index, flag, sink)
        self._raise_if_oob(flag)
        X_combined = ConcatCols.apply(q, pooled)
        result = {"output": self.mlp(X_combined), "valid_mask": series_cols[0] != self.padding_index}
        if self.return_series:
            result["X_series"] = self.embed(series.reshape(B, T * len(series_cols))).reshape(B, T, -1)
        return result


class FieldAwareInteractionLayer(Layer):
    """Holder of the field-aware table ``v`` [feature_dims, fields_cnt, embedding_dims] (2.FM/CustomLayers.py:428-434):
    v[id, c, :] is the vector id uses against field c.  The interaction itself is fused into Fn.FFM."""

    def __init__(self, fields_cnt, feature_dims=20, embedding_dims=16, **kwargs):
        super().__init__()
        self.v = torch.nn.Parameter(_uniform((feature_dims, fields_cnt, embedding_dims), 0.05))


class FFMLayer(Layer):
    """2.FM/CustomLayers.py:465-497: sigmoid(bias + sum_f w[x_f] + sum_{a<c} <v[x_a,c,:], v[x_c,a,:]>).  (The reference
    builds its FieldAwareInteractionLayer with the DEFAULT feature_dims=20 / embedding_dims=16, ignoring the layer's
    own arguments, :476; the layer's arguments are used here.)"""

    def __init__(self, feature_names=["item_tag1", "item_tag2", "item_tag3", "user_tag0", "user_tag1"], feature_dims=20,
                 embedding_dims=16, **kwargs):
        super().__init__()
        self.feature_names = feature_names
        self.feature_dims = feature_dims
        self.fields_cnt = len(feature_names)
        self.embedding_dims = embedding_dims
        self.bias = torch.nn.Parameter(_uniform((1,), 0.05))
        self.w = torch.nn.Parameter(_uniform((feature_dims, 1), 0.05))
        self.fa_interaction_layer = FieldAwareInteractionLayer(self.fields_cnt, feature_dims, embedding_dims)

    def logit(self, inputs):
        X = assemble_index(inputs, self.feature_names)
        flag = ops.new_flag(X.device) if self.check_ids else None
        z = Fn.FFM.apply(self.fa_interaction_layer.v, self.w, self.bias, X, flag)
        self._raise_if_oob(flag)
        return z

    def forward(self, inputs):
        z = self.logit(inputs)
        return {"output": Fn.Sigmoid.apply(z).reshape(-1, 1)}


class FFMRankingLayer(FFMLayer):
    """2.FM/CustomLayers.py:370-425, the loop form: ``embedding_list[i]`` is the table used against field i and
    ebd_out[i][:, j] * ebd_out[j][:, i] = table_i[x_j] * table_j[x_i].  Same numbers as FFMLayer with
    v[id, i, :] = embedding_list[i][id, :]; the F tables are kept interleaved per id (one id's F vectors contiguous)
    and exposed as strided views."""

    def __init__(self, feature_names=["item_tag1", "item_tag2", "item_tag3"], feature_dims=20, embedding_dims=16,
                 **kwargs):
        super().__init__(feature_names=feature_names, feature_dims=feature_dims, embedding_dims=embedding_dims)

    @property
    def embedding_list(self):
        v = self.fa_interaction_layer.v
        return [v[:, i, :] for i in range(self.fields_cnt)]


class PNNRankingLayer(PNNLayer):
    """2.FM/CustomLayers.py:536-598 (the loop form, InnerProductNetwork :601-624): same numbers as PNNLayer."""
