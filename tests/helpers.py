"""Shared test helpers: deterministic parameter builders used by oracle and GPU parity tests."""
import numpy as np


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def glorot(r, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return r.uniform(-lim, lim, size=(fan_in, fan_out)).astype(np.float32)


def det_table(V, E):
    """Hand-checkable table of SURVEY.md 8c: table[i,d] = ((i*31 + d*17) mod 97 - 48)/480."""
    i = np.arange(V, dtype=np.int64)[:, None]
    d = np.arange(E, dtype=np.int64)[None, :]
    return (((i * 31 + d * 17) % 97 - 48) / 480.0).astype(np.float32)


def deepfm_params(seed, V, F, E, mlp_dims=(32, 8), scale=0.05):
    r = rng(seed)
    dims = [F * E] + list(mlp_dims)
    return {
        "embed": r.uniform(-scale, scale, size=(V, E)).astype(np.float32),
        "w": r.uniform(-scale, scale, size=(V, 1)).astype(np.float32),
        "bias": r.uniform(-1, 1, size=(1,)).astype(np.float32),
        "k1": [glorot(r, dims[i], dims[i + 1]) for i in range(len(dims) - 1)],
        "b1": [r.uniform(-0.1, 0.1, size=(dims[i + 1],)).astype(np.float32) for i in range(len(dims) - 1)],
        "k2": [glorot(r, dims[-1], 1)],
        "b2": [r.uniform(-0.1, 0.1, size=(1,)).astype(np.float32)],
    }


def tower_params(seed, V, F, E, mlp_dims=(64, 32), final_dim=8, scale=0.05):
    r = rng(seed)
    dims = [F * E] + list(mlp_dims)
    return {
        "embed": r.uniform(-scale, scale, size=(V, E)).astype(np.float32),
        "mlp_k": [glorot(r, dims[i], dims[i + 1]) for i in range(len(dims) - 1)],
        "mlp_b": [r.uniform(-0.1, 0.1, size=(dims[i + 1],)).astype(np.float32) for i in range(len(dims) - 1)],
        "final_k": [glorot(r, dims[-1], final_dim)],
        "final_b": [r.uniform(-0.1, 0.1, size=(final_dim,)).astype(np.float32)],
    }


def dcn_params(seed, V, F, E, n_cont=3, units=(64, 8), layer_num=3, kind="vec"):
    r = rng(seed)
    D = n_cont + F * E
    dims = [D] + list(units)
    if kind == "vec":
        cw = [r.normal(0, 0.05, size=(D, 1)).astype(np.float32) for _ in range(layer_num)]
    else:
        cw = [r.normal(0, 0.05, size=(D, D)).astype(np.float32) for _ in range(layer_num)]
    return {
        "embed": r.uniform(-0.05, 0.05, size=(V, E)).astype(np.float32),
        "cross_w": cw,
        "cross_b": [r.normal(0, 0.02, size=(D, 1)).astype(np.float32) for _ in range(layer_num)],
        "dnn_k": [glorot(r, dims[i], dims[i + 1]) for i in range(len(dims) - 1)],
        "dnn_b": [r.uniform(-0.1, 0.1, size=(dims[i + 1],)).astype(np.float32) for i in range(len(dims) - 1)],
        "out_k": glorot(r, D + units[-1], 1),
        "out_b": r.uniform(-0.1, 0.1, size=(1,)).astype(np.float32),
    }


def din_params(seed, V, E, n_user=5, n_item=3, act="dice", H=36, mlp_units=(200, 80)):
    r = rng(seed)
    D = n_item * E

    def mk_act(width):
        if act == "dice":
            return {"kind": "dice", "alpha": r.uniform(-0.2, 0.2, size=(width,)).astype(np.float32),
                    "mean": r.uniform(-0.1, 0.1, size=(width,)).astype(np.float32),
                    "var": r.uniform(0.5, 1.5, size=(width,)).astype(np.float32)}
        if act == "prelu":
            return {"kind": "prelu", "alpha": r.uniform(-0.2, 0.3, size=(width,)).astype(np.float32)}
        return {"kind": act}

    in_dim = (n_user + n_item) * E + D
    dims = [in_dim] + list(mlp_units)
    mlp = []
    for i in range(len(mlp_units)):
        mlp.append({"K": glorot(r, dims[i], dims[i + 1]),
                    "b": r.uniform(-0.1, 0.1, size=(dims[i + 1],)).astype(np.float32),
                    "gamma": r.uniform(0.8, 1.2, size=(dims[i + 1],)).astype(np.float32),
                    "beta": r.uniform(-0.1, 0.1, size=(dims[i + 1],)).astype(np.float32),
                    "act": mk_act(dims[i + 1])})
    return {
        "embed": r.uniform(-0.5, 0.5, size=(V, E)).astype(np.float32),
        "att": {"W1": glorot(r, 3 * D + D * D, H), "b1": r.uniform(-0.1, 0.1, size=(H,)).astype(np.float32),
                "act": mk_act(H), "W2": glorot(r, H, 1), "b2": r.uniform(-0.1, 0.1, size=(1,)).astype(np.float32)},
        "mlp": mlp,
        "out_k": glorot(r, dims[-1], 2),
        "out_b": r.uniform(-0.1, 0.1, size=(2,)).astype(np.float32),
    }


def nfm_params(seed, V, E, n_cont=3, units=(64, 8), scale=0.3):
    """NeuralFactorizationMachineLayer (3.DCN/CustomLayers.py:451-474): embed, BatchNormalization(E+n_cont), MLP."""
    r = rng(seed)
    n = E + n_cont
    dims = [n] + list(units)
    return {
        "embed": r.uniform(-scale, scale, size=(V, E)).astype(np.float32),
        "bn_gamma": r.uniform(0.5, 1.5, size=(n,)).astype(np.float32),
        "bn_beta": r.uniform(-0.2, 0.2, size=(n,)).astype(np.float32),
        "bn_mean": r.uniform(-0.1, 0.1, size=(n,)).astype(np.float32),
        "bn_var": r.uniform(0.5, 1.5, size=(n,)).astype(np.float32),
        "k1": [glorot(r, dims[i], dims[i + 1]) for i in range(len(dims) - 1)],
        "b1": [r.uniform(-0.1, 0.1, size=(dims[i + 1],)).astype(np.float32) for i in range(len(dims) - 1)],
        "k2": [glorot(r, dims[-1], 1)],
        "b2": [r.uniform(-0.1, 0.1, size=(1,)).astype(np.float32)],
    }


def pnn_params(seed, V, F, E, mlp_dims=(32, 8), scale=0.3):
    """PNNLayer, method='inner' (2.FM/CustomLayers.py:705-727)."""
    r = rng(seed)
    dims = [F * E + F * (F - 1) // 2] + list(mlp_dims)
    return {
        "embed": r.uniform(-scale, scale, size=(V, E)).astype(np.float32),
        "k1": [glorot(r, dims[i], dims[i + 1]) for i in range(len(dims) - 1)],
        "b1": [r.uniform(-0.1, 0.1, size=(dims[i + 1],)).astype(np.float32) for i in range(len(dims) - 1)],
        "k2": [glorot(r, dims[-1], 1)],
        "b2": [r.uniform(-0.1, 0.1, size=(1,)).astype(np.float32)],
    }


def to_torch(obj, dtype=None, requires_grad=False, device=None):
    import torch
    if isinstance(obj, dict):
        return {k: to_torch(v, dtype, requires_grad, device) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [to_torch(v, dtype, requires_grad, device) for v in obj]
    if isinstance(obj, np.ndarray):
        t = torch.from_numpy(obj.copy())
        if t.is_floating_point():
            if dtype is not None:
                t = t.to(dtype)
            if device is not None:
                t = t.to(device)
            t.requires_grad_(requires_grad)
        elif device is not None:
            t = t.to(device)
        return t
    return obj
