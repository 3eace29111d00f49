import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests.test_gpu_engine import make16, oracle_grads
from explicit_tf2_recommendation_amd import engine, data
from oracle import layers_np as L
for (B, F, V, dist) in [(8192, 26, 1000000, "uniform"), (2048, 26, 200000, "zipf"), (64, 26, 1000, "uniform")]:
    layer, names, gen = make16(B, F, V, 11, dist)
    step = engine.DeepFMFusedStep(layer, B, gen.dims, gen.offsets, use_graph=False)
    gstep = engine.DeepFMTrainStep(layer, B, use_graph=False)
    batch = gen.batch(B)
    db = data.to_device(batch)
    loss = step(db).item(); l2 = gstep(db).item()
    ref_loss, ref = oracle_grads(layer, names, batch)
    print(B, F, "loss", loss, l2, ref_loss)
    g = step.gradients(); g2 = gstep.gradients()
    for name in ("MLP_layer1.kernel_0", "MLP_layer1.bias_0", "MLP_layer1.kernel_1", "MLP_layer1.bias_1", "MLP_layer2.kernel_0", "MLP_layer2.bias_0", "bias"):
        a = g[name].cpu().numpy().astype(np.float64); b = ref[name]; c = g2[name].cpu().numpy().astype(np.float64)
        d = np.abs(a - b); d2 = np.abs(c - b)
        i = np.unravel_index(d.argmax(), d.shape)
        print("  %-22s max|ref| %.3e  fused err %.3e at %s (ref %.4e got %.4e)  generic err %.3e" % (name, np.abs(b).max(), d.max(), i, b[i], a[i], d2.max()))
