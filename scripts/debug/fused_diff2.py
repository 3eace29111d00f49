import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests.test_gpu_engine import make16, oracle_grads
from explicit_tf2_recommendation_amd import engine, data
for use_graph in (False, True):
  for (B, F, V, dist) in [(1000, 3, 300, "zipf")]:
    layer, names, gen = make16(B, F, V, 11, dist)
    step = engine.DeepFMFusedStep(layer, B, gen.dims, gen.offsets, use_graph=use_graph)
    for it in range(3):
        batch = gen.batch(B)
        db = data.to_device(batch)
        loss = step(db).item()
        if use_graph: loss = step(db).item()
        ref_loss, ref = oracle_grads(layer, names, batch)
        print("graph", use_graph, "it", it, "loss", loss, ref_loss)
        g = step.gradients()
        for name in ("MLP_layer1.kernel_0", "MLP_layer1.bias_0", "MLP_layer1.kernel_1", "MLP_layer1.bias_1", "MLP_layer2.kernel_0", "MLP_layer2.bias_0", "bias"):
            a = g[name].cpu().numpy().astype(np.float64); b = ref[name]
            d = np.abs(a - b)
            i = np.unravel_index(d.argmax(), d.shape)
            print("  %-22s max|ref| %.3e err %.3e at %s (ref %.4e got %.4e)" % (name, np.abs(b).max(), d.max(), i, b[i], a[i]))
