#!/usr/bin/env python3
"""Print the figures of a bench.py JSON line that matter when iterating on a kernel."""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.load(open(path))
    except (OSError, ValueError) as e:
        print(path, "unreadable:", e)
        continue
    r, g = d.get("roofline", {}), d.get("roofline_gather", {})
    print("%s: %.1f M ex/s  %.4f ms/step  loss %.6f | fused %.2f us frac %.3f | gather %.2f us frac %.3f | keras-adam %.3f ms"
          % (path, d["value"] / 1e6, d["ms_per_step"], d.get("loss", float("nan")), r.get("avg_launch_us", 0), r.get("frac", 0),
             g.get("avg_launch_us", 0), g.get("frac", 0), d.get("train_step_keras_adam_ms", float("nan"))))
    for k in ("roofline_gather_e32", "roofline_gather_e64"):
        if k in d:
            x = d[k]
            print("   %s: %.1f us for %d lookups = %.0f GB/s (frac %.3f); at config size %.2f us (frac %.3f)"
                  % (k, x["avg_launch_us"], x["n_lookups"], x["achieved"], x["frac"], x["at_config_size"]["avg_launch_us"],
                     x["at_config_size"]["frac"]))
