#!/usr/bin/env python3
"""profiles/r03_pmc_traffic.json from the two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md's HBM section prescribes):

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o f -- python3 bench.py --steps 20 --warmup 8 \
        --no-cpu-baseline --adam-steps 0
    rocprofv3 --pmc WRITE_SIZE ... -d gpurun_out/pmc_w -o w -- (same command)
    python scripts/pmc_traffic.py gpurun_out/pmc_f/f_counter_collection.csv gpurun_out/pmc_w/w_counter_collection.csv [commit]

``commit``: the commit of the tree the passes ran on (a GPU box has no .git: pass `git rev-parse --short HEAD` from here);
bench.py prints it beside ``traffic`` so that a counter file older than the kernels is visible.

Units: KB per dispatch.  gfx950 correction: FETCH_SIZE reads exactly 1/2 of the fetched bytes, WRITE_SIZE is exact
-> hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.  Kernels that bench.py launches at two sizes (the row gathers: 2^20
lookups and the config's batch) are taken at their LARGEST grid, the one the roofline records quote.
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEEP = ["colsort_onewg_kernel", "deepfm3_kernel", "deepfm_post_direct_kernel", "emb_fm_fwd_vec_kernel",
        "gather_rows_kernel_e32", "gather_rows_kernel_e64"]


def key_of(name):
    m = re.search(r"gather_rows_kernel<(\d+)", name)
    if m:
        return "gather_rows_kernel_e%d" % (4 * int(m.group(1)))
    m = re.search(r"(\w+_kernel)\b", name)
    return m.group(1) if m else None        # template arguments dropped: deepfm3_kernel<true, true, false> -> deepfm3_kernel


def per_kernel(path, counter):
    rows = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = key_of(r["Kernel_Name"])
        if k not in KEEP:
            continue
        rows.setdefault(k, []).append((int(r.get("Grid_Size", 0) or 0), float(r["Counter_Value"])))
    acc = {}
    for k, lst in rows.items():
        gmax = max(g for g, _ in lst)
        vals = [v for g, v in lst if g == gmax]
        acc[k] = (sum(vals), len(vals))
    return acc


def main(fetch_csv, write_csv, commit=None):
    f, w = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, bench.py --steps 20 --warmup 8 "
                   "(1 x MI355X, DeepFM 10M x 16d, B=8192; row gathers at 2^20 lookups). Units: KB per dispatch. gfx950 "
                   "correction (MI355X_MICROARCH.md, section HBM): FETCH_SIZE reads exactly 1/2 of the fetched bytes, "
                   "WRITE_SIZE is exact -> hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024. Calibration on a known count: the "
                   "gather + FM kernel reads 212,992 lookups x one 128-B line + 1.7 MB of ids = 29.0 MB.",
           "commit": commit, "kernels": {}}
    for k in KEEP:
        if k in f and k in w:
            fa, wa = f[k][0] / f[k][1], w[k][0] / w[k][1]
            out["kernels"][k] = {"FETCH_SIZE_KB_avg": round(fa, 1), "dispatches_fetch": f[k][1],
                                 "WRITE_SIZE_KB_avg": round(wa, 1), "dispatches_write": w[k][1],
                                 "hbm_bytes_per_launch_corrected": int((2 * fa + wa) * 1024)}
    json.dump(out, open(os.path.join(ROOT, "profiles", "r03_pmc_traffic.json"), "w"), indent=1)
    for k, v in out["kernels"].items():
        print("%-28s %8.2f MB" % (k, v["hbm_bytes_per_launch_corrected"] / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
