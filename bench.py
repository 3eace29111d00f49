#!/usr/bin/env python3
"""Headline benchmark: examples/sec, forward + backward, DeepFM 10M-vocab x 16d, batch 8192 per GPU
(BASELINE.json `metric`), plus achieved HBM GB/s of the embedding-gather kernel against the gfx950 roofline
and the CPU baseline (oracle/torch_ref.py, the op-for-op TF2-CPU stand-in) timed on the same box.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one train_loop iteration without the optimizer (2.FM/ModelManager.py:171-177): index assembly,
fused gather + FM, DNN part, sigmoid, Keras BCE, and every gradient materialised (dense parameters as dense
tensors, the two tables as sorted-unique row sums).  Inputs (26 int64 [B,1] feature tensors + the label, the
DataGenerator contract) are resident in HBM before the timed region starts.  One JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)

CFG = dict(workload="DeepFM fwd+bwd, 26 categorical fields, 10M-vocab x 16d, batch 8192 (BASELINE.json metric)",
           vocab=10_000_000, fields=26, embedding_dims=16, mlp_dims=[32, 8], batch=8192)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--vocab", type=int, default=CFG["vocab"])
    ap.add_argument("--batch", type=int, default=CFG["batch"])
    ap.add_argument("--dist", default="uniform", choices=["uniform", "zipf"])
    ap.add_argument("--no-graph", action="store_true", help="enqueue every step eagerly (no hipGraph replay)")
    ap.add_argument("--resident", type=int, default=32,
                    help="distinct batches resident in HBM and cycled through (32 x 27 MB of table rows > the 256 MB "
                         "memory-side cache: no step finds its rows cached from the previous cycle)")
    ap.add_argument("--cycle", type=int, default=0,
                    help="steps per captured hipGraph (<= 32); default: the largest divisor of --steps that is <= 32")
    ap.add_argument("--step-graphs", action="store_true", help="one hipGraph per step instead of one per 4-step cycle")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: REHEARSAL of the multi-rank control flow on a box with fewer GPUs than ranks -- every rank "
                         "uses GPU (local_rank mod device count), exchanges hop through host memory (not a measurement)")
    ap.add_argument("--sharded-eager", action="store_true",
                    help="row-sharded step: issue every step eagerly.  The default captures each cycle of steps, RCCL "
                         "collectives included, in one hipGraph (fixed-capacity exchanges: a fixed program); a capture "
                         "that fails ends the run with a non-zero exit code, there is no silent fallback")
    ap.add_argument("--sharded-graph", action="store_true", help="(accepted for compatibility: graphs are the default)")
    ap.add_argument("--repeats", type=int, default=15,
                    help="further timed regions of K steps after the reported one, for median / p10 / p90 per step")
    ap.add_argument("--no-dssm", action="store_true", help="skip the sharded DSSM (config D) scaling record")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary records (Zipf value, B sweep, CrossNet GEMMs, sharded DSSM, optimizers)")
    ap.add_argument("--generic", action="store_true", help="use the generic ~35-kernel step instead of the fused one")
    ap.add_argument("--replicas", action="store_true",
                    help="N > 1: independent full-table replicas instead of the row-sharded table + RCCL all-to-all")
    ap.add_argument("--sharded", action="store_true", help="force the row-sharded step (the default for N > 1) at N = 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline sample")
    ap.add_argument("--adam-steps", type=int, default=10, help="extra: full train steps with Keras Adam as the reference applies it (dense sweep)")
    return ap.parse_args()


def host_cores():
    """Cores this process may actually use: the cgroup CPU quota when there is one (the GPU box gives a
    16-CPU share of a 256-thread host), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(args, names, seconds):
    """oracle/torch_ref.py DeepFM forward + BCE + backward (sparse table grads, de-duplicated) on the host
    cores: the "TF2-CPU stand-in" of BASELINE.md section 2.  Bounded sample of the same workload."""
    from oracle import torch_ref as T
    from explicit_tf2_recommendation_amd import data
    cores = host_cores()
    torch.set_num_threads(cores)
    V, F, E, B = args.vocab, CFG["fields"], CFG["embedding_dims"], args.batch
    g = torch.Generator().manual_seed(1234)
    u = lambda *s: (torch.rand(*s, generator=g) - 0.5) * 0.1
    dims = [F * E] + CFG["mlp_dims"]
    p = {"embed": u(V, E).requires_grad_(), "w": u(V, 1).requires_grad_(), "bias": u(1).requires_grad_(),
         "k1": [u(dims[i], dims[i + 1]).requires_grad_() for i in range(2)],
         "b1": [torch.zeros(dims[i + 1]).requires_grad_() for i in range(2)],
         "k2": [u(dims[-1], 1).requires_grad_()], "b2": [torch.zeros(1).requires_grad_()]}
    leaves = [p["embed"], p["w"], p["bias"]] + p["k1"] + p["b1"] + p["k2"] + p["b2"]
    gen = data.SyntheticGenerator(names, V, dist=args.dist, seed=0)
    batches = []
    for _ in range(2):
        b = gen.batch(B)
        batches.append(({k: torch.from_numpy(b[k]) for k in names}, torch.from_numpy(b["label"])))

    def one(i):
        ins, y = batches[i % len(batches)]
        X = T.index_assemble(ins, names)
        loss = T.keras_bce(y, T.deepfm_forward(p, X, sparse=True))
        grads = torch.autograd.grad(loss, leaves)
        return [gr.coalesce() if gr.is_sparse else gr for gr in grads]     # de-duplicated IndexedSlices

    one(0)
    n, t0 = 0, time.perf_counter()
    while True:
        one(n)
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= 200:
            break
    return {"value": n * B / el, "unit": "examples/s", "cores": cores, "kind": "port",
            "sample": "%d fwd+bwd steps of the same DeepFM config (B=%d, V=%d) in %.1f s, torch-CPU eager "
                      "op-for-op restatement (oracle/torch_ref.py), %d threads" % (n, B, V, el, cores)}


def gather_sweep(lib, check, layer, V, F, E, names, timed):
    """B sweep of the gather + FM forward kernel (rec_emb_fm_fwd_f32), fresh ids every launch: SURVEY.md section 7's
    "sweep B to 64 k to show the asymptote".  frac = algorithmic bytes / time / 8 TB/s; frac_lines counts the 128-byte
    lines actually fetched (one per lookup)."""
    import ctypes as C
    from explicit_tf2_recommendation_amd import ops
    vp = lambda t: C.c_void_p(t.data_ptr())
    emb, w, bias = layer.embed.embeddings, layer.w.embeddings, layer.bias
    dims = [V // F] * F
    dims[-1] += V - sum(dims)
    offs = np.concatenate([[0], np.cumsum(dims[:-1])])
    rng = np.random.Generator(np.random.PCG64(5))
    out = []
    for Bs in (8192, 16384, 32768, 65536, 131072):
        nset = max(4, (16 * 8192) // Bs)                  # >= 436 MB of distinct lines between two uses of an id set
        Xs = [ops.index_pack([torch.from_numpy(rng.integers(0, dims[f], size=Bs) + offs[f]).cuda() for f in range(F)])
              for _ in range(nset)]
        z = torch.empty(Bs, dtype=torch.float32, device="cuda")

        def launch(n, Xs=Xs, z=z, Bs=Bs, nset=nset):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            for i in range(n):
                check(lib.rec_emb_fm_fwd_f32(vp(emb), emb.stride(0), vp(w), w.stride(0), vp(bias), V, E, vp(Xs[i % nset]),
                                             Bs, F, vp(z), None, None, None, None, st), "rec_emb_fm_fwd_f32")

        us = timed(launch, 2 * nset)
        n = Bs * F
        nbytes = n * (8 + 4 * E + 4) + 4 * Bs
        out.append({"B": Bs, "n_lookups": n, "avg_launch_us": us, "G_lookups_per_s": n / us / 1e3,
                    "frac": nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                    "frac_lines": n * 128 / (us * 1e-6) / 1e9 / HBM_PEAK_GBS})
        del Xs
    torch.cuda.empty_cache()
    return out


def crossnet_record(ops, timed):
    """fp32 MFMA utilisation of the three GEMMs of a matrix-mode CrossNet layer (3.DCN/CustomLayers.py:297-305 and its
    backward) at BASELINE config C: M = 16384 rows, D = 323 (10 categorical fields, the reference default) and D = 835
    (26 fields).  Live-timed like `roofline`; peak = 157.3 TFLOP/s (fp32 matrix, MI355X_MICROARCH.md)."""
    PEAK = 157.3
    out = []
    M = 16384
    for D in (323, 835):
        X = torch.randn((M, D), device="cuda")
        X0 = torch.randn((M, D), device="cuda")
        W = torch.randn((D, D), device="cuda") * 0.05
        bvec = torch.zeros(D, device="cuda")
        H = torch.randn((M, D), device="cuda")
        U = torch.empty((M, D), device="cuda")
        forms = (("fwd X.W^T + cross epilogue x0*(u+b)+x (keeps U)",
                  lambda: ops.gemm(X, W, False, True, epi=ops.EPI_CROSS, bias=bvec, e0=X0, e1=X, aux=U)),
                 ("bwd dX = H.W", lambda: ops.gemm(H, W, False, False)),
                 ("bwd dW = H^T.X (split-K over the batch)", lambda: ops.gemm(H, X, True, False)))
        for name, fn in forms:
            def launch(n, fn=fn):
                for _ in range(n):
                    fn()
            us = timed(launch, 20)
            tf = 2.0 * M * D * D / us / 1e6
            out.append({"bound": "mfma", "D": D, "M": M, "gemm": name, "avg_launch_us": us, "achieved": tf, "peak": PEAK,
                        "unit": "TFLOP/s", "frac": tf / PEAK})
        del X, X0, W, H, U
    torch.cuda.empty_cache()
    return out


def dssm_record(world, rank, local_rank, barrier, reduce_max):
    """BASELINE config D as the north star states it: DSSM two-tower (2.FM/CustomLayers.py:208-239), item table 100M x 64d and
    user table 10M x 64d ROW-SHARDED over the ranks (layers.DSSMTwoTowerRetrievalLayer(sharded=True): de-duplicated
    fixed-capacity all-to-all of ids / rows / row gradients), dense parameters data-parallel (one flat all-reduce), B = 8192
    examples per rank (weak scaling), forward + Keras BCE + backward.  N = 1: the whole step replays from one hipGraph (no
    collective: the rank's own slab stays out of RCCL); N > 1: enqueued eagerly (RCCL inside a captured graph is verified at
    world size 1 only)."""
    import torch.distributed as dist
    from explicit_tf2_recommendation_amd import layers, data, engine, sharded, functional as Fn
    Vi, Vu, E, Bd = 100_000_000, 10_000_000, 64, 8192
    un, inn = ["user_tag1", "user_tag2"], ["item_tag1", "item_tag2", "item_tag3"]
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
    old_check = layers.Layer.check_ids
    layers.Layer.check_ids = False
    try:
        layers.set_init_seed(7)
        layer = layers.DSSMTwoTowerRetrievalLayer(u_feature_names=un, i_feature_names=inn, u_feature_dims=1000,
                                                  i_feature_dims=1000, u_embedding_dims=E, i_embedding_dims=E).cuda()
        comm = sharded.DistComm(None, separate_count_channel=True)
        # per-owner capacity of the exchange from the field layout (engine.exchange_capacity: sum over the fields that meet
        # the owner's block of min(B, overlap)), as the sharded DeepFM step does
        def cap(V, nf):
            dims, offs, _ = data.field_layout(V, nf)
            return engine.exchange_capacity(dims, offs, Bd, -(-V // world), world)
        layer.i_tower.embed = sharded.ShardedEmbedding(Vi, E, comm=comm, capacity=cap(Vi, 3), device="cuda")
        layer.u_tower.embed = sharded.ShardedEmbedding(Vu, E, comm=comm, capacity=cap(Vu, 2), device="cuda")
        dense = [p for n, p in layer.named_parameters() if "embeddings_shard" not in n]
        nb = 4
        bs = []
        for k in range(nb):
            gi = data.SyntheticGenerator(inn, Vi, seed=10 * rank + k).batch(Bd)
            gu = data.SyntheticGenerator(un, Vu, seed=1000 + 10 * rank + k).batch(Bd)
            bs.append(data.to_device({**{n: gu[n] for n in un}, **{n: gi[n] for n in inn}, "label": gi["label"]}))
        graphed = world == 1
        if graphed:
            gstep = engine.GraphedTrainStep(layer, bs[0])
            one = lambda i: gstep(bs[i % nb])
        else:
            def one(i):
                b = bs[i % nb]
                for p in layer.parameters():
                    p.grad = None
                out = layer({k: b[k] for k in un + inn})["output"]
                Fn.KerasBCE.apply(out, b["label"]).backward()
                sharded.allreduce_dense_grads(dense)
        for i in range(6):
            one(i)
        K = 40
        barrier()
        t0 = time.perf_counter()
        for i in range(K):
            one(i)
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            el = reduce_max(el)
        for t in (layer.i_tower.embed, layer.u_tower.embed):
            t.check_flags()
        rec = {"config": "D: DSSM two-tower, item table 100M x 64d + user table 10M x 64d row-sharded over %d rank(s), "
                         "fwd + Keras BCE + bwd, B = 8192 per rank" % world,
               "n_gpus": world, "batch_per_gpu": Bd, "steps": K, "ms_per_step": el / K * 1e3,
               "value": world * Bd * K / el, "unit": "examples/s", "scaling": "weak",
               "mode": "one hipGraph per step" if graphed else "eager (RCCL collectives outside graphs)",
               "exchange_slots_per_owner": {"item": cap(Vi, 3), "user": cap(Vu, 2)}}
        del layer, bs
        torch.cuda.empty_cache()
        return rec
    finally:
        layers.Layer.check_ids = old_check


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves -- one fresh child
    process per GPU under torch.distributed.run (rendezvous on 127.0.0.1) -- and relay rank 0's JSON line and the exit
    code.  This parent never touches the GPU (nothing here initialises HIP), so nothing is re-executed over a live
    GPU context; the children are ordinary subprocesses."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    line = None
    for ln in proc.stdout.decode(errors="replace").splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            sys.stderr.write(ln + "\n")
    if proc.returncode != 0 or line is None:
        raise SystemExit(proc.returncode or 1)
    sys.stdout.write(line + "\n")
    sys.stdout.flush()


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %s ranks: the launcher's count is used\n"
                         % (args.gpus, os.environ["WORLD_SIZE"]))
    # RCCL prints its version banner on stdout: keep stdout for the ONE JSON line, send everything else to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    rehearsal = args.dist_backend == "gloo"
    if rehearsal:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    if world > 1:
        import datetime
        # a collective that never completes surfaces as an error after 5 minutes instead of the 10-minute default
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=5))
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=5),
                                    device_id=torch.device("cuda", local_rank))

    def reduce_max(x):
        """max over ranks of a host scalar (the slowest rank's time is the job's time)"""
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    from explicit_tf2_recommendation_amd import layers, engine, data, ops

    V, F, E, B = args.vocab, CFG["fields"], CFG["embedding_dims"], args.batch
    names = ["C%d" % (i + 1) for i in range(F)]
    layers.set_init_seed(1234)
    layer = layers.DeepFMRankingLayer(feature_names=names, feature_dims=V, embedding_dims=E,
                                      mlp_dims=CFG["mlp_dims"]).cuda()
    gen = data.SyntheticGenerator(names, V, dist=args.dist, seed=rank)
    n_batches = args.resident
    # steps per captured hipGraph: the largest divisor of K that a call can hold (32).  A graph launch leaves the GPU idle for
    # ~30 us (kernel trace: profiles/r03_*), so K = 20 runs as ONE 20-step graph and K = 200 as eight 25-step graphs
    # (no batch twice in one call: the second occurrence would not find a prefetched plan and be sorted in line)
    Cy = args.cycle or max(d for d in range(1, min(32, n_batches) + 1) if args.steps % d == 0)
    if Cy > min(32, n_batches):
        raise SystemExit("--cycle must be <= min(32, --resident)")
    batches = [data.to_device(gen.batch(B)) for _ in range(n_batches)]
    sharded_mode = args.sharded or (world > 1 and not args.replicas)
    if sharded_mode and not args.cycle:          # the sharded step's graphs hold a slice of the resident batches
        Cy = 8 if args.steps % 8 == 0 else 4
    if sharded_mode:    # table rows block-partitioned over the ranks; ids / rows / row gradients by RCCL all-to-all
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        if rehearsal:
            from explicit_tf2_recommendation_amd import sharded as _sh
            args.sharded_eager = True                    # exchanges through host memory cannot be captured in a hipGraph
            step = engine.ShardedDeepFMStep(layer, B, gen.dims, gen.offsets, comm=_sh.HostStagedComm(None, True))
        else:
            step = engine.ShardedDeepFMStep(layer, B, gen.dims, gen.offsets)
    elif args.generic:
        step = engine.DeepFMTrainStep(layer, B, optimizer=None, use_graph=not args.no_graph)
    else:   # per step: fused fwd+bwd kernel, reduction + segment sums (one launch), per-column sort on a second stream
        step = engine.DeepFMFusedStep(layer, B, gen.dims, gen.offsets, optimizer=None, use_graph=not args.no_graph)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(i, n):
        if args.generic:
            step(batches[i % n_batches])
        else:   # the next batch is announced: its de-duplication plan is built while this one is differentiated (the
                # last step of a run announces the first batch of the next run: every run starts at batch 0)
            step(batches[i % n_batches], next_inputs=batches[(i + 1) % n_batches if i + 1 < n else 0])

    # launch-bound inner loop: the resident batches are cycled through --cycle steps per captured hipGraph (one graph
    # launch costs ~20 us of idle GPU); --step-graphs keeps one graph per step
    cycle = (not sharded_mode) and (not args.generic) and (not args.no_graph) and (not args.step_graphs)

    def calls_of(n, batches=batches):
        """The calls of one run of n steps, each with the batches it announces as next: every run starts at batch 0, so
        its last call announces the first batches of the next run (an input pipeline that knows what comes next) and
        all runs of the same n are identical -- the rehearsals below capture exactly the graphs the timed run replays."""
        seq = lambda i: [batches[(i + j) % len(batches)] for j in range(min(Cy, n - i))]
        out, i = [], 0
        while i < n:
            cur = seq(i)
            nxt = seq(i + len(cur)) if i + len(cur) < n else seq(0)
            out.append((cur, nxt))
            i += len(cur)
        return out

    def run_steps(n):
        i = 0
        if cycle:       # every call announces the batches of the NEXT call: their plans are built beside this call's steps
            for cur, nxt in calls_of(n):
                step.many(cur, then=nxt)
            return
        if sharded_mode and not args.sharded_eager:
            while n - i >= Cy:
                b0 = i % n_batches
                step.many(batches[b0:b0 + Cy])
                i += Cy
        while i < n:
            run(i, n)
            i += 1

    nw = max(args.warmup, 2 * n_batches)                 # warm-up also captures the hipGraphs of the resident batches
    nw += (-nw) % n_batches                              # ... and ends where the timed loop starts (batch 0)
    # a failure of the row-sharded exchange is a failure of the run: no silent switch to replicas (ask for
    # --replicas explicitly to measure those)
    run_steps(nw)
    # A full CPython garbage collection walks every tracked object of the imported modules (~45 ms with torch loaded):
    # collect now and keep the collector off inside the timed region (the loop allocates a few tuples per call).
    import gc
    gc.collect()
    gc.disable()
    # The last untimed steps are the timed sequence itself, twice (the plan-buffer ring has two halves), enqueued right
    # before the opening barrier: (1) no graph is captured inside the timed region whatever K is; (2) the GPU goes into it
    # busy -- after an idle stretch (the 45-ms collection above is one) the first ~150 us of work run at idle clocks,
    # which at K = 20 is a tenth of the region.
    # a graph is captured at the SECOND sighting of a call's addresses (that call itself still runs eagerly) and every call
    # has two forms (the plan-buffer ring has two halves): four rehearsals leave nothing to capture, and two more REPLAY
    # each captured graph once -- the first launch of a graph uploads it (40-500 us once per graph, which at K = 20 was
    # 2-25 us per step of the first timed region)
    for _ in range(6):
        run_steps(args.steps)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    # further regions of the same K steps (same graphs, same batches): the distribution behind the one reported number
    rep_ms = []
    for _ in range(max(0, args.repeats)):
        barrier()
        t1 = time.perf_counter()
        run_steps(args.steps)
        barrier()
        e1 = time.perf_counter() - t1
        if world > 1:
            e1 = reduce_max(e1)
        rep_ms.append(e1 / args.steps * 1e3)
    gc.enable()
    if world > 1:
        elapsed = reduce_max(elapsed)
    try:
        step.check_flags() if hasattr(step, "check_flags") else None
        if int(step.oob.item()) != 0:
            raise IndexError("embedding id out of range")
    except (IndexError, ValueError) as e:
        raise SystemExit("out-of-range id seen by the gather / sort kernels: %s" % e)
    loss = float(step.loss.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed

    replicas = None
    if sharded_mode and not args.generic:
        # reference point for the cost of the exchange: the same batches through the single-GPU fused step on a full
        # table replica per rank (no collective at all), timed the same way.  NOT the reported value.
        rstep = engine.DeepFMFusedStep(layer, B, gen.dims, gen.offsets, optimizer=None, use_graph=not args.no_graph)

        def rrun(n):
            if not args.no_graph:
                for cur, nxt in calls_of(n):
                    rstep.many(cur, then=nxt)
                return
            for i in range(n):
                rstep(batches[i % n_batches], next_inputs=batches[(i + 1) % n_batches])

        rrun(nw)
        gc.collect()
        gc.disable()
        for _ in range(4):
            rrun(args.steps)
        barrier()
        t0 = time.perf_counter()
        rrun(args.steps)
        barrier()
        gc.enable()
        rel = time.perf_counter() - t0
        if world > 1:
            rel = reduce_max(rel)
        replicas = {"value": world * B * args.steps / rel, "unit": "examples/s", "ms_per_step": rel / args.steps * 1e3,
                    "note": "independent full-table replicas, no exchange (upper bound; not the reported value)"}

    # ---- rooflines.  Average launch durations are measured live with HIP events around `reps` back-to-back launches
    # replayed from a hipGraph (events and replay share torch's current stream); `traffic` = HBM bytes per launch
    # from the PMC passes committed in profiles/r01_pmc_traffic.json (separate --pmc FETCH_SIZE / WRITE_SIZE runs of
    # this same command, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).
    import ctypes as C
    from explicit_tf2_recommendation_amd._lib import lib, check
    vp = lambda t: C.c_void_p(t.data_ptr())
    pmc, pmc_file = {}, {}
    try:
        pmc_file = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")))
        pmc = pmc_file["kernels"]
    except (OSError, ValueError, KeyError):
        pass

    def timed(launch, reps):
        launch(10)
        torch.cuda.synchronize()
        gg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gg, capture_error_mode=engine.CAPTURE_MODE):
            launch(reps)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gg.replay()
        torch.cuda.synchronize()
        ev0.record()
        gg.replay()
        ev1.record()
        torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) * 1e3 / reps

    def roof(kernel, algo_bytes, us, traffic_keys, note):
        achieved = algo_bytes / (us * 1e-6) / 1e9
        tr = [pmc[k]["hbm_bytes_per_launch_corrected"] for k in traffic_keys if k in pmc]
        return {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": sum(tr) if len(tr) == len(traffic_keys) else None,
                "traffic_source": {"file": "profiles/r03_pmc_traffic.json", "commit": pmc_file.get("commit"),
                                   "note": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, "
                                           "FETCH_SIZE doubled (gfx950); not re-measured in this run"},
                "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_us": us, "note": note}

    # (1) standalone gather + FM forward (what FMRankingLayer's forward launches): ids + embed rows + w + logit
    gather_bytes = B * F * (8 + E * 4 + 4) + B * 4
    # every launch of the timed replay takes another resident batch: no launch finds its rows in the memory-side cache
    Xs = [ops.index_pack([b[k] for k in names]) for b in batches]
    emb, w, bias = layer.embed.embeddings, layer.w.embeddings, layer.bias
    zbuf = torch.empty(B, dtype=torch.float32, device="cuda")

    def launch_gather(n):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for i in range(n):
            check(lib.rec_emb_fm_fwd_f32(vp(emb), emb.stride(0), vp(w), w.stride(0), vp(bias), V, E,
                                         vp(Xs[i % n_batches]), B, F, vp(zbuf), None, None, None, None, st),
                  "rec_emb_fm_fwd_f32")

    roofline_gather = roof("emb_fm_fwd_vec_kernel<8,2,fused> (embedding gather + FM, fused 128-B rows)", gather_bytes,
                           timed(launch_gather, 100), ["emb_fm_fwd_vec_kernel"],
                           "a random row read costs one 128-B line whatever the row size (76 algorithmic bytes per lookup "
                           "ride in 128 fetched).  At B = 8192 a launch is 213 k lookups: the bandwidth-delay regime -- the "
                           "bare line read of the same ids (scripts/exp/gather_sweep.hip) takes 6.8 us, ~3.4 us of it launch "
                           "ramp + two dependent HBM latencies; `sweep` shows the same kernel at larger batches, where the "
                           "chip sustains ~48-50 G random lines/s (6.1-6.4 TB/s of lines).  Every timed launch takes another "
                           "resident batch (no reuse through L2 / the 256-MB memory-side cache)")
    roofline_gather["physical_frac"] = (roofline_gather["traffic"] / (roofline_gather["avg_launch_us"] * 1e-6) / 1e9 /
                                        HBM_PEAK_GBS) if roofline_gather["traffic"] else None

    # (1b) the plain row gather at the row widths of the other configs, where the 128-byte-line granularity no longer
    # halves the useful bytes: E = 32 (128-B rows, V = 10M: config C) and E = 64 (256-B rows, V = 100M: config D).
    # Algorithmic bytes per lookup (materialising gather): 8 (id) + 2 * 4E (row read + row written).
    def gather_record(Eg, Vg, n_cfg, cfg_name):
        tab = torch.empty((Vg, Eg), dtype=torch.float32, device="cuda")
        tab.uniform_(-0.05, 0.05)                     # every page written: an untouched allocation reads unrealistically fast
        n_big = 1 << 20
        rs = np.random.Generator(np.random.PCG64(Eg))
        idsets = [torch.from_numpy(rs.integers(0, Vg, size=n_big)).cuda() for _ in range(8)]   # fresh rows per launch
        out = torch.empty((n_big, Eg), dtype=torch.float32, device="cuda")
        rec = {}
        for key, n_l in (("asymptote", n_big), ("config", n_cfg)):
            def launch(n, n_l=n_l):
                st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                for i in range(n):
                    check(lib.rec_emb_gather_f32(vp(tab), Vg, Eg, Eg, vp(idsets[i % 8]), n_l, vp(out), None, st),
                          "rec_emb_gather_f32")
            us = timed(launch, 40)
            nbytes = n_l * (8 + 2 * 4 * Eg)
            rec[key] = {"n_lookups": n_l, "algorithmic_bytes_per_launch": nbytes, "avg_launch_us": us,
                        "achieved": nbytes / (us * 1e-6) / 1e9, "frac": nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}
        tkey = "gather_rows_kernel_e%d" % Eg
        traffic = pmc[tkey]["hbm_bytes_per_launch_corrected"] if tkey in pmc else None
        a = rec["asymptote"]
        res = {"bound": "hbm", "kernel": "gather_rows_kernel<%d,4> (rec_emb_gather_f32, %d-byte rows, table %d x %dd)"
               % (Eg // 4, 4 * Eg, Vg, Eg), "achieved": a["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": a["frac"], "traffic": traffic,
               "physical_frac": (traffic / (a["avg_launch_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS) if traffic else None,
               "n_lookups": a["n_lookups"], "algorithmic_bytes_per_launch": a["algorithmic_bytes_per_launch"],
               "avg_launch_us": a["avg_launch_us"], "at_config_size": dict(rec["config"], config=cfg_name)}
        del tab, out, idsets
        torch.cuda.empty_cache()
        return res

    roofline_gather_e32 = gather_record(32, 10_000_000, 16384 * 10, "C: B=16384 x 10 categorical fields")
    roofline_gather_e64 = gather_record(64, 100_000_000, 8192 * 3, "D: B=8192 x 3 item fields")
    roofline = roofline_gather
    if not args.generic:
        # (2) the dominant kernel of the timed step: the fused forward+backward kernel.  Algorithmic bytes: the gather
        # above + labels + dL/dz + the IndexedSlices values it writes.
        fused_bytes = gather_bytes + B * 4 + B * 4 + B * F * E * 4
        fs = engine.DeepFMFusedStep(layer, B, gen.dims, gen.offsets, optimizer=None, use_graph=False)
        colss = [fs._cols(b) for b in batches]
        assert n_batches <= fs.NBUF
        for i in range(n_batches):                    # the plans exist before the kernel runs (direct mode), as in a step
            fs._sort(colss[i], i, torch.cuda.current_stream())
        torch.cuda.synchronize()

        def launch_fused(n):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            for i in range(n):
                fs._launch_main(colss[i % n_batches], batches[i % n_batches]["label"], st, i % n_batches)

        roofline = roof("deepfm3_kernel<direct, K0 in LDS> (gather, FM, MLP on fp32 MFMA, BCE, backward, IndexedSlices values "
                        "written to their de-duplicated slots; its workgroup partials are reduced in the next launch)",
                        fused_bytes, timed(launch_fused, 48), ["deepfm3_kernel"],
                        "one 8-wave workgroup (32 examples) per CU = two 16-example halves one phase apart: the backward of "
                        "half A runs on the matrix cores while the rows of half B are still landing.  What bounds a launch at "
                        "B = 8192 (phase stamps, profiles/r03_fused3_stamps.txt): launch ramp + ids (2 us) + the CU's 832 rows "
                        "at ~25 GB/s per CU (to 8.6 us) + 4.2 us of fp32 MFMA per CU (0.65 GFLOP at the f32 rate) + two head "
                        "latencies, then ~3 us for 27 MB of writes to leave the chip")

    extra = {}
    if rep_ms:
        r_ = np.array(rep_ms)
        extra["stats"] = {"regions": len(rep_ms), "steps_per_region": args.steps,
                          "ms_per_step_median": float(np.median(r_)), "ms_per_step_p10": float(np.percentile(r_, 10)),
                          "ms_per_step_p90": float(np.percentile(r_, 90)), "ms_per_step_min": float(r_.min()),
                          "value_median": world * B / (float(np.median(r_)) * 1e-3),
                          "note": "further timed regions of the same K steps in this run (same graphs, same batches, barrier "
                                  "+ synchronize on both sides each); `value` / `ms_per_step` are the FIRST region"}
    if not args.no_extras and not sharded_mode and not args.generic and world == 1:
        # the same measurement on Zipf(1.05) ids: hot ids form long runs in the de-duplication
        gz_ = data.SyntheticGenerator(names, V, dist="zipf", seed=rank)
        zb = [data.to_device(gz_.batch(B)) for _ in range(n_batches)]
        zstep = engine.DeepFMFusedStep(layer, B, gz_.dims, gz_.offsets, optimizer=None, use_graph=not args.no_graph)

        def zrun(n):
            for cur, nxt in calls_of(n, zb):
                zstep.many(cur, then=nxt)

        zrun(2 * n_batches)
        for _ in range(6):
            zrun(args.steps)
        zt = []
        for _ in range(5):
            barrier()
            t1 = time.perf_counter()
            zrun(args.steps)
            barrier()
            zt.append((time.perf_counter() - t1) / args.steps)
        zstep.check_flags()
        # ... and on four more draws of the uniform ids (seeds 0..4; `value` is seed 0): the same graphs'
        # worth of work on other addresses, median of five regions each
        seeds = {}
        for sd in (1, 2, 3, 4):
            gs_ = data.SyntheticGenerator(names, V, dist=args.dist, seed=1000 * sd + rank)
            sb = [data.to_device(gs_.batch(B)) for _ in range(n_batches)]
            for _ in range(6):
                for cur, nxt in calls_of(args.steps, sb):
                    zstep.many(cur, then=nxt)
            st_ = []
            for _ in range(5):
                barrier()
                t1 = time.perf_counter()
                for cur, nxt in calls_of(args.steps, sb):
                    zstep.many(cur, then=nxt)
                barrier()
                st_.append((time.perf_counter() - t1) / args.steps)
            seeds[str(sd)] = float(np.median(st_)) * 1e3
            del sb
        zstep.check_flags()
        extra["ms_per_step_seeds_1_to_4"] = seeds
        extra["value_zipf"] = world * B / float(np.median(zt))
        extra["ms_per_step_zipf"] = float(np.median(zt)) * 1e3
        del zb, zstep
        roofline_gather["sweep"] = gather_sweep(lib, check, layer, V, F, E, names, timed)
        extra["roofline_crossnet"] = crossnet_record(ops, timed)
    if not args.no_extras and not args.no_dssm:
        try:
            extra["scaling_dssm_d"] = dssm_record(world, rank, local_rank, barrier, reduce_max)
        except Exception as e:                               # a secondary record never takes the headline line down
            extra["scaling_dssm_d"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if args.adam_steps > 0 and rank == 0 and not args.no_extras:
        st2 = (engine.DeepFMTrainStep(layer, B, optimizer="keras_adam", lr=1e-3, use_graph=False) if args.generic else
               engine.DeepFMFusedStep(layer, B, gen.dims, gen.offsets, optimizer="keras_adam", lr=1e-3, use_graph=False))
        for i in range(2):
            st2(batches[i % n_batches])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.adam_steps):
            st2(batches[i % n_batches])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / args.adam_steps
        extra["train_step_keras_adam_ms"] = dt * 1e3
        extra["train_step_keras_adam_examples_per_s"] = B / dt
        extra["adam_sweep_bytes_per_step"] = 6 * (V * E + V) * 4
        if not args.generic:
            # touched-rows (lazy) Adam applied inside the post launch (non-reference semantics, opt-in);
            # the step counter lives on the device, so whole train steps replay from hipGraphs like the gradient-only ones
            st3 = engine.DeepFMFusedStep(layer, B, gen.dims, gen.offsets, optimizer="lazy_adam", lr=1e-3,
                                         use_graph=not args.no_graph)
            nl = Cy * max(1, (4 * args.adam_steps + Cy - 1) // Cy)

            def lazy_run(n):
                for cur, nxt in calls_of(n):
                    st3.many(cur, then=nxt)

            lazy_run(2 * n_batches)                               # captures the graphs of the resident batches
            for _ in range(4):                                    # ... and of the timed sequence itself (captured at the second
                lazy_run(nl)                                      # sighting, two forms per call: the plan ring has two halves)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            lazy_run(nl)
            torch.cuda.synchronize()
            extra["train_step_lazy_adam_ms"] = (time.perf_counter() - t1) / nl * 1e3
            # Keras Adam itself, evaluated lazily and exactly: rows skip the dense sweeps and replay them right before a
            # batch reads them (bit-identical tables, tests/test_gpu_engine.py).  What the replay costs depends on how long
            # rows stay untouched, so this runs on 64 resident batches (a row recurs after up to 64 steps; with fresh
            # uniform batches the mean gap at 213k of 10M rows per step is ~47) and is timed in the steady state, after
            # 256 steps; flush_ms = bringing every row up to date afterwards (the sweep that was saved, once).
            st3.release()                                         # one lazy-optimizer step per layer (table padding)
            st4 = engine.DeepFMFusedStep(layer, B, gen.dims, gen.offsets, optimizer="keras_adam_lazy", lr=1e-3,
                                         use_graph=not args.no_graph)
            fresh = [data.to_device(gen.batch(B)) for _ in range(64)]
            c4 = 4

            def exact_run():
                for i in range(0, 64, c4):
                    st4.many(fresh[i:i + c4], then=fresh[(i + c4) % 64:(i + c4) % 64 + c4])

            for _ in range(6):
                exact_run()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            exact_run()
            torch.cuda.synchronize()
            extra["train_step_keras_adam_lazy_exact_ms"] = (time.perf_counter() - t1) / 64 * 1e3
            t1 = time.perf_counter()
            st4.flush()
            torch.cuda.synchronize()
            extra["keras_adam_lazy_exact_flush_ms"] = (time.perf_counter() - t1) * 1e3
            del fresh

    if rank == 0:
        out = {"metric": "examples/sec fwd+bwd, DeepFM 10M-vocab x16d batch 8192", "value": value,
               "unit": "examples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": CFG["workload"], "vocab": V, "fields": F, "embedding_dims": E,
                          "mlp_dims": CFG["mlp_dims"], "batch_per_gpu": B, "id_distribution": args.dist,
                          "parallelism": ("1 process per GPU; fused table row-sharded (block partition), batch "
                                          "data-parallel, RCCL all-to-all of ids / rows / row gradients, flat "
                                          "all-reduce of dense gradients") if sharded_mode else
                          ("1 process per GPU, independent full-table replicas" if world > 1 else "single GPU"),
                          "global_batch": world * B,
                          "hipgraph": ("one graph per %d steps incl. the RCCL collectives" % Cy
                                       if not args.sharded_eager else False) if sharded_mode else
                          (not args.no_graph) and
                          ("one graph per %d steps, %d resident batches" % (Cy, n_batches) if cycle else "one per step"),
                          "step": "sharded, de-duplicate first, fixed-capacity exchanges (constant split sizes, nothing read "
                          "back by the host): per-column sort plan + id message + owner rank-merge for the next batch on a "
                          "second stream (own communicator), owner gather, all-to-all of rows, fused fwd+bwd on them, "
                          "per-id sums into the send slots, all-to-all of row gradients, owner sums"
                          if sharded_mode else
                          "generic" if args.generic else
                          "fused: fwd+bwd kernel writing value rows straight to their de-duplicated slots (plan complete "
                          "before the kernel), then reduction + remaining segment sums in one launch; de-duplication "
                          "plan of batch k+1 (per-column sort, second stream) overlaps step k"},
               "roofline": roofline, "roofline_gather": roofline_gather, "roofline_gather_e32": roofline_gather_e32,
               "roofline_gather_e64": roofline_gather_e64, "loss": loss}
        if rehearsal:
            out["config"]["rehearsal"] = ("--dist-backend gloo: %d ranks share the visible GPUs, exchanges hop through host "
                                          "memory -- a rehearsal of the multi-rank control flow, NOT a measurement" % world)
        if replicas is not None:
            out["replicas_no_exchange"] = replicas
        out.update(extra)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, names, args.cpu_seconds)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
