#!/usr/bin/env python
"""
bench.py -- seed-nodes/sec of the GraphSAGE training hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]            # N=1: plain python
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one `train_step` (reference models.py:97-104) over one batch of 512 seed nodes per
GPU: K1 sample (fanout 25, 10) -> K2 gather+mean -> K5 project (2 layers, hidden 128) -> loss ->
backward -> [RCCL grad all-reduce] -> clip -> Adam.  Workload = BASELINE config 2 ("Reddit mean
2-layer 25/10 h=128 bf16"), synthetic at Reddit's shape: N=232 965 nodes,
lognormal degrees clipped to [1, 21 657], D=602 bf16 features, 41 classes; graph + features are
resident in HBM before the timed region.  Timing: W untimed steps, then exactly K steps between
barrier + torch.cuda.synchronize() on both sides, max over ranks; rank 0 prints ONE JSON line.

Extra objects on the line (tier contract, section 4 of the task):
  roofline     the dominant kernel = the hop-2 k_gather_mean launch (250 of the 276 rows/seed):
               algorithmic bytes = B*f1*f2*D*2 per launch / its mean duration, HIP events on the
               launch stream, fresh frontier per launch; peak = 8 TB/s HBM3E.
  cpu_baseline the oracle (oracle/torch_ref.py + oracle/gsage_oracle.c: a port of the reference's
               CPU op sequence) timed on this host's cores on a bounded sample of the same workload.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_NODES, MAX_DEG, FEAT_DIM, N_CLASSES = 232965, 21657, 602, 41
FANOUT, HIDDEN, BATCH = (25, 10), (128, 128), 512
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s HBM3E (spec)


def synthetic_reddit(n_nodes=N_NODES, seed=0, feat_dim=FEAT_DIM, max_deg=MAX_DEG):
    """Reddit-shaped problem in the reference's sparse convention (ids 1-based, row 0 dummy).
    Returns dict(adj=scipy csr, feats=callable(device, dtype)->FeatureStore, feats_np=callable,
    train_ids, targets)."""
    from scipy import sparse
    gs = importlib.import_module("pytorch-graphsage_amd")
    rng = np.random.default_rng(seed)               # PCG64: ~10x faster than the legacy stream
    deg = np.clip(np.exp(rng.normal(5.2, 1.3, size=n_nodes + 1)).astype(np.int64), 1, max_deg)
    deg[0] = 0
    deg[1] = max_deg                                  # pins adj.shape[1] (the sel population)
    indptr = np.zeros(n_nodes + 2, dtype=np.int64)
    np.cumsum(deg, out=indptr[1:])
    nnz = int(indptr[-1])
    data = rng.integers(1, n_nodes + 1, size=nnz, dtype=np.int32)
    indices = gs.store.row_positions(indptr)
    adj = sparse.csr_matrix((data, indices, indptr), shape=(n_nodes + 1, max_deg))
    adj.has_sorted_indices = True
    n_train = int(0.6586 * n_nodes)                   # 153 431 / 232 965
    train_ids = rng.permutation(np.arange(1, n_nodes + 1))[:n_train]
    targets = rng.integers(0, N_CLASSES, size=(n_nodes + 1, 1))

    def feats_np():
        frng = np.random.default_rng(seed + 1)
        f = frng.standard_normal(size=(n_nodes + 1, feat_dim), dtype=np.float32)
        f[0] = 0
        return f

    def feats(device, dtype="bf16"):
        return gs.FeatureStore.from_array(feats_np(), torch.device(device), dtype=dtype)

    return {"adj": adj, "feats": feats, "feats_np": feats_np, "train_ids": train_ids,
            "targets": targets, "nnz": nnz}


def build_model(gs, adj, aggregator="mean", rng="philox", seed=123, fanout=FANOUT, hidden=HIDDEN):
    from torch.nn import functional as F
    gs.nn_modules.SparseUniformNeighborSampler.rng_default = rng
    specs = [{"n_train_samples": f, "n_val_samples": f, "output_dim": h,
              "activation": (lambda x: x) if i == len(fanout) - 1 else F.relu}
             for i, (f, h) in enumerate(zip(fanout, hidden))]
    model = gs.GSSupervised(
        sampler_class=gs.sampler_lookup["sparse_uniform_neighbor_sampler"], adj=adj, train_adj=adj,
        prep_class=gs.prep_lookup["identity"], aggregator_class=gs.aggregator_lookup[aggregator],
        input_dim=FEAT_DIM, n_nodes=adj.shape[0], n_classes=N_CLASSES,
        layer_specs=specs, lr_init=0.01, lr_schedule="constant", weight_decay=0.0)
    model.train_sampler.seed = seed
    model.val_sampler.seed = seed
    return model


def rows_per_seed(fanout):
    n, prod = 1, 1
    for f in fanout:
        prod *= f
        n += prod
    return n


def cpu_baseline(data, budget_s=15.0, batch=BATCH):
    """Oracle train_step (port of the reference CPU path) on the host cores, bounded sample."""
    from oracle import cpu as ocpu
    from oracle import torch_ref as tref
    ncpu = os.cpu_count() or 1
    adj = data["adj"]
    indptr, dat = adj.indptr.astype(np.int64), adj.data.astype(np.int64)
    feats = torch.from_numpy(data["feats_np"]())
    gen = torch.Generator().manual_seed(0)
    D, h = FEAT_DIM, HIDDEN[0]
    w = {"agg_layers.0.fc_x.weight": torch.randn(h, D, generator=gen) / 25,
         "agg_layers.0.fc_neib.weight": torch.randn(h, D, generator=gen) / 25,
         "agg_layers.1.fc_x.weight": torch.randn(h, 2 * h, generator=gen) / 16,
         "agg_layers.1.fc_neib.weight": torch.randn(h, 2 * h, generator=gen) / 16,
         "fc.weight": torch.randn(N_CLASSES, 2 * h, generator=gen) / 16,
         "fc.bias": torch.zeros(N_CLASSES)}
    opt = tref.Adam()
    rng = np.random.RandomState(0)
    stream = ocpu.LegacyMT19937(123 ** 2)

    def one_step():
        ids = data["train_ids"][rng.randint(0, len(data["train_ids"]), size=batch)]
        tg = torch.from_numpy(data["targets"][ids])
        sels = [stream.choice(adj.shape[1], (batch, FANOUT[0])),
                stream.choice(adj.shape[1], (batch * FANOUT[0], FANOUT[1]))]
        tref.train_step(w, opt, 0.01, "classification", ids, feats, tg, indptr, dat, FANOUT, sels,
                        "mean", "identity", adj.shape[0])

    # torch's default (one thread per hardware thread) oversubscribes this gather-heavy step on
    # big hosts; give the CPU its best thread count from a short probe, then time with it.
    one_step()                                       # page faults, lazy init
    best, cores = None, 1
    for th in sorted({t for t in (8, 16, 32, 64, 128, ncpu) if t <= ncpu}):
        torch.set_num_threads(th)
        one_step()
        t = time.time()
        one_step()
        dt1 = time.time() - t
        if best is None or dt1 < best:
            best, cores = dt1, th
    torch.set_num_threads(cores)
    done, t0 = 0, time.time()
    while True:
        one_step()
        done += 1
        if done >= 3 and time.time() - t0 > budget_s:
            break
    dt = time.time() - t0
    return {"value": done * batch / dt, "unit": "seed-nodes/sec", "cores": cores, "kind": "port",
            "sample": "%d train_steps of %d seeds (oracle/torch_ref.py fp32 + C sampler, "
                      "torch %d of %d host threads, %.1f s)" % (done, batch, cores, ncpu, dt)}


def dominant_kernel_roofline(gs, model, store, data, dev, reps=40, n_frontiers=8):
    """Mean duration of the hop-2 k_gather_mean launch on fresh frontiers, HIP events on the launch
    stream (torch's current stream is the stream ops.py launches on)."""
    ops = gs.ops
    rng = np.random.RandomState(7)
    fronts = []
    for _ in range(n_frontiers):
        ids0 = torch.from_numpy(data["train_ids"][rng.randint(0, len(data["train_ids"]), size=BATCH)]).to(dev)
        ids1 = model.train_sampler(ids0, n_samples=FANOUT[0])
        fronts.append(model.train_sampler(ids1, n_samples=FANOUT[1]))
    M = BATCH * FANOUT[0]
    cdt = ops.torch_dtype()
    for f in fronts:                                   # warm
        ops.gather_mean(store, f, M, FANOUT[1], out_dtype=cdt, out_ld=store.ld)
    torch.cuda.synchronize()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for r in range(reps):
        ops.gather_mean(store, fronts[r % n_frontiers], M, FANOUT[1], out_dtype=cdt, out_ld=store.ld)
    stop.record()
    torch.cuda.synchronize()
    dur_s = start.elapsed_time(stop) / 1e3 / reps
    elem = store.data.element_size()
    alg_bytes = BATCH * FANOUT[0] * FANOUT[1] * FEAT_DIM * elem
    achieved = alg_bytes / dur_s / 1e9
    # HBM bytes per launch of the same kernel from the PMC pass committed under profiles/
    # (rocprofv3 --pmc FETCH_SIZE, x2 gfx950 correction); only valid for the default workload
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_dominant_kernel.json")
    if os.path.exists(pmc) and elem == 2:
        with open(pmc) as f:
            traffic = json.load(f).get("hbm_read_bytes_per_launch")
    return {"bound": "hbm", "kernel": "k_gather_mean (hop 2: 250 of 276 rows/seed)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "alg_bytes_per_launch": alg_bytes, "avg_launch_us": dur_s * 1e6}


MFMA_PEAK_TFLOPS = 2500.0      # MI355X dense bf16 (MI355X_MICROARCH.md; no sparsity)


def pool_kernel_roofline(gs, model, store, data, dev, reps=20, n_frontiers=4):
    """BASELINE configs[2] (max-pool): the dominant kernel is K3 (pooling MLP + bias + ReLU + segment
    max fused) on the hop-2 frontier -- MFMA-bound.  FLOPs = 2 * rows * D * hidden of the contraction
    the reference runs as mlp(neibs) (nn_modules.py:224).  Timed exactly as engine.FusedPoolTrainStep
    launches it: gsage_pool_mlp_packed over the frontier's rows gathered beforehand (not timed)."""
    ops, nat = gs.ops, gs._native
    layer = list(model.agg_layers.children())[0]
    lin = layer.mlp[0]
    rng = np.random.RandomState(7)
    M, n = BATCH * FANOUT[0], FANOUT[1]
    H = lin.weight.shape[0]
    fronts = []
    for _ in range(n_frontiers):
        ids0 = torch.from_numpy(data["train_ids"][rng.randint(0, len(data["train_ids"]), size=BATCH)]).to(dev)
        ids2 = model.train_sampler(model.train_sampler(ids0, n_samples=FANOUT[0]), n_samples=n)
        fronts.append(ops.gather_mean(store, ids2, M * n, 1, out_dtype=torch.bfloat16, out_ld=store.ld))
    wp = ops.pack_weight(lin.weight.detach().float().contiguous())
    bias = lin.bias.detach().float().contiguous()
    pooled = torch.empty(M, H, dtype=torch.float32, device=dev)
    pooled_b = torch.empty(M, H, dtype=torch.bfloat16, device=dev)
    argmax = torch.empty(M, H, dtype=torch.int32, device=dev)

    def run(rows):
        nat.check(nat.lib().gsage_pool_mlp_packed(rows.data_ptr(), store.ld, None, wp.data_ptr(), bias.data_ptr(), M, n,
                                                  H, FEAT_DIM, nat.POOL_MAX, pooled.data_ptr(), H, argmax.data_ptr(),
                                                  pooled_b.data_ptr(), H, None, ops._stream()), "pool_mlp_packed")
    for f in fronts:
        run(f)
    torch.cuda.synchronize()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for r in range(reps):
        run(fronts[r % n_frontiers])
    stop.record()
    torch.cuda.synchronize()
    dur_s = start.elapsed_time(stop) / 1e3 / reps
    flops = 2.0 * M * n * FEAT_DIM * H
    achieved = flops / dur_s / 1e12
    return {"bound": "mfma", "kernel": "k_pool_mlp_packed (K3, hop 2: 128 000 rows x 602 -> 512)",
            "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_PEAK_TFLOPS,
            "traffic": None, "alg_flops_per_launch": flops, "avg_launch_us": dur_s * 1e6}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch-size", type=int, default=BATCH, help="seed nodes per GPU per step")
    ap.add_argument("--aggregator", type=str, default="mean")
    ap.add_argument("--fanout", type=str, default="25,10", help="per-layer fan-outs (BASELINE: 25,10)")
    ap.add_argument("--hidden", type=str, default="128,128", help="per-layer output dims")
    ap.add_argument("--precision", type=str, default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--launch", choices=["cmdlist", "graph", "eager"], default="cmdlist",
                    help="how a step's kernels are issued: native command lists (default), hipGraph "
                         "replay, or one Python call per kernel")
    ap.add_argument("--no-graph", action="store_true", help="same as --launch eager")
    ap.add_argument("--engine", type=str, default="fused", choices=["fused", "autograd"],
                    help="fused: engine.FusedMeanTrainStep (no autograd below the head); "
                         "autograd: GSSupervised.train_step (captured unless --no-graph)")
    ap.add_argument("--pipeline", action="store_true",
                    help="fused engine: overlap batch k+1's sampling/gathers with batch k's compute "
                         "on a second stream (bit-identical results; currently no faster)")
    ap.add_argument("--per-step-copy", action="store_true",
                    help="copy each batch into the engine's static buffers per step instead of walking "
                         "a device-resident batch queue")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    args = ap.parse_args()

    import __graft_entry__
    __graft_entry__.ensure_built()
    gs = importlib.import_module("pytorch-graphsage_amd")
    ops = gs.ops
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world)
    ddp = gs.dist.init_from_env(cuda=True)
    rank = ddp.rank if ddp is not None else 0
    dev = torch.device("cuda", torch.cuda.current_device())
    ops.set_compute_dtype(args.precision)
    ops.warmup(dev)

    data = synthetic_reddit(seed=0)
    store = data["feats"](dev, args.precision)
    fanout = tuple(int(v) for v in args.fanout.split(","))
    hidden = tuple(int(v) for v in args.hidden.split(","))
    assert len(fanout) == len(hidden)
    model = build_model(gs, data["adj"], aggregator=args.aggregator, rng="philox", fanout=fanout,
                        hidden=hidden).to(dev)
    if ddp is not None:
        gs.dist.attach(model, ddp, seed=123)
    model.train_sampler.csr(dev)                       # upload the CSR before timing
    loss_fn = gs.ProblemLosses.classification
    B = args.batch_size

    # seed batches resident in HBM before the timed region; rank r owns rows [r*B, (r+1)*B)
    total = args.steps + args.warmup
    rng = np.random.RandomState(1234)
    pick = rng.randint(0, len(data["train_ids"]), size=(total, world * B))
    ids_all = torch.from_numpy(data["train_ids"][pick][:, rank * B:(rank + 1) * B]).to(dev)
    tg_all = torch.from_numpy(data["targets"][data["train_ids"][pick]][:, rank * B:(rank + 1) * B]).to(dev)

    if args.no_graph:
        args.launch = "eager"
    use_graph = args.launch != "eager"
    step_fn = None
    engine = args.engine
    fused_cls = gs.engine.fused_engine_for(model, store)
    if engine == "fused" and fused_cls is None:
        engine = "autograd"
    if engine == "fused":
        step_fn = fused_cls(model, store, loss_fn, ids_all[0], tg_all[0], ddp=ddp,
                                               capture=args.launch if use_graph else False,
                                               pipelined=args.pipeline)
    elif use_graph:
        try:
            step_fn = gs.engine.CapturedTrainStep(model, store, loss_fn, ids_all[0], tg_all[0], ddp=ddp)
        except Exception as e:                          # report, then measure eager launches
            if rank == 0:
                print("graph capture failed, falling back to eager launches: %r" % (e,), file=sys.stderr)
            use_graph = False
    if step_fn is None:
        def step_fn(ids, tg):
            return model.train_step(ids=ids, feats=store, targets=tg, loss_fn=loss_fn)
    queued = engine == "fused" and not args.pipeline and not args.per_step_copy
    if queued:
        # the whole run's seed batches live in HBM (as the task's timing rule prescribes) and the
        # graph walks them through a device-side batch index: no per-step copies
        step_fn.load_epoch(ids_all, tg_all)
        run_step = lambda k: step_fn.step_queue()
    else:
        run_step = lambda k: step_fn(ids_all[k], tg_all[k])

    def sync():
        if ddp is not None:
            ddp.barrier()
        torch.cuda.synchronize()

    launches0 = gs._native.launch_count()
    for k in range(args.warmup):
        run_step(k)
    sync()
    t0 = time.perf_counter()
    for k in range(args.warmup, total):
        run_step(k)
    sync()
    elapsed = time.perf_counter() - t0
    if hasattr(step_fn, "flush"):
        step_fn.flush()          # pipelined engine: the timed region ran exactly K sample/gather
        torch.cuda.synchronize()  # stages and K compute stages; this drains the last compute stage
    if ddp is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    model.train_sampler.csr(dev).check()

    if rank == 0:
        value = args.steps * B * world / elapsed
        if args.aggregator == "max_pool" and fanout == FANOUT and B == BATCH:
            roof = pool_kernel_roofline(gs, model, store, data, dev)
        else:
            roof = dominant_kernel_roofline(gs, model, store, data, dev)
        line = {
            "metric": "seed-nodes/sec", "value": value, "unit": "seed-nodes/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "Reddit-shaped %s-aggregator %d-layer fanout %s hidden %s "
                                   "(BASELINE configs[1]%s); N=232965 D=602 nnz=%d"
                                   % (args.aggregator, len(fanout), "/".join(map(str, fanout)),
                                      "/".join(map(str, hidden)),
                                      "" if (args.aggregator, fanout, hidden) == ("mean", FANOUT, HIDDEN)
                                      else " variant", data["nnz"]),
                       "batch_per_gpu": B, "global_batch": B * world, "rng": "philox",
                       "engine": engine,
                       "launch": (getattr(step_fn, "capture_mode", None) or ("graph" if use_graph else "eager")),
                       "pipelined": bool(engine == "fused" and args.pipeline), "batch_queue": bool(queued), "parallelism": "dp%d" % world,
                       "kernel_launches_per_step": (gs._native.launch_count() - launches0) / max(total, 1)
                       if args.launch != "graph" else None},
            "frac_of_hbm_gather_roofline": value / world / (HBM_PEAK_GBS * 1e9 / (
                rows_per_seed(fanout) * FEAT_DIM * store.data.element_size())),
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(data, budget_s=args.cpu_budget)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
        sys.stdout.flush()
    if ddp is not None:
        ddp.barrier()
        ddp.close()


if __name__ == "__main__":
    main()
