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

`python bench.py --gpus N` without a launcher (WORLD_SIZE unset) starts the N ranks itself
(re-executes under torch.distributed.run on 127.0.0.1) and prints the one line of rank 0.

The timed region is repeated (R times exactly K steps, each between barrier + synchronize) until it
has run for >= 0.5 s; `value` comes from the MEDIAN repeat (K = 20 steps is ~2 ms: one repeat is
noise), min / max are in config.timing.

Extra objects on the line (tier contract, section 4 of the task):
  roofline     the step's dominant launch, timed IN PLACE (engine_roofline): the command list of the queue-mode
               step is re-recorded with HIP-event marks on the dispatch of k_gather_multi_adam (the launch that
               gathers the next batch's level-0 rows, with Adam and the sampler riding along) and of the
               seed-level launch that carries the rest of those gathers; achieved = the frontier rows that launch
               reads x D x 2 B / its mean duration over real steps; peak = 8 TB/s HBM3E.  `step` adds the
               whole-step figure (all 276 rows/seed / ms_per_step).  Every `extra` entry carries the same object
               for ITS dominant launch, timed the same way inside ITS step: K3 over the last hop (max-pool; MFMA),
               K4 over the last hop (attention, Reddit and Pokec shapes; HBM), the gather launch (papers).
  cpu_baseline two CPU restatements of train_step on the host, bounded sample each: the OpenMP C one
               (oracle/gsage_train_omp.c, OpenMP, register-blocked AVX2 / FMA products without BLAS) and the plain-torch port of the reference's
               op sequence (oracle/torch_ref.py, MKL GEMMs, fixed thread count); `value` is the faster.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
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


def cpu_baseline(data, budget_s=12.0, batch=BATCH):
    """train_step on the host cores, bounded sample of the bench workload (same graph, features,
    shapes, fp32).  Primary figure: the OpenMP C restatement (oracle/gsage_train_omp.c, the fastest of a few thread counts --
    SURVEY 8(d)(i)); second figure: the plain-torch port of the reference's op sequence
    (oracle/torch_ref.py) at a FIXED thread count (min(32, cores): more oversubscribes its gathers)."""
    from oracle import cpu as ocpu
    from oracle import torch_ref as tref
    ncpu = os.cpu_count() or 1
    adj = data["adj"]
    indptr, dat = adj.indptr.astype(np.int64), adj.data.astype(np.int64)
    feats_np = data["feats_np"]()
    gen = torch.Generator().manual_seed(0)
    D, h = FEAT_DIM, HIDDEN[0]
    w = {"agg_layers.0.fc_x.weight": torch.randn(h, D, generator=gen) / 25,
         "agg_layers.0.fc_neib.weight": torch.randn(h, D, generator=gen) / 25,
         "agg_layers.1.fc_x.weight": torch.randn(h, 2 * h, generator=gen) / 16,
         "agg_layers.1.fc_neib.weight": torch.randn(h, 2 * h, generator=gen) / 16,
         "fc.weight": torch.randn(N_CLASSES, 2 * h, generator=gen) / 16,
         "fc.bias": torch.zeros(N_CLASSES)}
    rng = np.random.RandomState(0)
    stream = ocpu.LegacyMT19937(123 ** 2)

    def batch_inputs():
        ids = data["train_ids"][rng.randint(0, len(data["train_ids"]), size=batch)]
        sels = [stream.choice(adj.shape[1], (batch, FANOUT[0])),
                stream.choice(adj.shape[1], (batch * FANOUT[0], FANOUT[1]))]
        return ids, data["targets"][ids], sels

    def timed(step, budget):
        step()                                       # page faults, lazy init
        done, t0 = 0, time.time()
        while True:
            step()
            done += 1
            if done >= 3 and time.time() - t0 > budget:
                break
        return done, time.time() - t0

    trainer = ocpu.MeanTrainerOMP({k: v.numpy() for k, v in w.items()}, FANOUT)

    def omp_step():
        ids, tg, sels = batch_inputs()
        trainer.step(0.01, ids, feats_np, tg, indptr, dat, sels)
    # thread count: the step is a chain of short parallel loops (512 .. 13 312 rows), so "all hardware threads" is not
    # the fastest on a 256-thread host; two steps at each candidate, the best one runs the timed sample
    cands = sorted({t for t in (ocpu.omp_threads(), max(1, ncpu // 2), max(1, ncpu // 4), 32, 16) if 1 <= t <= ncpu})
    omp_step()
    trial = {}
    for t in cands:
        ocpu.omp_set_threads(t)
        omp_step()
        t0 = time.time()
        omp_step(); omp_step()
        trial[t] = (time.time() - t0) / 2
    omp_threads = min(trial, key=trial.get)
    ocpu.omp_set_threads(omp_threads)
    n_omp, dt_omp = timed(omp_step, budget_s * 0.6)

    feats = torch.from_numpy(feats_np)
    opt = tref.Adam()
    threads = min(32, ncpu)
    torch.set_num_threads(threads)

    def torch_step():
        ids, tg, sels = batch_inputs()
        tref.train_step(w, opt, 0.01, "classification", ids, feats, torch.from_numpy(tg), indptr, dat, FANOUT,
                        sels, "mean", "identity", adj.shape[0])
    n_t, dt_t = timed(torch_step, budget_s * 0.4)
    omp = {"value": n_omp * batch / dt_omp, "cores": omp_threads,
           "sample": "%d train_steps of %d seeds, fp32, OpenMP C restatement (oracle/gsage_train_omp.c: "
                     "register-blocked AVX2 / FMA products, no BLAS) on %d threads of %d host threads (fastest of %s), "
                     "%.1f s; the restatement neither pins its threads nor places the 560 MB feature table by first "
                     "touch, so it stops scaling at about one NUMA domain of the host: context for the GPU figure, a "
                     "few times below what a tuned multi-socket port would reach"
                     % (n_omp, batch, omp_threads, ncpu, "/".join(str(t) for t in cands), dt_omp)}
    tp = {"value": n_t * batch / dt_t, "cores": threads,
          "sample": "%d train_steps of %d seeds, fp32, oracle/torch_ref.py (the reference's op sequence on stock "
                    "torch CPU kernels: MKL GEMMs) + C sampler, torch %d threads (fixed), %.1f s"
                    % (n_t, batch, threads, dt_t)}
    best, other, names = (tp, omp, ("torch_port", "openmp_c")) if tp["value"] >= omp["value"] else \
        (omp, tp, ("openmp_c", "torch_port"))
    # `value` = the FASTER of the two restatements (a slow baseline flatters nobody); both are listed
    return {"value": best["value"], "unit": "seed-nodes/sec", "cores": best["cores"], "kind": "port",
            "which": names[0], "sample": best["sample"], names[1]: other}


MFMA_PEAK_TFLOPS = 2500.0      # MI355X dense bf16 (MI355X_MICROARCH.md; no sparsity)
PMC_FILES = [os.path.join(ROOT, "profiles", f) for f in ("r06_pmc_launches.json", "r05_pmc_launches.json",
                                                          "r04_pmc_launches.json", "r03_pmc_launches.json")]
TIMING_METHOD = ("HIP start/stop events attached to the launch's dispatch inside the step's command list "
                 "(hipExtLaunchKernel), one read per step")


def timed_launches(eng, run_step, n_steps=48):
    """Mean duration (us) of the launches an engine can time IN PLACE (engine.TIMED): the command lists are
    re-recorded with HIP start / stop events attached to the dispatch of those launches
    (gsage_cmdlist_time_next), n_steps real steps follow (every step a fresh frontier, events on the stream the
    step runs on, one host sync per step to read them), then the lists are recorded again without the events."""
    eng.instrument(True)
    torch.cuda.synchronize()
    for k in range(4):
        run_step(k)
    torch.cuda.synchronize()
    acc = {}
    for k in range(n_steps):
        run_step(4 + k)
        for name, ms in eng.last_launch_ms().items():
            acc.setdefault(name, []).append(ms)
    eng.instrument(False)
    torch.cuda.synchronize()
    return {name: float(np.mean(v)) * 1e3 for name, v in acc.items()}


def exchange_in_list_us(eng, run_step, n_steps=16):
    """Data-parallel runs: the step's exchange (the collective node(s) of the step's command list) bracketed by events
    recorded in the list itself -- mean microseconds over n_steps real steps, on every rank (the steps are collective);
    None when the step is not one recorded list.  (Events between nodes over-state by an event packet's processing,
    ~5 us: an upper bound on what the collective adds to the step.)"""
    if getattr(eng, "capture_mode", None) != "cmdlist" or getattr(eng, "ddp", None) is None:
        return None
    try:
        eng.instrument(True)
        torch.cuda.synchronize()
        for k in range(2):
            run_step(k)
        torch.cuda.synchronize()
        acc = []
        for k in range(n_steps):
            run_step(2 + k)
            ms = eng.last_launch_ms().get("exchange")
            if ms is not None:
                acc.append(ms)
        eng.instrument(False)
        torch.cuda.synchronize()
        return float(np.mean(acc)) * 1e3 if acc else None
    except Exception as e:                      # never lose the line to the instrumentation
        return "error: %r" % (e,)


def pmc_traffic(key):
    """HBM bytes per launch of a timed kernel from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE
    and WRITE_SIZE in separate passes, the guide's gfx950 corrections, a calibration copy per pass): the newest
    profiles/rNN_pmc_launches.json that holds the key (written by tools/refresh_profiles.py from the same bench
    command; a kernel that did not change keeps the round it was last profiled in)."""
    for path in PMC_FILES:
        if not os.path.exists(path):
            continue
        with open(path) as f:
            rec = json.load(f).get(key)
        if rec:
            return rec.get("hbm_read_bytes_per_launch"), rec.get("hbm_write_bytes_per_launch"), \
                "profiles/%s[%s] (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, calibrated)" % (os.path.basename(path), key)
    return None, None, None


def hbm_roofline(kernel, alg_bytes, us, pmc_key, n_steps, **more):
    achieved = alg_bytes / (us * 1e-6) / 1e9
    rd, wr, src = pmc_traffic(pmc_key)
    out = {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": achieved / HBM_PEAK_GBS, "traffic": rd, "traffic_write": wr, "traffic_source": src,
           "alg_bytes_per_launch": alg_bytes, "avg_launch_us": us, "timed_steps": n_steps, "method": TIMING_METHOD}
    out.update(more)
    return out


def mfma_roofline(kernel, flops, us, pmc_key, n_steps, **more):
    achieved = flops / (us * 1e-6) / 1e12
    rd, wr, src = pmc_traffic(pmc_key)
    out = {"bound": "mfma", "kernel": kernel, "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": achieved / MFMA_PEAK_TFLOPS, "traffic": rd, "traffic_write": wr, "traffic_source": src,
           "alg_flops_per_launch": flops, "avg_launch_us": us, "timed_steps": n_steps, "method": TIMING_METHOD}
    out.update(more)
    return out


def engine_roofline(eng, run_step, pmc_key, n_steps=48):
    """The roofline object of an engine's dominant launch, timed in place inside the real step.
      FusedMeanTrainStep  k_gather_multi_adam: gathers of batch i+1 | Adam(i) | K1(i+2); HBM.  Algorithmic bytes =
                          the frontier rows that launch reads x D x sizeof (every sampled row is read exactly once
                          per step, by this launch or by the seed-level launch's gather role -- listed beside it).
      FusedPoolTrainStep  K3 (k_pool_mlp_packed) over the last hop, reading the rows through the frontier's row
                          list as the step does; MFMA.  FLOPs = 2 x rows x D x hidden (nn_modules.py:224).
      FusedAttnTrainStep  K4 (k_attn_aggregate_grp) over the last hop: every child row read once; HBM."""
    us = timed_launches(eng, run_step, n_steps)
    kind = type(eng).__name__
    L = eng.L
    if kind == "FusedMeanTrainStep":
        st = eng.store
        elem = st.data.element_size()
        rows_g, rows_t = eng.gather_launch_rows()
        k1_tail = bool(eng._k1_in_tail())
        seed_name = ("k_mean_tail_mfma (seed level, 16 seeds per workgroup on the matrix cores" if eng._tail_on_mfma()
                     else "k_mean_tail_ce (seed level") + " + gather role on the CUs it leaves idle: batch i+1's last-hop means" + \
            (" + sampler role: K1 of batch i+2)" if k1_tail else ")")
        if rows_t > rows_g and "seed_level" in us:
            # the launch that reads most of the step's frontier rows is the seed-level launch's gather role (round 5:
            # the whole last hop): it is the dominant kernel; the gather launch (hop-1 means | Adam | K1) is listed beside it
            # (the counters of THIS configuration's seed-level launch: reddit_seed_level / papers_seed_level)
            out = hbm_roofline(seed_name, rows_t * st.dim * elem, us["seed_level"], pmc_key.replace("_gather", "_seed_level"),
                               n_steps, rows_per_launch=rows_t)
            # (the rows the gather launch reads for the hop-1 means are the hop-1 nodes' own rows, which K5 / K5b also
            # read in place as x rows: the step's algorithmic bytes count them once, `rows_read` is what this launch moves)
            rows_h1 = sum(eng.size[1:L]) if rows_g == 0 else rows_g        # (children of every hop but the last)
            out["gather_launch"] = {"kernel": "k_gather_multi_adam (in-step: rest of the gathers of batch i+1 | Adam(i)" +
                                              ("" if k1_tail else " | K1(i+2)") + ")",
                                    "gather_rows": rows_g, "rows_read": rows_h1,
                                    "bytes_read_per_launch": rows_h1 * st.dim * elem, "avg_launch_us": us["gather"],
                                    "read_rate_gbs": rows_h1 * st.dim * elem / (us["gather"] * 1e-6) / 1e9,
                                    "note": "bounded by the update's dependent chain and a lane's row trips of the hop-1 "
                                            "means, not by its bytes" if k1_tail else
                                            "bounded by the update's and the sampler's dependent chains, not by its gathers"}
        else:
            out = hbm_roofline("k_gather_multi_adam (in-step: gathers of batch i+1 | Adam(i) | K1(i+2))",
                               rows_g * st.dim * elem, us["gather"], pmc_key, n_steps, rows_per_launch=rows_g)
            if rows_t and "seed_level" in us:
                out["seed_level_launch"] = {"kernel": seed_name, "gather_rows": rows_t,
                                            "alg_bytes_per_launch": rows_t * st.dim * elem,
                                            "avg_launch_us": us["seed_level"]}
        # MFMA utilisation of the step's two contractions (north_star: "MFMA utilisation on the GEMM"): K5 = the
        # level-0 projection [R0 x D] x [D x h], both concat halves in one grouped launch; K5b = every level's weight
        # gradient in one launch (the same FLOPs as the forward projections of those levels)
        if "k5" in us:
            fl = 2.0 * 2 * eng.rows[0] * eng.h[0] * eng.din[0]
            out["k5_launch"] = {"kernel": "k_linear_nt_packed (level-0 projection, x | mean against Wx | Wn)",
                                "alg_flops_per_launch": fl, "avg_launch_us": us["k5"],
                                "achieved_tflops": fl / (us["k5"] * 1e-6) / 1e12,
                                "frac_of_mfma_peak": fl / (us["k5"] * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS}
        if "k5b" in us:
            fl = sum(2.0 * 2 * eng.rows[l] * eng.h[l] * eng.din[l] for l in range(L))
            out["k5b_launch"] = {"kernel": "k_wgrad_multi (every level's weight gradient, one grouped launch)",
                                 "alg_flops_per_launch": fl, "avg_launch_us": us["k5b"],
                                 "achieved_tflops": fl / (us["k5b"] * 1e-6) / 1e12,
                                 "frac_of_mfma_peak": fl / (us["k5b"] * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS}
        return out
    rows = eng.size[L]                                    # rows of the last hop
    if kind == "FusedPoolTrainStep":
        D, Hm = eng.din[0], eng.Hm[0]
        out = mfma_roofline("k_pool_mlp_packed (K3 in-step, last hop: %d rows x %d -> %d, rows read through the "
                            "frontier's row list)" % (rows, D, Hm), 2.0 * rows * D * Hm, us["k3"], pmc_key, n_steps)
        if "k5b" in us:
            flops_b = sum(2.0 * eng.nrows[l] * eng.din[l] * eng.Hm[l] + 2.0 * eng.rows[l] * eng.h[l] *
                          (eng.din[l] + eng.Hm[l]) for l in range(L))
            out["k5b_launch"] = {"kernel": "k_wgrad_multi (every weight gradient of the step, one grouped launch)",
                                 "alg_flops_per_launch": flops_b, "avg_launch_us": us["k5b"],
                                 "achieved_tflops": flops_b / (us["k5b"] * 1e-6) / 1e12,
                                 "frac": flops_b / (us["k5b"] * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS}
        return out
    D = eng.din[0]
    elem = 2 if eng.tdt == torch.bfloat16 else 4
    fused = bool(getattr(eng, "fuse", [False])[0])
    out = hbm_roofline(("k_attn_fused_fwd (K4 with the attention MLP inside, in-step, last hop: %d child rows x %d %s, "
                        "each row read ONCE for att(.), the softmax weights and the weighted sum)" if fused else
                        "k_attn_aggregate_grp (K4 in-step, last hop: %d child rows x %d %s)")
                       % (rows, D, "bf16" if elem == 2 else "fp32"), rows * D * elem, us["k4"], pmc_key, n_steps,
                       rows_per_launch=rows)
    if "k4_bwd" in us:
        out["k4_bwd_launch"] = {"kernel": ("k_attn_fused_bwd (K4' with the attention MLP's backward inside" if fused else
                                           "k_attn_bwd_grp (K4'") + " in-step, last hop: the same rows read once)",
                                "alg_bytes_per_launch": rows * D * elem, "avg_launch_us": us["k4_bwd"],
                                "achieved": rows * D * elem / (us["k4_bwd"] * 1e-6) / 1e9}
    return out


def standalone_gather_probe(gs, model, store, data, dev, fanout, B, reps=40, n_frontiers=8):
    """Fallback roofline object when the step cannot be instrumented (data-parallel runs, eager / hipGraph
    launching): the last hop's gather+mean as ONE stand-alone k_gather_mean launch on fresh
    frontiers, HIP events on the launch stream.  Labelled as a probe: it is not a kernel of the step."""
    ops = gs.ops
    rng = np.random.RandomState(7)
    fronts = []
    for _ in range(n_frontiers):
        ids = torch.from_numpy(data["train_ids"][rng.randint(0, len(data["train_ids"]), size=B)]).to(dev)
        M = B
        for f in fanout[:-1]:
            ids = model.train_sampler(ids, n_samples=f)
            M *= f
        fronts.append(model.train_sampler(ids, n_samples=fanout[-1]))
    cdt = store.data.dtype
    for f in fronts:                                   # warm
        ops.gather_mean(store, f, M, fanout[-1], out_dtype=cdt, out_ld=store.ld)
    torch.cuda.synchronize()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for r in range(reps):
        ops.gather_mean(store, fronts[r % n_frontiers], M, fanout[-1], out_dtype=cdt, out_ld=store.ld)
    stop.record()
    torch.cuda.synchronize()
    dur_s = start.elapsed_time(stop) / 1e3 / reps
    alg = M * fanout[-1] * store.dim * store.data.element_size()
    achieved = alg / dur_s / 1e9
    return {"bound": "hbm", "kernel": "k_gather_mean (stand-alone probe of the last hop, NOT a launch of the step)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": None, "alg_bytes_per_launch": alg, "avg_launch_us": dur_s * 1e6}


def _lognormal_graph(gs, n_nodes, mu, sigma, max_deg, seed=0):
    from scipy import sparse
    rng = np.random.default_rng(seed)
    deg = np.clip(np.exp(rng.normal(mu, sigma, size=n_nodes + 1)).astype(np.int64), 1, max_deg)
    deg[0], deg[1] = 0, max_deg
    indptr = np.zeros(n_nodes + 2, dtype=np.int64)
    np.cumsum(deg, out=indptr[1:])
    data = rng.integers(1, n_nodes + 1, size=int(indptr[-1]), dtype=np.int32)
    adj = sparse.csr_matrix((data, gs.store.row_positions(indptr), indptr), shape=(n_nodes + 1, max_deg))
    adj.has_sorted_indices = True
    return adj, rng


def _timed_steps(step, n_warm, n_steps, finish=None, ddp=None):
    """finish: work the steps deferred (an engine's sync_rows) -- inside the timed region.  With a process group: a
    barrier + synchronize on both sides of the timed region, the MAX over the ranks."""
    def sync():
        if ddp is not None:
            ddp.barrier()
        torch.cuda.synchronize()
    for k in range(n_warm):
        step(k)
    if finish is not None:
        finish()
    sync()
    t0 = time.perf_counter()
    for k in range(n_warm, n_warm + n_steps):
        step(k)
    if finish is not None:
        finish()
    sync()
    dt = time.perf_counter() - t0
    if ddp is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    return dt / n_steps


def _placeholder_adj():
    from scipy import sparse
    return sparse.csr_matrix((np.array([1, 1]), np.array([0, 0]), np.array([0, 0, 1, 2])), shape=(3, 1))


def extra_papers(gs, dev, steps=60, warmup=5, n_nodes=111_059_956, B=BATCH, ddp=None):
    """BASELINE configs[4] on one GPU at its REAL size: 111 059 956 nodes, ~3.2e9 edges (int64 row offsets, 13 GB of
    int32 neighbour ids), 128-d bf16 features (28 GB), mean aggregator, three layers, fan-out 15/10/5 -- graph and
    table generated on the device (store.DeviceCSR.synthetic / FeatureStore.synthetic; ~43 GB of the 288 GB).
    Parity at this size: tests/test_gpu_large.py::test_three_layer_engine_step_at_papers_scale."""
    from torch.nn import functional as F
    n_rows = n_nodes + 1
    csr = gs.DeviceCSR.synthetic(n_rows, 14, 44, dev, max_deg=4096, seed=1, empty_every=1000)
    store = gs.FeatureStore.synthetic(n_rows, 128, dev, dtype="bf16", seed=2)
    fan, dims = (15, 10, 5), (128, 128, 128)
    gs.nn_modules.SparseUniformNeighborSampler.rng_default = "philox"
    specs = [{"n_train_samples": f, "n_val_samples": f, "output_dim": h,
              "activation": (lambda x: x) if i == 2 else F.relu} for i, (f, h) in enumerate(zip(fan, dims))]
    model = gs.GSSupervised(sampler_class=gs.sampler_lookup["sparse_uniform_neighbor_sampler"], adj=_placeholder_adj(),
                            train_adj=_placeholder_adj(), prep_class=gs.prep_lookup["identity"],
                            aggregator_class=gs.aggregator_lookup["mean"], input_dim=128, n_nodes=n_rows,
                            n_classes=N_CLASSES, layer_specs=specs, lr_init=0.01).to(dev)
    model.train_sampler.seed = 123
    model.train_sampler.use_device_csr(csr)
    world, rank = (ddp.world, ddp.rank) if ddp is not None else (1, 0)
    if ddp is not None:
        gs.dist.attach(model, ddp, seed=123)
    total = steps + warmup
    rng = np.random.default_rng(0)              # (every rank draws the global batches and keeps its columns)
    ids = torch.from_numpy(rng.integers(1, n_rows, size=(total, world * B))[:, rank * B:(rank + 1) * B].copy()).to(dev)
    tg = torch.from_numpy(rng.integers(0, N_CLASSES, size=(total, world * B, 1))[:, rank * B:(rank + 1) * B].copy()).to(dev)
    eng = gs.engine.FusedMeanTrainStep(model, store, gs.ProblemLosses.classification, ids[0], tg[0], ddp=ddp)
    eng.load_epoch(ids, tg)
    dt = _timed_steps(lambda k: eng.step_queue(), warmup, steps, ddp=ddp)
    csr.check()
    rows = 1 + 15 + 150 + 750
    rec = {"config": "BASELINE configs[4] at its real size (N=%d, nnz=%d): mean, 3 layers, fan-out 15/10/5, 128-d bf16 "
                     "features (a %.1f GB table: the uncached reference point for the gather), one GPU"
                     % (n_nodes, csr.nnz, store.data.numel() * 2 / 1e9),
           "ms_per_step": dt * 1e3, "value": world * B / dt, "unit": "seed-nodes/sec", "engine": "FusedMeanTrainStep",
           "alg_bytes_per_seed": rows * 128 * 2,
           "frac_of_hbm_gather_roofline": (B / dt) / (HBM_PEAK_GBS * 1e9 / (rows * 128 * 2))}
    if ddp is None:
        rec["roofline"] = engine_roofline(eng, lambda k: eng.step_queue(), "papers_gather", n_steps=32)
    else:
        rec["exchange_us"] = exchange_in_list_us(eng, lambda k: eng.step_queue())
    rec["_engine"] = eng
    return rec


def extra_pokec(gs, dev, steps=40, warmup=5, B=BATCH, precision=None, engine="fused", ddp=None):
    """BASELINE configs[3] at its SHAPE on one GPU: Pokec-sized graph (1.63 M nodes, ~6e7 edges), no features,
    trainable 64-d node embeddings (node_embedding prep), attention aggregator (hidden 32), fan-out 20/15,
    regression_mae.  The step keeps the reference's DENSE embedding-gradient semantics (every row of the 418 MB
    table moves every step; dense_table_bytes_per_step is what streaming it would cost), applied row by row when a
    row is touched or read (engine.FusedAttnTrainStep.sync_rows)."""
    from torch.nn import functional as F
    if precision:
        gs.ops.set_compute_dtype(precision)
    N = 1_632_803
    adj, rng = _lognormal_graph(gs, N, 3.0, 1.1, 8_763)
    fan, dims = (20, 15), (128, 128)
    gs.nn_modules.SparseUniformNeighborSampler.rng_default = "philox"
    specs = [{"n_train_samples": f, "n_val_samples": f, "output_dim": h,
              "activation": (lambda x: x) if i == 1 else F.relu} for i, (f, h) in enumerate(zip(fan, dims))]
    model = gs.GSSupervised(sampler_class=gs.sampler_lookup["sparse_uniform_neighbor_sampler"], adj=adj, train_adj=adj,
                            prep_class=gs.prep_lookup["node_embedding"],
                            aggregator_class=gs.aggregator_lookup["attention"], input_dim=None,
                            n_nodes=adj.shape[0], n_classes=1, layer_specs=specs, lr_init=0.01).to(dev)
    model.train_sampler.seed = 123
    model.train_sampler.csr(dev)
    world, rank = (ddp.world, ddp.rank) if ddp is not None else (1, 0)
    if ddp is not None:
        gs.dist.attach(model, ddp, seed=123)
    total = steps + warmup
    ids = torch.from_numpy(rng.integers(1, N + 1, size=(total, world * B))[:, rank * B:(rank + 1) * B].copy()).to(dev)
    tg = torch.from_numpy(rng.integers(15, 60, size=(total, world * B, 1))[:, rank * B:(rank + 1) * B].astype(np.float32)).to(dev)
    loss_fn = gs.ProblemLosses.regression_mae
    cls = gs.engine.fused_engine_for(model, None, ddp=ddp) if engine == "fused" else None
    if cls is not None:
        step_fn = cls(model, None, loss_fn, ids[0], tg[0], ddp=ddp, capture=os.environ.get("GSAGE_POKEC_LAUNCH", "cmdlist"))
        how = "%s (native attention step: K4 / K5 / K5b / K6, no autograd below the head), %s" % (
            cls.__name__, step_fn.capture_mode)
    else:
        step_fn = gs.engine.CapturedTrainStep(model, None, loss_fn, ids[0], tg[0])
        how = "CapturedTrainStep (native K4 / K5 / K5b / K6 kernels under autograd, hipGraph)"
    # deferred table rows (engine.sync_rows): the timed region ends with every row of the table settled, i.e. it
    # pays for one dense catch-up pass per `steps` steps (a training run pays one per epoch, before evaluation)
    deferred = bool(getattr(step_fn, "lazy_rows", False))
    dt = _timed_steps(lambda k: step_fn(ids[k], tg[k]), warmup, steps, finish=step_fn.sync_rows if deferred else None, ddp=ddp)
    if deferred:
        how += "; table rows updated when touched or read (gsage_rows_*: bit-identical to the dense update), " \
               "all rows settled inside the timed region"
    model.train_sampler.csr(dev).check()
    rows = 1 + 20 + 300
    rec = {"config": "BASELINE configs[3] shape on one GPU: Pokec-sized graph (N=%d, nnz=%d), node_embedding(64) + "
                     "attention(32), fan-out 20/15, regression_mae" % (N, adj.nnz),
           "ms_per_step": dt * 1e3, "value": world * B / dt, "unit": "seed-nodes/sec",
           "engine": how, "alg_bytes_per_seed": rows * 64 * 4, "dense_table_bytes_per_step": 7 * 4 * 64 * (N + 2),
           "frac_of_hbm_gather_roofline": (B / dt) / (HBM_PEAK_GBS * 1e9 / (rows * 64 * 4))}
    if cls is not None and step_fn.capture_mode == "cmdlist":
        if ddp is None:
            rec["roofline"] = engine_roofline(step_fn, lambda k: step_fn(ids[k % total], tg[k % total]), "pokec_k4",
                                              n_steps=32)
        else:
            rec["exchange_us"] = exchange_in_list_us(step_fn, lambda k: step_fn(ids[k % total], tg[k % total]))
        step_fn.sync_rows()
    rec["_engine"] = step_fn
    return rec


def _run_cli(gs, argv, problem):
    """train.main on a problem held in memory, stdout captured -> (list of JSON lines, wall seconds)"""
    import contextlib
    import io
    train = importlib.import_module("pytorch-graphsage_amd.train")
    buf, err = io.StringIO(), io.StringIO()
    t0 = time.time()
    os.environ["GSAGE_TRAIN_TIMING"] = "1"            # (train_fused then brackets every epoch's batch loop)
    try:
        with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(err):
            step = train.main(argv, problem=problem)
    finally:
        os.environ.pop("GSAGE_TRAIN_TIMING", None)
    torch.cuda.synchronize()
    wall = time.time() - t0
    lines = [json.loads(ln) for ln in buf.getvalue().splitlines() if ln.startswith("{")]
    engine = [ln for ln in err.getvalue().splitlines() if ln.startswith("gsage:")]
    return lines, wall, engine, list(getattr(step, "timing", []) or [])


def _epoch_rates(timing):
    """seed-nodes/s of every epoch's batch loop (train_fused's own bracket: per-batch log and metric included, the
    epoch's sampler draws and the validation pass not), and the same with the epoch's draws"""
    return ([t["seeds"] / t["loop_s"] for t in timing], [t["seeds"] / t["with_draws_s"] for t in timing])


def _whole_epoch_rates(timing):
    """seed-nodes/s of every WHOLE epoch: the epoch's sampler draws + its batch loop (per-batch log and metric) + its
    validation pass -- what a run of many epochs converges to; the command's fixed set-up (model, graph upload,
    engine construction and recording) is wall_s minus the epochs"""
    return [t["seeds"] / t["epoch_s"] for t in timing if "epoch_s" in t]


def extra_cli(gs, dev, data, store):
    """The drop-in surface itself (round-3 verdict, item 6): the reference's command lines through train.main --
    --engine auto, the reference's chunks and generators, one JSON line and one device-metric readback PER BATCH
    (train.py:150-158) -- on problems held in memory (NodeProblem.from_arrays; loading a problem file is not timed).
      reddit   run.sh:14-17 (sparse sampler, mean) on the bench's Reddit-shaped graph, 2 epochs: seed-nodes/s of the
               training loop (per-batch log included) beside the engine's own figure
      pokec    utils/pokec.sh:11-13 (DEFAULT dense sampler, node_embedding, mean, 3 epochs) on a Pokec-shaped problem
               (1.63 M nodes, K = 128 dense adjacency, 50 / 50 train / val as utils/convert-pokec.py:48): wall seconds
               of the whole command incl. its three validation passes, beside the reference's published 147.33 s
               (utils/pokec.sh:15 -- unknown hardware: context, not a baseline)."""
    out = {}
    N = data["adj"].shape[0] - 1
    folds = np.array(["test"] * (N + 1), dtype="<U5")
    folds[data["train_ids"]] = "train"
    rest = np.setdiff1d(np.arange(1, N + 1), data["train_ids"])
    folds[rest[:23_000]] = "val"
    folds[0] = "dummy"
    prob = gs.NodeProblem.from_arrays("classification", N_CLASSES, data["adj"], data["adj"], store, folds,
                                      data["targets"], cuda=True)
    epochs = 2
    try:
        lines, wall, eng, timing = _run_cli(gs, ["--problem-path", "<memory>", "--aggregator-class", "mean",
                                                 "--sampler-class", "sparse_uniform_neighbor_sampler", "--epochs",
                                                 str(epochs)], prob)
        n_train = int((folds == "train").sum())
        rates, rates_draws = _epoch_rates(timing)
        out["reddit"] = {"command": "run.sh:14-17 (--aggregator-class mean --sampler-class sparse_uniform_neighbor_sampler), "
                                    "--epochs %d, defaults otherwise (--engine auto, --rng compat, per-batch JSON)" % epochs,
                         "engine": eng[-1] if eng else None, "train_nodes": n_train,
                         "batches_per_epoch": len([ln for ln in lines if ln.get("epoch") == 0 and "epoch_progress" in ln]),
                         "cli_seeds_per_s": rates[-1] if rates else None, "cli_seeds_per_s_by_epoch": rates,
                         "cli_seeds_per_s_incl_epoch_draws": rates_draws,
                         # the headline of this entry: seeds of every epoch / wall seconds of the WHOLE command (engine
                         # construction, every epoch's sampler draws, the per-batch log, the validation passes)
                         "end_to_end_seeds_per_s": epochs * n_train / wall,
                         "whole_epoch_seeds_per_s": _whole_epoch_rates(timing),
                         "val_s_by_epoch": [t.get("val_s") for t in timing],
                         "setup_s": wall - sum(t.get("epoch_s", 0.0) for t in timing),
                         "wall_s": wall, "val_metric": lines[-1].get("val_metric") if lines else None}
    except Exception as e:
        out["reddit"] = {"error": repr(e)}
    del prob
    torch.cuda.empty_cache()
    try:
        rng = np.random.default_rng(3)
        Np, K = 1_632_803, 128
        adj = rng.integers(0, Np, size=(Np + 1, K), dtype=np.int64)
        adj[Np] = Np
        targets = rng.normal(25.0, 8.0, size=(Np + 1, 1)).astype(np.float64)
        pf = rng.choice(np.array(["train", "val"]), size=Np + 1)
        pf[Np] = "dummy"
        prob = gs.NodeProblem.from_arrays("regression_mae", None, adj, adj, None, pf, targets, cuda=True)
        lines, wall, eng, timing = _run_cli(gs, ["--problem-path", "<memory>", "--aggregator-class", "mean", "--prep-class",
                                                 "node_embedding", "--epochs", "3"], prob)
        n_train = int((pf == "train").sum())
        rates, _rd = _epoch_rates(timing)
        out["pokec"] = {"command": "utils/pokec.sh:11-13 (--aggregator-class mean --prep-class node_embedding --epochs 3; "
                                   "DEFAULT dense sampler), defaults otherwise",
                        "engine": eng[-1] if eng else None, "train_nodes": n_train, "wall_s": wall,
                        "end_to_end_seeds_per_s": 3 * n_train / wall,
                        "whole_epoch_seeds_per_s": _whole_epoch_rates(timing),
                        "val_s_by_epoch": [t.get("val_s") for t in timing],
                        "setup_s": wall - sum(t.get("epoch_s", 0.0) for t in timing),
                        "cli_seeds_per_s_by_epoch": rates, "final": lines[-1] if lines else None,
                        "reference_published_wall_s": 147.32675504684448,
                        "reference_note": "utils/pokec.sh:15, the reference's only published figure: unknown hardware, "
                                          "real Pokec data -- context only"}
    except Exception as e:
        out["pokec"] = {"error": repr(e)}
    return out


def extra_ddp_1rank(args):
    """The data-parallel form of the headline step with a ONE-rank RCCL group on this GPU (GSAGE_FORCE_DDP=1): the
    step as one command list whose exchange is an RCCL call issued by the library (gsage_comm_all_reduce_f32), the
    norm of the averaged gradient formed inside the launch that carries Adam -- in the order data-parallel runs use
    (inline) and in the opt-in order with the exchange on the list's side stream beside the next batch's gathers
    (GSAGE_DDP_OVERLAP=1).  What the driver's 1-GPU box can record of the multi-GPU tax, every round; the process
    group lives in a child process."""
    out = {}
    for name, ov in (("inline", "0"), ("overlapped", "1")):
        env = dict(os.environ)
        env.update({"GSAGE_FORCE_DDP": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(_free_port()), "GSAGE_DDP_OVERLAP": ov, "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--no-cpu-baseline", "--extra", "", "--min-time", "0.3", "--batch-size", str(args.batch_size),
               "--precision", args.precision]
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
            line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            out[name] = {"ms_per_step": line["ms_per_step"], "value": line["value"],
                         "collective": line["config"].get("collective"), "native_comm": line["config"].get("native_comm"),
                         "one_list": line["config"].get("one_list"),
                         "kernel_launches_per_step": line["config"].get("kernel_launches_per_step")}
        except Exception as e:
            out[name] = {"error": repr(e)}
    out["config"] = ("BASELINE configs[1] step with a 1-rank RCCL group (GSAGE_FORCE_DDP=1): `inline` is the order "
                     "data-parallel runs use (the collective on the step's own stream, Adam with the norm formed in the "
                     "gather launch), `overlapped` the opt-in order with the collective on the list's side stream "
                     "(GSAGE_DDP_OVERLAP=1: measured slower on this platform, DESIGN.md section 6)")
    return out


def extra_b4096(args):
    """The headline step at B = 4 096 seeds per GPU ("512/GPU (also sweep 4096)"): the same engine, eight
    times the work per launch -- fixed launch costs amortised, the seed level 256 workgroups wide.  A child process
    (its own seed queue and engine)."""
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--extra", "",
           "--min-time", "0.3", "--batch-size", "4096", "--precision", args.precision]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        return {"config": "BASELINE configs[1] at B = 4096 seeds per GPU", "ms_per_step": line["ms_per_step"],
                "value": line["value"], "unit": "seed-nodes/sec",
                "frac_of_hbm_gather_roofline": line["frac_of_hbm_gather_roofline"],
                "kernel_launches_per_step": line["config"].get("kernel_launches_per_step"),
                "gather_launch": {k: line["roofline"].get(k) for k in ("achieved", "frac", "avg_launch_us",
                                                                       "alg_bytes_per_launch")}}
    except Exception as e:
        return {"error": repr(e)}


def extra_fp32(args):
    """BASELINE configs[1] with fp32 storage and exact-fp32 arithmetic (`--precision fp32`: the reference's own number
    format, reference problem.py:119; BASELINE.md section 4's tight-parity row, roofline 8 TB/s / (276 x 602 x 4 B) =
    12.0 M seed-nodes/s): the same engine code instantiated on float.  A child process (its own feature table)."""
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--no-cpu-baseline", "--extra", "", "--min-time", "0.3", "--batch-size", str(args.batch_size), "--precision", "fp32"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        return {"config": "BASELINE configs[1] with fp32 features, fp32 activations and exact-fp32 MFMA "
                          "(v_mfma_f32_32x32x2_f32): the tight-parity instantiation of the same engine",
                "ms_per_step": line["ms_per_step"], "value": line["value"], "unit": "seed-nodes/sec", "dtype": "fp32",
                "frac_of_hbm_gather_roofline": line["frac_of_hbm_gather_roofline"],
                "roofline_seeds_per_s": HBM_PEAK_GBS * 1e9 / (rows_per_seed(FANOUT) * FEAT_DIM * 4),
                "kernel_launches_per_step": line["config"].get("kernel_launches_per_step"),
                "dominant_launch": {k: line["roofline"].get(k) for k in ("kernel", "achieved", "frac", "avg_launch_us",
                                                                         "alg_bytes_per_launch")}}
    except Exception as e:
        return {"error": repr(e)}


def _free_port():
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def self_launch(n):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks here (one process per
    GPU under torch.distributed.run, rendezvous on 127.0.0.1) with the same arguments; rank 0 prints the
    line.  With fewer than N GPUs visible the run only makes sense as a test of the data-parallel path:
    GSAGE_DIST_BACKEND=gloo lets the ranks share the visible GPU (RCCL refuses duplicate devices)."""
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if n_dev < n and env.get("GSAGE_DIST_BACKEND") != "gloo":
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (set GSAGE_DIST_BACKEND=gloo to let the "
                         "ranks share one GPU for a functional test)" % (n, n_dev))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch-size", type=int, default=BATCH, help="seed nodes per GPU per step")
    ap.add_argument("--aggregator", type=str, default="mean")
    ap.add_argument("--config", type=str, default="reddit", choices=["reddit", "pokec", "papers"],
                    help="reddit: BASELINE configs[1] / [2] (with --aggregator) on the Reddit-shaped graph (default); pokec / "
                         "papers: BASELINE configs[3] / configs[4] at their shapes on their own synthetic graphs -- all of "
                         "them compose with --gpus N (seed shards, one exchange per step)")
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
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--min-time", type=float, default=0.5,
                    help="repeat the K-step timed region until it has run this many seconds in total")
    ap.add_argument("--extra", type=str, default="max_pool,attention,papers,pokec,ddp_1rank,cli,b4096,fp32",
                    help="comma-separated additional configurations measured after the main line (N=1 only) and "
                         "reported under `extra`: an aggregator name (same graph), `papers` / `pokec` (BASELINE "
                         "configs[4] / configs[3] shapes on their own synthetic graphs); '' for none")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    import __graft_entry__
    __graft_entry__.ensure_built()
    gs = importlib.import_module("pytorch-graphsage_amd")
    ops = gs.ops
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d, "
                         "or without a launcher (bench.py then starts the ranks itself)" % (args.gpus, world, args.gpus))
    ddp = gs.dist.init_from_env(cuda=True)
    rank = ddp.rank if ddp is not None else 0
    dev = torch.device("cuda", torch.cuda.current_device())
    ops.set_compute_dtype(args.precision)
    ops.warmup(dev)

    if args.config != "reddit":
        # BASELINE configs[3] / configs[4] as the main line (one command per configuration and GPU count)
        fn = extra_pokec if args.config == "pokec" else extra_papers
        rec = fn(gs, dev, steps=args.steps, warmup=args.warmup, B=args.batch_size, ddp=ddp)
        rec.pop("_engine", None)
        if rank == 0:
            line = {"metric": "seed-nodes/sec", "value": rec["value"], "unit": "seed-nodes/sec", "n_gpus": world,
                    "steps": args.steps, "warmup": args.warmup, "ms_per_step": rec["ms_per_step"], "higher_is_better": True,
                    "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                    "config": {"workload": rec["config"], "batch_per_gpu": args.batch_size,
                               "global_batch": args.batch_size * world, "rng": "philox", "engine": rec["engine"],
                               "parallelism": "dp%d" % world, "ranks": world,
                               "collective": (torch.distributed.get_backend() if ddp is not None else None),
                               "process_group_ranks": (torch.distributed.get_world_size() if ddp is not None else None),
                               "exchange_in_list_us": rec.get("exchange_us")},
                    "frac_of_hbm_gather_roofline": rec.get("frac_of_hbm_gather_roofline"),
                    "roofline": rec.get("roofline"), "cpu_baseline": None, "extra": {}}
            print(json.dumps(line))
            sys.stdout.flush()
        if ddp is not None:
            ddp.barrier()
            ddp.close()
        return

    data = synthetic_reddit(seed=0)
    store = data["feats"](dev, args.precision)
    fanout = tuple(int(v) for v in args.fanout.split(","))
    hidden = tuple(int(v) for v in args.hidden.split(","))
    assert len(fanout) == len(hidden)
    B = args.batch_size
    loss_fn = gs.ProblemLosses.classification
    if args.no_graph:
        args.launch = "eager"
    total = args.steps + args.warmup
    rng = np.random.RandomState(1234)
    # seed batches resident in HBM before the timed region; rank r owns columns [r*B, (r+1)*B)
    pick = rng.randint(0, len(data["train_ids"]), size=(total, world * B))
    ids_all = torch.from_numpy(data["train_ids"][pick][:, rank * B:(rank + 1) * B]).to(dev)
    tg_all = torch.from_numpy(data["targets"][data["train_ids"][pick]][:, rank * B:(rank + 1) * B]).to(dev)

    def sync():
        if ddp is not None:
            ddp.barrier()
        torch.cuda.synchronize()

    def measure(aggregator, min_time):
        """Build model + engine for `aggregator`, warm up, time R x exactly K steps.  Returns a dict."""
        model = build_model(gs, data["adj"], aggregator=aggregator, rng="philox", fanout=fanout,
                            hidden=hidden).to(dev)
        if ddp is not None:
            gs.dist.attach(model, ddp, seed=123)
        model.train_sampler.csr(dev)                       # upload the CSR before timing
        use_graph = args.launch != "eager"
        step_fn, engine = None, args.engine
        fused_cls = gs.engine.fused_engine_for(model, store, explain=(rank == 0))
        if engine == "fused" and fused_cls is None:
            engine = "autograd"                             # (fused_engine_for said on stderr what is not covered)
        if engine == "fused":
            step_fn = fused_cls(model, store, loss_fn, ids_all[0], tg_all[0], ddp=ddp,
                                capture=args.launch if use_graph else False, pipelined=args.pipeline)
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
            # step walks them through a device-side batch index: no per-step copies
            step_fn.load_epoch(ids_all, tg_all)
            run_step = lambda k: step_fn.step_queue()
        else:
            run_step = lambda k: step_fn(ids_all[k % total], tg_all[k % total])

        for k in range(args.warmup):
            run_step(k)
        sync()
        times, launches, k = [], [], args.warmup
        while True:
            l0 = gs._native.launch_count()
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):                 # EXACTLY K steps per timed repeat
                run_step(k)
                k += 1
            sync()
            dt = time.perf_counter() - t0
            if hasattr(step_fn, "flush") and args.pipeline:
                step_fn.flush()      # pipelined engine: drain the last compute stage (outside the clock)
                torch.cuda.synchronize()
            if ddp is not None:
                t = torch.tensor([dt], device=dev, dtype=torch.float64)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                dt = float(t.item())
            times.append(dt)
            launches.append((gs._native.launch_count() - l0) / args.steps)
            stop = sum(times) >= min_time or len(times) >= 200
            if ddp is not None:                         # every rank must take the same decision
                flag = torch.tensor([1 if stop else 0], device=dev)
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
                stop = bool(flag.item())
            if stop:
                break
        model.train_sampler.csr(dev).check()
        xus = exchange_in_list_us(step_fn, run_step) if ddp is not None else None      # (collective steps: every rank)
        med = float(np.median(times))
        return {"model": model, "step_fn": step_fn, "engine": engine, "queued": queued, "use_graph": use_graph,
                "elapsed": med, "times": times, "launches_per_step": float(np.median(launches)), "exchange_us": xus}

    res = measure(args.aggregator, args.min_time)
    elapsed, step_fn, model = res["elapsed"], res["step_fn"], res["model"]
    elem = store.data.element_size()

    if rank == 0:
        value = args.steps * B * world / elapsed
        instrumentable = (res["queued"] and ddp is None and getattr(step_fn, "capture_mode", None) == "cmdlist")
        if instrumentable:
            roof = engine_roofline(step_fn, lambda k: step_fn.step_queue(),
                                   {"mean": "reddit_gather", "max_pool": "maxpool_k3",
                                    "attention": "attention_k4"}.get(args.aggregator, args.aggregator))
        else:
            roof = standalone_gather_probe(gs, model, store, data, dev, fanout, B)
        step_alg = rows_per_seed(fanout) * FEAT_DIM * elem * B
        roof["step"] = {"alg_bytes_per_step": step_alg, "ms_per_step": elapsed / args.steps * 1e3,
                        "achieved": step_alg / (elapsed / args.steps) / 1e9,
                        "frac": step_alg / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS}
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
                       "engine": res["engine"],
                       "launch": (getattr(step_fn, "capture_mode", None) or ("graph" if res["use_graph"] else "eager")),
                       "pipelined": bool(res["engine"] == "fused" and args.pipeline),
                       "batch_queue": bool(res["queued"]), "parallelism": "dp%d" % world,
                       "ranks": world, "collective": (torch.distributed.get_backend() if ddp is not None else None),
                       "process_group_ranks": (torch.distributed.get_world_size() if ddp is not None else None),
                       "exchange_in_list_us": res.get("exchange_us"),
                       "native_comm": bool(getattr(step_fn, "comm", None) is not None) if ddp is not None else None,
                       "one_list": bool(step_fn._one_list_ddp()) if (ddp is not None and hasattr(step_fn, "_one_list_ddp")) else None,
                       "kernel_launches_per_step": res["launches_per_step"] if args.launch != "graph" else None,
                       "timing": {"repeats": len(res["times"]), "steps_per_repeat": args.steps,
                                  "median_s": elapsed, "min_s": min(res["times"]), "max_s": max(res["times"]),
                                  "value_from": "median repeat"}},
            "frac_of_hbm_gather_roofline": value / world / (HBM_PEAK_GBS * 1e9 / (
                rows_per_seed(fanout) * FEAT_DIM * elem)),
            "roofline": roof,
        }
        # other BASELINE configurations on the same graph, so that the driver's run measures them too
        extra = {}
        if world == 1:
            del res, step_fn, model
            torch.cuda.empty_cache()
            names = [a for a in args.extra.split(",") if a and a != args.aggregator]
            for agg in [a for a in names if a not in ("papers", "pokec", "ddp_1rank", "cli", "b4096", "fp32")]:
                r2 = measure(agg, min(args.min_time, 0.3))
                e2 = r2["elapsed"]
                rec = {"config": {"max_pool": "BASELINE configs[2] shape on one GPU",
                                  "attention": "attention aggregators on the Reddit-shaped graph (config 4's aggregator "
                                               "on config 2's data), native engine"}.get(agg, agg),
                       "ms_per_step": e2 / args.steps * 1e3, "value": args.steps * B / e2,
                       "unit": "seed-nodes/sec", "engine": type(r2["step_fn"]).__name__,
                       "kernel_launches_per_step": r2["launches_per_step"], "repeats": len(r2["times"])}
                if r2["queued"] and getattr(r2["step_fn"], "capture_mode", None) == "cmdlist":
                    rec["roofline"] = engine_roofline(r2["step_fn"], lambda k, e=r2["step_fn"]: e.step_queue(),
                                                      {"max_pool": "maxpool_k3", "attention": "attention_k4"}.get(agg, agg),
                                                      n_steps=32)
                    rec["frac_of_hbm_gather_roofline"] = (args.steps * B / e2) / (HBM_PEAK_GBS * 1e9 / (
                        rows_per_seed(fanout) * FEAT_DIM * elem))
                extra[agg] = rec
                del r2
                torch.cuda.empty_cache()
            if "cli" in names:
                try:
                    extra["cli"] = extra_cli(gs, dev, data, store)
                except Exception as e:
                    extra["cli"] = {"error": repr(e)}
            store = None                         # the other shapes bring their own graphs and tables
            torch.cuda.empty_cache()
            for name, fn in (("papers", extra_papers), ("pokec", extra_pokec)):
                if name in names:
                    try:
                        extra[name] = fn(gs, dev)
                        extra[name].pop("_engine", None)
                    except Exception as e:                      # never lose the main line to an extra
                        extra[name] = {"error": repr(e)}
                    torch.cuda.empty_cache()
            if "ddp_1rank" in names and ddp is None:
                extra["ddp_1rank"] = extra_ddp_1rank(args)
            if "b4096" in names and args.batch_size != 4096:
                extra["b4096"] = extra_b4096(args)
            if "fp32" in names and args.precision != "fp32":
                extra["fp32"] = extra_fp32(args)
        line["extra"] = extra
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
