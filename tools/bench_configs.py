"""Step time at the SHAPES of BASELINE.json configs[3] (Pokec: attention, trainable 64-d node
embeddings, no features, fan-out 20/15, regression_mae) and configs[4] (ogbn-papers100M: mean,
3 layers, fan-out 15/10/5, 128-d bf16 features) on one GPU with synthetic graphs.  Parity for these
shapes is covered by tests (model golden vectors with the same modules, 3-layer engine tests, > 2^31
edge sampler tests); this tool only reports how fast the step is.  papers100M's node count is scaled
(--papers-nodes, default 8 M; the full 111 M x 128 bf16 table is 28 GB and fits one GPU, but building a
3.2e9-edge CSR on the host takes minutes).

    python tools/bench_configs.py pokec|papers [--steps K]
"""
import argparse, importlib, json, sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
from torch.nn import functional as F
import bench
gs = importlib.import_module('pytorch-graphsage_amd')


def graph(n_nodes, mu, sigma, max_deg, seed=0):
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


def timed(step, n_warm, n_steps):
    for k in range(n_warm):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n_warm, n_warm + n_steps):
        step(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n_steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=["pokec", "papers"])
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--papers-nodes", type=int, default=8_000_000)
    args = ap.parse_args()
    dev = torch.device("cuda")
    gs.ops.set_compute_dtype("bf16")
    gs.ops.warmup(dev)
    gs.nn_modules.SparseUniformNeighborSampler.rng_default = "philox"
    B, total = 512, args.steps + args.warmup
    if args.config == "pokec":
        N = 1_632_803
        adj, rng = graph(N, 3.0, 1.1, 8_763)                         # ~ 37 neighbours on average, heavy tail
        fan, dims = (20, 15), (128, 128)
        specs = [{"n_train_samples": f, "n_val_samples": f, "output_dim": h,
                  "activation": (lambda x: x) if i == 1 else F.relu} for i, (f, h) in enumerate(zip(fan, dims))]
        model = gs.GSSupervised(sampler_class=gs.sampler_lookup["sparse_uniform_neighbor_sampler"], adj=adj,
                                train_adj=adj, prep_class=gs.prep_lookup["node_embedding"],
                                aggregator_class=gs.aggregator_lookup["attention"], input_dim=None,
                                n_nodes=adj.shape[0], n_classes=1, layer_specs=specs, lr_init=0.01).to(dev)
        model.train_sampler.csr(dev)
        ids = torch.from_numpy(rng.integers(1, N + 1, size=(total, B))).to(dev)
        tg = torch.from_numpy(rng.integers(15, 60, size=(total, B, 1)).astype(np.float32)).to(dev)
        loss_fn = gs.ProblemLosses.regression_mae
        try:
            step_fn = gs.engine.CapturedTrainStep(model, None, loss_fn, ids[0], tg[0])
            how = "autograd path, hipGraph"
        except Exception as e:                                           # report and fall back to eager launches
            print("graph capture failed (%r): eager" % (e,), file=sys.stderr)
            step_fn = lambda i, t: model.train_step(ids=i, feats=None, targets=t, loss_fn=loss_fn)
            how = "autograd path, eager"
        dt = timed(lambda k: step_fn(ids[k], tg[k]), args.warmup, args.steps)
        rows = 1 + 20 + 300
        line = {"config": "BASELINE configs[3] shape: Pokec-sized graph (N=%d, nnz=%d), node_embedding(64) + "
                          "attention(32), fan-out 20/15, regression_mae" % (N, adj.nnz),
                "ms_per_step": dt * 1e3, "seed_nodes_per_s": B / dt, "engine": how,
                "alg_bytes_per_seed": rows * 64 * 4,
                "dense_table_bytes_per_step": 7 * 4 * 64 * (N + 2)}
    else:
        N = args.papers_nodes
        adj, rng = graph(N, 2.6, 1.2, 30_000)                           # ~ 28 neighbours on average
        data = {"adj": adj}
        feats = torch.zeros(N + 1, 128, dtype=torch.bfloat16, device=dev)
        feats[1:] = torch.randn(N, 128, device=dev).bfloat16()
        store = gs.FeatureStore(feats, 128)
        bench.FEAT_DIM = 128
        model = bench.build_model(gs, adj, aggregator="mean", rng="philox", fanout=(15, 10, 5),
                                  hidden=(128, 128, 128)).to(dev)
        model.train_sampler.csr(dev)
        ids = torch.from_numpy(rng.integers(1, N + 1, size=(total, B))).to(dev)
        tg = torch.from_numpy(rng.integers(0, bench.N_CLASSES, size=(total, B, 1))).to(dev)
        eng = gs.engine.FusedMeanTrainStep(model, store, gs.ProblemLosses.classification, ids[0], tg[0])
        eng.load_epoch(ids, tg)
        dt = timed(lambda k: eng.step_queue(), args.warmup, args.steps)
        rows = 1 + 15 + 150 + 750
        line = {"config": "BASELINE configs[4] shape at %d nodes (nnz=%d): mean, 3 layers, fan-out 15/10/5, "
                          "128-d bf16 features" % (N, adj.nnz),
                "ms_per_step": dt * 1e3, "seed_nodes_per_s": B / dt, "engine": "FusedMeanTrainStep, command list",
                "alg_bytes_per_seed": rows * 128 * 2,
                "frac_of_hbm_gather_roofline": (B / dt) / (bench.HBM_PEAK_GBS * 1e9 / (rows * 128 * 2))}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
