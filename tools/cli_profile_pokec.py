"""cProfile over train.main for utils/pokec.sh:11-13 on the bench's Pokec-shaped problem (run on the GPU box)."""
import cProfile, importlib, io, os, pstats, sys, time, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
gs = importlib.import_module("pytorch-graphsage_amd")
train = importlib.import_module("pytorch-graphsage_amd.train")
dev = torch.device("cuda")
gs.ops.set_compute_dtype("bf16"); gs.ops.warmup(dev)
rng = np.random.default_rng(3)
Np, K = 1_632_803, 128
adj = rng.integers(0, Np, size=(Np + 1, K), dtype=np.int64); adj[Np] = Np
targets = rng.normal(25.0, 8.0, size=(Np + 1, 1)).astype(np.float64)
pf = rng.choice(np.array(["train", "val"]), size=Np + 1); pf[Np] = "dummy"
prob = gs.NodeProblem.from_arrays("regression_mae", None, adj, adj, None, pf, targets, cuda=True)
argv = ["--problem-path", "<memory>", "--aggregator-class", "mean", "--prep-class", "node_embedding", "--epochs", "3"]
os.environ["GSAGE_TRAIN_TIMING"] = "1"
for rep in range(2):
    pr = cProfile.Profile()
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        pr.enable(); step = train.main(argv, problem=prob); torch.cuda.synchronize(); pr.disable()
    print("== run %d wall %.3f s" % (rep, time.time() - t0), getattr(step, "timing", None))
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(32)
    print("\n".join(l for l in out.getvalue().splitlines() if l.strip())[-6000:])
