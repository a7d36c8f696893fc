"""Where the reference's command line spends its wall time on the bench's Reddit-shaped problem (run on the GPU box):
cProfile over train.main, cumulative top entries.   python tools/cli_profile.py [epochs]"""
import cProfile, importlib, io, os, pstats, sys, time, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
gs = importlib.import_module("pytorch-graphsage_amd")
train = importlib.import_module("pytorch-graphsage_amd.train")
dev = torch.device("cuda")
gs.ops.set_compute_dtype("bf16"); gs.ops.warmup(dev)
data = bench.synthetic_reddit(seed=0)
store = data["feats"](dev, "bf16")
N = data["adj"].shape[0] - 1
folds = np.array(["test"] * (N + 1), dtype="<U5"); folds[data["train_ids"]] = "train"
rest = np.setdiff1d(np.arange(1, N + 1), data["train_ids"]); folds[rest[:23_000]] = "val"; folds[0] = "dummy"
prob = gs.NodeProblem.from_arrays("classification", bench.N_CLASSES, data["adj"], data["adj"], store, folds, data["targets"], cuda=True)
argv = ["--problem-path", "<memory>", "--aggregator-class", "mean", "--sampler-class", "sparse_uniform_neighbor_sampler",
        "--epochs", sys.argv[1] if len(sys.argv) > 1 else "2"]
for rep in range(2):          # (second run: everything that is cached per process -- library, jump table -- is warm)
    pr = cProfile.Profile()
    buf = io.StringIO()
    t0 = time.time()
    with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
        pr.enable(); train.main(argv, problem=prob); torch.cuda.synchronize(); pr.disable()
    print("== run %d wall %.3f s" % (rep, time.time() - t0))
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(28)
    print("\n".join(l for l in out.getvalue().splitlines() if l.strip())[-5500:])
