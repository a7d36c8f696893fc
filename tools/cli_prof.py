"""run.sh:14-17 through train.main on the bench's Reddit-shaped problem (what bench.py's `extra.cli.reddit` times), on its
own: for `rocprofv3 --kernel-trace --stats -- python tools/cli_prof.py [extra train.py flags]`."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
gs = importlib.import_module("pytorch-graphsage_amd")
dev = torch.device("cuda")
gs.ops.set_compute_dtype("bf16"); gs.ops.warmup(dev)
data = bench.synthetic_reddit(seed=0)
store = data["feats"](dev, "bf16")
N = data["adj"].shape[0] - 1
folds = np.array(["test"] * (N + 1), dtype="<U5")
folds[data["train_ids"]] = "train"
rest = np.setdiff1d(np.arange(1, N + 1), data["train_ids"])
folds[rest[:23_000]] = "val"; folds[0] = "dummy"
prob = gs.NodeProblem.from_arrays("classification", bench.N_CLASSES, data["adj"], data["adj"], store, folds, data["targets"], cuda=True)
lines, wall, eng, timing = bench._run_cli(gs, ["--problem-path", "<memory>", "--aggregator-class", "mean", "--sampler-class",
                                       "sparse_uniform_neighbor_sampler", "--epochs", "3"] + sys.argv[1:], prob)
print("rates", bench._epoch_rates(timing), "wall", wall, eng)
