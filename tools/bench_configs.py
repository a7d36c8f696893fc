"""Step time at the SHAPES of BASELINE.json configs[3] (Pokec: attention, trainable 64-d node embeddings,
no features, fan-out 20/15, regression_mae) and configs[4] (ogbn-papers100M: mean, 3 layers, fan-out 15/10/5,
128-d bf16 features) on one GPU with synthetic graphs -- the same measurements bench.py reports under
`extra` (bench.extra_pokec / bench.extra_papers), callable on their own.

    python tools/bench_configs.py pokec|papers [--steps K] [--papers-nodes N]
"""
import argparse, importlib, json, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
gs = importlib.import_module("pytorch-graphsage_amd")

ap = argparse.ArgumentParser()
ap.add_argument("config", choices=["pokec", "papers"])
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--papers-nodes", type=int, default=111_059_956)
ap.add_argument("--precision", type=str, default=None)
ap.add_argument("--engine", type=str, default="fused", choices=["fused", "autograd"])
args = ap.parse_args()
dev = torch.device("cuda")
gs.ops.set_compute_dtype("bf16")
gs.ops.warmup(dev)
if args.config == "pokec":
    rec = bench.extra_pokec(gs, dev, steps=args.steps, warmup=args.warmup, precision=args.precision, engine=args.engine)
else:
    rec = bench.extra_papers(gs, dev, steps=args.steps, warmup=args.warmup, n_nodes=args.papers_nodes)
rec.pop("_engine", None)
print(json.dumps(rec))
