"""Throughput of the BIT-IDENTICAL mode (rng="compat": sel drawn from numpy's legacy MT19937 stream exactly as
the reference does) on the eager module path at the bench workload: host-drawn sel (numpy + H2D per sampler
call) against the stream consumed on the device (gsage_mt_choice_device).  usage: python tools/compat_bench.py"""
import importlib, json, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import bench
gs = importlib.import_module("pytorch-graphsage_amd")
dev = torch.device("cuda")
gs.ops.warmup(dev)
data = bench.synthetic_reddit(seed=0)
store = data["feats"](dev, "bf16")
B, steps = 512, 30
rng = np.random.RandomState(1)
pick = rng.randint(0, len(data["train_ids"]), size=(steps + 5, B))
ids_all = torch.from_numpy(data["train_ids"][pick]).to(dev)
tg_all = torch.from_numpy(data["targets"][data["train_ids"][pick]]).to(dev)
out = {}
for mode in ("host", "device"):
    gs.helpers.legacy_stream.drop()
    gs.helpers.legacy_stream.enabled = mode == "device"
    model = bench.build_model(gs, data["adj"], rng="compat").to(dev)
    model.train_sampler.csr(dev)
    gs.set_seeds(123 ** 2)
    loss_fn = gs.ProblemLosses.classification
    for k in range(5):
        model.train_step(ids=ids_all[k], feats=store, targets=tg_all[k], loss_fn=loss_fn)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(5, steps + 5):
        model.train_step(ids=ids_all[k], feats=store, targets=tg_all[k], loss_fn=loss_fn)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    gs.helpers.legacy_stream.release()
    out[mode] = {"ms_per_step": dt * 1e3, "seed_nodes_per_s": B / dt}
gs.helpers.legacy_stream.enabled = False
print(json.dumps({"workload": "bench.py default (Reddit shape, B=512, fan-out 25/10), eager module path, rng=compat",
                  "sel_on_host": out["host"], "sel_on_device": out["device"]}))
