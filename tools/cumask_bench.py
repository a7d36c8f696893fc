"""Do CU-masked streams split the chip?  Times the step's kernels on streams restricted to N of the 256 CUs
(alone), then the gather launch and the forward/backward chain SIDE BY SIDE on complementary masks.
usage: python tools/cumask_bench.py"""
import importlib, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import bench
gs = importlib.import_module("pytorch-graphsage_amd")
ops, nat = gs.ops, gs._native
dev = torch.device("cuda")
ops.warmup(dev)
data = bench.synthetic_reddit(seed=0)
store = data["feats"](dev, "bf16")
model = bench.build_model(gs, data["adj"], rng="philox").to(dev)
B = 512
rng = np.random.RandomState(0)
pick = rng.randint(0, len(data["train_ids"]), size=(16, B))
ids_all = torch.from_numpy(data["train_ids"][pick]).to(dev)
tg_all = torch.from_numpy(data["targets"][data["train_ids"][pick]]).to(dev)
eng = gs.engine.FusedMeanTrainStep(model, store, gs.ProblemLosses.classification, ids_all[0], tg_all[0], capture=False)
eng.load_epoch(ids_all, tg_all)
for _ in range(4):
    eng.step_queue()
torch.cuda.synchronize()
fronts = []
for k in range(8):
    buf = torch.zeros_like(eng.ids_q[0]); buf[:B] = ids_all[k]
    eng.queue, q = None, eng.queue
    eng._stage_sample(0, ids=buf)
    eng.queue = q
    fronts.append(buf)
torch.cuda.synchronize()
i = [0]


def gather_all():
    i[0] += 1
    eng._stage_gather(0, ids=fronts[i[0] % 8], skip_rows=0)


def chain():
    eng._tail_gather = None
    eng._stage_compute(0)                       # K5, seed level (no gather role), K5b, finalise
    eng._stage_opt()                            # Adam as its own launch


def record(fn):
    with nat.CommandList.record() as cl:
        fn()
    return cl


# command lists: one C call replays a whole chain (no Python between its launches)
cl_g = [record(lambda k=k: eng._stage_gather(0, ids=fronts[k], skip_rows=0)) for k in range(8)]
cl_c = record(chain)
print("chain list: %d launches" % len(cl_c))


def sp(stream):
    import ctypes
    return ctypes.c_void_p(stream.cuda_stream)


def time_list(stream, lists, reps=40):
    for k in range(3):
        lists[k % len(lists)].replay(sp(stream))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(reps):
        lists[k % len(lists)].replay(sp(stream))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def ext(bits):
    return torch.cuda.ExternalStream(nat.masked_stream(bits))


full, full2 = torch.cuda.Stream(), torch.cuda.Stream()
print("full chip: gather %.1f us, chain %.1f us" % (time_list(full, cl_g), time_list(full, [cl_c])))


def both(sa, sb, reps=40):
    for k in range(3):
        cl_g[k % 8].replay(sp(sa)); cl_c.replay(sp(sb))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(reps):
        cl_g[k % 8].replay(sp(sa))
        cl_c.replay(sp(sb))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


print("two unmasked streams side by side: %.1f us/step" % both(full, full2))
for n in (64, 80, 96, 112, 128, 160):
    a, b = list(range(n)), list(range(n, 256))
    sa, sb = ext(a), ext(b)
    print("gather on %3d CUs: %6.1f us alone | chain on %3d CUs: %6.1f us alone | side by side %6.1f us/step"
          % (n, time_list(sa, cl_g), 256 - n, time_list(sb, [cl_c]), both(sa, sb)))
