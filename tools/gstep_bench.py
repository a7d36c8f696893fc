"""Where the in-step gather launch's time goes: the queue-mode launch of engine.FusedMeanTrainStep
(k_gather_multi_adam: gathers of the next batch | Adam | K1) and its parts, each timed as 40 launches in one
hipGraph over 8 different frontiers (1.1 GB of distinct rows: nothing is served from the Infinity Cache).
usage: python tools/gstep_bench.py"""
import importlib, sys
sys.path.insert(0, ".")
import numpy as np, torch
import bench
gs = importlib.import_module("pytorch-graphsage_amd")
ops = gs.ops
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
eng = gs.engine.FusedMeanTrainStep(model, store, gs.ProblemLosses.classification, ids_all[0], tg_all[0], capture="cmdlist")
eng.load_epoch(ids_all, tg_all)
for _ in range(4):
    eng.step_queue()
torch.cuda.synchronize()
fronts = []
for k in range(8):
    buf = torch.zeros_like(eng.ids_q[0])
    buf[:B] = ids_all[k]
    eng.queue, q = None, eng.queue                      # sample from buf's own seeds, not the queue
    eng._stage_sample(0, ids=buf)
    eng.queue = q
    fronts.append(buf)
torch.cuda.synchronize()


def timeit(fn, reps=40):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * reps) * 1e3


i = [0]
L, st, T = eng.L, store, eng._tail_rows
row_b = st.dim * 2


def front(with_adam, with_hops, skip):
    i[0] += 1
    ids = fronts[i[0] % 8]
    hops = eng._hops_desc(eng.ids_q[0], True) if with_hops else None
    eng._stage_gather(0, with_adam=with_adam, ids=ids, hops=hops, skip_rows=skip)


def only(which, skip):
    i[0] += 1
    ids = fronts[i[0] % 8]
    xa, R = eng.xa0_set[0], eng.rows[0]
    segs = {"h2": (st.data, ids[eng.off[2] + skip * 10:eng.off[3]], xa[1][eng.off[1] + skip:eng.off[2]], eng.size[1] - skip, 10),
            "h1": (st.data, ids[eng.off[1]:eng.off[2]], xa[1][:eng.off[1]], eng.size[0], 25),
            "x": (st.data, ids[:R], xa[0], R, 1)}
    ops.gather_mean_multi([segs[w] for w in which], st.ld, st.dim, st.ld)


rows_all = eng.off[3]
print("tail_rows (hop-2 means done by the seed-level launch):", T)
for name, fn, rows in [
        ("in-step launch: gather + Adam + K1", lambda: front(True, True, T), rows_all - T * 10),
        ("gather + K1", lambda: front(False, True, T), rows_all - T * 10),
        ("gather only (multi: h2' h1 x)", lambda: only(("h2", "h1", "x"), T), rows_all - T * 10),
        ("gather only, all of hop 2", lambda: only(("h2", "h1", "x"), 0), rows_all),
        ("h2' + h1 (no row copies)", lambda: only(("h2", "h1"), T), rows_all - T * 10 - eng.rows[0]),
        ("h2' alone", lambda: only(("h2",), T), (eng.size[1] - T) * 10),
        ("h2 alone (all 12 800 means)", lambda: only(("h2",), 0), eng.size[1] * 10),
        ("h1 alone", lambda: only(("h1",), 0), eng.size[0] * 25),
        ("x row copies alone", lambda: only(("x",), 0), eng.rows[0])]:
    t = timeit(fn)
    print("%-40s %6.1f us  %6.1f MB alg read  %.2f TB/s" % (name, t, rows * row_b / 1e6, rows * row_b / t / 1e6))
