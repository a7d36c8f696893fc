import importlib, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
gs = importlib.import_module('pytorch-graphsage_amd')
dev = torch.device('cuda'); gs.ops.warmup(dev)
data = bench.synthetic_reddit(seed=0)
store = data['feats'](dev, 'bf16')
for pipelined in (False, True):
    model = bench.build_model(gs, data['adj']).to(dev)
    model.train_sampler.csr(dev)
    B = 512; K = 200
    rng = np.random.RandomState(0)
    ids = torch.from_numpy(data['train_ids'][rng.randint(0, len(data['train_ids']), size=(K + 20, B))]).to(dev)
    tg = torch.from_numpy(data['targets'][data['train_ids'][rng.randint(0, 1000, size=(K + 20, B))]]).to(dev)
    eng = gs.engine.FusedMeanTrainStep(model, store, gs.ProblemLosses.classification, ids[0], tg[0], pipelined=pipelined)
    for k in range(20): eng(ids[k], tg[k])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(20, 20 + K): eng(ids[k], tg[k])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('pipelined=%s host enqueue %.1f us/step, total %.1f us/step' % (pipelined, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
    if pipelined: eng.flush()
    torch.cuda.synchronize()
