"""Can the NEXT batch's level-0 gathers (HBM-bound, stream B) run under THIS batch's compute chain
(K5, seed level, K5b, finalise: latency-bound, stream A)?  Timing experiment on the real stages of
the fused engine (two buffer sets), submission order and stream priorities varied."""
import importlib, sys, time
sys.path.insert(0, '.')
import torch, numpy as np, bench
gs = importlib.import_module('pytorch-graphsage_amd')
dev = torch.device('cuda'); gs.ops.warmup(dev); gs.ops.set_compute_dtype('bf16')
data = bench.synthetic_reddit(seed=0); store = data['feats'](dev, 'bf16'); tr = data['train_ids']
model = bench.build_model(gs, data['adj'], aggregator='mean', rng='philox').to(dev)
model.train_sampler.csr(dev)
pick = np.random.RandomState(1).randint(0, len(tr), size=(2, 512))
ids = torch.from_numpy(tr[pick]).to(dev); tg = torch.from_numpy(data['targets'][tr[pick]]).to(dev).view(2, 512, 1)
eng = gs.engine.FusedMeanTrainStep(model, store, gs.ProblemLosses.classification, ids[0], tg[0], capture=False, pipelined=True)
for s in (0, 1):
    eng._load(s, ids[s], tg[s]); eng._stage_sample_gather(s)
torch.cuda.synchronize()


def timed(fn, reps=60):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


print("compute chain alone   %.1f us" % timed(lambda: eng._stage_compute(0)))
print("gathers alone         %.1f us" % timed(lambda: eng._stage_gather(1)))
def seq():
    eng._stage_compute(0); eng._stage_gather(1)
print("sequential, 1 stream  %.1f us" % timed(seq))
for name, A, B in (("default/default", torch.cuda.Stream(), torch.cuda.Stream()),
                   ("chain high / gather low", torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0))):
    def gather_first():
        with torch.cuda.stream(B): eng._stage_gather(1)
        with torch.cuda.stream(A): eng._stage_compute(0)
    def chain_first():
        with torch.cuda.stream(A): eng._stage_compute(0)
        with torch.cuda.stream(B): eng._stage_gather(1)
    print("%-26s gather submitted first %.1f us | chain submitted first %.1f us" % (name, timed(gather_first), timed(chain_first)))
