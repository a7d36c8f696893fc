"""How much of a gradient exchange does the data-parallel step order hide?  (run on the GPU box)

One GPU cannot run a real multi-rank RCCL all-reduce, so the exchange is replaced by a kernel of a
chosen duration on a separate stream (exactly what torch's NCCL work handle does: the collective
runs on its own stream, `wait()` makes the compute stream wait for it).  Prints ms/step with the
engine's order (sample/gather of batch i+1 issued before the wait) and with the naive order
(wait immediately)."""
import importlib, os, sys, time
sys.path.insert(0, '.')
os.environ.update({"GSAGE_FORCE_DDP": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29541", "RANK": "0",
                   "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
import torch
import bench
gs = importlib.import_module('pytorch-graphsage_amd')
dev = torch.device('cuda')
gs.ops.warmup(dev)
ddp = gs.dist.init_from_env(cuda=True)
data = bench.synthetic_reddit(seed=0)
B = 512


def _calibrate():
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1000000); torch.cuda.synchronize()
    a.record(); torch.cuda._sleep(10000000); b.record(); torch.cuda.synchronize()
    return 10000000 / (a.elapsed_time(b) * 1e-3)


clock_hz = _calibrate()


class FakeWork(object):
    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


store = data["feats"](dev, "bf16")
gs.ops.set_compute_dtype("bf16")
tr = data["train_ids"]


def run(fake_us, naive, steps=300):
    model = bench.build_model(gs, data["adj"], aggregator="mean", rng="philox").to(dev)
    gs.dist.attach(model, ddp, seed=123)
    model.train_sampler.csr(dev)
    loss_fn = gs.ProblemLosses.classification
    import numpy as np
    pick = np.random.RandomState(1).randint(0, len(tr), size=(64, B))
    ids = torch.from_numpy(tr[pick]).to(dev)
    tg = torch.from_numpy(data["targets"][tr[pick]]).to(dev)
    eng = gs.engine.FusedMeanTrainStep(model, store, loss_fn, ids[0], tg[0], ddp=ddp)
    eng.load_epoch(ids, tg)
    side = torch.cuda.Stream()

    def fake(async_op=False):
        ev = torch.cuda.Event()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            if fake_us > 0:
                torch.cuda._sleep(int(fake_us * 1e-6 * clock_hz))
            ev.record(side)
        w = FakeWork(ev)
        if naive or not async_op:
            w.wait()
        return w
    eng._all_reduce = fake
    for _ in range(20):
        eng.step_queue()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.step_queue()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for us in (0, 30, 60, 100):
    print("exchange %3d us: overlapped order %.4f ms/step, naive order %.4f ms/step" % (us, run(us, False), run(us, True)))
ddp.close()
