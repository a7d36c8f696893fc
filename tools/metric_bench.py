"""GPU time of the per-batch metric launch(es) at the headline batch (512 x 41), alone: 40 launches per hipGraph."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
gs = importlib.import_module('pytorch-graphsage_amd')
ops, nat = gs.ops, gs._native
dev = torch.device('cuda'); ops.warmup(dev); L = nat.lib()

def timeit(fn, reps=40, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * reps) * 1e3

for B, C in ((512, 41), (512, 7), (64, 41), (2048, 41)):
    logits = torch.randn(B, C, device=dev)
    y = torch.randint(0, C, (B,), device=dev)
    counts = torch.zeros(3 * C + 1, dtype=torch.int32, device=dev)
    out = torch.zeros(3, dtype=torch.float64, device=dev)
    def f():
        nat.check(L.gsage_metric_f1(logits.data_ptr(), C, y.data_ptr(), 0, 0, 0, B, C, counts.data_ptr(), out.data_ptr(), ops._stream()))
    print("metric_f1 %d x %d: %.1f us per call" % (B, C, timeit(f)))
a, b = torch.randn(512, device=dev), torch.randn(512, device=dev)
o = torch.zeros(1, dtype=torch.float64, device=dev)
print("metric_mae 512: %.1f us" % timeit(lambda: nat.check(L.gsage_metric_mae(a.data_ptr(), b.data_ptr(), 512, o.data_ptr(), ops._stream()))))
x = torch.zeros(1, dtype=torch.int64, device=dev)
print("counter_add (an empty-ish launch): %.1f us" % timeit(lambda: nat.check(L.gsage_counter_add(x.data_ptr(), 1, ops._stream()))))
