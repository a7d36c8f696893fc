"""Launch-path micro-benchmarks on the GPU box: host cost of a ctypes launch, dependent-kernel boundary,
hipGraph replay floor, two-stream concurrency of trivial kernels (numbers quoted in DESIGN.md)."""
import importlib, sys, time
sys.path.insert(0, '.')
import torch
gs = importlib.import_module('pytorch-graphsage_amd')
nat = gs._native; L = nat.lib(); ops = gs.ops
dev = torch.device('cuda')
ctr = torch.zeros(1, dtype=torch.int64, device=dev)
def k(): L.gsage_counter_add(ctr.data_ptr(), 1, ops._stream())
def ev(fn, reps):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
for _ in range(10): k()
N = 400
print('eager trivial kernel: %.2f us' % ev(lambda: [k() for _ in range(N)], N))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(N): k()
g.replay()
print('graph trivial kernel: %.2f us' % ev(lambda: g.replay(), N))
x = torch.zeros(1024, device=dev)
print('eager torch add_: %.2f us' % ev(lambda: [x.add_(1) for _ in range(N)], N))
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    for _ in range(N): x.add_(1)
print('graph torch add_: %.2f us' % ev(lambda: g2.replay(), N))
# two parallel branches inside one graph
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
c2 = torch.zeros(1, dtype=torch.int64, device=dev)
g3 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g3):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        for _ in range(N // 2): L.gsage_counter_add(ctr.data_ptr(), 1, ops._stream())
    with torch.cuda.stream(s2):
        for _ in range(N // 2): L.gsage_counter_add(c2.data_ptr(), 1, ops._stream())
    cur.wait_stream(s1); cur.wait_stream(s2)
g3.replay()
print('graph 2 branches trivial kernel: %.2f us per kernel' % ev(lambda: g3.replay(), N))
import os
print('env', {k: v for k, v in os.environ.items() if 'HIP' in k or 'HSA' in k or 'ROC' in k})
