"""Micro-benchmark of K4 / K4' with the attention MLP inside (csrc/gsage_attn_fused.hip) at Reddit's last hop
(12 800 parents x 10 children x 602 bf16 columns from a 232 966-row table) or Pokec's (153 600 x 64): phase stops
and geometry overrides (GSAGE_AF_WAVES / GSAGE_AF_PER_CU) swept in one process.
    python tools/afbench.py [reddit|pokec] [sweep]"""
import importlib, os, sys
sys.path.insert(0, '.')
import torch
gs = importlib.import_module('pytorch-graphsage_amd')
ops, nat = gs.ops, gs._native
dev = torch.device('cuda')
ops.warmup(dev)
L = nat.lib()
BF = torch.bfloat16
shape = sys.argv[1] if len(sys.argv) > 1 else 'reddit'
if shape == 'reddit':
    N, D, ld, M, n = 232966, 602, 640, 12800, 10
else:
    N, D, ld, M, n = 164352, 64, 64, 10240, 15
torch.manual_seed(0)
table = torch.zeros(N, ld, dtype=BF, device=dev); table[:, :D] = torch.randn(N, D, device=dev).to(BF)
NF = 8
ids = [torch.randint(0, N, (M * n,), device=dev) for _ in range(NF)]
w0 = torch.zeros(32, ld, dtype=BF, device=dev); w0[:, :D] = (torch.randn(32, D, device=dev) / D ** 0.5).to(BF)
w2 = torch.zeros(32, 64, dtype=BF, device=dev); w2[:, :32] = (torch.randn(32, 32, device=dev) / 6).to(BF)
w2t = torch.zeros(32, 64, dtype=BF, device=dev); w2t[:, :32] = w2[:, :32].t()
xa = torch.randn(M, 32, device=dev); gout = torch.randn(M, ld, device=dev)
rows = M * n
hid = torch.zeros(rows, 64, dtype=BF, device=dev); a = torch.zeros(rows, 32, device=dev); ws = torch.zeros(rows, device=dev)
agg = torch.zeros(M, ld, device=dev); aggc = torch.zeros(M, ld, dtype=BF, device=dev)
da = torch.zeros(rows, 64, dtype=BF, device=dev); dhid = torch.zeros(rows, 64, dtype=BF, device=dev); dxa = torch.zeros(M, 32, device=dev)
st = ops._stream()


def fwd(k):
    nat.check(L.gsage_attn_fused_fwd(table.data_ptr(), nat.BF16, ld, ids[k % NF].data_ptr(), 0, w0.data_ptr(), ld, w2.data_ptr(), 64,
                                     xa.data_ptr(), 32, M, n, D, hid.data_ptr(), 64, a.data_ptr(), 32, ws.data_ptr(),
                                     aggc.data_ptr(), ld, st), "fwd")


def bwd(k):
    nat.check(L.gsage_attn_fused_bwd(table.data_ptr(), nat.BF16, ld, ids[k % NF].data_ptr(), 0, w2t.data_ptr(), 64, gout.data_ptr(), ld,
                                     ws.data_ptr(), a.data_ptr(), 32, xa.data_ptr(), 32, hid.data_ptr(), 64, M, n, D,
                                     da.data_ptr(), 64, dhid.data_ptr(), 64, dxa.data_ptr(), 32, st), "bwd")


def old_fwd(k):
    nat.check(L.gsage_attn_aggregate_lp(a.data_ptr(), 32, xa.data_ptr(), 32, table.data_ptr(), nat.BF16, ld, ids[k % NF].data_ptr(), M,
                                        n, 32, D, agg.data_ptr(), ld, ws.data_ptr(), aggc.data_ptr(), ld, st), "k4")


def timeit(fn, reps=24):
    for k in range(3): fn(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(reps): fn(k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


alg = rows * D * 2
print("%s: %d parents x %d children x %d columns = %.1f MB" % (shape, M, n, D, alg / 1e6))
print("separate K4 (weighted sum only): %.1f us" % timeit(old_fwd))
tf, tb = timeit(fwd), timeit(bwd)
print("fused  fwd %.1f us (%.2f TB/s)   bwd %.1f us (%.2f TB/s)" % (tf, alg / tf / 1e6, tb, alg / tb / 1e6))
if 'sweep' in sys.argv:
    for nw, pc in ((8, 1), (6, 1), (5, 1), (4, 1), (3, 1), (4, 2), (3, 2), (2, 2), (2, 4), (1, 8)):
        os.environ["GSAGE_AF_WAVES"], os.environ["GSAGE_AF_PER_CU"] = str(nw), str(pc)
        try:
            print("waves=%d per_cu=%d: fwd %.1f  bwd %.1f us" % (nw, pc, timeit(fwd), timeit(bwd)))
        except Exception as e:
            print("waves=%d per_cu=%d: %r" % (nw, pc, e))
