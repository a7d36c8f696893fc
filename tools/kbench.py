"""Per-kernel micro-benchmarks at the Reddit config-2 shapes (run on the GPU box)."""
import importlib, sys, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
gs = importlib.import_module('pytorch-graphsage_amd')
ops, nat = gs.ops, gs._native
dev = torch.device('cuda')
ops.warmup(dev)
L = nat.lib()

def timeit(fn, reps=40, warm=3):
    """GPU time per call: `reps` launches captured in one hipGraph (no host launch overhead)."""
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

B, f1, f2, D, ld, h = 512, 25, 10, 602, 640, 128
R0 = B * (1 + f1)
N = 232966
torch.manual_seed(0)
table = torch.zeros(N, ld, dtype=torch.bfloat16, device=dev); table[:, :D] = torch.randn(N, D, device=dev).bfloat16()
which = sys.argv[1:] or ['wgrad', 'linear', 'head', 'gather']

if 'wgrad' in which:
    import os
    dC = torch.randn(R0, 2 * h, device=dev).bfloat16()
    XA = torch.randn(2, R0, ld, device=dev).bfloat16()
    for rps in (272, 416, 560, 1120):
        S = (R0 + rps - 1) // rps
        slabs = torch.empty(S, 2 * h, 604, device=dev)
        def f():
            nat.check(L.gsage_wgrad(dC.data_ptr(), nat.BF16, 2 * h, XA.data_ptr(), ld, R0 * ld, R0, 2 * h, D, h, rps,
                                    slabs.data_ptr(), 604, None, h * D, ops._stream()))
        print('wgrad L0 rps=%d S=%d (%d workgroups): %.1f us' % (rps, S, S * 10, timeit(f)))

if 'linear' in which:
    XA = torch.randn(2, R0, ld, device=dev).bfloat16()
    H = torch.empty(R0, 2 * h, dtype=torch.bfloat16, device=dev)
    for ldw in (608, 640):
        W2 = torch.randn(2, h, ldw, device=dev).bfloat16()
        for M in (R0, R0 // 4, 512):
            def f():
                ops._linear_launch(XA.data_ptr(), ld, None, 0, W2.data_ptr(), ldw, None, H.data_ptr(), 2 * h, M, h, D,
                                   1, 2, R0 * ld, h * ldw, h, nat.BF16, nat.BF16)
            t = timeit(f)
            print('linear L0 grouped ldw=%d (%s) M=%d: %.1f us  (%.0f TF/s)' % (ldw, 'dma' if ldw == 640 else 'reg', M, t, 2 * 2 * M * 640 * h / t / 1e6))

if 'linK' in which:
    XA = torch.randn(2, R0, ld, device=dev).bfloat16()
    H = torch.empty(R0, 2 * h, dtype=torch.bfloat16, device=dev)
    W2 = torch.randn(2, h, 640, device=dev).bfloat16()
    for M in (512, R0):
        for K in (64, 128, 256, 384, 640):
            def f():
                ops._linear_launch(XA.data_ptr(), ld, None, 0, W2.data_ptr(), 640, None, H.data_ptr(), 2 * h, M, h, K,
                                   1, 2, R0 * ld, h * 640, h, nat.BF16, nat.BF16)
            print('linear dma M=%d K=%d: %.1f us' % (M, K, timeit(f)))

if 'head' in which:
    E = torch.randn(B, 2 * h, device=dev); W = torch.randn(41, 2 * h, device=dev) * 0.1; b = torch.zeros(41, device=dev)
    tg = torch.randint(0, 41, (B,), device=dev)
    preds = torch.empty(B, 41, device=dev); dE = torch.empty(B, 2 * h, dtype=torch.bfloat16, device=dev)
    dW = torch.empty(41, 2 * h, device=dev); db = torch.empty(41, device=dev); loss = torch.empty(1, device=dev)
    scr = torch.empty(L.gsage_head_ce_scratch(B, 41, 2 * h), device=dev)
    def f():
        nat.check(L.gsage_head_ce(E.data_ptr(), 2 * h, W.data_ptr(), b.data_ptr(), tg.data_ptr(), B, 41, 2 * h,
                                  preds.data_ptr(), dE.data_ptr(), nat.BF16, 2 * h, dW.data_ptr(), db.data_ptr(),
                                  loss.data_ptr(), scr.data_ptr(), None, 0, ops._stream()))
    print('head (2 launches): %.1f us' % timeit(f))

if 'gather' in which:
    store = gs.FeatureStore(table, D)
    ids = [torch.randint(1, N, (B * f1 * f2,), device=dev) for _ in range(8)]
    i = [0]
    def f():
        i[0] += 1
        ops.gather_mean(store, ids[i[0] % 8], B * f1, f2, out_dtype=torch.bfloat16, out_ld=ld)
    t = timeit(f)
    print('gather hop2: %.1f us (%.2f TB/s alg)' % (t, B * f1 * f2 * D * 2 / t / 1e6))

if 'sample' in which:
    import bench
    data = bench.synthetic_reddit(seed=0)
    csr = gs.DeviceCSR.from_scipy(data['adj'], dev)
    ids0 = torch.randint(1, N, (B,), device=dev)
    o1 = torch.empty(B * f1, dtype=torch.int64, device=dev); o2 = torch.empty(B * f1 * f2, dtype=torch.int64, device=dev)
    def f():
        ops.sample_csr(csr, ids0, f1, philox={"seed": 1, "call_base": 0}, out=o1)
        ops.sample_csr(csr, o1, f2, philox={"seed": 1, "call_base": 1}, out=o2)
    print('sample 2 hops: %.1f us' % timeit(f))

if 'tail' in which:
  for B in (128, 256, 512, 1024, 2048):
      n, C = 25, 41
      H = torch.randn(B * (1 + n), 256, device=dev).bfloat16()
      w2 = (torch.randn(2, 128, 256, device=dev) * 0.05).bfloat16(); w2t = w2.transpose(1, 2).contiguous()
      Wfc = torch.randn(C, 256, device=dev) * 0.05; bfc = torch.zeros(C, device=dev)
      tg = torch.randint(0, C, (B,), device=dev)
      agg = torch.empty(B, 256, device=dev, dtype=torch.bfloat16); dE = torch.empty_like(agg)
      preds = torch.empty(B, C, device=dev); dH = torch.empty_like(H)
      part = torch.empty(L.gsage_mean_tail_ce_scratch(B, C), device=dev)
      def f():
          nat.check(L.gsage_mean_tail_ce(H.data_ptr(), B, n, w2.data_ptr(), 256, w2t.data_ptr(), 128, Wfc.data_ptr(),
                                         bfc.data_ptr(), C, tg.data_ptr(), None, 0, agg.data_ptr(), dE.data_ptr(),
                                         preds.data_ptr(), dH.data_ptr(), part.data_ptr(), None, nat.BF16, ops._stream()))
      print('tail (B=%d n=%d): %.1f us' % (B, n, timeit(f)))

if 'tailm' in which:
  import os
  for B in (512,):
      n, C = 25, 41
      H = torch.randn(B * (1 + n), 256, device=dev).bfloat16()
      w2 = (torch.randn(2, 128, 256, device=dev) * 0.05).bfloat16(); w2t = w2.transpose(1, 2).contiguous()
      Wfc = torch.randn(C, 256, device=dev) * 0.05; bfc = torch.zeros(C, device=dev)
      tg = torch.randint(0, C, (B,), device=dev)
      agg = torch.empty(B, 256, device=dev, dtype=torch.bfloat16); dE = torch.empty_like(agg)
      preds = torch.empty(B, C, device=dev); dH = torch.empty_like(H)
      part = torch.empty(L.gsage_mean_tail_mfma_scratch(B, C), device=dev)
      def f():
          nat.check(L.gsage_mean_tail_mfma(H.data_ptr(), B, n, w2.data_ptr(), 256, w2t.data_ptr(), 128, Wfc.data_ptr(),
                                           bfc.data_ptr(), C, tg.data_ptr(), None, 0, agg.data_ptr(), dE.data_ptr(),
                                           preds.data_ptr(), dH.data_ptr(), part.data_ptr(), None, ops._stream()))
      for stop in (1, 2, 3, 4, 5, 0):
          os.environ["GSAGE_TAIL_STOP"] = str(stop)
          print('tailm B=%d stop after phase %d: %.1f us' % (B, stop, timeit(f)))
      os.environ["GSAGE_TAIL_STOP"] = "0"

if 'gmulti' in which:
    # the level-0 gathers of one step (x rows of 13 312 nodes | 512 means of 25 | 12 800 means of 10)
    # in one launch, for several segment orders
    store = gs.FeatureStore(table, D)
    tot = B * (1 + f1 + f1 * f2)
    idsets = [torch.randint(1, N, (tot,), device=dev) for _ in range(8)]
    xa = torch.zeros(2, R0, ld, dtype=torch.bfloat16, device=dev)
    i = [0]
    def segs(ids):
        return {"x": (table, ids[:R0], xa[0], R0, 1),
                "h1": (table, ids[B:R0], xa[1][:B], B, f1),
                "h2": (table, ids[R0:], xa[1][B:], B * f1, f2)}
    for order in (("x", "h1", "h2"), ("h2", "h1", "x"), ("h2", "x", "h1"), ("h1", "h2", "x"), ("h2",), ("x",), ("h1",)):
        def f():
            i[0] += 1
            sg = segs(idsets[i[0] % 8])
            ops.gather_mean_multi([sg[k] for k in order], ld, D, ld)
        t = timeit(f)
        rows = sum({"x": R0, "h1": B * f1, "h2": B * f1 * f2}[k] for k in order)
        print('gmulti %-12s: %.1f us (%.2f TB/s read)' % ('+'.join(order), t, rows * D * 2 / t / 1e6))

if 'linp' in which:
    # K5 at the layer-0 shape: LDS-DMA kernel against the packed-weight kernel
    for M in (R0 * 8, R0, R0 // 4, 512):
        XA = torch.randn(2, M, ld, device=dev).bfloat16()
        XA[:, :, D:] = 0
        H = torch.empty(M, 2 * h, dtype=torch.bfloat16, device=dev)
        W2 = torch.zeros(2, h, 640, device=dev, dtype=torch.bfloat16)
        W2[:, :, :D] = torch.randn(2, h, D, device=dev).bfloat16()
        Wp = ops.pack_weight(W2, K=D)
        def f0():
            ops._linear_launch(XA.data_ptr(), ld, None, 0, W2.data_ptr(), 640, None, H.data_ptr(), 2 * h, M, h, D,
                               1, 2, M * ld, h * 640, h, nat.BF16, nat.BF16)
        def f1():
            ops._linear_packed_launch(XA.data_ptr(), ld, None, 0, Wp.data_ptr(), None, H.data_ptr(), 2 * h, M, h, D,
                                      1, 2, M * ld, h, nat.BF16)
        f0(); ref = H.clone(); H.zero_(); f1()
        err = (H.float() - ref.float()).abs().max().item()
        t0, t1 = timeit(f0), timeit(f1)
        fl = 2 * 2 * M * 640 * h / 1e6
        print('linp M=%6d: dma %.1f us (%.0f TF/s)   packed %.1f us (%.0f TF/s)   max diff %.3g' % (M, t0, fl / t0, t1, fl / t1, err))

if 'linpk' in which:
    # K sweep: fixed cost vs cost per 64-wide k-tile, both K5 kernels
    for M in (512, R0):
        for K in (64, 256, 640, 1280, 2560):
            XA = torch.randn(2, M, K, device=dev).bfloat16()
            H = torch.empty(M, 2 * h, dtype=torch.bfloat16, device=dev)
            W2 = torch.randn(2, h, K, device=dev).bfloat16()
            Wp = ops.pack_weight(W2)
            def f0():
                ops._linear_launch(XA.data_ptr(), K, None, 0, W2.data_ptr(), K, None, H.data_ptr(), 2 * h, M, h, K,
                                   1, 2, M * K, h * K, h, nat.BF16, nat.BF16)
            def f1():
                ops._linear_packed_launch(XA.data_ptr(), K, None, 0, Wp.data_ptr(), None, H.data_ptr(), 2 * h, M, h, K,
                                          1, 2, M * K, h, nat.BF16)
            print('linpk M=%6d K=%5d (%2d k-tiles): dma %.1f us   packed %.1f us' % (M, K, K // 64, timeit(f0), timeit(f1)))

if 'pool' in which:
    # K3 on the hop-2 / hop-1 frontier of the max-pool configuration: 64 x 128 tiles against the packed kernel
    Hm = 512
    Wm = torch.zeros(Hm, ld, dtype=torch.bfloat16, device=dev); Wm[:, :D] = (torch.randn(Hm, D, device=dev) / 25).bfloat16()
    bm = torch.randn(Hm, device=dev) * 0.1
    Wp = ops.pack_weight(Wm, K=D)
    for M, n in ((B * f1, f2), (B, f1)):
        rows = torch.zeros(M * n, ld, dtype=torch.bfloat16, device=dev); rows[:, :D] = torch.randn(M * n, D, device=dev).bfloat16()
        pooled = torch.empty(M, Hm, device=dev); pb = torch.empty(M, Hm, dtype=torch.bfloat16, device=dev)
        arg = torch.empty(M, Hm, dtype=torch.int32, device=dev)
        def f0():
            nat.check(L.gsage_pool_mlp(rows.data_ptr(), nat.BF16, ld, None, Wm.data_ptr(), ld, bm.data_ptr(), M, n, Hm, D,
                                       nat.POOL_MAX, pooled.data_ptr(), Hm, arg.data_ptr(), pb.data_ptr(), Hm, None, ops._stream()))
        def f1():
            nat.check(L.gsage_pool_mlp_packed(rows.data_ptr(), ld, None, Wp.data_ptr(), bm.data_ptr(), M, n, Hm, D,
                                              nat.POOL_MAX, pooled.data_ptr(), Hm, arg.data_ptr(), pb.data_ptr(), Hm, None, ops._stream()))
        f0(); ref = pooled.clone(); pooled.zero_(); f1()
        err = (pooled - ref).abs().max().item()
        t0, t1 = timeit(f0, reps=10), timeit(f1, reps=10)
        fl = 2.0 * M * n * D * Hm / 1e6
        print('pool M=%6d n=%2d: tiles64x128 %.1f us (%.0f TF/s)   packed %.1f us (%.0f TF/s)   max diff %.3g' % (M, n, t0, fl / t0, t1, fl / t1, err))

if 'wgradpool' in which:
    # the max-pool MLP's weight gradient alone: dW[512 x 602] = ghc[128000 x 512]^T x rows[128000 x 602]
    M, Nt, K = 128000, 512, 602
    dC = torch.randn(M, Nt, device=dev).bfloat16()
    g = torch.Generator(device='cpu'); g.manual_seed(1)
    lists = {
        'random rows (as in the step)': torch.randint(0, N, (M,), generator=g).to(dev),
        'consecutive rows': (torch.arange(M) % N).to(dev),
        '4096 distinct rows (A hot in L2)': torch.randint(0, 4096, (M,), generator=g).to(dev),
    }
    for target in (160, 240, 480):
        rps, S, ldk = ops.wgrad_plan(M, Nt, K, target)
        slabs = torch.empty(S, Nt, ldk, device=dev)
        for name, rows in lists.items():
            def f():
                ops.wgrad_multi([(dC, table, ld, 0, M, Nt, K, Nt, slabs, target, rows)])
            t = timeit(f, reps=10)
            print('wgrad pool target=%d S=%d (%d workgroups) %s: %.1f us (%.0f TF/s)' % (target, S, S * 20, name, t, 2 * M * Nt * 640 / t / 1e6))
        A = table[:M].contiguous() if M <= N else torch.randn(M, ld, device=dev).bfloat16()
        def f():
            ops.wgrad_multi([(dC, A, ld, 0, M, Nt, K, Nt, slabs, target)])
        t = timeit(f, reps=10)
        print('wgrad pool target=%d S=%d contiguous A, no row list: %.1f us' % (target, S, t))

if 'wgradcold' in which:
    # the same problem with its operands NOT resident in the Infinity Cache: a 1 GB fill between launches
    M, Nt, K = 128000, 512, 602
    dC = torch.randn(M, Nt, device=dev).bfloat16()
    g = torch.Generator(device='cpu'); g.manual_seed(1)
    rows = torch.randint(0, N, (M,), generator=g).to(dev)
    flush = torch.empty(256 << 20, dtype=torch.float32, device=dev)
    rps, S, ldk = ops.wgrad_plan(M, Nt, K, 240)
    slabs = torch.empty(S, Nt, ldk, device=dev)
    def fl():
        flush.fill_(1.0)
    def both():
        flush.fill_(1.0)
        ops.wgrad_multi([(dC, table, ld, 0, M, Nt, K, Nt, slabs, 240, rows)])
    t0, t1 = timeit(fl, reps=10), timeit(both, reps=10)
    print('wgrad pool cold: fill %.1f us, fill + wgrad %.1f us -> wgrad %.1f us' % (t0, t1, t1 - t0))

if 'gemmref' in which:
    # Reference point for K3 / K5b (round-3 verdict, item 3): the vendor's bf16 GEMM (torch.mm -> hipBLASLt) at the
    # max-pool step's two big contractions, warm (operands re-read from the Infinity Cache launch after launch) and
    # cold (a 1 GB fill between launches: operands come from HBM, as inside the step).  Bench tooling only -- never
    # a product path.  Writes gpurun_out/gemmref.json.
    import json
    Mbig, K, Nh = 128000, 640, 512
    out = {}
    A = torch.randn(Mbig, K, device=dev).bfloat16()
    Wt = torch.randn(K, Nh, device=dev).bfloat16()
    dC = torch.randn(Mbig, Nh, device=dev).bfloat16()
    fill = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev)
    def cold(fn, reps=12):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            fill.fill_(1.0)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        return float(np.median(ts))
    shapes = {
        "k3_fwd  [128000 x 640] x [640 x 512]": (lambda: torch.mm(A, Wt), 2.0 * Mbig * K * Nh),
        "k5b_mlp [512 x 128000] x [128000 x 640] (dC^T A)": (lambda: torch.mm(dC.t(), A), 2.0 * Mbig * K * Nh),
        "k5b_mlp, fp32 result (out_dtype)": (lambda: torch.mm(dC.t(), A, out_dtype=torch.float32), 2.0 * Mbig * K * Nh),
    }
    for name, (fn, flops) in shapes.items():
        try:
            tw = timeit(fn, reps=10)
            tc = cold(fn)
        except Exception as e:
            out[name] = {"error": repr(e)}
            continue
        out[name] = {"warm_us": tw, "cold_us": tc, "warm_tflops": flops / tw / 1e6, "cold_tflops": flops / tc / 1e6,
                     "frac_of_2500_warm": flops / tw / 1e6 / 2500.0, "frac_of_2500_cold": flops / tc / 1e6 / 2500.0}
        print('gemmref %s: warm %.1f us (%.0f TF/s), cold %.1f us (%.0f TF/s)' % (name, tw, flops / tw / 1e6, tc, flops / tc / 1e6))
    import os
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(out, open('gpurun_out/gemmref.json', 'w'), indent=1)
