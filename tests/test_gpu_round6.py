"""-m gpu, round 6: K4 / K4' with the attention MLP inside (csrc/gsage_attn_fused.hip) against the launches they
replace (att.0 GEMM + tanh, gsage_attn_mlp2_fwd, gsage_attn_aggregate_lp; gsage_attn_bwd, gsage_attn_mlp2_bwd) and against
the definition in plain torch with the engine's rounding points (reference nn_modules.py:293-296, 309-315)."""
import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu
gs = pkg()
ops = gs.ops
nat = gs._native
DEV = "cuda"
BF = torch.bfloat16


@pytest.fixture(autouse=True)
def _setup():
    ops.set_compute_dtype("bf16")
    ops.warmup(torch.device(DEV))
    yield
    ops.set_compute_dtype("bf16")


def _r64(v):
    return -(-v // 64) * 64


def _case(D, n, M, N, with_ids, seed=3):
    g = torch.Generator(device="cpu"); g.manual_seed(seed)
    ld = _r64(D)
    rows = M * n
    table = torch.zeros(N if with_ids else rows, ld, dtype=BF, device=DEV)
    table[:, :D] = (torch.randn(table.shape[0], D, generator=g) * 0.7).to(DEV).to(BF)
    ids = torch.randint(0, N, (rows,), generator=g).to(DEV) if with_ids else None
    w0 = torch.zeros(32, ld, dtype=BF, device=DEV)
    w0[:, :D] = (torch.randn(32, D, generator=g) / np.sqrt(D)).to(DEV).to(BF)
    w2 = torch.zeros(32, 64, dtype=BF, device=DEV)
    w2[:, :32] = (torch.randn(32, 32, generator=g) / np.sqrt(32)).to(DEV).to(BF)
    w2t = torch.zeros(32, 64, dtype=BF, device=DEV)
    w2t[:, :32] = w2[:, :32].t()
    xa = torch.randn(M, 32, generator=g).to(DEV)
    gout = torch.zeros(M, ld, device=DEV)
    gout[:, :D] = torch.randn(M, D, generator=g).to(DEV)
    return dict(D=D, n=n, M=M, ld=ld, table=table, ids=ids, w0=w0, w2=w2, w2t=w2t, xa=xa, gout=gout)


def _fused(c):
    L, st = nat.lib(), ops._stream()
    D, n, M, ld = c["D"], c["n"], c["M"], c["ld"]
    rows = M * n
    hid = torch.zeros(rows, 64, dtype=BF, device=DEV); a = torch.zeros(rows, 32, device=DEV)
    ws = torch.zeros(rows, device=DEV)
    aggc = torch.full((M, ld), 7.0, dtype=BF, device=DEV)
    idp = c["ids"].data_ptr() if c["ids"] is not None else None
    assert L.gsage_attn_fused_ok(nat.BF16, ld, D, n, 32) == 1
    nat.check(L.gsage_attn_fused_fwd(c["table"].data_ptr(), nat.BF16, ld, idp, 0, c["w0"].data_ptr(), ld, c["w2"].data_ptr(),
                                     64, c["xa"].data_ptr(), 32, M, n, D, hid.data_ptr(), 64, a.data_ptr(), 32, ws.data_ptr(),
                                     aggc.data_ptr(), ld, st), "fused_fwd")
    da = torch.zeros(rows, 64, dtype=BF, device=DEV); dhid = torch.zeros(rows, 64, dtype=BF, device=DEV)
    dxa = torch.zeros(M, 32, device=DEV)
    nat.check(L.gsage_attn_fused_bwd(c["table"].data_ptr(), nat.BF16, ld, idp, 0, c["w2t"].data_ptr(), 64,
                                     c["gout"].data_ptr(), ld, ws.data_ptr(), a.data_ptr(), 32, c["xa"].data_ptr(), 32,
                                     hid.data_ptr(), 64, M, n, D, da.data_ptr(), 64, dhid.data_ptr(), 64, dxa.data_ptr(), 32,
                                     st), "fused_bwd")
    torch.cuda.synchronize()
    return dict(hid=hid, a=a, ws=ws, aggc=aggc, da=da, dhid=dhid, dxa=dxa)


def _separate(c):
    """the launches engine.FusedAttnTrainStep issued before round 6, on the same operands"""
    L, st = nat.lib(), ops._stream()
    D, n, M, ld = c["D"], c["n"], c["M"], c["ld"]
    rows = M * n
    hid = torch.zeros(rows, 64, dtype=BF, device=DEV); a = torch.zeros(rows, 32, device=DEV)
    idp = c["ids"].data_ptr() if c["ids"] is not None else None
    ops._linear_launch(c["table"].data_ptr(), ld, idp, 0, c["w0"].data_ptr(), ld, None, hid.data_ptr(), 64, rows, 32, D,
                       nat.ACT_TANH, 1, 0, 0, 0, nat.BF16, nat.BF16)
    nat.check(L.gsage_attn_mlp2_fwd(hid.data_ptr(), nat.BF16, 64, c["w2"].data_ptr(), 64, a.data_ptr(), 32, rows, 32, st), "mlp2")
    ws = torch.zeros(rows, device=DEV); agg = torch.zeros(M, ld, device=DEV); aggc = torch.zeros(M, ld, dtype=BF, device=DEV)
    nat.check(L.gsage_attn_aggregate_lp(a.data_ptr(), 32, c["xa"].data_ptr(), 32, c["table"].data_ptr(), nat.BF16, ld, idp, M,
                                        n, 32, D, agg.data_ptr(), ld, ws.data_ptr(), aggc.data_ptr(), ld, st), "k4")
    dan = torch.zeros(rows, 32, device=DEV); dax = torch.zeros(M, 32, device=DEV)
    nat.check(L.gsage_attn_bwd(c["gout"].data_ptr(), ld, ws.data_ptr(), a.data_ptr(), 32, c["xa"].data_ptr(), 32,
                               c["table"].data_ptr(), nat.BF16, ld, idp, M, n, 32, D, dan.data_ptr(), 32, dax.data_ptr(), 32,
                               st), "k4'")
    zero = torch.zeros(rows, 32, device=DEV)
    da = torch.zeros(rows, 64, dtype=BF, device=DEV); dhid = torch.zeros(rows, 64, dtype=BF, device=DEV)
    nat.check(L.gsage_attn_mlp2_bwd(dan.data_ptr(), 32, zero.data_ptr(), 32, hid.data_ptr(), nat.BF16, 64, c["w2t"].data_ptr(),
                                    64, da.data_ptr(), 64, dhid.data_ptr(), 64, rows, 32, st), "mlp2'")
    torch.cuda.synchronize()
    return dict(hid=hid, a=a, ws=ws, agg=agg, aggc=aggc, da=da, dhid=dhid, dxa=dax)


def _definition(c):
    """nn_modules.py:293-296, 309-315 and their autograd in fp64 with the engine's rounding points (hid, d a, d hid in
    bf16; operands bf16)"""
    D, n, M = c["D"], c["n"], c["M"]
    X = (c["table"][c["ids"]] if c["ids"] is not None else c["table"])[:, :D].double()
    W0, W2 = c["w0"][:, :D].double(), c["w2"][:, :32].double()
    rb = lambda t: t.to(torch.float32).to(BF).double()
    hid = rb(torch.tanh(X @ W0.t()))
    a = hid @ W2.t()
    xa = c["xa"].double()
    s = torch.bmm(a.view(M, n, 32), xa.unsqueeze(2)).squeeze(2)
    ws = torch.softmax(s, dim=1)
    agg = (X.view(M, n, D) * ws.unsqueeze(2)).sum(1)
    g = c["gout"][:, :D].double()
    dws = torch.bmm(X.view(M, n, D), g.unsqueeze(2)).squeeze(2)
    ds = ws * (dws - (dws * ws).sum(1, keepdim=True))
    dxa = (ds.unsqueeze(2) * a.view(M, n, 32)).sum(1)
    da = rb((ds.unsqueeze(2) * xa.unsqueeze(1)).reshape(M * n, 32))
    dhid = rb((da @ W2) * (1 - hid * hid))
    return dict(hid=hid, a=a, ws=ws.reshape(-1), agg=agg, da=da, dhid=dhid, dxa=dxa)


SHAPES = [(602, 10, 300, True), (602, 10, 37, False), (64, 15, 500, False), (64, 15, 129, True), (12, 3, 55, True),
          (40, 2, 64, True), (128, 16, 70, True), (256, 5, 33, False), (640, 7, 19, True), (100, 4, 21, True)]


@pytest.mark.parametrize("D,n,M,with_ids", SHAPES)
def test_fused_attention_hop_against_the_separate_launches_and_the_definition(D, n, M, with_ids):
    c = _case(D, n, M, 5000, with_ids)
    f, s, d = _fused(c), _separate(c), _definition(c)
    ld = c["ld"]
    # --- against the definition (fp64, the same rounding points)
    tol = dict(rtol=2e-3, atol=2e-3)
    # hid is a bf16 value: one ulp (2^-8 relative) where the pre-activation rounds differently
    torch.testing.assert_close(f["hid"][:, :32].double(), d["hid"], rtol=0, atol=1.0 / 128)
    torch.testing.assert_close(f["a"].double(), d["a"], rtol=0, atol=2e-2)
    torch.testing.assert_close(f["ws"].double(), d["ws"], rtol=0, atol=2e-2)
    torch.testing.assert_close(f["aggc"][:, :D].double(), d["agg"], rtol=1.0 / 128, atol=3e-2)
    torch.testing.assert_close(f["dxa"].double(), d["dxa"], rtol=5e-2, atol=5e-2 * float(d["dxa"].abs().max()))
    # --- against the separate launches: the same operands through the same rounding points; sums in another order
    assert torch.equal(f["hid"][:, 32:], torch.zeros_like(f["hid"][:, 32:]))
    dh = (f["hid"][:, :32].float() - s["hid"][:, :32].float()).abs()
    assert float(dh.max()) <= 1.0 / 128 and float((dh > 0).float().mean()) < 0.02, (float(dh.max()), float((dh > 0).float().mean()))
    torch.testing.assert_close(f["a"], s["a"], rtol=0, atol=2e-2)
    same = (dh.max(dim=1).values == 0)                      # rows whose hidden layer came out bit-identical
    assert float(same.float().mean()) > 0.5
    torch.testing.assert_close(f["a"][same], s["a"][same], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(f["ws"], s["ws"], rtol=0, atol=2e-2)
    # (the operand copy of the aggregate: bf16 of sums taken in another order -- an ulp where they straddle a boundary)
    torch.testing.assert_close(f["aggc"][:, :D].float(), s["agg"][:, :D], rtol=1.0 / 128, atol=3e-2)
    da_ = (f["aggc"][:, :D].float() - s["aggc"][:, :D].float()).abs()
    assert float((da_ > 0).float().mean()) < 0.05
    # pad columns: zeros up to the 32-column step, untouched beyond it
    cols = -(-D // 32) * 32
    assert float(f["aggc"][:, D:cols].float().abs().max() if cols > D else 0.0) == 0.0
    if ld > cols:
        assert torch.all(f["aggc"][:, cols:].float() == 7.0)
    # backward: fed the fused forward's own ws / a / hid, the definition fed the exact ones -- compare in norm
    for k in ("da", "dhid"):
        x, y = f[k][:, :32].double(), d[k]
        assert float((x - y).norm() / (y.norm() + 1e-30)) < 3e-2, (k, float((x - y).norm() / y.norm()))
        x2 = s[k][:, :32].double()
        assert float((x - x2).norm() / (x2.norm() + 1e-30)) < 3e-2, (k, "separate")
    assert float((f["dxa"] - s["dxa"]).norm() / s["dxa"].norm()) < 3e-2


def test_fused_attention_backward_on_identical_inputs_equals_the_separate_launches():
    """K4' + the second layer's backward fed the SAME ws / a / hid: d a and d hid are products of the same factors
    rounded at the same points -- equal up to the order of the fp32 sums inside dws."""
    c = _case(602, 10, 200, 5000, True)
    L, st = nat.lib(), ops._stream()
    s = _separate(c)
    D, n, M, ld = c["D"], c["n"], c["M"], c["ld"]
    rows = M * n
    da = torch.zeros(rows, 64, dtype=BF, device=DEV); dhid = torch.zeros(rows, 64, dtype=BF, device=DEV)
    dxa = torch.zeros(M, 32, device=DEV)
    nat.check(L.gsage_attn_fused_bwd(c["table"].data_ptr(), nat.BF16, ld, c["ids"].data_ptr(), 0, c["w2t"].data_ptr(), 64,
                                     c["gout"].data_ptr(), ld, s["ws"].data_ptr(), s["a"].data_ptr(), 32, c["xa"].data_ptr(), 32,
                                     s["hid"].data_ptr(), 64, M, n, D, da.data_ptr(), 64, dhid.data_ptr(), 64, dxa.data_ptr(), 32,
                                     st), "fused_bwd")
    torch.cuda.synchronize()
    torch.testing.assert_close(dxa, s["dxa"], rtol=1e-4, atol=1e-4 * float(s["dxa"].abs().max()))
    for got, want in ((da, s["da"]), (dhid, s["dhid"])):
        diff = (got.float() - want.float()).abs()
        # a bf16 ulp where the fp32 value sat on a rounding boundary
        assert float((diff > 0).float().mean()) < 0.02
        assert float(diff.max()) <= float(want.float().abs().max()) / 64


def test_fused_attention_refuses_what_it_does_not_cover():
    L = nat.lib()
    assert L.gsage_attn_fused_ok(nat.F32, 640, 602, 10, 32) == 0          # fp32 rows: the separate launches
    assert L.gsage_attn_fused_ok(nat.BF16, 640, 602, 25, 32) == 0         # a fan-out beyond one 16-row tile
    assert L.gsage_attn_fused_ok(nat.BF16, 640, 602, 1, 32) == 0
    assert L.gsage_attn_fused_ok(nat.BF16, 1280, 1200, 10, 32) == 0       # rows beyond 640 columns
    assert L.gsage_attn_fused_ok(nat.BF16, 600, 600, 10, 32) == 0         # ld below the 32-column step
    assert L.gsage_attn_fused_ok(nat.BF16, 640, 602, 10, 64) == 0
    rc = L.gsage_attn_fused_fwd(None, nat.F32, 640, None, 0, None, 640, None, 64, None, 32, 4, 10, 602, None, 64, None, 32,
                                None, None, 640, None)
    assert rc == -1 and b"not covered" in L.gsage_last_error()


@pytest.mark.parametrize("capture", ["cmdlist", False])
def test_attention_engine_with_the_fused_hop_tracks_the_separate_launches(monkeypatch, capture):
    """engine.FusedAttnTrainStep over bf16 features: the last hop of both levels through the fused kernels (default)
    against GSAGE_ATTN_FUSED=0 -- the first step's predictions, gradient norm and clipped gradient within the bf16
    engines' bounds (later steps only loosely: Adam moves every weight by ~lr whatever its gradient's size, so entries
    whose tiny gradients differ in sign drift apart by lr per step in ANY two correct implementations)."""
    from test_gpu_engine import _model, _problem
    from util import close_fro
    outs = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("GSAGE_ATTN_FUSED", flag)
        adj, feats, rng = _problem(n=900, D=40, seed=4)
        store = gs.FeatureStore.from_array(feats, torch.device(DEV), dtype="bf16")
        B, C = 48, 5
        ids = torch.from_numpy(rng.randint(1, adj.shape[0], size=(6, B))).to(DEV)
        tg = torch.from_numpy(rng.randint(0, C, size=(6, B, 1))).to(DEV)
        m = _model(adj, feats.shape[1], C, (32, 16), (5, 3), agg="attention")
        eng = gs.engine.FusedAttnTrainStep(m, store, gs.ProblemLosses.classification, ids[0], tg[0], capture=capture)
        assert eng.fuse == [flag == "1"] * 2          # (fan-outs 5 and 3: the last hop of both levels)
        p0 = eng(ids[0], tg[0]).clone().float().cpu().numpy()
        torch.cuda.synchronize()
        g0, n0 = eng.flat_g.clone().cpu().numpy(), float(eng.gnorm.item())
        rest = torch.stack([eng(ids[k], tg[k]).clone() for k in range(1, 4)]).float().cpu().numpy()
        outs[flag] = (p0, g0, n0, rest)
    a, b = outs["0"], outs["1"]
    assert np.abs(a[0] - b[0]).max() < 3e-3 * max(1.0, np.abs(a[0]).max())
    assert abs(a[2] - b[2]) < 5e-3 * a[2]
    close_fro(b[1], a[1], "clipped gradient of the first step", 2e-2)
    assert np.abs(a[3] - b[3]).max() < 0.5 * max(1.0, np.abs(a[3]).max())


# ---- the node-embedding prep as row pipelines (csrc/gsage_prep_rows.hip) ---------------------------------------------
@pytest.mark.parametrize("M,n_seed", [(1000, 48), (37, 16), (515, 0), (64, 64)])
def test_prep_rows_forward_equals_gather_then_projection(M, n_seed):
    """gsage_prep_rows_fwd against the launches it replaces and the definition (nn_modules.py:145-151): eraw bit for
    bit (the same rounding of the same fp32 rows), the projection to an ulp of bf16 (sums in another order)."""
    g = torch.Generator(device="cpu"); g.manual_seed(1)
    N, E = 3000, 64
    table = torch.randn(N + 2, E, generator=g).to(DEV)
    ids = torch.randint(1, N, (M,), generator=g).to(DEV)
    W = (torch.randn(E, E, generator=g) / 8).to(DEV)
    Wc = W.to(BF).contiguous(); bias = torch.randn(E, generator=g).to(DEV)
    eraw = torch.zeros(M, E, dtype=BF, device=DEV); out = torch.full((M, 128), 3.0, dtype=BF, device=DEV)
    L = nat.lib()
    assert L.gsage_prep_rows_ok(nat.BF16, 64) == 1 and L.gsage_prep_rows_ok(nat.F32, 64) == 0 and L.gsage_prep_rows_ok(nat.BF16, 32) == 0
    nat.check(L.gsage_prep_rows_fwd(table.data_ptr(), E, ids.data_ptr(), n_seed, N + 1, Wc.data_ptr(), E, bias.data_ptr(), M, E,
                                    eraw.data_ptr(), E, out.data_ptr() + 2 * 32, 128, ops._stream()), "prep_rows_fwd")
    torch.cuda.synchronize()
    rows = ids.clone(); rows[:n_seed] = N + 1
    want_e = table[rows].to(BF)
    assert torch.equal(eraw, want_e)
    want = want_e.double() @ Wc.double().t() + bias.double()
    torch.testing.assert_close(out[:, 32:96].double(), want, rtol=1.0 / 128, atol=1e-2)
    assert torch.all(out[:, :32].float() == 3.0) and torch.all(out[:, 96:].float() == 3.0)


@pytest.mark.parametrize("with_dhid,with_ws,deraw_mode", [(True, True, False), (False, False, False), (True, True, True)])
def test_prep_rows_backward_equals_merge_sums_projection_and_scatter(with_dhid, with_ws, deraw_mode):
    """gsage_prep_rows_bwd against the definition in fp64 (nn_modules.py:307-317 w.r.t. the level-0 rows, then 145-151):
    the input gradient of the level-0 rows, its bf16 operand copy, the bias partials, and the table's gradient (atomics,
    duplicates and the seeds' shared spare row included) or the d embedding rows."""
    import ctypes
    g = torch.Generator(device="cpu"); g.manual_seed(2)
    B, f1, f2, E, N = 24, 5, 3, 64, 400
    off = [0, B, B + B * f1, B + B * f1 + B * f1 * f2]
    R, r_x = off[3], off[2]
    ids = torch.randint(1, N, (R,), generator=g).to(DEV)
    dhid = (torch.randn(R, 64, generator=g) * 0.1).to(BF).to(DEV); dhid[:, 32:] = 0
    W0 = (torch.randn(32, E, generator=g) / 6).to(BF).to(DEV)
    W0T = torch.zeros(E, 64, dtype=BF, device=DEV); W0T[:, :32] = W0.t()
    datt = (dhid[:, :32].double() @ W0.double()).float()
    dx = torch.randn(r_x, E, generator=g).to(DEV); dagg = torch.randn(r_x, E, generator=g).to(DEV)
    ws = torch.rand(R - B, generator=g).to(DEV)
    Wp = (torch.randn(E, E, generator=g) / 8).to(BF).to(DEV); WpT = Wp.t().contiguous()
    din0 = torch.zeros(R, E, dtype=BF, device=DEV); bpart = torch.full((256, E), 9.0, device=DEV)
    gtab = torch.zeros(N + 2, E, device=DEV); deraw = torch.zeros(R, E, device=DEV)
    offh = (ctypes.c_int64 * 6)(*(off + [0, 0])); fanh = (ctypes.c_int32 * 6)(1, f1, f2, 1, 1, 1)
    L = nat.lib()
    nat.check(L.gsage_prep_rows_bwd(dhid.data_ptr() if with_dhid else None, 64, W0T.data_ptr() if with_dhid else None, 64,
                                    None if with_dhid else datt.data_ptr(), E, dx.data_ptr(), E, r_x, dagg.data_ptr(), E,
                                    ws.data_ptr() if with_ws else None, 3, offh, fanh, R, E, din0.data_ptr(), E, bpart.data_ptr(),
                                    256, WpT.data_ptr(), E, ids.data_ptr(), B, N + 1, gtab.data_ptr(), E,
                                    deraw.data_ptr() if deraw_mode else None, E, ops._stream()), "prep_rows_bwd")
    torch.cuda.synchronize()
    pos = torch.arange(R, device=DEV)
    parent = torch.zeros(R, dtype=torch.long, device=DEV)
    parent[off[1]:off[2]] = (pos[off[1]:off[2]] - off[1]) // f1
    parent[off[2]:] = off[1] + (pos[off[2]:] - off[2]) // f2
    w = torch.zeros(R, dtype=torch.float64, device=DEV)
    w[B:] = ws.double() if with_ws else torch.cat([torch.full((B * f1,), 1.0 / f1), torch.full((B * f1 * f2,), 1.0 / f2)]).double().to(DEV)
    v = datt.double().clone()
    v[:r_x] += dx.double()
    v += w.unsqueeze(1) * dagg.double()[parent]
    torch.testing.assert_close(din0.double(), v, rtol=1.0 / 128, atol=2e-2)
    torch.testing.assert_close(bpart.double().sum(0), v.sum(0), rtol=1e-4, atol=1e-3)
    d = din0.double() @ Wp.double()                     # (the kernel multiplies the rounded operand copy, as the GEMM did)
    if deraw_mode:
        torch.testing.assert_close(deraw.double(), d, rtol=1e-4, atol=1e-4)
        assert float(gtab.abs().max()) == 0.0
    else:
        rows = ids.clone(); rows[:B] = N + 1
        want = torch.zeros(N + 2, E, dtype=torch.float64, device=DEV).index_add_(0, rows, d)
        torch.testing.assert_close(gtab.double(), want, rtol=1e-4, atol=1e-3)
        assert float(deraw.abs().max()) == 0.0


# ---- K5b with two output tiles per workgroup (csrc/gsage_wgrad.hip, PAIR) ---------------------------------------------
@pytest.mark.parametrize("M,ntot,K,with_rows", [(70000, 512, 602, True), (65536, 256, 128, False), (66001, 512, 90, False)])
def test_wgrad_pairs_of_tiles_per_workgroup(monkeypatch, M, ntot, K, with_rows):
    """gsage_wgrad_multi in PAIR mode (GSAGE_WGRAD_PAIR=1: waves 0 / 1 and 2 / 3 share their rows of A) against fp64 and
    against the one-tile-per-workgroup launch: the same products, summed per slice in another grouping."""
    g = torch.Generator(device="cpu"); g.manual_seed(5)
    lda = _r64(K)
    N = 9000
    A = torch.zeros(N if with_rows else M, lda, dtype=BF, device=DEV)
    A[:, :K] = torch.randn(A.shape[0], K, generator=g).to(DEV).to(BF)
    rows = torch.randint(0, N, (M,), generator=g).to(DEV) if with_rows else None
    dC = (torch.randn(M, ntot, generator=g) * 0.1).to(DEV).to(BF)
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("GSAGE_WGRAD_PAIR", mode)
        target = 240
        rps, S, ldk = ops.wgrad_plan(M, ntot, K, target)
        pair = nat.lib().gsage_wgrad_pair_ok(nat.BF16, M, ntot, ntot, ops.wgrad_plan(M, ntot, K, 2 * target)[0])
        assert pair == int(mode)
        if pair:
            target *= 2
            rps, S, ldk = ops.wgrad_plan(M, ntot, K, target)
        slabs = torch.full((S, ntot, ldk), float("nan"), device=DEV)
        ops.wgrad_multi([(dC, A, lda, 0, M, ntot, K, ntot, slabs, target, rows)])
        torch.cuda.synchronize()
        outs[mode] = slabs[:, :, :K].double().sum(0)
        assert torch.isfinite(outs[mode]).all()
    x = (A[rows] if with_rows else A)[:, :K].double()
    want = dC.double().t() @ x
    scale = float(want.abs().max())
    for mode in ("0", "1"):
        assert float((outs[mode] - want).abs().max()) < 2e-5 * scale * (M / 1000.0) ** 0.5, mode
    assert float((outs["0"] - outs["1"]).abs().max()) < 2e-5 * scale * (M / 1000.0) ** 0.5


# ---- K5 for short reductions: weight-stationary persistent workgroups (csrc/gsage_packed.hip, k_linear_nt_packed_ws) -------
@pytest.mark.parametrize("M,N,K,groups,act", [(84992, 128, 128, 2, 1), (5000, 256, 128, 2, 1), (1031, 128, 64, 1, 0),
                                              (3000, 128, 192, 2, 1), (4097, 384, 256, 1, 1), (700, 128, 100, 2, 0),
                                              (256, 128, 128, 1, 1), (20000, 128, 40, 2, 1)])
def test_linear_nt_packed_weight_stationary_equals_the_tile_per_workgroup_kernel(monkeypatch, M, N, K, groups, act):
    """gsage_linear_nt_packed with bf16 output and K <= 256 runs the weight-stationary kernel (W in registers for the
    launch, a workgroup walks several row tiles, LDS-DMA ring across tile boundaries): the same MFMAs in the same order
    as the one-tile-per-workgroup kernel (GSAGE_K5_WS=0) -- torch.equal --, both within bf16 of fp64; ragged M, one to
    four k-tiles, several column blocks, two groups with the row list on group 0, rows beyond the output untouched."""
    g = torch.Generator(device="cpu"); g.manual_seed(M + N + K)
    ld = _r64(K)
    n_src = M + 37
    A = torch.zeros(groups, n_src, ld, dtype=BF, device=DEV)
    A[:, :, :K] = torch.randn(groups, n_src, K, generator=g).to(DEV).to(BF)
    W = (torch.randn(groups, N, K, generator=g) / np.sqrt(K)).to(DEV)
    bias = torch.randn(groups, N, generator=g).to(DEV)
    rows = torch.randperm(n_src, generator=g)[:M].to(DEV).contiguous()
    Wp = ops.pack_weight(W)
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("GSAGE_K5_WS", mode)
        C = torch.full((M + 3, groups * N + 8), -7.0, dtype=BF, device=DEV)
        ops._linear_packed_launch(A.data_ptr(), ld, rows.data_ptr(), 1, Wp.data_ptr(), bias.data_ptr(), C.data_ptr(),
                                  C.stride(0), M, N, K, act, groups, n_src * ld, N, nat.BF16)
        torch.cuda.synchronize()
        outs[mode] = C
    assert torch.equal(outs["0"], outs["1"])
    C = outs["1"]
    assert torch.all(C[M:].float() == -7.0) and torch.all(C[:, groups * N:].float() == -7.0)
    for gi in range(groups):
        a = (A[0][rows] if gi == 0 else A[gi][:M])[:, :K].double()
        want = a @ W[gi].to(BF).double().t() + bias[gi].double()
        if act == 1:
            want = torch.relu(want)
        torch.testing.assert_close(C[:M, gi * N:(gi + 1) * N].double(), want, rtol=2 ** -7, atol=2 ** -7)


# ---- inter-level backward routing, eight columns per thread (csrc/gsage_optim.hip, k_bwd_merge_v8) --------------------
@pytest.mark.parametrize("B,fans,D", [(512, (15, 10), 256), (37, (5, 3), 128), (64, (10,), 64), (9, (4, 3, 2), 40)])
def test_bwd_merge_eight_columns_per_thread_equals_the_four_column_kernel(monkeypatch, B, fans, D):
    """gsage_bwd_merge on bf16 rows of whole 16-byte chunks (k_bwd_merge_v8) against the 4-column kernel (GSAGE_MERGE_V8=0):
    torch.equal; and against the definition (reference: autograd of nn_modules.py:197-202 w.r.t. the level's input, the ReLU
    of models.py's layer below)."""
    import ctypes
    g = torch.Generator(device="cpu"); g.manual_seed(B + D)
    sizes = [B]
    for f in fans:
        sizes.append(sizes[-1] * f)
    off = [0]
    for sz in sizes:
        off.append(off[-1] + sz)
    n_hops = len(sizes)
    R, r_x = off[n_hops], off[n_hops - 1]
    H = torch.relu(torch.randn(R, D, generator=g)).to(BF).to(DEV)
    DG = torch.randn(r_x, 2 * D, generator=g).to(DEV)
    offh = (ctypes.c_int64 * 6)(*(off[:n_hops] + [0] * (6 - n_hops)))
    fanh = (ctypes.c_int32 * 6)(*([1] + list(fans) + [1] * (5 - len(fans))))
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("GSAGE_MERGE_V8", mode)
        dH = torch.full((R + 1, D), -3.0, dtype=BF, device=DEV)
        nat.check(nat.lib().gsage_bwd_merge(H.data_ptr(), nat.BF16, D, DG.data_ptr(), 2 * D, D, dH.data_ptr(), D, R, r_x, D,
                                            n_hops, offh, fanh, ops._stream()), "bwd_merge")
        torch.cuda.synchronize()
        outs[mode] = dH
    assert torch.equal(outs["0"], outs["1"]) and torch.all(outs["1"][R].float() == -3.0)
    want = torch.zeros(R, D, dtype=torch.float64, device=DEV)
    want[:r_x] += DG[:, :D].double()
    for k in range(1, n_hops):
        parent = off[k - 1] + torch.arange(sizes[k], device=DEV) // fans[k - 1]
        want[off[k]:off[k + 1]] += DG[parent, D:].double() / fans[k - 1]
    want = torch.where(H.double() > 0, want, torch.zeros_like(want))
    torch.testing.assert_close(outs["1"][:R].double(), want, rtol=2 ** -7, atol=1e-6)
