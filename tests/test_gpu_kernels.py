"""-m gpu: every entry point of the C ABI (include/gsage.h) against the CPU oracle on the same
seeded inputs.  Integer results bit-exact; fp32 paths 1e-5 relative; bf16-storage paths are
compared against the oracle evaluated on the bf16-rounded inputs (so only accumulation order and
the final rounding differ): 1e-5 relative for fp32 outputs, 2^-8 for bf16 outputs."""
import numpy as np
import pytest
import torch

from conftest import load_golden, pkg
from util import close, csr_of

pytestmark = pytest.mark.gpu
gs = pkg()
ops = gs.ops
nat = gs._native
DEV = "cuda"


def bf16_round(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float32)).bfloat16().float().numpy()


def dcsr(adj):
    return gs.DeviceCSR.from_scipy(adj, torch.device(DEV))


def test_device_and_library():
    info = nat.device_info()
    assert info is not None and info["arch"].startswith("gfx950"), info
    assert info["wave_size"] == 64
    assert nat.lib().gsage_abi_version() == nat.ABI_VERSION


def test_sampler_sel_bit_exact_all_golden_cases():
    from oracle import cpu as ocpu
    g = load_golden("sampler_kat.npz")
    csrs = [dcsr(csr_of(g, "g%d_" % i)) for i in range(3)]
    before = nat.launch_count()
    for c in range(int(g["n_cases"])):
        p = "c%d_" % c
        csr = csrs[int(g[p + "graph"])]
        ids = torch.from_numpy(g[p + "ids"]).to(DEV)
        sel = torch.from_numpy(g[p + "sel"].astype(np.int32)).to(DEV)
        out = ops.sample_csr(csr, ids, int(g[p + "n"]), sel=sel)
        assert out.dtype == torch.int64 and out.is_cuda
        assert np.array_equal(out.cpu().numpy(), g[p + "out"]), c
        csr.check()
    assert nat.launch_count() - before == int(g["n_cases"])       # the HIP path really ran


def test_sampler_philox_equals_oracle_and_counter():
    from oracle import cpu as ocpu
    g = load_golden("sampler_kat.npz")
    adj = csr_of(g, "g2_")
    csr = dcsr(adj)
    rng = np.random.RandomState(1)
    for (M, n, g0, call) in [(1, 1, 0, 0), (5, 3, 0, 7), (129, 25, 3, 1), (1000, 10, 123457, 2 ** 33 + 5)]:
        ids = rng.randint(0, adj.shape[0], size=M)
        sel_out = torch.empty(M * n, dtype=torch.int32, device=DEV)
        out = ops.sample_csr(csr, torch.from_numpy(ids).to(DEV), n,
                             philox={"seed": 2 ** 40 + 17, "call_base": call, "g0": g0, "sel_out": sel_out})
        sel = ocpu.philox_sel(2 ** 40 + 17, call, g0, M * n, adj.shape[1])
        assert np.array_equal(sel_out.cpu().numpy().astype(np.int64), sel)
        assert np.array_equal(out.cpu().numpy(), ocpu.sample_csr_sel(adj.indptr, adj.data, ids, n, sel))
    # device-side call counter (what a captured graph advances between replays)
    ctr = torch.zeros(1, dtype=torch.int64, device=DEV)
    ids = torch.from_numpy(rng.randint(0, adj.shape[0], size=40)).to(DEV)
    a = ops.sample_csr(csr, ids, 5, philox={"seed": 3, "call_ctr": ctr, "call_base": 1})
    nat.check(nat.lib().gsage_counter_add(ctr.data_ptr(), 2, None))
    torch.cuda.synchronize()
    assert int(ctr.item()) == 2
    b = ops.sample_csr(csr, ids, 5, philox={"seed": 3, "call_ctr": ctr, "call_base": 1})
    ref_a = ocpu.sample_csr_sel(adj.indptr, adj.data, ids.cpu().numpy(), 5, ocpu.philox_sel(3, 1, 0, 200, adj.shape[1]))
    ref_b = ocpu.sample_csr_sel(adj.indptr, adj.data, ids.cpu().numpy(), 5, ocpu.philox_sel(3, 3, 0, 200, adj.shape[1]))
    assert np.array_equal(a.cpu().numpy(), ref_a) and np.array_equal(b.cpu().numpy(), ref_b)


def test_sampler_edge_cases():
    g = load_golden("sampler_kat.npz")
    csr = dcsr(csr_of(g, "g0_"))
    assert ops.sample_csr(csr, torch.empty(0, dtype=torch.int64, device=DEV), 3,
                          sel=torch.empty(0, dtype=torch.int32, device=DEV)).numel() == 0
    with pytest.raises(AssertionError):
        ops.sample_csr(csr, torch.zeros(2, dtype=torch.int64, device=DEV), 0,
                       sel=torch.empty(0, dtype=torch.int32, device=DEV))
    out = ops.sample_csr(csr, torch.tensor([1, 10 ** 6, 0, 3], device=DEV), 2,
                         sel=torch.zeros(8, dtype=torch.int32, device=DEV))
    assert out[2:4].tolist() == [0, 0] and out[4:].tolist() == [0, 0, 0, 0]    # bad id / dummy / deg 0
    with pytest.raises(IndexError):
        csr.check()
    csr.check()                                                             # flag was cleared


@pytest.mark.parametrize("D,ld", [(602, 640), (602, 602), (64, 64), (7, 7), (130, 136), (1433, 1472)])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_gather_mean_vs_oracle(D, ld, dtype):
    from oracle import cpu as ocpu
    rng = np.random.RandomState(D + ld)
    R = 500
    tdt = ops.torch_dtype(dtype)
    host = rng.normal(size=(R, D)).astype(np.float32)
    host[0] = 0
    table = torch.zeros(R, ld, dtype=tdt, device=DEV)
    table[:, :D] = torch.from_numpy(host).to(DEV).to(tdt)
    ref_table = table[:, :D].float().cpu().numpy()
    store = gs.FeatureStore(table, D)
    for (M, n) in [(1, 1), (33, 10), (16, 25), (257, 3), (40, 1)]:
        ids = torch.from_numpy(rng.randint(0, R, size=M * n)).to(DEV)
        ref = ocpu.gather_mean_f32(ref_table, ids.cpu().numpy(), M, n)
        out = ops.gather_mean(store, ids, M, n, out_dtype=torch.float32)
        close(out.cpu().numpy(), ref, (D, ld, dtype, M, n), 1e-5, 1e-6)
        if n == 1:
            assert np.array_equal(out.cpu().numpy(), ref)            # a plain gather is exact
        outb = ops.gather_mean(store, ids, M, n, out_dtype=tdt, out_ld=ld if ld % 8 == 0 else None)
        assert outb.shape[1] == (ld if ld % 8 == 0 else D)
        close(outb[:, :D].float().cpu().numpy(), ref, ("lowp", D, ld, dtype, M, n), 2 ** -8, 2 ** -8)
        assert float(outb[:, D:].float().abs().sum()) == 0.0        # padding stays zero
    # in-order rows (ids == NULL): segment mean of a tensor
    M, n = 20, 5
    ref = ocpu.gather_mean_f32(ref_table[:M * n], None, M, n)
    out = ops._gather_mean_raw(table, D, None, M, n, torch.float32)
    close(out.cpu().numpy(), ref, "inorder", 1e-5, 1e-6)


def test_segment_mean_autograd_and_scatter_add():
    rng = np.random.RandomState(5)
    M, n, D = 37, 10, 50
    nb = torch.from_numpy(rng.normal(size=(M * n, D)).astype(np.float32)).to(DEV).requires_grad_(True)
    out = ops.segment_mean(nb, M)
    close(out.detach().cpu().numpy(), nb.detach().cpu().numpy().reshape(M, n, D).mean(1), "fwd", 1e-5, 1e-6)
    G = torch.from_numpy(rng.normal(size=(M, D)).astype(np.float32)).to(DEV)
    (out * G).sum().backward()
    close(nb.grad.cpu().numpy(), np.repeat(G.cpu().numpy() / n, n, axis=0), "bwd", 1e-6, 1e-7)
    # K6: dense embedding gradient
    table = torch.nn.Parameter(torch.from_numpy(rng.normal(size=(30, 16)).astype(np.float32)).to(DEV))
    ids = torch.from_numpy(rng.randint(0, 30, size=200)).to(DEV)
    rows = ops.embedding_rows(table, ids)
    assert np.array_equal(rows.detach().cpu().numpy(), table.detach().cpu().numpy()[ids.cpu().numpy()])
    G = torch.from_numpy(rng.normal(size=(200, 16)).astype(np.float32)).to(DEV)
    (rows * G).sum().backward()
    ref = np.zeros((30, 16), dtype=np.float64)
    np.add.at(ref, ids.cpu().numpy(), G.cpu().numpy().astype(np.float64))
    close(table.grad.cpu().numpy(), ref, "scatter", 1e-5, 1e-5)


def _lin_ref(A, W, b, act):
    y = A.astype(np.float64) @ W.astype(np.float64).T
    if b is not None:
        y = y + b.astype(np.float64)
    if act == 1:
        y = np.maximum(y, 0)
    elif act == 2:
        y = np.tanh(y)
    return y


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (64, 128, 64), (65, 129, 70), (513, 41, 256), (200, 512, 602), (31, 32, 1433)])
def test_linear_nt_mfma_layouts(dtype, M, N, K):
    """Asymmetric random operands: catches any row/column/k-slot mix-up of the MFMA fragments."""
    rng = np.random.RandomState(M * 7 + N * 3 + K)
    A = rng.normal(size=(M, K)).astype(np.float32)
    W = (rng.normal(size=(N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.normal(size=(N,)).astype(np.float32)
    if dtype == "bf16":
        A, W = bf16_round(A), bf16_round(W)
    At, Wt, bt = [torch.from_numpy(v).to(DEV) for v in (A, W, b)]
    for act in (0, 1, 2):
        for bias in (None, bt):
            out = ops.linear(At, Wt, bias, act, compute_dtype=dtype)
            ref = _lin_ref(A, W, b if bias is not None else None, act)
            close(out.cpu().numpy(), ref, (dtype, M, N, K, act, bias is not None), 2e-5, 2e-6)
    if dtype == "bf16":
        outb = ops.linear(At, Wt, bt, 1, compute_dtype=dtype, out_dtype=torch.bfloat16)
        close(outb.float().cpu().numpy(), _lin_ref(A, W, b, 1), "bf16 out", 2 ** -8, 2 ** -8)


@pytest.mark.parametrize("M,N,K,groups", [(13312, 128, 602, 2), (100, 128, 602, 2), (513, 96, 256, 1),
                                          (65, 41, 70, 1), (1000, 512, 640, 1), (77, 128, 64, 1),
                                          (300, 256, 200, 2), (64, 128, 320, 1), (2000, 160, 1433, 1)])
def test_linear_nt_packed_weight_operand(M, N, K, groups):
    """gsage_linear_nt_packed (W in MFMA fragment order, loaded straight into registers) against fp64:
    k-tile counts 1..23 (ring not full / exactly full / steady state + drain in every phase), ragged M
    and N, grouped launches, fp32 and bf16 outputs, bias + activation."""
    rng = np.random.RandomState(M + N + K + groups)
    ld = -(-K // 64) * 64
    A = np.zeros((groups, M, ld), dtype=np.float32)
    A[:, :, :K] = bf16_round(rng.normal(size=(groups, M, K)))
    W = bf16_round(rng.normal(size=(groups, N, K)) / np.sqrt(K))
    b = rng.normal(size=(groups, N)).astype(np.float32)
    At = torch.from_numpy(A).to(DEV).bfloat16().contiguous()
    Wt = torch.from_numpy(W).to(DEV)
    bt = torch.from_numpy(b).to(DEV)
    for wsrc in (Wt, Wt.bfloat16()):                           # the packer takes fp32 or bf16 weights
        Wp = ops.pack_weight(wsrc)
        for act, bias, cdt in ((0, None, torch.float32), (1, bt, torch.float32), (2, bt, torch.bfloat16)):
            C = torch.full((M, groups * N), -7.0, dtype=cdt, device=DEV)
            ops._linear_packed_launch(At.data_ptr(), ld, None, 0, Wp.data_ptr(), bias.data_ptr() if bias is not None else None,
                                      C.data_ptr(), groups * N, M, N, K, act, groups, M * ld, N,
                                      nat.BF16 if cdt == torch.bfloat16 else nat.F32)
            for g in range(groups):
                ref = _lin_ref(A[g][:, :K], W[g], b[g] if bias is not None else None, act)
                tol = (2 ** -8, 2 ** -8) if cdt == torch.bfloat16 else (2e-5, 2e-6)
                close(C[:, g * N:(g + 1) * N].float().cpu().numpy(), ref, (M, N, K, g, act), *tol)
    # row indirection on the A operand (group 0 only), as the fused projection uses it
    if groups == 2:
        rows = torch.from_numpy(rng.permutation(M)).to(DEV)
        C = torch.zeros(M, 2 * N, dtype=torch.float32, device=DEV)
        ops._linear_packed_launch(At.data_ptr(), ld, rows.data_ptr(), 1, Wp.data_ptr(), None, C.data_ptr(), 2 * N,
                                  M, N, K, 0, 2, M * ld, N, nat.F32)
        perm = rows.cpu().numpy()
        close(C[:, :N].cpu().numpy(), _lin_ref(A[0][perm][:, :K], W[0], None, 0), "a_rows g0", 2e-5, 2e-6)
        close(C[:, N:].cpu().numpy(), _lin_ref(A[1][:, :K], W[1], None, 0), "a_rows g1", 2e-5, 2e-6)


@pytest.mark.parametrize("mode", ["max", "mean"])
@pytest.mark.parametrize("M,n,D,H", [(1280, 10, 602, 512), (53, 25, 602, 512), (7, 1, 70, 130), (64, 3, 256, 64),
                                     (3, 64, 40, 128), (33, 7, 1433, 1024), (10, 10, 128, 512),
                                     # (round 6: fan-outs 5 / 10 / 15 / 20 / 25 pool in registers under max pooling)
                                     (41, 5, 602, 512), (21, 15, 100, 256), (9, 20, 602, 512), (2000, 10, 602, 300)])
def test_pool_mlp_packed_equals_pool_mlp(mode, M, n, D, H):
    """gsage_pool_mlp_packed (64 rows x 512 hidden columns per workgroup, W from the fragment-ordered
    operand) against gsage_pool_mlp: same k order inside every accumulator, so pooled values, argmax
    and sign bits must agree exactly; row indirection included."""
    rng = np.random.RandomState(M * 3 + n + D + H)
    ld = -(-D // 64) * 64
    R = 400
    table = torch.zeros(R, ld, dtype=torch.bfloat16, device=DEV)
    table[:, :D] = torch.from_numpy(rng.normal(size=(R, D)).astype(np.float32)).to(DEV).bfloat16()
    W = torch.zeros(H, ld, dtype=torch.bfloat16, device=DEV)
    W[:, :D] = torch.from_numpy((rng.normal(size=(H, D)) / np.sqrt(D)).astype(np.float32)).to(DEV).bfloat16()
    b = torch.from_numpy(rng.normal(size=(H,)).astype(np.float32)).to(DEV)
    ids = torch.from_numpy(rng.randint(0, R, size=M * n)).to(DEV)
    rows = table[ids].contiguous()
    Wp = ops.pack_weight(W, K=D)
    pm = nat.POOL_MAX if mode == "max" else nat.POOL_MEAN
    want_mask = mode == "mean" and H % 32 == 0

    def run(packed, A, a_rows):
        pooled = torch.full((M, H), -3.0, device=DEV)
        pooled_b = torch.zeros(M, H, dtype=torch.bfloat16, device=DEV)
        arg = torch.full((M, H), -1, dtype=torch.int32, device=DEV) if mode == "max" else None
        mask = torch.zeros(M * n, H // 32, dtype=torch.int32, device=DEV) if want_mask else None
        ap = arg.data_ptr() if arg is not None else None
        mp = mask.data_ptr() if mask is not None else None
        ar = a_rows.data_ptr() if a_rows is not None else None
        if packed:
            nat.check(nat.lib().gsage_pool_mlp_packed(A.data_ptr(), ld, ar, Wp.data_ptr(), b.data_ptr(), M, n, H, D, pm,
                                                      pooled.data_ptr(), H, ap, pooled_b.data_ptr(), H, mp, None),
                      "pool_mlp_packed")
        else:
            nat.check(nat.lib().gsage_pool_mlp(A.data_ptr(), nat.BF16, ld, ar, W.data_ptr(), ld, b.data_ptr(), M, n, H, D,
                                               pm, pooled.data_ptr(), H, ap, pooled_b.data_ptr(), H, mp, None), "pool_mlp")
        return pooled, pooled_b, arg, mask

    ref = run(False, rows, None)
    for A, a_rows in ((rows, None), (table, ids)):
        got = run(True, A, a_rows)
        for r, g_ in zip(ref, got):
            if r is not None:
                assert torch.equal(r, g_), (mode, M, n, D, H, a_rows is not None)
    # and the reference values themselves against fp64 (guards both kernels)
    hid = np.maximum(rows[:, :D].float().cpu().numpy().astype(np.float64) @ W[:, :D].float().cpu().numpy().astype(np.float64).T
                     + b.cpu().numpy(), 0).reshape(M, n, H)
    want = hid.max(1) if mode == "max" else hid.mean(1)
    close(ref[0].cpu().numpy(), want, ("pool", mode), 2e-5, 2e-5)


def test_linear_identity_times_asymmetric():
    """A = I: output must be exactly W^T laid out [m, j] (guide: transpose-detecting check)."""
    K = 64
    W = torch.arange(96 * K, dtype=torch.float32, device=DEV).view(96, K) / 64.0
    for dtype in ("fp32", "bf16"):
        out = ops.linear(torch.eye(K, device=DEV), W, None, 0, compute_dtype=dtype)
        ref = W.t() if dtype == "fp32" else W.bfloat16().float().t()
        assert torch.equal(out, ref.contiguous())


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_sage_project_grouped_gathered_and_backward(dtype):
    rng = np.random.RandomState(3)
    R, D, h, M, n = 400, 602, 128, 70, 10
    store = gs.FeatureStore.from_array(rng.normal(size=(R, D)).astype(np.float32), torch.device(DEV), dtype=dtype)
    tab = store.dense().cpu().numpy()
    ids_x = torch.from_numpy(rng.randint(0, R, size=M)).to(DEV)
    ids_n = torch.from_numpy(rng.randint(0, R, size=M * n)).to(DEV)
    Wx = torch.nn.Parameter(torch.from_numpy((rng.normal(size=(h, D)) / 25).astype(np.float32)).to(DEV))
    Wn = torch.nn.Parameter(torch.from_numpy((rng.normal(size=(h, D)) / 25).astype(np.float32)).to(DEV))
    ops.set_compute_dtype(dtype)
    try:
        before = nat.launch_count()
        agg = ops.gather_mean(store, ids_n, M, n, out_dtype=ops.torch_dtype(dtype), out_ld=store.ld)
        out = ops.sage_project(store[ids_x], agg, Wx, Wn, nat.ACT_RELU, dtype, torch.float32)
        assert nat.launch_count() - before == 2          # one gather+mean, ONE grouped GEMM
        wx = Wx.detach().cpu().numpy(); wn = Wn.detach().cpu().numpy()
        if dtype == "bf16":
            wx, wn = bf16_round(wx), bf16_round(wn)
        aggr = agg[:, :D].float().cpu().numpy()
        ref = np.maximum(np.concatenate([tab[ids_x.cpu().numpy()].astype(np.float64) @ wx.T.astype(np.float64),
                                         aggr.astype(np.float64) @ wn.T.astype(np.float64)], axis=1), 0)
        close(out.detach().cpu().numpy(), ref, ("fwd", dtype), 2e-5, 2e-6)
        G = torch.from_numpy(rng.normal(size=(M, 2 * h)).astype(np.float32)).to(DEV)
        (out * G).sum().backward()
        g = G.cpu().numpy() * (ref > 0)
        tol = (1e-4, 1e-5) if dtype == "fp32" else (2e-2, 2e-2)      # bf16: dOut is rounded to bf16
        close(Wx.grad.cpu().numpy(), g[:, :h].T @ tab[ids_x.cpu().numpy()], ("dWx", dtype), *tol)
        close(Wn.grad.cpu().numpy(), g[:, h:].T @ aggr, ("dWn", dtype), *tol)
    finally:
        ops.set_compute_dtype("bf16")


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("mode", ["max", "mean"])
@pytest.mark.parametrize("M,n,D,H", [(5, 10, 70, 512), (33, 25, 602, 512), (7, 1, 16, 130), (64, 3, 256, 64), (3, 64, 40, 128)])
def test_pool_mlp_vs_reference_math(dtype, mode, M, n, D, H):
    rng = np.random.RandomState(M + n + D)
    nb = rng.normal(size=(M * n, D)).astype(np.float32)
    W = (rng.normal(size=(H, D)) / np.sqrt(D)).astype(np.float32)
    b = rng.normal(size=(H,)).astype(np.float32)
    if dtype == "bf16":
        nb, W = bf16_round(nb), bf16_round(W)
    hid = np.maximum(nb.astype(np.float64) @ W.T.astype(np.float64) + b, 0).reshape(M, n, H)
    ref = hid.max(1) if mode == "max" else hid.mean(1)
    code = nat.POOL_MAX if mode == "max" else nat.POOL_MEAN
    nbt = torch.from_numpy(nb).to(DEV).requires_grad_(True)
    Wt = torch.nn.Parameter(torch.from_numpy(W).to(DEV))
    bt = torch.nn.Parameter(torch.from_numpy(b).to(DEV))
    out = ops.pool_mlp(nbt, Wt, bt, M, code, compute_dtype=dtype)
    close(out.detach().cpu().numpy(), ref, (dtype, mode, M, n, D, H), 2e-5, 2e-6)
    G = rng.normal(size=(M, H)).astype(np.float32)
    (out * torch.from_numpy(G).to(DEV)).sum().backward()
    if mode == "max":
        gh = np.zeros_like(hid)
        am = hid.argmax(1)
        np.put_along_axis(gh, am[:, None, :], (G * (ref > 0))[:, None, :], axis=1)
    else:
        gh = np.repeat((G / n)[:, None, :], n, axis=1) * (hid > 0)
    gh = gh.reshape(M * n, H)
    tol = (1e-4, 1e-5) if dtype == "fp32" else (2e-2, 2e-2)
    close(Wt.grad.cpu().numpy(), gh.T @ nb, ("dW", dtype, mode), *tol)
    close(bt.grad.cpu().numpy(), gh.sum(0), ("db", dtype, mode), *tol)
    close(nbt.grad.cpu().numpy(), gh @ W, ("dneibs", dtype, mode), *tol)
    # gathered variant: rows come from a table through ids
    store = gs.FeatureStore.from_array(nb, torch.device(DEV), dtype=dtype)
    perm = torch.from_numpy(rng.permutation(M * n)).to(DEV)
    inv = torch.argsort(perm)
    shuffled = gs.FeatureStore(store.data[perm].contiguous(), D)
    out2 = ops.pool_mlp(shuffled[inv], Wt, bt, M, code, compute_dtype=dtype)
    close(out2.detach().cpu().numpy(), ref, ("gathered", dtype, mode), 2e-5, 2e-6)


def test_attn_aggregate_forward_backward():
    rng = np.random.RandomState(9)
    for (M, n, Ha, D) in [(6, 5, 32, 20), (33, 20, 32, 64), (9, 15, 32, 602), (2, 64, 8, 9)]:
        na = torch.from_numpy(rng.normal(size=(M * n, Ha)).astype(np.float32)).to(DEV).requires_grad_(True)
        xa = torch.from_numpy(rng.normal(size=(M, Ha)).astype(np.float32)).to(DEV).requires_grad_(True)
        nb = torch.from_numpy(rng.normal(size=(M * n, D)).astype(np.float32)).to(DEV).requires_grad_(True)
        out = ops.attn_aggregate(na, xa, nb, M)
        na_c, xa_c, nb_c = [t.detach().cpu().double().requires_grad_(True) for t in (na, xa, nb)]
        s = torch.bmm(na_c.view(M, n, Ha), xa_c.view(M, Ha, 1)).squeeze(2)
        ref = (nb_c.view(M, n, D) * torch.softmax(s, dim=1).unsqueeze(-1)).sum(1)
        close(out.detach().cpu().numpy(), ref.detach().numpy(), ("attn fwd", M, n), 1e-5, 1e-6)
        G = torch.from_numpy(rng.normal(size=(M, D)).astype(np.float32))
        (out * G.to(DEV)).sum().backward()
        (ref * G.double()).sum().backward()
        close(na.grad.cpu().numpy(), na_c.grad.numpy(), "dna", 1e-4, 1e-5)
        close(xa.grad.cpu().numpy(), xa_c.grad.numpy(), "dxa", 1e-4, 1e-5)
        close(nb.grad.cpu().numpy(), nb_c.grad.numpy(), "dnb", 1e-4, 1e-5)


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
@pytest.mark.parametrize("D,n", [(64, 20), (64, 15), (100, 10), (256, 15), (256, 32), (602, 10), (602, 25), (760, 5),
                                 (800, 10), (24, 1), (602, 3)])
def test_attn_kernels_all_lane_groupings_vs_fp64(dtype, D, n):
    """K4 / K4' through the C ABI on table rows read through ids: every lane grouping of the grouped kernels
    (8 / 16 / 32 lanes per child row, 1..3 chunks per lane), ragged last passes (n not a multiple of the children
    in flight), and the wide fallback (rows too wide for the grouped kernels: 800 bf16 columns, fp32 beyond 384).
    Reference: fp64 torch on the same (bf16-rounded) rows."""
    T = torch.bfloat16 if dtype == "bf16" else torch.float32
    code = nat.BF16 if dtype == "bf16" else nat.F32
    vec = 8 if dtype == "bf16" else 4
    rng = np.random.RandomState(D * 7 + n)
    M, Ha, rows_in_table = 37, 32, 900
    ld = -(-D // vec) * vec
    table = torch.zeros(rows_in_table, ld, dtype=T, device=DEV)
    table[:, :D] = torch.from_numpy(rng.normal(size=(rows_in_table, D)).astype(np.float32)).to(DEV).to(T)
    ids = torch.from_numpy(rng.randint(0, rows_in_table, size=M * n)).to(DEV)
    na = torch.from_numpy(rng.normal(size=(M * n, Ha)).astype(np.float32)).to(DEV)
    xa = torch.from_numpy(rng.normal(size=(M, Ha)).astype(np.float32)).to(DEV)
    g = torch.from_numpy(rng.normal(size=(M, ld)).astype(np.float32)).to(DEV)
    agg = torch.full((M, ld), float("nan"), device=DEV)
    ws = torch.zeros(M * n, device=DEV)
    dna = torch.full((M * n, Ha), float("nan"), device=DEV)
    dxa = torch.full((M, Ha), float("nan"), device=DEV)
    lib, st = nat.lib(), ops._stream()
    nat.check(lib.gsage_attn_aggregate(na.data_ptr(), Ha, xa.data_ptr(), Ha, table.data_ptr(), code, ld, ids.data_ptr(),
                                       M, n, Ha, D, agg.data_ptr(), ld, ws.data_ptr(), st), "fwd")
    nat.check(lib.gsage_attn_bwd(g.data_ptr(), ld, ws.data_ptr(), na.data_ptr(), Ha, xa.data_ptr(), Ha, table.data_ptr(),
                                 code, ld, ids.data_ptr(), M, n, Ha, D, dna.data_ptr(), Ha, dxa.data_ptr(), Ha, st), "bwd")
    torch.cuda.synchronize()
    rows = table[ids][:, :D].double().cpu().view(M, n, D)
    na_c = na.double().cpu().requires_grad_(True)
    xa_c = xa.double().cpu().requires_grad_(True)
    w = torch.softmax(torch.bmm(na_c.view(M, n, Ha), xa_c.view(M, Ha, 1)).squeeze(2), dim=1)
    ref = (rows * w.unsqueeze(-1)).sum(1)
    close(agg[:, :D].cpu().numpy(), ref.detach().numpy(), ("K4 forward", dtype, D, n), 1e-5, 1e-5)
    close(ws.cpu().numpy(), w.detach().reshape(-1).numpy(), "softmax weights", 1e-5, 1e-6)
    (ref * g[:, :D].double().cpu()).sum().backward()
    close(dna.cpu().numpy(), na_c.grad.numpy(), ("d att(neibs)", dtype, D, n), 1e-4, 1e-4)
    close(dxa.cpu().numpy(), xa_c.grad.numpy(), ("d att(x)", dtype, D, n), 1e-4, 1e-4)
    # no row list: children are consecutive rows of the table
    seq = table[ids].contiguous()
    agg2 = torch.full((M, ld), float("nan"), device=DEV)
    nat.check(lib.gsage_attn_aggregate(na.data_ptr(), Ha, xa.data_ptr(), Ha, seq.data_ptr(), code, ld, None,
                                       M, n, Ha, D, agg2.data_ptr(), ld, ws.data_ptr(), st), "fwd")
    torch.cuda.synchronize()
    assert torch.equal(agg2[:, :D], agg[:, :D])


@pytest.mark.parametrize("M,N,K,groups", [(16, 128, 8, 1), (100, 128, 602, 2), (513, 128, 256, 2),
                                          (1300, 8, 70, 1), (3000, 128, 1433, 2), (64, 128, 128, 1), (37, 24, 200, 1)])
def test_wgrad_mfma_vs_fp64(M, N, K, groups):
    """dW_g = dC_g^T @ A_g (K5b): exact in fp32 up to summation order on bf16-rounded operands."""
    rng = np.random.RandomState(M + N + K)
    ld = ((K + 63) // 64) * 64
    tab = np.zeros((groups, M, ld), dtype=np.float32)
    tab[:, :, :K] = bf16_round(rng.normal(size=(groups, M, K)))
    dC = bf16_round(rng.normal(size=(M, groups * N)))
    tabt = torch.from_numpy(tab).to(DEV).bfloat16().contiguous()
    dCt = torch.from_numpy(dC).to(DEV).bfloat16().contiguous()
    out = ops.wgrad(dCt, tabt, ld, M * ld, M, groups * N, K, N)
    assert out.shape == (groups, N, K) and out.dtype == torch.float32
    for g in range(groups):
        ref = dC[:, g * N:(g + 1) * N].astype(np.float64).T @ tab[g][:, :K].astype(np.float64)
        close(out[g].cpu().numpy(), ref, ("wgrad", M, N, K, g), 2e-5, 2e-6)


def test_gather_mean_multi_equals_single_launches():
    rng = np.random.RandomState(4)
    store = gs.FeatureStore.from_array(rng.normal(size=(900, 602)).astype(np.float32), torch.device(DEV), "bf16")
    specs = [(301, 1), (12, 25), (300, 10), (7, 1)]      # row copies of a mixed launch go four rows per lane
    segs, refs = [], []
    for M, n in specs:
        ids = torch.from_numpy(rng.randint(0, 900, size=M * n)).to(DEV)
        out = torch.zeros(M, store.ld, dtype=torch.bfloat16, device=DEV)
        segs.append((store.data, ids, out, M, n))
        refs.append(ops.gather_mean(store, ids, M, n, out_dtype=torch.bfloat16, out_ld=store.ld))
    h = torch.from_numpy(rng.normal(size=(120, 256)).astype(np.float32)).to(DEV).bfloat16()
    o2 = torch.zeros(40, 256, dtype=torch.bfloat16, device=DEV)
    ops.gather_mean_multi(segs, store.ld, store.ld, store.ld)
    ops.gather_mean_multi([(h, None, o2, 40, 3)], 256, 256, 256)
    for (t, i, o, M, n), r in zip(segs, refs):
        assert torch.equal(o, r)
    assert torch.equal(o2, ops._gather_mean_raw(h, 256, None, 40, 3, torch.bfloat16))


@pytest.mark.parametrize("B,C,D", [(512, 41, 256), (33, 7, 32), (5, 64, 600), (100, 2, 1024)])
def test_head_ce_forward_backward_vs_torch(B, C, D):
    """normalize + fc + cross-entropy and all three gradients in two launches (fp32: 1e-5)."""
    import torch.nn.functional as F
    rng = np.random.RandomState(B + C + D)
    E = torch.from_numpy(rng.normal(size=(B, D)).astype(np.float32)).to(DEV)
    W = torch.from_numpy((rng.normal(size=(C, D)) * 0.3).astype(np.float32)).to(DEV)
    b = torch.from_numpy(rng.normal(size=(C,)).astype(np.float32)).to(DEV)
    t = torch.from_numpy(rng.randint(0, C, size=B)).to(DEV)
    L = nat.lib()
    preds = torch.empty(B, C, device=DEV)
    dE = torch.empty(B, D, device=DEV)
    dW, db, loss = torch.empty(C, D, device=DEV), torch.empty(C, device=DEV), torch.empty(1, device=DEV)
    scratch = torch.empty(L.gsage_head_ce_scratch(B, C, D), device=DEV)
    nat.check(L.gsage_head_ce(E.data_ptr(), D, W.data_ptr(), b.data_ptr(), t.data_ptr(), B, C, D,
                              preds.data_ptr(), dE.data_ptr(), nat.F32, D, dW.data_ptr(), db.data_ptr(),
                              loss.data_ptr(), scratch.data_ptr(), None, 0, None))
    Ed, Wd, bd = [v.detach().double().cpu().requires_grad_(True) for v in (E, W, b)]
    pr = F.normalize(Ed, dim=1) @ Wd.t() + bd
    ls = F.cross_entropy(pr, t.cpu())
    ls.backward()
    close(preds.cpu().numpy(), pr.detach().numpy(), "preds", 1e-5, 1e-6)
    assert abs(float(loss.item()) - float(ls)) < 1e-5 * max(1.0, float(ls))
    close(dE.cpu().numpy(), Ed.grad.numpy(), "dE", 1e-5, 1e-7)
    close(dW.cpu().numpy(), Wd.grad.numpy(), "dW", 1e-5, 1e-7)
    close(db.cpu().numpy(), bd.grad.numpy(), "db", 1e-5, 1e-7)


@pytest.mark.parametrize("B,fans", [(512, (25, 10)), (33, (5, 3)), (7, (4, 3, 2)), (1, (1,)), (130, (15, 10, 5))])
def test_fused_multi_hop_sampler_equals_per_hop_launches(B, fans):
    import ctypes
    g = load_golden("sampler_kat.npz")
    adj = csr_of(g, "g2_")
    csr = dcsr(adj)
    rng = np.random.RandomState(B)
    seeds = torch.from_numpy(rng.randint(0, adj.shape[0], size=B)).to(DEV)
    sizes = [B]
    for f in fans:
        sizes.append(sizes[-1] * f)
    ids = torch.zeros(sum(sizes), dtype=torch.int64, device=DEV)
    ids[:B] = seeds
    ctr = torch.full((1,), 6, dtype=torch.int64, device=DEV)
    rank = 3
    fan = (ctypes.c_int32 * len(fans))(*fans)
    nat.check(nat.lib().gsage_sample_hops_philox(csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.n_rows,
                                                 ids.data_ptr(), B, len(fans), fan, csr.max_deg, 99,
                                                 ctr.data_ptr(), 1, rank, None, None, 0, csr.err_flag.data_ptr(), None))
    cur, off = seeds, B
    for k, f in enumerate(fans):
        nxt = ops.sample_csr(csr, cur, f, philox={"seed": 99, "call_ctr": ctr, "call_base": 1 + k,
                                                  "g0": rank * sizes[k + 1]})
        assert torch.equal(ids[off:off + sizes[k + 1]], nxt), (B, fans, k)
        cur, off = nxt, off + sizes[k + 1]
    csr.check()


def test_gather_launch_carrying_a_sampler_equals_separate_launches():
    """gsage_gather_mean_multi_adam with hops: the frontier of a LATER batch sampled inside the gather
    launch (engine.step_queue) is the frontier the stand-alone K1 launch produces, and the gather's
    own output is unchanged."""
    import ctypes
    g = load_golden("sampler_kat.npz")
    adj = csr_of(g, "g2_")
    csr = dcsr(adj)
    rng = np.random.RandomState(11)
    B, fans, D, ld = 37, (6, 4), 40, 64
    sizes = [B, B * 6, B * 24]
    table = torch.zeros(adj.shape[0], ld, dtype=torch.bfloat16, device=DEV)
    table[:, :D] = torch.from_numpy(rng.normal(size=(adj.shape[0], D)).astype(np.float32)).to(DEV).bfloat16()
    queue = torch.from_numpy(rng.randint(0, adj.shape[0], size=(4, B))).to(DEV)
    bidx = torch.full((1,), 2, dtype=torch.int64, device=DEV)
    ctr = torch.full((1,), 8, dtype=torch.int64, device=DEV)

    def desc(ids, call_base, batch_base):
        d = nat.HopsDesc()
        d.rowptr, d.col, d.n_rows = csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.n_rows
        d.ids, d.B, d.n_hops = ids.data_ptr(), B, 2
        for k in range(5):
            d.fan[k] = fans[k] if k < 2 else 1
        d.max_deg, d.seed, d.call_ctr, d.call_base, d.rank = csr.max_deg, 5, ctr.data_ptr(), call_base, 1
        d.seed_queue, d.batch_idx, d.batch_base, d.n_batches = queue.data_ptr(), bidx.data_ptr(), batch_base, 4
        d.err_flag = csr.err_flag.data_ptr()
        return d

    def sample(ids, call_base, batch_base):
        d = desc(ids, call_base, batch_base)
        nat.check(nat.lib().gsage_sample_hops(ctypes.addressof(d), None), "sample_hops")

    cur = torch.zeros(sum(sizes), dtype=torch.int64, device=DEV)
    sample(cur, 0, 0)                                   # the frontier being gathered (batch 2)
    ref_next = torch.zeros_like(cur)
    sample(ref_next, 2, 1)                              # batch 3, the next two Philox call indices
    assert torch.equal(cur[:B], queue[2]) and torch.equal(ref_next[:B], queue[3])
    # ... which is what ticking both counters and sampling without offsets gives
    bidx += 1
    ctr += 2
    chk = torch.zeros_like(cur)
    sample(chk, 0, 0)
    bidx -= 1
    ctr -= 2
    assert torch.equal(chk, ref_next)

    def gather(hops):
        o1 = torch.zeros(B + sizes[1], ld, dtype=torch.bfloat16, device=DEV)
        o2 = torch.zeros(B, ld, dtype=torch.bfloat16, device=DEV)
        o3 = torch.zeros(sizes[1], ld, dtype=torch.bfloat16, device=DEV)
        ops.gather_mean_multi([(table, cur[:B + sizes[1]], o1, B + sizes[1], 1),
                               (table, cur[B:B + sizes[1]], o2, B, fans[0]),
                               (table, cur[B + sizes[1]:], o3, sizes[1], fans[1])], ld, D, ld, hops=hops)
        return o1, o2, o3

    ref = gather(None)
    got_next = torch.zeros_like(cur)
    got = gather(desc(got_next, 2, 1))
    torch.cuda.synchronize()
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    assert torch.equal(got_next, ref_next)
    assert int(bidx.item()) == 2 and int(ctr.item()) == 8      # nothing ticked
    csr.check()


def test_command_list_replay_equals_direct_launches():
    """include/gsage.h "Command lists": recorded launches do not run until replayed, replay
    re-issues them (on any stream) with the recorded arguments, and the launch counter counts them."""
    nat = gs._native
    rng = np.random.RandomState(5)
    R, D, ld, M, n = 300, 40, 64, 50, 7
    table = torch.zeros(R, ld, dtype=torch.bfloat16, device=DEV)
    table[:, :D] = torch.from_numpy(rng.normal(size=(R, D)).astype(np.float32)).to(DEV).bfloat16()
    store = gs.FeatureStore(table, D)
    ids = torch.from_numpy(rng.randint(0, R, size=M * n)).to(DEV)
    ref1 = ops.gather_mean(store, ids, M, n, out_dtype=torch.float32)
    ref2 = ops.gather_mean(store, ids[:M], M, 1, out_dtype=torch.bfloat16, out_ld=ld)
    out1 = torch.full_like(ref1, -7.0)
    out2 = torch.full_like(ref2, -7.0)
    torch.cuda.synchronize()
    with nat.CommandList.record() as cl:
        ops._gather_mean_raw(store.data, store.dim, ids, M, n, torch.float32, None, out=out1)
        ops._gather_mean_raw(store.data, store.ld, ids[:M], M, 1, torch.bfloat16, ld, out=out2)
    assert len(cl) == 2
    torch.cuda.synchronize()
    assert float(out1.min()) == -7.0 and float(out2.float().min()) == -7.0     # nothing ran yet
    before = nat.launch_count()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    cl.replay(side.cuda_stream)
    side.synchronize()
    assert nat.launch_count() == before + 2
    assert torch.equal(out1, ref1) and torch.equal(out2, ref2)
    # a list replays any number of times and tracks the DATA behind the recorded pointers
    table[:, :D] *= 2
    out1.fill_(-7.0)
    torch.cuda.synchronize()
    cl.replay(ops._stream())
    torch.cuda.synchronize()
    assert torch.equal(out1, ops.gather_mean(store, ids, M, n, out_dtype=torch.float32))


def test_wgrad_multi_equals_single_launches():
    """gsage_wgrad_multi: the partial tiles of every problem are bit-identical to one gsage_wgrad
    launch per problem (same tiles, same summation order), whatever mix of shapes shares the launch."""
    rng = np.random.RandomState(11)
    shapes = [(1300, 256, 602, 128), (512, 256, 256, 128), (70, 16, 40, 16), (33, 8, 24, 8)]
    probs, refs = [], []
    for (M, Ntot, K, npg) in shapes:
        lda = gs.store._round_up(K, 8)
        groups = Ntot // npg
        dC = torch.from_numpy(rng.normal(size=(M, Ntot)).astype(np.float32)).to(DEV).bfloat16()
        A = torch.zeros(groups, M, lda, dtype=torch.bfloat16, device=DEV)
        A[:, :, :K] = torch.from_numpy(rng.normal(size=(groups, M, K)).astype(np.float32)).to(DEV).bfloat16()
        rps, S, ldk = ops.wgrad_plan(M, Ntot, K)
        ref = ops.wgrad(dC, A, lda, M * lda, M, Ntot, K, npg, reduce=False).clone()
        slabs = torch.full((S, Ntot, ldk), float("nan"), dtype=torch.float32, device=DEV)
        probs.append((dC, A, lda, M * lda, M, Ntot, K, npg, slabs))
        refs.append(ref)
    ops.wgrad_multi(probs)
    torch.cuda.synchronize()
    for (M, Ntot, K, npg), pr, ref in zip(shapes, probs, refs):
        got = pr[8]
        assert torch.equal(got[:, :, :K], ref[:, :, :K]), (M, Ntot, K)


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_wgrad_reads_its_operand_rows_in_place_through_a_row_list(dtype):
    """gsage_wgrad_desc.a_rows: the A operand is a table read through a frontier's row list (duplicates, any
    order) -- bit-identical to the same launch over a gathered copy of those rows, for slices of 1, 2, 3 and many
    32-row steps per wave and ragged tails (every vmcnt path of the pipelined loop)."""
    T = torch.bfloat16 if dtype == "bf16" else torch.float32
    rng = np.random.RandomState(5)
    table_rows = 5000
    shapes = [(16384, 256, 602, 128), (1300, 256, 602, 128), (640, 128, 256, 128), (257, 32, 64, 32), (96, 16, 40, 16),
              (33, 8, 24, 8)]
    for wg_target in (None, 8, 64):
        probs, refs = [], []
        for (M, Ntot, K, npg) in shapes:
            lda = gs.store._round_up(K, 8)
            dC = torch.from_numpy(rng.normal(size=(M, Ntot)).astype(np.float32)).to(DEV).to(T)
            table = torch.zeros(table_rows, lda, dtype=T, device=DEV)
            table[:, :K] = torch.from_numpy(rng.normal(size=(table_rows, K)).astype(np.float32)).to(DEV).to(T)
            rows = torch.from_numpy(rng.randint(0, table_rows, size=M + 3)).to(DEV)[:M + 2]
            copy = table[rows[:M]].contiguous()
            rps, S, ldk = ops.wgrad_plan(M, Ntot, K, *([wg_target] if wg_target else []))
            ref = torch.full((S, Ntot, ldk), float("nan"), dtype=torch.float32, device=DEV)
            got = torch.full((S, Ntot, ldk), float("nan"), dtype=torch.float32, device=DEV)
            refs.append((copy, (dC, copy, lda, 0, M, Ntot, K, Ntot, ref, wg_target)))
            probs.append((dC, table, lda, 0, M, Ntot, K, Ntot, got, wg_target, rows))
        ops.wgrad_multi([r[1] for r in refs])
        ops.wgrad_multi(probs)
        torch.cuda.synchronize()
        for (M, Ntot, K, npg), pr, rf in zip(shapes, probs, refs):
            assert torch.equal(pr[8][:, :, :K], rf[1][8][:, :, :K]), (dtype, wg_target, M, Ntot, K)
            assert not torch.isnan(pr[8][:, :, :K]).any()


@pytest.mark.parametrize("variant", ["valu", "mfma"])
@pytest.mark.parametrize("B,n,C", [(512, 25, 41), (13, 7, 5), (6, 32, 64), (4, 1, 2), (33, 16, 41), (35, 17, 3),
                                   (16, 25, 17), (50, 25, 48), (48, 15, 41), (20, 10, 7)])
def test_seed_level_kernel_vs_fp64(B, n, C, variant):
    """gsage_mean_tail_ce / gsage_mean_tail_mfma = segment mean + both projections + normalize/fc/CE + every
    gradient down to the previous level, against fp64 torch autograd on the same bf16-rounded operands (ragged last
    workgroup, both neighbour-register variants, 1..64 classes) -- the VALU kernel (4 seeds per workgroup) and the
    matrix-core kernel (16 seeds per workgroup; its fan-out-25 specialisation and the generic one)."""
    import torch.nn.functional as F
    mfma = variant == "mfma"
    rng = np.random.RandomState(B * 131 + n * 7 + C)
    L = nat.lib()
    Hf = np.maximum(rng.normal(size=(B * (1 + n), 256)), 0).astype(np.float32)       # post-ReLU rows
    H = torch.from_numpy(Hf).to(DEV).bfloat16()
    w2 = torch.from_numpy((rng.normal(size=(2, 128, 256)) * 0.08).astype(np.float32)).to(DEV).bfloat16()
    w2t = w2.transpose(1, 2).contiguous()
    Wfc = torch.from_numpy((rng.normal(size=(C, 256)) * 0.3).astype(np.float32)).to(DEV)
    bfc = torch.from_numpy(rng.normal(size=(C,)).astype(np.float32)).to(DEV)
    tg = torch.from_numpy(rng.randint(0, C, size=B)).to(DEV)
    agg = torch.full((B, 256), 7.0, device=DEV, dtype=torch.bfloat16)
    dE = torch.full((B, 256), 7.0, device=DEV, dtype=torch.bfloat16)
    preds = torch.empty(B, C, device=DEV)
    dH = torch.full_like(H, 7.0)
    n_wg = (B + 15) // 16 if mfma else (B + 3) // 4
    part = torch.empty((L.gsage_mean_tail_mfma_scratch if mfma else L.gsage_mean_tail_ce_scratch)(B, C), device=DEV)
    assert part.numel() == n_wg * (C * 256 + C + 1)

    def run(outs, gather):
        args = (H.data_ptr(), B, n, w2.data_ptr(), 256, w2t.data_ptr(), 128, Wfc.data_ptr(), bfc.data_ptr(), C,
                tg.data_ptr(), None, 0, *[t.data_ptr() for t in outs], gather)
        if mfma:
            nat.check(L.gsage_mean_tail_mfma(*args, None), "mean_tail_mfma")
        else:
            nat.check(L.gsage_mean_tail_ce(*args, nat.BF16, None), "mean_tail_ce")
    run((agg, dE, preds, dH, part), None)
    torch.cuda.synchronize()
    if B == 512:
        # the same launch carrying a gather role on the CUs it leaves idle: every output of the seed
        # level unchanged, the gathered means those of gsage_gather_mean, bit for bit
        import ctypes
        rngg = np.random.RandomState(3)
        table = torch.zeros(5000, 640, dtype=torch.bfloat16, device=DEV)
        table[:, :602] = torch.from_numpy(rngg.normal(size=(5000, 602)).astype(np.float32)).to(DEV).bfloat16()
        store = gs.FeatureStore(table, 602)
        rows = 1234
        gids = torch.from_numpy(rngg.randint(0, 5000, size=rows * 10)).to(DEV)
        ref = ops.gather_mean(store, gids, rows, 10, out_dtype=torch.bfloat16, out_ld=640)
        out = torch.zeros(rows + 3, 640, dtype=torch.bfloat16, device=DEV)
        d = nat.TailGatherDesc()
        d.table, d.ids, d.out = table.data_ptr(), gids.data_ptr(), out.data_ptr()
        d.ld, d.out_ld, d.D, d.rows, d.n, d.n_workgroups = 640, 640, 602, rows, 10, (224 if mfma else 128)
        outs = [torch.zeros_like(t) for t in (agg, dE, preds, dH, part)]
        run(outs, ctypes.addressof(d))
        torch.cuda.synchronize()
        for a, b_ in zip((agg, dE, preds, dH, part), outs):
            assert torch.equal(a, b_)
        assert torch.equal(out[:rows], ref) and float(out[rows:].float().abs().max()) == 0.0
    Hd = H.double().cpu()
    x, nb = Hd[:B], Hd[B:].view(B, n, 256)
    mean = nb.mean(1)
    close(agg.float().cpu().numpy(), mean.numpy(), "agg", 2 ** -8, 2 ** -8)           # one bf16 rounding
    aggd = agg.double().cpu()                                                          # what the GEMM sees
    Wx, Wn = w2[0].double().cpu(), w2[1].double().cpu()
    emb = torch.cat([x @ Wx.t(), aggd @ Wn.t()], 1).requires_grad_(True)
    Wd, bd = Wfc.double().cpu().requires_grad_(True), bfc.double().cpu().requires_grad_(True)
    pr = F.normalize(emb, dim=1) @ Wd.t() + bd
    ls = F.cross_entropy(pr, tg.cpu())
    ls.backward()
    close(preds.cpu().numpy(), pr.detach().numpy(), "preds", 2e-5, 2e-5)
    scale = float(emb.grad.abs().max())
    assert float((dE.double().cpu() - emb.grad).abs().max()) <= 2 ** -8 * scale, "dE beyond one bf16 rounding"
    p3 = part.view(n_wg, C * 256 + C + 1).double().sum(0).cpu()
    close(p3[:C * 256].view(C, 256).numpy(), Wd.grad.numpy(), "d fc.weight", 2e-5, 1e-6)
    close(p3[C * 256:C * 256 + C].numpy(), bd.grad.numpy(), "d fc.bias", 2e-5, 1e-6)
    assert abs(float(p3[-1]) / B - float(ls)) < 1e-5 * max(1.0, float(ls))            # loss partials are sums
    # input gradients from the dE the kernel itself rounded to bf16 (that is what K5b consumes too)
    dEd = dE.double().cpu()
    dX = (dEd[:, :128] @ Wx) * (x > 0)
    dA = (dEd[:, 128:] @ Wn) / n
    ref = torch.cat([dX, (dA[:, None, :] * (nb > 0)).reshape(B * n, 256)], 0)
    got = dH.double().cpu()
    sc = max(float(ref.abs().max()), 1e-30)
    assert float((got - ref).abs().max()) <= 2 ** -8 * sc + 1e-12, "dH beyond one bf16 rounding"
    assert bool(((got == 0) == (ref == 0)).all()) or float((got - ref).abs().max()) <= 2 ** -8 * sc


@pytest.mark.parametrize("wd", [0.0, 1e-3])
def test_flat_adam_equals_torch_clip_plus_adam(wd):
    """optim.FlatAdam (Parameters as views of flat buckets, clip + Adam in two launches) against
    torch.nn.utils.clip_grad_norm_(params, 5) + torch.optim.Adam over 6 steps: gradients large enough
    to be clipped on some steps and not on others, a learning-rate change in between."""
    torch.manual_seed(0)
    shapes = [(37, 19), (128,), (5, 64, 3), (1,)]
    ref = [torch.nn.Parameter(torch.randn(*s, device=DEV)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    opt_r = torch.optim.Adam(ref, lr=0.01, weight_decay=wd)
    opt_m = gs.optim.FlatAdam(mine, lr=0.01, weight_decay=wd)
    assert opt_m.owns() and all(torch.equal(a, b) for a, b in zip(ref, mine))
    for step in range(6):
        scale = [0.01, 3.0, 0.2, 10.0, 0.001, 1.0][step]
        grads = [torch.randn(*s, device=DEV) * scale for s in shapes]
        opt_r.zero_grad()
        opt_m.zero_grad()
        for p, q, g in zip(ref, mine, grads):
            p.grad = g.clone()
            q.grad.add_(g)                                   # in place, as autograd accumulates
        if step == 3:
            gs.LRSchedule.set_lr(opt_r, 0.003)
            gs.LRSchedule.set_lr(opt_m, 0.003)
        total = torch.nn.utils.clip_grad_norm_(ref, 5)
        opt_r.step()
        opt_m.clip_and_step(5.0)
        assert abs(float(opt_m.grad_norm) - float(total)) <= 1e-5 * float(total)
        for p, q in zip(ref, mine):
            close(q.detach().cpu().numpy(), p.detach().cpu().numpy(), ("weights", step), 2e-6, 2e-7)
            close(q.grad.cpu().numpy(), p.grad.cpu().numpy(), ("clipped grad", step), 2e-6, 1e-7)
    assert int(opt_m.step_count) == 6


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
@pytest.mark.parametrize("M", [1, 15, 16, 17, 1000, 70001])
def test_attn_mlp_second_layer_kernels(dtype, M):
    """gsage_attn_mlp2_fwd / _bwd (the 32 -> 32 layer of the att MLP with its element-wise neighbours fused in)
    against torch on the same rounded operands: a = hid W2^T;  da = T(dan + dax), dhid = T((da W2) (1 - hid^2))."""
    T = torch.bfloat16 if dtype == "bf16" else torch.float32
    code = nat.BF16 if dtype == "bf16" else nat.F32
    gen = torch.Generator(device="cpu").manual_seed(M)
    lib, st = nat.lib(), ops._stream()
    hid = torch.zeros(M, 64, dtype=T, device=DEV)
    hid[:, :32] = torch.tanh(torch.randn(M, 32, generator=gen)).to(DEV).to(T)
    W2 = torch.zeros(32, 64, dtype=T, device=DEV)
    W2[:, :32] = (torch.randn(32, 32, generator=gen) * 0.3).to(DEV).to(T)
    W2T = torch.zeros(32, 64, dtype=T, device=DEV)
    W2T[:, :32] = W2[:, :32].t()
    a = torch.full((M, 32), float("nan"), device=DEV)
    nat.check(lib.gsage_attn_mlp2_fwd(hid.data_ptr(), code, 64, W2.data_ptr(), 64, a.data_ptr(), 32, M, 32, st), "fwd")
    ref = hid[:, :32].double() @ W2[:, :32].double().t()
    assert torch.allclose(a.double(), ref, rtol=1e-5, atol=1e-5)
    dan = torch.randn(M, 32, generator=gen).to(DEV)
    dax = torch.randn(M, 32, generator=gen).to(DEV)
    da = torch.zeros(M, 64, dtype=T, device=DEV)
    dhid = torch.zeros(M, 64, dtype=T, device=DEV)
    nat.check(lib.gsage_attn_mlp2_bwd(dan.data_ptr(), 32, dax.data_ptr(), 32, hid.data_ptr(), code, 64, W2T.data_ptr(), 64,
                                      da.data_ptr(), 64, dhid.data_ptr(), 64, M, 32, st), "bwd")
    torch.cuda.synchronize()
    da_ref = (dan + dax).to(T)
    assert torch.equal(da[:, :32], da_ref) and float(da[:, 32:].abs().max()) == 0.0
    dh_ref = (da_ref.double() @ W2[:, :32].double()) * (1 - hid[:, :32].double() ** 2)
    tol = 1e-2 if dtype == "bf16" else 1e-5
    assert torch.allclose(dhid[:, :32].double(), dh_ref, rtol=tol, atol=tol)


@pytest.mark.parametrize("B,D", [(512, 256), (48, 256), (7, 40), (2048, 128)])
@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_l1_head_with_the_reference_broadcast(B, D, dtype):
    """gsage_head_l1 against torch autograd of the reference's own expression: F.l1_loss(fc(F.normalize(E)),
    targets.squeeze()) -- [B,1] against [B], which broadcasts to [B,B] (problem.py:39-42 behind models.py:100)."""
    import warnings
    T = torch.bfloat16 if dtype == "bf16" else torch.float32
    gen = torch.Generator(device="cpu").manual_seed(B + D)
    E = torch.randn(B, D, generator=gen).to(DEV)
    W = (torch.randn(1, D, generator=gen) * 0.5).to(DEV)
    b = torch.randn(1, generator=gen).to(DEV)
    t = torch.randn(B, 1, generator=gen).to(DEV)
    preds = torch.full((B, 1), float("nan"), device=DEV)
    dE = torch.zeros(B, D, dtype=T, device=DEV)
    scratch = torch.full((nat.lib().gsage_head_l1_scratch(B, D),), float("nan"), device=DEV)
    nat.check(nat.lib().gsage_head_l1(E.data_ptr(), D, W.data_ptr(), b.data_ptr(), t.data_ptr(), B, D, preds.data_ptr(),
                                      dE.data_ptr(), nat.BF16 if dtype == "bf16" else nat.F32, D, scratch.data_ptr(),
                                      ops._stream()), "head_l1")
    n_wg = (B + 15) // 16
    part = scratch[:n_wg * (D + 2)].view(n_wg, D + 2).sum(0)            # what gsage_finalize_grads sums
    dwb, loss = part[:D + 1], part[D + 1]
    Ec, Wc, bc = [x.detach().double().cpu().requires_grad_(True) for x in (E, W, b)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        p_ref = torch.nn.functional.linear(torch.nn.functional.normalize(Ec, dim=1), Wc, bc)
        l_ref = torch.nn.functional.l1_loss(p_ref, t.double().cpu().squeeze())
    l_ref.backward()
    close(preds.cpu().numpy(), p_ref.detach().numpy(), "preds", 1e-5, 1e-5)
    assert abs(float(loss) - float(l_ref)) <= 1e-5 * max(1.0, float(l_ref))
    close(dwb[:D].cpu().numpy(), Wc.grad.numpy().reshape(-1), "d fc.weight", 1e-4, 1e-6)
    close(dwb[D:].cpu().numpy(), bc.grad.numpy(), "d fc.bias", 1e-4, 1e-6)
    tol = (2e-2, 1e-6) if dtype == "bf16" else (1e-4, 1e-7)
    close(dE.float().cpu().numpy(), Ec.grad.numpy(), "dE", *tol)
