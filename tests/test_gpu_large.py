"""-m gpu: maximum sizes (papers100M needs int64 row offsets and a feature
table far beyond 2^31 bytes).  Built directly on the device (10 GB CSR, 5 GB table: seconds on a
288 GB MI355X), checked through the definition: out = col[rowptr[id] + sel % deg]."""
import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu
gs = pkg()
ops, nat = gs.ops, gs._native
DEV = "cuda"


@pytest.fixture(scope="module")
def big_csr():
    if torch.cuda.get_device_properties(0).total_memory < 60 * 2 ** 30:
        pytest.skip("needs a large-memory GPU")
    n_rows = 1 << 20                                           # ids 0 .. 2^20-1 (0 = dummy, degree 0)
    deg = torch.full((n_rows,), 3000, dtype=torch.int64, device=DEV)
    deg[0] = 0
    deg[1::7] = 1                                              # ragged: some single-neighbour rows
    deg[5::11] = 0                                             # ... and some empty ones
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=DEV)
    rowptr[1:] = torch.cumsum(deg, 0)
    nnz = int(rowptr[-1])
    assert nnz > 2 ** 31, nnz
    col = torch.randint(1, n_rows, (nnz,), dtype=torch.int32, device=DEV)
    return gs.DeviceCSR(rowptr, col, n_rows, 4096), deg


def _check(csr, deg, ids, n, out, sel):
    ids_r = ids.repeat_interleave(n)
    d = deg[ids_r]
    pos = csr.rowptr[ids_r] + torch.where(d > 0, sel.long() % torch.clamp(d, min=1), torch.zeros_like(d))
    want = torch.where(d > 0, csr.col[torch.clamp(pos, max=csr.nnz - 1)].long(), torch.zeros_like(d))
    assert torch.equal(out, want)


def test_sampler_beyond_2_31_edges(big_csr):
    csr, deg = big_csr
    n_rows = csr.n_rows
    g = torch.Generator(device="cpu").manual_seed(0)
    # seeds from the END of the graph: their row offsets are > 2^31
    ids = (n_rows - 1 - torch.randint(0, 5000, (777,), generator=g)).to(DEV)
    assert int(csr.rowptr[ids].min()) > 2 ** 31
    sel_out = torch.empty(777 * 13, dtype=torch.int32, device=DEV)
    out = ops.sample_csr(csr, ids, 13, philox={"seed": 99, "call_base": 4, "sel_out": sel_out})
    csr.check()
    _check(csr, deg, ids, 13, out, sel_out)
    assert int(out.max()) < n_rows and int(out.min()) >= 0


def test_fused_hops_beyond_2_31_edges(big_csr):
    """3 hops in one launch over the same graph == three per-hop launches (papers100M fan-out 15/10/5)."""
    import ctypes
    csr, deg = big_csr
    B, fans = 64, (15, 10, 5)
    g = torch.Generator(device="cpu").manual_seed(1)
    seeds = (csr.n_rows - 1 - torch.randint(0, 100000, (B,), generator=g)).to(DEV)
    sizes = [B]
    for f in fans:
        sizes.append(sizes[-1] * f)
    all_ids = torch.zeros(sum(sizes), dtype=torch.int64, device=DEV)
    all_ids[:B] = seeds
    fan = (ctypes.c_int32 * 3)(*fans)
    nat.check(nat.lib().gsage_sample_hops_philox(csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.n_rows,
                                                 all_ids.data_ptr(), B, 3, fan, csr.max_deg, 5, None, 0, 0,
                                                 None, None, 0, csr.err_flag.data_ptr(), None), "hops")
    csr.check()
    cur, off = seeds, B
    for k, f in enumerate(fans):
        sel_out = torch.empty(cur.numel() * f, dtype=torch.int32, device=DEV)
        ref = ops.sample_csr(csr, cur, f, philox={"seed": 5, "call_base": k, "sel_out": sel_out})
        _check(csr, deg, cur, f, ref, sel_out)
        assert torch.equal(all_ids[off:off + ref.numel()], ref)
        cur, off = ref, off + ref.numel()


def test_gather_rows_beyond_2_31_bytes():
    """K2 addresses rows with 64-bit arithmetic: a 5 GB bf16 table, rows from its far end."""
    if torch.cuda.get_device_properties(0).total_memory < 60 * 2 ** 30:
        pytest.skip("needs a large-memory GPU")
    R, D = 20_000_000, 128                                     # 5.1 GB
    table = torch.empty(R, D, dtype=torch.bfloat16, device=DEV)
    table.view(torch.int16).copy_((torch.arange(R, device=DEV, dtype=torch.int64) % 251).to(torch.int16).view(-1, 1)
                                  .expand(R, D))               # every row filled with a row-dependent bit pattern
    store = gs.FeatureStore(table, D)
    g = torch.Generator(device="cpu").manual_seed(2)
    ids = (R - 1 - torch.randint(0, 1000, (4096 * 5,), generator=g)).to(DEV)
    assert int(ids.min()) * D * 2 > 2 ** 31
    rows = ops.gather_mean(store, ids, 4096 * 5, 1, out_dtype=torch.bfloat16)
    assert torch.equal(rows.view(torch.int16), table[ids].view(torch.int16))       # n == 1: exact copy
    mean = ops.gather_mean(store, ids, 4096, 5, out_dtype=torch.float32)
    want = table[ids].float().view(4096, 5, D).mean(1)
    assert float((mean - want).abs().max()) <= 1e-6 * float(want.abs().max() + 1)
