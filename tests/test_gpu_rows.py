"""-m gpu: deferred ("lazy") Adam over the rows of a trainable embedding table (gsage_rows_*, include/gsage.h).
The dense update the reference performs (torch.optim.Adam over nn.Embedding.weight: EVERY row moves every step)
is replayed row by row; the claim is bit-identity with the dense kernel, so the checks are torch.equal."""
import ctypes
import os

import numpy as np
import pytest
import torch
from scipy import sparse
from torch.nn import functional as F

from conftest import pkg

pytestmark = pytest.mark.gpu
gs = pkg()
ops, nat = gs.ops, gs._native
DEV = "cuda"


@pytest.fixture(autouse=True)
def _setup():
    ops.set_compute_dtype("bf16")
    ops.warmup(torch.device(DEV))
    yield
    gs.nn_modules.SparseUniformNeighborSampler.rng_default = "compat"
    ops.set_compute_dtype("bf16")
    os.environ.pop("GSAGE_DENSE_TABLE_ADAM", None)


class _Rows(object):
    def __init__(self, p, E, wd, max_norm, cap=64):
        n_rows = p.shape[0]
        z = lambda dt=torch.float32: torch.zeros(n_rows, E, dtype=dt, device=DEV)
        self.p, self.g, self.m, self.v = p.clone(), z(), z(), z()
        self.last = torch.zeros(n_rows, dtype=torch.int32, device=DEV)
        self.seen = torch.zeros(n_rows, dtype=torch.int32, device=DEV)
        self.hist = torch.zeros(2 * cap, dtype=torch.float32, device=DEV)
        self.lr = torch.zeros(1, dtype=torch.float32, device=DEV)
        self.step = torch.zeros(1, dtype=torch.int64, device=DEV)
        self.partial = torch.zeros(64, dtype=torch.float32, device=DEV)
        d = self.d = nat.RowAdamDesc()
        d.p, d.g, d.m, d.v = self.p.data_ptr(), self.g.data_ptr(), self.m.data_ptr(), self.v.data_ptr()
        d.last, d.seen, d.hist = self.last.data_ptr(), self.seen.data_ptr(), self.hist.data_ptr()
        d.lr, d.step, d.n_rows, d.E, d.hist_cap = self.lr.data_ptr(), self.step.data_ptr(), n_rows, E, cap
        d.beta1, d.beta2, d.eps, d.weight_decay, d.max_norm = 0.9, 0.999, 1e-8, wd, max_norm


@pytest.mark.parametrize("E,wd,max_norm", [(64, 0.0, 3e38), (64, 0.01, 3e38), (24, 0.0, 3e38), (200, 0.001, 3e38),
                                           (64, 0.0, 0.5)])
def test_deferred_rows_equal_the_dense_update_bit_for_bit(E, wd, max_norm):
    """12 updates over a 3 000-row table; every update has gradients on a random id list WITH duplicates (plus an
    always-present row, like the spare row every seed reads), the learning rate changes every step.  Dense side:
    gsage_clip_adam_step over the whole table.  Deferred side: catch-up of the listed rows (which must then equal
    the dense rows -- the forward reads them), norm over the listed rows, row update; rows never listed catch up
    at the end.  Without clipping everything is bit-identical; with clipping the norm's summation order differs,
    so the clipped case is held to 1e-6."""
    lib, st = nat.lib(), torch.cuda.current_stream().cuda_stream
    n_rows = 3000
    gen = torch.Generator(device="cpu").manual_seed(E)
    p0 = torch.randn(n_rows, E, generator=gen).to(DEV)
    lz = _Rows(p0, E, wd, max_norm)
    dn_p, dn_g = p0.clone(), torch.zeros(n_rows, E, device=DEV)
    dn_m, dn_v = torch.zeros_like(dn_g), torch.zeros_like(dn_g)
    dn_step = torch.zeros(1, dtype=torch.int64, device=DEV)
    dn_partial = torch.zeros(lib.gsage_adam_partials(n_rows * E), dtype=torch.float32, device=DEV)
    spare = torch.tensor([n_rows - 1], dtype=torch.int64, device=DEV)
    exact = max_norm > 1e30
    for t in range(1, 13):
        lr = 0.01 * (1.0 + 0.1 * t)
        lz.lr.fill_(lr)
        n_ids = int(torch.randint(1, 400, (1,), generator=gen))
        ids = torch.randint(0, 300 if t % 3 else n_rows, (n_ids,), generator=gen).to(DEV)
        ids = torch.cat([ids, ids[: n_ids // 3]])                       # duplicates
        uniq = torch.unique(torch.cat([ids, spare]))
        grad = torch.randn(uniq.shape[0], E, generator=gen).to(DEV) * (0.02 if t % 4 else 2.0)
        # ---- deferred side: rows the forward is about to read
        nat.check(lib.gsage_rows_catch_up(ctypes.byref(lz.d), spare.data_ptr(), 1, ids.data_ptr(), ids.shape[0], 0, st), "cu")
        if exact:
            assert torch.equal(lz.p[uniq], dn_p[uniq]), t
        else:
            assert torch.allclose(lz.p[uniq], dn_p[uniq], rtol=0, atol=1e-6), t
        lz.g[uniq] = grad
        dn_g[uniq] = grad
        lz.step += 1                                                      # (the engines' finalisation ticks it)
        nat.check(lib.gsage_rows_sqnorm(ctypes.byref(lz.d), spare.data_ptr(), 1, ids.data_ptr(), ids.shape[0], 0,
                                        lz.partial.data_ptr(), 64, st), "sq")
        assert abs(float(lz.partial.sum()) - float((grad.double() ** 2).sum())) <= 1e-5 * float((grad.double() ** 2).sum())
        nat.check(lib.gsage_rows_adam(ctypes.byref(lz.d), spare.data_ptr(), 1, ids.data_ptr(), ids.shape[0], 0,
                                      lz.partial.data_ptr(), 64, st), "ra")
        assert float(lz.g.abs().max()) == 0.0                             # consumed rows are zeroed
        # ---- dense side
        nat.check(lib.gsage_clip_adam_step(dn_p.data_ptr(), dn_g.data_ptr(), dn_m.data_ptr(), dn_v.data_ptr(), n_rows * E,
                                           dn_partial.data_ptr(), lz.lr.data_ptr(), dn_step.data_ptr(), 0.9, 0.999, 1e-8,
                                           wd, max_norm, None, 2 | 4, 0, None, 0, None, 0, None, 0, st), "dense")
        dn_g.zero_()
    assert int(lz.last.min()) < 12                                        # some rows are still behind
    nat.check(lib.gsage_rows_catch_up_all(ctypes.byref(lz.d), 0, st), "all")
    torch.cuda.synchronize()
    assert int(lz.last.min()) == 12 and int(lz.last.max()) == 12
    for a, b, name in ((lz.p, dn_p, "p"), (lz.m, dn_m, "exp_avg"), (lz.v, dn_v, "exp_avg_sq")):
        if exact:
            assert torch.equal(a, b), (name, float((a - b).abs().max()))
        else:
            assert torch.allclose(a, b, rtol=1e-6, atol=1e-6), (name, float((a - b).abs().max()))


def _emb_model(seed=0, n=3000, B=32):
    rng = np.random.RandomState(seed)
    deg = rng.randint(1, 12, size=n + 1)
    deg[0] = 0
    indptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    data = rng.randint(1, n, size=int(indptr[-1]))
    adj = sparse.csr_matrix((data, gs.store.row_positions(indptr), indptr), shape=(n + 1, int(deg.max())))
    torch.manual_seed(3)
    gs.nn_modules.SparseUniformNeighborSampler.rng_default = "philox"
    specs = [{"n_train_samples": 4, "n_val_samples": 4, "output_dim": 64, "activation": F.relu},
             {"n_train_samples": 3, "n_val_samples": 3, "output_dim": 64, "activation": lambda x: x}]
    model = gs.GSSupervised(sampler_class=gs.sampler_lookup["sparse_uniform_neighbor_sampler"], adj=adj, train_adj=adj,
                            prep_class=gs.prep_lookup["node_embedding"], aggregator_class=gs.aggregator_lookup["attention"],
                            input_dim=None, n_nodes=n, n_classes=1, layer_specs=specs, lr_init=0.01,
                            weight_decay=0.0).to(DEV)
    gs.nn_modules.SparseUniformNeighborSampler.rng_default = "compat"
    ids = torch.from_numpy(rng.randint(1, n, size=(10, B))).to(DEV)
    tg = torch.from_numpy(rng.normal(size=(10, B, 1)).astype(np.float32)).to(DEV)
    return model, ids, tg


@pytest.mark.parametrize("capture", [False, "cmdlist"])
def test_attention_embedding_engine_deferred_rows_equal_dense_table_updates(capture):
    """FusedAttnTrainStep over a trainable table, ten steps on ten different batches: deferred row updates
    (default) against the dense clip + Adam over the whole table (GSAGE_DENSE_TABLE_ADAM=1).  Same predictions
    every step; after state_dict() (which settles the deferred rows) the same table, exp_avg and exp_avg_sq.
    (The two sides differ in the summation order of the table's share of the gradient norm, and the scatter-add
    uses float atomics: 1e-6, not bit-identity.)"""
    res = {}
    for mode in ("dense", "deferred", "sorted"):
        os.environ.pop("GSAGE_SORTED_ROWS", None)
        if mode == "dense":
            os.environ["GSAGE_DENSE_TABLE_ADAM"] = "1"
        else:
            os.environ.pop("GSAGE_DENSE_TABLE_ADAM", None)
        if mode == "sorted":          # gradient rows by sort + segment sums (ABI 4; what data-parallel runs use)
            os.environ["GSAGE_SORTED_ROWS"] = "1"
        ops.set_compute_dtype("fp32")
        model, ids, tg = _emb_model()
        eng = gs.engine.FusedAttnTrainStep(model, None, gs.ProblemLosses.regression_mae, ids[0], tg[0], capture=capture)
        os.environ.pop("GSAGE_SORTED_ROWS", None)
        assert eng.emb and eng.lazy_rows == (mode != "dense") and (mode == "dense" or eng.sorted_rows == (mode == "sorted"))
        preds = []
        for s in range(10):
            eng.set_progress(s / 10.0)
            preds.append(eng(ids[s], tg[s]).detach().clone())
        if mode != "dense":
            assert int(eng.row_last.min()) < 10 and int(eng.row_last.max()) == 10     # rows ARE behind
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        if mode != "dense":
            assert int(eng.row_last.min()) == 10                                      # state_dict settled them
        nt = eng.n_tab
        res[mode] = (preds, sd, eng.flat_m[:nt].clone(), eng.flat_v[:nt].clone(), float(eng.gnorm.item()))
    a = res["dense"]
    for other in ("deferred", "sorted"):
        b = res[other]
        for s in range(10):
            assert torch.allclose(a[0][s], b[0][s], rtol=1e-5, atol=1e-5), (other, s)
        for k in a[1]:
            # (two DENSE runs differ by up to 3e-6 in the table: atomics order, amplified where Adam's v is tiny)
            assert torch.allclose(a[1][k], b[1][k], rtol=1e-5, atol=2e-5), (other, k, float((a[1][k] - b[1][k]).abs().max()))
        assert torch.allclose(a[2], b[2], rtol=1e-5, atol=1e-7) and torch.allclose(a[3], b[3], rtol=1e-5, atol=1e-9)
        assert abs(a[4] - b[4]) <= 1e-5 * max(1.0, a[4])
    moved = (a[1]["prep.embedding.weight"] != _emb_model()[0].state_dict()["prep.embedding.weight"]).any(dim=1)
    assert int(moved.sum()) > 500                      # Adam's moments keep moving rows after their last gradient


def test_module_forward_settles_deferred_rows():
    """Anything that reads the table through the module sees current rows: the eval forward after engine steps
    equals the eval forward of the dense run."""
    outs = []
    for mode in ("dense", "deferred"):
        if mode == "dense":
            os.environ["GSAGE_DENSE_TABLE_ADAM"] = "1"
        else:
            os.environ.pop("GSAGE_DENSE_TABLE_ADAM", None)
        ops.set_compute_dtype("fp32")
        model, ids, tg = _emb_model(seed=2)
        eng = gs.engine.FusedAttnTrainStep(model, None, gs.ProblemLosses.regression_mae, ids[0], tg[0], capture="cmdlist")
        for s in range(6):
            eng(ids[s], tg[s])
        rows = torch.arange(1, 3000, device=DEV)
        outs.append(model.prep.embedding(rows).detach().clone())
    assert torch.allclose(outs[0], outs[1], rtol=1e-5, atol=2e-5)


def test_model_forward_reads_settled_rows_after_deferred_engine_steps():
    """The evaluation path of train.py (`model(ids, feats, train=False)`, reference train.py:29-36) never calls the
    nn.Embedding module: NodeEmbeddingPrep reads `embedding.weight` itself.  After engine steps with deferred rows
    the model's OWN forward must therefore settle them (GSSupervised.forward) -- compared with the dense-table run
    on the same validation draws, over ids that include rows no training frontier touched."""
    outs, behind = [], None
    for mode in ("dense", "deferred"):
        if mode == "dense":
            os.environ["GSAGE_DENSE_TABLE_ADAM"] = "1"
        else:
            os.environ.pop("GSAGE_DENSE_TABLE_ADAM", None)
        ops.set_compute_dtype("fp32")
        model, ids, tg = _emb_model(seed=8)
        eng = gs.engine.FusedAttnTrainStep(model, None, gs.ProblemLosses.regression_mae, ids[0], tg[0], capture="cmdlist")
        for s in range(6):
            eng(ids[s], tg[s])
        if mode == "deferred":
            behind = int((eng.row_last < 6).sum())
            assert behind > 100                                  # most rows have pending zero-gradient updates
        model.val_sampler.seed, model.val_sampler.calls = 77, 0
        model.eval()
        ev = torch.arange(1, 257, device=DEV)
        with torch.no_grad():
            outs.append(model(ev, None, train=False).detach().clone())
        if mode == "deferred":
            assert int(eng.row_last.min()) == 6                  # the forward settled every row
    assert torch.allclose(outs[0], outs[1], rtol=1e-4, atol=1e-4), float((outs[0] - outs[1]).abs().max())


def test_close_settles_rows_and_detaches_and_a_second_engine_starts_from_current_rows():
    """engine A trains (rows deferred) -> engine B is built on the same model WITHOUT an explicit sync: B must
    start from A's settled table.  close() removes the hooks and refuses further steps."""
    ops.set_compute_dtype("fp32")
    model, ids, tg = _emb_model(seed=4)
    a = gs.engine.FusedAttnTrainStep(model, None, gs.ProblemLosses.regression_mae, ids[0], tg[0], capture=False)
    for s in range(4):
        a(ids[s], tg[s])
    assert int(a.row_last.min()) < 4
    b = gs.engine.FusedAttnTrainStep(model, None, gs.ProblemLosses.regression_mae, ids[0], tg[0], capture=False)
    assert int(a.row_last.min()) == 4                  # building B settled A's rows first
    assert torch.equal(b.table, a.table)
    a.close()
    assert not model.prep.embedding._forward_pre_hooks or len(model.prep.embedding._forward_pre_hooks) == 1
    with pytest.raises(RuntimeError):
        a(ids[0], tg[0])
    b(ids[5], tg[5])
    b.close()
    assert not hasattr(model, "_settle_rows") and len(model.prep.embedding._forward_pre_hooks) == 0


def test_ring_of_step_constants_wraps_safely(monkeypatch):
    """The constants of past updates live in a ring (ROW_HIST slots); the engine settles every row before a slot a
    deferred row still needs is overwritten.  With an 8-slot ring and 20 steps the table must still equal the dense
    run's."""
    outs = []
    for mode in ("dense", "deferred"):
        if mode == "dense":
            os.environ["GSAGE_DENSE_TABLE_ADAM"] = "1"
        else:
            os.environ.pop("GSAGE_DENSE_TABLE_ADAM", None)
            monkeypatch.setattr(gs.engine.FusedAttnTrainStep, "ROW_HIST", 8)
        ops.set_compute_dtype("fp32")
        model, ids, tg = _emb_model(seed=6)
        eng = gs.engine.FusedAttnTrainStep(model, None, gs.ProblemLosses.regression_mae, ids[0], tg[0], capture="cmdlist")
        settles = 0
        for s in range(20):
            before = eng._rows_since if mode == "deferred" else 0
            eng(ids[s % 10], tg[s % 10])
            if mode == "deferred" and eng._rows_since <= before:
                settles += 1
        if mode == "deferred":
            assert settles >= 2 and eng.row_hist.numel() == 16
        outs.append({k: v.detach().clone() for k, v in model.state_dict().items()})
    for k in outs[0]:
        # (20 steps: the scatter-add's float atomics land in a different order every run, and Adam amplifies that
        #  where v is tiny -- two DENSE runs differ by up to ~4e-5 in single table entries)
        assert torch.allclose(outs[0][k], outs[1][k], rtol=1e-5, atol=1e-4), (k, float((outs[0][k] - outs[1][k]).abs().max()))
