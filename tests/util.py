"""Shared helpers for the product tests."""
import numpy as np
import torch
from scipy import sparse
from torch.nn import functional as F


def csr_of(g, pre):
    return sparse.csr_matrix((g[pre + "data"], g[pre + "indices"], g[pre + "indptr"]),
                             shape=tuple(int(v) for v in g[pre + "shape"]))


def weights(g, prefix, device="cpu"):
    return {k[len(prefix):]: torch.from_numpy(g[k].copy()).to(device) for k in g.files
            if k.startswith(prefix)}


def close(a, b, what, rtol=1e-5, atol=2e-6):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    assert err <= (atol + rtol) * scale, (what, "max abs err %g at scale %g" % (err, scale))


def close_rel(a, b, what, tol):
    """max |a - b| <= tol * max |b|: element-wise, relative to the tensor's own scale (not to 1.0, so
    it stays meaningful for gradients and updates of magnitude 1e-2)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = float(np.abs(b).max())
    err = float(np.abs(a - b).max())
    assert err <= tol * scale + 1e-12, (what, "max abs err %g vs scale %g" % (err, scale))


def close_update(w, w_ref, w0, what, tol_fro=5e-3, tol_elem=1e-4):
    """Weights after Adam steps (lr = 0.01) against the reference's.  Adam's update
    lr * m / (sqrt(v) + 1e-8) is ill-conditioned exactly where a gradient entry (plus weight decay)
    cancels to ~1e-8 -- there an fp32 summation-order difference moves the weight by a fraction of lr
    -- so the check is (a) the UPDATE w - w0 matches in norm to tol_fro (a skipped or wrong optimizer
    step, a missing gradient term or a sign error changes it by O(1)), and (b) all but a handful of
    entries (1e-4 of the tensor, at least 2) match to tol_elem = 1 % of one step."""
    w, w_ref, w0 = (np.asarray(t, dtype=np.float64) for t in (w, w_ref, w0))
    assert w.shape == w_ref.shape == w0.shape, (what, w.shape, w_ref.shape)
    d, d_ref = w - w0, w_ref - w0
    err = float(np.linalg.norm(d - d_ref)) / max(float(np.linalg.norm(d_ref)), 1e-12)
    assert err <= tol_fro, (what, "update differs by %g of its norm" % err)
    bad = int((np.abs(w - w_ref) > tol_elem).sum())
    assert bad <= max(2, int(1e-4 * w.size)), (what, "%d of %d entries off by more than %g" % (bad, w.size, tol_elem))


def close_fro(a, b, what, tol):
    """Norm-wise check for low-precision gradients: a bf16 forward can flip the ReLU mask of a
    pre-activation that is ~0, which moves single gradient entries by O(1) -- bounded in
    Frobenius norm, not element-wise."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = float(np.linalg.norm(a - b)) / max(float(np.linalg.norm(b)), 1e-12)
    assert err <= tol, (what, "relative Frobenius error %g" % err)


class SelReplay(object):
    """Feeds recorded `sel` matrices to a compat-mode sampler (monkeypatches np.random.choice
    exactly where the reference, and the product in compat mode, draw them)."""

    def __init__(self, sels):
        self.sels = [np.asarray(s) for s in sels]
        self.orig = np.random.choice

    def __enter__(self):
        def fake(a, size=None, *args, **kw):
            s = self.sels.pop(0)
            assert tuple(s.shape) == tuple(size), (s.shape, size)
            return s.copy()
        np.random.choice = fake
        return self

    def __exit__(self, *exc):
        np.random.choice = self.orig


ACTS = {"relu": F.relu, "identity": (lambda x: x)}


def build_model(gs, g, p, device="cpu", feats_dtype=None):
    """GSSupervised for golden model case `p` with the recorded initial weights."""
    aggn, prepn, task, sched = [str(s) for s in g[p + "cfg"]]
    has_feats = bool(int(g[p + "has_feats"])) if (p + "has_feats") in g.files else True
    fan = [int(v) for v in g[p + "fanouts"]]
    odims = [int(v) for v in g[p + "out_dims"]]
    adj, tadj = csr_of(g, p + "adj_"), csr_of(g, p + "tadj_")
    feats = None
    if has_feats:
        if device == "cpu":
            feats = torch.from_numpy(g[p + "feats"].copy())
        else:
            feats = gs.FeatureStore.from_array(g[p + "feats"], torch.device(device),
                                               dtype=feats_dtype or gs.ops.config.compute_dtype)
    model = gs.GSSupervised(
        sampler_class=gs.sampler_lookup["sparse_uniform_neighbor_sampler"], adj=adj, train_adj=tadj,
        prep_class=gs.prep_lookup[prepn], aggregator_class=gs.aggregator_lookup[aggn],
        input_dim=g[p + "feats"].shape[1] if has_feats else None, n_nodes=adj.shape[0],
        n_classes=int(g[p + "n_classes"]),
        layer_specs=[{"n_train_samples": fan[l], "n_val_samples": fan[l], "output_dim": odims[l],
                      "activation": F.relu if l < len(fan) - 1 else (lambda x: x)} for l in range(len(fan))],
        lr_init=0.01, lr_schedule=sched, weight_decay=float(g[p + "weight_decay"]))
    model.load_state_dict(weights(g, p + "w0_"))
    model = model.to(device)
    # Adam state must live where the parameters are
    model.optimizer = torch.optim.Adam(model.parameters(), lr=model.lr,
                                       weight_decay=float(g[p + "weight_decay"]))
    return model, feats, task


# ---- small models for the data-parallel tests (tests/test_gpu_dist.py, tests/_ddp1_worker.py) ---------------------
DP_CASES = {
    # name: (aggregator, prep, task, output dims, fan-outs, storage precision)
    "mean": ("mean", "identity", "classification", (128, 128), (5, 3), "bf16"),
    "max_pool": ("max_pool", "identity", "classification", (128, 128), (5, 3), "bf16"),
    "attention": ("attention", "identity", "classification", (64, 64), (5, 3), "bf16"),
    "attention_emb": ("attention", "node_embedding", "classification", (64, 64), (4, 3), "fp32"),
    "attention_emb_mae": ("attention", "node_embedding", "regression_mae", (64, 64), (4, 3), "fp32"),
    "attention_emb_bf16": ("attention", "node_embedding", "regression_mae", (64, 64), (4, 3), "bf16"),
    "mean_emb": ("mean", "node_embedding", "classification", (64, 64), (4, 3), "fp32"),
}


def dp_case(gs, case, n_batch=3, global_batch=32, device="cuda"):
    """-> (model, feats or None, loss_fn, ids [n_batch, global_batch], targets, precision) of DP_CASES[case]: a 500-node
    graph with empty rows and duplicates, Philox sampler (seed 77), weight decay on."""
    agg, prep, task, dims, fan, prec = DP_CASES[case]
    rng = np.random.RandomState(0)
    n, D, C = 500, 40, 5
    deg = rng.randint(0, 30, size=n + 1)
    deg[0], deg[n] = 0, 3
    indptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    data = rng.randint(1, n + 1, size=int(indptr[-1]))
    adj = sparse.csr_matrix((data, gs.store.row_positions(indptr), indptr), shape=(n + 1, int(deg.max())))
    emb = prep == "node_embedding"
    feats = None
    if not emb:
        feats = rng.normal(size=(n + 1, D)).astype(np.float32)
        feats[0] = 0
    torch.manual_seed(5)
    gs.nn_modules.SparseUniformNeighborSampler.rng_default = "philox"
    specs = [{"n_train_samples": fan[0], "n_val_samples": fan[0], "output_dim": dims[0], "activation": F.relu},
             {"n_train_samples": fan[1], "n_val_samples": fan[1], "output_dim": dims[1], "activation": lambda x: x}]
    cls = task == "classification"
    model = gs.GSSupervised(sampler_class=gs.sampler_lookup["sparse_uniform_neighbor_sampler"], adj=adj,
                            train_adj=adj, prep_class=gs.prep_lookup[prep],
                            aggregator_class=gs.aggregator_lookup[agg], input_dim=None if emb else D, n_nodes=n + 1,
                            n_classes=C if cls else 1, layer_specs=specs, lr_init=0.01, weight_decay=1e-4)
    gs.nn_modules.SparseUniformNeighborSampler.rng_default = "compat"
    model.train_sampler.seed = model.val_sampler.seed = 77
    ids = torch.from_numpy(rng.randint(1, n + 1, size=(n_batch, global_batch)))
    if cls:
        tg = torch.from_numpy(rng.randint(0, C, size=(n_batch, global_batch, 1)))
    else:
        tg = torch.from_numpy(rng.normal(size=(n_batch, global_batch, 1)).astype(np.float32) * 2)
    loss_fn = gs.ProblemLosses.classification if cls else gs.ProblemLosses.regression_mae
    return model.to(device), feats, loss_fn, ids.to(device), tg.to(device), prec


def dp_run(gs, case, model, feats, loss_fn, ids, tg, prec, ddp, steps=4, capture=True):
    """`steps` train steps of the fused engine that covers the case (queue mode with the fused classification head,
    one call per batch otherwise) -> (predictions [steps, B, C], flat parameters with settled table rows, engine)."""
    gs.ops.set_compute_dtype(prec)
    gs.ops.warmup(torch.device("cuda"))
    store = gs.FeatureStore.from_array(feats, torch.device("cuda"), dtype=prec) if feats is not None else None
    cls = gs.engine.fused_engine_for(model, store, ddp=ddp)
    assert cls is not None, gs.engine.why_no_fused_engine(model, store, ddp)
    eng = cls(model, store, loss_fn, ids[0], tg[0], ddp=ddp, capture=capture)
    preds = []
    if eng.fused_head:
        eng.load_epoch(ids, tg)
        for _ in range(steps):
            preds.append(eng.step_queue().clone())
    else:
        for k in range(steps):
            preds.append(eng(ids[k % ids.shape[0]], tg[k % ids.shape[0]]).clone())
    eng.sync_rows()
    torch.cuda.synchronize()
    return torch.stack(preds).cpu(), eng.flat_p.clone().cpu(), eng
