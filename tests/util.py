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
