"""
oracle/torch_ref.py -- fp32 CPU ORACLE of the floating-point half of the hot path
(test infrastructure, NOT product; only tests/, smoke() and bench.py's cpu_baseline import it).

A functional, plain-torch restatement (weights are passed in as a dict with the reference's
state_dict key names) of:
  mean_aggregator        nn_modules.py:196-204
  pool_aggregator        nn_modules.py:223-232 (+ :235-256 for the max / mean pool_fn)
  attention_aggregator   nn_modules.py:305-321
  prep_*                 nn_modules.py:112-166
  forward                models.py:71-91      (frontier, gather, layer stacking, normalize, fc)
  loss_*                 problem.py:26-38
  clip / adam            models.py:97-104 -> torch.nn.utils.clip_grad_norm(5) + optim.Adam
Parity status: PINNED against tests/golden/{agg,prep,model}_kat.npz (tests/test_oracle_float.py),
tolerance 1e-5 relative (fp32, different summation order only).

The sampler step inside `forward` uses oracle/cpu.py (C) with an explicit `sel`, so the whole
forward is a deterministic function of (weights, ids, sels).

`rounding="bf16"` (mean / pool aggregators) additionally rounds to bf16 exactly where the fused
engines store bf16 -- gathered rows, neighbour means, hidden-level outputs, the weight operand copies,
and (through straight-through hooks) the gradients written between levels -- while every sum stays
fp32/fp64 as in the kernels.  With rounding=None this file is the pinned fp32 oracle; the bf16 mode is
the same code plus `_rb` calls, and lets the production (bf16) instantiation of the engines be checked
to ~1e-3 instead of the ~3e-2 an fp32 oracle allows.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import cpu as ocpu


class _RoundBF16(torch.autograd.Function):
    """value -> bf16 -> fp32 in the forward, the same rounding on the gradient in the backward
    (the engines store both the activation and its gradient in bf16)."""

    @staticmethod
    def forward(ctx, x, round_grad):
        ctx.round_grad = round_grad
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return (g.to(torch.bfloat16).to(g.dtype) if ctx.round_grad else g), None


def _rb(x, rounding, round_grad=False):
    return _RoundBF16.apply(x, round_grad) if rounding == "bf16" else x


def _act(name):
    return {"relu": torch.relu, "identity": (lambda t: t)}[name]


def _segments(x, neibs):
    return neibs.reshape(x.shape[0], -1, neibs.shape[1])


def _combine(x, agg, w, act, rounding=None):
    wx, wn = _rb(w["fc_x.weight"], rounding), _rb(w["fc_neib.weight"], rounding)
    out = torch.cat([x @ wx.t(), agg @ wn.t()], dim=1)
    return _act(act)(out)


def mean_aggregator(x, neibs, w, act, rounding=None):
    # the mean is a stored bf16 operand of the projection, and its gradient comes back in fp32
    return _combine(x, _rb(_segments(x, neibs).mean(dim=1), rounding), w, act, rounding)


def pool_aggregator(x, neibs, w, act, pool, rounding=None):
    pre = neibs @ _rb(w["mlp.0.weight"], rounding).t()
    if rounding == "bf16":
        # the hidden layer never leaves the chip in the forward; its GRADIENT is stored in bf16 as the
        # dC operand of the MLP's weight gradient and of its input gradient -- the bias gradient is
        # summed from the unrounded fp32 values (gsage_pool_bias_partials / _mean)
        pre = _GradRound.apply(pre)
    h = torch.relu(pre + w["mlp.0.bias"])
    seg = _segments(x, h)
    agg = seg.max(dim=1)[0] if pool == "max" else seg.mean(dim=1)
    return _combine(x, _rb(agg, rounding), w, act, rounding)


class _GradRound(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class _TanhStoredBF16(torch.autograd.Function):
    """tanh whose OUTPUT is stored in bf16 and whose backward uses that stored value, the result again stored
    in bf16 -- what engine.FusedAttnTrainStep does with the att MLP's hidden layer (K5 with fused tanh writes
    bf16; gsage_tanh_bwd reads it back and writes the bf16 operand of K5 / K5b)."""

    @staticmethod
    def forward(ctx, x):
        y = torch.tanh(x).to(torch.bfloat16).to(x.dtype)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return (g * (1 - y * y)).to(torch.bfloat16).to(g.dtype)


def att_mlp(t, w, rounding=None):
    """att(.) of nn_modules.py:293-297: Linear(D, 32, no bias) -> tanh -> Linear(32, 32, no bias)."""
    if rounding == "bf16":
        hid = _TanhStoredBF16.apply(t @ _rb(w["att.0.weight"], rounding).t())
        return _GradRound.apply(hid @ _rb(w["att.2.weight"], rounding).t())      # d a is stored in bf16
    return torch.tanh(t @ w["att.0.weight"].t()) @ w["att.2.weight"].t()


def attention_weighting(x, neibs, xa, na, w, act, rounding=None):
    """nn_modules.py:309-317 given att(x), att(neibs)."""
    scores = torch.bmm(_segments(x, na), xa.unsqueeze(2)).squeeze(2)      # [M, n]  (n>1, M>1: same as .squeeze())
    ws = torch.softmax(scores, dim=1)
    agg = (_segments(x, neibs) * ws.unsqueeze(-1)).sum(dim=1)
    return _combine(x, _rb(agg, rounding), w, act, rounding)


def attention_aggregator(x, neibs, w, act):
    return attention_weighting(x, neibs, att_mlp(x, w), att_mlp(neibs, w), w, act)


def aggregator(name, x, neibs, w, act, rounding=None):
    if name == "mean":
        return mean_aggregator(x, neibs, w, act, rounding)
    if name == "max_pool":
        return pool_aggregator(x, neibs, w, act, "max", rounding)
    if name == "mean_pool":
        return pool_aggregator(x, neibs, w, act, "mean", rounding)
    if name == "attention":
        assert rounding is None, "with rounding points the attention levels go through forward() (att once per row)"
        return attention_aggregator(x, neibs, w, act)
    raise KeyError(name)


def prep(name, ids, feats, w, n_nodes, layer_idx):
    if name == "identity":
        return feats
    if name == "linear":
        return feats @ w["fc.weight"].t()
    if name == "node_embedding":
        # seeds (layer_idx 0) all read the extra row `n_nodes`, never their own (nn_modules.py:145-149)
        rows = ids if layer_idx > 0 else torch.full_like(ids, n_nodes)
        e = w["embedding.weight"][rows] @ w["fc.weight"].t() + w["fc.bias"]
        return e if feats is None else torch.cat([feats, e], dim=1)
    raise KeyError(name)


def split_weights(w):
    """state_dict of GSSupervised -> (prep dict, [layer dicts], fc dict)."""
    prep_w = {k[len("prep."):]: v for k, v in w.items() if k.startswith("prep.")}
    layers = []
    li = 0
    while any(k.startswith("agg_layers.%d." % li) for k in w):
        pre = "agg_layers.%d." % li
        layers.append({k[len(pre):]: v for k, v in w.items() if k.startswith(pre)})
        li += 1
    fc = {"weight": w["fc.weight"], "bias": w["fc.bias"]}
    return prep_w, layers, fc


def forward(w, ids, feats, indptr, data, fanouts, sels, agg_name, prep_name, n_nodes,
            acts=None, rounding=None, frontier=None):
    """models.py:71-91 with the sampler's `sel` supplied per hop.  acts: one activation name per
    layer (default: the layer_specs of train.py:105-118 generalised in depth as models.py:85-86
    allows -- ReLU on every layer but the last).  frontier (optional): the sampled ids of every hop
    [hop 1, hop 2, ...] given directly instead of (indptr, data, sels) -- for graphs too large to
    hand to the CPU, whose samples are checked against the sampler's definition separately."""
    prep_w, layers, fc = split_weights(w)
    if acts is None:
        acts = ["relu"] * (len(layers) - 1) + ["identity"]
    assert len(acts) == len(layers) == len(fanouts)
    ids = torch.as_tensor(ids, dtype=torch.long)
    take = (lambda i: feats[i]) if feats is not None else (lambda i: None)
    hs = [prep(prep_name, ids, take(ids), prep_w, n_nodes, 0)]
    cur = ids
    for hop, n in enumerate(fanouts):
        if frontier is not None:
            nxt = np.asarray(frontier[hop], dtype=np.int64).reshape(-1)
            assert nxt.shape[0] == cur.shape[0] * int(n)
        else:
            nxt = ocpu.sample_csr_sel(indptr, data, cur.numpy(), int(n), sels[hop])
        cur = torch.from_numpy(nxt)
        hs.append(prep(prep_name, cur, take(cur), prep_w, n_nodes, hop + 1))
    for li, lw in enumerate(layers):
        if agg_name == "attention" and rounding is not None:
            # like the engine: att(.) ONCE per row of the level (the reference applies the same MLP to a row as
            # "x" and as a neighbour: same values), so that a row's two gradient contributions to att(.) are
            # summed in fp32 and rounded once
            sizes = [h.shape[0] for h in hs]
            a_all = torch.split(att_mlp(torch.cat(hs, dim=0), lw, rounding), sizes, dim=0)
            hs = [attention_weighting(hs[k], hs[k + 1], a_all[k], a_all[k + 1], lw, acts[li], rounding)
                  for k in range(len(hs) - 1)]
            if li < len(layers) - 1:
                hs = [_rb(h, rounding, round_grad=True) for h in hs]
            continue
        hs = [aggregator(agg_name, hs[k], hs[k + 1], lw, acts[li], rounding) for k in range(len(hs) - 1)]
        if li < len(layers) - 1:
            # hidden levels are stored (and their gradients written) in bf16; the last level stays fp32
            hs = [_rb(h, rounding, round_grad=True) for h in hs]
    assert len(hs) == 1
    emb = hs[0]
    if rounding == "bf16":
        emb = _GradRound.apply(emb)       # the embedding stays fp32, its gradient is stored in bf16
    out = F.normalize(emb, p=2, dim=1, eps=1e-12)
    return out @ fc["weight"].t() + fc["bias"]


def loss(task, preds, targets):
    t = targets.squeeze()
    if task == "classification":
        return F.cross_entropy(preds, t)
    if task == "multilabel_classification":
        return F.multilabel_soft_margin_loss(preds, t)
    if task == "regression_mae":
        return F.l1_loss(preds, t)
    raise KeyError(task)


def clip_(grads, max_norm=5.0):
    """torch.nn.utils.clip_grad_norm (models.py:101): global L2 norm, coef = max/(norm+1e-6),
    applied only when < 1."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = max_norm / (float(total) + 1e-6)
    if coef < 1.0:
        for g in grads.values():
            g.mul_(coef)
    return float(total)


class Adam(object):
    """torch.optim.Adam(betas=(0.9,0.999), eps=1e-8, L2 weight_decay) as used at models.py:69."""

    def __init__(self, weight_decay=0.0):
        self.t = 0
        self.m = {}
        self.v = {}
        self.wd = weight_decay

    def step(self, w, grads, lr):
        self.t += 1
        b1, b2, eps = 0.9, 0.999, 1e-8
        for k, g in grads.items():
            p = w[k]
            if self.wd:
                g = g + self.wd * p
            m = self.m.setdefault(k, torch.zeros_like(p))
            v = self.v.setdefault(k, torch.zeros_like(p))
            m.mul_(b1).add_(g, alpha=1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = v.sqrt() / np.sqrt(1 - b2 ** self.t) + eps
            p.addcdiv_(m, denom, value=-lr / (1 - b1 ** self.t))


def train_step(w, opt, lr, task, ids, feats, targets, indptr, data, fanouts, sels, agg_name,
               prep_name, n_nodes, acts=None, rounding=None, frontier=None):
    """models.py:97-104.  `w` is updated in place.  Returns dict(preds, loss, gradnorm, grads,
    clipped)."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in w.items()}
    preds = forward(params, ids, feats, indptr, data, fanouts, sels, agg_name, prep_name, n_nodes,
                    acts=acts, rounding=rounding, frontier=frontier)
    l = loss(task, preds, targets)
    l.backward()
    grads = {k: p.grad.detach().clone() for k, p in params.items() if p.grad is not None}
    raw = {k: g.clone() for k, g in grads.items()}
    gn = clip_(grads)
    with torch.no_grad():
        opt.step(w, grads, lr)
    return {"preds": preds.detach(), "loss": float(l.detach()), "gradnorm": gn, "grads": raw,
            "clipped": grads}
