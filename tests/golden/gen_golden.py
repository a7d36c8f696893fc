#!/usr/bin/env python
"""
gen_golden.py -- golden-vector generator (TEST INFRASTRUCTURE, runs ONLY in the build container).

Imports the *reference* (bkj/pytorch-graphsage, mounted read-only at /root/reference) under
Python 3 with the shims listed in SURVEY.md section 8(c), drives its own classes on small seeded
inputs and stores inputs + outputs as .npz fixtures next to this script.  The fixtures are DATA
(arrays only); no reference source or bytecode is written anywhere (`sys.dont_write_bytecode`).

    python -B tests/golden/gen_golden.py          # regenerates tests/golden/*.npz

The reference has no tests / golden vectors of its own, so these fixtures are
what pins both the oracle (oracle/) and the HIP path.  Nothing here is imported by the product.

Reference entry points exercised (file:line in /root/reference):
  SparseUniformNeighborSampler  nn_modules.py:52-101      UniformNeighborSampler  nn_modules.py:19-49
  Mean/MaxPool/MeanPool/Attention aggregators              nn_modules.py:185-321
  IdentityPrep / NodeEmbeddingPrep / LinearPrep            nn_modules.py:112-166
  GSSupervised.forward / train_step                        models.py:71-104
  NodeProblem.iterate                                      problem.py:141-153
  ProblemLosses / ProblemMetrics                           problem.py:26-64
  LRSchedule                                               lr.py:11-42
  set_seeds                                                helpers.py:14-18
"""
import os
import sys
import types
import warnings

sys.dont_write_bytecode = True
REF = os.environ.get("GSAGE_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
for _m in ("h5py", "cPickle", "ujson"):          # absent py2-era deps; never called here
    sys.modules.setdefault(_m, types.ModuleType(_m))

import numpy as np
import torch
from torch.nn import functional as F
from scipy import sparse

import helpers          # noqa: E402  (reference)
import nn_modules       # noqa: E402  (reference)
import models           # noqa: E402  (reference)
import problem          # noqa: E402  (reference)
import lr as ref_lr     # noqa: E402  (reference)

# helpers.to_numpy recurses forever on torch>=0.4 (helpers.py:22-23): harness-side shim only.
_to_numpy = lambda x: x.detach().cpu().numpy()
helpers.to_numpy = _to_numpy
nn_modules.to_numpy = _to_numpy

warnings.filterwarnings("ignore")
OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(1)


# ----------------------------------------------------------------------------- graph builders
def make_ref_csr(n_nodes, degs, rng):
    """CSR in the reference's sparse convention (utils/convert.py:100-126): node ids 1-based,
    row 0 = dummy (empty), row i holds neighbour ids (1-based) in columns 0..deg_i-1."""
    degs = np.asarray(degs).astype(np.int64)
    assert degs.shape[0] == n_nodes + 1 and degs[0] == 0
    rows = np.repeat(np.arange(n_nodes + 1), degs)
    cols = np.concatenate([np.arange(d) for d in degs]) if degs.sum() else np.zeros(0, int)
    vals = rng.randint(1, n_nodes + 1, size=rows.shape[0])
    return sparse.csr_matrix((vals, (rows, cols)))


def graph_small(rng):
    n = 50
    degs = rng.randint(0, 9, size=n + 1)
    degs[0] = 0
    degs[3] = 0          # a real node with no neighbours -> samples the dummy 0
    degs[4] = 1          # degree-1 row
    degs[5] = 8          # max-degree row, power of two
    degs[n] = 2          # keep the last row non-empty so shape[0] == n+1
    return make_ref_csr(n, degs, rng)


def graph_mid(rng):
    n = 300
    degs = rng.randint(0, 22, size=n + 1)
    degs[0] = 0
    degs[7] = 21         # max degree 21: not a power of two -> rejection sampling rejects
    degs[9] = 0
    degs[n] = 1
    return make_ref_csr(n, degs, rng)


def graph_big(rng):
    n = 5000
    degs = np.clip(np.exp(rng.normal(2.5, 1.2, size=n + 1)).astype(int), 0, 700)
    degs[0] = 0
    degs[11] = 700
    degs[n] = 3
    return make_ref_csr(n, degs, rng)


class ChoiceRecorder(object):
    """Records what np.random.choice returns inside the reference sampler (nn_modules.py:88)."""

    def __init__(self):
        self.orig = np.random.choice
        self.calls = []

    def __enter__(self):
        def wrapped(a, size=None, *args, **kw):
            out = self.orig(a, size, *args, **kw)
            self.calls.append((int(a), np.array(out, copy=True)))
            return out
        np.random.choice = wrapped
        return self

    def __exit__(self, *exc):
        np.random.choice = self.orig


class ChoiceReplayer(object):
    """Feeds a prescribed `sel` sequence to the reference sampler instead of the numpy stream."""

    def __init__(self, sels):
        self.orig = np.random.choice
        self.sels = list(sels)

    def __enter__(self):
        def wrapped(a, size=None, *args, **kw):
            s = self.sels.pop(0)
            assert tuple(s.shape) == tuple(size), (s.shape, size)
            assert s.max() < a
            return np.array(s, copy=True)
        np.random.choice = wrapped
        return self

    def __exit__(self, *exc):
        np.random.choice = self.orig


def csr_arrays(adj, prefix):
    return {
        prefix + "indptr": adj.indptr.astype(np.int64),
        prefix + "data": adj.data.astype(np.int64),
        prefix + "indices": adj.indices.astype(np.int64),
        prefix + "shape": np.array(adj.shape, dtype=np.int64),
    }


# ----------------------------------------------------------------------------- A. sampler KATs
def gen_sampler():
    out = {}
    graphs = [graph_small(np.random.RandomState(0)), graph_mid(np.random.RandomState(1)),
              graph_big(np.random.RandomState(2))]
    case = 0
    for gi, adj in enumerate(graphs):
        out.update(csr_arrays(adj, "g%d_" % gi))
        sampler = nn_modules.SparseUniformNeighborSampler(adj=adj)
        out["g%d_degrees" % gi] = sampler.degrees.astype(np.int64)
        idrng = np.random.RandomState(100 + gi)
        for seed in (0, 123, 123 ** 2):
            for n in (1, 5, 10, 25):
                M = int(idrng.choice([2, 7, 33, 129]))
                ids = idrng.randint(0, adj.shape[0], size=M)
                ids[0] = 0 if M > 2 else ids[0]                       # the dummy node itself
                ids[-1] = {0: 3, 1: 9, 2: 11}[gi]                     # degree-0 / max-degree rows
                ids[1] = {0: 5, 1: 7, 2: 11}[gi]
                np.random.seed(seed)
                with ChoiceRecorder() as rec:
                    res = sampler(torch.LongTensor(ids), n_samples=n)
                assert len(rec.calls) == 1
                tail = np.random.randint(0, 2 ** 31 - 1, size=4)       # pins words consumed
                p = "c%d_" % case
                out[p + "graph"] = np.array(gi)
                out[p + "seed"] = np.array(seed)
                out[p + "n"] = np.array(n)
                out[p + "ids"] = ids.astype(np.int64)
                out[p + "sel"] = rec.calls[0][1].astype(np.int64)
                out[p + "out"] = _to_numpy(res).reshape(-1).astype(np.int64)
                out[p + "tail"] = tail.astype(np.int64)
                case += 1
    out["n_cases"] = np.array(case)
    np.savez_compressed(os.path.join(OUT, "sampler_kat.npz"), **out)
    print("sampler_kat: %d cases" % case)


# ----------------------------------------------------------------------------- B/C. stream + iterate
def fake_problem(n_total, n_train, task, n_classes, rng):
    p = object.__new__(problem.NodeProblem)
    folds = np.array(["test"] * n_total, dtype=object)
    idx = rng.permutation(np.arange(1, n_total))
    folds[idx[:n_train]] = "train"
    folds[idx[n_train:n_train + n_total // 5]] = "val"
    folds[0] = "dummy"
    p.folds = folds
    p.task = task
    p.cuda = False
    if task == "classification":
        p.targets = rng.randint(0, n_classes, size=(n_total, 1))
    elif task == "multilabel_classification":
        p.targets = rng.randint(0, 2, size=(n_total, n_classes))
    else:
        p.targets = rng.normal(30, 8, size=(n_total, 1))
    p.nodes = {k: np.where(folds == k)[0] for k in ("train", "val", "test")}
    return p


def gen_iterate():
    out = {}
    case = 0
    for (n_total, n_train, bs, seed) in [(1200, 1030, 512, 15129), (1100, 1024, 512, 0),
                                        (300, 77, 16, 123), (64, 33, 512, 5), (400, 256, 128, 9)]:
        p = fake_problem(n_total, n_train, "classification", 7, np.random.RandomState(seed + 1))
        for shuffle in (True, False):
            helpers.set_seeds(seed)
            ids_l, tg_l, pr_l, sz_l = [], [], [], []
            for ids, targets, prog in p.iterate(mode="train", batch_size=bs, shuffle=shuffle):
                ids_l.append(_to_numpy(ids))
                tg_l.append(_to_numpy(targets).reshape(-1))
                pr_l.append(prog)
                sz_l.append(ids.shape[0])
            tail = np.random.randint(0, 2 ** 31 - 1, size=4)
            pre = "c%d_" % case
            out[pre + "nodes"] = p.nodes["train"].astype(np.int64)
            out[pre + "targets_all"] = p.targets.astype(np.int64)
            out[pre + "bs"] = np.array(bs)
            out[pre + "seed"] = np.array(seed)
            out[pre + "shuffle"] = np.array(int(shuffle))
            out[pre + "ids"] = np.concatenate(ids_l).astype(np.int64)
            out[pre + "targets"] = np.concatenate(tg_l).astype(np.int64)
            out[pre + "progress"] = np.array(pr_l, dtype=np.float64)
            out[pre + "sizes"] = np.array(sz_l, dtype=np.int64)
            out[pre + "tail"] = tail.astype(np.int64)
            case += 1
    out["n_cases"] = np.array(case)
    np.savez_compressed(os.path.join(OUT, "iterate_kat.npz"), **out)
    print("iterate_kat: %d cases" % case)


def gen_stream():
    """set_seeds(seed**2) -> iterate(shuffle=True) -> per chunk two sparse-sampler calls, exactly
    the RNG interleaving of train.py:133-148 + models.py:73-81."""
    out = {}
    adj = graph_mid(np.random.RandomState(1))
    n_total = adj.shape[0]
    p = fake_problem(n_total, 150, "classification", 5, np.random.RandomState(4))
    sampler = nn_modules.SparseUniformNeighborSampler(adj=adj)
    out.update(csr_arrays(adj, "g_"))
    out["nodes"] = p.nodes["train"].astype(np.int64)
    seed = 123
    helpers.set_seeds(seed ** 2)
    k = 0
    for ids, targets, prog in p.iterate(mode="train", batch_size=64, shuffle=True):
        with ChoiceRecorder() as rec:
            h1 = sampler(ids, n_samples=5).contiguous().view(-1)
            h2 = sampler(h1, n_samples=3).contiguous().view(-1)
        out["b%d_ids" % k] = _to_numpy(ids).astype(np.int64)
        out["b%d_h1" % k] = _to_numpy(h1).astype(np.int64)
        out["b%d_h2" % k] = _to_numpy(h2).astype(np.int64)
        out["b%d_sel1" % k] = rec.calls[0][1].astype(np.int64)
        out["b%d_sel2" % k] = rec.calls[1][1].astype(np.int64)
        k += 1
    out["n_batches"] = np.array(k)
    out["seed"] = np.array(seed)
    out["tail"] = np.random.randint(0, 2 ** 31 - 1, size=4).astype(np.int64)
    np.savez_compressed(os.path.join(OUT, "stream_kat.npz"), **out)
    print("stream_kat: %d batches" % k)


# ----------------------------------------------------------------------------- D. aggregators
def sd_arrays(module, prefix):
    return {prefix + k: _to_numpy(v).copy() for k, v in module.state_dict().items()}


def gen_aggregators():
    out = {}
    case = 0
    acts = {"relu": F.relu, "identity": (lambda x: x)}
    for name in ("mean", "max_pool", "mean_pool", "attention"):
        for (M, n, D, h, act) in [(6, 5, 20, 8, "relu"), (4, 1, 12, 6, "identity"),
                                  (33, 10, 50, 16, "relu"), (9, 25, 70, 32, "identity"),
                                  (17, 3, 16, 128, "relu")]:
            if name == "attention" and n == 1:
                continue      # .squeeze() quirk: reference errors out
            torch.manual_seed(1000 + case)
            agg = nn_modules.aggregator_lookup[name](input_dim=D, output_dim=h,
                                                     activation=acts[act])
            x = torch.randn(M, D, requires_grad=True)
            neibs = torch.randn(M * n, D, requires_grad=True)
            res = agg(x, neibs)
            G = torch.randn_like(res)
            (res * G).sum().backward()
            p = "c%d_" % case
            out[p + "name"] = np.array(name)
            out[p + "act"] = np.array(act)
            out[p + "dims"] = np.array([M, n, D, h], dtype=np.int64)
            out[p + "output_dim"] = np.array(agg.output_dim)
            out.update(sd_arrays(agg, p + "w_"))
            out[p + "x"] = _to_numpy(x).copy()
            out[p + "neibs"] = _to_numpy(neibs).copy()
            out[p + "G"] = _to_numpy(G).copy()
            out[p + "out"] = _to_numpy(res).copy()
            out[p + "dx"] = _to_numpy(x.grad).copy()
            out[p + "dneibs"] = _to_numpy(neibs.grad).copy()
            for k, v in agg.named_parameters():
                out[p + "g_" + k] = _to_numpy(v.grad).copy()
            case += 1
    out["n_cases"] = np.array(case)
    np.savez_compressed(os.path.join(OUT, "agg_kat.npz"), **out)
    print("agg_kat: %d cases" % case)


# ----------------------------------------------------------------------------- E. preps
def gen_preps():
    out = {}
    case = 0
    for name, input_dim in (("identity", 10), ("node_embedding", 10), ("node_embedding", None),
                            ("linear", 10)):
        for layer_idx in (0, 1):
            torch.manual_seed(50 + case)
            n_nodes = 40
            prep = nn_modules.prep_lookup[name](input_dim=input_dim, n_nodes=n_nodes)
            ids = torch.LongTensor(np.random.RandomState(case).randint(0, n_nodes, size=13))
            feats = torch.randn(13, input_dim) if input_dim else None
            res = prep(ids, feats, layer_idx=layer_idx)
            p = "c%d_" % case
            out[p + "name"] = np.array(name)
            out[p + "input_dim"] = np.array(input_dim if input_dim else 0)
            out[p + "n_nodes"] = np.array(n_nodes)
            out[p + "layer_idx"] = np.array(layer_idx)
            out[p + "output_dim"] = np.array(prep.output_dim)
            out[p + "ids"] = _to_numpy(ids)
            if feats is not None:
                out[p + "feats"] = _to_numpy(feats)
            out.update(sd_arrays(prep, p + "w_"))
            out[p + "out"] = _to_numpy(res).copy()
            if any(True for _ in prep.parameters()):
                G = torch.randn_like(res)
                (res * G).sum().backward()
                out[p + "G"] = _to_numpy(G)
                for k, v in prep.named_parameters():
                    out[p + "g_" + k] = _to_numpy(v.grad).copy()
            case += 1
    out["n_cases"] = np.array(case)
    np.savez_compressed(os.path.join(OUT, "prep_kat.npz"), **out)
    print("prep_kat: %d cases" % case)


# ----------------------------------------------------------------------------- F. full model
def gen_models():
    out = {}
    case = 0
    cfgs = [
        # (aggregator, prep, task, has_feats, fanouts, out_dims, lr_schedule, weight_decay)
        ("mean", "identity", "classification", True, (5, 3), (16, 16), "constant", 0.0),
        ("mean", "identity", "classification", True, (25, 10), (8, 8), "linear", 0.0),
        ("max_pool", "identity", "classification", True, (5, 3), (16, 16), "constant", 5e-4),
        ("mean_pool", "identity", "multilabel_classification", True, (4, 2), (8, 8), "constant", 0.0),
        ("attention", "node_embedding", "regression_mae", False, (4, 3), (16, 16), "constant", 0.0),
        ("mean", "node_embedding", "regression_mae", True, (5, 2), (8, 8), "constant", 0.0),
        ("attention", "identity", "classification", True, (5, 3), (16, 16), "constant", 0.0),
    ]
    for (aggn, prepn, task, has_feats, fan, odims, sched, wd) in cfgs:
        grng = np.random.RandomState(300 + case)
        n = 120
        degs = grng.randint(0, 12, size=n + 1)
        degs[0] = 0
        degs[2] = 0
        degs[n] = 3
        adj = make_ref_csr(n, degs, grng)
        tdegs = np.minimum(degs, grng.randint(0, 9, size=n + 1))
        tdegs[n] = 2
        train_adj = make_ref_csr(n, tdegs, grng)
        n_rows = adj.shape[0]
        D = 12
        n_classes = {"classification": 5, "multilabel_classification": 4, "regression_mae": 1}[task]
        feats_np = grng.normal(size=(n_rows, D)).astype(np.float32)
        feats_np[0] = 0
        feats = torch.FloatTensor(feats_np) if has_feats else None

        torch.manual_seed(7 + case)
        np.random.seed(7 + case)
        model = models.GSSupervised(**{
            "sampler_class": nn_modules.sampler_lookup["sparse_uniform_neighbor_sampler"],
            "adj": adj, "train_adj": train_adj,
            "prep_class": nn_modules.prep_lookup[prepn],
            "aggregator_class": nn_modules.aggregator_lookup[aggn],
            "input_dim": D if has_feats else None,
            "n_nodes": n_rows,
            "n_classes": n_classes,
            "layer_specs": [
                {"n_train_samples": fan[0], "n_val_samples": fan[0], "output_dim": odims[0],
                 "activation": F.relu},
                {"n_train_samples": fan[1], "n_val_samples": fan[1], "output_dim": odims[1],
                 "activation": lambda x: x},
            ],
            "lr_init": 0.01, "lr_schedule": sched, "weight_decay": wd,
        })
        p = "c%d_" % case
        out[p + "cfg"] = np.array([aggn, prepn, task, sched])
        out[p + "has_feats"] = np.array(int(has_feats))
        out[p + "fanouts"] = np.array(fan)
        out[p + "out_dims"] = np.array(odims)
        out[p + "weight_decay"] = np.array(wd)
        out[p + "n_classes"] = np.array(n_classes)
        out[p + "feats"] = feats_np
        out.update(csr_arrays(adj, p + "adj_"))
        out.update(csr_arrays(train_adj, p + "tadj_"))
        out.update(sd_arrays(model, p + "w0_"))

        B = 11
        ids = torch.LongTensor(grng.randint(1, n_rows, size=B))
        if task == "classification":
            targets = torch.LongTensor(grng.randint(0, n_classes, size=(B, 1)))
        elif task == "multilabel_classification":
            targets = torch.FloatTensor(grng.randint(0, 2, size=(B, n_classes)).astype(np.float32))
        else:
            targets = torch.FloatTensor(grng.normal(30, 8, size=(B, 1)).astype(np.float32))
        loss_fn = getattr(problem.ProblemLosses, task)
        out[p + "ids"] = _to_numpy(ids)
        out[p + "targets"] = _to_numpy(targets)

        # ---- eval forward (val sampler, full adj) with recorded sel
        np.random.seed(99)
        with ChoiceRecorder() as rec:
            ev = model(ids, feats, train=False)
        out[p + "eval_sel0"] = rec.calls[0][1].astype(np.int64)
        out[p + "eval_sel1"] = rec.calls[1][1].astype(np.int64)
        out[p + "eval_preds"] = _to_numpy(ev).copy()

        # ---- two train steps, recorded sel; progress set as train.py:142 would
        np.random.seed(1234 + case)
        for step in range(2):
            model.set_progress(0.25 * step)
            out[p + "lr%d" % step] = np.array(model.lr)
            w_before = {k: v.clone() for k, v in model.state_dict().items()}
            with ChoiceRecorder() as rec:
                preds = model.train_step(ids=ids, feats=feats, targets=targets, loss_fn=loss_fn)
            sels = [c[1] for c in rec.calls]
            out[p + "s%d_sel0" % step] = sels[0].astype(np.int64)
            out[p + "s%d_sel1" % step] = sels[1].astype(np.int64)
            out[p + "s%d_preds" % step] = _to_numpy(preds).copy()
            for k, v in model.named_parameters():
                out[p + "s%d_cg_%s" % (step, k)] = _to_numpy(v.grad).copy()      # clipped grads
            out.update(sd_arrays(model, p + "w%d_" % (step + 1)))
            # replay the same sel on the pre-step weights to get loss + unclipped grad norm
            w_after = {k: v.clone() for k, v in model.state_dict().items()}
            model.load_state_dict(w_before)
            model.optimizer.zero_grad()
            with ChoiceReplayer(sels):
                pr2 = model(ids, feats, train=True)
            loss = loss_fn(pr2, targets.squeeze())
            loss.backward()
            tn = torch.sqrt(sum((q.grad.detach() ** 2).sum() for q in model.parameters()
                                if q.grad is not None))
            assert np.allclose(_to_numpy(pr2), _to_numpy(preds), atol=1e-6)
            out[p + "s%d_loss" % step] = np.array(float(loss))
            out[p + "s%d_gradnorm" % step] = np.array(float(tn))
            for k, v in model.named_parameters():
                out[p + "s%d_g_%s" % (step, k)] = _to_numpy(v.grad).copy()       # raw grads
            model.load_state_dict(w_after)
        case += 1
    out["n_cases"] = np.array(case)
    np.savez_compressed(os.path.join(OUT, "model_kat.npz"), **out)
    print("model_kat: %d cases" % case)


# ----------------------------------------------------------------------------- G/H. metrics, losses, lr
def gen_misc():
    out = {}
    rng = np.random.RandomState(11)
    y = rng.randint(0, 6, size=(200, 1))
    logits = rng.normal(size=(200, 6)).astype(np.float32)
    m = problem.ProblemMetrics.classification(y, logits)
    out["cls_y"], out["cls_logits"] = y, logits
    out["cls_micro"], out["cls_macro"] = np.array(m["micro"]), np.array(m["macro"])
    out["cls_loss"] = np.array(float(problem.ProblemLosses.classification(
        torch.FloatTensor(logits), torch.LongTensor(y).squeeze())))
    y = rng.randint(0, 2, size=(150, 9))
    logits = rng.normal(size=(150, 9)).astype(np.float32)
    m = problem.ProblemMetrics.multilabel_classification(y, logits)
    out["ml_y"], out["ml_logits"] = y, logits
    out["ml_micro"], out["ml_macro"] = np.array(m["micro"]), np.array(m["macro"])
    out["ml_loss"] = np.array(float(problem.ProblemLosses.multilabel_classification(
        torch.FloatTensor(logits), torch.FloatTensor(y.astype(np.float32)))))
    y = rng.normal(30, 8, size=(90, 1)).astype(np.float32)
    pr = (y + rng.normal(size=(90, 1))).astype(np.float32)
    out["mae_y"], out["mae_pred"] = y, pr
    out["mae"] = np.array(problem.ProblemMetrics.regression_mae(y, pr))
    out["mae_loss"] = np.array(float(problem.ProblemLosses.regression_mae(
        torch.FloatTensor(pr), torch.FloatTensor(y))))
    xs = np.array([0.0, 0.1, 0.5, 0.99, 1.0, 1.5, 2.25])
    out["lr_x"] = xs
    for name in ("constant", "linear", "cyclical"):
        fn = getattr(ref_lr.LRSchedule, name)
        out["lr_" + name] = np.array([fn(float(x), lr_init=0.01) for x in xs])
        out["lr_%s_e4" % name] = np.array([fn(float(x), lr_init=0.05, epochs=4) for x in xs])
    np.savez_compressed(os.path.join(OUT, "misc_kat.npz"), **out)
    print("misc_kat done")


# ----------------------------------------------------------------------------- I. dense sampler
def gen_dense_sampler():
    out = {}
    rng = np.random.RandomState(3)
    n, K = 60, 16
    adj = rng.randint(0, n, size=(n + 1, K))
    adj[n] = n
    s = nn_modules.UniformNeighborSampler(adj=torch.LongTensor(adj))
    out["adj"] = adj.astype(np.int64)
    case = 0
    for seed in (0, 123, 15129):
        for ns in (1, 5, 16):
            ids = rng.randint(0, n + 1, size=9)
            helpers.set_seeds(seed)
            res = s(torch.LongTensor(ids), n_samples=ns)
            p = "c%d_" % case
            out[p + "seed"], out[p + "n"] = np.array(seed), np.array(ns)
            out[p + "ids"] = ids.astype(np.int64)
            out[p + "out"] = _to_numpy(res).astype(np.int64)
            case += 1
    out["n_cases"] = np.array(case)
    np.savez_compressed(os.path.join(OUT, "dense_sampler_kat.npz"), **out)
    print("dense_sampler_kat: %d cases" % case)


if __name__ == "__main__":
    gen_sampler()
    gen_iterate()
    gen_stream()
    gen_aggregators()
    gen_preps()
    gen_models()
    gen_misc()
    gen_dense_sampler()
    with open(os.path.join(OUT, "VERSIONS.txt"), "w") as f:
        f.write("python %s\ntorch %s\nnumpy %s\n" % (sys.version.split()[0], torch.__version__,
                                                      np.__version__))
        import scipy, sklearn
        f.write("scipy %s\nsklearn %s\n" % (scipy.__version__, sklearn.__version__))
