#!/usr/bin/env python
"""
gen_golden_round3.py -- golden vectors for what round 3 added to the fused engines (TEST INFRASTRUCTURE, runs
ONLY in the build container; same harness and shims as gen_golden.py, which imports the reference).

  p0, p1   the reference's utils/pokec.sh:5-8 configuration -- MEAN aggregators over the trainable node-embedding
           prep, NO features, regression_mae (F.l1_loss with the [B,1]-vs-[B] broadcast), sparse sampler -- two
           train steps of the reference's GSSupervised.train_step (models.py:97-104) with the `sel` its sampler
           drew (nn_modules.py:88), every weight incl. the embedding table after each step.
  d0, d1   the reference's DEFAULT sampler (UniformNeighborSampler, nn_modules.py:19-49; run.sh:8-10) under the
           mean aggregator: two train steps with the torch.randperm each sampler call drew (nn_modules.py:44),
           the sampled frontier of step 0, predictions, clipped gradients, weights after two Adam updates.
  k0       one epoch of the reference's training loop bookkeeping for the dense sampler: set_seeds(seed ** 2)
           (train.py:133), iterate(shuffle=True) (problem.py:141-153), per chunk the two sampler calls -- the
           chunk ids, both hops of the frontier and the state torch's CPU generator is left in.

    python -B tests/golden/gen_golden_round3.py      # writes tests/golden/round3_kat.npz
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np
import torch
from torch.nn import functional as F

import gen_golden as gg          # imports the reference with the harness shims

models, nn_modules, problem, helpers = gg.models, gg.nn_modules, gg.problem, gg.helpers
_np = gg._to_numpy


class PermRecorder(object):
    """Records what torch.randperm returns inside the reference's dense sampler (nn_modules.py:44)."""

    def __init__(self):
        self.orig = torch.randperm
        self.calls = []

    def __enter__(self):
        def wrapped(n, *args, **kw):
            out = self.orig(n, *args, **kw)
            self.calls.append(out.clone().numpy())
            return out
        torch.randperm = wrapped
        return self

    def __exit__(self, *exc):
        torch.randperm = self.orig


def two_steps(out, p, model, ids, feats, targets, loss_fn, recorder, record_key):
    """the reference's train_step twice; records the sampler's draws, predictions, clipped gradients of step 0,
    loss + pre-clip gradient norm (replayed on the pre-step weights), weights after each step"""
    for step in range(2):
        model.set_progress(0.25 * step)
        out[p + "lr%d" % step] = np.array(model.lr)
        w_before = {k: v.clone() for k, v in model.state_dict().items()}
        gen_state = torch.get_rng_state()
        with recorder() as rec:
            preds = model.train_step(ids=ids, feats=feats, targets=targets, loss_fn=loss_fn)
        draws = [c[1] if isinstance(c, tuple) else c for c in rec.calls]
        for h, dv in enumerate(draws):
            out[p + "s%d_%s%d" % (step, record_key, h)] = np.asarray(dv).astype(np.int32)
        out[p + "s%d_preds" % step] = _np(preds).copy()
        if step == 0:
            for k, v in model.named_parameters():
                out[p + "s0_cg_%s" % k] = _np(v.grad).copy()                 # clipped grads
        w_after = {k: v.clone() for k, v in model.state_dict().items()}
        out.update(gg.sd_arrays(model, p + "w%d_" % (step + 1)))
        # loss and pre-clip gradient norm: the same draws on the pre-step weights
        model.load_state_dict(w_before)
        model.optimizer.zero_grad()
        if record_key == "sel":
            with gg.ChoiceReplayer(draws):
                pr2 = model(ids, feats, train=True)
        else:
            after = torch.get_rng_state()
            torch.set_rng_state(gen_state)
            pr2 = model(ids, feats, train=True)
            torch.set_rng_state(after)
        loss = loss_fn(pr2, targets.squeeze())
        loss.backward()
        tn = torch.sqrt(sum((q.grad.detach() ** 2).sum() for q in model.parameters() if q.grad is not None))
        assert np.allclose(_np(pr2), _np(preds), atol=1e-6)
        out[p + "s%d_loss" % step] = np.array(float(loss))
        out[p + "s%d_gradnorm" % step] = np.array(float(tn))
        model.load_state_dict(w_after)


def gen_pokec_mean(out):
    cfgs = [((5, 3), (16, 16), 0.0, 1.0), ((4, 2), (64, 64), 5e-4, 0.05)]
    for case, (fan, odims, wd, fscale) in enumerate(cfgs):
        grng = np.random.RandomState(900 + case)
        n = 140
        degs = grng.randint(0, 12, size=n + 1)
        degs[0], degs[2], degs[n] = 0, 0, 3
        adj = gg.make_ref_csr(n, degs, grng)
        tdegs = np.minimum(degs, grng.randint(0, 9, size=n + 1))
        tdegs[n] = 2
        train_adj = gg.make_ref_csr(n, tdegs, grng)
        n_rows = adj.shape[0]
        torch.manual_seed(60 + case)
        np.random.seed(60 + case)
        model = models.GSSupervised(**{
            "sampler_class": nn_modules.sampler_lookup["sparse_uniform_neighbor_sampler"],
            "adj": adj, "train_adj": train_adj,
            "prep_class": nn_modules.prep_lookup["node_embedding"],
            "aggregator_class": nn_modules.aggregator_lookup["mean"],
            "input_dim": None, "n_nodes": n_rows, "n_classes": 1,
            "layer_specs": [{"n_train_samples": fan[0], "n_val_samples": fan[0], "output_dim": odims[0],
                             "activation": F.relu},
                            {"n_train_samples": fan[1], "n_val_samples": fan[1], "output_dim": odims[1],
                             "activation": lambda x: x}],
            "lr_init": 0.01, "lr_schedule": "constant", "weight_decay": wd,
        })
        with torch.no_grad():
            for prm in model.agg_layers.parameters():
                prm.mul_(fscale)
        p = "p%d_" % case
        out[p + "cfg"] = np.array(["mean", "node_embedding", "regression_mae", "constant"])
        out[p + "has_feats"] = np.array(0)
        out[p + "fanouts"], out[p + "out_dims"] = np.array(fan), np.array(odims)
        out[p + "weight_decay"], out[p + "n_classes"] = np.array(wd), np.array(1)
        out[p + "feats"] = np.zeros((n_rows, 1), dtype=np.float32)          # (unused: the problem has no features)
        out.update(gg.csr_arrays(adj, p + "adj_"))
        out.update(gg.csr_arrays(train_adj, p + "tadj_"))
        out.update(gg.sd_arrays(model, p + "w0_"))
        B = 11
        ids = torch.LongTensor(grng.randint(1, n_rows, size=B))
        ids[0] = 2                                   # a seed without neighbours: samples the dummy node
        targets = torch.FloatTensor(grng.normal(30, 8, size=(B, 1)).astype(np.float32))
        out[p + "ids"], out[p + "targets"] = _np(ids), _np(targets)
        np.random.seed(5555 + case)
        two_steps(out, p, model, ids, None, targets, problem.ProblemLosses.regression_mae, gg.ChoiceRecorder, "sel")
        print("%s mean + node_embedding fan %s dims %s: loss %.4f -> %.4f, |g| %.3f" % (
            p, fan, odims, float(out[p + "s0_loss"]), float(out[p + "s1_loss"]), float(out[p + "s0_gradnorm"])))
    out["n_pokec"] = np.array(len(cfgs))


def dense_adjacency(n, K, rng):
    """what utils/convert.py:71-98 writes: [n + 1, K] int64, 0-based ids, the dummy node n in the last row"""
    adj = rng.randint(0, n, size=(n + 1, K))
    adj[n] = n
    lonely = rng.randint(0, n, size=5)
    adj[lonely] = n                                   # nodes without neighbours point at the dummy
    return adj.astype(np.int64)


def gen_dense_steps(out):
    cfgs = [((5, 3), (128, 128), 24, 0.0, 1.0), ((4, 2), (16, 16), 12, 5e-4, 1.0)]
    for case, (fan, odims, D, wd, fscale) in enumerate(cfgs):
        grng = np.random.RandomState(950 + case)
        n, K, C, B = 180, 16, 5, 12
        adj, train_adj = dense_adjacency(n, K, grng), dense_adjacency(n, K, grng)
        feats_np = grng.normal(size=(n + 1, D)).astype(np.float32)
        feats_np[n] = 0
        feats = torch.FloatTensor(feats_np)
        torch.manual_seed(70 + case)
        model = models.GSSupervised(**{
            "sampler_class": nn_modules.sampler_lookup["uniform_neighbor_sampler"],
            "adj": torch.LongTensor(adj), "train_adj": torch.LongTensor(train_adj),
            "prep_class": nn_modules.prep_lookup["identity"],
            "aggregator_class": nn_modules.aggregator_lookup["mean"],
            "input_dim": D, "n_nodes": n + 1, "n_classes": C,
            "layer_specs": [{"n_train_samples": fan[0], "n_val_samples": fan[0], "output_dim": odims[0],
                             "activation": F.relu},
                            {"n_train_samples": fan[1], "n_val_samples": fan[1], "output_dim": odims[1],
                             "activation": lambda x: x}],
            "lr_init": 0.01, "lr_schedule": "constant", "weight_decay": wd,
        })
        p = "d%d_" % case
        out[p + "fanouts"], out[p + "out_dims"] = np.array(fan), np.array(odims)
        out[p + "weight_decay"], out[p + "n_classes"] = np.array(wd), np.array(C)
        out[p + "feats"], out[p + "adj"], out[p + "tadj"] = feats_np, adj, train_adj
        out.update(gg.sd_arrays(model, p + "w0_"))
        ids = torch.LongTensor(grng.randint(0, n, size=B))
        targets = torch.LongTensor(grng.randint(0, C, size=(B, 1)))
        out[p + "ids"], out[p + "targets"] = _np(ids), _np(targets)
        torch.manual_seed(8000 + case)
        out[p + "torch_seed"] = np.array(8000 + case)
        # the frontier the first step samples (same generator state)
        st = torch.get_rng_state()
        h1 = model.train_sampler(ids, n_samples=fan[0]).contiguous().view(-1)
        h2 = model.train_sampler(h1, n_samples=fan[1]).contiguous().view(-1)
        out[p + "s0_h1"], out[p + "s0_h2"] = _np(h1).astype(np.int64), _np(h2).astype(np.int64)
        torch.set_rng_state(st)
        two_steps(out, p, model, ids, feats, targets, problem.ProblemLosses.classification, PermRecorder, "perm")
        print("%s dense sampler fan %s dims %s: loss %.4f -> %.4f, |g| %.3f" % (
            p, fan, odims, float(out[p + "s0_loss"]), float(out[p + "s1_loss"]), float(out[p + "s0_gradnorm"])))
    out["n_dense"] = np.array(len(cfgs))


def gen_dense_epoch(out):
    rng = np.random.RandomState(11)
    n, K = 220, 16
    adj = dense_adjacency(n, K, rng)
    p = gg.fake_problem(n + 1, 150, "classification", 5, np.random.RandomState(4))
    sampler = nn_modules.UniformNeighborSampler(adj=torch.LongTensor(adj))
    out["k0_adj"] = adj
    out["k0_nodes"] = p.nodes["train"].astype(np.int64)
    seed = 123
    helpers.set_seeds(seed ** 2)
    k = 0
    for ids, targets, prog in p.iterate(mode="train", batch_size=64, shuffle=True):
        h1 = sampler(ids, n_samples=5).contiguous().view(-1)
        h2 = sampler(h1, n_samples=3).contiguous().view(-1)
        out["k0_b%d_ids" % k] = _np(ids).astype(np.int64)
        out["k0_b%d_h1" % k] = _np(h1).astype(np.int64)
        out["k0_b%d_h2" % k] = _np(h2).astype(np.int64)
        k += 1
    out["k0_n_batches"], out["k0_seed"] = np.array(k), np.array(seed)
    out["k0_tail_np"] = np.random.randint(0, 2 ** 31 - 1, size=4).astype(np.int64)
    out["k0_tail_torch"] = torch.randperm(16).numpy().astype(np.int64)
    print("k0 dense epoch: %d batches" % k)


def main():
    out = {}
    gen_pokec_mean(out)
    gen_dense_steps(out)
    gen_dense_epoch(out)
    path = os.path.join(HERE, "round3_kat.npz")
    np.savez_compressed(path, **out)
    print("round3_kat: %.2f MB" % (os.path.getsize(path) / 1e6))


if __name__ == "__main__":
    main()
