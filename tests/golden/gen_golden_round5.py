#!/usr/bin/env python
"""
gen_golden_round5.py -- golden vectors for round 5 (TEST INFRASTRUCTURE, runs ONLY in the build container; same
harness and shims as gen_golden.py, which imports the reference).

  e0, e1   the trainable node-embedding prep CONCATENATED with features (nn_modules.py:152-153:
           torch.cat([feats, fc(embedding)], dim=1); every hop's rows through the same prep, the seeds reading the
           spare row n_nodes: nn_modules.py:143-149) under MEAN aggregators (nn_modules.py:185-204), sparse sampler:
           e0 classification, feature width 24; e1 regression_mae (problem.py:39-42), feature width 40, weight decay
           on -- two train steps of the reference's GSSupervised.train_step (models.py:97-104) with the `sel` its
           sampler drew (nn_modules.py:88), predictions, clipped gradients, loss, gradient norm and every weight
           incl. the embedding table after each step.  (model_kat.npz c5 is the same model family at feature width 12.)
  f0 - f2  the other aggregators over the same prep: f0 max_pool (nn_modules.py:207-256) without features,
           classification; f1 mean_pool beside 24 feature columns, regression_mae; f2 attention (nn_modules.py:279-317)
           beside 24 feature columns, classification -- the same records.

    python -B tests/golden/gen_golden_round5.py      # writes tests/golden/round5_kat.npz
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
import gen_golden_round3 as g3

models, nn_modules, problem = gg.models, gg.nn_modules, gg.problem
_np = gg._to_numpy


def gen_embedding_beside_features(out):
    cfgs = [("classification", 24, (5, 3), (16, 16), 0.0, 1.0), ("regression_mae", 40, (4, 2), (64, 64), 5e-4, 0.05)]
    for case, (task, D, fan, odims, wd, fscale) in enumerate(cfgs):
        grng = np.random.RandomState(1500 + case)
        n = 150
        degs = grng.randint(0, 12, size=n + 1)
        degs[0], degs[2], degs[n] = 0, 0, 3
        adj = gg.make_ref_csr(n, degs, grng)
        tdegs = np.minimum(degs, grng.randint(0, 9, size=n + 1))
        tdegs[n] = 2
        train_adj = gg.make_ref_csr(n, tdegs, grng)
        n_rows = adj.shape[0]
        n_classes = 5 if task == "classification" else 1
        feats_np = grng.normal(size=(n_rows, D)).astype(np.float32)
        feats_np[0] = 0
        feats = torch.FloatTensor(feats_np)
        torch.manual_seed(120 + case)
        np.random.seed(120 + case)
        model = models.GSSupervised(**{
            "sampler_class": nn_modules.sampler_lookup["sparse_uniform_neighbor_sampler"],
            "adj": adj, "train_adj": train_adj,
            "prep_class": nn_modules.prep_lookup["node_embedding"],
            "aggregator_class": nn_modules.aggregator_lookup["mean"],
            "input_dim": D, "n_nodes": n_rows, "n_classes": n_classes,
            "layer_specs": [{"n_train_samples": fan[0], "n_val_samples": fan[0], "output_dim": odims[0],
                             "activation": F.relu},
                            {"n_train_samples": fan[1], "n_val_samples": fan[1], "output_dim": odims[1],
                             "activation": lambda x: x}],
            "lr_init": 0.01, "lr_schedule": "constant", "weight_decay": wd,
        })
        assert model.prep.output_dim == D + model.prep.embedding_dim
        with torch.no_grad():
            for prm in model.agg_layers.parameters():
                prm.mul_(fscale)
        p = "e%d_" % case
        out[p + "cfg"] = np.array(["mean", "node_embedding", task, "constant"])
        out[p + "has_feats"] = np.array(1)
        out[p + "fanouts"], out[p + "out_dims"] = np.array(fan), np.array(odims)
        out[p + "weight_decay"], out[p + "n_classes"] = np.array(wd), np.array(n_classes)
        out[p + "feats"] = feats_np
        out.update(gg.csr_arrays(adj, p + "adj_"))
        out.update(gg.csr_arrays(train_adj, p + "tadj_"))
        out.update(gg.sd_arrays(model, p + "w0_"))
        B = 11
        ids = torch.LongTensor(grng.randint(1, n_rows, size=B))
        ids[0] = 2                                   # a seed without neighbours: samples the dummy node
        if task == "classification":
            targets = torch.LongTensor(grng.randint(0, n_classes, size=(B, 1)))
        else:
            targets = torch.FloatTensor(grng.normal(30, 8, size=(B, 1)).astype(np.float32))
        out[p + "ids"], out[p + "targets"] = _np(ids), _np(targets)
        np.random.seed(7777 + case)
        g3.two_steps(out, p, model, ids, feats, targets, getattr(problem.ProblemLosses, task), gg.ChoiceRecorder, "sel")
        print("%s mean + [features %d | node_embedding] %s fan %s dims %s: loss %.4f -> %.4f, |g| %.3f" % (
            p, D, task, fan, odims, float(out[p + "s0_loss"]), float(out[p + "s1_loss"]), float(out[p + "s0_gradnorm"])))
    out["n_emb_feats"] = np.array(len(cfgs))


def gen_pool_over_embedding(out):
    """f0 / f1: the pooling aggregators (nn_modules.py:207-256) over the node-embedding prep -- f0 max_pool, no
    features, classification; f1 mean_pool beside 24 feature columns, regression_mae, weight decay on; f2: the
    attention aggregator (nn_modules.py:279-317) beside 24 feature columns, classification."""
    cfgs = [("max_pool", "classification", 0, (5, 3), (64, 64), 0.0, 1.0),
            ("mean_pool", "regression_mae", 24, (4, 2), (64, 64), 5e-4, 0.3),
            ("attention", "classification", 24, (4, 3), (16, 16), 5e-4, 1.0)]
    for case, (aggn, task, D, fan, odims, wd, fscale) in enumerate(cfgs):
        grng = np.random.RandomState(1700 + case)
        n = 150
        degs = grng.randint(0, 12, size=n + 1)
        degs[0], degs[2], degs[n] = 0, 0, 3
        adj = gg.make_ref_csr(n, degs, grng)
        tdegs = np.minimum(degs, grng.randint(0, 9, size=n + 1))
        tdegs[n] = 2
        train_adj = gg.make_ref_csr(n, tdegs, grng)
        n_rows = adj.shape[0]
        n_classes = 5 if task == "classification" else 1
        feats_np = grng.normal(size=(n_rows, max(D, 1))).astype(np.float32)
        feats_np[0] = 0
        feats = torch.FloatTensor(feats_np) if D else None
        torch.manual_seed(140 + case)
        np.random.seed(140 + case)
        model = models.GSSupervised(**{
            "sampler_class": nn_modules.sampler_lookup["sparse_uniform_neighbor_sampler"],
            "adj": adj, "train_adj": train_adj,
            "prep_class": nn_modules.prep_lookup["node_embedding"],
            "aggregator_class": nn_modules.aggregator_lookup[aggn],
            "input_dim": D if D else None, "n_nodes": n_rows, "n_classes": n_classes,
            "layer_specs": [{"n_train_samples": fan[0], "n_val_samples": fan[0], "output_dim": odims[0],
                             "activation": F.relu},
                            {"n_train_samples": fan[1], "n_val_samples": fan[1], "output_dim": odims[1],
                             "activation": lambda x: x}],
            "lr_init": 0.01, "lr_schedule": "constant", "weight_decay": wd,
        })
        with torch.no_grad():
            for prm in model.agg_layers.parameters():
                prm.mul_(fscale)
        p = "f%d_" % case
        out[p + "cfg"] = np.array([aggn, "node_embedding", task, "constant"])
        out[p + "has_feats"] = np.array(int(D > 0))
        out[p + "fanouts"], out[p + "out_dims"] = np.array(fan), np.array(odims)
        out[p + "weight_decay"], out[p + "n_classes"] = np.array(wd), np.array(n_classes)
        out[p + "feats"] = feats_np
        out.update(gg.csr_arrays(adj, p + "adj_"))
        out.update(gg.csr_arrays(train_adj, p + "tadj_"))
        out.update(gg.sd_arrays(model, p + "w0_"))
        B = 12
        ids = torch.LongTensor(grng.randint(1, n_rows, size=B))
        ids[0] = 2                                   # a seed without neighbours: samples the dummy node
        if task == "classification":
            targets = torch.LongTensor(grng.randint(0, n_classes, size=(B, 1)))
        else:
            targets = torch.FloatTensor(grng.normal(30, 8, size=(B, 1)).astype(np.float32))
        out[p + "ids"], out[p + "targets"] = _np(ids), _np(targets)
        np.random.seed(8888 + case)
        g3.two_steps(out, p, model, ids, feats, targets, getattr(problem.ProblemLosses, task), gg.ChoiceRecorder, "sel")
        print("%s %s + [features %d | node_embedding] %s fan %s dims %s: loss %.4f -> %.4f, |g| %.3f" % (
            p, aggn, D, task, fan, odims, float(out[p + "s0_loss"]), float(out[p + "s1_loss"]), float(out[p + "s0_gradnorm"])))
    out["n_pool_emb"] = np.array(len(cfgs))


def main():
    out = {}
    gen_embedding_beside_features(out)
    gen_pool_over_embedding(out)
    path = os.path.join(HERE, "round5_kat.npz")
    np.savez_compressed(path, **out)
    print("round5_kat: %.2f MB" % (os.path.getsize(path) / 1e6))


if __name__ == "__main__":
    main()
