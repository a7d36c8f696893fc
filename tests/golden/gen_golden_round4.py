#!/usr/bin/env python
"""
gen_golden_round4.py -- golden vectors for round 4 (TEST INFRASTRUCTURE, runs ONLY in the build container; same
harness and shims as gen_golden.py, which imports the reference).

  q0, q1   the reference's utils/pokec.sh:11-13 configuration -- the command behind the reference's ONLY published
           number (utils/pokec.sh:15): the DEFAULT dense sampler (UniformNeighborSampler, nn_modules.py:19-49) +
           the trainable node-embedding prep without features (nn_modules.py:126-155) + MEAN aggregators
           (nn_modules.py:185-204) + regression_mae (problem.py:39-42, the [B,1]-vs-[B] broadcast) -- two train
           steps of the reference's GSSupervised.train_step (models.py:97-104) from a recorded torch seed, with the
           torch.randperm each sampler call drew, the frontier of step 0, predictions, clipped gradients, loss,
           gradient norm, and every weight incl. the embedding table after each step.

    python -B tests/golden/gen_golden_round4.py      # writes tests/golden/round4_kat.npz
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


def gen_pokec_dense(out):
    cfgs = [((5, 3), (16, 16), 0.0, 1.0), ((4, 2), (64, 64), 5e-4, 0.05)]
    for case, (fan, odims, wd, fscale) in enumerate(cfgs):
        grng = np.random.RandomState(1200 + case)
        n, K, B = 160, 16, 13
        adj, train_adj = g3.dense_adjacency(n, K, grng), g3.dense_adjacency(n, K, grng)
        torch.manual_seed(90 + case)
        model = models.GSSupervised(**{
            "sampler_class": nn_modules.sampler_lookup["uniform_neighbor_sampler"],
            "adj": torch.LongTensor(adj), "train_adj": torch.LongTensor(train_adj),
            "prep_class": nn_modules.prep_lookup["node_embedding"],
            "aggregator_class": nn_modules.aggregator_lookup["mean"],
            "input_dim": None, "n_nodes": n + 1, "n_classes": 1,
            "layer_specs": [{"n_train_samples": fan[0], "n_val_samples": fan[0], "output_dim": odims[0],
                             "activation": F.relu},
                            {"n_train_samples": fan[1], "n_val_samples": fan[1], "output_dim": odims[1],
                             "activation": lambda x: x}],
            "lr_init": 0.01, "lr_schedule": "constant", "weight_decay": wd,
        })
        with torch.no_grad():
            for prm in model.agg_layers.parameters():
                prm.mul_(fscale)
        p = "q%d_" % case
        out[p + "fanouts"], out[p + "out_dims"] = np.array(fan), np.array(odims)
        out[p + "weight_decay"], out[p + "n_classes"] = np.array(wd), np.array(1)
        out[p + "adj"], out[p + "tadj"] = adj, train_adj
        out.update(gg.sd_arrays(model, p + "w0_"))
        ids = torch.LongTensor(grng.randint(0, n, size=B))
        targets = torch.FloatTensor(grng.normal(30, 8, size=(B, 1)).astype(np.float32))
        out[p + "ids"], out[p + "targets"] = _np(ids), _np(targets)
        torch.manual_seed(8100 + case)
        out[p + "torch_seed"] = np.array(8100 + case)
        st = torch.get_rng_state()
        h1 = model.train_sampler(ids, n_samples=fan[0]).contiguous().view(-1)
        h2 = model.train_sampler(h1, n_samples=fan[1]).contiguous().view(-1)
        out[p + "s0_h1"], out[p + "s0_h2"] = _np(h1).astype(np.int64), _np(h2).astype(np.int64)
        torch.set_rng_state(st)
        g3.two_steps(out, p, model, ids, None, targets, problem.ProblemLosses.regression_mae, g3.PermRecorder, "perm")
        print("%s dense sampler + node_embedding + mean fan %s dims %s: loss %.4f -> %.4f, |g| %.3f" % (
            p, fan, odims, float(out[p + "s0_loss"]), float(out[p + "s1_loss"]), float(out[p + "s0_gradnorm"])))
    out["n_pokec_dense"] = np.array(len(cfgs))


def main():
    out = {}
    gen_pokec_dense(out)
    path = os.path.join(HERE, "round4_kat.npz")
    np.savez_compressed(path, **out)
    print("round4_kat: %.2f MB" % (os.path.getsize(path) / 1e6))


if __name__ == "__main__":
    main()
