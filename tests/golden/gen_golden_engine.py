#!/usr/bin/env python
"""
gen_golden_engine.py -- golden vectors for the FUSED train-step engines (TEST INFRASTRUCTURE, runs
ONLY in the build container; same harness and shims as gen_golden.py).

model_kat.npz (gen_golden.py) pins the module-level path on tiny layers (output_dim 8 / 16).  The
fused engines that bench.py times have shape preconditions the tiny cases do not meet (the seed-level
kernel wants width 256 = 2 x 128, the pool engine output_dim % 64), and models.py:85-86 is generic
in depth while train.py:105-118 only ever builds two layers.  This script drives the REFERENCE's
GSSupervised.train_step (models.py:97-104) for two steps on configurations that reach every
kernel of the engines:

  e0  mean      2 layers  fan-out 25/10  output_dim 128/128   seed-level kernel, 13-row half-waves
  e1  mean      2 layers  fan-out  5/3   output_dim 128/128   seed-level kernel, short fan-out, L2 decay
  e2  mean      3 layers  fan-out  4/3/2 output_dim 16/16/8   generic per-level path, three levels
  e3  mean      3 layers  fan-out  5/4/3 output_dim 128 x 3   BASELINE configs[4]'s depth, scaled down
  e4  max_pool  2 layers  fan-out  5/3   output_dim 64/64     FusedPoolTrainStep, L2 decay
  e5  mean_pool 2 layers  fan-out  4/2   output_dim 64/64     FusedPoolTrainStep

and records inputs (graph, features, initial weights, seed ids, targets, the `sel` the sampler drew
at nn_modules.py:88 for every hop of both steps) and outputs (predictions, loss and pre-clip gradient
norm of both steps, clipped gradients of step 0, weights after the two Adam updates).

    python -B tests/golden/gen_golden_engine.py      # writes tests/golden/engine_kat.npz
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

models, nn_modules, problem = gg.models, gg.nn_modules, gg.problem
_np = gg._to_numpy

CASES = [
    # (aggregator, fanouts, out_dims, D, n_nodes, B, n_classes, lr_schedule, weight_decay, weight scale)
    # The embedding is normalised (models.py:90), so the loss does not change when the aggregator
    # weights are scaled by c while their gradients scale by 1/c: the cases with scale << 1 start from
    # the reference's own initial weights times c, have a pre-clip norm > 5, and so exercise
    # clip_grad_norm (models.py:101)
    ("mean", (25, 10), (128, 128), 40, 300, 11, 7, "constant", 0.0, 1.0),
    ("mean", (5, 3), (128, 128), 24, 200, 13, 5, "constant", 5e-4, 0.02),
    ("mean", (4, 3, 2), (16, 16, 8), 12, 150, 9, 5, "linear", 0.0, 1.0),
    ("mean", (5, 4, 3), (128, 128, 128), 16, 250, 6, 6, "constant", 0.0, 0.05),
    ("max_pool", (5, 3), (64, 64), 16, 160, 10, 5, "constant", 5e-4, 1.0),
    ("mean_pool", (4, 2), (64, 64), 16, 160, 10, 4, "constant", 0.0, 0.02),
]


def main():
    out = {}
    for case, (aggn, fan, odims, D, n, B, C, sched, wd, fscale) in enumerate(CASES):
        grng = np.random.RandomState(700 + case)
        degs = grng.randint(0, 14, size=n + 1)
        degs[0], degs[2], degs[n] = 0, 0, 3
        adj = gg.make_ref_csr(n, degs, grng)
        tdegs = np.minimum(degs, grng.randint(0, 11, size=n + 1))
        tdegs[n] = 2
        tdegs[5] = max(tdegs[5], 1)
        train_adj = gg.make_ref_csr(n, tdegs, grng)
        n_rows = adj.shape[0]
        feats_np = grng.normal(size=(n_rows, D)).astype(np.float32)
        feats_np[0] = 0
        feats = torch.FloatTensor(feats_np)

        torch.manual_seed(40 + case)
        np.random.seed(40 + case)
        L = len(fan)
        specs = [{"n_train_samples": fan[l], "n_val_samples": fan[l], "output_dim": odims[l],
                  "activation": F.relu if l < L - 1 else (lambda x: x)} for l in range(L)]
        model = models.GSSupervised(**{
            "sampler_class": nn_modules.sampler_lookup["sparse_uniform_neighbor_sampler"],
            "adj": adj, "train_adj": train_adj,
            "prep_class": nn_modules.prep_lookup["identity"],
            "aggregator_class": nn_modules.aggregator_lookup[aggn],
            "input_dim": D, "n_nodes": n_rows, "n_classes": C, "layer_specs": specs,
            "lr_init": 0.01, "lr_schedule": sched, "weight_decay": wd,
        })
        with torch.no_grad():
            for prm in model.agg_layers.parameters():
                prm.mul_(fscale)
        p = "e%d_" % case
        out[p + "cfg"] = np.array([aggn, "identity", "classification", sched])
        out[p + "fanouts"] = np.array(fan)
        out[p + "out_dims"] = np.array(odims)
        out[p + "weight_decay"] = np.array(wd)
        out[p + "n_classes"] = np.array(C)
        out[p + "feats"] = feats_np
        out.update(gg.csr_arrays(adj, p + "adj_"))
        out.update(gg.csr_arrays(train_adj, p + "tadj_"))
        out.update(gg.sd_arrays(model, p + "w0_"))

        ids = torch.LongTensor(grng.randint(1, n_rows, size=B))
        ids[0] = 2                                   # a seed without neighbours: samples the dummy node
        targets = torch.LongTensor(grng.randint(0, C, size=(B, 1)))
        loss_fn = problem.ProblemLosses.classification
        out[p + "ids"] = _np(ids)
        out[p + "targets"] = _np(targets)

        np.random.seed(4321 + case)
        for step in range(2):
            model.set_progress(0.25 * step)
            out[p + "lr%d" % step] = np.array(model.lr)
            w_before = {k: v.clone() for k, v in model.state_dict().items()}
            with gg.ChoiceRecorder() as rec:
                preds = model.train_step(ids=ids, feats=feats, targets=targets, loss_fn=loss_fn)
            sels = [c[1] for c in rec.calls]
            assert len(sels) == L
            for h, sv in enumerate(sels):
                out[p + "s%d_sel%d" % (step, h)] = sv.astype(np.int32)
            out[p + "s%d_preds" % step] = _np(preds).copy()
            if step == 0:
                for k, v in model.named_parameters():
                    out[p + "s0_cg_%s" % k] = _np(v.grad).copy()                 # clipped grads
            w_after = {k: v.clone() for k, v in model.state_dict().items()}
            # loss and pre-clip gradient norm: replay the same draws on the pre-step weights
            model.load_state_dict(w_before)
            model.optimizer.zero_grad()
            with gg.ChoiceReplayer(sels):
                pr2 = model(ids, feats, train=True)
            loss = loss_fn(pr2, targets.squeeze())
            loss.backward()
            tn = torch.sqrt(sum((q.grad.detach() ** 2).sum() for q in model.parameters() if q.grad is not None))
            assert np.allclose(_np(pr2), _np(preds), atol=1e-6)
            out[p + "s%d_loss" % step] = np.array(float(loss))
            out[p + "s%d_gradnorm" % step] = np.array(float(tn))
            model.load_state_dict(w_after)
        out.update(gg.sd_arrays(model, p + "w2_"))
        print("e%d %s fan %s dims %s: loss %.4f -> %.4f, |g| %.3f" % (
            case, aggn, fan, odims, float(out[p + "s0_loss"]), float(out[p + "s1_loss"]),
            float(out[p + "s0_gradnorm"])))
    out["n_cases"] = np.array(len(CASES))
    np.savez_compressed(os.path.join(HERE, "engine_kat.npz"), **out)
    print("engine_kat: %d cases, %.1f MB" % (len(CASES), os.path.getsize(os.path.join(HERE, "engine_kat.npz")) / 1e6))


if __name__ == "__main__":
    main()
