#!/usr/bin/env python
"""
train.py -- command-line driver, drop-in for the reference's train.py: every flag of
train.py:44-66 with the same defaults, the same seeding order (train.py:81,133), the same
per-batch / final / test JSON lines on stdout (train.py:151-157,165-170,174-176) and the
model repr on stderr, so run.sh / utils/pokec.sh work unchanged.

Additions (all optional): --rng {compat,philox}, --precision {bf16,fp32}, and data-parallel
execution when launched under torch.distributed.run (one process per GPU, RCCL grad all-reduce).
"""
from __future__ import division, print_function

import argparse
import importlib
import json
import os
import sys
from time import time

import numpy as np
import torch
from torch.nn import functional as F

if __package__ in (None, ""):                       # executed as a script: ./train.py
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    gs = importlib.import_module(os.path.basename(os.path.dirname(os.path.abspath(__file__))))
else:
    gs = importlib.import_module(__package__)

GSSupervised, NodeProblem = gs.GSSupervised, gs.NodeProblem
set_seeds, to_numpy = gs.set_seeds, gs.to_numpy
aggregator_lookup, prep_lookup, sampler_lookup = gs.aggregator_lookup, gs.prep_lookup, gs.sampler_lookup


def _round5(obj):
    """ujson.dumps(..., double_precision=5) of the reference: floats printed with 5 decimals."""
    if isinstance(obj, float):
        return round(obj, 5)
    if isinstance(obj, dict):
        return {k: _round5(v) for k, v in obj.items()}
    return obj


def dumps(obj):
    return json.dumps(_round5(obj), separators=(",", ":"))


def evaluate(model, problem, mode='val'):
    assert mode in ['test', 'val']
    preds, acts = [], []
    for (ids, targets, _) in problem.iterate(mode=mode, shuffle=False):
        preds.append(to_numpy(model(ids, problem.feats, train=False)))
        acts.append(to_numpy(targets))
    return problem.metric_fn(np.vstack([a.reshape(a.shape[0], -1) for a in acts]), np.vstack(preds))


def parse_args(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--problem-path', type=str, required=True)
    parser.add_argument('--no-cuda', action="store_true")
    # optimization
    parser.add_argument('--batch-size', type=int, default=512)
    parser.add_argument('--epochs', type=int, default=10)
    parser.add_argument('--lr-init', type=float, default=0.01)
    parser.add_argument('--lr-schedule', type=str, default='constant')
    parser.add_argument('--weight-decay', type=float, default=0.0)
    # architecture
    parser.add_argument('--sampler-class', type=str, default='uniform_neighbor_sampler')
    parser.add_argument('--aggregator-class', type=str, default='mean')
    parser.add_argument('--prep-class', type=str, default='identity')
    parser.add_argument('--n-train-samples', type=str, default='25,10')
    parser.add_argument('--n-val-samples', type=str, default='25,10')
    parser.add_argument('--output-dims', type=str, default='128,128')
    # logging
    parser.add_argument('--log-interval', default=10, type=int)
    parser.add_argument('--seed', default=123, type=int)
    parser.add_argument('--show-test', action="store_true")
    # build-specific (not in the reference)
    parser.add_argument('--rng', type=str, default='compat', choices=['compat', 'philox'])
    parser.add_argument('--precision', type=str, default='bf16', choices=['bf16', 'fp32'])

    args = parser.parse_args(argv)
    args.cuda = not args.no_cuda
    assert args.prep_class in prep_lookup.keys(), 'parse_args: prep_class not in %s' % str(prep_lookup.keys())
    assert args.aggregator_class in aggregator_lookup.keys(), \
        'parse_args: aggregator_class not in %s' % str(aggregator_lookup.keys())
    assert args.batch_size > 1, 'parse_args: batch_size must be > 1'
    return args


def build_model(args, problem):
    n_train = [int(v) for v in args.n_train_samples.split(',')]
    n_val = [int(v) for v in args.n_val_samples.split(',')]
    dims = [int(v) for v in args.output_dims.split(',')]
    depth = len(dims)
    specs = []
    for li in range(depth):
        last = li == depth - 1
        specs.append({
            "n_train_samples": n_train[li],
            "n_val_samples": n_val[li],
            "output_dim": dims[li],
            "activation": (lambda x: x) if last else F.relu,     # train.py:105-118
        })
    return GSSupervised(
        sampler_class=sampler_lookup[args.sampler_class], adj=problem.adj, train_adj=problem.train_adj,
        prep_class=prep_lookup[args.prep_class], aggregator_class=aggregator_lookup[args.aggregator_class],
        input_dim=problem.feats_dim, n_nodes=problem.n_nodes, n_classes=problem.n_classes,
        layer_specs=specs, lr_init=args.lr_init, lr_schedule=args.lr_schedule,
        weight_decay=args.weight_decay)


def main(argv=None):
    args = parse_args(argv)
    set_seeds(args.seed)
    gs.ops.set_compute_dtype(args.precision)
    gs.nn_modules.SparseUniformNeighborSampler.rng_default = args.rng

    ddp = gs.dist.init_from_env(args.cuda)            # no-op outside torch.distributed.run
    problem = NodeProblem(problem_path=args.problem_path, cuda=args.cuda)
    model = build_model(args, problem)
    if args.cuda:
        model = model.cuda()
        gs.ops.warmup(torch.device("cuda"))
    if ddp is not None:
        gs.dist.attach(model, ddp, seed=args.seed)
    print(model, file=sys.stderr)

    set_seeds(args.seed ** 2)                          # train.py:133
    start_time = time()
    val_metric = train_metric = None
    epoch = 0
    for epoch in range(args.epochs):
        model.train()
        for ids, targets, epoch_progress in problem.iterate(mode='train', shuffle=True,
                                                           batch_size=args.batch_size):
            if ddp is not None:
                ids, targets = ddp.shard(ids, targets)
            model.set_progress((epoch + epoch_progress) / args.epochs)
            preds = model.train_step(ids=ids, feats=problem.feats, targets=targets,
                                     loss_fn=problem.loss_fn)
            train_metric = problem.metric_fn(to_numpy(targets), to_numpy(preds))
            if ddp is None or ddp.rank == 0:
                print(dumps({"epoch": epoch, "epoch_progress": epoch_progress,
                             "train_metric": train_metric, "val_metric": val_metric,
                             "time": time() - start_time}))
                sys.stdout.flush()
        model.eval()
        val_metric = evaluate(model, problem, mode='val')

    print('-- done --', file=sys.stderr)
    if ddp is None or ddp.rank == 0:
        print(dumps({"epoch": epoch, "train_metric": train_metric, "val_metric": val_metric,
                     "time": time() - start_time}))
        sys.stdout.flush()
        if args.show_test:
            print(dumps({"test_f1": evaluate(model, problem, mode='test')}))
    if ddp is not None:
        ddp.close()


if __name__ == "__main__":
    main()
