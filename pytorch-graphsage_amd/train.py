#!/usr/bin/env python
"""
train.py -- command-line driver, drop-in for the reference's train.py: every flag of
train.py:44-66 with the same defaults, the same seeding order (train.py:81,133), the same
per-batch / final / test JSON lines on stdout (train.py:151-157,165-170,174-176) and the
model repr on stderr, so run.sh / utils/pokec.sh work unchanged.

Additions (all optional): --rng {compat,philox}, --precision {bf16,fp32}, data-parallel execution
when launched under torch.distributed.run (one process per GPU, RCCL grad all-reduce), and --engine:

  auto (default)  a fused engine (engine.Fused{Mean,Pool,Attn}TrainStep: a handful of kernel launches per step
                  instead of one framework op per tensor expression) whenever one covers the model, else the
                  module path -- with one stderr line saying which engine runs or why none does.  The run is
                  the reference's run: the epoch is shuffled with numpy's legacy stream and cut into the
                  reference's near-equal `array_split` chunks (problem.py:141-153; a recorded step has one
                  geometry, so chunks one seed short are padded and the head ignores the padding), the sampler
                  consumes the SAME generators in the SAME order (--rng compat: numpy's stream, on the device;
                  dense sampler: torch.randperm), so the sampled frontiers are bit-identical to the module
                  path's and to the reference's, and one JSON line is printed per batch (train.py:150-158).
  fused           the same engines, asked for explicitly: a model none covers is an error, and a line is printed
                  every --log-interval batches only (the reference parses that flag and ignores it,
                  train.py:64) -- the per-batch readback is the one thing left that costs a host sync per step.
  eager           the module path (GSSupervised.train_step), one launch per operator.
Data-parallel runs keep batches of one fixed size (--batch-size / world per rank, counter-based sampler).
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
set_seeds, to_numpy, batch_metric = gs.set_seeds, gs.to_numpy, gs.batch_metric
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


def check_samplers(model):
    """The GPU sampler maps an id outside the adjacency to the dummy node and raises a device flag
    instead of synchronising per call; the reference (and host mode) raise IndexError on the spot.  Read
    the flags here -- once per evaluation, so no sync is added to a train step."""
    for s in (model.train_sampler, model.val_sampler):
        for csr in getattr(s, "_dev", {}).values():          # the device copies a sparse sampler has made
            csr.check()


def evaluate(model, problem, mode='val'):
    assert mode in ['test', 'val']
    preds, acts = [], []
    # every rank evaluates the WHOLE fold: draw the samples a single process would (rank offset off)
    shard = getattr(model.val_sampler, "shard", None)
    if shard is not None:
        model.val_sampler.shard = (0, 1)
    try:
        for (ids, targets, _) in problem.iterate(mode=mode, shuffle=False):
            preds.append(model(ids, problem.feats, train=False).detach())
            acts.append(targets.reshape(targets.shape[0], -1))
    finally:
        if shard is not None:
            model.val_sampler.shard = shard
    check_samplers(model)
    if preds and preds[0].is_cuda:
        # scored on the device (problem.DeviceMetrics): the fold's predictions never travel to the host
        return batch_metric(problem.task, torch.cat(acts), torch.cat(preds))
    return problem.metric_fn(np.vstack([to_numpy(a) for a in acts]), np.vstack([to_numpy(p) for p in preds]))


class FusedEvaluator(object):
    """evaluate() on a fused engine's FORWARD launches (engine.FusedTrainStep(eval_only=True).evaluate_fold): the
    fold cut into the reference's chunks (problem.py:141-153 with shuffle=False, batch_size 512 as train.py:32 calls
    it), the validation sampler's frontier drawn from the generator -- and in the order -- the reference's evaluation
    would draw from, no autograd, no per-op launches, the metric on the device.  One engine per fold (their chunk
    sizes differ), built on first use; a model / store no engine covers keeps the module path (said once)."""

    def __init__(self, cls, model, problem):
        self.cls, self.model, self.problem = cls, model, problem
        self.engines, self.off = {}, False

    def _fold(self, mode, batch_size=512):
        nodes = self.problem.nodes[mode]
        chunks = np.array_split(np.arange(nodes.shape[0]), nodes.shape[0] // batch_size + 1)
        B = max(int(c.shape[0]) for c in chunks)
        mids = np.stack([np.concatenate([nodes[c], np.repeat(nodes[c[:1]], B - c.shape[0])]) for c in chunks])
        return torch.from_numpy(mids).cuda(), [int(c.shape[0]) for c in chunks]

    def prepare(self, mode='val'):
        """build the fold's engine now (part of a run's set-up, not of its first epoch)"""
        if not self.off and self.problem.nodes[mode].shape[0] >= 2:
            ids, live = self._fold(mode)
            if min(live) >= 2:
                self._engine(mode, ids)
        return self

    def _engine(self, mode, ids):
        model, problem = self.model, self.problem
        eng = self.engines.get(mode)
        if eng is None and not self.off:
            try:
                tg = torch.zeros(ids.shape[1], 1, dtype=torch.int64, device=ids.device) \
                    if problem.task == 'classification' else \
                    torch.zeros((ids.shape[1],) + tuple(np.asarray(problem.targets[:1]).shape[1:]), dtype=torch.float32,
                                device=ids.device)
                eng = self.engines[mode] = self.cls(model, problem.feats, problem.loss_fn, ids[0], tg, eval_only=True)
            except Exception as e:
                self.off = True
                print('gsage: evaluation stays on the module path (%s: %s)' % (type(e).__name__, e), file=sys.stderr)
        return eng

    def __call__(self, mode='val'):
        assert mode in ['test', 'val']
        model, problem = self.model, self.problem
        if self.off or problem.nodes[mode].shape[0] < 2:
            return evaluate(model, problem, mode=mode)
        ids, live = self._fold(mode)
        if min(live) < 2:
            return evaluate(model, problem, mode=mode)
        eng = self._engine(mode, ids)
        if eng is None:
            return evaluate(model, problem, mode=mode)
        try:
            preds = eng.evaluate_fold(ids, live)
        except Exception as e:            # (e.g. a validation fan-out the forward-only engine's kernels refuse)
            self.off = True
            self.engines.pop(mode, None)
            print('gsage: evaluation falls back to the module path (%s: %s)' % (type(e).__name__, e), file=sys.stderr)
            return evaluate(model, problem, mode=mode)
        nodes = problem.nodes[mode]
        _, acts = problem._batch(nodes, problem.targets[nodes])
        check_samplers(model)
        eng.csr.check()
        return batch_metric(problem.task, acts.reshape(acts.shape[0], -1), preds)


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
    parser.add_argument('--engine', type=str, default='auto', choices=['auto', 'eager', 'fused'])

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


def main(argv=None, problem=None):
    """problem: a NodeProblem already in memory (NodeProblem.from_arrays) instead of --problem-path's file."""
    args = parse_args(argv)
    set_seeds(args.seed)
    gs.ops.set_compute_dtype(args.precision)
    gs.nn_modules.SparseUniformNeighborSampler.rng_default = args.rng
    # compat mode on the GPU: the sampler consumes numpy's legacy stream on the device (same words, same
    # order; helpers.LegacyStreamOnDevice) instead of drawing on the host and copying `sel` over
    gs.helpers.legacy_stream.enabled = bool(args.cuda and args.rng == 'compat' and
                                            os.environ.get("GSAGE_HOST_SEL", "0") != "1")

    ddp = gs.dist.init_from_env(args.cuda)            # no-op outside torch.distributed.run
    if problem is None:
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
    if args.engine in ('auto', 'fused') and args.cuda:
        cls = choose_engine(args, problem, model, ddp)
        if cls is not None:
            print('gsage: train_step runs on %s' % cls.__name__, file=sys.stderr)
            return train_fused(args, problem, model, ddp, start_time, cls)
    for epoch in range(args.epochs):
        model.train()
        for ids, targets, epoch_progress in problem.iterate(mode='train', shuffle=True,
                                                           batch_size=args.batch_size):
            if ddp is not None:
                ids, targets = ddp.shard(ids, targets)
            model.set_progress((epoch + epoch_progress) / args.epochs)
            preds = model.train_step(ids=ids, feats=problem.feats, targets=targets,
                                     loss_fn=problem.loss_fn)
            train_metric = batch_metric(problem.task, targets, preds)      # on the device when the batch is
            if ddp is None or ddp.rank == 0:
                print(dumps({"epoch": epoch, "epoch_progress": epoch_progress,
                             "train_metric": train_metric, "val_metric": val_metric,
                             "time": time() - start_time}))
                sys.stdout.flush()
        model.eval()
        val_metric = evaluate(model, problem, mode='val')

    gs.helpers.legacy_stream.release()                 # hand numpy's stream back to the host
    print('-- done --', file=sys.stderr)
    if ddp is None or ddp.rank == 0:
        print(dumps({"epoch": epoch, "train_metric": train_metric, "val_metric": val_metric,
                     "time": time() - start_time}))
        sys.stdout.flush()
        if args.show_test:
            print(dumps({"test_f1": evaluate(model, problem, mode='test')}))
    if ddp is not None:
        ddp.close()


def choose_engine(args, problem, model, ddp):
    """The fused engine this run gets, or None for the module path.  Everything that could stop an engine AFTER it
    has re-pointed the model's Parameters into its buckets is decided here: the model / feature store / process
    group (engine.why_not), the head (engine.head_why_not: only a fused head can ignore the padding of the
    reference's unequal chunks) and the batch geometry.  --engine auto: one stderr line says why the module path
    runs; --engine fused: the same sentence is an error."""
    def give_up(why):
        if args.engine == 'fused':
            raise SystemExit('gsage: --engine fused: %s (use --engine auto / eager)' % why)
        print('gsage: %s; using the module path' % why, file=sys.stderr)
        return None
    cls = gs.engine.fused_engine_for(model, problem.feats, explain=True, ddp=ddp)
    if cls is None:
        return give_up('no fused engine covers this model')
    if ddp is not None and args.rng != 'philox':
        return give_up('data-parallel runs of the fused engines need --rng philox')
    nodes = problem.nodes['train']
    world = ddp.world if ddp is not None else 1
    if world > 1:
        B, padded = args.batch_size // world, False
        if B < 2 or nodes.shape[0] < B * world:
            return give_up('fewer training nodes (or a smaller --batch-size) than two seeds per rank')
    else:
        n_batches = nodes.shape[0] // args.batch_size + 1
        B = -(-nodes.shape[0] // n_batches)
        padded = nodes.shape[0] % n_batches != 0
        if nodes.shape[0] // n_batches < 2:           # the smallest of the reference's array_split chunks
            return give_up('chunks of fewer than two training nodes')
    example = torch.zeros(1, dtype=torch.int64 if problem.task == 'classification' else torch.float32)
    why = cls.head_why_not(model, problem.loss_fn, example, B, padded, world)
    if why is not None:
        return give_up(why)
    return cls


def epoch_chunks(nodes, batch_size):
    """The reference's batches of one epoch (problem.py:141-153): a permutation from numpy's legacy stream, cut
    into n // batch_size + 1 near-equal chunks (never all of one size: quirk 6).  -> list of index arrays."""
    gs.helpers.legacy_stream.release()               # the shuffle is a HOST draw from the shared stream
    order = np.random.permutation(np.arange(nodes.shape[0]))
    return np.array_split(order, order.shape[0] // batch_size + 1)


def train_fused(args, problem, model, ddp, start_time, cls):
    """The training run on a fused engine (see the module docstring)."""
    assert args.cuda
    world, rank = (ddp.world, ddp.rank) if ddp is not None else (1, 0)
    nodes = problem.nodes['train']
    dev = torch.device('cuda')
    cls_task = problem.task == 'classification'
    every = 1 if args.engine == 'auto' else max(args.log_interval, 1)

    def targets_of(mids, shape):
        tg = np.asarray(problem.targets[mids.reshape(-1)])
        if cls_task:
            return torch.from_numpy(tg.reshape(shape)).long().to(dev)
        return torch.from_numpy(tg.reshape(shape + (-1,)).astype(np.float32)).to(dev)   # multilabel / regression

    if world > 1:
        # data-parallel: batches of one fixed size (the < batch-size tail of an epoch is dropped)
        B = args.batch_size // world
        n_batches = nodes.shape[0] // (B * world)
        assert n_batches >= 1, 'fewer training nodes than one global batch'

        def epoch_batches():
            gs.helpers.legacy_stream.release()
            order = np.random.permutation(np.arange(nodes.shape[0]))[:n_batches * B * world]     # problem.py:146
            mids = nodes[order].reshape(n_batches, world, B)[:, rank]
            return torch.from_numpy(np.ascontiguousarray(mids)).to(dev), targets_of(mids, (n_batches, B)), None
    else:
        # the reference's own chunks, padded to the largest one with the chunk's first seed
        n_batches = nodes.shape[0] // args.batch_size + 1
        B = -(-nodes.shape[0] // n_batches)
        assert B >= 2, 'fewer than two training nodes per batch'

        def epoch_batches():
            chunks = epoch_chunks(nodes, args.batch_size)
            mids = np.stack([np.concatenate([nodes[c], np.repeat(nodes[c[:1]], B - c.shape[0])]) for c in chunks])
            return (torch.from_numpy(mids).to(dev), targets_of(mids, (n_batches, B)),
                    [int(c.shape[0]) for c in chunks])
    ids, tgs, live = epoch_batches()
    first = tgs[0].view(B, 1) if cls_task else tgs[0]
    step = cls(model, problem.feats, problem.loss_fn, ids[0], first, ddp=ddp)
    # engines with the fused classification head walk a device-resident queue of the epoch's batches; the others
    # (regression: the fused L1 head; multilabel: stock torch ops inside the captured step) take one batch per call
    queued = bool(step.fused_head)
    # (choose_engine has made sure that a head without a fused kernel never meets padded chunks)
    assert live is None or step.fused_head or step.fused_l1 or min(live) == B
    val_metric = train_metric = None
    epoch = 0
    fold_eval = FusedEvaluator(cls, model, problem).prepare('val') if os.environ.get("GSAGE_FUSED_EVAL", "1") == "1" else \
        (lambda mode='val': evaluate(model, problem, mode=mode))
    # The per-batch line (train.py:150-158) without a host sync per step: batch b is scored on the device right behind
    # its step into a small device ring (problem.MetricRing), and the ring is read back -- one copy -- every 32 batches
    # and at the end of every epoch: same lines, same order, same values, printed 32 at a time.
    # (the `time` of a line is taken when its batch is SUBMITTED, not when the ring is read back; GSAGE_METRIC_RING=1
    # makes the ring one slot deep: the reference's per-batch flush, one host sync per batch)
    ring = gs.problem.MetricRing(problem.task, dev, capacity=max(1, int(os.environ.get("GSAGE_METRIC_RING", "32"))))
    pending = []
    # (class ids outside [0, C) are scored the reference's way, on the host: the synchronous route)
    tg_ok = problem.task != 'classification' or (int(np.min(problem.targets)) >= 0 and
                                                 int(np.max(problem.targets)) < problem.n_classes)

    def flush():
        nonlocal train_metric
        for (prog, stamp), metric in zip(pending, ring.results()):
            train_metric = metric
            print(dumps({"epoch": epoch, "epoch_progress": prog, "train_metric": train_metric,
                         "val_metric": val_metric, "time": stamp}))
        del pending[:]
        sys.stdout.flush()
    # GSAGE_TRAIN_TIMING=1 (bench.py's CLI measurement): wall seconds of every epoch's batch loop, bracketed by device
    # synchronisations, in step.timing -- the log lines reach stdout a ring at a time, their `time` stamps do not
    # resolve single batches any more
    timing = os.environ.get("GSAGE_TRAIN_TIMING", "0") == "1"
    step.timing = []
    for epoch in range(args.epochs):
        model.train()
        t_epoch = time()
        if epoch > 0:
            ids, tgs, live = epoch_batches()
        if queued:
            step.load_epoch(ids, tgs, n_valid=live)          # (compat / dense sampler: draws the epoch's values)
        if timing:
            torch.cuda.synchronize()
        t_loop = time()
        try:
            for b in range(n_batches):
                nb = live[b] if live is not None else B
                step.set_progress((epoch + b / n_batches) / args.epochs)
                preds = step.step_queue() if queued else step(ids[b, :nb], tgs[b, :nb])
                if (b % every == 0 or b == n_batches - 1) and rank == 0:
                    if tg_ok and preds.dtype == torch.float32 and preds.is_contiguous():
                        # (a full batch costs one indexing op here: the loop's host time per batch is what bounds the CLI)
                        full = nb == B
                        ring.score(tgs[b] if full else tgs[b, :nb], preds if full else preds[:nb])
                        pending.append((b / n_batches, time() - start_time))
                        if ring.pending == ring.capacity:
                            flush()
                    else:
                        flush()
                        train_metric = batch_metric(problem.task, tgs[b, :nb].view(nb, -1), preds[:nb])
                        print(dumps({"epoch": epoch, "epoch_progress": b / n_batches, "train_metric": train_metric,
                                     "val_metric": val_metric, "time": time() - start_time}))
                        sys.stdout.flush()
        except BaseException:
            try:
                flush()                   # (a run that dies mid-epoch still prints the batches it scored ...
            except Exception:             #  ... but a failing read-back must not replace the error that ended it)
                pass
            raise
        flush()
        if timing:
            torch.cuda.synchronize()
            step.timing.append({"epoch": epoch, "batches": n_batches, "seeds": int(sum(live)) if live is not None
                                else n_batches * B * world, "loop_s": time() - t_loop, "with_draws_s": time() - t_epoch})
        model.eval()
        t_val = time()
        val_metric = fold_eval('val')
        if timing:
            torch.cuda.synchronize()
            step.timing[-1]["val_s"] = time() - t_val
            step.timing[-1]["epoch_s"] = time() - t_epoch            # draws + batch loop + validation
    if rank == 0 and args.show_test:
        test_metric = fold_eval('test')
    gs.helpers.legacy_stream.release()                 # hand numpy's stream back to the host
    print('-- done --', file=sys.stderr)
    if rank == 0:
        print(dumps({"epoch": epoch, "train_metric": train_metric, "val_metric": val_metric,
                     "time": time() - start_time}))
        sys.stdout.flush()
        if args.show_test:
            print(dumps({"test_f1": test_metric}))
    if ddp is not None:
        ddp.close()
    return step


if __name__ == "__main__":
    main()
