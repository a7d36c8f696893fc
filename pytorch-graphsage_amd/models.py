"""
models.py -- GSSupervised, the layer-stacking loop and train_step (reference models.py:21-104),
kept drop-in: same constructor keywords (as passed at train.py:94-123), same methods
(`forward(ids, feats, train)`, `set_progress`, `train_step`), same parameter names.

What changes underneath: the frontier is built on the device by K1, the
per-hop `feats[ids]` gathers are row *references* consumed inside the aggregator kernels
(K2-K5) instead of materialised [B*f1*f2, D] tensors, and `train_step` has a gradient-sync
hook between backward and clip for data-parallel runs (one flat RCCL all-reduce, dist.py).
"""
from functools import partial

import torch
from torch import nn
from torch.nn import functional as F

import os

from . import ops
from .lr import LRSchedule
from .optim import FlatAdam
from .store import FeatureStore


class GSSupervised(nn.Module):
    def __init__(self, input_dim, n_nodes, n_classes, layer_specs, aggregator_class, prep_class,
                 sampler_class, adj, train_adj, lr_init=0.01, weight_decay=0.0,
                 lr_schedule='constant', epochs=10):
        super(GSSupervised, self).__init__()

        # samplers: training walks train_adj, evaluation the full graph (models.py:41-44)
        self.train_sampler = sampler_class(adj=train_adj)
        self.val_sampler = sampler_class(adj=adj)
        self.train_sample_fns = [partial(self.train_sampler, n_samples=spec['n_train_samples'])
                                 for spec in layer_specs]
        self.val_sample_fns = [partial(self.val_sampler, n_samples=spec['n_val_samples'])
                               for spec in layer_specs]

        self.prep = prep_class(input_dim=input_dim, n_nodes=n_nodes)
        width = self.prep.output_dim

        stack = []
        for spec in layer_specs:
            layer = aggregator_class(input_dim=width, output_dim=spec['output_dim'],
                                     activation=spec['activation'])
            stack.append(layer)
            width = layer.output_dim          # 2 * output_dim for the concat (models.py:59)
        self.agg_layers = nn.Sequential(*stack)
        self.fc = nn.Linear(width, n_classes, bias=True)

        # schedule is a function of progress only: `epochs` is not forwarded (models.py:67)
        self.lr_scheduler = partial(getattr(LRSchedule, lr_schedule), lr_init=lr_init)
        self.lr = self.lr_scheduler(0.0)
        self.optimizer = torch.optim.Adam(self.parameters(), lr=self.lr, weight_decay=weight_decay)
        self._init_optimizer = self.optimizer

        self.grad_sync = None                 # set by dist.attach(): called after backward
        self._wrapped = {}

    # ------------------------------------------------------------------------------------
    def _rows_of(self, feats):
        """What `feats[ids]` should index: FeatureStore as is; a CUDA tensor is wrapped zero-copy
        so its gathers are fused too; CPU tensors keep stock indexing (host mode)."""
        if feats is None or isinstance(feats, FeatureStore) or not feats.is_cuda:
            return feats
        key = (feats.data_ptr(), tuple(feats.shape), feats.dtype)
        if key not in self._wrapped:
            self._wrapped = {key: FeatureStore.wrap(feats)}
        return self._wrapped[key]

    def forward(self, ids, feats, train=True):
        # a fused engine that defers the embedding table's zero-gradient Adam updates (engine.sync_rows)
        # settles them HERE: the prep reads `embedding.weight` directly, not through the nn.Embedding
        # module, so a hook on that module would never fire
        settle = getattr(self, "_settle_rows", None)
        if settle is not None:
            settle()
        sample_fns = self.train_sample_fns if train else self.val_sample_fns
        table = self._rows_of(feats)
        rows = (lambda i: table[i]) if table is not None else (lambda i: None)

        # frontier: [B], [B*f1], [B*f1*f2], ...  (models.py:75-81)
        hops = [self.prep(ids, rows(ids), layer_idx=0)]
        for hop, sample in enumerate(sample_fns):
            ids = sample(ids=ids).contiguous().view(-1)
            hops.append(self.prep(ids, rows(ids), layer_idx=hop + 1))

        # layer l maps every adjacent pair of hop representations; shared weights (models.py:85-86)
        for layer in self.agg_layers.children():
            hops = [layer(hops[k], hops[k + 1]) for k in range(len(hops) - 1)]
        assert len(hops) == 1, "len(all_feats) != 1"

        out = F.normalize(hops[0].float(), dim=1)
        if out.is_cuda:
            # the head's projection on K5 in exact fp32 (models.py:91 is an nn.Linear: a library GEMM on the GPU
            # otherwise -- the last one of the module path, forward and backward)
            return ops.linear(out, self.fc.weight, self.fc.bias, compute_dtype="fp32")
        return self.fc(out)

    def __getstate__(self):
        # per-process handles of a fused engine (a weakref and a bound method) do not pickle / deep-copy
        state = dict(self.__dict__)
        state.pop("_engine", None)
        state.pop("_settle_rows", None)
        return state

    def set_progress(self, progress):
        self.lr = self.lr_scheduler(progress)
        LRSchedule.set_lr(self.optimizer, self.lr)

    def _flat_optimizer(self):
        """On the GPU the optimizer built in __init__ (torch.optim.Adam, never stepped yet) is replaced,
        on the first train_step, by optim.FlatAdam (its state_dict keeps torch.optim.Adam's format, so
        checkpoints interchange; a reference to the ORIGINAL optimizer object taken before the first step
        goes stale -- hold `model.optimizer` instead): same arithmetic, Parameters and gradients become
        views of flat buckets, clip + Adam become two launches.  An optimizer assigned from outside,
        one that has state already, CPU parameters or GSAGE_TORCH_ADAM=1 keep the stock route."""
        opt = self.optimizer
        settle = getattr(self, "_settle_rows", None)
        if settle is not None:                    # a fused engine with deferred embedding-table rows
            settle()
        eng = getattr(self, "_engine", None)
        eng = eng() if eng is not None else None
        trained = eng is not None and eng.holds_parameters()     # a fused engine trained them last
        if isinstance(opt, FlatAdam):
            if not opt.owns():                    # somebody re-pointed the Parameters (e.g. a fused engine):
                opt._attach()                     # take them back WITH their current values ...
                if trained:
                    eng.export_optimizer_state(opt)   # ... and the engine's exp_avg / exp_avg_sq / step count
            return opt
        if opt is not self._init_optimizer or opt.state or os.environ.get("GSAGE_TORCH_ADAM", "0") == "1":
            if trained and getattr(self, "_handed_over", None) != id(eng):
                # the stock route takes over from a fused engine: Adam continues from the engine's exp_avg /
                # exp_avg_sq / step count when this optimizer has no state of its own, else say what is lost
                self._handed_over = id(eng)
                try:
                    if opt.state:
                        raise RuntimeError("the optimizer already holds state of its own")
                    opt.load_state_dict(eng.optimizer_state_dict())
                except Exception as e:
                    import warnings
                    warnings.warn("gsage: %s -- Adam's moments of the fused engine's steps are NOT carried over "
                                  "into model.optimizer" % (e,))
            return None
        params = [p for p in self.parameters() if p.requires_grad]
        if not params or not all(p.is_cuda and p.dtype == torch.float32 for p in params):
            return None
        g = opt.param_groups[0]
        self.optimizer = FlatAdam(params, lr=g["lr"], weight_decay=g.get("weight_decay", 0.0),
                                  betas=g.get("betas", (0.9, 0.999)), eps=g.get("eps", 1e-8))
        if trained:
            eng.export_optimizer_state(self.optimizer)
        return self.optimizer

    def optimizer_state_dict(self):
        """torch.optim.Adam-format state of whoever trained last: the fused engine's buckets when one holds
        the Parameters (`model.optimizer` is then still the never-stepped optimizer of __init__), else
        `model.optimizer.state_dict()`.  Loads into the torch.optim.Adam the reference builds (models.py:69)."""
        eng = getattr(self, "_engine", None)
        eng = eng() if eng is not None else None
        if eng is not None and eng.holds_parameters():
            return eng.optimizer_state_dict()
        return self.optimizer.state_dict()

    def train_step(self, ids, feats, targets, loss_fn):
        flat = self._flat_optimizer()
        self.optimizer.zero_grad()
        preds = self(ids, feats, train=True)
        loss = loss_fn(preds, targets.squeeze())
        loss.backward()
        if self.grad_sync is not None:
            self.grad_sync(self)
        if flat is not None:
            flat.clip_and_step(5.0)               # models.py:101-102 in two launches
        else:
            torch.nn.utils.clip_grad_norm_(self.parameters(), 5)
            self.optimizer.step()
        return preds
