"""
engine/common.py -- what the three fused train-step engines share: flat parameter / gradient / Adam buckets
(the model's Parameters become views), frontier geometry, the fused multi-hop sampler's descriptor, the
classification head, gradient finalisation + clip + Adam, recording into native command lists or hipGraphs,
the device-resident batch queue with its one-batch-ahead software pipeline, the data-parallel order around
the ONE gradient all-reduce, and the Adam-state hand-back to `GSSupervised.train_step`.

The aggregator-specific parts (per-level operands, forward / backward launches, which rows are gathered
ahead) live in mean.py / pool.py / attn.py: subclasses implement `why_not`, `_init_levels`, `_init_reduce`,
`_stage_gather`, `_stage_compute` and `_backward_levels`.
"""
import ctypes
import os

import numpy as np
import torch

from .. import _native as nat
from .. import ops
from ..nn_modules import IdentityPrep, NodeEmbeddingPrep, SparseUniformNeighborSampler, UniformNeighborSampler, \
    _split_activation, concat_combine
from ..store import FeatureStore


class _ReduceDesc(ctypes.Structure):         # mirrors gsage_reduce_desc (include/gsage.h)
    _fields_ = [("src", ctypes.c_void_p), ("stride", ctypes.c_int64), ("out_off", ctypes.c_int64),
                ("S", ctypes.c_int32), ("rows", ctypes.c_int32), ("cols", ctypes.c_int32),
                ("ld", ctypes.c_int32)]


class _PrepDesc(ctypes.Structure):           # mirrors gsage_prep_desc (include/gsage.h)
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("dst_t", ctypes.c_void_p),
                ("rows", ctypes.c_int32), ("cols", ctypes.c_int32), ("dst_ld", ctypes.c_int32),
                ("dst_t_ld", ctypes.c_int32), ("dst_p", ctypes.c_void_p), ("kc_p", ctypes.c_int64),
                ("dst_f32", ctypes.c_int32), ("reserved", ctypes.c_int32)]


def _r8(v):
    return (v + 63) // 64 * 64          # whole 128-byte bf16 lines: enables the LDS-DMA GEMM path


class _ListRunner(object):
    """hipGraph-like handle over a native command list: replay() issues it on the current stream."""

    def __init__(self, cl):
        self.cl = cl

    def replay(self):
        self.cl.replay(ops._stream())


def _r64(v):
    return (int(v) + 63) // 64 * 64


class FusedTrainStep(object):
    """train_step (reference models.py:97-104) without autograd below the loss and without framework glue
    kernels: the machinery common to the mean / pool / attention engines (see the module docstring).

    The arithmetic is that of GSSupervised.train_step.  Parameters and gradients live in flat fp32
    buckets; the model's Parameters become views of them, so `model.state_dict()`, evaluation and
    checkpointing keep working.  `__call__(ids, targets)` has the contract of train_step;
    `load_epoch()` + `step_queue()` walk a device-resident queue of seed batches with no host copies."""

    MEAN_ENGINE = False       # FusedMeanTrainStep only: seed-level kernel, gathers cut around the exchange
    ROW_HIST = 1 << 15        # updates whose constants are kept for deferred table rows (sync_rows() before it wraps)

    # ---- which (model, feature store) pairs an engine covers ----------------------------------------
    @classmethod
    def why_not(cls, model, feats, ddp=None):
        """None when this engine covers (model, feats) -- under the data-parallel handle `ddp` when one is given --,
        else one sentence saying what it does not cover."""
        raise NotImplementedError

    @classmethod
    def supports(cls, model, feats, ddp=None):
        return cls.why_not(model, feats, ddp) is None

    @classmethod
    def head_why_not(cls, model, loss_fn, example_targets, batch, padded, world=1):
        """What the head of (model, loss_fn) costs a caller BEFORE an engine is built (train.py decides between the
        engine and the module path with it): None when a fused head applies (cross-entropy with <= 64 classes and an
        fc input of <= 1024, or the L1 regression head with <= 2048 seeds), else a sentence.  Without a fused head an
        engine still trains (stock torch ops for the head inside the captured step) but cannot ignore padded seeds:
        `padded` says whether the caller will pad short batches (the reference's array_split chunks)."""
        from ..problem import ProblemLosses
        C, D2 = model.fc.weight.shape
        post = _split_activation(list(model.agg_layers.children())[-1].activation)[1]
        probe = torch.linspace(-2.0, 2.0, 12).view(3, 4)          # (no draw from torch's generator: the run's own)
        ident = post is None or torch.equal(post(probe), probe)
        if loss_fn is ProblemLosses.classification and ident and C <= 64 and D2 <= 1024 and \
                example_targets.dtype == torch.int64:
            return None
        if loss_fn is ProblemLosses.regression_mae and ident and C == 1 and 1 < batch <= 2048 and \
                example_targets.dtype == torch.float32:
            if world * batch > 8192:       # gsage_head_l1_sharded keeps the GLOBAL batch's targets in LDS
                return ("the sharded L1 head pairs every prediction with every target of the global batch, which it "
                        "holds in LDS: world x batch = %d x %d exceeds 8192" % (world, batch))
            return None
        if not padded:
            return None
        return ("the head of this problem (loss %s, %d outputs) has no fused kernel, and only a fused head can "
                "ignore the padding of the reference's unequal chunks" % (getattr(loss_fn, "__name__", "?"), int(C)))

    @staticmethod
    def _why_not_common(model, feats, agg_types, what):
        """The checks every engine shares: one aggregator family with the stock concat, ReLU on all but the last
        layer, a sampler whose frontier the fused K1 can produce."""
        layers = list(model.agg_layers.children())
        if not layers or {type(l) for l in layers} not in [{t} for t in agg_types]:
            return "the layers are not all %s aggregators" % what
        if not all(l.combine_fn is concat_combine for l in layers):
            return "a combine_fn other than the concat"
        codes = [_split_activation(l.activation)[0] for l in layers]
        if codes[:-1] != [nat.ACT_RELU] * (len(layers) - 1) or codes[-1] != nat.ACT_NONE:
            return "activations other than ReLU on the hidden layers and identity on the last (train.py:105-118)"
        s = model.train_sampler
        if isinstance(s, SparseUniformNeighborSampler):
            if s.rng not in ("philox", "compat"):
                return "an unknown sampler rng mode %r" % (s.rng,)
        elif isinstance(s, UniformNeighborSampler):
            if not (torch.is_tensor(s.adj) and s.adj.is_cuda):
                return "a dense adjacency that is not on the GPU"
        else:
            return "a sampler class the fused K1 does not know (%s)" % type(s).__name__
        return None

    @staticmethod
    def _why_not_input(model, feats, ddp=None, concat_ok=False):
        """Level-0 rows: an identity prep over a FeatureStore in HBM, or the trainable node-embedding prep
        (nn_modules.py:126-155) -- without features (BASELINE configs[3] / utils/pokec.sh) or, where the engine says
        concat_ok, concatenated behind them (nn_modules.py:152-153: [feats | fc(embedding)])."""
        if isinstance(model.prep, NodeEmbeddingPrep):
            if (feats is None) != (not model.prep.input_dim):
                return "a node-embedding prep whose input_dim disagrees with the features it is given"
            if feats is not None:
                if not concat_ok:
                    return "a node-embedding prep concatenated with features"
                if not isinstance(feats, FeatureStore) or not feats.is_cuda or feats.dim != int(model.prep.input_dim) \
                        or feats.dtype not in (torch.bfloat16, torch.float32):
                    return "features beside the node embedding that are not a bf16 / fp32 FeatureStore in HBM"
            if model.prep.embedding_dim % 8 != 0 or not model.prep.embedding.weight.is_cuda:
                return "an embedding width that is not a multiple of 8, or a table that is not in HBM"
            if ddp is not None and os.environ.get("GSAGE_DENSE_TABLE_ADAM", "0") == "1":
                # the dense mode zeroes only the rows of THIS rank's frontier after the update; rows that another
                # rank's frontier made non-zero (through the all-reduce of the whole bucket) would be re-applied forever
                return "GSAGE_DENSE_TABLE_ADAM=1 (the dense table update) in a data-parallel run: only the deferred row " \
                       "updates exchange and clear every rank's touched rows"
            return None
        if not isinstance(model.prep, IdentityPrep):
            return "a prep class other than identity / node_embedding (%s)" % type(model.prep).__name__
        if not isinstance(feats, FeatureStore):
            return "features that are not a FeatureStore"
        if feats.dtype not in (torch.bfloat16, torch.float32) or not feats.is_cuda:
            return "a feature table that is not bf16 / fp32 in HBM"
        return None

    def __init__(self, model, feats, loss_fn, example_ids, example_targets, ddp=None, capture=True,
                 warmup=2, pipelined=False, eval_only=False):
        """capture: "cmdlist" (or True) / "graph" / False; pipelined: two batches in flight on two streams (Philox
        sampler only); eval_only: the forward launches over the validation sampler (evaluate_fold)."""
        if not type(self).supports(model, feats, ddp):
            raise ValueError("%s does not cover this (model, feature store): %s"
                             % (type(self).__name__, type(self).why_not(model, feats, ddp)))
        if not (torch.is_tensor(example_ids) and example_ids.is_cuda and example_ids.dtype == torch.int64
                and example_ids.dim() == 1):
            raise ValueError("example_ids must be a CUDA int64 vector of seed ids (one batch)")
        if not (torch.is_tensor(example_targets) and example_targets.is_cuda
                and int(example_targets.shape[0]) == int(example_ids.shape[0])):
            raise ValueError("example_targets must be a CUDA tensor with one row per seed")
        # eval_only: the engine's FORWARD launches over the model's VALIDATION sampler (models.py:41-44: the full
        # graph, n_val_samples), for train.evaluate's folds (evaluate_fold below).  The Parameters stay where they are
        # (a training engine's bucket, FlatAdam's, or their own storage): only the operand copies are this engine's.
        self.eval_only = bool(eval_only)
        assert not (self.eval_only and (ddp is not None or pipelined)), "eval_only: single process, sequential"
        self._init_common(model, feats, loss_fn, example_ids, example_targets, ddp, pipelined)
        self._init_levels(example_ids, example_targets)
        self._init_head(loss_fn, example_targets)
        if self.eval_only:
            Cc, D2c = model.fc.weight.shape
            if self.fused_l1:                      # (what _install_reduce gives the training engines' heads)
                self.l1_scratch = torch.zeros(nat.lib().gsage_head_l1_scratch(self.B, D2c), dtype=torch.float32,
                                              device=self.dev)
            elif not self.fused_head:
                self.head_stage = torch.zeros(Cc * D2c + Cc, dtype=torch.float32, device=self.dev)
            self._param_ptrs = self._current_param_ptrs()
            self.refresh_weights()
        else:
            self._init_reduce()
        self._finish_init(capture, warmup)

    # ---- construction, in five steps (subclasses override the aggregator-specific ones) -----------
    def _init_common(self, model, feats, loss_fn, example_ids, example_targets, ddp, pipelined):
        """Everything that does not depend on the aggregator: exchange op, frontier geometry, flat
        parameter / gradient / Adam buckets (Parameters become views), device counters."""
        self.model, self.store, self.loss_fn, self.ddp = model, feats, loss_fn, ddp
        # pipelined: batch k+1's sampling + gathers (which do not depend on the weights) run on a
        # second graph branch WHILE batch k's GEMMs / backward / Adam run; results are identical
        # to the sequential order, `__call__` then returns the predictions of the previous batch.
        self.pipelined = bool(pipelined)
        self.nset = 2 if self.pipelined else 1
        # trainable node-embedding prep: the level-0 rows are weights (computed per step from the current table)
        self.emb = isinstance(model.prep, NodeEmbeddingPrep)
        # ... E columns wide, behind the D0 feature columns of the row when the problem has features too
        self.E = int(model.prep.embedding_dim) if self.emb else 0
        self.D0 = int(feats.dim) if (self.emb and feats is not None) else 0
        self.lazy_rows = self.emb and os.environ.get("GSAGE_DENSE_TABLE_ADAM", "0") != "1" and not self.eval_only
        # the step's collectives: issued by the library (RCCL through gsage_comm_*, a node of the step's list) when the
        # handle carries a native communicator, else torch.distributed (gloo in the tests) through a host-call node
        self.world = int(ddp.world) if ddp is not None else 1
        self.comm = getattr(ddp, "comm", None) if ddp is not None else None
        self._host_cbs = []                       # ctypes callbacks recorded into lists (kept alive with the engine)
        self._front_ready, self._qstep = False, 0
        self.g_prime, self.g_qfront, self.g_queue = None, None, None
        self._tail_gather, self._tail_rows = None, 0
        self._reduce_op = None
        if ddp is not None:
            # averaging inside the collective saves a launch; fall back to divide-then-sum where
            # the backend has no AVG
            self._reduce_op = torch.distributed.ReduceOp.AVG
            try:
                probe = torch.ones(8, device=next(model.parameters()).device)
                torch.distributed.all_reduce(probe, op=self._reduce_op)
                if abs(float(probe[0]) - 1.0) > 1e-6:
                    raise RuntimeError("AVG returned %r" % float(probe[0]))
            except Exception:
                self._reduce_op = torch.distributed.ReduceOp.SUM
        dev = feats.device if feats is not None else next(model.parameters()).device
        self.dev = dev
        # storage type of features, activations and weight operand copies
        self.tdt = feats.dtype if feats is not None else ops.torch_dtype()
        self.code = nat.BF16 if self.tdt == torch.bfloat16 else nat.F32
        self.esz = 2 if self.tdt == torch.bfloat16 else 4
        self.sel, self.sel_queue = None, None     # caller-supplied sampler draws (set_sel / load_epoch)
        self.layers = list(model.agg_layers.children())
        L = self.L = len(self.layers)
        self.post = _split_activation(self.layers[-1].activation)[1]
        the_sampler = model.val_sampler if self.eval_only else model.train_sampler
        self.fan = [1] + [fn.keywords["n_samples"] for fn in (model.val_sample_fns if self.eval_only
                                                              else model.train_sample_fns)]
        if isinstance(the_sampler, UniformNeighborSampler):
            # the dense sampler keeps `perm[:n_samples]` of the adjacency's K columns (nn_modules.py:43-49): asking for
            # more than K yields K -- the frontier's real geometry (run.sh:8-10's defaults 25 / 10 on a K = 16 file)
            K = int(the_sampler.adj.size(1))
            self.fan = [1] + [min(int(n), K) for n in self.fan[1:]]
        B = self.B = int(example_ids.shape[0])
        self.size = [B]
        for k in range(1, L + 1):
            self.size.append(self.size[-1] * self.fan[k])
        self.off = [0]
        for k in range(L + 1):
            self.off.append(self.off[-1] + self.size[k])          # off[k] = first row of hop k
        self.sampler = the_sampler
        self.csr = self.sampler.csr(dev)          # store.DeviceCSR, or store.DenseAdj for the dense sampler
        # where the sampler's draws come from:
        #   philox  in-kernel counter RNG (throughput runs)
        #   compat  numpy's legacy MT19937 stream, consumed ON THE DEVICE in the reference's order: the frontier is
        #           bit-identical to the eager compat run and to the reference (queue mode: load_epoch fills the
        #           epoch's draws with gsage_mt_choice_device; per call: one k_mt_choice launch per hop)
        #   dense   the reference's default sampler: one torch.randperm per sampler call, its head = the columns
        #           every parent keeps (nn_modules.py:43-49)
        self.dense = isinstance(self.sampler, UniformNeighborSampler)
        self.draws = "dense" if self.dense else self.sampler.rng
        assert not (self.pipelined and self.draws != "philox"), \
            "pipelined=True keeps two batches in flight: only the counter-based sampler can run ahead of the host"
        # number of int32 draws one batch's frontier consumes
        self.n_sel = sum(self.fan[1:]) if self.dense else self.off[L + 1] - self.off[1]
        if self.draws != "philox":
            self.sel = torch.zeros(self.n_sel, dtype=torch.int32, device=dev)
        self._in_list = False                     # True while a native command list is being recorded
        self._sel_user = False                    # set_sel(): the caller's draws win over the engine's own
        # live seeds of the batch (<= B): the reference's `iterate` yields near-equal chunks, never exactly
        # batch_size (problem.py:141-153), so a batch may be one seed short of the recorded geometry.  Padded seeds
        # run through the forward like any other; the head gives them no loss and no gradient (n_valid).
        self.n_valid = torch.full((1,), B, dtype=torch.int32, device=dev)
        self._nv_host, self.nv_queue = B, None
        self.off_host = (ctypes.c_int64 * 6)(*([int(v) for v in self.off[:L + 1]] + [0] * (5 - L)))
        self.fan_host = (ctypes.c_int32 * 6)(*([int(v) for v in self.fan[:L + 1]] + [1] * (5 - L)))

        # ---- flat parameter / gradient / Adam buckets; Parameters become views ----------------
        settle = getattr(model, "_settle_rows", None)
        if settle is not None:                    # an earlier engine's deferred table rows (sync_rows)
            settle()
        self.params = [p for p in model.parameters() if p.requires_grad]
        sizes = [p.numel() for p in self.params]
        self.poff = [0]
        for n in sizes:
            self.poff.append(self.poff[-1] + n)
        total = self.poff[-1]
        self.pidx = {id(p): i for i, p in enumerate(self.params)}
        self.step = torch.zeros(1, dtype=torch.int64, device=dev)
        if self.eval_only:
            self.flat_p = self.flat_g = self.flat_m = self.flat_v = None
        else:
            self.flat_p = torch.cat([p.detach().reshape(-1).float() for p in self.params]).contiguous()
            self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
            self.flat_m = torch.zeros_like(self.flat_g)
            self.flat_v = torch.zeros_like(self.flat_g)
            for p, o, n in zip(self.params, self.poff, sizes):
                p.data = self.flat_p[o:o + n].view_as(p)
                p.grad = self.flat_g[o:o + n].view_as(p)
            import weakref
            model._engine = weakref.ref(self)          # models.py hands the Adam state back through this
            self.lr = torch.tensor([float(model.lr)], dtype=torch.float32, device=dev)
            self.wd = float(model.optimizer.param_groups[0].get("weight_decay", 0.0))
            self.partial = torch.zeros(nat.lib().gsage_adam_partials(total), dtype=torch.float32, device=dev)
            self.gnorm = torch.zeros(1, dtype=torch.float32, device=dev)
            self._import_optimizer_state()

        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        self.preds = None
        self.n_calls = 0
        # optional device-resident batch queue (load_epoch): the graph then needs no per-step copies
        self.queue = None
        self._q_ids = None                            # frontier of the batch a queue-mode compute stage works on
        self.batch_idx = torch.zeros(1, dtype=torch.int64, device=dev)
        self.ids_set = [torch.zeros(self.off[L + 1], dtype=torch.int64, device=dev) for _ in range(self.nset)]
        self.tg_set = [example_targets.clone() for _ in range(self.nset)]
        self.ids_set[0][:B].copy_(example_ids)


    def _will_fuse_head(self, example_targets):
        from ..problem import ProblemLosses
        C, D2 = self.model.fc.weight.shape
        probe = torch.randn(3, 4, device=self.dev)
        ident = self.post is None or torch.equal(self.post(probe), probe)
        return bool(self.loss_fn is ProblemLosses.classification and ident and C <= 64 and D2 <= 1024 and
                    example_targets.dtype == torch.int64)

    def _init_head(self, loss_fn, example_targets):
        """Classification head as one fused kernel pair when it applies (else stock torch autograd)."""
        model, dev, L, B = self.model, self.dev, self.L, self.B
        C, D2 = model.fc.weight.shape
        self.fused_head = self._will_fuse_head(example_targets)
        self.fused_tail = bool(self._will_fuse_tail(example_targets))
        if self.fused_head:
            assert nat.lib().gsage_head_ce_scratch(B, C, D2) == nat.lib().gsage_mean_tail_ce_scratch(B, C) \
                or not self.fused_tail
            n_scr = nat.lib().gsage_head_ce_scratch(B, C, D2)
            if self.fused_tail and self._tail_on_mfma():       # one partial row per 16 seeds instead of per 4
                n_scr = nat.lib().gsage_mean_tail_mfma_scratch(B, C)
            self.head_scratch = torch.zeros(n_scr, dtype=torch.float32, device=dev)
            self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
            self.preds = torch.zeros(B, C, dtype=torch.float32, device=dev)
        # the regression head of the Pokec problem (F.l1_loss with the reference's [B,1]-vs-[B] broadcast) as one kernel
        from ..problem import ProblemLosses
        probe = torch.randn(3, 4, device=self.dev)
        ident = self.post is None or torch.equal(self.post(probe), probe)
        self.fused_l1 = bool(self.loss_fn is ProblemLosses.regression_mae and C == 1 and ident and self.B <= 2048 and
                             example_targets.dtype == torch.float32 and example_targets.numel() == self.B and
                             self.B > 1 and os.environ.get("GSAGE_TORCH_HEAD", "0") != "1")
        if self.fused_l1:
            self.preds = torch.zeros(self.B, 1, dtype=torch.float32, device=self.dev)
            if self.world > 1:
                # the reference's L1 loss pairs EVERY prediction with EVERY target of the batch (problem.py:39-42:
                # [B,1] against [B] broadcasts): a shard needs the global batch's targets -- gathered once per step,
                # 4 bytes per seed -- for its rows' share of the global loss (gsage_head_l1_sharded)
                assert self.world * self.B <= 8192, "gsage_head_l1_sharded: the global batch must stay <= 8192 seeds"
                self.tg_all = torch.zeros(self.world * self.B, dtype=torch.float32, device=self.dev)

    def _install_reduce(self, rdesc):
        """Append the head's gradient source, check that every parameter is covered, upload."""
        model, dev = self.model, self.dev
        f32 = torch.float32
        Cc, D2c = model.fc.weight.shape
        ifc = self.pidx[id(model.fc.weight)]
        assert self.pidx[id(model.fc.bias)] == ifc + 1
        if self.fused_head:
            width = Cc * D2c + Cc + 1
            rdesc.append(_ReduceDesc(self.head_scratch.data_ptr(), width, self.poff[ifc],
                                     self.head_scratch.numel() // width, 1, Cc * D2c + Cc, width))
        elif self.fused_l1:                               # gsage_head_l1: one partial row [dW | db | loss] per 16 seeds
            width, n_wg = D2c + 2, (self.B + 15) // 16
            self.l1_scratch = torch.zeros(nat.lib().gsage_head_l1_scratch(self.B, D2c), dtype=f32, device=dev)
            rdesc.append(_ReduceDesc(self.l1_scratch.data_ptr(), width, self.poff[ifc], n_wg, 1, D2c + 1, width))
        else:
            self.head_stage = torch.zeros(Cc * D2c + Cc, dtype=f32, device=dev)
            rdesc.append(_ReduceDesc(self.head_stage.data_ptr(), 0, self.poff[ifc], 1, 1, Cc * D2c + Cc,
                                     Cc * D2c + Cc))
        covered = sum(d.rows * d.cols for d in rdesc)
        uncovered = int(self.table.numel()) if self.emb else 0     # the scatter-added embedding table
        assert covered + uncovered == self.flat_p.numel(), "every parameter must be covered by a gradient source"
        self.rdescs = torch.frombuffer(bytearray(bytes((_ReduceDesc * len(rdesc))(*rdesc))),
                                       dtype=torch.uint8).to(dev)
        self.n_rdesc = len(rdesc)
        self._rdesc_max_S = max(int(d.S) for d in rdesc)
        self.r_max = max(d.rows * d.cols for d in rdesc)
        self.n_partial = nat.lib().gsage_finalize_partials(self.n_rdesc, self.r_max)
        self.partial = torch.zeros(max(self.n_partial, self.partial.numel()), dtype=f32, device=dev)
        self.refresh_weights()

    def _debug_addresses(self):
        """GSAGE_DEBUG_ADDR=1: the address range of every device buffer the engine's kernels are handed, on stderr
        -- what a 'Memory access fault ... on address' line of the runtime is matched against."""
        seen = {}
        def visit(name, v, depth=0):
            if torch.is_tensor(v):
                if v.is_cuda and v.numel():
                    st = v.untyped_storage()
                    seen.setdefault(st.data_ptr(), (name, st.nbytes()))
            elif isinstance(v, (list, tuple)) and depth < 2:
                for i, e in enumerate(v):
                    visit("%s[%d]" % (name, i), e, depth + 1)
            elif depth < 1 and hasattr(v, "__dict__") and not isinstance(v, torch.nn.Module):
                for k, e in vars(v).items():
                    visit("%s.%s" % (name, k), e, depth + 1)
        for k, v in vars(self).items():
            visit(k, v)
        for k, v in self.model.state_dict().items():
            visit("model." + k, v)
        text = "[gsage addr] %s B=%d\n" % (type(self).__name__, self.B)
        for ptr in sorted(seen):
            text += "[gsage addr]   %#x .. %#x  %s\n" % (ptr, ptr + seen[ptr][1], seen[ptr][0])
        os.write(2, text.encode())          # (fd 2, not sys.stderr: survives a test runner's sys-level capture)

    def _finish_init(self, capture, warmup):
        ddp = self.ddp
        if os.environ.get("GSAGE_DEBUG_ADDR", "0") == "1":
            self._debug_addresses()
        if self.eval_only:
            return self._finish_init_eval(capture, warmup)
        # warm-up (library handles, allocator) with state restored afterwards, then capture
        saved = self.flat_p.clone()
        saved_opt = (self.flat_m.clone(), self.flat_v.clone(), self.step.clone())     # (zeros, or an imported state)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._run_sequential(0)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.flat_p.copy_(saved)
        for t, v in zip((self.flat_m, self.flat_v, self.step), saved_opt):
            t.copy_(v)
        slots = getattr(self, "_norm_slots", None)     # (tags of the warm-up's update numbers must not meet the run's)
        for t in (self.counter,) + tuple(getattr(self, "_warm_reset", ())) + ((slots,) if slots is not None else ()):
            t.zero_()
        if self.lazy_rows and int(self.step.item()) > 0:      # deferred table rows: every row is current at that count
            self.row_last.fill_(int(self.step.item()))
        self.refresh_weights()
        torch.cuda.synchronize()
        self.g_main, self.g_opt, self.g_front = None, None, None
        # pipelined mode runs the two stages on two streams (two hardware queues): parallel
        # branches inside ONE hipGraph were measured to be serialised by the runtime.
        self.s_front = torch.cuda.Stream() if self.pipelined else None
        self.s_back = torch.cuda.Stream() if self.pipelined else None
        self.ev_front = [torch.cuda.Event() for _ in range(self.nset)]
        self.ev_back = [torch.cuda.Event() for _ in range(self.nset)]
        # capture: False = eager launches from Python; "cmdlist" (or True) = native command lists
        # (include/gsage.h: recorded launches replayed by one C call, no device-side start-up gap);
        # "graph" = hipGraphs.
        self.capture_mode = {True: "cmdlist", False: None, None: None}.get(capture, capture)
        assert self.capture_mode in (None, "cmdlist", "graph")
        if self.capture_mode == "cmdlist" and not (self.fused_head or self.fused_l1):
            self.capture_mode = "graph"              # the stock-torch head cannot be recorded
        if self.capture_mode == "graph" and self.world > 1 and self.fused_l1:
            self.capture_mode = "cmdlist"            # (the sharded L1 head gathers the global targets INSIDE the step:
            #                                           a collective cannot sit in a hipGraph, it can in a list)
        self._pool = None
        if self.capture_mode:
            self._record_main()
        torch.cuda.synchronize()

    def _record_main(self):
        """(Re-)record the per-call command lists / graphs of __call__."""
        ddp = self.ddp
        self._host_cbs = self._cbs_main = []      # (callbacks of the lists recorded below live as long as those lists)
        if self.capture_mode == "graph":
            # re-recording: let go of the old hipGraphs (and their private pool) before capturing new ones
            self.g_main, self.g_opt, self.g_front, self._pool = None, None, None, None
            torch.cuda.synchronize()
        self.g_main = []
        if self.pipelined:
            self.g_front = [self._record(lambda st_=st_: self._stage_sample_gather(st_), self.s_front)
                            for st_ in range(2)]

        # data-parallel: on command lists the exchange is a node of the step's ONE list (_stage_exchange); hipGraphs
        # cannot hold it, so "graph" mode (and the two-stream pipelined engine) keep it between two recordings
        one_list = self._one_list_ddp()

        def main(st_):
            if not self.pipelined:
                self._stage_sample_gather(0)
            self._stage_compute(st_)
            if one_list:
                self._stage_exchange()
            if ddp is None or one_list:
                self._stage_opt()
        for st_ in range(self.nset):
            self.g_main.append(self._record(lambda st_=st_: main(st_),
                                            self.s_back if self.pipelined else None))
        self.g_opt = None
        if ddp is not None and not one_list:
            self.g_opt = self._record(self._stage_opt, self.s_back if self.pipelined else None)

    def _one_list_ddp(self):
        """data-parallel step as ONE command list (collectives as nodes)?  GSAGE_DDP_ONE_LIST=0: round 3's three
        lists around a torch.distributed call."""
        return bool(self.ddp is not None and self.capture_mode == "cmdlist" and not self.pipelined and
                    os.environ.get("GSAGE_DDP_ONE_LIST", "1") == "1")

    def _ddp_overlap(self):
        """run the exchange on the list's side stream, beside the bulk of the next batch's gathers (GSAGE_DDP_OVERLAP=1)?
        Default: no -- inline, on the step's own stream.  Measured on the MI355X (profiles/r04_ddp_1rank_timeline_*,
        DESIGN.md section 6): a fork to the side stream delays the collective's kernel by ~13 us, the fork and the
        join each leave a ~7 us hole on the main stream, and the launch that follows the join is as long as its Adam
        role (~19 us: nothing left to hide it behind) -- 46 us of fixed cost to hide at most the ~23 us the bulk of
        the gathers takes.  Inline, the step is the single-GPU step plus the collective: everything else (Adam with
        the norm formed in the launch, the sampler) still rides in the ONE gather launch."""
        return os.environ.get("GSAGE_DDP_OVERLAP", "0") == "1"

    # ---- helpers ------------------------------------------------------------------------------
    def _record(self, fn, stream=None):
        """Record the launches of fn() once; returns an object whose replay() re-issues them on the
        current stream (command list) or on the capture stream (hipGraph)."""
        if self.capture_mode == "cmdlist":
            self._in_list = True
            try:
                with nat.CommandList.record() as cl:
                    fn()
            finally:
                self._in_list = False
            return _ListRunner(cl)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self._pool, stream=stream):
            fn()
        self._pool = g.pool()
        return g

    def refresh_weights(self):
        """Rebuild the bf16 operand copies from the fp32 Parameters.  Adam keeps them current;
        call this after changing the weights from outside (load_state_dict, manual edits)."""
        nat.check(nat.lib().gsage_prep_weights(self.descs.data_ptr(), self.n_desc, self.max_elems,
                                               None, 0, None, 0, ops._stream()), "prep_weights")

    # ---- optimizer state for whoever trains next / writes a checkpoint -----------------------------
    def holds_parameters(self):
        """True while the model's Parameters still are views of this engine's flat bucket."""
        base, es = self.flat_p.data_ptr(), 4
        return all(p.data_ptr() == base + o * es for p, o in zip(self.params, self.poff))

    def _settled(self):
        self.sync_rows()

    def optimizer_state_dict(self):
        """torch.optim.Adam's format (as optim.FlatAdam.state_dict): the engine's exp_avg / exp_avg_sq / step."""
        self._settled()
        state, n = {}, int(self.step.item())
        if n > 0:
            step = torch.tensor(float(n))
            for i, (p, o) in enumerate(zip(self.params, self.poff)):
                k = p.numel()
                state[i] = {"step": step.clone(), "exp_avg": self.flat_m[o:o + k].view_as(p).clone(),
                            "exp_avg_sq": self.flat_v[o:o + k].view_as(p).clone()}
        g = self.model.optimizer.param_groups[0]
        group = {k: v for k, v in g.items() if k != "params"}
        group.update({"lr": float(self.lr.item()), "weight_decay": self.wd, "params": list(range(len(self.params)))})
        return {"state": state, "param_groups": [group]}

    def export_optimizer_state(self, flat_adam):
        """Copy exp_avg / exp_avg_sq / the step count into an optim.FlatAdam over the same Parameters (models.py
        calls this when GSSupervised.train_step takes over from the engine: Adam's bias correction and moments
        continue instead of restarting)."""
        assert [p.numel() for p in flat_adam.params] == [p.numel() for p in self.params], \
            "export_optimizer_state: different parameter lists"
        self._settled()
        flat_adam.flat_m.copy_(self.flat_m)
        flat_adam.flat_v.copy_(self.flat_v)
        flat_adam.step_count.copy_(self.step)

    def _import_optimizer_state(self):
        """An engine built AFTER the model has trained through `model.optimizer` (GSSupervised.train_step: FlatAdam
        or a torch.optim.Adam) continues from that optimizer's exp_avg / exp_avg_sq / step count instead of
        restarting Adam's moments and bias correction from zero."""
        try:
            sd = self.model.optimizer.state_dict()
        except Exception:
            return
        st = sd.get("state") or {}
        if not st:
            return
        if len(st) != len(self.params):
            import warnings
            warnings.warn("gsage: the optimizer's state does not match the trainable parameters; the fused engine "
                          "starts Adam's moments from zero")
            return
        steps = set()
        for i, (p, o) in enumerate(zip(self.params, self.poff)):
            e = st.get(i, st.get(str(i)))
            k = p.numel()
            self.flat_m[o:o + k].copy_(e["exp_avg"].reshape(-1))
            self.flat_v[o:o + k].copy_(e["exp_avg_sq"].reshape(-1))
            steps.add(int(e["step"]))
        assert len(steps) == 1, "the engines keep ONE step count: per-parameter counts differ (%r)" % (steps,)
        self.step.fill_(steps.pop())

    def _grad_slice(self, prm):
        i = self.pidx[id(prm)]
        return self.flat_g[self.poff[i]:self.poff[i + 1]]

    def _linear(self, A, lda, a_rows, a_g0, W, ldw, C, c_dtype, ldc, M, N, K, act, a_gs, w_gs, c_gs):
        ops._linear_launch(A, lda, a_rows, a_g0, W, ldw, None, C, ldc, M, N, K, act, 2, a_gs, w_gs,
                           c_gs, self.code, c_dtype)

    # ---- stages (each is a sequence of kernel launches on the current stream) ---------------
    def _hops_desc(self, ids, ahead, counters=None):
        """gsage_hops_desc that samples a whole frontier into `ids`; ahead=True: the batch AFTER the
        one the device counters point at (call_base / batch_base offsets, the counters themselves
        are not touched).  counters: (philox call counter, batch index) to read instead of the step's own."""
        L = self.L
        ctr, bidx = counters if counters is not None else (self.counter, self.batch_idx)
        d = nat.HopsDesc()
        if self.dense:
            d.dense_adj, d.dense_ld, d.n_rows = self.csr.adj.data_ptr(), self.csr.K, self.csr.n_rows
        else:
            d.rowptr, d.col, d.n_rows = self.csr.rowptr.data_ptr(), self.csr.col.data_ptr(), self.csr.n_rows
        d.ids, d.B, d.n_hops = ids.data_ptr(), self.B, L
        for k in range(5):
            d.fan[k] = int(self.fan[k + 1]) if k < L else 1
        d.max_deg, d.seed = self.csr.max_deg, int(getattr(self.sampler, "seed", 0))
        d.call_ctr, d.call_base = ctr.data_ptr(), L * int(ahead)       # (ahead: how many batches past the counters)
        # (evaluation: every rank scores the WHOLE fold with the samples a single process would draw)
        d.rank = 0 if self.eval_only else getattr(self.sampler, "shard", (0, 1))[0]
        d.seed_queue = self.queue[0].data_ptr() if self.queue else None
        d.batch_idx = bidx.data_ptr() if self.queue else None
        d.batch_base, d.n_batches = int(ahead), (self.queue[2] if self.queue else 0)
        d.err_flag = self.csr.err_flag.data_ptr()
        if self.queue and self.sel_queue is not None:
            d.sel, d.sel_stride = self.sel_queue.data_ptr(), int(self.sel_queue.shape[1])
        elif self.sel is not None:
            d.sel, d.sel_stride = self.sel.data_ptr(), 0
        return d

    def _hop_draw_counts(self, b):
        """[(offset into a batch's draws, number of draws)] per hop for a batch of b live seeds: the reference's
        sampler call of hop k draws b * fan_1 ... fan_k values (sparse, nn_modules.py:88) -- padded seeds draw none."""
        out, live = [], b
        for k in range(1, self.L + 1):
            live *= self.fan[k]
            out.append((self.off[k] - self.off[1], live))
        return out

    def _draw_batch(self, b):
        """compat / dense draws of ONE batch into self.sel, consumed from the same generators, in the same order,
        as the reference's sampler calls of that batch (nn_modules.py:88 / :44)."""
        if self.draws == "dense":
            keep = torch.cat([UniformNeighborSampler.draw_keep(self.csr.K, n) for n in self.fan[1:]])
            self.sel.copy_(keep.to(torch.int32), non_blocking=True)
        elif self.draws == "compat":
            from ..helpers import legacy_stream
            st = legacy_stream.acquire(self.dev)
            segs = [(o, c) for o, c in self._hop_draw_counts(b)]
            ops.mt_choice_segments(st, self.csr.max_deg, segs, self.sel)

    def _draw_epoch(self, n_valid):
        """The draws of a whole epoch queue, batch after batch (the order the reference's training loop consumes
        its generators in): -> int32 [n_batches, n_sel] on the device."""
        nb = len(n_valid)
        if self.draws == "dense":
            keep = torch.stack([torch.cat([UniformNeighborSampler.draw_keep(self.csr.K, n) for n in self.fan[1:]])
                                for _ in range(nb)])
            return keep.to(device=self.dev, dtype=torch.int32).contiguous()
        from ..helpers import legacy_stream
        st = legacy_stream.acquire(self.dev)
        out = torch.zeros(nb, self.n_sel, dtype=torch.int32, device=self.dev)
        segs = [(i * self.n_sel + o, c) for i, b in enumerate(n_valid) for o, c in self._hop_draw_counts(int(b))]
        ops.mt_choice_segments(st, self.csr.max_deg, segs, out)
        return out

    def _stage_sample(self, s, ids=None, ahead=False, counters=None):
        """K1: every hop in one launch, frontier written in place into the concatenated ids."""
        d = self._hops_desc(self.ids_set[s] if ids is None else ids, ahead, counters)
        nat.check(nat.lib().gsage_sample_hops(ctypes.addressof(d), ops._stream()), "sample_hops")

    def _stage_sample_gather(self, s):
        """K1 for every hop + the level-0 gathers of batch set `s`; independent of the weights."""
        self._stage_sample(s)
        self._stage_gather(s)
        # the next batch's samples use the next L Philox call indices (sequential mode: ticked by
        # gsage_finalize_grads instead of a launch of its own)
        if self.pipelined:
            nat.check(nat.lib().gsage_counter_add(self.counter.data_ptr(), self.L, ops._stream()), "counter_add")

    def _adam_desc(self):
        d = nat.AdamDesc()
        d.p, d.g, d.m, d.v = (self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.flat_m.data_ptr(),
                              self.flat_v.data_ptr())
        d.n, d.partial, d.lr, d.step = (self.flat_p.numel(), self.partial.data_ptr(), self.lr.data_ptr(),
                                        self.step.data_ptr())
        d.beta1, d.beta2, d.eps, d.weight_decay, d.max_norm = 0.9, 0.999, 1e-8, self.wd, 5.0
        d.norm_out, d.step_is_current = self.gnorm.data_ptr(), 1
        d.n_partial_ready = 0 if self.ddp is not None else self.n_partial
        d.prep_descs, d.n_prep = self.descs.data_ptr(), self.n_desc
        d.tick1, d.inc1, d.tick2, d.inc2 = None, 0, None, 0     # the finalisation (or K5b) ticks the counters
        if self._fold_finalize():
            # no finalisation launch: the update's workgroups sum the partial buffers of their own elements, form the
            # norm among themselves and store the (clipped) gradient (gsage_adam_desc.reduce_descs)
            d.n_partial_ready = 0
            d.norm_slots = self._slots().data_ptr()
            d.reduce_descs, d.n_reduce = self.rdescs.data_ptr(), self.n_rdesc
            return d
        if self.ddp is not None and self._norm_in_launch():
            # data-parallel: the norm of the AVERAGED gradient is formed by the update's own workgroups inside the
            # launch that carries Adam (gsage_adam_desc.norm_slots) -- no norm launch behind the collective
            d.norm_slots = self._slots().data_ptr()
        return d

    def _slots(self):
        if getattr(self, "_norm_slots", None) is None:
            self._norm_slots = torch.zeros(1024, dtype=torch.int64, device=self.dev)
        return self._norm_slots

    def _meet_fits(self):
        """The in-launch norm is a meeting of the update's workgroups (one per 1 024 parameters): every one of them
        must be RESIDENT at once in the launch that carries them -- asked of the device (occupancy of the gather
        launch with the sampler role's LDS, times the CU count, one workgroup per CU kept as margin), not assumed."""
        if getattr(self, "_meet_ok", None) is None:
            n_wg = -(-(-(-self.flat_p.numel() // 4)) // 256)
            widest, width = 1, 1
            for k in range(1, self.L + 1):
                width *= self.fan[k]
                widest = max(widest, width)
            spw = int(os.environ.get("GSAGE_HOPS_SPW", "1"))
            lds = 16 * (spw if 1 <= spw <= 16 else 1) * widest
            cap = int(nat.lib().gsage_gather_adam_capacity(self.code, lds))
            self._meet_ok = bool(n_wg <= 1024 and n_wg <= cap)
        return self._meet_ok

    def _norm_in_launch(self):
        """can the launch that carries Adam form the gradient norm itself?  (the mean engine's gather launch)"""
        return bool(self.MEAN_ENGINE and not self.emb and self._meet_fits() and
                    os.environ.get("GSAGE_DDP_NORM_IN_LAUNCH", "1") == "1")

    def _fold_finalize(self):
        """Single GPU: no finalisation launch -- the update's workgroups sum K5b's partial tiles and the head's
        partial rows for the 1 024 elements each is about to update (gsage_adam_desc.reduce_descs), the norm is
        formed among them (norm_slots), and the step's ticks ride in the K5b launch (gsage_wgrad_ticks_next).
        Four launches per step instead of five at BASELINE configs[1].  Not with a process group (the exchange
        wants the flat bucket), a trainable table (its gradient is scatter-added), or more partial buffers per
        element than the update can sum inside its launch without becoming its longest role.  Every launch mode of
        an engine takes the same path (per call, queue, pipelined): their results stay bit-identical.
        GSAGE_FOLD_FINALIZE: 0 (default) = never, 1 = the mean engine, all = every engine that qualifies.
        OPT-IN, because it measured slower on the MI355X: inside the launch that also
        gathers, an update workgroup's six to eight dependent rounds of partial loads (16 in flight per lane is what
        the launch's 72-register cap leaves) take ~28 us against 6.5 + 16 for the finalisation launch and the
        update without them -- 0.0916 against 0.0836 ms/step at configs[1]."""
        if getattr(self, "_fold", None) is None:
            mode = os.environ.get("GSAGE_FOLD_FINALIZE", "0")
            ok = (mode in ("1", "all") and (self.MEAN_ENGINE or mode == "all") and self.ddp is None and not self.emb
                  and (self.fused_head or self.fused_l1) and getattr(self, "n_rdesc", 99) <= 16
                  and getattr(self, "_rdesc_max_S", 1 << 30) <= 32 and self._meet_fits())
            self._fold = bool(ok)
        return self._fold

    def _wgrad_ticks(self):
        """(before the step's K5b launch) without a finalisation launch the step's ticks ride in K5b"""
        if self._fold_finalize():
            nat.check(nat.lib().gsage_wgrad_ticks_next(
                self.step.data_ptr(), None if self.pipelined else self.counter.data_ptr(), self.L,
                self.batch_idx.data_ptr() if self.queue else None, 1), "wgrad_ticks_next")
            self._ticks_issued = True

    def _head_live_rows(self):
        """tell the next head launch how many seeds of the batch are live (padded chunks, see _pad_batch)"""
        if os.environ.get("GSAGE_NO_NVALID", "0") == "1":      # (measurement: the head without the live-row word)
            return
        nat.check(nat.lib().gsage_head_n_valid_next(self._nv_ptr()), "head_n_valid_next")

    def _stage_head_ce(self, s):
        """normalize + fc + softmax cross-entropy + their gradients (models.py:90-91,100) in one launch: predictions,
        d loss / d embedding into dc[L-1], one partial row [dW | db | loss] per workgroup for the finalisation."""
        m, L, B = self.model, self.L, self.B
        C, D2 = m.fc.weight.shape
        E = self.hout[L - 1]
        tg = self.queue[1] if self.queue else self.tg_set[s].view(-1)
        self._head_live_rows()
        nat.check(nat.lib().gsage_head_ce(E.data_ptr(), E.stride(0), m.fc.weight.data_ptr(), m.fc.bias.data_ptr(),
                                          tg.data_ptr(), B, C, D2, self.preds.data_ptr(), self.dc[L - 1].data_ptr(),
                                          self.code, self.dc[L - 1].stride(0), None, None, None,
                                          self.head_scratch.data_ptr(),
                                          self.batch_idx.data_ptr() if self.queue else None,
                                          self.queue[2] if self.queue else 0, ops._stream()), "head_ce")

    def _stage_head_l1(self, s):
        """normalize + fc + F.l1_loss with the reference's [B,1]-vs-[B] broadcast (problem.py:39-42) + gradients"""
        m, L = self.model, self.L
        out = self.hout[L - 1]
        if self.world > 1:
            assert self._nv_host == self.B, "data-parallel batches have one fixed size (no padded seeds)"
            self._x_all_gather(self.tg_set[s].view(-1), self.tg_all)
            nat.check(nat.lib().gsage_head_l1_sharded(
                out.data_ptr(), out.stride(0), m.fc.weight.data_ptr(), m.fc.bias.data_ptr(), self.tg_all.data_ptr(),
                self.tg_all.numel(), self.B, out.shape[1], self.preds.data_ptr(), self.dc[L - 1].data_ptr(), self.code,
                self.dc[L - 1].stride(0), self.l1_scratch.data_ptr(), ops._stream()), "head_l1_sharded")
            return
        self._head_live_rows()
        nat.check(nat.lib().gsage_head_l1(out.data_ptr(), out.stride(0), m.fc.weight.data_ptr(), m.fc.bias.data_ptr(),
                                          self.tg_set[s].data_ptr(), self.B, out.shape[1], self.preds.data_ptr(),
                                          self.dc[L - 1].data_ptr(), self.code, self.dc[L - 1].stride(0),
                                          self.l1_scratch.data_ptr(), ops._stream()), "head_l1")

    def _stage_head(self, s):
        """the head this model gets: fused cross-entropy, fused L1, or stock torch autograd over three ops"""
        if self.fused_head:
            self._stage_head_ce(s)
        elif self.fused_l1:
            self._stage_head_l1(s)
        else:
            self._torch_head(s)

    def _will_fuse_tail(self, example_targets):
        return False

    def _tail_on_mfma(self):
        return False

    def _wg_target(self):
        return 240

    def _tail_gather_rows(self):
        return 0

    def _ahead_rows(self):
        """rows of the next batch's last-hop means that launches of the CURRENT step gather (the seed-level launch's
        and the level-0 projection's gather roles, the side section): the gather launch skips them"""
        return self._tail_rows

    def _queue_compute_body(self, par):
        self._stage_compute(self._qset(par))

    def _torch_head(self, s):
        # head: normalize + fc + loss (stock torch, autograd confined to these few ops)
        m, L = self.model, self.L
        emb = self.hout[L - 1].detach().requires_grad_(True)
        z = self.post(emb) if self.post is not None else emb
        preds = m.fc(torch.nn.functional.normalize(z, dim=1))
        loss = self.loss_fn(preds, self.tg_set[s].squeeze())
        d_emb, d_w, d_b = torch.autograd.grad(loss, [emb, m.fc.weight, m.fc.bias])
        nw = d_w.numel()
        self.head_stage[:nw].copy_(d_w.reshape(-1))
        self.head_stage[nw:].copy_(d_b.reshape(-1))
        self.dc[L - 1].copy_(d_emb)
        if self.preds is None:
            self.preds = torch.empty_like(preds)
        self.preds.copy_(preds.detach())

    def _stage_finalize(self, s):
        """Every partial buffer -> flat gradient bucket, + squared-norm partials, + the step's ticks:
        Adam step, Philox call counter, batch-queue index (nothing else in this launch reads them)."""
        L, lib, stream = self.L, nat.lib(), ops._stream()
        if self._fold_finalize():
            assert getattr(self, "_ticks_issued", False), "no finalisation launch: _wgrad_ticks() must precede K5b"
            self._ticks_issued = False
            return
        nat.check(lib.gsage_finalize_grads(self.rdescs.data_ptr(), self.n_rdesc, self.r_max,
                                           self.flat_g.data_ptr(), self.partial.data_ptr(),
                                           self.step.data_ptr(),
                                           None if self.pipelined else self.counter.data_ptr(), L,
                                           self.batch_idx.data_ptr() if self.queue else None, 1,
                                           stream), "finalize_grads")

    # ---- the step's collectives (seed shards, ONE exchange per step) -----------------
    # Three carriers, one call site each: (1) the library's own RCCL communicator (dist.DataParallel.comm,
    # gsage_comm_*): the collective is a node of the step's command list, on the list's side stream when it
    # overlaps the next batch's gathers; (2) torch.distributed inside a host-call node (gsage_host_call: backends
    # without a C entry point -- gloo in the tests -- keep the same list structure); (3) torch.distributed directly
    # when nothing is being recorded.  None of them may run while a hipGraph is being captured ("graph" mode keeps
    # the exchange between two graphs).
    def _on_stream(self, s):
        import contextlib
        if not s:
            return contextlib.nullcontext()
        return torch.cuda.stream(torch.cuda.ExternalStream(int(s)))

    def _x_all_reduce(self, t):
        """average t (a contiguous fp32 slice of the gradient bucket) over the ranks, in place"""
        if self.comm is not None:
            return self.comm.all_reduce(t, True, ops._stream())

        def run(s=None):
            with self._on_stream(s):
                if self._reduce_op == torch.distributed.ReduceOp.SUM:
                    t.div_(self.world)
                torch.distributed.all_reduce(t, op=self._reduce_op)
        if self._in_list:
            self._host_cbs.append(nat.host_call(run))
        else:
            run()

    def _x_all_gather(self, send, recv):
        """recv[r] = rank r's send (recv: [world * send.numel()] of send's type)"""
        if self.comm is not None:
            return self.comm.all_gather(send, recv, ops._stream())
        outs = list(recv.view(self.world, -1).unbind(0))

        def run(s=None):
            with self._on_stream(s):
                torch.distributed.all_gather(outs, send.reshape(-1))
        if self._in_list:
            self._host_cbs.append(nat.host_call(run))
        else:
            run()

    def _stage_exchange(self):
        """The step's exchange.  Dense parameters: ONE all-reduce (average) of the flat fp32 gradient bucket.  A
        trainable embedding table with deferred row updates (configs[3]): its dense 418 MB gradient never travels --
        every rank contributes the row ids of its frontier and their fp32 gradient rows (all-gather, grouped with
        the all-reduce of the other parameters), and every rank then reduces the SAME concatenated list in the
        SAME order (_stage_opt_emb: sort + segment sums), so the replicas' tables stay bit-identical and all of
        them apply the identical touched-row set to the deferred-row Adam."""
        if self.ddp is None:
            return
        if getattr(self, "_marks", False) and self._in_list:           # (instrument(): events around the exchange)
            nat.check(nat.lib().gsage_cmdlist_mark(8), "cmdlist_mark")
            try:
                self._exchange_nodes()
            finally:
                nat.check(nat.lib().gsage_cmdlist_mark(9), "cmdlist_mark")
            return
        self._exchange_nodes()

    def _exchange_nodes(self):
        if self.emb and self.lazy_rows:
            B, RA0 = self.B, self.off[self.L + 1]
            ids = self._cur_ids
            if self.comm is not None:
                self.comm.group(True, ops._stream())
            self._x_all_reduce(self.flat_g[self.n_tab:])
            self._x_all_gather(ids[B:RA0], self.xids)
            self._x_all_gather(self.deraw[B:], self.xrows)
            self._x_all_gather(self.seed_grad, self.xseed)
            if self.comm is not None:
                self.comm.group(False, ops._stream())
            return
        self._x_all_reduce(self.flat_g)

    def _all_reduce(self, async_op=False):
        """("graph" mode and the pipelined engine: the exchange as a torch.distributed call between two graphs)"""
        if self.emb and self.lazy_rows:
            assert not async_op
            return self._stage_exchange()
        if self._reduce_op == torch.distributed.ReduceOp.SUM:
            self.flat_g.div_(self.ddp.world)
        return torch.distributed.all_reduce(self.flat_g, op=self._reduce_op, async_op=async_op)

    def _stage_opt(self):
        """clip_grad_norm(5) + Adam over the flat bucket."""
        if self.emb:
            return self._stage_opt_emb()
        d = self._adam_desc()
        if d.reduce_descs:               # (no finalisation launch ran: the update sums the partial buffers itself)
            nat.check(nat.lib().gsage_clip_adam_meet(ctypes.addressof(d), ops._stream()), "clip_adam_meet")
            return
        nat.check(nat.lib().gsage_clip_adam_step(d.p, d.g, d.m, d.v, d.n, d.partial, d.lr, d.step, d.beta1,
                                                 d.beta2, d.eps, d.weight_decay, d.max_norm, d.norm_out,
                                                 d.step_is_current, d.n_partial_ready, d.prep_descs,
                                                 d.n_prep, d.tick1, d.inc1, d.tick2, d.inc2, ops._stream()),
                  "clip_adam_step")

    def _run_sequential(self, s):
        if self.emb:
            self._cur_ids = self.ids_set[s]
        self._stage_sample_gather(s)
        self._stage_compute(s)
        self._stage_exchange()
        self._stage_opt()


    # =================================================================================================
    # Trainable node-embedding prep (reference nn_modules.py:126-155; BASELINE configs[3], utils/pokec.sh).
    # The level-0 rows are prep.fc(embedding[ids]) -- computed at the start of the step from the CURRENT table,
    # their gradient scattered back into the table's (dense) gradient at the end.  A problem WITH features
    # (nn_modules.py:152-153) has rows [features (D0 columns) | prep.fc(embedding) (E columns)]: the feature columns
    # are copied beside the prep's output each step, and only columns [D0, D0 + E) of the level-0 input gradient
    # are formed (the features take none).  Shared by the mean and the
    # attention engines; a subclass calls _init_emb() from _init_levels, adds _emb_wgrad_problem() to its K5b
    # problems of level 0, forms the level-0 input gradient (din0f / din0) and hands over to _prep_backward().
    # =================================================================================================
    def _init_emb(self, copies):
        """copies(parameter, need_transposed) -> (operand copy, transposed copy): the subclass's operand-copy factory
        (it also records the refresh descriptor).  Allocates the prep's work buffers."""
        prep, dev, T, f32 = self.model.prep, self.dev, self.tdt, torch.float32
        RA0, E = self.off[self.L + 1], self.E
        ld0 = self.ldE = -(-E // 8) * 8 if T == torch.bfloat16 else E          # (the prep's own operands: E wide)
        assert self.din[0] == self.D0 + E and tuple(prep.fc.weight.shape) == (E, E)
        self.wprep, self.wprepT = copies(prep.fc.weight, True)   # (operand copies of prep.fc.weight)
        self.table = prep.embedding.weight                 # a view of the flat parameter bucket
        assert self.pidx[id(self.table)] == 0 and self.table.shape[1] == E and self.table.numel() % 4 == 0
        self.seed_rows = torch.full((self.B,), int(prep.n_nodes), dtype=torch.int64, device=dev)
        self.eraw = torch.zeros(RA0, ld0, dtype=T, device=dev)              # embedding rows as gathered (operand type)
        self.din0f = torch.zeros(RA0, E, dtype=f32, device=dev)             # d prep output
        self.din0 = self.din0f if T == f32 else torch.zeros(RA0, ld0, dtype=T, device=dev)
        self.deraw = torch.zeros(RA0, E, dtype=f32, device=dev)             # d embedding rows
        self.prep_bpart = torch.zeros(256, E, dtype=f32, device=dev)             # prep.fc.bias gradient partials
        self.seed_grad = torch.zeros(min(16, self.B), E, dtype=f32, device=dev)   # partial sums of the gradient of
        #                                                                             the spare row the seeds read
        self._cur_ids = self.ids_set[0]
        # the prep as row pipelines (csrc/gsage_prep_rows.hip: gather + prep.fc in one launch; the level-0 input
        # gradient, the bias sums, the product through prep.fc^T and the table's atomics in another);
        # GSAGE_PREP_ROWS=0: the separate launches
        self.rows_ok = bool(os.environ.get("GSAGE_PREP_ROWS", "1") != "0" and self.D0 % 4 == 0 and
                            nat.lib().gsage_prep_rows_ok(self.code, E))

    def _emb_wgrad_problem(self):
        """(dC, A, lda, M, Ntot, K, parameter, row list) of the prep's affine: d out^T x embedding rows"""
        E = self.E
        return (self.din0, self.eraw, self.eraw.stride(0), self.off[self.L + 1], E, E, self.model.prep.fc.weight, None)

    def _emb_reduce_desc(self):
        """finalisation source of prep.fc.bias (column sums of the level-0 input gradient)"""
        E = self.E
        ib = self.pidx[id(self.model.prep.fc.bias)]
        return _ReduceDesc(self.prep_bpart.data_ptr(), E, self.poff[ib], self.prep_bpart.shape[0], 1, E, E)

    def _init_emb_optimizer(self):
        """after _install_reduce: the table's gradient comes from scatter-adds, its squared norm from a pass of its
        own whose partials sit behind the finalisation's in the same array; deferred row updates (gsage_rows_*)."""
        dev, f32 = self.dev, torch.float32
        self.n_tab = int(self.table.numel())
        self.n_tab_partial = 1024 if self.lazy_rows else nat.lib().gsage_adam_partials(self.n_tab)
        n_dense_sq = nat.lib().gsage_adam_partials(max(1, self.flat_p.numel() - self.n_tab))
        self.partial = torch.zeros(max(self.n_partial, n_dense_sq) + self.n_tab_partial, dtype=f32, device=dev)
        if not self.lazy_rows:
            return
        # a step touches its frontier's rows, everything else is replayed -- bit for bit -- when it is next read
        # (sync_rows: GSSupervised.forward, state_dict, another optimizer / engine taking the Parameters)
        n_rows, E = int(self.table.shape[0]), self.E
        i32 = torch.int32
        self.row_last = torch.zeros(n_rows, dtype=i32, device=dev)
        self.row_seen = torch.zeros(n_rows, dtype=i32, device=dev)
        self.row_hist = torch.zeros(2 * self.ROW_HIST, dtype=f32, device=dev)
        d = self.row_desc = nat.RowAdamDesc()
        o, n = 0, self.n_tab
        d.p, d.g, d.m, d.v = (self.flat_p[o:o + n].data_ptr(), self.flat_g[o:o + n].data_ptr(),
                              self.flat_m[o:o + n].data_ptr(), self.flat_v[o:o + n].data_ptr())
        d.last, d.seen, d.hist = self.row_last.data_ptr(), self.row_seen.data_ptr(), self.row_hist.data_ptr()
        d.lr, d.step, d.n_rows, d.E, d.hist_cap = self.lr.data_ptr(), self.step.data_ptr(), n_rows, E, self.ROW_HIST
        d.beta1, d.beta2, d.eps, d.weight_decay, d.max_norm = 0.9, 0.999, 1e-8, self.wd, 5.0
        self._rows_dirty, self._rows_since = False, 0
        self._warm_reset = (self.row_last, self.row_seen)
        # The table's gradient, deterministically (csrc/gsage_rowsum.hip): the frontier's ids -- every rank's, in a
        # data-parallel run -- are sorted, each run of equal ids is summed in list order and stored (no atomics, no
        # zero-fill), and the norm / Adam passes walk the sorted list (gsage_row_adam.sorted_ids).
        # Single GPU: opt-in (GSAGE_SORTED_ROWS=1) -- the sort's nine launches and the segment sum are ~0.1 ms of the
        # Pokec-shaped step where the atomics cost ~0.02; data-parallel: always (replicas
        # that add the same rows in different orders would drift apart).
        self.sorted_rows = self.ddp is not None or os.environ.get("GSAGE_SORTED_ROWS", "0") == "1"
        if self.sorted_rows:
            W, B, RA0, ns = self.world, self.B, self.off[self.L + 1], int(self.seed_grad.shape[0])
            n0, n1 = W * (RA0 - B), W * ns
            self.x_n0, self.x_n1 = n0, n1
            if self.ddp is not None:
                self.xids = torch.zeros(n0, dtype=torch.int64, device=dev)
                self.xrows = torch.zeros(n0, E, dtype=f32, device=dev)
                self.xseed = torch.zeros(n1, E, dtype=f32, device=dev)
            self.key_bits = max(1, int(n_rows - 1).bit_length())
            self.sids = torch.zeros(n0 + n1, dtype=torch.int64, device=dev)
            self.spos = torch.zeros(n0 + n1, dtype=i32, device=dev)
            nb = int(nat.lib().gsage_sort_rows_temp_bytes(n0 + n1, self.key_bits))
            assert nb > 0, "gsage_sort_rows_temp_bytes failed"
            self.sort_temp = torch.zeros(nb, dtype=torch.uint8, device=dev)
            ds = self.row_desc_sorted = nat.RowAdamDesc()
            ctypes.memmove(ctypes.addressof(ds), ctypes.addressof(d), ctypes.sizeof(d))
            ds.sorted_ids = 1
        self.model._settle_rows = self.sync_rows
        emb_mod = self.model.prep.embedding
        self._row_hooks = [emb_mod.register_forward_pre_hook(lambda *_: self.sync_rows()),
                           emb_mod.register_state_dict_pre_hook(lambda *_: self.sync_rows())]

    def _prep_forward(self, s):
        lib, stream, prep = nat.lib(), ops._stream(), self.model.prep
        ids, B, RA0, E = self._cur_ids, self.B, self.off[self.L + 1], self.E
        tab = self.table
        if self.lazy_rows:       # the rows this step reads, brought up to the last update
            nat.check(lib.gsage_rows_catch_up(ctypes.byref(self.row_desc), self.seed_rows.data_ptr(), 1,
                                              ids[B:RA0].data_ptr(), RA0 - B, 0, stream), "rows_catch_up")
        g0 = self.g0_set[s]
        if self.rows_ok:         # table row -> bf16 -> prep.fc + bias, one launch
            if self.D0:
                st = self.store
                ops.gather_mean_multi([(st.data, ids[:RA0], g0, RA0, 1)], st.ld, st.dim, self.ldin[0])
            nat.check(lib.gsage_prep_rows_fwd(tab.data_ptr(), tab.stride(0), ids.data_ptr(), B, int(prep.n_nodes),
                                              self.wprep.data_ptr(), self.wprep.shape[1], prep.fc.bias.data_ptr(), RA0, E,
                                              self.eraw.data_ptr(), self.eraw.stride(0), g0.data_ptr() + self.D0 * self.esz,
                                              self.ldin[0], stream), "prep_rows_fwd")
            return
        # fp32 table rows -> the operand type in the gather itself (seeds read the spare row n_nodes)
        segs = [(tab, self.seed_rows, self.eraw[:B], B, 1), (tab, ids[B:RA0], self.eraw[B:], RA0 - B, 1)]
        ops.gather_mean_multi(segs, E, E, self.eraw.stride(0))
        if self.D0:              # [features | ...]: the frontier's feature rows (whole 16-byte chunks: before the affine)
            st = self.store
            ops.gather_mean_multi([(st.data, ids[:RA0], g0, RA0, 1)], st.ld, st.dim, self.ldin[0])
        ops._linear_launch(self.eraw.data_ptr(), self.eraw.stride(0), None, 0, self.wprep.data_ptr(), self.wprep.shape[1],
                           prep.fc.bias.data_ptr(), g0.data_ptr() + self.D0 * self.esz, self.ldin[0], RA0, E, E,
                           nat.ACT_NONE, 1, 0, 0, 0, self.code, self.code)

    def _prep_backward_rows(self, s, dhid, w0t, datt, ldatt, dx, ldx, r_x, dagg, ldagg, ws):
        """_input_grad0 + _prep_backward of an engine as ONE launch (self.rows_ok): the sources of the level-0 input
        gradient as gsage_attn_merge_bwd2 takes them (pointers already at the prep's columns), or d hid + the transposed
        operand copy of att.0 instead of the gradient through att(.) -> din0, the bias partials, the table's gradient."""
        lib, stream = nat.lib(), ops._stream()
        ids, B, RA0, E, L = self._cur_ids, self.B, self.off[self.L + 1], self.E, self.L
        g = self._grad_slice(self.table)
        sorted_rows = self.lazy_rows and self.sorted_rows       # (the rows then meet in _stage_opt_emb, from deraw)
        nat.check(lib.gsage_prep_rows_bwd(
            dhid, self.HA_LD if dhid else 0, w0t.data_ptr() if w0t is not None else None, w0t.shape[1] if w0t is not None else 0,
            datt, ldatt, dx, ldx, r_x, dagg, ldagg, ws, L + 1, self.off_host, self.fan_host, RA0, E, self.din0.data_ptr(),
            self.din0.stride(0), self.prep_bpart.data_ptr(), self.prep_bpart.shape[0], self.wprepT.data_ptr(),
            self.wprepT.shape[1], ids.data_ptr(), B, int(self.model.prep.n_nodes), g.data_ptr(), E,
            self.deraw.data_ptr() if sorted_rows else None, E, stream), "prep_rows_bwd")
        if sorted_rows:
            ns = self.seed_grad.shape[0]
            nat.check(lib.gsage_colsum_partials(self.deraw.data_ptr(), E, B, E, self.seed_grad.data_ptr(), ns, stream),
                      "colsum_partials")

    def _prep_backward(self, s):
        """level 0's input gradient (din0f fp32, din0 = its operand copy; formed by the subclass) -> prep.fc (weight:
        a K5b problem of level 0; bias: column sums) -> the table's gradient."""
        lib, stream = nat.lib(), ops._stream()
        ids, B, RA0, E = self._cur_ids, self.B, self.off[self.L + 1], self.E
        nat.check(lib.gsage_colsum_partials(self.din0f.data_ptr(), E, RA0, E, self.prep_bpart.data_ptr(),
                                            self.prep_bpart.shape[0], stream), "colsum_partials")
        ops._linear_launch(self.din0.data_ptr(), self.din0.stride(0), None, 0, self.wprepT.data_ptr(), self.wprepT.shape[1],
                           None, self.deraw.data_ptr(), E, RA0, E, E, nat.ACT_NONE, 1, 0, 0, 0, self.code, nat.F32)
        g = self._grad_slice(self.table)
        # every seed reads the SAME spare row: its B gradient rows are summed first (B atomics onto one row took 15 us)
        ns = self.seed_grad.shape[0]             # (partial sums: one workgroup summing B rows alone took 14 us)
        nat.check(lib.gsage_colsum_partials(self.deraw.data_ptr(), E, B, E, self.seed_grad.data_ptr(), ns, stream),
                  "colsum_partials")
        if self.lazy_rows and self.sorted_rows:
            return            # the rows meet in _stage_opt_emb (after the exchange, in a data-parallel run)
        for rows, idv, M in ((self.seed_grad, self.seed_rows, ns), (self.deraw[B:], ids[B:RA0], RA0 - B)):
            nat.check(lib.gsage_scatter_add_rows(rows.data_ptr(), E, idv.data_ptr(), M, 1, E, 1.0, g.data_ptr(), E,
                                                 stream), "scatter_add_rows")

    def _stage_opt_emb(self):
        lib, stream = nat.lib(), ops._stream()
        d = self._adam_desc()
        nt, B, RA0, E = self.n_tab, self.B, self.off[self.L + 1], self.E
        g = self._grad_slice(self.table)
        n_all = self.n_partial + self.n_tab_partial
        if self.lazy_rows:
            ids = self._cur_ids
            base = self.n_partial             # norm partials of the other parameters: the finalisation's ...
            if self.ddp is not None:          # ... or, after an exchange, those of the AVERAGED gradient
                nd = self.flat_p.numel() - nt
                base = lib.gsage_adam_partials(nd)
                nat.check(lib.gsage_grad_sqnorm(self.flat_g[nt:].data_ptr(), nd, self.partial.data_ptr(), base, stream),
                          "grad_sqnorm")
                n_all = base + self.n_tab_partial
            if self.sorted_rows:
                # every rank's (ids, gradient rows) in rank order [+ the seeds' spare row, 16 partial rows per rank]:
                # one stable sort, one segment sum per distinct row (scaled by 1 / world: the all-reduce's average)
                dp = self.ddp is not None
                src = self.xids if dp else ids[B:RA0]
                rows0, rows1 = (self.xrows, self.xseed) if dp else (self.deraw[B:], self.seed_grad)
                n0, n1 = self.x_n0, self.x_n1
                nat.check(lib.gsage_sort_rows(src.data_ptr(), n0, int(self.model.prep.n_nodes), n1, self.key_bits,
                                              self.sids.data_ptr(), self.spos.data_ptr(), self.sort_temp.data_ptr(),
                                              self.sort_temp.numel(), stream), "sort_rows")
                nat.check(lib.gsage_segment_sum_rows(self.sids.data_ptr(), self.spos.data_ptr(), n0 + n1,
                                                     rows0.data_ptr(), E, n0, rows1.data_ptr(), E, E, 1.0 / self.world,
                                                     g.data_ptr(), E, stream), "segment_sum_rows")
                rd = ctypes.byref(self.row_desc_sorted)
                lists = (self.sids.data_ptr(), n0 + n1, None, 0, 0)
            else:
                rd = ctypes.byref(self.row_desc)
                lists = (self.seed_rows.data_ptr(), 1, ids[B:RA0].data_ptr(), RA0 - B, 0)
            nat.check(lib.gsage_rows_sqnorm(rd, *lists, self.partial[base:].data_ptr(), self.n_tab_partial,
                                            stream), "rows_sqnorm")
            nat.check(lib.gsage_rows_adam(rd, *lists, self.partial.data_ptr(), n_all, stream), "rows_adam")
            o, n = nt, self.flat_p.numel() - nt
            nat.check(lib.gsage_clip_adam_step(self.flat_p[o:].data_ptr(), self.flat_g[o:].data_ptr(),
                                               self.flat_m[o:].data_ptr(), self.flat_v[o:].data_ptr(), n,
                                               self.partial.data_ptr(), d.lr, d.step, d.beta1, d.beta2, d.eps,
                                               d.weight_decay, d.max_norm, d.norm_out, 1, n_all, d.prep_descs, d.n_prep,
                                               None, 0, None, 0, stream), "clip_adam_step")
            return
        base = self.n_partial
        if self.ddp is not None:              # (dense table, data-parallel: the whole bucket was averaged)
            nd = self.flat_p.numel() - nt
            base = lib.gsage_adam_partials(nd)
            nat.check(lib.gsage_grad_sqnorm(self.flat_g[nt:].data_ptr(), nd, self.partial.data_ptr(), base, stream),
                      "grad_sqnorm")
            n_all = base + self.n_tab_partial
        nat.check(lib.gsage_grad_sqnorm(g.data_ptr(), nt, self.partial[base:].data_ptr(), self.n_tab_partial,
                                        stream), "grad_sqnorm")
        # the table (16-byte lanes, no operand copies), then everything else (operand copies refreshed)
        # (flag 2 on the table: its gradient is zeroed below, no need to write the clipped values back; flag 4: the
        # arithmetic of the deferred row updates, so dense and deferred runs of the table stay bit-comparable)
        for (o, n, prep, n_prep, cur) in ((0, nt, None, 0, 7), (nt, self.flat_p.numel() - nt, d.prep_descs, d.n_prep, 1)):
            nat.check(lib.gsage_clip_adam_step(self.flat_p[o:].data_ptr(), self.flat_g[o:].data_ptr(),
                                               self.flat_m[o:].data_ptr(), self.flat_v[o:].data_ptr(), n,
                                               self.partial.data_ptr(), d.lr, d.step, d.beta1, d.beta2, d.eps,
                                               d.weight_decay, d.max_norm, d.norm_out, cur, n_all, prep, n_prep, None,
                                               0, None, 0, stream), "clip_adam_step")
        # the table's gradient goes back to zero by touching the rows this step wrote
        ids = self._cur_ids
        for idv, M in ((self.seed_rows[:1], 1), (ids[B:RA0], RA0 - B)):
            nat.check(lib.gsage_zero_rows(g.data_ptr(), E, idv.data_ptr(), M, E, stream), "zero_rows")

    def sync_rows(self):
        """Deferred table rows: apply every pending update to every row (table, exp_avg, exp_avg_sq all current
        afterwards).  Runs by itself before GSSupervised.forward, the embedding module's forward and state_dict;
        call it before reading `prep.embedding.weight` or the optimizer buckets directly."""
        if not self.lazy_rows or not self._rows_dirty:
            return
        nat.check(nat.lib().gsage_rows_catch_up_all(ctypes.byref(self.row_desc), 0, ops._stream()), "rows_catch_up_all")
        self._rows_dirty, self._rows_since = False, 0

    def close(self):
        """Settle the deferred rows and detach from the model (hooks on the embedding module, model._settle_rows):
        call before pickling the model or when the engine is done with."""
        if self.lazy_rows:
            self.sync_rows()
            for h in self._row_hooks:
                h.remove()
            self._row_hooks = []
            if getattr(self.model, "_settle_rows", None) == self.sync_rows:
                del self.model._settle_rows
            self._closed = True

    def _rows_tick(self):
        if getattr(self, "_closed", False):
            raise RuntimeError("this engine was closed (deferred table rows settled, hooks removed): build a new one")
        if self.lazy_rows:
            if self._rows_since >= self.ROW_HIST - 2:
                self.sync_rows()
            self._rows_dirty = True
            self._rows_since += 1

    # ---- forward only: train.evaluate's folds on the engine's launches (eval_only=True) ----------------------
    EVAL_BATCHES = 64          # batches per replay of the evaluation queue's buffers (longer folds run in pieces)

    def _current_param_ptrs(self):
        return tuple(int(p.data_ptr()) for p in self.params)

    def _finish_init_eval(self, capture, warmup):
        dev, B, nb = self.dev, self.B, self.EVAL_BATCHES
        self.capture_mode = {True: "cmdlist", False: None, None: None}.get(capture, capture)
        if self.capture_mode == "cmdlist" and not (self.fused_head or self.fused_l1):
            self.capture_mode = None               # (the stock-torch head cannot be recorded: eager launches)
        assert self.capture_mode in (None, "cmdlist"), "eval_only records command lists (or launches eagerly)"
        # the fold's batches as a device-resident queue (ids, draws): one recorded list serves every batch
        self.q_ids = torch.zeros(nb, B, dtype=torch.int64, device=dev)
        tdt = torch.int64 if self.fused_head else torch.float32
        self.q_tg = torch.zeros(nb, B, dtype=tdt, device=dev)             # (the heads want targets: zeros)
        self.queue = (self.q_ids, self.q_tg, nb)
        self.nv_queue = torch.full((nb,), B, dtype=torch.int32, device=dev)
        self.sel_queue = None
        if self.draws != "philox":
            self.sel_queue = torch.zeros(nb, self.n_sel, dtype=torch.int32, device=dev)
        self.tg_set[0].zero_()
        self.q_ids[:] = self.ids_set[0][:B]
        for _ in range(max(1, warmup)):
            self._eval_step()
        torch.cuda.synchronize()
        self.batch_idx.zero_()
        self.counter.zero_()
        self.g_eval = self._record(self._eval_step) if self.capture_mode else None
        torch.cuda.synchronize()

    def _eval_step(self):
        """sample (the queue's next batch, its recorded draws) -> gather -> forward -> tick"""
        lib, st = nat.lib(), ops._stream()
        if self.emb:
            self._cur_ids = self.ids_set[0]
        self._stage_sample(0)
        self._stage_gather(0)
        self._stage_compute(0)
        nat.check(lib.gsage_counter_add(self.batch_idx.data_ptr(), 1, st), "counter_add")
        nat.check(lib.gsage_counter_add(self.counter.data_ptr(), self.L, st), "counter_add")

    def evaluate_fold(self, ids, live):
        """Predictions of a fold cut into batches the way problem.iterate(mode, shuffle=False) cuts it
        (reference train.py:29-36 + problem.py:141-153): ids int64 [n_batches, B] (short chunks padded with their
        first id), live = the chunks' sizes.  The frontier of every batch is the validation sampler's -- drawn from
        the generator the reference's evaluation would draw from, in its order (compat: numpy's legacy stream,
        padded seeds draw nothing) -- and the forward is the engine's (no autograd, no per-op launches).
        -> float32 [sum(live), n_outputs] on the device, the fold's rows in order."""
        assert self.eval_only and ids.dim() == 2 and int(ids.shape[1]) == self.B and len(live) == int(ids.shape[0])
        settle = getattr(self.model, "_settle_rows", None)
        if settle is not None:                      # a training engine's deferred table rows
            settle()
        if self._current_param_ptrs() != self._param_ptrs:
            raise RuntimeError("the model's Parameters moved (another engine or optimizer re-pointed them) after this "
                               "evaluation engine was built: build a new one")
        self.refresh_weights()                      # operand copies of the CURRENT weights (one launch)
        nb, B, nbuf = int(ids.shape[0]), self.B, self.EVAL_BATCHES
        C = int(self.preds.shape[1])
        out = torch.empty(nb, B, C, dtype=torch.float32, device=self.dev)
        for b0 in range(0, nb, nbuf):
            n = min(nbuf, nb - b0)
            self.q_ids[:n].copy_(ids[b0:b0 + n])
            if self.draws != "philox":
                self.sel_queue[:n].copy_(self._draw_epoch(live[b0:b0 + n]))
            self.batch_idx.zero_()
            for b in range(n):
                if self.g_eval is not None:
                    self.g_eval.replay()
                else:
                    self._eval_step()
                out[b0 + b].copy_(self.preds)
        if all(int(v) == B for v in live):
            return out.view(nb * B, C)
        keep = torch.cat([torch.arange(b * B, b * B + int(v)) for b, v in enumerate(live)]).to(self.dev)
        return out.view(nb * B, C)[keep]

    # ---- per-batch entry --------------------------------------------------------------------------
    def set_progress(self, progress):
        self.model.lr = self.model.lr_scheduler(progress)
        lr = float(self.model.lr)
        if lr != getattr(self, "_lr_host", None):      # (a constant schedule costs no launch per batch)
            self._lr_host = lr
            self.lr.fill_(lr)

    def set_sel(self, sels):
        """Replace the Philox draws of the sampler by caller-supplied ones for the following steps
        (None: back to Philox).  sels: one integer array per hop, [M_k, fan_k] or flat, each value in
        [0, max_deg) -- exactly what the reference draws with np.random.choice at nn_modules.py:88, so a
        recorded reference step can be replayed through the engine (parity level 1).  Re-records the
        command lists / graphs (their kernels now read the sel buffer); later set_sel calls with
        same-sized draws only overwrite the buffer."""
        if sels is None:
            assert self.draws == "philox", "only the counter-based sampler can do without caller-supplied draws"
            changed, self.sel, self._sel_user = self.sel is not None, None, False
        else:
            flat = torch.cat([torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x).reshape(-1)
                              .to(device=self.dev, dtype=torch.int32) for x in sels])
            assert flat.numel() == self.n_sel, "sel must cover every draw of the frontier"
            changed = self.sel is None
            self._sel_user = True
            if changed:
                self.sel = flat.clone()
            else:
                self.sel.copy_(flat)
        if changed and self.g_main is not None:
            torch.cuda.synchronize()
            self._record_main()

    def load_epoch(self, ids_epoch, targets_epoch, sel_epoch=None, n_valid=None):
        """Device-resident batch queue: ids_epoch int64 [n_batches, B], targets_epoch int64
        [n_batches, B(,1)] on the GPU.  Afterwards `step_queue()` runs one train_step on the next batch
        of the queue (wrapping around) with no host->device or device->device copies at all.
        Needs the fused classification head and the sequential (non-pipelined) mode; the graph is
        re-captured because its kernels now read the queue.

        n_valid (optional, one int per batch, <= B): the live seeds of each batch -- the reference's near-equal
        `array_split` chunks padded to B (train.py pads with the chunk's first id; padded seeds get no loss and no
        gradient).  sel_epoch: recorded sampler draws (see set_sel).  With a compat-mode or dense sampler and no
        sel_epoch the engine draws the epoch's values itself, from the generator and in the order the reference's
        training loop would consume them."""
        if not self.fused_head:
            raise ValueError("load_epoch needs the fused classification head (classification loss, <= 64 classes, "
                             "int64 targets); this model runs per batch through __call__")
        if self.pipelined:
            raise ValueError("load_epoch is the sequential engine's queue mode; build the engine without pipelined=True")
        if ids_epoch.dim() != 2 or int(ids_epoch.shape[1]) != self.B or ids_epoch.dtype != torch.int64:
            raise ValueError("ids_epoch must be int64 [n_batches, %d], got %s %s"
                             % (self.B, ids_epoch.dtype, tuple(ids_epoch.shape)))
        n_batches = int(ids_epoch.shape[0])
        if targets_epoch.dtype != torch.int64 or targets_epoch.numel() != n_batches * self.B:
            raise ValueError("targets_epoch must be int64 class ids, one per seed: [n_batches, %d(, 1)], got %s %s"
                             % (self.B, targets_epoch.dtype, tuple(targets_epoch.shape)))
        if not (ids_epoch.is_cuda and targets_epoch.is_cuda):
            raise ValueError("the epoch queue lives in HBM: pass CUDA tensors")
        # the previous epoch's last launches still tick the sampler counters and write the frontier buffers zeroed below
        torch.cuda.synchronize()
        tq = targets_epoch.reshape(n_batches, self.B).contiguous()
        self.queue = (ids_epoch.contiguous(), tq, n_batches)
        nv = [self.B] * n_batches if n_valid is None else [int(v) for v in n_valid]
        assert len(nv) == n_batches and all(2 <= v <= self.B for v in nv), "n_valid: one count in [2, B] per batch"
        self.nv_queue = torch.tensor(nv, dtype=torch.int32, device=self.dev)
        self.sel_queue = None
        if sel_epoch is not None:               # [n_batches, draws per frontier] recorded draws (see set_sel)
            self.sel_queue = sel_epoch.to(device=self.dev, dtype=torch.int32).reshape(n_batches, -1).contiguous()
            assert self.sel_queue.shape[1] == self.n_sel
        elif self.draws != "philox":
            self.sel_queue = self._draw_epoch(nv)
        self.batch_idx.zero_()
        # From here on the step is software-pipelined (see step_queue): two frontier buffers, batch
        # i+2 is sampled while batch i+1 is gathered and batch i is updated.
        self.ids_q = [self.ids_set[0], torch.zeros_like(self.ids_set[0])]
        # K1 inside the level-0 projection's launch (mean engine, _k1_in_k5): batch i+2 is sampled at the START of
        # step i, while K5(i) and K5b(i) still read batch i's frontier as their row list -- a ring of THREE
        # WHERE batch i+2 is sampled is decided HERE, once per epoch queue (the predicates read the environment: asked
        # again per step they could disagree with the ring built below and leave a batch unsampled): "k5" / "tail" = a
        # sampler role of that launch (ring of three), "front" = the launch that carries the update (ring of two).
        # The seed-level launch (B / 4 workgroups: half the chip at B = 512) also gathers the first
        # rows of the NEXT batch's last-hop means on the CUs it leaves idle; K5b of the current batch
        # still reads the current operands afterwards, so the level-0 operand buffers alternate too.
        self._k1_where = None
        in_k5, in_tail = self._k1_in_k5(), self._k1_in_tail()
        self._tail_rows = self._tail_gather_rows()
        if in_tail and not self._tail_rows:        # (the sampler role rides behind the gather role: none, no role)
            in_tail = False
        self._k1_where = "k5" if in_k5 else "tail" if in_tail else "front"
        self.P = 3 if self._k1_where != "front" else 2
        if self.P == 3:
            self.ids_q.append(torch.zeros_like(self.ids_set[0]))
        while self._tail_rows and len(self.xa0_set) < self.P:
            self.xa0_set.append(torch.zeros_like(self.xa0_set[0]))
        self._front_ready, self._qstep = False, 0
        self._record_queue()
        return self

    def _record_queue(self):
        self.g_prime, self.g_qfront, self.g_queue = None, None, None
        self._host_cbs = self._cbs_queue = []     # (the previous epoch's lists, and their callbacks, are let go)
        # data-parallel, mean engine: the gathers are cut around the exchange (bulk before the join, rest with Adam)
        self._ddp_split = bool(self.ddp is not None and self.MEAN_ENGINE and not self.emb and
                               self.size[self.L - 1] > self._tail_rows and
                               os.environ.get("GSAGE_DDP_SPLIT", "1") == "1")
        if self.g_main is not None:
            torch.cuda.synchronize()
            self.g_prime = self._record(self._queue_prime)
            if self.ddp is None:
                self.g_queue = [self._record(lambda par=par: self._queue_step(par)) for par in range(self.P)]
            else:
                if self._one_list_ddp():
                    # ONE list per step, the exchange a node of it (on the side stream when it overlaps the gathers)
                    self.g_queue = [self._record(lambda par=par: self._queue_step_ddp(par)) for par in range(2)]
                    self.g_qfront, self.g_opt = None, None
                    return
                # "graph" mode: three pieces so that the exchange can overlap the NEXT batch's gathers (see step_queue)
                self.g_queue = [self._record(lambda par=par: self._queue_compute(par)) for par in range(2)]
                if self.emb:
                    self.g_qfront = [self._record(lambda par=par: self._queue_front(par, True)) for par in range(2)]
                    self.g_opt = None
                elif self._ddp_split:
                    # the bulk of the gathers runs while the exchange is in flight; what follows the exchange
                    # is ONE norm pass and ONE launch: the remaining gathers with Adam(i) and K1(i+2) riding along
                    self.g_qfront = [self._record(lambda par=par: self._queue_front_means(par)) for par in range(2)]
                    self.g_opt = [self._record(lambda par=par: self._queue_front_rest(par)) for par in range(2)]
                else:
                    self.g_qfront = [self._record(lambda par=par: self._queue_front(par, False)) for par in range(2)]
                    self.g_opt = self._record(self._stage_opt)

    # launches an engine can time in place: name -> (start mark, stop mark); subclasses add theirs
    TIMED = {"gather": (0, 1)}

    def instrument(self, on=True):
        """Measurement only (bench.py's roofline objects): re-record the command lists with HIP start / stop events
        attached to the DISPATCH of the launches named in TIMED (gsage_cmdlist_time_next -> hipExtLaunchKernel):
        `gather` = the launch that gathers the next batch's level-0 rows, `seed_level` = the seed-level launch
        that carries the first part of those gathers, and per engine its dominant kernel (pool: K3 on the last
        hop; attention: K4 on the last hop).  `last_launch_ms()` then returns their durations for the step just
        replayed: timed in place, on the stream the step runs on."""
        assert self.capture_mode == "cmdlist"          # (with a process group: `exchange` = the collective nodes)
        self._marks = bool(on)
        torch.cuda.synchronize()
        if self.queue is not None:
            self._record_queue()
        else:
            self._record_main()

    def last_launch_ms(self):
        if self.queue is not None:
            par = (self._qstep - 1) % getattr(self, "P", 2)
            cl = self.g_queue[par].cl
            front = cl
        else:
            cl = front = self.g_main[0].cl
        out = {}
        timed = dict(self.TIMED) if self.ddp is None else {"exchange": (8, 9)}
        for name, (a, b) in timed.items():
            try:
                out[name] = (front if name == "gather" else cl).elapsed_ms(a, b)
            except Exception:                     # this step has no such launch (e.g. nothing is gathered ahead)
                pass
        return out

    def _time_next(self, a, b):
        """While recording an instrumented list: attach start / stop events a, b to the next kernel."""
        if getattr(self, "_marks", False) and self.capture_mode == "cmdlist":
            nat.check(nat.lib().gsage_cmdlist_time_next(a, b), "cmdlist_time_next")

    # the queue pipeline's pieces; par = parity of the step: batch i+1 is gathered from ids_q[1 - par]
    # (into operand set 1 - par when the sets alternate) while batch i+2 is sampled into ids_q[par]
    def _qset(self, par):
        return par if self._tail_rows else 0

    def _k1_in_k5(self):
        return False

    def _k1_in_tail(self):
        return False

    def _nx(self, par):
        """ring position of the NEXT batch (gathered during this step)"""
        return (par + 1) % getattr(self, "P", 2)

    def _nx2(self, par):
        """ring position the batch AFTER the next is sampled into"""
        return (par + 2) % getattr(self, "P", 2)

    def _queue_prime(self):
        self._stage_sample(0, ids=self.ids_q[0])
        self._stage_sample(0, ids=self.ids_q[1], ahead=True)
        if not self.emb:             # (embedding prep: sampling runs ahead, nothing else can -- the rows are weights)
            self._stage_gather(0, ids=self.ids_q[0])

    def _queue_front(self, par, with_adam):
        if self.emb:
            assert with_adam
            self._cur_ids = self.ids_q[par]
            self._stage_opt()                                  # Adam(i) + zeroing of the rows batch i touched
            self._stage_sample(0, ids=self.ids_q[par], ahead=True)      # batch i+2 (batch i's frontier is done with)
            return
        self._time_next(0, 1)
        nx = self._nx(par)
        if self.dense:           # (the gather launch's sampler role walks a CSR: the dense frontier is a launch of its own)
            self._stage_gather(self._qset(nx), with_adam=with_adam, ids=self.ids_q[nx],
                               skip_rows=self._ahead_rows())
            self._stage_sample(0, ids=self.ids_q[par], ahead=True)
            return
        # (P == 3: batch i+2 was sampled by the projection's launch of this step: nothing to sample here)
        self._stage_gather(self._qset(nx), with_adam=with_adam, ids=self.ids_q[nx],
                           hops=self._hops_desc(self.ids_q[self._nx2(par)], True) if self.P == 2 else None,
                           skip_rows=self._ahead_rows())

    def _queue_compute(self, par):
        self._q_ids = self.ids_q[par]
        try:
            return self._queue_compute_body(par)
        finally:
            self._q_ids = None

    def _queue_step(self, par):
        self._queue_compute(par)
        self._queue_front(par, True)              # Adam(i) || gathers(i+1) || sampling(i+2)

    def _queue_step_ddp(self, par):
        """One data-parallel step of the queue pipeline as ONE sequence of launches and collective nodes:
        compute(i) -> exchange(i) [side stream] || bulk of the gathers(i+1) + sampling(i+2) -> join -> norm of the
        averaged gradient -> Adam(i) || rest of the gathers(i+1).  Recorded into one command list (or issued
        eagerly, then without the side stream)."""
        lib = nat.lib()
        self._queue_compute(par)
        if self.emb:
            # the level-0 rows are weights: nothing of batch i+1 can run before Adam(i) -- exchange, then update
            self._cur_ids = self.ids_q[par]
            self._stage_exchange()
            self._queue_front(par, True)
            return
        side = self._in_list and self._ddp_overlap()
        if not side and self._norm_in_launch():
            # nothing to hide the exchange behind (one rank, or GSAGE_DDP_OVERLAP=0): the single-GPU order -- ONE launch
            # for the gathers of batch i+1, Adam(i) with the norm formed in the launch, and the sampling of batch i+2
            self._stage_exchange()
            self._queue_front(par, True)
            return
        if side:
            nat.check(lib.gsage_cmdlist_side_begin(), "cmdlist_side_begin")
        self._stage_exchange()
        if side:
            nat.check(lib.gsage_cmdlist_side_end(), "cmdlist_side_end")
        if self._ddp_split:
            self._queue_front_means(par)
        else:
            self._queue_front(par, False)
        if side:
            nat.check(lib.gsage_cmdlist_join(), "cmdlist_join")
        if self._ddp_split:
            self._queue_front_rest(par)
        else:
            self._stage_opt()

    def step_queue(self):
        """One train_step on the next batch of the loaded epoch queue -> preds (static buffer).

        Software-pipelined, because sampling and the level-0 gathers do not depend on the weights:
        the launch that gathers batch i+1 also carries Adam(i) and the frontier sampling of batch i+2
        (two short latency-bound jobs that are free beside the HBM-bound gather); in data-parallel
        runs the gathers + sampling run while batch i's gradient all-reduce is in flight on RCCL's
        stream, Adam(i) following both.  Every call performs exactly one sampling, one gather, one
        forward/backward, (one exchange) and one optimizer step, and the weights are up to date when
        it returns; the first call after load_epoch() additionally samples batches 0 and 1 and
        gathers batch 0."""
        assert self.queue is not None, "call load_epoch() first"
        self._rows_tick()
        rec = self.g_queue is not None
        if not self._front_ready:
            if rec:
                self.g_prime.replay()
            else:
                self._queue_prime()
            self._front_ready = True
        par = self._qstep % self.P
        self._qstep += 1
        if self.ddp is None:
            if rec:
                self.g_queue[par].replay()
            else:
                self._queue_step(par)
            return self.preds
        if self._one_list_ddp() or not rec:
            if rec:
                self.g_queue[par].replay()
            else:
                self._queue_step_ddp(par)
            return self.preds
        self.g_queue[par].replay()
        # Order matters: the collective is submitted BEFORE the gathers.  Submitted after them (from a
        # side stream that only waits for the gradients, which would hide the ~25 us the collective
        # call costs the host) it did not start until the 8 320-workgroup gather launch had been
        # dispatched completely -- no overlap at all (tools/overlap_check.py).
        # (Order and priority were measured on one rank, 0.1205 ms/step as written: the means submitted BEFORE
        # the collective 0.137; RCCL's stream at high priority 0.46 -- its kernel then preempts the gathers.)
        if self.emb:
            self._cur_ids = self.ids_q[par]
            self._stage_exchange()
            self.g_qfront[par].replay()
            return self.preds
        work = self._all_reduce(async_op=True)
        self.g_qfront[par].replay()                  # batch i+1's gathers overlap the exchange
        work.wait()                                  # stream-level wait, the host does not block
        (self.g_opt[par] if isinstance(self.g_opt, list) else self.g_opt).replay()
        return self.preds

    def _pad_batch(self, ids, targets):
        """A batch one or a few seeds short of B (the reference's near-equal chunks): pad with its first seed."""
        b = int(ids.shape[0])
        if not (2 <= b < self.B) or not (self.fused_head or getattr(self, "fused_l1", False)):
            raise ValueError("this engine was recorded for batches of %d seeds (got %d); shorter batches need one of "
                             "the fused heads" % (self.B, b))
        pad = self.B - b
        ids = torch.cat([ids, ids[:1].expand(pad)])
        targets = torch.cat([targets, targets[:1].expand(pad, *targets.shape[1:])])
        return ids.contiguous(), targets.contiguous()

    def _nv_ptr(self):
        """device pointer to the live-seed count(s) the head reads: the epoch queue's array or the per-call word"""
        return (self.nv_queue if self.queue else self.n_valid).data_ptr()

    def _load(self, s, ids, targets):
        dst_i, dst_t = self.ids_set[s][:self.B], self.tg_set[s]
        if (ids.is_cuda and targets.is_cuda and ids.dtype == dst_i.dtype and targets.dtype == dst_t.dtype and
                ids.is_contiguous() and targets.is_contiguous() and ids.numel() == dst_i.numel() and
                targets.numel() == dst_t.numel() and targets.element_size() * targets.numel() % 4 == 0):
            # one launch instead of two blits (hipMemcpyAsync: ~7 us each on the step's stream)
            nat.check(nat.lib().gsage_copy_pair(dst_i.data_ptr(), ids.data_ptr(), ids.numel() * 8, dst_t.data_ptr(),
                                                targets.data_ptr(), targets.numel() * targets.element_size(),
                                                ops._stream()), "copy_pair")
            return
        dst_i.copy_(ids, non_blocking=True)
        dst_t.copy_(targets, non_blocking=True)

    def __call__(self, ids, targets):
        """Sequential mode: same contract as GSSupervised.train_step -> preds of THIS batch.
        Pipelined mode: submits this batch (its sampling/gathers start now), finishes the previous
        one and returns ITS preds (None on the very first call); call flush() after the last batch."""
        self._rows_tick()
        k = self.n_calls
        self.n_calls += 1
        b = int(ids.shape[0])
        if b != self.B:
            ids, targets = self._pad_batch(ids, targets)
        if b != self._nv_host:
            self._nv_host = b
            self.n_valid.fill_(b)
        if self.draws != "philox" and not self._sel_user:
            self._draw_batch(b)
        if not self.pipelined:
            self._load(0, ids, targets)
            if self.g_main is None:
                self._run_sequential(0)
            else:
                self.g_main[0].replay()
                if self.g_opt is not None:
                    self._all_reduce()
                    self.g_opt.replay()
            return self.preds
        s = k % 2
        cur = torch.cuda.current_stream()
        # front stream: wait until the compute stage that last used buffer set s is done, load the
        # batch, sample + gather
        self.s_front.wait_stream(cur)
        with torch.cuda.stream(self.s_front):
            self.s_front.wait_event(self.ev_back[s])
            self._load(s, ids, targets)
            if self.g_front is None:
                self._stage_sample_gather(s)
            else:
                self.g_front[s].replay()
            self.ev_front[s].record(self.s_front)
        if k == 0:
            return None                              # pipeline primed
        self._launch_back((k - 1) % 2)
        cur.wait_event(self.ev_back[(k - 1) % 2])    # whoever reads preds on this stream sees them
        return self.preds

    def _launch_back(self, par):
        with torch.cuda.stream(self.s_back):
            self.s_back.wait_event(self.ev_front[par])
            if self.g_main is None:
                self._stage_compute(par)
                if self.ddp is not None:
                    self._all_reduce()
                self._stage_opt()
            else:
                self.g_main[par].replay()
                if self.g_opt is not None:
                    self._all_reduce()
                    self.g_opt.replay()
            self.ev_back[par].record(self.s_back)

    def flush(self):
        """Pipelined mode: finish the last submitted batch; returns its preds."""
        if not self.pipelined or self.n_calls == 0:
            return self.preds
        self._launch_back((self.n_calls - 1) % 2)
        torch.cuda.current_stream().wait_stream(self.s_back)
        torch.cuda.current_stream().wait_stream(self.s_front)
        self.n_calls = 0                              # the next call primes a fresh pipeline
        return self.preds
