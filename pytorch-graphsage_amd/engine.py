"""
engine.py -- the whole train_step as replayed HIP graphs.

At the reference's batch size (512 seeds) one step moves only ~170 MB and ~10 GFLOP: on an
MI355X that is tens of microseconds of work spread over dozens of kernels, so eager launching
(Python + ~3-5 us per launch) dominates.  `CapturedTrainStep` captures
    zero_grad -> K1 x hops -> K2/K5 forward -> loss -> backward -> [flatten grads]
    [-> RCCL all-reduce (eager, between the two graphs) ->]
    [unflatten] -> clip_grad_norm(5) -> Adam -> advance the Philox call counter
once (hipGraph via torch.cuda.CUDAGraph; the ctypes-launched kernels are recorded because they
are enqueued on torch's current stream) and replays it per batch.  Same arithmetic as
GSSupervised.train_step (reference models.py:97-104); inputs are copied into static buffers.

Requirements: model on CUDA, sparse sampler in rng="philox" mode (the compat stream is host
numpy and cannot live in a graph); the sampler's call index is read from a device counter that
the graph itself advances, so every replay draws fresh samples.
"""
import torch

from . import _native as nat


class CapturedTrainStep(object):
    def __init__(self, model, feats, loss_fn, example_ids, example_targets, ddp=None, warmup=3):
        assert example_ids.is_cuda, "CapturedTrainStep needs CUDA tensors"
        self.model, self.feats, self.loss_fn, self.ddp = model, feats, loss_fn, ddp
        self.ids = example_ids.clone()
        self.targets = example_targets.clone()
        dev = self.ids.device

        # clip + Adam over flat buckets (optim.FlatAdam: step count and learning rate live on the
        # device, two launches, capturable; the gradient bucket is also what data-parallel runs exchange)
        from .optim import FlatAdam
        old = model.optimizer
        if not isinstance(old, FlatAdam):
            wd = old.param_groups[0].get("weight_decay", 0.0)
            model.optimizer = FlatAdam([p for p in model.parameters() if p.requires_grad],
                                       lr=float(model.lr), weight_decay=wd)
        self.opt = model.optimizer
        self.lr = self.opt.lr_t
        self.params = self.opt.params

        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        self.samplers = [s for s in (model.train_sampler,) if hasattr(s, "begin_capture")]
        for s in self.samplers:
            assert s.rng == "philox", "captured steps need the counter-based sampler (rng='philox')"
            s.begin_capture(self.counter)

        self.flat = self.opt.flat_g if ddp is not None else None

        # warm-up on a side stream (allocator, lazy Adam state, library handles), then put weights,
        # optimizer state and the sample counter back so the captured run starts from step 0
        saved = [p.detach().clone() for p in self.params]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._front()
                if ddp is not None:
                    torch.distributed.all_reduce(self.flat)
                    self._back()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            for p, q in zip(self.params, saved):
                p.copy_(q)
            for t in (self.opt.flat_m, self.opt.flat_v, self.opt.step_count):
                t.zero_()
            self.counter.zero_()
        torch.cuda.synchronize()

        self.g_front = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_front):
            self.preds = self._front()
        self.g_back = None
        if ddp is None:
            pass                                   # _front already ran clip + Adam (single graph)
        else:
            self.g_back = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_back, pool=self.g_front.pool()):
                self._back()

    # ---- graph bodies ------------------------------------------------------------------------
    def _front(self):
        m = self.model
        for s in self.samplers:
            s._static_calls = 0
        self.opt.zero_grad()
        preds = m(self.ids, self.feats, train=True)
        loss = self.loss_fn(preds, self.targets.squeeze())
        loss.backward()
        if self.ddp is None:
            self._finish()
        else:
            self.flat.div_(self.ddp.world)         # the gradients already are one flat bucket
        return preds

    def _back(self):
        self._finish()

    def _finish(self):
        self.opt.clip_and_step(5.0)
        calls = sum(s.calls_in_capture() for s in self.samplers)
        if calls:
            nat.check(nat.lib().gsage_counter_add(self.counter.data_ptr(), calls,
                                                  torch.cuda.current_stream().cuda_stream),
                      "counter_add")

    # ---- per-batch entry ----------------------------------------------------------------------
    def set_progress(self, progress):
        self.model.lr = self.model.lr_scheduler(progress)
        self.opt.param_groups[0]["lr"] = float(self.model.lr)
        self.opt._lr_seen = float(self.model.lr)
        self.lr.fill_(float(self.model.lr))

    def __call__(self, ids, targets):
        """Same contract as GSSupervised.train_step(ids, feats, targets, loss_fn) -> preds
        (the returned tensor is a static buffer, overwritten by the next call)."""
        self.ids.copy_(ids, non_blocking=True)
        self.targets.copy_(targets, non_blocking=True)
        self.g_front.replay()
        if self.g_back is not None:
            torch.distributed.all_reduce(self.flat)
            self.g_back.replay()
        return self.preds


# =================================================================================================
# Fused engine: the SAGE layers without autograd
# =================================================================================================
import ctypes
import os

import numpy as np

from . import ops
from .nn_modules import IdentityPrep, MaxPoolAggregator, MeanAggregator, MeanPoolAggregator, \
    SparseUniformNeighborSampler, \
    _split_activation, concat_combine
from .store import FeatureStore


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


class FusedMeanTrainStep(object):
    """train_step (reference models.py:97-104) for the north-star configuration -- sparse sampler,
    identity prep over a bf16 FeatureStore, mean aggregators (ReLU on all but the last layer) --
    without autograd below the loss and without framework glue kernels.  For the 2-layer Reddit
    shape a step is FIVE launches, recorded once into a native command list (or a hipGraph) and
    replayed per batch:

        K5            level 0: ONE grouped MFMA GEMM (x | agg against Wx | Wn), bf16 out
        seed level    segment mean + both projections + normalize/fc/CE + all gradients down to the
                      level-0 activations in one kernel (gsage_mean_tail_ce; generic models use K2 + K5
                      + gsage_head_ce + K5/merge per level instead); in queue mode the CUs its B / 4
                      workgroups leave idle gather the first part of the NEXT batch's last-hop means
        K5b           every level's weight gradient in one grouped launch (partial tiles -> slabs)
        finalise      partial tiles + head partials -> flat gradient bucket + norm partials; ticks the
                      step's device counters
        [RCCL]        one all-reduce of the flat gradient bucket (data-parallel runs only)
        Adam          clip + Adam + refresh of the bf16 operand copies, side by side with the level-0
                      gathers (x rows | neighbour means of every hop) of the NEXT batch and with K1
                      (all hops) for the batch after that (queue mode; otherwise K1 and the gathers
                      open the step as launches of their own)

    The arithmetic is that of GSSupervised.train_step.  Parameters and gradients live in flat fp32
    buckets; the model's Parameters become views of them, so `model.state_dict()`, evaluation and
    checkpointing keep working.  `__call__(ids, targets)` has the contract of train_step;
    `load_epoch()` + `step_queue()` walk a device-resident queue of seed batches with no host copies.
    """

    @staticmethod
    def supports(model, feats):
        layers = list(model.agg_layers.children())
        if not layers or not all(type(l) is MeanAggregator and l.combine_fn is concat_combine for l in layers):
            return False
        codes = [_split_activation(l.activation)[0] for l in layers]
        if codes[:-1] != [nat.ACT_RELU] * (len(layers) - 1) or codes[-1] != nat.ACT_NONE:
            return False
        if not isinstance(model.prep, IdentityPrep) or not isinstance(feats, FeatureStore):
            return False
        # bf16 storage = the production path; fp32 storage = the exact-arithmetic parity mode (same
        # engine, same kernel sources instantiated on fp32: golden fixtures replay at 2e-4)
        if feats.dtype not in (torch.bfloat16, torch.float32) or not feats.is_cuda:
            return False
        if not isinstance(model.train_sampler, SparseUniformNeighborSampler) or model.train_sampler.rng != "philox":
            return False
        return all(l.output_dim_ % 8 == 0 for l in layers)

    def __init__(self, model, feats, loss_fn, example_ids, example_targets, ddp=None, capture=True,
                 warmup=2, pipelined=False, gather_cus=None):
        """gather_cus (queue mode, single GPU, command lists): run the weight-independent half of the step
        -- sampling of batch i+2 and the level-0 gathers of batch i+1 -- on a stream restricted to that many
        compute units while the forward / backward / update chain of batch i runs on a stream restricted
        to the others (see _split_* below).  None: GSAGE_GATHER_CUS from the environment, else off."""
        if not type(self).supports(model, feats):
            raise ValueError("%s does not cover this (model, feature store): see %s.supports"
                             % (type(self).__name__, type(self).__name__))
        if not (torch.is_tensor(example_ids) and example_ids.is_cuda and example_ids.dtype == torch.int64
                and example_ids.dim() == 1):
            raise ValueError("example_ids must be a CUDA int64 vector of seed ids (one batch)")
        if not (torch.is_tensor(example_targets) and example_targets.is_cuda
                and int(example_targets.shape[0]) == int(example_ids.shape[0])):
            raise ValueError("example_targets must be a CUDA tensor with one row per seed")
        if gather_cus is None:
            gather_cus = int(os.environ.get("GSAGE_GATHER_CUS", "0"))
        self.gather_cus = int(gather_cus) if (ddp is None and not pipelined and type(self) is FusedMeanTrainStep) else 0
        self._init_common(model, feats, loss_fn, example_ids, example_targets, ddp, pipelined)
        self._init_levels(example_ids, example_targets)
        self._init_head(loss_fn, example_targets)
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
        self.fan = [1] + [fn.keywords["n_samples"] for fn in model.train_sample_fns]
        B = self.B = int(example_ids.shape[0])
        self.size = [B]
        for k in range(1, L + 1):
            self.size.append(self.size[-1] * self.fan[k])
        self.off = [0]
        for k in range(L + 1):
            self.off.append(self.off[-1] + self.size[k])          # off[k] = first row of hop k
        self.sampler = model.train_sampler
        self.csr = self.sampler.csr(dev)

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
        self.flat_p = torch.cat([p.detach().reshape(-1).float() for p in self.params]).contiguous()
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros_like(self.flat_g)
        self.flat_v = torch.zeros_like(self.flat_g)
        for p, o, n in zip(self.params, self.poff, sizes):
            p.data = self.flat_p[o:o + n].view_as(p)
            p.grad = self.flat_g[o:o + n].view_as(p)
        self.pidx = {id(p): i for i, p in enumerate(self.params)}
        self.step = torch.zeros(1, dtype=torch.int64, device=dev)
        self.lr = torch.tensor([float(model.lr)], dtype=torch.float32, device=dev)
        self.wd = float(model.optimizer.param_groups[0].get("weight_decay", 0.0))
        self.partial = torch.zeros(nat.lib().gsage_adam_partials(total), dtype=torch.float32, device=dev)
        self.gnorm = torch.zeros(1, dtype=torch.float32, device=dev)

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

    def _init_levels(self, example_ids, example_targets):
        """Mean aggregator: per-level shapes, bf16 operand copies (+ their refresh descriptors) and
        work buffers."""
        model, feats, dev, L, B = self.model, self.store, self.dev, self.L, self.B
        # ---- per-level shapes, operand copies and work buffers ------------------------------
        self.h = [l.output_dim_ for l in self.layers]
        self.din = [feats.dim] + [2 * h for h in self.h[:-1]]
        self.rows = [self.off[L - l] for l in range(L)]           # R_l = rows of level l
        self.w2, self.w2t, self.wp, descs = [], [], [], []
        for l, layer in enumerate(self.layers):
            h, din = self.h[l], self.din[l]
            assert tuple(layer.fc_x.weight.shape) == (h, din) == tuple(layer.fc_neib.weight.shape)
            ix, inb = self.pidx[id(layer.fc_x.weight)], self.pidx[id(layer.fc_neib.weight)]
            assert inb == ix + 1, "fc_x / fc_neib must be adjacent in the parameter order"
            w2 = torch.zeros(2, h, _r8(din), dtype=self.tdt, device=dev)
            w2t = torch.zeros(2, din, _r8(h), dtype=self.tdt, device=dev) if l > 0 else None
            tail_level = self._will_fuse_tail(example_targets) and l == L - 1
            self.w2.append(w2)
            self.w2t.append(w2t)
            # levels whose forward runs on K5 read the weights in MFMA fragment order
            # (gsage_linear_nt_packed; needs whole-line operand rows); the seed-level kernel reads w2
            lda = feats.ld if l == 0 else din
            # the fragment-ordered copy feeds K5's forward; the seed-level kernel reads w2 / w2t instead
            packed = (self.code == nat.BF16 and lda % 64 == 0 and lda >= -(-din // 64) * 64 and not tail_level)
            gstride = nat.lib().gsage_packed_weight_elems(h, din, 1)
            wp = torch.zeros(2 * gstride, dtype=torch.bfloat16, device=dev) if packed else None
            self.wp.append(wp)
            for g, prm in enumerate((layer.fc_x.weight, layer.fc_neib.weight)):
                # (a copy nobody reads is not refreshed: w2 serves the unpacked K5 and the seed-level kernel)
                descs.append(_PrepDesc(prm.data_ptr(), w2[g].data_ptr() if (not packed or tail_level) else None,
                                       w2t[g].data_ptr() if w2t is not None else None,
                                       h, din, w2.shape[2], w2t.shape[2] if w2t is not None else 0,
                                       wp[g * gstride:].data_ptr() if packed else None, 4 * (-(-din // 64)),
                                       int(self.code == nat.F32), 0))
        raw = bytes((_PrepDesc * len(descs))(*descs))
        self.descs = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        self.n_desc = len(descs)
        self.max_elems = max(h * d for h, d in zip(self.h, self.din))

        bf, f32 = self.tdt, torch.float32
        # per-batch inputs of the compute stage, one set per batch in flight:
        #   ids   the concatenated frontier [hop 0 | hop 1 | ... | hop L]
        #   xa0   level-0 operands [x rows | neighbour means], gathered ONCE per step so the forward
        #         GEMM and the weight-gradient kernel both read plain row-major operands
        self.xa0_set = [torch.zeros(2, self.rows[0], feats.ld, dtype=bf, device=dev) for _ in range(self.nset)]
        # K5 and K5b read the x rows of level 0 in place through the frontier's row list (gsage_linear_nt_packed
        # a_rows / gsage_wgrad_desc.a_rows) instead of from xa0[0]: no row copies in the gather launch (28.5 vs
        # 32.9 us in-step, 0.092 vs 0.095 ms/step at config 2; GSAGE_MEAN_INPLACE_X=0 brings the copies back)
        self.inplace_x = (os.environ.get("GSAGE_MEAN_INPLACE_X", "1") == "1" and self.L >= 2 and
                          not getattr(self, "gather_cus", 0))         # (split mode keeps frontier rings of its own;
        #                                                                one level: the copies ARE the "rest" launch
        #                                                                that carries Adam in data-parallel runs)
        self.agg, self.hout, self.dc, self.dg = [], [], [], []
        for l in range(L):
            R = self.rows[l]
            ld = feats.ld if l == 0 else self.din[l]
            assert ld % 8 == 0
            self.agg.append(None if l == 0 else torch.zeros(R, ld, dtype=bf, device=dev))
            last = l == L - 1
            self.hout.append(torch.zeros(R, 2 * self.h[l], dtype=f32 if last else bf, device=dev))
            self.dc.append(torch.zeros(R, 2 * self.h[l], dtype=bf, device=dev))
            self.dg.append(torch.zeros(R, 2 * self.din[l], dtype=f32, device=dev) if l > 0 else None)
        self.off_host = (ctypes.c_int64 * 6)(*([int(v) for v in self.off[:L + 1]] + [0] * (5 - L)))
        self.fan_host = (ctypes.c_int32 * 6)(*([int(v) for v in self.fan[:L + 1]] + [1] * (5 - L)))


    def _will_fuse_head(self, example_targets):
        from .problem import ProblemLosses
        C, D2 = self.model.fc.weight.shape
        probe = torch.randn(3, 4, device=self.dev)
        ident = self.post is None or torch.equal(self.post(probe), probe)
        return bool(self.loss_fn is ProblemLosses.classification and ident and C <= 64 and D2 <= 1024 and
                    example_targets.dtype == torch.int64)

    def _will_fuse_tail(self, example_targets):
        """The seed level (segment mean + GEMM + head + input gradients + mask/route) as ONE kernel?"""
        L = self.L
        hs = [l.output_dim_ for l in self.layers]
        return bool(type(self) is FusedMeanTrainStep and self._will_fuse_head(example_targets) and L >= 2 and
                    2 * hs[L - 2] == 256 and 2 * hs[L - 1] == 256 and self.fan[1] <= 32 and
                    os.environ.get("GSAGE_NO_FUSED_TAIL", "0") != "1")

    def _init_head(self, loss_fn, example_targets):
        """Classification head as one fused kernel pair when it applies (else stock torch autograd)."""
        model, dev, L, B = self.model, self.dev, self.L, self.B
        C, D2 = model.fc.weight.shape
        self.fused_head = self._will_fuse_head(example_targets)
        self.fused_tail = self._will_fuse_tail(example_targets)
        if self.fused_head:
            assert nat.lib().gsage_head_ce_scratch(B, C, D2) == nat.lib().gsage_mean_tail_ce_scratch(B, C) \
                or not self.fused_tail
            self.head_scratch = torch.zeros(nat.lib().gsage_head_ce_scratch(B, C, D2),
                                            dtype=torch.float32, device=dev)
            self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
            self.preds = torch.zeros(B, C, dtype=torch.float32, device=dev)


    def _init_reduce(self):
        """Gradient partial buffers + the descriptor table gsage_finalize_grads sums them with."""
        model, dev, L = self.model, self.dev, self.L
        f32 = torch.float32
        rdesc, self.slabs = [], []
        for l in range(L):
            h, din, R = self.h[l], self.din[l], self.rows[l]
            ix = self.pidx[id(self.layers[l].fc_x.weight)]
            parts = [(2 * h, 0)] if h % 128 == 0 else [(h, 0), (h, 1)]
            bufs = []
            for ntot, g in parts:
                rps, S, ldk = ops.wgrad_plan(R, ntot, din, self._wg_target())
                buf = torch.zeros(S, ntot, ldk, dtype=f32, device=dev)
                bufs.append(buf)
                rdesc.append(_ReduceDesc(buf.data_ptr(), ntot * ldk, self.poff[ix + g], S, ntot, din, ldk))
            self.slabs.append(bufs)
        self._install_reduce(rdesc)

    def _wg_target(self):
        """K5b workgroups to plan for: the chip, or the chain's share of it in split mode."""
        if not self.gather_cus:
            return 240
        n_cu = int(torch.cuda.get_device_properties(self.dev).multi_processor_count)
        return max(32, n_cu - self.gather_cus - 8)

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
        elif getattr(self, "fused_l1", False):           # gsage_head_l1: one partial row [dW | db | loss] per 16 seeds
            width, n_wg = D2c + 2, (self.B + 15) // 16
            self.l1_scratch = torch.zeros(nat.lib().gsage_head_l1_scratch(self.B, D2c), dtype=f32, device=dev)
            rdesc.append(_ReduceDesc(self.l1_scratch.data_ptr(), width, self.poff[ifc], n_wg, 1, D2c + 1, width))
        else:
            self.head_stage = torch.zeros(Cc * D2c + Cc, dtype=f32, device=dev)
            rdesc.append(_ReduceDesc(self.head_stage.data_ptr(), 0, self.poff[ifc], 1, 1, Cc * D2c + Cc,
                                     Cc * D2c + Cc))
        covered = sum(d.rows * d.cols for d in rdesc)
        uncovered = self._uncovered() if hasattr(self, "_uncovered") else 0     # e.g. a scatter-added embedding table
        assert covered + uncovered == self.flat_p.numel(), "every parameter must be covered by a gradient source"
        self.rdescs = torch.frombuffer(bytearray(bytes((_ReduceDesc * len(rdesc))(*rdesc))),
                                       dtype=torch.uint8).to(dev)
        self.n_rdesc = len(rdesc)
        self.r_max = max(d.rows * d.cols for d in rdesc)
        self.n_partial = nat.lib().gsage_finalize_partials(self.n_rdesc, self.r_max)
        self.partial = torch.zeros(max(self.n_partial, self.partial.numel()), dtype=f32, device=dev)
        self.refresh_weights()

    def _finish_init(self, capture, warmup):
        ddp = self.ddp
        # warm-up (library handles, allocator) with state restored afterwards, then capture
        saved = self.flat_p.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._run_sequential(0)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.flat_p.copy_(saved)
        for t in (self.flat_m, self.flat_v, self.step, self.counter) + tuple(getattr(self, "_warm_reset", ())):
            t.zero_()
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
        if self.capture_mode == "cmdlist" and not self.fused_head:
            self.capture_mode = "graph"              # the stock-torch head cannot be recorded
        self._pool = None
        if self.capture_mode:
            self._record_main()
        torch.cuda.synchronize()

    def _record_main(self):
        """(Re-)record the per-call command lists / graphs of __call__."""
        ddp = self.ddp
        if self.capture_mode == "graph":
            # re-recording: let go of the old hipGraphs (and their private pool) before capturing new ones
            self.g_main, self.g_opt, self.g_front, self._pool = None, None, None, None
            torch.cuda.synchronize()
        self.g_main = []
        if self.pipelined:
            self.g_front = [self._record(lambda st_=st_: self._stage_sample_gather(st_), self.s_front)
                            for st_ in range(2)]

        def main(st_):
            if not self.pipelined:
                self._stage_sample_gather(0)
            self._stage_compute(st_)
            if ddp is None:
                self._stage_opt()
        for st_ in range(self.nset):
            self.g_main.append(self._record(lambda st_=st_: main(st_),
                                            self.s_back if self.pipelined else None))
        if ddp is not None:
            self.g_opt = self._record(self._stage_opt, self.s_back if self.pipelined else None)

    # ---- helpers ------------------------------------------------------------------------------
    def _record(self, fn, stream=None):
        """Record the launches of fn() once; returns an object whose replay() re-issues them on the
        current stream (command list) or on the capture stream (hipGraph)."""
        if self.capture_mode == "cmdlist":
            with nat.CommandList.record() as cl:
                fn()
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
        d.rowptr, d.col, d.n_rows = self.csr.rowptr.data_ptr(), self.csr.col.data_ptr(), self.csr.n_rows
        d.ids, d.B, d.n_hops = ids.data_ptr(), self.B, L
        for k in range(5):
            d.fan[k] = int(self.fan[k + 1]) if k < L else 1
        d.max_deg, d.seed = self.csr.max_deg, self.sampler.seed
        d.call_ctr, d.call_base, d.rank = ctr.data_ptr(), (L if ahead else 0), self.sampler.shard[0]
        d.seed_queue = self.queue[0].data_ptr() if self.queue else None
        d.batch_idx = bidx.data_ptr() if self.queue else None
        d.batch_base, d.n_batches = (1 if ahead else 0), (self.queue[2] if self.queue else 0)
        d.err_flag = self.csr.err_flag.data_ptr()
        if self.queue and self.sel_queue is not None:
            d.sel, d.sel_stride = self.sel_queue.data_ptr(), int(self.sel_queue.shape[1])
        elif self.sel is not None:
            d.sel, d.sel_stride = self.sel.data_ptr(), 0
        return d

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
        d.tick1, d.inc1, d.tick2, d.inc2 = None, 0, None, 0     # the finalisation ticks the counters
        return d

    def _stage_gather(self, s, with_adam=False, ids=None, hops=None, skip_rows=0, part=None, adam=None):
        """Level-0 gathers of batch set s (x rows of every hop + each hop's neighbour means), one
        launch; with_adam: the clip + Adam update of the batch just finished rides along; hops: so
        does the sampling of a later batch's frontier (a gsage_hops_desc writing ANOTHER buffer).
        ids: frontier to gather from (default: the set's own).  part: "means" = only the last hop's
        neighbour means (the bulk: what runs while a gradient exchange is in flight), "rest" = the other
        segments (what then shares a launch with Adam and the sampler); None = everything."""
        L, st = self.L, self.store
        if ids is None:
            ids = self.ids_set[s]
        # one launch: x rows of every hop + the mean of each hop's sampled neighbours
        R = self.rows[0]
        xa = self.xa0_set[s]
        # work items are dealt out in segment order: the means first (many dependent loads per item,
        # smallest hop first), the row copies last so that short items fill the launch's tail
        # (tools/kbench.py gmulti: 36.1 us against 41.9 us for copies first at Reddit shapes)
        segs = []
        for k in range(L):
            n, r0 = self.fan[k + 1], (skip_rows if k == L - 1 else 0)   # rows the seed-level launch gathered
            if self.size[k] > r0 and (part is None or (part == "means") == (k == L - 1)):
                segs.append((st.data, ids[self.off[k + 1] + r0 * n:self.off[k + 2]],
                             xa[1][self.off[k] + r0:self.off[k + 1]], self.size[k] - r0, n))
        if part != "means" and not self.inplace_x:
            segs.append((st.data, ids[:R], xa[0], R, 1))
        if not segs:
            return False
        # (D = the real width: the pad columns of the operand buffers were zeroed once and stay zero)
        if adam is None and with_adam:
            adam = self._adam_desc()
        ops.gather_mean_multi(segs, st.ld, st.dim, st.ld, adam=adam, hops=hops)
        return True

    def _stage_compute(self, s):
        """Forward GEMMs, head, backward; everything that needs the current weights."""
        L, B, st, lib = self.L, self.B, self.store, nat.lib()
        stream = ops._stream()
        esz = self.esz
        for l in range(L - 1 if self.fused_tail else L):
            R, h, din = self.rows[l], self.h[l], self.din[l]
            rows = None
            if l == 0 and self.inplace_x:      # the x rows of every hop, read in place through the frontier
                xbuf, agg, lda = st.data, self.xa0_set[s][1], st.ld
                rows = (self._q_ids if self._q_ids is not None else self.ids_set[s]).data_ptr()
            elif l == 0:
                xbuf, agg, lda = self.xa0_set[s][0], self.xa0_set[s][1], st.ld
            else:
                xbuf, agg, lda = self.hout[l - 1], self.agg[l], din
                segs = [(xbuf[self.off[k + 1]:self.off[k + 2]], None, agg[self.off[k]:self.off[k + 1]],
                         self.size[k], self.fan[k + 1]) for k in range(L - l)]
                ops.gather_mean_multi(segs, din, din, din)
            delta = agg.data_ptr() - xbuf.data_ptr()
            assert delta % esz == 0 and agg.stride(0) == lda
            last = l == L - 1
            if self.wp[l] is not None:
                ops._linear_packed_launch(xbuf.data_ptr(), lda, rows, int(rows is not None), self.wp[l].data_ptr(), None,
                                          self.hout[l].data_ptr(), 2 * h, R, h, din,
                                          nat.ACT_NONE if last else nat.ACT_RELU, 2, delta // esz, h,
                                          nat.F32 if last else nat.BF16)
                continue
            self._linear(xbuf.data_ptr(), lda, rows, int(rows is not None), self.w2[l].data_ptr(), self.w2[l].shape[2],
                         self.hout[l].data_ptr(), nat.F32 if last else self.code, 2 * h, R, h, din,
                         nat.ACT_NONE if last else nat.ACT_RELU, delta // esz,
                         h * self.w2[l].shape[2], h)

        m = self.model
        if self.fused_tail:
            C = m.fc.weight.shape[0]
            tg = self.queue[1] if self.queue else self.tg_set[s].view(-1)
            self._time_next(2, 3)
            nat.check(lib.gsage_mean_tail_ce(
                self.hout[L - 2].data_ptr(), B, self.fan[1], self.w2[L - 1].data_ptr(),
                self.w2[L - 1].shape[2], self.w2t[L - 1].data_ptr(), self.w2t[L - 1].shape[2],
                m.fc.weight.data_ptr(), m.fc.bias.data_ptr(), C, tg.data_ptr(),
                self.batch_idx.data_ptr() if self.queue else None, self.queue[2] if self.queue else 0,
                self.agg[L - 1].data_ptr(), self.dc[L - 1].data_ptr(), self.preds.data_ptr(),
                self.dc[L - 2].data_ptr(), self.head_scratch.data_ptr(),
                ctypes.addressof(self._tail_gather) if self._tail_gather is not None else None, self.code,
                stream), "mean_tail_ce")
        elif self.fused_head:
            C, D2 = m.fc.weight.shape
            tg = self.queue[1] if self.queue else self.tg_set[s].view(-1)
            nat.check(lib.gsage_head_ce(self.hout[L - 1].data_ptr(), self.hout[L - 1].stride(0),
                                        m.fc.weight.data_ptr(), m.fc.bias.data_ptr(), tg.data_ptr(),
                                        B, C, D2, self.preds.data_ptr(), self.dc[L - 1].data_ptr(),
                                        self.code, self.dc[L - 1].stride(0), None, None, None,
                                        self.head_scratch.data_ptr(),
                                        self.batch_idx.data_ptr() if self.queue else None,
                                        self.queue[2] if self.queue else 0, stream), "head_ce")
        else:
            self._torch_head(s)
        self._backward_levels(s)

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

    def _backward_levels(self, s):
        L, st, lib = self.L, self.store, nat.lib()
        stream = ops._stream()
        esz = self.esz
        # (1) the chain of input gradients down the levels: dC[l] -> (dX | dAgg) -> mask/route -> dC[l-1]
        for l in range(L - 1, 0, -1):
            if self.fused_tail and l == L - 1:
                continue                                  # k_mean_tail_ce wrote dC[L-2] already
            R, h, din = self.rows[l], self.h[l], self.din[l]
            w2t = self.w2t[l]
            # NT GEMM against the transposed operand copies
            self._linear(self.dc[l].data_ptr(), 2 * h, None, 0, w2t.data_ptr(), w2t.shape[2],
                         self.dg[l].data_ptr(), nat.F32, 2 * din, R, din, h, nat.ACT_NONE, h,
                         din * w2t.shape[2], din)
            below = self.hout[l - 1]
            nat.check(lib.gsage_bwd_merge(below.data_ptr(), self.code, below.stride(0), self.dg[l].data_ptr(),
                                          2 * din, din, self.dc[l - 1].data_ptr(),
                                          self.dc[l - 1].stride(0), self.rows[l - 1], R, din,
                                          L - l + 1, self.off_host, self.fan_host, stream),
                      "bwd_merge")
        # (2) every level's weight gradient in ONE launch: each alone fills a fraction of the chip
        probs = []
        for l in range(L - 1, -1, -1):
            R, h, din = self.rows[l], self.h[l], self.din[l]
            dc = self.dc[l]
            xbuf, lda = (self.xa0_set[s][0], st.ld) if l == 0 else (self.hout[l - 1], din)
            rows = None
            if l == 0 and self.inplace_x:
                xbuf, rows = st.data, (self._q_ids if self._q_ids is not None else self.ids_set[s])
            aggl = self.xa0_set[s][1] if l == 0 else self.agg[l]
            delta = (aggl.data_ptr() - xbuf.data_ptr()) // esz
            if h % 128 == 0:
                probs.append((dc, xbuf, lda, delta, R, 2 * h, din, h, self.slabs[l][0], self._wg_target(), rows))
            else:
                for g in range(2):
                    probs.append((dc[:, g * h:], xbuf if g == 0 else aggl, lda, 0, R, h, din, h,
                                  self.slabs[l][g], self._wg_target(), rows if g == 0 else None))
        for i in range(0, len(probs), 8):
            ops.wgrad_multi(probs[i:i + 8])
        self._stage_finalize(s)

    def _stage_finalize(self, s):
        """Every partial buffer -> flat gradient bucket, + squared-norm partials, + the step's ticks:
        Adam step, Philox call counter, batch-queue index (nothing else in this launch reads them)."""
        L, lib, stream = self.L, nat.lib(), ops._stream()
        nat.check(lib.gsage_finalize_grads(self.rdescs.data_ptr(), self.n_rdesc, self.r_max,
                                           self.flat_g.data_ptr(), self.partial.data_ptr(),
                                           self.step.data_ptr(),
                                           None if self.pipelined else self.counter.data_ptr(), L,
                                           self.batch_idx.data_ptr() if self.queue else None, 1,
                                           stream), "finalize_grads")

    def _all_reduce(self, async_op=False):
        """The step's ONE exchange: average the flat fp32 gradient bucket over the ranks (RCCL)."""
        if self._reduce_op == torch.distributed.ReduceOp.SUM:
            self.flat_g.div_(self.ddp.world)
        return torch.distributed.all_reduce(self.flat_g, op=self._reduce_op, async_op=async_op)

    def _stage_opt(self):
        """clip_grad_norm(5) + Adam over the flat bucket."""
        d = self._adam_desc()
        nat.check(nat.lib().gsage_clip_adam_step(d.p, d.g, d.m, d.v, d.n, d.partial, d.lr, d.step, d.beta1,
                                                 d.beta2, d.eps, d.weight_decay, d.max_norm, d.norm_out,
                                                 d.step_is_current, d.n_partial_ready, d.prep_descs,
                                                 d.n_prep, d.tick1, d.inc1, d.tick2, d.inc2, ops._stream()),
                  "clip_adam_step")

    def _run_sequential(self, s):
        self._stage_sample_gather(s)
        self._stage_compute(s)
        if self.ddp is not None:
            self._all_reduce()
        self._stage_opt()

    # ---- per-batch entry --------------------------------------------------------------------------
    def set_progress(self, progress):
        self.model.lr = self.model.lr_scheduler(progress)
        self.lr.fill_(float(self.model.lr))
        self._user_dirty = True           # split mode: the chain stream must see this write

    def set_sel(self, sels):
        """Replace the Philox draws of the sampler by caller-supplied ones for the following steps
        (None: back to Philox).  sels: one integer array per hop, [M_k, fan_k] or flat, each value in
        [0, max_deg) -- exactly what the reference draws with np.random.choice at nn_modules.py:88, so a
        recorded reference step can be replayed through the engine (parity level 1).  Re-records the
        command lists / graphs (their kernels now read the sel buffer); later set_sel calls with
        same-sized draws only overwrite the buffer."""
        if sels is None:
            changed, self.sel = self.sel is not None, None
        else:
            flat = torch.cat([torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x).reshape(-1)
                              .to(device=self.dev, dtype=torch.int32) for x in sels])
            assert flat.numel() == self.off[self.L + 1] - self.off[1], "sel must cover every sample of the frontier"
            changed = self.sel is None
            if changed:
                self.sel = flat.clone()
            else:
                self.sel.copy_(flat)
        if changed and self.g_main is not None:
            torch.cuda.synchronize()
            self._record_main()

    def load_epoch(self, ids_epoch, targets_epoch, sel_epoch=None):
        """Device-resident batch queue: ids_epoch int64 [n_batches, B], targets_epoch int64
        [n_batches, B(,1)] on the GPU.  Afterwards `step_queue()` runs one train_step on the next batch
        of the queue (wrapping around) with no host->device or device->device copies at all.
        Needs the fused classification head and the sequential (non-pipelined) mode; the graph is
        re-captured because its kernels now read the queue."""
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
        tq = targets_epoch.reshape(n_batches, self.B).contiguous()
        self.queue = (ids_epoch.contiguous(), tq, n_batches)
        self.sel_queue = None
        if sel_epoch is not None:               # [n_batches, samples per frontier] recorded draws (see set_sel)
            self.sel_queue = sel_epoch.to(device=self.dev, dtype=torch.int32).reshape(n_batches, -1).contiguous()
            assert self.sel_queue.shape[1] == self.off[self.L + 1] - self.off[1]
        self.batch_idx.zero_()
        # From here on the step is software-pipelined (see step_queue): two frontier buffers, batch
        # i+2 is sampled while batch i+1 is gathered and batch i is updated.
        self.ids_q = [self.ids_set[0], torch.zeros_like(self.ids_set[0])]
        # The seed-level launch (B / 4 workgroups: half the chip at B = 512) also gathers the first
        # rows of the NEXT batch's last-hop means on the CUs it leaves idle; K5b of the current batch
        # still reads the current operands afterwards, so the level-0 operand buffers alternate too.
        self.split = bool(self.gather_cus and self.capture_mode == "cmdlist" and self.ddp is None)
        self._tail_rows = 0 if self.split else self._tail_gather_rows()
        if (self._tail_rows or self.split) and len(self.xa0_set) == 1:
            self.xa0_set = [self.xa0_set[0], torch.zeros_like(self.xa0_set[0])]
        self._front_ready, self._qstep = False, 0
        if self.split:
            self._split_setup()
        self._record_queue()
        return self

    # ---- split mode: gathers and chain side by side on disjoint halves of the chip ------------------
    # The level-0 gathers of batch i+1 (170 MB of HBM reads, no weights involved) and the chain of batch i
    # (K5 -> seed level -> K5b -> finalise -> Adam: ~65 us of latency-bound launches that move little)
    # want different things from the chip, and on one stream they can only take turns.  Two streams
    # alone do not help (tools/overlap2_check.py, round 1): the chain's kernels own their CUs through
    # registers / LDS and a gather squeezed in beside them runs at a fraction of its bandwidth.  So each
    # gets CUs of its own (hipExtStreamCreateWithCUMask): `gather_cus` for the gathers (they need ~96
    # to finish inside the chain's time), the rest for the chain.  Per step: the gather stream waits
    # for chain(i-1) (whose K5 / K5b read the operand buffers it is about to overwrite), gathers batch
    # i+1 and samples batch i+2; the chain stream waits for the gathers of batch i.  The sampler reads
    # counters of its own, advanced on the gather stream (the chain's finalisation ticks the step's).
    def _split_setup(self):
        n_cu = int(torch.cuda.get_device_properties(self.dev).multi_processor_count)
        assert 16 <= self.gather_cus <= n_cu - 32, "gather_cus out of range"
        if getattr(self, "_sG", None) is None:
            self._sG = nat.masked_stream(range(self.gather_cus))
            self._sC = nat.masked_stream(range(self.gather_cus, n_cu))
            self._evG = [nat.new_event() for _ in range(2)]
            self._evC = [nat.new_event() for _ in range(2)]
            self.g_ctr = torch.zeros(1, dtype=torch.int64, device=self.dev)
            self.g_bidx = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.g_ctr.zero_()
        self.g_bidx.zero_()
        self._user_dirty = True

    def _split_prime(self):
        c = (self.g_ctr, self.g_bidx)
        self._stage_sample(0, ids=self.ids_q[0], counters=c)
        self._stage_sample(0, ids=self.ids_q[1], ahead=True, counters=c)
        self._stage_gather(0, ids=self.ids_q[0])
        self._split_tick(2)

    def _split_tick(self, n):
        lib, st = nat.lib(), ops._stream()
        nat.check(lib.gsage_counter_add(self.g_ctr.data_ptr(), n * self.L, st), "counter_add")
        nat.check(lib.gsage_counter_add(self.g_bidx.data_ptr(), n, st), "counter_add")

    def _split_front(self, par):
        """gather stream, step i (par = i % 2): gathers of batch i+1 || sampling of batch i+2."""
        self._time_next(0, 1)
        self._stage_gather(1 - par, ids=self.ids_q[1 - par],
                           hops=self._hops_desc(self.ids_q[par], False, (self.g_ctr, self.g_bidx)))
        self._split_tick(1)

    def _split_chain(self, par):
        """chain stream, step i: everything that needs the current weights, then the update."""
        self._tail_gather = None
        self._stage_compute(par)
        self._stage_opt()

    def _record_queue(self):
        self.g_prime, self.g_qfront, self.g_queue = None, None, None
        if getattr(self, "split", False):
            torch.cuda.synchronize()
            self.g_prime = self._record(self._split_prime)
            self.g_qfront = [self._record(lambda par=par: self._split_front(par)) for par in range(2)]
            self.g_queue = [self._record(lambda par=par: self._split_chain(par)) for par in range(2)]
            return
        if self.g_main is not None:
            torch.cuda.synchronize()
            self.g_prime = self._record(self._queue_prime)
            if self.ddp is None:
                self.g_queue = [self._record(lambda par=par: self._queue_step(par)) for par in range(2)]
            else:
                # data-parallel: three pieces so that the exchange can overlap the NEXT batch's
                # gathers (see step_queue)
                self.g_queue = [self._record(lambda par=par: self._queue_compute(par)) for par in range(2)]
                self._ddp_split = bool(type(self) is FusedMeanTrainStep and self.size[self.L - 1] > self._tail_rows
                                       and os.environ.get("GSAGE_DDP_SPLIT", "1") == "1")
                if self._ddp_split:
                    # the bulk of the gathers runs while the exchange is in flight; what follows the exchange
                    # is ONE norm pass and ONE launch: the remaining gathers with Adam(i) and K1(i+2) riding along
                    self.g_qfront = [self._record(lambda par=par: self._queue_front_means(par)) for par in range(2)]
                    self.g_opt = [self._record(lambda par=par: self._queue_front_rest(par)) for par in range(2)]
                else:
                    self.g_qfront = [self._record(lambda par=par: self._queue_front(par, False)) for par in range(2)]
                    self.g_opt = self._record(self._stage_opt)

    def instrument(self, on=True):
        """Measurement only (bench.py's roofline object): re-record the queue-mode command lists with
        HIP start / stop events attached to the dispatch of the step's dominant launch -- the one that
        gathers the next batch's level-0 rows (events 0/1) -- and of the seed-level launch that carries
        the first part of those gathers (events 2/3).  `last_launch_ms()` then returns their durations
        for the step just replayed: timed in place, on the stream the step runs on."""
        assert self.queue is not None and self.capture_mode == "cmdlist" and self.ddp is None
        self._marks = bool(on)
        self._record_queue()

    def last_launch_ms(self):
        par = (self._qstep - 1) % 2
        cl = self.g_queue[par].cl
        out = {"gather": (self.g_qfront[par].cl if self.split else cl).elapsed_ms(0, 1)}
        if self.fused_tail:
            out["seed_level"] = cl.elapsed_ms(2, 3)
        return out

    def _time_next(self, a, b):
        """While recording an instrumented list: attach start / stop events a, b to the next kernel."""
        if getattr(self, "_marks", False) and self.capture_mode == "cmdlist":
            nat.check(nat.lib().gsage_cmdlist_time_next(a, b), "cmdlist_time_next")

    def gather_launch_rows(self):
        """(rows the queue-mode gather launch reads, rows the seed-level launch's gather role reads) per
        step: every sampled frontier row is read exactly once, by one of the two."""
        total = self.off[self.L + 1]
        tail = self._tail_rows * self.fan[self.L]
        if getattr(self, "inplace_x", False):          # K5 / K5b read the x rows themselves
            total -= self.rows[0]
        return total - tail, tail

    def _tail_gather_rows(self):
        """Rows of the last hop's neighbour means that the seed-level launch of the previous step
        gathers (0: none)."""
        if not getattr(self, "fused_tail", False) or self.fan[self.L] not in (5, 10, 15) or self.code != nat.BF16:
            return 0
        n_cu = int(torch.cuda.get_device_properties(self.dev).multi_processor_count)      # MI355X: 256
        n_idle = n_cu - (self.B + 3) // 4                 # one seed-level workgroup per CU
        if n_idle < 32:
            return 0
        # what an idle CU moves while the ~27 us launch lasts does not depend on B: ~480 KB, i.e. ~40
        # means of ten 1.2 KB rows (sweep at B = 512, DESIGN.md section 3: 40 % of 12 800 rows on 128 CUs);
        # other row sizes / fan-outs get the same bytes per idle CU
        frac = float(os.environ.get("GSAGE_TAIL_GATHER_FRAC", "0.4"))
        self._tail_wgs = n_idle
        per_cu = 100.0 * frac * (10.0 * 1204.0) / (self.fan[self.L] * max(self.store.dim * self.esz, 256))
        return min(int(self.size[self.L - 1]), max(int(per_cu * n_idle), 0))

    # the queue pipeline's pieces; par = parity of the step: batch i+1 is gathered from ids_q[1 - par]
    # (into operand set 1 - par when the sets alternate) while batch i+2 is sampled into ids_q[par]
    def _qset(self, par):
        return par if self._tail_rows else 0

    def _queue_prime(self):
        self._stage_sample(0, ids=self.ids_q[0])
        self._stage_sample(0, ids=self.ids_q[1], ahead=True)
        self._stage_gather(0, ids=self.ids_q[0])

    def _queue_front(self, par, with_adam):
        self._time_next(0, 1)
        self._stage_gather(self._qset(1 - par), with_adam=with_adam, ids=self.ids_q[1 - par],
                           hops=self._hops_desc(self.ids_q[par], True), skip_rows=self._tail_rows)

    def _queue_front_means(self, par):
        self._stage_gather(self._qset(1 - par), ids=self.ids_q[1 - par], skip_rows=self._tail_rows, part="means")

    def _queue_front_rest(self, par):
        """after the exchange: squared norm of the averaged gradient, then the rest of the gathers || Adam || K1"""
        n = self.flat_g.numel()
        n_sq = nat.lib().gsage_adam_partials(n)
        nat.check(nat.lib().gsage_grad_sqnorm(self.flat_g.data_ptr(), n, self.partial.data_ptr(), n_sq, ops._stream()),
                  "grad_sqnorm")
        d = self._adam_desc()
        d.n_partial_ready = n_sq
        self._stage_gather(self._qset(1 - par), ids=self.ids_q[1 - par], part="rest", adam=d,
                           hops=self._hops_desc(self.ids_q[par], True))

    def _queue_compute(self, par):
        self._q_ids = self.ids_q[par]
        try:
            return self._queue_compute_body(par)
        finally:
            self._q_ids = None

    def _queue_compute_body(self, par):
        if self._tail_rows:
            L, st, nxt = self.L, self.store, self.ids_q[1 - par]
            d = nat.TailGatherDesc()
            d.table, d.ids = st.data.data_ptr(), nxt[self.off[L]:].data_ptr()
            d.out = self.xa0_set[1 - par][1][self.off[L - 1]:].data_ptr()
            d.ld, d.out_ld, d.D, d.rows = st.ld, st.ld, st.dim, self._tail_rows
            d.n, d.n_workgroups = self.fan[L], self._tail_wgs
            self._tail_gather = d
        try:
            self._stage_compute(self._qset(par))
        finally:
            self._tail_gather = None

    def _queue_step(self, par):
        self._queue_compute(par)
        self._queue_front(par, True)              # Adam(i) || gathers(i+1) || sampling(i+2)

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
        if getattr(self, "split", False):
            return self._step_queue_split()
        rec = self.g_queue is not None
        if not self._front_ready:
            if rec:
                self.g_prime.replay()
            else:
                self._queue_prime()
            self._front_ready = True
        par = self._qstep % 2
        self._qstep += 1
        if self.ddp is None:
            if rec:
                self.g_queue[par].replay()
            else:
                self._queue_step(par)
            return self.preds
        if rec:
            self.g_queue[par].replay()
        else:
            self._queue_compute(par)
        # Order matters: the collective is submitted BEFORE the gathers.  Submitted after them (from a
        # side stream that only waits for the gradients, which would hide the ~25 us the collective
        # call costs the host) it did not start until the 8 320-workgroup gather launch had been
        # dispatched completely -- no overlap at all (tools/overlap_check.py).
        # (Order and priority were measured on one rank, 0.1205 ms/step as written: the means submitted BEFORE
        # the collective 0.137; RCCL's stream at high priority 0.46 -- its kernel then preempts the gathers.)
        split = getattr(self, "_ddp_split", False)
        work = self._all_reduce(async_op=True)
        if rec:                                      # batch i+1's gathers overlap the exchange
            self.g_qfront[par].replay()
        elif split:
            self._queue_front_means(par)
        else:
            self._queue_front(par, False)
        work.wait()                                  # stream-level wait, the host does not block
        if rec:
            (self.g_opt[par] if isinstance(self.g_opt, list) else self.g_opt).replay()
        elif getattr(self, "_ddp_split", False):
            self._queue_front_rest(par)
        else:
            self._stage_opt()
        return self.preds

    def _step_queue_split(self):
        user = torch.cuda.current_stream()
        if not self._front_ready:
            self.g_prime.replay()                       # on the caller's stream, once per epoch
            torch.cuda.synchronize()
            self._front_ready = True
        i = self._qstep
        par = i % 2
        self._qstep += 1
        if self._user_dirty:                            # e.g. set_progress wrote the learning rate
            ev = torch.cuda.Event()
            ev.record(user)
            for h in (self._sC, self._sG):
                torch.cuda.ExternalStream(h).wait_event(ev)
            self._user_dirty = False
        vp = ctypes.c_void_p
        nat.check(nat.lib().gsage_cmdlist_replay_pair(
            self.g_qfront[par].cl._h, vp(self._sG), vp(self._evC[1 - par]) if i > 0 else None, vp(self._evG[par]),
            self.g_queue[par].cl._h, vp(self._sC), vp(self._evG[1 - par]) if i > 0 else None, vp(self._evC[par]),
            vp(user.cuda_stream), 1), "cmdlist_replay_pair")
        return self.preds

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
        k = self.n_calls
        self.n_calls += 1
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


def _r64(v):
    return (int(v) + 63) // 64 * 64


class FusedPoolTrainStep(FusedMeanTrainStep):
    """train_step for max-pool / mean-pool aggregators (reference nn_modules.py:207-256; BASELINE config 3) with no
    autograd and no framework glue, on the same machinery as FusedMeanTrainStep (flat buckets, fused
    multi-hop sampler, head kernel, finalisation + Adam, command lists, batch queue, data-parallel
    order).  Per level l, rows = hops 0 .. L-l-1 ("x"), neighbour rows = hops 1 .. L-l:

      forward    K3   pooled = max_j relu(Wm nb_j + bm) per hop (hidden [M*n, Hm] never leaves the chip),
                      fp32 + bf16 operand copy + argmax
                 K5   out[:, :h] = act(x Wx^T);  K5  out[:, h:] = act(pooled Wn^T)
      backward   K5   d pooled = dC[:, h:] Wn            (NT GEMM against the transposed operand copy)
                 route d pooled through the max / ReLU -> bf16 d hidden [M*n, Hm]   (gsage_pool_route_bwd;
                       mean pool: g / n where K3's sign mask is set, gsage_pool_route_mean_bwd)
                 bias partials of the MLP                                           (gsage_pool_bias_partials)
                 l > 0: K5 dX = dC[:, :h] Wx, K5 dN = d hidden Wm, merge + ReLU mask -> dC of level l-1
                 K5b  all weight gradients of all levels in one grouped launch (fc_x, fc_neib, mlp)
    Level 0 reads its operands from two row buffers gathered once per step (x rows, neighbour rows):
    K3, K5 and K5b all want plain row-major operands, and the gathers run one batch ahead beside Adam.
    """

    NPART = 256          # partial rows per hop for the MLP bias gradient (summed by the finalisation)
    # K5b workgroups per problem: the MLP's weight gradient is 20x the work of the two projections that
    # share its launch, so those take few, long M-slices (fewer partial tiles to write and to sum)
    WG_TARGET = {"m": 240, "x": 40, "n": 40}

    @staticmethod
    def supports(model, feats):
        layers = list(model.agg_layers.children())
        kinds = {type(l) for l in layers}
        if not layers or kinds not in ({MaxPoolAggregator}, {MeanPoolAggregator}):
            return False
        if not all(l.combine_fn is concat_combine for l in layers):
            return False
        codes = [_split_activation(l.activation)[0] for l in layers]
        if codes[:-1] != [nat.ACT_RELU] * (len(layers) - 1) or codes[-1] != nat.ACT_NONE:
            return False
        if not isinstance(model.prep, IdentityPrep) or not isinstance(feats, FeatureStore):
            return False
        if feats.dtype not in (torch.bfloat16, torch.float32) or not feats.is_cuda:
            return False
        if feats.ld % (64 if feats.dtype == torch.bfloat16 else 4) != 0:      # whole lines for the LDS-DMA kernels
            return False
        if not isinstance(model.train_sampler, SparseUniformNeighborSampler) or model.train_sampler.rng != "philox":
            return False
        if any(fn.keywords["n_samples"] > 64 for fn in model.train_sample_fns) or len(layers) > 2:
            return False                                     # K3 tiles hold whole segments up to 64 rows
        return all(l.output_dim_ % 64 == 0 and l.mlp[0].weight.shape[0] % 128 == 0 for l in layers)

    # ---- construction ------------------------------------------------------------------------------
    def _init_levels(self, example_ids, example_targets):
        feats, dev, L = self.store, self.dev, self.L
        bf, f32, i32 = self.tdt, torch.float32, torch.int32
        is_bf = self.code == nat.BF16
        self.pool_mode = nat.POOL_MAX if type(self.layers[0]) is MaxPoolAggregator else nat.POOL_MEAN
        self.h = [l.output_dim_ for l in self.layers]
        self.Hm = [int(l.mlp[0].weight.shape[0]) for l in self.layers]
        self.din = [feats.dim] + [2 * h for h in self.h[:-1]]
        self.rows = [self.off[L - l] for l in range(L)]                     # x rows of level l
        self.nrows = [self.off[L - l + 1] - self.off[1] for l in range(L)]  # neighbour rows of level l
        assert all(d % 64 == 0 for d in self.din[1:]), "hidden widths must be multiples of 32"
        descs = []

        def copies(prm, need_t, packed=False):
            r, c = prm.shape
            w = torch.zeros(r, _r64(c), dtype=bf, device=dev)
            wt = torch.zeros(c, _r64(r), dtype=bf, device=dev) if need_t else None
            # forward operands of K3 / K5 also in MFMA fragment order (gsage_*_packed)
            wp = (torch.zeros(nat.lib().gsage_packed_weight_elems(r, c, 1), dtype=bf, device=dev)
                  if packed and is_bf else None)
            descs.append(_PrepDesc(prm.data_ptr(), w.data_ptr(), wt.data_ptr() if need_t else None, r, c,
                                   w.shape[1], wt.shape[1] if need_t else 0,
                                   wp.data_ptr() if wp is not None else None, 4 * (-(-c // 64)),
                                   int(not is_bf), 0))
            return (w, wt, wp) if packed else (w, wt)
        self.wm, self.wx, self.wn, self.wmT, self.wxT, self.wnT = [], [], [], [], [], []
        self.wm_p, self.wx_p, self.wn_p = [], [], []
        for l, layer in enumerate(self.layers):
            order = [self.pidx[id(p)] for p in (layer.mlp[0].weight, layer.mlp[0].bias, layer.fc_x.weight,
                                                layer.fc_neib.weight)]
            assert order == list(range(order[0], order[0] + 4)), "unexpected parameter order"
            wm, wmT, wm_p = copies(layer.mlp[0].weight, l > 0, True)
            wx, wxT, wx_p = copies(layer.fc_x.weight, l > 0, True)
            wn, wnT, wn_p = copies(layer.fc_neib.weight, True, True)
            self.wm.append(wm); self.wmT.append(wmT); self.wx.append(wx); self.wxT.append(wxT)
            self.wn.append(wn); self.wnT.append(wnT)
            self.wm_p.append(wm_p); self.wx_p.append(wx_p); self.wn_p.append(wn_p)
        self.descs = torch.frombuffer(bytearray(bytes((_PrepDesc * len(descs))(*descs))), dtype=torch.uint8).to(dev)
        self.n_desc = len(descs)
        self.max_elems = max(d.rows * d.cols for d in descs)

        # level-0 operands: x rows (hops 0..L-1) gathered once per step; the neighbour rows (hops 1..L: 141 k rows,
        # 180 MB at Reddit's shape) are read IN PLACE through the frontier's row list by K3 and by K5b
        # (gsage_wgrad_desc.a_rows; its list starts at entry B of the frontier and must be 16-byte aligned: even B) --
        # GSAGE_POOL_COPY_ROWS=1 brings back the gathered copy
        self.inplace0 = os.environ.get("GSAGE_POOL_COPY_ROWS", "0") != "1" and self.B % 2 == 0
        self.x0_set = [torch.zeros(self.rows[0], feats.ld, dtype=bf, device=dev) for _ in range(self.nset)]
        self.xn0_set = [None if self.inplace0 else torch.zeros(self.nrows[0], feats.ld, dtype=bf, device=dev)
                        for _ in range(self.nset)]
        self._q_ids = None
        self.pooled, self.pooled_b, self.argmax, self.hout, self.dc = [], [], [], [], []
        self.dpool, self.ghc, self.dxb, self.dnb, self.bpart = [], [], [], [], []
        for l in range(L):
            R, NR, Hm, h, din = self.rows[l], self.nrows[l], self.Hm[l], self.h[l], self.din[l]
            last = l == L - 1
            self.pooled.append(torch.zeros(R, Hm, dtype=f32, device=dev))
            # operand copy of `pooled` for the fc_neib projection and its weight gradient (parity mode:
            # the fp32 result itself)
            self.pooled_b.append(torch.zeros(R, _r64(Hm), dtype=bf, device=dev) if is_bf else self.pooled[l])
            # what the backward needs of the hidden layer: the winning row (max) / the ReLU sign bits (mean)
            self.argmax.append(torch.zeros(R, Hm, dtype=i32, device=dev) if self.pool_mode == nat.POOL_MAX
                               else torch.zeros(NR, Hm // 32, dtype=i32, device=dev))
            self.hout.append(torch.zeros(R, 2 * h, dtype=f32 if last else bf, device=dev))
            self.dc.append(torch.zeros(R, 2 * h, dtype=bf, device=dev))
            self.dpool.append(torch.zeros(R, Hm, dtype=f32, device=dev))
            self.ghc.append(torch.zeros(NR, Hm, dtype=bf, device=dev))
            self.dxb.append(torch.zeros(R, din, dtype=f32, device=dev) if l > 0 else None)
            self.dnb.append(torch.zeros(NR, din, dtype=f32, device=dev) if l > 0 else None)
            self.bpart.append(torch.zeros((L - l) * self.NPART, Hm, dtype=f32, device=dev))

    def _init_head(self, loss_fn, example_targets):
        super(FusedPoolTrainStep, self)._init_head(loss_fn, example_targets)
        self.fused_tail = False
        assert self.fused_head, "FusedPoolTrainStep needs the fused classification head"

    def _x_operand(self, l, s):
        return (self.x0_set[s], self.store.ld) if l == 0 else (self.hout[l - 1], self.din[l])

    def _nb_operand(self, l, s):
        """neighbour rows of level l: (row block, leading dimension, row list or None).  With a row list, neighbour
        row i is block[list[i]] (level 0 read in place from the feature table)."""
        if l == 0:
            if self.inplace0:       # the frontier of the batch being computed: the queue's, else the set's own
                ids = self._q_ids if self._q_ids is not None else self.ids_set[s]
                return self.store.data, self.store.ld, ids[self.off[1]:]
            return self.xn0_set[s], self.store.ld, None
        return self.hout[l - 1][self.off[1]:], self.din[l], None

    def _init_reduce(self):
        dev, L, f32 = self.dev, self.L, torch.float32
        rdesc, self.slabs = [], []
        for l, layer in enumerate(self.layers):
            R, NR, Hm, h, din = self.rows[l], self.nrows[l], self.Hm[l], self.h[l], self.din[l]
            bufs = {}
            for key, prm, M, ntot, K in (("m", layer.mlp[0].weight, NR, Hm, din), ("x", layer.fc_x.weight, R, h, din),
                                         ("n", layer.fc_neib.weight, R, h, Hm)):
                rps, S, ldk = ops.wgrad_plan(M, ntot, K, self.WG_TARGET[key])
                buf = torch.zeros(S, ntot, ldk, dtype=f32, device=dev)
                bufs[key] = buf
                rdesc.append(_ReduceDesc(buf.data_ptr(), ntot * ldk, self.poff[self.pidx[id(prm)]], S, ntot, K, ldk))
            ib = self.pidx[id(layer.mlp[0].bias)]
            rdesc.append(_ReduceDesc(self.bpart[l].data_ptr(), Hm, self.poff[ib], self.bpart[l].shape[0], 1, Hm, Hm))
            self.slabs.append(bufs)
        self._install_reduce(rdesc)

    # ---- stages ----------------------------------------------------------------------------------------
    def _stage_gather(self, s, with_adam=False, ids=None, hops=None, skip_rows=0):
        L, st = self.L, self.store
        if ids is None:
            ids = self.ids_set[s]
        segs = [(st.data, ids[:self.rows[0]], self.x0_set[s], self.rows[0], 1)]
        if not self.inplace0:
            segs.append((st.data, ids[self.off[1]:self.off[L + 1]], self.xn0_set[s], self.nrows[0], 1))
        # (D = the real width: the pad columns of the operand buffers were zeroed once and stay zero)
        ops.gather_mean_multi(segs, st.ld, st.dim, st.ld, adam=self._adam_desc() if with_adam else None,
                              hops=hops)

    def _queue_compute(self, par):
        self._q_ids = self.ids_q[par]
        try:
            return super(FusedPoolTrainStep, self)._queue_compute(par)
        finally:
            self._q_ids = None

    def _gemm(self, A, lda, W, C, c_code, ldc, M, N, K, act, Wp=None):
        if Wp is not None and lda % 64 == 0 and lda >= -(-K // 64) * 64:
            ops._linear_packed_launch(A, lda, None, 0, Wp.data_ptr(), None, C, ldc, M, N, K, act, 1, 0, 0, c_code)
            return
        ops._linear_launch(A, lda, None, 0, W.data_ptr(), W.shape[1], None, C, ldc, M, N, K, act, 1, 0, 0, 0,
                           self.code, c_code)

    def _stage_compute(self, s):
        L, B, lib, stream, m = self.L, self.B, nat.lib(), ops._stream(), self.model
        for l, layer in enumerate(self.layers):
            R, Hm, h, din = self.rows[l], self.Hm[l], self.h[l], self.din[l]
            nb, ldnb, nrows = self._nb_operand(l, s)
            for k in range(L - l):                       # one K3 launch per hop: its fan-out is the segment
                r0, r1 = self.off[k], self.off[k + 1]
                a0 = self.off[k + 1] - self.off[1]
                a_ptr = nb.data_ptr() if nrows is not None else nb[a0:].data_ptr()
                r_ptr = nrows[a0:].data_ptr() if nrows is not None else None
                is_max = self.pool_mode == nat.POOL_MAX
                is_bf = self.code == nat.BF16
                tail = (self.pooled[l][r0:r1].data_ptr(), Hm, self.argmax[l][r0:r1].data_ptr() if is_max else None,
                        self.pooled_b[l][r0:r1].data_ptr() if is_bf else None, self.pooled_b[l].shape[1],
                        None if is_max else self.argmax[l][a0:].data_ptr(), stream)
                if self.wm_p[l] is not None and ldnb % 64 == 0 and ldnb >= -(-din // 64) * 64:
                    nat.check(lib.gsage_pool_mlp_packed(
                        a_ptr, ldnb, r_ptr, self.wm_p[l].data_ptr(), layer.mlp[0].bias.data_ptr(),
                        self.size[k], self.fan[k + 1], Hm, din, self.pool_mode, *tail), "pool_mlp_packed")
                else:
                    nat.check(lib.gsage_pool_mlp(
                        a_ptr, self.code, ldnb, r_ptr, self.wm[l].data_ptr(), self.wm[l].shape[1],
                        layer.mlp[0].bias.data_ptr(), self.size[k], self.fan[k + 1], Hm, din, self.pool_mode,
                        *tail), "pool_mlp")
            x, ldx = self._x_operand(l, s)
            last = l == L - 1
            out, code = self.hout[l], (nat.F32 if last else self.code)
            act = nat.ACT_NONE if last else nat.ACT_RELU
            self._gemm(x.data_ptr(), ldx, self.wx[l], out.data_ptr(), code, 2 * h, R, h, din, act, self.wx_p[l])
            self._gemm(self.pooled_b[l].data_ptr(), self.pooled_b[l].shape[1], self.wn[l],
                       out.data_ptr() + h * out.element_size(), code, 2 * h, R, h, Hm, act, self.wn_p[l])
        C, D2 = m.fc.weight.shape
        tg = self.queue[1] if self.queue else self.tg_set[s].view(-1)
        nat.check(lib.gsage_head_ce(self.hout[L - 1].data_ptr(), 2 * self.h[L - 1], m.fc.weight.data_ptr(),
                                    m.fc.bias.data_ptr(), tg.data_ptr(), B, C, D2, self.preds.data_ptr(),
                                    self.dc[L - 1].data_ptr(), self.code, 2 * self.h[L - 1], None, None, None,
                                    self.head_scratch.data_ptr(),
                                    self.batch_idx.data_ptr() if self.queue else None,
                                    self.queue[2] if self.queue else 0, stream), "head_ce")
        self._backward_levels(s)

    def _backward_levels(self, s):
        L, lib, stream = self.L, nat.lib(), ops._stream()
        for l in range(L - 1, -1, -1):
            R, NR, Hm, h, din = self.rows[l], self.nrows[l], self.Hm[l], self.h[l], self.din[l]
            dc = self.dc[l]
            # d pooled = dC[:, h:] Wn
            self._gemm(dc.data_ptr() + h * self.esz, 2 * h, self.wnT[l], self.dpool[l].data_ptr(), nat.F32, Hm, R, Hm,
                       h, nat.ACT_NONE)
            for k in range(L - l):
                r0, r1 = self.off[k], self.off[k + 1]
                a0 = self.off[k + 1] - self.off[1]
                if self.pool_mode == nat.POOL_MEAN:
                    nat.check(lib.gsage_pool_route_mean_bwd(self.dpool[l][r0:r1].data_ptr(), Hm,
                                                            self.argmax[l][a0:].data_ptr(), self.size[k],
                                                            self.fan[k + 1], Hm, self.ghc[l][a0:].data_ptr(), self.code,
                                                            Hm, self.bpart[l][k * self.NPART:].data_ptr(), self.NPART,
                                                            stream), "pool_route_mean_bwd")
                    continue
                nat.check(lib.gsage_pool_route_bwd(self.dpool[l][r0:r1].data_ptr(), Hm, self.pooled[l][r0:r1].data_ptr(),
                                                   Hm, self.argmax[l][r0:r1].data_ptr(), Hm, self.size[k],
                                                   self.fan[k + 1], Hm, self.ghc[l][a0:].data_ptr(), self.code, Hm,
                                                   stream), "pool_route_bwd")
                nat.check(lib.gsage_pool_bias_partials(self.dpool[l][r0:r1].data_ptr(), Hm,
                                                       self.pooled[l][r0:r1].data_ptr(), Hm, self.size[k], Hm,
                                                       self.bpart[l][k * self.NPART:].data_ptr(), self.NPART,
                                                       stream),
                          "pool_bias_partials")
            if l > 0:
                self._gemm(dc.data_ptr(), 2 * h, self.wxT[l], self.dxb[l].data_ptr(), nat.F32, din, R, din, h,
                           nat.ACT_NONE)
                self._gemm(self.ghc[l].data_ptr(), Hm, self.wmT[l], self.dnb[l].data_ptr(), nat.F32, din, NR, din,
                           Hm, nat.ACT_NONE)
                below = self.hout[l - 1]
                nat.check(lib.gsage_pool_merge_bwd(below.data_ptr(), self.code, below.stride(0), self.dxb[l].data_ptr(),
                                                   din, R,
                                                   self.dnb[l].data_ptr(), din, self.off[1],
                                                   self.dc[l - 1].data_ptr(), self.dc[l - 1].stride(0),
                                                   self.rows[l - 1], din, stream), "pool_merge_bwd")
        probs = []
        for l in range(L - 1, -1, -1):
            R, NR, Hm, h, din = self.rows[l], self.nrows[l], self.Hm[l], self.h[l], self.din[l]
            x, ldx = self._x_operand(l, s)
            nb, ldnb, nrows = self._nb_operand(l, s)
            dc = self.dc[l]
            T = self.WG_TARGET
            probs.append((dc[:, :h], x, ldx, 0, R, h, din, h, self.slabs[l]["x"], T["x"]))
            probs.append((dc[:, h:], self.pooled_b[l], self.pooled_b[l].shape[1], 0, R, h, Hm, h, self.slabs[l]["n"],
                          T["n"]))
            probs.append((self.ghc[l], nb, ldnb, 0, NR, Hm, din, Hm, self.slabs[l]["m"], T["m"], nrows))
        for i in range(0, len(probs), 8):
            ops.wgrad_multi(probs[i:i + 8])
        self._stage_finalize(s)


class FusedAttnTrainStep(FusedMeanTrainStep):
    """train_step for attention aggregators (reference nn_modules.py:289-321; BASELINE config 4's aggregator)
    with no autograd and no framework glue, on the machinery of FusedMeanTrainStep (flat buckets, fused
    multi-hop sampler, head kernel, finalisation + Adam, command lists, batch queue).  Level l turns the rows
    of hops 0 .. L-l ("In") into the rows of hops 0 .. L-l-1:

      forward    K5   hid = tanh(In W0^T);  K5  a = hid W2^T        att(.) ONCE per row: the reference applies the
                                                                    same MLP to a row as "x" and as a neighbour
                 K4   per hop: scores <a_child, a_parent>, softmax over the fan-out, agg = sum w * raw child row
                 K5   out[:, :h] = act(In[:rows_x] Wx^T);  K5  out[:, h:] = act(agg Wn^T)
      backward   K5   d agg = dC[:, h:] Wn
                 K4'  per hop: d a(child), d a(parent)  (softmax backward inside)
                 K5   (d a W2), tanh backward -> d hid;   l > 0: K5 d In(att) = d hid W0, K5 dX = dC[:, :h] Wx,
                      one merge kernel adds them to ws * d agg(parent) and applies the ReLU mask -> dC of level l-1
                 K5b  fc_x, fc_neib, att.0, att.2 weight gradients of every level in one grouped launch
    Level 0 reads its rows from ONE buffer gathered per step (all hops, next batch, beside Adam and K1): the att
    MLP, K4, the x projection and two of the four weight gradients all want plain row-major operands."""

    HA_LD = 64            # leading dimension of the 32-wide att activations (whole 128-byte bf16 lines)
    ROW_HIST = 1 << 15    # updates whose constants are kept for deferred table rows (sync_rows() before it wraps)
    WG_TARGET = 120       # K5b workgroups per problem (eight problems share the launch)

    @staticmethod
    def supports(model, feats):
        from .nn_modules import AttentionAggregator
        layers = list(model.agg_layers.children())
        if not layers or not all(type(l) is AttentionAggregator and l.combine_fn is concat_combine for l in layers):
            return False
        codes = [_split_activation(l.activation)[0] for l in layers]
        if codes[:-1] != [nat.ACT_RELU] * (len(layers) - 1) or codes[-1] != nat.ACT_NONE:
            return False
        from .nn_modules import NodeEmbeddingPrep
        if isinstance(model.prep, NodeEmbeddingPrep):
            # BASELINE config 4: no features, trainable node embeddings (+ affine) are the level-0 rows
            if feats is not None or model.prep.input_dim or model.prep.embedding_dim % 8 != 0:
                return False
            if not model.prep.embedding.weight.is_cuda:
                return False
        elif not isinstance(model.prep, IdentityPrep) or not isinstance(feats, FeatureStore):
            return False
        elif feats.dtype not in (torch.bfloat16, torch.float32) or not feats.is_cuda or feats.ld % 8 != 0:
            return False
        if not isinstance(model.train_sampler, SparseUniformNeighborSampler) or model.train_sampler.rng != "philox":
            return False
        if any(not (2 <= fn.keywords["n_samples"] <= 32) for fn in model.train_sample_fns) or len(layers) > 4:
            return False                                     # K4 keeps a parent's softmax in one wave's lanes
        ha = {int(l.att[0].weight.shape[0]) for l in layers}
        return all(l.output_dim_ % 8 == 0 for l in layers) and ha <= {32} and \
            all(tuple(l.att[2].weight.shape) == (32, 32) for l in layers)

    # ---- construction ------------------------------------------------------------------------------
    def _init_levels(self, example_ids, example_targets):
        from .nn_modules import NodeEmbeddingPrep
        feats, dev, L = self.store, self.dev, self.L
        T, f32 = self.tdt, torch.float32
        self.Ha = 32
        self.h = [l.output_dim_ for l in self.layers]
        self.emb = isinstance(self.model.prep, NodeEmbeddingPrep)
        self.lazy_rows = self.emb and os.environ.get("GSAGE_DENSE_TABLE_ADAM", "0") != "1"
        if self.emb:
            assert self.ddp is None, "the embedding-prep engine is single-GPU (data-parallel runs use the module path)"
            E = int(self.model.prep.embedding_dim)
            d0, ld0 = E, _r64(E) if T == torch.bfloat16 else E
        else:
            d0, ld0 = feats.dim, feats.ld
        self.din = [d0] + [2 * h for h in self.h[:-1]]
        self.ldin = [ld0] + [2 * h for h in self.h[:-1]]
        self.rows = [self.off[L - l] for l in range(L)]              # x rows of level l
        self.rall = [self.off[L - l + 1] for l in range(L)]          # all input rows of level l
        assert all(d % 8 == 0 for d in self.ldin)
        descs = []

        def copies(prm, need_t):
            r, c = prm.shape
            w = torch.zeros(r, _r64(c), dtype=T, device=dev)
            wt = torch.zeros(c, _r64(r), dtype=T, device=dev) if need_t else None
            descs.append(_PrepDesc(prm.data_ptr(), w.data_ptr(), wt.data_ptr() if need_t else None, r, c, w.shape[1],
                                   wt.shape[1] if need_t else 0, None, 0, int(self.code == nat.F32), 0))
            return w, wt
        self.w0, self.w0T, self.w2, self.w2T, self.wx, self.wxT, self.wn, self.wnT = ([] for _ in range(8))
        for l, layer in enumerate(self.layers):
            ing = l > 0 or self.emb                  # does this level's input need a gradient?
            a, b = copies(layer.att[0].weight, ing); self.w0.append(a); self.w0T.append(b)
            a, b = copies(layer.att[2].weight, True); self.w2.append(a); self.w2T.append(b)
            a, b = copies(layer.fc_x.weight, ing); self.wx.append(a); self.wxT.append(b)
            a, b = copies(layer.fc_neib.weight, True); self.wn.append(a); self.wnT.append(b)
        if self.emb:
            self.wp, self.wpT = copies(self.model.prep.fc.weight, True)
        self.descs = torch.frombuffer(bytearray(bytes((_PrepDesc * len(descs))(*descs))), dtype=torch.uint8).to(dev)
        self.n_desc = len(descs)
        self.max_elems = max(d.rows * d.cols for d in descs)

        # level-0 rows of every hop.  Feature rows are read IN PLACE through the frontier's row list (K5 / K4 / K4' /
        # K5b all take one; GSAGE_ATTN_COPY_ROWS=1: gathered once per step into one buffer per batch in flight, as
        # the embedding prep needs anyway for its output rows); the gather launch then only carries the seeds' rows
        self.inplace0 = (not self.emb) and os.environ.get("GSAGE_ATTN_COPY_ROWS", "0") != "1"
        self._q_ids, self._cur_ids = None, self.ids_set[0]
        self.g0_set = [torch.zeros(self.B if self.inplace0 else self.rall[0], self.ldin[0], dtype=T, device=dev)
                       for _ in range(self.nset)]
        if self.emb:
            prep, RA0, E = self.model.prep, self.rall[0], self.din[0]
            self.table = prep.embedding.weight                 # a view of the flat parameter bucket
            assert self.pidx[id(self.table)] == 0 and self.table.shape[1] == E and self.table.numel() % 4 == 0
            self.seed_rows = torch.full((self.B,), int(prep.n_nodes), dtype=torch.int64, device=dev)
            self.eraw = torch.zeros(RA0, self.ldin[0], dtype=T, device=dev)     # embedding rows as gathered (operand type)
            self.din0f = torch.zeros(RA0, E, dtype=f32, device=dev)             # d prep output
            self.din0 = self.din0f if T == f32 else torch.zeros(RA0, self.ldin[0], dtype=T, device=dev)
            self.deraw = torch.zeros(RA0, E, dtype=f32, device=dev)             # d embedding rows
            self.bpart = torch.zeros(256, E, dtype=f32, device=dev)             # prep.fc.bias gradient partials
            self.seed_grad = torch.zeros(min(16, self.B), E, dtype=f32, device=dev)   # partial sums of the gradient of
            #                                                                             the spare row the seeds read
            self._cur_ids = self.ids_set[0]
        Ha, HL = self.Ha, self.HA_LD
        z = lambda *shape, dt=f32: torch.zeros(*shape, dtype=dt, device=dev)
        self.hid, self.a, self.agg, self.aggc, self.ws, self.hout, self.dc = ([] for _ in range(7))
        self.dagg, self.dan, self.dax, self.da, self.dhid, self.datt, self.dx = ([] for _ in range(7))
        for l in range(L):
            R, RA, ld, h = self.rows[l], self.rall[l], self.ldin[l], self.h[l]
            last = l == L - 1
            self.hid.append(z(RA, HL, dt=T)); self.a.append(z(RA, Ha))
            self.agg.append(z(R, ld)); self.aggc.append(z(R, ld, dt=T)); self.ws.append(z(RA - self.off[1]))
            self.hout.append(z(R, 2 * h, dt=f32 if last else T)); self.dc.append(z(R, 2 * h, dt=T))
            self.dagg.append(z(R, ld)); self.dan.append(z(RA, Ha)); self.dax.append(z(RA, Ha))
            self.da.append(z(RA, HL, dt=T)); self.dhid.append(z(RA, HL, dt=T))
            ing = l > 0 or self.emb
            self.datt.append(z(RA, ld) if ing else None); self.dx.append(z(R, ld) if ing else None)
        self.off_host = (ctypes.c_int64 * 6)(*([int(v) for v in self.off[:L + 1]] + [0] * (5 - L)))
        self.fan_host = (ctypes.c_int32 * 6)(*([int(v) for v in self.fan[:L + 1]] + [1] * (5 - L)))

    def _init_head(self, loss_fn, example_targets):
        super(FusedAttnTrainStep, self)._init_head(loss_fn, example_targets)
        self.fused_tail = False
        # the regression head of the Pokec problem (F.l1_loss with the reference's [B,1]-vs-[B] broadcast) as one kernel
        from .problem import ProblemLosses
        C, D2 = self.model.fc.weight.shape
        probe = torch.randn(3, 4, device=self.dev)
        ident = self.post is None or torch.equal(self.post(probe), probe)
        self.fused_l1 = bool(self.loss_fn is ProblemLosses.regression_mae and C == 1 and ident and self.B <= 2048 and
                             example_targets.dtype == torch.float32 and example_targets.numel() == self.B and
                             self.B > 1 and os.environ.get("GSAGE_TORCH_HEAD", "0") != "1")
        if self.fused_l1:
            self.preds = torch.zeros(self.B, 1, dtype=torch.float32, device=self.dev)

    def _in(self, l, s):
        """input rows of level l (all hops it reads): (row block, leading dimension, row list or None) -- with a
        row list, input row i is block[list[i]] (level 0 read in place from the feature table)"""
        if l > 0:
            return self.hout[l - 1], self.ldin[l], None
        if self.inplace0:
            return self.store.data, self.ldin[0], self._cur_ids
        return self.g0_set[s], self.ldin[0], None

    def _wg_problems(self, l, s):
        """(dC, A, lda, M, Ntot, K, parameter) of the four weight gradients of level l"""
        inp, ld, rows = self._in(l, s)
        h, Ha, D, layer = self.h[l], self.Ha, self.din[l], self.layers[l]
        probs = [(self.dc[l][:, :h], inp, ld, self.rows[l], h, D, layer.fc_x.weight, rows),
                 (self.dc[l][:, h:], self.aggc[l], ld, self.rows[l], h, D, layer.fc_neib.weight, None),
                 (self.da[l], self.hid[l], self.HA_LD, self.rall[l], Ha, Ha, layer.att[2].weight, None),
                 (self.dhid[l], inp, ld, self.rall[l], Ha, D, layer.att[0].weight, rows)]
        if l == 0 and self.emb:      # the prep's affine: d out^T x embedding rows
            probs.append((self.din0, self.eraw, self.eraw.stride(0), self.rall[0], D, D, self.model.prep.fc.weight, None))
        return probs

    def _init_reduce(self):
        dev, f32 = self.dev, torch.float32
        rdesc, self.slabs = [], []
        for l in range(self.L):
            bufs = []
            for (dC, A, lda, M, ntot, K, prm, _rows) in self._wg_problems(l, 0):
                rps, S, ldk = ops.wgrad_plan(M, ntot, K, self.WG_TARGET)
                buf = torch.zeros(S, ntot, ldk, dtype=f32, device=dev)
                bufs.append(buf)
                rdesc.append(_ReduceDesc(buf.data_ptr(), ntot * ldk, self.poff[self.pidx[id(prm)]], S, ntot, K, ldk))
            self.slabs.append(bufs)
        if self.emb:
            E = self.din[0]
            ib = self.pidx[id(self.model.prep.fc.bias)]
            rdesc.append(_ReduceDesc(self.bpart.data_ptr(), E, self.poff[ib], self.bpart.shape[0], 1, E, E))
        self._install_reduce(rdesc)
        if self.emb:
            # the table's gradient comes from scatter-adds, its squared norm from a pass of its own whose
            # partials sit behind the finalisation's in the same array
            self.n_tab = int(self.table.numel())
            self.n_tab_partial = 1024 if self.lazy_rows else nat.lib().gsage_adam_partials(self.n_tab)
            self.partial = torch.zeros(self.n_partial + self.n_tab_partial, dtype=torch.float32, device=self.dev)
        if self.emb and self.lazy_rows:
            # deferred row updates (gsage_rows_*): a step touches its frontier's rows, everything else is
            # replayed -- bit for bit -- when it is next read (sync_rows: module forward, state_dict, eval)
            n_rows, E = int(self.table.shape[0]), self.din[0]
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
            self.model._settle_rows = self.sync_rows
            emb_mod = self.model.prep.embedding
            self._row_hooks = [emb_mod.register_forward_pre_hook(lambda *_: self.sync_rows()),
                               emb_mod.register_state_dict_pre_hook(lambda *_: self.sync_rows())]

    def _uncovered(self):
        return int(self.table.numel()) if getattr(self, "emb", False) else 0

    # ---- stages ----------------------------------------------------------------------------------------
    def _stage_gather(self, s, with_adam=False, ids=None, hops=None, skip_rows=0):
        if self.emb:
            return               # nothing to gather ahead: the embedding rows are weights (read after Adam)
        st = self.store
        if ids is None:
            ids = self.ids_set[s]
        n0 = self.B if self.inplace0 else self.rall[0]        # (in place: a token segment carries Adam and K1)
        segs = [(st.data, ids[:n0], self.g0_set[s], n0, 1)]
        ops.gather_mean_multi(segs, st.ld, st.dim, st.ld, adam=self._adam_desc() if with_adam else None, hops=hops)

    # embedding prep: the level-0 rows are prep.fc(embedding[ids]) (nn_modules.py:144-155) -- computed at the start of
    # the step from the CURRENT table, their gradient scattered back into the table's (dense) gradient at the end
    def _prep_forward(self, s):
        lib, stream, prep = nat.lib(), ops._stream(), self.model.prep
        ids, B, RA0, E = self._cur_ids, self.B, self.rall[0], self.din[0]
        tab = self.table
        if self.lazy_rows:       # the rows this step reads, brought up to the last update
            nat.check(lib.gsage_rows_catch_up(ctypes.byref(self.row_desc), self.seed_rows.data_ptr(), 1,
                                              ids[B:RA0].data_ptr(), RA0 - B, 0, stream), "rows_catch_up")
        # fp32 table rows -> the operand type in the gather itself (seeds read the spare row n_nodes)
        segs = [(tab, self.seed_rows, self.eraw[:B], B, 1), (tab, ids[B:RA0], self.eraw[B:], RA0 - B, 1)]
        ops.gather_mean_multi(segs, E, E, self.eraw.stride(0))
        ops._linear_launch(self.eraw.data_ptr(), self.eraw.stride(0), None, 0, self.wp.data_ptr(), self.wp.shape[1],
                           prep.fc.bias.data_ptr(), self.g0_set[s].data_ptr(), self.ldin[0], RA0, E, E, nat.ACT_NONE, 1,
                           0, 0, 0, self.code, self.code)

    def _prep_backward(self, s):
        """level 0's input gradient -> prep.fc (weight: K5b problem of level 0; bias: column sums) -> table gradient."""
        lib, stream = nat.lib(), ops._stream()
        ids, B, RA0, E, L = self._cur_ids, self.B, self.rall[0], self.din[0], self.L
        ld = self.ldin[0]
        lp = self.din0 is not self.din0f          # fp32 for the bias gradient's column sums + the GEMMs' operand copy
        nat.check(lib.gsage_attn_merge_bwd2(None, self.code, 0, self.datt[0].data_ptr(), ld, self.dx[0].data_ptr(), ld,
                                            self.rows[0], self.dagg[0].data_ptr(), ld, self.ws[0].data_ptr(),
                                            self.din0f.data_ptr(), nat.F32, E, RA0, E, L + 1, self.off_host,
                                            self.fan_host, self.din0.data_ptr() if lp else None,
                                            self.din0.stride(0) if lp else 0, stream), "attn_merge_bwd")
        nat.check(lib.gsage_colsum_partials(self.din0f.data_ptr(), E, RA0, E, self.bpart.data_ptr(),
                                            self.bpart.shape[0], stream), "colsum_partials")
        self._gemm(self.din0.data_ptr(), self.din0.stride(0), self.wpT, self.deraw.data_ptr(), nat.F32, E, RA0, E, E,
                   nat.ACT_NONE)
        g = self._grad_slice(self.table)
        # every seed reads the SAME spare row: its B gradient rows are summed first (B atomics onto one row took 15 us)
        ns = self.seed_grad.shape[0]             # (partial sums: one workgroup summing B rows alone took 14 us)
        nat.check(lib.gsage_colsum_partials(self.deraw.data_ptr(), E, B, E, self.seed_grad.data_ptr(), ns, stream),
                  "colsum_partials")
        for rows, idv, M in ((self.seed_grad, self.seed_rows, ns), (self.deraw[B:], ids[B:RA0], RA0 - B)):
            nat.check(lib.gsage_scatter_add_rows(rows.data_ptr(), E, idv.data_ptr(), M, 1, E, 1.0, g.data_ptr(), E,
                                                 stream), "scatter_add_rows")

    def _stage_opt(self):
        if not self.emb:
            return super(FusedAttnTrainStep, self)._stage_opt()
        lib, stream = nat.lib(), ops._stream()
        d = self._adam_desc()
        nt, B, RA0, E = self.n_tab, self.B, self.rall[0], self.din[0]
        g = self._grad_slice(self.table)
        n_all = self.n_partial + self.n_tab_partial
        if self.lazy_rows:
            ids, rd = self._cur_ids, ctypes.byref(self.row_desc)
            lists = (self.seed_rows.data_ptr(), 1, ids[B:RA0].data_ptr(), RA0 - B, 0)
            nat.check(lib.gsage_rows_sqnorm(rd, *lists, self.partial[self.n_partial:].data_ptr(), self.n_tab_partial,
                                            stream), "rows_sqnorm")
            nat.check(lib.gsage_rows_adam(rd, *lists, self.partial.data_ptr(), n_all, stream), "rows_adam")
            o, n = nt, self.flat_p.numel() - nt
            nat.check(lib.gsage_clip_adam_step(self.flat_p[o:].data_ptr(), self.flat_g[o:].data_ptr(),
                                               self.flat_m[o:].data_ptr(), self.flat_v[o:].data_ptr(), n,
                                               self.partial.data_ptr(), d.lr, d.step, d.beta1, d.beta2, d.eps,
                                               d.weight_decay, d.max_norm, d.norm_out, 1, n_all, d.prep_descs, d.n_prep,
                                               None, 0, None, 0, stream), "clip_adam_step")
            return
        nat.check(lib.gsage_grad_sqnorm(g.data_ptr(), nt, self.partial[self.n_partial:].data_ptr(), self.n_tab_partial,
                                        stream), "grad_sqnorm")
        # the table (16-byte lanes, no operand copies), then everything else (operand copies refreshed)
        # (flag 2 on the table: its gradient is zeroed below, no need to write the clipped values back)
        for (o, n, prep, n_prep, cur) in ((0, nt, None, 0, 3), (nt, self.flat_p.numel() - nt, d.prep_descs, d.n_prep, 1)):
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
        afterwards).  Runs by itself before the embedding module's forward and state_dict; call it before reading
        `prep.embedding.weight` or the optimizer buckets directly."""
        if not getattr(self, "lazy_rows", False) or not self._rows_dirty:
            return
        nat.check(nat.lib().gsage_rows_catch_up_all(ctypes.byref(self.row_desc), 0, ops._stream()), "rows_catch_up_all")
        self._rows_dirty, self._rows_since = False, 0

    def close(self):
        """Settle the deferred rows and detach from the model (hooks on the embedding module, model._settle_rows):
        call before pickling the model or when the engine is done with."""
        if getattr(self, "lazy_rows", False):
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
        if getattr(self, "lazy_rows", False):
            if self._rows_since >= self.ROW_HIST - 2:
                self.sync_rows()
            self._rows_dirty = True
            self._rows_since += 1

    def __call__(self, *args, **kwargs):
        self._rows_tick()
        return super(FusedAttnTrainStep, self).__call__(*args, **kwargs)

    def step_queue(self):
        self._rows_tick()
        return super(FusedAttnTrainStep, self).step_queue()

    # queue mode with an embedding prep: sampling runs ahead, nothing else can (the rows are weights)
    def _queue_prime(self):
        if not self.emb:
            return super(FusedAttnTrainStep, self)._queue_prime()
        self._stage_sample(0, ids=self.ids_q[0])
        self._stage_sample(0, ids=self.ids_q[1], ahead=True)

    def _queue_compute(self, par):
        self._q_ids = self.ids_q[par]
        try:
            return super(FusedAttnTrainStep, self)._queue_compute(par)
        finally:
            self._q_ids = None

    def _queue_front(self, par, with_adam):
        if not self.emb:
            return super(FusedAttnTrainStep, self)._queue_front(par, with_adam)
        assert with_adam
        self._cur_ids = self.ids_q[par]
        self._stage_opt()                                  # Adam(i) + zeroing of the rows batch i touched
        self._stage_sample(0, ids=self.ids_q[par], ahead=True)      # batch i+2 (batch i's frontier is done with)

    def _run_sequential(self, s):
        if self.emb:
            self._cur_ids = self.ids_set[s]
        return super(FusedAttnTrainStep, self)._run_sequential(s)

    def _gemm(self, A, lda, W, C, c_code, ldc, M, N, K, act, rows=None):
        ops._linear_launch(A, lda, rows.data_ptr() if rows is not None else None, 0, W.data_ptr(), W.shape[1], None, C,
                           ldc, M, N, K, act, 1, 0, 0, 0, self.code, c_code)

    @staticmethod
    def _child_rows(inp, rows, c0):
        """(table pointer, id pointer) of the rows from position c0 on, for K4 / K4'"""
        if rows is None:
            return inp[c0:].data_ptr(), None
        return inp.data_ptr(), rows[c0:].data_ptr()

    def _stage_compute(self, s):
        L, B, lib, stream, m = self.L, self.B, nat.lib(), ops._stream(), self.model
        Ha, HL, esz = self.Ha, self.HA_LD, self.esz
        self._cur_ids = self._q_ids if self._q_ids is not None else self.ids_set[s]
        if self.emb:
            self._prep_forward(s)
        for l in range(L):
            R, RA, h, D = self.rows[l], self.rall[l], self.h[l], self.din[l]
            inp, ld, rows = self._in(l, s)
            self._gemm(inp.data_ptr(), ld, self.w0[l], self.hid[l].data_ptr(), self.code, HL, RA, Ha, D, nat.ACT_TANH, rows)
            nat.check(lib.gsage_attn_mlp2_fwd(self.hid[l].data_ptr(), self.code, HL, self.w2[l].data_ptr(),
                                              self.w2[l].shape[1], self.a[l].data_ptr(), Ha, RA, Ha, stream), "attn_mlp2_fwd")
            for k in range(L - l):                   # K4 writes the aggregate as fp32 and as the next GEMMs' operand
                r0, c0 = self.off[k], self.off[k + 1]
                tab, idp = self._child_rows(inp, rows, c0)
                nat.check(lib.gsage_attn_aggregate_lp(
                    self.a[l][c0:].data_ptr(), Ha, self.a[l][r0:].data_ptr(), Ha, tab, self.code, ld,
                    idp, self.size[k], self.fan[k + 1], Ha, D, self.agg[l][r0:].data_ptr(), ld,
                    self.ws[l][c0 - self.off[1]:].data_ptr(), self.aggc[l][r0:].data_ptr(), ld, stream), "attn_aggregate")
            last = l == L - 1
            out, code = self.hout[l], (nat.F32 if last else self.code)
            act = nat.ACT_NONE if last else nat.ACT_RELU
            self._gemm(inp.data_ptr(), ld, self.wx[l], out.data_ptr(), code, 2 * h, R, h, D, act, rows)
            self._gemm(self.aggc[l].data_ptr(), ld, self.wn[l], out.data_ptr() + h * out.element_size(), code, 2 * h,
                       R, h, D, act)
        if self.fused_head:
            C, D2 = m.fc.weight.shape
            tg = self.queue[1] if self.queue else self.tg_set[s].view(-1)
            nat.check(lib.gsage_head_ce(self.hout[L - 1].data_ptr(), 2 * self.h[L - 1], m.fc.weight.data_ptr(),
                                        m.fc.bias.data_ptr(), tg.data_ptr(), B, C, D2, self.preds.data_ptr(),
                                        self.dc[L - 1].data_ptr(), self.code, 2 * self.h[L - 1], None, None, None,
                                        self.head_scratch.data_ptr(),
                                        self.batch_idx.data_ptr() if self.queue else None,
                                        self.queue[2] if self.queue else 0, stream), "head_ce")
        elif self.fused_l1:
            out = self.hout[L - 1]
            nat.check(lib.gsage_head_l1(out.data_ptr(), out.stride(0), m.fc.weight.data_ptr(), m.fc.bias.data_ptr(),
                                        self.tg_set[s].data_ptr(), B, out.shape[1], self.preds.data_ptr(),
                                        self.dc[L - 1].data_ptr(), self.code, self.dc[L - 1].stride(0),
                                        self.l1_scratch.data_ptr(), stream), "head_l1")
        else:
            self._torch_head(s)
        self._backward_levels(s)

    def _backward_levels(self, s):
        L, lib, stream = self.L, nat.lib(), ops._stream()
        Ha, HL, esz = self.Ha, self.HA_LD, self.esz
        for l in range(L - 1, -1, -1):
            R, RA, h, D = self.rows[l], self.rall[l], self.h[l], self.din[l]
            inp, ld, rows = self._in(l, s)
            dc = self.dc[l]
            # d agg = dC[:, h:] Wn
            self._gemm(dc.data_ptr() + h * esz, 2 * h, self.wnT[l], self.dagg[l].data_ptr(), nat.F32, ld, R, D, h,
                       nat.ACT_NONE)
            for k in range(L - l):
                r0, c0 = self.off[k], self.off[k + 1]
                tab, idp = self._child_rows(inp, rows, c0)
                nat.check(lib.gsage_attn_bwd(
                    self.dagg[l][r0:].data_ptr(), ld, self.ws[l][c0 - self.off[1]:].data_ptr(),
                    self.a[l][c0:].data_ptr(), Ha, self.a[l][r0:].data_ptr(), Ha, tab, self.code, ld,
                    idp, self.size[k], self.fan[k + 1], Ha, D, self.dan[l][c0:].data_ptr(), Ha,
                    self.dax[l][r0:].data_ptr(), Ha, stream), "attn_bwd")
            # d a = (as a child) + (as a parent); hop 0 is never a child, the last hop never a parent (zeros);
            # d hid = (d a W2) * tanh' -- one pass
            nat.check(lib.gsage_attn_mlp2_bwd(self.dan[l].data_ptr(), Ha, self.dax[l].data_ptr(), Ha,
                                              self.hid[l].data_ptr(), self.code, HL, self.w2T[l].data_ptr(),
                                              self.w2T[l].shape[1], self.da[l].data_ptr(), HL, self.dhid[l].data_ptr(), HL,
                                              RA, Ha, stream), "attn_mlp2_bwd")
            if l > 0 or self.emb:
                self._gemm(self.dhid[l].data_ptr(), HL, self.w0T[l], self.datt[l].data_ptr(), nat.F32, ld, RA, D, Ha,
                           nat.ACT_NONE)
                self._gemm(dc.data_ptr(), 2 * h, self.wxT[l], self.dx[l].data_ptr(), nat.F32, ld, R, D, h, nat.ACT_NONE)
            if l == 0 and self.emb:
                self._prep_backward(s)
            if l > 0:
                below = self.hout[l - 1]
                nat.check(lib.gsage_attn_merge_bwd(
                    below.data_ptr(), self.code, below.stride(0), self.datt[l].data_ptr(), ld, self.dx[l].data_ptr(), ld,
                    R, self.dagg[l].data_ptr(), ld, self.ws[l].data_ptr(), self.dc[l - 1].data_ptr(), self.code,
                    self.dc[l - 1].stride(0), RA, D, L - l + 1, self.off_host, self.fan_host, stream), "attn_merge_bwd")
        probs = []
        for l in range(L - 1, -1, -1):
            for (dC, A, lda, M, ntot, K, prm, rows), slab in zip(self._wg_problems(l, s), self.slabs[l]):
                probs.append((dC, A, lda, 0, M, ntot, K, ntot, slab, self.WG_TARGET, rows))
        for i in range(0, len(probs), 8):
            ops.wgrad_multi(probs[i:i + 8])
        self._stage_finalize(s)


def fused_engine_for(model, feats):
    """The fused train-step engine that covers (model, feats), or None (callers then fall back to
    GSSupervised.train_step, optionally captured by CapturedTrainStep)."""
    for cls in (FusedMeanTrainStep, FusedPoolTrainStep, FusedAttnTrainStep):
        if cls.supports(model, feats):
            return cls
    return None
