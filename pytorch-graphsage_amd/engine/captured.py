"""
engine/captured.py -- the whole train_step as replayed HIP graphs.

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

from .. import _native as nat


class CapturedTrainStep(object):
    def __init__(self, model, feats, loss_fn, example_ids, example_targets, ddp=None, warmup=3):
        assert example_ids.is_cuda, "CapturedTrainStep needs CUDA tensors"
        self.model, self.feats, self.loss_fn, self.ddp = model, feats, loss_fn, ddp
        self.ids = example_ids.clone()
        self.targets = example_targets.clone()
        dev = self.ids.device

        # clip + Adam over flat buckets (optim.FlatAdam: step count and learning rate live on the
        # device, two launches, capturable; the gradient bucket is also what data-parallel runs exchange)
        from ..optim import FlatAdam
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
