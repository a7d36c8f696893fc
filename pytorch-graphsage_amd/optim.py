"""
optim.py -- clip_grad_norm(5) + Adam over ONE flat fp32 bucket, for the autograd path on the GPU.

The reference ends every train_step with torch.nn.utils.clip_grad_norm(params, 5) and
optimizer.step() (models.py:101-102; Adam betas (0.9, 0.999), eps 1e-8, L2 weight decay as at
models.py:69): in stock PyTorch that is a norm kernel per parameter, a stack, a scale per parameter and
the multi-tensor Adam kernels -- ~25 launches and 4-5 passes over every gradient.  Here the Parameters
(and their .grad) become views of flat fp32 buckets and the tail of the step is two launches of
libgsage_hip.so (gsage_clip_adam_step: squared-norm partials, then clip + Adam in one pass; the same
kernels the fused engines use).  With a trainable node-embedding table (Pokec: 1.6 M x 64 fp32, dense
gradient semantics kept) that tail is most of the step.

Same arithmetic as torch.optim.Adam (bias correction, eps outside the square root, L2 decay added to the
gradient); the learning rate is read from `param_groups[0]['lr']` every step (LRSchedule.set_lr), the
step count lives on the device, so a step is hipGraph-capturable.  Data-parallel runs all-reduce
`flat_g` directly (dist.DataParallel.sync).
"""
import torch

from . import _native as nat


class FlatAdam(object):
    def __init__(self, params, lr=0.01, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params if p.requires_grad]
        assert self.params and all(p.is_cuda and p.dtype == torch.float32 for p in self.params), \
            "FlatAdam: fp32 CUDA parameters only"
        dev = self.params[0].device
        self.sizes = [p.numel() for p in self.params]
        self.offsets = [0]
        for n in self.sizes:
            self.offsets.append(self.offsets[-1] + n)
        total = self.offsets[-1]
        self.flat_p = torch.cat([p.detach().reshape(-1) for p in self.params]).contiguous()
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros_like(self.flat_g)
        self.flat_v = torch.zeros_like(self.flat_g)
        self.step_count = torch.zeros(1, dtype=torch.int64, device=dev)
        self.lr_t = torch.tensor([float(lr)], dtype=torch.float32, device=dev)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)        # pre-clip norm of the last step
        self.partial = torch.zeros(nat.lib().gsage_adam_partials(total), dtype=torch.float32, device=dev)
        self.param_groups = [{"params": self.params, "lr": lr, "weight_decay": weight_decay,
                              "betas": tuple(betas), "eps": eps}]
        self._lr_seen = float(lr)
        self._attach()

    # ---- views ----------------------------------------------------------------------------------
    def _attach(self):
        """Make the Parameters (and their .grad) views of this optimizer's buckets.  A Parameter that
        currently lives elsewhere (first attach, or something re-pointed it since -- a fused engine
        trains through buckets of its own) brings its LIVE value along: the bucket must never resurrect
        the weights it held before somebody else trained them."""
        es = self.flat_p.element_size()
        with torch.no_grad():
            for p, o, n in zip(self.params, self.offsets, self.sizes):
                if p.data_ptr() != self.flat_p.data_ptr() + o * es:
                    self.flat_p[o:o + n].copy_(p.data.reshape(-1))
                p.data = self.flat_p[o:o + n].view_as(p)
                p.grad = self.flat_g[o:o + n].view_as(p)

    def owns(self, params=None):
        """True while every Parameter (and its .grad) still is the view this optimizer made."""
        for p, o in zip(self.params, self.offsets):
            es = self.flat_p.element_size()
            if p.data_ptr() != self.flat_p.data_ptr() + o * es:
                return False
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + o * es:
                return False
        return True

    # ---- torch.optim surface ----------------------------------------------------------------------
    def zero_grad(self, set_to_none=False):
        self.flat_g.zero_()
        for p, o, n in zip(self.params, self.offsets, self.sizes):
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + o * 4:
                p.grad = self.flat_g[o:o + n].view_as(p)

    def clip_and_step(self, max_norm=5.0):
        """clip_grad_norm_(params, max_norm) followed by Adam's step, two launches."""
        g = self.param_groups[0]
        lr = g["lr"]
        if torch.is_tensor(lr):
            self.lr_t.copy_(lr.reshape(1))
        elif float(lr) != self._lr_seen:
            self._lr_seen = float(lr)
            self.lr_t.fill_(self._lr_seen)
        b1, b2 = g["betas"]
        nat.check(nat.lib().gsage_clip_adam_step(
            self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.flat_m.data_ptr(), self.flat_v.data_ptr(),
            self.flat_p.numel(), self.partial.data_ptr(), self.lr_t.data_ptr(), self.step_count.data_ptr(),
            b1, b2, g["eps"], g["weight_decay"], float(max_norm), self.grad_norm.data_ptr(), 0, 0, None, 0,
            None, 0, None, 0, torch.cuda.current_stream().cuda_stream), "clip_adam_step")

    def step(self):
        self.clip_and_step(max_norm=3.0e38)           # no clipping

    def state_dict(self):
        """torch.optim.Adam's format ({"state": {i: {step, exp_avg, exp_avg_sq}}, "param_groups":
        [...]}): a checkpoint written here loads into the torch.optim.Adam the reference builds
        (models.py:69) and vice versa."""
        state = {}
        if int(self.step_count.item()) > 0:
            step = self.step_count.to(torch.float32).reshape(())
            for i, (p, o, n) in enumerate(zip(self.params, self.offsets, self.sizes)):
                state[i] = {"step": step.clone(), "exp_avg": self.flat_m[o:o + n].view_as(p).clone(),
                            "exp_avg_sq": self.flat_v[o:o + n].view_as(p).clone()}
        groups = []
        for g in self.param_groups:
            d = {k: v for k, v in g.items() if k != "params"}
            d["params"] = list(range(len(self.params)))
            groups.append(d)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        if "state" not in sd:                          # round-1 format of this class: flat buckets
            self.flat_m.copy_(sd["m"])
            self.flat_v.copy_(sd["v"])
            self.step_count.copy_(sd["step"])
        else:
            st = sd["state"]
            self.flat_m.zero_()
            self.flat_v.zero_()
            steps = set()
            for i, (p, o, n) in enumerate(zip(self.params, self.offsets, self.sizes)):
                e = st.get(i, st.get(str(i)))
                if e is None:
                    continue
                self.flat_m[o:o + n].copy_(e["exp_avg"].reshape(-1))
                self.flat_v[o:o + n].copy_(e["exp_avg_sq"].reshape(-1))
                steps.add(int(e["step"]))
            assert len(steps) <= 1, "FlatAdam keeps one step count: per-parameter counts differ (%r)" % (steps,)
            self.step_count.fill_(steps.pop() if steps else 0)
        for g, s in zip(self.param_groups, sd.get("param_groups", [])):
            g.update({k: v for k, v in s.items() if k != "params"})
