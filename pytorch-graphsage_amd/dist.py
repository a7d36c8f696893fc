"""
dist.py -- data-parallel execution of the hot path: one process per GPU, seed-node minibatches
sharded across ranks, ONE flat fp32 gradient all-reduce per step over RCCL/xGMI
(torch.distributed backend "nccl" is RCCL on ROCm; "gloo" for the CPU tests).

The reference is single-process (no collective anywhere), so this has no
reference counterpart; it hooks into GSSupervised.train_step between backward and clip
(models.py:100-101 in the reference's numbering).  Design:
  * graph + features are replicated per GPU (they fit in 288 GB), so the only exchange is the
    gradient: 0.92 MB (mean) / 2.77 MB (max-pool) -- latency-bound, hence a single bucket and a
    single collective rather than per-parameter reductions;
  * every rank takes an equal, contiguous slice of each seed batch; gradients are averaged, which
    equals the single-process gradient of the mean loss over the whole (truncated) batch;
  * sampling is sharding-invariant: in philox mode the counter is the GLOBAL sample index
    (rank offset passed to K1); in compat mode every rank draws the global `sel` matrix from the
    same legacy stream and keeps its rows.
"""
import os

import torch
import torch.distributed as dist


class DataParallel(object):
    def __init__(self, rank, world, device, owns_group):
        self.rank, self.world, self.device = rank, world, device
        self._owns = owns_group
        self._flat = None
        # the library's own RCCL communicator (_native.NativeComm, include/gsage.h "The step's collectives"): with it
        # the fused engines issue the step's collectives as nodes of the step's command list; None (gloo, or
        # GSAGE_NATIVE_COMM=0): the same nodes call torch.distributed through a host callback
        self.comm = None

    def _all_agree(self, ok):
        """True iff `ok` holds on every rank (one tiny all-reduce over the process group)."""
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(int(flag.item()))

    def create_native_comm(self):
        """One RCCL communicator owned by libgsage_hip.so, set up over the process group torch.distributed already
        has (rank 0's 128-byte id travels through broadcast_object_list).  Collective: every rank calls it.  The
        ranks agree after each stage -- a rank that cannot load RCCL must not leave the others waiting inside
        ncclCommInitRank, and either every rank issues the step's collectives through the library or none does."""
        import sys
        from . import _native as nat

        def exchange(raw):
            box = [raw]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        why = None
        try:
            nat.NativeComm.load()
        except Exception as e:
            why = e
        if not self._all_agree(why is None):
            if self.rank == 0:
                print("gsage: no native RCCL communicator (%s); collectives go through torch.distributed"
                      % (why if why is not None else "another rank could not load RCCL"), file=sys.stderr)
            self.comm = None
            return None
        try:
            self.comm = nat.NativeComm(self.rank, self.world, exchange)
        except Exception as e:
            why, self.comm = e, None
        if not self._all_agree(self.comm is not None):
            if self.comm is not None:
                self.comm.close()
            self.comm = None
            if self.rank == 0:
                print("gsage: the native RCCL communicator could not be created on every rank (%s); collectives go "
                      "through torch.distributed" % (why,), file=sys.stderr)
        return self.comm

    # ---- batch sharding -------------------------------------------------------------------
    def shard(self, ids, targets=None):
        """Equal contiguous slices; the (< world) left-over seeds of a batch are dropped so every
        rank has the same M (keeps the mean exact and the compat-mode stream aligned)."""
        per = ids.shape[0] // self.world
        if per < 1:
            raise ValueError("data-parallel batch of %d seeds cannot be split over %d ranks: every rank would "
                             "train on an empty batch (NaN loss)" % (ids.shape[0], self.world))
        lo, hi = self.rank * per, (self.rank + 1) * per
        if targets is None:
            return ids[lo:hi]
        return ids[lo:hi], targets[lo:hi]

    # ---- gradient exchange -------------------------------------------------------------------
    def sync(self, model):
        """Average gradients across ranks with one all-reduce of one flat fp32 bucket."""
        opt = getattr(model, "optimizer", None)
        if hasattr(opt, "flat_g") and opt.owns():
            # optim.FlatAdam: the gradients already ARE one flat bucket -- no copy in, no copy out
            opt.flat_g.div_(self.world)
            dist.all_reduce(opt.flat_g, op=dist.ReduceOp.SUM)
            return
        params = [p for p in model.parameters() if p.requires_grad]
        total = sum(p.numel() for p in params)
        if self._flat is None or self._flat.numel() != total or self._flat.device != params[0].device:
            self._flat = torch.zeros(total, dtype=torch.float32, device=params[0].device)
        flat = self._flat
        off = 0
        for p in params:
            n = p.numel()
            if p.grad is None:
                flat[off:off + n].zero_()
            else:
                flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        flat.div_(self.world)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        off = 0
        for p in params:
            n = p.numel()
            if p.grad is None:
                p.grad = flat[off:off + n].view_as(p).clone()
            else:
                p.grad.copy_(flat[off:off + n].view_as(p))
            off += n

    def broadcast_parameters(self, model):
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src=0)

    def barrier(self):
        dist.barrier()

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None
        if self._owns and dist.is_initialized():
            dist.destroy_process_group()


def init_from_env(cuda=True):
    """Returns a DataParallel handle when launched by torch.distributed.run with WORLD_SIZE > 1,
    else None.  Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and os.environ.get("GSAGE_FORCE_DDP", "0") != "1":
        return None      # (GSAGE_FORCE_DDP=1: exercise the collective path with a 1-rank group)
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("MASTER_PORT", "29511")
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if cuda:
        if os.environ.get("GSAGE_DIST_BACKEND") == "gloo" and torch.cuda.device_count() == 1:
            local = 0                 # several ranks sharing the only GPU (single-GPU test boxes)
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
        # "nccl" is RCCL on ROCm.  GSAGE_DIST_BACKEND=gloo lets several ranks share ONE GPU (RCCL
        # refuses duplicate devices): tests/test_gpu_dist.py runs the data-parallel engine that way.
        backend = os.environ.get("GSAGE_DIST_BACKEND", "nccl")
    else:
        device = torch.device("cpu")
        backend = "gloo"
    owns = not dist.is_initialized()
    if owns:
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    ddp = DataParallel(rank, world, device, owns)
    if cuda and backend == "nccl" and os.environ.get("GSAGE_NATIVE_COMM", "1") == "1":
        ddp.create_native_comm()
    return ddp


def attach(model, ddp, seed=0):
    """Wire a GSSupervised replica into the group: same initial weights everywhere, gradient sync
    after backward, sharding-invariant sampling."""
    ddp.broadcast_parameters(model)
    model.grad_sync = ddp.sync
    for s in (model.train_sampler, model.val_sampler):
        if hasattr(s, "shard"):
            s.shard = (ddp.rank, ddp.world)
            s.seed = int(seed)
    return model
