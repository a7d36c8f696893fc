"""Data-parallel path on CPU: world_size 2, gloo backend (the N>1 path of dist.py; on the GPU box
the same code runs over RCCL).  A 2-rank step on a sharded batch must equal the single-process step
on the whole batch: same samples (sharding-invariant sampler), same averaged gradient, same weights."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp
from scipy import sparse
from torch.nn import functional as F

from conftest import ROOT, pkg


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _build(gs, rng_mode, agg="mean"):
    rng = np.random.RandomState(0)
    n, D, C = 200, 12, 4
    deg = rng.randint(0, 9, size=n + 1)
    deg[0], deg[n] = 0, 2
    indptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    data = rng.randint(1, n + 1, size=int(indptr[-1]))
    adj = sparse.csr_matrix((data, gs.store.row_positions(indptr), indptr), shape=(n + 1, int(deg.max())))
    feats = torch.from_numpy(rng.normal(size=(n + 1, D)).astype(np.float32))
    torch.manual_seed(5)
    gs.nn_modules.SparseUniformNeighborSampler.rng_default = rng_mode
    model = gs.GSSupervised(
        sampler_class=gs.sampler_lookup["sparse_uniform_neighbor_sampler"], adj=adj, train_adj=adj,
        prep_class=gs.prep_lookup["identity"], aggregator_class=gs.aggregator_lookup[agg], input_dim=D,
        n_nodes=n + 1, n_classes=C,
        layer_specs=[{"n_train_samples": 4, "n_val_samples": 4, "output_dim": 8, "activation": F.relu},
                     {"n_train_samples": 3, "n_val_samples": 3, "output_dim": 8, "activation": lambda x: x}])
    gs.nn_modules.SparseUniformNeighborSampler.rng_default = "compat"
    model.train_sampler.seed = 11
    ids = torch.from_numpy(rng.randint(1, n + 1, size=24))
    tg = torch.from_numpy(rng.randint(0, C, size=(24, 1)))
    return model, feats, ids, tg


def _worker(rank, world, port, rng_mode, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    torch.set_num_threads(1)
    gs = pkg()
    ddp = gs.dist.init_from_env(cuda=False)
    assert ddp is not None and ddp.world == world
    model, feats, ids, tg = _build(gs, rng_mode)
    gs.dist.attach(model, ddp, seed=11)
    gs.set_seeds(77)                                   # every rank: same legacy stream
    for _ in range(2):
        i, t = ddp.shard(ids, tg)
        model.train_step(ids=i, feats=feats, targets=t, loss_fn=gs.ProblemLosses.classification)
    if rank == 0:
        torch.save({k: v.clone() for k, v in model.state_dict().items()}, os.path.join(out_dir, "w.pt"))
    ddp.barrier()
    ddp.close()


@pytest.mark.parametrize("rng_mode", ["compat", "philox"])
def test_two_rank_step_equals_single_process(tmp_path, rng_mode):
    gs = pkg()
    model, feats, ids, tg = _build(gs, rng_mode)
    gs.set_seeds(77)
    for _ in range(2):
        model.train_step(ids=ids, feats=feats, targets=tg, loss_fn=gs.ProblemLosses.classification)
    ref = {k: v.clone() for k, v in model.state_dict().items()}

    mp.spawn(_worker, args=(2, _free_port(), rng_mode, str(tmp_path)), nprocs=2, join=True)
    got = torch.load(os.path.join(str(tmp_path), "w.pt"))
    for k in ref:
        assert torch.allclose(got[k], ref[k], rtol=1e-4, atol=1e-5), (rng_mode, k, float((got[k] - ref[k]).abs().max()))


def test_shard_is_equal_and_contiguous():
    gs = pkg()
    d = gs.dist.DataParallel(rank=1, world=3, device=torch.device("cpu"), owns_group=False)
    ids = torch.arange(10)
    a, b = d.shard(ids, ids * 2)
    assert a.tolist() == [3, 4, 5] and b.tolist() == [6, 8, 10]
