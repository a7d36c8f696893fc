"""-m gpu: the data-parallel fused engine with TWO ranks.  A gpurun box has one GPU and RCCL refuses
two ranks on one device, so both ranks use cuda:0 and exchange gradients over gloo
(GSAGE_DIST_BACKEND=gloo; the collective is the only thing that differs from the RCCL run -- sharding,
sharding-invariant Philox sampling, the one-stage software pipeline around the all-reduce, the
divide-then-sum fallback for backends without AVG and Adam after the exchange are all exercised).
Two ranks x B seeds must reproduce one process x 2B seeds: same samples, averaged gradient = gradient
of the global batch mean, same weights (up to fp32 summation order)."""
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

pytestmark = pytest.mark.gpu
B_RANK, N_BATCH, STEPS = 16, 3, 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _build(gs, agg="mean"):
    rng = np.random.RandomState(0)
    n, D, C = 500, 40, 5
    deg = rng.randint(0, 30, size=n + 1)
    deg[0], deg[n] = 0, 3
    indptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    data = rng.randint(1, n + 1, size=int(indptr[-1]))
    adj = sparse.csr_matrix((data, gs.store.row_positions(indptr), indptr), shape=(n + 1, int(deg.max())))
    feats = rng.normal(size=(n + 1, D)).astype(np.float32)
    feats[0] = 0
    torch.manual_seed(5)
    gs.nn_modules.SparseUniformNeighborSampler.rng_default = "philox"
    specs = [{"n_train_samples": 5, "n_val_samples": 5, "output_dim": 128, "activation": F.relu},
             {"n_train_samples": 3, "n_val_samples": 3, "output_dim": 128, "activation": lambda x: x}]
    model = gs.GSSupervised(sampler_class=gs.sampler_lookup["sparse_uniform_neighbor_sampler"], adj=adj,
                            train_adj=adj, prep_class=gs.prep_lookup["identity"],
                            aggregator_class=gs.aggregator_lookup[agg], input_dim=D, n_nodes=n + 1,
                            n_classes=C, layer_specs=specs, lr_init=0.01, weight_decay=1e-4)
    gs.nn_modules.SparseUniformNeighborSampler.rng_default = "compat"
    model.train_sampler.seed = model.val_sampler.seed = 77
    ids = torch.from_numpy(rng.randint(1, n + 1, size=(N_BATCH, 2 * B_RANK)))
    tg = torch.from_numpy(rng.randint(0, C, size=(N_BATCH, 2 * B_RANK, 1)))
    return model.to("cuda"), feats, ids.to("cuda"), tg.to("cuda")


def _run(gs, model, feats, ids, tg, ddp):
    gs.ops.set_compute_dtype("bf16")
    gs.ops.warmup(torch.device("cuda"))
    store = gs.FeatureStore.from_array(feats, torch.device("cuda"), dtype="bf16")
    eng = gs.engine.fused_engine_for(model, store)(model, store, gs.ProblemLosses.classification, ids[0], tg[0],
                                                    ddp=ddp)
    eng.load_epoch(ids, tg)
    preds = [eng.step_queue().clone() for _ in range(STEPS)]
    torch.cuda.synchronize()
    return torch.stack(preds).cpu(), eng.flat_p.clone().cpu()


def _worker(rank, world, port, out_dir, agg):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": "0", "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "GSAGE_DIST_BACKEND": "gloo"})
    gs = pkg()
    ddp = gs.dist.init_from_env(cuda=True)
    assert ddp is not None and ddp.world == world
    model, feats, ids, tg = _build(gs, agg)
    gs.dist.attach(model, ddp, seed=77)
    lo, hi = rank * B_RANK, (rank + 1) * B_RANK
    preds, w = _run(gs, model, feats, ids[:, lo:hi].contiguous(), tg[:, lo:hi].contiguous(), ddp)
    torch.save({"preds": preds, "w": w}, os.path.join(out_dir, "r%d.pt" % rank))
    ddp.barrier()
    ddp.close()


@pytest.mark.parametrize("agg", ["mean", "max_pool"])
def test_two_rank_engine_equals_single_process_global_batch(tmp_path, agg):
    gs = pkg()
    model, feats, ids, tg = _build(gs, agg)
    ref_preds, ref_w = _run(gs, model, feats, ids, tg, None)
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), agg), nprocs=2, join=True)
    r0 = torch.load(os.path.join(str(tmp_path), "r0.pt"))
    r1 = torch.load(os.path.join(str(tmp_path), "r1.pt"))
    assert torch.equal(r0["w"], r1["w"]), "replicas diverged"
    got = torch.cat([r0["preds"], r1["preds"]], dim=1)
    scale = float(ref_preds.abs().max())
    assert float((got - ref_preds).abs().max()) <= 3e-3 * scale, float((got - ref_preds).abs().max())
    assert float((r0["w"] - ref_w).abs().max()) <= 3e-3 * float(ref_w.abs().max())
