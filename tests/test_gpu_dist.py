"""-m gpu: the data-parallel fused engines with TWO ranks.  A gpurun box has one GPU and RCCL refuses two ranks on one
device, so both ranks use cuda:0 and exchange over gloo (GSAGE_DIST_BACKEND=gloo): the collectives then are
host-call nodes of the SAME one-list step that carries RCCL calls on a multi-GPU node (tests/test_gpu_round4.py runs
that form with a 1-rank RCCL group) -- sharding, sharding-invariant Philox sampling, the exchange on the list's side
stream beside the next batch's gathers, the divide-then-sum fallback for backends without AVG, the norm of the
averaged gradient and Adam after the exchange, and for a trainable embedding table (BASELINE configs[3]) the
sparse-row exchange: every rank's touched row ids + fp32 gradient rows, reduced in one order everywhere.
Two ranks x B seeds must reproduce one process x 2B seeds: same samples, averaged gradient = gradient of the global
batch mean (for the L1 head: of the global [B,1]-vs-[B] pair mean), same weights up to fp32 summation order, and
the two replicas bit-identical to each other."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, pkg
import util

pytestmark = pytest.mark.gpu
B_RANK, N_BATCH, STEPS = 16, 3, 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir, case, capture):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": "0", "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "GSAGE_DIST_BACKEND": "gloo"})
    if capture == "cmdlist+overlap":          # the exchange on the list's side stream, beside the bulk of the gathers
        os.environ["GSAGE_DDP_OVERLAP"] = "1"
        capture = "cmdlist"
    gs = pkg()
    ddp = gs.dist.init_from_env(cuda=True)
    assert ddp is not None and ddp.world == world and ddp.comm is None
    model, feats, loss_fn, ids, tg, prec = util.dp_case(gs, case, n_batch=N_BATCH, global_batch=world * B_RANK)
    gs.dist.attach(model, ddp, seed=77)
    lo, hi = rank * B_RANK, (rank + 1) * B_RANK
    preds, w, eng = util.dp_run(gs, case, model, feats, loss_fn, ids[:, lo:hi].contiguous(), tg[:, lo:hi].contiguous(),
                                prec, ddp, steps=STEPS, capture=capture)
    torch.save({"preds": preds, "w": w, "one_list": eng._one_list_ddp(), "mode": eng.capture_mode},
               os.path.join(out_dir, "r%d.pt" % rank))
    ddp.barrier()
    ddp.close()


@pytest.mark.parametrize("case,capture", [("mean", "cmdlist"), ("max_pool", "cmdlist"), ("attention", "cmdlist"),
                                          ("attention_emb", "cmdlist"), ("attention_emb_mae", "cmdlist"),
                                          ("attention_emb_bf16", "cmdlist"), ("mean_emb", "cmdlist"),
                                          ("mean", "cmdlist+overlap"), ("max_pool", "cmdlist+overlap"),
                                          ("mean", "graph"), ("mean", False), ("attention_emb_mae", False)])
def test_two_rank_engine_equals_single_process_global_batch(tmp_path, case, capture):
    gs = pkg()
    model, feats, loss_fn, ids, tg, prec = util.dp_case(gs, case, n_batch=N_BATCH, global_batch=2 * B_RANK)
    ref_preds, ref_w, _eng = util.dp_run(gs, case, model, feats, loss_fn, ids, tg, prec, None, steps=STEPS)
    del _eng
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), case, capture), nprocs=2, join=True)
    r0 = torch.load(os.path.join(str(tmp_path), "r0.pt"))
    r1 = torch.load(os.path.join(str(tmp_path), "r1.pt"))
    assert r0["one_list"] == (capture in ("cmdlist", "cmdlist+overlap"))
    assert torch.equal(r0["w"], r1["w"]), "replicas diverged: %g" % float((r0["w"] - r1["w"]).abs().max())
    got = torch.cat([r0["preds"], r1["preds"]], dim=1)
    tol = 2e-5 if prec == "fp32" else 3e-3
    scale = float(ref_preds.abs().max())
    assert float((got - ref_preds).abs().max()) <= tol * scale, float((got - ref_preds).abs().max())
    assert float((r0["w"] - ref_w).abs().max()) <= tol * float(ref_w.abs().max()), float((r0["w"] - ref_w).abs().max())
