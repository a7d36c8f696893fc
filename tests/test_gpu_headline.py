"""-m gpu: the HEADLINE configurations checked against the oracle at the size they are benched (round-3 verdict,
"what's weak" #1): BASELINE configs[1] (mean) and configs[2] (max-pool) on bench.py's Reddit-shaped graph --
N = 232 965, D = 602 (rows padded to 640), B = 512, fan-out 25/10, hidden 128 -- in the launch mode the bench times
(command lists, device-resident batch queue, software-pipelined steps: the `k_mean_tail_ce<16,10>` seed level, the
gather roles, Adam riding in the gather launch), bf16 storage against the rounding-aware oracle and fp32 storage
against the plain fp32 oracle.  The oracle runs on the frontier's rows relabelled to a compact table (the technique
of test_gpu_large.py): two consecutive train steps, predictions / gradient norm / weights after each."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import pkg
from util import close, close_fro, close_update

_LOG = os.environ.get("GSAGE_PARITY_LOG")


def _note(key, **vals):
    """measured errors -> profiles/rNN_parity_errors.jsonl (tools/gpu_round.sh sets GSAGE_PARITY_LOG)"""
    if _LOG:
        with open(_LOG, "a") as f:
            f.write(json.dumps(dict(key=key, **{k: float(v) for k, v in vals.items()})) + "\n")

pytestmark = pytest.mark.gpu
gs = pkg()
ops, nat = gs.ops, gs._native
DEV = "cuda"
B, FAN, SEED = 512, (25, 10), 123


@pytest.fixture(scope="module")
def reddit():
    """the bench's Reddit-shaped synthetic graph + features (built once: ~25 s of host time)"""
    import bench
    return bench.synthetic_reddit(seed=0)


@pytest.fixture(autouse=True)
def _setup():
    ops.set_compute_dtype("bf16")
    ops.warmup(torch.device(DEV))
    yield
    ops.set_compute_dtype("bf16")
    gs.nn_modules.SparseUniformNeighborSampler.rng_default = "compat"


# relative Frobenius bound on the two-step Adam update, bf16 engines against the rounding-aware oracle.  Max pool: an
# argmax that flips under another summation order moves a whole route; attention: tiny attention-MLP gradients whose
# sign flips move an Adam update by a full lr (att.0.weight; predictions and gradient norm agree to 2e-6 in the same
# run).  Measured (profiles/r06_parity_errors.jsonl, headline/*/update): mean 3.2e-3, max pool 3.9e-2, attention 6.3e-2
# -- the bounds are 1.3 - 2 x those (round-5 review: "2 x the measured error"; max pool's 5e-2 already sits below that).
UPDATE_BOUND = {"mean": 5e-3, "max_pool": 5e-2, "attention": 0.13}


def _frontier(csr, seeds, batch, L=2):
    """the queue pipeline's frontier of batch `batch`: hop k is Philox call batch * L + k of the sampler's seed"""
    cur, hops = seeds, []
    for k, f in enumerate(FAN):
        cur = ops.sample_csr(csr, cur, f, philox={"seed": SEED, "call_base": batch * L + k})
        hops.append(cur)
    return hops


@pytest.mark.parametrize("agg,prec", [("mean", "bf16"), ("mean", "fp32"), ("max_pool", "bf16"), ("attention", "bf16")])
def test_bench_workload_steps_equal_the_oracle(reddit, agg, prec):
    import bench
    from oracle import torch_ref as tref
    data, dev = reddit, torch.device(DEV)
    ops.set_compute_dtype(prec)
    store = data["feats"](dev, prec)
    assert store.dim == 602 and store.ld == (640 if prec == "bf16" else 608) and store.data.shape[0] == 232966
    n_steps = 2
    rng = np.random.RandomState(17)
    pick = rng.randint(0, len(data["train_ids"]), size=(n_steps + 2, B))
    ids = torch.from_numpy(data["train_ids"][pick]).to(dev)
    tg = torch.from_numpy(data["targets"][data["train_ids"][pick]]).to(dev).view(n_steps + 2, B, 1)
    torch.manual_seed(11)
    model = bench.build_model(gs, data["adj"], aggregator=agg, rng="philox", seed=SEED).to(dev)
    w0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    csr = model.train_sampler.csr(dev)
    eng = gs.engine.fused_engine_for(model, store)(model, store, gs.ProblemLosses.classification, ids[0], tg[0],
                                                   capture="cmdlist")
    assert eng.capture_mode == "cmdlist" and eng.fused_head
    if agg == "attention":
        assert eng.fuse[0] and not eng.fuse[1]            # the last hop (fan-out 10) through K4 / K4' with the MLP inside
    if agg == "mean":
        assert eng.fused_tail and eng.B == B and eng.fan[1:] == [25, 10]        # the seed-level kernel (k_mean_tail_mfma<25, 10> at B = 512, bf16)
    eng.load_epoch(ids, tg)
    preds, norms = [], []
    before = nat.launch_count()
    for s in range(n_steps):
        preds.append(eng.step_queue().detach().float().cpu().numpy().copy())
        norms.append(float(eng.gnorm.item()))
        if s == 0:
            # the pipeline's buffers: batch 1 sampled by the prime launch, batch 2 inside step 0 (its projection's launch)
            torch.cuda.synchronize()
            for b, buf in ((1, eng.ids_q[1]), (2, eng.ids_q[2 % eng.P])):     # (a ring of eng.P frontier buffers)
                hops = _frontier(csr, ids[b], b)
                assert torch.equal(buf[B:B + B * 25], hops[0]) and torch.equal(buf[B + B * 25:], hops[1]), b
    torch.cuda.synchronize()
    assert nat.launch_count() - before >= 4 * n_steps         # (mean: K5 | seed level | K5b | gather + Adam + K1)
    csr.check()
    if agg == "mean" and prec == "bf16":
        assert eng._tail_rows > 0                         # the seed-level launch's gather role is part of the step

    # the oracle on the frontier's rows, relabelled to a compact table, step after step from the same weights
    w = {k: v.clone() for k, v in w0.items()}
    opt = tref.Adam()
    bf = prec == "bf16"
    for s in range(n_steps):
        hops = [h.cpu().numpy() for h in _frontier(csr, ids[s], s)]
        uniq, inv = np.unique(np.concatenate([ids[s].cpu().numpy()] + hops), return_inverse=True)
        small = store.data[torch.from_numpy(uniq).to(dev), :store.dim].float().cpu()
        parts = np.split(inv, np.cumsum([B, hops[0].shape[0]]))
        r = tref.train_step(w, opt, 0.01, "classification", parts[0], small, tg[s].cpu(), None, None, FAN, None, agg,
                            "identity", 232966, rounding="bf16" if bf else None, frontier=parts[1:])
        _note("headline/%s/%s/step%d" % (agg, prec, s), preds=np.abs(preds[s] - r["preds"].numpy()).max(),
              gnorm_rel=abs(norms[s] - r["gradnorm"]) / max(1.0, r["gradnorm"]))
        close(preds[s], r["preds"].numpy(), "preds vs oracle, step %d (%s %s)" % (s, agg, prec), *(((1e-3, 1e-3) if agg == "mean" else (3e-3, 3e-3)) if bf else (2e-4, 2e-4)))
        assert abs(norms[s] - r["gradnorm"]) <= ((1e-3 if agg == "mean" else 5e-3) if bf else 2e-4) * max(1.0, r["gradnorm"]), \
            (s, norms[s], r["gradnorm"])
    worst = 0.0
    for k, v in model.named_parameters():
        if bf:
            got, want = v.detach().cpu().numpy() - w0[k].numpy(), w[k].numpy() - w0[k].numpy()
            worst = max(worst, float(np.linalg.norm(got - want)) / max(float(np.linalg.norm(want)), 1e-12))
            close_fro(got, want, ("Adam updates", k), UPDATE_BOUND[agg])
        else:
            close_update(v.detach().cpu().numpy(), w[k].numpy(), w0[k].numpy(), ("weights after 2 steps", k))
    if bf:
        _note("headline/%s/%s/update" % (agg, prec), upd_fro=worst)
