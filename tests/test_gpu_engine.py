"""-m gpu: the hipGraph engines (engine.py) against the eager autograd path and the oracle.
Both engines draw the same Philox samples as the eager product path for the same seed, so
predictions / gradients / updated weights are directly comparable."""
import copy

import numpy as np
import pytest
import torch
from scipy import sparse
from torch.nn import functional as F

from conftest import pkg
from util import close, close_fro

pytestmark = pytest.mark.gpu
gs = pkg()
ops = gs.ops
DEV = "cuda"


def _problem(n=600, D=40, C=5, seed=0, max_deg=30):
    rng = np.random.RandomState(seed)
    deg = rng.randint(0, max_deg, size=n + 1)
    deg[0], deg[7], deg[n] = 0, 0, 3
    indptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    data = rng.randint(1, n + 1, size=int(indptr[-1]))
    adj = sparse.csr_matrix((data, gs.store.row_positions(indptr), indptr), shape=(n + 1, int(deg.max())))
    feats = rng.normal(size=(n + 1, D)).astype(np.float32)
    feats[0] = 0
    return adj, feats, rng


def _model(adj, D, C, dims, fans, seed=3, agg="mean"):
    torch.manual_seed(seed)
    gs.nn_modules.SparseUniformNeighborSampler.rng_default = "philox"
    specs = [{"n_train_samples": f, "n_val_samples": f, "output_dim": h,
              "activation": (lambda x: x) if i == len(dims) - 1 else F.relu}
             for i, (h, f) in enumerate(zip(dims, fans))]
    m = gs.GSSupervised(sampler_class=gs.sampler_lookup["sparse_uniform_neighbor_sampler"], adj=adj,
                        train_adj=adj, prep_class=gs.prep_lookup["identity"],
                        aggregator_class=gs.aggregator_lookup[agg], input_dim=D,
                        n_nodes=adj.shape[0], n_classes=C, layer_specs=specs, lr_init=0.01,
                        weight_decay=1e-4)
    gs.nn_modules.SparseUniformNeighborSampler.rng_default = "compat"
    m.train_sampler.seed = m.val_sampler.seed = 77
    return m.to(DEV)


@pytest.fixture(autouse=True)
def _setup():
    ops.set_compute_dtype("bf16")
    ops.warmup(torch.device(DEV))
    yield
    ops.set_compute_dtype("bf16")


@pytest.mark.parametrize("dims,fans,B", [((128, 128), (25, 10), 64), ((16, 8), (5, 3), 33),
                                         ((32, 16, 8), (4, 3, 2), 20)])
@pytest.mark.parametrize("capture", [False, "cmdlist", "graph"])
def test_fused_engine_matches_eager_autograd_path(dims, fans, B, capture):
    adj, feats, rng = _problem()
    D, C = feats.shape[1], 5
    store = gs.FeatureStore.from_array(feats, torch.device(DEV), dtype="bf16")
    ref_model = _model(adj, D, C, dims, fans)
    eng_model = _model(adj, D, C, dims, fans)
    eng_model.load_state_dict(ref_model.state_dict())
    ref_model.optimizer = torch.optim.Adam(ref_model.parameters(), lr=0.01, weight_decay=1e-4)
    loss_fn = gs.ProblemLosses.classification
    assert gs.engine.FusedMeanTrainStep.supports(eng_model, store)
    batches = [(torch.from_numpy(rng.randint(1, adj.shape[0], size=B)).to(DEV),
                torch.from_numpy(rng.randint(0, C, size=(B, 1))).to(DEV)) for _ in range(3)]
    eng = gs.engine.FusedMeanTrainStep(eng_model, store, loss_fn, batches[0][0], batches[0][1], capture=capture)
    for step, (ids, tg) in enumerate(batches):
        # every step is checked from identical weights (Adam's sign-like first updates would
        # otherwise amplify bf16 round-off into O(lr) weight differences)
        ref_model.load_state_dict(eng_model.state_dict())
        p_ref = ref_model.train_step(ids=ids, feats=store, targets=tg, loss_fn=loss_fn).detach().float().cpu().numpy()
        p_eng = eng(ids, tg).detach().float().cpu().numpy()
        # same samples, same forward kernels: predictions agree to bf16 round-off of the weights
        close(p_eng, p_ref, ("preds", step), 3e-2, 3e-2)
        for (k, a), (_, b) in zip(eng_model.named_parameters(), ref_model.named_parameters()):
            close_fro(a.grad.cpu().numpy(), b.grad.cpu().numpy(), ("grad", step, k), 0.1)
            if step == 0:
                close_fro(a.detach().cpu().numpy(), b.detach().cpu().numpy(), ("weight", k), 0.05)
    # the Parameters are views of the flat bucket: state_dict stays usable, eval path still works
    ev = eng_model(batches[0][0], store, train=False)
    assert ev.shape == (B, C) and torch.isfinite(ev).all()
    eng_model.train_sampler.csr(DEV).check()


@pytest.mark.parametrize("dims,fans,B", [((128, 128), (25, 10), 64), ((64, 64), (5, 3), 33), ((64,), (7,), 20),
                                         ((64, 128, 64), (4, 3, 2), 20)])
@pytest.mark.parametrize("capture", [False, "cmdlist"])
@pytest.mark.parametrize("agg", ["max_pool", "mean_pool"])
def test_fused_pool_engine_matches_eager_autograd_path(dims, fans, B, capture, agg):
    """FusedPoolTrainStep (max-pool aggregators, BASELINE config 3) against GSSupervised.train_step on
    the eager product path: same Philox samples, same K3 forward kernel; the engine's own backward
    (argmax routing, K5 input gradients, grouped K5b) against autograd."""
    adj, feats, rng = _problem()
    D, C = feats.shape[1], 5
    store = gs.FeatureStore.from_array(feats, torch.device(DEV), dtype="bf16")
    ref_model = _model(adj, D, C, dims, fans, agg=agg)
    eng_model = _model(adj, D, C, dims, fans, agg=agg)
    eng_model.load_state_dict(ref_model.state_dict())
    ref_model.optimizer = torch.optim.Adam(ref_model.parameters(), lr=0.01, weight_decay=1e-4)
    loss_fn = gs.ProblemLosses.classification
    assert gs.engine.fused_engine_for(eng_model, store) is gs.engine.FusedPoolTrainStep
    batches = [(torch.from_numpy(rng.randint(1, adj.shape[0], size=B)).to(DEV),
                torch.from_numpy(rng.randint(0, C, size=(B, 1))).to(DEV)) for _ in range(3)]
    eng = gs.engine.FusedPoolTrainStep(eng_model, store, loss_fn, batches[0][0], batches[0][1], capture=capture)
    for step, (ids, tg) in enumerate(batches):
        ref_model.load_state_dict(eng_model.state_dict())
        p_ref = ref_model.train_step(ids=ids, feats=store, targets=tg, loss_fn=loss_fn).detach().float().cpu().numpy()
        p_eng = eng(ids, tg).detach().float().cpu().numpy()
        close(p_eng, p_ref, ("preds", step), 3e-2, 3e-2)
        for (k, a), (_, b) in zip(eng_model.named_parameters(), ref_model.named_parameters()):
            close_fro(a.grad.cpu().numpy(), b.grad.cpu().numpy(), ("grad", step, k), 0.1)
            if step == 0:
                close_fro(a.detach().cpu().numpy(), b.detach().cpu().numpy(), ("weight", k), 0.05)
    ev = eng_model(batches[0][0], store, train=False)
    assert ev.shape == (B, C) and torch.isfinite(ev).all()
    eng_model.train_sampler.csr(DEV).check()
    # queue mode (next batch sampled / gathered one step ahead) == per-step copies, bit for bit
    ids_all = torch.stack([b[0] for b in batches])
    tg_all = torch.stack([b[1] for b in batches])
    outs = []
    for queued in (False, True):
        mdl = _model(adj, D, C, dims, fans, agg=agg)
        e2 = gs.engine.FusedPoolTrainStep(mdl, store, loss_fn, batches[0][0], batches[0][1], capture=capture)
        if queued:
            e2.load_epoch(ids_all, tg_all)
            preds = [e2.step_queue().clone() for _ in range(4)]
        else:
            preds = [e2(ids_all[k % 3], tg_all[k % 3]).clone() for k in range(4)]
        outs.append((torch.stack(preds), e2.flat_p.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("capture", [False, "cmdlist", "graph"])
def test_pipelined_engine_is_bit_identical_to_sequential(capture):
    """Overlapping batch k+1's sampling/gathers with batch k's compute must not change anything:
    sampling does not depend on the weights and every kernel is deterministic."""
    adj, feats, rng = _problem(seed=2)
    D, C, B, dims, fans = feats.shape[1], 5, 40, (128, 128), (6, 4)
    store = gs.FeatureStore.from_array(feats, torch.device(DEV), dtype="bf16")
    batches = [(torch.from_numpy(rng.randint(1, adj.shape[0], size=B)).to(DEV),
                torch.from_numpy(rng.randint(0, C, size=(B, 1))).to(DEV)) for _ in range(5)]
    outs = {}
    for mode in (False, True):
        model = _model(adj, D, C, dims, fans)
        eng = gs.engine.FusedMeanTrainStep(model, store, gs.ProblemLosses.classification, batches[0][0],
                                           batches[0][1], capture=capture, pipelined=mode)
        preds = []
        for ids, tg in batches:
            r = eng(ids, tg)
            if r is not None:
                preds.append(r.clone())
        if mode:
            preds.append(eng.flush().clone())
        torch.cuda.synchronize()
        outs[mode] = (torch.stack(preds), eng.flat_p.clone(), int(eng.counter.item()), int(eng.step.item()))
    assert outs[False][2] == outs[True][2] == 10 and outs[False][3] == outs[True][3] == 5
    assert torch.equal(outs[False][0], outs[True][0])
    assert torch.equal(outs[False][1], outs[True][1])


@pytest.mark.parametrize("B,fans", [(24, (5, 3)), (40, (25, 10)), (800, (25, 10)), (40, (6, 5)), (40, (7, 15)),
                                    (24, (4, 3, 5))])
@pytest.mark.parametrize("capture", [False, "cmdlist", "graph"])
def test_batch_queue_equals_per_step_copies(capture, B, fans):
    """step_queue (batch i+2 sampled and batch i+1 gathered in the launch that applies Adam(i)) is
    the plain one-batch-at-a-time engine, bit for bit, across the queue's wrap-around.  Fan-out
    (25, 10) also moves hop-2 rows of the next batch into the seed-level launch: all of them at
    B = 40 (the gather launch's segment disappears), about a tenth at B = 800 (56 idle CUs on an MI355X)."""
    adj, feats, rng = _problem(seed=4)
    D, C, dims = feats.shape[1], 5, (128,) * len(fans)
    store = gs.FeatureStore.from_array(feats, torch.device(DEV), dtype="bf16")
    ids_all = torch.from_numpy(rng.randint(1, adj.shape[0], size=(3, B))).to(DEV)
    tg_all = torch.from_numpy(rng.randint(0, C, size=(3, B, 1))).to(DEV)
    res = []
    for queued in (False, True):
        model = _model(adj, D, C, dims, fans)
        eng = gs.engine.FusedMeanTrainStep(model, store, gs.ProblemLosses.classification, ids_all[0], tg_all[0],
                                           capture=capture)
        preds = []
        if queued:
            eng.load_epoch(ids_all, tg_all)
            for k in range(7):                       # wraps around the 3-batch queue
                preds.append(eng.step_queue().clone())
        else:
            for k in range(7):
                preds.append(eng(ids_all[k % 3], tg_all[k % 3]).clone())
        res.append((torch.stack(preds), eng.flat_p.clone()))
        if queued and fans[-1] in (5, 10, 15):
            assert eng._tail_rows > 0              # the seed-level launch carried part of the last hop's means
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("B,fans", [(24, (5, 3)), (200, (25, 10))])
@pytest.mark.parametrize("capture", [False, "cmdlist", "graph"])
def test_data_parallel_queue_order_matches_single_process(capture, B, fans):
    """dist.py + engine.step_queue with a ONE-rank RCCL group: the exchange is an identity, so the
    software-pipelined order (next batch's sample/gather issued while the all-reduce is in flight,
    Adam afterwards) must reproduce the plain sequential engine.  The clip norm comes from a
    different kernel in this mode (k_grad_sqnorm), hence a tolerance instead of torch.equal."""
    import os
    import torch.distributed as dist
    adj, feats, rng = _problem(seed=6)
    D, C, dims = feats.shape[1], 5, (128, 128)     # fans (25, 10): the seed-level launch also gathers
    store = gs.FeatureStore.from_array(feats, torch.device(DEV), dtype="bf16")
    ids_all = torch.from_numpy(rng.randint(1, adj.shape[0], size=(3, B))).to(DEV)
    tg_all = torch.from_numpy(rng.randint(0, C, size=(3, B, 1))).to(DEV)

    def run(ddp):
        model = _model(adj, D, C, dims, fans)
        if ddp is not None:
            gs.dist.attach(model, ddp, seed=77)
        eng = gs.engine.FusedMeanTrainStep(model, store, gs.ProblemLosses.classification, ids_all[0],
                                           tg_all[0], ddp=ddp, capture=capture)
        eng.load_epoch(ids_all, tg_all)
        preds = [eng.step_queue().clone() for _ in range(5)]
        torch.cuda.synchronize()
        return torch.stack(preds), eng.flat_p.clone()

    ref = run(None)
    env = {"GSAGE_FORCE_DDP": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533", "RANK": "0",
           "WORLD_SIZE": "1", "LOCAL_RANK": "0"}
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    ddp = None
    try:
        ddp = gs.dist.init_from_env(cuda=True)
        assert ddp is not None and ddp.world == 1
        got = run(ddp)
    finally:
        if ddp is not None:
            ddp.close()
        if dist.is_initialized():
            dist.destroy_process_group()
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    close(got[0].cpu().numpy(), ref[0].cpu().numpy(), "preds of 5 queue steps", rtol=2e-3, atol=2e-3)
    close(got[1].cpu().numpy(), ref[1].cpu().numpy(), "weights after 5 queue steps", rtol=2e-3, atol=2e-3)


def test_fused_engine_first_step_against_oracle():
    """One engine step vs the fp32 CPU oracle fed the same Philox sel (bf16 tolerance)."""
    from oracle import cpu as ocpu
    from oracle import torch_ref as tref
    adj, feats, rng = _problem(seed=5)
    D, C, B, fans, dims = feats.shape[1], 5, 48, (6, 4), (128, 128)
    store = gs.FeatureStore.from_array(feats, torch.device(DEV), dtype="bf16")
    model = _model(adj, D, C, dims, fans)
    w0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    ids = rng.randint(1, adj.shape[0], size=B)
    tg = rng.randint(0, C, size=(B, 1))
    eng = gs.engine.FusedMeanTrainStep(model, store, gs.ProblemLosses.classification,
                                       torch.from_numpy(ids).to(DEV), torch.from_numpy(tg).to(DEV))
    preds = eng(torch.from_numpy(ids).to(DEV), torch.from_numpy(tg).to(DEV)).cpu().numpy()
    sels = [ocpu.philox_sel(77, 0, 0, B * fans[0], adj.shape[1]).reshape(B, fans[0]),
            ocpu.philox_sel(77, 1, 0, B * fans[0] * fans[1], adj.shape[1]).reshape(-1, fans[1])]
    fb = store.dense().cpu()                                  # the bf16-rounded table, as fp32
    ref = tref.train_step(w0, tref.Adam(weight_decay=1e-4), 0.01, "classification", ids, fb,
                          torch.from_numpy(tg), adj.indptr.astype(np.int64), adj.data.astype(np.int64),
                          fans, sels, "mean", "identity", adj.shape[0])
    close(preds, ref["preds"].numpy(), "preds vs oracle", 3e-2, 3e-2)
    assert abs(float(eng.gnorm.item()) - ref["gradnorm"]) < 0.05 * max(1.0, ref["gradnorm"])
    for k, v in model.named_parameters():
        close_fro(v.grad.cpu().numpy(), ref["clipped"][k].numpy(), ("grad vs oracle", k), 0.1)


def test_captured_autograd_engine_runs_and_advances_samples():
    adj, feats, rng = _problem(seed=9)
    D, C, B = feats.shape[1], 5, 32
    store = gs.FeatureStore.from_array(feats, torch.device(DEV), dtype="bf16")
    model = _model(adj, D, C, (16, 16), (5, 3))
    ids = torch.from_numpy(rng.randint(1, adj.shape[0], size=B)).to(DEV)
    tg = torch.from_numpy(rng.randint(0, C, size=(B, 1))).to(DEV)
    eng = gs.engine.CapturedTrainStep(model, store, gs.ProblemLosses.classification, ids, tg)
    a = eng(ids, tg).clone()
    b = eng(ids, tg).clone()
    assert torch.isfinite(a).all() and not torch.equal(a, b)     # new samples + updated weights
    assert int(eng.counter.item()) == 4


def test_engine_hands_its_adam_state_back_to_train_step_and_checkpoints():
    """An engine trains three steps, then GSSupervised.train_step takes the Parameters back: FlatAdam must continue
    from the engine's exp_avg / exp_avg_sq / step count (the moments used to restart from zero), and
    `model.optimizer_state_dict()` while the engine holds the Parameters is torch.optim.Adam's format with that state."""
    adj, feats, rng = _problem()
    D, C, B = feats.shape[1], 5, 32
    store = gs.FeatureStore.from_array(feats, torch.device(DEV), dtype="bf16")
    model = _model(adj, D, C, (16, 8), (5, 3))
    loss_fn = gs.ProblemLosses.classification
    batches = [(torch.from_numpy(rng.randint(1, adj.shape[0], size=B)).to(DEV),
                torch.from_numpy(rng.randint(0, C, size=(B, 1))).to(DEV)) for _ in range(4)]
    eng = gs.engine.FusedMeanTrainStep(model, store, loss_fn, batches[0][0], batches[0][1], capture="cmdlist")
    for ids, tg in batches[:3]:
        eng(ids, tg)
    torch.cuda.synchronize()
    assert eng.holds_parameters()
    sd = model.optimizer_state_dict()
    assert int(sd["state"][0]["step"]) == 3 and set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    ref = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in model.parameters()], lr=0.01)
    ref.load_state_dict(sd)                                    # the reference's optimizer class accepts it
    m3, v3 = eng.flat_m.clone(), eng.flat_v.clone()
    assert float(m3.abs().max()) > 0
    model.train_step(ids=batches[3][0], feats=store, targets=batches[3][1], loss_fn=loss_fn)
    torch.cuda.synchronize()
    opt = model.optimizer
    assert isinstance(opt, gs.optim.FlatAdam) and not eng.holds_parameters()
    assert int(opt.step_count.item()) == 4
    g = opt.flat_g                                             # m4 = 0.9 m3 + 0.1 g4 (weight decay folded into g)
    wd_g = g + 1e-4 * (opt.flat_p - 0)                         # p moved by <= lr; bound loosely below
    assert torch.allclose(opt.flat_m, 0.9 * m3 + 0.1 * wd_g, rtol=0, atol=2e-4 * float(1 + wd_g.abs().max()))
    assert float((opt.flat_v - 0.999 * v3).min()) >= -1e-12    # v4 = 0.999 v3 + 0.001 g^2 >= 0.999 v3
    import pickle
    pickle.dumps(model.state_dict())


