"""-m gpu: the reference-generated golden fixtures replayed through the FUSED engines -- the path
bench.py times (engine.FusedMeanTrainStep / FusedPoolTrainStep), not the module-level path.

Two instantiations of the same engine and kernel sources:
  * fp32 storage ("parity mode"): every kernel of the step runs on fp32 operands (the seed-level
    kernel, the multi-segment gather and K5 are the same templates instantiated on float; K5b has an
    fp32 twin with the same slab layout), the sampler replays the `sel` the reference drew
    (gsage_hops_desc.sel).  Predictions, pre-clip gradient norm, clipped gradients and the weights
    after two Adam steps are compared with tests/golden/engine_kat.npz -- outputs of the reference's
    own train_step (models.py:97-104) -- at fp32 tolerance.
  * bf16 storage (production): compared with the oracle restated with the engines' bf16 rounding
    points (oracle/torch_ref.py rounding="bf16"), same `sel`: bounds of a few 1e-3 instead of the few
    1e-2 an fp32 oracle allows for a bf16 path (measured: 6e-8 / 7e-6; the bounds below are 1e-4 / 1e-3).
Each check fails if Adam, the clip, or one gradient term is skipped (see test_*_detects_*)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, pkg
from util import build_model, close, close_fro, close_rel, close_update, weights

pytestmark = pytest.mark.gpu
gs = pkg()
ops = gs.ops
DEV = "cuda"
N_CASES = 6
_LOG = os.environ.get("GSAGE_PARITY_LOG")


def _note(key, **vals):
    if _LOG:
        with open(_LOG, "a") as f:
            f.write(json.dumps(dict(key=key, **{k: float(v) for k, v in vals.items()})) + "\n")


@pytest.fixture(autouse=True)
def _setup():
    ops.set_compute_dtype("bf16")
    ops.warmup(torch.device(DEV))
    gs.nn_modules.SparseUniformNeighborSampler.rng_default = "philox"
    yield
    gs.nn_modules.SparseUniformNeighborSampler.rng_default = "compat"
    ops.set_compute_dtype("bf16")


def _case(g, c, dtype):
    p = "e%d_" % c
    model, store, task = build_model(gs, g, p, device=DEV, feats_dtype=dtype)
    fan = [int(v) for v in g[p + "fanouts"]]
    ids = torch.from_numpy(g[p + "ids"]).to(DEV)
    tg = torch.from_numpy(g[p + "targets"]).to(DEV)
    sels = [[g[p + "s%d_sel%d" % (st, h)] for h in range(len(fan))] for st in range(2)]
    return p, model, store, fan, ids, tg, sels


def _engine(model, store, ids, tg, capture):
    cls = gs.engine.fused_engine_for(model, store)
    assert cls is not None, "fixture case not covered by a fused engine"
    return cls(model, store, gs.ProblemLosses.classification, ids, tg, capture=capture)


@pytest.mark.parametrize("capture", [False, "cmdlist"])
@pytest.mark.parametrize("c", range(N_CASES))
def test_fp32_engine_replays_reference_train_steps(c, capture):
    g = load_golden("engine_kat.npz")
    p, model, store, fan, ids, tg, sels = _case(g, c, "fp32")
    w0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    eng = _engine(model, store, ids, tg, capture)
    before = gs._native.launch_count()
    for step in range(2):
        eng.set_progress(0.25 * step)
        assert abs(float(eng.lr.item()) - float(g[p + "lr%d" % step])) < 1e-9
        eng.set_sel(sels[step])
        preds = eng(ids, tg).detach().cpu().numpy()
        close(preds, g[p + "s%d_preds" % step], (c, step, "preds"), 2e-4, 2e-5)
        gn, gn_ref = float(eng.gnorm.item()), float(g[p + "s%d_gradnorm" % step])
        assert abs(gn - gn_ref) <= 2e-4 * max(1.0, gn_ref), (c, step, "gradnorm", gn, gn_ref)
        _note("fp32/%d/%s/%d" % (c, capture, step), preds=np.abs(preds - g[p + "s%d_preds" % step]).max(),
              gnorm=abs(gn - gn_ref))
        if step == 0:
            for k, v in model.named_parameters():          # p.grad holds the CLIPPED gradient, like the reference
                close_rel(v.grad.cpu().numpy(), g[p + "s0_cg_" + k], (c, "clipped grad", k), 2e-4)
    assert gs._native.launch_count() > before
    err = 0.0
    for k, v in model.state_dict().items():
        ref = g[p + "w2_" + k]
        err = max(err, float(np.abs(v.detach().cpu().numpy() - ref).max()))
        close_update(v.detach().cpu().numpy(), ref, w0[k].cpu().numpy(), (c, "weights after 2 steps", k))
    _note("fp32/%d/%s/w2" % (c, capture), werr=err)
    model.train_sampler.csr(DEV).check()


@pytest.mark.parametrize("c", [0, 1, 3, 4])
def test_fp32_engine_queue_mode_replays_reference(c):
    """The software-pipelined queue order bench.py uses (batch i+2 sampled and batch i+1 gathered inside
    the launch that applies Adam(i)) with the recorded draws as a device-resident sel queue."""
    g = load_golden("engine_kat.npz")
    p, model, store, fan, ids, tg, sels = _case(g, c, "fp32")
    w0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    eng = _engine(model, store, ids, tg, "cmdlist")
    sel_q = torch.stack([torch.cat([torch.from_numpy(np.asarray(x)).reshape(-1) for x in sels[st]]) for st in range(2)])
    eng.load_epoch(torch.stack([ids, ids]), torch.stack([tg, tg]), sel_epoch=sel_q)
    for step in range(2):
        eng.set_progress(0.25 * step)
        preds = eng.step_queue().detach().cpu().numpy()
        close(preds, g[p + "s%d_preds" % step], (c, step, "preds"), 2e-4, 2e-5)
    torch.cuda.synchronize()
    for k, v in model.state_dict().items():
        close_update(v.detach().cpu().numpy(), g[p + "w2_" + k], w0[k].numpy(), (c, "weights after 2 queue steps", k))


def _oracle_bf16(g, p, store, fan, ids, tg, sels, steps=2):
    from oracle import torch_ref as tref
    aggn = str(g[p + "cfg"][0])
    w = weights(g, p + "w0_")
    opt = tref.Adam(weight_decay=float(g[p + "weight_decay"]))
    fb = store.dense().cpu()                         # the bf16-rounded table, as fp32
    out = []
    for st in range(steps):
        r = tref.train_step(w, opt, float(g[p + "lr%d" % st]), "classification", ids.cpu().numpy(), fb, tg.cpu(),
                            g[p + "tadj_indptr"], g[p + "tadj_data"], fan,
                            [np.asarray(x).astype(np.int64) for x in sels[st]], aggn, "identity",
                            int(g[p + "adj_shape"][0]), rounding="bf16")
        out.append(r)
    return out, w


@pytest.mark.parametrize("mode", ["call", "queue"])
@pytest.mark.parametrize("c", range(N_CASES))
def test_bf16_engine_against_rounding_aware_oracle(c, mode):
    """The PRODUCTION instantiation (bf16 storage, packed-weight K5, MFMA K5b, seed-level kernel) with
    the reference's recorded draws, against the oracle with the engines' bf16 rounding points."""
    g = load_golden("engine_kat.npz")
    p, model, store, fan, ids, tg, sels = _case(g, c, "bf16")
    w0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    ref, w_ref = _oracle_bf16(g, p, store, fan, ids, tg, sels)
    eng = _engine(model, store, ids, tg, "cmdlist")
    if mode == "queue":
        sel_q = torch.stack([torch.cat([torch.from_numpy(np.asarray(x)).reshape(-1) for x in sels[st]])
                             for st in range(2)])
        eng.load_epoch(torch.stack([ids, ids]), torch.stack([tg, tg]), sel_epoch=sel_q)
    for step in range(2):
        eng.set_progress(0.25 * step)
        if mode == "queue":
            preds = eng.step_queue().detach().cpu().numpy()
        else:
            eng.set_sel(sels[step])
            preds = eng(ids, tg).detach().cpu().numpy()
        torch.cuda.synchronize()
        r = ref[step]
        perr = float(np.abs(preds - r["preds"].numpy()).max())
        close(preds, r["preds"].numpy(), (c, step, "preds vs bf16-aware oracle"), 1e-4, 1e-4)
        # and against the reference's own fp32 outputs at the looser bound bf16 storage allows
        close(preds, g[p + "s%d_preds" % step], (c, step, "preds vs reference"), 3e-2, 3e-2)
        gn = float(eng.gnorm.item())
        assert abs(gn - r["gradnorm"]) <= 1e-4 * max(1.0, r["gradnorm"]), (c, step, gn, r["gradnorm"])
        gerr = 0.0
        if step == 0 or mode == "call":
            for k, v in model.named_parameters():
                a, b = v.grad.cpu().numpy(), r["clipped"][k].numpy()
                e = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-12))
                gerr = max(gerr, e)
                close_fro(a, b, (c, step, "clipped grad", k), 1e-3)
        _note("bf16/%d/%s/%d" % (c, mode, step), preds=perr, gnorm=abs(gn - r["gradnorm"]), grad_fro=gerr)
    # two Adam steps: compare the UPDATE (weights minus initial weights), which a skipped / wrong
    # optimizer step or a sign error in any gradient term changes by O(1)
    worst = 0.0
    for k, v in model.state_dict().items():
        d_eng = v.detach().cpu().numpy() - w0[k].numpy()
        d_ref = w_ref[k].numpy() - w0[k].numpy()
        e = float(np.linalg.norm(d_eng - d_ref) / max(np.linalg.norm(d_ref), 1e-12))
        worst = max(worst, e)
        close_fro(d_eng, d_ref, (c, "weight update", k), 1e-2)
    _note("bf16/%d/%s/w2" % (c, mode), upd_fro=worst)


def test_checks_detect_a_skipped_update_and_a_dropped_gradient_term():
    """The tolerances above are not vacuous: (a) weights that miss the second Adam step, (b) an update
    computed without the neighbour half of the level-0 gradient both fail the same comparisons."""
    g = load_golden("engine_kat.npz")
    p, model, store, fan, ids, tg, sels = _case(g, 1, "fp32")
    w0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    eng = _engine(model, store, ids, tg, False)
    eng.set_progress(0.0)
    eng.set_sel(sels[0])
    eng(ids, tg)
    torch.cuda.synchronize()
    # (a) one step instead of two
    k = "agg_layers.0.fc_x.weight"
    d_ref = g[p + "w2_" + k] - w0[k].numpy()
    d_one = model.state_dict()[k].detach().cpu().numpy() - w0[k].numpy()
    with pytest.raises(AssertionError):
        close_fro(d_one, d_ref, "one step is not two", 5e-3)
    # (b) the clipped gradient with one term zeroed
    kk = "agg_layers.0.fc_neib.weight"
    bad = dict(model.named_parameters())[kk].grad.cpu().numpy() * 0.0
    with pytest.raises(AssertionError):
        close_rel(bad, g[p + "s0_cg_" + kk], "dropped term", 2e-4)


@pytest.mark.parametrize("capture", [False, "cmdlist"])
def test_fp32_attention_engine_replays_reference_train_steps(capture):
    """engine.FusedAttnTrainStep (native attention step: K4 forward / backward, the att MLP on K5, K5b, the merge
    kernel -- no autograd) on the reference's attention / identity / classification fixture (model_kat case 6):
    two train steps in fp32 storage with the recorded draws, against the reference's predictions, clipped
    gradients of BOTH steps and weights after each Adam update."""
    g = load_golden("model_kat.npz")
    p = "c6_"
    assert [str(v) for v in g[p + "cfg"]][:3] == ["attention", "identity", "classification"]
    model, store, task = build_model(gs, g, p, device=DEV, feats_dtype="fp32")
    fan = [int(v) for v in g[p + "fanouts"]]
    ids = torch.from_numpy(g[p + "ids"]).to(DEV)
    tg = torch.from_numpy(g[p + "targets"]).to(DEV)
    w0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    cls = gs.engine.fused_engine_for(model, store)
    assert cls is gs.engine.FusedAttnTrainStep
    eng = cls(model, store, gs.ProblemLosses.classification, ids, tg, capture=capture)
    for step in range(2):
        eng.set_progress(0.25 * step)
        eng.set_sel([g[p + "s%d_sel%d" % (step, h)] for h in range(len(fan))])
        preds = eng(ids, tg).detach().cpu().numpy()
        close(preds, g[p + "s%d_preds" % step], (step, "preds"), 2e-4, 2e-5)
        gn = float(eng.gnorm.item())
        assert abs(gn - float(g[p + "s%d_gradnorm" % step])) <= 2e-4 * max(1.0, float(g[p + "s%d_gradnorm" % step]))
        for k, v in model.named_parameters():
            close_rel(v.grad.cpu().numpy(), g[p + "s%d_cg_%s" % (step, k)], (step, "clipped grad", k), 2e-4)
        for k, v in model.state_dict().items():
            close_update(v.detach().cpu().numpy(), g[p + "w%d_%s" % (step + 1, k)], w0[k].numpy(), (step, "weights", k))
    model.train_sampler.csr(DEV).check()


def test_bf16_attention_engine_matches_reference_and_queue_mode():
    """The bf16 instantiation of the same engine: against the reference's fp32 outputs at bf16 tolerance, and
    its queue mode (gathers of batch i+1 + sampling of batch i+2 beside Adam(i)) == per-call mode, bit for bit."""
    g = load_golden("model_kat.npz")
    p = "c6_"
    fan = [int(v) for v in g[p + "fanouts"]]
    outs = []
    for queued in (False, True):
        model, store, task = build_model(gs, g, p, device=DEV, feats_dtype="bf16")
        ids = torch.from_numpy(g[p + "ids"]).to(DEV)
        tg = torch.from_numpy(g[p + "targets"]).to(DEV)
        eng = gs.engine.FusedAttnTrainStep(model, store, gs.ProblemLosses.classification, ids, tg, capture="cmdlist")
        sels = [[g[p + "s%d_sel%d" % (st, h)] for h in range(len(fan))] for st in range(2)]
        preds = []
        if queued:
            sel_q = torch.stack([torch.cat([torch.from_numpy(np.asarray(x)).reshape(-1) for x in sels[st]])
                                 for st in range(2)])
            eng.load_epoch(torch.stack([ids, ids]), torch.stack([tg, tg]), sel_epoch=sel_q)
            for st in range(2):
                preds.append(eng.step_queue().clone())
        else:
            for st in range(2):
                eng.set_sel(sels[st])
                preds.append(eng(ids, tg).clone())
        torch.cuda.synchronize()
        outs.append((torch.stack(preds), eng.flat_p.clone()))
        for st in range(2):
            close(preds[st].cpu().numpy(), g[p + "s%d_preds" % st], (queued, st, "preds vs reference"), 4e-2, 4e-2)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("capture", [False, "graph"])
def test_fp32_attention_embedding_engine_replays_reference_train_steps(capture):
    """BASELINE config 4's model family through the native engine: attention aggregators over TRAINABLE node
    embeddings (node_embedding prep, no features), regression_mae -- the reference's fixture (model_kat case 4),
    two train steps in fp32 with the recorded draws.  The embedding table's dense-gradient semantics are checked
    through the weights: every row of the table after each Adam step (touched rows moved like the reference's,
    the others not at all)."""
    g = load_golden("model_kat.npz")
    p = "c4_"
    assert [str(v) for v in g[p + "cfg"]][:3] == ["attention", "node_embedding", "regression_mae"]
    ops.set_compute_dtype("fp32")
    model, store, task = build_model(gs, g, p, device=DEV, feats_dtype="fp32")
    assert store is None
    fan = [int(v) for v in g[p + "fanouts"]]
    ids = torch.from_numpy(g[p + "ids"]).to(DEV)
    tg = torch.from_numpy(g[p + "targets"]).to(DEV)
    w0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    cls = gs.engine.fused_engine_for(model, None)
    assert cls is gs.engine.FusedAttnTrainStep
    eng = cls(model, None, gs.ProblemLosses.regression_mae, ids, tg, capture=capture)
    assert eng.emb and eng.tdt == torch.float32
    for step in range(2):
        eng.set_progress(0.25 * step)
        eng.set_sel([g[p + "s%d_sel%d" % (step, h)] for h in range(len(fan))])
        preds = eng(ids, tg).detach().cpu().numpy()
        close(preds, g[p + "s%d_preds" % step], (step, "preds"), 2e-4, 2e-5)
        gn = float(eng.gnorm.item())
        assert abs(gn - float(g[p + "s%d_gradnorm" % step])) <= 2e-4 * max(1.0, float(g[p + "s%d_gradnorm" % step]))
        for k, v in model.named_parameters():
            if k != "prep.embedding.weight":             # the table's gradient is consumed (zeroed) by the step
                close_rel(v.grad.cpu().numpy(), g[p + "s%d_cg_%s" % (step, k)], (step, "clipped grad", k), 2e-4)
        for k, v in model.state_dict().items():
            close_update(v.detach().cpu().numpy(), g[p + "w%d_%s" % (step + 1, k)], w0[k].numpy(), (step, "weights", k))
        assert float(eng._grad_slice(eng.table).abs().max()) == 0.0
    model.train_sampler.csr(DEV).check()


def test_bf16_attention_engine_against_rounding_aware_oracle():
    """The production (bf16) instantiation of FusedAttnTrainStep on the reference's attention fixture with the
    recorded draws, against the oracle with the engine's rounding points (att MLP hidden layer and d a stored in
    bf16, aggregated rows and level outputs rounded, att(.) once per row): two steps, predictions, gradient
    norm, clipped gradients, and the Adam update."""
    g = load_golden("model_kat.npz")
    p = "c6_"
    model, store, task = build_model(gs, g, p, device=DEV, feats_dtype="bf16")
    fan = [int(v) for v in g[p + "fanouts"]]
    ids = torch.from_numpy(g[p + "ids"]).to(DEV)
    tg = torch.from_numpy(g[p + "targets"]).to(DEV)
    sels = [[g[p + "s%d_sel%d" % (st, h)] for h in range(len(fan))] for st in range(2)]
    w0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    ref, w_ref = _oracle_bf16(g, p, store, fan, ids, tg, sels)
    eng = gs.engine.FusedAttnTrainStep(model, store, gs.ProblemLosses.classification, ids, tg, capture="cmdlist")
    for step in range(2):
        eng.set_progress(0.25 * step)
        eng.set_sel(sels[step])
        preds = eng(ids, tg).detach().cpu().numpy()
        torch.cuda.synchronize()
        r = ref[step]
        close(preds, r["preds"].numpy(), (step, "preds vs bf16-aware oracle"), 1e-4, 1e-4)
        gn = float(eng.gnorm.item())
        assert abs(gn - r["gradnorm"]) <= 1e-4 * max(1.0, r["gradnorm"]), (step, gn, r["gradnorm"])
        gerr = 0.0
        for k, v in model.named_parameters():
            a, b = v.grad.cpu().numpy(), r["clipped"][k].numpy()
            gerr = max(gerr, float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-12)))
            close_fro(a, b, (step, "clipped grad", k), 1e-3)
        _note("bf16attn/%d" % step, preds=np.abs(preds - r["preds"].numpy()).max(), gnorm=abs(gn - r["gradnorm"]),
              grad_fro=gerr)
    worst = 0.0
    for k, v in model.state_dict().items():
        d_eng = v.detach().cpu().numpy() - w0[k].numpy()
        d_ref = w_ref[k].numpy() - w0[k].numpy()
        worst = max(worst, float(np.linalg.norm(d_eng - d_ref) / max(np.linalg.norm(d_ref), 1e-12)))
        close_fro(d_eng, d_ref, ("weight update", k), 1e-2)
    _note("bf16attn/w2", upd_fro=worst)


@pytest.mark.parametrize("c,flag", [(0, "GSAGE_MEAN_INPLACE_X"), (3, "GSAGE_MEAN_INPLACE_X"), (4, "GSAGE_POOL_COPY_ROWS"),
                                    (5, "GSAGE_POOL_COPY_ROWS")])
@pytest.mark.parametrize("mode", ["call", "queue"])
def test_operands_read_in_place_equal_the_gathered_copies_bit_for_bit(c, flag, mode):
    """Level-0 operands read in place through the frontier's row list (default) against the gathered-copy paths
    (GSAGE_MEAN_INPLACE_X=0 / GSAGE_POOL_COPY_ROWS=1): same rows, same arithmetic, same summation order -> the same
    predictions and the same weights after two steps, bit for bit."""
    g = load_golden("engine_kat.npz")
    res = []
    for copy in (False, True):
        if copy:
            os.environ[flag] = "0" if flag == "GSAGE_MEAN_INPLACE_X" else "1"
        try:
            p, model, store, fan, ids, tg, sels = _case(g, c, "bf16")
            eng = _engine(model, store, ids, tg, "cmdlist")
            in_place = getattr(eng, "inplace_x", False) or getattr(eng, "inplace0", False)
            assert in_place == (not copy)
            if mode == "queue":
                sel_q = torch.stack([torch.cat([torch.from_numpy(np.asarray(x)).reshape(-1) for x in sels[st]])
                                     for st in range(2)])
                eng.load_epoch(torch.stack([ids, ids]), torch.stack([tg, tg]), sel_epoch=sel_q)
            preds = []
            for step in range(2):
                eng.set_progress(0.25 * step)
                if mode == "queue":
                    preds.append(eng.step_queue().detach().clone())
                else:
                    eng.set_sel(sels[step])
                    preds.append(eng(ids, tg).detach().clone())
            torch.cuda.synchronize()
            res.append((preds, {k: v.detach().clone() for k, v in model.state_dict().items()}))
        finally:
            os.environ.pop(flag, None)
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(a, b)
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k


def test_attention_rows_in_place_equal_the_gathered_copy():
    """FusedAttnTrainStep on the reference's attention fixture: level-0 rows read in place (default) against
    GSAGE_ATTN_COPY_ROWS=1 -- identical predictions and weights (K4's float atomics are not involved: no embeddings)."""
    g = load_golden("model_kat.npz")
    p = "c6_"
    res = []
    for copy in (False, True):
        if copy:
            os.environ["GSAGE_ATTN_COPY_ROWS"] = "1"
        try:
            model, store, task = build_model(gs, g, p, device=DEV, feats_dtype="bf16")
            fan = [int(v) for v in g[p + "fanouts"]]
            ids = torch.from_numpy(g[p + "ids"]).to(DEV)
            tg = torch.from_numpy(g[p + "targets"]).to(DEV)
            eng = gs.engine.FusedAttnTrainStep(model, store, gs.ProblemLosses.classification, ids, tg, capture="cmdlist")
            assert eng.inplace0 == (not copy)
            preds = []
            for step in range(2):
                eng.set_progress(0.25 * step)
                eng.set_sel([g[p + "s%d_sel%d" % (step, h)] for h in range(len(fan))])
                preds.append(eng(ids, tg).detach().clone())
            torch.cuda.synchronize()
            res.append((preds, {k: v.detach().clone() for k, v in model.state_dict().items()}))
        finally:
            os.environ.pop("GSAGE_ATTN_COPY_ROWS", None)
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(a, b)
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k
