"""-m gpu: what round 3 added behind the reference's command lines.

  * the reference's run scripts reach the fused engines unchanged (run.sh:8-10 dense sampler, run.sh:14-17 sparse
    sampler with the compat stream, utils/pokec.sh:5-8 mean + node_embedding) -- argument lists through train.main,
    the engine that ran, the stream positions left behind;
  * the fused sampler consumes the reference's generators in the reference's order: frontiers bit-identical to
    stream_kat.npz (numpy's legacy stream, on the device) and to round3_kat.npz k0 (torch.randperm, dense sampler);
  * batches one seed short of the recorded geometry (the reference's array_split chunks) are padded and the head
    ignores the padding: same predictions / gradients / update as an engine recorded for the short batch;
  * the mean engine over the trainable node-embedding prep and over the dense sampler against two train steps of
    the reference (round3_kat.npz p0/p1, d0/d1) in fp32.
"""
import importlib
import json
import os

import numpy as np
import pytest
import torch
from scipy import sparse
from torch.nn import functional as F

from conftest import load_golden, pkg
from util import build_model, close, close_rel, close_update, csr_of, weights

pytestmark = pytest.mark.gpu
gs = pkg()
ops, nat = gs.ops, gs._native
DEV = "cuda"


@pytest.fixture(autouse=True)
def _setup():
    ops.set_compute_dtype("bf16")
    ops.warmup(torch.device(DEV))
    yield
    gs.nn_modules.SparseUniformNeighborSampler.rng_default = "compat"
    gs.helpers.legacy_stream.drop()
    gs.helpers.legacy_stream.enabled = False
    ops.set_compute_dtype("bf16")
    os.environ.pop("GSAGE_DENSE_TABLE_ADAM", None)


def _specs(fan, dims):
    return [{"n_train_samples": f, "n_val_samples": f, "output_dim": h,
             "activation": (lambda x: x) if i == len(fan) - 1 else F.relu} for i, (f, h) in enumerate(zip(fan, dims))]


# ------------------------------------------------------------------------------------------------------------
# frontiers: the fused sampler on the reference's generators
# ------------------------------------------------------------------------------------------------------------
def _stream_model(adj, D=12, C=5, fan=(5, 3), dims=(16, 16), rng="compat"):
    torch.manual_seed(5)
    gs.nn_modules.SparseUniformNeighborSampler.rng_default = rng
    m = gs.GSSupervised(sampler_class=gs.sampler_lookup["sparse_uniform_neighbor_sampler"], adj=adj, train_adj=adj,
                        prep_class=gs.prep_lookup["identity"], aggregator_class=gs.aggregator_lookup["mean"],
                        input_dim=D, n_nodes=adj.shape[0], n_classes=C, layer_specs=_specs(fan, dims), lr_init=0.01)
    gs.nn_modules.SparseUniformNeighborSampler.rng_default = "compat"
    return m.to(DEV)


def _padded(nodes, chunks, B):
    mids = np.stack([np.concatenate([nodes[c], np.repeat(nodes[c[:1]], B - c.shape[0])]) for c in chunks])
    return torch.from_numpy(mids).to(DEV), [int(c.shape[0]) for c in chunks]


@pytest.mark.parametrize("mode", ["call", "queue"])
def test_fused_sampler_consumes_numpys_stream_like_the_reference(mode):
    """stream_kat.npz = the reference's set_seeds(seed ** 2) -> iterate(shuffle=True) -> two sparse-sampler calls per
    chunk.  The fused mean engine with a compat-mode sampler on the same seed: the epoch's chunks (padded to one
    geometry), every hop of every frontier and the words left in numpy's stream are the reference's, bit for bit
    -- per call (one k_mt_choice launch per batch) and through the epoch queue (one launch per epoch)."""
    g = load_golden("stream_kat.npz")
    adj, nodes = csr_of(g, "g_"), g["nodes"]
    nb = int(g["n_batches"])
    train = importlib.import_module("pytorch-graphsage_amd.train")
    rng = np.random.RandomState(0)
    feats = rng.normal(size=(adj.shape[0], 12)).astype(np.float32)
    store = gs.FeatureStore.from_array(feats, torch.device(DEV), dtype="bf16")
    model = _stream_model(adj)
    gs.helpers.legacy_stream.enabled = True
    gs.set_seeds(int(g["seed"]) ** 2)
    chunks = train.epoch_chunks(nodes, 64)
    assert len(chunks) == nb and all(np.array_equal(nodes[c], g["b%d_ids" % b]) for b, c in enumerate(chunks))
    B = max(c.shape[0] for c in chunks) + 2               # every batch is padded: the padding path is exercised
    ids, live = _padded(nodes, chunks, B)
    tg = torch.from_numpy(rng.randint(0, 5, size=(nb, B))).to(DEV)
    eng = gs.engine.FusedMeanTrainStep(model, store, gs.ProblemLosses.classification, ids[0], tg[0].view(B, 1))
    assert eng.draws == "compat"
    off = eng.off

    def check(front, b):
        n1, n2 = live[b] * 5, live[b] * 15
        assert np.array_equal(front[:live[b]].cpu().numpy(), g["b%d_ids" % b]), b
        assert np.array_equal(front[off[1]:off[1] + n1].cpu().numpy(), g["b%d_h1" % b]), ("hop 1", b)
        assert np.array_equal(front[off[2]:off[2] + n2].cpu().numpy(), g["b%d_h2" % b]), ("hop 2", b)

    before = nat.launch_count()
    if mode == "call":
        for b in range(nb):
            eng(ids[b, :live[b]], tg[b, :live[b]].view(-1, 1))
            check(eng.ids_set[0], b)
    else:
        eng.load_epoch(ids, tg, n_valid=live)             # draws the epoch's values on the device, in order
        eng.g_prime.replay()                              # samples batches 0 and 1, gathers batch 0
        eng._front_ready = True
        check(eng.ids_q[0], 0)
        if nb > 1:
            check(eng.ids_q[1], 1)
        for b in range(nb):
            eng.step_queue()
            if b + 2 < nb:                                # step b sampled batch b + 2 (ring of eng.P buffers)
                check(eng.ids_q[(b + 2) % eng.P], b + 2)     # (a ring of eng.P frontier buffers)
    torch.cuda.synchronize()
    assert nat.launch_count() > before
    model.train_sampler.csr(DEV).check()
    gs.helpers.legacy_stream.release()
    assert np.array_equal(np.random.randint(0, 2 ** 31 - 1, size=4), g["tail"])


@pytest.mark.parametrize("mode", ["call", "queue"])
def test_fused_dense_sampler_consumes_torchs_generator_like_the_reference(mode):
    """round3_kat k0 = the reference's default sampler (UniformNeighborSampler, one torch.randperm per call) over one
    shuffled epoch.  The fused mean engine over the dense adjacency: chunks, both hops of every frontier and the
    state both generators are left in are the reference's."""
    g = load_golden("round3_kat.npz")
    adj_np, nodes, nb = g["k0_adj"], g["k0_nodes"], int(g["k0_n_batches"])
    train = importlib.import_module("pytorch-graphsage_amd.train")
    rng = np.random.RandomState(0)
    feats = rng.normal(size=(adj_np.shape[0], 12)).astype(np.float32)
    store = gs.FeatureStore.from_array(feats, torch.device(DEV), dtype="bf16")
    adj = torch.from_numpy(adj_np).to(DEV)
    torch.manual_seed(5)
    model = gs.GSSupervised(sampler_class=gs.sampler_lookup["uniform_neighbor_sampler"], adj=adj, train_adj=adj,
                            prep_class=gs.prep_lookup["identity"], aggregator_class=gs.aggregator_lookup["mean"],
                            input_dim=12, n_nodes=adj.shape[0], n_classes=5, layer_specs=_specs((5, 3), (16, 16)),
                            lr_init=0.01).to(DEV)
    assert gs.engine.fused_engine_for(model, store) is gs.engine.FusedMeanTrainStep
    gs.set_seeds(int(g["k0_seed"]) ** 2)
    chunks = train.epoch_chunks(nodes, 64)
    B = max(c.shape[0] for c in chunks) + 1               # every batch is padded by one seed
    ids, live = _padded(nodes, chunks, B)
    tg = torch.from_numpy(rng.randint(0, 5, size=(nb, B))).to(DEV)
    gen = torch.get_rng_state()
    eng = gs.engine.FusedMeanTrainStep(model, store, gs.ProblemLosses.classification, ids[0], tg[0].view(B, 1))
    assert eng.draws == "dense" and torch.equal(torch.get_rng_state(), gen)     # building the engine draws nothing
    off = eng.off

    def check(front, b):
        assert np.array_equal(front[:live[b]].cpu().numpy(), g["k0_b%d_ids" % b]), b
        assert np.array_equal(front[off[1]:off[1] + live[b] * 5].cpu().numpy(), g["k0_b%d_h1" % b]), ("hop 1", b)
        assert np.array_equal(front[off[2]:off[2] + live[b] * 15].cpu().numpy(), g["k0_b%d_h2" % b]), ("hop 2", b)

    if mode == "call":
        for b in range(nb):
            eng(ids[b, :live[b]], tg[b, :live[b]].view(-1, 1))
            check(eng.ids_set[0], b)
    else:
        eng.load_epoch(ids, tg, n_valid=live)
        eng.g_prime.replay()
        eng._front_ready = True
        check(eng.ids_q[0], 0)
        check(eng.ids_q[1], 1)
        for b in range(nb):
            eng.step_queue()
            if b + 2 < nb:
                check(eng.ids_q[(b + 2) % eng.P], b + 2)     # (a ring of eng.P frontier buffers)
    torch.cuda.synchronize()
    model.train_sampler.table(DEV).check()
    assert np.array_equal(np.random.randint(0, 2 ** 31 - 1, size=4), g["k0_tail_np"])
    assert np.array_equal(torch.randperm(16).numpy(), g["k0_tail_torch"])


# ------------------------------------------------------------------------------------------------------------
# padded batches
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("agg,dims", [("mean", (128, 128)), ("mean", (16, 16)), ("max_pool", (64, 64)),
                                      ("attention", (16, 16))])
def test_a_padded_batch_equals_the_short_batch(agg, dims):
    """An engine recorded for B seeds, fed b < B (padded with the batch's first seed; head told n_valid = b),
    against an engine recorded for exactly b seeds on the same recorded draws: predictions of the live seeds, the
    gradient norm and the weights after the step agree to fp32 round-off -- the padding contributes nothing."""
    g = load_golden("engine_kat.npz")
    p = "e1_"
    adj = csr_of(g, p + "tadj_")
    feats = g[p + "feats"]
    C, fan, b, B = int(g[p + "n_classes"]), (5, 3), 13, 16
    ops.set_compute_dtype("fp32")
    store = gs.FeatureStore.from_array(feats, torch.device(DEV), dtype="fp32")
    rng = np.random.RandomState(3)
    ids = torch.from_numpy(rng.randint(1, adj.shape[0], size=b)).to(DEV)
    tg = torch.from_numpy(rng.randint(0, C, size=(b, 1))).to(DEV)
    sel_b = [rng.randint(0, adj.shape[1], size=(b, 5)), rng.randint(0, adj.shape[1], size=(b * 5, 3))]
    pad = B - b
    sel_B = [np.concatenate([sel_b[0], np.zeros((pad, 5), dtype=np.int64)]),
             np.concatenate([sel_b[1], np.zeros((pad * 5, 3), dtype=np.int64)])]
    res = []
    for size, sels in ((b, sel_b), (B, sel_B)):
        torch.manual_seed(9)
        gs.nn_modules.SparseUniformNeighborSampler.rng_default = "philox"
        m = gs.GSSupervised(sampler_class=gs.sampler_lookup["sparse_uniform_neighbor_sampler"], adj=adj, train_adj=adj,
                            prep_class=gs.prep_lookup["identity"], aggregator_class=gs.aggregator_lookup[agg],
                            input_dim=feats.shape[1], n_nodes=adj.shape[0], n_classes=C,
                            layer_specs=_specs(fan, dims), lr_init=0.01, weight_decay=1e-4).to(DEV)
        ex_ids = torch.cat([ids, ids[:1].expand(size - b)])
        ex_tg = torch.cat([tg, tg[:1].expand(size - b, 1)])
        cls = gs.engine.fused_engine_for(m, store)
        eng = cls(m, store, gs.ProblemLosses.classification, ex_ids, ex_tg, capture="cmdlist")
        eng.set_sel(sels)
        w0 = eng.flat_p.detach().cpu().numpy().copy()
        preds = eng(ids, tg).detach().clone()             # (size == B: the engine pads and sets n_valid = b)
        torch.cuda.synchronize()
        res.append((preds[:b].cpu().numpy(), float(eng.gnorm.item()), eng.flat_p.detach().cpu().numpy().copy(), w0))
    (p_s, g_s, w_s, w0), (p_l, g_l, w_l, _) = res
    close(p_l, p_s, "preds of the live seeds", 1e-5, 1e-5)
    assert abs(g_l - g_s) <= 1e-5 * max(1.0, g_s), (g_l, g_s)
    # (Adam's first update is lr * sign(g) wherever |g| >> eps: entries whose gradient cancels to round-off may
    # differ by a step between two summation orders -- close_update allows a handful of those)
    close_update(w_l, w_s, w0, "weights after the step", tol_fro=5e-3, tol_elem=1e-5)


def test_l1_head_ignores_padded_rows():
    """gsage_head_l1 with n_valid = b on B rows == the same launch on b rows (predictions, d E of the live rows,
    the partial [dW | db | loss] rows summed); padded rows get a zero gradient."""
    torch.manual_seed(0)
    B, b, D = 48, 37, 128
    E = torch.randn(B, D, device=DEV)
    W, bias = torch.randn(1, D, device=DEV) * 0.1, torch.zeros(1, device=DEV)
    t = torch.randn(B, device=DEV) * 3 + 20
    L = nat.lib()
    outs = []
    for rows, nv in ((b, None), (B, torch.tensor([b], dtype=torch.int32, device=DEV))):
        preds = torch.zeros(rows, 1, device=DEV)
        dE = torch.full((rows, D), 7.0, device=DEV)
        scratch = torch.zeros(L.gsage_head_l1_scratch(rows, D), device=DEV)
        if nv is not None:
            nat.check(L.gsage_head_n_valid_next(nv.data_ptr()), "nv")
        nat.check(L.gsage_head_l1(E.data_ptr(), D, W.data_ptr(), bias.data_ptr(), t.data_ptr(), rows, D, preds.data_ptr(),
                                  dE.data_ptr(), nat.F32, D, scratch.data_ptr(), ops._stream()), "head_l1")
        torch.cuda.synchronize()
        n_wg = (rows + 15) // 16
        outs.append((preds[:b].clone(), dE.clone(), scratch[:n_wg * (D + 2)].view(n_wg, D + 2).sum(0)))
    (p_s, d_s, s_s), (p_l, d_l, s_l) = outs
    assert torch.allclose(p_l, p_s) and torch.allclose(d_l[:b], d_s, rtol=1e-5, atol=1e-7)
    assert float(d_l[b:].abs().max()) == 0.0
    assert torch.allclose(s_l, s_s, rtol=1e-4, atol=1e-6)


# ------------------------------------------------------------------------------------------------------------
# reference fixtures through the new engine modes
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", [0, 1])
@pytest.mark.parametrize("table", ["deferred", "dense"])
@pytest.mark.parametrize("capture", [False, "graph"])
def test_fp32_mean_embedding_engine_replays_reference_train_steps(case, table, capture):
    """utils/pokec.sh:5-8 through FusedMeanTrainStep: mean aggregators over the TRAINABLE node-embedding prep (no
    features), regression_mae -- two train steps of the reference (round3_kat p0 / p1) in fp32 with the recorded
    draws; every weight incl. every row of the embedding table after each step."""
    if table == "dense":
        os.environ["GSAGE_DENSE_TABLE_ADAM"] = "1"
    g = load_golden("round3_kat.npz")
    p = "p%d_" % case
    ops.set_compute_dtype("fp32")
    model, store, task = build_model(gs, g, p, device=DEV, feats_dtype="fp32")
    assert store is None and task == "regression_mae"
    fan = [int(v) for v in g[p + "fanouts"]]
    ids = torch.from_numpy(g[p + "ids"]).to(DEV)
    tg = torch.from_numpy(g[p + "targets"]).to(DEV)
    w0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    cls = gs.engine.fused_engine_for(model, None)
    assert cls is gs.engine.FusedMeanTrainStep
    eng = cls(model, None, gs.ProblemLosses.regression_mae, ids, tg, capture=capture)
    assert eng.emb and eng.fused_l1 and eng.lazy_rows == (table == "deferred") and eng.tdt == torch.float32
    for step in range(2):
        eng.set_progress(0.25 * step)
        eng.set_sel([g[p + "s%d_sel%d" % (step, h)] for h in range(len(fan))])
        preds = eng(ids, tg).detach().cpu().numpy()
        close(preds, g[p + "s%d_preds" % step], (step, "preds"), 2e-4, 2e-5)
        gn = float(eng.gnorm.item())
        assert abs(gn - float(g[p + "s%d_gradnorm" % step])) <= 2e-4 * max(1.0, float(g[p + "s%d_gradnorm" % step]))
        if step == 0:
            for k, v in model.named_parameters():
                if k != "prep.embedding.weight":         # the table's gradient is consumed (zeroed) by the step
                    close_rel(v.grad.cpu().numpy(), g[p + "s0_cg_%s" % k], (step, "clipped grad", k), 2e-4)
        for k, v in model.state_dict().items():          # (state_dict settles the deferred rows)
            close_update(v.detach().cpu().numpy(), g[p + "w%d_%s" % (step + 1, k)], w0[k].numpy(), (step, "weights", k))
        assert float(eng._grad_slice(eng.table).abs().max()) == 0.0
    model.train_sampler.csr(DEV).check()


@pytest.mark.parametrize("case", [0, 1])
@pytest.mark.parametrize("capture", [False, "cmdlist"])
def test_fp32_mean_engine_over_the_dense_sampler_replays_reference_train_steps(case, capture):
    """run.sh:8-10 (the reference's DEFAULT sampler) through FusedMeanTrainStep: two train steps of the reference
    (round3_kat d0: width-256 seed level, d1: generic path) in fp32 from the recorded torch seed -- the engine
    draws the permutations itself, so the frontier, the predictions and the weights after two Adam updates are
    the reference's."""
    g = load_golden("round3_kat.npz")
    p = "d%d_" % case
    ops.set_compute_dtype("fp32")
    fan, dims = [int(v) for v in g[p + "fanouts"]], [int(v) for v in g[p + "out_dims"]]
    adj, tadj = torch.from_numpy(g[p + "adj"]).to(DEV), torch.from_numpy(g[p + "tadj"]).to(DEV)
    feats = g[p + "feats"]
    store = gs.FeatureStore.from_array(feats, torch.device(DEV), dtype="fp32")
    model = gs.GSSupervised(sampler_class=gs.sampler_lookup["uniform_neighbor_sampler"], adj=adj, train_adj=tadj,
                            prep_class=gs.prep_lookup["identity"], aggregator_class=gs.aggregator_lookup["mean"],
                            input_dim=feats.shape[1], n_nodes=adj.shape[0], n_classes=int(g[p + "n_classes"]),
                            layer_specs=_specs(fan, dims), lr_init=0.01, weight_decay=float(g[p + "weight_decay"]))
    model.load_state_dict(weights(g, p + "w0_"))
    model = model.to(DEV)
    model.optimizer = torch.optim.Adam(model.parameters(), lr=model.lr, weight_decay=float(g[p + "weight_decay"]))
    w0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    ids = torch.from_numpy(g[p + "ids"]).to(DEV)
    tg = torch.from_numpy(g[p + "targets"]).to(DEV)
    eng = gs.engine.FusedMeanTrainStep(model, store, gs.ProblemLosses.classification, ids, tg, capture=capture)
    assert eng.draws == "dense"
    torch.manual_seed(int(g[p + "torch_seed"]))
    for step in range(2):
        eng.set_progress(0.25 * step)
        preds = eng(ids, tg).detach().cpu().numpy()
        if step == 0:
            front = eng.ids_set[0].cpu().numpy()
            assert np.array_equal(front[eng.off[1]:eng.off[2]], g[p + "s0_h1"])
            assert np.array_equal(front[eng.off[2]:eng.off[3]], g[p + "s0_h2"])
        close(preds, g[p + "s%d_preds" % step], (step, "preds"), 2e-4, 2e-5)
        gn = float(eng.gnorm.item())
        assert abs(gn - float(g[p + "s%d_gradnorm" % step])) <= 2e-4 * max(1.0, float(g[p + "s%d_gradnorm" % step]))
        if step == 0:
            for k, v in model.named_parameters():
                close_rel(v.grad.cpu().numpy(), g[p + "s0_cg_%s" % k], (step, "clipped grad", k), 2e-4)
        for k, v in model.state_dict().items():
            close_update(v.detach().cpu().numpy(), g[p + "w%d_%s" % (step + 1, k)], w0[k].numpy(), (step, "weights", k))
    model.train_sampler.table(DEV).check()


# ------------------------------------------------------------------------------------------------------------
# the reference's run scripts, argument lists unchanged
# ------------------------------------------------------------------------------------------------------------
def _toy_problem(tmp_path, kind, n=900, D=12, C=4):
    rng = np.random.RandomState(0)
    folds = np.array(["train"] * 700 + ["val"] * 150 + ["test"] * (n + 1 - 850))
    if kind == "pokec_dense":                             # utils/convert-pokec.py -> problem.h5: dense adjacency, NO features
        K = 16
        adj = rng.randint(0, n, size=(n + 1, K))
        adj[n] = n
        folds[n] = "dummy"
        targets = (20.0 + 10.0 * (np.arange(n + 1) % 3)).astype(np.float32).reshape(-1, 1)
        prob = {"task": "regression_mae", "n_classes": 1, "folds": folds, "targets": targets, "sparse": False,
                "adj": adj, "train_adj": adj}
    elif kind == "dense":                                 # utils/convert.py:71-98: 0-based ids, dummy = last row
        K = 16
        adj = rng.randint(0, n, size=(n + 1, K))
        adj[n] = n
        feats = rng.normal(size=(n + 1, D)).astype(np.float32)
        feats[n] = 0
        folds[n] = "dummy"
        prob = {"task": "classification", "n_classes": C, "feats": feats, "folds": folds,
                "targets": feats[:, :C].argmax(1).reshape(-1, 1), "sparse": False, "adj": adj, "train_adj": adj}
    else:
        degs = rng.randint(1, 12, size=n + 1)
        degs[0] = 0
        rows = np.repeat(np.arange(n + 1), degs)
        cols = np.concatenate([np.arange(d) for d in degs])
        adj = sparse.csr_matrix((rng.randint(1, n + 1, size=rows.shape[0]), (rows, cols)))
        folds[0] = "dummy"
        if kind == "sparse":
            feats = rng.normal(size=(n + 1, D)).astype(np.float32)
            feats[0] = 0
            prob = {"task": "classification", "n_classes": C, "feats": feats, "folds": folds,
                    "targets": feats[:, :C].argmax(1).reshape(-1, 1), "sparse": True, "adj": adj, "train_adj": adj}
        else:                                             # pokec: no features, regression on the node id's residue
            targets = (20.0 + 10.0 * (np.arange(n + 1) % 3)).astype(np.float32).reshape(-1, 1)
            prob = {"task": "regression_mae", "n_classes": 1, "folds": folds, "targets": targets, "sparse": True,
                    "adj": adj, "train_adj": adj}
    path = os.path.join(str(tmp_path), "%s-problem.npz" % kind)
    gs.problem.save_problem_npz(path, prob)
    return path


def _run_cli(argv, capsys):
    train = importlib.import_module("pytorch-graphsage_amd.train")
    step = train.main(argv)
    cap = capsys.readouterr()
    lines = [json.loads(l) for l in cap.out.strip().split("\n") if l.startswith("{")]
    np_tail = np.random.randint(0, 2 ** 31 - 1, size=4)
    torch_tail = torch.randperm(16)
    return step, lines, cap.err, np_tail, torch_tail


@pytest.mark.parametrize("script,kind,extra,engine", [
    ("run.sh:8-10", "dense", ["--aggregator-class", "mean"], "FusedMeanTrainStep"),
    ("run.sh:14-17", "sparse", ["--aggregator-class", "mean", "--sampler-class", "sparse_uniform_neighbor_sampler"],
     "FusedMeanTrainStep"),
    ("pokec.sh:5-8", "pokec", ["--aggregator-class", "mean", "--sampler-class", "sparse_uniform_neighbor_sampler",
                               "--prep-class", "node_embedding", "--epochs", "3"], "FusedMeanTrainStep"),
    # the command behind the reference's only published number (utils/pokec.sh:11-15): DEFAULT dense sampler
    ("pokec.sh:11-13", "pokec_dense", ["--aggregator-class", "mean", "--prep-class", "node_embedding", "--epochs", "3"],
     "FusedMeanTrainStep"),
])
def test_reference_run_scripts_reach_the_fused_engines(tmp_path, capsys, script, kind, extra, engine):
    """The argument lists of the reference's run scripts, unchanged (only --problem-path points at a toy problem in
    the .npz twin of problem.h5), through train.main with its defaults (--engine auto, --rng compat): a fused
    engine runs (and says so on stderr), one JSON line per batch in the reference's schema, and -- the run being
    the reference's run -- numpy's and torch's generators end where the module path (--engine eager) leaves them,
    with train metrics that track it."""
    path = _toy_problem(tmp_path, kind)
    argv = ["--problem-path", path] + extra
    step, out, err, np_tail, torch_tail = _run_cli(argv, capsys)
    assert step is not None and type(step).__name__ == engine, (script, err[-500:])
    assert "train_step runs on %s" % engine in err
    assert set(out[-1]) == {"epoch", "train_metric", "val_metric", "time"}
    logged = [o for o in out if "epoch_progress" in o]
    epochs = 3 if "--epochs" in extra else 10
    assert len(logged) == epochs * 2                      # 700 train nodes, batch 512 -> two chunks per epoch
    # the same command line on the module path: same generators, consumed in the same order
    step2, out2, err2, np_tail2, torch_tail2 = _run_cli(argv + ["--engine", "eager"], capsys)
    assert step2 is None
    assert np.array_equal(np_tail, np_tail2), "numpy's stream ends elsewhere"
    assert torch.equal(torch_tail, torch_tail2), "torch's CPU generator ends elsewhere"
    logged2 = [o for o in out2 if "epoch_progress" in o]
    assert [o["epoch_progress"] for o in logged] == [o["epoch_progress"] for o in logged2]
    if kind.startswith("pokec"):
        a, b = out[-1]["train_metric"], out2[-1]["train_metric"]
        assert abs(a - b) <= 0.15 * abs(b) + 0.5, (a, b)
        assert out[-1]["train_metric"] < logged[0]["train_metric"]            # the MAE falls
    else:
        a, b = out[-1]["val_metric"]["micro"], out2[-1]["val_metric"]["micro"]
        assert abs(a - b) <= 0.08, (a, b)


# ------------------------------------------------------------------------------------------------------------
# end metric of the production (bf16) engine against the fp32 instantiation
# ------------------------------------------------------------------------------------------------------------
def test_bf16_engine_trains_like_the_fp32_engine_at_the_bench_shape():
    """The bf16 engines are tight only against an oracle that mirrors their rounding points; the fp32 instantiation is
    what is pinned to the reference's outputs (engine_kat, 2e-4).  This ties the two together by an END metric at the
    bench workload's shape (fan-out 25/10, 128/128, 602-d features, 41 classes; a 30 k-node graph with learnable
    labels): the same 60 Philox-sampled batches through both instantiations -- the loss curves must track each other
    (bf16 storage perturbs a trajectory, it must not bend it) and both must learn."""
    rng = np.random.RandomState(0)
    n, D, C, B, steps = 30_000, 602, 41, 512, 60
    deg = np.clip(np.exp(rng.normal(3.0, 1.0, size=n + 1)).astype(np.int64), 1, 400)
    deg[0] = 0
    indptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    adj = sparse.csr_matrix((rng.randint(1, n + 1, size=int(indptr[-1])), gs.store.row_positions(indptr), indptr),
                            shape=(n + 1, int(deg.max())))
    proto = rng.normal(size=(C, D)).astype(np.float32)
    labels = rng.randint(0, C, size=n + 1)
    feats = (proto[labels] * 0.5 + rng.normal(size=(n + 1, D))).astype(np.float32)
    feats[0] = 0
    ids = torch.from_numpy(rng.randint(1, n + 1, size=(steps, B))).to(DEV)
    tg = torch.from_numpy(labels[ids.cpu().numpy()]).to(DEV)
    curves = {}
    for prec in ("fp32", "bf16"):
        ops.set_compute_dtype(prec)
        store = gs.FeatureStore.from_array(feats, torch.device(DEV), dtype=prec)
        torch.manual_seed(4)
        gs.nn_modules.SparseUniformNeighborSampler.rng_default = "philox"
        m = gs.GSSupervised(sampler_class=gs.sampler_lookup["sparse_uniform_neighbor_sampler"], adj=adj, train_adj=adj,
                            prep_class=gs.prep_lookup["identity"], aggregator_class=gs.aggregator_lookup["mean"],
                            input_dim=D, n_nodes=n + 1, n_classes=C, layer_specs=_specs((25, 10), (128, 128)),
                            lr_init=0.01).to(DEV)
        m.train_sampler.seed = 5
        eng = gs.engine.FusedMeanTrainStep(m, store, gs.ProblemLosses.classification, ids[0], tg[0].view(B, 1))
        eng.load_epoch(ids, tg)
        losses = []
        for s in range(steps):
            preds = eng.step_queue()
            losses.append(float(F.cross_entropy(preds, tg[s])))
        curves[prec] = np.array(losses)
    a, b = curves["fp32"], curves["bf16"]
    assert a[-5:].mean() < 0.6 * a[:5].mean() and b[-5:].mean() < 0.6 * b[:5].mean(), (a[:3], a[-3:], b[-3:])
    assert np.abs(a - b).max() <= 0.05 * a[0] + 0.02, float(np.abs(a - b).max())
    assert abs(a[-10:].mean() - b[-10:].mean()) <= 0.03 * a[-10:].mean() + 0.01


@pytest.mark.gpu
@pytest.mark.parametrize("D,n,dtype", [(602, 25, "bf16"), (256, 25, "bf16"), (64, 20, "bf16"), (602, 25, "fp32"), (100, 32, "fp32")])
def test_attention_one_parent_per_workgroup_equals_one_per_wave(D, n, dtype):
    """K4 / K4' give a hop with few parents and long fan-outs one WORKGROUP per parent (four waves share its children;
    gsage_attn.hip, WPP = 4) instead of one wave: the same parents as the head of a launch large enough to take the
    one-wave path must come out equal -- the backward bit for bit (per-child dot products are formed the same way),
    the forward up to the order of the fp32 partial sums (nn_modules.py:309-315)."""
    dev = torch.device("cuda")
    ops = gs.ops
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    g = torch.Generator(device="cpu"); g.manual_seed(5)
    M_small, M_big, Ha, N = 512, 8192, 32, 50_000
    vec = 8 if dtype == "bf16" else 4
    ld = (D + vec - 1) // vec * vec
    table = torch.zeros(N, ld, dtype=tdt, device=dev)
    table[:, :D] = torch.randn(N, D, generator=g).to(dev).to(tdt)
    ids = torch.randint(0, N, (M_big * n,), generator=g).to(dev)
    na = torch.randn(M_big * n, Ha, generator=g).to(dev)
    xa = torch.randn(M_big, Ha, generator=g).to(dev)
    gout = torch.randn(M_big, D, generator=g).to(dev)
    lib, nat = gs._native.lib(), gs._native

    def run(M):
        agg = torch.empty(M, D, device=dev); ws = torch.empty(M, n, device=dev)
        nat.check(lib.gsage_attn_aggregate(na.data_ptr(), Ha, xa.data_ptr(), Ha, table.data_ptr(), ops._code(tdt), ld,
                                           ids.data_ptr(), M, n, Ha, D, agg.data_ptr(), D, ws.data_ptr(), ops._stream()), "fwd")
        dna = torch.empty(M * n, Ha, device=dev); dxa = torch.empty(M, Ha, device=dev)
        nat.check(lib.gsage_attn_bwd(gout.data_ptr(), D, ws.data_ptr(), na.data_ptr(), Ha, xa.data_ptr(), Ha,
                                     table.data_ptr(), ops._code(tdt), ld, ids.data_ptr(), M, n, Ha, D, dna.data_ptr(), Ha,
                                     dxa.data_ptr(), Ha, ops._stream()), "bwd")
        torch.cuda.synchronize()
        return agg, ws, dna, dxa

    a_s, w_s, dn_s, dx_s = run(M_small)
    a_b, w_b, dn_b, dx_b = run(M_big)
    assert torch.equal(w_s, w_b[:M_small])
    torch.testing.assert_close(a_s, a_b[:M_small], rtol=2e-6, atol=2e-6)
    assert torch.equal(dn_s, dn_b[:M_small * n]) and torch.equal(dx_s, dx_b[:M_small])
    # and against the definition
    rows = table[ids[:M_small * n], :D].float().view(M_small, n, D)
    w = torch.softmax(torch.bmm(na[:M_small * n].view(M_small, n, Ha), xa[:M_small].unsqueeze(2)).squeeze(2), dim=1)
    torch.testing.assert_close(a_s, (rows * w.unsqueeze(2)).sum(1), rtol=1e-4, atol=1e-4)
