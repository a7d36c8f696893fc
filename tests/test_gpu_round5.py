"""-m gpu, round 5: the step without a finalisation launch (the update's workgroups sum the partial buffers
themselves, gsage_adam_desc.reduce_descs; ticks in the K5b launch) and the seed level on the matrix cores
(gsage_mean_tail_mfma) -- through the engines, against the launches they replace."""
import numpy as np
import pytest
import torch

from conftest import pkg
from test_gpu_engine import _model, _problem
from util import close, close_fro

pytestmark = pytest.mark.gpu
gs = pkg()
ops = gs.ops
nat = gs._native
DEV = "cuda"


@pytest.fixture(autouse=True)
def _setup():
    ops.set_compute_dtype("bf16")
    ops.warmup(torch.device(DEV))
    yield
    ops.set_compute_dtype("bf16")


def _run(monkeypatch, env, dims, fans, B, mode, n_steps=5, C=5, clip_scale=1.0):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    adj, feats, rng = _problem(n=900, D=40, seed=4)
    store = gs.FeatureStore.from_array(feats * clip_scale, torch.device(DEV), dtype="bf16")
    ids = torch.from_numpy(rng.randint(1, adj.shape[0], size=(n_steps + 2, B))).to(DEV)
    tg = torch.from_numpy(rng.randint(0, C, size=(n_steps + 2, B))).to(DEV)
    m = _model(adj, feats.shape[1], C, dims, fans)
    eng = gs.engine.FusedMeanTrainStep(m, store, gs.ProblemLosses.classification, ids[0], tg[0].view(B, 1),
                                       capture=mode if mode != "queue" else "cmdlist")
    before = nat.launch_count()
    if mode == "queue":
        eng.load_epoch(ids, tg)
        before = nat.launch_count()
        preds = torch.stack([eng.step_queue().clone() for _ in range(n_steps)])
    else:
        preds = torch.stack([eng(ids[k], tg[k].view(B, 1)).clone() for k in range(n_steps)])
    torch.cuda.synchronize()
    return {"preds": preds.float().cpu().numpy(), "p": eng.flat_p.clone().cpu().numpy(),
            "g": eng.flat_g.clone().cpu().numpy(), "norm": float(eng.gnorm.item()), "fold": eng._fold_finalize(),
            "launches": nat.launch_count() - before, "step": int(eng.step.item()), "ctr": int(eng.counter.item()),
            "mfma": eng.fused_tail and eng._tail_on_mfma(), "k1_in_tail": eng._k1_in_tail()}


@pytest.mark.parametrize("mode", ["queue", "cmdlist", False])
@pytest.mark.parametrize("dims,fans,B,scale", [((128, 128), (25, 10), 64, 1.0), ((128, 128), (25, 10), 64, 30.0),
                                               ((16, 8), (5, 3), 33, 1.0), ((32, 16, 8), (4, 3, 2), 20, 1.0)])
def test_update_that_sums_the_partial_buffers_equals_the_finalisation_launch(monkeypatch, mode, dims, fans, B, scale):
    """GSAGE_FOLD_FINALIZE=1 (opt-in: it measured slower): no gsage_finalize_grads launch -- every update workgroup sums the partial
    buffers of its own 1 024 elements in the finalisation's order (the SAME gradient bits), forms the norm with the
    others inside the launch and ticks ride in K5b.  Against GSAGE_FOLD_FINALIZE=0: identical gradients and
    counters; weights equal unless the clip is active (the norm's terms are added in another order: scale = 30
    makes it active), one launch fewer per step."""
    a = _run(monkeypatch, {"GSAGE_FOLD_FINALIZE": "0"}, dims, fans, B, mode, clip_scale=scale)
    b = _run(monkeypatch, {"GSAGE_FOLD_FINALIZE": "1"}, dims, fans, B, mode, clip_scale=scale)
    assert not a["fold"] and b["fold"]
    assert a["step"] == b["step"] == 5 and a["ctr"] == b["ctr"]
    assert b["launches"] == a["launches"] - 5, (a["launches"], b["launches"])
    assert abs(a["norm"] - b["norm"]) <= 2e-6 * max(1.0, a["norm"])
    if a["norm"] < 4.9 and scale == 1.0:
        assert np.array_equal(a["p"], b["p"]) and np.array_equal(a["g"], b["g"]) and np.array_equal(a["preds"], b["preds"])
    else:
        close(b["preds"], a["preds"], "preds", 1e-4, 1e-5)
        close_fro(b["p"], a["p"], "weights", 1e-5)


@pytest.mark.parametrize("mode", ["queue", "cmdlist"])
@pytest.mark.parametrize("B,C", [(64, 5), (50, 41), (512, 41)])
def test_matrix_core_seed_level_through_the_engine(monkeypatch, mode, B, C):
    """FusedMeanTrainStep with the seed level on the matrix cores (16 seeds per workgroup) against the same engine on
    the VALU seed-level kernel: five steps, width-256 levels, fan-out 25 / 10 (the gather role rides along in queue
    mode): predictions and weights agree to fp32 round-off plus the bf16 roundings whose inputs moved by it."""
    a = _run(monkeypatch, {"GSAGE_TAIL_MFMA": "0"}, (128, 128), (25, 10), B, mode, C=C)
    b = _run(monkeypatch, {"GSAGE_TAIL_MFMA": "1"}, (128, 128), (25, 10), B, mode, C=C)
    assert b["mfma"] and not a["mfma"]
    close(b["preds"], a["preds"], "preds", 2e-3, 2e-4)
    close_fro(b["p"], a["p"], "weights after five steps", 3e-3)      # (Adam: ~1e-2 of the UPDATES)
    assert abs(a["norm"] - b["norm"]) <= 2e-3 * max(1.0, a["norm"])


def test_in_launch_norm_is_admitted_by_the_device_not_by_a_constant():
    """gsage_gather_adam_capacity: resident workgroups of the gather launch for a given sampler-role LDS footprint
    (occupancy x CUs, one per CU kept as margin) -- what bounds the meeting of the update's workgroups."""
    L = nat.lib()
    small, big = L.gsage_gather_adam_capacity(nat.BF16, 4000), L.gsage_gather_adam_capacity(nat.BF16, 150 * 1024)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert small >= 3 * cus and small % cus == 0
    assert big == 0                                       # one workgroup per CU fits: nothing left after the margin
    assert L.gsage_gather_adam_capacity(nat.F32, 4000) >= 3 * cus


def _np_draw(high, counts):
    return [np.random.choice(high, c).astype(np.int32) for c in counts]


@pytest.mark.parametrize("high,counts,gaps", [(21657, [140800 * 3, 512 * 25, 999999, 7, 2000000], True),
                                             (4096, [450000, 450000], False), (3, [1500000], False),
                                             (2 ** 31 + 5, [600000, 1], True)])
def test_many_workgroups_consume_numpys_stream_like_numpy(high, counts, gaps):
    """gsage_mt_choice_par through ops.mt_choice_segments: np.random.choice(high, .) requests served by up to 256
    workgroups (jump-ahead) -- every value, the state and the position equal numpy's; the stream continues on the
    device (one-workgroup kernel) and on the host."""
    from importlib import import_module
    helpers = import_module("pytorch-graphsage_amd.helpers")
    ls = helpers.LegacyStreamOnDevice()
    np.random.seed(99)
    np.random.choice(1000, 333)                            # (somewhere inside a block)
    state0 = np.random.get_state()
    want = _np_draw(high, counts)
    tail_want = np.random.choice(high, 5000).astype(np.int32)
    after = np.random.get_state()
    np.random.set_state(state0)
    st = ls.acquire(torch.device(DEV))
    offs, o = [], 0
    for c in counts:
        offs.append(o)
        o += c + (13 if gaps else 0)
    out = torch.full((o + 5,), -7, dtype=torch.int32, device=DEV)
    before = nat.launch_count()
    ops.mt_choice_segments(st, high, list(zip(offs, counts)), out)
    assert nat.launch_count() - before == 5                # stream | count | scan | write | finisher
    tail = torch.zeros(5000, dtype=torch.int32, device=DEV)
    ops.mt_choice_segments(st, high, [(0, 5000)], tail)     # small request: the one-workgroup kernel continues
    got = out.cpu().numpy()
    for off, c, w in zip(offs, counts, want):
        assert np.array_equal(got[off:off + c], w), (high, c)
        if gaps:
            assert (got[off + c:off + c + 13] == -7).all()
    assert np.array_equal(tail.cpu().numpy(), tail_want)
    ls.release()
    s1 = np.random.get_state()
    assert s1[2] == after[2] and np.array_equal(s1[1], after[1])


def test_parallel_stream_finisher_and_block_zero():
    """The C entry point directly: (a) chunks sized BELOW the request (an unlucky acceptance estimate): the serial
    finisher completes it; (b) a request that ends inside the block the stream already stands in."""
    L = nat.lib()
    table = ops.mt_jump_table(torch.device(DEV))
    for seed, count, n_wg, per in ((5, 700000, 3, 4), (6, 100, 2, 1), (7, 1300000, 16, 2)):
        np.random.seed(seed)
        np.random.choice(50, 17)
        name, key, pos, *_ = np.random.get_state()
        want = np.random.choice(21657, count).astype(np.int32)
        a_key, a_pos = np.random.get_state()[1:3]
        host = np.concatenate([np.asarray(key, dtype=np.uint32), np.array([pos], dtype=np.uint32)])
        st = torch.from_numpy(host.view(np.int32).copy()).to(DEV)
        cum = torch.tensor([0, count], dtype=torch.int64, device=DEV)
        off = torch.tensor([3], dtype=torch.int64, device=DEV)
        out = torch.full((count + 8,), -1, dtype=torch.int32, device=DEV)
        scratch = torch.empty(int(L.gsage_mt_choice_par_scratch(n_wg)), dtype=torch.uint8, device=DEV)
        nat.check(L.gsage_mt_choice_par(st.data_ptr(), 21657, 1, cum.data_ptr(), off.data_ptr(), count, out.data_ptr(),
                                        table.data_ptr(), scratch.data_ptr(), scratch.numel(), n_wg, per, None), "par")
        got = out.cpu().numpy()
        assert np.array_equal(got[3:3 + count], want) and (got[:3] == -1).all() and (got[3 + count:] == -1).all()
        s = st.cpu().numpy().view(np.uint32)
        assert int(s[624]) == int(a_pos) and np.array_equal(s[:624], np.asarray(a_key, dtype=np.uint32)), (seed, count)


@pytest.mark.parametrize("kind,extra", [
    ("sparse", ["--aggregator-class", "mean", "--sampler-class", "sparse_uniform_neighbor_sampler"]),
    ("dense", ["--aggregator-class", "mean"]),
    ("pokec_dense", ["--aggregator-class", "mean", "--prep-class", "node_embedding"]),
    ("sparse", ["--aggregator-class", "max_pool", "--sampler-class", "sparse_uniform_neighbor_sampler"]),
    ("sparse", ["--aggregator-class", "attention", "--sampler-class", "sparse_uniform_neighbor_sampler"]),
    ("sparse", ["--aggregator-class", "mean", "--prep-class", "node_embedding", "--sampler-class",
                "sparse_uniform_neighbor_sampler"]),                       # [features | embedding] rows
])
def test_fused_evaluation_equals_the_module_path(tmp_path, kind, extra):
    """train.FusedEvaluator (engine forward launches over the validation sampler, the fold's draws taken from the
    reference's generators in the reference's order) against train.evaluate (the module path, reference
    train.py:29-36) on the same weights: the same metric up to bf16 rounding, and numpy's / torch's generators end
    in the same place."""
    from importlib import import_module
    from test_gpu_round3 import _toy_problem
    train = import_module("pytorch-graphsage_amd.train")
    helpers = import_module("pytorch-graphsage_amd.helpers")
    args = train.parse_args(["--problem-path", _toy_problem(tmp_path, kind)] + extra)
    gs.set_seeds(5)
    gs.nn_modules.SparseUniformNeighborSampler.rng_default = "compat"
    helpers.legacy_stream.enabled = True
    try:
        problem = gs.NodeProblem(problem_path=args.problem_path, cuda=True)
        model = train.build_model(args, problem).cuda()
        cls = gs.engine.fused_engine_for(model, problem.feats)
        assert cls is not None
        ev = train.FusedEvaluator(cls, model, problem)
        outs = []
        for fused in (False, True):
            gs.set_seeds(11)
            torch.manual_seed(11)
            m = [(ev(mode) if fused else train.evaluate(model, problem, mode=mode)) for mode in ("val", "test", "val")]
            helpers.legacy_stream.release()
            outs.append((m, np.random.randint(0, 2 ** 31 - 1, size=4), torch.randperm(16)))
        assert not ev.off and set(ev.engines) == {"val", "test"}
        assert np.array_equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
        for a, b in zip(outs[0][0], outs[1][0]):
            if isinstance(a, dict):
                assert abs(a["micro"] - b["micro"]) <= 0.05 and abs(a["macro"] - b["macro"]) <= 0.08, (a, b)
            else:
                assert abs(a - b) <= 0.02 * abs(a) + 1e-3, (a, b)
    finally:
        helpers.legacy_stream.enabled = False
        helpers.legacy_stream.drop()


@pytest.mark.parametrize("mode", ["queue", False])
def test_sampler_role_of_the_projection_launch_changes_nothing(monkeypatch, mode):
    """GSAGE_K1_IN_K5=1: batch i+2 is sampled by a z-slice of the level-0 projection's launch of step i
    (gsage_hops_role_next; a ring of three frontier buffers) instead of in the launch that carries the update: the
    same Philox words, the same frontier, the same weights bit for bit."""
    a = _run(monkeypatch, {"GSAGE_K1_IN_K5": "0", "GSAGE_K1_IN_TAIL": "0"}, (128, 128), (25, 10), 64, mode, n_steps=7)
    b = _run(monkeypatch, {"GSAGE_K1_IN_K5": "1", "GSAGE_K1_IN_TAIL": "0"}, (128, 128), (25, 10), 64, mode, n_steps=7)
    assert np.array_equal(a["preds"], b["preds"]) and np.array_equal(a["p"], b["p"])
    assert a["ctr"] == b["ctr"] and a["step"] == b["step"] == 7


@pytest.mark.parametrize("B,wgs", [(64, "32"), (200, "7"), (512, "32"), (512, "128")])
def test_sampler_role_of_the_seed_level_launch_changes_nothing(monkeypatch, B, wgs):
    """GSAGE_K1_IN_TAIL=1 (the default where it applies): batch i+2 is sampled by workgroups of the SEED-LEVEL launch
    of step i (gsage_hops_role_next before gsage_mean_tail_mfma; sample_hops_wide: many seeds per 512-thread
    workgroup, four samples per lane and trip) instead of in the launch that carries the update: the same Philox
    words, the same frontier, the same weights bit for bit -- for whole and ragged seed groups."""
    monkeypatch.setenv("GSAGE_TAIL_SMP_WGS", wgs)
    a = _run(monkeypatch, {"GSAGE_K1_IN_K5": "0", "GSAGE_K1_IN_TAIL": "0"}, (128, 128), (25, 10), B, "queue", n_steps=7)
    b = _run(monkeypatch, {"GSAGE_K1_IN_K5": "0", "GSAGE_K1_IN_TAIL": "1"}, (128, 128), (25, 10), B, "queue", n_steps=7)
    assert b["k1_in_tail"] and not a["k1_in_tail"]
    assert np.array_equal(a["preds"], b["preds"]) and np.array_equal(a["p"], b["p"])
    assert a["ctr"] == b["ctr"] and a["step"] == b["step"] == 7


@pytest.mark.parametrize("agg", ["mean", "max_pool", "mean_pool", "attention"])
def test_module_path_runs_no_library_gemm(monkeypatch, agg):
    """GSSupervised.train_step -- the literal aggregator_lookup[...] plug-ins under autograd -- forward AND backward
    on this library's kernels only: with every library-GEMM entry of torch poisoned the step still runs (round 4's
    backward called torch.mm for the input gradients and the head was an nn.Linear)."""
    import torch.nn.functional as F

    def boom(*a, **k):
        raise AssertionError("a library GEMM was called in the module path")
    adj, feats, rng = _problem(n=500, D=24, seed=2)
    B, C = 48, 5
    store = gs.FeatureStore.from_array(feats, torch.device(DEV), dtype="bf16")
    m = _model(adj, feats.shape[1], C, (16, 8), (5, 3), agg=agg)
    ids = torch.from_numpy(rng.randint(1, adj.shape[0], size=B)).to(DEV)
    tg = torch.from_numpy(rng.randint(0, C, size=(B, 1))).to(DEV)
    w0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    for name in ("mm", "matmul", "addmm", "bmm", "einsum"):
        monkeypatch.setattr(torch, name, boom)
    monkeypatch.setattr(F, "linear", boom)
    for _ in range(2):
        preds = m.train_step(ids=ids, feats=store, targets=tg, loss_fn=gs.ProblemLosses.classification)
    torch.cuda.synchronize()
    assert preds.shape == (B, C) and bool(torch.isfinite(preds).all())
    moved = [k for k, v in m.state_dict().items() if not torch.equal(v, w0[k])]
    assert len(moved) == len(w0), "every parameter received a gradient and moved"


# ------------------------------------------------------------------------------------------------------------
# the node-embedding prep concatenated with features (nn_modules.py:152-153) in the mean engine
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("src,p", [("model_kat.npz", "c5_"), ("round5_kat.npz", "e0_"), ("round5_kat.npz", "e1_")])
@pytest.mark.parametrize("table", ["deferred", "dense"])
@pytest.mark.parametrize("capture", [False, "cmdlist"])
def test_fp32_mean_engine_over_embedding_beside_features_replays_reference_train_steps(monkeypatch, src, p, table, capture):
    """[features | prep.fc(embedding[ids])] rows (reference nn_modules.py:143-155) through FusedMeanTrainStep: two
    train steps of the reference (model_kat c5: feature width 12, regression_mae; round5_kat e0: 24, classification;
    e1: 40, regression_mae with weight decay) in fp32 with the recorded draws -- predictions, gradient norm, clipped
    gradients, and every weight incl. every row of the embedding table after each step."""
    from conftest import load_golden
    from util import build_model, close_rel, close_update
    if table == "dense":
        monkeypatch.setenv("GSAGE_DENSE_TABLE_ADAM", "1")
    g = load_golden(src)
    ops.set_compute_dtype("fp32")
    model, store, task = build_model(gs, g, p, device=DEV, feats_dtype="fp32")
    assert store is not None and type(model.prep).__name__ == "NodeEmbeddingPrep" and model.prep.input_dim == store.dim
    fan = [int(v) for v in g[p + "fanouts"]]
    ids = torch.from_numpy(g[p + "ids"]).to(DEV)
    tg = torch.from_numpy(g[p + "targets"]).to(DEV)
    w0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    cls = gs.engine.fused_engine_for(model, store)
    assert cls is gs.engine.FusedMeanTrainStep, gs.engine.FusedMeanTrainStep.why_not(model, store)
    eng = cls(model, store, getattr(gs.ProblemLosses, task), ids, tg, capture=capture)
    assert eng.emb and eng.D0 == store.dim and eng.din[0] == store.dim + eng.E and eng.tdt == torch.float32
    assert eng.lazy_rows == (table == "deferred")
    for step in range(2):
        eng.set_progress(0.25 * step)
        eng.set_sel([g[p + "s%d_sel%d" % (step, h)] for h in range(len(fan))])
        preds = eng(ids, tg).detach().cpu().numpy()
        close(preds, g[p + "s%d_preds" % step], (step, "preds"), 2e-4, 2e-5)
        gn, want = float(eng.gnorm.item()), float(g[p + "s%d_gradnorm" % step])
        assert abs(gn - want) <= 2e-4 * max(1.0, want), (gn, want)
        if step == 0:
            for k, v in model.named_parameters():
                if k != "prep.embedding.weight":         # the table's gradient is consumed (zeroed) by the step
                    close_rel(v.grad.cpu().numpy(), g[p + "s0_cg_%s" % k], (step, "clipped grad", k), 2e-4)
        for k, v in model.state_dict().items():          # (state_dict settles the deferred rows)
            close_update(v.detach().cpu().numpy(), g[p + "w%d_%s" % (step + 1, k)], w0[k].numpy(), (step, "weights", k))


def _emb_feats_model(adj, D, C, dims, fans, task):
    import torch.nn.functional as F
    torch.manual_seed(9)
    gs.nn_modules.SparseUniformNeighborSampler.rng_default = "philox"
    specs = [{"n_train_samples": f, "n_val_samples": f, "output_dim": h,
              "activation": (lambda x: x) if i == len(dims) - 1 else F.relu} for i, (h, f) in enumerate(zip(dims, fans))]
    m = gs.GSSupervised(sampler_class=gs.sampler_lookup["sparse_uniform_neighbor_sampler"], adj=adj, train_adj=adj,
                        prep_class=gs.prep_lookup["node_embedding"], aggregator_class=gs.aggregator_lookup["mean"],
                        input_dim=D, n_nodes=adj.shape[0], n_classes=C, layer_specs=specs, lr_init=0.01, weight_decay=1e-4)
    gs.nn_modules.SparseUniformNeighborSampler.rng_default = "compat"
    m.train_sampler.seed = m.val_sampler.seed = 77
    return m.to(DEV)


@pytest.mark.parametrize("dims,fans,D", [((128, 128), (10, 5), 40), ((64, 32), (5, 3), 24)])
def test_bf16_engine_over_embedding_beside_features_tracks_the_fp32_engine(dims, fans, D):
    """The production precision of the same rows: bf16 storage ([features | prep output] operand rows, the seed level
    in one launch at 128 / 128) against the fp32 instantiation on the same Philox-sampled batches through the device
    queue -- same frontier, predictions and the weights after six steps within bf16's rounding; the features' columns
    of the level-0 weights learn too (the fc_x / fc_neib gradients span D0 + E columns)."""
    adj, feats, rng = _problem(n=700, D=D, seed=6)
    B, C, steps = 64, 5, 6
    ids = torch.from_numpy(rng.randint(1, adj.shape[0], size=(steps, B))).to(DEV)
    tg = torch.from_numpy(rng.randint(0, C, size=(steps, B))).to(DEV)
    res = {}
    for prec in ("fp32", "bf16"):
        ops.set_compute_dtype(prec)
        store = gs.FeatureStore.from_array(feats, torch.device(DEV), dtype=prec)
        m = _emb_feats_model(adj, D, C, dims, fans, "classification")
        w0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
        eng = gs.engine.FusedMeanTrainStep(m, store, gs.ProblemLosses.classification, ids[0], tg[0].view(B, 1),
                                           capture="cmdlist")
        assert eng.emb and eng.D0 == D and eng.fused_tail == (dims == (128, 128))
        eng.load_epoch(ids, tg)
        preds = [eng.step_queue().detach().float().cpu().numpy().copy() for _ in range(steps)]
        w = {k: v.detach().float().cpu().numpy() for k, v in m.state_dict().items()}
        moved = {k: float((m.state_dict()[k] - w0[k]).abs().max()) for k in w0}
        assert all(v > 0 for v in moved.values()), moved
        fx = m.state_dict()["agg_layers.0.fc_x.weight"] - w0["agg_layers.0.fc_x.weight"]
        assert float(fx[:, :D].abs().max()) > 0 and float(fx[:, D:].abs().max()) > 0
        res[prec] = (preds, w)
    for s in range(steps):                  # (a trajectory: the bf16 run's rounding compounds over the updates)
        close_fro(res["bf16"][0][s], res["fp32"][0][s], ("preds", s), 2e-2 if s == 0 else 8e-2)
    for k in res["fp32"][1]:                # (Adam's first updates are sign-like: entries whose gradient is noise move
        close_fro(res["bf16"][1][k], res["fp32"][1][k], ("weights", k), 0.2)   # by O(lr) either way -- a loose bound)


@pytest.mark.parametrize("p", ["f0_", "f1_", "f2_"])
@pytest.mark.parametrize("capture", [False, "cmdlist"])
def test_fp32_pool_and_attention_engines_over_node_embedding_replay_reference_train_steps(p, capture):
    """The other aggregators over the node-embedding prep: max_pool / mean_pool (nn_modules.py:207-256) through
    FusedPoolTrainStep -- round5_kat f0 (max_pool, no features, classification) and f1 (mean_pool beside 24 feature
    columns, regression_mae: the L1 head under the pool engine) -- and attention (nn_modules.py:279-317) beside 24
    feature columns through FusedAttnTrainStep (f2): two train steps of the reference in fp32 with the recorded
    draws; every weight incl. every row of the embedding table after each step."""
    from conftest import load_golden
    from util import build_model, close_rel, close_update
    g = load_golden("round5_kat.npz")
    ops.set_compute_dtype("fp32")
    model, store, task = build_model(gs, g, p, device=DEV, feats_dtype="fp32")
    fan = [int(v) for v in g[p + "fanouts"]]
    ids = torch.from_numpy(g[p + "ids"]).to(DEV)
    tg = torch.from_numpy(g[p + "targets"]).to(DEV)
    w0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    cls = gs.engine.fused_engine_for(model, store)
    want_cls = gs.engine.FusedAttnTrainStep if p == "f2_" else gs.engine.FusedPoolTrainStep
    assert cls is want_cls, want_cls.why_not(model, store)
    eng = cls(model, store, getattr(gs.ProblemLosses, task), ids, tg, capture=capture)
    assert eng.emb and eng.D0 == (store.dim if store is not None else 0) and eng.tdt == torch.float32
    assert eng.fused_l1 == (task == "regression_mae") and eng.fused_head == (task == "classification")
    for step in range(2):
        eng.set_progress(0.25 * step)
        eng.set_sel([g[p + "s%d_sel%d" % (step, h)] for h in range(len(fan))])
        preds = eng(ids, tg).detach().cpu().numpy()
        close(preds, g[p + "s%d_preds" % step], (step, "preds"), 2e-4, 2e-5)
        gn, want = float(eng.gnorm.item()), float(g[p + "s%d_gradnorm" % step])
        assert abs(gn - want) <= 2e-4 * max(1.0, want), (gn, want)
        if step == 0:
            for k, v in model.named_parameters():
                if k != "prep.embedding.weight":         # the table's gradient is consumed (zeroed) by the step
                    close_rel(v.grad.cpu().numpy(), g[p + "s0_cg_%s" % k], (step, "clipped grad", k), 2e-4)
        for k, v in model.state_dict().items():          # (state_dict settles the deferred rows)
            close_update(v.detach().cpu().numpy(), g[p + "w%d_%s" % (step + 1, k)], w0[k].numpy(), (step, "weights", k))


@pytest.mark.parametrize("agg,D", [("max_pool", 0), ("mean_pool", 40)])
def test_bf16_pool_engine_over_node_embedding_tracks_the_fp32_engine(agg, D):
    """The production precision of the same model family (K3 / K5 on the packed operands over the prep's output
    rows): bf16 against the fp32 instantiation over the same Philox-sampled batches through the device queue."""
    import torch.nn.functional as F
    adj, feats, rng = _problem(n=700, D=max(D, 8), seed=8)
    B, C, steps, dims, fans = 64, 5, 5, (64, 64), (6, 4)
    ids = torch.from_numpy(rng.randint(1, adj.shape[0], size=(steps, B))).to(DEV)
    tg = torch.from_numpy(rng.randint(0, C, size=(steps, B))).to(DEV)
    res = {}
    for prec in ("fp32", "bf16"):
        ops.set_compute_dtype(prec)
        store = gs.FeatureStore.from_array(feats, torch.device(DEV), dtype=prec) if D else None
        torch.manual_seed(9)
        gs.nn_modules.SparseUniformNeighborSampler.rng_default = "philox"
        specs = [{"n_train_samples": f, "n_val_samples": f, "output_dim": h,
                  "activation": (lambda x: x) if i == len(dims) - 1 else F.relu} for i, (h, f) in enumerate(zip(dims, fans))]
        m = gs.GSSupervised(sampler_class=gs.sampler_lookup["sparse_uniform_neighbor_sampler"], adj=adj, train_adj=adj,
                            prep_class=gs.prep_lookup["node_embedding"], aggregator_class=gs.aggregator_lookup[agg],
                            input_dim=D if D else None, n_nodes=adj.shape[0], n_classes=C, layer_specs=specs,
                            lr_init=0.01, weight_decay=1e-4)
        gs.nn_modules.SparseUniformNeighborSampler.rng_default = "compat"
        m.train_sampler.seed = m.val_sampler.seed = 77
        m = m.to(DEV)
        w0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
        assert gs.engine.fused_engine_for(m, store) is gs.engine.FusedPoolTrainStep
        eng = gs.engine.FusedPoolTrainStep(m, store, gs.ProblemLosses.classification, ids[0], tg[0].view(B, 1),
                                           capture="cmdlist")
        eng.load_epoch(ids, tg)
        preds = [eng.step_queue().detach().float().cpu().numpy().copy() for _ in range(steps)]
        sd = m.state_dict()
        assert all(float((sd[k] - w0[k]).abs().max()) > 0 for k in w0)
        res[prec] = (preds, {k: v.detach().float().cpu().numpy() for k, v in sd.items()})
    for s in range(steps):                  # (the weights are not compared: a pooled maximum that changes hands under
        close_fro(res["bf16"][0][s], res["fp32"][0][s], ("preds", s), 2e-2 if s == 0 else 8e-2)   # bf16 reroutes a whole
    #                                          gradient row, and Adam's first updates are sign-like)
    assert all(np.isfinite(v).all() for v in res["bf16"][1].values())


def test_sampler_role_with_three_hops_and_when_it_does_not_fit(monkeypatch):
    """BASELINE configs[4]'s geometry (three layers, fan-outs 15 / 10 / 5: 750 ids per seed at the widest hop): with
    two seeds per sampler workgroup the role runs and changes nothing; at B = 512 sixteen seeds' frontiers (twice 96 KB)
    do not fit the launch's LDS -- gsage_mean_tail_mfma_sampler_wgs says 0 and K1 stays in the launch that carries
    the update."""
    dims, fans = (128, 128, 128), (15, 10, 5)
    a = _run(monkeypatch, {"GSAGE_K1_IN_K5": "0", "GSAGE_K1_IN_TAIL": "0"}, dims, fans, 64, "queue", n_steps=5)
    b = _run(monkeypatch, {"GSAGE_K1_IN_K5": "0", "GSAGE_K1_IN_TAIL": "1"}, dims, fans, 64, "queue", n_steps=5)
    assert b["k1_in_tail"] and not a["k1_in_tail"] and b["mfma"]
    assert np.array_equal(a["preds"], b["preds"]) and np.array_equal(a["p"], b["p"]) and a["ctr"] == b["ctr"]
    assert nat.lib().gsage_mean_tail_mfma_sampler_wgs(512, 750) == 0 and nat.lib().gsage_mean_tail_mfma_sampler_wgs(512, 250) == 32
    c = _run(monkeypatch, {"GSAGE_K1_IN_K5": "0", "GSAGE_K1_IN_TAIL": "1"}, dims, fans, 512, "queue", n_steps=3)
    assert not c["k1_in_tail"] and c["step"] == 3 and np.isfinite(c["preds"]).all()
