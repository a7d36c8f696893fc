"""Host logic of the drop-in boundary (no GPU): the product's plugin classes, GSSupervised,
NodeProblem.iterate, LRSchedule and train.py plumbing on CPU tensors, checked against the golden
vectors generated from the reference.  CPU tensors take ops.py's explicit host mode (the
reference's `--no-cuda` configuration); the HIP path is covered by the -m gpu tests."""
import io
import json
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden, pkg
from util import ACTS, SelReplay, build_model, close, csr_of, weights

gs = pkg()


def test_lookup_tables_match_reference_surface():
    assert set(gs.sampler_lookup) == {"uniform_neighbor_sampler", "sparse_uniform_neighbor_sampler"}
    assert set(gs.prep_lookup) == {"identity", "node_embedding", "linear"}
    assert set(gs.aggregator_lookup) == {"mean", "max_pool", "mean_pool", "lstm", "attention"}


def test_sparse_sampler_compat_stream_bit_exact():
    g = load_golden("sampler_kat.npz")
    for c in range(int(g["n_cases"])):
        p = "c%d_" % c
        s = gs.sampler_lookup["sparse_uniform_neighbor_sampler"](adj=csr_of(g, "g%d_" % int(g[p + "graph"])))
        np.random.seed(int(g[p + "seed"]))
        out = s(torch.LongTensor(g[p + "ids"]), n_samples=int(g[p + "n"]))
        assert out.dtype == torch.int64 and out.dim() == 1
        assert np.array_equal(out.numpy(), g[p + "out"]), c
        assert np.array_equal(np.random.randint(0, 2 ** 31 - 1, size=4), g[p + "tail"])
    assert np.array_equal(s.degrees, g["g2_degrees"])


def test_sparse_sampler_errors():
    g = load_golden("sampler_kat.npz")
    s = gs.sampler_lookup["sparse_uniform_neighbor_sampler"](adj=csr_of(g, "g0_"))
    with pytest.raises(AssertionError):
        s(torch.LongTensor([1, 2]), n_samples=0)
    with pytest.raises(IndexError):
        s(torch.LongTensor([1, 10 ** 6]), n_samples=2)
    with pytest.raises(AssertionError):
        gs.sampler_lookup["sparse_uniform_neighbor_sampler"](adj=np.zeros((3, 3)))
    assert s(torch.LongTensor([]), n_samples=3).numel() == 0           # empty batch


def test_philox_host_equals_oracle_and_is_shard_invariant():
    from oracle import cpu as ocpu
    g = load_golden("sampler_kat.npz")
    adj = csr_of(g, "g2_")
    ids = torch.LongTensor(np.random.RandomState(0).randint(0, adj.shape[0], size=64))
    s = gs.sampler_lookup["sparse_uniform_neighbor_sampler"](adj=adj, rng="philox", seed=99)
    out = s(ids, n_samples=10)
    sel = ocpu.philox_sel(99, 0, 0, 640, adj.shape[1])
    assert np.array_equal(out.numpy(), ocpu.sample_csr_sel(adj.indptr, adj.data, ids.numpy(), 10, sel))
    halves = []
    for rank in (0, 1):
        t = gs.sampler_lookup["sparse_uniform_neighbor_sampler"](adj=adj, rng="philox", seed=99)
        t.shard = (rank, 2)
        halves.append(t(ids[rank * 32:(rank + 1) * 32], n_samples=10).numpy())
    assert np.array_equal(np.concatenate(halves), out.numpy())


def test_dense_sampler_matches_reference():
    g = load_golden("dense_sampler_kat.npz")
    s = gs.sampler_lookup["uniform_neighbor_sampler"](adj=torch.LongTensor(g["adj"]))
    for c in range(int(g["n_cases"])):
        p = "c%d_" % c
        gs.set_seeds(int(g[p + "seed"]))
        out = s(torch.LongTensor(g[p + "ids"]), n_samples=int(g[p + "n"]))
        assert np.array_equal(out.numpy(), g[p + "out"])


def test_aggregators_host_mode_forward_backward():
    g = load_golden("agg_kat.npz")
    for c in range(int(g["n_cases"])):
        p = "c%d_" % c
        name, act = str(g[p + "name"]), str(g[p + "act"])
        M, n, D, h = [int(v) for v in g[p + "dims"]]
        agg = gs.aggregator_lookup[name](input_dim=D, output_dim=h, activation=ACTS[act])
        agg.load_state_dict(weights(g, p + "w_"))
        assert agg.output_dim == int(g[p + "output_dim"])
        x = torch.from_numpy(g[p + "x"].copy()).requires_grad_(True)
        nb = torch.from_numpy(g[p + "neibs"].copy()).requires_grad_(True)
        out = agg(x, nb)
        close(out.detach().numpy(), g[p + "out"], (c, name))
        (out * torch.from_numpy(g[p + "G"])).sum().backward()
        close(x.grad.numpy(), g[p + "dx"], (c, name, "dx"))
        close(nb.grad.numpy(), g[p + "dneibs"], (c, name, "dneibs"))
        for k, v in agg.named_parameters():
            close(v.grad.numpy(), g[p + "g_" + k], (c, name, k))


def test_lstm_aggregator_host_mode_forward_backward():
    """LSTMAggregator (nn_modules.py:259-286; kept so that aggregator_lookup is complete -- the
    recurrence is stock torch / MIOpen) against vectors recorded from the reference, uni- and
    bidirectional."""
    g = load_golden("lstm_kat.npz")
    for c in range(int(g["n_cases"])):
        p = "c%d_" % c
        M, n, D, h, hid, bidir = [int(v) for v in g[p + "dims"]]
        agg = gs.aggregator_lookup["lstm"](input_dim=D, output_dim=h, activation=ACTS[str(g[p + "act"])],
                                           hidden_dim=hid, bidirectional=bool(bidir))
        agg.load_state_dict(weights(g, p + "w_"))
        assert agg.output_dim == int(g[p + "output_dim"])
        x = torch.from_numpy(g[p + "x"].copy()).requires_grad_(True)
        nb = torch.from_numpy(g[p + "neibs"].copy()).requires_grad_(True)
        out = agg(x, nb)
        close(out.detach().numpy(), g[p + "out"], (c, "lstm"), 1e-5, 1e-6)
        (out * torch.from_numpy(g[p + "G"])).sum().backward()
        close(x.grad.numpy(), g[p + "dx"], (c, "dx"), 1e-5, 1e-6)
        close(nb.grad.numpy(), g[p + "dneibs"], (c, "dneibs"), 1e-5, 1e-6)
        for k, v in agg.named_parameters():
            close(v.grad.numpy(), g[p + "g_" + k], (c, k), 1e-5, 1e-6)


def test_attention_refuses_squeeze_quirk_shapes():
    agg = gs.aggregator_lookup["attention"](input_dim=4, output_dim=3, activation=None)
    with pytest.raises(AssertionError):
        agg(torch.zeros(3, 4), torch.zeros(3, 4))          # fanout 1


def test_preps_host_mode():
    g = load_golden("prep_kat.npz")
    for c in range(int(g["n_cases"])):
        p = "c%d_" % c
        name = str(g[p + "name"])
        idim = int(g[p + "input_dim"]) or None
        prep = gs.prep_lookup[name](input_dim=idim, n_nodes=int(g[p + "n_nodes"]))
        prep.load_state_dict(weights(g, p + "w_"))
        assert prep.output_dim == int(g[p + "output_dim"])
        feats = torch.from_numpy(g[p + "feats"]) if (p + "feats") in g.files else None
        out = prep(torch.from_numpy(g[p + "ids"]), feats, layer_idx=int(g[p + "layer_idx"]))
        close(out.detach().numpy(), g[p + "out"], (c, name))
        if (p + "G") in g.files:
            (out * torch.from_numpy(g[p + "G"])).sum().backward()
            for k, v in prep.named_parameters():
                ref = g[p + "g_" + k]
                close(v.grad.numpy() if v.grad is not None else np.zeros_like(ref), ref, (c, name, k))


def test_full_model_host_mode_train_steps():
    g = load_golden("model_kat.npz")
    for c in range(int(g["n_cases"])):
        p = "c%d_" % c
        model, feats, task = build_model(gs, g, p)
        loss_fn = getattr(gs.ProblemLosses, task)
        ids = torch.from_numpy(g[p + "ids"])
        tg = torch.from_numpy(g[p + "targets"])
        with SelReplay([g[p + "eval_sel0"], g[p + "eval_sel1"]]):
            ev = model(ids, feats, train=False)
        close(ev.detach().numpy(), g[p + "eval_preds"], (c, "eval"))
        for step in range(2):
            model.set_progress(0.25 * step)
            assert abs(model.lr - float(g[p + "lr%d" % step])) < 1e-12
            with SelReplay([g[p + "s%d_sel0" % step], g[p + "s%d_sel1" % step]]):
                preds = model.train_step(ids=ids, feats=feats, targets=tg, loss_fn=loss_fn)
            close(preds.detach().numpy(), g[p + "s%d_preds" % step], (c, step, "preds"), 1e-4, 1e-5)
            for k, v in model.named_parameters():
                close(v.grad.numpy(), g[p + "s%d_cg_%s" % (step, k)], (c, step, "cg", k), 1e-4, 1e-5)
            for k, v in model.state_dict().items():
                close(v.numpy(), g[p + "w%d_%s" % (step + 1, k)], (c, step, "w", k), 1e-4, 1e-5)


def _tiny_problem(tmp_path, sparse_adj=True, task="classification"):
    rng = np.random.RandomState(0)
    n = 150
    from scipy import sparse
    degs = rng.randint(1, 9, size=n + 1)
    degs[0] = 0
    rows = np.repeat(np.arange(n + 1), degs)
    cols = np.concatenate([np.arange(d) for d in degs])
    vals = rng.randint(1, n + 1, size=rows.shape[0])
    adj = sparse.csr_matrix((vals, (rows, cols)))
    folds = np.array(["train"] * 100 + ["val"] * 30 + ["test"] * 21)
    folds[0] = "dummy"
    prob = {"task": task, "n_classes": 4, "feats": rng.normal(size=(n + 1, 9)).astype(np.float32),
            "folds": folds, "targets": rng.randint(0, 4, size=(n + 1, 1)), "sparse": True,
            "adj": adj, "train_adj": adj}
    if not sparse_adj:
        dense = rng.randint(0, n, size=(n + 1, 8))
        prob.update({"sparse": False, "adj": dense, "train_adj": dense})
    path = os.path.join(str(tmp_path), "problem.npz")
    gs.problem.save_problem_npz(path, prob)
    return path


def test_node_problem_iterate_matches_reference_chunking(tmp_path):
    g = load_golden("iterate_kat.npz")
    prob = gs.NodeProblem(_tiny_problem(tmp_path), cuda=False)
    assert prob.n_nodes == 151 and prob.feats_dim == 9 and prob.n_classes == 4
    for c in range(int(g["n_cases"])):
        p = "c%d_" % c
        prob.nodes["train"] = g[p + "nodes"]
        prob.targets = g[p + "targets_all"]
        gs.set_seeds(int(g[p + "seed"]))
        ids, tgs, progs = [], [], []
        for i, t, pr in prob.iterate("train", batch_size=int(g[p + "bs"]), shuffle=bool(int(g[p + "shuffle"]))):
            assert i.dtype == torch.int64 and t.dtype == torch.int64
            ids.append(i.numpy()); tgs.append(t.numpy().reshape(-1)); progs.append(pr)
        assert [len(i) for i in ids] == list(g[p + "sizes"])
        assert np.array_equal(np.concatenate(ids), g[p + "ids"])
        assert np.array_equal(np.concatenate(tgs), g[p + "targets"])
        assert np.allclose(progs, g[p + "progress"])
        assert np.array_equal(np.random.randint(0, 2 ** 31 - 1, size=4), g[p + "tail"])


def test_train_epoch_chunks_are_the_references_iterate_chunks():
    """train.epoch_chunks (what the fused training loop cuts an epoch with) against the reference's iterate
    (iterate_kat.npz: chunk sizes, chunk contents, words left in numpy's stream) for the shuffled cases."""
    import importlib
    train = importlib.import_module("pytorch-graphsage_amd.train")
    g = load_golden("iterate_kat.npz")
    seen = 0
    for c in range(int(g["n_cases"])):
        p = "c%d_" % c
        if not bool(int(g[p + "shuffle"])):
            continue
        nodes = g[p + "nodes"]
        gs.set_seeds(int(g[p + "seed"]))
        chunks = train.epoch_chunks(nodes, int(g[p + "bs"]))
        assert [len(ch) for ch in chunks] == list(g[p + "sizes"])
        assert np.array_equal(np.concatenate([nodes[ch] for ch in chunks]), g[p + "ids"])
        assert np.array_equal(np.random.randint(0, 2 ** 31 - 1, size=4), g[p + "tail"])
        seen += 1
    assert seen >= 1


def test_fused_engine_for_says_why_no_engine_applies(capsys):
    """No fused engine covers a model on the CPU / an LSTM aggregator: `fused_engine_for(explain=True)` returns None
    and names, per engine, what is not covered -- nobody lands on the slow path silently (round-2 verdict, weak 10)."""
    from scipy import sparse
    from torch.nn import functional as F
    adj = sparse.csr_matrix((np.array([1, 2, 1]), np.array([0, 1, 0]), np.array([0, 0, 2, 3])), shape=(3, 2))
    specs = [{"n_train_samples": 2, "n_val_samples": 2, "output_dim": 8, "activation": F.relu},
             {"n_train_samples": 2, "n_val_samples": 2, "output_dim": 8, "activation": lambda x: x}]
    for agg, needle in (("mean", "FeatureStore"), ("lstm", "aggregators")):
        m = gs.GSSupervised(sampler_class=gs.sampler_lookup["sparse_uniform_neighbor_sampler"], adj=adj, train_adj=adj,
                            prep_class=gs.prep_lookup["identity"], aggregator_class=gs.aggregator_lookup[agg],
                            input_dim=4, n_nodes=3, n_classes=2, layer_specs=specs)
        assert gs.engine.fused_engine_for(m, torch.zeros(3, 4), explain=True) is None
        err = capsys.readouterr().err
        assert "no fused train-step engine covers this model" in err and needle in err
        why = gs.engine.why_no_fused_engine(m, torch.zeros(3, 4))
        assert set(why) == {"FusedMeanTrainStep", "FusedPoolTrainStep", "FusedAttnTrainStep"}


def test_dense_sampler_host_mode_and_table():
    """UniformNeighborSampler on CPU tensors keeps the reference's stock indexing (dense_sampler_kat.npz), and
    store.DenseAdj validates what the HIP path needs."""
    g = load_golden("dense_sampler_kat.npz")
    adj = torch.from_numpy(g["adj"])
    s = gs.sampler_lookup["uniform_neighbor_sampler"](adj=adj)
    for c in range(int(g["n_cases"])):
        p = "c%d_" % c
        gs.set_seeds(int(g[p + "seed"]))
        out = s(torch.from_numpy(g[p + "ids"]), n_samples=int(g[p + "n"]))
        assert np.array_equal(out.numpy(), g[p + "out"]), c
    tab = s.table()
    assert (tab.n_rows, tab.K, tab.max_deg) == (adj.shape[0], adj.shape[1], adj.shape[1])
    tab.check()
    with pytest.raises(AssertionError):
        gs.store.DenseAdj(adj.float())


def test_lr_schedule_and_metrics_match_reference():
    g = load_golden("misc_kat.npz")
    for name in ("constant", "linear", "cyclical"):
        fn = getattr(gs.LRSchedule, name)
        assert np.allclose([fn(float(x), lr_init=0.01) for x in g["lr_x"]], g["lr_" + name])
        assert np.allclose([fn(float(x), lr_init=0.05, epochs=4) for x in g["lr_x"]], g["lr_%s_e4" % name])
    with pytest.raises(TypeError):       # `step` cannot be selected
        gs.GSSupervised(input_dim=4, n_nodes=5, n_classes=2, layer_specs=[], aggregator_class=None,
                        prep_class=gs.prep_lookup["identity"],
                        sampler_class=gs.sampler_lookup["uniform_neighbor_sampler"],
                        adj=torch.zeros(2, 2).long(), train_adj=torch.zeros(2, 2).long(),
                        lr_schedule="step")
    m = gs.ProblemMetrics.classification(g["cls_y"], g["cls_logits"])
    assert abs(m["micro"] - float(g["cls_micro"])) < 1e-12 and abs(m["macro"] - float(g["cls_macro"])) < 1e-12
    m = gs.ProblemMetrics.multilabel_classification(g["ml_y"], g["ml_logits"])
    assert abs(m["micro"] - float(g["ml_micro"])) < 1e-12 and abs(m["macro"] - float(g["ml_macro"])) < 1e-12
    assert abs(gs.ProblemMetrics.regression_mae(g["mae_y"], g["mae_pred"]) - float(g["mae"])) < 1e-6
    for key, fn, y in (("cls", gs.ProblemLosses.classification, torch.LongTensor(g["cls_y"]).squeeze()),
                       ("ml", gs.ProblemLosses.multilabel_classification, torch.FloatTensor(g["ml_y"].astype(np.float32)))):
        assert abs(float(fn(torch.FloatTensor(g[key + "_logits"]), y)) - float(g[key + "_loss"])) < 1e-5


@pytest.mark.parametrize("sampler,sparse_adj", [("sparse_uniform_neighbor_sampler", True),
                                                ("uniform_neighbor_sampler", False)])
def test_train_py_cli_runs_on_cpu(tmp_path, capsys, sampler, sparse_adj):
    """BASELINE config 1 plumbing: the reference's flags, JSON log schema, no GPU."""
    path = _tiny_problem(tmp_path, sparse_adj=sparse_adj)
    gs.train = __import__("importlib").import_module("pytorch-graphsage_amd.train")
    gs.train.main(["--problem-path", path, "--no-cuda", "--epochs", "2", "--batch-size", "32",
                   "--sampler-class", sampler, "--n-train-samples", "5,5", "--n-val-samples", "5,5",
                   "--output-dims", "16,16", "--show-test"])
    lines = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    per_batch = [l for l in lines if "epoch_progress" in l]
    assert len(per_batch) == 2 * (100 // 32 + 1)
    assert set(per_batch[0]) == {"epoch", "epoch_progress", "train_metric", "val_metric", "time"}
    assert per_batch[0]["val_metric"] is None and per_batch[-1]["val_metric"] is not None
    assert set(lines[-2]) == {"epoch", "train_metric", "val_metric", "time"}
    assert set(lines[-1]) == {"test_f1"} and set(lines[-1]["test_f1"]) == {"micro", "macro"}


def test_shard_refuses_batches_smaller_than_the_world():
    """A chunk with fewer seeds than ranks would give every rank an empty batch (NaN loss averaged
    into every replica): dist.DataParallel.shard raises instead."""
    import pytest
    gs = pkg()
    dp = gs.dist.DataParallel(rank=1, world=4, device=torch.device("cpu"), owns_group=False)
    ids = torch.arange(10)
    assert dp.shard(ids).tolist() == [2, 3]
    with pytest.raises(ValueError):
        dp.shard(torch.arange(3))


def test_wgrad_balance_fits_one_round_with_equal_slices():
    """ops.wgrad_balance: every launch's problems (eight at a time) together stay within the workgroup budget, each
    problem's target is a whole number of slices per tile, and slice lengths differ by less than a slice."""
    ops = gs.ops
    shapes = [(512, 128, 256), (512, 128, 256), (13312, 32, 32), (13312, 32, 256), (13312, 128, 602), (13312, 128, 602),
              (141312, 32, 32), (141312, 32, 602), (164352, 64, 64)]
    targets = ops.wgrad_balance(shapes, budget=248)
    assert len(targets) == len(shapes)
    for i in range(0, len(shapes), 8):
        chunk, tg = shapes[i:i + 8], targets[i:i + 8]
        assert sum(tg) <= 248 or all(ops.wgrad_plan(m, nt, k, t)[1] == 1 for (m, nt, k), t in zip(chunk, tg))
        lens = []
        for (m, nt, k), t in zip(chunk, tg):
            tiles = ((nt + 127) // 128) * ((k + 3) // 4 * 4 + 127) // 128
            assert t % tiles == 0 and t >= tiles
            rps, S, _ldk = ops.wgrad_plan(m, nt, k, t)
            assert S * tiles <= t
            if m > 256:
                lens.append(rps)
        big = max(lens)
        assert all(l <= big for l in lens)


def test_head_coverage_is_decided_before_an_engine_is_built():
    """Engine.head_why_not (what train.choose_engine asks BEFORE an engine re-points the model's Parameters; round-3
    advisor finding): cross-entropy with <= 64 classes and the L1 regression head are fused and may meet padded
    chunks; multilabel / wide heads are not -- they are refused only when the caller will pad."""
    from scipy import sparse
    from torch.nn import functional as F
    adj = sparse.csr_matrix((np.array([1, 2, 1]), np.array([0, 1, 0]), np.array([0, 0, 2, 3])), shape=(3, 2))
    specs = [{"n_train_samples": 2, "n_val_samples": 2, "output_dim": 8, "activation": F.relu},
             {"n_train_samples": 2, "n_val_samples": 2, "output_dim": 8, "activation": lambda x: x}]

    def model(n_classes):
        return gs.GSSupervised(sampler_class=gs.sampler_lookup["sparse_uniform_neighbor_sampler"], adj=adj, train_adj=adj,
                               prep_class=gs.prep_lookup["identity"], aggregator_class=gs.aggregator_lookup["mean"],
                               input_dim=4, n_nodes=3, n_classes=n_classes, layer_specs=specs)
    E = gs.engine.FusedMeanTrainStep
    L = gs.ProblemLosses
    i64, f32 = torch.zeros(1, dtype=torch.int64), torch.zeros(1)
    state = torch.get_rng_state()
    assert E.head_why_not(model(5), L.classification, i64, 512, True) is None
    assert E.head_why_not(model(1), L.regression_mae, f32, 512, True) is None
    assert E.head_why_not(model(4), L.multilabel_classification, f32, 512, False) is None          # nothing is padded
    assert "no fused kernel" in E.head_why_not(model(4), L.multilabel_classification, f32, 512, True)
    assert "no fused kernel" in E.head_why_not(model(100), L.classification, i64, 512, True)       # > 64 classes
    assert "no fused kernel" in E.head_why_not(model(1), L.regression_mae, f32, 4096, True)        # > 2048 seeds
    state2 = torch.get_rng_state()
    m = model(5)                                     # (building a model draws; head_why_not itself must not)
    torch.set_rng_state(state2)
    E.head_why_not(m, L.classification, i64, 512, True)
    assert torch.equal(torch.get_rng_state(), state2), "head_why_not consumed torch's generator (the run's own)"
    torch.set_rng_state(state)


def test_train_main_takes_a_problem_held_in_memory(capsys):
    """NodeProblem.from_arrays + train.main(problem=...): the reference's loop on arrays already in memory (what
    bench.py's CLI measurements use instead of writing a multi-GB problem file) -- host mode, same JSON protocol."""
    from scipy import sparse
    rng = np.random.RandomState(0)
    n, D, C = 120, 8, 3
    degs = rng.randint(1, 6, size=n + 1)
    degs[0] = 0
    rows = np.repeat(np.arange(n + 1), degs)
    cols = np.concatenate([np.arange(d) for d in degs])
    adj = sparse.csr_matrix((rng.randint(1, n + 1, size=rows.shape[0]), (rows, cols)))
    feats = rng.normal(size=(n + 1, D)).astype(np.float32)
    feats[0] = 0
    folds = np.array(["train"] * 80 + ["val"] * 25 + ["test"] * (n + 1 - 105))
    folds[0] = "dummy"
    prob = gs.NodeProblem.from_arrays("classification", C, adj, adj, feats, folds, feats[:, :C].argmax(1).reshape(-1, 1),
                                      cuda=False)
    assert prob.feats_dim == D and prob.n_nodes == adj.shape[0] and prob.nodes["train"].shape[0] == 79
    train = __import__("importlib").import_module("pytorch-graphsage_amd.train")
    train.main(["--problem-path", "<memory>", "--no-cuda", "--epochs", "2", "--batch-size", "32", "--sampler-class",
                "sparse_uniform_neighbor_sampler", "--n-train-samples", "3,2", "--n-val-samples", "3,2",
                "--output-dims", "8,8"], problem=prob)
    lines = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    per_batch = [l for l in lines if "epoch_progress" in l]
    assert len(per_batch) == 2 * (79 // 32 + 1) and per_batch[-1]["val_metric"] is not None
