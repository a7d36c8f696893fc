"""-m gpu: the drop-in boundary end to end on the MI355X -- aggregator / prep / full-model golden
vectors through the HIP path (fp32 mode: tight; bf16 mode: loose), plus size-independent
properties at BASELINE.json's full Reddit shapes where the oracle is too slow to run."""
import numpy as np
import pytest
import torch

from conftest import load_golden, pkg
from util import ACTS, SelReplay, build_model, close, close_fro, csr_of, weights

pytestmark = pytest.mark.gpu
gs = pkg()
ops = gs.ops
nat = gs._native
DEV = "cuda"


@pytest.fixture(autouse=True)
def _warm():
    ops.warmup(torch.device(DEV))
    yield
    ops.set_compute_dtype("bf16")


TOL = {"fp32": (1e-4, 1e-5), "bf16": (4e-2, 4e-2)}


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_aggregator_golden_vectors_on_gpu(dtype):
    ops.set_compute_dtype(dtype)
    g = load_golden("agg_kat.npz")
    before = nat.launch_count()
    for c in range(int(g["n_cases"])):
        p = "c%d_" % c
        name, act = str(g[p + "name"]), str(g[p + "act"])
        M, n, D, h = [int(v) for v in g[p + "dims"]]
        agg = gs.aggregator_lookup[name](input_dim=D, output_dim=h, activation=ACTS[act])
        agg.load_state_dict(weights(g, p + "w_"))
        agg = agg.to(DEV)
        x = torch.from_numpy(g[p + "x"].copy()).to(DEV).requires_grad_(True)
        nb = torch.from_numpy(g[p + "neibs"].copy()).to(DEV).requires_grad_(True)
        out = agg(x, nb)
        assert out.shape == g[p + "out"].shape
        close(out.detach().float().cpu().numpy(), g[p + "out"], (c, name, "out"), *TOL[dtype])
        (out.float() * torch.from_numpy(g[p + "G"]).to(DEV)).sum().backward()
        if dtype == "fp32":
            chk = lambda a, b, w: close(a, b, w, *TOL[dtype])
        else:           # ReLU-mask flips of ~0 pre-activations: norm-wise bound (util.close_fro)
            chk = lambda a, b, w: close_fro(a, b, w, 0.15)
        chk(x.grad.cpu().numpy(), g[p + "dx"], (c, name, "dx"))
        chk(nb.grad.cpu().numpy(), g[p + "dneibs"], (c, name, "dneibs"))
        for k, v in agg.named_parameters():
            chk(v.grad.cpu().numpy(), g[p + "g_" + k], (c, name, k))
    assert nat.launch_count() - before >= 2 * int(g["n_cases"])


def test_lstm_aggregator_golden_vectors_on_gpu():
    """LSTMAggregator on the GPU (MIOpen recurrence + K5 projection, fp32 compute mode) against the
    vectors recorded from the reference."""
    ops.set_compute_dtype("fp32")
    g = load_golden("lstm_kat.npz")
    for c in range(int(g["n_cases"])):
        p = "c%d_" % c
        M, n, D, h, hid, bidir = [int(v) for v in g[p + "dims"]]
        agg = gs.aggregator_lookup["lstm"](input_dim=D, output_dim=h, activation=ACTS[str(g[p + "act"])],
                                           hidden_dim=hid, bidirectional=bool(bidir))
        agg.load_state_dict(weights(g, p + "w_"))
        agg = agg.to(DEV)
        x = torch.from_numpy(g[p + "x"].copy()).to(DEV).requires_grad_(True)
        nb = torch.from_numpy(g[p + "neibs"].copy()).to(DEV).requires_grad_(True)
        out = agg(x, nb)
        close(out.detach().float().cpu().numpy(), g[p + "out"], (c, "lstm out"), 2e-4, 2e-5)
        (out.float() * torch.from_numpy(g[p + "G"]).to(DEV)).sum().backward()
        close(x.grad.cpu().numpy(), g[p + "dx"], (c, "dx"), 2e-4, 2e-5)
        close(nb.grad.cpu().numpy(), g[p + "dneibs"], (c, "dneibs"), 2e-4, 2e-5)
        for k, v in agg.named_parameters():
            close(v.grad.cpu().numpy(), g[p + "g_" + k], (c, k), 2e-4, 2e-5)


def test_aggregator_rowref_equals_tensor_path():
    """feats[ids] as a lazy RowRef (fused gather) must equal the materialised-tensor route."""
    ops.set_compute_dtype("fp32")
    rng = np.random.RandomState(0)
    R, D, h, M, n = 300, 50, 16, 40, 7
    feats = torch.from_numpy(rng.normal(size=(R, D)).astype(np.float32)).to(DEV)
    store = gs.FeatureStore.from_array(feats.cpu().numpy(), torch.device(DEV), dtype="fp32")
    idx = torch.from_numpy(rng.randint(0, R, size=M)).to(DEV)
    idn = torch.from_numpy(rng.randint(0, R, size=M * n)).to(DEV)
    for name in ("mean", "max_pool", "mean_pool", "attention"):
        torch.manual_seed(1)
        agg = gs.aggregator_lookup[name](input_dim=D, output_dim=h, activation=torch.relu).to(DEV)
        a = agg(store[idx], store[idn])
        b = agg(feats[idx], feats[idn])
        close(a.detach().float().cpu().numpy(), b.detach().float().cpu().numpy(), name, 1e-5, 1e-6)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_full_model_golden_train_steps_on_gpu(dtype):
    ops.set_compute_dtype(dtype)
    g = load_golden("model_kat.npz")
    rtol, atol = TOL[dtype]
    for c in range(int(g["n_cases"])):
        p = "c%d_" % c
        model, feats, task = build_model(gs, g, p, device=DEV)
        loss_fn = getattr(gs.ProblemLosses, task)
        ids = torch.from_numpy(g[p + "ids"]).to(DEV)
        tg = torch.from_numpy(g[p + "targets"]).to(DEV)
        with SelReplay([g[p + "eval_sel0"], g[p + "eval_sel1"]]):
            ev = model(ids, feats, train=False)
        close(ev.detach().cpu().numpy(), g[p + "eval_preds"], (c, "eval"), rtol, atol)
        for step in range(2):
            model.set_progress(0.25 * step)
            with SelReplay([g[p + "s%d_sel0" % step], g[p + "s%d_sel1" % step]]):
                preds = model.train_step(ids=ids, feats=feats, targets=tg, loss_fn=loss_fn)
            close(preds.detach().cpu().numpy(), g[p + "s%d_preds" % step], (c, step, "preds"), rtol, atol)
            if dtype == "fp32":
                for k, v in model.named_parameters():
                    close(v.grad.cpu().numpy(), g[p + "s%d_cg_%s" % (step, k)], (c, step, "cg", k), 2e-4, 2e-5)
                for k, v in model.state_dict().items():
                    close(v.cpu().numpy(), g[p + "w%d_%s" % (step + 1, k)], (c, step, "w", k), 2e-4, 2e-5)
        model.train_sampler.csr(DEV).check()


def test_stream_kat_on_gpu_compat_mode_is_bit_identical():
    """Same seed -> same epoch shuffle AND same sampled frontier as the reference (level 2)."""
    g = load_golden("stream_kat.npz")
    adj = csr_of(g, "g_")
    s = gs.sampler_lookup["sparse_uniform_neighbor_sampler"](adj=adj, rng="compat")
    nodes = g["nodes"]
    gs.set_seeds(int(g["seed"]) ** 2)
    order = np.random.permutation(np.arange(nodes.shape[0]))
    for b, chunk in enumerate(np.array_split(order, nodes.shape[0] // 64 + 1)):
        ids = torch.from_numpy(nodes[chunk]).to(DEV)
        assert np.array_equal(ids.cpu().numpy(), g["b%d_ids" % b])
        h1 = s(ids, n_samples=5)
        h2 = s(h1, n_samples=3)
        assert np.array_equal(h1.cpu().numpy(), g["b%d_h1" % b])
        assert np.array_equal(h2.cpu().numpy(), g["b%d_h2" % b])
    assert np.array_equal(np.random.randint(0, 2 ** 31 - 1, size=4), g["tail"])


# ------------------------------------------------------------------------ full-size properties
def _reddit_like(n_nodes=232965, seed=0):
    import bench
    return bench.synthetic_reddit(n_nodes=n_nodes, seed=seed)


@pytest.fixture(scope="module")
def reddit():
    """the bench's Reddit-shaped synthetic graph + features (built once: ~25 s of host time)"""
    import bench
    return bench.synthetic_reddit(seed=0)


def test_full_size_reddit_properties(reddit):
    """BASELINE config 2 shapes (N=232 965, D=602, B=512, fanout 25/10): sampled ids are real
    neighbours; gather+mean is linear in the table; means of a constant table are that constant."""
    ops.set_compute_dtype("bf16")
    data = reddit
    adj = data["adj"]
    s = gs.sampler_lookup["sparse_uniform_neighbor_sampler"](adj=adj, rng="philox", seed=1)
    ids0 = torch.from_numpy(np.random.RandomState(0).randint(1, adj.shape[0], size=512)).to(DEV)
    ids1 = s(ids0, n_samples=25)
    ids2 = s(ids1, n_samples=10)
    assert ids1.numel() == 12800 and ids2.numel() == 128000
    s.csr(DEV).check()
    # membership: every sample of parent p is one of p's neighbours (or 0 for an isolated parent)
    ip, dat = adj.indptr, adj.data
    par = ids1.cpu().numpy().repeat(10)
    ch = ids2.cpu().numpy()
    for k in np.random.RandomState(1).randint(0, ch.shape[0], size=2000):
        row = dat[ip[par[k]]:ip[par[k] + 1]]
        assert (ch[k] == 0 and row.size == 0) or ch[k] in row
    # determinism + counter sensitivity
    s2 = gs.sampler_lookup["sparse_uniform_neighbor_sampler"](adj=adj, rng="philox", seed=1)
    assert torch.equal(s2(ids0, n_samples=25), ids1)
    assert not torch.equal(s2(ids0, n_samples=25), ids1)            # next call index
    store = data["feats"](DEV, "bf16")
    a = ops.gather_mean(store, ids2, 12800, 10, out_dtype=torch.float32)
    ones = gs.FeatureStore(torch.ones_like(store.data), store.dim)
    assert torch.equal(ops.gather_mean(ones, ids2, 12800, 10, out_dtype=torch.float32),
                       torch.ones(12800, store.dim, device=DEV))
    dbl = gs.FeatureStore(store.data * 2, store.dim)
    assert torch.equal(ops.gather_mean(dbl, ids2, 12800, 10, out_dtype=torch.float32), a * 2)
    # spot-check 64 rows against the oracle
    from oracle import cpu as ocpu
    sub = ids2[:640].cpu().numpy()
    uniq, inv = np.unique(sub, return_inverse=True)
    small = store.data[torch.from_numpy(uniq).to(DEV), :store.dim].float().cpu().numpy()
    close(a[:64].cpu().numpy(), ocpu.gather_mean_f32(small, inv, 64, 10), "oracle spot", 1e-5, 1e-6)


@pytest.mark.parametrize("agg", ["mean", "max_pool"])
def test_full_size_fused_step_vs_eager_step(reddit, agg):
    """The exact bench workload (BASELINE configs[1] / [2] shapes: Reddit-sized graph, B = 512, fan-out
    25/10, hidden 128): one fused-engine step against one step of GSSupervised.train_step on the eager
    product path from the same weights and the same Philox samples; and size-independent invariants
    of the step -- the loss gradient w.r.t. the logits sums to zero per row, so d fc.bias sums to 0;
    gradient clipping bounds the applied update; a second run reproduces the first bit for bit."""
    import bench
    ops.set_compute_dtype("bf16")
    data = reddit
    dev = torch.device(DEV)
    store = data["feats"](dev, "bf16")
    rng = np.random.RandomState(3)
    pick = rng.randint(0, len(data["train_ids"]), size=(2, 512))
    ids = torch.from_numpy(data["train_ids"][pick]).to(dev)
    tg = torch.from_numpy(data["targets"][data["train_ids"][pick]]).to(dev).view(2, 512, 1)
    loss_fn = gs.ProblemLosses.classification

    def fresh():
        torch.manual_seed(11)                                  # identical initial weights every time
        m = bench.build_model(gs, data["adj"], aggregator=agg, rng="philox").to(dev)
        m.train_sampler.csr(dev)
        return m
    ref, mdl = fresh(), fresh()
    mdl.load_state_dict(ref.state_dict())
    eng = gs.engine.fused_engine_for(mdl, store)(mdl, store, loss_fn, ids[0], tg[0])
    w0 = eng.flat_p.clone()
    p_eng = eng(ids[0], tg[0]).float().cpu().numpy()
    p_ref = ref.train_step(ids=ids[0], feats=store, targets=tg[0], loss_fn=loss_fn).detach().float().cpu().numpy()
    close(p_eng, p_ref, "preds at full size", 3e-2, 3e-2)
    for (k, a), (_, b) in zip(mdl.named_parameters(), ref.named_parameters()):
        close_fro(a.grad.cpu().numpy(), b.grad.cpu().numpy(), ("grad", k), 0.1)
    # invariants
    gb = mdl.fc.bias.grad
    assert abs(float(gb.sum())) <= 1e-5 * float(gb.abs().sum() + 1e-12)
    assert float(eng.gnorm) > 0 and np.isfinite(float(eng.gnorm))
    clipped = float(torch.linalg.vector_norm(eng.flat_g))             # the bucket holds the CLIPPED gradient
    assert clipped <= 5.0 * (1 + 1e-4) and abs(clipped - min(float(eng.gnorm), 5.0)) <= 1e-3 * clipped + 1e-6
    assert float((eng.flat_p - w0).abs().max()) <= 0.01 * 1.001      # Adam's first step moves each weight by <= lr
    mdl.train_sampler.csr(dev).check()
    # determinism of the whole step (sampling, split reductions, Adam): an identical second engine
    # reproduces predictions and weights bit for bit
    outs = []
    for _ in range(2):
        m = fresh()
        e = gs.engine.fused_engine_for(m, store)(m, store, loss_fn, ids[0], tg[0])
        e.load_epoch(ids, tg)
        pr = torch.stack([e.step_queue().clone() for _ in range(3)])
        outs.append((pr, e.flat_p.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("agg,dims", [("mean", "128,128"), ("max_pool", "64,64")])
def test_train_cli_with_fused_engine(tmp_path, capsys, agg, dims):
    """train.py --engine fused: the reference's command line driving the fused engines through the
    device batch queue -- a learnable toy problem (class = arg-max feature of the node) must be learnt,
    and the stdout protocol (one JSON object per logged batch, a final one, train.py:151-170) holds."""
    import importlib
    import json
    import os
    from scipy import sparse
    rng = np.random.RandomState(0)
    n, D, C = 900, 12, 4
    degs = rng.randint(1, 12, size=n + 1)
    degs[0] = 0
    rows = np.repeat(np.arange(n + 1), degs)
    cols = np.concatenate([np.arange(d) for d in degs])
    vals = rng.randint(1, n + 1, size=rows.shape[0])
    adj = sparse.csr_matrix((vals, (rows, cols)))
    feats = rng.normal(size=(n + 1, D)).astype(np.float32)
    feats[0] = 0
    targets = feats[:, :C].argmax(1).reshape(-1, 1)
    folds = np.array(["train"] * 700 + ["val"] * 150 + ["test"] * 51)
    folds[0] = "dummy"
    path = os.path.join(str(tmp_path), "problem.npz")
    gs.problem.save_problem_npz(path, {"task": "classification", "n_classes": C, "feats": feats, "folds": folds,
                                       "targets": targets, "sparse": True, "adj": adj, "train_adj": adj})
    train = importlib.import_module("pytorch-graphsage_amd.train")
    train.main(["--problem-path", path, "--engine", "fused", "--rng", "philox", "--batch-size", "64",
                "--epochs", "6", "--lr-init", "0.01", "--sampler-class", "sparse_uniform_neighbor_sampler",
                "--aggregator-class", agg, "--n-train-samples", "5,3", "--n-val-samples", "5,3",
                "--output-dims", dims, "--log-interval", "4", "--show-test"])
    out = [json.loads(l) for l in capsys.readouterr().out.strip().split("\n") if l.startswith("{")]
    assert "test_f1" in out[-1] and set(out[-2]) == {"epoch", "train_metric", "val_metric", "time"}
    logged = [o for o in out if "epoch_progress" in o]
    assert len(logged) >= 6 * 3 and logged[0]["val_metric"] is None
    first, last = logged[0]["train_metric"]["micro"], out[-2]["train_metric"]["micro"]
    assert last > max(0.6, first + 0.2), (first, last)           # chance = 0.25
    assert out[-2]["val_metric"]["micro"] > 0.6
