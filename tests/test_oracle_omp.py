"""The OpenMP C restatement of train_step (oracle/gsage_train_omp.c, bench.py's CPU baseline) against
the fixtures generated from the reference: 2- and 3-layer mean-aggregator cases, two steps each."""
import numpy as np

from conftest import load_golden
from util import close_update
from oracle import cpu as ocpu


def _run(g, p, prefix_sel):
    fan = [int(v) for v in g[p + "fanouts"]]
    w0 = {k[len(p + "w0_"):]: g[k] for k in g.files if k.startswith(p + "w0_")}
    tr = ocpu.MeanTrainerOMP(w0, fan, weight_decay=float(g[p + "weight_decay"]))
    feats = np.ascontiguousarray(g[p + "feats"], dtype=np.float32)
    out = []
    for step in range(2):
        sels = [g[p + "s%d_sel%d" % (step, h)] for h in range(len(fan))]
        out.append(tr.step(float(g[p + "lr%d" % step]), g[p + "ids"], feats, g[p + "targets"],
                           g[p + "tadj_indptr"], g[p + "tadj_data"], sels))
        out[-1]["clipped"] = {k: v.copy() for k, v in out[-1]["clipped"].items()}
    return tr, out


def _check(g, p, tr, out, w_final_prefix):
    for step, r in enumerate(out):
        ref = g[p + "s%d_preds" % step]
        assert np.abs(r["preds"] - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max()), (p, step)
        assert abs(r["loss"] - float(g[p + "s%d_loss" % step])) < 1e-4
        gn = float(g[p + "s%d_gradnorm" % step])
        assert abs(r["gradnorm"] - gn) <= 2e-4 * max(1.0, gn)
        for k, v in r["clipped"].items():
            key = p + "s%d_cg_%s" % (step, k)
            if key in g.files:
                assert np.abs(v - g[key]).max() <= 2e-4 * np.abs(g[key]).max() + 1e-9, (p, step, k)
    for k, v in tr.weights().items():
        close_update(v, g[w_final_prefix + k], g[p + "w0_" + k], (p, k))


def test_omp_train_step_on_module_fixture():
    g = load_golden("model_kat.npz")
    for c in (0, 1):                                   # the two mean / identity / classification cases
        p = "c%d_" % c
        assert [str(s) for s in g[p + "cfg"]][:3] == ["mean", "identity", "classification"]
        tr, out = _run(g, p, None)
        _check(g, p, tr, out, p + "w2_")


def test_omp_train_step_on_engine_fixture_incl_three_layers():
    g = load_golden("engine_kat.npz")
    seen3 = 0
    for c in range(int(g["n_cases"])):
        p = "e%d_" % c
        if str(g[p + "cfg"][0]) != "mean":
            continue
        seen3 += len(g[p + "fanouts"]) == 3
        tr, out = _run(g, p, None)
        _check(g, p, tr, out, p + "w2_")
    assert seen3 == 2
    assert ocpu.omp_threads() >= 1
