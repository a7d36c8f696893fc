"""-m gpu: ABI 4 -- collectives and host calls as command-list nodes, the deterministic (sorted) gradient of a trainable
embedding table, the sharded L1 head; the data-parallel step as ONE list with a 1-rank RCCL group on the box's GPU
(two ranks: tests/test_gpu_dist.py over gloo)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, pkg

pytestmark = pytest.mark.gpu
gs = pkg()
ops, nat = gs.ops, gs._native
DEV = "cuda"


@pytest.fixture(autouse=True)
def _setup():
    ops.set_compute_dtype("bf16")
    ops.warmup(torch.device(DEV))
    yield
    ops.set_compute_dtype("bf16")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _sort(ids, tail_id, n_tail, key_bits):
    lib = nat.lib()
    n = ids.numel() + n_tail
    sids = torch.zeros(n, dtype=torch.int64, device=DEV)
    spos = torch.zeros(n, dtype=torch.int32, device=DEV)
    nb = lib.gsage_sort_rows_temp_bytes(n, key_bits)
    assert nb > 0
    temp = torch.zeros(nb, dtype=torch.uint8, device=DEV)
    nat.check(lib.gsage_sort_rows(ids.data_ptr(), ids.numel(), tail_id, n_tail, key_bits, sids.data_ptr(),
                                  spos.data_ptr(), temp.data_ptr(), nb, _stream()), "sort_rows")
    return sids, spos, temp


@pytest.mark.parametrize("n0,n_tail,n_rows,E", [(5000, 16, 700, 64), (164_000, 32, 1_632_805, 64), (1, 0, 9, 8),
                                                (3000, 7, 40, 200), (0, 5, 100, 64), (2048, 0, 257, 8), (2049, 3, 70_000, 8),
                                                (1_310_720, 128, 1_632_805, 8)])
def test_sort_rows_is_stable_and_segment_sums_are_the_in_order_sums(n0, n_tail, n_rows, E):
    """gsage_sort_rows: the library's stable radix sort over (id, position) (round 6; rocPRIM's until then: the same
    contract, the same test) -- ids ascending, positions ascending within equal
    ids (stable), tail entries (one spare row) behind the frontier's.  gsage_segment_sum_rows: table[id] = scale *
    (rows of the run added IN LIST ORDER): compared bit for bit with a sequential fp32 sum on the host for the
    longest runs, with a float64 index_add for all rows; rows in no run are not touched; two launches agree bit
    for bit (no atomics)."""
    lib = nat.lib()
    gen = torch.Generator().manual_seed(n0 + E)
    hot = max(2, n_rows // 50)
    ids = torch.where(torch.rand(n0, generator=gen) < 0.5, torch.randint(1, hot, (n0,), generator=gen),
                      torch.randint(1, n_rows - 1, (n0,), generator=gen)).to(DEV)
    tail_id = n_rows - 1
    key_bits = max(1, int(n_rows - 1).bit_length())
    sids, spos, _temp = _sort(ids, tail_id, n_tail, key_bits)
    keys = torch.cat([ids, torch.full((n_tail,), tail_id, dtype=torch.int64, device=DEV)])
    ref_k, ref_p = torch.sort(keys, stable=True)
    assert torch.equal(sids, ref_k) and torch.equal(spos.long(), ref_p)

    rows0 = torch.randn(max(n0, 1), E, generator=gen).to(DEV)
    rows1 = torch.randn(max(n_tail, 1), E, generator=gen).to(DEV)
    table = torch.full((n_rows, E), 7.0, device=DEV)
    scale = 0.5
    n = n0 + n_tail

    def seg(out):
        nat.check(lib.gsage_segment_sum_rows(sids.data_ptr(), spos.data_ptr(), n, rows0.data_ptr(), E, n0,
                                             rows1.data_ptr(), E, E, scale, out.data_ptr(), E, _stream()), "segsum")
    seg(table)
    allrows = torch.cat([rows0[:n0], rows1[:n_tail]])
    ref = torch.zeros(n_rows, E, dtype=torch.float64, device=DEV).index_add_(0, keys, allrows.double()) * scale
    touched = torch.zeros(n_rows, dtype=torch.bool, device=DEV)
    touched[keys] = True
    assert torch.equal(table[~touched], torch.full_like(table[~touched], 7.0))          # stores only where a run is
    uniq, counts = torch.unique(keys, return_counts=True)
    # (fp32 sums of up to max(counts) terms against float64: the bound grows with the run; bit-exactness is below)
    assert torch.allclose(table[touched].double(), ref[touched], rtol=1e-5, atol=1e-5 + 2e-6 * int(counts.max()))
    # the order of additions is the list's: sequential fp32 sums on the host for the longest runs
    rows_cpu, keys_cpu = allrows.cpu().numpy(), keys.cpu().numpy()
    for r in uniq[torch.argsort(counts, descending=True)[:3]].tolist():
        acc = None
        for p in np.nonzero(keys_cpu == r)[0]:
            acc = rows_cpu[p].copy() if acc is None else (acc + rows_cpu[p]).astype(np.float32)
        assert np.array_equal(table[r].cpu().numpy(), (acc * np.float32(scale)).astype(np.float32)), r
    again = torch.full((n_rows, E), 7.0, device=DEV)
    seg(again)
    assert torch.equal(again, table)


@pytest.mark.parametrize("key_bits,n", [(31, 40_000), (8, 5000), (9, 70_001), (63, 3000)])
def test_sort_rows_over_every_key_width(key_bits, n):
    """the passes' ping-pong ends in the output pair for one .. eight passes; keys up to 2^key_bits - 1"""
    gen = torch.Generator().manual_seed(key_bits)
    top = 2 ** min(key_bits, 62) - 1
    ids = torch.randint(0, top, (n,), generator=gen, dtype=torch.int64).to(DEV)
    ids[::7] = ids[0]                                   # duplicates: stability
    sids, spos, _t = _sort(ids, int(ids[0]), 3, key_bits)
    keys = torch.cat([ids, ids[:1].repeat(3)])
    ref_k, ref_p = torch.sort(keys, stable=True)
    assert torch.equal(sids, ref_k) and torch.equal(spos.long(), ref_p)


def test_sort_and_segment_sum_replay_from_a_command_list():
    """The sort (round 6: the library's own kernels, three launches per 8-bit pass) and the segment sum are kernel nodes
    of a command list: a replay on fresh ids equals the direct calls."""
    lib = nat.lib()
    n0, n_rows, E = 20_000, 3000, 64
    gen = torch.Generator().manual_seed(1)
    ids = torch.randint(1, n_rows, (n0,), generator=gen).to(DEV)
    rows = torch.randn(n0, E, generator=gen).to(DEV)
    spare = torch.randn(4, E, generator=gen).to(DEV)
    n = n0 + 4
    sids = torch.zeros(n, dtype=torch.int64, device=DEV)
    spos = torch.zeros(n, dtype=torch.int32, device=DEV)
    nb = lib.gsage_sort_rows_temp_bytes(n, 12)
    temp = torch.zeros(nb, dtype=torch.uint8, device=DEV)
    table = torch.zeros(n_rows, E, device=DEV)
    with nat.CommandList.record() as cl:
        nat.check(lib.gsage_sort_rows(ids.data_ptr(), n0, n_rows - 1, 4, 12, sids.data_ptr(), spos.data_ptr(),
                                      temp.data_ptr(), nb, None), "sort_rows")
        nat.check(lib.gsage_segment_sum_rows(sids.data_ptr(), spos.data_ptr(), n, rows.data_ptr(), E, n0,
                                             spare.data_ptr(), E, E, 1.0, table.data_ptr(), E, None), "segsum")
    assert len(cl) == 2 * 3 + 1                     # (12 key bits: two passes of count / scan / place, + the segment sum)
    for trial in range(3):
        ids.copy_(torch.randint(1, n_rows, (n0,), generator=gen))
        table.zero_()
        cl.replay(_stream())
        keys = torch.cat([ids, torch.full((4,), n_rows - 1, dtype=torch.int64, device=DEV)])
        ref = torch.zeros(n_rows, E, dtype=torch.float64, device=DEV).index_add_(0, keys, torch.cat([rows, spare]).double())
        assert torch.allclose(table.double(), ref, rtol=1e-5, atol=1e-5), trial


def test_sorted_row_lists_equal_the_stamped_lists_and_are_deterministic():
    """gsage_rows_* over a SORTED list (gsage_row_adam.sorted_ids: runs settled by comparing neighbours, no atomics)
    against the same list unsorted (stamps + atomicMax): same table, exp_avg, exp_avg_sq bit for bit; the sorted
    norm partials are identical from launch to launch (the stamped ones need not be)."""
    from test_gpu_rows import _Rows
    lib, st = nat.lib(), _stream()
    n_rows, E = 5000, 64
    gen = torch.Generator().manual_seed(5)
    p0 = torch.randn(n_rows, E, generator=gen).to(DEV)
    a, b = _Rows(p0, E, 0.01, 0.7), _Rows(p0, E, 0.01, 0.7)
    b.d.sorted_ids = 1
    partials = []
    for t in range(1, 9):
        for r in (a, b):
            r.lr.fill_(0.01 + 0.001 * t)
        ids = torch.randint(0, 400 if t % 2 else n_rows, (3000,), generator=gen).to(DEV)
        sids, _pos, _tmp = _sort(ids, 0, 0, 13)
        uniq = torch.unique(ids)
        grad = torch.randn(uniq.shape[0], E, generator=gen).to(DEV) * (0.05 if t % 3 else 3.0)
        for r, lst in ((a, ids), (b, sids)):
            nat.check(lib.gsage_rows_catch_up(ctypes.byref(r.d), lst.data_ptr(), lst.numel(), None, 0, 0, st), "cu")
            r.g[uniq] = grad
            r.step += 1
            nat.check(lib.gsage_rows_sqnorm(ctypes.byref(r.d), lst.data_ptr(), lst.numel(), None, 0, 0,
                                            r.partial.data_ptr(), 64, st), "sq")
        first = b.partial.clone()
        nat.check(lib.gsage_rows_sqnorm(ctypes.byref(b.d), sids.data_ptr(), sids.numel(), None, 0, 0,
                                        b.partial.data_ptr(), 64, st), "sq")
        assert torch.equal(first, b.partial)                               # the order of the norm's terms is the list's
        want = float((grad.double() ** 2).sum())
        assert abs(float(b.partial.double().sum()) - want) <= 1e-5 * want
        partials.append((float(a.partial.sum()), float(b.partial.sum())))
        # (the two sides would clip with norms that differ in the last bits: give both the sorted side's partials)
        a.partial.copy_(b.partial)
        for r, lst in ((a, ids), (b, sids)):
            nat.check(lib.gsage_rows_adam(ctypes.byref(r.d), lst.data_ptr(), lst.numel(), None, 0, 0,
                                          r.partial.data_ptr(), 64, st), "ra")
    for r in (a, b):
        nat.check(lib.gsage_rows_catch_up_all(ctypes.byref(r.d), 0, st), "all")
    torch.cuda.synchronize()
    assert torch.equal(a.p, b.p) and torch.equal(a.m, b.m) and torch.equal(a.v, b.v)
    assert float(b.g.abs().max()) == 0.0 and int(b.last.min()) == 8
    rc = lib.gsage_rows_adam(ctypes.byref(b.d), ids.data_ptr(), 10, ids.data_ptr(), 10, 0, b.partial.data_ptr(), 64, st)
    assert rc == -1 and b"sorted" in lib.gsage_last_error()


def test_head_l1_shards_average_to_the_global_head():
    """gsage_head_l1_sharded on the two halves of a batch (each with the GLOBAL targets) against gsage_head_l1 on the
    whole batch (the reference's [B,1]-vs-[B] broadcast, problem.py:39-42): the mean over the shards of every
    partial sum [d fc.weight | d fc.bias | loss] equals the global one, and d E of a shard's rows is world x the
    global rows (the all-reduce's average undoes the factor)."""
    lib, st = nat.lib(), _stream()
    GB, D, W = 96, 256, 2
    B = GB // W
    gen = torch.Generator().manual_seed(11)
    Emb = torch.randn(GB, D, generator=gen).to(DEV)
    Wt = (torch.randn(D, generator=gen) / 4).to(DEV)
    bias = torch.tensor([0.3], device=DEV)
    tg = (torch.randn(GB, generator=gen) * 2).to(DEV)

    def run(E_rows, targets, T):
        b = E_rows.shape[0]
        preds = torch.zeros(b, 1, device=DEV)
        dE = torch.zeros(b, D, device=DEV)
        scratch = torch.zeros(lib.gsage_head_l1_scratch(b, D), device=DEV)
        if T:
            nat.check(lib.gsage_head_l1_sharded(E_rows.data_ptr(), D, Wt.data_ptr(), bias.data_ptr(), targets.data_ptr(), T,
                                                b, D, preds.data_ptr(), dE.data_ptr(), nat.F32, D, scratch.data_ptr(), st), "l1s")
        else:
            nat.check(lib.gsage_head_l1(E_rows.data_ptr(), D, Wt.data_ptr(), bias.data_ptr(), targets.data_ptr(), b, D,
                                        preds.data_ptr(), dE.data_ptr(), nat.F32, D, scratch.data_ptr(), st), "l1")
        n_wg = (b + 15) // 16
        return preds, dE, scratch[:n_wg * (D + 2)].view(n_wg, D + 2).sum(0)
    p_all, dE_all, part_all = run(Emb, tg, 0)
    # the global head against plain torch (the reference's call)
    e = Emb.clone().requires_grad_(True)
    pr = torch.nn.functional.normalize(e, dim=1) @ Wt[:, None] + bias
    loss = (pr - tg[None, :]).abs().mean()
    loss.backward()
    assert torch.allclose(p_all, pr.detach(), rtol=1e-5, atol=1e-5)
    assert abs(float(part_all[D + 1]) - float(loss)) <= 1e-5 * float(loss)
    assert torch.allclose(dE_all, e.grad, rtol=1e-4, atol=1e-7)
    acc = torch.zeros_like(part_all)
    for r in range(W):
        p_r, dE_r, part_r = run(Emb[r * B:(r + 1) * B].contiguous(), tg, GB)
        assert torch.equal(p_r, p_all[r * B:(r + 1) * B])
        assert torch.allclose(dE_r, W * dE_all[r * B:(r + 1) * B], rtol=1e-5, atol=1e-8)
        acc += part_r / W
    assert torch.allclose(acc, part_all, rtol=1e-5, atol=1e-7)


def test_host_call_nodes_run_in_list_order_on_the_replays_stream():
    """gsage_host_call inside a list: the callee is handed the replay's stream (the list's side stream inside a side
    section) and what it enqueues there runs between the neighbouring kernel nodes."""
    lib = nat.lib()
    ctr = torch.zeros(1, dtype=torch.int64, device=DEV)
    seen = torch.zeros(3, dtype=torch.int64, device=DEV)
    streams = []

    def probe(slot):
        def fn(s):
            streams.append(int(s or 0))
            with torch.cuda.stream(torch.cuda.ExternalStream(int(s))) if s else torch.cuda.stream(torch.cuda.current_stream()):
                seen[slot:slot + 1].copy_(ctr, non_blocking=True)
        return fn
    keep = []
    with nat.CommandList.record() as cl:
        nat.check(lib.gsage_counter_add(ctr.data_ptr(), 5, None), "counter_add")
        keep.append(nat.host_call(probe(0)))
        nat.check(lib.gsage_counter_add(ctr.data_ptr(), 7, None), "counter_add")
        nat.check(lib.gsage_cmdlist_side_begin(), "side_begin")
        keep.append(nat.host_call(probe(1)))
        nat.check(lib.gsage_cmdlist_side_end(), "side_end")
        nat.check(lib.gsage_cmdlist_join(), "join")
        nat.check(lib.gsage_counter_add(ctr.data_ptr(), 1, None), "counter_add")
        keep.append(nat.host_call(probe(2)))
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        cl.replay(side.cuda_stream)
    side.synchronize()
    torch.cuda.synchronize()
    assert seen.tolist() == [5, 12, 13]
    assert streams[0] == side.cuda_stream and streams[2] == side.cuda_stream and streams[1] not in (0, side.cuda_stream)

    def boom(_s):
        raise ValueError("expected failure of a host-call node")
    with nat.CommandList.record() as bad:
        keep.append(nat.host_call(boom))
    with pytest.raises(RuntimeError, match="host call failed"):
        bad.replay(_stream())


def _run_worker(name, env=None, timeout=600):
    e = dict(os.environ)
    e.update({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "GSAGE_FORCE_DDP": "1",
              "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    e["MASTER_PORT"] = str(s.getsockname()[1])
    s.close()
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_ddp1_worker.py"), name], env=e,
                         capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0 and "__OK__" in out.stdout, (out.stdout[-3000:], out.stderr[-3000:])
    return out.stdout


def test_native_rccl_communicator_with_one_rank():
    """gsage_comm_*: communicator set-up through torch.distributed's store, all-reduce / all-gather issued directly
    and as nodes of a command list's side section (a 1-rank RCCL group on the box's one GPU)."""
    _run_worker("comm")


@pytest.mark.parametrize("case", ["mean", "max_pool", "attention", "attention_emb_mae", "mean_emb"])
@pytest.mark.parametrize("overlap", ["0", "1"])
def test_one_rank_data_parallel_step_equals_the_plain_step(case, overlap):
    """The data-parallel step -- ONE command list whose collectives are RCCL calls issued by the library
    (gsage_comm_*), inline or on the list's side stream beside the next batch's gathers -- with a 1-rank group
    against the engine built without a process group: same predictions, same weights (the norm of the averaged
    gradient is summed in another order: 1e-6), and for the embedding engines the same table after settling."""
    out = _run_worker(case, {"GSAGE_DDP_OVERLAP": overlap})
    assert "native_comm=1" in out and "one_list=1" in out


# ------------------------------------------------------------------------------------------------------------
# the command behind the reference's only published number (utils/pokec.sh:11-13) through the fused engine
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", [0, 1])
@pytest.mark.parametrize("table", ["deferred", "sorted", "dense"])
@pytest.mark.parametrize("capture", [False, "cmdlist"])
def test_fp32_mean_embedding_engine_over_the_dense_sampler_replays_reference_train_steps(case, table, capture):
    """utils/pokec.sh:11-13 -- the reference's DEFAULT dense sampler + the trainable node-embedding prep (no
    features) + mean aggregators + regression_mae -- through FusedMeanTrainStep: two train steps of the reference
    (round4_kat q0 / q1) in fp32 from the recorded torch seed (the engine draws the permutations itself): the
    frontier, predictions, gradient norm, clipped gradients and every weight incl. every row of the embedding table
    after each step are the reference's."""
    from torch.nn import functional as F
    from conftest import load_golden
    from util import close, close_rel, close_update, weights
    os.environ.pop("GSAGE_DENSE_TABLE_ADAM", None)
    os.environ.pop("GSAGE_SORTED_ROWS", None)
    if table == "dense":
        os.environ["GSAGE_DENSE_TABLE_ADAM"] = "1"
    if table == "sorted":             # the table's gradient by sort + segment sums (what data-parallel runs use)
        os.environ["GSAGE_SORTED_ROWS"] = "1"
    try:
        g = load_golden("round4_kat.npz")
        p = "q%d_" % case
        ops.set_compute_dtype("fp32")
        fan, dims = [int(v) for v in g[p + "fanouts"]], [int(v) for v in g[p + "out_dims"]]
        adj, tadj = torch.from_numpy(g[p + "adj"]).to(DEV), torch.from_numpy(g[p + "tadj"]).to(DEV)
        specs = [{"n_train_samples": f, "n_val_samples": f, "output_dim": h,
                  "activation": (lambda x: x) if i == len(fan) - 1 else F.relu} for i, (f, h) in enumerate(zip(fan, dims))]
        wd = float(g[p + "weight_decay"])
        model = gs.GSSupervised(sampler_class=gs.sampler_lookup["uniform_neighbor_sampler"], adj=adj, train_adj=tadj,
                                prep_class=gs.prep_lookup["node_embedding"], aggregator_class=gs.aggregator_lookup["mean"],
                                input_dim=None, n_nodes=adj.shape[0], n_classes=1, layer_specs=specs, lr_init=0.01,
                                weight_decay=wd)
        model.load_state_dict(weights(g, p + "w0_"))
        model = model.to(DEV)
        model.optimizer = torch.optim.Adam(model.parameters(), lr=model.lr, weight_decay=wd)
        w0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        ids = torch.from_numpy(g[p + "ids"]).to(DEV)
        tg = torch.from_numpy(g[p + "targets"]).to(DEV)
        cls = gs.engine.fused_engine_for(model, None)
        assert cls is gs.engine.FusedMeanTrainStep
        eng = cls(model, None, gs.ProblemLosses.regression_mae, ids, tg, capture=capture)
        assert eng.draws == "dense" and eng.emb and eng.fused_l1 and eng.lazy_rows == (table != "dense")
        assert eng.tdt == torch.float32 and (not eng.lazy_rows or eng.sorted_rows == (table == "sorted"))
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
                    if k != "prep.embedding.weight":         # the table's gradient is consumed (zeroed) by the step
                        close_rel(v.grad.cpu().numpy(), g[p + "s0_cg_%s" % k], (step, "clipped grad", k), 2e-4)
            for k, v in model.state_dict().items():          # (state_dict settles the deferred rows)
                close_update(v.detach().cpu().numpy(), g[p + "w%d_%s" % (step + 1, k)], w0[k].numpy(), (step, "weights", k))
            assert float(eng._grad_slice(eng.table).abs().max()) == 0.0
        model.train_sampler.table(DEV).check()
    finally:
        os.environ.pop("GSAGE_DENSE_TABLE_ADAM", None)
        os.environ.pop("GSAGE_SORTED_ROWS", None)


# ------------------------------------------------------------------------------------------------------------
# train.py decides what an engine covers BEFORE building it (round-3 advisor finding, severity high)
# ------------------------------------------------------------------------------------------------------------
def _multilabel_problem(tmp_path, n_train):
    from scipy import sparse
    rng = np.random.RandomState(0)
    n, D, C = 900, 12, 4
    degs = rng.randint(1, 12, size=n + 1)
    degs[0] = 0
    rows = np.repeat(np.arange(n + 1), degs)
    cols = np.concatenate([np.arange(d) for d in degs])
    adj = sparse.csr_matrix((rng.randint(1, n + 1, size=rows.shape[0]), (rows, cols)))
    feats = rng.normal(size=(n + 1, D)).astype(np.float32)
    feats[0] = 0
    folds = np.array(["train"] * n_train + ["val"] * 150 + ["test"] * (n + 1 - n_train - 150))
    folds[0] = "dummy"
    targets = (feats[:, :C] > 0).astype(np.float32)                 # utils/run-convert.sh's task: one bit per class
    path = os.path.join(str(tmp_path), "ml-problem.npz")
    gs.problem.save_problem_npz(path, {"task": "multilabel_classification", "n_classes": C, "feats": feats,
                                       "folds": folds, "targets": targets, "sparse": True, "adj": adj, "train_adj": adj})
    return path


def test_default_cli_run_of_a_multilabel_problem_trains(tmp_path, capsys):
    """multilabel_classification (the task of the reference's utils/run-convert.sh) has no fused head; the
    reference's unequal array_split chunks would have to be padded, which only a fused head can ignore.  The default
    --engine auto must therefore take the module path BEFORE an engine has re-pointed the Parameters (and say why on
    stderr), and train; --engine fused on the same problem is an error with the same sentence.  With chunks of one
    size nothing is padded and the engine runs (stock torch ops for the head inside the captured step)."""
    import importlib
    import json
    train = importlib.import_module("pytorch-graphsage_amd.train")
    argv = ["--aggregator-class", "mean", "--sampler-class", "sparse_uniform_neighbor_sampler", "--epochs", "4",
            "--n-train-samples", "5,3", "--n-val-samples", "5,3"]
    path = _multilabel_problem(tmp_path, 702)                       # 701 training nodes -> chunks of 351 and 350
    step = train.main(["--problem-path", path] + argv)
    cap = capsys.readouterr()
    assert step is None and "has no fused kernel" in cap.err and "using the module path" in cap.err
    out = [json.loads(l) for l in cap.out.strip().split("\n") if l.startswith("{")]
    logged = [o for o in out if "epoch_progress" in o]
    assert len(logged) == 4 * 2 and set(out[-1]) == {"epoch", "train_metric", "val_metric", "time"}
    assert out[-1]["train_metric"]["micro"] > logged[0]["train_metric"]["micro"]
    with pytest.raises(SystemExit, match="has no fused kernel"):
        train.main(["--problem-path", path, "--engine", "fused"] + argv)
    capsys.readouterr()
    path = _multilabel_problem(tmp_path, 701)                       # 700 training nodes, two chunks of 350: no padding
    step = train.main(["--problem-path", path] + argv)
    cap = capsys.readouterr()
    assert step is not None and type(step).__name__ == "FusedMeanTrainStep" and not step.fused_head
    out = [json.loads(l) for l in cap.out.strip().split("\n") if l.startswith("{")]
    assert out[-1]["train_metric"]["micro"] > 0.5


def test_one_launch_batch_metric_and_metric_ring_equal_the_reference_route():
    """A training batch (B x C <= 64 k) is scored by ONE single-workgroup launch, larger inputs by the three-launch
    route: both equal the reference's host route (ProblemMetrics: sklearn) on the same data, for classification and
    both multilabel target types; MetricRing (results written into a device ring, read back a ring at a time) returns
    the same values, in order, as the synchronous calls."""
    gen = torch.Generator().manual_seed(3)
    for B, C in ((512, 41), (4096, 41), (37, 7), (300, 500), (256, 64)):
        logits = torch.randn(B, C, generator=gen).to(DEV)
        y = torch.randint(0, C, (B,), generator=gen).to(DEV)
        before = nat.launch_count()
        dev = gs.DeviceMetrics.classification(y, logits)
        assert nat.launch_count() - before == (1 if B * C <= 64 * 1024 else 3), (B, C)
        host = gs.ProblemMetrics.classification(y.cpu().numpy(), logits.cpu().numpy())
        assert abs(dev["micro"] - host["micro"]) < 1e-9 and abs(dev["macro"] - host["macro"]) < 1e-9, (B, C)
        ring = gs.problem.MetricRing("classification", torch.device(DEV), capacity=4)
        want = []
        for k in range(4):
            lg = logits.roll(k, dims=1).contiguous()
            ring.score(y.view(B, 1), lg)
            want.append(gs.DeviceMetrics.classification(y, lg))
        assert ring.results() == want and ring.results() == []
        for cast in (torch.float32, torch.int64):
            ym = (torch.rand(B, C, generator=gen) < 0.3).to(cast).to(DEV)
            dev = gs.DeviceMetrics.multilabel_classification(ym, logits)
            host = gs.ProblemMetrics.multilabel_classification(ym.cpu().numpy(), logits.cpu().numpy())
            assert abs(dev["micro"] - host["micro"]) < 1e-9 and abs(dev["macro"] - host["macro"]) < 1e-9, (B, C, cast)
            mr = gs.problem.MetricRing("multilabel_classification", torch.device(DEV))
            mr.score(ym, logits)
            assert mr.results() == [dev]
    a, b = torch.randn(777, 1, generator=gen).to(DEV), torch.randn(777, 1, generator=gen).to(DEV)
    mr = gs.problem.MetricRing("regression_mae", torch.device(DEV))
    mr.score(a, b)
    assert mr.results() == [gs.DeviceMetrics.regression_mae(a, b)]


@pytest.mark.parametrize("spw", ["2", "4"])
def test_fused_sampler_with_several_seeds_per_workgroup(spw):
    """GSAGE_HOPS_SPW (seeds per workgroup of the fused multi-hop sampler; read once per process): the frontier does
    not depend on it -- the reference-stream test of tests/test_gpu_round3.py (frontiers equal to the reference's
    golden ones) and the queue-mode tests of tests/test_gpu_engine.py (the sampler as a role of the gather launch
    against the stand-alone launch) again, in a process that sets it."""
    env = dict(os.environ, GSAGE_HOPS_SPW=spw)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_round3.py"),
                        os.path.join(ROOT, "tests", "test_gpu_engine.py"), "-q", "-x", "-p", "no:cacheprovider", "-m", "gpu",
                        "-k", "fused_sampler_consumes or queue"], env=env, cwd=ROOT, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
