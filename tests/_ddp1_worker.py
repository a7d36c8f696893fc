"""Worker of tests/test_gpu_round4.py: ONE rank with a real RCCL group on the box's GPU (GSAGE_FORCE_DDP=1), so that
the library's own communicator (gsage_comm_*) and the one-list data-parallel step run on hardware; the process group
lives and dies with this process.  usage: _ddp1_worker.py comm | <tests/util.py DP_CASES name>"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from conftest import pkg          # noqa: E402
import util                       # noqa: E402

gs = pkg()
nat, ops = gs._native, gs.ops
DEV = "cuda"


def comm_case(ddp):
    lib = nat.lib()
    comm = ddp.comm
    assert comm is not None and comm.world == 1
    st = torch.cuda.current_stream().cuda_stream
    x = torch.arange(1000, dtype=torch.float32, device=DEV) / 7
    want = x.clone()
    comm.all_reduce(x, True, st)
    send = torch.arange(64, dtype=torch.int64, device=DEV) * 3
    recv = torch.zeros(64, dtype=torch.int64, device=DEV)
    comm.group(True, st)
    comm.all_gather(send, recv, st)
    comm.all_reduce(x, False, st)
    comm.group(False, st)
    torch.cuda.synchronize()
    assert torch.equal(x, want) and torch.equal(recv, send)
    # as nodes of a list, on its side stream, between two kernel nodes that write what the collective reads / read
    # what it wrote
    ctr = torch.zeros(2, dtype=torch.int64, device=DEV)
    got = torch.zeros(2, dtype=torch.int64, device=DEV)
    with nat.CommandList.record() as cl:
        nat.check(lib.gsage_counter_add(ctr.data_ptr(), 3, None), "counter_add")
        nat.check(lib.gsage_cmdlist_side_begin(), "side_begin")
        comm.all_gather(ctr, got, None)
        nat.check(lib.gsage_cmdlist_side_end(), "side_end")
        nat.check(lib.gsage_counter_add(ctr[1:].data_ptr(), 1, None), "counter_add")
        nat.check(lib.gsage_cmdlist_join(), "join")
        nat.check(lib.gsage_counter_add(got.data_ptr(), 100, None), "counter_add")
    for k in range(1, 4):
        cl.replay(st)
        torch.cuda.synchronize()
        assert int(got[0]) == 3 * k + 100, (k, got.tolist())
    print("native_comm=1 one_list=1")


def engine_case(case, ddp):
    res = {}
    for name, handle in (("plain", None), ("ddp", ddp)):
        model, feats, loss_fn, ids, tg, prec = util.dp_case(gs, case, global_batch=16)
        if handle is not None:
            gs.dist.attach(model, handle, seed=77)
        res[name] = util.dp_run(gs, case, model, feats, loss_fn, ids, tg, prec, handle, steps=5)
    (p0, w0, e0), (p1, w1, e1) = res["plain"], res["ddp"]
    assert e1.ddp is not None and e0.ddp is None
    scale_p, scale_w = float(p0.abs().max()), float(w0.abs().max())
    ep, ew = float((p1 - p0).abs().max()), float((w1 - w0).abs().max())
    print("case %s: |d preds| %.3e of %.3e, |d w| %.3e of %.3e" % (case, ep, scale_p, ew, scale_w))
    # fp32 storage: only the order in which the norm of the (averaged) gradient is summed differs; bf16 storage: a
    # last-bit difference can flip the rounding of an operand copy
    tol = 2e-5 if util.DP_CASES[case][5] == "fp32" else 3e-3
    assert ep <= tol * scale_p + 1e-6 and ew <= tol * scale_w + 1e-6, (ep, ew)
    if e1.emb:
        assert e1.lazy_rows and e1.sorted_rows
    print("native_comm=%d one_list=%d" % (int(e1.comm is not None), int(e1._one_list_ddp())))


def main():
    what = sys.argv[1]
    ddp = gs.dist.init_from_env(cuda=True)
    assert ddp is not None and ddp.world == 1, "run with GSAGE_FORCE_DDP=1 RANK=0 WORLD_SIZE=1"
    ops.warmup(torch.device(DEV))
    try:
        if what == "comm":
            comm_case(ddp)
        else:
            engine_case(what, ddp)
    finally:
        ddp.close()
    print("__OK__")


if __name__ == "__main__":
    main()
