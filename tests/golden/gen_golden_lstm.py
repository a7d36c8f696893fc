#!/usr/bin/env python
"""
gen_golden_lstm.py -- golden vectors for the LSTM aggregator (TEST INFRASTRUCTURE, build container
only; same rules and shims as gen_golden.py).  Drives the reference's LSTMAggregator
(nn_modules.py:259-286, uni- and bidirectional) on small seeded inputs, forward and backward, and
stores inputs / weights / outputs / gradients in the layout of agg_kat.npz.

    python -B tests/golden/gen_golden_lstm.py        # regenerates tests/golden/lstm_kat.npz
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REF = os.environ.get("GSAGE_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
for _m in ("h5py", "cPickle", "ujson"):
    sys.modules.setdefault(_m, types.ModuleType(_m))

import numpy as np
import torch
from torch.nn import functional as F

import nn_modules          # noqa: E402  (reference)

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(1)


def main():
    out = {}
    acts = {"relu": F.relu, "identity": (lambda x: x)}
    case = 0
    for (M, n, D, h, hid, bidir, act) in [(6, 5, 20, 8, 16, False, "relu"), (9, 10, 12, 6, 32, True, "identity"),
                                          (4, 1, 7, 5, 8, False, "relu"), (17, 3, 16, 32, 64, True, "relu")]:
        torch.manual_seed(3000 + case)
        agg = nn_modules.LSTMAggregator(input_dim=D, output_dim=h, activation=acts[act], hidden_dim=hid,
                                        bidirectional=bidir)
        x = torch.randn(M, D, requires_grad=True)
        neibs = torch.randn(M * n, D, requires_grad=True)
        res = agg(x, neibs)
        G = torch.randn_like(res)
        (res * G).sum().backward()
        p = "c%d_" % case
        out[p + "name"], out[p + "act"] = np.array("lstm"), np.array(act)
        out[p + "dims"] = np.array([M, n, D, h, hid, int(bidir)], dtype=np.int64)
        out[p + "output_dim"] = np.array(agg.output_dim)
        for k, v in agg.state_dict().items():
            out[p + "w_" + k] = v.detach().numpy().copy()
        out[p + "x"], out[p + "neibs"], out[p + "G"] = x.detach().numpy().copy(), neibs.detach().numpy().copy(), G.numpy().copy()
        out[p + "out"] = res.detach().numpy().copy()
        out[p + "dx"], out[p + "dneibs"] = x.grad.numpy().copy(), neibs.grad.numpy().copy()
        for k, v in agg.named_parameters():
            out[p + "g_" + k] = v.grad.numpy().copy()
        case += 1
    out["n_cases"] = np.array(case)
    np.savez_compressed(os.path.join(OUT, "lstm_kat.npz"), **out)
    print("lstm_kat: %d cases" % case)


if __name__ == "__main__":
    main()
