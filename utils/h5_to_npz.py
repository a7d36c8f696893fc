#!/usr/bin/env python
"""
utils/h5_to_npz.py -- problem.h5 (what the reference's converters write: utils/convert.py:192-202,
utils/convert-pokec.py:73-88, utils/convert-cora.py:77-78) -> the .npz twin `NodeProblem` reads without h5py.

    python utils/h5_to_npz.py data/reddit/sparse-problem.h5 [data/reddit/sparse-problem.npz]

Same keys, same arrays (task, n_classes, sparse, feats, folds, targets, adj, train_adj); sparse adjacencies stay the
[3, nnz] (v, r, c) array of utils/convert.py:128-131.  Needs h5py -- run it on the machine that has the .h5 files;
the GPU image has no h5py, which is why the twin exists (pytorch-graphsage_amd/problem.py:_read_problem).
"""
import sys

import numpy as np


def convert(src, dst=None):
    try:
        import h5py
    except ImportError:
        raise SystemExit("h5_to_npz.py needs h5py (pip install h5py) on the machine that holds %s" % src)
    dst = dst or (src[:-3] if src.endswith(".h5") else src) + ".npz"
    out = {}
    with h5py.File(src, "r") as f:
        for k in f.keys():
            v = f[k][()]
            if isinstance(v, bytes):                       # h5py scalar strings (task)
                v = v.decode()
            v = np.asarray(v)
            if v.dtype.kind in "OS":                       # folds: variable-length / byte strings -> unicode
                v = np.array([s.decode() if isinstance(s, bytes) else str(s) for s in v.reshape(-1)]).reshape(v.shape)
            out[k] = v
    np.savez(dst, **out)
    print("%s -> %s (%s)" % (src, dst, ", ".join("%s %s" % (k, tuple(v.shape)) for k, v in sorted(out.items()))))
    return dst


if __name__ == "__main__":
    if len(sys.argv) not in (2, 3):
        raise SystemExit(__doc__)
    convert(*sys.argv[1:])
