#!/usr/bin/env python
"""
gen_golden_convert.py -- golden vectors for the problem-file converters (TEST INFRASTRUCTURE, runs
ONLY in the build container; same rules as gen_golden.py: fixtures are arrays, no reference source
or bytecode is written anywhere).

Imports the reference's utils/convert.py (/root/reference) under Python 3 and drives its FUNCTIONS
on small seeded graphs:
    make_adjacency          utils/convert.py:71-98      make_sparse_adjacency   utils/convert.py:100-126
    spadj2edgelist          utils/convert.py:128-131
Shims (harness side only): empty modules for h5py / cPickle / ujson, and networkx's version string
is reported as 1.x while the module is imported (it asserts networkx < 2 at import, convert.py:25;
the three functions only use G.nodes() / G.neighbors(), which behave the same in networkx 3).  The
script bodies (convert.py __main__, convert-cora.py, convert-pokec.py) are py2-only / stale
 and cannot run: their intended behaviour is restated in the product and
tested through properties instead.

    python -B tests/golden/gen_golden_convert.py     # regenerates tests/golden/convert_kat.npz
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REF = os.environ.get("GSAGE_REFERENCE", "/root/reference")
for _m in ("h5py", "cPickle", "ujson"):
    sys.modules.setdefault(_m, types.ModuleType(_m))

import numpy as np
import networkx as nx

_real = nx.__version__
nx.__version__ = "1.11"
sys.path.insert(0, os.path.join(REF, "utils"))
import convert as ref_convert            # noqa: E402  (reference)
nx.__version__ = _real

OUT = os.path.dirname(os.path.abspath(__file__))


def build(n, edges, preadd):
    G = nx.Graph()
    if preadd:
        G.add_nodes_from(range(n))
    G.add_edges_from([tuple(e) for e in edges])
    return G


def main():
    out = {}
    rng = np.random.RandomState(7)
    case = 0
    for (n, n_edges, max_degree, preadd) in [(30, 70, 8, True), (12, 40, 4, True), (25, 60, 16, False),
                                             (50, 400, 6, True)]:
        edges = rng.randint(0, n, size=(n_edges, 2))
        if not preadd:                       # labels must be exactly 0..n-1: make every node appear
            edges = np.vstack([edges, np.stack([rng.permutation(n), rng.permutation(n)], 1)])
        G = build(n, edges, preadd)
        assert sorted(G.nodes()) == list(range(n))
        sel = rng.rand(n) < 0.6
        for tag, s in (("all", None), ("sel", sel)):
            seed = 100 + case
            np.random.seed(seed)
            adj = ref_convert.make_adjacency(G, max_degree, sel=s)
            words_after = np.random.randint(0, 2 ** 31 - 1)          # position of the numpy stream afterwards
            sp = ref_convert.make_sparse_adjacency(G, sel=s)
            el = ref_convert.spadj2edgelist(sp)
            p = "c%d_%s_" % (case, tag)
            out[p + "adj"] = np.asarray(adj, dtype=np.int64)
            out[p + "stream_after"] = np.int64(words_after)
            out[p + "seed"] = np.int64(seed)
            out[p + "sp_data"], out[p + "sp_indices"], out[p + "sp_indptr"] = sp.data, sp.indices, sp.indptr
            out[p + "sp_shape"] = np.asarray(sp.shape, dtype=np.int64)
            out[p + "edgelist"] = np.asarray(el, dtype=np.int64)
        out["c%d_edges" % case] = edges.astype(np.int64)
        out["c%d_meta" % case] = np.asarray([n, max_degree, int(preadd)], dtype=np.int64)
        out["c%d_sel" % case] = sel
        out["c%d_node_order" % case] = np.asarray(list(G.nodes()), dtype=np.int64)
        case += 1
    out["n_cases"] = np.int64(case)
    np.savez_compressed(os.path.join(OUT, "convert_kat.npz"), **out)
    print("convert_kat: %d cases" % case)


if __name__ == "__main__":
    main()
