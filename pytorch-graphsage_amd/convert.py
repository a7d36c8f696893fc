"""
convert.py -- offline converters: raw datasets -> the problem file NodeProblem loads
(SURVEY section 8(f) rank 1 + 3).  CPU-side, not on the timed path.

Restates the reference's utils/convert.py (functions, pinned by tests/golden/convert_kat.npz) and
the INTENDED behaviour of its three scripts, which cannot run as shipped:
  * utils/convert.py __main__ (:148-202)  GraphSAGE json format -> dense problem   (py2 + networkx 1.x)
  * utils/convert-cora.py                 calls make_adjacency with a signature that no longer exists
                                          (:77-78 vs convert.py:71) and never adds the dummy row
  * utils/convert-pokec.py                converts the sparse adjacency to an edge list twice
                                          (:85-86 and again inside save_problem, convert.py:59-61)
Differences on purpose: no networkx / h5py dependency (a 20-line adjacency-list graph; `.npz` twin of
problem.h5 with the same keys, plus .h5 when h5py is importable), one conversion of sparse
adjacencies, dummy rows always added.  Conventions kept: dense files put the dummy
node LAST (id = n_nodes), sparse files FIRST (ids 1-based, 0 = dummy); padded dense rows are filled
by sampling the node's own neighbours; random choices come from the global numpy legacy stream in
the reference's order, so a seeded run reproduces the reference's adjacency bit for bit.

    python -m pytorch-graphsage_amd.convert graphsage --inpath data/reddit [--sparse]
    python -m pytorch-graphsage_amd.convert cora      --inpath data/cora
    python -m pytorch-graphsage_amd.convert pokec     --inpath data/pokec
"""
from __future__ import print_function

import argparse
import json
import os
import sys

import numpy as np
from scipy.sparse import csr_matrix

from .problem import save_problem_npz


# ---- graph: nodes in insertion order, neighbours in edge-insertion order (what networkx's
#      G.nodes() / G.neighbors() give the reference, utils/convert.py:73,84,102,112) -------------
class Graph(object):
    def __init__(self, n_nodes=0):
        self._nb = {}
        for i in range(n_nodes):
            self._nb[i] = {}

    def add_edge(self, u, v):
        u, v = int(u), int(v)
        self._nb.setdefault(u, {})
        self._nb.setdefault(v, {})
        self._nb[u].setdefault(v, True)
        self._nb[v].setdefault(u, True)

    def nodes(self):
        return list(self._nb.keys())

    def neighbors(self, node):
        return list(self._nb[int(node)].keys())

    def __len__(self):
        return len(self._nb)


def graph_from_edgelist(edges, n_nodes=None):
    """nx.from_edgelist(edges) (n_nodes None: nodes appear in order of first mention), or a graph
    whose nodes 0..n_nodes-1 exist up front (node-link files list the nodes first)."""
    G = Graph(n_nodes or 0)
    for u, v in np.asarray(edges).reshape(-1, 2):
        G.add_edge(u, v)
    return G


# ---- adjacency builders ---------------------------------------------------------------------------
def make_adjacency(G, max_degree, sel=None):
    """Dense [n_nodes + 1, max_degree] adjacency, every entry initialised to the dummy node id
    n_nodes; rows of nodes with more than max_degree neighbours are subsampled without replacement,
    shorter rows are padded by resampling with replacement (utils/convert.py:71-98).  sel: boolean
    mask over nodes; only selected nodes get a row and only selected neighbours count."""
    all_nodes = np.array(G.nodes())
    n_nodes = len(all_nodes)
    adj = np.full((n_nodes + 1, max_degree), n_nodes, dtype=np.int64)
    if sel is not None:
        all_nodes = all_nodes[sel]
    for node in all_nodes:
        neibs = np.array(G.neighbors(node), dtype=np.int64)
        if sel is not None and len(neibs):
            neibs = neibs[sel[neibs]]
        if len(neibs) > 0:
            if len(neibs) > max_degree:
                neibs = np.random.choice(neibs, max_degree, replace=False)
            elif len(neibs) < max_degree:
                extra = np.random.choice(neibs, max_degree - neibs.shape[0], replace=True)
                neibs = np.concatenate([neibs, extra])
            adj[node, :] = neibs
    return adj


def make_sparse_adjacency(G, sel=None):
    """CSR adjacency in the (v, r, c) convention NodeProblem reads back (problem.py:70-72): row
    node+1 holds the 1-based neighbour ids in columns 0..deg-1; row 0 / id 0 is the dummy
    (utils/convert.py:100-126)."""
    all_nodes = np.array(G.nodes())
    if sel is not None:
        all_nodes = all_nodes[sel]
    r, c, v = [], [], []
    for node in all_nodes:
        neibs = np.array(G.neighbors(node), dtype=np.int64)
        if sel is not None and len(neibs):
            neibs = neibs[sel[neibs]]
        if len(neibs) > 0:
            r.append(np.full(len(neibs), node + 1, dtype=np.int64))
            c.append(np.arange(len(neibs), dtype=np.int64))
            v.append(neibs + 1)
    return csr_matrix((np.hstack(v), (np.hstack(r), np.hstack(c))))


def spadj2edgelist(spadj):
    """[3, nnz] array (v; r; c), utils/convert.py:128-131."""
    rr, cc = spadj.nonzero()
    return np.vstack([spadj.data, rr, cc])


# ---- problem files --------------------------------------------------------------------------------
def parse_fold(x):
    return 'test' if x.get('test') else ('val' if x.get('val') else 'train')


def validate_problem(problem):
    """utils/convert.py:38-53."""
    for k in ('adj', 'train_adj', 'targets', 'folds'):
        assert problem.get(k) is not None, "problem[%r] is None" % k
    if problem.get('feats') is not None:
        assert problem['feats'].shape[0] == problem['targets'].shape[0], "feats / targets rows differ"
        assert problem['feats'].shape[0] == problem['folds'].shape[0], "feats / folds rows differ"
        if not problem.get('sparse'):
            assert problem['adj'].shape[0] == problem['feats'].shape[0], "adj / feats rows differ"
    assert problem['adj'].shape[0] == problem['train_adj'].shape[0], "adj / train_adj rows differ"
    assert len(problem['targets'].shape) == 2, "targets must be 2-D"
    return True


def save_problem(problem, outpath):
    """Writes `outpath` (.npz: the twin NodeProblem reads when h5py is absent; .h5: through h5py when
    it is importable).  Sparse adjacencies (scipy matrices) are stored ONCE as the (v, r, c) edge
    list of spadj2edgelist."""
    assert validate_problem(problem)
    assert not os.path.exists(outpath), 'save_problem: %s already exists' % outpath
    problem = dict(problem)
    if problem.get('sparse'):
        for k in ('adj', 'train_adj'):
            if hasattr(problem[k], 'nonzero') and hasattr(problem[k], 'data') and not isinstance(problem[k], np.ndarray):
                problem[k] = spadj2edgelist(problem[k])
    if outpath.endswith('.h5'):
        try:
            import h5py
        except ImportError:
            raise RuntimeError("h5py is not installed: write the .npz twin instead (%s)" % (outpath[:-3] + '.npz'))
        f = h5py.File(outpath, 'w')
        for k, v in problem.items():
            if v is not None:
                f[k] = np.asarray(v).astype('S') if np.asarray(v).dtype.kind == 'U' else v
        f.close()
    else:
        save_problem_npz(outpath, problem)
    return outpath


def _augment(feats, targets, folds, dummy_first):
    """One extra row for the dummy node: last for dense files (utils/convert.py:187-189), first for
    sparse ones (:209-211)."""
    zt = np.zeros((1, targets.shape[1]), dtype=targets.dtype)
    zf = None if feats is None else np.zeros((1, feats.shape[1]), dtype=feats.dtype)
    if dummy_first:
        return (None if feats is None else np.vstack([zf, feats]), np.vstack([zt, targets]),
                np.hstack([['dummy'], folds]))
    return (None if feats is None else np.vstack([feats, zf]), np.vstack([targets, zt]),
            np.hstack([folds, ['dummy']]))


def build_problem(G, feats, targets, folds, task, n_classes, max_degree=128, sparse=False):
    """Adjacencies (all edges / training-fold edges) + dummy-augmented arrays, as utils/convert.py
    :181-202 (dense) and the commented-out sparse variant :204-225."""
    train = (folds == 'train')
    if sparse:
        # NB the file stores (v, r, c) only, so the loader infers both dimensions from the largest
        # row / column present: max_deg of adj and train_adj differ.
        adj = make_sparse_adjacency(G, sel=None)
        train_adj = make_sparse_adjacency(G, sel=train)
    else:
        adj = make_adjacency(G, max_degree, sel=None)
        train_adj = make_adjacency(G, max_degree, sel=train)
    f, t, fo = _augment(feats, targets, folds, dummy_first=sparse)
    out = {"task": task, "n_classes": n_classes, "adj": adj, "train_adj": train_adj, "feats": f,
           "targets": t, "folds": fo}
    if sparse:
        out["sparse"] = True
    return out


# ---- dataset front ends ---------------------------------------------------------------------------
def convert_graphsage(inpath, outpath, max_degree=128, task='classification', sparse=False,
                      links_are_indices=True):
    """GraphSAGE's json format (G.json node-link graph, id_map.json, class_map.json, feats.npy), the
    body of utils/convert.py:148-202.  links_are_indices: networkx 1.x node-link files (what the
    reference requires) store link endpoints as indices into the node list."""
    from sklearn.preprocessing import StandardScaler
    assert task in ('classification', 'multilabel_classification'), 'unknown task'
    id2target = json.load(open(os.path.join(inpath, 'class_map.json')))
    id2idx = json.load(open(os.path.join(inpath, 'id_map.json')))
    feats = np.load(os.path.join(inpath, 'feats.npy'))
    raw = json.load(open(os.path.join(inpath, 'G.json')))
    node_ids = [nd['id'] for nd in raw['nodes']]
    pos = {nid: i for i, nid in enumerate(node_ids)}
    G = Graph(len(node_ids))                       # == nx.convert_node_labels_to_integers(G)
    for lk in raw['links']:
        u, v = lk['source'], lk['target']
        if not links_are_indices:
            u, v = pos[u], pos[v]
        G.add_edge(u, v)
    feats = np.vstack([feats[id2idx[str(i)]] for i in node_ids])
    targets = np.vstack([id2target[str(i)] for i in node_ids])
    folds = np.array([parse_fold(nd) for nd in raw['nodes']])
    feats = StandardScaler().fit(feats[folds == 'train']).transform(feats)
    n_classes = len(np.unique(targets)) if task == 'classification' else targets.shape[1]
    return save_problem(build_problem(G, feats, targets, folds, task, n_classes, max_degree, sparse), outpath)


def load_cora(path, dataset='cora', drop_last_feature=False):
    """<path>/<dataset>.content (id, 1433 binary features, label) and .cites (cited, citing) ->
    (features [N, D] float32, symmetric 0/1 adjacency [N, N], integer labels [N]).
    drop_last_feature reproduces utils/convert-cora.py:19 (`[:, 1:-2]`, which loses a column)."""
    raw = np.loadtxt(os.path.join(path, dataset + '.content'), dtype=np.dtype(str))
    feats = raw[:, 1:(-2 if drop_last_feature else -1)].astype(np.float32)
    names = sorted(set(raw[:, -1]))
    labels = np.array([names.index(s) for s in raw[:, -1]], dtype=np.int64)
    idx_map = {j: i for i, j in enumerate(raw[:, 0].astype(np.int64))}
    cites = np.loadtxt(os.path.join(path, dataset + '.cites'), dtype=np.int64).reshape(-1, 2)
    n = feats.shape[0]
    A = np.zeros((n, n), dtype=np.float32)
    for a, b in cites:
        if a in idx_map and b in idx_map:
            A[idx_map[a], idx_map[b]] = 1
            A[idx_map[b], idx_map[a]] = 1
    return feats, A, labels


def convert_cora(inpath, outpath, max_degree=128, dataset='cora', drop_last_feature=False):
    """Planetoid-style split of utils/convert-cora.py:60-69 -- 140 train, 300 val (rows 200..499), the
    rest from row 500 on is test, rows 140..199 and 500 - 440 = 60 trailing rows fall out (the
    script truncates every array to the fold vector's length) -- row-normalised features,
    self-loops, dense adjacency with the dummy row."""
    feats, A, labels = load_cora(inpath, dataset, drop_last_feature)
    folds = np.array(['train'] * 140 + ['val'] * 300 + ['test'] * (feats.shape[0] - 500))
    n = folds.shape[0]
    rs = feats.sum(axis=1, keepdims=True)
    feats = (feats / np.where(rs == 0, 1, rs))[:n]
    targets = labels[:n].reshape(-1, 1)
    A = A[:n][:, :n] + np.identity(n, dtype=np.float32)
    G = graph_from_edgelist(np.vstack(np.where(A)).T, n_nodes=n)
    prob = build_problem(G, feats, targets, folds, 'classification', int(np.unique(targets).shape[0]), max_degree)
    return save_problem(prob, outpath)


def convert_pokec(inpath, out_dense, out_sparse, max_degree=128, seed=123):
    """soc-pokec-ages.tsv (id, age) + soc-pokec-relationships.txt (src, trg) -> age regression
    problems without features (utils/convert-pokec.py): users with a positive age that take part in
    at least one edge between such users, ids renumbered in file order, random 50/50 train/val."""
    import pandas as pd
    np.random.seed(seed)
    ages = pd.read_csv(os.path.join(inpath, 'soc-pokec-ages.tsv'), header=None, sep='\t', names=('id', 'age'),
                       dtype={'id': np.int64, 'age': str}, keep_default_na=False)
    ages = ages[ages.age != 'null']
    ages = ages.assign(age=ages.age.astype(int))
    ages = ages[ages.age > 0]
    edges = pd.read_csv(os.path.join(inpath, 'soc-pokec-relationships.txt'), header=None, sep='\t',
                        names=('src', 'trg'))
    edges = edges[edges.src.isin(ages.id) & edges.trg.isin(ages.id)]
    ages = ages[ages.id.isin(edges.src) | ages.id.isin(edges.trg)]
    uid = dict(zip(ages.id.values, np.arange(ages.shape[0])))
    e = np.stack([edges.src.map(uid).values, edges.trg.map(uid).values], 1)
    targets = ages.age.values.astype(float).reshape(-1, 1)
    folds = np.random.choice(['train', 'val'], targets.shape[0], p=[0.5, 0.5])
    G = graph_from_edgelist(e, n_nodes=targets.shape[0])
    outs = []
    for path, sp in ((out_dense, False), (out_sparse, True)):
        if path:
            prob = build_problem(G, None, targets, folds, 'regression_mae', None, max_degree, sparse=sp)
            prob['train_adj'] = prob['adj']                 # utils/convert-pokec.py:69,86: one adjacency
            outs.append(save_problem(prob, path))
    return outs


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('dataset', choices=['graphsage', 'cora', 'pokec'])
    ap.add_argument('--inpath', type=str, required=True)
    ap.add_argument('--outpath', type=str)
    ap.add_argument('--max-degree', type=int, default=128)
    ap.add_argument('--task', type=str, default='classification')
    ap.add_argument('--sparse', action='store_true')
    ap.add_argument('--seed', type=int, default=123)
    args = ap.parse_args(argv)
    ext = '.h5' if 'h5py' in sys.modules else '.npz'
    if args.dataset == 'graphsage':
        out = args.outpath or os.path.join(args.inpath, ('sparse-problem' if args.sparse else 'problem') + ext)
        print(convert_graphsage(args.inpath, out, args.max_degree, args.task, args.sparse))
    elif args.dataset == 'cora':
        print(convert_cora(args.inpath, args.outpath or os.path.join(args.inpath, 'problem' + ext), args.max_degree))
    else:
        print(convert_pokec(args.inpath, os.path.join(args.inpath, 'problem' + ext),
                            os.path.join(args.inpath, 'sparse-problem' + ext), args.max_degree, args.seed))


if __name__ == "__main__":
    main()
