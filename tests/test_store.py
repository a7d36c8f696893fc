"""CPU: store.DeviceCSR.from_scipy accepts the reference's sparse adjacency convention (row i holds its neighbours in
columns 0..deg_i-1, 1-based ids: nn_modules.py:80-101) and refuses anything else instead of re-interpreting it."""
import numpy as np
import pytest
import torch
from scipy import sparse

from conftest import pkg

gs = pkg()
CPU = torch.device("cpu")


def _csr(rows, n_cols=None):
    """rows: list of (columns, ids) per row"""
    indptr = np.concatenate([[0], np.cumsum([len(c) for c, _ in rows])]).astype(np.int64)
    idx = np.concatenate([np.asarray(c, dtype=np.int32) for c, _ in rows] + [np.zeros(0, dtype=np.int32)])
    dat = np.concatenate([np.asarray(d, dtype=np.int64) for _, d in rows] + [np.zeros(0, dtype=np.int64)])
    n_cols = n_cols or max([len(c) for c, _ in rows] + [1]) + 2
    return sparse.csr_matrix((dat, idx, indptr), shape=(len(rows), n_cols))


def test_reference_convention_is_accepted_and_uploaded_as_is():
    rng = np.random.RandomState(0)
    n = 300
    deg = rng.randint(0, 9, size=n)
    deg[0] = deg[5] = 0
    rows = [(np.arange(d), rng.randint(1, n, size=d)) for d in deg]
    adj = _csr(rows, n_cols=12)
    g = gs.DeviceCSR.from_scipy(adj, CPU)
    assert g.n_rows == n and g.max_deg == 12 and g.nnz == int(deg.sum())
    assert np.array_equal(g.rowptr.numpy(), np.concatenate([[0], np.cumsum(deg)]))
    assert g.col.dtype == torch.int32 and np.array_equal(g.col.numpy(), adj.data.astype(np.int32))
    assert np.array_equal(gs.store.row_positions(adj.indptr), adj.indices)       # (what the O(rows) check stands for)
    # unsorted columns are sorted first, as before
    perm = _csr([(np.array([2, 0, 1]), np.array([7, 8, 9])), (np.array([0]), np.array([3]))])
    assert not perm.has_sorted_indices
    g2 = gs.DeviceCSR.from_scipy(perm, CPU)
    assert g2.col.tolist() == [8, 9, 7, 3]
    empty = gs.DeviceCSR.from_scipy(_csr([(np.zeros(0), np.zeros(0)), (np.zeros(0), np.zeros(0))]), CPU)
    assert empty.nnz == 0


@pytest.mark.parametrize("rows,what", [
    ([([0, 2], [4, 5]), ([0], [6])], "a gap: the last column is not deg - 1"),
    ([([1, 2], [4, 5]), ([0], [6])], "the first column is not 0"),
    ([([0, 1, 1], [4, 5, 6]), ([0], [6])], "a duplicate column"),
    ([([0, 1, 3], [4, 5, 6]), ([0, 1], [6, 7])], "columns skip one"),
])
def test_other_layouts_are_refused(rows, what):
    with pytest.raises(ValueError, match="sparse convention"):
        gs.DeviceCSR.from_scipy(_csr([(np.asarray(c), np.asarray(d)) for c, d in rows], n_cols=6), CPU)


@pytest.mark.parametrize("bad", [0, -3, 2 ** 31, 2 ** 40, -(2 ** 32) + 5, -(2 ** 33) + 1])      # (the last two wrap into range when narrowed to int32)
def test_ids_outside_one_based_int32_are_refused(bad):
    with pytest.raises(ValueError, match="1-based positive int32"):
        gs.DeviceCSR.from_scipy(_csr([(np.arange(3), np.array([5, bad, 7])), (np.arange(1), np.array([2]))]), CPU)
