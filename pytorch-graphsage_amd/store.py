"""
store.py -- HBM-resident data layouts of the hot path.

  DeviceCSR     the adjacency the sampler walks: (rowptr int64 [n_rows+1], col int32 [nnz]) built
                once from the reference's scipy csr_matrix (problem.py:70-72: csr_matrix((v,(r,c)))
                in the convention of utils/convert.py:100-126 -- ids 1-based, row 0 the dummy,
                row i's neighbours in columns 0..deg_i-1).
  FeatureStore  the node-feature table feats[N+1, D] (problem.py:118-121) kept in HBM as a
                row-major [n_rows, ld] matrix, bf16 (default on GPU) or fp32, rows padded with zeros
                to a multiple of 128 bytes so every row is a whole number of cache lines and
                16-byte lane loads are aligned.
  RowRef        what `feats[ids]` (models.py:76,80) returns for a FeatureStore: a *reference* to
                rows, consumed by the fused gather kernels; the [B*f1*f2, D] frontier the reference
                materialises per batch never exists unless someone asks for `.materialize()`.
"""
import numpy as np
import torch


def _round_up(v, m):
    return (v + m - 1) // m * m


def row_positions(indptr):
    """[0..deg_0-1, 0..deg_1-1, ...] for a CSR row pointer, via one cumsum (np.repeat on 1e8
    entries is ~10x slower)."""
    indptr = np.asarray(indptr, dtype=np.int64)
    nnz = int(indptr[-1])
    if nnz == 0:
        return np.zeros(0, dtype=np.int32)
    deg = np.diff(indptr)
    rows = np.flatnonzero(deg > 0)
    step = np.ones(nnz, dtype=np.int32)
    step[0] = 0
    step[indptr[rows[1:]]] -= deg[rows[:-1]].astype(np.int32)
    return np.cumsum(step, dtype=np.int32)


class DeviceCSR(object):
    def __init__(self, rowptr, col, n_rows, max_deg):
        self.rowptr = rowptr          # int64 [n_rows + 1]
        self.col = col                # int32 [nnz]
        self.n_rows = int(n_rows)     # adj.shape[0]  (counts the dummy row)
        self.max_deg = int(max_deg)   # adj.shape[1]: the population `sel` is drawn from
        self.err_flag = torch.zeros(1, dtype=torch.int32, device=rowptr.device)

    @property
    def device(self):
        return self.rowptr.device

    @property
    def nnz(self):
        return int(self.col.shape[0])

    @staticmethod
    def from_scipy(adj, device):
        """adj: scipy.sparse matrix in the reference convention.  Checks the convention instead
        of silently re-interpreting it: the reference indexes a row by COLUMN (nn_modules.py:90-93),
        which equals the position inside the row only when columns are 0..deg-1."""
        from scipy import sparse
        assert sparse.issparse(adj), "SparseUniformNeighborSampler: not sparse.issparse(adj)"
        adj = adj.tocsr()
        if not adj.has_sorted_indices:
            adj = adj.sorted_indices()
        indptr = np.asarray(adj.indptr, dtype=np.int64)
        # columns 0..deg-1 in every row: strictly increasing columns (scipy's canonical-format flag: one C pass over the
        # indices, cached on the matrix) whose LAST one is deg - 1 -- an O(rows) look instead of building and comparing
        # an nnz-long position array (0.36 of the 0.47 s the reference's Reddit command line took, two adjacencies)
        deg = np.diff(indptr)
        rows = np.flatnonzero(deg > 0)
        indices = np.asarray(adj.indices)
        ok = bool(adj.has_canonical_format) and (rows.size == 0 or (
            np.array_equal(indices[indptr[rows + 1] - 1], deg[rows] - 1) and bool((indices[indptr[rows]] == 0).all())))
        if not ok:
            raise ValueError("adjacency is not in the reference's sparse convention "
                             "(row i must hold its neighbours in columns 0..deg_i-1)")
        data = np.asarray(adj.data)
        # (ids wider than 32 bits: both bounds BEFORE narrowing -- -(2**32) + 5 would wrap to 5 and pass a check made after)
        if data.size and data.dtype.itemsize > 4 and (data.max() >= 2 ** 31 or data.min() < 1):
            raise ValueError("neighbour ids must be 1-based positive int32 values")
        rowptr = torch.from_numpy(indptr).to(device)
        col = torch.from_numpy(data.astype(np.int32, copy=False)).to(device)
        # (the lower bound is looked at where the narrowed ids already are: one pass on the device, not on the host)
        if data.size and int(col.min()) <= 0:
            raise ValueError("neighbour ids must be 1-based positive int32 values")
        return DeviceCSR(rowptr, col, adj.shape[0], adj.shape[1])

    def check(self):
        """Raise IndexError if a kernel saw an id outside the graph (synchronises)."""
        if int(self.err_flag.item()) != 0:
            self.err_flag.zero_()
            raise IndexError("sampler: node id out of range of the adjacency")

    @staticmethod
    def synthetic(n_rows, deg_lo, deg_hi, device, max_deg=None, seed=0, empty_every=0, chunk=1 << 29):
        """A random graph in the reference's sparse convention built ON THE DEVICE (benchmarks and the
        BASELINE-size tests: papers100M's 3.2e9 edges are 13 GB of int32 that no host round trip should carry):
        row 0 = the dummy (no neighbours), degrees uniform in [deg_lo, deg_hi], neighbour ids uniform in
        [1, n_rows); every `empty_every`-th row (if > 0) has no neighbours -> samples the dummy.  Row offsets are
        int64 (they pass 2^31), the edge array is filled in chunks (a single randint of > 2^31 elements is not
        something to rely on)."""
        gen = torch.Generator(device=device).manual_seed(int(seed))
        deg = torch.randint(int(deg_lo), int(deg_hi) + 1, (int(n_rows),), dtype=torch.int64, device=device, generator=gen)
        deg[0] = 0
        if empty_every:
            deg[3::int(empty_every)] = 0
        rowptr = torch.zeros(int(n_rows) + 1, dtype=torch.int64, device=device)
        torch.cumsum(deg, 0, out=rowptr[1:])
        del deg
        nnz = int(rowptr[-1])
        col = torch.empty(nnz, dtype=torch.int32, device=device)
        for o in range(0, nnz, chunk):
            n = min(chunk, nnz - o)
            col[o:o + n] = torch.randint(1, int(n_rows), (n,), dtype=torch.int32, device=device, generator=gen)
        return DeviceCSR(rowptr, col, n_rows, max_deg if max_deg is not None else int(deg_hi))


class DenseAdj(object):
    """The dense adjacency of the reference's default sampler (UniformNeighborSampler, nn_modules.py:19-49): an
    int64 [n_rows, K] table in HBM, every row pre-sampled to exactly K neighbours by the converter
    (utils/convert.py:71-98; ids 0-based, the dummy node is the LAST row).  Quacks like DeviceCSR where the fused
    engines need it (n_rows, max_deg = K, err_flag, check())."""

    def __init__(self, adj):
        assert torch.is_tensor(adj) and adj.dim() == 2 and adj.dtype == torch.int64, \
            "UniformNeighborSampler: adj must be a LongTensor [n_nodes + 1, K]"
        self.adj = adj.contiguous()
        self.n_rows, self.K = int(adj.shape[0]), int(adj.shape[1])
        self.max_deg = self.K
        self.err_flag = torch.zeros(1, dtype=torch.int32, device=adj.device)

    @property
    def device(self):
        return self.adj.device

    def check(self):
        """Raise IndexError if a kernel saw an id outside the table (synchronises)."""
        if self.adj.is_cuda and int(self.err_flag.item()) != 0:
            self.err_flag.zero_()
            raise IndexError("sampler: node id out of range of the adjacency")


class FeatureStore(object):
    """Device-resident node-feature table.  Quacks enough like the reference's `problem.feats`
    tensor for models.py / train.py: `.shape`, `.size()`, `feats[ids]`, `.is_cuda`."""

    def __init__(self, data, dim):
        assert data.dim() == 2 and data.is_contiguous()
        if int(data.shape[0]) >= 2 ** 31:            # the gather kernels read the low 32-bit word of a node id
            raise ValueError("FeatureStore: %d rows; node ids must stay below 2^31 (include/gsage.h, gsage_gather_mean)"
                             % int(data.shape[0]))
        self.data = data               # [n_rows, ld]
        self.dim = int(dim)            # logical D (columns [D, ld) are zero)

    @staticmethod
    def from_array(feats, device, dtype="bf16"):
        feats = torch.as_tensor(np.asarray(feats) if not torch.is_tensor(feats) else feats)
        n_rows, dim = feats.shape
        tdt = {"bf16": torch.bfloat16, "fp32": torch.float32}[dtype]
        ld = _round_up(dim, 128 // (2 if dtype == "bf16" else 4))
        data = torch.zeros(n_rows, ld, dtype=tdt, device=device)
        data[:, :dim] = feats.to(device=device, dtype=torch.float32).to(tdt)
        return FeatureStore(data, dim)

    @staticmethod
    def synthetic(n_rows, dim, device, dtype="bf16", seed=0, chunk=1 << 23):
        """N(0, 1) rows generated on the device in chunks (a 111 M x 128 bf16 table is 28 GB; its fp32 staging copy
        would be twice that); row 0 (the dummy node) is zero."""
        tdt = {"bf16": torch.bfloat16, "fp32": torch.float32}[dtype]
        ld = _round_up(int(dim), 128 // (2 if dtype == "bf16" else 4))
        data = torch.zeros(int(n_rows), ld, dtype=tdt, device=device)
        gen = torch.Generator(device=device).manual_seed(int(seed))
        for o in range(0, int(n_rows), chunk):
            n = min(chunk, int(n_rows) - o)
            data[o:o + n, :dim] = torch.randn(n, int(dim), device=device, generator=gen).to(tdt)
        data[0].zero_()
        return FeatureStore(data, dim)

    @staticmethod
    def wrap(t):
        """Zero-copy view of an existing [n_rows, D] fp32 / bf16 tensor (ld == D)."""
        return FeatureStore(t.detach().contiguous(), t.shape[1])

    @property
    def shape(self):
        return (self.data.shape[0], self.dim)

    def size(self, d=None):
        return self.shape if d is None else self.shape[d]

    @property
    def ld(self):
        return int(self.data.shape[1])

    @property
    def is_cuda(self):
        return self.data.is_cuda

    @property
    def device(self):
        return self.data.device

    @property
    def dtype(self):
        return self.data.dtype

    def cuda(self):
        return FeatureStore(self.data.cuda(), self.dim)

    def __getitem__(self, ids):
        return RowRef(self, ids)

    def dense(self):
        """fp32 [n_rows, D] copy (tests / CPU mode)."""
        return self.data[:, :self.dim].float()


class RowRef(object):
    """`store[ids]`: rows of a FeatureStore selected by a LongTensor, not yet gathered."""

    def __init__(self, store, ids):
        self.store = store
        self.ids = ids.contiguous().view(-1)

    @property
    def shape(self):
        return (int(self.ids.shape[0]), self.store.dim)

    def size(self, d=None):
        return self.shape if d is None else self.shape[d]

    @property
    def is_cuda(self):
        return self.store.is_cuda

    @property
    def device(self):
        return self.store.device

    def materialize(self, dtype=torch.float32):
        """The reference's `feats[ids]` as a real [M, D] tensor (one gather kernel)."""
        from . import ops
        return ops.gather_rows(self.store, self.ids, out_dtype=dtype)
