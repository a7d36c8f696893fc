"""
nn_modules.py -- the reference's plugin surface, backed by the gfx950 kernels.

Same three lookup tables, class names, constructor keywords, parameter names/shapes (so
state_dicts interchange) and call signatures as the reference's nn_modules.py
(sampler_lookup :104-107, prep_lookup :169-173, aggregator_lookup :324-330); the bodies are
calls into ops.py (C ABI of libgsage_hip.so).  What differs on purpose:

  * `feats[ids]` may arrive as a store.RowRef instead of a materialised tensor; the aggregators
    then gather inside their kernels.  Plain tensors are accepted everywhere, as in the reference.
  * The sparse sampler never leaves the device: no D2H/H2D of ids, no scipy row slicing.
    rng="compat" draws `sel` from numpy's global legacy stream exactly like nn_modules.py:88
    (bit-identical samples for the same seed); rng="philox" draws it in-kernel (counter based).
"""
import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from . import _native as nat
from . import ops
from .store import DenseAdj, DeviceCSR, RowRef


# --------------------------------------------------------------------------------------------
# Samplers
# --------------------------------------------------------------------------------------------
class UniformNeighborSampler(object):
    """Dense `[n_nodes+1, K]` LongTensor adjacency (reference nn_modules.py:19-49): every row was
    pre-sampled to exactly K neighbours offline; a call picks the same n random columns for the
    whole batch (ONE torch.randperm per call, from torch's global CPU generator -- SURVEY quirk 4).
    On the GPU one launch of gsage_sample_dense computes `adj[ids][:, perm][:, :n]` without the
    [M, K] intermediate; CPU tensors keep stock indexing (host mode)."""

    def __init__(self, adj):
        self.adj = adj
        self._dev = {}                                    # device -> store.DenseAdj (train.check_samplers reads it)

    def table(self, device=None):
        """store.DenseAdj over the adjacency on `device` (default: where `adj` lives)."""
        adj = self.adj if torch.is_tensor(self.adj) else torch.as_tensor(np.asarray(self.adj), dtype=torch.int64)
        device = adj.device if device is None else torch.device(device)
        key = (device.type, device.index if device.index is not None or device.type == "cpu"
               else torch.cuda.current_device())
        if key not in self._dev:
            self._dev[key] = DenseAdj(adj.to(device))
        return self._dev[key]

    csr = table                                           # what the fused engines ask a sampler for

    @staticmethod
    def draw_keep(K, n_samples):
        """The columns a call keeps: head of ONE permutation of the K columns, drawn exactly where the
        reference draws it (nn_modules.py:44); n_samples = -1: all but the last (python slicing)."""
        return torch.randperm(K)[:n_samples]

    def __call__(self, ids, n_samples=-1):
        keep = self.draw_keep(self.adj.size(1), n_samples)
        if not ids.is_cuda:
            return self.adj[ids][:, keep]
        tab = self.table(ids.device)
        ids = ids.contiguous().view(-1)
        M, n = int(ids.shape[0]), int(keep.shape[0])
        out = torch.empty(M, n, dtype=torch.int64, device=ids.device)
        keep = keep.to(ids.device, non_blocking=True)
        nat.check(nat.lib().gsage_sample_dense(tab.adj.data_ptr(), tab.K, tab.n_rows, ids.data_ptr(), M,
                                               keep.data_ptr(), n, out.data_ptr(), tab.err_flag.data_ptr(),
                                               ops._stream()), "sample_dense")
        return out


class SparseUniformNeighborSampler(object):
    """CSR uniform neighbour sampling with replacement (reference nn_modules.py:52-101) as the
    K1 kernel.  Contract: out[i*n+j] = row_i[sel[i,j] % deg_i], or the
    dummy node 0 when deg_i == 0, with sel ~ U[0, adj.shape[1]) -- modulo bias included."""

    rng_default = "compat"

    def __init__(self, adj, rng=None, seed=0):
        self.adj = adj
        self._host = DeviceCSR.from_scipy(adj, torch.device("cpu"))
        self._dev = {}
        self.rng = rng or SparseUniformNeighborSampler.rng_default
        assert self.rng in ("compat", "philox")
        self.seed = int(seed)
        self.calls = 0                # host-side call index (philox, eager)
        self.call_ctr = None          # device counter tensor (philox inside a captured graph)
        self.shard = (0, 1)           # (rank, world): offsets the global sample index

    @property
    def degrees(self):
        """nn_modules.py:76-78: stored-entry count per row == rowptr differences."""
        return np.diff(self._host.rowptr.numpy())

    def csr(self, device):
        device = torch.device(device)
        if device.type == "cpu":
            return self._host
        key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
        if key not in self._dev:
            self._dev[key] = DeviceCSR(self._host.rowptr.to(device), self._host.col.to(device),
                                       self._host.n_rows, self._host.max_deg)
        return self._dev[key]

    def use_device_csr(self, csr):
        """Walk `csr` (a store.DeviceCSR already in HBM) on its device instead of uploading the scipy matrix this
        sampler was built with -- for graphs that only ever exist on the GPU (store.DeviceCSR.synthetic: a
        papers100M-sized CSR is 14 GB; the plugin API's scipy matrix is then a placeholder)."""
        dev = csr.device
        self._dev[(dev.type, dev.index if dev.index is not None else torch.cuda.current_device())] = csr
        return csr

    def __call__(self, ids, n_samples=128):
        assert n_samples > 0, 'SparseUniformNeighborSampler: n_samples must be set explicitly'
        ids = ids.contiguous().view(-1)
        csr = self.csr(ids.device)
        M = int(ids.shape[0])
        rank, world = self.shard
        if self.rng == "compat":
            # the reference's own draw (nn_modules.py:88).  Data-parallel: every rank draws the
            # whole job's matrix from the same stream and keeps its rows (dist.py).
            from .helpers import legacy_stream
            if ids.is_cuda and legacy_stream.enabled:
                # the same words, consumed on the device (gsage_mt_choice_device): no host draw, no H2D
                st = legacy_stream.acquire(ids.device)
                full = torch.empty(M * world * n_samples, dtype=torch.int32, device=ids.device)
                nat.check(nat.lib().gsage_mt_choice_device(st.data_ptr(), csr.max_deg, full.numel(),
                                                           full.data_ptr(), ops._stream()), "mt_choice_device")
                sel = full[rank * M * n_samples:(rank + 1) * M * n_samples]
                return ops.sample_csr(csr, ids, n_samples, sel=sel)
            legacy_stream.release()
            sel = np.random.choice(csr.max_deg, (M * world, n_samples))[rank * M:(rank + 1) * M]
            sel = torch.from_numpy(np.ascontiguousarray(sel, dtype=np.int32))
            if ids.is_cuda:
                sel = sel.to(ids.device, non_blocking=True)
            return ops.sample_csr(csr, ids, n_samples, sel=sel)
        ph = {"seed": self.seed, "g0": rank * M * n_samples}
        if self.call_ctr is not None:
            ph["call_ctr"] = self.call_ctr
            ph["call_base"] = self._static_calls
            self._static_calls += 1
        else:
            ph["call_base"] = self.calls
            self.calls += 1
        return ops.sample_csr(csr, ids, n_samples, philox=ph)

    # graph capture support: the call index lives in device memory and is advanced in-graph
    _static_calls = 0

    def begin_capture(self, counter):
        self.call_ctr = counter
        self._static_calls = 0

    def calls_in_capture(self):
        return self._static_calls


sampler_lookup = {
    "uniform_neighbor_sampler": UniformNeighborSampler,
    "sparse_uniform_neighbor_sampler": SparseUniformNeighborSampler,
}


# --------------------------------------------------------------------------------------------
# Preprocessers
# --------------------------------------------------------------------------------------------
def _as_tensor(feats):
    return feats.materialize() if isinstance(feats, RowRef) else feats


class IdentityPrep(nn.Module):
    """nn_modules.py:112-123: passes the (possibly lazy) feature rows through."""

    def __init__(self, input_dim, n_nodes=None):
        super(IdentityPrep, self).__init__()
        self.input_dim = input_dim

    @property
    def output_dim(self):
        return self.input_dim

    def forward(self, ids, feats, layer_idx=0):
        return feats


class NodeEmbeddingPrep(nn.Module):
    """nn_modules.py:126-155: trainable per-node embedding (+ affine), concatenated to the
    features when there are any.  Seeds (layer_idx 0) all read the spare row `n_nodes` so a node
    never sees its own embedding.  Gather = K2, dense gradient = K6 scatter-add."""

    def __init__(self, input_dim, n_nodes, embedding_dim=64):
        super(NodeEmbeddingPrep, self).__init__()
        self.n_nodes = n_nodes
        self.input_dim = input_dim
        self.embedding_dim = embedding_dim
        self.embedding = nn.Embedding(num_embeddings=n_nodes + 1, embedding_dim=embedding_dim)
        self.fc = nn.Linear(embedding_dim, embedding_dim)

    @property
    def output_dim(self):
        return (self.input_dim or 0) + self.embedding_dim

    def forward(self, ids, feats, layer_idx=0):
        rows = ids if layer_idx > 0 else torch.full_like(ids, self.n_nodes)
        embs = ops.embedding_rows(self.embedding.weight, rows)
        embs = ops.linear(embs, self.fc.weight, self.fc.bias)
        if self.input_dim:
            return torch.cat([_as_tensor(feats).float(), embs], dim=1)
        return embs


class LinearPrep(nn.Module):
    """nn_modules.py:158-166."""

    def __init__(self, input_dim, n_nodes, output_dim=32):
        super(LinearPrep, self).__init__()
        self.fc = nn.Linear(input_dim, output_dim, bias=False)
        self.output_dim = output_dim

    def forward(self, ids, feats, layer_idx=0):
        return ops.linear(_as_tensor(feats), self.fc.weight)


prep_lookup = {
    "identity": IdentityPrep,
    "node_embedding": NodeEmbeddingPrep,
    "linear": LinearPrep,
}


# --------------------------------------------------------------------------------------------
# Aggregators
# --------------------------------------------------------------------------------------------
def concat_combine(parts):
    return torch.cat(parts, dim=1)


def _split_activation(act):
    """(fusable code, leftover callable).  ReLU goes into the GEMM epilogue; anything else
    (e.g. the reference's `lambda x: x`, train.py:116) is applied afterwards."""
    if act is None:
        return nat.ACT_NONE, None
    if act in (F.relu, torch.relu) or isinstance(act, nn.ReLU):
        return nat.ACT_RELU, None
    return nat.ACT_NONE, act


class AggregatorMixin(object):
    @property
    def output_dim(self):
        """Width after combine_fn (nn_modules.py:178-182); 2 * output_dim_ for the concat."""
        probe = torch.zeros((1, self.output_dim_))
        return self.combine_fn([probe, probe]).size(1)

    def _project(self, x, agg):
        """combine_fn([fc_x(x), fc_neib(agg)]) + activation; one grouped MFMA launch for the
        stock concat."""
        code, post = _split_activation(self.activation)
        if self.combine_fn is concat_combine:
            hidden = code == nat.ACT_RELU and agg.is_cuda
            out = ops.sage_project(x, agg, self.fc_x.weight, self.fc_neib.weight, code,
                                   out_dtype=ops.torch_dtype() if hidden else torch.float32)
        else:
            xt = _as_tensor(x)
            out = self.combine_fn([ops.linear(xt, self.fc_x.weight),
                                   ops.linear(agg, self.fc_neib.weight)])
            if code == nat.ACT_RELU:
                out = torch.relu(out)
        return post(out) if post is not None else out


class MeanAggregator(nn.Module, AggregatorMixin):
    """nn_modules.py:185-204."""

    def __init__(self, input_dim, output_dim, activation, combine_fn=concat_combine):
        super(MeanAggregator, self).__init__()
        self.fc_x = nn.Linear(input_dim, output_dim, bias=False)
        self.fc_neib = nn.Linear(input_dim, output_dim, bias=False)
        self.output_dim_ = output_dim
        self.activation = activation
        self.combine_fn = combine_fn

    def forward(self, x, neibs):
        M = x.size(0)
        n = neibs.size(0) // M
        cdt = ops.torch_dtype() if neibs.is_cuda else torch.float32
        if isinstance(neibs, RowRef):
            same = isinstance(x, RowRef) and x.store is neibs.store and neibs.store.dtype == cdt
            agg = ops.gather_mean(neibs.store, neibs.ids, M, n, out_dtype=cdt,
                                  out_ld=neibs.store.ld if same else None)
        else:
            agg = ops.segment_mean(neibs, M, out_dtype=cdt)
        return self._project(x, agg)


class PoolAggregator(nn.Module, AggregatorMixin):
    """nn_modules.py:207-232.  pool_fn: "max" / "mean" run fused (K3: the [M*n, hidden] MLP
    output never reaches HBM); any other callable gets the unfused MLP output [M, n, hidden]."""

    def __init__(self, input_dim, output_dim, pool_fn, activation, hidden_dim=512,
                 combine_fn=concat_combine):
        super(PoolAggregator, self).__init__()
        self.mlp = nn.Sequential(nn.Linear(input_dim, hidden_dim, bias=True), nn.ReLU())
        self.fc_x = nn.Linear(input_dim, output_dim, bias=False)
        self.fc_neib = nn.Linear(hidden_dim, output_dim, bias=False)
        self.output_dim_ = output_dim
        self.activation = activation
        self.pool_fn = pool_fn
        self.combine_fn = combine_fn

    def forward(self, x, neibs):
        M = x.size(0)
        lin = self.mlp[0]
        if self.pool_fn in ("max", "mean"):
            mode = nat.POOL_MAX if self.pool_fn == "max" else nat.POOL_MEAN
            agg = ops.pool_mlp(neibs, lin.weight, lin.bias, M, mode)
        else:
            hid = ops.linear(_as_tensor(neibs), lin.weight, lin.bias, nat.ACT_RELU)
            agg = self.pool_fn(hid.view(M, -1, hid.size(1)))
        return self._project(x, agg)


class MaxPoolAggregator(PoolAggregator):
    """nn_modules.py:235-244."""

    def __init__(self, input_dim, output_dim, activation, hidden_dim=512, combine_fn=concat_combine):
        super(MaxPoolAggregator, self).__init__(input_dim=input_dim, output_dim=output_dim,
                                                pool_fn="max", activation=activation,
                                                hidden_dim=hidden_dim, combine_fn=combine_fn)


class MeanPoolAggregator(PoolAggregator):
    """nn_modules.py:247-256."""

    def __init__(self, input_dim, output_dim, activation, hidden_dim=512, combine_fn=concat_combine):
        super(MeanPoolAggregator, self).__init__(input_dim=input_dim, output_dim=output_dim,
                                                 pool_fn="mean", activation=activation,
                                                 hidden_dim=hidden_dim, combine_fn=combine_fn)


class LSTMAggregator(nn.Module, AggregatorMixin):
    """nn_modules.py:259-286.  Not on the north-star path: the
    recurrence runs on the stock torch/MIOpen LSTM, only the projection uses K5."""

    def __init__(self, input_dim, output_dim, activation, hidden_dim=512, bidirectional=False,
                 combine_fn=concat_combine):
        super(LSTMAggregator, self).__init__()
        assert not hidden_dim % 2, "LSTMAggregator: hiddem_dim % 2 != 0"
        self.lstm = nn.LSTM(input_dim, hidden_dim // (1 + bidirectional),
                            bidirectional=bidirectional, batch_first=True)
        self.fc_x = nn.Linear(input_dim, output_dim, bias=False)
        self.fc_neib = nn.Linear(hidden_dim, output_dim, bias=False)
        self.output_dim_ = output_dim
        self.activation = activation
        self.combine_fn = combine_fn

    def forward(self, x, neibs):
        xt, nt = _as_tensor(x).float(), _as_tensor(neibs).float()
        seq, _ = self.lstm(nt.view(xt.size(0), -1, nt.size(1)))
        return self._project(xt, seq[:, -1, :].contiguous())


class AttentionAggregator(nn.Module, AggregatorMixin):
    """nn_modules.py:289-321: scores = att(neibs) . att(x), softmax over the fanout, weighted sum
    of the RAW neighbour rows.  The two tiny att GEMMs run on K5 (tanh fused), the weighting on
    K4 with the neighbour gather fused."""

    def __init__(self, input_dim, output_dim, activation, hidden_dim=32, combine_fn=concat_combine):
        super(AttentionAggregator, self).__init__()
        self.att = nn.Sequential(nn.Linear(input_dim, hidden_dim, bias=False), nn.Tanh(),
                                 nn.Linear(hidden_dim, hidden_dim, bias=False))
        self.fc_x = nn.Linear(input_dim, output_dim, bias=False)
        self.fc_neib = nn.Linear(input_dim, output_dim, bias=False)
        self.output_dim_ = output_dim
        self.activation = activation
        self.combine_fn = combine_fn

    @staticmethod
    def _rows(t):
        """Operand of the att MLP.  Lazy feature rows of a bf16 table are gathered ONCE, in storage
        precision and with the table's zero padding, i.e. already in the layout K5 wants (the generic
        route materialises fp32 rows and re-casts them: 3x the bytes and two extra passes)."""
        if isinstance(t, RowRef) and t.store.data.is_cuda and t.store.data.dtype == torch.bfloat16 \
                and ops.config.compute_dtype == "bf16":
            st = t.store
            ids = t.ids.contiguous().view(-1)
            buf = ops._gather_mean_raw(st.data, st.ld, ids, int(ids.shape[0]), 1, torch.bfloat16, st.ld)
            return ops.mark_zero_padded(buf[:, :st.dim])
        return _as_tensor(t)

    def _att(self, t):
        hid = ops.linear(t, self.att[0].weight, None, nat.ACT_TANH)
        return ops.linear(hid, self.att[2].weight)

    def forward(self, x, neibs):
        M = x.size(0)
        n = neibs.size(0) // M
        # the reference's bare .squeeze() (nn_modules.py:311) changes meaning for M == 1 or
        # fanout == 1 (train.py:75 forbids batch 1); refuse instead of silently diverging
        assert M > 1 and n > 1, "AttentionAggregator: needs batch > 1 and fanout > 1"
        xt, nt = self._rows(x), self._rows(neibs)
        agg = ops.attn_aggregate(self._att(nt), self._att(xt), neibs if isinstance(neibs, RowRef)
                                 else nt, M)
        return self._project(x, agg)


aggregator_lookup = {
    "mean": MeanAggregator,
    "max_pool": MaxPoolAggregator,
    "mean_pool": MeanPoolAggregator,
    "lstm": LSTMAggregator,
    "attention": AttentionAggregator,
}
