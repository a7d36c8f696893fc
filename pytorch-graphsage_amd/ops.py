"""
ops.py -- tensor-level operators of the hot path, bound to libgsage_hip.so through ctypes.

PyTorch is plumbing here: it owns device memory, streams and autograd bookkeeping; the work is
done by the HIP kernels behind include/gsage.h.  Dispatch rule (no silent fallback):
  * CUDA tensors  -> the native library, always; a missing / stale library raises
                     NativeLibraryError (see _native.py).
  * CPU tensors   -> "host mode": the same operator written with stock torch ops, used only for
                     the reference's CPU configuration (`train.py --no-cuda`, BASELINE config 1)
                     and for the multi-process gloo tests.  Never reached from a CUDA tensor.

Reference call sites each operator replaces are cited per function.
"""
import ctypes
import os

import numpy as np
import torch
import torch.nn.functional as F

from . import _native as nat
from .store import FeatureStore, RowRef, _round_up

_vp = ctypes.c_void_p


class _Config(object):
    compute_dtype = "bf16"      # "bf16": MFMA bf16 in / fp32 accumulate;  "fp32": exact-fp32 MFMA


config = _Config()


def set_compute_dtype(name):
    assert name in ("bf16", "fp32")
    config.compute_dtype = name


def torch_dtype(name=None):
    return {"bf16": torch.bfloat16, "fp32": torch.float32}[name or config.compute_dtype]


def _code(dtype):
    if dtype == torch.float32:
        return nat.F32
    if dtype == torch.bfloat16:
        return nat.BF16
    raise TypeError("gsage: unsupported dtype %s" % dtype)


def _ptr(t):
    return _vp(t.data_ptr()) if t is not None else None


def _stream():
    return _vp(torch.cuda.current_stream().cuda_stream)


def warmup(device):
    """Load the library (raises loudly when it is missing) before any graph capture."""
    nat.lib()


def _dgrad(gc, wa, K):
    """d input = gc @ W[:, :K] -- gc [M, N] in the compute type, wa [N, >= K] the weight's operand copy -- on K5
    (gsage_linear_nt: the NT kernel against the TRANSPOSED operand copy), fp32 result [M, K].  The backward of the
    module path (the literal aggregator_lookup[...] plug-in under autograd, reference nn_modules.py:196-204,
    :223-232) runs no library GEMM: round 4 still called torch.mm here."""
    cdt = gc.dtype
    epc = 8 if cdt == torch.bfloat16 else 4
    M, N = gc.shape
    out = torch.empty(M, K, dtype=torch.float32, device=gc.device)
    if M == 0:
        return out
    ga = _pad_cast(gc, cdt, 8 * epc)
    wt = _pad_cast(wa[:, :K].t(), cdt, 8 * epc)              # [K, round_up(N)]: rows = this product's outputs
    _linear_launch(_ptr(ga), ga.stride(0), None, 0, _ptr(wt), wt.stride(0), None, _ptr(out), K, M, K, N,
                   nat.ACT_NONE, 1, 0, 0, 0, _code(cdt), nat.F32)
    return out


def _wgrad_any(gc, xa, K):
    """d W = gc^T @ xa[:, :K] on K5b (gsage_wgrad: bf16 MFMA, or its fp32 twin in the parity mode) for ANY shape:
    the columns of gc are padded to whole 16-byte chunks (zero columns: zero rows of the result, dropped), the rows of
    xa taken as they are when they already are.  fp32 result [N, K]."""
    cdt = gc.dtype
    mult = 8 if cdt == torch.bfloat16 else 4
    M, N = gc.shape
    if M == 0:
        return torch.zeros(N, K, dtype=torch.float32, device=gc.device)
    g = _pad_cast(gc, cdt, mult)
    ok = (xa.dtype == cdt and xa.stride(1) == 1 and xa.stride(0) % mult == 0 and xa.data_ptr() % 16 == 0 and
          xa.stride(0) >= _round_up(K, 4))
    x = xa if ok else _pad_cast(xa[:, :K], cdt, mult)
    Np = g.shape[1]
    return wgrad(g, x, x.stride(0), 0, M, Np, K, Np)[0][:N]


def mark_zero_padded(view):
    """Tag a [:, :D] view of a buffer THIS library filled (rows gathered with their zero padding) so
    that _pad_cast may hand it to the GEMM kernels as is."""
    view._gsage_zero_padded = True
    return view


def _trusted(t):
    return bool(getattr(t, "_gsage_zero_padded", False))


def _pad_cast(t, dtype, mult, trusted=False):
    """[M, D] -> contiguous [M, round_up(D, mult)] of `dtype`, zero padded (no copy if already so).
    The GEMM kernels read whole padded rows, so a strided [:, :D] view is passed through only when the
    caller vouches for its pad columns (`trusted`: buffers this library produced, see
    mark_zero_padded); a user tensor sliced out of a wider buffer is copied into a zeroed one."""
    M, D = t.shape
    ld = _round_up(D, mult)
    if t.dtype == dtype and ld == D and t.is_contiguous():
        return t
    if trusted and t.dtype == dtype and t.stride(1) == 1 and t.stride(0) == ld and t.data_ptr() % 16 == 0:
        return t              # [:, :D] view of rows that are already padded (pad columns zero: gathered rows)
    out = torch.zeros(M, ld, dtype=dtype, device=t.device) if ld != D else \
        torch.empty(M, ld, dtype=dtype, device=t.device)
    out[:, :D] = t
    return out


# =============================================================================================
# K1  sampler
# =============================================================================================
def _philox_host(seed, call, g0, count, max_deg):
    """numpy Philox4x32-10 for host mode; same definition as the kernel (include/gsage.h)."""
    g = np.arange(g0, g0 + count, dtype=np.uint64)
    blk = g >> np.uint64(2)
    c = [(blk & np.uint64(0xFFFFFFFF)), (blk >> np.uint64(32)),
         np.full(count, call & 0xFFFFFFFF, dtype=np.uint64),
         np.full(count, (call >> 32) & 0xFFFFFFFF, dtype=np.uint64)]
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]
        p1 = np.uint64(0xCD9E8D57) * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0)) & mask, p1 & mask,
             ((p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1)) & mask, p0 & mask]
        k0 = (k0 + 0x9E3779B9) & 0xFFFFFFFF
        k1 = (k1 + 0xBB67AE85) & 0xFFFFFFFF
    words = np.stack(c, axis=1)[np.arange(count), (g & np.uint64(3)).astype(np.int64)]
    return ((words * np.uint64(max_deg)) >> np.uint64(32)).astype(np.int64)


_JUMP_TABLE = {}
MT_PAR_MIN = 400000          # requests below this many values stay on the one-workgroup kernel (~1 ms)
MT_PAR_MAX = 80000000        # values per parallel call: 4096 jump units of 64 refills reach 1.6e8 raw words


JUMP_TABLE_SHA256 = "2d7b0a6ba6b6c1566432f44eda5f747868391536bc0ceb09fd7efa9bc4420669"    # 39 936 words


def jump_table_ok(tab):
    """The jump polynomials are a constant of MT19937 (no seed enters them): a file is trusted by content, not by size."""
    import hashlib
    return hashlib.sha256(tab.tobytes()).hexdigest() == JUMP_TABLE_SHA256


def mt_jump_table(device):
    """The seed-independent jump polynomials of numpy's MT19937 (gsage_mt_jump_table: 128 x 312 words) on `device`.
    Computed by the library (~1 s: Berlekamp-Massey + square-and-multiply) the first time ever, then read from
    mt19937_jump_table.bin next to the library (written by __graft_entry__.build(), or here when missing)."""
    key = str(device)
    if key not in _JUMP_TABLE:
        import numpy as np
        L = nat.lib()
        words = int(L.gsage_mt_jump_table_words())
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mt19937_jump_table.bin")
        tab = None
        if os.path.exists(path) and os.path.getsize(path) == 8 * words:
            tab = np.fromfile(path, dtype=np.uint64)
            if not jump_table_ok(tab):           # a torn or stale file: recompute (the table is seed-independent)
                tab = None
        if tab is None:
            tab = np.zeros(words, dtype=np.uint64)
            nat.check(L.gsage_mt_jump_table(tab.ctypes.data, words), "mt_jump_table")
            assert jump_table_ok(tab), "gsage_mt_jump_table produced a table that is not MT19937's"
            tmp = "%s.%d.tmp" % (path, os.getpid())        # (every rank of a data-parallel run may get here at once)
            try:
                tab.tofile(tmp)
                os.replace(tmp, path)
            except OSError:
                try:
                    os.remove(tmp)
                except OSError:
                    pass
        _JUMP_TABLE[key] = torch.from_numpy(tab.view(np.int64)).to(device)
    return _JUMP_TABLE[key]


def _mt_choice_par(state, high, segs, flat):
    """the requests served by up to 256 workgroups at once (gsage_mt_choice_par); segs as in mt_choice_segments"""
    L = nat.lib()
    top = high - 1
    mask = (1 << top.bit_length()) - 1
    rate = (top + 1) / (mask + 1)
    table = mt_jump_table(flat.device)
    i = 0
    while i < len(segs):                          # (pieces of <= MT_PAR_MAX values: the jump table's reach)
        j, total = i, 0
        while j < len(segs) and (j == i or total + segs[j][1] <= MT_PAR_MAX):
            total += segs[j][1]
            j += 1
        part = segs[i:j]
        if total > MT_PAR_MAX:                    # one request larger than the reach: cut it
            o, c = part[0]
            part, segs = [(o, MT_PAR_MAX)], segs[:i] + [(o, MT_PAR_MAX), (o + MT_PAR_MAX, c - MT_PAR_MAX)] + segs[i + 1:]
            total, j = MT_PAR_MAX, i + 1
        units = int(total / (624.0 * rate) * 1.015 / 64) + 2
        per = -(-units // 256)
        n_wg = -(-units // per)
        cum = [0]
        for _, c in part:
            cum.append(cum[-1] + c)
        host = torch.tensor([cum, [o for o, _ in part] + [0]], dtype=torch.int64)
        dev = host.to(flat.device)
        scratch = torch.empty(int(L.gsage_mt_choice_par_scratch(n_wg)), dtype=torch.uint8, device=flat.device)
        nat.check(L.gsage_mt_choice_par(_ptr(state), int(high), len(part), _ptr(dev[0]), _ptr(dev[1]), total, _ptr(flat),
                                        _ptr(table), _ptr(scratch), scratch.numel(), n_wg, per, _stream()),
                  "mt_choice_par")
        i = j
    return flat


def mt_choice_segments(state, high, segs, out):
    """numpy's legacy stream on the device (`state`: helpers.legacy_stream.acquire), many np.random.choice(high, .)
    requests in ONE call: segs = [(offset into out, count)], served in order; out: int32 CUDA tensor (any shape,
    offsets address its flat view).  Large requests (an epoch's sampler draws) are served by many workgroups at
    once (gsage_mt_choice_par: jump-ahead), small ones by the one-workgroup kernel; values, state and position are
    numpy's either way."""
    segs = [(int(o), int(c)) for o, c in segs if c > 0]
    if not segs:
        return out
    flat = out.view(-1)
    assert flat.dtype == torch.int32 and flat.is_cuda and max(o + c for o, c in segs) <= flat.numel()
    if high < 2:                                   # a range of one value: numpy draws nothing
        for o, c in segs:
            flat[o:o + c].zero_()
        return out
    if sum(c for _, c in segs) >= MT_PAR_MIN and os.environ.get("GSAGE_MT_PARALLEL", "1") == "1":
        _mt_choice_par(state, int(high), segs, flat)
        return out
    host = torch.tensor([[o for o, _ in segs], [c for _, c in segs]], dtype=torch.int64)
    dev = host.to(flat.device)
    nat.check(nat.lib().gsage_mt_choice_segments(_ptr(state), int(high), len(segs), _ptr(dev[0]), _ptr(dev[1]),
                                                 _ptr(flat), _stream()), "mt_choice_segments")
    return out              # (`dev` may be freed: the caching allocator reuses it in stream order only)


def sample_csr(csr, ids, n, sel=None, philox=None, out=None):
    """SparseUniformNeighborSampler.__call__ (nn_modules.py:80-101).

    ids: LongTensor [M] on csr.device.  Exactly one of
      sel    : IntTensor [M*n] in [0, max_deg)   (parity level 1 / compat mode)
      philox : dict(seed, call_base, g0, call_ctr=None|cuda uint64-as-int64 tensor[1])
    Returns LongTensor [M*n] on the same device."""
    assert n > 0, "SparseUniformNeighborSampler: n_samples must be set explicitly"
    ids = ids.contiguous().view(-1)
    M = int(ids.shape[0])
    if ids.is_cuda:
        L = nat.lib()
        if out is None:
            out = torch.empty(M * n, dtype=torch.int64, device=ids.device)
        assert out.dtype == torch.int64 and out.numel() == M * n and out.is_contiguous()
        if sel is not None:
            sel = sel.contiguous().view(-1)
            assert sel.dtype == torch.int32 and sel.shape[0] == M * n and sel.is_cuda
            nat.check(L.gsage_sample_csr_sel(_ptr(csr.rowptr), _ptr(csr.col), csr.n_rows, _ptr(ids),
                                             M, n, _ptr(sel), _ptr(out), _ptr(csr.err_flag),
                                             _stream()), "sample_csr_sel")
        else:
            ctr = philox.get("call_ctr")
            sel_out = philox.get("sel_out")
            nat.check(L.gsage_sample_csr_philox(_ptr(csr.rowptr), _ptr(csr.col), csr.n_rows,
                                                _ptr(ids), M, n, csr.max_deg, int(philox["seed"]),
                                                _ptr(ctr), int(philox.get("call_base", 0)),
                                                int(philox.get("g0", 0)), _ptr(out), _ptr(sel_out),
                                                _ptr(csr.err_flag), _stream()), "sample_csr_philox")
        return out
    # ---- host mode
    idn = ids.numpy()
    if idn.size and (idn.min() < 0 or idn.max() >= csr.n_rows):
        raise IndexError("sampler: node id out of range of the adjacency")
    if sel is None:
        ctr = philox.get("call_ctr")
        call = int(philox.get("call_base", 0)) + (int(ctr.item()) if ctr is not None else 0)
        seln = _philox_host(int(philox["seed"]), call, int(philox.get("g0", 0)), M * n, csr.max_deg)
    else:
        seln = sel.numpy().reshape(-1).astype(np.int64)
    rp = csr.rowptr.numpy()
    beg = np.repeat(rp[idn], n)
    deg = np.repeat(rp[idn + 1] - rp[idn], n)
    off = np.where(deg > 0, seln % np.maximum(deg, 1), 0)
    col = csr.col.numpy()
    vals = col[np.minimum(beg + off, max(col.shape[0] - 1, 0))] if col.shape[0] else np.zeros_like(off)
    return torch.from_numpy(np.where(deg > 0, vals, 0).astype(np.int64))


# =============================================================================================
# K2 / K6  gather + mean, segment mean, scatter-add
# =============================================================================================
def _gather_mean_raw(table, D, ids, M, n, out_dtype, out_ld=None, out=None):
    """out[i] = mean_j table[ids[i*n+j]] (ids None: rows i*n+j).  table: [R, ld] tensor.
    `out`: optional preallocated [M, out_ld] destination (a row slice of a larger buffer)."""
    if out_ld is None:
        out_ld = D
    if table.is_cuda:
        if out is None:
            out = (torch.zeros if out_ld != D else torch.empty)(M, out_ld, dtype=out_dtype,
                                                                device=table.device)
        else:
            assert out.dtype == out_dtype and out.stride(0) == out_ld and out.shape[0] == M
        nat.check(nat.lib().gsage_gather_mean(_ptr(table), _code(table.dtype), table.stride(0),
                                              _ptr(ids), M, n, D, _ptr(out), _code(out_dtype),
                                              out_ld, _stream()), "gather_mean")
        return out
    rows = table[ids.view(-1), :D] if ids is not None else table[:M * n, :D]
    res = rows.float().view(M, n, D).mean(dim=1) if n > 1 else rows.float().view(M, D)
    out = torch.zeros(M, out_ld, dtype=out_dtype)
    out[:, :D] = res.to(out_dtype)
    return out


class _GatherTrainable(torch.autograd.Function):
    """Row gather from a TRAINABLE fp32 table with the reference's dense gradient
    (nn.Embedding, nn_modules.py:134,146-149): backward = K6 scatter-add into a zeroed table."""

    @staticmethod
    def forward(ctx, table, ids):
        ctx.save_for_backward(ids)
        ctx.tshape = table.shape
        ctx.table = table if (table.is_leaf and table.requires_grad) else None
        return _gather_mean_raw(table.detach(), table.shape[1], ids, int(ids.shape[0]), 1,
                                torch.float32)

    @staticmethod
    def backward(ctx, g):
        (ids,) = ctx.saved_tensors
        g = g.contiguous().float()
        t = ctx.table
        if (g.is_cuda and t is not None and t.grad is not None and t.grad.dtype == torch.float32 and
                t.grad.shape == t.shape and t.grad.is_contiguous() and not torch.is_grad_enabled()):
            # The parameter already has a (zeroed, persistent) gradient buffer -- optim.FlatAdam's view:
            # scatter-add straight into it.  The stock route materialises a dense zero table per use of
            # the embedding and lets autograd add it to .grad: 4 extra passes over the table per use
            # (Pokec: 3 uses x 418 MB).  Same result: the adds commute.
            nat.check(nat.lib().gsage_scatter_add_rows(_ptr(g), g.stride(0), _ptr(ids), int(ids.shape[0]), 1,
                                                       g.shape[1], 1.0, _ptr(t.grad), t.grad.stride(0),
                                                       _stream()), "scatter_add_rows")
            return None, None
        grad = torch.zeros(ctx.tshape, dtype=torch.float32, device=g.device)
        if g.is_cuda:
            nat.check(nat.lib().gsage_scatter_add_rows(_ptr(g), g.stride(0), _ptr(ids),
                                                       int(ids.shape[0]), 1, g.shape[1], 1.0,
                                                       _ptr(grad), grad.stride(0), _stream()),
                      "scatter_add_rows")
        else:
            grad.index_add_(0, ids, g)
        return grad, None


def embedding_rows(table, ids):
    """table[ids] for a trainable parameter (dense-grad semantics of the reference)."""
    return _GatherTrainable.apply(table, ids.contiguous().view(-1))


def gather_rows(store, ids, out_dtype=torch.float32):
    """feats[ids] (models.py:76,80) materialised: [M, D] tensor of `out_dtype`."""
    ids = ids.contiguous().view(-1)
    return _gather_mean_raw(store.data, store.dim, ids, int(ids.shape[0]), 1, out_dtype)


def gather_mean(store, ids, M, n, out_dtype=torch.float32, out_ld=None):
    """feats[ids].view(M, n, D).mean(1) without materialising feats[ids]
    (models.py:80 + nn_modules.py:197-198)."""
    ids = ids.contiguous().view(-1)
    assert ids.shape[0] == M * n
    if out_ld is not None and out_ld == store.ld:
        # the table's pad columns are zero, so averaging them writes the output's pad columns
        return _gather_mean_raw(store.data, store.ld, ids, M, n, out_dtype, out_ld)
    return _gather_mean_raw(store.data, store.dim, ids, M, n, out_dtype, out_ld)


class _SegmentMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, neibs, M, out_dtype):
        n = neibs.shape[0] // M
        ctx.dims = (M, n, neibs.shape[1], neibs.dtype)
        src = neibs.detach()
        if not src.is_contiguous():
            src = src.contiguous()
        return _gather_mean_raw(src, src.shape[1], None, M, n, out_dtype)

    @staticmethod
    def backward(ctx, g):
        M, n, D, dt = ctx.dims
        g = g.contiguous().float()
        if g.is_cuda:
            out = torch.empty(M * n, D, dtype=torch.float32, device=g.device)
            nat.check(nat.lib().gsage_segment_mean_bwd(_ptr(g), g.stride(0), M, n, D, _ptr(out), D,
                                                       _stream()), "segment_mean_bwd")
        else:
            out = (g / n).repeat_interleave(n, dim=0)
        return out.to(dt), None, None


def segment_mean(neibs, M, out_dtype=torch.float32):
    """neibs.view(M, -1, D).mean(dim=1) (nn_modules.py:197-198) for an in-order tensor."""
    assert neibs.shape[0] % M == 0
    return _SegmentMean.apply(neibs, M, out_dtype)


# =============================================================================================
# K5  projection GEMM
# =============================================================================================
def _linear_launch(A, lda, a_rows, a_rows_g0, W, ldw, bias, C, ldc, M, N, K, act, groups,
                   a_gs, w_gs, c_gs, dtype_code, c_code):
    nat.check(nat.lib().gsage_linear_nt(A, dtype_code, lda, a_rows, a_rows_g0, W, ldw, bias, C,
                                        c_code, ldc, M, N, K, act, groups, a_gs, w_gs, c_gs,
                                        _stream()), "linear_nt")


def pack_weight(W, K=None, out=None):
    """[groups, N, ld] (or [N, ld]) fp32 / bf16 weights -> the MFMA-fragment-ordered bf16 operand of
    gsage_linear_nt_packed (include/gsage.h).  K: logical inner size (default: the last dimension)."""
    W3 = W if W.dim() == 3 else W.unsqueeze(0)
    assert W3.stride(2) == 1 and W3.is_cuda
    groups, N, _ = W3.shape
    K = int(W3.shape[2] if K is None else K)
    n = nat.lib().gsage_packed_weight_elems(N, K, groups)
    if out is None:
        out = torch.empty(n, dtype=torch.bfloat16, device=W.device)
    assert out.numel() >= n and out.dtype == torch.bfloat16
    code = nat.BF16 if W3.dtype == torch.bfloat16 else nat.F32
    assert W3.dtype in (torch.bfloat16, torch.float32)
    nat.check(nat.lib().gsage_pack_weight(W3.data_ptr(), code, W3.stride(1), W3.stride(0) if groups > 1 else 0,
                                          N, K, groups, out.data_ptr(), _stream()), "pack_weight")
    return out


def _linear_packed_launch(A, lda, a_rows, a_rows_g0, Wp, bias, C, ldc, M, N, K, act, groups, a_gs, c_gs, c_code):
    nat.check(nat.lib().gsage_linear_nt_packed(A, lda, a_rows, a_rows_g0, Wp, bias, C, c_code, ldc, M, N, K,
                                               act, groups, a_gs, c_gs, _stream()), "linear_nt_packed")


def _prep_weight(W, cdt, epc):
    """[N, K] fp32 parameter -> [N, round_up(K, epc)] compute dtype, zero padded."""
    return _pad_cast(W.detach(), cdt, epc)


def gather_mean_multi(segments, ld, D, out_ld, adam=None, hops=None):
    """All hops of a level in one K2 launch.  segments: list of (table, ids|None, out, M, n) with
    bf16 (or, parity mode, fp32) row-major tensors sharing ld / out_ld.  adam: optional _native.AdamDesc -- the clip + Adam
    update of the previous batch rides in the same launch; hops: optional _native.HopsDesc -- so does
    the frontier sampling of a later batch (gsage_gather_mean_multi_adam)."""
    k = len(segments)
    T = (ctypes.c_void_p * k)(*[s[0].data_ptr() for s in segments])
    I = (ctypes.c_void_p * k)(*[(s[1].data_ptr() if s[1] is not None else None) for s in segments])
    O = (ctypes.c_void_p * k)(*[s[2].data_ptr() for s in segments])
    Ms = (ctypes.c_int64 * k)(*[int(s[3]) for s in segments])
    ns = (ctypes.c_int32 * k)(*[int(s[4]) for s in segments])
    code, ocode = _code(segments[0][0].dtype), _code(segments[0][2].dtype)
    assert all(s[0].dtype == segments[0][0].dtype and s[2].dtype == segments[0][2].dtype for s in segments)
    if adam is not None or hops is not None:
        nat.check(nat.lib().gsage_gather_mean_multi_adam(
            k, T, I, O, Ms, ns, code, ld, D, code, out_ld,
            ctypes.addressof(adam) if adam is not None else None,
            ctypes.addressof(hops) if hops is not None else None, _stream()), "gather_mean_multi_adam")
        return
    nat.check(nat.lib().gsage_gather_mean_multi(k, T, I, O, Ms, ns, code, ld, D, ocode, out_ld,
                                                _stream()), "gather_mean_multi")


def wgrad_plan(M, Ntot, K, target=240):
    """(rows_per_split, n_slabs, ldk) used by wgrad.  A workgroup owns one 128 x 128 output tile and
    one M-slice (its four waves quarter the slice and meet in LDS), one workgroup fits per CU, and
    every slice costs a partial tile in HBM: so aim at ~240 workgroups (MI355X: 256 CUs, the rest is
    left to the small problems sharing a gsage_wgrad_multi launch) and never below 64 rows per wave.
    target: workgroups to aim at (small problems that share a launch with a big one take fewer, longer
    slices: fewer partial tiles to write and to sum)."""
    ldk = _round_up(K, 4)
    tiles = ((Ntot + 127) // 128) * ((ldk + 127) // 128)
    s_target = max(1, target // tiles)
    rps = max(256, _round_up((M + s_target - 1) // s_target, 16))
    return rps, (M + rps - 1) // rps, ldk


def wgrad_balance(shapes, budget=248, group=8):
    """Workgroup targets for K5b problems that share gsage_wgrad_multi launches (issued `group` at a time, in this
    order).  One workgroup fits per CU, so a launch with more workgroups than CUs runs in rounds and its small problems
    then cost a round of their own (the max-pool step's six problems used 572 workgroups: 150 us where the big one
    alone takes 98).  Per launch: the smallest slice length R (rows, multiple of 16, >= 256) for which
    sum_i tiles_i * ceil(M_i / R) <= budget -- every problem gets slices of about the same length, so they finish
    together, in one round.  shapes: [(M, Ntot, K)]; returns the per-problem `target` for wgrad_plan."""
    targets = []
    for i in range(0, len(shapes), group):
        chunk = shapes[i:i + group]
        tiles = [((nt + 127) // 128) * ((_round_up(k, 4) + 127) // 128) for (_m, nt, k) in chunk]
        R = 256
        while sum(t * ((m + R - 1) // R) for t, (m, _nt, _k) in zip(tiles, chunk)) > budget and R < max(m for m, _, _ in chunk):
            R += 16
        targets += [t * ((m + R - 1) // R) for t, (m, _nt, _k) in zip(tiles, chunk)]
    return targets


def wgrad(dC, A, lda, a_gstride, M, Ntot, K, n_per_group, out=None, slabs=None, reduce=True):
    """dW_g = dC_g^T @ A_g on the matrix cores (K5b), bf16 operands, fp32 result
    [groups, n_per_group, K].  dC: contiguous bf16 [M, >=Ntot]; A: bf16 [M, lda] row-major.
    reduce=False leaves the per-slice partial tiles in `slabs` ([S, Ntot, ldk]) for
    gsage_finalize_grads and returns the slabs."""
    groups = (Ntot + n_per_group - 1) // n_per_group
    rps, S, ldk = wgrad_plan(M, Ntot, K)
    if slabs is None:
        slabs = torch.empty(S, Ntot, ldk, dtype=torch.float32, device=dC.device)
    if reduce and out is None:
        out = torch.empty(groups, n_per_group, K, dtype=torch.float32, device=dC.device)
    nat.check(nat.lib().gsage_wgrad(_ptr(dC), _code(dC.dtype), dC.stride(0), _ptr(A), lda, a_gstride, M, Ntot, K,
                                    n_per_group, rps, _ptr(slabs), ldk, _ptr(out) if reduce else None,
                                    n_per_group * K, _stream()), "wgrad")
    return out if reduce else slabs


def wgrad_multi(problems):
    """Several K5b problems in one launch; partial tiles stay in each problem's slabs
    (gsage_finalize_grads sums them).  problems: list of (dC, A, lda, a_gstride, M, Ntot, K,
    n_per_group, slabs[, workgroup target[, a_rows]]) with the meaning of wgrad(); a_rows (int64 [M], 16-byte
    aligned): reduction index m reads row a_rows[m] of A -- a frontier's table rows in place."""
    descs = (nat.WgradDesc * len(problems))()
    for d, prob in zip(descs, problems):
        dC, A, lda, a_gs, M, Ntot, K, npg, slabs = prob[:9]
        rps, S, ldk = wgrad_plan(M, Ntot, K, *[v for v in prob[9:10] if v is not None])
        assert tuple(slabs.shape) == (S, Ntot, ldk) and slabs.is_contiguous()
        rows = prob[10] if len(prob) > 10 else None
        assert rows is None or (rows.dtype == torch.int64 and rows.is_contiguous() and rows.numel() >= M)
        d.a_rows = _ptr(rows) if rows is not None else None
        d.dC, d.A, d.slabs = _ptr(dC), _ptr(A), _ptr(slabs)
        d.ldc, d.lda, d.a_gstride = dC.stride(0), lda, a_gs
        d.M, d.Ntot, d.K, d.n_per_group, d.ldk, d.rows_per_split = M, Ntot, K, npg, ldk, rps
    code = _code(problems[0][0].dtype)
    assert all(prob[0].dtype == prob[1].dtype == problems[0][0].dtype for prob in problems)
    nat.check(nat.lib().gsage_wgrad_multi(len(problems), ctypes.cast(descs, ctypes.c_void_p), code, _stream()),
              "wgrad_multi")


class _Linear(torch.autograd.Function):
    """act(x @ W^T + b) on the matrix cores; the backward contractions too (K5 against the transposed operand copy for the
    input gradient, K5b for the weight gradient)."""

    @staticmethod
    def forward(ctx, x, W, b, act, cdt_name, out_dtype):
        cdt = torch_dtype(cdt_name)
        epc = 8 if cdt == torch.bfloat16 else 4
        M, K = x.shape
        N = W.shape[0]
        xa = _pad_cast(x.detach(), cdt, 8 * epc, _trusted(x))   # whole 128-byte rows: LDS-DMA GEMM path
        wa = _prep_weight(W, cdt, 8 * epc)
        out = torch.empty(M, N, dtype=out_dtype, device=x.device)
        bf = b.detach().float().contiguous() if b is not None else None
        _linear_launch(_ptr(xa), xa.stride(0), None, 0, _ptr(wa), wa.stride(0), _ptr(bf),
                       _ptr(out), N, M, N, K, act, 1, 0, 0, 0, _code(cdt), _code(out_dtype))
        ctx.save_for_backward(xa, wa, out if act != nat.ACT_NONE else None)
        ctx.meta = (act, K, x.dtype, b is not None, W.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        xa, wa, out = ctx.saved_tensors
        act, K, xdt, has_b, wdt = ctx.meta
        g = g.float()
        if act == nat.ACT_RELU:
            g = g * (out > 0)
        elif act == nat.ACT_TANH:
            o = out.float()
            g = g * (1 - o * o)
        gc = g.to(xa.dtype)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _dgrad(gc, wa, K).to(xdt)
        if ctx.needs_input_grad[1]:
            # K5b (the library GEMM behind gc.t() @ xa ran at ~20 TF/s on these skinny shapes)
            dw = _wgrad_any(gc, xa, K).to(wdt)
        if has_b and ctx.needs_input_grad[2]:
            db = g.sum(dim=0)
        return dx, dw, db, None, None, None


def linear(x, W, b=None, act=nat.ACT_NONE, compute_dtype=None, out_dtype=torch.float32):
    """nn.Linear (+ fused activation) on CUDA through K5; host mode uses F.linear."""
    if not x.is_cuda:
        y = F.linear(x.float(), W, b)
        if act == nat.ACT_RELU:
            y = torch.relu(y)
        elif act == nat.ACT_TANH:
            y = torch.tanh(y)
        return y
    return _Linear.apply(x, W, b, act, compute_dtype or config.compute_dtype, out_dtype)


class _SageProject(torch.autograd.Function):
    """act(cat[x @ Wx^T, agg @ Wn^T], dim=1) (nn_modules.py:200-202 and its twins at :228-230,
    :317-319) as ONE grouped MFMA launch writing both halves of the concat.  x may be a row
    reference (table, ids): then the A tile of group 0 is gathered inside the kernel."""

    @staticmethod
    def forward(ctx, x, agg, Wx, Wn, x_table, x_ids, x_dim, act, cdt_name, out_dtype):
        cdt = torch_dtype(cdt_name)
        epc = 8 if cdt == torch.bfloat16 else 4
        h = Wx.shape[0]
        M = agg.shape[0]
        Dn = Wn.shape[1]
        an = _pad_cast(agg.detach(), cdt, 8 * epc, _trusted(agg))
        if x_ids is not None:
            Dx = x_dim
            if x_table.dtype != cdt or x_table.stride(0) % epc != 0:
                xa = _gather_mean_raw(x_table, Dx, x_ids, M, 1, cdt, _round_up(Dx, epc))
                a_rows = None
            else:
                xa, a_rows = x_table, x_ids
        else:
            Dx = x.shape[1]
            xa, a_rows = _pad_cast(x.detach(), cdt, 8 * epc, _trusted(x)), None
        out = torch.empty(M, 2 * h, dtype=out_dtype, device=agg.device)
        esz = xa.element_size()
        delta = an.data_ptr() - xa.data_ptr()
        # one grouped launch when both halves share K and a leading dimension: group 1's A
        # operand is addressed as A + a_gstride, i.e. the agg buffer relative to the x operand
        grouped = (Dx == Dn and delta % esz == 0 and xa.stride(0) == an.stride(0))
        if grouped:
            ldw = _round_up(Dx, 8 * epc)
            w2 = torch.zeros(2, h, ldw, dtype=cdt, device=agg.device)
            w2[0, :, :Dx] = Wx.detach()
            w2[1, :, :Dn] = Wn.detach()
        if grouped:
            _linear_launch(_ptr(xa), xa.stride(0), _ptr(a_rows), 1, _ptr(w2), ldw, None, _ptr(out),
                           2 * h, M, h, Dx, act, 2, delta // esz, h * ldw, h, _code(cdt),
                           _code(out_dtype))
            wxa, wna = w2[0], w2[1]
        else:
            wxa = _prep_weight(Wx, cdt, 8 * epc)
            wna = _prep_weight(Wn, cdt, 8 * epc)
            _linear_launch(_ptr(xa), xa.stride(0), _ptr(a_rows), 1, _ptr(wxa), wxa.stride(0), None,
                           _ptr(out), 2 * h, M, h, Dx, act, 1, 0, 0, 0, _code(cdt), _code(out_dtype))
            _linear_launch(_ptr(an), an.stride(0), None, 0, _ptr(wna), wna.stride(0), None,
                           _vp(out.data_ptr() + h * out.element_size()), 2 * h, M, h, Dn, act, 1,
                           0, 0, 0, _code(cdt), _code(out_dtype))
        ctx.save_for_backward(xa, a_rows, an, wxa, wna, out if act == nat.ACT_RELU else None)
        ctx.grouped = (delta // esz) if grouped else None
        ctx.meta = (act, h, Dx, Dn, x.dtype if x is not None else None, agg.dtype, Wx.dtype, cdt, epc)
        return out

    @staticmethod
    def backward(ctx, g):
        xa, a_rows, an, wxa, wna, out = ctx.saved_tensors
        act, h, Dx, Dn, xdt, adt, wdt, cdt, epc = ctx.meta
        g = g.float()
        if act == nat.ACT_RELU:
            g = g * (out > 0)
        gc = g.to(cdt)
        gx, gn = gc[:, :h], gc[:, h:]
        dx = dagg = dwx = dwn = None
        fused_w = (cdt == torch.bfloat16 and ctx.grouped is not None and h % 128 == 0 and
                   xa.stride(0) % 8 == 0 and ctx.needs_input_grad[2] and ctx.needs_input_grad[3])
        # halves with different K (pool / attention aggregators): one K5b launch each -- the library
        # GEMM behind gx.t() @ xm ran at ~20 TF/s on these skinny, transposed operands
        k5b_ok = cdt == torch.bfloat16 and h % 8 == 0 and gc.shape[0] >= 64 and gc.is_contiguous()
        if fused_w:
            # both weight gradients in one MFMA launch (x rows re-gathered when they were lazy)
            xm = xa if a_rows is None else _gather_mean_raw(xa, an.stride(0), a_rows, an.shape[0], 1,
                                                            cdt, an.stride(0))
            dw = wgrad(gc.contiguous(), xm, xm.stride(0), (an.data_ptr() - xm.data_ptr()) // 2,
                       an.shape[0], 2 * h, Dx, h)
            dwx, dwn = dw[0].to(wdt), dw[1].to(wdt)
        if ctx.needs_input_grad[2] and not fused_w:
            if a_rows is not None:
                M = an.shape[0]
                xm = _gather_mean_raw(xa, Dx, a_rows, M, 1, cdt, _round_up(Dx, epc))
            else:
                xm = xa
            if k5b_ok and xm.stride(0) % 8 == 0:
                dwx = wgrad(gx, xm, xm.stride(0), 0, gc.shape[0], h, Dx, h)[0].to(wdt)
            else:
                dwx = _wgrad_any(gx, xm, Dx).to(wdt)
        if ctx.needs_input_grad[3] and not fused_w:
            if k5b_ok and an.stride(0) % 8 == 0:
                dwn = wgrad(gn, an, an.stride(0), 0, gc.shape[0], h, Dn, h)[0].to(wdt)
            else:
                dwn = _wgrad_any(gn, an, Dn).to(wdt)
        if ctx.needs_input_grad[0]:
            dx = _dgrad(gx, wxa, Dx).to(xdt)
        if ctx.needs_input_grad[1]:
            dagg = _dgrad(gn, wna, Dn).to(adt)
        return dx, dagg, dwx, dwn, None, None, None, None, None, None


def sage_project(x, agg, Wx, Wn, act=nat.ACT_NONE, compute_dtype=None, out_dtype=torch.float32):
    """x: Tensor [M, Dx] or RowRef; agg: Tensor [M, Dn].  Returns [M, 2h]."""
    if not agg.is_cuda:
        xt = x.materialize() if isinstance(x, RowRef) else x
        y = torch.cat([F.linear(xt.float(), Wx), F.linear(agg.float(), Wn)], dim=1)
        return torch.relu(y) if act == nat.ACT_RELU else y
    cd = compute_dtype or config.compute_dtype
    if isinstance(x, RowRef):
        return _SageProject.apply(None, agg, Wx, Wn, x.store.data, x.ids, x.store.dim, act, cd,
                                  out_dtype)
    return _SageProject.apply(x, agg, Wx, Wn, None, None, 0, act, cd, out_dtype)


# =============================================================================================
# K3  pooling MLP
# =============================================================================================
class _PoolMLP(torch.autograd.Function):
    """pool_r relu(neibs @ Wm^T + bm) over each group of n rows (nn_modules.py:224-226,:240,:252);
    the [M*n, H] hidden activations stay on chip.  Backward recomputes nothing it can route
    through argmax: d hidden is non-zero only at the winning row of each (segment, channel)."""

    @staticmethod
    def forward(ctx, neibs, Wm, bm, table, ids, dim, M, n, mode, cdt_name):
        cdt = torch_dtype(cdt_name)
        epc = 8 if cdt == torch.bfloat16 else 4
        H = Wm.shape[0]
        if ids is not None:
            K = dim
            if table.dtype != cdt or table.stride(0) % epc != 0:
                A = _gather_mean_raw(table, K, ids, M * n, 1, cdt, _round_up(K, epc))
                a_rows = None
            else:
                A, a_rows = table, ids
        else:
            K = neibs.shape[1]
            A, a_rows = _pad_cast(neibs.detach(), cdt, epc, _trusted(neibs)), None
        wa = _prep_weight(Wm, cdt, epc)
        bf = bm.detach().float().contiguous()
        pooled = torch.empty(M, H, dtype=torch.float32, device=A.device)
        argmax = torch.empty(M, H, dtype=torch.int32, device=A.device) if mode == nat.POOL_MAX else None
        nat.check(nat.lib().gsage_pool_mlp(_ptr(A), _code(cdt), A.stride(0), _ptr(a_rows), _ptr(wa),
                                           wa.stride(0), _ptr(bf), M, n, H, K, mode, _ptr(pooled),
                                           H, _ptr(argmax), None, 0, None, _stream()), "pool_mlp")
        ctx.save_for_backward(A, a_rows, wa, bf, pooled, argmax)
        ctx.meta = (M, n, H, K, mode, cdt, neibs.dtype if neibs is not None else None, Wm.dtype, epc)
        return pooled

    @staticmethod
    def backward(ctx, g):
        A, a_rows, wa, bf, pooled, argmax = ctx.saved_tensors
        M, n, H, K, mode, cdt, ndt, wdt, epc = ctx.meta
        g = g.float()
        rows = A if a_rows is None else _gather_mean_raw(A, K, a_rows, M * n, 1, cdt,
                                                         _round_up(K, epc))
        dn = dw = db = None
        fast = mode == nat.POOL_MAX and cdt == torch.bfloat16 and H % 128 == 0 and g.is_cuda
        if fast:
            # routed gradient written once as the bf16 operand K5b wants (no fp32 [M, n, H] scatter)
            g = g.contiguous()
            ghc = torch.empty(M * n, H, dtype=torch.bfloat16, device=g.device)
            nat.check(nat.lib().gsage_pool_route_bwd(_ptr(g), H, _ptr(pooled), H, _ptr(argmax), H, M, n, H,
                                                     _ptr(ghc), nat.BF16, H, _stream()), "pool_route_bwd")
            if ctx.needs_input_grad[2]:
                db = (g * (pooled > 0)).sum(dim=0)
            if ctx.needs_input_grad[1]:
                dw = wgrad(ghc, rows, rows.stride(0), 0, M * n, H, K, H)[0].to(wdt)
        else:
            if mode == nat.POOL_MAX:
                gh = torch.zeros(M, n, H, dtype=torch.float32, device=g.device)
                gh.scatter_(1, argmax.long().unsqueeze(1), (g * (pooled > 0)).unsqueeze(1))
                gh = gh.view(M * n, H)
            else:
                hid = torch.empty(M * n, H, dtype=torch.float32, device=g.device)     # recompute (mean pool only): K5
                ra = _pad_cast(rows, cdt, 8 * epc, True) if rows.stride(0) % (8 * epc) == 0 else _pad_cast(rows[:, :K], cdt, 8 * epc)
                wl = _pad_cast(wa[:, :K], cdt, 8 * epc)
                _linear_launch(_ptr(ra), ra.stride(0), None, 0, _ptr(wl), wl.stride(0), _ptr(bf), _ptr(hid), H, M * n, H, K,
                               nat.ACT_RELU, 1, 0, 0, 0, _code(cdt), nat.F32)
                gh = (g / n).repeat_interleave(n, dim=0) * (hid > 0)
            ghc = gh.to(cdt)
            if ctx.needs_input_grad[1]:
                dw = _wgrad_any(ghc, rows, K).to(wdt)
            if ctx.needs_input_grad[2]:
                db = gh.sum(dim=0)
        if ctx.needs_input_grad[0]:
            dn = _dgrad(ghc, wa, K).to(ndt)
        return dn, dw, db, None, None, None, None, None, None, None


def pool_mlp(neibs, Wm, bm, M, mode, compute_dtype=None):
    """neibs: Tensor [M*n, D] or RowRef.  Returns pooled [M, H] fp32."""
    total = neibs.shape[0]
    assert total % M == 0
    n = total // M
    if not neibs.is_cuda:
        nb = neibs.materialize() if isinstance(neibs, RowRef) else neibs
        hid = torch.relu(F.linear(nb.float(), Wm, bm)).view(M, n, -1)
        return hid.max(dim=1)[0] if mode == nat.POOL_MAX else hid.mean(dim=1)
    cd = compute_dtype or config.compute_dtype
    if n > 64:      # tile holds whole segments only up to 64 rows: unfused route, still K5
        nb = neibs.materialize() if isinstance(neibs, RowRef) else neibs
        hid = linear(nb, Wm, bm, nat.ACT_RELU, cd).view(M, n, -1)
        return hid.max(dim=1)[0] if mode == nat.POOL_MAX else hid.mean(dim=1)
    if isinstance(neibs, RowRef):
        return _PoolMLP.apply(None, Wm, bm, neibs.store.data, neibs.ids, neibs.store.dim, M, n,
                              mode, cd)
    return _PoolMLP.apply(neibs, Wm, bm, None, None, 0, M, n, mode, cd)


# =============================================================================================
# K4  attention weighting
# =============================================================================================
class _AttnAggregate(torch.autograd.Function):
    """softmax_r(<na[i,r], xa[i]>) weighted sum of the raw neighbour rows (nn_modules.py:309-315)."""

    @staticmethod
    def forward(ctx, na, xa, neibs, table, ids, dim, M, n):
        na = na.contiguous().float()
        xa = xa.contiguous().float()
        if ids is not None:
            T, D, rows_ids = table, dim, ids
        else:
            T = neibs.detach()
            T = T if T.is_contiguous() else T.contiguous()
            D, rows_ids = T.shape[1], None
        agg = torch.empty(M, D, dtype=torch.float32, device=na.device)
        ws = torch.empty(M, n, dtype=torch.float32, device=na.device)
        nat.check(nat.lib().gsage_attn_aggregate(_ptr(na), na.stride(0), _ptr(xa), xa.stride(0),
                                                 _ptr(T), _code(T.dtype), T.stride(0),
                                                 _ptr(rows_ids), M, n, na.shape[1], D, _ptr(agg), D,
                                                 _ptr(ws), _stream()), "attn_aggregate")
        ctx.save_for_backward(na, xa, T, rows_ids, ws)
        ctx.meta = (M, n, D, neibs.dtype if neibs is not None else None)
        return agg

    @staticmethod
    def backward(ctx, g):
        na, xa, T, rows_ids, ws = ctx.saved_tensors
        M, n, D, ndt = ctx.meta
        g = g.float().contiguous()
        vec = 8 if T.dtype == torch.bfloat16 else 4
        wide = (T.dtype in (torch.bfloat16, torch.float32) and n <= 32 and T.stride(0) % vec == 0 and
                T.data_ptr() % 16 == 0 and _round_up(D, vec) <= T.stride(0))
        if wide:
            # dws, the softmax backward and both products in one launch; the rows are read once, in
            # storage precision, straight from the table
            Ha = na.shape[1]
            dna = torch.empty(M * n, Ha, dtype=torch.float32, device=g.device)
            dxa = torch.empty(M, Ha, dtype=torch.float32, device=g.device)
            nat.check(nat.lib().gsage_attn_bwd(_ptr(g), g.stride(0), _ptr(ws), _ptr(na), na.stride(0), _ptr(xa),
                                               xa.stride(0), _ptr(T), _code(T.dtype), T.stride(0), _ptr(rows_ids),
                                               M, n, Ha, D, _ptr(dna), Ha, _ptr(dxa), Ha, _stream()), "attn_bwd")
            dneibs = None
            if ctx.needs_input_grad[2]:
                dneibs = (ws.unsqueeze(2) * g.unsqueeze(1)).reshape(M * n, D).to(ndt)
            return dna, dxa, dneibs, None, None, None, None, None
        rows = T if rows_ids is None else _gather_mean_raw(T, D, rows_ids, M * n, 1, torch.float32)
        rows = rows[:, :D].float().view(M, n, D)
        dws = torch.bmm(rows, g.unsqueeze(2)).squeeze(2)                    # [M, n]
        ds = ws * (dws - (dws * ws).sum(dim=1, keepdim=True))               # softmax backward
        nav = na.view(M, n, -1)
        dna = (ds.unsqueeze(2) * xa.unsqueeze(1)).reshape(M * n, -1)
        dxa = torch.bmm(ds.unsqueeze(1), nav).squeeze(1)
        dneibs = None
        if ctx.needs_input_grad[2]:
            dneibs = (ws.unsqueeze(2) * g.unsqueeze(1)).reshape(M * n, D).to(ndt)
        return dna, dxa, dneibs, None, None, None, None, None


def attn_aggregate(na, xa, neibs, M):
    """na: att(neibs) [M*n, Ha]; xa: att(x) [M, Ha]; neibs: Tensor [M*n, D] or RowRef."""
    n = na.shape[0] // M
    if not na.is_cuda:
        nb = neibs.materialize() if isinstance(neibs, RowRef) else neibs
        s = torch.bmm(na.view(M, n, -1), xa.view(M, -1, 1)).squeeze(2)
        w = torch.softmax(s, dim=1)
        return (nb.float().view(M, n, -1) * w.unsqueeze(-1)).sum(dim=1)
    if n > 64:
        nb = neibs.materialize() if isinstance(neibs, RowRef) else neibs
        s = torch.bmm(na.view(M, n, -1), xa.view(M, -1, 1)).squeeze(2)
        w = torch.softmax(s, dim=1)
        return (nb.float().view(M, n, -1) * w.unsqueeze(-1)).sum(dim=1)
    if isinstance(neibs, RowRef):
        return _AttnAggregate.apply(na, xa, None, neibs.store.data, neibs.ids, neibs.store.dim, M, n)
    return _AttnAggregate.apply(na, xa, neibs, None, None, 0, M, n)
