"""
engine/pool.py -- FusedPoolTrainStep: max-pool / mean-pool aggregators (reference nn_modules.py:207-256;
BASELINE configs[2]).
"""
import os

import torch

from .. import _native as nat
from .. import ops
from ..nn_modules import IdentityPrep, MaxPoolAggregator, MeanPoolAggregator, NodeEmbeddingPrep
from .common import FusedTrainStep, _PrepDesc, _ReduceDesc, _r8, _r64


class FusedPoolTrainStep(FusedTrainStep):
    """train_step for max-pool / mean-pool aggregators (reference nn_modules.py:207-256; BASELINE config 3) with no
    autograd and no framework glue, on the same machinery as FusedMeanTrainStep (flat buckets, fused
    multi-hop sampler, head kernel, finalisation + Adam, command lists, batch queue, data-parallel
    order).  Per level l, rows = hops 0 .. L-l-1 ("x"), neighbour rows = hops 1 .. L-l:

      forward    K3   pooled = max_j relu(Wm nb_j + bm) per hop (hidden [M*n, Hm] never leaves the chip),
                      fp32 + bf16 operand copy + argmax
                 K5   out[:, :h] = act(x Wx^T);  K5  out[:, h:] = act(pooled Wn^T)
      backward   K5   d pooled = dC[:, h:] Wn            (NT GEMM against the transposed operand copy)
                 route d pooled through the max / ReLU -> bf16 d hidden [M*n, Hm]   (gsage_pool_route_bwd;
                       mean pool: g / n where K3's sign mask is set, gsage_pool_route_mean_bwd)
                 bias partials of the MLP                                           (gsage_pool_bias_partials)
                 l > 0: K5 dX = dC[:, :h] Wx, K5 dN = d hidden Wm, merge + ReLU mask -> dC of level l-1
                 K5b  all weight gradients of all levels in one grouped launch (fc_x, fc_neib, mlp)
    Level 0 reads its operands from two row buffers gathered once per step (x rows, neighbour rows):
    K3, K5 and K5b all want plain row-major operands, and the gathers run one batch ahead beside Adam.
    Under the trainable node-embedding prep (nn_modules.py:126-155; with or without features beside it) the level-0
    rows of the whole frontier are the prep's output, computed per step into one buffer (common._prep_forward): x rows
    = its first rows, neighbour rows = the rest; level 0 then passes a gradient down too (dX through fc_x, dN through
    the pooling MLP, added row by row -- no ReLU below an affine prep) and common._prep_backward takes it to
    prep.fc and the table.
    """

    NPART = 256          # partial rows per hop for the MLP bias gradient (summed by the finalisation)
    TIMED = {"gather": (0, 1), "k3": (4, 5), "k5b": (6, 7)}      # K3 = level 0's launch over the LAST hop (the bulk)

    @classmethod
    def why_not(cls, model, feats, ddp=None):
        why = cls._why_not_common(model, feats, (MaxPoolAggregator, MeanPoolAggregator), "max-pool / mean-pool")
        if why:
            return why
        if not isinstance(model.prep, (IdentityPrep, NodeEmbeddingPrep)):
            return "a prep class other than identity / node_embedding (%s)" % type(model.prep).__name__
        why = cls._why_not_input(model, feats, ddp, concat_ok=True)
        if why:
            return why
        layers = list(model.agg_layers.children())
        # whole lines for the LDS-DMA kernels (the embedding prep's output rows are padded to them by the engine)
        if isinstance(model.prep, IdentityPrep) and feats.ld % (64 if feats.dtype == torch.bfloat16 else 4) != 0:
            return "feature rows that are not whole 128-byte lines"
        if any(fn.keywords["n_samples"] > 64 for fn in model.train_sample_fns) or len(layers) > 3:
            return "a fan-out above 64 (K3 tiles hold whole segments up to 64 rows) or more than three layers"
        if not all(l.output_dim_ % 64 == 0 and l.mlp[0].weight.shape[0] % 128 == 0 for l in layers):
            return "output dims that are not multiples of 64, or a pooling MLP whose width is not a multiple of 128"
        return None

    # ---- construction ------------------------------------------------------------------------------
    def _init_levels(self, example_ids, example_targets):
        feats, dev, L = self.store, self.dev, self.L
        bf, f32, i32 = self.tdt, torch.float32, torch.int32
        is_bf = self.code == nat.BF16
        self.pool_mode = nat.POOL_MAX if type(self.layers[0]) is MaxPoolAggregator else nat.POOL_MEAN
        self.h = [l.output_dim_ for l in self.layers]
        self.Hm = [int(l.mlp[0].weight.shape[0]) for l in self.layers]
        if self.emb:                                 # level-0 rows = [features |] prep.fc(embedding[ids])
            d0 = self.D0 + self.E
            ld0 = _r64(d0) if is_bf else _r8(d0)
        else:
            d0, ld0 = feats.dim, feats.ld
        self.din = [d0] + [2 * h for h in self.h[:-1]]
        self.ldin = [ld0] + [2 * h for h in self.h[:-1]]
        self.rows = [self.off[L - l] for l in range(L)]                     # x rows of level l
        self.nrows = [self.off[L - l + 1] - self.off[1] for l in range(L)]  # neighbour rows of level l
        assert all(d % 64 == 0 for d in self.din[1:]), "hidden widths must be multiples of 32"
        descs = []

        def copies(prm, need_t, packed=False):
            r, c = prm.shape
            w = torch.zeros(r, _r64(c), dtype=bf, device=dev)
            wt = torch.zeros(c, _r64(r), dtype=bf, device=dev) if need_t else None
            # forward operands of K3 / K5 also in MFMA fragment order (gsage_*_packed)
            wp = (torch.zeros(nat.lib().gsage_packed_weight_elems(r, c, 1), dtype=bf, device=dev)
                  if packed and is_bf else None)
            descs.append(_PrepDesc(prm.data_ptr(), w.data_ptr(), wt.data_ptr() if need_t else None, r, c,
                                   w.shape[1], wt.shape[1] if need_t else 0,
                                   wp.data_ptr() if wp is not None else None, 4 * (-(-c // 64)),
                                   int(not is_bf), 0))
            return (w, wt, wp) if packed else (w, wt)
        self.wm, self.wx, self.wn, self.wmT, self.wxT, self.wnT = [], [], [], [], [], []
        self.wm_p, self.wx_p, self.wn_p = [], [], []
        for l, layer in enumerate(self.layers):
            order = [self.pidx[id(p)] for p in (layer.mlp[0].weight, layer.mlp[0].bias, layer.fc_x.weight,
                                                layer.fc_neib.weight)]
            assert order == list(range(order[0], order[0] + 4)), "unexpected parameter order"
            ing = l > 0 or self.emb                  # does this level's input need a gradient?
            wm, wmT, wm_p = copies(layer.mlp[0].weight, ing, True)
            wx, wxT, wx_p = copies(layer.fc_x.weight, ing, True)
            wn, wnT, wn_p = copies(layer.fc_neib.weight, True, True)
            self.wm.append(wm); self.wmT.append(wmT); self.wx.append(wx); self.wxT.append(wxT)
            self.wn.append(wn); self.wnT.append(wnT)
            self.wm_p.append(wm_p); self.wx_p.append(wx_p); self.wn_p.append(wn_p)
        if self.emb:
            self._init_emb(lambda prm, need_t: copies(prm, need_t))
        self.descs = torch.frombuffer(bytearray(bytes((_PrepDesc * len(descs))(*descs))), dtype=torch.uint8).to(dev)
        self.n_desc = len(descs)
        self.max_elems = max(d.rows * d.cols for d in descs)

        # level-0 operands: x rows (hops 0..L-1) gathered once per step; the neighbour rows (hops 1..L: 141 k rows,
        # 180 MB at Reddit's shape) are read IN PLACE through the frontier's row list by K3 and by K5b
        # (gsage_wgrad_desc.a_rows; its list starts at entry B of the frontier and must be 16-byte aligned: even B) --
        # GSAGE_POOL_COPY_ROWS=1 brings back the gathered copy
        # (embedding prep: ONE buffer per batch in flight holds the prep's output for every row of the frontier)
        self.inplace0 = os.environ.get("GSAGE_POOL_COPY_ROWS", "0") != "1" and self.B % 2 == 0 and not self.emb
        if self.emb:
            self.g0_set = [torch.zeros(self.off[L + 1], ld0, dtype=bf, device=dev) for _ in range(self.nset)]
            self.x0_set = self.xn0_set = [None] * self.nset
        else:
            self.x0_set = [torch.zeros(self.rows[0], feats.ld, dtype=bf, device=dev) for _ in range(self.nset)]
            self.xn0_set = [None if self.inplace0 else torch.zeros(self.nrows[0], feats.ld, dtype=bf, device=dev)
                            for _ in range(self.nset)]
        self._q_ids = None
        self.pooled, self.pooled_b, self.argmax, self.hout, self.dc = [], [], [], [], []
        self.dpool, self.ghc, self.dxb, self.dnb, self.bpart = [], [], [], [], []
        for l in range(L):
            R, NR, Hm, h, din = self.rows[l], self.nrows[l], self.Hm[l], self.h[l], self.din[l]
            last = l == L - 1
            self.pooled.append(torch.zeros(R, Hm, dtype=f32, device=dev))
            # operand copy of `pooled` for the fc_neib projection and its weight gradient (parity mode:
            # the fp32 result itself)
            self.pooled_b.append(torch.zeros(R, _r64(Hm), dtype=bf, device=dev) if is_bf else self.pooled[l])
            # what the backward needs of the hidden layer: the winning row (max) / the ReLU sign bits (mean)
            self.argmax.append(torch.zeros(R, Hm, dtype=i32, device=dev) if self.pool_mode == nat.POOL_MAX
                               else torch.zeros(NR, Hm // 32, dtype=i32, device=dev))
            self.hout.append(torch.zeros(R, 2 * h, dtype=f32 if last else bf, device=dev))
            self.dc.append(torch.zeros(R, 2 * h, dtype=bf, device=dev))
            self.dpool.append(torch.zeros(R, Hm, dtype=f32, device=dev))
            self.ghc.append(torch.zeros(NR, Hm, dtype=bf, device=dev))
            if l == 0 and self.emb:      # the prep's E columns only (features take no gradient); the neighbour rows'
                E = self.E                # gradient sits at its rows of the frontier, above zeros for the seeds
                self.dn_all = torch.zeros(self.off[L + 1], E, dtype=f32, device=dev)
                self.dzero = torch.zeros(R, E, dtype=f32, device=dev)
                self.dxb.append(torch.zeros(R, E, dtype=f32, device=dev))
                self.dnb.append(self.dn_all[self.off[1]:])
            else:
                self.dxb.append(torch.zeros(R, din, dtype=f32, device=dev) if l > 0 else None)
                self.dnb.append(torch.zeros(NR, din, dtype=f32, device=dev) if l > 0 else None)
            self.bpart.append(torch.zeros((L - l) * self.NPART, Hm, dtype=f32, device=dev))

    def _x_operand(self, l, s):
        if l == 0 and self.emb:
            return self.g0_set[s], self.ldin[0]
        return (self.x0_set[s], self.store.ld) if l == 0 else (self.hout[l - 1], self.din[l])

    def _nb_operand(self, l, s):
        """neighbour rows of level l: (row block, leading dimension, row list or None).  With a row list, neighbour
        row i is block[list[i]] (level 0 read in place from the feature table)."""
        if l == 0 and self.emb:
            return self.g0_set[s][self.off[1]:], self.ldin[0], None
        if l == 0:
            if self.inplace0:       # the frontier of the batch being computed: the queue's, else the set's own
                ids = self._q_ids if self._q_ids is not None else self.ids_set[s]
                return self.store.data, self.store.ld, ids[self.off[1]:]
            return self.xn0_set[s], self.store.ld, None
        return self.hout[l - 1][self.off[1]:], self.din[l], None

    def _wg_shapes(self, l):
        """(key, parameter, M, Ntot, K) of level l's three weight gradients, in the order they are issued"""
        layer = self.layers[l]
        R, NR, Hm, h, din = self.rows[l], self.nrows[l], self.Hm[l], self.h[l], self.din[l]
        return (("x", layer.fc_x.weight, R, h, din), ("n", layer.fc_neib.weight, R, h, Hm),
                ("m", layer.mlp[0].weight, NR, Hm, din))

    def _init_reduce(self):
        dev, L, f32 = self.dev, self.L, torch.float32
        # K5b workgroups per problem: the MLP's weight gradients get the chip's worth of slices each (the small one's
        # many short workgroups fill the tail of the big one's), the two projections few, long ones (fewer partial
        # tiles to write and to sum).  Sizing all six for ONE round of equal-length slices (ops.wgrad_balance: 232
        # workgroups, the big problem on 200) measured SLOWER here -- 180 against 150 us: the launch is bound by what
        # its 240 big workgroups pull from HBM (785 MB per launch), and fewer, longer slices only stretch that.
        self.wg_target = {(l, key): {"m": 240, "x": 40, "n": 40}[key] for l in range(L) for key in "mxn"}
        for l in range(L):
            # two output tiles per workgroup (gsage_wgrad_pair_ok): slices half as long keep the workgroup count
            for key, _prm, M, ntot, K in self._wg_shapes(l):
                t = self.wg_target[(l, key)]
                if nat.lib().gsage_wgrad_pair_ok(self.code, M, ntot, ntot, ops.wgrad_plan(M, ntot, K, 2 * t)[0]):
                    self.wg_target[(l, key)] = 2 * t
        rdesc, self.slabs = [], []
        for l, layer in enumerate(self.layers):
            Hm = self.Hm[l]
            bufs = {}
            for key, prm, M, ntot, K in sorted(self._wg_shapes(l), key=lambda spec: "mxn".index(spec[0])):
                rps, S, ldk = ops.wgrad_plan(M, ntot, K, self.wg_target[(l, key)])
                buf = torch.zeros(S, ntot, ldk, dtype=f32, device=dev)
                bufs[key] = buf
                rdesc.append(_ReduceDesc(buf.data_ptr(), ntot * ldk, self.poff[self.pidx[id(prm)]], S, ntot, K, ldk))
            ib = self.pidx[id(layer.mlp[0].bias)]
            rdesc.append(_ReduceDesc(self.bpart[l].data_ptr(), Hm, self.poff[ib], self.bpart[l].shape[0], 1, Hm, Hm))
            self.slabs.append(bufs)
        if self.emb:                                  # prep.fc: weight through K5b, bias through column sums
            dC, A, lda, M_, ntot, K, prm, _rows = self._emb_wgrad_problem()
            self.wg_target[("prep", 0)] = 40
            rps, S, ldk = ops.wgrad_plan(M_, ntot, K, 40)
            self.slab_prep = torch.zeros(S, ntot, ldk, dtype=f32, device=dev)
            rdesc.append(_ReduceDesc(self.slab_prep.data_ptr(), ntot * ldk, self.poff[self.pidx[id(prm)]], S, ntot, K, ldk))
            rdesc.append(self._emb_reduce_desc())
        self._install_reduce(rdesc)
        if self.emb:
            self._init_emb_optimizer()

    # ---- stages ----------------------------------------------------------------------------------------
    def _stage_gather(self, s, with_adam=False, ids=None, hops=None, skip_rows=0):
        L, st = self.L, self.store
        if self.emb:
            return               # nothing to gather ahead: the level-0 rows are weights (read after Adam)
        if ids is None:
            ids = self.ids_set[s]
        segs = [(st.data, ids[:self.rows[0]], self.x0_set[s], self.rows[0], 1)]
        if not self.inplace0:
            segs.append((st.data, ids[self.off[1]:self.off[L + 1]], self.xn0_set[s], self.nrows[0], 1))
        # (D = the real width: the pad columns of the operand buffers were zeroed once and stay zero)
        ops.gather_mean_multi(segs, st.ld, st.dim, st.ld, adam=self._adam_desc() if with_adam else None,
                              hops=hops)

    def _gemm(self, A, lda, W, C, c_code, ldc, M, N, K, act, Wp=None):
        if Wp is not None and lda % 64 == 0 and lda >= -(-K // 64) * 64:
            ops._linear_packed_launch(A, lda, None, 0, Wp.data_ptr(), None, C, ldc, M, N, K, act, 1, 0, 0, c_code)
            return
        ops._linear_launch(A, lda, None, 0, W.data_ptr(), W.shape[1], None, C, ldc, M, N, K, act, 1, 0, 0, 0,
                           self.code, c_code)

    def _stage_compute(self, s):
        L, B, lib, stream, m = self.L, self.B, nat.lib(), ops._stream(), self.model
        if self.emb:
            self._cur_ids = self._q_ids if self._q_ids is not None else self.ids_set[s]
            self._prep_forward(s)
        for l, layer in enumerate(self.layers):
            R, Hm, h, din = self.rows[l], self.Hm[l], self.h[l], self.din[l]
            nb, ldnb, nrows = self._nb_operand(l, s)
            for k in range(L - l):                       # one K3 launch per hop: its fan-out is the segment
                r0, r1 = self.off[k], self.off[k + 1]
                a0 = self.off[k + 1] - self.off[1]
                a_ptr = nb.data_ptr() if nrows is not None else nb[a0:].data_ptr()
                r_ptr = nrows[a0:].data_ptr() if nrows is not None else None
                is_max = self.pool_mode == nat.POOL_MAX
                is_bf = self.code == nat.BF16
                tail = (self.pooled[l][r0:r1].data_ptr(), Hm, self.argmax[l][r0:r1].data_ptr() if is_max else None,
                        self.pooled_b[l][r0:r1].data_ptr() if is_bf else None, self.pooled_b[l].shape[1],
                        None if is_max else self.argmax[l][a0:].data_ptr(), stream)
                if l == 0 and k == L - 1:
                    self._time_next(4, 5)
                if self.wm_p[l] is not None and ldnb % 64 == 0 and ldnb >= -(-din // 64) * 64:
                    nat.check(lib.gsage_pool_mlp_packed(
                        a_ptr, ldnb, r_ptr, self.wm_p[l].data_ptr(), layer.mlp[0].bias.data_ptr(),
                        self.size[k], self.fan[k + 1], Hm, din, self.pool_mode, *tail), "pool_mlp_packed")
                else:
                    nat.check(lib.gsage_pool_mlp(
                        a_ptr, self.code, ldnb, r_ptr, self.wm[l].data_ptr(), self.wm[l].shape[1],
                        layer.mlp[0].bias.data_ptr(), self.size[k], self.fan[k + 1], Hm, din, self.pool_mode,
                        *tail), "pool_mlp")
            x, ldx = self._x_operand(l, s)
            last = l == L - 1
            out, code = self.hout[l], (nat.F32 if last else self.code)
            act = nat.ACT_NONE if last else nat.ACT_RELU
            self._gemm(x.data_ptr(), ldx, self.wx[l], out.data_ptr(), code, 2 * h, R, h, din, act, self.wx_p[l])
            self._gemm(self.pooled_b[l].data_ptr(), self.pooled_b[l].shape[1], self.wn[l],
                       out.data_ptr() + h * out.element_size(), code, 2 * h, R, h, Hm, act, self.wn_p[l])
        self._stage_head(s)
        if self.eval_only:
            return                                    # (forward only: train.evaluate's folds)
        self._backward_levels(s)

    def _backward_levels(self, s):
        L, lib, stream = self.L, nat.lib(), ops._stream()
        for l in range(L - 1, -1, -1):
            R, NR, Hm, h, din = self.rows[l], self.nrows[l], self.Hm[l], self.h[l], self.din[l]
            dc = self.dc[l]
            # d pooled = dC[:, h:] Wn
            self._gemm(dc.data_ptr() + h * self.esz, 2 * h, self.wnT[l], self.dpool[l].data_ptr(), nat.F32, Hm, R, Hm,
                       h, nat.ACT_NONE)
            for k in range(L - l):
                r0, r1 = self.off[k], self.off[k + 1]
                a0 = self.off[k + 1] - self.off[1]
                if self.pool_mode == nat.POOL_MEAN:
                    nat.check(lib.gsage_pool_route_mean_bwd(self.dpool[l][r0:r1].data_ptr(), Hm,
                                                            self.argmax[l][a0:].data_ptr(), self.size[k],
                                                            self.fan[k + 1], Hm, self.ghc[l][a0:].data_ptr(), self.code,
                                                            Hm, self.bpart[l][k * self.NPART:].data_ptr(), self.NPART,
                                                            stream), "pool_route_mean_bwd")
                    continue
                nat.check(lib.gsage_pool_route_bwd(self.dpool[l][r0:r1].data_ptr(), Hm, self.pooled[l][r0:r1].data_ptr(),
                                                   Hm, self.argmax[l][r0:r1].data_ptr(), Hm, self.size[k],
                                                   self.fan[k + 1], Hm, self.ghc[l][a0:].data_ptr(), self.code, Hm,
                                                   stream), "pool_route_bwd")
                nat.check(lib.gsage_pool_bias_partials(self.dpool[l][r0:r1].data_ptr(), Hm,
                                                       self.pooled[l][r0:r1].data_ptr(), Hm, self.size[k], Hm,
                                                       self.bpart[l][k * self.NPART:].data_ptr(), self.NPART,
                                                       stream),
                          "pool_bias_partials")
            if l > 0:
                self._gemm(dc.data_ptr(), 2 * h, self.wxT[l], self.dxb[l].data_ptr(), nat.F32, din, R, din, h,
                           nat.ACT_NONE)
                self._gemm(self.ghc[l].data_ptr(), Hm, self.wmT[l], self.dnb[l].data_ptr(), nat.F32, din, NR, din,
                           Hm, nat.ACT_NONE)
                below = self.hout[l - 1]
                nat.check(lib.gsage_pool_merge_bwd(below.data_ptr(), self.code, below.stride(0), self.dxb[l].data_ptr(),
                                                   din, R,
                                                   self.dnb[l].data_ptr(), din, self.off[1],
                                                   self.dc[l - 1].data_ptr(), self.dc[l - 1].stride(0),
                                                   self.rows[l - 1], din, stream), "pool_merge_bwd")
            if l == 0 and self.emb:
                # level 0's input is the prep's output: dX (x rows) + dN (neighbour rows, at their rows of the frontier),
                # only through the prep's columns [D0, D0 + E) of the transposed copies, no ReLU below
                E, D0, RA0 = self.E, self.D0, self.off[L + 1]
                self._gemm(dc.data_ptr(), 2 * h, self.wxT[0][D0:D0 + E], self.dxb[0].data_ptr(), nat.F32, E, R, E, h,
                           nat.ACT_NONE)
                self._gemm(self.ghc[0].data_ptr(), Hm, self.wmT[0][D0:D0 + E], self.dnb[0].data_ptr(), nat.F32, E, NR, E,
                           Hm, nat.ACT_NONE)
                if self.rows_ok:    # (one row pipeline: common._prep_backward_rows)
                    self._prep_backward_rows(s, None, None, self.dn_all.data_ptr(), E, self.dxb[0].data_ptr(), E, R,
                                             self.dzero.data_ptr(), E, None)
                else:
                    lp = self.din0 is not self.din0f
                    nat.check(lib.gsage_attn_merge_bwd2(
                        None, self.code, 0, self.dn_all.data_ptr(), E, self.dxb[0].data_ptr(), E, R, self.dzero.data_ptr(), E,
                        None, self.din0f.data_ptr(), nat.F32, E, RA0, E, L + 1, self.off_host, self.fan_host,
                        self.din0.data_ptr() if lp else None, self.din0.stride(0) if lp else 0, stream), "merge_bwd (level 0)")
                    self._prep_backward(s)
        probs = []
        if self.emb:
            dC, A, lda, M_, ntot, K, _prm, _rows = self._emb_wgrad_problem()
            probs.append((dC, A, lda, 0, M_, ntot, K, ntot, self.slab_prep, self.wg_target[("prep", 0)], None))
        for l in range(L - 1, -1, -1):
            R, NR, Hm, h, din = self.rows[l], self.nrows[l], self.Hm[l], self.h[l], self.din[l]
            x, ldx = self._x_operand(l, s)
            nb, ldnb, nrows = self._nb_operand(l, s)
            dc = self.dc[l]
            T = self.wg_target
            probs.append((dc[:, :h], x, ldx, 0, R, h, din, h, self.slabs[l]["x"], T[(l, "x")]))
            probs.append((dc[:, h:], self.pooled_b[l], self.pooled_b[l].shape[1], 0, R, h, Hm, h, self.slabs[l]["n"],
                          T[(l, "n")]))
            probs.append((self.ghc[l], nb, ldnb, 0, NR, Hm, din, Hm, self.slabs[l]["m"], T[(l, "m")], nrows))
        for i in range(0, len(probs), 8):
            if i == 0:
                self._wgrad_ticks()
                self._time_next(6, 7)
            ops.wgrad_multi(probs[i:i + 8])
        self._stage_finalize(s)
