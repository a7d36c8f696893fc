"""
engine/mean.py -- FusedMeanTrainStep: the north-star configuration (BASELINE configs[1], [4]): mean aggregators
(reference nn_modules.py:185-204) over an identity prep and a bf16 / fp32 FeatureStore.
"""
import ctypes
import os

import torch

from .. import _native as nat
from .. import ops
from ..nn_modules import IdentityPrep, MeanAggregator
from .common import FusedTrainStep, _PrepDesc, _ReduceDesc, _r8


class FusedMeanTrainStep(FusedTrainStep):
    """train_step (reference models.py:97-104) for the north-star configuration -- sparse sampler,
    identity prep over a bf16 FeatureStore, mean aggregators (ReLU on all but the last layer) --
    without autograd below the loss and without framework glue kernels.  For the 2-layer Reddit
    shape a step is FIVE launches, recorded once into a native command list (or a hipGraph) and
    replayed per batch:

        K5            level 0: ONE grouped MFMA GEMM (x | agg against Wx | Wn), bf16 out
        seed level    segment mean + both projections + normalize/fc/CE + all gradients down to the
                      level-0 activations in one kernel (gsage_mean_tail_mfma: 16 seeds per workgroup on
                      the matrix cores; gsage_mean_tail_ce for fp32 storage; generic models use K2 + K5
                      + gsage_head_ce + K5/merge per level instead); in queue mode the CUs its B / 16
                      workgroups leave idle gather the NEXT batch's last-hop means (at B = 512: all of
                      them) and, where it fits, sample the batch after that (K1: _k1_in_tail)
        K5b           every level's weight gradient in one grouped launch (partial tiles -> slabs)
        finalise      partial tiles + head partials -> flat gradient bucket + norm partials; ticks the
                      step's device counters
        [RCCL]        one all-reduce of the flat gradient bucket (data-parallel runs only)
        Adam          clip + Adam + refresh of the bf16 operand copies, side by side with what is left of
                      the level-0 gathers of the NEXT batch (the neighbour means of the other hops; the x
                      rows are read in place by K5 / K5b) and -- unless the seed-level launch carried it --
                      with K1 (all hops) for the batch after that (queue mode; otherwise K1 and the
                      gathers open the step as launches of their own)

    The arithmetic is that of GSSupervised.train_step.  Parameters and gradients live in flat fp32
    buckets; the model's Parameters become views of them, so `model.state_dict()`, evaluation and
    checkpointing keep working.  `__call__(ids, targets)` has the contract of train_step;
    `load_epoch()` + `step_queue()` walk a device-resident queue of seed batches with no host copies.
    """

    MEAN_ENGINE = True
    TIMED = {"gather": (0, 1), "seed_level": (2, 3), "k5": (4, 5), "k5b": (6, 7)}


    @classmethod
    def why_not(cls, model, feats, ddp=None):
        # bf16 storage = the production path; fp32 storage = the exact-arithmetic parity mode (same engine, same
        # kernel sources instantiated on fp32: golden fixtures replay at 2e-4).  With the node-embedding prep
        # (utils/pokec.sh:5-13) the level-0 rows are weights: computed per step, nothing is gathered ahead.
        why = cls._why_not_common(model, feats, (MeanAggregator,), "mean") or cls._why_not_input(model, feats, ddp, concat_ok=True)
        if why:
            return why
        if not all(l.output_dim_ % 8 == 0 for l in model.agg_layers.children()):
            return "output dims that are not multiples of 8"
        return None

    def _init_levels(self, example_ids, example_targets):
        """Mean aggregator: per-level shapes, bf16 operand copies (+ their refresh descriptors) and
        work buffers."""
        model, feats, dev, L, B = self.model, self.store, self.dev, self.L, self.B
        # ---- per-level shapes, operand copies and work buffers ------------------------------
        self.h = [l.output_dim_ for l in self.layers]
        if self.emb:                                 # level-0 rows = [features |] prep.fc(embedding[ids])
            d0 = self.D0 + self.E
            ld0 = _r8(d0) if (self.tdt == torch.bfloat16 or self.D0) else d0
        else:
            d0, ld0 = feats.dim, feats.ld
        self.din = [d0] + [2 * h for h in self.h[:-1]]
        self.ldin = [ld0] + [2 * h for h in self.h[:-1]]
        self.rows = [self.off[L - l] for l in range(L)]           # R_l = rows of level l
        self.w2, self.w2t, self.wp, descs = [], [], [], []
        for l, layer in enumerate(self.layers):
            h, din = self.h[l], self.din[l]
            assert tuple(layer.fc_x.weight.shape) == (h, din) == tuple(layer.fc_neib.weight.shape)
            ix, inb = self.pidx[id(layer.fc_x.weight)], self.pidx[id(layer.fc_neib.weight)]
            assert inb == ix + 1, "fc_x / fc_neib must be adjacent in the parameter order"
            w2 = torch.zeros(2, h, _r8(din), dtype=self.tdt, device=dev)
            w2t = torch.zeros(2, din, _r8(h), dtype=self.tdt, device=dev) if (l > 0 or self.emb) else None
            tail_level = self._will_fuse_tail(example_targets) and l == L - 1
            self.w2.append(w2)
            self.w2t.append(w2t)
            # levels whose forward runs on K5 read the weights in MFMA fragment order
            # (gsage_linear_nt_packed; needs whole-line operand rows); the seed-level kernel reads w2
            lda = self.ldin[l]
            # the fragment-ordered copy feeds K5's forward; the seed-level kernel reads w2 / w2t instead
            packed = (self.code == nat.BF16 and lda % 64 == 0 and lda >= -(-din // 64) * 64 and not tail_level)
            gstride = nat.lib().gsage_packed_weight_elems(h, din, 1)
            wp = torch.zeros(2 * gstride, dtype=torch.bfloat16, device=dev) if packed else None
            self.wp.append(wp)
            for g, prm in enumerate((layer.fc_x.weight, layer.fc_neib.weight)):
                # (a copy nobody reads is not refreshed: w2 serves the unpacked K5 and the seed-level kernel)
                descs.append(_PrepDesc(prm.data_ptr(), w2[g].data_ptr() if (not packed or tail_level) else None,
                                       w2t[g].data_ptr() if w2t is not None else None,
                                       h, din, w2.shape[2], w2t.shape[2] if w2t is not None else 0,
                                       wp[g * gstride:].data_ptr() if packed else None, 4 * (-(-din // 64)),
                                       int(self.code == nat.F32), 0))
        if self.emb:
            def copies(prm, need_t):                              # operand copies of prep.fc.weight
                r, c = prm.shape
                w = torch.zeros(r, _r8(c), dtype=self.tdt, device=dev)
                wt = torch.zeros(c, _r8(r), dtype=self.tdt, device=dev) if need_t else None
                descs.append(_PrepDesc(prm.data_ptr(), w.data_ptr(), wt.data_ptr() if need_t else None, r, c,
                                       w.shape[1], wt.shape[1] if need_t else 0, None, 0, int(self.code == nat.F32), 0))
                return w, wt
            self._init_emb(copies)
        raw = bytes((_PrepDesc * len(descs))(*descs))
        self.descs = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        self.n_desc = len(descs)
        self.max_elems = max(d.rows * d.cols for d in descs)

        bf, f32 = self.tdt, torch.float32
        # per-batch inputs of the compute stage, one set per batch in flight:
        #   ids   the concatenated frontier [hop 0 | hop 1 | ... | hop L]
        #   xa0   level-0 operands [x rows | neighbour means], gathered ONCE per step so the forward
        #         GEMM and the weight-gradient kernel both read plain row-major operands
        # (embedding prep: xa0[0] holds the prep's output for EVERY row of the frontier, computed per step)
        r0 = self.off[L + 1] if self.emb else self.rows[0]
        self.xa0_set = [torch.zeros(2, r0, ld0, dtype=bf, device=dev) for _ in range(self.nset)]
        self.g0_set = [xa[0] for xa in self.xa0_set]
        # K5 and K5b read the x rows of level 0 in place through the frontier's row list (gsage_linear_nt_packed
        # a_rows / gsage_wgrad_desc.a_rows) instead of from xa0[0]: no row copies in the gather launch (28.5 vs
        # 32.9 us in-step, 0.092 vs 0.095 ms/step at config 2; GSAGE_MEAN_INPLACE_X=0 brings the copies back)
        self.inplace_x = (os.environ.get("GSAGE_MEAN_INPLACE_X", "1") == "1" and self.L >= 2 and not self.emb)
        #                                                               (one level: the copies ARE the "rest" launch
        #                                                                that carries Adam in data-parallel runs)
        self.agg, self.hout, self.dc, self.dg = [], [], [], []
        for l in range(L):
            R = self.rows[l]
            ld = self.ldin[l]
            assert ld % 8 == 0
            self.agg.append(None if l == 0 else torch.zeros(R, ld, dtype=bf, device=dev))
            last = l == L - 1
            self.hout.append(torch.zeros(R, 2 * self.h[l], dtype=f32 if last else bf, device=dev))
            self.dc.append(torch.zeros(R, 2 * self.h[l], dtype=bf, device=dev))
            # (level 0 under the embedding prep: only the prep's E columns of the row have a gradient to pass on)
            self.dg.append(torch.zeros(R, 2 * (self.din[l] if l > 0 else self.E), dtype=f32, device=dev)
                           if (l > 0 or self.emb) else None)

    def _will_fuse_tail(self, example_targets):
        """The seed level (segment mean + GEMM + head + input gradients + mask/route) as ONE kernel?"""
        L = self.L
        hs = [l.output_dim_ for l in self.layers]
        return bool(self._will_fuse_head(example_targets) and L >= 2 and
                    2 * hs[L - 2] == 256 and 2 * hs[L - 1] == 256 and self.fan[1] <= 32 and
                    os.environ.get("GSAGE_NO_FUSED_TAIL", "0") != "1")

    def _tail_on_mfma(self):
        """... and on the matrix cores (gsage_mean_tail_mfma: 16 seeds per 512-thread workgroup; bf16 storage)?
        GSAGE_TAIL_MFMA=0: the 4-seeds-per-workgroup VALU kernel (gsage_mean_tail_ce; the fp32 parity mode's)."""
        return self.code == nat.BF16 and os.environ.get("GSAGE_TAIL_MFMA", "1") == "1"


    def _init_reduce(self):
        """Gradient partial buffers + the descriptor table gsage_finalize_grads sums them with."""
        model, dev, L = self.model, self.dev, self.L
        f32 = torch.float32
        rdesc, self.slabs = [], []
        # K5b workgroups per problem.  Two layers: every problem aims at the chip (196 + 16 workgroups at Reddit's
        # shapes).  Deeper models would ask for more workgroups than there are CUs (one fits per CU: config 5's three
        # levels 240 + 128 + 8) and run in rounds: those are sized together (ops.wgrad_balance), in issue order.
        def parts_of(l):
            return [(2 * self.h[l], 0)] if self.h[l] % 128 == 0 else [(self.h[l], 0), (self.h[l], 1)]
        order = [("prep", 0)] if self.emb else []
        order += [(l, g) for l in range(L - 1, -1, -1) for _nt, g in parts_of(l)]
        def shape_of(key):
            if key[0] == "prep":
                _dC, _A, _lda, M_, ntot, K, _prm, _rows = self._emb_wgrad_problem()
                return (M_, ntot, K)
            return (self.rows[key[0]], parts_of(key[0])[0][0], self.din[key[0]])
        shapes = [shape_of(k) for k in order]
        plain = [self._wg_target()] * len(order)
        def n_wg(targets):
            return sum((ops.wgrad_plan(m, nt, k, t)[1]) * ((nt + 127) // 128) * ((k + 127) // 128)
                       for (m, nt, k), t in zip(shapes, targets))
        cus = int(torch.cuda.get_device_properties(dev).multi_processor_count)
        balanced = n_wg(plain) > cus
        self.wg_target = dict(zip(order, ops.wgrad_balance(shapes, budget=cus - 8) if balanced else plain))
        for l in range(L):
            h, din, R = self.h[l], self.din[l], self.rows[l]
            ix = self.pidx[id(self.layers[l].fc_x.weight)]
            parts = parts_of(l)
            bufs = []
            for ntot, g in parts:
                rps, S, ldk = ops.wgrad_plan(R, ntot, din, self.wg_target[(l, g)])
                buf = torch.zeros(S, ntot, ldk, dtype=f32, device=dev)
                bufs.append(buf)
                rdesc.append(_ReduceDesc(buf.data_ptr(), ntot * ldk, self.poff[ix + g], S, ntot, din, ldk))
            self.slabs.append(bufs)
        if self.emb:                                  # prep.fc: weight through K5b, bias through column sums
            dC, A, lda, M_, ntot, K, prm, _rows = self._emb_wgrad_problem()
            rps, S, ldk = ops.wgrad_plan(M_, ntot, K, self.wg_target[("prep", 0)])
            self.slab_prep = torch.zeros(S, ntot, ldk, dtype=f32, device=dev)
            rdesc.append(_ReduceDesc(self.slab_prep.data_ptr(), ntot * ldk, self.poff[self.pidx[id(prm)]], S, ntot, K, ldk))
            rdesc.append(self._emb_reduce_desc())
        self._install_reduce(rdesc)
        if self.emb:
            self._init_emb_optimizer()

    def _wg_target(self):
        """K5b workgroups to plan for: the chip (one workgroup fits per CU)"""
        return int(os.environ.get("GSAGE_WGRAD_TARGET", "240"))

    def _stage_gather(self, s, with_adam=False, ids=None, hops=None, skip_rows=0, part=None, adam=None, stop_rows=None):
        """Level-0 gathers of batch set s (x rows of every hop + each hop's neighbour means), one
        launch; with_adam: the clip + Adam update of the batch just finished rides along; hops: so
        does the sampling of a later batch's frontier (a gsage_hops_desc writing ANOTHER buffer).
        ids: frontier to gather from (default: the set's own).  part: "means" = only the last hop's
        neighbour means (the bulk: what runs while a gradient exchange is in flight), "rest" = the other
        segments (what then shares a launch with Adam and the sampler); None = everything."""
        L, st = self.L, self.store
        if self.emb:
            return False         # nothing to gather ahead: the level-0 rows are weights (read after Adam)
        if ids is None:
            ids = self.ids_set[s]
        # one launch: x rows of every hop + the mean of each hop's sampled neighbours
        R = self.rows[0]
        xa = self.xa0_set[s]
        # work items are dealt out in segment order: the means first (many dependent loads per item,
        # smallest hop first), the row copies last so that short items fill the launch's tail
        # (tools/kbench.py gmulti: 36.1 us against 41.9 us for copies first at Reddit shapes)
        segs = []
        for k in range(L):
            # last hop: rows [skip_rows, stop_rows) -- the seed-level launch and the side section gathered the others
            n, r0 = self.fan[k + 1], (skip_rows if k == L - 1 else 0)
            r1 = self.size[k] if (k < L - 1 or stop_rows is None) else min(int(stop_rows), self.size[k])
            if r1 > r0 and (part is None or (part == "means") == (k == L - 1)):
                segs.append((st.data, ids[self.off[k + 1] + r0 * n:self.off[k + 1] + r1 * n],
                             xa[1][self.off[k] + r0:self.off[k] + r1], r1 - r0, n))
        if part != "means" and not self.inplace_x:
            segs.append((st.data, ids[:R], xa[0], R, 1))
        if not segs:
            return False
        # (D = the real width: the pad columns of the operand buffers were zeroed once and stay zero)
        if adam is None and with_adam:
            adam = self._adam_desc()
        ops.gather_mean_multi(segs, st.ld, st.dim, st.ld, adam=adam, hops=hops)
        return True

    def _stage_compute(self, s):
        """Forward GEMMs, head, backward; everything that needs the current weights."""
        L, B, st, lib = self.L, self.B, self.store, nat.lib()
        stream = ops._stream()
        esz = self.esz
        if self.emb:
            self._cur_ids = self._q_ids if self._q_ids is not None else self.ids_set[s]
            self._prep_forward(s)
        for l in range(L - 1 if self.fused_tail else L):
            R, h, din = self.rows[l], self.h[l], self.din[l]
            rows = None
            if l == 0 and self.emb:            # x rows = the first rows of the prep's output; means of its later rows
                xa, ld0 = self.xa0_set[s], self.ldin[0]
                xbuf, agg, lda = xa[0], xa[1], ld0
                segs = [(xa[0][self.off[k + 1]:self.off[k + 2]], None, xa[1][self.off[k]:self.off[k + 1]],
                         self.size[k], self.fan[k + 1]) for k in range(L)]
                ops.gather_mean_multi(segs, ld0, din, ld0)
            elif l == 0 and self.inplace_x:      # the x rows of every hop, read in place through the frontier
                xbuf, agg, lda = st.data, self.xa0_set[s][1], st.ld
                rows = (self._q_ids if self._q_ids is not None else self.ids_set[s]).data_ptr()
            elif l == 0:
                xbuf, agg, lda = self.xa0_set[s][0], self.xa0_set[s][1], st.ld
            else:
                xbuf, agg, lda = self.hout[l - 1], self.agg[l], din
                segs = [(xbuf[self.off[k + 1]:self.off[k + 2]], None, agg[self.off[k]:self.off[k + 1]],
                         self.size[k], self.fan[k + 1]) for k in range(L - l)]
                ops.gather_mean_multi(segs, din, din, din)
            delta = agg.data_ptr() - xbuf.data_ptr()
            assert delta % esz == 0 and agg.stride(0) == lda
            last = l == L - 1
            if self.wp[l] is not None:
                if l == 0 and getattr(self, "_k5_hops", None) is not None and self._k1_in_k5():
                    # the projection's spare workgroup slots sample the frontier of the batch after the next
                    nat.check(lib.gsage_hops_role_next(ctypes.addressof(self._k5_hops)), "hops_role_next")
                if l == 0:
                    self._time_next(4, 5)
                ops._linear_packed_launch(xbuf.data_ptr(), lda, rows, int(rows is not None), self.wp[l].data_ptr(), None,
                                          self.hout[l].data_ptr(), 2 * h, R, h, din,
                                          nat.ACT_NONE if last else nat.ACT_RELU, 2, delta // esz, h,
                                          nat.F32 if last else nat.BF16)
                continue
            self._linear(xbuf.data_ptr(), lda, rows, int(rows is not None), self.w2[l].data_ptr(), self.w2[l].shape[2],
                         self.hout[l].data_ptr(), nat.F32 if last else self.code, 2 * h, R, h, din,
                         nat.ACT_NONE if last else nat.ACT_RELU, delta // esz,
                         h * self.w2[l].shape[2], h)

        m = self.model
        if self.fused_tail:
            C = m.fc.weight.shape[0]
            tg = self.queue[1] if self.queue else self.tg_set[s].view(-1)
            self._time_next(2, 3)
            self._head_live_rows()
            args = (self.hout[L - 2].data_ptr(), B, self.fan[1], self.w2[L - 1].data_ptr(),
                    self.w2[L - 1].shape[2], self.w2t[L - 1].data_ptr(), self.w2t[L - 1].shape[2],
                    m.fc.weight.data_ptr(), m.fc.bias.data_ptr(), C, tg.data_ptr(),
                    self.batch_idx.data_ptr() if self.queue else None, self.queue[2] if self.queue else 0,
                    self.agg[L - 1].data_ptr(), self.dc[L - 1].data_ptr(), self.preds.data_ptr(),
                    self.dc[L - 2].data_ptr(), self.head_scratch.data_ptr(),
                    ctypes.addressof(self._tail_gather) if self._tail_gather is not None else None)
            if self._tail_on_mfma():
                if getattr(self, "_k5_hops", None) is not None and self._tail_gather is not None and self._k1_in_tail():
                    # workgroups behind the seed-level ones sample the frontier of the batch after the next
                    nat.check(lib.gsage_hops_role_next(ctypes.addressof(self._k5_hops)), "hops_role_next")
                nat.check(lib.gsage_mean_tail_mfma(*args, stream), "mean_tail_mfma")
            else:
                nat.check(lib.gsage_mean_tail_ce(*args, self.code, stream), "mean_tail_ce")
        else:
            self._stage_head(s)
        if self.eval_only:
            return                                    # (forward only: train.evaluate's folds)
        self._backward_levels(s)

    def _backward_levels(self, s):
        L, st, lib = self.L, self.store, nat.lib()
        stream = ops._stream()
        esz = self.esz
        # (1) the chain of input gradients down the levels: dC[l] -> (dX | dAgg) -> mask/route -> dC[l-1]
        for l in range(L - 1, 0, -1):
            if self.fused_tail and l == L - 1:
                continue                                  # k_mean_tail_ce wrote dC[L-2] already
            R, h, din = self.rows[l], self.h[l], self.din[l]
            w2t = self.w2t[l]
            # NT GEMM against the transposed operand copies
            self._linear(self.dc[l].data_ptr(), 2 * h, None, 0, w2t.data_ptr(), w2t.shape[2],
                         self.dg[l].data_ptr(), nat.F32, 2 * din, R, din, h, nat.ACT_NONE, h,
                         din * w2t.shape[2], din)
            below = self.hout[l - 1]
            nat.check(lib.gsage_bwd_merge(below.data_ptr(), self.code, below.stride(0), self.dg[l].data_ptr(),
                                          2 * din, din, self.dc[l - 1].data_ptr(),
                                          self.dc[l - 1].stride(0), self.rows[l - 1], R, din,
                                          L - l + 1, self.off_host, self.fan_host, stream),
                      "bwd_merge")
        if self.emb:
            # level 0's input is the prep's output: (dX | dAgg) = dC[0] against the transposed copies, merged into
            # one gradient row per frontier row (x rows: dX; children: dAgg of the parent / fan-out; no ReLU below).
            # With features in front ([features | prep output]) only rows [D0, D0 + E) of the transposed copies --
            # the prep's columns -- are multiplied: the features are data, they take no gradient.
            R, h, E, d0 = self.rows[0], self.h[0], self.E, self.din[0]
            w2t, dg = self.w2t[0], self.dg[0]
            self._linear(self.dc[0].data_ptr(), 2 * h, None, 0, w2t.data_ptr() + self.D0 * w2t.shape[2] * esz,
                         w2t.shape[2], dg.data_ptr(), nat.F32, 2 * E, R, E, h, nat.ACT_NONE, h, d0 * w2t.shape[2], E)
            if self.rows_ok:        # merge, bias sums, product through prep.fc^T and the table's atomics: one row pipeline
                self._prep_backward_rows(s, None, None, None, 0, dg.data_ptr(), 2 * E, R, dg.data_ptr() + 4 * E, 2 * E, None)
            else:
                lp = self.din0 is not self.din0f
                nat.check(lib.gsage_attn_merge_bwd2(
                    None, self.code, 0, None, 0, dg.data_ptr(), 2 * E, R, dg.data_ptr() + 4 * E, 2 * E, None,
                    self.din0f.data_ptr(), nat.F32, E, self.off[L + 1], E, L + 1, self.off_host, self.fan_host,
                    self.din0.data_ptr() if lp else None, self.din0.stride(0) if lp else 0, stream), "merge_bwd (level 0)")
                self._prep_backward(s)
        # (2) every level's weight gradient in ONE launch: each alone fills a fraction of the chip
        probs = []
        if self.emb:
            dC, A, lda, M_, ntot, K, _prm, _rows = self._emb_wgrad_problem()
            probs.append((dC, A, lda, 0, M_, ntot, K, ntot, self.slab_prep, self.wg_target[("prep", 0)], None))
        for l in range(L - 1, -1, -1):
            R, h, din = self.rows[l], self.h[l], self.din[l]
            dc = self.dc[l]
            xbuf, lda = (self.xa0_set[s][0], self.ldin[0]) if l == 0 else (self.hout[l - 1], din)
            rows = None
            if l == 0 and self.inplace_x:
                xbuf, rows = st.data, (self._q_ids if self._q_ids is not None else self.ids_set[s])
            aggl = self.xa0_set[s][1] if l == 0 else self.agg[l]
            delta = (aggl.data_ptr() - xbuf.data_ptr()) // esz
            if h % 128 == 0:
                probs.append((dc, xbuf, lda, delta, R, 2 * h, din, h, self.slabs[l][0], self.wg_target[(l, 0)], rows))
            else:
                for g in range(2):
                    probs.append((dc[:, g * h:], xbuf if g == 0 else aggl, lda, 0, R, h, din, h,
                                  self.slabs[l][g], self.wg_target[(l, g)], rows if g == 0 else None))
        for i in range(0, len(probs), 8):
            if i == 0:
                self._wgrad_ticks()
                self._time_next(6, 7)
            ops.wgrad_multi(probs[i:i + 8])
        self._stage_finalize(s)

    def gather_launch_rows(self):
        """(rows the queue-mode gather launch reads, rows the seed-level launch's gather role reads) per
        step: every sampled frontier row is read exactly once, by one of the two."""
        total = self.off[self.L + 1]
        tail = self._tail_rows * self.fan[self.L]
        if getattr(self, "inplace_x", False):          # K5 / K5b read the x rows themselves
            total -= self.rows[0]
        return total - tail, tail

    def _tail_gather_rows(self):
        """Rows of the last hop's neighbour means that the seed-level launch of the previous step
        gathers (0: none)."""
        if not self.fused_tail or self.emb or self.fan[self.L] not in (5, 10, 15) or self.code != nat.BF16:
            return 0
        mfma = self._tail_on_mfma()
        n_idle = self._tail_idle_cus(self._k1_in_tail())
        if n_idle < 32:
            return 0
        # what an idle CU moves while the launch lasts does not depend on B: ~480 KB in the VALU kernel's ~27 us,
        # i.e. ~40 means of ten 1.2 KB rows (sweep at B = 512, DESIGN.md section 3: 40 % of 12 800 rows on 128 CUs).
        # The matrix-core kernel's 32 workgroups take as long in the step (~28 us: each moves 770 KB through one CU)
        # but leave 224 CUs: ~60 means per CU, i.e. the WHOLE last hop at B = 512 (round-5 sweep, DESIGN.md section 5:
        # 0.0905 / 0.0883 / 0.0842 / 0.0837 ms/step at 30 / 40 / 50 / >= 60 per CU; the gather launch then carries only
        # the hop-1 means, Adam and the sampler).  Other row sizes / fan-outs get the same bytes per idle CU
        # (with the sampler role on 32 of those CUs the other 192 still take the whole last hop: 67 means each, the launch
        #  31.4 instead of 30.3 us and the launch that carries the update 11.8 instead of 13.3)
        frac = float(os.environ.get("GSAGE_TAIL_GATHER_FRAC", "0.7" if mfma else "0.4"))
        n_idle = max(32, min(n_idle, int(os.environ.get("GSAGE_TAIL_GATHER_WGS", n_idle))))     # (a sweep knob)
        self._tail_wgs = n_idle
        per_cu = 100.0 * frac * (10.0 * 1204.0) / (self.fan[self.L] * max(self.store.dim * self.esz, 256))
        return min(int(self.size[self.L - 1]), max(int(per_cu * n_idle), 0))

    def _tail_idle_cus(self, with_sampler):
        """CUs the seed-level launch leaves to its gather role: every workgroup of that launch owns a CU"""
        n_cu = int(torch.cuda.get_device_properties(self.dev).multi_processor_count)      # MI355X: 256
        n_idle = n_cu - ((self.B + 15) // 16 if self._tail_on_mfma() else (self.B + 3) // 4)
        if with_sampler:
            n_smp = self._tail_sampler_wgs()
            n_idle = n_idle - n_smp if n_smp > 0 else -1           # (0: the role does not fit -- no sampler role)
        return n_idle

    def _tail_sampler_wgs(self):
        widest, width = 1, 1
        for k in range(1, self.L + 1):
            width *= int(self.fan[k])
            widest = max(widest, width)
        return int(nat.lib().gsage_mean_tail_mfma_sampler_wgs(self.B, widest))

    def _k1_early(self):
        """data-parallel order: K1(i+2) rides in the launch that gathers the bulk of batch i+1 WHILE the exchange is in
        flight instead of in the launch that follows it (its ~9 us chain of dependent loads would otherwise sit on
        the critical path behind the collective, beside Adam).  Sampling reads no weights: same frontier either way."""
        return (not self.dense) and os.environ.get("GSAGE_DDP_K1_EARLY", "1") == "1"

    def _queue_front_means(self, par):
        self._stage_gather(self._qset(1 - par), ids=self.ids_q[1 - par], skip_rows=self._ahead_rows(), part="means",
                           hops=self._hops_desc(self.ids_q[par], True) if self._k1_early() else None)

    def _queue_front_rest(self, par):
        """after the exchange: the rest of the gathers || Adam (which forms the norm of the averaged gradient inside
        the launch) [|| K1, unless it rode with the bulk of the gathers]"""
        d = self._adam_desc()
        if not d.norm_slots:             # (a bucket too large for the in-launch norm: a norm pass of its own)
            n = self.flat_g.numel()
            n_sq = nat.lib().gsage_adam_partials(n)
            nat.check(nat.lib().gsage_grad_sqnorm(self.flat_g.data_ptr(), n, self.partial.data_ptr(), n_sq, ops._stream()),
                      "grad_sqnorm")
            d.n_partial_ready = n_sq
        self._stage_gather(self._qset(1 - par), ids=self.ids_q[1 - par], part="rest", adam=d,
                           hops=None if (self.dense or self._k1_early()) else self._hops_desc(self.ids_q[par], True))
        if self.dense:
            self._stage_sample(0, ids=self.ids_q[par], ahead=True)

    def _k1_in_k5(self):
        """Sample batch i+2 inside the launch of the level-0 projection of step i (gsage_hops_role_next) instead of
        in the launch that carries the update?  OPT-IN (GSAGE_K1_IN_K5=1): measured on the MI355X (round 5, DESIGN.md
        section 5) the launch that carries the update gets 1.6 us shorter (16.2 -> 14.6: what is left is the update's
        own chain) and the projection 2.8 us longer (14.0 -> 16.8: the sampler's 171 workgroups hold slots its 416
        want) -- 0.0846 against 0.0835 ms/step.  Needs the packed ReLU projection at level 0, a CSR sampler, one GPU;
        a ring of three frontier buffers (K5 / K5b of step i still read batch i's as their row list)."""
        if getattr(self, "_k1_where", None) is not None:      # (decided in load_epoch)
            return self._k1_where == "k5"
        return bool(self.wp and self.wp[0] is not None and self.L >= 2 and not self.dense and not self.emb and
                    self.ddp is None and os.environ.get("GSAGE_K1_IN_K5", "0") == "1")

    def _k1_in_tail(self):
        """Sample batch i+2 inside the SEED-LEVEL launch of step i (a sampler role behind its seed-level workgroups,
        gsage_hops_role_next) instead of in the launch that carries the update?  That launch lasts as long as its
        gather role (~31 us at config 2) whatever else rides in it; K1's chain of dependent loads was the longer of the
        last launch's two.  Needs the matrix-core seed level with a gather role, a CSR sampler, one GPU; a ring of three
        frontier buffers as for _k1_in_k5.  GSAGE_K1_IN_TAIL=0: K1 stays in the launch that carries the update."""
        if getattr(self, "_k1_where", None) is not None:      # (decided in load_epoch)
            return self._k1_where == "tail"
        return bool(self.fused_tail and self._tail_on_mfma() and self.L >= 2 and not self.dense and not self.emb and
                    self.ddp is None and self.fan[self.L] in (5, 10, 15) and not self._k1_in_k5() and
                    self._tail_idle_cus(True) >= 32 and
                    float(os.environ.get("GSAGE_TAIL_GATHER_FRAC", "0.7")) > 0 and
                    os.environ.get("GSAGE_K1_IN_TAIL", "1") == "1")

    def _queue_compute_body(self, par):
        nx = self._nx(par)
        if self.P == 3:
            self._k5_hops = self._hops_desc(self.ids_q[self._nx2(par)], 2)
        if self._tail_rows:
            L, st, nxt = self.L, self.store, self.ids_q[nx]
            d = nat.TailGatherDesc()
            d.table, d.ids = st.data.data_ptr(), nxt[self.off[L]:].data_ptr()
            d.out = self.xa0_set[nx][1][self.off[L - 1]:].data_ptr()
            d.ld, d.out_ld, d.D, d.rows = st.ld, st.ld, st.dim, self._tail_rows
            d.n, d.n_workgroups = self.fan[L], self._tail_wgs
            self._tail_gather = d
        try:
            self._stage_compute(self._qset(par))
        finally:
            self._tail_gather = None
            self._k5_hops = None

