"""
engine/attn.py -- FusedAttnTrainStep: attention aggregators (reference nn_modules.py:289-321) over feature rows
or over the trainable node-embedding prep of BASELINE configs[3] (nn_modules.py:126-155), with the deferred row
updates of the embedding table (gsage_rows_*).
"""
import ctypes
import os

import torch

from .. import _native as nat
from .. import ops
from ..nn_modules import AttentionAggregator, IdentityPrep, NodeEmbeddingPrep
from .common import FusedTrainStep, _PrepDesc, _ReduceDesc, _r64


class FusedAttnTrainStep(FusedTrainStep):
    """train_step for attention aggregators (reference nn_modules.py:289-321; BASELINE config 4's aggregator)
    with no autograd and no framework glue, on the machinery of FusedMeanTrainStep (flat buckets, fused
    multi-hop sampler, head kernel, finalisation + Adam, command lists, batch queue).  Level l turns the rows
    of hops 0 .. L-l ("In") into the rows of hops 0 .. L-l-1:

      forward    K5   hid = tanh(In W0^T);  K5  a = hid W2^T        att(.) ONCE per row: the reference applies the
                                                                    same MLP to a row as "x" and as a neighbour
                 K4   per hop: scores <a_child, a_parent>, softmax over the fan-out, agg = sum w * raw child row
                 K5   out[:, :h] = act(In[:rows_x] Wx^T);  K5  out[:, h:] = act(agg Wn^T)
      backward   K5   d agg = dC[:, h:] Wn
                 K4'  per hop: d a(child), d a(parent)  (softmax backward inside)
                 K5   (d a W2), tanh backward -> d hid;   l > 0: K5 d In(att) = d hid W0, K5 dX = dC[:, :h] Wx,
                      one merge kernel adds them to ws * d agg(parent) and applies the ReLU mask -> dC of level l-1
                 K5b  fc_x, fc_neib, att.0, att.2 weight gradients of every level in one grouped launch
    Level 0 reads its rows from ONE buffer gathered per step (all hops, next batch, beside Adam and K1): the att
    MLP, K4, the x projection and two of the four weight gradients all want plain row-major operands."""

    HA_LD = 64            # leading dimension of the 32-wide att activations (whole 128-byte bf16 lines)
    TIMED = {"gather": (0, 1), "k4": (4, 5), "k4_bwd": (6, 7)}   # K4 / K4' of level 0 over the LAST hop (the bulk)

    @classmethod
    def why_not(cls, model, feats, ddp=None):
        why = cls._why_not_common(model, feats, (AttentionAggregator,), "attention") or \
            cls._why_not_input(model, feats, ddp, concat_ok=True)
        if why:
            return why
        layers = list(model.agg_layers.children())
        if feats is not None and feats.ld % 8 != 0:
            return "feature rows that are not whole 16-byte chunks"
        if any(not (2 <= fn.keywords["n_samples"] <= 32) for fn in model.train_sample_fns) or len(layers) > 4:
            return "a fan-out outside 2..32 or more than four layers (K4 keeps a parent's softmax in one wave's lanes)"
        ha = {int(l.att[0].weight.shape[0]) for l in layers}
        if not (all(l.output_dim_ % 8 == 0 for l in layers) and ha <= {32} and
                all(tuple(l.att[2].weight.shape) == (32, 32) for l in layers)):
            return "an attention MLP that is not 32 wide, or output dims that are not multiples of 8"
        return None

    # ---- construction ------------------------------------------------------------------------------
    def _init_levels(self, example_ids, example_targets):
        feats, dev, L = self.store, self.dev, self.L
        T, f32 = self.tdt, torch.float32
        self.Ha = 32
        self.h = [l.output_dim_ for l in self.layers]
        if self.emb:                                 # level-0 rows = [features |] prep.fc(embedding[ids])
            d0 = self.D0 + self.E
            ld0 = _r64(d0) if T == torch.bfloat16 else (-(-d0 // 8) * 8 if self.D0 else d0)
        else:
            d0, ld0 = feats.dim, feats.ld
        self.din = [d0] + [2 * h for h in self.h[:-1]]
        self.ldin = [ld0] + [2 * h for h in self.h[:-1]]
        self.rows = [self.off[L - l] for l in range(L)]              # x rows of level l
        self.rall = [self.off[L - l + 1] for l in range(L)]          # all input rows of level l
        assert all(d % 8 == 0 for d in self.ldin)
        descs = []

        def copies(prm, need_t):
            r, c = prm.shape
            w = torch.zeros(r, _r64(c), dtype=T, device=dev)
            wt = torch.zeros(c, _r64(r), dtype=T, device=dev) if need_t else None
            descs.append(_PrepDesc(prm.data_ptr(), w.data_ptr(), wt.data_ptr() if need_t else None, r, c, w.shape[1],
                                   wt.shape[1] if need_t else 0, None, 0, int(self.code == nat.F32), 0))
            return w, wt
        self.w0, self.w0T, self.w2, self.w2T, self.wx, self.wxT, self.wn, self.wnT = ([] for _ in range(8))
        for l, layer in enumerate(self.layers):
            ing = l > 0 or self.emb                  # does this level's input need a gradient?
            a, b = copies(layer.att[0].weight, ing); self.w0.append(a); self.w0T.append(b)
            a, b = copies(layer.att[2].weight, True); self.w2.append(a); self.w2T.append(b)
            a, b = copies(layer.fc_x.weight, ing); self.wx.append(a); self.wxT.append(b)
            a, b = copies(layer.fc_neib.weight, True); self.wn.append(a); self.wnT.append(b)
        if self.emb:
            self._init_emb(copies)
        self.descs = torch.frombuffer(bytearray(bytes((_PrepDesc * len(descs))(*descs))), dtype=torch.uint8).to(dev)
        self.n_desc = len(descs)
        self.max_elems = max(d.rows * d.cols for d in descs)

        # level-0 rows of every hop.  Feature rows are read IN PLACE through the frontier's row list (K5 / K4 / K4' /
        # K5b all take one; GSAGE_ATTN_COPY_ROWS=1: gathered once per step into one buffer per batch in flight, as
        # the embedding prep needs anyway for its output rows); the gather launch then only carries the seeds' rows
        self.inplace0 = (not self.emb) and os.environ.get("GSAGE_ATTN_COPY_ROWS", "0") != "1"
        self._q_ids, self._cur_ids = None, self.ids_set[0]
        self.g0_set = [torch.zeros(self.B if self.inplace0 else self.rall[0], self.ldin[0], dtype=T, device=dev)
                       for _ in range(self.nset)]
        Ha, HL = self.Ha, self.HA_LD
        z = lambda *shape, dt=f32: torch.zeros(*shape, dtype=dt, device=dev)
        self.hid, self.a, self.aggc, self.ws, self.hout, self.dc = ([] for _ in range(6))
        self.dagg, self.dan, self.dax, self.da, self.dhid, self.datt, self.dx = ([] for _ in range(7))
        for l in range(L):
            R, RA, ld, h = self.rows[l], self.rall[l], self.ldin[l], self.h[l]
            last = l == L - 1
            self.hid.append(z(RA, HL, dt=T)); self.a.append(z(RA, Ha))
            self.aggc.append(z(R, ld, dt=T)); self.ws.append(z(RA - self.off[1]))   # (the aggregate: operand copy only)
            self.hout.append(z(R, 2 * h, dt=f32 if last else T)); self.dc.append(z(R, 2 * h, dt=T))
            self.dagg.append(z(R, ld)); self.dan.append(z(RA, Ha)); self.dax.append(z(RA, Ha))
            self.da.append(z(RA, HL, dt=T)); self.dhid.append(z(RA, HL, dt=T))
            ing = l > 0 or self.emb
            self.datt.append(z(RA, ld) if ing else None); self.dx.append(z(R, ld) if ing else None)
        # the LAST hop of a level (most of its rows) through K4 / K4' with the attention MLP inside
        # (csrc/gsage_attn_fused.hip: the rows are read once per direction); GSAGE_ATTN_FUSED=0: the separate launches
        on = os.environ.get("GSAGE_ATTN_FUSED", "1") != "0"
        self.fuse = [bool(on and nat.lib().gsage_attn_fused_ok(self.code, self.ldin[l], self.din[l], self.fan[L - l], Ha))
                     for l in range(L)]

    def _in(self, l, s):
        """input rows of level l (all hops it reads): (row block, leading dimension, row list or None) -- with a
        row list, input row i is block[list[i]] (level 0 read in place from the feature table)"""
        if l > 0:
            return self.hout[l - 1], self.ldin[l], None
        if self.inplace0:
            return self.store.data, self.ldin[0], self._cur_ids
        return self.g0_set[s], self.ldin[0], None

    def _wg_problems(self, l, s):
        """(dC, A, lda, M, Ntot, K, parameter) of the four weight gradients of level l"""
        inp, ld, rows = self._in(l, s)
        h, Ha, D, layer = self.h[l], self.Ha, self.din[l], self.layers[l]
        probs = [(self.dc[l][:, :h], inp, ld, self.rows[l], h, D, layer.fc_x.weight, rows),
                 (self.dc[l][:, h:], self.aggc[l], ld, self.rows[l], h, D, layer.fc_neib.weight, None),
                 (self.da[l], self.hid[l], self.HA_LD, self.rall[l], Ha, Ha, layer.att[2].weight, None),
                 (self.dhid[l], inp, ld, self.rall[l], Ha, D, layer.att[0].weight, rows)]
        if l == 0 and self.emb:      # the prep's affine: d out^T x embedding rows
            probs.append(self._emb_wgrad_problem())
        return probs

    def _init_reduce(self):
        dev, f32 = self.dev, torch.float32
        rdesc, self.slabs = [], []
        # K5b workgroups per problem: the problems go out eight to a launch in the order of _stage_compute and one
        # workgroup fits per CU, so each launch's problems are sized together to fill the chip once
        # (ops.wgrad_balance)
        order = [(l, i, p) for l in range(self.L - 1, -1, -1) for i, p in enumerate(self._wg_problems(l, 0))]
        targets = ops.wgrad_balance([(p[3], p[4], p[5]) for _l, _i, p in order])
        self.wg_target = {(l, i): t for (l, i, _p), t in zip(order, targets)}
        for l in range(self.L):
            bufs = []
            for i, (dC, A, lda, M, ntot, K, prm, _rows) in enumerate(self._wg_problems(l, 0)):
                rps, S, ldk = ops.wgrad_plan(M, ntot, K, self.wg_target[(l, i)])
                buf = torch.zeros(S, ntot, ldk, dtype=f32, device=dev)
                bufs.append(buf)
                rdesc.append(_ReduceDesc(buf.data_ptr(), ntot * ldk, self.poff[self.pidx[id(prm)]], S, ntot, K, ldk))
            self.slabs.append(bufs)
        if self.emb:
            rdesc.append(self._emb_reduce_desc())
        self._install_reduce(rdesc)
        if self.emb:
            self._init_emb_optimizer()

    # ---- stages ----------------------------------------------------------------------------------------
    def _stage_gather(self, s, with_adam=False, ids=None, hops=None, skip_rows=0):
        if self.emb:
            return               # nothing to gather ahead: the embedding rows are weights (read after Adam)
        st = self.store
        if ids is None:
            ids = self.ids_set[s]
        n0 = self.B if self.inplace0 else self.rall[0]        # (in place: a token segment carries Adam and K1)
        segs = [(st.data, ids[:n0], self.g0_set[s], n0, 1)]
        ops.gather_mean_multi(segs, st.ld, st.dim, st.ld, adam=self._adam_desc() if with_adam else None, hops=hops)

    def _input_grad0(self):
        """level 0's input gradient with an embedding prep: through att(.), through fc_x, ws * d agg of the parent
        -- no ReLU below (the prep's output is affine) -> din0f (fp32) + din0 (operand copy).  With features in front
        of the prep's output only columns [D0, D0 + E) of the three sources are merged (the features take no gradient)"""
        ld, E, RA0, L, o = self.ldin[0], self.E, self.rall[0], self.L, 4 * self.D0
        lp = self.din0 is not self.din0f          # fp32 for the bias gradient's column sums + the GEMMs' operand copy
        nat.check(nat.lib().gsage_attn_merge_bwd2(
            None, self.code, 0, self.datt[0].data_ptr() + o, ld, self.dx[0].data_ptr() + o, ld, self.rows[0],
            self.dagg[0].data_ptr() + o, ld, self.ws[0].data_ptr(), self.din0f.data_ptr(), nat.F32, E, RA0, E, L + 1,
            self.off_host, self.fan_host, self.din0.data_ptr() if lp else None, self.din0.stride(0) if lp else 0,
            ops._stream()), "attn_merge_bwd")

    def _gemm(self, A, lda, W, C, c_code, ldc, M, N, K, act, rows=None):
        ops._linear_launch(A, lda, rows.data_ptr() if rows is not None else None, 0, W.data_ptr(), W.shape[1], None, C,
                           ldc, M, N, K, act, 1, 0, 0, 0, self.code, c_code)

    @staticmethod
    def _child_rows(inp, rows, c0):
        """(table pointer, id pointer) of the rows from position c0 on, for K4 / K4'"""
        if rows is None:
            return inp[c0:].data_ptr(), None
        return inp.data_ptr(), rows[c0:].data_ptr()

    def _stage_compute(self, s):
        L, B, lib, stream, m = self.L, self.B, nat.lib(), ops._stream(), self.model
        Ha, HL, esz = self.Ha, self.HA_LD, self.esz
        self._cur_ids = self._q_ids if self._q_ids is not None else self.ids_set[s]
        if self.emb:
            self._prep_forward(s)
        for l in range(L):
            R, RA, h, D = self.rows[l], self.rall[l], self.h[l], self.din[l]
            inp, ld, rows = self._in(l, s)
            RM = R if self.fuse[l] else RA            # rows whose att(.) is a launch of its own (fused: all but the last hop)
            self._gemm(inp.data_ptr(), ld, self.w0[l], self.hid[l].data_ptr(), self.code, HL, RM, Ha, D, nat.ACT_TANH, rows)
            nat.check(lib.gsage_attn_mlp2_fwd(self.hid[l].data_ptr(), self.code, HL, self.w2[l].data_ptr(),
                                              self.w2[l].shape[1], self.a[l].data_ptr(), Ha, RM, Ha, stream), "attn_mlp2_fwd")
            for k in range(L - l):                   # K4 writes the aggregate as fp32 and as the next GEMMs' operand
                r0, c0 = self.off[k], self.off[k + 1]
                tab, idp = self._child_rows(inp, rows, c0)
                if l == 0 and k == L - 1:
                    self._time_next(4, 5)
                if self.fuse[l] and k == L - l - 1:  # the last hop: att(children), the weights and the sum in one pass
                    nat.check(lib.gsage_attn_fused_fwd(
                        tab, self.code, ld, idp, 0, self.w0[l].data_ptr(), self.w0[l].shape[1], self.w2[l].data_ptr(),
                        self.w2[l].shape[1], self.a[l][r0:].data_ptr(), Ha, self.size[k], self.fan[k + 1], D,
                        self.hid[l][c0:].data_ptr(), HL, self.a[l][c0:].data_ptr(), Ha,
                        self.ws[l][c0 - self.off[1]:].data_ptr(), self.aggc[l][r0:].data_ptr(), ld, stream), "attn_fused_fwd")
                    continue
                nat.check(lib.gsage_attn_aggregate_lp(
                    self.a[l][c0:].data_ptr(), Ha, self.a[l][r0:].data_ptr(), Ha, tab, self.code, ld,
                    idp, self.size[k], self.fan[k + 1], Ha, D, None, ld,
                    self.ws[l][c0 - self.off[1]:].data_ptr(), self.aggc[l][r0:].data_ptr(), ld, stream), "attn_aggregate")
            last = l == L - 1
            out, code = self.hout[l], (nat.F32 if last else self.code)
            act = nat.ACT_NONE if last else nat.ACT_RELU
            self._gemm(inp.data_ptr(), ld, self.wx[l], out.data_ptr(), code, 2 * h, R, h, D, act, rows)
            self._gemm(self.aggc[l].data_ptr(), ld, self.wn[l], out.data_ptr() + h * out.element_size(), code, 2 * h,
                       R, h, D, act)
        self._stage_head(s)
        if self.eval_only:
            return                                    # (forward only: train.evaluate's folds)
        self._backward_levels(s)

    def _backward_levels(self, s):
        L, lib, stream = self.L, nat.lib(), ops._stream()
        Ha, HL, esz = self.Ha, self.HA_LD, self.esz
        for l in range(L - 1, -1, -1):
            R, RA, h, D = self.rows[l], self.rall[l], self.h[l], self.din[l]
            inp, ld, rows = self._in(l, s)
            dc = self.dc[l]
            # d agg = dC[:, h:] Wn
            self._gemm(dc.data_ptr() + h * esz, 2 * h, self.wnT[l], self.dagg[l].data_ptr(), nat.F32, ld, R, D, h,
                       nat.ACT_NONE)
            for k in range(L - l - 1, -1, -1):       # (the last hop first: it leaves d a of its PARENTS, hop k's rows)
                r0, c0 = self.off[k], self.off[k + 1]
                tab, idp = self._child_rows(inp, rows, c0)
                if l == 0 and k == L - 1:
                    self._time_next(6, 7)
                if self.fuse[l] and k == L - l - 1:
                    nat.check(lib.gsage_attn_fused_bwd(
                        tab, self.code, ld, idp, 0, self.w2T[l].data_ptr(), self.w2T[l].shape[1],
                        self.dagg[l][r0:].data_ptr(), ld, self.ws[l][c0 - self.off[1]:].data_ptr(),
                        self.a[l][c0:].data_ptr(), Ha, self.a[l][r0:].data_ptr(), Ha, self.hid[l][c0:].data_ptr(), HL,
                        self.size[k], self.fan[k + 1], D, self.da[l][c0:].data_ptr(), HL, self.dhid[l][c0:].data_ptr(), HL,
                        self.dax[l][r0:].data_ptr(), Ha, stream), "attn_fused_bwd")
                    continue
                nat.check(lib.gsage_attn_bwd(
                    self.dagg[l][r0:].data_ptr(), ld, self.ws[l][c0 - self.off[1]:].data_ptr(),
                    self.a[l][c0:].data_ptr(), Ha, self.a[l][r0:].data_ptr(), Ha, tab, self.code, ld,
                    idp, self.size[k], self.fan[k + 1], Ha, D, self.dan[l][c0:].data_ptr(), Ha,
                    self.dax[l][r0:].data_ptr(), Ha, stream), "attn_bwd")
            # d a = (as a child) + (as a parent); hop 0 is never a child, the last hop never a parent (zeros);
            # d hid = (d a W2) * tanh' -- one pass
            nat.check(lib.gsage_attn_mlp2_bwd(self.dan[l].data_ptr(), Ha, self.dax[l].data_ptr(), Ha,
                                              self.hid[l].data_ptr(), self.code, HL, self.w2T[l].data_ptr(),
                                              self.w2T[l].shape[1], self.da[l].data_ptr(), HL, self.dhid[l].data_ptr(), HL,
                                              R if self.fuse[l] else RA, Ha, stream), "attn_mlp2_bwd")
            rows0 = l == 0 and self.emb and self.rows_ok      # level 0 over the prep: one row pipeline (common.py)
            if (l > 0 or self.emb) and not rows0:
                self._gemm(self.dhid[l].data_ptr(), HL, self.w0T[l], self.datt[l].data_ptr(), nat.F32, ld, RA, D, Ha,
                           nat.ACT_NONE)
            if l > 0 or self.emb:
                self._gemm(dc.data_ptr(), 2 * h, self.wxT[l], self.dx[l].data_ptr(), nat.F32, ld, R, D, h, nat.ACT_NONE)
            if rows0:
                o = 4 * self.D0               # (features in front of the prep's output take no gradient)
                self._prep_backward_rows(s, self.dhid[0].data_ptr(), self.w0T[0][self.D0:], None, 0,
                                         self.dx[0].data_ptr() + o, ld, self.rows[0], self.dagg[0].data_ptr() + o, ld,
                                         self.ws[0].data_ptr())
            elif l == 0 and self.emb:
                self._input_grad0()
                self._prep_backward(s)
            if l > 0:
                below = self.hout[l - 1]
                nat.check(lib.gsage_attn_merge_bwd(
                    below.data_ptr(), self.code, below.stride(0), self.datt[l].data_ptr(), ld, self.dx[l].data_ptr(), ld,
                    R, self.dagg[l].data_ptr(), ld, self.ws[l].data_ptr(), self.dc[l - 1].data_ptr(), self.code,
                    self.dc[l - 1].stride(0), RA, D, L - l + 1, self.off_host, self.fan_host, stream), "attn_merge_bwd")
        probs = []
        for l in range(L - 1, -1, -1):
            for i, ((dC, A, lda, M, ntot, K, prm, rows), slab) in enumerate(zip(self._wg_problems(l, s), self.slabs[l])):
                probs.append((dC, A, lda, 0, M, ntot, K, ntot, slab, self.wg_target[(l, i)], rows))
        for i in range(0, len(probs), 8):
            if i == 0:
                self._wgrad_ticks()
            ops.wgrad_multi(probs[i:i + 8])
        self._stage_finalize(s)
