"""Pyramid projection: S1 (upsample+concat), S2 (pixel sampling) and the projection half of S3
factorised through the nearest-neighbour upsampling.

The reference projects the 5376-channel descriptor of every SAMPLED pixel (20 000 x 5 patches x
2 views x 110 GFLOP).  The descriptor is a concatenation of nearest-upsampled taps and the
projection is linear, so   z(p) = sum_l  Wp[:, cols_l] . tap_l[src_l(p)]   — every tap can be
projected once at its NATIVE resolution (31 GFLOP per view) and the levels summed up the
pyramid into one full-resolution table Z; z(p) is then a single 2 KB row of Z.  The backward is
the exact adjoint (scatter-add, 2x2 sum-pooling, one TN GEMM per tap).  Same real-number result
as swav_clustering.py:108-130 + :160-171 (and autograd of them); only the fp32 summation order
differs (13 partial dot products instead of one 5376-long chain).

With K <= 16 prototypes (`project_tables` / `backward_lowrank`) the loss only sees a pixel's K
prototype scores and 1 / |z|, and the top of the pyramid is never formed:
  * levels without taps (hlen cuts them off: a 512^2 generator at hlen = 5376) are skipped — the
    tables live at the highest level that has taps and the pick lists index it with a shift;
  * a finest level with taps enters through CF-wide forms of its taps (csrc/gx_fineq.hip: one
    (CF + 16)-column contraction whose epilogue reduces every row to K + 1 numbers), so its
    (pixels x D) table Z — 4.3 GB at 16 images of 256^2 — and the gradient table of the same size
    are never written; the backward is a weighted Gram GEMM of the taps plus GEMMs on the level below.
"""
import torch

from . import ops


class Pyramid:
    def __init__(self, tapset, hlen):
        """tapset: ops.TapSet of NHWC taps (B, r, r, C) in concat order; hlen: channels kept."""
        self.B = tapset.B
        self.size = tapset.size
        levels = {}
        off = 0
        for t in tapset.taps:
            r, c = t.shape[1], t.shape[3]
            cu = max(0, min(c, hlen - off))
            levels.setdefault(r, []).append((t.reshape(-1, c), off, cu))
            off += c
        self.res = sorted(levels)
        self.levels = [levels[r] for r in self.res]
        if any(b != 2 * a for a, b in zip(self.res, self.res[1:])) or self.res[-1] != self.size:
            raise RuntimeError("pyramid projection needs taps at successive powers of two")
        self._lr = None

    # ------------------------------------------------------------------------------------- forward
    def _project_levels(self, wp, nlev):
        """Z of level nlev - 1 (B, r, r, D): the projections of the first nlev levels summed up the pyramid"""
        d = wp.shape[0]
        prev = None
        for r, taps in zip(self.res[:nlev], self.levels[:nlev]):
            p = torch.empty(self.B * r * r, d, dtype=torch.float32, device=wp.device)
            used = [(t, off, cu) for t, off, cu in taps if cu > 0]
            # the level below rides on the epilogue of this level's (first) GEMM: p = taps . Wp^T + upsample2(prev)
            up = prev.reshape(-1, d) if (prev is not None and used and (r & (r - 1)) == 0) else None
            if not used:
                p.zero_()
            elif (len(used) <= 2 and used[0][2] == used[0][0].shape[1] and
                  (len(used) == 1 or used[1][1] == used[0][1] + used[0][2]) and
                  ops.gemm_planes_ok(used[0][0], used[1][0][:, :used[1][2]] if len(used) == 2 else None, d, p)):
                # the level's weight columns as bf16x6 planes, split once per call (Wp changes once per step):
                # both operands of the projection then reach LDS by DMA (gemm_xw_kernel)
                ta, off, ca = used[0]
                tb = used[1][0][:, :used[1][2]] if len(used) == 2 else None
                kk = ca + (used[1][2] if len(used) == 2 else 0)
                # (f16x3 form: the taps are generator activations, whose range is the x-planes convs' contract)
                ops.gemm_nt_planes(ta, tb, ops.split_gemm_weights(wp[:, off:off + kk], h2=True), p, coarse=up,
                                   fine_res=r if up is not None else 0, a_range=ops.FEATURE_RANGE)
            elif (len(used) == 2 and used[0][2] == used[0][0].shape[1] and used[0][2] % 32 == 0 and
                  used[1][1] == used[0][1] + used[0][2]):
                (ta, off, ca), (tb, _, cb) = used
                if up is not None:
                    ops.gemm_upadd(ta, tb[:, :cb], wp[:, off:off + ca + cb], p, up, r)
                else:
                    ops.gemm_cat(ta, tb[:, :cb], wp[:, off:off + ca + cb], p)
            else:
                for i, (t, off, cu) in enumerate(used):
                    if up is not None and i == 0:
                        ops.gemm_upadd(t[:, :cu], None, wp[:, off:off + cu], p, up, r)
                    else:
                        ops.gemm(t[:, :cu], wp[:, off:off + cu], False, True, out=p, beta=0.0 if i == 0 else 1.0)
            p = p.reshape(self.B, r, r, d)
            if prev is not None and up is None:
                ops.upsample2_add_(prev, p)
            prev = p
        return prev

    def project(self, wp):
        """wp (D, hlen) -> Z (B, H, W, D): the projection of every pixel's descriptor."""
        return self._project_levels(wp, len(self.res))

    def _used(self, li):
        return [(t, off, cu) for t, off, cu in self.levels[li] if cu > 0]

    def project_tables(self, wp, wc, bias, quad=True):
        """The K <= 16 path: (rn_table, sc_table, shift) — 1 / |z| and the K prototype scores of every pixel of the
        table level, which lies `shift` levels below the finest (levels without taps are never formed).
        quad=False keeps the finest level's projected table (the form every engine takes; A/B and parity tests)."""
        k = wc.shape[0]
        top = len(self.res) - 1
        while top > 0 and not self._used(top):
            top -= 1
        shift = len(self.res) - 1 - top
        used = self._used(top)
        r = self.res[top]
        fq = None
        if quad and top > 0 and r >= 32 and 1 <= len(used) <= 2 and all(cu == t.shape[1] for t, _, cu in used) and \
                (len(used) == 1 or used[1][1] == used[0][1] + used[0][2]) and (r & (r - 1)) == 0 and \
                ops.fineq_ok(used[0][2], used[1][2] if len(used) == 2 else 0, k) and wp.shape[0] % 16 == 0 and \
                all(t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0 for t, _, _ in used):
            fq = used
        if fq is None:
            z = self._project_levels(wp, top + 1)
            if top > 0:
                rn, sc = ops.upsample2_add_scores_(None, z, wc, bias)
            else:                                    # single-level pyramid
                rn, sc = ops.proto_fwd_lowrank(z, None, None, wc, bias)
            self._lr = dict(mode="table", top=top, shift=shift, z=z)
            return rn, sc, shift
        # quadratic-form finest level (csrc/gx_fineq.hip): only the level below is projected
        zc = self._project_levels(wp, top)
        d = wp.shape[0]
        t1, off, c1 = fq[0]
        t2 = fq[1][0] if len(fq) == 2 else None
        cf = c1 + (fq[1][2] if len(fq) == 2 else 0)
        wf = wp[:, off:off + cf]                                        # (D, CF) view, row stride hlen
        mg = ops.gemm(wf, wf, True, False)                              # Wf^T Wf (CF, CF)
        wcf = ops.gemm(wc, wf, False, False)                            # Wc Wf (K, CF)
        bq = ops.fineq_pack_weights(mg, wcf)
        zc2 = zc.reshape(-1, d)
        yt = ops.fineq_ytable(zc2.shape[0], wp.device)                  # per parent pixel: 2 Zc Wf | Zc Wc^T | |Zc|^2
        ops.proto_raw_tables(zc2, wc, ytable=yt)
        # a bound of |Zc| for the f16x3 forms of the two GEMMs that contract it (here and Zc^T Tt in the backward): the largest
        # row norm, from the |Zc|^2 column the pass above just wrote — 1 MB read instead of a pass over the table
        zc_amax = ops.absmax(yt[:, ops.FINEQ_NC_COL], sqrt=True) if ops.GEMM_H2 else None
        ops.gemm_nt_planes(zc2, None, ops.split_gemm_weights(wf.t().contiguous(), h2=True), yt[:, :cf], alpha=2.0, a_amax=zc_amax)
        rn, sc, t_amax = ops.fineq_fwd(t1, t2, ops.split_gemm_weights(bq, h2=True), yt, bias, k, r, a_range=ops.FEATURE_RANGE,
                                       want_amax=True)
        self._lr = dict(mode="quad", top=top, shift=shift, z=zc, t1=t1, t2=t2, off=off, cf=cf, z_amax=zc_amax,
                        t_amax=t_amax if ops.GEMM_H2 else None)
        return rn, sc, shift

    def table_pixels(self):
        """pixels per image of the table level of the last project_tables call"""
        r = self.res[self._lr["top"]]
        return r * r

    # ------------------------------------------------------------------------------------- backward
    def backward(self, g, gwp, accumulate, g_pooled=None, on_level=None, nlev=None, g_amax=None, g_mul=1.0):
        """g (B, r, r, D) = d loss / d Z of level nlev - 1 (default: the finest) -> gwp (D, hlen) (+)= sum over the
        taps of the first nlev levels of G_level^T tap.  g_pooled: sumpool2(g) if the caller already has it.
        on_level([(off, width), ...]) is called once the GEMMs of a level are queued (its columns of gwp are final:
        a sharded caller starts their all-reduce there).  g_amax (one-element HIP tensor, optional): max |g| as its producer left
        it — the level GEMMs then run the f16x3 form (the taps' range is the convs' contract; a 2x2 sum-pooling multiplies the
        bound by 4); g_mul: the factor between *g_amax and the bound of |g| (a scatter leaves its largest single contribution)."""
        first = not accumulate
        gmul = float(g_mul)
        nlev = len(self.res) if nlev is None else nlev
        res, levels = self.res[:nlev], self.levels[:nlev]
        for r, taps in zip(reversed(res), reversed(levels)):
            rows = self.B * r * r
            g2 = g.reshape(rows, -1)
            used = [(t, off, cu) for t, off, cu in taps if cu > 0]
            if (len(used) == 2 and used[0][2] % 128 == 0 and used[1][1] == used[0][1] + used[0][2]):
                # both taps of the level against ONE pass over the level's dZ
                (ta, off, ca), (tb, _, cb) = used
                tiles = ((g2.shape[1] + 127) // 128) * ((ca + cb + 127) // 128)
                split = int(max(1, min(512 // tiles, rows // 256)))
                ops.gemm_tn_cat(g2, ta[:, :ca], tb[:, :cb], gwp[:, off:off + ca + cb], beta=0.0 if first else 1.0, split_k=split,
                                a_amax=g_amax, a_mul=gmul, b_range=ops.FEATURE_RANGE if g_amax is not None else None)
                used = []
            for t, off, cu in used:
                tiles = ((g2.shape[1] + 127) // 128) * ((cu + 127) // 128)
                split = int(max(1, min(512 // tiles, rows // 256)))
                ops.gemm(g2, t[:, :cu], True, False, out=gwp[:, off:off + cu], beta=0.0 if first else 1.0,
                         split_k=split, a_amax=g_amax, a_range=gmul if g_amax is not None else None,
                         b_range=ops.FEATURE_RANGE if g_amax is not None else None)
            if on_level is not None and any(cu > 0 for _, _, cu in taps):
                on_level([(off, cu) for _, off, cu in taps if cu > 0])
            if r != res[0]:
                g = g_pooled if (g_pooled is not None and r == res[-1]) else ops.sumpool2(g)
                gmul *= 4.0
        return gwp

    def backward_lowrank(self, coeffs, wp, wc, gwp, gwc, on_level=None, on_heads=None, fused_finish=True):
        """The adjoint of project_tables from the coefficient grid C (table pixels x (K + 1)) of the low-rank path
        (csrc/gx_swav.hip: G[q] = C[q][:K] Wc + C[q][K] z(q)): gwp (D, hlen) = d loss / d Wp, gwc (K, D) = d loss / d Wc
        (both overwritten).  on_heads() is called when gwc is final, on_level as in `backward`."""
        lr = self._lr
        top, zc = lr["top"], lr["z"]
        if top + 1 < len(self.res):                  # columns of levels above the table level (hlen cut them off)
            pass                                     # carry no weight columns: nothing to zero
        if lr["mode"] == "table":
            pool = top > 0
            out = ops.proto_lowrank_finish(coeffs, zc, wc, gwc, 0.0, want_pool=pool)
            g, gp = out if pool else (out, None)
            if on_heads is not None:
                on_heads()
            return self.backward(g, gwp, accumulate=False, g_pooled=gp, on_level=on_level, nlev=top + 1)
        t1, t2, off, cf = lr["t1"], lr["t2"], lr["off"], lr["cf"]
        r = self.res[top]
        k, d = wc.shape[0], wp.shape[0]
        wf = wp[:, off:off + cf]
        zc2 = zc.reshape(-1, d)
        # ONE pass over the level's taps: S = sum_q e[q] t t^T (e = C[:, K]), P = C[:, :K]^T t, Tt = 2x2-pooled e t, Cpool
        # (f16x3: the taps' true maximum as fineq_fwd met it + the coefficient grid's two maxima — a product of tight bounds)
        s, pm, tt, cpool, tt_amax = ops.fineq_bwd(coeffs, t1, t2, k, r, want_amax=True, t_amax=lr.get("t_amax"))
        if not ops.GEMM_H2:
            tt_amax = None
        g_amax = None
        # gradient of the level below: Gc = Tt Wf^T + Cpool[:, :K] Wc + epool .* Zc  (+ its 2x2 sum-pooling), gWc
        pool = top > 1
        rc = zc.shape[1]
        if fused_finish and ops.get_gemm_mode() == 1 and ops.gemm_finish_ok(d, cf, rc, k):
            # ... in ONE kernel: the finish rides on the GEMM's epilogue, G is written once (csrc/gx_gemm.hip)
            # (f16x3: Tt's bound is the maximum fineq_bwd left on the device)
            gcz, gp, g_amax = ops.gemm_nt_planes_finish(tt, ops.split_gemm_weights(wf, h2=True), cpool, zc, wc, gwc, 0.0, rc, pool,
                                                        want_amax=True, a_amax=tt_amax)
        else:
            gc = torch.empty_like(zc2)
            ops.gemm_nt_planes(tt, None, ops.split_gemm_weights(wf), gc)
            out = ops.proto_lowrank_finish(cpool, zc, wc, gwc, 0.0, want_pool=pool, add_to=gc.reshape(zc.shape))
            gcz, gp = out if pool else (out, None)
        ops.gemm(pm, wf, False, True, out=gwc, beta=1.0)                # gWc += P Wf^T
        if on_heads is not None:
            on_heads()
        # dWf = Zc^T Tt + Wf S + Wc^T P
        gcols = gwp[:, off:off + cf]
        rc_rows = zc2.shape[0]
        tiles = ((d + 127) // 128) * ((cf + 127) // 128)
        ops.gemm(zc2, tt, True, False, out=gcols, beta=0.0, split_k=int(max(1, min(512 // tiles, rc_rows // 256))),
                 a_amax=lr.get("z_amax") if tt_amax is not None else None, b_amax=tt_amax)
        ops.gemm(wf, s, False, False, out=gcols, beta=1.0)
        ops.gemm(wc, pm, True, False, out=gcols, beta=1.0)
        if on_level is not None:
            on_level([(o, cu) for _, o, cu in self._used(top)])
        return self.backward(gcz, gwp, accumulate=False, g_pooled=gp, on_level=on_level, nlev=top, g_amax=g_amax)
