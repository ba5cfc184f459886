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

    def project(self, wp, score_head=None):
        """wp (D, hlen) -> Z (B, H, W, D): the projection of every pixel's descriptor.
        score_head=(Wc, bias) with K <= 16: the last upsample-add also emits the prototype score
        table of every pixel; returns (Z, (rn_table, score_table))."""
        d = wp.shape[0]
        prev = None
        tables = None
        for r, taps in zip(self.res, self.levels):
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
                ops.gemm_nt_planes(ta, tb, ops.split_gemm_weights(wp[:, off:off + kk]), p, coarse=up, fine_res=r if up is not None else 0)
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
            if prev is not None:
                if score_head is not None and r == self.res[-1]:
                    tables = ops.upsample2_add_scores_(prev if up is None else None, p, *score_head)
                elif up is None:
                    ops.upsample2_add_(prev, p)
            prev = p
        if score_head is not None:
            if tables is None:                       # single-level pyramid: no upsample pass to ride on
                tables = ops.proto_fwd_lowrank(prev, None, None, *score_head)
            return prev, tables
        return prev

    def backward(self, g, gwp, accumulate, g_pooled=None, on_level=None):
        """g (B, H, W, D) = d loss / d Z -> gwp (D, hlen) (+)= sum over taps of G_level^T tap.
        g_pooled: sumpool2(g) if the caller already has it.  on_level([(off, width), ...]) is called once
        the GEMMs of a level are queued (its columns of gwp are final: a sharded caller starts their
        all-reduce there)."""
        first = not accumulate
        touched = set()
        for r, taps in zip(reversed(self.res), reversed(self.levels)):
            rows = self.B * r * r
            g2 = g.reshape(rows, -1)
            used = [(t, off, cu) for t, off, cu in taps if cu > 0]
            if (len(used) == 2 and used[0][2] % 128 == 0 and used[1][1] == used[0][1] + used[0][2]):
                # both taps of the level against ONE pass over the level's dZ
                (ta, off, ca), (tb, _, cb) = used
                tiles = ((g2.shape[1] + 127) // 128) * ((ca + cb + 127) // 128)
                split = int(max(1, min(512 // tiles, rows // 256)))
                ops.gemm_tn_cat(g2, ta[:, :ca], tb[:, :cb], gwp[:, off:off + ca + cb], beta=0.0 if first else 1.0, split_k=split)
                touched.update(((off, ca), (off + ca, cb)))
                used = []
            for t, off, cu in used:
                tiles = ((g2.shape[1] + 127) // 128) * ((cu + 127) // 128)
                split = int(max(1, min(512 // tiles, rows // 256)))
                ops.gemm(g2, t[:, :cu], True, False, out=gwp[:, off:off + cu], beta=0.0 if first else 1.0,
                         split_k=split)
                touched.add((off, cu))
            if on_level is not None and any(cu > 0 for _, _, cu in taps):
                on_level([(off, cu) for _, off, cu in taps if cu > 0])
            if r != self.res[0]:
                g = g_pooled if (g_pooled is not None and r == self.res[-1]) else ops.sumpool2(g)
        if first:      # columns of hlen not covered by any tap cannot exist (sum C >= hlen); nothing to zero
            pass
        return gwp
