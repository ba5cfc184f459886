"""The layers that follow the first Linear of the projection network (hfc_with_swav/swav_clustering.py:244-269).

    'linear'   nothing (every shipped config)
    '1-layer'  LeakyReLU(0.01)
    '2-layer'  BatchNorm1d -> LeakyReLU(0.01) -> Linear(D, D, bias=False) -> BatchNorm1d -> Tanh

The first Linear stays where the step has it (the tap pyramid, or the panel GEMM); this class runs the rest on the
sampled rows with the row kernels of csrc/gx_projhead.hip and the fp32 GEMM, forward and backward, and owns the
BatchNorm bookkeeping.  BatchNorm is in train mode everywhere — the reference never calls .eval(), so also
`predict_swav_codes` normalises with the statistics of the pixels it is given and moves the running averages.
A BatchNorm batch is the rows of ONE Sinkhorn problem: with sinkhorn_scope='image' the N sampled pixels of one view
of one image (the reference's call, :171-177), with 'global' the rows of every image and rank of that view and patch
(statistics and backward sums all-reduced), so that W ranks equal one.
"""
import torch
from torch import nn

from . import ops

KINDS = ('linear', '1-layer', '2-layer')
SLOPE = 0.01            # nn.LeakyReLU() default


def build_projection(kind, hlen, nclasses):
    """the torch module of swav_clustering.py:244-269 (what projection.pt pickles), same construction order"""
    if kind not in KINDS:
        raise ValueError(f"projn_nw={kind!r}: the reference knows {KINDS}")
    mods = [nn.Linear(hlen, nclasses, bias=False)]
    if kind == '1-layer':
        mods += [nn.LeakyReLU(inplace=True)]
    elif kind == '2-layer':
        mods += [nn.BatchNorm1d(nclasses), nn.LeakyReLU(inplace=True), nn.Linear(nclasses, nclasses, bias=False),
                 nn.BatchNorm1d(nclasses), nn.Tanh()]
    return nn.Sequential(*mods)


class ProjectionHead:
    def __init__(self, projection, kind):
        if kind not in KINDS:
            raise ValueError(f"projn_nw={kind!r}: the reference knows {KINDS}")
        self.kind = kind
        self.linear = kind == 'linear'
        self.projection = projection
        if kind == '2-layer':
            self.bn1, self.lin2, self.bn2 = projection[1], projection[3], projection[4]

    def extra_params(self):
        """the trainable tensors after the first Linear's weight, in nn.Module.parameters() order"""
        if self.kind != '2-layer':
            return []
        return [self.bn1.weight.data, self.bn1.bias.data, self.lin2.weight.data, self.bn2.weight.data, self.bn2.bias.data]

    @staticmethod
    def _track(bn, mean, var, n):
        """running statistics as nn.BatchNorm1d keeps them in train mode (momentum 0.1, unbiased variance)"""
        m = bn.momentum
        bn.running_mean.mul_(1.0 - m).add_(mean, alpha=m)
        bn.running_var.mul_(1.0 - m).add_(var, alpha=m * n / max(1, n - 1))
        bn.num_batches_tracked += 1

    def _bn(self, bn, x, act, groups, reduce, world, want_xhat):
        rows = x.shape[0]
        per = rows // groups
        a = torch.empty_like(x)
        xhat = torch.empty_like(x) if want_xhat else None
        vs = []
        for g in range(groups):
            sl = slice(g * per, (g + 1) * per)
            mean, var = ops.batch_stats(x[sl], reduce, world)
            self._track(bn, mean, var, per * world)
            ops.bn_act_fwd(x[sl], mean, var, bn.weight.data, bn.bias.data, bn.eps, act, SLOPE, want_xhat=want_xhat,
                           out=a[sl], xhat_out=xhat[sl] if want_xhat else None)
            vs.append(var)
        return a, xhat, vs

    def forward(self, z, groups=1, reduce=None, world=1, want_ctx=True):
        """z (rows, D) = first Linear's output rows (consumed) -> (head output rows, ctx for backward)"""
        if self.kind == 'linear':
            return z, None
        if self.kind == '1-layer':
            return ops.leaky_relu_(z, SLOPE), None
        a1, xh1, v1 = self._bn(self.bn1, z, ops.BN_LRELU, groups, reduce, world, want_ctx)
        h2 = ops.gemm(a1, self.lin2.weight.data, False, True)
        y, xh2, v2 = self._bn(self.bn2, h2, ops.BN_TANH, groups, reduce, world, want_ctx)
        return y, ((xh1, v1, a1, xh2, v2, groups, reduce, world) if want_ctx else None)

    def _bn_bwd(self, bn, da, xhat, vs, act, groups, reduce, world, gw, gb, beta):
        rows = da.shape[0]
        per = rows // groups
        dx = torch.empty_like(da)
        for g in range(groups):
            sl = slice(g * per, (g + 1) * per)
            _, dg, db = ops.bn_act_bwd(da[sl], xhat[sl], vs[g], bn.weight.data, bn.bias.data, bn.eps, act, SLOPE,
                                       reduce=reduce, world=world, out=dx[sl])
            if g == 0 and beta == 0.0:          # overwrite: the gradient buffers start uninitialised
                gw.copy_(dg)
                gb.copy_(db)
            else:
                gw.add_(dg)
                gb.add_(db)
        return dx

    def backward(self, dy, y, ctx, grads, beta, split_k=1):
        """dy = d loss / d (head output rows) -> d loss / d (first Linear's output rows); the gradients of
        extra_params() go into `grads` (same order), scaled in by `beta` (0 = overwrite) like the step's other GEMMs.
        `y` is the head output or any positive multiple of it (the step's normalised rows)."""
        if self.kind == 'linear':
            return dy
        if self.kind == '1-layer':
            return ops.leaky_relu_bwd_(dy, y, SLOPE)
        xh1, v1, a1, xh2, v2, groups, reduce, world = ctx
        g_bn1w, g_bn1b, g_w2, g_bn2w, g_bn2b = grads
        dh2 = self._bn_bwd(self.bn2, dy, xh2, v2, ops.BN_TANH, groups, reduce, world, g_bn2w, g_bn2b, beta)
        ops.gemm(dh2, a1, True, False, out=g_w2, beta=beta, split_k=split_k)
        da1 = ops.gemm(dh2, self.lin2.weight.data, False, False)
        return self._bn_bwd(self.bn1, da1, xh1, v1, ops.BN_LRELU, groups, reduce, world, g_bn1w, g_bn1b, beta)
