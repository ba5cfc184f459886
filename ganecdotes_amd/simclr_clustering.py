"""SimCLR baseline preprocessor on MI355X (`--method=hfc_with_simclr`, BASELINE.json config 5).

Mirrors the reference class `SimCLRClustering` (baseline/hfc_with_simclr/simclr_clustering.py:
14-400): same constructor keywords, `preprocess` / `pretrain` / `predict_simclr_codes`, the
`projection` torch module (Linear -> BatchNorm1d -> LeakyReLU -> Linear) and `projection.pt`.
Views, descriptor gather and the Linear layers reuse the SwAV path's kernels; the head (batch-norm
in train mode, the reference's inline loss over the first 2*batch feature rows with its wrap-around
pairs, and their backward) is `csrc/gx_simclr.hip`.  Reference quirks are kept (SURVEY.md
appendix 11); `predict_simclr_codes` uses batch statistics because the reference never leaves
train mode.
"""
import os
import time

import numpy as np
import torch
from torch import nn

from . import ops
from .draws import ImageDraws, ViewDraws, draw_image
from .hostcfg import limit_host_threads
from .pyramid import Pyramid
from .swav_clustering import _NullLogger
from .views import ViewMixin, check_hf_interp


def draw_simclr_iteration(style_dim, gen_num_layers, n_perturb_layers, hw, transforms=True, degrees=10.0,
                          layer_no=None, cpu_reference_rng=False):
    """Host draws of one SimCLR iteration in the reference's order: z; view s (+ its angle/flip);
    view t (+ its angle/flip); one randperm (simclr_clustering.py:177-207)."""
    z = torch.randn(1, style_dim)
    views = []
    for _ in range(2):
        if cpu_reference_rng:
            for i in range(gen_num_layers):
                r = 2 ** ((i + 5) // 2)
                torch.empty(1, 1, r, r).normal_()
        lno = layer_no if layer_no is not None else int(np.random.choice(list(range(n_perturb_layers))))
        if cpu_reference_rng:
            maps = [torch.randn(1, style_dim) for _ in range(2 * n_perturb_layers)]
            vd = ViewDraws(lno, torch.cat([maps[2 * lno], maps[2 * lno + 1]], 0))
        else:
            vd = ViewDraws(lno, torch.randn(2, style_dim))
        if transforms:
            vd.angle = float(torch.empty(1).uniform_(-degrees, degrees).item())
            vd.flip = bool(torch.rand(1) < 0.5)
        views.append(vd)
    return ImageDraws(z, views, [torch.randperm(hw)])


class SimCLRClustering(ViewMixin):

    def __init__(self, model, model_config, perturb_args, simclr_args, logger=None, train=True, out_dir=None,
                 device='cuda', tb=None, layer_hf_dim=None):
        self.model = model
        self.model_config = model_config
        self.perturb_args = perturb_args
        self.simclr_args = simclr_args
        self._interp = check_hf_interp(simclr_args, 'simclr_args', allowed=('nearest', 'bilinear'))
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError("SimCLRClustering (ganecdotes_amd) needs a HIP device: there is no CPU fallback")
        ops.lib()
        limit_host_threads()
        self.writer = tb
        self.nclasses = simclr_args['nclasses']
        self._hlen = simclr_args['hlen']
        self.logger = logger or _NullLogger()
        self.train = train
        self.out_dir = out_dir
        self.projection = None
        self.projection_file = None
        if out_dir is not None:
            self.swav_dir = os.path.join(out_dir, 'simclr')
            os.makedirs(self.swav_dir, exist_ok=True)
            self.projection_file = os.path.join(out_dir, 'projection.pt')
            if not self.train:
                if os.path.exists(self.projection_file):
                    self.projection = torch.load(self.projection_file, weights_only=False).to(self.device)
                else:
                    self.logger.info("Projection File not found - pretraining ...")
        with torch.no_grad():
            self.mean_latent = self.model.mean_latent(self.model_config.num_latents_for_mean)
            self.truncation = self.model_config.truncation
        self.feature_peak = self.probe_feature_range()          # a checkpoint outside the f16x3 window -> bf16x6, with a warning
        self.layer_hf_dim = layer_hf_dim
        self._bufs = None
        self._step_no = 0

    def _pick_spec(self):
        return self.simclr_args['batch_size'], 1, 'host'

    def preprocess(self, input_latent):
        if self.train or self.projection_file is None or not os.path.exists(self.projection_file):
            self.pretrain(input_latent)

    def build_heads(self):
        """Linear(hlen, D, no bias) -> BatchNorm1d(D) -> LeakyReLU -> Linear(D, D, no bias)
        (simclr_clustering.py:147-158)."""
        d = self.nclasses
        self.projection = nn.Sequential(nn.Linear(self._hlen, d, bias=False), nn.BatchNorm1d(d),
                                        nn.LeakyReLU(inplace=True), nn.Linear(d, d, bias=False)).to(self.device)
        for p in self.projection.parameters():
            p.requires_grad_(False)
        self._bufs = None
        self._step_no = 0

    def _params(self):
        pr = self.projection
        return [pr[0].weight.data, pr[1].weight.data, pr[1].bias.data, pr[3].weight.data]

    # ------------------------------------------------------------------------------------- one iteration
    def train_step(self, image):
        """One optimizer step on one ImageDraws (2 views, batch_size picked pixel pairs).  Returns the
        loss as a 1-element HIP tensor."""
        sa = self.simclr_args
        bsz, hlen = sa['batch_size'], self._hlen
        w1, gamma, beta, w2 = self._params()
        bn = self.projection[1]
        pb = self.prepare_batch([image])
        views = self.make_views(pb, want_image=False)                          # acquires pb on this stream
        pb_picks = pb.picks[0]                                                 # (1, bsz) int32
        xs = []
        for ts, maps, _ in views:
            x = ops.gather_taps(ts, pb_picks, hlen, maps, interp=self._interp)  # (bsz, hlen)
            ops.row_l2_normalize_(x, want_rnorm=False)                         # F.normalize(dim=1) per pixel
            xs.append(x)
        x = torch.stack(xs, 1).reshape(2 * bsz, hlen).contiguous()             # s0, t0, s1, t1, ...
        n = 2 * bsz
        h1 = ops.gemm(x, w1, False, True)                                      # (n, D)
        mean, var = ops.col_stats(h1)
        a, xhat = ops.bn_lrelu_fwd(h1, mean, var, gamma, beta, bn.eps, 0.01)
        out = ops.gemm(a, w2, False, True)                                     # (n, D)
        loss, dout = ops.simclr_ntxent(out, sa['temperature'])
        # backward
        gw2 = ops.gemm(dout, a, True, False)                                   # (D, D)
        da = ops.gemm(dout, w2, False, False)                                  # (n, D)
        dh1, dgamma, dbeta = ops.bn_lrelu_bwd(da, xhat, var, gamma, beta, bn.eps, 0.01)
        gw1 = ops.gemm(dh1, x, True, False)                                    # (D, hlen)
        # BatchNorm running statistics (module state; momentum 0.1, unbiased variance)
        bn.running_mean.mul_(1 - bn.momentum).add_(mean, alpha=bn.momentum)
        bn.running_var.mul_(1 - bn.momentum).add_(var, alpha=bn.momentum * n / (n - 1))
        bn.num_batches_tracked += 1
        grads = [gw1, dgamma, dbeta, gw2]
        if self._bufs is None:
            self._bufs = [torch.zeros_like(p) for p in (w1, gamma, beta, w2)]
        ta = sa['train_args']
        for p, g, buf in zip((w1, gamma, beta, w2), grads, self._bufs):
            ops.larc_sgd_(p, g.contiguous(), buf, ta['lr'], ta.get('momentum', 0.0), sa['trust_coeff'],
                          first_step=(self._step_no == 0), use_larc=sa.get('use_larc', True))
        self._step_no += 1
        self._grads = grads
        return loss

    def draw_iteration(self):
        sa = self.simclr_args
        hw = self.model.size * self.model.size
        return draw_simclr_iteration(self.model.style_dim, self.model.num_layers, self.perturb_args['n_layers'], hw,
                                     transforms=sa.get('transforms', True), layer_no=self.perturb_args['layer_no'],
                                     cpu_reference_rng=sa.get('cpu_reference_rng', False))

    def pretrain(self, input_latent, num_test_samples=2):
        """(simclr_clustering.py:131-281) num_iters optimizer steps, then projection.pt."""
        sa = self.simclr_args
        self.build_heads()
        self.logger.info("Projection Network:")
        self.logger.info(self.projection.__str__())
        self.logger.info("SimCLR Parameters:")
        for k, v in sa.items():
            self.logger.info(f"{k}: {v}")
        t0 = time.time()
        self.loss_history = []
        for e in range(sa['num_iters']):
            loss = self.train_step(self.draw_iteration())
            self.loss_history.append(loss)
            if self.writer is not None:
                self.writer.add_scalar('simclr/loss', loss.item(), e)
            lv = float(loss.item())
            ops.check_finite(lv, "SimCLR loss")
            self.logger.info(f" (Iter:{e}):\tLoss: {lv:.03f},\tTime: {time.time() - t0:.03f}\t")
        if self.projection_file is not None:
            torch.save(self.projection, self.projection_file)

    # ------------------------------------------------------------------------------------- inference
    def predict_simclr_codes(self, input_latent):
        """(simclr_clustering.py:365-400) -> (out_preds (b,D,h,w), out_labels (b,h,w)).  The first
        Linear of every normalised pixel descriptor is the pyramid projection divided by the
        descriptor norm; BatchNorm uses the image's own pixel statistics (train mode) and, like the
        reference, updates the running statistics."""
        lat = input_latent.to(self.device).float()
        feats = self._features_only(lat)
        ts = ops.TapSet([f.permute(0, 2, 3, 1) for f in feats])
        hw = ts.size
        w1, gamma, beta, w2 = self._params()
        bn = self.projection[1]
        if self._interp == 'nearest':
            z = Pyramid(ts, self._hlen).project(w1).reshape(ts.B * hw * hw, -1)
            ops.row_div_(z, ops.norm_map(ts, self._hlen).reshape(-1))
        else:       # hf_interp='bilinear': the descriptor panel itself (the pyramid factorisation is the nearest upsampling)
            picks = torch.arange(hw * hw, dtype=torch.int32, device=self.device).repeat(ts.B, 1)
            x = ops.gather_taps(ts, picks, self._hlen, interp=self._interp)
            ops.row_l2_normalize_(x, want_rnorm=False)
            z = ops.gemm(x, w1, False, True)
        mean, var = ops.col_stats(z)
        a, _ = ops.bn_lrelu_fwd(z, mean, var, gamma, beta, bn.eps, 0.01, want_xhat=False)
        n = z.shape[0]
        bn.running_mean.mul_(1 - bn.momentum).add_(mean, alpha=bn.momentum)
        bn.running_var.mul_(1 - bn.momentum).add_(var, alpha=bn.momentum * n / (n - 1))
        bn.num_batches_tracked += 1
        out = ops.gemm(a, w2, False, True)
        labels = ops.row_argmax(out).reshape(ts.B, hw, hw)
        preds = out.reshape(ts.B, hw, hw, -1).permute(0, 3, 1, 2)
        return preds, labels
