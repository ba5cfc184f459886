"""Per-pixel SwAV clustering on MI355X — the `preprocessor` plugin of `--method=hfc_with_swav`.

Mirrors the reference class `SwAVClustering` (hfc_with_swav/swav_clustering.py:28-693): same
constructor keywords, `preprocess` / `pretrain` / `predict_swav_codes`, the `projection` /
`prototype` torch modules and the `projection.pt` / `prototypes.pt` files.  The arithmetic is
not the reference's op chain but hand-written gfx950 kernels reached through the C ABI
(`ganecdotes_amd.ops`):

  views      : both views of the batch as ONE 2B-image NHWC generator forward (the reference's discarded
               first forward is replaced by the 3-line latent computation it was run for)
  S1/S1b/S2/S3 (default, projection_mode='pyramid'): every tap is projected ONCE at its native resolution
               and the levels are summed up the pyramid into the table Z (2B, H, W, D) (pyramid.py); the
               1.3 GiB stack and the N x 5376 panel never exist — rotation / flip / sampling are an index
               into Z.  K <= 16: the finest level also emits every pixel's prototype scores, a patch is a
               gather of (K+1)-float rows and the backward a scatter of K+1 coefficients (low-rank path).
               projection_mode='panel' keeps the reference-shaped gather + N x 5376 GEMM.
  S4         : log-domain scaling-vector Sinkhorn, all views x patches of a step as ONE batched problem
               (one read of the scores per iteration, Q never formed)
  S5/S6      : fused codes + swapped-CE + d(scores); dense finish / MFMA GEMMs for dWc, dz; the pyramid's
               adjoint (sum-pooling + one TN GEMM per tap) for dWp
  S7/S8      : row normalise, LARC-SGD

Extensions over the reference (opt-in `swav_args` keys, defaults = reference behaviour):
  projection_mode 'pyramid' (default: project every tap once at native resolution, see
                  pyramid.py) or 'panel' (gather the sampled descriptors and run the reference's
                  N x 5376 GEMM per view-patch)
  batch_size      images per optimizer step on this rank (reference: 1)
  sinkhorn_scope  'image' (per-image Sinkhorn, == reference steps with gradient averaging) or
                  'global' (Sinkhorn over all pixels of the global batch, SURVEY.md par. 8e)
  pick_source     'host' (torch.randperm on the host generator, the reference's stream) or
                  'device' (same distribution, drawn on the device generator: no host work)
  use_larc, transforms, cpu_reference_rng   switches used by parity tests (the last one makes
                  the host RNG stream that of a CPU run of the reference, see draws.py)
Multi-GPU: pass `dist_group` (torch.distributed, backend nccl = RCCL); images shard over
ranks, the Sinkhorn column sums / maxima and the parameter gradients are all-reduced.
"""
import contextlib
import math
import os
import time
from types import SimpleNamespace
from typing import NamedTuple

import numpy as np
import torch
from torch import nn

from . import ops, sharded
from .draws import ImageDraws, draw_image
from .hostcfg import limit_host_threads
from .views import PreparedBatch, ViewMixin, check_hf_interp
from .pyramid import Pyramid
from .projection_head import ProjectionHead, build_projection


class _NullLogger:
    def info(self, *_):
        pass


# output pixels of a view that the nearest rotate / flip map can send to ONE source pixel (two, rarely three: rotated lattice
# points inside a unit cell), rounded up: the factor between the largest single row a scatter adds into G and a bound of |G|
SCATTER_MEET = 4.0


class Route(NamedTuple):
    """What one optimizer step runs — decided in ONE place (`SwAVClustering.plan_route`) from the configuration and the batch
    shape.  `train_step` dispatches on `name`; `bench.py` prints `describe()` into `config.workload`."""
    name: str               # 'lowrank' (K <= 16 on score tables of every pixel) | 'tables' (many prototypes on per-pixel
    #                         d(score) / score tables) | 'per_sample' (the reference-shaped chain per patch and view)
    mode: str               # projection_mode: 'pyramid' | 'panel'
    small: bool             # per_sample: the fused K <= 16 row kernels
    k_run: int              # prototype columns the kernels see (nprototypes padded to a multiple of 4 past 64 columns)
    planes: bool            # prototype weights as bf16x6 planes, once per step (weight-plane GEMMs)
    pad_ds: bool            # d(scores) rows padded to 16-column chunks: dz on the weight-plane kernel
    score_table: bool       # tables: the scores live on per-pixel tables too (one score GEMM per view)
    sinkhorn_table: bool    # tables: every Sinkhorn problem of the step solved on the tables, a row read once per pass
    pgroup: int             # patches per Sinkhorn batch
    fineq: bool             # lowrank: the finest pyramid level through quadratic forms of its taps (gx_fineq.hip)

    def describe(self):
        if self.name == 'lowrank':
            how = "score/rnorm tables of every pixel" + (", finest level as quadratic forms of its taps" if self.fineq else "")
        elif self.name == 'tables':
            how = ("d(scores)" + (" and scores" if self.score_table else "") + " on per-pixel tables, backward GEMMs over pixels" +
                   (", table Sinkhorn" if self.sinkhorn_table else ", per-problem Sinkhorn sweeps"))
        else:
            how = ("fused small-K row kernels" if self.small else "per-patch score / gradient GEMMs" +
                   (" on weight planes" if self.planes else ""))
        return f"route={self.name} ({self.mode} projection; {how}; {self.k_run} prototype columns, {self.pgroup} patch(es) per Sinkhorn batch)"


class SwAVClustering(ViewMixin):


    def __init__(self, model, model_config, perturb_args, swav_args, sinkhorn_args, logger=None, train=True,
                 out_dir=None, device='cuda', tb=None, layer_hf_dim=None, dist_group=None):
        self.model = model
        self.model_config = model_config
        self.perturb_args = perturb_args
        self.swav_args = swav_args
        # before anything touches a device: a config error, not a device error
        self._interp = check_hf_interp(swav_args, 'swav_args', allowed=('nearest', 'bilinear'))
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError("SwAVClustering (ganecdotes_amd) needs a HIP device: there is no CPU fallback")
        ops.lib()                                   # fail loudly if the HIP extension is missing
        limit_host_threads()                        # the host only feeds the GPU: no more threads than CPUs it may use (hostcfg.py)
        self.writer = tb
        self._hlen = swav_args['hlen']
        self.nclasses = swav_args['nclasses']
        self.nprototypes = swav_args['nprototypes']
        self.niters = sinkhorn_args['niters']
        self.eps = sinkhorn_args['eps']
        self.sinkhorn_args = sinkhorn_args.copy()
        self.logger = logger or _NullLogger()
        self.train = train
        self.out_dir = out_dir
        self.dist_group = dist_group
        self.world = 1
        self.rank = 0
        # a process group => the sharded protocol with its collectives, also when it has ONE rank (the
        # exchange steps then run through RCCL as identities: tests/test_hip_multirank.py)
        self.sharded = dist_group is not None
        if dist_group is not None:
            import torch.distributed as dist
            self.world = dist.get_world_size(dist_group)
            self.rank = dist.get_rank(dist_group)
        if out_dir is not None:
            self.swav_dir = os.path.join(out_dir, 'swav')
            os.makedirs(self.swav_dir, exist_ok=True)
            self.prototype_file = os.path.join(out_dir, 'prototypes.pt')
            self.projection_file = os.path.join(out_dir, 'projection.pt')
        else:
            self.prototype_file = self.projection_file = None
        self.projection = None
        self.prototype = None
        if not self.train and self.projection_file is not None:
            if os.path.exists(self.projection_file):
                self.projection = torch.load(self.projection_file, weights_only=False).to(self.device)
                self.prototype = torch.load(self.prototype_file, weights_only=False).to(self.device)
            else:
                self.logger.info("Prototype File not found - pretraining ...")
        with torch.no_grad():
            self.mean_latent = self.model.mean_latent(self.model_config.num_latents_for_mean)
            self.truncation = self.model_config.truncation
        self.feature_peak = self.probe_feature_range()          # a checkpoint outside the f16x3 window -> bf16x6, with a warning
        self.layer_hf_dim = layer_hf_dim
        self._bufs = None
        self._step_no = 0
        self._sk_probe = None                        # a list: train_step appends the table Sinkhorn's last step statistics (tools/soak.py)

    def _patch_mode(self):
        """'patch' sampling (swav_clustering.py:150-158, :383-385): a square crop of patch_size x patch_size
        pixels at one random offset instead of patch_size random pixels; here it is only a different pick
        list (draws.patch_pixels).  patch_size None / == image side means every pixel in both modes."""
        sa = self.swav_args
        method = sa.get('sampling_method', 'random')
        if method not in ('random', 'patch'):
            raise NotImplementedError(f"sampling_method={method!r}: the reference knows 'random' and 'patch'")
        return method == 'patch' and sa['patch_size'] is not None and sa['patch_size'] < self.model.size

    def _rows_per_image(self):
        ps = self.swav_args['patch_size']
        return ps * ps if (ps is not None and self._patch_mode()) else ps

    def _pick_spec(self):
        sa = self.swav_args
        # a crop offset is one host draw per patch (numpy's stream, like the reference): never the device path
        return self._rows_per_image(), sa['num_patches'], ('host' if self._patch_mode() else sa.get('pick_source', 'host'))

    # ------------------------------------------------------------------------------------- modules
    def build_heads(self):
        """projection + prototype layers as torch modules (swav_clustering.py:244-271) — the
        objects that are pickled to projection.pt / prototypes.pt."""
        hlen = self.swav_args['hlen']
        self.projection = build_projection(self.swav_args['projn_nw'], hlen, self.nclasses).to(self.device)
        self.prototype = nn.Linear(self.nclasses, self.nprototypes).to(self.device)
        self._head = None
        for p in list(self.projection.parameters()) + list(self.prototype.parameters()):
            p.requires_grad_(False)
        self._bufs = None
        self._step_no = 0

    def _proj_head(self):
        """the layers after the first Linear (projn_nw '1-layer' / '2-layer'; 'linear': none) — projection_head.py"""
        if getattr(self, '_head', None) is None or self._head.projection is not self.projection:
            self._head = ProjectionHead(self.projection, self.swav_args.get('projn_nw', 'linear'))
        return self._head

    def _params(self):
        """[Wp, Wc, b] and, for projn_nw='2-layer', the BatchNorm affines and the second Linear after them"""
        return [self.projection[0].weight.data, self.prototype.weight.data, self.prototype.bias.data] + \
            self._proj_head().extra_params()

    # ------------------------------------------------------------------------------------- reference API
    def preprocess(self, input_latent):
        if self.train or self.projection_file is None or not os.path.exists(self.projection_file):
            self.pretrain(input_latent)

    # ------------------------------------------------------------------------------------- views
    def create_hidden_features_from_perturbed_vectors(self, layer_no=None, input_latent=None,
                                                      input_is_latent=True):
        """Reference-shaped single-view helper (swav_clustering.py:574-656): returns
        (stack (1,hlen,H,W), image, layer_no).  Draws from the global RNGs like the reference."""
        sd = self.model.style_dim
        if input_latent is None:
            input_latent = torch.randn(1, sd).to(self.device)
            input_is_latent = False
        w = input_latent if input_is_latent else self.model.style(input_latent)
        im = draw_image(sd, self.model.num_layers, self.perturb_args['n_layers'], 1, 0, transforms=False,
                        layer_no=layer_no)
        vd = im.views[0]
        mix = torch.zeros(1, self.model.n_latent, 2)
        sig = float(self.perturb_args['perturb_std'][vd.layer_no])
        mix[0, 2 * vd.layer_no, 0] = mix[0, 2 * vd.layer_no + 1, 1] = sig
        lat = self._view_latents(w.reshape(1, sd), vd.map_noise.to(self.device), mix.to(self.device))
        img, feats = self.model([lat], truncation=self.truncation, truncation_latent=self.mean_latent,
                                input_is_latent=True, randomize_noise=False)
        return self.create_pixel_feature_vectors(feats), img, vd.layer_no

    # ------------------------------------------------------------------------------------- collectives
    def _reduce(self, t, op):
        if getattr(self, "_reduce_fn", None) is None:
            self._reduce_fn = sharded.make_reduce(self.dist_group)
        self._reduce_fn(t, op)

    # ------------------------------------------------------------------------------------- one step
    def _marginals(self, img, k, nrows, shared=False):
        """'image' source pdf (swav_clustering.py:523-532): r from a K-bin, c from an nrows-bin
        histogram of the whole norm map."""
        red = self._reduce if (self.sharded and shared) else None
        r = ops.hist_marginal(img, k, red)
        c = ops.hist_marginal(img, nrows * (self.world if shared else 1), red)
        if red is not None:             # this rank's rows of the global per-sample marginal
            c = c[self.rank * nrows:(self.rank + 1) * nrows].contiguous()
        return r, c

    def _split_k(self, m, n, k):
        """k split of a long-reduction TN GEMM (m x n outputs): enough workgroups to fill the chip, and — these GEMMs run
        the wave-specialised 256 x 128 tiles, ONE workgroup per CU — a count that fills whole rounds of the 256 CUs
        (K = 5000, m x n = 5000 x 512: 80 tiles; 6 splits = 480 workgroups = 1.9 rounds ran 8.8 ms, 16 = 5 rounds 8.4 ms)"""
        tiles128 = ((m + 127) // 128) * ((n + 127) // 128)
        want = max(1, 1024 // tiles128)
        cap = max(1, k // 256)
        if m <= 128 or k // max(1, min(want, cap)) < 768:
            return int(max(1, min(want, cap)))
        tiles = ((m + 255) // 256) * ((n + 127) // 128)
        best, best_eff = 1, 0.0
        for s in range(max(1, want // 2), min(cap, 4 * want, max(1, k // 768)) + 1):
            wgs = tiles * s
            eff = wgs / (((wgs + 255) // 256) * 256.0)
            if wgs >= 256 and eff > best_eff + 1e-9:
                best, best_eff = s, eff
        return int(best if best_eff > 0 else max(1, min(want, cap)))

    def _score_tables(self, pb, views, Zt, P, B, hw, N, D, K, wc_planes, bc, c_one, by_mask=False, by_image=False):
        """The per-pixel score tables of a many-prototype step (swav_clustering.py:171-182 once per view over the
        pixels instead of once per patch over the samples): the drawn pixels first, their count on the device, each patch's samples
        as sorted table rows, z_hat of the table's pixels and `S = z_hat Wc^T + b` per view.  c_one (2, rows): the 'image' source
        pdf's row marginals (None = uniform).  by_mask: the drawn pixels grouped by which patches drew them (what the table
        Sinkhorn wants), by_image: image by image first (`seg`: an image's run of the table; the per-image scope).
        Returns None when the weight-plane GEMM does not take the shape."""
        dev = self.device
        npix = B * hw * hw
        pix_base = (torch.arange(B, dtype=torch.int32, device=dev) * (hw * hw)).unsqueeze(1)
        # Table rows = the pixels some patch drew, FIRST (84 % of them at 5 x 20 000 of 65 536), in pixel order; their count
        # `live` stays on the device and bounds the three GEMMs there (no host sync: the launches cover all B * HW rows)
        flat = [(pb.picks[pno] + pix_base).reshape(-1).long() for pno in range(P)]
        drawn = torch.zeros(npix, dtype=torch.bool, device=dev)
        for f_ in flat:
            drawn[f_] = True
        seg = None
        if by_mask and P <= 14:
            # ... and, for the table Sinkhorn (csrc/gx_sktab.hip), the drawn pixels grouped by WHICH patches drew them: the rows of
            # a 4-row group then go through the same patches together.  Per-image scope: image by image (an image's problems live
            # on ITS run of the table, `seg`), by membership inside an image.
            mkey = torch.zeros(npix, dtype=torch.int32, device=dev)
            for p_, f_ in enumerate(flat):
                mkey[f_] += 1 << p_
            if by_image:
                img = torch.arange(npix, dtype=torch.int32, device=dev) // (hw * hw)
                key = torch.where(mkey > 0, (img << 14) + mkey, torch.full_like(mkey, 1 << 30))
                seg = torch.zeros(B + 1, dtype=torch.int32, device=dev)
                seg[1:] = torch.cumsum(drawn.view(B, hw * hw).sum(1, dtype=torch.int32), 0, dtype=torch.int32)
            else:
                key = torch.where(mkey > 0, mkey, torch.full_like(mkey, 1 << 30))
            order = torch.argsort(key, stable=True)
        else:
            order = torch.argsort((~drawn).to(torch.int8), stable=True)               # drawn pixel ids first
        live = drawn.sum(dtype=torch.int32).reshape(1)
        live_frac = 1.0 - (1.0 - N / float(hw * hw)) ** P                            # its expectation (FLOP accounting)
        ar32 = torch.arange(npix, dtype=torch.int32, device=dev)
        cid = torch.empty(npix, dtype=torch.int32, device=dev)
        cid[order] = ar32                                                            # pixel id -> table row
        # a patch's samples as table rows, ASCENDING: a Sinkhorn problem does not care about the order of its rows, and sorted the
        # patches of a view walk the table side by side (ops.SinkhornBatch.from_rows)
        # (with the 'image' source pdf the row marginal belongs to the sample's POSITION in its patch: it moves with the row)
        sorted_all = [torch.sort(cid[f_]) for f_ in flat]
        rowmaps_all = [s_.values for s_ in sorted_all]
        if c_one is None:
            c_sorted = None
        elif c_one.dim() == 2:                 # global scope: (2, rows)
            c_sorted = [[c_one[v][s_.indices] for v in range(2)] for s_ in sorted_all]
        else:                                  # per-image scope: (2, B, N) — an image's samples stay together under the sort
            c_sorted = [[torch.gather(c_one[v], 1, s_.indices.view(B, N) - (torch.arange(B, device=dev) * N).unsqueeze(1))
                         for v in range(2)] for s_ in sorted_all]
        del sorted_all
        row_group = (order // (hw * hw)).to(torch.int32)                             # table row -> image (per-image Sinkhorn scope)
        order32 = order.to(torch.int32)
        sentinel = torch.full((1,), -1, dtype=torch.int32, device=dev)
        scat_picks = torch.where(ar32 < live, order32, torch.full_like(order32, npix)).view(1, npix)   # rows past live: dropped
        maps_flat = []
        for v in range(2):
            m_ = views[v][1]
            mf = ar32 if m_ is None else torch.where(m_ >= 0, m_ + pix_base, m_).reshape(-1)
            maps_flat.append(torch.cat([mf, sentinel]).view(1, npix + 1))
        zt_tab, rn_tab, S_tab = [], [], []
        for v in range(2):
            with self._timed("gather_rows"):       # z_hat of the table's pixels (the image batch as ONE image of B * H rows)
                zt, rnt = ops.gather_rows_normalize(Zt[v].reshape(1, B * hw, hw, D), order32.view(1, npix), maps_flat[v])
            st = torch.empty(npix, K, dtype=torch.float32, device=dev)
            if not ops.gemm_planes_ok(zt, None, K, st):
                return None
            ops.gemm_nt_planes_live(zt, wc_planes, st, live, live_frac, col_bias=bc, a_range=1.0)      # |z_hat| <= 1
            zt_tab.append(zt); rn_tab.append(rnt); S_tab.append(st)
        return SimpleNamespace(live=live, live_frac=live_frac, seg=seg, rowmaps=rowmaps_all, c_sorted=c_sorted, scat_picks=scat_picks,
                               maps_flat=maps_flat, zt=zt_tab, rn=rn_tab, S=S_tab, row_group=row_group)

    def _tables_backward(self, dS_tab, tb, views, Zt, Gt, Wc, wct_planes, gWc, B, hw, ds_range):
        """Backward of the prototype layer from the per-pixel d(scores) tables (swav_clustering.py:459 through
        :171-182), once per view over the pixels: gWc = dS^T z_hat, dz = dS Wc, normalise-backward + scatter into the projected
        tables Gt through the rotate / flip maps.  tb: the score tables of _score_tables (their z_hat, pixel order and live
        count are reused) or None (z_hat of every pixel gathered here)."""
        dev = self.device
        K, D = Wc.shape
        npix = B * hw * hw
        picks_all = None if tb is not None else torch.arange(hw * hw, dtype=torch.int32, device=dev).repeat(B, 1)
        # the largest single row the scatters add into the G tables: a table row is scattered once, and the nearest rotate / flip
        # map sends at most two or three output pixels to one source pixel — x SCATTER_MEET it bounds |G| for the level GEMMs
        g_amax = torch.zeros(1, dtype=torch.float32, device=dev) if ops.GEMM_H2 else None
        for v in range(2):
            dz = torch.empty(npix, D, dtype=torch.float32, device=dev)
            if tb is not None:
                zt, rnt = tb.zt[v], tb.rn[v]
                ops.gemm(dS_tab[v][:, :K], zt, True, False, out=gWc, beta=0.0 if v == 0 else 1.0,
                         split_k=self._split_k(K, D, npix), live=tb.live, live_frac=tb.live_frac, a_range=ds_range, b_range=1.0)
                ops.gemm_nt_planes_live(dS_tab[v], wct_planes, dz, tb.live, tb.live_frac, a_range=ds_range)
                with self._timed("scatter_rows"):
                    ops.normalize_bwd_scatter_(dz, zt, rnt, tb.scat_picks, tb.maps_flat[v], Gt[v].reshape(1, B * hw, hw, D), amax=g_amax)
                continue
            with self._timed("gather_rows"):
                zt, rnt = ops.gather_rows_normalize(Zt[v], picks_all, views[v][1])       # z_hat of every (output) pixel
            ops.gemm(dS_tab[v][:, :K], zt, True, False, out=gWc, beta=0.0 if v == 0 else 1.0, split_k=self._split_k(K, D, npix),
                     a_range=ds_range, b_range=1.0)
            if ops.gemm_planes_ok(dS_tab[v], None, D, dz):
                ops.gemm_nt_planes(dS_tab[v], None, wct_planes, dz, a_range=ds_range)
            else:
                ops.gemm(dS_tab[v][:, :K], Wc, False, False, out=dz)
            with self._timed("scatter_rows"):
                ops.normalize_bwd_scatter_(dz, zt, rnt, picks_all, views[v][1], Gt[v], amax=g_amax)
            del zt, rnt, dz
        return g_amax

    # ------------------------------------------------------------------------------------- route of a step
    def plan_route(self, B, hw, N):
        """Which kernels a step over B images of hw x hw pixels with N samples per patch and image will run — decided HERE,
        once, from the configuration and the shape (`Route`, above); train_step dispatches on it and bench.py prints it."""
        sa = self.swav_args
        P = sa['num_patches']
        head = self._proj_head()
        Wp, Wc = self.projection[0].weight.data, self.prototype.weight.data
        D, K = Wp.shape[0], Wc.shape[0]
        mode = sa.get('projection_mode', 'pyramid')
        if mode not in ('pyramid', 'panel'):
            raise NotImplementedError(f"projection_mode={mode!r}: 'pyramid' or 'panel'")
        if self._interp != 'nearest':
            mode = 'panel'      # the pyramid factorisation IS the nearest upsampling (a pixel's descriptor = its ancestors' taps)
        # the few-prototype routes (fused gather + scores, score tables) need a LINEAR map from descriptor to z
        small = bool(sa.get('small_proto_kernels', True) and ops.proto_small_ok(K, D) and head.linear)
        lowrank = bool(mode == 'pyramid' and sa.get('lowrank_proto', True) and ops.lowrank_ok(K, D) and head.linear)
        rows = B * N
        # Past 64 columns the row kernels (Sinkhorn sweeps, fused CE, the weight-plane GEMMs) read 16-byte pieces of a row: the
        # prototype count is padded to a multiple of 4 with columns that can take no mass (zero weights, bias -16, marginal 1e-30;
        # _pad_prototypes) instead of falling to the team-per-row kernels (K = 4001: 132 ms per Sinkhorn against 10 at 4000)
        k_run = K if (K <= 64 or K % 4 == 0) else (K + 3) // 4 * 4
        # The pyramid path holds only (rows, K) scores per patch, so all patches go through ONE batched Sinkhorn (as many as fit
        # 16 GB of scores); the panel path keeps a (rows, hlen) descriptor panel per view alive until its backward: patch by patch
        pgroup = max(1, min(P, int(16e9 // (8.0 * rows * k_run)))) if mode == 'pyramid' else 1
        if lowrank:
            return Route('lowrank', mode, small, K, False, False, False, False, pgroup, bool(sa.get('fineq', True)))
        # generic-K scores: the prototype weights as bf16x6 planes, once per step (gemm_xw_kernel)
        planes = bool(not (mode == 'pyramid' and small) and ops.get_gemm_mode() == 1 and D % 16 == 0)
        # dz = dS Wc as the NT contraction dS x (Wc^T)^T on the weight-plane kernel: dS rows padded to whole 16-column
        # chunks (zero pad), planes of Wc^T once per step (many prototypes; 7.95 -> 7.45 ms at K = 5000)
        pad_ds = planes and k_run % 4 == 0 and k_run > 64
        Kp = (k_run + 15) // 16 * 16
        # Many prototypes: d(scores) of every patch land on per-PIXEL rows (one table per view): a pixel drawn by
        # several patches has its rows added up by the CE kernel, and the two backward GEMMs (dWc, dz) run ONCE over the
        # B * HW pixels instead of once per patch over P * B * N samples (ffhq: 65 536 pixels against 100 000 samples per image)
        # (memory: the two d(scores) tables + either the two score tables or every patch's per-sample scores, held to half the HBM)
        want_stab = bool(sa.get('score_table', True))
        tab_bytes = 8.0 * B * hw * hw * Kp + (8.0 * B * hw * hw * k_run if want_stab else 8.0 * rows * k_run * P)
        # worth it where the tables' live rows are fewer than the samples: always with more samples than pixels, and with uniformly
        # drawn picks where the patches overlap enough — the pixels some patch drew, HW (1 - (1 - N / HW)^P) in expectation, against
        # the P N samples (the kernels only touch the drawn pixels: ffhq 0.55, the car configuration's 20 000 of 262 144 pixels x 5
        # patches 0.86: 94.7 -> 87.9 ms per step)
        live_exp = hw * hw * (1.0 - (1.0 - min(1.0, N / float(hw * hw))) ** P)
        tab_default = hw * hw < P * N or (sa.get('sampling_method', 'random') == 'random' and live_exp <= 0.9 * P * N)
        dstab = bool(mode == 'pyramid' and pad_ds and not small and head.linear and 64 < k_run <= 8192 and P <= 8 and
                     sa.get('ds_table', tab_default) and
                     tab_bytes < 0.5 * torch.cuda.get_device_properties(self.device).total_memory)
        if not dstab:
            return Route('per_sample', mode, small, k_run, planes, pad_ds, False, False, pgroup, False)
        # ... and, with one Sinkhorn problem per patch and view (global scope) or per image too, the SCORES are a per-pixel table:
        # one score GEMM per view over the pixels, the Sinkhorn problems and the CE read their rows of it through the pick index;
        # every patch's scores stay until the one pixel-major CE pass of the step (pgroup = P)
        sk_tab = bool(want_stab and sa.get('sinkhorn_table', True) and ops.sinkhorn_table_ok(P, k_run))
        return Route('tables', mode, small, k_run, planes, pad_ds, want_stab, sk_tab, P, False)

    def _pad_prototypes(self, c):
        """K -> route.k_run columns: rows of zeros under Wc, bias -16 — a padded column's score is -16 for every
        sample: exp((-16 - max) / T) and exp((-16 - max) / eps) are 0 in fp32 for every T, eps <= 0.1 against scores that are
        cosines + a small bias — and EXPLICIT marginals whose padded entries are 1e-30 (a Sinkhorn column receives exactly its
        marginal, so the padded columns hold 1e-30 of a code's mass whatever their score; 0 would make log r = -inf and
        poison the step statistics of the shifted sweeps).  The real columns keep r = 1 / K (or the 'image' histogram over
        K bins): the problem the reference poses (swav_clustering.py:522-537), to 1e-30.  Gradients come back through
        _unpad_prototypes."""
        K, Kr, dev = c.K, c.route.k_run, self.device
        c.K_true = K
        c.Wc_true, c.bc_true, c.gWc_true, c.gb_true = c.Wc, c.bc, c.gWc, c.gb
        Wc = torch.zeros(Kr, c.D, dtype=torch.float32, device=dev)
        Wc[:K] = c.Wc
        bc = torch.full((Kr,), -16.0, dtype=torch.float32, device=dev)
        bc[:K] = c.bc
        c.Wc, c.bc, c.K = Wc, bc, Kr
        c.gWc = torch.empty_like(Wc)
        c.gb = torch.empty_like(bc)
        if c.r_one is None:                        # uniform marginals, spelled out
            shp_r = (2, K) if c.ngroups == 1 else (2, c.B, K)
            shp_c = (2, c.rows) if c.ngroups == 1 else (2, c.B, c.N)
            c.r_one = torch.full(shp_r, 1.0 / K, dtype=torch.float32, device=dev)
            c.c_one = torch.full(shp_c, 1.0 / c.nglob, dtype=torch.float32, device=dev)
        pad = torch.full((*c.r_one.shape[:-1], Kr - K), 1e-30, dtype=torch.float32, device=dev)
        c.r_one = torch.cat([c.r_one, pad], dim=-1).contiguous()

    def _unpad_prototypes(self, c):
        K = c.K_true
        c.gWc_true.copy_(c.gWc[:K])
        c.gb_true.copy_(c.gb[:K])
        for key in ('scores_s', 'scores_t', 'q_s', 'q_t'):
            if key in c.aux:
                c.aux[key] = c.aux[key][:, :K].contiguous()
        c.Wc, c.bc, c.gWc, c.gb, c.K = c.Wc_true, c.bc_true, c.gWc_true, c.gb_true, K

    # ------------------------------------------------------------------------------------- one step
    def train_step(self, images, return_aux=False):
        """One optimizer step over this rank's images (list of ImageDraws).  Returns the loss
        as a 1-element HIP tensor (global mean; no host sync).  Prologue (views, pyramid forward, marginals) -> ONE of
        `_step_lowrank` / `_step_tables` / `_step_per_sample`, chosen by plan_route -> epilogue (pyramid backward, gradient
        all-reduce, LARC-SGD)."""
        c = self._step_prologue(images, return_aux)
        getattr(self, '_step_' + c.route.name)(c)
        return self._step_epilogue(c)

    def _step_prologue(self, images, return_aux):
        sa = self.swav_args
        c = SimpleNamespace(sa=sa, return_aux=return_aux, aux={}, first_acc=True)
        c.P, c.N, c.hlen, c.T = sa['num_patches'], self._rows_per_image(), sa['hlen'], sa['temperature']
        c.scope = sa.get('sinkhorn_scope', 'image')
        c.pb = pb = self.prepare_batch(images)
        c.B = B = pb.n
        c.Bg = B * self.world
        c.Wp, c.Wc, c.bc, *c.extra = self._params()
        c.head = self._proj_head()
        c.D, c.K = c.Wp.shape[0], c.Wc.shape[0]
        dev = self.device

        # S7: prototype rows to unit norm (bias untouched)
        ops.row_l2_normalize_(c.Wc, want_rnorm=False)

        c.views, ts_all, imgs = self.take_views(pb)      # prefetched on the view stream (prefetch_views) or made here
        c.hw = hw = c.views[0][0].size
        if c.N is None or c.N >= hw * hw:
            c.N = hw * hw
        if imgs is None and self.sinkhorn_args['source_pdf'] == 'image':
            imgs = [ops.norm_map(ts, c.hlen, maps, interp=self._interp) for ts, maps, _ in c.views]

        if self._bufs is None:
            self._bufs = [torch.zeros_like(p) for p in (c.Wp, c.Wc, c.bc, *c.extra)]
            self._grads = [torch.empty_like(p) for p in (c.Wp, c.Wc, c.bc, *c.extra)]
        c.gWp, c.gWc, c.gb, *c.gextra = self._grads
        c.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        c.rows = B * c.N
        c.gscale = 1.0 / (c.N * c.P * c.Bg)
        c.route = route = self.plan_route(B, hw, c.N)
        self.last_route = route
        if sa.get('deterministic', False) and route.name == 'per_sample':
            # this route scatters 2 KB rows with float atomics (normalize_bwd_scatter) and sums the Sinkhorn columns with them:
            # a promise it cannot keep is refused.  'lowrank' switches its coefficient scatter to fixed point; 'tables' needs
            # nothing — every d(scores) row is written once, by the pixel's own thread, and every block sum has a fixed order
            raise NotImplementedError("swav_args['deterministic']=True: bitwise run-to-run reproducibility holds on the routes "
                                      f"'lowrank' (K <= 16) and 'tables' (64 < K <= 8192); this step would take {route.describe()}")
        c.mode = route.mode
        c.pyr = c.Z_all = c.Zt = c.G_all = c.Gt = None
        if route.mode == 'pyramid':
            with self._timed("pyramid_fwd"):
                # ONE pyramid over the 2B images of both views: each level is one GEMM, the table Z_all
                # (2B, H, W, D) holds view s in [0, B) and view t in [B, 2B)
                c.pyr = Pyramid(ts_all, c.hlen)
                if route.name == 'lowrank':
                    # K <= 16: the score / rnorm tables of every pixel of the table level; the finest level's projected
                    # table is never formed (pyramid.py, csrc/gx_fineq.hip) and levels without taps are skipped (`lshift`)
                    c.rn_all, c.sc_all, c.lshift = c.pyr.project_tables(c.Wp, c.Wc, c.bc, quad=route.fineq)
                else:
                    c.Z_all = c.pyr.project(c.Wp)
                    c.Zt = [c.Z_all[v * B:(v + 1) * B] for v in range(2)]          # (B, H, W, D) per view
            if route.name != 'lowrank':
                c.G_all = torch.zeros_like(c.Z_all)
                c.Gt = [c.G_all[v * B:(v + 1) * B] for v in range(2)]
        c.ngroups = 1 if c.scope == 'global' else B                  # Sinkhorn problems per (patch, view)
        c.sk_reduce = self._reduce if (c.scope == 'global' and self.sharded) else None    # also: BatchNorm batch = Sinkhorn problem
        c.grows = c.rows // c.ngroups
        c.nglob = c.grows * (self.world if c.scope == 'global' else 1)
        # 'image' source pdf: the marginals depend on the view (and image), not on the patch
        c.r_one = c.c_one = None
        if imgs is not None:
            if c.scope == 'global':
                marg = [self._marginals(imgs[v].contiguous(), c.K, c.rows, shared=True) for v in range(2)]
                c.r_one = torch.stack([m[0] for m in marg])                       # (2, K)
                c.c_one = torch.stack([m[1] for m in marg])                       # (2, rows)
            else:
                marg = [[self._marginals(imgs[v][b:b + 1].contiguous(), c.K, c.N) for b in range(B)] for v in range(2)]
                c.r_one = torch.stack([torch.stack([m[0] for m in mv]) for mv in marg])   # (2, B, K)
                c.c_one = torch.stack([torch.stack([m[1] for m in mv]) for mv in marg])   # (2, B, N)
            if return_aux:                                            # what the 'image' pdf prescribes (tests)
                c.aux.update(norm_map_s=imgs[0], norm_map_t=imgs[1], r_marginals=c.r_one, c_marginals=c.c_one)
        c.K_true = c.K
        if route.k_run != c.K:
            self._pad_prototypes(c)
        return c

    def _step_epilogue(self, c):
        sa = self.swav_args
        if c.K_true != c.K:
            self._unpad_prototypes(c)
        pending = []
        if c.mode == 'pyramid':
            on_level = on_heads = None
            if self.sharded:
                # once the prototype gradients and the loss are final their all-reduce travels while the
                # pyramid backward runs, and every pyramid level's slice of dWp follows as soon as its GEMMs
                # are queued (finest levels first), so only the coarsest levels' message is exposed
                on_heads = lambda: pending.append(sharded.allreduce_flat_async([c.gWc, c.gb, c.loss] + c.gextra, self.dist_group))
                on_level = lambda cols: pending.append(sharded.allreduce_cols_async(c.gWp, cols, self.dist_group))
            with self._timed("pyramid_bwd"):
                if c.route.name == 'lowrank':      # from the coefficient grid: dense finish at the table level, the finest level's Gram forms
                    c.pyr.backward_lowrank(c.Ct_all, c.Wp, c.Wc, c.gWp, c.gWc, on_level=on_level, on_heads=on_heads)
                else:
                    if on_heads is not None:
                        on_heads()
                    # (tables route: the scatters left the largest single row of G on the device -> the level GEMMs' f16x3 form)
                    ga = getattr(c, "g_amax", None)
                    c.pyr.backward(c.G_all, c.gWp, accumulate=False, on_level=on_level, g_amax=ga, g_mul=SCATTER_MEET)
            c.Zt = c.Z_all = c.G_all = c.Gt = None
        elif self.sharded:
            pending.append(sharded.allreduce_flat_async(list(self._grads) + [c.loss], self.dist_group))
        for fin in pending:
            fin()
        # S8: LARC + momentum SGD
        lr = sa['train_args']['lr']
        mom = sa['train_args'].get('momentum', 0.0)
        if sa.get('use_scheduler', False) and getattr(self, '_lr_schedule', None) is not None:
            lr = float(self._lr_schedule[min(self._step_no, len(self._lr_schedule) - 1)])
        for p, g, buf in zip((c.Wp, c.Wc, c.bc, *c.extra), self._grads, self._bufs):
            ops.larc_sgd_(p, g, buf, lr, mom, sa['trust_coeff'], first_step=(self._step_no == 0), use_larc=sa.get('use_larc', True))
        self._step_no += 1
        if getattr(self, "_prefetched_step", False):     # a pipelined step: its end bounds how far prefetch_views may run ahead
            self._prefetched_step = False
            ev = torch.cuda.Event()
            ev.record()
            self._step_done.append(ev)
        return (c.loss, c.aux) if c.return_aux else c.loss

    # --- route 'lowrank': K <= 16, pyramid mode, linear projection ---
    def _step_lowrank(self, c):
        """gather / Sinkhorn / loss / coefficient scatter of all patches and both views in single launches on the score and
        rnorm tables of every pixel; the backward of the prototype layer and of the projection comes from the coefficient grid
        (`Pyramid.backward_lowrank`, epilogue)."""
        B, K, hw, dev = c.B, c.K, c.hw, self.device
        views = c.views
        maps_all = None
        if views[0][1] is not None or views[1][1] is not None:        # (2B, HW) rotate / flip index maps
            ident = None
            if views[0][1] is None or views[1][1] is None:
                ident = torch.arange(hw * hw, dtype=torch.int32, device=dev).repeat(B, 1)
            maps_all = torch.cat([m if m is not None else ident for m in (views[0][1], views[1][1])])
        c.Ct_all = ops.lowrank_coeffs(2 * B, c.pyr.table_pixels(), K, dev)   # (2B * table pixels, K+1 padded) coefficient grid
        # swav_args['deterministic']: the coefficient scatter accumulates fixed-point integers (order-independent); with the
        # fixed-order block sums everywhere else on this route two runs from the same state are then bitwise equal
        ct_fix = torch.zeros(c.Ct_all.shape, dtype=torch.int64, device=dev) if c.sa.get('deterministic', False) else None
        for p0 in range(0, c.P, c.route.pgroup):
            pnos = list(range(p0, min(c.P, p0 + c.route.pgroup)))
            G = len(pnos)
            Sc_all = torch.empty(G, 2, c.rows, K, dtype=torch.float32, device=dev)
            gpicks = c.pb.picks[pnos[0]:pnos[-1] + 1].contiguous()
            Rn_all = ops.gather_scores_batch(c.sc_all, c.rn_all, c.bc, gpicks, maps_all, B, hw * hw, Sc_all, shift=c.lshift)
            skb = self._sinkhorn_batch(c, Sc_all, G, {})
            dS_all = torch.empty_like(Sc_all)
            ops.swapped_ce_batch(Sc_all, c.ngroups, skb, c.T, c.gscale, c.loss, c.gscale, 1.0, dS_all)
            if c.return_aux and pnos[0] == 0:
                c.aux.update(scores_s=Sc_all[0, 0, :c.grows].clone(), scores_t=Sc_all[0, 1, :c.grows].clone(),
                             q_s=skb[0].codes(), q_t=skb[c.ngroups].codes())
            with self._timed("proto_bwd_lowrank"):
                ops.proto_bwd_lowrank_batch_(dS_all, Sc_all, Rn_all, c.bc, gpicks, maps_all, c.Ct_all, B, hw * hw, c.gb,
                                             0.0 if c.first_acc else 1.0, shift=c.lshift, coeffs_fix=ct_fix)
            c.first_acc = False
            del dS_all, Rn_all, Sc_all, skb

    def _sinkhorn_batch(self, c, Sc_all, G, cmax_pre):
        """S4 on per-sample score matrices: every (patch, view[, image]) problem of a patch group in one batched Sinkhorn;
        cmax_pre {(patch in group, view): (column maxima, first-pass sums)} where the score GEMM's epilogue left them"""
        K, ngroups, grows = c.K, c.ngroups, c.grows
        r_all = c_all = None
        if c.r_one is not None:
            r_all = c.r_one.unsqueeze(0).expand(G, *c.r_one.shape).reshape(G * 2 * ngroups, K).contiguous()
            c_all = c.c_one.unsqueeze(0).expand(G, *c.c_one.shape).reshape(G * 2 * ngroups, grows).contiguous()
        with self._timed("sinkhorn"):
            skb = ops.SinkhornBatch(Sc_all.reshape(G * 2 * ngroups, grows, K), self.eps, self.niters, r=r_all,
                                    c=c_all, c_uniform=1.0 / c.nglob)
            for (gi, v), (cm, uf) in cmax_pre.items():
                for g in range(ngroups):
                    sk_ = skb[(gi * 2 + v) * ngroups + g]
                    sk_.cmax_preset = cm if ngroups == 1 else cm[g]
                    sk_.u_first_preset = uf if ngroups == 1 else uf[g]
            skb.run(c.sk_reduce)
        return skb

    # --- route 'tables': many prototypes, d(scores) (and the scores) on per-pixel tables ---
    def _step_tables(self, c):
        """64 < K <= 8192, pyramid mode, linear projection: the CE of ALL patches pixel-major into one d(scores) table per view,
        the backward GEMMs once per view over the pixels; with `route.score_table` the score GEMM too runs once per view over
        the pixels and the Sinkhorn problems are row selections of the two tables (`route.sinkhorn_table`: solved ON the tables,
        every row read once per pass)."""
        B, K, hw, P, dev = c.B, c.K, c.hw, c.P, self.device
        rows, ngroups = c.rows, c.ngroups
        Kp = (K + 15) // 16 * 16
        npix = B * hw * hw
        # f16x3 forms of the two big NT GEMMs: z_hat rows are unit vectors, and |d(scores)| <= P gscale / T (a pixel is in at
        # most P patches, |softmax - code| <= 1 per entry)
        wc_planes = ops.split_gemm_weights(c.Wc, h2=True)
        wct_planes = ops.split_gemm_weights(c.Wc.t().contiguous(), h2=True)
        ds_range = P * c.gscale / c.T
        dS_tab = [torch.empty(npix, Kp, dtype=torch.float32, device=dev) for _ in range(2)]
        pix_base = (torch.arange(B, dtype=torch.int32, device=dev) * (hw * hw)).unsqueeze(1)
        tb = None
        if c.route.score_table:
            tb = self._score_tables(c.pb, c.views, c.Zt, P, B, hw, c.N, c.D, K, wc_planes, c.bc, c.c_one,
                                    by_mask=c.route.sinkhorn_table, by_image=ngroups > 1)
        pnos = list(range(P))
        nrow = torch.arange(rows, dtype=torch.int32, device=dev)
        if tb is not None:
            rowmaps = [tb.rowmaps[pno] for pno in pnos]                               # a patch's samples -> table rows
            inv = torch.full((P, npix), -1, dtype=torch.int32, device=dev)           # table row -> its row in patch p's problem
            for gi in range(P):
                inv[gi].scatter_(0, rowmaps[gi].long(), nrow)
            with self._timed("sinkhorn"):
                if c.route.sinkhorn_table and (ngroups == 1 or tb.seg is not None):
                    # ALL problems of the step in one pass over the live rows of the two tables per iteration — a row several
                    # patches drew is read once (csrc/gx_sktab.hip); per-image scope: a workgroup stays inside
                    # one image's run of the table
                    r_tab = c_tab = None
                    if c.r_one is not None:              # problem index = (patch * 2 + view) * ngroups + image
                        if ngroups == 1:
                            r_tab = torch.stack([c.r_one[v] for _ in pnos for v in range(2)])
                        else:
                            r_tab = torch.stack([c.r_one[v][g] for _ in pnos for v in range(2) for g in range(ngroups)])
                        c_tab = torch.stack([tb.c_sorted[pno][v].reshape(-1) for pno in pnos for v in range(2)])
                    skb = ops.SinkhornTable(tb.S[0], tb.S[1], inv, tb.live, self.eps, self.niters, 1.0 / c.nglob, rows,
                                            r=r_tab, c=c_tab, rowmaps=rowmaps if c.return_aux else None, ngroups=ngroups,
                                            seg=tb.seg)
                    with ops.record("sinkhorn_pass_kernel", 0.0, 8.0 * npix * K * tb.live_frac * self.niters, launches=self.niters):
                        skb.run(c.sk_reduce)
                    if self._sk_probe is not None:                            # tools/soak.py: what the last update vouched for
                        self._sk_probe.append(skb.stats.clone())
                else:
                    # problems in (patch, view, image) order; per-image scope: an image's N samples are a contiguous run of the
                    # patch's sorted rows (its pixels are a contiguous run of the table)
                    tabs, maps, rl, cl = [], [], [], []
                    for gi, pno in enumerate(pnos):
                        for v in range(2):
                            for g in range(ngroups):
                                tabs.append(tb.S[v])
                                maps.append(rowmaps[gi] if ngroups == 1 else rowmaps[gi].view(B, c.N)[g])
                                if c.r_one is not None:
                                    rl.append(c.r_one[v] if ngroups == 1 else c.r_one[v][g])
                                    cl.append(tb.c_sorted[pno][v] if ngroups == 1 else tb.c_sorted[pno][v][g])
                    skb = ops.SinkhornBatch.from_rows(tabs, maps, self.eps, self.niters, 1.0 / c.nglob, sorted_rows=True,
                                                      r=rl or None, c=cl or None)
                    if ngroups == 1:
                        for sk_ in skb.probs:  # column maxima + first pass of every problem in ONE sweep of its rows
                            sk_.cmax_preset, sk_.u_first_preset = ops.sinkhorn_first_rows(sk_.S, self.eps, sk_.rowmap)
                    # (per-image scope: 2 B P small problems — their maxima and first pass are two more launches of the batch)
                    skb.run(c.sk_reduce)
                    del tabs, maps, rl, cl
            ops.swapped_ce_pixels_tab(tb.S[0], tb.S[1], skb, ngroups, inv, c.T, c.gscale, c.loss, c.gscale, 1.0, dS_tab[0], dS_tab[1],
                                      Kp, c.gb, 0.0, live=tb.live, live_frac=tb.live_frac,
                                      row_group=tb.row_group if ngroups > 1 else None)
            if c.return_aux:
                c.aux.update(scores_s=skb[0].scores(), scores_t=skb[ngroups].scores(), q_s=skb[0].codes(),
                             q_t=skb[ngroups].codes())
            del inv, skb, rowmaps
        else:
            # per-sample scores of every patch (one score GEMM per patch and view, the Sinkhorn maxima and first pass from its
            # epilogue), then ONE pass over them all: loss, bias gradient, and each pixel's d(scores) summed over the patches
            # that drew it, written once (inv[p][pixel] = the pixel's row in patch p, -1 = not drawn)
            Sc_all = torch.empty(P, 2, rows, K, dtype=torch.float32, device=dev)
            cmax_pre = {}
            for gi, pno in enumerate(pnos):
                for v in range(2):
                    with self._timed("gather_rows"):
                        z, _ = ops.gather_rows_normalize(c.Zt[v], c.pb.picks[pno], c.views[v][1])  # z_hat (B*N, D)
                    self._scores_gemm(c, z, Sc_all[gi, v], wc_planes, cmax_pre, (gi, v))
                    del z
            skb = self._sinkhorn_batch(c, Sc_all, P, cmax_pre)
            inv = torch.full((P, npix), -1, dtype=torch.int32, device=dev)
            for gi, pno in enumerate(pnos):
                inv[gi].scatter_(0, (c.pb.picks[pno] + pix_base).reshape(-1).long(), nrow)
            ops.swapped_ce_pixels(Sc_all, skb, ngroups, inv, c.T, c.gscale, c.loss, c.gscale, 1.0, dS_tab[0], dS_tab[1], Kp, c.gb, 0.0)
            if c.return_aux:
                c.aux.update(scores_s=Sc_all[0, 0, :c.grows].clone(), scores_t=Sc_all[0, 1, :c.grows].clone(),
                             q_s=skb[0].codes(), q_t=skb[ngroups].codes())
            del inv, Sc_all, skb
        c.first_acc = False
        c.g_amax = self._tables_backward(dS_tab, tb, c.views, c.Zt, c.Gt, c.Wc, wct_planes, c.gWc, B, hw, ds_range)

    def _scores_gemm(self, c, z, sc, wc_planes, cmax_pre, key):
        """scores = z_hat Wc^T + b into sc (rows, K)"""
        K = c.K
        if wc_planes is not None and ops.gemm_planes_ok(z, None, K, sc):
            # Wc split once per step; the Sinkhorn column maxima AND its first pass (the column sums of
            # exp((scores - max) / eps)) come out of the epilogue: nine sweeps of the score matrix, not eleven
            if K > 64 and (c.ngroups == 1 or c.grows % 32 == 0):
                _, cm_, uf_ = ops.gemm_nt_planes(z, None, wc_planes, sc, col_bias=c.bc, a_range=1.0,
                                                 want_colmax=True if c.ngroups == 1 else c.ngroups, sk_eps=self.eps)
                cmax_pre[key] = (cm_, uf_)
            else:
                ops.gemm_nt_planes(z, None, wc_planes, sc, col_bias=c.bc, a_range=1.0)
        else:
            ops.gemm(z, c.Wc, False, True, col_bias=c.bc, out=sc)

    # --- route 'per_sample': everything else (panel mode, projection heads, 16 < K <= 64, shapes the tables do not take) ---
    def _step_per_sample(self, c):
        """The reference-shaped chain per patch and view (swav_clustering.py:377-418 with :133-182): gather (or the N x hlen panel
        and its GEMM) -> [projection head] -> normalise -> scores -> batched Sinkhorn -> swapped CE -> backward GEMMs -> scatter
        into the projected tables (pyramid mode) or dWp (panel mode)."""
        B, K, D, hw, dev = c.B, c.K, c.D, c.hw, self.device
        rows, ngroups, grows = c.rows, c.ngroups, c.grows
        mode, head, small, views = c.mode, c.head, c.route.small, c.views
        Wc, bc, gWc, gb = c.Wc, c.bc, c.gWc, c.gb
        wc_planes = ops.split_gemm_weights(Wc) if c.route.planes else None
        pad_ds = c.route.pad_ds
        Kp = (K + 15) // 16 * 16
        wct_planes = ops.split_gemm_weights(Wc.t().contiguous()) if pad_ds else None
        for p0 in range(0, c.P, c.route.pgroup):
            pnos = list(range(p0, min(c.P, p0 + c.route.pgroup)))
            G = len(pnos)
            Sc_all = torch.empty(G, 2, rows, K, dtype=torch.float32, device=dev)
            keep = []
            cmax_pre = {}
            for gi, pno in enumerate(pnos):
                picks = c.pb.picks[pno]
                X, Zh, Rn, Hc = [], [], [], []
                for v, (ts, maps, _) in enumerate(views):
                    sc = Sc_all[gi, v]
                    if mode == 'pyramid' and small:
                        with self._timed("proto_fwd_small"):
                            z, rn, _ = ops.proto_fwd_small(c.Zt[v], picks, maps, Wc, bc, out=sc)
                        Zh.append(z); Rn.append(rn)
                        continue
                    if mode == 'pyramid' and head.linear:
                        with self._timed("gather_rows"):
                            z, rn = ops.gather_rows_normalize(c.Zt[v], picks, maps)  # z_hat (B*N, D)
                    else:
                        if mode == 'pyramid':
                            with self._timed("gather_rows"):
                                z = ops.gather_rows(c.Zt[v], picks, maps)            # rows of the first Linear's output
                        else:
                            with self._timed("gather"):
                                x = ops.gather_taps(ts, picks, c.hlen, maps, interp=self._interp)         # (B*N, hlen)
                            with self._timed("proj_fwd"):
                                z = ops.gemm(x, c.Wp, False, True)                   # (B*N, D)
                            X.append(x)
                        if not head.linear:        # '1-layer' / '2-layer': one BatchNorm batch per Sinkhorn problem
                            with self._timed("proj_head"):
                                z, hc = head.forward(z, ngroups, c.sk_reduce, self.world if c.sk_reduce else 1)
                            Hc.append(hc)
                        rn = ops.row_l2_normalize_(z)                              # z -> z_hat
                    self._scores_gemm(c, z, sc, wc_planes, cmax_pre, (gi, v))
                    Zh.append(z); Rn.append(rn)
                keep.append((X, Zh, Rn, Hc))
            skb = self._sinkhorn_batch(c, Sc_all, G, cmax_pre)
            for gi, pno in enumerate(pnos):
                picks = c.pb.picks[pno]
                X, Zh, Rn, Hc = keep[gi]
                Sc = [Sc_all[gi, 0], Sc_all[gi, 1]]
                if pad_ds:   # rows padded to whole 16-column chunks (zero pad) so that dS can be the A operand of the dz GEMM
                    dSp = [torch.empty(rows, Kp, dtype=torch.float32, device=dev) for _ in range(2)]
                    for t_ in dSp:
                        if Kp > K:
                            t_[:, K:].zero_()
                    dS = [t_[:, :K] for t_ in dSp]
                else:
                    dS = [torch.empty_like(Sc[0]), torch.empty_like(Sc[1])]
                # many prototypes: loss, dS and the bias gradient in one pass over the score matrices
                gb_in_ce = not (mode == 'pyramid' and small) and ops.swapped_ce_fused_ok(skb[0], dS[0])
                for g in range(ngroups):
                    r0, r1 = g * grows, (g + 1) * grows
                    sk = [skb[(gi * 2 + v) * ngroups + g] for v in range(2)]
                    if gb_in_ce:
                        ops.swapped_ce_fused(sk[0], sk[1], c.T, c.gscale, c.loss, c.gscale, 1.0, dS[0][r0:r1], dS[1][r0:r1], gb,
                                             0.0 if (c.first_acc and g == 0) else 1.0)
                    else:
                        ops.swapped_ce(sk[0], sk[1], c.T, c.gscale, c.loss, c.gscale, 1.0, dS[0][r0:r1], dS[1][r0:r1])
                    if c.return_aux and pno == 0 and g == 0:
                        c.aux.update(scores_s=Sc[0][r0:r1].clone(), scores_t=Sc[1][r0:r1].clone(),
                                     q_s=sk[0].codes(), q_t=sk[1].codes())
                # S6: backward (gradients accumulate over views and patches)
                for v in range(2):
                    beta = 0.0 if c.first_acc else 1.0
                    if mode == 'pyramid' and small:
                        with self._timed("proto_bwd_small"):
                            ops.proto_bwd_small_(dS[v], Zh[v], Rn[v], picks, views[v][1], Wc, c.Gt[v], gWc, gb, beta)
                        c.first_acc = False
                        continue
                    ops.gemm(dS[v], Zh[v], True, False, out=gWc, beta=beta, split_k=self._split_k(K, D, rows))
                    if not gb_in_ce:
                        ops.colsum(dS[v], gb, alpha=1.0, beta=beta)
                    if pad_ds and ops.gemm_planes_ok(dSp[v], None, D, Zh[v]):
                        dz = torch.empty(rows, D, dtype=torch.float32, device=dev)
                        ops.gemm_nt_planes(dSp[v], None, wct_planes, dz)
                    else:
                        dz = ops.gemm(dS[v], Wc, False, False)                     # d z_hat (B*N, D)
                    if mode == 'pyramid' and head.linear:
                        with self._timed("scatter_rows"):
                            ops.normalize_bwd_scatter_(dz, Zh[v], Rn[v], picks, views[v][1], c.Gt[v])
                    else:
                        ops.l2_normalize_bwd_(dz, Zh[v], Rn[v])
                        if not head.linear:
                            with self._timed("proj_head"):
                                dz = head.backward(dz, Zh[v], Hc[v], c.gextra, beta, split_k=self._split_k(D, D, rows))
                        if mode == 'pyramid':
                            with self._timed("scatter_rows"):
                                ops.scatter_rows_add_(dz, picks, views[v][1], c.Gt[v])
                        else:
                            with self._timed("proj_dW"):
                                ops.gemm(dz, X[v], True, False, out=c.gWp, beta=beta, split_k=self._split_k(D, c.hlen, rows))
                    c.first_acc = False
                keep[gi] = None
                del X, Zh, Rn, Hc, Sc, dS
            del keep, Sc_all, skb

    # ------------------------------------------------------------------------------------- training loop
    def _schedule(self):
        sa = self.swav_args
        ns = sa['num_samples']
        warm = np.linspace(sa['start_warmup'], sa['base_lr'], ns * sa['warmup_epochs'])
        it = np.arange(ns * (sa['num_epochs'] - sa['warmup_epochs']))
        cos = np.array([sa['final_lr'] + 0.5 * (sa['base_lr'] - sa['final_lr']) *
                        (1 + math.cos(math.pi * t / (sa['num_epochs'] - sa['warmup_epochs']))) for t in it])
        return np.concatenate((warm, cos))

    def draw_batch(self, batch):
        sa = self.swav_args
        hw = self.model.size * self.model.size
        patch = self._patch_mode()
        npick = 0 if (sa.get('pick_source', 'host') == 'device' and not patch) else sa['num_patches']
        return [draw_image(self.model.style_dim, self.model.num_layers, self.perturb_args['n_layers'], hw,
                           npick, transforms=sa.get('transforms', True),
                           layer_no=self.perturb_args['layer_no'],
                           cpu_reference_rng=sa.get('cpu_reference_rng', False),
                           sampling_method='patch' if patch else 'random', patch_size=sa['patch_size'])
                for _ in range(batch)]

    def pretrain(self, input_latent, num_test_samples=5):
        """Self-supervised pretraining loop (swav_clustering.py:205-505): num_epochs x num_samples
        optimizer steps, each over `batch_size` freshly drawn latents; logs ` E:<e> | Loss | T`
        and saves prototypes.pt / projection.pt."""
        sa = self.swav_args
        sd = self.model.style_dim
        if sa.get('add_local_loss', False):
            # swav_clustering.py:420-425: `sum(self.layer_hf_dim[l_s])` sums ONE int of the configs' flat list and raises
            # TypeError in the reference's first step; there is no defined behaviour to reproduce
            raise NotImplementedError("add_local_loss=True raises in the reference itself (sum() of an int, "
                                      "swav_clustering.py:424); every shipped config sets it False")
        if sa.get('plot_test_images', False):
            self.logger.info("plot_test_images: the tensorboard image grids (swav_clustering.py:467-500) are not drawn")
        # reference RNG order at entry: test latents + their randomized-noise forwards (:224-237)
        for _ in range(num_test_samples):
            torch.randn(1, sd)
        for _ in range(num_test_samples):
            for i in range(self.model.num_layers):
                r = 2 ** ((i + 5) // 2)
                torch.empty(1, 1, r, r).normal_()
        self.build_heads()
        if self.world > 1:
            # data parallel: every rank starts from rank 0's heads and then draws its OWN images — the host
            # and device generators are re-seeded per rank (the reference seeds 42 at import,
            # lib/util/util.py:21-24, which would otherwise hand every rank the same z, layers, angles, picks)
            import torch.distributed as dist
            for p in self._params():
                dist.broadcast(p, 0, group=self.dist_group)
            seed = int(sa.get('seed', 42)) + self.rank
            torch.manual_seed(seed)
            np.random.seed(seed)
        self.logger.info("Projection Network:")
        self.logger.info(self.projection.__str__())
        self.logger.info("Prototype Matrix:")
        self.logger.info(self.prototype.__str__())
        self.logger.info("SwAV Parameters:")
        for k, v in sa.items():
            self.logger.info(f"{k}: {v}")
        self.logger.info("Sinkhorn-Knopp Algo. Parameters:")
        for k, v in self.sinkhorn_args.items():
            self.logger.info(f"{k}: {v}")
        self._lr_schedule = self._schedule()
        batch = sa.get('batch_size', 1)
        t0 = time.time()
        self.loss_history = []
        # the next batch is drawn (host RNG, same order) and staged BEFORE a step's loss is read back for the log: the read-back
        # waits for the GPU, and the host-side draws (80 randperms of 65 536 per ffhq batch) would otherwise start only after it
        # ... and (swav_args['overlap_views'], default on) its views are queued on the view stream before the step itself is queued:
        # the frozen generator's work for batch i + 1 runs beside step i (ViewMixin.prefetch_views; same arithmetic, same draws)
        total = sa['num_epochs'] * sa['num_samples']
        overlap = bool(sa.get('overlap_views', True))
        stage = (lambda b: self.prefetch_views(b)) if overlap else (lambda b: b)
        nxt = stage(self.draw_batch(batch)) if total > 0 else None
        done = 0
        for e in range(sa['num_epochs']):
            for _ in range(sa['num_samples']):
                cur, done = nxt, done + 1
                nxt = stage(self.draw_batch(batch)) if done < total else None
                loss = self.train_step(cur)
                self.loss_history.append(loss)
                if self.writer is not None:
                    self.writer.add_scalar('swav/loss', loss.item(), e)
            if e % sa['epoch_print_freq'] == 0:
                lv = float(loss.item())
                ops.check_finite(lv, "SwAV loss")                      # (where the host reads the loss back anyway)
                self.logger.info(f" E:{e}\t|\tLoss: {lv:.03f} \t|\tT: {time.time() - t0:.03f}")
        if self.loss_history:
            ops.check_finite(float(self.loss_history[-1].item()), "SwAV loss")
        self.logger.info("Finished pretraining - Saving projection file")
        if self.prototype_file is not None and self.rank == 0:
            torch.save(self.prototype, self.prototype_file)
            torch.save(self.projection, self.projection_file)

    # ------------------------------------------------------------------------------------- inference
    def predict_swav_codes(self, input_latent, input_is_latent=True):
        """(swav_clustering.py:659-693) generator with fixed noise and ONE truncation ->
        projection of every pixel -> (out_preds (b,D,h,w) fp32, out_labels (b,h,w) int64 =
        argmax over the D projection channels).  Like the reference, the latent is always
        treated as a W latent."""
        lat = input_latent.to(self.device).float()
        feats = self._features_only(lat)
        ts = ops.TapSet([f.permute(0, 2, 3, 1) for f in feats])
        hw = ts.size
        wp = self.projection[0].weight.data
        if self.swav_args.get('projection_mode', 'pyramid') == 'pyramid' and self._interp == 'nearest':
            z = Pyramid(ts, self.swav_args['hlen']).project(wp).reshape(ts.B * hw * hw, -1)
        else:
            picks = torch.arange(hw * hw, dtype=torch.int32, device=self.device).repeat(ts.B, 1)
            x = ops.gather_taps(ts, picks, self.swav_args['hlen'], interp=self._interp)
            z = ops.gemm(x, wp, False, True)                                   # (b*h*w, D) == NHWC
        head = self._proj_head()
        if not head.linear:          # BatchNorm in train mode here too (the reference never calls .eval()): per image
            z, _ = head.forward(z.contiguous(), ts.B, want_ctx=False)
        labels = ops.row_argmax(z).reshape(ts.B, hw, hw)
        preds = z.reshape(ts.B, hw, hw, wp.shape[0]).permute(0, 3, 1, 2)
        return preds, labels
