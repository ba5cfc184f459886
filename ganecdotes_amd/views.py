"""View creation shared by the SwAV and SimCLR preprocessors (reference:
hfc_with_swav/swav_clustering.py:574-656, lib/oneshot/image_augmentor.py:8-104): batch staging
through pinned memory, latent perturbation, the batched NHWC generator forward and the
rotate/flip index maps."""
import logging
import os

import torch

from . import ops

_LOG = logging.getLogger("ganecdotes_amd")


class PreparedBatch:
    """Device-side form of a batch of ImageDraws: staged through pinned host memory and copied
    on a side stream, so drawing / uploading step i+1 overlaps the kernels of step i."""
    __slots__ = ("n", "z", "map_noise", "mix", "xforms", "picks", "ready", "_keep", "views")

    def acquire(self):
        """Make the batch usable on the CURRENT stream: wait for the side-stream upload, and tell the
        caching allocator that this stream reads the side-stream-owned buffers — otherwise their memory
        returns to the side stream's pool the moment Python drops them and the NEXT batch's upload /
        key sort overwrites picks that kernels already queued on this stream have yet to read (a host
        running many steps ahead turned that into out-of-range pixel indices)."""
        cur = torch.cuda.current_stream()
        cur.wait_event(self.ready)
        for t in tuple(self._keep) + ((self.picks,) if self.picks is not None else ()):
            if t is not None and t.is_cuda:
                t.record_stream(cur)


def check_hf_interp(args, where, allowed=("nearest",)):
    """`hf_interp` is the `mode` of the reference's F.interpolate over the taps (hfc_with_swav/swav_clustering.py:123-124,
    baseline/hfc_with_simclr/simclr_clustering.py:102-103).  Every shipped config sets 'nearest'; hfc_with_swav also runs
    'bilinear' (align_corners=False — on the reference-shaped panel route: the pyramid factorisation needs the nearest
    upsampling).  Any other value would change every descriptor, so it is refused by name rather than silently computed as
    nearest."""
    mode = args.get('hf_interp', 'nearest')
    if mode not in allowed:
        raise NotImplementedError(
            f"{where}['hf_interp']={mode!r}: the HIP path implements {' / '.join(repr(a) for a in allowed)}; the feature "
            f"upsampling would silently differ from F.interpolate(mode={mode!r})")
    return mode


class ViewMixin(object):
    """Needs self.model / mean_latent / truncation / perturb_args / device, `self._hlen` and
    `_pick_spec() -> (pixels per patch, patches, 'host'|'device')`."""

    def _features_only(self, lat):
        """The generator's taps for the predict paths (swav_clustering.py:673-677, simclr_clustering.py predict): fixed noise, ONE
        truncation, and no RGB image — the reference discards it there (`perturbed_img` is never read; run_tests renders its
        images with its own generator call, src/one_shot_pipeline.py:618), so the ToRGB branch is not run."""
        prev = getattr(self.model, "want_image", True)
        self.model.want_image = False
        try:
            _, feats = self.model([lat], truncation=self.model_config.truncation, truncation_latent=self.mean_latent,
                                  input_is_latent=True, randomize_noise=False)
        finally:
            self.model.want_image = prev
        # a NaN anywhere in the conv chain reaches the last tap (every later 3x3 conv spreads it): one small reduction here
        # keeps the predict paths from turning an out-of-range activation into silent garbage labels
        ops.check_finite(feats[-1][:, :, ::8, ::8], "generator taps")
        return feats

    def probe_feature_range(self, n=4, margin=4.0):
        """Called once by the preprocessors' constructors.  The f16x3 engine (DESIGN.md par. 4.21) holds fp32-class error inside
        a window of magnitudes: |activation| < 8188 into a conv, taps under ops.FEATURE_RANGE into the projection — and the
        generator is frozen, so whether a CHECKPOINT lives inside that window can be found before the first step: n untruncated
        latents (the widest styles the mapping network gives; a private RNG stream, the draws of a run are untouched) go through it
        once.  A tap that is not finite, or within `margin` of the window's edge, switches the process to the bf16x6 forms (fp32's
        range, ops.set_conv_planes(0) + ops.GEMM_H2 = False) with a warning — instead of the first optimizer step returning NaN.
        Nothing happens on the other engines, or when GX_CONV_PLANES / GX_GEMM_PLANES were set by hand.  -> the largest |tap|."""
        if ops.get_gemm_mode() != 1 or (ops.get_conv_planes() != 1 and not ops.GEMM_H2):
            return None
        if "GX_CONV_PLANES" in os.environ or "GX_GEMM_PLANES" in os.environ:
            return None
        gen = torch.Generator(device=self.device)
        gen.manual_seed(0x5EED)
        z = torch.randn(n, self.model.style_dim, device=self.device, generator=gen)
        prev = getattr(self.model, "want_image", True)
        self.model.want_image = False
        try:
            with torch.no_grad():
                _, feats = self.model([z], randomize_noise=False)
        finally:
            self.model.want_image = prev
        peak = float(torch.stack([f.abs().max() for f in feats]).max())
        limit = ops.FEATURE_RANGE / margin
        if not (peak < limit):                                    # (NaN compares false)
            ops.set_conv_planes(0)
            ops.GEMM_H2 = False
            msg = (f"generator activations reach {peak:.4g} (f16x3 window / {margin:g} = {limit:g}): the convs and the weight-plane "
                   f"GEMMs of this process run their bf16x6 forms (fp32's range) from here on — GX_CONV_PLANES / GX_GEMM_PLANES "
                   f"set by hand override this probe")
            _LOG.warning(msg)
            lg = getattr(self, "logger", None)
            if lg is not None and hasattr(lg, "warning"):
                lg.warning(msg)
        return peak

    def _timed(self, tag, flops=0.0, nbytes=0.0):
        """HIP-event bracket of a phase of the step (active only when ops.PROFILE is a dict)."""
        return ops.record("phase:" + tag, flops, nbytes)

    def create_pixel_feature_vectors(self, features, pred=False):
        """API-compatibility shim of swav_clustering.py:108-130: returns the materialised
        (B, hlen, H, W) stack through the gather kernel with the identity pick list."""
        taps = [f.permute(0, 2, 3, 1).contiguous() for f in features]
        ts = ops.TapSet(taps)
        hw = ts.size * ts.size
        picks = torch.arange(hw, dtype=torch.int32, device=self.device).repeat(ts.B, 1)
        x = ops.gather_taps(ts, picks, self._hlen, interp=getattr(self, "_interp", "nearest"))
        return x.reshape(ts.B, ts.size, ts.size, -1).permute(0, 3, 1, 2)

    def prepare_batch(self, images):
        """Pack a list of ImageDraws into two pinned staging buffers (floats, int32 picks) and
        upload them with non-blocking copies on a side stream."""
        if isinstance(images, PreparedBatch):
            return images
        B, S, nl = len(images), self.model.style_dim, self.model.n_latent
        hw = self.model.size * self.model.size
        N, npatch, psrc = self._pick_spec()
        N = hw if (N is None or N >= hw) else N
        P = len(images[0].picks)
        dev_picks = (P == 0 and psrc == 'device')
        nf = B * S + 2 * (2 * B * S) + 2 * (B * nl * 2)
        if not hasattr(self, "_staging"):
            self._staging, self._stage_i, self._copy_stream = {}, 0, torch.cuda.Stream(device=self.device)
        key = (nf, P * B * N)
        ring = self._staging.setdefault(key, [])
        if len(ring) < 3:
            ring.append([torch.empty(nf, dtype=torch.float32).pin_memory(),
                         torch.empty(max(1, P * B * N), dtype=torch.int32).pin_memory(), None])
            slot = ring[-1]
        else:
            slot = ring[self._stage_i % 3]
            self._stage_i += 1
            if slot[2] is not None:
                slot[2].synchronize()              # its previous upload has been consumed
        hf, hi, _ = slot
        o = 0
        hf[o:o + B * S].view(B, S).copy_(torch.cat([im.z for im in images], 0)); o += B * S
        for v in range(2):
            hf[o:o + 2 * B * S].view(2 * B, S).copy_(torch.cat([im.views[v].map_noise for im in images], 0))
            o += 2 * B * S
        for v in range(2):
            mix = hf[o:o + B * nl * 2].view(B, nl, 2)
            mix.zero_()
            for b, im in enumerate(images):
                vd = im.views[v]
                sig = float(self.perturb_args['perturb_std'][vd.layer_no])
                mix[b, 2 * vd.layer_no, 0] = sig
                mix[b, 2 * vd.layer_no + 1, 1] = sig
            o += B * nl * 2
        if P:
            hi.view(P, B, N).copy_(torch.stack([torch.stack([im.picks[p][:N] for im in images]) for p in range(P)]))
        pb = PreparedBatch()
        pb.n = B
        pb.views = None
        with torch.cuda.stream(self._copy_stream):
            df = hf.to(self.device, non_blocking=True)
            di = hi.to(self.device, non_blocking=True)
            pb.ready = torch.cuda.Event()
            pb.ready.record()
        slot[2] = pb.ready
        o = 0
        pb.z = df[o:o + B * S].view(B, S); o += B * S
        pb.map_noise = []
        for v in range(2):
            pb.map_noise.append(df[o:o + 2 * B * S].view(2 * B, S)); o += 2 * B * S
        pb.mix = []
        for v in range(2):
            pb.mix.append(df[o:o + B * nl * 2].view(B, nl, 2)); o += B * nl * 2
        pb.picks = di[:P * B * N].view(P, B, N) if P else None
        if dev_picks:
            # picks = the pixels the first N entries of a uniform random permutation select, per image and
            # patch: the N smallest of H*W uniform keys drawn on the device generator (radix select)
            with torch.cuda.stream(self._copy_stream):
                keys = torch.rand(npatch * B, hw, device=self.device)
                pb.picks = ops.random_subset_picks(keys, N).reshape(npatch, B, N)
                pb.ready = torch.cuda.Event()
                pb.ready.record()
            slot[2] = pb.ready
        pb.xforms = [[(im.views[v].angle, im.views[v].flip) for im in images] for v in range(2)]
        pb._keep = (df, di)
        return pb

    def _view_latents(self, w, map_noise, mix, nw=None):
        """Latents of the feature-producing forward for a batch of images and ONE view each
        (swav_clustering.py:599-640, image_augmentor.py:42-53).  w (B,S) mapped latents;
        map_noise (2B,S) (or nw = its mapping, if the caller already has it); mix (B, n_latent, 2) =
        sigma at the two perturbed latents:  lat[b, 2l+j] = (1-sigma) lat + sigma style(noise_j)."""
        if nw is None:
            nw = self.model.style(map_noise)
        return ops.view_latents(w.contiguous(), self.mean_latent.reshape(-1).contiguous(), float(self.truncation),
                                nw.reshape(w.shape[0], 2, -1).contiguous(), mix.contiguous())

    def prefetch_views(self, images):
        """Queue the views of a batch (training form: all 2B images as one batch, no RGB) on the VIEW stream and attach them to its
        PreparedBatch; `train_step` picks them up (take_views).  The generator is frozen, so the views of batch i + 1 depend on
        nothing step i computes: queued before step i they run beside it — the step's HBM- and latency-bound passes (blur,
        Sinkhorn, CE, scatter, the mapping network) and the tails of its matrix-core kernels share the chip with the other
        stream's work instead of leaving it idle.  At a pipeline start the view stream waits for everything
        queued on the current stream so far; in the steady pipeline it waits for nothing (see below), and the host's lead is
        bounded to two steps."""
        pb = self.prepare_batch(images)
        if getattr(pb, "views", None) is not None:
            return pb
        if not hasattr(self, "_view_stream"):
            self._view_stream = torch.cuda.Stream(device=self.device)     # (stream priority either way: no effect)
        # Bounded lead.  The taps live in the VIEW stream's memory pool and are recorded on the step's stream, so a freed set
        # returns to the pool only once the step that read it has run on the GPU: a host that queues many steps ahead (it can:
        # a step is ~125 launches) would pin one 4 GB set of taps per step of lead.  Two steps of lead keep both streams fed.
        done = getattr(self, "_step_done", None) or ()
        while len(done) > 2:
            done.popleft().synchronize()
        while done and done[0].query():            # steps that have already run
            done.popleft()
        vs = self._view_stream
        if not done:
            # pipeline start (or restart after a gap): everything queued so far — the mean latent, a head just built, a predict
            # call — is ordered before the first views.  In the steady pipeline the views depend on nothing the step's stream
            # does (frozen generator, draws staged on the copy stream), and the generator orders its own calls across streams
            # (generator._OneStreamAtATime): no wait, so the views of batch i + 2 may start before step i has ended.
            vs.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(vs):
            with self._timed("views"):
                out = self.make_views(pb, want_all=True, want_image=False)
            norms = None
            if getattr(self, "sinkhorn_args", {}).get("source_pdf") == "image":
                # the norm maps of the 'image' source pdf (swav_clustering.py:361-362) depend on the taps alone: an HBM-bound
                # pass over every tap that belongs beside the step, not in front of it
                norms = [ops.norm_map(ts, self._hlen, maps, interp=getattr(self, "_interp", "nearest")) for ts, maps, _ in out[0]]
            ev = torch.cuda.Event()
            ev.record()
        self._last_view_event = ev
        pb.views = (out, ev, norms)
        return pb

    def take_views(self, pb):
        """(views, TapSet of all 2B images, norm maps or None) of a batch: the prefetched ones (the current stream waits for them and is recorded
        as a user of their memory, which belongs to the view stream's pool) or, without a prefetch, made here and now"""
        pre = getattr(pb, "views", None)
        if pre is None:
            ev = getattr(self, "_last_view_event", None)
            if ev is not None:                      # the generator's plane buffers: one user at a time
                torch.cuda.current_stream(self.device).wait_event(ev)
            with self._timed("views"):
                return (*self.make_views(pb, want_all=True, want_image=False), None)
        (out, ts_all), ev, norms = pre
        pb.views = None
        if not hasattr(self, "_step_done"):
            import collections
            self._step_done = collections.deque()
        self._prefetched_step = True               # _step_epilogue marks the end of this step for prefetch_views' bounded lead
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        pb.acquire()
        for t in ts_all.taps:
            t.record_stream(cur)
        for _, maps, _ in out:
            if maps is not None:
                maps.record_stream(cur)
        for t in norms or ():
            t.record_stream(cur)
        return out, ts_all, norms

    def make_views(self, images, want_all=False, want_image=True):
        """Both views of a batch (list of ImageDraws or PreparedBatch) ->
        per view (TapSet, maps or None, image); want_all: also the TapSet of the whole 2B-image batch
        (view s = images [0, B), view t = [B, 2B)).  want_image=False skips the ToRGB branch: the RGB image of a
        view enters no loss term — the reference keeps it for plots only (swav_clustering.py:347-356, :474-499) — so
        the training step does not compute it (image = None)."""
        pb = self.prepare_batch(images)
        pb.acquire()
        n = pb.n
        # the latents and both views' perturbation noises go through the mapping network together
        allw = self.model.style(torch.cat([pb.z, pb.map_noise[0], pb.map_noise[1]]))
        w = allw[:n]
        # both views go through the generator as ONE batch of 2B images (the reference runs them one after
        # the other, swav_clustering.py:340-345: same arithmetic per image, half the launches, and the
        # low-resolution layers get enough tiles to fill the chip)
        lat = torch.cat([self._view_latents(w, None, pb.mix[v], nw=allw[n + 2 * n * v:n + 2 * n * (v + 1)])
                         for v in range(2)])
        prev = getattr(self.model, "want_image", True)
        self.model.want_image = bool(want_image)
        try:
            img, feats = self.model([lat], truncation=self.truncation, truncation_latent=self.mean_latent,
                                    input_is_latent=True, randomize_noise=False)
        finally:
            self.model.want_image = prev
        out = []
        for v in range(2):
            ts = ops.TapSet([f[v * n:(v + 1) * n].permute(0, 2, 3, 1) for f in feats])
            maps = None
            if any(a is not None for a, _ in pb.xforms[v]):
                hw = ts.size
                maps = torch.empty(pb.n, hw * hw, dtype=torch.int32, device=self.device)
                ops.rotflip_map_batch([a for a, _ in pb.xforms[v]], [f for _, f in pb.xforms[v]], hw, hw, maps)
            out.append((ts, maps, None if img is None else img[v * n:(v + 1) * n]))
        if want_all:
            return out, ops.TapSet([f.permute(0, 2, 3, 1) for f in feats])
        return out
