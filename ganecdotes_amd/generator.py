"""StyleGAN2 generator (feature source of the SwAV path) on hand-written gfx950 kernels.

Same constructor, call signature, `style` / `mean_latent` helpers and state_dict keys as the
reference's `Generator` (models/stylegan2/model.py:457-648), so rosinality `g_ema` checkpoints
load unchanged (src/one_shot_pipeline.py:138-147).  Underneath, nothing is shared with it:
activations are NHWC, the modulated convolutions run as batch-shared-weight implicit GEMMs on
the fp32 MFMA with noise+bias+leaky-ReLU epilogues, and derived weights are cached once for
the frozen network (see csrc/gx_conv.hip).  There is no CPU path: calling it on a non-HIP
device raises RuntimeError.

`forward` returns the 2*log2(size)-3 hidden-feature taps as tensors of logical shape
(B, C, H, W) whose memory is NHWC (`tap.permute(0, 2, 3, 1)` is contiguous) — exactly what
the gather kernel consumes.
"""
import math
import os
import random

import torch
from torch import nn

from . import ops


class _Bag(nn.Module):
    """Parameter container (gives the reference's state_dict key structure)."""


def _equal_linear_params(in_dim, out_dim, bias_init=0.0, lr_mul=1.0):
    m = _Bag()
    m.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
    m.bias = nn.Parameter(torch.full((out_dim,), float(bias_init)))
    m.lr_mul = lr_mul
    return m


MAP_SPLIT_K = 8          # k split of the mapping-MLP GEMMs (tiny M: 15.1 -> 14.9 ms per step of views against unsplit)
SKINNY_ROWS = 256        # up to this many latents the mapping layers run as ops.equal_linear_skinny (one launch per layer)


def _fir(taps):
    k = torch.tensor(taps, dtype=torch.float32)
    k = torch.outer(k, k)
    return k / k.sum()


def _styled_conv_params(cin, cout, ksize, style_dim, upsample, blur_kernel):
    m = _Bag()
    m.conv = _Bag()
    m.conv.weight = nn.Parameter(torch.randn(1, cout, cin, ksize, ksize))
    m.conv.modulation = _equal_linear_params(style_dim, cin, bias_init=1.0)
    if upsample:
        m.conv.blur = _Bag()
        m.conv.blur.register_buffer("kernel", _fir(blur_kernel) * 4.0)
    m.noise = _Bag()
    m.noise.weight = nn.Parameter(torch.zeros(1))
    m.activate = _Bag()
    m.activate.bias = nn.Parameter(torch.zeros(cout))
    m.upsample = upsample
    return m


def _to_rgb_params(cin, style_dim, upsample, blur_kernel):
    m = _Bag()
    if upsample:
        m.upsample = _Bag()
        m.upsample.register_buffer("kernel", _fir(blur_kernel) * 4.0)
    m.conv = _Bag()
    m.conv.weight = nn.Parameter(torch.randn(1, 3, cin, 1, 1))
    m.conv.modulation = _equal_linear_params(style_dim, cin, bias_init=1.0)
    m.bias = nn.Parameter(torch.zeros(1, 3, 1, 1))
    return m


# x-planes conv path (activations split once by their producer, both conv operands by LDS-DMA: csrc/gx_conv.hip
# modconv_xp_kernel) for every layer big enough to fill the chip; GX_CONV_XP=0 keeps the in-kernel split everywhere
CONV_XP = os.environ.get("GX_CONV_XP", "1") != "0"
# per-sample power-of-two normalisation of every styled conv's style, folded into its demod coefficients (gx_style_norm: exact);
# False only in the test that shows what it is for (huge styles against the f16x3 planes' range)
STYLE_NORM = True


class _OneStreamAtATime:
    """Guard of a module's persistent device buffers (the generator's x-plane buffers and derived-weight caches): they are
    reused call after call, so two calls must not be in flight on two streams at once.  A call on another stream than the
    previous one first waits for that call's end; calls on one stream are ordered anyway (no wait, no cost).  This is what lets
    the views of the next batch run on their own stream (ViewMixin.prefetch_views) while predict / plotting calls on the
    default stream stay correct."""

    def __init__(self, owner):
        self.owner = owner

    def __enter__(self):
        cur = torch.cuda.current_stream()
        last = self.owner.__dict__.get("_last_use")
        if last is not None and last[0] != cur.cuda_stream:
            cur.wait_event(last[1])
        return self

    def __exit__(self, *exc):
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(cur)
        self.owner.__dict__["_last_use"] = (cur.cuda_stream, ev)
        return False


class MappingNetwork(nn.Module):
    """`Generator.style`: PixelNorm + n_mlp EqualLinear(fused lrelu), lr_mul 0.01
    (model.py:473-482).  Children are named "1".."n_mlp" like the reference Sequential."""

    def __init__(self, style_dim, n_mlp, lr_mlp):
        super().__init__()
        self.n_mlp = n_mlp
        self.lr_mlp = lr_mlp
        for i in range(n_mlp):
            self.add_module(str(i + 1), _equal_linear_params(style_dim, style_dim, lr_mul=lr_mlp))

    def _transposed(self, i, w):
        """the layer's weight as (in, out), remade when the parameter changes (load_state_dict, .to(), an optimizer step)"""
        cache = self.__dict__.setdefault("_wt_cache", {})
        key = (w.data_ptr(), w._version, w.device)
        hit = cache.get(i)
        if hit is None or hit[0] != key:
            cache[i] = hit = (key, w.t().contiguous())
        return hit[1]

    def forward(self, z):
        if not z.is_cuda:
            raise RuntimeError("ganecdotes_amd.Generator runs on HIP devices only (no CPU fallback)")
        with _OneStreamAtATime(self):               # the transposed-weight cache is made by the first call
            return self._forward(z)

    def _forward(self, z):
        shape = z.shape
        h = ops.pixel_norm(z.reshape(-1, shape[-1]).contiguous().float())
        for i in range(self.n_mlp):
            lin = getattr(self, str(i + 1))
            w = lin.weight.detach()
            if h.shape[0] <= SKINNY_ROWS and w.shape[1] % 16 == 0:
                # a batch of latents: one latency-bound launch per layer instead of {GEMM split over k, slab reduce, bias + act}
                h = ops.equal_linear_skinny(h, self._transposed(i, w), (1.0 / math.sqrt(w.shape[1])) * self.lr_mlp,
                                            lin.bias.detach(), self.lr_mlp, act=True)
                continue
            # few rows x 512 x 512: four output tiles of a 32-k-tile loop are latency-bound; split the k loop over 8 x 4 tiles
            y = ops.gemm(h, w, False, True, split_k=MAP_SPLIT_K if h.shape[0] <= 256 else 1)
            h = ops.bias_act_(y, alpha=(1.0 / math.sqrt(w.shape[1])) * self.lr_mlp, bias=lin.bias.detach(),
                              bscale=self.lr_mlp, act=True)
        return h.reshape(shape)


class Generator(nn.Module):
    def __init__(self, size, style_dim, n_mlp, channel_multiplier=2, blur_kernel=(1, 3, 3, 1), lr_mlp=0.01,
                 channels=None):
        super().__init__()
        self.size = size
        self.style_dim = style_dim
        self.style = MappingNetwork(style_dim, n_mlp, lr_mlp)
        self.channels = channels or {
            4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * channel_multiplier, 128: 128 * channel_multiplier,
            256: 64 * channel_multiplier, 512: 32 * channel_multiplier, 1024: 16 * channel_multiplier}
        ch = self.channels
        self.input = _Bag()
        self.input.input = nn.Parameter(torch.randn(1, ch[4], 4, 4))
        self.conv1 = _styled_conv_params(ch[4], ch[4], 3, style_dim, False, blur_kernel)
        self.to_rgb1 = _to_rgb_params(ch[4], style_dim, False, blur_kernel)
        self.log_size = int(math.log(size, 2))
        self.num_layers = (self.log_size - 2) * 2 + 1
        self.convs = nn.ModuleList()
        self.to_rgbs = nn.ModuleList()
        self.noises = _Bag()
        for layer_idx in range(self.num_layers):
            res = (layer_idx + 5) // 2
            self.noises.register_buffer(f"noise_{layer_idx}", torch.randn(1, 1, 2 ** res, 2 ** res))
        cin = ch[4]
        for i in range(3, self.log_size + 1):
            cout = ch[2 ** i]
            self.convs.append(_styled_conv_params(cin, cout, 3, style_dim, True, blur_kernel))
            self.convs.append(_styled_conv_params(cout, cout, 3, style_dim, False, blur_kernel))
            self.to_rgbs.append(_to_rgb_params(cout, style_dim, True, blur_kernel))
            cin = cout
        self.n_latent = self.log_size * 2 - 2
        self._derived = None
        self._xplanes = {}              # this generator's x-planes buffers, one per styled conv (ops.xplanes_buffer)
        self.want_image = True          # ToRGB branch can be skipped by feature-only callers

    # ------------------------------------------------------------------ derived (frozen) weights
    def _invalidate(self):
        self._derived = None
        self._xplanes = {}
        self.style.__dict__.pop("_wt_cache", None)

    def load_state_dict(self, *a, **k):
        self._invalidate()
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._invalidate()
        return super()._apply(fn, *a, **k)

    def _prepare(self):
        """Weight re-layouts for the kernels, computed once (the generator is frozen on this
        path): conv weights tap-major [k*k][Cout][Cin], sum-of-squares for demodulation,
        flipped FIR kernels, the NHWC constant input."""
        fmt = (ops.get_gemm_mode(), ops.get_conv_planes()) if self.input.input.is_cuda else None
        if self._derived is not None and self._derived["plane_format"] == fmt:
            return self._derived
        d = {"plane_format": fmt}          # the frozen weight planes below are in the conv engine's format of the moment

        def conv(m, key):
            w = m.conv.weight.detach()[0]                                  # (Cout, Cin, k, k)
            cout, cin, k, _ = w.shape
            d[key + ".wt"] = w.permute(2, 3, 0, 1).reshape(k * k, cout, cin).contiguous()
            # the generator is frozen on this path: split the weights into bf16 planes once (bf16x6 engine)
            d[key + ".wtp"] = ops.split_conv_weights(d[key + ".wt"]) if w.is_cuda else None
            d[key + ".wsq"] = w.pow(2).sum(dim=(2, 3)).contiguous()
            d[key + ".scale"] = 1.0 / math.sqrt(cin * k * k)
            if hasattr(m.conv, "blur"):
                d[key + ".kf"] = torch.flip(m.conv.blur.kernel.detach(), [0, 1]).contiguous()

        def rgb(m, key):
            w = m.conv.weight.detach()[0, :, :, 0, 0]
            d[key + ".w3"] = w.contiguous()
            d[key + ".scale"] = 1.0 / math.sqrt(w.shape[1])
            d[key + ".bias"] = m.bias.detach().reshape(3).contiguous()
            if hasattr(m, "upsample"):
                d[key + ".kf"] = torch.flip(m.upsample.kernel.detach(), [0, 1]).contiguous()

        conv(self.conv1, "conv1")
        rgb(self.to_rgb1, "to_rgb1")
        for i, m in enumerate(self.convs):
            conv(m, f"convs.{i}")
        for i, m in enumerate(self.to_rgbs):
            rgb(m, f"to_rgbs.{i}")
        d["const"] = self.input.input.detach().permute(0, 2, 3, 1).contiguous()     # (1,4,4,C)
        # every layer's style affine as ONE GEMM (weights stacked along the output columns), and the
        # per-layer slices / demod inputs as tables for gx_style_pick / gx_demod_batch
        layers = [("conv1", self.conv1, 0, True), ("to_rgb1", self.to_rgb1, 1, False)]
        i = 1
        for blk in range(len(self.to_rgbs)):
            layers.append((f"convs.{2 * blk}", self.convs[2 * blk], i, True))
            layers.append((f"convs.{2 * blk + 1}", self.convs[2 * blk + 1], i + 1, True))
            layers.append((f"to_rgbs.{blk}", self.to_rgbs[blk], i + 2, False))
            i += 2
        dev = self.input.input.device
        ws, bs, pick, dtab, dscale, wsq = [], [], [], [], [], []
        off = doff = woff = 0
        for key, m, idx, styled in layers:
            w = m.conv.modulation.weight.detach()
            cin = w.shape[0]
            ws.append(w); bs.append(m.conv.modulation.bias.detach())
            pick += [idx, off, cin]
            d[key + ".soff"] = (off, cin)
            if styled:
                cout = d[key + ".wsq"].shape[0]
                dtab += [off, cin, cout, woff, doff]
                dscale.append(d[key + ".scale"])
                wsq.append(d[key + ".wsq"].reshape(-1))
                d[key + ".doff"] = (doff, cout)
                doff += cout; woff += cout * cin
            off += cin
        d["mod.W"] = torch.cat(ws).contiguous()
        d["mod.b"] = torch.cat(bs).contiguous()
        d["mod.pick"] = torch.tensor(pick, dtype=torch.int32, device=dev)
        d["mod.nlayer"] = len(layers)
        d["mod.dtab"] = torch.tensor(dtab, dtype=torch.int32, device=dev)
        d["mod.dscale"] = torch.tensor(dscale, dtype=torch.float32, device=dev)
        d["mod.wsq"] = torch.cat(wsq).contiguous()
        d["mod.nconv"] = len(dscale)
        d["mod.drows"] = doff
        self._derived = d
        return d

    # ------------------------------------------------------------------ helpers of the reference API
    def make_noise(self):
        device = self.input.input.device
        noises = [torch.randn(1, 1, 4, 4, device=device)]
        for i in range(3, self.log_size + 1):
            for _ in range(2):
                noises.append(torch.randn(1, 1, 2 ** i, 2 ** i, device=device))
        return noises

    def mean_latent(self, n_latent):
        """mean of style(z) over n_latent draws (model.py:554-560)."""
        z = torch.randn(n_latent, self.style_dim, device=self.input.input.device)
        w = self.style(z)
        out = torch.empty(self.style_dim, dtype=torch.float32, device=w.device)
        ops.colsum(w, out, alpha=1.0 / n_latent)
        return out.reshape(1, self.style_dim)

    def get_latent(self, z):
        return self.style(z)

    # ------------------------------------------------------------------ layers
    def _styles(self, latent, d):
        """(packed styles, packed demod') of every layer for latent (B, n_latent, S): one GEMM of all
        latents through all affines (EqualLinear, model.py:155-169 inside ModulatedConv2d :239), one
        pick, one demod launch."""
        bsz, nl, sd = latent.shape
        s_all = ops.gemm(latent.reshape(bsz * nl, sd), d["mod.W"], False, True, alpha=1.0 / math.sqrt(sd),
                         col_bias=d["mod.b"])
        sp = ops.style_pick(s_all, bsz, nl, d["mod.pick"], d["mod.nlayer"])
        dm = ops.demod_batch(sp, d["mod.wsq"], bsz, d["mod.dtab"], d["mod.dscale"], d["mod.nconv"], d["mod.drows"])
        if STYLE_NORM:
            # every styled conv's style to magnitude < 4 per sample, the power of two folded into its demod coefficients
            # (exact): the conv engines' operand range then only has to hold the activations (the ToRGB styles stay as they are)
            ops.style_norm_(sp, dm, bsz, d["mod.dtab"], d["mod.nconv"])
        return sp, dm

    @staticmethod
    def _slab(packed, bsz, span):
        off, n = span
        return packed[bsz * off:bsz * (off + n)].view(bsz, n)

    def _uses_xp(self, m, key, bsz, h, w, d):
        """does the styled conv `key`, fed a (bsz, h, w, cin) input, take the x-planes path?"""
        cout, cin = d[key + ".wt"].shape[1], d[key + ".wt"].shape[2]
        return CONV_XP and ops.conv_xp_ok(bsz, h, w, cout, cin, d[key + ".wtp"], up=m.upsample)

    def _styled(self, m, key, x, sp, dmp, noise, d, xp=None, nxt=None):
        """One StyledConv.  xp: this layer's pre-split modulated input if its producer already made it;
        nxt = (module, key) of the next styled conv: if that one reads x planes, this layer's output pass also
        writes them (out * next style).  Returns (out, planes for the next conv or None)."""
        bsz, h, w, cin = x.shape
        s = self._slab(sp, bsz, d[key + ".soff"])
        dm = self._slab(dmp, bsz, d[key + ".doff"])
        nw = m.noise.weight.detach()
        b = m.activate.bias.detach()
        wtp = d[key + ".wtp"]
        cout = d[key + ".wt"].shape[1]
        if xp is None and self._uses_xp(m, key, bsz, h, w, d):
            xp = ops.modsplit(x, s, out=ops.xplanes_buffer(bsz, h, w, cin, x.device, key, cache=self._xplanes))
        oh, ow = (2 * h, 2 * w) if m.upsample else (h, w)
        ns = npl = None
        if nxt is not None and self._uses_xp(nxt[0], nxt[1], bsz, oh, ow, d):
            ns = self._slab(sp, bsz, d[nxt[1] + ".soff"])
            npl = ops.xplanes_buffer(bsz, oh, ow, cout, x.device, nxt[1], cache=self._xplanes)
        if m.upsample:
            out = ops.modconv_up(x, d[key + ".wt"], s, dm, d[key + ".kf"], noise, nw, b, True, wt_planes=wtp, x_planes=xp,
                                 next_style=ns, next_planes=npl)
        else:
            out = ops.modconv(x, d[key + ".wt"], 3, s, dm, noise, nw, b, True, wt_planes=wtp, x_planes=xp,
                              next_style=ns, next_planes=npl)
        return out, npl

    def _rgb(self, m, key, x, sp, skip, d):
        s = self._slab(sp, x.shape[0], d[key + ".soff"])
        return ops.torgb(x, d[key + ".w3"], s, d[key + ".scale"], d[key + ".bias"], skip, d.get(key + ".kf"))

    def synthesis(self, latent, noise):
        """latent (B, n_latent, S) -> (image (B,3,size,size) or None, [NHWC taps])."""
        with _OneStreamAtATime(self):
            return self._synthesis(latent, noise)

    def _synthesis(self, latent, noise):
        d = self._prepare()
        bsz = latent.shape[0]
        sp, dmp = self._styles(latent.contiguous(), d)

        def nz(i, res):
            if noise[i] is None:
                return torch.randn(bsz, 1, res, res, device=latent.device)
            return noise[i].contiguous()

        chain = [(self.conv1, "conv1")] + [(m, f"convs.{i}") for i, m in enumerate(self.convs)]
        x = d["const"].expand(bsz, -1, -1, -1).contiguous()
        x, xp = self._styled(self.conv1, "conv1", x, sp, dmp, nz(0, 4), d, None, chain[1] if len(chain) > 1 else None)
        taps = [x]
        skip = self._rgb(self.to_rgb1, "to_rgb1", x, sp, None, d) if self.want_image else None
        for blk in range(len(self.to_rgbs)):
            res = 2 ** (blk + 3)
            i = 2 * blk + 1                                            # index of this block's up-conv in `chain`
            x, xp = self._styled(*chain[i], x, sp, dmp, nz(2 * blk + 1, res), d, xp, chain[i + 1])
            taps.append(x)
            x, xp = self._styled(*chain[i + 1], x, sp, dmp, nz(2 * blk + 2, res), d, xp,
                                 chain[i + 2] if i + 2 < len(chain) else None)
            taps.append(x)
            if self.want_image:
                skip = self._rgb(self.to_rgbs[blk], f"to_rgbs.{blk}", x, sp, skip, d)
        return skip, taps

    # ------------------------------------------------------------------ reference call signature
    @torch.no_grad()
    def forward(self, styles, return_latents=False, inject_index=None, truncation=1, truncation_latent=None,
                input_is_latent=False, noise=None, randomize_noise=True):
        if not self.input.input.is_cuda:
            raise RuntimeError("ganecdotes_amd.Generator runs on HIP devices only (no CPU fallback)")
        styles = [s.float() for s in styles]
        if not input_is_latent:
            styles = [self.style(s) for s in styles]        # 3-D styles are mapped layer-wise too
        if noise is None:
            if randomize_noise:
                noise = [None] * self.num_layers
            else:
                noise = [getattr(self.noises, f"noise_{i}") for i in range(self.num_layers)]
        if truncation < 1:
            base = truncation_latent.reshape(-1).contiguous().float()
            styles = [ops.lerp_rows(s.contiguous(), base, float(truncation)) for s in styles]
        if len(styles) < 2:
            inject_index = self.n_latent
            latent = styles[0] if styles[0].ndim == 3 else styles[0].unsqueeze(1).repeat(1, inject_index, 1)
        else:
            if inject_index is None:
                inject_index = random.randint(1, self.n_latent - 1)
            latent = torch.cat([styles[0].unsqueeze(1).repeat(1, inject_index, 1),
                                styles[1].unsqueeze(1).repeat(1, self.n_latent - inject_index, 1)], 1)
        image, taps = self.synthesis(latent.contiguous(), noise)
        if return_latents:
            return image, latent
        return image, [t.permute(0, 3, 1, 2) for t in taps]
