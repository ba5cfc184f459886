"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU fp32 restatement (plain PyTorch ops, no custom kernels) of the StyleGAN2
generator forward that the per-pixel SwAV path taps for hidden features.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
this package; `ganecdotes_amd/` never does.

Written from the maths of the reference, as pure functions over a state_dict;
each function cites the reference lines it follows (paths relative to the
upstream repo avm-debatr/ganecdotes).  Pinned by tests/golden/gv1..gv4 (arrays
captured from the reference's own functions by tools/capture_golden.py).

Layout here is the reference's: activations NCHW, taps returned as a list of
(B, C_l, r_l, r_l) tensors.
"""
import math

import torch
import torch.nn.functional as F

SQRT2 = math.sqrt(2.0)


def lrelu_bias_scale(x, bias=None):
    """fused_leaky_relu: lrelu(x + b, 0.2) * sqrt(2)   (models/stylegan2/model.py:32-43;
    native spec lib/gan/optim/fused_bias_act_kernel.cu:18-85, act=lrelu, grad=0)."""
    if bias is not None:
        shape = [1, -1] + [1] * (x.ndim - 2)
        x = x + bias.reshape(shape)
    return F.leaky_relu(x, 0.2) * SQRT2


def pixel_norm(x):
    """PixelNorm (model.py:105-110)."""
    return x * torch.rsqrt(x.pow(2).mean(dim=1, keepdim=True) + 1e-8)


def equal_linear(x, weight, bias, lr_mul=1.0, activation=False):
    """EqualLinear (model.py:223-252): W*(lr_mul/sqrt(in)), b*lr_mul, optional fused lrelu."""
    scale = (1.0 / math.sqrt(weight.shape[1])) * lr_mul
    if activation:
        return lrelu_bias_scale(F.linear(x, weight * scale), bias * lr_mul)
    return F.linear(x, weight * scale, bias * lr_mul)


def n_mlp_of(sd):
    n = 0
    while f"style.{n + 1}.weight" in sd:
        n += 1
    return n


def mapping(z, sd, lr_mlp=0.01):
    """Generator.style = PixelNorm + n_mlp x EqualLinear(fused_lrelu, lr_mul=0.01)
    (model.py:473-482)."""
    h = pixel_norm(z)
    for i in range(1, n_mlp_of(sd) + 1):
        h = equal_linear(h, sd[f"style.{i}.weight"], sd[f"style.{i}.bias"], lr_mlp, True)
    return h


def fir_kernel(taps=(1.0, 3.0, 3.0, 1.0)):
    """make_kernel (model.py:113-121): separable outer product, sum-normalised."""
    k = torch.tensor(taps, dtype=torch.float32)
    k = torch.outer(k, k)
    return k / k.sum()


def upfirdn2d(x, kernel, up=1, pad=(0, 0)):
    """upfirdn2d_native with down=1 (model.py:61-102; native spec
    lib/gan/optim/upfirdn2d_kernel.cu:114-215): zero-insert upsample by `up`, pad
    (pad0 before, pad1 after), correlate with the flipped FIR."""
    b, c, h, w = x.shape
    if up > 1:
        z = x.new_zeros(b, c, h * up, w * up)
        z[:, :, ::up, ::up] = x
        x = z
    p0, p1 = pad
    x = F.pad(x, [p0, p1, p0, p1])
    kh, kw = kernel.shape
    wgt = torch.flip(kernel, [0, 1]).reshape(1, 1, kh, kw)
    y = F.conv2d(x.reshape(b * c, 1, x.shape[2], x.shape[3]), wgt)
    return y.reshape(b, c, y.shape[2], y.shape[3])


def upfirdn2d_general(x, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """The native op's own signature (lib/gan/optim/upfirdn2d.cpp:18-35; arithmetic of
    upfirdn2d_kernel.cu:53-112 == upfirdn2d_native, model.py:61-102) on x [major][in_h][in_w][minor]:
    zero-insert upsample, pad (negative pad = crop), correlate with the flipped kernel, decimate."""
    major, ih, iw, minor = x.shape
    kh, kw = kernel.shape
    u = x.new_zeros(major, ih * up_y, iw * up_x, minor)
    u[:, ::up_y, ::up_x] = x
    u = F.pad(u, [0, 0, max(pad_x0, 0), max(pad_x1, 0), max(pad_y0, 0), max(pad_y1, 0)])
    u = u[:, max(-pad_y0, 0):u.shape[1] - max(-pad_y1, 0), max(-pad_x0, 0):u.shape[2] - max(-pad_x1, 0)]
    img = u.permute(0, 3, 1, 2).reshape(major * minor, 1, u.shape[1], u.shape[2])
    y = F.conv2d(img, torch.flip(kernel, [0, 1]).reshape(1, 1, kh, kw))
    y = y.reshape(major, minor, y.shape[2], y.shape[3])[:, :, ::down_y, ::down_x]
    return y.permute(0, 2, 3, 1).contiguous()


def fused_bias_act(x, bias=None, refer=None, act=3, grad=0, alpha=0.2, scale=1.0):
    """fused_bias_act_kernel.cu:18-85 with the op's own argument list (fused_bias_act.cpp:18-31): bias over
    dim 1; act 1 linear / 3 leaky-ReLU; grad 0 forward, 1 first gradient (gated by sign of refer), 2 -> 0."""
    if bias is not None:
        x = x + bias.reshape([1, -1] + [1] * (x.ndim - 2))
    if grad == 2:
        return torch.zeros_like(x)
    if act == 1:
        return x * scale
    gate = x if grad == 0 else refer
    return torch.where(gate > 0, x, x * alpha) * scale


def modulated_conv(x, style, weight, mod_w, mod_b, demodulate=True, upsample=False,
                   blur_kernel=None):
    """ModulatedConv2d.forward (model.py:327-368).

    x (B,Cin,H,W); style (B,S); weight (1,Cout,Cin,k,k).  Per-sample weights
    w' = W * s / sqrt(Cin k^2), demodulated by rsqrt(sum w'^2 + 1e-8); plain: conv
    pad k//2; upsample: conv_transpose stride 2 pad 0, then blur with pad (1,1)."""
    b, cin, h, w = x.shape
    _, cout, _, k, _ = weight.shape
    s = equal_linear(style, mod_w, mod_b)                       # (B,Cin), bias_init=1
    wgt = weight * (1.0 / math.sqrt(cin * k * k)) * s.reshape(b, 1, cin, 1, 1)
    if demodulate:
        d = torch.rsqrt(wgt.pow(2).sum(dim=(2, 3, 4)) + 1e-8)
        wgt = wgt * d.reshape(b, cout, 1, 1, 1)
    outs = []
    for i in range(b):                                          # groups=batch, one sample at a time
        if upsample:
            y = F.conv_transpose2d(x[i:i + 1], wgt[i].transpose(0, 1), stride=2, padding=0)
        else:
            y = F.conv2d(x[i:i + 1], wgt[i], padding=k // 2)
        outs.append(y)
    y = torch.cat(outs, 0)
    if upsample:
        # Blur(pad=(pad0,pad1), upsample_factor=2): p = (4-2)-(k-1) = 0 -> pad (1,1); kernel*4
        y = upfirdn2d(y, blur_kernel, up=1, pad=(1, 1))
    return y


def styled_conv(x, style, sd, prefix, noise, upsample=False):
    """StyledConv.forward (model.py:426-432) = modulated conv -> + noise_w*noise
    (NoiseInjection, :377-382) -> lrelu(x+b)*sqrt2."""
    blur = sd.get(prefix + "conv.blur.kernel") if upsample else None
    y = modulated_conv(x, style, sd[prefix + "conv.weight"], sd[prefix + "conv.modulation.weight"],
                       sd[prefix + "conv.modulation.bias"], True, upsample, blur)
    y = y + sd[prefix + "noise.weight"] * noise
    return lrelu_bias_scale(y, sd[prefix + "activate.bias"])


def to_rgb(x, style, sd, prefix, skip=None):
    """ToRGB.forward (model.py:445-454): 1x1 modulated conv (no demod) + bias; skip is
    FIR-upsampled x2 (Upsample: kernel*4, pad (2,1), model.py:124-142) and added."""
    y = modulated_conv(x, style, sd[prefix + "conv.weight"], sd[prefix + "conv.modulation.weight"],
                       sd[prefix + "conv.modulation.bias"], demodulate=False)
    y = y + sd[prefix + "bias"]
    if skip is not None:
        y = y + upfirdn2d(skip, sd[prefix + "upsample.kernel"], up=2, pad=(2, 1))
    return y


def num_layers_of(sd):
    n = 0
    while f"noises.noise_{n}" in sd:
        n += 1
    return n


def truncate(w, mean_latent, psi):
    """truncation branch of Generator.forward (model.py:594-602)."""
    return mean_latent + psi * (w - mean_latent)


def synthesis(latent, sd, noises=None):
    """Generator.forward from `out = self.input(latent)` on (model.py:622-648).

    latent (B, n_latent, S) already truncated; noises: list of num_layers tensors
    (1|B,1,r,r) or None for the registered `noises.noise_i` buffers
    (randomize_noise=False).  Returns image (B,3,size,size) and the 13 (2*log2-3)
    taps in NCHW."""
    nl = num_layers_of(sd)
    if noises is None:
        noises = [sd[f"noises.noise_{i}"] for i in range(nl)]
    b = latent.shape[0]
    x = sd["input.input"].repeat(b, 1, 1, 1)
    x = styled_conv(x, latent[:, 0], sd, "conv1.", noises[0])
    taps = [x]
    skip = to_rgb(x, latent[:, 1], sd, "to_rgb1.")
    i = 1
    for blk in range((nl - 1) // 2):
        x = styled_conv(x, latent[:, i], sd, f"convs.{2 * blk}.", noises[2 * blk + 1], upsample=True)
        taps.append(x)
        x = styled_conv(x, latent[:, i + 1], sd, f"convs.{2 * blk + 1}.", noises[2 * blk + 2])
        taps.append(x)
        skip = to_rgb(x, latent[:, i + 2], sd, f"to_rgbs.{blk}.", skip)
        i += 2
    return skip, taps


def generator_forward(styles, sd, truncation=1.0, truncation_latent=None, input_is_latent=False,
                      noises=None):
    """Generator.forward for a single style tensor (model.py:565-648): optional mapping,
    truncation whenever psi<1 (regardless of input_is_latent), broadcast to
    n_latent = 2*log2(size)-2 layers unless already 3-D."""
    w = styles
    if not input_is_latent:
        w = mapping(w, sd)
    if truncation < 1:
        w = truncate(w, truncation_latent, truncation)
    n_latent = num_layers_of(sd) + 1
    latent = w if w.ndim == 3 else w.unsqueeze(1).repeat(1, n_latent, 1)
    img, taps = synthesis(latent, sd, noises)
    return img, taps, latent


def mean_latent(sd, n, style_dim, generator=None):
    """Generator.mean_latent (model.py:554-560)."""
    z = torch.randn(n, style_dim, generator=generator)
    return mapping(z, sd).mean(0, keepdim=True)
