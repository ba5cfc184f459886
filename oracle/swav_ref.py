"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU fp32 restatement (plain PyTorch ops) of the reference's per-pixel SwAV
clustering hot path (hfc_with_swav/swav_clustering.py + lib/oneshot/
image_augmentor.py of avm-debatr/ganecdotes).  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.

It keeps the reference's op sequence on purpose (materialised upsample+concat,
full-permutation pixel sampling, unfused Sinkhorn on the K x B matrix), because it
doubles as the "reference CPU path" timed by bench.py.  Pinned by
tests/golden/gv5..gv11.  Two third-party pieces of arithmetic are NOT in the
reference tree and are restated from their public definitions — PARITY UNPINNED
for those two only:
  * torchvision RandomRotation(10)+RandomHorizontalFlip(0.5) on a tensor  (S1b)
  * NVIDIA apex LARC(trust_coefficient, clip=False)                         (S8)
"""
import math
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import stylegan2_ref as G


# ----------------------------------------------------------------------------- S1
def pixel_features(taps, hlen, mode="nearest"):
    """create_pixel_feature_vectors (swav_clustering.py:108-130): nearest-upsample every
    tap to the largest H x W, concat on channels, keep the first `hlen` channels."""
    h = max(t.shape[-2] for t in taps)
    w = max(t.shape[-1] for t in taps)
    return torch.cat([F.interpolate(t, (h, w), mode=mode) for t in taps], dim=1)[:, :hlen]


# ----------------------------------------------------------------------------- S1b
def rotflip_index_map(angle_deg, flip, h, w):
    """Source-pixel map of torchvision `RandomRotation` (nearest, no expand, centre,
    zero fill) followed by `RandomHorizontalFlip` for one drawn (angle, flip)
    (call sites swav_clustering.py:98-102, :358-359).  RESTATED from torchvision's
    public tensor path (inverse affine matrix for -angle about the image centre,
    pixel-centre base grid, grid_sample nearest / zeros / align_corners=False);
    parity unpinned.  Returns int64 (h*w): flat source index, or -1 where the
    rotated image is zero-filled."""
    rot = math.radians(-angle_deg)
    # inverse affine of a pure rotation: [cos, sin, 0; -sin, cos, 0] (rot = -angle)
    theta = torch.tensor([[math.cos(rot), math.sin(rot), 0.0],
                          [-math.sin(rot), math.cos(rot), 0.0]], dtype=torch.float32).reshape(1, 2, 3)
    base = torch.empty(1, h, w, 3, dtype=torch.float32)
    base[..., 0].copy_(torch.linspace(-w * 0.5 + 0.5, w * 0.5 + 0.5 - 1, steps=w))
    base[..., 1].copy_(torch.linspace(-h * 0.5 + 0.5, h * 0.5 + 0.5 - 1, steps=h).unsqueeze(-1))
    base[..., 2].fill_(1)
    rescaled = theta.transpose(1, 2) / torch.tensor([0.5 * w, 0.5 * h], dtype=torch.float32)
    grid = base.view(1, h * w, 3).bmm(rescaled).view(1, h, w, 2)
    ids = (torch.arange(h * w, dtype=torch.float32) + 1).reshape(1, 1, h, w)
    src = F.grid_sample(ids, grid, mode="nearest", padding_mode="zeros", align_corners=False)
    src = src.reshape(h, w).round().long() - 1
    if flip:
        src = torch.flip(src, dims=[1])
    return src.reshape(-1)


def apply_index_map(stack, index_map):
    """stack (1,C,H,W) resampled through a flat source map (-1 -> 0)."""
    b, c, h, w = stack.shape
    flat = stack.reshape(b, c, h * w)
    safe = index_map.clamp(min=0)
    out = flat[:, :, safe] * (index_map >= 0).to(flat.dtype)
    return out.reshape(b, c, h, w)


# ----------------------------------------------------------------------------- S1c
def norm_map(stack):
    """torch.norm(stack, p=2, dim=1) (swav_clustering.py:361-362)."""
    return torch.norm(stack, p=2, dim=1)


# ----------------------------------------------------------------------------- S2 + S3
def sample_pixels(stack, picks, patch_size):
    """'random' sampling (swav_clustering.py:160-167): permute all columns by `picks`,
    keep the first patch_size; 'patch' sampling (:150-158): `picks` is ONE integer offset (np.random.choice(h -
    patch_size), :383-385) and the pixels are the square crop [o:o+ps, o:o+ps] in row-major order.
    Returns (N, C)."""
    if isinstance(picks, (int, np.integer)):
        stack = stack[:, :, picks:picks + patch_size, picks:picks + patch_size]
        picks = None
    flat = torch.squeeze(stack).flatten(1)
    if picks is not None:
        flat = flat[:, picks][:, :patch_size]
    return flat.t()


@dataclass
class HeadParams:
    """the projection network's layers after the first Linear (swav_clustering.py:250-269):
    '1-layer' LeakyReLU(0.01); '2-layer' BatchNorm1d -> LeakyReLU(0.01) -> Linear(D, D) -> BatchNorm1d -> Tanh, both
    BatchNorms in train mode (the reference never calls .eval()); `running` = [mean1, var1, mean2, var2, batches]
    follows nn.BatchNorm1d's bookkeeping (momentum 0.1, unbiased variance)."""
    kind: str
    bn1_w: torch.Tensor = None
    bn1_b: torch.Tensor = None
    w2: torch.Tensor = None
    bn2_w: torch.Tensor = None
    bn2_b: torch.Tensor = None
    running: list = None

    def leaves(self):
        return [] if self.kind != "2-layer" else [self.bn1_w, self.bn1_b, self.w2, self.bn2_w, self.bn2_b]

    @staticmethod
    def fresh(kind, d):
        """as the reference constructs it: nn.Linear's default init draw for the second Linear, BatchNorm at (1, 0)"""
        if kind != "2-layer":
            return HeadParams(kind)
        w2, _ = init_linear_like_torch(d, d, False)
        return HeadParams(kind, torch.ones(d), torch.zeros(d), w2, torch.ones(d), torch.zeros(d),
                          [torch.zeros(d), torch.ones(d), torch.zeros(d), torch.ones(d), 0])

    def apply(self, z):
        if self.kind == "1-layer":
            return F.leaky_relu(z, 0.01)
        if self.running is None:
            self.running = [torch.zeros(z.shape[1]), torch.ones(z.shape[1]), torch.zeros(z.shape[1]),
                            torch.ones(z.shape[1]), 0]
        rm1, rv1, rm2, rv2, _ = self.running
        z = F.leaky_relu(F.batch_norm(z, rm1, rv1, self.bn1_w, self.bn1_b, training=True, momentum=0.1, eps=1e-5), 0.01)
        z = F.batch_norm(F.linear(z, self.w2), rm2, rv2, self.bn2_w, self.bn2_b, training=True, momentum=0.1, eps=1e-5)
        self.running[4] += 1
        return torch.tanh(z)


def codes(x, proj_w, proto_w=None, proto_b=None, train=True, head=None):
    """get_swav_codes_from_hidden_features after sampling (swav_clustering.py:171-177):
    z = projection(x) (x Wp^T, then the head's layers if projn_nw is not 'linear'); train:
    scores = normalize(z) Wc^T + b; predict: z^T (D, N)."""
    z = F.linear(x, proj_w)
    if head is not None and head.kind != "linear":
        z = head.apply(z)
    if train:
        return F.linear(F.normalize(z, p=2, dim=1), proto_w, proto_b)
    return z.t()


# ----------------------------------------------------------------------------- S4
def image_marginals(img, k, b):
    """'image' source pdf (swav_clustering.py:523-532): histc over the whole norm map
    (range = data min/max) with B and K bins, +1e-9, bin0 := bin1, normalised."""
    hb = torch.histc(img, b) + 1e-9
    hb[0] = hb[1]
    hb = hb / hb.sum()
    hk = torch.histc(img, k) + 1e-9
    hk[0] = hk[1]
    hk = hk / hk.sum()
    return hk, hb


def sinkhorn_knopp(scores, eps, niters, source_pdf="uniform", img=None, fp64=False):
    """sinkhorn_knopp (swav_clustering.py:509-544), literally: Q = exp(scores/eps)^T (K x B),
    Q /= sum(Q); niters x {row-normalise to r, column-normalise to c}; return column-
    normalised Q^T (B x K).  `fp64=True` runs the SAME literal algorithm in double
    precision (result cast back to fp32): the checker for inputs where the reference's
    un-shifted fp32 exp() overflows (scores/eps > 88.7; fp64 holds up to 709)."""
    s = scores.double() if fp64 else scores
    q = torch.exp(s / eps).t()
    q = q / q.sum()
    k, b = q.shape
    if source_pdf == "image":
        r, c = image_marginals(img, k, b)
    else:
        r, c = torch.ones(k) / k, torch.ones(b) / b
    r, c = r.to(q.dtype), c.to(q.dtype)
    for _ in range(niters):
        u = q.sum(dim=1)
        q = q * (r / u).unsqueeze(1)
        q = q * (c / q.sum(dim=0)).unsqueeze(0)
    return (q / q.sum(dim=0, keepdim=True)).t().to(scores.dtype)


# ----------------------------------------------------------------------------- S5
def swapped_prediction_loss(p_s, p_t, q_s, q_t):
    """calculate_swapped_prediction_loss (swav_clustering.py:547-570)."""
    lst = (q_s * F.log_softmax(p_t, dim=1)).sum(dim=1).mean()
    lts = (q_t * F.log_softmax(p_s, dim=1)).sum(dim=1).mean()
    return -0.5 * (lst + lts)


# ----------------------------------------------------------------------------- S7, S8
def normalize_prototypes(proto_w):
    """prototype row renormalisation (swav_clustering.py:328-331); bias untouched."""
    return F.normalize(proto_w, dim=1, p=2)


def larc_sgd_step(params, grads, bufs, lr, momentum, trust, use_larc=True, eps=1e-8):
    """apex LARC(clip=False, weight_decay=0) around torch SGD(momentum) (call site
    swav_clustering.py:286-292) — RESTATED from apex's public LARC.step, parity
    unpinned: g *= trust*|p|/(|g|+eps) when both norms are non-zero; then
    buf = g (first step) or momentum*buf + g; p -= lr*buf.  In place; returns bufs."""
    out = []
    for p, g, buf in zip(params, grads, bufs):
        g = g.clone()
        if use_larc:
            pn, gn = torch.norm(p), torch.norm(g)
            if pn != 0 and gn != 0:
                g = g * (trust * pn / (gn + eps))
        buf = g.clone() if buf is None else buf * momentum + g
        p.add_(buf, alpha=-lr)
        out.append(buf)
    return out


# ----------------------------------------------------------------------------- draws
@dataclass
class ViewDraws:
    layer_no: int
    map_noise: torch.Tensor          # (2, S): z-noise fed to the mapping MLP for latents 2l, 2l+1
    angle: Optional[float] = None    # degrees; None = rotation/flip disabled
    flip: bool = False
    index_map: Optional[torch.Tensor] = None   # explicit source map (overrides angle/flip) for parity tests


@dataclass
class ImageDraws:
    z: torch.Tensor                  # (1, S)
    views: List[ViewDraws] = field(default_factory=list)   # [s, t]
    picks: List[torch.Tensor] = field(default_factory=list)  # num_patches x randperm(H*W)


def draw_image_reference_order(style_dim, gen_num_layers, n_perturb_layers, hw, num_patches,
                               transforms=True, degrees=10.0):
    """Consume torch / numpy GLOBAL RNG exactly as one reference training image does
    (SURVEY appendix 16): z (swav_clustering.py:323); per view: gen_num_layers noise
    normal_() draws of the discarded first forward (model.py:380), np.random.choice
    (:611), 2*n_perturb_layers randn_like (image_augmentor.py:47); then angle/flip for s
    and t (:358-359); then one randperm per patch (:388)."""
    z = torch.randn(1, style_dim)
    views = []
    for _ in range(2):
        for i in range(gen_num_layers):
            r = 2 ** ((i + 5) // 2)
            torch.empty(1, 1, r, r).normal_()
        lno = int(np.random.choice(list(range(n_perturb_layers))))
        maps = [torch.randn(1, style_dim) for _ in range(2 * n_perturb_layers)]
        views.append(ViewDraws(lno, torch.cat([maps[2 * lno], maps[2 * lno + 1]], 0)))
    if transforms:
        for v in views:
            v.angle = float(torch.empty(1).uniform_(-degrees, degrees).item())
            v.flip = bool(torch.rand(1) < 0.5)
    picks = [torch.randperm(hw) for _ in range(num_patches)]
    return ImageDraws(z, views, picks)


# ----------------------------------------------------------------------------- V1-V3
def view_latents(w, mean_latent, psi, n_latent, vd: ViewDraws, sd, perturb_std=1.0):
    """Latents fed to the feature-producing forward of one view
    (create_hidden_features_from_perturbed_vectors, swav_clustering.py:574-651;
    create_perturbed_vectors_from_latents, image_augmentor.py:8-55): first forward
    truncates w once and broadcasts; latents 2l,2l+1 := (1-sigma)*lat + sigma*style(noise)."""
    lat = G.truncate(w, mean_latent, psi).unsqueeze(1).repeat(1, n_latent, 1)
    noise_w = G.mapping(vd.map_noise, sd)                       # (2, S)
    new = lat.clone()
    for j in range(2):
        n = 2 * vd.layer_no + j
        new[:, n] = (1 - perturb_std) * lat[0, n] + perturb_std * noise_w[j]
    return new


def make_view(sd, w, mean_latent, psi, vd: ViewDraws, hlen, perturb_std=1.0, hf_interp="nearest"):
    """One augmented view -> (stack (1,hlen,H,W), image).  The second forward truncates
    AGAIN (image_augmentor.py:75-79 -> model.py:594-602) and uses the registered
    noise buffers."""
    n_latent = G.num_layers_of(sd) + 1
    new = view_latents(w, mean_latent, psi, n_latent, vd, sd, perturb_std)
    img, taps, _ = G.generator_forward(new, sd, truncation=psi, truncation_latent=mean_latent,
                                       input_is_latent=True)
    stack = pixel_features(taps, hlen, hf_interp)          # F.interpolate(mode=swav_args['hf_interp']), swav_clustering.py:123-124
    if vd.index_map is not None:
        stack = apply_index_map(stack, vd.index_map)
    elif vd.angle is not None:
        stack = apply_index_map(stack, rotflip_index_map(vd.angle, vd.flip, stack.shape[2],
                                                         stack.shape[3]))
    return stack, img


# ----------------------------------------------------------------------------- one training step
@dataclass
class SwavParams:
    proj_w: torch.Tensor             # (D, C)
    proto_w: torch.Tensor            # (K, D)
    proto_b: torch.Tensor            # (K,)
    head: "HeadParams" = None        # projn_nw '1-layer' / '2-layer' (None: 'linear')


def swav_step_loss(sd, mean_latent, psi, images: List[ImageDraws], params: SwavParams, *, hlen,
                   patch_size, num_patches, temperature, eps, niters, source_pdf="uniform",
                   sinkhorn_scope="image", fp64=False, perturb_std=1.0, return_aux=False, hf_interp="nearest"):
    """Loss of one optimizer step over a batch of images (SURVEY §8e semantics).

    scope 'image' : per-image Sinkhorn; loss = mean over images of the reference's
                    per-image loss (== B reference steps with gradient averaging;
                    B=1 is the reference exactly, swav_clustering.py:320-460).
    scope 'global': Sinkhorn over the concatenated (B*N) pixels of each patch index."""
    proj_w, proto_w, proto_b = params.proj_w, params.proto_w, params.proto_b
    stacks, imgs = [], []
    with torch.no_grad():
        for im in images:
            w = G.mapping(im.z, sd)
            vs = [make_view(sd, w, mean_latent, psi, vd, hlen, perturb_std, hf_interp)[0] for vd in im.views]
            stacks.append(vs)
            imgs.append([norm_map(v) for v in vs])
    loss = 0.0
    aux = {}
    nb = len(images)
    for pno in range(num_patches):
        sc = [[], []]
        head = params.head
        if head is not None and head.kind == "2-layer" and sinkhorn_scope == "global":
            # a BatchNorm batch is the rows of one Sinkhorn problem: every image's rows of that view together
            for v in range(2):
                xs = torch.cat([sample_pixels(stacks[b][v], im.picks[pno], patch_size) for b, im in enumerate(images)], 0)
                sc[v] = list(codes(xs, proj_w, proto_w, proto_b, train=True, head=head).chunk(nb, 0))
        else:
            for b, im in enumerate(images):
                for v in range(2):
                    x = sample_pixels(stacks[b][v], im.picks[pno], patch_size)
                    sc[v].append(codes(x, proj_w, proto_w, proto_b, train=True, head=head))
        if sinkhorn_scope == "global":
            s_all, t_all = torch.cat(sc[0], 0), torch.cat(sc[1], 0)
            with torch.no_grad():
                img_s = torch.cat([i[0] for i in imgs], 0)
                img_t = torch.cat([i[1] for i in imgs], 0)
                q_s = sinkhorn_knopp(s_all, eps, niters, source_pdf, img_s, fp64)
                q_t = sinkhorn_knopp(t_all, eps, niters, source_pdf, img_t, fp64)
            loss = loss + swapped_prediction_loss(s_all / temperature, t_all / temperature, q_s, q_t)
            if return_aux and pno == 0:
                aux.update(scores_s=s_all.detach(), scores_t=t_all.detach(), q_s=q_s, q_t=q_t)
        else:
            for b in range(nb):
                with torch.no_grad():
                    q_s = sinkhorn_knopp(sc[0][b], eps, niters, source_pdf, imgs[b][0], fp64)
                    q_t = sinkhorn_knopp(sc[1][b], eps, niters, source_pdf, imgs[b][1], fp64)
                loss = loss + swapped_prediction_loss(sc[0][b] / temperature, sc[1][b] / temperature,
                                                      q_s, q_t) / nb
                if return_aux and pno == 0 and b == 0:
                    aux.update(scores_s=sc[0][b].detach(), scores_t=sc[1][b].detach(), q_s=q_s, q_t=q_t)
    loss = loss / num_patches
    return (loss, aux) if return_aux else loss


def swav_train_step(sd, mean_latent, psi, images, params: SwavParams, bufs, *, lr, momentum, trust,
                    use_larc=True, **kw):
    """S7 renorm -> loss -> backward (S6) -> LARC-SGD (S8).  Mutates params in place.
    Returns (loss value, grads, new momentum buffers)."""
    with torch.no_grad():
        params.proto_w.copy_(normalize_prototypes(params.proto_w))
    leaves = [params.proj_w, params.proto_w, params.proto_b] + (params.head.leaves() if params.head is not None else [])
    for p in leaves:
        p.requires_grad_(True)
        p.grad = None
    loss = swav_step_loss(sd, mean_latent, psi, images, params, **kw)
    loss.backward()
    grads = [p.grad.detach().clone() for p in leaves]
    for p in leaves:
        p.requires_grad_(False)
    with torch.no_grad():
        bufs = larc_sgd_step(leaves, grads, bufs, lr, momentum, trust, use_larc)
    return float(loss.detach()), grads, bufs


def init_linear_like_torch(out_f, in_f, bias):
    """torch.nn.Linear default init draws (kaiming_uniform(a=sqrt5) -> U(-1/sqrt(in), 1/sqrt(in))
    for weight, then the same bound for bias) from the global RNG, in nn.Linear's order."""
    lin = torch.nn.Linear(in_f, out_f, bias=bias)
    return lin.weight.detach().clone(), (lin.bias.detach().clone() if bias else None)


def pretrain_reference_order(sd, style_dim, *, num_latents_for_mean, psi, n_perturb_layers,
                             num_test_samples, num_epochs, num_patches, patch_size, hlen, nclasses,
                             nprototypes, temperature, eps, niters, lr, momentum, trust,
                             use_larc, transforms, source_pdf="uniform", projn_nw="linear"):
    """The reference's constructor + pretrain() RNG order and update rule
    (swav_clustering.py:93-95, :224-237, :245-271, :320-460) with num_samples=1.
    Returns (losses per epoch, SwavParams)."""
    nl = G.num_layers_of(sd)
    with torch.no_grad():
        mean_lat = G.mean_latent(sd, num_latents_for_mean, style_dim)
        for _ in range(num_test_samples):
            torch.randn(1, style_dim)
        for _ in range(num_test_samples):          # test images: randomize_noise=True draws
            for i in range(nl):
                r = 2 ** ((i + 5) // 2)
                torch.empty(1, 1, r, r).normal_()
    proj_w, _ = init_linear_like_torch(nclasses, hlen, False)
    head = None if projn_nw == "linear" else HeadParams.fresh(projn_nw, nclasses)      # Sequential's construction order
    proto_w, proto_b = init_linear_like_torch(nprototypes, nclasses, True)
    params = SwavParams(proj_w, proto_w, proto_b, head)
    bufs = [None] * (3 + (len(head.leaves()) if head is not None else 0))
    size = 2 ** ((nl + 3) // 2)
    losses = []
    for _ in range(num_epochs):
        im = draw_image_reference_order(style_dim, nl, n_perturb_layers, size * size, num_patches,
                                        transforms=transforms)
        loss, _, bufs = swav_train_step(sd, mean_lat, psi, [im], params, bufs, lr=lr,
                                        momentum=momentum, trust=trust, use_larc=use_larc,
                                        hlen=hlen, patch_size=patch_size, num_patches=num_patches,
                                        temperature=temperature, eps=eps, niters=niters,
                                        source_pdf=source_pdf)
        losses.append(loss)
    return losses, params, mean_lat


# ----------------------------------------------------------------------------- P1
def predict_swav_codes(sd, w, mean_latent, psi, proj_w, hlen, head=None, hf_interp="nearest"):
    """predict_swav_codes (swav_clustering.py:659-693): one truncation, registered noise,
    projection on every pixel, argmax over the D projection channels."""
    with torch.no_grad():
        _, taps, _ = G.generator_forward(w, sd, truncation=psi, truncation_latent=mean_latent,
                                         input_is_latent=True)
        stack = pixel_features(taps, hlen, hf_interp)
        b, c, h, wd = stack.shape
        x = torch.squeeze(stack).flatten(1).t()
        preds = codes(x, proj_w, train=False, head=head).reshape(b, proj_w.shape[0], h, wd)
        labels = preds.max(1)[1]
    return preds, labels
