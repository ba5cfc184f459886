"""Host-side randomness of one training image, drawn in the reference's order.

"Same latent seeds" (BASELINE north star) only holds if the torch / numpy GLOBAL RNGs are
consumed exactly as the reference consumes them.  Everything random
in a step is drawn here, on the host, once — and then handed explicitly to the kernels, so
the same draws can be replayed through the CPU oracle for parity tests.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch


@dataclass
class ViewDraws:
    layer_no: int                      # perturbed StyleGAN block (latents 2l, 2l+1)
    map_noise: torch.Tensor            # (2, S) z-noise pushed through the mapping MLP
    angle: Optional[float] = None      # degrees, U(-10, 10); None = rotation/flip disabled
    flip: bool = False


@dataclass
class ImageDraws:
    z: torch.Tensor                                         # (1, S)
    views: List[ViewDraws] = field(default_factory=list)    # [s, t]
    picks: List[torch.Tensor] = field(default_factory=list)  # num_patches x randperm(H*W) (int64, CPU)
    patch_offsets: Optional[List[int]] = None               # 'patch' sampling: the crop offset of every patch


def patch_pixels(offset, patch_size, side):
    """pixel indices of the reference's 'patch' sampling (swav_clustering.py:150-158): the square crop
    [o:o+ps, o:o+ps] of a side x side image — the SAME offset on both axes — in row-major order"""
    ax = offset + torch.arange(patch_size)
    return (ax[:, None] * side + ax[None, :]).reshape(-1)


def draw_image(style_dim, gen_num_layers, n_perturb_layers, hw, num_patches, transforms=True, degrees=10.0,
               layer_no=None, cpu_reference_rng=False, sampling_method='random', patch_size=None):
    """One image's draws in reference order: z (swav_clustering.py:323); per view
    `np.random.choice` of the layer (:611) and the mapping noises of the two perturbed latents
    (lib/oneshot/image_augmentor.py:47); rotation angle then flip for s, then for t (:358-359);
    one randperm per patch (:388) — or, with sampling_method='patch', one `np.random.choice(h - patch_size)`
    per patch (:383-385, numpy's global RNG), kept in `patch_offsets` next to the crop's pixel list.

    On a GPU the reference draws the per-layer noise maps of its discarded first generator call
    (models/stylegan2/model.py:380) and all 2*n_perturb_layers mapping noises from the DEVICE
    generator, so they never touch the host RNG stream.  `cpu_reference_rng=True` reproduces the
    host stream of a CPU run of the reference instead (noise maps drawn and thrown away, all
    2*n_perturb_layers mapping noises drawn): that is what the golden fixtures were captured with."""
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
            views.append(ViewDraws(lno, torch.cat([maps[2 * lno], maps[2 * lno + 1]], 0)))
        else:
            views.append(ViewDraws(lno, torch.randn(2, style_dim)))
    if transforms:
        for v in views:
            v.angle = float(torch.empty(1).uniform_(-degrees, degrees).item())
            v.flip = bool(torch.rand(1) < 0.5)
    if sampling_method == 'patch' and num_patches > 0:
        side = int(round(hw ** 0.5))
        offs = [int(np.random.choice(side - patch_size)) for _ in range(num_patches)]
        im = ImageDraws(z, views, [patch_pixels(o, patch_size, side) for o in offs])
        im.patch_offsets = offs
        return im
    picks = [torch.randperm(hw) for _ in range(num_patches)] if num_patches > 0 else []
    return ImageDraws(z, views, picks)
