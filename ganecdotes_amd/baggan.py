"""BagGAN generator (pidray-* models; SURVEY.md par. 8f rank 3) on the HIP generator.

The reference's `StyleGANGenerator` (models/baggan/models.py:96-378, blocks.py) is the rosinality StyleGAN2
generator re-spelled: same mapping network, modulated convs with demodulation, transposed-conv + FIR
upsampling, noise injection, FusedLeakyReLU and skip-connected image converters, same forward.  Only the
parameter names and the width table differ (GV14 pins this: the reference BagGAN's outputs are reproduced
through the key translation below).  So the class is `ganecdotes_amd.generator.Generator` with the
reference's constructor, the BagGAN width table and a state_dict that speaks the BagGAN key names — real
`pidray` checkpoints (`generator.module` of BagGANHQ, src/one_shot_pipeline.py:150-154) load unchanged.
"""
import math
import re

import torch
from torch import nn

from .generator import Generator

CHL_MULTIPLIER = 2                                           # models.py:14
RES_TO_CHANNEL_MAP = {4: 512, 8: 512, 16: 256 * CHL_MULTIPLIER, 32: 128 * CHL_MULTIPLIER, 64: 64 * CHL_MULTIPLIER,
                      128: 32 * CHL_MULTIPLIER, 256: 16 * CHL_MULTIPLIER, 512: 8 * CHL_MULTIPLIER}     # models.py:25-32

_TO_SG2 = [
    (r"^style\.mapper\.(\d+)\.(weight|bias)$", r"style.\1.\2"),
    (r"^const_input_block\.const_block$", "input.input"),
    (r"^conv_init\.style_block\.weight$", "conv1.conv.weight"),
    (r"^conv_init\.style_block\.mod\.(weight|bias)$", r"conv1.conv.modulation.\1"),
    (r"^conv_init\.noise_block\.weight$", "conv1.noise.weight"),
    (r"^conv_init\.activation\.bias$", "conv1.activate.bias"),
    (r"^x_to_img_init\.conv\.weight$", "to_rgb1.conv.weight"),
    (r"^x_to_img_init\.conv\.mod\.(weight|bias)$", r"to_rgb1.conv.modulation.\1"),
    (r"^x_to_img_init\.bias$", "to_rgb1.bias"),
    (r"^conv_blks\.(\d+)\.style_block\.weight$", r"convs.\1.conv.weight"),
    (r"^conv_blks\.(\d+)\.style_block\.blur\.kernel$", r"convs.\1.conv.blur.kernel"),
    (r"^conv_blks\.(\d+)\.style_block\.mod\.(weight|bias)$", r"convs.\1.conv.modulation.\2"),
    (r"^conv_blks\.(\d+)\.noise_block\.weight$", r"convs.\1.noise.weight"),
    (r"^conv_blks\.(\d+)\.activation\.bias$", r"convs.\1.activate.bias"),
    (r"^x_to_img_blks\.(\d+)\.conv\.weight$", r"to_rgbs.\1.conv.weight"),
    (r"^x_to_img_blks\.(\d+)\.conv\.mod\.(weight|bias)$", r"to_rgbs.\1.conv.modulation.\2"),
    (r"^x_to_img_blks\.(\d+)\.bias$", r"to_rgbs.\1.bias"),
    (r"^x_to_img_blks\.(\d+)\.upsample\.kernel$", r"to_rgbs.\1.upsample.kernel"),
    (r"^noise_blks\.noise_(\d+)$", r"noises.noise_\1"),
]
_FROM_SG2 = [
    (r"^style\.(\d+)\.(weight|bias)$", r"style.mapper.\1.\2"),
    (r"^input\.input$", "const_input_block.const_block"),
    (r"^conv1\.conv\.weight$", "conv_init.style_block.weight"),
    (r"^conv1\.conv\.modulation\.(weight|bias)$", r"conv_init.style_block.mod.\1"),
    (r"^conv1\.noise\.weight$", "conv_init.noise_block.weight"),
    (r"^conv1\.activate\.bias$", "conv_init.activation.bias"),
    (r"^to_rgb1\.conv\.weight$", "x_to_img_init.conv.weight"),
    (r"^to_rgb1\.conv\.modulation\.(weight|bias)$", r"x_to_img_init.conv.mod.\1"),
    (r"^to_rgb1\.bias$", "x_to_img_init.bias"),
    (r"^convs\.(\d+)\.conv\.weight$", r"conv_blks.\1.style_block.weight"),
    (r"^convs\.(\d+)\.conv\.blur\.kernel$", r"conv_blks.\1.style_block.blur.kernel"),
    (r"^convs\.(\d+)\.conv\.modulation\.(weight|bias)$", r"conv_blks.\1.style_block.mod.\2"),
    (r"^convs\.(\d+)\.noise\.weight$", r"conv_blks.\1.noise_block.weight"),
    (r"^convs\.(\d+)\.activate\.bias$", r"conv_blks.\1.activation.bias"),
    (r"^to_rgbs\.(\d+)\.conv\.weight$", r"x_to_img_blks.\1.conv.weight"),
    (r"^to_rgbs\.(\d+)\.conv\.modulation\.(weight|bias)$", r"x_to_img_blks.\1.conv.mod.\2"),
    (r"^to_rgbs\.(\d+)\.bias$", r"x_to_img_blks.\1.bias"),
    (r"^to_rgbs\.(\d+)\.upsample\.kernel$", r"x_to_img_blks.\1.upsample.kernel"),
    (r"^noises\.noise_(\d+)$", r"noise_blks.noise_\1"),
]


def _translate(key, rules):
    for pat, rep in rules:
        if re.match(pat, key):
            return re.sub(pat, rep, key)
    return None


class StyleGANGenerator(Generator):
    """models/baggan/models.py:96-256: `StyleGANGenerator(latent_dims, img_resolution, mlp_layers, mlp_lr,
    img_chls, fir_filter, res2chlmap)`; forward / mean_latent / get_latent / make_noise as the reference."""

    def __init__(self, latent_dims, img_resolution, mlp_layers=8, mlp_lr=0.01, img_chls=1, fir_filter=(1, 3, 3, 1),
                 res2chlmap=None, _channels=None):
        if not (img_resolution > 16 and img_resolution & (img_resolution - 1) == 0):
            raise AssertionError("img_resolution must be a power of two > 16")                 # models.py:111
        z_dim, w_dim = latent_dims
        # the reference ignores a passed res2chlmap (models.py:121-122 only assigns the default); `_channels`
        # is this build's opt-in for fixture-sized models
        super().__init__(img_resolution, w_dim, mlp_layers, channel_multiplier=CHL_MULTIPLIER,
                         blur_kernel=list(fir_filter), lr_mlp=mlp_lr, channels=dict(_channels or RES_TO_CHANNEL_MAP))
        self.z_dim, self.w_dim = z_dim, w_dim
        self.res_log = self.log_size
        # constructed and never used by forward (models.py:204-208); kept so that checkpoints round-trip
        self.head_m = nn.Sequential(nn.Conv2d(1, 1, 3, 1, 1), nn.Conv2d(1, 1, 3, 1, 1), nn.Conv2d(1, 1, 3, 1, 1),
                                    nn.Conv2d(1, 1, 3, 1, 1), nn.Tanh())

    # ------------------------------------------------------------------ BagGAN-named checkpoints
    def state_dict(self, *a, **k):
        out = type(super().state_dict(*a, **k))()
        for key, v in super().state_dict(*a, **k).items():
            out[key if key.startswith("head_m.") else _translate(key, _FROM_SG2)] = v
        return out

    def load_state_dict(self, sd, strict=True):
        conv = {}
        unexpected = []
        for key, v in sd.items():
            if key.startswith("head_m."):
                conv[key] = v
                continue
            t = _translate(key, _TO_SG2)
            if t is None:
                unexpected.append(key)
            else:
                conv[t] = v
        res = super().load_state_dict(conv, strict=False)
        missing = [(_translate(m, _FROM_SG2) or m) for m in res.missing_keys]
        unexpected += list(res.unexpected_keys)
        if strict and (missing or unexpected):
            raise RuntimeError(f"BagGAN load_state_dict: missing {missing}, unexpected {unexpected}")
        return nn.modules.module._IncompatibleKeys(missing, unexpected)
