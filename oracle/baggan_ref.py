"""TEST INFRASTRUCTURE ONLY — the BagGAN generator (pidray-* models) restated on
top of oracle/stylegan2_ref.py.  The reference's models/baggan/models.py:96-378 + blocks.py is a re-spelling
of the rosinality StyleGAN2 generator: same mapping network (blocks.py:59-130 == EqualLinear), constant input
(:155-200), modulated conv with the same demodulation and the same transposed-conv + blur upsampling
(:346-505), noise injection (:523-558), FusedLeakyReLU, skip-connected x->image converters (:615-685) and the
same forward (models.py:257-378).  Only the parameter names differ, so the restatement is a key translation.
Pinned by tests/golden/gv14_baggan.npz (tools/capture_golden_baggan.py)."""
import re

from . import stylegan2_ref as G

_RULES = [
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


def to_stylegan2_keys(sd):
    """BagGAN state_dict -> the rosinality key names stylegan2_ref uses; `head_m.*` (constructed at
    models.py:204-208, never used by forward) is dropped."""
    out = {}
    for k, v in sd.items():
        if k.startswith("head_m."):
            continue
        for pat, rep in _RULES:
            if re.match(pat, k):
                out[re.sub(pat, rep, k)] = v
                break
        else:
            raise KeyError(f"unexpected BagGAN key {k}")
    return out


def generator_forward(styles, sd, **kw):
    return G.generator_forward(styles, to_stylegan2_keys(sd), **kw)


def mapping(z, sd):
    return G.mapping(z, to_stylegan2_keys(sd))
