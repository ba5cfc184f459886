"""`from models.baggan.models import StyleGANGenerator` (src/one_shot_pipeline.py:150-154 via BagGANHQ): the
MI355X-native BagGAN generator — same constructor, forward signature and state_dict keys as
models/baggan/models.py:96-378."""
from ganecdotes_amd.baggan import RES_TO_CHANNEL_MAP as DEFAULT_RES_TO_CHANNEL_MAP    # noqa: F401
from ganecdotes_amd.baggan import StyleGANGenerator                                      # noqa: F401
