"""Model config (plugin surface; attribute names follow the reference's configs/models/ffhq_256_rp_eyeg.py)."""
from ganecdotes_amd.config_defaults import stylegan_model

globals().update(stylegan_model('ffhq_256', 'stylegan2-ffhq-config-f.pt', ['background', 'glasses'], one_shot_ind=21, latents_dir='ffhq_256_rp_eyeg'))
