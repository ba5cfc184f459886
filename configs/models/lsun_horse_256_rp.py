"""Model config (plugin surface; attribute names follow the reference's configs/models/lsun_horse_256_rp.py)."""
from ganecdotes_amd.config_defaults import stylegan_model

globals().update(stylegan_model('horse_256', 'stylegan2-horse-config-f.pt', ['background', 'rider'], one_shot_ind=4, latents_dir='horse_256_rp'))
