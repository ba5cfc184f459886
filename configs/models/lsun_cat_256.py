"""Model config (plugin surface; attribute names follow the reference's configs/models/lsun_cat_256.py)."""
from ganecdotes_amd.config_defaults import stylegan_model

globals().update(stylegan_model('lsun_cat_256', 'stylegan2-cat-config-f.pt', ['background', 'head', 'eye', 'nose', 'ear', 'neck', 'torso', 'tail', 'f_leg', 'b_leg']))
