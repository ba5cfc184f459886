"""Model config (plugin surface; attribute names follow the reference's configs/models/lsun_horse_256.py)."""
from ganecdotes_amd.config_defaults import stylegan_model

globals().update(stylegan_model('horse_256', 'stylegan2-horse-config-f.pt', ['background', 'f_leg', 'b_leg', 'torso', 'neck', 'tail', 'head', 'rider'], one_shot_ind=4))
