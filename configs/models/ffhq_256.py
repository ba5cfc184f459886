"""Model config (plugin surface; attribute names follow the reference's configs/models/ffhq_256.py)."""
from ganecdotes_amd.config_defaults import stylegan_model

globals().update(stylegan_model('ffhq_256', 'stylegan2-ffhq-config-f.pt', ['background', 'skin', 'hair', 'eye', 'eyebrow', 'nose', 'mouth', 'ear']))
