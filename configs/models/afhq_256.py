"""Model config (plugin surface; attribute names follow the reference's configs/models/afhq_256.py)."""
from ganecdotes_amd.config_defaults import stylegan_model

globals().update(stylegan_model('afhq_256', 'stylegan2-afhq.pth', ['background', 'head', 'neck', 'mouth', 'eye', 'nose', 'ear']))
