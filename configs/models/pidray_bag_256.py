"""Model config (plugin surface; attribute names follow the reference's configs/models/pidray_bag_256.py)."""
from ganecdotes_amd.config_defaults import baggan_model

globals().update(baggan_model('pidray_256', ['background', 'bag', 'metal', 'liquid', 'hammer', 'tools', 'organic', 'weapon'], truncation=0.9))
