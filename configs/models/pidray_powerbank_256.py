"""Model config (plugin surface; attribute names follow the reference's configs/models/pidray_powerbank_256.py)."""
from ganecdotes_amd.config_defaults import baggan_model

globals().update(baggan_model('pidray_powerbank_256', ['background', 'powerbank'], 19))
