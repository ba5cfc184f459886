"""Model config (plugin surface; attribute names follow the reference's configs/models/lsun_car_512.py)."""
from ganecdotes_amd.config_defaults import stylegan_model

globals().update(stylegan_model('car_512', 'stylegan2-car-config-f.pt', ['background', 'body', 'door', 'window', 'windshield', 'hood', 'wheel', 'lights'], one_shot_ind=3))
