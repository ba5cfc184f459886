"""Segmentor / preprocessor config (plugin surface; values follow the reference's
configs/segmentors/hfc_with_swav_pidray_config.py)."""
from ganecdotes_amd.config_defaults import BAGGAN_256_HF_DIM, swav_segmentor

globals().update(swav_segmentor(4000, 0.005, 'XXS', layer_hf_dim=BAGGAN_256_HF_DIM, hlen=2528))
