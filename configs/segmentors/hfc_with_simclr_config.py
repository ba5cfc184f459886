"""SimCLR baseline config (values follow the reference's configs/segmentors/hfc_with_simclr_config.py)."""
from ganecdotes_amd.config_defaults import simclr_segmentor

globals().update(simclr_segmentor())
