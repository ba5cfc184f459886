"""k-means baseline config (values follow the reference's configs/segmentors/hfc_kmeans_config.py)."""
from ganecdotes_amd.config_defaults import kmeans_segmentor

globals().update(kmeans_segmentor())
