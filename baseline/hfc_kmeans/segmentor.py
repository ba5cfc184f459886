"""Import path of the reference's baseline/hfc_kmeans/segmentor.py."""
from ganecdotes_amd.segmentor import OneShotSegmentor                   # noqa: F401
from ganecdotes_amd.kmeans import HFCPreprocessor                       # noqa: F401
