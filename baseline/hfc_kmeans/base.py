"""Plugin export of --method=hfc_kmeans (same two names as the reference's
baseline/hfc_kmeans/base.py): the MI355X-native classes."""
from ganecdotes_amd.segmentor import OneShotSegmentor as segmentor      # noqa: F401  ('S' / 'M' / 'L' plans)
from ganecdotes_amd.kmeans import HFCPreprocessor as preprocessor       # noqa: F401
