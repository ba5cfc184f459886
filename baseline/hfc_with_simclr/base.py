"""Plugin export of --method=hfc_with_simclr (same two names as the reference's
baseline/hfc_with_simclr/base.py): the MI355X-native classes."""
from ganecdotes_amd.segmentor import OneShotSegmentor as segmentor              # noqa: F401
from ganecdotes_amd.simclr_clustering import SimCLRClustering as preprocessor   # noqa: F401
