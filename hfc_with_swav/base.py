"""Plugin export of --method=hfc_with_swav (same two names as the reference's
hfc_with_swav/base.py:1-2): the MI355X-native classes."""
from ganecdotes_amd.segmentor import OneShotSegmentor as segmentor            # noqa: F401
from ganecdotes_amd.swav_clustering import SwAVClustering as preprocessor     # noqa: F401
