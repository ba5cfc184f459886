"""Import path of the reference's baseline/hfc_kmeans/hfc_kmeans_clustering.py for the classes this build has."""
from ganecdotes_amd.kmeans import BaseHFCModel, FlatKMeansHFC, HierarchicalKMeansHFC           # noqa: F401
