"""Same-box A/B of one product choice on the driver's own command: runs bench.py in THIS process with one function of
ganecdotes_amd.ops wrapped so that a choice falls back to its previous form — no switch in the product for it.
    python tools/lab/ab_patch.py none|fqb|lvl -- [bench.py arguments]
      fqb   fineq_bwd without the taps' maximum  -> its bf16x6 form
      lvl   tables route: the scatters' bound of |G| withheld -> the pyramid-backward level GEMMs on bf16x6
"""
import os
import runpy
import sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, root)
from ganecdotes_amd import ops  # noqa: E402

which = sys.argv[1]
rest = sys.argv[3:] if len(sys.argv) > 2 and sys.argv[2] == "--" else sys.argv[2:]
if which == "fqb":
    _bwd = ops.fineq_bwd

    def fineq_bwd(*a, t_amax=None, **k):
        return _bwd(*a, **k)
    ops.fineq_bwd = fineq_bwd
if which == "lvl":
    from ganecdotes_amd.swav_clustering import SwAVClustering
    _tb = SwAVClustering._tables_backward

    def _tables_backward(self, *a, **k):
        _tb(self, *a, **k)
        return None
    SwAVClustering._tables_backward = _tables_backward
if which not in ("none", "fqb", "lvl"):
    raise SystemExit(f"unknown patch {which!r}")
sys.argv = [os.path.join(root, "bench.py")] + rest
runpy.run_path(os.path.join(root, "bench.py"), run_name="__main__")
