#!/usr/bin/env python3
"""GV17: golden vectors for the projection networks `projn_nw='1-layer'` and `'2-layer'`.

Runs only in the build container; imports the reference from /root/reference at run time and writes plain arrays.
hfc_with_swav/swav_clustering.py:250-269 builds, beside the 'linear' projection every shipped config uses,
  '1-layer'  Linear(hlen, D, bias=False) -> LeakyReLU
  '2-layer'  Linear -> BatchNorm1d -> LeakyReLU -> Linear(D, D, bias=False) -> BatchNorm1d -> Tanh
Captured per kind, on the mini generator of tools/capture_golden.py (seed 42, rotation / flip disabled, LARC
pass-through, like GV10 / GV11):
  * three epochs of the reference's own pretrain(): the loss per epoch, the projection's full state_dict (BatchNorm
    running statistics and batch counters included) and the prototype layer afterwards;
  * predict_swav_codes on a fresh latent (BatchNorm still in train mode: the reference never calls .eval()), and the
    running statistics after it;
  * get_swav_codes_from_hidden_features(train=True / False) on one fixed feature map with the trained modules.
"""
import functools
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from capture_golden import REF, _install_placeholders, _np, build_mini_generator, build_swav, out_dir_arg   # noqa: E402


def main():
    out_dir = out_dir_arg(__doc__.splitlines()[0])
    Writer = _install_placeholders()
    sys.path.insert(0, REF)
    import matplotlib
    matplotlib.use("agg")
    import torch
    torch.set_num_threads(1)              # BatchNorm's column sums: one thread, so the fixture regenerates bit for bit
    from models.stylegan2 import model as M
    from hfc_with_swav import swav_clustering as SC
    make_swav = functools.partial(build_swav, SC, Writer)
    tmp = tempfile.mkdtemp(prefix="gv17_")

    arrs = {}
    for kind in ("1-layer", "2-layer"):
        tag = kind[0] + "l"
        g = build_mini_generator(M, 117)
        torch.manual_seed(42)
        np.random.seed(42)
        sw = make_swav(g, tmp, hlen=56, nclasses=8, nproto=12, patch=300, eps=0.05, num_epochs=3, num_patches=2,
                       projn_nw=kind)
        sw.pretrain(None, num_test_samples=2)
        if kind == "1-layer":
            arrs.update({f"sd_{k_}": _np(v) for k_, v in g.state_dict().items()})
        arrs[f"{tag}_losses"] = np.asarray([s[1] for s in sw.writer.scalars], np.float64)
        arrs.update({f"{tag}_proj_{k_}": _np(v) for k_, v in sw.projection.state_dict().items()})
        arrs[f"{tag}_proto_weight"] = _np(sw.prototype.weight)
        arrs[f"{tag}_proto_bias"] = _np(sw.prototype.bias)
        # fixed feature map through the trained modules (train=True takes the first patch_size picks)
        torch.manual_seed(1170)
        hfeat = torch.randn(1, 56, 20, 20)
        picks = torch.randperm(20 * 20)
        with torch.no_grad():
            sc_train = sw.get_swav_codes_from_hidden_features(hfeat, picks=picks, train=True)
            state_mid = {k_: v.clone() for k_, v in sw.projection.state_dict().items() if "running" in k_}
            sc_pred = sw.get_swav_codes_from_hidden_features(hfeat, (1, 8, 20, 20), train=False)
        arrs.update({f"{tag}_hfeat": _np(hfeat), f"{tag}_picks": _np(picks), f"{tag}_scores_train": _np(sc_train),
                     f"{tag}_scores_pred": _np(sc_pred)})
        arrs.update({f"{tag}_proj_after_train_call_{k_}": _np(v) for k_, v in state_mid.items()})
        arrs.update({f"{tag}_proj_after_pred_call_{k_}": _np(v) for k_, v in sw.projection.state_dict().items()
                     if "running" in k_})
        torch.manual_seed(1171)
        lat = torch.randn(1, 16)
        with torch.no_grad():
            wlat = g.style(lat)
            preds, labels = sw.predict_swav_codes(wlat)
        arrs.update({f"{tag}_pred_w": _np(wlat), f"{tag}_mean_latent": _np(sw.mean_latent), f"{tag}_preds": _np(preds),
                     f"{tag}_labels": _np(labels)})
        arrs.update({f"{tag}_proj_after_predict_{k_}": _np(v) for k_, v in sw.projection.state_dict().items()
                     if "running" in k_})
    path = os.path.join(out_dir, "gv17_projection_heads.npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {path}: {len(arrs)} arrays ({os.path.getsize(path)} B)")


if __name__ == "__main__":
    main()
