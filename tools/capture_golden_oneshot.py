#!/usr/bin/env python3
"""GV13: golden vectors for the one-shot fine-tune head and the IoU metrics.

Like tools/capture_golden.py it runs only in the build container, imports the reference from
/root/reference at run time and writes plain arrays.  Captured:
  * the reference's OneShotSegmentor (hfc_with_swav/swav_clustering.py:697-758), sizes 'XXS' and
    'XS' (the two every hfc_with_swav / hfc_with_simclr config uses), trained exactly as
    OneShotPipeline.setup_trainer / run_trainer do (src/one_shot_pipeline.py:250-270, :533-581):
    torch.optim.Adam(lr, betas), nn.CrossEntropyLoss on the head's output, StepLR.step() per epoch,
    one labelled sample;  -> initial weights, first forward, per-epoch losses, final weights;
  * lib/metrics/segmentation.py get_mask_iou / get_weighted_iou / get_bin_iou on small masks.
The pipeline class itself is not constructed (it loads checkpoints and datasets at __init__).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from capture_golden import REF, _install_placeholders, _np, out_dir_arg   # noqa: E402


def main():
    out_dir = out_dir_arg(__doc__.splitlines()[0])
    _install_placeholders()
    import types
    for name in ("skimage.segmentation", "skimage.metrics"):     # imported by lib/metrics/segmentation.py, unused
        sys.modules[name] = types.ModuleType(name)               # by the three functions captured here
    sys.path.insert(0, REF)
    import matplotlib
    matplotlib.use("agg")
    import torch
    from torch import nn
    torch.set_num_threads(4)
    from hfc_with_swav import swav_clustering as SC
    from lib.metrics import segmentation as MET
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_supervised_config",
                                                  os.path.join(REF, "configs/trainers/supervised_config.py"))
    tc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tc)

    arrs = dict(lr=np.float64(tc.lr), beta1=np.float64(tc.beta1), beta2=np.float64(tc.beta2),
                step_size=np.int64(tc.scheduler_args["step_size"]), gamma=np.float64(tc.scheduler_args["gamma"]))
    in_ch, n_class, hw, epochs = 16, 5, 20, 6
    for size in ("XXS", "XS", "Lin"):         # 'Lin': the per-pixel Linear + LeakyReLU head (used by no shipped config)
        torch.manual_seed(1300 + len(size) + (7 if size == "Lin" else 0))
        seg = SC.OneShotSegmentor(in_ch=in_ch, n_class=n_class, size=size).train()
        feats = torch.randn(1, in_ch, hw, hw)
        labels = torch.randint(0, n_class, (1, hw, hw))
        arrs[f"{size}_feats"] = _np(feats)
        arrs[f"{size}_labels"] = labels.numpy().astype(np.int64)
        for k, v in seg.state_dict().items():
            arrs[f"{size}_init_{k}"] = _np(v)
        with torch.no_grad():
            arrs[f"{size}_out0"] = _np(seg(feats))
        opt = torch.optim.Adam(seg.parameters(), lr=tc.lr, betas=(tc.beta1, tc.beta2))
        sched = torch.optim.lr_scheduler.StepLR(opt, **tc.scheduler_args)
        losses = []
        for _ in range(epochs):
            out = seg(feats)
            loss = 0.0
            loss = loss + 1.0 * nn.CrossEntropyLoss()(out, labels)
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(loss.item())
            sched.step()
        arrs[f"{size}_losses"] = np.asarray(losses, np.float64)
        for k, v in seg.state_dict().items():
            arrs[f"{size}_final_{k}"] = _np(v)
        with torch.no_grad():
            arrs[f"{size}_pred"] = seg.eval()(feats).max(1)[1].numpy().astype(np.int64)

    # metrics on small masks (classes: background + 3 parts)
    rng = np.random.RandomState(7)
    gt = rng.randint(0, 4, (12, 12)).astype(np.float32)
    pred = gt.copy()
    flip = rng.rand(12, 12) < 0.3
    pred[flip] = rng.randint(0, 4, int(flip.sum()))
    classes = ["background", "a", "b", "c"]
    miou = {c: MET.get_mask_iou(gt, pred, i) for i, c in enumerate(classes)}
    arrs.update(met_gt=gt, met_pred=pred, met_mask_iou=np.asarray([miou[c] for c in classes], np.float64),
                met_w_iou=np.float64(MET.get_weighted_iou(gt, miou, classes)),
                met_bin_iou=np.float64(MET.get_bin_iou(gt, pred)),
                met_empty_iou=np.float64(MET.get_mask_iou(np.zeros((4, 4)), np.zeros((4, 4)), 3)))
    path = os.path.join(out_dir, "gv13_oneshot.npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, {k: np.asarray(v).shape for k, v in list(arrs.items())[:6]})


if __name__ == "__main__":
    main()
