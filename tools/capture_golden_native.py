#!/usr/bin/env python3
"""GV16: golden vectors for the reference's two native-op SIGNATURES and the rest of its test metrics.

Runs only in the build container (imports /root/reference at run time, writes plain arrays):
  * `upfirdn2d_native(input [major,h,w,minor]-shaped call, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1,
    pad_y0, pad_y1)` of models/stylegan2/model.py:61-102 — the torch-op twin of the CUDA op
    lib/gan/optim/upfirdn2d_kernel.cu (same signature, upfirdn2d.cpp:18-35) — on general parameter sets:
    up/down 1-3, asymmetric and negative pads, non-square kernels, minor > 1;
  * `fused_leaky_relu(input, bias, negative_slope, scale)` (model.py:32-43), forward and — through autograd —
    the input gradient the CUDA op produces with grad=1 (fused_act.py:61-67: gated by the sign of the OUTPUT);
  * lib/metrics/segmentation.py get_bb_iou / get_mask_dice / get_bb_dice / get_pd_at_iou_threshold /
    get_iou_vs_pd_curve (the remaining entries of run_tests' `results`, src/one_shot_pipeline.py:726-884).
"""
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from capture_golden import REF, _install_placeholders, _np, out_dir_arg   # noqa: E402

UPFIRDN_CASES = [   # (major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)
    (3, 9, 9, 1, 4, 4, 1, 1, 1, 1, 1, 1, 1, 1),        # mode 1 (blur after the transposed conv)
    (3, 8, 8, 1, 4, 4, 2, 2, 1, 1, 2, 1, 2, 1),        # mode 3 (skip upsample)
    (2, 11, 7, 1, 4, 4, 1, 1, 2, 2, 1, 1, 1, 1),       # downsample (discriminator blur + stride, mode 5 shape)
    (2, 6, 10, 3, 3, 5, 2, 3, 1, 2, 0, 2, 3, 1),       # non-square kernel, mixed up / down, minor = 3
    (2, 12, 12, 4, 4, 4, 1, 1, 1, 1, -1, 2, 1, -2),    # negative pads = crops
    (1, 5, 5, 2, 2, 2, 3, 3, 2, 2, 1, 1, 1, 1),        # up 3 / down 2
]


def main():
    out_dir = out_dir_arg(__doc__.splitlines()[0])
    _install_placeholders()
    for name in ("skimage.segmentation", "skimage.metrics"):
        sys.modules[name] = types.ModuleType(name)
    sys.path.insert(0, REF)
    import matplotlib
    matplotlib.use("agg")
    import pandas as pd
    import torch
    torch.set_num_threads(4)
    from models.stylegan2 import model as M
    from lib.metrics import segmentation as MET

    arrs = dict(upfirdn_cases=np.asarray(UPFIRDN_CASES, np.int64))
    torch.manual_seed(1600)
    for i, (major, ih, iw, minor, kh, kw, ux, uy, dx, dy, px0, px1, py0, py1) in enumerate(UPFIRDN_CASES):
        x = torch.randn(major, ih, iw, minor)
        k = torch.randn(kh, kw)
        # upfirdn2d_native takes (N, C, H, W) and views it as (-1, H, W, 1) (:64-66): feed the minor
        # channels as C of a channels-first copy and bring the result back to [major][h][w][minor]
        y = M.upfirdn2d_native(x.permute(0, 3, 1, 2).contiguous(), k, ux, uy, dx, dy, px0, px1, py0, py1)
        arrs[f"up{i}_x"], arrs[f"up{i}_k"] = _np(x), _np(k)
        arrs[f"up{i}_y"] = _np(y.permute(0, 2, 3, 1).contiguous())

    x = torch.randn(2, 6, 5, 7, requires_grad=True)
    b = torch.randn(6)
    y = M.fused_leaky_relu(x, b, 0.2, 2 ** 0.5)
    gy = torch.randn_like(y)
    (gx,) = torch.autograd.grad(y, x, gy)
    arrs.update(fl_x=_np(x), fl_b=_np(b), fl_y=_np(y), fl_gy=_np(gy), fl_gx=_np(gx))
    x2 = torch.randn(9, 4)
    # (the torch twin ignores its negative_slope argument — 0.2 is hard-coded, model.py:37,43 — while the CUDA op
    #  honours alpha, fused_bias_act_kernel.cu:60; captured at 0.2, where the two agree)
    arrs.update(fl2_x=_np(x2), fl2_b=_np(b[:4]), fl2_y=_np(M.fused_leaky_relu(x2, b[:4], 0.2, 1.5)))

    # metrics: 6 samples, classes background + 3 parts
    classes = ["background", "a", "b", "c"]
    rng = np.random.RandomState(16)
    gts, preds = [], []
    for s in range(6):
        gt = np.zeros((16, 16), np.float32)
        for c in range(1, 4):
            y0, x0 = rng.randint(0, 10, 2)
            gt[y0:y0 + rng.randint(2, 6), x0:x0 + rng.randint(2, 6)] = c
        pred = gt.copy()
        flip = rng.rand(16, 16) < 0.15 * (s + 1) / 3
        pred[flip] = rng.randint(0, 4, int(flip.sum()))
        if s == 4:
            pred[pred == 2] = 0                       # a class the prediction misses entirely
        if s == 5:
            gt[gt == 3] = 0                           # a class absent from the ground truth
        gts.append(gt); preds.append(pred)
    arrs["met_gt"], arrs["met_pred"] = np.stack(gts), np.stack(preds)
    mi = np.array([[MET.get_mask_iou(g, p, i) for i in range(4)] for g, p in zip(gts, preds)], np.float64)
    arrs["met_mask_iou"] = mi
    arrs["met_bb_iou"] = np.array([[MET.get_bb_iou(g, p, i) for i in range(4)] for g, p in zip(gts, preds)], np.float64)
    arrs["met_mask_dice"] = np.array([[MET.get_mask_dice(g, p, i) for i in range(4)] for g, p in zip(gts, preds)], np.float64)
    arrs["met_bb_dice"] = np.array([[MET.get_bb_dice(g, p, i) for i in range(4)] for g, p in zip(gts, preds)], np.float64)
    df = pd.DataFrame(data=mi, columns=classes)
    pdv = MET.get_pd_at_iou_threshold(df, iou_thr=0.5, classes=classes)
    arrs["met_pd"] = np.array([pdv[c] for c in classes], np.float64)
    curve = MET.get_iou_vs_pd_curve(iou_pd=df, classes=classes)
    arrs["met_curve"] = np.array([np.asarray(curve[c], np.float64) for c in classes + ["Mean"]])
    import io
    buf = io.StringIO()
    df.to_csv(buf)                                     # the exact text of tests/mask_iou_results.csv
    arrs["met_csv"] = np.frombuffer(buf.getvalue().encode(), np.uint8)

    path = os.path.join(out_dir, "gv16_native_ops.npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
