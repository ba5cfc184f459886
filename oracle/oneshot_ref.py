"""TEST INFRASTRUCTURE ONLY — CPU restatement of the one-shot fine-tune head and the IoU metrics
(SURVEY.md §8f rank 1).  Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline only;
the product path never touches it.  Pinned by tests/golden/gv13_oneshot.npz (captured from the
reference's own OneShotSegmentor / torch.optim.Adam / nn.CrossEntropyLoss composition and from
lib/metrics/segmentation.py by tools/capture_golden_oneshot.py).

Each function cites the reference lines it follows.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- head
def segmentor_plan(size, in_ch, n_class):
    """[(c_in, c_out, dilation, leaky_relu_after)] of OneShotSegmentor
    (hfc_with_swav/swav_clustering.py:697-741).  zip() stops at len(dilations) and the trailing
    LeakyReLU is dropped, so 'XXS' is ONE conv in_ch -> 12 whatever n_class is.  'Lin' (:708-712) is a per-pixel
    nn.Linear(in_ch, n_class) FOLLOWED by its LeakyReLU(0.2): dilation 0 marks the 1x1 form."""
    if size == 'Lin':
        return [(in_ch, n_class, 0, True)]
    dil = {'XXS': [1], 'XS': [1, 2, 1], 'S': [1, 2, 1, 2, 1], 'M': [1, 2, 4, 1, 2, 4, 1],
           'L': [1, 2, 4, 8, 1, 2, 4, 8, 1]}[size]
    ch = {'XXS': [12], 'XS': [16, 8], 'S': [128, 64, 64, 32], 'M': [128, 64, 64, 64, 64, 32],
          'L': [128, 64, 64, 64, 64, 64, 64, 32]}[size]
    ch = [in_ch] + ch + [n_class]
    plan = [(a, b, d, True) for d, a, b in zip(dil, ch[:-1], ch[1:])]
    plan[-1] = plan[-1][:3] + (False,)
    return plan


def segmentor_forward(x, params, plan):
    """OneShotSegmentor.forward (:744-758) for the conv sizes.  x (B, C, H, W); params = [w0, b0, w1, ...]
    with w (c_out, c_in, 3, 3)."""
    for i, (_, _, d, act) in enumerate(plan):
        w = params[2 * i]
        if w.dim() == 2:                      # 'Lin' (:748-755): view / permute / Linear / permute back == a 1x1 conv
            x = F.conv2d(x, w[:, :, None, None], params[2 * i + 1])
        else:
            x = F.conv2d(x, w, params[2 * i + 1], padding=d, dilation=d)
        if act:
            x = F.leaky_relu(x, 0.2)
    return x


def adam_step(params, grads, m, v, step, lr, beta1, beta2, eps=1e-8):
    """torch.optim.Adam (the optimizer of setup_trainer, src/one_shot_pipeline.py:257-261; defaults
    eps=1e-8, weight_decay=0, amsgrad=False), restated from its public definition:
    m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)."""
    bc1, bc2 = 1.0 - beta1 ** step, 1.0 - beta2 ** step
    for p, g, mi, vi in zip(params, grads, m, v):
        mi.mul_(beta1).add_(g, alpha=1.0 - beta1)
        vi.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
        denom = (vi.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(mi, denom, value=-lr / bc1)


def finetune(feats, labels, params, plan, epochs, lr, beta1, beta2, step_size=500, gamma=0.1, orders=None):
    """run_trainer (src/one_shot_pipeline.py:533-581): per epoch, for every labelled sample (in
    `orders[epoch]`, the reference's random.shuffle of range(n)): out = head(features[idx]);
    loss = CrossEntropyLoss(out, label); Adam step; then StepLR.step() per epoch (:578).
    (The reference's Resize calls are identities when the head output is at image_size.)
    Returns (last loss of every epoch, params)."""
    params = [p.clone().requires_grad_(True) for p in params]
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    n, step, losses = feats.shape[0], 0, []
    for e in range(epochs):
        cur_lr = lr * gamma ** (e // step_size)
        order = list(range(n)) if orders is None else orders[e]
        for idx in order:
            out = segmentor_forward(feats[idx:idx + 1], params, plan)
            loss = F.cross_entropy(out, labels[idx:idx + 1].long())
            grads = torch.autograd.grad(loss, params)
            step += 1
            with torch.no_grad():
                adam_step(params, grads, m, v, step, cur_lr, beta1, beta2)
        losses.append(float(loss.detach()))
    return losses, [p.detach() for p in params]


# ----------------------------------------------------------------------------- metrics
def mask_iou(gt_mask, pred_mask, label):
    """get_mask_iou (lib/metrics/segmentation.py:9-29)."""
    gt, pred = (np.asarray(gt_mask) == label), (np.asarray(pred_mask) == label)
    union = np.count_nonzero(gt | pred)
    return np.count_nonzero(gt & pred) / union if union > 0 else 0


def weighted_iou(gt_mask, mask_iou_by_class, classes):
    """get_weighted_iou (:94-108): area-weighted sum over the non-'background' classes."""
    gt_mask = np.asarray(gt_mask)
    area = gt_mask.shape[0] * gt_mask.shape[1]
    return sum(np.count_nonzero(gt_mask == i) / area * mask_iou_by_class[c]
               for i, c in enumerate(classes) if c != 'background')


def bin_iou(gt_mask, pred_mask):
    """get_bin_iou (:111-125): foreground (label > 0) IoU."""
    gt, pred = np.asarray(gt_mask) > 0, np.asarray(pred_mask) > 0
    union = np.count_nonzero(gt | pred)
    return np.count_nonzero(gt & pred) / union if union > 0 else 0


def run_tests_metrics(gt_masks, pred_masks, classes):
    """the per-sample IoU block of run_tests (src/one_shot_pipeline.py:726-748): per-class mask IoU,
    area-weighted IoU and binary IoU for every test sample."""
    res = dict(mask_iou=[], w_iou=[], bin_iou=None)
    for gt, pred in zip(gt_masks, pred_masks):
        mi = {c: mask_iou(gt, pred, i) for i, c in enumerate(classes)}
        res['mask_iou'].append(mi)
        res['w_iou'].append(weighted_iou(gt, mi, classes))
        res['bin_iou'] = bin_iou(gt, pred)            # the reference keeps the LAST sample's value (:737)
    return res


def bb_iou(gt_mask, pred_mask, label):
    """get_bb_iou (lib/metrics/segmentation.py:32-68), quirks kept: the class's bounding box is filled with
    EXCLUSIVE upper bounds ([min:max], so the last row / column keeps only its own pixels), 0 when either
    mask lacks the class."""
    gt = (np.asarray(gt_mask) == label).astype(np.float64)
    pred = (np.asarray(pred_mask) == label).astype(np.float64)
    for m in (gt, pred):
        if m.max() == 0:
            return 0
        nz = m.nonzero()
        m[nz[0].min():nz[0].max(), nz[1].min():nz[1].max()] = 1
    union = np.count_nonzero(gt + pred)
    return np.count_nonzero(gt * pred) / union if union > 0 else 0


def dice_of_iou(iou):
    """get_mask_dice / get_bb_dice (:71-78)"""
    return 2 * iou / (iou + 1)


def pd_at_iou_threshold(iou_table, classes, thr=0.5):
    """get_pd_at_iou_threshold (:126-133): iou_table (samples, classes) -> fraction of samples above thr"""
    t = np.asarray(iou_table, np.float64)
    return {c: float((t[:, i] > thr).mean()) for i, c in enumerate(classes)}


def iou_vs_pd_curve(iou_table, classes, inc=0.05):
    """get_iou_vs_pd_curve (:136-162): int(1/inc) thresholds on linspace(0, 1), plus the class mean"""
    vals = np.linspace(0, 1, int(1 / inc))
    out = {c: [pd_at_iou_threshold(iou_table, classes, t)[c] for t in vals] for c in classes}
    out['Mean'] = np.mean([out[c] for c in classes], axis=0)
    return out
