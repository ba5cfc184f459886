"""One-shot fine-tune head on HIP (SURVEY.md par. 8f rank 1).

Mirrors the reference's fine-tune + test loop for the hfc_with_swav / hfc_with_simclr methods:
  * `OneShotSegmentor` (hfc_with_swav/swav_clustering.py:697-758): a stack of (dilated) 3x3 convs with
    LeakyReLU(0.2) between them — 'XXS' (ffhq, generic) is a single conv in_ch -> 12, 'XS' (cat, car,
    simclr) is in_ch -> 16 -> 8 -> n_class with dilations 1, 2, 1;
  * `OneShotPipeline.run_trainer` (src/one_shot_pipeline.py:533-581): Adam(lr, betas) on
    nn.CrossEntropyLoss of the head's output for every labelled sample, `num_epochs` epochs, StepLR per epoch;
  * `run_tests` (:595-886): argmax of the head over predict_swav_codes' code map, then
    lib/metrics/segmentation.py's per-class mask IoU, area-weighted IoU and binary IoU.

Everything runs on the GPU through the C ABI (csrc/gx_head.hip + gx_gemm_f32): activations are zero-padded
flat NHWC so that a dilated 3x3 tap is a row shift and conv forward / dW / dX are nine GEMMs each on
row-offset views of the same buffers; the 134 MB code map of a 256^2 image is read in place.
No CPU fallback: constructing the head without a HIP device raises.
"""
import os

import numpy as np
import torch
from torch import nn

from . import _lib, ops
from .ops import _p, _stream, lib, workspace
from .segmentor import _PLANS, OneShotSegmentor


def _plan(size, in_ch, n_class):
    """[(c_in, c_out, dilation, LeakyReLU after it, taps)]: 9-tap dilated convs, or — 'Lin' — ONE 1-tap layer (a per-pixel
    nn.Linear) that KEEPS its LeakyReLU (hfc_with_swav/swav_clustering.py:708-712)"""
    if size == 'Lin':
        return [(in_ch, n_class, 0, True, 1)]
    if size not in _PLANS:
        raise NotImplementedError(f"OneShotHead: unknown size {size!r}")
    dil, hidden = _PLANS[size]
    widths = [in_ch] + hidden + [n_class]
    plan = [[a, b, d, True, 9] for d, a, b in zip(dil, widths[:-1], widths[1:])]     # zip stops at len(dil) (reference)
    plan[-1][3] = False
    return [tuple(p) for p in plan]


class OneShotHead:
    """HIP twin of OneShotSegmentor: same plan, same `layers.{2i}.weight/bias` state_dict keys."""

    def __init__(self, in_ch, n_class, size='S', device='cuda'):
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError("OneShotHead runs on HIP devices only (no CPU fallback)")
        lib()
        self.plan = _plan(size, in_ch, n_class)
        self.size, self.in_ch, self.n_class = size, in_ch, n_class
        self.out_ch = self.plan[-1][1]
        self.P = max(p[2] for p in self.plan)
        self.load_module(OneShotSegmentor(in_ch, n_class, size))          # nn.Conv2d's default init
        self._step = 0
        self.stacked_taps = True       # layers with 9 * Cout <= 128: one stacked-tap GEMM + gx_head_tap_sum (False: nine GEMMs; A/B, tests)

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd):
        self.W, self.b = [], []
        for i, (cin, cout, _, _, ntap) in enumerate(self.plan):
            w = sd[f"layers.{2 * i}.weight"].detach().float()
            want = (cout, cin, 3, 3) if ntap == 9 else (cout, cin)
            if tuple(w.shape) != want:
                raise RuntimeError(f"layers.{2 * i}.weight: expected {want}, got {tuple(w.shape)}")
            # tap-major [taps][Cout][Cin]: one contiguous (Cout, Cin) GEMM operand per tap
            w = w.permute(2, 3, 0, 1).reshape(9, cout, cin) if ntap == 9 else w.reshape(1, cout, cin)
            self.W.append(w.contiguous().to(self.device))
            self.b.append(sd[f"layers.{2 * i}.bias"].detach().float().contiguous().to(self.device))
        self._m = self._v = None
        self._step = 0

    def load_module(self, module):
        self.load_state_dict(module.state_dict())

    def state_dict(self):
        sd = {}
        for i, (cin, cout, _, _, ntap) in enumerate(self.plan):
            sd[f"layers.{2 * i}.weight"] = (self.W[i].reshape(3, 3, cout, cin).permute(2, 3, 0, 1).contiguous() if ntap == 9
                                            else self.W[i].reshape(cout, cin).clone())
            sd[f"layers.{2 * i}.bias"] = self.b[i].clone()
        return sd

    def to_module(self):
        """a torch OneShotSegmentor carrying these weights (what the reference pickles / moves to CPU)"""
        m = OneShotSegmentor(self.in_ch, self.n_class, self.size)
        m.load_state_dict({k: v.cpu() for k, v in self.state_dict().items()})
        return m

    # ------------------------------------------------------------------ geometry
    G = 4     # zero guard rows before and after every flat buffer (see _range)

    def _geom(self, b, h, w):
        hp, wp = h + 2 * self.P, w + 2 * self.P
        rows = b * hp * wp
        s = self.P * wp + self.P                 # largest |row shift| of any tap of any layer
        return hp, wp, rows, s

    def _range(self, rows, s):
        """row range [lo, hi) the GEMMs run over: it covers every row whose taps stay in the buffer,
        [s, rows - s), widened to multiples of 4 rows (16-B vector loads of the m-contiguous GEMM
        operands); the few extra rows are border rows (masked to zero), read at most 3 guard rows and
        never reach past the buffer's guard rows (hi <= rows - s + 3: inside the buffer when s >= 3, at most 3 guard rows
        — zeros, G = 4 — past it for the 1-tap 'Lin' layer, whose s is 0)."""
        lo = s & ~3
        hi = lo + ((rows - s - lo + 3) & ~3)
        return lo, hi

    def _shift(self, tap, d, wp, ntap=9):
        return 0 if ntap == 1 else ((tap // 3) - 1) * d * wp + ((tap % 3) - 1) * d

    def _buf(self, rows, c):
        """zeroed flat buffer with guard rows; returns (storage, logical view of `rows` rows)"""
        st = torch.zeros(rows + 2 * self.G, c, dtype=torch.float32, device=self.device)
        return st, st[self.G:self.G + rows]

    def pad(self, feats):
        """feats (B, H, W, C) NHWC (a view of the projection table is fine) -> (storage, padded flat view)"""
        feats = feats.contiguous()
        b, h, w, c = feats.shape
        hp, wp, rows, _ = self._geom(b, h, w)
        st, out = self._buf(rows, c)
        _lib.check(lib().gx_head_pad(_p(feats), b, h, w, c, self.P, _p(out), _stream()), "gx_head_pad")
        return st, out

    # ------------------------------------------------------------------ forward / backward
    def _forward(self, xst, b, h, w):
        """xst = padded input storage -> list of (storage, view) of every layer's activation (last = logits)"""
        hp, wp, rows, s = self._geom(b, h, w)
        lo, hi = self._range(rows, s)
        g = self.G
        acts, x = [], xst
        for i, (cin, cout, d, act, ntap) in enumerate(self.plan):
            yst, y = self._buf(rows, cout)
            if ntap == 9 and 9 * cout <= 128 and self.stacked_taps:
                # few output channels (the 'XXS' head: 512 -> 12): nine GEMMs would each fill 12 of a tile's 128 columns.  ONE
                # GEMM over the un-shifted rows with the taps stacked along N (108 columns), then the tap sum + bias +
                # activation + border mask in one pass (gx_head_tap_sum): the same sums in the same order
                y9 = ops.gemm(x, self.W[i].reshape(9 * cout, cin), False, True)
                _lib.check(lib().gx_head_tap_sum(_p(y9[g:]), y9.stride(0), b, h, w, cout, self.P, d, _p(self.b[i]), int(act),
                                                 _p(y), _stream()), "gx_head_tap_sum")
                acts.append((yst, y))
                x = yst
                continue
            for tap in range(ntap):
                sh = self._shift(tap, d, wp, ntap)
                ops.gemm(x[g + lo + sh:g + hi + sh], self.W[i][tap], False, True, out=yst[g + lo:g + hi],
                         beta=0.0 if tap == 0 else 1.0, col_bias=self.b[i] if tap == 0 else None)
            _lib.check(lib().gx_head_act_mask(_p(y), b, h, w, cout, self.P, int(act), _stream()), "gx_head_act_mask")
            acts.append((yst, y))
            x = yst
        return acts

    def _backward(self, xst, acts, dlst, b, h, w):
        """gradients [(dW (9,Cout,Cin), db (Cout))] of every layer from d loss / d logits (storage)"""
        hp, wp, rows, s = self._geom(b, h, w)
        lo, hi = self._range(rows, s)
        g = self.G
        grads = [None] * len(self.plan)
        dyst = dlst
        for i in reversed(range(len(self.plan))):
            cin, cout, d, act, ntap = self.plan[i]
            x = xst if i == 0 else acts[i - 1][0]
            dy = dyst[g:g + rows]
            if act:                              # through the LeakyReLU + border mask that follow layer i
                _lib.check(lib().gx_head_act_mask_bwd(_p(dy), _p(acts[i][1]), b, h, w, cout, self.P, int(act), _stream()),
                           "gx_head_act_mask_bwd")
            dw = torch.empty(ntap, cout, cin, dtype=torch.float32, device=self.device)
            split = int(max(1, min(256, (hi - lo) // 2048)))
            for tap in range(ntap):
                sh = self._shift(tap, d, wp, ntap)
                ops.gemm(dyst[g + lo:g + hi], x[g + lo + sh:g + hi + sh], True, False, out=dw[tap], split_k=split)
            db = torch.empty(cout, dtype=torch.float32, device=self.device)
            ops.colsum(dy, db)
            grads[i] = (dw, db)
            if i > 0:
                dxst, _ = self._buf(rows, cin)
                for tap in range(ntap):
                    sh = self._shift(tap, d, wp, ntap)
                    ops.gemm(dyst[g + lo - sh:g + hi - sh], self.W[i][tap], False, False, out=dxst[g + lo:g + hi],
                             beta=0.0 if tap == 0 else 1.0)
                dyst = dxst
        return grads

    def _ce(self, logits, labels, b, h, w, want_grad):
        """mean CE of the padded logits view; returns (loss (1,), storage of d loss / d logits or None)"""
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        dlst = dl = None
        if want_grad:
            dlst, dl = self._buf(logits.shape[0], logits.shape[1])
        ws = workspace("head_ce", 4096, self.device)
        _lib.check(lib().gx_head_softmax_ce(_p(logits), _p(labels), b, h, w, self.out_ch, self.P, _p(dl), _p(loss), _p(ws),
                                            4096, _stream()), "gx_head_softmax_ce")
        return loss, dlst

    def _adam(self, grads, lr, beta1, beta2, eps=1e-8):
        if self._m is None:
            self._m = [(torch.zeros_like(w), torch.zeros_like(b)) for w, b in zip(self.W, self.b)]
            self._v = [(torch.zeros_like(w), torch.zeros_like(b)) for w, b in zip(self.W, self.b)]
        self._step += 1
        for i, (dw, db) in enumerate(grads):
            for p, g, m, v in ((self.W[i], dw, self._m[i][0], self._v[i][0]), (self.b[i], db, self._m[i][1], self._v[i][1])):
                _lib.check(lib().gx_head_adam(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, self._step,
                                              _stream()), "gx_head_adam")

    # ------------------------------------------------------------------ reference-shaped API
    def fit(self, feats, labels, num_epochs, lr, beta1, beta2, step_size=500, gamma=0.1, orders=None, print_freq=0,
            logger=None):
        """run_trainer: feats (n, H, W, C) NHWC code maps of the labelled samples, labels (n, H, W) integer.
        Returns the last loss of every epoch (HIP scalars are read back once, at the end)."""
        n, h, w, _ = feats.shape
        labels = labels.to(self.device, torch.int64).contiguous()
        xps = [self.pad(feats[i:i + 1])[0] for i in range(n)]         # the features are fixed: pad once
        losses = []
        for e in range(num_epochs):
            cur_lr = lr * gamma ** (e // step_size)                   # StepLR.step() once per epoch (:578)
            order = list(range(n)) if orders is None else orders[e]
            for idx in order:
                acts = self._forward(xps[idx], 1, h, w)
                loss, dlst = self._ce(acts[-1][1], labels[idx:idx + 1], 1, h, w, True)
                self._adam(self._backward(xps[idx], acts, dlst, 1, h, w), cur_lr, beta1, beta2)
            losses.append(loss)
            if logger is not None and print_freq and e % print_freq == 0:
                logger.info(f'{e:5}-th epoch | loss: {loss.item():6.4f}')
        return [float(x) for x in torch.cat(losses).cpu()]

    def logits(self, feats):
        """feats (B, H, W, C) NHWC -> logits (B, out_ch, H, W) like OneShotSegmentor.forward"""
        b, h, w, _ = feats.shape
        out = self._forward(self.pad(feats)[0], b, h, w)[-1][1]
        hp, wp = h + 2 * self.P, w + 2 * self.P
        return out.reshape(b, hp, wp, self.out_ch)[:, self.P:self.P + h, self.P:self.P + w].permute(0, 3, 1, 2)

    def predict(self, feats):
        """pred (B, H, W) int64 = segmentor(features).max(1)[1]  (run_tests :664-665)"""
        b, h, w, _ = feats.shape
        out = self._forward(self.pad(feats)[0], b, h, w)[-1][1]
        pred = torch.empty(b, h, w, dtype=torch.int64, device=self.device)
        _lib.check(lib().gx_head_argmax(_p(out), b, h, w, self.out_ch, self.P, _p(pred), _stream()), "gx_head_argmax")
        return pred


# --------------------------------------------------------------------------------- metrics (run_tests)
def confusion(gt, pred, n_class):
    """conf[g][p] (n_class x n_class, int64 on the host) for HIP label maps of equal shape"""
    gt = gt.to(pred.device, torch.int64).contiguous()
    pred = pred.contiguous()
    conf = torch.empty(n_class, n_class, dtype=torch.int64, device=pred.device)
    _lib.check(lib().gx_confusion(_p(gt), _p(pred), gt.numel(), n_class, _p(conf), _stream()), "gx_confusion")
    return conf.cpu().numpy()


def iou_from_confusion(conf, classes):
    """lib/metrics/segmentation.py get_mask_iou (:9-29) for every class, get_weighted_iou (:94-108) and
    get_bin_iou (:111-125) from one confusion matrix of a (gt, pred) pair."""
    conf = np.asarray(conf, dtype=np.float64)
    area = conf.sum()
    mask_iou = {}
    for i, c in enumerate(classes):
        union = conf[i, :].sum() + conf[:, i].sum() - conf[i, i]
        mask_iou[c] = conf[i, i] / union if union > 0 else 0
    w_iou = sum(conf[i, :].sum() / area * mask_iou[c] for i, c in enumerate(classes) if c != 'background')
    fg_inter = conf[1:, 1:].sum()
    fg_union = area - conf[0, 0]
    return mask_iou, w_iou, (fg_inter / fg_union if fg_union > 0 else 0)


def bb_iou(gt_mask, pred_mask, label):
    """lib/metrics/segmentation.py get_bb_iou (:32-68) on host label maps: IoU of the class's bounding boxes,
    filled with the reference's exclusive upper bounds; 0 when either map lacks the class."""
    gt = (np.asarray(gt_mask) == label).astype(np.float64)
    pred = (np.asarray(pred_mask) == label).astype(np.float64)
    for m in (gt, pred):
        if m.max() == 0:
            return 0
        nz = m.nonzero()
        m[nz[0].min():nz[0].max(), nz[1].min():nz[1].max()] = 1
    union = np.count_nonzero(gt + pred)
    return np.count_nonzero(gt * pred) / union if union > 0 else 0


def pd_at_iou_threshold(iou_table, classes, thr=0.5):
    """get_pd_at_iou_threshold (:126-133) on a (samples, classes) IoU table"""
    t = np.asarray(iou_table, np.float64)
    return {c: float((t[:, i] > thr).mean()) for i, c in enumerate(classes)}


def iou_vs_pd_curve(iou_table, classes, inc=0.05):
    """get_iou_vs_pd_curve (:136-162)"""
    vals = np.linspace(0, 1, int(1 / inc))
    out = {c: [pd_at_iou_threshold(iou_table, classes, t)[c] for t in vals] for c in classes}
    out['Mean'] = np.mean([out[c] for c in classes], axis=0)
    return out


def table_csv(rows, classes):
    """The text pandas.DataFrame(data, columns=classes).to_csv() writes (src/one_shot_pipeline.py:776-791):
    an unnamed index column, then one column per class; floats in repr form."""
    lines = ["," + ",".join(classes)]
    for i, r in enumerate(rows):
        lines.append(",".join([str(i)] + [repr(float(r[c])) for c in classes]))
    return "\n".join(lines) + "\n"


def write_result_files(test_dir, results, classes):
    """tests/mask_iou_results.csv, tests/bb_iou_results.csv and tests/results.npz exactly as run_tests leaves
    them (src/one_shot_pipeline.py:775-791, :884: np.savez_compressed(**results), the per-sample dict lists
    as object arrays)."""
    with open(os.path.join(test_dir, 'mask_iou_results.csv'), 'w') as f:
        f.write(table_csv(results['mask_iou'], classes))
    with open(os.path.join(test_dir, 'bb_iou_results.csv'), 'w') as f:
        f.write(table_csv(results['bb_iou'], classes))
    np.savez_compressed(os.path.join(test_dir, 'results.npz'), **results)


def run_tests(preprocessor, head, test_latents, test_labels, classes, batch=16):
    """The numeric core of OneShotPipeline.run_tests (:611-748, tester 'all'): per batch of test latents
    predict_swav_codes -> head -> argmax; then per sample the IoU / Dice block, and the PD-vs-IoU summaries.
    test_labels (n, H, W).  Returns (pred_labels (n, H, W) int64 on the host, results dict with the
    reference's keys).  batch: latents per predict call (the reference: 1, MAX_TEST_BATCH :30 — per-pixel
    preprocessors give the same codes at any batch; SimCLR's train-mode BatchNorm does not, so its caller
    passes 1)."""
    preds = []
    for s in range(0, test_latents.shape[0], batch):
        codes, _ = preprocessor.predict_swav_codes(test_latents[s:s + batch])
        preds.append(head.predict(codes.permute(0, 2, 3, 1)))
    pred = torch.cat(preds)
    nconf = max(len(classes), head.out_ch)
    results = dict(mask_iou=[], bb_iou=[], w_iou=[], bin_iou=None, mask_dice=[], bb_dice=[])
    pred_h = pred.cpu()
    gt_h = test_labels.cpu().numpy() if isinstance(test_labels, torch.Tensor) else np.asarray(test_labels)
    for i in range(pred.shape[0]):
        conf = confusion(test_labels[i], pred[i], nconf)
        mi, wi, bi = iou_from_confusion(conf, list(classes) + [f"_{j}" for j in range(len(classes), nconf)])
        results['mask_iou'].append({c: mi[c] for c in classes})
        results['w_iou'].append(wi)
        results['bin_iou'] = bi                     # the reference keeps the last sample's value (:737)
        bb = {c: bb_iou(gt_h[i], pred_h[i].numpy(), k) for k, c in enumerate(classes)}
        results['bb_iou'].append(bb)
        results['mask_dice'].append({c: 2 * mi[c] / (mi[c] + 1) for c in classes})
        results['bb_dice'].append({c: 2 * bb[c] / (bb[c] + 1) for c in classes})
    table = [[m[c] for c in classes] for m in results['mask_iou']]
    results['pd'] = pd_at_iou_threshold(table, classes, 0.5)
    results['iou_pd_curve'] = iou_vs_pd_curve(table, classes)
    return pred_h, results
