"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU fp32 restatement of the SimCLR baseline loss path (BASELINE.json config 5;
baseline/hfc_with_simclr/simclr_clustering.py of avm-debatr/ganecdotes).  Pinned by
tests/golden/gv12_simclr.npz (two iterations of the reference's own pretrain()).
The reference's quirks are reproduced on purpose:
  * `scores = projection(x).t()` is (D, 2b) and the loops index `scores[i]`, so the
    "similarity" is taken between the first 2b FEATURE ROWS (each a 2b-vector over
    the samples), not between samples (simclr_clustering.py:230-242);
  * the positive pairs are (2k-1, 2k) and (2k, 2k-1), k = 0 wrapping to index -1 (:261-263).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import stylegan2_ref as G
from . import swav_ref as S


def projection_mlp(x, w1, gamma, beta, w2, eps=1e-5):
    """Linear(no bias) -> BatchNorm1d in TRAIN mode (batch statistics, biased variance) ->
    LeakyReLU(0.01) -> Linear(no bias)   (simclr_clustering.py:147-158).  Returns (out, mean, var)."""
    h1 = F.linear(x, w1)
    mean = h1.mean(0)
    var = h1.var(0, unbiased=False)
    a = F.leaky_relu((h1 - mean) * torch.rsqrt(var + eps) * gamma + beta, 0.01)
    return F.linear(a, w2), mean, h1.var(0, unbiased=True)


def ntxent_feature_rows(out, batch, temperature):
    """The reference's inline loss (simclr_clustering.py:224-264) on out (2b, D) = projection of
    the interleaved samples [s0, t0, s1, t1, ...]."""
    n = 2 * batch
    rows = out.t()[:n]                                            # scores[i], i < 2b: feature rows
    nrm = rows.norm(dim=1)
    sim = (rows @ rows.t()) / torch.clamp(nrm[:, None] * nrm[None, :], min=1e-8) / temperature
    e = torch.exp(sim)
    den = e.sum(1) - e.diagonal()                                 # sum over m != i
    loss_m = -torch.log(e / den[:, None])
    tot = 0.0
    for k in range(batch):
        tot = tot + loss_m[2 * k - 1, 2 * k] + loss_m[2 * k, 2 * k - 1]
    return tot / n


def sample_pair(stack_s, stack_t, picks, batch):
    """per-pixel F.normalize(dim=1) of both stacks, first `batch` picked pixels, interleaved
    (simclr_clustering.py:191-222) -> (2b, C)"""
    xs = F.normalize(stack_s, dim=1).squeeze(0).flatten(1)[:, picks][:, :batch]
    xt = F.normalize(stack_t, dim=1).squeeze(0).flatten(1)[:, picks][:, :batch]
    x = torch.zeros(xs.shape[0], 2 * batch)
    x[:, ::2] = xs
    x[:, 1::2] = xt
    return x.t()


def draw_iteration_reference_order(style_dim, gen_num_layers, n_perturb_layers, hw, transforms=False,
                                   degrees=10.0):
    """RNG order of one SimCLR iteration: z; view s draws (+ its angle/flip); view t draws (+ its
    angle/flip); one randperm (simclr_clustering.py:177-207)."""
    z = torch.randn(1, style_dim)
    views = []
    for _ in range(2):
        for i in range(gen_num_layers):
            r = 2 ** ((i + 5) // 2)
            torch.empty(1, 1, r, r).normal_()
        lno = int(np.random.choice(list(range(n_perturb_layers))))
        maps = [torch.randn(1, style_dim) for _ in range(2 * n_perturb_layers)]
        vd = S.ViewDraws(lno, torch.cat([maps[2 * lno], maps[2 * lno + 1]], 0))
        if transforms:
            vd.angle = float(torch.empty(1).uniform_(-degrees, degrees).item())
            vd.flip = bool(torch.rand(1) < 0.5)
        views.append(vd)
    return S.ImageDraws(z, views, [torch.randperm(hw)])


def simclr_step_loss(sd, mean_latent, psi, im, w1, gamma, beta, w2, *, hlen, batch, temperature, hf_interp="nearest"):
    with torch.no_grad():
        w = G.mapping(im.z, sd)
        stacks = [S.make_view(sd, w, mean_latent, psi, vd, hlen, hf_interp=hf_interp)[0] for vd in im.views]
        x = sample_pair(stacks[0], stacks[1], im.picks[0], batch)
    out, mean, var_unbiased = projection_mlp(x, w1, gamma, beta, w2)
    return ntxent_feature_rows(out, batch, temperature), mean.detach(), var_unbiased.detach()


def pretrain_reference_order(sd, style_dim, *, num_latents_for_mean, psi, n_perturb_layers, num_iters, batch,
                             hlen, nclasses, temperature, lr, momentum, trust, use_larc, transforms):
    """constructor + pretrain() of SimCLRClustering in the reference's RNG order.
    Returns (losses, params dict incl. BatchNorm running stats, mean_latent)."""
    nl = G.num_layers_of(sd)
    with torch.no_grad():
        mean_lat = G.mean_latent(sd, num_latents_for_mean, style_dim)
    w1, _ = S.init_linear_like_torch(nclasses, hlen, False)
    gamma, beta = torch.ones(nclasses), torch.zeros(nclasses)
    w2, _ = S.init_linear_like_torch(nclasses, nclasses, False)
    run_mean, run_var = torch.zeros(nclasses), torch.ones(nclasses)
    leaves = [w1, gamma, beta, w2]
    bufs = [None] * 4
    size = 2 ** ((nl + 3) // 2)
    losses = []
    for _ in range(num_iters):
        im = draw_iteration_reference_order(style_dim, nl, n_perturb_layers, size * size, transforms)
        for p in leaves:
            p.requires_grad_(True)
            p.grad = None
        loss, mean, var_u = simclr_step_loss(sd, mean_lat, psi, im, *leaves, hlen=hlen, batch=batch,
                                             temperature=temperature)
        loss.backward()
        grads = [p.grad.detach().clone() for p in leaves]
        for p in leaves:
            p.requires_grad_(False)
        with torch.no_grad():
            run_mean = 0.9 * run_mean + 0.1 * mean
            run_var = 0.9 * run_var + 0.1 * var_u
            bufs = S.larc_sgd_step(leaves, grads, bufs, lr, momentum, trust, use_larc)
        losses.append(float(loss.detach()))
    return losses, dict(w1=w1, gamma=gamma, beta=beta, w2=w2, running_mean=run_mean, running_var=run_var), mean_lat


def predict_simclr_codes(sd, w, mean_latent, psi, params, hlen, hf_interp="nearest"):
    """predict_simclr_codes (simclr_clustering.py:365-400): the projection module is still in
    train mode, so BatchNorm uses the statistics of the image's own H*W pixels."""
    with torch.no_grad():
        _, taps, _ = G.generator_forward(w, sd, truncation=psi, truncation_latent=mean_latent,
                                         input_is_latent=True)
        stack = F.normalize(S.pixel_features(taps, hlen, hf_interp), dim=1)
        b, c, h, wd = stack.shape
        x = stack.squeeze(0).flatten(1).t()
        out, _, _ = projection_mlp(x, params["w1"], params["gamma"], params["beta"], params["w2"])
        preds = out.t().reshape(b, params["w2"].shape[0], h, wd)
    return preds, preds.max(1)[1]
