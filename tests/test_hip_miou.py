"""Segmentation-agreement half of the metric (SURVEY.md par. 8d): no ground-truth labels exist
offline, so "seg mIoU vs ref" is the mean over classes of the mask IoU (same definition as the
reference's lib/metrics/segmentation.py:9-29: |A & B| / |A | B| per class) between the HIP
cluster map and the CPU-oracle cluster map on the same seeds; target >= 99.5 %."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def mean_mask_iou(a, b, classes):
    ious = []
    for c in classes:
        ma, mb = (a == c), (b == c)
        union = np.logical_or(ma, mb).sum()
        if union:
            ious.append(np.logical_and(ma, mb).sum() / union)
    return float(np.mean(ious))


def test_cluster_map_miou_hip_vs_oracle():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from ganecdotes_amd.generator import Generator
    from ganecdotes_amd.swav_clustering import SwAVClustering
    from oracle import stylegan2_ref as G
    from oracle import swav_ref as S
    torch.manual_seed(13)
    np.random.seed(13)
    g = Generator(64, 512, 8)
    for n, p in g.named_parameters():
        if n.endswith("noise.weight"):
            p.data.fill_(0.1)
    sd = {k: v.detach().clone() for k, v in g.state_dict().items()}
    g = g.eval().cuda()

    class C:
        num_latents_for_mean, truncation, latent_dim = 128, 0.7, 512
    hlen, d = 512 * 9, 64
    sw = SwAVClustering(model=g, model_config=C, out_dir=None, device="cuda",
                        perturb_args=dict(truncation=0.7, n_layers=4, n_samples=1, layer_no=None,
                                          perturb_std=[1.0] * 4),
                        swav_args=dict(nclasses=d, nprototypes=40, hlen=hlen, projn_nw="linear", patch_size=1500,
                                       num_patches=2, temperature=0.1, train_args=dict(lr=0.01, momentum=0.9),
                                       trust_coeff=0.01),
                        sinkhorn_args=dict(source_pdf="uniform", niters=10, eps=0.05))
    ml = G.mean_latent(sd, 128, 512)
    sw.mean_latent = ml.cuda()
    sw.build_heads()
    wp = sw.projection[0].weight.data.cpu()
    scores = []
    for i in range(3):
        w = G.mapping(torch.randn(1, 512), sd)
        preds, labels = sw.predict_swav_codes(w.cuda())
        ref_preds, ref_labels = S.predict_swav_codes(sd, w, ml, 0.7, wp, hlen)
        a, b = labels.cpu().numpy()[0], ref_labels.numpy()[0]
        scores.append(mean_mask_iou(a, b, np.union1d(np.unique(a), np.unique(b))))
        assert (a == b).mean() > 0.995
    assert np.mean(scores) >= 0.995, scores


def test_cluster_map_miou_after_a_ten_step_trajectory_hip_vs_oracle():
    """The mIoU half of the metric after TRAINING, not only on untrained heads: ten optimizer steps
    of the HIP path and ten of the oracle on the same draws (64^2 generator at ffhq width, 9 taps x 512 channels, 40
    prototypes, 2 patches of 1500 of 4096 pixels, rotate / flip on) — float atomics and the bf16x6 contraction order accumulate
    over a trajectory — then the cluster maps of three fresh latents: mean mask IoU >= 0.995 (north_star: within 0.5 pt), the loss
    trajectory to 1e-3 and the trained projection to 1e-3."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from ganecdotes_amd import ops
    from ganecdotes_amd.draws import draw_image
    from ganecdotes_amd.generator import Generator
    from ganecdotes_amd.swav_clustering import SwAVClustering
    from oracle import stylegan2_ref as G
    from oracle import swav_ref as S
    torch.set_num_threads(min(16, torch.get_num_threads()))
    torch.manual_seed(14)
    np.random.seed(14)
    g = Generator(64, 512, 8)
    for n, p in g.named_parameters():
        if n.endswith("noise.weight"):
            p.data.fill_(0.1)
    sd = {k: v.detach().clone() for k, v in g.state_dict().items()}
    g = g.eval().cuda()

    class C:
        num_latents_for_mean, truncation, latent_dim = 128, 0.7, 512
    hlen, d, K, N, P = 512 * 9, 64, 40, 1500, 2
    sw = SwAVClustering(model=g, model_config=C, out_dir=None, device="cuda",
                        perturb_args=dict(truncation=0.7, n_layers=4, n_samples=1, layer_no=None, perturb_std=[1.0] * 4),
                        swav_args=dict(nclasses=d, nprototypes=K, hlen=hlen, projn_nw="linear", patch_size=N, num_patches=P,
                                       temperature=0.1, train_args=dict(lr=0.01, momentum=0.9), trust_coeff=0.01,
                                       sinkhorn_scope="global", batch_size=2),
                        sinkhorn_args=dict(source_pdf="uniform", niters=10, eps=0.05))
    ml = G.mean_latent(sd, 128, 512)
    sw.mean_latent = ml.cuda()
    sw.build_heads()
    params = S.SwavParams(*[p.detach().cpu().clone() for p in sw._params()])
    bufs = [None] * 3
    for step in range(10):
        images = [draw_image(512, 9, 4, 64 * 64, P, transforms=True) for _ in range(2)]
        loss = sw.train_step(images)
        oimgs = [S.ImageDraws(im.z, [S.ViewDraws(v.layer_no, v.map_noise, v.angle, v.flip,
                                                 ops.rotflip_map(v.angle, v.flip, 64, 64, "cuda").cpu().long())
                                     for v in im.views], im.picks) for im in images]
        ref_loss, _, bufs = S.swav_train_step(sd, ml, 0.7, oimgs, params, bufs, lr=0.01, momentum=0.9, trust=0.01, use_larc=True,
                                              hlen=hlen, patch_size=N, num_patches=P, temperature=0.1, eps=0.05, niters=10,
                                              source_pdf="uniform", sinkhorn_scope="global")
        np.testing.assert_allclose(loss.item(), ref_loss, rtol=1e-3)
    wp = sw.projection[0].weight.data.cpu()
    assert float((wp - params.proj_w).abs().max() / params.proj_w.abs().max()) < 1e-3
    scores = []
    for i in range(3):
        w = G.mapping(torch.randn(1, 512), sd)
        _, labels = sw.predict_swav_codes(w.cuda())
        _, ref_labels = S.predict_swav_codes(sd, w, ml, 0.7, params.proj_w, hlen)       # the ORACLE's trained projection
        a, b = labels.cpu().numpy()[0], ref_labels.numpy()[0]
        scores.append(mean_mask_iou(a, b, np.union1d(np.unique(a), np.unique(b))))
    assert np.mean(scores) >= 0.995, scores
