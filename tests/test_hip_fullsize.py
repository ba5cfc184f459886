"""BASELINE.json's configurations at their FULL sizes (256^2 x 5376 channels x 20 000 pixels;
512^2 for car) on a real MI355X:

  * the cat-256 (K = 16, 'image' source pdf, eps 0.003), car-512 (train step at 512^2, K = 16 and
    4000, eps 0.01) and SimCLR (ffhq width, 20 pixel pairs) configurations: one optimizer step each
    against the CPU oracle on the same draws;
  * one whole optimizer step of the ffhq-256 / K=8 workload against the CPU oracle on the same
    draws (one image, one patch: what the oracle finishes in seconds on the box's host cores);
  * the car-512 generator + all-pixel projection + cluster map against the oracle;
  * size-independent properties at the bench batch (16 images): the pyramid projection and its
    backward are exact adjoints, the Sinkhorn codes carry the prescribed sample marginal, the
    generator is batch-independent, predict is idempotent and the step is reproducible.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from ganecdotes_amd import ops
    ops.lib()
    return torch.device("cuda")


# north_star asks for 1e-3 relative in fp32; the step measures ~3e-7 on the loss and ~2e-6 (max-norm) on the gradients
# against the oracle on either engine (tools/lab/step_errors.py), so the full-size tests hold it to 1e-4
LOSS_TOL = 1e-4
GRAD_TOL = 1e-4


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def full_generator(size, seed=42):
    from ganecdotes_amd.generator import Generator
    torch.manual_seed(seed)
    g = Generator(size, 512, 8)
    for n, p in g.named_parameters():
        if n.endswith("noise.weight"):
            p.data.fill_(0.1)            # the checkpoint-free init is 0: exercise the noise path
    sd = {k: v.detach().clone() for k, v in g.state_dict().items()}
    return g.eval(), sd


class Cfg:
    num_latents_for_mean, truncation, latent_dim = 128, 0.7, 512


def ffhq_args(batch, nproto, num_patches=5, pick_source="device", **over):
    swav = dict(num_epochs=1, num_samples=1, num_patches=num_patches, sampling_method="random", patch_size=20000,
                hf_interp="nearest", warmup_epochs=1, start_warmup=0.01, use_scheduler=False, base_lr=0.01,
                final_lr=0.0001, trust_coeff=0.01, train_args=dict(lr=0.01, momentum=0.9), projn_nw="linear",
                temperature=0.01, nprototypes=nproto, nclasses=512, hlen=5376, add_local_loss=False,
                plot_test_images=False, epoch_print_freq=5, max_masks=4, batch_size=batch,
                sinkhorn_scope="global", pick_source=pick_source)
    swav.update(over)
    return dict(perturb_args=dict(truncation=0.7, n_layers=6, n_samples=1, layer_no=None, perturb_std=[1.0] * 6),
                swav_args=swav, sinkhorn_args=dict(source_pdf="uniform", niters=10, eps=0.005))


# --------------------------------------------------------------------------- full-size step vs oracle
@pytest.fixture(params=[1, 0], ids=["bf16x6", "f32mfma"])
def engine(request, gpu):
    """both contraction engines (GEMMs and modulated convs): bf16x6 (default) and exact-fp32 MFMA"""
    from ganecdotes_amd import ops
    prev = ops.set_gemm_mode(request.param)
    yield request.param
    ops.set_gemm_mode(prev)


@pytest.mark.parametrize("nproto", [8, 8000])
def test_ffhq256_full_size_step_vs_oracle(gpu, engine, nproto):
    """ffhq-256, 13 taps / 5376 channels, 20 000 sampled pixels, D=512, K=8, T=0.01, eps=0.005
    (configs[1] at batch 1, one patch): loss, the three gradients and the LARC update against the
    oracle on identical draws — on either contraction engine.  K = 8000: the generic `hfc_with_swav` configuration's own
    prototype count (configs/segmentors/hfc_with_swav_config.py), on the 8192-column forms of the row kernels and the per-pixel
    tables like the bench step."""
    from oracle import swav_ref as S
    from ganecdotes_amd import ops
    from ganecdotes_amd.draws import draw_image
    from ganecdotes_amd.swav_clustering import SwAVClustering
    if nproto > 8 and ops.get_gemm_mode() == 0:
        pytest.skip("the 8000-prototype case runs on the default engine only (suite time)")
    torch.set_num_threads(min(16, torch.get_num_threads()))
    g, sd = full_generator(256)
    g = g.to(gpu)
    sw = SwAVClustering(model=g, model_config=Cfg, out_dir=None, device="cuda",
                        **ffhq_args(1, nproto, num_patches=1, pick_source="host", **(dict(ds_table=True) if nproto > 64 else {})))
    torch.manual_seed(5)
    np.random.seed(5)
    with torch.no_grad():
        ml = S.G.mean_latent(sd, 128, 512)
    sw.mean_latent = ml.cuda()
    sw.build_heads()
    params = S.SwavParams(*[p.detach().cpu().clone() for p in sw._params()])
    images = [draw_image(512, 13, 6, 256 * 256, 1, transforms=True)]
    loss = sw.train_step(images)
    oimgs = [S.ImageDraws(im.z, [S.ViewDraws(v.layer_no, v.map_noise, v.angle, v.flip,
                                             ops.rotflip_map(v.angle, v.flip, 256, 256, "cuda").cpu().long())
                                 for v in im.views], im.picks) for im in images]
    ref_loss, grads, _ = S.swav_train_step(
        sd, ml, 0.7, oimgs, params, [None] * 3, lr=0.01, momentum=0.9, trust=0.01, use_larc=True, hlen=5376,
        patch_size=20000, num_patches=1, temperature=0.01, eps=0.005, niters=10, source_pdf="uniform",
        sinkhorn_scope="global")
    np.testing.assert_allclose(loss.item(), ref_loss, rtol=LOSS_TOL)
    for got, want in zip(sw._grads, grads):
        assert rel(got, want) < GRAD_TOL
    for got, want in zip(sw._params(), (params.proj_w, params.proto_w, params.proto_b)):
        assert rel(got, want) < 1e-5



@pytest.mark.parametrize("nproto,patches", [(5000, 3), (4001, 1)])
def test_ffhq256_table_sinkhorn_overlapping_patches_vs_oracle(gpu, nproto, patches):
    """What the table Sinkhorn (csrc/gx_sktab.hip) does that nothing else does — ONE pass over rows that SEVERAL
    patches drew — against the oracle, directly, at full size and with the configuration's real eps = 0.005 / T = 0.01
    (VERDICT round 4, item 4): ffhq-256 width, 1 image, K = 5000, 3 patches of 20 000 of the 65 536 pixels (60 % of the drawn pixels
    are drawn twice or more), scores and d(scores) on per-pixel tables.  Loss, the three gradients and the LARC update.
    (4001, 1 patch): a prototype count that is not a multiple of 4 — padded to 4004 columns that take no mass
    (SwAVClustering._pad_prototypes; item 5) — against the oracle's unpadded problem.  Default engine only (oracle time)."""
    from oracle import swav_ref as S
    from ganecdotes_amd import ops
    from ganecdotes_amd.draws import draw_image
    from ganecdotes_amd.swav_clustering import SwAVClustering
    torch.set_num_threads(min(16, torch.get_num_threads()))
    g, sd = full_generator(256)
    g = g.to(gpu)
    sw = SwAVClustering(model=g, model_config=Cfg, out_dir=None, device="cuda",
                        **ffhq_args(1, nproto, num_patches=patches, pick_source="host", ds_table=True, score_table=True,
                                    sinkhorn_table=True))
    torch.manual_seed(6)
    np.random.seed(6)
    with torch.no_grad():
        ml = S.G.mean_latent(sd, 128, 512)
    sw.mean_latent = ml.cuda()
    sw.build_heads()
    params = S.SwavParams(*[p.detach().cpu().clone() for p in sw._params()])
    images = [draw_image(512, 13, 6, 256 * 256, patches, transforms=True)]
    if patches > 1:       # the patches do overlap: pixels drawn by two or more of them
        cnt = torch.zeros(256 * 256, dtype=torch.int32)
        for pk in images[0].picks:
            cnt[pk[:20000].long()] += 1
        assert int((cnt >= 2).sum()) > 0.3 * int((cnt >= 1).sum())
    loss = sw.train_step(images)
    rt = sw.last_route
    assert rt.k_run == (nproto + 3) // 4 * 4, rt
    if ops.get_gemm_mode() == 1:     # (the per-pixel tables need the weight-plane GEMMs of the default engine; GX_GEMM_MODE=0 runs per sample)
        assert rt.name == "tables" and rt.score_table and rt.sinkhorn_table, rt
    oimgs = [S.ImageDraws(im.z, [S.ViewDraws(v.layer_no, v.map_noise, v.angle, v.flip,
                                             ops.rotflip_map(v.angle, v.flip, 256, 256, "cuda").cpu().long())
                                 for v in im.views], im.picks) for im in images]
    ref_loss, grads, _ = S.swav_train_step(
        sd, ml, 0.7, oimgs, params, [None] * 3, lr=0.01, momentum=0.9, trust=0.01, use_larc=True, hlen=5376,
        patch_size=20000, num_patches=patches, temperature=0.01, eps=0.005, niters=10, source_pdf="uniform",
        sinkhorn_scope="global")
    np.testing.assert_allclose(loss.item(), ref_loss, rtol=LOSS_TOL)
    for got, want in zip(sw._grads, grads):
        assert got.shape == want.shape and rel(got, want) < GRAD_TOL
    for got, want in zip(sw._params(), (params.proj_w, params.proto_w, params.proto_b)):
        assert rel(got, want) < 1e-5


@pytest.mark.parametrize("nproto", [8, 5000])
def test_ffhq256_batch2_two_patches_step_vs_oracle(gpu, engine, nproto):
    """the batched form of the step against the oracle, not only its properties: 2 images x 2 patches x 2 views of
    20 000 pixels in ONE global Sinkhorn problem per (patch, view) (40 000 rows), gradients averaged over images and
    patches — BASELINE.json's 8 prototypes (quadratic-form finest level, batched gather / loss / scatter) and the
    reference's own 5000 (Sinkhorn first pass out of the score GEMM's epilogue)."""
    from oracle import swav_ref as S
    from ganecdotes_amd import ops
    from ganecdotes_amd.draws import draw_image
    from ganecdotes_amd.swav_clustering import SwAVClustering
    torch.set_num_threads(min(16, torch.get_num_threads()))
    g, sd = full_generator(256)
    g = g.to(gpu)
    patches = 2 if nproto == 8 else 1
    # K = 5000: through the per-pixel d(scores) tables and the pixel-major CE, which the bench step takes on its own
    # (5 patches draw more samples than the image has pixels) and one patch would not
    extra = dict(ds_table=True) if nproto > 64 else {}
    sw = SwAVClustering(model=g, model_config=Cfg, out_dir=None, device="cuda",
                        **ffhq_args(2, nproto, num_patches=patches, pick_source="host", **extra))
    torch.manual_seed(5)
    np.random.seed(5)
    with torch.no_grad():
        ml = S.G.mean_latent(sd, 128, 512)
    sw.mean_latent = ml.cuda()
    sw.build_heads()
    params = S.SwavParams(*[p.detach().cpu().clone() for p in sw._params()])
    images = [draw_image(512, 13, 6, 256 * 256, patches, transforms=True) for _ in range(2)]
    loss = sw.train_step(images)
    # the oracle draws its OWN rotate / flip source maps here (oracle/swav_ref.py:rotflip_index_map), it is not handed the HIP
    # ones as in the tests above: the two restatements of torchvision's rotation agree on every pixel of these four views
    oimgs = [S.ImageDraws(im.z, [S.ViewDraws(v.layer_no, v.map_noise, v.angle, v.flip) for v in im.views], im.picks)
             for im in images]
    for im in images:
        for v in im.views:
            assert v.angle is not None and abs(v.angle) > 0.05
            same = ops.rotflip_map(v.angle, v.flip, 256, 256, "cuda").cpu().long() == S.rotflip_index_map(v.angle, v.flip, 256, 256)
            assert float(same.float().mean()) > 0.9999
    ref_loss, grads, _ = S.swav_train_step(
        sd, ml, 0.7, oimgs, params, [None] * 3, lr=0.01, momentum=0.9, trust=0.01,
        use_larc=True, hlen=5376, patch_size=20000, num_patches=patches, temperature=0.01, eps=0.005, niters=10,
        source_pdf="uniform", sinkhorn_scope="global")
    np.testing.assert_allclose(loss.item(), ref_loss, rtol=LOSS_TOL)
    for got, want in zip(sw._grads, grads):
        assert rel(got, want) < GRAD_TOL
    for got, want in zip(sw._params(), (params.proj_w, params.proto_w, params.proto_b)):
        assert rel(got, want) < 1e-5


def test_car512_generator_and_cluster_map_vs_oracle(gpu, engine):
    """configs[3] shape: a 512^2 generator (15 taps, sum C = 5504 > hlen = 5376: the two
    native-resolution taps fall outside the descriptor) -> image, taps, projection of all 262 144
    pixels and the argmax cluster map against the oracle."""
    from oracle import swav_ref as S
    from ganecdotes_amd.swav_clustering import SwAVClustering
    torch.set_num_threads(min(16, torch.get_num_threads()))
    g, sd = full_generator(512, seed=9)
    g = g.to(gpu)
    a = ffhq_args(1, 16)
    a["perturb_args"] = dict(truncation=0.7, n_layers=7, n_samples=1, layer_no=None, perturb_std=[1.0] * 7)
    sw = SwAVClustering(model=g, model_config=Cfg, out_dir=None, device="cuda", **a)
    torch.manual_seed(2)
    with torch.no_grad():
        ml = S.G.mean_latent(sd, 64, 512)
        w = S.G.mapping(torch.randn(1, 512), sd)
    sw.mean_latent = ml.cuda()
    sw.build_heads()
    proj_w = sw.projection[0].weight.data.cpu().clone()
    with torch.no_grad():
        img_ref, taps_ref, _ = S.G.generator_forward(w, sd, truncation=0.7, truncation_latent=ml,
                                                     input_is_latent=True)
    img, taps = g([w.cuda()], truncation=0.7, truncation_latent=ml.cuda(), input_is_latent=True,
                  randomize_noise=False)
    assert len(taps) == 15 and taps[-1].shape == (1, 64, 512, 512)
    assert rel(img, img_ref) < 1e-4
    for a_, b_ in zip(taps, taps_ref):
        assert rel(a_, b_) < 1e-4
    del img_ref, taps_ref
    preds_ref, labels_ref = S.predict_swav_codes(sd, w, ml, 0.7, proj_w, 5376)
    preds, labels = sw.predict_swav_codes(w.cuda())
    assert preds.shape == (1, 512, 512, 512) and labels.shape == (1, 512, 512)
    assert rel(preds, preds_ref) < 1e-4
    agree = (labels.cpu() == labels_ref).float().mean().item()
    assert agree > 0.995, agree
    # idempotence: no atomics anywhere on the predict path
    preds2, labels2 = sw.predict_swav_codes(w.cuda())
    assert torch.equal(preds, preds2) and torch.equal(labels, labels2)


def _oracle_images(images, size):
    """the oracle's draws of the same images, fed the HIP rotate/flip source maps (the maps
    themselves are compared with the oracle's restatement in test_hip_kernels.py)"""
    from oracle import swav_ref as S
    from ganecdotes_amd import ops
    return [S.ImageDraws(im.z, [S.ViewDraws(v.layer_no, v.map_noise, v.angle, v.flip,
                                            ops.rotflip_map(v.angle, v.flip, size, size, "cuda").cpu().long())
                                for v in im.views], im.picks) for im in images]


def test_cat256_full_size_step_vs_oracle(gpu, engine):
    """configs[2] (cat-256) parameters at full size: 16 prototypes, source_pdf='image' (marginals from
    the histograms of the whole 256^2 norm map, swav_clustering.py:523-532), eps = 0.003
    (configs/segmentors/hfc_with_swav_cat_config.py:52-65), one image / one patch of 20 000 pixels:
    loss, the three gradients and the LARC update against the oracle on identical draws.
    Tolerance: the marginals are HISTOGRAMS of the norm map, so a norm within rounding of a bin edge moves one count
    between two of the 20 000 / 16 bins and with it the Sinkhorn targets — a discontinuity of the reference's own
    definition, not arithmetic error (the same step with 'uniform' marginals agrees to 3e-7): 3.3e-4 on the loss
    measured here, held to north_star's 1e-3."""
    from oracle import swav_ref as S
    from ganecdotes_amd.draws import draw_image
    from ganecdotes_amd.swav_clustering import SwAVClustering
    torch.set_num_threads(min(16, torch.get_num_threads()))
    g, sd = full_generator(256, seed=43)
    g = g.to(gpu)
    a = ffhq_args(1, 16, num_patches=1, pick_source="host")
    a["sinkhorn_args"] = dict(source_pdf="image", niters=10, eps=0.003)
    sw = SwAVClustering(model=g, model_config=Cfg, out_dir=None, device="cuda", **a)
    torch.manual_seed(6)
    np.random.seed(6)
    with torch.no_grad():
        ml = S.G.mean_latent(sd, 128, 512)
    sw.mean_latent = ml.cuda()
    sw.build_heads()
    params = S.SwavParams(*[p.detach().cpu().clone() for p in sw._params()])
    images = [draw_image(512, 13, 6, 256 * 256, 1, transforms=True)]
    loss = sw.train_step(images)
    ref_loss, grads, _ = S.swav_train_step(
        sd, ml, 0.7, _oracle_images(images, 256), params, [None] * 3, lr=0.01, momentum=0.9, trust=0.01,
        use_larc=True, hlen=5376, patch_size=20000, num_patches=1, temperature=0.01, eps=0.003, niters=10,
        source_pdf="image", sinkhorn_scope="global")
    np.testing.assert_allclose(loss.item(), ref_loss, rtol=1e-3)
    for got, want in zip(sw._grads, grads):
        assert rel(got, want) < 2e-3
    for got, want in zip(sw._params(), (params.proj_w, params.proto_w, params.proto_b)):
        assert rel(got, want) < 1e-4


@pytest.mark.parametrize("nproto", [16, 4000])
def test_car512_full_size_train_step_vs_oracle(gpu, nproto):
    """configs[3] (car-512) TRAIN step at full size: a true 512^2 generator (15 taps, sum C = 5504;
    hlen = 5376 drops the two native-resolution taps, so every channel of the descriptor is
    upsampled), 20 000 of 262 144 pixels, eps = 0.01 (configs/segmentors/hfc_with_swav_car_config.py:
    52-65), 7 perturbable layers; BASELINE.json's 16 prototypes (low-rank path) and the reference's own
    4000 (generic MFMA path) at batch 1, one patch."""
    from oracle import swav_ref as S
    from ganecdotes_amd.draws import draw_image
    from ganecdotes_amd.swav_clustering import SwAVClustering
    torch.set_num_threads(min(16, torch.get_num_threads()))
    g, sd = full_generator(512, seed=9)
    g = g.to(gpu)
    a = ffhq_args(1, nproto, num_patches=1, pick_source="host")
    a["perturb_args"] = dict(truncation=0.7, n_layers=7, n_samples=1, layer_no=None, perturb_std=[1.0] * 7)
    a["sinkhorn_args"] = dict(source_pdf="uniform", niters=10, eps=0.01)
    sw = SwAVClustering(model=g, model_config=Cfg, out_dir=None, device="cuda", **a)
    torch.manual_seed(8)
    np.random.seed(8)
    with torch.no_grad():
        ml = S.G.mean_latent(sd, 64, 512)
    sw.mean_latent = ml.cuda()
    sw.build_heads()
    params = S.SwavParams(*[p.detach().cpu().clone() for p in sw._params()])
    images = [draw_image(512, 15, 7, 512 * 512, 1, transforms=True)]
    loss = sw.train_step(images)
    ref_loss, grads, _ = S.swav_train_step(
        sd, ml, 0.7, _oracle_images(images, 512), params, [None] * 3, lr=0.01, momentum=0.9, trust=0.01,
        use_larc=True, hlen=5376, patch_size=20000, num_patches=1, temperature=0.01, eps=0.01, niters=10,
        source_pdf="uniform", sinkhorn_scope="global")
    np.testing.assert_allclose(loss.item(), ref_loss, rtol=LOSS_TOL)
    for got, want in zip(sw._grads, grads):
        assert rel(got, want) < GRAD_TOL
    for got, want in zip(sw._params(), (params.proj_w, params.proto_w, params.proto_b)):
        assert rel(got, want) < 1e-5


def test_pidray256_full_size_train_step_vs_oracle(gpu):
    """hfc_with_swav_pidray TRAIN step at full size: the BagGAN generator at its own widths (512, 512, 512,
    256, 128, 64, 32 channels from 4^2 to 256^2: 13 taps, sum C = 2528 = hlen, hfc_with_swav_pidray_config.py:5-7), the
    reference's own 4000 prototypes, eps = 0.005 (:52, :66-68), batch 1, one patch of 20 000 pixels — against the oracle
    on the BagGAN-named weights (oracle/baggan_ref.py) and identical draws."""
    from oracle import baggan_ref as Bg
    from oracle import swav_ref as S
    from models.baggan.models import StyleGANGenerator
    from ganecdotes_amd.draws import draw_image
    from ganecdotes_amd.swav_clustering import SwAVClustering
    torch.set_num_threads(min(16, torch.get_num_threads()))
    torch.manual_seed(21)
    g = StyleGANGenerator((512, 512), 256, img_chls=3)
    for n, p in g.named_parameters():
        if "noise" in n and n.endswith("weight"):
            p.data.fill_(0.1)
    sd = Bg.to_stylegan2_keys({k: v.detach().clone() for k, v in g.state_dict().items()})
    g = g.eval().to(gpu)
    a = ffhq_args(1, 4000, num_patches=1, pick_source="host", hlen=2528)
    sw = SwAVClustering(model=g, model_config=Cfg, out_dir=None, device="cuda", **a)
    torch.manual_seed(12)
    np.random.seed(12)
    with torch.no_grad():
        ml = S.G.mean_latent(sd, 64, 512)
    sw.mean_latent = ml.cuda()
    sw.build_heads()
    assert sw._params()[0].shape == (512, 2528)
    params = S.SwavParams(*[p.detach().cpu().clone() for p in sw._params()])
    images = [draw_image(512, 13, 6, 256 * 256, 1, transforms=True)]
    loss = sw.train_step(images)
    ref_loss, grads, _ = S.swav_train_step(
        sd, ml, 0.7, _oracle_images(images, 256), params, [None] * 3, lr=0.01, momentum=0.9, trust=0.01,
        use_larc=True, hlen=2528, patch_size=20000, num_patches=1, temperature=0.01, eps=0.005, niters=10,
        source_pdf="uniform", sinkhorn_scope="global")
    np.testing.assert_allclose(loss.item(), ref_loss, rtol=LOSS_TOL)
    for got, want in zip(sw._grads, grads):
        assert rel(got, want) < GRAD_TOL
    for got, want in zip(sw._params(), (params.proj_w, params.proto_w, params.proto_b)):
        assert rel(got, want) < 1e-5


def test_simclr_ffhq256_iteration_vs_oracle(gpu):
    """configs[4] (ffhq-256 hfc_with_simclr) at full width: one optimizer iteration — 20 picked pixel
    pairs of two 256^2 x 5376 views, Linear -> BatchNorm1d(train) -> LeakyReLU -> Linear, the reference's
    inline NT-Xent (T = 1.0) — against the oracle on identical draws: loss, the four gradients, the
    LARC update and the BatchNorm running statistics."""
    from oracle import simclr_ref as C
    from oracle import swav_ref as S
    from ganecdotes_amd.simclr_clustering import SimCLRClustering, draw_simclr_iteration
    torch.set_num_threads(min(16, torch.get_num_threads()))
    g, sd = full_generator(256, seed=44)
    g = g.to(gpu)
    args = dict(num_iters=1, batch_size=20, patch_size=20000, hf_interp="nearest", trust_coeff=0.01,
                train_args=dict(lr=0.01, momentum=0.9), temperature=1.0, nclasses=512, hlen=5376,
                epoch_print_freq=1, max_masks=4, use_larc=True, transforms=True)
    sim = SimCLRClustering(model=g, model_config=Cfg, out_dir=None, device="cuda", simclr_args=args,
                           perturb_args=dict(truncation=0.7, n_layers=6, n_samples=1, layer_no=None,
                                             perturb_std=[1.0] * 6))
    torch.manual_seed(12)
    np.random.seed(12)
    with torch.no_grad():
        ml = S.G.mean_latent(sd, 128, 512)
    sim.mean_latent = ml.cuda()
    sim.build_heads()
    leaves = [p.detach().cpu().clone() for p in sim._params()]
    im = draw_simclr_iteration(512, 13, 6, 256 * 256, transforms=True)
    loss = sim.train_step(im)
    oim = _oracle_images([im], 256)[0]
    for p in leaves:
        p.requires_grad_(True)
    ref_loss, mean, var_u = C.simclr_step_loss(sd, ml, 0.7, oim, *leaves, hlen=5376, batch=20, temperature=1.0)
    ref_loss.backward()
    grads = [p.grad.detach().clone() for p in leaves]
    np.testing.assert_allclose(loss.item(), float(ref_loss.detach()), rtol=LOSS_TOL)
    for got, want in zip(sim._grads, grads):
        assert rel(got, want) < GRAD_TOL
    for p in leaves:
        p.requires_grad_(False)
    with torch.no_grad():
        S.larc_sgd_step(leaves, grads, [None] * 4, 0.01, 0.9, 0.01, True)
    for got, want in zip(sim._params(), leaves):
        assert rel(got, want) < 1e-4
    bn = sim.projection[1]
    assert rel(bn.running_mean, 0.1 * mean) < GRAD_TOL
    assert rel(bn.running_var, 0.9 + 0.1 * var_u) < GRAD_TOL


# --------------------------------------------------------------------------- properties at the bench batch
@pytest.fixture(scope="module")
def bench_state(gpu):
    from ganecdotes_amd.swav_clustering import SwAVClustering
    g, _ = full_generator(256)
    g = g.to(gpu)
    torch.manual_seed(1)
    np.random.seed(1)
    ml = g.mean_latent(128)

    def make(nproto, batch=16, **over):
        sw = SwAVClustering(model=g, model_config=Cfg, out_dir=None, device="cuda",
                            **ffhq_args(batch, nproto, **over))
        sw.mean_latent = ml
        torch.manual_seed(77)
        sw.build_heads()
        return sw
    return g, ml, make


def test_pyramid_projection_adjoint_at_batch16(gpu, bench_state):
    """<project(W), G> == <W, backward(G)> over 16 x 256^2 pixels x 512 dims: the table
    projection and the gradient route (sum-pool + TN GEMMs) are exact adjoints of each other."""
    from ganecdotes_amd import ops
    from ganecdotes_amd.pyramid import Pyramid
    g, ml, _ = bench_state
    torch.manual_seed(3)
    w = g.style(torch.randn(16, 512, device="cuda"))
    _, feats = g([w], truncation=0.7, truncation_latent=ml, input_is_latent=True, randomize_noise=False)
    ts = ops.TapSet([f.permute(0, 2, 3, 1) for f in feats])
    pyr = Pyramid(ts, 5376)
    wp = torch.randn(512, 5376, device="cuda") / 73.0
    z = pyr.project(wp)
    assert z.shape == (16, 256, 256, 512)
    # linearity in W
    z2 = pyr.project(2.5 * wp)
    assert rel(z2, 2.5 * z) < 1e-5
    del z2
    gz = torch.randn_like(z)
    lhs = float((z.double() * gz.double()).sum())
    gw = torch.empty_like(wp)
    pyr.backward(gz, gw, accumulate=False)
    rhs = float((wp.double() * gw.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), abs(rhs), 1.0), (lhs, rhs)
    # accumulate=True adds
    pyr.backward(gz, gw, accumulate=True)
    rhs2 = float((wp.double() * gw.double()).sum())
    assert abs(rhs2 - 2 * rhs) <= 1e-4 * abs(2 * rhs)


def test_generator_is_batch_independent(gpu, bench_state):
    """Image b of a 16-image batch equals the same latent run alone (weights are shared by the
    batch, styles live in the A operand): same taps to fp32 rounding of split-K order."""
    g, ml, _ = bench_state
    torch.manual_seed(4)
    w = g.style(torch.randn(16, 512, device="cuda"))
    kw = dict(truncation=0.7, truncation_latent=ml, input_is_latent=True, randomize_noise=False)
    img, taps = g([w], **kw)
    taps = [t.clone() for t in taps]
    img = img.clone()
    for b in (0, 9, 15):
        img1, taps1 = g([w[b:b + 1]], **kw)
        assert rel(img1, img[b:b + 1]) < 2e-5
        for a_, b_ in zip(taps1, taps):
            assert rel(a_, b_[b:b + 1]) < 2e-5


@pytest.mark.parametrize("nproto", [8, 16, 5000])
def test_bench_step_properties(gpu, bench_state, nproto):
    """configs[1] (K=8, batch 16), configs[2]'s per-rank shape (K=16) and the reference's own
    K=5000: codes are a transport plan with the prescribed per-sample marginal, the loss is
    the swapped cross-entropy of those codes (recomputed in fp64 with torch on the device), and
    the step is reproducible."""
    g, ml, make = bench_state
    batch = 16 if nproto <= 16 else 4
    sw = make(nproto, batch=batch, num_patches=1)
    torch.manual_seed(10)
    np.random.seed(10)
    images = sw.draw_batch(batch)
    torch.manual_seed(20)                       # device picks are drawn from the torch generator
    loss, aux = sw.train_step(images, return_aux=True)
    rows = batch * 20000
    for tag in "st":
        q = aux[f"q_{tag}"].double()
        assert q.shape == (rows, nproto) and bool((q >= 0).all())
        # the last Sinkhorn half-iteration normalises every sample's row to c = 1/rows, and the
        # reference then multiplies by rows (swav_clustering.py:571) -> rows sum to 1
        assert float((q.sum(1) - 1).abs().max()) < 1e-4
        col = q.sum(0) / rows
        assert abs(float(col.sum()) - 1) < 1e-6
        assert float((col * nproto - 1).abs().max()) < 0.2, float((col * nproto - 1).abs().max())
    ps = torch.log_softmax(aux["scores_s"].double() / 0.01, 1)
    pt = torch.log_softmax(aux["scores_t"].double() / 0.01, 1)
    want = -0.5 * ((aux["q_t"].double() * ps).sum(1).mean() + (aux["q_s"].double() * pt).sum(1).mean())
    np.testing.assert_allclose(loss.item(), float(want), rtol=1e-4)
    # reproducibility from the same state and draws (float atomics reorder sums only)
    g0 = [x.clone() for x in sw._grads]
    sw2 = make(nproto, batch=batch, num_patches=1)
    torch.manual_seed(20)
    loss2 = sw2.train_step(images)
    np.testing.assert_allclose(loss2.item(), loss.item(), rtol=1e-5)
    for a_, b_ in zip(sw2._grads, g0):
        assert rel(a_, b_) < 1e-4


@pytest.mark.parametrize("nproto,pdf", [(8, "uniform"), (16, "image")])
def test_deterministic_mode_is_bitwise_reproducible_at_the_bench_batch(gpu, bench_state, nproto, pdf):
    """swav_args['deterministic'] = True on the K <= 16 route (BASELINE configs 2 and 3 per rank, 16 images, 5 patches): the
    coefficient scatter accumulates 64-bit fixed-point integers instead of float atomics, and every block sum on the route has
    a fixed order — two runs of three steps from the same state and draws give BITWISE equal losses, gradients and parameters
    (the reference sets cudnn.deterministic, lib/util/util.py:21-28); and the deterministic step is the default step to fp32
    rounding (loss 1e-6, gradients 1e-5)."""
    g, ml, make = bench_state
    runs = []
    for det in (True, True, False):
        sw = make(nproto, batch=16, deterministic=det)
        if pdf == "image":
            sw.sinkhorn_args["source_pdf"] = "image"
            sw.eps = sw.sinkhorn_args["eps"] = 0.003
        torch.manual_seed(31)
        np.random.seed(31)
        losses = []
        for _ in range(3):
            losses.append(sw.train_step(sw.draw_batch(16)))
        runs.append((torch.cat(losses).cpu(), [x.clone() for x in sw._grads], [p.clone() for p in sw._params()]))
    assert torch.equal(runs[0][0], runs[1][0])
    for a_, b_ in zip(runs[0][1] + runs[0][2], runs[1][1] + runs[1][2]):
        assert torch.equal(a_, b_)
    np.testing.assert_allclose(runs[0][0].numpy(), runs[2][0].numpy(), rtol=1e-5)
    for a_, b_ in zip(runs[0][1], runs[2][1]):
        assert rel(a_, b_) < 1e-4


@pytest.mark.parametrize("nproto,scope", [(5000, "global"), (4001, "image")])
def test_table_route_is_bitwise_reproducible(gpu, bench_state, nproto, scope):
    """The route of the reference's own prototype counts needs no switch: every d(scores) row is written once by its pixel's own
    thread, the table Sinkhorn and the loss / bias-gradient sums reduce block partials in a fixed order, and the backward is
    GEMMs — two runs of three steps from the same state and draws are BITWISE equal with swav_args['deterministic'] (accepted
    here as a no-op) and without it.  (`make` seeds the constructor's mean latent: bench_state builds every instance alike.)"""
    from ganecdotes_amd import ops
    if ops.get_gemm_mode() != 1:
        pytest.skip("the 'tables' route is the bf16x6 engine's")
    g, ml, make = bench_state
    runs = []
    for det in (True, False, False):
        sw = make(nproto, batch=4, deterministic=det, sinkhorn_scope=scope)
        torch.manual_seed(33)
        np.random.seed(33)
        losses = [sw.train_step(sw.draw_batch(4)) for _ in range(3)]
        assert sw.last_route.name == "tables"
        runs.append((torch.cat(losses).cpu(), [x.clone() for x in sw._grads], [p.clone() for p in sw._params()]))
    for other in runs[1:]:
        assert torch.equal(runs[0][0], other[0])
        for a_, b_ in zip(runs[0][1] + runs[0][2], other[1] + other[2]):
            assert torch.equal(a_, b_)
    # the route that cannot promise it says so
    sw = make(40, batch=4, deterministic=True)
    with pytest.raises(NotImplementedError, match="deterministic"):
        sw.train_step(sw.draw_batch(4))


@pytest.mark.parametrize("nproto", [8, 5000])
def test_prefetched_views_equal_inline_views_at_the_bench_batch(gpu, bench_state, nproto):
    """bench.py's pipeline at its own shape: the views of the NEXT batch are made on the view stream while the step of the
    current one runs (two full-size generator passes and a training step in flight together).  Three pipelined steps against
    the same three steps on one stream from the same state and draws: every loss and the final gradients agree to what the
    float atomics allow — i.e. nothing one stream reads is overwritten by the other (plane buffers, scratch arenas, taps)."""
    g, ml, make = bench_state
    batch = 16 if nproto == 8 else 4
    res = []
    for overlap in (True, False):
        sw = make(nproto, batch=batch)
        torch.manual_seed(11)
        np.random.seed(11)
        stage = sw.prefetch_views if overlap else sw.prepare_batch
        losses = []
        nxt = stage(sw.draw_batch(batch))
        for i in range(3):
            cur = nxt
            nxt = stage(sw.draw_batch(batch)) if i < 2 else None
            losses.append(sw.train_step(cur))
        res.append((torch.cat(losses).cpu().numpy(), [x.clone() for x in sw._grads]))
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=2e-5)
    for a_, b_ in zip(res[0][1], res[1][1]):
        assert rel(a_, b_) < 1e-4


@pytest.mark.parametrize("scope,nproto", [("global", 5000), ("image", 5000), ("global", 4000), ("global", 8000), ("image", 8000)])
def test_table_sinkhorn_step_at_bench_patches(gpu, bench_state, scope, nproto):
    """The reference's own prototype counts with the bench's 5 patches of 20 000 samples (4 images per step here): scores and
    d(scores) on per-pixel tables and the step's Sinkhorn problems solved ON the tables, every patch in one pass per iteration
    (DESIGN §4.16).  Size-independent properties of the codes (a transport plan with the prescribed sample marginal), the loss
    = the swapped cross-entropy of those codes recomputed in fp64, and the same step with the per-problem sweeps
    (swav_args['sinkhorn_table'] = False: an independent implementation on differently ordered tables) — loss and gradients."""
    g, ml, make = bench_state
    batch, P = 4, 5
    sw = make(nproto, batch=batch, num_patches=P, sinkhorn_scope=scope)
    sw2 = make(nproto, batch=batch, num_patches=P, sinkhorn_scope=scope, sinkhorn_table=False)
    torch.manual_seed(10)
    np.random.seed(10)
    images = sw.draw_batch(batch)
    torch.manual_seed(20)
    loss, aux = sw.train_step(images, return_aux=True)
    torch.manual_seed(20)
    loss2 = sw2.train_step(images)
    rows = batch * 20000 if scope == "global" else 20000          # rows of the problem aux hands out (patch 0; image 0)
    for tag in "st":
        q = aux[f"q_{tag}"].double()
        assert q.shape == (rows, nproto) and bool((q >= 0).all())
        assert float((q.sum(1) - 1).abs().max()) < 1e-4
    np.testing.assert_allclose(loss2.item(), loss.item(), rtol=2e-5)
    for a_, b_ in zip(sw._grads, sw2._grads):
        assert rel(a_, b_) < 1e-4
    if scope == "global":                                             # patch 0's share of the loss from its codes and scores
        ps = torch.log_softmax(aux["scores_s"].double() / 0.01, 1)
        pt = torch.log_softmax(aux["scores_t"].double() / 0.01, 1)
        part = -0.5 * ((aux["q_t"].double() * ps).sum(1).mean() + (aux["q_s"].double() * pt).sum(1).mean())
        assert np.isfinite(float(part)) and 0.2 * loss.item() < float(part) < 5 * loss.item()


def test_cat256_per_rank_step_properties(gpu, bench_state):
    """configs[2]'s per-rank shape exactly as hfc_with_swav_cat_config.py:52-65 sets it: 16 images, 16 prototypes,
    source_pdf='image', eps = 0.003.  The marginals the step prescribes are the reference's histograms of the whole norm
    map (oracle image_marginals: K bins for the prototypes, one bin per sample), the codes are a transport plan onto
    them (rows exactly — the last half-iteration —, prototype mass to Sinkhorn's residual), and the loss is the swapped
    cross-entropy of those codes."""
    from oracle import swav_ref as S
    g, ml, make = bench_state
    sw = make(16, batch=16, num_patches=1)
    sw.sinkhorn_args.update(source_pdf="image", eps=0.003)
    sw.eps = 0.003
    torch.manual_seed(12)
    np.random.seed(12)
    images = sw.draw_batch(16)
    torch.manual_seed(22)
    loss, aux = sw.train_step(images, return_aux=True)
    rows = 16 * 20000
    assert np.isfinite(loss.item())
    for v, tag in enumerate("st"):
        img = aux[f"norm_map_{tag}"].cpu()
        assert img.shape == (16, 256, 256)
        r_ref, c_ref = S.image_marginals(img, 16, rows)
        r, c = aux["r_marginals"][v].cpu(), aux["c_marginals"][v].cpu()
        # a sample on a bin edge may fall either side of it in fp32: a few counts out of 1 M
        assert float((r - r_ref).abs().max()) < 8.0 / img.numel() and float((c - c_ref).abs().max()) < 8.0 / img.numel()
        q = aux[f"q_{tag}"].double().cpu()
        assert q.shape == (rows, 16) and bool((q >= 0).all())
        assert float((q.sum(1) - 1).abs().max()) < 1e-4
        mass = (q * c.double()[:, None]).sum(0)                    # prototype mass of the plan diag(c) q
        assert abs(float(mass.sum()) - 1) < 1e-5
        assert float((mass - r.double()).abs().max()) < 0.3 * float(r.max()), (mass, r)   # Sinkhorn's residual after 10 iterations (random heads)
    ps = torch.log_softmax(aux["scores_s"].double() / 0.01, 1)
    pt = torch.log_softmax(aux["scores_t"].double() / 0.01, 1)
    want = -0.5 * ((aux["q_t"].double() * ps).sum(1).mean() + (aux["q_s"].double() * pt).sum(1).mean())
    np.testing.assert_allclose(loss.item(), float(want), rtol=1e-4)


def test_car512_per_rank_step_properties(gpu):
    """configs[3]'s per-rank shape (hfc_with_swav_car_config.py:52-65): a true 512^2 generator, 4 images, 16 prototypes,
    eps = 0.01, all 5 patches of 20 000 of 262 144 pixels.  hlen = 5376 cuts the 512^2 taps off, so the score tables
    live at 256^2 and the picks index them through the shift (pyramid.py); transport-plan, loss and reproducibility
    properties as in test_bench_step_properties."""
    from ganecdotes_amd.swav_clustering import SwAVClustering
    g, _ = full_generator(512, seed=9)
    g = g.to(gpu)
    torch.manual_seed(2)
    ml = g.mean_latent(64)

    def make():
        a = ffhq_args(4, 16)
        a["perturb_args"] = dict(truncation=0.7, n_layers=7, n_samples=1, layer_no=None, perturb_std=[1.0] * 7)
        a["sinkhorn_args"] = dict(source_pdf="uniform", niters=10, eps=0.01)
        sw = SwAVClustering(model=g, model_config=Cfg, out_dir=None, device="cuda", **a)
        sw.mean_latent = ml
        torch.manual_seed(78)
        sw.build_heads()
        return sw
    sw = make()
    torch.manual_seed(14)
    np.random.seed(14)
    images = sw.draw_batch(4)
    torch.manual_seed(24)
    loss, aux = sw.train_step(images, return_aux=True)
    rows = 4 * 20000
    for tag in "st":
        q = aux[f"q_{tag}"].double()
        assert q.shape == (rows, 16) and bool((q >= 0).all())
        assert float((q.sum(1) - 1).abs().max()) < 1e-4
        col = q.sum(0) / rows
        assert float((col * 16 - 1).abs().max()) < 0.2
    assert np.isfinite(loss.item())
    g0 = [x.clone() for x in sw._grads]
    assert all(bool(torch.isfinite(x).all()) for x in g0)
    sw2 = make()
    torch.manual_seed(24)
    loss2 = sw2.train_step(images)
    np.testing.assert_allclose(loss2.item(), loss.item(), rtol=1e-5)
    for a_, b_ in zip(sw2._grads, g0):
        assert rel(a_, b_) < 1e-4


def test_bench_cli_contract_one_json_line(gpu):
    """`python bench.py --gpus N --steps K --warmup W` as the driver runs it: exactly one JSON line on stdout with the
    contract's keys (metric / unit of BASELINE.json, whole-job value, roofline of the dominant kernel with
    frac = achieved / peak, the oracle timed as cpu_baseline at N = 1), and `--gpus 2` starting its own two ranks."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = json.load(open(os.path.join(root, "BASELINE.json")))

    def run(*args):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], cwd=root, capture_output=True, text=True,
                           timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, p.stdout[-2000:]
        return json.loads(lines[0])

    d = run("--steps", "3", "--warmup", "1")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == base.get("unit", d["unit"]) and d["vs_baseline"] is None and d["scaling"] == "weak"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 16 * 1e3 / d["ms_per_step"]) < 1e-3 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("mfma", "hbm") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] < 1.0
    assert 0.0 < r["step_frac"] < 1.0 and r["step_tflop"] > 0                  # the whole step against the same peak
    # the timed region runs the next batch's views beside the step: the kernel ALONE is timed on one stream right after it
    tr = r["timed_region"]                       # kernel figures: one-stream steps (agree with rocprofv3); the two-stream brackets beside them
    assert r["concurrent_streams"] == 2 and tr["frac"] <= r["frac"] < 1.0 and r["one_stream_ms_per_step"] > 0
    assert 0.0 < r["one_stream_step_frac"] < 1.0
    h = r["hbm"]                                                                # the largest HBM-bound pass
    assert h["bound"] == "hbm" and abs(h["frac"] - h["achieved"] / h["peak"]) < 1e-9 and 0.0 < h["frac"] < 1.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == d["unit"]
    rf = d["reference_faithful"]                                                # the reference's own nprototypes, same line
    assert rf["prototypes"] == 5000 and rf["unit"] == d["unit"] and rf["value"] > 0
    assert abs(rf["value"] - 16 * 1e3 / rf["ms_per_step"]) < 1e-3 * rf["value"]
    assert 0.0 < rf["roofline"]["mfma_frac"] < 1.0 and 0.0 < rf["roofline"]["hbm_frac"] < 1.0
    x6 = os.environ.get("GX_GEMM_MODE", "1") != "0"          # (the exact-fp32 engine has no weight-plane GEMMs: many prototypes run per sample)
    assert not x6 or ("route=tables" in rf["route"] and "table Sinkhorn" in rf["route"])
    rc = rf["cpu_baseline"]                                                     # ... with ITS CPU figure beside it (BASELINE.md par. 3)
    assert rc["kind"] == "port" and rc["value"] > 0 and "K=5000" in rc["sample"]
    assert "route=lowrank" in d["config"]["workload"]
    assert r["traffic_source"] is None or "profiles/" in r["traffic_source"]
    # every other BASELINE.json configuration, driver-timed in the same line
    per_step = {"cat_per_rank": 16, "car512_per_rank": 4, "car512_per_rank_k4000": 4, "simclr": 1, "predict": 16}
    for key, imgs in per_step.items():
        n = d[key]
        assert n["unit"] == "images/s" and n["value"] > 0 and n["ms_per_step"] > 0 and n["workload"], key
        assert abs(n["value"] - imgs * 1e3 / n["ms_per_step"]) < 1e-3 * n["value"], key
        nr = n["roofline"]
        assert nr["kernel"] and 0.0 < nr["frac"] < 1.0 and 0.0 < nr["step_frac"] < 1.0, key
    assert "source_pdf='image'" in d["cat_per_rank"]["workload"] and "route=lowrank" in d["cat_per_rank"]["workload"]
    assert (not x6 or "route=tables" in d["car512_per_rank_k4000"]["workload"]) and "512x512" in d["car512_per_rank"]["workload"]
    assert d["simclr"]["iterations_per_s"] > 0 and np.isfinite(d["simclr"]["loss"])
    d2 = run("--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert d2["n_gpus"] == 2 and d2["cpu_baseline"] is None and "reference_faithful" not in d2
    assert abs(d2["value"] - 2 * 16 * 1e3 / d2["ms_per_step"]) < 1e-3 * d2["value"]
