"""End-to-end parity of the HIP product path (generator taps -> SwAV step -> predict) against
the golden vectors captured from the reference and against the CPU oracle on the same draws.
Runs on a real MI355X only."""
import copy

import numpy as np
import pytest
import math

import torch

from conftest import golden_sd, load_golden

pytestmark = pytest.mark.gpu
T = torch.from_numpy
MINI = {4: 16, 8: 16, 16: 8, 32: 8, 64: 8}


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from ganecdotes_amd import ops
    ops.lib()
    return torch.device("cuda")


def mini_gen(sd, gpu, size=32):
    from ganecdotes_amd.generator import Generator
    g = Generator(size, 16, 2, channels=MINI)
    missing, unexpected = g.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    return g.eval().to(gpu)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


class Cfg:
    num_latents_for_mean = 64
    truncation = 0.7
    latent_dim = 16


def toy_args(**over):
    swav = dict(num_epochs=3, num_samples=1, num_patches=2, sampling_method="random", patch_size=300,
                hf_interp="nearest", warmup_epochs=3, start_warmup=0.01, use_scheduler=False, base_lr=0.01,
                final_lr=0.0001, trust_coeff=0.01, train_args=dict(lr=0.01, momentum=0.9), projn_nw="linear",
                temperature=0.1, nprototypes=12, nclasses=8, hlen=56, add_local_loss=False,
                plot_test_images=False, epoch_print_freq=1, max_masks=4)
    swav.update(over)
    return dict(perturb_args=dict(truncation=0.7, n_layers=3, n_samples=1, layer_no=None, perturb_std=[1.0] * 3),
                swav_args=swav, sinkhorn_args=dict(source_pdf="uniform", niters=10, eps=0.05))


def make_swav(g, gpu, tmp_path, **over):
    from ganecdotes_amd.swav_clustering import SwAVClustering
    sk = over.pop("sinkhorn_args", None)
    a = toy_args(**over)
    if sk:
        a["sinkhorn_args"].update(sk)
    return SwAVClustering(model=g, model_config=Cfg, out_dir=str(tmp_path), device="cuda", **a)


# ------------------------------------------------------------------------------------- generator
def test_generator_golden_gv4(gpu):
    d = load_golden("gv4_mini_generator")
    g = mini_gen(golden_sd(d), gpu)
    w, ml = T(d["w"]).cuda(), T(d["mean_latent"]).cuda()
    assert rel(g.style(T(d["z"]).cuda()), T(d["w"])) < 1e-5
    img, taps = g([w], truncation=0.7, truncation_latent=ml, input_is_latent=True, randomize_noise=False)
    assert rel(img, T(d["img_once"])) < 2e-5
    for i, t in enumerate(taps):
        assert t.shape == d[f"tap_once_{i}"].shape
        assert rel(t, T(d[f"tap_once_{i}"])) < 2e-5, i
    _, lat = g([w], return_latents=True, truncation=0.7, truncation_latent=ml, input_is_latent=True)
    assert rel(lat, T(d["latent_trunc_once"])) < 1e-6
    img2, taps2 = g([lat], truncation=0.7, truncation_latent=ml, input_is_latent=True, randomize_noise=False)
    assert rel(img2, T(d["img_twice"])) < 2e-5
    for i, t in enumerate(taps2):
        assert rel(t, T(d[f"tap_twice_{i}"])) < 2e-5, i
    imgz, tapsz = g([T(d["z"]).cuda()], randomize_noise=False)
    assert rel(imgz, T(d["img_z"])) < 2e-5
    # state_dict keys are the reference's
    assert set(g.state_dict().keys()) == set(golden_sd(d).keys())


def test_generator_full_width_vs_oracle(gpu):
    """ffhq-width generator at 64x64 (512-channel layers) against the oracle."""
    from ganecdotes_amd.generator import Generator
    from oracle import stylegan2_ref as G
    torch.manual_seed(7)
    g = Generator(64, 512, 8)
    for n, p in g.named_parameters():
        if n.endswith("noise.weight"):
            p.data.fill_(0.1)
    sd = {k: v.clone() for k, v in g.state_dict().items()}
    g = g.eval().to(gpu)
    z = torch.randn(2, 512)
    ml = G.mean_latent(sd, 256, 512)
    with torch.no_grad():
        w = G.mapping(z, sd)
        img_ref, taps_ref, _ = G.generator_forward(w, sd, 0.7, ml, input_is_latent=True)
    assert rel(g.style(z.cuda()), w) < 2e-5
    img, taps = g([w.cuda()], truncation=0.7, truncation_latent=ml.cuda(), input_is_latent=True,
                  randomize_noise=False)
    assert rel(img, img_ref) < 5e-5
    for a, b in zip(taps, taps_ref):
        assert rel(a, b) < 5e-5


def test_generator_xp_path_equals_in_kernel_split(gpu):
    """ffhq-width generator at 64 x 64, 16 images: with the x-planes conv path (activations split once by the
    producing epilogue / blur pass, LDS-DMA main loop) the layers that qualify (>= 256 tiles) give the same taps
    and image as the in-kernel-split path — the same six bf16 products per fp32 product; the 16x16x32 MFMA form adds
    them pairwise inside one instruction, so the fp32 summation order (only) differs: 1e-5 after 9 layers."""
    from ganecdotes_amd import generator as GM
    from ganecdotes_amd import ops
    torch.manual_seed(17)
    g = GM.Generator(64, 512, 8)
    for n, p in g.named_parameters():
        if n.endswith("noise.weight"):
            p.data.fill_(0.1)
    g = g.eval().to(gpu)
    w = g.style(torch.randn(16, 512, device="cuda"))
    ml = g.mean_latent(64)
    kw = dict(truncation=0.7, truncation_latent=ml, input_is_latent=True, randomize_noise=False)
    prev, prev_mode = GM.CONV_XP, ops.set_gemm_mode(1)
    try:
        GM.CONV_XP = False
        img0, taps0 = g([w], **kw)
        taps0 = [t.clone() for t in taps0]
        GM.CONV_XP = True
        g._xplanes.clear()
        img1, taps1 = g([w], **kw)
        users = set(g._xplanes)                              # consumer layers that read x planes (the generator's own buffers)
        assert {"convs.4", "convs.5", "convs.6", "convs.7"} <= users, users
        # one buffer per layer: another batch size replaces it instead of pinning a second set
        g([w[:4]], **kw)
        assert set(g._xplanes) == users and any(shape[0] == 4 for shape, _ in g._xplanes.values())
        img1, taps1 = g([w], **kw)
        assert rel(img1, img0) < 1e-5
        for a, b in zip(taps1, taps0):
            assert rel(a, b) < 1e-5
    finally:
        GM.CONV_XP = prev
        ops.set_gemm_mode(prev_mode)


def test_generator_plane_formats_agree_and_the_f16x3_range_is_loud(gpu, tmp_path):
    """ffhq-width generator at 64 x 64, 16 images under the two plane formats of the x-planes convs (ops.set_conv_planes: three
    bf16 planes / six products, two scaled fp16 planes / three products): both within 1e-5 of the exact-fp32 MFMA engine, the
    frozen weight planes re-made when the format changes.  The price of f16x3 is range: activations scaled past |x * style| =
    32752 come out NaN — and predict_swav_codes says so (FloatingPointError naming the switch) instead of returning labels,
    while bf16x6 (fp32's range) still computes them."""
    from ganecdotes_amd import generator as GM
    from ganecdotes_amd import ops
    torch.manual_seed(18)
    g = GM.Generator(64, 512, 8)
    for n, p in g.named_parameters():
        if n.endswith("noise.weight"):
            p.data.fill_(0.1)
    g = g.eval().to(gpu)
    w = g.style(torch.randn(16, 512, device="cuda"))
    ml = g.mean_latent(64)
    kw = dict(truncation=0.7, truncation_latent=ml, input_is_latent=True, randomize_noise=False)
    prev_mode, prev_fmt = ops.set_gemm_mode(0), ops.get_conv_planes()
    try:
        ref = [t.clone() for t in g([w], **kw)[1]]
        ops.set_gemm_mode(1)
        outs = {}
        for fmt in (1, 0, 1):
            ops.set_conv_planes(fmt)
            outs[fmt] = [t.clone() for t in g([w], **kw)[1]]
            assert g._derived["plane_format"] == (1, fmt)
            for a, b in zip(outs[fmt], ref):
                assert rel(a, b) < 1e-5
        assert any(not torch.equal(a, b) for a, b in zip(outs[0], outs[1]))       # (two different engines did run)
        # out of range: the constant input x 3e5 (demodulation keeps the later layers at that scale)
        sw = make_swav(g, gpu, tmp_path)
        sw.build_heads()
        lat = w[:2]
        sw.predict_swav_codes(lat)
        g.input.input.data.mul_(3e5)
        g._invalidate()
        with pytest.raises(FloatingPointError, match="GX_CONV_PLANES=0"):
            sw.predict_swav_codes(lat)
        ops.set_conv_planes(0)
        preds, labels = sw.predict_swav_codes(lat)
        assert bool(torch.isfinite(preds).all())
    finally:
        ops.set_conv_planes(prev_fmt)
        ops.set_gemm_mode(prev_mode)


def test_feature_range_probe_falls_back_to_bf16x6(gpu, tmp_path):
    """The preprocessors' constructors send a few untruncated latents through the frozen generator once: a checkpoint whose
    activations sit outside the f16x3 window (here: the constant input x 3e5) switches the process to the bf16x6 forms with a
    warning, and training / predicting then work; an ordinary generator leaves the engines alone."""
    import os
    from ganecdotes_amd import generator as GM
    from ganecdotes_amd import ops
    if ops.get_gemm_mode() != 1 or ops.get_conv_planes() != 1 or not ops.GEMM_H2 or \
            "GX_CONV_PLANES" in os.environ or "GX_GEMM_PLANES" in os.environ:
        pytest.skip("the probe only acts on the default engines")
    torch.manual_seed(21)
    g = GM.Generator(64, 512, 8).eval().to(gpu)
    try:
        sw = make_swav(g, gpu, tmp_path)
        assert sw.feature_peak is not None and sw.feature_peak < ops.FEATURE_RANGE / 4
        assert ops.get_conv_planes() == 1 and ops.GEMM_H2
        state = torch.cuda.get_rng_state()
        assert sw.probe_feature_range() == sw.feature_peak                     # a private RNG stream: repeatable ...
        assert torch.equal(torch.cuda.get_rng_state(), state)                   # ... and the run's own draws are untouched
        g.input.input.data.mul_(3e5)
        g._invalidate()
        sw2 = make_swav(g, gpu, tmp_path)
        assert not (sw2.feature_peak < ops.FEATURE_RANGE / 4)
        assert ops.get_conv_planes() == 0 and not ops.GEMM_H2
        sw2.build_heads()
        preds, labels = sw2.predict_swav_codes(g.style(torch.randn(2, 512, device="cuda")))
        assert bool(torch.isfinite(preds).all())
    finally:
        ops.set_conv_planes(1)
        ops.GEMM_H2 = True


def test_style_norm_keeps_huge_styles_inside_the_f16x3_range(gpu):
    """gx_style_norm: per (styled conv, sample) the style divided by the power of two that brings its largest magnitude into
    [2, 4), the demod coefficients multiplied by it — exact (bitwise the same taps on the exact-fp32 engine), and what makes the
    f16x3 planes' fixed range a bound of the ACTIVATIONS only: with every style affine scaled 30000 x (styles far beyond what
    trained checkpoints reach) the un-normalised f16x3 convs overflow to NaN, the normalised ones stay within 1e-5 of
    the exact engine."""
    from ganecdotes_amd import generator as GM
    from ganecdotes_amd import ops
    # the kernel on a packed table: two "convs" (cin 70 / 300, cout 5 / 9), 3 samples
    torch.manual_seed(5)
    b = 3
    table = torch.tensor([0, 70, 5, 0, 0, 70, 300, 9, 0, 5], dtype=torch.int32, device="cuda")
    sp = torch.randn(b * 370, device="cuda") * torch.logspace(-3, 3, b * 370, device="cuda")[torch.randperm(b * 370, device="cuda")]
    sp[b * 70 + 300:b * 70 + 600] = 0.0                                   # conv 1, sample 1: an all-zero style stays as it is
    dm = torch.randn(b * 14, device="cuda")
    s0, d0 = sp.clone(), dm.clone()
    ops.style_norm_(sp, dm, b, table, 2)
    for j, (soff, cin, cout, doff) in enumerate(((0, 70, 5, 0), (70, 300, 9, 5))):
        for i in range(b):
            a0, a1 = s0[b * soff + i * cin:b * soff + (i + 1) * cin], sp[b * soff + i * cin:b * soff + (i + 1) * cin]
            e0, e1 = d0[b * doff + i * cout:b * doff + (i + 1) * cout], dm[b * doff + i * cout:b * doff + (i + 1) * cout]
            m = float(a0.abs().max())
            if m == 0.0:
                assert torch.equal(a0, a1) and torch.equal(e0, e1)
                continue
            c = 2.0 ** (math.floor(math.log2(m)) - 1)
            assert 2.0 <= float(a1.abs().max()) < 4.0
            assert torch.equal(a1, a0 / c) and torch.equal(e1, e0 * c)
    # the generator
    torch.manual_seed(19)
    g = GM.Generator(64, 512, 8)
    for n, p in g.named_parameters():
        if n.endswith("noise.weight"):
            p.data.fill_(0.1)
        if "modulation" in n and "to_rgb" not in n:
            p.data.mul_(30000.0)
    g = g.eval().to(gpu)
    w = g.style(torch.randn(8, 512, device="cuda"))
    kw = dict(truncation=0.7, truncation_latent=g.mean_latent(64), input_is_latent=True, randomize_noise=False)
    prev_mode, prev_fmt, prev_norm = ops.set_gemm_mode(0), ops.get_conv_planes(), GM.STYLE_NORM
    try:
        GM.STYLE_NORM = False
        ref = [t.clone() for t in g([w], **kw)[1]]
        GM.STYLE_NORM = True
        for a, r in zip(g([w], **kw)[1], ref):
            assert torch.equal(a, r)                                      # exact-fp32 engine: powers of two commute with it
        ops.set_gemm_mode(1)
        for fmt in (1, 0):
            ops.set_conv_planes(fmt)
            for a, r in zip(g([w], **kw)[1], ref):
                assert rel(a, r) < 1e-5
        ops.set_conv_planes(1)
        GM.STYLE_NORM = False
        assert not all(bool(torch.isfinite(t).all()) for t in g([w], **kw)[1])          # what the normalisation is for
    finally:
        GM.STYLE_NORM = prev_norm
        ops.set_conv_planes(prev_fmt)
        ops.set_gemm_mode(prev_mode)


def test_baggan_generator_golden_gv14(gpu):
    """pidray-* models: the reference BagGAN StyleGANGenerator's mapping, image and
    7 feature taps, loaded from its own state_dict through the key translation"""
    from models.baggan.models import StyleGANGenerator
    d = load_golden("gv14_baggan")
    ref = {k[3:]: T(v) for k, v in d.items() if k.startswith("sd_")}
    g = StyleGANGenerator(latent_dims=(16, 16), img_resolution=32, mlp_layers=2, _channels={4: 16, 8: 16, 16: 8, 32: 8})
    g.load_state_dict(ref)
    g = g.eval().to(gpu)
    w, ml = T(d["w"]).cuda(), T(d["mean_latent"]).cuda()
    assert rel(g.style(T(d["z"]).cuda()), T(d["w"])) < 1e-5
    img, taps = g([w], truncation=0.7, truncation_latent=ml, input_is_latent=True, randomize_noise=False)
    assert rel(img, T(d["img"])) < 2e-5
    assert len(taps) == 7
    for i, t in enumerate(taps):
        assert rel(t, T(d[f"tap_{i}"])) < 2e-5, i
    imgz, tapsz = g([T(d["z"]).cuda()], randomize_noise=False, truncation_latent=ml)
    assert rel(imgz, T(d["img_z"])) < 2e-5 and rel(tapsz[-1], T(d["tap_z_last"])) < 2e-5
    _, lat = g([w], return_latents=True, truncation=0.7, truncation_latent=ml, input_is_latent=True)
    assert rel(lat, T(d["latent"])) < 1e-6


# ------------------------------------------------------------------------------------- views
def test_views_golden_gv9(gpu, tmp_path):
    from ganecdotes_amd.draws import ImageDraws, ViewDraws
    d = load_golden("gv9_view_rng")
    g = mini_gen(golden_sd(d), gpu)
    sw = make_swav(g, gpu, tmp_path)
    sw.mean_latent = T(d["mean_latent"]).cuda()
    im = ImageDraws(T(d["z"]), [ViewDraws(int(d[f"replay_layer_no_{v}"]),
                                          T(d[f"replay_map_noise_{v}"])[[2 * int(d[f"replay_layer_no_{v}"]),
                                                                          2 * int(d[f"replay_layer_no_{v}"]) + 1], 0])
                                for v in "st"], [])
    views = sw.make_views([im])
    from ganecdotes_amd import ops
    picks = torch.arange(32 * 32, dtype=torch.int32, device="cuda").reshape(1, -1)
    for (ts, maps, img), tag in zip(views, "st"):
        x = ops.gather_taps(ts, picks, 56)
        ref = T(d[f"stack_{tag}"]).reshape(56, -1).t()
        assert rel(x, ref) < 2e-5
        assert rel(img, T(d[f"img_{tag}"])) < 2e-5


# ------------------------------------------------------------------------------------- GV10 / GV11
def test_pretrain_trajectory_golden_gv10_and_predict_gv11(gpu, tmp_path):
    d = load_golden("gv10_pretrain")
    g = mini_gen(golden_sd(d), gpu)
    sw = make_swav(g, gpu, tmp_path, use_larc=False, transforms=False, cpu_reference_rng=True)
    torch.manual_seed(42)
    np.random.seed(42)
    z = torch.randn(64, 16)                              # the reference constructor's mean_latent draw
    sw.mean_latent = g.style(z.cuda()).mean(0, keepdim=True)
    sw.pretrain(None, num_test_samples=2)
    losses = [float(l.item()) for l in sw.loss_history]
    np.testing.assert_allclose(losses, d["losses"], rtol=1e-3)
    assert rel(sw.projection[0].weight.data, T(d["proj_weight"])) < 1e-3
    assert rel(sw.prototype.weight.data, T(d["proto_weight"])) < 1e-3
    assert rel(sw.prototype.bias.data, T(d["proto_bias"])) < 1e-3
    # files are loadable torch modules, callable like the reference's
    proj = torch.load(tmp_path / "projection.pt", weights_only=False)
    proto = torch.load(tmp_path / "prototypes.pt", weights_only=False)
    assert proto(proj(torch.zeros(1, 56, device="cuda"))).shape == (1, 12)
    # GV11: predict with the golden projection
    d11 = load_golden("gv11_predict")
    sw.projection[0].weight.data.copy_(T(d["proj_weight"]).cuda())
    sw.mean_latent = T(d11["mean_latent"]).cuda()
    preds, labels = sw.predict_swav_codes(T(d11["w"]).cuda())
    assert preds.shape == d11["preds"].shape and labels.dtype == torch.int64
    assert rel(preds, T(d11["preds"])) < 5e-5
    assert (labels.cpu().numpy() == d11["labels"]).mean() > 0.995


# ------------------------------------------------------------------------------------- one step vs oracle
@pytest.mark.parametrize("mode", ["pyramid", "panel"])
@pytest.mark.parametrize("scope,batch,pdf", [("image", 1, "uniform"), ("image", 2, "uniform"),
                                             ("global", 3, "uniform"), ("image", 1, "image")])
def test_train_step_vs_oracle(gpu, tmp_path, scope, batch, pdf, mode):
    from oracle import swav_ref as S
    from ganecdotes_amd.draws import draw_image
    d = load_golden("gv10_pretrain")
    sd = golden_sd(d)
    g = mini_gen(sd, gpu)
    sw = make_swav(g, gpu, tmp_path, sinkhorn_scope=scope, batch_size=batch, projection_mode=mode,
                   sinkhorn_args=dict(source_pdf=pdf))
    torch.manual_seed(123)
    np.random.seed(123)
    ml = S.G.mean_latent(sd, 64, 16)
    sw.mean_latent = ml.cuda()
    sw.build_heads()
    params = S.SwavParams(*[p.detach().cpu().clone() for p in sw._params()])
    bufs = [None] * 3
    for step in range(2):
        images = [draw_image(16, 7, 3, 32 * 32, 2, transforms=True) for _ in range(batch)]
        # the device rotate/flip map and the oracle's may differ on nearest-rounding ties only;
        # parity of everything downstream is checked on identical draws
        loss = sw.train_step(images)
        from ganecdotes_amd import ops
        oimgs = [S.ImageDraws(im.z, [S.ViewDraws(v.layer_no, v.map_noise, v.angle, v.flip,
                                                 ops.rotflip_map(v.angle, v.flip, 32, 32, "cuda").cpu().long())
                                     for v in im.views], im.picks) for im in images]
        ref_loss, grads, bufs = S.swav_train_step(
            sd, ml, 0.7, oimgs, params, bufs, lr=0.01, momentum=0.9, trust=0.01, use_larc=True, hlen=56,
            patch_size=300, num_patches=2, temperature=0.1, eps=0.05, niters=10, source_pdf=pdf,
            sinkhorn_scope=scope)
        np.testing.assert_allclose(loss.item(), ref_loss, rtol=1e-3)
        for got, want in zip(sw._grads, grads):
            assert rel(got, want) < 2e-3
        for got, want in zip(sw._params(), (params.proj_w, params.proto_w, params.proto_b)):
            assert rel(got, want) < 1e-4


@pytest.mark.parametrize("scope,batch,pdf,nproto", [("image", 1, "uniform", 12), ("global", 2, "image", 12), ("global", 2, "uniform", 200)])
def test_train_step_and_predict_with_bilinear_hf_interp_vs_oracle(gpu, tmp_path, scope, batch, pdf, nproto):
    """swav_args['hf_interp'] = 'bilinear': F.interpolate's other upsampling of the taps (hfc_with_swav/swav_clustering.py:
    123-124; every shipped config says 'nearest').  The pyramid factorisation IS the nearest upsampling, so the step runs the
    reference-shaped panel route (Route.mode == 'panel') on bilinearly gathered descriptors; two steps (loss, gradients,
    updated parameters; the 'image' marginals from the bilinear stack's norm maps) and predict_swav_codes against the oracle,
    whose stack is torch's own F.interpolate(mode='bilinear')."""
    from oracle import swav_ref as S
    from ganecdotes_amd import ops
    from ganecdotes_amd.draws import draw_image
    d = load_golden("gv10_pretrain")
    sd = golden_sd(d)
    g = mini_gen(sd, gpu)
    sw = make_swav(g, gpu, tmp_path, sinkhorn_scope=scope, batch_size=batch, hf_interp="bilinear", nprototypes=nproto,
                   sinkhorn_args=dict(source_pdf=pdf))
    torch.manual_seed(124)
    np.random.seed(124)
    ml = S.G.mean_latent(sd, 64, 16)
    sw.mean_latent = ml.cuda()
    sw.build_heads()
    params = S.SwavParams(*[p.detach().cpu().clone() for p in sw._params()])
    bufs = [None] * 3
    for step in range(2):
        images = [draw_image(16, 7, 3, 32 * 32, 2, transforms=True) for _ in range(batch)]
        loss = sw.train_step(images)
        assert sw.last_route.mode == "panel" and sw.last_route.name == "per_sample"
        oimgs = [S.ImageDraws(im.z, [S.ViewDraws(v.layer_no, v.map_noise, v.angle, v.flip,
                                                 ops.rotflip_map(v.angle, v.flip, 32, 32, "cuda").cpu().long())
                                     for v in im.views], im.picks) for im in images]
        ref_loss, grads, bufs = S.swav_train_step(
            sd, ml, 0.7, oimgs, params, bufs, lr=0.01, momentum=0.9, trust=0.01, use_larc=True, hlen=56,
            patch_size=300, num_patches=2, temperature=0.1, eps=0.05, niters=10, source_pdf=pdf,
            sinkhorn_scope=scope, hf_interp="bilinear")
        np.testing.assert_allclose(loss.item(), ref_loss, rtol=1e-3)
        for got, want in zip(sw._grads, grads):
            assert rel(got, want) < 2e-3
        for got, want in zip(sw._params(), (params.proj_w, params.proto_w, params.proto_b)):
            assert rel(got, want) < 1e-4
    for _ in range(2):                                   # (the reference's predict is one latent per call, :685-690)
        w = S.G.mapping(torch.randn(1, 16), sd)
        preds, labels = sw.predict_swav_codes(w.cuda())
        ref_preds, ref_labels = S.predict_swav_codes(sd, w, ml, 0.7, params.proj_w, 56, hf_interp="bilinear")
        assert rel(preds, ref_preds) < 5e-5
        assert (labels.cpu() == ref_labels).float().mean() > 0.995
        # the nearest stack gives other codes: the option is not silently ignored
        nn_preds, _ = S.predict_swav_codes(sd, w, ml, 0.7, params.proj_w, 56)
        assert rel(preds, nn_preds) > 1e-2


@pytest.mark.parametrize("scope,batch,transforms,score_table,nproto,pdf", [
    ("global", 3, True, True, 200, "uniform"), ("global", 3, True, False, 200, "uniform"), ("image", 2, True, True, 200, "uniform"),
    ("global", 2, False, True, 200, "uniform"), ("global", 2, True, True, 6148, "uniform"), ("image", 2, True, True, 6148, "uniform"),
    ("global", 3, True, True, 200, "image"), ("global", 2, True, False, 200, "image"), ("image", 2, True, True, 200, "image")])
def test_train_step_with_per_pixel_dscore_tables_vs_oracle(gpu, tmp_path, scope, batch, transforms, score_table, nproto, pdf):
    """Many prototypes with d(scores) accumulated on per-pixel rows (swav_args['ds_table']): 3 patches of 600
    of the 1024 pixels, so most pixels are drawn by more than one patch; two steps against the oracle, and the same step
    with the per-sample backward GEMMs.  6148 prototypes: the 8192-column forms of the row kernels (the generic configuration's
    8000 prototypes take them), where the per-sample form is the two-kernel CE; 'image' source pdf (the cat configuration): the row
    marginals follow the samples through the sorted row maps of the score tables"""
    from oracle import swav_ref as S
    from ganecdotes_amd import ops
    from ganecdotes_amd.draws import draw_image
    d = load_golden("gv10_pretrain")
    sd = golden_sd(d)
    g = mini_gen(sd, gpu)
    kw = dict(sinkhorn_scope=scope, batch_size=batch, nprototypes=nproto, nclasses=16, num_patches=3, patch_size=600,
              transforms=transforms, sinkhorn_args=dict(source_pdf=pdf))
    sw = make_swav(g, gpu, tmp_path, ds_table=True, score_table=score_table, **kw)       # score_table: the scores per pixel too
    sw2 = make_swav(g, gpu, tmp_path, ds_table=False, **kw)
    torch.manual_seed(123)
    np.random.seed(123)
    ml = S.G.mean_latent(sd, 64, 16)
    sw.mean_latent = sw2.mean_latent = ml.cuda()
    sw.build_heads()
    sw2.build_heads()
    for a, b in zip(sw2._params(), sw._params()):
        a.copy_(b)
    params = S.SwavParams(*[p.detach().cpu().clone() for p in sw._params()])
    bufs = [None] * 3
    for step in range(2):
        images = [draw_image(16, 7, 3, 32 * 32, 3, transforms=transforms) for _ in range(batch)]
        loss = sw.train_step(images)
        loss2 = sw2.train_step(images)
        oimgs = [S.ImageDraws(im.z, [S.ViewDraws(v.layer_no, v.map_noise, v.angle, v.flip,
                                                 None if v.angle is None else
                                                 ops.rotflip_map(v.angle, v.flip, 32, 32, "cuda").cpu().long())
                                     for v in im.views], im.picks) for im in images]
        ref_loss, grads, bufs = S.swav_train_step(
            sd, ml, 0.7, oimgs, params, bufs, lr=0.01, momentum=0.9, trust=0.01, use_larc=True, hlen=56,
            patch_size=600, num_patches=3, temperature=0.1, eps=0.05, niters=10, source_pdf=pdf, sinkhorn_scope=scope)
        np.testing.assert_allclose(loss.item(), ref_loss, rtol=1e-3)
        np.testing.assert_allclose(loss.item(), loss2.item(), rtol=1e-5)
        for got, other, want in zip(sw._grads, sw2._grads, grads):
            assert rel(got, want) < 2e-3 and rel(got, other) < 1e-4
        for got, want in zip(sw._params(), (params.proj_w, params.proto_w, params.proto_b)):
            assert rel(got, want) < 1e-4


@pytest.mark.parametrize("scope,batch,nproto,pdf", [("global", 3, 200, "uniform"), ("image", 2, 200, "uniform"),
                                                    ("global", 2, 203, "uniform"), ("image", 2, 4001, "uniform"),
                                                    ("global", 2, 203, "image")])
def test_train_step_tables_at_the_real_eps_and_temperature_vs_oracle(gpu, tmp_path, scope, batch, nproto, pdf):
    """The per-pixel table route (score tables + table Sinkhorn) with the configurations' OWN eps = 0.005
    and T = 0.01 (hfc_with_swav_ffhq_config.py:53, :65) — the toy tests above run at eps 0.05 / T 0.1, where the codes are soft —
    on 3 overlapping patches of 600 of 1024 pixels (most pixels drawn twice or more): two steps against the oracle, gradients
    to 1e-3.  nprototypes 203 / 4001: not a multiple of 4 — the prototype columns are padded to 204 /
    4004 with columns that take no mass (SwAVClustering._pad_prototypes; item 5), the oracle solves the unpadded problem."""
    from oracle import swav_ref as S
    from ganecdotes_amd import ops
    from ganecdotes_amd.draws import draw_image
    d = load_golden("gv10_pretrain")
    sd = golden_sd(d)
    g = mini_gen(sd, gpu)
    sw = make_swav(g, gpu, tmp_path, sinkhorn_scope=scope, batch_size=batch, nprototypes=nproto, nclasses=16, num_patches=3,
                   patch_size=600, temperature=0.01, ds_table=True, sinkhorn_args=dict(source_pdf=pdf, eps=0.005))
    torch.manual_seed(77)
    np.random.seed(77)
    ml = S.G.mean_latent(sd, 64, 16)
    sw.mean_latent = ml.cuda()
    sw.build_heads()
    params = S.SwavParams(*[p.detach().cpu().clone() for p in sw._params()])
    bufs = [None] * 3
    for step in range(2):
        images = [draw_image(16, 7, 3, 32 * 32, 3, transforms=True) for _ in range(batch)]
        loss = sw.train_step(images)
        rt = sw.last_route
        assert rt.k_run == (nproto + 3) // 4 * 4, rt
        if ops.get_gemm_mode() == 1:      # (GX_GEMM_MODE=0, the exact-fp32 engine, has no weight-plane GEMMs: it runs per sample)
            assert rt.name == "tables" and rt.score_table and rt.sinkhorn_table, rt
        oimgs = [S.ImageDraws(im.z, [S.ViewDraws(v.layer_no, v.map_noise, v.angle, v.flip,
                                                 ops.rotflip_map(v.angle, v.flip, 32, 32, "cuda").cpu().long())
                                     for v in im.views], im.picks) for im in images]
        ref_loss, grads, bufs = S.swav_train_step(
            sd, ml, 0.7, oimgs, params, bufs, lr=0.01, momentum=0.9, trust=0.01, use_larc=True, hlen=56,
            patch_size=600, num_patches=3, temperature=0.01, eps=0.005, niters=10, source_pdf=pdf, sinkhorn_scope=scope,
            fp64=True)     # the toy generator's cosines pass 0.44: the reference's un-shifted fp32 exp(scores / 0.005) overflows
        #                    there (GV7 documents its NaN); fp64 runs the same literal algorithm where it is finite
        assert np.isfinite(ref_loss)
        np.testing.assert_allclose(loss.item(), ref_loss, rtol=1e-3)
        for got, want in zip(sw._grads, grads):
            assert got.shape == want.shape and rel(got, want) < 1e-3
        for got, want in zip(sw._params(), (params.proj_w, params.proto_w, params.proto_b)):
            assert rel(got, want) < 1e-4


def test_prototype_count_not_a_multiple_of_four_costs_what_the_next_multiple_does(gpu, tmp_path):
    """K = 4001 takes the route K = 4004 takes (padded columns, `Route.k_run`), not the team-per-row kernels it used to fall to
    (132 ms per Sinkhorn against 10 ms at 4000), and returns K-shaped gradients; the per-sample route pads too
    (K = 201 without the tables)."""
    from ganecdotes_amd.draws import draw_image
    d = load_golden("gv10_pretrain")
    g = mini_gen(golden_sd(d), gpu)
    for nproto, extra, name in ((4001, dict(ds_table=True), "tables"), (201, dict(ds_table=False), "per_sample")):
        sw = make_swav(g, gpu, tmp_path, sinkhorn_scope="global", batch_size=2, nprototypes=nproto, nclasses=16, num_patches=3,
                       patch_size=600, **extra)
        torch.manual_seed(3)
        np.random.seed(3)
        sw.build_heads()
        images = [draw_image(16, 7, 3, 32 * 32, 3, transforms=True) for _ in range(2)]
        loss, aux = sw.train_step(images, return_aux=True)
        rt = sw.last_route
        assert rt.k_run == (nproto + 3) // 4 * 4
        from ganecdotes_amd import ops
        if ops.get_gemm_mode() == 1:
            assert rt.name == name and rt.pad_ds
        assert sw._grads[1].shape == (nproto, 16) and sw._grads[2].shape == (nproto,)
        assert aux["q_s"].shape[1] == nproto and aux["scores_s"].shape[1] == nproto
        # the padded columns took no mass: the real columns of every code row sum to 1
        q = aux["q_s"].double()
        assert float((q.sum(1) - 1).abs().max()) < 1e-5
        assert torch.isfinite(loss).all() and all(torch.isfinite(g_).all() for g_ in sw._grads)


@pytest.mark.parametrize("nproto", [6148, 7001])
def test_train_step_past_the_fused_ce_limit_vs_oracle(gpu, tmp_path, nproto):
    """More prototypes than the one-pass CE takes (K > 6144): the step falls back to gx_swapped_ce + gx_colsum.  With
    K % 4 == 0 its dS rows are padded to whole 16-column chunks (row-strided views, the A operand of the dz GEMM) — both
    kernels take the stride (this shape raised 'expected a contiguous tensor'); K = 7001 is the unpadded
    form.  One step against the oracle."""
    from oracle import swav_ref as S
    from ganecdotes_amd import ops
    from ganecdotes_amd.draws import draw_image
    d = load_golden("gv10_pretrain")
    sd = golden_sd(d)
    g = mini_gen(sd, gpu)
    sw = make_swav(g, gpu, tmp_path, sinkhorn_scope="global", batch_size=2, nprototypes=nproto, nclasses=16)
    torch.manual_seed(321)
    np.random.seed(321)
    ml = S.G.mean_latent(sd, 64, 16)
    sw.mean_latent = ml.cuda()
    sw.build_heads()
    params = S.SwavParams(*[p.detach().cpu().clone() for p in sw._params()])
    images = [draw_image(16, 7, 3, 32 * 32, 2, transforms=True) for _ in range(2)]
    loss = sw.train_step(images)
    oimgs = [S.ImageDraws(im.z, [S.ViewDraws(v.layer_no, v.map_noise, v.angle, v.flip,
                                             ops.rotflip_map(v.angle, v.flip, 32, 32, "cuda").cpu().long())
                                 for v in im.views], im.picks) for im in images]
    ref_loss, grads, _ = S.swav_train_step(
        sd, ml, 0.7, oimgs, params, [None] * 3, lr=0.01, momentum=0.9, trust=0.01, use_larc=True, hlen=56,
        patch_size=300, num_patches=2, temperature=0.1, eps=0.05, niters=10, source_pdf="uniform", sinkhorn_scope="global")
    np.testing.assert_allclose(loss.item(), ref_loss, rtol=1e-3)
    for got, want in zip(sw._grads, grads):
        assert rel(got, want) < 2e-3


@pytest.mark.parametrize("mode", ["pyramid", "panel"])
def test_patch_sampling_vs_oracle(gpu, tmp_path, mode):
    """sampling_method='patch' (swav_clustering.py:150-158, :383-385): one np.random.choice(h - patch_size)
    offset per patch, the pixels of the square crop [o:o+ps, o:o+ps]; loss, gradients and update against
    the oracle's crop on identical draws (both rotated / flipped views, batch 2)."""
    from oracle import swav_ref as S
    from ganecdotes_amd import ops
    d = load_golden("gv10_pretrain")
    sd = golden_sd(d)
    g = mini_gen(sd, gpu)
    sw = make_swav(g, gpu, tmp_path, sampling_method="patch", patch_size=12, batch_size=2, sinkhorn_scope="image",
                   projection_mode=mode)
    torch.manual_seed(321)
    np.random.seed(321)
    ml = S.G.mean_latent(sd, 64, 16)
    sw.mean_latent = ml.cuda()
    sw.build_heads()
    params = S.SwavParams(*[p.detach().cpu().clone() for p in sw._params()])
    images = sw.draw_batch(2)
    assert all(len(im.patch_offsets) == 2 and all(0 <= o < 32 - 12 for o in im.patch_offsets) for im in images)
    assert all(p.numel() == 144 for im in images for p in im.picks)
    loss = sw.train_step(images)
    oimgs = [S.ImageDraws(im.z, [S.ViewDraws(v.layer_no, v.map_noise, v.angle, v.flip,
                                             ops.rotflip_map(v.angle, v.flip, 32, 32, "cuda").cpu().long())
                                 for v in im.views], list(im.patch_offsets)) for im in images]
    ref_loss, grads, _ = S.swav_train_step(
        sd, ml, 0.7, oimgs, params, [None] * 3, lr=0.01, momentum=0.9, trust=0.01, use_larc=True, hlen=56,
        patch_size=12, num_patches=2, temperature=0.1, eps=0.05, niters=10, source_pdf="uniform",
        sinkhorn_scope="image")
    np.testing.assert_allclose(loss.item(), ref_loss, rtol=1e-3)
    for got, want in zip(sw._grads, grads):
        assert rel(got, want) < 2e-3
    for got, want in zip(sw._params(), (params.proj_w, params.proto_w, params.proto_b)):
        assert rel(got, want) < 1e-4


@pytest.mark.parametrize("hlen", [512 * 8 + 300, 512 * 7])
def test_pyramid_equals_panel_at_ffhq_width(gpu, tmp_path, hlen):
    """Full-width 64x64 generator (512-channel taps): the pyramid projection and the
    reference-shaped panel GEMM give the same loss / gradients / predictions.  hlen cuts inside the
    last tap, or drops the whole finest level (the car-512 case: a true 512^2 generator has
    sum(C) = 5504 > hlen = 5376, so its two native-resolution taps are never used)."""
    from ganecdotes_amd.draws import draw_image
    from ganecdotes_amd.generator import Generator
    from ganecdotes_amd.swav_clustering import SwAVClustering
    torch.manual_seed(3)
    np.random.seed(3)
    g = Generator(64, 512, 8)
    for n, p in g.named_parameters():
        if n.endswith("noise.weight"):
            p.data.fill_(0.1)
    g = g.eval().to(gpu)

    class C:
        num_latents_for_mean, truncation, latent_dim = 128, 0.7, 512
    out = {}
    imgs = [draw_image(512, 9, 4, 64 * 64, 2) for _ in range(2)]
    wlat = g.style(torch.randn(1, 512, device="cuda"))
    ml = g.mean_latent(128)
    for mode in ("panel", "pyramid"):
        a = toy_args(projection_mode=mode, hlen=hlen, nclasses=64, nprototypes=40, patch_size=1500,
                     sinkhorn_scope="global", temperature=0.1)
        a["perturb_args"] = dict(truncation=0.7, n_layers=4, n_samples=1, layer_no=None, perturb_std=[1.0] * 4)
        a["sinkhorn_args"] = dict(source_pdf="uniform", niters=10, eps=0.05)
        sw = SwAVClustering(model=g, model_config=C, out_dir=None, device="cuda", **a)
        sw.mean_latent = ml
        torch.manual_seed(11)
        sw.build_heads()
        loss = sw.train_step(imgs)
        preds, labels = sw.predict_swav_codes(wlat)
        out[mode] = (loss.item(), [x.clone() for x in sw._grads], preds.clone(), labels.clone())
    np.testing.assert_allclose(out["pyramid"][0], out["panel"][0], rtol=1e-4)
    for a, b in zip(out["pyramid"][1], out["panel"][1]):
        assert rel(a, b) < 1e-3
    assert rel(out["pyramid"][2], out["panel"][2]) < 1e-4
    assert (out["pyramid"][3] == out["panel"][3]).float().mean() > 0.999


@pytest.mark.parametrize("nproto", [8, 16, 11])
def test_small_prototype_paths_agree(gpu, tmp_path, nproto):
    """D=256, K=8 / 16 (BASELINE configs[1] / [2]) / 11 on a 512-channel generator: the generic GEMM
    chain, the fused small-prototype kernels and the low-rank coefficient path (score table of
    every pixel + per-patch gathers) give the same loss, gradients and update."""
    from ganecdotes_amd.draws import draw_image
    from ganecdotes_amd.generator import Generator
    from ganecdotes_amd.swav_clustering import SwAVClustering
    torch.manual_seed(5)
    np.random.seed(5)
    g = Generator(32, 512, 8)
    for n, p in g.named_parameters():
        if n.endswith("noise.weight"):
            p.data.fill_(0.1)
    g = g.eval().to(gpu)

    class C:
        num_latents_for_mean, truncation, latent_dim = 128, 0.7, 512
    imgs = [draw_image(512, 7, 3, 32 * 32, 3) for _ in range(3)]
    ml = g.mean_latent(128)
    out = {}
    for tag, kw in (("generic", dict(small_proto_kernels=False, lowrank_proto=False)),
                    ("small", dict(small_proto_kernels=True, lowrank_proto=False)),
                    ("lowrank", dict(small_proto_kernels=True, lowrank_proto=True))):
        a = toy_args(hlen=512 * 7, nclasses=256, nprototypes=nproto, patch_size=700, num_patches=3,
                     sinkhorn_scope="global", temperature=0.1, **kw)
        a["sinkhorn_args"] = dict(source_pdf="uniform", niters=10, eps=0.05)
        sw = SwAVClustering(model=g, model_config=C, out_dir=None, device="cuda", **a)
        sw.mean_latent = ml
        torch.manual_seed(11)
        sw.build_heads()
        l1 = sw.train_step(imgs).item()
        l2 = sw.train_step(imgs).item()
        out[tag] = (l1, l2, [x.clone() for x in sw._grads], [x.clone() for x in sw._params()])
    for tag in ("small", "lowrank"):
        np.testing.assert_allclose(out[tag][:2], out["generic"][:2], rtol=1e-4)
        for a_, b_ in zip(out[tag][2], out["generic"][2]):
            assert rel(a_, b_) < 1e-3, tag
        for a_, b_ in zip(out[tag][3], out["generic"][3]):
            assert rel(a_, b_) < 1e-4, tag


def test_host_running_ahead_keeps_batches_intact(gpu, tmp_path):
    """60 optimizer steps queued without a host sync, device-drawn picks: the side-stream-owned staging /
    pick buffers of batch i must not be recycled for batch i+1 while kernels of step i are still queued
    (PreparedBatch.acquire records the consuming stream); the run matches the same run synced every step."""
    d = load_golden("gv10_pretrain")
    g = mini_gen(golden_sd(d), gpu)
    ml = g.mean_latent(64)
    out = []
    for sync in (False, True):
        sw = make_swav(g, gpu, tmp_path, pick_source="device", batch_size=4, sinkhorn_scope="global")
        sw.mean_latent = ml
        torch.manual_seed(77)
        np.random.seed(77)
        sw.build_heads()
        losses = []
        for _ in range(60):
            losses.append(sw.train_step(sw.draw_batch(4)))
            if sync:
                torch.cuda.synchronize()
        out.append(torch.cat(losses).cpu().numpy())
    assert np.isfinite(out[0]).all()
    np.testing.assert_allclose(out[0], out[1], rtol=2e-3)


def test_views_prefetched_on_their_own_stream_keep_the_trajectory(gpu, tmp_path):
    """The views of batch i + 1 queued on the view stream BEFORE step i (ViewMixin.prefetch_views — what pretrain() and
    bench.py do): 60 steps without a host sync, the host running ahead, taps / maps / picks crossing from the view stream's
    memory pool to the step's stream.  Same draws, same arithmetic: the losses equal the one-stream run's, step by step; so
    do the trained parameters."""
    d = load_golden("gv10_pretrain")
    g = mini_gen(golden_sd(d), gpu)
    ml = g.mean_latent(64)
    out, par = [], []
    for overlap in (True, False):
        sw = make_swav(g, gpu, tmp_path, pick_source="device", batch_size=4, sinkhorn_scope="global")
        sw.mean_latent = ml
        torch.manual_seed(78)
        np.random.seed(78)
        sw.build_heads()
        losses = []
        stage = sw.prefetch_views if overlap else sw.prepare_batch
        nxt = stage(sw.draw_batch(4))
        for i in range(60):
            cur = nxt
            nxt = stage(sw.draw_batch(4)) if i < 59 else None
            if overlap:
                assert cur.views is not None
            losses.append(sw.train_step(cur))
            assert cur.views is None                      # taken (or never made)
        out.append(torch.cat(losses).cpu().numpy())
        par.append([p.clone() for p in sw._params()])
    assert np.isfinite(out[0]).all()
    np.testing.assert_allclose(out[0], out[1], rtol=2e-3)
    for a_, b_ in zip(*par):
        assert rel(a_, b_) < 1e-3


def test_no_cpu_fallback(tmp_path):
    from ganecdotes_amd.generator import Generator
    g = Generator(32, 16, 2, channels=MINI)
    with pytest.raises(RuntimeError):
        g([torch.randn(1, 16)])
    from ganecdotes_amd.swav_clustering import SwAVClustering
    with pytest.raises(RuntimeError):
        SwAVClustering(model=g, model_config=Cfg, out_dir=str(tmp_path), device="cpu", **toy_args())


# ------------------------------------------------------------------------------------- SimCLR (config 5)
def test_simclr_head_kernels_vs_autograd(gpu):
    from ganecdotes_amd import ops
    from oracle import simclr_ref as C
    torch.manual_seed(31)
    n, d = 40, 64
    h1 = torch.randn(n, d)
    gamma, beta = torch.rand(d) + 0.5, 0.1 * torch.randn(d)
    w2 = torch.randn(d, d) / 8
    h1r = h1.clone().requires_grad_(True)
    gr, br, w2r = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True), w2.clone().requires_grad_(True)
    mean = h1r.mean(0)
    var = h1r.var(0, unbiased=False)
    a_ref = torch.nn.functional.leaky_relu((h1r - mean) * torch.rsqrt(var + 1e-5) * gr + br, 0.01)
    out_ref = a_ref @ w2r.t()
    loss_ref = C.ntxent_feature_rows(out_ref, n // 2, 0.7)
    loss_ref.backward()
    d_ = lambda t: t.cuda().contiguous()
    m, v = ops.col_stats(d_(h1))
    a, xhat = ops.bn_lrelu_fwd(d_(h1), m, v, d_(gamma), d_(beta))
    out = ops.gemm(a, d_(w2), False, True)
    loss, dout = ops.simclr_ntxent(out, 0.7)
    np.testing.assert_allclose(loss.item(), loss_ref.item(), rtol=1e-5)
    gw2 = ops.gemm(dout, a, True, False)
    da = ops.gemm(dout, d_(w2), False, False)
    dh1, dg, db = ops.bn_lrelu_bwd(da, xhat, v, d_(gamma), d_(beta))
    assert rel(gw2, w2r.grad) < 1e-4 and rel(dh1, h1r.grad) < 1e-3
    assert rel(dg, gr.grad) < 1e-4 and rel(db, br.grad) < 1e-4


def test_simclr_iteration_and_predict_with_bilinear_hf_interp_vs_oracle(gpu, tmp_path):
    """simclr_args['hf_interp'] = 'bilinear' (baseline/hfc_with_simclr/simclr_clustering.py:102-103): two iterations (loss, the
    four gradients) and predict_simclr_codes against the oracle on torch's own bilinear stack."""
    from oracle import simclr_ref as C
    from oracle import swav_ref as S
    from ganecdotes_amd import ops
    from ganecdotes_amd.simclr_clustering import SimCLRClustering
    d = load_golden("gv12_simclr")
    sd = golden_sd(d)
    g = mini_gen(sd, gpu)
    args = dict(num_iters=2, batch_size=4, patch_size=30, hf_interp="bilinear", trust_coeff=0.01,
                train_args=dict(lr=0.01, momentum=0.9), temperature=1.0, nclasses=8, hlen=56, epoch_print_freq=1,
                max_masks=4, use_larc=True, transforms=True)
    sim = SimCLRClustering(model=g, model_config=Cfg, out_dir=None, device="cuda", simclr_args=args,
                           perturb_args=dict(truncation=0.7, n_layers=3, n_samples=1, layer_no=None, perturb_std=[1.0] * 3))
    torch.manual_seed(5)
    np.random.seed(5)
    ml = S.G.mean_latent(sd, 64, 16)
    sim.mean_latent = ml.cuda()
    sim.build_heads()
    leaves = [p.detach().cpu().clone() for p in sim._params()]
    for it in range(2):
        im = sim.draw_iteration()
        loss = sim.train_step(im)
        oim = S.ImageDraws(im.z, [S.ViewDraws(v.layer_no, v.map_noise, v.angle, v.flip,
                                              ops.rotflip_map(v.angle, v.flip, 32, 32, "cuda").cpu().long()) for v in im.views], im.picks)
        for p in leaves:
            p.requires_grad_(True)
            p.grad = None
        ref, _, _ = C.simclr_step_loss(sd, ml, 0.7, oim, *leaves, hlen=56, batch=4, temperature=1.0, hf_interp="bilinear")
        ref.backward()
        np.testing.assert_allclose(loss.item(), float(ref.detach()), rtol=1e-3)
        for got, p in zip(sim._grads, leaves):
            assert rel(got, p.grad) < 2e-3
        for p, q in zip(leaves, sim._params()):           # continue from the HIP parameters
            p.requires_grad_(False)
            p.copy_(q.cpu())
    w = S.G.mapping(torch.randn(1, 16), sd)
    params = dict(w1=leaves[0], gamma=leaves[1], beta=leaves[2], w2=leaves[3])
    preds, labels = sim.predict_simclr_codes(w.cuda())
    ref_preds, ref_labels = C.predict_simclr_codes(sd, w, ml, 0.7, params, 56, hf_interp="bilinear")
    assert rel(preds, ref_preds) < 1e-3
    assert (labels.cpu() == ref_labels).float().mean() > 0.99
    assert rel(preds, C.predict_simclr_codes(sd, w, ml, 0.7, params, 56)[0]) > 1e-2      # not the nearest stack


def test_simclr_pretrain_golden_gv12(gpu, tmp_path):
    from ganecdotes_amd.simclr_clustering import SimCLRClustering
    d = load_golden("gv12_simclr")
    g = mini_gen(golden_sd(d), gpu)
    args = dict(num_iters=2, batch_size=3, patch_size=30, hf_interp="nearest", trust_coeff=0.01,
                train_args=dict(lr=0.01, momentum=0.9), temperature=1.0, nclasses=8, hlen=56, epoch_print_freq=1,
                max_masks=4, use_larc=False, transforms=False, cpu_reference_rng=True)
    sim = SimCLRClustering(model=g, model_config=Cfg, out_dir=str(tmp_path), device="cuda", simclr_args=args,
                           perturb_args=dict(truncation=0.7, n_layers=3, n_samples=1, layer_no=None,
                                             perturb_std=[1.0] * 3))
    torch.manual_seed(42)
    np.random.seed(42)
    z = torch.randn(64, 16)                               # the reference constructor's mean_latent draw
    sim.mean_latent = g.style(z.cuda()).mean(0, keepdim=True)
    sim.pretrain(None)
    np.testing.assert_allclose([l.item() for l in sim.loss_history], d["losses"], rtol=1e-3)
    sdp = sim.projection.state_dict()
    for key in ("0.weight", "1.weight", "1.bias", "3.weight", "1.running_mean", "1.running_var"):
        assert rel(sdp[key], T(d["proj_" + key])) < 2e-3, key
    assert int(sdp["1.num_batches_tracked"]) == 2
    sim.mean_latent = T(d["mean_latent"]).cuda()
    for key in ("0.weight", "1.weight", "1.bias", "3.weight"):
        sdp[key].copy_(T(d["proj_" + key]).cuda())
    preds, labels = sim.predict_simclr_codes(T(d["pred_w"]).cuda())
    assert preds.shape == d["preds"].shape
    assert rel(preds, T(d["preds"])) < 1e-3
    assert (labels.cpu().numpy() == d["labels"]).mean() > 0.99
    assert rel(sim.projection[1].running_mean, T(d["proj_after_predict_1.running_mean"])) < 2e-3
    proj = torch.load(tmp_path / "projection.pt", weights_only=False)
    assert proj(torch.zeros(4, 56, device="cuda")).shape == (4, 8)
