"""The oracle (CPU restatement) against the golden vectors captured from the
reference's own functions (tools/capture_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import golden_sd, load_golden
from oracle import stylegan2_ref as G
from oracle import swav_ref as S

T = torch.from_numpy


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else a
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_gv1_upfirdn2d():
    d = load_golden("gv1_upfirdn2d")
    k = T(d["kernel"])
    close(G.fir_kernel(), d["kernel"])
    close(G.upfirdn2d(T(d["x_mode1"]), k * 4, up=1, pad=(1, 1)), d["y_mode1"])
    close(G.upfirdn2d(T(d["x_mode3"]), k * 4, up=2, pad=(2, 1)), d["y_mode3"])


def test_gv2_modconv():
    d = load_golden("gv2_modconv")
    blur = G.fir_kernel() * 4
    for tag, kw in (("plain", {}), ("up", dict(upsample=True, blur_kernel=blur)),
                    ("rgb", dict(demodulate=False))):
        y = G.modulated_conv(T(d[f"{tag}_x"]), T(d[f"{tag}_style"]), T(d[f"{tag}_weight"]),
                             T(d[f"{tag}_mod_weight"]), T(d[f"{tag}_mod_bias"]), **kw)
        close(y, d[f"{tag}_y"], rtol=1e-4, atol=1e-5)


def test_gv3_layers():
    d = load_golden("gv3_layers")
    for tag, up in (("plain", False), ("up", True)):
        sd = {k[len(tag) + 4:]: T(v) for k, v in d.items() if k.startswith(f"{tag}_sd_")}
        y = G.styled_conv(T(d[f"{tag}_x"]), T(d[f"{tag}_style"]), sd, "", T(d[f"{tag}_noise"]), up)
        close(y, d[f"{tag}_y"], rtol=1e-4, atol=1e-5)
    sd = {k[7:]: T(v) for k, v in d.items() if k.startswith("rgb_sd_")}
    close(G.to_rgb(T(d["rgb_x"]), T(d["rgb_style"]), sd, "", T(d["rgb_skip"])), d["rgb_y_skip"],
          rtol=1e-4, atol=1e-5)
    close(G.to_rgb(T(d["rgb_x"]), T(d["rgb_style"]), sd, ""), d["rgb_y_noskip"], rtol=1e-4, atol=1e-5)
    x = T(d["el_x"])
    close(G.equal_linear(x, T(d["el_weight"]), T(d["el_bias"]), 0.01, True), d["el_y"])
    close(G.equal_linear(x, T(d["el2_weight"]), T(d["el2_bias"])), d["el2_y"])
    close(G.pixel_norm(x), d["pn_y"])


def test_gv4_mini_generator():
    d = load_golden("gv4_mini_generator")
    sd = golden_sd(d)
    w, ml = T(d["w"]), T(d["mean_latent"])
    close(G.mapping(T(d["z"]), sd), d["w"], rtol=1e-5, atol=1e-6)
    img, taps, lat = G.generator_forward(w, sd, 0.7, ml, input_is_latent=True)
    close(lat, d["latent_trunc_once"])
    close(img, d["img_once"], rtol=1e-4, atol=1e-5)
    for i, t in enumerate(taps):
        close(t, d[f"tap_once_{i}"], rtol=1e-4, atol=1e-5)
    img2, taps2, _ = G.generator_forward(lat, sd, 0.7, ml, input_is_latent=True)
    close(img2, d["img_twice"], rtol=1e-4, atol=1e-5)
    for i, t in enumerate(taps2):
        close(t, d[f"tap_twice_{i}"], rtol=1e-4, atol=1e-5)
    imgz, tapsz, _ = G.generator_forward(T(d["z"]), sd, 1.0, None, input_is_latent=False)
    close(imgz, d["img_z"], rtol=1e-4, atol=1e-5)
    assert len(taps) == 7


def test_gv5_pixel_features():
    d = load_golden("gv5_pixel_features")
    taps = [T(d[f"tap_{i}"]) for i in range(4)]
    close(S.pixel_features(taps, int(d["hlen"])), d["stack"], 0, 0)
    close(S.pixel_features(taps, 42), d["stack_full"], 0, 0)


def test_gv6_codes():
    d = load_golden("gv6_codes")
    x = S.sample_pixels(T(d["hfeat"]), T(d["picks"]), int(d["patch_size"]))
    close(S.codes(x, T(d["proj_weight"]), T(d["proto_weight"]), T(d["proto_bias"])), d["scores_train"])
    preds, labels = None, None
    xa = S.sample_pixels(T(d["hfeat"]), None, None)
    preds = S.codes(xa, T(d["proj_weight"]), train=False).reshape(d["scores_pred"].shape)
    close(preds, d["scores_pred"])
    assert (preds.max(1)[1].numpy() == d["labels_pred"]).all()


def test_gv7_sinkhorn():
    d = load_golden("gv7_sinkhorn")
    sc, img = T(d["scores"]), T(d["img"])
    for tag, pdf in (("uniform_e05", "uniform"), ("image_e05", "image"),
                     ("uniform_e005", "uniform"), ("image_e003", "image")):
        s = sc * float(d[f"scale_{tag}"])
        eps = float(d[f"eps_{tag}"])
        q = S.sinkhorn_knopp(s, eps, int(d["niters"]), pdf, img)
        close(q, d[f"q_{tag}"], rtol=1e-5, atol=1e-9)
        # the fp64 run of the same algorithm is the same matrix on what the loss sees
        qs = S.sinkhorn_knopp(s, eps, int(d["niters"]), pdf, img, fp64=True).numpy()
        ref = d[f"q_{tag}"]
        assert np.all(np.abs(qs - ref) <= 1e-3 * ref.max(axis=1, keepdims=True))
    close(S.sinkhorn_knopp(T(d["scores_big"]), 0.01, 10), d["q_big_uniform_e01"], rtol=1e-5, atol=1e-9)
    # the reference overflows (non-finite) on un-shifted fp32 exp; the fp64 run stays finite
    assert not bool(d["overflow_all_finite"])
    assert not torch.isfinite(S.sinkhorn_knopp(sc * float(d["overflow_scale"]), 0.005, 10)).all()
    q_over = S.sinkhorn_knopp(sc * float(d["overflow_scale"]), 0.005, 10, fp64=True)
    assert torch.isfinite(q_over).all()
    close(q_over.sum(1), np.ones(sc.shape[0], np.float32), rtol=1e-5)


def test_gv8_loss_and_grads():
    d = load_golden("gv8_loss_grads")
    pw = T(d["proj_weight"]).clone().requires_grad_(True)
    cw = T(d["proto_weight"]).clone().requires_grad_(True)
    cb = T(d["proto_bias"]).clone().requires_grad_(True)
    n = d["scores_s"].shape[0]
    ss = S.codes(S.sample_pixels(T(d["hs"]), T(d["picks"]), n), pw, cw, cb)
    st = S.codes(S.sample_pixels(T(d["ht"]), T(d["picks"]), n), pw, cw, cb)
    close(ss, d["scores_s"])
    close(st, d["scores_t"])
    with torch.no_grad():
        qs = S.sinkhorn_knopp(ss, float(d["eps"]), 10)
        qt = S.sinkhorn_knopp(st, float(d["eps"]), 10)
    close(qs, d["q_s"], rtol=1e-5, atol=1e-9)
    tt = float(d["temperature"])
    loss = S.swapped_prediction_loss(ss / tt, st / tt, qs, qt)
    close(loss, d["loss"], rtol=1e-6)
    loss.backward()
    close(pw.grad, d["g_proj"], rtol=1e-4, atol=1e-7)
    close(cw.grad, d["g_proto_w"], rtol=1e-4, atol=1e-7)
    close(cb.grad, d["g_proto_b"], rtol=1e-4, atol=1e-7)


def test_gv9_view_and_rng_order():
    d = load_golden("gv9_view_rng")
    sd = golden_sd(d)
    torch.manual_seed(42)
    np.random.seed(42)
    ml = G.mean_latent(sd, 64, 16)
    close(ml, d["mean_latent"], rtol=1e-5, atol=1e-6)
    im = S.draw_image_reference_order(16, G.num_layers_of(sd), 3, 32 * 32, 0, transforms=False)
    close(im.z, d["z"], 0, 0)
    assert torch.equal(torch.randn(4), T(d["after"]))           # RNG state identical afterwards
    w = G.mapping(im.z, sd)
    close(w, d["w"], rtol=1e-5, atol=1e-6)
    for v, tag in zip(im.views, "st"):
        assert v.layer_no == int(d[f"layer_no_{tag}"])
        stack, img = S.make_view(sd, w, ml, 0.7, v, hlen=56)
        close(stack, d[f"stack_{tag}"], rtol=1e-4, atol=1e-5)
        close(img, d[f"img_{tag}"], rtol=1e-4, atol=1e-5)


def _gv10_run(d):
    sd = golden_sd(d)
    torch.manual_seed(42)
    np.random.seed(42)
    return S.pretrain_reference_order(
        sd, 16, num_latents_for_mean=64, psi=0.7, n_perturb_layers=3, num_test_samples=2,
        num_epochs=3, num_patches=2, patch_size=300, hlen=56, nclasses=8, nprototypes=12,
        temperature=0.1, eps=0.05, niters=10, lr=0.01, momentum=0.9, trust=0.01,
        use_larc=False, transforms=False)


def test_gv10_pretrain_trajectory():
    d = load_golden("gv10_pretrain")
    losses, params, _ = _gv10_run(d)
    np.testing.assert_allclose(losses, d["losses"], rtol=1e-4)
    close(params.proj_w, d["proj_weight"], rtol=1e-4, atol=1e-6)
    close(params.proto_w, d["proto_weight"], rtol=1e-4, atol=1e-6)
    close(params.proto_b, d["proto_bias"], rtol=1e-4, atol=1e-6)


def test_gv11_predict():
    d10, d = load_golden("gv10_pretrain"), load_golden("gv11_predict")
    sd = golden_sd(d10)
    preds, labels = S.predict_swav_codes(sd, T(d["w"]), T(d["mean_latent"]), 0.7,
                                         T(d10["proj_weight"]), hlen=56)
    close(preds, d["preds"], rtol=1e-4, atol=1e-5)
    assert (labels.numpy() == d["labels"]).mean() > 0.999


def _gv17_head(d, tag, kind, prefix="proj_", running_from=None):
    """the oracle's HeadParams from a captured projection state_dict"""
    if kind == "1-layer":
        return S.HeadParams(kind)
    r = running_from or prefix
    return S.HeadParams(kind, T(d[f"{tag}_{prefix}1.weight"]), T(d[f"{tag}_{prefix}1.bias"]), T(d[f"{tag}_{prefix}3.weight"]),
                        T(d[f"{tag}_{prefix}4.weight"]), T(d[f"{tag}_{prefix}4.bias"]),
                        [T(d[f"{tag}_{r}1.running_mean"]).clone(), T(d[f"{tag}_{r}1.running_var"]).clone(),
                         T(d[f"{tag}_{r}4.running_mean"]).clone(), T(d[f"{tag}_{r}4.running_var"]).clone(), 0])


@pytest.mark.parametrize("kind", ["1-layer", "2-layer"])
def test_gv17_projection_heads(kind):
    """projn_nw '1-layer' / '2-layer' (swav_clustering.py:250-269): the reference's own pretrain() trajectory, final
    modules and BatchNorm running statistics; its code calls on a fixed feature map; predict_swav_codes afterwards"""
    d = load_golden("gv17_projection_heads")
    tag = kind[0] + "l"
    sd = golden_sd(d)
    torch.manual_seed(42)
    np.random.seed(42)
    losses, params, _ = S.pretrain_reference_order(
        sd, 16, num_latents_for_mean=64, psi=0.7, n_perturb_layers=3, num_test_samples=2, num_epochs=3, num_patches=2,
        patch_size=300, hlen=56, nclasses=8, nprototypes=12, temperature=0.1, eps=0.05, niters=10, lr=0.01, momentum=0.9,
        trust=0.01, use_larc=False, transforms=False, projn_nw=kind)
    np.testing.assert_allclose(losses, d[f"{tag}_losses"], rtol=1e-4)
    close(params.proj_w, d[f"{tag}_proj_0.weight"], rtol=1e-4, atol=1e-6)
    close(params.proto_w, d[f"{tag}_proto_weight"], rtol=1e-4, atol=1e-6)
    close(params.proto_b, d[f"{tag}_proto_bias"], rtol=1e-4, atol=1e-6)
    if kind == "2-layer":
        h = params.head
        for got, key in ((h.bn1_w, "1.weight"), (h.bn1_b, "1.bias"), (h.w2, "3.weight"), (h.bn2_w, "4.weight"),
                         (h.bn2_b, "4.bias"), (h.running[0], "1.running_mean"), (h.running[1], "1.running_var"),
                         (h.running[2], "4.running_mean"), (h.running[3], "4.running_var")):
            close(got, d[f"{tag}_proj_{key}"], rtol=1e-4, atol=1e-6)
        assert h.running[4] == int(d[f"{tag}_proj_1.num_batches_tracked"]) == 12     # 3 epochs x 2 patches x 2 views
    # the code calls on a fixed feature map, from the CAPTURED modules
    head = _gv17_head(d, tag, kind)
    pw, cw, cb = T(d[f"{tag}_proj_0.weight"]), T(d[f"{tag}_proto_weight"]), T(d[f"{tag}_proto_bias"])
    stack, picks = T(d[f"{tag}_hfeat"]), T(d[f"{tag}_picks"]).long()
    with torch.no_grad():
        x = S.sample_pixels(stack, picks, 300)
        close(S.codes(x, pw, cw, cb, train=True, head=head), d[f"{tag}_scores_train"], rtol=1e-4, atol=1e-5)
        if kind == "2-layer":
            close(head.running[0], d[f"{tag}_proj_after_train_call_1.running_mean"], rtol=1e-4, atol=1e-6)
            close(head.running[3], d[f"{tag}_proj_after_train_call_4.running_var"], rtol=1e-4, atol=1e-6)
        xall = torch.squeeze(stack).flatten(1).t()
        close(S.codes(xall, pw, train=False, head=head).reshape(1, 8, 20, 20), d[f"{tag}_scores_pred"], rtol=1e-4, atol=1e-5)
        preds, labels = S.predict_swav_codes(sd, T(d[f"{tag}_pred_w"]), T(d[f"{tag}_mean_latent"]), 0.7, pw, hlen=56, head=head)
    close(preds, d[f"{tag}_preds"], rtol=1e-4, atol=1e-5)
    assert (labels.numpy() == d[f"{tag}_labels"]).mean() > 0.999
    if kind == "2-layer":
        close(head.running[0], d[f"{tag}_proj_after_predict_1.running_mean"], rtol=1e-4, atol=1e-6)
        close(head.running[1], d[f"{tag}_proj_after_predict_1.running_var"], rtol=1e-4, atol=1e-6)
        close(head.running[2], d[f"{tag}_proj_after_predict_4.running_mean"], rtol=1e-4, atol=1e-6)
        close(head.running[3], d[f"{tag}_proj_after_predict_4.running_var"], rtol=1e-4, atol=1e-6)


def test_rotflip_map_properties():
    m0 = S.rotflip_index_map(0.0, False, 16, 16)
    assert torch.equal(m0, torch.arange(256))
    mf = S.rotflip_index_map(0.0, True, 8, 12).reshape(8, 12)
    assert torch.equal(mf, torch.arange(96).reshape(8, 12).flip(1))
    m = S.rotflip_index_map(7.5, False, 64, 64)
    assert (m == -1).sum() > 0 and (m >= 0).float().mean() > 0.9
    # centre pixel neighbourhood maps to itself for a small angle
    c = 32 * 64 + 32
    assert abs(int(m[c]) - c) <= 65


def test_larc_restatement():
    torch.manual_seed(0)
    p, g = torch.randn(5, 7), torch.randn(5, 7)
    p0 = p.clone()
    bufs = S.larc_sgd_step([p], [g], [None], lr=0.01, momentum=0.9, trust=0.01)
    scale = 0.01 * p0.norm() / (g.norm() + 1e-8)
    close(p, (p0 - 0.01 * scale * g).numpy(), rtol=1e-6)
    p1 = p.clone()
    bufs2 = S.larc_sgd_step([p], [g], bufs, lr=0.01, momentum=0.9, trust=0.01)
    scale2 = 0.01 * p1.norm() / (g.norm() + 1e-8)
    close(bufs2[0], (0.9 * bufs[0] + scale2 * g).numpy(), rtol=1e-6)


def test_gv12_simclr_trajectory_and_predict():
    from oracle import simclr_ref as C
    d = load_golden("gv12_simclr")
    sd = golden_sd(d)
    torch.manual_seed(42)
    np.random.seed(42)
    losses, p, ml = C.pretrain_reference_order(
        sd, 16, num_latents_for_mean=64, psi=0.7, n_perturb_layers=3, num_iters=2, batch=3, hlen=56, nclasses=8,
        temperature=1.0, lr=0.01, momentum=0.9, trust=0.01, use_larc=False, transforms=False)
    np.testing.assert_allclose(losses, d["losses"], rtol=1e-4)
    close(p["w1"], d["proj_0.weight"], rtol=1e-4, atol=1e-6)
    close(p["gamma"], d["proj_1.weight"], rtol=1e-4, atol=1e-6)
    close(p["beta"], d["proj_1.bias"], rtol=1e-3, atol=1e-6)
    close(p["w2"], d["proj_3.weight"], rtol=1e-4, atol=1e-6)
    close(p["running_mean"], d["proj_1.running_mean"], rtol=1e-4, atol=1e-6)
    close(p["running_var"], d["proj_1.running_var"], rtol=1e-4, atol=1e-6)
    close(ml, d["mean_latent"], rtol=1e-5, atol=1e-6)
    preds, labels = C.predict_simclr_codes(sd, T(d["pred_w"]), ml, 0.7, p, 56)
    close(preds, d["preds"], rtol=1e-3, atol=1e-4)
    assert (labels.numpy() == d["labels"]).mean() > 0.99


# ------------------------------------------------------------------------------------- GV13: one-shot head
@pytest.mark.parametrize("size", ["XXS", "XS", "Lin"])
def test_oneshot_head_finetune_golden_gv13(size):
    from oracle import oneshot_ref as O
    d = load_golden("gv13_oneshot")
    feats, labels = torch.from_numpy(d[f"{size}_feats"]), torch.from_numpy(d[f"{size}_labels"])
    plan = O.segmentor_plan(size, 16, 5)
    keys = sorted((k for k in d if k.startswith(f"{size}_init_")),
                  key=lambda k: (int(k.split("layers.")[1].split(".")[0]), "bias" in k))
    params = [torch.from_numpy(d[k]) for k in keys]
    assert [tuple(p.shape) for p in params[::2]] == [(b, a, 3, 3) if dd else (b, a) for a, b, dd, _ in plan]
    np.testing.assert_allclose(O.segmentor_forward(feats, params, plan).numpy(), d[f"{size}_out0"], rtol=1e-5, atol=1e-6)
    losses, final = O.finetune(feats, labels, params, plan, epochs=len(d[f"{size}_losses"]), lr=float(d["lr"]),
                               beta1=float(d["beta1"]), beta2=float(d["beta2"]), step_size=int(d["step_size"]),
                               gamma=float(d["gamma"]))
    np.testing.assert_allclose(losses, d[f"{size}_losses"], rtol=1e-5)
    for k, p in zip(keys, final):
        np.testing.assert_allclose(p.numpy(), d[k.replace("_init_", "_final_")], rtol=1e-4, atol=1e-6)
    pred = O.segmentor_forward(feats, final, plan).max(1)[1].numpy()
    assert (pred == d[f"{size}_pred"]).mean() > 0.995


def test_iou_metrics_golden_gv13():
    from oracle import oneshot_ref as O
    d = load_golden("gv13_oneshot")
    classes = ["background", "a", "b", "c"]
    gt, pred = d["met_gt"], d["met_pred"]
    mi = {c: O.mask_iou(gt, pred, i) for i, c in enumerate(classes)}
    np.testing.assert_allclose([mi[c] for c in classes], d["met_mask_iou"], rtol=1e-12)
    np.testing.assert_allclose(O.weighted_iou(gt, mi, classes), float(d["met_w_iou"]), rtol=1e-12)
    np.testing.assert_allclose(O.bin_iou(gt, pred), float(d["met_bin_iou"]), rtol=1e-12)
    assert O.mask_iou(np.zeros((4, 4)), np.zeros((4, 4)), 3) == float(d["met_empty_iou"]) == 0


# ------------------------------------------------------------------------------------- GV14: BagGAN generator
def test_baggan_generator_golden_gv14():
    from oracle import baggan_ref as B
    d = load_golden("gv14_baggan")
    sd = {k[3:]: torch.from_numpy(v) for k, v in d.items() if k.startswith("sd_")}
    z, w, ml = (torch.from_numpy(d[k]) for k in ("z", "w", "mean_latent"))
    with torch.no_grad():
        np.testing.assert_allclose(B.mapping(z, sd).numpy(), d["w"], rtol=1e-5, atol=1e-6)
        img, taps, lat = B.generator_forward(w, sd, truncation=0.7, truncation_latent=ml, input_is_latent=True)
    np.testing.assert_allclose(img.numpy(), d["img"], rtol=1e-4, atol=1e-5)
    assert len(taps) == 7
    for i, t in enumerate(taps):
        np.testing.assert_allclose(t.numpy(), d[f"tap_{i}"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(lat.numpy(), d["latent"], rtol=1e-6)
    with torch.no_grad():
        imgz, tapsz, _ = B.generator_forward(z, sd)
    np.testing.assert_allclose(imgz.numpy(), d["img_z"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(tapsz[-1].numpy(), d["tap_z_last"], rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------------------------------------ GV15
def _gv15():
    return load_golden("gv15_kmeans")


def _same_partition(a, b):
    return len(set(zip(a.tolist(), b.tolist()))) == max(len(set(a.tolist())), len(set(b.tolist())))


@pytest.mark.parametrize("tag,kw", [("explicit", {}), ("maxiter2", dict(max_iter=2)), ("far", {})])
def test_gv15_lloyd_matches_sklearn(tag, kw):
    """oracle Lloyd (centring, tolerance, E/M step, empty-cluster relocation, final E step) vs sklearn 1.7.2"""
    from oracle import kmeans_ref as K
    g = _gv15()
    init = g["A_init_far"] if tag == "far" else g["A_init"]
    m = K.kmeans_fit(g["A_x"], init.shape[0], n_init=1, init=init, **kw)
    assert m["n_iter_"] == int(g[f"A_{tag}_n_iter"])
    np.testing.assert_array_equal(m["labels_"], g[f"A_{tag}_labels"])
    np.testing.assert_allclose(m["cluster_centers_"], g[f"A_{tag}_centers"], rtol=0, atol=2e-5)
    assert abs(m["inertia_"] - float(g[f"A_{tag}_inertia"])) <= 1e-5 * float(g[f"A_{tag}_inertia"])


@pytest.mark.parametrize("i", [0, 1])
def test_gv15_degenerate_duplicates_match_sklearn(i):
    """3 distinct points, 5 clusters: no relocation, and sklearn's in-order _average_centers quirk for the empties"""
    from oracle import kmeans_ref as K
    g = _gv15()
    m = K.kmeans_fit(g["A_dup_x"], 5, n_init=1, init=g[f"A_dup{i}_init"])
    np.testing.assert_array_equal(m["labels_"], g[f"A_dup{i}_labels"])
    np.testing.assert_allclose(m["cluster_centers_"], g[f"A_dup{i}_centers"], rtol=0, atol=2e-5)
    assert m["n_iter_"] == int(g[f"A_dup{i}_n_iter"])


def test_gv15_kmeans_plusplus_stream_matches_sklearn():
    """seeded greedy k-means++: the same RandomState reproduces sklearn's chosen indices and final clustering"""
    from oracle import kmeans_ref as K
    g = _gv15()
    x = g["A_x"]
    xc = x - x.mean(axis=0)
    _, ids = K.kmeans_plusplus(xc, 6, K._row_sumsq(xc), np.random.RandomState(5))
    np.testing.assert_array_equal(ids, g["A_pp_indices"])
    for tag, k, n_init, seed in (("pp1", 5, 1, 3), ("pp3", 7, 3, 11)):
        m = K.kmeans_fit(x, k, n_init=n_init, random_state=seed)
        np.testing.assert_array_equal(m["labels_"], g[f"A_{tag}_labels"])
        np.testing.assert_allclose(m["cluster_centers_"], g[f"A_{tag}_centers"], rtol=0, atol=2e-5)
        assert m["n_iter_"] == int(g[f"A_{tag}_n_iter"])
    np.testing.assert_array_equal(K.kmeans_predict(_blobs16(), g["A_explicit_centers"]), g["A_pred_labels"])


def _blobs16():
    rng = np.random.RandomState(16)
    centres = rng.randn(5, 24).astype(np.float32) * 2.0 + 3.0
    return np.ascontiguousarray(centres[rng.randint(0, 5, 700)] + rng.randn(700, 24).astype(np.float32), np.float32)


def test_gv15_reference_hfc_preprocessor():
    """the reference's HFCPreprocessor.train_hfc_model / predict_hfc_vectors + FlatKMeansHFC on its mini generator"""
    import torch
    from oracle import kmeans_ref as K
    g = _gv15()
    sd = golden_sd(g)
    w = torch.from_numpy(g["B_w"])
    clusters, nl = g["B_clusters"].tolist(), len(g["B_clusters"])
    torch.manual_seed(int(g["B_seed_train"]))
    feats, new_list, _, _ = K.train_features(sd, w[0:1], 16, num_latents_for_mean=64, truncation=0.7, n_layers=nl,
                                             n_samples=int(g["B_n_samples"]), perturb_std=g["B_perturb_std"].tolist())
    for n in range(nl):
        np.testing.assert_allclose(feats[n], g[f"B_feat_{n}"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(np.stack([t.numpy() for t in new_list]), g["B_new_latents"], rtol=1e-5, atol=1e-6)
    models = K.flat_fit([g[f"B_feat_{n}"] for n in range(nl)], clusters, random_state=0, n_init=2, verbose=0)
    for n, m in enumerate(models):
        assert _same_partition(m["labels_"], g[f"B_labels_{n}"])
        np.testing.assert_array_equal(m["labels_"], g[f"B_labels_{n}"])
        np.testing.assert_allclose(m["cluster_centers_"], g[f"B_centers_{n}"], rtol=0, atol=1e-5)
        assert abs(m["inertia_"] - float(g[f"B_inertia_{n}"])) <= 1e-5 * float(g[f"B_inertia_{n}"])
        # `self.stds = self.means = [None] * n` (hfc_kmeans_clustering.py:54) aliases the two lists: both hold the std
        np.testing.assert_allclose(m["std"], g[f"B_std_{n}"], rtol=1e-5, atol=1e-6)
        np.testing.assert_array_equal(g[f"B_mean_{n}"], g[f"B_std_{n}"])
    torch.manual_seed(int(g["B_seed_predict"]))
    pf, _ = K.predict_features(sd, w[1:2], 16, num_latents_for_mean=64, truncation=0.7, n_layers=nl)
    maps, labels = K.flat_predict(pf, [g[f"B_centers_{n}"] for n in range(nl)], int(g["B_out_size"]))
    for n in range(nl):
        np.testing.assert_array_equal(labels[n], g[f"B_pred_labels_{n}"])
    np.testing.assert_array_equal(maps * 2 - 1, g["B_pred_maps"])
    assert sorted(g["B_saved_files"].tolist()) == ["clusterer_layer_0.sav", "clusterer_layer_1.sav", "clusterer_layer_2.sav",
                                                   "model_stats.npz"]


def test_gv15_reference_hierarchical_kmeans():
    """hfc_algo='hfc_kmeans_hier' (HierarchicalKMeansHFC, hfc_kmeans_clustering.py:212-390) as the reference ran it on
    its mini generator: the propagated initial centres, the fits they lead to and predict; and what the reference itself
    does with hier_encode=True (it raises: the build refuses that flag with the same reason)"""
    from oracle import kmeans_ref as K
    g = _gv15()
    clusters, nl = g["C_clusters"].tolist(), len(g["C_clusters"])
    assert clusters == [3, 6, 12]
    feats = [g[f"C_feat_{n}"] for n in range(nl)]
    models = K.hier_fit(feats, clusters, random_state=0, n_init=2, verbose=0)
    for n, m in enumerate(models):
        if n:
            np.testing.assert_allclose(m["init"], g[f"C_init_{n}"], rtol=1e-6, atol=1e-7)
            assert np.array_equal(m["init"][0::2], m["init"][1::2]) and np.ptp(m["init"], axis=1).max() == 0
        np.testing.assert_array_equal(m["labels_"], g[f"C_labels_{n}"])
        np.testing.assert_allclose(m["cluster_centers_"], g[f"C_centers_{n}"], rtol=0, atol=1e-5)
        assert abs(m["inertia_"] - float(g[f"C_inertia_{n}"])) <= 1e-5 * float(g[f"C_inertia_{n}"])
        assert m["n_iter_"] == int(g[f"C_n_iter_{n}"])
    assert sorted(g["C_saved_files"].tolist()) == [f"clusterer_layer_{n}.sav" for n in range(nl)]    # no model_stats.npz
    assert str(g["D_hier_encode_outcome"]).startswith("ValueError: X has 12 features, but KMeans is expecting 16")


# ------------------------------------------------------------------------------------- GV16: native-op signatures, metrics
def test_gv16_upfirdn2d_general_and_fused_bias_act():
    d = load_golden("gv16_native_ops")
    for i, case in enumerate(d["upfirdn_cases"]):
        major, ih, iw, minor, kh, kw, *rest = [int(v) for v in case]
        y = G.upfirdn2d_general(T(d[f"up{i}_x"]), T(d[f"up{i}_k"]), *rest)
        assert tuple(y.shape) == d[f"up{i}_y"].shape, i
        close(y, d[f"up{i}_y"], rtol=1e-5, atol=1e-5)
    close(G.fused_bias_act(T(d["fl_x"]), T(d["fl_b"]), None, 3, 0, 0.2, 2 ** 0.5), d["fl_y"])
    close(G.fused_bias_act(T(d["fl_gy"]), None, T(d["fl_y"]), 3, 1, 0.2, 2 ** 0.5), d["fl_gx"])
    close(G.fused_bias_act(T(d["fl2_x"]), T(d["fl2_b"]), None, 3, 0, 0.2, 1.5), d["fl2_y"])
    assert float(G.fused_bias_act(T(d["fl_x"]), None, T(d["fl_y"]), 3, 2, 0.2, 1.0).abs().max()) == 0.0


def test_gv16_remaining_test_metrics():
    from oracle import oneshot_ref as O
    d = load_golden("gv16_native_ops")
    classes = ["background", "a", "b", "c"]
    gts, preds = d["met_gt"], d["met_pred"]
    mi = np.array([[O.mask_iou(g, p, i) for i in range(4)] for g, p in zip(gts, preds)])
    bb = np.array([[O.bb_iou(g, p, i) for i in range(4)] for g, p in zip(gts, preds)])
    np.testing.assert_allclose(mi, d["met_mask_iou"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(bb, d["met_bb_iou"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(O.dice_of_iou(mi), d["met_mask_dice"], atol=1e-15)
    np.testing.assert_allclose(O.dice_of_iou(bb), d["met_bb_dice"], atol=1e-15)
    pdv = O.pd_at_iou_threshold(mi, classes)
    np.testing.assert_allclose([pdv[c] for c in classes], d["met_pd"])
    curve = O.iou_vs_pd_curve(mi, classes)
    np.testing.assert_allclose(np.array([curve[c] for c in classes + ["Mean"]]), d["met_curve"], atol=1e-15)


def test_fixtures_regenerate_from_the_reference():
    """tools/verify_golden.py: every capture script, run against /root/reference into a scratch directory, reproduces the
    tracked tests/golden/*.npz array for array (only GV10's wall-clock log string may differ) and leaves the tracked files
    untouched.  The reference exists in the build container only; elsewhere the fixtures are the pin."""
    import os
    import subprocess
    import sys
    if not os.path.isdir("/root/reference/hfc_with_swav"):
        pytest.skip("the reference tree is not on this machine")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    golden = os.path.join(root, "tests", "golden")
    before = {n: os.path.getmtime(os.path.join(golden, n)) for n in os.listdir(golden)}
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "verify_golden.py")], capture_output=True, text=True, timeout=1900)   # >= the sum of verify_golden's per-script budgets (6 x 300 s)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "0 unexpected difference(s)" in p.stdout
    assert before == {n: os.path.getmtime(os.path.join(golden, n)) for n in os.listdir(golden)}
