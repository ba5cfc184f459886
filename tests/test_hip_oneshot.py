"""One-shot fine-tune head + test loop on HIP (SURVEY.md par. 8f rank 1) against the golden vectors
captured from the reference (GV13) and against the CPU oracle.  Runs on a real MI355X only."""
import numpy as np
import pytest
import torch

from conftest import golden_sd, load_golden

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from ganecdotes_amd import ops
    ops.lib()
    return torch.device("cuda")


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a.detach() - b.detach()).abs().max() / (b.detach().abs().max() + 1e-30))


@pytest.mark.parametrize("size", ["XXS", "XS", "Lin"])
def test_head_finetune_golden_gv13(gpu, size):
    """the reference's OneShotSegmentor trained the run_trainer way (Adam + CrossEntropyLoss, 6 epochs):
    first forward, per-epoch losses, final weights and predictions"""
    from ganecdotes_amd.oneshot import OneShotHead
    d = load_golden("gv13_oneshot")
    head = OneShotHead(16, 5, size)
    head.load_state_dict({k[len(size) + 6:]: T(d[k]) for k in d if k.startswith(f"{size}_init_")})
    feats = T(d[f"{size}_feats"]).cuda().permute(0, 2, 3, 1).contiguous()          # NHWC
    labels = T(d[f"{size}_labels"]).cuda()
    assert rel(head.logits(feats), T(d[f"{size}_out0"])) < 1e-5
    losses = head.fit(feats, labels, num_epochs=len(d[f"{size}_losses"]), lr=float(d["lr"]), beta1=float(d["beta1"]),
                      beta2=float(d["beta2"]), step_size=int(d["step_size"]), gamma=float(d["gamma"]))
    np.testing.assert_allclose(losses, d[f"{size}_losses"], rtol=1e-4)
    for k, v in head.state_dict().items():
        assert rel(v, T(d[f"{size}_final_{k}"])) < 1e-3, k
    pred = head.predict(feats).cpu().numpy()
    assert (pred == d[f"{size}_pred"]).mean() > 0.99
    # round trip through the torch module the registry exposes
    m = head.to_module()
    assert rel(m(T(d[f"{size}_feats"])), head.logits(feats)) < 1e-5


def test_head_stacked_taps_equal_nine_gemms(gpu):
    """A 3x3 head layer with few output channels (9 Cout <= 128: the ffhq 'XXS' head, 512 -> 12) runs as ONE GEMM over the
    un-shifted padded rows with the nine taps stacked along N plus gx_head_tap_sum (tap sum + bias + LeakyReLU + border mask);
    the nine accumulating GEMMs it replaces added the same per-tap dot products in the same order: logits bitwise equal, at the
    evaluate shape (256 x 256 x 512, 2 images) and with a dilated, activated first layer ('XS')."""
    from ganecdotes_amd.oneshot import OneShotHead
    torch.manual_seed(3)
    for size, ncls, hw, cin in (("XXS", 12, 256, 512), ("XS", 12, 64, 64)):
        head = OneShotHead(cin, ncls, size)
        feats = torch.randn(2, hw, hw, cin, device="cuda")
        a = head.logits(feats).clone()
        pa = head.predict(feats)
        head.stacked_taps = False
        b = head.logits(feats)
        assert torch.equal(a, b), float((a - b).abs().max())
        assert torch.equal(pa, head.predict(feats))


def test_confusion_and_iou_metrics(gpu):
    from oracle import oneshot_ref as O
    from ganecdotes_amd.oneshot import confusion, iou_from_confusion
    d = load_golden("gv13_oneshot")
    classes = ["background", "a", "b", "c"]
    conf = confusion(T(d["met_gt"]).long().cuda(), T(d["met_pred"]).long().cuda(), 4)
    mi, wi, bi = iou_from_confusion(conf, classes)
    np.testing.assert_allclose([mi[c] for c in classes], d["met_mask_iou"], rtol=1e-12)
    np.testing.assert_allclose(wi, float(d["met_w_iou"]), rtol=1e-12)
    np.testing.assert_allclose(bi, float(d["met_bin_iou"]), rtol=1e-12)
    # big random masks, 12 classes, against the restated reference metrics
    rng = np.random.RandomState(3)
    gt = rng.randint(0, 12, (256, 256))
    pred = np.where(rng.rand(256, 256) < 0.7, gt, rng.randint(0, 12, (256, 256)))
    cls = ["background"] + [f"c{i}" for i in range(1, 12)]
    conf = confusion(T(gt).cuda(), T(pred).cuda(), 12)
    assert conf.sum() == 256 * 256
    mi, wi, bi = iou_from_confusion(conf, cls)
    ref = {c: O.mask_iou(gt, pred, i) for i, c in enumerate(cls)}
    np.testing.assert_allclose([mi[c] for c in cls], [ref[c] for c in cls], rtol=1e-12)
    np.testing.assert_allclose(wi, O.weighted_iou(gt, ref, cls), rtol=1e-12)
    np.testing.assert_allclose(bi, O.bin_iou(gt, pred), rtol=1e-12)


@pytest.mark.parametrize("size,n_class", [("XXS", 12), ("XS", 8)])
def test_head_full_size_vs_oracle(gpu, size, n_class):
    """the ffhq shape: a (1, 512, 256, 256) code map, 'XXS' (one conv 512 -> 12) / 'XS' (512 -> 16 -> 8 -> n),
    3 epochs of the reference's trainer settings against the oracle"""
    from oracle import oneshot_ref as O
    from ganecdotes_amd.oneshot import OneShotHead
    torch.manual_seed(5)
    head = OneShotHead(512, n_class, size)
    sd = {k: v.cpu().clone() for k, v in head.state_dict().items()}
    plan = O.segmentor_plan(size, 512, n_class)
    params = []
    for i in range(len(plan)):
        params += [sd[f"layers.{2 * i}.weight"], sd[f"layers.{2 * i}.bias"]]
    feats = torch.randn(1, 512, 256, 256) * 0.5
    labels = torch.randint(0, n_class, (1, 256, 256))
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref_losses, ref_final = O.finetune(feats, labels, params, plan, epochs=3, lr=1e-3, beta1=0.9, beta2=0.99)
    f = feats.cuda().permute(0, 2, 3, 1).contiguous()
    with torch.no_grad():
        assert rel(head.logits(f), O.segmentor_forward(feats, params, plan)) < 2e-5
    losses = head.fit(f, labels.cuda(), num_epochs=3, lr=1e-3, beta1=0.9, beta2=0.99)
    np.testing.assert_allclose(losses, ref_losses, rtol=1e-4)
    # Adam divides by sqrt(v): where a gradient element is at rounding-noise level its sign, hence a
    # +-lr step, is arbitrary in BOTH implementations, so weights are compared as a population
    got = head.state_dict()
    for i in range(len(plan)):
        for a, b in ((got[f"layers.{2 * i}.weight"], ref_final[2 * i]), (got[f"layers.{2 * i}.bias"], ref_final[2 * i + 1])):
            diff = (a.cpu() - b).abs()
            assert float((diff <= 1e-4).float().mean()) > 0.9, float((diff <= 1e-4).float().mean())
            assert float(diff.max()) <= 2 * 3 * 1e-3 + 1e-6            # never more than a sign flip per step
    with torch.no_grad():
        ref_pred = O.segmentor_forward(feats, ref_final, plan).max(1)[1]
    assert (head.predict(f).cpu() == ref_pred).float().mean() > 0.995


@pytest.mark.parametrize("size,hw", [("S", 96), ("L", 64)])
def test_head_kmeans_plans_vs_oracle(gpu, size, hw):
    """the hfc_kmeans head: 124 label-map channels into the 5-layer 'S' plan (dilations 1, 2, 1, 2, 1;
    configs/segmentors/hfc_kmeans_config.py:74-75) and the 9-layer 'L' plan (dilations up to 8) — forward,
    3 Adam epochs and predictions against the oracle"""
    from oracle import oneshot_ref as O
    from ganecdotes_amd.oneshot import OneShotHead
    torch.manual_seed(6)
    n_class, in_ch = 6, 124
    head = OneShotHead(in_ch, n_class, size)
    sd = {k: v.cpu().clone() for k, v in head.state_dict().items()}
    plan = O.segmentor_plan(size, in_ch, n_class)
    params = []
    for i in range(len(plan)):
        params += [sd[f"layers.{2 * i}.weight"], sd[f"layers.{2 * i}.bias"]]
    lab_maps = torch.randint(0, 8, (1, 1, hw, hw))
    feats = (torch.arange(in_ch).reshape(1, -1, 1, 1) % 8 == lab_maps).float() * 2 - 1        # +-1 one-hot-like maps
    labels = torch.randint(0, n_class, (1, hw, hw))
    ref_losses, ref_final = O.finetune(feats, labels, params, plan, epochs=3, lr=1e-3, beta1=0.9, beta2=0.99)
    f = feats.cuda().permute(0, 2, 3, 1).contiguous()
    with torch.no_grad():
        assert rel(head.logits(f), O.segmentor_forward(feats, params, plan)) < 2e-5
    losses = head.fit(f, labels.cuda(), num_epochs=3, lr=1e-3, beta1=0.9, beta2=0.99)
    np.testing.assert_allclose(losses, ref_losses, rtol=2e-4)
    got = head.state_dict()
    for i in range(len(plan)):
        for a, b in ((got[f"layers.{2 * i}.weight"], ref_final[2 * i]), (got[f"layers.{2 * i}.bias"], ref_final[2 * i + 1])):
            diff = (a.cpu() - b).abs()
            assert float((diff <= 1e-4).float().mean()) > 0.9
            assert float(diff.max()) <= 2 * 3 * 1e-3 + 1e-6
    with torch.no_grad():
        ref_pred = O.segmentor_forward(feats, ref_final, plan).max(1)[1]
    assert (head.predict(f).cpu() == ref_pred).float().mean() > 0.99


def test_one_shot_pipeline_end_to_end_vs_oracle(gpu, tmp_path):
    """pretrained projection -> code map of the labelled latent -> fine-tuned head -> run_tests over
    test latents: predictions and the IoU results of the HIP pipeline against the oracle's
    (synthetic labels: there are no datasets in this environment)"""
    from oracle import oneshot_ref as O
    from oracle import swav_ref as S
    from ganecdotes_amd.generator import Generator
    from ganecdotes_amd.oneshot import OneShotHead, run_tests
    from ganecdotes_amd.swav_clustering import SwAVClustering
    MINI = {4: 16, 8: 16, 16: 8, 32: 8, 64: 8}
    d = load_golden("gv10_pretrain")
    sd = golden_sd(d)
    g = Generator(32, 16, 2, channels=MINI)
    g.load_state_dict(sd, strict=False)
    g = g.eval().to(gpu)

    class Cfg:
        num_latents_for_mean, truncation, latent_dim = 64, 0.7, 16
    swav = dict(num_epochs=1, num_samples=1, num_patches=2, sampling_method="random", patch_size=300, hf_interp="nearest",
                warmup_epochs=1, start_warmup=0.01, use_scheduler=False, base_lr=0.01, final_lr=0.0001, trust_coeff=0.01,
                train_args=dict(lr=0.01, momentum=0.9), projn_nw="linear", temperature=0.1, nprototypes=12, nclasses=8,
                hlen=56, add_local_loss=False, plot_test_images=False, epoch_print_freq=1, max_masks=4)
    sw = SwAVClustering(model=g, model_config=Cfg, out_dir=str(tmp_path), device="cuda",
                        perturb_args=dict(truncation=0.7, n_layers=3, n_samples=1, layer_no=None, perturb_std=[1.0] * 3),
                        swav_args=swav, sinkhorn_args=dict(source_pdf="uniform", niters=10, eps=0.05))
    torch.manual_seed(9)
    ml = S.G.mean_latent(sd, 64, 16)
    sw.mean_latent = ml.cuda()
    sw.build_heads()
    proj_w = sw.projection[0].weight.data.cpu().clone()
    classes = ["background", "a", "b"]
    with torch.no_grad():
        w_all = S.G.mapping(torch.randn(5, 16), sd)
    # synthetic ground truth: a fixed partition of the image plane
    yy, xx = torch.meshgrid(torch.arange(32), torch.arange(32), indexing="ij")
    gt = ((xx > 10).long() + (yy > 20).long()).unsqueeze(0).repeat(5, 1, 1)
    # HIP pipeline
    head = OneShotHead(8, len(classes), "XS")
    sd_head = {k: v.cpu().clone() for k, v in head.state_dict().items()}
    codes, _ = sw.predict_swav_codes(w_all[:1].cuda())
    head.fit(codes.permute(0, 2, 3, 1), gt[:1].cuda(), num_epochs=20, lr=1e-3, beta1=0.9, beta2=0.99)
    pred, res = run_tests(sw, head, w_all[1:].cuda(), gt[1:], classes, batch=2)
    # oracle pipeline
    plan = O.segmentor_plan("XS", 8, len(classes))
    params = []
    for i in range(len(plan)):
        params += [sd_head[f"layers.{2 * i}.weight"], sd_head[f"layers.{2 * i}.bias"]]
    f0, _ = S.predict_swav_codes(sd, w_all[:1], ml, 0.7, proj_w, 56)
    _, final = O.finetune(f0, gt[:1], params, plan, epochs=20, lr=1e-3, beta1=0.9, beta2=0.99)
    ref_pred = []
    with torch.no_grad():
        for i in range(1, 5):
            fi, _ = S.predict_swav_codes(sd, w_all[i:i + 1], ml, 0.7, proj_w, 56)
            ref_pred.append(O.segmentor_forward(fi, final, plan).max(1)[1])
    ref_pred = torch.cat(ref_pred)
    assert (pred == ref_pred).float().mean() > 0.99
    ref = O.run_tests_metrics([m.numpy() for m in gt[1:]], [m.numpy() for m in ref_pred], classes)
    for a, b in zip(res["mask_iou"], ref["mask_iou"]):
        for c in classes:
            assert abs(a[c] - b[c]) < 0.005          # BASELINE.json: mIoU within +-0.5 pt of the reference
    np.testing.assert_allclose(res["w_iou"], ref["w_iou"], atol=0.005)
    assert abs(res["bin_iou"] - ref["bin_iou"]) < 0.005


def test_pipeline_on_reference_format_files(gpu, tmp_path):
    """OneShotPipeline on FILES in the reference's formats instead of the synthetic mode:
    a rosinality checkpoint {'g_ema': state_dict, ...} with an extra key (strict=False,
    src/one_shot_pipeline.py:140-147), `latents.pt` as the (latents, noises) tuple, numpy labels stored in
    [0, 1) (x255 rule, :356-374), one_shot_ind != 0; run_pipeline() then leaves projection.pt / prototypes.pt /
    cluster_map.pt / tests/label_predictions.pt / tests/mask_iou_results.csv / bb_iou_results.csv / results.npz."""
    from oracle import swav_ref as S
    from src.one_shot_pipeline import OneShotPipeline
    d = load_golden("gv10_pretrain")
    sd = golden_sd(d)
    torch.save({"g_ema": {**sd, "not.a.generator.key": torch.zeros(3)}, "g": {}, "latent_avg": torch.zeros(16)},
               tmp_path / "ckpt.pt")
    torch.manual_seed(21)
    with torch.no_grad():
        w = S.G.mapping(torch.randn(5, 16), sd)
    torch.save((w, [torch.zeros(1, 1, 4, 4)]), tmp_path / "latents.pt")
    yy, xx = np.meshgrid(np.arange(32), np.arange(32), indexing="ij")
    ids = ((xx > 12).astype(np.int64) + (yy > 18).astype(np.int64))
    torch.save(np.repeat((ids / 255.0).astype(np.float32)[None], 5, 0), tmp_path / "labels.pt")
    (tmp_path / "model_cfg.py").write_text(
        f"model_path = {str(tmp_path / 'ckpt.pt')!r}\nnum_latents_for_mean = 64\ntruncation = 0.7\nimage_size = 32\n"
        "latent_dim = 16\ngen_args = dict(size=32, style_dim=16, n_mlp=2, channels={4: 16, 8: 16, 16: 8, 32: 8, 64: 8})\n"
        f"is_baggan = False\nsample_latents = {str(tmp_path / 'latents.pt')!r}\n"
        f"sample_labels = {str(tmp_path / 'labels.pt')!r}\none_shot_ind = 1\nclasses = ['background', 'a', 'b']\n")
    (tmp_path / "seg_cfg.py").write_text(
        "from ganecdotes_amd.config_defaults import swav_segmentor\n"
        "globals().update(swav_segmentor(12, 0.05, 'XS', train_hfc=True, num_epochs=2, layer_hf_dim=[16, 32, 8], hlen=56,\n"
        "                                nclasses=8, n_hfc_layers=3))\n"
        "hfc_prep_args['swav_args'].update(patch_size=300, num_patches=2, temperature=0.1, epoch_print_freq=1)\n")
    out = tmp_path / "run"
    pipe = OneShotPipeline(out_dir=str(out), exp_name="file formats", model="ffhq-256", segmentor="hfc_with_swav_ffhq",
                           custom=dict(model=str(tmp_path / "model_cfg.py"), seg=str(tmp_path / "seg_cfg.py")),
                           num_test_samples=4, device="cuda:0")
    got = pipe.model.state_dict()
    for k, v in sd.items():                                    # the g_ema entry was the one loaded
        assert torch.equal(got[k].cpu(), v), k
    pipe.trainer_config.num_epochs = 30
    torch.manual_seed(77)                  # the preprocessor's mean_latent is its first device draw
    pipe.run_pipeline()
    assert pipe.one_shot_ind == 1 and torch.equal(pipe.one_shot_latent.cpu(), w[1])
    assert float(pipe.test_labels.max()) == 2.0               # x255 brought the stored [0, 1) labels back to ids
    for f in ("projection.pt", "prototypes.pt", "cluster_map.pt", "tests/label_predictions.pt",
              "tests/mask_iou_results.csv", "tests/bb_iou_results.csv", "tests/results.npz"):
        assert (out / f).exists(), f
    pred = torch.load(out / "tests" / "label_predictions.pt")
    assert pred.shape == (4, 32, 32) and pred.dtype == torch.int64
    rows = (out / "tests" / "mask_iou_results.csv").read_text().splitlines()
    assert rows[0] == ",background,a,b" and len(rows) == 5
    res = np.load(out / "tests" / "results.npz", allow_pickle=True)
    assert {"mask_iou", "bb_iou", "w_iou", "bin_iou", "mask_dice", "bb_dice", "pd", "iou_pd_curve"} <= set(res.files)
    # the CSV is the table of results['mask_iou']
    np.testing.assert_allclose([float(v) for v in rows[1].split(",")[1:]],
                               [res["mask_iou"][0][c] for c in ("background", "a", "b")])
    proj = torch.load(out / "projection.pt", weights_only=False)
    assert proj[0].weight.shape == (8, 56)
    # evaluate.py's route (evaluate.py:93-95): train_hfc / train off -> the constructor loads projection.pt /
    # prototypes.pt back (swav_clustering.py:82-86) and the same code map, head fit and predictions come out
    code0 = pipe.one_shot_features.clone()
    pipe2 = OneShotPipeline(out_dir=str(out), exp_name="evaluate", model="ffhq-256", segmentor="hfc_with_swav_ffhq",
                            custom=dict(model=str(tmp_path / "model_cfg.py"), seg=str(tmp_path / "seg_cfg.py")),
                            num_test_samples=4, device="cuda:0")
    pipe2.seg_config.train_hfc = False
    pipe2.seg_config.hfc_prep_args['train'] = False
    pipe2.trainer_config.num_epochs = 30
    pipe2.run_pipeline(blocks_to_run=('setup',))
    torch.manual_seed(77)                  # same mean_latent draw as the training run
    pipe2.run_trainer()
    assert torch.equal(pipe2.preprocessor.projection[0].weight.data.cpu(), proj[0].weight.data.cpu())
    assert float((pipe2.one_shot_features - code0).abs().max()) == 0.0      # same heads, same generator: same codes


def test_simclr_predict_one_latent_per_call(gpu, tmp_path):
    """hfc_with_simclr's run_tests route: the train-mode BatchNorm of predict_simclr_codes pools
    its statistics over the rows of a call, so the test loop feeds ONE latent per call like the reference
    (MAX_TEST_BATCH = 1, src/one_shot_pipeline.py:30): run_tests(batch=1) == per-latent predictions, and a
    4-latent call would differ."""
    from ganecdotes_amd.generator import Generator
    from ganecdotes_amd.oneshot import OneShotHead, run_tests
    from ganecdotes_amd.simclr_clustering import SimCLRClustering
    d = load_golden("gv12_simclr")
    g = Generator(32, 16, 2, channels={4: 16, 8: 16, 16: 8, 32: 8, 64: 8})
    g.load_state_dict(golden_sd(d), strict=False)
    g = g.eval().to(gpu)

    class Cfg:
        num_latents_for_mean, truncation, latent_dim = 64, 0.7, 16
    args = dict(num_iters=1, batch_size=3, patch_size=30, hf_interp="nearest", trust_coeff=0.01,
                train_args=dict(lr=0.01, momentum=0.9), temperature=1.0, nclasses=8, hlen=56, epoch_print_freq=1,
                max_masks=4)
    sim = SimCLRClustering(model=g, model_config=Cfg, out_dir=None, device="cuda", simclr_args=args,
                           perturb_args=dict(truncation=0.7, n_layers=3, n_samples=1, layer_no=None, perturb_std=[1.0] * 3))
    torch.manual_seed(3)
    sim.build_heads()
    lat = g.style(torch.randn(4, 16, device="cuda"))
    one = torch.cat([sim.predict_simclr_codes(lat[i:i + 1])[0] for i in range(4)])
    four = sim.predict_simclr_codes(lat)[0]
    assert float((one - four).abs().max()) > 1e-4              # pooled statistics are a different function
    head = OneShotHead(8, 3, "XS")

    class Prep:
        predict_swav_codes = staticmethod(lambda l: sim.predict_simclr_codes(l))
    labels = torch.zeros(4, 32, 32)
    pred1, _ = run_tests(Prep, head, lat, labels, ["background", "a", "b"], batch=1)
    want = torch.cat([head.predict(one[i:i + 1].permute(0, 2, 3, 1)) for i in range(4)]).cpu()
    assert torch.equal(pred1, want)
