#!/usr/bin/env python3
"""Capture golden vectors from the reference's own functions.

Runs ONLY in the build container, where the upstream reference is mounted
read-only at /root/reference.  It imports the reference at run time (nothing is
copied), feeds it small seeded inputs and writes plain arrays to
tests/golden/*.npz.  Only arrays are committed: no reference source, bytecode or
derived file ships.

Third-party packages that the reference imports at module level but that are not
installed here (torchvision, astropy, skimage, imageio, tensorboard, apex) get
inert placeholder modules in sys.modules.  None of them contributes arithmetic to
the captured functions:
  * torchvision.transforms.Compose -> identity callable (rotation/flip disabled;
    S1b is restated from torchvision's public definition, "parity unpinned")
  * apex LARC -> pass-through around the wrapped optimizer (S8 LARC is restated
    from apex's public definition, "parity unpinned"); GV10 therefore pins the
    pretrain loop with plain momentum-SGD.

Usage:  python tools/capture_golden.py [--out tests/golden]
"""
import argparse
import functools
import os
import sys
import types

import numpy as np

REF = "/root/reference"


# --------------------------------------------------------------------------
# inert placeholders for absent third-party modules
# --------------------------------------------------------------------------
def _install_placeholders():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Identity:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x

    class _Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            return x  # rotation / flip disabled in captures

    tvt = mod("torchvision.transforms", Compose=_Compose, RandomRotation=_Identity,
              RandomHorizontalFlip=_Identity, RandomVerticalFlip=_Identity,
              ToPILImage=_Identity, ToTensor=_Identity)
    tvu = mod("torchvision.utils", make_grid=lambda *a, **k: None)
    mod("torchvision", transforms=tvt, utils=tvu)

    fits = mod("astropy.io.fits")
    aio = mod("astropy.io", fits=fits)
    mod("astropy", io=aio)

    mod("skimage.measure", regionprops=lambda *a, **k: None)
    mod("skimage.transform", rescale=lambda *a, **k: None)
    mod("skimage")
    mod("imageio")

    class _Writer:
        def __init__(self, *a, **k):
            self.scalars = []

        def add_scalar(self, tag, val, step):
            self.scalars.append((tag, float(val), int(step)))

        def add_image(self, *a, **k):
            pass

    mod("torch.utils.tensorboard", SummaryWriter=_Writer)

    class _LARC:
        """pass-through: delegates to the wrapped optimizer (no LARC arithmetic)."""

        def __init__(self, optimizer, trust_coefficient=0.02, clip=True, eps=1e-8):
            self.optim = optimizer

        @property
        def param_groups(self):
            return self.optim.param_groups

        def zero_grad(self):
            self.optim.zero_grad()

        def step(self):
            self.optim.step()

    larc = mod("apex.parallel.LARC", LARC=_LARC)
    par = mod("apex.parallel", LARC=larc)
    mod("apex", parallel=par)
    return _Writer


def _np(t):
    return t.detach().cpu().numpy().copy()


def out_dir_arg(description=None):
    """`--out DIR` of every capture script (default: tests/golden): capturing into a scratch directory leaves the
    tracked fixtures untouched (tools/verify_golden.py compares the two)"""
    ap = argparse.ArgumentParser(description=description)
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    return args.out


class _Logger:
    def __init__(self):
        self.lines = []

    def info(self, msg):
        self.lines.append(str(msg))


def build_mini_generator(M, seed, size=32, style_dim=16, n_mlp=2, channels=None):
    """the reference's own Generator class (M = models.stylegan2.model) with small channel counts"""
    import torch
    channels = channels or {4: 16, 8: 16, 16: 8, 32: 8, 64: 8}
    torch.manual_seed(seed)
    g = M.Generator(size, style_dim, n_mlp)        # builds the full-width net ...
    g.channels = channels
    g.input = M.ConstantInput(channels[4])         # ... whose blocks are replaced
    g.conv1 = M.StyledConv(channels[4], channels[4], 3, style_dim)
    g.to_rgb1 = M.ToRGB(channels[4], style_dim, upsample=False)
    g.convs = torch.nn.ModuleList()
    g.to_rgbs = torch.nn.ModuleList()
    cin = channels[4]
    for i in range(3, g.log_size + 1):
        cout = channels[2 ** i]
        g.convs.append(M.StyledConv(cin, cout, 3, style_dim, upsample=True))
        g.convs.append(M.StyledConv(cout, cout, 3, style_dim))
        g.to_rgbs.append(M.ToRGB(cout, style_dim))
        cin = cout
    # make every path numerically active
    for m in g.modules():
        if isinstance(m, M.NoiseInjection):
            m.weight.data.fill_(0.1)
        if isinstance(m, M.FusedLeakyReLU):
            m.bias.data.normal_(0, 0.2)
        if isinstance(m, M.ToRGB):
            m.bias.data.normal_(0, 0.2)
    for m in g.style:
        if isinstance(m, M.EqualLinear):
            m.bias.data.normal_(0, 10.0)
    g.eval()
    return g


class MiniCfg:
    num_latents_for_mean = 64
    truncation = 0.7
    latent_dim = 16

def build_swav(SC, Writer, gen, out_dir, hlen, nclasses, nproto, patch, source_pdf="uniform", eps=0.05,
               niters=10, num_epochs=3, num_patches=2, layer_hf_dim=None, projn_nw="linear"):
    """the reference's SwAVClustering (SC = hfc_with_swav.swav_clustering) on a mini generator, CPU"""
    perturb_args = dict(truncation=0.7, n_layers=3, n_samples=1, layer_no=None,
                        perturb_std=[1.0] * 3)
    swav_args = dict(num_epochs=num_epochs, num_samples=1, num_patches=num_patches,
                     sampling_method="random", patch_size=patch, hf_interp="nearest",
                     warmup_epochs=num_epochs, start_warmup=0.01, use_scheduler=False,
                     base_lr=0.01, final_lr=0.0001, trust_coeff=0.01,
                     train_args=dict(lr=0.01, momentum=0.9), projn_nw=projn_nw,
                     temperature=0.1, nprototypes=nproto, nclasses=nclasses, hlen=hlen,
                     add_local_loss=False, plot_test_images=False, epoch_print_freq=1,
                     max_masks=4)
    sk = dict(source_pdf=source_pdf, niters=niters, eps=eps)
    return SC.SwAVClustering(model=gen, model_config=MiniCfg, perturb_args=perturb_args,
                             swav_args=swav_args, sinkhorn_args=sk, logger=_Logger(),
                             train=True, out_dir=out_dir, device="cpu", tb=Writer(),
                             layer_hf_dim=layer_hf_dim)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "tests", "golden"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)

    Writer = _install_placeholders()
    sys.path.insert(0, REF)
    import matplotlib
    matplotlib.use("agg")
    import torch
    torch.set_num_threads(4)
    import torch.nn.functional as F
    from models.stylegan2 import model as M
    from hfc_with_swav import swav_clustering as SC
    from lib.oneshot import image_augmentor as IA

    def save(name, **arrs):
        path = os.path.join(args.out, name + ".npz")
        np.savez_compressed(path, **arrs)
        print(f"wrote {path}: {sorted(arrs)} ({os.path.getsize(path)} B)")

    # ---------------------------------------------------------------- GV1
    torch.manual_seed(101)
    k = M.make_kernel([1, 3, 3, 1])
    x1 = torch.randn(2, 3, 9, 9)
    y1 = M.upfirdn2d(x1, k * 4, pad=(1, 1))                    # Blur after up-conv
    x3 = torch.randn(2, 3, 8, 8)
    y3 = M.upfirdn2d(x3, k * 4, up=2, down=1, pad=(2, 1))      # Upsample (RGB skip)
    save("gv1_upfirdn2d", kernel=_np(k), x_mode1=_np(x1), y_mode1=_np(y1),
         x_mode3=_np(x3), y_mode3=_np(y3))

    # ---------------------------------------------------------------- GV2
    torch.manual_seed(102)
    out = {}
    for tag, kw in (("plain", dict(kernel_size=3)),
                    ("up", dict(kernel_size=3, upsample=True)),
                    ("rgb", dict(kernel_size=1, demodulate=False))):
        cout = 3 if tag == "rgb" else 12
        mc = M.ModulatedConv2d(8, cout, style_dim=16, **kw)
        mc.modulation.bias.data.normal_(1.0, 0.3)
        x = torch.randn(2, 8, 8, 8)
        s = torch.randn(2, 16)
        with torch.no_grad():
            y = mc(x, s)
        out.update({f"{tag}_weight": _np(mc.weight), f"{tag}_mod_weight": _np(mc.modulation.weight),
                    f"{tag}_mod_bias": _np(mc.modulation.bias), f"{tag}_x": _np(x),
                    f"{tag}_style": _np(s), f"{tag}_y": _np(y)})
    save("gv2_modconv", **out)

    # ---------------------------------------------------------------- GV3
    torch.manual_seed(103)
    out = {}
    for tag, up in (("plain", False), ("up", True)):
        sc = M.StyledConv(8, 12, 3, 16, upsample=up)
        sc.noise.weight.data.fill_(0.37)
        sc.activate.bias.data.normal_(0, 0.5)
        sc.conv.modulation.bias.data.normal_(1.0, 0.3)
        x = torch.randn(2, 8, 8, 8)
        s = torch.randn(2, 16)
        hw = 16 if up else 8
        nz = torch.randn(1, 1, hw, hw)
        with torch.no_grad():
            y = sc(x, s, noise=nz)
        sd = {f"{tag}_sd_{k_}": _np(v) for k_, v in sc.state_dict().items()}
        out.update(sd)
        out.update({f"{tag}_x": _np(x), f"{tag}_style": _np(s), f"{tag}_noise": _np(nz),
                    f"{tag}_y": _np(y)})
    rgb = M.ToRGB(12, 16)
    rgb.bias.data.normal_(0, 0.5)
    x = torch.randn(2, 12, 16, 16)
    s = torch.randn(2, 16)
    skip = torch.randn(2, 3, 8, 8)
    with torch.no_grad():
        y_skip = rgb(x, s, skip)
        y_noskip = rgb(x, s, None)
    out.update({f"rgb_sd_{k_}": _np(v) for k_, v in rgb.state_dict().items()})
    out.update(dict(rgb_x=_np(x), rgb_style=_np(s), rgb_skip=_np(skip), rgb_y_skip=_np(y_skip),
                    rgb_y_noskip=_np(y_noskip)))
    el = M.EqualLinear(16, 16, lr_mul=0.01, activation="fused_lrelu")
    el.bias.data.normal_(0, 30.0)
    el2 = M.EqualLinear(16, 8, bias_init=1)
    xin = torch.randn(5, 16)
    with torch.no_grad():
        out.update(dict(el_weight=_np(el.weight), el_bias=_np(el.bias), el_x=_np(xin),
                        el_y=_np(el(xin)), el2_weight=_np(el2.weight), el2_bias=_np(el2.bias),
                        el2_y=_np(el2(xin)), pn_y=_np(M.PixelNorm()(xin))))
    save("gv3_layers", **out)

    # ---------------------------------------------------------------- GV4
    # mini generator built from the reference's own classes with small channels,
    # run through the reference's own Generator.forward.
    mini_generator = functools.partial(build_mini_generator, M)

    g = mini_generator(104)
    torch.manual_seed(1040)
    with torch.no_grad():
        mean_lat = g.mean_latent(64)
        z = torch.randn(2, 16)
        w = g.style(z)
        img1, taps1 = g([w], truncation=0.7, truncation_latent=mean_lat, input_is_latent=True,
                        randomize_noise=False)
        _, lat = g([w], return_latents=True, truncation=0.7, truncation_latent=mean_lat,
                   input_is_latent=True)
        img2, taps2 = g([lat], truncation=0.7, truncation_latent=mean_lat, input_is_latent=True,
                        randomize_noise=False)
        imgz, tapsz = g([z], truncation=1, input_is_latent=False, randomize_noise=False)
    out = {f"sd_{k_}": _np(v) for k_, v in g.state_dict().items()}
    out.update(dict(mean_latent=_np(mean_lat), z=_np(z), w=_np(w), latent_trunc_once=_np(lat),
                    img_once=_np(img1), img_twice=_np(img2), img_z=_np(imgz)))
    for i, (a, b, c) in enumerate(zip(taps1, taps2, tapsz)):
        out[f"tap_once_{i}"] = _np(a)
        out[f"tap_twice_{i}"] = _np(b)
        out[f"tap_z_{i}"] = _np(c)
    save("gv4_mini_generator", **out)

    # ---------------------------------------------------------------- helper: SwAV object
    _Cfg = MiniCfg
    make_swav = functools.partial(build_swav, SC, Writer)

    import tempfile
    tmp = tempfile.mkdtemp(prefix="gv_")

    # ---------------------------------------------------------------- GV5
    g5 = mini_generator(105)
    torch.manual_seed(1050)
    sw = make_swav(g5, tmp, hlen=37, nclasses=8, nproto=12, patch=30)
    taps = [torch.randn(1, 6, 2, 2), torch.randn(1, 10, 4, 4), torch.randn(1, 12, 8, 8),
            torch.randn(1, 14, 16, 16)]
    stack = sw.create_pixel_feature_vectors(taps)             # hlen=37 cuts the 4th tap
    sw.swav_args["hlen"] = 42
    stack_full = sw.create_pixel_feature_vectors(taps)
    save("gv5_pixel_features", hlen=np.int64(37), stack=_np(stack), stack_full=_np(stack_full),
         **{f"tap_{i}": _np(t) for i, t in enumerate(taps)})

    # ---------------------------------------------------------------- GV6
    torch.manual_seed(106)
    C, D, K, HW = 16, 8, 12, 8
    sw = make_swav(g5, tmp, hlen=C, nclasses=D, nproto=K, patch=30)
    sw.projection = torch.nn.Sequential(torch.nn.Linear(C, D, bias=False))
    sw.prototype = torch.nn.Linear(D, K)
    hfeat = torch.randn(1, C, HW, HW)
    picks = torch.randperm(HW * HW)
    with torch.no_grad():
        sc_train = sw.get_swav_codes_from_hidden_features(hfeat, picks=picks, train=True)
        sc_pred = sw.get_swav_codes_from_hidden_features(hfeat, (1, D, HW, HW), train=False)
    save("gv6_codes", hfeat=_np(hfeat), picks=_np(picks), patch_size=np.int64(30),
         proj_weight=_np(sw.projection[0].weight), proto_weight=_np(sw.prototype.weight),
         proto_bias=_np(sw.prototype.bias), scores_train=_np(sc_train), scores_pred=_np(sc_pred),
         labels_pred=_np(sc_pred.max(1)[1]))

    # ---------------------------------------------------------------- GV7
    torch.manual_seed(107)
    out = {}
    N, K = 64, 12
    scores = F.normalize(torch.randn(N, 8), dim=1) @ F.normalize(torch.randn(K, 8), dim=1).t()
    scores = scores + 0.01 * torch.randn(K)
    img = torch.rand(1, 16, 16) * 3.0 + 0.5
    out["scores"] = _np(scores)
    out["img"] = _np(img)
    for tag, pdf, eps in (("uniform_e05", "uniform", 0.05), ("image_e05", "image", 0.05),
                          ("uniform_e005", "uniform", 0.005), ("image_e003", "image", 0.003)):
        sw = make_swav(g5, tmp, hlen=C, nclasses=D, nproto=K, patch=30, source_pdf=pdf, eps=eps)
        sc_in = scores * (0.3 if eps < 0.01 else 1.0)      # keep exp() finite at small eps
        with torch.no_grad():
            q = sw.sinkhorn_knopp(sc_in.clone(), img)
        out[f"q_{tag}"] = _np(q)
        out[f"scale_{tag}"] = np.float32(0.3 if eps < 0.01 else 1.0)
        out[f"eps_{tag}"] = np.float32(eps)
    # bigger uniform case with more prototypes than a wave is wide
    N2, K2 = 300, 200
    scores2 = 0.25 * (F.normalize(torch.randn(N2, 16), dim=1) @
                      F.normalize(torch.randn(K2, 16), dim=1).t())
    sw = make_swav(g5, tmp, hlen=C, nclasses=D, nproto=K2, patch=30, eps=0.01)
    with torch.no_grad():
        out["scores_big"] = _np(scores2)
        out["q_big_uniform_e01"] = _np(sw.sinkhorn_knopp(scores2.clone(), img))
    # overflow documentation: un-shifted exp() -> non-finite output in the reference
    sw = make_swav(g5, tmp, hlen=C, nclasses=D, nproto=K, patch=30, eps=0.005)
    with torch.no_grad():
        q_over = sw.sinkhorn_knopp(scores.clone() * 2.0, img)
    out["overflow_scale"] = np.float32(2.0)
    out["overflow_all_finite"] = np.bool_(bool(torch.isfinite(q_over).all()))
    out["niters"] = np.int64(10)
    save("gv7_sinkhorn", **out)

    # ---------------------------------------------------------------- GV8
    torch.manual_seed(108)
    C, D, K, N = 16, 8, 12, 40
    sw = make_swav(g5, tmp, hlen=C, nclasses=D, nproto=K, patch=N, eps=0.05)
    sw.projection = torch.nn.Sequential(torch.nn.Linear(C, D, bias=False))
    sw.prototype = torch.nn.Linear(D, K)
    hs = torch.randn(1, C, 8, 8)
    ht = hs + 0.3 * torch.randn(1, C, 8, 8)
    picks = torch.randperm(64)
    T = 0.1
    ss = sw.get_swav_codes_from_hidden_features(hs, picks=picks, train=True)
    st = sw.get_swav_codes_from_hidden_features(ht, picks=picks, train=True)
    with torch.no_grad():
        qs = sw.sinkhorn_knopp(ss, None)
        qt = sw.sinkhorn_knopp(st, None)
    loss = sw.calculate_swapped_prediction_loss(ss / T, st / T, qs, qt)
    loss.backward()
    save("gv8_loss_grads", hs=_np(hs), ht=_np(ht), picks=_np(picks), temperature=np.float32(T),
         eps=np.float32(0.05), proj_weight=_np(sw.projection[0].weight),
         proto_weight=_np(sw.prototype.weight), proto_bias=_np(sw.prototype.bias),
         scores_s=_np(ss), scores_t=_np(st), q_s=_np(qs), q_t=_np(qt), loss=_np(loss),
         g_proj=_np(sw.projection[0].weight.grad), g_proto_w=_np(sw.prototype.weight.grad),
         g_proto_b=_np(sw.prototype.bias.grad))

    # ---------------------------------------------------------------- GV9
    # RNG order of one view under seed 42 (torch) / 42 (numpy)
    g9 = mini_generator(109)
    torch.manual_seed(42)
    np.random.seed(42)
    sw = make_swav(g9, tmp, hlen=56, nclasses=8, nproto=12, patch=30)   # draws mean_latent(64)
    mean_lat9 = sw.mean_latent.clone()
    z9 = torch.randn(1, 16)
    with torch.no_grad():
        w9 = g9.style(z9)
    views = {}
    for v in ("s", "t"):
        hfeat, img, lno = sw.create_hidden_features_from_perturbed_vectors(
            input_latent=w9, layer_no=None)
        views[f"stack_{v}"] = _np(hfeat)
        views[f"img_{v}"] = _np(img)
        views[f"layer_no_{v}"] = np.int64(lno)
    after = torch.randn(4)
    # the same quantities from explicitly replayed draws (documents the order)
    torch.manual_seed(42)
    np.random.seed(42)
    _ml = torch.randn(64, 16)
    _z = torch.randn(1, 16)
    replay = {}
    for v in ("s", "t"):
        noises = [torch.empty(1, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2)).normal_()
                  for i in range(g9.num_layers)]
        lno = np.random.choice(list(range(3)))
        maps = [torch.randn(1, 16) for _ in range(6)]
        replay[f"replay_layer_no_{v}"] = np.int64(lno)
        replay[f"replay_map_noise_{v}"] = _np(torch.stack(maps))
    after_replay = torch.randn(4)
    out = {f"sd_{k_}": _np(v) for k_, v in g9.state_dict().items()}
    out.update(views)
    out.update(replay)
    out.update(dict(mean_latent=_np(mean_lat9), z=_np(z9), w=_np(w9), after=_np(after),
                    after_replay=_np(after_replay)))
    save("gv9_view_rng", **out)

    # ---------------------------------------------------------------- GV10
    # three epochs of the reference's own pretrain() (rotation/flip disabled, LARC
    # pass-through -> plain momentum SGD), seed 42.
    g10 = mini_generator(110)
    torch.manual_seed(42)
    np.random.seed(42)
    sw = make_swav(g10, tmp, hlen=56, nclasses=8, nproto=12, patch=300, eps=0.05,
                   num_epochs=3, num_patches=2)
    sw.pretrain(None, num_test_samples=2)
    losses = [s[1] for s in sw.writer.scalars]
    out = {f"sd_{k_}": _np(v) for k_, v in g10.state_dict().items()}
    out.update(dict(losses=np.asarray(losses, np.float64),
                    proj_weight=_np(sw.projection[0].weight),
                    proto_weight=_np(sw.prototype.weight), proto_bias=_np(sw.prototype.bias),
                    log=np.asarray([l for l in sw.logger.lines if l.startswith(" E:")])))
    save("gv10_pretrain", **out)

    # ---------------------------------------------------------------- GV11
    torch.manual_seed(1110)
    lat = torch.randn(1, 16)
    with torch.no_grad():
        wlat = g10.style(lat)
        preds, labels = sw.predict_swav_codes(wlat)
    save("gv11_predict", w=_np(wlat), mean_latent=_np(sw.mean_latent), preds=_np(preds),
         labels=_np(labels))

    # ---------------------------------------------------------------- GV12
    # two iterations of the reference's own SimCLRClustering.pretrain() (baseline config 5),
    # batch_size=3 pixel pairs, rotation/flip disabled, LARC pass-through, seed 42.  The loss is
    # inline in pretrain() (simclr_clustering.py:224-264), so the trajectory + final weights pin it.
    from baseline.hfc_with_simclr import simclr_clustering as SIM
    g12 = mini_generator(112)
    torch.manual_seed(42)
    np.random.seed(42)
    sim_args = dict(num_iters=2, batch_size=3, patch_size=30, hf_interp="nearest", trust_coeff=0.01,
                    train_args=dict(lr=0.01, momentum=0.9), temperature=1.0, nclasses=8, hlen=56,
                    epoch_print_freq=1, max_masks=4)
    sim = SIM.SimCLRClustering(model=g12, model_config=_Cfg,
                               perturb_args=dict(truncation=0.7, n_layers=3, n_samples=1, layer_no=None,
                                                 perturb_std=[1.0] * 3),
                               simclr_args=sim_args, logger=_Logger(), train=True, out_dir=tmp, device="cpu",
                               tb=Writer(), layer_hf_dim=None)
    sim.pretrain(None)
    out = {f"sd_{k_}": _np(v) for k_, v in g12.state_dict().items()}
    out.update({f"proj_{k_}": _np(v) for k_, v in sim.projection.state_dict().items()})
    out["losses"] = np.asarray([s_[1] for s_ in sim.writer.scalars], np.float64)
    torch.manual_seed(1120)
    wl = g12.style(torch.randn(1, 16))
    preds, labels = sim.predict_simclr_codes(wl)
    out.update(dict(pred_w=_np(wl), mean_latent=_np(sim.mean_latent), preds=_np(preds), labels=_np(labels)))
    out.update({f"proj_after_predict_{k_}": _np(v) for k_, v in sim.projection.state_dict().items()
                if "running" in k_})
    save("gv12_simclr", **out)

    print("done")


if __name__ == "__main__":
    main()
