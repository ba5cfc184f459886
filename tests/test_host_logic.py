"""Host-side logic that needs no GPU: RNG draw order, the C-ABI library's exported symbols,
config/plugin surface."""
import ctypes
import os
import re

import numpy as np
import torch

from conftest import ROOT, load_golden

T = torch.from_numpy


def test_draws_follow_reference_rng_order():
    from ganecdotes_amd.draws import draw_image
    d = load_golden("gv9_view_rng")
    torch.manual_seed(42)
    np.random.seed(42)
    torch.randn(64, 16)                                  # mean_latent draw of the constructor
    im = draw_image(16, 7, 3, 32 * 32, 0, transforms=False, cpu_reference_rng=True)
    assert torch.equal(im.z, T(d["z"]))
    assert torch.equal(torch.randn(4), T(d["after"]))
    for v, tag in zip(im.views, "st"):
        lno = int(d[f"layer_no_{tag}"])
        assert v.layer_no == lno
        ref = T(d[f"replay_map_noise_{tag}"])[[2 * lno, 2 * lno + 1], 0]
        assert torch.equal(v.map_noise, ref)


def test_draws_with_transforms_and_picks():
    from ganecdotes_amd.draws import draw_image
    torch.manual_seed(0)
    np.random.seed(0)
    im = draw_image(16, 7, 3, 64, 5)
    assert len(im.picks) == 5 and all(sorted(p.tolist()) == list(range(64)) for p in im.picks)
    assert all(-10 <= v.angle <= 10 for v in im.views)


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads and exports every function include/ganecdotes_hip.h declares
    (no compute calls: there is no GPU here)."""
    from ganecdotes_amd import _lib, build
    build.build(verbose=False)
    header = open(os.path.join(ROOT, "include", "ganecdotes_hip.h")).read()
    declared = set(re.findall(r"\b(gx_[a-z0-9_]+)\s*\(", header))
    assert len(declared) > 30
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    L = _lib.load()
    assert b"gfx950" in L.gx_version()
    assert L.gx_sinkhorn_workspace_bytes(8) == (32 + 2048 * 8) * 4
    assert L.gx_gemm_f32_workspace_bytes(512, 5376, 8) == 8 * 512 * 5376 * 4


def test_ctypes_table_matches_the_header_prototypes():
    """Every prototype of include/ganecdotes_hip.h, parsed, against the hand-written ctypes table: same
    return type, same argument count, same argument classes (pointer / int / long / size_t / float) in the
    same order.  (The .hip definitions are checked against the same header by the compiler: gx_common.h
    includes it.)"""
    from ganecdotes_amd import _lib
    import ctypes as C
    header = open(os.path.join(ROOT, "include", "ganecdotes_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", " ", header, flags=re.S)
    protos = re.findall(r"\b(int|size_t|const char\*|void\*|float\*)\s+(gx_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", header)
    assert len(protos) == len(_lib.SIGNATURES), (len(protos), len(_lib.SIGNATURES))

    def ctype_of(decl):
        decl = decl.strip()
        if decl in ("void", ""):
            return None
        if "*" in decl:
            return C.c_void_p
        base = re.sub(r"\b(const|unsigned|signed)\b", " ", decl)
        base = " ".join(base.split()[:-1]) if len(base.split()) > 1 else base.strip()      # drop the name
        return {"int": C.c_int, "long": C.c_long, "long long": C.c_longlong, "size_t": C.c_size_t,
                "float": C.c_float, "double": C.c_double}[base]
    ret_of = {"int": C.c_int, "size_t": C.c_size_t, "const char*": C.c_char_p, "void*": C.c_void_p,
              "float*": C.c_void_p}
    for ret, name, args in protos:
        res, argtypes = _lib.SIGNATURES[name]
        assert res is ret_of[ret], (name, ret, res)
        want = [t for t in (ctype_of(a) for a in args.split(",")) if t is not None]
        assert list(argtypes) == want, (name, argtypes, want)


def test_result_tables_and_files_match_the_reference_format(tmp_path):
    """bb IoU / Dice / PD summaries and the text of tests/mask_iou_results.csv against GV16 (captured from
    lib/metrics/segmentation.py and pandas.DataFrame.to_csv, src/one_shot_pipeline.py:775-791, :884)."""
    from conftest import load_golden
    from ganecdotes_amd import oneshot as H
    d = load_golden("gv16_native_ops")
    classes = ["background", "a", "b", "c"]
    gts, preds = d["met_gt"], d["met_pred"]
    bb = np.array([[H.bb_iou(g, p, i) for i in range(4)] for g, p in zip(gts, preds)])
    np.testing.assert_allclose(bb, d["met_bb_iou"], rtol=0, atol=1e-15)
    mi = d["met_mask_iou"]
    pdv = H.pd_at_iou_threshold(mi, classes)
    np.testing.assert_allclose([pdv[c] for c in classes], d["met_pd"])
    curve = H.iou_vs_pd_curve(mi, classes)
    np.testing.assert_allclose(np.array([curve[c] for c in classes + ["Mean"]]), d["met_curve"], atol=1e-15)
    rows = [{c: mi[i, j] for j, c in enumerate(classes)} for i in range(mi.shape[0])]
    assert H.table_csv(rows, classes) == bytes(d["met_csv"]).decode()
    res = dict(mask_iou=rows, bb_iou=[{c: bb[i, j] for j, c in enumerate(classes)} for i in range(bb.shape[0])],
               w_iou=[0.5] * 6, bin_iou=0.25, pd=pdv, iou_pd_curve=curve)
    H.write_result_files(str(tmp_path), res, classes)
    assert (tmp_path / "mask_iou_results.csv").read_text() == bytes(d["met_csv"]).decode()
    assert (tmp_path / "bb_iou_results.csv").read_text().splitlines()[0] == ",background,a,b,c"
    back = np.load(tmp_path / "results.npz", allow_pickle=True)
    assert set(back.files) == set(res) and float(back["bin_iou"]) == 0.25
    assert back["mask_iou"][2]["b"] == rows[2]["b"]


def test_sample_loader_rules(tmp_path):
    """OneShotPipeline's latents / labels loading rules (src/one_shot_pipeline.py:356-384): the
    (latents, noises) tuple form, numpy labels, the x255 rule for labels stored in [0, 1) and the LSUN-car
    canvas (384 rows centred in 512)."""
    from src.one_shot_pipeline import OneShotPipeline
    lat = torch.randn(3, 14, 8)
    lab01 = (np.random.RandomState(0).randint(0, 5, (3, 16, 16)) / 255.0).astype(np.float32)
    torch.save((lat, [torch.zeros(1)]), tmp_path / "latents.pt")
    torch.save(lab01, tmp_path / "labels.pt")

    class MC:
        sample_latents, sample_labels, one_shot_ind = str(tmp_path / "latents.pt"), str(tmp_path / "labels.pt"), 2
    p = object.__new__(OneShotPipeline)
    p.synthetic, p.device, p.model_config, p.model_str = False, "cpu", MC, "ffhq-256"
    p._load_samples()
    assert torch.equal(p.test_latents, lat) and p.one_shot_ind == 2
    np.testing.assert_allclose(p.test_labels.numpy(), lab01 * 255, rtol=1e-6)
    assert p.test_labels.max() >= 1
    # plain tensors, class ids already; LSUN-car canvas
    torch.save(lat, tmp_path / "latents.pt")
    torch.save(torch.ones(3, 384, 512) * 2, tmp_path / "labels.pt")
    p.model_str = "exp-car-512"
    p._load_samples()
    assert p.test_labels.shape == (3, 512, 512)
    assert float(p.test_labels[:, :64].abs().max()) == 0 and float(p.test_labels[:, 64:448].min()) == 2
    assert float(p.test_labels[:, 448:].abs().max()) == 0


def test_product_path_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ganecdotes_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, re.M):
                    bad.append(f)
    assert not bad, bad


def test_registry_and_config_surface():
    """configs/mapper.py keys and the attribute names of the config modules = plugin surface
    (reference: configs/mapper.py:9-67, hfc_with_swav_ffhq_config.py:1-78, ffhq_256.py:3-35)."""
    from configs import mapper
    from lib.util.util import load_config
    for key in ('hfc_with_swav', 'hfc_with_swav_ffhq', 'hfc_with_swav_cat', 'hfc_with_swav_car',
                'hfc_with_swav_horse', 'hfc_with_swav_pidray', 'hfc_with_simclr'):
        cfg = load_config(mapper.segmentors[key], 'seg_config')
        for attr in ('hfc_prep_args', 'seg_args', 'train_hfc', 'n_layers', 'n_hfc_layers', 'layer_hf_dim', 'hlen',
                     'nclasses'):
            assert hasattr(cfg, attr), (key, attr)
        assert set(cfg.hfc_prep_args) >= {'perturb_args', 'train', 'layer_hf_dim'}
    expect = {'hfc_with_swav_ffhq': (5000, 0.005, 'uniform', 'XXS', 5376), 'hfc_with_swav_cat': (5000, 0.003, 'image', 'XS', 5376),
              'hfc_with_swav_car': (4000, 0.01, 'uniform', 'XS', 5376), 'hfc_with_swav_horse': (5000, 0.003, 'uniform', 'XXS', 5376),
              'hfc_with_swav_pidray': (4000, 0.005, 'uniform', 'XXS', 2528), 'hfc_with_swav': (8000, 0.005, 'uniform', 'XXS', 5376)}
    for key, (k, eps, pdf, size, hlen) in expect.items():
        cfg = load_config(mapper.segmentors[key], 'seg_config')
        sa, sk = cfg.hfc_prep_args['swav_args'], cfg.hfc_prep_args['sinkhorn_args']
        assert (sa['nprototypes'], sk['eps'], sk['source_pdf'], cfg.seg_args['size'], sa['hlen']) == (k, eps, pdf, size, hlen)
        assert sa['patch_size'] == 20000 and sa['num_patches'] == 5 and sk['niters'] == 10 and sa['temperature'] == 0.01
        assert set(sa) >= {'num_epochs', 'num_samples', 'sampling_method', 'hf_interp', 'warmup_epochs', 'start_warmup',
                           'use_scheduler', 'base_lr', 'final_lr', 'trust_coeff', 'train_args', 'projn_nw',
                           'nclasses', 'add_local_loss', 'plot_test_images', 'epoch_print_freq', 'max_masks'}
    for key in ('ffhq-256', 'cat-256', 'car-512', 'horse-256', 'afhq-256', 'pidray-256'):
        m = load_config(mapper.models[key], 'model_config')
        for attr in ('num_latents_for_mean', 'truncation', 'image_size', 'latent_dim', 'gen_args', 'is_baggan',
                     'sample_latents', 'sample_labels', 'one_shot_ind', 'classes'):
            assert hasattr(m, attr), (key, attr)
    assert set(mapper.losses) == {'bce', 'softmax', 'sigmoid', 'tanh', 'logloss', 'cross_entropy'}
    assert set(mapper.lr_scheduler) == {'step', 'plateau', 'cosine'} and 'supervised' in mapper.trainer


def test_cli_method_remap():
    import pretrain, evaluate
    assert pretrain.remap('hfc_with_swav', 'ffhq-256') == 'hfc_with_swav_ffhq'
    assert pretrain.remap('hfc_with_swav', 'cat-256') == 'hfc_with_swav_cat'
    assert pretrain.remap('hfc_with_swav', 'car-512') == 'hfc_with_swav_car'
    assert pretrain.remap('hfc_with_swav', 'horse-256') == 'hfc_with_swav_horse'
    assert pretrain.remap('hfc_with_swav', 'pidray-hammer-256') == 'hfc_with_swav_pidray'
    assert pretrain.remap('hfc_with_swav', 'afhq-256') == 'hfc_with_swav'
    assert evaluate.remap('hfc_with_simclr', 'ffhq-256') == 'hfc_with_simclr'


def test_plugin_exports_and_generator_state_dict_keys():
    import hfc_with_swav.base as base
    from models.stylegan2.model import Generator
    assert base.preprocessor.__name__ == 'SwAVClustering' and base.segmentor.__name__ == 'OneShotSegmentor'
    for m in ('preprocess', 'pretrain', 'predict_swav_codes'):
        assert callable(getattr(base.preprocessor, m))
    g = Generator(256, 512, 8)
    keys = set(g.state_dict())
    assert len(keys) == 135          # SURVEY par. 8b: 135 entries at 256^2
    for k in ('style.1.weight', 'style.8.bias', 'input.input', 'conv1.conv.weight', 'conv1.conv.modulation.weight',
              'conv1.noise.weight', 'conv1.activate.bias', 'to_rgb1.bias', 'convs.0.conv.blur.kernel',
              'convs.11.activate.bias', 'to_rgbs.5.upsample.kernel', 'noises.noise_12'):
        assert k in keys, k
    assert g.state_dict()['conv1.conv.weight'].shape == (1, 512, 512, 3, 3)
    assert g.state_dict()['input.input'].shape == (1, 512, 4, 4)
    seg = base.segmentor(512, 8, 'XXS')
    assert [type(l).__name__ for l in seg.layers] == ['Conv2d']
    assert len(base.segmentor(512, 8, 'XS').layers) == 5


def test_baggan_generator_speaks_the_reference_keys():
    """pidray-* models: state_dict keys of the HIP BagGAN generator == the reference StyleGANGenerator's
    (GV14 fixture), checkpoints round-trip through the key translation, and the pidray configs build it."""
    import numpy as np
    import torch
    from conftest import load_golden
    from models.baggan.models import StyleGANGenerator
    d = load_golden("gv14_baggan")
    ref = {k[3:]: torch.from_numpy(v) for k, v in d.items() if k.startswith("sd_")}
    g = StyleGANGenerator(latent_dims=(16, 16), img_resolution=32, mlp_layers=2, _channels={4: 16, 8: 16, 16: 8, 32: 8})
    assert set(g.state_dict().keys()) == set(ref.keys())
    res = g.load_state_dict(ref)
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in g.state_dict().items():
        assert tuple(v.shape) == tuple(ref[k].shape), k
        np.testing.assert_array_equal(v.numpy(), ref[k].numpy())
    from configs import mapper
    from lib.util.util import load_config
    m = load_config(mapper.models['pidray-256'], 'model_config')
    assert m.is_baggan and m.truncation == 0.9 and m.gen_args['img_resolution'] == 256
    assert load_config(mapper.models['pidray-hammer-256'], 'model_config').truncation == 0.95


def test_host_thread_cap_only_lowers():
    """hostcfg.limit_host_threads: torch's intra-op pool is capped at the CPUs the process may use (affinity, cgroup quota) and
    at GX_HOST_THREADS when set (no other cap: ADVICE round 4); it never raises the count and acts once per process"""
    import torch
    from ganecdotes_amd import hostcfg
    before = torch.get_num_threads()
    n = hostcfg.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    got = hostcfg.limit_host_threads()
    assert got == torch.get_num_threads() and got <= before and got == min(before, n)
    torch.set_num_threads(max(1, got - 1) if got > 1 else 1)
    again = hostcfg.limit_host_threads()                 # once per process: a later call leaves the caller's own setting alone
    assert again == torch.get_num_threads()
    torch.set_num_threads(before)


def test_hf_interp_values_the_path_does_not_implement_are_refused_by_name():
    """`hf_interp` is F.interpolate's mode in the reference (hfc_with_swav/swav_clustering.py:123-124; simclr_clustering.py:102-103).
    Both preprocessors implement 'nearest' (every shipped config) and 'bilinear'.  Anything else is
    refused by name instead of being computed as nearest silently; the check runs before any device
    is touched."""
    import pytest
    from ganecdotes_amd.swav_clustering import SwAVClustering
    from ganecdotes_amd.simclr_clustering import SimCLRClustering
    for mode in ("bicubic", "area", "nearest-exact"):
        with pytest.raises(NotImplementedError, match="hf_interp"):
            SwAVClustering(model=None, model_config=None, perturb_args={}, swav_args=dict(hf_interp=mode), sinkhorn_args={},
                           device='cpu')
    with pytest.raises(NotImplementedError, match="hf_interp"):
        SimCLRClustering(model=None, model_config=None, perturb_args={}, simclr_args=dict(hf_interp='bicubic'), device='cpu')
    # 'nearest' and (hfc_with_swav) 'bilinear' pass the check and reach the next one (no HIP device here / a CPU device was asked for)
    for mode in ("nearest", "bilinear"):
        with pytest.raises(RuntimeError, match="HIP device"):
            SwAVClustering(model=None, model_config=None, perturb_args={}, swav_args=dict(hf_interp=mode), sinkhorn_args={},
                           device='cpu')
