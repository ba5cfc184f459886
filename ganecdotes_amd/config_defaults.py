"""Builders for the plugin-surface config modules (configs/segmentors/*.py, configs/models/*.py).

The reference's config files are plain Python modules whose ATTRIBUTE NAMES are the plugin
surface (configs/segmentors/hfc_with_swav_ffhq_config.py:1-78, configs/models/ffhq_256.py:3-35).
Each config module here calls one builder and publishes the result with globals().update(),
so the names and values match while the per-dataset differences stay one line each.
"""
import os

ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STYLEGAN2_256_HF_DIM = [512, 1024, 1024, 1024, 1024, 512, 256]   # per-resolution tap widths, 4^2..256^2
BAGGAN_256_HF_DIM = [512, 1024, 512, 256, 128, 64, 32]


def swav_segmentor(nprototypes, eps, seg_size, *, train_hfc=False, num_epochs=100, source_pdf="uniform",
                   layer_hf_dim=None, hlen=None, nclasses=512, n_hfc_layers=6):
    """Attributes of a `hfc_with_swav*` segmentor config."""
    dims = list(layer_hf_dim or STYLEGAN2_256_HF_DIM)
    hlen = sum(dims) if hlen is None else hlen
    prep = dict(
        perturb_args=dict(truncation=0.7, n_layers=n_hfc_layers, n_samples=1, layer_no=None,
                          perturb_std=[1.0] * n_hfc_layers),
        swav_args=dict(num_epochs=num_epochs, num_samples=1, num_patches=5, sampling_method="random",
                       patch_size=20000, hf_interp="nearest", warmup_epochs=100, start_warmup=0.01,
                       use_scheduler=False, base_lr=0.01, final_lr=0.0001, trust_coeff=0.01,
                       freeze_prototype_niters=313, train_args=dict(lr=0.01, momentum=0.9),
                       projn_nw="linear", temperature=0.01, nprototypes=nprototypes, nclasses=nclasses,
                       hlen=hlen, add_local_loss=False, plot_test_images=False, epoch_print_freq=5,
                       max_masks=4),
        sinkhorn_args=dict(source_pdf=source_pdf, niters=10, eps=eps),
        train=train_hfc,
        layer_hf_dim=dims)
    return dict(n_layers=13, n_hfc_layers=n_hfc_layers, train_hfc=train_hfc, layer_hf_dim=dims, hlen=hlen,
                nclasses=nclasses, hfc_prep_args=prep, seg_args=dict(size=seg_size, in_ch=nclasses))


def simclr_segmentor(*, nclasses=512, n_hfc_layers=6):
    """Attributes of the `hfc_with_simclr` config (configs/segmentors/hfc_with_simclr_config.py)."""
    dims = list(STYLEGAN2_256_HF_DIM)
    prep = dict(
        perturb_args=dict(truncation=0.7, n_layers=n_hfc_layers, n_samples=1, layer_no=None,
                          perturb_std=[1.0] * n_hfc_layers),
        simclr_args=dict(num_iters=100, batch_size=20, patch_size=20000, hf_interp="nearest", trust_coeff=0.01,
                         train_args=dict(lr=0.01, momentum=0.9), temperature=1.0, nclasses=nclasses,
                         hlen=sum(dims), epoch_print_freq=5, max_masks=4),
        train=True,
        layer_hf_dim=dims)
    return dict(n_layers=13, n_hfc_layers=n_hfc_layers, train_hfc=True, layer_hf_dim=dims, hlen=sum(dims),
                nclasses=nclasses, hfc_prep_args=prep, seg_args=dict(size="XS", in_ch=nclasses))


def kmeans_segmentor(*, n_hfc_layers=5, clusters_per_layer=(4, 8, 16, 32, 64), train_hfc=True):
    """Attributes of the `hfc_kmeans` config (configs/segmentors/hfc_kmeans_config.py:1-75): flat k-means per
    generator layer, 4 perturbed samples per layer, label maps at 256, the 5-layer 'S' head on sum(K) channels."""
    clusters = list(clusters_per_layer)
    prep = dict(
        perturb_args=dict(truncation=0.7, n_layers=n_hfc_layers, n_samples=4, perturb_std=[1.0] * n_hfc_layers),
        hfc_algo='hfc_kmeans',
        hfc_args=dict(kmeans_args=dict(verbose=0),
                      base_args=dict(out_dir=None, n_layers=n_hfc_layers, clusters_per_layer=clusters, out_size=256,
                                     presaved=not train_hfc)),
        hier_encode=False, hle_samples=100, train=train_hfc)
    return dict(n_layers=13, n_hfc_layers=n_hfc_layers, clusters_per_layer=clusters, train_hfc=train_hfc,
                hfc_prep_args=prep, seg_args=dict(size='S', in_ch=sum(clusters)))


def stylegan_model(subdir, ckpt, classes, one_shot_ind=0, image_size=256, latents_dir=None):
    """Attributes of a standard (non-BagGAN) model config."""
    base = ROOT_DIR + "/checkpoints/standard/"
    ldir = base + (latents_dir or subdir)
    return dict(model_path=base + subdir + "/" + ckpt, num_latents_for_mean=4096, truncation=0.7,
                image_size=image_size, latent_dim=512, gen_args=dict(size=image_size, style_dim=512, n_mlp=8),
                is_baggan=False, sample_latents=ldir + "/latents.pt", sample_labels=ldir + "/labels.pt",
                sample_images=ldir + "/images/", one_shot_ind=one_shot_ind, classes=list(classes))


def baggan_model(subdir, classes, one_shot_ind=0, truncation=0.95):
    """pidray-* models: the BagGAN generator (ganecdotes_amd/baggan.py).  `model_path` is where the
    reference's BagGANHQ finds the generator (models/baggan/config/config_pidray_unlabeled.py:12-28, load_epoch
    740: `G_net_e_740_i_0.pth`); `gen_args` are its generator_params (:80-86)."""
    base = ROOT_DIR + "/checkpoints/baggan/" + subdir
    return dict(config_path=ROOT_DIR + "/models/baggan/config/config_pidray_unlabeled.py",
                model_path=ROOT_DIR + "/checkpoints/baggan/pidray_baggan_presaved/models/expt_v4.0.1/G_net_e_740_i_0.pth",
                num_latents_for_mean=4096, truncation=truncation, image_size=256, latent_dim=512, is_baggan=True,
                gen_args=dict(latent_dims=(512, 512), img_resolution=256, mlp_layers=8, mlp_lr=0.01, img_chls=3,
                              fir_filter=[1, 3, 3, 1], res2chlmap=None),
                sample_latents=base + "/latents.pt", sample_labels=base + "/labels.pt",
                sample_images=base + "/images/", one_shot_ind=one_shot_ind, classes=list(classes))
