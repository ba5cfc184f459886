"""Slim orchestrator with the reference's contract (src/one_shot_pipeline.py:33-465):
`OneShotPipeline(out_dir, exp_name, model, segmentor, trainer, tester, mode, inputs, custom,
device, num_test_samples)`, `load_model`, `load_segmentor`, `load_trainer`,
`run_pipeline(blocks_to_run)`, `run_trainer`, `run_tests`.  It drives the accelerated path: generator
load, `preprocessor.preprocess` (SwAV pretrain) / `train_hfc_model` (k-means), `predict_swav_codes` /
`predict_hfc_vectors`, the one-shot fine-tune of the
head (`run_trainer`, reference :466-581) and the IoU test loop (`run_tests`, :595-886) — all on HIP
(ganecdotes_amd/oneshot.py).  Plotting, tensorboard images and the labelling GUI are outside this build.

`synthetic=True` (opt-in extra): random-init generator under seed 42, random latents and a synthetic
label partition, because no checkpoints / latents / labels exist offline (reference: checkpoints/ is
empty) — the loop is exercised, the IoU numbers mean nothing.
"""
import os
import time

import torch

from configs import mapper as config_mapper
from lib.util.util import get_logger, load_config
from models.stylegan2.model import Generator
import hfc_with_swav.base as hfc_with_swav


class OneShotPipeline(object):

    def __init__(self, out_dir, exp_name='', model='ffhq-256', segmentor='hfc_kmeans', trainer='supervised',
                 tester='all', mode='offline', inputs='saved', custom=None, device='cuda:0',
                 num_test_samples=None, synthetic=False, dist_group=None):
        self.out_dir = out_dir
        os.makedirs(self.out_dir, exist_ok=True)
        self.start_time = time.strftime("%m%d%Y_%H%M%S", time.localtime())
        self.rank = 0
        if dist_group is not None:
            import torch.distributed as dist
            self.rank = dist.get_rank(dist_group)
        # one log per rank; every other file (heads, cluster map, predictions, IoU tables) is rank 0's
        self.logfile = os.path.join(self.out_dir, f'one_shot_learner_{self.start_time}' +
                                    (f'_rank{self.rank}' if self.rank else '') + '.log')
        self.logger = get_logger('OneShot', self.logfile)
        self.summary_writer = None            # tensorboard is not a dependency of this build
        self.model_str, self.seg_str, self.train_str, self.test_str = model, segmentor, trainer, tester
        self.mode, self.inputs, self.device, self.exp_name = mode, inputs, device, exp_name
        self.synthetic = synthetic
        self.dist_group = dist_group
        self.logger.info("=" * 80)
        self.logger.info("One-Shot Learning Pipeline for StyleGANs (MI355X-native hot path)")
        self.logger.info("=" * 80 + '\n')
        self.logger.info(self.exp_name)
        self.configs = dict(model=config_mapper.models[self.model_str],
                            seg=config_mapper.segmentors[self.seg_str],
                            trainer=config_mapper.trainer[self.train_str])
        if custom is not None:
            self.configs.update(custom)
        self.load_model()
        self.load_segmentor()
        self.load_trainer()
        self.num_test_samples = num_test_samples

    # -------------------------------------------------------------------------
    def load_model(self):
        self.model_config = load_config(self.configs['model'], 'model_config')
        gen_cls = Generator
        if self.model_config.is_baggan:               # pidray-*: BagGANHQ(...).generator.module (reference :150-154)
            from models.baggan.models import StyleGANGenerator as gen_cls
        self.model = gen_cls(**self.model_config.gen_args)
        if self.synthetic:
            torch.manual_seed(42)
            self.model = gen_cls(**self.model_config.gen_args)
            for n, p in self.model.named_parameters():
                if n.endswith("noise.weight"):
                    p.data.fill_(0.1)
        else:
            sd = torch.load(self.model_config.model_path, map_location='cpu', weights_only=False)
            self.model.load_state_dict(sd["g_ema"] if 'g_ema' in sd.keys() else sd, strict=False)
        self.logger.info(f"Model Name: {self.model_str}")
        for k, v in self.model_config.gen_args.items():
            self.logger.info(f"{k}: {v}")
        self.model.eval()
        self.model = self.model.to(self.device)
        self.mean_latent = self.model.mean_latent(self.model_config.num_latents_for_mean)

    def load_segmentor(self):
        self.seg_config = load_config(self.configs['seg'], 'seg_config')
        if self.seg_str.find('hfc_with_swav') != -1:
            self.segmentor = hfc_with_swav.segmentor
            self.preprocessor = hfc_with_swav.preprocessor
        elif self.seg_str == 'hfc_with_simclr':
            import baseline.hfc_with_simclr.base as hfc_with_simclr
            self.segmentor = hfc_with_simclr.segmentor
            self.preprocessor = hfc_with_simclr.preprocessor
        elif self.seg_str == 'hfc_kmeans':               # built here, like the reference (:211-219)
            import baseline.hfc_kmeans.base as hfc_kmeans
            self.segmentor = hfc_kmeans.segmentor
            self.preprocessor = hfc_kmeans.preprocessor(model=self.model, model_config=self.model_config,
                                                        out_dir=self.out_dir, logger=self.logger,
                                                        **self.seg_config.hfc_prep_args)
        else:
            raise NotImplementedError(f"method '{self.seg_str}' is outside this build (SURVEY.md par. 8f)")

    def load_trainer(self):
        self.trainer_config = load_config(self.configs['trainer'], 'trainer_config')

    # -------------------------------------------------------------------------
    def _load_samples(self):
        """test latents / labels (src/one_shot_pipeline.py:356-384): `latents.pt` is either a tensor or a
        (latents, noises) tuple; labels may be a numpy array, are scaled by 255 when stored in [0, 1), and
        LSUN-car labels (384 rows) are centred in a square canvas."""
        if self.synthetic:
            n = (self.num_test_samples or 4) + 1
            torch.manual_seed(1234)
            self.test_latents = self.model.style(torch.randn(n, self.model_config.latent_dim, device=self.device))
            size, k = self.model.size, len(self.model_config.classes)
            yy, xx = torch.meshgrid(torch.arange(size), torch.arange(size), indexing="ij")
            lab = ((xx * k) // size + (yy * k) // size) % k              # synthetic partition: NOT ground truth
            self.test_labels = lab.unsqueeze(0).repeat(n, 1, 1).float()
            self.one_shot_ind = 0
            return
        # weights_only=False: labels.pt may hold a pickled numpy array (the reference predates torch's safe default)
        lat = torch.load(self.model_config.sample_latents, weights_only=False)
        self.test_latents = (lat[0] if isinstance(lat, (tuple, list)) else lat).to(self.device)
        lab = torch.load(self.model_config.sample_labels, weights_only=False)
        if not isinstance(lab, torch.Tensor):
            lab = torch.from_numpy(lab)
        if lab.max() < 1:
            lab = lab * 255
        if self.model_str.find('p-car') != -1:
            full = torch.zeros(lab.shape[0], lab.shape[2], lab.shape[2])
            full[:, 256 - 192:256 + 192, :] = lab.clone()
            lab = full
        self.test_labels = lab
        self.one_shot_ind = self.model_config.one_shot_ind

    def _resize_labels(self, lab, size):
        """transforms.Resize(image_size, interpolation=NEAREST) of the reference (:562-563), as an index"""
        if lab.shape[-1] == size and lab.shape[-2] == size:
            return lab
        return torch.nn.functional.interpolate(lab[:, None].float(), size=(size, size), mode="nearest")[:, 0]

    def run_pipeline(self, blocks_to_run=('setup', 'train', 'test')):
        if 'setup' in blocks_to_run:
            self._load_samples()
            ind = self.one_shot_ind
            self.one_shot_latent = self.test_latents[ind]
            self.one_shot_label = self.test_labels[ind:ind + 1]
            if self.num_test_samples is None:
                self.num_test_samples = self.test_latents.shape[0]
            self.num_test_samples = min(self.num_test_samples, self.test_latents.shape[0])
        if 'train' in blocks_to_run:
            self.run_trainer()
        if 'test' in blocks_to_run:
            self.run_tests()

    def run_trainer(self):
        """(:466-581) pretrain / load the clustering model, take the code map of the labelled latent and
        fine-tune the head on it with the trainer config (Adam + CrossEntropyLoss, StepLR per epoch)."""
        from ganecdotes_amd.oneshot import OneShotHead
        if self.seg_str == 'hfc_kmeans':                 # (:484-495) fit the per-layer k-means, take the label maps
            if self.seg_config.train_hfc:
                self.preprocessor.train_hfc_model(self.one_shot_latent.reshape(1, -1))
            self._predict = self.preprocessor.predict_hfc_vectors
            self.one_shot_features, self.one_shot_clusters = self._predict(self.one_shot_latent.reshape(1, -1))
            torch.save([l.cpu() for l in self.one_shot_clusters], os.path.join(self.out_dir, 'cluster_map.pt'))
        else:
            kw = dict(self.seg_config.hfc_prep_args)
            self.preprocessor = self.preprocessor(model=self.model, model_config=self.model_config,
                                                  out_dir=self.out_dir, logger=self.logger,
                                                  tb=self.summary_writer, device=self.device,
                                                  **({'dist_group': self.dist_group} if self.dist_group else {}), **kw)
            if self.seg_config.train_hfc:
                self.preprocessor.preprocess(self.one_shot_latent)
            if self.rank != 0:           # the pretraining was the data-parallel part: rank 0 fits the head and tests
                return
            self._predict = getattr(self.preprocessor, 'predict_swav_codes', None) or self.preprocessor.predict_simclr_codes
            self.one_shot_features, self.one_shot_clusters = self._predict(self.one_shot_latent.reshape(1, -1))
            self.one_shot_features = self.one_shot_features.detach().clone()
            torch.save(self.one_shot_clusters.cpu(), os.path.join(self.out_dir, 'cluster_map.pt'))
        tc = self.trainer_config
        self.head = OneShotHead(**self.seg_config.seg_args, n_class=len(self.model_config.classes), device=self.device)
        size = self.one_shot_features.shape[-1]
        label = self._resize_labels(self.one_shot_label, size).to(self.device, torch.long)
        t0 = time.time()
        self.head_losses = self.head.fit(self.one_shot_features.permute(0, 2, 3, 1), label, num_epochs=tc.num_epochs,
                                         lr=tc.lr, beta1=tc.beta1, beta2=tc.beta2, **tc.scheduler_args,
                                         print_freq=tc.print_freq, logger=self.logger)
        self.segmentor = self.head.to_module()          # the torch module the reference keeps (and moves to CPU, :610)
        self.logger.info(f'******* Training Complete ({time.time() - t0:.1f}s, final loss {self.head_losses[-1]:.4f}) ********')

    def run_tests(self):
        """(:595-886) predictions of the fine-tuned head on the test latents, label_predictions.pt and the
        IoU block (per-class mask IoU, area-weighted IoU, binary IoU)."""
        from ganecdotes_amd.oneshot import run_tests, write_result_files
        if self.rank != 0:
            return None

        class _Prep:                                    # run_tests only needs predict_swav_codes-shaped access
            predict_swav_codes = staticmethod(lambda lat: self._predict(lat))
        n = self.num_test_samples
        self.test_dir = os.path.join(self.out_dir, 'tests')
        os.makedirs(self.test_dir, exist_ok=True)
        size = self.one_shot_features.shape[-1]
        labels = self._resize_labels(self.test_labels[:n], size)
        # SimCLR's BatchNorm takes its statistics over the rows of ONE predict call: one latent per call, like the
        # reference (MAX_TEST_BATCH = 1, :30, :614-639); the per-pixel preprocessors are batch-independent
        # (16 latents per call at 256^2: 1147 images/s against 919 at 4 — the low-resolution layers fill the chip; 8 at 512^2)
        per_call = 1 if self.seg_str == 'hfc_with_simclr' else (16 if size <= 256 else 8)
        pred, self.results = run_tests(_Prep, self.head, self.test_latents[:n], labels, self.model_config.classes,
                                       batch=per_call)
        torch.save(pred, os.path.join(self.test_dir, 'label_predictions.pt'))
        classes = self.model_config.classes
        write_result_files(self.test_dir, self.results, classes)     # mask / bb IoU tables, results.npz (:775-791, :884)
        mean_iou = {c: float(sum(m[c] for m in self.results['mask_iou']) / n) for c in classes}
        self.results['mean_mask_iou'] = mean_iou
        self.results['mIoU'] = float(sum(mean_iou.values()) / len(classes))
        self.logger.info(f"mask IoU per class: {mean_iou}")
        self.logger.info(f"mIoU {self.results['mIoU']:.4f} | weighted IoU {sum(self.results['w_iou']) / n:.4f} | "
                         f"binary IoU (last sample) {self.results['bin_iou']:.4f}"
                         + ("   [synthetic labels: not a quality measure]" if self.synthetic else ""))
        return self.results
