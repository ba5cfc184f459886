"""Slim orchestrator with the reference's contract (src/one_shot_pipeline.py:33-465):
`OneShotPipeline(out_dir, exp_name, model, segmentor, trainer, tester, mode, inputs, custom,
device, num_test_samples)`, `load_model`, `load_segmentor`, `load_trainer`,
`run_pipeline(blocks_to_run)`.  It drives ONLY the accelerated path: generator load,
`preprocessor.preprocess` (SwAV pretrain) and `predict_swav_codes`; the one-shot fine-tune,
IoU test loop, plotting and the GUI are outside this build (SURVEY.md par. 8f).

`synthetic=True` (opt-in extra): random-init generator under seed 42 and random latents,
because no checkpoints / latents / labels exist offline (reference: checkpoints/ is empty).
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
        self.logfile = os.path.join(self.out_dir, f'one_shot_learner_{self.start_time}.log')
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
        if self.model_config.is_baggan:
            raise NotImplementedError("BagGAN (pidray-*) generators are outside this build (SURVEY.md par. 8f)")
        self.model = Generator(**self.model_config.gen_args)
        if self.synthetic:
            torch.manual_seed(42)
            self.model = Generator(**self.model_config.gen_args)
            for n, p in self.model.named_parameters():
                if n.endswith("noise.weight"):
                    p.data.fill_(0.1)
        else:
            sd = torch.load(self.model_config.model_path, map_location='cpu')
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
        else:
            raise NotImplementedError(f"method '{self.seg_str}' is outside this build (SURVEY.md par. 8f)")

    def load_trainer(self):
        self.trainer_config = load_config(self.configs['trainer'], 'trainer_config')

    # -------------------------------------------------------------------------
    def run_pipeline(self, blocks_to_run=('setup', 'train')):
        if 'setup' in blocks_to_run:
            if self.synthetic:
                z = torch.randn(1, self.model_config.latent_dim).to(self.device)
                self.one_shot_latent = self.model.style(z)
            else:
                lat = torch.load(self.model_config.sample_latents)
                lat = lat[0] if isinstance(lat, tuple) else lat
                self.one_shot_latent = lat[self.model_config.one_shot_ind].to(self.device)
        if 'train' in blocks_to_run:
            self.run_trainer()

    def run_trainer(self):
        kw = dict(self.seg_config.hfc_prep_args)
        self.preprocessor = self.preprocessor(model=self.model, model_config=self.model_config,
                                              out_dir=self.out_dir, logger=self.logger,
                                              tb=self.summary_writer, device=self.device,
                                              **({'dist_group': self.dist_group} if self.dist_group else {}), **kw)
        if self.seg_config.train_hfc:
            self.preprocessor.preprocess(self.one_shot_latent)
        predict = getattr(self.preprocessor, 'predict_swav_codes', None) or self.preprocessor.predict_simclr_codes
        self.one_shot_features, self.one_shot_clusters = predict(self.one_shot_latent.reshape(1, -1))
        torch.save(self.one_shot_clusters.cpu(), os.path.join(self.out_dir, 'cluster_map.pt'))
        self.logger.info(f"cluster map {tuple(self.one_shot_clusters.shape)} saved; one-shot fine-tune / IoU "
                         f"test loop are not part of this build")
