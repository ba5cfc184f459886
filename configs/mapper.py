"""Registry of config modules: names -> paths, exactly the keys of the reference's
configs/mapper.py:9-67 (models, segmentors, trainer, losses, lr_scheduler)."""
import os

import torch.nn as nn
import torch.optim.lr_scheduler as sch

CONFIGS_DIR = os.path.dirname(os.path.abspath(__file__))


def _m(name):
    return os.path.join(CONFIGS_DIR, 'models', name + '.py')


def _s(name):
    return os.path.join(CONFIGS_DIR, 'segmentors', name + '_config.py')


models = {key: _m(mod) for key, mod in {
    'ffhq-256': 'ffhq_256', 'ffhq-256-er': 'ffhq_256_rp_earr', 'ffhq-256-eg': 'ffhq_256_rp_eyeg',
    'car-512': 'lsun_car_512', 'cat-256': 'lsun_cat_256', 'horse-256': 'lsun_horse_256',
    'horse-256-rp': 'lsun_horse_256_rp', 'church-256': 'lsun_church_256', 'church-512': 'lsun_church_512',
    'pidray-256': 'pidray_bag_256', 'pidray-pliers-256': 'pidray_pliers_256',
    'pidray-hammer-256': 'pidray_hammer_256', 'pidray-powerbank-256': 'pidray_powerbank_256',
    'pidray-wrench-256': 'pidray_wrench_256', 'pidray-handcuffs-256': 'pidray_handcuffs_256',
    'celeba-256': 'celebamask_ffhq_im_256_n_100', 'p-horse-256': 'pascal_horse_256',
    'p-car-512': 'pascal_car_512', 'afhq-256': 'afhq_256'}.items()}

segmentors = {name: _s(name) for name in (
    'repurposegan', 'datasetgan', 'hfc_with_swav', 'hfc_with_simclr', 'hfc_kmeans', 'hfc_with_swav_cat',
    'hfc_with_swav_car', 'hfc_with_swav_ffhq', 'hfc_with_swav_horse', 'hfc_with_swav_pidray')}

trainer = {'supervised': os.path.join(CONFIGS_DIR, 'trainers', 'supervised_config.py')}

tester = {name: os.path.join(CONFIGS_DIR, 'tester', fname + '_config.py') for name, fname in (
    ('iou', 'iou'), ('roc', 'roc'), ('prcurve', 'prcurve'), ('dice', 'dice'), ('conf_mat', 'conf_mat'),
    ('all', 'all'))}      # "tester module - not used" in the reference (mapper.py:48)

losses = {'bce': nn.BCEWithLogitsLoss, 'softmax': nn.Softmax, 'sigmoid': nn.Sigmoid, 'tanh': nn.Tanh,
          'logloss': nn.LogSoftmax, 'cross_entropy': nn.CrossEntropyLoss}

lr_scheduler = {'step': sch.StepLR, 'plateau': sch.ReduceLROnPlateau, 'cosine': sch.CosineAnnealingLR}
