"""`load_config` / `get_logger` with the reference's semantics (lib/util/util.py:31-83) and its
import-time seeding (numpy 42, torch 42; lib/util/util.py:21-28)."""
import importlib.util
import logging
import random
import sys

import numpy as np
import torch

seed = 42
np.random.seed(seed)
random.seed(seed)
torch.manual_seed(seed)


def load_config(cfg_path, cfg_name='name'):
    """exec a config file as a module and return it"""
    import os
    assert os.path.exists(cfg_path), "Config file not found!"
    spec = importlib.util.spec_from_file_location(cfg_name, cfg_path)
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module


def get_logger(lname, logfile):
    logger = logging.getLogger(lname)
    logger.setLevel(logging.INFO)
    fmt = logging.Formatter('[%(asctime)s] [%(name)s] %(levelname)s: %(message)s')
    for h in (logging.FileHandler(logfile, mode='a'), logging.StreamHandler(sys.stdout)):
        h.setFormatter(fmt)
        logger.addHandler(h)
    return logger
