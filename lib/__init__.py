import os

ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS_DIR = os.path.join(ROOT_DIR, 'configs')
CHECKPOINT_DIR = os.path.join(ROOT_DIR, 'checkpoints')
RESULTS_DIR = os.path.join(ROOT_DIR, 'results')
