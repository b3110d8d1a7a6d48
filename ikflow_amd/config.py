"""Global configuration (mirror of ikflow/config.py:6-8: DEVICE re-exported from jrl.config, DEFAULT_TORCH_DTYPE)."""
import os

import torch

# jrl.config.DEVICE is "cuda:<idx>" when a GPU is visible, else "cpu". On "cpu" this package can still be imported
# and its host logic exercised, but every compute call raises: there is no CPU path.
DEVICE = f"cuda:{int(os.environ.get('LOCAL_RANK', 0))}" if torch.cuda.is_available() else "cpu"
DEFAULT_TORCH_DTYPE = torch.float32

# ~/.cache/ikflow/ - same cache layout as the reference (ikflow/config.py:12-18), so weight files it downloaded are found
DEFAULT_DATA_DIR = os.path.join(os.path.expanduser("~"), ".cache/ikflow/")
MODELS_DIR = os.path.join(DEFAULT_DATA_DIR, "models/")
