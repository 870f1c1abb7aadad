"""Logger and seeding (reference: lib/helpers/utils_helper.py:6-26)."""
import logging
import random

import numpy as np
import torch


def create_logger(log_file, rank=0):
    log_format = "%(asctime)s  %(levelname)5s  %(message)s"
    logging.basicConfig(level=logging.INFO if rank == 0 else "ERROR", format=log_format, filename=log_file)
    console = logging.StreamHandler()
    console.setLevel(logging.INFO if rank == 0 else "ERROR")
    console.setFormatter(logging.Formatter(log_format))
    logging.getLogger(__name__).addHandler(console)
    return logging.getLogger(__name__)


def set_random_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed ** 2)
    torch.cuda.manual_seed(seed ** 3)
    torch.backends.cudnn.deterministic = True      # MIOpen honours the same switch on ROCm
    torch.backends.cudnn.benchmark = False
