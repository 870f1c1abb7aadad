"""``build_dataloader(cfg['dataset']) -> (train_loader, test_loader)``
(reference: lib/helpers/dataloader_helper.py:12-36: batch_size from the config, 4 workers, shuffle on
train).  ``dataset.type: KITTI`` reads a KITTI-format directory (monosowa_amd/kitti_dataset.py), ``synthetic`` serves seeded KITTI-shaped samples; under torch.distributed each
rank gets a disjoint shard (DistributedSampler), which the reference (single process) never needed."""
import numpy as np
import torch
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

from ..monodetr.misc import get_world_size, is_dist_avail_and_initialized
from ..synthetic import SyntheticKITTI


def my_worker_init_fn(worker_id):
    np.random.seed(np.random.get_state()[1][0] + worker_id)


def build_dataset(cfg, split):
    if cfg["type"] == "synthetic":
        return SyntheticKITTI(split=split, cfg=cfg)
    if cfg["type"] == "KITTI":
        from ..kitti_dataset import KITTI_Dataset
        return KITTI_Dataset(split=split, cfg=cfg)
    raise NotImplementedError("%s dataset is not supported" % cfg["type"])


def build_dataloader(cfg, workers=4, drop_last=False, test=True):
    """``drop_last`` / ``test=False`` (no test loader: None) are extensions for bench.py's DataLoader-fed leg."""
    train_set = build_dataset(cfg, cfg["train_split"])
    test_set = build_dataset(cfg, cfg["test_split"]) if test else None
    sampler = DistributedSampler(train_set, shuffle=True) if is_dist_avail_and_initialized() and get_world_size() > 1 else None
    train_loader = DataLoader(train_set, batch_size=cfg["batch_size"], num_workers=workers, worker_init_fn=my_worker_init_fn,
                              shuffle=sampler is None, sampler=sampler, pin_memory=torch.cuda.is_available(), drop_last=drop_last)
    if test_set is None:
        return train_loader, None
    test_loader = DataLoader(test_set, batch_size=cfg["batch_size"], num_workers=workers, worker_init_fn=my_worker_init_fn,
                             shuffle=False, pin_memory=torch.cuda.is_available(), drop_last=False)
    return train_loader, test_loader
