"""Checkpoint dictionary {'epoch','model_state','optimizer_state','best_result','best_epoch'} saved
as '<name>.pth' (reference: lib/helpers/save_helper.py:6-45).  DataParallel / DDP wrappers are
unwrapped on save so checkpoints are interchangeable with the reference's."""
import os

import torch
import torch.nn as nn


def unwrap(model):
    return model.module if isinstance(model, (nn.DataParallel, nn.parallel.DistributedDataParallel)) else model


def model_state_to_cpu(model_state):
    out = type(model_state)()
    for k, v in model_state.items():
        out[k] = v.cpu()
    return out


def get_checkpoint_state(model=None, optimizer=None, epoch=None, best_result=None, best_epoch=None):
    optim_state = optimizer.state_dict() if optimizer is not None else None
    if model is None:
        model_state = None
    elif unwrap(model) is not model:
        model_state = model_state_to_cpu(unwrap(model).state_dict())
    else:
        model_state = model.state_dict()
    return {"epoch": epoch, "model_state": model_state, "optimizer_state": optim_state,
            "best_result": best_result, "best_epoch": best_epoch}


def save_checkpoint(state, filename):
    torch.save(state, "{}.pth".format(filename))


def load_checkpoint(model, optimizer, filename, map_location, logger=None):
    if not os.path.isfile(filename):
        raise FileNotFoundError(filename)
    if logger is not None:
        logger.info("==> Loading from checkpoint '{}'".format(filename))
    checkpoint = torch.load(filename, map_location, weights_only=False)
    epoch = checkpoint.get("epoch", -1)
    best_result = checkpoint.get("best_result", 0.0)
    best_epoch = checkpoint.get("best_epoch", 0.0)
    if model is not None and checkpoint["model_state"] is not None:
        unwrap(model).load_state_dict(checkpoint["model_state"])
    if optimizer is not None and checkpoint["optimizer_state"] is not None:
        optimizer.load_state_dict(checkpoint["optimizer_state"])
    if logger is not None:
        logger.info("==> Done")
    return epoch, best_result, best_epoch
