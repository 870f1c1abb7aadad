"""Small shared utilities of the detector (the used subset of the reference's utils/misc.py).

* ``NestedTensor``                       utils/misc.py:287-332  (tensors + padding mask pair)
* ``inverse_sigmoid``                    utils/misc.py:473-477
* ``accuracy``                           utils/misc.py:435-450  (top-k precision, logging only)
* ``get_world_size`` / ``is_dist_avail_and_initialized`` / ``reduce_dict``   utils/misc.py:135-159,381-400
"""
import torch
import torch.distributed as dist


class NestedTensor(object):
    def __init__(self, tensors, mask, all_valid=False):
        self.tensors = tensors
        self.mask = mask
        # True when the producer built ``mask`` as all-False (no padding anywhere): consumers may then skip
        # masking passes without looking at the device tensor
        self.all_valid = all_valid

    def to(self, device, non_blocking=False):
        m = self.mask.to(device, non_blocking=non_blocking) if self.mask is not None else None
        return NestedTensor(self.tensors.to(device, non_blocking=non_blocking), m)

    def decompose(self):
        return self.tensors, self.mask

    def __repr__(self):
        return str(self.tensors)


def inverse_sigmoid(x, eps=1e-5):
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


@torch.no_grad()
def accuracy(output, target, topk=(1,)):
    """Precision@k in percent for each k."""
    if target.numel() == 0:
        return [torch.zeros([], device=output.device)]
    maxk = max(topk)
    _, pred = output.topk(maxk, 1, True, True)
    correct = pred.t().eq(target.view(1, -1).expand_as(pred.t()))
    return [correct[:k].reshape(-1).float().sum(0).mul_(100.0 / target.size(0)) for k in topk]


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def reduce_dict(input_dict, average=True):
    """All-reduce a dict of scalar tensors (logging).  One stacked collective, sorted keys."""
    world_size = get_world_size()
    if world_size < 2:
        return input_dict
    with torch.no_grad():
        names = sorted(input_dict.keys())
        values = torch.stack([input_dict[k].detach().float().reshape(()) for k in names], dim=0)
        dist.all_reduce(values)
        if average:
            values /= world_size
        return {k: v for k, v in zip(names, values)}
