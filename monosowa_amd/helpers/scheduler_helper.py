"""Step-decay LR schedule with optional 5-epoch cosine warm-up
(reference: lib/helpers/scheduler_helper.py:6-18, CosineWarmupLR :60-76)."""
import math

import torch.optim.lr_scheduler as lr_sched


class CosineWarmupLR(lr_sched._LRScheduler):
    def __init__(self, optimizer, num_epoch, init_lr=0.0, last_epoch=-1):
        self.num_epoch = num_epoch
        self.init_lr = init_lr
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        return [self.init_lr + (base_lr - self.init_lr) * (1 - math.cos(math.pi * self.last_epoch / self.num_epoch)) / 2
                for base_lr in self.base_lrs]


def build_lr_scheduler(cfg, optimizer, last_epoch):
    def lr_lbmd(cur_epoch):
        decay = 1
        for step in cfg["decay_list"]:
            if cur_epoch >= step:
                decay = decay * cfg["decay_rate"]
        return decay

    lr_scheduler = lr_sched.LambdaLR(optimizer, lr_lbmd, last_epoch=last_epoch)
    warmup = CosineWarmupLR(optimizer, num_epoch=5, init_lr=0.00001) if cfg["warmup"] else None
    return lr_scheduler, warmup
