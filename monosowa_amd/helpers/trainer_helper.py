"""Epoch loop (reference: lib/helpers/trainer_helper.py -- ``Trainer.__init__`` :15-63, ``train``
:65-114, ``train_one_epoch`` :116-178, ``prepare_targets`` :180-191).  Same constructor signature and
checkpoint naming.  Differences that do not change results:

* ``torch.distributed`` aware: under ``torchrun`` the model is wrapped in DistributedDataParallel
  (bucketed RCCL all-reduce overlapped with backward); rank 0 alone writes checkpoints and logs.
* the per-iteration ``.item()`` of ~30 loss terms (a host sync every step, :153-157) happens only on
  the logging iterations (every 30th, as printed by the reference).
"""
import os

import numpy as np
import torch
import tqdm

from ..monodetr import misc
from ..monodetr.criterion import weighted_total
from ..synthetic import attach_host_mask
from ..synthetic import prepare_targets as _prepare_targets
from .save_helper import get_checkpoint_state, load_checkpoint, save_checkpoint, unwrap


def wrap_ddp(model, device):
    """DDP with few large buckets (xGMI rings are per-link bound) and the never-used parameters frozen."""
    force = os.environ.get("MONOSOWA_FORCE_DDP") == "1"        # rehearse the DDP path on a 1-GPU box
    if not (misc.is_dist_avail_and_initialized() and (misc.get_world_size() > 1 or force)):
        return model
    core = unwrap(model)
    unused = set(core.unused_parameter_names()) if hasattr(core, "unused_parameter_names") else set()
    for n, p in core.named_parameters():
        if n in unused:
            p.requires_grad_(False)
    ids = [device.index] if device.type == "cuda" else None
    return torch.nn.parallel.DistributedDataParallel(core, device_ids=ids, bucket_cap_mb=64,
                                                     gradient_as_bucket_view=True, broadcast_buffers=False)


_EVAL_WAIT_GROUP = []


def wait_for_evaluation(timeout_hours=6.0):
    """All ranks meet here after rank 0's checkpoint / evaluation block: a barrier on a dedicated gloo group (host-side, created
    on first use by every rank at the same point of the epoch loop) whose timeout covers a whole KITTI val inference + AP run."""
    import datetime
    if not _EVAL_WAIT_GROUP:
        _EVAL_WAIT_GROUP.append(torch.distributed.new_group(backend="gloo", timeout=datetime.timedelta(hours=timeout_hours)))
    torch.distributed.monitored_barrier(group=_EVAL_WAIT_GROUP[0], timeout=datetime.timedelta(hours=timeout_hours))


def stage_batch(raw, device):
    """One collated loader batch ``(inputs, calibs, targets, info)`` onto ``device`` (reference: trainer_helper.py:121-127, a
    per-key ``.to(device)``): non-blocking copies (pinned source buffers when the loader pins), images to channels-last for the
    MIOpen NHWC kernels, and the object mask kept on the host next to its device copy so that ``prepare_targets`` needs no
    device -> host synchronisation."""
    inputs, calibs, targets, info = raw
    inputs = inputs.to(device, non_blocking=True)
    if inputs.is_cuda:
        inputs = inputs.contiguous(memory_format=torch.channels_last)
    calibs = calibs.to(device, non_blocking=True)
    host_mask = targets["mask_2d"].numpy() if not targets["mask_2d"].is_cuda else None
    targets = {k: v.to(device, non_blocking=True) for k, v in targets.items()}
    if host_mask is not None:
        attach_host_mask(targets["mask_2d"], host_mask)      # prepare_targets then needs no device sync
    return inputs, calibs, targets, info


class Trainer(object):
    def __init__(self, cfg, model, optimizer, train_loader, test_loader, lr_scheduler, warmup_lr_scheduler,
                 logger, loss, model_name):
        self.cfg = cfg
        self.optimizer = optimizer
        self.train_loader = train_loader
        self.test_loader = test_loader
        self.lr_scheduler = lr_scheduler
        self.warmup_lr_scheduler = warmup_lr_scheduler
        self.logger = logger
        self.epoch = 0
        self.best_result = 0
        self.best_epoch = 0
        if torch.cuda.is_available():
            self.device = torch.device("cuda", torch.cuda.current_device())
        else:
            self.device = torch.device("cpu")
        if self.device.type == "cuda":
            from .model_helper import to_mi355x_layout
            to_mi355x_layout(unwrap(model))
        self.model = wrap_ddp(model, self.device)
        self.detr_loss = loss
        self.model_name = model_name
        self.output_dir = os.path.join("./" + cfg["save_path"], model_name)
        self.tester = None
        self.log_interval = 30

        if cfg.get("pretrain_model"):
            assert os.path.exists(cfg["pretrain_model"])
            load_checkpoint(model=self.model, optimizer=None, filename=cfg["pretrain_model"],
                            map_location=self.device, logger=self.logger)
        if cfg.get("resume_model", None):
            resume = os.path.join(self.output_dir, "checkpoint.pth")
            assert os.path.exists(resume)
            self.epoch, self.best_result, self.best_epoch = load_checkpoint(
                model=self.model.to(self.device), optimizer=self.optimizer, filename=resume,
                map_location=self.device, logger=self.logger)
            self.lr_scheduler.last_epoch = self.epoch - 1
            self.logger.info("Loading Checkpoint... Best Result:{}, Best Epoch:{}".format(self.best_result, self.best_epoch))

    def train(self):
        start_epoch = self.epoch
        best_result, best_epoch = self.best_result, self.best_epoch
        main = misc.is_main_process()
        bar = tqdm.tqdm(range(start_epoch, self.cfg["max_epoch"]), dynamic_ncols=True, leave=True, desc="epochs", disable=not main)
        for epoch in range(start_epoch, self.cfg["max_epoch"]):
            np.random.seed(np.random.get_state()[1][0] + epoch)
            sampler = getattr(self.train_loader, "sampler", None)
            if hasattr(sampler, "set_epoch"):
                sampler.set_epoch(epoch)
            self.train_one_epoch(epoch)
            self.epoch += 1
            if self.warmup_lr_scheduler is not None and epoch < 5:
                self.warmup_lr_scheduler.step()
            else:
                self.lr_scheduler.step()
            if (self.epoch % self.cfg["save_frequency"]) == 0 and main:
                os.makedirs(self.output_dir, exist_ok=True)
                name = "checkpoint_epoch_%d" % self.epoch if self.cfg["save_all"] else "checkpoint"
                save_checkpoint(get_checkpoint_state(self.model, self.optimizer, self.epoch, best_result, best_epoch),
                                os.path.join(self.output_dir, name))
                if self.tester is not None:
                    self.logger.info("Test Epoch {}".format(self.epoch))
                    self.tester.inference()
                    cur = self.tester.evaluate()
                    if cur > best_result:
                        best_result, best_epoch = cur, self.epoch
                        save_checkpoint(get_checkpoint_state(self.model, self.optimizer, self.epoch, best_result, best_epoch),
                                        os.path.join(self.output_dir, "checkpoint_best"))
                    self.logger.info("Best Result:{}, epoch:{}".format(best_result, best_epoch))
            if (self.epoch % self.cfg["save_frequency"]) == 0 and misc.is_dist_avail_and_initialized():
                # rank 0 saved / evaluated alone: the other ranks wait here instead of inside the next epoch's first
                # gradient all-reduce.  The wait runs on its own gloo group with a timeout sized for a full validation pass
                # (the default group's collective timeout -- 10 minutes under RCCL -- would abort a long evaluation just the
                # same, only inside this barrier)
                wait_for_evaluation()
            bar.update()
        self.logger.info("Best Result:{}, epoch:{}".format(best_result, best_epoch))
        return None

    def train_step(self, inputs, calibs, targets, info=None):
        """One optimizer step on a device-resident batch; returns (total loss tensor, loss dict)."""
        img_sizes = targets["img_size"]
        target_list = self.prepare_targets(targets, inputs.shape[0])
        self.optimizer.zero_grad(set_to_none=True)
        outputs = self.model(inputs, calibs, target_list, img_sizes, dn_args=None)
        loss_dict = self.detr_loss(outputs, target_list, None, info)
        weight_dict = self.detr_loss.weight_dict
        total = weighted_total(loss_dict, weight_dict)
        total.backward()
        self.optimizer.step()
        return total, loss_dict

    def train_one_epoch(self, epoch):
        torch.set_grad_enabled(True)
        self.model.train()
        self.detr_loss.train()
        main = misc.is_main_process()
        if main:
            print(">>>>>>> Epoch:", str(epoch) + ":")
        bar = tqdm.tqdm(total=len(self.train_loader), leave=(self.epoch + 1 == self.cfg["max_epoch"]), desc="iters", disable=not main)
        for batch_idx, raw in enumerate(self.train_loader):
            inputs, calibs, targets, info = stage_batch(raw, self.device)
            total, loss_dict = self.train_step(inputs, calibs, targets, info)
            if batch_idx % self.log_interval == 0:
                self._log(batch_idx, loss_dict)
            bar.update()
        bar.close()
        # the device assignment solver reports an invalid cost matrix through a status word (no exception from a kernel, no wait in
        # the step): look at it for certain before the epoch's checkpoint is written
        matcher = getattr(self.detr_loss, "matcher", None)
        if matcher is not None and hasattr(matcher, "check_device_status"):
            matcher.check_device_status(block=True)

    def _log(self, batch_idx, loss_dict):
        weight_dict = self.detr_loss.weight_dict
        reduced = misc.reduce_dict({k: v for k, v in loss_dict.items() if k in weight_dict})
        if not misc.is_main_process():
            return
        logged = {k: (reduced[k] * weight_dict[k]).item() for k in reduced}
        print("----", batch_idx, "----")
        print("%s: %.2f, " % ("loss_detr", sum(logged.values())))
        seen = set()
        for key, val in logged.items():
            if key[-1].isdigit() and key[-1] not in seen:
                print("")
                seen.add(key[-1])
            print("%s: %.2f, " % (key, val), end="")
        print("\n")

    @staticmethod
    def prepare_targets(targets, batch_size):
        return _prepare_targets(targets, batch_size)
