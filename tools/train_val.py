#!/usr/bin/env python
"""Train / evaluate MonoDETR on MI355X -- same CLI as the reference's MonoDETR/tools/train_val.py:30-33
(`--config X [-e]`), same flow (:36-125): seed, output dir + config copy, logger, dataloaders, model + criterion,
optimizer, LR schedule, Trainer (+ Tester for validation), or evaluation only.

Single GPU:   python tools/train_val.py --config configs/monodetr.yaml
Multi GPU:    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/train_val.py --config ...
(one process per GPU, DDP over RCCL; the reference's single-process nn.DataParallel / `gpu_ids` is not used).
"""
import argparse
import datetime
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# measured MIOpen / GEMM kernel choices for the shipped workload (harmless for other shapes: misses fall back to the
# libraries' heuristics); must be set before torch loads MIOpen
from monosowa_amd import miopen_tuning                                    # noqa: E402
miopen_tuning.use_shipped_db(int(os.environ.get("RANK", "0")))

import torch                                                               # noqa: E402
import yaml                                                                # noqa: E402

from monosowa_amd.helpers.dataloader_helper import build_dataloader      # noqa: E402
from monosowa_amd.helpers.model_helper import build_model                 # noqa: E402
from monosowa_amd.helpers.optimizer_helper import build_optimizer         # noqa: E402
from monosowa_amd.helpers.save_helper import load_checkpoint              # noqa: E402,F401
from monosowa_amd.helpers.scheduler_helper import build_lr_scheduler      # noqa: E402
from monosowa_amd.helpers.tester_helper import Tester                     # noqa: E402
from monosowa_amd.helpers.trainer_helper import Trainer                   # noqa: E402
from monosowa_amd.helpers.utils_helper import create_logger, set_random_seed   # noqa: E402


def main():
    parser = argparse.ArgumentParser(description="Depth-aware Transformer for Monocular 3D Object Detection (MI355X)")
    parser.add_argument("--config", dest="config", help="settings of detection in yaml format")
    parser.add_argument("-e", "--evaluate_only", action="store_true", default=False, help="evaluation only")
    parser.add_argument("--workers", type=int, default=4)
    args = parser.parse_args()
    assert os.path.exists(args.config)
    cfg = yaml.load(open(args.config, "r"), Loader=yaml.Loader)
    set_random_seed(cfg.get("random_seed", 444))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.distributed.init_process_group(backend="nccl" if torch.cuda.is_available() else "gloo")

    model_name = cfg["model_name"]
    output_path = os.path.join("./" + cfg["trainer"]["save_path"], model_name)
    os.makedirs(output_path, exist_ok=True)
    if rank == 0 and os.path.abspath(os.path.dirname(args.config)) != os.path.abspath(output_path):
        shutil.copy(args.config, output_path)
    log_file = os.path.join(output_path, "train.log.%s" % datetime.datetime.now().strftime("%Y%m%d_%H%M%S"))
    logger = create_logger(log_file, rank)

    train_loader, test_loader = build_dataloader(cfg["dataset"], workers=args.workers)
    model, loss = build_model(cfg["model"])
    device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    model = model.to(device)
    res = cfg["dataset"].get("resolution", (1280, 384))
    loss.depth_map_size = (res[0] // 16, res[1] // 16)

    if args.evaluate_only:
        logger.info("###################  Evaluation Only  ##################")
        tester = Tester(cfg=cfg["tester"], model=model, dataloader=test_loader, logger=logger, train_cfg=cfg["trainer"],
                        model_name=model_name)
        tester.test()
        return

    optimizer = build_optimizer(cfg["optimizer"], model)
    lr_scheduler, warmup_lr_scheduler = build_lr_scheduler(cfg["lr_scheduler"], optimizer, last_epoch=-1)
    if cfg["continue_train"]:
        path = os.path.join(output_path, "checkpoint_best.pth")
        if os.path.exists(path):
            print("Loading checkpoint from %s" % path)
            load_checkpoint(model, optimizer, path, device, logger)

    trainer = Trainer(cfg=cfg["trainer"], model=model, optimizer=optimizer, train_loader=train_loader,
                      test_loader=test_loader, lr_scheduler=lr_scheduler, warmup_lr_scheduler=warmup_lr_scheduler,
                      logger=logger, loss=loss, model_name=model_name)
    tester = Tester(cfg=cfg["tester"], model=trainer.model, dataloader=test_loader, logger=logger,
                    train_cfg=cfg["trainer"], model_name=model_name)
    if cfg["dataset"]["test_split"] != "test":
        trainer.tester = tester

    logger.info("###################  Training  ##################")
    logger.info("Batch Size: %d" % (cfg["dataset"]["batch_size"]))
    logger.info("Learning Rate: %f" % (cfg["optimizer"]["lr"]))
    trainer.train()
    if cfg["dataset"]["test_split"] == "test":
        return
    logger.info("###################  Testing  ##################")
    logger.info("Batch Size: %d" % (cfg["dataset"]["batch_size"]))
    logger.info("Split: %s" % (cfg["dataset"]["test_split"]))
    if rank == 0:
        tester.test()


if __name__ == "__main__":
    main()
