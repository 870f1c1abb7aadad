"""Inference loop and KITTI result writer (reference: lib/helpers/tester_helper.py -- ``Tester``
:14-28, ``test`` :28-78, ``inference`` :80-166, ``save_results`` :168-188, ``evaluate`` :190-194).

Model time is measured with device synchronisation (the reference's ``time.time()`` pair at :94-99
brackets an asynchronous launch).  Open3D visualisation (:196-258) is out of scope; KITTI AP
evaluation is delegated to ``dataset.eval`` when the dataset provides it."""
import glob
import os
import time

import numpy as np
import torch
import tqdm

from .decode_helper import PinholeCalib, decode_detections, extract_dets_from_outputs
from .save_helper import unwrap, load_checkpoint


class GraphedForward:
    """Eval forward replayed from a captured hipGraph.  The eval forward enqueues ~1,500 small kernels and
    never synchronises with the host, so at batch 16 it is launch-bound; one graph launch replaces them.
    Inputs are copied into static buffers; outputs are the graph's static output tensors (clone to keep)."""

    def __init__(self, model, images, calibs, img_sizes, warmup=2):
        self.model = model.eval()
        self.images, self.calibs, self.img_sizes = images.clone(), calibs.clone(), img_sizes.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):
                self.model(self.images, self.calibs, None, self.img_sizes)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.outputs = self.model(self.images, self.calibs, None, self.img_sizes)

    def __call__(self, images, calibs, img_sizes):
        self.images.copy_(images, non_blocking=True)
        self.calibs.copy_(calibs, non_blocking=True)
        self.img_sizes.copy_(img_sizes, non_blocking=True)
        self.graph.replay()
        return self.outputs


class Tester(object):
    def __init__(self, cfg, model, dataloader, logger, train_cfg=None, model_name="monodetr"):
        self.cfg = cfg
        # the bare module: inference runs on one rank at a time, and a DistributedDataParallel forward would start
        # collectives (buffer broadcast) that the other ranks never join
        self.model = unwrap(model)
        self.dataloader = dataloader
        self.max_objs = dataloader.dataset.max_objs
        self.class_name = dataloader.dataset.class_name
        self.output_dir = os.path.join("./" + train_cfg["save_path"], model_name)
        self.dataset_type = cfg.get("type", "KITTI")
        self.device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self.logger = logger
        self.train_cfg = train_cfg
        self.model_name = model_name
        self.last_img_per_s = None

    def test(self):
        assert self.cfg["mode"] in ["single", "all"]
        if self.cfg["mode"] == "single":
            path = os.path.join(self.output_dir, "checkpoint_epoch_{}.pth".format(self.cfg["checkpoint"]))
            load_checkpoint(model=self.model, optimizer=None, filename=path, map_location=self.device, logger=self.logger)
            self.model.to(self.device)
            self.inference()
            return self.evaluate()
        ckpts = sorted(glob.glob(os.path.join(self.output_dir, "checkpoint_epoch_*.pth")), key=os.path.getmtime)
        result = None
        for path in ckpts:
            load_checkpoint(model=self.model, optimizer=None, filename=path, map_location=self.device, logger=self.logger)
            self.model.to(self.device)
            self.inference()
            result = self.evaluate()
        return result

    @torch.no_grad()
    def inference(self):
        self.model.eval()
        results, model_time, n_img = {}, 0.0, 0
        bar = tqdm.tqdm(total=len(self.dataloader), leave=True, desc="Evaluation Progress")
        for inputs, calibs, targets, info in self.dataloader:
            inputs = inputs.to(self.device)
            calibs_dev = calibs.to(self.device)
            img_sizes = info["img_size"].to(self.device).clone()
            img_sizes[:, 1] = img_sizes[:, 1] / info["height_crop"].to(self.device)
            if self.device.type == "cuda":
                torch.cuda.synchronize()
            t0 = time.time()
            outputs = self.model(inputs, calibs_dev, targets, img_sizes, dn_args=0)
            if self.device.type == "cuda":
                torch.cuda.synchronize()
            model_time += time.time() - t0
            n_img += inputs.shape[0]
            dets = extract_dets_from_outputs(outputs=outputs, K=self.max_objs, topk=self.cfg["topk"]).cpu().numpy()
            dataset = self.dataloader.dataset
            if hasattr(dataset, "get_calib"):
                cal = [dataset.get_calib(int(i)) for i in info["img_id"]]
            else:
                cal = [PinholeCalib(p) for p in calibs.numpy()]
            info_np = {k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in info.items()}
            results.update(decode_detections(dets=dets, info=info_np, calibs=cal, cls_mean_size=dataset.cls_mean_size,
                                             threshold=self.cfg.get("threshold", 0.2)))
            bar.update()
        bar.close()
        self.last_img_per_s = n_img / max(model_time, 1e-9)
        print("inference on {} images: {:.2f} img/s (model only)".format(n_img, self.last_img_per_s))
        self.logger.info("==> Saving ...")
        self.save_results(results)
        return results

    def save_results(self, results):
        """One KITTI label file per image: 'Class 0.0 0 alpha x1 y1 x2 y2 h w l x y z ry score', '%.2f'."""
        output_dir = os.path.join(self.output_dir, "outputs", "data")
        os.makedirs(output_dir, exist_ok=True)
        for img_id, preds in results.items():
            with open(os.path.join(output_dir, "{:06d}.txt".format(int(img_id))), "w") as f:
                for p in preds:
                    f.write("{} 0.0 0".format(self.class_name[int(p[0])]))
                    for j in range(1, len(p)):
                        f.write(" {:.2f}".format(p[j]))
                    f.write("\n")

    def evaluate(self):
        results_dir = os.path.join(self.output_dir, "outputs", "data")
        assert os.path.exists(results_dir)
        dataset = self.dataloader.dataset
        if hasattr(dataset, "eval"):
            return dataset.eval(results_dir=results_dir, logger=self.logger)
        if getattr(dataset, "label_dir", None) and hasattr(dataset, "idx_list"):
            return evaluate_kitti_results(results_dir, dataset.label_dir, [int(i) for i in dataset.idx_list],
                                          getattr(dataset, "writelist", ["Car"]), self.logger)
        self.logger.info("dataset has no ground-truth label directory: nothing to evaluate; returning 0")
        return 0.0


def evaluate_kitti_results(results_dir, label_dir, img_ids, categories, logger):
    """What KITTI_Dataset.eval does with the written results (kitti_dataset.py:144-159): the official report per category
    through monosowa_amd.kitti_eval (HIP overlap kernels + native matching); returns Car-moderate 3D AP_R40."""
    from .. import kitti_eval as kitti
    logger.info("==> Loading detections and GTs...")
    dt_annos = kitti.get_label_annos(results_dir, img_ids)
    gt_annos = kitti.get_label_annos(label_dir, img_ids)
    test_id = {"Car": 0, "Pedestrian": 1, "Cyclist": 2}
    logger.info("==> Evaluating (official) ...")
    car_moderate = 0
    for category in categories:
        text, _, ap3d_r40 = kitti.get_official_eval_result(gt_annos, dt_annos, test_id[category])
        if category == "Car":
            car_moderate = ap3d_r40
        logger.info(text)
    return car_moderate
