"""Synthetic KITTI-shaped data honouring the reference dataloader's batch contract
``(inputs f32[B,3,H,W], calibs f32[B,3,4], targets{... [B,50,.]}, info{...})``
(lib/datasets/kitti/kitti_dataset.py:271-283,397-412,465-489) -- SURVEY.md section 8d, config 2:
images ~ N(0,1); P2 with fu = fv = 707.05 * (1280/1242), cu = 640, cv = 192; img_size (1242, 375);
per image n ~ U{1..10} Car objects, boxes_3d = (cx, cy, l, r, t, b) with cx, cy ~ U(0.1, 0.9) and
l, r, t, b ~ U(0.01, 0.1); boxes = the matching cxcywh; depth ~ U(5, 60);
size_3d ~ N((1.53, 1.63, 3.88), 0.1); heading_bin ~ U{0..11}; heading_res ~ U(-pi/12, pi/12);
labels = 1 (Car); padded to 50 slots with mask_2d.  These values are chosen for the synthetic
benchmark, not taken from the reference."""
import math

import numpy as np
import torch
from torch.utils.data import Dataset

MAX_OBJS = 50


MIXED_CAMERA_FU = (721.5, 552.6, 2055.0)      # KITTI, KITTI-360, Waymo front camera: typical values picked for the synthetic
                                              # mixed-dataset batch of SURVEY.md 8d config 5 (the reference reads fu per calib file)


def synthetic_sample(rng, resolution=(1280, 384), canonical_focal_length=500.0, fu=None):
    """``fu``: focal length of this sample's camera in pixels (default: the KITTI-like value scaled to the resolution).  The
    depth labels live in Canonical Object Space: scaled by canonical_focal_length / fu PER SAMPLE (kitti_dataset.py:232-237)."""
    W, H = resolution
    img = rng.standard_normal((3, H, W), dtype=np.float32)
    fu = 707.05 * (W / 1242.0) if fu is None else float(fu)
    calib = np.array([[fu, 0, W / 2.0, 0], [0, fu, H / 2.0, 0], [0, 0, 1, 0]], dtype=np.float32)
    n = int(rng.integers(1, 11))
    t = {
        "calibs": np.zeros((MAX_OBJS, 3, 4), dtype=np.float32),      # P2 per object slot (kitti_dataset.py:271,392)
        "indices": np.zeros((MAX_OBJS,), dtype=np.int64),
        "img_size": np.array([1242, 375], dtype=np.int32),
        "labels": np.zeros((MAX_OBJS,), dtype=np.int8),
        "boxes": np.zeros((MAX_OBJS, 4), dtype=np.float32),
        "boxes_3d": np.zeros((MAX_OBJS, 6), dtype=np.float32),
        "depth": np.zeros((MAX_OBJS, 1), dtype=np.float32),
        "size_2d": np.zeros((MAX_OBJS, 2), dtype=np.float32),
        "size_3d": np.zeros((MAX_OBJS, 3), dtype=np.float32),
        "src_size_3d": np.zeros((MAX_OBJS, 3), dtype=np.float32),
        "heading_bin": np.zeros((MAX_OBJS, 1), dtype=np.int64),
        "heading_res": np.zeros((MAX_OBJS, 1), dtype=np.float32),
        "mask_2d": np.zeros((MAX_OBJS,), dtype=bool),
    }
    c = rng.uniform(0.1, 0.9, (n, 2)).astype(np.float32)
    lrtb = rng.uniform(0.01, 0.1, (n, 4)).astype(np.float32)
    t["boxes_3d"][:n] = np.concatenate([c, lrtb], 1)
    x0, x1 = c[:, 0] - lrtb[:, 0], c[:, 0] + lrtb[:, 1]
    y0, y1 = c[:, 1] - lrtb[:, 2], c[:, 1] + lrtb[:, 3]
    t["boxes"][:n] = np.stack([(x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0], 1)
    t["size_2d"][:n] = t["boxes"][:n, 2:] * np.array([W, H], dtype=np.float32)
    canonical_scale = canonical_focal_length / fu          # Canonical Object Space (kitti_dataset.py:232-237)
    t["depth"][:n, 0] = rng.uniform(5, 60, n).astype(np.float32) * canonical_scale
    t["size_3d"][:n] = (np.array([1.53, 1.63, 3.88]) + 0.1 * rng.standard_normal((n, 3))).astype(np.float32)
    t["src_size_3d"][:n] = t["size_3d"][:n]
    t["heading_bin"][:n, 0] = rng.integers(0, 12, n)
    t["heading_res"][:n, 0] = rng.uniform(-math.pi / 12, math.pi / 12, n).astype(np.float32)
    t["labels"][:n] = 1
    t["calibs"][:n] = calib
    t["mask_2d"][:n] = True
    t["indices"][:n] = np.arange(n)
    info = {"img_id": 0, "img_size": np.array([1242, 375], dtype=np.int32),
            "bbox_downsample_ratio": np.array([1242 / (W / 16), 375 / (H / 16)], dtype=np.float32),
            "height_crop": np.float32(1.0), "canonical_scale": np.float32(canonical_scale)}
    return img, calib, t, info


class SyntheticKITTI(Dataset):
    """Index-seeded synthetic dataset with the KITTI_Dataset item contract (img, calib, targets, info)."""

    def __init__(self, split="train", cfg=None, num_samples=None, seed=444):
        cfg = cfg or {}
        self.split = split
        self.resolution = tuple(cfg.get("resolution", (1280, 384)))
        self.num_samples = int(num_samples if num_samples is not None else cfg.get("num_samples", 64))
        self.canonical_focal_length = float(cfg.get("canonical_focal_length", 500.0))
        self.seed = seed + int(cfg.get("seed_offset", 0)) + (0 if split == "train" else 100003)
        self.max_objs = MAX_OBJS
        self.class_name = ["Pedestrian", "Car", "Cyclist"]
        self.cls_mean_size = np.zeros((3, 3), dtype=np.float32)   # meanshape: False

    def __len__(self):
        return self.num_samples

    def __getitem__(self, i):
        rng = np.random.default_rng(self.seed * 1000003 + i)
        img, calib, t, info = synthetic_sample(rng, self.resolution, self.canonical_focal_length)
        info["img_id"] = i
        return img, calib, t, info


def make_batch(batch_size, device, seed=444, resolution=(1280, 384), mixed_cameras=False):
    """One pre-collated batch resident on ``device``: (inputs, calibs, targets dict of [B,50,...], info).
    ``mixed_cameras``: the samples cycle through MIXED_CAMERA_FU (a mixed-dataset batch: every sample has its own
    canonical_scale and its own fu in ``calibs``, which the depth head reads per sample, monodetr.py:248)."""
    rng = np.random.default_rng(seed)
    samples = [synthetic_sample(rng, resolution, fu=MIXED_CAMERA_FU[i % len(MIXED_CAMERA_FU)] if mixed_cameras else None)
               for i in range(batch_size)]
    inputs = torch.from_numpy(np.stack([s[0] for s in samples])).to(device)
    calibs = torch.from_numpy(np.stack([s[1] for s in samples])).to(device)
    targets = {k: torch.from_numpy(np.stack([s[2][k] for s in samples])).to(device) for k in samples[0][2]}
    attach_host_mask(targets["mask_2d"], np.stack([s[2]["mask_2d"] for s in samples]))
    info = {k: np.stack([np.asarray(s[3][k]) for s in samples]) for k in samples[0][3]}
    return inputs, calibs, targets, info


USE_HOST_MASK = True


def attach_host_mask(mask_device, mask_host):
    """The loader knows the object mask on the host before it ships the batch; keeping that copy next to the device
    tensor lets ``prepare_targets`` build its gather indices without a device->host sync."""
    mask_device._host_mask = np.asarray(mask_host).astype(bool)
    return mask_device


def prepare_targets(targets, batch_size):
    """Padded [B,50,...] dict -> list of per-image dicts of the valid objects (trainer_helper.py:180-191).
    The reference indexes every key of every image with a boolean mask (8 x B device->host syncs); here the
    mask is resolved once and each key is gathered once for the whole batch, then split into views."""
    keys = ("labels", "boxes", "calibs", "depth", "size_3d", "heading_bin", "heading_res", "boxes_3d")
    host = getattr(targets["mask_2d"], "_host_mask", None)
    if host is not None and USE_HOST_MASK:
        hb, hs = np.nonzero(host[:batch_size])                     # row-major = per image, slot order; no device sync
        counts = np.bincount(hb, minlength=batch_size).tolist()
        dev = targets["mask_2d"].device
        b_idx = torch.from_numpy(hb).to(dev, non_blocking=True)
        s_idx = torch.from_numpy(hs).to(dev, non_blocking=True)
    else:
        mask = targets["mask_2d"][:batch_size]
        b_idx, s_idx = mask.nonzero(as_tuple=True)                 # one sync
        counts = torch.bincount(b_idx, minlength=batch_size).tolist() if b_idx.numel() else [0] * batch_size
    flat = {k: v[b_idx, s_idx] for k, v in targets.items() if k in keys}
    per_key = {k: v.split(counts) for k, v in flat.items()}
    out = TargetList({k: per_key[k][b] for k in per_key} for b in range(batch_size))
    out.flat = flat                 # the per-image entries are views of these: the criterion need not concatenate them again
    return out


class TargetList(list):
    """The reference's list of per-image target dicts, remembering the batch-flat tensors its entries are views of."""
    flat = None
