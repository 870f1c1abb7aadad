"""KITTI-format file dataset: the loader side of the hot path, with the reference's batch contract.

    KITTI_Dataset(split, cfg)[i] -> (img f32 [3, 384, 1280], P2 [3, 4], targets dict of [50, ...] arrays, info dict)

follows lib/datasets/kitti/kitti_dataset.py:27-489 (directory layout `ImageSets/<split>.txt`, `training/image_2/%06d.png`,
`calib/%06d.txt` with `P2:` on line 3, 15-field `label_2/%06d.txt`; flip / crop augmentation with the reference's draw
order from `np.random`; affine resampling to 1280 x 384 with PIL; 3D-centre / l,r,t,b box encoding; 12-bin heading;
Canonical Object Space: depth labels scaled by canonical_focal_length / fu of the augmented intrinsics, :232-237) and
lib/datasets/kitti/kitti_utils.py (Object3d :13-52, Calibration :137-330, affine helpers :332-388).  The same layout is
what the converters k360_to_k.py / waymo_to_kitti_projected.py write.

`aug_pd` (photometric distortion, lib/datasets/kitti/pd.py:114-416, ON in the reference's shipped mixed-dataset config
checkpoints/best_kitti_k360_to_kitti/monodetr_kk360_05.yaml:18): monosowa_amd/photometric.py.

Not carried over (off in both shipped configs and tied to files of the pseudo-label pipeline): `use_add_data` (per-car
masks / lidar in dill+zstd), `use_depth`, `output_lidar` -- they raise.
"""
import os

import numpy as np
import torch.utils.data as data
from PIL import Image, ImageFile

from .photometric import PhotometricDistort

ImageFile.LOAD_TRUNCATED_IMAGES = True

NUM_HEADING_BIN = 12


def angle2class(angle):
    """Continuous angle -> (bin of 30 degrees, residual) (lib/datasets/utils.py:8-16)."""
    angle = angle % (2 * np.pi)
    per_class = 2 * np.pi / float(NUM_HEADING_BIN)
    shifted = (angle + per_class / 2) % (2 * np.pi)
    class_id = int(shifted / per_class)
    return class_id, shifted - (class_id * per_class + per_class / 2)


class Object3d:
    """One label line: type truncated occluded alpha x1 y1 x2 y2 h w l x y z ry [score] (kitti_utils.py:13-52)."""

    def __init__(self, line):
        f = line.strip().split(" ")
        self.src = line
        self.cls_type = f[0]
        self.trucation = float(f[1])           # (sic: the reference's attribute name)
        self.occlusion = float(f[2])
        self.alpha = float(f[3])
        self.box2d = np.array([float(v) for v in f[4:8]], dtype=np.float32)
        self.h, self.w, self.l = float(f[8]), float(f[9]), float(f[10])
        self.pos = np.array([float(v) for v in f[11:14]], dtype=np.float32)
        self.dis_to_cam = np.linalg.norm(self.pos)
        self.ry = float(f[14])
        self.score = float(f[15]) if len(f) == 16 else -1.0
        height = float(self.box2d[3]) - float(self.box2d[1]) + 1
        if self.trucation == -1:
            self.level_str, self.level = "DontCare", 0
        elif height >= 40 and self.trucation <= 0.15 and self.occlusion <= 0:
            self.level_str, self.level = "Easy", 1
        elif height >= 25 and self.trucation <= 0.3 and self.occlusion <= 1:
            self.level_str, self.level = "Moderate", 2
        elif height >= 25 and self.trucation <= 0.5 and self.occlusion <= 2:
            self.level_str, self.level = "Hard", 3
        else:
            self.level_str, self.level = "UnKnown", 4


def get_objects_from_label(label_file):
    with open(label_file, "r") as f:
        return [Object3d(line) for line in f.readlines()]


def get_calib_from_file(calib_file):
    """P2 / P3 / R0_rect / Tr_velo_to_cam from lines 3-6 of a KITTI calib file (kitti_utils.py:118-134)."""
    with open(calib_file) as f:
        lines = f.readlines()
    row = lambda i: np.array(lines[i].strip().split(" ")[1:], dtype=np.float32)
    return {"P2": row(2).reshape(3, 4), "P3": row(3).reshape(3, 4), "R0": row(4).reshape(3, 3), "Tr_velo2cam": row(5).reshape(3, 4)}


class Calibration:
    def __init__(self, calib_file):
        calib = get_calib_from_file(calib_file) if isinstance(calib_file, str) else calib_file
        self.P2, self.R0, self.V2C = calib["P2"], calib["R0"], calib["Tr_velo2cam"]
        self._intrinsics()

    def _intrinsics(self):
        self.cu, self.cv = self.P2[0, 2], self.P2[1, 2]
        self.fu, self.fv = self.P2[0, 0], self.P2[1, 1]
        self.tx, self.ty = self.P2[0, 3] / (-self.fu), self.P2[1, 3] / (-self.fv)

    def rect_to_img(self, pts_rect):
        """[N, 3] rectified camera points -> pixel coordinates [N, 2], depth [N] (kitti_utils.py:180-189)."""
        hom = np.hstack((pts_rect, np.ones((pts_rect.shape[0], 1), dtype=np.float32)))
        p = np.dot(hom, self.P2.T)
        return (p[:, 0:2].T / hom[:, 2]).T, p[:, 2] - self.P2.T[3, 2]

    def img_to_rect(self, u, v, depth_rect):
        x = ((u - self.cu) * depth_rect) / self.fu + self.tx
        y = ((v - self.cv) * depth_rect) / self.fv + self.ty
        return np.concatenate((x.reshape(-1, 1), y.reshape(-1, 1), depth_rect.reshape(-1, 1)), axis=1)

    def alpha2ry(self, alpha, u):
        ry = alpha + np.arctan2(u - self.cu, self.fu)
        return ry - 2 * np.pi if ry > np.pi else (ry + 2 * np.pi if ry < -np.pi else ry)

    def ry2alpha(self, ry, u):
        alpha = ry - np.arctan2(u - self.cu, self.fu)
        return alpha - 2 * np.pi if alpha > np.pi else (alpha + 2 * np.pi if alpha < -np.pi else alpha)

    def flip(self, img_size):
        """Refit P2 to the horizontally mirrored camera (aug_calib; kitti_utils.py:296-330): eight image points at depths
        2..78 m are back-projected, mirrored, and a pinhole with equal focal lengths is fitted by the null vector of the
        projection equations."""
        ws, hs = 4, 2
        us = np.tile(np.linspace(0, img_size[0], ws)[None, :], [hs, 1])
        vs = np.tile(np.linspace(0, img_size[1], hs)[:, None], [1, ws])
        p2 = np.stack([us, vs, np.linspace(2, 78, ws * hs).reshape(hs, ws)], -1).reshape(-1, 3)
        p3 = self.img_to_rect(p2[:, 0:1], p2[:, 1:2], p2[:, 2:3])
        p3[:, 0] *= -1
        p2[:, 0] = img_size[0] - p2[:, 0]
        eq = np.zeros([ws * hs, 2, 7])
        eq[:, 0, 0], eq[:, 1, 0] = p3[:, 0], p3[:, 1]
        eq[:, 0, 1] = eq[:, 1, 2] = p3[:, 2]
        eq[:, 0, 3] = eq[:, 1, 4] = 1
        eq[:, :, -2] = -p2[:, :2]
        eq[:, :, -1] = -p2[:, :2] * p3[:, 2:3]
        sol = np.linalg.svd(eq.reshape(-1, 7))[-1][-1]
        sol /= sol[-1]
        m = np.zeros([4, 3]).astype(np.float32)
        m[0, 0] = m[1, 1] = sol[0]
        m[2, 0:2] = sol[1:3]
        m[3, :] = sol[3:6]
        m[-1, -1] = self.P2[-1, -1]
        self.P2 = m.T
        self._intrinsics()


def _three_point_affine(src, dst):
    """The 2 x 3 matrix that maps three source points onto three destination points (the definition of
    cv2.getAffineTransform), solved in float64."""
    a = np.hstack([np.asarray(src, dtype=np.float64), np.ones((3, 1))])
    return np.linalg.solve(a, np.asarray(dst, dtype=np.float64)).T


def get_affine_transform(center, scale, rot, output_size, shift=np.array([0, 0], dtype=np.float32), inv=0):
    """Crop (centre, size) -> output image, as three point pairs like the reference (kitti_utils.py:347-381)."""
    if not isinstance(scale, (np.ndarray, list)):
        scale = np.array([scale, scale], dtype=np.float32)
    src_w, dst_w, dst_h = scale[0], output_size[0], output_size[1]
    rad = np.pi * rot / 180
    sn, cs = np.sin(rad), np.cos(rad)
    src_dir = [-(src_w * -0.5) * sn, (src_w * -0.5) * cs]
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src, dst = np.zeros((3, 2), dtype=np.float32), np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center + scale * shift
    src[1, :] = center + src_dir + scale * shift
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32) + dst_dir
    third = lambda a, b: b + np.array([-(a - b)[1], (a - b)[0]], dtype=np.float32)
    src[2, :], dst[2, :] = third(src[0, :], src[1, :]), third(dst[0, :], dst[1, :])
    trans = _three_point_affine(src, dst)
    return (trans, _three_point_affine(dst, src)) if inv else trans


def affine_transform(pt, t):
    return np.dot(t, np.array([pt[0], pt[1], 1.0], dtype=np.float32).T)[:2]


class KITTI_Dataset(data.Dataset):
    def __init__(self, split, cfg):
        self.root_dir = cfg.get("root_dir")
        self.split = split
        self.num_classes, self.max_objs = 3, 50
        self.class_name = ["Pedestrian", "Car", "Cyclist"]
        self.cls2id = {"Pedestrian": 0, "Car": 1, "Cyclist": 2}
        self.resolution = np.array([1280, 384])                      # W, H
        self.use_3d_center = cfg.get("use_3d_center", True)
        self.writelist = list(cfg.get("writelist", ["Car"]))
        self.bbox2d_type = cfg.get("bbox2d_type", "anno")
        assert self.bbox2d_type in ["anno", "proj"]
        self.meanshape = cfg.get("meanshape", False)
        if cfg.get("class_merging", False):
            self.writelist.extend(["Van", "Truck"])
        if cfg.get("use_dontcare", False):
            self.writelist.extend(["DontCare"])
        for key in ("use_add_data", "use_depth", "output_lidar"):
            if cfg.get(key, False):
                raise NotImplementedError("dataset.%s needs the pseudo-label pipeline's side files (per-car masks / lidar, depth maps) "
                                          "and is off in both shipped configs" % key)
        self.aug_pd = cfg.get("aug_pd", False)                       # on in checkpoints/.../monodetr_kk360_05.yaml:18
        self.pd = PhotometricDistort()
        assert split in ["train", "val", "trainval", "test"]
        with open(os.path.join(self.root_dir, "ImageSets", split + ".txt")) as f:
            self.idx_list = [x.strip() for x in f.readlines()]
        self.data_dir = os.path.join(self.root_dir, "testing" if split == "test" else "training")
        self.image_dir = os.path.join(self.data_dir, "image_2")
        self.calib_dir = os.path.join(self.data_dir, "calib")
        self.label_dir = os.path.join(self.data_dir, "label_2")
        self.data_augmentation = split in ["train", "trainval"]
        self.aug_crop, self.aug_calib = cfg.get("aug_crop", False), cfg.get("aug_calib", False)
        self.random_flip, self.random_crop = cfg.get("random_flip", 0.5), cfg.get("random_crop", 0.5)
        self.scale, self.shift = cfg.get("scale", 0.4), cfg.get("shift", 0.1)
        self.depth_scale = cfg.get("depth_scale", "normal")
        self.mean = np.array([0.485, 0.456, 0.406], dtype=np.float32)
        self.std = np.array([0.229, 0.224, 0.225], dtype=np.float32)
        self.cls_mean_size = np.array([[1.76255119, 0.66068622, 0.84422524], [1.52563191462, 1.62856739989, 3.88311640418],
                                       [1.73698127, 0.59706367, 1.76282397]])
        if not self.meanshape:
            self.cls_mean_size = np.zeros_like(self.cls_mean_size, dtype=np.float32)
        self.downsample = 32
        self.clip_2d = cfg.get("clip_2d", False)
        self.template_width, self.template_height = cfg.get("template_width", 1.63), cfg.get("template_height", 1.526)
        self.template_length = cfg.get("template_length", 3.88)
        self.use_canonical_module = cfg.get("use_canonical_module", False)
        self.canonical_focal_length = cfg.get("canonical_focal_length", 1000.0)

    def __len__(self):
        return len(self.idx_list)

    def get_image(self, idx):
        return Image.open(os.path.join(self.image_dir, "%06d.png" % idx))

    def get_label(self, idx):
        return get_objects_from_label(os.path.join(self.label_dir, "%06d.txt" % idx))

    def get_calib(self, idx):
        return Calibration(os.path.join(self.calib_dir, "%06d.txt" % idx))

    def eval(self, results_dir, logger):
        from .helpers.tester_helper import evaluate_kitti_results
        return evaluate_kitti_results(results_dir, self.label_dir, [int(i) for i in self.idx_list], self.writelist, logger)

    def adjust_intrinsics(self, fx, fy, cx, cy, img_size, center, crop_scale, crop_size, flipped):
        """Intrinsics of the augmented, resampled image (kitti_dataset.py:491-526) -> fx, fy, cx, cy, cy / (H / 2)."""
        if flipped:
            cx = img_size[0] - 1 - cx
        fx, fy, cx, cy = fx * crop_scale, fy * crop_scale, cx * crop_scale, cy * crop_scale
        cx, cy = cx - (center[0] - img_size[0] / 2), cy - (center[1] - img_size[1] / 2)
        s = self.resolution[0] / crop_size[0]
        fx, fy, cx, cy = fx * s, fy * s, cx * s, cy * s
        return fx, fy, cx, cy, cy / (self.resolution[1] / 2.0)

    def __getitem__(self, item):
        index = int(self.idx_list[item])
        img = self.get_image(index)
        calib = self.get_calib(index)
        img_size = np.array(img.size)
        features_size = self.resolution // self.downsample
        center = np.array(img_size) / 2
        crop_size, crop_scale = img_size, 1
        flipped = False
        if self.data_augmentation:                                   # draw order of the reference: [photometric], flip, crop?, scale, shift x, shift y
            if self.aug_pd:                                          # kitti_dataset.py:182-185
                img = Image.fromarray(self.pd(np.array(img).astype(np.float32)).astype(np.uint8))
            if np.random.random() < self.random_flip:
                flipped = True
                img = img.transpose(Image.FLIP_LEFT_RIGHT)
            if self.aug_crop and np.random.random() < self.random_crop:
                crop_scale = np.clip(np.random.randn() * self.scale + 1, 1 - self.scale, 1 + self.scale)
                crop_size = img_size * crop_scale
                center[0] += img_size[0] * np.clip(np.random.randn() * self.shift, -2 * self.shift, 2 * self.shift)
                center[1] += img_size[1] * np.clip(np.random.randn() * self.shift, -2 * self.shift, 2 * self.shift)
        trans, trans_inv = get_affine_transform(center, crop_size, 0, self.resolution, inv=1)
        img = img.transform(tuple(self.resolution.tolist()), method=Image.AFFINE, data=tuple(trans_inv.reshape(-1).tolist()),
                            resample=Image.BILINEAR)
        img = ((np.array(img).astype(np.float32) / 255.0 - self.mean) / self.std).transpose(2, 0, 1)
        fu, fv, cu, cv, height_cropped = self.adjust_intrinsics(calib.fu, calib.fv, calib.cu, calib.cv, img_size, center, crop_scale,
                                                                 crop_size, flipped)
        canonical_scale = self.canonical_focal_length / fu if self.use_canonical_module else 1.0
        info = {"img_id": index, "img_size": img_size, "bbox_downsample_ratio": img_size / features_size,
                "canonical_scale": canonical_scale, "height_crop": height_cropped}
        if self.split == "test":
            return img, calib.P2, img, info

        objects = self.get_label(index)
        if flipped:
            if self.aug_calib:
                calib.flip(img_size)
            for o in objects:
                x1, _, x2, _ = o.box2d
                o.box2d[0], o.box2d[2] = img_size[0] - x2, img_size[0] - x1
                o.alpha, o.ry = np.pi - o.alpha, np.pi - o.ry
                if self.aug_calib:
                    o.pos[0] *= -1
                wrap = lambda a: a - 2 * np.pi if a > np.pi else (a + 2 * np.pi if a < -np.pi else a)
                o.alpha, o.ry = wrap(o.alpha), wrap(o.ry)

        n = self.max_objs
        calibs = np.zeros((n, 3, 4), dtype=np.float32)
        indices = np.zeros((n,), dtype=np.int64)
        mask_2d = np.zeros((n,), dtype=bool)
        labels = np.zeros((n,), dtype=np.int8)
        depth = np.zeros((n, 1), dtype=np.float32)
        heading_bin = np.zeros((n, 1), dtype=np.int64)
        heading_res = np.zeros((n, 1), dtype=np.float32)
        size_2d = np.zeros((n, 2), dtype=np.float32)
        size_3d = np.zeros((n, 3), dtype=np.float32)
        src_size_3d = np.zeros((n, 3), dtype=np.float32)
        boxes = np.zeros((n, 4), dtype=np.float32)
        boxes_3d = np.zeros((n, 6), dtype=np.float32)
        objects_out = np.zeros((n, 7), dtype=np.float32)
        for i in range(min(len(objects), n)):
            o = objects[i]
            if o.cls_type not in self.writelist or o.level_str == "UnKnown" or o.pos[-1] < 2 or o.pos[-1] > 65:
                continue
            bbox_2d = o.box2d.copy()
            bbox_2d[:2] = affine_transform(bbox_2d[:2], trans)
            bbox_2d[2:] = affine_transform(bbox_2d[2:], trans)
            center_2d = np.array([(bbox_2d[0] + bbox_2d[2]) / 2, (bbox_2d[1] + bbox_2d[3]) / 2], dtype=np.float32)
            corner_2d = bbox_2d.copy()
            center_3d = (o.pos + [0, -o.h / 2, 0]).reshape(-1, 3)             # the box centre, not the bottom centre
            center_3d = calib.rect_to_img(center_3d)[0][0]
            if flipped and not self.aug_calib:
                center_3d[0] = img_size[0] - center_3d[0]
            center_3d = affine_transform(center_3d.reshape(-1), trans)
            if center_3d[0] < 0 or center_3d[0] >= self.resolution[0] or center_3d[1] < 0 or center_3d[1] >= self.resolution[1]:
                continue
            labels[i] = self.cls2id[o.cls_type]
            size_2d[i] = bbox_2d[2] - bbox_2d[0], bbox_2d[3] - bbox_2d[1]
            center_2d_norm = center_2d / self.resolution
            size_2d_norm = size_2d[i] / self.resolution
            corner_2d[0:2] = corner_2d[0:2] / self.resolution
            corner_2d[2:4] = corner_2d[2:4] / self.resolution
            center_3d_norm = center_3d / self.resolution
            l, r = center_3d_norm[0] - corner_2d[0], corner_2d[2] - center_3d_norm[0]
            t, b = center_3d_norm[1] - corner_2d[1], corner_2d[3] - center_3d_norm[1]
            if l < 0 or r < 0 or t < 0 or b < 0:
                if not self.clip_2d:
                    continue
                l, r, t, b = (np.clip(v, 0, 1) for v in (l, r, t, b))
            boxes[i] = center_2d_norm[0], center_2d_norm[1], size_2d_norm[0], size_2d_norm[1]
            boxes_3d[i] = center_3d_norm[0], center_3d_norm[1], l, r, t, b
            if self.use_canonical_module:
                o.pos[-1] *= canonical_scale                                  # Canonical Object Space: depth as seen by the canonical camera
            depth[i] = {"normal": o.pos[-1] * crop_scale, "inverse": o.pos[-1] / crop_scale, "none": o.pos[-1]}[self.depth_scale]
            heading = calib.ry2alpha(o.ry, (o.box2d[0] + o.box2d[2]) / 2)
            heading = heading - 2 * np.pi if heading > np.pi else (heading + 2 * np.pi if heading < -np.pi else heading)
            heading_bin[i], heading_res[i] = angle2class(heading)
            src_size_3d[i] = np.array([o.h, o.w, o.l], dtype=np.float32)
            size_3d[i] = src_size_3d[i] - self.cls_mean_size[self.cls2id[o.cls_type]]
            if o.trucation <= 0.5 and o.occlusion <= 2:
                mask_2d[i] = 1
            calibs[i] = calib.P2
            objects_out[i] = np.array([o.h, o.w, o.l, o.pos[0], o.pos[1], o.pos[2], o.ry], dtype=np.float32)
        targets = {"calibs": calibs, "indices": indices, "img_size": img_size, "labels": labels, "boxes": boxes, "boxes_3d": boxes_3d,
                   "depth": depth, "size_2d": size_2d, "size_3d": size_3d, "src_size_3d": src_size_3d, "heading_bin": heading_bin,
                   "heading_res": heading_res, "mask_2d": mask_2d, "objects": objects_out}
        info.update({"affine": trans, "affine_inv": trans_inv, "scale_depth": crop_scale, "calib_P2": calib.P2, "calib_R0": calib.R0,
                     "calib_V2C": calib.V2C, "resolution": self.resolution, "flip": flipped,
                     "templates_dimensions": np.array([self.template_height, self.template_width, self.template_length], dtype=np.float32)})
        return img, calib.P2, targets, info
