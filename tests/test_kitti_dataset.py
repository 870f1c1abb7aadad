"""KITTI file dataset against the REFERENCE's KITTI_Dataset (oracle/gen_golden.py kitti_dataset: the
reference class on a small synthetic KITTI directory whose files travel inside the fixture)."""
import json
import os

import numpy as np
import pytest
import torch


@pytest.fixture()
def kitti_root(golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, "kitti_dataset.npz"), allow_pickle=False)
    for i, name in enumerate(g["file_names"]):
        path = tmp_path / str(name)
        os.makedirs(path.parent, exist_ok=True)
        path.write_bytes(g["file_%03d" % i].tobytes())
    cfg = dict(json.loads(str(g["cfg_json"])), root_dir=str(tmp_path))
    return g, cfg


@pytest.mark.parametrize("split,seeds", [("val", [0]), ("train", [11, 12, 13])])
def test_samples_equal_the_reference_dataset(kitti_root, split, seeds):
    """Every target array bit for bit (boxes, 3D-centre / l,r,t,b encoding, canonical-object-space depth, heading bins,
    masks, which objects survive the filters), the affine transforms and intrinsics bookkeeping, and the resampled image
    (an 8x8-strided sample of it exactly, its sum to 1e-9) -- also under flip + crop augmentation with fixed numpy seeds."""
    from monosowa_amd.kitti_dataset import KITTI_Dataset
    g, cfg = kitti_root
    ds = KITTI_Dataset(split, cfg)
    assert len(ds) == len(g["ids"])
    flips = kept = 0
    for seed in seeds:
        for item in range(len(ds)):
            np.random.seed(seed * 100 + item)
            img, P2, targets, info = ds[item]
            key = "%s_s%d_i%d__" % (split, seed, item)
            assert img.dtype == np.float32 and img.shape == (3, 384, 1280)
            assert np.array_equal(img[:, ::8, ::8], g[key + "img_sub"])
            assert abs(img.astype(np.float64).sum() - float(g[key + "img_sum"])) <= 1e-9 * max(1.0, abs(float(g[key + "img_sum"])))
            assert np.array_equal(np.asarray(P2), g[key + "P2"])
            assert set(targets) == {k[len(key) + 2:] for k in g.files if k.startswith(key + "t_")}
            for k, v in targets.items():
                ref = g[key + "t_" + k]
                assert np.asarray(v).dtype == ref.dtype and np.array_equal(np.asarray(v), ref), (key, k)
            for k in ("img_id", "img_size", "bbox_downsample_ratio", "canonical_scale", "height_crop", "affine", "affine_inv", "scale_depth", "flip"):
                assert np.array_equal(np.asarray(info[k]), g[key + "info_" + k]), (key, k)
            flips += int(info["flip"])
            kept += int(targets["mask_2d"].sum())
    assert kept > 0 and (split == "val" or 0 < flips < len(seeds) * len(ds))       # the fixture exercises both branches


def test_loader_batches_feed_the_training_step_contract(kitti_root):
    """build_dataloader(type KITTI) -> the (inputs, calibs, targets, info) batch the trainer consumes; prepare_targets accepts it."""
    from monosowa_amd.helpers.dataloader_helper import build_dataloader
    from monosowa_amd.synthetic import prepare_targets
    g, cfg = kitti_root
    cfg = dict(cfg, type="KITTI", train_split="train", test_split="val", batch_size=3)
    train_loader, test_loader = build_dataloader(cfg, workers=0)
    inputs, calibs, targets, info = next(iter(test_loader))
    assert inputs.shape == (3, 3, 384, 1280) and inputs.dtype == torch.float32 and calibs.shape == (3, 3, 4)
    assert targets["boxes_3d"].shape == (3, 50, 6) and targets["labels"].dtype == torch.int8 and targets["mask_2d"].dtype == torch.bool
    tl = prepare_targets(targets, 3)
    assert len(tl) == 3 and all(set(t) >= {"labels", "boxes", "boxes_3d", "depth", "size_3d", "heading_bin", "heading_res"} for t in tl)
    assert sum(len(t["labels"]) for t in tl) == int(targets["mask_2d"].sum())
    assert len(train_loader.dataset) == len(g["ids"])


def test_photometric_distortion_equals_the_reference_class(golden_dir):
    """monosowa_amd.photometric.PhotometricDistort against the reference class (pd.py:398-416, run by oracle/gen_golden.py
    kitti_dataset_pd with cv2.cvtColor restated from OpenCV's documented float formulas) on a seeded image under 24 numpy
    seeds: same draws in the same order, same floats."""
    from monosowa_amd.photometric import PhotometricDistort
    g = np.load(os.path.join(golden_dir, "kitti_dataset_pd.npz"), allow_pickle=False)
    pd = PhotometricDistort()
    image = g["pd_image"]
    changed = permuted = 0
    for seed in range(24):
        np.random.seed(1000 + seed)
        out = pd(image)
        ref = g["pd_out_%02d" % seed]
        assert out.dtype == np.float32 and out.shape == ref.shape
        assert np.array_equal(out, ref), (seed, float(np.abs(out - ref).max()))
        changed += int(not np.array_equal(ref, image))
        permuted += int(np.abs(ref[..., 0] - image[..., 0]).mean() > 40)
    assert changed >= 20 and permuted >= 1 and np.array_equal(image, g["pd_image"])      # branches taken; the input is not modified


def test_shipped_mixed_dataset_config_constructs_and_matches_the_reference(kitti_root, golden_dir):
    """The dataset section of the reference's shipped mixed-dataset config (checkpoints/best_kitti_k360_to_kitti/
    monodetr_kk360_05.yaml: aug_pd and aug_crop on, canonical focal length 1000) builds a KITTI_Dataset here, and its train
    samples equal the reference class's: image (strided sample exactly, sum to 1e-9), intrinsics, box / depth targets."""
    from monosowa_amd.kitti_dataset import KITTI_Dataset
    _, base = kitti_root
    g = np.load(os.path.join(golden_dir, "kitti_dataset_pd.npz"), allow_pickle=False)
    cfg = dict(json.loads(str(g["cfg_json"])), root_dir=base["root_dir"])
    assert cfg["aug_pd"] is True and cfg["aug_crop"] is True
    ds = KITTI_Dataset("train", cfg)
    n = 0
    for seed in (21, 22):
        for item in range(len(ds)):
            np.random.seed(seed * 100 + item)
            img, P2, targets, info = ds[item]
            key = "train_s%d_i%d__" % (seed, item)
            assert np.array_equal(img[:, ::8, ::8], g[key + "img_sub"]), key
            assert abs(img.astype(np.float64).sum() - float(g[key + "img_sum"])) <= 1e-9 * max(1.0, abs(float(g[key + "img_sum"])))
            assert np.array_equal(np.asarray(P2), g[key + "P2"])
            for k in ("boxes_3d", "depth", "mask_2d", "labels"):
                assert np.array_equal(np.asarray(targets[k]), g[key + "t_" + k]), (key, k)
            assert np.array_equal(np.asarray(info["flip"]), g[key + "info_flip"])
            n += 1
    assert n == 2 * len(ds)


def test_unshipped_side_inputs_are_refused(kitti_root):
    from monosowa_amd.kitti_dataset import KITTI_Dataset
    _, cfg = kitti_root
    for key in ("use_add_data", "use_depth", "output_lidar"):
        with pytest.raises(NotImplementedError):
            KITTI_Dataset("val", dict(cfg, **{key: True}))
