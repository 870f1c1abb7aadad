"""KITTI evaluation against fixtures made by the REFERENCE's own evaluation code
(oracle/gen_golden.py kitti_eval: kitti_eval_python/eval.py run as plain Python under an identity `numba.jit`, its
rotated-IoU DEVICE function, rotate_iou.py:17-259, called pair by pair on float32 numpy arrays)."""
import logging
import os

import numpy as np
import pytest

from oracle import rotate_iou_oracle as RO

KEYS = ("name", "truncated", "occluded", "alpha", "bbox", "dimensions", "location", "rotation_y", "score")


def _annos(g, prefix):
    out, o = [], 0
    for c in g[prefix + "_count"]:
        out.append({k: g[prefix + "_" + k][o:o + c] for k in KEYS})
        o += c
    return out


def _overlaps(g, metric, dts, gts):
    flat, out, o = g["m%d_overlaps" % metric], [], 0
    for d, t in zip(dts, gts):
        n = len(d["name"]) * len(t["name"])
        out.append(flat[o:o + n].reshape(len(d["name"]), len(t["name"])))
        o += n
    return out


@pytest.fixture()
def ap(golden_dir):
    g = np.load(os.path.join(golden_dir, "kitti_ap.npz"), allow_pickle=False)
    return g, _annos(g, "gt"), _annos(g, "dt")


def _exact_rotated_overlaps(monkeypatch):
    """bev / 3D overlaps from the exact float64 oracle instead of the HIP kernels (CPU tests)."""
    from monosowa_amd import kitti_eval as K
    monkeypatch.setattr(K, "bev_box_overlap", lambda b, q, criterion=-1: RO.rotate_iou(b, q, criterion))
    monkeypatch.setattr(K, "d3_box_overlap", lambda b, q, criterion=-1, device_id=0: RO.box3d_overlap(b, q, criterion))
    return K


def test_recall_thresholds_equal_the_reference(ap):
    from monosowa_amd import kitti_eval as K
    g = ap[0]
    assert np.array_equal(np.array(K.get_thresholds(g["thr_scores"], int(g["thr_num_gt"]))), g["thr_out"])
    assert K.get_thresholds(np.zeros(0), 5) == []


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_matching_and_pr_curves_equal_the_reference_bit_for_bit(ap, metric):
    """clean_data + the native matcher (mono_kitti_tp_scores_f64 / mono_kitti_pr_f64) + the recall-threshold and envelope
    steps, fed with the overlap matrices the reference used: precision / recall / orientation [3 classes, 3 difficulties,
    2 overlap settings, 41 points], NaNs included."""
    from monosowa_amd import kitti_eval as K
    g, gts, dts = ap
    ret = K.eval_class(gts, dts, [0, 1, 2], [0, 1, 2], metric, g["min_overlaps"], compute_aos=(metric == 0),
                       overlaps=_overlaps(g, metric, dts, gts))
    for key in ("precision", "recall") + (("orientation",) if metric == 0 else ()):
        assert np.array_equal(ret[key], g["m%d_%s" % (metric, key)], equal_nan=True), key
    assert np.nanmax(ret["precision"]) > 0.3          # the fixture does exercise true positives


def test_image_box_overlaps_equal_the_reference(ap):
    from monosowa_amd import kitti_eval as K
    g, gts, dts = ap
    mine = K.calculate_iou_partly(dts, gts, 0)[0]
    assert np.array_equal(np.concatenate([o.reshape(-1) for o in mine]), g["m0_overlaps"])


def test_official_report_equals_the_reference_text(ap, monkeypatch):
    """get_official_eval_result for Car / Pedestrian / Cyclist: the report text character for character, every dictionary
    entry, the returned Car-moderate AP_R40 (rotated overlaps from the exact oracle here; from the HIP kernels in the GPU test)."""
    g, gts, dts = ap
    K = _exact_rotated_overlaps(monkeypatch)
    _check_official(K, g, gts, dts)


def test_distance_range_report_equals_the_reference_text(ap, monkeypatch):
    """get_distance_eval_result (clean_data_by_distance: ranges 0-30 / 30-50 / 50-70 m) for Car and Pedestrian."""
    g, gts, dts = ap
    K = _exact_rotated_overlaps(monkeypatch)
    texts = []
    for cls in (0, 1):
        text, ret = K.get_distance_eval_result(gts, dts, cls)
        texts.append(text)
        assert set(ret) == {k.split("__")[1] for k in g.files if k.startswith("distance_%d__" % cls)}
        for key, val in ret.items():
            ref = float(g["distance_%d__%s" % (cls, key)])
            assert abs(val - ref) <= 1e-9 or (np.isnan(val) and np.isnan(ref)), (cls, key, val, ref)
    assert "\n=====\n".join(texts) == str(g["distance_text"])


def _check_official(K, g, gts, dts):
    texts = []
    for cls in (0, 1, 2):
        text, ret, car = K.get_official_eval_result(gts, dts, cls)
        texts.append(text)
        assert set(ret) == {k.split("__")[1] for k in g.files if k.startswith("official_%d__" % cls)}
        for key, val in ret.items():
            ref = float(g["official_%d__%s" % (cls, key)])
            assert abs(val - ref) <= 1e-9 or (np.isnan(val) and np.isnan(ref)), (cls, key, val, ref)
        assert abs(car - float(g["official_%d_return" % cls])) <= 1e-9
    assert "\n=====\n".join(texts) == str(g["official_text"])


def test_exact_oracle_agrees_with_the_reference_device_function(golden_dir):
    """oracle/rotate_iou_oracle.py (Sutherland-Hodgman, float64) against the reference's rotate_iou device function on every
    non-degenerate pair.  EXACTLY identical boxes are excluded: there the reference's vertex collection depends on float32
    rounding of on-edge tests and returns 0 or 1/3 instead of 1 in this emulation (rotate_iou.py:161-200)."""
    r = np.load(os.path.join(golden_dir, "kitti_rotate_iou.npz"))
    same = (r["boxes"][:, None, :] == r["qboxes"][None, :, :]).all(-1)
    for c in (-1, 0, 1, 2):
        ref = r["iou_crit%d" % c]
        diff = np.abs(RO.rotate_iou(r["boxes"], r["qboxes"], c) - ref)
        assert diff[~same].max() <= 3e-6 * max(1.0, ref.max()), (c, diff[~same].max())          # criterion 2 returns areas (up to ~9)
    assert same.sum() == 8 and (r["iou_crit-1"] > 0.05).sum() > 150


def test_written_results_round_trip_through_the_label_reader(ap, tmp_path, monkeypatch):
    """Result files as tester_helper.save_results writes them ('%.2f') and label files, read back by get_label_annos and
    evaluated through the tester's entry point."""
    from monosowa_amd.helpers.tester_helper import evaluate_kitti_results
    g, gts, dts = ap
    _exact_rotated_overlaps(monkeypatch)
    for sub, annos, with_score in (("label_2", gts, False), ("data", dts, True)):
        os.makedirs(tmp_path / sub)
        for i, a in enumerate(annos):
            with open(tmp_path / sub / ("%06d.txt" % i), "w") as f:
                for j in range(len(a["name"])):
                    l, h, w = a["dimensions"][j]
                    vals = [a["alpha"][j], *a["bbox"][j], h, w, l, *a["location"][j], a["rotation_y"][j]] + ([a["score"][j]] if with_score else [])
                    f.write("%s %.2f %d " % (a["name"][j], a["truncated"][j], a["occluded"][j]) + " ".join("%.2f" % v for v in vals) + "\n")
    from monosowa_amd import kitti_eval as K
    back = K.get_label_annos(str(tmp_path / "data"))
    assert len(back) == len(dts) and all(len(b["name"]) == len(d["name"]) for b, d in zip(back, dts))
    assert np.allclose(back[3]["dimensions"], np.round(dts[3]["dimensions"], 2), atol=6e-3) and back[3]["score"].shape == dts[3]["score"].shape
    car = evaluate_kitti_results(str(tmp_path / "data"), str(tmp_path / "label_2"), list(range(len(gts))), ["Car", "Pedestrian"],
                                 logging.getLogger("kitti-ap-test"))
    assert 0.0 <= car <= 100.0


@pytest.mark.gpu
@pytest.mark.parametrize("criterion", [-1, 0, 1, 2])
def test_rotate_iou_kernel_equals_the_reference_device_function(golden_dir, criterion):
    """The HIP kernel behind rotate_iou_gpu_eval against the reference's own device function (non-degenerate pairs)."""
    from monosowa_amd import kitti_eval as K
    r = np.load(os.path.join(golden_dir, "kitti_rotate_iou.npz"))
    same = (r["boxes"][:, None, :] == r["qboxes"][None, :, :]).all(-1)
    got = K.rotate_iou_gpu_eval(r["boxes"], r["qboxes"], criterion)
    ref = r["iou_crit%d" % criterion]
    assert np.abs(got - ref)[~same].max() <= 2e-5 * max(1.0, ref.max()), np.abs(got - ref)[~same].max()


@pytest.mark.gpu
def test_official_report_on_the_gpu_equals_the_reference_text(ap):
    from monosowa_amd import kitti_eval as K
    g, gts, dts = ap
    _check_official(K, g, gts, dts)
