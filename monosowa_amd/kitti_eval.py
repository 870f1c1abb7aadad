"""Rotated-box overlaps of the KITTI evaluation on MI355X, with the reference's function names and
numpy-in / numpy-out contracts so that `kitti_eval_python/eval.py` can call them in place of its numba-CUDA kernel:

    rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1, device_id=0)    rotate_iou.py:293-330
    bev_box_overlap(boxes, qboxes, criterion=-1)                          eval.py:192-194
    d3_box_overlap(boxes, qboxes, criterion=-1)                           eval.py:226-230

and the AP accumulation on top of them (same names, arguments and results as the reference's numba code):

    get_label_anno / get_label_annos                                      kitti_common.py:294-347
    clean_data, get_thresholds, calculate_iou_partly, eval_class          eval.py:10-80, :413-640
    get_mAP, get_mAP_R40, do_eval, get_official_eval_result               eval.py:643-700, :863-985

The per-image matching runs in C++ (csrc/kitti_ap.h: mono_kitti_tp_scores_f64 / mono_kitti_pr_f64), the rotated overlaps on
the GPU; everything else is a few numpy expressions per (class, difficulty).
"""
import ctypes
import os

import numpy as np
import torch

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libmonosowa_kitti.so")
SYMBOLS = ("mono_rotate_iou_f32", "mono_box3d_overlap_f32", "mono_extract_dets_f32", "mono_kitti_tp_scores_f64", "mono_kitti_pr_f64")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise RuntimeError("HIP extension %s is missing: run `python -m monosowa_amd.build`" % _PATH)
        lib = ctypes.CDLL(_PATH)
        P, I, LL = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
        for name in SYMBOLS[:2]:
            fn = getattr(lib, name)
            fn.restype = I
            fn.argtypes = [P, P, P, LL, LL, I, P]
        lib.mono_extract_dets_f32.restype = I
        lib.mono_extract_dets_f32.argtypes = [P] * 6 + [I] * 4 + [P]
        D = ctypes.c_double
        lib.mono_kitti_tp_scores_f64.restype = I
        lib.mono_kitti_tp_scores_f64.argtypes = [LL] + [P] * 9 + [I, D, P, P]
        lib.mono_kitti_pr_f64.restype = I
        lib.mono_kitti_pr_f64.argtypes = [LL] + [P] * 9 + [I, D, P, LL, I, P]
        _lib = lib
    return _lib


def _overlap(fn_name, boxes, query, width, criterion, device_id):
    dtype = boxes.dtype
    N, K = boxes.shape[0], query.shape[0]
    if N == 0 or K == 0:
        return np.zeros((N, K), dtype=dtype)
    if not torch.cuda.is_available():
        raise RuntimeError("%s needs the GPU (the reference's kernel is GPU-only as well)" % fn_name)
    dev = torch.device("cuda", device_id)
    b = torch.from_numpy(np.ascontiguousarray(boxes, dtype=np.float32).reshape(N, width)).to(dev)
    q = torch.from_numpy(np.ascontiguousarray(query, dtype=np.float32).reshape(K, width)).to(dev)
    out = torch.empty((N, K), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        code = getattr(load(), fn_name)(b.data_ptr(), q.data_ptr(), out.data_ptr(), N, K, int(criterion),
                                        torch.cuda.current_stream().cuda_stream)
    if code:
        raise RuntimeError("%s failed with code %d" % (fn_name, code))
    return out.cpu().numpy().astype(dtype)


def rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1, device_id=0):
    """boxes [N,5], query_boxes [K,5] = (cx, cy, w, h, angle) -> overlap matrix [N,K] in boxes.dtype."""
    return _overlap("mono_rotate_iou_f32", boxes, query_boxes, 5, criterion, device_id)


def bev_box_overlap(boxes, qboxes, criterion=-1):
    return rotate_iou_gpu_eval(boxes, qboxes, criterion)


def d3_box_overlap(boxes, qboxes, criterion=-1, device_id=0):
    """Camera-frame 3D boxes [*,7] -> 3D overlap [N,K] (BEV intersection x height overlap, one kernel)."""
    return _overlap("mono_box3d_overlap_f32", boxes, qboxes, 7, criterion, device_id)


def extract_dets_device(outputs, topk=50):
    """decode_helper.py:58-111 in one launch: outputs dict of float32 CUDA tensors -> detections [B, topk, 37]."""
    logits = outputs["pred_logits"].contiguous()
    B, Q, C = logits.shape
    t = lambda k: outputs[k].contiguous()
    out = torch.empty((B, topk, 37), dtype=torch.float32, device=logits.device)
    with torch.cuda.device(logits.device):
        code = load().mono_extract_dets_f32(logits.data_ptr(), t("pred_boxes").data_ptr(), t("pred_angle").data_ptr(),
                                            t("pred_3d_dim").data_ptr(), t("pred_depth").data_ptr(), out.data_ptr(), B, Q, C,
                                            int(topk), torch.cuda.current_stream().cuda_stream)
    if code:
        raise RuntimeError("mono_extract_dets_f32 failed with code %d" % code)
    return out


# ================================================================================================ AP accumulation
import pathlib
import re

CLASS_NAMES = ("car", "pedestrian", "cyclist", "van", "person_sitting", "truck")
CLASS_TO_NAME = {0: "Car", 1: "Pedestrian", 2: "Cyclist", 3: "Van", 4: "Person_sitting", 5: "Truck"}
MIN_HEIGHT = (40, 25, 25)            # per difficulty (easy, moderate, hard): eval.py:31-33
MAX_OCCLUSION = (0, 1, 2)
MAX_TRUNCATION = (0.15, 0.3, 0.5)
N_SAMPLE_PTS = 41


def get_label_anno(label_path):
    """One KITTI label / result file -> dict of arrays (kitti_common.py:294-330); dimensions come back as (l, h, w)."""
    with open(label_path, "r") as f:
        rows = [line.strip().split(" ") for line in f.readlines()]
    num = lambda lo, hi: np.array([[float(v) for v in r[lo:hi]] for r in rows], dtype=np.float64).reshape(-1, hi - lo)
    anno = {"name": np.array([r[0] for r in rows]), "truncated": num(1, 2).reshape(-1),
            "occluded": np.array([int(r[2]) for r in rows]), "alpha": num(3, 4).reshape(-1), "bbox": num(4, 8),
            "dimensions": num(8, 11)[:, [2, 0, 1]], "location": num(11, 14), "rotation_y": num(14, 15).reshape(-1)}
    anno["score"] = num(15, 16).reshape(-1) if rows and len(rows[0]) == 16 else np.zeros(len(rows))
    return anno


def get_label_annos(label_folder, image_ids=None):
    folder = pathlib.Path(label_folder)
    if image_ids is None:
        image_ids = sorted(int(p.stem) for p in folder.glob("*.txt") if re.match(r"^\d{6}.txt$", p.name))
    if not isinstance(image_ids, list):
        image_ids = list(range(image_ids))
    return [get_label_anno(folder / ("%06d.txt" % idx)) for idx in image_ids]


def clean_data(gt_anno, dt_anno, current_class, difficulty):
    """Which boxes count for (class, difficulty): -> num_valid_gt, ignored_gt, ignored_dt (0 evaluate / 1 ignore / -1 other
    class, int64 arrays), DontCare boxes [n, 4] (eval.py:29-80)."""
    cls = CLASS_NAMES[current_class]
    gname = np.char.lower(np.asarray(gt_anno["name"], dtype=str)) if len(gt_anno["name"]) else np.zeros(0, dtype=str)
    same = gname == cls
    neighbour = ((gname == "person_sitting") & (cls == "pedestrian")) | ((gname == "van") & (cls == "car"))
    bbox = np.asarray(gt_anno["bbox"], dtype=np.float64).reshape(-1, 4)
    hard = (np.asarray(gt_anno["occluded"]) > MAX_OCCLUSION[difficulty]) | (np.asarray(gt_anno["truncated"]) > MAX_TRUNCATION[difficulty]) \
        | ((bbox[:, 3] - bbox[:, 1]) <= MIN_HEIGHT[difficulty])
    ignored_gt = np.where(same & ~hard, 0, np.where(neighbour | (same & hard), 1, -1)).astype(np.int64)
    dc = bbox[np.asarray(gt_anno["name"], dtype=str) == "DontCare"] if len(bbox) else np.zeros((0, 4))
    dname = np.char.lower(np.asarray(dt_anno["name"], dtype=str)) if len(dt_anno["name"]) else np.zeros(0, dtype=str)
    dbox = np.asarray(dt_anno["bbox"], dtype=np.float64).reshape(-1, 4)
    small = np.abs(dbox[:, 3] - dbox[:, 1]) < MIN_HEIGHT[difficulty]
    ignored_dt = np.where(small, 1, np.where(dname == cls, 0, -1)).astype(np.int64)
    return int((ignored_gt == 0).sum()), ignored_gt, ignored_dt, dc.reshape(-1, 4)


MAX_DISTANCE = (30, 50, 70)          # distance ranges of get_distance_eval_result: [0, 30], (30, 50], (50, 70] metres


def clean_data_by_distance(gt_anno, dt_anno, current_class, difficulty):
    """As clean_data, with the difficulty index selecting a DISTANCE range (norm of the object's location) and the "hard"
    occlusion / truncation / height limits for every range (eval.py:83-157, DISTANCE_COVER = False)."""
    cls = CLASS_NAMES[current_class]
    gname = np.char.lower(np.asarray(gt_anno["name"], dtype=str)) if len(gt_anno["name"]) else np.zeros(0, dtype=str)
    same = gname == cls
    neighbour = ((gname == "person_sitting") & (cls == "pedestrian")) | ((gname == "van") & (cls == "car"))
    bbox = np.asarray(gt_anno["bbox"], dtype=np.float64).reshape(-1, 4)
    dis = np.linalg.norm(np.asarray(gt_anno["location"], dtype=np.float64).reshape(-1, 3), axis=1)
    hard = (np.asarray(gt_anno["occluded"]) > MAX_OCCLUSION[2]) | (np.asarray(gt_anno["truncated"]) > MAX_TRUNCATION[2]) \
        | ((bbox[:, 3] - bbox[:, 1]) <= MIN_HEIGHT[2]) | (dis > MAX_DISTANCE[difficulty])
    if difficulty > 0:
        hard = hard | (dis <= MAX_DISTANCE[difficulty - 1])
    ignored_gt = np.where(same & ~hard, 0, np.where(neighbour | (same & hard), 1, -1)).astype(np.int64)
    dc = bbox[np.asarray(gt_anno["name"], dtype=str) == "DontCare"] if len(bbox) else np.zeros((0, 4))
    dname = np.char.lower(np.asarray(dt_anno["name"], dtype=str)) if len(dt_anno["name"]) else np.zeros(0, dtype=str)
    dbox = np.asarray(dt_anno["bbox"], dtype=np.float64).reshape(-1, 4)
    small = np.abs(dbox[:, 3] - dbox[:, 1]) < MIN_HEIGHT[2]
    ignored_dt = np.where(small, 1, np.where(dname == cls, 0, -1)).astype(np.int64)
    return int((ignored_gt == 0).sum()), ignored_gt, ignored_dt, dc.reshape(-1, 4)


def get_thresholds(scores, num_gt, num_sample_pts=N_SAMPLE_PTS):
    """Score thresholds at which recall crosses the 41 sample points (eval.py:10-26)."""
    scores = np.sort(np.asarray(scores, dtype=np.float64))[::-1]
    picked, current_recall, n = [], 0.0, len(scores)
    for i, score in enumerate(scores):
        l_recall = (i + 1) / num_gt
        r_recall = (i + 2) / num_gt if i < n - 1 else l_recall
        if (r_recall - current_recall) < (current_recall - l_recall) and i < n - 1:
            continue
        picked.append(score)
        current_recall += 1 / (num_sample_pts - 1.0)
    return picked


def image_box_overlap(boxes, query_boxes, criterion=-1):
    """Axis-aligned image boxes, no +1 convention (eval.py:160-190): IoU, or intersection / area of `boxes` (0), of
    `query_boxes` (1)."""
    b, q = np.asarray(boxes, dtype=np.float64).reshape(-1, 1, 4), np.asarray(query_boxes, dtype=np.float64).reshape(1, -1, 4)
    iw = np.minimum(b[..., 2], q[..., 2]) - np.maximum(b[..., 0], q[..., 0])
    ih = np.minimum(b[..., 3], q[..., 3]) - np.maximum(b[..., 1], q[..., 1])
    inter = iw * ih
    area_b, area_q = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1]), (q[..., 2] - q[..., 0]) * (q[..., 3] - q[..., 1])
    ua = area_b + area_q - inter if criterion == -1 else (area_b + 0 * area_q if criterion == 0 else (area_q + 0 * area_b if criterion == 1 else np.ones_like(inter)))
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where((iw > 0) & (ih > 0), inter / ua, 0.0)


def _metric_boxes(annos, metric):
    if metric == 0:
        return [np.asarray(a["bbox"], dtype=np.float64).reshape(-1, 4) for a in annos]
    cols = [0, 2] if metric == 1 else [0, 1, 2]
    return [np.concatenate([np.asarray(a["location"], dtype=np.float64).reshape(-1, 3)[:, cols],
                            np.asarray(a["dimensions"], dtype=np.float64).reshape(-1, 3)[:, cols],
                            np.asarray(a["rotation_y"], dtype=np.float64).reshape(-1, 1)], 1) for a in annos]


def calculate_iou_partly(gt_annos, dt_annos, metric, num_parts=50, max_boxes=4096):
    """Per-image overlap matrices [n_first, n_second] (first = `gt_annos` argument; eval_class passes the detections first, as
    the reference does, eval.py:538).  metric 0: image boxes, 1: BEV, 2: 3D (camera frame).  Images are batched into launches
    of at most `max_boxes` boxes per side; only the per-image diagonal blocks are kept.  `num_parts` is accepted for
    signature compatibility.  -> overlaps, total_first_num, total_second_num"""
    assert len(gt_annos) == len(dt_annos)
    first, second = _metric_boxes(gt_annos, metric), _metric_boxes(dt_annos, metric)
    n1, n2 = np.array([len(b) for b in first], dtype=np.int64), np.array([len(b) for b in second], dtype=np.int64)
    overlaps = [None] * len(first)
    if metric == 0:
        for i in range(len(first)):
            overlaps[i] = image_box_overlap(first[i], second[i])
        return overlaps, n1, n2
    fn = bev_box_overlap if metric == 1 else d3_box_overlap
    i = 0
    while i < len(first):
        j, a, b = i, 0, 0
        while j < len(first) and (j == i or (a + n1[j] <= max_boxes and b + n2[j] <= max_boxes)):
            a, b, j = a + n1[j], b + n2[j], j + 1
        block = fn(np.concatenate(first[i:j]), np.concatenate(second[i:j])).astype(np.float64) if a and b else np.zeros((a, b))
        r = c = 0
        for k in range(i, j):
            overlaps[k] = block[r:r + n1[k], c:c + n2[k]]
            r, c = r + n1[k], c + n2[k]
        i = j
    return overlaps, n1, n2


def _prepare_data(gt_annos, dt_annos, current_class, difficulty, DIForDIS=True):
    """Flat arrays for the native matcher: counts, gt rows (bbox, alpha), dt rows (bbox, alpha, score), ignore flags, DontCare."""
    n_gt, n_dt, n_dc, gt_rows, dt_rows, ign_gt, ign_dt, dcs, valid = [], [], [], [], [], [], [], [], 0
    for g, d in zip(gt_annos, dt_annos):
        nv, ig, idt, dc = (clean_data if DIForDIS else clean_data_by_distance)(g, d, current_class, difficulty)
        valid += nv
        ign_gt.append(ig); ign_dt.append(idt); dcs.append(dc)
        n_gt.append(len(ig)); n_dt.append(len(idt)); n_dc.append(len(dc))
        gt_rows.append(np.concatenate([np.asarray(g["bbox"], dtype=np.float64).reshape(-1, 4), np.asarray(g["alpha"], dtype=np.float64).reshape(-1, 1)], 1))
        dt_rows.append(np.concatenate([np.asarray(d["bbox"], dtype=np.float64).reshape(-1, 4), np.asarray(d["alpha"], dtype=np.float64).reshape(-1, 1),
                                       np.asarray(d["score"], dtype=np.float64).reshape(-1, 1)], 1))
    cat = lambda xs, w, dt=np.float64: np.ascontiguousarray(np.concatenate(xs) if xs else np.zeros((0, w)), dtype=dt)
    return {"n_gt": np.array(n_gt, dtype=np.int64), "n_dt": np.array(n_dt, dtype=np.int64), "n_dc": np.array(n_dc, dtype=np.int64),
            "gt": cat(gt_rows, 5), "dt": cat(dt_rows, 6), "ign_gt": cat(ign_gt, 0, np.int64).reshape(-1),
            "ign_dt": cat(ign_dt, 0, np.int64).reshape(-1), "dc": cat(dcs, 4), "num_valid_gt": valid}


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def eval_class(gt_annos, dt_annos, current_classes, difficultys, metric, min_overlaps, compute_aos=False, num_parts=50,
               overlaps=None, DIForDIS=True):
    """Precision / recall / orientation similarity at the 41 recall sample points, [class, difficulty, min_overlap, 41]
    (eval.py:508-633).  `overlaps`: optional per-image [n_dt, n_gt] matrices (default: computed here, on the GPU for the
    rotated metrics)."""
    assert len(gt_annos) == len(dt_annos)
    lib = load()
    if overlaps is None:
        overlaps = calculate_iou_partly(dt_annos, gt_annos, metric, num_parts)[0]
    flat = np.ascontiguousarray(np.concatenate([np.asarray(o, dtype=np.float64).reshape(-1) for o in overlaps]) if overlaps else np.zeros(0))
    shape = [len(current_classes), len(difficultys), len(min_overlaps), N_SAMPLE_PTS]
    precision, recall, aos = np.zeros(shape), np.zeros(shape), np.zeros(shape)
    for m, current_class in enumerate(current_classes):
        for l, difficulty in enumerate(difficultys):
            d = _prepare_data(gt_annos, dt_annos, current_class, difficulty, DIForDIS)
            assert int((d["n_gt"] * d["n_dt"]).sum()) == flat.size, "overlap matrices do not match the annotations"
            common = (len(gt_annos), _ptr(d["n_gt"]), _ptr(d["n_dt"]), _ptr(d["n_dc"]), _ptr(flat), _ptr(d["gt"]), _ptr(d["dt"]),
                      _ptr(d["ign_gt"]), _ptr(d["ign_dt"]), _ptr(d["dc"]))
            for k, min_overlap in enumerate(min_overlaps[:, metric, m]):
                scores = np.zeros(max(1, int(d["n_gt"].sum())))
                count = ctypes.c_longlong(0)
                if lib.mono_kitti_tp_scores_f64(*common, int(metric), float(min_overlap), _ptr(scores), ctypes.byref(count)):
                    raise RuntimeError("mono_kitti_tp_scores_f64 failed")
                thresholds = np.ascontiguousarray(get_thresholds(scores[:count.value], d["num_valid_gt"]), dtype=np.float64)
                pr = np.zeros((len(thresholds), 4))
                if lib.mono_kitti_pr_f64(*common, int(metric), float(min_overlap), _ptr(thresholds), len(thresholds), int(bool(compute_aos)),
                                         _ptr(pr)):
                    raise RuntimeError("mono_kitti_pr_f64 failed")
                n = len(thresholds)
                with np.errstate(divide="ignore", invalid="ignore"):
                    recall[m, l, k, :n] = pr[:, 0] / (pr[:, 0] + pr[:, 2])
                    precision[m, l, k, :n] = pr[:, 0] / (pr[:, 0] + pr[:, 1])
                    if compute_aos:
                        aos[m, l, k, :n] = pr[:, 3] / (pr[:, 0] + pr[:, 1])
                # monotone envelope from the right over the first n points (eval.py:618-624); like np.max, a NaN spreads leftwards
                for arr in (precision, recall) + ((aos,) if compute_aos else ()):
                    if n:
                        arr[m, l, k, :n] = np.maximum.accumulate(arr[m, l, k, :n][::-1])[::-1]
    return {"recall": recall, "precision": precision, "orientation": aos}


def get_mAP(prec):
    return prec[..., ::4].sum(-1) / 11 * 100


def get_mAP_R40(prec):
    return prec[..., 1:].sum(-1) / 40 * 100


def do_eval(gt_annos, dt_annos, current_classes, min_overlaps, compute_aos=False, PR_detail_dict=None, DIForDIS=True):
    """-> mAP_bbox, mAP_bev, mAP_3d, mAP_aos, and the four R40 variants; each [class, difficulty, min_overlap] (eval.py:655-700)."""
    difficultys = [0, 1, 2]
    out, out40 = [], []
    for metric, key in ((0, "bbox"), (1, "bev"), (2, "3d")):
        ret = eval_class(gt_annos, dt_annos, current_classes, difficultys, metric, min_overlaps, compute_aos and metric == 0,
                         DIForDIS=DIForDIS)
        out.append(get_mAP(ret["precision"])); out40.append(get_mAP_R40(ret["precision"]))
        if PR_detail_dict is not None:
            PR_detail_dict[key] = ret["precision"]
        if metric == 0:
            aos = (get_mAP(ret["orientation"]), get_mAP_R40(ret["orientation"])) if compute_aos else (None, None)
            if compute_aos and PR_detail_dict is not None:
                PR_detail_dict["aos"] = ret["orientation"]
    return out[0], out[1], out[2], aos[0], out40[0], out40[1], out40[2], aos[1]


def _report(gt_annos, dt_annos, current_classes, overlap_tables, levels, PR_detail_dict, DIForDIS):
    """Shared body of the two report functions: per class and overlap setting the AP / AP_R40 lines (same text layout as
    eval.py:900-985), and the dictionary of headline numbers at the first overlap setting keyed by `levels`."""
    name_to_class = {v: k for k, v in CLASS_TO_NAME.items()}
    if not isinstance(current_classes, (list, tuple)):
        current_classes = [current_classes]
    current_classes = [name_to_class[c] if isinstance(c, str) else c for c in current_classes]
    min_overlaps = np.stack(overlap_tables, 0)[:, :, current_classes]
    compute_aos = False
    for anno in dt_annos:                      # the first non-empty detection file decides (alpha == -10: no orientation)
        if np.asarray(anno["alpha"]).shape[0] != 0:
            compute_aos = bool(anno["alpha"][0] != -10)
            break
    bbox, bev, d3, aos, bbox40, bev40, d340, aos40 = do_eval(gt_annos, dt_annos, current_classes, min_overlaps, compute_aos, PR_detail_dict,
                                                             DIForDIS)
    text, ret = "", {}
    for j, cls in enumerate(current_classes):
        name = CLASS_TO_NAME[cls]
        for i in range(min_overlaps.shape[0]):
            for tag, (vb, ve, v3, va) in (("AP", (bbox, bev, d3, aos)), ("AP_R40", (bbox40, bev40, d340, aos40))):
                text += "%s %s@%.2f, %.2f, %.2f:\n" % ((name, tag) + tuple(min_overlaps[i, :, j]))
                text += "bbox AP:%.4f, %.4f, %.4f\n" % tuple(vb[j, :, i])
                text += "bev  AP:%.4f, %.4f, %.4f\n" % tuple(ve[j, :, i])
                text += "3d   AP:%.4f, %.4f, %.4f\n" % tuple(v3[j, :, i])
                if compute_aos:
                    text += "aos  AP:%.2f, %.2f, %.2f\n" % tuple(va[j, :, i])
            if i == 0:
                for suffix, (vb, ve, v3, va) in (("", (bbox, bev, d3, aos)), ("_R40", (bbox40, bev40, d340, aos40))):
                    for di, level in enumerate(levels):
                        if compute_aos:
                            ret["%s_aos_%s%s" % (name, level, suffix)] = va[j, di, 0]
                        ret["%s_3d_%s%s" % (name, level, suffix)] = v3[j, di, 0]
                        ret["%s_bev_%s%s" % (name, level, suffix)] = ve[j, di, 0]
                        ret["%s_image_%s%s" % (name, level, suffix)] = vb[j, di, 0]
    return text, ret, d340[0, 1, 0]


def get_official_eval_result(gt_annos, dt_annos, current_classes, PR_detail_dict=None):
    """The KITTI report: text, dict of headline numbers, Car-moderate 3D AP_R40 (eval.py:863-985; same overlap table, same
    text layout, same dictionary keys)."""
    overlap_0_7 = np.array([[0.7, 0.5, 0.5, 0.7, 0.5, 0.7]] * 3)
    overlap_0_5 = np.array([[0.5, 0.5, 0.5, 0.5, 0.5, 0.5], [0.5, 0.25, 0.25, 0.5, 0.25, 0.5], [0.5, 0.25, 0.25, 0.5, 0.25, 0.5]])
    overlap_0_3 = np.array([[0.3, 0.5, 0.5, 0.3, 0.5, 0.5], [0.3, 0.25, 0.25, 0.3, 0.25, 0.5], [0.3, 0.25, 0.25, 0.3, 0.25, 0.5]])
    return _report(gt_annos, dt_annos, current_classes, [overlap_0_7, overlap_0_5, overlap_0_3], ("easy", "moderate", "hard"),
                   PR_detail_dict, True)


def get_distance_eval_result(gt_annos, dt_annos, current_classes, PR_detail_dict=None):
    """The report by distance range (0-30 m, 30-50 m, 50-70 m) instead of difficulty (eval.py:988-1090): text and dictionary."""
    overlap_0_7 = np.array([[0.7, 0.5, 0.5, 0.7, 0.5, 0.7]] * 3)
    overlap_0_5 = np.array([[0.7, 0.5, 0.5, 0.7, 0.5, 0.5], [0.5, 0.25, 0.25, 0.5, 0.25, 0.5], [0.5, 0.25, 0.25, 0.5, 0.25, 0.5]])
    text, ret, _ = _report(gt_annos, dt_annos, current_classes, [overlap_0_7, overlap_0_5], ("30m", "50m", "70m"), PR_detail_dict, False)
    return text, ret
