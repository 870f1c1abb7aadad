"""CPU checker for the rotated-box overlap kernels (test infrastructure only; never imported by the product).

The reference's implementation is a numba-CUDA kernel (lib/datasets/kitti/kitti_eval_python/rotate_iou.py:263-330) whose
launch code cannot run here (numba and CUDA are absent).  PINNED since round 2: its DEVICE function (rotate_iou.py:17-259)
runs pair by pair as plain Python on float32 numpy arrays (oracle/gen_golden.py kitti_eval -> tests/golden/
kitti_rotate_iou.npz), and this oracle agrees with it to 3e-6 on every non-degenerate pair (tests/test_kitti_ap.py).
This oracle computes the same quantity -- area of the intersection of two rotated rectangles, combined by `criterion` exactly as
rotate_iou.py:249-260 and eval.py:197-230 do -- with an independent exact method in float64: Sutherland-Hodgman clipping
of one convex quadrilateral by the other and the shoelace formula.  Box convention as in the reference
(rotate_iou.py:217-238): (cx, cy, w, h, angle), corners (-w/2,-h/2), (-w/2,h/2), (w/2,h/2), (w/2,-h/2) rotated by
[[cos, sin], [-sin, cos]].
"""
import numpy as np


def corners(b):
    cx, cy, w, h, a = [float(x) for x in b]
    ca, sa = np.cos(a), np.sin(a)
    xs = np.array([-w / 2, -w / 2, w / 2, w / 2])
    ys = np.array([-h / 2, h / 2, h / 2, -h / 2])
    return np.stack([ca * xs + sa * ys + cx, -sa * xs + ca * ys + cy], 1)


def _clip(subject, clipper):
    out = [tuple(p) for p in subject]
    area2 = sum(clipper[i][0] * clipper[(i + 1) % 4][1] - clipper[(i + 1) % 4][0] * clipper[i][1] for i in range(4))
    sign = 1.0 if area2 > 0 else -1.0
    for i in range(4):
        a, b = clipper[i], clipper[(i + 1) % 4]
        side = lambda p: sign * ((b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0]))
        inp, out = out, []
        for j in range(len(inp)):
            p, q = inp[j], inp[(j + 1) % len(inp)]
            sp, sq = side(p), side(q)
            if sp >= 0:
                out.append(p)
            if (sp >= 0) != (sq >= 0):
                t = sp / (sp - sq)
                out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
        if not out:
            return []
    return out


def intersection_area(b1, b2):
    poly = _clip(corners(b1), corners(b2))
    if len(poly) < 3:
        return 0.0
    x = np.array([p[0] for p in poly])
    y = np.array([p[1] for p in poly])
    return 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(np.roll(x, -1), y))


def _ratio(inter, a1, a2, criterion):
    if criterion == -1:
        return inter / (a1 + a2 - inter)
    if criterion == 0:
        return inter / a1
    if criterion == 1:
        return inter / a2
    return inter


def rotate_iou(boxes, query, criterion=-1):
    """boxes [N,5], query [K,5] -> [N,K]; criterion 0 divides by the QUERY box's area, 1 by the box's
    (devRotateIoUEval is called as (query, box), rotate_iou.py:289)."""
    out = np.zeros((len(boxes), len(query)))
    for i, b in enumerate(boxes):
        for j, q in enumerate(query):
            out[i, j] = _ratio(intersection_area(q, b), q[2] * q[3], b[2] * b[3], criterion)
    return out


def box3d_overlap(boxes, query, criterion=-1):
    """Camera-frame boxes [*,7] = (x, y, z, d3, d4, d5, ry) (eval.py:197-230)."""
    out = np.zeros((len(boxes), len(query)))
    for i, b in enumerate(boxes):
        for j, q in enumerate(query):
            bev = intersection_area(q[[0, 2, 3, 5, 6]], b[[0, 2, 3, 5, 6]])
            if bev > 0:
                ih = min(b[1], q[1]) - max(b[1] - b[4], q[1] - q[4])
                if ih > 0:
                    out[i, j] = _ratio(ih * bev, b[3] * b[4] * b[5], q[3] * q[4] * q[5], criterion)
    return out
