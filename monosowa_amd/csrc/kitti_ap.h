// KITTI AP accumulation on the host: the detection <-> ground-truth matching of the official protocol as the
// reference evaluates it (kitti_eval_python/eval.py:234-349 `compute_statistics_jit`, :363-410 `fused_compute_statistics`,
// :160-190 `image_box_overlap` for the DontCare regions).  The reference compiles these loops with numba; here they are C++
// behind two C entry points, one call per (class, difficulty, min_overlap) over ALL images.
//
// Per image i: n_dt[i] detections, n_gt[i] ground-truth boxes, n_dc[i] DontCare boxes; overlaps = the images' matrices
// [n_dt[i], n_gt[i]] (row = detection), concatenated; gt_data rows = (bbox x1 y1 x2 y2, alpha), dt_data rows = (bbox, alpha,
// score); ignored_* = 0 evaluate / 1 ignore (neighbouring class, too hard, too small) / -1 other class (eval.py:29-80).
#pragma once
#include <math.h>
#include <stdint.h>
#include <vector>

namespace kitti_ap {

struct Image {
  const double *overlaps, *gt, *dt, *dc;
  const int64_t *ign_gt, *ign_dt;
  long long n_gt, n_dt, n_dc;
};

struct Stats { long long tp = 0, fp = 0, fn = 0; double similarity = 0; };

// One image at one score threshold.  with_fp = false: the first pass (every detection competes by SCORE; returns the scores
// of the true positives in `tp_scores`).  with_fp = true: detections below `thresh` are out, a ground truth takes the
// unassigned detection with the LARGEST overlap, false positives and (metric 0) DontCare hits are counted.
inline Stats match_image(const Image &im, int metric, double min_overlap, double thresh, bool with_fp, bool with_aos,
                         std::vector<double> *tp_scores, std::vector<char> &assigned, std::vector<char> &below,
                         std::vector<double> &delta) {
  Stats st;
  const long long n_dt = im.n_dt, n_gt = im.n_gt;
  assigned.assign((size_t)n_dt, 0);
  below.assign((size_t)n_dt, 0);
  delta.clear();
  auto score = [&](long long j) { return im.dt[j * 6 + 5]; };
  if (with_fp)
    for (long long j = 0; j < n_dt; ++j) below[j] = score(j) < thresh;
  for (long long i = 0; i < n_gt; ++i) {
    if (im.ign_gt[i] == -1) continue;
    long long det = -1;
    bool found = false, found_ignored = false;     // found_ignored: the candidate so far is an "ignore" detection
    double best_score = 0, best_overlap = 0;
    for (long long j = 0; j < n_dt; ++j) {
      if (im.ign_dt[j] == -1 || assigned[j] || below[j]) continue;
      const double ov = im.overlaps[j * n_gt + i];
      if (!(ov > min_overlap)) continue;
      if (!with_fp) {
        if (!found || score(j) > best_score) { det = j; best_score = score(j); found = true; }
      } else if ((ov > best_overlap || found_ignored) && im.ign_dt[j] == 0) {
        best_overlap = ov; det = j; found = true; found_ignored = false;
      } else if (!found && im.ign_dt[j] == 1) {
        det = j; found = true; found_ignored = true;
      }
    }
    if (!found) {
      if (im.ign_gt[i] == 0) ++st.fn;
    } else if (im.ign_gt[i] == 1 || im.ign_dt[det] == 1) {
      assigned[det] = 1;
    } else {
      ++st.tp;
      if (tp_scores) tp_scores->push_back(score(det));
      if (with_aos) delta.push_back(im.gt[i * 5 + 4] - im.dt[det * 6 + 4]);
      assigned[det] = 1;
    }
  }
  if (with_fp) {
    for (long long j = 0; j < n_dt; ++j)
      if (!(assigned[j] || im.ign_dt[j] == -1 || im.ign_dt[j] == 1 || below[j])) ++st.fp;
    long long stuff = 0;
    if (metric == 0) {
      // detections inside DontCare regions are not false positives: intersection / detection area (eval.py:160-190, criterion 0)
      for (long long k = 0; k < im.n_dc; ++k) {
        const double *q = im.dc + k * 4;
        for (long long j = 0; j < n_dt; ++j) {
          if (assigned[j] || im.ign_dt[j] == -1 || im.ign_dt[j] == 1 || below[j]) continue;
          const double *b = im.dt + j * 6;
          const double iw = fmin(b[2], q[2]) - fmax(b[0], q[0]);
          if (!(iw > 0)) continue;
          const double ih = fmin(b[3], q[3]) - fmax(b[1], q[1]);
          if (!(ih > 0)) continue;
          if (iw * ih / ((b[2] - b[0]) * (b[3] - b[1])) > min_overlap) { assigned[j] = 1; ++stuff; }
        }
      }
    }
    st.fp -= stuff;
    if (with_aos) {
      if (st.tp > 0 || st.fp > 0) {
        double s = 0;                                  // false positives count as similarity 0 (eval.py:339-347)
        for (double d : delta) s += (1.0 + cos(d)) / 2.0;
        st.similarity = s;
      } else {
        st.similarity = -1;
      }
    }
  }
  return st;
}

struct Cursor {
  const int64_t *n_gt, *n_dt, *n_dc;
  const double *overlaps, *gt, *dt, *dc;
  const int64_t *ign_gt, *ign_dt;
  Image next(long long i) {
    Image im{overlaps, gt, dt, dc, ign_gt, ign_dt, (long long)n_gt[i], (long long)n_dt[i], (long long)n_dc[i]};
    overlaps += im.n_gt * im.n_dt;
    gt += im.n_gt * 5; dt += im.n_dt * 6; dc += im.n_dc * 4;
    ign_gt += im.n_gt; ign_dt += im.n_dt;
    return im;
  }
};

}  // namespace kitti_ap

extern "C" {

// First pass: scores of the detections matched to an evaluated ground truth when every detection takes part
// (compute_statistics_jit with compute_fp = False, thresh = 0; eval.py:563-577).  scores_out holds up to sum(n_gt) values.
int mono_kitti_tp_scores_f64(long long n_images, const int64_t *n_gt, const int64_t *n_dt, const int64_t *n_dc,
                             const double *overlaps, const double *gt_data, const double *dt_data, const int64_t *ignored_gt,
                             const int64_t *ignored_dt, const double *dc_boxes, int metric, double min_overlap,
                             double *scores_out, long long *n_scores) {
  if (n_images < 0 || !n_gt || !n_dt || !n_dc || !scores_out || !n_scores) return -1;
  kitti_ap::Cursor cur{n_gt, n_dt, n_dc, overlaps, gt_data, dt_data, dc_boxes, ignored_gt, ignored_dt};
  std::vector<double> scores, delta;
  std::vector<char> assigned, below;
  for (long long i = 0; i < n_images; ++i) {
    const kitti_ap::Image im = cur.next(i);
    kitti_ap::match_image(im, metric, min_overlap, 0.0, false, false, &scores, assigned, below, delta);
  }
  for (size_t k = 0; k < scores.size(); ++k) scores_out[k] = scores[k];
  *n_scores = (long long)scores.size();
  return 0;
}

// Second pass: pr[t] = (tp, fp, fn, similarity) summed over the images at each score threshold (fused_compute_statistics,
// eval.py:363-410).  pr [n_thresholds, 4] is ACCUMULATED into (the caller zeroes it).
int mono_kitti_pr_f64(long long n_images, const int64_t *n_gt, const int64_t *n_dt, const int64_t *n_dc, const double *overlaps,
                      const double *gt_data, const double *dt_data, const int64_t *ignored_gt, const int64_t *ignored_dt,
                      const double *dc_boxes, int metric, double min_overlap, const double *thresholds, long long n_thresholds,
                      int compute_aos, double *pr) {
  if (n_images < 0 || !n_gt || !n_dt || !n_dc || (n_thresholds > 0 && (!thresholds || !pr))) return -1;
  kitti_ap::Cursor cur{n_gt, n_dt, n_dc, overlaps, gt_data, dt_data, dc_boxes, ignored_gt, ignored_dt};
  std::vector<double> delta;
  std::vector<char> assigned, below;
  for (long long i = 0; i < n_images; ++i) {
    const kitti_ap::Image im = cur.next(i);
    for (long long t = 0; t < n_thresholds; ++t) {
      const kitti_ap::Stats st = kitti_ap::match_image(im, metric, min_overlap, thresholds[t], true, compute_aos != 0, nullptr,
                                                       assigned, below, delta);
      pr[t * 4 + 0] += (double)st.tp;
      pr[t * 4 + 1] += (double)st.fp;
      pr[t * 4 + 2] += (double)st.fn;
      if (st.similarity != -1) pr[t * 4 + 3] += st.similarity;
    }
  }
  return 0;
}

}  // extern "C"
