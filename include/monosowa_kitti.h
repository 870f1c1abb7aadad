/* monosowa_kitti.h -- C ABI of the inference / evaluation side: the rotated-box overlap kernels of the KITTI evaluation
 *  and the detection extraction of the inference loop (row f3).
 *
 * Replaces lib/datasets/kitti/kitti_eval_python/rotate_iou.py:263-330 (`rotate_iou_gpu_eval`, a numba-CUDA kernel) and
 * the CPU loop of eval.py:197-230 (`d3_box_overlap`).  Device pointers, asynchronous on `stream`.
 * criterion: -1 IoU, 0 intersection / area of the first operand, 1 intersection / area of the second, 2 intersection.
 * Return value: 0, -1 (NULL pointer), -2 (bad size / criterion) or a hipError_t.
 */
#ifndef MONOSOWA_KITTI_H
#define MONOSOWA_KITTI_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* boxes [N, 5], query [K, 5] = (cx, cy, w, h, angle; clockwise positive) -> out [N, K].
 * As in the reference, criterion 0 divides by the QUERY box's area and 1 by the box's (rotate_iou.py:249-260, :289). */
int mono_rotate_iou_f32(const float *boxes, const float *query, float *out, long long N, long long K, int criterion, void *stream);

/* Camera-frame 3D boxes [*, 7] = (x, y, z, d3, d4, d5, ry): BEV rectangle (x, z, d3, d5, ry), bottom at y, height d4;
 * out [N, K] = 3D overlap (criterion 0: / volume of the box, 1: / volume of the query box, as eval.py:210-221). */
int mono_box3d_overlap_f32(const float *boxes, const float *query, float *out, long long N, long long K, int criterion, void *stream);

/* Inference post-process: lib/helpers/decode_helper.py:58-111 `extract_dets_from_outputs` as one kernel.
 * logits [B, Q, C], boxes [B, Q, 6] (cx, cy, l, r, t, b), angle [B, Q, 24], size3d [B, Q, 3], depth [B, Q, 2] ->
 * out [B, K, 37] = [cls, score, x2d, y2d, w2d, h2d, depth, heading(24), size3d(3), x3d, y3d, sigma], the K best
 * sigmoid(logit) scores of each image in descending order (exact ties: smaller flat index first).  Q * C <= 8192. */
int mono_extract_dets_f32(const float *logits, const float *boxes, const float *angle, const float *size3d, const float *depth,
                          float *out, int B, int Q, int C, int K, void *stream);

/* KITTI AP accumulation, HOST functions: the official matching protocol as kitti_eval_python/eval.py:234-410
 * runs it (numba there), one call per (class, difficulty, min_overlap) over all images.  Host pointers.
 *   n_gt / n_dt / n_dc [n_images]: boxes per image;   overlaps: the images' [n_dt, n_gt] matrices (row = detection), concatenated;
 *   gt_data [sum n_gt, 5] = (x1, y1, x2, y2, alpha);   dt_data [sum n_dt, 6] = (x1, y1, x2, y2, alpha, score);
 *   ignored_gt / ignored_dt: 0 evaluate, 1 ignore, -1 other class (clean_data, eval.py:29-80);   dc_boxes [sum n_dc, 4].
 * mono_kitti_tp_scores_f64: scores of the true positives when every detection takes part (first pass, eval.py:563-577);
 *   scores_out has room for sum n_gt values.
 * mono_kitti_pr_f64: pr[t] += (tp, fp, fn, orientation similarity) at each score threshold (second pass, eval.py:363-410);
 *   metric 0 also forgives detections inside DontCare boxes.  Return 0, or -1 for a NULL pointer. */
int mono_kitti_tp_scores_f64(long long n_images, const int64_t *n_gt, const int64_t *n_dt, const int64_t *n_dc,
                             const double *overlaps, const double *gt_data, const double *dt_data, const int64_t *ignored_gt,
                             const int64_t *ignored_dt, const double *dc_boxes, int metric, double min_overlap,
                             double *scores_out, long long *n_scores);
int mono_kitti_pr_f64(long long n_images, const int64_t *n_gt, const int64_t *n_dt, const int64_t *n_dc, const double *overlaps,
                      const double *gt_data, const double *dt_data, const int64_t *ignored_gt, const int64_t *ignored_dt,
                      const double *dc_boxes, int metric, double min_overlap, const double *thresholds, long long n_thresholds,
                      int compute_aos, double *pr);

#ifdef __cplusplus
}
#endif
#endif
