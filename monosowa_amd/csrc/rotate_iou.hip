// Rotated-box overlaps for the KITTI evaluation: BEV IoU of rotated rectangles and camera-frame 3D
// box IoU, N x K pairs per call.  Replaces the numba-CUDA kernel of
// lib/datasets/kitti/kitti_eval_python/rotate_iou.py:263-330 (+ the CPU loop eval.py:197-223 for the 3D case).
//
// Same definition as the reference: the intersection polygon's vertices are the corners of either rectangle that lie
// inside the other plus the pairwise edge intersections; they are ordered around their centroid and the polygon is
// summed as a triangle fan.  One thread per (box, query) pair, the 64 x 64 tile's boxes staged in LDS, everything in
// registers (at most 24 candidate vertices); pure f32 like the reference.
#include <hip/hip_runtime.h>

#include "../../include/monosowa_kitti.h"

namespace kitti {

struct P2 { float x, y; };

__device__ __forceinline__ void corners_of(const float *b, P2 c[4]) {
  // (cx, cy, w, h, angle): corners (-w/2,-h/2), (-w/2,h/2), (w/2,h/2), (w/2,-h/2) rotated clockwise by angle
  const float ca = cosf(b[4]), sa = sinf(b[4]);
  const float hx = b[2] * 0.5f, hy = b[3] * 0.5f;
  const float xs[4] = {-hx, -hx, hx, hx}, ys[4] = {-hy, hy, hy, -hy};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    c[i].x = ca * xs[i] + sa * ys[i] + b[0];
    c[i].y = -sa * xs[i] + ca * ys[i] + b[1];
  }
}

__device__ __forceinline__ bool inside(P2 p, const P2 q[4]) {          // projections on two adjacent edges of q
  const float abx = q[1].x - q[0].x, aby = q[1].y - q[0].y, adx = q[3].x - q[0].x, ady = q[3].y - q[0].y;
  const float apx = p.x - q[0].x, apy = p.y - q[0].y;
  const float abab = abx * abx + aby * aby, abap = abx * apx + aby * apy;
  const float adad = adx * adx + ady * ady, adap = adx * apx + ady * apy;
  return abab >= abap && abap >= 0.f && adad >= adap && adap >= 0.f;
}

__device__ __forceinline__ bool cross_point(P2 A, P2 B, P2 C, P2 D, P2 &out) {   // segments AB x CD (proper crossing)
  const float bax = B.x - A.x, bay = B.y - A.y, dax = D.x - A.x, cax = C.x - A.x, day = D.y - A.y, cay = C.y - A.y;
  const bool acd = day * cax > cay * dax;
  const bool bcd = (D.y - B.y) * (C.x - B.x) > (C.y - B.y) * (D.x - B.x);
  if (acd == bcd) return false;
  const bool abc = cay * bax > bay * cax, abd = day * bax > bay * dax;
  if (abc == abd) return false;
  const float dcx = D.x - C.x, dcy = D.y - C.y;
  const float abba = A.x * B.y - B.x * A.y, cddc = C.x * D.y - D.x * C.y;
  const float dh = bay * dcx - bax * dcy;
  out.x = (abba * dcx - bax * cddc) / dh;
  out.y = (abba * dcy - bay * cddc) / dh;
  return true;
}

__device__ float intersection_area(const float *b1, const float *b2) {
  P2 c1[4], c2[4], v[24];
  corners_of(b1, c1);
  corners_of(b2, c2);
  int n = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (inside(c1[i], c2)) v[n++] = c1[i];
    if (inside(c2[i], c1)) v[n++] = c2[i];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      P2 t;
      if (cross_point(c1[i], c1[(i + 1) & 3], c2[j], c2[(j + 1) & 3], t)) v[n++] = t;
    }
  if (n < 3) return 0.f;
  // order around the centroid by a monotone pseudo-angle: x / |v| for the upper half plane, -2 - x / |v| below
  float cx = 0.f, cy = 0.f;
  for (int i = 0; i < n; ++i) { cx += v[i].x; cy += v[i].y; }
  cx /= n; cy /= n;
  float key[24];
  for (int i = 0; i < n; ++i) {
    const float dx = v[i].x - cx, dy = v[i].y - cy;
    const float d = sqrtf(dx * dx + dy * dy);
    float k = dx / d;
    if (dy / d < 0.f) k = -2.f - k;
    key[i] = k;
  }
  for (int i = 1; i < n; ++i) {                         // insertion sort (n <= 24, usually <= 8)
    const float k = key[i];
    const P2 p = v[i];
    int j = i;
    while (j > 0 && key[j - 1] > k) { key[j] = key[j - 1]; v[j] = v[j - 1]; --j; }
    key[j] = k;
    v[j] = p;
  }
  float area = 0.f;
  for (int i = 0; i < n - 2; ++i)
    area += fabsf(((v[0].x - v[i + 2].x) * (v[i + 1].y - v[i + 2].y) - (v[0].y - v[i + 2].y) * (v[i + 1].x - v[i + 2].x)) * 0.5f);
  return area;
}

__device__ __forceinline__ float ratio(float inter, float a1, float a2, int criterion) {
  if (criterion == -1) return inter / (a1 + a2 - inter);
  if (criterion == 0) return inter / a1;
  if (criterion == 1) return inter / a2;
  return inter;
}

// MODE 0: BEV boxes [*, 5] = (cx, cy, w, h, angle), out = overlap by `criterion` (areas: a1 = QUERY, a2 = box, the
// argument order of the reference's device function).  MODE 1: camera-frame 3D boxes [*, 7] = (x, y, z, d3, d4, d5, ry)
// with BEV rectangle (x, z, d3, d5, ry), bottom y and height d4 (eval.py:197-230).
template <int MODE>
__global__ __launch_bounds__(64) void overlap_kernel(const float *__restrict__ boxes, const float *__restrict__ query,
                                                     float *__restrict__ out, long long N, long long K, int criterion) {
  constexpr int W = MODE ? 7 : 5;
  __shared__ float sb[64 * W], sq[64 * W];
  const long long r0 = (long long)blockIdx.x * 64, c0 = (long long)blockIdx.y * 64;
  const int rows = (int)min(64LL, N - r0), cols = (int)min(64LL, K - c0);
  const int tx = threadIdx.x;
  if (tx < rows)
    for (int i = 0; i < W; ++i) sb[tx * W + i] = boxes[(r0 + tx) * W + i];
  if (tx < cols)
    for (int i = 0; i < W; ++i) sq[tx * W + i] = query[(c0 + tx) * W + i];
  __syncthreads();
  if (tx >= rows) return;
  const float *b = sb + tx * W;
  for (int j = 0; j < cols; ++j) {
    const float *q = sq + j * W;
    float res;
    if (MODE == 0) {
      res = ratio(intersection_area(q, b), q[2] * q[3], b[2] * b[3], criterion);
    } else {
      const float bb[5] = {b[0], b[2], b[3], b[5], b[6]}, qq[5] = {q[0], q[2], q[3], q[5], q[6]};
      const float bev = intersection_area(qq, bb);
      res = 0.f;
      if (bev > 0.f) {
        const float ih = fminf(b[1], q[1]) - fmaxf(b[1] - b[4], q[1] - q[4]);
        if (ih > 0.f) res = ratio(ih * bev, b[3] * b[4] * b[5], q[3] * q[4] * q[5], criterion);
      }
    }
    out[(r0 + tx) * K + c0 + j] = res;
  }
}

}  // namespace kitti

// ---- detections from network outputs: lib/helpers/decode_helper.py:58-111 (extract_dets_from_outputs) as ONE kernel ----
// prob = sigmoid(logits); top-K over the Q * C scores of an image; query = index / C, class = index % C; gather of the
// query's box / heading / depth / size; cxcylrtb -> xyxy -> cxcywh; sigma = exp(-log variance).  One workgroup per image.
// The rank of a score is counted directly (N = Q * C is 150 in inference): rank = #(larger scores) + #(equal scores with a
// smaller index) -- the order torch.topk produces on distinct scores, and a fixed, documented order on exact ties (where
// torch's is unspecified).  Row layout (37 floats): [cls, score, x2d, y2d, w2d, h2d, depth, heading(24), size3d(3), x3d,
// y3d, sigma].  Every arithmetic step is the reference's own expression, individually rounded (no fused multiply-add).
namespace dets {

constexpr int kThreads = 1024;
constexpr int kMaxScores = 8192;        // Q * C the kernel accepts (32 KB of LDS)

__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}

__global__ __launch_bounds__(kThreads) void extract_dets_kernel(
    const float *__restrict__ logits, const float *__restrict__ boxes, const float *__restrict__ angle,
    const float *__restrict__ size3d, const float *__restrict__ depth, float *__restrict__ out, int Q, int C, int K) {
  __shared__ float score[kMaxScores];
  const int b = blockIdx.x, N = Q * C;
  const float *lg = logits + (long long)b * N;
  for (int i = threadIdx.x; i < N; i += kThreads) score[i] = 1.0f / (1.0f + expf(-lg[i]));      // at::sigmoid
  __syncthreads();
  for (int i = threadIdx.x; i < N; i += kThreads) {
    const float s = score[i];
    int rank = 0;
    for (int j = 0; j < N; ++j) {
      const float t = score[j];
      rank += (t > s) || (t == s && j < i);
    }
    if (rank >= K) continue;
    const int q = i / C, cls = i - q * C;
    const float *bx = boxes + ((long long)b * Q + q) * 6;
    const float cx = bx[0], cy = bx[1];
    const float x0 = sub_rn(cx, bx[2]), y0 = sub_rn(cy, bx[4]), x1 = add_rn(cx, bx[3]), y1 = add_rn(cy, bx[5]);   // box_cxcylrtb_to_xyxy
    float *o = out + ((long long)b * K + rank) * 37;
    o[0] = (float)cls;
    o[1] = s;
    o[2] = add_rn(x0, x1) / 2;                                                                                   // box_xyxy_to_cxcywh
    o[3] = add_rn(y0, y1) / 2;
    o[4] = sub_rn(x1, x0);
    o[5] = sub_rn(y1, y0);
    const float *dp = depth + ((long long)b * Q + q) * 2;
    o[6] = dp[0];
    const float *an = angle + ((long long)b * Q + q) * 24;
    for (int k = 0; k < 24; ++k) o[7 + k] = an[k];
    const float *sz = size3d + ((long long)b * Q + q) * 3;
    o[31] = sz[0]; o[32] = sz[1]; o[33] = sz[2];
    o[34] = cx;
    o[35] = cy;
    o[36] = expf(-dp[1]);
  }
}

}  // namespace dets

extern "C" {

int mono_extract_dets_f32(const float *logits, const float *boxes, const float *angle, const float *size3d, const float *depth,
                          float *out, int B, int Q, int C, int K, void *stream_) {
  if (!logits || !boxes || !angle || !size3d || !depth || !out) return -1;
  if (B <= 0 || Q <= 0 || C <= 0 || K <= 0 || K > Q * C || (long long)Q * C > dets::kMaxScores) return -2;
  dets::extract_dets_kernel<<<B, dets::kThreads, 0, (hipStream_t)stream_>>>(logits, boxes, angle, size3d, depth, out, Q, C, K);
  return (int)hipGetLastError();
}

int mono_rotate_iou_f32(const float *boxes, const float *query, float *out, long long N, long long K, int criterion, void *stream_) {
  if (!boxes || !query || !out) return -1;
  if (N <= 0 || K <= 0 || criterion < -1 || criterion > 2) return -2;
  const dim3 grid((unsigned)((N + 63) / 64), (unsigned)((K + 63) / 64));
  kitti::overlap_kernel<0><<<grid, 64, 0, (hipStream_t)stream_>>>(boxes, query, out, N, K, criterion);
  return (int)hipGetLastError();
}

int mono_box3d_overlap_f32(const float *boxes, const float *query, float *out, long long N, long long K, int criterion, void *stream_) {
  if (!boxes || !query || !out) return -1;
  if (N <= 0 || K <= 0 || criterion < -1 || criterion > 2) return -2;
  const dim3 grid((unsigned)((N + 63) / 64), (unsigned)((K + 63) / 64));
  kitti::overlap_kernel<1><<<grid, 64, 0, (hipStream_t)stream_>>>(boxes, query, out, N, K, criterion);
  return (int)hipGetLastError();
}

}  // extern "C"

#include "kitti_ap.h"      // host-side AP accumulation of the same library (mono_kitti_tp_scores_f64, mono_kitti_pr_f64)
