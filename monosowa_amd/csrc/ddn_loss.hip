// Depth-map supervision of MonoDETR (DDNLoss: depth_predictor/ddn_loss/ddn_loss.py:12-127, balancer.py:7-81,
// focalloss.py:55-136) as one forward and one backward kernel, EIGHT lanes per depth-map pixel (each lane every eighth bin; round 4:
// one thread per pixel left the chip at 480 waves of serial 81-bin loops -- 161 + 75 us for 10 MB of logits):
//   target depth  = centre depth of the NEAREST ground-truth box covering the pixel (the reference paints boxes far to
//                   near, ddn_loss.py:56-62; box = floor(top-left) .. ceil(bottom-right), :48-50), 0 without a box;
//   target bin    = LID index floor(-0.5 + 0.5 sqrt(1 + 8 (d - d_min) / bin_size)), out of range / not finite -> num_bins
//                   (:85-100);
//   pixel loss    = sum_c (one_hot_c + eps) * (-alpha (1 - p_c)^gamma log p_c), p = softmax over the 81 bin logits;
//   weighting     = fg_weight on pixels inside some box, bg_weight elsewhere; total = sum / number of pixels.
// PyTorch needs ~35 launches over [B, 81, H, W] for this; here the logits are read once per direction.
// Logits are addressed with (batch, channel, pixel) strides, so NCHW and channels-last maps are used in place.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace mono {

struct DdnParams {
  int B, C, H, W, N;                 // C = num_bins + 1 logits per pixel, N = padded boxes per image
  long long sb, sc, sp;              // strides of the logits (floats): batch, channel, pixel (row-major h * W + w)
  float alpha, gamma, fg_weight, bg_weight, depth_min, depth_max, eps;
};

// target bin and weight of pixel (b, y, x): boxes [B, N, 4] = xyxy in depth-map pixels (float), depth [B, N], valid [B, N]
__device__ __forceinline__ void ddn_target(const DdnParams &p, const float *__restrict__ boxes, const float *__restrict__ depth,
                                           const unsigned char *__restrict__ valid, int b, int y, int x, int &bin, float &weight) {
  float nearest = INFINITY;
  bool fg = false;
  for (int i = 0; i < p.N; ++i) {
    if (!valid[b * p.N + i]) continue;
    const float *bx = boxes + ((long long)b * p.N + i) * 4;
    // Python slicing [v1:v2, u1:u2] of the integer box: a negative start counts from the end, then clamp into [0, n]
    long long u1 = (long long)floorf(bx[0]), v1 = (long long)floorf(bx[1]), u2 = (long long)ceilf(bx[2]), v2 = (long long)ceilf(bx[3]);
    auto norm = [](long long a, int n) { a = a < 0 ? a + n : a; return a < 0 ? 0LL : (a > n ? (long long)n : a); };
    u1 = norm(u1, p.W); u2 = norm(u2, p.W); v1 = norm(v1, p.H); v2 = norm(v2, p.H);
    if (y >= v1 && y < v2 && x >= u1 && x < u2) {
      fg = true;
      nearest = fminf(nearest, depth[b * p.N + i]);
    }
  }
  const float d = fg ? nearest : 0.f;
  const int num_bins = p.C - 1;
  const float bin_size = 2.f * (p.depth_max - p.depth_min) / (float)(num_bins * (1 + num_bins));
  const float idx = -0.5f + 0.5f * sqrtf(1.f + 8.f * (d - p.depth_min) / bin_size);
  const bool bad = (idx < 0.f) || (idx > (float)num_bins) || !isfinite(idx);
  bin = bad ? num_bins : (int)idx;
  weight = fg ? p.fg_weight : p.bg_weight;
}

constexpr int kDdnLanes = 8;          // lanes of one pixel: lane j takes bins j, j + 8, ... (an aligned group of 8 lanes of a wave)
__device__ __forceinline__ float group8_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 1)); v = fmaxf(v, __shfl_xor(v, 2)); return fmaxf(v, __shfl_xor(v, 4));
}
__device__ __forceinline__ float group8_sum(float v) {
  v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); return v + __shfl_xor(v, 4);
}

// partial[block] = sum over the block's pixels of weight * pixel loss   (the host sums the partials and divides)
__global__ __launch_bounds__(256) void ddn_loss_fwd_kernel(const float *__restrict__ logits, const float *__restrict__ boxes,
                                                           const float *__restrict__ depth, const unsigned char *__restrict__ valid,
                                                           float *__restrict__ partial, const DdnParams p) {
  __shared__ float red[4];
  const int slot = blockIdx.x * 256 + threadIdx.x, n_pix = p.B * p.H * p.W;
  const int sub = slot & (kDdnLanes - 1);
  const bool live = slot / kDdnLanes < n_pix;
  const int pix = live ? slot / kDdnLanes : n_pix - 1;                 // (dead lanes shadow the last pixel: the shuffles stay uniform)
  const int b = pix / (p.H * p.W), hw = pix - b * p.H * p.W, y = hw / p.W, x = hw - y * p.W;
  int bin; float weight;
  ddn_target(p, boxes, depth, valid, b, y, x, bin, weight);
  const float *z = logits + b * p.sb + hw * p.sp;
  float mx = -INFINITY;
  for (int c = sub; c < p.C; c += kDdnLanes) mx = fmaxf(mx, z[c * p.sc]);
  mx = group8_max(mx);
  float sum = 0.f;
  for (int c = sub; c < p.C; c += kDdnLanes) sum += expf(z[c * p.sc] - mx);
  const float lse = mx + logf(group8_sum(sum));
  float loss = 0.f;
  for (int c = sub; c < p.C; c += kDdnLanes) {
    const float ls = z[c * p.sc] - lse, pc = expf(ls);
    const float focal = -p.alpha * powf(1.f - pc, p.gamma) * ls;
    loss += ((c == bin ? 1.f : 0.f) + p.eps) * focal;
  }
  loss = live ? loss * weight : 0.f;
  for (int o = 32; o > 0; o >>= 1) loss += __shfl_xor(loss, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = loss;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// f'(p) p of the focal term f(p) = -alpha (1 - p)^gamma log p, with the factor p folded in analytically:
//   f'(p) p = -alpha (-gamma (1 - p)^(gamma - 1) p log p + (1 - p)^gamma)
// -- no division by p, so a bin whose softmax probability underflows to 0 (logit gap beyond ~103) contributes a finite
// term (p log p -> 0) like the reference's autograd through (1 - p)^gamma * log_softmax, instead of inf * 0 = NaN.
__device__ __forceinline__ float focal_dp_times_p(const DdnParams &p, float ls, float pc, float om) {
  return -p.alpha * (-p.gamma * powf(om, p.gamma - 1.f) * (ls * pc) + powf(om, p.gamma));
}

// grad_logits = g_scale * weight * d(pixel loss) / d logits, g_scale = grad of the total / number of pixels (device scalar)
__global__ __launch_bounds__(256) void ddn_loss_bwd_kernel(const float *__restrict__ logits, const float *__restrict__ boxes,
                                                           const float *__restrict__ depth, const unsigned char *__restrict__ valid,
                                                           const float *__restrict__ grad_total, float *__restrict__ grad_logits,
                                                           const DdnParams p) {
  const int slot = blockIdx.x * 256 + threadIdx.x, n_pix = p.B * p.H * p.W;
  const int sub = slot & (kDdnLanes - 1);
  const bool live = slot / kDdnLanes < n_pix;
  const int pix = live ? slot / kDdnLanes : n_pix - 1;
  const int b = pix / (p.H * p.W), hw = pix - b * p.H * p.W, y = hw / p.W, x = hw - y * p.W;
  int bin; float weight;
  ddn_target(p, boxes, depth, valid, b, y, x, bin, weight);
  const float *z = logits + b * p.sb + hw * p.sp;
  float *gz = grad_logits + b * p.sb + hw * p.sp;
  float mx = -INFINITY;
  for (int c = sub; c < p.C; c += kDdnLanes) mx = fmaxf(mx, z[c * p.sc]);
  mx = group8_max(mx);
  float sum = 0.f;
  for (int c = sub; c < p.C; c += kDdnLanes) sum += expf(z[c * p.sc] - mx);
  const float lse = mx + logf(group8_sum(sum));
  // L = sum_c a_c f(p_c), f = -alpha (1 - p)^gamma log p;  dL/dz_k = t_k - p_k sum_c t_c  with  t_c = a_c f'(p_c) p_c
  float T = 0.f;
  for (int c = sub; c < p.C; c += kDdnLanes) {
    const float ls = z[c * p.sc] - lse, pc = expf(ls), om = 1.f - pc;
    T += ((c == bin ? 1.f : 0.f) + p.eps) * focal_dp_times_p(p, ls, pc, om);
  }
  T = group8_sum(T);
  if (!live) return;
  const float scale = grad_total[0] * weight / (float)n_pix;
  for (int c = sub; c < p.C; c += kDdnLanes) {
    const float ls = z[c * p.sc] - lse, pc = expf(ls), om = 1.f - pc;
    const float t = ((c == bin ? 1.f : 0.f) + p.eps) * focal_dp_times_p(p, ls, pc, om);
    gz[c * p.sc] = scale * (t - pc * T);
  }
}

// ---- expected depth of the bin distribution (depth_predictor.py:90-91): weighted_depth = sum_c softmax(logits)_c * value_c ----
// PyTorch: softmax, a broadcast multiply and a strided channel reduction over [B, 81, H, W] (the reduction alone 158 us at
// B = 16); here one thread per pixel each way.  d logits_c = g * p_c * (value_c - E).
__global__ __launch_bounds__(256) void depth_expect_fwd_kernel(const float *__restrict__ logits, const float *__restrict__ values,
                                                               float *__restrict__ out, int n_pix, int HW, int C, long long sb,
                                                               long long sc, long long sp) {
  const int slot = blockIdx.x * 256 + threadIdx.x, sub = slot & (kDdnLanes - 1);
  const bool live = slot / kDdnLanes < n_pix;
  const int pix = live ? slot / kDdnLanes : n_pix - 1;
  const int b = pix / HW, hw = pix - b * HW;
  const float *z = logits + b * sb + hw * sp;
  float mx = -INFINITY;
  for (int c = sub; c < C; c += kDdnLanes) mx = fmaxf(mx, z[c * sc]);
  mx = group8_max(mx);
  float sum = 0.f, acc = 0.f;
  for (int c = sub; c < C; c += kDdnLanes) {
    const float e = expf(z[c * sc] - mx);
    sum += e;
    acc += e * values[c];
  }
  sum = group8_sum(sum);
  acc = group8_sum(acc);
  if (live && sub == 0) out[pix] = acc / sum;
}

__global__ __launch_bounds__(256) void depth_expect_bwd_kernel(const float *__restrict__ logits, const float *__restrict__ values,
                                                               const float *__restrict__ expect, const float *__restrict__ grad_out,
                                                               float *__restrict__ grad_logits, int n_pix, int HW, int C, long long sb,
                                                               long long sc, long long sp) {
  const int slot = blockIdx.x * 256 + threadIdx.x, sub = slot & (kDdnLanes - 1);
  const bool live = slot / kDdnLanes < n_pix;
  const int pix = live ? slot / kDdnLanes : n_pix - 1;
  const int b = pix / HW, hw = pix - b * HW;
  const float *z = logits + b * sb + hw * sp;
  float *gz = grad_logits + b * sb + hw * sp;
  float mx = -INFINITY;
  for (int c = sub; c < C; c += kDdnLanes) mx = fmaxf(mx, z[c * sc]);
  mx = group8_max(mx);
  float sum = 0.f;
  for (int c = sub; c < C; c += kDdnLanes) sum += expf(z[c * sc] - mx);
  sum = group8_sum(sum);
  if (!live) return;
  const float g = grad_out[pix] / sum, E = expect[pix];
  for (int c = sub; c < C; c += kDdnLanes) gz[c * sc] = g * expf(z[c * sc] - mx) * (values[c] - E);
}

}  // namespace mono
