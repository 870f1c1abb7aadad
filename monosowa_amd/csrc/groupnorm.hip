// GroupNorm(32 groups, 256 channels) (+ ReLU) on channels-last [B, HW, 256] tensors for gfx950 -- the normalisation
// of MonoDETR's input projections and depth-predictor head (reference: monodetr.py:68-88 `input_proj`,
// depth_predictor.py:27-52 `downsample / proj / upsample / depth_head`, all nn.GroupNorm(32, 256)).
//
// PyTorch's GroupNorm kernels are NCHW-only: on the channels-last activations MIOpen's convolutions produce, every
// call pays a layout copy in and out (forward and backward).  Here a wave owns one pixel per iteration (64 lanes x
// float4 = the pixel's 256 channels, 1 KB coalesced), a lane pair owns one 8-channel group, statistics are
// accumulated in f64 (sum / sum of squares; v_fma_f64 is full rate on CDNA4 and the kernels are HBM-bound).
//   forward : gn_stats_kernel  -> stats[b][g] = {sum, sumsq}          (f64 atomics, 64 per workgroup)
//             gn_apply_kernel  -> y = (x - mean) * rstd * gamma + beta (ReLU optional); mean/rstd saved
//   backward: gn_bwd_stats_kernel -> part[b][c] = {sum gy' * xhat, sum gy'}   (gy' = gy masked by the ReLU)
//             gn_bwd_apply_kernel -> gx = rstd * (gy' * gamma - (sum_g gy' gamma + xhat * sum_g gy' gamma xhat) / n)
// ggamma / gbeta are the sums of part over b: by one wave of gn_bwd_apply_kernel's first workgroup (ggamma_gbeta [2][256]), or by the caller.
#pragma once
#include <hip/hip_runtime.h>

namespace mono {

constexpr int kGnC = 256, kGnG = 32, kGnPix = 64;   // channels, groups, pixels per workgroup

__device__ __forceinline__ double shfl_xor_f64(double v, int m) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, m);
  hi = __shfl_xor(hi, m);
  return __hiloint2double(hi, lo);
}

// pre_bias (nullable): a per-channel bias added to x before the normalisation (the preceding convolution's bias, so
// that the convolution runs without its bias pass and the bias gradient falls out of the backward below).
__device__ __forceinline__ float4 load_bias(const float *pre_bias, int lane) {
  return pre_bias ? reinterpret_cast<const float4 *>(pre_bias)[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
}

__global__ __launch_bounds__(256) void gn_stats_kernel(const float *__restrict__ x, const float *__restrict__ pre_bias,
                                                       double *__restrict__ stats, int HW) {
  __shared__ double red[4][kGnG][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.y;
  const int p0 = blockIdx.x * kGnPix, p1 = min(p0 + kGnPix, HW);
  const float *xb = x + ((long long)b * HW) * kGnC + lane * 4;
  double s = 0.0, ss = 0.0;
  const float4 pb = load_bias(pre_bias, lane);
#pragma unroll 4
  for (int p = p0 + wave; p < p1; p += 4) {
    float4 v = *reinterpret_cast<const float4 *>(xb + (long long)p * kGnC);
    v.x += pb.x; v.y += pb.y; v.z += pb.z; v.w += pb.w;
    s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
    ss = fma((double)v.x, (double)v.x, ss); ss = fma((double)v.y, (double)v.y, ss);
    ss = fma((double)v.z, (double)v.z, ss); ss = fma((double)v.w, (double)v.w, ss);
  }
  s += shfl_xor_f64(s, 1);
  ss += shfl_xor_f64(ss, 1);
  if (!(lane & 1)) { red[wave][lane >> 1][0] = s; red[wave][lane >> 1][1] = ss; }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int g = threadIdx.x >> 1, k = threadIdx.x & 1;
    const double t = (red[0][g][k] + red[1][g][k]) + (red[2][g][k] + red[3][g][k]);
    atomicAdd(stats + ((long long)b * kGnG + g) * 2 + k, t);
  }
}

template <bool RELU>
__global__ __launch_bounds__(256) void gn_apply_kernel(const float *__restrict__ x, const float *__restrict__ pre_bias,
                                                       const double *__restrict__ stats,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta,
                                                       float *__restrict__ y, float *__restrict__ mean_rstd, int HW, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.y, g = lane >> 1;
  const double inv_n = 1.0 / (8.0 * HW);
  const double m = stats[((long long)b * kGnG + g) * 2] * inv_n;
  const double var = fmax(stats[((long long)b * kGnG + g) * 2 + 1] * inv_n - m * m, 0.0);
  const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + (double)eps));
  if (blockIdx.x == 0 && wave == 0 && !(lane & 1)) {
    mean_rstd[((long long)b * kGnG + g) * 2] = mean;
    mean_rstd[((long long)b * kGnG + g) * 2 + 1] = rstd;
  }
  const float4 ga = reinterpret_cast<const float4 *>(gamma)[lane], be = reinterpret_cast<const float4 *>(beta)[lane];
  const float4 sc = make_float4(rstd * ga.x, rstd * ga.y, rstd * ga.z, rstd * ga.w);
  const float4 pb = load_bias(pre_bias, lane);
  const int p0 = blockIdx.x * kGnPix, p1 = min(p0 + kGnPix, HW);
  const long long base = ((long long)b * HW) * kGnC + lane * 4;
#pragma unroll 4
  for (int p = p0 + wave; p < p1; p += 4) {
    const float4 v = *reinterpret_cast<const float4 *>(x + base + (long long)p * kGnC);
    float4 o = make_float4((v.x + pb.x - mean) * sc.x + be.x, (v.y + pb.y - mean) * sc.y + be.y,
                           (v.z + pb.z - mean) * sc.z + be.z, (v.w + pb.w - mean) * sc.w + be.w);
    if (RELU) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    *reinterpret_cast<float4 *>(y + base + (long long)p * kGnC) = o;
  }
}

// y is the saved forward output when RELU (its sign is the ReLU mask), unused otherwise.
template <bool RELU>
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const float *__restrict__ gy, const float *__restrict__ x,
                                                           const float *__restrict__ pre_bias, const float *__restrict__ y,
                                                           const float *__restrict__ mean_rstd, double *__restrict__ part, int HW) {
  __shared__ double red[3][64][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.y, g = lane >> 1;
  const float mean = mean_rstd[((long long)b * kGnG + g) * 2], rstd = mean_rstd[((long long)b * kGnG + g) * 2 + 1];
  const int p0 = blockIdx.x * kGnPix, p1 = min(p0 + kGnPix, HW);
  const long long base = ((long long)b * HW) * kGnC + lane * 4;
  double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const float4 pb = load_bias(pre_bias, lane);
#pragma unroll 2
  for (int p = p0 + wave; p < p1; p += 4) {
    float4 gv = *reinterpret_cast<const float4 *>(gy + base + (long long)p * kGnC);
    float4 v = *reinterpret_cast<const float4 *>(x + base + (long long)p * kGnC);
    v.x += pb.x; v.y += pb.y; v.z += pb.z; v.w += pb.w;
    if (RELU) {
      const float4 yo = *reinterpret_cast<const float4 *>(y + base + (long long)p * kGnC);
      gv.x = yo.x > 0.f ? gv.x : 0.f; gv.y = yo.y > 0.f ? gv.y : 0.f;
      gv.z = yo.z > 0.f ? gv.z : 0.f; gv.w = yo.w > 0.f ? gv.w : 0.f;
    }
    a[0] = fma((double)gv.x, (double)((v.x - mean) * rstd), a[0]); a[1] += (double)gv.x;
    a[2] = fma((double)gv.y, (double)((v.y - mean) * rstd), a[2]); a[3] += (double)gv.y;
    a[4] = fma((double)gv.z, (double)((v.z - mean) * rstd), a[4]); a[5] += (double)gv.z;
    a[6] = fma((double)gv.w, (double)((v.w - mean) * rstd), a[6]); a[7] += (double)gv.w;
  }
  if (wave) {
#pragma unroll
    for (int k = 0; k < 8; ++k) red[wave - 1][lane][k] = a[k];
  }
  __syncthreads();
  if (!wave) {
    double *dst = part + ((long long)b * kGnC + lane * 4) * 2;
#pragma unroll
    for (int k = 0; k < 8; ++k) atomicAdd(dst + k, a[k] + (red[0][lane][k] + red[1][lane][k]) + red[2][lane][k]);
  }
}

// gbias_partials (with pre_bias): one row of 256 per workgroup = sum of gx over the workgroup's pixels (the bias gradient
// is the sum of gx over batch and pixels; partial_sum_kernel adds the rows).
template <bool RELU>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float *__restrict__ gy, const float *__restrict__ x,
                                                           const float *__restrict__ pre_bias, const float *__restrict__ y,
                                                           const float *__restrict__ mean_rstd, const float *__restrict__ gamma,
                                                           const double *__restrict__ part, float *__restrict__ gx,
                                                           float *__restrict__ gbias_partials, int HW,
                                                           float *__restrict__ ggamma_gbeta) {
  __shared__ float4 red[3][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.y, g = lane >> 1;
  if (ggamma_gbeta && blockIdx.x == 0 && blockIdx.y == 0 && wave == 3) {
    // ggamma[c] = sum_b part[b][c][0], gbeta[c] = sum_b part[b][c][1]: `part` is complete when this kernel starts; one wave of
    // one workgroup adds the gridDim.y batches in order (ATen did it in three launches on the tiny tensor: cast, sum, transpose copy)
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int bb = 0; bb < (int)gridDim.y; ++bb) {
      const double *q = part + ((long long)bb * kGnC + lane * 4) * 2;
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] += q[k];
    }
    reinterpret_cast<float4 *>(ggamma_gbeta)[lane] = make_float4((float)s[0], (float)s[2], (float)s[4], (float)s[6]);
    reinterpret_cast<float4 *>(ggamma_gbeta + kGnC)[lane] = make_float4((float)s[1], (float)s[3], (float)s[5], (float)s[7]);
  }
  const float mean = mean_rstd[((long long)b * kGnG + g) * 2], rstd = mean_rstd[((long long)b * kGnG + g) * 2 + 1];
  const float4 ga = reinterpret_cast<const float4 *>(gamma)[lane];
  const double *pp = part + ((long long)b * kGnC + lane * 4) * 2;
  double A = pp[0] * ga.x + pp[2] * ga.y + pp[4] * ga.z + pp[6] * ga.w;     // sum_c gamma_c * sum gy' xhat
  double Bs = pp[1] * ga.x + pp[3] * ga.y + pp[5] * ga.z + pp[7] * ga.w;    // sum_c gamma_c * sum gy'
  A += shfl_xor_f64(A, 1);
  Bs += shfl_xor_f64(Bs, 1);
  const double inv_n = 1.0 / (8.0 * HW);
  const float a = (float)(A * inv_n), bs = (float)(Bs * inv_n);
  const int p0 = blockIdx.x * kGnPix, p1 = min(p0 + kGnPix, HW);
  const long long base = ((long long)b * HW) * kGnC + lane * 4;
  const float4 pb = load_bias(pre_bias, lane);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 2
  for (int p = p0 + wave; p < p1; p += 4) {
    float4 gv = *reinterpret_cast<const float4 *>(gy + base + (long long)p * kGnC);
    float4 v = *reinterpret_cast<const float4 *>(x + base + (long long)p * kGnC);
    v.x += pb.x; v.y += pb.y; v.z += pb.z; v.w += pb.w;
    if (RELU) {
      const float4 yo = *reinterpret_cast<const float4 *>(y + base + (long long)p * kGnC);
      gv.x = yo.x > 0.f ? gv.x : 0.f; gv.y = yo.y > 0.f ? gv.y : 0.f;
      gv.z = yo.z > 0.f ? gv.z : 0.f; gv.w = yo.w > 0.f ? gv.w : 0.f;
    }
    float4 o;
    o.x = rstd * (gv.x * ga.x - bs - (v.x - mean) * rstd * a);
    o.y = rstd * (gv.y * ga.y - bs - (v.y - mean) * rstd * a);
    o.z = rstd * (gv.z * ga.z - bs - (v.z - mean) * rstd * a);
    o.w = rstd * (gv.w * ga.w - bs - (v.w - mean) * rstd * a);
    *reinterpret_cast<float4 *>(gx + base + (long long)p * kGnC) = o;
    acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
  }
  if (gbias_partials) {
    if (wave) red[wave - 1][lane] = acc;
    __syncthreads();
    if (!wave) {
      const float4 r0 = red[0][lane], r1 = red[1][lane], r2 = red[2][lane];
      float *dst = gbias_partials + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * kGnC + lane * 4;
      *reinterpret_cast<float4 *>(dst) = make_float4(acc.x + r0.x + r1.x + r2.x, acc.y + r0.y + r1.y + r2.y,
                                                     acc.z + r0.z + r1.z + r2.z, acc.w + r0.w + r1.w + r2.w);
    }
  }
}

}  // namespace mono
