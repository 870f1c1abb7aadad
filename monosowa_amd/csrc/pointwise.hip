// Pointwise HIP kernels for the dense (library) part of the path: what PyTorch-ROCm issues as two or three
// elementwise passes after a MIOpen convolution -- bias add, residual add, ReLU -- in one in-place pass.
//   y[r, c] = act(y[r, c] + bias[c] (+ residual[r, c])),  y: [rows, C] contiguous = an NHWC activation
// (frozen batch-norm is folded into the conv weights and `bias`, lib/models/monodetr/backbone.py:28-65).
// HBM-bound: one read (+ one for the residual) and one write per element, float4 per lane.
#include <hip/hip_runtime.h>

#include "groupnorm.hip"
#include "matched_losses.hip"
#include "ddn_loss.hip"
#include "head_tail.hip"
#include "lsap_device.hip"
#include "conv1x1_fused.hip"
#include "small_wgrad.hip"
#include <stdint.h>
#include <algorithm>

namespace mono {

template <bool RELU, bool RES>
__global__ __launch_bounds__(256) void bias_act_kernel(float *__restrict__ y, const float *__restrict__ bias,
                                                       const float *__restrict__ residual, long long n_vec, int c_vec) {
  // n_vec = rows * C / 4 float4 elements; c_vec = C / 4
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
    float4 v = reinterpret_cast<float4 *>(y)[i];
    const float4 b = reinterpret_cast<const float4 *>(bias)[i % c_vec];
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    if (RES) {
      const float4 r = reinterpret_cast<const float4 *>(residual)[i];
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    reinterpret_cast<float4 *>(y)[i] = v;
  }
}

// The frozen stem of the backbone (backbone.py:72-74 of the reference freezes conv1 / layer1): relu(y + bias) followed by the 3 x 3 /
// stride 2 / padding 1 max-pool, in ONE pass over the convolution's output -- relu(max(y) + b) = max(relu(y + b)): both are monotonic and
// the bias is constant over a window.  Channels-last; a thread owns 4 channels of one output pixel (9 float4 loads, neighbours' windows
// overlap in L2).  Forward only: nothing below layer 2 receives a gradient.
__global__ __launch_bounds__(256) void bias_relu_maxpool_kernel(const float *__restrict__ y, const float *__restrict__ bias,
                                                                float *__restrict__ out, int N, int H, int W, int c_vec, int OH, int OW) {
  const long long n_items = (long long)N * OH * OW * c_vec, stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += stride) {
    const int c = (int)(i % c_vec);
    long long t = i / c_vec;
    const int ox = (int)(t % OW);
    t /= OW;
    const int oy = (int)(t % OH), n = (int)(t / OH);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int iy = 2 * oy - 1 + dy;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int ix = 2 * ox - 1 + dx;
        if (ix < 0 || ix >= W) continue;
        const float4 v = reinterpret_cast<const float4 *>(y)[(((long long)n * H + iy) * W + ix) * c_vec + c];
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    }
    const float4 b = reinterpret_cast<const float4 *>(bias)[c];
    reinterpret_cast<float4 *>(out)[i] = make_float4(fmaxf(m.x + b.x, 0.f), fmaxf(m.y + b.y, 0.f), fmaxf(m.z + b.z, 0.f), fmaxf(m.w + b.w, 0.f));
  }
}

// ReLU variants with a BYTE MASK per float4 (bit k = element k positive): the forward writes 1 B per 16 B of output, the
// backward reads that instead of y (4 B per element from HBM: y was written long before its backward runs).
template <bool RES>
__global__ __launch_bounds__(256) void bias_relu_mask_kernel(float *__restrict__ y, const float *__restrict__ bias,
                                                             const float *__restrict__ residual, unsigned char *__restrict__ mask,
                                                             long long n_vec, int c_vec) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
    float4 v = reinterpret_cast<float4 *>(y)[i];
    const float4 b = reinterpret_cast<const float4 *>(bias)[i % c_vec];
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    if (RES) {
      const float4 r = reinterpret_cast<const float4 *>(residual)[i];
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    mask[i] = (unsigned char)((v.x > 0.f) | ((v.y > 0.f) << 1) | ((v.z > 0.f) << 2) | ((v.w > 0.f) << 3));
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    reinterpret_cast<float4 *>(y)[i] = v;
  }
}
// three gradients meeting at one ReLU output (a stage output of the backbone: the next stage's first convolution, its identity
// branch, and the feature-pyramid projection outside the body)
__global__ __launch_bounds__(256) void relu_grad_mask3_kernel(const float *__restrict__ ga, const float *__restrict__ gb,
                                                              const float *__restrict__ gc, const unsigned char *__restrict__ mask,
                                                              float *__restrict__ grad_in, long long n_vec) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
    float4 a = reinterpret_cast<const float4 *>(ga)[i];
    const float4 b = reinterpret_cast<const float4 *>(gb)[i], c = reinterpret_cast<const float4 *>(gc)[i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w;
    const unsigned m = mask[i];
    reinterpret_cast<float4 *>(grad_in)[i] = make_float4((m & 1u) ? a.x : 0.f, (m & 2u) ? a.y : 0.f, (m & 4u) ? a.z : 0.f, (m & 8u) ? a.w : 0.f);
  }
}
template <bool TWO>
__global__ __launch_bounds__(256) void relu_grad_mask_kernel(const float *__restrict__ ga, const float *__restrict__ gb,
                                                             const unsigned char *__restrict__ mask, float *__restrict__ grad_in,
                                                             long long n_vec) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
    float4 a = reinterpret_cast<const float4 *>(ga)[i];
    if (TWO) {
      const float4 b = reinterpret_cast<const float4 *>(gb)[i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    const unsigned m = mask[i];
    reinterpret_cast<float4 *>(grad_in)[i] = make_float4((m & 1u) ? a.x : 0.f, (m & 2u) ? a.y : 0.f, (m & 4u) ? a.z : 0.f, (m & 8u) ? a.w : 0.f);
  }
}

// The affine form for convolutions WITHOUT a residual: y = relu(y * scale[c] + shift[c]) (frozen batch-norm applied here
// instead of folded into the weights: no per-step weight multiply and no multiply in the weight's backward), byte mask as
// above; backward grad_in = grad * mask * scale[c].
__global__ __launch_bounds__(256) void affine_relu_mask_kernel(float *__restrict__ y, const float *__restrict__ scale,
                                                               const float *__restrict__ shift, unsigned char *__restrict__ mask,
                                                               long long n_vec, int c_vec) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
    float4 v = reinterpret_cast<float4 *>(y)[i];
    const float4 a = reinterpret_cast<const float4 *>(scale)[i % c_vec], b = reinterpret_cast<const float4 *>(shift)[i % c_vec];
    v.x = v.x * a.x + b.x; v.y = v.y * a.y + b.y; v.z = v.z * a.z + b.z; v.w = v.w * a.w + b.w;
    mask[i] = (unsigned char)((v.x > 0.f) | ((v.y > 0.f) << 1) | ((v.z > 0.f) << 2) | ((v.w > 0.f) << 3));
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    reinterpret_cast<float4 *>(y)[i] = v;
  }
}
__global__ __launch_bounds__(256) void affine_relu_grad_kernel(const float *__restrict__ g, const unsigned char *__restrict__ mask,
                                                               const float *__restrict__ scale, float *__restrict__ grad_in,
                                                               long long n_vec, int c_vec) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
    const float4 a = reinterpret_cast<const float4 *>(g)[i], sc = reinterpret_cast<const float4 *>(scale)[i % c_vec];
    const unsigned m = mask[i];
    reinterpret_cast<float4 *>(grad_in)[i] = make_float4((m & 1u) ? a.x * sc.x : 0.f, (m & 2u) ? a.y * sc.y : 0.f,
                                                         (m & 4u) ? a.z * sc.z : 0.f, (m & 8u) ? a.w * sc.w : 0.f);
  }
}

// grad_in = grad_out * (y > 0), optionally also written to a second buffer (the residual branch's gradient)
__global__ __launch_bounds__(256) void relu_grad_kernel(const float *__restrict__ grad_out, const float *__restrict__ y,
                                                        float *__restrict__ grad_in, long long n_vec) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
    const float4 g = reinterpret_cast<const float4 *>(grad_out)[i];
    const float4 v = reinterpret_cast<const float4 *>(y)[i];
    reinterpret_cast<float4 *>(grad_in)[i] =
        make_float4(v.x > 0.f ? g.x : 0.f, v.y > 0.f ? g.y : 0.f, v.z > 0.f ? g.z : 0.f, v.w > 0.f ? g.w : 0.f);
  }
}

// grad_in = scale[channel] * grad_out * (y > 0) on a channels-last tensor (c_vec = C / 4 float4 per pixel): the ReLU backward of a
// trainable 1 x 1 convolution + frozen norm WITHOUT an identity branch, with the norm's scale already on the gradient -- both of its
// consumers (dX = g' W, dW = g'^T x) then need no scaled copy of the weight / no rescaling of the weight gradient (2 launches per node)
__global__ __launch_bounds__(256) void relu_grad_scale_kernel(const float *__restrict__ grad_out, const float *__restrict__ y,
                                                              const float *__restrict__ scale, float *__restrict__ grad_in, long long n_vec,
                                                              int c_vec) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
    const float4 g = reinterpret_cast<const float4 *>(grad_out)[i], v = reinterpret_cast<const float4 *>(y)[i];
    const float4 sc = reinterpret_cast<const float4 *>(scale)[i % c_vec];
    reinterpret_cast<float4 *>(grad_in)[i] = make_float4(v.x > 0.f ? g.x * sc.x : 0.f, v.y > 0.f ? g.y * sc.y : 0.f,
                                                         v.z > 0.f ? g.z * sc.z : 0.f, v.w > 0.f ? g.w * sc.w : 0.f);
  }
}

// grad_in = (grad_a + grad_b) * (y > 0): the two consumers of a block output (next conv and the identity branch) and
// the ReLU backward in one pass (autograd would run an accumulation add and then the ReLU backward)
__global__ __launch_bounds__(256) void relu_grad2_kernel(const float *__restrict__ ga, const float *__restrict__ gb,
                                                         const float *__restrict__ y, float *__restrict__ grad_in, long long n_vec) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
    const float4 a = reinterpret_cast<const float4 *>(ga)[i], b = reinterpret_cast<const float4 *>(gb)[i];
    const float4 v = reinterpret_cast<const float4 *>(y)[i];
    reinterpret_cast<float4 *>(grad_in)[i] = make_float4(v.x > 0.f ? a.x + b.x : 0.f, v.y > 0.f ? a.y + b.y : 0.f,
                                                         v.z > 0.f ? a.z + b.z : 0.f, v.w > 0.f ? a.w + b.w : 0.f);
  }
}

// three consumers (the last block of a stage feeds the next stage, its identity branch and an input projection): their gradients meet
// here, masked by the sign of the block's OUTPUT (the epilogue-GEMM bottleneck path keeps no byte mask)
__global__ __launch_bounds__(256) void relu_grad3_kernel(const float *__restrict__ ga, const float *__restrict__ gb, const float *__restrict__ gc,
                                                         const float *__restrict__ y, float *__restrict__ grad_in, long long n_vec) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
    const float4 a = reinterpret_cast<const float4 *>(ga)[i], b = reinterpret_cast<const float4 *>(gb)[i], c = reinterpret_cast<const float4 *>(gc)[i];
    const float4 v = reinterpret_cast<const float4 *>(y)[i];
    reinterpret_cast<float4 *>(grad_in)[i] = make_float4(v.x > 0.f ? (a.x + b.x) + c.x : 0.f, v.y > 0.f ? (a.y + b.y) + c.y : 0.f,
                                                         v.z > 0.f ? (a.z + b.z) + c.z : 0.f, v.w > 0.f ? (a.w + b.w) + c.w : 0.f);
  }
}

// ---- LayerNorm(x + dropout(z)) over rows of 256 channels: one wave per row, float4 per lane -----------------
// (reference: `src = self.norm1(src + self.dropout1(src2))`, depthaware_transformer.py:339-354,500-515).
// The keep mask is a counter-based hash of (seed, element index), recomputed in the backward: no mask tensor.
__device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float keep_scale(unsigned long long seed, long long idx, unsigned threshold, float scale) {
  const unsigned h = mix32((unsigned)idx * 0x9E3779B1u ^ (unsigned)seed) ^ mix32((unsigned)(idx >> 32) + (unsigned)(seed >> 32));
  return h >= threshold ? scale : 0.f;        // P(drop) = threshold / 2^32
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ __launch_bounds__(256) void dropout_add_ln_fwd_kernel(
    const float *__restrict__ x, const float *__restrict__ z, const float *__restrict__ gamma,
    const float *__restrict__ beta, float *__restrict__ y, float *__restrict__ s_out, float *__restrict__ mean_out,
    float *__restrict__ rstd_out, long long rows, unsigned threshold, float scale, unsigned long long seed, float eps) {
  const int lane = threadIdx.x & 63;
  const float4 g = reinterpret_cast<const float4 *>(gamma)[lane], b = reinterpret_cast<const float4 *>(beta)[lane];
  const long long wave_stride = (long long)gridDim.x * 4;
  for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += wave_stride) {
    const long long e = r * 256 + lane * 4;
    const float4 xv = *reinterpret_cast<const float4 *>(x + e);
    float4 zv = *reinterpret_cast<const float4 *>(z + e);
    if (threshold) {
      zv.x *= keep_scale(seed, e, threshold, scale); zv.y *= keep_scale(seed, e + 1, threshold, scale);
      zv.z *= keep_scale(seed, e + 2, threshold, scale); zv.w *= keep_scale(seed, e + 3, threshold, scale);
    }
    const float4 sv = make_float4(xv.x + zv.x, xv.y + zv.y, xv.z + zv.z, xv.w + zv.w);
    const float mean = wave_sum(sv.x + sv.y + sv.z + sv.w) * (1.f / 256.f);
    const float dx = sv.x - mean, dy = sv.y - mean, dz = sv.z - mean, dw = sv.w - mean;
    const float var = wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.f / 256.f);
    const float rstd = rsqrtf(var + eps);
    *reinterpret_cast<float4 *>(s_out + e) = sv;
    *reinterpret_cast<float4 *>(y + e) = make_float4(dx * rstd * g.x + b.x, dy * rstd * g.y + b.y, dz * rstd * g.z + b.z, dw * rstd * g.w + b.w);
    if (lane == 0) { mean_out[r] = mean; rstd_out[r] = rstd; }
  }
}

// gs = rstd * (gy*gamma - mean(gy*gamma) - xhat * mean(gy*gamma*xhat));  gx = gs;  gz = gs * keep_scale;
// ggamma = sum_rows gy * xhat, gbeta = sum_rows gy, gzsum = sum_rows gz: per-workgroup partial rows [gridDim.x][768], added up by
// partial_sum_kernel (no same-address atomics)
__global__ __launch_bounds__(256) void dropout_add_ln_bwd_kernel(
    const float *__restrict__ gy, const float *__restrict__ s, const float *__restrict__ mean_in,
    const float *__restrict__ rstd_in, const float *__restrict__ gamma, float *__restrict__ gx, float *__restrict__ gz,
    float *__restrict__ partials, long long rows, unsigned threshold, float scale, unsigned long long seed) {
  __shared__ float4 part[3][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float4 g = reinterpret_cast<const float4 *>(gamma)[lane];
  float4 acc_g = make_float4(0.f, 0.f, 0.f, 0.f), acc_b = make_float4(0.f, 0.f, 0.f, 0.f), acc_z = make_float4(0.f, 0.f, 0.f, 0.f);
  const long long wave_stride = (long long)gridDim.x * 4;
  for (long long r = (long long)blockIdx.x * 4 + wave; r < rows; r += wave_stride) {
    const long long e = r * 256 + lane * 4;
    const float4 gv = *reinterpret_cast<const float4 *>(gy + e);
    const float4 sv = *reinterpret_cast<const float4 *>(s + e);
    const float mean = mean_in[r], rstd = rstd_in[r];
    const float4 xh = make_float4((sv.x - mean) * rstd, (sv.y - mean) * rstd, (sv.z - mean) * rstd, (sv.w - mean) * rstd);
    const float4 t = make_float4(gv.x * g.x, gv.y * g.y, gv.z * g.z, gv.w * g.w);
    const float m1 = wave_sum(t.x + t.y + t.z + t.w) * (1.f / 256.f);
    const float m2 = wave_sum(t.x * xh.x + t.y * xh.y + t.z * xh.z + t.w * xh.w) * (1.f / 256.f);
    const float4 gs = make_float4(rstd * (t.x - m1 - xh.x * m2), rstd * (t.y - m1 - xh.y * m2),
                                  rstd * (t.z - m1 - xh.z * m2), rstd * (t.w - m1 - xh.w * m2));
    *reinterpret_cast<float4 *>(gx + e) = gs;
    float4 gzv = gs;
    if (threshold) {
      gzv.x *= keep_scale(seed, e, threshold, scale); gzv.y *= keep_scale(seed, e + 1, threshold, scale);
      gzv.z *= keep_scale(seed, e + 2, threshold, scale); gzv.w *= keep_scale(seed, e + 3, threshold, scale);
    }
    *reinterpret_cast<float4 *>(gz + e) = gzv;
    acc_z.x += gzv.x; acc_z.y += gzv.y; acc_z.z += gzv.z; acc_z.w += gzv.w;       // sum_rows gz: the bias gradient of the linear behind z
    acc_g.x += gv.x * xh.x; acc_g.y += gv.y * xh.y; acc_g.z += gv.z * xh.z; acc_g.w += gv.w * xh.w;
    acc_b.x += gv.x; acc_b.y += gv.y; acc_b.z += gv.z; acc_b.w += gv.w;
  }
  part[0][wave][lane] = acc_g;
  part[1][wave][lane] = acc_b;
  part[2][wave][lane] = acc_z;
  __syncthreads();
  if (wave < 3) {                                                   // wave k adds up quantity k
    float4 a = part[wave][0][lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) { const float4 pa = part[wave][w][lane]; a.x += pa.x; a.y += pa.y; a.z += pa.z; a.w += pa.w; }
    *reinterpret_cast<float4 *>(partials + (long long)blockIdx.x * 768 + wave * 256 + lane * 4) = a;
  }
}

// ---- y = dropout(relu(h)) (FFN hidden activation, depthaware_transformer.py:352 `self.dropout2(F.relu(self.linear1(src)))`)
// forward: one pass, hash mask; backward: grad_h = grad_y * scale where y > 0 (y > 0 <=> kept and h > 0), zero elsewhere
__global__ __launch_bounds__(256) void relu_dropout_fwd_kernel(const float *__restrict__ h, float *__restrict__ y, long long n_vec,
                                                               unsigned threshold, float scale, unsigned long long seed) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
    float4 v = reinterpret_cast<const float4 *>(h)[i];
    const long long e = i * 4;
    v.x = v.x > 0.f ? v.x * keep_scale(seed, e, threshold, scale) : 0.f;
    v.y = v.y > 0.f ? v.y * keep_scale(seed, e + 1, threshold, scale) : 0.f;
    v.z = v.z > 0.f ? v.z * keep_scale(seed, e + 2, threshold, scale) : 0.f;
    v.w = v.w > 0.f ? v.w * keep_scale(seed, e + 3, threshold, scale) : 0.f;
    reinterpret_cast<float4 *>(y)[i] = v;
  }
}
__global__ __launch_bounds__(256) void relu_dropout_bwd_kernel(const float *__restrict__ gy, const float *__restrict__ y,
                                                               float *__restrict__ gh, long long n_vec, float scale) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
    const float4 g = reinterpret_cast<const float4 *>(gy)[i], v = reinterpret_cast<const float4 *>(y)[i];
    reinterpret_cast<float4 *>(gh)[i] = make_float4(v.x > 0.f ? g.x * scale : 0.f, v.y > 0.f ? g.y * scale : 0.f,
                                                    v.z > 0.f ? g.z * scale : 0.f, v.w > 0.f ? g.w * scale : 0.f);
  }
}

// the same over a [rows, 256] matrix, also leaving the column sums of grad_h (the bias gradient of the linear in front) as
// per-workgroup partial rows [gridDim.x][256] for partial_sum_kernel
__global__ __launch_bounds__(256) void relu_dropout_bwd_colsum_kernel(const float *__restrict__ gy, const float *__restrict__ y,
                                                                      float *__restrict__ gh, float *__restrict__ partials,
                                                                      long long rows, float scale) {
  __shared__ float4 part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const long long wave_stride = (long long)gridDim.x * 4;
  for (long long r = (long long)blockIdx.x * 4 + wave; r < rows; r += wave_stride) {
    const long long e = r * 256 + lane * 4;
    const float4 g = *reinterpret_cast<const float4 *>(gy + e), v = *reinterpret_cast<const float4 *>(y + e);
    const float4 o = make_float4(v.x > 0.f ? g.x * scale : 0.f, v.y > 0.f ? g.y * scale : 0.f, v.z > 0.f ? g.z * scale : 0.f,
                                 v.w > 0.f ? g.w * scale : 0.f);
    *reinterpret_cast<float4 *>(gh + e) = o;
    acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int w = 1; w < 4; ++w) { const float4 pa = part[w][lane]; acc.x += pa.x; acc.y += pa.y; acc.z += pa.z; acc.w += pa.w; }
    *reinterpret_cast<float4 *>(partials + (long long)blockIdx.x * 256 + lane * 4) = acc;
  }
}

// ---- column sums of a row-major [rows, C] matrix (bias gradients): out[c] += sum_r g[r][c] --------------------
// PyTorch's reduce kernel takes 26 us for [8800, 256] (few workgroups); here a wave walks rows with float4 lanes,
// the 4 waves of a workgroup combine through LDS into one partial row; partial_sum_kernel adds the partial rows.
// (Same-address global atomics serialise in L2: 1275 workgroups x 256 atomics cost 100 us, hence two stages.)
template <int JJ>
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ g, float *__restrict__ partials, long long rows,
                                                     int C, int rows_per_block, long long rows_per_batch, long long batch_stride) {
  __shared__ float4 red[3][JJ][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cv = C >> 2;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float4 acc[JJ];
#pragma unroll
  for (int j = 0; j < JJ; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
  for (long long r = r0 + wave; r < r1; r += 4) {
#pragma unroll
    for (int j = 0; j < JJ; ++j) {
      const int c = lane + 64 * j;
      if (c < cv) {
        // row r of the virtual [batch * rows_per_batch, C] matrix (contiguous input: rows_per_batch = rows, one batch)
        const long long bidx = r / rows_per_batch;
        const float4 v = *reinterpret_cast<const float4 *>(g + bidx * batch_stride + (r - bidx * rows_per_batch) * C + c * 4);
        acc[j].x += v.x; acc[j].y += v.y; acc[j].z += v.z; acc[j].w += v.w;
      }
    }
  }
  if (wave) {
#pragma unroll
    for (int j = 0; j < JJ; ++j) red[wave - 1][j][lane] = acc[j];
  }
  __syncthreads();
  if (!wave) {
#pragma unroll
    for (int j = 0; j < JJ; ++j) {
      const int c = lane + 64 * j;
      if (c < cv) {
        const float4 a = red[0][j][lane], b = red[1][j][lane], d = red[2][j][lane];
        *reinterpret_cast<float4 *>(partials + (long long)blockIdx.x * C + c * 4) =
            make_float4(acc[j].x + a.x + b.x + d.x, acc[j].y + a.y + b.y + d.y, acc[j].z + a.z + b.z + d.z, acc[j].w + a.w + b.w + d.w);
      }
    }
  }
}

// Column sums of a SHORT matrix in one launch (round 4): the decoder's bias gradients are [8800, 256..512] -- 2 - 4 MB, where the two-stage
// form above spends its time in two launches' fixed costs (8 + 5.8 us for 2.25 MB).  Here a 16-wave workgroup owns a strip of four float4
// columns (64 bytes of every row) over ALL rows: thread = (row lane of 256, column of 4), 8 rows in flight per thread, then the 256 row
// lanes are folded by wave shuffles and one LDS exchange -- no partial rows, no second launch.  C / 16 workgroups; the strips of one row
// share its 128-byte lines through L2 (the matrix is read from HBM once).
// OFF by default (0): back to back on one L2-warm matrix it wins ([8800, 256]: 7.6 against 10.3 us), but inside the train step, where its
// input has just been written by a GEMM on other CUs, it loses -- 17.2 us per call against 7.9 + 5.8 for the two stages (16 workgroups pull
// the 9 MB through 16 CUs, and two strips share every 128-byte line from different XCDs); the step measured 68.45 / 68.38 ms with it and
// 68.25 / 68.09 ms without, in alternating same-box runs.  -DMONO_COLSUM_STRIP_ROWS=12288 brings it back (tools/debug/colsum_time.py).
#ifndef MONO_COLSUM_STRIP_ROWS
#define MONO_COLSUM_STRIP_ROWS 0
#endif
constexpr long long kColsumStripRows = MONO_COLSUM_STRIP_ROWS;        // up to here mono_colsum_f32 takes the one-launch strip kernel
__global__ __launch_bounds__(1024) void colsum_strip_kernel(const float *__restrict__ g, float *__restrict__ out, long long rows, int C) {
  __shared__ float4 red[16][4];
  const int col = threadIdx.x & 3, rl = threadIdx.x >> 2, wave = threadIdx.x >> 6;
  const int c4 = blockIdx.x * 4 + col;                      // this thread's float4 column
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c4 * 4 < C) {
    const float *base = g + (long long)c4 * 4;
#pragma unroll 8
    for (long long r = rl; r < rows; r += 256) {
      const float4 v = *reinterpret_cast<const float4 *>(base + r * C);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  // fold the wave's 16 row lanes (lanes with equal `col`: lane ^ 4, 8, 16, 32)
#pragma unroll
  for (int o = 4; o < 64; o <<= 1) {
    acc.x += __shfl_xor(acc.x, o); acc.y += __shfl_xor(acc.y, o); acc.z += __shfl_xor(acc.z, o); acc.w += __shfl_xor(acc.w, o);
  }
  if ((threadIdx.x & 63) < 4) red[wave][col] = acc;
  __syncthreads();
  if (threadIdx.x < 4 && c4 * 4 < C) {
    float4 a = red[0][col];
#pragma unroll
    for (int w = 1; w < 16; ++w) { const float4 b = red[w][col]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    *reinterpret_cast<float4 *>(out + c4 * 4) = a;
  }
}

// Column sums for ANY C <= 1024 (81 depth bins, 3 classes ...): thread = column (+ 256, + 512, ...), a workgroup walks its rows;
// the partial rows are added by partial_sum_scalar_kernel.  (ATen's reduction takes 308 us for [30720, 81].)
__global__ __launch_bounds__(256) void colsum_scalar_kernel(const float *__restrict__ g, float *__restrict__ partials, long long rows,
                                                            int C, int rows_per_block) {
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  for (int c = threadIdx.x; c < C; c += 256) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    long long r = r0;
    for (; r + 3 < r1; r += 4) { a0 += g[r * C + c]; a1 += g[(r + 1) * C + c]; a2 += g[(r + 2) * C + c]; a3 += g[(r + 3) * C + c]; }
    for (; r < r1; ++r) a0 += g[r * C + c];
    partials[(long long)blockIdx.x * C + c] = (a0 + a1) + (a2 + a3);
  }
}
__global__ __launch_bounds__(256) void partial_sum_scalar_kernel(const float *__restrict__ partials, float *__restrict__ out, int n, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float a = 0.f;
  for (int k = 0; k < n; ++k) a += partials[(long long)k * C + c];
  out[c] = a;
}

// out[c] = sum_k partials[k][c], k < n: one 16-wave workgroup per 64 float4 columns, the waves split the rows.
__global__ __launch_bounds__(1024) void partial_sum_kernel(const float *__restrict__ partials, float *__restrict__ out, int n, int C) {
  __shared__ float4 red[15][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = blockIdx.x * 64 + lane;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < (C >> 2)) {
#pragma unroll 8
    for (int k = wave; k < n; k += 16) {
      const float4 v = *reinterpret_cast<const float4 *>(partials + (long long)k * C + c * 4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  if (wave) red[wave - 1][lane] = acc;
  __syncthreads();
  if (!wave && c < (C >> 2)) {
#pragma unroll
    for (int w = 0; w < 15; ++w) {
      const float4 a = red[w][lane];
      acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
    }
    *reinterpret_cast<float4 *>(out + c * 4) = acc;
  }
}

// Column sums of SEVERAL row ranges ("levels") of a [batch, S, C] tensor in one launch pair: rows [row0[l], row0[l] + rows[l]) of every
// batch for level l -> out[l][C].  The visual encoder's level_embed gradient and the bias gradient of its merged offsets / logits
// projection need the sums of d proj [16, 10200, 384] over each of the four pyramid levels (depthaware_transformer.py:232-240): four
// launch pairs of 960 / 480 / 120 / 30 workgroups each read their level at 2.2 TB/s (28 us per launch on average, 250 MB in 112 us);
// one grid over all levels reads the tensor once, all CUs busy to the end.
struct LevelPlan {
  int n;
  int row0[8], rows[8], rpb[8];
  int blk0[9];                                  // first workgroup of each level; blk0[n] = the grid
};
template <int JJ>
__global__ __launch_bounds__(256) void colsum_levels_kernel(const float *__restrict__ g, float *__restrict__ partials, const LevelPlan plan,
                                                            int batch, long long batch_stride, int C) {
  __shared__ float4 red[3][JJ][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cv = C >> 2;
  int l = 0;
  while (l + 1 < plan.n && (int)blockIdx.x >= plan.blk0[l + 1]) ++l;
  const long long rows_l = plan.rows[l], total = (long long)batch * rows_l;
  const long long r0 = (long long)((int)blockIdx.x - plan.blk0[l]) * plan.rpb[l];
  const long long r1 = r0 + plan.rpb[l] < total ? r0 + plan.rpb[l] : total;
  const float *base = g + (long long)plan.row0[l] * C;
  float4 acc[JJ];
#pragma unroll
  for (int j = 0; j < JJ; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
  for (long long r = r0 + wave; r < r1; r += 4) {
    const long long bidx = r / rows_l;
    const float *row = base + bidx * batch_stride + (r - bidx * rows_l) * C;
#pragma unroll
    for (int j = 0; j < JJ; ++j) {
      const int c = lane + 64 * j;
      if (c < cv) {
        const float4 v = *reinterpret_cast<const float4 *>(row + c * 4);
        acc[j].x += v.x; acc[j].y += v.y; acc[j].z += v.z; acc[j].w += v.w;
      }
    }
  }
  if (wave) {
#pragma unroll
    for (int j = 0; j < JJ; ++j) red[wave - 1][j][lane] = acc[j];
  }
  __syncthreads();
  if (!wave) {
#pragma unroll
    for (int j = 0; j < JJ; ++j) {
      const int c = lane + 64 * j;
      if (c < cv) {
        const float4 a = red[0][j][lane], b = red[1][j][lane], d = red[2][j][lane];
        *reinterpret_cast<float4 *>(partials + (long long)blockIdx.x * C + c * 4) =
            make_float4(acc[j].x + a.x + b.x + d.x, acc[j].y + a.y + b.y + d.y, acc[j].z + a.z + b.z + d.z, acc[j].w + a.w + b.w + d.w);
      }
    }
  }
}
// out[l][c] = sum of level l's partial rows (blockIdx.y = level); blockIdx.y = plan.n (when launched): out[n][c] = the sum over ALL levels
__global__ __launch_bounds__(1024) void partial_sum_levels_kernel(const float *__restrict__ partials, float *__restrict__ out, const LevelPlan plan, int C) {
  __shared__ float4 red[15][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = blockIdx.x * 64 + lane, l = blockIdx.y;
  const int k0 = l < plan.n ? plan.blk0[l] : 0, k1 = l < plan.n ? plan.blk0[l + 1] : plan.blk0[plan.n];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < (C >> 2)) {
#pragma unroll 8
    for (int k = k0 + wave; k < k1; k += 16) {
      const float4 v = *reinterpret_cast<const float4 *>(partials + (long long)k * C + c * 4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  if (wave) red[wave - 1][lane] = acc;
  __syncthreads();
  if (!wave && c < (C >> 2)) {
#pragma unroll
    for (int w = 0; w < 15; ++w) {
      const float4 a = red[w][lane];
      acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
    }
    *reinterpret_cast<float4 *>(out + (long long)l * C + c * 4) = acc;
  }
}

// ---- the reference's AdamW variant (lib/helpers/optimizer_helper.py:69-129) over every parameter in ONE launch ------
//   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g g;  p -= step_size * (wd p + m / (sqrt(v) + eps))
// One workgroup per chunk (<= 32768 elements of one tensor); the chunk table (four pointer arrays, lengths, decay) is
// rebuilt by the host each step (gradient buffers move).  float4 path when all four pointers are 16-byte aligned
// (DDP's bucket-view gradients need not be).
__device__ __forceinline__ void adamw_one(float &p, float g, float &m, float &v, float wd, float b1, float omb1, float b2,
                                          float omb2, float eps, float neg_step) {
  m = __fmaf_rn(omb1, g, __fmul_rn(m, b1));
  v = __fmaf_rn(__fmul_rn(omb2, g), g, __fmul_rn(v, b2));
  const float denom = __fadd_rn(__fsqrt_rn(v), eps);
  const float upd = __fadd_rn(__fmul_rn(p, wd), __fdiv_rn(m, denom));
  p = __fmaf_rn(neg_step, upd, p);
}

__global__ __launch_bounds__(256) void adamw_kernel(const unsigned long long *__restrict__ pp, const unsigned long long *__restrict__ gp,
                                                    const unsigned long long *__restrict__ mp, const unsigned long long *__restrict__ vp,
                                                    const int *__restrict__ ns, const float *__restrict__ wds, float b1, float omb1,
                                                    float b2, float omb2, float eps, float step_size) {
  const int c = blockIdx.x;
  float *p = reinterpret_cast<float *>(pp[c]), *m = reinterpret_cast<float *>(mp[c]), *v = reinterpret_cast<float *>(vp[c]);
  const float *g = reinterpret_cast<const float *>(gp[c]);
  const int n = ns[c];
  const float wd = wds[c], neg = -step_size;
  if (((pp[c] | gp[c] | mp[c] | vp[c]) & 15ull) == 0) {
    const int n4 = n >> 2;
    for (int i = threadIdx.x; i < n4; i += 256) {
      float4 P = reinterpret_cast<float4 *>(p)[i], M = reinterpret_cast<float4 *>(m)[i], V = reinterpret_cast<float4 *>(v)[i];
      const float4 G = reinterpret_cast<const float4 *>(g)[i];
      adamw_one(P.x, G.x, M.x, V.x, wd, b1, omb1, b2, omb2, eps, neg);
      adamw_one(P.y, G.y, M.y, V.y, wd, b1, omb1, b2, omb2, eps, neg);
      adamw_one(P.z, G.z, M.z, V.z, wd, b1, omb1, b2, omb2, eps, neg);
      adamw_one(P.w, G.w, M.w, V.w, wd, b1, omb1, b2, omb2, eps, neg);
      reinterpret_cast<float4 *>(p)[i] = P; reinterpret_cast<float4 *>(m)[i] = M; reinterpret_cast<float4 *>(v)[i] = V;
    }
    for (int i = (n4 << 2) + threadIdx.x; i < n; i += 256) adamw_one(p[i], g[i], m[i], v[i], wd, b1, omb1, b2, omb2, eps, neg);
  } else {
    for (int i = threadIdx.x; i < n; i += 256) adamw_one(p[i], g[i], m[i], v[i], wd, b1, omb1, b2, omb2, eps, neg);
  }
}

inline int grid_for_vec(long long n_vec) {
  long long g = (n_vec + 255) / 256;
  if (g > 256LL * 32) g = 256LL * 32;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace mono

extern "C" {

int mono_reduce_blocks(long long rows);

// y [rows, C] is updated in place; residual may be NULL; C % 4 == 0 and 16-byte aligned pointers required.
int mono_bias_act_f32(float *y, const float *bias, const float *residual, long long rows, int C, int relu, void *stream_) {
  if (!y || !bias) return -1;
  if (rows <= 0 || C <= 0 || (C & 3) || ((uintptr_t)y & 15) || ((uintptr_t)bias & 15) || ((uintptr_t)residual & 15)) return -2;
  hipStream_t stream = (hipStream_t)stream_;
  const long long n_vec = rows * C / 4;
  const int g = mono::grid_for_vec(n_vec);
  if (relu && residual) mono::bias_act_kernel<true, true><<<g, 256, 0, stream>>>(y, bias, residual, n_vec, C / 4);
  else if (relu) mono::bias_act_kernel<true, false><<<g, 256, 0, stream>>>(y, bias, nullptr, n_vec, C / 4);
  else if (residual) mono::bias_act_kernel<false, true><<<g, 256, 0, stream>>>(y, bias, residual, n_vec, C / 4);
  else mono::bias_act_kernel<false, false><<<g, 256, 0, stream>>>(y, bias, nullptr, n_vec, C / 4);
  return (int)hipGetLastError();
}

// out [N, OH, OW, C] = max_pool2d(relu(y + bias), kernel 3, stride 2, padding 1) of the channels-last y [N, H, W, C] (C % 4 == 0):
// OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1.
int mono_bias_relu_maxpool_nhwc_f32(const float *y, const float *bias, float *out, int N, int H, int W, int C, void *stream_) {
  if (!y || !bias || !out) return -1;
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || ((uintptr_t)y & 15) || ((uintptr_t)bias & 15) || ((uintptr_t)out & 15)) return -2;
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const long long n_items = (long long)N * OH * OW * (C / 4);
  mono::bias_relu_maxpool_kernel<<<mono::grid_for_vec(n_items), 256, 0, (hipStream_t)stream_>>>(y, bias, out, N, H, W, C / 4, OH, OW);
  return (int)hipGetLastError();
}

// Y[M, 256] = relu(relu(X[M, 64] + b_in) W[64, 256] + b_out + R[M, 256]): the tail of a frozen bottleneck in one pass (conv1x1_fused.hip).
// Rows are channels-last pixels; w is [K][N] (the 1 x 1 convolution's weight, transposed, norm scale folded in).  y may be res.
int mono_conv1x1_tail_f32(const float *x, const float *b_in, const float *w, const float *b_out, const float *res, float *y,
                          long long M, int K, int N, void *stream_) {
  if (!x || !b_in || !w || !b_out || !res || !y) return -1;
  if (M <= 0 || K != mono::kC1K || N != mono::kC1N) return -2;
  if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)res & 15) || ((uintptr_t)y & 15)) return -2;
  if ((const float *)y == res || (const float *)y == x) return -2;      // __restrict__ operands; res is read a channel block ahead of the y stores
  const long long strips = (M + 31) / 32, per_wg = mono::kC1Threads / 64;
  const int grid = (int)std::min<long long>(512, (strips + per_wg - 1) / per_wg);
  mono::conv1x1_tail_kernel<<<grid, mono::kC1Threads, 0, (hipStream_t)stream_>>>(x, b_in, w, b_out, res, y, M);
  return (int)hipGetLastError();
}

// Y[M, 256] = relu(relu(X[M, 64] + b_in) W[64, 256] + X0[M, 64] Wd[64, 256] + b_out): the tail of a stage's FIRST frozen bottleneck, the
// 1 x 1 downsample convolution of the block's input x0 (stride 1) evaluated into the same accumulator (conv1x1_fused.hip).
int mono_conv1x1_tail_ds_f32(const float *x, const float *b_in, const float *w, const float *x0, const float *wd, const float *b_out,
                             float *y, long long M, int K, int N, void *stream_) {
  if (!x || !b_in || !w || !x0 || !wd || !b_out || !y) return -1;
  if (M <= 0 || K != mono::kC1K || N != mono::kC1N) return -2;
  if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)x0 & 15) || ((uintptr_t)wd & 15) || ((uintptr_t)y & 15)) return -2;
  if ((const float *)y == x || (const float *)y == x0) return -2;
  const long long strips = (M + 31) / 32, per_wg = mono::kC1Threads2 / 64;
  const int grid = (int)std::min<long long>(256, (strips + per_wg - 1) / per_wg);
  mono::conv1x1_tail_ds_kernel<<<grid, mono::kC1Threads2, 0, (hipStream_t)stream_>>>(x, b_in, w, x0, wd, b_out, y, M);
  return (int)hipGetLastError();
}

// H[M, 64] = relu(X[M, K] W[K, 64] + b): the head of a frozen bottleneck in one pass (conv1x1_fused.hip); K = 64 or 256, w is [K][64].
int mono_conv1x1_head_f32(const float *x, const float *w, const float *b_out, float *y, long long M, int K, int N, void *stream_) {
  if (!x || !w || !b_out || !y) return -1;
  if (M <= 0 || (K != 64 && K != 256) || N != 64) return -2;
  if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 15)) return -2;
  const long long strips = (M + 31) / 32, per_wg = mono::kC1Threads / 64;
  const int grid = (int)std::min<long long>(512, (strips + per_wg - 1) / per_wg);
  if (K == 64) mono::conv1x1_head_kernel<64><<<grid, mono::kC1Threads, 0, (hipStream_t)stream_>>>(x, w, b_out, y, M);
  else mono::conv1x1_head_kernel<256><<<grid, mono::kC1Threads, 0, (hipStream_t)stream_>>>(x, w, b_out, y, M);
  return (int)hipGetLastError();
}

// y = relu(y + bias (+ residual)) in place, and mask[i] (one byte per 4 consecutive elements) = their sign bits.
int mono_bias_relu_mask_f32(float *y, const float *bias, const float *residual, unsigned char *mask, long long rows, int C,
                            void *stream_) {
  if (!y || !bias || !mask) return -1;
  if (rows <= 0 || C <= 0 || (C & 3) || ((uintptr_t)y & 15) || ((uintptr_t)bias & 15) || ((uintptr_t)residual & 15)) return -2;
  const long long n_vec = rows * C / 4;
  const int g = mono::grid_for_vec(n_vec);
  if (residual) mono::bias_relu_mask_kernel<true><<<g, 256, 0, (hipStream_t)stream_>>>(y, bias, residual, mask, n_vec, C / 4);
  else mono::bias_relu_mask_kernel<false><<<g, 256, 0, (hipStream_t)stream_>>>(y, bias, nullptr, mask, n_vec, C / 4);
  return (int)hipGetLastError();
}

// y = relu(y * scale[c] + shift[c]) in place + byte mask; backward: grad_in = grad * mask * scale[c]   ([rows, C], C % 4 == 0)
int mono_affine_relu_mask_f32(float *y, const float *scale, const float *shift, unsigned char *mask, long long rows, int C,
                              void *stream_) {
  if (!y || !scale || !shift || !mask) return -1;
  if (rows <= 0 || C <= 0 || (C & 3) || ((uintptr_t)y & 15) || ((uintptr_t)scale & 15) || ((uintptr_t)shift & 15)) return -2;
  const long long n_vec = rows * C / 4;
  mono::affine_relu_mask_kernel<<<mono::grid_for_vec(n_vec), 256, 0, (hipStream_t)stream_>>>(y, scale, shift, mask, n_vec, C / 4);
  return (int)hipGetLastError();
}
int mono_affine_relu_grad_f32(const float *grad, const unsigned char *mask, const float *scale, float *grad_in, long long rows,
                              int C, void *stream_) {
  if (!grad || !mask || !scale || !grad_in) return -1;
  if (rows <= 0 || C <= 0 || (C & 3) || ((uintptr_t)grad & 15) || ((uintptr_t)scale & 15) || ((uintptr_t)grad_in & 15)) return -2;
  const long long n_vec = rows * C / 4;
  mono::affine_relu_grad_kernel<<<mono::grid_for_vec(n_vec), 256, 0, (hipStream_t)stream_>>>(grad, mask, scale, grad_in, n_vec, C / 4);
  return (int)hipGetLastError();
}

// grad_in = (grad_a (+ grad_b)) where the mask bit is set, else 0; grad_b may be NULL.  n % 4 == 0.
int mono_relu_grad_mask_f32(const float *grad_a, const float *grad_b, const unsigned char *mask, float *grad_in, long long n,
                            void *stream_) {
  if (!grad_a || !mask || !grad_in) return -1;
  if (n <= 0 || (n & 3) || ((uintptr_t)grad_a & 15) || ((uintptr_t)grad_b & 15) || ((uintptr_t)grad_in & 15)) return -2;
  const int g = mono::grid_for_vec(n / 4);
  if (grad_b) mono::relu_grad_mask_kernel<true><<<g, 256, 0, (hipStream_t)stream_>>>(grad_a, grad_b, mask, grad_in, n / 4);
  else mono::relu_grad_mask_kernel<false><<<g, 256, 0, (hipStream_t)stream_>>>(grad_a, nullptr, mask, grad_in, n / 4);
  return (int)hipGetLastError();
}

int mono_relu_grad_mask3_f32(const float *grad_a, const float *grad_b, const float *grad_c, const unsigned char *mask, float *grad_in,
                             long long n, void *stream_) {
  if (!grad_a || !grad_b || !grad_c || !mask || !grad_in) return -1;
  if (n <= 0 || (n & 3) || ((uintptr_t)grad_a & 15) || ((uintptr_t)grad_b & 15) || ((uintptr_t)grad_c & 15) || ((uintptr_t)grad_in & 15)) return -2;
  mono::relu_grad_mask3_kernel<<<mono::grid_for_vec(n / 4), 256, 0, (hipStream_t)stream_>>>(grad_a, grad_b, grad_c, mask, grad_in, n / 4);
  return (int)hipGetLastError();
}

int mono_relu_grad2_f32(const float *grad_a, const float *grad_b, const float *y, float *grad_in, long long n, void *stream_) {
  if (!grad_a || !grad_b || !y || !grad_in) return -1;
  if (n <= 0 || (n & 3) || ((uintptr_t)grad_a & 15) || ((uintptr_t)grad_b & 15) || ((uintptr_t)y & 15) || ((uintptr_t)grad_in & 15)) return -2;
  mono::relu_grad2_kernel<<<mono::grid_for_vec(n / 4), 256, 0, (hipStream_t)stream_>>>(grad_a, grad_b, y, grad_in, n / 4);
  return (int)hipGetLastError();
}

int mono_relu_grad3_f32(const float *grad_a, const float *grad_b, const float *grad_c, const float *y, float *grad_in, long long n,
                        void *stream_) {
  if (!grad_a || !grad_b || !grad_c || !y || !grad_in) return -1;
  if (n <= 0 || (n & 3) || ((uintptr_t)grad_a & 15) || ((uintptr_t)grad_b & 15) || ((uintptr_t)grad_c & 15) || ((uintptr_t)y & 15) ||
      ((uintptr_t)grad_in & 15))
    return -2;
  mono::relu_grad3_kernel<<<mono::grid_for_vec(n / 4), 256, 0, (hipStream_t)stream_>>>(grad_a, grad_b, grad_c, y, grad_in, n / 4);
  return (int)hipGetLastError();
}

int mono_relu_grad_f32(const float *grad_out, const float *y, float *grad_in, long long n, void *stream_) {
  if (!grad_out || !y || !grad_in) return -1;
  if (n <= 0 || (n & 3) || ((uintptr_t)grad_out & 15) || ((uintptr_t)y & 15) || ((uintptr_t)grad_in & 15)) return -2;
  mono::relu_grad_kernel<<<mono::grid_for_vec(n / 4), 256, 0, (hipStream_t)stream_>>>(grad_out, y, grad_in, n / 4);
  return (int)hipGetLastError();
}

// grad_in = scale[c] * grad_out * (y > 0), channels-last [.., C] with n elements in all; C % 4 == 0, n % C == 0.
int mono_relu_grad_scale_f32(const float *grad_out, const float *y, const float *scale, float *grad_in, long long n, int C, void *stream_) {
  if (!grad_out || !y || !scale || !grad_in) return -1;
  if (n <= 0 || C <= 0 || (C & 3) || n % C || ((uintptr_t)grad_out & 15) || ((uintptr_t)y & 15) || ((uintptr_t)grad_in & 15) || ((uintptr_t)scale & 15))
    return -2;
  mono::relu_grad_scale_kernel<<<mono::grid_for_vec(n / 4), 256, 0, (hipStream_t)stream_>>>(grad_out, y, scale, grad_in, n / 4, C / 4);
  return (int)hipGetLastError();
}


// LayerNorm over C = 256 of x + dropout(z), rows x 256 contiguous.  p in [0, 1): drop probability (0 = no dropout);
// the keep mask depends only on (seed, element index).  s (the pre-norm sum), mean, rstd are saved for the backward.
int mono_dropout_add_layernorm_fwd_f32(const float *x, const float *z, const float *gamma, const float *beta, float *y,
                                       float *s, float *mean, float *rstd, long long rows, int C, float p,
                                       unsigned long long seed, float eps, void *stream_) {
  if (!x || !z || !gamma || !beta || !y || !s || !mean || !rstd) return -1;
  if (rows <= 0 || C != 256 || !(p >= 0.f && p < 1.f)) return -2;
  const unsigned threshold = (unsigned)((double)p * 4294967296.0);
  const float scale = 1.f / (1.f - p);
  long long g = (rows + 3) / 4;
  if (g > 256 * 16) g = 256 * 16;
  mono::dropout_add_ln_fwd_kernel<<<(int)g, 256, 0, (hipStream_t)stream_>>>(x, z, gamma, beta, y, s, mean, rstd, rows,
                                                                           threshold, scale, seed, eps);
  return (int)hipGetLastError();
}

// ggamma_gbeta: [3, 256] contiguous (ggamma, gbeta, column sums of gz), overwritten.  partials: scratch of
// mono_reduce_blocks(rows) * 768 floats.
int mono_dropout_add_layernorm_bwd_f32(const float *gy, const float *s, const float *mean, const float *rstd,
                                       const float *gamma, float *gx, float *gz, float *ggamma_gbeta, float *partials,
                                       long long rows, int C, float p, unsigned long long seed, void *stream_) {
  if (!gy || !s || !mean || !rstd || !gamma || !gx || !gz || !ggamma_gbeta || !partials) return -1;
  if (rows <= 0 || C != 256 || !(p >= 0.f && p < 1.f)) return -2;
  const unsigned threshold = (unsigned)((double)p * 4294967296.0);
  const float scale = 1.f / (1.f - p);
  const int g = mono_reduce_blocks(rows);
  hipStream_t st = (hipStream_t)stream_;
  mono::dropout_add_ln_bwd_kernel<<<g, 256, 0, st>>>(gy, s, mean, rstd, gamma, gx, gz, partials, rows, threshold, scale, seed);
  mono::partial_sum_kernel<<<3, 1024, 0, st>>>(partials, ggamma_gbeta, g, 768);
  return (int)hipGetLastError();
}

// y = dropout_p(relu(h)), n contiguous floats (n % 4 == 0); the mask is a hash of (seed, element index).
int mono_relu_dropout_fwd_f32(const float *h, float *y, long long n, float p, unsigned long long seed, void *stream_) {
  if (!h || !y) return -1;
  if (n <= 0 || (n & 3) || !(p >= 0.f && p < 1.f) || ((uintptr_t)h & 15) || ((uintptr_t)y & 15)) return -2;
  const unsigned threshold = (unsigned)((double)p * 4294967296.0);
  mono::relu_dropout_fwd_kernel<<<mono::grid_for_vec(n / 4), 256, 0, (hipStream_t)stream_>>>(h, y, n / 4, threshold, 1.f / (1.f - p), seed);
  return (int)hipGetLastError();
}

// grad_h = grad_y / (1 - p) where y > 0, else 0  (y = the forward output).
int mono_relu_dropout_bwd_f32(const float *grad_y, const float *y, float *grad_h, long long n, float p, void *stream_) {
  if (!grad_y || !y || !grad_h) return -1;
  if (n <= 0 || (n & 3) || !(p >= 0.f && p < 1.f) || ((uintptr_t)grad_y & 15) || ((uintptr_t)y & 15) || ((uintptr_t)grad_h & 15)) return -2;
  mono::relu_dropout_bwd_kernel<<<mono::grid_for_vec(n / 4), 256, 0, (hipStream_t)stream_>>>(grad_y, y, grad_h, n / 4, 1.f / (1.f - p));
  return (int)hipGetLastError();
}

// grad_h [rows, 256] as above plus colsum[256] = its column sums; partials: mono_reduce_blocks(rows) * 256 floats of scratch.
int mono_relu_dropout_bwd_colsum_f32(const float *grad_y, const float *y, float *grad_h, float *colsum, float *partials, long long rows,
                                     float p, void *stream_) {
  if (!grad_y || !y || !grad_h || !colsum || !partials) return -1;
  if (rows <= 0 || !(p >= 0.f && p < 1.f) || ((uintptr_t)grad_y & 15) || ((uintptr_t)y & 15) || ((uintptr_t)grad_h & 15)) return -2;
  const int g = mono_reduce_blocks(rows);
  hipStream_t st = (hipStream_t)stream_;
  mono::relu_dropout_bwd_colsum_kernel<<<g, 256, 0, st>>>(grad_y, y, grad_h, partials, rows, 1.f / (1.f - p));
  mono::partial_sum_kernel<<<1, 1024, 0, st>>>(partials, colsum, g, 256);
  return (int)hipGetLastError();
}

// Matched-pair losses of SetCriterion for all decoder layers (see matched_losses.hip).  Predictions [NL, B, Q, 6|2|3|24]
// contiguous; idx [3, NL, K] int64 = (image, query, flat target); targets flat over the batch.  out [NL, 6] = per-layer
// sums {center, bbox, giou, depth, dim, angle}; comp [NL] is saved for the backward.
int mono_matched_losses_fwd_f32(const float *boxes, const float *depth, const float *dims, const float *angle,
                                const long long *idx, const float *t_box, const float *t_depth, const float *t_size,
                                const long long *t_bin, const float *t_res, float *out, float *comp, int NL, int B, int Q,
                                int K, void *stream_) {
  if (!boxes || !depth || !dims || !angle || !idx || !t_box || !t_depth || !t_size || !t_bin || !t_res || !out || !comp) return -1;
  if (NL <= 0 || B <= 0 || Q <= 0 || K <= 0) return -2;
  const mono::MatchedArgs a{boxes, depth, dims, angle, idx, t_box, t_depth, t_size, t_res, t_bin, NL, B, Q, K};
  mono::matched_fwd_kernel<<<NL, 256, 0, (hipStream_t)stream_>>>(a, out, comp);
  return (int)hipGetLastError();
}

// grad_out [NL, 6]; g_* are the gradients of the four prediction tensors, ZERO on entry (only matched rows are written).
int mono_matched_losses_bwd_f32(const float *boxes, const float *depth, const float *dims, const float *angle,
                                const long long *idx, const float *t_box, const float *t_depth, const float *t_size,
                                const long long *t_bin, const float *t_res, const float *comp, const float *grad_out,
                                float *g_boxes, float *g_depth, float *g_dims, float *g_angle, int NL, int B, int Q, int K,
                                void *stream_) {
  if (!boxes || !depth || !dims || !angle || !idx || !t_box || !t_depth || !t_size || !t_bin || !t_res || !comp || !grad_out ||
      !g_boxes || !g_depth || !g_dims || !g_angle)
    return -1;
  if (NL <= 0 || B <= 0 || Q <= 0 || K <= 0) return -2;
  const mono::MatchedArgs a{boxes, depth, dims, angle, idx, t_box, t_depth, t_size, t_res, t_bin, NL, B, Q, K};
  mono::matched_bwd_kernel<<<NL, 256, 0, (hipStream_t)stream_>>>(a, comp, grad_out, g_boxes, g_depth, g_dims, g_angle);
  return (int)hipGetLastError();
}

// One AdamW step (the reference's variant, see adamw_kernel) for n_chunks chunks.  table: device buffer holding, back to
// back, p[n_chunks], g[n_chunks], m[n_chunks], v[n_chunks] (64-bit device addresses), n[n_chunks] (int32, elements
// per chunk), wd[n_chunks] (float).  step_size = lr * sqrt(1 - b2^t) / (1 - b1^t).  Scalars arrive as doubles so that
// 1 - beta is rounded to float once, like the Python scalars of the reference.
int mono_adamw_step_f32(const void *table, int n_chunks, double beta1, double beta2, double eps, double step_size, void *stream_) {
  if (!table) return -1;
  if (n_chunks <= 0) return -2;
  const unsigned long long *pp = reinterpret_cast<const unsigned long long *>(table);
  const int *ns = reinterpret_cast<const int *>(pp + 4 * (size_t)n_chunks);
  const float *wds = reinterpret_cast<const float *>(ns + n_chunks);
  mono::adamw_kernel<<<n_chunks, 256, 0, (hipStream_t)stream_>>>(pp, pp + n_chunks, pp + 2 * (size_t)n_chunks, pp + 3 * (size_t)n_chunks,
                                                               ns, wds, (float)beta1, (float)(1.0 - beta1), (float)beta2,
                                                               (float)(1.0 - beta2), (float)eps, (float)step_size);
  return (int)hipGetLastError();
}

// Number of workgroups (= partial rows of C floats the caller provides) the row reductions below use for `rows` rows.
int mono_reduce_blocks(long long rows) {
  long long rpb = 64;
  while ((rows + rpb - 1) / rpb > 1024) rpb *= 2;
  return (int)((rows + rpb - 1) / rpb);
}

// out[c] = sum over rows of g[r][c];  g row-major [rows, C], C % 4 == 0, C <= 1024.
// partials: scratch of mono_reduce_blocks(rows) * C floats.
int mono_colsum_f32(const float *g, float *out, float *partials, long long rows, int C, void *stream_) {
  if (!g || !out || !partials) return -1;
  if (rows <= 0 || C <= 0 || C % 4 || C > 1024) return -2;
  hipStream_t st = (hipStream_t)stream_;
  // (one-launch strip kernel: disabled by default, see MONO_COLSUM_STRIP_ROWS)
  if (rows <= mono::kColsumStripRows && C <= 384) {
    mono::colsum_strip_kernel<<<(C / 4 + 3) / 4, 1024, 0, st>>>(g, out, rows, C);
    return (int)hipGetLastError();
  }
  const int grid = mono_reduce_blocks(rows);
  const int rpb = (int)((rows + grid - 1) / grid + 63) / 64 * 64;
  if (C <= 256) mono::colsum_kernel<1><<<grid, 256, 0, st>>>(g, partials, rows, C, rpb, rows, 0);
  else if (C <= 512) mono::colsum_kernel<2><<<grid, 256, 0, st>>>(g, partials, rows, C, rpb, rows, 0);
  else mono::colsum_kernel<4><<<grid, 256, 0, st>>>(g, partials, rows, C, rpb, rows, 0);
  mono::partial_sum_kernel<<<(C / 4 + 63) / 64, 1024, 0, st>>>(partials, out, grid, C);
  return (int)hipGetLastError();
}

// Weight and bias gradient of a linear over R tokens (small_wgrad.hip).  mono_linear_wgrad_workspace: floats of scratch the call needs,
// 0 when the shape is not served (M, N multiples of 64, 64 <= R < 2^24).
long long mono_linear_wgrad_workspace(int R, int M, int N) {
  if (R < 64 || R >= (1 << 24) || M <= 0 || N <= 0 || M % 64 || N % 64 || M > 4096 || N > 4096) return 0;
  return (long long)mono::linear_wgrad_splits(R, M, N) * ((long long)M * N + M);
}
int mono_linear_wgrad_f32(const float *dy, long long ldy, const float *x, long long ldx, float *dw, float *db, float *ws, int R, int M,
                          int N, void *stream_) {
  if (!dy || !x || !dw || !ws) return -1;
  if (!mono_linear_wgrad_workspace(R, M, N) || ldy < M || ldx < N || ldy % 4 || ldx % 4 || ((size_t)dy | (size_t)x | (size_t)dw | (size_t)ws) % 16 ||
      (db && (size_t)db % 16))
    return -2;
  hipStream_t st = (hipStream_t)stream_;
  const int S = mono::linear_wgrad_splits(R, M, N);
  // whole stages per split (only the matrix's last rows leave a tail: a tail is a chain of exposed memory latencies)
  const int rows_per_split = ((R + S - 1) / S + mono::kWgStageRows - 1) / mono::kWgStageRows * mono::kWgStageRows;
  const int tiles = (M / 64) * (N / 64);
  mono::linear_wgrad_partial_kernel<<<tiles * S, 256, 0, st>>>(dy, ldy, x, ldx, ws, R, M, N, S, rows_per_split);
  const long long MN = (long long)M * N;
  mono::linear_wgrad_reduce_kernel<<<(int)(((MN + M) / 4 + 255) / 256), 256, 0, st>>>(ws, dw, db, MN, M, S);
  return (int)hipGetLastError();
}

// out[c] = sum_r g[r][c] for any C <= 1024 (no alignment or width requirement); partials: mono_colsum_any_blocks(rows) * C floats.
int mono_colsum_any_blocks(long long rows) {
  const long long b = (rows + 127) / 128;
  return (int)(b < 1 ? 1 : (b > 512 ? 512 : b));
}
int mono_colsum_any_f32(const float *g, float *out, float *partials, long long rows, int C, void *stream_) {
  if (!g || !out || !partials) return -1;
  if (rows <= 0 || C <= 0 || C > 1024) return -2;
  const int grid = mono_colsum_any_blocks(rows);
  const int rpb = (int)((rows + grid - 1) / grid);
  hipStream_t st = (hipStream_t)stream_;
  mono::colsum_scalar_kernel<<<grid, 256, 0, st>>>(g, partials, rows, C, rpb);
  mono::partial_sum_scalar_kernel<<<(C + 255) / 256, 256, 0, st>>>(partials, out, grid, C);
  return (int)hipGetLastError();
}

// out[c] = sum_k x[k][c] for a contiguous [n, C] stack of n slices (split-K partial products: n = 16..64, C = out * in of a weight
// matrix): the generic PyTorch reduction over dim 0 runs this shape at 0.24 TB/s (70 us for 64 x 256 x 256).  C % 4 == 0.
int mono_sum_slices_f32(const float *x, float *out, int n, long long C, void *stream_) {
  if (!x || !out) return -1;
  if (n <= 0 || C <= 0 || (C & 3) || C > (1ll << 30) || ((uintptr_t)x & 15) || ((uintptr_t)out & 15)) return -2;
  mono::partial_sum_kernel<<<(unsigned)((C / 4 + 63) / 64), 1024, 0, (hipStream_t)stream_>>>(x, out, n, (int)C);
  return (int)hipGetLastError();
}

// Column sums of a [batch, rows, C] view whose batches are batch_stride floats apart (rows of a batch contiguous):
// out[c] = sum_{b, r} g[b * batch_stride + r * C + c].  partials: mono_reduce_blocks(batch * rows) * C floats.
int mono_colsum_strided_f32(const float *g, float *out, float *partials, int batch, long long rows, long long batch_stride,
                            int C, void *stream_) {
  if (!g || !out || !partials) return -1;
  if (batch <= 0 || rows <= 0 || C <= 0 || C % 4 || C > 512 || batch_stride % 4) return -2;
  hipStream_t st = (hipStream_t)stream_;
  const long long total = (long long)batch * rows;
  const int grid = mono_reduce_blocks(total);
  const int rpb = (int)((total + grid - 1) / grid + 63) / 64 * 64;
  if (C <= 256) mono::colsum_kernel<1><<<grid, 256, 0, st>>>(g, partials, total, C, rpb, rows, batch_stride);
  else mono::colsum_kernel<2><<<grid, 256, 0, st>>>(g, partials, total, C, rpb, rows, batch_stride);
  mono::partial_sum_kernel<<<(C / 4 + 63) / 64, 1024, 0, st>>>(partials, out, grid, C);
  return (int)hipGetLastError();
}

// Per-level column sums of a [batch, S, C] tensor: out[l][c] = sum over batches and rows [bounds[2 l], bounds[2 l + 1]) of g; n_levels <= 8,
// C % 4 == 0, C <= 512.  mono_colsum_levels_blocks: partial rows (of C floats) the call needs, 0 for unusable bounds.
static int colsum_levels_plan(mono::LevelPlan &p, int batch, long long S, int n_levels, const int *bounds) {
  if (batch <= 0 || n_levels <= 0 || n_levels > 8 || !bounds) return 0;
  p.n = n_levels;
  p.blk0[0] = 0;
  for (int l = 0; l < n_levels; ++l) {
    const long long a = bounds[2 * l], b = bounds[2 * l + 1];
    if (a < 0 || b <= a || b > S) return 0;
    const long long total = (long long)batch * (b - a);
    const int grid = mono_reduce_blocks(total);
    p.row0[l] = (int)a; p.rows[l] = (int)(b - a);
    p.rpb[l] = (int)((total + grid - 1) / grid + 63) / 64 * 64;
    p.blk0[l + 1] = p.blk0[l] + grid;
  }
  return p.blk0[n_levels];
}
int mono_colsum_levels_blocks(int batch, long long S, int n_levels, const int *bounds) {
  mono::LevelPlan p;
  return colsum_levels_plan(p, batch, S, n_levels, bounds);
}
int mono_colsum_levels_f32(const float *g, float *out, float *partials, int batch, long long S, int C, int n_levels, const int *bounds,
                           int with_total, void *stream_) {
  if (!g || !out || !partials) return -1;
  mono::LevelPlan p;
  const int grid = colsum_levels_plan(p, batch, S, n_levels, bounds);
  if (!grid || C <= 0 || C % 4 || C > 512) return -2;
  hipStream_t st = (hipStream_t)stream_;
  if (C <= 256) mono::colsum_levels_kernel<1><<<grid, 256, 0, st>>>(g, partials, p, batch, S * C, C);
  else mono::colsum_levels_kernel<2><<<grid, 256, 0, st>>>(g, partials, p, batch, S * C, C);
  mono::partial_sum_levels_kernel<<<dim3((C / 4 + 63) / 64, n_levels + (with_total ? 1 : 0)), 1024, 0, st>>>(partials, out, p, C);
  return (int)hipGetLastError();
}

// GroupNorm(G = 32, C = 256) (+ ReLU) on channels-last x [B, HW, 256] (+ pre_bias[256] when not NULL: the preceding
// convolution's bias).  stats [B, 32, 2] f64 must be zero on entry; mean_rstd [B, 32, 2] f32 is written for the backward.
int mono_groupnorm_nhwc_fwd_f32(const float *x, const float *pre_bias, const float *gamma, const float *beta, float *y,
                                double *stats, float *mean_rstd, int B, int HW, int C, int G, float eps, int relu,
                                void *stream_) {
  if (!x || !gamma || !beta || !y || !stats || !mean_rstd) return -1;
  if (B <= 0 || HW <= 0 || C != mono::kGnC || G != mono::kGnG || B > 65535) return -2;
  hipStream_t st = (hipStream_t)stream_;
  const dim3 grid((HW + mono::kGnPix - 1) / mono::kGnPix, B);
  mono::gn_stats_kernel<<<grid, 256, 0, st>>>(x, pre_bias, stats, HW);
  if (relu) mono::gn_apply_kernel<true><<<grid, 256, 0, st>>>(x, pre_bias, stats, gamma, beta, y, mean_rstd, HW, eps);
  else mono::gn_apply_kernel<false><<<grid, 256, 0, st>>>(x, pre_bias, stats, gamma, beta, y, mean_rstd, HW, eps);
  return (int)hipGetLastError();
}

// Rows of gbias scratch the backward needs: one per workgroup.
int mono_groupnorm_blocks(int B, int HW) { return B * ((HW + mono::kGnPix - 1) / mono::kGnPix); }

// part [B, 256, 2] f64 must be zero on entry; on return part[b][c] = {sum gy' xhat, sum gy'} (ggamma / gbeta are its
// sums over b).  y (the forward output) is read only when relu != 0.  With pre_bias: gbias [256] receives the bias
// gradient (= sum of gx over batch and pixels), gbias_partials is scratch of mono_groupnorm_blocks(B, HW) * 256 floats.
int mono_groupnorm_nhwc_bwd_f32(const float *gy, const float *x, const float *pre_bias, const float *y,
                                const float *mean_rstd, const float *gamma, float *gx, double *part, float *gbias,
                                float *gbias_partials, float *ggamma_gbeta, int B, int HW, int C, int G, int relu, void *stream_) {
  if (!gy || !x || !mean_rstd || !gamma || !gx || !part || (relu && !y) || (pre_bias && (!gbias || !gbias_partials))) return -1;
  if (B <= 0 || HW <= 0 || C != mono::kGnC || G != mono::kGnG || B > 65535) return -2;
  hipStream_t st = (hipStream_t)stream_;
  const dim3 grid((HW + mono::kGnPix - 1) / mono::kGnPix, B);
  float *gp = pre_bias ? gbias_partials : nullptr;
  if (relu) {
    mono::gn_bwd_stats_kernel<true><<<grid, 256, 0, st>>>(gy, x, pre_bias, y, mean_rstd, part, HW);
    mono::gn_bwd_apply_kernel<true><<<grid, 256, 0, st>>>(gy, x, pre_bias, y, mean_rstd, gamma, part, gx, gp, HW, ggamma_gbeta);
  } else {
    mono::gn_bwd_stats_kernel<false><<<grid, 256, 0, st>>>(gy, x, pre_bias, y, mean_rstd, part, HW);
    mono::gn_bwd_apply_kernel<false><<<grid, 256, 0, st>>>(gy, x, pre_bias, y, mean_rstd, gamma, part, gx, gp, HW, ggamma_gbeta);
  }
  if (pre_bias) mono::partial_sum_kernel<<<1, 1024, 0, st>>>(gbias_partials, gbias, (int)(grid.x * grid.y), mono::kGnC);
  return (int)hipGetLastError();
}


// ---- DDN depth-map loss (ddn_loss.hip) ---------------------------------------------------------------------------------
int mono_ddn_loss_blocks(int B, int H, int W) { return (B * H * W * mono::kDdnLanes + 255) / 256; }

static mono::DdnParams ddn_params(int B, int C, int H, int W, int N, long long sb, long long sc, long long sp, float alpha,
                                  float gamma, float fg_weight, float bg_weight, float depth_min, float depth_max) {
  mono::DdnParams p;
  p.B = B; p.C = C; p.H = H; p.W = W; p.N = N;
  p.sb = sb; p.sc = sc; p.sp = sp;
  p.alpha = alpha; p.gamma = gamma; p.fg_weight = fg_weight; p.bg_weight = bg_weight;
  p.depth_min = depth_min; p.depth_max = depth_max; p.eps = 1e-6f;
  return p;
}

int mono_ddn_loss_fwd_f32(const float *logits, const float *boxes, const float *depth, const unsigned char *valid, float *partial,
                          int B, int C, int H, int W, int N, long long sb, long long sc, long long sp, float alpha, float gamma,
                          float fg_weight, float bg_weight, float depth_min, float depth_max, void *stream) {
  if (!logits || !boxes || !depth || !valid || !partial) return -1;
  if (B <= 0 || C <= 1 || H <= 0 || W <= 0 || N <= 0) return -2;
  const mono::DdnParams p = ddn_params(B, C, H, W, N, sb, sc, sp, alpha, gamma, fg_weight, bg_weight, depth_min, depth_max);
  mono::ddn_loss_fwd_kernel<<<mono_ddn_loss_blocks(B, H, W), 256, 0, (hipStream_t)stream>>>(logits, boxes, depth, valid, partial, p);
  return (int)hipGetLastError();
}

int mono_ddn_loss_bwd_f32(const float *logits, const float *boxes, const float *depth, const unsigned char *valid,
                          const float *grad_total, float *grad_logits, int B, int C, int H, int W, int N, long long sb, long long sc,
                          long long sp, float alpha, float gamma, float fg_weight, float bg_weight, float depth_min, float depth_max,
                          void *stream) {
  if (!logits || !boxes || !depth || !valid || !grad_total || !grad_logits) return -1;
  if (B <= 0 || C <= 1 || H <= 0 || W <= 0 || N <= 0) return -2;
  const mono::DdnParams p = ddn_params(B, C, H, W, N, sb, sc, sp, alpha, gamma, fg_weight, bg_weight, depth_min, depth_max);
  mono::ddn_loss_bwd_kernel<<<mono_ddn_loss_blocks(B, H, W), 256, 0, (hipStream_t)stream>>>(logits, boxes, depth, valid, grad_total,
                                                                                           grad_logits, p);
  return (int)hipGetLastError();
}


// ---- expected depth over the bin distribution (ddn_loss.hip) -------------------------------------------------------------
int mono_depth_expect_fwd_f32(const float *logits, const float *values, float *out, int B, int C, int H, int W, long long sb,
                              long long sc, long long sp, void *stream) {
  if (!logits || !values || !out) return -1;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return -2;
  const int n = B * H * W;
  mono::depth_expect_fwd_kernel<<<(n * mono::kDdnLanes + 255) / 256, 256, 0, (hipStream_t)stream>>>(logits, values, out, n, H * W, C, sb, sc, sp);
  return (int)hipGetLastError();
}

int mono_depth_expect_bwd_f32(const float *logits, const float *values, const float *expect, const float *grad_out, float *grad_logits,
                              int B, int C, int H, int W, long long sb, long long sc, long long sp, void *stream) {
  if (!logits || !values || !expect || !grad_out || !grad_logits) return -1;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return -2;
  const int n = B * H * W;
  mono::depth_expect_bwd_kernel<<<(n * mono::kDdnLanes + 255) / 256, 256, 0, (hipStream_t)stream>>>(logits, values, expect, grad_out, grad_logits, n,
                                                                               H * W, C, sb, sc, sp);
  return (int)hipGetLastError();
}


// ---- classification side of the criterion (matched_losses.hip): focal sums, class error, cardinality error -----------------
int mono_focal_fwd_f32(const float *logits, const long long *idx, const long long *labels, const float *sizes, float *out, int NL,
                       int B, int Q, int C, int K, float alpha, float gamma, void *stream) {
  if (!logits || !out || !sizes || (K > 0 && (!idx || !labels))) return -1;
  if (NL <= 0 || B <= 0 || Q <= 0 || C <= 0 || C > 255 || K < 0 || B > 256 || (long long)B * Q > mono::kFocalMaxCells) return -2;
  const mono::FocalArgs a{logits, idx, labels, sizes, NL, B, Q, C, K, alpha, gamma};
  mono::focal_fwd_kernel<<<NL, mono::kFocalThreads, 0, (hipStream_t)stream>>>(a, out);
  return (int)hipGetLastError();
}

int mono_focal_bwd_f32(const float *logits, const long long *idx, const long long *labels, const float *grad_out, float *grad_logits,
                       int NL, int B, int Q, int C, int K, float alpha, float gamma, void *stream) {
  if (!logits || !grad_out || !grad_logits || (K > 0 && (!idx || !labels))) return -1;
  if (NL <= 0 || B <= 0 || Q <= 0 || C <= 0 || C > 255 || K < 0 || (long long)B * Q > mono::kFocalMaxCells) return -2;
  const mono::FocalArgs a{logits, idx, labels, nullptr, NL, B, Q, C, K, alpha, gamma};
  mono::focal_bwd_kernel<<<NL, mono::kFocalThreads, 0, (hipStream_t)stream>>>(a, grad_out, grad_logits);
  return (int)hipGetLastError();
}


// ---- the matcher's per-image cost blocks (matched_losses.hip) ------------------------------------------------------------------
int mono_match_cost_f32(const float *logits, const float *boxes, const long long *labels, const float *tboxes, const long long *cols,
                        float *out, int NL, int B, int Q, int C, int N, float w_class, float w_3d, float w_bbox, float w_giou,
                        void *stream) {
  if (!logits || !boxes || !labels || !tboxes || !cols || !out) return -1;
  if (NL <= 0 || B <= 0 || Q <= 0 || C <= 0 || N <= 0) return -2;
  const long long n = (long long)NL * B * Q * N;
  if (n > (1ll << 31)) return -2;
  const mono::CostArgs a{logits, boxes, labels, tboxes, cols, NL, B, Q, C, N, w_class, w_3d, w_bbox, w_giou};
  mono::match_cost_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(a, out);
  return (int)hipGetLastError();
}


// ---- per-level tail of the detection heads (head_tail.hip) --------------------------------------------------------------------
int mono_head_tail_fwd_f32(const float *tmp, const float *size3d, const float *depth_reg, const float *wdepth, const float *fu,
                           const float *img_h, float *coords, float *depth_ave, int B, int Q, int H, int W, const float *ref,
                           int ref_dim, void *stream) {
  if (!tmp || !size3d || !depth_reg || !wdepth || !fu || !img_h || !coords || !depth_ave) return -1;
  if (B <= 0 || Q <= 0 || H <= 0 || W <= 0 || (ref && (ref_dim < 1 || ref_dim > 6))) return -2;
  const mono::HeadTailArgs a{tmp, size3d, depth_reg, wdepth, fu, img_h, B, Q, H, W, ref, ref_dim};
  mono::head_tail_fwd_kernel<<<(B * Q + 255) / 256, 256, 0, (hipStream_t)stream>>>(a, coords, depth_ave);
  return (int)hipGetLastError();
}

int mono_head_tail_bwd_f32(const float *tmp, const float *size3d, const float *depth_reg, const float *wdepth, const float *fu,
                           const float *img_h, const float *g_coords, const float *g_depth_ave, float *g_tmp, float *g_size3d,
                           float *g_depth_reg, float *g_wdepth, int B, int Q, int H, int W, const float *ref, int ref_dim,
                           void *stream) {
  if (!tmp || !size3d || !depth_reg || !wdepth || !fu || !img_h || !g_tmp || !g_size3d || !g_depth_reg || !g_wdepth) return -1;
  if (B <= 0 || Q <= 0 || H <= 0 || W <= 0 || (ref && (ref_dim < 1 || ref_dim > 6))) return -2;
  const mono::HeadTailArgs a{tmp, size3d, depth_reg, wdepth, fu, img_h, B, Q, H, W, ref, ref_dim};
  mono::head_tail_bwd_kernel<<<(B * Q + 255) / 256, 256, 0, (hipStream_t)stream>>>(a, g_coords, g_depth_ave, g_tmp, g_size3d,
                                                                                   g_depth_reg, g_wdepth);
  return (int)hipGetLastError();
}

int mono_refine_reference_f32(const float *tmp, const float *ref, float *out, int n, int ref_dim, void *stream) {
  if (!tmp || !ref || !out) return -1;
  if (n <= 0 || ref_dim < 1 || ref_dim > 6 || (long long)n * 6 > (1ll << 30)) return -2;
  mono::refine_reference_kernel<<<(n * 6 + 255) / 256, 256, 0, (hipStream_t)stream>>>(tmp, ref, out, n, ref_dim);
  return (int)hipGetLastError();
}

/* The matcher's assignments on the device (csrc/lsap_device.hip): see include/monosowa_pointwise.h. */
int mono_lsap_match_flat_f32(const float *cost, int NL, int B, int Q, int T, int G, const int *meta, long long *out_idx, long long K,
                             int *status, void *stream) {
  if (!cost || !meta || !out_idx || !status) return -1;
  if (NL <= 0 || B <= 0 || Q <= 0 || T <= 0 || G <= 0 || Q % G != 0 || K < 0) return -2;
  if (Q / G > lsapd::kMaxDim || (long long)NL * B * G > (1ll << 30)) return -3;          // the host solver keeps such shapes
  if (K == 0) return 0;
  lsapd::match_flat_kernel<<<NL * B * G, 64, 0, (hipStream_t)stream>>>(cost, NL, B, Q, T, G, meta, out_idx, K, status);
  return (int)hipGetLastError();
}

}  // extern "C"
