// The tail of a FROZEN ResNet bottleneck as one kernel (reference backbone.py:72-74 freezes conv1 / layer1; torchvision Bottleneck:
// out = relu(bn3(conv3(relu(bn2(conv2_out)))) + identity)):
//
//     Y[m, n] = relu( sum_k relu(X[m, k] + b_in[k]) * W[k, n]  +  b_out[n]  +  R[m, n] )        K = 64, N = 256, channels-last rows
//
// X = the 3 x 3 convolution's raw output (BN scale folded into its weights, shift b_in not yet added), W = conv3's weights with
// bn3's scale folded in, b_out = bn3's shift (+ the downsample norm's), R = the block's identity.  Replaces three passes over the
// network's largest activations -- the bias + ReLU pass over X (read + write), the library's 1 x 1 convolution (read X, write Y) and the
// shift + residual + ReLU pass (read Y, read R, write Y): 2.4 GB at [16, 96, 320] -- by one: read X and R, write Y (1.13 GB).  Forward
// only: nothing below layer 2 receives a gradient.
//
// A wave owns 32 pixels (rows of X) at a time and all 256 output channels: its lanes keep relu(X + b_in) of their pixel in registers
// (lane = (pixel m, half kh): the 32 contiguous floats X[m][32 kh ..], i.e. MFMA step s contracts k = s and k = 32 + s), W sits in LDS
// for the whole workgroup (64 KB), and the product is evaluated TRANSPOSED per 32-channel block -- D[n][m] = sum_k W[k][n] X[m][k],
// v_mfma_f32_32x32x2_f32, exact f32 -- so that a lane ends up with 4 CONSECUTIVE channels of its pixel per accumulator quad: the
// epilogue reads R and writes Y as float4.  One accumulator (16 registers) is live at a time (142 VGPRs).
#pragma once
#include <hip/hip_runtime.h>

namespace mono {

typedef float c1_f32x16 __attribute__((ext_vector_type(16)));
constexpr int kC1K = 64, kC1N = 256, kC1Threads = 512;

__global__ __launch_bounds__(kC1Threads) void conv1x1_tail_kernel(const float *__restrict__ x, const float *__restrict__ b_in,
                                                                  const float *__restrict__ w, const float *__restrict__ b_out,
                                                                  const float *__restrict__ res, float *__restrict__ y, long long M) {
  __shared__ float Ws[kC1K * kC1N];                      // W[k][n]
  __shared__ float Bi[kC1K], Bo[kC1N];
  for (int i = threadIdx.x; i < kC1K * kC1N / 4; i += kC1Threads)
    reinterpret_cast<float4 *>(Ws)[i] = reinterpret_cast<const float4 *>(w)[i];
  if (threadIdx.x < kC1K) Bi[threadIdx.x] = b_in[threadIdx.x];
  if (threadIdx.x < kC1N) Bo[threadIdx.x] = b_out[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m_l = lane & 31, kh = lane >> 5;
  const long long n_strips = (M + 31) / 32, strip_stride = (long long)gridDim.x * (kC1Threads / 64);
  for (long long strip = (long long)blockIdx.x * (kC1Threads / 64) + wave; strip < n_strips; strip += strip_stride) {
    const long long m = strip * 32 + m_l;
    const bool live = m < M;
    const long long mc = live ? m : M - 1;
    // this lane's 32 inputs: relu(X[m][32 kh + s] + b_in[32 kh + s])
    float xr[32];
    const float *xp = x + mc * kC1K + 32 * kh;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 v = *reinterpret_cast<const float4 *>(xp + 4 * i), b = *reinterpret_cast<const float4 *>(&Bi[32 * kh + 4 * i]);
      xr[4 * i] = fmaxf(v.x + b.x, 0.f); xr[4 * i + 1] = fmaxf(v.y + b.y, 0.f);
      xr[4 * i + 2] = fmaxf(v.z + b.z, 0.f); xr[4 * i + 3] = fmaxf(v.w + b.w, 0.f);
    }
    const float *rp = res + mc * kC1N + 4 * kh;
    float *yp = y + mc * kC1N + 4 * kh;
    // the identity's 16 values of a channel block are requested one block AHEAD: the wait in front of their use then leaves the
    // previous block's stores in flight (loads and stores return through one in-order counter); 304 -> 276 us at [16, 96, 320]
    float4 r_next[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) r_next[g] = *reinterpret_cast<const float4 *>(rp + 8 * g);
#pragma unroll 1
    for (int nb = 0; nb < kC1N / 32; ++nb) {
      float4 r4[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) r4[g] = r_next[g];
      if (nb + 1 < kC1N / 32) {
#pragma unroll
        for (int g = 0; g < 4; ++g) r_next[g] = *reinterpret_cast<const float4 *>(rp + (nb + 1) * 32 + 8 * g);
      }
      c1_f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const float *wp = &Ws[(32 * kh) * kC1N + nb * 32 + m_l];          // A operand: W[k = 32 kh + s][n = 32 nb + lane % 32]
#pragma unroll
      for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wp[s * kC1N], xr[s], acc, 0, 0, 0);
      // acc[4 g + t] = D[n = 32 nb + 8 g + 4 kh + t][m]
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bo = *reinterpret_cast<const float4 *>(&Bo[nb * 32 + 8 * g + 4 * kh]);
        const float4 o = make_float4(fmaxf(acc[4 * g] + bo.x + r4[g].x, 0.f), fmaxf(acc[4 * g + 1] + bo.y + r4[g].y, 0.f),
                                     fmaxf(acc[4 * g + 2] + bo.z + r4[g].z, 0.f), fmaxf(acc[4 * g + 3] + bo.w + r4[g].w, 0.f));
        if (live) *reinterpret_cast<float4 *>(yp + nb * 32 + 8 * g) = o;
      }
    }
  }
}

// The tail of the FIRST frozen bottleneck of a stage, whose identity is a 1 x 1 downsample convolution of the block's input:
//     Y[m, n] = relu( sum_k relu(X[m, k] + b_in[k]) W[k, n]  +  sum_k X0[m, k] Wd[k, n]  +  b_out[n] ),      K = K0 = 64, N = 256
// -- the downsample product joins the accumulator instead of being written (503 MB at [16, 96, 320]) by a convolution of its own and read
// back as the identity.  Both weight matrices sit in LDS (128 KB: one 16-wave workgroup per CU).
constexpr int kC1Threads2 = 1024;
__global__ __launch_bounds__(kC1Threads2) void conv1x1_tail_ds_kernel(const float *__restrict__ x, const float *__restrict__ b_in,
                                                                      const float *__restrict__ w, const float *__restrict__ x0,
                                                                      const float *__restrict__ wd, const float *__restrict__ b_out,
                                                                      float *__restrict__ y, long long M) {
  __shared__ float Ws[2 * kC1K * kC1N];                  // W[k][n], then Wd[k][n]
  __shared__ float Bi[kC1K], Bo[kC1N];
  for (int i = threadIdx.x; i < kC1K * kC1N / 4; i += kC1Threads2) {
    reinterpret_cast<float4 *>(Ws)[i] = reinterpret_cast<const float4 *>(w)[i];
    reinterpret_cast<float4 *>(Ws + kC1K * kC1N)[i] = reinterpret_cast<const float4 *>(wd)[i];
  }
  if (threadIdx.x < kC1K) Bi[threadIdx.x] = b_in[threadIdx.x];
  if (threadIdx.x < kC1N) Bo[threadIdx.x] = b_out[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m_l = lane & 31, kh = lane >> 5;
  const long long n_strips = (M + 31) / 32, strip_stride = (long long)gridDim.x * (kC1Threads2 / 64);
  for (long long strip = (long long)blockIdx.x * (kC1Threads2 / 64) + wave; strip < n_strips; strip += strip_stride) {
    const long long m = strip * 32 + m_l;
    const bool live = m < M;
    const long long mc = live ? m : M - 1;
    float xr[32], x0r[32];
    const float *xp = x + mc * kC1K + 32 * kh, *x0p = x0 + mc * kC1K + 32 * kh;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 v = *reinterpret_cast<const float4 *>(xp + 4 * i), b = *reinterpret_cast<const float4 *>(&Bi[32 * kh + 4 * i]);
      const float4 u = *reinterpret_cast<const float4 *>(x0p + 4 * i);
      xr[4 * i] = fmaxf(v.x + b.x, 0.f); xr[4 * i + 1] = fmaxf(v.y + b.y, 0.f);
      xr[4 * i + 2] = fmaxf(v.z + b.z, 0.f); xr[4 * i + 3] = fmaxf(v.w + b.w, 0.f);
      x0r[4 * i] = u.x; x0r[4 * i + 1] = u.y; x0r[4 * i + 2] = u.z; x0r[4 * i + 3] = u.w;
    }
    float *yp = y + mc * kC1N + 4 * kh;
#pragma unroll 1
    for (int nb = 0; nb < kC1N / 32; ++nb) {
      c1_f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const float *wp = &Ws[(32 * kh) * kC1N + nb * 32 + m_l], *wdp = wp + kC1K * kC1N;
#pragma unroll
      for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wp[s * kC1N], xr[s], acc, 0, 0, 0);
#pragma unroll
      for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wdp[s * kC1N], x0r[s], acc, 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bo = *reinterpret_cast<const float4 *>(&Bo[nb * 32 + 8 * g + 4 * kh]);
        const float4 o = make_float4(fmaxf(acc[4 * g] + bo.x, 0.f), fmaxf(acc[4 * g + 1] + bo.y, 0.f),
                                     fmaxf(acc[4 * g + 2] + bo.z, 0.f), fmaxf(acc[4 * g + 3] + bo.w, 0.f));
        if (live) *reinterpret_cast<float4 *>(yp + nb * 32 + 8 * g) = o;
      }
    }
  }
}

// The head of a FROZEN bottleneck: H[m, n] = relu( sum_k X[m, k] W[k, n] + b[n] ),  K = 256 (64 for the first block), N = 64 --
// conv1 + bn1 + ReLU in one pass over the block's input (the library's 1 x 1 convolution plus a bias + ReLU pass over its output
// before).  Same mapping as the tail kernel, with the contraction in chunks of 64 (a lane holds 32 inputs of its pixel at a time:
// k = (K / 2) kh + 32 chunk + s) and both 32-channel accumulators alive across the chunks.
template <int K>
__global__ __launch_bounds__(kC1Threads) void conv1x1_head_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                                  const float *__restrict__ b_out, float *__restrict__ y, long long M) {
  constexpr int N = 64;
  __shared__ float Ws[K * N];                            // W[k][n]
  __shared__ float Bo[N];
  for (int i = threadIdx.x; i < K * N / 4; i += kC1Threads)
    reinterpret_cast<float4 *>(Ws)[i] = reinterpret_cast<const float4 *>(w)[i];
  if (threadIdx.x < N) Bo[threadIdx.x] = b_out[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m_l = lane & 31, kh = lane >> 5;
  const long long n_strips = (M + 31) / 32, strip_stride = (long long)gridDim.x * (kC1Threads / 64);
  for (long long strip = (long long)blockIdx.x * (kC1Threads / 64) + wave; strip < n_strips; strip += strip_stride) {
    const long long m = strip * 32 + m_l;
    const bool live = m < M;
    const long long mc = live ? m : M - 1;
    const float *xp = x + mc * K + (K / 2) * kh;
    c1_f32x16 acc0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    float4 xv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) xv[i] = *reinterpret_cast<const float4 *>(xp + 4 * i);
#pragma unroll 1
    for (int c = 0; c < K / 64; ++c) {
      float xr[32];
#pragma unroll
      for (int i = 0; i < 8; ++i) { xr[4 * i] = xv[i].x; xr[4 * i + 1] = xv[i].y; xr[4 * i + 2] = xv[i].z; xr[4 * i + 3] = xv[i].w; }
      if (c + 1 < K / 64) {                                // the next chunk's inputs: in flight under this chunk's products
#pragma unroll
        for (int i = 0; i < 8; ++i) xv[i] = *reinterpret_cast<const float4 *>(xp + 32 * (c + 1) + 4 * i);
      }
      const float *wp = &Ws[((K / 2) * kh + 32 * c) * N + m_l];         // A operand: W[k][n = lane % 32 (+ 32)]
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wp[s * N], xr[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wp[s * N + 32], xr[s], acc1, 0, 0, 0);
      }
    }
    float *yp = y + mc * N + 4 * kh;
    if (live) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b0 = *reinterpret_cast<const float4 *>(&Bo[8 * g + 4 * kh]), b1 = *reinterpret_cast<const float4 *>(&Bo[32 + 8 * g + 4 * kh]);
        *reinterpret_cast<float4 *>(yp + 8 * g) = make_float4(fmaxf(acc0[4 * g] + b0.x, 0.f), fmaxf(acc0[4 * g + 1] + b0.y, 0.f),
                                                              fmaxf(acc0[4 * g + 2] + b0.z, 0.f), fmaxf(acc0[4 * g + 3] + b0.w, 0.f));
        *reinterpret_cast<float4 *>(yp + 32 + 8 * g) = make_float4(fmaxf(acc1[4 * g] + b1.x, 0.f), fmaxf(acc1[4 * g + 1] + b1.y, 0.f),
                                                                   fmaxf(acc1[4 * g + 2] + b1.z, 0.f), fmaxf(acc1[4 * g + 3] + b1.w, 0.f));
      }
    }
  }
}

}  // namespace mono
