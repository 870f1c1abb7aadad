// Pointwise HIP kernels for the dense (library) part of the path: what PyTorch-ROCm issues as two or three
// elementwise passes after a MIOpen convolution -- bias add, residual add, ReLU -- in one in-place pass.
//   y[r, c] = act(y[r, c] + bias[c] (+ residual[r, c])),  y: [rows, C] contiguous = an NHWC activation
// (frozen batch-norm is folded into the conv weights and `bias`, lib/models/monodetr/backbone.py:28-65).
// HBM-bound: one read (+ one for the residual) and one write per element, float4 per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>

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

inline int grid_for_vec(long long n_vec) {
  long long g = (n_vec + 255) / 256;
  if (g > 256LL * 32) g = 256LL * 32;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace mono

extern "C" {

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

int mono_relu_grad_f32(const float *grad_out, const float *y, float *grad_in, long long n, void *stream_) {
  if (!grad_out || !y || !grad_in) return -1;
  if (n <= 0 || (n & 3) || ((uintptr_t)grad_out & 15) || ((uintptr_t)y & 15) || ((uintptr_t)grad_in & 15)) return -2;
  mono::relu_grad_kernel<<<mono::grid_for_vec(n / 4), 256, 0, (hipStream_t)stream_>>>(grad_out, y, grad_in, n / 4);
  return (int)hipGetLastError();
}

}  // extern "C"
