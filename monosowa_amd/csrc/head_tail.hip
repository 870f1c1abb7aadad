// The per-level tail of MonoDETR's detection heads (monodetr.py:238-263) as one kernel each way: box coordinates
//   coords = sigmoid(tmp)                                            [B, Q, 6]  (cx, cy, l, r, t, b)
// and the fused depth estimate
//   depth_ave[..., 0] = ( (1 / (sigmoid(depth_reg0) + 1e-6) - 1)                      -- regressed (inverse sigmoid coding)
//                       + size3d[..., 0] / max((coords4 + coords5) * img_h, 1) * fu  -- geometric: 3D height / 2D box height
//                       + bilinear(weighted_depth; coords[..., :2]) ) / 3            -- read from the depth map at the projected
//   depth_ave[..., 1] = depth_reg1                                                      3D centre (F.grid_sample, align_corners)
// PyTorch: ~20 launches forward and ~30 backward PER decoder level (every `x[:, :, i]` costs a zero fill, a copy and an add in
// the backward), and those are the first thing the backward enqueues behind the matcher's synchronisation, where the GPU
// waits for the host: 3 x 30 exposed launches.  One thread per (image, query).  The sampling location is detached in the
// reference (no gradient to the centre through the depth map); the depth map itself receives its bilinear weights by atomics.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace mono {

struct HeadTailArgs {
  const float *tmp;         // [B, Q, 6] box logits (reference already added)
  const float *size3d;      // [B, Q, 3]
  const float *depth_reg;   // [B, Q, 2]
  const float *wdepth;      // [B, H, W] weighted depth map
  const float *fu, *img_h;  // [B]
  int B, Q, H, W;
  const float *ref;         // optional [B, Q, ref_dim] (no gradient): the box logits are tmp + inverse_sigmoid(ref) on the first ref_dim
  int ref_dim;              // coordinates (monodetr.py:224-232; util/misc.py inverse_sigmoid, eps 1e-5)
};

__device__ __forceinline__ float inverse_sigmoidf_(float x) {
  x = fminf(fmaxf(x, 0.f), 1.f);
  return logf(fmaxf(x, 1e-5f) / fmaxf(1.f - x, 1e-5f));
}
// box logit k of cell i
__device__ __forceinline__ float head_logit(const HeadTailArgs &a, int i, int k) {
  const float t = a.tmp[i * 6 + k];
  return (a.ref && k < a.ref_dim) ? t + inverse_sigmoidf_(a.ref[i * a.ref_dim + k]) : t;
}

struct HeadTap { int x0, y0; float wnw, wne, wsw, wse; bool nw, ne, sw, se; };

// F.grid_sample(bilinear, align_corners=True, padding zeros) of the location (cx, cy) in [0, 1]^2
__device__ __forceinline__ HeadTap head_tap(float cx, float cy, int H, int W) {
  HeadTap t;
  const float gx = (cx - 0.5f) * 2.f, gy = (cy - 0.5f) * 2.f;
  const float ix = ((gx + 1.f) / 2.f) * (float)(W - 1), iy = ((gy + 1.f) / 2.f) * (float)(H - 1);
  const float fx = floorf(ix), fy = floorf(iy);
  t.x0 = (int)fx; t.y0 = (int)fy;
  const float ex = fx + 1.f, ey = fy + 1.f;
  t.wnw = (ex - ix) * (ey - iy); t.wne = (ix - fx) * (ey - iy); t.wsw = (ex - ix) * (iy - fy); t.wse = (ix - fx) * (iy - fy);
  const bool xl = t.x0 >= 0 && t.x0 < W, xr = t.x0 + 1 >= 0 && t.x0 + 1 < W, yt = t.y0 >= 0 && t.y0 < H, yb = t.y0 + 1 >= 0 && t.y0 + 1 < H;
  t.nw = xl && yt; t.ne = xr && yt; t.sw = xl && yb; t.se = xr && yb;
  return t;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void head_tail_fwd_kernel(const HeadTailArgs a, float *__restrict__ coords, float *__restrict__ depth_ave) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.B * a.Q) return;
  const int b = i / a.Q;
  float oc[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) { oc[k] = sigmoidf_(head_logit(a, i, k)); coords[i * 6 + k] = oc[k]; }
  const float bh = fmaxf((oc[4] + oc[5]) * a.img_h[b], 1.0f);
  const float geo = a.size3d[i * 3] / bh * a.fu[b];
  const float s = sigmoidf_(a.depth_reg[i * 2]);
  const float inv = 1.f / (s + 1e-6f) - 1.f;
  const HeadTap t = head_tap(oc[0], oc[1], a.H, a.W);
  const float *m = a.wdepth + (long long)b * a.H * a.W + t.y0 * a.W + t.x0;
  float dm = 0.f;
  if (t.nw) dm += m[0] * t.wnw;
  if (t.ne) dm += m[1] * t.wne;
  if (t.sw) dm += m[a.W] * t.wsw;
  if (t.se) dm += m[a.W + 1] * t.wse;
  depth_ave[i * 2] = ((inv + geo) + dm) / 3.f;
  depth_ave[i * 2 + 1] = a.depth_reg[i * 2 + 1];
}

// g_wdepth must be ZERO on entry.
__global__ __launch_bounds__(256) void head_tail_bwd_kernel(const HeadTailArgs a, const float *__restrict__ g_coords,
                                                            const float *__restrict__ g_dave, float *__restrict__ g_tmp,
                                                            float *__restrict__ g_size3d, float *__restrict__ g_dreg,
                                                            float *__restrict__ g_wdepth) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.B * a.Q) return;
  const int b = i / a.Q;
  float oc[6], goc[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) { oc[k] = sigmoidf_(head_logit(a, i, k)); goc[k] = g_coords ? g_coords[i * 6 + k] : 0.f; }
  const float g0 = g_dave ? g_dave[i * 2] / 3.f : 0.f, g1 = g_dave ? g_dave[i * 2 + 1] : 0.f;
  const float h = (oc[4] + oc[5]) * a.img_h[b], bh = fmaxf(h, 1.0f), fu = a.fu[b], s3 = a.size3d[i * 3];
  g_size3d[i * 3] = g0 * fu / bh;
  g_size3d[i * 3 + 1] = 0.f;
  g_size3d[i * 3 + 2] = 0.f;
  const float g_h = h >= 1.0f ? g0 * (-(s3 / bh) / bh * fu) : 0.f;        // d (s3 / bh * fu) / d bh
  goc[4] += g_h * a.img_h[b];
  goc[5] += g_h * a.img_h[b];
  const float s = sigmoidf_(a.depth_reg[i * 2]), se = s + 1e-6f;
  g_dreg[i * 2] = g0 * (-1.f / (se * se)) * (s * (1.f - s));
  g_dreg[i * 2 + 1] = g1;
  if (g0 != 0.f) {
    const HeadTap t = head_tap(oc[0], oc[1], a.H, a.W);
    float *m = g_wdepth + (long long)b * a.H * a.W + t.y0 * a.W + t.x0;
    if (t.nw) atomicAdd(m, g0 * t.wnw);
    if (t.ne) atomicAdd(m + 1, g0 * t.wne);
    if (t.sw) atomicAdd(m + a.W, g0 * t.wsw);
    if (t.se) atomicAdd(m + a.W + 1, g0 * t.wse);
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) g_tmp[i * 6 + k] = goc[k] * (oc[k] * (1.f - oc[k]));
}

// The decoder's iterative reference refinement (depthaware_transformer.py:602-613), detached in the reference:
// out [n, 6] = sigmoid(tmp + inverse_sigmoid(ref) on the first ref_dim coordinates).
__global__ __launch_bounds__(256) void refine_reference_kernel(const float *__restrict__ tmp, const float *__restrict__ ref,
                                                               float *__restrict__ out, int n, int ref_dim) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n * 6) return;
  const int cell = i / 6, k = i - cell * 6;
  const float t = tmp[i];
  out[i] = sigmoidf_(k < ref_dim ? t + inverse_sigmoidf_(ref[cell * ref_dim + k]) : t);
}

}  // namespace mono
