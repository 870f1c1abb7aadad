// Shared device helpers for the MSDA kernels (gfx950 / CDNA4, wave64).
//
// The sampling rule restated here is the reference's (ops/src/cuda/ms_deform_im2col_cuda.cuh):
//   h_im = loc_y * H - 0.5, w_im = loc_x * W - 0.5                              (cuh:271-272)
//   a point contributes iff h_im > -1 && w_im > -1 && h_im < H && w_im < W       (cuh:274)
//   corners (h_low,w_low) .. (h_low+1,w_low+1), each dropped when outside        (cuh:56-79)
//   corner weights hh*hw, hh*lw, lh*hw, lh*lw with lh = h_im - h_low, hh = 1-lh  (cuh:44-46,81)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace msda {

// One sampling point resolved against one level: clamped corner coordinates (always
// addressable), corner weights with dropped corners zeroed, and the fractional parts.
template <typename T>
struct Tap {
  int y0, y1, x0, x1;   // clamped rows / columns of the 2x2 footprint
  T w1, w2, w3, w4;     // (y0,x0) (y0,x1) (y1,x0) (y1,x1); 0 where the corner is outside
  T lh, lw, hh, hw;     // fractional offsets (for the location gradient)
  bool t, b, l, r;      // which rows / columns are inside
  bool valid;           // the cuh:274 test
};

// Product rounded once, then the subtraction: the reference evaluates `loc * size - 0.5` with a
// double literal, so the float product is rounded before the subtraction (no fused multiply-add).
__device__ __forceinline__ float scale_loc(float loc, int size) {
  return __fsub_rn(__fmul_rn(loc, (float)size), 0.5f);
}
__device__ __forceinline__ double scale_loc(double loc, int size) {
  return __dsub_rn(__dmul_rn(loc, (double)size), 0.5);
}

template <typename T>
__device__ __forceinline__ Tap<T> make_tap(T loc_x, T loc_y, int H, int W) {
  Tap<T> tp;
  const T h_im = scale_loc(loc_y, H);
  const T w_im = scale_loc(loc_x, W);
  tp.valid = (h_im > (T)-1) && (w_im > (T)-1) && (h_im < (T)H) && (w_im < (T)W);
  // Invalid points (including NaN / huge coordinates) are parked at 0 so that the integer
  // conversion below is defined; their weights are zeroed.
  const T hs = tp.valid ? h_im : (T)0;
  const T ws = tp.valid ? w_im : (T)0;
  const T hf = floor(hs), wf = floor(ws);
  const int h_low = (int)hf, w_low = (int)wf;
  tp.lh = hs - hf;
  tp.lw = ws - wf;
  tp.hh = (T)1 - tp.lh;
  tp.hw = (T)1 - tp.lw;
  tp.t = tp.valid && (h_low >= 0);
  tp.b = tp.valid && (h_low + 1 <= H - 1);
  tp.l = (w_low >= 0);
  tp.r = (w_low + 1 <= W - 1);
  tp.w1 = (tp.t && tp.l) ? tp.hh * tp.hw : (T)0;
  tp.w2 = (tp.t && tp.r) ? tp.hh * tp.lw : (T)0;
  tp.w3 = (tp.b && tp.l) ? tp.lh * tp.hw : (T)0;
  tp.w4 = (tp.b && tp.r) ? tp.lh * tp.lw : (T)0;
  tp.y0 = max(h_low, 0);
  tp.y1 = min(h_low + 1, H - 1);
  tp.x0 = max(w_low, 0);
  tp.x1 = min(w_low + 1, W - 1);
  return tp;
}

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

}  // namespace msda
