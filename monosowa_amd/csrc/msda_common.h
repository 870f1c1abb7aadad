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
  int h_low, w_low;     // unclamped top-left corner (0 for a point that fails the cuh:274 test)
  T w1, w2, w3, w4;     // (y0,x0) (y0,x1) (y1,x0) (y1,x1); 0 where the corner is outside
  T lh, lw, hh, hw;     // fractional offsets (for the location gradient)
  bool t, b, l, r;      // which rows / columns are inside
  bool valid;           // the cuh:274 test
};

// Individually rounded operations.  hipcc compiles device code with -ffp-contract=fast and HIP's "_rn" intrinsics for
// multiply / add are plain `x * y` / `x + y` (__clang_hip_math.h), so a product handed to a subtraction through them still
// becomes one v_fma -- found on a point whose rounded product lands exactly on k + 0.5 (loc * W = 8.49999964 -> 8.5 -> w_im
// = 8.0, but 7.9999995 when fused): floor() and with it grad_loc change side.  The pragma removes the `contract` flag from
// these operations themselves, which survives inlining.
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}
__device__ __forceinline__ double mul_rn(double a, double b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ double sub_rn(double a, double b) {
#pragma clang fp contract(off)
  return a - b;
}

// Product rounded once, then the subtraction: the reference evaluates `loc * size - 0.5` with a
// double literal, so the float product is rounded before the subtraction (no fused multiply-add).
__device__ __forceinline__ float scale_loc(float loc, int size) { return sub_rn(mul_rn(loc, (float)size), 0.5f); }
__device__ __forceinline__ double scale_loc(double loc, int size) { return sub_rn(mul_rn(loc, (double)size), 0.5); }

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
  tp.h_low = h_low;
  tp.w_low = w_low;
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

// ---- "near" points of the self-attention shape (Lq == S: query i sits at token i's pixel) --------------------------------
// A query at pixel c of a level of extent Nq has its centre at (c + 0.5) / Nq * N - 0.5 in pixels of a level of extent N;
// centre_floor is the floor of that (the same integer formula on the host: msda_window.h / msda_scatter_plan.h).  A sampling
// point is NEAR when the top-left corner of its bilinear footprint lies within `reach` pixels of that, in both axes.  The
// row-tile scatter (msda_scatter_rows.hip) finds the near points of a value tile by scanning the queries around it; every
// other point ("far": long learned offsets) is added to grad_value by the gather kernel with global atomics.  Both kernels
// evaluate THIS function, so the two sets partition the points exactly.
__device__ __forceinline__ int centre_floor(int c, int Nq, int N) {
  // floor(num / den), den > 0, |num| < 2^23 (host: level extents <= 2048): a float estimate, then an exact integer fix-up
  const int num = (2 * c + 1) * N - Nq, den = 2 * Nq;
  int q = (int)floorf((float)num / (float)den);
  const int r = num - q * den;
  q += (r >= den) - (r < 0);
  return q;
}
__device__ __forceinline__ bool near_point(int h_low, int w_low, int cy, int cx, int reach) {
  return abs(h_low - cy) <= reach && abs(w_low - cx) <= reach;
}

// XCD-aware work mapping (speed only, never correctness): workgroups are dealt round-robin over the 8
// XCDs, so workgroup b and b+8 share an L2.  Giving XCD x the x-th contiguous eighth of the work keeps
// each L2's working set to the value rows around one stretch of queries instead of the whole tensor.
__device__ __forceinline__ long long xcd_chunked_block(long long n_blocks_padded8) {
  const long long per_xcd = n_blocks_padded8 >> 3;
  return (long long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
}

// Ordering point for wave-private LDS hand-offs (lane A writes, lane B of the SAME wave reads): the LDS
// executes one wave's instructions in order, so only the compiler has to be kept from moving accesses
// across.  (A wavefront-scope release/acquire fence would also emit s_waitcnt vmcnt(0) lgkmcnt(0) and drain
// every prefetch and LDS atomic in flight -- measured: +40 % on the scatter kernel.)
__device__ __forceinline__ void wave_lds_order() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

}  // namespace msda

namespace msda {

// Tap from already-scaled image coordinates of a point known to pass the cuh:274 test.
__device__ __forceinline__ Tap<float> make_tap_im(float h_im, float w_im, int H, int W) {
  Tap<float> tp;
  tp.valid = true;
  const float hf = floorf(h_im), wf = floorf(w_im);
  const int h_low = (int)hf, w_low = (int)wf;
  tp.h_low = h_low;
  tp.w_low = w_low;
  tp.lh = h_im - hf;
  tp.lw = w_im - wf;
  tp.hh = 1.f - tp.lh;
  tp.hw = 1.f - tp.lw;
  tp.t = (h_low >= 0);
  tp.b = (h_low + 1 <= H - 1);
  tp.l = (w_low >= 0);
  tp.r = (w_low + 1 <= W - 1);
  tp.w1 = (tp.t && tp.l) ? tp.hh * tp.hw : 0.f;
  tp.w2 = (tp.t && tp.r) ? tp.hh * tp.lw : 0.f;
  tp.w3 = (tp.b && tp.l) ? tp.lh * tp.hw : 0.f;
  tp.w4 = (tp.b && tp.r) ? tp.lh * tp.lw : 0.f;
  tp.y0 = max(h_low, 0);
  tp.y1 = min(h_low + 1, H - 1);
  tp.x0 = max(w_low, 0);
  tp.x1 = min(w_low + 1, W - 1);
  return tp;
}

template <int CTRL>
__device__ __forceinline__ float dpp_x(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}

// ---- fused prologue (FUSED kernels): the module-level arithmetic of ops/modules/ms_deform_attn.py:146-155 --------
// attention weights = softmax over the pair's L*P = 16 logits; sampling location of point (l, p):
//   ref_dim 2: ref[l] + offset / (W_l, H_l)                                       (ms_deform_attn.py:149-152)
//   ref_dim 6: ref[l][:2] + offset / P * (ref[l][2]+ref[l][3], ref[l][4]+ref[l][5]) * 0.5     (:153-155)
// evaluated operation by operation (no contraction) like the PyTorch expressions.
struct RefScale { float rx, ry, sx, sy; };   // location = (rx, ry) + f(offset, sx, sy)

__device__ __forceinline__ RefScale load_ref(const float *rp, int ref_dim, int H, int W) {
  RefScale r;
  r.rx = rp[0];
  r.ry = rp[1];
  if (ref_dim == 2) { r.sx = (float)W; r.sy = (float)H; }
  else { r.sx = add_rn(rp[2], rp[3]); r.sy = add_rn(rp[4], rp[5]); }
  return r;
}
template <int P>
__device__ __forceinline__ float loc_from_offset(float ref, float off, float s, int ref_dim) {
  if (ref_dim == 2) return add_rn(ref, __fdiv_rn(off, s));
  return add_rn(ref, mul_rn(mul_rn(__fdiv_rn(off, (float)P), s), 0.5f));
}
// a / b for a divisor whose correctly rounded reciprocal rc = RN(1 / b) is at hand (a level extent: a small integer, constant
// per lane): two residual corrections, each `r = a - q b` exact in the FMA.  q1 is within one ulp of a / b (its error before
// rounding is the first estimate's 2^-23 times rc's 2^-24), and a correction of a faithful quotient with a correctly rounded
// reciprocal rounds correctly unless b's significand is all ones (Markstein 1990) -- never for an integer below 2^24.  Five
// full-rate instructions where the IEEE division expands to ~11 with a quarter-rate v_rcp_f32.  Finite operands with a
// normal quotient only (an infinite offset gives NaN instead of inf: either fails the cuh:274 test).
__device__ __forceinline__ float div_rc(float a, float b, float rc) {
  float q = mul_rn(a, rc);
  q = __builtin_fmaf(__builtin_fmaf(-q, b, a), rc, q);
  q = __builtin_fmaf(__builtin_fmaf(-q, b, a), rc, q);
  return q;
}
template <int P>
__device__ __forceinline__ float offset_grad(float g, float s, int ref_dim) {       // autograd of the line above
  if (ref_dim == 2) return __fdiv_rn(g, s);
  return __fdiv_rn(mul_rn(mul_rn(g, 0.5f), s), (float)P);
}
// sum / max over the 8 lanes of a (query, head) group (all lanes get the result)
__device__ __forceinline__ float group_sum(float v) {
  v += dpp_x<0x141>(v); v += dpp_x<0x4E>(v); v += dpp_x<0xB1>(v);
  return v;
}
__device__ __forceinline__ float group_max(float v) {
  v = fmaxf(v, dpp_x<0x141>(v)); v = fmaxf(v, dpp_x<0x4E>(v)); v = fmaxf(v, dpp_x<0xB1>(v));
  return v;
}

// ---- host/device shared plan of the tile-owner backward (passed to the kernel by value) -------
// Work items are never materialised: item -> (level, tile, query chunk) is derived on the device
// from per-level tilings, so any image size fits the kernel argument.
constexpr int kMaxLevels = 8;
constexpr int kTileRows = 256;        // LDS accumulator rows (x 32 doubles = 256 B) per workgroup

struct BwdPlan {
  int n_items;                 // sum over levels of n_ty * n_tx * n_chunks
  int n_levels;
  int H[kMaxLevels], W[kMaxLevels], start[kMaxLevels];
  int th[kMaxLevels], tw[kMaxLevels];        // nominal tile extent (th * tw <= kTileRows)
  int n_ty[kMaxLevels], n_tx[kMaxLevels];
  int n_chunks[kMaxLevels];                  // workgroups sharing one tile (split by query range)
  int order[kMaxLevels];                     // levels sorted by work per item, heaviest first
  int first_item[kMaxLevels + 1];            // prefix over `order`
};

}  // namespace msda
