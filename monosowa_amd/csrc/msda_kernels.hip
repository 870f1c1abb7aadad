// MSDA forward / backward kernels for gfx950 (MI355X, CDNA4, wave64) -- hand-written HIP.
//
// What is computed is the reference operator (see include/monosowa_msda.h and msda_common.h for
// the citations into ops/src/cuda/ms_deform_im2col_cuda.cuh); how it is mapped to the machine
// is new:
//
//   * fwd, D == 32, L*P == 16, f32 ("d32" path, the shipped MonoDETR geometry):
//     a (query, head) pair owns 8 lanes x float4 = one 128-B value row per corner, so a wave
//     covers 8 pairs and every corner fetch is a full 128-B segment (global_load_dwordx4).
//     The pair's 16 (x, y) locations and 16 weights are fetched ONCE, coalesced (one float4 +
//     one float2 per lane), and broadcast inside the 8-lane group with wave shuffles.
//   * generic path (any D, L, P; f32 / f64): 32 lanes per pair, lane = channel (strided over D).
//   * bwd: lane = channel so that each corner's scatter is one contiguous 128-B row segment per
//     half-wave -- the full-rate shape for global_atomic_add_f32 on gfx950; the per-point
//     channel reductions for grad_loc / grad_attn_w are wave shuffles (the reference serialises
//     them through shared memory on thread 0, cuh:376-394).
#include "msda_common.h"

namespace msda {

// ----------------------------------------------------------------------------------------------
// Forward, d32 path.
// ----------------------------------------------------------------------------------------------
template <int L, int P>
__global__ __launch_bounds__(256) void fwd_d32_kernel(
    const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const float *__restrict__ loc,
    const float *__restrict__ attw, float *__restrict__ out, int S, int M, int Lq,
    long long n_pairs) {
  static_assert(L * P == 16, "d32 path stages 16 points per pair");
  const int lane = threadIdx.x & 63;
  const int sub = threadIdx.x & 7;     // which float4 of the 32-channel row
  const int grp = lane & ~7;           // first lane of this pair's group
  int Hs[L], Ws[L], st[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    Hs[l] = (int)shapes[2 * l];
    Ws[l] = (int)shapes[2 * l + 1];
    st[l] = (int)lsi[l];
  }
  const int tok = M * 32;              // floats between consecutive tokens
  // exact grid (padded to a multiple of 8 workgroups), XCD x takes the x-th contiguous eighth of the pairs
  const long long pair = xcd_chunked_block(gridDim.x) * 32 + (threadIdx.x >> 3);
  if (pair < n_pairs) {
    const int m = (int)(pair % M);
    const int b = (int)(pair / ((long long)M * Lq));
    const float *vb = value + ((long long)b * S * M + m) * 32 + sub * 4;
    const float4 lc = ld4(loc + pair * 32 + sub * 4);                       // points 2*sub, 2*sub+1
    const float2 aw = *reinterpret_cast<const float2 *>(attw + pair * 16 + sub * 2);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const float *vl = vb + (long long)st[l] * tok;
      const int H = Hs[l], W = Ws[l];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int pt = l * P + p;
        const int src = grp | (pt >> 1);
        const float lx = __shfl((pt & 1) ? lc.z : lc.x, src);
        const float ly = __shfl((pt & 1) ? lc.w : lc.y, src);
        const float wt = __shfl((pt & 1) ? aw.y : aw.x, src);
        const Tap<float> tp = make_tap<float>(lx, ly, H, W);
        const int r0 = tp.y0 * W, r1 = tp.y1 * W;
        const float4 v1 = ld4(vl + (r0 + tp.x0) * tok);
        const float4 v2 = ld4(vl + (r0 + tp.x1) * tok);
        const float4 v3 = ld4(vl + (r1 + tp.x0) * tok);
        const float4 v4 = ld4(vl + (r1 + tp.x1) * tok);
        acc.x += (tp.w1 * v1.x + tp.w2 * v2.x + tp.w3 * v3.x + tp.w4 * v4.x) * wt;
        acc.y += (tp.w1 * v1.y + tp.w2 * v2.y + tp.w3 * v3.y + tp.w4 * v4.y) * wt;
        acc.z += (tp.w1 * v1.z + tp.w2 * v2.z + tp.w3 * v3.z + tp.w4 * v4.z) * wt;
        acc.w += (tp.w1 * v1.w + tp.w2 * v2.w + tp.w3 * v3.w + tp.w4 * v4.w) * wt;
      }
    }
    st4(out + pair * 32 + sub * 4, acc);
  }
}

// ----------------------------------------------------------------------------------------------
// Forward, generic path: 32 lanes per (query, head) pair, lane = channel (strided over D).
// ----------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void fwd_generic_kernel(
    const T *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const T *__restrict__ loc, const T *__restrict__ attw, T *__restrict__ out, int S, int M, int D,
    int L, int Lq, int P, long long n_pairs) {
  const int c0 = threadIdx.x & 31;
  const long long tok = (long long)M * D;
  const long long stride = (long long)gridDim.x * (blockDim.x >> 5);
  for (long long pair = (long long)blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5); pair < n_pairs;
       pair += stride) {
    const int m = (int)(pair % M);
    const long long b = pair / ((long long)M * Lq);
    const T *vb = value + (b * S * M + m) * D;
    const T *lp = loc + pair * L * P * 2;
    const T *wp = attw + pair * L * P;
    for (int c = c0; c < D; c += 32) {
      T col = 0;
      for (int l = 0; l < L; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const T *vl = vb + lsi[l] * tok + c;
        for (int p = 0; p < P; ++p) {
          const Tap<T> tp = make_tap<T>(lp[(l * P + p) * 2], lp[(l * P + p) * 2 + 1], H, W);
          if (!tp.valid) continue;   // uniform across the pair's lanes
          const long long r0 = (long long)tp.y0 * W, r1 = (long long)tp.y1 * W;
          const T v1 = vl[(r0 + tp.x0) * tok], v2 = vl[(r0 + tp.x1) * tok];
          const T v3 = vl[(r1 + tp.x0) * tok], v4 = vl[(r1 + tp.x1) * tok];
          col += (tp.w1 * v1 + tp.w2 * v2 + tp.w3 * v3 + tp.w4 * v4) * wp[l * P + p];
        }
      }
      out[pair * D + c] = col;
    }
  }
}

// ----------------------------------------------------------------------------------------------
// Backward, generic path (also the first d32 implementation): 32 lanes per pair, lane = channel.
// grad_value is accumulated with global atomics (must be zero on entry); grad_loc and
// grad_attn_w are written exactly once per point (zeros for points that do not contribute,
// which is what the reference's zeros_like leaves there, cu:122-123).
// ----------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T half_wave_sum(T v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 32);
  return v;
}

template <typename T>
__global__ __launch_bounds__(256) void bwd_generic_kernel(
    const T *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const T *__restrict__ loc, const T *__restrict__ attw, const T *__restrict__ grad_out,
    T *__restrict__ grad_value, T *__restrict__ grad_loc, T *__restrict__ grad_attw, int S, int M,
    int D, int L, int Lq, int P, long long n_pairs) {
  const int c0 = threadIdx.x & 31;
  const long long tok = (long long)M * D;
  const long long stride = (long long)gridDim.x * (blockDim.x >> 5);
  for (long long pair = (long long)blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5); pair < n_pairs;
       pair += stride) {
    const int m = (int)(pair % M);
    const long long b = pair / ((long long)M * Lq);
    const long long voff = (b * S * M + m) * D;
    const T *lp = loc + pair * L * P * 2;
    const T *wp = attw + pair * L * P;
    const T *gp = grad_out + pair * D;
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const long long loff = voff + lsi[l] * tok;
      for (int p = 0; p < P; ++p) {
        const int pt = l * P + p;
        const Tap<T> tp = make_tap<T>(lp[pt * 2], lp[pt * 2 + 1], H, W);
        T ga = 0, gw = 0, gh = 0;
        if (tp.valid) {   // uniform across the pair's lanes
          const T wt = wp[pt];
          const long long i1 = loff + ((long long)tp.y0 * W + tp.x0) * tok;
          const long long i2 = loff + ((long long)tp.y0 * W + tp.x1) * tok;
          const long long i3 = loff + ((long long)tp.y1 * W + tp.x0) * tok;
          const long long i4 = loff + ((long long)tp.y1 * W + tp.x1) * tok;
          for (int c = c0; c < D; c += 32) {
            const T top = gp[c];
            const T tgv = top * wt;
            // dropped corners: weight 0 and value treated as 0 (cuh:114-152)
            const T v1 = (tp.t && tp.l) ? value[i1 + c] : (T)0;
            const T v2 = (tp.t && tp.r) ? value[i2 + c] : (T)0;
            const T v3 = (tp.b && tp.l) ? value[i3 + c] : (T)0;
            const T v4 = (tp.b && tp.r) ? value[i4 + c] : (T)0;
            if (tp.t && tp.l) atomicAdd(grad_value + i1 + c, tp.w1 * tgv);
            if (tp.t && tp.r) atomicAdd(grad_value + i2 + c, tp.w2 * tgv);
            if (tp.b && tp.l) atomicAdd(grad_value + i3 + c, tp.w3 * tgv);
            if (tp.b && tp.r) atomicAdd(grad_value + i4 + c, tp.w4 * tgv);
            const T ghw = -tp.hw * v1 - tp.lw * v2 + tp.hw * v3 + tp.lw * v4;
            const T gww = -tp.hh * v1 + tp.hh * v2 - tp.lh * v3 + tp.lh * v4;
            ga += top * (tp.w1 * v1 + tp.w2 * v2 + tp.w3 * v3 + tp.w4 * v4);
            gw += (T)W * gww * tgv;
            gh += (T)H * ghw * tgv;
          }
        }
        ga = half_wave_sum(ga);
        gw = half_wave_sum(gw);
        gh = half_wave_sum(gh);
        if (c0 == 0) {
          grad_loc[(pair * L * P + pt) * 2] = gw;
          grad_loc[(pair * L * P + pt) * 2 + 1] = gh;
          grad_attw[pair * L * P + pt] = ga;
        }
      }
    }
  }
}

}  // namespace msda
