// Backward, d32 path (D == 32, L*P == 16, f32): three kernels, no global atomics on the fine levels.
//
// The reference scatters grad_value with 64 global float atomics per output element
// (cuh:125-152); on gfx950 global f32 atomics run at ~1.3 TB/s chip-wide whatever their locality,
// which makes that formulation ~7 ms per encoder layer at B=16.  Here grad_value is accumulated
// on chip instead:
//
//   K1 bwd_prep_kernel      transposes the sampling points into per-(batch, head, level) record
//                           lists {h_im, w_im} / {attn_w} (12 B per point, coalesced for the
//                           scans below); points failing the cuh:274 test get a sentinel.
//   K2 bwd_scatter_kernel   "tile owner": one workgroup owns a tile of one level of one
//                           (batch, head) -- 256 value rows x 32 double accumulators in LDS.  It
//                           scans the level's record list, queues the points with a corner inside
//                           its tile, and adds w_corner * attn_w * grad_out[b,q,m,:] into LDS with
//                           ds_add_f64 (8 lanes x 4 channels per point, channel order rotated per
//                           lane group to spread the LDS banks).
//                           Tiles owned by a single workgroup are written back with plain stores
//                           (no zero-fill, no atomics); coarse levels, whose few rows receive a
//                           quarter of all points each, are split over several workgroups by
//                           query range and flushed with full-row atomic adds into a zeroed region.
//   K3 bwd_gather_kernel    grad_loc / grad_attn_w: the forward's gather (8 lanes x float4 per
//                           (query, head)), channel sums by an 8-lane transposing butterfly that
//                           leaves lane j holding points 2j, 2j+1 -- the layout of the coalesced
//                           float4 / float2 stores.
#include "msda_common.h"

namespace msda {

constexpr float kInvalidCoord = -8.0f;   // floor() = -8: no corner can fall inside any tile

// ------------------------------------------------------------------------------------------ K1
// one thread per output record; rec index o = (((b*M + m)*L + l)*Lq + q)*P + p
__global__ __launch_bounds__(256) void bwd_prep_kernel(
    const float *__restrict__ loc, const float *__restrict__ attw, const int64_t *__restrict__ shapes,
    float2 *__restrict__ rec_hw, float *__restrict__ rec_aw, int M, int L, int Lq, int P, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += stride) {
    long long t = o;
    const int p = (int)(t % P); t /= P;
    const int q = (int)(t % Lq); t /= Lq;
    const int l = (int)(t % L); t /= L;
    const int m = (int)(t % M);
    const long long b = t / M;
    const long long src = (((b * Lq + q) * M + m) * L + l) * P + p;
    const float2 xy = *reinterpret_cast<const float2 *>(loc + src * 2);
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const float h_im = scale_loc(xy.y, H), w_im = scale_loc(xy.x, W);
    const bool ok = (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W);
    rec_hw[o] = ok ? make_float2(h_im, w_im) : make_float2(kInvalidCoord, kInvalidCoord);
    rec_aw[o] = attw[src];
  }
}

// ------------------------------------------------------------------------------------------ K2
// LDS float atomics (ds_add_f32) are serialised per lane on gfx950 (0.33 lane-adds/clk/CU measured,
// tools/ubench/lds_atomic*.hip) while ds_add_f64 runs at 3.9 and ds_add_u32 at 11-13: the tile is
// accumulated in double, which is also more accurate than the reference's float atomics.
constexpr int kQueue = 128;              // hit queue slots per wave (ring)
constexpr int kScatterThreads = 512;
constexpr int kScatterWaves = kScatterThreads / 64;
constexpr int kDrainBatches = 4;         // 8-hit batches whose grad_out rows are fetched together

__device__ __forceinline__ float pick(const float4 &v, int i) {
  return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}

__global__ __launch_bounds__(kScatterThreads, 4) void bwd_scatter_kernel(
    const float2 *__restrict__ rec_hw, const float *__restrict__ rec_aw,
    const float *__restrict__ grad_out, float *__restrict__ grad_value, const BwdPlan plan, int B, int S,
    int M, int Lq, int P) {
  __shared__ double acc[kTileRows * 32];                 // 64 KB
  __shared__ float4 queue[kScatterWaves][kQueue];        // 16 KB

  // blockIdx -> (batch*head, item); all items of one (batch, head) share blockIdx % 8, i.e. one XCD
  // (speed only: they re-scan the same record lists out of that XCD's L2).
  const int BM = B * M;
  const int bm = (int)(blockIdx.x % 8) + 8 * (int)(blockIdx.x / (8 * plan.n_items));
  const int it = (int)((blockIdx.x / 8) % plan.n_items);
  if (bm >= BM) return;
  const int b = bm / M, m = bm % M;
  // item -> (level, tile, chunk)
  int oi = 0;
  while (oi + 1 < plan.n_levels && it >= plan.first_item[oi + 1]) ++oi;
  const int l = plan.order[oi];
  const int local = it - plan.first_item[oi];
  const int n_chunks = plan.n_chunks[l];
  const int chunk = local % n_chunks, tile = local / n_chunks;
  const int H = plan.H[l], W = plan.W[l];
  const int y0 = (tile / plan.n_tx[l]) * plan.th[l], x0 = (tile % plan.n_tx[l]) * plan.tw[l];
  const int th = min(plan.th[l], H - y0), tw = min(plan.tw[l], W - x0);
  const int n_rows = th * tw;
  const long long n_pts = (long long)Lq * P;
  const int pt_begin = (int)(n_pts * chunk / n_chunks), pt_end = (int)(n_pts * (chunk + 1) / n_chunks);
  const bool exclusive = n_chunks == 1;

  for (int i = threadIdx.x; i < n_rows * 32; i += kScatterThreads) acc[i] = 0.0;
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane >> 3, sub = lane & 7;
  const long long list = ((long long)bm * plan.n_levels + l) * n_pts;        // this level's record list
  const float2 *hw_list = rec_hw + list;
  const float *aw_list = rec_aw + list;
  const float *go_base = grad_out + ((long long)b * Lq * M + m) * 32 + sub * 4;     // + q * M*32
  float4 *myq = queue[wave];
  int qhead = 0, qtail = 0;
  // rotate the channel order by the group index: in an LDS pass the groups of a 32-lane half then
  // touch different banks of every 4-bank quad
  const int r0 = grp & 3;
  const int c0 = sub * 4 + r0, c1 = sub * 4 + ((r0 + 1) & 3), c2 = sub * 4 + ((r0 + 2) & 3),
            c3 = sub * 4 + ((r0 + 3) & 3);

  // Adds up to NB x 8 queued hits into the LDS tile, one hit per 8-lane group and batch; the
  // grad_out rows of all batches are requested before the first is consumed.
  auto drain = [&](const int nb) {
    float4 rec[kDrainBatches], g[kDrainBatches];
    bool active[kDrainBatches];
#pragma unroll
    for (int k = 0; k < kDrainBatches; ++k) {
      active[k] = (k < nb) && (qhead + 8 * k + grp) < qtail;
      rec[k] = myq[(qhead + 8 * k + grp) & (kQueue - 1)];
      const int q = __float_as_int(rec[k].w) / P;
      g[k] = active[k] ? ld4(go_base + (long long)q * M * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    qhead += 8 * nb;
#pragma unroll
    for (int k = 0; k < kDrainBatches; ++k) {
      if (!active[k]) continue;
      const Tap<float> tp = make_tap_im(rec[k].x, rec[k].y, H, W);
      const float aw = rec[k].z;
      const float t0 = pick(g[k], r0) * aw, t1 = pick(g[k], (r0 + 1) & 3) * aw;
      const float t2 = pick(g[k], (r0 + 2) & 3) * aw, t3 = pick(g[k], (r0 + 3) & 3) * aw;
      const int ry0 = tp.y0 - y0, ry1 = tp.y1 - y0, rx0 = tp.x0 - x0, rx1 = tp.x1 - x0;
      const bool iy0 = tp.t && (unsigned)ry0 < (unsigned)th, iy1 = tp.b && (unsigned)ry1 < (unsigned)th;
      const bool ix0 = tp.l && (unsigned)rx0 < (unsigned)tw, ix1 = tp.r && (unsigned)rx1 < (unsigned)tw;
      auto add_row = [&](int ry, int rx, float w) {
        double *row = acc + (ry * tw + rx) * 32;
        atomicAdd(row + c0, (double)(w * t0));
        atomicAdd(row + c1, (double)(w * t1));
        atomicAdd(row + c2, (double)(w * t2));
        atomicAdd(row + c3, (double)(w * t3));
      };
      if (iy0 && ix0) add_row(ry0, rx0, tp.w1);
      if (iy0 && ix1) add_row(ry0, rx1, tp.w2);
      if (iy1 && ix0) add_row(ry1, rx0, tp.w3);
      if (iy1 && ix1) add_row(ry1, rx1, tp.w4);
    }
  };

  const int step = kScatterWaves * 64;
  int base = pt_begin + wave * 64;
  float2 hw_next = make_float2(kInvalidCoord, kInvalidCoord);
  if (base + lane < pt_end) hw_next = hw_list[base + lane];
  for (; base < pt_end; base += step) {
    const int idx = base + lane;
    const float2 hw = hw_next;
    hw_next = make_float2(kInvalidCoord, kInvalidCoord);
    if (idx + step < pt_end) hw_next = hw_list[idx + step];          // prefetch the next slice
    const int h_low = (int)floorf(hw.x), w_low = (int)floorf(hw.y);
    // a corner row/col counts if it lies inside the tile (tiles lie inside the level, so this also
    // implies the cuh:56-79 bounds checks)
    const bool row_in = ((unsigned)(h_low - y0) < (unsigned)th) || ((unsigned)(h_low + 1 - y0) < (unsigned)th);
    const bool col_in = ((unsigned)(w_low - x0) < (unsigned)tw) || ((unsigned)(w_low + 1 - x0) < (unsigned)tw);
    const bool hit = row_in && col_in;
    const unsigned long long mask = __ballot(hit);
    if (hit) {
      const int rank = __popcll(mask & ((1ull << lane) - 1ull));
      myq[(qtail + rank) & (kQueue - 1)] = make_float4(hw.x, hw.y, aw_list[idx], __int_as_float(idx));
    }
    __builtin_amdgcn_wave_barrier();      // queue writes above are read by other lanes of this wave below
    qtail += __popcll(mask);
    while (qtail - qhead >= 8 * kDrainBatches) drain(kDrainBatches);
  }
  while (qhead < qtail) drain(min(kDrainBatches, (qtail - qhead + 7) / 8));
  __syncthreads();

  // write the tile back (rounded to float once)
  const long long tok0 = (long long)b * S + plan.start[l];
  if (exclusive) {
    for (int r = threadIdx.x >> 3; r < n_rows; r += kScatterThreads / 8) {
      const int ry = r / tw, rx = r - ry * tw;
      float *dst = grad_value + ((tok0 + (long long)(y0 + ry) * W + (x0 + rx)) * M + m) * 32 + sub * 4;
      const double *a = acc + r * 32 + sub * 4;
      st4(dst, make_float4((float)a[0], (float)a[1], (float)a[2], (float)a[3]));
    }
  } else {
    const int ch = threadIdx.x & 31;
    for (int r = threadIdx.x >> 5; r < n_rows; r += kScatterThreads / 32) {
      const float v = (float)acc[r * 32 + ch];
      const int ry = r / tw, rx = r - ry * tw;
      if (v != 0.f)
        atomicAdd(grad_value + ((tok0 + (long long)(y0 + ry) * W + (x0 + rx)) * M + m) * 32 + ch, v);
    }
  }
}

// ------------------------------------------------------------------------------------------ K3
// DPP lane exchanges inside an 8-lane group (no LDS traffic): partner = sub ^ 1, sub ^ 2
// (quad_perm) and 7 - sub (row_half_mirror; note it flips lane parity).
template <int CTRL>
__device__ __forceinline__ float dpp_xchg(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}
constexpr int kDppHalfMirror = 0x141, kDppXor2 = 0x4E, kDppXor1 = 0xB1;

template <int L, int P>
__global__ __launch_bounds__(256, 4) void bwd_gather_kernel(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const float *__restrict__ loc, const float *__restrict__ attw, const float *__restrict__ grad_out,
    float *__restrict__ grad_loc, float *__restrict__ grad_attw, int S, int M, int Lq, long long n_pairs) {
  static_assert(L == 4 && P == 4, "d32 path: 4 levels x 4 points, one level per lane pair");
  const int lane = threadIdx.x & 63;
  const int sub = threadIdx.x & 7;
  const int grp = lane & ~7;
  const bool odd = sub & 1;
  const int tok = M * 32;
  // exact grid (padded to a multiple of 8 workgroups), XCD x takes the x-th contiguous eighth of the pairs
  const long long pair = xcd_chunked_block(gridDim.x) * 32 + (threadIdx.x >> 3);
  if (pair < n_pairs) {
    const int m = (int)(pair % M);
    const int b = (int)(pair / ((long long)M * Lq));
    const float *vb = value + ((long long)b * S * M + m) * 32 + sub * 4;
    const float4 lc = ld4(loc + pair * 32 + sub * 4);                       // points 2*sub, 2*sub+1
    const float2 aw = *reinterpret_cast<const float2 *>(attw + pair * 16 + sub * 2);
    const float4 g = ld4(grad_out + pair * 32 + sub * 4);
    float4 out_loc = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 out_aw = make_float2(0.f, 0.f);
    // a real loop over levels bounds the corner loads in flight (4 points x 4 corners x float4)
#pragma unroll 1
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const float *vl = vb + (long long)lsi[l] * tok;
      float ga[P], gx[P], gy[P];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        // the level's points 4l .. 4l+3 sit in lanes 2l (p = 0, 1) and 2l + 1 (p = 2, 3)
        const int src = grp | (2 * l + (p >> 1));
        const float lx = __shfl((p & 1) ? lc.z : lc.x, src);
        const float ly = __shfl((p & 1) ? lc.w : lc.y, src);
        const float wt = __shfl((p & 1) ? aw.y : aw.x, src);
        const Tap<float> tp = make_tap<float>(lx, ly, H, W);
        const int r0 = tp.y0 * W, r1 = tp.y1 * W;
        const float4 v1 = ld4(vl + (r0 + tp.x0) * tok);
        const float4 v2 = ld4(vl + (r0 + tp.x1) * tok);
        const float4 v3 = ld4(vl + (r1 + tp.x0) * tok);
        const float4 v4 = ld4(vl + (r1 + tp.x1) * tok);
        // per-corner dot products with grad_out over this lane's 4 channels
        const float d1 = g.x * v1.x + g.y * v1.y + g.z * v1.z + g.w * v1.w;
        const float d2 = g.x * v2.x + g.y * v2.y + g.z * v2.z + g.w * v2.w;
        const float d3 = g.x * v3.x + g.y * v3.y + g.z * v3.z + g.w * v3.w;
        const float d4 = g.x * v4.x + g.y * v4.y + g.z * v4.z + g.w * v4.w;
        // dropped corners contribute neither value nor slope (cuh:114-152)
        const float e1 = (tp.t && tp.l) ? d1 : 0.f, e2 = (tp.t && tp.r) ? d2 : 0.f;
        const float e3 = (tp.b && tp.l) ? d3 : 0.f, e4 = (tp.b && tp.r) ? d4 : 0.f;
        ga[p] = tp.w1 * e1 + tp.w2 * e2 + tp.w3 * e3 + tp.w4 * e4;
        gx[p] = (float)W * wt * (tp.hh * (e2 - e1) + tp.lh * (e4 - e3));
        gy[p] = (float)H * wt * (tp.hw * (e3 - e1) + tp.lw * (e4 - e2));
      }
      // channel sums over the 8 lanes.  (1) all-reduce lane <-> 7 - lane on all 12 values, which
      // leaves both quads holding the same four pair sums; (2) inside a quad, even lanes collect
      // points 0,1 and odd lanes points 2,3 (exchange with sub ^ 1); (3) add sub ^ 2.
#pragma unroll
      for (int p = 0; p < P; ++p) {
        ga[p] += dpp_xchg<kDppHalfMirror>(ga[p]);
        gx[p] += dpp_xchg<kDppHalfMirror>(gx[p]);
        gy[p] += dpp_xchg<kDppHalfMirror>(gy[p]);
      }
      float r[6];
      {
        const float k0 = odd ? gx[2] : gx[0], s0 = odd ? gx[0] : gx[2];
        const float k1 = odd ? gy[2] : gy[0], s1 = odd ? gy[0] : gy[2];
        const float k2 = odd ? gx[3] : gx[1], s2 = odd ? gx[1] : gx[3];
        const float k3 = odd ? gy[3] : gy[1], s3 = odd ? gy[1] : gy[3];
        const float k4 = odd ? ga[2] : ga[0], s4 = odd ? ga[0] : ga[2];
        const float k5 = odd ? ga[3] : ga[1], s5 = odd ? ga[1] : ga[3];
        r[0] = k0 + dpp_xchg<kDppXor1>(s0); r[1] = k1 + dpp_xchg<kDppXor1>(s1);
        r[2] = k2 + dpp_xchg<kDppXor1>(s2); r[3] = k3 + dpp_xchg<kDppXor1>(s3);
        r[4] = k4 + dpp_xchg<kDppXor1>(s4); r[5] = k5 + dpp_xchg<kDppXor1>(s5);
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) r[i] += dpp_xchg<kDppXor2>(r[i]);
      if ((sub >> 1) == l) {          // lanes 2l, 2l+1 own this level's points
        out_loc = make_float4(r[0], r[1], r[2], r[3]);
        out_aw = make_float2(r[4], r[5]);
      }
    }
    st4(grad_loc + pair * 32 + sub * 4, out_loc);
    *reinterpret_cast<float2 *>(grad_attw + pair * 16 + sub * 2) = out_aw;
  }
}

}  // namespace msda
