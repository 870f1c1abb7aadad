// Directional launch plan of the self-attention kernels: sampling statistics -> per-head tables (msda_plan.h).
#include "msda_common.h"
#include "msda_plan.h"

namespace msda {

constexpr int kStatsBlocks = 256;        // workgroups of dir_stats_kernel = partial statistics per (head, level)
constexpr int kStatsThreads = 256;

__device__ __forceinline__ int plan_floor_div_dev(int a, int b) {        // b > 0
  const int q = a / b, r = a - q * b;
  return r < 0 ? q - 1 : q;
}

// ---- 1. statistics of d = (h_low - cf_y, w_low - cf_x) over a sample of the call's points ---------------------------------------
// MODE 0: `loc` = raw sampling offsets [B, Lq, .] with row stride loc_rs (the fused operator's input), `ref` = 2-d reference
//         points [B, Lq, 4, 2]; locations are evaluated exactly like the kernels do (msda_common.h: loc_from_offset).
// MODE 1: `loc` = the forward's saved sampling locations, level-major [B, M, L, Lq, P, 2].
// Every workgroup leaves ONE partial record per (head, level) in partial[blockIdx.x][M][4] (plain stores: no zero fill, no
// same-address atomics -- 1000 workgroups adding into 32 records would serialise in L2).
template <int MODE>
__global__ __launch_bounds__(kStatsThreads) void dir_stats_kernel(const float *__restrict__ loc, const float *__restrict__ ref,
                                                                  const PlanGeom g, int B, int loc_rs,
                                                                  DirStats *__restrict__ partial) {
  __shared__ int sh[kPlanMaxHeads * 4][10];       // n, up_y, dn_y, up_x, dn_x, sum_y, sum_x, sq_y, sq_x, -
  const int M = g.M, S = g.S;
  for (int i = threadIdx.x; i < M * 4 * 10; i += kStatsThreads) (&sh[0][0])[i] = 0;
  __syncthreads();
  const int ns = min(kPlanSamples, S);
  const long long total = (long long)B * ns * M * 4;
  for (long long t = (long long)blockIdx.x * kStatsThreads + threadIdx.x; t < total; t += (long long)gridDim.x * kStatsThreads) {
    const int l = (int)(t & 3);
    const long long u = t >> 2;
    const int m = (int)(u % M);
    const long long v = u / M;
    const int i = (int)(v % ns), b = (int)(v / ns);
    // a pseudo-random token of this batch element (a regular stride would alias with the image columns)
    const unsigned q = (unsigned)((((unsigned long long)((unsigned)(i + b * ns) * 2654435761u)) * (unsigned long long)S) >> 32);
    int lq = 0;
    while (lq < 3 && (int)q >= g.start[lq + 1]) ++lq;
    const int rel = (int)q - g.start[lq], yq = rel / g.W[lq], xq = rel - yq * g.W[lq];
    const int H = g.H[l], W = g.W[l];
    const int cy = centre_floor(yq, g.H[lq], H), cx = centre_floor(xq, g.W[lq], W);
    float xy[8];
    if (MODE == 0) {
      const float *op = loc + ((long long)b * S + q) * loc_rs + (m * 4 + l) * 8;
      const float4 a = ld4(op), c = ld4(op + 4);
      const float2 r = *reinterpret_cast<const float2 *>(ref + (((long long)b * S + q) * 4 + l) * 2);
      const float o[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        xy[2 * k] = add_rn(r.x, __fdiv_rn(o[2 * k], (float)W));
        xy[2 * k + 1] = add_rn(r.y, __fdiv_rn(o[2 * k + 1], (float)H));
      }
    } else {
      const float *lp = loc + ((((long long)(b * M + m) * 4 + l) * S + q) * 4) * 2;
      const float4 a = ld4(lp), c = ld4(lp + 4);
      xy[0] = a.x; xy[1] = a.y; xy[2] = a.z; xy[3] = a.w; xy[4] = c.x; xy[5] = c.y; xy[6] = c.z; xy[7] = c.w;
    }
    int n = 0, up_y = 0, dn_y = 0, up_x = 0, dn_x = 0, sy = 0, sx = 0, qy = 0, qx = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const Tap<float> tp = make_tap<float>(xy[2 * k], xy[2 * k + 1], H, W);
      if (!tp.valid) continue;
      const int dy = max(-kPlanClip, min(kPlanClip, tp.h_low - cy)), dx = max(-kPlanClip, min(kPlanClip, tp.w_low - cx));
      ++n;
      up_y = max(up_y, dy + kPlanClip); dn_y = max(dn_y, kPlanClip - dy);
      up_x = max(up_x, dx + kPlanClip); dn_x = max(dn_x, kPlanClip - dx);
      sy += dy; sx += dx; qy += dy * dy; qx += dx * dx;
    }
    if (n) {
      int *rec = sh[m * 4 + l];
      atomicAdd(&rec[0], n);
      atomicMax(&rec[1], up_y); atomicMax(&rec[2], dn_y); atomicMax(&rec[3], up_x); atomicMax(&rec[4], dn_x);
      atomicAdd(&rec[5], sy); atomicAdd(&rec[6], sx); atomicAdd(&rec[7], qy); atomicAdd(&rec[8], qx);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < M * 4; i += kStatsThreads) {
    DirStats s;
    s.n = sh[i][0]; s.up_y = sh[i][1]; s.dn_y = sh[i][2]; s.up_x = sh[i][3]; s.dn_x = sh[i][4];
    s.sum_y = sh[i][5]; s.sum_x = sh[i][6]; s.pad = 0;
    s.sq_y = (unsigned long long)(unsigned)sh[i][7]; s.sq_x = (unsigned long long)(unsigned)sh[i][8];
    partial[(long long)blockIdx.x * M * 4 + i] = s;
  }
}

// ---- 2. per-head tables ----------------------------------------------------------------------------------------------------------
// run of query pixels c in [0, Nq) whose centre floor (in pixels of the extent-N level) lies in [lo, hi]
__device__ __forceinline__ void scan_run_dev(int Nq, int N, int lo, int hi, short &q0, short &qn) {
  // centre floor(c) = floor(((2 c + 1) N - Nq) / (2 Nq)) is monotone in c:
  //   >= lo  <=>  (2 c + 1) N >= Nq (2 lo + 1)      <= hi  <=>  (2 c + 1) N < Nq (2 hi + 3)
  int first = plan_floor_div_dev(Nq * (2 * lo + 1) - N + 2 * N - 1, 2 * N);           // ceil((Nq (2 lo + 1) - N) / (2 N))
  int last = plan_floor_div_dev(Nq * (2 * hi + 3) - N + 2 * N - 1, 2 * N) - 1;
  first = max(first, 0);
  last = min(last, Nq - 1);
  q0 = (short)(last < first ? 0 : first);
  qn = (short)(last < first ? 0 : last - first + 1);
}

__global__ __launch_bounds__(256) void dir_plan_kernel(const DirStats *__restrict__ partial, int n_partial, const PlanGeom g,
                                                       const RowPlan rp, HeadPlan *__restrict__ plans) {
  const int m = blockIdx.x, tid = threadIdx.x;
  __shared__ DirBounds s_win[4], s_near[4];
  __shared__ float s_mean[4][2];
  __shared__ RowAxis s_rax[kRowMaxAxisTiles];
  __shared__ int s_cand_max[4];
  HeadPlan &hp = plans[m];

  // ---- A. bounds: [min, max] of the sample, cut at mean +- kPlanSigmas sigma (one stray point must not size every window) ------
  // (the partial records are summed by all threads: one dependent global load per thread instead of n_partial per level)
  __shared__ int s_acc[4][8];                    // n, up_y, dn_y, up_x, dn_x, sum_y, sum_x, -
  __shared__ unsigned long long s_sq[4][2];
  if (tid < 32) (&s_acc[0][0])[tid] = 0;
  if (tid < 8) (&s_sq[0][0])[tid] = 0ull;
  __syncthreads();
  for (int i = tid; i < n_partial * 4; i += 256) {
    const int l = i & 3;
    const DirStats s = partial[((long long)(i >> 2) * g.M + m) * 4 + l];
    if (s.n) {
      atomicAdd(&s_acc[l][0], s.n);
      atomicMax(&s_acc[l][1], s.up_y); atomicMax(&s_acc[l][2], s.dn_y); atomicMax(&s_acc[l][3], s.up_x); atomicMax(&s_acc[l][4], s.dn_x);
      atomicAdd(&s_acc[l][5], s.sum_y); atomicAdd(&s_acc[l][6], s.sum_x);
      atomicAdd(&s_sq[l][0], s.sq_y); atomicAdd(&s_sq[l][1], s.sq_x);
    }
  }
  __syncthreads();
  if (tid < 4) {
    const int l = tid;
    const long long n = s_acc[l][0], sy = s_acc[l][5], sx = s_acc[l][6];
    const unsigned long long qy = s_sq[l][0], qx = s_sq[l][1];
    const int up_y = s_acc[l][1], dn_y = s_acc[l][2], up_x = s_acc[l][3], dn_x = s_acc[l][4];
    DirBounds b;
    float my = 0.f, mx = 0.f;
    if (n == 0) {
      b.ylo = b.xlo = (short)-g.default_halo;
      b.yhi = b.xhi = (short)(g.default_halo - 1);
    } else {
      my = (float)sy / (float)n; mx = (float)sx / (float)n;
      const float vy = fmaxf((float)qy / (float)n - my * my, 0.f), vx = fmaxf((float)qx / (float)n - mx * mx, 0.f);
      const float dy = kPlanSigmas * sqrtf(vy) + 0.5f, dx = kPlanSigmas * sqrtf(vx) + 0.5f;
      b.ylo = (short)max(kPlanClip - dn_y, (int)floorf(my - dy)); b.yhi = (short)min(up_y - kPlanClip, (int)ceilf(my + dy));
      b.xlo = (short)max(kPlanClip - dn_x, (int)floorf(mx - dx)); b.xhi = (short)min(up_x - kPlanClip, (int)ceilf(mx + dx));
    }
    s_mean[l][0] = my; s_mean[l][1] = mx;
    DirBounds nb;                         // scatter: the candidate tables hold |d| <= reach (both ends clamped into that range)
    const int R = g.reach;
    nb.ylo = (short)min(max((int)b.ylo, -R), R); nb.yhi = (short)max(min((int)b.yhi, R), -R);
    nb.xlo = (short)min(max((int)b.xlo, -R), R); nb.xhi = (short)max(min((int)b.xhi, R), -R);
    s_near[l] = nb;
    // windows: no extent beyond 32 pixels around the mean enters the fit below (nothing that wide fits the LDS budget)
    const int cy = (int)floorf(my), cx = (int)floorf(mx);
    b.ylo = (short)max((int)b.ylo, cy - 16); b.yhi = (short)min((int)b.yhi, cy + 16);
    b.xlo = (short)max((int)b.xlo, cx - 16); b.xhi = (short)min((int)b.xhi, cx + 16);
    if (b.yhi < b.ylo) b.yhi = b.ylo;
    if (b.xhi < b.xlo) b.xhi = b.xlo;
    s_win[l] = b;
  }
  __syncthreads();

  // (round 3 also fitted per-head LDS windows here -- 80 shrink-and-refit trips -- which no kernel read: directional windows on the
  // round-3 tiling measured the same time as isotropic ones; removed together with the HeadPlan fields that held them)
  if (tid < 4) { hp.win[tid] = s_win[tid]; hp.near[tid] = s_near[tid]; }
  if (!g.want_rows) return;

  // ---- C. row-tile scatter: per (level, axis tile) the runs of query pixels whose points can reach it ---------------------------
  int n_axis_total = 0;
#pragma unroll
  for (int l = 0; l < 4; ++l) n_axis_total += rp.n_ty[l] + rp.n_tx[l];
  for (int a = tid; a < n_axis_total; a += 256) {
    int l = 0;
    while (l < 3 && a >= rp.axis0[l + 1]) ++l;
    const bool is_y = a - rp.axis0[l] < rp.n_ty[l];
    RowAxis ra = rp.ax[a];                       // r0 / rn: the static tiling
    const DirBounds b = s_near[l];
    // a point with top-left pixel t = cf + d touches pixels t, t + 1: it reaches [r0, r0 + rn) iff t in [r0 - 1, r0 + rn - 1]
    const int dlo = is_y ? b.ylo : b.xlo, dhi = is_y ? b.yhi : b.xhi;
    const int lo = ra.r0 - 1 - dhi, hi = ra.r0 + ra.rn - 1 - dlo;
#pragma unroll
    for (int lq = 0; lq < 4; ++lq) scan_run_dev(is_y ? rp.H[lq] : rp.W[lq], is_y ? rp.H[l] : rp.W[l], lo, hi, ra.q0[lq], ra.qn[lq]);
    s_rax[a] = ra;
    hp.rax[a] = ra;
  }
  if (tid < 4) s_cand_max[tid] = 0;
  __syncthreads();
  int n_tiles_total = 0;
#pragma unroll
  for (int l = 0; l < 4; ++l) n_tiles_total += rp.n_ty[l] * rp.n_tx[l];
  for (int t = tid; t < n_tiles_total; t += 256) {
    int l = 0, tl = t;
    while (l < 3 && tl >= rp.n_ty[l] * rp.n_tx[l]) { tl -= rp.n_ty[l] * rp.n_tx[l]; ++l; }
    const int ty = tl / rp.n_tx[l], tx = tl - ty * rp.n_tx[l];
    const RowAxis ay = s_rax[rp.axis0[l] + ty], ax = s_rax[rp.axis0[l] + rp.n_ty[l] + tx];
    int c = 0;
#pragma unroll
    for (int lq = 0; lq < 4; ++lq) c += (int)ay.qn[lq] * (int)ax.qn[lq];
    atomicMax(&s_cand_max[l], c);
  }
  __syncthreads();
  if (tid == 0) {
    float work[4];
    for (int l = 0; l < 4; ++l) {
      const int nc = max(1, min(rp.n_chunks[l], (s_cand_max[l] + kRowChunkQueries - 1) / kRowChunkQueries));
      hp.n_chunks[l] = nc;
      work[l] = (float)s_cand_max[l] / (float)nc;
      hp.order[l] = l;
    }
    for (int i = 0; i < 4; ++i)                    // levels by work per item, heaviest first
      for (int j = i + 1; j < 4; ++j)
        if (work[hp.order[j]] > work[hp.order[i]]) { const int t = hp.order[i]; hp.order[i] = hp.order[j]; hp.order[j] = t; }
    hp.first_item[0] = 0;
    for (int i = 0; i < 4; ++i) {
      const int l = hp.order[i];
      hp.first_item[i + 1] = hp.first_item[i] + rp.n_ty[l] * rp.n_tx[l] * hp.n_chunks[l];
    }
    hp.n_items = hp.first_item[4];
  }
  __syncthreads();
  // the items, ready to use (RowItem): what scatter_rows_kernel would otherwise derive with a chain of dependent loads
  const int n_items = min(hp.n_items, kPlanMaxItems);
  for (int it = tid; it < n_items; it += 256) {
    int oi = 0;
    while (oi < 3 && it >= hp.first_item[oi + 1]) ++oi;
    const int l = hp.order[oi], local = it - hp.first_item[oi], nc = hp.n_chunks[l];
    const int chunk = local % nc, tile = local / nc;
    const int ty = tile / rp.n_tx[l], tx = tile - ty * rp.n_tx[l];
    const RowAxis ay = s_rax[rp.axis0[l] + ty], ax = s_rax[rp.axis0[l] + rp.n_ty[l] + tx];
    int n_cand = 0;
#pragma unroll
    for (int lq = 0; lq < 4; ++lq) n_cand += (int)ay.qn[lq] * (int)ax.qn[lq];
    RowItem d{};
    d.level = (short)l; d.chunk = (short)chunk; d.n_chunks = (short)nc;
    d.y0 = ay.r0; d.th = ay.rn; d.x0 = ax.r0; d.tw = ax.rn;
    d.c_begin = (int)((long long)n_cand * chunk / nc);
    d.c_end = (int)((long long)n_cand * (chunk + 1) / nc);
    d.cand_off = rp.cand_base[l] + tile * rp.cand_stride[l];
    d.near = s_near[l];
    hp.items[it] = d;
  }
}

// ---- 3. all of it in ONE launch (round 4) ---------------------------------------------------------------------------------------------
// The three kernels above ran back to back in front of every self-attention backward: 6.8 + 19.8 + 10.1 us plus two launch gaps, 4 % of
// the backward, most of it the latency of tiny dependent grids (8 workgroups planning, each reading 256 partial records).  Here
// workgroup (head m, tile) does everything ITS tile needs by itself:
//   a. the head's statistics from a sample of the call's points -- the same deterministic sample in every workgroup of the head
//      (integer sums: order-independent), so all of them derive IDENTICAL bounds without talking to each other; the sample is smaller
//      than dir_stats_kernel's (>= 4 k points per (head, level) instead of 16 k: every workgroup pays for it) -- the bounds stay hints;
//   b. the head's axis runs, tile counts, chunking and item order (a few hundred integers, recomputed per workgroup: cheaper than a
//      dependent launch);
//   c. its tile's RowItems and candidate list.  The workgroup of tile 0 also leaves the head-wide fields (bounds, item count).
// MODE as dir_stats_kernel.  grid = M * n_tiles, 256 threads.
constexpr int kFusedUnitsPerThread = 2;

// centre_floor (msda_common.h) with the hardware reciprocal: the estimate is within 1 of the quotient (|num| < 2^23, 1 ulp of rcp),
// the integer fix-up makes it exact -- a third of the instructions of the IEEE division; this kernel evaluates it ~10 times per thread
__device__ __forceinline__ int centre_floor_fast(int c, int Nq, int N) {
  const int num = (2 * c + 1) * N - Nq, den = 2 * Nq;
  int q = (int)floorf((float)num * __builtin_amdgcn_rcpf((float)den));
  int r = num - q * den;
  if (r < 0) { --q; r += den; }
  if (r < 0) { --q; r += den; }
  if (r >= den) { ++q; r -= den; }
  if (r >= den) ++q;
  return q;
}
// x / d for 0 <= x < 2^23, 0 < d < 2^12 (a level's pixel index by its width)
__device__ __forceinline__ int small_div(int x, int d) {
  int q = (int)((float)x * __builtin_amdgcn_rcpf((float)d));
  const int r = x - q * d;
  q += (r >= d) - (r < 0);
  return q;
}

template <int MODE>
__global__ __launch_bounds__(256) void plan_fused_kernel(const float *__restrict__ loc, const float *__restrict__ ref, const PlanGeom g,
                                                         const RowPlan rp, int B, int loc_rs, int n_tiles, HeadPlan *__restrict__ plans,
                                                         RowCandidate *__restrict__ table) {
  const int m = blockIdx.x / n_tiles, tile_g = blockIdx.x - m * n_tiles, tid = threadIdx.x;
  __shared__ int sh[4][10];                      // n, up_y, dn_y, up_x, dn_x, sum_y, sum_x, sq_y, sq_x, -
  __shared__ DirBounds s_near[4], s_win[4];
  __shared__ RowAxis s_rax[kRowMaxAxisTiles];
  __shared__ int s_cand_max[4], s_nchunks[4], s_order[4], s_first[5];
  HeadPlan &hp = plans[m];
  if (tid < 40) (&sh[0][0])[tid] = 0;
  if (tid < 4) s_cand_max[tid] = 0;
  __syncthreads();

  // ---- a. statistics of d = (h_low - cf_y, w_low - cf_x) for this head ---------------------------------------------------------------
  const int S = g.S, M = g.M;
  const int want = 256 * kFusedUnitsPerThread / 4;                      // sampled (batch element, query) pairs
  const int ns = max(1, min(S, (want + B - 1) / B));                  // queries per batch element
  const int total = B * ns * 4;
  for (int t = tid; t < total; t += 256) {
    const int l = t & 3, v = t >> 2;
    const int i = v % ns, b = v / ns;
    const unsigned q = (unsigned)((((unsigned long long)((unsigned)(i + b * ns) * 2654435761u)) * (unsigned long long)S) >> 32);
    int lq = 0;
    while (lq < 3 && (int)q >= g.start[lq + 1]) ++lq;
    const int rel = (int)q - g.start[lq], yq = small_div(rel, g.W[lq]), xq = rel - yq * g.W[lq];
    const int H = g.H[l], W = g.W[l];
    const int cy = centre_floor_fast(yq, g.H[lq], H), cx = centre_floor_fast(xq, g.W[lq], W);
    float xy[8];
    if (MODE == 0) {
      const float *op = loc + ((long long)b * S + q) * loc_rs + (m * 4 + l) * 8;
      const float4 a = ld4(op), c = ld4(op + 4);
      const float2 r = *reinterpret_cast<const float2 *>(ref + (((long long)b * S + q) * 4 + l) * 2);
      const float o[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        xy[2 * k] = add_rn(r.x, __fdiv_rn(o[2 * k], (float)W));
        xy[2 * k + 1] = add_rn(r.y, __fdiv_rn(o[2 * k + 1], (float)H));
      }
    } else {
      const float *lp = loc + ((((long long)(b * M + m) * 4 + l) * S + q) * 4) * 2;
      const float4 a = ld4(lp), c = ld4(lp + 4);
      xy[0] = a.x; xy[1] = a.y; xy[2] = a.z; xy[3] = a.w; xy[4] = c.x; xy[5] = c.y; xy[6] = c.z; xy[7] = c.w;
    }
    int n = 0, up_y = 0, dn_y = 0, up_x = 0, dn_x = 0, sy = 0, sx = 0, qy = 0, qx = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // (only the footprint's top-left pixel is needed: the scaling, the cuh:274 test and the floors of make_tap)
      const float h_im = scale_loc(xy[2 * k + 1], H), w_im = scale_loc(xy[2 * k], W);
      if (!((h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W))) continue;
      const int dy = max(-kPlanClip, min(kPlanClip, (int)floorf(h_im) - cy)), dx = max(-kPlanClip, min(kPlanClip, (int)floorf(w_im) - cx));
      ++n;
      up_y = max(up_y, dy + kPlanClip); dn_y = max(dn_y, kPlanClip - dy);
      up_x = max(up_x, dx + kPlanClip); dn_x = max(dn_x, kPlanClip - dx);
      sy += dy; sx += dx; qy += dy * dy; qx += dx * dx;
    }
    if (n) {
      int *rec = sh[l];
      atomicAdd(&rec[0], n);
      atomicMax(&rec[1], up_y); atomicMax(&rec[2], dn_y); atomicMax(&rec[3], up_x); atomicMax(&rec[4], dn_x);
      atomicAdd(&rec[5], sy); atomicAdd(&rec[6], sx); atomicAdd(&rec[7], qy); atomicAdd(&rec[8], qx);      // <= 16 k x 64^2: fits 32 bits
    }
  }
  __syncthreads();
  if (tid < 4) {                                 // bounds: [min, max] of the sample, cut at mean +- kPlanSigmas sigma (dir_plan_kernel, A)
    const int l = tid;
    const int n = sh[l][0];
    DirBounds b;
    if (n == 0) {
      b.ylo = b.xlo = (short)-g.default_halo;
      b.yhi = b.xhi = (short)(g.default_halo - 1);
    } else {
      const float my = (float)sh[l][5] / (float)n, mx = (float)sh[l][6] / (float)n;
      const float vy = fmaxf((float)(unsigned)sh[l][7] / (float)n - my * my, 0.f), vx = fmaxf((float)(unsigned)sh[l][8] / (float)n - mx * mx, 0.f);
      const float dy = kPlanSigmas * sqrtf(vy) + 0.5f, dx = kPlanSigmas * sqrtf(vx) + 0.5f;
      b.ylo = (short)max(kPlanClip - sh[l][2], (int)floorf(my - dy)); b.yhi = (short)min(sh[l][1] - kPlanClip, (int)ceilf(my + dy));
      b.xlo = (short)max(kPlanClip - sh[l][4], (int)floorf(mx - dx)); b.xhi = (short)min(sh[l][3] - kPlanClip, (int)ceilf(mx + dx));
    }
    s_win[l] = b;
    DirBounds nb;                                // scatter: the candidate tables hold |d| <= reach (both ends clamped into that range)
    const int R = g.reach;
    nb.ylo = (short)min(max((int)b.ylo, -R), R); nb.yhi = (short)max(min((int)b.yhi, R), -R);
    nb.xlo = (short)min(max((int)b.xlo, -R), R); nb.xhi = (short)max(min((int)b.xhi, R), -R);
    s_near[l] = nb;
  }
  __syncthreads();

  // ---- b. the head's axis runs, tile counts, chunking, item order (dir_plan_kernel, C) --------------------------------------------------
  int n_axis_total = 0;
#pragma unroll
  for (int l = 0; l < 4; ++l) n_axis_total += rp.n_ty[l] + rp.n_tx[l];
  for (int a = tid; a < n_axis_total; a += 256) {
    int l = 0;
    while (l < 3 && a >= rp.axis0[l + 1]) ++l;
    const bool is_y = a - rp.axis0[l] < rp.n_ty[l];
    RowAxis ra = rp.ax[a];
    const DirBounds b = s_near[l];
    const int dlo = is_y ? b.ylo : b.xlo, dhi = is_y ? b.yhi : b.xhi;
    const int lo = ra.r0 - 1 - dhi, hi = ra.r0 + ra.rn - 1 - dlo;
#pragma unroll
    for (int lq = 0; lq < 4; ++lq) scan_run_dev(is_y ? rp.H[lq] : rp.W[lq], is_y ? rp.H[l] : rp.W[l], lo, hi, ra.q0[lq], ra.qn[lq]);
    s_rax[a] = ra;
    if (tile_g == 0) hp.rax[a] = ra;
  }
  __syncthreads();
  for (int t = tid; t < n_tiles; t += 256) {
    int l = 0, tl = t;
    while (l < 3 && tl >= rp.n_ty[l] * rp.n_tx[l]) { tl -= rp.n_ty[l] * rp.n_tx[l]; ++l; }
    const int ty = tl / rp.n_tx[l], tx = tl - ty * rp.n_tx[l];
    const RowAxis ay = s_rax[rp.axis0[l] + ty], ax = s_rax[rp.axis0[l] + rp.n_ty[l] + tx];
    int c = 0;
#pragma unroll
    for (int lq = 0; lq < 4; ++lq) c += (int)ay.qn[lq] * (int)ax.qn[lq];
    atomicMax(&s_cand_max[l], c);
  }
  __syncthreads();
  if (tid == 0) {
    float work[4];
    for (int l = 0; l < 4; ++l) {
      const int nc = max(1, min(rp.n_chunks[l], (s_cand_max[l] + kRowChunkQueries - 1) / kRowChunkQueries));
      s_nchunks[l] = nc;
      work[l] = (float)s_cand_max[l] / (float)nc;
      s_order[l] = l;
    }
    for (int i = 0; i < 4; ++i)                    // levels by work per item, heaviest first
      for (int j = i + 1; j < 4; ++j)
        if (work[s_order[j]] > work[s_order[i]]) { const int t = s_order[i]; s_order[i] = s_order[j]; s_order[j] = t; }
    s_first[0] = 0;
    for (int i = 0; i < 4; ++i) {
      const int l = s_order[i];
      s_first[i + 1] = s_first[i] + rp.n_ty[l] * rp.n_tx[l] * s_nchunks[l];
    }
    if (tile_g == 0) {
      for (int l = 0; l < 4; ++l) { hp.n_chunks[l] = s_nchunks[l]; hp.order[l] = s_order[l]; hp.win[l] = s_win[l]; hp.near[l] = s_near[l]; }
      for (int i = 0; i < 5; ++i) hp.first_item[i] = s_first[i];
      hp.n_items = s_first[4];
    }
  }
  __syncthreads();

  // ---- c. this tile: its items (one per chunk) and its candidates (row_candidates_kernel) ----------------------------------------------
  int l = 0, t = tile_g;
  while (l < 3 && t >= rp.n_ty[l] * rp.n_tx[l]) { t -= rp.n_ty[l] * rp.n_tx[l]; ++l; }
  const int ty = t / rp.n_tx[l], tx = t - ty * rp.n_tx[l];
  const RowAxis ay = s_rax[rp.axis0[l] + ty], ax = s_rax[rp.axis0[l] + rp.n_ty[l] + tx];
  int n_cand = 0;
#pragma unroll
  for (int lq = 0; lq < 4; ++lq) n_cand += (int)ay.qn[lq] * (int)ax.qn[lq];
  const int nc = s_nchunks[l];
  int oi = 0;
  while (oi < 3 && s_order[oi] != l) ++oi;
  if (tid < nc) {
    const int it = s_first[oi] + t * nc + tid;
    if (it < kPlanMaxItems) {
      RowItem d{};
      d.level = (short)l; d.chunk = (short)tid; d.n_chunks = (short)nc;
      d.y0 = ay.r0; d.th = ay.rn; d.x0 = ax.r0; d.tw = ax.rn;
      d.c_begin = (int)((long long)n_cand * tid / nc);
      d.c_end = (int)((long long)n_cand * (tid + 1) / nc);
      d.cand_off = rp.cand_base[l] + t * rp.cand_stride[l];
      d.near = s_near[l];
      hp.items[it] = d;
    }
  }
  RowCandidate *out = table + (long long)m * rp.cand_total + rp.cand_base[l] + (long long)t * rp.cand_stride[l];
  int first = 0;
  for (int lq = 0; lq < 4; ++lq) {
    const int n = (int)ay.qn[lq] * (int)ax.qn[lq], w = ax.qn[lq];
    for (int k = tid; k < n; k += 256) {
      const int dy = small_div(k, w), dx = k - dy * w;
      const int yq = ay.q0[lq] + dy, xq = ax.q0[lq] + dx;
      RowCandidate c;
      c.token = rp.start[lq] + yq * rp.W[lq] + xq;
      c.cy = (short)centre_floor_fast(yq, rp.H[lq], rp.H[l]);
      c.cx = (short)centre_floor_fast(xq, rp.W[lq], rp.W[l]);
      out[first + k] = c;
    }
    first += n;
  }
}

}  // namespace msda
