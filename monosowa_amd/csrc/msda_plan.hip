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

}  // namespace msda
