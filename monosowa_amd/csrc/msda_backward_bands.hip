// grad_value of the cross-attention shape (few queries: the decoder's 550 / 50), d32 path: ROW BANDS of one level of one
// (batch, head), all of that plane's grad_out rows resident in LDS, sums in registers.
//
// The tile-owner scatter (msda_backward_tiled.hip, K2) gives every 16 x 16 tile of every level its own workgroup and adds
// every (corner, channel) with a 64-bit fixed-point LDS atomic: 144 M ds_add_u64 per launch at B = 16, Lq = 550 -- measured
// 181 us, of which ~160 are the atomics (a build without them: 20 us): the coarse levels' 600 rows take half of all additions,
// eight at a time from one wave onto the same few rows.
// Here a workgroup (16 waves, the CU's whole LDS) takes a run of BANDS -- kBandRows consecutive value rows of a level -- of one
// (batch, head):
//   once:      the plane's grad_out rows (Lq x 128 B <= 72 KB) into LDS, this thread's <= 4 records of the level's list (K1's
//              {h_im, w_im}, {attn_w}) into registers with their taps resolved;
//   per band:  (1) every corner that falls into the band takes a rank in its ROW's list (one ds_add_rtn_u32 per corner: 1/32
//              of the per-channel atomics, and 32-bit integer LDS atomics run 1.6x the 64-bit rate), (2) a prefix sum over the
//              band's row counts turns ranks into list positions, (3) the corner's {weight, query} goes there, (4) eight lanes
//              per row (4 channels each) walk the row's list -- ds_read_b64 of the entry, ds_read_b128 of the grad_out row,
//              4 FMAs -- and store the row: plain 128-byte rows on every level, no zero fill, no global atomics, no
//              fixed-point scale.
// Semantics: cuh:125-152 (the four corner additions of a point), restated in msda_common.h.
#include "msda_common.h"

namespace msda {

constexpr int kBandThreads = 1024;
constexpr int kBandWaves = kBandThreads / 64;
constexpr int kBandMaxQueries = 576;                 // grad_out rows held in LDS (72 KB)
constexpr int kBandPoints = 4;                       // records per thread: Lq * P <= 4096
constexpr int kBandRows = 2048;                      // rows per band (two per thread in the prefix sum); 16-bit counters
constexpr int kBandMaxHits = kBandMaxQueries * 16;   // corner entries of one band: at most every corner of the level's list (72 KB)
#ifndef MSDA_BANDS_PER_ITEM
#define MSDA_BANDS_PER_ITEM 2
#endif
constexpr int kBandsPerItem = MSDA_BANDS_PER_ITEM;   // bands one workgroup walks (level 0 of 1280 x 384: 4 bands = 2 workgroups)

struct BandPlan {
  int H[4], W[4], start[4];
  int first_item[5];                                 // items of level l: [first_item[l], first_item[l + 1])
  int n_items;
};

__global__ __launch_bounds__(kBandThreads) void bwd_scatter_bands_kernel(
    const float2 *__restrict__ rec_hw, const float *__restrict__ rec_aw, const float *__restrict__ grad_out,
    float *__restrict__ grad_value, const BandPlan plan, int B, int S, int M, int Lq, const unsigned char *__restrict__ vmask) {
  __shared__ float4 go_lds[kBandMaxQueries * 8];               // 72 KB: grad_out[b, :, m, :]
  __shared__ uint2 entries[kBandMaxHits + 1];                      // 72 KB: {weight bits, query} of the band's corners, row after row
  __shared__ unsigned count[kBandRows];                        // 8 KB (32-bit: LDS atomics)
  __shared__ unsigned short base[kBandRows];                   // 4 KB: first entry of the row's list (< kBandMaxHits = 9216)
  __shared__ unsigned wave_tot[kBandWaves];

  // blockIdx -> (batch * head, item); all items of one (batch, head) share blockIdx % 8, i.e. one XCD (speed only)
  const int bm = (int)(blockIdx.x % 8) + 8 * (int)(blockIdx.x / (8 * plan.n_items));
  if (bm >= B * M) return;
  const int it = (int)((blockIdx.x / 8) % plan.n_items);
  const int b = bm / M, m = bm - b * M;
  int l = 0;
  while (l < 3 && it >= plan.first_item[l + 1]) ++l;
  const int H = plan.H[l], W = plan.W[l], n_rows_level = H * W;
  const int band0 = (it - plan.first_item[l]) * kBandsPerItem;
  const int n_bands = min(kBandsPerItem, (n_rows_level + kBandRows - 1) / kBandRows - band0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // ---- once: grad_out rows of this (batch, head) -> LDS (chunk c of query q at q * 8 + c) -----------------------------------
  for (int i = tid; i < Lq * 8; i += kBandThreads) {
    const int q = i >> 3, c = i & 7;
    go_lds[i] = ld4(grad_out + (((long long)b * Lq + q) * M + m) * 32 + c * 4);
  }
  // ---- once: this thread's records, taps resolved (rows of the level, -1: corner dropped / point invalid) ---------------------
  const long long n_pts = (long long)Lq * 4;
  const long long list = ((long long)bm * 4 + l) * n_pts;
  int rows[kBandPoints][4];
  float wts[kBandPoints][4];
  int qry[kBandPoints];
#pragma unroll
  for (int k = 0; k < kBandPoints; ++k) {
    const long long idx = tid + (long long)k * kBandThreads;
#pragma unroll
    for (int c = 0; c < 4; ++c) { rows[k][c] = -1; wts[k][c] = 0.f; }
    qry[k] = 0;
    if (idx < n_pts) {
      const float2 hw = rec_hw[list + idx];
      const float aw = rec_aw[list + idx];
      qry[k] = (int)(idx >> 2);
      if (hw.x > kInvalidCoord) {
        const Tap<float> tp = make_tap_im(hw.x, hw.y, H, W);
        if (tp.t && tp.l) rows[k][0] = tp.y0 * W + tp.x0;
        if (tp.t && tp.r) rows[k][1] = tp.y0 * W + tp.x1;
        if (tp.b && tp.l) rows[k][2] = tp.y1 * W + tp.x0;
        if (tp.b && tp.r) rows[k][3] = tp.y1 * W + tp.x1;
        wts[k][0] = tp.w1 * aw; wts[k][1] = tp.w2 * aw; wts[k][2] = tp.w3 * aw; wts[k][3] = tp.w4 * aw;
      }
    }
  }
  const long long tok0 = (long long)b * S + plan.start[l];

  for (int bi = 0; bi < n_bands; ++bi) {
    const int row0 = (band0 + bi) * kBandRows, n_rows = min(kBandRows, n_rows_level - row0);
    count[tid] = 0;
    count[tid + kBandThreads] = 0;
    __syncthreads();                                   // (first trip: also the grad_out rows; later: the previous walk is done)
    // ---- (1) ranks ---------------------------------------------------------------------------------------------------------
    unsigned rank[kBandPoints][4];
#pragma unroll
    for (int k = 0; k < kBandPoints; ++k)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int r = rows[k][c] - row0;
        rank[k][c] = 0xFFFFFFFFu;
        if (rows[k][c] >= 0 && (unsigned)r < (unsigned)n_rows) rank[k][c] = atomicAdd(&count[r], 1u);
      }
    __syncthreads();
    // ---- (2) exclusive prefix sum of the row counts: two rows per thread, a shuffle scan per wave, the waves' totals through LDS
    static_assert(kBandRows == 2 * kBandThreads, "two rows per thread");
    const unsigned s0 = 2 * tid < n_rows ? count[2 * tid] : 0u, s1 = 2 * tid + 1 < n_rows ? count[2 * tid + 1] : 0u;
    unsigned incl = s0 + s1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned up = __shfl_up(incl, o);
      if (lane >= o) incl += up;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    {
      unsigned off = 0;
      for (int w = 0; w < wave; ++w) off += wave_tot[w];
      const unsigned excl = off + incl - (s0 + s1);
      base[2 * tid] = (unsigned short)excl;
      base[2 * tid + 1] = (unsigned short)(excl + s0);
    }
    __syncthreads();
    // ---- (3) entries ---------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < kBandPoints; ++k)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (rank[k][c] != 0xFFFFFFFFu)
          entries[base[rows[k][c] - row0] + rank[k][c]] = make_uint2(__float_as_uint(wts[k][c]), (unsigned)qry[k]);
    __syncthreads();
    // ---- (4) eight lanes per row walk its list and store the row -------------------------------------------------------------------
    for (int task = tid; task < n_rows * 8; task += kBandThreads) {
      const int r = task >> 3, c = task & 7;
      const unsigned n = count[r], b0 = base[r];
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      uint2 e_next = entries[b0];                      // entry i + 1 is requested before entry i's row (b0 + n < kBandMaxHits + 1: see below)
      for (unsigned i = 0; i < n; ++i) {
        const uint2 e = e_next;
        e_next = entries[b0 + i + 1];
        const float w = __uint_as_float(e.x);
        const float4 g = go_lds[e.y * 8 + c];
        a.x += w * g.x; a.y += w * g.y; a.z += w * g.z; a.w += w * g.w;
      }
      const long long token = tok0 + row0 + r;
      if (vmask && vmask[token]) a = make_float4(0.f, 0.f, 0.f, 0.f);
      st4(grad_value + (token * M + m) * 32 + c * 4, a);
    }
    // (the next trip's first barrier orders this walk before the counters are cleared and the entries rewritten:
    // count[] is cleared BEFORE that barrier, by threads that may still be ahead of a slow walker -- so one more barrier here)
    __syncthreads();
  }
}

inline bool make_band_plan(const int64_t *shapes_host, const int64_t *lsi_host, int Lq, BandPlan &p) {
  if (Lq > kBandMaxQueries || (long long)Lq * 4 > (long long)kBandPoints * kBandThreads) return false;
  p.first_item[0] = 0;
  for (int l = 0; l < 4; ++l) {
    p.H[l] = (int)shapes_host[2 * l]; p.W[l] = (int)shapes_host[2 * l + 1]; p.start[l] = (int)lsi_host[l];
    const long long rows = (long long)p.H[l] * p.W[l];
    if (rows > (1 << 24)) return false;
    const int bands = (int)((rows + kBandRows - 1) / kBandRows);
    p.first_item[l + 1] = p.first_item[l] + (bands + kBandsPerItem - 1) / kBandsPerItem;
  }
  p.n_items = p.first_item[4];
  return true;
}

}  // namespace msda
