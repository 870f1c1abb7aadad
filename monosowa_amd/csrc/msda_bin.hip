// Exact scan lists for the cell scatter (msda_scatter_rows.hip) from the forward's saved sampling locations: which (query, level)
// units have a point in which tile.
//
// The geometric scan (msda_scatter_plan.h / msda_plan.h) lets a tile look at every query whose pixel lies within the head's
// bounds of it -- 132 k candidates per (batch, head) plane at 1280 x 384 where 57 k (query, level, tile) incidences exist, two
// thirds of the tap arithmetic for points that land elsewhere, and every point beyond the bounds left to global atomics.  Here one
// streaming pass over the saved locations (level-major [B, M, L, Lq, P, 2]: 32 contiguous bytes per unit) resolves each unit's
// four footprints ONCE and appends the unit -- its query and a 4-bit mask of the points concerned -- to the list of every tile
// whose cells [y0 - 1, y0 + th) x [x0 - 1, x0 + tw) hold one of them (1.4 lists per unit on average: cells on a tile's apron
// belong to its neighbour as well).  The scatter then scans exactly its tile's list: no bounds, no near / far classes, any offset
// distribution.  Appends are aggregated per wave (lanes are neighbouring queries: one or two distinct tiles per wave) -- one
// returning atomic per (wave, tile).  A list that is full (capacity: twice the uniform share) sends the unit's points to the
// gather kernel's row-atomic path through `far_mask`.
#include "msda_common.h"

namespace msda {

struct BinPlan {
  int H[4], W[4];
  int th[4], tw[4], n_ty[4], n_tx[4];     // output tiles of level l: uniform extents (the last one may be smaller)
  int tile0[4];                           // global index of level l's first tile (counters are [B * M][n_tiles])
  int cap[4];                             // entries one list of level l holds
  int list_off[4];                        // entry offset of level l's first list within a plane
  int n_tiles, plane_entries;
};

__global__ __launch_bounds__(256) void bin_points_kernel(const float *__restrict__ loc, const BinPlan p, long long n_units, int S,
                                                         unsigned *__restrict__ counts, unsigned *__restrict__ lists,
                                                         unsigned char *__restrict__ far_mask) {
  const long long u = (long long)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool live = u < n_units;
  const long long uu = live ? u : n_units - 1;
  const int q = (int)(uu % S);
  const int l = (int)((uu / S) & 3);
  const long long bm = uu / (4LL * S);
  const int H = p.H[l], W = p.W[l], th = p.th[l], tw = p.tw[l];
  const float4 a = ld4(loc + uu * 8), c = ld4(loc + uu * 8 + 4);
  const float xs[4] = {a.x, a.z, c.x, c.z}, ys[4] = {a.y, a.w, c.y, c.w};
  int cy[4], cx[4];
  unsigned valid = 0;
  int ty_lo = 1 << 20, ty_hi = -1, tx_lo = 1 << 20, tx_hi = -1;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float h_im = scale_loc(ys[k], H), w_im = scale_loc(xs[k], W);
    const bool ok = live && (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W);       // cuh:274
    cy[k] = ok ? (int)floorf(h_im) : 0;
    cx[k] = ok ? (int)floorf(w_im) : 0;
    if (ok) {
      valid |= 1u << k;
      // the cell feeds output rows cy and cy + 1 (columns cx, cx + 1) where they exist
      ty_lo = min(ty_lo, max(cy[k], 0) / th); ty_hi = max(ty_hi, min(cy[k] + 1, H - 1) / th);
      tx_lo = min(tx_lo, max(cx[k], 0) / tw); tx_hi = max(tx_hi, min(cx[k] + 1, W - 1) / tw);
    }
  }
  const int n_y = valid ? ty_hi - ty_lo + 1 : 0, n_x = valid ? tx_hi - tx_lo + 1 : 0;
  int trips = n_y * n_x;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) trips = max(trips, __shfl_xor(trips, o));            // wave-uniform trip count
  unsigned far = 0;
  const unsigned *cnt = counts + bm * p.n_tiles + p.tile0[l];
  unsigned *lst = lists + bm * (long long)p.plane_entries + p.list_off[l];
  const int cap = p.cap[l];
  for (int it = 0; it < trips; ++it) {
    int tile = -1;
    unsigned mask = 0;
    if (it < n_y * n_x) {
      const int ty = ty_lo + it / n_x, tx = tx_lo + it % n_x;
      const int y0 = ty * th - 1, y1 = min((ty + 1) * th, H) - 1, x0 = tx * tw - 1, x1 = min((tx + 1) * tw, W) - 1;     // the tile's cells
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if ((valid >> k & 1) && cy[k] >= y0 && cy[k] <= y1 && cx[k] >= x0 && cx[k] <= x1) mask |= 1u << k;
      if (mask) tile = ty * p.n_tx[l] + tx;
    }
    // wave-aggregated append: one returning atomic per distinct tile of this trip
    unsigned long long todo = __ballot(tile >= 0);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int key = __shfl(tile, leader);
      // (lanes of one wave may belong to different planes / levels at a boundary: the list address is part of the key)
      const unsigned long long same = __ballot(tile == key && cnt == (const unsigned *)__shfl((unsigned long long)cnt, leader));
      unsigned base = 0;
      if (lane == leader) base = atomicAdd(const_cast<unsigned *>(cnt) + key, (unsigned)__popcll(same));
      base = __shfl(base, leader);
      if (same >> lane & 1) {
        const unsigned idx = base + (unsigned)__popcll(same & ((1ull << lane) - 1ull));
        if (idx < (unsigned)cap) lst[(long long)key * cap + idx] = (unsigned)q | (mask << 28);
        else far |= mask;
      }
      todo &= ~same;
    }
  }
  if (live) far_mask[u] = (unsigned char)far;
}

}  // namespace msda
