// Host-side plan of the row-tile scatter (msda_scatter_rows.hip): grad_value of the self-attention shape (Lq == S).
//
// Every level is cut into exclusive tiles of at most kRowTileRows value rows; a workgroup owns one tile of one
// (batch, head) and accumulates it in REGISTERS (2 lanes per row x 16 channels), so fine-level tiles are written with plain
// stores: no atomics, no zero fill.  The points that can reach a tile are found geometrically: along each axis, tile t of
// level l lists, per query level, the run of query pixels whose centre lies within `reach` + 1 pixels (of level l) of the
// tile -- separable, tabulated here, passed to the kernel as an argument.  Coarse levels, whose few rows receive a quarter
// of all points each, are split over several workgroups by candidate range ("chunks") and combined with row atomics into
// a zeroed region, as in the tile-owner generation.
#pragma once
#include <stdint.h>
#include <algorithm>

namespace msda {

#ifndef MSDA_ROW_CELLS
#define MSDA_ROW_CELLS 128
#endif
#ifndef MSDA_ROW_THREADS
#define MSDA_ROW_THREADS 512
#endif
constexpr int kRowTileRows = MSDA_ROW_CELLS;   // bilinear cells per tile: 512 threads = 128 cells x 4 lanes (8 channels x 4 corner sums each);
                                           // a tile of th x tw output rows has (th + 1) x (tw + 1) cells
#ifndef MSDA_ROW_CAP
#define MSDA_ROW_CAP 32
#endif
constexpr int kRowCellCap = MSDA_ROW_CAP;  // points one cell's bucket holds per batch and round
constexpr int kRowThreads = MSDA_ROW_THREADS;
#ifndef MSDA_ROW_SUB
#define MSDA_ROW_SUB 2
#endif
constexpr int kRowSub = MSDA_ROW_SUB;                 // candidate points per thread and batch
constexpr int kRowBatchQueries = kRowThreads / 4 * kRowSub;   // candidates x the level's 4 points per batch
constexpr int kRowMaxAxisTiles = 96;       // sum over levels of (n_ty + n_tx)
constexpr int kRowChunkQueries = 3584;     // candidates per workgroup (14 batches); longer scan lists are split

struct RowAxis {
  short r0, rn;            // the tile's rows (or columns) of its level
  short q0[4], qn[4];      // per query level: the run of query pixels scanned for this tile
};

struct RowPlan {
  int H[4], W[4], start[4];
  int th[4], tw[4], n_ty[4], n_tx[4], n_chunks[4];
  int axis0[4];            // index of level l's first y-tile in `ax`; its x-tiles follow the n_ty y-tiles
  int order[4], first_item[5];      // levels by work per item, heaviest first; prefix of items over `order`
  int n_items;
  int reach;               // a point is NEAR when its top-left corner is within `reach` pixels of the query's centre floor
  int cand_base[4], cand_stride[4];   // scan list of tile t of level l: table[cand_base[l] + t * cand_stride[l] ...)
  int cand_total;
  RowAxis ax[kRowMaxAxisTiles];
};

// One candidate of a tile's scan list (built on the device once per call, read by every (batch, head)):
// the query's token and its centre floor in pixels of the tile's level.
struct RowCandidate { int token; short cy, cx; };

inline int plan_floor_div(int a, int b) {
  const int q = a / b, r = a - q * b;
  return r < 0 ? q - 1 : q;
}

// queries of extent-Nq level whose centre floor (in pixels of the extent-N level) lies in [lo, hi]
inline void scan_run(int Nq, int N, int lo, int hi, short &q0, short &qn) {
  int first = -1, last = -2;
  for (int c = 0; c < Nq; ++c) {
    const int cf = plan_floor_div((2 * c + 1) * N - Nq, 2 * Nq);
    if (cf >= lo && cf <= hi) { if (first < 0) first = c; last = c; }
  }
  q0 = (short)(first < 0 ? 0 : first);
  qn = (short)(first < 0 ? 0 : last - first + 1);
}

// Returns false when the shape does not fit the tables (the caller keeps the tile-owner kernels).
inline bool make_row_plan(const int64_t *shapes_host, const int64_t *lsi_host, int reach, RowPlan &p) {
  p = RowPlan{};
  p.reach = reach;
  int n_axis = 0;
  double work[4];
  for (int l = 0; l < 4; ++l) {
    const int H = (int)shapes_host[2 * l], W = (int)shapes_host[2 * l + 1];
    if (H > 2048 || W > 2048) return false;                  // centre_floor's float estimate (msda_common.h)
    p.H[l] = H; p.W[l] = W; p.start[l] = (int)lsi_host[l];
  }
  for (int l = 0; l < 4; ++l) {
    const int H = p.H[l], W = p.W[l];
    // tiles whose cells -- (th + 1) x (tw + 1): one apron row above, one apron column to the left -- fit kRowTileRows: the
    // split that scans the fewest candidates, i.e. minimises the summed area of the tiles grown by a typical scan margin
    int tw = W, th = H;
    if ((H + 1) * (W + 1) > kRowTileRows) {
      const int mg = 8;
      long long best = -1;
      for (int nty = 1; nty <= H; ++nty) {
        const int h = (H + nty - 1) / nty, w_max = kRowTileRows / (h + 1) - 1;
        if (w_max < 1) continue;
        const int ntx = (W + w_max - 1) / w_max, w = (W + ntx - 1) / ntx;
        const long long cost = (long long)nty * ntx * (h + mg) * (w + mg);
        if (best < 0 || cost < best) { best = cost; th = h; tw = w; }
      }
    }
    p.n_tx[l] = (W + tw - 1) / tw; p.tw[l] = (W + p.n_tx[l] - 1) / p.n_tx[l];
    p.n_ty[l] = (H + th - 1) / th; p.th[l] = (H + p.n_ty[l] - 1) / p.n_ty[l];
    p.axis0[l] = n_axis;
    if (n_axis + p.n_ty[l] + p.n_tx[l] > kRowMaxAxisTiles) return false;
    long long cand_max = 0;
    for (int axis = 0; axis < 2; ++axis) {
      const int n_t = axis == 0 ? p.n_ty[l] : p.n_tx[l], t_ext = axis == 0 ? p.th[l] : p.tw[l], N = axis == 0 ? H : W;
      for (int t = 0; t < n_t; ++t) {
        RowAxis &a = p.ax[n_axis++];
        a.r0 = (short)(t * t_ext);
        a.rn = (short)std::min(t_ext, N - t * t_ext);
        // a near point's top-left corner is within `reach` of the centre floor and must land in [r0 - 1, r0 + rn - 1]
        const int lo = a.r0 - 1 - reach, hi = a.r0 + a.rn - 1 + reach;
        for (int lq = 0; lq < 4; ++lq) scan_run(axis == 0 ? p.H[lq] : p.W[lq], N, lo, hi, a.q0[lq], a.qn[lq]);
      }
    }
    // scan lists: per tile the candidates of all four query levels, one after another
    for (int ty = 0; ty < p.n_ty[l]; ++ty)
      for (int tx = 0; tx < p.n_tx[l]; ++tx) {
        long long c = 0;
        for (int lq = 0; lq < 4; ++lq) c += (long long)p.ax[p.axis0[l] + ty].qn[lq] * p.ax[p.axis0[l] + p.n_ty[l] + tx].qn[lq];
        cand_max = std::max(cand_max, c);
      }
    p.cand_base[l] = p.cand_total;
    p.cand_stride[l] = (int)cand_max;
    p.cand_total += (int)cand_max * p.n_ty[l] * p.n_tx[l];
    p.n_chunks[l] = (int)std::max(1LL, std::min(16LL, (cand_max + kRowChunkQueries - 1) / kRowChunkQueries));
    work[l] = (double)cand_max / p.n_chunks[l];
    p.order[l] = l;
  }
  std::sort(p.order, p.order + 4, [&](int a, int b) { return work[a] > work[b]; });
  p.first_item[0] = 0;
  for (int i = 0; i < 4; ++i) {
    const int l = p.order[i];
    p.first_item[i + 1] = p.first_item[i] + p.n_ty[l] * p.n_tx[l] * p.n_chunks[l];
  }
  p.n_items = p.first_item[4];
  return true;
}

inline size_t row_plan_table_bytes(const RowPlan &p) { return (size_t)p.cand_total * sizeof(RowCandidate); }

}  // namespace msda
