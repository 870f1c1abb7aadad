// Tile-window geometry of the self-attention ("encoder") gather kernels, shared by host and device.
//
// When the queries ARE the value tokens (Lq == S: the encoder's deformable self-attention, depthaware_transformer.py
// :345-354 -> ms_deform_attn.py:122-162), query i sits at the pixel centre of token i and -- at initialisation, and for
// as long as the learned offsets stay a few pixels long (ms_deform_attn.py:106-114: |offset| <= n_points pixels) -- samples
// every level within a few pixels of its own position.  The image plane is cut into n_ty x n_tx tiles in NORMALISED
// coordinates; a workgroup takes the queries of all four levels whose pixel falls into its tile and stages, per level, the
// value window those queries can reach with offsets shorter than `halo` pixels.  Taps that leave the window are still
// correct: they are fetched from global memory (the window is a cache, never a contract).
//
// All of it is host-side integer arithmetic; the device receives the finished table (WinTable) as a kernel argument.
#pragma once
#include <stdint.h>

namespace msda {

constexpr int kWinLevels = 4;
#ifndef MSDA_WIN_THREADS_FWD
#define MSDA_WIN_THREADS_FWD 1024
#endif
constexpr int kWinThreads = MSDA_WIN_THREADS_FWD;    // forward: 16 waves, one workgroup per CU owns the LDS (measured: 12 waves 0.376 ms, 8 waves 0.411 ms against 0.323 with 16)
#ifndef MSDA_WIN_THREADS_BWD
#define MSDA_WIN_THREADS_BWD 768
#endif
constexpr int kWinThreadsBwd = MSDA_WIN_THREADS_BWD;  // backward: 12 waves = 170 VGPRs per lane (grad_out row + prefetch need ~157;
                                                      // measured 1.11 ms per backward with 8 waves, 1.07 with 12; 16 waves would spill)
constexpr int kWinPairsPerPass = kWinThreads / 8;    // 128 (query, head) pairs in flight
constexpr int kWinMaxPasses = 2;                     // queries of a tile <= 256
constexpr int kWinLdsBudget = 160 * 1024 - 128 - 256;    // - the zero row
constexpr int kWinMaxRows = kWinLdsBudget / 128;     // value rows (32 floats) the windows of one tile may hold
constexpr int kWinGoRows = kWinThreadsBwd / 64 * 8;   // backward: one 1-KiB block of grad_out rows per wave (its 8 pairs), at the end
constexpr int kWinMaxRowsBwd = kWinMaxRows - kWinGoRows;

struct WinGeom {
  int H[kWinLevels], W[kWinLevels], start[kWinLevels];
  int n_ty, n_tx;      // tiles
  int halo;            // window = query span +- halo pixels of the sampled level
};

// The geometry is separable: along one axis, tile t owns a run of query pixels of every level and needs a run of value
// pixels of every level.  The host tabulates both per (axis tile, level); the kernel takes the table as an argument and
// does no division.
struct AxisSpec { short w0, wn, q0, qn; };           // window start / extent, query start / extent (pixels of the level)
constexpr int kWinMaxAxisTiles = 64;                 // n_ty + n_tx
struct WinTable {
  int H[kWinLevels], W[kWinLevels], start[kWinLevels];
  int n_ty, n_tx;
  AxisSpec ax[kWinMaxAxisTiles][kWinLevels];         // [0, n_ty): rows of tiles, [n_ty, n_ty + n_tx): columns of tiles
};

// Per (tile, query slot, sampled level): the query's token and the floor of its centre at that level (centre_floor of
// msda_common.h, y << 16 | x & 0xFFFF).  Geometry only -- one table serves every (batch, head) plane; the kernels read it
// instead of re-deriving (level, row, column) from the slot number with compares and divisions in every lane.
struct WinQuery { int q, cf; };                      // q = -1: empty slot
constexpr int kWinMaxQueries = kWinPairsPerPass * kWinMaxPasses;

inline int floor_div(int a, int b) {                 // b > 0
  const int q = a / b, r = a - q * b;
  return r < 0 ? q - 1 : q;
}

// Along one axis with level extents N[l]: queries of level l owned by tile t of n_t = pixels [t N_l / n_t, (t+1) N_l / n_t)
// (a partition of the level); value window of level lv = the span of those queries' centres, mapped to level lv
// (a query at pixel c of level l sits at floor((c + 0.5) / N_l * N_lv - 0.5)), widened by `halo`, clipped to the level.
inline AxisSpec axis_spec(const int *N, int n_t, int t, int lv, int halo) {
  AxisSpec a;
  a.q0 = (short)(t * N[lv] / n_t);
  a.qn = (short)((t + 1) * N[lv] / n_t - a.q0);
  int lo = 1 << 30, hi = -(1 << 30);
  for (int l = 0; l < kWinLevels; ++l) {
    const int q0 = t * N[l] / n_t, q1 = (t + 1) * N[l] / n_t;
    if (q1 <= q0) continue;
    const int c = floor_div((2 * q0 + 1) * N[lv] - N[l], 2 * N[l]);
    const int d = floor_div((2 * (q1 - 1) + 1) * N[lv] - N[l], 2 * N[l]);
    lo = c < lo ? c : lo;
    hi = d > hi ? d : hi;
  }
  if (hi < lo) { a.w0 = 0; a.wn = 0; return a; }
  const int w0 = lo - halo < 0 ? 0 : lo - halo, w1 = hi + halo > N[lv] - 1 ? N[lv] - 1 : hi + halo;
  a.w0 = (short)w0;
  a.wn = (short)(w1 - w0 + 1);
  return a;
}

inline void fill_window_table(const WinGeom &g, WinTable &t) {
  for (int l = 0; l < kWinLevels; ++l) { t.H[l] = g.H[l]; t.W[l] = g.W[l]; t.start[l] = g.start[l]; }
  t.n_ty = g.n_ty;
  t.n_tx = g.n_tx;
  for (int ty = 0; ty < g.n_ty; ++ty)
    for (int l = 0; l < kWinLevels; ++l) t.ax[ty][l] = axis_spec(g.H, g.n_ty, ty, l, g.halo);
  for (int tx = 0; tx < g.n_tx; ++tx)
    for (int l = 0; l < kWinLevels; ++l) t.ax[g.n_ty + tx][l] = axis_spec(g.W, g.n_tx, tx, l, g.halo);
}

// [n_tiles][kWinMaxQueries][kWinLevels] entries; a tile's queries in level order, row-major within the level.
inline void fill_query_table(const WinTable &t, WinQuery *out) {
  for (int ty = 0; ty < t.n_ty; ++ty)
    for (int tx = 0; tx < t.n_tx; ++tx) {
      WinQuery *tile = out + (size_t)(ty * t.n_tx + tx) * kWinMaxQueries * kWinLevels;
      int i = 0;
      for (int l = 0; l < kWinLevels; ++l) {
        const AxisSpec ay = t.ax[ty][l], ax = t.ax[t.n_ty + tx][l];
        for (int dy = 0; dy < ay.qn; ++dy)
          for (int dx = 0; dx < ax.qn; ++dx, ++i) {
            const int y = ay.q0 + dy, x = ax.q0 + dx;
            for (int lv = 0; lv < kWinLevels; ++lv) {
              const int cy = floor_div((2 * y + 1) * t.H[lv] - t.H[l], 2 * t.H[l]);
              const int cx = floor_div((2 * x + 1) * t.W[lv] - t.W[l], 2 * t.W[l]);
              tile[i * kWinLevels + lv] = WinQuery{t.start[l] + y * t.W[l] + x, (int)(((unsigned)cy << 16) | ((unsigned)cx & 0xFFFFu))};
            }
          }
      }
      for (; i < kWinMaxQueries; ++i)
        for (int lv = 0; lv < kWinLevels; ++lv) tile[i * kWinLevels + lv] = WinQuery{-1, 0};
    }
}

// Host: choose the tiling.  Cost model (LDS cycles per tile, measured orders of magnitude): ~5 per staged row (L2 -> LDS),
// ~80 per (query, head) pair slot of a pass (16 points x 4 row reads + records), so halo rows are cheap next to idle pair
// slots.  Returns false when no tiling fits (tiny or huge levels): the caller keeps the plain gather kernels.
// tile_cost: the fixed part of a tile in the same units (a full pass of 128 pair slots = 10,240).  400 for the forward.  The BACKWARD
// gather pays much more per tile than its row loops -- a block of grad_out rows per wave, the softmax / offset backward epilogue, two
// more barriers -- and where the LDS budget forbids 256-query tiles it is faster on FEWER, partly filled two-pass tiles than on
// one-pass tiles: measured at 1920 x 1280 (config 5, B = 4; interior tiles carry the halo on all four sides: 1,304 rows for 256
// queries against a budget of 1,181) 10 x 30 tiles of <= 170 queries 1.510 ms per backward, 20 x 20 tiles of 128 queries (this
// model's choice with 400) 1.645 ms; the forward prefers the latter (0.436 against 0.485 ms).  18,000 reproduces both choices and
// leaves 1280 x 384 / 1408 x 376 (2 x 20 / 2 x 22 full tiles, the minimum count) as they are.
inline bool choose_window_tiling(const int64_t *shapes_host, const int64_t *lsi_host, int halo, int max_rows, WinGeom &best,
                                 int force_ty = 0, int force_tx = 0, double tile_cost = 400.0) {
  // force_ty x force_tx (> 0, measurement runs: MSDA_WIN_TILES): only that tiling is considered (still checked against the budgets)
  WinGeom g;
  for (int l = 0; l < kWinLevels; ++l) {
    g.H[l] = (int)shapes_host[2 * l];
    g.W[l] = (int)shapes_host[2 * l + 1];
    g.start[l] = (int)lsi_host[l];
    if (g.H[l] > 16384 || g.W[l] > 16384) return false;        // (2c + 1) * N fits 32 bits, pixels fit 16
  }
  g.halo = halo;
  double best_cost = -1;
  AxisSpec ys[kWinMaxAxisTiles][kWinLevels], xs[kWinMaxAxisTiles][kWinLevels];
  for (int n_ty = 1; n_ty <= g.H[0] && n_ty < kWinMaxAxisTiles; ++n_ty) {
    if (force_ty > 0 && n_ty != force_ty) continue;
    const int th = (g.H[0] + n_ty - 1) / n_ty;
    if (th > 64) continue;
    if (th < 4 && n_ty > 1) break;
    for (int ty = 0; ty < n_ty; ++ty)
      for (int l = 0; l < kWinLevels; ++l) ys[ty][l] = axis_spec(g.H, n_ty, ty, l, halo);
    for (int n_tx = 1; n_tx <= g.W[0] && n_ty + n_tx <= kWinMaxAxisTiles; ++n_tx) {
      if (force_tx > 0 && n_tx != force_tx) continue;
      const int tw = (g.W[0] + n_tx - 1) / n_tx;
      if (tw > 64) continue;
      if (tw < 4 && n_tx > 1) break;
      for (int tx = 0; tx < n_tx; ++tx)
        for (int l = 0; l < kWinLevels; ++l) xs[tx][l] = axis_spec(g.W, n_tx, tx, l, halo);
      double cost = 0;
      bool ok = true;
      for (int ty = 0; ty < n_ty && ok; ++ty)
        for (int tx = 0; tx < n_tx && ok; ++tx) {
          int rows = 0, queries = 0;
          for (int l = 0; l < kWinLevels; ++l) {
            rows += (ys[ty][l].wn * xs[tx][l].wn + 7) / 8 * 8;     // windows start on 8-row (1 KiB) boundaries
            queries += ys[ty][l].qn * xs[tx][l].qn;
          }
          if (rows > max_rows || queries > kWinPairsPerPass * kWinMaxPasses) { ok = false; break; }
          const int passes = (queries + kWinPairsPerPass - 1) / kWinPairsPerPass;
          cost += 5.0 * rows + 80.0 * passes * kWinPairsPerPass + tile_cost;      // + fixed per-tile overhead
        }
      if (ok && (best_cost < 0 || cost < best_cost)) { best_cost = cost; best = g; best.n_ty = n_ty; best.n_tx = n_tx; }
    }
  }
  return best_cost >= 0;
}

}  // namespace msda
