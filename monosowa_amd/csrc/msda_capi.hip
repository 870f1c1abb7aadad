// C-ABI entry points (include/monosowa_msda.h): argument checks, kernel selection, launch.
// Replaces the reference host launchers ms_deform_attn_cuda_forward / _backward
// (ops/src/cuda/ms_deform_attn_cuda.cu:20-153).  The reference slices the batch into
// `im2col_step` chunks only to bound its 32-bit indexing; indexing here is 64-bit, so one
// launch covers the whole batch and the chunking precondition lives in the Python shim.
#include "../../include/monosowa_msda.h"
#include "msda_kernels.hip"
#include "msda_backward_tiled.hip"
#include "msda_backward_sorted.hip"
#include "msda_backward_bands.hip"
#include "msda_gather_rec.hip"
#include "msda_gather_win.hip"
#include "msda_scatter_rows.hip"
#include "msda_plan.hip"
#include <map>
#include <mutex>
#include <cstdlib>
#include <string>
#include <algorithm>
#include <vector>

namespace {

inline int grid_for(long long work_items, int items_per_block) {
  long long g = (work_items + items_per_block - 1) / items_per_block;
  const long long cap = 256LL * 64;   // 256 CUs x 64 resident-or-queued blocks; grid-stride beyond
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// exact grid for the 8-lanes-per-pair kernels: 32 pairs per workgroup, padded to a multiple of 8 workgroups
inline int grid_pairs(long long n_pairs) { return (int)(((n_pairs + 31) / 32 + 7) / 8 * 8); }

inline int check_dims(int B, int S, int M, int D, int L, int Lq, int P) {
  if (B <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq <= 0 || P <= 0) return MSDA_E_SHAPE;
  // per-sample offsets are kept in 32 bits inside the d32 kernel
  if ((long long)S * M * D >= (1LL << 31)) return MSDA_E_SHAPE;
  if ((long long)B * Lq * M >= (1LL << 35)) return MSDA_E_SHAPE;      // exact grids: (pairs / 32) workgroups
  return 0;
}

// MSDA_GATHER = 0: first-generation gather kernels; 1: tap records; 2 (default): records + coarse levels in LDS.
// A tuning / A-B switch only; every mode computes the same function.
// Kernel-generation switches (A/B measurement and test coverage; every setting computes the same function).
//   gather        0: first-generation gather kernels, 1: tap records, 2 (default): records + coarse levels in LDS
//   scatter_sorted 0 (default): never, 1: counting sort by bilinear cell + register sums (msda_backward_sorted.hip)
//                  when a level's record list has >= 8192 points, 2: always.  Measured at B = 16, Lq = 10200:
//                  0.90 vs 1.04 ms on random +-4 px offsets (tools/msda_kernel_bench.py) but 2.02 vs 1.88 ms per
//                  backward inside the train step, whose freshly initialised offsets are whole pixels with ~5
//                  points per cell -- hence opt-in.  Otherwise LDS-atomic tile accumulators, flavour by scatter_fixed:
//   scatter_fixed 1 (default): 64-bit fixed-point tile accumulators (ds_add_u64), 0: double (ds_add_f64)
// Initialised from MSDA_GATHER / MSDA_SCATTER_FIXED, changed at run time with msda_set_option().
struct Options {
  int gather = 2;
  int scatter_fixed = 1;
  int scatter_sorted = 0;
  int gather_chunks_fwd = 16, gather_chunks_bwd = 8;      // query slices per (batch, head) of the staged gather kernels
  int window = 3;              // tile-window gather kernels when Lq == S (msda_gather_win.hip): bit 0 forward, bit 1 backward
  int window_halo = 5;         // pixels of the sampled level a window extends beyond its tile's queries
  int window_persistent = 1;   // 1 (default): one workgroup per CU walks the items, inputs prefetched a unit ahead; 0: one per item
  int scatter_rows = 1;        // 1 (default): row-tile scatter (msda_scatter_rows.hip) with the window backward when Lq == S
  int scatter_reach = 6;       // its near-point reach in pixels (farther points: global atomics in the gather kernel)
  int directional = 1;         // 1 (default): per-head directional bounds measured on the call's own offsets (msda_plan.h) size the
                               // scatter's scan regions (and the windows); 0: isotropic reach / halo for every head
  int scatter_lists = 0;       // 1: exact scan lists from the saved locations (msda_bin.hip) for the saved backward instead of the
                               // geometric scan behind the directional plan.  Opt-in: measured at B = 16 the binning pass costs 273 us and
                               // the list-driven scatter 560 us (its workgroups have ONE batch of work each: the per-workgroup load chain
                               // dominates) against 35 + 473 us -- it only wins beyond sigma = 8 px (2.35 vs 2.65 ms)
  int scatter_lists_cap = 0;   // > 0: capacity of every exact scan list (tests: forces list overflow -> the far path); 0: sized from Lq
  int scatter_bands = 1;       // 1 (default): row-band scatter (msda_backward_bands.hip) for short record lists (Lq <= 576: the decoder)
  int plan_fused = 1;          // 1 (default): statistics + per-head plan + candidate tables in one launch (plan_fused_kernel); 0: three kernels
  int plan_reach = 8;          // capacity of the directional scan: |footprint - centre| beyond this many pixels is "far" in any case
  Options() {                                               // the environment is read ONCE, at first use
    if (const char *e = std::getenv("MSDA_GATHER")) gather = std::atoi(e);
    if (const char *e = std::getenv("MSDA_SCATTER_FIXED")) scatter_fixed = std::atoi(e) != 0;
    if (const char *e = std::getenv("MSDA_SCATTER_SORTED")) scatter_sorted = std::atoi(e);
    if (const char *e = std::getenv("MSDA_GATHER_CHUNKS")) gather_chunks_fwd = std::max(1, std::atoi(e));
    if (const char *e = std::getenv("MSDA_GATHER_CHUNKS_BWD")) gather_chunks_bwd = std::max(1, std::atoi(e));
    if (const char *e = std::getenv("MSDA_WINDOW")) window = std::atoi(e) & 3;
    if (const char *e = std::getenv("MSDA_WINDOW_PERSISTENT")) window_persistent = std::atoi(e) != 0;
    if (const char *e = std::getenv("MSDA_SCATTER_ROWS")) scatter_rows = std::atoi(e) != 0;
    if (const char *e = std::getenv("MSDA_SCATTER_REACH")) scatter_reach = std::min(16, std::max(1, std::atoi(e)));
    if (const char *e = std::getenv("MSDA_WINDOW_HALO")) window_halo = std::min(32, std::max(0, std::atoi(e)));
    if (const char *e = std::getenv("MSDA_DIRECTIONAL")) directional = std::atoi(e) != 0;
    if (const char *e = std::getenv("MSDA_SCATTER_BANDS")) scatter_bands = std::atoi(e) != 0;
    if (const char *e = std::getenv("MSDA_SCATTER_LISTS")) scatter_lists = std::atoi(e) != 0;
    if (const char *e = std::getenv("MSDA_PLAN_FUSED")) plan_fused = std::atoi(e) != 0;
    if (const char *e = std::getenv("MSDA_PLAN_REACH")) plan_reach = std::min(16, std::max(1, std::atoi(e)));
  }
};
inline Options &options() {
  static Options o;
  return o;
}
inline bool scatter_fixed_point() { return options().scatter_fixed != 0; }
inline int gather_mode() { return options().gather; }
inline int gather_chunks_fwd() { return options().gather_chunks_fwd; }
inline int gather_chunks_bwd() { return options().gather_chunks_bwd; }

// A host copy of the pyramid is caller-provided (a Python attribute in the shim): before any kernel trusts it, the
// levels must be positive and tile [0, S) in order -- otherwise a kernel would read past the value tensor.
inline int check_host_geometry(const int64_t *shapes_host, const int64_t *lsi_host, int L, int S) {
  long long next = 0;
  for (int l = 0; l < L; ++l) {
    const long long H = shapes_host[2 * l], W = shapes_host[2 * l + 1];
    if (H <= 0 || W <= 0 || H > 32000 || W > 32000) return MSDA_E_SHAPE;     // 16-bit chunk boxes / window coordinates
    if (lsi_host[l] != next) return MSDA_E_SHAPE;
    next += H * W;
  }
  return next == S ? 0 : MSDA_E_SHAPE;
}

// Geometry of the LDS-resident gather kernels: stage the longest tail of levels that fits kLdsRows.
// Returns false when nothing would be staged (the plain gather kernels are used instead).
bool make_gather_geom(const int64_t *shapes_host, const int64_t *lsi_host, int B, int M, int Lq, int S,
                      msda::GatherGeom &g, bool bwd = false) {
  int first = 4;
  long long rows = 0;
  for (int l = 3; l >= 0; --l) {
    const long long n = shapes_host[2 * l] * shapes_host[2 * l + 1];
    if (lsi_host[l] + n != (l == 3 ? S : lsi_host[l + 1])) break;      // levels must tile the token axis in order
    if (rows + n > msda::kLdsRows) break;
    rows += n;
    first = l;
  }
  for (int l = 0; l < 4; ++l) {
    g.H[l] = (int)shapes_host[2 * l];
    g.W[l] = (int)shapes_host[2 * l + 1];
    g.start[l] = (int)lsi_host[l];
  }
  g.first_lds_level = first;
  if (first == 4) { g.lds_token0 = 0; g.n_lds_rows = 0; g.n_chunks = 1; return false; }
  g.lds_token0 = (int)lsi_host[first];
  g.n_lds_rows = (int)rows;
  // Query slices per (batch, head).  Workgroups of one (batch, head) run back to back on one XCD (32 CUs):
  // with c slices an XCD has 32 / c value planes (1.2 MB of fine levels each at 1280x384) live in its 4 MB
  // L2 -- measured L2 hit rate 31 % at c = 2.  The forward (one 16-wave workgroup per CU) runs 16 slices = 2 planes live
  // (0.420 -> 0.382 ms against 8 slices, tools/msda_kernel_bench.py; -0.10 ms/step in situ); the backward's gather (two
  // 8-wave workgroups per CU) is best at 8.  Each slice still gets >= 256 queries so the 76 KB of staging stays amortised.
  (void)B;
  const long long chunks = bwd ? gather_chunks_bwd() : gather_chunks_fwd();
  const long long max_chunks = std::max<long long>(1, Lq / 256);
  g.n_chunks = (int)std::max<long long>(1, std::min(chunks, max_chunks));
  return true;
}


// Tiling of the tile-window kernels for a pyramid (searched once per geometry and halo, then cached).
bool window_tiling(const int64_t *shapes_host, const int64_t *lsi_host, int halo, bool bwd, msda::WinGeom &out) {
  static std::mutex mu;
  static std::map<std::vector<int64_t>, std::pair<bool, msda::WinGeom>> cache;
  std::vector<int64_t> key(shapes_host, shapes_host + 8);
  key.insert(key.end(), lsi_host, lsi_host + 4);
  key.push_back(halo);
  key.push_back(bwd);
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(key);
  if (it == cache.end()) {
    msda::WinGeom g{};
    // the backward keeps a block of grad_out rows per wave behind the windows
    int fy = 0, fx = 0;
    if (const char *e = std::getenv("MSDA_WIN_TILES")) std::sscanf(e, "%dx%d", &fy, &fx);      // measurement runs: "NYxNX"
    const bool ok = msda::choose_window_tiling(shapes_host, lsi_host, halo, bwd ? msda::kWinMaxRowsBwd : msda::kWinMaxRows, g, fy, fx,
                                               bwd ? 18000.0 : 400.0);
    it = cache.emplace(key, std::make_pair(ok, g)).first;
  }
  out = it->second.second;
  return it->second.first;
}

// Device copy of a tiling's query table (msda_window.h: WinQuery), made on first use per (device, geometry) and kept for the
// life of the process (0.3 MB for 40 tiles).  The copy is a blocking hipMemcpy: it is complete before any later launch on any
// stream.  First use must therefore happen outside a stream capture (like every library's lazy initialisation).
const msda::WinQuery *query_table(const int64_t *shapes_host, const int64_t *lsi_host, int halo, bool bwd, const msda::WinTable &wt) {
  static std::mutex mu;
  static std::map<std::vector<int64_t>, msda::WinQuery *> cache;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::vector<int64_t> key(shapes_host, shapes_host + 8);
  key.insert(key.end(), lsi_host, lsi_host + 4);
  key.push_back(halo);
  key.push_back(bwd);
  key.push_back(dev);
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(key);
  if (it == cache.end()) {
    std::vector<msda::WinQuery> host((size_t)wt.n_ty * wt.n_tx * msda::kWinMaxQueries * msda::kWinLevels);
    msda::fill_query_table(wt, host.data());
    msda::WinQuery *devp = nullptr;
    if (hipMalloc(&devp, host.size() * sizeof(msda::WinQuery)) != hipSuccess) return nullptr;
    if (hipMemcpy(devp, host.data(), host.size() * sizeof(msda::WinQuery), hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipFree(devp);
      return nullptr;
    }
    it = cache.emplace(key, devp).first;
  }
  return it->second;
}

// One workgroup per CU for the persistent kernels (a multiple of 8: the XCD-aware item order relies on it).
inline int persistent_grid() {
  static const int n = [] {
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
    return std::max(8, cus / 8 * 8);
  }();
  return n;
}

// The window kernels address a (batch, head) plane with 32-bit lane offsets built from 24-bit products: S and the widest row
// stride (floats) are bounded so that a plane's span stays below 2^31 bytes.
constexpr int kWinMaxStride = 4096;
inline bool window_fits(int S, long long widest_stride) { return S < (1 << 17) && widest_stride <= kWinMaxStride; }

inline bool window_applies(bool bwd, const int64_t *shapes_host, const int64_t *lsi_host, int Lq, int S, long long widest_stride) {
  msda::WinGeom wg;
  return Lq == S && (options().window & (bwd ? 2 : 1)) && gather_mode() > 1 && window_fits(S, widest_stride) &&
         window_tiling(shapes_host, lsi_host, options().window_halo, bwd, wg);
}

// A launch helper that could not do what its caller had planned (a device allocation failed) leaves its code here; the entry
// points return it in place of hipGetLastError().
thread_local int tl_launch_error = 0;
inline int launch_status() {
  const int e = tl_launch_error;
  tl_launch_error = 0;
  return e ? e : (int)hipGetLastError();
}

// Launch of the record-based gather kernels (forward or the backward's grad_loc / grad_attn_w pass), staged when
// the tail of the pyramid fits LDS and the option allows it.
// How the caller's `value` is laid out beyond the reference contract (fused entry points, ABI v7): floats between consecutive
// tokens (0 = M * 32) and an optional padding mask [B, S] whose tokens read as zero rows.
struct ValueView {
  int token_stride = 0;
  const unsigned char *mask = nullptr;
};

template <bool BWD, bool FUSED, bool SAVED = false>
void launch_gather(const float *value, const float *loc, const float *attw, const float *grad_out, float *out,
                   float *grad_loc, float *grad_attw, const float *ref, int ref_dim, const int64_t *shapes_host,
                   const int64_t *lsi_host, int B, int S, int M, int Lq, hipStream_t stream, int loc_rs = 0, int aw_rs = 0,
                   float *grad_value = nullptr, int far_reach = -1, ValueView vv = ValueView(),
                   const msda::HeadPlan *plans = nullptr, const unsigned char *far_mask = nullptr) {
  if (!loc_rs) loc_rs = M * 32;
  if (!aw_rs) aw_rs = M * 16;
  const int vts = vv.token_stride ? vv.token_stride : M * 32;
  if (Lq == S && (options().window & (BWD ? 2 : 1)) && gather_mode() > 1 && (!FUSED || ref_dim == 2)) {
    // self-attention shape (the window kernels evaluate the 2-d reference-point formula only): query i sits at token i's
    // pixel -> tile-local value windows (msda_gather_win.hip)
    msda::WinGeom wg;
    // the kernels address a (batch, head) plane with 32-bit lane offsets built from 24-bit products
    const long long widest = std::max<long long>(std::max(loc_rs, aw_rs), std::max(M * 32, vts));
    msda::WinTable wt;
    const msda::WinQuery *qtab = nullptr;
    if (window_fits(S, widest) && window_tiling(shapes_host, lsi_host, options().window_halo, BWD, wg)) {
      msda::fill_window_table(wg, wt);
      qtab = query_table(shapes_host, lsi_host, options().window_halo, BWD, wt);
      if (!qtab) { tl_launch_error = (int)hipErrorOutOfMemory; return; }
    }
    if (qtab) {
      const int bm_groups = (B * M + 7) / 8;
      // persistent: one workgroup per CU walks the (batch * head, tile) items (window_persistent = 0: one workgroup per item)
      const int n_virtual = 8 * wg.n_ty * wg.n_tx * bm_groups;
      const int grid = options().window_persistent ? std::min(n_virtual, persistent_grid()) : n_virtual;
      const int threads = BWD ? msda::kWinThreadsBwd : msda::kWinThreads;
      if (vv.mask)
        msda::gather_win_kernel<BWD, FUSED, SAVED, true><<<grid, threads, 0, stream>>>(
            value, loc, attw, grad_out, out, grad_loc, grad_attw, ref, ref_dim, wt, B, S, M, loc_rs, aw_rs, grad_value, far_reach,
            n_virtual, vts, vv.mask, qtab, plans, far_mask);
      else
        msda::gather_win_kernel<BWD, FUSED, SAVED, false><<<grid, threads, 0, stream>>>(
            value, loc, attw, grad_out, out, grad_loc, grad_attw, ref, ref_dim, wt, B, S, M, loc_rs, aw_rs, grad_value, far_reach,
            n_virtual, vts, nullptr, qtab, plans, far_mask);
      return;
    }
  }
  // callers that planned on the window kernel (saved prologue, far points of the row-tile scatter) cannot fall back
  if (SAVED || far_reach >= 0) { tl_launch_error = MSDA_E_UNSUPPORTED; return; }
  msda::GatherGeom geom;
  const long long n_pairs = (long long)B * Lq * M;
  const bool staged = make_gather_geom(shapes_host, lsi_host, B, M, Lq, S, geom, BWD) && gather_mode() > 1;
  if (staged) {
    const int bm_groups = (B * M + 7) / 8;
    msda::gather_rec_kernel<BWD, true, FUSED>
        <<<8 * geom.n_chunks * bm_groups, BWD ? msda::kStagedThreadsBwd : msda::kStagedThreadsFwd, 0, stream>>>(
            value, loc, attw, grad_out, out, grad_loc, grad_attw, ref, ref_dim, geom, B, S, M, Lq, n_pairs, loc_rs, aw_rs, vts,
            vv.mask);
  } else {
    geom.first_lds_level = 4;
    msda::gather_rec_kernel<BWD, false, FUSED><<<grid_pairs(n_pairs), msda::kPlainThreads, 0, stream>>>(
        value, loc, attw, grad_out, out, grad_loc, grad_attw, ref, ref_dim, geom, B, S, M, Lq, n_pairs, loc_rs, aw_rs, vts, vv.mask);
  }
}

template <typename T>
int forward_impl(const T *value, const int64_t *shapes, const int64_t *lsi, const T *loc,
                 const T *attw, T *out, int B, int S, int M, int D, int L, int Lq, int P,
                 const int64_t *shapes_host, const int64_t *lsi_host, void *stream_) {
  if (!value || !shapes || !lsi || !loc || !attw || !out) return MSDA_E_NULLPTR;
  if (int e = check_dims(B, S, M, D, L, Lq, P)) return e;
  hipStream_t stream = (hipStream_t)stream_;
  const long long n_pairs = (long long)B * Lq * M;
  if constexpr (sizeof(T) == 4) {
    if (D == 32 && L == 4 && P == 4) {
      if (shapes_host && lsi_host && gather_mode() > 0) {
        if (int e = check_host_geometry(shapes_host, lsi_host, L, S)) return e;
        launch_gather<false, false>(value, loc, attw, nullptr, out, nullptr, nullptr, nullptr, 0, shapes_host, lsi_host,
                                    B, S, M, Lq, stream);
        return launch_status();
      }
      msda::fwd_d32_kernel<4, 4><<<grid_pairs(n_pairs), 256, 0, stream>>>(
          value, shapes, lsi, loc, attw, out, S, M, Lq, n_pairs);
      return launch_status();
    }
  }
  msda::fwd_generic_kernel<T><<<grid_for(n_pairs, 8), 256, 0, stream>>>(
      value, shapes, lsi, loc, attw, out, S, M, D, L, Lq, P, n_pairs);
  return launch_status();
}

inline bool tiled_backward_applies(int elem_bytes, int D, int L, int P) {
  return elem_bytes == 4 && D == 32 && L == 4 && P == 4 && L <= msda::kMaxLevels;
}
// K1 runs M * L * 8 threads per workgroup

inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }

// workspace layout: [rec_hw: 8 B/point][rec_aw: 4 B/point][chunk boxes: 8 B per 64 points][bounds: 2 floats per (b, m)]
struct TiledWorkspace {
  size_t n_points, n_chunks_per_list, n_lists, off_aw, off_boxes, off_bounds, bytes;
};
inline TiledWorkspace tiled_workspace(int B, int M, int L, int Lq, int P) {
  TiledWorkspace w;
  w.n_lists = (size_t)B * M * L;
  w.n_points = w.n_lists * Lq * P;
  w.n_chunks_per_list = ((size_t)Lq * P + 63) / 64;
  w.off_aw = align256(w.n_points * 8);
  w.off_boxes = align256(w.off_aw + w.n_points * 4);
  w.off_bounds = align256(w.off_boxes + w.n_lists * w.n_chunks_per_list * sizeof(msda::ChunkBox));
  w.bytes = w.off_bounds + align256((size_t)B * M * 2 * sizeof(float));
  return w;
}
inline size_t tiled_workspace_bytes(int B, int M, int L, int Lq, int P) { return tiled_workspace(B, M, L, Lq, P).bytes; }

// Tiling of every level for the tile-owner scatter (msda_backward_tiled.hip, K2).
msda::BwdPlan make_plan(const int64_t *shapes_host, const int64_t *lsi_host, int L, int Lq, int P) {
  msda::BwdPlan plan{};
  plan.n_levels = L;
  const long long n_pts = (long long)Lq * P;
  const long long target = 8192;                 // sampling points scanned per workgroup, roughly
  double work[msda::kMaxLevels];
  for (int l = 0; l < L; ++l) {
    const int H = (int)shapes_host[2 * l], W = (int)shapes_host[2 * l + 1];
    plan.H[l] = H;
    plan.W[l] = W;
    plan.start[l] = (int)lsi_host[l];
    if ((long long)H * W <= msda::kTileRows) {
      plan.th[l] = H; plan.tw[l] = W; plan.n_ty[l] = 1; plan.n_tx[l] = 1;
    } else {
      // balanced tiles of at most 16 x 16 (or kTileRows / W rows when the level is narrow)
      const int max_tw = std::min(W, 16), max_th = std::max(1, std::min(H, msda::kTileRows / max_tw));
      plan.n_tx[l] = (W + max_tw - 1) / max_tw;
      plan.tw[l] = (W + plan.n_tx[l] - 1) / plan.n_tx[l];
      plan.n_ty[l] = (H + max_th - 1) / max_th;
      plan.th[l] = (H + plan.n_ty[l] - 1) / plan.n_ty[l];
    }
    const long long tiles = (long long)plan.n_ty[l] * plan.n_tx[l];
    const long long per_tile = (n_pts + tiles - 1) / tiles;
    plan.n_chunks[l] = (int)std::max(1LL, std::min(8LL, (per_tile + target / 2) / target));
    work[l] = (double)per_tile / plan.n_chunks[l];
    plan.order[l] = l;
  }
  std::sort(plan.order, plan.order + L, [&](int a, int b) { return work[a] > work[b]; });
  plan.first_item[0] = 0;
  for (int i = 0; i < L; ++i) {
    const int l = plan.order[i];
    plan.first_item[i + 1] = plan.first_item[i] + plan.n_ty[l] * plan.n_tx[l] * plan.n_chunks[l];
  }
  plan.n_items = plan.first_item[L];
  return plan;
}

template <typename T>
int backward_impl(const T *value, const int64_t *shapes, const int64_t *lsi, const T *loc,
                  const T *attw, const T *grad_out, T *grad_value, T *grad_loc, T *grad_attw,
                  int B, int S, int M, int D, int L, int Lq, int P, const int64_t *shapes_host,
                  const int64_t *lsi_host, void *workspace, size_t workspace_bytes, void *stream_,
                  const float *fused_ref = nullptr, int fused_ref_dim = 0, int loc_rs = 0, int aw_rs = 0, bool saved = false,
                  ValueView vv = ValueView(), int plan_mode = 0) {
  // plan_mode (saved self-attention backward, ABI v9): 0 = plan and run; 1 = PLAN ONLY -- the directional statistics, the per-head
  // plan and the candidate tables go into `workspace` and nothing else happens (they depend on the forward's saved locations
  // only, so the caller runs them early, on a side stream); 2 = run from the plan a mode-1 call left in `workspace`
  // saved: `loc` / `attw` are the sampling locations / attention weights the fused forward stored (contiguous); the
  // gradients still go back to raw offsets / logits through `fused_ref` and the row strides (self-attention shape only)
  if (plan_mode == 1 ? (!loc || !shapes || !lsi) :
      (!value || !shapes || !lsi || !loc || !attw || !grad_out || !grad_value || !grad_loc || !grad_attw))
    return MSDA_E_NULLPTR;
  if (int e = check_dims(B, S, M, D, L, Lq, P)) return e;
  hipStream_t stream = (hipStream_t)stream_;
  const long long n_pairs = (long long)B * Lq * M;
  if ((vv.token_stride || vv.mask) && !(sizeof(T) == 4 && fused_ref && tiled_backward_applies(4, D, L, P) && M * L * 8 <= 1024))
    return MSDA_E_UNSUPPORTED;          // strided / masked value: fused d32 path only

  if constexpr (sizeof(T) == 4) {
    if (tiled_backward_applies(4, D, L, P) && M * L * 8 <= 1024) {
      const size_t need = tiled_workspace_bytes(B, M, L, Lq, P);
      if (!workspace || workspace_bytes < need) return MSDA_E_WORKSPACE;
      int64_t host_geom[3 * msda::kMaxLevels];
      if (!shapes_host || !lsi_host) {          // blocking fetch of the pyramid (see header)
        hipError_t e = hipMemcpyAsync(host_geom, shapes, sizeof(int64_t) * 2 * L, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess)
          e = hipMemcpyAsync(host_geom + 2 * L, lsi, sizeof(int64_t) * L, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return (int)e;
        shapes_host = host_geom;
        lsi_host = host_geom + 2 * L;
      }
      if (int e = check_host_geometry(shapes_host, lsi_host, L, S)) return e;
      // self-attention shape: row-tile scatter + window gather (no transposed lists, no LDS atomics per channel)
      const long long widest = std::max<long long>(std::max(loc_rs, aw_rs), std::max(M * 32, vv.token_stride));
      if (options().scatter_rows && window_applies(true, shapes_host, lsi_host, Lq, S, widest) && (!fused_ref || fused_ref_dim == 2)) {
        msda::RowPlan rp;
        // ---- saved backward: exact scan lists (msda_bin.hip) -- one streaming pass over the saved locations bins every (query,
        // level) unit into the tiles its points fall into; the cell scatter scans exactly those, whatever the offsets look like
        if (plan_mode && options().scatter_lists) return MSDA_E_UNSUPPORTED;      // the exact lists have no plan to run ahead
        if (saved && options().scatter_lists && msda::make_row_plan(shapes_host, lsi_host, options().scatter_reach, rp)) {
          msda::BinPlan bp{};
          int n_tiles = 0, entries = 0;
          double work[4];
          for (int l = 0; l < 4; ++l) {
            bp.H[l] = rp.H[l]; bp.W[l] = rp.W[l];
            bp.th[l] = rp.th[l]; bp.tw[l] = rp.tw[l]; bp.n_ty[l] = rp.n_ty[l]; bp.n_tx[l] = rp.n_tx[l];
            bp.tile0[l] = n_tiles;
            const int tiles = rp.n_ty[l] * rp.n_tx[l];
            // capacity: twice the share of a uniform spread (1.5 lists per unit: cells on an apron belong to two tiles), + slack
            // (never more than the 16 chunks of kRowChunkQueries the scatter's grid scans: the excess takes the far path; `scatter_lists_cap`
            // forces a small capacity so that tests reach that path)
            bp.cap[l] = (int)std::min<long long>((3LL * Lq + tiles - 1) / tiles + 256, 16LL * msda::kRowChunkQueries);
            if (options().scatter_lists_cap > 0) bp.cap[l] = std::min(bp.cap[l], options().scatter_lists_cap);
            bp.list_off[l] = entries;
            n_tiles += tiles;
            entries += tiles * bp.cap[l];
            rp.n_chunks[l] = std::max(1, std::min(16, (bp.cap[l] + msda::kRowChunkQueries - 1) / msda::kRowChunkQueries));
            work[l] = (double)bp.cap[l] / rp.n_chunks[l];
            rp.order[l] = l;
          }
          bp.n_tiles = n_tiles; bp.plane_entries = entries;
          std::sort(rp.order, rp.order + 4, [&](int a, int b) { return work[a] > work[b]; });
          rp.first_item[0] = 0;
          for (int i = 0; i < 4; ++i) rp.first_item[i + 1] = rp.first_item[i] + rp.n_ty[rp.order[i]] * rp.n_tx[rp.order[i]] * rp.n_chunks[rp.order[i]];
          rp.n_items = rp.first_item[4];
          const size_t counts_b = align256(sizeof(unsigned) * (size_t)B * M * n_tiles);
          const size_t lists_b = align256(sizeof(unsigned) * (size_t)B * M * entries);
          const size_t far_b = align256((size_t)B * M * 4 * S);
          if ((long long)B * M * entries < (1LL << 31) && counts_b + lists_b + far_b <= workspace_bytes) {
            char *wsp = reinterpret_cast<char *>(workspace);
            unsigned *counts = reinterpret_cast<unsigned *>(wsp);
            unsigned *lists = reinterpret_cast<unsigned *>(wsp + counts_b);
            unsigned char *far_mask = reinterpret_cast<unsigned char *>(wsp + counts_b + lists_b);
            hipError_t e = hipMemsetAsync(counts, 0, counts_b, stream);
            if (e != hipSuccess) return (int)e;
            const long long n_units = (long long)B * M * 4 * S;
            msda::bin_points_kernel<<<(unsigned)((n_units + 255) / 256), 256, 0, stream>>>(loc, bp, n_units, S, counts, lists, far_mask);
            for (int l = 0; l < L; ++l) {          // levels shared by several workgroups are accumulated with atomics
              if (rp.n_chunks[l] == 1) continue;
              e = hipMemset2DAsync(grad_value + (size_t)rp.start[l] * M * 32, sizeof(float) * (size_t)S * M * 32, 0,
                                   sizeof(float) * (size_t)rp.H[l] * rp.W[l] * M * 32, B, stream);
              if (e != hipSuccess) return (int)e;
            }
            const int groups = (B * M + 7) / 8;
            msda::scatter_rows_kernel<false, true><<<8 * rp.n_items * groups, msda::kRowThreads, 0, stream>>>(
                loc, attw, grad_out, grad_value, nullptr, 0, nullptr, rp, B, S, M, 0, 0, vv.mask, nullptr, lists, counts, bp, far_mask);
            launch_gather<true, true, true>(value, loc, attw, grad_out, nullptr, grad_loc, grad_attw, fused_ref, fused_ref_dim,
                                            shapes_host, lsi_host, B, S, M, Lq, stream, loc_rs, aw_rs, grad_value, 0, vv, nullptr, far_mask);
            return launch_status();
          }
        }
        // directional plan (msda_plan.h): fused operator with 2-d reference points, or the forward's saved locations
        bool directional = options().directional && (saved || fused_ref) && M <= msda::kPlanMaxHeads;
        int reach = 0;
        size_t plan_b = 0, stats_b = 0;
        bool ok = false;
        for (int attempt = 0; attempt < 2 && !ok; ++attempt) {        // a workspace too small for the per-head tables: isotropic plan
          reach = directional ? std::max(options().scatter_reach, options().plan_reach) : options().scatter_reach;
          plan_b = directional ? align256(msda::plan_bytes(M)) : 0;
          stats_b = directional ? align256(msda::plan_stats_bytes(M) * msda::kStatsBlocks) : 0;
          ok = msda::make_row_plan(shapes_host, lsi_host, reach, rp) && (!directional || rp.n_items <= msda::kPlanMaxItems) &&
               plan_b + stats_b + msda::row_plan_table_bytes(rp) * (directional ? M : 1) <= workspace_bytes;
          if (!ok) { if (!directional) break; directional = false; }
        }
        if (ok) {
          char *wsp = reinterpret_cast<char *>(workspace);
          msda::HeadPlan *plans = directional ? reinterpret_cast<msda::HeadPlan *>(wsp) : nullptr;
          msda::RowCandidate *table = reinterpret_cast<msda::RowCandidate *>(wsp + plan_b + stats_b);
          int n_tiles = 0;
          for (int l = 0; l < 4; ++l) n_tiles += rp.n_ty[l] * rp.n_tx[l];
          if (plan_mode && !(directional && saved)) return MSDA_E_UNSUPPORTED;
          if (directional && plan_mode != 2) {
            msda::PlanGeom pg{};
            for (int l = 0; l < 4; ++l) { pg.H[l] = rp.H[l]; pg.W[l] = rp.W[l]; pg.start[l] = rp.start[l]; }
            pg.S = S; pg.M = M;
            pg.default_halo = options().window_halo;
            pg.reach = reach;
            pg.want_rows = 1;
            if (options().plan_fused) {
              // statistics, per-head plan and candidate tables in ONE launch (msda_plan.hip: plan_fused_kernel)
              if (saved)
                msda::plan_fused_kernel<1><<<M * n_tiles, 256, 0, stream>>>(loc, nullptr, pg, rp, B, 0, n_tiles, plans, table);
              else
                msda::plan_fused_kernel<0><<<M * n_tiles, 256, 0, stream>>>(loc, fused_ref, pg, rp, B, loc_rs ? loc_rs : M * 32, n_tiles,
                                                                            plans, table);
            } else {
              msda::DirStats *partial = reinterpret_cast<msda::DirStats *>(wsp + plan_b);
              if (saved)
                msda::dir_stats_kernel<1><<<msda::kStatsBlocks, msda::kStatsThreads, 0, stream>>>(loc, nullptr, pg, B, 0, partial);
              else
                msda::dir_stats_kernel<0><<<msda::kStatsBlocks, msda::kStatsThreads, 0, stream>>>(
                    loc, fused_ref, pg, B, loc_rs ? loc_rs : M * 32, partial);
              msda::dir_plan_kernel<<<M, 256, 0, stream>>>(partial, msda::kStatsBlocks, pg, rp, plans);
            }
          }
          if (plan_mode != 2 && !(directional && options().plan_fused))
            msda::row_candidates_kernel<<<n_tiles * (directional ? M : 1), 256, 0, stream>>>(rp, table, plans, n_tiles);
          if (plan_mode == 1) return launch_status();
          for (int l = 0; l < L; ++l) {          // levels shared by several workgroups are accumulated with atomics
            if (rp.n_chunks[l] == 1) continue;
            hipError_t e = hipMemset2DAsync(grad_value + (size_t)rp.start[l] * M * 32, sizeof(float) * (size_t)S * M * 32, 0,
                                            sizeof(float) * (size_t)rp.H[l] * rp.W[l] * M * 32, B, stream);
            if (e != hipSuccess) return (int)e;
          }
          const int groups = (B * M + 7) / 8;
          const int lrs = loc_rs ? loc_rs : M * 32, ars = aw_rs ? aw_rs : M * 16;
          if (fused_ref && !saved)
            msda::scatter_rows_kernel<true><<<8 * rp.n_items * groups, msda::kRowThreads, 0, stream>>>(
                loc, attw, grad_out, grad_value, fused_ref, fused_ref_dim, table, rp, B, S, M, lrs, ars, vv.mask, plans);
          else if (saved)
            msda::scatter_rows_kernel<false, true><<<8 * rp.n_items * groups, msda::kRowThreads, 0, stream>>>(
                loc, attw, grad_out, grad_value, nullptr, 0, table, rp, B, S, M, 0, 0, vv.mask, plans);
          else
            msda::scatter_rows_kernel<false><<<8 * rp.n_items * groups, msda::kRowThreads, 0, stream>>>(
                loc, attw, grad_out, grad_value, nullptr, 0, table, rp, B, S, M, lrs, ars);
          if (fused_ref && saved)
            launch_gather<true, true, true>(value, loc, attw, grad_out, nullptr, grad_loc, grad_attw, fused_ref, fused_ref_dim,
                                            shapes_host, lsi_host, B, S, M, Lq, stream, loc_rs, aw_rs, grad_value, rp.reach, vv, plans);
          else if (fused_ref)
            launch_gather<true, true>(value, loc, attw, grad_out, nullptr, grad_loc, grad_attw, fused_ref, fused_ref_dim,
                                      shapes_host, lsi_host, B, S, M, Lq, stream, loc_rs, aw_rs, grad_value, rp.reach, vv, plans);
          else
            launch_gather<true, false>(value, loc, attw, grad_out, nullptr, grad_loc, grad_attw, nullptr, 0, shapes_host,
                                       lsi_host, B, S, M, Lq, stream, 0, 0, grad_value, rp.reach);
          return launch_status();
        }
      }
      if (saved || plan_mode) return MSDA_E_UNSUPPORTED;          // msda_fused_save_supported() said otherwise
      const msda::BwdPlan plan = make_plan(shapes_host, lsi_host, L, Lq, P);
      if ((long long)Lq * P >= (1LL << (62 - msda::kFixBits))) return MSDA_E_SHAPE;   // fixed-point headroom
      const TiledWorkspace ws = tiled_workspace(B, M, L, Lq, P);
      char *wsp = reinterpret_cast<char *>(workspace);
      float2 *rec_hw = reinterpret_cast<float2 *>(wsp);
      float *rec_aw = reinterpret_cast<float *>(wsp + ws.off_aw);
      msda::ChunkBox *boxes = reinterpret_cast<msda::ChunkBox *>(wsp + ws.off_boxes);
      float *bounds = reinterpret_cast<float *>(wsp + ws.off_bounds);
      if (fused_ref)
        msda::bwd_prep_kernel<4, true><<<(unsigned)(B * ws.n_chunks_per_list), M * L * 8, 0, stream>>>(
            loc, attw, grad_out, shapes, rec_hw, rec_aw, boxes, fused_ref, fused_ref_dim, M, L, Lq, (int)ws.n_chunks_per_list,
            loc_rs ? loc_rs : M * 32, aw_rs ? aw_rs : M * 16);
      else
        msda::bwd_prep_kernel<4, false><<<(unsigned)(B * ws.n_chunks_per_list), M * L * 8, 0, stream>>>(
            loc, attw, grad_out, shapes, rec_hw, rec_aw, boxes, nullptr, 0, M, L, Lq, (int)ws.n_chunks_per_list, M * 32, M * 16);
      // short record lists (the decoder's cross-attention): the row-band scatter (msda_backward_bands.hip) -- every level written
      // with plain stores, f32 sums in registers: no fixed-point bounds, no zero fill
      msda::BandPlan bp;
      const bool bands = options().scatter_bands && options().scatter_sorted != 2 && L == 4 && P == 4 &&
                         msda::make_band_plan(shapes_host, lsi_host, Lq, bp);
      if (!bands) msda::bwd_bounds_kernel<<<B * M, 256, 0, stream>>>(boxes, bounds, (int)(L * ws.n_chunks_per_list));
      // levels shared by several workgroups are accumulated with atomics: zero exactly those rows
      for (int l = 0; l < L && !bands; ++l) {
        if (plan.n_chunks[l] == 1) continue;
        hipError_t e = hipMemset2DAsync(grad_value + (size_t)plan.start[l] * M * 32, sizeof(float) * (size_t)S * M * 32,
                                        0, sizeof(float) * (size_t)plan.H[l] * plan.W[l] * M * 32, B, stream);
        if (e != hipSuccess) return (int)e;
      }
      const int bm_groups = (B * M + 7) / 8;
      if (bands) {
        msda::bwd_scatter_bands_kernel<<<8 * bp.n_items * bm_groups, msda::kBandThreads, 0, stream>>>(
            rec_hw, rec_aw, grad_out, grad_value, bp, B, S, M, Lq, vv.mask);
      } else
      // measured (B = 16, KITTI pyramid): sorted 0.90 ms vs 1.04 ms at Lq = 10200; 0.259 vs 0.245 at Lq = 550; 0.137 vs
      // 0.091 at Lq = 50 -- the batches' barriers only pay off on long record lists
      if (options().scatter_sorted == 1 ? Lq * P >= 8192 : options().scatter_sorted == 2)
        msda::bwd_scatter_sorted_kernel<<<8 * plan.n_items * bm_groups, msda::kSortThreads, 0, stream>>>(
            rec_hw, rec_aw, boxes, grad_out, grad_value, plan, B, S, M, Lq, P, (int)ws.n_chunks_per_list, vv.mask);
      else if (scatter_fixed_point())
        msda::bwd_scatter_kernel<true><<<8 * plan.n_items * bm_groups, msda::kScatterThreads, 0, stream>>>(
            rec_hw, rec_aw, boxes, bounds, grad_out, grad_value, plan, B, S, M, Lq, P, (int)ws.n_chunks_per_list, vv.mask);
      else
        msda::bwd_scatter_kernel<false><<<8 * plan.n_items * bm_groups, msda::kScatterThreads, 0, stream>>>(
            rec_hw, rec_aw, boxes, bounds, grad_out, grad_value, plan, B, S, M, Lq, P, (int)ws.n_chunks_per_list, vv.mask);
      if (gather_mode() > 0 || fused_ref) {
        if (fused_ref)
          launch_gather<true, true>(value, loc, attw, grad_out, nullptr, grad_loc, grad_attw, fused_ref, fused_ref_dim,
                                    shapes_host, lsi_host, B, S, M, Lq, stream, loc_rs, aw_rs, nullptr, -1, vv);
        else
          launch_gather<true, false>(value, loc, attw, grad_out, nullptr, grad_loc, grad_attw, nullptr, 0, shapes_host,
                                     lsi_host, B, S, M, Lq, stream);
      } else {
        msda::bwd_gather_kernel<4, 4><<<grid_pairs(n_pairs), 256, 0, stream>>>(
            value, shapes, lsi, loc, attw, grad_out, grad_loc, grad_attw, S, M, Lq, n_pairs);
      }
      return launch_status();
    }
  }
  hipError_t err = hipMemsetAsync(grad_value, 0, sizeof(T) * (size_t)B * S * M * D, stream);
  if (err != hipSuccess) return (int)err;
  msda::bwd_generic_kernel<T><<<grid_for(n_pairs, 8), 256, 0, stream>>>(
      value, shapes, lsi, loc, attw, grad_out, grad_value, grad_loc, grad_attw, S, M, D, L, Lq, P,
      n_pairs);
  return launch_status();
}

}  // namespace

extern "C" {

int msda_abi_version(void) { return MSDA_ABI_VERSION; }

int msda_set_option(const char *name, int value) {
  if (!name) return MSDA_E_NULLPTR;
  const std::string n(name);
  if (n == "gather" && value >= 0 && value <= 2) { options().gather = value; return 0; }
  if (n == "scatter_fixed" && (value == 0 || value == 1)) { options().scatter_fixed = value; return 0; }
  if (n == "scatter_sorted" && value >= 0 && value <= 2) { options().scatter_sorted = value; return 0; }
  if (n == "window" && value >= 0 && value <= 3) { options().window = value; return 0; }
  if (n == "window_halo" && value >= 0 && value <= 32) { options().window_halo = value; return 0; }
  if (n == "window_persistent" && (value == 0 || value == 1)) { options().window_persistent = value; return 0; }
  if (n == "scatter_rows" && (value == 0 || value == 1)) { options().scatter_rows = value; return 0; }
  if (n == "scatter_reach" && value >= 1 && value <= 16) { options().scatter_reach = value; return 0; }
  if (n == "directional" && (value == 0 || value == 1)) { options().directional = value; return 0; }
  if (n == "scatter_bands" && (value == 0 || value == 1)) { options().scatter_bands = value; return 0; }
  if (n == "scatter_lists" && (value == 0 || value == 1)) { options().scatter_lists = value; return 0; }
  if (n == "plan_reach" && value >= 1 && value <= 16) { options().plan_reach = value; return 0; }
  if (n == "plan_fused" && (value == 0 || value == 1)) { options().plan_fused = value; return 0; }
  if (n == "scatter_lists_cap" && value >= 0) { options().scatter_lists_cap = value; return 0; }
  return MSDA_E_UNSUPPORTED;
}

int msda_options_stamp(void) {
  // everything a plan made by msda_saved_plan_f32 depends on besides the call's own arguments
  const Options &o = options();
  unsigned h = 2166136261u;
  for (int v : {o.directional, o.scatter_lists, o.scatter_rows, o.scatter_reach, o.plan_reach, o.plan_fused, o.window, o.window_halo, o.gather}) {
    h ^= (unsigned)v + 0x9E3779B9u;
    h *= 16777619u;
  }
  return (int)(h & 0x7FFFFFFFu);
}

int msda_debug_counter(const char *name, unsigned long long *out) {
  if (!name || !out) return MSDA_E_NULLPTR;
  const std::string n(name);
  static const char *const scan_names[5] = {"scan_candidates", "scan_delivering", "scan_point_tests", "scan_delivered", "scan_points_skipped"};
  int scan = -1;
  for (int i = 0; i < 5; ++i)
    if (n == scan_names[i]) scan = i;
  if (n != "scatter_overflow_rounds" && scan < 0) return MSDA_E_UNSUPPORTED;
  const unsigned long long zero = 0;
  // (blocking copies on the null stream: every launch issued before this call has finished when the value is read)
  hipError_t e = hipDeviceSynchronize();
  if (scan >= 0) {
    // the cell scatter's scan census: counted by measurement builds only (-DMSDA_ROWS_COUNT=1, msda_scatter_rows.hip); 0 otherwise
    if (e == hipSuccess) e = hipMemcpyFromSymbol(out, HIP_SYMBOL(msda::g_rows_scan), sizeof(zero), scan * sizeof(zero));
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(msda::g_rows_scan), &zero, sizeof(zero), scan * sizeof(zero));
    return (int)e;
  }
  if (e == hipSuccess) e = hipMemcpyFromSymbol(out, HIP_SYMBOL(msda::g_rows_overflow_rounds), sizeof(zero));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(msda::g_rows_overflow_rounds), &zero, sizeof(zero));
  return (int)e;
}

const char *msda_strerror(int code) {
  switch (code) {
    case 0: return "success";
    case MSDA_E_NULLPTR: return "msda: a required pointer is NULL";
    case MSDA_E_SHAPE: return "msda: a dimension is <= 0 or exceeds the indexing range";
    case MSDA_E_UNSUPPORTED: return "msda: unsupported configuration";
    case MSDA_E_WORKSPACE: return "msda: workspace is NULL or smaller than msda_backward_workspace_bytes()";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "msda: unknown error";
  }
}

size_t msda_backward_workspace_bytes(int B, int S, int M, int D, int L, int Lq, int P, int elem_bytes) {
  (void)S;
  if (B <= 0 || M <= 0 || L <= 0 || Lq <= 0 || P <= 0) return 0;
  return (tiled_backward_applies(elem_bytes, D, L, P) && M * L * 8 <= 1024) ? tiled_workspace_bytes(B, M, L, Lq, P) : 0;
}

int msda_forward_f32(const float *value, const int64_t *shapes, const int64_t *level_start,
                     const float *loc, const float *attn_w, float *out, int B, int S, int M, int D,
                     int L, int Lq, int P, const int64_t *shapes_host, const int64_t *level_start_host,
                     void *stream) {
  return forward_impl<float>(value, shapes, level_start, loc, attn_w, out, B, S, M, D, L, Lq, P, shapes_host,
                             level_start_host, stream);
}

// ---- ABI v7: the fused operator on a value VIEW (token stride + padding mask) -------------------------------------------------
int msda_fused_forward_view_f32(const float *value, int value_token_stride, const unsigned char *value_mask,
                                const int64_t *shapes, const int64_t *level_start, const float *offsets, const float *logits,
                                const float *ref, int ref_dim, float *out, float *loc_save, float *attn_save, int B, int S, int M,
                                int D, int L, int Lq, int P, int offsets_row_stride, int logits_row_stride,
                                const int64_t *shapes_host, const int64_t *level_start_host, void *stream) {
  if (!value || !shapes || !level_start || !offsets || !logits || !ref || !out || !shapes_host || !level_start_host)
    return MSDA_E_NULLPTR;
  if ((loc_save == nullptr) != (attn_save == nullptr)) return MSDA_E_NULLPTR;
  if (int e = check_dims(B, S, M, D, L, Lq, P)) return e;
  if (!(D == 32 && L == 4 && P == 4) || (ref_dim != 2 && ref_dim != 6)) return MSDA_E_UNSUPPORTED;
  if (offsets_row_stride < M * 32 || logits_row_stride < M * 16 || (offsets_row_stride & 3) || (logits_row_stride & 3))
    return MSDA_E_SHAPE;
  if (value_token_stride < M * 32 || (value_token_stride & 3) || (long long)S * value_token_stride >= (1LL << 31)) return MSDA_E_SHAPE;
  if (int e = check_host_geometry(shapes_host, level_start_host, L, S)) return e;
  ValueView vv;
  vv.token_stride = value_token_stride;
  vv.mask = value_mask;
  if (loc_save) {
    if (!msda_fused_save_supported_view(S, M, D, L, Lq, P, ref_dim, value_token_stride, offsets_row_stride, logits_row_stride,
                                        shapes_host, level_start_host))
      return MSDA_E_UNSUPPORTED;
    launch_gather<false, true, true>(value, offsets, logits, nullptr, out, loc_save, attn_save, ref, ref_dim, shapes_host,
                                     level_start_host, B, S, M, Lq, (hipStream_t)stream, offsets_row_stride, logits_row_stride,
                                     nullptr, -1, vv);
  } else {
    launch_gather<false, true>(value, offsets, logits, nullptr, out, nullptr, nullptr, ref, ref_dim, shapes_host,
                               level_start_host, B, S, M, Lq, (hipStream_t)stream, offsets_row_stride, logits_row_stride, nullptr,
                               -1, vv);
  }
  return launch_status();
}

int msda_fused_backward_view_f32(const float *value, int value_token_stride, const unsigned char *value_mask,
                                 const int64_t *shapes, const int64_t *level_start, const float *offsets_or_loc,
                                 const float *logits_or_attn, int saved, const float *ref, int ref_dim, const float *grad_out,
                                 float *grad_value, float *grad_offsets, float *grad_logits, int B, int S, int M, int D, int L,
                                 int Lq, int P, int offsets_row_stride, int logits_row_stride, const int64_t *shapes_host,
                                 const int64_t *level_start_host, void *workspace, size_t workspace_bytes, void *stream) {
  if (!ref || !shapes_host || !level_start_host) return MSDA_E_NULLPTR;
  if (!(D == 32 && L == 4 && P == 4) || (ref_dim != 2 && ref_dim != 6)) return MSDA_E_UNSUPPORTED;
  if (offsets_row_stride < M * 32 || logits_row_stride < M * 16 || (offsets_row_stride & 3) || (logits_row_stride & 3))
    return MSDA_E_SHAPE;
  if (value_token_stride < M * 32 || (value_token_stride & 3) || (long long)S * value_token_stride >= (1LL << 31)) return MSDA_E_SHAPE;
  if (saved && !msda_fused_save_supported_view(S, M, D, L, Lq, P, ref_dim, value_token_stride, offsets_row_stride, logits_row_stride,
                                               shapes_host, level_start_host))
    return MSDA_E_UNSUPPORTED;
  ValueView vv;
  vv.token_stride = value_token_stride;
  vv.mask = value_mask;
  return backward_impl<float>(value, shapes, level_start, offsets_or_loc, logits_or_attn, grad_out, grad_value, grad_offsets,
                              grad_logits, B, S, M, D, L, Lq, P, shapes_host, level_start_host, workspace, workspace_bytes,
                              stream, ref, ref_dim, offsets_row_stride, logits_row_stride, saved != 0, vv);
}

// ---- ABI v9: the saved backward's plan ahead of the backward ----------------------------------------------------------------------
int msda_saved_plan_f32(const float *loc_saved, const int64_t *shapes, const int64_t *level_start, int value_token_stride, int B, int S,
                        int M, int D, int L, int Lq, int P, int offsets_row_stride, int logits_row_stride, const int64_t *shapes_host,
                        const int64_t *level_start_host, void *workspace, size_t workspace_bytes, void *stream) {
  if (!loc_saved || !shapes_host || !level_start_host || !workspace) return MSDA_E_NULLPTR;
  if (!msda_fused_save_supported_view(S, M, D, L, Lq, P, 2, value_token_stride, offsets_row_stride, logits_row_stride, shapes_host,
                                      level_start_host))
    return MSDA_E_UNSUPPORTED;
  ValueView vv;
  vv.token_stride = value_token_stride;
  static const float dummy_ref = 0.f;            // (a 2-d reference is what the saved path requires; never read in plan mode)
  return backward_impl<float>(nullptr, shapes, level_start, loc_saved, nullptr, nullptr, nullptr, nullptr, nullptr, B, S, M, D, L, Lq, P,
                              shapes_host, level_start_host, workspace, workspace_bytes, stream, &dummy_ref, 2, offsets_row_stride,
                              logits_row_stride, true, vv, 1);
}

int msda_fused_backward_view_planned_f32(const float *value, int value_token_stride, const unsigned char *value_mask,
                                         const int64_t *shapes, const int64_t *level_start, const float *loc_saved,
                                         const float *attn_saved, const float *ref, int ref_dim, const float *grad_out,
                                         float *grad_value, float *grad_offsets, float *grad_logits, int B, int S, int M, int D, int L,
                                         int Lq, int P, int offsets_row_stride, int logits_row_stride, const int64_t *shapes_host,
                                         const int64_t *level_start_host, void *planned_workspace, size_t workspace_bytes, void *stream) {
  if (!ref || !shapes_host || !level_start_host || !planned_workspace) return MSDA_E_NULLPTR;
  if (!(D == 32 && L == 4 && P == 4) || ref_dim != 2) return MSDA_E_UNSUPPORTED;
  if (offsets_row_stride < M * 32 || logits_row_stride < M * 16 || (offsets_row_stride & 3) || (logits_row_stride & 3))
    return MSDA_E_SHAPE;
  if (value_token_stride < M * 32 || (value_token_stride & 3) || (long long)S * value_token_stride >= (1LL << 31)) return MSDA_E_SHAPE;
  if (!msda_fused_save_supported_view(S, M, D, L, Lq, P, ref_dim, value_token_stride, offsets_row_stride, logits_row_stride,
                                      shapes_host, level_start_host))
    return MSDA_E_UNSUPPORTED;
  ValueView vv;
  vv.token_stride = value_token_stride;
  vv.mask = value_mask;
  return backward_impl<float>(value, shapes, level_start, loc_saved, attn_saved, grad_out, grad_value, grad_offsets, grad_logits, B, S, M,
                              D, L, Lq, P, shapes_host, level_start_host, planned_workspace, workspace_bytes, stream, ref, ref_dim,
                              offsets_row_stride, logits_row_stride, true, vv, 2);
}

int msda_forward_f64(const double *value, const int64_t *shapes, const int64_t *level_start,
                     const double *loc, const double *attn_w, double *out, int B, int S, int M, int D,
                     int L, int Lq, int P, const int64_t *shapes_host, const int64_t *level_start_host,
                     void *stream) {
  return forward_impl<double>(value, shapes, level_start, loc, attn_w, out, B, S, M, D, L, Lq, P, shapes_host,
                              level_start_host, stream);
}

int msda_backward_f32(const float *value, const int64_t *shapes, const int64_t *level_start,
                      const float *loc, const float *attn_w, const float *grad_out, float *grad_value,
                      float *grad_loc, float *grad_attn_w, int B, int S, int M, int D, int L, int Lq,
                      int P, const int64_t *shapes_host, const int64_t *level_start_host, void *workspace,
                      size_t workspace_bytes, void *stream) {
  return backward_impl<float>(value, shapes, level_start, loc, attn_w, grad_out, grad_value, grad_loc,
                              grad_attn_w, B, S, M, D, L, Lq, P, shapes_host, level_start_host, workspace,
                              workspace_bytes, stream);
}

// ---- fused operator: prologue of ops/modules/ms_deform_attn.py:146-155 inside the kernels --------------------
int msda_fused_forward_f32(const float *value, const int64_t *shapes, const int64_t *level_start,
                           const float *offsets, const float *logits, const float *ref, int ref_dim, float *out,
                           int B, int S, int M, int D, int L, int Lq, int P, const int64_t *shapes_host,
                           const int64_t *level_start_host, void *stream) {
  if (!value || !shapes || !level_start || !offsets || !logits || !ref || !out || !shapes_host || !level_start_host)
    return MSDA_E_NULLPTR;
  if (int e = check_dims(B, S, M, D, L, Lq, P)) return e;
  if (!(D == 32 && L == 4 && P == 4) || (ref_dim != 2 && ref_dim != 6)) return MSDA_E_UNSUPPORTED;
  return msda_fused_forward_strided_f32(value, shapes, level_start, offsets, logits, ref, ref_dim, out, B, S, M, D, L, Lq, P,
                                        M * 32, M * 16, shapes_host, level_start_host, stream);
}

int msda_fused_forward_strided_f32(const float *value, const int64_t *shapes, const int64_t *level_start,
                                   const float *offsets, const float *logits, const float *ref, int ref_dim, float *out,
                                   int B, int S, int M, int D, int L, int Lq, int P, int offsets_row_stride,
                                   int logits_row_stride, const int64_t *shapes_host, const int64_t *level_start_host,
                                   void *stream) {
  if (!value || !shapes || !level_start || !offsets || !logits || !ref || !out || !shapes_host || !level_start_host)
    return MSDA_E_NULLPTR;
  if (int e = check_dims(B, S, M, D, L, Lq, P)) return e;
  if (!(D == 32 && L == 4 && P == 4) || (ref_dim != 2 && ref_dim != 6)) return MSDA_E_UNSUPPORTED;
  if (offsets_row_stride < M * 32 || logits_row_stride < M * 16 || (offsets_row_stride & 3) || (logits_row_stride & 3))
    return MSDA_E_SHAPE;
  if (int e = check_host_geometry(shapes_host, level_start_host, L, S)) return e;
  launch_gather<false, true>(value, offsets, logits, nullptr, out, nullptr, nullptr, ref, ref_dim, shapes_host,
                             level_start_host, B, S, M, Lq, (hipStream_t)stream, offsets_row_stride, logits_row_stride);
  return launch_status();
}

int msda_fused_backward_f32(const float *value, const int64_t *shapes, const int64_t *level_start,
                            const float *offsets, const float *logits, const float *ref, int ref_dim,
                            const float *grad_out, float *grad_value, float *grad_offsets, float *grad_logits,
                            int B, int S, int M, int D, int L, int Lq, int P, const int64_t *shapes_host,
                            const int64_t *level_start_host, void *workspace, size_t workspace_bytes, void *stream) {
  return msda_fused_backward_strided_f32(value, shapes, level_start, offsets, logits, ref, ref_dim, grad_out, grad_value,
                                         grad_offsets, grad_logits, B, S, M, D, L, Lq, P, M * 32, M * 16, shapes_host,
                                         level_start_host, workspace, workspace_bytes, stream);
}

int msda_fused_backward_strided_f32(const float *value, const int64_t *shapes, const int64_t *level_start,
                                    const float *offsets, const float *logits, const float *ref, int ref_dim,
                                    const float *grad_out, float *grad_value, float *grad_offsets, float *grad_logits,
                                    int B, int S, int M, int D, int L, int Lq, int P, int offsets_row_stride,
                                    int logits_row_stride, const int64_t *shapes_host, const int64_t *level_start_host,
                                    void *workspace, size_t workspace_bytes, void *stream) {
  if (!ref || !shapes_host || !level_start_host) return MSDA_E_NULLPTR;
  if (!(D == 32 && L == 4 && P == 4) || (ref_dim != 2 && ref_dim != 6) || M * L * 8 > 1024) return MSDA_E_UNSUPPORTED;
  if (offsets_row_stride < M * 32 || logits_row_stride < M * 16 || (offsets_row_stride & 3) || (logits_row_stride & 3))
    return MSDA_E_SHAPE;
  return backward_impl<float>(value, shapes, level_start, offsets, logits, grad_out, grad_value, grad_offsets,
                              grad_logits, B, S, M, D, L, Lq, P, shapes_host, level_start_host, workspace,
                              workspace_bytes, stream, ref, ref_dim, offsets_row_stride, logits_row_stride);
}

// ---- ABI v6: the fused forward hands the backward the locations / weights it evaluated -------------------------------
int msda_fused_save_supported_view(int S, int M, int D, int L, int Lq, int P, int ref_dim, int value_token_stride,
                                   int offsets_row_stride, int logits_row_stride, const int64_t *shapes_host,
                                   const int64_t *level_start_host) {
  // (the tile-window kernels evaluate the 2-d reference-point formula only; 6-d reference points stay on the v5 pair)
  if (!shapes_host || !level_start_host || !(D == 32 && L == 4 && P == 4) || M * L * 8 > 1024 || ref_dim != 2) return 0;
  if (check_host_geometry(shapes_host, level_start_host, L, S)) return 0;
  msda::RowPlan rp;
  // the widest row stride the view entry points will address a plane with (their 32-bit lane offsets: window_fits())
  const long long widest = std::max<long long>(std::max(offsets_row_stride, logits_row_stride), std::max(M * 32, value_token_stride));
  return options().scatter_rows && window_applies(true, shapes_host, level_start_host, Lq, S, widest) &&
         window_applies(false, shapes_host, level_start_host, Lq, S, widest) &&
         msda::make_row_plan(shapes_host, level_start_host, options().scatter_reach, rp) ? 1 : 0;
}

int msda_fused_save_supported(int S, int M, int D, int L, int Lq, int P, int ref_dim, const int64_t *shapes_host,
                              const int64_t *level_start_host) {
  return msda_fused_save_supported_view(S, M, D, L, Lq, P, ref_dim, M * 32, M * 48, M * 48, shapes_host, level_start_host);
}

int msda_fused_forward_save_f32(const float *value, const int64_t *shapes, const int64_t *level_start,
                                const float *offsets, const float *logits, const float *ref, int ref_dim, float *out,
                                float *loc_save, float *attn_save, int B, int S, int M, int D, int L, int Lq, int P,
                                int offsets_row_stride, int logits_row_stride, const int64_t *shapes_host,
                                const int64_t *level_start_host, void *stream) {
  if (!value || !shapes || !level_start || !offsets || !logits || !ref || !out || !loc_save || !attn_save || !shapes_host ||
      !level_start_host)
    return MSDA_E_NULLPTR;
  if (int e = check_dims(B, S, M, D, L, Lq, P)) return e;
  if (!(D == 32 && L == 4 && P == 4) || (ref_dim != 2 && ref_dim != 6)) return MSDA_E_UNSUPPORTED;
  if (offsets_row_stride < M * 32 || logits_row_stride < M * 16 || (offsets_row_stride & 3) || (logits_row_stride & 3))
    return MSDA_E_SHAPE;
  if (!msda_fused_save_supported(S, M, D, L, Lq, P, ref_dim, shapes_host, level_start_host)) return MSDA_E_UNSUPPORTED;
  launch_gather<false, true, true>(value, offsets, logits, nullptr, out, loc_save, attn_save, ref, ref_dim, shapes_host,
                                   level_start_host, B, S, M, Lq, (hipStream_t)stream, offsets_row_stride, logits_row_stride);
  return launch_status();
}

int msda_fused_backward_saved_f32(const float *value, const int64_t *shapes, const int64_t *level_start,
                                  const float *loc_saved, const float *attn_saved, const float *ref, int ref_dim,
                                  const float *grad_out, float *grad_value, float *grad_offsets, float *grad_logits,
                                  int B, int S, int M, int D, int L, int Lq, int P, int offsets_row_stride,
                                  int logits_row_stride, const int64_t *shapes_host, const int64_t *level_start_host,
                                  void *workspace, size_t workspace_bytes, void *stream) {
  if (!ref || !shapes_host || !level_start_host) return MSDA_E_NULLPTR;
  if (!msda_fused_save_supported(S, M, D, L, Lq, P, ref_dim, shapes_host, level_start_host))
    return MSDA_E_UNSUPPORTED;
  if (offsets_row_stride < M * 32 || logits_row_stride < M * 16 || (offsets_row_stride & 3) || (logits_row_stride & 3))
    return MSDA_E_SHAPE;
  return backward_impl<float>(value, shapes, level_start, loc_saved, attn_saved, grad_out, grad_value, grad_offsets,
                              grad_logits, B, S, M, D, L, Lq, P, shapes_host, level_start_host, workspace,
                              workspace_bytes, stream, ref, ref_dim, offsets_row_stride, logits_row_stride, true);
}

int msda_backward_f64(const double *value, const int64_t *shapes, const int64_t *level_start,
                      const double *loc, const double *attn_w, const double *grad_out,
                      double *grad_value, double *grad_loc, double *grad_attn_w, int B, int S, int M,
                      int D, int L, int Lq, int P, const int64_t *shapes_host,
                      const int64_t *level_start_host, void *workspace, size_t workspace_bytes,
                      void *stream) {
  return backward_impl<double>(value, shapes, level_start, loc, attn_w, grad_out, grad_value, grad_loc,
                               grad_attn_w, B, S, M, D, L, Lq, P, shapes_host, level_start_host, workspace,
                               workspace_bytes, stream);
}

#if MSDA_ROWS_STAMP
int msda_debug_rows_stamps(unsigned long long *out8) {
  unsigned long long zero[8] = {0};
  hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(msda::g_rows_stamp), sizeof(zero));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(msda::g_rows_stamp), zero, sizeof(zero));
  return (int)e;
}
#endif
#if MSDA_WIN_STAMP
// measurement builds: read and reset the window kernels' phase clocks ([forward | backward][12])
int msda_debug_stamps(unsigned long long *out24) {
  unsigned long long zero[24] = {0};
  hipError_t e = hipMemcpyFromSymbol(out24, HIP_SYMBOL(msda::g_win_stamp), sizeof(zero));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(msda::g_win_stamp), zero, sizeof(zero));
  return (int)e;
}
#endif
}  // extern "C"
