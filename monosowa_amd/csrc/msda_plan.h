// Device-side launch plan of the self-attention ("encoder") kernels: DIRECTIONAL bounds per (head, level).
//
// In deformable self-attention (Lq == S) every query sits at its token's pixel and samples each level around its own
// position; what differs between heads is WHERE around it: the module initialises head m to look along the angle 2 pi m / M
// (ms_deform_attn.py:106-114) and training bends that pattern without scattering it.  The round-2 kernels sized their LDS value
// windows and their scatter scan regions ISOTROPICALLY (halo 5 / reach 6 pixels on every side for every head); measured, a
// head's points occupy a box a fraction of that size -- e.g. 2 x 5 instead of 11 x 11 pixels per level.  This plan is that
// box, per (head, level): the bounds of
//     d = (h_low - cf_y, w_low - cf_x)      h_low / w_low = top-left pixel of the point's bilinear footprint (cuh:271-281),
//                                           cf = centre_floor() of the query's own pixel at the sampled level
// measured on the call's own sampling offsets by a sampling pre-pass (dir_stats_kernel), turned into tables by one small
// kernel (dir_plan_kernel) and read by the gather / scatter kernels from device memory: no host synchronisation, no state
// between calls, capture-safe.  The bounds are HINTS: a point outside its head's window is fetched from global memory, a point
// outside its head's scan bounds is added with global atomics by the gather kernel (exactly as with the isotropic plan).
//
// Reference semantics are untouched (offsets are unbounded there, ms_deform_attn.py:145-155); only the work layout adapts.
#pragma once
#include "msda_window.h"
#include "msda_scatter_plan.h"

namespace msda {

constexpr int kPlanMaxHeads = 32;        // M * L * 8 <= 1024 threads in the d32 kernels
#ifndef MSDA_PLAN_SIGMAS
#define MSDA_PLAN_SIGMAS 3.0f
#endif
// bounds = [min, max] of the sampled d, cut at mean +- this many standard deviations: the scan (and window) area grows with the
// square of the bounds, the points beyond them (0.3 % per axis of a normal distribution at 3 sigma) take the full-rate atomic /
// global-load paths of the gather kernel
constexpr float kPlanSigmas = MSDA_PLAN_SIGMAS;
constexpr int kPlanClip = 64;            // |d| is clipped here before it enters the statistics
constexpr int kPlanSamples = 256;        // sampled queries per batch element (of Lq): >= 16 k points per (head, level) at B = 16

struct DirBounds { short ylo, yhi, xlo, xhi; };      // inclusive bounds of d
#if defined(__HIPCC__)
__device__ __forceinline__ bool inside_bounds(int dy, int dx, const DirBounds b) {
  return dy >= b.ylo && dy <= b.yhi && dx >= b.xlo && dx <= b.xhi;
}
#endif

// Raw statistics of one (head, level) over the sampled points that pass the cuh:274 test (zeroed by the host per call).
struct DirStats {
  int n;
  // extremes as running MAXIMA of non-negative codes, so that the all-zero fill means "no sample yet":
  // up = max(d + kPlanClip), dn = max(kPlanClip - d)   ->   max d = up - kPlanClip, min d = kPlanClip - dn
  int up_y, dn_y, up_x, dn_x;
  int sum_y, sum_x;
  int pad;
  unsigned long long sq_y, sq_x;
};

// One work item of the row-tile scatter, ready to use: a workgroup reads its descriptor with one (scalar) load instead of
// walking first_item -> order -> n_chunks -> two axis records -> candidate counts, each a dependent global load (phase clocks:
// that chain was 35 % of the scatter's wave time -- its workgroups live for three or four batches only).
struct RowItem {
  short level, chunk, n_chunks, pad0;
  short y0, th, x0, tw;                    // the tile's output rows
  int c_begin, c_end;                      // this chunk's candidates within the tile's scan list
  int cand_off;                            // the tile's scan list: table[head * cand_total + cand_off ...)
  int pad1;
  DirBounds near;                          // the head's near-bounds at this level
  int pad2[2];
};
constexpr int kPlanMaxItems = 768;         // per head; a pyramid with more items keeps the isotropic host plan

// Everything the kernels read for one head.
struct HeadPlan {
  DirBounds win[kWinLevels];               // the measured bounds (mean +- kPlanSigmas sigma, within 16 pixels of the mean): diagnostics
  DirBounds near[kWinLevels];              // scatter bounds (clamped to the host's reach): near <=> inside
  // row-tile scatter
  int n_chunks[4], order[4], first_item[5];
  int n_items;
  RowAxis rax[kRowMaxAxisTiles];
  RowItem items[kPlanMaxItems];
};

// Static inputs of the plan kernel (kernel argument).
struct PlanGeom {
  int H[4], W[4], start[4];
  int S, M;
  int default_halo;                        // bounds when a (head, level) has no valid sample: [-halo, halo - 1]
  int reach;                               // scatter: |d| <= reach at most (capacity of the candidate tables)
  int want_rows;                           // 1: also plan the row-tile scatter
};

inline size_t plan_bytes(int M) { return sizeof(HeadPlan) * (size_t)M; }
inline size_t plan_stats_bytes(int M) { return sizeof(DirStats) * (size_t)M * 4; }

}  // namespace msda
