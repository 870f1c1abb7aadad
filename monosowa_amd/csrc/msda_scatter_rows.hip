// grad_value of the self-attention shape (Lq == S), d32 path: exclusive tiles of bilinear CELLS accumulated in registers.
// Plan and geometry: msda_scatter_plan.h (static tiling), msda_plan.h (per-head directional scan runs).
//
// A sampling point adds w_corner * attn_w * grad_out[q, m, :] to the four value rows of its bilinear footprint (cuh:125-152).
// Round 2 made every footprint CORNER a "hit" in its value row's bucket and let the row's lanes sum their hits: per corner one
// LDS slot reservation, one record, 128 bytes of grad_out re-read from LDS -- counters: LDS 47 % busy (44 % of it bank
// conflicts), waves parked 56 % of the time; the cost follows the 83.6 M corner hits of a launch, not the candidates scanned.
// Here the unit is the POINT: a workgroup owns a tile of output rows [y0, y0 + th) x [x0, x0 + tw) of one level of one
// (batch, head) and accumulates, per CELL (the top-left pixel of a footprint; cells [y0 - 1, y0 + th) x [x0 - 1, x0 + tw)),
// the four corner-weighted sums  S00, S01, S10, S11 = sum over the cell's points of w_corner * attn_w * grad_out  in registers
// (4 lanes per cell x 8 channels x 4 sums = 32 accumulator registers; 2 lanes x 16 channels would need 64 and spilled the scan).  One reservation, one 2-byte record and ONE read of the grad_out row per point;
// at the end  grad_value[y, x] = S00[y, x] + S01[y, x - 1] + S10[y - 1, x] + S11[y - 1, x - 1]  (three shifted adds through
// LDS, once per workgroup).  Cells on the tile's top / left apron are accumulated by two neighbouring tiles each (the
// candidate scan already covers them: a point reaches a tile's outputs iff its cell lies in the tile's cell range).
// Per batch of 256 candidates (the queries around the tile, msda_plan.h):
//   1. thread = (candidate, one of the level's 4 points): resolves its tap, leaves the four corner coefficients in a per-batch
//      table and -- when the cell is the tile's -- the point's index in the cell's bucket; the batch's grad_out rows go to LDS;
//   2. the cell's 4 lanes walk the bucket: per point 1 + 1 + 2 LDS reads and 16 packed FMAs per lane.
// Single-workgroup tiles are written with plain 128-byte rows (no zero fill, no global atomics); levels whose scan lists
// are split over workgroups add full rows atomically into a zeroed region.
// Only NEAR points are scanned (inside the head's bounds, msda_plan.h); the rest is added by the gather kernel.
#include "msda_common.h"
#include "msda_scatter_plan.h"
#include "msda_plan.h"
#include "msda_bin.hip"

#ifndef MSDA_ROWS_SKIP
#define MSDA_ROWS_SKIP 0         // measurement builds only (bits): 1 no bucket walk, 2 no appends, 16 no grad_out row loads, 32 no tap
                                 // arithmetic, 64 no epilogue (shifted adds + stores), 128 one batch per item (phase budget)
#endif
#ifndef MSDA_ROWS_STAMP
#define MSDA_ROWS_STAMP 0        // measurement builds only: per-phase shader-clock totals (tools/debug/win_stamps.py)
#endif

namespace msda {

#if MSDA_ROWS_STAMP
__device__ unsigned long long g_rows_stamp[8];
#define ROWS_STAMP(i) do { const long long t_ = clock64(); st_acc[i] += t_ - st_t; st_t = t_; } while (0)
#else
#define ROWS_STAMP(i) do { } while (0)
#endif

typedef float rows_v2f __attribute__((ext_vector_type(2)));      // packed pair: v_pk_fma_f32

// Diagnostics (msda_debug_counter("scatter_overflow_rounds")): extra bucket rounds taken because a cell's bucket was full -- bumped on
// that rare path only, by one thread per workgroup and round.  Tests read it to prove that they reached the path.
__device__ unsigned long long g_rows_overflow_rounds;

#ifndef MSDA_ROWS_PRIO
#define MSDA_ROWS_PRIO 0         // measurement builds only: wave priority by phase (1: scan above walk, 2: walk above scan)
#endif
#ifndef MSDA_ROWS_COUNT
#define MSDA_ROWS_COUNT 0        // measurement builds only: scan census (msda_debug_counter("scan_*")) -- what the scan looked at and what it delivered
#endif
// [0] candidates scanned (query, level, tile), [1] candidates with a point in the tile, [2] point tests (taps evaluated),
// [3] points delivered to a cell of the tile, [4] points skipped before their loads (the round-5 per-point reach test:
// tools/debug/experiments/r05_point_bounds_merged_order.patch; 0 in this tree)
__device__ unsigned long long g_rows_scan[5];

// Workgroup barrier for LDS hand-offs that leaves global loads in flight: __syncthreads() carries a workgroup-scope fence
// that lowers to s_waitcnt vmcnt(0) as well, i.e. every barrier would wait for the next batch's prefetch.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// scan lists: one workgroup per (level, tile) -- and per head when the directional plan (msda_plan.h) supplies the runs
__global__ __launch_bounds__(256) void row_candidates_kernel(const RowPlan p, RowCandidate *__restrict__ table,
                                                             const HeadPlan *__restrict__ plans, int n_tiles) {
  const int head = blockIdx.x / n_tiles;                   // 0 without a directional plan (one table for all heads)
  int l = 0, t = blockIdx.x - head * n_tiles;
  while (l < 3 && t >= p.n_ty[l] * p.n_tx[l]) { t -= p.n_ty[l] * p.n_tx[l]; ++l; }
  const int ty = t / p.n_tx[l], tx = t - ty * p.n_tx[l];
  const RowAxis *axes = plans ? plans[head].rax : p.ax;
  const RowAxis ay = axes[p.axis0[l] + ty], ax = axes[p.axis0[l] + p.n_ty[l] + tx];
  RowCandidate *out = table + (long long)head * p.cand_total + p.cand_base[l] + (long long)t * p.cand_stride[l];
  int first = 0;
  for (int lq = 0; lq < 4; ++lq) {
    const int n = (int)ay.qn[lq] * (int)ax.qn[lq], w = ax.qn[lq];
    for (int k = threadIdx.x; k < n; k += 256) {
      const int dy = k / w, dx = k - dy * w;
      const int yq = ay.q0[lq] + dy, xq = ax.q0[lq] + dx;
      RowCandidate c;
      c.token = p.start[lq] + yq * p.W[lq] + xq;
      c.cy = (short)centre_floor(yq, p.H[lq], p.H[l]);
      c.cx = (short)centre_floor(xq, p.W[lq], p.W[l]);
      out[first + k] = c;
    }
    first += n;
  }
}

// FUSED: `loc` / `attw` carry raw sampling offsets / attention logits, `ref` the reference points [B, Lq, 4, ref_dim].
// LEVEL_MAJOR (not FUSED): `loc` / `attw` are the fused forward's saved tensors, [B, M, L, Lq, P(, 2)].
#ifndef MSDA_ROW_WAVES
#define MSDA_ROW_WAVES 4
#endif
template <bool FUSED, bool LEVEL_MAJOR = false>
__global__ __launch_bounds__(kRowThreads, MSDA_ROW_WAVES) void scatter_rows_kernel(
    const float *__restrict__ loc, const float *__restrict__ attw, const float *__restrict__ grad_out,
    float *__restrict__ grad_value, const float *__restrict__ ref, int ref_dim, const RowCandidate *__restrict__ table,
    const RowPlan p, int B, int S, int M, int loc_rs, int aw_rs, const unsigned char *__restrict__ vmask = nullptr,
    const HeadPlan *__restrict__ plans = nullptr, const unsigned *__restrict__ lists = nullptr,
    const unsigned *__restrict__ counts = nullptr, const BinPlan bp = BinPlan(),
    const unsigned char *__restrict__ far_mask = nullptr) {
  // lists / counts / bp (optional): EXACT scan lists (msda_bin.hip) -- a candidate entry is a query with a 4-bit mask of its points
  // that fall into this tile's cells; no bounds, no near / far classes.  `p` supplies the tiling and the (capacity-derived) chunking.
  // far_mask [B, M, L, Lq] (with the lists): points that did not fit SOME tile's list.  A point is far as a whole -- the gather kernel
  // adds all four of its corner rows -- so every tile skips it here, also the one whose list still took it (a cell on an apron is
  // listed by two tiles: with only one of the two lists full the other tile would add its rows a second time).
  // plans (optional): the directional plan (msda_plan.h) -- per head the scan runs, chunking and near-bounds; `p` then only
  // supplies the static tiling and the table capacities (the grid is sized for its isotropic worst case: surplus items exit)
  // vmask [B, S] (optional): padded value tokens -- their grad_value rows come out as zero (ms_deform_attn.py:139-140)
  // one pool: the epilogue exchanges all three shifted corner sums through it at once (kRowThreads x 96 bytes)
  __shared__ float4 lds_pool[kRowBatchQueries * 8 + kRowBatchQueries * 4];
  float4 *const go_lds = lds_pool;                               // grad_out rows of the batch's candidates (this head), 32 KB
  float4 *const coef_tab = lds_pool + kRowBatchQueries * 8;      // per (candidate slot, point): the four corner coefficients, 16 KB
  static_assert(kRowThreads * 6 <= kRowBatchQueries * 12, "the epilogue's exchange fits the pool");
  __shared__ unsigned short bucket[kRowTileRows * (kRowCellCap + 1)];   // per cell: point indices (slot * 4 + point), 16.5 KB
  __shared__ unsigned count[kRowTileRows];
  __shared__ int overflow;                                 // some bucket was full: the batch needs another round

  // blockIdx -> (batch * head, item); all items of one (batch, head) share blockIdx % 8, i.e. one XCD (speed only)
  const int bm = (int)(blockIdx.x % 8) + 8 * (int)(blockIdx.x / (8 * p.n_items));
  if (bm >= B * M) return;
  const int it = (int)((blockIdx.x / 8) % p.n_items);
  const int b = bm / M, m = bm - b * M;
  const HeadPlan *hp = plans ? plans + m : nullptr;
  int l, n_chunks, y0, th, x0, tw, c_begin, c_end;
  const RowCandidate *cands;
  DirBounds nb;              // near <=> the footprint's top-left pixel lies inside these bounds around the query's centre floor
  bool by_list = false;
  const unsigned *list = nullptr;
  if (lists) {
    int oi = 0;
    while (oi < 3 && it >= p.first_item[oi + 1]) ++oi;
    l = p.order[oi];
    const int local = it - p.first_item[oi];
    n_chunks = p.n_chunks[l];
    const int chunk = local % n_chunks, tile = local / n_chunks;
    const int ty = tile / p.n_tx[l], tx = tile - ty * p.n_tx[l];
    const RowAxis ay = p.ax[p.axis0[l] + ty], ax = p.ax[p.axis0[l] + p.n_ty[l] + tx];
    y0 = ay.r0; x0 = ax.r0; th = ay.rn; tw = ax.rn;
    const int n_list = (int)min(counts[(long long)bm * bp.n_tiles + bp.tile0[l] + tile], (unsigned)bp.cap[l]);
    c_begin = chunk * kRowChunkQueries;
    c_end = min(n_list, c_begin + kRowChunkQueries);
    if (c_end <= c_begin) {
      if (chunk > 0) return;                 // nothing left for this chunk (chunk 0 still writes / zeroes the tile when it owns it alone)
      c_end = c_begin;
    }
    list = lists + (long long)bm * bp.plane_entries + bp.list_off[l] + (long long)tile * bp.cap[l];
    cands = nullptr;
    nb.ylo = nb.xlo = nb.yhi = nb.xhi = 0;
    by_list = true;
  } else if (hp) {
    // planned call: ONE descriptor (msda_plan.h: RowItem)
    if (it >= hp->n_items) return;
    const RowItem d = hp->items[it];
    l = d.level; n_chunks = d.n_chunks;
    y0 = d.y0; th = d.th; x0 = d.x0; tw = d.tw;
    c_begin = d.c_begin; c_end = d.c_end;
    cands = table + (long long)m * p.cand_total + d.cand_off;
    nb = d.near;
  } else {
    int oi = 0;
    while (oi < 3 && it >= p.first_item[oi + 1]) ++oi;
    l = p.order[oi];
    const int local = it - p.first_item[oi];
    n_chunks = p.n_chunks[l];
    const int chunk = local % n_chunks, tile = local / n_chunks;
    const int ty = tile / p.n_tx[l], tx = tile - ty * p.n_tx[l];
    const RowAxis ay = p.ax[p.axis0[l] + ty], ax = p.ax[p.axis0[l] + p.n_ty[l] + tx];
    nb.ylo = nb.xlo = (short)-p.reach; nb.yhi = nb.xhi = (short)p.reach;
    y0 = ay.r0; x0 = ax.r0; th = ay.rn; tw = ax.rn;                    // the tile's output rows
    int n_cand = 0;
#pragma unroll
    for (int lq = 0; lq < 4; ++lq) n_cand += (int)ay.qn[lq] * (int)ax.qn[lq];
    c_begin = (int)((long long)n_cand * chunk / n_chunks);
    c_end = (int)((long long)n_cand * (chunk + 1) / n_chunks);
    cands = table + p.cand_base[l] + (long long)tile * p.cand_stride[l];
  }
  const int H = p.H[l], W = p.W[l];
  const int cw = tw + 1, n_cells = (th + 1) * cw;                    // its cells: one more row above, one more column to the left
  constexpr int cap = kRowCellCap, bstride = kRowCellCap + 1;        // (odd stride in 2-byte units: the buckets start on different banks)

  const int tid = threadIdx.x;
  const int slot0 = tid >> 2, pt = tid & 3;                // scan role: candidate slots slot0, slot0 + 128 of the batch; point
  const int cell = tid >> 2, quarter = tid & 3;            // walk role: cell of the tile, quarter of its 32 channels
  if (tid < kRowTileRows) count[tid] = 0;
  if (tid == 0) overflow = 0;
  // this lane's 4 x 8 channel sums as packed pairs (v_pk_fma_f32): sums[j][2 k], [2 k + 1] = channels 4 k .. 4 k + 3 (of the
  // lane's 8) of corner j
  rows_v2f sums[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) sums[j][i] = (rows_v2f){0.f, 0.f};

  // Candidate entries are read two batches ahead, a batch's POINTS one batch ahead (in flight during the previous batch's
  // bucket / walk phases: lds_barrier keeps them in flight); its grad_out rows are requested at the top of the batch and
  // written to LDS behind the taps -- not live across the walk.
  // Batch k takes candidates k, k + n_batches, k + 2 n_batches, ... of the chunk: a batch then samples the whole scan
  // region (a batch of NEIGHBOURS lands on a handful of cells of a coarse level and overflows their buckets).
  struct Points { float4 lg; float2 xy; float wt; RefScale rs; };
  const int n_batches = (MSDA_ROWS_SKIP & 128) ? 1 : (c_end - c_begin + kRowBatchQueries - 1) / kRowBatchQueries;
  auto cand_index = [&](const int k, const int u) {
    return k < n_batches ? c_begin + k + (slot0 + u * (kRowThreads / 4)) * n_batches : c_end;
  };
  auto candidate = [&](const int j) {
    RowCandidate c{-1, 0, 0};
    if (j < c_end) {
      if (by_list) {
        const unsigned e = list[j];
        c.token = (int)(e & 0x0FFFFFFFu);
        // the unit's point mask, minus the points the binning pass handed to the gather kernel's atomic path
        c.cy = (short)((e >> 28) & ~(unsigned)far_mask[((long long)bm * 4 + l) * S + c.token]);
      } else {
        c = cands[j];
      }
    }
    return c;
  };
#if MSDA_ROWS_COUNT
  unsigned n_scan[5] = {0, 0, 0, 0, 0};
#endif
  auto fetch = [&](const RowCandidate c) {
    Points in{};
    if (c.token >= 0) {
      const long long q_lin = (long long)b * S + c.token;                          // Lq == S
      const long long pl = (((long long)(b * M + m) * 4 + l) * S + c.token) * 4 + pt;          // level-major point index
      in.xy = LEVEL_MAJOR ? *reinterpret_cast<const float2 *>(loc + pl * 2)
                          : *reinterpret_cast<const float2 *>(loc + q_lin * loc_rs + ((m * 4 + l) * 4 + pt) * 2);
      if (FUSED) {
        const float *lg = attw + q_lin * aw_rs + m * 16;
        in.lg = ld4(lg + pt * 4);                  // the candidate's 4 threads take one level's four logits each
        in.wt = lg[l * 4 + pt];                    // own logit
        in.rs = load_ref(ref + (q_lin * 4 + l) * ref_dim, ref_dim, H, W);
      } else {
        in.wt = LEVEL_MAJOR ? attw[pl] : attw[q_lin * aw_rs + (m * 4 + l) * 4 + pt];
      }
    }
    return in;
  };
  RowCandidate cur[kRowSub], c_next[kRowSub];
  Points in[kRowSub];
#pragma unroll
  for (int u = 0; u < kRowSub; ++u) {
    cur[u] = candidate(cand_index(0, u));
    in[u] = fetch(cur[u]);
    c_next[u] = candidate(cand_index(1, u));
  }
  lds_barrier();
#if MSDA_ROWS_STAMP
  long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = clock64();
#endif

  for (int k = 0; k < n_batches; ++k) {
    ROWS_STAMP(0);
#if MSDA_ROWS_PRIO == 1
    __builtin_amdgcn_s_setprio(2);           // measurement builds: the load-issuing part of a batch ahead of the other workgroup's walk
#elif MSDA_ROWS_PRIO == 2
    __builtin_amdgcn_s_setprio(0);
#endif
    // ---- 0. the batch's grad_out rows: this thread's 32 bytes of its candidates' rows ---------------------------------------
    float4 g0[kRowSub], g1[kRowSub];
#pragma unroll
    for (int u = 0; u < kRowSub; ++u) {
      g0[u] = g1[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (cur[u].token >= 0 && !(MSDA_ROWS_SKIP & 16)) {
        const float *gp = grad_out + (((long long)b * S + cur[u].token) * M + m) * 32 + pt * 8;
        g0[u] = ld4(gp);
        g1[u] = ld4(gp + 4);
      }
    }
    // ---- 1. this thread's two points -------------------------------------------------------------------------------
    unsigned pend = 0;                                     // bit u: the point of sub-slot u still has to be placed
    int cells[kRowSub];
#pragma unroll
    for (int u = 0; u < kRowSub; ++u) {
      const int slot = slot0 + u * (kRowThreads / 4);
      float lx = in[u].xy.x, ly = in[u].xy.y, wt = in[u].wt;
      if (FUSED) {
        // softmax over the pair's 16 logits (quad reductions over the candidate's 4 threads)
        const float4 lg = in[u].lg;
        float mx = fmaxf(fmaxf(lg.x, lg.y), fmaxf(lg.z, lg.w));
        mx = fmaxf(mx, dpp_x<0xB1>(mx)); mx = fmaxf(mx, dpp_x<0x4E>(mx));
        float sum = expf(lg.x - mx) + expf(lg.y - mx) + expf(lg.z - mx) + expf(lg.w - mx);
        sum += dpp_x<0xB1>(sum); sum += dpp_x<0x4E>(sum);
        wt = expf(wt - mx) / sum;
        lx = loc_from_offset<4>(in[u].rs.rx, lx, in[u].rs.sx, ref_dim);
        ly = loc_from_offset<4>(in[u].rs.ry, ly, in[u].rs.sy, ref_dim);
      }
      cells[u] = 0;
      if (cur[u].token >= 0 && (!(MSDA_ROWS_SKIP & 32) || lx == 123.456f)) {
#if MSDA_ROWS_COUNT
        n_scan[0] += pt == 0;
        ++n_scan[2];
#endif
        const Tap<float> tp = make_tap<float>(lx, ly, H, W);
        if (tp.valid && (by_list ? (cur[u].cy >> pt & 1) != 0 : inside_bounds(tp.h_low - cur[u].cy, tp.w_low - cur[u].cx, nb))) {
          const int cy = tp.h_low - (y0 - 1), cx = tp.w_low - (x0 - 1);
          if ((unsigned)cy <= (unsigned)th && (unsigned)cx <= (unsigned)tw) {
            cells[u] = cy * cw + cx;
            // corner coefficients (a corner outside the level has weight 0, msda_common.h): read by the walk through the
            // point's index -- only a point of this tile is ever looked up
            coef_tab[slot * 4 + pt] = make_float4(tp.w1 * wt, tp.w2 * wt, tp.w3 * wt, tp.w4 * wt);
            if (!(MSDA_ROWS_SKIP & 2)) pend |= 1u << u;
          }
        }
      }
    }
    // the batch's grad_out rows (requested at the top: the taps ran under their latency), then the next batch's points:
    // requested now, consumed in the next iteration
#pragma unroll
    for (int u = 0; u < kRowSub; ++u) {
      // (only for a candidate with a point in this tile: the quad's four threads hold its four points)
      unsigned any = (pend >> u) & 1u;
#if MSDA_ROWS_COUNT
      n_scan[3] += any;
#endif
      any |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)any, 0xB1, 0xF, 0xF, false);
      any |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)any, 0x4E, 0xF, 0xF, false);
#if MSDA_ROWS_COUNT
      n_scan[1] += any && pt == 0;
#endif
      if (any) {
        const int slot = slot0 + u * (kRowThreads / 4);
        go_lds[slot * 8 + ((pt * 2) ^ (slot & 7))] = g0[u];          // chunk c of slot s at c ^ (s & 7): the walk's lanes
        go_lds[slot * 8 + ((pt * 2 + 1) ^ (slot & 7))] = g1[u];      // read one chunk of different rows -> different banks
      }
      cur[u] = c_next[u];
      in[u] = fetch(cur[u]);
      c_next[u] = candidate(cand_index(k + 2, u));
    }
    ROWS_STAMP(1);
#if MSDA_ROWS_PRIO == 1
    __builtin_amdgcn_s_setprio(0);
#elif MSDA_ROWS_PRIO == 2
    __builtin_amdgcn_s_setprio(2);           // (the inverse: appends + walk first)
#endif
    // ---- 2./3. buckets and cell sums; a cell whose bucket overflows (many points on one pixel) takes more rounds -----------
    bool again;
    do {
      unsigned rank[kRowSub];
#pragma unroll
      for (int u = 0; u < kRowSub; ++u) {
        rank[u] = 0xFFFFFFFFu;
        if (pend & (1u << u)) rank[u] = atomicAdd(&count[cells[u]], 1u);
      }
#pragma unroll
      for (int u = 0; u < kRowSub; ++u)
        if (rank[u] < (unsigned)cap) {
          bucket[cells[u] * bstride + rank[u]] = (unsigned short)((slot0 + u * (kRowThreads / 4)) * 4 + pt);
          pend &= ~(1u << u);
        }
      if (pend) overflow = 1;                // rare
      ROWS_STAMP(2);
      lds_barrier();
      ROWS_STAMP(3);
      // read BETWEEN the round's two barriers: every `overflow = 1` of this round lies before the first one, the next write
      // to the flag (the reset below, or the next batch's appends) behind the second
      again = overflow != 0;
      if (cell < n_cells) {
        const int n = min((int)count[cell], cap);
        const unsigned short *bk = bucket + cell * bstride;
        const char *go_b = reinterpret_cast<const char *>(go_lds);
        const int hs = quarter << 5;
        // index i + 1 is requested before point i's rows (bk[n] is at worst the cell's pad slot)
        unsigned idx_next = bk[0];
        for (int i = 0; i < ((MSDA_ROWS_SKIP & 1) ? 0 : n); ++i) {
          const unsigned idx = idx_next;
          idx_next = bk[i + 1];
          const float4 c4 = coef_tab[idx];
          // the row's 16-byte chunks are stored XOR-swizzled by the slot: chunk c of slot s at (c ^ (s & 7)); this lane's
          // chunks are 2 quarter + kk  ->  byte (s * 128 + (s & 7) * 16) ^ (quarter * 32) ^ (kk * 16)
          const unsigned s4 = idx >> 2, a0 = ((s4 << 7) | ((s4 & 7) << 4)) ^ hs;
          float4 a[2];
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) a[kk] = *reinterpret_cast<const float4 *>(go_b + (a0 ^ (kk << 4)));
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const rows_v2f lo = (rows_v2f){a[kk].x, a[kk].y}, hi = (rows_v2f){a[kk].z, a[kk].w};
            sums[0][2 * kk] = __builtin_elementwise_fma((rows_v2f){c4.x, c4.x}, lo, sums[0][2 * kk]);
            sums[0][2 * kk + 1] = __builtin_elementwise_fma((rows_v2f){c4.x, c4.x}, hi, sums[0][2 * kk + 1]);
            sums[1][2 * kk] = __builtin_elementwise_fma((rows_v2f){c4.y, c4.y}, lo, sums[1][2 * kk]);
            sums[1][2 * kk + 1] = __builtin_elementwise_fma((rows_v2f){c4.y, c4.y}, hi, sums[1][2 * kk + 1]);
            sums[2][2 * kk] = __builtin_elementwise_fma((rows_v2f){c4.z, c4.z}, lo, sums[2][2 * kk]);
            sums[2][2 * kk + 1] = __builtin_elementwise_fma((rows_v2f){c4.z, c4.z}, hi, sums[2][2 * kk + 1]);
            sums[3][2 * kk] = __builtin_elementwise_fma((rows_v2f){c4.w, c4.w}, lo, sums[3][2 * kk]);
            sums[3][2 * kk + 1] = __builtin_elementwise_fma((rows_v2f){c4.w, c4.w}, hi, sums[3][2 * kk + 1]);
          }
        }
        if (quarter == 0) count[cell] = 0;   // the cell's four lanes sit in one wave: all have read it
      }
      ROWS_STAMP(4);
      lds_barrier();                         // orders the walk before the next appends / coefficients / grad_out rows
      ROWS_STAMP(5);
      if (again) {                           // everyone has read the flag (in front of the barrier above)
        if (tid == 0) { overflow = 0; atomicAdd(&g_rows_overflow_rounds, 1ull); }
        lds_barrier();
      }
    } while (again);
  }

#if MSDA_ROWS_COUNT
  for (int i = 0; i < 5; ++i) {
    unsigned v = n_scan[i];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&g_rows_scan[i], (unsigned long long)v);
  }
#endif
#if MSDA_ROWS_STAMP
  ROWS_STAMP(6);
  if ((threadIdx.x & 63) == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&g_rows_stamp[i], (unsigned long long)st_acc[i]);
#endif
  if ((MSDA_ROWS_SKIP & 64) && sums[0][0].x != 123.456f) return;
  // ---- corner sums -> rows: out[y, x] = S00[y, x] + S01[y, x - 1] + S10[y - 1, x] + S11[y - 1, x - 1] ---------------------------
  // ONE exchange through LDS (the batch tables: every wave is past its last walk): each lane leaves its S01, S10, S11 (96 bytes),
  // one barrier, each output cell's lanes pick up the three neighbours' sums.  (Three separate 32-byte exchanges cost six barriers:
  // the epilogue was 9 % of the kernel.)
  const int cy = cell / cw, cx = cell - cy * cw;
  const bool is_out = cell < n_cells && cy >= 1 && cx >= 1;
  {
    float4 *mine = lds_pool + (cell * 4 + quarter) * 6;
#pragma unroll
    for (int j = 1; j < 4; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        mine[(j - 1) * 2 + kk] = make_float4(sums[j][2 * kk].x, sums[j][2 * kk].y, sums[j][2 * kk + 1].x, sums[j][2 * kk + 1].y);
    lds_barrier();
    if (is_out) {
      const int src[3] = {cell - 1, cell - cw, cell - cw - 1};              // S01 from the left cell, S10 from above, S11 from above-left
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const float4 v = lds_pool[(src[j] * 4 + quarter) * 6 + j * 2 + kk];
          sums[0][2 * kk] += (rows_v2f){v.x, v.y};
          sums[0][2 * kk + 1] += (rows_v2f){v.z, v.w};
        }
    }
  }

  // ---- write the tile -------------------------------------------------------------------------------------------------
  float4 acc[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) acc[k] = make_float4(sums[0][2 * k].x, sums[0][2 * k].y, sums[0][2 * k + 1].x, sums[0][2 * k + 1].y);
  const long long tok0 = (long long)b * S + p.start[l];
  if (n_chunks == 1) {
    if (is_out) {
      const long long token = tok0 + (long long)(y0 + cy - 1) * W + (x0 + cx - 1);
      float *dst = grad_value + (token * M + m) * 32 + quarter * 8;
      const bool padded = vmask && vmask[token];
#pragma unroll
      for (int k = 0; k < 2; ++k) st4(dst + 4 * k, padded ? make_float4(0.f, 0.f, 0.f, 0.f) : acc[k]);
    }
  } else {
    // several workgroups share the tile: full 128-byte rows of atomics (lane = channel), through LDS
    lds_barrier();                           // every lane has picked up its neighbours' sums: the pool is free again
    float *rows_lds = reinterpret_cast<float *>(go_lds);
    if (is_out) {
#pragma unroll
      for (int k = 0; k < 2; ++k) *reinterpret_cast<float4 *>(rows_lds + cell * 32 + quarter * 8 + 4 * k) = acc[k];
    }
    __syncthreads();
    const int ch = tid & 31;
    for (int rr = tid >> 5; rr < n_cells; rr += kRowThreads / 32) {
      const int ry = rr / cw, rx = rr - ry * cw;
      if (ry < 1 || rx < 1) continue;
      const float v = rows_lds[rr * 32 + ch];
      const long long token = tok0 + (long long)(y0 + ry - 1) * W + (x0 + rx - 1);
      if (v != 0.f && !(vmask && vmask[token])) atomicAdd(grad_value + (token * M + m) * 32 + ch, v);
    }
  }
}

}  // namespace msda
