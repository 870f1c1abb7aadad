// Backward, d32 path (D == 32, L*P == 16, f32): three kernels, no global atomics on the fine levels.
//
// The reference scatters grad_value with 64 global float atomics per output element
// (cuh:125-152); on gfx950 global f32 atomics run at ~1.3 TB/s chip-wide whatever their locality,
// which makes that formulation ~7 ms per encoder layer at B=16.  Here grad_value is accumulated
// on chip instead:
//
//   K1 bwd_prep_kernel      transposes the sampling points into per-(batch, head, level) record
//                           lists {h_im, w_im} / {attn_w} (12 B per point, coalesced for the
//                           scans below); points failing the cuh:274 test get a sentinel.  Per
//                           64-point chunk it also emits the bounding box of the corner pixels,
//                           and per (batch, head) max|attn_w|, max|grad_out|.
//   K2 bwd_scatter_kernel   "tile owner": one workgroup owns a tile of one level of one
//                           (batch, head) -- 256 value rows x 32 64-bit fixed-point accumulators in
//                           LDS.  It walks the level's chunk boxes, scans only chunks that touch
//                           its tile, queues the points with a corner inside it, and adds
//                           w_corner * attn_w * grad_out[b,q,m,:] into LDS with ds_add_u64 (8 lanes
//                           x 4 channels per point, channel order rotated per lane group).
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
// record index o = (((b*M + m)*L + l)*Lq + q)*P + p.  64 consecutive records of one (b, m, l) list = one
// "chunk" (16 queries): K1 also emits the chunk's bounding box of corner pixels (so that tile workgroups can
// skip whole chunks) and its max|attn_w|, max|grad_out| (K1b folds those per (batch, head): the scale of the
// fixed-point accumulators in K2).
struct ChunkBox {
  short y_lo, y_hi, x_lo, x_hi;      // inclusive corner-pixel ranges; empty: y_lo > y_hi
  float a_max, g_max;                // max|attn_w|, max|grad_out| seen by the chunk (folded per (b, m) by K1b)
};

// K1b: bounds[bm] = {max|attn_w|, max|grad_out|} over all chunks of the (batch, head); one workgroup per bm.
// NaN / inf gradients poison the bound and therefore the whole (batch, head) -- as they poison the sums.
__global__ __launch_bounds__(256) void bwd_bounds_kernel(const ChunkBox *__restrict__ boxes, float *__restrict__ bounds,
                                                         int chunks_per_bm) {
  __shared__ float red[2][4];
  const ChunkBox *mine = boxes + (long long)blockIdx.x * chunks_per_bm;
  float a = 0.f, g = 0.f;
  bool bad = false;
  for (int i = threadIdx.x; i < chunks_per_bm; i += 256) {
    const float ca = mine[i].a_max, cg = mine[i].g_max;
    bad |= !(ca == ca) || !(cg == cg);
    a = fmaxf(a, ca);
    g = fmaxf(g, cg);
  }
  if (bad) a = g = __int_as_float(0x7FC00000);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float oa = __shfl_xor(a, o), og = __shfl_xor(g, o);
    a = (oa != oa || a != a) ? __int_as_float(0x7FC00000) : fmaxf(a, oa);
    g = (og != og || g != g) ? __int_as_float(0x7FC00000) : fmaxf(g, og);
  }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = g; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      a = (red[0][w] != red[0][w] || a != a) ? __int_as_float(0x7FC00000) : fmaxf(a, red[0][w]);
      g = (red[1][w] != red[1][w] || g != g) ? __int_as_float(0x7FC00000) : fmaxf(g, red[1][w]);
    }
    bounds[2 * blockIdx.x] = a;
    bounds[2 * blockIdx.x + 1] = g;
  }
}

// Workgroup = (batch, 16 consecutive queries) x all (head, level) lists: thread t serves list t / 8 and the
// two queries 2*(t % 8), +1.  The 16 queries' loc / attn_w / grad_out are 16 KB / 8 KB / 16 KB contiguous
// blocks read in 256-byte runs, every list receives one 64-point chunk = 512 contiguous bytes, and the chunk
// box is an 8-lane reduction.  (The first version walked list-major and fetched 3.3x its input.)
// FUSED: `loc` / `attw` are the raw offsets / logits and `ref` [B, Lq, L, ref_dim] the reference points; the
// prologue (softmax over the pair's 16 logits, location arithmetic) is evaluated here exactly as in the gather
// kernels (msda_gather_rec.hip), so K2 scatters with the same locations and weights the forward used.
template <int P, bool FUSED>
__global__ __launch_bounds__(1024) void bwd_prep_kernel(
    const float *__restrict__ loc, const float *__restrict__ attw, const float *__restrict__ grad_out,
    const int64_t *__restrict__ shapes, float2 *__restrict__ rec_hw, float *__restrict__ rec_aw,
    ChunkBox *__restrict__ boxes, const float *__restrict__ ref, int ref_dim, int M, int L, int Lq,
    int n_chunks_per_list, int loc_rs, int aw_rs) {
  static_assert(P == 4, "one float4 of attention weights / two float4 of locations per (query, head, level)");
  const int chunk = blockIdx.x % n_chunks_per_list;
  const long long b = blockIdx.x / n_chunks_per_list;
  const int ml = threadIdx.x >> 3;                   // m * L + l
  const int j = threadIdx.x & 7;
  const int m = ml / L, l = ml - m * L;
  const long long n_pts = (long long)Lq * P;
  const long long list = (b * M + m) * L + l;
  const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
  int y_lo = 32767, y_hi = -32768, x_lo = 32767, x_hi = -32768;
  float a_abs = 0.f, g_abs = 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int q = chunk * 16 + 2 * j + k;
    if (q >= Lq) continue;
    const long long pair = (b * Lq + q) * M + m;
    const long long q_lin = b * Lq + q;                              // loc_rs / aw_rs: floats per query row (see gather_rec_kernel)
    const float *lp = loc + q_lin * loc_rs + (m * L + l) * P * 2;
    float4 xy01 = ld4(lp), xy23 = ld4(lp + 4);
    float4 aw = ld4(attw + q_lin * aw_rs + (m * L + l) * P);
    if (FUSED) {
      const float *lg = attw + q_lin * aw_rs + m * L * P;          // the pair's 16 logits
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, lg[i]);
      // same association as the 8-lane reduction of the gather kernels: (lane pairs) mirror, xor 2, xor 1
      float e[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) e[i] = expf(lg[i] - mx);
      float part[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) part[i] = e[2 * i] + e[2 * i + 1];
      float m1[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) m1[i] = part[i] + part[7 - i];
      float m2[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) m2[i] = m1[i] + m1[i ^ 2];
      const float denom = m2[0] + m2[1];
      aw = make_float4(e[l * 4] / denom, e[l * 4 + 1] / denom, e[l * 4 + 2] / denom, e[l * 4 + 3] / denom);
      const RefScale rs = load_ref(ref + ((b * Lq + q) * L + l) * ref_dim, ref_dim, H, W);
      xy01 = make_float4(loc_from_offset<4>(rs.rx, xy01.x, rs.sx, ref_dim), loc_from_offset<4>(rs.ry, xy01.y, rs.sy, ref_dim),
                         loc_from_offset<4>(rs.rx, xy01.z, rs.sx, ref_dim), loc_from_offset<4>(rs.ry, xy01.w, rs.sy, ref_dim));
      xy23 = make_float4(loc_from_offset<4>(rs.rx, xy23.x, rs.sx, ref_dim), loc_from_offset<4>(rs.ry, xy23.y, rs.sy, ref_dim),
                         loc_from_offset<4>(rs.rx, xy23.z, rs.sx, ref_dim), loc_from_offset<4>(rs.ry, xy23.w, rs.sy, ref_dim));
    }
    const float4 g0 = ld4(grad_out + pair * 32 + l * 8), g1 = ld4(grad_out + pair * 32 + l * 8 + 4);
    const float xs[4] = {xy01.x, xy01.z, xy23.x, xy23.z}, ys[4] = {xy01.y, xy01.w, xy23.y, xy23.w};
    float2 hw[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const float h_im = scale_loc(ys[p], H), w_im = scale_loc(xs[p], W);
      const bool ok = (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W);
      hw[p] = ok ? make_float2(h_im, w_im) : make_float2(kInvalidCoord, kInvalidCoord);
      if (ok) {
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        y_lo = min(y_lo, h_low); y_hi = max(y_hi, h_low + 1);
        x_lo = min(x_lo, w_low); x_hi = max(x_hi, w_low + 1);
      }
    }
    float4 *dst = reinterpret_cast<float4 *>(rec_hw + list * n_pts + (long long)q * P);
    dst[0] = make_float4(hw[0].x, hw[0].y, hw[1].x, hw[1].y);
    dst[1] = make_float4(hw[2].x, hw[2].y, hw[3].x, hw[3].y);
    st4(rec_aw + list * n_pts + (long long)q * P, aw);
    // fmaxf would drop a NaN: keep it so that it poisons the bound
    const float av[4] = {aw.x, aw.y, aw.z, aw.w};
    const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) a_abs = (av[i] != av[i]) ? av[i] : fmaxf(a_abs, fabsf(av[i]));
#pragma unroll
    for (int i = 0; i < 8; ++i) g_abs = (gv[i] != gv[i]) ? gv[i] : fmaxf(g_abs, fabsf(gv[i]));
  }
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) {                  // the 8 threads of a list are adjacent lanes
    y_lo = min(y_lo, __shfl_xor(y_lo, o)); y_hi = max(y_hi, __shfl_xor(y_hi, o));
    x_lo = min(x_lo, __shfl_xor(x_lo, o)); x_hi = max(x_hi, __shfl_xor(x_hi, o));
    const float oa = __shfl_xor(a_abs, o), og = __shfl_xor(g_abs, o);
    a_abs = (oa != oa || a_abs != a_abs) ? __int_as_float(0x7FC00000) : fmaxf(a_abs, oa);
    g_abs = (og != og || g_abs != g_abs) ? __int_as_float(0x7FC00000) : fmaxf(g_abs, og);
  }
  if (j == 0) {
    ChunkBox bx;
    bx.y_lo = (short)max(y_lo, -32768); bx.y_hi = (short)min(y_hi, 32767);
    bx.x_lo = (short)max(x_lo, -32768); bx.x_hi = (short)min(x_hi, 32767);
    bx.a_max = a_abs;
    bx.g_max = g_abs;
    boxes[list * n_chunks_per_list + chunk] = bx;
  }
}

// ------------------------------------------------------------------------------------------ K2
// LDS float atomics (ds_add_f32) are serialised per lane on gfx950: 0.33 lane-adds/clk/CU measured, against
// 3.9 for ds_add_f64, 7.1 for ds_add_u64 and 11-13 for ds_add_u32 (tools/ubench/lds_atomic*.hip).  The tile
// is therefore accumulated in 64-bit FIXED POINT: a contribution c = w_corner * attn_w * grad_out is scaled
// by a power of two chosen from the (batch, head)'s max|attn_w| * max|grad_out| (K1) such that |c| * scale
// <= 2^kFixBits, rounded once to an integer, and added with ds_add_u64.  Integer sums are exact and
// order-independent; the result is rounded to float once at write-back.  With up to 2^(62 - kFixBits)
// contributions per accumulator there is no overflow (the host checks Lq*P against that).
constexpr int kQueue = 64;               // hit records per wave (one scanned chunk at most)
constexpr int kScatterThreads = 512;
constexpr int kScatterWaves = kScatterThreads / 64;
constexpr int kDrainBatches = 4;         // 8-hit batches whose grad_out rows are fetched together
constexpr int kFixBits = 42;             // |contribution| <= 2^42 after scaling; sums of < 2^20 terms fit int64
constexpr unsigned kNoRow = 0xFFFFu;

// round-to-nearest integer of x (|x| < 2^51) as two's-complement int64, by the 1.5 * 2^52 magic constant
__device__ __forceinline__ unsigned long long to_fixed(double x) {
  const double magic = 6755399441055744.0;
  return (unsigned long long)(__double_as_longlong(x + magic) - __double_as_longlong(magic));
}

// FIXED = true: 64-bit fixed point (ds_add_u64).  FIXED = false: double (ds_add_f64).
template <bool FIXED>
__global__ __launch_bounds__(kScatterThreads, 4) void bwd_scatter_kernel(
    const float2 *__restrict__ rec_hw, const float *__restrict__ rec_aw, const ChunkBox *__restrict__ boxes,
    const float *__restrict__ bounds, const float *__restrict__ grad_out, float *__restrict__ grad_value,
    const BwdPlan plan, int B, int S, int M, int Lq, int P, int n_chunks_per_list,
    const unsigned char *__restrict__ vmask = nullptr) {
  // vmask [B, S] (optional): padded value tokens (ms_deform_attn.py:139-140 fills their value rows with 0): their grad_value
  // rows are written as zero / left out of the atomics
  __shared__ unsigned long long acc[kTileRows * 32];     // 64 KB
  // per wave: hit records {w1..w4 premultiplied by attn_w} , {rows(1,2), rows(3,4), query, -}
  __shared__ float4 queue[kScatterWaves][kQueue][2];     // 16 KB

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

  for (int i = threadIdx.x; i < n_rows * 32; i += kScatterThreads) acc[i] = 0ull;
  __syncthreads();

  // fixed-point scale of this (batch, head): 2^(kFixBits - ceil(log2(bound)))
  const float bound = bounds[2 * bm] * bounds[2 * bm + 1];
  int e2 = 0;
  (void)frexpf(bound, &e2);                               // bound = f * 2^e2, f in [0.5, 1)  =>  bound <= 2^e2
  const bool representable = bound > 0.f && bound < 3.0e38f;      // 0, inf or NaN: nothing meaningful to accumulate
  // scale = 2^(kFixBits - e2) can exceed the float range for tiny bounds: clamp the exponent (coarser than
  // necessary only when bound < 2^-84, where the values are denormal-scale anyway)
  const int sh = min(kFixBits - e2, 126);
  const float scale_f = representable ? ldexpf(1.0f, sh) : 0.f;
  const double inv_scale = representable ? ldexp(1.0, -sh) : (bound == 0.f ? 0.0 : (double)NAN);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane >> 3, sub = lane & 7;
  const long long list_id = (long long)bm * plan.n_levels + l;
  const long long list = list_id * n_pts;                                   // this level's record list
  const float2 *hw_list = rec_hw + list;
  const float *aw_list = rec_aw + list;
  const ChunkBox *box_list = boxes + list_id * n_chunks_per_list;
  const float *go_base = grad_out + ((long long)b * Lq * M + m) * 32 + sub * 4;     // + q * M*32
  float4 (*myq)[2] = queue[wave];
  // rotate the channel order by the group index: in an LDS pass the groups then touch different banks
  const int r0 = grp & 3;
  const int c0 = sub * 4 + r0, c1 = sub * 4 + ((r0 + 1) & 3), c2 = sub * 4 + ((r0 + 2) & 3),
            c3 = sub * 4 + ((r0 + 3) & 3);

  // Adds the queued hits [first, n) into the LDS tile, one hit per 8-lane group and batch; the grad_out rows
  // of up to kDrainBatches batches are requested before the first is consumed.  The scanning lane has already
  // resolved the tap: a record holds the four corner weights (times attn_w) and the LDS rows they go to.
  auto drain = [&](const int first, const int n) {
    float4 wts[kDrainBatches], meta[kDrainBatches], g[kDrainBatches];
    bool active[kDrainBatches];
#pragma unroll
    for (int k = 0; k < kDrainBatches; ++k) {
      const int slot = first + 8 * k + grp;
      active[k] = slot < n;
      wts[k] = myq[slot & (kQueue - 1)][0];
      meta[k] = myq[slot & (kQueue - 1)][1];
      g[k] = active[k] ? ld4(go_base + (long long)__float_as_int(meta[k].z) * M * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < kDrainBatches; ++k) {
      if (!active[k]) continue;
      const float g0 = r0 == 0 ? g[k].x : (r0 == 1 ? g[k].y : (r0 == 2 ? g[k].z : g[k].w));
      const float g1 = r0 == 0 ? g[k].y : (r0 == 1 ? g[k].z : (r0 == 2 ? g[k].w : g[k].x));
      const float g2 = r0 == 0 ? g[k].z : (r0 == 1 ? g[k].w : (r0 == 2 ? g[k].x : g[k].y));
      const float g3 = r0 == 0 ? g[k].w : (r0 == 1 ? g[k].x : (r0 == 2 ? g[k].y : g[k].z));
      const unsigned rows12 = (unsigned)__float_as_int(meta[k].x), rows34 = (unsigned)__float_as_int(meta[k].y);
      auto add_row = [&](unsigned r, float w) {
        if (r == kNoRow) return;                        // corner outside the level or owned by another tile
        unsigned long long *row = acc + r * 32;
        if (FIXED) {
          // power-of-two scaling is exact in float (|c| * scale <= 2^42); one conversion, one rounding
          const float ws = w * scale_f;
          atomicAdd(row + c0, to_fixed((double)(ws * g0)));
          atomicAdd(row + c1, to_fixed((double)(ws * g1)));
          atomicAdd(row + c2, to_fixed((double)(ws * g2)));
          atomicAdd(row + c3, to_fixed((double)(ws * g3)));
        } else {
          double *drow = reinterpret_cast<double *>(row);
          atomicAdd(drow + c0, (double)(w * g0));
          atomicAdd(drow + c1, (double)(w * g1));
          atomicAdd(drow + c2, (double)(w * g2));
          atomicAdd(drow + c3, (double)(w * g3));
        }
      };
      add_row(rows12 & 0xFFFFu, wts[k].x);
      add_row(rows12 >> 16, wts[k].y);
      add_row(rows34 & 0xFFFFu, wts[k].z);
      add_row(rows34 >> 16, wts[k].w);
    }
  };

  // scan: chunks are dealt round-robin to the 8 waves (chunk c -> wave c % 8); a wave looks at 64 of its chunk
  // boxes per step and scans only those that touch the tile
  const int first_chunk = pt_begin >> 6, last_chunk = (pt_end + 63) >> 6;     // [first, last)
  for (int cbase = first_chunk + wave; cbase < last_chunk; cbase += kScatterWaves * 64) {
    const int my_chunk = cbase + lane * kScatterWaves;
    bool touch = false;
    if (my_chunk < last_chunk) {
      const ChunkBox bx = box_list[my_chunk];
      touch = bx.y_lo <= bx.y_hi && bx.y_hi >= y0 && bx.y_lo < y0 + th && bx.x_hi >= x0 && bx.x_lo < x0 + tw;
    }
    unsigned long long todo = __ballot(touch);
    // software pipeline over the touched chunks: the next chunk's records are in flight while this one is tested
    auto fetch = [&](unsigned long long bits, float &aw) {
      float2 hw = make_float2(kInvalidCoord, kInvalidCoord);
      aw = 0.f;
      if (bits) {
        const int idx = (cbase + __builtin_ctzll(bits) * kScatterWaves) * 64 + lane;
        if (idx >= pt_begin && idx < pt_end) { hw = hw_list[idx]; aw = aw_list[idx]; }
      }
      return hw;
    };
    float aw_next;
    float2 hw_next = fetch(todo, aw_next);
    while (todo) {
      const int idx = (cbase + __builtin_ctzll(todo) * kScatterWaves) * 64 + lane;
      todo &= todo - 1;
      const float2 hw = hw_next;
      const float aw = aw_next;
      hw_next = fetch(todo, aw_next);
      // resolve this lane's point against the tile: which of its corners exist (cuh:56-79) and are owned
      bool hit = false;
      float4 wts = make_float4(0.f, 0.f, 0.f, 0.f);
      unsigned rows12 = 0, rows34 = 0;
      if (hw.x > kInvalidCoord) {
        const Tap<float> tp = make_tap_im(hw.x, hw.y, H, W);
        const int ry0 = tp.y0 - y0, ry1 = tp.y1 - y0, rx0 = tp.x0 - x0, rx1 = tp.x1 - x0;
        const bool iy0 = tp.t && (unsigned)ry0 < (unsigned)th, iy1 = tp.b && (unsigned)ry1 < (unsigned)th;
        const bool ix0 = tp.l && (unsigned)rx0 < (unsigned)tw, ix1 = tp.r && (unsigned)rx1 < (unsigned)tw;
        const unsigned q1 = (iy0 && ix0) ? (unsigned)(ry0 * tw + rx0) : kNoRow, q2 = (iy0 && ix1) ? (unsigned)(ry0 * tw + rx1) : kNoRow;
        const unsigned q3 = (iy1 && ix0) ? (unsigned)(ry1 * tw + rx0) : kNoRow, q4 = (iy1 && ix1) ? (unsigned)(ry1 * tw + rx1) : kNoRow;
        hit = (iy0 || iy1) && (ix0 || ix1);
        rows12 = q1 | (q2 << 16);
        rows34 = q3 | (q4 << 16);
        wts = make_float4(tp.w1 * aw, tp.w2 * aw, tp.w3 * aw, tp.w4 * aw);
      }
      const unsigned long long mask = __ballot(hit);
      const int n_hits = __popcll(mask);
      if (hit) {
        const int rank = __popcll(mask & ((1ull << lane) - 1ull));
        myq[rank][0] = wts;
        myq[rank][1] = make_float4(__int_as_float((int)rows12), __int_as_float((int)rows34), __int_as_float(idx / P), 0.f);
      }
      wave_lds_order();                          // records written above are read by other lanes of this wave
      for (int first = 0; first < n_hits; first += 8 * kDrainBatches) drain(first, n_hits);
      wave_lds_order();                          // the queue is rewritten by the next chunk
    }
  }
  __syncthreads();

  // write the tile back (converted to float once)
  const long long tok0 = (long long)b * S + plan.start[l];
  auto to_float = [&](unsigned long long v) {
    return FIXED ? (float)((double)(long long)v * inv_scale) : (float)__longlong_as_double((long long)v);
  };
  if (exclusive) {
    for (int r = threadIdx.x >> 3; r < n_rows; r += kScatterThreads / 8) {
      const int ry = r / tw, rx = r - ry * tw;
      const long long token = tok0 + (long long)(y0 + ry) * W + (x0 + rx);
      float *dst = grad_value + (token * M + m) * 32 + sub * 4;
      const unsigned long long *a = acc + r * 32 + sub * 4;
      const bool padded = vmask && vmask[token];
      st4(dst, padded ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(to_float(a[0]), to_float(a[1]), to_float(a[2]), to_float(a[3])));
    }
  } else {
    const int ch = threadIdx.x & 31;
    for (int r = threadIdx.x >> 5; r < n_rows; r += kScatterThreads / 32) {
      const unsigned long long raw = acc[r * 32 + ch];
      const int ry = r / tw, rx = r - ry * tw;
      const long long token = tok0 + (long long)(y0 + ry) * W + (x0 + rx);
      if ((raw != 0ull || (FIXED && !representable)) && !(vmask && vmask[token]))
        atomicAdd(grad_value + (token * M + m) * 32 + ch, to_float(raw));
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
