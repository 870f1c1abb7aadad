// Gather kernels, fourth generation: tile-local VALUE WINDOWS in LDS for the self-attention shape (Lq == S), forward and
// the backward's grad_loc / grad_attn_w pass.  d32 path: D = 32, L = P = 4, f32.  Geometry: msda_window.h.
//
// The third generation (msda_gather_rec.hip) fetches every corner row of levels 0 / 1 from L1 / L2: 10.7 GB of 128-byte
// row reads per launch at B = 16 (18x the algorithmic bytes) at ~21 TB/s, and its (batch, head)-wide LDS stage covers only
// the two coarse levels.  Here a workgroup (16 waves = one per CU, it owns the LDS) takes the <= 256 queries whose pixels
// fall into one image tile -- all four query levels -- for one (batch, head) and keeps, for EVERY sampled level, the value
// window those queries reach with offsets shorter than `halo` pixels (1280x384, 12x16 level-0 tiles, halo 5: 572 + 288 +
// 182 + 72 rows = 139 KB).  All corner reads of in-window taps are ds_read_b128 (bank-conflict-free for the regular
// neighbour patterns of a tile: consecutive queries read consecutive rows); a tap that leaves its window (long learned
// offsets, a caller whose queries are not at their token's pixel) is fetched from global memory instead -- same result.
//
// A workgroup: (1) every lane loads its pair's inputs (lane j of a pair holds points 2j, 2j + 1 after the coalesced load),
// then the four windows are filled by LDS-DMA (global_load_lds_dwordx4: 8 rows = 1 KiB per wave instruction, per-lane
// source row) while the lanes resolve THEIR OWN two points; (2) each lane reads the WHOLE 128-byte corner rows of its two
// points (8 x ds_read_b128 per corner) and works on all 32 channels: forward = 32 accumulators per lane, reduce-scattered
// over the pair's 8 lanes at the end (28 DPP adds) so that lane j holds channels 4j .. 4j+3 = the coalesced store layout;
// backward = the four corner dot products with grad_out in registers, no cross-lane step at all.  A corner the reference
// drops (outside the level, cuh:56-79 / :114-152) points at an all-zero LDS row: no validity masks, and a real value is
// never multiplied by a zero weight.
//
// Why whole rows per lane: with channels split over the pair's lanes (generation three, and the first versions of this
// kernel) every point's tap -- 4 addresses + 4..12 coefficients -- has to reach all 8 lanes.  Through wave-private LDS
// records that exchange cost more LDS cycles than the corner rows themselves (counters: LDS 70 % busy in the compute
// phase, 2/3 of it records), through DPP broadcasts 16 VALU instructions per point.  Owning the points removes the exchange;
// the price is bank conflicts (64 lanes read one 16-byte slot of 64 different rows: the slot a lane starts at is rotated
// by its position in the pair so that a pair's 8 lanes cover all 8 slots).
#include "msda_common.h"
#include "msda_window.h"
#include "msda_plan.h"
#include <type_traits>

#ifndef MSDA_WIN_PK_FWD
#define MSDA_WIN_PK_FWD 0        // forward row FMAs as v_pk_fma_f32
#endif
#ifndef MSDA_WIN_PARITY
#define MSDA_WIN_PARITY 1        // 1: left / right corner reads ordered by LDS row parity (round 2: bank conflicts halved, time unchanged); 0: none
                                 // of that per-point swap logic (A/B round 4: DESIGN 4.0)
#endif
#ifndef MSDA_WIN_NT_SAVES
#define MSDA_WIN_NT_SAVES 0      // 1: the saving forward writes locations / weights with non-temporal stores (A/B: see DESIGN 4.0)
#endif
#ifndef MSDA_WIN_SKIP
#define MSDA_WIN_SKIP 0          // measurement builds only: 1 no fill, 2 no row reads (forward), 4 no output stores (forward), 8 no prefetch, 16 no saves
#endif

#ifndef MSDA_WIN_STAMP
#define MSDA_WIN_STAMP 0         // measurement builds only: per-phase shader-clock totals of the window kernels (tools/debug/win_stamps.py)
#endif

namespace msda {

#if MSDA_WIN_STAMP
__device__ unsigned long long g_win_stamp[2][12];              // [backward][phase], summed over all waves
#define MSDA_STAMP(i) do { const long long t_ = clock64(); st_acc[i] += t_ - st_t; st_t = t_; } while (0)
#else
#define MSDA_STAMP(i) do { } while (0)
#endif

typedef float v2f __attribute__((ext_vector_type(2)));        // packed pair: v_pk_fma_f32 does two FMAs per lane and issue slot
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void global_cvoid_t;

// BWD = false: out[pair] = sum of sampled rows.   BWD = true: grad_loc / grad_attn_w of the pair.
// FUSED: `loc` / `attw` carry raw sampling offsets / attention logits, `ref` the reference points [B, Lq, 4, 2] (the host
// keeps 6-d reference points on the record kernels: ref_dim must be 2 here).
// SAVED (with FUSED): forward -- also store the sampling locations / attention weights it evaluated (`grad_loc` / `grad_attw`
// double as the contiguous save buffers); backward -- `loc` / `attw` ARE those saved tensors (contiguous), only the chain
// rule back to offsets / logits is evaluated here: no softmax, no reference-point arithmetic in the backward's hot loops.
//
// PERSISTENT workgroups: the grid is one workgroup per CU (the host passes n_virtual = the number of (batch * head, tile)
// items); a workgroup walks items blockIdx.x, blockIdx.x + gridDim.x, ... and, within an item, passes of 128 (query, head)
// pairs.  The unit of the software pipeline is one (item, pass): while a unit's corner rows are read from LDS, the NEXT
// unit's per-lane inputs (offsets / logits / reference point, the lane's 16 bytes of grad_out) are already in flight, so
// the only exposed memory phase of an item is the LDS-DMA fill of its windows -- and this lane's tap arithmetic runs under
// that.  With one workgroup per item (the previous form) launch, input latency, fill and row reads ran back to back, 20
// times per CU: measured 0.35 ms forward of which ~0.10 ms was that prologue.
// MASKED: `vmask` [B, S] marks padded value tokens; their rows count as zero (ms_deform_attn.py:139-140) -- folded into the
// corner weights (forward) / the corner dot products (backward) of the lane that owns the point.  `vts`: floats between
// consecutive value tokens.
template <bool BWD, bool FUSED, bool SAVED = false, bool MASKED = false>
__global__ __launch_bounds__(BWD ? kWinThreadsBwd : kWinThreads, BWD ? (kWinThreadsBwd + 255) / 256 : 4) void gather_win_kernel(
    const float *__restrict__ value, const float *__restrict__ loc, const float *__restrict__ attw,
    const float *__restrict__ grad_out, float *__restrict__ out, float *__restrict__ grad_loc,
    float *__restrict__ grad_attw, const float *__restrict__ ref, int ref_dim, const WinTable g, int B, int S, int M,
    int loc_rs, int aw_rs, float *__restrict__ grad_value, int far_reach, int n_virtual, int vts,
    const unsigned char *__restrict__ vmask, const WinQuery *__restrict__ qtab, const HeadPlan *__restrict__ plans = nullptr,
    const unsigned char *__restrict__ far_mask = nullptr) {
  // far_mask [B, M, L, Lq] (backward with the exact scan lists, msda_bin.hip): bit p = point p of the unit did not fit its tile's
  // list and takes the row-atomic path here (normally all zero)
  // plans (optional): the directional plan (msda_plan.h); backward: the head's near-bounds replace the isotropic far_reach
  // far_reach >= 0 (backward, with msda_scatter_rows.hip): points that are not near_point(.., far_reach) add their
  // grad_value contributions here with global atomics -- the row-tile scatter handles exactly the near ones
  __shared__ float4 win[(kWinMaxRows + 1) * 8];                 // value windows, 8 float4 = one 128-byte row; + the zero row
  constexpr int kZeroOff = kWinMaxRows * 128;                   // byte offset of the zero row
  // backward: the last kWinGoRows rows hold grad_out rows, one 1-KiB block per wave (its 8 pairs); the host's tiling
  // leaves them free (kWinMaxRowsBwd)
  constexpr int kGoOff = (kWinMaxRows - kWinGoRows) * 128;
  constexpr int kThreads = BWD ? kWinThreadsBwd : kWinThreads, kPairs = kThreads / 8;       // (query, head) pairs per pass

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = threadIdx.x & 7;
  const int tok = vts;
  const int n_tiles = g.n_ty * g.n_tx;
  if (threadIdx.x < 8) win[kWinMaxRows * 8 + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);

  // ---- an item's scalars: virtual index -> (batch * head, tile); the tiles of one (batch, head) share v % 8, which is
  // blockIdx.x % 8 for every item of this workgroup (gridDim.x is a multiple of 8) = one XCD's L2 (speed only)
  struct Item { int b, m, ty, tx, tile, n_queries; };
  auto make_item = [&](const int v) {
    Item it;
    const int bm = (v & 7) + 8 * (v / (8 * n_tiles));
    it.tile = (v >> 3) % n_tiles;
    it.ty = it.tile / g.n_tx;
    it.tx = it.tile - it.ty * g.n_tx;
    it.b = bm / M;
    it.m = bm - it.b * M;
    const AxisSpec *ay_tab = g.ax[it.ty], *ax_tab = g.ax[g.n_ty + it.tx];
    it.n_queries = (int)ay_tab[0].qn * (int)ax_tab[0].qn + (int)ay_tab[1].qn * (int)ax_tab[1].qn +
                   (int)ay_tab[2].qn * (int)ax_tab[2].qn + (int)ay_tab[3].qn * (int)ax_tab[3].qn;
    return it;
  };
  auto item_valid = [&](const int v) { return v < n_virtual && (v & 7) + 8 * (v / (8 * n_tiles)) < B * M; };

  // ---- a unit's per-lane inputs (lane j of a pair holds points 2j, 2j + 1 after the coalesced load) -----------------------
  const int l_mine = sub >> 1;                                           // level of this lane's two points
  const int Hm = g.H[l_mine], Wm = g.W[l_mine], start_m = g.start[l_mine];
  const float2 ref_scale = make_float2((float)Wm, (float)Hm);            // FUSED: the offset normaliser of this lane's level (ref_dim == 2)
  const float2 ref_rc = make_float2(__fdiv_rn(1.f, ref_scale.x), __fdiv_rn(1.f, ref_scale.y));
  struct In {
    float4 lc, go;           // offsets / locations of the two points; this lane's 16 bytes of the pair's grad_out row
    float2 aw;
    float2 rf;               // FUSED forward: the reference point (x, y) of this lane's level (ref_dim == 2 only, see the host)
    int q, cf_y, cf_x;       // the query's token (-1 for a lane without a query); its centre floor at this lane's level
  };
  // A unit's per-lane inputs in two steps, each a unit ahead of its use: the query-table entry (tile, slot, this lane's level),
  // then -- from the token it names -- the loads.  All per-lane addressing is 32-bit: a scalar base of the unit's (batch, head)
  // plus a lane offset that is one 24-bit multiply-add (host: S and the row strides < 2^24, a plane's span < 2^31 bytes).
  auto load_entry = [&](const Item &it, const int ps) {
    const int i = ps * kPairs + (threadIdx.x >> 3);             // query slot of the tile
    WinQuery e{-1, 0};
    if (i < it.n_queries) e = qtab[(unsigned)((it.tile * kWinMaxQueries + i) * kWinLevels + l_mine)];
    return e;
  };
  const unsigned saved_lane = (unsigned)l_mine * (unsigned)S * 4u + (sub & 1) * 2;      // level-major saved tensors: + q * 4
  auto load_unit = [&](const Item &it, const WinQuery e) {
    In in{};
    in.q = e.q;
    in.cf_y = e.cf >> 16;
    in.cf_x = (int)(short)(e.cf & 0xFFFF);
    if (e.q >= 0) {
      const unsigned q = (unsigned)e.q;
      const long long bS = (long long)it.b * S;
      if (BWD && SAVED) {
        // the forward's saved tensors are LEVEL-MAJOR, [B, M, L, Lq, P(, 2)]: a level's points of neighbouring queries are
        // neighbours in memory, which is what the row-tile scatter's scan wants (whole lines instead of 32-byte quarters)
        const long long plane = (long long)(it.b * M + it.m) * 4 * S * 4;
        const unsigned pl = q * 4u + saved_lane;
        in.lc = ld4(loc + plane * 2 + pl * 2u);
        in.aw = *reinterpret_cast<const float2 *>(attw + plane + pl);
      } else {
        in.lc = ld4(loc + (bS * loc_rs + it.m * 32) + (__umul24(q, (unsigned)loc_rs) + sub * 4u));
        in.aw = *reinterpret_cast<const float2 *>(attw + (bS * aw_rs + it.m * 16) + (__umul24(q, (unsigned)aw_rs) + sub * 2u));
      }
      if (FUSED && !(BWD && SAVED)) in.rf = *reinterpret_cast<const float2 *>(ref + bS * 8 + (q * 8u + l_mine * 2u));
      if (BWD) in.go = ld4(grad_out + (bS * M + it.m) * 32 + (__umul24(q, (unsigned)(M * 32)) + sub * 4u));
    }
    return in;
  };

  char *wbytes = reinterpret_cast<char *>(win);
  const int rot = sub ^ (lane >> 3);                                     // see row() below

  // the walk over (item, pass) units; `ok` = false past the last one
  struct Unit { Item it; int v, ps; bool ok; };
  auto next_unit = [&](const Unit &u) {
    Unit n = u;
    if (!u.ok) return n;
    n.ps = u.ps + 1;
    if (n.ps * kPairs >= u.it.n_queries) {
      n.ps = 0;
      n.v = u.v + gridDim.x;
      while (n.v < n_virtual && !item_valid(n.v)) n.v += gridDim.x;
      n.ok = n.v < n_virtual;
      if (n.ok) n.it = make_item(n.v);
    }
    return n;
  };
  Unit cur;
  cur.v = blockIdx.x;
  cur.ps = 0;
  while (cur.v < n_virtual && !item_valid(cur.v)) cur.v += gridDim.x;
  if (cur.v >= n_virtual) return;
  cur.ok = true;
  cur.it = make_item(cur.v);
  Unit u1 = next_unit(cur);
  In nxt = load_unit(cur.it, load_entry(cur.it, 0));
  WinQuery e1 = u1.ok ? load_entry(u1.it, u1.ps) : WinQuery{-1, 0};
  bool first = true;

#if MSDA_WIN_STAMP
  long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_t = clock64();
#endif
  while (true) {
    MSDA_STAMP(0);
    const Item it = cur.it;
    const int ps = cur.ps;
    In in = (MSDA_WIN_SKIP & 8) ? load_unit(it, load_entry(it, ps)) : nxt;
    // the unit's inputs have arrived long ago; consuming them HERE keeps their wait (which the compiler can only express as
    // vmcnt(0) once the fill below is in flight) in front of the fill
    asm volatile("" : "+v"(in.lc.x), "+v"(in.lc.y), "+v"(in.lc.z), "+v"(in.lc.w), "+v"(in.aw.x), "+v"(in.aw.y), "+v"(in.rf.x),
                 "+v"(in.rf.y) : : "memory");
    if (BWD) asm volatile("" : "+v"(in.go.x), "+v"(in.go.y), "+v"(in.go.z), "+v"(in.go.w) : : "memory");
#if MSDA_WIN_STAMP
    __builtin_amdgcn_s_waitcnt(0x0F70);
    MSDA_STAMP(7);
#endif
    const AxisSpec *ay_tab = g.ax[it.ty], *ax_tab = g.ax[g.n_ty + it.tx];
    const float *value_bm = value + (long long)it.b * S * tok + it.m * 32;
    int base1, base2, base3;
    {
      int rows = 0;
      rows += ((int)ay_tab[0].wn * (int)ax_tab[0].wn + 7) & ~7; base1 = rows;
      rows += ((int)ay_tab[1].wn * (int)ax_tab[1].wn + 7) & ~7; base2 = rows;
      rows += ((int)ay_tab[2].wn * (int)ax_tab[2].wn + 7) & ~7; base3 = rows;
    }

    // ---- this lane's two taps: pure VALU, evaluated UNDER the fill when the unit starts an item.  Fill, taps and the fill's wait
    // sit in one branch: with the wait in a second `if (ps == 0)` the compiler's wait-count pass sees a path from the fill to the
    // LDS reads that skips it and puts its own vmcnt(0) in front of the first row read -- behind the next unit's input loads.
    const bool live = in.q >= 0;
    const unsigned q_u = live ? (unsigned)in.q : 0u;
    // the tile's window of this lane's level: the four levels' specs are wave-uniform (scalar loads), picked per lane
    int wy_lo, wy_hi, wx_lo, wx_hi, ww_m;
    {
      auto packed = [](const AxisSpec a) { return (int)(unsigned short)a.w0 | ((int)(unsigned short)a.wn << 16); };
      const int y0 = packed(ay_tab[0]), y1 = packed(ay_tab[1]), y2 = packed(ay_tab[2]), y3 = packed(ay_tab[3]);
      const int x0 = packed(ax_tab[0]), x1 = packed(ax_tab[1]), x2 = packed(ax_tab[2]), x3 = packed(ax_tab[3]);
      const int yp = l_mine == 0 ? y0 : (l_mine == 1 ? y1 : (l_mine == 2 ? y2 : y3));
      const int xp = l_mine == 0 ? x0 : (l_mine == 1 ? x1 : (l_mine == 2 ? x2 : x3));
      wy_lo = yp & 0xFFFF; wy_hi = wy_lo + (yp >> 16) - 1;
      wx_lo = xp & 0xFFFF; ww_m = xp >> 16; wx_hi = wx_lo + ww_m - 1;
    }
    const int base_m = l_mine == 0 ? 0 : (l_mine == 1 ? base1 : (l_mine == 2 ? base2 : base3));
    float2 a2 = in.aw;
    float4 l4 = in.lc;
    if (FUSED && !(BWD && SAVED)) {
      // softmax over the pair's 16 logits (2 per lane), then this lane's two sampling locations (msda_common.h)
      // (hardware exp2 and reciprocal, ~1 ulp each: the weights are smooth in them; the LOCATIONS below decide floor()s and
      // are rounded exactly like the reference's `offset / normalizer`, ms_deform_attn.py:149-152)
      const float mx = group_max(fmaxf(a2.x, a2.y));
      const float e0 = __expf(a2.x - mx), e1 = __expf(a2.y - mx);
      const float inv = __builtin_amdgcn_rcpf(group_sum(e0 + e1));
      a2 = make_float2(e0 * inv, e1 * inv);
      if (live)
        l4 = make_float4(add_rn(in.rf.x, div_rc(l4.x, ref_scale.x, ref_rc.x)), add_rn(in.rf.y, div_rc(l4.y, ref_scale.y, ref_rc.y)),
                         add_rn(in.rf.x, div_rc(l4.z, ref_scale.x, ref_rc.x)), add_rn(in.rf.y, div_rc(l4.w, ref_scale.y, ref_rc.y)));
    }
    DirBounds near_b;        // backward with the row-tile scatter: the points it covers (everything else is "far": atomics below)
    near_b.ylo = near_b.xlo = (short)-far_reach; near_b.yhi = near_b.xhi = (short)far_reach;
    if (BWD && plans) near_b = plans[it.m].near[l_mine];
    unsigned far_bits = 0;   // this lane's two points: bits 2 (sub & 1), 2 (sub & 1) + 1 of the unit's byte
    if (BWD && far_mask && live)
      far_bits = (unsigned)far_mask[((long long)(it.b * M + it.m) * 4 + l_mine) * S + q_u] >> (2 * (sub & 1));
    int off[2][4];
    float cw[2][4];          // forward: corner weights x attn_w.  backward: lh, lw, W attn_w, H attn_w
    int far_points = 0;      // backward with the row-tile scatter: which of the two points it does not cover
    int padded = 0;          // MASKED backward: bit 4 k2 + c = corner c of point k2 sits on a padded token
    int out_pts = 0;         // bit k2: point k2 lies outside its window (fetched from global memory in the fix-up pass)
    int swapped = 0;         // backward: bit 2 k2 / 2 k2 + 1 = the top / bottom corner pair of point k2 is read right corner first
    const int zero_rot = kZeroOff + rot * 16;
    auto taps = [&]() {
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      const float lx = k2 ? l4.z : l4.x, ly = k2 ? l4.w : l4.y, wt = k2 ? a2.y : a2.x;
      const Tap<float> tp = make_tap<float>(lx, ly, Hm, Wm);
      // The footprint's clamped corner coordinates against the window: one test for the point (a dropped corner's clamped
      // coordinate lies next to the kept ones, so this is only stricter at a level border, where the fallback is still right).
      const bool inwin = tp.y0 >= wy_lo && tp.y1 <= wy_hi && tp.x0 >= wx_lo && tp.x1 <= wx_hi;
      // corner -> LDS byte offset (in window), kOutside (fetched from global memory in the fix-up pass, which re-derives the
      // token from the location: rare), or the zero row (a corner the reference drops, a lane without a query)
      const int lds00 = (base_m + (tp.y0 - wy_lo) * ww_m + (tp.x0 - wx_lo)) * 128 + rot * 16, ldx = (tp.x1 - tp.x0) * 128,
                ldy = (tp.y1 - tp.y0) * ww_m * 128;
      // (the offsets carry this lane's slot rotation: the row loops XOR the slot number into them and nothing else)
      const bool use = inwin && live;
      auto pick = [&](const int in_lds, const bool keep) { return (keep && use) ? in_lds : zero_rot; };
      off[k2][0] = pick(lds00, tp.t && tp.l);
      off[k2][1] = pick(lds00 + ldx, tp.t && tp.r);
      off[k2][2] = pick(lds00 + ldy, tp.b && tp.l);
      off[k2][3] = pick(lds00 + ldy + ldx, tp.b && tp.r);
      if (live && !inwin && tp.valid) out_pts |= 1 << k2;                   // its corners read the zero row in the row loops
      if (BWD) { cw[k2][0] = tp.lh; cw[k2][1] = tp.lw; cw[k2][2] = (float)Wm * wt; cw[k2][3] = (float)Hm * wt; }
      else {
        // a dropped corner reads the zero row: its weight need not be zeroed (the products below are finite)
        const float hw_w = tp.hw * wt, lw_w = tp.lw * wt;
        cw[k2][0] = tp.hh * hw_w; cw[k2][1] = tp.hh * lw_w; cw[k2][2] = tp.lh * hw_w; cw[k2][3] = tp.lh * lw_w;
      }
      if (MASKED) {                                                       // corners on padded tokens (coordinates are clamped into the level)
        const unsigned char *mk = vmask + (long long)it.b * S + start_m;
        const int r0 = tp.y0 * Wm, r1 = tp.y1 * Wm;
        const bool p0 = mk[r0 + tp.x0] != 0, p1 = mk[r0 + tp.x1] != 0, p2 = mk[r1 + tp.x0] != 0, p3 = mk[r1 + tp.x1] != 0;
        if (BWD) padded |= ((p0 ? 1 : 0) | (p1 ? 2 : 0) | (p2 ? 4 : 0) | (p3 ? 8 : 0)) << (4 * k2);
        else { if (p0) cw[k2][0] = 0.f; if (p1) cw[k2][1] = 0.f; if (p2) cw[k2][2] = 0.f; if (p3) cw[k2][3] = 0.f; }
      }
      if (BWD && far_mask) {
        if (live && tp.valid && (far_bits >> k2 & 1)) far_points |= 1 << k2;
      } else if (BWD && far_reach >= 0 && live && tp.valid && !inside_bounds(tp.h_low - in.cf_y, tp.w_low - in.cf_x, near_b))
        far_points |= 1 << k2;
      // Bank parity: a ds_read_b128 is served in groups of 16 lanes over 16 sixteen-byte bank slots; a 128-byte row covers the
      // 8 slots of its parity, and in every group exactly two lanes share a slot position (rot ^ s) -- lanes 16 apart.  The
      // left / right corners of a footprint are neighbouring rows (opposite parity): the lane with bit 4 clear reads the
      // even one first, its partner the odd one, so the two never meet (counters: every group took 2 cycles before).
      if (MSDA_WIN_PARITY) {
        // (a point outside its window keeps its order: the fix-up pass pairs corners and weights by index)
        const bool sw_t = use && (((off[k2][0] >> 7) ^ (lane >> 4)) & 1) != 0, sw_b = use && (((off[k2][2] >> 7) ^ (lane >> 4)) & 1) != 0;
        const int o0 = off[k2][0], o1 = off[k2][1], o2 = off[k2][2], o3 = off[k2][3];
        off[k2][0] = sw_t ? o1 : o0; off[k2][1] = sw_t ? o0 : o1;
        off[k2][2] = sw_b ? o3 : o2; off[k2][3] = sw_b ? o2 : o3;
        if (BWD) swapped |= ((sw_t ? 1 : 0) | (sw_b ? 2 : 0)) << (2 * k2);
        else {
          const float w0 = cw[k2][0], w1 = cw[k2][1], w2 = cw[k2][2], w3 = cw[k2][3];
          cw[k2][0] = sw_t ? w1 : w0; cw[k2][1] = sw_t ? w0 : w1;
          cw[k2][2] = sw_b ? w3 : w2; cw[k2][3] = sw_b ? w2 : w3;
        }
      }
    }
      // the taps are final HERE (the asm statements consume them): otherwise the compiler sinks the tap arithmetic into the
      // row loops below -- behind the fill's wait instead of under it, with all its intermediates alive across the loops
      asm volatile("" : "+v"(off[0][0]), "+v"(off[0][1]), "+v"(off[0][2]), "+v"(off[0][3]), "+v"(off[1][0]), "+v"(off[1][1]),
                   "+v"(off[1][2]), "+v"(off[1][3]) : : "memory");
      asm volatile("" : "+v"(cw[0][0]), "+v"(cw[0][1]), "+v"(cw[0][2]), "+v"(cw[0][3]), "+v"(cw[1][0]), "+v"(cw[1][1]),
                   "+v"(cw[1][2]), "+v"(cw[1][3]), "+v"(out_pts), "+v"(swapped) : : "memory");
    };
    if (ps == 0) {
      // ---- LDS-DMA fill of the four windows: thread -> (row, 16-byte slot); a wave instruction lands 8 consecutive rows.
      // Level l's window starts at LDS row base_l (multiple of 8), levels in order.
      MSDA_STAMP(8);
      if (!first) __syncthreads();                                        // every wave has finished with the previous windows
      MSDA_STAMP(9);
      int rows = 0;
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        const AxisSpec ay = ay_tab[l], ax = ax_tab[l];
        const int ww = ax.wn, n_rows = (int)ay.wn * ww, Wl = g.W[l];
        const int first_tok = g.start[l] + (int)ay.w0 * Wl + ax.w0;
        const float inv_ww = 1.0f / (float)ww;
        for (int r0 = wave * 8; r0 < n_rows; r0 += kThreads / 8) {
          const int r = r0 + (lane >> 3);
          float4 *dst = win + (size_t)(rows + r0) * 8;                           // wave-uniform; lane i lands at dst + i
          if (r < n_rows && !(MSDA_WIN_SKIP & 1)) {
            const int y = (int)(((float)r + 0.5f) * inv_ww), x = r - y * ww;     // exact: r < 2^11, ww <= 2^7
            const float *src = value_bm + (long long)(first_tok + y * Wl + x) * tok + sub * 4;
            __builtin_amdgcn_global_load_lds((global_cvoid_t *)src, (lds_void_t *)dst, 16, 0, 0);
          }
        }
        rows += (n_rows + 7) & ~7;
      }
      MSDA_STAMP(10);
      taps();
      MSDA_STAMP(1);
      // this wave's share of the windows has landed.  The builtin, not inline asm: the wait-count pass must KNOW that nothing
      // is outstanding here.   (vmcnt 0, expcnt 7, lgkmcnt 15)
      __builtin_amdgcn_s_waitcnt(0x0F70);
      MSDA_STAMP(2);
      __syncthreads();                                                    // ... and everyone else's
      MSDA_STAMP(3);
      first = false;
    } else {
      taps();
      MSDA_STAMP(1);
    }

    // ---- the next unit's inputs: requested now, consumed at the top of the next iteration; behind them the table entry of
    // the unit after it --------------------------------------------------------------------------------------------------------
    const bool more = u1.ok;
    if (more && !(MSDA_WIN_SKIP & 8)) nxt = load_unit(u1.it, e1);
    const Unit u2 = next_unit(u1);
    if (u2.ok) e1 = load_entry(u2.it, u2.ps);

    if (!BWD && FUSED && SAVED && live && !(MSDA_WIN_SKIP & 16)) {        // hand the backward what was evaluated here (level-major)
      const long long plane = (long long)(it.b * M + it.m) * 4 * S * 4;
      const unsigned pl = q_u * 4u + saved_lane;
#if MSDA_WIN_NT_SAVES
      // (read again only by the backward, a whole decoder pass later: non-temporal, so that 251 MB of saves do not push the value
      // windows' lines out of L2)
      typedef float nt_v4 __attribute__((ext_vector_type(4)));
      typedef float nt_v2 __attribute__((ext_vector_type(2)));
      __builtin_nontemporal_store((nt_v4){l4.x, l4.y, l4.z, l4.w}, reinterpret_cast<nt_v4 *>(grad_loc + plane * 2 + pl * 2u));
      __builtin_nontemporal_store((nt_v2){a2.x, a2.y}, reinterpret_cast<nt_v2 *>(grad_attw + plane + pl));
#else
      st4(grad_loc + plane * 2 + pl * 2u, l4);
      *reinterpret_cast<float2 *>(grad_attw + plane + pl) = a2;
#endif
    }

    MSDA_STAMP(4);
    // one corner: the whole 128-byte row; register s receives 16-byte slot s ^ rot with rot = sub ^ (pair of the wave): the
    // 8 lanes of a pair start on 8 different slots and so do the 8 lanes with equal `sub` (whose rows are neighbours when
    // neighbouring queries sample alike: same bank half) -- and the forward's reduce-scatter needs no lane-dependent selects.
    // A corner outside its window reads the zero row here and is fetched from global memory by `row_outside` (rare; the
    // branch is skipped unless some lane of the wave needs it).
    // Half a row (4 x ds_read_b128) at a time: 16 registers of row data in flight instead of 32 -- the prefetched inputs of
    // the next unit need the room (launch bound: 128 VGPRs at 16 waves per CU).
    auto half_row = [&](const int o, const int h, float4 (&vv)[4]) {
#pragma unroll
      for (int s = 0; s < 4; ++s) vv[s] = *reinterpret_cast<const float4 *>(wbytes + (o ^ ((4 * h + s) * 16)));
    };
    // value token of corner c of this lane's point k2, -1 for a corner the reference drops (cold path: re-derived)
    auto outside_token = [&](const int k2, const int c) {
      const Tap<float> tp = make_tap<float>(k2 ? l4.z : l4.x, k2 ? l4.w : l4.y, Hm, Wm);
      const bool keep = ((c & 2) ? tp.b : tp.t) && ((c & 1) ? tp.r : tp.l);
      return keep ? start_m + ((c & 2) ? tp.y1 : tp.y0) * Wm + ((c & 1) ? tp.x1 : tp.x0) : -1;
    };

    // Serves the wave's outside points slot by slot (lane-in-pair j, point k2): `add(k2, c, weight, slot data)` for each kept
    // corner c with this lane's 16-byte share of the corner row, then `done(j, k2)`.  Wave-uniform control flow.
    auto fixup_outside = [&](auto add, auto done) {
#pragma unroll 1
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const bool mine = sub == j && (out_pts & (1 << k2));
          if (!__any(mine)) continue;
          const int src = ((lane & ~7) | j) << 2;                       // ds_bpermute address of the pair's lane j
          int tk[4];
          float wk[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int t_own = mine ? outside_token(k2, c) : -1;
            tk[c] = __builtin_amdgcn_ds_bpermute(src, t_own);
            wk[c] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(cw[k2][c])));
          }
          float4 x[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            x[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tk[c] >= 0) x[c] = ld4(value_bm + (long long)tk[c] * tok + 4 * rot);
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) add(k2, c, wk[c], x[c]);
          done(j, k2);
        }
    };
    auto fixup_nop = [](const int, const int) {};

    if (!BWD) {
      float4 acc[8];
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float w = cw[k2][c];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            float4 vv[4];
            if (MSDA_WIN_SKIP & 2) { for (int s = 0; s < 4; ++s) vv[s] = make_float4(w, w, w, w); } else
            half_row(off[k2][c], h, vv);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#if MSDA_WIN_PK_FWD
              // two channels per issue slot (v_pk_fma_f32 / v_pk_mul_f32; the weight is the same in both halves)
              const v2f w2 = (v2f){w, w}, lo = (v2f){vv[s].x, vv[s].y}, hi = (v2f){vv[s].z, vv[s].w};
              v2f a_lo = (v2f){acc[4 * h + s].x, acc[4 * h + s].y}, a_hi = (v2f){acc[4 * h + s].z, acc[4 * h + s].w};
              if (k2 == 0 && c == 0) { a_lo = w2 * lo; a_hi = w2 * hi; }
              else { a_lo = __builtin_elementwise_fma(w2, lo, a_lo); a_hi = __builtin_elementwise_fma(w2, hi, a_hi); }
              acc[4 * h + s] = make_float4(a_lo.x, a_lo.y, a_hi.x, a_hi.y);
#else
              if (k2 == 0 && c == 0)        // the first corner initialises the accumulators (32 moves fewer)
                acc[4 * h + s] = make_float4(w * vv[s].x, w * vv[s].y, w * vv[s].z, w * vv[s].w);
              else {
                acc[4 * h + s].x += w * vv[s].x; acc[4 * h + s].y += w * vv[s].y; acc[4 * h + s].z += w * vv[s].z; acc[4 * h + s].w += w * vv[s].w;
              }
#endif
            }
            // one half row (4 loads) at a time: the accumulation has to be finished HERE (the asm statement consumes it),
            // before the next loads -- otherwise the compiler loads many rows first and spills
            asm volatile("" : "+v"(acc[4 * h].x), "+v"(acc[4 * h].y), "+v"(acc[4 * h].z), "+v"(acc[4 * h].w), "+v"(acc[4 * h + 1].x),
                         "+v"(acc[4 * h + 1].y), "+v"(acc[4 * h + 1].z), "+v"(acc[4 * h + 1].w), "+v"(acc[4 * h + 2].x),
                         "+v"(acc[4 * h + 2].y), "+v"(acc[4 * h + 2].z), "+v"(acc[4 * h + 2].w), "+v"(acc[4 * h + 3].x),
                         "+v"(acc[4 * h + 3].y), "+v"(acc[4 * h + 3].z), "+v"(acc[4 * h + 3].w) : : "memory");
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      MSDA_STAMP(5);
      // points outside their window (their corners read the zero row above): from global memory, by the PAIR.  Every lane of
      // a pair accumulates all 32 channels and the reduce-scatter below sums over the pair's lanes -- so any lane of the pair may
      // add a share: the owner hands the point's four corner tokens and weights to its pair (ds_bpermute), lane i fetches the
      // 16-byte slot its acc[0] stands for (slot `rot`: the pair's 8 lanes cover the 8 slots) of all four corner rows in ONE
      // round trip.  (Round 2 let the owner lane fetch 4 x 8 slots one after the other, each waited for: 64 dependent global
      // loads per wave as soon as ONE of its 128 points left its window -- 3 % of points outside cost the forward 60 %.)
      if (__any(out_pts != 0)) fixup_outside([&](const int, const int, const float w, const float4 x) {
        acc[0].x += w * x.x; acc[0].y += w * x.y; acc[0].z += w * x.z; acc[0].w += w * x.w;
      }, fixup_nop);
      // reduce-scatter over the pair's 8 lanes.  acc[s] holds slot s ^ rot, so in a butterfly with partners sub ^ 7, sub ^ 2,
      // sub ^ 1 every lane keeps its low registers and receives the partner's registers of the same slots (no selects):
      // acc[0] ends as slot rot = channels 4 rot .. 4 rot + 3 summed over the pair -- a permutation of the pair's lanes, still
      // one full 128-byte store per pair
      // first step with partner sub ^ 7 (row_half_mirror: one DPP operand, where sub ^ 4 needs two shifted moves): its register
      // 7 - s = s ^ 7 holds slot s ^ 7 ^ (rot ^ 7) = this lane's slot of register s
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[s].x += dpp_x<0x141>(acc[7 - s].x); acc[s].y += dpp_x<0x141>(acc[7 - s].y);
        acc[s].z += dpp_x<0x141>(acc[7 - s].z); acc[s].w += dpp_x<0x141>(acc[7 - s].w);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        acc[s].x += dpp_x<0x4E>(acc[s + 2].x); acc[s].y += dpp_x<0x4E>(acc[s + 2].y);
        acc[s].z += dpp_x<0x4E>(acc[s + 2].z); acc[s].w += dpp_x<0x4E>(acc[s + 2].w);
      }
      const float4 res = make_float4(acc[0].x + dpp_x<0xB1>(acc[1].x), acc[0].y + dpp_x<0xB1>(acc[1].y),
                                     acc[0].z + dpp_x<0xB1>(acc[1].z), acc[0].w + dpp_x<0xB1>(acc[1].w));
      if (live && (!(MSDA_WIN_SKIP & 4) || res.x == 123.456f))
        st4(out + ((long long)it.b * S * M + it.m) * 32 + (__umul24(q_u, (unsigned)(M * 32)) + rot * 4u), res);
    } else {
      // grad_out of the pair, all 32 channels in every lane, rotated as row(): each lane fetched 16 bytes of the row a unit
      // ahead; the pair's 8 lanes exchange them through the wave's private 1-KiB LDS block (no workgroup barrier: only this
      // wave touches it, and its previous reads are ordered before these writes by the LDS queue)
      float4 gq[8];
      {
        char *gow = wbytes + kGoOff + wave * 1024 + (lane >> 3) * 128;
        *reinterpret_cast<float4 *>(gow + sub * 16) = in.go;              // zeros for a lane that is not live
        wave_lds_order();
#pragma unroll
        for (int s = 0; s < 8; ++s) gq[s] = *reinterpret_cast<const float4 *>(gow + ((s ^ rot) * 16));
        wave_lds_order();
      }
      if (__any(far_points != 0)) {
        // far points: their corner contributions w_corner attn_w grad_out[q, m, :] (cuh:125-152), which the row-tile scatter
        // leaves out, with global atomics -- by HALF-WAVES, lane = channel: one wave instruction adds two full 128-byte rows,
        // the shape global float atomics run at full rate in (MI355X_MICROARCH.md, "Global float atomics").  Round 2 let the
        // owner lane add its 4 x 32 values itself, 64 lanes in 64 different rows per instruction (17x slower): fine for a
        // stray point, a cliff as soon as a trained layer puts a few per cent of its points beyond the scan bounds.
        float *gv = grad_value + ((long long)it.b * S * M + it.m) * 32;
        const char *go_w = wbytes + kGoOff + wave * 1024;                 // the wave's eight grad_out rows (written above)
        const int hw_half = lane >> 5, ch = lane & 31;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          unsigned long long todo = __ballot((far_points >> k2) & 1);
          if (!todo) continue;
          int tkn[4] = {-1, -1, -1, -1};                                  // owner lanes: the point's corner tokens (-1: dropped / padded)
          float cfs[4] = {0.f, 0.f, 0.f, 0.f};
          if (far_points & (1 << k2)) {
            const float lx = k2 ? l4.z : l4.x, ly = k2 ? l4.w : l4.y, wt = k2 ? a2.y : a2.x;
            const Tap<float> tp = make_tap<float>(lx, ly, Hm, Wm);
            auto corner = [&](const int c, const int y, const int x, const bool keep, const float w) {
              const int t = start_m + y * Wm + x;
              if (keep && !(MASKED && vmask[(long long)it.b * S + t])) { tkn[c] = t; cfs[c] = w * wt; }
            };
            corner(0, tp.y0, tp.x0, tp.t && tp.l, tp.w1);
            corner(1, tp.y0, tp.x1, tp.t && tp.r, tp.w2);
            corner(2, tp.y1, tp.x0, tp.b && tp.l, tp.w3);
            corner(3, tp.y1, tp.x1, tp.b && tp.r, tp.w4);
          }
          while (todo) {                                                  // two points per trip: lower / upper half-wave
            const int la = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            int lb = -1;
            if (todo) { lb = __ffsll((long long)todo) - 1; todo &= todo - 1; }
            const int owner = hw_half ? lb : la;
            const int src = owner >= 0 ? owner : lane;                    // (every lane takes part in the exchange)
            const float g = *reinterpret_cast<const float *>(go_w + (src >> 3) * 128 + ch * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const int t = __builtin_amdgcn_ds_bpermute(src << 2, tkn[c]);
              const float cf = __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(cfs[c])));
              if (owner >= 0 && t >= 0) atomicAdd(gv + (long long)t * (M * 32) + ch, cf * g);     // grad_value is dense, whatever `vts` is
            }
          }
        }
      }
      float d[2][4];
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          d[k2][c] = 0.f;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            float4 vv[4];
            half_row(off[k2][c], h, vv);
            // 16 products as 8 packed FMAs into two pair accumulators (independent chains), then three adds: the backward has
            // the registers for aligned pairs (8-wave workgroups), and instruction issue is what bounds these loops
            v2f pa = (v2f){0.f, 0.f}, pb = (v2f){0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; s += 2) {
              const float4 ga = gq[4 * h + s], gb = gq[4 * h + s + 1];
              pa = __builtin_elementwise_fma((v2f){ga.x, ga.y}, (v2f){vv[s].x, vv[s].y}, pa);
              pa = __builtin_elementwise_fma((v2f){ga.z, ga.w}, (v2f){vv[s].z, vv[s].w}, pa);
              pb = __builtin_elementwise_fma((v2f){gb.x, gb.y}, (v2f){vv[s + 1].x, vv[s + 1].y}, pb);
              pb = __builtin_elementwise_fma((v2f){gb.z, gb.w}, (v2f){vv[s + 1].z, vv[s + 1].w}, pb);
            }
            pa += pb;
            d[k2][c] += pa.x + pa.y;                                      // dropped corner: zero row -> 0 (cuh:114-152)
            // one half row (4 loads) at a time: the dot product has to be finished HERE (the asm consumes it), before the
            // next loads -- otherwise the compiler loads many rows first and packs the products across corners
            asm volatile("" : "+v"(d[k2][c]) : : "memory");
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      MSDA_STAMP(5);
      if (swapped) {                                                      // back to corner order
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const bool sw_t = (swapped & (1 << (2 * k2))) != 0, sw_b = (swapped & (2 << (2 * k2))) != 0;
          const float d0 = d[k2][0], d1 = d[k2][1], d2 = d[k2][2], d3 = d[k2][3];
          d[k2][0] = sw_t ? d1 : d0; d[k2][1] = sw_t ? d0 : d1;
          d[k2][2] = sw_b ? d3 : d2; d[k2][3] = sw_b ? d2 : d3;
        }
      }
      if (__any(out_pts != 0)) {                                          // see the forward; here: partial dot products of the lane's slot
        float part[4] = {0.f, 0.f, 0.f, 0.f};
        fixup_outside([&](const int, const int c, const float, const float4 x) {
          part[c] = gq[0].x * x.x + gq[0].y * x.y + gq[0].z * x.z + gq[0].w * x.w;       // this lane's slot of corner c (gq[0] = slot rot)
        }, [&](const int j, const int k2) {
          // one served point done: the pair's partial sums -> the owner's corner dot products
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float tot = group_sum(part[c]);
            if (sub == j) { if (k2 == 0) d[0][c] += tot; else d[1][c] += tot; }
          }
        });
      }
      if (MASKED && padded) {
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (padded & (1 << (4 * k2 + c))) d[k2][c] = 0.f;
      }
      float4 ol = make_float4(0.f, 0.f, 0.f, 0.f);
      float2 oa = make_float2(0.f, 0.f);
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        // grad_attn_w = sum_i w_i d_i;  grad_x = W wt (hh (d2 - d1) + lh (d4 - d3));  grad_y = H wt (hw (d3 - d1) + lw (d4 - d2))
        const float lh = cw[k2][0], lw = cw[k2][1], hh = 1.f - lh, hw = 1.f - lw;
        const float ga = hh * (hw * d[k2][0] + lw * d[k2][1]) + lh * (hw * d[k2][2] + lw * d[k2][3]);
        const float gx = cw[k2][2] * (hh * (d[k2][1] - d[k2][0]) + lh * (d[k2][3] - d[k2][2]));
        const float gy = cw[k2][3] * (hw * (d[k2][2] - d[k2][0]) + lw * (d[k2][3] - d[k2][1]));
        if (k2 == 0) { ol.x = gx; ol.y = gy; oa.x = ga; } else { ol.z = gx; ol.w = gy; oa.y = ga; }
      }
      if (FUSED) {
        // chain rule through the prologue for this lane's own two points: softmax backward and the offset scale
        const float dot = group_sum(oa.x * a2.x + oa.y * a2.y);
        oa = make_float2((oa.x - dot) * a2.x, (oa.y - dot) * a2.y);
        ol = make_float4(div_rc(ol.x, ref_scale.x, ref_rc.x), div_rc(ol.y, ref_scale.y, ref_rc.y),
                         div_rc(ol.z, ref_scale.x, ref_rc.x), div_rc(ol.w, ref_scale.y, ref_rc.y));
      }
      if (live) {
        const long long bS = (long long)it.b * S;
        st4(grad_loc + (bS * loc_rs + it.m * 32) + (__umul24(q_u, (unsigned)loc_rs) + sub * 4u), ol);
        *reinterpret_cast<float2 *>(grad_attw + (bS * aw_rs + it.m * 16) + (__umul24(q_u, (unsigned)aw_rs) + sub * 2u)) = oa;
      }
    }

    MSDA_STAMP(6);
    if (!more) break;
    cur = u1;
    u1 = u2;
  }
#if MSDA_WIN_STAMP
  if (lane == 0)
    for (int i = 0; i < 12; ++i) atomicAdd(&g_win_stamp[BWD ? 1 : 0][i], (unsigned long long)st_acc[i]);
#endif
}

}  // namespace msda
