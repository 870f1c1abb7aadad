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
#include <type_traits>

namespace msda {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void global_cvoid_t;

// BWD = false: out[pair] = sum of sampled rows.   BWD = true: grad_loc / grad_attn_w of the pair.
// FUSED: `loc` / `attw` carry raw sampling offsets / attention logits, `ref` the reference points [B, Lq, 4, ref_dim].
// SAVED (with FUSED): forward -- also store the sampling locations / attention weights it evaluated (`grad_loc` / `grad_attw`
// double as the contiguous save buffers); backward -- `loc` / `attw` ARE those saved tensors (contiguous), only the chain
// rule back to offsets / logits is evaluated here: no softmax, no reference-point arithmetic in the backward's hot loops.
template <bool BWD, bool FUSED, bool SAVED = false>
__global__ __launch_bounds__(kWinThreads, 4) void gather_win_kernel(
    const float *__restrict__ value, const float *__restrict__ loc, const float *__restrict__ attw,
    const float *__restrict__ grad_out, float *__restrict__ out, float *__restrict__ grad_loc,
    float *__restrict__ grad_attw, const float *__restrict__ ref, int ref_dim, const WinTable g, int B, int S, int M,
    int loc_rs, int aw_rs, float *__restrict__ grad_value, int far_reach) {
  // far_reach >= 0 (backward, with msda_scatter_rows.hip): points that are not near_point(.., far_reach) add their
  // grad_value contributions here with global atomics -- the row-tile scatter handles exactly the near ones
  __shared__ float4 win[(kWinMaxRows + 1) * 8];                 // value windows, 8 float4 = one 128-byte row; + the zero row
  constexpr int kZeroOff = kWinMaxRows * 128;                   // byte offset of the zero row

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = threadIdx.x & 7;
  const int tok = M * 32;

  // workgroup -> (batch * head, tile); the tiles of one (batch, head) share blockIdx % 8 = one XCD's L2 (speed only)
  const int n_tiles = g.n_ty * g.n_tx;
  const int bm = (int)(blockIdx.x % 8) + 8 * (int)(blockIdx.x / (8 * n_tiles));
  if (bm >= B * M) return;
  const int tile = (int)((blockIdx.x / 8) % n_tiles);
  const int ty = tile / g.n_tx, tx = tile - ty * g.n_tx;
  const int b = bm / M, m = bm - b * M;

  // ---- tile geometry from the host's separable table (kernel argument: no division, no register arrays)
  const AxisSpec *ay_tab = g.ax[ty], *ax_tab = g.ax[g.n_ty + tx];
  const float *value_bm = value + ((long long)b * S * M + m) * 32;
  if (threadIdx.x < 8) win[kWinMaxRows * 8 + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);

  int base1 = 0, base2 = 0, base3 = 0, first1 = 0, first2 = 0, first3 = 0, n_queries = 0;
  {
    int rows = 0;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      if (l == 1) { base1 = rows; first1 = n_queries; }
      if (l == 2) { base2 = rows; first2 = n_queries; }
      if (l == 3) { base3 = rows; first3 = n_queries; }
      rows += ((int)ay_tab[l].wn * (int)ax_tab[l].wn + 7) & ~7;
      n_queries += (int)ay_tab[l].qn * (int)ax_tab[l].qn;
    }
  }

  // ---- (1a) per-pair inputs of both passes: issued BEFORE the window fill so that their wait does not sit behind it ------
  const int l_mine = sub >> 1;                                           // level of this lane's two points
  float4 lc[kWinMaxPasses];
  float2 aw[kWinMaxPasses];
  int q_lin[kWinMaxPasses];                                              // b * S + q (host: B * S < 2^31)
  int cf_y[kWinMaxPasses], cf_x[kWinMaxPasses];                          // the query's centre floor at this lane's level
  bool live[kWinMaxPasses];
#pragma unroll
  for (int ps = 0; ps < kWinMaxPasses; ++ps) {
    const int i = ps * kWinPairsPerPass + (threadIdx.x >> 3);             // query of the tile
    live[ps] = i < n_queries;
    lc[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
    aw[ps] = make_float2(0.f, 0.f);
    q_lin[ps] = cf_y[ps] = cf_x[ps] = 0;
    if (live[ps]) {
      // query i of the tile -> (level, row, column): the level by three compares, then ONE small division
      const int ql = (i >= first1) + (i >= first2) + (i >= first3);
      const AxisSpec ayq = ay_tab[ql], axq = ax_tab[ql];
      const int k = i - (ql == 0 ? 0 : (ql == 1 ? first1 : (ql == 2 ? first2 : first3)));
      const int dy = (int)(((float)k + 0.5f) / (float)axq.qn), dx = k - dy * axq.qn;          // exact: k < 256
      q_lin[ps] = b * S + g.start[ql] + (ayq.q0 + dy) * g.W[ql] + axq.q0 + dx;                // Lq == S
      if (BWD && far_reach >= 0) {
        cf_y[ps] = centre_floor(ayq.q0 + dy, g.H[ql], g.H[sub >> 1]);
        cf_x[ps] = centre_floor(axq.q0 + dx, g.W[ql], g.W[sub >> 1]);
      }
      const long long ql64 = q_lin[ps];
      if (BWD && SAVED) {
        // the forward's saved tensors are LEVEL-MAJOR, [B, M, L, Lq, P(, 2)]: a level's points of neighbouring queries are
        // neighbours in memory, which is what the row-tile scatter's scan wants (whole lines instead of 32-byte quarters)
        const long long pl = (((long long)(b * M + m) * 4 + (sub >> 1)) * S + (q_lin[ps] - b * S)) * 4 + (sub & 1) * 2;
        lc[ps] = ld4(loc + pl * 2);
        aw[ps] = *reinterpret_cast<const float2 *>(attw + pl);
      } else {
        lc[ps] = ld4(loc + ql64 * loc_rs + m * 32 + sub * 4);
        aw[ps] = *reinterpret_cast<const float2 *>(attw + ql64 * aw_rs + m * 16 + sub * 2);
      }
    }
  }

  // ---- (1b) LDS-DMA fill of the four windows: thread -> (row, 16-byte slot); a wave instruction lands 8 consecutive rows.
  // Level l's window starts at LDS row base_l (multiple of 8), levels in order.
  {
    int rows = 0;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const AxisSpec ay = ay_tab[l], ax = ax_tab[l];
      const int ww = ax.wn, n_rows = (int)ay.wn * ww, Wl = g.W[l];
      const int first_tok = g.start[l] + (int)ay.w0 * Wl + ax.w0;
      const float inv_ww = 1.0f / (float)ww;
      for (int r0 = wave * 8; r0 < n_rows; r0 += kWinThreads / 8) {
        const int r = r0 + (lane >> 3);
        float4 *dst = win + (size_t)(rows + r0) * 8;                           // wave-uniform; lane i lands at dst + i
        if (r < n_rows) {
          const int y = (int)(((float)r + 0.5f) * inv_ww), x = r - y * ww;     // exact: r < 2^11, ww <= 2^7
          const float *src = value_bm + (long long)(first_tok + y * Wl + x) * tok + sub * 4;
          __builtin_amdgcn_global_load_lds((global_cvoid_t *)src, (lds_void_t *)dst, 16, 0, 0);
        }
      }
      rows += (n_rows + 7) & ~7;
    }
  }

  // this lane's level: extent, window, LDS base
  const int Hm = g.H[l_mine], Wm = g.W[l_mine], start_m = g.start[l_mine];
  const AxisSpec aym = ay_tab[l_mine], axm = ax_tab[l_mine];
  const int wy_lo = aym.w0, wy_hi = wy_lo + aym.wn - 1, wx_lo = axm.w0, wx_hi = wx_lo + axm.wn - 1, ww_m = axm.wn;
  const int base_m = l_mine == 0 ? 0 : (l_mine == 1 ? base1 : (l_mine == 2 ? base2 : base3));
  const char *wbytes = reinterpret_cast<const char *>(win);
  const int rot = sub ^ (lane >> 3);                                     // see row() below

  bool waited = false;
#pragma unroll
  for (int ps = 0; ps < kWinMaxPasses; ++ps) {
    if (ps * kWinPairsPerPass >= n_queries) break;                        // wave-uniform
    const long long ql64 = q_lin[ps];
    // ---- this lane's two taps ------------------------------------------------------------------------------------------
    float2 a2 = aw[ps];
    float4 l4 = lc[ps];
    float2 ref_scale = make_float2(1.f, 1.f);                             // FUSED backward: d location / d offset
    if (FUSED && !(BWD && SAVED)) {
      // softmax over the pair's 16 logits (2 per lane), then this lane's two sampling locations (msda_common.h)
      const float mx = group_max(fmaxf(a2.x, a2.y));
      const float e0 = expf(a2.x - mx), e1 = expf(a2.y - mx);
      const float denom = group_sum(e0 + e1);
      a2 = make_float2(e0 / denom, e1 / denom);
      if (live[ps]) {
        const RefScale rs = load_ref(ref + (ql64 * 4 + l_mine) * ref_dim, ref_dim, Hm, Wm);
        ref_scale = make_float2(rs.sx, rs.sy);
        l4 = make_float4(loc_from_offset<4>(rs.rx, l4.x, rs.sx, ref_dim), loc_from_offset<4>(rs.ry, l4.y, rs.sy, ref_dim),
                         loc_from_offset<4>(rs.rx, l4.z, rs.sx, ref_dim), loc_from_offset<4>(rs.ry, l4.w, rs.sy, ref_dim));
        if (!BWD && SAVED) {                                            // hand the backward what was evaluated here (level-major)
          const long long pl = (((long long)(b * M + m) * 4 + l_mine) * S + (q_lin[ps] - b * S)) * 4 + (sub & 1) * 2;
          st4(grad_loc + pl * 2, l4);
          *reinterpret_cast<float2 *>(grad_attw + pl) = a2;
        }
      }
    }
    if (FUSED && BWD && SAVED && live[ps]) {
      const RefScale rs = load_ref(ref + (ql64 * 4 + l_mine) * ref_dim, ref_dim, Hm, Wm);
      ref_scale = make_float2(rs.sx, rs.sy);
    }
    int off[2][4];
    float cw[2][4];          // forward: corner weights x attn_w.  backward: lh, lw, W attn_w, H attn_w
    int far_points = 0;      // backward with the row-tile scatter: which of the two points it does not cover
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      const float lx = k2 ? l4.z : l4.x, ly = k2 ? l4.w : l4.y, wt = k2 ? a2.y : a2.x;
      const Tap<float> tp = make_tap<float>(lx, ly, Hm, Wm);
      // only corners the reference keeps must be reachable (a dropped corner reads the zero row; a point failing the
      // cuh:274 test has all four dropped -- tp.l / tp.r do not include that test)
      const bool inwin = (!tp.t || (tp.y0 >= wy_lo && tp.y0 <= wy_hi)) && (!tp.b || (tp.y1 >= wy_lo && tp.y1 <= wy_hi)) &&
                         (!tp.l || (tp.x0 >= wx_lo && tp.x0 <= wx_hi)) && (!tp.r || (tp.x1 >= wx_lo && tp.x1 <= wx_hi));
      // corner -> LDS byte offset (in window), (token << 4) | 1 (outside: global fallback), or the zero row (dropped)
      const int lds00 = (base_m + (tp.y0 - wy_lo) * ww_m + (tp.x0 - wx_lo)) * 128, ldx = (tp.x1 - tp.x0) * 128,
                ldy = (tp.y1 - tp.y0) * ww_m * 128;
      const int mem00 = start_m + tp.y0 * Wm + tp.x0, mdx = tp.x1 - tp.x0, mdy = (tp.y1 - tp.y0) * Wm;
      auto pick = [&](const int in_lds, const int in_mem, const bool keep) {
        return (keep && live[ps]) ? (inwin ? in_lds : ((in_mem << 4) | 1)) : kZeroOff;
      };
      off[k2][0] = pick(lds00, mem00, tp.t && tp.l);
      off[k2][1] = pick(lds00 + ldx, mem00 + mdx, tp.t && tp.r);
      off[k2][2] = pick(lds00 + ldy, mem00 + mdy, tp.b && tp.l);
      off[k2][3] = pick(lds00 + ldy + ldx, mem00 + mdy + mdx, tp.b && tp.r);
      if (BWD) { cw[k2][0] = tp.lh; cw[k2][1] = tp.lw; cw[k2][2] = (float)Wm * wt; cw[k2][3] = (float)Hm * wt; }
      else { cw[k2][0] = tp.w1 * wt; cw[k2][1] = tp.w2 * wt; cw[k2][2] = tp.w3 * wt; cw[k2][3] = tp.w4 * wt; }
      if (BWD && far_reach >= 0 && live[ps] && tp.valid && !near_point(tp.h_low, tp.w_low, cf_y[ps], cf_x[ps], far_reach))
        far_points |= 1 << k2;
    }
    if (!waited) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // this wave's share of the windows has landed
      __syncthreads();                                                    // ... and everyone else's
      waited = true;
    }

    // one corner: the whole 128-byte row; register s receives 16-byte slot s ^ rot with rot = sub ^ (pair of the wave): the
    // 8 lanes of a pair start on 8 different slots and so do the 8 lanes with equal `sub` (whose rows are neighbours when
    // neighbouring queries sample alike: same bank half) -- and the forward's reduce-scatter needs no lane-dependent selects.
    // A corner outside its window reads the zero row here and is fetched from global memory by `row_outside` (rare; the
    // branch is skipped unless some lane of the wave needs it).
    auto row = [&](const int o, float4 (&v)[8]) {
      const int a = ((o & 1) ? kZeroOff : o) + rot * 16;                  // LDS offsets are multiples of 128
#pragma unroll
      for (int s = 0; s < 8; ++s) v[s] = *reinterpret_cast<const float4 *>(wbytes + (a ^ (s * 16)));
    };
    auto row_outside = [&](const int o, float4 (&v)[8]) {
      const float *p = value_bm + (long long)(o >> 4) * tok;
#pragma unroll
      for (int s = 0; s < 8; ++s) v[s] = ld4(p + 4 * (s ^ rot));
    };

    if (!BWD) {
      float4 acc[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) acc[s] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float w = cw[k2][c];
          float4 v[8];
          row(off[k2][c], v);
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            acc[s].x += w * v[s].x; acc[s].y += w * v[s].y; acc[s].z += w * v[s].z; acc[s].w += w * v[s].w;
          }
          if (off[k2][c] & 1) {
            row_outside(off[k2][c], v);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
              acc[s].x += w * v[s].x; acc[s].y += w * v[s].y; acc[s].z += w * v[s].z; acc[s].w += w * v[s].w;
            }
          }
          // one corner row (8 loads) at a time: the accumulation has to be finished HERE (the asm statements consume it),
          // before the next row's loads -- otherwise the compiler loads all 8 rows first and spills
          asm volatile("" : "+v"(acc[0].x), "+v"(acc[0].y), "+v"(acc[0].z), "+v"(acc[0].w), "+v"(acc[1].x), "+v"(acc[1].y),
                       "+v"(acc[1].z), "+v"(acc[1].w), "+v"(acc[2].x), "+v"(acc[2].y), "+v"(acc[2].z), "+v"(acc[2].w),
                       "+v"(acc[3].x), "+v"(acc[3].y), "+v"(acc[3].z), "+v"(acc[3].w) : : "memory");
          asm volatile("" : "+v"(acc[4].x), "+v"(acc[4].y), "+v"(acc[4].z), "+v"(acc[4].w), "+v"(acc[5].x), "+v"(acc[5].y),
                       "+v"(acc[5].z), "+v"(acc[5].w), "+v"(acc[6].x), "+v"(acc[6].y), "+v"(acc[6].z), "+v"(acc[6].w),
                       "+v"(acc[7].x), "+v"(acc[7].y), "+v"(acc[7].z), "+v"(acc[7].w) : : "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
      // reduce-scatter over the pair's 8 lanes.  acc[s] holds slot s ^ rot, so in the butterfly with partner sub ^ 4 every
      // lane keeps registers 0..3 and hands over 4..7 (the partner's register s ^ 4 is the same slot), then sub ^ 2, sub ^ 1:
      // acc[0] ends as slot rot = channels 4 rot .. 4 rot + 3 summed over the pair -- a permutation of the pair's lanes, still
      // one full 128-byte store per pair
      auto xor4 = [](const float x) {
        int r = __builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x104, 0xF, 0x5, false);      // row_shl:4 -> lanes 0..3 of 8
        r = __builtin_amdgcn_update_dpp(r, __float_as_int(x), 0x114, 0xF, 0xA, false);          // row_shr:4 -> lanes 4..7
        return __int_as_float(r);
      };
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[s].x += xor4(acc[s + 4].x); acc[s].y += xor4(acc[s + 4].y); acc[s].z += xor4(acc[s + 4].z); acc[s].w += xor4(acc[s + 4].w);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        acc[s].x += dpp_x<0x4E>(acc[s + 2].x); acc[s].y += dpp_x<0x4E>(acc[s + 2].y);
        acc[s].z += dpp_x<0x4E>(acc[s + 2].z); acc[s].w += dpp_x<0x4E>(acc[s + 2].w);
      }
      const float4 res = make_float4(acc[0].x + dpp_x<0xB1>(acc[1].x), acc[0].y + dpp_x<0xB1>(acc[1].y),
                                     acc[0].z + dpp_x<0xB1>(acc[1].z), acc[0].w + dpp_x<0xB1>(acc[1].w));
      if (live[ps]) st4(out + (ql64 * M + m) * 32 + rot * 4, res);
    } else {
      // grad_out of the pair, all 32 channels in every lane (8 x 16-byte broadcast loads from L1), rotated as row()
      float4 gq[8];
      const float *gp = grad_out + (ql64 * M + m) * 32;
#pragma unroll
      for (int s = 0; s < 8; ++s) gq[s] = live[ps] ? ld4(gp + 4 * (s ^ rot)) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (far_points) {
        // far points (rare): their corner contributions w_corner attn_w grad_out[q, m, :] (cuh:125-152), which the row-tile
        // scatter leaves out, with global atomics
        float *gv = grad_value + ((long long)b * S * M + m) * 32;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          if (!(far_points & (1 << k2))) continue;
          const float lx = k2 ? l4.z : l4.x, ly = k2 ? l4.w : l4.y, wt = k2 ? a2.y : a2.x;
          const Tap<float> tp = make_tap<float>(lx, ly, Hm, Wm);
          auto corner_add = [&](const int y, const int x, const bool keep, const float w) {
            if (!keep) return;
            float *row_p = gv + (long long)(start_m + y * Wm + x) * tok;
            const float cwt = w * wt;
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) {
              float *d4 = row_p + 4 * (s2 ^ rot);
              atomicAdd(d4, cwt * gq[s2].x); atomicAdd(d4 + 1, cwt * gq[s2].y); atomicAdd(d4 + 2, cwt * gq[s2].z); atomicAdd(d4 + 3, cwt * gq[s2].w);
            }
          };
          corner_add(tp.y0, tp.x0, tp.t && tp.l, tp.w1);
          corner_add(tp.y0, tp.x1, tp.t && tp.r, tp.w2);
          corner_add(tp.y1, tp.x0, tp.b && tp.l, tp.w3);
          corner_add(tp.y1, tp.x1, tp.b && tp.r, tp.w4);
        }
      }
      float4 ol = make_float4(0.f, 0.f, 0.f, 0.f);
      float2 oa = make_float2(0.f, 0.f);
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        float d[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          auto dot8 = [&](const float4 (&v)[8]) {
            float dx = 0.f, dy = 0.f;                                     // two partial sums: independent FMA chains
#pragma unroll
            for (int s = 0; s < 8; s += 2) {
              dx += gq[s].x * v[s].x + gq[s].y * v[s].y + gq[s].z * v[s].z + gq[s].w * v[s].w;
              dy += gq[s + 1].x * v[s + 1].x + gq[s + 1].y * v[s + 1].y + gq[s + 1].z * v[s + 1].z + gq[s + 1].w * v[s + 1].w;
            }
            return dx + dy;
          };
          float4 v[8];
          row(off[k2][c], v);
          d[c] = dot8(v);                                                 // dropped corner: zero row -> 0 (cuh:114-152)
          if (off[k2][c] & 1) {
            row_outside(off[k2][c], v);
            d[c] += dot8(v);
          }
          // one corner row (8 loads) at a time: the dot product has to be finished HERE (the asm consumes it), before the
          // next row's loads -- otherwise the compiler loads all 8 rows first, packs the products across corners and spills
          asm volatile("" : "+v"(d[c]) : : "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
        // grad_attn_w = sum_i w_i d_i;  grad_x = W wt (hh (d2 - d1) + lh (d4 - d3));  grad_y = H wt (hw (d3 - d1) + lw (d4 - d2))
        const float lh = cw[k2][0], lw = cw[k2][1], hh = 1.f - lh, hw = 1.f - lw;
        const float ga = hh * (hw * d[0] + lw * d[1]) + lh * (hw * d[2] + lw * d[3]);
        const float gx = cw[k2][2] * (hh * (d[1] - d[0]) + lh * (d[3] - d[2]));
        const float gy = cw[k2][3] * (hw * (d[2] - d[0]) + lw * (d[3] - d[1]));
        if (k2 == 0) { ol.x = gx; ol.y = gy; oa.x = ga; } else { ol.z = gx; ol.w = gy; oa.y = ga; }
      }
      if (FUSED) {
        // chain rule through the prologue for this lane's own two points: softmax backward and the offset scale
        const float dot = group_sum(oa.x * a2.x + oa.y * a2.y);
        oa = make_float2((oa.x - dot) * a2.x, (oa.y - dot) * a2.y);
        ol = make_float4(offset_grad<4>(ol.x, ref_scale.x, ref_dim), offset_grad<4>(ol.y, ref_scale.y, ref_dim),
                         offset_grad<4>(ol.z, ref_scale.x, ref_dim), offset_grad<4>(ol.w, ref_scale.y, ref_dim));
      }
      if (live[ps]) {
        st4(grad_loc + ql64 * loc_rs + m * 32 + sub * 4, ol);
        *reinterpret_cast<float2 *>(grad_attw + ql64 * aw_rs + m * 16 + sub * 2) = oa;
      }
    }
  }
  if (!waited) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // never leave with LDS-DMA in flight
}

}  // namespace msda
