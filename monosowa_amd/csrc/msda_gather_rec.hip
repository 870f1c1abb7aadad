// Gather kernels, third generation (d32 path: D = 32, L = P = 4, f32): forward and the backward's
// grad_loc / grad_attn_w pass.
//
// Measured on the first two generations: 1152 VALU + 441 SALU instructions per 64 corner loads -- every
// one of the 8 lanes of a (query, head) group recomputed the same bilinear tap, so the kernels were
// VALU-bound and staging value rows in LDS bought nothing.  Here the tap of a sampling point is computed
// ONCE, by the lane that already holds its (x, y, weight) after the coalesced load (lane j of a group
// holds points 2j, 2j+1), and published as a record in wave-private LDS:
//     forward : {4 corner byte offsets, 4 coefficients  w_corner * attn_w}                      32 B
//     backward: {4 corner byte offsets, 4 w_corner, 4 d/dx coefficients, 4 d/dy coefficients}   64 B
// All 8 lanes then stream the 16 records of their pair with broadcast ds_read_b128 and issue each corner
// fetch as base + offset with no further arithmetic: per point 4 loads + 16 (28) FMAs.
// STAGED variants additionally keep the trailing pyramid levels that fit kLdsRows value rows in LDS
// (at 1280x384: levels 2+3 = 600 rows = 77 KB per (batch, head), half of all taps) and serve those taps
// with ds_read_b128; a workgroup is then tied to one (batch, head) and all workgroups of a (batch, head)
// share blockIdx % 8, i.e. one XCD.
#include "msda_common.h"
#include <type_traits>

namespace msda {

constexpr int kLdsRows = 608;              // value rows staged per workgroup
constexpr int kRowPad = 36;                // floats per staged row: 32 + 4 so that rows of equal parity do not
                                           // collide on the 64 LDS banks of ds_read_b128 (measured: the
                                           // unpadded layout spent 6x its LDS cycles in bank conflicts)
constexpr int kStagedThreadsFwd = 1024;    // one workgroup per CU shares the staged rows: 16 waves (32-B records) ...
constexpr int kStagedThreadsBwd = 512;     // ... or 8 waves (64-B records): 76 KB rows + 64 KB records
constexpr int kPlainThreads = 256;

struct GatherGeom {
  int H[4], W[4], start[4];
  int first_lds_level;                     // levels >= this are staged (contiguous tail of the token axis); 4 = none
  int lds_token0;                          // first staged token
  int n_lds_rows;
  int n_chunks;                            // query slices per (batch, head) (staged variants)
};

// Corner offsets (in floats, relative to the (batch, head) base of the source the level lives in) of one
// tap: rows are `row_stride` floats apart (tok in global memory, kRowPad when staged), `base` = level origin.
__device__ __forceinline__ int4 corner_offsets(const Tap<float> &tp, int W, int base, int row_stride) {
  const int r0 = tp.y0 * W, r1 = tp.y1 * W;
  return make_int4(base + (r0 + tp.x0) * row_stride, base + (r0 + tp.x1) * row_stride,
                   base + (r1 + tp.x0) * row_stride, base + (r1 + tp.x1) * row_stride);
}

__device__ __forceinline__ int sel4(int i, const int (&a)[4]) {        // lane-varying index into a uniform array
  return i == 0 ? a[0] : (i == 1 ? a[1] : (i == 2 ? a[2] : a[3]));
}

// BWD = false: out[pair] = sum of sampled rows.   BWD = true: grad_loc / grad_attn_w of the pair.
// FUSED: `loc` / `attw` carry the raw sampling offsets / attention logits, `ref` the per-level reference points
// [B, Lq, 4, ref_dim]; the backward then writes grad_offsets / grad_logits in place of grad_loc / grad_attn_w.
template <bool BWD, bool STAGED, bool FUSED>
__global__ __launch_bounds__(STAGED ? (BWD ? kStagedThreadsBwd : kStagedThreadsFwd) : kPlainThreads, BWD && STAGED ? 2 : 4)
void gather_rec_kernel(
    const float *__restrict__ value, const float *__restrict__ loc, const float *__restrict__ attw,
    const float *__restrict__ grad_out, float *__restrict__ out, float *__restrict__ grad_loc,
    float *__restrict__ grad_attw, const float *__restrict__ ref, int ref_dim, const GatherGeom g, int B, int S,
    int M, int Lq, long long n_pairs, int loc_rs, int aw_rs, int vts = 0, const unsigned char *__restrict__ vmask = nullptr) {
  // vts: floats between consecutive value tokens (0: M * 32; 768 when `value` is a column block of one projection shared by
  // three layers).  vmask [B, S] (optional): padded value tokens read as zero rows (ms_deform_attn.py:139-140) -- folded into
  // the corner weights when the lane builds its records, the row reads stay as they are.
  // loc_rs / aw_rs: floats between consecutive queries in loc / attw (and their gradients): M*32 / M*16 when contiguous;
  // larger when both live in one [B, Lq, 384] buffer (offsets | logits of a merged projection)
  constexpr int kThreads = STAGED ? (BWD ? kStagedThreadsBwd : kStagedThreadsFwd) : kPlainThreads;
  constexpr int kWaves = kThreads / 64;
  constexpr int kRecF4 = BWD ? 4 : 2;                          // float4 slots per record
  // wave-private records: 8 pairs x 16 points; each pair's block is padded by one float4 so that the 8
  // broadcast reads of a wave-instruction fall on different banks
  constexpr int kPairF4 = 16 * kRecF4 + 1;
  __shared__ float4 records[kWaves][8 * kPairF4];
  __shared__ float staged[STAGED ? kLdsRows * kRowPad : 4];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = threadIdx.x & 7, grp8 = lane >> 3;
  const int tok = vts ? vts : M * 32;
  float4 *rec = records[wave] + grp8 * kPairF4;                // this pair's 16 records

  long long pair, pair_stride;
  int n_iter, bm = 0;
  if (STAGED) {
    // workgroup -> (batch*head, query slice); workgroups of one (batch, head) share blockIdx % 8
    bm = (int)(blockIdx.x % 8) + 8 * (int)(blockIdx.x / (8 * g.n_chunks));
    if (bm >= B * M) return;
    const int chunk = (int)((blockIdx.x / 8) % g.n_chunks);
    const int q_begin = (int)((long long)Lq * chunk / g.n_chunks), q_end = (int)((long long)Lq * (chunk + 1) / g.n_chunks);
    const float *value_bm = value + (long long)(bm / M) * S * tok + (bm % M) * 32;
    for (int r = threadIdx.x >> 3; r < g.n_lds_rows; r += kThreads / 8)
      *reinterpret_cast<float4 *>(staged + r * kRowPad + sub * 4) = ld4(value_bm + (long long)(g.lds_token0 + r) * tok + sub * 4);
    __syncthreads();
    const int q0 = q_begin + (threadIdx.x >> 3);
    n_iter = q0 < q_end ? (q_end - q0 + kThreads / 8 - 1) / (kThreads / 8) : 0;
    pair = ((long long)(bm / M) * Lq + q0) * M + (bm % M);
    pair_stride = (long long)(kThreads / 8) * M;
  } else {
    // exact grid, XCD x takes the x-th contiguous eighth of the pairs
    pair = xcd_chunked_block(gridDim.x) * (kThreads / 8) + (threadIdx.x >> 3);
    n_iter = pair < n_pairs ? 1 : 0;
    pair_stride = 0;
  }

  for (int it = 0; it < n_iter; ++it, pair += pair_stride) {
    const int m = (int)(pair % M);
    const int b = (int)(pair / ((long long)M * Lq));
    const float *vb = value + (long long)b * S * tok + m * 32 + sub * 4;       // + corner offset
    const float *sb = staged + sub * 4;
    const long long q_lin = pair / M;
    const long long loc_at = q_lin * loc_rs + m * 32 + sub * 4, aw_at = q_lin * aw_rs + m * 16 + sub * 2;
    float4 lc = ld4(loc + loc_at);                                            // points 2*sub, 2*sub+1: x,y,x,y
    float2 aw = *reinterpret_cast<const float2 *>(attw + aw_at);
    float4 go = make_float4(0.f, 0.f, 0.f, 0.f);
    if (BWD) go = ld4(grad_out + pair * 32 + sub * 4);

    // ---- phase 1: this lane's two taps -> records --------------------------------------------------
    const int l_mine = sub >> 1;                                              // level of points 2*sub, 2*sub+1
    const int H = sel4(l_mine, g.H), W = sel4(l_mine, g.W), lvl_start = sel4(l_mine, g.start);
    const bool mine_staged = STAGED && l_mine >= g.first_lds_level;
    const int row_stride = mine_staged ? kRowPad : tok;
    const int lvl_base = mine_staged ? (lvl_start - g.lds_token0) * kRowPad : lvl_start * tok;
    RefScale rs{};
    if (FUSED) {
      // softmax over the pair's 16 logits (2 per lane), then the two sampling locations of this lane
      const float mx = group_max(fmaxf(aw.x, aw.y));
      const float e0 = expf(aw.x - mx), e1 = expf(aw.y - mx);
      const float denom = group_sum(e0 + e1);
      aw = make_float2(e0 / denom, e1 / denom);
      const long long q_lin = pair / M;                                       // b * Lq + q
      rs = load_ref(ref + (q_lin * 4 + l_mine) * ref_dim, ref_dim, H, W);
      lc = make_float4(loc_from_offset<4>(rs.rx, lc.x, rs.sx, ref_dim), loc_from_offset<4>(rs.ry, lc.y, rs.sy, ref_dim),
                       loc_from_offset<4>(rs.rx, lc.z, rs.sx, ref_dim), loc_from_offset<4>(rs.ry, lc.w, rs.sy, ref_dim));
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float lx = k ? lc.z : lc.x, ly = k ? lc.w : lc.y, wt = k ? aw.y : aw.x;
      const Tap<float> tp = make_tap<float>(lx, ly, H, W);
      const int4 off = corner_offsets(tp, W, lvl_base, row_stride);
      float4 *r = rec + (2 * sub + k) * kRecF4;
      r[0] = make_float4(__int_as_float(off.x), __int_as_float(off.y), __int_as_float(off.z), __int_as_float(off.w));
      float q1 = 1.f, q2 = 1.f, q3 = 1.f, q4 = 1.f;                            // 0 for a corner on a padded token
      if (vmask) {
        const unsigned char *mk = vmask + (long long)b * S + lvl_start;         // (corner coordinates are clamped into the level)
        const int r0 = tp.y0 * W, r1 = tp.y1 * W;
        q1 = mk[r0 + tp.x0] ? 0.f : 1.f; q2 = mk[r0 + tp.x1] ? 0.f : 1.f;
        q3 = mk[r1 + tp.x0] ? 0.f : 1.f; q4 = mk[r1 + tp.x1] ? 0.f : 1.f;
      }
      if (!BWD) {
        r[1] = make_float4(tp.w1 * wt * q1, tp.w2 * wt * q2, tp.w3 * wt * q3, tp.w4 * wt * q4);
      } else {
        // grad_attn_w = sum_i w_i d_i; grad_x = W*wt*(hh(d2-d1) + lh(d4-d3)); grad_y = H*wt*(hw(d3-d1) + lw(d4-d2)),
        // a dropped corner (cuh:114-152) contributing nothing: its coefficients are zeroed
        const float k1 = (tp.t && tp.l) ? q1 : 0.f, k2 = (tp.t && tp.r) ? q2 : 0.f;
        const float k3 = (tp.b && tp.l) ? q3 : 0.f, k4 = (tp.b && tp.r) ? q4 : 0.f;
        const float sx = (float)W * wt, sy = (float)H * wt;
        r[1] = make_float4(tp.w1 * q1, tp.w2 * q2, tp.w3 * q3, tp.w4 * q4);
        r[2] = make_float4(-sx * tp.hh * k1, sx * tp.hh * k2, -sx * tp.lh * k3, sx * tp.lh * k4);
        r[3] = make_float4(-sy * tp.hw * k1, -sy * tp.lw * k2, sy * tp.hw * k3, sy * tp.lw * k4);
      }
    }
    wave_lds_order();

    // ---- phase 2: stream the pair's 16 records ------------------------------------------------------
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 out_loc = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 out_aw = make_float2(0.f, 0.f);
    const bool odd = sub & 1;
    float ga[4], gx[4], gy[4];
    auto points = [&](const int l, auto lds_tag) {
      constexpr bool kInLds = decltype(lds_tag)::value;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const float4 *r = rec + (l * 4 + p) * kRecF4;
        const float4 o = r[0];
        const float4 c = r[1];
        float4 v1, v2, v3, v4;
        if (kInLds) {
          v1 = *reinterpret_cast<const float4 *>(sb + __float_as_int(o.x));
          v2 = *reinterpret_cast<const float4 *>(sb + __float_as_int(o.y));
          v3 = *reinterpret_cast<const float4 *>(sb + __float_as_int(o.z));
          v4 = *reinterpret_cast<const float4 *>(sb + __float_as_int(o.w));
        } else {
          v1 = ld4(vb + __float_as_int(o.x));
          v2 = ld4(vb + __float_as_int(o.y));
          v3 = ld4(vb + __float_as_int(o.z));
          v4 = ld4(vb + __float_as_int(o.w));
        }
        if (!BWD) {
          acc.x += c.x * v1.x + c.y * v2.x + c.z * v3.x + c.w * v4.x;
          acc.y += c.x * v1.y + c.y * v2.y + c.z * v3.y + c.w * v4.y;
          acc.z += c.x * v1.z + c.y * v2.z + c.z * v3.z + c.w * v4.z;
          acc.w += c.x * v1.w + c.y * v2.w + c.z * v3.w + c.w * v4.w;
        } else {
          const float4 cx = r[2], cy = r[3];
          const float d1 = go.x * v1.x + go.y * v1.y + go.z * v1.z + go.w * v1.w;
          const float d2 = go.x * v2.x + go.y * v2.y + go.z * v2.z + go.w * v2.w;
          const float d3 = go.x * v3.x + go.y * v3.y + go.z * v3.z + go.w * v3.w;
          const float d4 = go.x * v4.x + go.y * v4.y + go.z * v4.z + go.w * v4.w;
          ga[p] = c.x * d1 + c.y * d2 + c.z * d3 + c.w * d4;
          gx[p] = cx.x * d1 + cx.y * d2 + cx.z * d3 + cx.w * d4;
          gy[p] = cy.x * d1 + cy.y * d2 + cy.z * d3 + cy.w * d4;
        }
      }
    };
#pragma unroll 1
    for (int l = 0; l < 4; ++l) {                                             // a real loop bounds the loads in flight
      if (STAGED && l >= g.first_lds_level) points(l, std::true_type{});
      else points(l, std::false_type{});
      if (BWD) {
        // channel sums over the 8 lanes: all-reduce lane <-> 7 - lane on all 12 values (both quads then hold
        // the same four pair sums), even lanes collect points 0,1 / odd lanes points 2,3 (sub ^ 1), add sub ^ 2;
        // lanes 2l, 2l+1 own level l's points = the coalesced store layout
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          ga[p] += dpp_x<0x141>(ga[p]);
          gx[p] += dpp_x<0x141>(gx[p]);
          gy[p] += dpp_x<0x141>(gy[p]);
        }
        float rr[6];
        rr[0] = (odd ? gx[2] : gx[0]) + dpp_x<0xB1>(odd ? gx[0] : gx[2]);
        rr[1] = (odd ? gy[2] : gy[0]) + dpp_x<0xB1>(odd ? gy[0] : gy[2]);
        rr[2] = (odd ? gx[3] : gx[1]) + dpp_x<0xB1>(odd ? gx[1] : gx[3]);
        rr[3] = (odd ? gy[3] : gy[1]) + dpp_x<0xB1>(odd ? gy[1] : gy[3]);
        rr[4] = (odd ? ga[2] : ga[0]) + dpp_x<0xB1>(odd ? ga[0] : ga[2]);
        rr[5] = (odd ? ga[3] : ga[1]) + dpp_x<0xB1>(odd ? ga[1] : ga[3]);
#pragma unroll
        for (int i = 0; i < 6; ++i) rr[i] += dpp_x<0x4E>(rr[i]);
        if ((sub >> 1) == l) {
          out_loc = make_float4(rr[0], rr[1], rr[2], rr[3]);
          out_aw = make_float2(rr[4], rr[5]);
        }
      }
    }
    if (!BWD) {
      st4(out + pair * 32 + sub * 4, acc);
    } else {
      if (FUSED) {
        // chain rule through the prologue for this lane's own two points: softmax backward
        // (grad - sum(grad * prob)) * prob and the offset scale
        const float dot = group_sum(out_aw.x * aw.x + out_aw.y * aw.y);
        out_aw = make_float2((out_aw.x - dot) * aw.x, (out_aw.y - dot) * aw.y);
        out_loc = make_float4(offset_grad<4>(out_loc.x, rs.sx, ref_dim), offset_grad<4>(out_loc.y, rs.sy, ref_dim),
                              offset_grad<4>(out_loc.z, rs.sx, ref_dim), offset_grad<4>(out_loc.w, rs.sy, ref_dim));
      }
      st4(grad_loc + loc_at, out_loc);
      *reinterpret_cast<float2 *>(grad_attw + aw_at) = out_aw;
    }
    wave_lds_order();          // records are rewritten by the next iteration
  }
}

}  // namespace msda
