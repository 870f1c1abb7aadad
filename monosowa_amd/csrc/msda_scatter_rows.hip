// grad_value of the self-attention shape (Lq == S), d32 path: exclusive ROW TILES accumulated in registers.
// Plan and geometry: msda_scatter_plan.h.
//
// The tile-owner generation (msda_backward_tiled.hip) is bound by LDS atomics: one ds_add_u64 per (point, corner, channel)
// = 2.7 G of them per launch at B = 16, behind a transposing pre-pass (K1: 0.8 GB of traffic) and with grad_out rows
// fetched per hit from L2.  Here a workgroup owns <= 256 value rows of one level of one (batch, head); each row belongs to 2
// lanes (16 channels = four float4 accumulators each).  The points that can reach the tile come from a per-tile scan list of
// candidate queries (the queries whose pixel centre lies within reach + 1 pixels of the tile; built once per call by
// row_candidates_kernel and shared by all batch x head planes).  Per batch of 128 candidates:
//   1. thread = (candidate, one of the level's 4 points): loads the point straight from loc / attn_w (fused: offsets /
//      logits / reference points -- no transposed copy), resolves its tap; the batch's grad_out rows go to LDS (16 KB);
//   2. every corner that lands in the tile takes a slot in its ROW's bucket (one ds_add_rtn_u32 per corner -- 1/32 of the
//      atomics of the per-channel scheme) and leaves {weight, candidate};
//   3. the row's 2 lanes walk the bucket: 4 x ds_read_b128 of grad_out + 16 FMAs per hit, sums stay in registers.
// The inputs of a batch are fetched one batch ahead (candidate entries two ahead).
// Single-workgroup tiles are written with plain 128-byte rows (no zero fill, no global atomics); levels whose scan lists
// are split over workgroups add full rows atomically into a zeroed region.
// Only NEAR points are scanned (near_point(), msda_common.h); the rest is added by the gather kernel (msda_gather_win.hip).
#include "msda_common.h"
#include "msda_scatter_plan.h"

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

// Workgroup barrier for LDS hand-offs that leaves global loads in flight: __syncthreads() carries a workgroup-scope fence
// that lowers to s_waitcnt vmcnt(0) as well, i.e. every barrier would wait for the next batch's prefetch.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// scan lists: one workgroup per (level, tile)
__global__ __launch_bounds__(256) void row_candidates_kernel(const RowPlan p, RowCandidate *__restrict__ table) {
  int l = 0, t = blockIdx.x;
  while (l < 3 && t >= p.n_ty[l] * p.n_tx[l]) { t -= p.n_ty[l] * p.n_tx[l]; ++l; }
  const int ty = t / p.n_tx[l], tx = t - ty * p.n_tx[l];
  const RowAxis ay = p.ax[p.axis0[l] + ty], ax = p.ax[p.axis0[l] + p.n_ty[l] + tx];
  RowCandidate *out = table + p.cand_base[l] + (long long)t * p.cand_stride[l];
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
template <bool FUSED, bool LEVEL_MAJOR = false>
__global__ __launch_bounds__(kRowThreads, kRowSub == 1 ? 6 : 4) void scatter_rows_kernel(
    const float *__restrict__ loc, const float *__restrict__ attw, const float *__restrict__ grad_out,
    float *__restrict__ grad_value, const float *__restrict__ ref, int ref_dim, const RowCandidate *__restrict__ table,
    const RowPlan p, int B, int S, int M, int loc_rs, int aw_rs, const unsigned char *__restrict__ vmask = nullptr) {
  // vmask [B, S] (optional): padded value tokens -- their grad_value rows come out as zero (ms_deform_attn.py:139-140)
  __shared__ float4 go_lds[kRowBatchQueries * 8];          // grad_out rows of the batch's candidates (this head), 16 KB
  __shared__ uint2 bucket[kRowBucketEntries];              // per row: {weight bits, candidate slot}, 32 KB
  __shared__ unsigned count[kRowTileRows];
  __shared__ int overflow;                                 // some bucket was full: the batch needs another round

  // blockIdx -> (batch * head, item); all items of one (batch, head) share blockIdx % 8, i.e. one XCD (speed only)
  const int bm = (int)(blockIdx.x % 8) + 8 * (int)(blockIdx.x / (8 * p.n_items));
  if (bm >= B * M) return;
  const int it = (int)((blockIdx.x / 8) % p.n_items);
  const int b = bm / M, m = bm - b * M;
  int oi = 0;
  while (oi < 3 && it >= p.first_item[oi + 1]) ++oi;
  const int l = p.order[oi];
  const int local = it - p.first_item[oi];
  const int n_chunks = p.n_chunks[l];
  const int chunk = local % n_chunks, tile = local / n_chunks;
  const int ty = tile / p.n_tx[l], tx = tile - ty * p.n_tx[l];
  const RowAxis ay = p.ax[p.axis0[l] + ty], ax = p.ax[p.axis0[l] + p.n_ty[l] + tx];
  const int H = p.H[l], W = p.W[l];
  const int y0 = ay.r0, x0 = ax.r0, th = ay.rn, tw = ax.rn, n_rows = th * tw;
  // bucket slots per row and round (>= 15); rows are cap + 1 entries apart: with a power-of-two stride every row's bucket
  // would start on the same LDS bank (counters: 58 % of the LDS cycles were bank conflicts)
  const int cap = kRowBucketEntries / n_rows - 1, bstride = cap + 1;
  int n_cand = 0;
#pragma unroll
  for (int lq = 0; lq < 4; ++lq) n_cand += (int)ay.qn[lq] * (int)ax.qn[lq];
  const int c_begin = (int)((long long)n_cand * chunk / n_chunks), c_end = (int)((long long)n_cand * (chunk + 1) / n_chunks);
  const RowCandidate *cands = table + p.cand_base[l] + (long long)tile * p.cand_stride[l];

  const int tid = threadIdx.x;
  const int slot0 = tid >> 2, pt = tid & 3;                // scan role: candidate slots slot0, slot0 + 128 of the batch; point
  const int r = tid >> 1, half = tid & 1;                  // gather role: row of the tile, half of its 32 channels
  if (tid < kRowTileRows) count[tid] = 0;
  if (tid == 0) overflow = 0;
  // this lane's 16 channel sums as 8 packed pairs: a hit costs 8 v_pk_fma_f32 instead of 16 v_fmac_f32 (the bucket walk is
  // bound by instruction issue and LDS latency, not by arithmetic)
  rows_v2f ap[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) ap[i] = (rows_v2f){0.f, 0.f};

  // The scan is a chain of two dependent memory round trips per batch (candidate -> point).  Candidate entries are read two
  // batches ahead; a batch's points are requested as soon as the previous batch's inputs have been turned into taps, so the
  // loads are in flight during that batch's bucket / gather phases (lds_barrier keeps them in flight).
  // Batch k takes candidates k, k + n_batches, k + 2 n_batches, ... of the chunk: a batch then samples the whole scan
  // region.  Neighbouring queries sample alike (the model's offsets are a per-head pattern), so a batch of NEIGHBOURS
  // lands on a handful of rows of a coarse level -- hundreds of hits per row against a bucket of a few dozen (many overflow
  // rounds with most lanes idle: measured 1.95 ms per backward in the train step against 1.45 ms on random offsets).
  struct Inputs { float4 g0, g1, lg; float2 xy; float wt; RefScale rs; };
  const int n_batches = (c_end - c_begin + kRowBatchQueries - 1) / kRowBatchQueries;
  auto cand_index = [&](const int k, const int u) {
    return k < n_batches ? c_begin + k + (slot0 + u * (kRowThreads / 4)) * n_batches : c_end;
  };
  auto candidate = [&](const int j) {
    RowCandidate c{0, 0, 0};
    if (j < c_end) c = cands[j];
    return c;
  };
  auto fetch = [&](const int j, const RowCandidate c) {
    Inputs in{};
    if (j < c_end) {
      const long long q_lin = (long long)b * S + c.token;                          // Lq == S
      const float *gp = grad_out + (q_lin * M + m) * 32 + pt * 8;
      in.g0 = ld4(gp);
      in.g1 = ld4(gp + 4);
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
  Inputs in[kRowSub];
#pragma unroll
  for (int u = 0; u < kRowSub; ++u) {
    cur[u] = candidate(cand_index(0, u));
    in[u] = fetch(cand_index(0, u), cur[u]);
    c_next[u] = candidate(cand_index(1, u));
  }
  lds_barrier();
#if MSDA_ROWS_STAMP
  long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = clock64();
#endif

  for (int k = 0; k < n_batches; ++k) {
    ROWS_STAMP(0);
#if MSDA_ROWS_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ROWS_STAMP(7);
#endif
    // ---- 1. this thread's two points -------------------------------------------------------------------------------
    unsigned pend = 0;                                     // bit 4 u + c: corner c of sub-slot u still has to be placed
    int rows[kRowSub][4];
    float coef[kRowSub][4];
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
#pragma unroll
      for (int c = 0; c < 4; ++c) { rows[u][c] = 0; coef[u][c] = 0.f; }
      if (cand_index(k, u) < c_end) {
        go_lds[slot * 8 + ((pt * 2) ^ (slot & 7))] = in[u].g0;          // chunk c of slot s at c ^ (s & 7): the gather's lanes
        go_lds[slot * 8 + ((pt * 2 + 1) ^ (slot & 7))] = in[u].g1;      // read one chunk of different rows -> different banks
        const Tap<float> tp = make_tap<float>(lx, ly, H, W);
        if (tp.valid && near_point(tp.h_low, tp.w_low, cur[u].cy, cur[u].cx, p.reach)) {
          const int ry0 = tp.y0 - y0, ry1 = tp.y1 - y0, rx0 = tp.x0 - x0, rx1 = tp.x1 - x0;
          const bool iy0 = tp.t && (unsigned)ry0 < (unsigned)th, iy1 = tp.b && (unsigned)ry1 < (unsigned)th;
          const bool ix0 = tp.l && (unsigned)rx0 < (unsigned)tw, ix1 = tp.r && (unsigned)rx1 < (unsigned)tw;
          rows[u][0] = ry0 * tw + rx0; rows[u][1] = ry0 * tw + rx1; rows[u][2] = ry1 * tw + rx0; rows[u][3] = ry1 * tw + rx1;
          coef[u][0] = tp.w1 * wt; coef[u][1] = tp.w2 * wt; coef[u][2] = tp.w3 * wt; coef[u][3] = tp.w4 * wt;
          pend |= ((iy0 && ix0 ? 1u : 0u) | (iy0 && ix1 ? 2u : 0u) | (iy1 && ix0 ? 4u : 0u) | (iy1 && ix1 ? 8u : 0u)) << (4 * u);
        }
      }
      // the next batch's points: requested now, consumed in the next iteration
      cur[u] = c_next[u];
      in[u] = fetch(cand_index(k + 1, u), cur[u]);
      c_next[u] = candidate(cand_index(k + 2, u));
    }
    ROWS_STAMP(1);
    // ---- 2./3. buckets and row sums; a row whose bucket overflows (many points on one pixel) takes more rounds -----------
    bool again;
    do {
      // all slot reservations first (independent LDS atomics with return: in flight together), then the entries -- one atomic,
      // its wait and its store per corner in turn cost eight LDS round trips per thread and batch
      unsigned rank[kRowSub][4];
#pragma unroll
      for (int u = 0; u < kRowSub; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          rank[u][c] = 0xFFFFFFFFu;
          if (pend & (1u << (4 * u + c))) rank[u][c] = atomicAdd(&count[rows[u][c]], 1u);
        }
#pragma unroll
      for (int u = 0; u < kRowSub; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (rank[u][c] < (unsigned)cap) {
            bucket[rows[u][c] * bstride + rank[u][c]] = make_uint2(__float_as_uint(coef[u][c]), (unsigned)(slot0 + u * (kRowThreads / 4)));
            pend &= ~(1u << (4 * u + c));
          }
      if (pend) overflow = 1;                // rare
      ROWS_STAMP(2);
      lds_barrier();
      ROWS_STAMP(3);
      // read BETWEEN the round's two barriers: every `overflow = 1` of this round lies before the first one, and the next
      // write to the flag (the reset below, or the next batch's appends) lies behind the second -- read after it, a fast wave's
      // next-batch append could reach a slow wave's read of this batch (waves disagreeing on the number of barriers)
      again = overflow != 0;
      if (r < n_rows) {
        const int n = min((int)count[r], cap);
        const uint2 *bk = bucket + r * bstride;
        for (int i = 0; i < n; ++i) {
          const uint2 e = bk[i];
          const float w = __uint_as_float(e.x);
          const float4 *g4 = go_lds + e.y * 8;
          const int sw = (half * 4) ^ (e.y & 7);     // grad_out rows are stored with their 16-byte chunks XOR-swizzled by the slot
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const float4 a = g4[kk ^ sw];
            ap[2 * kk] = __builtin_elementwise_fma((rows_v2f){w, w}, (rows_v2f){a.x, a.y}, ap[2 * kk]);
            ap[2 * kk + 1] = __builtin_elementwise_fma((rows_v2f){w, w}, (rows_v2f){a.z, a.w}, ap[2 * kk + 1]);
          }
        }
        if (half == 0) count[r] = 0;         // the row's two lanes sit in one wave: both have read it
      }
      ROWS_STAMP(4);
      lds_barrier();                         // orders the gather before the next appends / grad_out rows
      ROWS_STAMP(5);
      if (again) {                           // everyone has read the flag (in front of the barrier above)
        if (tid == 0) overflow = 0;
        lds_barrier();
      }
    } while (again);
  }

#if MSDA_ROWS_STAMP
  ROWS_STAMP(6);
  if ((threadIdx.x & 63) == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&g_rows_stamp[i], (unsigned long long)st_acc[i]);
#endif
  // ---- write the tile -------------------------------------------------------------------------------------------------
  float4 acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) acc[k] = make_float4(ap[2 * k].x, ap[2 * k].y, ap[2 * k + 1].x, ap[2 * k + 1].y);
  const long long tok0 = (long long)b * S + p.start[l];
  if (n_chunks == 1) {
    if (r < n_rows) {
      const int ry = r / tw, rx = r - ry * tw;
      const long long token = tok0 + (long long)(y0 + ry) * W + (x0 + rx);
      float *dst = grad_value + (token * M + m) * 32 + half * 16;
      const bool padded = vmask && vmask[token];
#pragma unroll
      for (int k = 0; k < 4; ++k) st4(dst + 4 * k, padded ? make_float4(0.f, 0.f, 0.f, 0.f) : acc[k]);
    }
  } else {
    // several workgroups share the tile: full 128-byte rows of atomics (lane = channel), through LDS
    float *rows_lds = reinterpret_cast<float *>(bucket);
    if (r < n_rows) {
#pragma unroll
      for (int k = 0; k < 4; ++k) *reinterpret_cast<float4 *>(rows_lds + r * 32 + half * 16 + 4 * k) = acc[k];
    }
    __syncthreads();
    const int ch = tid & 31;
    for (int rr = tid >> 5; rr < n_rows; rr += kRowThreads / 32) {
      const float v = rows_lds[rr * 32 + ch];
      const int ry = rr / tw, rx = rr - ry * tw;
      const long long token = tok0 + (long long)(y0 + ry) * W + (x0 + rx);
      if (v != 0.f && !(vmask && vmask[token])) atomicAdd(grad_value + (token * M + m) * 32 + ch, v);
    }
  }
}

}  // namespace msda
