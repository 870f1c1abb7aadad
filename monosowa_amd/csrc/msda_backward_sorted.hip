// Backward K2, third generation: "sort, then sum in registers" -- no floating-point atomics on the fine levels
// and no LDS atomics per contribution.
//
// The tile-owner kernel of msda_backward_tiled.hip adds every contribution (4 corners x 32 channels per sampling
// point) into LDS with ds_add_u64: 128 lane-adds per point at 7.1 lane-adds/clk/CU = 18 clk/CU/point, the whole
// cost of that kernel (1.04 ms per encoder layer at B = 16).  Here a workgroup still owns a tile (<= 16 x 16 value
// rows of one level of one (batch, head)) and scans the same K1 record lists through the same chunk boxes, but
//   1. the points that touch the tile are binned by bilinear CELL (the pixel pair (floor y, floor x); a tile has
//      (th + 1) x (tw + 1) cells) with one ds_add_rtn_u32 per point -- a counting sort in LDS;
//   2. an 8-lane group (8 x float4 = the head's 32 channels) takes a cell, walks its points, and accumulates the
//      four corner rows in REGISTERS: 16 FMAs per lane and point, grad_out rows prefetched four points ahead;
//   3. the four sums are added to the f32 tile in LDS with plain read-modify-writes: cells are processed in four
//      parity classes (cy % 2, cx % 2), and two cells of one class never share a corner row.
// Points are buffered in batches of <= kSortCap per workgroup; tiles are written back exactly as before (plain
// stores when one workgroup owns the tile, row atomics into a zeroed region for the coarse levels that are split
// by query range).  Sums are plain f32 in a data-dependent order (like the reference's atomics, cuh:125-152).
// (included by msda_capi.hip after msda_backward_tiled.hip: ChunkBox, kInvalidCoord, BwdPlan come from there)

namespace msda {

constexpr int kSortCap = 1536;            // buffered points per batch
constexpr int kSortThreads = 512;
constexpr int kSortWaves = kSortThreads / 64;
constexpr int kSortRowPad = 36;           // floats per LDS tile row: 8 groups x ds_read_b128 spread over the banks
constexpr int kMaxCells = 17 * 17;

__global__ __launch_bounds__(kSortThreads, 2) void bwd_scatter_sorted_kernel(
    const float2 *__restrict__ rec_hw, const float *__restrict__ rec_aw, const ChunkBox *__restrict__ boxes,
    const float *__restrict__ grad_out, float *__restrict__ grad_value, const BwdPlan plan, int B, int S, int M,
    int Lq, int P, int n_chunks_per_list, const unsigned char *__restrict__ vmask = nullptr) {
  __shared__ float tile[kTileRows * kSortRowPad];     // 36 KB
  __shared__ float4 r_wts[kSortCap];                  // corner weights x attn_w
  __shared__ int r_q[kSortCap];                       // query of the point
  __shared__ unsigned r_key[kSortCap];                // cell << 16 | rank inside the cell
  __shared__ unsigned short order[kSortCap];          // sorted position -> record
  __shared__ unsigned hist[kMaxCells + 3], cstart[kMaxCells + 3];
  __shared__ unsigned n_rec;

  const int BM = B * M;
  const int bm = (int)(blockIdx.x % 8) + 8 * (int)(blockIdx.x / (8 * plan.n_items));
  const int it = (int)((blockIdx.x / 8) % plan.n_items);
  if (bm >= BM) return;
  const int b = bm / M, m = bm % M;
  int oi = 0;
  while (oi + 1 < plan.n_levels && it >= plan.first_item[oi + 1]) ++oi;
  const int l = plan.order[oi];
  const int local = it - plan.first_item[oi];
  const int n_chunks = plan.n_chunks[l];
  const int chunk = local % n_chunks, tile_id = local / n_chunks;
  const int H = plan.H[l], W = plan.W[l];
  const int y0 = (tile_id / plan.n_tx[l]) * plan.th[l], x0 = (tile_id % plan.n_tx[l]) * plan.tw[l];
  const int th = min(plan.th[l], H - y0), tw = min(plan.tw[l], W - x0);
  const int n_rows = th * tw;
  const int cw = tw + 1, n_cells = (th + 1) * cw;
  const long long n_pts = (long long)Lq * P;
  const int pt_begin = (int)(n_pts * chunk / n_chunks), pt_end = (int)(n_pts * (chunk + 1) / n_chunks);
  const bool exclusive = n_chunks == 1;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gid = threadIdx.x >> 3, sub = threadIdx.x & 7;
  for (int i = threadIdx.x; i < n_rows * kSortRowPad; i += kSortThreads) tile[i] = 0.f;

  const long long list_id = (long long)bm * plan.n_levels + l;
  const float2 *hw_list = rec_hw + list_id * n_pts;
  const float *aw_list = rec_aw + list_id * n_pts;
  const ChunkBox *box_list = boxes + list_id * n_chunks_per_list;
  const float *go_base = grad_out + ((long long)b * Lq * M + m) * 32 + sub * 4;     // + q * M*32

  // scan state of this wave (survives the batches): chunks are dealt round-robin to the waves, 64 boxes per look
  const int first_chunk = pt_begin >> 6, last_chunk = (pt_end + 63) >> 6;
  int cbase = first_chunk + wave, cur_base = 0;
  unsigned long long todo = 0ull;
  bool wave_done = false;
  float2 hw_next = make_float2(kInvalidCoord, kInvalidCoord);
  float aw_next = 0.f;
  auto fetch = [&](unsigned long long bits) {
    hw_next = make_float2(kInvalidCoord, kInvalidCoord);
    aw_next = 0.f;
    if (bits) {
      const int idx = (cur_base + __builtin_ctzll(bits) * kSortWaves) * 64 + lane;
      if (idx >= pt_begin && idx < pt_end) { hw_next = hw_list[idx]; aw_next = aw_list[idx]; }
    }
  };

  for (;;) {
    for (int i = threadIdx.x; i < n_cells; i += kSortThreads) hist[i] = 0u;
    if (threadIdx.x == 0) n_rec = 0u;
    __syncthreads();

    // ---- 1. scan + bin -----------------------------------------------------------------------------------
    while (!wave_done) {
      if (!todo) {
        if (cbase >= last_chunk) { wave_done = true; break; }
        const int my_chunk = cbase + lane * kSortWaves;
        bool touch = false;
        if (my_chunk < last_chunk) {
          const ChunkBox bx = box_list[my_chunk];
          touch = bx.y_lo <= bx.y_hi && bx.y_hi >= y0 && bx.y_lo < y0 + th && bx.x_hi >= x0 && bx.x_lo < x0 + tw;
        }
        todo = __ballot(touch);
        cur_base = cbase;
        cbase += kSortWaves * 64;
        if (!todo) continue;
        fetch(todo);
      }
      // up to 8 waves pass this test together and add <= 64 records each
      if (*reinterpret_cast<volatile unsigned *>(&n_rec) > (unsigned)(kSortCap - kSortWaves * 64)) break;
      const int idx = (cur_base + __builtin_ctzll(todo) * kSortWaves) * 64 + lane;
      todo &= todo - 1;
      const float2 hw = hw_next;
      const float aw = aw_next;
      fetch(todo);
      bool hit = false;
      float4 wts = make_float4(0.f, 0.f, 0.f, 0.f);
      int cell = 0;
      if (hw.x > kInvalidCoord) {
        const Tap<float> tp = make_tap_im(hw.x, hw.y, H, W);
        const int cy = (int)floorf(hw.x) - y0 + 1, cx = (int)floorf(hw.y) - x0 + 1;      // cell: corners (cy-1..cy, cx-1..cx)
        const bool iy0 = tp.t && (unsigned)(cy - 1) < (unsigned)th, iy1 = tp.b && (unsigned)cy < (unsigned)th;
        const bool ix0 = tp.l && (unsigned)(cx - 1) < (unsigned)tw, ix1 = tp.r && (unsigned)cx < (unsigned)tw;
        hit = (iy0 || iy1) && (ix0 || ix1);
        cell = cy * cw + cx;
        wts = make_float4(tp.w1 * aw, tp.w2 * aw, tp.w3 * aw, tp.w4 * aw);
      }
      const unsigned long long mask = __ballot(hit);
      if (mask) {
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(&n_rec, (unsigned)__popcll(mask));
        base = __shfl(base, 0);
        if (hit) {
          const unsigned slot = base + __popcll(mask & ((1ull << lane) - 1ull));
          r_wts[slot] = wts;
          r_q[slot] = idx / P;
          r_key[slot] = ((unsigned)cell << 16) | atomicAdd(&hist[cell], 1u);
        }
      }
    }
    const int more = __syncthreads_or(wave_done ? 0 : 1);
    const unsigned total = n_rec;

    // ---- 2. cell offsets, sorted order ------------------------------------------------------------------
    if (wave == 0) {
      unsigned carry = 0;
      for (int base = 0; base < n_cells; base += 64) {
        const int i = base + lane;
        const unsigned v = i < n_cells ? hist[i] : 0u;
        unsigned incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const unsigned t = __shfl_up(incl, o);
          if (lane >= o) incl += t;
        }
        if (i < n_cells) cstart[i] = carry + incl - v;
        carry += __shfl(incl, 63);
      }
    }
    __syncthreads();
    for (unsigned i = threadIdx.x; i < total; i += kSortThreads) {
      const unsigned key = r_key[i];
      order[cstart[key >> 16] + (key & 0xFFFFu)] = (unsigned short)i;
    }
    __syncthreads();

    // ---- 3. per-cell register sums, four parity classes -------------------------------------------------
    if (total) {
      for (int ph = 0; ph < 4; ++ph) {
        const int py = ph >> 1, px = ph & 1;
        const int ncy = (th + 2 - py) >> 1, ncx = (cw + 1 - px) >> 1;     // cells cy = py, py+2, .. <= th; cx likewise < cw
        for (int k = gid; k < ncy * ncx; k += kSortThreads / 8) {
          const int cy = py + 2 * (k / ncx), cx = px + 2 * (k % ncx);
          const int cell = cy * cw + cx;
          const unsigned n = hist[cell];
          if (!n) continue;
          const unsigned s = cstart[cell];
          float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f), a2 = a1, a3 = a1, a4 = a1;
          for (unsigned k0 = 0; k0 < n; k0 += 4) {
            float4 w[4], g[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const bool on = k0 + j < n;
              const unsigned i = on ? order[s + k0 + j] : 0u;
              w[j] = on ? r_wts[i] : make_float4(0.f, 0.f, 0.f, 0.f);
              g[j] = on ? ld4(go_base + (long long)r_q[i] * M * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              a1.x += w[j].x * g[j].x; a1.y += w[j].x * g[j].y; a1.z += w[j].x * g[j].z; a1.w += w[j].x * g[j].w;
              a2.x += w[j].y * g[j].x; a2.y += w[j].y * g[j].y; a2.z += w[j].y * g[j].z; a2.w += w[j].y * g[j].w;
              a3.x += w[j].z * g[j].x; a3.y += w[j].z * g[j].y; a3.z += w[j].z * g[j].z; a3.w += w[j].z * g[j].w;
              a4.x += w[j].w * g[j].x; a4.y += w[j].w * g[j].y; a4.z += w[j].w * g[j].z; a4.w += w[j].w * g[j].w;
            }
          }
          auto add_row = [&](int ry, int rx, const float4 a) {
            if ((unsigned)ry >= (unsigned)th || (unsigned)rx >= (unsigned)tw) return;     // owned by a neighbour tile
            float4 *p = reinterpret_cast<float4 *>(tile + (ry * tw + rx) * kSortRowPad + sub * 4);
            float4 t = *p;
            t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w;
            *p = t;
          };
          add_row(cy - 1, cx - 1, a1);
          add_row(cy - 1, cx, a2);
          add_row(cy, cx - 1, a3);
          add_row(cy, cx, a4);
        }
        __syncthreads();
      }
    }
    if (!more) break;
  }

  // ---- write the tile back -------------------------------------------------------------------------------
  const long long tok0 = (long long)b * S + plan.start[l];
  if (exclusive) {
    for (int r = gid; r < n_rows; r += kSortThreads / 8) {
      const int ry = r / tw, rx = r - ry * tw;
      const long long token = tok0 + (long long)(y0 + ry) * W + (x0 + rx);
      st4(grad_value + (token * M + m) * 32 + sub * 4,
          (vmask && vmask[token]) ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4 *>(tile + r * kSortRowPad + sub * 4));
    }
  } else {
    const int ch = threadIdx.x & 31;
    for (int r = threadIdx.x >> 5; r < n_rows; r += kSortThreads / 32) {
      const float v = tile[r * kSortRowPad + ch];
      const int ry = r / tw, rx = r - ry * tw;
      const long long token = tok0 + (long long)(y0 + ry) * W + (x0 + rx);
      if (v != 0.f && !(vmask && vmask[token])) atomicAdd(grad_value + (token * M + m) * 32 + ch, v);
    }
  }
}

}  // namespace msda
