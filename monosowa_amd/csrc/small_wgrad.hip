// Weight AND bias gradient of a linear layer over a few thousand tokens, two launches (round 5):
//
//     dW[M, N] = dY[R, M]^T . X[R, N]          db[M] = sum_r dY[r, m]            R >= 64 rows (the step: 8,800 and 30,720), M, N multiples of 64
//
// (autograd's AddmmBackward / MmBackward of the nn.Linear layers of the reference's decoder and depth-token encoder,
// depthaware_transformer.py:339-354,440-515: 8,800 rows of queries, 30,720 of depth tokens.)  The product's output is tiny and its
// contraction long: the library runs [256, 8800] x [8800, 256] as 138 workgroups + a split-K reduction in 25.7 us, the column sums of dY
// take two more launches (7.9 + 5.8 us) and read dY a second time.  Here:
//
//   linear_wgrad_partial_kernel   workgroup = (64 x 64 tile of dW, split s of the rows), four waves, each a 32 x 32 sub-tile, two rows per
//        v_mfma_f32_32x32x2_f32 (exact f32 products): lane (c = lane % 32, h = lane / 32) feeds dY[k + h][m0 + c] as the A operand and
//        X[k + h][n0 + c] as the B operand.  The two 16-row x 64-column slabs of a stage go global -> registers (16 bytes per lane, four
//        stages in flight) -> LDS ring -> operand registers, all issued by the MFMA waves in the gaps between their MFMAs; one barrier per
//        stage, in the middle of the MFMA run.  The A operand IS dY, so the waves of the first column of tiles add it up on the way: the
//        bias gradient costs one v_add per MFMA and no second pass.  Splits are the FAST index of the grid and a multiple of 8: a split's
//        rows are read by one XCD only (blockIdx % 8), each L2 sees R / 8 rows of both matrices; every split is a whole number of stages.
//   linear_wgrad_reduce_kernel    sums the S partial [M N + M] images in split order -- no atomics, the result is deterministic.
//        (Folded into the first launch -- every workgroup releases its image with an agent-scope fence and counts itself in, the last one
//        of a tile adds the tile's images -- the pair took 50.6 us instead of 21.8 and the step +1.9 ms: the fence is an L2 write-back
//        per wave.  Bit-identical, measured, removed.)
//
// Measured, [8800, 256] x [8800, 256] stand-alone (tools/debug/r05_wgrad_prof.sh, in-kernel s_memtime stamps since removed, tools/ubench/mfma_f32_chain.hip): 17.3 us +
// 4.4 us for the reduction, against the library's 18.6 (25.7 inside the step) + 7.5 + 5.1 for the column sums.  The matrix pipe's floor is
// 275 MFMAs x 64 cycles = 8.4 us per wave; the loop runs at 104 cycles per MFMA (93 without its global loads; the bare schedule -- MFMAs,
// operand reads, barrier, LDS writes behind MFMA 0 / 1 -- does 78 in the microbenchmark).  Every other arrangement tried landed on the same
// 17 - 18 us: each wave streaming its own 32 columns straight into registers (32 dword loads per stage: issue-bound) or through a private
// LDS-DMA ring (512 bytes per MFMA: at the CU's ~34 bytes / clock intake); shared slabs by LDS-DMA issued from the MFMA waves (an LDS-DMA
// piece costs the issuing wave 100 - 160 cycles and slows the others' LDS reads); separate loader waves, by LDS-DMA or through registers (a
// loader that shares a SIMD with an MFMA wave only gets to run while that wave waits at the barrier: the MFMA waves spent 370 of 890
// cycles per stage there); 64 x 32 per wave straight from global memory with 8-byte loads and no LDS (18.6, and twice the partial images).
// The column sums' one v_add_f32 per MFMA costs 1.9 of the 17.4 us (15.6 with the adds compiled out); written as packed adds the compiler splits them again.
#pragma once
#include <hip/hip_runtime.h>

namespace mono {

typedef float wg_f32x16 __attribute__((ext_vector_type(16)));
#ifndef MONO_WGRAD_WORKGROUPS
#define MONO_WGRAD_WORKGROUPS 256
#endif
constexpr int kWgPairs = 8;                        // row pairs (MFMAs per wave) per stage
constexpr int kWgStageRows = 2 * kWgPairs;         // 16 rows of dY and X per stage: an 8 KiB slab, 2 KiB (two float4 per lane) per wave
constexpr int kWgSlots = 3;                        // LDS ring: a stage being read, one published, one being written

// number of row splits for an [R] x [M, N] problem (multiple of 8; about MONO_WGRAD_WORKGROUPS workgroups, at least 64 rows each)
inline int linear_wgrad_splits(int R, int M, int N) {
  const int tiles = (M / 64) * (N / 64);
  int S = ((MONO_WGRAD_WORKGROUPS + tiles - 1) / tiles + 7) / 8 * 8;
  while (S > 8 && R / S < 64) S -= 8;
  return S;
}

__global__ __launch_bounds__(256) void linear_wgrad_partial_kernel(const float *__restrict__ dy, long long ldy,
                                                                   const float *__restrict__ x, long long ldx, float *__restrict__ ws,
                                                                   int R, int M, int N, int S, int rows_per_split) {
  // the workgroup's operand slabs: [slot][dY | X][piece = 4 rows][row pair u][32-column half][row of the pair h][32 columns] -- the
  // order in which an MFMA's 64 lanes (h, c) read one half of one row pair: 64 consecutive floats, no bank conflict
  __shared__ float ring[kWgSlots][2][kWgStageRows * 64];
  const int s = blockIdx.x % S, tile = blockIdx.x / S;
  const int n_tiles = N / 64, tm = tile / n_tiles, tn = tile % n_tiles;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), c = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = tm * 64 + wm * 32, n0 = tn * 64 + wn * 32;
  const int k_begin = s * rows_per_split, k_end = min(R, k_begin + rows_per_split);
  const int n_stages = k_end > k_begin ? (k_end - k_begin) / kWgStageRows : 0;
  wg_f32x16 acc0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  float bsum = 0.f;
  // staging: wave w brings pieces 2 w, 2 w + 1 of the 8 (4 of dY, then 4 of X); lane l holds floats 4 l .. 4 l + 3 of a piece's image
  //   = (u = l / 32, half = (l / 16) % 2, h = (l / 8) % 2, column 4 (l % 8))  ->  row 2 u + h of the piece, column 32 half + 4 (l % 8)
  const int src_row = 4 * ((2 * wave) & 3) + 2 * (lane >> 5) + ((lane >> 3) & 1), src_col = 32 * ((lane >> 4) & 1) + 4 * (lane & 7);
  const float *gsrc = wave < 2 ? dy + (long long)(k_begin + src_row) * ldy + tm * 64 + src_col
                               : x + (long long)(k_begin + src_row) * ldx + tn * 64 + src_col;
  const long long g_ld = wave < 2 ? ldy : ldx;
  float *const lds_dst = &ring[0][wave >> 1][((2 * wave) & 3) * 256 + 4 * lane];
  constexpr int kSlotFloats = 2 * kWgStageRows * 64;
  const float *const al = &ring[0][0][wm * 64 + lane], *const bl = &ring[0][1][wn * 64 + lane];
  float4 g0a, g0b, g1a, g1b, g2a, g2b, g3a, g3b;                     // four stages in flight in registers (1 us of loads ahead of their use)
  float a0[kWgPairs], b0[kWgPairs], a1[kWgPairs], b1[kWgPairs];       // two operand sets, used alternately
  // (past the last stage the last one is loaded and written again -- into a slot nobody reads any more: the steady-state loop has no
  // branches, so the compiler's wait counts stay exact: `vmcnt(2)` in front of a stage's LDS writes leaves the next stage's loads in flight)
#define WG_GLOAD(GA, GB, STAGE)                                                                                      \
  {                                                                                                                  \
    const float *p_ = gsrc + (long long)min((STAGE), n_stages - 1) * kWgStageRows * g_ld;              \
    GA = *reinterpret_cast<const float4 *>(p_); GB = *reinterpret_cast<const float4 *>(p_ + 4 * g_ld);               \
  }
#define WG_LWRITE(GA, GB, STAGE)                                                                                     \
  {                                                                                                                  \
    float *d_ = lds_dst + ((STAGE) % kWgSlots) * kSlotFloats;                                                        \
    *reinterpret_cast<float4 *>(d_) = GA; *reinterpret_cast<float4 *>(d_ + 256) = GB;                                \
  }
  // One iteration = one stage i (its operands are in registers AC / BC):
  //   MFMA 0 .. 3, stage i + 2: registers -> LDS behind MFMA 0 and 1 | lgkmcnt(0), barrier i | stage i + 6: global -> registers |
  //   MFMA 4 .. 7, each followed by its share of stage i + 1's operand reads (LDS -> AN / BN)
  // The barrier sits in the middle of the MFMA run, in the shadow of an executing MFMA; everything else rides in the gaps between MFMAs,
  // issued by the MFMA waves themselves (see the header for what other arrangements cost).  Stage j is in LDS from barrier j - 2 on and
  // read after barrier j - 1; its slot is rewritten (stage j + 3) after barrier j: every wave's reads of it were retired before that.
#define WG_STAGE(AC, BC, AN, BN, GA, GB, I)                                                                          \
  {                                                                                                                  \
    float *d_ = lds_dst + (((I) + 2) % kWgSlots) * kSlotFloats;                                                      \
    _Pragma("unroll") for (int u = 0; u < kWgPairs / 2; ++u) {                                                       \
      if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(AC[u], BC[u], acc1, 0, 0, 0);                           \
      else acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(AC[u], BC[u], acc0, 0, 0, 0);                                 \
      bsum += AC[u];                                                                                                 \
      if (u < 2) {                          /* the LDS writes right behind MFMA 0 and 1: done by the time of the barrier */ \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        *reinterpret_cast<float4 *>(d_ + 256 * u) = u ? GB : GA;                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
      }                                                                                                              \
    }                                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                               \
    __builtin_amdgcn_s_barrier();                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    WG_GLOAD(GA, GB, (I) + 6)                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    const float *an_ = al + (((I) + 1) % kWgSlots) * kSlotFloats, *bn_ = bl + (((I) + 1) % kWgSlots) * kSlotFloats;  \
    _Pragma("unroll") for (int u = kWgPairs / 2; u < kWgPairs; ++u) {                                                \
      if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(AC[u], BC[u], acc1, 0, 0, 0);                           \
      else acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(AC[u], BC[u], acc0, 0, 0, 0);                                 \
      bsum += AC[u];                                                                                                 \
      const int v = (u - kWgPairs / 2) * 2;                                                                          \
      AN[v] = an_[v * 128]; BN[v] = bn_[v * 128]; AN[v + 1] = an_[v * 128 + 128]; BN[v + 1] = bn_[v * 128 + 128];    \
      __builtin_amdgcn_sched_barrier(0);                                                                             \
    }                                                                                                                \
  }
  if (n_stages > 0) {
  // prologue: stages 0, 1 -> LDS, stages 2 .. 5 -> registers; stage 0's operands -> a0 / b0
  // (the four register stages are loaded oldest first, as in the loop: the wait counts there are the minimum over both ways in)
  WG_GLOAD(g0a, g0b, 0)
  WG_GLOAD(g1a, g1b, 1)
  WG_LWRITE(g0a, g0b, 0)
  WG_LWRITE(g1a, g1b, 1)
  WG_GLOAD(g0a, g0b, 2)
  WG_GLOAD(g1a, g1b, 3)
  WG_GLOAD(g2a, g2b, 4)
  WG_GLOAD(g3a, g3b, 5)
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kWgPairs; ++u) { a0[u] = al[u * 128]; b0[u] = bl[u * 128]; }
  int i = 0;
#pragma unroll 1
  for (; i + 4 <= n_stages; i += 4) {
    WG_STAGE(a0, b0, a1, b1, g0a, g0b, i)
    WG_STAGE(a1, b1, a0, b0, g1a, g1b, i + 1)
    WG_STAGE(a0, b0, a1, b1, g2a, g2b, i + 2)
    WG_STAGE(a1, b1, a0, b0, g3a, g3b, i + 3)
  }
  if (i < n_stages) WG_STAGE(a0, b0, a1, b1, g0a, g0b, i)
  if (i + 1 < n_stages) WG_STAGE(a1, b1, a0, b0, g1a, g1b, i + 1)
  if (i + 2 < n_stages) WG_STAGE(a0, b0, a1, b1, g2a, g2b, i + 2)
  }
#undef WG_STAGE
#undef WG_GLOAD
#undef WG_LWRITE
  // the matrix's last rows (fewer than a stage, possibly an odd count): straight from global memory, all loads first
  {
    const int k_tail = k_begin + n_stages * kWgStageRows;
    if (k_tail < k_end) {
      const float *ap = dy + (long long)(k_tail + h) * ldy + m0 + c, *bp = x + (long long)(k_tail + h) * ldx + n0 + c;
      float ta[kWgPairs], tb[kWgPairs];
#pragma unroll
      for (int u = 0; u < kWgPairs; ++u) {
        const bool live = k_tail + 2 * u + h < k_end;
        ta[u] = live ? ap[2 * u * ldy] : 0.f; tb[u] = live ? bp[2 * u * ldx] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < kWgPairs; ++u) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[u], tb[u], acc0, 0, 0, 0);
        bsum += ta[u];
      }
    }
  }
  // acc[4 g + t] = D[m0 + 8 g + 4 h + t][n0 + c]
  float *wp = ws + (long long)s * ((long long)M * N + M);
#pragma unroll
  for (int v = 0; v < 16; ++v)
    wp[(long long)(m0 + 8 * (v >> 2) + 4 * h + (v & 3)) * N + n0 + c] = acc0[v] + acc1[v];
  if (tn == 0 && wn == 0) {
    const float t = bsum + __shfl_xor(bsum, 32);
    if (h == 0) wp[(long long)M * N + m0 + c] = t;
  }
}

// dw[i] (i < MN) / db[i - MN] = sum_s ws[s][i]; one float4 per thread, splits in order
__global__ __launch_bounds__(256) void linear_wgrad_reduce_kernel(const float *__restrict__ ws, float *__restrict__ dw, float *__restrict__ db,
                                                                  long long MN, int M, int S) {
  const long long i4 = (long long)blockIdx.x * 256 + threadIdx.x, stride4 = (MN + M) / 4;
  if (i4 >= stride4) return;
  const float4 *p = reinterpret_cast<const float4 *>(ws) + i4;
  float4 a = p[0];
#pragma unroll 4
  for (int s = 1; s < S; ++s) { const float4 b = p[s * stride4]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
  if (i4 * 4 < MN) reinterpret_cast<float4 *>(dw)[i4] = a;
  else if (db) reinterpret_cast<float4 *>(db)[i4 - MN / 4] = a;
}

}  // namespace mono
