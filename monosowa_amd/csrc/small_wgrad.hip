// Weight AND bias gradient of a linear layer over a few thousand tokens, two launches (round 5):
//
//     dW[M, N] = dY[R, M]^T . X[R, N]          db[M] = sum_r dY[r, m]            R = 2,048 .. 32,767 rows, M, N multiples of 64
//
// (autograd's AddmmBackward / MmBackward of the nn.Linear layers of the reference's decoder and depth-token encoder,
// depthaware_transformer.py:339-354,440-515: 8,800 rows of queries, 30,720 of depth tokens.)  The product's output is tiny and its
// contraction long: the library runs [256, 8800] x [8800, 256] as 138 workgroups + a split-K reduction in 25.7 us, the column sums of dY
// take two more launches (7.9 + 5.8 us) and read dY a second time.  Here:
//
//   linear_wgrad_partial_kernel   workgroup = (64 x 64 tile of dW, split s of the rows).  A wave owns a 32 x 32 sub-tile and walks its
//        split two rows per v_mfma_f32_32x32x2_f32 (exact f32 products): lane (c = lane % 32, h = lane / 32) feeds dY[k + h][m0 + c] as
//        the A operand and X[k + h][n0 + c] as the B operand -- both are 128-byte row segments, read straight from global memory, 8 row
//        pairs in flight ahead of the matrix pipe.  The A operand IS dY, so the waves of the first column of tiles add it up on the way:
//        the bias gradient costs one v_add per MFMA and no second pass.  Splits are the FAST index of the grid and a multiple of 8: a
//        split's rows are then read by one XCD only (blockIdx % 8), each L2 sees R / 8 rows of both matrices.
//   linear_wgrad_reduce_kernel    sums the S partial [M N + M] images in split order -- no atomics, the result is deterministic.
#pragma once
#include <hip/hip_runtime.h>

namespace mono {

typedef float wg_f32x16 __attribute__((ext_vector_type(16)));
#ifndef MONO_WGRAD_PAIRS
#define MONO_WGRAD_PAIRS 8
#endif
#ifndef MONO_WGRAD_SLOTS
#define MONO_WGRAD_SLOTS 3
#endif
#ifndef MONO_WGRAD_DEBUG
#define MONO_WGRAD_DEBUG 0
#endif
#ifndef MONO_WGRAD_WORKGROUPS
#define MONO_WGRAD_WORKGROUPS 512
#endif
constexpr int kWgPairs = MONO_WGRAD_PAIRS;            // row pairs per software-pipeline stage

// number of row splits for an [R] x [M, N] problem (multiple of 8; about 512 workgroups, at least 64 rows each)
inline int linear_wgrad_splits(int R, int M, int N) {
  const int tiles = (M / 64) * (N / 64);
  int S = ((MONO_WGRAD_WORKGROUPS + tiles - 1) / tiles + 7) / 8 * 8;
  while (S > 8 && R / S < 64) S -= 8;
  return S;
}

constexpr int kWgStageRows = 2 * kWgPairs;      // rows of dY and X per ring slot
constexpr int kWgSlots = MONO_WGRAD_SLOTS;       // ring depth: kWgSlots - 1 stages in flight beyond the one being read
constexpr int kWgPieces = kWgStageRows / 4;      // 1 KiB LDS-DMA pieces (4 rows x 64 columns) per operand and stage
constexpr int kWgPiecesPerWave = 2 * kWgPieces / 4;
static_assert(kWgStageRows % 8 == 0 && kWgSlots >= 2, "a wave issues whole pieces; one slot being read, the others landing");

#define WG_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define WG_LGKMCNT0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

__global__ __launch_bounds__(256) void linear_wgrad_partial_kernel(const float *__restrict__ dy, long long ldy,
                                                                   const float *__restrict__ x, long long ldx, float *__restrict__ ws,
                                                                   int R, int M, int N, int S, int rows_per_split) {
  // the workgroup's operand slabs, a ring of stages: [slot][dY | X][piece = 4 rows][row pair u][32-column half][row of the pair h][32 columns]
  // -- the order in which an MFMA's 64 lanes (h, c) read one half of one row pair: 64 consecutive floats, no bank conflict
  __shared__ float ring[kWgSlots][2][kWgPieces * 256];
  const int s = blockIdx.x % S, tile = blockIdx.x / S;
  const int n_tiles = N / 64, tm = tile / n_tiles, tn = tile % n_tiles;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), c = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = tm * 64 + wm * 32, n0 = tn * 64 + wn * 32;
  const int k_begin = s * rows_per_split, k_end = min(R, k_begin + rows_per_split);
  wg_f32x16 acc0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  float bsum = 0.f;
  const int n_full = k_end > k_begin ? (k_end - k_begin) / kWgStageRows : 0;
  // LDS-DMA: a piece lands lane-linear (lane l -> floats 4 l .. 4 l + 3 of the piece), so the SOURCE address carries the permutation:
  // float 4 l of the image = (u = l / 32, half = (l / 16) % 2, h = (l / 8) % 2, column 4 (l % 8)) -> row 2 u + h, column 32 half + 4 (l % 8)
  const int src_row = 2 * (lane >> 5) + ((lane >> 3) & 1), src_col = 32 * ((lane >> 4) & 1) + 4 * (lane & 7);
  // this wave's pieces of a stage: kWgPiecesPerWave consecutive ones of the 2 kWgPieces (dY's first, then X's)
  const float *src[kWgPiecesPerWave];
  long long src_step[kWgPiecesPerWave];
  int dst_off[kWgPiecesPerWave];                                   // float offset inside a slot
#pragma unroll
  for (int q = 0; q < kWgPiecesPerWave; ++q) {
    const int p = wave * kWgPiecesPerWave + q, is_x = p >= kWgPieces, pp = is_x ? p - kWgPieces : p;
    src[q] = is_x ? x + (long long)(k_begin + 4 * pp + src_row) * ldx + tn * 64 + src_col
                  : dy + (long long)(k_begin + 4 * pp + src_row) * ldy + tm * 64 + src_col;
    src_step[q] = (is_x ? ldx : ldy) * kWgStageRows;
    dst_off[q] = is_x * kWgPieces * 256 + pp * 256;
  }
  typedef __attribute__((address_space(3))) void lds_void;
  auto issue_piece = [&](int slot, int q) {
#if MONO_WGRAD_DEBUG != 2 && MONO_WGRAD_DEBUG != 5
    __builtin_amdgcn_global_load_lds(src[q], (lds_void *)(&ring[slot][0][0] + dst_off[q]), 16, 0, 0);
#endif
    src[q] += src_step[q];
  };
  float a_c[kWgPairs], b_c[kWgPairs], a_n[kWgPairs], b_n[kWgPairs];
  auto read_ops = [&](int slot, float *a, float *b) {
    const float *al = &ring[slot][0][wm * 64 + lane], *bl = &ring[slot][1][wn * 64 + lane];
#pragma unroll
    for (int u = 0; u < kWgPairs; ++u) { a[u] = al[u * 128]; b[u] = bl[u * 128]; }
  };
  auto mma = [&](int u) {
#if MONO_WGRAD_DEBUG != 1
    if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c[u], b_c[u], acc1, 0, 0, 0);
    else acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c[u], b_c[u], acc0, 0, 0, 0);
#endif
    bsum += a_c[u];
  };
  // Measured on the way here (tools/debug/r05_wgrad_prof.sh, [8800, 256] x [8800, 256]): with every wave streaming its OWN 32 columns of
  // both operands (512 bytes per MFMA, through registers or through a private LDS ring, any depth, any issue order) the kernel takes
  // 17.5 us -- its loads alone 8.7, its products alone 12: a CU takes in about 34 bytes per clock from L2 (MI355X_MICROARCH.md's 66 - 73
  // GB/s per CU) and four waves at 512 bytes per 64-cycle MFMA ask for 32.  Shared slabs halve that.
#if MONO_WGRAD_DEBUG == 3 || MONO_WGRAD_DEBUG == 5
  const long long t0 = clock64(), w0 = wall_clock64();
  long long t_lgkm = 0, t_vm = 0, t_bar = 0, t_rd = 0, t_body = 0;
#endif
  // prologue: stages 0 .. kWgSlots - 1 in flight (every slot); stage 0 landed everywhere -> registers
  for (int j = 0; j < kWgSlots && j < n_full; ++j)
#pragma unroll
    for (int q = 0; q < kWgPiecesPerWave; ++q) issue_piece(j, q);
  if (n_full >= kWgSlots) WG_VMCNT(kWgPiecesPerWave * (kWgSlots - 1)); else WG_VMCNT(0);
  __builtin_amdgcn_s_barrier();
  if (n_full > 0) read_ops(0, a_c, b_c);
  int i = 0, slot = 1 % kWgSlots, fill = 0;                         // slot = where stage i + 1 lands, fill = stage i's slot: where stage i + kWgSlots goes
  constexpr int kIssueEvery = kWgPairs / kWgPiecesPerWave;
  // steady state, iteration i: MFMAs of stage i from registers; operands of stage i + 1 LDS -> registers; this wave's pieces of stage
  // i + kWgSlots issued in the MFMAs' shadows into the slot stage i was read from -- by every wave in iteration i - 1, retired by the
  // lgkmcnt(0) in front of this iteration's barrier
#pragma unroll 1
  for (; i + kWgSlots < n_full; ++i) {
#if MONO_WGRAD_DEBUG == 3 || MONO_WGRAD_DEBUG == 5
    const long long s0 = clock64();
#endif
    WG_LGKMCNT0();
#if MONO_WGRAD_DEBUG == 3 || MONO_WGRAD_DEBUG == 5
    const long long s1 = clock64();
#endif
    WG_VMCNT(kWgPiecesPerWave * (kWgSlots - 2));                  // this wave's pieces of stage i + 1 have landed ...
#if MONO_WGRAD_DEBUG == 3 || MONO_WGRAD_DEBUG == 5
    const long long s2 = clock64();
#endif
    __builtin_amdgcn_s_barrier();                                  // ... and so have everybody else's
#if MONO_WGRAD_DEBUG == 3 || MONO_WGRAD_DEBUG == 5
    const long long s3 = clock64();
    t_lgkm += s1 - s0; t_vm += s2 - s1; t_bar += s3 - s2;
#endif
    read_ops(slot, a_n, b_n);
    __builtin_amdgcn_sched_barrier(0);
#if MONO_WGRAD_DEBUG == 3 || MONO_WGRAD_DEBUG == 5
    const long long s4 = clock64();
    t_rd += s4 - s3;
#endif
#pragma unroll
    for (int u = 0; u < kWgPairs; ++u) {
      mma(u);
      if (u % kIssueEvery == 0 && u / kIssueEvery < kWgPiecesPerWave) {
        __builtin_amdgcn_sched_barrier(0);
        issue_piece(fill, u / kIssueEvery);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#if MONO_WGRAD_DEBUG == 3 || MONO_WGRAD_DEBUG == 5
    __builtin_amdgcn_sched_barrier(0);
    t_body += clock64() - s4;
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int u = 0; u < kWgPairs; ++u) { a_c[u] = a_n[u]; b_c[u] = b_n[u]; }
    slot = slot + 1 == kWgSlots ? 0 : slot + 1;
    fill = fill + 1 == kWgSlots ? 0 : fill + 1;
  }
  // drain: everything issued has to land; stage i is in registers
  WG_LGKMCNT0();
  WG_VMCNT(0);
  __builtin_amdgcn_s_barrier();
#pragma unroll 1
  for (; i < n_full; ++i) {
    if (i + 1 < n_full) read_ops(slot, a_n, b_n);
#pragma unroll
    for (int u = 0; u < kWgPairs; ++u) mma(u);
#pragma unroll
    for (int u = 0; u < kWgPairs; ++u) { a_c[u] = a_n[u]; b_c[u] = b_n[u]; }
    slot = slot + 1 == kWgSlots ? 0 : slot + 1;
  }
#if MONO_WGRAD_DEBUG == 3 || MONO_WGRAD_DEBUG == 5
  if ((threadIdx.x & 63) == 0 && (blockIdx.x == 0 || blockIdx.x == 100))
    printf("block %d: %d stages, %lld shader clocks, %lld wall ticks (100 MHz); waiting: lgkmcnt %lld, vmcnt %lld, barrier %lld; read issue %lld, mfma + dma issue %lld\n", blockIdx.x, n_full,
           clock64() - t0, wall_clock64() - w0, t_lgkm, t_vm, t_bar, t_rd, t_body);
#endif
  // the split's last rows (fewer than a stage, possibly an odd count): straight from global memory, guarded
  {
    const int k_tail = k_begin + n_full * kWgStageRows;
    const float *ap = dy + (long long)(k_tail + h) * ldy + m0 + c, *bp = x + (long long)(k_tail + h) * ldx + n0 + c;
    for (int k = k_tail; k < k_end; k += 2) {
      const bool live = k + h < k_end;
      const float a = live ? *ap : 0.f, b = live ? *bp : 0.f;
      ap += 2 * ldy; bp += 2 * ldx;
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
      bsum += a;
    }
  }
  // acc[4 g + t] = D[m0 + 8 g + 4 h + t][n0 + c]
  float *wp = ws + (long long)s * ((long long)M * N + M);
#pragma unroll
  for (int v = 0; v < 16; ++v)
    wp[(long long)(m0 + 8 * (v >> 2) + 4 * h + (v & 3)) * N + n0 + c] = acc0[v] + acc1[v];
  if (tn == 0 && (wave & 1) == 0) {
    const float t = bsum + __shfl_xor(bsum, 32);
    if (h == 0) wp[(long long)M * N + m0 + c] = t;
  }
}
#undef WG_VMCNT
#undef WG_LGKMCNT0

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
