// What one wave per SIMD sustains on v_mfma_f32_32x32x2_f32 with the small-linear weight-gradient kernel's ingredients added one at a time:
//   0 bare chain (2 accumulators)   1 + 2 LDS dwords per MFMA (ds_read2st64), read one stage ahead   2 + a workgroup barrier per 8 MFMAs
//   3 = 1 + 2 with a second, idle-at-barriers wave per SIMD      4 = 2 with the operands read right before use
// Prints shader clocks per MFMA (wave 0 of block 0) and the kernel's wall time.   hipcc --offload-arch=gfx950 -O3 -o mfma_f32_chain mfma_f32_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kStages = 275 / 8 * 8 / 8;   // 34 stages of 8 MFMAs

template <int MODE, int WAVES, int AHEAD> __global__ __launch_bounds__(WAVES * 64) void k(float *out, long long *clk, const float *src) {
  __shared__ float ring[3][2][1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 3 * 2 * 1024; i += WAVES * 64) (&ring[0][0][0])[i] = 1e-3f * (i % 97);
  __syncthreads();
  if (wave >= 4) {                                   // extra waves: meet the barriers; modes 4 / 5 / 6: also act as loaders
    if (MODE >= 4) {
      const int lw = wave - 4;
      int passed = -1;
      long long t_work = 0, t_lbar = 0;
      for (int j = lw; j < kStages; j += 4) {
        float4 r[8];
#pragma unroll
        for (int p = 0; p < 8; ++p)
          r[p] = (MODE == 4) ? make_float4(1.f, 2.f, 3.f, (float)j) : *reinterpret_cast<const float4 *>(src + ((size_t)(blockIdx.x * 34 + j) * 8 + p) * 256 + 4 * lane);
        for (; passed < j - 1 - AHEAD; ++passed) __builtin_amdgcn_s_barrier();
        const long long l0 = clock64();
        float *slot = &ring[j % 3][0][0];
        if (MODE != 5) {
#pragma unroll
          for (int p = 0; p < 8; ++p) *reinterpret_cast<float4 *>(slot + p * 256 + 4 * lane) = r[p];
        } else {
          float t = 0.f;
#pragma unroll
          for (int p = 0; p < 8; ++p) t += r[p].x;
          if (t == 1.2345e-30f) slot[lane] = t;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const long long l1 = clock64();
        __builtin_amdgcn_s_barrier();
        const long long l2 = clock64();
        t_work += l1 - l0; t_lbar += l2 - l1;
        ++passed;
      }
      if (blockIdx.x == 0 && threadIdx.x == 256) { clk[2] = t_work; clk[3] = t_lbar; }
      for (; passed < kStages - 1; ++passed) __builtin_amdgcn_s_barrier();
      return;
    }
    if (MODE >= 2) for (int s = 0; s < kStages; ++s) __builtin_amdgcn_s_barrier();
    return;
  }
  f32x16 acc0 = {0}, acc1 = {0};
  float a[8], b[8], an[8], bn[8];
  const float *al = &ring[0][0][(wave >> 1) * 64 + lane], *bl = &ring[0][1][(wave & 1) * 64 + lane];
#pragma unroll
  for (int u = 0; u < 8; ++u) { a[u] = al[u * 128]; b[u] = bl[u * 128]; }
  float a2[8], b2[8];
  if (AHEAD == 2) {
#pragma unroll
    for (int u = 0; u < 8; ++u) { an[u] = al[2048 + u * 128]; bn[u] = bl[2048 + u * 128]; }
  }
  long long t_cbar = 0;
  float4 g0 = make_float4(1.f, 2.f, 3.f, 4.f), g1 = g0;
  const float *gp = src + ((size_t)blockIdx.x * 34 * 8 + wave * 2) * 256 + 4 * lane;
  const long long t0 = clock64();
  int slot = AHEAD == 2 ? 2 : 1;
#define STAGE(A, B, AN, BN)                                                                   \
    {                                                                                         \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                         \
        if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[u], B[u], acc1, 0, 0, 0);    \
        else acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[u], B[u], acc0, 0, 0, 0);          \
        if (MODE == 11 && u < 2) {                                                            \
          __builtin_amdgcn_sched_barrier(0);                                                  \
          float *d_ = &ring[(slot + 1) % 3][wave >> 1][((2 * wave) & 3) * 256 + 4 * lane];    \
          *reinterpret_cast<float4 *>(d_ + 256 * u) = u ? g1 : g0;                            \
          __builtin_amdgcn_sched_barrier(0);                                                  \
        }                                                                                     \
      }                                                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                      \
      if (MODE == 8 || MODE == 9) {                                                           \
        float *d_ = &ring[(slot + 1) % 3][wave >> 1][((2 * wave) & 3) * 256 + 4 * lane];      \
        *reinterpret_cast<float4 *>(d_) = g0; *reinterpret_cast<float4 *>(d_ + 256) = g1;     \
      }                                                                                       \
      if (MODE == 10) { if (g0.x + g1.y == 1.2345e-30f) ring[0][0][lane] = g0.x; }            \
      if (MODE >= 2) { if (AHEAD == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long long c0 = clock64(); __builtin_amdgcn_s_barrier(); t_cbar += clock64() - c0; } \
      __builtin_amdgcn_sched_barrier(0);                                                      \
      if (MODE == 9 || MODE == 10) {                                                          \
        g0 = *reinterpret_cast<const float4 *>(gp); g1 = *reinterpret_cast<const float4 *>(gp + 1024); gp += 2048; \
      }                                                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                      \
      const float *an_ = al + slot * 2048, *bn_ = bl + slot * 2048;                           \
      _Pragma("unroll") for (int u = 4; u < 8; ++u) {                                         \
        if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[u], B[u], acc1, 0, 0, 0);    \
        else acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[u], B[u], acc0, 0, 0, 0);          \
        if (MODE >= 1) {                                                                      \
          const int v = (u - 4) * 2;                                                          \
          AN[v] = an_[v * 128]; BN[v] = bn_[v * 128]; AN[v + 1] = an_[v * 128 + 128]; BN[v + 1] = bn_[v * 128 + 128]; \
        }                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                    \
      }                                                                                       \
      slot = slot == 2 ? 0 : slot + 1;                                                        \
    }
  if (AHEAD == 2) {
#pragma unroll 1
    for (int s = 0; s + 3 <= kStages; s += 3) { STAGE(a, b, a2, b2) STAGE(an, bn, a, b) STAGE(a2, b2, an, bn) }
    STAGE(a, b, a2, b2)                                   // 34 = 33 + 1
  } else {
#pragma unroll 1
    for (int s = 0; s < kStages; s += 2) {
      if (MODE >= 1) { STAGE(a, b, an, bn) STAGE(an, bn, a, b) } else { STAGE(a, b, a, b) STAGE(a, b, a, b) }
    }
  }
  const long long t1 = clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = t_cbar; }
  float r = 0.f;
#pragma unroll
  for (int v = 0; v < 16; ++v) r += acc0[v] + acc1[v];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE, int WAVES, int AHEAD = 1> void run(const char *what, int blocks, float *out, long long *clk, const float *src) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) k<MODE, WAVES, AHEAD><<<blocks, WAVES * 64>>>(out, clk, src);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 50; ++i) k<MODE, WAVES, AHEAD><<<blocks, WAVES * 64>>>(out, clk, src);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long c[4]; (void)hipMemcpy(c, clk, 32, hipMemcpyDeviceToHost);
  printf("%-64s %d blocks x %d waves: %6.1f clocks per MFMA, %.2f us per launch | per stage: MFMA wave at the barrier %lld; loader: work %lld, at its barrier %lld\n", what,
         blocks, WAVES, (double)c[0] / (kStages * 8), ms * 1e3 / 50, c[1] / kStages, c[2] / 9, c[3] / 9);
  (void)hipMemset(clk, 0, 32);
}

int main() {
  float *out; long long *clk; (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&clk, 32); (void)hipMemset(clk, 0, 32);
  float *src; (void)hipMalloc(&src, (size_t)512 * 34 * 8 * 256 * 4); (void)hipMemset(src, 0, (size_t)512 * 34 * 8 * 256 * 4);
  for (int blocks : {256}) {
    run<0, 4>("0 bare MFMA chain", blocks, out, clk, src);
    run<1, 4>("1 + 2 LDS dwords per MFMA, a stage ahead, in the MFMA gaps", blocks, out, clk, src);
    run<2, 4>("2 + barrier per 8 MFMAs (mid-run)", blocks, out, clk, src);
    run<2, 8>("3 + four more waves that only meet the barriers", blocks, out, clk, src);
    run<4, 8>("4 + those waves write the stages to LDS (8 ds_write_b128 each)", blocks, out, clk, src);
    run<5, 8>("5 + those waves load the stages from memory, no LDS writes", blocks, out, clk, src);
    run<6, 8>("6 + both: load, then write to LDS", blocks, out, clk, src);
    run<8, 4>("8 = 2, + each MFMA wave writes 2 ds_write_b128 per stage (constants)", blocks, out, clk, src);
    run<11, 4>("11 = 8 with the two writes right after MFMA 0 and MFMA 1 (three / two MFMAs before the barrier)", blocks, out, clk, src);
    run<10, 4>("10 = 2, + each MFMA wave loads 2 x 16 B per lane per stage, no LDS writes", blocks, out, clk, src);
    run<9, 4>("9 = 8 + 10: load (one stage ahead), write to LDS", blocks, out, clk, src);
    run<4, 8, 2>("4' as 4, operands read TWO stages ahead (no lgkmcnt wait at the barrier: a deeper ring would allow that)", blocks, out, clk, src);
    run<6, 8, 2>("6' as 6, operands read TWO stages ahead", blocks, out, clk, src);
  }
  return 0;
}
