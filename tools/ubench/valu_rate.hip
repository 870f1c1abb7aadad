// VALU issue rate on gfx950: v_fma_f32 vs v_pk_fma_f32, N waves per SIMD (tools/ubench/valu_rate.hip; hipcc --offload-arch=gfx950).
// Every wave runs ITER trips of 16 independent accumulator pairs: PK = 0: 32 v_fma_f32, PK = 1: 16 v_pk_fma_f32 (the same 64 FMAs per lane).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096;
template <int PK>
__global__ __launch_bounds__(1024) void rate_kernel(float *out, float a, float b, long long *clk) {
  v2f acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (v2f){(float)threadIdx.x + i, (float)i};
  const v2f va = (v2f){a, a * 0.5f}, vb = (v2f){b, b + 1.f};
  const long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (PK) acc[i] = __builtin_elementwise_fma(acc[i], va, vb);
      else { acc[i].x = __builtin_fmaf(acc[i].x, va.x, vb.x); acc[i].y = __builtin_fmaf(acc[i].y, va.y, vb.y); }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(acc[i]));
  }
  const long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}
int main() {
  float *out; long long *clk, h;
  hipMalloc(&out, sizeof(float) * 256 * 1024 * 4); hipMalloc(&clk, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int threads : {256, 512, 1024}) for (int pk = 0; pk < 2; ++pk) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (pk) rate_kernel<1><<<256, threads>>>(out, 1.0001f, 0.5f, clk); else rate_kernel<0><<<256, threads>>>(out, 1.0001f, 0.5f, clk);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    const double fma_per_s = 256.0 * threads * ITER * 32 / (ms * 1e-3);
    printf("%-14s %2d waves/SIMD: %.3f ms, %6.1f TFLOP/s, wave clocks per trip %.1f (s_memtime ticks)\n", pk ? "v_pk_fma_f32" : "v_fma_f32",
           threads / 256, ms, 2 * fma_per_s / 1e12, (double)h / ITER);
  }
  return 0;
}
