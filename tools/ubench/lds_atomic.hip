// Micro-benchmark: LDS float atomic-add throughput on gfx950 vs. plain LDS read-modify-write.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, int stride_rows) {
  __shared__ float acc[512 * 32];
  for (int i = threadIdx.x; i < 512 * 32; i += 256) acc[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane >> 3, sub = lane & 7;
  float v = 1.0f + lane;
  int row = (wave * 8 + grp) * stride_rows;
  for (int it = 0; it < iters; ++it) {
    row = (row * 5 + 17 + it) & 511;
    float *r = acc + row * 32;
    if (MODE == 0) {            // 8 lanes x 4 channels, rotated (conflict-free banks)
      const int r0 = grp & 3;
#pragma unroll
      for (int k = 0; k < 4; ++k) atomicAdd(r + sub * 4 + ((r0 + k) & 3), v);
    } else if (MODE == 1) {     // 8 lanes x 4 channels, unrotated (4-way bank conflicts)
#pragma unroll
      for (int k = 0; k < 4; ++k) atomicAdd(r + sub * 4 + k, v);
    } else if (MODE == 2) {     // lane = channel: half-wave per row (2 rows per instruction)
      float *rr = acc + ((row + (lane >> 5)) & 511) * 32;
#pragma unroll
      for (int k = 0; k < 4; ++k) atomicAdd(rr + (lane & 31), v);
    } else if (MODE == 3) {     // non-atomic float4 RMW (racy; throughput reference)
      float4 *p = reinterpret_cast<float4 *>(r + sub * 4);
      float4 x = *p;
      x.x += v; x.y += v; x.z += v; x.w += v;
      *p = x;
    } else if (MODE == 4) {     // returning atomic
      float s = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) s += atomicAdd(r + sub * 4 + ((grp + k) & 3), v);
      v += s * 1e-30f;
    }
  }
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = acc[threadIdx.x];
}

template <int MODE>
void run(const char *name, float *d, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 2;
  k<MODE><<<blocks, 256>>>(d, 10, 1);
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(d, iters, 1);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // lane-adds: blocks*256 threads * iters * 4
  double lane_ops = (double)blocks * 256 * iters * 4;
  double per_cu_per_clk = lane_ops / (ms * 1e-3) / 256 / 2.4e9;
  printf("%-44s %8.3f ms  %6.2f lane-adds/clk/CU (at 2.4 GHz)\n", name, ms, per_cu_per_clk);
}

int main() {
  float *d; hipMalloc(&d, 512 * 256 * 4);
  const int iters = 20000;
  run<0>("ds_add_f32 8x4 rotated (no bank conflict)", d, iters);
  run<1>("ds_add_f32 8x4 unrotated", d, iters);
  run<2>("ds_add_f32 lane=channel (2 rows/instr)", d, iters);
  run<3>("plain float4 RMW (racy reference)", d, iters);
  run<4>("ds_add_rtn_f32 8x4 rotated", d, iters);
  hipFree(d);
  return 0;
}
