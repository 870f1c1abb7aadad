// Micro-benchmark 2: integer LDS atomics (u32 / u64), f64, vs ds_add_f32, on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d\n", (int)e_); return; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
  __shared__ unsigned long long acc64[256 * 32];     // 64 KB
  unsigned int *acc32 = reinterpret_cast<unsigned int *>(acc64);
  float *accf = reinterpret_cast<float *>(acc64);
  double *accd = reinterpret_cast<double *>(acc64);
  for (int i = threadIdx.x; i < 256 * 32; i += 256) acc64[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane >> 3, sub = lane & 7;
  int row = (wave * 8 + grp);
  const unsigned v = 1 + lane;
  for (int it = 0; it < iters; ++it) {
    row = (row * 5 + 17 + it) & 255;
    const int r0 = grp & 3;
    if (MODE == 0) {            // u32, 8 lanes x 4 channels rotated
#pragma unroll
      for (int k = 0; k < 4; ++k) atomicAdd(acc32 + row * 32 + sub * 4 + ((r0 + k) & 3), v);
    } else if (MODE == 1) {     // u64, 8 lanes x 4 channels rotated
#pragma unroll
      for (int k = 0; k < 4; ++k) atomicAdd(acc64 + row * 32 + sub * 4 + ((r0 + k) & 3), (unsigned long long)v);
    } else if (MODE == 2) {     // f32
#pragma unroll
      for (int k = 0; k < 4; ++k) atomicAdd(accf + row * 32 + sub * 4 + ((r0 + k) & 3), (float)v);
    } else if (MODE == 3) {     // f64
#pragma unroll
      for (int k = 0; k < 4; ++k) atomicAdd(accd + row * 32 + sub * 4 + ((r0 + k) & 3), (double)v);
    } else if (MODE == 4) {     // u32, lane = channel (half-wave per row)
#pragma unroll
      for (int k = 0; k < 4; ++k) atomicAdd(acc32 + ((row + (lane >> 5) + k) & 255) * 32 + (lane & 31), v);
    } else if (MODE == 5) {     // u32 returning
      unsigned s = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) s += atomicAdd(acc32 + row * 32 + sub * 4 + ((r0 + k) & 3), v);
      row += s & 1;
    }
  }
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = (float)acc64[threadIdx.x];
}

template <int MODE>
void run(const char *name, float *d, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int blocks = 512;
  k<MODE><<<blocks, 256>>>(d, 10);
  CK(hipEventRecord(e0));
  k<MODE><<<blocks, 256>>>(d, iters);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double lane_ops = (double)blocks * 256 * iters * 4;
  printf("%-40s %8.3f ms  %6.2f lane-adds/clk/CU (at 2.4 GHz)\n", name, ms, lane_ops / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
  float *d;
  if (hipMalloc(&d, 512 * 256 * 4) != hipSuccess) return 1;
  const int iters = 4000;
  run<0>("ds_add_u32 8x4 rotated", d, iters);
  run<1>("ds_add_u64 8x4 rotated", d, iters);
  run<2>("ds_add_f32 8x4 rotated", d, iters);
  run<3>("ds_add_f64 8x4 rotated", d, iters);
  run<4>("ds_add_u32 lane=channel", d, iters);
  run<5>("ds_add_rtn_u32 8x4 rotated", d, iters);
  return 0;
}
