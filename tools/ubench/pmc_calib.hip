// Calibration kernels for rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM section:
// FETCH_SIZE reads 1/2 of a wide coalesced stream; other access shapes must be calibrated on a known
// byte count).  Two shapes with known HBM bytes, both larger than the 256 MiB Infinity Cache:
//   calib_stream_f4 : every lane reads consecutive float4 (1 KiB per wave instruction), writes 1/16 of it
//   calib_rowgather : 8 lanes x float4 fetch one random 128-B row of a 1-KiB-strided table (the MSDA
//                     corner fetch), each row fetched exactly once
//   calib_seg32     : 4 lanes x 8 bytes fetch one 32-byte segment of a random 1-KiB token (the row-tile scatter's read of
//                     one level's four sampling points): which request size does the counter see per segment?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

__global__ __launch_bounds__(256) void calib_stream_f4(const float4 *in, float4 *out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float4 acc = make_float4(0, 0, 0, 0);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 15 * stride < n; i += 16 * stride) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float4 v = in[i + k * stride];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  // plain streaming copy of the first n/4 elements: n*4 bytes written exactly once
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n / 4; j += stride) {
    float4 v = in[j];
    v.x += acc.x * 1e-30f;
    out[j] = v;
  }
}

__global__ __launch_bounds__(256) void calib_rowgather(const float *table, const int *rows, float4 *out, int n_rows) {
  const int sub = threadIdx.x & 7;
  const int stride = gridDim.x * 32;
  float4 acc = make_float4(0, 0, 0, 0);
  for (int r = blockIdx.x * 32 + (threadIdx.x >> 3); r < n_rows; r += stride) {
    const float4 v = *reinterpret_cast<const float4 *>(table + (size_t)rows[r] * 256 + sub * 4);   // 1 KiB token stride
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void calib_seg32(const float *table, const int *rows, float2 *out, int n_rows) {
  const int sub = threadIdx.x & 3;
  const int stride = gridDim.x * 64;
  float2 acc = make_float2(0, 0);
  for (int r = blockIdx.x * 64 + (threadIdx.x >> 2); r < n_rows; r += stride) {
    const float2 v = *reinterpret_cast<const float2 *>(table + (size_t)rows[r] * 256 + 40 + sub * 2);   // bytes 160..191 of the token
    acc.x += v.x; acc.y += v.y;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
  const size_t n_f4 = (size_t)64 << 20;               // 1 GiB stream
  float4 *in, *out;
  if (hipMalloc(&in, n_f4 * 16) != hipSuccess || hipMalloc(&out, n_f4 * 16 / 4 + (1 << 20)) != hipSuccess) return 1;
  (void)hipMemset(in, 0, n_f4 * 16);
  calib_stream_f4<<<2048, 256>>>(in, out, n_f4);
  // row gather: table of 2M tokens x 1 KiB = 2 GiB, each token's head-0 row (128 B) read once in random order
  const int n_rows = 2 << 20;
  float *table; int *rows; float4 *out2;
  if (hipMalloc(&table, (size_t)n_rows * 1024) != hipSuccess || hipMalloc(&rows, n_rows * 4) != hipSuccess ||
      hipMalloc(&out2, 2048 * 256 * 16) != hipSuccess) return 1;
  (void)hipMemset(table, 0, (size_t)n_rows * 1024);
  std::vector<int> h(n_rows);
  std::iota(h.begin(), h.end(), 0);
  std::shuffle(h.begin(), h.end(), std::mt19937(1));
  (void)hipMemcpy(rows, h.data(), n_rows * 4, hipMemcpyHostToDevice);
  calib_rowgather<<<2048, 256>>>(table, rows, out2, n_rows);
  (void)hipDeviceSynchronize();
  (void)hipMemset(in, 0, n_f4 * 16);                  // push the table out of the Infinity Cache again
  calib_seg32<<<2048, 256>>>(table, rows, reinterpret_cast<float2 *>(out2), n_rows);
  (void)hipDeviceSynchronize();
  printf("calib_stream_f4: read %zu bytes, wrote %zu bytes\n", n_f4 * 16 + n_f4 * 4, n_f4 * 4);
  printf("calib_rowgather: read %zu bytes of rows (+%d index bytes)\n", (size_t)n_rows * 128, n_rows * 4);
  printf("calib_seg32: read %zu bytes of 32-byte segments (+%d index bytes)\n", (size_t)n_rows * 32, n_rows * 4);
  return 0;
}
