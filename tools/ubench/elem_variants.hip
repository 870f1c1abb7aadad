// Which spelling of exp / log / sigmoid reproduces the floats of PyTorch's elementwise kernels on this ROCm build?
// usage: elem_variants x.bin exp.bin log.bin sigmoid.bin n     (files of float32 written by tools/ubench/elem_variants.py)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
extern "C" __device__ float __ocml_exp_f32(float);
extern "C" __device__ float __ocml_log_f32(float);
extern "C" __device__ float __ocml_native_exp_f32(float);
extern "C" __device__ float __ocml_native_log_f32(float);
__device__ __forceinline__ float rn_mul(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float rn_add(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float rn_sub(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}
__global__ void k(const float *x, const float *a1, float *o, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = x[i], a = fabsf(v) + 1e-8f;
  o[0 * n + i] = expf(v);
  o[1 * n + i] = __ocml_exp_f32(v);
  o[2 * n + i] = __builtin_expf(v);
  o[3 * n + i] = __expf(v);
  o[4 * n + i] = logf(a);
  o[5 * n + i] = __ocml_log_f32(a);
  o[6 * n + i] = __builtin_logf(a);
  o[7 * n + i] = __logf(a);
  o[8 * n + i] = 1.0f / (1.0f + expf(-v));
  o[9 * n + i] = 1.0f / (1.0f + __ocml_exp_f32(-v));
  o[10 * n + i] = __frcp_rn(1.0f + expf(-v));
  o[11 * n + i] = 1.0f / (1.0f + __expf(-v));
  o[12 * n + i] = __fdividef(1.0f, 1.0f + expf(-v));
  {
    const float x = v;
    const float p = 1.0f / rn_add(1.0f, expf(-x));
    const float neg = rn_mul(rn_mul(0.75f, rn_mul(p, p)), -logf(rn_add(rn_sub(1.0f, p), 1e-8f)));
    const float omp = rn_sub(1.0f, p);
    const float pos = rn_mul(rn_mul(0.25f, rn_mul(omp, omp)), -logf(rn_add(p, 1e-8f)));
    o[13 * n + i] = rn_sub(pos, neg);
    o[14 * n + i] = rn_add(rn_sub(1.0f, p), 1e-8f);
    o[15 * n + i] = -logf(a1[i]);
    o[17 * n + i] = __int_as_float(__float_as_int(logf(a1[i])) ^ 0x80000000);
    o[16 * n + i] = -logf(rn_add(rn_sub(1.0f, p), 1e-8f));
  }
}
static std::vector<float> rd(const char *p, int n) { std::vector<float> v(n); FILE *f = fopen(p, "rb"); if (!f || fread(v.data(), 4, n, f) != (size_t)n) { printf("cannot read %s\n", p); exit(1); } fclose(f); return v; }
int main(int argc, char **argv) {
  const int n = atoi(argv[5]);
  auto x = rd(argv[1], n), e = rd(argv[2], n), l = rd(argv[3], n), s = rd(argv[4], n);
  float *dx, *dout; hipMalloc(&dx, 4 * n); hipMalloc(&dout, 4 * n * 18);
  hipMemcpy(dx, x.data(), 4 * n, hipMemcpyHostToDevice);
  float *da; hipMalloc(&da, 4 * n); { auto a1 = rd(argv[7], n); hipMemcpy(da, a1.data(), 4 * n, hipMemcpyHostToDevice); }
  k<<<(n + 255) / 256, 256>>>(dx, da, dout, n); hipDeviceSynchronize();
  std::vector<float> o(18 * (size_t)n); hipMemcpy(o.data(), dout, 4 * (size_t)n * 18, hipMemcpyDeviceToHost);
  const char *names[18] = {"expf", "__ocml_exp_f32", "__builtin_expf", "__expf", "logf", "__ocml_log_f32", "__builtin_logf", "__logf",
                           "1/(1+expf(-x))", "1/(1+ocml_exp(-x))", "frcp_rn(1+expf(-x))", "1/(1+__expf(-x))", "fdividef(1,1+expf(-x))", "class cost", "arg (1-p)+1e-8 vs torch", "-logf(torch arg) vs torch", "neglog(1-p)", "logf(torch arg) sign-flipped as bits"};
  auto cc = rd(argv[6], n), ng = rd(argv[7], n), ps = rd(argv[8], n), nl = rd(argv[9], n), pp = rd(argv[10], n);
  for (int v = 0; v < 18; ++v) {
    const std::vector<float> &ref = v < 4 ? e : (v < 8 ? l : (v < 13 ? s : (v == 13 ? cc : (v == 14 ? ng : (v == 15 ? nl : (v == 16 ? nl : nl))))));
    long bad = 0;
    for (int i = 0; i < n; ++i) bad += !(o[(size_t)v * n + i] == ref[i] || (o[(size_t)v * n + i] != o[(size_t)v * n + i] && ref[i] != ref[i]));
    printf("%-26s differs from torch in %ld of %d\n", names[v], bad, n);
    if (v == 16) { int shown = 0; for (int i = 0; i < n && shown < 6; ++i) if (o[(size_t)v * n + i] != ref[i]) { printf("   x=%.9g  mine=%.9g torch=%.9g  sig=%.9g\n", x[i], o[(size_t)v * n + i], ref[i], s[i]); ++shown; } }
  }
  return 0;
}
