// Which spelling of log reproduces PyTorch's aten::log floats?  One kernel per variant (no cross-variant CSE).
// usage: log_variants arg.bin neglog.bin n
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
extern "C" __device__ float __ocml_log_f32(float);
extern "C" __device__ float __ocml_native_log_f32(float);
template <int V> __global__ void k(const float *a, float *o, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = a[i];
  float r;
  if (V == 0) r = logf(x);
  else if (V == 1) r = __ocml_log_f32(x);
  else if (V == 2) r = __logf(x);
  else if (V == 3) r = __ocml_native_log_f32(x);
  else if (V == 4) r = log2f(x) * 0.6931471805599453f;
  else if (V == 5) r = (float)log((double)x);
  else r = __builtin_amdgcn_logf(x) * 0.6931471805599453f;
  o[i] = r;
}
static std::vector<float> rd(const char *p, int n) { std::vector<float> v(n); FILE *f = fopen(p, "rb"); if (!f || fread(v.data(), 4, n, f) != (size_t)n) { printf("cannot read %s\n", p); exit(1); } fclose(f); return v; }
int main(int argc, char **argv) {
  const int n = atoi(argv[3]);
  auto a = rd(argv[1], n), ref = rd(argv[2], n);
  float *da, *dout; (void)hipMalloc(&da, 4 * n); (void)hipMalloc(&dout, 4 * n);
  (void)hipMemcpy(da, a.data(), 4 * n, hipMemcpyHostToDevice);
  const char *names[7] = {"logf", "__ocml_log_f32", "__logf", "__ocml_native_log_f32", "log2f * ln2", "(float)log((double)x)", "v_log_f32 * ln2"};
  std::vector<float> o(n);
  for (int v = 0; v < 7; ++v) {
    switch (v) {
      case 0: k<0><<<(n + 255) / 256, 256>>>(da, dout, n); break;
      case 1: k<1><<<(n + 255) / 256, 256>>>(da, dout, n); break;
      case 2: k<2><<<(n + 255) / 256, 256>>>(da, dout, n); break;
      case 3: k<3><<<(n + 255) / 256, 256>>>(da, dout, n); break;
      case 4: k<4><<<(n + 255) / 256, 256>>>(da, dout, n); break;
      case 5: k<5><<<(n + 255) / 256, 256>>>(da, dout, n); break;
      default: k<6><<<(n + 255) / 256, 256>>>(da, dout, n); break;
    }
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(o.data(), dout, 4 * n, hipMemcpyDeviceToHost);
    long bad = 0; double worst = 0;
    for (int i = 0; i < n; ++i) if (o[i] != -ref[i] && !(o[i] != o[i])) { ++bad; double e = fabs((double)o[i] + ref[i]) / fabs(ref[i] + 1e-30); if (e > worst) worst = e; }
    printf("%-26s differs from -torch(-log) in %ld of %d  (worst rel %.2e)\n", names[v], bad, n, worst);
  }
  return 0;
}
