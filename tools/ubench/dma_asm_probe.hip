#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void_t;
// saddr form + M0 from an AS3 pointer; 160 KB LDS, destination beyond 64 KB
__global__ __launch_bounds__(256) void k(const float *src, float *dst) {
  __shared__ float4 buf[10000];                         // 160,000 bytes
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t *)buf;
  const unsigned row = 9000 + (threadIdx.x >> 6) * 64;   // float4 index: byte 144,000+
  const unsigned voff = threadIdx.x * 16;               // byte offset of this lane's 16 bytes
  const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + row * 16);
  if ((threadIdx.x & 63) < 48)                          // partial EXEC
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(m0v), "v"(voff), "s"(src) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float4 v = buf[9000 + (threadIdx.x ^ 1)];
  dst[threadIdx.x] = v.x + v.y * 0.001f;
}
int main() {
  float *s, *d; (void)hipMalloc(&s, 4096 * 4); (void)hipMalloc(&d, 1024 * 4);
  static float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = i; (void)hipMemcpy(s, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 256>>>(s, d); float o[256]; (void)hipMemcpy(o, d, sizeof(o), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) { int j = i ^ 1; float want = ((j & 63) < 48) ? 4 * j + (4 * j + 1) * 0.001f : o[i]; if (o[i] != want) ++bad; }
  printf("dma2: o[0]=%g o[1]=%g o[70]=%g bad=%d err=%s\n", o[0], o[1], o[70], bad, hipGetErrorString(hipGetLastError())); return bad != 0;
}
