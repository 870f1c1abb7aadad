// Issue cost of the instruction classes the MSDA window / scatter kernels are made of, gfx950 (tools/ubench/issue_rate.hip).
//   hipcc -O3 --offload-arch=gfx950 -o tools/ubench/issue_rate tools/ubench/issue_rate.hip && tools/ubench/issue_rate
// Every wave runs ITER trips of 16 INDEPENDENT instances of one instruction (16 registers, no chain shorter than 16 issues),
// at 1 / 2 / 4 waves per SIMD, one workgroup per CU.  Reported: SIMD clocks per wave-instruction = kernel time x clock /
// (ITER x 16 x waves per SIMD); the clock is measured alongside with s_memtime (100 MHz) against clock64-free wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

constexpr int ITER = 2048;     // x 16 instructions; the loops below run ITER / 8 trips of 8 x 16 (128 instructions per branch)

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

// One kernel per instruction: BODY is an asm template over %0 (in/out register of the instance), %1, %2 (loop-invariant inputs)
#define DEFINE_KERNEL(NAME, ASM)                                                                                         \
  __global__ __launch_bounds__(1024) void k_##NAME(float *out, float a, float b) {                                     \
    float r[16];                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) r[i] = (float)(threadIdx.x + i) * 1.0001f;                          \
    float x = a + (float)(threadIdx.x & 7), y = b;                                                                      \
    _Pragma("unroll 1") for (int it = 0; it < ITER / 8; ++it) {                                                         \
      _Pragma("unroll") for (int rep = 0; rep < 8; ++rep) {                                                            \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(x), "v"(y));               \
      }                                                                                                                  \
    }                                                                                                                    \
    float s = 0.f;                                                                                                       \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) s += r[i];                                                          \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                      \
  }

DEFINE_KERNEL(fma, "v_fma_f32 %0, %0, %1, %2")
DEFINE_KERNEL(fmac, "v_fmac_f32 %0, %1, %2")
DEFINE_KERNEL(mul, "v_mul_f32 %0, %0, %1")
DEFINE_KERNEL(add, "v_add_f32 %0, %0, %1")
DEFINE_KERNEL(max, "v_max_f32 %0, %0, %1")
DEFINE_KERNEL(addu, "v_add_u32 %0, %0, %1")
DEFINE_KERNEL(subu, "v_sub_u32 %0, %0, %1")
DEFINE_KERNEL(xor_, "v_xor_b32 %0, %0, %1")
DEFINE_KERNEL(xor_imm, "v_xor_b32 %0, 0x70, %0")
DEFINE_KERNEL(and_, "v_and_b32 %0, %0, %1")
DEFINE_KERNEL(lshl, "v_lshlrev_b32 %0, 3, %0")
DEFINE_KERNEL(lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
DEFINE_KERNEL(add3, "v_add3_u32 %0, %0, %1, %2")
DEFINE_KERNEL(mad24, "v_mad_u32_u24 %0, %0, %1, %2")
DEFINE_KERNEL(mul_lo, "v_mul_lo_u32 %0, %0, %1")
DEFINE_KERNEL(mov, "v_mov_b32 %0, %1")
DEFINE_KERNEL(cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEFINE_KERNEL(cndmask_s, "v_cndmask_b32 %0, %0, %1, s[20:21]")
DEFINE_KERNEL(cmp_cnd_vcc, "v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc")
DEFINE_KERNEL(cmp_cnd_sgpr, "v_cmp_gt_f32 s[20:21], %0, %1\n\tv_cndmask_b32 %0, %0, %2, s[20:21]")
DEFINE_KERNEL(cnd_vcc_fma, "v_cndmask_b32 %0, %0, %1, vcc\n\tv_fma_f32 %0, %0, %1, %2")
DEFINE_KERNEL(cnd_vcc_e64, "v_cndmask_b32_e64 %0, %0, %1, vcc")
DEFINE_KERNEL(cmp, "v_cmp_gt_f32 vcc, %0, %1")
DEFINE_KERNEL(cmp_s, "v_cmp_gt_f32 s[20:21], %0, %1")
DEFINE_KERNEL(floor, "v_floor_f32 %0, %0")
DEFINE_KERNEL(cvt_i, "v_cvt_i32_f32 %0, %0")
DEFINE_KERNEL(cvt_f, "v_cvt_f32_i32 %0, %0")
DEFINE_KERNEL(exp, "v_exp_f32 %0, %0")
DEFINE_KERNEL(rcp, "v_rcp_f32 %0, %0")
DEFINE_KERNEL(mov_dpp_quad, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
DEFINE_KERNEL(mov_dpp_mirror, "v_mov_b32_dpp %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf")
DEFINE_KERNEL(add_dpp_quad, "v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
DEFINE_KERNEL(add_dpp_shr, "v_add_f32_dpp %0, %1, %0 row_shr:4 row_mask:0xf bank_mask:0xf")
DEFINE_KERNEL(fmac_dpp_bcast, "v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf")
DEFINE_KERNEL(addu_dpp_bcast, "v_add_u32_dpp %0, %1, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf")
DEFINE_KERNEL(perm, "v_perm_b32 %0, %0, %1, %2")
DEFINE_KERNEL(bfe, "v_bfe_u32 %0, %0, 3, 5")
DEFINE_KERNEL(and_or, "v_and_or_b32 %0, %0, %1, %2")
DEFINE_KERNEL(xad, "v_xad_u32 %0, %0, %1, %2")
DEFINE_KERNEL(min_i, "v_min_i32 %0, %0, %1")
DEFINE_KERNEL(med3, "v_med3_i32 %0, %0, %1, %2")
DEFINE_KERNEL(readlane, "v_readlane_b32 s20, %0, 3")
DEFINE_KERNEL(s_add, "s_add_u32 s20, s20, s21")
DEFINE_KERNEL(s_mul, "s_mul_i32 s20, s20, s21")

// packed: 16 register PAIRS
typedef float v2f __attribute__((ext_vector_type(2)));
#define DEFINE_PK(NAME, ASM)                                                                                             \
  __global__ __launch_bounds__(1024) void k_##NAME(float *out, float a, float b) {                                     \
    v2f r[16];                                                                                                           \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) r[i] = (v2f){(float)(threadIdx.x + i), (float)i};                   \
    v2f x = (v2f){a, a * 0.5f}, y = (v2f){b, b + 1.f};                                                                  \
    _Pragma("unroll 1") for (int it = 0; it < ITER / 8; ++it) {                                                         \
      _Pragma("unroll") for (int rep = 0; rep < 8; ++rep) {                                                            \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(x), "v"(y));               \
      }                                                                                                                  \
    }                                                                                                                    \
    float s = 0.f;                                                                                                       \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) s += r[i].x + r[i].y;                                               \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                      \
  }
DEFINE_PK(pk_fma, "v_pk_fma_f32 %0, %0, %1, %2")
DEFINE_PK(pk_mul, "v_pk_mul_f32 %0, %0, %1")
DEFINE_PK(pk_add, "v_pk_add_f32 %0, %0, %1")
DEFINE_PK(pk_mov, "v_pk_mov_b32 %0, %1, %2")

// LDS reads: 16 independent destinations per trip, conflict-free addresses (lane * 16 bytes), drained once per trip
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void k_ds_read_b128(float *out, float a, float b) {
  __shared__ float4 buf[2048];
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) buf[i] = make_float4(a, b, a, b);
  __syncthreads();
  v4f r[16];
  const unsigned base = (unsigned)(size_t)reinterpret_cast<char *>(buf) + (threadIdx.x & 63) * 16;
#pragma unroll 1
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[i]) : "v"(base), "n"(i * 1024));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(r[i]));
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += r[i].x;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the row block of the window forward as it stands, in inline asm (the compiler must not re-shape it): per half row 4 x
// (address instruction + ds_read_b128), one wait, 16 v_fmac.  XOR = 1: v_xor per read; XOR = 0: immediate offsets, no address instruction.
template <int XOR>
__global__ __launch_bounds__(1024) void k_rowmix(float *out, float a, float b) {
  __shared__ float4 buf[2048];
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) buf[i] = make_float4(a, b, a, b);
  __syncthreads();
  float acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = 0.f;
  // lane -> a row of its own; rotation by lane PAIR (row parity = lane parity picks the bank half, the rotation the slot: the 16 lanes
  // of an LDS group land on 16 different bank slots) so that the XOR form is conflict-free; the immediate form reads slot k of 64 rows at
  // a pitch of 144 bytes (conflict-free for consecutive rows)
  unsigned o = XOR ? (unsigned)(size_t)reinterpret_cast<char *>(buf) + (threadIdx.x & 63) * 128 + ((threadIdx.x >> 1) & 7) * 16
                   : (unsigned)(size_t)reinterpret_cast<char *>(buf) + (threadIdx.x & 63) * 144;
  float w = a;
#pragma unroll 1
  for (int it = 0; it < ITER / 4; ++it) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {                 // 8 "corners": 64 row reads, 256 FMAs, 64 (or 0) address instructions
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        v4f vv[4];
        if (XOR) {
          unsigned t[4];
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) {
            asm volatile("v_xor_b32 %0, %1, %2" : "=v"(t[s2]) : "n"((4 * h + s2) * 16), "v"(o));
            asm volatile("ds_read_b128 %0, %1" : "=v"(vv[s2]) : "v"(t[s2]));
          }
        } else {
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(vv[s2]) : "v"(o), "n"((4 * h + s2) * 16));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[(4 * h + s2) * 4 + 0]) : "v"(w), "v"(vv[s2].x));
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[(4 * h + s2) * 4 + 1]) : "v"(w), "v"(vv[s2].y));
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[(4 * h + s2) * 4 + 2]) : "v"(w), "v"(vv[s2].z));
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[(4 * h + s2) * 4 + 3]) : "v"(w), "v"(vv[s2].w));
        }
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 32; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

struct Case { const char *name; void (*fn)(float *, float, float); double per_trip; };

int main(int argc, char **argv) {
  float *out;
  (void)hipMalloc(&out, sizeof(float) * 256 * 1024);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  int clk_khz = 0;
  (void)hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  printf("device %s, %d CUs, nominal clock %.2f GHz\n", prop.name, prop.multiProcessorCount, clk_khz * 1e-6);
#define C(NAME) {#NAME, k_##NAME, 16.0}
  Case cases[] = {C(fma), C(fmac), C(mul), C(add), C(max), C(addu), C(subu), C(xor_), C(xor_imm), C(and_), C(lshl), C(lshl_add), C(add3),
                  C(mad24), C(mul_lo), C(mov), C(cndmask), C(cndmask_s), {"cmp+cnd via vcc (2 instr)", k_cmp_cnd_vcc, 32.0}, {"cmp+cnd via sgpr (2 instr)", k_cmp_cnd_sgpr, 32.0}, {"cnd_vcc+fma (2 instr)", k_cnd_vcc_fma, 32.0}, C(cnd_vcc_e64), C(cmp), C(cmp_s), C(floor), C(cvt_i), C(cvt_f), C(exp), C(rcp),
                  C(mov_dpp_quad), C(mov_dpp_mirror), C(add_dpp_quad), C(add_dpp_shr), C(fmac_dpp_bcast), C(addu_dpp_bcast), C(perm), C(bfe),
                  C(and_or), C(xad), C(min_i), C(med3), C(readlane), C(s_add), C(s_mul), C(pk_fma), C(pk_mul), C(pk_add), C(pk_mov),
                  {"ds_read_b128", k_ds_read_b128, 16.0},
                  {"rowmix_xor(64rd+64xor+256fma)/4", k_rowmix<1>, (64 + 64 + 256) / 4.0},
                  {"rowmix_imm(64rd+256fma)/4", k_rowmix<0>, (64 + 256) / 4.0}};
  const int n_cu = prop.multiProcessorCount;
  printf("%-34s %10s %10s %10s   (SIMD clocks per wave-instruction at the NOMINAL clock)\n", "instruction", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD");
  for (const Case &c : cases) {
    printf("%-34s", c.name);
    for (int threads : {256, 512, 1024}) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(c.fn, dim3(n_cu), dim3(threads), 0, 0, out, 1.0001f, 0.5f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
      }
      const double instr_per_simd = (double)ITER * c.per_trip * (threads / 256);
      printf(" %10.2f", best * 1e-3 * clk_khz * 1e3 / instr_per_simd);
    }
    printf("\n");
  }
  return 0;
}
