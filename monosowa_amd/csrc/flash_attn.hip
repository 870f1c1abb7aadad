// fp32 multi-head attention with head_dim 32 for gfx950 (MI355X): softmax(Q K^T / sqrt(d)) V with dropout on the
// attention probabilities, never materialising the [Lq, Lk] score matrix ("flash" formulation), on the exact-f32
// matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces the core of nn.MultiheadAttention in MonoDETR's depth encoder (depth_predictor/transformer.py:57-65,
// 1920 x 1920 tokens) and the decoder's depth cross-attention (depthaware_transformer.py:417-423, 550 x 1920),
// which PyTorch runs through a generic fp32 kernel at ~30 % of the f32 MFMA rate.
//
// Layout trick (all three kernels): scores are computed TRANSPOSED, S^T[key][query] = K Q^T, so that in the
// 32x32 accumulator layout (col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) a lane owns ONE
// query and 16 keys.  Row statistics (max, sum, the rescale factor, dropout seed) are then per-lane scalars, the
// only cross-lane step is one exchange between lanes l and l ^ 32, and P^T feeds the second product straight from
// its accumulator registers: O^T[d][q] += sum_key V^T[d][key] P^T[key][q] takes B = P^T register t at MFMA step
// t when A = V[key(t, half)][d = lane & 31] with key(t, half) = 8 (t >> 2) + 4 half + (t & 3).
#include <hip/hip_runtime.h>

#include "../../include/monosowa_attn.h"

namespace attn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef ATTN_FWD_WAVES
#define ATTN_FWD_WAVES 4        // waves per SIMD the forward is compiled for: 128 VGPRs, no spills (2: 130 VGPRs with dropout = 3 waves; in the step 703 -> 686 us on the 1920 x 1920 launch)
#endif
#ifndef ATTN_BWD_WAVES
#define ATTN_BWD_WAVES 3        // waves per SIMD the two backward kernels are compiled for (170 VGPRs; 2 waves / 256 VGPRs measures the same)
#endif
#ifndef ATTN_PREFETCH
#define ATTN_PREFETCH 0         // measurement builds only (with ATTN_BWD_WAVES=2: 210 VGPRs): all LDS operands of a 32-row half requested at its
                                // top instead of right in front of the MFMA that consumes them.  Measured round 4: 2.12 vs 2.09 ms on the
                                // 1920 x 1920 backward -- the LDS round trips are not what the MFMA pipe waits for
#endif
#ifndef ATTN_OPDEPTH
#define ATTN_OPDEPTH 0          // measurement builds: software-pipelining depth of the second-stage MFMAs' LDS operands (0: as the compiler places them)
#endif
#ifndef ATTN_OPDEPTH_KV
#define ATTN_OPDEPTH_KV 0       // the same for bwd_dkdv's two operand streams (no scheduling directives there: with them the 170-register build spilled)
#endif
#ifndef ATTN_EARLY_STORE
#define ATTN_EARLY_STORE 0   // measurement (round 5: 1.7 % SLOWER backward, forward +1 %, 7 instances spill): the next tile's LDS writes in
                             // front of the LAST MFMA block of an iteration instead of behind it -- at 3 - 4 waves per SIMD the other
                             // waves already cover the writes' completion latency in front of the barrier
#endif
#ifndef ATTN_PRIO
#define ATTN_PRIO 0             // measurement builds only: 1 = waves inside an MFMA cluster issue ahead of waves in their vector sections
                                // (s_setprio 2 around the clusters), 2 = the inverse (vector sections first)
#endif
#define ATTN_PRIO_MFMA()  do { if (ATTN_PRIO == 1) __builtin_amdgcn_s_setprio(2); else if (ATTN_PRIO == 2) __builtin_amdgcn_s_setprio(0); } while (0)
#define ATTN_PRIO_VALU()  do { if (ATTN_PRIO == 1) __builtin_amdgcn_s_setprio(0); else if (ATTN_PRIO == 2) __builtin_amdgcn_s_setprio(2); } while (0)
#ifndef ATTN_SKIP
#define ATTN_SKIP 0             // measurement builds only (bits, backward kernels): 1 no exp2 of the recomputed scores, 2 no dropout hash,
                                // 4 no second-stage MFMAs (dQ / dK, dV), 8 no recomputation MFMAs (S, dP), 16 no LDS reads of the A operands,
                                // 32 the hash's input word in place of the hash (NOT a usable mask: prices a mask of ~4 instructions per key pair)
#endif
constexpr int kBlockQ = 128;     // queries per workgroup (4 waves x 32)
constexpr int kTileK = 64;       // keys per LDS tile
constexpr int kKStride = 36;     // floats per K row in LDS: 16-lane ds_read_b128 phases hit 64 distinct banks
constexpr int kVStride = 40;     // floats per V row: rows r and r + 4 (the two lane halves) are 32 banks apart

struct Strides { long long b, h, t; };   // in floats; the 32 channels of a head are contiguous

struct Args {
  const float *q, *k, *v;
  float *o, *lse;                // lse: [B*H, Lq], log2 domain: row max + log2(row sum) of the scaled scores
  const float *dout, *delta;     // backward only
  float *dq, *dk, *dv;           // backward only
  int H, Lq, Lk;
  Strides sq, sk, sv, so;        // so also addresses dout
  Strides sdq, sdk, sdv;
  float scale_log2;              // softmax scale * log2(e)
  float scale;                   // softmax scale
  unsigned drop_thr16;           // drop when the element's 16 random bits < thr (0: no dropout)
  float drop_scale;              // 1 / (1 - p)
  unsigned seed_lo, seed_hi;
  const unsigned char *kmask;    // MASK kernels: key padding mask [B, Lk], non-zero = the key takes no part (its score is -inf)
  unsigned *keep;                // dropout keep bits (optional): written by the forward, read by the backward kernels instead of
                                 // re-hashing.  One word per (forward lane, key tile): [B*H][ceil(Lq/128)][ceil(Lk/64)][256 threads];
                                 // bit 16 * half + v = the lane's score register v of the tile's 32-key half (key = 64 kt + 32 half +
                                 // 4 (lane >> 5) + 8 (v >> 2) + (v & 3))
};
__device__ __forceinline__ long long keep_index(const Args &a, int bh, int q_block, int k_tile, int slot) {
  const int n_qb = (a.Lq + 127) >> 7, n_kt = (a.Lk + 63) >> 6;
  return (((long long)bh * n_qb + q_block) * n_kt + k_tile) * 256 + slot;
}

__device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
// 32 random bits for the key pair (2 kp, 2 kp + 1) of query q; element (q, key) uses half key & 1.
__device__ __forceinline__ unsigned drop_bits(unsigned head_seed, unsigned q, unsigned kp) {
  if (ATTN_SKIP & 32) return head_seed + q * 0x9E3779B1u + kp * 0x85EBCA77u;      // measurement build: what a mask that costs ~4 instructions per pair would buy
  return mix32(head_seed + q * 0x9E3779B1u + kp * 0x85EBCA77u);
}
__device__ __forceinline__ unsigned head_seed_of(const Args &a, unsigned bh) {
  return mix32(a.seed_lo ^ (bh * 0xC2B2AE3Du)) ^ a.seed_hi;
}
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32); }      // lane l <-> l ^ 32
__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// ---------------------------------------------------------------------------------------------- forward
// MASK: a key padding mask (Args::kmask) -- the reference passes one to both attention modules (depthaware_transformer.py:456-459,
// depth_predictor/transformer.py:57-60: key_padding_mask -> -inf before the softmax, torch.nn.functional.multi_head_attention_forward).
// The tile's keys carry an additive bias in LDS, 0 or -inf (also for the keys past Lk of a ragged last tile): one ds_read_b128 per
// four scores.  A query whose keys are ALL masked comes out as NaN, as it does from torch's softmax.
// Waves per SIMD by instance: the training instance of the shipped pipeline (dropout + keep bits, no padding) and the inference
// instances fit the 128 VGPRs of 4 waves; dropout together with the pad-bit test or with nothing to amortise the hash against does not
// (4 - 6 registers spilled inside the tile loop), so those are compiled for 3 waves (170 VGPRs): no instance spills.
template <bool DROP, bool MASK, bool BITS>
constexpr int fwd_waves() { return (DROP && (MASK || !BITS)) ? 3 : ATTN_FWD_WAVES; }
template <bool DROP, bool MASK = false, bool BITS = false>       // BITS (with DROP): also store the keep bits (Args::keep)
__global__ __launch_bounds__(256, (fwd_waves<DROP, MASK, BITS>())) void fwd_kernel(const Args a) {
  __shared__ float Ks[2][kTileK * kKStride];
  __shared__ float Vs[2][kTileK * kVStride];
  __shared__ unsigned long long Ms[2];                 // MASK: bit k = key k of the tile takes no part (padded, or past Lk)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.H, hd = bh % a.H;
  const int q = blockIdx.x * kBlockQ + wave * 32 + r;
  const int qc = min(q, a.Lq - 1);
  const float *qp = a.q + b * a.sq.b + hd * a.sq.h + (long long)qc * a.sq.t + 16 * h;
  float qreg[16];                                       // Q[q][16 h + t], pre-scaled: scores come out in the log2 domain
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4 t = ld4(qp + 4 * i);
    qreg[4 * i] = t.x * a.scale_log2; qreg[4 * i + 1] = t.y * a.scale_log2;
    qreg[4 * i + 2] = t.z * a.scale_log2; qreg[4 * i + 3] = t.w * a.scale_log2;
  }
  const float *kb = a.k + b * a.sk.b + hd * a.sk.h, *vb = a.v + b * a.sv.b + hd * a.sv.h;
  const int n_tiles = (a.Lk + kTileK - 1) / kTileK;
  float4 kr[2], vr[2];
  unsigned long long pad_next = 0;                      // MASK: the tile being loaded, as wave 0 sees it (a wave-uniform value: SGPRs)
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = threadIdx.x + 256 * j, key = kt * kTileK + (idx >> 3), c = (idx & 7) * 4;
      const bool ok = key < a.Lk;
      kr[j] = ok ? ld4(kb + (long long)key * a.sk.t + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      vr[j] = ok ? ld4(vb + (long long)key * a.sv.t + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (MASK && wave == 0) {
      const int key = kt * kTileK + lane;
      pad_next = __ballot(!(key < a.Lk && a.kmask[(long long)b * a.Lk + key] == 0));
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = threadIdx.x + 256 * j, row = idx >> 3, c = (idx & 7) * 4;
      *reinterpret_cast<float4 *>(&Ks[buf][row * kKStride + c]) = kr[j];
      *reinterpret_cast<float4 *>(&Vs[buf][row * kVStride + c]) = vr[j];
    }
    if (MASK && threadIdx.x == 0) Ms[buf] = pad_next;
  };
  load_tile(0);
  store_tile(0);
  __syncthreads();

  const unsigned hseed = DROP ? head_seed_of(a, (unsigned)bh) : 0u;
  f32x16 o = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, lsum = 0.f;
  for (int kt = 0; kt < n_tiles; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < n_tiles) load_tile(kt + 1);            // in flight during the products below
    // S^T = K Q^T for the two 32-key halves of the tile
    f32x16 s0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, s1 = s0;
    const float *k0 = &Ks[buf][r * kKStride + 16 * h], *k1 = k0 + 32 * kKStride;
    ATTN_PRIO_MFMA();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 a0 = ld4(k0 + 4 * i), a1 = ld4(k1 + 4 * i);
      s0 = mfma(a0.x, qreg[4 * i], s0);     s1 = mfma(a1.x, qreg[4 * i], s1);
      s0 = mfma(a0.y, qreg[4 * i + 1], s0); s1 = mfma(a1.y, qreg[4 * i + 1], s1);
      s0 = mfma(a0.z, qreg[4 * i + 2], s0); s1 = mfma(a1.z, qreg[4 * i + 2], s1);
      s0 = mfma(a0.w, qreg[4 * i + 3], s0); s1 = mfma(a1.w, qreg[4 * i + 3], s1);
    }
    ATTN_PRIO_VALU();
    const int key0 = kt * kTileK + 4 * h;                // + 8 (v >> 2) + (v & 3) (+ 32 for s1)
    if (MASK) {
      // The tile's 64 pad bits come through the scalar path: one uniform LDS read, two readfirstlanes, and per score a bit test of a
      // lane word -- 2 VGPRs.  (Round 4 kept a float bias per key in LDS and added it with eight ds_read_b128 per tile: at the 128-VGPR
      // cap of 4 waves per SIMD the masked instances spilled 8 - 13 registers.)
      const unsigned long long tm = Ms[buf];
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)tm), hi = __builtin_amdgcn_readfirstlane((unsigned)(tm >> 32));
      const unsigned w0 = lo >> (4 * h), w1 = hi >> (4 * h);       // this lane's keys: bit 8 (v >> 2) + (v & 3)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int bit = 8 * (v >> 2) + (v & 3);
        if ((w0 >> bit) & 1u) s0[v] = -INFINITY;
        if ((w1 >> bit) & 1u) s1[v] = -INFINITY;
      }
    } else if (kt == n_tiles - 1 && (a.Lk & (kTileK - 1))) {
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int key = key0 + 8 * (v >> 2) + (v & 3);
        if (key >= a.Lk) s0[v] = -INFINITY;
        if (key + 32 >= a.Lk) s1[v] = -INFINITY;
      }
    }
    float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
    for (int v = 1; v < 16; ++v) mx = fmaxf(mx, fmaxf(s0[v], s1[v]));
    mx = fmaxf(mx, xhalf(mx));
    const float m_new = fmaxf(m, mx);
    // MASK: while every key seen so far is padded the running maximum is -inf, and (-inf) - (-inf) would turn alpha and every
    // probability into NaN for the rest of the row (a leading tile of padded keys in front of valid ones: torch gives finite
    // output).  Subtracting 0 instead leaves alpha = 0 on an empty accumulator and p = exp2(-inf) = 0; a row whose keys are ALL
    // padded still ends with lsum = 0 and comes out as NaN, like torch's softmax.
    const float m_sub = (MASK && m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m - m_sub);
    float rs = 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      s0[v] = __builtin_amdgcn_exp2f(s0[v] - m_sub);
      s1[v] = __builtin_amdgcn_exp2f(s1[v] - m_sub);
      rs += s0[v] + s1[v];
    }
    lsum = lsum * alpha + rs;
    m = m_new;
#pragma unroll
    for (int v = 0; v < 16; ++v) o[v] *= alpha;
    if (DROP) {
      unsigned word = 0;                                  // this lane's 32 keep bits of the tile (Args::keep)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const unsigned kp = (unsigned)(key0 + 8 * g + 2 * pr) >> 1;
          const unsigned b0 = drop_bits(hseed, (unsigned)q, kp), b1 = drop_bits(hseed, (unsigned)q, kp + 16u);
          const int v = 4 * g + 2 * pr;
          const bool k00 = (b0 & 0xFFFFu) >= a.drop_thr16, k01 = (b0 >> 16) >= a.drop_thr16;
          const bool k10 = (b1 & 0xFFFFu) >= a.drop_thr16, k11 = (b1 >> 16) >= a.drop_thr16;
          s0[v] = k00 ? s0[v] * a.drop_scale : 0.f;
          s0[v + 1] = k01 ? s0[v + 1] * a.drop_scale : 0.f;
          s1[v] = k10 ? s1[v] * a.drop_scale : 0.f;
          s1[v + 1] = k11 ? s1[v + 1] * a.drop_scale : 0.f;
          if (BITS) word |= (k00 ? 1u << v : 0u) | (k01 ? 2u << v : 0u) | (k10 ? 0x10000u << v : 0u) | (k11 ? 0x20000u << v : 0u);
        }
      }
      if (BITS) a.keep[keep_index(a, bh, blockIdx.x, kt, threadIdx.x)] = word;
    }
#if ATTN_EARLY_STORE
    if (kt + 1 < n_tiles) store_tile(buf ^ 1);
#endif
    // O^T += V^T P^T
    ATTN_PRIO_MFMA();
    const float *v0 = &Vs[buf][(4 * h) * kVStride + r], *v1 = v0 + 32 * kVStride;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int row = 8 * (t >> 2) + (t & 3);
      o = mfma(v0[row * kVStride], s0[t], o);
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int row = 8 * (t >> 2) + (t & 3);
      o = mfma(v1[row * kVStride], s1[t], o);
    }
    ATTN_PRIO_VALU();
#if !ATTN_EARLY_STORE
    if (kt + 1 < n_tiles) store_tile(buf ^ 1);
#endif
    __syncthreads();
  }
  lsum += xhalf(lsum);
  if (q < a.Lq) {
    const float inv = 1.f / lsum;
    float *op = a.o + b * a.so.b + hd * a.so.h + (long long)q * a.so.t + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4 *>(op + 8 * g) = make_float4(o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv);
    if (h == 0) a.lse[(long long)bh * a.Lq + q] = m + __builtin_amdgcn_logf(lsum);     // v_log_f32 = log2
  }
}


// ---------------------------------------------------------------------------------------------- backward
// dQ: same decomposition as the forward (a lane owns one query); per 32-key half: S^T (recomputed), dP^T = V dO^T,
// dS^T = P^T o (dP^T - delta), dQ^T += K^T dS^T -- 48 MFMAs.
template <bool DROP, bool MASK = false, bool BITS = false>       // BITS (with DROP): the forward's keep bits instead of the hash
__global__ __launch_bounds__(256, ATTN_BWD_WAVES) void bwd_dq_kernel(const Args a) {
  __shared__ float Ks[2][kTileK * kKStride];
  __shared__ float Vs[2][kTileK * kKStride];
  __shared__ float Bs[2][MASK ? kTileK : 4];            // additive key bias, as in the forward
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.H, hd = bh % a.H;
  const int q = blockIdx.x * kBlockQ + wave * 32 + r;
  const int qc = min(q, a.Lq - 1);
  const float *qp = a.q + b * a.sq.b + hd * a.sq.h + (long long)qc * a.sq.t + 16 * h;
  const float *dop = a.dout + b * a.so.b + hd * a.so.h + (long long)qc * a.so.t + 16 * h;
  // delta[q] = sum_d dO[q][d] O[q][d] is evaluated here (the lane holds its half of the dO row anyway) and left in Args::delta
  // for bwd_dkdv, which runs behind this kernel on the same stream: no separate pass over dO and O
  const float *op = a.o + b * a.so.b + hd * a.so.h + (long long)qc * a.so.t + 16 * h;
  float qreg[16], doreg[16];
  float delta = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4 t = ld4(qp + 4 * i), u = ld4(dop + 4 * i), w = ld4(op + 4 * i);
    qreg[4 * i] = t.x * a.scale_log2; qreg[4 * i + 1] = t.y * a.scale_log2;
    qreg[4 * i + 2] = t.z * a.scale_log2; qreg[4 * i + 3] = t.w * a.scale_log2;
    doreg[4 * i] = u.x; doreg[4 * i + 1] = u.y; doreg[4 * i + 2] = u.z; doreg[4 * i + 3] = u.w;
    delta += u.x * w.x + u.y * w.y + u.z * w.z + u.w * w.w;
  }
  delta += xhalf(delta);
  if (h == 0 && q < a.Lq) const_cast<float *>(a.delta)[(long long)bh * a.Lq + q] = delta;
  const float lse = a.lse[(long long)bh * a.Lq + qc];
  const float *kb = a.k + b * a.sk.b + hd * a.sk.h, *vb = a.v + b * a.sv.b + hd * a.sv.h;
  const int n_tiles = (a.Lk + kTileK - 1) / kTileK;
  float4 kr[2], vr[2];
  float br = 0.f;
  constexpr bool saved_bits = DROP && BITS;              // the forward's keep bits instead of the hash (the same mask)
  unsigned word_next = 0;                                // this lane's keep word of the tile being loaded (the forward's lane layout)
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = threadIdx.x + 256 * j, key = kt * kTileK + (idx >> 3), c = (idx & 7) * 4;
      const bool ok = key < a.Lk;
      kr[j] = ok ? ld4(kb + (long long)key * a.sk.t + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      vr[j] = ok ? ld4(vb + (long long)key * a.sv.t + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (MASK && threadIdx.x < kTileK) {
      const int key = kt * kTileK + threadIdx.x;
      br = (key < a.Lk && a.kmask[(long long)b * a.Lk + key] == 0) ? 0.f : -INFINITY;
    }
    if (saved_bits) word_next = a.keep[keep_index(a, bh, blockIdx.x, kt, threadIdx.x)];
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = threadIdx.x + 256 * j, row = idx >> 3, c = (idx & 7) * 4;
      *reinterpret_cast<float4 *>(&Ks[buf][row * kKStride + c]) = kr[j];
      *reinterpret_cast<float4 *>(&Vs[buf][row * kKStride + c]) = vr[j];
    }
    if (MASK && threadIdx.x < kTileK) Bs[buf][threadIdx.x] = br;
  };
  load_tile(0);
  store_tile(0);
  __syncthreads();
  const unsigned hseed = DROP ? head_seed_of(a, (unsigned)bh) : 0u;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 dq = zero;
  for (int kt = 0; kt < n_tiles; ++kt) {
    const int buf = kt & 1;
    const unsigned word = word_next;
    if (kt + 1 < n_tiles) load_tile(kt + 1);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x16 s = zero, dp = zero;
      ATTN_PRIO_MFMA();
      const float *k0 = &Ks[buf][(half * 32 + r) * kKStride + 16 * h], *v0 = &Vs[buf][(half * 32 + r) * kKStride + 16 * h];
      const float *kc = &Ks[buf][(half * 32 + 4 * h) * kKStride + r];          // dQ^T += K^T dS^T : A = K[key(t, h)][d = r]
#if ATTN_PREFETCH
      float4 ka4[4], va4[4];
      float kq[16];
#pragma unroll
      for (int i = 0; i < 4; ++i) { ka4[i] = ld4(k0 + 4 * i); va4[i] = ld4(v0 + 4 * i); }
#pragma unroll
      for (int t = 0; t < 16; ++t) kq[t] = kc[(8 * (t >> 2) + (t & 3)) * kKStride];
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#if ATTN_PREFETCH
        const float4 ka = ka4[i], va = va4[i];
#else
        const float4 ka = (ATTN_SKIP & 16) ? make_float4(qreg[0], qreg[1], qreg[2], qreg[3]) : ld4(k0 + 4 * i);
        const float4 va = (ATTN_SKIP & 16) ? make_float4(doreg[0], doreg[1], doreg[2], doreg[3]) : ld4(v0 + 4 * i);
#endif
        if (ATTN_SKIP & 8) { s[i] += ka.x + ka.y + ka.z + ka.w; dp[i] += va.x + va.y + va.z + va.w; continue; }
        s = mfma(ka.x, qreg[4 * i], s);     dp = mfma(va.x, doreg[4 * i], dp);
        s = mfma(ka.y, qreg[4 * i + 1], s); dp = mfma(va.y, doreg[4 * i + 1], dp);
        s = mfma(ka.z, qreg[4 * i + 2], s); dp = mfma(va.z, doreg[4 * i + 2], dp);
        s = mfma(ka.w, qreg[4 * i + 3], s); dp = mfma(va.w, doreg[4 * i + 3], dp);
      }
      ATTN_PRIO_VALU();
      const int key0 = kt * kTileK + half * 32 + 4 * h;
      if (MASK) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b4 = ld4(&Bs[buf][half * 32 + 4 * h + 8 * g]);
          s[4 * g] += b4.x; s[4 * g + 1] += b4.y; s[4 * g + 2] += b4.z; s[4 * g + 3] += b4.w;
        }
      }
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int key = key0 + 8 * (v >> 2) + (v & 3);
        float p = (ATTN_SKIP & 1) ? s[v] - lse : __builtin_amdgcn_exp2f(s[v] - lse);       // a masked key: exp2(-inf) = 0
        if (!MASK && key >= a.Lk) p = 0.f;
        float dpe = dp[v];
        if (DROP && !(ATTN_SKIP & 2)) {
          bool keep;
          if (saved_bits) keep = (word & (1u << (16 * half + v))) != 0;
          else {
            const unsigned bits = drop_bits(hseed, (unsigned)q, (unsigned)key >> 1);      // CSE'd across the pair
            keep = ((key & 1) ? (bits >> 16) : (bits & 0xFFFFu)) >= a.drop_thr16;
          }
          dpe = keep ? dpe * a.drop_scale : 0.f;
        }
        s[v] = p * (dpe - delta);                                                          // dS^T
      }
#if ATTN_EARLY_STORE
      if (half == 1 && kt + 1 < n_tiles) store_tile(buf ^ 1);
#endif
      ATTN_PRIO_MFMA();
#if ATTN_OPDEPTH > 0 && !ATTN_PREFETCH
      // the A operands of the second-stage MFMAs ATTN_OPDEPTH steps ahead of their use (an LDS round trip is longer than the one
      // MFMA + ~14 vector instructions the compiler leaves between a read and its MFMA)
      float kop[ATTN_OPDEPTH];
#pragma unroll
      for (int j = 0; j < ATTN_OPDEPTH; ++j) kop[j] = kc[(8 * (j >> 2) + (j & 3)) * kKStride];
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const float aop = kop[t % ATTN_OPDEPTH];
        if (t + ATTN_OPDEPTH < 16) {
          const int u = t + ATTN_OPDEPTH;
          kop[t % ATTN_OPDEPTH] = kc[(8 * (u >> 2) + (u & 3)) * kKStride];
        }
        dq = mfma(aop, s[t], dq);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // this MFMA ...
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // ... then the read for step t + depth, not later
      }
#else
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        if (ATTN_SKIP & 4) { dq[t] += s[t]; continue; }
#if ATTN_PREFETCH
        dq = mfma(kq[t], s[t], dq);
#else
        dq = mfma((ATTN_SKIP & 16) ? qreg[t] : kc[(8 * (t >> 2) + (t & 3)) * kKStride], s[t], dq);
#endif
      }
#endif
    }
#if !ATTN_EARLY_STORE
    if (kt + 1 < n_tiles) store_tile(buf ^ 1);
#endif
    __syncthreads();
  }
  if (q < a.Lq) {
    float *dst = a.dq + b * a.sdq.b + hd * a.sdq.h + (long long)q * a.sdq.t + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4 *>(dst + 8 * g) = make_float4(dq[4 * g] * a.scale, dq[4 * g + 1] * a.scale, dq[4 * g + 2] * a.scale, dq[4 * g + 3] * a.scale);
  }
}

// dK, dV: a lane owns one KEY (S = Q K^T untransposed: col = key, rows = 16 queries); per 32-query half:
// S, dP = dO V^T, dV^T += dO^T P_drop, dK^T += Q^T dS -- 64 MFMAs.  Row statistics (lse, delta) come from LDS.
template <bool DROP, bool MASK = false, bool BITS = false>
__global__ __launch_bounds__(256, ATTN_BWD_WAVES) void bwd_dkdv_kernel(const Args a) {
  __shared__ float Qs[2][kTileK * kKStride];
  __shared__ float Ds[2][kTileK * kKStride];
  __shared__ float Ls[2][kTileK], Es[2][kTileK];          // lse, delta of the tile's queries
  // saved keep bits (Args::keep) of the tile's 64 queries against this workgroup's two 64-key tiles: [key tile][forward lane half][query]
  __shared__ unsigned Ws[2][DROP && BITS ? 256 : 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.H, hd = bh % a.H;
  const int key = blockIdx.x * kBlockQ + wave * 32 + r;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);          // the wave's number as a scalar (for the epilogue)
  const int kc_ = min(key, a.Lk - 1);
  const float *kp = a.k + b * a.sk.b + hd * a.sk.h + (long long)kc_ * a.sk.t + 16 * h;
  const float *vp = a.v + b * a.sv.b + hd * a.sv.h + (long long)kc_ * a.sv.t + 16 * h;
  float kreg[16], vreg[16];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4 t = ld4(kp + 4 * i), u = ld4(vp + 4 * i);
    kreg[4 * i] = t.x * a.scale_log2; kreg[4 * i + 1] = t.y * a.scale_log2;
    kreg[4 * i + 2] = t.z * a.scale_log2; kreg[4 * i + 3] = t.w * a.scale_log2;
    vreg[4 * i] = u.x; vreg[4 * i + 1] = u.y; vreg[4 * i + 2] = u.z; vreg[4 * i + 3] = u.w;
  }
  // MASK: this lane's key takes no part in any softmax -- P = 0 in its whole column, dK = dV = 0
  const bool dead = MASK && a.kmask[(long long)b * a.Lk + kc_] != 0;
  const float *qb = a.q + b * a.sq.b + hd * a.sq.h, *dob = a.dout + b * a.so.b + hd * a.so.h;
  const float *lb = a.lse + (long long)bh * a.Lq, *eb = a.delta + (long long)bh * a.Lq;
  const int n_tiles = (a.Lq + kTileK - 1) / kTileK;
  float4 qr[2], dr[2];
  float lr = 0.f, er = 0.f;
  constexpr bool saved_bits = DROP && BITS;              // the forward's keep bits instead of the hash (the same mask)
  unsigned wr = 0;
  // this lane's key in the forward's layout: key tile, lane half and bit of the word the forward lane of a query wrote
  const int key_in_tile = (wave & 1) * 32 + r;
  const int my_half_f = (r >> 2) & 1, my_bit = 16 * (wave & 1) + 4 * (r >> 3) + (r & 3), my_ktl = wave >> 1;
  const int n_ktiles = (a.Lk + kTileK - 1) / kTileK;
  (void)key_in_tile;
  auto load_tile = [&](int qt) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = threadIdx.x + 256 * j, qq = qt * kTileK + (idx >> 3), c = (idx & 7) * 4;
      const bool ok = qq < a.Lq;
      qr[j] = ok ? ld4(qb + (long long)qq * a.sq.t + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      dr[j] = ok ? ld4(dob + (long long)qq * a.so.t + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (threadIdx.x < kTileK) {
      const int qq = qt * kTileK + threadIdx.x;
      lr = qq < a.Lq ? lb[qq] : INFINITY;               // exp2(s - inf) = 0: padded queries contribute nothing
      er = qq < a.Lq ? eb[qq] : 0.f;
    }
    if (saved_bits) {
      // thread -> (key tile of this workgroup, forward lane half, query of the tile): the word of the forward lane that owned the query
      const int ktl = threadIdx.x >> 7, half_f = (threadIdx.x >> 6) & 1, qq = qt * kTileK + (threadIdx.x & 63);
      const int kt_f = min(2 * (int)blockIdx.x + ktl, n_ktiles - 1);                     // (a key tile past Lk: its lanes store nothing)
      wr = a.keep[keep_index(a, bh, qq >> 7, kt_f, ((qq & 127) >> 5) * 64 + half_f * 32 + (qq & 31))];
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = threadIdx.x + 256 * j, row = idx >> 3, c = (idx & 7) * 4;
      *reinterpret_cast<float4 *>(&Qs[buf][row * kKStride + c]) = qr[j];
      *reinterpret_cast<float4 *>(&Ds[buf][row * kKStride + c]) = dr[j];
    }
    if (threadIdx.x < kTileK) { Ls[buf][threadIdx.x] = lr; Es[buf][threadIdx.x] = er; }
    if (saved_bits) Ws[buf][threadIdx.x] = wr;
  };
  load_tile(0);
  store_tile(0);
  __syncthreads();
  const unsigned hseed = DROP ? head_seed_of(a, (unsigned)bh) : 0u;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 dk = zero, dv = zero;
  for (int qt = 0; qt < n_tiles; ++qt) {
    const int buf = qt & 1;
    if (qt + 1 < n_tiles) load_tile(qt + 1);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x16 s = zero, dp = zero;
      ATTN_PRIO_MFMA();
      const float *q0 = &Qs[buf][(half * 32 + r) * kKStride + 16 * h], *d0 = &Ds[buf][(half * 32 + r) * kKStride + 16 * h];
      const float *dc = &Ds[buf][(half * 32 + 4 * h) * kKStride + r], *qc = &Qs[buf][(half * 32 + 4 * h) * kKStride + r];
#if ATTN_PREFETCH
      float4 qa4[4], da4[4];
      float dcv[16], qcv[16];
#pragma unroll
      for (int i = 0; i < 4; ++i) { qa4[i] = ld4(q0 + 4 * i); da4[i] = ld4(d0 + 4 * i); }
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int row = (8 * (t >> 2) + (t & 3)) * kKStride;
        dcv[t] = dc[row]; qcv[t] = qc[row];
      }
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#if ATTN_PREFETCH
        const float4 qa = qa4[i], da = da4[i];
#else
        const float4 qa = (ATTN_SKIP & 16) ? make_float4(kreg[0], kreg[1], kreg[2], kreg[3]) : ld4(q0 + 4 * i);
        const float4 da = (ATTN_SKIP & 16) ? make_float4(vreg[0], vreg[1], vreg[2], vreg[3]) : ld4(d0 + 4 * i);
#endif
        if (ATTN_SKIP & 8) { s[i] += qa.x + qa.y + qa.z + qa.w; dp[i] += da.x + da.y + da.z + da.w; continue; }
        s = mfma(qa.x, kreg[4 * i], s);     dp = mfma(da.x, vreg[4 * i], dp);
        s = mfma(qa.y, kreg[4 * i + 1], s); dp = mfma(da.y, vreg[4 * i + 1], dp);
        s = mfma(qa.z, kreg[4 * i + 2], s); dp = mfma(da.z, vreg[4 * i + 2], dp);
        s = mfma(qa.w, kreg[4 * i + 3], s); dp = mfma(da.w, vreg[4 * i + 3], dp);
      }
      ATTN_PRIO_VALU();
      const int q0i = qt * kTileK + half * 32 + 4 * h;       // query of register v: q0i + 8 (v >> 2) + (v & 3)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 l4 = ld4(&Ls[buf][half * 32 + 4 * h + 8 * g]), e4 = ld4(&Es[buf][half * 32 + 4 * h + 8 * g]);
        const float ls[4] = {l4.x, l4.y, l4.z, l4.w}, es[4] = {e4.x, e4.y, e4.z, e4.w};
        uint4 w4 = make_uint4(0u, 0u, 0u, 0u);           // the keep words of the four queries of this group
        if (saved_bits) w4 = *reinterpret_cast<const uint4 *>(&Ws[buf][my_ktl * 128 + my_half_f * 64 + half * 32 + 4 * h + 8 * g]);
        const unsigned ws[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int v = 4 * g + j;
          const float p = dead ? 0.f : ((ATTN_SKIP & 1) ? s[v] - ls[j] : __builtin_amdgcn_exp2f(s[v] - ls[j]));
          float pd = p, dpe = dp[v];
          if (DROP && !(ATTN_SKIP & 2)) {
            bool keep;
            if (saved_bits) keep = ((ws[j] >> my_bit) & 1u) != 0;
            else {
              const unsigned bits = drop_bits(hseed, (unsigned)(q0i + 8 * g + j), (unsigned)key >> 1);
              keep = ((key & 1) ? (bits >> 16) : (bits & 0xFFFFu)) >= a.drop_thr16;
            }
            pd = keep ? p * a.drop_scale : 0.f;
            dpe = keep ? dpe * a.drop_scale : 0.f;
          }
          dp[v] = pd;                                          // P_drop   (operand of dV)
          s[v] = p * (dpe - es[j]);                            // dS       (operand of dK)
        }
      }
      ATTN_PRIO_MFMA();
#if ATTN_EARLY_STORE
      if (half == 1 && qt + 1 < n_tiles) store_tile(buf ^ 1);
#endif
#if ATTN_OPDEPTH_KV > 0 && !ATTN_PREFETCH
      float dop[ATTN_OPDEPTH_KV], qop[ATTN_OPDEPTH_KV];      // (as in bwd_dq: the two A operands of step t + depth requested at step t)
#pragma unroll
      for (int j = 0; j < ATTN_OPDEPTH_KV; ++j) {
        dop[j] = dc[(8 * (j >> 2) + (j & 3)) * kKStride];
        qop[j] = qc[(8 * (j >> 2) + (j & 3)) * kKStride];
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const float da_ = dop[t % ATTN_OPDEPTH_KV], qa_ = qop[t % ATTN_OPDEPTH_KV];
        if (t + ATTN_OPDEPTH_KV < 16) {
          const int u = t + ATTN_OPDEPTH_KV, rowu = (8 * (u >> 2) + (u & 3)) * kKStride;
          dop[t % ATTN_OPDEPTH_KV] = dc[rowu];
          qop[t % ATTN_OPDEPTH_KV] = qc[rowu];
        }
        dv = mfma(da_, dp[t], dv);
        dk = mfma(qa_, s[t], dk);
      }
#else
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        if (ATTN_SKIP & 4) { dv[t] += dp[t]; dk[t] += s[t]; continue; }
#if ATTN_PREFETCH
        dv = mfma(dcv[t], dp[t], dv);
        dk = mfma(qcv[t], s[t], dk);
#else
        const int row = (8 * (t >> 2) + (t & 3)) * kKStride;
        dv = mfma((ATTN_SKIP & 16) ? vreg[t] : dc[row], dp[t], dv);
        dk = mfma((ATTN_SKIP & 16) ? kreg[t] : qc[row], s[t], dk);
#endif
      }
#endif
    }
#if !ATTN_EARLY_STORE
    if (qt + 1 < n_tiles) store_tile(buf ^ 1);
#endif
    __syncthreads();
  }
  // (the lane's key and half re-derived from the lane counter and a scalar wave number: kept in a VGPR across the tile loop, `key`
  // was the one register the 170-VGPR build spilled)
  const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int key_e = (int)blockIdx.x * kBlockQ + wave_s * 32 + (lane_e & 31), h_e = lane_e >> 5;
  if (key_e < a.Lk) {
    float *dkp = a.dk + b * a.sdk.b + hd * a.sdk.h + (long long)key_e * a.sdk.t + 4 * h_e;
    float *dvp = a.dv + b * a.sdv.b + hd * a.sdv.h + (long long)key_e * a.sdv.t + 4 * h_e;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      *reinterpret_cast<float4 *>(dkp + 8 * g) = make_float4(dk[4 * g] * a.scale, dk[4 * g + 1] * a.scale, dk[4 * g + 2] * a.scale, dk[4 * g + 3] * a.scale);
      *reinterpret_cast<float4 *>(dvp + 8 * g) = make_float4(dv[4 * g], dv[4 * g + 1], dv[4 * g + 2], dv[4 * g + 3]);
    }
  }
}

}  // namespace attn

namespace {

inline attn::Strides cvt(mono_attn_strides s) { return attn::Strides{s.batch, s.head, s.token}; }

inline void fill_common(attn::Args &a, int H, int Lq, int Lk, float scale, float p, unsigned long long seed) {
  a.H = H; a.Lq = Lq; a.Lk = Lk;
  a.scale = scale;
  a.scale_log2 = scale * 1.4426950408889634f;
  a.drop_thr16 = (unsigned)(p * 65536.0f + 0.5f);
  a.drop_scale = a.drop_thr16 ? 65536.0f / (65536.0f - (float)a.drop_thr16) : 1.f;     // exactly 1 / P(keep)
  a.seed_lo = (unsigned)seed;
  a.seed_hi = (unsigned)(seed >> 32);
}

inline bool aligned16(const void *p, mono_attn_strides s) {
  return ((uintptr_t)p % 16 == 0) && s.batch % 4 == 0 && s.head % 4 == 0 && s.token % 4 == 0;
}

}  // namespace

extern "C" {

int mono_attn_forward_f32(const float *q, const float *k, const float *v, float *o, float *lse, int B, int H, int Lq,
                          int Lk, int head_dim, mono_attn_strides sq, mono_attn_strides sk, mono_attn_strides sv,
                          mono_attn_strides so, float softmax_scale, float dropout_p, unsigned long long seed,
                          void *stream_) {
  return mono_attn_forward_masked_f32(q, k, v, nullptr, o, lse, B, H, Lq, Lk, head_dim, sq, sk, sv, so, softmax_scale, dropout_p,
                                      seed, stream_);
}

int mono_attn_forward_masked_f32(const float *q, const float *k, const float *v, const unsigned char *key_padding_mask, float *o,
                                 float *lse, int B, int H, int Lq, int Lk, int head_dim, mono_attn_strides sq,
                                 mono_attn_strides sk, mono_attn_strides sv, mono_attn_strides so, float softmax_scale,
                                 float dropout_p, unsigned long long seed, void *stream_) {
  return mono_attn_forward_keep_f32(q, k, v, key_padding_mask, nullptr, o, lse, B, H, Lq, Lk, head_dim, sq, sk, sv, so, softmax_scale,
                                    dropout_p, seed, stream_);
}

long long mono_attn_keep_words(int B, int H, int Lq, int Lk) {
  if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return 0;
  return (long long)B * H * ((Lq + 127) / 128) * ((Lk + 63) / 64) * 256;
}

int mono_attn_forward_keep_f32(const float *q, const float *k, const float *v, const unsigned char *key_padding_mask,
                               unsigned *keep_bits, float *o, float *lse, int B, int H, int Lq, int Lk, int head_dim,
                               mono_attn_strides sq, mono_attn_strides sk, mono_attn_strides sv, mono_attn_strides so,
                               float softmax_scale, float dropout_p, unsigned long long seed, void *stream_) {
  if (!q || !k || !v || !o || !lse) return MONO_ATTN_E_NULLPTR;
  if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0 || head_dim != 32 || !(dropout_p >= 0.f && dropout_p < 1.f) ||
      (long long)B * H > 65535)
    return MONO_ATTN_E_SHAPE;
  if (!aligned16(q, sq) || !aligned16(k, sk) || !aligned16(v, sv) || !aligned16(o, so)) return MONO_ATTN_E_SHAPE;
  attn::Args a{};
  a.q = q; a.k = k; a.v = v; a.o = o; a.lse = lse;
  a.sq = cvt(sq); a.sk = cvt(sk); a.sv = cvt(sv); a.so = cvt(so);
  fill_common(a, H, Lq, Lk, softmax_scale, dropout_p, seed);
  const dim3 grid((Lq + attn::kBlockQ - 1) / attn::kBlockQ, B * H);
  hipStream_t st = (hipStream_t)stream_;
  a.kmask = key_padding_mask;
  a.keep = a.drop_thr16 ? keep_bits : nullptr;
  if (key_padding_mask) {
    if (a.keep) attn::fwd_kernel<true, true, true><<<grid, 256, 0, st>>>(a);
    else if (a.drop_thr16) attn::fwd_kernel<true, true><<<grid, 256, 0, st>>>(a);
    else attn::fwd_kernel<false, true><<<grid, 256, 0, st>>>(a);
  } else if (a.keep) attn::fwd_kernel<true, false, true><<<grid, 256, 0, st>>>(a);
  else if (a.drop_thr16) attn::fwd_kernel<true><<<grid, 256, 0, st>>>(a);
  else attn::fwd_kernel<false><<<grid, 256, 0, st>>>(a);
  return (int)hipGetLastError();
}


int mono_attn_backward_f32(const float *q, const float *k, const float *v, const float *o, const float *lse,
                           const float *dout, float *dq, float *dk, float *dv, float *delta, int B, int H, int Lq,
                           int Lk, int head_dim, mono_attn_strides sq, mono_attn_strides sk, mono_attn_strides sv,
                           mono_attn_strides so, mono_attn_strides sdq, mono_attn_strides sdk, mono_attn_strides sdv,
                           float softmax_scale, float dropout_p, unsigned long long seed, void *stream_) {
  return mono_attn_backward_masked_f32(q, k, v, nullptr, o, lse, dout, dq, dk, dv, delta, B, H, Lq, Lk, head_dim, sq, sk, sv, so, sdq,
                                       sdk, sdv, softmax_scale, dropout_p, seed, stream_);
}

int mono_attn_backward_masked_f32(const float *q, const float *k, const float *v, const unsigned char *key_padding_mask,
                                  const float *o, const float *lse, const float *dout, float *dq, float *dk, float *dv, float *delta,
                                  int B, int H, int Lq, int Lk, int head_dim, mono_attn_strides sq, mono_attn_strides sk,
                                  mono_attn_strides sv, mono_attn_strides so, mono_attn_strides sdq, mono_attn_strides sdk,
                                  mono_attn_strides sdv, float softmax_scale, float dropout_p, unsigned long long seed,
                                  void *stream_) {
  return mono_attn_backward_keep_f32(q, k, v, key_padding_mask, nullptr, o, lse, dout, dq, dk, dv, delta, B, H, Lq, Lk, head_dim, sq, sk,
                                     sv, so, sdq, sdk, sdv, softmax_scale, dropout_p, seed, stream_);
}

int mono_attn_backward_keep_f32(const float *q, const float *k, const float *v, const unsigned char *key_padding_mask,
                                const unsigned *keep_bits, const float *o, const float *lse, const float *dout, float *dq, float *dk,
                                float *dv, float *delta, int B, int H, int Lq, int Lk, int head_dim, mono_attn_strides sq,
                                mono_attn_strides sk, mono_attn_strides sv, mono_attn_strides so, mono_attn_strides sdq,
                                mono_attn_strides sdk, mono_attn_strides sdv, float softmax_scale, float dropout_p,
                                unsigned long long seed, void *stream_) {
  if (!q || !k || !v || !o || !lse || !dout || !dq || !dk || !dv || !delta) return MONO_ATTN_E_NULLPTR;
  if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0 || head_dim != 32 || !(dropout_p >= 0.f && dropout_p < 1.f) ||
      (long long)B * H > 65535)
    return MONO_ATTN_E_SHAPE;
  if (!aligned16(q, sq) || !aligned16(k, sk) || !aligned16(v, sv) || !aligned16(o, so) || !aligned16(dout, so) ||
      !aligned16(dq, sdq) || !aligned16(dk, sdk) || !aligned16(dv, sdv))
    return MONO_ATTN_E_SHAPE;
  attn::Args a{};
  a.q = q; a.k = k; a.v = v; a.o = const_cast<float *>(o); a.lse = const_cast<float *>(lse);
  a.dout = dout; a.delta = delta; a.dq = dq; a.dk = dk; a.dv = dv;
  a.sq = cvt(sq); a.sk = cvt(sk); a.sv = cvt(sv); a.so = cvt(so);
  a.sdq = cvt(sdq); a.sdk = cvt(sdk); a.sdv = cvt(sdv);
  fill_common(a, H, Lq, Lk, softmax_scale, dropout_p, seed);
  hipStream_t st = (hipStream_t)stream_;
  const dim3 gq((Lq + attn::kBlockQ - 1) / attn::kBlockQ, B * H), gk((Lk + attn::kBlockQ - 1) / attn::kBlockQ, B * H);
  a.kmask = key_padding_mask;
  a.keep = a.drop_thr16 ? const_cast<unsigned *>(keep_bits) : nullptr;
  if (key_padding_mask && a.keep) {
    attn::bwd_dq_kernel<true, true, true><<<gq, 256, 0, st>>>(a);
    attn::bwd_dkdv_kernel<true, true, true><<<gk, 256, 0, st>>>(a);
  } else if (a.keep) {
    attn::bwd_dq_kernel<true, false, true><<<gq, 256, 0, st>>>(a);
    attn::bwd_dkdv_kernel<true, false, true><<<gk, 256, 0, st>>>(a);
  } else if (key_padding_mask && a.drop_thr16) {
    attn::bwd_dq_kernel<true, true><<<gq, 256, 0, st>>>(a);
    attn::bwd_dkdv_kernel<true, true><<<gk, 256, 0, st>>>(a);
  } else if (key_padding_mask) {
    attn::bwd_dq_kernel<false, true><<<gq, 256, 0, st>>>(a);
    attn::bwd_dkdv_kernel<false, true><<<gk, 256, 0, st>>>(a);
  } else if (a.drop_thr16) {
    attn::bwd_dq_kernel<true><<<gq, 256, 0, st>>>(a);
    attn::bwd_dkdv_kernel<true><<<gk, 256, 0, st>>>(a);
  } else {
    attn::bwd_dq_kernel<false><<<gq, 256, 0, st>>>(a);
    attn::bwd_dkdv_kernel<false><<<gk, 256, 0, st>>>(a);
  }
  return (int)hipGetLastError();
}

}  // extern "C"
