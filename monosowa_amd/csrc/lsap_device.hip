// Rectangular linear sum assignment ON THE DEVICE: the Hungarian matcher's 528 problems per step (3 decoder layers x 16 images x
// 11 query groups, 50 queries x <= 50 targets each; reference matcher.py:94-103 calls scipy.optimize.linear_sum_assignment for each)
// without the device -> host copy of the cost blocks, the host wait and the copy back -- the train step's ONE host synchronisation
// (bench: 1.7 - 2.9 ms of a 71 ms step: behind the wait the GPU queue is empty and every launch of the criterion tail and the
// decoder backward is paid at host speed).
//
// Same algorithm, same arithmetic, same tie-breaks as csrc/lsap.cpp (the shortest-augmenting-path method of Crouse 2016, which is
// what scipy implements): double duals, the reduced cost  r = ((min_val + c) - u[i]) - v[j]  rounded operation by operation, the
// remaining-column list in scipy's order (initialised descending, a used column is replaced by the list's last one), and the
// choice among equal shortest paths: scanning the list in order, a strictly shorter path always wins and an equal one wins iff
// its column is unassigned -- i.e. the LAST unassigned column among the minima if there is one, else the FIRST minimum.  One
// wavefront per problem: lane p owns list positions p, p + 64, ...; the scan is data-parallel, the selection a wave reduction
// (butterfly shuffles: one double minimum, one packed integer maximum), everything else wave-uniform.  All tables live in LDS (the cost block as float:
// the float -> double conversion is exact), so the augmenting loop never touches global memory.
//
// tests/test_lsap_device.py compares the assignments with csrc/lsap.cpp and scipy on random, tied, rectangular and degenerate
// problems.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lsapd {

constexpr int kMaxDim = 128;             // rows / columns of one problem (the list positions of a lane: kMaxDim / 64)
constexpr int kMaxCells = 8192;          // rows x columns kept in LDS as float (32 KB)

__device__ __forceinline__ double wave_min_f64(double v) {
  // butterfly over the 64 lanes: ds_swizzle-free shuffles (6 steps); every lane ends with the minimum
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
    const double o = __shfl_xor(v, m, 64);
    v = o < v ? o : v;
  }
  return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v = max(v, __shfl_xor(v, m, 64));
  return v;
}
// LDS hand-offs between the lanes of the one wave (lane 0 writes, every lane reads): the LDS executes a wave's instructions in
// order, only the compiler has to be kept from moving accesses across
__device__ __forceinline__ void order() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

// Problem (layer l, image b, group g): queries [g * gq, (g + 1) * gq) of cost[l, b, :, 0 .. sizes[b]) (row stride T).
// meta [3, B] int32: sizes | first (output position of image b's pairs within a layer) | toff (targets of the images before b).
// out_idx [3, NL, K] int64 = (image, query, target + toff) per pair, (image, group) order, pairs of a group sorted by query --
// exactly lsap_match_flat_f32 (csrc/lsap.cpp).  status: bit 0 set when a cost is NaN / -inf or a problem is infeasible
// (scipy raises ValueError; the caller reads the flag without stalling the step), bit 1 when a problem exceeds the LDS tables.
__global__ __launch_bounds__(64) void match_flat_kernel(const float *__restrict__ cost, int NL, int B, int Q, int T, int G,
                                                        const int *__restrict__ meta, long long *__restrict__ out_idx, long long K,
                                                        int *__restrict__ status) {
  __shared__ float costf[kMaxCells];
  __shared__ double u[kMaxDim], v[kMaxDim], shortest[kMaxDim];
  __shared__ int path[kMaxDim], col4row[kMaxDim], row4col[kMaxDim], remaining[kMaxDim];
  __shared__ unsigned char SR[kMaxDim], SC[kMaxDim];
  const int lane = threadIdx.x;
  const int g = blockIdx.x % G, b = (blockIdx.x / G) % B, l = blockIdx.x / (G * B);
  const int gq = Q / G, n = meta[b], first = meta[B + b], toff = meta[2 * B + b];
  if (n <= 0) return;
  const bool transpose = gq > n;
  const int R = transpose ? n : gq, C = transpose ? gq : n;               // R <= C
  const int k_pairs = R;
  long long *ob = out_idx + (long long)l * K + first + (long long)g * k_pairs, *oq = ob + (long long)NL * K, *ot = oq + (long long)NL * K;
  if (C > kMaxDim || R * C > kMaxCells) {
    // beyond the tables (the Python wrapper keeps such shapes on the host solver): identity pairs, flagged
    if (lane == 0) atomicOr(status, 2);
    for (int i = lane; i < k_pairs; i += 64) { ob[i] = b; oq[i] = g * gq + i; ot[i] = toff + i; }
    return;
  }
  // ---- the cost block, in the solver's orientation (rows = the smaller side) ---------------------------------------------------
  const float *c = cost + (((long long)l * B + b) * Q + (long long)g * gq) * T;
  bool bad = false;
  for (int e = lane; e < R * C; e += 64) {
    const int i = e / C, j = e - i * C;
    const float x = transpose ? c[(long long)j * T + i] : c[(long long)i * T + j];
    bad |= (x != x) || (x == -__builtin_inff());
    costf[e] = x;
  }
  for (int i = lane; i < R; i += 64) { u[i] = 0.0; col4row[i] = -1; }
  for (int j = lane; j < C; j += 64) { v[j] = 0.0; path[j] = -1; row4col[j] = -1; }
  bool failed = __any(bad);
  order();

  const double inf = __builtin_inf();
  for (int cur = 0; cur < R && !failed; ++cur) {
    double min_val = 0.0;
    int num_remaining = C;
    for (int it = lane; it < C; it += 64) { remaining[it] = C - it - 1; shortest[it] = inf; SC[it] = 0; }
    for (int i = lane; i < R; i += 64) SR[i] = 0;
    order();
    int sink = -1, i = cur;
    while (sink == -1) {
      if (lane == 0) SR[i] = 1;
      const double ui = u[i];
      // this lane's part of the scan, in list order: local minimum, first position attaining it, last UNASSIGNED position attaining it
      double lowest = inf;
      int first_it = 0x7fffffff, last_free = -1;
      for (int it = lane; it < num_remaining; it += 64) {
        const int j = remaining[it];
        const double r = ((min_val + (double)costf[i * C + j]) - ui) - v[j];
        double s = shortest[j];
        if (r < s) { path[j] = i; shortest[j] = r; s = r; }
        const bool free_col = row4col[j] == -1;
        if (s < lowest) { lowest = s; first_it = it; last_free = free_col ? it : -1; }
        else if (s == lowest) { if (it < first_it) first_it = it; if (free_col) last_free = it; }
      }
      const double m = wave_min_f64(lowest);
      if (!(m < inf)) { failed = true; break; }                            // infeasible (wave-uniform)
      // one max-reduction decides: any unassigned minimum beats every assigned one (bit 16), the largest position among the
      // unassigned ones, the smallest among the assigned ones
      const int key = lowest == m ? (last_free >= 0 ? (0x10000 | last_free) : (0xFFFF - first_it)) : -1;
      const int best = wave_max_i32(key);
      const int index = (best & 0x10000) ? (best & 0xFFFF) : (0xFFFF - best);
      min_val = m;
      order();
      const int j = remaining[index];
      const int owner = row4col[j];
      if (owner == -1) sink = j; else i = owner;
      order();
      if (lane == 0) { SC[j] = 1; remaining[index] = remaining[num_remaining - 1]; }
      --num_remaining;
      order();
    }
    if (failed) break;
    // ---- dual updates (lsap.cpp: u[cur] += min_val; u[r] += min_val - shortest[col4row[r]]; v[j] -= min_val - shortest[j]) --------
    for (int r = lane; r < R; r += 64) {
      if (r == cur) u[r] = u[r] + min_val;
      else if (SR[r]) u[r] = u[r] + (min_val - shortest[col4row[r]]);
    }
    for (int j = lane; j < C; j += 64)
      if (SC[j]) v[j] = v[j] - (min_val - shortest[j]);
    order();
    // ---- augment along the path (serial; every lane walks it, lane 0 writes) -----------------------------------------------------
    int j = sink;
    while (true) {
      const int r = path[j];
      const int prev = col4row[r];
      order();
      if (lane == 0) { row4col[j] = r; col4row[r] = j; }
      order();
      j = prev;
      if (r == cur) break;
    }
  }
  if (failed) {
    if (lane == 0) atomicOr(status, 1);
    for (int i = lane; i < k_pairs; i += 64) { ob[i] = b; oq[i] = g * gq + i; ot[i] = toff + i; }      // valid indices; the flag says the rest
    return;
  }
  // ---- pairs sorted by query (lsap.cpp: solve_strided) -----------------------------------------------------------------------------
  if (!transpose) {
    for (int i = lane; i < R; i += 64) { ob[i] = b; oq[i] = g * gq + i; ot[i] = toff + col4row[i]; }
  } else {
    int base = 0;                                                          // queries = the solver's columns, in increasing order
    for (int j0 = 0; j0 < C; j0 += 64) {
      const int j = j0 + lane;
      const int r = j < C ? row4col[j] : -1;
      const unsigned long long m = __ballot(r != -1);
      if (r != -1) {
        const int k = base + __popcll(m & ((1ull << lane) - 1ull));
        ob[k] = b; oq[k] = g * gq + j; ot[k] = toff + r;
      }
      base += __popcll(m);
    }
  }
}

}  // namespace lsapd
