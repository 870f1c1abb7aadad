// libmonosowa_gemm.so: hipBLASLt f32 GEMMs with epilogues behind a C-ABI (include/monosowa_gemm.h).
//
// Dense contractions stay on the library's MFMA kernels (the same Tensile kernels PyTorch's addmm reaches: Cijk_..._Bias_HA_S_SAV_...);
// this file only asks them for what they can already do and the PyTorch front end does not expose: a per-row alpha vector (frozen-BN
// scale), bias, ReLU and a beta * C residual in the epilogue, and the bias gradient as a by-product of the weight-gradient GEMM.
// Row-major callers are mapped onto the library's column-major convention by computing the transposed product.
// Kernel choice: the library's heuristic list, timed on first use per problem key (mono_gemm_set_autotune).
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "../../include/monosowa_gemm.h"

namespace {

constexpr size_t kWorkspaceBytes = 64u << 20;

struct DeviceState {
  hipblasLtHandle_t handle = nullptr;
  void *workspace = nullptr;
};

std::mutex g_mutex;
std::map<int, DeviceState> g_devices;
int g_autotune = 32;

// (kind, m, n, k, lda, ldb, ldc, ldd, flags) in the library's column-major terms
typedef std::tuple<int, int, int, int, long long, long long, long long, long long, int> Key;
std::map<std::pair<int, Key>, hipblasLtMatmulAlgo_t> g_algos;      // (device, key) -> chosen kernel

int device_state(DeviceState &out) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  auto it = g_devices.find(dev);
  if (it == g_devices.end()) {
    DeviceState s;
    hipblasStatus_t st = hipblasLtCreate(&s.handle);
    if (st != HIPBLAS_STATUS_SUCCESS) return (int)st == 0 ? 1 : (int)st;
    e = hipMalloc(&s.workspace, kWorkspaceBytes);
    if (e != hipSuccess) {                       // nothing half-made stays behind: the next call starts over
      hipblasLtDestroy(s.handle);
      return (int)e;
    }
    it = g_devices.emplace(dev, s).first;
  }
  out = it->second;
  return 0;
}

struct Problem {
  // column-major problem  Dc[m x n] = epilogue(alpha (x) op(Ac)[m x k] . op(Bc)[k x n] + beta Cc)
  int kind;                       // cache key component
  hipblasOperation_t ta, tb;
  int m, n, k;
  const float *a, *b, *c;
  float *d;
  long long lda, ldb, ldc, ldd;   // leading dimensions of the STORED matrices
  const float *alpha_vec;         // per row of Dc (NULL: alpha = 1)
  float beta;
  const float *bias;              // per row of Dc (epilogue input), or the bias-gradient output
  uint32_t epilogue;
};

#define LT_CHECK(x)                              \
  do {                                           \
    hipblasStatus_t st_ = (x);                   \
    if (st_ != HIPBLAS_STATUS_SUCCESS) {         \
      rc = (int)st_ == 0 ? 1 : (int)st_;         \
      goto done;                                 \
    }                                            \
  } while (0)

int run(const Problem &p, hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_mutex);
  DeviceState ds;
  int rc = device_state(ds);
  if (rc) return rc;
  int dev = 0;
  (void)hipGetDevice(&dev);
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr, ld = nullptr;
  hipblasLtMatmulPreference_t pref = nullptr;
  const float one = 1.f;
  const void *alpha_ptr = &one;
  {
    LT_CHECK(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    LT_CHECK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &p.ta, sizeof(p.ta)));
    LT_CHECK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &p.tb, sizeof(p.tb)));
    LT_CHECK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &p.epilogue, sizeof(p.epilogue)));
    if (p.bias) {
      const void *bp = p.bias;
      const int32_t bt = (int32_t)HIP_R_32F;
      LT_CHECK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bp, sizeof(bp)));
      LT_CHECK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)));
    }
    if (p.alpha_vec) {
      const int32_t mode = (int32_t)HIPBLASLT_POINTER_MODE_ALPHA_DEVICE_VECTOR_BETA_HOST;
      LT_CHECK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_POINTER_MODE, &mode, sizeof(mode)));
      alpha_ptr = p.alpha_vec;
    }
    const bool a_t = p.ta != HIPBLAS_OP_N, b_t = p.tb != HIPBLAS_OP_N;
    LT_CHECK(hipblasLtMatrixLayoutCreate(&la, HIP_R_32F, a_t ? p.k : p.m, a_t ? p.m : p.k, p.lda));
    LT_CHECK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_32F, b_t ? p.n : p.k, b_t ? p.k : p.n, p.ldb));
    LT_CHECK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_32F, p.m, p.n, p.ldc));
    LT_CHECK(hipblasLtMatrixLayoutCreate(&ld, HIP_R_32F, p.m, p.n, p.ldd));

    const int flags = (int)p.epilogue * 4 + (p.alpha_vec ? 1 : 0) + (p.beta != 0.f ? 2 : 0);
    const auto key = std::make_pair(dev, Key(p.kind, p.m, p.n, p.k, p.lda, p.ldb, p.ldc, p.ldd, flags));
    auto it = g_algos.find(key);
    if (it == g_algos.end()) {
      LT_CHECK(hipblasLtMatmulPreferenceCreate(&pref));
      const uint64_t ws = kWorkspaceBytes;
      LT_CHECK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws, sizeof(ws)));
      const bool aliased = p.beta != 0.f && (const float *)p.d == p.c;       // repeated launches would accumulate: no timing
      const int want = (g_autotune > 1 && !aliased) ? g_autotune : 1;
      std::vector<hipblasLtMatmulHeuristicResult_t> res(want);
      int got = 0;
      LT_CHECK(hipblasLtMatmulAlgoGetHeuristic(ds.handle, desc, la, lb, lc, ld, pref, want, res.data(), &got));
      int best = -1;
      float best_ms = 0.f;
      if (got > 1) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        for (int i = 0; i < got; ++i) {
          if (res[i].state != HIPBLAS_STATUS_SUCCESS || res[i].workspaceSize > kWorkspaceBytes) continue;
          bool ok = true;
          float ms = 0.f;
          for (int r = 0; r < 4 && ok; ++r) {                   // 1 warm-up + 3 timed
            if (r == 1) (void)hipEventRecord(e0, stream);
            ok = hipblasLtMatmul(ds.handle, desc, alpha_ptr, p.a, la, p.b, lb, &p.beta, p.c ? p.c : p.d, lc, p.d, ld, &res[i].algo,
                                 ds.workspace, kWorkspaceBytes, stream) == HIPBLAS_STATUS_SUCCESS;
          }
          (void)hipEventRecord(e1, stream);
          if (hipEventSynchronize(e1) != hipSuccess) ok = false;
          if (ok) (void)hipEventElapsedTime(&ms, e0, e1);
          if (ok && (best < 0 || ms < best_ms)) { best = i; best_ms = ms; }
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
      } else if (got == 1 && res[0].state == HIPBLAS_STATUS_SUCCESS) {
        best = 0;
      }
      if (best < 0) { rc = MONO_GEMM_E_NO_ALGO; goto done; }
      it = g_algos.emplace(key, res[best].algo).first;
    }
    {
      const hipblasStatus_t st = hipblasLtMatmul(ds.handle, desc, alpha_ptr, p.a, la, p.b, lb, &p.beta, p.c ? p.c : p.d, lc, p.d, ld, &it->second,
                                                 ds.workspace, kWorkspaceBytes, stream);
      if (st != HIPBLAS_STATUS_SUCCESS) {
        g_algos.erase(it);                       // (a kernel that refuses these operands is not kept for the key)
        rc = (int)st == 0 ? 1 : (int)st;
        goto done;
      }
    }
  }
done:
  if (pref) hipblasLtMatmulPreferenceDestroy(pref);
  if (la) hipblasLtMatrixLayoutDestroy(la);
  if (lb) hipblasLtMatrixLayoutDestroy(lb);
  if (lc) hipblasLtMatrixLayoutDestroy(lc);
  if (ld) hipblasLtMatrixLayoutDestroy(ld);
  if (desc) hipblasLtMatmulDescDestroy(desc);
  return rc;
}

inline bool misaligned(const void *p) { return ((uintptr_t)p & 15) != 0; }

}  // namespace

extern "C" {

int mono_gemm_nt_epilogue_f32(const float *A, long long lda, const float *W, long long ldw, const float *C, long long ldc, float *D,
                              long long ldd, int M, int N, int K, const float *scale, float beta, const float *bias, int relu,
                              void *stream) {
  if (!A || !W || !D) return MONO_GEMM_E_NULLPTR;
  if (M <= 0 || N <= 0 || K <= 0 || lda < K || ldw < K || ldd < N || (C && ldc < N)) return MONO_GEMM_E_SHAPE;
  if (misaligned(A) || misaligned(W) || misaligned(D) || misaligned(C) || misaligned(scale) || misaligned(bias)) return MONO_GEMM_E_SHAPE;
  // row-major D[M, N] = A W^T   <=>   column-major D^T[N x M] = W[N x K] A^T[K x M]: Ac = W stored [K x N] (op T), Bc = A stored [K x M] (op N)
  Problem p{};
  p.kind = 0;
  p.ta = HIPBLAS_OP_T; p.tb = HIPBLAS_OP_N;
  p.m = N; p.n = M; p.k = K;
  p.a = W; p.lda = ldw;
  p.b = A; p.ldb = lda;
  p.c = (C && beta != 0.f) ? C : nullptr; p.ldc = p.c ? ldc : ldd;
  p.d = D; p.ldd = ldd;
  p.alpha_vec = scale;
  p.beta = p.c ? beta : 0.f;
  p.bias = bias;
  p.epilogue = bias ? (relu ? HIPBLASLT_EPILOGUE_RELU_BIAS : HIPBLASLT_EPILOGUE_BIAS) : (relu ? HIPBLASLT_EPILOGUE_RELU : HIPBLASLT_EPILOGUE_DEFAULT);
  return run(p, (hipStream_t)stream);
}

int mono_gemm_tn_bgrad_f32(const float *dY, long long lddy, const float *X, long long ldx, float *dW, long long lddw, float *dbias,
                           int M, int N, int K, void *stream) {
  if (!dY || !X || !dW) return MONO_GEMM_E_NULLPTR;
  if (M <= 0 || N <= 0 || K <= 0 || lddy < N || ldx < K || lddw < K) return MONO_GEMM_E_SHAPE;
  if (misaligned(dY) || misaligned(X) || misaligned(dW) || misaligned(dbias)) return MONO_GEMM_E_SHAPE;
  // row-major dW[N, K] = dY^T X   <=>   column-major dW^T[K x N] = X^T[K x M] dY[M x N]: Ac = X stored [K x M] (op N), Bc = dY stored
  // [N x M] (op T); the bias gradient is the reduction of op(Bc) over the inner dimension: one value per column of the result = per n
  Problem p{};
  p.kind = 1;
  p.ta = HIPBLAS_OP_N; p.tb = HIPBLAS_OP_T;
  p.m = K; p.n = N; p.k = M;
  p.a = X; p.lda = ldx;
  p.b = dY; p.ldb = lddy;
  p.c = nullptr; p.ldc = lddw;
  p.d = dW; p.ldd = lddw;
  p.beta = 0.f;
  p.bias = dbias;
  p.epilogue = dbias ? HIPBLASLT_EPILOGUE_BGRADB : HIPBLASLT_EPILOGUE_DEFAULT;
  return run(p, (hipStream_t)stream);
}

int mono_gemm_nn_f32(const float *dY, long long lddy, const float *W, long long ldw, float *dX, long long lddx, int M, int N, int K,
                     void *stream) {
  if (!dY || !W || !dX) return MONO_GEMM_E_NULLPTR;
  if (M <= 0 || N <= 0 || K <= 0 || lddy < N || ldw < K || lddx < K) return MONO_GEMM_E_SHAPE;
  if (misaligned(dY) || misaligned(W) || misaligned(dX)) return MONO_GEMM_E_SHAPE;
  // row-major dX[M, K] = dY W   <=>   column-major dX^T[K x M] = W^T[K x N] dY^T[N x M]: Ac = W stored [K x N] (op N), Bc = dY stored [N x M] (op N)
  Problem p{};
  p.kind = 2;
  p.ta = HIPBLAS_OP_N; p.tb = HIPBLAS_OP_N;
  p.m = K; p.n = M; p.k = N;
  p.a = W; p.lda = ldw;
  p.b = dY; p.ldb = lddy;
  p.c = nullptr; p.ldc = lddx;
  p.d = dX; p.ldd = lddx;
  p.beta = 0.f;
  p.epilogue = HIPBLASLT_EPILOGUE_DEFAULT;
  return run(p, (hipStream_t)stream);
}

int mono_gemm_set_autotune(int n) {
  std::lock_guard<std::mutex> lock(g_mutex);
  const int prev = g_autotune;
  g_autotune = n;
  return prev;
}

int mono_gemm_cache_size(void) {
  std::lock_guard<std::mutex> lock(g_mutex);
  return (int)g_algos.size();
}

}  // extern "C"
