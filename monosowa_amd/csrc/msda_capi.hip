// C-ABI entry points (include/monosowa_msda.h): argument checks, kernel selection, launch.
// Replaces the reference host launchers ms_deform_attn_cuda_forward / _backward
// (ops/src/cuda/ms_deform_attn_cuda.cu:20-153).  The reference slices the batch into
// `im2col_step` chunks only to bound its 32-bit indexing; indexing here is 64-bit, so one
// launch covers the whole batch and the chunking precondition lives in the Python shim.
#include "../../include/monosowa_msda.h"
#include "msda_kernels.hip"

namespace {

inline int grid_for(long long work_items, int items_per_block) {
  long long g = (work_items + items_per_block - 1) / items_per_block;
  const long long cap = 256LL * 64;   // 256 CUs x 64 resident-or-queued blocks; grid-stride beyond
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

inline int check_dims(int B, int S, int M, int D, int L, int Lq, int P) {
  if (B <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq <= 0 || P <= 0) return MSDA_E_SHAPE;
  // per-sample offsets are kept in 32 bits inside the d32 kernel
  if ((long long)S * M * D >= (1LL << 31)) return MSDA_E_SHAPE;
  return 0;
}

template <typename T>
int forward_impl(const T *value, const int64_t *shapes, const int64_t *lsi, const T *loc,
                 const T *attw, T *out, int B, int S, int M, int D, int L, int Lq, int P,
                 void *stream_) {
  if (!value || !shapes || !lsi || !loc || !attw || !out) return MSDA_E_NULLPTR;
  if (int e = check_dims(B, S, M, D, L, Lq, P)) return e;
  hipStream_t stream = (hipStream_t)stream_;
  const long long n_pairs = (long long)B * Lq * M;
  if constexpr (sizeof(T) == 4) {
    if (D == 32 && L == 4 && P == 4) {
      msda::fwd_d32_kernel<4, 4><<<grid_for(n_pairs, 32), 256, 0, stream>>>(
          value, shapes, lsi, loc, attw, out, S, M, Lq, n_pairs);
      return (int)hipGetLastError();
    }
  }
  msda::fwd_generic_kernel<T><<<grid_for(n_pairs, 8), 256, 0, stream>>>(
      value, shapes, lsi, loc, attw, out, S, M, D, L, Lq, P, n_pairs);
  return (int)hipGetLastError();
}

template <typename T>
int backward_impl(const T *value, const int64_t *shapes, const int64_t *lsi, const T *loc,
                  const T *attw, const T *grad_out, T *grad_value, T *grad_loc, T *grad_attw,
                  int B, int S, int M, int D, int L, int Lq, int P, void *, size_t, void *stream_) {
  if (!value || !shapes || !lsi || !loc || !attw || !grad_out || !grad_value || !grad_loc || !grad_attw)
    return MSDA_E_NULLPTR;
  if (int e = check_dims(B, S, M, D, L, Lq, P)) return e;
  hipStream_t stream = (hipStream_t)stream_;
  const long long n_pairs = (long long)B * Lq * M;
  hipError_t err = hipMemsetAsync(grad_value, 0, sizeof(T) * (size_t)B * S * M * D, stream);
  if (err != hipSuccess) return (int)err;
  msda::bwd_generic_kernel<T><<<grid_for(n_pairs, 8), 256, 0, stream>>>(
      value, shapes, lsi, loc, attw, grad_out, grad_value, grad_loc, grad_attw, S, M, D, L, Lq, P,
      n_pairs);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" {

int msda_abi_version(void) { return MSDA_ABI_VERSION; }

const char *msda_strerror(int code) {
  switch (code) {
    case 0: return "success";
    case MSDA_E_NULLPTR: return "msda: a required pointer is NULL";
    case MSDA_E_SHAPE: return "msda: a dimension is <= 0 or exceeds the indexing range";
    case MSDA_E_UNSUPPORTED: return "msda: unsupported configuration";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "msda: unknown error";
  }
}

size_t msda_backward_workspace_bytes(int, int, int, int, int, int, int, int) { return 0; }

int msda_forward_f32(const float *value, const int64_t *shapes, const int64_t *level_start,
                     const float *loc, const float *attn_w, float *out, int B, int S, int M, int D,
                     int L, int Lq, int P, void *stream) {
  return forward_impl<float>(value, shapes, level_start, loc, attn_w, out, B, S, M, D, L, Lq, P, stream);
}

int msda_forward_f64(const double *value, const int64_t *shapes, const int64_t *level_start,
                     const double *loc, const double *attn_w, double *out, int B, int S, int M, int D,
                     int L, int Lq, int P, void *stream) {
  return forward_impl<double>(value, shapes, level_start, loc, attn_w, out, B, S, M, D, L, Lq, P, stream);
}

int msda_backward_f32(const float *value, const int64_t *shapes, const int64_t *level_start,
                      const float *loc, const float *attn_w, const float *grad_out, float *grad_value,
                      float *grad_loc, float *grad_attn_w, int B, int S, int M, int D, int L, int Lq,
                      int P, void *workspace, size_t workspace_bytes, void *stream) {
  return backward_impl<float>(value, shapes, level_start, loc, attn_w, grad_out, grad_value, grad_loc,
                              grad_attn_w, B, S, M, D, L, Lq, P, workspace, workspace_bytes, stream);
}

int msda_backward_f64(const double *value, const int64_t *shapes, const int64_t *level_start,
                      const double *loc, const double *attn_w, const double *grad_out,
                      double *grad_value, double *grad_loc, double *grad_attn_w, int B, int S, int M,
                      int D, int L, int Lq, int P, void *workspace, size_t workspace_bytes,
                      void *stream) {
  return backward_impl<double>(value, shapes, level_start, loc, attn_w, grad_out, grad_value, grad_loc,
                               grad_attn_w, B, S, M, D, L, Lq, P, workspace, workspace_bytes, stream);
}

}  // extern "C"
