/* C-ABI of libmonosowa_gemm.so: f32 library GEMMs (hipBLASLt) with the epilogues MonoDETR's dense layers end in.
 *
 * Row a10 / a3 of SURVEY.md section 8: the ResNet bottlenecks' 1 x 1 convolutions (reference backbone.py:28-115 -> torchvision
 * Bottleneck: conv -> frozen BN -> (+ identity) -> ReLU) and the nn.Linear layers of the transformer are DENSE contractions; they stay
 * on the library's MFMA kernels.  What this shim adds is the part PyTorch's front end cannot ask the library for: the per-channel
 * scale (frozen BN's scale), shift, residual and ReLU evaluated in the GEMM's epilogue, and a bias gradient produced by the
 * weight-gradient GEMM itself -- passes over the activations that otherwise run as separate kernels.
 *
 * All matrices are ROW-major f32 in device memory, leading dimensions in elements, 16-byte aligned.  Every call is asynchronous on
 * `stream`.  Returns 0, a negative MONO_GEMM_E_* code, or a positive hipblasStatus_t / hipError_t.  No CPU path.
 * State: one hipBLASLt handle and one 64 MiB split-K workspace per device, created on first use and kept for the life of the process;
 * host calls are serialised by a mutex, but launches on DIFFERENT streams of one device share that workspace and must not overlap
 * (MonoDETR's step issues all of them on one stream).  The first call of a problem key times candidates (host-synchronising, see
 * mono_gemm_set_autotune): it must happen outside a stream capture. */
#ifndef MONOSOWA_GEMM_H
#define MONOSOWA_GEMM_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MONO_GEMM_E_NULLPTR (-1)
#define MONO_GEMM_E_SHAPE (-2)
#define MONO_GEMM_E_NO_ALGO (-3)   /* the library offers no kernel for this problem + epilogue */

/* D[M, N] = act( scale[n] * (A[M, K] . W[N, K]^T) + beta * C[M, N] + bias[n] )
 *   scale: per output column (NULL: 1), bias: per output column (NULL: none), C: residual (NULL or beta == 0: none; may be D),
 *   relu != 0: act = max(., 0).
 * The forward of y = relu(bn(conv1x1(x)) (+ identity)) with A = the channels-last pixel matrix, W = the convolution's [out, in]
 * weight, scale / bias = the frozen norm's affine map; also F.linear(x, W, b) (+ ReLU). */
int mono_gemm_nt_epilogue_f32(const float *A, long long lda, const float *W, long long ldw, const float *C, long long ldc, float *D,
                              long long ldd, int M, int N, int K, const float *scale, float beta, const float *bias, int relu,
                              void *stream);

/* dW[N, K] = dY[M, N]^T . X[M, K]   and, when dbias != NULL,  dbias[n] = sum_m dY[m, n]  from the same launch (the library's
 * bias-gradient epilogue): the weight and bias gradients of y = x W^T + b.  Correct and -- measured on MI355X, ROCm 7.2 -- slow with
 * dbias (about 1 ms per call at M = 8800, N = K = 256): the model does not call it, it stays for the measurement. */
int mono_gemm_tn_bgrad_f32(const float *dY, long long lddy, const float *X, long long ldx, float *dW, long long lddw, float *dbias,
                           int M, int N, int K, void *stream);

/* dX[M, K] = dY[M, N] . W[N, K]  (plain; here so that one library serves a layer's three GEMMs with one tuning cache). */
int mono_gemm_nn_f32(const float *dY, long long lddy, const float *W, long long ldw, float *dX, long long lddx, int M, int N, int K,
                     void *stream);

/* Kernel selection: the first call of a (shape, epilogue) key times the library's `n` best candidates on the call's own operands
 * (3 launches each, on `stream`, host-synchronising) and keeps the fastest; n <= 1 takes the library's first choice without
 * timing (no synchronisation, e.g. under stream capture).  Default 32 (66.50 -> 66.33 ms per train step against 8, round 5).  Returns the previous value. */
int mono_gemm_set_autotune(int n);

/* Number of (shape, epilogue) keys selected so far (tests / diagnostics). */
int mono_gemm_cache_size(void);

#ifdef __cplusplus
}
#endif
#endif
