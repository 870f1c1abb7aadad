/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for MultiScaleDeformableAttention (MSDA).
 *
 * A scalar restatement, in plain C, of the arithmetic the reference's device code
 * performs.  It is the checker for tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py; nothing under monosowa_amd/ may call it.
 *
 * Reference (paths relative to /root/reference/MonoDETR/lib/models/monodetr/ops/src/cuda):
 *   forward  : ms_deformable_im2col_gpu_kernel          ms_deform_im2col_cuda.cuh:237-299
 *              ms_deform_attn_im2col_bilinear           ms_deform_im2col_cuda.cuh:33-84
 *   backward : ..._shm_blocksize_aware_reduce_v1        ms_deform_im2col_cuda.cuh:301-403
 *              ms_deform_attn_col2im_bilinear           ms_deform_im2col_cuda.cuh:87-159
 *   host     : ms_deform_attn_cuda_forward/backward     ms_deform_attn_cuda.cu:20-153
 *
 * Parity pin: checked against golden vectors produced by the reference's own
 * Python definition (ops/functions/ms_deform_attn_func.py:41-61) -- see
 * oracle/gen_golden.py and tests/test_oracle_golden.py.
 *
 * Layouts (all contiguous, row-major):
 *   value  [B, S, M, D]          shapes [L, 2] int64 (H, W)     lsi [L] int64
 *   loc    [B, Lq, M, L, P, 2]   (x, y) normalised               w   [B, Lq, M, L, P]
 *   out / grad_out [B, Lq, M, D]
 *
 * Build: -O2 -ffp-contract=off so that every product and sum rounds once, in the
 * order written (the reference's `loc*size - 0.5` is a rounded product followed by
 * a subtraction, cuh:271-272).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define MSDA_ORACLE_IMPL(T, SUF, FLOOR)                                                        \
  void msda_oracle_forward_##SUF(const T *value, const int64_t *shapes, const int64_t *lsi,    \
                                 const T *loc, const T *attw, int B, int S, int M, int D,      \
                                 int L, int Lq, int P, T *out) {                               \
    for (int b = 0; b < B; ++b)                                                                \
      for (int q = 0; q < Lq; ++q)                                                             \
        for (int m = 0; m < M; ++m) {                                                          \
          const int64_t qm = ((int64_t)b * Lq + q) * M + m;                                    \
          const T *lp = loc + qm * L * P * 2;                                                  \
          const T *wp = attw + qm * L * P;                                                     \
          T *op = out + qm * D;                                                                \
          for (int c = 0; c < D; ++c) op[c] = 0;                                               \
          for (int l = 0; l < L; ++l) {                                                        \
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                      \
            const T *vl = value + ((int64_t)b * S + lsi[l]) * M * D;                           \
            for (int p = 0; p < P; ++p) {                                                      \
              const T lw = lp[(l * P + p) * 2], lh = lp[(l * P + p) * 2 + 1];                  \
              const T wt = wp[l * P + p];                                                      \
              const T h_im = lh * H - (T)0.5, w_im = lw * W - (T)0.5;                          \
              if (!(h_im > -1 && w_im > -1 && h_im < H && w_im < W)) continue;                 \
              const int h_low = (int)FLOOR(h_im), w_low = (int)FLOOR(w_im);                    \
              const int h_high = h_low + 1, w_high = w_low + 1;                                \
              const T dh = h_im - h_low, dw = w_im - w_low;                                    \
              const T hh = 1 - dh, hw = 1 - dw;                                                \
              const T w1 = hh * hw, w2 = hh * dw, w3 = dh * hw, w4 = dh * dw;                  \
              for (int c = 0; c < D; ++c) {                                                    \
                T v1 = 0, v2 = 0, v3 = 0, v4 = 0;                                              \
                if (h_low >= 0 && w_low >= 0)                                                  \
                  v1 = vl[((int64_t)h_low * W + w_low) * M * D + m * D + c];                   \
                if (h_low >= 0 && w_high <= W - 1)                                             \
                  v2 = vl[((int64_t)h_low * W + w_high) * M * D + m * D + c];                  \
                if (h_high <= H - 1 && w_low >= 0)                                             \
                  v3 = vl[((int64_t)h_high * W + w_low) * M * D + m * D + c];                  \
                if (h_high <= H - 1 && w_high <= W - 1)                                        \
                  v4 = vl[((int64_t)h_high * W + w_high) * M * D + m * D + c];                 \
                const T val = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);                         \
                op[c] += val * wt;                                                             \
              }                                                                                \
            }                                                                                  \
          }                                                                                    \
        }                                                                                      \
  }                                                                                            \
                                                                                               \
  /* grad_value / grad_loc / grad_attw must be zero-filled by the caller, as the        */    \
  /* reference host does with zeros_like (ms_deform_attn_cuda.cu:121-123).               */   \
  void msda_oracle_backward_##SUF(const T *value, const int64_t *shapes, const int64_t *lsi,   \
                                  const T *loc, const T *attw, const T *grad_out, int B,       \
                                  int S, int M, int D, int L, int Lq, int P, T *grad_value,    \
                                  T *grad_loc, T *grad_attw) {                                 \
    for (int b = 0; b < B; ++b)                                                                \
      for (int q = 0; q < Lq; ++q)                                                             \
        for (int m = 0; m < M; ++m) {                                                          \
          const int64_t qm = ((int64_t)b * Lq + q) * M + m;                                    \
          const T *lp = loc + qm * L * P * 2;                                                  \
          const T *wp = attw + qm * L * P;                                                     \
          const T *gp = grad_out + qm * D;                                                     \
          T *glp = grad_loc + qm * L * P * 2;                                                  \
          T *gwp = grad_attw + qm * L * P;                                                     \
          for (int l = 0; l < L; ++l) {                                                        \
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                      \
            const int64_t voff = ((int64_t)b * S + lsi[l]) * M * D;                            \
            const T *vl = value + voff;                                                        \
            T *gvl = grad_value + voff;                                                        \
            for (int p = 0; p < P; ++p) {                                                      \
              const T lw = lp[(l * P + p) * 2], lh = lp[(l * P + p) * 2 + 1];                  \
              const T wt = wp[l * P + p];                                                      \
              const T h_im = lh * H - (T)0.5, w_im = lw * W - (T)0.5;                          \
              if (!(h_im > -1 && w_im > -1 && h_im < H && w_im < W)) continue;                 \
              const int h_low = (int)FLOOR(h_im), w_low = (int)FLOOR(w_im);                    \
              const int h_high = h_low + 1, w_high = w_low + 1;                                \
              const T dh = h_im - h_low, dw = w_im - w_low;                                    \
              const T hh = 1 - dh, hw = 1 - dw;                                                \
              const T w1 = hh * hw, w2 = hh * dw, w3 = dh * hw, w4 = dh * dw;                  \
              T acc_gw = 0, acc_gh = 0, acc_ga = 0;                                            \
              for (int c = 0; c < D; ++c) {                                                    \
                const T top_grad = gp[c];                                                      \
                const T tgv = top_grad * wt;                                                   \
                T ghw = 0, gww = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;                            \
                if (h_low >= 0 && w_low >= 0) {                                                \
                  const int64_t i1 = ((int64_t)h_low * W + w_low) * M * D + m * D + c;         \
                  v1 = vl[i1]; ghw -= hw * v1; gww -= hh * v1; gvl[i1] += w1 * tgv;            \
                }                                                                              \
                if (h_low >= 0 && w_high <= W - 1) {                                           \
                  const int64_t i2 = ((int64_t)h_low * W + w_high) * M * D + m * D + c;        \
                  v2 = vl[i2]; ghw -= dw * v2; gww += hh * v2; gvl[i2] += w2 * tgv;            \
                }                                                                              \
                if (h_high <= H - 1 && w_low >= 0) {                                           \
                  const int64_t i3 = ((int64_t)h_high * W + w_low) * M * D + m * D + c;        \
                  v3 = vl[i3]; ghw += hw * v3; gww -= dh * v3; gvl[i3] += w3 * tgv;            \
                }                                                                              \
                if (h_high <= H - 1 && w_high <= W - 1) {                                      \
                  const int64_t i4 = ((int64_t)h_high * W + w_high) * M * D + m * D + c;       \
                  v4 = vl[i4]; ghw += dw * v4; gww += dh * v4; gvl[i4] += w4 * tgv;            \
                }                                                                              \
                const T val = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);                         \
                acc_ga += top_grad * val;                                                      \
                acc_gw += W * gww * tgv;                                                       \
                acc_gh += H * ghw * tgv;                                                       \
              }                                                                                \
              glp[(l * P + p) * 2] = acc_gw;                                                   \
              glp[(l * P + p) * 2 + 1] = acc_gh;                                               \
              gwp[l * P + p] = acc_ga;                                                         \
            }                                                                                  \
          }                                                                                    \
        }                                                                                      \
  }

MSDA_ORACLE_IMPL(float, f32, floorf)
MSDA_ORACLE_IMPL(double, f64, floor)
