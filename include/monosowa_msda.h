/*
 * monosowa_msda.h -- C-ABI of the MI355X-native MultiScaleDeformableAttention (MSDA) library.
 *
 * Drop-in boundary for the reference's compiled extension module
 * `MultiScaleDeformableAttention` (MonoDETR/lib/models/monodetr/ops/src/vision.cpp:13-16):
 *
 *   msda_forward_*   replaces ms_deform_attn_forward   (ops/src/ms_deform_attn.h:20-39
 *                    -> ops/src/cuda/ms_deform_attn_cuda.cu:20-80 -> ms_deform_im2col_cuda.cuh:923-954)
 *   msda_backward_*  replaces ms_deform_attn_backward  (ops/src/ms_deform_attn.h:41-61
 *                    -> ops/src/cuda/ms_deform_attn_cuda.cu:83-153 -> ms_deform_im2col_cuda.cuh:956-1326)
 *
 * Plain pointers and sizes only (no torch / ATen types).  All pointers are DEVICE pointers on
 * the current HIP device; tensors are dense row-major:
 *
 *   value        [B, S, M, D]        S = sum_l H_l*W_l   (head-interleaved: one token = M*D scalars)
 *   shapes       [L, 2]  int64       (H_l, W_l)                       (cu:67)
 *   level_start  [L]     int64       first token of level l           (cu:68)
 *   loc          [B, Lq, M, L, P, 2] (x, y) normalised to [0,1] over the level, may lie outside
 *   attn_w       [B, Lq, M, L, P]
 *   out / grad_out               [B, Lq, M, D]
 *   grad_value / grad_loc / grad_attn_w   shaped like value / loc / attn_w
 *
 * Semantics: out[b,q,m,:] = sum_{l,p} attn_w * bilinear(value_l[b,:,m,:]; loc*(W_l,H_l) - 0.5),
 * zero outside the level, a point contributing iff -1 < h < H_l and -1 < w < W_l (cuh:274).
 *
 * Ownership: the caller owns every buffer.  Inputs are never written.  Output buffers may be
 * uninitialised: the library zero-fills what its algorithm needs on `stream` (the reference host
 * code allocates zeroed outputs, cu:54,121-123).  Everything is enqueued on `stream`
 * (a hipStream_t; NULL = the null stream), asynchronously, with no host synchronisation and no
 * allocation (`workspace` entry points below), so calls may be captured into a hipGraph.
 *
 * Level geometry on the host: `shapes_host` / `level_start_host` are optional HOST copies of
 * `shapes` / `level_start` (same values).  The launch plans of the fast kernels depend on H_l, W_l.
 * Backward: when the host copies are NULL the library fetches them from the device with a blocking
 * copy on `stream` (correct, but a host synchronisation and not graph-capturable).  Forward: NULL
 * selects the kernel that needs no plan (no synchronisation, somewhat slower).  Callers that know the
 * pyramid (they built it) should pass them.
 *
 * Errors: 0 on success; a positive value is a hipError_t from the launch (the reference only
 * printf's launch errors, cuh:948-952 -- this library returns them); negative values are
 * MSDA_E_* argument errors.  msda_strerror() names either kind.  The compute entry points hold no
 * per-call state and may be called from several threads (the autograd engine does).  Process-wide
 * state, all of it behind mutexes or benign: (1) the kernel-generation option table (msda_set_option /
 * MSDA_* environment variables, read once at first use): every setting computes the same function, and
 * changing it while other threads launch is a race on plain ints, not a correctness hazard; (2) caches
 * keyed by the pyramid: the window tiling (host) and, per device, the tile-window kernels' query table
 * (0.3 MB of device memory, allocated and filled with a blocking copy on the first call for a geometry,
 * kept for the life of the process) -- so the first call for a geometry must not happen inside a stream
 * capture.
 */
#ifndef MONOSOWA_MSDA_H_
#define MONOSOWA_MSDA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSDA_ABI_VERSION 10

#define MSDA_E_NULLPTR (-1)   /* a required pointer is NULL                        */
#define MSDA_E_SHAPE (-2)     /* a dimension is <= 0 or exceeds the indexing range */
#define MSDA_E_UNSUPPORTED (-3)
#define MSDA_E_WORKSPACE (-4)  /* workspace is NULL or smaller than msda_backward_workspace_bytes() */

int msda_abi_version(void);
const char *msda_strerror(int code);

/* Kernel-generation switches for A/B measurements and tests (process-wide; not thread-safe against
 * concurrent launches).  "gather": 0 | 1 | 2 (default 2), "scatter_fixed": 0 | 1 (default 1),
 * "scatter_sorted": 0 | 1 | 2 (default 0; 1: the sort-then-sum scatter on long record lists, 2: always).  Every setting
 * computes the same function.  Returns 0, or MSDA_E_UNSUPPORTED for an unknown name / value. */
int msda_set_option(const char *name, int value);

/* A hash of the option table entries a plan made by msda_saved_plan_f32() depends on: a caller that keeps such a plan across other
 * calls compares the stamp taken at plan time with the one at backward time and lets the backward plan for itself when they differ
 * (msda_fused_backward_view_planned_f32 would otherwise run on tables laid out for another reach / scan source). */
int msda_options_stamp(void);

/* Diagnostics for tests: reads AND resets a device-side event counter of the current device (synchronises the device).
 * "scatter_overflow_rounds": extra bucket rounds of the self-attention backward's cell scatter (a cell received more points
 * from one batch of candidates than its bucket holds).
 * "scan_candidates", "scan_delivering", "scan_point_tests", "scan_delivered", "scan_points_skipped": the cell scatter's scan census --
 * (query, level, tile) candidates looked at / with a point in the tile, sampling points whose tap was evaluated / that landed in a
 * cell of the tile, points skipped by the per-point reach test; counted only by measurement builds of the library
 * (-DMSDA_ROWS_COUNT=1), 0 otherwise.  Returns 0, MSDA_E_UNSUPPORTED for an unknown name, or a hipError_t. */
int msda_debug_counter(const char *name, unsigned long long *out);

/* Bytes of device scratch the backward needs for this geometry (0 if none); the caller allocates
 * it (any alignment >= 16) and passes it to msda_backward_*; it may be NULL when the answer is 0.
 * Contents need not be preserved between calls. */
size_t msda_backward_workspace_bytes(int B, int S, int M, int D, int L, int Lq, int P, int elem_bytes);

int msda_forward_f32(const float *value, const int64_t *shapes, const int64_t *level_start,
                     const float *loc, const float *attn_w, float *out,
                     int B, int S, int M, int D, int L, int Lq, int P,
                     const int64_t *shapes_host, const int64_t *level_start_host, void *stream);

int msda_forward_f64(const double *value, const int64_t *shapes, const int64_t *level_start,
                     const double *loc, const double *attn_w, double *out,
                     int B, int S, int M, int D, int L, int Lq, int P,
                     const int64_t *shapes_host, const int64_t *level_start_host, void *stream);

int msda_backward_f32(const float *value, const int64_t *shapes, const int64_t *level_start,
                      const float *loc, const float *attn_w, const float *grad_out,
                      float *grad_value, float *grad_loc, float *grad_attn_w,
                      int B, int S, int M, int D, int L, int Lq, int P,
                      const int64_t *shapes_host, const int64_t *level_start_host,
                      void *workspace, size_t workspace_bytes, void *stream);

int msda_backward_f64(const double *value, const int64_t *shapes, const int64_t *level_start,
                      const double *loc, const double *attn_w, const double *grad_out,
                      double *grad_value, double *grad_loc, double *grad_attn_w,
                      int B, int S, int M, int D, int L, int Lq, int P,
                      const int64_t *shapes_host, const int64_t *level_start_host,
                      void *workspace, size_t workspace_bytes, void *stream);

/*
 * Fused operator: the pointwise prologue of the reference MODULE,
 * ops/modules/ms_deform_attn.py:146-155, evaluated inside the kernels so that neither the sampling locations nor
 * the attention weights make a round trip through HBM:
 *     attn_w = softmax over the L*P logits of each (query, head)
 *     loc    = ref[l] + offset / (W_l, H_l)                                           (ref_dim == 2)
 *            = ref[l][:2] + offset / P * (ref[l][2]+ref[l][3], ref[l][4]+ref[l][5]) * 0.5   (ref_dim == 6)
 *   offsets [B, Lq, M, L, P, 2]  raw output of the `sampling_offsets` projection
 *   logits  [B, Lq, M, L, P]     raw output of the `attention_weights` projection
 *   ref     [B, Lq, L, ref_dim]  per-level reference points (no gradient is produced for them)
 * The backward returns grad_value, grad_offsets, grad_logits.  Only the fast geometry is supported
 * (float, D = 32, L = P = 4): anything else returns MSDA_E_UNSUPPORTED and the caller uses the unfused entry
 * points.  The host copies of the pyramid are required.
 */
int msda_fused_forward_f32(const float *value, const int64_t *shapes, const int64_t *level_start,
                           const float *offsets, const float *logits, const float *ref, int ref_dim, float *out,
                           int B, int S, int M, int D, int L, int Lq, int P,
                           const int64_t *shapes_host, const int64_t *level_start_host, void *stream);

int msda_fused_backward_f32(const float *value, const int64_t *shapes, const int64_t *level_start,
                            const float *offsets, const float *logits, const float *ref, int ref_dim,
                            const float *grad_out, float *grad_value, float *grad_offsets, float *grad_logits,
                            int B, int S, int M, int D, int L, int Lq, int P,
                            const int64_t *shapes_host, const int64_t *level_start_host,
                            void *workspace, size_t workspace_bytes, void *stream);

/* ABI v5: the same two operators with row strides -- offsets_row_stride / logits_row_stride = floats between consecutive
 * queries in `offsets` / `logits` (and, in the backward, in grad_offsets / grad_logits); M*32 / M*16 when contiguous,
 * e.g. 384 / 384 when both are column blocks of one [B, Lq, 384] projection output (offsets | logits), which lets the
 * module run sampling_offsets and attention_weights (ms_deform_attn.py:142-145) as ONE GEMM. */
int msda_fused_forward_strided_f32(const float *value, const int64_t *shapes, const int64_t *level_start,
                                   const float *offsets, const float *logits, const float *ref, int ref_dim, float *out,
                                   int B, int S, int M, int D, int L, int Lq, int P, int offsets_row_stride,
                                   int logits_row_stride, const int64_t *shapes_host, const int64_t *level_start_host,
                                   void *stream);
int msda_fused_backward_strided_f32(const float *value, const int64_t *shapes, const int64_t *level_start,
                                    const float *offsets, const float *logits, const float *ref, int ref_dim,
                                    const float *grad_out, float *grad_value, float *grad_offsets, float *grad_logits,
                                    int B, int S, int M, int D, int L, int Lq, int P, int offsets_row_stride,
                                    int logits_row_stride, const int64_t *shapes_host, const int64_t *level_start_host,
                                    void *workspace, size_t workspace_bytes, void *stream);

/* ABI v6 (training): the fused forward stores what it evaluated -- loc_save [B, M, L, Lq, P, 2] sampling locations and
 * attn_save [B, M, L, Lq, P] softmax weights, LEVEL-MAJOR (a level's points of neighbouring queries are neighbours in
 * memory: the scatter's scan reads whole lines) -- and the backward of the self-attention shape
 * (Lq == S: msda_gather_win.hip / msda_scatter_rows.hip) reads those instead of re-evaluating the prologue in its two
 * kernels (the row-tile scatter evaluates every point ~2.3x: measured 0.94 -> 0.66 ms per launch at B = 16).
 * grad_offsets / grad_logits still refer to the RAW projection outputs (row strides as in v5).
 * msda_fused_save_supported() -> 1 when msda_fused_backward_saved_f32 covers the geometry and the reference-point form
 * (2-d reference points on the self-attention shape); else use the v5 pair. */
int msda_fused_save_supported(int S, int M, int D, int L, int Lq, int P, int ref_dim, const int64_t *shapes_host,
                              const int64_t *level_start_host);
/* ABI v8: the same question for the VIEW entry points (msda_fused_forward_view_f32 with loc_save / msda_fused_backward_view_f32
 * with saved = 1), whose value token stride and projection row strides also bound the kernels' 32-bit plane addressing: a value
 * column block of a very wide projection (token stride > 4096 floats) is answered 0 here, and the caller takes the v7 pair that
 * re-evaluates the prologue.  (msda_fused_save_supported() answers for dense operands.) */
int msda_fused_save_supported_view(int S, int M, int D, int L, int Lq, int P, int ref_dim, int value_token_stride,
                                   int offsets_row_stride, int logits_row_stride, const int64_t *shapes_host,
                                   const int64_t *level_start_host);
int msda_fused_forward_save_f32(const float *value, const int64_t *shapes, const int64_t *level_start,
                                const float *offsets, const float *logits, const float *ref, int ref_dim, float *out,
                                float *loc_save, float *attn_save, int B, int S, int M, int D, int L, int Lq, int P,
                                int offsets_row_stride, int logits_row_stride, const int64_t *shapes_host,
                                const int64_t *level_start_host, void *stream);
int msda_fused_backward_saved_f32(const float *value, const int64_t *shapes, const int64_t *level_start,
                                  const float *loc_saved, const float *attn_saved, const float *ref, int ref_dim,
                                  const float *grad_out, float *grad_value, float *grad_offsets, float *grad_logits,
                                  int B, int S, int M, int D, int L, int Lq, int P, int offsets_row_stride,
                                  int logits_row_stride, const int64_t *shapes_host, const int64_t *level_start_host,
                                  void *workspace, size_t workspace_bytes, void *stream);

/* ABI v7: the fused operator on a value VIEW -- the two remaining steps of the module around the operator
 * (ops/modules/ms_deform_attn.py:138-141) without passes of their own:
 *   value_token_stride  floats between consecutive tokens of `value` (M*32 when dense; the batch stride is S * value_token_stride).
 *                       768 when `value` is one 256-column block of a [B, S, 768] projection: the three decoder layers'
 *                       value_proj(memory) as ONE GEMM, each layer reading its block in place.
 *   value_mask          optional [B, S] bytes, non-zero = padded token: its value row counts as zero
 *                       (`value.masked_fill(input_padding_mask[..., None], 0)`, :139-140) and its grad_value row comes out zero.
 * Everything else as in v5 / v6: loc_save / attn_save both NULL or both given (v6 forward); `saved` != 0: offsets_or_loc /
 * logits_or_attn are the tensors the v6 forward stored.  grad_value is always dense [B, S, M, D].  Requirements: float,
 * D = 32, L = P = 4, host pyramid; MSDA_E_SHAPE for a stride < M*32 or not a multiple of 4. */
int msda_fused_forward_view_f32(const float *value, int value_token_stride, const unsigned char *value_mask,
                                const int64_t *shapes, const int64_t *level_start, const float *offsets, const float *logits,
                                const float *ref, int ref_dim, float *out, float *loc_save, float *attn_save, int B, int S, int M,
                                int D, int L, int Lq, int P, int offsets_row_stride, int logits_row_stride,
                                const int64_t *shapes_host, const int64_t *level_start_host, void *stream);
int msda_fused_backward_view_f32(const float *value, int value_token_stride, const unsigned char *value_mask,
                                 const int64_t *shapes, const int64_t *level_start, const float *offsets_or_loc,
                                 const float *logits_or_attn, int saved, const float *ref, int ref_dim, const float *grad_out,
                                 float *grad_value, float *grad_offsets, float *grad_logits, int B, int S, int M, int D, int L,
                                 int Lq, int P, int offsets_row_stride, int logits_row_stride, const int64_t *shapes_host,
                                 const int64_t *level_start_host, void *workspace, size_t workspace_bytes, void *stream);

/* ABI v9 (training): the saved backward's PLAN ahead of the backward.  What the self-attention backward launches in front of its
 * scatter -- the directional statistics, the per-head plan and the tiles' candidate tables (msda_plan.h, ~35 us of three small
 * dependent kernels at B = 16) -- depends on the forward's saved sampling locations only.  msda_saved_plan_f32() runs just those
 * into `workspace` (msda_backward_workspace_bytes() bytes, laid out as the backward lays it out): called right behind the v6 / v7
 * forward on a SIDE stream, they overlap whatever the caller's main stream does next.  msda_fused_backward_view_planned_f32() is
 * msda_fused_backward_view_f32(saved = 1) that starts from such a workspace; the caller orders the two (an event) and keeps the
 * workspace alive and untouched in between.  Same options (msda_set_option) at both calls.  MSDA_E_UNSUPPORTED when the call would
 * not run on a directional plan (then msda_fused_backward_view_f32 plans for itself).  The reference has no counterpart: its
 * backward is one kernel (ms_deform_attn_cuda.cu:88-153). */
int msda_saved_plan_f32(const float *loc_saved, const int64_t *shapes, const int64_t *level_start, int value_token_stride, int B, int S,
                        int M, int D, int L, int Lq, int P, int offsets_row_stride, int logits_row_stride, const int64_t *shapes_host,
                        const int64_t *level_start_host, void *workspace, size_t workspace_bytes, void *stream);
int msda_fused_backward_view_planned_f32(const float *value, int value_token_stride, const unsigned char *value_mask,
                                         const int64_t *shapes, const int64_t *level_start, const float *loc_saved,
                                         const float *attn_saved, const float *ref, int ref_dim, const float *grad_out,
                                         float *grad_value, float *grad_offsets, float *grad_logits, int B, int S, int M, int D, int L,
                                         int Lq, int P, int offsets_row_stride, int logits_row_stride, const int64_t *shapes_host,
                                         const int64_t *level_start_host, void *planned_workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MONOSOWA_MSDA_H_ */
