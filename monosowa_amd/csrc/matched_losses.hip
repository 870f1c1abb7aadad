// The matched-pair losses of MonoDETR's SetCriterion in two launches (forward, backward) for all decoder layers:
//   loss_center / loss_bbox  L1 on the 3D-centre projection and the l, r, t, b distances     (monodetr.py:1010-1030)
//   loss_giou                1 - GIoU of the (cx-l, cy-t, cx+r, cy+b) boxes                  (:1032-1044, box_ops)
//   loss_depth               Laplacian aleatoric uncertainty, 1.4142 exp(-s) |d - d*| + s      (:1046-1059)
//   loss_dim                 dimension-aware L1 |s - s*| / s* with the per-layer compensation weight (:1061-1078)
//   loss_angle               12-bin cross entropy + L1 of the target bin's residual           (:1080-1103)
// PyTorch evaluates them with ~60 small kernels forward and ~150 backward (gathers, elementwise ops, index_put
// scatters), enqueued right behind the matcher's host sync where the GPU queue is empty and every launch is exposed.
// One workgroup per decoder layer walks that layer's K matched (image, query, target) triples.
// Outputs are the per-layer SUMS (the caller divides by num_boxes); the backward scatters into zeroed gradients.
#pragma once
#include <hip/hip_runtime.h>

namespace mono {

struct MatchedArgs {
  const float *boxes, *depth, *dims, *angle;        // predictions [NL, B, Q, 6 | 2 | 3 | 24]
  const long long *idx;                             // [3, NL, K]: image, query, flat target index
  const float *t_box, *t_depth, *t_size, *t_res;    // targets [T, 6], [T], [T, 3], [T]
  const long long *t_bin;                           // [T]
  int NL, B, Q, K;
};

__device__ __forceinline__ float sgn(float d) { return (float)(d > 0.f) - (float)(d < 0.f); }

__device__ __forceinline__ float block_sum(float v, float *scratch) {      // 256 threads; result to all
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}

// GIoU of a = (x1, y1, x2, y2) against the constant box b, and its gradient with respect to a.
__device__ __forceinline__ float giou_grad(const float a[4], const float b[4], float g[4]) {
  const float aw = a[2] - a[0], ah = a[3] - a[1];
  const float area_a = aw * ah, area_b = (b[2] - b[0]) * (b[3] - b[1]);
  const float iw = fminf(a[2], b[2]) - fmaxf(a[0], b[0]), ih = fminf(a[3], b[3]) - fmaxf(a[1], b[1]);
  const float iwc = fmaxf(iw, 0.f), ihc = fmaxf(ih, 0.f);
  const float inter = iwc * ihc, uni = area_a + area_b - inter;
  const float cw0 = fmaxf(a[2], b[2]) - fminf(a[0], b[0]), ch0 = fmaxf(a[3], b[3]) - fminf(a[1], b[1]);
  const float cw = fmaxf(cw0, 0.f), ch = fmaxf(ch0, 0.f), area_c = cw * ch;
  const float di[4] = {(iw > 0.f && a[0] > b[0]) ? -ihc : 0.f, (ih > 0.f && a[1] > b[1]) ? -iwc : 0.f,
                       (iw > 0.f && a[2] < b[2]) ? ihc : 0.f, (ih > 0.f && a[3] < b[3]) ? iwc : 0.f};
  const float da[4] = {-ah, -aw, ah, aw};
  const float dc[4] = {(cw0 > 0.f && a[0] < b[0]) ? -ch : 0.f, (ch0 > 0.f && a[1] < b[1]) ? -cw : 0.f,
                       (cw0 > 0.f && a[2] > b[2]) ? ch : 0.f, (ch0 > 0.f && a[3] > b[3]) ? cw : 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float du = da[k] - di[k];
    g[k] = (di[k] * uni - inter * du) / (uni * uni) + (du * area_c - uni * dc[k]) / (area_c * area_c);
  }
  return inter / uni - (area_c - uni) / area_c;
}

struct Pair {
  float box[6], tbox[6], dep[2], tdep, dim[3], tdim[3], ang[24], tres;
  int bin;
  long long row;      // (l * B + b) * Q + q
};

__device__ __forceinline__ Pair load_pair(const MatchedArgs &a, int l, int k) {
  Pair p;
  const long long *ip = a.idx + (long long)l * a.K + k;
  const long long stride = (long long)a.NL * a.K;
  const long long b = ip[0], q = ip[stride], t = ip[2 * stride];
  p.row = ((long long)l * a.B + b) * a.Q + q;
#pragma unroll
  for (int i = 0; i < 6; ++i) { p.box[i] = a.boxes[p.row * 6 + i]; p.tbox[i] = a.t_box[t * 6 + i]; }
  p.dep[0] = a.depth[p.row * 2]; p.dep[1] = a.depth[p.row * 2 + 1]; p.tdep = a.t_depth[t];
#pragma unroll
  for (int i = 0; i < 3; ++i) { p.dim[i] = a.dims[p.row * 3 + i]; p.tdim[i] = a.t_size[t * 3 + i]; }
#pragma unroll
  for (int i = 0; i < 24; ++i) p.ang[i] = a.angle[p.row * 24 + i];
  p.bin = (int)a.t_bin[t];
  p.tres = a.t_res[t];
  return p;
}

__device__ __forceinline__ void to_xyxy(const float c[6], float o[4]) {
  o[0] = c[0] - c[2]; o[1] = c[1] - c[4]; o[2] = c[0] + c[3]; o[3] = c[1] + c[5];
}

// out[l] = {center, bbox, giou, depth, dim, angle} sums; comp[l] = sum |s - s*| / sum |s - s*| / s*
__global__ __launch_bounds__(256) void matched_fwd_kernel(const MatchedArgs a, float *__restrict__ out, float *__restrict__ comp) {
  __shared__ float scratch[4];
  const int l = blockIdx.x;
  float s_center = 0.f, s_bbox = 0.f, s_giou = 0.f, s_depth = 0.f, s_l1 = 0.f, s_dl = 0.f, s_angle = 0.f;
  for (int k = threadIdx.x; k < a.K; k += 256) {
    const Pair p = load_pair(a, l, k);
    s_center += fabsf(p.box[0] - p.tbox[0]) + fabsf(p.box[1] - p.tbox[1]);
    s_bbox += (fabsf(p.box[2] - p.tbox[2]) + fabsf(p.box[3] - p.tbox[3])) + (fabsf(p.box[4] - p.tbox[4]) + fabsf(p.box[5] - p.tbox[5]));
    float xa[4], xb[4], g[4];
    to_xyxy(p.box, xa);
    to_xyxy(p.tbox, xb);
    s_giou += 1.f - giou_grad(xa, xb, g);
    s_depth += 1.4142f * expf(-p.dep[1]) * fabsf(p.dep[0] - p.tdep) + p.dep[1];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float d = fabsf(p.dim[i] - p.tdim[i]);
      s_l1 += d;
      s_dl += d / p.tdim[i];
    }
    float mx = p.ang[0];
#pragma unroll
    for (int i = 1; i < 12; ++i) mx = fmaxf(mx, p.ang[i]);
    float se = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) se += expf(p.ang[i] - mx);
    float logit_t = 0.f, res_p = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) { if (i == p.bin) { logit_t = p.ang[i]; res_p = p.ang[12 + i]; } }
    s_angle += (logf(se) + mx - logit_t) + fabsf(res_p - p.tres);
  }
  const float c = block_sum(s_center, scratch), bb = block_sum(s_bbox, scratch), gi = block_sum(s_giou, scratch);
  const float dp = block_sum(s_depth, scratch), l1 = block_sum(s_l1, scratch), dl = block_sum(s_dl, scratch);
  const float an = block_sum(s_angle, scratch);
  if (threadIdx.x == 0) {
    const float cp = l1 / dl;                 // mean(l1) / mean(l1 / s*): same element count
    comp[l] = cp;
    out[l * 6 + 0] = c; out[l * 6 + 1] = bb; out[l * 6 + 2] = gi; out[l * 6 + 3] = dp; out[l * 6 + 4] = dl * cp; out[l * 6 + 5] = an;
  }
}

// gradients of sum_l sum_j go[l][j] * out[l][j] scattered to the matched rows (the buffers are zero elsewhere; a
// (layer, image, query) row is matched at most once, so plain stores)
__global__ __launch_bounds__(256) void matched_bwd_kernel(const MatchedArgs a, const float *__restrict__ comp, const float *__restrict__ go,
                                                          float *__restrict__ g_boxes, float *__restrict__ g_depth,
                                                          float *__restrict__ g_dims, float *__restrict__ g_angle) {
  const int l = blockIdx.x;
  const float w_center = go[l * 6], w_bbox = go[l * 6 + 1], w_giou = go[l * 6 + 2], w_depth = go[l * 6 + 3];
  const float w_dim = go[l * 6 + 4] * comp[l], w_angle = go[l * 6 + 5];
  for (int k = threadIdx.x; k < a.K; k += 256) {
    const Pair p = load_pair(a, l, k);
    float gb[6];
    gb[0] = w_center * sgn(p.box[0] - p.tbox[0]);
    gb[1] = w_center * sgn(p.box[1] - p.tbox[1]);
#pragma unroll
    for (int i = 2; i < 6; ++i) gb[i] = w_bbox * sgn(p.box[i] - p.tbox[i]);
    float xa[4], xb[4], g[4];
    to_xyxy(p.box, xa);
    to_xyxy(p.tbox, xb);
    (void)giou_grad(xa, xb, g);
    const float s = -w_giou;                                  // loss = 1 - giou
    gb[0] += s * (g[0] + g[2]); gb[1] += s * (g[1] + g[3]);
    gb[2] -= s * g[0]; gb[3] += s * g[2]; gb[4] -= s * g[1]; gb[5] += s * g[3];
#pragma unroll
    for (int i = 0; i < 6; ++i) g_boxes[p.row * 6 + i] = gb[i];
    const float e = 1.4142f * expf(-p.dep[1]), d = p.dep[0] - p.tdep;
    g_depth[p.row * 2] = w_depth * e * sgn(d);
    g_depth[p.row * 2 + 1] = w_depth * (1.f - e * fabsf(d));
#pragma unroll
    for (int i = 0; i < 3; ++i) g_dims[p.row * 3 + i] = w_dim * sgn(p.dim[i] - p.tdim[i]) / p.tdim[i];
    float mx = p.ang[0];
#pragma unroll
    for (int i = 1; i < 12; ++i) mx = fmaxf(mx, p.ang[i]);
    float ex[12], se = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) { ex[i] = expf(p.ang[i] - mx); se += ex[i]; }
    const float inv = 1.f / se;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      g_angle[p.row * 24 + i] = w_angle * (ex[i] * inv - (i == p.bin ? 1.f : 0.f));
      g_angle[p.row * 24 + 12 + i] = (i == p.bin) ? w_angle * sgn(p.ang[12 + i] - p.tres) : 0.f;
    }
  }
}

// ---- classification side of the criterion for all decoder layers (monodetr.py:396-449, sigmoid_focal_loss :302-330) -------------
//   out[l] = { sum over (image, query, class) of the sigmoid focal loss against the matched one-hot targets,
//              class_error = 100 - top-1 accuracy of the matched queries in %,
//              cardinality_error = mean over images of | #(queries whose arg-max is not the last class) - #targets | }
// PyTorch: ~25 small kernels forward (one-hot scatter, sigmoid, BCE, p_t, modulation, alpha weighting, reductions, arg-max
// bookkeeping) and ~30 backward, in the stretch behind the matcher's synchronisation where every launch is exposed.  One
// workgroup per layer: the layer's matched (image, query) -> class map lives in LDS.
constexpr int kFocalThreads = 1024;
constexpr int kFocalMaxCells = 32768;                 // B * Q cells of the class map (bytes of LDS)

struct FocalArgs {
  const float *logits;                // [NL, B, Q, C]
  const long long *idx;               // [3, NL, K]: image, query, flat target
  const long long *labels;            // [T] class of every target
  const float *sizes;                 // [B] targets per image (cardinality)
  int NL, B, Q, C, K;
  float alpha, gamma;
};

__device__ __forceinline__ void focal_class_map(const FocalArgs &a, int l, unsigned char *cls) {
  const int cells = a.B * a.Q;
  for (int i = threadIdx.x; i < cells; i += kFocalThreads) cls[i] = (unsigned char)a.C;          // C = "no object"
  __syncthreads();
  const long long *bi = a.idx + (long long)l * a.K, *qi = a.idx + ((long long)a.NL + l) * a.K, *ti = a.idx + ((long long)2 * a.NL + l) * a.K;
  for (int k = threadIdx.x; k < a.K; k += kFocalThreads) cls[bi[k] * a.Q + qi[k]] = (unsigned char)a.labels[ti[k]];
  __syncthreads();
}

// focal term of one logit x against target t in {0, 1} and its derivative (alpha < 0: no alpha weighting)
__device__ __forceinline__ float focal_term(float x, bool t, float alpha, float gamma, float *dfdx) {
  // log p = -softplus(-x), log(1 - p) = -softplus(x); softplus(z) = max(z, 0) + log1p(exp(-|z|))
  const float lse = log1pf(expf(-fabsf(x)));
  const float log_p = -(fmaxf(-x, 0.f) + lse), log_1p = -(fmaxf(x, 0.f) + lse);
  const float p = 1.f / (1.f + expf(-x));
  const float q = t ? 1.f - p : p;                       // 1 - p_t
  const float ce = t ? -log_p : -log_1p;
  const float w = alpha >= 0.f ? (t ? alpha : 1.f - alpha) : 1.f;
  const float mod = gamma == 2.f ? q * q : powf(q, gamma);
  if (dfdx) {
    // d q / dx = -+ p (1 - p); d ce / dx = -(1 - p) (t = 1), p (t = 0)
    const float dq = t ? -p * (1.f - p) : p * (1.f - p);
    const float dce = t ? -(1.f - p) : p;
    const float dmod = gamma == 2.f ? 2.f * q : gamma * powf(q, gamma - 1.f);
    *dfdx = w * (dmod * dq * ce + mod * dce);
  }
  return w * mod * ce;
}

__global__ __launch_bounds__(kFocalThreads) void focal_fwd_kernel(const FocalArgs a, float *__restrict__ out) {
  __shared__ unsigned char cls[kFocalMaxCells];
  __shared__ float red[kFocalThreads / 64];
  __shared__ int card[256], correct;
  const int l = blockIdx.x, cells = a.B * a.Q;
  if (threadIdx.x < 256) card[threadIdx.x] = 0;
  if (threadIdx.x == 0) correct = 0;
  focal_class_map(a, l, cls);
  const float *lg = a.logits + (long long)l * cells * a.C;
  float sum = 0.f;
  for (int i = threadIdx.x; i < cells; i += kFocalThreads) {
    const int t = cls[i];
    int best = 0;
    float bv = lg[(long long)i * a.C];
    for (int c = 0; c < a.C; ++c) {
      const float x = lg[(long long)i * a.C + c];
      sum += focal_term(x, c == t, a.alpha, a.gamma, nullptr);
      if (x > bv) { bv = x; best = c; }                 // first maximum, like torch.argmax
    }
    if (best != a.C - 1) atomicAdd(&card[i / a.Q], 1);
    if (t < a.C && best == t) atomicAdd(&correct, 1);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int w = 0; w < kFocalThreads / 64; ++w) tot += red[w];
    float ce = 0.f;
    for (int b = 0; b < a.B; ++b) ce += fabsf((float)card[b] - a.sizes[b]);
    out[l * 3] = tot;
    out[l * 3 + 1] = a.K > 0 ? 100.f - (float)correct * (100.f / (float)a.K) : 100.f;
    out[l * 3 + 2] = ce / (float)a.B;
  }
}

// grad_logits[l, b, q, c] = grad_out[l] * d focal / d logit     (grad_out: gradient of the per-layer focal SUM)
__global__ __launch_bounds__(kFocalThreads) void focal_bwd_kernel(const FocalArgs a, const float *__restrict__ grad_out,
                                                                  float *__restrict__ grad_logits) {
  __shared__ unsigned char cls[kFocalMaxCells];
  const int l = blockIdx.x, cells = a.B * a.Q;
  focal_class_map(a, l, cls);
  const float *lg = a.logits + (long long)l * cells * a.C;
  float *gl = grad_logits + (long long)l * cells * a.C;
  const float g = grad_out[l];
  for (int i = threadIdx.x; i < cells * a.C; i += kFocalThreads) {
    const int cell = i / a.C, c = i - cell * a.C;
    float d;
    (void)focal_term(lg[i], c == cls[cell], a.alpha, a.gamma, &d);
    gl[i] = g * d;
  }
}


// ---- the matcher's cost blocks (matcher.py:60-99 of this repo = the reference's matcher.py:53-88 restricted to each image's own
// targets) as one launch.  One thread per (layer, image, query, target slot).  Every operation is the PyTorch expression's own,
// rounded individually and in its order (no contraction: rn()), with the same elementary functions (expf / logf of the device
// library): the floats -- and with them the assignments -- are those of the ~70 elementwise launches it replaces.
// exp / log exactly as PyTorch's elementwise kernels evaluate them: the device library's functions by name.  (hipcc 7.2 expands a
// plain `logf` through the backend's own lowering, which differs from __ocml_log_f32 in the last bit for a third of the arguments
// in (0, 1): tools/ubench/log_variants.hip.)
extern "C" __device__ float __ocml_exp_f32(float);
extern "C" __device__ float __ocml_log_f32(float);

struct CostArgs {
  const float *logits, *boxes;         // [NL, B, Q, C], [NL, B, Q, 6]
  const long long *labels;             // [T]
  const float *tboxes;                 // [T, 6]
  const long long *cols;               // [B, N]: flat target of slot j of image b (padding slots repeat a valid target)
  int NL, B, Q, C, N;
  float w_class, w_3d, w_bbox, w_giou;
};

__device__ __forceinline__ float rn_mul(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float rn_add(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float rn_sub(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}

__global__ __launch_bounds__(256) void match_cost_kernel(const CostArgs a, float *__restrict__ out) {
  const long long n = (long long)a.NL * a.B * a.Q * a.N;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int j = (int)(i % a.N);
  const long long cell = i / a.N;                          // (l, b, q)
  const int b = (int)((cell / a.Q) % a.B);
  const long long t = a.cols[(long long)b * a.N + j];
  const long long id = a.labels[t];
  // focal-style classification cost of the target's class
  const float x = a.logits[cell * a.C + id];
  const float p = 1.0f / rn_add(1.0f, __ocml_exp_f32(-x));                                           // sigmoid
  const float neg = rn_mul(rn_mul(0.75f, rn_mul(p, p)), -__ocml_log_f32(rn_add(rn_sub(1.0f, p), 1e-8f)));
  const float omp = rn_sub(1.0f, p);
  const float pos = rn_mul(rn_mul(0.25f, rn_mul(omp, omp)), -__ocml_log_f32(rn_add(p, 1e-8f)));
  const float cost_class = rn_sub(pos, neg);
  const float *pb = a.boxes + cell * 6, *tb = a.tboxes + t * 6;
  float pa[6], ta[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) { pa[k] = pb[k]; ta[k] = tb[k]; }
  const float c3d = rn_add(fabsf(rn_sub(pa[0], ta[0])), fabsf(rn_sub(pa[1], ta[1])));
  float cbb = rn_add(fabsf(rn_sub(pa[2], ta[2])), fabsf(rn_sub(pa[3], ta[3])));
  cbb = rn_add(cbb, fabsf(rn_sub(pa[4], ta[4])));
  cbb = rn_add(cbb, fabsf(rn_sub(pa[5], ta[5])));
  // (cx, cy, l, r, t, b) -> (cx - l, cy - t, cx + r, cy + b)
  const float ax0 = rn_sub(pa[0], pa[2]), ay0 = rn_sub(pa[1], pa[4]), ax1 = rn_add(pa[0], pa[3]), ay1 = rn_add(pa[1], pa[5]);
  const float bx0 = rn_sub(ta[0], ta[2]), by0 = rn_sub(ta[1], ta[4]), bx1 = rn_add(ta[0], ta[3]), by1 = rn_add(ta[1], ta[5]);
  const float area1 = rn_mul(rn_sub(ax1, ax0), rn_sub(ay1, ay0)), area2 = rn_mul(rn_sub(bx1, bx0), rn_sub(by1, by0));
  // torch.min / torch.max / clamp(min=0) hand NaNs on (fminf / fmaxf would drop them)
  auto tmin = [](const float u, const float v) { return (u != u || u < v) ? u : v; };
  auto tmax = [](const float u, const float v) { return (u != u || u > v) ? u : v; };
  auto relu = [](const float u) { return u < 0.f ? 0.f : u; };
  const float w = relu(rn_sub(tmin(ax1, bx1), tmax(ax0, bx0))), h = relu(rn_sub(tmin(ay1, by1), tmax(ay0, by0)));
  const float inter = rn_mul(w, h);
  const float uni = rn_sub(rn_add(area1, area2), inter);
  const float iou = inter / uni;
  const float wc = relu(rn_sub(tmax(ax1, bx1), tmin(ax0, bx0))), hc = relu(rn_sub(tmax(ay1, by1), tmin(ay0, by0)));
  const float area = rn_mul(wc, hc);
  const float cost_giou = -rn_sub(iou, rn_sub(area, uni) / area);
  float c = rn_add(rn_mul(a.w_bbox, cbb), rn_mul(a.w_3d, c3d));
  c = rn_add(c, rn_mul(a.w_class, cost_class));
  c = rn_add(c, rn_mul(a.w_giou, cost_giou));
  out[i] = c;
}

}  // namespace mono
