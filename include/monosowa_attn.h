/* monosowa_attn.h -- C ABI of the fp32 head_dim-32 attention kernels for MI355X (gfx950).
 *
 * Replaces the scaled-dot-product core of the two nn.MultiheadAttention modules on MonoDETR's hot path:
 *   depth encoder self-attention    lib/models/monodetr/depth_predictor/transformer.py:57-65   (1920 x 1920 tokens)
 *   decoder depth cross-attention   lib/models/monodetr/depthaware_transformer.py:417-423      (550 x 1920 tokens)
 * i.e. what torch.nn.functional.multi_head_attention_forward computes between its input and output projections:
 *   P = softmax(Q K^T / sqrt(32)),  O = dropout_p(P) V          per (batch, head)
 *
 * Tensors are addressed as [B, H, L, 32] with explicit strides (in floats) for batch, head and token; the 32
 * channels of a head are contiguous.  All pointers are device pointers; launches are asynchronous on `stream`.
 * Dropout keeps an element when its 16 counter-based random bits (a hash of seed, batch*head, query, key) are
 * >= round(p * 65536); the backward regenerates the same mask from the same seed, or reads the forward's keep bits (the *_keep_*
 * pair at the end of this file).
 * Return value: 0, or MONO_ATTN_E_* / a hipError_t.
 */
#ifndef MONOSOWA_ATTN_H
#define MONOSOWA_ATTN_H
#ifdef __cplusplus
extern "C" {
#endif

#define MONO_ATTN_E_NULLPTR (-1)
#define MONO_ATTN_E_SHAPE (-2)

typedef struct { long long batch, head, token; } mono_attn_strides;

/* o [B,H,Lq,32] (strides so), lse [B*H, Lq] contiguous (log2 domain; saved for the backward).  head_dim must be 32. */
int mono_attn_forward_f32(const float *q, const float *k, const float *v, float *o, float *lse, int B, int H, int Lq,
                          int Lk, int head_dim, mono_attn_strides sq, mono_attn_strides sk, mono_attn_strides sv,
                          mono_attn_strides so, float softmax_scale, float dropout_p, unsigned long long seed,
                          void *stream);

/* dq / dk / dv from dout (addressed with so), the forward's o and lse.  delta: scratch [B*H, Lq] floats. */
int mono_attn_backward_f32(const float *q, const float *k, const float *v, const float *o, const float *lse,
                           const float *dout, float *dq, float *dk, float *dv, float *delta, int B, int H, int Lq,
                           int Lk, int head_dim, mono_attn_strides sq, mono_attn_strides sk, mono_attn_strides sv,
                           mono_attn_strides so, mono_attn_strides sdq, mono_attn_strides sdk, mono_attn_strides sdv,
                           float softmax_scale, float dropout_p, unsigned long long seed, void *stream);

/* The same two operations with a key padding mask: key_padding_mask [B, Lk] bytes, non-zero = the key is padding and takes no
 * part in any query's softmax (score -inf, exactly nn.MultiheadAttention's key_padding_mask -- the reference passes one at
 * depthaware_transformer.py:456-459 and depth_predictor/transformer.py:57-60); dk / dv of a masked key come out as zero.  NULL =
 * no mask (the functions above).  A query whose keys are all masked yields NaN, as torch's softmax does. */
int mono_attn_forward_masked_f32(const float *q, const float *k, const float *v, const unsigned char *key_padding_mask, float *o,
                                 float *lse, int B, int H, int Lq, int Lk, int head_dim, mono_attn_strides sq,
                                 mono_attn_strides sk, mono_attn_strides sv, mono_attn_strides so, float softmax_scale,
                                 float dropout_p, unsigned long long seed, void *stream);
int mono_attn_backward_masked_f32(const float *q, const float *k, const float *v, const unsigned char *key_padding_mask,
                                  const float *o, const float *lse, const float *dout, float *dq, float *dk, float *dv, float *delta,
                                  int B, int H, int Lq, int Lk, int head_dim, mono_attn_strides sq, mono_attn_strides sk,
                                  mono_attn_strides sv, mono_attn_strides so, mono_attn_strides sdq, mono_attn_strides sdk,
                                  mono_attn_strides sdv, float softmax_scale, float dropout_p, unsigned long long seed,
                                  void *stream);

/* The same two operations with the dropout mask handed from the forward to the backward instead of being re-hashed there (round 4:
 * the hash is 14 % of the backward).  keep_bits: mono_attn_keep_words(B, H, Lq, Lk) 32-bit words, written by the forward when
 * dropout_p > 0 (one word per forward lane and 64-key tile, in the kernels' own register order: opaque to the caller), read by the
 * backward of the SAME call geometry; NULL = the functions above (the backward regenerates the mask from the seed -- the same mask).
 * key_padding_mask may be NULL. */
long long mono_attn_keep_words(int B, int H, int Lq, int Lk);
int mono_attn_forward_keep_f32(const float *q, const float *k, const float *v, const unsigned char *key_padding_mask,
                               unsigned *keep_bits, float *o, float *lse, int B, int H, int Lq, int Lk, int head_dim,
                               mono_attn_strides sq, mono_attn_strides sk, mono_attn_strides sv, mono_attn_strides so,
                               float softmax_scale, float dropout_p, unsigned long long seed, void *stream);
int mono_attn_backward_keep_f32(const float *q, const float *k, const float *v, const unsigned char *key_padding_mask,
                                const unsigned *keep_bits, const float *o, const float *lse, const float *dout, float *dq, float *dk,
                                float *dv, float *delta, int B, int H, int Lq, int Lk, int head_dim, mono_attn_strides sq,
                                mono_attn_strides sk, mono_attn_strides sv, mono_attn_strides so, mono_attn_strides sdq,
                                mono_attn_strides sdk, mono_attn_strides sdv, float softmax_scale, float dropout_p,
                                unsigned long long seed, void *stream);

#ifdef __cplusplus
}
#endif
#endif
