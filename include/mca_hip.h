/*
 * mca_hip.h — C ABI of libmca_hip.so: the gfx950 (MI355X) kernels behind the MCA / MMA fusion
 * training step.
 *
 * The reference (josiahbjorgaard/mca-paper) has no FFI: its hot path is stock ATen ops called from
 * Python.  Each entry point below therefore names the reference Python lines whose
 * arithmetic it replaces.  Conventions:
 *   - plain pointers to DEVICE memory, sizes as integers, `stream` is a hipStream_t passed as void*;
 *   - caller allocates every output and workspace; nothing is allocated, freed or synchronised here,
 *     so every call is stream-ordered and graph-capturable;
 *   - return value 0 = launched, negative = argument error (MCA_E_*), nothing launched;
 *   - bf16 tensors are uint16_t (raw bits, round-to-nearest-even); all accumulation is fp32.
 *   - "ld" = leading dimension in ELEMENTS of a row-major matrix.
 */
#ifndef MCA_HIP_H
#define MCA_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mca_stream_t;

#define MCA_OK 0
#define MCA_E_BADARG (-1)
#define MCA_E_ALIGN (-2)
#define MCA_E_UNSUPPORTED (-3)
#define MCA_E_LAUNCH (-4)

/* version / build info: returns a static string */
const char* mca_version(void);

/* ---------------------------------------------------------------------------------------------
 * GEMM  (all Linear layers: encoders.py:190, model.py:49-51,69-71; their autograd backward)
 * --------------------------------------------------------------------------------------------- */
/* C[M,N] = A[M,K] · B[N,K]^T (+ bias[N]) (+ residual[M,N]).   K % 64 == 0, lda/ldb % 8 == 0.
 * out_bf16 != 0: C is bf16, else fp32.  bias / residual may be NULL.  residual rows are indexed
 * row % res_period when res_period > 0 (broadcast over batch), else row.                         */
int mca_gemm_nt(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb,
                void* C, int64_t ldc, int out_bf16,
                const float* bias, const float* residual, int64_t ldres, int64_t res_period,
                int64_t M, int64_t N, int64_t K, mca_stream_t stream);

/* C[M,N] (fp32) = A[M,K]·B[N,K]^T + LayerNorm(x)[M,N], the LayerNorm (model.py:24-31, beta = 0) recomputed in the epilogue
 * as (x - mean[m]) * rstd[m] * gamma[n] from the saved pre-norm tensor and the statistics mca_layernorm_fwd wrote: the
 * residual branches x = Attn(LN(x)) + LN(x), x = FF(LN(x)) + LN(x) of MCALayer.forward (model.py:117-122) without the normed
 * fp32 tensor ever being stored.  M >= 2048, N % 128 == 0, K >= 512 (MCA_E_UNSUPPORTED otherwise: use mca_gemm_nt).      */
int mca_gemm_nt_lnres(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* C, int64_t ldc,
                      const float* x, int64_t ldx, const float* mean, const float* rstd, const float* gamma,
                      int64_t M, int64_t N, int64_t K, mca_stream_t stream);

/* Fused data-gradient GEMM + GEGLU backward (model.py:35-54 autograd): dg = A[M,K]·B[ip,K]^T is never stored;
 * dh[:, n] = dg*gelu(gate), dh[:, ip+n] = dg*a*gelu'(gate) with h = [a | gate].  h, dh: bf16 [M, 2*ip], row stride ldh. */
int mca_gemm_nt_geglu_bwd(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const uint16_t* h,
                          uint16_t* dh, int64_t ldh, int64_t ip, int64_t M, int64_t K, mca_stream_t stream);

/* Fused FF1 GEMM + GEGLU forward (model.py:35-38,49-54): h[M, 2*ip] = A[M,K]·W1[2*ip,K]^T (bf16, both halves kept for the
 * backward) and g[M, ip] = h[:, :ip] * gelu(h[:, ip:]) in one pass (h is not read back).  ip % 64 == 0, ldh >= 2*ip,
 * ldg >= ip, both % 8 == 0.                                                                                          */
int mca_gemm_nt_geglu_fwd(const uint16_t* A, int64_t lda, const uint16_t* W1, int64_t ldb, uint16_t* h, int64_t ldh,
                          uint16_t* g, int64_t ldg, int64_t ip, int64_t M, int64_t K, mca_stream_t stream);

/* C[N,K] += A[R,N]^T · B[R,K]   (weight gradient: reduction over the R token rows; fp32 atomics
 * into C, which the caller zeroes once per step).  lda/ldb % 8 == 0; N and K are arbitrary but the
 * rows of A / B must be readable up to the next multiple of 8 columns (lda >= roundup8(N) etc.).  */
int mca_gemm_tn_acc(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb,
                    float* C, int64_t ldc, int64_t R, int64_t N, int64_t K, mca_stream_t stream);

/* Several weight gradients over the same R token rows in one launch: C_i[N_i,K_i] += A_i[R,N_i]^T · B_i[R,K_i] for
 * i < n <= MCA_TN_MAX_GROUP (the four nn.Linear weight gradients of an MCALayer backward, model.py:109-131: to_q/to_kv,
 * to_out, feedforward[0] (two halves) and feedforward[2]).  Same operand rules as mca_gemm_tn_acc per member; members may
 * not alias each other's C.  Groups the kernel does not suit run as n single launches (same results).              */
#define MCA_TN_MAX_GROUP 8
typedef struct mca_tn_desc {
  const uint16_t* A; int64_t lda;
  const uint16_t* B; int64_t ldb;
  float* C; int64_t ldc;
  int64_t N, K;
} mca_tn_desc;
int mca_gemm_tn_acc_group(const mca_tn_desc* d, int n, int64_t R, mca_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * LayerNorm (model.py:24-31; nn.LayerNorm in encoders.py:51,189,192)
 * --------------------------------------------------------------------------------------------- */
/* y = LN(x)*gamma (+beta), eps; one row = `cols` contiguous floats at x + row*ldx.
 * Optional outputs (NULL to skip): y (fp32) at y + (row / period)*y_bstride + (row % period)*ldy
 * (period <= 0: row*ldy); y_bf16 at row*ld_bf16, zero-filled up to cols_pad.
 * rowmask (u8, 1 = padded row): the row's output is `add` only (fp32) / zeros (bf16) and its stats
 * are (0, 0).  add (fp32 [period, cols]) is added after masking (positional table).
 * mean/rstd: per-row statistics saved for the backward.                                          */
int mca_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta,
                      const uint8_t* rowmask, const float* add, int64_t period,
                      float* y, int64_t ldy, int64_t y_bstride,
                      uint16_t* y_bf16, int64_t ld_bf16, int cols_pad,
                      float* mean, float* rstd, int64_t rows, int cols, float eps, mca_stream_t stream);

/* Backward of the above.  dy is read with the same (period, ldy, y_bstride) row mapping as y was
 * written; masked rows have zero gradient.  dx / dx_bf16 optional.  dgamma/dbeta (fp32[cols]) are
 * ACCUMULATED (atomics).  dxsum (fp32[cols], may be NULL): += the column sums of dx, i.e. the bias
 * gradient of the nn.Linear whose output this norm takes (encoders.py:189-192: Linear -> LayerNorm),
 * without a second pass over dx.                                                                 */
int mca_layernorm_bwd(const float* dy, int64_t ldy, int64_t y_bstride, int64_t period,
                      const float* x, int64_t ldx, const float* gamma,
                      const float* mean, const float* rstd, const uint8_t* rowmask,
                      float* dx, int64_t lddx, uint16_t* dx_bf16, int64_t ld_bf16,
                      float* dgamma, float* dbeta, float* dxsum, int64_t rows, int cols, mca_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * GEGLU (model.py:35-38): h = [a | gate] (two halves of width ip, ld 2*ip) -> g = gelu(gate)*a
 * --------------------------------------------------------------------------------------------- */
int mca_geglu_fwd(const uint16_t* h, uint16_t* g, int64_t rows, int ip, mca_stream_t stream);
int mca_geglu_bwd(const uint16_t* dg, const uint16_t* h, uint16_t* dh, int64_t rows, int ip,
                  mca_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * small data-movement helpers
 * --------------------------------------------------------------------------------------------- */
/* dst[bf16, rows_pad x cols_pad, ld] = src[fp32, rows x cols, lds] (transposed if transpose != 0:
 * dst[c][r] = src[r][c]); padding is zero-filled.  Used to refresh the bf16 weight copies.       */
int mca_cast_pad_bf16(const float* src, int64_t lds, int64_t rows, int64_t cols,
                      uint16_t* dst, int64_t ldd, int64_t rows_pad, int64_t cols_pad, int transpose,
                      mca_stream_t stream);
/* the same for n tensors in ONE launch; descs_dev is a device array (built once by the caller)   */
/* scale: every element is multiplied by it before rounding (0 is read as 1; used to fold scale * log2(e) into W_q)  */
typedef struct { const void* src; void* dst; int64_t lds, rows, cols, ldd, rows_pad, cols_pad; int32_t transpose; float scale; } mca_cast_desc;
int mca_cast_pad_bf16_multi(const mca_cast_desc* descs_dev, int n, mca_stream_t stream);
/* dst[r*ldd + c] = bf16(src[r*lds + c] * scale) */
int mca_f32_to_bf16(const float* src, int64_t lds, uint16_t* dst, int64_t ldd, int64_t rows, int64_t cols,
                    float scale, mca_stream_t stream);
/* dst[(i/period)*dst_bstride + (i%period)*ldd + c] = src[(i % period)*lds + c]  for i < rows
 * (broadcast learned tokens over the batch: model.py:460,472)                                    */
int mca_bcast_rows(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t dst_bstride,
                   int64_t period, int64_t rows, int cols, mca_stream_t stream);
/* dst[(i % period)*ldd + c] += sum over i of src[(i/period)*src_bstride + (i%period)*lds + c]
 * (period == 1: plain column sum -> bias gradients)                                              */
int mca_reduce_rows(const float* src, int64_t lds, int64_t src_bstride, int64_t period,
                    float* dst, int64_t ldd, int64_t rows, int cols, mca_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * TabularEncoder (encoders.py:17-96): nn.Embedding(max_norm=1) in-place row renormalisation;
 * ContinuousValueEncoder front end Linear(1,D)+ReLU on min(x, max_value), and its backward.
 * --------------------------------------------------------------------------------------------- */
int mca_embedding_renorm(float* weight, int64_t rows, int cols, float max_norm, mca_stream_t stream);
/* h1[r, :] = relu(min(x[r], max_value) * w1 + b1) as bf16 (rows x cols); padmask[r] = (x[r] == padding_value) */
int mca_tab_value_fwd(const float* x, const float* w1, const float* b1, uint16_t* h1, uint8_t* padmask,
                      int64_t rows, int cols, float max_value, float padding_value, mca_stream_t stream);
/* dw1 += sum_r dh1*[h1>0]*min(x,max_value);  db1 += sum_r dh1*[h1>0]   (dh1 fp32, ld in elements)            */
int mca_tab_value_bwd(const float* dh1, int64_t ld, const uint16_t* h1, const float* x, float* dw1, float* db1,
                      int64_t rows, int cols, float max_value, mca_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Block-masked fused attention (model.py:73-105 as used by MCALayer :119 and attn_pool :472-473)
 * --------------------------------------------------------------------------------------------- */
/* Packing of the per-modality attention masks (model.py:455-466; encoders.py:196-214 for the row masks): ONE launch
 * instead of ~8 small tensor ops per modality.  For every modality i < n_mod, mask i is (b, n_i) of 1-byte (bool) or
 * 8-byte (int64) elements, non-zero = padded:
 *   padding[s, offset_i + j] = mask_i[s, j] != 0;   rowmask_i[s*n_i + j] = the same (dense copy, skipped when NULL);
 *   present[s] bit i = any token of modality i valid in sample s (MCA.forward's modality_sample_mask);
 *   padding[s, n_tokens - n_fusion ...] = 0 (fusion tokens are never padded).                                        */
#define MCA_MAX_MODALITIES 16
typedef struct { const void* mask; uint8_t* rowmask; int32_t elem_bytes, n, offset, pad_; } mca_mask_desc;
typedef struct { mca_mask_desc m[MCA_MAX_MODALITIES]; int32_t n_mod, batch, n_tokens, n_fusion; } mca_pack_masks_args;
int mca_pack_masks(const mca_pack_masks_args* args, uint8_t* padding, int32_t* present, mca_stream_t stream);

/* Finite check of encoder inputs / pooled outputs without a host sync (encoders.py:197-198,206-213: the reference raises
 * after `.sum()` syncs).  *flag |= bit if any of the n[i] floats at p[i], i < count, is Inf or NaN.  The caller zeroes the
 * flag word, reads it once per step and passes it to mca_adamw_step as skip_flag.                                       */
typedef struct { const float* p[MCA_MAX_MODALITIES]; int64_t n[MCA_MAX_MODALITIES]; int32_t count, pad_; } mca_finite_args;
int mca_nonfinite_flag(const mca_finite_args* args, int32_t* flag, int bit, mca_stream_t stream);
/* *host_pinned = *flag, by a kernel (host_pinned: pinned, device-visible host memory): the host polls it after an event,
 * no copy node in a captured step                                                                                        */
int mca_flag_to_host(const int32_t* flag, int32_t* host_pinned, mca_stream_t stream);

/* keyinfo[b, nk_pad] = padded ? 31 : kgroup[j]; entries >= nk are 31.
 * ktile_flags[b, n_ktiles] = 0 no valid key in the 64-key tile, 1 mixed, 2 all valid.
 * padding: u8 (b, nk), 1 = padded key (model.py:465-466).                                       */
int mca_build_keyinfo(const uint8_t* padding, const uint8_t* kgroup, uint8_t* keyinfo,
                      uint8_t* ktile_flags, int batch, int nk, int nk_pad, mca_stream_t stream);

/* khot[b, nk_pad, 16] (bf16) = one-hot of min(keyinfo, 15): the B^T operand of the MASK PRODUCT.  A structure with at most
 * 15 key groups lets the attention kernels add -32768 * (query may not see the key's group | key padded) to the scores on
 * the matrix pipe (one 32x32x16 MFMA per 32-key block: S += Khot . Qblk^T, Qblk[i][g] = bit g of qmask[i] ? 0 : -32768,
 * Qblk[i][15] = -32768) instead of ~4 vector instructions per score; exp2 of such a score is exactly 0, a row whose every
 * key is blocked ends with a running maximum below -16384 and takes the uniform-row path.                               */
int mca_build_keyhot(const uint8_t* keyinfo, uint16_t* khot, int batch, int nk_pad, mca_stream_t stream);

/* vmean[b, h*64+d] = mean over ALL nk keys of V  (value of a fully-masked softmax row); fixed summation order:
 * bitwise reproducible                                                                            */
int mca_attn_vmean(const uint16_t* V, int64_t kv_bstride, int64_t kv_ld, float* vmean,
                   int batch, int nk, int heads, mca_stream_t stream);
/* the same, skipping the samples whose presence bits (mca_pack_masks: bit m = modality m has a valid token) equal full_bits:
 * with every modality present no query row of the fusion / EAO structures is fully masked, vmean[b] is never read and is
 * left as it was                                                                                                        */
int mca_attn_vmean_if_needed(const uint16_t* V, int64_t kv_bstride, int64_t kv_ld, float* vmean, int batch, int nk,
                             int heads, const int32_t* present, int32_t full_bits, mca_stream_t stream);

typedef struct {
  const uint16_t* q; int64_t q_bstride; int64_t q_ld;     /* q[b*q_bstride + i*q_ld + h*64 + d]   */
  const uint16_t* k; const uint16_t* v; int64_t kv_bstride; int64_t kv_ld;
  uint16_t* o; int64_t o_bstride; int64_t o_ld;
  float* lse;                       /* (b, heads, nq) log2-domain; +inf marks a uniform row       */
  const uint32_t* qmask;            /* (nq) allowed key groups per query row                       */
  const uint8_t* keyinfo;           /* (b, nk_pad)                                                 */
  const uint8_t* ktile_flags;       /* (b, n_ktiles)                                               */
  const int32_t* q_ptr; const uint32_t* q_kt; const int32_t* q_order;   /* q_kt: key-tile index | (full << 31) */
  const float* vmean;               /* (b, heads*64)                                               */
  int batch, heads, nq, nk, nk_pad, n_qtiles, n_ktiles;
  float scale;                      /* dim_head ** -0.5                                            */
  int flags;                        /* MCA_ATTN_* bits                                             */
  const uint16_t* khot;             /* optional (mca_build_keyhot): one-hot key groups, the mask as a matrix product */
} mca_attn_fwd_args;
/* q already carries scale * log2(e) (folded into the bf16 copy of to_q.weight by mca_cast_pad_bf16_multi's per-tensor
 * scale): the kernels then take q.k as the log2-domain logit and never multiply a score.  lse, o, dq, dk, dv keep their
 * meaning (dq is the gradient w.r.t. the UNSCALED q, so the data- and weight-gradient GEMMs are unchanged).              */
#define MCA_ATTN_Q_PRESCALED 1
/* mca_attn_fwd only: LAZY softmax reference.  The score accumulators start from -m (the MFMA C operand) and m moves only when
 * a score exceeds it by more than 12 (log2 units) or a row meets its first real key, instead of following every new row maximum:
 * a third fewer vector instructions per tile (CMU b = 32: 304 against 336 us per layer).  Same contract (o, lse, uniform rows) and
 * the same distance from the exact softmax; P is rounded to bf16 against another reference, so results differ from the textbook
 * form by rounding (2-3e-3 rel-L2 of o).  The engine sets it by default since round 5: over 8 data seeds x {MCA, MMA} the two
 * forms are statistically indistinguishable in their distance to the fp64 oracle (profiles/r05_lazy_softmax_seed_study.txt).   */
#define MCA_ATTN_LAZY_REFERENCE 2
/* dim_head is fixed at 64; query tile 128 rows, key tile 64.                                     */
int mca_attn_fwd(const mca_attn_fwd_args* args, mca_stream_t stream);

/* ---- EAO baseline (model.py:481-596): every modality alone and every combination of modalities is one SEGMENT of a
 * super-sequence; the attention kernels' key groups make the attention block-diagonal over segments.
 * dst[b, r, :] (+)= src[b, r, :], r < rows: replicates a modality's encoded token block into its segments (accumulate = 0)
 * and sums the gradients of the replicas back (accumulate = 1); strides in floats.                                        */
int mca_rows_copy_add(const float* src, int64_t src_bstride, float* dst, int64_t dst_bstride, int64_t rows, int cols,
                      int batch, int accumulate, mca_stream_t stream);
/* MeanTokenProjectionPool without token types / projection (model.py:255-276): out[b, s, :] = mean of the un-padded rows
 * seg_start[s] .. seg_start[s+1]-1 of x[b] (zeros if none); counts[b, s] = their number.  Fixed summation order.          */
int mca_segment_mean_fwd(const float* x, const uint8_t* padding, const int32_t* seg_start, int n_seg, float* out,
                         int32_t* counts, int batch, int n_tokens, int cols, mca_stream_t stream);
/* dx[b, r, :] = padded(b, r) ? 0 : d_out[b, seg_of_row[r], :] / counts[b, seg_of_row[r]]                                  */
int mca_segment_mean_bwd(const float* d_out, const uint8_t* padding, const uint8_t* seg_of_row, const int32_t* counts,
                         int n_seg, float* dx, int batch, int n_tokens, int cols, mca_stream_t stream);

/* MX-fp8 operands of the fusion attention forward (BASELINE configs[4], "fp8 MFMA attention"): OCP e4m3 elements with one
 * E8M0 power-of-two scale byte per 32 elements along the contraction (d for Q and K; keys 0..31 | 32..63 of a 64-key tile
 * for V^T), the operand format of v_mfma_scale_f32_32x32x64_f8f6f4.  npad = n_ktiles * 64 token rows per (sample, head), rows >= n are zero.
 *   q8, k8 : [batch][heads][npad][64]            qs, ks : [batch][heads][npad][2]
 *   v8t    : [batch][heads][n_ktiles][64 d][64]  V transposed per 64-key tile; position p of a row holds key
 *            32 (p >> 5) + 8 ((p >> 2) & 3) + 4 ((p >> 4) & 1) + (p & 3) of the tile (the order in which the S^T
 *            accumulators of the forward kernel hold a query's keys);      vs : [batch][heads][n_ktiles][64 d][2]          */
typedef struct { uint8_t* q8; uint8_t* qs; uint8_t* k8; uint8_t* ks; uint8_t* v8t; uint8_t* vs; int n_ktiles; } mca_attn_fp8_operands;
/* q | k | v (bf16, strides as in mca_attn_fwd_args; q pre-scaled by scale * log2 e) -> the operands above              */
int mca_attn_quant_mxfp8(const uint16_t* q, int64_t q_bstride, int64_t q_ld, const uint16_t* k, const uint16_t* v,
                         int64_t kv_bstride, int64_t kv_ld, const mca_attn_fp8_operands* f, int batch, int heads, int n,
                         mca_stream_t stream);
/* mca_attn_fwd with Q K^T and P V on the block-scaled fp8 matrix instruction (self-attention, MCA_ATTN_Q_PRESCALED only;
 * the q / k / v pointers of args are not read).  o (bf16) and lse as mca_attn_fwd; the backward stays in bf16.            */
int mca_attn_fwd_fp8(const mca_attn_fwd_args* args, const mca_attn_fp8_operands* f, mca_stream_t stream);

/* Backward of the fusion attention with the two score recomputes (S = Q K^T, dP = dO V^T: 4 of the 7 matrix products of
 * the two-pass backward) on the block-scaled fp8 matrix instruction (BASELINE configs[4]; autograd of model.py:87-99).
 * Operands: e4m3 copies of q, k, v, dO, all [batch][heads][npad][64] with [batch][heads][npad][2] E8M0 scale bytes (blocks of
 * 32 ALONG d, the contraction of both products), npad = n_ktiles * 64, rows >= n zero.  q8 / k8 are the arrays of
 * mca_attn_quant_mxfp8, so S is bit for bit the S of mca_attn_fwd_fp8 and P = 2^(S - lse) is consistent with its lse.   */
typedef struct { uint8_t* q8; uint8_t* qs; uint8_t* k8; uint8_t* ks; uint8_t* v8; uint8_t* vs; uint8_t* do8; uint8_t* dos; int n_ktiles; } mca_attn_fp8_bwd_operands;
/* which: bit 0 q, bit 1 k, bit 2 v, bit 3 dO are quantised by this call (15 = all; 12 = v and dO when f->q8 / f->k8 point at
 * the arrays the forward's mca_attn_quant_mxfp8 wrote for the same layer)                                              */
int mca_attn_quant_bwd_mxfp8(const uint16_t* q, int64_t q_bstride, int64_t q_ld, const uint16_t* k, const uint16_t* v,
                             int64_t kv_bstride, int64_t kv_ld, const uint16_t* d_o, int64_t o_bstride, int64_t o_ld,
                             const mca_attn_fp8_bwd_operands* f, int which, int batch, int heads, int n, mca_stream_t stream);
/* mca_attn_bwd_dq / mca_attn_bwd_dkv (arguments as documented below) with those operands; self-attention,
 * MCA_ATTN_Q_PRESCALED and the mask product (khot / qblk) required, kblock_keys = 128 for the dkv pass; the q / v pointers
 * of args are read only by the dkv pass (bf16 Q tile of the dK product); dQ, dK, dV products stay bf16 / fp32.            */
/* delta[b,h,i] = sum_d dO*O ; dvmean[b,h*64+d] = (1/nk) * sum over uniform rows i of dO          */
int mca_attn_bwd_prep(const uint16_t* o, const uint16_t* d_o, int64_t o_bstride, int64_t o_ld,
                      const float* lse, float* delta, float* dvmean,
                      int batch, int heads, int nq, int nk, mca_stream_t stream);

/* The backward runs in two passes WITHOUT atomics (float atomics run at ~1.3 TB/s chip-wide on MI355X: the 0.9 GB of fp32 dQ
 * adds per layer put a 690 us floor under a one-pass kernel, which left the library in round 3): every output element has
 * one owner, results are bitwise reproducible.  MCA_ATTN_Q_PRESCALED is required (MCA_E_UNSUPPORTED otherwise).
 *   mca_attn_bwd_dq : one workgroup per 128-row query tile, key tiles of 64 (the forward's schedule); dq written once,
 *                     bf16 (dq_f32 == 0) or fp32, no pre-zeroing.  Needs q_ptr / q_kt / q_order, n_qtiles128, n_ktiles64.
 *   mca_attn_bwd_dkv: one workgroup per 256- or 128-key block (kblock_keys), query steps of 64; dk, dv bf16.  Needs k_qt
 *                     (per key block the list of 64-row query tiles | full << 31), k_wg (per launch slot the four int32
 *                     {key block, first list entry, number of entries, query tile of the first entry}: one 16-byte load at
 *                     workgroup start), n_qtiles64, n_kblocks256, dvmean.
 * Both need mca_attn_bwd_prep's delta (and dvmean).  The two launches are independent of each other.                     */
typedef struct {
  const uint16_t* q; int64_t q_bstride; int64_t q_ld;
  const uint16_t* k; const uint16_t* v; int64_t kv_bstride; int64_t kv_ld;
  const uint16_t* d_o; int64_t o_bstride; int64_t o_ld;
  const float* lse; const float* delta; const float* dvmean;
  void* dq; int64_t dq_bstride; int64_t dq_ld; int dq_f32;
  uint16_t* dk; uint16_t* dv; int64_t dkv_bstride; int64_t dkv_ld;
  const uint32_t* qmask; const uint8_t* keyinfo; const uint8_t* ktile_flags;
  const int32_t* q_ptr; const uint32_t* q_kt; const int32_t* q_order; int n_qtiles128, n_ktiles64;
  const int32_t* k_wg; const uint32_t* k_qt; int n_qtiles64, n_kblocks256;
  int batch, heads, nq, nk, nk_pad;
  float scale;
  int flags;
  /* optional, both or neither: the mask as a matrix product (see mca_build_keyhot).  qblk[nq, 16] bf16 is the static query
   * side: qblk[i][g] = bit g of qmask[i] ? 0 : -32768 for g < 15, qblk[i][15] = -32768                                  */
  const uint16_t* khot; const uint16_t* qblk;
  /* keys per workgroup of the dkv pass: 0 / 256 (8 wavefronts) or 128 (4 wavefronts, two workgroups per CU: a few % faster
   * for short sequences, slower for long ones); k_wg / k_qt / n_kblocks256 describe key blocks of THIS size              */
  int kblock_keys;
} mca_attn_bwd2_args;
int mca_attn_bwd_dq(const mca_attn_bwd2_args* args, mca_stream_t stream);
int mca_attn_bwd_dkv(const mca_attn_bwd2_args* args, mca_stream_t stream);
int mca_attn_bwd_dq_fp8(const mca_attn_bwd2_args* args, const mca_attn_fp8_bwd_operands* f, mca_stream_t stream);
int mca_attn_bwd_dkv_fp8(const mca_attn_bwd2_args* args, const mca_attn_fp8_bwd_operands* f, mca_stream_t stream);

/* ONE-PASS backward of the fusion self-attention (autograd of model.py:87-99 as called by MCALayer.forward :119), the five
 * matrix products the gradient needs instead of the two passes' seven, still without atomics: ONE workgroup owns a whole
 * (sample, head), walks its key blocks (<= 256 keys: dK / dV in registers) in order and inside a block the query tiles
 * (<= 64 rows) the structure allows; dQ of a tile is summed over its key blocks by a private fp32 read-modify-write in dq_acc
 * (first visit starts from zero, last visit writes bf16 dq): fixed order, bitwise reproducible.  Needs the mask product
 * operands (khot / qblk: at most 15 key groups), MCA_ATTN_Q_PRESCALED, nq == nk == n.  Tables (structure.build_onepass_schedule):
 *   qt_desc[n_qtiles]  = {first row, rows (1..64)}                       query tiles, a partition of 0..n-1
 *   kb_desc[n_kblocks] = {first key, keys (1..256), first entry, entries} key blocks, a partition of 0..n-1, 16-byte aligned
 *   kb_qt[]            = per key block its query tiles | (every pair structurally allowed << 31), ascending
 *   visit[n_kblocks][n_qtiles] = 1 where the block lists the tile;  max_list = longest list (<= 250), n_entries = all lists
 * rowc (b, heads, n_qtiles + 1, 2, 64) fp32: -lse | -delta of the tile's rows (mca_attn_bwd_prep_onepass; positions past a
 * tile's rows, and the whole last tile - the NULL tile every key block's sweep ends on - hold -inf | 0, written once by the caller).  dq_acc: workspace of batch * heads * (n_qtiles + 1) * 4096 floats (the last slot of a
 * (sample, head) belongs to the null tile: written, never read back into a result), contents irrelevant on entry.  dq, dk, dv bf16, every element written.  n_qtiles < 256, n_kblocks <= 64, max_list <= 250, else MCA_E_UNSUPPORTED
 * (the caller keeps the two-pass form).                                                                                  */
typedef struct {
  const uint16_t* q; int64_t q_bstride; int64_t q_ld;           /* q[b*q_bstride + head*q_hstride + i*q_ld + d]                       */
  const uint16_t* k; const uint16_t* v; int64_t kv_bstride; int64_t kv_ld;
  const uint16_t* d_o; int64_t o_bstride; int64_t o_ld;        /* d_o[b*o_bstride + head*o_hstride + i*o_ld + d]                     */
  int64_t q_hstride, o_hstride;                                /* elements between heads: 64 in a (b, n, heads*64) matrix (0 = 64);   */
                                                               /* n*64 in the head-major packed copies of mca_attn_bwd_prep_onepass  */
  const float* rowc; const float* dvmean;
  uint16_t* dq; int64_t dq_bstride; int64_t dq_ld;
  uint16_t* dk; uint16_t* dv; int64_t dkv_bstride; int64_t dkv_ld;
  float* dq_acc;
  const uint8_t* keyinfo; const uint8_t* ktile_flags; const uint16_t* khot; const uint16_t* qblk;
  const int32_t* qt_desc; const int32_t* kb_desc; const uint32_t* kb_qt; const uint8_t* visit;
  int n_qtiles, n_kblocks, max_list;
  int n_entries;                                               /* length of kb_qt: n_entries + 4 * n_kblocks <= 768                 */
  int batch, heads, n, nk_pad, n_ktiles64;
  float scale;
  int flags;
  int split;                                                   /* 0 / 1: one workgroup per (sample, head).  S = 2..8 (small batches): S workgroups per  */
                                                               /* (sample, head), key block kb swept by workgroup kb mod S, dq_acc holds S slices of   */
                                                               /* batch * heads * (n_qtiles + 1) slots (slice-major per (sample, head)), summed in      */
                                                               /* slice order by a second launch of the same call: still no atomics, bitwise repeatable  */
} mca_attn_bwd1_args;
int mca_attn_bwd_onepass(const mca_attn_bwd1_args* args, mca_stream_t stream);
/* mca_attn_bwd_prep for the one-pass form: rowc in tile order (row_slot[q] = tile * 64 + position) instead of delta; dvmean as
 * mca_attn_bwd_prep.  Optional (q_hm and do_hm both or neither): head-major packed copies q_hm / do_hm [batch][heads][n][64] of
 * q (read with q_bstride / q_ld) and d_o - the one-pass kernel then reads a 64-row tile of a head as 8 KiB of contiguous memory
 * (pass them as its q / d_o with q_bstride = heads*n*64, q_hstride = n*64, q_ld = 64: the form the pipelined kernel takes; other
 * strides run the plain form of the same algorithm).  Both buffers need 64 rows (8 KiB) of readable slack behind the last row:
 * the kernel reads whole 64-row tiles.                                                                                    */
int mca_attn_bwd_prep_onepass(const uint16_t* o, const uint16_t* d_o, int64_t o_bstride, int64_t o_ld, const float* lse,
                              const int32_t* row_slot, float* rowc, float* dvmean, int batch, int heads, int n, int n_qtiles,
                              const uint16_t* q, int64_t q_bstride, int64_t q_ld, uint16_t* q_hm, uint16_t* do_hm,
                              mca_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * All-pairs contrastive loss with temperature
 * (model.py:175-233 + torchmultimodal ContrastiveLossWithTemperature, formula per
 *  utils/contrastive_loss_with_temperature.py:40-108,178-195)
 * --------------------------------------------------------------------------------------------- */
typedef struct { int32_t slot_a, slot_b; uint32_t and_bits, or_bits; } mca_loss_term;
/* pooled_all: (B, R, D) fp32, all ranks concatenated rank-major; present_all: (B) u32 bit m = modality
 * m present in the sample.  Local rows are [row0, row0+b).
 * Outputs: term_loss[T] (NaN where a term has no valid row on this rank), loss[1] (NaN-aware mean),
 * d_pooled (b,R,D) = d(sum over ranks of loss_r)/d(local pooled) given every rank's n_valid/n_terms
 * are computed here from present_all; d_logit_scale[1] = d loss_local / d logit_scale.
 * workspace: >= mca_contrastive_workspace_bytes(...)                                              */
int64_t mca_contrastive_workspace_bytes(int B, int T);
int mca_contrastive_fwd_bwd(const float* pooled_all, const uint32_t* present_all,
                            const mca_loss_term* terms, int T, const float* logit_scale,
                            int B, int b_local, int row0, int R, int D,
                            float* term_loss, float* loss, float* d_pooled, float* d_logit_scale,
                            void* workspace, mca_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * clip_grad_norm_ + AdamW over one flat buffer (train_accel_gpu.py:116-118; torch AdamW defaults)
 * --------------------------------------------------------------------------------------------- */
/* sqnorm[0] = sum g^2, sqnorm[MCA_SQNORM_WORDS - 1] = its square root (the total gradient norm, train_accel_gpu.py:116).
 * sqnorm points at MCA_SQNORM_WORDS floats: words 1 .. MCA_SQNORM_WORDS - 2 are scratch for the per-block partial sums, added
 * in a fixed order (the same bits on every launch and on every data-parallel replica); the library itself holds no state,
 * calls on different buffers may overlap.                                                          */
#define MCA_SQNORM_WORDS 1026
int mca_grad_sqnorm(const float* g, int64_t n, float* sqnorm, mca_stream_t stream);
/* grads scaled by min(1, max_norm/(sqrt(sqnorm)+1e-6)) when max_norm > 0, then decoupled AdamW.
 * skip_flag (may be NULL): device word written by mca_nonfinite_flag; non-zero = the step is skipped,
 * parameters and moments untouched (the reference raises before optimizer.step(), encoders.py:197-213). */
int mca_adamw_step(float* p, const float* g, float* m, float* v, int64_t n,
                   float lr, float beta1, float beta2, float eps, float weight_decay,
                   float bias_corr1, float bias_corr2, float max_norm, const float* sqnorm,
                   const int32_t* skip_flag, const float* hyper, mca_stream_t stream);
/* hyper (may be NULL): three device floats {lr, bias_corr1, bias_corr2} that replace the scalar arguments, so that a step
 * captured in a hipGraph takes this step's learning rate and Adam bias corrections at REPLAY time; written by
 * mca_adamw_hyper, a one-thread kernel launched outside the graph (scalar arguments: no host buffer to race with).      */
int mca_adamw_hyper(float* hyper, float lr, float bias_corr1, float bias_corr2, mca_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
