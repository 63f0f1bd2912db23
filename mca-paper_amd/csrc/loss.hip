// All-pairs contrastive loss with temperature, forward + backward in one pass.
//
// Reference: MCAPretrainingLoss.forward (model.py:175-233) calls, for each of its 4..60 (a, b) slot pairs,
// torchmultimodal's ContrastiveLossWithTemperature (formula per utils/contrastive_loss_with_temperature.py
// :40-108): logits_a = a·b_allᵀ·e^s, logits_b = b·a_allᵀ·e^s, rows selected by the pair's sample mask,
// loss = (CE(logits_a) + CE(logits_b)) / 2 with labels = global row index; a pair with no selected row is
// NaN and is dropped from the NaN-aware mean (model.py:223-232).  Every call launches its own kernels and
// two all-gathers there; here all terms are evaluated together on the gathered (B, R, D) block.
//
// Note logits_b[i][j] = G[j][i] with G = a_all·b_allᵀ, so one Gram matrix per term serves both directions.
//
// Data-parallel semantics (utils/distributed.py:23-56 + DDP): rank r's loss uses its b local rows against
// all B columns; the autograd all-gather sends column-side gradients back to the owning rank.  Every rank
// holds the whole (B, R, D) block, so this kernel evaluates all ranks' row statistics and returns
// d(sum_r loss_r)/d(local pooled) directly — no reduce-scatter is needed.
#include "common.h"

#define LT 16

// G_t[i][j] = sum_d A_t[i][d] * B_t[j][d]          grid (B/16, B/16, T), block 16x16
__global__ __launch_bounds__(256) void gram_kernel(const float* __restrict__ pooled, const mca_loss_term* __restrict__ terms,
                                                    float* __restrict__ G, int B, int R, int D) {
  __shared__ float As[LT][LT + 1], Bs[LT][LT + 1];
  const int t = blockIdx.z, ti = blockIdx.y * LT, tj = blockIdx.x * LT;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int sa = terms[t].slot_a, sb = terms[t].slot_b;
  float acc = 0.f;
  for (int d0 = 0; d0 < D; d0 += LT) {
    const int ia = ti + ty, jb = tj + ty;
    As[ty][tx] = (ia < B && d0 + tx < D) ? pooled[((int64_t)ia * R + sa) * D + d0 + tx] : 0.f;
    Bs[ty][tx] = (jb < B && d0 + tx < D) ? pooled[((int64_t)jb * R + sb) * D + d0 + tx] : 0.f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < LT; k++) acc += As[ty][k] * Bs[tx][k];
    __syncthreads();
  }
  if (ti + ty < B && tj + tx < B) G[((int64_t)t * B + ti + ty) * B + tj + tx] = acc;
}

__device__ __forceinline__ bool term_valid(const mca_loss_term& tm, uint32_t present) {
  return ((present & tm.and_bits) == tm.and_bits) && (tm.or_bits == 0 || (present & tm.or_bits) != 0);
}

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); i++) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = -INFINITY;
  for (int i = 0; i < (int)(blockDim.x >> 6); i++) t = fmaxf(t, red[i]);
  return t;
}

// per (term, row i): log-sum-exp of row i and of column i of T*G, the two cross-entropies, and the
// temperature-derivative terms.  grid (B, T), block 256.
//   stats[t][i] = {lse_row, lse_col, ce_row + ce_col, dsum}   dsum = d(ce_row+ce_col)/ds
__global__ __launch_bounds__(256) void rowstats_kernel(const float* __restrict__ G, const float* __restrict__ logit_scale,
                                                        float* __restrict__ stats, int B) {
  __shared__ float red[4];
  const int i = blockIdx.x, t = blockIdx.y;
  const float T = expf(*logit_scale);
  const float* Gt = G + (int64_t)t * B * B;
  float mr = -INFINITY, mc = -INFINITY;
  for (int j = threadIdx.x; j < B; j += 256) {
    mr = fmaxf(mr, T * Gt[(int64_t)i * B + j]);
    mc = fmaxf(mc, T * Gt[(int64_t)j * B + i]);
  }
  mr = block_max(mr, red);
  mc = block_max(mc, red);
  float sr = 0.f, sc = 0.f, er = 0.f, ec = 0.f;
  for (int j = threadIdx.x; j < B; j += 256) {
    const float lr = T * Gt[(int64_t)i * B + j], lc = T * Gt[(int64_t)j * B + i];
    const float pr = expf(lr - mr), pc = expf(lc - mc);
    sr += pr; sc += pc; er += pr * lr; ec += pc * lc;
  }
  sr = block_sum(sr, red); sc = block_sum(sc, red); er = block_sum(er, red); ec = block_sum(ec, red);
  if (threadIdx.x == 0) {
    const float diag = T * Gt[(int64_t)i * B + i];
    const float lse_r = mr + logf(sr), lse_c = mc + logf(sc);
    float* st = stats + ((int64_t)t * B + i) * 4;
    st[0] = lse_r; st[1] = lse_c;
    st[2] = (lse_r - diag) + (lse_c - diag);
    st[3] = (er / sr - diag) + (ec / sc - diag);       // d/ds of (ce_row + ce_col), since d(T G)/ds = T G
  }
}

// one block: per (term, rank) valid counts and losses, per-rank number of non-NaN terms; emits the local
// rank's outputs and the per-(term, rank) gradient weights w[t][r] = 1 / (2 n_{t,r} nl_r)  (0 if n = 0)
__global__ __launch_bounds__(256) void reduce_terms_kernel(const float* __restrict__ stats, const uint32_t* __restrict__ present,
                                                            const mca_loss_term* __restrict__ terms, int T, int B, int b_local,
                                                            int row0, float* __restrict__ w, float* __restrict__ term_loss,
                                                            float* __restrict__ loss, float* __restrict__ d_logit_scale) {
  extern __shared__ float sh[];          // [T*W] n, [T*W] sum, [T*W] dsum
  const int W = B / b_local;
  float* n = sh; float* s = sh + T * W; float* ds = sh + 2 * T * W;
  for (int idx = threadIdx.x; idx < T * W; idx += 256) {
    const int t = idx / W, r = idx % W;
    float cnt = 0.f, sum = 0.f, dsum = 0.f;
    for (int i = r * b_local; i < (r + 1) * b_local; i++) {
      if (term_valid(terms[t], present[i])) {
        cnt += 1.f; sum += stats[((int64_t)t * B + i) * 4 + 2]; dsum += stats[((int64_t)t * B + i) * 4 + 3];
      }
    }
    n[idx] = cnt; s[idx] = sum; ds[idx] = dsum;
  }
  __syncthreads();
  // per rank: number of non-NaN terms
  for (int r = threadIdx.x; r < W; r += 256) {
    float nl = 0.f;
    for (int t = 0; t < T; t++) nl += n[t * W + r] > 0.f ? 1.f : 0.f;
    const float denom = nl > 0.f ? nl : 1.f;
    for (int t = 0; t < T; t++) w[t * W + r] = n[t * W + r] > 0.f ? 1.f / (2.f * n[t * W + r] * denom) : 0.f;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int r = row0 / b_local;
    float tot = 0.f, nl = 0.f, dtot = 0.f;
    for (int t = 0; t < T; t++) {
      const float cnt = n[t * W + r];
      if (cnt > 0.f) {
        const float l = s[t * W + r] / (2.f * cnt);
        term_loss[t] = l; tot += l; nl += 1.f; dtot += ds[t * W + r] / (2.f * cnt);
      } else term_loss[t] = __uint_as_float(0x7fc00000u);
    }
    const float denom = nl > 0.f ? nl : 1.f;
    *loss = tot / denom;
    *d_logit_scale = dtot / denom;
  }
}

// dG_t[i][j] = T * ( v_i w_i (softmax_row_i[j] - [i==j]) + v_j w_j (softmax_col_j[i] - [i==j]) )   in place
__global__ __launch_bounds__(256) void dgram_kernel(float* __restrict__ G, const float* __restrict__ stats,
                                                     const uint32_t* __restrict__ present, const mca_loss_term* __restrict__ terms,
                                                     const float* __restrict__ w, const float* __restrict__ logit_scale,
                                                     int B, int b_local) {
  const int t = blockIdx.y, W = B / b_local;
  const float T = expf(*logit_scale);
  const int64_t total = (int64_t)B * B;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int i = (int)(idx / B), j = (int)(idx % B);
    const float l = T * G[(int64_t)t * total + idx];
    float g = 0.f;
    const float wi = term_valid(terms[t], present[i]) ? w[t * W + i / b_local] : 0.f;
    const float wj = term_valid(terms[t], present[j]) ? w[t * W + j / b_local] : 0.f;
    const float eq = i == j ? 1.f : 0.f;
    if (wi != 0.f) g += wi * (expf(l - stats[((int64_t)t * B + i) * 4 + 0]) - eq);
    if (wj != 0.f) g += wj * (expf(l - stats[((int64_t)t * B + j) * 4 + 1]) - eq);
    G[(int64_t)t * total + idx] = T * g;
  }
}

// d_pooled[i_local][slot][:] = sum over terms with slot_a == slot of sum_j dG[i][j] b_j
//                            + sum over terms with slot_b == slot of sum_i' dG[i'][i] a_i'
// grid (R, b_local), block = 4 slices of the j / i' range x 128 columns, partial sums combined through LDS
__global__ __launch_bounds__(512) void dpooled_kernel(const float* __restrict__ dG, const float* __restrict__ pooled,
                                                       const mca_loss_term* __restrict__ terms, int T, int B, int R, int D,
                                                       int row0, float* __restrict__ d_pooled) {
  __shared__ float red[4][128];
  const int slot = blockIdx.x, il = blockIdx.y, i = row0 + il;
  const int dl = threadIdx.x & 127, sl = threadIdx.x >> 7;
  for (int d0 = 0; d0 < D; d0 += 128) {
    const int d = d0 + dl;
    float acc = 0.f;
    if (d < D) {
      for (int t = 0; t < T; t++) {
        const float* g = dG + (int64_t)t * B * B;
        if (terms[t].slot_a == slot) {
          const int sb = terms[t].slot_b;
#pragma unroll 4
          for (int j = sl; j < B; j += 4) acc += g[(int64_t)i * B + j] * pooled[((int64_t)j * R + sb) * D + d];
        }
        if (terms[t].slot_b == slot) {
          const int sa = terms[t].slot_a;
#pragma unroll 4
          for (int k = sl; k < B; k += 4) acc += g[(int64_t)k * B + i] * pooled[((int64_t)k * R + sa) * D + d];
        }
      }
    }
    red[sl][dl] = acc;
    __syncthreads();
    if (sl == 0 && d < D) d_pooled[((int64_t)il * R + slot) * D + d] = red[0][dl] + red[1][dl] + red[2][dl] + red[3][dl];
    __syncthreads();
  }
}

extern "C" int64_t mca_contrastive_workspace_bytes(int B, int T) {
  const int64_t g = (int64_t)T * B * B * 4, st = (int64_t)T * B * 4 * 4, w = (int64_t)T * B * 4;
  return g + st + w + 256;
}

extern "C" int mca_contrastive_fwd_bwd(const float* pooled_all, const uint32_t* present_all, const mca_loss_term* terms,
                                       int T, const float* logit_scale, int B, int b_local, int row0, int R, int D,
                                       float* term_loss, float* loss, float* d_pooled, float* d_logit_scale,
                                       void* workspace, mca_stream_t stream) {
  if (!pooled_all || !present_all || !terms || !logit_scale || !term_loss || !loss || !d_pooled || !d_logit_scale || !workspace)
    return MCA_E_BADARG;
  if (T <= 0 || B <= 0 || b_local <= 0 || B % b_local || row0 % b_local || row0 < 0 || row0 + b_local > B || R <= 0 || D <= 0)
    return MCA_E_BADARG;
  const int W = B / b_local;
  if ((size_t)3 * T * W * sizeof(float) > 60000) return MCA_E_UNSUPPORTED;
  float* G = reinterpret_cast<float*>(workspace);
  float* stats = G + (int64_t)T * B * B;
  float* w = stats + (int64_t)T * B * 4;
  hipStream_t s = as_stream(stream);
  const int tb = (B + LT - 1) / LT;
  hipLaunchKernelGGL(gram_kernel, dim3(tb, tb, T), dim3(256), 0, s, pooled_all, terms, G, B, R, D);
  hipLaunchKernelGGL(rowstats_kernel, dim3(B, T), dim3(256), 0, s, G, logit_scale, stats, B);
  hipLaunchKernelGGL(reduce_terms_kernel, dim3(1), dim3(256), 3 * T * W * sizeof(float), s, stats, present_all, terms, T, B,
                     b_local, row0, w, term_loss, loss, d_logit_scale);
  int gx = (int)(((int64_t)B * B + 255) / 256); if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(dgram_kernel, dim3(gx, T), dim3(256), 0, s, G, stats, present_all, terms, w, logit_scale, B, b_local);
  hipLaunchKernelGGL(dpooled_kernel, dim3(R, b_local), dim3(512), 0, s, G, pooled_all, terms, T, B, R, D, row0, d_pooled);
  return launch_status();
}
