// HBM-bound kernels of the MCA step: LayerNorm fwd/bwd (with the encoder's pad-mask / positional-table
// fusions), GEGLU fwd/bwd, bf16 casts, row broadcast / row reduction.  One wavefront (64 lanes) per
// row for the norms; 16-byte accesses wherever the layout allows.
#include "common.h"

// =====================================================================================================
// LayerNorm forward.  Reference: model.py:24-31 (gamma only), encoders.py:189-192,199-209 (affine,
// pad rows zeroed, + positional table), encoders.py:68-70 (tabular value norm).
// =====================================================================================================
template <int VEC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma, const float* __restrict__ beta,
    const uint8_t* __restrict__ rowmask, const float* __restrict__ add, int64_t period,
    float* __restrict__ y, int64_t ldy, int64_t y_bstride,
    u16* __restrict__ y_bf16, int64_t ld_bf16, int cols_pad,
    float* __restrict__ mean_out, float* __restrict__ rstd_out, int64_t rows, int cols, float eps) {
  constexpr int NI = 16 / VEC;             // up to 1024 columns
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const bool masked = rowmask && rowmask[row];
    const float* xr = x + row * ldx;
    float v[NI][VEC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; i++) {
      const int c = (lane + 64 * i) * VEC;
      if (c < cols) {
        if (VEC == 4) {
          float4 t = masked ? make_float4(0, 0, 0, 0) : *reinterpret_cast<const float4*>(xr + c);
          v[i][0] = t.x; v[i][1 % VEC] = t.y; v[i][2 % VEC] = t.z; v[i][3 % VEC] = t.w;
        } else {
          v[i][0] = masked ? 0.f : xr[c];
        }
#pragma unroll
        for (int j = 0; j < VEC; j++) s += v[i][j];
      } else {
#pragma unroll
        for (int j = 0; j < VEC; j++) v[i][j] = 0.f;
      }
    }
    const float mean = wave_sum(s) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NI; i++) {
      const int c = (lane + 64 * i) * VEC;
      if (c < cols) {
#pragma unroll
        for (int j = 0; j < VEC; j++) { const float d = v[i][j] - mean; q += d * d; }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)cols + eps);
    if (lane == 0) {
      if (mean_out) mean_out[row] = masked ? 0.f : mean;
      if (rstd_out) rstd_out[row] = masked ? 0.f : rstd;
    }
    const int64_t prow = period > 0 ? row % period : row;
    float* yr = y ? (period > 0 ? y + (row / period) * y_bstride + prow * ldy : y + row * ldy) : nullptr;
    const float* ar = add ? add + prow * (int64_t)cols : nullptr;
    u16* br = y_bf16 ? y_bf16 + row * ld_bf16 : nullptr;
#pragma unroll
    for (int i = 0; i < NI; i++) {
      const int c = (lane + 64 * i) * VEC;
      if (c < cols) {
        float o[VEC];
#pragma unroll
        for (int j = 0; j < VEC; j++) {
          float t = (v[i][j] - mean) * rstd * gamma[c + j];
          if (beta) t += beta[c + j];
          o[j] = masked ? 0.f : t;
        }
        if (br) {
          if (VEC == 4) {
            uint2 pk; pk.x = pack2bf(o[0], o[1 % VEC]); pk.y = pack2bf(o[2 % VEC], o[3 % VEC]);
            *reinterpret_cast<uint2*>(br + c) = pk;
          } else {
            br[c] = f2bf(o[0]);
          }
        }
        if (yr) {
          if (ar) {
#pragma unroll
            for (int j = 0; j < VEC; j++) o[j] += ar[c + j];
          }
          if (VEC == 4) *reinterpret_cast<float4*>(yr + c) = make_float4(o[0], o[1 % VEC], o[2 % VEC], o[3 % VEC]);
          else yr[c] = o[0];
        }
      }
    }
    if (br) for (int c = cols + lane; c < cols_pad; c += 64) br[c] = 0;
  }
}

// Trunk form (gamma only, bf16 output + statistics, cols % 256 == 0, no mask / positional table): TWO rows per wavefront at
// once.  A wavefront of the general kernel has one 2 KB row in flight and spends ~4 us per row on load -> mean -> variance
// -> store; with two independent rows per wavefront the loads and both reduction chains overlap.
template <int NI>
__global__ __launch_bounds__(256) void ln_fwd_trunk_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                            u16* __restrict__ y_bf16, int64_t ld_bf16, float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out, int64_t rows, int cols, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * 2;
  if (r0 >= rows) return;
  const bool two = r0 + 1 < rows;
  const float* xa = x + r0 * ldx;
  const float* xb = x + (two ? r0 + 1 : r0) * ldx;
  float4 va[NI], vb[NI], gv[NI];
#pragma unroll
  for (int i = 0; i < NI; i++) { va[i] = *reinterpret_cast<const float4*>(xa + (lane + 64 * i) * 4); vb[i] = *reinterpret_cast<const float4*>(xb + (lane + 64 * i) * 4); }
#pragma unroll
  for (int i = 0; i < NI; i++) gv[i] = *reinterpret_cast<const float4*>(gamma + (lane + 64 * i) * 4);          // with the rows, not after the reductions
  float sa = 0.f, sb = 0.f;
#pragma unroll
  for (int i = 0; i < NI; i++) {          // same summation order as ln_fwd_kernel
    sa += va[i].x; sa += va[i].y; sa += va[i].z; sa += va[i].w;
    sb += vb[i].x; sb += vb[i].y; sb += vb[i].z; sb += vb[i].w;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o, WAVE); sb += __shfl_xor(sb, o, WAVE); }
  const float ma = sa / (float)cols, mb = sb / (float)cols;
  float qa = 0.f, qb = 0.f;
#pragma unroll
  for (int i = 0; i < NI; i++) {
    float d;
    d = va[i].x - ma; qa += d * d; d = va[i].y - ma; qa += d * d; d = va[i].z - ma; qa += d * d; d = va[i].w - ma; qa += d * d;
    d = vb[i].x - mb; qb += d * d; d = vb[i].y - mb; qb += d * d; d = vb[i].z - mb; qb += d * d; d = vb[i].w - mb; qb += d * d;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { qa += __shfl_xor(qa, o, WAVE); qb += __shfl_xor(qb, o, WAVE); }
  const float ra = rsqrtf(qa / (float)cols + eps), rb = rsqrtf(qb / (float)cols + eps);
  if (lane == 0) {
    if (mean_out) { mean_out[r0] = ma; if (two) mean_out[r0 + 1] = mb; }
    if (rstd_out) { rstd_out[r0] = ra; if (two) rstd_out[r0 + 1] = rb; }
  }
#pragma unroll
  for (int i = 0; i < NI; i++) {
    const int c = (lane + 64 * i) * 4;
    const float4 g = gv[i];
    uint2 pk;
    pk.x = pack2bf((va[i].x - ma) * ra * g.x, (va[i].y - ma) * ra * g.y); pk.y = pack2bf((va[i].z - ma) * ra * g.z, (va[i].w - ma) * ra * g.w);
    *reinterpret_cast<uint2*>(y_bf16 + r0 * ld_bf16 + c) = pk;
    if (two) {
      pk.x = pack2bf((vb[i].x - mb) * rb * g.x, (vb[i].y - mb) * rb * g.y); pk.y = pack2bf((vb[i].z - mb) * rb * g.z, (vb[i].w - mb) * rb * g.w);
      *reinterpret_cast<uint2*>(y_bf16 + (r0 + 1) * ld_bf16 + c) = pk;
    }
  }
}

// Narrow rows (an encoder's input norm: 35 ... 713 features, bf16 output padded to the GEMM's K, no fp32 output): SIXTEEN lanes
// per row, four rows per wavefront.  With a wavefront per row a 74-column row keeps 74 of 1,024 lane slots of the general kernel
// busy and the loop is bound by its per-row latency (48,000 x 74: 32 us for 14 MB).
__global__ __launch_bounds__(256) void ln_fwd_narrow_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const uint8_t* __restrict__ rowmask,
                                                             u16* __restrict__ y_bf16, int64_t ld_bf16, int cols_pad,
                                                             float* __restrict__ mean_out, float* __restrict__ rstd_out, int64_t rows,
                                                             int cols, float eps) {
  const int sub = threadIdx.x & 15, rloc = threadIdx.x >> 4;
  for (int64_t row = (int64_t)blockIdx.x * 16 + rloc; row < rows; row += (int64_t)gridDim.x * 16) {
    const bool masked = rowmask && rowmask[row];
    const float* xr = x + row * ldx;
    float v[16];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int c = sub + 16 * k;
      v[k] = (c < cols && !masked) ? xr[c] : 0.f;
      s += v[k];
    }
    s += __shfl_xor(s, 8, WAVE); s += __shfl_xor(s, 4, WAVE); s += __shfl_xor(s, 2, WAVE); s += __shfl_xor(s, 1, WAVE);
    const float mean = s / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const float d = v[k] - mean;
      if (sub + 16 * k < cols) q += d * d;
    }
    q += __shfl_xor(q, 8, WAVE); q += __shfl_xor(q, 4, WAVE); q += __shfl_xor(q, 2, WAVE); q += __shfl_xor(q, 1, WAVE);
    const float rstd = rsqrtf(q / (float)cols + eps);
    if (sub == 0) {
      if (mean_out) mean_out[row] = masked ? 0.f : mean;
      if (rstd_out) rstd_out[row] = masked ? 0.f : rstd;
    }
    u16* br = y_bf16 + row * ld_bf16;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int c = sub + 16 * k;
      if (c < cols) {
        float t = (v[k] - mean) * rstd * gamma[c];
        if (beta) t += beta[c];
        br[c] = f2bf(masked ? 0.f : t);
      }
    }
    for (int c = cols + sub; c < cols_pad; c += 16) br[c] = 0;
  }
}

extern "C" int mca_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta,
                                 const uint8_t* rowmask, const float* add, int64_t period,
                                 float* y, int64_t ldy, int64_t y_bstride,
                                 uint16_t* y_bf16, int64_t ld_bf16, int cols_pad,
                                 float* mean, float* rstd, int64_t rows, int cols, float eps,
                                 mca_stream_t stream) {
  if (!x || !gamma || rows < 0 || cols <= 0 || cols > 1024) return MCA_E_BADARG;
  if (rows == 0) return MCA_OK;
  if (!y && y_bf16 && !add && cols <= 256 && mca_knobs[12] != 1) {          // narrow rows, bf16 output only (knob 12 = 1: general kernel)
    int64_t nb = (rows + 15) / 16; if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(ln_fwd_narrow_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), x, ldx, gamma, beta, rowmask, y_bf16, ld_bf16,
                       cols_pad, mean, rstd, rows, cols, eps);
    return launch_status();
  }
  const bool vec = (cols % 4 == 0) && (ldx % 4 == 0) && (!y || (ldy % 4 == 0 && y_bstride % 4 == 0)) &&
                   (!y_bf16 || ld_bf16 % 4 == 0) &&
                   ((uintptr_t)x % 16 == 0) && (!y || (uintptr_t)y % 16 == 0) && (!add || (uintptr_t)add % 16 == 0);
  if (vec && !beta && !rowmask && !add && !y && y_bf16 && cols_pad <= cols && (cols == 256 || cols == 512 || cols == 1024) &&
      (uintptr_t)gamma % 16 == 0 && mca_knobs[12] != 1) {          // knob 12 = 1: general kernel (A/B); any row count, so that a
                                                                   // given call form always takes the same arithmetic path
    const unsigned nb = (unsigned)((rows + 7) / 8);
    if (cols == 256) hipLaunchKernelGGL(ln_fwd_trunk_kernel<1>, dim3(nb), dim3(256), 0, as_stream(stream), x, ldx, gamma, y_bf16, ld_bf16, mean, rstd, rows, cols, eps);
    else if (cols == 512) hipLaunchKernelGGL(ln_fwd_trunk_kernel<2>, dim3(nb), dim3(256), 0, as_stream(stream), x, ldx, gamma, y_bf16, ld_bf16, mean, rstd, rows, cols, eps);
    else hipLaunchKernelGGL(ln_fwd_trunk_kernel<4>, dim3(nb), dim3(256), 0, as_stream(stream), x, ldx, gamma, y_bf16, ld_bf16, mean, rstd, rows, cols, eps);
    return launch_status();
  }
  int64_t blocks = (rows + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  if (vec)
    hipLaunchKernelGGL(ln_fwd_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, ldx, gamma, beta,
                       rowmask, add, period, y, ldy, y_bstride, y_bf16, ld_bf16, cols_pad, mean, rstd, rows, cols, eps);
  else
    hipLaunchKernelGGL(ln_fwd_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, ldx, gamma, beta,
                       rowmask, add, period, y, ldy, y_bstride, y_bf16, ld_bf16, cols_pad, mean, rstd, rows, cols, eps);
  return launch_status();
}

// =====================================================================================================
// LayerNorm backward:  g = dy*gamma;  dx = rstd*(g - mean(g) - xhat*mean(g*xhat));
// dgamma += sum_rows dy*xhat;  dbeta += sum_rows dy.   Per-block partials in registers, one atomic
// per column per block.
// =====================================================================================================
template <int VEC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(
    const float* __restrict__ dy, int64_t ldy, int64_t y_bstride, int64_t period,
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
    const float* __restrict__ mean_in, const float* __restrict__ rstd_in, const uint8_t* __restrict__ rowmask,
    float* __restrict__ dx, int64_t lddx, u16* __restrict__ dx_bf16, int64_t ld_bf16,
    float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dxsum, int64_t rows, int cols) {
  constexpr int NI = 16 / VEC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float dg[NI][VEC], db[NI][VEC], ds[NI][VEC];
#pragma unroll
  for (int i = 0; i < NI; i++)
#pragma unroll
    for (int j = 0; j < VEC; j++) { dg[i][j] = 0.f; db[i][j] = 0.f; ds[i][j] = 0.f; }
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const bool masked = rowmask && rowmask[row];
    const float* xr = x + row * ldx;
    const float* dyr = period > 0 ? dy + (row / period) * y_bstride + (row % period) * ldy : dy + row * ldy;
    const float mean = mean_in[row], rstd = rstd_in[row];
    float xh[NI][VEC], g[NI][VEC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NI; i++) {
      const int c = (lane + 64 * i) * VEC;
#pragma unroll
      for (int j = 0; j < VEC; j++) { xh[i][j] = 0.f; g[i][j] = 0.f; }
      if (c < cols && !masked) {
        float xv[VEC], dv[VEC];
        if (VEC == 4) {
          float4 a = *reinterpret_cast<const float4*>(xr + c), d = *reinterpret_cast<const float4*>(dyr + c);
          xv[0] = a.x; xv[1 % VEC] = a.y; xv[2 % VEC] = a.z; xv[3 % VEC] = a.w;
          dv[0] = d.x; dv[1 % VEC] = d.y; dv[2 % VEC] = d.z; dv[3 % VEC] = d.w;
        } else { xv[0] = xr[c]; dv[0] = dyr[c]; }
#pragma unroll
        for (int j = 0; j < VEC; j++) {
          xh[i][j] = (xv[j] - mean) * rstd;
          g[i][j] = dv[j] * gamma[c + j];
          dg[i][j] += dv[j] * xh[i][j];
          db[i][j] += dv[j];
          s1 += g[i][j];
          s2 += g[i][j] * xh[i][j];
        }
      }
    }
    s1 = wave_sum(s1) / (float)cols;
    s2 = wave_sum(s2) / (float)cols;
    float* dxr = dx ? dx + row * lddx : nullptr;
    u16* br = dx_bf16 ? dx_bf16 + row * ld_bf16 : nullptr;
#pragma unroll
    for (int i = 0; i < NI; i++) {
      const int c = (lane + 64 * i) * VEC;
      if (c < cols) {
        float o[VEC];
#pragma unroll
        for (int j = 0; j < VEC; j++) { o[j] = masked ? 0.f : rstd * (g[i][j] - s1 - xh[i][j] * s2); ds[i][j] += o[j]; }
        if (dxr) {
          if (VEC == 4) *reinterpret_cast<float4*>(dxr + c) = make_float4(o[0], o[1 % VEC], o[2 % VEC], o[3 % VEC]);
          else dxr[c] = o[0];
        }
        if (br) {
          if (VEC == 4) {
            uint2 pk; pk.x = pack2bf(o[0], o[1 % VEC]); pk.y = pack2bf(o[2 % VEC], o[3 % VEC]);
            *reinterpret_cast<uint2*>(br + c) = pk;
          } else br[c] = f2bf(o[0]);
        }
      }
    }
  }
  // block reduction of the parameter-gradient partials through LDS, then one atomic per column
  __shared__ float red[4][1024];
  for (int pass = 0; pass < 3; pass++) {
    float* target = pass == 0 ? dgamma : (pass == 1 ? dbeta : dxsum);
    if (!target) continue;
#pragma unroll
    for (int i = 0; i < NI; i++) {
      const int c = (lane + 64 * i) * VEC;
#pragma unroll
      for (int j = 0; j < VEC; j++)
        if (c + j < 1024) red[wave][c + j] = pass == 0 ? dg[i][j] : (pass == 1 ? db[i][j] : ds[i][j]);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < cols; c += 256) {
      const float t = red[0][c] + red[1][c] + red[2][c] + red[3][c];
      if (t != 0.f) atomicAdd(target + c, t);
    }
    __syncthreads();
  }
}

// Trunk form of the backward (gamma only, no mask / period, fp32 dx + bf16 copy, cols % 256 == 0): two rows per wavefront
// and iteration, as in ln_fwd_trunk_kernel (8 KB of loads in flight per wavefront instead of 4).
template <int NI>
__global__ __launch_bounds__(256) void ln_bwd_trunk_kernel(const float* __restrict__ dy, int64_t ldy, const float* __restrict__ x, int64_t ldx,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                            const float* __restrict__ rstd_in, float* __restrict__ dx, int64_t lddx,
                                                            u16* __restrict__ dx_bf16, int64_t ld_bf16, float* __restrict__ dgamma,
                                                            int64_t rows, int cols) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 gm[NI], dg[NI];
#pragma unroll
  for (int i = 0; i < NI; i++) { gm[i] = *reinterpret_cast<const float4*>(gamma + (lane + 64 * i) * 4); dg[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
  const float inv_cols = 1.f / (float)cols;
  for (int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * 2; r0 < rows; r0 += (int64_t)gridDim.x * 8) {
    const bool two = r0 + 1 < rows;
    const int64_t r1 = two ? r0 + 1 : r0;
    float4 xa[NI], xb[NI], da[NI], db[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) {
      const int c = (lane + 64 * i) * 4;
      xa[i] = *reinterpret_cast<const float4*>(x + r0 * ldx + c); da[i] = *reinterpret_cast<const float4*>(dy + r0 * ldy + c);
      xb[i] = *reinterpret_cast<const float4*>(x + r1 * ldx + c); db[i] = *reinterpret_cast<const float4*>(dy + r1 * ldy + c);
    }
    const float ma = mean_in[r0], ra = rstd_in[r0], mb = mean_in[r1], rb = rstd_in[r1];
    float s1a = 0.f, s2a = 0.f, s1b = 0.f, s2b = 0.f;
    // xhat overwrites x, g = dy * gamma overwrites dy
#define LNB_ELEM(X, D, G, DG, M, R, S1, S2, LIVE)  { const float xh_ = ((X) - (M)) * (R); const float g_ = (D) * (G); if (LIVE) (DG) += (D) * xh_; \
                                                    (S1) += g_; (S2) += g_ * xh_; (X) = xh_; (D) = g_; }
#pragma unroll
    for (int i = 0; i < NI; i++) {
      LNB_ELEM(xa[i].x, da[i].x, gm[i].x, dg[i].x, ma, ra, s1a, s2a, true)  LNB_ELEM(xa[i].y, da[i].y, gm[i].y, dg[i].y, ma, ra, s1a, s2a, true)
      LNB_ELEM(xa[i].z, da[i].z, gm[i].z, dg[i].z, ma, ra, s1a, s2a, true)  LNB_ELEM(xa[i].w, da[i].w, gm[i].w, dg[i].w, ma, ra, s1a, s2a, true)
      LNB_ELEM(xb[i].x, db[i].x, gm[i].x, dg[i].x, mb, rb, s1b, s2b, two)   LNB_ELEM(xb[i].y, db[i].y, gm[i].y, dg[i].y, mb, rb, s1b, s2b, two)
      LNB_ELEM(xb[i].z, db[i].z, gm[i].z, dg[i].z, mb, rb, s1b, s2b, two)   LNB_ELEM(xb[i].w, db[i].w, gm[i].w, dg[i].w, mb, rb, s1b, s2b, two)
    }
#undef LNB_ELEM
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s1a += __shfl_xor(s1a, o, WAVE); s2a += __shfl_xor(s2a, o, WAVE); s1b += __shfl_xor(s1b, o, WAVE); s2b += __shfl_xor(s2b, o, WAVE);
    }
    s1a *= inv_cols; s2a *= inv_cols; s1b *= inv_cols; s2b *= inv_cols;
#pragma unroll
    for (int i = 0; i < NI; i++) {
      const int c = (lane + 64 * i) * 4;
      float4 o;
      o.x = ra * (da[i].x - s1a - xa[i].x * s2a); o.y = ra * (da[i].y - s1a - xa[i].y * s2a);
      o.z = ra * (da[i].z - s1a - xa[i].z * s2a); o.w = ra * (da[i].w - s1a - xa[i].w * s2a);
      if (dx) *reinterpret_cast<float4*>(dx + r0 * lddx + c) = o;
      if (dx_bf16) { uint2 pk; pk.x = pack2bf(o.x, o.y); pk.y = pack2bf(o.z, o.w); *reinterpret_cast<uint2*>(dx_bf16 + r0 * ld_bf16 + c) = pk; }
      if (two) {
        o.x = rb * (db[i].x - s1b - xb[i].x * s2b); o.y = rb * (db[i].y - s1b - xb[i].y * s2b);
        o.z = rb * (db[i].z - s1b - xb[i].z * s2b); o.w = rb * (db[i].w - s1b - xb[i].w * s2b);
        if (dx) *reinterpret_cast<float4*>(dx + r1 * lddx + c) = o;
        if (dx_bf16) { uint2 pk; pk.x = pack2bf(o.x, o.y); pk.y = pack2bf(o.z, o.w); *reinterpret_cast<uint2*>(dx_bf16 + r1 * ld_bf16 + c) = pk; }
      }
    }
  }
  // block reduction of the dgamma partials through LDS, then one atomic per column
  __shared__ float red[4][1024];
#pragma unroll
  for (int i = 0; i < NI; i++) *reinterpret_cast<float4*>(&red[wave][(lane + 64 * i) * 4]) = dg[i];
  __syncthreads();
  for (int c = threadIdx.x; c < cols; c += 256) {
    const float t = red[0][c] + red[1][c] + red[2][c] + red[3][c];
    if (t != 0.f) atomicAdd(dgamma + c, t);
  }
}

// Parameter gradients only (no dx wanted: an encoder's input norm, encoders.py:189).  dgamma / dbeta are column sums over the
// rows: a thread owns a column and every (256 / CW)-th row of its workgroup's slab (CW = 64 / 128 / 256 columns per
// workgroup: a 74-column norm keeps 2 x 74 of 256 threads busy instead of 74), coalesced along the columns, no wavefront
// reductions; the row sub-lanes are added through LDS, then one atomic per column (the row form spent 33 us on 12,000 x 74 values).
template <int CW>
__global__ __launch_bounds__(256) void ln_bwd_params_kernel(const float* __restrict__ dy, int64_t ldy, int64_t y_bstride, int64_t period,
                                                             const float* __restrict__ x, int64_t ldx, const float* __restrict__ mean_in,
                                                             const float* __restrict__ rstd_in, const uint8_t* __restrict__ rowmask,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t rows, int cols,
                                                             int64_t rows_per_block) {
  constexpr int RY = 256 / CW;
  __shared__ float red[2][RY][CW];
  const int tx = threadIdx.x % CW, ty = threadIdx.x / CW;
  const int c = blockIdx.x * CW + tx;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  int64_t r1 = r0 + rows_per_block; if (r1 > rows) r1 = rows;
  float dg = 0.f, db = 0.f;
  if (c < cols) {
#pragma unroll 4
    for (int64_t row = r0 + ty; row < r1; row += RY) {          // (no branch on the mask: the loads of four rows stay in flight together)
      const bool masked = rowmask && rowmask[row];          // selects, not products: a non-finite value in a padded row adds an exact 0
      const float* dyr = period > 0 ? dy + (row / period) * y_bstride + (row % period) * ldy : dy + row * ldy;
      const float dv_ = dyr[c], xv_ = (x[row * ldx + c] - mean_in[row]) * rstd_in[row];
      const float d = masked ? 0.f : dv_, xh = masked ? 0.f : xv_;
      dg += d * xh;
      db += d;
    }
  }
  if (RY > 1) {
    red[0][ty][tx] = dg; red[1][ty][tx] = db;
    __syncthreads();
    if (ty != 0) return;
    dg = 0.f; db = 0.f;
#pragma unroll
    for (int k = 0; k < RY; k++) { dg += red[0][k][tx]; db += red[1][k][tx]; }          // fixed order
  }
  if (c >= cols) return;
  if (dgamma && dg != 0.f) atomicAdd(dgamma + c, dg);
  if (dbeta && db != 0.f) atomicAdd(dbeta + c, db);
}

extern "C" int mca_layernorm_bwd(const float* dy, int64_t ldy, int64_t y_bstride, int64_t period,
                                 const float* x, int64_t ldx, const float* gamma,
                                 const float* mean, const float* rstd, const uint8_t* rowmask,
                                 float* dx, int64_t lddx, uint16_t* dx_bf16, int64_t ld_bf16,
                                 float* dgamma, float* dbeta, float* dxsum, int64_t rows, int cols, mca_stream_t stream) {
  if (!dy || !x || !gamma || !mean || !rstd || rows < 0 || cols <= 0 || cols > 1024) return MCA_E_BADARG;
  if (rows == 0) return MCA_OK;
  if (!dx && !dx_bf16 && !dxsum && mca_knobs[12] != 1) {          // parameter gradients only
    if (!dgamma && !dbeta) return MCA_OK;
    const int cw = cols <= 64 ? 64 : (cols <= 128 ? 128 : 256);
    const int chunks = (cols + cw - 1) / cw;
    int64_t slabs = 1024 / chunks;          // four rounds of workgroups at most (each ends with one atomic per column)
    if (slabs > (rows + 31) / 32) slabs = (rows + 31) / 32;
    if (slabs < 1) slabs = 1;
    const int64_t rpb = (rows + slabs - 1) / slabs;
    const dim3 grid(chunks, (unsigned)((rows + rpb - 1) / rpb));
#define LNP_LAUNCH(CW) hipLaunchKernelGGL(ln_bwd_params_kernel<CW>, grid, dim3(256), 0, as_stream(stream), dy, ldy, y_bstride, period, x, ldx, mean, rstd, \
                                          rowmask, dgamma, dbeta, rows, cols, rpb)
    if (cw == 64) LNP_LAUNCH(64); else if (cw == 128) LNP_LAUNCH(128); else LNP_LAUNCH(256);
#undef LNP_LAUNCH
    return launch_status();
  }
  const bool vec = (cols % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && (y_bstride % 4 == 0) &&
                   (!dx || lddx % 4 == 0) && (!dx_bf16 || ld_bf16 % 4 == 0) &&
                   ((uintptr_t)x % 16 == 0) && ((uintptr_t)dy % 16 == 0) && (!dx || (uintptr_t)dx % 16 == 0);
  if (vec && !rowmask && period <= 0 && !dbeta && !dxsum && dgamma && (cols == 256 || cols == 512 || cols == 1024) &&
      (uintptr_t)gamma % 16 == 0 && mca_knobs[12] != 1) {          // knob 12 = 1: general kernel (A/B)
    // one workgroup per CU at most: every workgroup ends with one atomic per column on dgamma, and 1024 of them on the same
    // 512 addresses cost more than the extra loads in flight bring (b = 8: 40.8 -> 28.5 us, b = 32: 108.7 -> 104.0 us;
    // knob 14 = another cap, tools/bench_ln.py)
    int64_t nb = (rows + 7) / 8;
    const int64_t cap = mca_knobs[14] > 0 ? mca_knobs[14] : 256;
    if (nb > cap) nb = cap;
    if (cols == 256) hipLaunchKernelGGL(ln_bwd_trunk_kernel<1>, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), dy, ldy, x, ldx, gamma, mean, rstd, dx, lddx, dx_bf16, ld_bf16, dgamma, rows, cols);
    else if (cols == 512) hipLaunchKernelGGL(ln_bwd_trunk_kernel<2>, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), dy, ldy, x, ldx, gamma, mean, rstd, dx, lddx, dx_bf16, ld_bf16, dgamma, rows, cols);
    else hipLaunchKernelGGL(ln_bwd_trunk_kernel<4>, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), dy, ldy, x, ldx, gamma, mean, rstd, dx, lddx, dx_bf16, ld_bf16, dgamma, rows, cols);
    return launch_status();
  }
  int64_t blocks = (rows + 3) / 4;
  // every workgroup ends with one atomic per column on the same dgamma / dbeta / dxsum addresses, and a wavefront has one row in
  // flight: few workgroups at few rows, up to four per CU at many (measured, tools/bench_ln_encoder.py: 3,600 rows 14.5 us at
  // 256 against 30.7 at 1,024; 48,000 rows 134 us at 256 against 70 at 1,024)
  int64_t bcap = rows / 32;
  if (bcap < 256) bcap = 256;
  if (bcap > 1024) bcap = 1024;
  if (mca_knobs[14] > 0) bcap = mca_knobs[14];
  if (blocks > bcap) blocks = bcap;
  if (vec)
    hipLaunchKernelGGL(ln_bwd_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), dy, ldy, y_bstride,
                       period, x, ldx, gamma, mean, rstd, rowmask, dx, lddx, dx_bf16, ld_bf16, dgamma, dbeta, dxsum, rows, cols);
  else
    hipLaunchKernelGGL(ln_bwd_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), dy, ldy, y_bstride,
                       period, x, ldx, gamma, mean, rstd, rowmask, dx, lddx, dx_bf16, ld_bf16, dgamma, dbeta, dxsum, rows, cols);
  return launch_status();
}

// =====================================================================================================
// GEGLU (model.py:35-38).  h = [a | gate], g = gelu_erf(gate) * a.  8 bf16 per lane per access.
// =====================================================================================================
__global__ __launch_bounds__(256) void geglu_fwd_kernel(const u16* __restrict__ h, u16* __restrict__ g,
                                                         int64_t rows, int ip) {
  const int chunks = ip / 8;
  const int64_t total = rows * chunks;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t r = idx / chunks;
    const int c = (int)(idx % chunks) * 8;
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(h + r * 2 * ip + c);
    const bf16x8 gt = *reinterpret_cast<const bf16x8*>(h + r * 2 * ip + ip + c);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = (short)f2bf(gelu_erf(bf2f((u16)gt[j])) * bf2f((u16)a[j]));
    *reinterpret_cast<bf16x8*>(g + r * ip + c) = o;
  }
}

__global__ __launch_bounds__(256) void geglu_bwd_kernel(const u16* __restrict__ dg, const u16* __restrict__ h,
                                                         u16* __restrict__ dh, int64_t rows, int ip) {
  const int chunks = ip / 8;
  const int64_t total = rows * chunks;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t r = idx / chunks;
    const int c = (int)(idx % chunks) * 8;
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(h + r * 2 * ip + c);
    const bf16x8 gt = *reinterpret_cast<const bf16x8*>(h + r * 2 * ip + ip + c);
    const bf16x8 d = *reinterpret_cast<const bf16x8*>(dg + r * ip + c);
    bf16x8 da, dgt;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const float gv = bf2f((u16)gt[j]), av = bf2f((u16)a[j]), dv = bf2f((u16)d[j]);
      da[j] = (short)f2bf(dv * gelu_erf(gv));
      dgt[j] = (short)f2bf(dv * av * gelu_erf_grad(gv));
    }
    *reinterpret_cast<bf16x8*>(dh + r * 2 * ip + c) = da;
    *reinterpret_cast<bf16x8*>(dh + r * 2 * ip + ip + c) = dgt;
  }
}

static inline unsigned stream_grid(int64_t items) {
  int64_t b = (items + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (unsigned)b;
}

extern "C" int mca_geglu_fwd(const uint16_t* h, uint16_t* g, int64_t rows, int ip, mca_stream_t stream) {
  if (!h || !g || rows < 0 || ip <= 0 || ip % 8) return MCA_E_BADARG;
  if (rows == 0) return MCA_OK;
  hipLaunchKernelGGL(geglu_fwd_kernel, dim3(stream_grid(rows * (ip / 8))), dim3(256), 0, as_stream(stream), h, g, rows, ip);
  return launch_status();
}
extern "C" int mca_geglu_bwd(const uint16_t* dg, const uint16_t* h, uint16_t* dh, int64_t rows, int ip,
                             mca_stream_t stream) {
  if (!dg || !h || !dh || rows < 0 || ip <= 0 || ip % 8) return MCA_E_BADARG;
  if (rows == 0) return MCA_OK;
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(stream_grid(rows * (ip / 8))), dim3(256), 0, as_stream(stream), dg, h, dh, rows, ip);
  return launch_status();
}

// =====================================================================================================
// casts / broadcast / reductions
// =====================================================================================================
__global__ __launch_bounds__(256) void cast_pad_kernel(const float* __restrict__ src, int64_t lds, int64_t rows,
                                                        int64_t cols, u16* __restrict__ dst, int64_t ldd,
                                                        int64_t rows_pad, int64_t cols_pad, int transpose) {
  // dst is (rows_pad x cols_pad); element (r, c) of dst comes from src[r][c] or src[c][r]
  const int64_t total = rows_pad * cols_pad;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t r = idx / cols_pad, c = idx % cols_pad;
    float v = 0.f;
    if (!transpose) { if (r < rows && c < cols) v = src[r * lds + c]; }
    else { if (c < rows && r < cols) v = src[c * lds + r]; }
    dst[r * ldd + c] = f2bf(v);
  }
}
extern "C" int mca_cast_pad_bf16(const float* src, int64_t lds, int64_t rows, int64_t cols, uint16_t* dst,
                                 int64_t ldd, int64_t rows_pad, int64_t cols_pad, int transpose,
                                 mca_stream_t stream) {
  if (!src || !dst || rows <= 0 || cols <= 0 || rows_pad <= 0 || cols_pad <= 0) return MCA_E_BADARG;
  if (!transpose && (rows_pad < rows || cols_pad < cols)) return MCA_E_BADARG;
  if (transpose && (rows_pad < cols || cols_pad < rows)) return MCA_E_BADARG;
  hipLaunchKernelGGL(cast_pad_kernel, dim3(stream_grid(rows_pad * cols_pad)), dim3(256), 0, as_stream(stream), src, lds,
                     rows, cols, dst, ldd, rows_pad, cols_pad, transpose);
  return launch_status();
}

// multi-tensor form of the above: one launch refreshes every bf16 weight copy (descriptor table in device memory).
// 64x64 tiles through LDS: source rows are read as 256-byte runs and destination rows written as 128-byte runs for the
// plain AND the transposed copies (element-wise transposed reads fetched 435 MB for 70 MB of parameters).
__global__ __launch_bounds__(256) void cast_pad_multi_kernel(const mca_cast_desc* __restrict__ descs) {
  __shared__ float tile[64][65];
  const mca_cast_desc d = descs[blockIdx.y];
  const float* src = reinterpret_cast<const float*>(d.src);
  u16* dst = reinterpret_cast<u16*>(d.dst);
  const int tid = threadIdx.x;
  const int64_t tiles_c = (d.cols_pad + 63) / 64, tiles_r = (d.rows_pad + 63) / 64;
  // source extent in DESTINATION coordinates: plain (rows x cols), transposed (cols x rows)
  const int64_t vr = d.transpose ? d.cols : d.rows, vc = d.transpose ? d.rows : d.cols;
  const float sc = d.scale == 0.f ? 1.f : d.scale;
  for (int64_t t = blockIdx.x; t < tiles_r * tiles_c; t += gridDim.x) {
    const int64_t r0 = (t / tiles_c) * 64, c0 = (t % tiles_c) * 64;          // destination tile origin
#pragma unroll 4
    for (int i = 0; i < 16; i++) {
      const int a = i * 4 + (tid >> 6), bq = tid & 63;          // source-major walk: `bq` runs along a source row
      float v = 0.f;
      if (!d.transpose) { if (r0 + a < vr && c0 + bq < vc) v = src[(r0 + a) * d.lds + c0 + bq] * sc; tile[a][bq] = v; }
      else { if (c0 + a < vc && r0 + bq < vr) v = src[(c0 + a) * d.lds + r0 + bq] * sc; tile[bq][a] = v; }          // tile[dst row][dst col]
    }
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < 8; i++) {
      const int r = i * 8 + (tid >> 5), c = (tid & 31) * 2;
      if (r0 + r < d.rows_pad) {
        u16* o = dst + (r0 + r) * d.ldd + c0 + c;
        if (c0 + c + 1 < d.cols_pad && (((uintptr_t)o) & 3) == 0) *reinterpret_cast<uint32_t*>(o) = pack2bf(tile[r][c], tile[r][c + 1]);
        else {
          if (c0 + c < d.cols_pad) o[0] = f2bf(tile[r][c]);
          if (c0 + c + 1 < d.cols_pad) o[1] = f2bf(tile[r][c + 1]);
        }
      }
    }
    __syncthreads();
  }
}
extern "C" int mca_cast_pad_bf16_multi(const mca_cast_desc* descs_dev, int n, mca_stream_t stream) {
  if (!descs_dev || n <= 0 || n > 65535) return MCA_E_BADARG;
  hipLaunchKernelGGL(cast_pad_multi_kernel, dim3(48, n), dim3(256), 0, as_stream(stream), descs_dev);
  return launch_status();
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, int64_t lds,
                                                           u16* __restrict__ dst, int64_t ldd, int64_t rows,
                                                           int64_t cols4, float scale) {
  const int64_t total = rows * cols4;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t r = idx / cols4, c = (idx % cols4) * 4;
    const float4 t = *reinterpret_cast<const float4*>(src + r * lds + c);
    uint2 pk; pk.x = pack2bf(t.x * scale, t.y * scale); pk.y = pack2bf(t.z * scale, t.w * scale);
    *reinterpret_cast<uint2*>(dst + r * ldd + c) = pk;
  }
}
extern "C" int mca_f32_to_bf16(const float* src, int64_t lds, uint16_t* dst, int64_t ldd, int64_t rows,
                               int64_t cols, float scale, mca_stream_t stream) {
  if (!src || !dst || rows < 0 || cols <= 0) return MCA_E_BADARG;
  if (cols % 4 || lds % 4 || ldd % 4 || (uintptr_t)src % 16 || (uintptr_t)dst % 8) return MCA_E_ALIGN;
  if (rows == 0) return MCA_OK;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(stream_grid(rows * cols / 4)), dim3(256), 0, as_stream(stream), src, lds,
                     dst, ldd, rows, cols / 4, scale);
  return launch_status();
}

__global__ __launch_bounds__(256) void bcast_rows_kernel(const float* __restrict__ src, int64_t lds,
                                                          float* __restrict__ dst, int64_t ldd, int64_t dst_bstride,
                                                          int64_t period, int64_t rows, int cols) {
  const int64_t total = rows * cols;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t i = idx / cols; const int c = (int)(idx % cols);
    dst[(i / period) * dst_bstride + (i % period) * ldd + c] = src[(i % period) * lds + c];
  }
}
extern "C" int mca_bcast_rows(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t dst_bstride,
                              int64_t period, int64_t rows, int cols, mca_stream_t stream) {
  if (!src || !dst || rows < 0 || cols <= 0 || period <= 0) return MCA_E_BADARG;
  if (rows == 0) return MCA_OK;
  hipLaunchKernelGGL(bcast_rows_kernel, dim3(stream_grid(rows * cols)), dim3(256), 0, as_stream(stream), src, lds, dst,
                     ldd, dst_bstride, period, rows, cols);
  return launch_status();
}

// dst[(i % period), c] += sum_i src[...]: each block owns a slab of rows, one atomic per (prow, c) per block
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ src, int64_t lds,
                                                           int64_t src_bstride, int64_t period,
                                                           float* __restrict__ dst, int64_t ldd, int64_t rows,
                                                           int cols, int64_t groups_per_block) {
  // grid.x = column chunks of 256, grid.y = prow, grid.z = slabs over the i/period ("group") index
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int64_t prow = blockIdx.y;
  const int64_t ngroups = (rows + period - 1) / period;
  const int64_t g0 = (int64_t)blockIdx.z * groups_per_block;
  int64_t g1 = g0 + groups_per_block; if (g1 > ngroups) g1 = ngroups;
  if (c >= cols) return;
  float acc = 0.f;
#pragma unroll 4
  for (int64_t g = g0; g < g1; g++) {
    if (g * period + prow < rows) acc += src[g * src_bstride + prow * lds + c];
  }
  if (acc != 0.f) atomicAdd(dst + prow * ldd + c, acc);
}
extern "C" int mca_reduce_rows(const float* src, int64_t lds, int64_t src_bstride, int64_t period, float* dst,
                               int64_t ldd, int64_t rows, int cols, mca_stream_t stream) {
  if (!src || !dst || rows < 0 || cols <= 0 || period <= 0) return MCA_E_BADARG;
  if (rows == 0) return MCA_OK;
  const int64_t ngroups = (rows + period - 1) / period;
  // Two rounds of workgroups at most: every slab ends with one atomic per (prow, column), and the slabs of one column all hit
  // the same address (a bias gradient over 12,000 rows in 1,024 slabs: 38 us for 25 MB, all of it same-address atomics; knob 14
  // = another number of workgroups, tools/bench_ln.py)
  const int64_t per_slab = period * ((cols + 255) / 256);
  const int64_t wgs = mca_knobs[14] > 0 ? mca_knobs[14] : 512;
  int64_t slabs = wgs / per_slab > 0 ? wgs / per_slab : 1;
  if (slabs > ngroups) slabs = ngroups;
  const int64_t gpb = (ngroups + slabs - 1) / slabs;
  if (period > 65535) return MCA_E_UNSUPPORTED;
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((cols + 255) / 256, (unsigned)period, (unsigned)slabs), dim3(256), 0,
                     as_stream(stream), src, lds, src_bstride, period, dst, ldd, rows, cols, gpb);
  return launch_status();
}

// =====================================================================================================
// TabularEncoder pieces (encoders.py:17-96)
// =====================================================================================================
// nn.Embedding(max_norm): rows with L2 norm > max_norm are rescaled IN PLACE by max_norm / (norm + 1e-7)
__global__ __launch_bounds__(256) void embedding_renorm_kernel(float* __restrict__ w, int64_t rows, int cols, float max_norm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += (int64_t)gridDim.x * 4) {
    float* wr = w + r * cols;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += wr[c] * wr[c];
    const float nrm = sqrtf(wave_sum(s));
    if (nrm > max_norm) {
      const float sc = max_norm / (nrm + 1e-7f);
      for (int c = lane; c < cols; c += 64) wr[c] *= sc;
    }
  }
}
extern "C" int mca_embedding_renorm(float* weight, int64_t rows, int cols, float max_norm, mca_stream_t stream) {
  if (!weight || rows <= 0 || cols <= 0) return MCA_E_BADARG;
  int64_t blocks = (rows + 3) / 4; if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(embedding_renorm_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), weight, rows, cols, max_norm);
  return launch_status();
}

// h1[r][d] = relu(min(x[r], max_value) * w1[d] + b1[d])  (Linear(1, D) + ReLU), bf16 out; padmask[r] = (x[r] == padding_value)
__global__ __launch_bounds__(256) void tab_value_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                             const float* __restrict__ b1, u16* __restrict__ h1,
                                                             uint8_t* __restrict__ padmask, int64_t rows, int cols,
                                                             float max_value, float padding_value) {
  const int chunks = cols / 8;
  const int64_t total = rows * chunks;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t r = idx / chunks;
    const int c = (int)(idx % chunks) * 8;
    const float xv = x[r];
    const float xc = fminf(xv, max_value);
    if (c == 0) padmask[r] = xv == padding_value ? 1 : 0;
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = (short)f2bf(fmaxf(xc * w1[c + j] + b1[c + j], 0.f));
    *reinterpret_cast<bf16x8*>(h1 + r * cols + c) = o;
  }
}
extern "C" int mca_tab_value_fwd(const float* x, const float* w1, const float* b1, uint16_t* h1, uint8_t* padmask, int64_t rows,
                                 int cols, float max_value, float padding_value, mca_stream_t stream) {
  if (!x || !w1 || !b1 || !h1 || !padmask || rows <= 0 || cols <= 0 || cols % 8) return MCA_E_BADARG;
  hipLaunchKernelGGL(tab_value_fwd_kernel, dim3(stream_grid(rows * (cols / 8))), dim3(256), 0, as_stream(stream), x, w1, b1, h1,
                     padmask, rows, cols, max_value, padding_value);
  return launch_status();
}

// dw1[d] += sum_r dh1[r][d] * [h1[r][d] > 0] * min(x[r], max_value);  db1[d] += sum_r dh1[r][d] * [h1 > 0]
__global__ __launch_bounds__(256) void tab_value_bwd_kernel(const float* __restrict__ dh1, int64_t ld, const u16* __restrict__ h1,
                                                             const float* __restrict__ x, float* __restrict__ dw1,
                                                             float* __restrict__ db1, int64_t rows, int cols, float max_value,
                                                             int64_t rows_per_block) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  int64_t r1 = r0 + rows_per_block; if (r1 > rows) r1 = rows;
  if (c >= cols) return;
  float aw = 0.f, ab = 0.f;
  for (int64_t r = r0; r < r1; r++) {
    const float g = bf2f(h1[r * cols + c]) > 0.f ? dh1[r * ld + c] : 0.f;
    aw += g * fminf(x[r], max_value);
    ab += g;
  }
  atomicAdd(dw1 + c, aw);
  atomicAdd(db1 + c, ab);
}
extern "C" int mca_tab_value_bwd(const float* dh1, int64_t ld, const uint16_t* h1, const float* x, float* dw1, float* db1,
                                 int64_t rows, int cols, float max_value, mca_stream_t stream) {
  if (!dh1 || !h1 || !x || !dw1 || !db1 || rows <= 0 || cols <= 0) return MCA_E_BADARG;
  int64_t slabs = rows < 512 ? rows : 512;
  const int64_t rpb = (rows + slabs - 1) / slabs;
  slabs = (rows + rpb - 1) / rpb;
  hipLaunchKernelGGL(tab_value_bwd_kernel, dim3((cols + 255) / 256, (unsigned)slabs), dim3(256), 0, as_stream(stream), dh1, ld, h1, x,
                     dw1, db1, rows, cols, max_value, rpb);
  return launch_status();
}


// =====================================================================================================
// mask packing: one workgroup per sample walks the modalities (see include/mca_hip.h)
// =====================================================================================================
__global__ __launch_bounds__(256) void pack_masks_kernel(mca_pack_masks_args a, uint8_t* __restrict__ padding, int32_t* __restrict__ present) {
  __shared__ int any_valid[MCA_MAX_MODALITIES];
  const int s = blockIdx.x, tid = threadIdx.x;
  if (tid < MCA_MAX_MODALITIES) any_valid[tid] = 0;
  __syncthreads();
  uint8_t* prow = padding + (int64_t)s * a.n_tokens;
  for (int i = 0; i < a.n_mod; i++) {
    const mca_mask_desc d = a.m[i];
    bool valid = false;
    for (int j = tid; j < d.n; j += 256) {
      const int64_t idx = (int64_t)s * d.n + j;
      const bool pad = d.elem_bytes == 8 ? reinterpret_cast<const int64_t*>(d.mask)[idx] != 0 : reinterpret_cast<const uint8_t*>(d.mask)[idx] != 0;
      prow[d.offset + j] = pad ? 1 : 0;
      if (d.rowmask) d.rowmask[idx] = pad ? 1 : 0;
      valid |= !pad;
    }
    if (__any(valid) && (tid & 63) == 0) any_valid[i] = 1;          // benign race: every writer stores 1
  }
  for (int j = a.n_tokens - a.n_fusion + tid; j < a.n_tokens; j += 256) prow[j] = 0;
  __syncthreads();
  if (tid == 0) {
    int bits = 0;
    for (int i = 0; i < a.n_mod; i++) bits |= any_valid[i] << i;
    present[s] = bits;
  }
}

extern "C" int mca_pack_masks(const mca_pack_masks_args* args, uint8_t* padding, int32_t* present, mca_stream_t stream) {
  if (!args || !padding || !present) return MCA_E_BADARG;
  if (args->n_mod <= 0 || args->n_mod > MCA_MAX_MODALITIES || args->batch <= 0 || args->n_tokens <= 0 || args->n_fusion < 0) return MCA_E_BADARG;
  int64_t tot = args->n_fusion;
  for (int i = 0; i < args->n_mod; i++) {
    const mca_mask_desc& d = args->m[i];
    if (!d.mask || d.n <= 0 || d.offset < 0 || (d.elem_bytes != 1 && d.elem_bytes != 8)) return MCA_E_BADARG;
    if (d.elem_bytes == 8 && (uintptr_t)d.mask % 8) return MCA_E_ALIGN;
    if ((int64_t)d.offset + d.n > args->n_tokens) return MCA_E_BADARG;
    tot += d.n;
  }
  if (tot > args->n_tokens) return MCA_E_BADARG;
  hipLaunchKernelGGL(pack_masks_kernel, dim3(args->batch), dim3(256), 0, as_stream(stream), *args, padding, present);
  return launch_status();
}


// =====================================================================================================
// Device-side finite check (encoders.py:197-213: the reference raises on non-finite encoder inputs / outputs after
// `.sum()` host syncs; here the tensors are scanned on the device and a bit is OR-ed into one flag word the host reads
// once per step, and that mca_adamw_step honours so a bad step never reaches the weights).
// grid (blocks, n tensors): block row y scans tensor y with 16-byte loads where the pointer allows.
// =====================================================================================================
__global__ __launch_bounds__(256) void nonfinite_flag_kernel(mca_finite_args a, int32_t* __restrict__ flag, int bit) {
  const float* __restrict__ p = a.p[blockIdx.y];
  const int64_t n = a.n[blockIdx.y];
  bool bad = false;
  const bool vec = ((uintptr_t)p & 15) == 0;
  const int64_t n4 = vec ? n / 4 : 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const uint4 v = reinterpret_cast<const uint4*>(p)[i];
    bad |= ((v.x & 0x7f800000u) == 0x7f800000u) | ((v.y & 0x7f800000u) == 0x7f800000u) | ((v.z & 0x7f800000u) == 0x7f800000u) |
           ((v.w & 0x7f800000u) == 0x7f800000u);
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    bad |= (__float_as_uint(p[i]) & 0x7f800000u) == 0x7f800000u;
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, bit);
}
// the flag word, written by a kernel straight into PINNED host memory (device-visible): a copy NODE of a captured step is not
// ordered reliably against the kernels around it (see mca_attn_bwd_prep), a kernel is
__global__ void flag_to_host_kernel(const int32_t* __restrict__ flag, volatile int32_t* __restrict__ host_word) {
  if (threadIdx.x == 0) { *host_word = *flag; __threadfence_system(); }
}
extern "C" int mca_flag_to_host(const int32_t* flag, int32_t* host_pinned, mca_stream_t stream) {
  if (!flag || !host_pinned) return MCA_E_BADARG;
  hipLaunchKernelGGL(flag_to_host_kernel, dim3(1), dim3(64), 0, as_stream(stream), flag, host_pinned);
  return launch_status();
}
extern "C" int mca_nonfinite_flag(const mca_finite_args* args, int32_t* flag, int bit, mca_stream_t stream) {
  if (!args || !flag || args->count <= 0 || args->count > MCA_MAX_MODALITIES || bit == 0) return MCA_E_BADARG;
  int64_t nmax = 0;
  for (int i = 0; i < args->count; i++) {
    if (!args->p[i] || args->n[i] < 0) return MCA_E_BADARG;
    if ((uintptr_t)args->p[i] % 4) return MCA_E_ALIGN;
    if (args->n[i] > nmax) nmax = args->n[i];
  }
  if (nmax == 0) return MCA_OK;
  int64_t blocks = (nmax / 4 + 255) / 256; if (blocks > 1024) blocks = 1024; if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(nonfinite_flag_kernel, dim3((unsigned)blocks, (unsigned)args->count), dim3(256), 0, as_stream(stream), *args, flag, bit);
  return launch_status();
}

// =====================================================================================================
// EAO baseline ("everything at once", reference model.py:481-596): every modality alone and every modality combination is one
// SEGMENT of a super-sequence (block-diagonal attention = the key-group masks of the fusion kernels); the token blocks of a
// modality are replicated into its segments, and the pooled token of a segment is the mean of its un-padded rows
// (MeanTokenProjectionPool without token types and without projection, model.py:255-276).
// =====================================================================================================
// dst[b, r, :] (+)= src[b, r, :]   rows x cols contiguous per sample (16-byte pieces)
__global__ __launch_bounds__(256) void rows_copy_add_kernel(const float4* __restrict__ src, int64_t src_bstride4, float4* __restrict__ dst,
                                                             int64_t dst_bstride4, int64_t per_sample4, int64_t total4, int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / per_sample4, r = i - b * per_sample4;
    float4 v = src[b * src_bstride4 + r];
    float4* d = dst + b * dst_bstride4 + r;
    if (accumulate) { const float4 o = *d; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
    *d = v;
  }
}
extern "C" int mca_rows_copy_add(const float* src, int64_t src_bstride, float* dst, int64_t dst_bstride, int64_t rows, int cols,
                                 int batch, int accumulate, mca_stream_t stream) {
  if (!src || !dst || rows < 0 || cols <= 0 || batch <= 0) return MCA_E_BADARG;
  if (cols % 4 || src_bstride % 4 || dst_bstride % 4 || (uintptr_t)src % 16 || (uintptr_t)dst % 16) return MCA_E_ALIGN;
  if (rows == 0) return MCA_OK;
  const int64_t per4 = rows * cols / 4, total4 = per4 * batch;
  hipLaunchKernelGGL(rows_copy_add_kernel, dim3(stream_grid(total4)), dim3(256), 0, as_stream(stream), reinterpret_cast<const float4*>(src),
                     src_bstride / 4, reinterpret_cast<float4*>(dst), dst_bstride / 4, per4, total4, accumulate);
  return launch_status();
}

// out[b, s, :] = mean over the un-padded rows of segment s (zeros when there is none: model.py:267-268); counts[b, s] = their
// number.  grid (segments, batch), 512 threads = 8 row groups x 64 lanes (4 columns each, columns strided by 256); the row
// groups meet in LDS in a fixed order: bitwise reproducible.
__global__ __launch_bounds__(512) void segment_mean_fwd_kernel(const float* __restrict__ x, const uint8_t* __restrict__ padding,
                                                                const int32_t* __restrict__ seg_start, float* __restrict__ out,
                                                                int32_t* __restrict__ counts, int n_tokens, int cols) {
  __shared__ float red[8][1024];
  __shared__ int cnt_s[8];
  const int s = blockIdx.x, b = blockIdx.y, lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int r0 = seg_start[s], r1 = seg_start[s + 1];
  const uint8_t* pad = padding + (int64_t)b * n_tokens;
  const float* xb = x + (int64_t)b * n_tokens * cols;
  float4 acc[4];
#pragma unroll
  for (int k = 0; k < 4; k++) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  int cnt = 0;
  for (int r = r0 + rg; r < r1; r += 8) {
    if (pad[r]) continue;
    cnt++;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int c = lane * 4 + 256 * k;
      if (c < cols) {
        const float4 v = *reinterpret_cast<const float4*>(xb + (int64_t)r * cols + c);
        acc[k].x += v.x; acc[k].y += v.y; acc[k].z += v.z; acc[k].w += v.w;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int c = lane * 4 + 256 * k;
    if (c < cols) { red[rg][c] = acc[k].x; red[rg][c + 1] = acc[k].y; red[rg][c + 2] = acc[k].z; red[rg][c + 3] = acc[k].w; }
  }
  if (lane == 0) cnt_s[rg] = cnt;
  __syncthreads();
  int total = 0;
#pragma unroll
  for (int g = 0; g < 8; g++) total += cnt_s[g];
  const float inv = total > 0 ? 1.f / (float)total : 0.f;
  for (int c = threadIdx.x; c < cols; c += 512) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 8; g++) t += red[g][c];
    out[((int64_t)b * gridDim.x + s) * cols + c] = t * inv;
  }
  if (threadIdx.x == 0) counts[(int64_t)b * gridDim.x + s] = total;
}
extern "C" int mca_segment_mean_fwd(const float* x, const uint8_t* padding, const int32_t* seg_start, int n_seg, float* out,
                                    int32_t* counts, int batch, int n_tokens, int cols, mca_stream_t stream) {
  if (!x || !padding || !seg_start || !out || !counts || n_seg <= 0 || batch <= 0 || n_tokens <= 0 || cols <= 0) return MCA_E_BADARG;
  if (cols % 4 || (uintptr_t)x % 16) return MCA_E_ALIGN;
  if (cols > 1024 || batch > 65535) return MCA_E_UNSUPPORTED;
  hipLaunchKernelGGL(segment_mean_fwd_kernel, dim3(n_seg, batch), dim3(512), 0, as_stream(stream), x, padding, seg_start, out, counts,
                     n_tokens, cols);
  return launch_status();
}

// dx[b, r, :] = padded(b, r) ? 0 : d_out[b, seg(r), :] / counts[b, seg(r)]          one wavefront per row
__global__ __launch_bounds__(256) void segment_mean_bwd_kernel(const float* __restrict__ d_out, const uint8_t* __restrict__ padding,
                                                                const uint8_t* __restrict__ seg_of_row, const int32_t* __restrict__ counts,
                                                                int n_seg, float* __restrict__ dx, int64_t rows_total, int n_tokens, int cols) {
  const int lane = threadIdx.x & 63;
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows_total; row += (int64_t)gridDim.x * 4) {
    const int64_t b = row / n_tokens;
    const int r = (int)(row - b * n_tokens);
    const int s = seg_of_row[r];
    const bool live = padding[row] == 0;
    const float inv = live ? 1.f / (float)counts[b * n_seg + s] : 0.f;
    const float* g = d_out + (b * n_seg + s) * (int64_t)cols;
    for (int c = lane * 4; c < cols; c += 256) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live) { v = *reinterpret_cast<const float4*>(g + c); v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv; }
      *reinterpret_cast<float4*>(dx + row * cols + c) = v;
    }
  }
}
extern "C" int mca_segment_mean_bwd(const float* d_out, const uint8_t* padding, const uint8_t* seg_of_row, const int32_t* counts,
                                    int n_seg, float* dx, int batch, int n_tokens, int cols, mca_stream_t stream) {
  if (!d_out || !padding || !seg_of_row || !counts || !dx || n_seg <= 0 || batch <= 0 || n_tokens <= 0 || cols <= 0) return MCA_E_BADARG;
  if (cols % 4 || (uintptr_t)d_out % 16 || (uintptr_t)dx % 16) return MCA_E_ALIGN;
  const int64_t rows = (int64_t)batch * n_tokens;
  int64_t blocks = (rows + 3) / 4; if (blocks > 65535 * 16) blocks = 65535 * 16;
  hipLaunchKernelGGL(segment_mean_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), d_out, padding, seg_of_row, counts,
                     n_seg, dx, rows, n_tokens, cols);
  return launch_status();
}
