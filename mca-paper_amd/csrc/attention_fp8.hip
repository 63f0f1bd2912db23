// Fusion attention forward with MX-fp8 operands (BASELINE configs[4]: "fp8 MFMA attention"; reference arithmetic
// model.py:87-99): Q K^T and P V on the block-scaled matrix instruction v_mfma_scale_f32_32x32x64_f8f6f4 (OCP e4m3 elements,
// one E8M0 power-of-two scale per 32 elements along the contraction; 2x the bf16 rate per MI355X_MICROARCH.md, measured
// 4.5-4.8 PFLOP/s by tools/probe_fp8.py).  Operand layout of the instruction as probed (tools/probe_fp8.py for the pairing,
// tools/probe_fp8_scale.py for the scales): lane l holds row (A) / column (B) l & 31; its bytes 0..15 are contraction elements
// 16 (l >> 5) .. +15 (scale block 0 = K 0..31) and its bytes 16..31 elements 32 + 16 (l >> 5) .. +15 (scale block 1); byte j
// of lane half h meets byte j of the other operand's half h.  The scale of block 0 is byte 0 of the scale VGPR of lane
// (l & 31), that of block 1 of lane 32 + (l & 31).
//
//   mca_attn_quant_mxfp8   q | k | v (bf16, as the QKV GEMM wrote them; q already carries scale * log2 e) ->
//                          Q8, K8 [b][h][token][64] e4m3 + scale bytes [b][h][token][2] (blocks of 32 along d), and
//                          V8T [b][h][key tile][d][64 positions] e4m3 + scales [..][d][2] (blocks = keys 0..31 | 32..63 of the
//                          tile): V TRANSPOSED per 64-key tile, the keys inside each 32-key block PERMUTED into the order the
//                          S^T accumulators hand P to the P.V product (position p <-> key 32 (p >> 5) + 8 ((p >> 2) & 3) +
//                          4 ((p >> 4) & 1) + (p & 3)), so that the P.V A operand is two 16-byte row reads and P goes from
//                          the accumulators to the B operand with 16 conversions.
//   mca_attn_fwd_fp8       the forward kernel of attention_fwd.hip (128-query workgroups, 64-key tiles, static tile skipping,
//                          exact running maximum, uniform-row semantics) on those operands: 4 matrix instructions per
//                          32 x 64 block instead of 16, K / V tiles of 4 KiB each instead of 8, no transposed LDS reads.
//                          P is fed as 128 * 2^(S - m) (unit scale; the sum carries the same factor), e4m3 range down to
//                          2^-16 of the row maximum.
// The backward stays in bf16 (attention_bwd2.hip) with the log-sum-exp this forward wrote.
#include "common.h"

#define AQ 128
#define AK 64
#define DH 64
#define MAX_KTILES 512
typedef int v8i __attribute__((ext_vector_type(8)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------
// quantisation
// ---------------------------------------------------------------------------------------------------------
// 16 floats of one 32-element block half -> 16 e4m3 bytes with the block's power-of-two scale; amax over the 32-element block
// = this lane and its xor-1 neighbour.  Scale exponent e = floor(log2(amax)) - 7: |x| / 2^e < 256 <= 448 (no saturation
// needed; v_cvt_pk_fp8_f32 returns NaN above 448).  Returns the E8M0 byte (e + 127).
__device__ __forceinline__ uint32_t quant_block_half(const float (&x)[16], uint4& out) {
  float am = 0.f;
#pragma unroll
  for (int i = 0; i < 16; i++) am = fmaxf(am, fabsf(x[i]));
  am = fmaxf(am, __shfl_xor(am, 1, WAVE));
  const uint32_t E = (__float_as_uint(am) >> 23) & 0xffu;          // biased exponent of amax (0 for zero / subnormal)
  const uint32_t sb = E > 7u ? E - 7u : 0u;                        // E8M0 scale byte: 2^(sb - 127)
  const float mult = __uint_as_float((254u - sb) << 23);           // 2^(127 - sb)
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int v = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * i] * mult, x[4 * i + 1] * mult, 0, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * i + 2] * mult, x[4 * i + 3] * mult, v, true);
    w[i] = (uint32_t)v;
  }
  out = make_uint4(w[0], w[1], w[2], w[3]);
  return sb;
}

__global__ __launch_bounds__(256) void attn_quant_mxfp8_kernel(const u16* __restrict__ q, int64_t q_bstride, int64_t q_ld,
                                                                const u16* __restrict__ k, const u16* __restrict__ v,
                                                                int64_t kv_bstride, int64_t kv_ld, uint8_t* __restrict__ q8,
                                                                uint8_t* __restrict__ qs, uint8_t* __restrict__ k8,
                                                                uint8_t* __restrict__ ks, uint8_t* __restrict__ v8t,
                                                                uint8_t* __restrict__ vs, int heads, int n, int ntiles) {
  __shared__ u16 vt[AK][DH + 2];          // V tile, row-major, padded: the transposed gather below walks a column
  const int kt = blockIdx.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int r = tid >> 2, c = tid & 3;          // row of the tile, 16-element quarter of the 64-wide row
  const int tok = kt * AK + r;
  const bool live = tok < n;
  const int64_t rowi = ((int64_t)b * heads + h) * ((int64_t)ntiles * AK) + tok;
  // ---- q and k: blocks of 32 along d
  for (int which = 0; which < 2; which++) {
    const u16* src = which == 0 ? q + (int64_t)b * q_bstride + (int64_t)(live ? tok : 0) * q_ld + h * DH + 16 * c
                                : k + (int64_t)b * kv_bstride + (int64_t)(live ? tok : 0) * kv_ld + h * DH + 16 * c;
    float x[16];
    const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(src), a1 = *reinterpret_cast<const bf16x8*>(src + 8);
#pragma unroll
    for (int i = 0; i < 8; i++) { x[i] = live ? bf2f((u16)a0[i]) : 0.f; x[8 + i] = live ? bf2f((u16)a1[i]) : 0.f; }
    uint4 o;
    const uint32_t sb = quant_block_half(x, o);
    uint8_t* dst = (which == 0 ? q8 : k8) + rowi * DH + 16 * c;
    *reinterpret_cast<uint4*>(dst) = o;
    if ((c & 1) == 0) (which == 0 ? qs : ks)[rowi * 2 + (c >> 1)] = (uint8_t)sb;
  }
  // ---- v: transposed per tile, keys permuted, blocks of 32 along the (permuted) key positions
  {
    const u16* src = v + (int64_t)b * kv_bstride + (int64_t)(live ? tok : 0) * kv_ld + h * DH + 16 * c;
    const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(src), a1 = *reinterpret_cast<const bf16x8*>(src + 8);
#pragma unroll
    for (int i = 0; i < 8; i++) { vt[r][16 * c + i] = live ? (u16)a0[i] : (u16)0; vt[r][16 * c + 8 + i] = live ? (u16)a1[i] : (u16)0; }
  }
  __syncthreads();
  {
    const int d = r;          // this thread: row d of V^T, positions 16 c .. 16 c + 15 = 32-key block c >> 1, lane half c & 1
    float x[16];
#pragma unroll
    for (int jj = 0; jj < 16; jj++) {
      const int key = (c >> 1) * 32 + (jj >> 2) * 8 + 4 * (c & 1) + (jj & 3);
      x[jj] = bf2f(vt[key][d]);
    }
    uint4 o;
    const uint32_t sb = quant_block_half(x, o);
    const int64_t ti = (((int64_t)b * heads + h) * ntiles + kt) * DH + d;
    *reinterpret_cast<uint4*>(v8t + ti * AK + 16 * c) = o;
    if ((c & 1) == 0) vs[ti * 2 + (c >> 1)] = (uint8_t)sb;
  }
}

extern "C" int mca_attn_quant_mxfp8(const uint16_t* q, int64_t q_bstride, int64_t q_ld, const uint16_t* k, const uint16_t* v,
                                    int64_t kv_bstride, int64_t kv_ld, const mca_attn_fp8_operands* f, int batch, int heads,
                                    int n, mca_stream_t stream) {
  if (!q || !k || !v || !f || !f->q8 || !f->qs || !f->k8 || !f->ks || !f->v8t || !f->vs || batch <= 0 || heads <= 0 || n <= 0)
    return MCA_E_BADARG;
  if (q_ld % 8 || kv_ld % 8 || q_bstride % 8 || kv_bstride % 8 || (uintptr_t)q % 16 || (uintptr_t)k % 16 || (uintptr_t)v % 16) return MCA_E_ALIGN;
  if ((uintptr_t)f->q8 % 16 || (uintptr_t)f->k8 % 16 || (uintptr_t)f->v8t % 16) return MCA_E_ALIGN;
  const int ntiles = (n + AK - 1) / AK;
  if (f->n_ktiles != ntiles) return MCA_E_BADARG;
  if (heads > 65535 || batch > 65535) return MCA_E_UNSUPPORTED;
  hipLaunchKernelGGL(attn_quant_mxfp8_kernel, dim3(ntiles, heads, batch), dim3(256), 0, as_stream(stream), q, q_bstride, q_ld, k, v,
                     kv_bstride, kv_ld, f->q8, f->qs, f->k8, f->ks, f->v8t, f->vs, heads, n, ntiles);
  return launch_status();
}

// ---------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------
// [64 rows][64 bytes] tile image: 16-byte chunk c (0..3) of row r at c ^ ((r >> 2) & 3): the 16 rows of a ds_read_b128 lane
// group then cover all 64 banks (64-byte rows: four rows per 256-byte bank row)
__device__ __forceinline__ int t8_off(int r, int c) { return r * 64 + ((c ^ ((r >> 2) & 3)) << 4); }

// DMA: the K8 / V8T / scale / one-hot tiles go global -> LDS by LDS-DMA (needs the mask product's one-hot operand), the swizzle of
// the tile images applied on the source chunk; otherwise through registers (structures with more than 15 key groups)
template <bool DMA>
__global__ __launch_bounds__(256) void attn_fwd8_kernel(mca_attn_fwd_args a, mca_attn_fp8_operands f, int dbg) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * 2 * AK * DH];   // K8, V8T double-buffered: 16 KiB
  __shared__ __attribute__((aligned(16))) uint8_t sc_s[2][2][AK * 2];     // [buffer][K | V][row][half] scale bytes
  __shared__ __attribute__((aligned(16))) uint8_t kinfo[2][AK];
  __shared__ __attribute__((aligned(16))) u16 hot_s[2][AK * 16];          // one-hot key groups (mask product, attention_fwd.hip)
  __shared__ uint8_t flags_s[MAX_KTILES];
  __shared__ uint32_t live_s[MAX_KTILES];
  __shared__ int n_live_s;
  uint8_t* Ks = lds;
  uint8_t* Vs = lds + 2 * AK * DH;

  const int lin0 = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
  const int lin = (dbg & 16) ? lin0 : xcd_remap(lin0, (int)(gridDim.x * gridDim.y * gridDim.z));
  const int qt = a.q_order[lin % (int)gridDim.x];
  const int h = (lin / (int)gridDim.x) % (int)gridDim.y, b = lin / (int)(gridDim.x * gridDim.y);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int q0 = qt * AQ + wave * 32;
  int qrow = q0 + l31;
  const bool qvalid = qrow < a.nq;
  if (qrow > a.nq - 1) qrow = a.nq - 1;
  const int ntiles = f.n_ktiles;
  const int64_t bh = (int64_t)b * a.heads + h;

  // Q8 fragment (B operand of S^T = K Q^T): lane = query, bytes 16 lh .. +15 of both 32-element blocks of its row; lane half lh
  // supplies the scale of block lh
  v8i qf;
  int qscale;
  {
    const int64_t rowi = bh * ((int64_t)ntiles * AK) + qrow;
    const uint4 lo = *reinterpret_cast<const uint4*>(f.q8 + rowi * DH + 16 * lh), hi = *reinterpret_cast<const uint4*>(f.q8 + rowi * DH + 32 + 16 * lh);
    qf[0] = lo.x; qf[1] = lo.y; qf[2] = lo.z; qf[3] = lo.w; qf[4] = hi.x; qf[5] = hi.y; qf[6] = hi.z; qf[7] = hi.w;
    qscale = f.qs[rowi * 2 + lh];
  }
  const uint32_t qm = a.qmask[qrow];
  // the mask as a matrix product (mca_build_keyhot; a bf16 32x32x16 product into the same fp32 accumulators)
  const bool use_hot = DMA || a.khot != nullptr;
  bf16x8 qblk;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int g = 8 * lh + j;
    qblk[j] = (g < 15 && ((qm >> g) & 1u)) ? (short)0 : (short)0xC700;
  }
  const u16* khot_g = use_hot ? a.khot + (int64_t)b * a.nk_pad * 16 : nullptr;

  f32x16 o[2];
#pragma unroll
  for (int n = 0; n < 2; n++)
#pragma unroll
    for (int r = 0; r < 16; r++) o[n][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const uint8_t* k8b = f.k8 + bh * ((int64_t)ntiles * AK) * DH;
  const uint8_t* ksb = f.ks + bh * ((int64_t)ntiles * AK) * 2;
  const uint8_t* v8b = f.v8t + bh * (int64_t)ntiles * DH * AK;
  const uint8_t* vsb = f.vs + bh * (int64_t)ntiles * DH * 2;
  const uint8_t* kinfo_g = a.keyinfo + (int64_t)b * a.nk_pad;
  {
    const uint8_t* flags_g = a.ktile_flags + (int64_t)b * a.n_ktiles;
    for (int i = tid; i < a.n_ktiles; i += 256) flags_s[i] = flags_g[i];
  }
  __syncthreads();

  // staging: a K8 / V8T tile is 4 KiB = 256 chunks of 16 B: one of each per thread; the 128 + 128 scale bytes and the 64 key
  // group bytes by the first threads
  const int srow = tid >> 2, scol = tid & 3;
  uint4 rk, rv;
  uint32_t rsk = 0, rsv = 0, rinfo = 0;
  uint2 rhot = make_uint2(0, 0);
  auto gload = [&](int kt) {
    if (use_hot) rhot = *reinterpret_cast<const uint2*>(khot_g + (int64_t)kt * (AK * 16) + tid * 4);
    rk = *reinterpret_cast<const uint4*>(k8b + ((int64_t)kt * AK + srow) * DH + 16 * scol);
    rv = *reinterpret_cast<const uint4*>(v8b + ((int64_t)kt * DH + srow) * AK + 16 * scol);
    if (tid < 32) { rsk = *reinterpret_cast<const uint32_t*>(ksb + (int64_t)kt * AK * 2 + tid * 4); rsv = *reinterpret_cast<const uint32_t*>(vsb + (int64_t)kt * DH * 2 + tid * 4); }
    if (!use_hot && tid < 16) rinfo = *reinterpret_cast<const uint32_t*>(kinfo_g + kt * AK + tid * 4);
  };
  auto swrite = [&](int buf) {
    *reinterpret_cast<uint4*>(Ks + buf * AK * DH + t8_off(srow, scol)) = rk;
    *reinterpret_cast<uint4*>(Vs + buf * AK * DH + t8_off(srow, scol)) = rv;
    if (tid < 32) { *reinterpret_cast<uint32_t*>(&sc_s[buf][0][tid * 4]) = rsk; *reinterpret_cast<uint32_t*>(&sc_s[buf][1][tid * 4]) = rsv; }
    if (use_hot) *reinterpret_cast<uint2*>(&hot_s[buf][tid * 4]) = rhot;
    else if (tid < 16) *reinterpret_cast<uint32_t*>(&kinfo[buf][tid * 4]) = rinfo;
  };

  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto stage_dma = [&](int kt, int buf) {
    const int sw = scol ^ ((srow >> 2) & 3);          // t8_off: LDS slot scol of row srow holds source chunk scol ^ swizzle(row)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(k8b + ((int64_t)kt * AK + srow) * DH + 16 * sw),
                                     (__attribute__((address_space(3))) void*)(Ks + buf * AK * DH + wave_u * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(v8b + ((int64_t)kt * DH + srow) * AK + 16 * sw),
                                     (__attribute__((address_space(3))) void*)(Vs + buf * AK * DH + wave_u * 1024), 16, 0, 0);
    if (wave_u < 2)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(khot_g + (int64_t)kt * (AK * 16) + tid * 8),
                                       (__attribute__((address_space(3))) void*)(&hot_s[buf][wave_u * 512]), 16, 0, 0);
    if (wave_u == 2) {          // 128 + 128 scale bytes: lanes 0..31 the K scales, 32..63 the V scales, 4 bytes each
      const uint8_t* sp = (lane < 32 ? ksb + (int64_t)kt * AK * 2 : vsb + (int64_t)kt * DH * 2) + (lane & 31) * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp,
                                       (__attribute__((address_space(3))) void*)(&sc_s[buf][0][0]), 4, 0, 0);
    }
  };

  if (wave == 0) {
    const int lb = a.q_ptr[qt], le = a.q_ptr[qt + 1];
    int n = 0;
    for (int i0 = lb; i0 < le; i0 += 64) {
      const int i = i0 + lane;
      const uint32_t e = i < le ? a.q_kt[i] : 0u;
      const bool keep = i < le && flags_s[e & 0x7fffffffu] != 0;
      const unsigned long long m = __ballot(keep);
      if (keep) live_s[n + __popcll(m & ((1ull << lane) - 1ull))] = e;
      n += __popcll(m);
    }
    if (lane == 0) n_live_s = n;
  }
  __syncthreads();
  const int it_end = n_live_s;
  int it = 0, buf = 0;
  if (it < it_end) {
    if (DMA) { stage_dma((int)(live_s[0] & 0x7fffffffu), 0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    else { gload((int)(live_s[0] & 0x7fffffffu)); swrite(0); }
  }
  __syncthreads();

  while (it < it_end) {
    const uint32_t ent = live_s[it];
    const int kt = (int)(ent & 0x7fffffffu);
    const bool need_mask = (ent >> 31) == 0 || (flags_s[kt] != 2);
    const int nit = it + 1;
    if (nit < it_end) { if (DMA) stage_dma((int)(live_s[nit] & 0x7fffffffu), buf ^ 1); else gload((int)(live_s[nit] & 0x7fffffffu)); }

    const uint8_t* ks = Ks + buf * AK * DH;
    const uint8_t* vs = Vs + buf * AK * DH;
    // ---- S^T = K Q^T: one block-scaled MFMA per 32-key block (A = K8 row of this lane's key, its 16 bytes of each d block)
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
      const int row = kb * 32 + l31;
      const uint4 lo = *reinterpret_cast<const uint4*>(ks + t8_off(row, lh)), hi = *reinterpret_cast<const uint4*>(ks + t8_off(row, 2 + lh));
      v8i kf;
      kf[0] = lo.x; kf[1] = lo.y; kf[2] = lo.z; kf[3] = lo.w; kf[4] = hi.x; kf[5] = hi.y; kf[6] = hi.z; kf[7] = hi.w;
      const int kscale = sc_s[buf][0][row * 2 + lh];
#pragma unroll
      for (int r = 0; r < 16; r++) s[kb][r] = 0.f;
      s[kb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf, qf, s[kb], 0, 0, 0, kscale, 0, qscale);
    }
    if (need_mask && use_hot) {
#pragma unroll
      for (int kb = 0; kb < 2; kb++) {
        const bf16x8 hf = *reinterpret_cast<const bf16x8*>(&hot_s[buf][(kb * 32 + l31) * 16 + 8 * lh]);
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf, qblk, s[kb], 0, 0, 0);
      }
    }
    if (need_mask && !use_hot) {
#pragma unroll
      for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const uint32_t info4 = *reinterpret_cast<const uint32_t*>(&kinfo[buf][kb * 32 + 8 * g + 4 * lh]);
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const uint32_t grp = (info4 >> (8 * e)) & 0xffu;
            const bool ok = (qm >> grp) & 1u;
            s[kb][4 * g + e] = ok ? s[kb][4 * g + e] : -INFINITY;
          }
        }
    }
    // ---- online softmax, exact running maximum (q carries scale * log2 e: S is the log2-domain logit)
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int r = 0; r < 16; r++) mx = fmaxf(mx, s[kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, WAVE));
    const float m_new = fmaxf(m_run, mx);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
    const float m7 = m_use - 7.f;          // P is fed as 128 * 2^(S - m): e4m3 keeps 2^-16 of the row maximum; l carries the 128 too
    float rs = 0.f;
    v8i pf;                                 // P^T as the B operand: byte j of this lane half = key 32 (j >> 4) + 8 ((j >> 2) & 3) + 4 lh + (j & 3)
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const float p0 = __builtin_amdgcn_exp2f(s[kb][4 * g] - m7), p1 = __builtin_amdgcn_exp2f(s[kb][4 * g + 1] - m7);
        const float p2 = __builtin_amdgcn_exp2f(s[kb][4 * g + 2] - m7), p3 = __builtin_amdgcn_exp2f(s[kb][4 * g + 3] - m7);
        rs += (p0 + p1) + (p2 + p3);
        int w = __builtin_amdgcn_cvt_pk_fp8_f32(p0, p1, 0, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(p2, p3, w, true);
        pf[kb * 4 + g] = w;
      }
    rs += __shfl_xor(rs, 32, WAVE);
    l_run = l_run * alpha + rs;
    m_run = m_new;
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int r = 0; r < 16; r++) o[n][r] *= alpha;
    // ---- O^T += V^T P^T: A = V8T row d (this lane half's 16 positions of each 32-key block), unit scale for P
#pragma unroll
    for (int n = 0; n < 2; n++) {
      const int row = n * 32 + l31;
      const uint4 lo = *reinterpret_cast<const uint4*>(vs + t8_off(row, lh)), hi = *reinterpret_cast<const uint4*>(vs + t8_off(row, 2 + lh));
      v8i vf;
      vf[0] = lo.x; vf[1] = lo.y; vf[2] = lo.z; vf[3] = lo.w; vf[4] = hi.x; vf[5] = hi.y; vf[6] = hi.z; vf[7] = hi.w;
      const int vscale = sc_s[buf][1][row * 2 + lh];
      o[n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf, pf, o[n], 0, 0, 0, vscale, 0, 127);
    }

    if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wavefront's pieces of the next tile have landed
    else if (nit < it_end) swrite(buf ^ 1);
    __syncthreads();
    buf ^= 1;
    it = nit;
  }

  const bool uniform = !(l_run > 0.f) || m_run < -16384.f;          // (every score carried the -32768 of the mask product)
  const float inv = uniform ? 0.f : 1.f / l_run;
  if (qvalid) {
    if (lh == 0) a.lse[bh * a.nq + qrow] = uniform ? INFINITY : m_run + log2f(l_run) - 7.f;
    u16* op = a.o + (int64_t)b * a.o_bstride + (int64_t)qrow * a.o_ld + h * DH;
    const float* vm = a.vmean + (int64_t)b * a.heads * DH + h * DH;
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int d = n * 32 + 8 * g + 4 * lh;
        float v0, v1, v2, v3;
        if (uniform) { v0 = vm[d]; v1 = vm[d + 1]; v2 = vm[d + 2]; v3 = vm[d + 3]; }
        else { v0 = o[n][4 * g] * inv; v1 = o[n][4 * g + 1] * inv; v2 = o[n][4 * g + 2] * inv; v3 = o[n][4 * g + 3] * inv; }
        uint2 pk; pk.x = pack2bf(v0, v1); pk.y = pack2bf(v2, v3);
        *reinterpret_cast<uint2*>(op + d) = pk;
      }
  }
}

extern "C" int mca_attn_fwd_fp8(const mca_attn_fwd_args* a, const mca_attn_fp8_operands* f, mca_stream_t stream) {
  if (!a || !f || !a->o || !a->lse || !a->qmask || !a->keyinfo || !a->ktile_flags || !a->q_ptr || !a->q_kt || !a->q_order || !a->vmean)
    return MCA_E_BADARG;
  if (!f->q8 || !f->qs || !f->k8 || !f->ks || !f->v8t || !f->vs) return MCA_E_BADARG;
  if (a->batch <= 0 || a->heads <= 0 || a->nq <= 0 || a->nk <= 0) return MCA_E_BADARG;
  if (a->nq != a->nk) return MCA_E_UNSUPPORTED;          // self-attention only: the query rows are tokens of the same quantised block
  if (!(a->flags & MCA_ATTN_Q_PRESCALED)) return MCA_E_UNSUPPORTED;
  if (a->n_qtiles != (a->nq + AQ - 1) / AQ || a->n_ktiles != (a->nk + AK - 1) / AK || f->n_ktiles != a->n_ktiles) return MCA_E_BADARG;
  if (a->nk_pad < a->n_ktiles * AK || a->nk_pad % 4 || (uintptr_t)a->keyinfo % 4) return MCA_E_BADARG;
  if (a->o_ld % 4 || a->o_bstride % 4 || (uintptr_t)a->o % 8 || (uintptr_t)f->q8 % 16 || (uintptr_t)f->k8 % 16 || (uintptr_t)f->v8t % 16 ||
      (uintptr_t)f->ks % 4 || (uintptr_t)f->vs % 4 || (uintptr_t)a->khot % 16)
    return MCA_E_ALIGN;
  if (a->heads > 65535 || a->batch > 65535 || a->n_ktiles > MAX_KTILES) return MCA_E_UNSUPPORTED;
  const dim3 grid(a->n_qtiles, a->heads, a->batch);
  if (a->khot && !(mca_knobs[9] & 32)) hipLaunchKernelGGL(attn_fwd8_kernel<true>, grid, dim3(256), 0, as_stream(stream), *a, *f, mca_knobs[9]);          // knob 9 bit 32: register staging (A/B)
  else hipLaunchKernelGGL(attn_fwd8_kernel<false>, grid, dim3(256), 0, as_stream(stream), *a, *f, mca_knobs[9]);
  return launch_status();
}

// =========================================================================================================
// Backward with MX-fp8 score recomputes (BASELINE configs[4]).  The two-pass backward of attention_bwd2.hip recomputes
// S = Q K^T and dP = dO V^T in both passes: here those two products (4 of the 7 per tile) run on the block-scaled fp8 matrix
// instruction, from e4m3 copies of q, k, v and dO with one E8M0 scale per 32 elements ALONG d (the contraction of both
// products); the three products that build gradients (dQ = dS K, dK = dS^T Q, dV = P^T dO) keep bf16 operands and fp32
// accumulation.  S is then bit for bit the S of mca_attn_fwd_fp8 (same q8 / k8), so P = 2^(S - lse) is consistent with the
// log-sum-exp that forward wrote (the bf16 recompute was not).  delta stays rowsum(dO o O) of mca_attn_bwd_prep.
//   mca_attn_quant_bwd_mxfp8   q | k | v | dO (bf16) -> four [b][h][npad][64] e4m3 arrays + [b][h][npad][2] scale bytes
//   mca_attn_bwd_dq_fp8        attn_bwd_dq_kernel's structure (128-query workgroups, 64-key tiles, query on the lane)
//   mca_attn_bwd_dkv_fp8       attn_bwd_dkv_kernel's structure, 4 wavefronts x 32 keys, key on the lane
// Self-attention, MCA_ATTN_Q_PRESCALED and the mask product (khot / qblk) only.
// =========================================================================================================
__global__ __launch_bounds__(256) void attn_quant_rows4_kernel(const u16* __restrict__ q, int64_t q_bstride, int64_t q_ld,
                                                                const u16* __restrict__ k, const u16* __restrict__ v,
                                                                int64_t kv_bstride, int64_t kv_ld, const u16* __restrict__ d_o,
                                                                int64_t o_bstride, int64_t o_ld, mca_attn_fp8_bwd_operands f,
                                                                int heads, int n, int ntiles, int which_mask) {
  const int kt = blockIdx.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int r = tid >> 2, c = tid & 3;
  const int tok = kt * AK + r;
  const bool live = tok < n;
  const int64_t rowi = ((int64_t)b * heads + h) * ((int64_t)ntiles * AK) + tok;
  const int64_t t = live ? tok : 0;
#pragma unroll
  for (int which = 0; which < 4; which++) {
    if (!((which_mask >> which) & 1)) continue;
    const u16* src = which == 0 ? q + (int64_t)b * q_bstride + t * q_ld
                   : which == 1 ? k + (int64_t)b * kv_bstride + t * kv_ld
                   : which == 2 ? v + (int64_t)b * kv_bstride + t * kv_ld
                                : d_o + (int64_t)b * o_bstride + t * o_ld;
    src += h * DH + 16 * c;
    float x[16];
    const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(src), a1 = *reinterpret_cast<const bf16x8*>(src + 8);
#pragma unroll
    for (int i = 0; i < 8; i++) { x[i] = live ? bf2f((u16)a0[i]) : 0.f; x[8 + i] = live ? bf2f((u16)a1[i]) : 0.f; }
    uint4 o;
    const uint32_t sb = quant_block_half(x, o);
    uint8_t* d8 = which == 0 ? f.q8 : which == 1 ? f.k8 : which == 2 ? f.v8 : f.do8;
    uint8_t* ds = which == 0 ? f.qs : which == 1 ? f.ks : which == 2 ? f.vs : f.dos;
    *reinterpret_cast<uint4*>(d8 + rowi * DH + 16 * c) = o;
    if ((c & 1) == 0) ds[rowi * 2 + (c >> 1)] = (uint8_t)sb;
  }
}

static int check_fp8_bwd_operands(const mca_attn_fp8_bwd_operands* f) {
  if (!f || !f->q8 || !f->qs || !f->k8 || !f->ks || !f->v8 || !f->vs || !f->do8 || !f->dos) return MCA_E_BADARG;
  if ((uintptr_t)f->q8 % 16 || (uintptr_t)f->k8 % 16 || (uintptr_t)f->v8 % 16 || (uintptr_t)f->do8 % 16) return MCA_E_ALIGN;
  if ((uintptr_t)f->qs % 4 || (uintptr_t)f->ks % 4 || (uintptr_t)f->vs % 4 || (uintptr_t)f->dos % 4) return MCA_E_ALIGN;
  return MCA_OK;
}

extern "C" int mca_attn_quant_bwd_mxfp8(const uint16_t* q, int64_t q_bstride, int64_t q_ld, const uint16_t* k, const uint16_t* v,
                                        int64_t kv_bstride, int64_t kv_ld, const uint16_t* d_o, int64_t o_bstride, int64_t o_ld,
                                        const mca_attn_fp8_bwd_operands* f, int which, int batch, int heads, int n, mca_stream_t stream) {
  if (!q || !k || !v || !d_o || batch <= 0 || heads <= 0 || n <= 0 || which <= 0 || which > 15) return MCA_E_BADARG;
  const int rc = check_fp8_bwd_operands(f);
  if (rc != MCA_OK) return rc;
  if (q_ld % 8 || kv_ld % 8 || o_ld % 8 || q_bstride % 8 || kv_bstride % 8 || o_bstride % 8 || (uintptr_t)q % 16 || (uintptr_t)k % 16 ||
      (uintptr_t)v % 16 || (uintptr_t)d_o % 16)
    return MCA_E_ALIGN;
  const int ntiles = (n + AK - 1) / AK;
  if (f->n_ktiles != ntiles) return MCA_E_BADARG;
  if (heads > 65535 || batch > 65535) return MCA_E_UNSUPPORTED;
  hipLaunchKernelGGL(attn_quant_rows4_kernel, dim3(ntiles, heads, batch), dim3(256), 0, as_stream(stream), q, q_bstride, q_ld, k, v,
                     kv_bstride, kv_ld, d_o, o_bstride, o_ld, *f, heads, n, ntiles, which);
  return launch_status();
}

// ---------------------------------------------------------------------------------------------------------
// dQ pass
// ---------------------------------------------------------------------------------------------------------
// bf16 K image for the transposed reads of the dQ product (attention_bwd2.hip rt_off)
__device__ __forceinline__ int rt8_off(int r, int c) {
  const int s = ((r >> 1) & 7) ^ (((r >> 1) & 1) << 2);
  return r * 64 + ((c ^ s) << 3);
}

template <bool OUT_F32>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq8_kernel(mca_attn_bwd2_args a, mca_attn_fp8_bwd_operands f, int dbg) {
  __shared__ __attribute__((aligned(16))) u16 kbf_s[2 * AK * DH];          // bf16 K tile, double-buffered: 16 KiB
  __shared__ __attribute__((aligned(16))) uint8_t t8_s[2 * 2 * AK * DH];   // K8, V8 double-buffered: 16 KiB
  __shared__ __attribute__((aligned(16))) uint8_t sc_s[2][2][AK * 2];      // [buffer][K | V][row][block] scale bytes
  __shared__ __attribute__((aligned(16))) u16 hot_s[2][AK * 16];
  __shared__ uint8_t flags_s[MAX_KTILES];
  __shared__ uint32_t live_s[MAX_KTILES];
  __shared__ int n_live_s;
  uint8_t* K8s = t8_s;
  uint8_t* V8s = t8_s + 2 * AK * DH;

  const int lin0 = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
  const int lin = (dbg & 16) ? lin0 : xcd_remap(lin0, (int)(gridDim.x * gridDim.y * gridDim.z));
  const int qt = a.q_order[lin % (int)gridDim.x];
  const int h = (lin / (int)gridDim.x) % (int)gridDim.y, b = lin / (int)(gridDim.x * gridDim.y);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int q0 = qt * AQ + wave * 32;
  int qrow = q0 + l31;
  const bool qvalid = qrow < a.nq;
  if (qrow > a.nq - 1) qrow = a.nq - 1;
  const int ntiles = f.n_ktiles;
  const int64_t bh = (int64_t)b * a.heads + h;

  // B operands held for the whole kernel: this lane's query row of Q8 and dO8 (bytes 16 lh .. +15 of both 32-element blocks)
  v8i qf, dof;
  int qscale, doscale;
  {
    const int64_t rowi = bh * ((int64_t)ntiles * AK) + qrow;
    const uint4 lo = *reinterpret_cast<const uint4*>(f.q8 + rowi * DH + 16 * lh), hi = *reinterpret_cast<const uint4*>(f.q8 + rowi * DH + 32 + 16 * lh);
    qf[0] = lo.x; qf[1] = lo.y; qf[2] = lo.z; qf[3] = lo.w; qf[4] = hi.x; qf[5] = hi.y; qf[6] = hi.z; qf[7] = hi.w;
    qscale = f.qs[rowi * 2 + lh];
    const uint4 l2 = *reinterpret_cast<const uint4*>(f.do8 + rowi * DH + 16 * lh), h2 = *reinterpret_cast<const uint4*>(f.do8 + rowi * DH + 32 + 16 * lh);
    dof[0] = l2.x; dof[1] = l2.y; dof[2] = l2.z; dof[3] = l2.w; dof[4] = h2.x; dof[5] = h2.y; dof[6] = h2.z; dof[7] = h2.w;
    doscale = f.dos[rowi * 2 + lh];
  }
  const uint32_t qm = a.qmask[qrow];
  const uint32_t qm8 = ((qm >> (8 * lh)) & 0xffu) & (lh ? 0x7fu : 0xffu);
  f32x16 neglse, negdel;
  {
    const int64_t ri = bh * a.nq + qrow;
    const float nl = -a.lse[ri], nd = -a.delta[ri];
#pragma unroll
    for (int r = 0; r < 16; r++) { neglse[r] = nl; negdel[r] = nd; }
  }
  f32x16 dq[2];
#pragma unroll
  for (int n = 0; n < 2; n++)
#pragma unroll
    for (int r = 0; r < 16; r++) dq[n][r] = 0.f;

  const u16* kbase = a.k + (int64_t)b * a.kv_bstride + h * DH;
  const uint8_t* k8b = f.k8 + bh * ((int64_t)ntiles * AK) * DH;
  const uint8_t* ksb = f.ks + bh * ((int64_t)ntiles * AK) * 2;
  const uint8_t* v8b = f.v8 + bh * ((int64_t)ntiles * AK) * DH;
  const uint8_t* vsb = f.vs + bh * ((int64_t)ntiles * AK) * 2;
  const u16* khot_g = a.khot + (int64_t)b * a.nk_pad * 16;
  {
    const uint8_t* flags_g = a.ktile_flags + (int64_t)b * a.n_ktiles64;
    for (int i = tid; i < a.n_ktiles64; i += 256) flags_s[i] = flags_g[i];
  }
  __syncthreads();

  // staging: the bf16 K tile (512 pieces of 16 bytes), K8 and V8 (256 each), the 256 scale bytes and the one-hot tile (128 pieces)
  // go global -> LDS by LDS-DMA; the swizzles of the tile images sit on the SOURCE chunk (piece p lands at byte 16 p)
  int srow[2];
  unsigned koff[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int id = tid + 256 * i, r = id >> 3, sl = id & 7;
    srow[i] = r;
    koff[i] = (unsigned)(r * (int)a.kv_ld + ((sl ^ (((r >> 1) & 7) ^ (((r >> 1) & 1) << 2))) << 3));          // rt8_off
  }
  const int s8row = tid >> 2, s8sw = (tid & 3) ^ ((s8row >> 2) & 3);                                           // t8_off
  const int last_kt = a.n_ktiles64 - 1;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto stage = [&](int kt, int buf) {
    const u16* kb = kbase + (int64_t)kt * AK * a.kv_ld;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      unsigned ko = koff[i];
      if (kt == last_kt) { const int over = kt * AK + srow[i] - (a.nk - 1); if (over > 0) ko -= (unsigned)(over * (int)a.kv_ld); }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kb + ko),
                                       (__attribute__((address_space(3))) void*)(kbf_s + buf * AK * DH + (i * 4 + wave_u) * 512), 16, 0, 0);
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(k8b + ((int64_t)kt * AK + s8row) * DH + 16 * s8sw),
                                     (__attribute__((address_space(3))) void*)(K8s + buf * AK * DH + wave_u * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(v8b + ((int64_t)kt * AK + s8row) * DH + 16 * s8sw),
                                     (__attribute__((address_space(3))) void*)(V8s + buf * AK * DH + wave_u * 1024), 16, 0, 0);
    if (wave_u < 2)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(khot_g + (int64_t)kt * (AK * 16) + tid * 8),
                                       (__attribute__((address_space(3))) void*)(&hot_s[buf][wave_u * 512]), 16, 0, 0);
    if (wave_u == 2) {          // lanes 0..31: the 128 K scale bytes, 32..63: the V scale bytes
      const uint8_t* sp = (lane < 32 ? ksb : vsb) + (int64_t)kt * AK * 2 + (lane & 31) * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp, (__attribute__((address_space(3))) void*)(&sc_s[buf][0][0]), 4, 0, 0);
    }
  };

  if (wave == 0) {
    const int lb = a.q_ptr[qt], le = a.q_ptr[qt + 1];
    int n = 0;
    for (int i0 = lb; i0 < le; i0 += 64) {
      const int i = i0 + lane;
      const uint32_t e = i < le ? a.q_kt[i] : 0u;
      const uint8_t fl = i < le ? flags_s[e & 0x7fffffffu] : (uint8_t)0;
      const bool keep = fl != 0;
      const unsigned long long m = __ballot(keep);
      if (keep) live_s[n + __popcll(m & ((1ull << lane) - 1ull))] = (e & 0x7fffffffu) | ((e >> 31) && fl == 2 ? 0x80000000u : 0u);
      n += __popcll(m);
    }
    if (lane == 0) n_live_s = n;
  }
  __syncthreads();
  const int it_end = n_live_s;
  int buf = 0;
  if (it_end > 0) { stage((int)(live_s[0] & 0x7fffffffu), 0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  __syncthreads();

  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  for (int it = 0; it < it_end; it++) {
    const uint32_t ent = live_s[it];
    const bool need_mask = (ent >> 31) == 0;
    if (it + 1 < it_end) stage((int)(live_s[it + 1] & 0x7fffffffu), buf ^ 1);
    const u16* ks = kbf_s + buf * AK * DH;
    const uint8_t* k8 = K8s + buf * AK * DH;
    const uint8_t* v8 = V8s + buf * AK * DH;
    // S^T - lse and dP^T - delta: ONE block-scaled MFMA each per 32-key block (row constants as the accumulator start values)
    f32x16 s[2], dp[2];
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
      const int row = kb * 32 + l31;
      const uint4 klo = *reinterpret_cast<const uint4*>(k8 + t8_off(row, lh)), khi = *reinterpret_cast<const uint4*>(k8 + t8_off(row, 2 + lh));
      const uint4 vlo = *reinterpret_cast<const uint4*>(v8 + t8_off(row, lh)), vhi = *reinterpret_cast<const uint4*>(v8 + t8_off(row, 2 + lh));
      v8i kf, vf;
      kf[0] = klo.x; kf[1] = klo.y; kf[2] = klo.z; kf[3] = klo.w; kf[4] = khi.x; kf[5] = khi.y; kf[6] = khi.z; kf[7] = khi.w;
      vf[0] = vlo.x; vf[1] = vlo.y; vf[2] = vlo.z; vf[3] = vlo.w; vf[4] = vhi.x; vf[5] = vhi.y; vf[6] = vhi.z; vf[7] = vhi.w;
      const int kscale = sc_s[buf][0][row * 2 + lh], vscale = sc_s[buf][1][row * 2 + lh];
      s[kb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf, qf, neglse, 0, 0, 0, kscale, 0, qscale);
      dp[kb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf, dof, negdel, 0, 0, 0, vscale, 0, doscale);
    }
    if (need_mask) {          // blocked (query, key) pairs: -32768 on top of the score, exp2 is exactly 0
      u32x4v qb;
#pragma unroll
      for (int w = 0; w < 4; w++)
        qb[w] = (((qm8 >> (2 * w)) & 1u) ? 0u : 0xC700u) | (((qm8 >> (2 * w + 1)) & 1u) ? 0u : 0xC7000000u);
#pragma unroll
      for (int kb = 0; kb < 2; kb++) {
        const bf16x8 hf = *reinterpret_cast<const bf16x8*>(&hot_s[buf][(kb * 32 + l31) * 16 + 8 * lh]);
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf, *reinterpret_cast<const bf16x8*>(&qb), s[kb], 0, 0, 0);
      }
    }
    // dS^T = P^T o (dP^T - delta), packed as the B operand of the dQ^T product (bf16)
    u32x4v dsw[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int sp = 0; sp < 2; sp++)
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          const float d0 = __builtin_amdgcn_exp2f(s[kb][8 * sp + j]) * dp[kb][8 * sp + j];
          const float d1 = __builtin_amdgcn_exp2f(s[kb][8 * sp + j + 1]) * dp[kb][8 * sp + j + 1];
          dsw[kb][sp][j >> 1] = pack2bf_pk(d0, d1);
        }
    // dQ^T[d][q] += K^T[d][key] dS^T[key][q]: bf16, A = K^T by transposed reads of the bf16 K image
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int sp = 0; sp < 2; sp++)
#pragma unroll
        for (int n = 0; n < 2; n++) {
          bf16x8 ktf;
#pragma unroll
          for (int t = 0; t < 2; t++) {
            const int key = kb * 32 + 16 * sp + 8 * t + 4 * lh + tq;
            const int d = n * 32 + 16 * tg + 4 * tp;
            const bf16x4 k4 = lds_read_tr16(ks + rt8_off(key, d >> 3) + (d & 7));
#pragma unroll
            for (int e = 0; e < 4; e++) ktf[4 * t + e] = k4[e];
          }
          dq[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, *reinterpret_cast<const bf16x8*>(&dsw[kb][sp]), dq[n], 0, 0, 0);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wavefront's pieces of the next tile have landed
    __syncthreads();
    buf ^= 1;
  }

  if (qvalid) {
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int d = n * 32 + 8 * g + 4 * lh;
        const float v0 = dq[n][4 * g] * a.scale, v1 = dq[n][4 * g + 1] * a.scale, v2 = dq[n][4 * g + 2] * a.scale, v3 = dq[n][4 * g + 3] * a.scale;
        if (OUT_F32) {
          float* p = reinterpret_cast<float*>(a.dq) + (int64_t)b * a.dq_bstride + (int64_t)qrow * a.dq_ld + h * DH + d;
          *reinterpret_cast<f32x4*>(p) = f32x4{v0, v1, v2, v3};
        } else {
          u16* p = reinterpret_cast<u16*>(a.dq) + (int64_t)b * a.dq_bstride + (int64_t)qrow * a.dq_ld + h * DH + d;
          uint2 pk; pk.x = pack2bf(v0, v1); pk.y = pack2bf(v2, v3);
          *reinterpret_cast<uint2*>(p) = pk;
        }
      }
  }
}

static int check_bwd8(const mca_attn_bwd2_args* a, const mca_attn_fp8_bwd_operands* f) {
  if (!a || !a->k || !a->d_o || !a->lse || !a->delta || !a->qmask || !a->keyinfo || !a->ktile_flags) return MCA_E_BADARG;
  const int rc = check_fp8_bwd_operands(f);
  if (rc != MCA_OK) return rc;
  if (a->batch <= 0 || a->heads <= 0 || a->nq <= 0 || a->nk <= 0) return MCA_E_BADARG;
  if (a->nq != a->nk) return MCA_E_UNSUPPORTED;                                   // self-attention: one quantised token block
  if (!(a->flags & MCA_ATTN_Q_PRESCALED) || !a->khot || !a->qblk) return MCA_E_UNSUPPORTED;
  if (f->n_ktiles != (a->nk + AK - 1) / AK) return MCA_E_BADARG;
  if (a->q_ld % 8 || a->kv_ld % 8 || a->o_ld % 8 || a->q_bstride % 8 || a->kv_bstride % 8 || a->o_bstride % 8) return MCA_E_ALIGN;
  if ((uintptr_t)a->q % 16 || (uintptr_t)a->k % 16 || (uintptr_t)a->d_o % 16 || (uintptr_t)a->khot % 16 || (uintptr_t)a->qblk % 16) return MCA_E_ALIGN;
  if (a->heads > 65535 || a->batch > 65535) return MCA_E_UNSUPPORTED;
  return MCA_OK;
}

extern "C" int mca_attn_bwd_dq_fp8(const mca_attn_bwd2_args* a, const mca_attn_fp8_bwd_operands* f, mca_stream_t stream) {
  const int rc = check_bwd8(a, f);
  if (rc != MCA_OK) return rc;
  if (!a->dq || !a->q_ptr || !a->q_kt || !a->q_order) return MCA_E_BADARG;
  if (a->n_qtiles128 != (a->nq + AQ - 1) / AQ || a->n_ktiles64 != (a->nk + AK - 1) / AK) return MCA_E_BADARG;
  if (a->nk_pad < a->n_ktiles64 * AK) return MCA_E_BADARG;
  if (a->dq_ld % 4 || a->dq_bstride % 4 || (uintptr_t)a->dq % (a->dq_f32 ? 16 : 8)) return MCA_E_ALIGN;
  if (a->n_ktiles64 > MAX_KTILES) return MCA_E_UNSUPPORTED;
  const dim3 grid(a->n_qtiles128, a->heads, a->batch);
  if (a->dq_f32) hipLaunchKernelGGL(attn_bwd_dq8_kernel<true>, grid, dim3(256), 0, as_stream(stream), *a, *f, mca_knobs[9]);
  else hipLaunchKernelGGL(attn_bwd_dq8_kernel<false>, grid, dim3(256), 0, as_stream(stream), *a, *f, mca_knobs[9]);
  return launch_status();
}

// ---------------------------------------------------------------------------------------------------------
// dK / dV pass (4 wavefronts x 32 keys per workgroup, 64-query steps)
// ---------------------------------------------------------------------------------------------------------
#define BQ8 64
#define MAX_QTILES8 2048
#define DKV8_LDS_BYTES (2 * BQ8 * DH * 2 * 2 + 2 * BQ8 * DH * 2 + 2 * 2 * BQ8 * 2 + 2 * 192 * 4 + MAX_QTILES8 * 4 + DH * 4 + 2 * BQ8 * 16 * 2)

__global__ __launch_bounds__(256) void attn_bwd_dkv8_kernel(mca_attn_bwd2_args a, mca_attn_fp8_bwd_operands f, int dbg) {
  constexpr int BKEYS = 128, NT = 256;
  extern __shared__ __attribute__((aligned(16))) u16 lds[];
  u16* Qs = lds;                               // bf16 Q tile, 2 x 64 x 64 (transposed reads of the dK product)
  u16* Os = Qs + 2 * BQ8 * DH;                 // bf16 dO tile
  uint8_t* Q8s = reinterpret_cast<uint8_t*>(Os + 2 * BQ8 * DH);      // 2 x 64 x 64 bytes
  uint8_t* O8s = Q8s + 2 * BQ8 * DH;
  uint8_t* sc8 = O8s + 2 * BQ8 * DH;                                  // [2][Q | dO][64 rows][2]
  float* rowc = reinterpret_cast<float*>(sc8 + 2 * 2 * BQ8 * 2);      // [2][3][64]: -lse, -delta, (unused)
  uint32_t* qlist = reinterpret_cast<uint32_t*>(rowc + 2 * 192);
  float* dvm_s = reinterpret_cast<float*>(qlist + MAX_QTILES8);
  u16* qblk_s = reinterpret_cast<u16*>(dvm_s + DH);                   // [2][64][16]

  const int lin0 = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
  const int lin = (dbg & 16) ? lin0 : xcd_remap(lin0, (int)(gridDim.x * gridDim.y * gridDim.z));
  const int4 w = reinterpret_cast<const int4*>(a.k_wg)[lin % (int)gridDim.x];
  const int kbi = w.x, it_begin = w.y, first_qt = w.w;
  const int h = (lin / (int)gridDim.x) % (int)gridDim.y, b = lin / (int)(gridDim.x * gridDim.y);
  int n_it = w.z;
  {
    const uint8_t* fl = a.ktile_flags + (int64_t)b * a.n_ktiles64;
    int live = 0;
    for (int t = 0; t < BKEYS / AK; t++) { const int kt = kbi * (BKEYS / AK) + t; if (kt < a.n_ktiles64) live |= fl[kt]; }
    if (!live) n_it = 0;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int key0 = kbi * BKEYS;
  const int mykey = key0 + wave * 32 + l31;
  const int ntiles = f.n_ktiles;
  const int64_t bh = (int64_t)b * a.heads + h;
  const int64_t row0 = bh * ((int64_t)ntiles * AK);

  // B operands held for the whole kernel: this lane's key row of K8 and V8
  v8i kf, vf;
  int kscale, vscale;
  {
    // (a key past the quantised block, 128-key blocks against 64-row padding, is a padded key: masked by its khot row)
    const int64_t rowi = row0 + (mykey < ntiles * AK ? mykey : ntiles * AK - 1);
    const uint4 lo = *reinterpret_cast<const uint4*>(f.k8 + rowi * DH + 16 * lh), hi = *reinterpret_cast<const uint4*>(f.k8 + rowi * DH + 32 + 16 * lh);
    kf[0] = lo.x; kf[1] = lo.y; kf[2] = lo.z; kf[3] = lo.w; kf[4] = hi.x; kf[5] = hi.y; kf[6] = hi.z; kf[7] = hi.w;
    kscale = f.ks[rowi * 2 + lh];
    const uint4 l2 = *reinterpret_cast<const uint4*>(f.v8 + rowi * DH + 16 * lh), h2 = *reinterpret_cast<const uint4*>(f.v8 + rowi * DH + 32 + 16 * lh);
    vf[0] = l2.x; vf[1] = l2.y; vf[2] = l2.z; vf[3] = l2.w; vf[4] = h2.x; vf[5] = h2.y; vf[6] = h2.z; vf[7] = h2.w;
    vscale = f.vs[rowi * 2 + lh];
  }
  const uint32_t kinfo = a.keyinfo[(int64_t)b * a.nk_pad + mykey];
  const bf16x8 khf = *reinterpret_cast<const bf16x8*>(a.khot + ((int64_t)b * a.nk_pad + mykey) * 16 + 8 * lh);
  const bool key_ok = kinfo != 31u;
  const bool wave_keys_ok = __all(key_ok);
  const bool wave_dead = !__any(key_ok);
  const float dk_scale = 0.6931471805599453f;          // q carries scale * log2 e: dK^T carries that factor

  f32x16 dk[2], dv[2];
#pragma unroll
  for (int n = 0; n < 2; n++)
#pragma unroll
    for (int r = 0; r < 16; r++) { dk[n][r] = 0.f; dv[n][r] = 0.f; }

  const u16* qbase = a.q + (int64_t)b * a.q_bstride + h * DH;
  const u16* obase = a.d_o + (int64_t)b * a.o_bstride + h * DH;
  const float* lse_g = a.lse + bh * a.nq;
  const float* delta_g = a.delta + bh * a.nq;

  // staging: the bf16 Q and dO tiles (512 pieces of 16 bytes each), Q8 and dO8 (256 each), the 256 scale bytes and the qblk tile
  // (128 pieces) go global -> LDS by LDS-DMA, the tile swizzles on the SOURCE chunk (piece p lands at byte 16 p); the two
  // per-row constants pass through a register (negated, +inf / 0 for rows past nq)
  const int srow = tid >> 3;                                                                           // (+32 for the second piece: same swizzle)
  const int ssw = ((tid & 7) ^ (((srow >> 1) & 7) ^ (((srow >> 1) & 1) << 2))) << 3;                   // rt8_off
  const int s8row = tid >> 2, s8sw = 16 * ((tid & 3) ^ ((s8row >> 2) & 3));                            // t8_off
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  float stage_c = 0.f;
  bool stage_oob = false;
  const int cwhich = tid >> 6 < 2 ? tid >> 6 : 2, crow = tid & 63;
  const float* cbase = cwhich == 0 ? lse_g : delta_g;
  auto stage = [&](int qt, int buf) {
#pragma unroll
    for (int u = 0; u < 2; u++) {
      int q = qt * BQ8 + srow + u * 32; if (q > a.nq - 1) q = a.nq - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qbase + (int64_t)q * a.q_ld + ssw),
                                       (__attribute__((address_space(3))) void*)(Qs + buf * BQ8 * DH + (u * 4 + wave_u) * 512), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(obase + (int64_t)q * a.o_ld + ssw),
                                       (__attribute__((address_space(3))) void*)(Os + buf * BQ8 * DH + (u * 4 + wave_u) * 512), 16, 0, 0);
    }
    const int64_t r8 = row0 + (int64_t)qt * BQ8 + s8row;          // rows past nq are zero rows of the quantised arrays
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(f.q8 + r8 * DH + s8sw),
                                     (__attribute__((address_space(3))) void*)(Q8s + buf * BQ8 * DH + wave_u * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(f.do8 + r8 * DH + s8sw),
                                     (__attribute__((address_space(3))) void*)(O8s + buf * BQ8 * DH + wave_u * 1024), 16, 0, 0);
    if (wave_u == 3) {          // scale bytes of the 64 rows: 128 bytes of Q (lanes 0..31), 128 of dO (32..63)
      const uint8_t* sp = (lane < 32 ? f.qs : f.dos) + (row0 + (int64_t)qt * BQ8) * 2 + (lane & 31) * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp,
                                       (__attribute__((address_space(3))) void*)(sc8 + buf * (2 * BQ8 * 2)), 4, 0, 0);
    }
    if (wave_u < 2) {
      int qb = qt * BQ8 + (tid >> 1); if (qb > a.nq - 1) qb = a.nq - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.qblk + (int64_t)qb * 16 + (tid & 1) * 8),
                                       (__attribute__((address_space(3))) void*)(qblk_s + buf * (BQ8 * 16) + wave_u * 512), 16, 0, 0);
      int qq = qt * BQ8 + crow;
      stage_oob = qq >= a.nq;
      if (qq > a.nq - 1) qq = a.nq - 1;
      stage_c = cbase[qq];
    }
  };
  auto swrite = [&](int buf) {          // rows past nq contribute nothing: lse = +inf (P = 0), delta = 0
    if (tid < 128) rowc[buf * 192 + tid] = -(stage_oob ? (cwhich == 0 ? INFINITY : 0.f) : stage_c);
  };

  if (n_it > 0) stage(first_qt, 0);
  for (int i = tid; i < n_it; i += NT) qlist[i] = a.k_qt[it_begin + i];
  if (tid < DH) dvm_s[tid] = a.dvmean[(int64_t)b * a.heads * DH + h * DH + tid];
  int buf = 0;
  if (n_it > 0) swrite(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  for (int it = 0; it < n_it; it++) {
    const uint32_t ent = __builtin_amdgcn_readfirstlane(qlist[it]);
    const bool full = (ent >> 31) != 0;
    if (it + 1 < n_it) stage((int)(__builtin_amdgcn_readfirstlane(qlist[it + 1]) & 0x7fffffffu), buf ^ 1);
    if (!wave_dead)
#pragma unroll
    for (int sub = 0; sub < 2; sub++) {
      const u16* qs = Qs + buf * BQ8 * DH + sub * 32 * DH;
      const u16* os = Os + buf * BQ8 * DH + sub * 32 * DH;
      const uint8_t* q8 = Q8s + buf * BQ8 * DH;
      const uint8_t* o8 = O8s + buf * BQ8 * DH;
      const float* rc = rowc + buf * 192 + sub * 32;
      const int row = sub * 32 + l31;
      // A operands: this lane's QUERY row of the sub-tile (Q8 for S, dO8 for dP); accumulator start values = the row constants
      // of the rows this lane's registers hold (rows 8 g + 4 lh + e)
      const uint4 qlo = *reinterpret_cast<const uint4*>(q8 + t8_off(row, lh)), qhi = *reinterpret_cast<const uint4*>(q8 + t8_off(row, 2 + lh));
      const uint4 olo = *reinterpret_cast<const uint4*>(o8 + t8_off(row, lh)), ohi = *reinterpret_cast<const uint4*>(o8 + t8_off(row, 2 + lh));
      v8i qa, oa;
      qa[0] = qlo.x; qa[1] = qlo.y; qa[2] = qlo.z; qa[3] = qlo.w; qa[4] = qhi.x; qa[5] = qhi.y; qa[6] = qhi.z; qa[7] = qhi.w;
      oa[0] = olo.x; oa[1] = olo.y; oa[2] = olo.z; oa[3] = olo.w; oa[4] = ohi.x; oa[5] = ohi.y; oa[6] = ohi.z; oa[7] = ohi.w;
      const int qsc = sc8[buf * (2 * BQ8 * 2) + row * 2 + lh], osc = sc8[buf * (2 * BQ8 * 2) + BQ8 * 2 + row * 2 + lh];
      f32x16 s, dp;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const f32x4 cl = *reinterpret_cast<const f32x4*>(rc + 8 * g + 4 * lh), cd = *reinterpret_cast<const f32x4*>(rc + 64 + 8 * g + 4 * lh);
#pragma unroll
        for (int e = 0; e < 4; e++) { s[4 * g + e] = cl[e]; dp[4 * g + e] = cd[e]; }
      }
      s = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qa, kf, s, 0, 0, 0, qsc, 0, kscale);
      dp = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(oa, vf, dp, 0, 0, 0, osc, 0, vscale);
      const bool clean = full && wave_keys_ok;
      if (!clean) {          // blocked (query, key) pairs: -32768 on top of the score, exp2 is exactly 0
        const bf16x8 qbf = *reinterpret_cast<const bf16x8*>(qblk_s + buf * (BQ8 * 16) + (sub * 32 + l31) * 16 + 8 * lh);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qbf, khf, s, 0, 0, 0);
      }
      u32x4v pbw[2], sbw[2];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = __builtin_amdgcn_exp2f(s[r]), p1 = __builtin_amdgcn_exp2f(s[r + 1]);
        pbw[r >> 3][(r & 7) >> 1] = pack2bf_pk(p0, p1);
        sbw[r >> 3][(r & 7) >> 1] = pack2bf_pk(p0 * dp[r], p1 * dp[r + 1]);
      }
      // dV^T += dO^T P ; dK^T += Q^T dS  (bf16 operands, transposed reads of the bf16 tiles)
#pragma unroll
      for (int sp = 0; sp < 2; sp++)
#pragma unroll
        for (int n = 0; n < 2; n++) {
          bf16x8 ot, qtf;
#pragma unroll
          for (int t = 0; t < 2; t++) {
            const int qr = 16 * sp + 8 * t + 4 * lh + tq;
            const int d = n * 32 + 16 * tg + 4 * tp;
            const bf16x4 o4 = lds_read_tr16(os + rt8_off(qr, d >> 3) + (d & 7));
            const bf16x4 q4 = lds_read_tr16(qs + rt8_off(qr, d >> 3) + (d & 7));
#pragma unroll
            for (int e = 0; e < 4; e++) { ot[4 * t + e] = o4[e]; qtf[4 * t + e] = q4[e]; }
          }
          dv[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ot, *reinterpret_cast<const bf16x8*>(&pbw[sp]), dv[n], 0, 0, 0);
          dk[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, *reinterpret_cast<const bf16x8*>(&sbw[sp]), dk[n], 0, 0, 0);
        }
    }
    if (it + 1 < n_it) swrite(buf ^ 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wavefront's pieces of the next tile have landed
    __syncthreads();
    buf ^= 1;
  }

  if (mykey < a.nk) {
    u16* dkp = a.dk + (int64_t)b * a.dkv_bstride + (int64_t)mykey * a.dkv_ld + h * DH;
    u16* dvp = a.dv + (int64_t)b * a.dkv_bstride + (int64_t)mykey * a.dkv_ld + h * DH;
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int d = n * 32 + 8 * g + 4 * lh;
        uint2 pk;
        pk.x = pack2bf(dk[n][4 * g] * dk_scale, dk[n][4 * g + 1] * dk_scale);
        pk.y = pack2bf(dk[n][4 * g + 2] * dk_scale, dk[n][4 * g + 3] * dk_scale);
        *reinterpret_cast<uint2*>(dkp + d) = pk;
        const f32x4 dvm = *reinterpret_cast<const f32x4*>(dvm_s + d);
        pk.x = pack2bf(dv[n][4 * g] + dvm[0], dv[n][4 * g + 1] + dvm[1]);
        pk.y = pack2bf(dv[n][4 * g + 2] + dvm[2], dv[n][4 * g + 3] + dvm[3]);
        *reinterpret_cast<uint2*>(dvp + d) = pk;
      }
  }
}

extern "C" int mca_attn_bwd_dkv_fp8(const mca_attn_bwd2_args* a, const mca_attn_fp8_bwd_operands* f, mca_stream_t stream) {
  const int rc = check_bwd8(a, f);
  if (rc != MCA_OK) return rc;
  if (!a->dk || !a->dv || !a->dvmean || !a->k_wg || !a->k_qt) return MCA_E_BADARG;
  if (a->kblock_keys != 128) return MCA_E_UNSUPPORTED;
  if (a->n_qtiles64 != (a->nq + BQ8 - 1) / BQ8 || a->n_kblocks256 != (a->nk + 127) / 128) return MCA_E_BADARG;
  if (a->nk_pad < a->n_kblocks256 * 128) return MCA_E_BADARG;
  if (a->dkv_ld % 4 || a->dkv_bstride % 4 || (uintptr_t)a->dk % 8 || (uintptr_t)a->dv % 8 || (uintptr_t)a->k_wg % 16) return MCA_E_ALIGN;
  if (a->n_qtiles64 > MAX_QTILES8) return MCA_E_UNSUPPORTED;
  int dev = 0;
  static bool attr_set[64] = {false};          // the attribute is per device
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MCA_E_LAUNCH;
  if (!attr_set[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DKV8_LDS_BYTES) != hipSuccess)
      return MCA_E_LAUNCH;
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL(attn_bwd_dkv8_kernel, dim3(a->n_kblocks256, a->heads, a->batch), dim3(256), DKV8_LDS_BYTES, as_stream(stream), *a, *f, mca_knobs[9]);
  return launch_status();
}
