// Block-masked fused attention, backward in ONE pass without atomics (autograd of model.py:73-105 as called by MCALayer :119).
//
// The two-pass backward (attention_bwd2.hip) pays for its atomic-free dQ with S and dP computed twice: 7 matrix products for
// the 5 the gradient needs.  Here ONE workgroup owns a whole (sample, head): it walks that pair's key blocks in order (256
// keys: 4 wavefronts x 64 keys, dK^T / dV^T and the K / V operand fragments in registers, ONE wavefront per SIMD on the whole
// 512-entry register file) and, inside a key block, the query tiles the structure allows (<= 64 rows per step).  Per step:
//   S^T-lse, dP^T-delta (key on the lane: the accumulators ARE the B operands of the next two products), P, dS,
//   dV^T += dO^T P, dK^T += Q^T dS, and dS crosses LDS once ([key][query] image) for dQ^T += K^T dS^T: five products.
// dQ of a query tile is summed over the key blocks that visit it.  Those visits all happen in THIS workgroup, one after the
// other, so the sum is a plain private read-modify-write of an fp32 tile (the first visit starts from zero, the last one
// writes bf16 to dq): no atomics (1.3 TB/s chip-wide on this part: the 0.9 GB of adds per layer put a 690 us floor under the
// round-1 kernel), no hand-off between workgroups, no fence, and the order of the sum is fixed: bitwise reproducible.  A lane
// reads back exactly the 16-byte pieces it wrote itself (dq_acc is laid out [tile][wavefront][register group][lane][4]), so
// the read-after-write is program order of one work-item.
// Tiles follow the structure (structure.build_onepass_schedule): query tiles and key blocks are cut along the modality
// boundaries, described by {first row, rows} tables; a step whose every pair is allowed and whose keys are all valid skips
// the mask product.
#include "common.h"

#define DH 64
#define TQ 64                     // query rows per step (at most)
#define TKB 256                   // keys per block (at most): 4 wavefronts x 64
#define B1_NST 2                  // Q / dO / row-constant / mask-operand stages
#define B1_STAGE_U16 (TQ * DH * 2 + 256 + TQ * 16)          // u16 per stage: Q | dO | rowc (128 floats) | qblk = 18,944 bytes
#define B1_MAX_QT 256
#define B1_MAX_KB 64
#define B1_MAX_LIST 256
#define B1_MAX_META 768              // step records of all key blocks of a (sample, head) (pipelined kernel)
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));

// [rows][64] bf16 image (128-byte rows) serving ds_read_b128 row reads AND ds_read_b64_tr_b16 transposed reads (as attention_bwd2.hip)
__device__ __forceinline__ int b1_off(int r, int c) {
  const int s = ((r >> 1) & 7) ^ (((r >> 1) & 1) << 2);
  return r * 64 + ((c ^ s) << 3);
}
__device__ __forceinline__ int b1_swz(int r) { return ((r >> 1) & 7) ^ (((r >> 1) & 1) << 2); }

#define B1_DMA(SRC, DST, BYTES) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC), (__attribute__((address_space(3))) void*)(DST), BYTES, 0, 0)

__global__ __launch_bounds__(256, 1) void attn_bwd1_kernel(mca_attn_bwd1_args a, int dbg) {
  extern __shared__ __attribute__((aligned(16))) u16 lds[];
  u16* stage_s = lds;                                             // B1_NST stages
  u16* kimg = stage_s + B1_NST * B1_STAGE_U16;                    // [256 keys][64 d]: A operand of the dQ product (transposed reads)
  u16* dsimg = kimg + TKB * DH;                                   // 2 x [256 keys][64 queries]: dS^T of a step
  uint32_t* list_s = reinterpret_cast<uint32_t*>(dsimg + 2 * TKB * TQ);   // the key block's query tiles
  uint8_t* first_s = reinterpret_cast<uint8_t*>(list_s + B1_MAX_LIST);    // per query tile: first / last LIVE key block that visits it
  uint8_t* last_s = first_s + B1_MAX_QT;
  uint8_t* live_s = last_s + B1_MAX_QT;                           // per key block: has a valid key in this sample
  float* dvm_s = reinterpret_cast<float*>(live_s + B1_MAX_KB);

  const int lin = (dbg & 16) ? (int)blockIdx.x : xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int h = lin % a.heads, b = lin / a.heads;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  const int64_t bh = (int64_t)b * a.heads + h;

  const u16* qbase = a.q + (int64_t)b * a.q_bstride + h * (a.q_hstride ? a.q_hstride : (int64_t)DH);
  const u16* obase = a.d_o + (int64_t)b * a.o_bstride + h * (a.o_hstride ? a.o_hstride : (int64_t)DH);
  const u16* kbase = a.k + (int64_t)b * a.kv_bstride + h * DH;
  const u16* vbase = a.v + (int64_t)b * a.kv_bstride + h * DH;
  const float* rowc_g = a.rowc + bh * (int64_t)(a.n_qtiles + 1) * 128;          // (+ the null tile)
  float* acc_g = a.dq_acc + bh * (int64_t)(a.n_qtiles + 1) * (TQ * DH);          // (+ the null tile's slot: only the pipelined kernel writes it)
  const uint8_t* kinfo_g = a.keyinfo + (int64_t)b * a.nk_pad;
  const u16* khot_g = a.khot + (int64_t)b * a.nk_pad * 16;

  // ---- which key blocks hold a valid key in this sample (64-key tile flags of mca_build_keyinfo: conservative at block edges),
  // and per query tile the first / last of them that visits it
  if (tid < a.n_kblocks) {
    const int4 d = reinterpret_cast<const int4*>(a.kb_desc)[tid];
    const uint8_t* fl = a.ktile_flags + (int64_t)b * a.n_ktiles64;
    int live = 0;
    for (int t = d.x >> 6; t <= (d.x + d.y - 1) >> 6; t++) live |= fl[t];
    live_s[tid] = live ? 1 : 0;
  }
  if (tid < DH) dvm_s[tid] = a.dvmean[(int64_t)b * a.heads * DH + h * DH + tid];
  __syncthreads();
  for (int qt = tid; qt < a.n_qtiles; qt += 256) {
    int first = 255, last = 255;
    for (int kb = 0; kb < a.n_kblocks; kb++)
      if (live_s[kb] && a.visit[kb * a.n_qtiles + qt]) { if (first == 255) first = kb; last = kb; }
    first_s[qt] = (uint8_t)first; last_s[qt] = (uint8_t)last;
  }
  __syncthreads();
  // query tiles nobody visits (every key they may see is padded in this sample): dq = 0
  for (int qt = 0; qt < a.n_qtiles; qt++) {
    if (first_s[qt] != 255) continue;
    const int2 qd = reinterpret_cast<const int2*>(a.qt_desc)[qt];
    for (int i = tid; i < qd.y * 8; i += 256)
      *reinterpret_cast<uint4*>(a.dq + (int64_t)b * a.dq_bstride + (int64_t)(qd.x + (i >> 3)) * a.dq_ld + h * DH + (i & 7) * 8) = make_uint4(0, 0, 0, 0);
  }

  // loop-invariant LDS offsets (u16 units)
  int rowfrag[4];          // row fragment (row l31, k-step st) of a 32-row block of a [rows][64] image
#pragma unroll
  for (int st = 0; st < 4; st++) rowfrag[st] = b1_off(l31, 2 * st + lh);
  const float dk_scale = 0.6931471805599453f;          // q carries scale * log2 e: dK^T carries that factor

  for (int kbi = 0; kbi < a.n_kblocks; kbi++) {
    if (!live_s[kbi]) continue;          // (uniform over the workgroup)
    const int4 kd = reinterpret_cast<const int4*>(a.kb_desc)[kbi];
    const int key_start = __builtin_amdgcn_readfirstlane(kd.x), n_keys = __builtin_amdgcn_readfirstlane(kd.y);
    const int e_begin = __builtin_amdgcn_readfirstlane(kd.z), n_ent = __builtin_amdgcn_readfirstlane(kd.w);
    for (int i = tid; i < n_ent; i += 256) list_s[i] = a.kb_qt[e_begin + i];

    // ---- this lane's two keys (slots wave * 64 + kb * 32 + l31): operand fragments for the whole block
    bf16x8 kf[2][4], vf[2][4], khf[2];
    bool ok[2];
    int keyrow[2];
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
      const int slot = wave * 64 + kb * 32 + l31;
      const bool valid = slot < n_keys;
      const int key = key_start + (valid ? slot : n_keys - 1);
      keyrow[kb] = valid ? key : -1;
      ok[kb] = valid && kinfo_g[key] != 31;
#pragma unroll
      for (int s = 0; s < 4; s++) {
        kf[kb][s] = *reinterpret_cast<const bf16x8*>(kbase + (int64_t)key * a.kv_ld + 16 * s + 8 * lh);
        vf[kb][s] = *reinterpret_cast<const bf16x8*>(vbase + (int64_t)key * a.kv_ld + 16 * s + 8 * lh);
      }
      khf[kb] = *reinterpret_cast<const bf16x8*>(khot_g + (int64_t)key * 16 + 8 * lh);
      if (!valid) {          // a slot past the block: one-hot of the pad group (15), blocked for every query
#pragma unroll
        for (int j = 0; j < 8; j++) khf[kb][j] = 0;
        if (lh) khf[kb][7] = (short)0x3F80;
      }
    }
    const bool wave_ok[2] = {__all(ok[0]) != 0, __all(ok[1]) != 0};
    const bool wave_dead = !__any(ok[0] || ok[1]);
    // ---- K image: wavefront w moves the rows of its own 64 slots (8 pieces of 1 KiB; swizzle on the source chunk)
#pragma unroll
    for (int p = 0; p < 8; p++) {
      const int r = wave * 64 + p * 8 + (lane >> 3);
      const int key = key_start + (r < n_keys ? r : n_keys - 1);
      B1_DMA(kbase + (int64_t)key * a.kv_ld + (((lane & 7) ^ b1_swz(r)) << 3), kimg + (wave * 8 + p) * 512, 16);
    }
    f32x16 dk[2][2], dv[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int n = 0; n < 2; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) { dk[kb][n][r] = 0.f; dv[kb][n][r] = 0.f; }
    if (wave_dead) {          // its 64 rows of both dS^T images stay zero for the whole block
      for (int i = lane; i < 2 * 64 * 8; i += 64) {
        const int buf = i >> 9, r = (i >> 3) & 63, c = i & 7;
        *reinterpret_cast<uint4*>(dsimg + buf * (TKB * TQ) + (wave * 64 + r) * 64 + c * 8) = make_uint4(0, 0, 0, 0);
      }
    }

    // ---- staging of one step: Q / dO tiles (8 pieces each, two per wavefront), row constants (wavefront 0), mask operand (1, 2)
    auto issue = [&](uint32_t ent, int st) {
      const int qt = (int)(ent & 0x7fffffffu);
      const int2 qd = reinterpret_cast<const int2*>(a.qt_desc)[qt];
      const int row0 = __builtin_amdgcn_readfirstlane(qd.x), nrows = __builtin_amdgcn_readfirstlane(qd.y);
      u16* sb = stage_s + st * B1_STAGE_U16;
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int p = wave * 2 + u, r = p * 8 + (lane >> 3);
        const int row = row0 + (r < nrows ? r : nrows - 1);
        const int sw = ((lane & 7) ^ b1_swz(r)) << 3;
        B1_DMA(qbase + (int64_t)row * a.q_ld + sw, sb + p * 512, 16);
        B1_DMA(obase + (int64_t)row * a.o_ld + sw, sb + TQ * DH + p * 512, 16);
      }
      if (wave == 0) {
        B1_DMA(rowc_g + (int64_t)qt * 128 + lane, sb + 2 * TQ * DH, 4);
        B1_DMA(rowc_g + (int64_t)qt * 128 + 64 + lane, sb + 2 * TQ * DH + 128, 4);
      } else if (wave < 3) {
        const int r = (wave - 1) * 32 + (lane >> 1);
        const int row = row0 + (r < nrows ? r : nrows - 1);
        B1_DMA(a.qblk + (int64_t)row * 16 + (lane & 1) * 8, sb + 2 * TQ * DH + 256 + (wave - 1) * 512, 16);
      }
    };
    __syncthreads();          // list_s visible; every wavefront is past the previous block's last reads of the stages
    if (n_ent > 0) issue(list_s[0], 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int it = 0; it < n_ent; it++) {
      const uint32_t ent = __builtin_amdgcn_readfirstlane(list_s[it]);
      const int qt = (int)(ent & 0x7fffffffu);
      const bool full = (ent >> 31) != 0;
      const int st = it & 1;
      if (it + 1 < n_ent) issue(__builtin_amdgcn_readfirstlane(list_s[it + 1]), st ^ 1);
      const u16* Qs = stage_s + st * B1_STAGE_U16;
      const u16* Os = Qs + TQ * DH;
      const float* rc = reinterpret_cast<const float*>(Qs + 2 * TQ * DH);
      const u16* qblk_s = Qs + 2 * TQ * DH + 256;
      u16* dsw = dsimg + (it & 1) * (TKB * TQ);

      if (!wave_dead) {
#pragma unroll
        for (int qb = 0; qb < 2; qb++) {
          // S^T - lse and dP^T - delta: the row constants are the accumulators' start values
          f32x16 s[2], dp[2];
#pragma unroll
          for (int g = 0; g < 4; g++) {
            const f32x4 cl = *reinterpret_cast<const f32x4*>(rc + qb * 32 + 8 * g + 4 * lh);
            const f32x4 cd = *reinterpret_cast<const f32x4*>(rc + 64 + qb * 32 + 8 * g + 4 * lh);
#pragma unroll
            for (int e = 0; e < 4; e++) { s[0][4 * g + e] = cl[e]; s[1][4 * g + e] = cl[e]; dp[0][4 * g + e] = cd[e]; dp[1][4 * g + e] = cd[e]; }
          }
#pragma unroll
          for (int ks = 0; ks < 4; ks++) {
            const bf16x8 qf = *reinterpret_cast<const bf16x8*>(Qs + qb * 32 * DH + rowfrag[ks]);
            const bf16x8 of = *reinterpret_cast<const bf16x8*>(Os + qb * 32 * DH + rowfrag[ks]);
            s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, kf[0][ks], s[0], 0, 0, 0);
            s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, kf[1][ks], s[1], 0, 0, 0);
            dp[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(of, vf[0][ks], dp[0], 0, 0, 0);
            dp[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(of, vf[1][ks], dp[1], 0, 0, 0);
          }
          // blocked (query, key) pairs: -32768 on top of the score, exp2 is exactly 0 (the mask as a matrix product)
          if (!(full && wave_ok[0] && wave_ok[1])) {
            const bf16x8 qbf = *reinterpret_cast<const bf16x8*>(qblk_s + (qb * 32 + l31) * 16 + 8 * lh);
            s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qbf, khf[0], s[0], 0, 0, 0);
            s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qbf, khf[1], s[1], 0, 0, 0);
          }
          u32x4v pbw[2][2], sbw[2][2];
#pragma unroll
          for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              const float p0 = __builtin_amdgcn_exp2f(s[kb][r]), p1 = __builtin_amdgcn_exp2f(s[kb][r + 1]);
              pbw[kb][r >> 3][(r & 7) >> 1] = pack2bf_pk(p0, p1);
              sbw[kb][r >> 3][(r & 7) >> 1] = pack2bf_pk(p0 * dp[kb][r], p1 * dp[kb][r + 1]);
            }
          // dV^T += dO^T P ; dK^T += Q^T dS   (element j of k-step sp carries query 16 sp + 8 (j >> 2) + 4 lh + (j & 3))
#pragma unroll
          for (int sp = 0; sp < 2; sp++)
#pragma unroll
            for (int n = 0; n < 2; n++) {
              bf16x8 ot, qtf;
#pragma unroll
              for (int t = 0; t < 2; t++) {
                const int qr = qb * 32 + 16 * sp + 8 * t + 4 * lh + tq;
                const int d = n * 32 + 16 * tg + 4 * tp;
                const bf16x4 o4 = lds_read_tr16(Os + b1_off(qr, d >> 3) + (d & 7));
                const bf16x4 q4 = lds_read_tr16(Qs + b1_off(qr, d >> 3) + (d & 7));
#pragma unroll
                for (int e = 0; e < 4; e++) { ot[4 * t + e] = o4[e]; qtf[4 * t + e] = q4[e]; }
              }
#pragma unroll
              for (int kb = 0; kb < 2; kb++) {
                dv[kb][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ot, *reinterpret_cast<const bf16x8*>(&pbw[kb][sp]), dv[kb][n], 0, 0, 0);
                dk[kb][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, *reinterpret_cast<const bf16x8*>(&sbw[kb][sp]), dk[kb][n], 0, 0, 0);
              }
            }
          // dS^T -> [key][query] image: this lane's key row, four consecutive queries per 8-byte store
#pragma unroll
          for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int sp = 0; sp < 2; sp++)
#pragma unroll
              for (int t = 0; t < 2; t++) {
                const int q0 = qb * 32 + 16 * sp + 8 * t + 4 * lh;
                *reinterpret_cast<uint2*>(dsw + b1_off(wave * 64 + kb * 32 + l31, q0 >> 3) + (q0 & 7)) = make_uint2(sbw[kb][sp][2 * t], sbw[kb][sp][2 * t + 1]);
              }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wavefront's pieces of the next stage have landed
      __syncthreads();                                           // dS^T of the step complete; next stage visible

      // ---- dQ^T[d][q] (+)= K^T[d][key] dS^T[key][q] over the block's 256 keys: wavefront w owns the 32 x 32 block
      // (queries 32 (w >> 1) .., d 32 (w & 1) ..) of the 64 x 64 tile
      {
        const int qbw = wave >> 1, nw = wave & 1;
        const bool first = first_s[qt] == kbi, last = last_s[qt] == kbi;
        float* accp = acc_g + (int64_t)qt * (TQ * DH) + wave * 1024 + lane * 4;
        f32x16 dq;
        if (first) {
#pragma unroll
          for (int r = 0; r < 16; r++) dq[r] = 0.f;
        } else {
#pragma unroll
          for (int g = 0; g < 4; g++) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(accp + g * 256);
#pragma unroll
            for (int e = 0; e < 4; e++) dq[4 * g + e] = v[e];
          }
        }
        const u16* dsr = dsimg + (it & 1) * (TKB * TQ);
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
          bf16x8 af, bfr;
#pragma unroll
          for (int t = 0; t < 2; t++) {
            const int key = 16 * kk + 8 * lh + 4 * t + tq;
            const int d = nw * 32 + 16 * tg + 4 * tp, qc = qbw * 32 + 16 * tg + 4 * tp;
            const bf16x4 k4 = lds_read_tr16(kimg + b1_off(key, d >> 3) + (d & 7));
            const bf16x4 s4 = lds_read_tr16(dsr + b1_off(key, qc >> 3) + (qc & 7));
#pragma unroll
            for (int e = 0; e < 4; e++) { af[4 * t + e] = k4[e]; bfr[4 * t + e] = s4[e]; }
          }
          dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, dq, 0, 0, 0);
        }
        if (last) {
          const int2 qd = reinterpret_cast<const int2*>(a.qt_desc)[qt];
          const int r = qbw * 32 + l31;
          if (r < qd.y) {
            u16* p = a.dq + (int64_t)b * a.dq_bstride + (int64_t)(qd.x + r) * a.dq_ld + h * DH + nw * 32 + 4 * lh;
#pragma unroll
            for (int g = 0; g < 4; g++) {
              uint2 pk;
              pk.x = pack2bf(dq[4 * g] * a.scale, dq[4 * g + 1] * a.scale);
              pk.y = pack2bf(dq[4 * g + 2] * a.scale, dq[4 * g + 3] * a.scale);
              *reinterpret_cast<uint2*>(p + 8 * g) = pk;
            }
          }
        } else {
#pragma unroll
          for (int g = 0; g < 4; g++) *reinterpret_cast<f32x4*>(accp + g * 256) = f32x4{dq[4 * g], dq[4 * g + 1], dq[4 * g + 2], dq[4 * g + 3]};
        }
      }
    }

    // ---- the block's dK = ln 2 * dK^T, dV = dV^T + dvmean (uniform rows spread over every key)
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
      if (keyrow[kb] < 0) continue;
      u16* dkp = a.dk + (int64_t)b * a.dkv_bstride + (int64_t)keyrow[kb] * a.dkv_ld + h * DH;
      u16* dvp = a.dv + (int64_t)b * a.dkv_bstride + (int64_t)keyrow[kb] * a.dkv_ld + h * DH;
#pragma unroll
      for (int n = 0; n < 2; n++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const int d = n * 32 + 8 * g + 4 * lh;
          uint2 pk;
          pk.x = pack2bf(dk[kb][n][4 * g] * dk_scale, dk[kb][n][4 * g + 1] * dk_scale);
          pk.y = pack2bf(dk[kb][n][4 * g + 2] * dk_scale, dk[kb][n][4 * g + 3] * dk_scale);
          *reinterpret_cast<uint2*>(dkp + d) = pk;
          const f32x4 dvm = *reinterpret_cast<const f32x4*>(dvm_s + d);
          pk.x = pack2bf(dv[kb][n][4 * g] + dvm[0], dv[kb][n][4 * g + 1] + dvm[1]);
          pk.y = pack2bf(dv[kb][n][4 * g + 2] + dvm[2], dv[kb][n][4 * g + 3] + dvm[3]);
          *reinterpret_cast<uint2*>(dvp + d) = pk;
        }
    }
    __syncthreads();          // every wavefront is done with the K image, the list and the stages before the next block's prologue
  }
  // key blocks without a valid key in this sample sweep nothing: dK = 0, dV = dvmean
  for (int kbi = 0; kbi < a.n_kblocks; kbi++) {
    if (live_s[kbi]) continue;
    const int4 kd = reinterpret_cast<const int4*>(a.kb_desc)[kbi];
    for (int i = tid; i < kd.y * 8; i += 256) {
      const int key = kd.x + (i >> 3), c = i & 7;
      *reinterpret_cast<uint4*>(a.dk + (int64_t)b * a.dkv_bstride + (int64_t)key * a.dkv_ld + h * DH + c * 8) = make_uint4(0, 0, 0, 0);
      uint4 pv;
      pv.x = pack2bf(dvm_s[c * 8], dvm_s[c * 8 + 1]); pv.y = pack2bf(dvm_s[c * 8 + 2], dvm_s[c * 8 + 3]);
      pv.z = pack2bf(dvm_s[c * 8 + 4], dvm_s[c * 8 + 5]); pv.w = pack2bf(dvm_s[c * 8 + 6], dvm_s[c * 8 + 7]);
      *reinterpret_cast<uint4*>(a.dv + (int64_t)b * a.dkv_bstride + (int64_t)key * a.dkv_ld + h * DH + c * 8) = pv;
    }
  }
}
// =====================================================================================================
// The production form of the kernel above: the same work decomposition, as a SOFTWARE PIPELINE whose issue order is generated
// (tools/gen_bwd1_schedule.py -> attention_bwd1_sched.inc: one slot = one MFMA and the single-issue work in its shadow).
//   * register classes by hand (every MFMA is inline asm): dK^T / dV^T accumulators, the K / V operand fragments and the Q / dO
//     row fragments and the two dQ accumulators live in the accumulator half; scores (the row constants are read straight into
//     them), packed P / dS, the V fragments and the transposed fragments in the vector half.  (Left to hipcc, the plain form moves 730 registers between the halves per step.)
//   * the loop is rotated: iteration X computes the scores of blocks 0..3 of step X, and finishes step X-1 (block 3's vector work
//     fills the shadow of A0; the dV / dK products of its blocks 2 and 3; its dQ product behind the iteration's ONE barrier); Q / dO / row-constant /
//     mask-operand tiles arrive by LDS-DMA two steps ahead (three stages), the dQ partial of a step's tile is requested a
//     whole iteration ahead of its product.  The loop's 13 vector-memory operations are unconditional and in a fixed order, so
//     its two waits are compile-time vmcnt immediates (B1_W1_YOUNGER / B1_W2_YOUNGER of the generated schedule).
//   * key blocks are pipelined across their boundaries (the next block's images, stages and flags are requested behind the barrier
//     that ends a block's loop, ahead of its dK / dV epilogue); every step record is built once at kernel start; small batches
//     deal the key blocks of a (sample, head) to several workgroups (split) and add their dQ slices in a second launch.
//   * an inline-asm MFMA gets no hazard padding from hipcc: the schedule keeps LAG slots between a matrix instruction and the
//     first vector reader of its result; the two places outside the schedule (dQ store, block epilogue) pad by hand.
// =====================================================================================================
#define B1P_NST 3
#ifndef B1_ABL          // timing-only ablation builds (tools/ablate_bwd1.py): 1 no vector work, 2 no dQ product, 4 no transposed reads,
#define B1_ABL 0        // 8 no row reads, 16 no dS^T stores, 32 no score MFMAs, 64 no dV / dK MFMAs, 128 no vector memory in the loop.  0 in the product.
#endif
#define B1_SB() __builtin_amdgcn_sched_barrier(0)
// (PAD: "s_nop 1" ahead of the MFMA inside the statement - two wait states between a compiler-placed register copy or select
//  and the MFMA that reads it.  The first and the drain iteration of a key block are padded (hipcc copies the zeroed accumulators
//  and the carried state into place just ahead of their first reader there: tools/audit_bwd1_isa.py found them 0-1 states
//  ahead); the steady-state loop is not, and the audit checks every build of it.)
#define MF_ACC_AV(D, A, B) do { if (STEADY) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(D) : "a"(A), "v"(B)); \
                                else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(D) : "a"(A), "v"(B)); } while (0)
#define MF_ACC_AA(D, A, B) do { if (STEADY) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(D) : "a"(A), "a"(B)); \
                                else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(D) : "a"(A), "a"(B)); } while (0)
#define MF_ACCA_VV(D, A, B) do { if (STEADY) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(D) : "v"(A), "v"(B)); \
                                 else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(D) : "v"(A), "v"(B)); } while (0)
#define MF_ACC_VV(D, A, B) do { if (STEADY) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(D) : "v"(A), "v"(B)); \
                                else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(D) : "v"(A), "v"(B)); } while (0)
// The two dQ accumulators are OWNED registers, a[224:239] and a[240:255]: they are the landing registers of asynchronous loads (the
// old partial arrives while other work runs), and a value hipcc can see it may copy, spill or reuse the moment the loading
// statement ends - it did: copies at the loop's back edge and a fragment read into registers whose load was still in flight.
// Every statement that touches them names them and lists them as clobbers; tools/audit_bwd1_isa.py proves nothing else does.
#define ACC_CLOB_0 "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239"
#define ACC_CLOB_1 "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
// one 1-KiB piece G of this wavefront's 4-KiB slot: SBASE wave-uniform (scalar registers), VOFF = 16 * lane
template <int BUF, int G> __device__ __forceinline__ void b1_acc_load1(unsigned voff, const float* sbase) {
#define L_(R, O) asm volatile("global_load_dwordx4 " R ", %0, %1 offset:" O :: "v"(voff), "s"(sbase) : "memory", ACC_CLOB_0, ACC_CLOB_1)
  if (BUF == 0) { if (G == 0) L_("a[224:227]", "0"); else if (G == 1) L_("a[228:231]", "1024"); else if (G == 2) L_("a[232:235]", "2048"); else L_("a[236:239]", "3072"); }
  else { if (G == 0) L_("a[240:243]", "0"); else if (G == 1) L_("a[244:247]", "1024"); else if (G == 2) L_("a[248:251]", "2048"); else L_("a[252:255]", "3072"); }
#undef L_
}
template <int BUF, int G> __device__ __forceinline__ void b1_acc_store1(unsigned voff, float* sbase) {
#define S_(R, O) asm volatile("global_store_dwordx4 %0, " R ", %1 offset:" O :: "v"(voff), "s"(sbase) : "memory")
  if (BUF == 0) { if (G == 0) S_("a[224:227]", "0"); else if (G == 1) S_("a[228:231]", "1024"); else if (G == 2) S_("a[232:235]", "2048"); else S_("a[236:239]", "3072"); }
  else { if (G == 0) S_("a[240:243]", "0"); else if (G == 1) S_("a[244:247]", "1024"); else if (G == 2) S_("a[248:251]", "2048"); else S_("a[252:255]", "3072"); }
#undef S_
}
template <int BUF, int G> __device__ __forceinline__ void b1_acc_read4(float (&t)[4]) {
#define R_(A, B, C, D) asm volatile("v_accvgpr_read_b32 %0, " A "\n\tv_accvgpr_read_b32 %1, " B "\n\tv_accvgpr_read_b32 %2, " C "\n\tv_accvgpr_read_b32 %3, " D \
                                    : "=v"(t[0]), "=v"(t[1]), "=v"(t[2]), "=v"(t[3]))
  if (BUF == 0) { if (G == 0) R_("a224", "a225", "a226", "a227"); else if (G == 1) R_("a228", "a229", "a230", "a231"); else if (G == 2) R_("a232", "a233", "a234", "a235"); else R_("a236", "a237", "a238", "a239"); }
  else { if (G == 0) R_("a240", "a241", "a242", "a243"); else if (G == 1) R_("a244", "a245", "a246", "a247"); else if (G == 2) R_("a248", "a249", "a250", "a251"); else R_("a252", "a253", "a254", "a255"); }
#undef R_
}
// the dQ product's MFMAs on an owned accumulator (always padded: two per iteration carry the pad, the other fourteen do not need
// it - their operands come from LDS reads - but one form keeps the strings few); ZERO: the first product of a tile's first visit
template <int BUF, bool ZERO, bool PAD> __device__ __forceinline__ void b1_acc_mfma(const bf16x8& a_, const bf16x8& b_) {
  if (BUF == 0) {
    if (ZERO) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[224:239], %0, %1, 0" :: "v"(a_), "v"(b_) : ACC_CLOB_0);
    else if (PAD) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[224:239], %0, %1, a[224:239]" :: "v"(a_), "v"(b_) : ACC_CLOB_0);
    else asm volatile("v_mfma_f32_32x32x16_bf16 a[224:239], %0, %1, a[224:239]" :: "v"(a_), "v"(b_) : ACC_CLOB_0);
  } else {
    if (ZERO) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[240:255], %0, %1, 0" :: "v"(a_), "v"(b_) : ACC_CLOB_1);
    else if (PAD) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[240:255], %0, %1, a[240:255]" :: "v"(a_), "v"(b_) : ACC_CLOB_1);
    else asm volatile("v_mfma_f32_32x32x16_bf16 a[240:255], %0, %1, a[240:255]" :: "v"(a_), "v"(b_) : ACC_CLOB_1);
  }
}
// LDS-DMA by hand (M0 written in the same statement): hipcc then knows of no LDS write in flight - with the builtin it drained
// vmcnt(0) in front of the next LDS read, i.e. every iteration waited for ALL of its memory traffic.  BASE: wave-uniform 64-bit
// pointer, OFF: this lane's 32-bit byte offset, LDS_DST: wave-uniform LDS byte address (lane l lands at + 16 l / + 4 l).  The kernel's
// own counted s_waitcnt vmcnt + s_barrier order the data.
#define B1_DMA16(BASE, OFF, LDS_DST)                                                                                                       \
  do { unsigned keep_;                                                                                                                     \
       asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"    \
                    : "=&s"(keep_) : "v"(OFF), "s"(BASE), "s"(LDS_DST) : "memory"); } while (0)
#define B1_DMA4(BASE, OFF, LDS_DST)                                                                                                        \
  do { unsigned keep_;                                                                                                                     \
       asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"      \
                    : "=&s"(keep_) : "v"(OFF), "s"(BASE), "s"(LDS_DST) : "memory"); } while (0)
// the loop's form: BASE / LDS_DST come out of scalar arithmetic (no vector-written scalar register: no five-state hazard); M0 is
// not saved (a reserved register for hipcc: it holds no value across a statement, and writes it itself right ahead of the
// instructions that read it); IMM: the instruction's immediate byte offset - added to the global address AND to the LDS address
#define B1_DMA16L(BASE, OFF, LDS_DST, IMM)                                                                                                 \
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:" #IMM :: "v"(OFF), "s"(BASE), "s"(LDS_DST) : "memory")
#define B1_DMA4L(BASE, OFF, LDS_DST)                                                                                                       \
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" :: "v"(OFF), "s"(BASE), "s"(LDS_DST) : "memory")
// LDS accesses by 32-bit byte address = per-lane base register + compile-time immediate (pointer arithmetic on u16* made hipcc
// keep one address register per constant offset and spill them)
#define LDS_P(T, ADDR) reinterpret_cast<__attribute__((address_space(3))) T*>(static_cast<uintptr_t>(ADDR))
#define LDS_R8(ADDR) (*LDS_P(const bf16x8, ADDR))
#define LDS_RF4(ADDR) (*LDS_P(const f32x4, ADDR))
#define LDS_TR(ADDR) __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_P(bf16x4, ADDR))

MCA_TRACE_BUFFER(attn_bwd1)      // trace build, knob 9 bit 8: s_memtime stamps of wavefront 0 of workgroup 0 at the block-level points (tools/trace_bwd1.py)
#ifdef MCA_TRACE_BUILD
#define B1_STAMP() do { if (trace_on) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) mca_trace_attn_bwd1[tr_n & 1023] = t_; tr_n++; } } while (0)
#define B1_NOTE(V) do { if (trace_on) { if (lane == 0) mca_trace_attn_bwd1[tr_n & 1023] = (unsigned long long)(V); tr_n++; } } while (0)
// inside the step loop: s_memtime into scalar registers (no wait at the stamp: read at the end of the iteration)
#define B1_TR(K_) asm volatile("s_memtime %0" : "=s"(tr_t[K_]))
#define B1_TR_END() do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr_t[6]) :: "memory");                      \
                         if (trace_on && it == 8 && lane == 0) {                                                                 \
                           for (int k_ = 0; k_ < 7; k_++) mca_trace_attn_bwd1[512 + tr_blk * 8 + k_] = tr_t[k_]; } } while (0)
#else
#define B1_STAMP() do {} while (0)
#define B1_NOTE(V) do {} while (0)
#define B1_TR(K_) do {} while (0)
#define B1_TR_END() do {} while (0)
#endif
// a step's record {tile | (1 first, 2 last visit of the tile by this (sample, head)) << 16 | rows << 20, first row}
#define M_QT(M) ((M).x & 0xffff)
#define M_FIRST(M) (((M).x >> 16) & 1)
#define M_LAST(M) (((M).x >> 17) & 1)
#define M_ROWS(M) ((int)((unsigned)(M).x >> 20))
#define M_ROW0(M) ((M).y)

// (the body is a __device__ function: the host pass of hipcc checks the register constraints of inline asm in a __global__ body
//  against the host's register classes)
__device__ __forceinline__ void attn_bwd1p_body(const mca_attn_bwd1_args& a, const int dbg, u16* lds) {
  u16* stage_s = lds;                                             // B1P_NST stages
  u16* kimg = stage_s + B1P_NST * B1_STAGE_U16;
  u16* dsimg = kimg + TKB * DH;
  int2* meta_s = reinterpret_cast<int2*>(dsimg + 2 * TKB * TQ);           // every live key block's steps: {tile | (1 first, 2 last visit) << 16 | rows << 20, first row}
  int4* kbd_s = reinterpret_cast<int4*>(meta_s + B1_MAX_META);            // key block table {first key, keys, first record in meta_s, steps}
  // tables only the set-up below reads live in the (still unused) first dS^T image: the query tile table {first row, rows} and
  // each tile's first / last live key block (2.5 KiB that the step records can have instead: LONG needs 720 of them)
  int2* qtd_s = reinterpret_cast<int2*>(dsimg);
  uint8_t* first_s = reinterpret_cast<uint8_t*>(qtd_s + B1_MAX_QT);
  uint8_t* last_s = first_s + B1_MAX_QT;
  uint8_t* live_s = reinterpret_cast<uint8_t*>(kbd_s + B1_MAX_KB);
  float* dvm_s = reinterpret_cast<float*>(live_s + B1_MAX_KB);

  const int lin = (dbg & 16) ? (int)blockIdx.x : xcd_remap((int)blockIdx.x, (int)gridDim.x);
  // SPLIT mode (a.split = S > 1; batches too small to give every CU a (sample, head)): S workgroups share a (sample, head), workgroup
  // s owns the key blocks kb with kb mod S == s.  Each keeps its own fp32 dQ partials in its own slice of dq_acc (no visit is a
  // "last" one: nothing is written to dq here); attn_bwd1_reduce_kernel adds the S slices in a fixed order afterwards.
  const int S = a.split > 1 ? a.split : 1;
  const int bh_ = lin / S, sid = lin - bh_ * S;
  const int h = bh_ % a.heads, b = bh_ / a.heads;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef MCA_TRACE_BUILD
  const bool trace_on = lin == 0 && wave == 0 && (dbg & 8);
  int tr_n = 0, tr_blk = 0;
#endif
  const int l31 = lane & 31, lh = lane >> 5;
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  const int64_t bh = (int64_t)b * a.heads + h;

  const u16* qbase = a.q + (int64_t)b * a.q_bstride + h * (a.q_hstride ? a.q_hstride : (int64_t)DH);
  const u16* obase = a.d_o + (int64_t)b * a.o_bstride + h * (a.o_hstride ? a.o_hstride : (int64_t)DH);
  const u16* kbase = a.k + (int64_t)b * a.kv_bstride + h * DH;
  const u16* vbase = a.v + (int64_t)b * a.kv_bstride + h * DH;
  const float* rowc_g = a.rowc + bh * (int64_t)(a.n_qtiles + 1) * 128;          // (+ the null tile)
  float* acc_s = a.dq_acc + (bh * S + sid) * (int64_t)(a.n_qtiles + 1) * (TQ * DH) + wave * 1024;          // (wave-uniform; lane l's slot at + 4 l)
  const uint8_t* kinfo_g = a.keyinfo + (int64_t)b * a.nk_pad;
  const u16* khot_g = a.khot + (int64_t)b * a.nk_pad * 16;

  if (tid < a.n_kblocks) {
    const int4 d = reinterpret_cast<const int4*>(a.kb_desc)[tid];
    const uint8_t* fl = a.ktile_flags + (int64_t)b * a.n_ktiles64;
    int live = 0;
    for (int t = d.x >> 6; t <= (d.x + d.y - 1) >> 6; t++) live |= fl[t];
    const bool mine = tid % S == sid;
    live_s[tid] = (live && mine) ? 1 : ((!live && mine) ? 2 : 0);          // 1: this workgroup sweeps it; 2: its keys are all padded, this workgroup writes dK = 0, dV = dvmean
    kbd_s[tid] = d;
  }
  if (tid < DH) dvm_s[tid] = a.dvmean[(int64_t)b * a.heads * DH + h * DH + tid];
  for (int qt = tid; qt < a.n_qtiles; qt += 256) qtd_s[qt] = reinterpret_cast<const int2*>(a.qt_desc)[qt];
  if (tid == 0) { qtd_s[a.n_qtiles] = make_int2(0, 1); first_s[a.n_qtiles] = 255; last_s[a.n_qtiles] = 255; }          // the null tile
  __syncthreads();
  for (int qt = tid; qt < a.n_qtiles; qt += 256) {
    int first = 255, last = 255;
    for (int kb = 0; kb < a.n_kblocks; kb++)
      if (live_s[kb] == 1 && a.visit[kb * a.n_qtiles + qt]) { if (first == 255) first = kb; last = kb; }
    first_s[qt] = (uint8_t)first; last_s[qt] = (uint8_t)last;
  }
  // ---- the step records of EVERY live key block, once (per block this was a global read of the block's list in front of its
  // first DMA: one memory latency per block, eleven per workgroup).  A block's records: its list, then the NULL steps - tile
  // n_qtiles = {row 0, one row}, row constants -inf | 0: one ends every sweep (the loop is rotated: it finishes the last real
  // step), a second one makes the iteration count even (the loop is unrolled by two), two more are read ahead by the last iterations
  int* mtot_s = reinterpret_cast<int*>(dvm_s + DH);
  if (tid == 0) {
    int off = 0;
    for (int kb = 0; kb < a.n_kblocks; kb++) {
      int4 d = kbd_s[kb];
      const int e_begin = d.z;
      d.z = off;
      if (live_s[kb] == 1) off += ((d.w + 2) & ~1) + 2;
      kbd_s[kb] = make_int4(d.x, d.y, d.z | (e_begin << 16), d.w);          // (record offset < 2^16, list offset < 2^15)
    }
    *mtot_s = off;
  }
  __syncthreads();
  {          // (one flat pass: every thread's global read of its list entry is in flight at once)
    const int total = *mtot_s;
    for (int e = tid; e < total; e += 256) {
      int kb = 0;
      for (int k = 1; k < a.n_kblocks; k++) if ((kbd_s[k].z & 0xffff) <= e) kb = k;          // the LAST block whose records start at or before e
      while (live_s[kb] != 1) kb--;          // (a dead block shares its offset with the next live one; block offsets ascend)
      const int4 d = kbd_s[kb];
      const int mo = d.z & 0xffff, e_begin = d.z >> 16, n_ent = d.w, i = e - mo;
      const int qt = i < n_ent ? (int)(a.kb_qt[e_begin + i] & 0x7fffffffu) : a.n_qtiles;
      const int2 qd = qtd_s[qt];
      meta_s[e] = make_int2(qt | (((int)first_s[qt] == kb ? 1 : 0) << 16) | (((int)last_s[qt] == kb && S == 1 ? 2 : 0) << 16) | (qd.y << 20), qd.x);
    }
  }
  __syncthreads();
  for (int qt = 0; qt < a.n_qtiles; qt++) {          // query tiles this workgroup never visits: dq = 0 (split mode: its partial = 0)
    if (first_s[qt] != 255) continue;
    const int2 qd = qtd_s[qt];
    if (S == 1) {
      for (int i = tid; i < qd.y * 8; i += 256)
        *reinterpret_cast<uint4*>(a.dq + (int64_t)b * a.dq_bstride + (int64_t)(qd.x + (i >> 3)) * a.dq_ld + h * DH + (i & 7) * 8) = make_uint4(0, 0, 0, 0);
    } else {
      float* slot = a.dq_acc + ((bh * S + sid) * (int64_t)(a.n_qtiles + 1) + qt) * (TQ * DH);
      for (int i = tid; i < TQ * DH / 4; i += 256) *reinterpret_cast<f32x4*>(slot + 4 * i) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __syncthreads();          // (the set-up tables inside the first dS^T image are dead from here on)

  const unsigned lds_b = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const u16*)lds;
  const unsigned STAGE_B = 2u * B1_STAGE_U16, KIMG_B = lds_b + 2u * B1P_NST * B1_STAGE_U16, DSIMG_B = KIMG_B + 2u * TKB * DH;
  const float dk_scale = 0.6931471805599453f;

  // ---- the key blocks of this (sample, head), software-pipelined ACROSS blocks: the next block's K image, first two stages and
  // K / V fragments are requested right behind the barrier that ends a block's loop, ahead of that block's dK / dV epilogue (as a
  // prologue of its own they were three memory latencies in a row per block, eleven blocks per workgroup)
    // ---- staging of one step: five 1-KiB / 256-byte pieces per wavefront (Q and dO rows 16 w .. 16 w + 15, and one of: -lse, -delta,
  // the two halves of the mask operand): every wavefront's vmcnt sees the same count
  // per-lane byte offsets of this wavefront's two Q / dO pieces inside a tile (row 8 p + lane / 8, swizzled 16-byte chunk): the
  // packed copies are read WITHOUT clamping to the tile's rows (rows past them belong to the next tile or to the 63 rows of
  // slack behind the buffer: finite values behind row constants of -inf)
  // piece P of a stage: 0 / 2 the two Q pieces of this wavefront, 1 / 3 the dO pieces, 4 its share of the row constants / mask operand
  auto dma_piece = [&]<int P>(const int2& m, int st) __attribute__((always_inline)) {          // m: the step's record (scalar registers)
    const unsigned sb_ = lds_b + (unsigned)st * STAGE_B;
    int ln = lane;          // (offsets derived from the lane id HERE: hoisted out of the loop they were registers to spill)
    asm volatile("" : "+v"(ln));
    if (P < 4) {
      constexpr int u = P >> 1;
      const u16* rowp = ((P & 1) ? obase : qbase) + (int64_t)M_ROW0(m) * DH;
      const int r = (wave * 2 + u) * 8 + (ln >> 3);
      const unsigned o_dma = (unsigned)(r * 128 + (((ln & 7) ^ b1_swz(r)) << 4));
      B1_DMA16(rowp, o_dma, sb_ + ((P & 1) ? 8192u : 0u) + (unsigned)(wave * 2 + u) * 1024u);
    } else if (wave < 2) {
      const float* rb = rowc_g + (int64_t)M_QT(m) * 128 + wave * 64;
      B1_DMA4(rb, (unsigned)ln * 4u, sb_ + 16384u + (unsigned)wave * 256u);
    } else {
      const int r = (wave - 2) * 32 + (ln >> 1), nrows = M_ROWS(m);
      const int row = M_ROW0(m) + (r < nrows ? r : nrows - 1);
      B1_DMA16(a.qblk, (unsigned)row * 32u + (unsigned)(ln & 1) * 16u, sb_ + 16896u + (unsigned)(wave - 2) * 1024u);
    }
  };
  auto next_live = [&](int from) { int k = from; while (k < a.n_kblocks && live_s[k] != 1) k++; return k < a.n_kblocks ? k : -1; };
  bf16x8 kf[2][4], vf[2][4], khf[2];
  uint8_t kflag[2];          // (raw key flags: compared at the top of the block, so that the wait for them sits there)
  int key_start = 0, n_keys = 0, meta_off = 0, n_ent = 0;
  // everything of key block kb that comes from memory: K and V images (LDS-DMA, 8 + 8 pieces of 1 KiB: whole 128-byte rows - as
  // fragment loads straight from memory, one 16-byte piece of its own row per lane, they were 64 cache lines per instruction and,
  // with the epilogue's stores, more operations than the 63 a wavefront may have in flight: ~20,000 cycles per block), the stages
  // of its first two steps (10 pieces; a key block has at least one real step and the null step), this wavefront's mask operand
  // and key flags
  auto issue_block = [&](const int kb) __attribute__((always_inline)) {
    const int4 kd = kbd_s[kb];
    key_start = __builtin_amdgcn_readfirstlane(kd.x); n_keys = __builtin_amdgcn_readfirstlane(kd.y);
    meta_off = __builtin_amdgcn_readfirstlane(kd.z) & 0xffff; n_ent = __builtin_amdgcn_readfirstlane(kd.w);
#pragma unroll
    for (int p = 0; p < 8; p++) {          // K image: wavefront w moves the rows of its own 64 slots
      const int r = wave * 64 + p * 8 + (lane >> 3);
      const int key = key_start + (r < n_keys ? r : n_keys - 1);
      B1_DMA16(kbase, (unsigned)key * (unsigned)(a.kv_ld * 2) + (unsigned)(((lane & 7) ^ b1_swz(r)) << 4), KIMG_B + (unsigned)(wave * 8 + p) * 1024u);
    }
#pragma unroll
    for (int p = 0; p < 8; p++) {          // V rows of the same slots, into this wavefront's own rows of the second dS^T image (free until the loop)
      const int r = wave * 64 + p * 8 + (lane >> 3);
      const int key = key_start + (r < n_keys ? r : n_keys - 1);
      B1_DMA16(vbase, (unsigned)key * (unsigned)(a.kv_ld * 2) + (unsigned)(((lane & 7) ^ b1_swz(r)) << 4), DSIMG_B + 2u * TKB * TQ + (unsigned)(wave * 8 + p) * 1024u);
    }
    int2 r0 = meta_s[meta_off], r1 = meta_s[meta_off + 1];
    r0 = make_int2(__builtin_amdgcn_readfirstlane(r0.x), __builtin_amdgcn_readfirstlane(r0.y));
    r1 = make_int2(__builtin_amdgcn_readfirstlane(r1.x), __builtin_amdgcn_readfirstlane(r1.y));
    dma_piece.template operator()<0>(r0, 0); dma_piece.template operator()<1>(r0, 0); dma_piece.template operator()<2>(r0, 0);
    dma_piece.template operator()<3>(r0, 0); dma_piece.template operator()<4>(r0, 0);
    dma_piece.template operator()<0>(r1, 1); dma_piece.template operator()<1>(r1, 1); dma_piece.template operator()<2>(r1, 1);
    dma_piece.template operator()<3>(r1, 1); dma_piece.template operator()<4>(r1, 1);
#pragma unroll
    for (int kb2 = 0; kb2 < 2; kb2++) {          // (32-byte rows of consecutive keys: coalesced)
      const int slot = wave * 64 + kb2 * 32 + l31;
      const int key = key_start + (slot < n_keys ? slot : n_keys - 1);
      khf[kb2] = *reinterpret_cast<const bf16x8*>(khot_g + (int64_t)key * 16 + 8 * lh);          // (raw: a slot without a key is fixed at the top of the block)
      kflag[kb2] = kinfo_g[key];
    }
  };
  int kbi = __builtin_amdgcn_readfirstlane(next_live(0));
  if (kbi >= 0) issue_block(kbi);
  while (kbi >= 0) {
    B1_STAMP();          // 0: top of the block
    const int n_it = (n_ent + 2) & ~1;          // (the real steps, the null step, and one more to make the count even)
    const int2* meta_b = meta_s + meta_off;
    const int ks_cur = key_start, nk_cur = n_keys;
    // everything issue_block requested has arrived (and the previous block's epilogue stores have left)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {          // this wavefront's K / V operand fragments: rows of its own 64 slots of the two images
      int ln = lane;
      asm volatile("" : "+v"(ln));
#pragma unroll
      for (int kb2 = 0; kb2 < 2; kb2++)
#pragma unroll
        for (int st = 0; st < 4; st++) {
          const unsigned o = (unsigned)((wave * 64 + kb2 * 32) * 128) + 2u * (unsigned)b1_off(ln & 31, 2 * st + (ln >> 5));
          kf[kb2][st] = LDS_R8(KIMG_B + o);
          vf[kb2][st] = LDS_R8(DSIMG_B + 2u * TKB * TQ + o);
        }
    }
    const bool wave_dead = !__any((wave * 64 + l31 < n_keys && kflag[0] != 31) || (wave * 64 + 32 + l31 < n_keys && kflag[1] != 31));
    // (the mask operands exist HERE: left to hipcc, their selects sank to just ahead of the first inline-asm MFMA that reads them -
    //  a vector write needs two wait states before an MFMA reads it, and hipcc pads nothing around inline asm: wrong scores on
    //  some wavefronts)
#pragma unroll
    for (int kb2 = 0; kb2 < 2; kb2++)
      if (!(wave * 64 + kb2 * 32 + l31 < n_keys)) {          // a slot past the block's keys: the padding group's one-hot (masked against every query)
#pragma unroll
        for (int j = 0; j < 8; j++) khf[kb2][j] = 0;
        if (lh) khf[kb2][7] = (short)0x3F80;
      }
    asm volatile("" : "+v"(khf[0]), "+v"(khf[1]));
    B1_STAMP();          // 1: the block's fragments and flags have arrived
    f32x16 dk[2][2], dv[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int n = 0; n < 2; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) { dk[kb][n][r] = 0.f; dv[kb][n][r] = 0.f; }
    if (wave_dead) {          // its 64 rows of both dS^T images stay zero for the whole block
      for (int i = lane; i < 2 * 64 * 8; i += 64) {
        const int buf = i >> 9, r = (i >> 3) & 63, c = i & 7;
        *reinterpret_cast<uint4*>(dsimg + buf * (TKB * TQ) + (wave * 64 + r) * 64 + c * 8) = make_uint4(0, 0, 0, 0);
      }
    }

    // K image, stages 0 and 1 of this block have landed (older than the fragment loads waited for above); every wavefront is here
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    B1_STAMP();          // 2: behind the block's first barrier

    // ---- pipeline state
    // (the dQ accumulators of two steps are the owned registers a[224:255]: buffer s & 1 receives the OLD partial of step s's tile
    //  two iterations before step s's dQ product - iteration s + 1 - accumulates onto it in place)
    f32x16 S[2], dP[2];
    u32x4v pb[2][2], sb[2][2];
    bf16x8 rfQ[4], rfO[4], qbf, trO[2][2], trQ[2][2], da[2], db[2];
    // (deliberately uninitialised: every element is written before its first real use, and a zero held 'just in case' stays live -
    //  48 registers of zeros through the loop in an earlier build)
    // ---- LDS BYTE offsets of this lane inside a stage / an image: derived HERE from an opaque copy of the lane id (held across the
    // key-block loop they were sixteen more live registers through the pipeline)
    int ln_ = lane;
    asm volatile("" : "+v"(ln_));
    const int l31_ = ln_ & 31, lh_ = ln_ >> 5, tq_ = (ln_ & 15) >> 2, tp_ = ln_ & 3, tg_ = (ln_ >> 4) & 1;
    unsigned o_rf[4];          // row fragment (row l31_, k-step ks) of a 32-row block of a [rows][64] image
  #pragma unroll
    for (int ks = 0; ks < 4; ks++) o_rf[ks] = 2u * (unsigned)b1_off(l31_, 2 * ks + lh_);
    unsigned o_tr[2][2], o_dqa[2], o_dqb[2];          // transposed reads: [t][n] of a Q / dO tile (rows 8 t + 4 lh_ + tq_); dQ operands (rows 8 lh_ + 4 t + tq_)
  #pragma unroll
    for (int t = 0; t < 2; t++) {
  #pragma unroll
      for (int n = 0; n < 2; n++) o_tr[t][n] = 2u * (unsigned)(b1_off(8 * t + 4 * lh_ + tq_, n * 4 + 2 * tg_ + (tp_ >> 1)) + 4 * (tp_ & 1));
      o_dqa[t] = 2u * (unsigned)(b1_off(8 * lh_ + 4 * t + tq_, (wave & 1) * 4 + 2 * tg_ + (tp_ >> 1)) + 4 * (tp_ & 1));
      o_dqb[t] = 2u * (unsigned)(b1_off(8 * lh_ + 4 * t + tq_, (wave >> 1) * 4 + 2 * tg_ + (tp_ >> 1)) + 4 * (tp_ & 1));
    }
    unsigned o_ds[2];          // dS^T store: (row * 128 + 8 lh_) ^ (swizzle << 4); the 16-byte chunk c of the row is at o_ds ^ (c << 4)
  #pragma unroll
    for (int kb = 0; kb < 2; kb++) { const int r = wave * 64 + kb * 32 + l31_; o_ds[kb] = (unsigned)((r * 128 + 8 * lh_) ^ (b1_swz(r) << 4)); }
    const unsigned o_rc = 16u * (unsigned)lh_, o_qb = (unsigned)(l31_ * 32 + 16 * lh_);
    unsigned a_rf[4], a_tr[2][2], a_da[2], a_db[2], a_rc = lds_b + 16384u + o_rc, a_qb = lds_b + 16896u + o_qb;
    // dS^T store addresses of this lane in the two images (the image bases are multiples of 128 bytes, so the chunk swizzle - an xor on
    // bits 4..6 - applies to the sum: one xor per store instead of an xor and an add)
    unsigned a_dsw[2][2] = {{DSIMG_B + o_ds[0], DSIMG_B + o_ds[1]}, {DSIMG_B + 2u * TKB * TQ + o_ds[0], DSIMG_B + 2u * TKB * TQ + o_ds[1]}};
#pragma unroll
    for (int ks = 0; ks < 4; ks++) a_rf[ks] = lds_b + o_rf[ks];
#pragma unroll
    for (int t = 0; t < 2; t++) {
      a_tr[t][0] = lds_b + o_tr[t][0]; a_tr[t][1] = lds_b + o_tr[t][1];
      a_da[t] = KIMG_B + o_dqa[t]; a_db[t] = DSIMG_B + (2u * TKB * TQ) + o_dqb[t];          // (iteration 0 reads nothing from it)
    }
    // The vector-memory operations of an iteration are UNCONDITIONAL and in a fixed order - four stores of the dQ block of step
    // it - 2 (a step that does not exist has the null record: tile n_qtiles, a slot of dq_acc nobody reads), four loads of the
    // old partial of step it's tile (a first visit loads its slot's stale contents: the first product of a first visit starts from
    // the constant 0 instead), five DMA pieces of step it + 2's stage (behind the last step: the null tile's) - so that the two
    // waits of an iteration are compile-time counts (attention_bwd1_sched.inc: B1_W1_YOUNGER, B1_W2_YOUNGER) and no counter,
    // branch or jump table is needed (they were ~200 scalar instructions per iteration, 4 issue cycles each, in a loop whose
    // vector pipe is the scarce resource).
    // records of steps it + 1, it, it - 1, it - 2, in scalar registers
    const int null_rec = a.n_qtiles | (1 << 20);
    int2 m1 = meta_b[1], m0 = meta_b[0], mp = make_int2(null_rec | (1 << 16), 0), mpp = make_int2(null_rec, 0);
    m1 = make_int2(__builtin_amdgcn_readfirstlane(m1.x), __builtin_amdgcn_readfirstlane(m1.y));
    m0 = make_int2(__builtin_amdgcn_readfirstlane(m0.x), __builtin_amdgcn_readfirstlane(m0.y));
#define B1_STR_(X) #X
#define B1_STR(X) B1_STR_(X)
    // piece G of the dQ block of a tile's LAST visit (buffer BUF): scaled, as bf16, into dq
    auto store_dq_last = [&]<int BUF, int G>(const int2& m) __attribute__((always_inline)) {
      int ln = lane;          // (addresses from an opaque copy of the lane id: held across the loop they were spilled)
      asm volatile("" : "+v"(ln));
      float t[4];
      b1_acc_read4<BUF, G>(t);
      const int r = (wave >> 1) * 32 + (ln & 31);
      u16* dqb = a.dq + (int64_t)b * a.dq_bstride + h * DH + (wave & 1) * 32 + 8 * G;          // (wave-uniform)
      // (a lane past the tile's rows stores to its own dq_acc slot, which nobody reads: every lane issues the store)
      u16* p = r < M_ROWS(m) ? dqb + (size_t)(unsigned)((M_ROW0(m) + r) * (int)a.dq_ld + 4 * (ln >> 5))
                             : reinterpret_cast<u16*>(acc_s + ((int64_t)M_QT(m) << 12) + G * 256 + ln * 4);
      uint2 pk;
      pk.x = pack2bf_pk(t[0] * a.scale, t[1] * a.scale);
      pk.y = pack2bf_pk(t[2] * a.scale, t[3] * a.scale);
      *reinterpret_cast<uint2*>(p) = pk;
    };
    // this wavefront's share of the small pieces of a stage: -lse | -delta (wavefronts 0, 1), the mask operand's halves (2, 3)
    auto dma_small = [&](const int2& m, unsigned stage_b) __attribute__((always_inline)) {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      if (wave < 2) {
        const float* rb = rowc_g + (int64_t)M_QT(m) * 128 + wave * 64;
        B1_DMA4L(rb, (unsigned)ln * 4u, stage_b + 16384u + (unsigned)wave * 256u);
      } else {
        const int r = (wave - 2) * 32 + (ln >> 1), nrows = M_ROWS(m);
        const int row = M_ROW0(m) + (r < nrows ? r : nrows - 1);
        B1_DMA16L(a.qblk, (unsigned)row * 32u + (unsigned)(ln & 1) * 16u, stage_b + 16896u + (unsigned)(wave - 2) * 1024u, 0);
      }
    };
    unsigned st_c = lds_b, st_n = lds_b + STAGE_B, st_nn = lds_b + 2u * STAGE_B;          // stages of steps it, it + 1, it + 2
    unsigned a_meta = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const int2*)meta_b + 16u;          // record it + 2
    // ONE loop body (every iteration, the first and the last included, runs it whole: the first on a neutral "previous step" -
    // scores of -inf, packed operands and fragments of zero: P = dS = 0 -, the last on the block's NULL step, whose row constants are
    // -inf.  Six specialised copies of a 500-register body made hipcc spill accumulators right behind the MFMAs that wrote them.)
    // LIVE: this wavefront has a valid key; PAR = it & 1.
    auto iter = [&]<bool LIVE, int PAR>(const int it) __attribute__((always_inline)) {
      constexpr bool PREV = true, CUR = true;
      constexpr bool STEADY = LIVE;          // (a wavefront without a valid key runs its few MFMAs padded)
      // byte addresses of this lane (carried across the iterations: advanced by wave-uniform deltas at the end of each, so that no
      // loop-invariant offset registers stay live): a_rf / a_tr / a_rc / a_qb point into the stage of step `it`, a_db into the dS^T
      // image of step it - 1, ds_cur is the image of step `it`; st_c / st_n / st_nn: the stages of steps it, it + 1, it + 2
      constexpr unsigned DS_HALF = 2u * TKB * TQ;
      const unsigned ds_cur = DSIMG_B + (PAR ? DS_HALF : 0u), ds_prev = DSIMG_B + (PAR ? 0u : DS_HALF);
      // (opaque to the optimiser: loop strength reduction made every (base + immediate) pair an induction variable of its own -
      //  45 registers, spilled - instead of one base register per fragment and an immediate in the instruction)
      asm volatile("" : "+v"(a_rf[0]), "+v"(a_rf[1]), "+v"(a_rf[2]), "+v"(a_rf[3]), "+v"(a_rc), "+v"(a_qb));
      asm volatile("" : "+v"(a_tr[0][0]), "+v"(a_tr[0][1]), "+v"(a_tr[1][0]), "+v"(a_tr[1][1]), "+v"(a_da[0]), "+v"(a_da[1]), "+v"(a_db[0]), "+v"(a_db[1]));
      unsigned n_off = st_n - st_c;          // (wave-uniform: the next stage's same addresses)
      asm volatile("" : "+s"(n_off));
      asm volatile("" : "+v"(a_dsw[0][0]), "+v"(a_dsw[0][1]), "+v"(a_dsw[1][0]), "+v"(a_dsw[1][1]));          // (opaque: the xor-ed variants are recomputed, not hoisted and spilled)
      const u32x2v m2v = *LDS_P(const u32x2v, a_meta);          // record of step it + 2 (read here, ahead of the barrier's lgkmcnt(0); made scalar behind it)
      const bool prev_first = M_FIRST(mp) != 0;
      float* st_base = acc_s + ((int64_t)M_QT(mpp) << 12);          // (wave-uniform: scalar registers)
      const float* ld_base = acc_s + ((int64_t)M_QT(m0) << 12);
      unsigned ln16 = (unsigned)lane << 4;
      asm volatile("" : "+v"(ln16));
#ifdef MCA_TRACE_BUILD
      unsigned long long tr_t[7];
#endif
      int2 m2s = m1;
      unsigned o_dma = 0;
      const u16 *qrow = qbase, *orow = obase;
      asm volatile("s_nop 1" ::: "memory");          // register copies of the loop's back edge before the first MFMA reads them
      B1_SB();

#define A_M(J) if (!(B1_ABL & 32)) MF_ACC_VV(S[(J) & 1], qbf, khf[(J) & 1])
#define A_S(J, KS) if (!(B1_ABL & 32)) MF_ACC_AA(S[(J) & 1], rfQ[KS], kf[(J) & 1][KS])
#define A_P(J, KS) if (!(B1_ABL & 32)) MF_ACC_AV(dP[(J) & 1], rfO[KS], vf[(J) & 1][KS])
#define VE2_(SET, I) do { if (B1_ABL & 1) break; const float p0_ = __builtin_amdgcn_exp2f(S[SET][2 * (I)]), p1_ = __builtin_amdgcn_exp2f(S[SET][2 * (I) + 1]); \
                          S[SET][2 * (I)] = p0_; S[SET][2 * (I) + 1] = p1_; dP[SET][2 * (I)] *= p0_; dP[SET][2 * (I) + 1] *= p1_; } while (0)
#define VC_(SET, I) do { if (B1_ABL & 1) break; pb[SET][(I) >> 2][(I) & 3] = pack2bf_pk(S[SET][2 * (I)], S[SET][2 * (I) + 1]); \
                         sb[SET][(I) >> 2][(I) & 3] = pack2bf_pk(dP[SET][2 * (I)], dP[SET][2 * (I) + 1]); \
                         asm volatile("" :: "v"(pb[SET][(I) >> 2][(I) & 3]), "v"(sb[SET][(I) >> 2][(I) & 3])); } while (0)
#define VE2(J, I) VE2_((J) & 1, I)
#define VC(J, I) VC_((J) & 1, I)
#define VEP2(I) VE2_(1, I)
#define VCP(I) VC_(1, I)
#define C_V(J, SP, N) if (!(B1_ABL & 64)) MF_ACCA_VV(dv[(J) & 1][N], trO[SP][N], pb[(J) & 1][SP])
#define C_K(J, SP, N) if (!(B1_ABL & 64)) MF_ACCA_VV(dk[(J) & 1][N], trQ[SP][N], sb[(J) & 1][SP])
#define CP_V(J, SP, N) if (!(B1_ABL & 64)) MF_ACCA_VV(dv[(J) & 1][N], trO[SP][N], pb[(J) & 1][SP])
#define CP_K(J, SP, N) if (!(B1_ABL & 64)) MF_ACCA_VV(dk[(J) & 1][N], trQ[SP][N], sb[(J) & 1][SP])
#define DSW_(IMG, QB, KB, SET, SP, T) if (!(B1_ABL & 16)) *LDS_P(u32x2v, a_dsw[IMG][KB] ^ (unsigned)(((QB) * 4 + 2 * (SP) + (T)) << 4)) = u32x2v{sb[SET][SP][2 * (T)], sb[SET][SP][2 * (T) + 1]}
#define DSW(J, SP, T) DSW_(PAR, (J) >> 1, (J) & 1, (J) & 1, SP, T)
#define DSWP(SP, T) DSW_(PAR ^ 1, 1, 1, 1, SP, T)
#define RC4_(DST, ADDR, G) do { if (B1_ABL & 8) break; const f32x4 t_ = LDS_RF4(ADDR); DST[4 * (G)] = t_[0]; DST[4 * (G) + 1] = t_[1]; DST[4 * (G) + 2] = t_[2]; DST[4 * (G) + 3] = t_[3]; } while (0)
#define RC_L(J, G) RC4_(S[(J) & 1], a_rc + (unsigned)(((J) >> 1) * 128 + 32 * (G)), G)
#define RC_D(J, G) RC4_(dP[(J) & 1], a_rc + (unsigned)(256 + ((J) >> 1) * 128 + 32 * (G)), G)
#define NRC_L(G) RC4_(S[0], a_rc + n_off + (unsigned)(32 * (G)), G)
#define NRC_D(G) RC4_(dP[0], a_rc + n_off + (unsigned)(256 + 32 * (G)), G)
#define RQB(QB) if (!(B1_ABL & 8)) qbf = LDS_R8(a_qb + (unsigned)((QB) * 1024))
#define NRQB() if (!(B1_ABL & 8)) qbf = LDS_R8(a_qb + n_off)
#define RF_Q(QB, KS) if (!(B1_ABL & 8)) rfQ[KS] = LDS_R8(a_rf[KS] + (unsigned)((QB) * 4096))
#define RF_O(QB, KS) if (!(B1_ABL & 8)) rfO[KS] = LDS_R8(a_rf[KS] + (unsigned)(8192 + (QB) * 4096))
#define NRF_Q(KS) if (!(B1_ABL & 8)) rfQ[KS] = LDS_R8(a_rf[KS] + n_off)
#define NRF_O(KS) if (!(B1_ABL & 8)) rfO[KS] = LDS_R8(a_rf[KS] + n_off + 8192u)
#define TR2_(DST, A0, A1, IMM) do { const bf16x4 lo_ = LDS_TR((A0) + (unsigned)(IMM)), hi_ = LDS_TR((A1) + (unsigned)(IMM)); DST = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7); } while (0)
#define TR_O(QB, SP, N) if (!(B1_ABL & 4)) TR2_(trO[SP][N], a_tr[0][N], a_tr[1][N], 8192 + (QB) * 4096 + (SP) * 2048)
#define TR_Q(QB, SP, N) if (!(B1_ABL & 4)) TR2_(trQ[SP][N], a_tr[0][N], a_tr[1][N], (QB) * 4096 + (SP) * 2048)
#define DQR(K) do { if (B1_ABL & 2) break; TR2_(da[(K) & 1], a_da[0], a_da[1], (K) * 2048); TR2_(db[(K) & 1], a_db[0], a_db[1], (K) * 2048); } while (0)
      // (the first product of a tile's first visit starts from the constant 0; behind the counted wait in either case: an unused load
      //  that lands later would overwrite the product)
#define DQM(K) do { if (B1_ABL & 2) break;                                                                                   \
                    if ((K) == 0) { asm volatile("s_waitcnt vmcnt(" B1_STR(B1_W2_YOUNGER) ")" ::: "memory");                  \
                                    if (prev_first) b1_acc_mfma<PAR ^ 1, true, true>(da[0], db[0]);                          \
                                    else b1_acc_mfma<PAR ^ 1, false, true>(da[0], db[0]); }                                  \
                    else b1_acc_mfma<PAR ^ 1, false, !STEADY>(da[(K) & 1], db[(K) & 1]); } while (0)
      // the barrier: this wavefront's dS^T stores are done (lgkmcnt) and its pieces of the NEXT step's stage have landed
#define BARRIER() do { asm volatile("s_waitcnt vmcnt(" B1_STR(B1_W1_YOUNGER) ") lgkmcnt(0)" ::: "memory");                    \
                       __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
      // piece P of step it + 2's stage: 0 / 2 the two Q pieces of this wavefront, 1 / 3 the dO pieces, 4 its share of the row
      // constants / mask operand
#define DMA(P) do { if ((B1_ABL & 128) && (P) != 0) break;                                                                    \
                    if ((P) == 0) { m2s = make_int2(__builtin_amdgcn_readfirstlane((int)m2v[0]), __builtin_amdgcn_readfirstlane((int)m2v[1])); \
                                    qrow = qbase + (int64_t)M_ROW0(m2s) * DH; orow = obase + (int64_t)M_ROW0(m2s) * DH;            \
                                    int ln_ = lane; asm volatile("" : "+v"(ln_));                                                  \
                                    const int r_ = wave * 16 + (ln_ >> 3);                                                         \
                                    o_dma = (unsigned)(r_ * 128 + (((ln_ & 7) ^ b1_swz(r_)) << 4)); }                              \
                    if (B1_ABL & 128) break;                                                                                        \
                    if ((P) == 0) B1_DMA16L(qrow, o_dma, st_nn + (unsigned)(wave * 2048), 0);                                       \
                    else if ((P) == 1) B1_DMA16L(orow, o_dma, st_nn + 8192u + (unsigned)(wave * 2048), 0);                          \
                    else if ((P) == 2) { o_dma ^= 64u; B1_DMA16L(qrow, o_dma, st_nn + (unsigned)(wave * 2048), 1024); }     \
                    else if ((P) == 3) B1_DMA16L(orow, o_dma, st_nn + 8192u + (unsigned)(wave * 2048), 1024);               \
                    else dma_small(m2s, st_nn); } while (0)
      // the dQ block of step it - 2 (buffer PAR: its last MFMA ran six slots before the previous iteration ended), then the old
      // partial of step it's tile into the same registers
#define ST(G) do { if (B1_ABL & 128) break; if (M_LAST(mpp)) store_dq_last.template operator()<PAR, G>(mpp); else b1_acc_store1<PAR, G>(ln16, st_base); } while (0)
#define LD(G) do { if (B1_ABL & 128) break; b1_acc_load1<PAR, G>(ln16, ld_base); } while (0)
#include "attention_bwd1_sched.inc"
#undef A_M
#undef A_S
#undef A_P
#undef VE2_
#undef VC_
#undef VE2
#undef VC
#undef VEP2
#undef VCP
#undef C_V
#undef C_K
#undef CP_V
#undef CP_K
#undef DSW_
#undef DSW
#undef DSWP
#undef RC4_
#undef RC_L
#undef RC_D
#undef NRC_L
#undef NRC_D
#undef RQB
#undef NRQB
#undef RF_Q
#undef RF_O
#undef NRF_Q
#undef NRF_O
#undef TR2_
#undef TR_O
#undef TR_Q
#undef DQR
#undef DQM
#undef BARRIER
#undef DMA
#undef ST
#undef LD
      {          // addresses of the next iteration
        const unsigned d_st = n_off, d_ds = ds_cur - ds_prev;          // (wave-uniform; unsigned wrap-around is the subtraction)
#pragma unroll
        for (int ks = 0; ks < 4; ks++) a_rf[ks] += d_st;
#pragma unroll
        for (int t = 0; t < 2; t++) { a_tr[t][0] += d_st; a_tr[t][1] += d_st; a_db[t] += d_ds; }
        a_rc += d_st; a_qb += d_st;
      }
      // the records move on: step it + 2's becomes step it + 1's, ...
      B1_TR_END();
      mpp = mp; mp = m0; m0 = m1; m1 = m2s;
      { const unsigned t_ = st_c; st_c = st_n; st_n = st_nn; st_nn = t_; }
      a_meta += 8u;
      B1_SB();
    };

    if (!wave_dead) {
      // the first step's first block: row constants (into score set 0), mask operand, row fragments - the loop reads them an
      // iteration ahead; and the neutral previous step: block 3' scores of -inf (P = 0), packed operands of block 2' and every
      // transposed fragment zero (0 x finite: the products of iteration 0 that belong to no step add exact zeros)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const f32x4 t0 = LDS_RF4(lds_b + 16384u + o_rc + 32u * g), t1 = LDS_RF4(lds_b + 16384u + 256u + o_rc + 32u * g);
#pragma unroll
        for (int e = 0; e < 4; e++) { S[0][4 * g + e] = t0[e]; dP[0][4 * g + e] = t1[e]; }
      }
      qbf = LDS_R8(lds_b + 16896u + o_qb);
#pragma unroll
      for (int ks = 0; ks < 4; ks++) { rfQ[ks] = LDS_R8(lds_b + o_rf[ks]); rfO[ks] = LDS_R8(lds_b + 8192u + o_rf[ks]); }
#pragma unroll
      for (int r = 0; r < 16; r++) { S[1][r] = -INFINITY; dP[1][r] = 0.f; }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int w = 0; w < 4; w++) { pb[0][i][w] = 0u; sb[0][i][w] = 0u; pb[1][i][w] = 0u; sb[1][i][w] = 0u; }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
          for (int e = 0; e < 8; e++) { trO[i][j][e] = 0; trQ[i][j][e] = 0; }
    }
    B1_STAMP();          // 3: loop start
    // iterations 0 .. n_it - 1 (n_it even), parity of `it` as a template argument
    if (!wave_dead) {
      for (int it = 0; it < n_it; it += 2) { iter.template operator()<true, 0>(it); iter.template operator()<true, 1>(it + 1); }
    } else {
      for (int it = 0; it < n_it; it += 2) { iter.template operator()<false, 0>(it); iter.template operator()<false, 1>(it + 1); }
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");          // the last dK / dV MFMAs before the epilogue reads the accumulators
    B1_SB();
    B1_STAMP();          // 4: loop end
    // an odd number of steps: the last one's dQ block (buffer 0, finished by the null iteration) has no iteration left to go out in
    if (n_ent & 1) {
      float* slot = acc_s + ((int64_t)M_QT(mpp) << 12);
      unsigned ln16 = (unsigned)lane << 4;
      asm volatile("" : "+v"(ln16));
      if (M_LAST(mpp)) {
        store_dq_last.template operator()<0, 0>(mpp); store_dq_last.template operator()<0, 1>(mpp);
        store_dq_last.template operator()<0, 2>(mpp); store_dq_last.template operator()<0, 3>(mpp);
      } else {
        b1_acc_store1<0, 0>(ln16, slot); b1_acc_store1<0, 1>(ln16, slot); b1_acc_store1<0, 2>(ln16, slot); b1_acc_store1<0, 3>(ln16, slot);
      }
    }

    // every wavefront is done with the K image, the stages and the dS^T images: the next block's memory traffic starts here
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (this block's dQ stores stay in flight)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    B1_STAMP();          // 5: behind the barrier that ends the block's loop
    const int nxt = __builtin_amdgcn_readfirstlane(next_live(kbi + 1));
    if (nxt >= 0) issue_block(nxt);
    B1_STAMP();          // 6: the next block's memory traffic is requested

    // ---- the block's dK = ln 2 * dK^T, dV = dV^T + dvmean, through this wavefront's own rows of the first dS^T image: a lane holds 8
    // bytes of ITS key's row per register group - stored from the registers that was 64 cache lines per instruction (the epilogue
    // took ~11,000 cycles per block); transposed through LDS (144-byte rows) a store instruction writes eight whole 128-byte rows.
    // (a row slot without a key stores to the null tile's slot of dq_acc, which nobody reads: every lane issues every store)
    {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const unsigned stg = DSIMG_B + (unsigned)wave * 8192u;          // (wave-local: no barrier, the LDS counter orders write -> read)
      const unsigned w_off = stg + (unsigned)((ln & 31) * 144 + 8 * (ln >> 5)), r_off = stg + (unsigned)((ln >> 3) * 144 + (ln & 7) * 16);
      u16* dummy = reinterpret_cast<u16*>(acc_s + ((int64_t)a.n_qtiles << 12)) + (ln & 31) * 64;          // (inside this wavefront's 4 KiB of the slot)
#pragma unroll
      for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int m = 0; m < 2; m++) {          // m = 0: dK, 1: dV
#pragma unroll
          for (int n = 0; n < 2; n++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
              u32x2v pk;
              if (m == 0) {
                pk[0] = pack2bf(dk[kb][n][4 * g] * dk_scale, dk[kb][n][4 * g + 1] * dk_scale);
                pk[1] = pack2bf(dk[kb][n][4 * g + 2] * dk_scale, dk[kb][n][4 * g + 3] * dk_scale);
              } else {
                const f32x4 dvm = *reinterpret_cast<const f32x4*>(dvm_s + n * 32 + 8 * g + 4 * lh);
                pk[0] = pack2bf(dv[kb][n][4 * g] + dvm[0], dv[kb][n][4 * g + 1] + dvm[1]);
                pk[1] = pack2bf(dv[kb][n][4 * g + 2] + dvm[2], dv[kb][n][4 * g + 3] + dvm[3]);
              }
              *LDS_P(u32x2v, w_off + (unsigned)(n * 64 + 16 * g)) = pk;
            }
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const u32x4v v4 = *LDS_P(const u32x4v, r_off + (unsigned)(i * 8 * 144));
            const int slot = wave * 64 + kb * 32 + i * 8 + (ln >> 3);
            u16* base = (m == 0 ? a.dk : a.dv) + (int64_t)b * a.dkv_bstride + (int64_t)(ks_cur + slot) * a.dkv_ld + h * DH + (ln & 7) * 8;
            *reinterpret_cast<u32x4v*>(slot < nk_cur ? base : dummy) = v4;
          }
        }
    }
    B1_STAMP();          // 7: epilogue stores issued
    B1_NOTE(n_it);
#ifdef MCA_TRACE_BUILD
    tr_blk++;
#endif
    kbi = nxt;
  }
#ifdef MCA_TRACE_BUILD
  if (trace_on && lane == 0) mca_trace_attn_bwd1[1023] = (unsigned long long)tr_n;
#endif
  for (int kbi = 0; kbi < a.n_kblocks; kbi++) {          // (this workgroup's) key blocks without a valid key in this sample: dK = 0, dV = dvmean
    if (live_s[kbi] != 2) continue;
    const int4 kd = reinterpret_cast<const int4*>(a.kb_desc)[kbi];
    for (int i = tid; i < kd.y * 8; i += 256) {
      const int key = kd.x + (i >> 3), c = i & 7;
      *reinterpret_cast<uint4*>(a.dk + (int64_t)b * a.dkv_bstride + (int64_t)key * a.dkv_ld + h * DH + c * 8) = make_uint4(0, 0, 0, 0);
      uint4 pv;
      pv.x = pack2bf(dvm_s[c * 8], dvm_s[c * 8 + 1]); pv.y = pack2bf(dvm_s[c * 8 + 2], dvm_s[c * 8 + 3]);
      pv.z = pack2bf(dvm_s[c * 8 + 4], dvm_s[c * 8 + 5]); pv.w = pack2bf(dvm_s[c * 8 + 6], dvm_s[c * 8 + 7]);
      *reinterpret_cast<uint4*>(a.dv + (int64_t)b * a.dkv_bstride + (int64_t)key * a.dkv_ld + h * DH + c * 8) = pv;
    }
  }
}
// split mode: dq = scale * (sum over the S slices, in slice order) of a tile's fp32 partials, as bf16.  One workgroup per (sample,
// head, query tile); thread (wave w, lane l) holds, per piece g, the four dims (w & 1) * 32 + 8 g + 4 (l >> 5) + 0..3 of row
// (w >> 1) * 32 + (l & 31) - the accumulator layout the main kernel stores
__global__ __launch_bounds__(256) void attn_bwd1_reduce_kernel(mca_attn_bwd1_args a) {
  const int qt = blockIdx.x, bh = blockIdx.y, h = bh % a.heads, b = bh / a.heads, S = a.split;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int2 qd = reinterpret_cast<const int2*>(a.qt_desc)[qt];
  const int r = (wave >> 1) * 32 + (lane & 31);
#pragma unroll
  for (int g = 0; g < 4; g++) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < S; s++) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(a.dq_acc + (((int64_t)bh * S + s) * (a.n_qtiles + 1) + qt) * (TQ * DH) + wave * 1024 + g * 256 + lane * 4);
      acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
    if (r < qd.y) {
      uint2 pk;
      pk.x = pack2bf(acc[0] * a.scale, acc[1] * a.scale); pk.y = pack2bf(acc[2] * a.scale, acc[3] * a.scale);
      *reinterpret_cast<uint2*>(a.dq + (int64_t)b * a.dq_bstride + (int64_t)(qd.x + r) * a.dq_ld + h * DH + (wave & 1) * 32 + 8 * g + 4 * (lane >> 5)) = pk;
    }
  }
}
__global__ __launch_bounds__(256, 1) void attn_bwd1p_kernel(mca_attn_bwd1_args a, int dbg) {
  extern __shared__ __attribute__((aligned(16))) u16 lds_dyn[];
  attn_bwd1p_body(a, dbg, lds_dyn);
}
#define B1P_LDS_BYTES ((B1P_NST * B1_STAGE_U16 + TKB * DH + 2 * TKB * TQ) * 2 + B1_MAX_META * 8 + B1_MAX_KB * 16 + B1_MAX_KB + DH * 4 + 16)

#define B1_LDS_BYTES ((B1_NST * B1_STAGE_U16 + TKB * DH + 2 * TKB * TQ) * 2 + B1_MAX_LIST * 4 + 2 * B1_MAX_QT + B1_MAX_KB + DH * 4)

extern "C" int mca_attn_bwd_onepass(const mca_attn_bwd1_args* a, mca_stream_t stream) {
  if (!a || !a->q || !a->k || !a->v || !a->d_o || !a->rowc || !a->dvmean || !a->dq || !a->dk || !a->dv || !a->dq_acc) return MCA_E_BADARG;
  if (!a->keyinfo || !a->ktile_flags || !a->khot || !a->qblk || !a->qt_desc || !a->kb_desc || !a->kb_qt || !a->visit) return MCA_E_BADARG;
  if (a->batch <= 0 || a->heads <= 0 || a->n <= 0 || a->n_qtiles <= 0 || a->n_kblocks <= 0) return MCA_E_BADARG;
  if (a->n_qtiles >= B1_MAX_QT || a->n_kblocks > B1_MAX_KB || a->max_list + 6 > B1_MAX_LIST ||
      a->n_entries + 4 * a->n_kblocks > B1_MAX_META) return MCA_E_UNSUPPORTED;          // (lists + null steps + the records read ahead)
  if (a->q_ld % 8 || a->kv_ld % 8 || a->o_ld % 8 || a->q_bstride % 8 || a->kv_bstride % 8 || a->o_bstride % 8 || a->q_hstride % 8 || a->o_hstride % 8) return MCA_E_ALIGN;
  if ((uintptr_t)a->q % 16 || (uintptr_t)a->k % 16 || (uintptr_t)a->v % 16 || (uintptr_t)a->d_o % 16 || (uintptr_t)a->khot % 16 || (uintptr_t)a->qblk % 16) return MCA_E_ALIGN;
  if (a->dq_ld % 8 || a->dq_bstride % 8 || (uintptr_t)a->dq % 16 || a->dkv_ld % 8 || a->dkv_bstride % 8 || (uintptr_t)a->dk % 16 || (uintptr_t)a->dv % 16) return MCA_E_ALIGN;
  if ((uintptr_t)a->dq_acc % 16 || (uintptr_t)a->rowc % 4 || (uintptr_t)a->kb_desc % 16 || (uintptr_t)a->qt_desc % 8) return MCA_E_ALIGN;
  if (a->nk_pad < a->n || a->n_ktiles64 != (a->n + 63) / 64) return MCA_E_BADARG;
  if (!(a->flags & MCA_ATTN_Q_PRESCALED)) return MCA_E_UNSUPPORTED;
  if ((int64_t)a->batch * a->heads > 0x7fffffff) return MCA_E_UNSUPPORTED;
  static bool attr_set[64] = {false};
  bool* done = mca_dev_flag(attr_set);
  if (!*done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, B1_LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd1p_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, B1P_LDS_BYTES) != hipSuccess)
      return MCA_E_LAUNCH;
    *done = true;
  }
  // knob 9 bit 64: the plain (compiler-scheduled) form of the same algorithm instead of the pipelined one (A/B and cross-check)
  // the pipelined kernel reads q / dO from the head-major packed copies only (64-element rows: a tile is 8 KiB contiguous)
  const bool packed = a->q_ld == DH && a->o_ld == DH && a->q_hstride && a->o_hstride;
  const int split = a->split > 1 ? a->split : 1;
  if (split > 8) return MCA_E_UNSUPPORTED;
  if (split > 1 && !packed) return MCA_E_UNSUPPORTED;          // (split mode exists in the pipelined kernel only)
  if ((mca_knobs[9] & 64 && split == 1) || !packed) hipLaunchKernelGGL(attn_bwd1_kernel, dim3(a->batch * a->heads), dim3(256), B1_LDS_BYTES, as_stream(stream), *a, mca_knobs[9]);
  else hipLaunchKernelGGL(attn_bwd1p_kernel, dim3(a->batch * a->heads * split), dim3(256), B1P_LDS_BYTES, as_stream(stream), *a, mca_knobs[9]);
  if (split > 1) hipLaunchKernelGGL(attn_bwd1_reduce_kernel, dim3(a->n_qtiles, a->batch * a->heads), dim3(256), 0, as_stream(stream), *a);
  return launch_status();
}
