// Block-masked fused attention, backward in TWO passes without atomics (autograd of model.py:73-105).
//
// Why two passes: a one-pass kernel (rounds 1-2, removed in round 3) owns a 256-key block and has to SUM dQ across key blocks.  On
// MI355X float atomics execute at the memory side at ~1.3 TB/s chip-wide (MI355X_MICROARCH.md, "Global float atomics"): the
// 0.9 GB of fp32 dQ adds per layer are a 690 us floor under a 1,000 us kernel, on top of a 166 MB memset and a 250 MB
// fp32 -> bf16 pass per layer.  Here every output element has exactly one owner:
//   mca_attn_bwd_dq   one workgroup = a 128-row QUERY tile of one (sample, head), sweeping the key tiles the structure allows
//                     (the forward's schedule): recomputes P^T and dP^T with the query on the lane, forms dS^T in registers
//                     and accumulates dQ^T += K^T . dS^T (dS^T accumulators ARE the B operand).  dQ leaves once, as bf16
//                     (or fp32), no LDS round trip of dS, no atomics, no memset, no conversion pass.
//   mca_attn_bwd_dkv  one workgroup = a 256-key block (8 wavefronts x 32 keys), sweeping 64-query steps: the one-pass kernel
//                     without its dS^T image, K image, dQ product and atomics.
// The price is S and dP computed twice (7 MFMA products instead of 5); both passes are bitwise reproducible.
#include "common.h"

#define DH 64
#define AQ 128      // dq pass: query rows per workgroup
#define AK 64       // dq pass: keys per tile
#define MAX_KTILES 512
// dkv pass: W wavefronts per workgroup, 32 keys each: W = 8 (256 keys, one workgroup per CU) or W = 4 (128 keys, two
// independent workgroups per CU, free to drift apart: 3 % faster at N = 2538, equal at N = 6088); chosen by the caller through
// mca_attn_bwd2_args.kblock_keys
#define BQ 64       // dkv pass: query rows per step
#define MAX_QTILES 2048
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

// [rows][64 d] bf16 image (128-byte rows) serving BOTH ds_read_b128 row reads and ds_read_b64_tr_b16 transposed reads
__device__ __forceinline__ int rt_off(int r, int c) {
  const int s = ((r >> 1) & 7) ^ (((r >> 1) & 1) << 2);
  return r * 64 + ((c ^ s) << 3);
}
// row reads only: chunk c (16 B) of row r at c ^ ((r>>1)&7)
__device__ __forceinline__ int row_off(int r, int c) { return r * 64 + ((c ^ ((r >> 1) & 7)) << 3); }

// =====================================================================================================
// dQ pass
// =====================================================================================================
template <bool OUT_F32, bool HOT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(mca_attn_bwd2_args a, int dbg) {
  constexpr bool PRESCALED = true;          // q arrives pre-scaled by scale * log2 e (MCA_ATTN_Q_PRESCALED, required)
  __shared__ __attribute__((aligned(16))) u16 lds[2 * 2 * AK * DH];   // K,V double-buffered: 32 KiB
  __shared__ __attribute__((aligned(16))) uint8_t kinfo[2][AK];
  __shared__ __attribute__((aligned(16))) u16 hot_s[2][AK * 16];   // one-hot key groups of the tile (mask product operand)
  __shared__ uint8_t flags_s[MAX_KTILES];
  __shared__ uint32_t live_s[MAX_KTILES + 2];
  __shared__ int n_live_s;
  u16* Ks = lds;
  u16* Vs = lds + 2 * AK * DH;

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

  // B operands held for the whole kernel: lane = query, 8 consecutive d per k-step
  bf16x8 qf[4], dof[4];
  {
    const u16* qp = a.q + (int64_t)b * a.q_bstride + (int64_t)qrow * a.q_ld + h * DH + 8 * lh;
    const u16* op = a.d_o + (int64_t)b * a.o_bstride + (int64_t)qrow * a.o_ld + h * DH + 8 * lh;
#pragma unroll
    for (int s = 0; s < 4; s++) { qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s); dof[s] = *reinterpret_cast<const bf16x8*>(op + 16 * s); }
  }
  const uint32_t qm = a.qmask[qrow];
  // the mask as a matrix product (mca_hip.h, mca_build_keyhot): this lane's query-side operand, groups 8 lh .. 8 lh + 7
  // (rebuilt from qm in every masked tile: four registers held for the whole kernel cost nine spills)
  constexpr bool use_hot = HOT;          // (a compile-time choice: with both mask paths in one kernel the register file spills)
  const uint32_t qm8 = ((qm >> (8 * lh)) & 0xffu) & (lh ? 0x7fu : 0xffu);          // visible-bits of this lane's eight groups; slot 15 never
  const float c2 = a.scale * 1.4426950408889634f;
  // row constants as accumulator start values: S - lse (log2 domain; lse = +inf marks a uniform row: P = 0) and dP - delta
  f32x16 neglse, negdel;
  {
    const int64_t ri = ((int64_t)b * a.heads + h) * a.nq + qrow;
    const float nl = -a.lse[ri], nd = -a.delta[ri];
#pragma unroll
    for (int r = 0; r < 16; r++) { neglse[r] = nl; negdel[r] = nd; }
  }
  f32x16 sinit = neglse;          // un-prescaled q: the scores are scaled AFTER the product, so they start from zero
  if (!PRESCALED) {
#pragma unroll
    for (int r = 0; r < 16; r++) sinit[r] = 0.f;
  }
  f32x16 dq[2];
#pragma unroll
  for (int n = 0; n < 2; n++)
#pragma unroll
    for (int r = 0; r < 16; r++) dq[n][r] = 0.f;

  const u16* kbase = a.k + (int64_t)b * a.kv_bstride + h * DH;
  const u16* vbase = a.v + (int64_t)b * a.kv_bstride + h * DH;
  const uint8_t* kinfo_g = a.keyinfo + (int64_t)b * a.nk_pad;
  const u16* khot_g = use_hot ? a.khot + (int64_t)b * a.nk_pad * 16 : nullptr;
  {
    const uint8_t* flags_g = a.ktile_flags + (int64_t)b * a.n_ktiles64;
    for (int i = tid; i < a.n_ktiles64; i += 256) flags_s[i] = flags_g[i];
  }
  __syncthreads();

  int srow[2], sc[2];
  unsigned loff[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int id = tid + 256 * i; srow[i] = id >> 3; sc[i] = id & 7;
    loff[i] = (unsigned)(srow[i] * (int)a.kv_ld + sc[i] * 8);
  }
  const int last_kt = a.n_ktiles64 - 1;
  bf16x8 rk[2], rv[2];
  const unsigned hoff = (unsigned)tid * 4u;          // u16 elements into the one-hot image | bytes into the key-group bytes
  uint2 rhot = make_uint2(0, 0);          // 8 bytes of the tile's one-hot image per thread (element-mask path: .x = four key-group bytes)
  auto gload = [&](int kt) {
    // (wave-uniform tile base + 32-bit lane offset: a per-lane 64-bit address held across the loop was spilled, and its
    // reload waited vmcnt(0) in front of the prefetch)
    if (use_hot) { const u16* hb = khot_g + (int64_t)kt * (AK * 16); rhot = *reinterpret_cast<const uint2*>(hb + hoff); }
    else if (tid < 16) { const uint8_t* ib = kinfo_g + kt * AK; rhot.x = *reinterpret_cast<const uint32_t*>(ib + hoff); }
    const u16* kb = kbase + (int64_t)kt * AK * a.kv_ld;
    const u16* vb = vbase + (int64_t)kt * AK * a.kv_ld;
    if (kt != last_kt) {
#pragma unroll
      for (int i = 0; i < 2; i++) {
        rk[i] = *reinterpret_cast<const bf16x8*>(kb + loff[i]);
        rv[i] = *reinterpret_cast<const bf16x8*>(vb + loff[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; i++) {
        int key = kt * AK + srow[i]; if (key > a.nk - 1) key = a.nk - 1;
        rk[i] = *reinterpret_cast<const bf16x8*>(kbase + (int64_t)key * a.kv_ld + sc[i] * 8);
        rv[i] = *reinterpret_cast<const bf16x8*>(vbase + (int64_t)key * a.kv_ld + sc[i] * 8);
      }
    }
  };
  auto swrite = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      *reinterpret_cast<bf16x8*>(Ks + buf * AK * DH + rt_off(srow[i], sc[i])) = rk[i];          // rows for S, columns for dQ
      *reinterpret_cast<bf16x8*>(Vs + buf * AK * DH + row_off(srow[i], sc[i])) = rv[i];
    }
    if (use_hot) *reinterpret_cast<uint2*>(&hot_s[buf][tid * 4]) = rhot;
    else if (tid < 16) *reinterpret_cast<uint32_t*>(&kinfo[buf][tid * 4]) = rhot.x;
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
      // bit 31 = no element-wise mask needed (structurally full AND every key of the tile valid in this sample)
      if (keep) live_s[n + __popcll(m & ((1ull << lane) - 1ull))] = (e & 0x7fffffffu) | ((e >> 31) && fl == 2 ? 0x80000000u : 0u);
      n += __popcll(m);
    }
    if (lane == 0) { n_live_s = n; live_s[n] = 0u; live_s[n + 1] = 0u; }
  }
  __syncthreads();
  const int it_end = n_live_s;
  int it = 0;
  int buf = 0;
  uint32_t e_cur = __builtin_amdgcn_readfirstlane(live_s[0]), e_nxt = __builtin_amdgcn_readfirstlane(live_s[1]);
  if (it < it_end) { gload((int)(e_cur & 0x7fffffffu)); swrite(0); }
  __syncthreads();

  // row-fragment addresses (bytes, LDS address space) of this lane's K / V rows: buffer 0, key block 0 (block 1: +4096 bytes,
  // buffer 1: +8192; the swizzles of both images depend on the row modulo 16 only)
  unsigned kaddr[4], vaddr[4];
  {
    const unsigned kb_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const u16*)Ks;
    const unsigned vb_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const u16*)Vs;
#pragma unroll
    for (int st = 0; st < 4; st++) {
      kaddr[st] = kb_lds + 2u * (unsigned)rt_off(l31, 2 * st + lh);
      vaddr[st] = vb_lds + 2u * (unsigned)row_off(l31, 2 * st + lh);
    }
  }
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  while (it < it_end) {
    const int kt = (int)(e_cur & 0x7fffffffu);
    const bool need_mask = (e_cur >> 31) == 0;
    const int nit = it + 1;
    if (nit < it_end) gload((int)(e_nxt & 0x7fffffffu));
    const uint32_t e_nn_v = live_s[it + 2];

    const u16* ks = Ks + buf * AK * DH;
    // S^T - lse and dP^T - delta: sixteen row fragments (K and V, four k-steps, two key blocks) through a rolling window of
    // eight registers sets: a fragment's register is re-requested as soon as its MFMA has issued, every MFMA waits for exactly
    // its own fragment (counted lgkmcnt; the compiler's form was read, wait, MFMA sixteen times in a row)
    f32x16 s[2], dp[2];
    {
      const unsigned bo = (unsigned)buf * (unsigned)(AK * DH * 2);
      u32x4v f[8];
      __builtin_amdgcn_sched_barrier(0);
#define DQ_READ(I, ST, KB, ISV) DQ_DSREAD128(f[I], (ISV ? vaddr[ST] : kaddr[ST]) + bo, KB)
#define DQ_DSREAD128(dst, addr, KB) do { if (KB) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(dst) : "v"(addr)); \
                                         else asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr)); } while (0)
      // fragment order = MFMA order: (st, K kb0), (st, V kb0), (st, K kb1), (st, V kb1)
      DQ_READ(0, 0, 0, 0); DQ_READ(1, 0, 0, 1); DQ_READ(2, 0, 1, 0); DQ_READ(3, 0, 1, 1);
      DQ_READ(4, 1, 0, 0); DQ_READ(5, 1, 0, 1); DQ_READ(6, 1, 1, 0); DQ_READ(7, 1, 1, 1);
#define DQ_WAIT(N) do { asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define DQ_S(I, ST, KB) s[KB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&f[I]), qf[ST], ST == 0 ? sinit : s[KB], 0, 0, 0); __builtin_amdgcn_sched_barrier(0)
#define DQ_P(I, ST, KB) dp[KB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&f[I]), dof[ST], ST == 0 ? negdel : dp[KB], 0, 0, 0); __builtin_amdgcn_sched_barrier(0)
      DQ_WAIT(7); DQ_S(0, 0, 0); DQ_READ(0, 2, 0, 0);
      DQ_WAIT(7); DQ_P(1, 0, 0); DQ_READ(1, 2, 0, 1);
      DQ_WAIT(7); DQ_S(2, 0, 1); DQ_READ(2, 2, 1, 0);
      DQ_WAIT(7); DQ_P(3, 0, 1); DQ_READ(3, 2, 1, 1);
      DQ_WAIT(7); DQ_S(4, 1, 0); DQ_READ(4, 3, 0, 0);
      DQ_WAIT(7); DQ_P(5, 1, 0); DQ_READ(5, 3, 0, 1);
      DQ_WAIT(7); DQ_S(6, 1, 1); DQ_READ(6, 3, 1, 0);
      DQ_WAIT(7); DQ_P(7, 1, 1); DQ_READ(7, 3, 1, 1);
      DQ_WAIT(7); DQ_S(0, 2, 0);
      DQ_WAIT(6); DQ_P(1, 2, 0);
      DQ_WAIT(5); DQ_S(2, 2, 1);
      DQ_WAIT(4); DQ_P(3, 2, 1);
      DQ_WAIT(3); DQ_S(4, 3, 0);
      DQ_WAIT(2); DQ_P(5, 3, 0);
      DQ_WAIT(1); DQ_S(6, 3, 1);
      DQ_WAIT(0); DQ_P(7, 3, 1);
#undef DQ_S
#undef DQ_P
#undef DQ_WAIT
#undef DQ_READ
#undef DQ_DSREAD128
    }
    if (need_mask && use_hot) {          // blocked (query, key) pairs: -32768 on top of the score, exp2 is exactly 0
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
    if (!PRESCALED) {
#pragma unroll
      for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int r = 0; r < 16; r++) s[kb][r] = fmaf(s[kb][r], c2, neglse[r]);
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
    // dS^T = P^T o (dP^T - delta), packed as the B operand of the dQ^T product
    u32x4v dsw[2][2];          // (one packed conversion per operand dword, no re-packing)
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
    // dQ^T[d][q] += K^T[d][key] dS^T[key][q]: A = K^T by transposed reads of the K image (element j of k-step sp carries
    // key 16sp + 8(j>>2) + 4lh + (j&3) of the 32-key block, the accumulator-as-operand order)
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
            const bf16x4 k4 = lds_read_tr16(ks + rt_off(key, d >> 3) + (d & 7));
#pragma unroll
            for (int e = 0; e < 4; e++) ktf[4 * t + e] = k4[e];
          }
          dq[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, *reinterpret_cast<const bf16x8*>(&dsw[kb][sp]), dq[n], 0, 0, 0);
        }

    if (nit < it_end) swrite(buf ^ 1);
    __syncthreads();
    buf ^= 1;
    it = nit;
    e_cur = e_nxt;
    e_nxt = __builtin_amdgcn_readfirstlane(e_nn_v);
  }

  // ---- epilogue: dq = scale * dS K (gradient w.r.t. the unscaled q), one owner per element
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

static int check_bwd2(const mca_attn_bwd2_args* a) {
  if (!a || !a->q || !a->k || !a->v || !a->d_o || !a->lse || !a->delta || !a->qmask || !a->keyinfo || !a->ktile_flags) return MCA_E_BADARG;
  if (a->batch <= 0 || a->heads <= 0 || a->nq <= 0 || a->nk <= 0) return MCA_E_BADARG;
  if (a->q_ld % 8 || a->kv_ld % 8 || a->o_ld % 8 || a->q_bstride % 8 || a->kv_bstride % 8 || a->o_bstride % 8) return MCA_E_ALIGN;
  if ((uintptr_t)a->q % 16 || (uintptr_t)a->k % 16 || (uintptr_t)a->v % 16 || (uintptr_t)a->d_o % 16) return MCA_E_ALIGN;
  if (a->heads > 65535 || a->batch > 65535) return MCA_E_UNSUPPORTED;
  return MCA_OK;
}

extern "C" int mca_attn_bwd_dq(const mca_attn_bwd2_args* a, mca_stream_t stream) {
  const int rc = check_bwd2(a);
  if (rc != MCA_OK) return rc;
  if (!a->dq || !a->q_ptr || !a->q_kt || !a->q_order) return MCA_E_BADARG;
  if (a->n_qtiles128 != (a->nq + AQ - 1) / AQ || a->n_ktiles64 != (a->nk + AK - 1) / AK) return MCA_E_BADARG;
  if (a->nk_pad < a->n_ktiles64 * AK || a->nk_pad % 4 || (uintptr_t)a->keyinfo % 4) return MCA_E_BADARG;
  if (a->dq_ld % 4 || a->dq_bstride % 4 || (uintptr_t)a->dq % (a->dq_f32 ? 16 : 8)) return MCA_E_ALIGN;
  if (a->n_ktiles64 > MAX_KTILES) return MCA_E_UNSUPPORTED;
  const dim3 grid(a->n_qtiles128, a->heads, a->batch);
  if (!(a->flags & MCA_ATTN_Q_PRESCALED)) return MCA_E_UNSUPPORTED;          // (the un-prescaled forms left the library in round 3:
                                                                              //  one of them spilled 579 registers)
  const int dbg = mca_knobs[9];
  const bool hot = a->khot != nullptr;
  if ((uintptr_t)a->khot % 16) return MCA_E_ALIGN;
#define DQ_LAUNCH(F, H) hipLaunchKernelGGL((attn_bwd_dq_kernel<F, H>), grid, dim3(256), 0, as_stream(stream), *a, dbg)
  if (!a->dq_f32) { if (hot) DQ_LAUNCH(false, true); else DQ_LAUNCH(false, false); }
  else { if (hot) DQ_LAUNCH(true, true); else DQ_LAUNCH(true, false); }
#undef DQ_LAUNCH
  return launch_status();
}

// =====================================================================================================
// dK / dV pass: one workgroup = 256 keys (8 wavefronts x 32 keys, K / V fragments and dK^T / dV^T in registers), sweeping the
// 64-query steps the structure allows for the block; the key sits on the MFMA lane, so P and dS are directly the B operands
// of the dV^T / dK^T products (guide, Appendix B "Attention backward").  Q / dO tiles in one LDS image for row AND column reads.
// =====================================================================================================
template <int W>
__global__ __launch_bounds__(64 * W) void attn_bwd_dkv_kernel(mca_attn_bwd2_args a, int dbg) {
  constexpr bool PRESCALED = true;          // (MCA_ATTN_Q_PRESCALED, required)
  constexpr int BKEYS = 32 * W, DKV_NT = 64 * W, DKV_SU = 512 / DKV_NT;
  extern __shared__ __attribute__((aligned(16))) u16 lds[];
  u16* Qs = lds;                               // 2 x 64 x 64
  u16* Os = Qs + 2 * BQ * DH;                  // 2 x 64 x 64   (dO)
  float* rowc = reinterpret_cast<float*>(Os + 2 * BQ * DH);       // [2][3][64]: -lse, -delta, qmask(bits)
  uint32_t* qlist = reinterpret_cast<uint32_t*>(rowc + 2 * 192);  // [MAX_QTILES]
  float* dvm_s = reinterpret_cast<float*>(qlist + MAX_QTILES);    // [64]
  u16* qblk_s = reinterpret_cast<u16*>(dvm_s + DH);               // [2][64][16]: query side of the mask product

  const int lin0 = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
  const int lin = (dbg & 16) ? lin0 : xcd_remap(lin0, (int)(gridDim.x * gridDim.y * gridDim.z));
  const int4 w = reinterpret_cast<const int4*>(a.k_wg)[lin % (int)gridDim.x];
  const int kbi = w.x, it_begin = w.y, first_qt = w.w;
  const int h = (lin / (int)gridDim.x) % (int)gridDim.y, b = lin / (int)(gridDim.x * gridDim.y);
  // a key block whose four 64-key tiles hold no valid key in this sample (ktile_flags == 0: a dropped modality) sweeps nothing:
  // dK = 0, dV = dvmean straight from the epilogue
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
  int keyc = mykey; if (keyc > a.nk - 1) keyc = a.nk - 1;
  // q pre-scaled by scale * log2(e): q.k is the log2-domain logit (no multiply per score), dK^T carries that factor
  const float c2 = PRESCALED ? 1.f : a.scale * 1.4426950408889634f, inv_c2 = 1.f / c2;
  const float dk_scale = PRESCALED ? 0.6931471805599453f : a.scale;

  const u16* kbase = a.k + (int64_t)b * a.kv_bstride + h * DH;
  const u16* vbase = a.v + (int64_t)b * a.kv_bstride + h * DH;
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int s = 0; s < 4; s++) {
    kf[s] = *reinterpret_cast<const bf16x8*>(kbase + (int64_t)keyc * a.kv_ld + 16 * s + 8 * lh);
    vf[s] = *reinterpret_cast<const bf16x8*>(vbase + (int64_t)keyc * a.kv_ld + 16 * s + 8 * lh);
  }
  const uint32_t kinfo = a.keyinfo[(int64_t)b * a.nk_pad + mykey];     // nk_pad covers every key block
  // the mask as a matrix product (mca_hip.h, mca_build_keyhot): this lane's key-side operand (groups 8 lh .. 8 lh + 7 of its key)
  const bool use_hot = a.khot != nullptr && a.qblk != nullptr;
  bf16x8 khf;
#pragma unroll
  for (int j = 0; j < 8; j++) khf[j] = 0;
  if (use_hot) khf = *reinterpret_cast<const bf16x8*>(a.khot + ((int64_t)b * a.nk_pad + mykey) * 16 + 8 * lh);
  const bool key_ok = kinfo != 31u;
  const uint32_t keybit = key_ok ? (1u << kinfo) : 0u;
  const bool wave_keys_ok = __all(key_ok);
  // a wavefront whose 32 keys are all padded (a dropped modality, the tail of a short sequence) has P = 0 everywhere: it
  // keeps staging and meeting the barriers, but skips its products and arithmetic; dK stays 0, dV becomes dvmean
  const bool wave_dead = !__any(key_ok);

  f32x16 dk[2], dv[2];
#pragma unroll
  for (int n = 0; n < 2; n++)
#pragma unroll
    for (int r = 0; r < 16; r++) { dk[n][r] = 0.f; dv[n][r] = 0.f; }

  const u16* qbase = a.q + (int64_t)b * a.q_bstride + h * DH;
  const u16* obase = a.d_o + (int64_t)b * a.o_bstride + h * DH;
  const float* lse_g = a.lse + ((int64_t)b * a.heads + h) * a.nq;
  const float* delta_g = a.delta + ((int64_t)b * a.heads + h) * a.nq;

  const int srow = tid >> 3, sc = tid & 7;          // (+ 64 * DKV_NT / 512 rows per further piece)
  bf16x8 stage_q[DKV_SU], stage_o[DKV_SU];
  uint4 stage_b = make_uint4(0, 0, 0, 0);
  float stage_c = 0.f;
  bool stage_oob = false;
  const int cwhich = tid >> 6 < 2 ? tid >> 6 : 2, crow = tid & 63;
  const float* cbase = cwhich == 0 ? lse_g : (cwhich == 1 ? delta_g : reinterpret_cast<const float*>(a.qmask));
  auto gload = [&](int qt) {
#pragma unroll
    for (int u = 0; u < DKV_SU; u++) {
      int q = qt * BQ + srow + u * (DKV_NT / 8); if (q > a.nq - 1) q = a.nq - 1;
      stage_q[u] = *reinterpret_cast<const bf16x8*>(qbase + (int64_t)q * a.q_ld + sc * 8);
      stage_o[u] = *reinterpret_cast<const bf16x8*>(obase + (int64_t)q * a.o_ld + sc * 8);
    }
    int qq = qt * BQ + crow;
    stage_oob = qq >= a.nq;
    if (qq > a.nq - 1) qq = a.nq - 1;
    stage_c = cbase[qq];
    if (use_hot && tid < 128) {
      int qb = qt * BQ + (tid >> 1); if (qb > a.nq - 1) qb = a.nq - 1;
      stage_b = *reinterpret_cast<const uint4*>(a.qblk + (int64_t)qb * 16 + (tid & 1) * 8);
    }
  };
  auto swrite = [&](int buf) {
#pragma unroll
    for (int u = 0; u < DKV_SU; u++) {
      *reinterpret_cast<bf16x8*>(Qs + buf * BQ * DH + rt_off(srow + u * (DKV_NT / 8), sc)) = stage_q[u];
      *reinterpret_cast<bf16x8*>(Os + buf * BQ * DH + rt_off(srow + u * (DKV_NT / 8), sc)) = stage_o[u];
    }
    if (tid < 192) {          // rows past nq contribute nothing: lse = +inf (P = 0), delta = 0, qmask = 0
      float v = stage_oob ? (cwhich == 0 ? INFINITY : 0.f) : stage_c;
      if (cwhich == 0) v = -v * inv_c2; else if (cwhich == 1) v = -v;
      rowc[buf * 192 + tid] = v;
    }
    if (use_hot && tid < 128) *reinterpret_cast<uint4*>(qblk_s + buf * (BQ * 16) + (tid >> 1) * 16 + (tid & 1) * 8) = stage_b;
  };

  if (n_it > 0) gload(first_qt);
  // the K / V fragment loads are older than the first tile's: naming them as asm inputs here makes hipcc wait for them NOW
  // (vmcnt counted against the first tile's three loads) instead of inside the loop, where its conservative vmcnt would make
  // every step wait for the next tile's loads it has just issued
#pragma unroll
  for (int s4 = 0; s4 < 4; s4++) asm volatile("" :: "v"(kf[s4]), "v"(vf[s4]));
  for (int i = tid; i < n_it; i += DKV_NT) qlist[i] = a.k_qt[it_begin + i];
  if (tid < DH) dvm_s[tid] = a.dvmean[(int64_t)b * a.heads * DH + h * DH + tid];
  int buf = 0;
  if (n_it > 0) swrite(0);
  __syncthreads();

  // row-fragment addresses (bytes, LDS address space) of this lane's Q / dO rows: buffer 0, sub-tile 0
  unsigned qaddr[4], oaddr[4];
  {
    const unsigned q_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const u16*)Qs;
    const unsigned o_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const u16*)Os;
#pragma unroll
    for (int st = 0; st < 4; st++) { qaddr[st] = q_lds + 2u * (unsigned)rt_off(l31, 2 * st + lh); oaddr[st] = o_lds + 2u * (unsigned)rt_off(l31, 2 * st + lh); }
  }
  const unsigned caddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)rowc + 16u * (unsigned)lh;
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  for (int it = 0; it < n_it; it++) {
    const uint32_t ent = __builtin_amdgcn_readfirstlane(qlist[it]);          // wave-uniform: scalar branch on `full`
    const bool full = (ent >> 31) != 0;
    if (it + 1 < n_it) gload((int)(__builtin_amdgcn_readfirstlane(qlist[it + 1]) & 0x7fffffffu));
    if (!wave_dead)
#pragma unroll
    for (int sub = 0; sub < 2; sub++) {
      const u16* qs = Qs + buf * BQ * DH + sub * 32 * DH;
      const u16* os = Os + buf * BQ * DH + sub * 32 * DH;
      const float* rc = rowc + buf * 192 + sub * 32;
      f32x16 s, dp;
      {          // the row constants (accumulator start values) and the eight Q / dO row fragments of the sub-tile are requested
                 // together, all by hand: a compiler-issued read in between would make hipcc wait lgkmcnt(0) at its first use,
                 // i.e. for every fragment; every MFMA waits for exactly its own fragment
        const unsigned so = (unsigned)buf * (unsigned)(BQ * DH * 2) + (unsigned)sub * (unsigned)(32 * DH * 2);
        const unsigned co = caddr + (unsigned)(buf * 192 + sub * 32) * 4u;
        u32x4v f[8], cl[4], cd[4];
        __builtin_amdgcn_sched_barrier(0);
#define KV_CREAD(DST, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(DST) : "v"(co))
        KV_CREAD(cl[0], 0); KV_CREAD(cl[1], 32); KV_CREAD(cl[2], 64); KV_CREAD(cl[3], 96);
        KV_CREAD(cd[0], 256); KV_CREAD(cd[1], 288); KV_CREAD(cd[2], 320); KV_CREAD(cd[3], 352);
#undef KV_CREAD
#define KV_READ(I, ADDR) asm volatile("ds_read_b128 %0, %1" : "=v"(f[I]) : "v"(ADDR + so))
        KV_READ(0, qaddr[0]); KV_READ(1, oaddr[0]); KV_READ(2, qaddr[1]); KV_READ(3, oaddr[1]);
        KV_READ(4, qaddr[2]); KV_READ(5, oaddr[2]); KV_READ(6, qaddr[3]); KV_READ(7, oaddr[3]);
#define KV_WAIT(N) do { asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define KV_S(I, ST) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&f[I]), kf[ST], s, 0, 0, 0)
#define KV_P(I, ST) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&f[I]), vf[ST], dp, 0, 0, 0)
        KV_WAIT(7);          // fragment 0 is back, so are the eight older constant reads
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
          for (int e = 0; e < 4; e++) { s[4 * g + e] = __uint_as_float(cl[g][e]); dp[4 * g + e] = __uint_as_float(cd[g][e]); }
        KV_S(0, 0); KV_WAIT(6); KV_P(1, 0); KV_WAIT(5); KV_S(2, 1); KV_WAIT(4); KV_P(3, 1);
        KV_WAIT(3); KV_S(4, 2); KV_WAIT(2); KV_P(5, 2); KV_WAIT(1); KV_S(6, 3); KV_WAIT(0); KV_P(7, 3);
        __builtin_amdgcn_sched_barrier(0);
#undef KV_READ
#undef KV_WAIT
#undef KV_S
#undef KV_P
      }
      // P and dS as the B operands of the products: the packed conversion's dword IS two adjacent operand elements (splitting
      // it into shorts and re-packing cost a v_perm per dword)
      u32x4v pbw[2], sbw[2];
      const bool clean = full && wave_keys_ok;
      if (!clean && use_hot) {          // blocked (query, key) pairs: -32768 on top of the score, exp2 is exactly 0
        const bf16x8 qbf = *reinterpret_cast<const bf16x8*>(qblk_s + buf * (BQ * 16) + (sub * 32 + l31) * 16 + 8 * lh);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qbf, khf, s, 0, 0, 0);
      }
      if (clean || use_hot) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float p0 = __builtin_amdgcn_exp2f(PRESCALED ? s[r] : s[r] * c2), p1 = __builtin_amdgcn_exp2f(PRESCALED ? s[r + 1] : s[r + 1] * c2);
          const uint32_t pp = pack2bf_pk(p0, p1), ss = pack2bf_pk(p0 * dp[r], p1 * dp[r + 1]);
          pbw[r >> 3][(r & 7) >> 1] = pp;
          sbw[r >> 3][(r & 7) >> 1] = ss;
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const f32x4 qm4 = *reinterpret_cast<const f32x4*>(rc + 128 + 8 * g + 4 * lh);
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            const int r = 4 * g + e;
            const uint32_t q0 = full ? 0xffffffffu : __float_as_uint(qm4[e]), q1 = full ? 0xffffffffu : __float_as_uint(qm4[e + 1]);
            float p0 = __builtin_amdgcn_exp2f(PRESCALED ? s[r] : s[r] * c2), p1 = __builtin_amdgcn_exp2f(PRESCALED ? s[r + 1] : s[r + 1] * c2);
            p0 = (q0 & keybit) ? p0 : 0.f; p1 = (q1 & keybit) ? p1 : 0.f;
            const uint32_t pp = pack2bf_pk(p0, p1), ss = pack2bf_pk(p0 * dp[r], p1 * dp[r + 1]);
            pbw[r >> 3][(r & 7) >> 1] = pp;
            sbw[r >> 3][(r & 7) >> 1] = ss;
          }
        }
      }
      // dV^T += dO^T P ; dK^T += Q^T dS   (element j of k-step sp carries q = 16sp + 8(j>>2) + 4lh + (j&3))
#pragma unroll
      for (int sp = 0; sp < 2; sp++)
#pragma unroll
        for (int n = 0; n < 2; n++) {
          bf16x8 ot, qtf;
#pragma unroll
          for (int t = 0; t < 2; t++) {
            const int qr = 16 * sp + 8 * t + 4 * lh + tq;
            const int d = n * 32 + 16 * tg + 4 * tp;
            const bf16x4 o4 = lds_read_tr16(os + rt_off(qr, d >> 3) + (d & 7));
            const bf16x4 q4 = lds_read_tr16(qs + rt_off(qr, d >> 3) + (d & 7));
#pragma unroll
            for (int e = 0; e < 4; e++) { ot[4 * t + e] = o4[e]; qtf[4 * t + e] = q4[e]; }
          }
          dv[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ot, *reinterpret_cast<const bf16x8*>(&pbw[sp]), dv[n], 0, 0, 0);
          dk[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, *reinterpret_cast<const bf16x8*>(&sbw[sp]), dk[n], 0, 0, 0);
        }
    }
    if (it + 1 < n_it) swrite(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // ---- epilogue: dK = dk_scale * dK^T, dV = dV^T + dvmean (uniform rows spread over every key)
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
#define DKV_LDS_BYTES (2 * BQ * DH * 2 * 2 + 2 * 192 * 4 + MAX_QTILES * 4 + DH * 4 + 2 * BQ * 16 * 2)

extern "C" int mca_attn_bwd_dkv(const mca_attn_bwd2_args* a, mca_stream_t stream) {
  const int rc = check_bwd2(a);
  if (rc != MCA_OK) return rc;
  if (!a->dk || !a->dv || !a->dvmean || !a->k_wg || !a->k_qt) return MCA_E_BADARG;
  const int BKEYS = a->kblock_keys ? a->kblock_keys : 256;
  if (BKEYS != 256 && BKEYS != 128) return MCA_E_UNSUPPORTED;
  if (a->n_qtiles64 != (a->nq + BQ - 1) / BQ || a->n_kblocks256 != (a->nk + BKEYS - 1) / BKEYS) return MCA_E_BADARG;
  if (a->nk_pad < a->n_kblocks256 * BKEYS) return MCA_E_BADARG;
  if (a->dkv_ld % 4 || a->dkv_bstride % 4 || (uintptr_t)a->dk % 8 || (uintptr_t)a->dv % 8 || (uintptr_t)a->k_wg % 16) return MCA_E_ALIGN;
  if (a->n_qtiles64 > MAX_QTILES) return MCA_E_UNSUPPORTED;
  if ((a->khot != nullptr) != (a->qblk != nullptr)) return MCA_E_BADARG;
  if ((uintptr_t)a->khot % 16 || (uintptr_t)a->qblk % 16) return MCA_E_ALIGN;
  if (!(a->flags & MCA_ATTN_Q_PRESCALED)) return MCA_E_UNSUPPORTED;
  int dev = 0;
  static bool attr_set[64] = {false};          // hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MCA_E_LAUNCH;
  if (!attr_set[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, DKV_LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, DKV_LDS_BYTES) != hipSuccess)
      return MCA_E_LAUNCH;
    attr_set[dev] = true;
  }
  const dim3 grid(a->n_kblocks256, a->heads, a->batch);
#define DKV_LAUNCH(W) hipLaunchKernelGGL((attn_bwd_dkv_kernel<W>), grid, dim3(64 * W), DKV_LDS_BYTES, as_stream(stream), *a, mca_knobs[9])
  if (BKEYS == 256) DKV_LAUNCH(8); else DKV_LAUNCH(4);
#undef DKV_LAUNCH
  return launch_status();
}

// =====================================================================================================
// delta[b,h,q] = sum_d dO[q,h,d] * O[q,h,d]   (dvmean: attn_dvmean_block below, same launch)
// one wavefront per (b, q) row: lane covers 8 contiguous columns of the 512-wide row -> head = lane / 8
// =====================================================================================================
#define PREP_ROWS 32
// row_slot != NULL (one-pass backward, attention_bwd1.hip): the row constants go out NEGATED and in that kernel's tile order
// instead: rowc[b, h, tile, 0, pos] = -lse[b, h, q], rowc[b, h, tile, 1, pos] = -delta, row_slot[q] = tile * 64 + pos (n_qtiles + 1
// tiles per (b, h): the last one is the one-pass kernel's null tile, never written here); positions
// past a tile's rows keep what the caller initialised them with (-inf | 0: a row that contributes nothing)
__device__ __forceinline__ void attn_bwd_prep_block(const int bx, const int by, const u16* __restrict__ o, const u16* __restrict__ d_o,
                                                    int64_t bstride, int64_t ld, float* __restrict__ delta, int heads, int nq,
                                                    const float* __restrict__ lse, const int32_t* __restrict__ row_slot,
                                                    float* __restrict__ rowc, int n_qtiles,
                                                    const u16* __restrict__ qsrc, int64_t q_bstride, int64_t q_ld,
                                                    u16* __restrict__ q_hm, u16* __restrict__ do_hm) {
  // delta is (b, head, q): a row touches it at a stride of nq floats per head.  It crosses LDS so that the global accesses are
  // 128-byte runs along q (one row at a time they were 4-byte accesses in 8 different lines per row).
  __shared__ float del_s[8][PREP_ROWS];
  const int b = by, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cols = heads * DH;
  const int q_begin = bx * PREP_ROWS;
  int q_end = q_begin + PREP_ROWS; if (q_end > nq) q_end = nq;
  for (int c0 = 0; c0 < cols; c0 += 512) {
    const int h0 = c0 / DH;
    const int c = c0 + lane * 8;
    const int hl = lane >> 3;          // head of this lane inside the 512-column slab
    // all PREP_ROWS / 4 rows of this wavefront are requested before any is used (a row at a time the loop is latency-bound)
    bf16x8 ovs[PREP_ROWS / 4], dvs[PREP_ROWS / 4], qvs[PREP_ROWS / 4];
#pragma unroll
    for (int k = 0; k < PREP_ROWS / 4; k++) {
      int q = q_begin + wave + 4 * k; if (q > nq - 1) q = nq - 1;
      const int cc = c < cols ? c : 0;
      ovs[k] = *reinterpret_cast<const bf16x8*>(o + (int64_t)b * bstride + (int64_t)q * ld + cc);
      dvs[k] = *reinterpret_cast<const bf16x8*>(d_o + (int64_t)b * bstride + (int64_t)q * ld + cc);
      if (do_hm) qvs[k] = *reinterpret_cast<const bf16x8*>(qsrc + (int64_t)b * q_bstride + (int64_t)q * q_ld + cc);          // (with the others: a row at a time it was one more latency per row)
    }
#pragma unroll
    for (int k = 0; k < PREP_ROWS / 4; k++) {
      const int q = q_begin + wave + 4 * k;
      if (q >= q_end) break;
      // head-major packed copies for the one-pass backward ([b][head][row][64]: a 64-row tile of one head is 8 KiB contiguous; read
      // from rows 3 KiB apart its LDS-DMA pieces were eight 128-byte segments in eight DRAM pages: 200 us per layer at b = 32)
      if (do_hm && c < cols) {
        const int64_t dst = (((int64_t)b * heads + h0 + hl) * nq + q) * DH + (lane & 7) * 8;
        *reinterpret_cast<bf16x8*>(do_hm + dst) = dvs[k];
        *reinterpret_cast<bf16x8*>(q_hm + dst) = qvs[k];
      }
      float part = 0.f;
      if (c < cols) {
#pragma unroll
        for (int j = 0; j < 8; j++) part += bf2f((u16)dvs[k][j]) * bf2f((u16)ovs[k][j]);
      }
      // 8 lanes per head
      part += __shfl_xor(part, 1, WAVE); part += __shfl_xor(part, 2, WAVE); part += __shfl_xor(part, 4, WAVE);
      if ((lane & 7) == 0) del_s[hl][q - q_begin] = part;
    }
    __syncthreads();
    {
      const int hh = tid / PREP_ROWS, r = tid % PREP_ROWS;          // 8 heads x 32 rows = 256 threads
      if (h0 + hh < heads && q_begin + r < q_end) {
        if (!row_slot) delta[((int64_t)b * heads + h0 + hh) * nq + q_begin + r] = del_s[hh][r];
        else {
          const int q = q_begin + r, slot = row_slot[q];
          float* dst = rowc + (((int64_t)b * heads + h0 + hh) * (n_qtiles + 1) + (slot >> 6)) * 128 + (slot & 63);          // (+ 1: the null tile)
          dst[0] = -lse[((int64_t)b * heads + h0 + hh) * nq + q];
          dst[64] = -del_s[hh][r];
        }
      }
    }
    __syncthreads();
  }
}

// dvmean[b, h*64 + d] = (1/nk) * sum over the uniform rows (lse = +inf) of dO, in ROW ORDER: one workgroup per (sample, head)
// lists its uniform rows (usually none: they exist only where a dropped modality empties a query's key set) and adds them in
// a fixed order, so the value - and with it dV of every key of that sample - is bitwise reproducible (per-block atomics,
// rounds 1-2, made dV differ from run to run at the 2e-4 level whenever a modality was dropped).  Also the zeroing of dvmean
// (a kernel, not hipMemsetAsync: as a memset NODE of a single-chain captured step the zeroing was not ordered against the
// kernels around it - the whole backward then started from a dvmean full of stale sums).
#define DVM_MAX_LIST 4096
__device__ __forceinline__ void attn_dvmean_block(const int bx, const int by, const u16* __restrict__ d_o, int64_t bstride, int64_t ld,
                                                  const float* __restrict__ lse, float* __restrict__ dvmean,
                                                  int heads, int nq, float inv_nk) {
  __shared__ int list_s[DVM_MAX_LIST];
  __shared__ int n_s;
  __shared__ float red[32][DH + 1];
  const int h = bx, b = by, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* l = lse + ((int64_t)b * heads + h) * nq;
  float* out = dvmean + (int64_t)b * heads * DH + h * DH;
  {          // the common case first: no uniform row at all (one barrier)
    int anyu = 0;
    for (int q = tid; q < nq; q += 256) anyu |= (l[q] == INFINITY) ? 1 : 0;
    if (tid == 0) n_s = 0;
    if (!__syncthreads_or(anyu)) { if (tid < DH) out[tid] = 0.f; return; }
  }
  // ordered compaction of the uniform rows: wavefront w of round r owns rows 256 r + 64 w .. + 63; the rounds and the
  // wavefronts of a round append in order (a barrier per wavefront turn).  8 lanes per row (8 columns each), 32 row groups
  // striding the list; a full list is added up and emptied (the order stays a function of the data alone).
  const int cl = tid & 7, rg = tid >> 3;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto flush = [&]() {          // every thread, behind a barrier that made the list visible
    const int n = n_s;
    for (int i = rg; i < n; i += 32) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(d_o + (int64_t)b * bstride + (int64_t)list_s[i] * ld + h * DH + cl * 8);
#pragma unroll
      for (int j = 0; j < 8; j++) acc[j] += bf2f((u16)v[j]);
    }
    __syncthreads();
    if (tid == 0) n_s = 0;
    __syncthreads();
  };
  for (int q0 = 0; q0 < nq; q0 += 256) {
    const int q = q0 + tid;
    const bool uni = q < nq && l[q] == INFINITY;
    const unsigned long long m = __ballot(uni);
    for (int w = 0; w < 4; w++) {
      if (wave == w && m) {
        const int base = n_s;
        if (uni) list_s[base + __popcll(m & ((1ull << lane) - 1ull))] = q;
        if (lane == 0) n_s = base + __popcll(m);
      }
      __syncthreads();
    }
    if (n_s > DVM_MAX_LIST - 256) flush();          // (uniform over the workgroup: n_s is read behind the barrier)
  }
  flush();
#pragma unroll
  for (int j = 0; j < 8; j++) red[rg][cl * 8 + j] = acc[j];
  __syncthreads();
  if (tid < DH) {
    float t = 0.f;
    for (int r = 0; r < 32; r++) t += red[r][tid];
    out[tid] = t * inv_nk;
  }
}
// delta and dvmean are independent of each other: ONE launch (a node of the replayed step costs ~4 us whatever it does); blocks
// 0 .. n_prep - 1 of a sample compute delta, the `heads` blocks behind them dvmean
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const u16* __restrict__ o, const u16* __restrict__ d_o, int64_t bstride, int64_t ld,
                                                             const float* __restrict__ lse, float* __restrict__ delta,
                                                             float* __restrict__ dvmean, int heads, int nq, float inv_nk, int n_prep,
                                                             const int32_t* __restrict__ row_slot, float* __restrict__ rowc, int n_qtiles,
                                                             const u16* __restrict__ qsrc, int64_t q_bstride, int64_t q_ld,
                                                             u16* __restrict__ q_hm, u16* __restrict__ do_hm) {
  if ((int)blockIdx.x < n_prep) attn_bwd_prep_block((int)blockIdx.x, (int)blockIdx.y, o, d_o, bstride, ld, delta, heads, nq, lse, row_slot, rowc, n_qtiles,
                                                    qsrc, q_bstride, q_ld, q_hm, do_hm);
  else attn_dvmean_block((int)blockIdx.x - n_prep, (int)blockIdx.y, d_o, bstride, ld, lse, dvmean, heads, nq, inv_nk);
}
extern "C" int mca_attn_bwd_prep(const uint16_t* o, const uint16_t* d_o, int64_t o_bstride, int64_t o_ld,
                                 const float* lse, float* delta, float* dvmean, int batch, int heads, int nq, int nk,
                                 mca_stream_t stream) {
  if (!o || !d_o || !lse || !delta || !dvmean || batch <= 0 || heads <= 0 || nq <= 0 || nk <= 0) return MCA_E_BADARG;
  if (o_ld % 8 || o_bstride % 8 || (uintptr_t)o % 16 || (uintptr_t)d_o % 16) return MCA_E_ALIGN;
  if (heads > 65535 || batch > 65535) return MCA_E_UNSUPPORTED;
  const int n_prep = (nq + PREP_ROWS - 1) / PREP_ROWS;
  hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3(n_prep + heads, batch), dim3(256), 0, as_stream(stream), o, d_o, o_bstride, o_ld, lse, delta,
                     dvmean, heads, nq, 1.f / (float)nk, n_prep, (const int32_t*)nullptr, (float*)nullptr, 0, (const u16*)nullptr, (int64_t)0, (int64_t)0,
                     (u16*)nullptr, (u16*)nullptr);
  return launch_status();
}
extern "C" int mca_attn_bwd_prep_onepass(const uint16_t* o, const uint16_t* d_o, int64_t o_bstride, int64_t o_ld, const float* lse,
                                         const int32_t* row_slot, float* rowc, float* dvmean, int batch, int heads, int n, int n_qtiles,
                                         const uint16_t* q, int64_t q_bstride, int64_t q_ld, uint16_t* q_hm, uint16_t* do_hm,
                                         mca_stream_t stream) {
  if (!o || !d_o || !lse || !row_slot || !rowc || !dvmean || batch <= 0 || heads <= 0 || n <= 0 || n_qtiles <= 0) return MCA_E_BADARG;
  if ((q_hm != nullptr) != (do_hm != nullptr) || (q_hm && !q)) return MCA_E_BADARG;
  if (q_hm && (q_ld % 8 || q_bstride % 8 || (uintptr_t)q % 16 || (uintptr_t)q_hm % 16 || (uintptr_t)do_hm % 16)) return MCA_E_ALIGN;
  if (o_ld % 8 || o_bstride % 8 || (uintptr_t)o % 16 || (uintptr_t)d_o % 16) return MCA_E_ALIGN;
  if (heads > 65535 || batch > 65535) return MCA_E_UNSUPPORTED;
  const int n_prep = (n + PREP_ROWS - 1) / PREP_ROWS;
  hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3(n_prep + heads, batch), dim3(256), 0, as_stream(stream), o, d_o, o_bstride, o_ld, lse, (float*)nullptr,
                     dvmean, heads, n, 1.f / (float)n, n_prep, row_slot, rowc, n_qtiles, q, q_bstride, q_ld, q_hm, do_hm);
  return launch_status();
}
