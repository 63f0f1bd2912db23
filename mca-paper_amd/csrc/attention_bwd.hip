// Block-masked fused attention, backward (autograd of model.py:73-105).  P is recomputed from Q, K and the
// forward's log-sum-exp; nothing of size N x N is ever stored.
//
// One workgroup = 8 wavefronts = a block of 256 keys of one (sample, head); wavefront w owns keys
// [32w, 32w+32) and keeps K, V fragments plus the dK^T and dV^T accumulators of its keys in registers for
// the whole kernel (no cross-workgroup sum for dK/dV).  The workgroup sweeps the 32-row query tiles the
// fusion structure allows for its key block (CSR list from the host).  Per query tile and wavefront, with
// the KEY on the MFMA lane (cdna guide, Appendix B "Attention backward"):
//     S[q][key]   = Q · K^T           A = Q rows (LDS, ds_read_b128),  B = K fragments (registers)
//     dP[q][key]  = dO · V^T          A = dO rows (LDS),               B = V fragments (registers)
//     P = exp2(c·S − lse),  dS = P ∘ (dP − delta)     (−lse/c and −delta are the accumulators' start values)
//     dV^T[d][key] += dO^T · P        A = dO^T (ds_read_b64_tr_b16 of the same LDS image), B = P accumulators
//     dK^T[d][key] += Q^T · dS        A = Q^T  (transposed reads),                         B = dS accumulators
//     dQ[q][d]     += dS · K          dS crosses LDS once (transposed image), 16x16x32 MFMA, each wavefront
//                                     owns one 16x16 block of the 32x64 dQ tile -> fp32 atomics to HBM
// One barrier per 64-query step (two 32-row sub-tiles back to back, so the MFMAs of one overlap the VALU of the other).
#include "common.h"

#define BKEYS 256
#define BQ 64        // query rows per step: two 32-row sub-tiles between barriers
#define DH 64
#define MAX_QTILES 2048

// [32 rows][64 d] image used for BOTH row reads and transposed reads (Q and dO tiles)
__device__ __forceinline__ int qd_off(int r, int c) {
  const int s = ((r >> 1) & 7) ^ (((r >> 1) & 1) << 2);
  return r * 64 + ((c ^ s) << 3);
}
// [256 keys][64 d] K image, transposed reads of 8 consecutive keys x 16 d
__device__ __forceinline__ int kt_off(int key, int c) { return key * 64 + ((c ^ (((key >> 1) & 3) << 1)) << 3); }
// [256 keys][32 q] dS^T image (64-byte rows)
__device__ __forceinline__ int ds_off(int key, int c) { return key * 32 + ((c ^ ((key >> 1) & 3)) << 3); }

// timeline probe (knob 6 bit 3): wave 0 of one workgroup writes s_memtime stamps, read back by tools/trace_attn_bwd.py
MCA_TRACE_BUFFER(attn_bwd)
#ifdef MCA_TRACE_BUILD      // the stamps pin the instruction order, so the production build carries none (build.py: trace=True)
#define AB_STAMP() do { if (tracing && ti < 1000) mca_trace_attn_bwd[ti++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AB_STAMP() do { } while (0)
#endif

__global__ __launch_bounds__(512) void attn_bwd_kernel(mca_attn_bwd_args a, int dbg) {
  extern __shared__ __attribute__((aligned(16))) u16 lds[];
  u16* Qs = lds;                               // 2 x 64 x 64
  u16* Os = Qs + 2 * BQ * DH;                  // 2 x 64 x 64   (dO)
  u16* Ds = Os + 2 * BQ * DH;                  // 2 x [2 sub-tiles] x 256 x 32  (dS^T)
  u16* Kimg = Ds + 2 * BKEYS * BQ;             // 256 x 64
  float* rowc = reinterpret_cast<float*>(Kimg + BKEYS * DH);      // [2][3][64]: lse, delta, qmask(bits)
  // This key block's query-tile list, copied once and read with inline-asm LDS reads: a global load of the next entry at
  // the top of every step carries s_waitcnt vmcnt(0), i.e. waits for the dQ atomics the previous step has just issued.
  uint32_t* qlist = reinterpret_cast<uint32_t*>(rowc + 2 * 192);   // [MAX_QTILES]
  float* dvm_s = reinterpret_cast<float*>(qlist + MAX_QTILES);     // [64] dvmean of this (sample, head): read in the epilogue

  // XCD-aware order: the key blocks of one (sample, head) sweep the same Q / dO tiles: one XCD takes them all
#ifdef MCA_TRACE_BUILD
  const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif
  const int lin0 = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
  const int lin = (dbg & 16) ? lin0 : xcd_remap(lin0, (int)(gridDim.x * gridDim.y * gridDim.z));          // knob 9 = 16: launch order (A/B)
  // launch slot -> key block, list bounds and first query tile: one 16-byte scalar load when the host passes the table
  int kbi, it_begin, n_it, first_qt;
  if (a.k_wg) {
    const int4 w = reinterpret_cast<const int4*>(a.k_wg)[lin % (int)gridDim.x];
    kbi = w.x; it_begin = w.y; n_it = w.z; first_qt = w.w;
  } else {
    kbi = a.k_order[lin % (int)gridDim.x];
    it_begin = a.k_ptr[kbi];
    n_it = a.k_ptr[kbi + 1] - it_begin;          // <= n_qtiles <= MAX_QTILES (checked by the host)
    first_qt = n_it > 0 ? (int)(a.k_qt[it_begin] & 0x7fffffffu) : 0;
  }
  const int h = (lin / (int)gridDim.x) % (int)gridDim.y, b = lin / (int)(gridDim.x * gridDim.y);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int key0 = kbi * BKEYS;
  const int mykey = key0 + wave * 32 + l31;
  int keyc = mykey; if (keyc > a.nk - 1) keyc = a.nk - 1;
  // q pre-scaled by scale * log2(e) (MCA_ATTN_Q_PRESCALED): q.k is the log2-domain logit; dK^T then carries that factor too
  const bool prescaled = (a.flags & MCA_ATTN_Q_PRESCALED) != 0;
  const float c2 = prescaled ? 1.f : a.scale * 1.4426950408889634f, inv_c2 = 1.f / c2;
  const float dk_scale = prescaled ? 0.6931471805599453f : a.scale;
  // One workgroup per CU: nothing hides this prologue, so its dependent loads are kept few (slot table above); the first
  // Q / dO tile, the V fragments and the K image are all in flight together, and the LDS writes and the barrier come after
  // all of them.  The K fragments are read back from the K image (the direct form is 16-byte pieces of 32 rows per instruction).

  const u16* kbase = a.k + (int64_t)b * a.kv_bstride + h * DH;
  const u16* vbase = a.v + (int64_t)b * a.kv_bstride + h * DH;
  // K / V fragments (B operands): lane = key, k = d
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int s = 0; s < 4; s++) vf[s] = *reinterpret_cast<const bf16x8*>(vbase + (int64_t)keyc * a.kv_ld + 16 * s + 8 * lh);
  const uint32_t kinfo = a.keyinfo[(int64_t)b * a.nk_pad + mykey];     // nk_pad covers every key block
  const bool key_ok = kinfo != 31u;
  const uint32_t keybit = key_ok ? (1u << kinfo) : 0u;          // bit of this lane's key group; 0 = padded key
  const bool wave_keys_ok = __all(key_ok);                      // wave-uniform

  f32x16 dk[2], dv[2];
#pragma unroll
  for (int n = 0; n < 2; n++)
#pragma unroll
    for (int r = 0; r < 16; r++) { dk[n][r] = 0.f; dv[n][r] = 0.f; }

  const u16* qbase = a.q + (int64_t)b * a.q_bstride + h * DH;
  const u16* obase = a.d_o + (int64_t)b * a.o_bstride + h * DH;
  const float* lse_g = a.lse + ((int64_t)b * a.heads + h) * a.nq;
  const float* delta_g = a.delta + ((int64_t)b * a.heads + h) * a.nq;

  // staging: a 64-row tile is 512 chunks of 16 B per operand: one Q chunk and one dO chunk per thread; threads
  // 0..191 also carry the row constants (lse, delta, qmask)
  const int srow = tid >> 3, sc = tid & 7;
  bf16x8 stage_q, stage_o;
  float stage_c = 0.f;
  bool stage_oob = false;
  // row constants: ONE unconditional load through a selected pointer (three predicated loads into the same register made
  // hipcc wait vmcnt(0) between them, i.e. for the Q / dO loads just issued and every dQ atomic in flight)
  const int cwhich = tid >> 6 < 2 ? tid >> 6 : 2, crow = tid & 63;
  const float* cbase = cwhich == 0 ? lse_g : (cwhich == 1 ? delta_g : reinterpret_cast<const float*>(a.qmask));
  auto gload = [&](int qt) {
    int q = qt * BQ + srow; if (q > a.nq - 1) q = a.nq - 1;
    stage_q = *reinterpret_cast<const bf16x8*>(qbase + (int64_t)q * a.q_ld + sc * 8);
    stage_o = *reinterpret_cast<const bf16x8*>(obase + (int64_t)q * a.o_ld + sc * 8);
    int qq = qt * BQ + crow;
    stage_oob = qq >= a.nq;
    if (qq > a.nq - 1) qq = a.nq - 1;
    stage_c = cbase[qq];
  };
  auto swrite = [&](int buf) {
    *reinterpret_cast<bf16x8*>(Qs + buf * BQ * DH + qd_off(srow, sc)) = stage_q;
    *reinterpret_cast<bf16x8*>(Os + buf * BQ * DH + qd_off(srow, sc)) = stage_o;
    // rows past nq contribute nothing: lse = +inf (P = 0), delta = 0, qmask = 0
    // stored as the accumulator start values: -lse / c2 (so that exp2(c2 * acc) = P; +inf lse -> -inf -> P = 0), -delta
    if (tid < 192) {
      float v = stage_oob ? (cwhich == 0 ? INFINITY : 0.f) : stage_c;
      if (cwhich == 0) v = -v * inv_c2; else if (cwhich == 1) v = -v;
      rowc[buf * 192 + tid] = v;
    }
  };

  if (n_it > 0) gload(first_qt);
#ifdef MCA_TRACE_BUILD
  const unsigned long long t_chain = __builtin_amdgcn_s_memtime();          // the scalar chain has delivered first_qt
#endif
  // K image for the dQ product: the four chunks of a thread are loaded before any is written
  {
    bf16x8 kimg_r[BKEYS * 8 / 512];
#pragma unroll
    for (int u = 0; u < BKEYS * 8 / 512; u++) {
      const int id = tid + 512 * u, r = id >> 3, c = id & 7;
      int kk = key0 + r; if (kk > a.nk - 1) kk = a.nk - 1;
      kimg_r[u] = *reinterpret_cast<const bf16x8*>(kbase + (int64_t)kk * a.kv_ld + c * 8);
    }
    for (int i = tid; i < n_it; i += 512) qlist[i] = a.k_qt[it_begin + i];
    if (tid < DH) dvm_s[tid] = a.dvmean[(int64_t)b * a.heads * DH + h * DH + tid];          // (the epilogue would wait a memory round trip for it)
#pragma unroll
    for (int u = 0; u < BKEYS * 8 / 512; u++) {
      const int id = tid + 512 * u;
      *reinterpret_cast<bf16x8*>(Kimg + kt_off(id >> 3, id & 7)) = kimg_r[u];
    }
  }
  int buf = 0;
#ifdef MCA_TRACE_BUILD
  const unsigned long long t_kimg = __builtin_amdgcn_s_memtime();          // K image written (its loads have landed)
#endif
  if (n_it > 0) swrite(0);
#ifdef MCA_TRACE_BUILD
  const unsigned long long t_sw = __builtin_amdgcn_s_memtime();
#endif
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 4; s++) kf[s] = *reinterpret_cast<const bf16x8*>(Kimg + kt_off(wave * 32 + l31, 2 * s + lh));
  const unsigned lds_qlist = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const uint32_t*)qlist;
  auto entry_issue = [&](int i, uint32_t& v) {          // uniform LDS read, result valid after the next lgkmcnt(0)
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(lds_qlist + 4u * (unsigned)i));
  };
  uint32_t e_cur_v = 0, e_nxt_v = 0;
  if (n_it > 0) entry_issue(0, e_cur_v);
  entry_issue(n_it > 1 ? 1 : 0, e_nxt_v);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  uint32_t e_cur = __builtin_amdgcn_readfirstlane(e_cur_v), e_nxt = __builtin_amdgcn_readfirstlane(e_nxt_v);

  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1, g4 = lane >> 4;
  const int qb = wave & 1, db = wave >> 1;       // this wavefront's 16x16 block of the dQ tile

  const bool tracing = (dbg & 8) && blockIdx.x == 2 && blockIdx.y == 0 && blockIdx.z == (gridDim.z > 16 ? 16 : 0) && tid == 0;          // a workgroup of a middle round, not of the cold first one
  int ti = 0;
#ifdef MCA_TRACE_BUILD
  if (tracing) {
    mca_trace_attn_bwd[1000] = t_start; mca_trace_attn_bwd[1001] = __builtin_amdgcn_s_memtime();
    mca_trace_attn_bwd[1004] = t_chain; mca_trace_attn_bwd[1005] = t_kimg; mca_trace_attn_bwd[1006] = t_sw;
  }
#endif
  for (int it = 0; it < n_it; it++) {
    AB_STAMP();
    const uint32_t ent = e_cur;
    const int qt = (int)(ent & 0x7fffffffu);
    const bool full = (ent >> 31) != 0;
    if (it + 1 < n_it) gload((int)(e_nxt & 0x7fffffffu));
    uint32_t e_nn_v;
    entry_issue(it + 2 < n_it ? it + 2 : it, e_nn_v);          // entry it+2: lands under this step's first LDS wait
#pragma unroll
    for (int sub = 0; sub < 2; sub++) {
    const u16* qs = Qs + buf * BQ * DH + sub * 32 * DH;
    const u16* os = Os + buf * BQ * DH + sub * 32 * DH;
    const float* rc = rowc + buf * 192 + sub * 32;
    u16* ds = Ds + buf * BKEYS * BQ + sub * BKEYS * 32;

    // ---- S and dP (rows = q in registers, column = key on the lane)
    // The accumulators start from the row constants (-lse / c2 and -delta, stored that way by swrite): the products come out
    // as S - lse / c2 and dP - delta, and the LDS latency of the constants hides behind the fragment reads instead of
    // standing in the middle of the exp / multiply phase.  (Computing S / dP of both sub-tiles first, so that the MFMAs of
    // the second run under the exp phase of the first, needs 32 more registers than the 256 a wavefront has here: 55 spills.)
    f32x16 s, dp;
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const f32x4 lse4 = *reinterpret_cast<const f32x4*>(rc + 8 * g + 4 * lh);
      const f32x4 del4 = *reinterpret_cast<const f32x4*>(rc + 64 + 8 * g + 4 * lh);
#pragma unroll
      for (int e = 0; e < 4; e++) { s[4 * g + e] = lse4[e]; dp[4 * g + e] = del4[e]; }
    }
#pragma unroll
    for (int st = 0; st < 4; st++) {
      const bf16x8 qfrag = *reinterpret_cast<const bf16x8*>(qs + qd_off(l31, 2 * st + lh));
      const bf16x8 ofrag = *reinterpret_cast<const bf16x8*>(os + qd_off(l31, 2 * st + lh));
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfrag, kf[st], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ofrag, vf[st], dp, 0, 0, 0);
    }
    AB_STAMP();
    // ---- P, dS.  Fast path: a structurally full tile whose 32 keys (this wavefront's) are all valid needs no mask.
    bf16x8 pb[2], sb[2];
    if (full && wave_keys_ok) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float p = __builtin_amdgcn_exp2f(s[r] * c2);
        const float dsv = p * dp[r];
        pb[r >> 3][r & 7] = (short)f2bf(p);
        sb[r >> 3][r & 7] = (short)f2bf(dsv);
      }
    } else {
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const f32x4 qm4 = *reinterpret_cast<const f32x4*>(rc + 128 + 8 * g + 4 * lh);
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int r = 4 * g + e;
          const uint32_t qsel = full ? 0xffffffffu : __float_as_uint(qm4[e]);
          float p = __builtin_amdgcn_exp2f(s[r] * c2);
          p = (qsel & keybit) ? p : 0.f;
          const float dsv = p * dp[r];
          pb[r >> 3][r & 7] = (short)f2bf(p);
          sb[r >> 3][r & 7] = (short)f2bf(dsv);
        }
      }
    }
    AB_STAMP();
    // ---- dS^T to LDS: lane = key row, 4 consecutive q per store
#pragma unroll
    for (int g = 0; g < 4; g++) {
      uint2 pk;
      pk.x = (uint32_t)(u16)sb[g >> 1][4 * (g & 1)] | ((uint32_t)(u16)sb[g >> 1][4 * (g & 1) + 1] << 16);
      pk.y = (uint32_t)(u16)sb[g >> 1][4 * (g & 1) + 2] | ((uint32_t)(u16)sb[g >> 1][4 * (g & 1) + 3] << 16);
      *reinterpret_cast<uint2*>(ds + ds_off(wave * 32 + l31, g) + 4 * lh) = pk;
    }
    // ---- dV^T += dO^T P ; dK^T += Q^T dS   (element j of k-step sp carries q = 16sp + 8(j>>2) + 4lh + (j&3))
#pragma unroll
    for (int sp = 0; sp < 2; sp++)
#pragma unroll
      for (int n = 0; n < 2; n++) {
        bf16x8 ot, qtf;
#pragma unroll
        for (int t = 0; t < 2; t++) {
          const int qr = 16 * sp + 8 * t + 4 * lh + tq;
          const int d = n * 32 + 16 * tg + 4 * tp;
          const bf16x4 o4 = lds_read_tr16(os + qd_off(qr, d >> 3) + (d & 7));
          const bf16x4 q4 = lds_read_tr16(qs + qd_off(qr, d >> 3) + (d & 7));
#pragma unroll
          for (int e = 0; e < 4; e++) { ot[4 * t + e] = o4[e]; qtf[4 * t + e] = q4[e]; }
        }
        dv[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ot, pb[sp], dv[n], 0, 0, 0);
        dk[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, sb[sp], dk[n], 0, 0, 0);
      }
    AB_STAMP();

    }
    if (it + 1 < n_it) swrite(buf ^ 1);
    AB_STAMP();
    __syncthreads();
    AB_STAMP();

    // ---- dQ tile (64 x 64) = dS (64 x 256) · K (256 x 64): per 32-row sub-tile this wavefront owns one 16x16 block,
    //      8 k-steps of 32 keys.  key carried by (lane group g4, read t, element e): 32ks + 16(g4>>1) + 8t + 4(g4&1) + e
    {
      const u16* dsb = Ds + buf * BKEYS * BQ;
      f32x4 dq[2];
      dq[0] = f32x4{0.f, 0.f, 0.f, 0.f}; dq[1] = dq[0];
#pragma unroll
      for (int ks = 0; ks < 8; ks++) {
        bf16x8 af[2], bfr;
#pragma unroll
        for (int t = 0; t < 2; t++) {
          const int key = 32 * ks + 16 * (g4 >> 1) + 8 * t + 4 * (g4 & 1) + tq;
          const int qc = qb * 16 + 4 * tp;          // q column inside the dS^T row
          const int dc = db * 16 + 4 * tp;          // d column inside the K row
          const bf16x4 b4 = lds_read_tr16(Kimg + kt_off(key, dc >> 3) + (dc & 7));
#pragma unroll
          for (int sub = 0; sub < 2; sub++) {
            const bf16x4 a4 = lds_read_tr16(dsb + sub * BKEYS * 32 + ds_off(key, qc >> 3) + (qc & 7));
#pragma unroll
            for (int e = 0; e < 4; e++) af[sub][4 * t + e] = a4[e];
          }
#pragma unroll
          for (int e = 0; e < 4; e++) bfr[4 * t + e] = b4[e];
        }
#pragma unroll
        for (int sub = 0; sub < 2; sub++) dq[sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[sub], bfr, dq[sub], 0, 0, 0);
      }
      AB_STAMP();
      float* dqp = a.dq + (int64_t)b * a.dq_bstride + h * DH + db * 16 + (lane & 15);
#pragma unroll
      for (int sub = 0; sub < 2; sub++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int q = qt * BQ + sub * 32 + qb * 16 + 4 * g4 + e;
          if (q < a.nq && !(dbg & 1)) atomicAdd(dqp + (int64_t)q * a.dq_ld, dq[sub][e] * a.scale);
          if ((dbg & 1) && dq[sub][e] == 123.456f) dqp[0] = 1.f;
        }
    }
    AB_STAMP();
    buf ^= 1;
    e_cur = e_nxt;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (long done: the dQ product above has waited on younger reads)
    e_nxt = __builtin_amdgcn_readfirstlane(e_nn_v);
  }

#ifdef MCA_TRACE_BUILD
  if (tracing) mca_trace_attn_bwd[1002] = __builtin_amdgcn_s_memtime();
#endif
  // ---- epilogue: dK = scale * dK^T, dV = dV^T + dvmean (uniform rows spread over every key)
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
#ifdef MCA_TRACE_BUILD
  if (tracing) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); mca_trace_attn_bwd[1003] = __builtin_amdgcn_s_memtime(); }
#endif
}

#define BWD_LDS_BYTES ((2 * BQ * DH * 2 + 2 * BKEYS * BQ + BKEYS * DH) * 2 + 2 * 192 * 4 + MAX_QTILES * 4 + DH * 4)

extern "C" int mca_attn_bwd(const mca_attn_bwd_args* a, mca_stream_t stream) {
  if (!a || !a->q || !a->k || !a->v || !a->d_o || !a->lse || !a->delta || !a->dvmean || !a->dq || !a->dk || !a->dv ||
      !a->qmask || !a->keyinfo || !a->k_ptr || !a->k_qt || !a->k_order)
    return MCA_E_BADARG;
  if (a->batch <= 0 || a->heads <= 0 || a->nq <= 0 || a->nk <= 0) return MCA_E_BADARG;
  if (a->n_qtiles != (a->nq + BQ - 1) / BQ || a->n_ktiles != (a->nk + BKEYS - 1) / BKEYS) return MCA_E_BADARG;
  if (a->nk_pad < a->n_ktiles * BKEYS) return MCA_E_BADARG;
  if (a->q_ld % 8 || a->kv_ld % 8 || a->o_ld % 8 || a->dkv_ld % 4 || a->q_bstride % 8 || a->kv_bstride % 8 ||
      a->o_bstride % 8 || a->dkv_bstride % 4)
    return MCA_E_ALIGN;
  if ((uintptr_t)a->q % 16 || (uintptr_t)a->k % 16 || (uintptr_t)a->v % 16 || (uintptr_t)a->d_o % 16 ||
      (uintptr_t)a->dk % 8 || (uintptr_t)a->dv % 8)
    return MCA_E_ALIGN;
  if (a->heads > 65535 || a->batch > 65535 || a->n_qtiles > MAX_QTILES) return MCA_E_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            BWD_LDS_BYTES) != hipSuccess)
      return MCA_E_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(a->n_ktiles, a->heads, a->batch), dim3(512), BWD_LDS_BYTES, as_stream(stream), *a, mca_knobs[6] | mca_knobs[9]);
  return launch_status();
}

// =====================================================================================================
// delta[b,h,q] = sum_d dO[q,h,d] * O[q,h,d];  dvmean[b, h*64+d] = (1/nk) sum over uniform rows (lse = +inf) of dO
// one wavefront per (b, q) row: lane covers 8 contiguous columns of the 512-wide row -> head = lane / 8
// =====================================================================================================
#define PREP_ROWS 32
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const u16* __restrict__ o, const u16* __restrict__ d_o,
                                                             int64_t bstride, int64_t ld, const float* __restrict__ lse,
                                                             float* __restrict__ delta, float* __restrict__ dvmean,
                                                             int heads, int nq, float inv_nk) {
  // lse / delta are (b, head, q): a row touches them at a stride of nq floats per head.  They cross LDS so that the global
  // accesses are 128-byte runs along q (one row at a time they were 4-byte accesses in 8 different lines per row).
  __shared__ float lse_s[8][PREP_ROWS], del_s[8][PREP_ROWS];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cols = heads * DH;
  const int q_begin = blockIdx.x * PREP_ROWS;
  int q_end = q_begin + PREP_ROWS; if (q_end > nq) q_end = nq;
  for (int c0 = 0; c0 < cols; c0 += 512) {
    const int h0 = c0 / DH;
    const int c = c0 + lane * 8;
    const int hl = lane >> 3;          // head of this lane inside the 512-column slab
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool any = false;
    // all PREP_ROWS / 4 rows of this wavefront are requested before any is used (a row at a time the loop is latency-bound),
    // and before the lse values the block waits for at its first barrier
    bf16x8 ovs[PREP_ROWS / 4], dvs[PREP_ROWS / 4];
#pragma unroll
    for (int k = 0; k < PREP_ROWS / 4; k++) {
      int q = q_begin + wave + 4 * k; if (q > nq - 1) q = nq - 1;
      const int cc = c < cols ? c : 0;
      ovs[k] = *reinterpret_cast<const bf16x8*>(o + (int64_t)b * bstride + (int64_t)q * ld + cc);
      dvs[k] = *reinterpret_cast<const bf16x8*>(d_o + (int64_t)b * bstride + (int64_t)q * ld + cc);
    }
    {
      const int hh = tid / PREP_ROWS, r = tid % PREP_ROWS;          // 8 heads x 32 rows = 256 threads
      if (h0 + hh < heads && q_begin + r < q_end) lse_s[hh][r] = lse[((int64_t)b * heads + h0 + hh) * nq + q_begin + r];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PREP_ROWS / 4; k++) {
      const int q = q_begin + wave + 4 * k;
      if (q >= q_end) break;
      float part = 0.f;
      if (c < cols) {
        const bool uni = lse_s[hl][q - q_begin] == INFINITY;
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const float dvj = bf2f((u16)dvs[k][j]);
          part += dvj * bf2f((u16)ovs[k][j]);
          if (uni) { acc[j] += dvj; any = true; }
        }
      }
      // 8 lanes per head
      part += __shfl_xor(part, 1, WAVE); part += __shfl_xor(part, 2, WAVE); part += __shfl_xor(part, 4, WAVE);
      if ((lane & 7) == 0) del_s[hl][q - q_begin] = part;
    }
    __syncthreads();
    {
      const int hh = tid / PREP_ROWS, r = tid % PREP_ROWS;
      if (h0 + hh < heads && q_begin + r < q_end) delta[((int64_t)b * heads + h0 + hh) * nq + q_begin + r] = del_s[hh][r];
    }
    if (any && c < cols) {
#pragma unroll
      for (int j = 0; j < 8; j++)
        if (acc[j] != 0.f) atomicAdd(dvmean + (int64_t)b * cols + c + j, acc[j] * inv_nk);
    }
    __syncthreads();
  }
}
// (a kernel, not hipMemsetAsync: as a memset NODE of a captured step the zeroing was not ordered against the kernels around it
// when the captured graph was a single chain - the whole backward then started from a dvmean full of stale sums)
__global__ __launch_bounds__(256) void zero_f32_kernel(float* __restrict__ p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0.f;
}
extern "C" int mca_attn_bwd_prep(const uint16_t* o, const uint16_t* d_o, int64_t o_bstride, int64_t o_ld,
                                 const float* lse, float* delta, float* dvmean, int batch, int heads, int nq, int nk,
                                 mca_stream_t stream) {
  if (!o || !d_o || !lse || !delta || !dvmean || batch <= 0 || heads <= 0 || nq <= 0 || nk <= 0) return MCA_E_BADARG;
  if (o_ld % 8 || o_bstride % 8 || (uintptr_t)o % 16 || (uintptr_t)d_o % 16) return MCA_E_ALIGN;
  const int64_t nz = (int64_t)batch * heads * DH;
  hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)((nz + 255) / 256)), dim3(256), 0, as_stream(stream), dvmean, nz);
  hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3((nq + PREP_ROWS - 1) / PREP_ROWS, batch), dim3(256), 0, as_stream(stream), o, d_o,
                     o_bstride, o_ld, lse, delta, dvmean, heads, nq, 1.f / (float)nk);
  return launch_status();
}
