/*
 * mca_hip_debug.h — measurement hooks of libmca_hip.so.  NOT part of the drop-in ABI (include/mca_hip.h): these entry
 * points exist for the A/B tools under tools/ and for the tests that compare the production kernels with their
 * conservative forms.  The knob table is process-global; production code never sets it, and every test that does restores
 * it in a fixture finaliser (tests/conftest.py).
 */
#ifndef MCA_HIP_DEBUG_H
#define MCA_HIP_DEBUG_H
#ifdef __cplusplus
extern "C" {
#endif
/* knob[key] = value for key in [0, 16); returns 0.  All knobs are 0 in production.  What the library reads them for:
 *    0  persistent NT GEMMs: 8 = s_memtime stamps of workgroup 0 (tools/trace_persist.py), 32 * P = P row panels per tile group
 *    1  = 1: the 128 x 128 NT kernel for every shape;   2  bit 1: weight-gradient 128 x 128 kernel without its atomics (timing)
 *    3  > 0: number of (uniform) row splits of the weight-gradient kernels;   4  = 1: no residual prefetch in the fp32 + residual NT kernel
 *    6  = r + 1: the grouped weight-gradient launch gives its two-tile workgroups 32 r rows less than the others; -(r + 1): the
 *       same with a purely tile-major line (no owner segments)
 *    5  weight-gradient kernel: 1 = 128 x 128, 2 = 256 x 128 where 256 x 256 would be chosen
 *    7  1 = one-tile-per-workgroup NT kernels only, 3 = persistent kernel for fp32 + residual as well;   10  = 1: 256 x 128 persistent
 *       kernel where the 256 x 256 one would be chosen;   11  = 1: one launch per member of a weight-gradient group
 *    8 / 9  attention kernels (9 also the weight-gradient kernels), bit mask: 8 = s_memtime stamps / clock probe of the launch
 *       (trace build; gemm_tn_256x256_group_kernel in every build), 16 = launch order without the XCD remap, 32 = fp8 forward with
 *       register staging instead of LDS-DMA
 *   12  = 1: general LayerNorm kernels where the trunk forms would be chosen
 *   14  > 0: cap on the workgroups of the LayerNorm backward kernels (default 256) and of mca_reduce_rows (default 512)
 *   (13 and 15 selected / tuned the forward attention forms of round 4; gone with tools/overlays/fwd64)
 *   one-pass attention backward (attention_bwd1.hip), knob 9: 8 = s_memtime stamps (trace build, tools/trace_bwd1.py), 16 = launch
 *       order without the XCD remap, 64 = the plain (compiler-scheduled) kernel instead of the pipelined one (cross-check) */
int mca_debug_set(int key, int value);
/* every knob back to 0 */
int mca_debug_reset(void);
#ifdef __cplusplus
}
#endif
#endif
