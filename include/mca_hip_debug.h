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
/* knob[key] = value for key in [0, 16); returns 0.  All knobs are 0 in production. */
int mca_debug_set(int key, int value);
/* every knob back to 0 */
int mca_debug_reset(void);
#ifdef __cplusplus
}
#endif
#endif
