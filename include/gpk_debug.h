/* gpk_debug.h -- the ONE development call of libgpk.so: NOT part of the drop-in boundary (include/gpk.h, include/gpk_mg.h).
 *
 * Kernel-variant / tuning switches for A/B measurements and for tests that force every instantiation.  Everything else that is
 * development-only -- the superseded kernel designs, shader-clock stamps, hardware probes, micro-benchmarks -- lives in
 * libgpk_dev.so (include/gpk_dev.h; csrc/dev/), which the product never loads.  gpk_tune changes ONE handle (no process-wide state in the library since round 4);
 * the product path never calls it.
 */
#ifndef GPK_DEBUG_H
#define GPK_DEBUG_H

#include "gpk.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Development / tuning switch of ONE handle: key -> a field of the handle's GpkTune (csrc/gpk_common.h documents every key next to
 * its default; tools/README.md lists them).  Examples: key 0 = force the GEMM tile configuration (0 auto, 1 = 128x128, 2 = 64x64);
 * key 10 = 0: substitution strips even when Dinv is given; key 12 = 0: SYRK then right-looking Cholesky on one stream instead of the
 * two-partition pipeline; key 13 = CUs of the chain partition (default 32); key 23 = 0: dense schedule for the Eikonal, Burgers and
 * Darcy systems; key 24 = workgroups per split-K product launch of the pipeline (default 1000, 0 = no split); key 52 = the loss
 * gpk_gn_step reports (1, default: true substitution, one vector, its chain on the handle's internal side stream next to the end of the
 * step -- include/gpk.h, gpk_gn_step; 2: the same chain on the handle's stream in front of the solve phase; 0: the approximate number
 * taken from the F column of the GEMM-only solve, rounds 2-4).
 * No process-wide state (round 4): handles of one process can run different variants side by side.  Unknown key: error. */
int gpk_tune(gpk_handle h, int key, int value);
#ifdef __cplusplus
}
#endif
#endif /* GPK_DEBUG_H */
