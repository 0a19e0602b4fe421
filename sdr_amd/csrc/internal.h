/* internal.h -- hooks of libsdr_amd.so that are NOT part of the C ABI (include/sdr_amd.h):
 * used by the repository's own tests and experiment scripts only. */
#pragma once
#include "../../include/sdr_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* sdr_fir_create with the kernel family pinned: path -1 = automatic (what sdr_fir_create does),
 * 0 = direct form, 1 = overlap-save, 2 = matrix-core direct form (fir_mfma); grid > 0 overrides the number of persistent workgroups of the
 * overlap-save kernels (tests make runs of tiles cross channel boundaries with tiny grids). */
int sdr_amd_internal_fir_create(sdr_fir_t** out, const float* taps, size_t ntaps, int sample_kind,
                                size_t channels, int device, int path, int grid);

#ifdef __cplusplus
}
#endif
