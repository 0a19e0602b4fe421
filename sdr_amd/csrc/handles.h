// handles.h -- internal: what chain.hip needs from the FIR and generator handles
// that live in sdr_capi.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "../../include/sdr_amd.h"

namespace sdr_amd {

// Makes `device` current for the scope, restores the caller's device after.
struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) {
            err = hipSetDevice(device);
            changed = err == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (changed) (void)hipSetDevice(prev);
    }
};

// True while `s` is recording into a HIP graph (sdr_capi.hip).
bool stream_is_capturing(hipStream_t s);

// Like sdr_fir_process_device with the generator's next `blocks` packets as the input,
// synthesised inside the FIR kernel (nothing is written but the blocks*B outputs).
// Needs the overlap-save path, real samples, one channel, 256 <= B <= 2^23.
int fir_process_generated(sdr_fir_t* f, sdr_gen_t* g, size_t blocks, float* d_out, hipStream_t stream);
// SDR_OK when fir_process_generated would accept the pair, else the error (message set).
int fir_generated_supported(const sdr_fir_t* f, const sdr_gen_t* g);
size_t gen_block_size(const sdr_gen_t* g);
// sdr_fir_create with the overlap-save path chosen whatever the tap count (the fused
// generator lives in those kernels only); and which path a handle uses.
int fir_create_overlap_save(sdr_fir_t** out, const float* taps, size_t ntaps, int sample_kind, size_t channels,
                            int device);
bool fir_uses_overlap_save(const sdr_fir_t* f);

} // namespace sdr_amd
