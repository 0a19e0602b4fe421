// gen_kernels.h -- internal: device waveform generator launch interface.
#pragma once
#include <hip/hip_runtime.h>

#include "gen_device.h"

namespace sdr_amd {

struct GenArgs {
    float* out;        // blocks*block_size samples (complex: float pairs)
    const float* phi;  // per-block start phase, cycles in [0,1); unused with pstate
    GenDevState* pstate; // non-NULL: closed-form phases from the device's own position (gen_device.h), moved on
    unsigned pincr;      // by `blocks` blocks when the launch's last workgroup finishes
    long long blocks;
    long long block_size;
    float amp;
    float cps;         // cycles per sample
    int waveform;      // 0 cosine 1 sine 2 square 3 triangle 4 sawtooth
    int complex_samples;
};

hipError_t launch_gen(const GenArgs& a, hipStream_t stream);
// out[s] = {re[s], scale * q[s]}, s in [0, n)
hipError_t launch_gen_interleave(float* out, const float* re, const float* q, float scale, long long n, hipStream_t stream);

} // namespace sdr_amd
