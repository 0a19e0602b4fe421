// fir_kernels.h -- internal interface between the C ABI (sdr_capi.hip) and the
// gfx950 kernels (fir_kernels.hip).  Not installed; the public surface is
// include/sdr_amd.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace sdr_amd {

struct FirArgs {
    const float* in;      // channel 0, sample 0
    float* out;           // channel 0, output 0
    const float* hist;    // [channels][hist_len] samples preceding `in`
    const float* taps;    // compacted (every STRIDE-th) taps, zero-padded to nchunks*R/STRIDE
    long long n;          // samples per channel
    long long in_stride;  // samples between channels
    long long out_stride;
    int nchunks;          // tap chunks of R sample offsets each; even
    int hist_len;         // samples of history per channel (>= nchunks*R + shift, multiple of 4)
    int shift;            // phase of the first kept tap for STRIDE=2 (0 or 1)
    int vec_in;           // every channel's input and history are 16-byte aligned
    int vec_out;
};

struct KernelInfo {
    const void* fn;
    const char* name;
    int R;
    int stride;
    bool real;
};

const KernelInfo* find_kernel(int R, int stride, bool real);
size_t kernel_lds_bytes(const KernelInfo& k, int nchunks);
hipError_t prepare_kernel(const KernelInfo& k, int nchunks);
hipError_t launch_fir(const KernelInfo& k, const FirArgs& a, size_t channels, hipStream_t stream);
hipError_t launch_carry_history(const float* in, long long in_stride, long long n, const float* old_hist,
                                float* new_hist, int hist_len, int w, size_t channels, hipStream_t stream);

} // namespace sdr_amd
