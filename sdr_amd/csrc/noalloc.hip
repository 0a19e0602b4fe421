// noalloc.hip -- sdr_fir_apply_device_noalloc: the streaming FIR of sdr_fir_create +
// sdr_fir_process_device as ONE stateless call that allocates nothing, for callers that cannot
// call hipMalloc (code under HIP graph capture, tight serving loops, harnesses that forbid it).
//
// The existing entry points keep what their kernels need beyond input and output -- the
// compacted taps and the carried history -- in buffers hipMalloc'ed by sdr_fir_create.  Here the
// same two buffers live in `scratch`, a file-scope device array of fixed size (8 KiB; every GPU
// the library runs on holds its own copy until the process ends).  Host code never passes the
// array's name to a kernel: its device address is looked up once per device with
// hipGetSymbolAddress and kept in g_scratch[device].
//
// Every call on one GPU shares that GPU's array: a small set-up kernel rewrites the taps and
// zeroes the history at the start of each call (stream-ordered before the filter kernel), so
// calls may follow each other on one stream, but calls that OVERLAP on one GPU (two streams, two
// threads) race on it.
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>

#include "../../include/sdr_amd.h"
#include "errors.h"
#include "fir_kernels.h"
#include "handles.h"

namespace {

constexpr int kTapsFloats = 512;                      // [0, 512): compacted taps
constexpr int kScratchFloats = SDR_NOALLOC_SCRATCH_BYTES / 4; // [512, 2048): zero history
constexpr int kMaxDevices = 64;

static __device__ __align__(256) float scratch[kScratchFloats];

std::mutex g_mu;
float* g_scratch[kMaxDevices]; // device address of `scratch`, per device (0 = not looked up yet)

// compacted taps by value: the set-up kernel needs no host-to-device copy
struct TapsArg {
    float g[SDR_NOALLOC_MAX_TAPS + 33];
};

__global__ void noalloc_setup(TapsArg t, int ntap_floats, float* s, int total) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x)
        s[i] = i < ntap_floats ? t.g[i] : 0.0f; // taps, then zeros up to the end of the history region
}

int scratch_for(int device, float** out) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_scratch[device]) {
        // with `device` current (the caller holds a DeviceGuard): each GPU has its own copy
        float* p = nullptr;
        hipError_t e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return sdr_amd::fail(SDR_ERR_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
        g_scratch[device] = p;
    }
    *out = g_scratch[device];
    return SDR_OK;
}

} // namespace

using namespace sdr_amd;

extern "C" int sdr_fir_apply_device_noalloc(const float* taps, size_t ntaps, int sample_kind, const void* d_in,
                                            size_t n, void* d_out, int device, void* hip_stream) {
    clear_error();
    if (!taps || ntaps == 0) return fail(SDR_ERR_INVALID, "sdr_fir_apply_device_noalloc: no taps");
    if (ntaps > SDR_NOALLOC_MAX_TAPS)
        return fail(SDR_ERR_INVALID, "sdr_fir_apply_device_noalloc: more than SDR_NOALLOC_MAX_TAPS (255) taps: the fixed "
                                     "scratch array holds the taps and history of the direct-form kernels only");
    if (sample_kind != SDR_REAL && sample_kind != SDR_COMPLEX)
        return fail(SDR_ERR_INVALID, "sdr_fir_apply_device_noalloc: sample_kind must be SDR_REAL or SDR_COMPLEX");
    if (n == 0) return SDR_OK;
    if (n > SDR_NOALLOC_MAX_SAMPLES)
        return fail(SDR_ERR_INVALID, "sdr_fir_apply_device_noalloc: more than SDR_NOALLOC_MAX_SAMPLES (2^36) samples");
    if (!d_in || !d_out) return fail(SDR_ERR_INVALID, "sdr_fir_apply_device_noalloc: NULL buffer");
    const int w = sample_kind == SDR_COMPLEX ? 2 : 1;
    const size_t wbytes = w * sizeof(float);
    if (((uintptr_t)d_in % wbytes) || ((uintptr_t)d_out % wbytes))
        return fail(SDR_ERR_INVALID, "sdr_fir_apply_device_noalloc: buffers must be aligned to the sample size");
    {
        const uintptr_t i0 = (uintptr_t)d_in, o0 = (uintptr_t)d_out;
        if (i0 < o0 + n * wbytes && o0 < i0 + n * wbytes)
            return fail(SDR_ERR_INVALID, "sdr_fir_apply_device_noalloc: in and out overlap");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(SDR_ERR_NO_DEVICE, "sdr_fir_apply_device_noalloc: no HIP device (libsdr_amd has no CPU fallback)");
    if (device < 0 || device >= ndev || device >= kMaxDevices)
        return fail(SDR_ERR_INVALID, "sdr_fir_apply_device_noalloc: device out of range");

    DirectPlan plan;
    int rc = plan_direct(taps, ntaps, sample_kind, &plan);
    if (rc != SDR_OK) return rc;
    if (plan.taps_floats > sizeof(TapsArg) / sizeof(float) || plan.taps_floats > (size_t)kTapsFloats ||
        (size_t)plan.hist_len * w > (size_t)(kScratchFloats - kTapsFloats))
        return fail(SDR_ERR_INVALID, "sdr_fir_apply_device_noalloc: filter does not fit the fixed scratch array");
    if (kernel_lds_bytes(*plan.k, plan.nchunks) > 64 * 1024)
        return fail(SDR_ERR_INVALID, "sdr_fir_apply_device_noalloc: filter needs more LDS than a plain launch provides");

    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return fail(SDR_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(guard.err));
    float* ptr = nullptr;
    rc = scratch_for(device, &ptr);
    if (rc != SDR_OK) return rc;
    hipStream_t stream = (hipStream_t)hip_stream;

    TapsArg t;
    plan_direct_fill(taps, plan, t.g);
    const int total = kTapsFloats + plan.hist_len * w; // taps region, then the history the kernel may read
    hipLaunchKernelGGL(noalloc_setup, dim3(4), dim3(256), 0, stream, t, (int)plan.taps_floats, ptr, total);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SDR_ERR_HIP, std::string("noalloc_setup: ") + hipGetErrorString(e));

    FirArgs a;
    a.in = (const float*)d_in;
    a.out = (float*)d_out;
    a.hist[0] = a.hist[1] = ptr + kTapsFloats;
    a.state = nullptr; // stateless: always the zero history
    a.taps = ptr;
    a.n = (long long)n;
    a.in_stride = a.out_stride = (long long)n;
    a.nchunks = plan.nchunks;
    a.hist_len = plan.hist_len;
    a.shift = plan.shift;
    a.vec_in = (uintptr_t)d_in % 16 == 0;
    a.vec_out = (uintptr_t)d_out % 16 == 0;
    e = launch_fir(*plan.k, a, 1, stream);
    if (e != hipSuccess) return fail(SDR_ERR_HIP, std::string("launch_fir: ") + hipGetErrorString(e));
    return SDR_OK;
}
