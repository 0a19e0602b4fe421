// chain.hip -- a cascade of FIR stages evaluated as one filter (C ABI sdr_chain_*).
//
// The reference chains blocks through unix pipes, one kfr::fir per process
// (gen | hilbert | ...).  On the GPU every intermediate stream would cost a full
// HBM write + read; since every stage is a real-tap FIR the cascade is itself an FIR
// whose taps are the convolution of the stage taps (the stage spectra multiply), so
// the whole chain runs as ONE pass of the FIR kernels.  What does not commute is the
// group-delay skip at the start of a stream: stage k drops its first skips[k] outputs,
// and the next stage never sees them.  That makes the first
//      M0 = sum_{k>=2} max(0, ntaps[k] - 1 - skips[k])
// chain outputs differ from the single filter's; those few are computed stage by stage
// on a short prefix of the input and patched over the single-filter result.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sdr_amd.h"
#include "errors.h"
#include "handles.h"

using sdr_amd::fail;

struct sdr_chain {
    int device = 0, kind = SDR_REAL, w = 1;
    size_t channels = 1;
    std::vector<sdr_fir_t*> stage; // stage-by-stage filters (warm-up prefix only)
    std::vector<size_t> skips;
    std::vector<size_t> skip_left; // per stage, for the staged prefix
    sdr_fir_t* fused = nullptr;    // the single equivalent filter
    size_t fused_ntaps = 0;
    std::vector<float> fused_taps; // kept to rebuild the fused filter on the overlap-save path (generate_device)
    size_t total_skip = 0; // D: outputs of the fused filter dropped at stream start
    size_t warm = 0;       // M0: leading chain outputs taken from the staged path
    size_t prefix = 0;     // P = M0 + D inputs go through the staged path too
    size_t pos = 0;        // inputs consumed so far
    size_t emitted = 0;    // outputs produced so far
    bool dirty = false;    // the fused filter has taken samples since the last reset / seek
    float* scratch[3] = {nullptr, nullptr, nullptr}; // [channels][prefix] each
    float* warm_in = nullptr; // sdr_chain_generate_device: the generated packets that cover the warm-up prefix
    size_t warm_cap = 0;      // floats
};

static int hipfail(hipError_t e, const char* what) {
    return fail(SDR_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

extern "C" void sdr_chain_destroy(sdr_chain_t* c) {
    if (!c) return;
    for (sdr_fir_t* f : c->stage) sdr_fir_destroy(f);
    sdr_fir_destroy(c->fused);
    for (float* p : c->scratch)
        if (p) (void)hipFree(p);
    if (c->warm_in) (void)hipFree(c->warm_in);
    delete c;
}

extern "C" int sdr_chain_create(sdr_chain_t** out, size_t nstages, const float* const* taps, const size_t* ntaps,
                                const size_t* skips, int sample_kind, size_t channels, int device) {
    if (!out) return fail(SDR_ERR_INVALID, "sdr_chain_create: out is NULL");
    *out = nullptr;
    if (nstages == 0 || nstages > 16 || !taps || !ntaps) return fail(SDR_ERR_INVALID, "sdr_chain_create: 1..16 stages");
    for (size_t k = 0; k < nstages; ++k)
        if (!taps[k] || ntaps[k] == 0) return fail(SDR_ERR_INVALID, "sdr_chain_create: empty stage");
    sdr_chain* c = new (std::nothrow) sdr_chain();
    if (!c) return fail(SDR_ERR_NOMEM, "sdr_chain_create: out of memory");
    c->device = device;
    c->kind = sample_kind;
    c->w = sample_kind == SDR_COMPLEX ? 2 : 1;
    c->channels = channels;

    // taps of the cascade: convolution of the stage taps, in double
    std::vector<double> h(taps[0], taps[0] + ntaps[0]);
    for (size_t k = 1; k < nstages; ++k) {
        std::vector<double> g(h.size() + ntaps[k] - 1, 0.0);
        for (size_t i = 0; i < h.size(); ++i)
            for (size_t j = 0; j < ntaps[k]; ++j) g[i + j] += h[i] * (double)taps[k][j];
        h.swap(g);
    }
    std::vector<float> hf(h.begin(), h.end());
    c->fused_ntaps = hf.size();
    c->fused_taps = hf;
    for (size_t k = 0; k < nstages; ++k) {
        const size_t d = skips ? skips[k] : 0;
        c->skips.push_back(d);
        c->total_skip += d;
        if (k >= 1) c->warm += ntaps[k] - 1 > d ? ntaps[k] - 1 - d : 0;
    }
    c->prefix = nstages > 1 ? c->warm + c->total_skip : 0;
    c->skip_left = c->skips;

    int rc = sdr_fir_create(&c->fused, hf.data(), hf.size(), sample_kind, channels, device);
    if (rc == SDR_OK && nstages > 1) {
        for (size_t k = 0; k < nstages && rc == SDR_OK; ++k) {
            sdr_fir_t* f = nullptr;
            rc = sdr_fir_create(&f, taps[k], ntaps[k], sample_kind, channels, device);
            if (rc == SDR_OK) c->stage.push_back(f);
        }
    }
    if (rc == SDR_OK) {
        sdr_amd::DeviceGuard guard(device);
        if (guard.err != hipSuccess) rc = hipfail(guard.err, "hipSetDevice");
        const size_t bytes = std::max<size_t>(c->prefix + c->total_skip, 1) * channels * c->w * sizeof(float);
        for (float*& p : c->scratch) {
            if (rc != SDR_OK) break;
            hipError_t e = hipMalloc(&p, bytes);
            if (e != hipSuccess) rc = hipfail(e, "hipMalloc(chain scratch)");
        }
    }
    if (rc != SDR_OK) {
        std::string keep = sdr_last_error();
        sdr_chain_destroy(c);
        return fail(rc, keep);
    }
    *out = c;
    return SDR_OK;
}

extern "C" size_t sdr_chain_ntaps(const sdr_chain_t* c) { return c ? c->fused_ntaps : 0; }

extern "C" int sdr_chain_reset(sdr_chain_t* c) {
    if (!c) return fail(SDR_ERR_INVALID, "sdr_chain_reset: NULL handle");
    int rc = sdr_fir_reset(c->fused);
    for (sdr_fir_t* f : c->stage)
        if (rc == SDR_OK) rc = sdr_fir_reset(f);
    c->skip_left = c->skips;
    c->pos = c->emitted = 0;
    c->dirty = false;
    return rc;
}

// Smallest position sdr_chain_seek accepts (besides 0): past the start-of-stream region -- the
// staged warm-up prefix or the swallowed outputs, whichever is longer -- plus one filter length.
extern "C" size_t sdr_chain_min_seek(const sdr_chain_t* c) {
    if (!c) return 0;
    return std::max(c->prefix, c->total_skip) + c->fused_ntaps;
}

extern "C" int sdr_chain_seek(sdr_chain_t* c, size_t position) {
    if (!c) return fail(SDR_ERR_INVALID, "sdr_chain_seek: NULL handle");
    if (position == 0) return sdr_chain_reset(c);
    if (position < sdr_chain_min_seek(c))
        return fail(SDR_ERR_INVALID, "sdr_chain_seek: position lies inside the start-of-stream region (see sdr_chain_min_seek)");
    int rc = sdr_chain_reset(c);
    if (rc != SDR_OK) return rc;
    for (size_t& s : c->skip_left) s = 0;   // (sdr_chain_reset above zeroed the histories: dirty = false)
    c->pos = position;
    c->emitted = position - c->total_skip;
    return SDR_OK;
}

extern "C" int sdr_chain_process_device(sdr_chain_t* c, const void* d_in, size_t in_stride, size_t n, void* d_out,
                                        size_t out_stride, size_t* n_out, void* hip_stream) {
    if (!c) return fail(SDR_ERR_INVALID, "sdr_chain_process_device: NULL handle");
    if (n_out) *n_out = 0;
    if (n == 0) return SDR_OK;
    if (!d_in || !d_out) return fail(SDR_ERR_INVALID, "sdr_chain_process_device: NULL buffer");
    hipStream_t stream = (hipStream_t)hip_stream;
    // every copy and launch below belongs to the chain's device, whatever device is current
    sdr_amd::DeviceGuard guard(c->device);
    if (guard.err != hipSuccess) return hipfail(guard.err, "hipSetDevice");
    // Past the start of the stream a chain call is one call of the fused filter, whose stream position is
    // device state: safe to record into a HIP graph and replay.  Inside the start-of-stream region the work
    // depends on the host-side position (swallowed outputs, staged warm-up): refused rather than recorded.
    if (c->pos < std::max(c->prefix, c->total_skip) && sdr_amd::stream_is_capturing(stream))
        return fail(SDR_ERR_INVALID, "sdr_chain_process_device: the start-of-stream region (group-delay skip, staged warm-up) "
                                     "cannot be recorded into a HIP graph; capture a later call");
    c->dirty = true;
    const size_t w = c->w, fw = w * sizeof(float);
    const float* in = (const float*)d_in;
    float* out = (float*)d_out;
    int rc;

    // ---- the single-pass filter over all n inputs ---------------------------------
    // its first (total_skip - pos) outputs, if any, are swallowed: they go to scratch
    const size_t swallow = c->pos < c->total_skip ? std::min(n, c->total_skip - c->pos) : 0;
    if (swallow) {
        rc = sdr_fir_process_device(c->fused, in, in_stride, swallow, c->scratch[2], c->total_skip, stream);
        if (rc != SDR_OK) return rc;
    }
    const size_t produced = n - swallow;
    if (produced) {
        rc = sdr_fir_process_device(c->fused, in + swallow * w, in_stride, produced, out, out_stride, stream);
        if (rc != SDR_OK) return rc;
    }

    // ---- stage by stage over the inputs that fall inside the warm-up prefix ---------
    if (!c->stage.empty() && c->pos < c->prefix) {
        size_t m = std::min(n, c->prefix - c->pos); // samples entering stage 0
        const float* src = in;
        size_t src_stride = in_stride;
        const size_t cap = c->prefix + c->total_skip;
        for (size_t k = 0; k < c->stage.size() && m; ++k) {
            float* dst = c->scratch[k & 1];
            rc = sdr_fir_process_device(c->stage[k], src, src_stride, m, dst, cap, stream);
            if (rc != SDR_OK) return rc;
            const size_t drop = std::min(m, c->skip_left[k]);
            c->skip_left[k] -= drop;
            src = dst + drop * w;
            src_stride = cap;
            m -= drop;
        }
        // m chain outputs, global indices [emitted, emitted + m): patch those below `warm`
        if (m && c->emitted < c->warm) {
            const size_t take = std::min(m, c->warm - c->emitted);
            for (size_t ch = 0; ch < c->channels; ++ch) {
                hipError_t e = hipMemcpyAsync(out + ch * out_stride * w, src + ch * src_stride * w, take * fw,
                                              hipMemcpyDeviceToDevice, stream);
                if (e != hipSuccess) return hipfail(e, "hipMemcpyAsync(chain warm-up)");
            }
        }
    }
    c->pos += n;
    c->emitted += produced;
    if (n_out) *n_out = produced;
    return SDR_OK;
}

extern "C" int sdr_chain_generate_device(sdr_chain_t* c, sdr_gen_t* g, size_t blocks, void* d_out, size_t* n_out,
                                         void* hip_stream) {
    if (n_out) *n_out = 0;
    if (!c || !g) return fail(SDR_ERR_INVALID, "sdr_chain_generate_device: NULL handle");
    // The generator is fused into the overlap-save kernels only.  A chain whose fused filter is
    // short enough for the direct form is moved over before its first sample (allocates once).
    sdr_amd::DeviceGuard guard(c->device);
    if (guard.err != hipSuccess) return hipfail(guard.err, "hipSetDevice");
    if (!sdr_amd::fir_uses_overlap_save(c->fused) && c->kind == SDR_REAL && c->channels == 1) {
        // moving the filter is safe while its history is still zero: at the start of the stream,
        // after a reset, and after a seek (which zeroes the histories)
        if (c->dirty)
            return fail(SDR_ERR_INVALID, "sdr_chain_generate_device: this short chain has already filtered samples through the "
                                         "direct-form kernel; generate right after create, sdr_chain_reset or sdr_chain_seek");
        sdr_fir_t* f = nullptr;
        int rc0 = sdr_amd::fir_create_overlap_save(&f, c->fused_taps.data(), c->fused_taps.size(), c->kind, c->channels, c->device);
        if (rc0 != SDR_OK) return rc0;
        sdr_fir_destroy(c->fused);
        c->fused = f;
    }
    int rc = sdr_amd::fir_generated_supported(c->fused, g);
    if (rc != SDR_OK) return rc;
    if (blocks == 0) return SDR_OK;
    if (!d_out) return fail(SDR_ERR_INVALID, "sdr_chain_generate_device: NULL buffer");
    hipStream_t stream = (hipStream_t)hip_stream;
    const size_t B = sdr_amd::gen_block_size(g);
    float* out = (float*)d_out;
    size_t done = 0;

    // The packets that overlap the start-of-stream region (swallowed outputs, staged
    // warm-up) are materialised once and take the ordinary path.
    const size_t lead = std::max(c->prefix, c->total_skip);
    if (c->pos < lead) {
        const size_t k = std::min(blocks, (lead - c->pos + B - 1) / B);
        if (k * B > c->warm_cap) {
            if (c->warm_in) (void)hipFree(c->warm_in);
            c->warm_in = nullptr;
            c->warm_cap = 0;
            hipError_t e = hipMalloc(&c->warm_in, k * B * sizeof(float));
            if (e != hipSuccess) return hipfail(e, "hipMalloc(chain warm-up input)");
            c->warm_cap = k * B;
        }
        rc = sdr_gen_generate_device(g, k, c->warm_in, stream);
        if (rc != SDR_OK) return rc;
        size_t n1 = 0;
        rc = sdr_chain_process_device(c, c->warm_in, k * B, k * B, out, k * B, &n1, stream);
        if (rc != SDR_OK) return rc;
        done = n1;
        blocks -= k;
    }
    if (blocks) {
        rc = sdr_amd::fir_process_generated(c->fused, g, blocks, out + done, stream);
        if (rc != SDR_OK) return rc;
        c->dirty = true;
        c->pos += blocks * B;
        c->emitted += blocks * B;
        done += blocks * B;
    }
    if (n_out) *n_out = done;
    return SDR_OK;
}
