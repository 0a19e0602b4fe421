// sdr_capi.hip -- the C ABI of libsdr_amd.so (include/sdr_amd.h) over the gfx950
// kernels in fir_kernels.hip.  There is no CPU compute path in this library:
// without a HIP device every compute entry point returns SDR_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sdr_amd.h"
#include "errors.h"
#include "fir_kernels.h"
#include "gen_kernels.h"
#include "handles.h"

namespace sdr_amd {

static thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}
void clear_error() { g_last_error.clear(); }

static int hip_fail(hipError_t e, const char* what) {
    int code = (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? SDR_ERR_NO_DEVICE
               : (e == hipErrorOutOfMemory)                          ? SDR_ERR_NOMEM
                                                                     : SDR_ERR_HIP;
    return fail(code, std::string(what) + ": " + hipGetErrorString(e));
}

// A stream under HIP graph capture records work instead of running it.  The FIR entry points are safe to
// capture (their stream position is device state); calls whose work depends on host-side stream position
// ask this and refuse with a message instead of recording something that replays wrongly.
bool stream_is_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return st != hipStreamCaptureStatusNone;
}

#define SDR_HIP(expr)                                              \
    do {                                                           \
        hipError_t e_ = (expr);                                    \
        if (e_ != hipSuccess) return sdr_amd::hip_fail(e_, #expr); \
    } while (0)

} // namespace sdr_amd

using namespace sdr_amd;

struct sdr_fir {
    int device = 0;
    int kind = SDR_REAL;
    size_t channels = 1;
    size_t ntaps = 0;
    int w = 1; // floats per sample
    const KernelInfo* k = nullptr;
    int nchunks = 0;
    int hist_len = 0;
    int shift = 0;
    float* d_taps = nullptr;
    bool use_mfma = false;  // matrix-core direct form (fir_mfma.hip)
    int mfma_smax = 0;
    float* d_atab = nullptr;
    bool use_fft = false;   // overlap-save FFT kernel instead of the direct form
    int fft_halo = 0;
    int fft_parts = 1;
    int fft_grid = 0;
#ifdef SDR_AMD_DIAG
    int fft_ablate = 0, fft_contig = 0; // diagnostics (SDR_AMD_FFT_ABLATE / SDR_AMD_FFT_CONTIG), read at create
#endif
    float* d_hperm = nullptr;
    const float* d_wtab = nullptr; // per-lane twiddle tables, stored behind the spectra in d_hperm
    // Real samples, every other tap zero (the Hilbert design: non-zero taps on the even indices): y[n] depends on
    // inputs of n's parity only, so the stream IS a complex stream of half the length -- (x[2m], x[2m+1]) -- filtered
    // by the kept taps.  Calls that allow it (even n and strides, 8-byte aligned buffers) run the complex kernel on
    // that view: 8-byte accesses instead of 4-byte ones, half the halo, and 2049..4097 such taps need no partitions.
    bool fft_pair = false;
    int pair_halo = 0, pair_parts = 1, pair_grid = 0, pair_resident = 0;
    float* d_hperm_pair = nullptr; // spectra of the kept taps
    // 16384-point kernel (fir_fft16k.hip): more than 2049 taps (partitioned at 8192 taps beyond 8193)
    int use16k = 0; // 0 no, 1 by the size of the call, 2 always (asked for by name)
    int halo16k = 0;
    float* d_hperm16k = nullptr;   // its one spectrum, then its twiddles
    float* d_hperm16k_pair = nullptr; // the kept taps' spectrum for the complex-pair view of an every-other-tap design
    int pair_halo16k = 0;
    int parts16 = 1;               // > 1: more than 8193 taps, partitions of 8192 taps
    const float* d_wtab16k = nullptr;
    int cus = 256;
    unsigned* d_tickets = nullptr; // tile-ticket counters behind the twiddle table (FftArgs::tickets): zero between calls
    int fft_resident = 0;
    // The stream's history: the last hist_len samples, in one of two buffers; WHICH one is device state
    // (d_state->cur), read and flipped by the kernels themselves.  The host keeps no position of the
    // stream, so graph replays of a captured call and direct calls continue one another.
    float* d_hist[2] = {nullptr, nullptr};
    FirDevState* d_state = nullptr;
    // Onset of the stream.  kfr::fir is a direct form: each output is rounded relative to itself.  The overlap-save
    // kernels round at the level of the whole filter's spectrum, which a stream still SHORTER than the filter has
    // not reached (a windowed design's first taps are tiny: its onset lies orders of magnitude below the settled
    // level).  Calls that end inside the first ntaps samples of a stream therefore run the direct form (same
    // history buffers, same device state), when its staging fits the LDS (up to ~14k taps).  Host-side count,
    // saturating at ntaps; a captured call counts as past the onset (nobody knows how often it will be replayed).
    size_t onset_seen = 0;
    bool onset_direct = false;
    // Host-buffer entry points: one stream per staging slot, so that the device-to-host copy of
    // batch k (slot s) overlaps the host-to-device copy of batch k+1 (slot s^1) -- PCIe is full
    // duplex.  The filter kernels still run in submission order: each waits for the event recorded
    // behind the previous batch's kernels (the history carries from batch to batch).
    hipStream_t stream = nullptr;  // slot 0; also what reset / destroy synchronise first
    hipStream_t stream1 = nullptr; // slot 1
    hipEvent_t filtered = nullptr; // behind the kernels of the latest submission
    bool have_filtered = false;
    struct Slot {
        float* d_in = nullptr;
        float* d_out = nullptr;
        size_t cap = 0; // floats
        hipEvent_t done = nullptr;
        bool pending = false;
    } slot[2];
    hipStream_t slot_stream(int s) const { return s ? stream1 : stream; }
};

extern "C" const char* sdr_version(void) { return "sdr_amd 0.1 (gfx950)"; }
extern "C" const char* sdr_last_error(void) { return g_last_error.c_str(); }

extern "C" int sdr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// Diagnostic builds only (make diag -> libsdr_amd_diag.so, used by scripts/exp): kernel-family,
// tile and grid overrides from the environment.  The release library reads no environment
// variable on the compute path and has no switch that skips arithmetic.
#ifdef SDR_AMD_DIAG
static int env_int(const char* name, int dflt) {
    const char* s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}
#else
static inline int env_int(const char*, int dflt) { return dflt; }
#endif

// Direct-form plan for a tap vector: tap structure (every other tap zero => a kernel that skips
// them: the Hilbert transformer, reference include/sdr/hilbert.hpp:46-55), outputs per lane,
// chunk count.  plan_direct_fill writes the compacted taps g[i] = h[stride*i + shift], zero-padded
// to whole chunk pairs (plan.taps_floats floats).  Neither touches the heap on success.
int sdr_amd::plan_direct(const float* taps, size_t ntaps, int sample_kind, DirectPlan* p) {
    bool odd_zero = ntaps >= 8, even_zero = ntaps >= 8;
    for (size_t j = 0; j < ntaps; ++j) {
        if (taps[j] != 0.0f) {
            if (j & 1) odd_zero = false;
            else even_zero = false;
        }
    }
    int stride = 1, shift = 0;
    if (odd_zero && !even_zero) stride = 2, shift = 0;
    else if (even_zero && !odd_zero) stride = 2, shift = 1;
    if (env_int("SDR_AMD_DENSE", 0)) stride = 1, shift = 0;
    const size_t kept = (ntaps - shift + stride - 1) / stride;
    // Outputs per lane: 4 (8 KB tiles, more workgroups in flight: 5.9-6.0 TB/s) while the filter
    // is short enough to stay memory-bound that way, else 8 (profiles/r01_direct_r_sweep.txt)
    const int R = env_int("SDR_AMD_R", kept <= 31 ? 4 : 8);
    if (R != 4 && R != 8 && R != 16) return fail(SDR_ERR_INVALID, "SDR_AMD_R must be 4, 8 or 16");
    const KernelInfo* k = find_kernel(R, stride, sample_kind == SDR_REAL);
    if (!k) return fail(SDR_ERR_INVALID, "sdr_fir_create: no kernel for this configuration");
    const size_t span = (kept - 1) * stride + 1; // sample offsets covered
    int nchunks = (int)((span + R - 1) / R);
    nchunks += nchunks & 1;
    if (nchunks < 2) nchunks = 2;
    // complex samples, four outputs per lane: at least four chunks -- the tile's halo is then 16 samples = one 128-byte line,
    // where the 64-byte halo of two chunks leaves every tile's first line half used (2..8 taps: 0.733 -> 0.757 of HBM peak,
    // profiles/r03_direct_vs_mfma.txt; the extra taps are zeros)
    if (sample_kind == SDR_COMPLEX && R == 4 && nchunks < 4) nchunks = 4;
    p->k = k;
    p->stride = stride;
    p->shift = shift;
    p->R = R;
    p->nchunks = nchunks;
    p->hist_len = nchunks * R + 4;
    p->kept = kept;
    p->taps_floats = (size_t)nchunks * (R / stride);
    return SDR_OK;
}

void sdr_amd::plan_direct_fill(const float* taps, const DirectPlan& p, float* g) {
    for (size_t i = 0; i < p.taps_floats; ++i) g[i] = i < p.kept ? taps[p.stride * i + p.shift] : 0.0f;
}

// whether sdr_fir_create picks the 16384-point tiles by itself for complex filters of 2050..8193 taps
#ifndef SDR_FFT16K_AUTO
#define SDR_FFT16K_AUTO 1
#endif

// in-place radix-2 FFT in double (forward, exp(-i ...)); n = re.size() a power of two; cs / sn = cos / sin(2 pi m / n)
static void fft_radix2(std::vector<double>& re, std::vector<double>& im, const std::vector<double>& cs, const std::vector<double>& sn) {
    const size_t n = re.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const size_t step = n / len;
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const double wr = cs[k * step], wi = -sn[k * step];
                const size_t a = i + k, b = i + k + len / 2;
                const double xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
                re[b] = re[a] - xr; im[b] = im[a] - xi;
                re[a] += xr; im[a] += xi;
            }
    }
}

// path: SDR_PATH_AUTO = choose by tap count (measured crossovers), or one kernel family forced
// (internal callers and the test hook below; the public sdr_fir_create always passes AUTO).
// grid: workgroups of the persistent overlap-save kernels (0 = the default for the device).
static int fir_create_impl(sdr_fir_t** out, const float* taps, size_t ntaps, int sample_kind, size_t channels,
                           int device, int path, int grid) {
    clear_error();
    if (!out) return fail(SDR_ERR_INVALID, "sdr_fir_create: out is NULL");
    *out = nullptr;
    if (!taps || ntaps == 0) return fail(SDR_ERR_INVALID, "sdr_fir_create: no taps");
    if (ntaps > SDR_MAX_TAPS) return fail(SDR_ERR_INVALID, "sdr_fir_create: more than SDR_MAX_TAPS (131072) taps");
    if (sample_kind != SDR_REAL && sample_kind != SDR_COMPLEX)
        return fail(SDR_ERR_INVALID, "sdr_fir_create: sample_kind must be SDR_REAL or SDR_COMPLEX");
    if (channels == 0 || channels > 65535) return fail(SDR_ERR_INVALID, "sdr_fir_create: channels must be 1..65535");
    if (path < SDR_PATH_AUTO || path > SDR_PATH_DIRECT_FORMS) return fail(SDR_ERR_INVALID, "sdr_fir_create: unknown kernel family");
    const bool want_fft = path == SDR_PATH_FFT || path == SDR_PATH_FFT4K || path == SDR_PATH_FFT16K;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(SDR_ERR_NO_DEVICE, "sdr_fir_create: no HIP device (libsdr_amd has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(SDR_ERR_INVALID, "sdr_fir_create: device out of range");

    DirectPlan plan;
    int prc = plan_direct(taps, ntaps, sample_kind, &plan);
    if (prc != SDR_OK) return prc;
    const int stride = plan.stride, shift = plan.shift, nchunks = plan.nchunks;
    const KernelInfo* k = plan.k;
    std::vector<float> g(plan.taps_floats);
    plan_direct_fill(taps, plan, g.data());

    // Path choice.  Direct form: ntaps (or ntaps/2 with the Hilbert structure) packed
    // FMAs per complex sample; overlap-save FFT (N = 4096): ~50 packed ops per sample
    // times 4096/(4096-halo).
    int fft_halo = 0, fft_parts = 1;
    fir_fft_plan(ntaps, &fft_halo, &fft_parts);
    // Measured crossover (profiles/r01_direct_vs_fft_crossover.txt): the direct form holds
    // its memory-bound floor (5.7-6.0 TB/s) up to 32 dense taps (40 on real samples) or ~87
    // Hilbert-structured taps; from there the fast convolution is quicker.
    const size_t fft_from = stride == 2 ? 88u : sample_kind == SDR_COMPLEX ? 33u : 41u;
    bool use_fft = ntaps >= fft_from && fft_parts <= 64;   // (the matrix-core kernel may still take over below)
#ifdef SDR_AMD_DIAG
    if (const char* ev = getenv("SDR_AMD_FFT")) use_fft = atoi(ev) != 0 && fft_parts <= 64;
#endif
    if (path != SDR_PATH_AUTO) use_fft = want_fft && fft_parts <= 64;

    sdr_fir* f = new (std::nothrow) sdr_fir();
    if (!f) return fail(SDR_ERR_NOMEM, "sdr_fir_create: out of memory");
    f->device = device;
    f->kind = sample_kind;
    f->channels = channels;
    f->ntaps = ntaps;
    f->w = sample_kind == SDR_COMPLEX ? 2 : 1;
    f->k = k;
    f->nchunks = nchunks;
    f->shift = shift;
    f->hist_len = plan.hist_len;
    f->use_fft = use_fft;
    // matrix-core direct form: dense (or every-other-tap) filters of up to 129 kept taps
    // (complex samples: the 12-step variant from the start -- with 8 steps, up to 17 taps, the kernel runs 0.735-0.741 of HBM peak,
    // with 12, from 18 taps on, 0.752-0.758: profiles/r03_direct_vs_mfma.txt; the extra steps multiply zeros)
    const int mfma_smax = (shift == 0) ? mfma_steps_for(sample_kind == SDR_COMPLEX && plan.kept < 18 ? 18 : plan.kept) : 0;
    // Measured three-way crossover (profiles/r02_direct_vs_fft_vs_mfma.txt): with 17..49 kept taps
    // (dense, or every other tap of a Hilbert design up to 97 taps) the matrix-core kernel holds
    // 5.4-6.0 TB/s (0.67-0.75 of HBM peak) where the direct form has turned VALU-bound and the
    // overlap-save kernel sits at 5.1-5.3; below, the direct form is level or ahead; from 50 kept
    // taps up the fp32 matrix pipe is the bottleneck and the overlap-save kernel is faster.
    // (Round 3: the overlap-save kernel gained 1.4-3.7 % from its third ticket counter, and for dense filters the
    // crossover moved down with it -- level at 41 kept taps, overlap-save ahead by 0.6 % at 45 and 1.2-2 % at 49,
    // complex and real: profiles/r03_mfma_vs_fft.txt.  Every-other-tap designs stay level up to 49 kept taps.)
    // Real samples: the matrix-core kernel is ahead of the vector-ALU direct form at EVERY tap count -- 2..16 kept taps, dense or
    // every other tap: 0.772-0.785 of HBM peak against 0.72-0.737 (profiles/r03_direct_vs_mfma.txt; the direct form's 4-byte
    // accesses hold it back, the matrix-core kernel stages whole rows through the LDS).  Complex samples: the direct form stays
    // ahead up to 16 kept taps (0.754-0.759 against 0.735-0.741).
    const size_t mfma_first = sample_kind == SDR_REAL ? 1 : 17;
    const size_t mfma_last = stride == 2 ? 49 : 44;
    const bool mfma_auto = (path == SDR_PATH_AUTO || path == SDR_PATH_DIRECT_FORMS) && mfma_smax != 0 && plan.kept >= mfma_first && plan.kept <= mfma_last;
    if (mfma_auto) f->use_fft = use_fft = false;
    const bool use_mfma = mfma_auto || (!use_fft && path == SDR_PATH_MFMA);
    if (use_mfma && mfma_smax == 0) {
        delete f;
        return fail(SDR_ERR_INVALID, "sdr_fir_create: no matrix-core kernel for this filter (more than 129 kept taps, or odd-phase taps)");
    }
    if (use_mfma) {
        f->use_mfma = true;
        f->mfma_smax = mfma_smax;
        f->hist_len = std::max<int>(f->hist_len, (int)mfma_hist_samples(mfma_smax, stride) + 4);
    }
    if (use_fft) {
        f->fft_halo = fft_halo;
        f->fft_parts = fft_parts;
        f->hist_len = std::max((fft_parts > 1 ? fft_parts * 2048 : fft_halo) + 4, plan.hist_len); // (both even)
        if (sample_kind == SDR_REAL && stride == 2 && shift == 0 && plan.kept >= 2) {
            fir_fft_plan(plan.kept, &f->pair_halo, &f->pair_parts);
            f->fft_pair = true;
            const int need = 2 * (f->pair_parts > 1 ? f->pair_parts * 2048 : f->pair_halo) + 4; // real samples
            f->hist_len = std::max(f->hist_len, need); // even either way: both views share the history buffers
        }
    }

    DeviceGuard guard(device);
    int rc = SDR_OK;
    auto bail = [&](hipError_t err, const char* what) {
        rc = hip_fail(err, what);
        std::string keep = g_last_error;
        sdr_fir_destroy(f);
        g_last_error = keep;
        return rc;
    };
    if (guard.err != hipSuccess) return bail(guard.err, "hipSetDevice");
    if (!use_fft && (e = prepare_kernel(*k, nchunks)) != hipSuccess)
        return bail(e, "tap count too large for the direct-form kernel");
    if (use_fft) f->onset_direct = prepare_kernel(*k, nchunks) == hipSuccess; // the direct form takes the stream's onset
    const size_t hist_bytes = channels * (size_t)f->hist_len * f->w * sizeof(float);
    if ((e = hipMalloc(&f->d_taps, g.size() * sizeof(float))) != hipSuccess) return bail(e, "hipMalloc(taps)");
    if ((e = hipMalloc(&f->d_hist[0], hist_bytes)) != hipSuccess) return bail(e, "hipMalloc(history)");
    if ((e = hipMalloc(&f->d_hist[1], hist_bytes)) != hipSuccess) return bail(e, "hipMalloc(history)");
    if ((e = hipMalloc(&f->d_state, 256)) != hipSuccess) return bail(e, "hipMalloc(state)");
    if ((e = hipMemset(f->d_state, 0, 256)) != hipSuccess) return bail(e, "hipMemset(state)");
    if ((e = hipMemcpy(f->d_taps, g.data(), g.size() * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess)
        return bail(e, "hipMemcpy(taps)");
    if (use_mfma) {
        std::vector<float> at((size_t)mfma_smax * 64);
        mfma_fill_atab(g.data(), plan.kept, mfma_smax, at.data());
        if ((e = hipMalloc(&f->d_atab, at.size() * sizeof(float))) != hipSuccess) return bail(e, "hipMalloc(taps operand)");
        if ((e = hipMemcpy(f->d_atab, at.data(), at.size() * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess)
            return bail(e, "hipMemcpy(taps operand)");
    }
    if (use_fft) {
        // Filter spectrum H[k] = (1/N) sum_j h[j] exp(-2 pi i j k / N), N = 4096, in
        // double on the host, stored in the order the kernel's lanes consume it.
        const int N = 4096;
        std::vector<double> cs(N), sn(N);
        for (int m = 0; m < N; ++m) {
            cs[m] = std::cos(2.0 * M_PI * m / N);
            sn[m] = std::sin(2.0 * M_PI * m / N);
        }
        // partition p holds taps [p*2048, p*2048+2048) (all taps when parts == 1)
        auto spectra = [&](const float* t, size_t nt, int parts, float* dst) {
            const size_t plen = parts > 1 ? 2048 : nt;
            for (int p = 0; p < parts; ++p) {
                const size_t j0 = (size_t)p * plen, j1 = std::min(nt, j0 + plen);
                for (int k = 0; k < N; ++k) {
                    double re = 0.0, im = 0.0;
                    for (size_t j = j0; j < j1; ++j) {
                        const int m = (int)(((j - j0) * (size_t)k) & (N - 1));
                        re += t[j] * cs[m];
                        im -= t[j] * sn[m];
                    }
                    const int k0 = k & 15, k1 = (k >> 4) & 15, k2 = k >> 8;
                    const size_t pos = (size_t)p * N + k2 * 256 + 16 * k0 + k1;
                    dst[2 * pos] = (float)(re / N);
                    dst[2 * pos + 1] = (float)(im / N);
                }
            }
        };
        // spectra, the per-lane twiddle tables (LaneConstsT::init: W_4096^(tid k) at [256 k + tid], W_256^(n k) at
        // [4096 + 16 k + n]), zeroed ticket counters
        constexpr size_t TW = 4096 + 256;
        std::vector<float> hp((size_t)2 * N * fft_parts + 2 * TW + FFT_TICKET_WORDS);
        {
            float* tw = hp.data() + (size_t)2 * N * fft_parts;
            for (int k = 0; k < 16; ++k)
                for (int t = 0; t < 256; ++t) {
                    const int m = (t * k) & (N - 1);
                    tw[2 * (256 * k + t)] = (float)cs[m];
                    tw[2 * (256 * k + t) + 1] = (float)-sn[m];
                }
            for (int k = 0; k < 16; ++k)
                for (int n0 = 0; n0 < 16; ++n0) {
                    const int m = 16 * ((n0 * k) & 255);
                    tw[2 * (4096 + 16 * k + n0)] = (float)cs[m];
                    tw[2 * (4096 + 16 * k + n0) + 1] = (float)-sn[m];
                }
        }
        spectra(taps, ntaps, fft_parts, hp.data());
        if (f->fft_pair) {
            std::vector<float> hq((size_t)2 * N * f->pair_parts);
            spectra(g.data(), plan.kept, f->pair_parts, hq.data()); // g = the kept taps, in order
            if ((e = hipMalloc(&f->d_hperm_pair, hq.size() * sizeof(float))) != hipSuccess) return bail(e, "hipMalloc(spectrum)");
            if ((e = hipMemcpy(f->d_hperm_pair, hq.data(), hq.size() * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess)
                return bail(e, "hipMemcpy(spectrum)");
        }
        if ((e = hipMalloc(&f->d_hperm, hp.size() * sizeof(float))) != hipSuccess) return bail(e, "hipMalloc(spectrum)");
        if ((e = hipMemcpy(f->d_hperm, hp.data(), hp.size() * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess)
            return bail(e, "hipMemcpy(spectrum)");
        f->d_wtab = f->d_hperm + (size_t)2 * N * fft_parts;
        f->d_tickets = reinterpret_cast<unsigned*>(f->d_hperm + (size_t)2 * N * fft_parts + 2 * TW);
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
        // three workgroups are resident per CU (two for the partitioned variants); three rounds of
        // them leave a shorter tail than two (same-box sweep: 9 per CU 1 % ahead of 6, 12 and 16 level)
        f->fft_grid = cus * env_int("SDR_AMD_FFT_BLOCKS_PER_CU", fft_parts > 1 ? 2 : 9);
        f->fft_resident = cus * (fft_parts > 1 ? 2 : 3);
        f->cus = cus;
        f->pair_grid = grid > 0 ? grid : cus * (f->pair_parts > 1 ? 2 : 9);
        f->pair_resident = cus * (f->pair_parts > 1 ? 2 : 3);
        // tests: a tiny grid makes every workgroup walk long runs of tiles across channels
        f->fft_grid = std::max(1, env_int("SDR_AMD_FFT_GRID", grid > 0 ? grid : f->fft_grid));
#ifdef SDR_AMD_DIAG
        f->fft_ablate = env_int("SDR_AMD_FFT_ABLATE", 0);
        f->fft_contig = env_int("SDR_AMD_FFT_CONTIG", 0);
#endif
    }
    // 16384-point tiles: more than one and at most four 2048-tap partitions (2050..8193 taps); real samples ride
    // two stretches at a time; calls that take the complex-pair view of an every-other-tap design run the complex
    // kernel with the spectrum of the kept taps (up to 8193 kept taps: a 16385-tap Hilbert design)
    const bool want16k = use_fft && path != SDR_PATH_FFT4K && (path == SDR_PATH_FFT16K || SDR_FFT16K_AUTO);
    // (beyond 8193 taps: 16384-point tiles partitioned at 8192 taps, up to 16 partitions = 131072 taps)
    const bool main16k = want16k && (fft_parts >= 2 || fft_halo == 2048);   // from 1026 taps on
    const int parts16 = ntaps - 1 <= 8192 ? 1 : (int)((ntaps + 8191) / 8192);
    const bool pair16k = want16k && f->fft_pair && ((f->pair_parts >= 2 && f->pair_parts <= 4) || (f->pair_parts == 1 && f->pair_halo == 2048));
    if (main16k || pair16k) {
        const int NK = 16384;
        std::vector<double> cs(NK), sn(NK);
        for (int m = 0; m < NK; ++m) {
            cs[m] = std::cos(2.0 * M_PI * m / NK);
            sn[m] = std::sin(2.0 * M_PI * m / NK);
        }
        // spectra H_p / N of `nt` taps cut into `np` pieces (one piece: all of them; else 8192 taps each) in the kernel's lane
        // order, back to back, followed (with_twiddles) by the twiddle tables
        auto upload = [&](const float* t, size_t nt, int np, bool with_twiddles, float** dst) -> hipError_t {
            std::vector<float> nat((size_t)2 * NK), hk((size_t)2 * NK * np + (with_twiddles ? 2 * FFT16K_TWIDDLE_COMPLEX : 0));
            const size_t plen = np > 1 ? 8192 : nt;
            for (int p = 0; p < np; ++p) {
                std::vector<double> re(NK, 0.0), im(NK, 0.0);
                const size_t j0 = (size_t)p * plen, j1 = std::min(nt, j0 + plen);
                for (size_t j = j0; j < j1; ++j) re[j - j0] = t[j];
                fft_radix2(re, im, cs, sn);
                for (int k = 0; k < NK; ++k) {
                    nat[2 * k] = (float)(re[k] / NK);
                    nat[2 * k + 1] = (float)(im[k] / NK);
                }
                fir_fft16k_permute_spectrum(nat.data(), hk.data() + (size_t)2 * NK * p);
            }
            if (with_twiddles) fir_fft16k_fill_twiddles(cs.data(), sn.data(), hk.data() + (size_t)2 * NK * np);
            hipError_t err = hipMalloc(dst, hk.size() * sizeof(float));
            if (err != hipSuccess) return err;
            return hipMemcpy(*dst, hk.data(), hk.size() * sizeof(float), hipMemcpyHostToDevice);
        };
        // (the twiddles live behind whichever spectrum comes first)
        if (main16k) {
            if ((e = upload(taps, ntaps, parts16, true, &f->d_hperm16k)) != hipSuccess) return bail(e, "16384-point spectrum");
            f->d_wtab16k = f->d_hperm16k + (size_t)2 * NK * parts16;
            f->halo16k = parts16 > 1 ? 8192 : std::max(2048, 1024 * (int)((ntaps - 1 + 1023) / 1024)); // whole rows of 1024 samples
            f->parts16 = parts16;
        }
        if (pair16k) {
            if ((e = upload(g.data(), plan.kept, 1, !main16k, &f->d_hperm16k_pair)) != hipSuccess) return bail(e, "16384-point spectrum");
            if (!main16k) f->d_wtab16k = f->d_hperm16k_pair + (size_t)2 * NK;
            f->pair_halo16k = std::max(2048, 1024 * (int)((plan.kept - 1 + 1023) / 1024));
        }
        f->use16k = path == SDR_PATH_FFT16K ? 2 : 1;
    }
    if ((e = hipMemset(f->d_hist[0], 0, hist_bytes)) != hipSuccess) return bail(e, "hipMemset(history)");
    if ((e = hipMemset(f->d_hist[1], 0, hist_bytes)) != hipSuccess) return bail(e, "hipMemset(history)");
    if ((e = hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    if ((e = hipStreamCreateWithFlags(&f->stream1, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    if ((e = hipEventCreateWithFlags(&f->filtered, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    for (auto& s : f->slot)
        if ((e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    *out = f;
    return SDR_OK;
}

extern "C" int sdr_fir_create(sdr_fir_t** out, const float* taps, size_t ntaps, int sample_kind,
                              size_t channels, int device) {
    return fir_create_impl(out, taps, ntaps, sample_kind, channels, device, SDR_PATH_AUTO, 0);
}

extern "C" int sdr_fir_create_numerics(sdr_fir_t** out, const float* taps, size_t ntaps, int sample_kind,
                                       size_t channels, int device, int numerics) {
    if (numerics != SDR_NUMERICS_FAST && numerics != SDR_NUMERICS_PER_OUTPUT) {
        clear_error();
        if (out) *out = nullptr;
        return fail(SDR_ERR_INVALID, "sdr_fir_create_numerics: numerics must be SDR_NUMERICS_FAST or SDR_NUMERICS_PER_OUTPUT");
    }
    return fir_create_impl(out, taps, ntaps, sample_kind, channels, device,
                           numerics == SDR_NUMERICS_PER_OUTPUT ? SDR_PATH_DIRECT_FORMS : SDR_PATH_AUTO, 0);
}

// Test hook (declared in csrc/internal.h, not part of include/sdr_amd.h): sdr_fir_create with the
// kernel family and the persistent grid chosen by the caller, so that parity tests can hold every
// family to its parity bar at every tap count without any environment switch in the library.
extern "C" int sdr_amd_internal_fir_create(sdr_fir_t** out, const float* taps, size_t ntaps, int sample_kind,
                                           size_t channels, int device, int path, int grid) {
    return fir_create_impl(out, taps, ntaps, sample_kind, channels, device, path, grid);
}

int sdr_amd::fir_create_overlap_save(sdr_fir_t** out, const float* taps, size_t ntaps, int sample_kind,
                                     size_t channels, int device) {
    return fir_create_impl(out, taps, ntaps, sample_kind, channels, device, SDR_PATH_FFT, 0);
}

bool sdr_amd::fir_uses_overlap_save(const sdr_fir_t* f) { return f && f->use_fft; }

extern "C" void sdr_fir_destroy(sdr_fir_t* f) {
    if (!f) return;
    DeviceGuard guard(f->device);
    if (f->stream) (void)hipStreamSynchronize(f->stream);
    if (f->stream1) (void)hipStreamSynchronize(f->stream1);
    for (auto& s : f->slot) {
        if (s.d_in) (void)hipFree(s.d_in);
        if (s.d_out) (void)hipFree(s.d_out);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    if (f->d_taps) (void)hipFree(f->d_taps);
    if (f->d_atab) (void)hipFree(f->d_atab);
    if (f->d_hperm) (void)hipFree(f->d_hperm);
    if (f->d_hperm_pair) (void)hipFree(f->d_hperm_pair);
    if (f->d_hperm16k) (void)hipFree(f->d_hperm16k);
    if (f->d_hperm16k_pair) (void)hipFree(f->d_hperm16k_pair);
    if (f->d_hist[0]) (void)hipFree(f->d_hist[0]);
    if (f->d_hist[1]) (void)hipFree(f->d_hist[1]);
    if (f->d_state) (void)hipFree(f->d_state);
    if (f->stream) (void)hipStreamDestroy(f->stream);
    if (f->stream1) (void)hipStreamDestroy(f->stream1);
    if (f->filtered) (void)hipEventDestroy(f->filtered);
    delete f;
}

// what every overlap-save launch of a handle shares: history buffers, device state, ticket counters (zero
// between calls: the call's last workgroup to finish zeroes them), twiddles
static void fft_common_args(sdr_fir* f, FftArgs& fa) {
    fa.hist[0] = f->d_hist[0];
    fa.hist[1] = f->d_hist[1];
    fa.state = f->d_state;
    fa.wtab = f->d_wtab;
    fa.tickets = f->d_tickets;
    fa.resident = env_int("SDR_AMD_FFT_RESIDENT", f->fft_resident);
    if (!env_int("SDR_AMD_FFT_TICKETS", 1)) fa.tickets = nullptr; // diagnostic builds only
#ifdef SDR_AMD_DIAG
    fa.ablate = f->fft_ablate;
    fa.contig = f->fft_contig;
#endif
}

static int process_device_impl(sdr_fir* f, const float* d_in, size_t in_stride, size_t n, float* d_out,
                               size_t out_stride, hipStream_t stream) {
    if (n == 0) return SDR_OK;
    if (f->channels > 1 && (in_stride < n || out_stride < n))
        return fail(SDR_ERR_INVALID, "sdr_fir_process_device: channel stride smaller than n");
    const size_t wbytes = f->w * sizeof(float);
    if (((uintptr_t)d_in % wbytes) || ((uintptr_t)d_out % wbytes))
        return fail(SDR_ERR_INVALID, "sdr_fir_process_device: buffers must be aligned to the sample size");
    {
        // tiles re-read their neighbours' inputs (halo): filtering in place would feed outputs back in
        const uintptr_t i0 = (uintptr_t)d_in, o0 = (uintptr_t)d_out;
        const uintptr_t i1 = i0 + ((f->channels - 1) * in_stride + n) * wbytes;
        const uintptr_t o1 = o0 + ((f->channels - 1) * out_stride + n) * wbytes;
        if (i0 < o1 && o0 < i1) return fail(SDR_ERR_INVALID, "sdr_fir_process_device: in and out overlap");
    }
    bool onset_call = false;
    if (f->use_fft && f->onset_seen < f->ntaps) {
        if (stream_is_capturing(stream)) f->onset_seen = f->ntaps;
        else if (f->onset_direct && f->onset_seen + n <= f->ntaps) onset_call = true;
        f->onset_seen = std::min(f->ntaps, f->onset_seen + n);
    }
    if (f->use_fft && !onset_call) {
        FftArgs fa = {};
        fa.in = d_in;
        fa.out = d_out;
        fa.channels = (long long)f->channels;
        fa.t_lo = fa.t_hi = 0;
        fft_common_args(f, fa);
        // real samples as complex pairs (see sdr_fir::fft_pair) when this call's shape allows it; either view
        // leaves the same history behind, so calls of both kinds mix freely
        const bool pair = f->fft_pair && n % 2 == 0 && ((uintptr_t)d_in % 8 == 0) && ((uintptr_t)d_out % 8 == 0) &&
                          (f->channels == 1 || (in_stride % 2 == 0 && out_stride % 2 == 0));
        if (pair) {
            fa.hperm = f->d_hperm_pair;
            fa.n = (long long)(n / 2);
            fa.in_stride = (long long)(in_stride / 2);
            fa.out_stride = (long long)(out_stride / 2);
            fa.halo = f->pair_halo;
            fa.hist_len = f->hist_len / 2;
            fa.parts = f->pair_parts;
            fa.resident = env_int("SDR_AMD_FFT_RESIDENT", f->pair_resident);
            if (f->use16k && f->d_hperm16k_pair) {
                fa.use16k = f->use16k;
                fa.halo16k = f->pair_halo16k;
                fa.parts16 = 1;
                fa.hperm16k = f->d_hperm16k_pair;
                fa.wtab16k = f->d_wtab16k;
                fa.cus16k = f->cus;
            }
            SDR_HIP(launch_fir_fft(fa, false, f->pair_grid, stream));
        } else {
            fa.hperm = f->d_hperm;
            fa.n = (long long)n;
            fa.in_stride = (long long)in_stride;
            fa.out_stride = (long long)out_stride;
            fa.halo = f->fft_halo;
            fa.hist_len = f->hist_len;
            fa.parts = f->fft_parts;
            if (f->use16k && f->d_hperm16k) {
                fa.use16k = f->use16k;
                fa.halo16k = f->halo16k;
                fa.parts16 = f->parts16;
                fa.hperm16k = f->d_hperm16k;
                fa.wtab16k = f->d_wtab16k;
                fa.cus16k = f->cus; // one 1024-lane workgroup per CU
            }
            SDR_HIP(launch_fir_fft(fa, f->kind == SDR_REAL, f->fft_grid, stream));
        }
        return SDR_OK;
    }
    const bool multi = f->channels > 1;
    const int vec_in = ((uintptr_t)d_in % 16 == 0) && (!multi || (in_stride * f->w) % 4 == 0);
    const int vec_out = ((uintptr_t)d_out % 16 == 0) && (!multi || (out_stride * f->w) % 4 == 0);
    if (f->use_mfma) {
        MfmaArgs m;
        m.in = d_in;
        m.out = d_out;
        m.hist[0] = f->d_hist[0];
        m.hist[1] = f->d_hist[1];
        m.state = f->d_state;
        m.atab = f->d_atab;
        m.n = (long long)n;
        m.in_stride = (long long)in_stride;
        m.out_stride = (long long)out_stride;
        m.hist_len = f->hist_len;
        m.vec_in = vec_in;
        m.vec_out = vec_out;
        SDR_HIP(launch_fir_mfma(m, f->kind == SDR_REAL, f->k->stride, f->mfma_smax, f->channels, stream));
        SDR_HIP(launch_carry_history(d_in, (long long)in_stride, (long long)n, f->d_hist, f->d_state, f->hist_len, f->w,
                                     f->channels, stream));
        return SDR_OK;
    }
    FirArgs a;
    a.in = d_in;
    a.out = d_out;
    a.hist[0] = f->d_hist[0];
    a.hist[1] = f->d_hist[1];
    a.state = f->d_state;
    a.taps = f->d_taps;
    a.n = (long long)n;
    a.in_stride = (long long)in_stride;
    a.out_stride = (long long)out_stride;
    a.nchunks = f->nchunks;
    a.hist_len = f->hist_len;
    a.shift = f->shift;
    a.vec_in = vec_in;
    a.vec_out = vec_out;
    SDR_HIP(launch_fir(*f->k, a, f->channels, stream));
    SDR_HIP(launch_carry_history(d_in, (long long)in_stride, (long long)n, f->d_hist, f->d_state, f->hist_len, f->w,
                                 f->channels, stream));
    return SDR_OK;
}

extern "C" int sdr_fir_process_device(sdr_fir_t* f, const void* d_in, size_t in_stride, size_t n, void* d_out,
                                      size_t out_stride, void* hip_stream) {
    clear_error();
    if (!f) return fail(SDR_ERR_INVALID, "sdr_fir_process_device: NULL handle");
    if (n && (!d_in || !d_out)) return fail(SDR_ERR_INVALID, "sdr_fir_process_device: NULL buffer");
    DeviceGuard guard(f->device);
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    return process_device_impl(f, (const float*)d_in, in_stride, n, (float*)d_out, out_stride, (hipStream_t)hip_stream);
}

static int ensure_slot(sdr_fir* f, sdr_fir::Slot& s, size_t floats) {
    if (s.cap >= floats) return SDR_OK;
    // growing a slot: wait for anything that may still use the old buffers
    SDR_HIP(hipStreamSynchronize(f->stream));
    SDR_HIP(hipStreamSynchronize(f->stream1));
    if (s.d_in) (void)hipFree(s.d_in);
    if (s.d_out) (void)hipFree(s.d_out);
    s.d_in = s.d_out = nullptr;
    s.cap = 0;
    size_t cap = std::max<size_t>(floats, 1 << 16);
    cap = (cap + 1023) & ~(size_t)1023;
    SDR_HIP(hipMalloc(&s.d_in, cap * sizeof(float)));
    SDR_HIP(hipMalloc(&s.d_out, cap * sizeof(float)));
    s.cap = cap;
    return SDR_OK;
}

extern "C" int sdr_fir_submit(sdr_fir_t* f, int slot, const void* in, size_t n, void* out) {
    clear_error();
    if (!f) return fail(SDR_ERR_INVALID, "sdr_fir_submit: NULL handle");
    if (slot < 0 || slot > 1) return fail(SDR_ERR_INVALID, "sdr_fir_submit: slot must be 0 or 1");
    if (n && (!in || !out)) return fail(SDR_ERR_INVALID, "sdr_fir_submit: NULL buffer");
    DeviceGuard guard(f->device);
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    sdr_fir::Slot& s = f->slot[slot];
    if (s.pending) {
        SDR_HIP(hipEventSynchronize(s.done));
        s.pending = false;
    }
    if (n == 0) return SDR_OK;
    const size_t floats = n * f->channels * f->w;
    int rc = ensure_slot(f, s, floats);
    if (rc != SDR_OK) return rc;
    hipStream_t st = f->slot_stream(slot);
    SDR_HIP(hipMemcpyAsync(s.d_in, in, floats * sizeof(float), hipMemcpyHostToDevice, st));
    if (f->have_filtered) SDR_HIP(hipStreamWaitEvent(st, f->filtered, 0)); // history order across the two streams
    rc = process_device_impl(f, s.d_in, n, n, s.d_out, n, st);
    if (rc != SDR_OK) return rc;
    SDR_HIP(hipEventRecord(f->filtered, st));
    f->have_filtered = true;
    SDR_HIP(hipMemcpyAsync(out, s.d_out, floats * sizeof(float), hipMemcpyDeviceToHost, st));
    SDR_HIP(hipEventRecord(s.done, st));
    s.pending = true;
    return SDR_OK;
}

extern "C" int sdr_fir_wait(sdr_fir_t* f, int slot) {
    clear_error();
    if (!f) return fail(SDR_ERR_INVALID, "sdr_fir_wait: NULL handle");
    if (slot < 0 || slot > 1) return fail(SDR_ERR_INVALID, "sdr_fir_wait: slot must be 0 or 1");
    sdr_fir::Slot& s = f->slot[slot];
    if (!s.pending) return SDR_OK;
    DeviceGuard guard(f->device);
    SDR_HIP(hipEventSynchronize(s.done));
    s.pending = false;
    return SDR_OK;
}

extern "C" int sdr_fir_process(sdr_fir_t* f, const void* in, size_t n, void* out) {
    int rc = sdr_fir_submit(f, 0, in, n, out);
    if (rc != SDR_OK) return rc;
    return sdr_fir_wait(f, 0);
}

extern "C" int sdr_fir_reset(sdr_fir_t* f) {
    clear_error();
    if (!f) return fail(SDR_ERR_INVALID, "sdr_fir_reset: NULL handle");
    DeviceGuard guard(f->device);
    // The handle's own streams are drained; work the caller queued on a stream of their own
    // (sdr_fir_process_device) is theirs to order before a reset, like before a destroy.  The clears go
    // through the handle's stream and are waited for, so the next call on any stream sees them.
    SDR_HIP(hipStreamSynchronize(f->stream));
    SDR_HIP(hipStreamSynchronize(f->stream1));
    f->have_filtered = false;
    f->onset_seen = 0;
    const size_t hist_bytes = f->channels * (size_t)f->hist_len * f->w * sizeof(float);
    SDR_HIP(hipMemsetAsync(f->d_hist[0], 0, hist_bytes, f->stream));
    SDR_HIP(hipMemsetAsync(f->d_hist[1], 0, hist_bytes, f->stream));
    SDR_HIP(hipMemsetAsync(f->d_state, 0, 256, f->stream));
    if (f->d_tickets) SDR_HIP(hipMemsetAsync(f->d_tickets, 0, FFT_TICKET_WORDS * sizeof(unsigned), f->stream));
    SDR_HIP(hipStreamSynchronize(f->stream));
    return SDR_OK;
}

extern "C" size_t sdr_fir_ntaps(const sdr_fir_t* f) { return f ? f->ntaps : 0; }
extern "C" int sdr_fir_tap_stride(const sdr_fir_t* f) { return f ? (f->use_fft ? (f->fft_pair ? 2 : 1) : f->k->stride) : 0; }
extern "C" const char* sdr_fir_kernel_name(const sdr_fir_t* f) {
    if (!f) return "";
    if (f->use_mfma) return fir_mfma_kernel_name(f->kind == SDR_REAL, f->k->stride);
    if (f->use_fft && f->use16k && !f->fft_pair) {
        if (f->kind == SDR_REAL) return f->use16k == 2 ? "fir_fft16k<REAL>" : "fir_fft16k<REAL> (short calls: fir_fft4096<REAL>)";
        return f->use16k == 2 ? "fir_fft16k<COMPLEX>" : "fir_fft16k<COMPLEX> (short calls: fir_fft4096<COMPLEX>)";
    }
    if (f->use_fft && f->fft_pair && f->use16k && f->d_hperm16k_pair)
        return f->use16k == 2 ? "fir_fft16k<REAL as complex pairs>" : "fir_fft16k<REAL as complex pairs> (short calls: fir_fft4096<REAL as complex pairs>)";
    if (f->use_fft && f->fft_pair) return "fir_fft4096<REAL as complex pairs>";
    return f->use_fft ? fir_fft_kernel_name(f->kind == SDR_REAL) : f->k->name;
}

extern "C" void* sdr_host_alloc(size_t bytes) {
    clear_error();
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        hip_fail(e, "hipHostMalloc");
        return nullptr;
    }
    return p;
}

extern "C" void sdr_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}


// ---- generator -----------------------------------------------------------------

struct sdr_gen {
    int device = 0;
    int waveform = 0;
    int kind = SDR_REAL;
    float amp = 1.0f, cps = 0.0f, phi = 0.0f, phi_incr = 0.0f;
    size_t block_size = 0;
    hipStream_t stream = nullptr;
    // ring of per-call phase tables (pinned host + device copy): the host fills table
    // k+1 while the GPU still reads table k, so back-to-back calls never stall
    static constexpr int kRing = 4;
    struct Table {
        hipEvent_t free = nullptr; // recorded after the kernel that reads the table
        hipEvent_t ready = nullptr; // recorded after the table's upload on the generator's own stream
        float* h = nullptr;        // pinned
        float* d = nullptr;
        size_t cap = 0;
    } table[kRing];
    int next_table = 0;
    float* d_out = nullptr; // staging for the host entry point
    size_t out_cap = 0;
    // Closed-form phase chain (phase_chain_is_exact): the generator's position lives in DEVICE memory and the
    // kernels move it on themselves -- no per-call phase table, nothing of the position on the host, so a captured
    // graph of one generating call replays as the next call.  dev_phase: d_state->phase, not `phi`, is the truth.
    GenDevState* d_state = nullptr;
    bool dev_phase = false;
    unsigned pincr = 0;          // phi_incr in units of 2^-23 (dev_phase)
    hipEvent_t last = nullptr;   // behind the latest generating launch (what to_host waits for)
    bool have_last = false;
    // complex square / triangle / sawtooth (blocks/gen.cpp:138, 147-164, 231-260): the
    // waveform plus +-j times its Hilbert transform from a real FIR
    size_t hilbert_taps = 127;
    bool preroll = true;      // hilb_delay samples through the filter before the first block
    bool prerolled = false;
    sdr_fir_t* quad = nullptr;
    float* d_quad[3] = {nullptr, nullptr, nullptr}; // waveform, advanced waveform, filtered
    size_t quad_cap = 0;                            // samples each
    bool quadrature() const { return kind == SDR_COMPLEX && waveform > SDR_WAVE_SINE; }
};

static inline float fractf_host(float x) { return x - floorf(x); }

static bool phase_chain_is_exact(float phi, float incr, unsigned* P, unsigned* I);

// Move the generator's position to the device when its phase chain has a closed form (not for the quadrature
// waveforms, whose second table is built on the host), or back to the host (for a change of frequency, an
// advance, or a chain that stops being exact).  to_host waits for the latest generating launch the library
// itself queued; replays of a captured graph are the caller's to finish first.
static int gen_to_device(sdr_gen* g) {
    unsigned P, I;
    if (g->dev_phase || g->quadrature() || !phase_chain_is_exact(g->phi, g->phi_incr, &P, &I)) return SDR_OK;
    GenDevState st = {P, 0u};
    SDR_HIP(hipMemcpy(g->d_state, &st, sizeof st, hipMemcpyHostToDevice));
    g->pincr = I;
    g->dev_phase = true;
    return SDR_OK;
}
static int gen_to_host(sdr_gen* g) {
    if (!g->dev_phase) return SDR_OK;
    if (g->have_last) SDR_HIP(hipEventSynchronize(g->last));
    GenDevState st;
    SDR_HIP(hipMemcpy(&st, g->d_state, sizeof st, hipMemcpyDeviceToHost));
    g->phi = (float)(st.phase & 0x7fffffu) * (1.0f / 8388608.0f);
    g->dev_phase = false;
    return SDR_OK;
}
// after a generating launch on `stream`
static int gen_mark_launch(sdr_gen* g, hipStream_t stream) {
    if (stream_is_capturing(stream)) return SDR_OK; // a captured launch runs later, at the caller's command
    SDR_HIP(hipEventRecord(g->last, stream));
    g->have_last = true;
    return SDR_OK;
}

extern "C" size_t sdr_optimal_block_size(uint64_t element_size, uint64_t sample_rate) {
    if (element_size == 0) return 0;
    const uint64_t page = 4096, header = 16;
    uint64_t size = std::max<uint64_t>(1, (2 * page - header) / element_size);
    if (sample_rate == 0) return (size_t)size;
    uint64_t a = header, b = element_size;
    while (b) { uint64_t t = a % b; a = b; b = t; }
    const uint64_t step = (header / a * element_size) / element_size; // lcm(header, elt)/elt
    const uint64_t limit = std::max<uint64_t>((page - header) / element_size, step);
    while (size > limit && (size * 1000000000ull) % sample_rate > 0) size -= step;
    return (size_t)size;
}

extern "C" int sdr_gen_create(sdr_gen_t** out, int waveform, int sample_kind, float amplitude,
                              float cycles_per_sample, float phase_degrees, size_t block_size, int device) {
    clear_error();
    if (!out) return fail(SDR_ERR_INVALID, "sdr_gen_create: out is NULL");
    *out = nullptr;
    if (waveform < SDR_WAVE_COSINE || waveform > SDR_WAVE_SAWTOOTH) return fail(SDR_ERR_INVALID, "sdr_gen_create: unknown waveform");
    if (sample_kind != SDR_REAL && sample_kind != SDR_COMPLEX) return fail(SDR_ERR_INVALID, "sdr_gen_create: bad sample_kind");
    if (block_size == 0 || block_size > (1u << 24)) return fail(SDR_ERR_INVALID, "sdr_gen_create: bad block_size");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(SDR_ERR_NO_DEVICE, "sdr_gen_create: no HIP device (libsdr_amd has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(SDR_ERR_INVALID, "sdr_gen_create: device out of range");
    sdr_gen* g = new (std::nothrow) sdr_gen();
    if (!g) return fail(SDR_ERR_NOMEM, "sdr_gen_create: out of memory");
    g->device = device;
    g->waveform = waveform;
    g->kind = sample_kind;
    g->amp = amplitude;
    g->cps = cycles_per_sample;
    g->phi = fractf_host(phase_degrees / 360.0f);                      // gen.cpp:123
    g->phi_incr = fractf_host(cycles_per_sample * (float)block_size);  // gen.cpp:124
    g->block_size = block_size;
    DeviceGuard guard(device);
    hipError_t e;
    e = hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking);
    for (auto& t : g->table) {
        if (e == hipSuccess) e = hipEventCreateWithFlags(&t.free, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&t.ready, hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->last, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc(&g->d_state, 256);
    if (e == hipSuccess) e = hipMemset(g->d_state, 0, 256);
    if (e == hipSuccess && gen_to_device(g) != SDR_OK) e = hipErrorUnknown;
    if (e != hipSuccess) {
        int rc = hip_fail(e, "sdr_gen_create");
        std::string keep = g_last_error;
        sdr_gen_destroy(g);
        g_last_error = keep;
        return rc;
    }
    *out = g;
    return SDR_OK;
}

extern "C" int sdr_gen_set_frequency(sdr_gen_t* g, float cycles_per_sample) {
    clear_error();
    if (!g) return fail(SDR_ERR_INVALID, "sdr_gen_set_frequency: NULL handle");
    DeviceGuard guard(g->device);
    int rc = gen_to_host(g);
    if (rc != SDR_OK) return rc;
    g->cps = cycles_per_sample;
    g->phi_incr = fractf_host(cycles_per_sample * (float)g->block_size);
    return gen_to_device(g);
}

// The per-block phase recurrence phi' = fract(phi + incr) in fp32 is a serial chain (~3 ns per block on the host:
// for 2044-sample blocks that alone caps a generator at ~650 Gsample/s, below what the fused kernel sustains).
// When phi and incr are both whole multiples of 2^-23 below 1, no step of it ever rounds: every such multiple
// below 2 is a float, so phi + incr is exact, and so is the subtraction of 1 -- the chain is
// phi_k = ((P + k I) mod 2^23) / 2^23 in integers, bit for bit (tests/test_host_cpu.py checks the arithmetic).
// incr = fract(cps * B) is such a multiple whenever |cps * B| >= 1 (its ulp is then 2^-23 or coarser), phi after
// the first wrap, or from the start for phases like 0, 45 or 90 degrees.
static bool phase_chain_is_exact(float phi, float incr, unsigned* P, unsigned* I) {
    if (!(phi >= 0.0f && phi < 1.0f && incr >= 0.0f && incr < 1.0f)) return false;
    const float p = phi * 8388608.0f, i = incr * 8388608.0f; // exact scalings
    if (p != floorf(p) || i != floorf(i)) return false;
    *P = (unsigned)p;
    *I = (unsigned)i;
    return true;
}

extern "C" int sdr_gen_advance(sdr_gen_t* g, size_t blocks) {
    clear_error();
    if (!g) return fail(SDR_ERR_INVALID, "sdr_gen_advance: NULL handle");
    if (g->quadrature() && blocks)
        return fail(SDR_ERR_INVALID, "sdr_gen_advance: a quadrature generator carries a filter history that cannot be skipped");
    DeviceGuard guard(g->device);
    int rc = gen_to_host(g);
    if (rc != SDR_OK) return rc;
    float phi = g->phi;
    const float incr = g->phi_incr;
    unsigned P, I;
    if (phase_chain_is_exact(phi, incr, &P, &I)) {
        phi = (float)((P + (unsigned)blocks * I) & 0x7fffffu) * (1.0f / 8388608.0f);
    } else {
        for (size_t b = 0; b < blocks; ++b) phi = fractf_host(phi + incr);
    }
    g->phi = phi;
    return gen_to_device(g);
}

// The next `blocks` per-block start phases (plus one spare entry, read but never used
// by kernels that look one block ahead) into the next table of the ring, uploaded on
// `stream`.  The caller records t->free after the kernels that read the table.
// second: a second table follows at [cap, cap + blocks]: the same phases advanced by
// `second_offset` (one fp32 addition per block: the quadrature path's phi + (cps*hilb_delay)).
static int gen_next_table(sdr_gen* g, size_t blocks, hipStream_t stream, sdr_gen::Table** out,
                          bool second = false, float second_offset = 0.0f) {
    if (stream_is_capturing(stream))
        return fail(SDR_ERR_INVALID, "generator: this call uploads a per-call phase table computed on the host (the reference's "
                                     "sequential fp32 phase recurrence, blocks/gen.cpp:185) and cannot be recorded into a HIP graph");
    sdr_gen::Table& t = g->table[g->next_table];
    g->next_table = (g->next_table + 1) % sdr_gen::kRing;
    SDR_HIP(hipEventSynchronize(t.free)); // the call that used this slot 4 calls ago
    if (blocks + 1 > t.cap) {
        if (t.h) (void)hipHostFree(t.h);
        if (t.d) (void)hipFree(t.d);
        t.h = t.d = nullptr;
        t.cap = 0;
        const size_t cap = std::max<size_t>(blocks + 1, 4096);
        SDR_HIP(hipHostMalloc(&t.h, 2 * cap * sizeof(float), hipHostMallocDefault));
        SDR_HIP(hipMalloc(&t.d, 2 * cap * sizeof(float)));
        t.cap = cap;
    }
    // sequential fp32 phase recurrence, one value per block (gen.cpp:185,192,227)
    // phi and phi_incr lie in [0,1), so their fp32 sum s lies in [0,2) and
    // fract(s) = s - floor(s) is s or s - 1 (exact): no libm call on this serial chain,
    // which the host must finish faster than the GPU consumes a table (one entry per
    // block_size samples).
    float phi = g->phi;
    const float incr = g->phi_incr;
    float* h = t.h;
    unsigned P, I;
    if (phase_chain_is_exact(phi, incr, &P, &I)) {
        // closed form, vectorisable: 262k blocks of a 2^29-sample call take ~30 us instead of 0.85 ms
        for (size_t b = 0; b <= blocks; ++b) h[b] = (float)((P + (unsigned)b * I) & 0x7fffffu) * (1.0f / 8388608.0f);
        phi = h[blocks];
    } else {
        for (size_t b = 0; b < blocks; ++b) {
            h[b] = phi;
            const float s = phi + incr;
            phi = s >= 1.0f ? s - 1.0f : s;
            if (__builtin_expect(phi >= 1.0f, 0)) phi -= 1.0f; // only if a fract() above rounded up to 1.0
        }
        h[blocks] = phi;
    }
    g->phi = phi;
    if (second) {
        for (size_t b = 0; b <= blocks; ++b) h[t.cap + b] = h[b] + second_offset;
        SDR_HIP(hipMemcpyAsync(t.d + t.cap, t.h + t.cap, (blocks + 1) * sizeof(float), hipMemcpyHostToDevice, g->stream));
    }
    // upload on the generator's own stream: the host runs ahead of the GPU, so table k+1
    // crosses PCIe while the kernels of call k are still running on `stream`
    SDR_HIP(hipMemcpyAsync(t.d, t.h, (blocks + 1) * sizeof(float), hipMemcpyHostToDevice, g->stream));
    SDR_HIP(hipEventRecord(t.ready, g->stream));
    SDR_HIP(hipStreamWaitEvent(stream, t.ready, 0));
    *out = &t;
    return SDR_OK;
}

static int gen_quadrature_impl(sdr_gen* g, size_t blocks, float* d_out, hipStream_t stream);

static int gen_device_impl(sdr_gen* g, size_t blocks, float* d_out, hipStream_t stream) {
    if (blocks == 0) return SDR_OK;
    if (g->quadrature()) return gen_quadrature_impl(g, blocks, d_out, stream);
    if (blocks > 0xffffffffull) return fail(SDR_ERR_INVALID, "generator: more than 2^32 - 1 blocks in one call");
    GenArgs a;
    a.out = d_out;
    a.blocks = (long long)blocks;
    a.block_size = (long long)g->block_size;
    a.amp = g->amp;
    a.cps = g->cps;
    a.waveform = g->waveform;
    a.complex_samples = g->kind == SDR_COMPLEX;
    if (g->dev_phase) {
        // closed-form phases from the device's own position: no table, nothing of this call's position on the host
        a.phi = nullptr;
        a.pstate = g->d_state;
        a.pincr = g->pincr;
        SDR_HIP(launch_gen(a, stream));
        return gen_mark_launch(g, stream);
    }
    sdr_gen::Table* t = nullptr;
    int rc = gen_next_table(g, blocks, stream, &t);
    if (rc != SDR_OK) return rc;
    a.pstate = nullptr;
    a.pincr = 0;
    a.phi = t->d;
    SDR_HIP(launch_gen(a, stream));
    SDR_HIP(hipEventRecord(t->free, stream));
    return SDR_OK;
}

// blocks/gen.cpp:231-260.  Per block k, in-block index i:
//   re = A * wave(phi_k + cps*i)
//   im = A * (+-1) * fir(hilb, wave(phi_k + (cps*hilb_delay) + cps*i)),  - for cps < 0
// hilb = real FIR with sdr::hilbert taps whose history runs on from block to block (and from
// the pre-roll, :147-164).  Two generator launches, one pass of the FIR path, one interleave.
static int gen_quadrature_impl(sdr_gen* g, size_t blocks, float* d_out, hipStream_t stream) {
    if (stream_is_capturing(stream))
        return fail(SDR_ERR_INVALID, "generator: the quadrature waveforms (complex square / triangle / sawtooth) build per-call "
                                     "phase tables on the host and cannot be recorded into a HIP graph");
    const size_t n = blocks * g->block_size;
    const size_t hd = (g->hilbert_taps - 1) / 2;
    if (!g->quad) {
        std::vector<float> h(g->hilbert_taps);
        int rc = sdr_hilbert_taps(h.data(), h.size());
        if (rc != SDR_OK) return rc;
        rc = sdr_fir_create(&g->quad, h.data(), h.size(), SDR_REAL, 1, g->device);
        if (rc != SDR_OK) return rc;
    }
    if (n > g->quad_cap || hd > g->quad_cap) {
        SDR_HIP(hipStreamSynchronize(stream));
        for (float*& p : g->d_quad) {
            if (p) (void)hipFree(p);
            p = nullptr;
        }
        g->quad_cap = 0;
        const size_t cap = std::max(n, hd);
        for (float*& p : g->d_quad) SDR_HIP(hipMalloc(&p, cap * sizeof(float)));
        g->quad_cap = cap;
    }
    GenArgs a;
    a.pstate = nullptr; // the quadrature path keeps its position on the host (its second table is built there)
    a.pincr = 0;
    a.amp = 1.0f;
    a.cps = g->cps;
    a.waveform = g->waveform;
    a.complex_samples = 0;
    if (g->preroll && !g->prerolled && hd > 0) {
        // one "block" of hilb_delay samples at the initial phase, filtered and discarded
        const float save_phi = g->phi, save_incr = g->phi_incr;
        sdr_gen::Table* t0 = nullptr;
        g->phi_incr = 0.0f;
        int rc = gen_next_table(g, 1, stream, &t0);
        g->phi = save_phi;
        g->phi_incr = save_incr;
        if (rc != SDR_OK) return rc;
        a.out = g->d_quad[1];
        a.phi = t0->d;
        a.blocks = 1;
        a.block_size = (long long)hd;
        SDR_HIP(launch_gen(a, stream));
        SDR_HIP(hipEventRecord(t0->free, stream));
        rc = sdr_fir_process_device(g->quad, g->d_quad[1], hd, hd, g->d_quad[2], hd, stream);
        if (rc != SDR_OK) return rc;
    }
    g->prerolled = true;
    sdr_gen::Table* t = nullptr;
    const float adv = g->cps * (float)hd; // (cycles_per_sample*hilb_delay), gen.cpp:236
    int rc = gen_next_table(g, blocks, stream, &t, true, adv);
    if (rc != SDR_OK) return rc;
    a.blocks = (long long)blocks;
    a.block_size = (long long)g->block_size;
    a.out = g->d_quad[0];
    a.phi = t->d;
    a.amp = g->amp;
    SDR_HIP(launch_gen(a, stream));
    a.out = g->d_quad[1];
    a.phi = t->d + t->cap;
    a.amp = 1.0f;
    SDR_HIP(launch_gen(a, stream));
    SDR_HIP(hipEventRecord(t->free, stream));
    rc = sdr_fir_process_device(g->quad, g->d_quad[1], n, n, g->d_quad[2], n, stream);
    if (rc != SDR_OK) return rc;
    SDR_HIP(launch_gen_interleave(d_out, g->d_quad[0], g->d_quad[2], g->cps < 0 ? -g->amp : g->amp, (long long)n, stream));
    return SDR_OK;
}

extern "C" int sdr_gen_set_hilbert(sdr_gen_t* g, size_t hilbert_taps, int preroll) {
    clear_error();
    if (!g) return fail(SDR_ERR_INVALID, "sdr_gen_set_hilbert: NULL handle");
    if (g->prerolled || g->quad) return fail(SDR_ERR_INVALID, "sdr_gen_set_hilbert: the generator has already produced samples");
    if (hilbert_taps < 5 || hilbert_taps > SDR_MAX_TAPS) return fail(SDR_ERR_INVALID, "sdr_gen_set_hilbert: taps must be 5..131072");
    g->hilbert_taps = hilbert_taps;
    g->preroll = preroll != 0;
    return SDR_OK;
}

size_t sdr_amd::gen_block_size(const sdr_gen_t* g) { return g ? g->block_size : 0; }

int sdr_amd::fir_generated_supported(const sdr_fir_t* f, const sdr_gen_t* g) {
    if (!f || !g) return fail(SDR_ERR_INVALID, "fused generator: NULL handle");
    if (f->device != g->device) return fail(SDR_ERR_INVALID, "fused generator: generator and filter live on different devices");
    if (g->kind != SDR_REAL || f->kind != SDR_REAL || f->channels != 1)
        return fail(SDR_ERR_INVALID, "fused generator: needs a real generator and a real one-channel filter");
    if (!f->use_fft) return fail(SDR_ERR_INVALID, "fused generator: the filter is too short for the overlap-save kernels");
    if (g->block_size < 256 || g->block_size > (1u << 23))
        return fail(SDR_ERR_INVALID, "fused generator: block size must be 256..2^23 samples");
    return SDR_OK;
}

int sdr_amd::fir_process_generated(sdr_fir_t* f, sdr_gen_t* g, size_t blocks, float* d_out, hipStream_t stream) {
    int rc = fir_generated_supported(f, g);
    if (rc != SDR_OK) return rc;
    if (blocks == 0) return SDR_OK;
    DeviceGuard guard(f->device);
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    if (blocks > 0xffffffffull) return fail(SDR_ERR_INVALID, "fused generator: more than 2^32 - 1 blocks in one call");
    sdr_gen::Table* t = nullptr;
    const float incr = g->phi_incr; // of this call's blocks
    if (!g->dev_phase) {
        rc = gen_next_table(g, blocks, stream, &t);
        if (rc != SDR_OK) return rc;
    }
    FftArgs fa = {};
    fa.in = nullptr;
    fa.out = d_out;
    fa.hperm = f->d_hperm;
    fa.n = (long long)(blocks * g->block_size);
    fa.in_stride = fa.out_stride = fa.n;
    fa.channels = 1;
    fa.halo = f->fft_halo;
    fa.hist_len = f->hist_len;
    fa.parts = f->fft_parts;
    fa.gen.phi = t ? t->d : nullptr;
    fa.gen.pstate = g->dev_phase ? g->d_state : nullptr;
    fa.gen.pincr = g->pincr;
    fa.gen.nblocks = (unsigned)blocks;
    fa.gen.phi_incr = incr;
    fa.gen.block_size = (long long)g->block_size;
    fa.gen.inv_block = 1.0 / (double)g->block_size;
    fa.gen.amp = g->amp;
    fa.gen.cps = g->cps;
    fa.gen.waveform = g->waveform;
    fa.gen.enabled = 1;
    fft_common_args(f, fa);
    f->onset_seen = std::min(f->ntaps, f->onset_seen + (size_t)fa.n);
    SDR_HIP(launch_fir_fft(fa, true, f->fft_grid, stream));
    if (t) SDR_HIP(hipEventRecord(t->free, stream));
    else return gen_mark_launch(g, stream);
    return SDR_OK;
}

extern "C" int sdr_gen_generate_device(sdr_gen_t* g, size_t blocks, void* d_out, void* hip_stream) {
    clear_error();
    if (!g) return fail(SDR_ERR_INVALID, "sdr_gen_generate_device: NULL handle");
    if (blocks && !d_out) return fail(SDR_ERR_INVALID, "sdr_gen_generate_device: NULL buffer");
    DeviceGuard guard(g->device);
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    return gen_device_impl(g, blocks, (float*)d_out, (hipStream_t)hip_stream);
}

extern "C" int sdr_gen_generate(sdr_gen_t* g, size_t blocks, void* out) {
    clear_error();
    if (!g) return fail(SDR_ERR_INVALID, "sdr_gen_generate: NULL handle");
    if (blocks == 0) return SDR_OK;
    if (!out) return fail(SDR_ERR_INVALID, "sdr_gen_generate: NULL buffer");
    DeviceGuard guard(g->device);
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    const size_t floats = blocks * g->block_size * (g->kind == SDR_COMPLEX ? 2 : 1);
    if (floats > g->out_cap) {
        SDR_HIP(hipStreamSynchronize(g->stream));
        if (g->d_out) (void)hipFree(g->d_out);
        g->d_out = nullptr;
        g->out_cap = 0;
        SDR_HIP(hipMalloc(&g->d_out, floats * sizeof(float)));
        g->out_cap = floats;
    }
    int rc = gen_device_impl(g, blocks, g->d_out, g->stream);
    if (rc != SDR_OK) return rc;
    SDR_HIP(hipMemcpyAsync(out, g->d_out, floats * sizeof(float), hipMemcpyDeviceToHost, g->stream));
    SDR_HIP(hipStreamSynchronize(g->stream));
    return SDR_OK;
}

extern "C" void sdr_gen_destroy(sdr_gen_t* g) {
    if (!g) return;
    DeviceGuard guard(g->device);
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    for (auto& t : g->table) {
        if (t.h) (void)hipHostFree(t.h);
        if (t.d) (void)hipFree(t.d);
        if (t.free) (void)hipEventDestroy(t.free);
        if (t.ready) (void)hipEventDestroy(t.ready);
    }
    if (g->d_out) (void)hipFree(g->d_out);
    if (g->d_state) (void)hipFree(g->d_state);
    if (g->last) (void)hipEventDestroy(g->last);
    for (float* p : g->d_quad)
        if (p) (void)hipFree(p);
    if (g->quad) sdr_fir_destroy(g->quad);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
}
