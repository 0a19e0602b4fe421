// fir_kernels.h -- internal interface between the C ABI (sdr_capi.hip) and the
// gfx950 kernels (fir_kernels.hip).  Not installed; the public surface is
// include/sdr_amd.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "gen_device.h"

namespace sdr_amd {

// Where a handle's stream stands, in DEVICE memory: every launch reads it, the last workgroup of a
// call's last launch to finish moves it on.  Nothing about the stream's position is host state, so
// a captured graph of one call replays as the next call of the stream, and replays mix freely with
// direct calls (the reference's kfr::fir_state lives across calls, libs/kfr/dsp/fir.hpp:57-68).
struct FirDevState {
    unsigned cur;  // which of the two history buffers holds the last hist_len samples of the stream
    unsigned done; // workgroups of the running launch that have finished (back to 0 when it ends)
};

// The workgroup that finishes last gets true (thread 0 only; call from every workgroup of the launch,
// `nblk` of them, behind a barrier that every wave reaches after its last read of *st).
__device__ __forceinline__ bool last_workgroup_done(FirDevState* st, unsigned nblk) {
    if (threadIdx.x != 0) return false;
    // Relaxed on purpose.  What must precede the flip is every workgroup's READ of the state, and that is
    // program order (the value was used before the barrier); what the winner writes is first read by the NEXT
    // launch on the stream, and the kernel boundary orders that.  An acq_rel atomic here costs every workgroup
    // an L2 write-back (buffer_wbl2) at its end: C2 0.771 -> 0.798 ms, profiles/r03_device_state_ab.txt.
    return __hip_atomic_fetch_add(&st->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == nblk;
}

// the history buffer state->cur names / the other one, without indexing the kernel-argument struct by a run-time
// value (that keeps the compiler from holding the arguments in registers: it reloads them at every use)
template <class P>
__device__ __forceinline__ P pick(P const (&h)[2], unsigned cur) { return cur ? h[1] : h[0]; }

struct FirArgs {
    const float* in;      // channel 0, sample 0
    float* out;           // channel 0, output 0
    const float* hist[2]; // [channels][hist_len] samples preceding `in`: the buffer state->cur names
    const FirDevState* state; // NULL: hist[0]
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

// overlap-save FFT path (fir_fft.hip)
struct FftArgs {
    const float* in;
    float* out;
    float* hist[2];       // [channels][hist_len] each: state->cur names the one that precedes `in`; the launch
                          // fills the other and flips state->cur when its last workgroup is done
    FirDevState* state;
    const float* hperm;   // parts x 4096 complex: H_p[k]/N stored at [p*4096 + k2*256 + 16*k0 + k1], k = k0 + 16 k1 + 256 k2
    const float* wtab;    // 4096 + 256 complex: W_4096^(tid k) at [256 k + tid], W_256^(n k) at [4096 + 16 k + n]
    long long n;
    long long in_stride;
    long long out_stride;
    long long channels;
    int halo;             // old samples per tile: >= ntaps-1; 64 * {1,2,4,8,16,32}
    int hist_len;         // >= halo
    int parts;            // filter partitions of 2048 taps (1 = plain overlap-save)
#ifdef SDR_AMD_DIAG       // diagnostic builds only (make diag, scripts/exp)
    int ablate;           // 1 = skip the arithmetic (memory pattern only)
    int contig;           // every workgroup walks a contiguous run of tiles instead of a grid-strided set
#endif
    long long t_lo, t_hi; // set by launch_fir_fft: tiles [t_lo, t_hi) of each channel are interior tiles
    int grid_carry, grid_edge; // set by launch_fir_fft: leading workgroups that carry the history / take edge tiles
    GenSrc gen;           // gen.enabled: samples 0..n-1 are synthesised (real, one channel), `in` is unused
    // Ordered tile hand-out (fir_fft.hip, interior_tiles): FFT_TICKETS counters, FFT_TICKET_STRIDE words
    // apart, all zero when the launch starts; the call's last workgroup to finish zeroes them again
    // (launches on one handle are stream-ordered).  NULL: tiles are grid-strided.
    unsigned* tickets;
    int resident;         // workgroups of the single-partition kernels resident on the device at once (0: unknown)
    // 16384-point kernel (fir_fft16k.hip; more than 2049 taps): its spectrum (parts16 of them beyond 8193 taps, in that kernel's lane
    // order) and twiddles; use16k: the interior of every channel runs on it in tiles of 16384 - halo16k outputs
    // (launch_fir_fft16k sets big_first / big_tiles / grid_edge / grid_carry: one launch, edge and carry workgroups included)
    const float* hperm16k;
    const float* wtab16k;
    int use16k, halo16k, cus16k;
    int parts16;          // > 1: more than 8193 taps, partitions of 8192 taps (hperm16k holds parts16 spectra back to back)
    long long big_first, big_tiles;
};
constexpr size_t FFT16K_TWIDDLE_COMPLEX = 16384 + 1024 + 64;
hipError_t launch_fir_fft16k(const FftArgs& a, bool real, int cus, hipStream_t stream);
void fir_fft16k_tiles(long long n, int halo16k, bool real, int parts16, long long* per_tile, long long* first, long long* interior);
void fir_fft16k_permute_spectrum(const float* h_natural, float* hperm); // 16384 complex
void fir_fft16k_fill_twiddles(const double* cs, const double* sn, float* wtab);
#ifndef SDR_FFT_TICKETS
// measured (profiles/r03_ticket_counters.txt): 1 counter serialises (C2 0.858 ms), 2 / 4 / 6: 0.776 / 0.779 / 0.779, 3 / 5 / 7 / 9: 0.765 /
// 0.769 / 0.775 / 0.777: odd counts beat even ones, three is best (and the counter must be bid % 3: (bid >> 3) % 3 loses 2.7 %, two
// counters chosen by bid >> 3 or bid >> 8 lose 1 % / 11 %).  With bid % 3 a workgroup's tiles all have one residue mod 3 (the grid, 768, is a
// multiple of 3) and every XCD (bid % 8) holds an equal mix of the three sequences; the mechanism behind the odd / even split was not isolated.
#define SDR_FFT_TICKETS 3
#endif
constexpr int FFT_TICKETS = SDR_FFT_TICKETS;
constexpr int FFT_TICKET_STRIDE = 64;
constexpr size_t FFT_TICKET_WORDS = FFT_TICKETS * FFT_TICKET_STRIDE;
hipError_t launch_fir_fft(const FftArgs& a, bool real, int grid_blocks, hipStream_t stream);
const char* fir_fft_kernel_name(bool real);
void fir_fft_plan(size_t ntaps, int* halo, int* parts);

// kernel families a handle can be pinned to (internal; the public create chooses by tap count)
// FFT4K: overlap-save with 4096-point tiles only (tests hold both tile sizes to parity where AUTO picks 16384)
// DIRECT_FORMS: the public SDR_NUMERICS_PER_OUTPUT -- no overlap-save; matrix-core kernel where AUTO would pick it
enum { SDR_PATH_AUTO = -1, SDR_PATH_DIRECT = 0, SDR_PATH_FFT = 1, SDR_PATH_MFMA = 2, SDR_PATH_FFT4K = 3, SDR_PATH_FFT16K = 4,
       SDR_PATH_DIRECT_FORMS = 5 };

// direct-form plan of a tap vector (sdr_capi.hip: plan_direct)
struct DirectPlan {
    const KernelInfo* k = nullptr;
    int stride = 1, shift = 0, R = 8, nchunks = 2, hist_len = 0;
    size_t kept = 0;        // taps the kernel applies (every stride-th)
    size_t taps_floats = 0; // compacted taps, zero-padded: nchunks * R / stride
};
int plan_direct(const float* taps, size_t ntaps, int sample_kind, DirectPlan* p);
void plan_direct_fill(const float* taps, const DirectPlan& p, float* g);

// matrix-core direct form (fir_mfma.hip)
struct MfmaArgs {
    const float* in;
    float* out;
    const float* hist[2]; // [channels][hist_len]: the buffer state->cur names
    const FirDevState* state;
    const float* atab;   // smax x 64: the taps operand of every step, per lane
    long long n;         // samples per channel
    long long in_stride;
    long long out_stride;
    int hist_len;        // >= 4 (smax - 4) * stride, multiple of 4
    int vec_in, vec_out; // 16-byte aligned input + history / output
};
int mfma_steps_for(size_t dense_taps);                 // 0: too many taps for this kernel
size_t mfma_hist_samples(int smax, int stride);
void mfma_fill_atab(const float* g, size_t ntaps, int smax, float* atab);
hipError_t launch_fir_mfma(const MfmaArgs& a, bool real, int stride, int smax, size_t channels, hipStream_t stream);
const char* fir_mfma_kernel_name(bool real, int stride);

const KernelInfo* find_kernel(int R, int stride, bool real);
size_t kernel_lds_bytes(const KernelInfo& k, int nchunks);
hipError_t prepare_kernel(const KernelInfo& k, int nchunks);
hipError_t launch_fir(const KernelInfo& k, const FirArgs& a, size_t channels, hipStream_t stream);
// the last hist_len samples of (history state->cur | in) into the other history buffer, then state->cur flips
hipError_t launch_carry_history(const float* in, long long in_stride, long long n, float* const hist[2],
                                FirDevState* state, int hist_len, int w, size_t channels, hipStream_t stream);

} // namespace sdr_amd
