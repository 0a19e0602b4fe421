/* sdr_amd.h -- C ABI of libsdr_amd.so: the MI355X (gfx950) implementation of
 * fbbdev/sdr's sample-stream FIR hot path.
 *
 * The reference has no FFI on this path: kfr::fir is a header-only template
 * inlined into each block's main().  These entry points are what a binding for
 * the path would bind; each cites the reference interface it replaces.  Plain C
 * types only, integer status returns (0 = success, negative = error; text via
 * sdr_last_error()), no exceptions cross the boundary.  One caller thread per
 * handle.
 *
 * Sample layout (include/sdr/signal.hpp:36-37 of the reference): real samples
 * are float32, complex samples are interleaved {float re, float im}.  Taps are
 * real float32 (kfr::fir_state<float, U>).
 */
#ifndef SDR_AMD_H
#define SDR_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    SDR_OK = 0,
    SDR_ERR_INVALID = -1,   /* bad argument */
    SDR_ERR_NO_DEVICE = -2, /* no usable HIP device: there is NO CPU fallback */
    SDR_ERR_HIP = -3,       /* a HIP runtime call failed */
    SDR_ERR_NOMEM = -4
};

enum { SDR_REAL = 0, SDR_COMPLEX = 1 };
enum { SDR_LOWPASS = 0, SDR_HIGHPASS = 1, SDR_BANDPASS = 2, SDR_BANDSTOP = 3 };

typedef struct sdr_fir sdr_fir_t;

/* Library / device ------------------------------------------------------- */
const char* sdr_version(void);
/* Text of the last error on the calling thread ("" if none). */
const char* sdr_last_error(void);
/* Number of HIP devices visible (0 when none; never fails). */
int sdr_device_count(void);

/* Tap design (host code) -------------------------------------------------- */
/* Replaces sdr::hilbert<float>(n)  (reference include/sdr/hilbert.hpp:33-81,
 * called at blocks/hilbert.cpp:45 and blocks/gen.cpp:138).  n >= 5: for
 * n <= 4 the reference's design degenerates to 0/0 (all-NaN taps). */
int sdr_hilbert_taps(float* taps, size_t n);
/* Replaces kfr::fir_lowpass/highpass/bandpass/bandstop with
 * kfr::window_blackman (alpha 0.16, symmetric) and normalize=true
 * (reference libs/kfr/dsp/fir_design.hpp:35-116, window.hpp:290-311).
 * Frequencies are normalised (Hz / sample_rate); f2 only for band kinds. */
int sdr_fir_design(int kind, float* taps, size_t n, float f1, float f2);
/* The same designs under another of kfr's windows (reference libs/kfr/dsp/window.hpp:
 * :120-139 rectangular, :204-223 hann, :247-267 hamming alpha 0.54, :290-311 blackman
 * alpha 0.16, :314-334 blackman_harris, :336-358 kaiser with window_param = beta over
 * kfr::modzerobessel, libs/kfr/base/modzerobessel.hpp:42-94; all symmetric).  window_param
 * is read by the Kaiser window only. */
enum {
    SDR_WINDOW_BLACKMAN = 0,
    SDR_WINDOW_RECTANGULAR = 1,
    SDR_WINDOW_HANN = 2,
    SDR_WINDOW_HAMMING = 3,
    SDR_WINDOW_BLACKMAN_HARRIS = 4,
    SDR_WINDOW_KAISER = 5
};
int sdr_fir_design_window(int kind, int window, float window_param, float* taps, size_t n, float f1, float f2);

/* Longest filter: 64 partitions of 2048 taps in the partitioned fast convolution. */
#define SDR_MAX_TAPS 131072

/* Streaming FIR ----------------------------------------------------------- */
/* Replaces the kfr::fir_state<T,U> constructor (reference
 * libs/kfr/dsp/fir.hpp:57-68; blocks/hilbert.cpp:45, blocks/gen.cpp:138):
 * `ntaps` real taps h[0..ntaps), zeroed history, `channels` independent
 * streams sharing the taps, samples SDR_REAL or SDR_COMPLEX, on HIP device
 * `device`.  The tap vector is analysed once: all-zero odd (or even) taps --
 * the Hilbert structure -- select a kernel that skips them. */
int sdr_fir_create(sdr_fir_t** out, const float* taps, size_t ntaps, int sample_kind,
                   size_t channels, int device);

/* The same constructor with the handle's NUMERICS chosen by the caller.  kfr::fir is a direct
 * form (reference libs/kfr/dsp/fir.hpp:129-146: one dot product per output), so each output is
 * rounded relative to itself and a NaN / Inf input sample reaches the next `ntaps` outputs and no
 * other.  SDR_NUMERICS_FAST (what sdr_fir_create uses) picks the quickest kernel family per
 * filter; from 45 dense taps (99 of an every-other-tap design) up that is overlap-save, whose rounding error is relative to a 4096-
 * or 16384-point tile (~3e-7 of the TILE's RMS: weak stretches next to strong ones see the strong
 * ones' noise floor) and which loses the whole tile a non-finite sample sits in.
 * SDR_NUMERICS_PER_OUTPUT keeps the handle on the direct-form kernels (VALU or fp32 matrix
 * cores) whatever the tap count, with the reference's character: error relative to each output,
 * non-finite samples confined to the ntaps outputs they reach (plus the kernels' zero padding,
 * at most 144 neighbours) -- at the direct form's cost: hilbert's 127 taps 0.50 ms instead of
 * 0.39 per 2^28 real samples, 127 dense taps 1.5 ms instead of 0.77 per 2^28 complex samples.
 * Up to 14336 dense taps (the direct form's LDS window); longer filters are refused
 * (SDR_ERR_INVALID).  The blocks take it from the environment: SDR_AMD_NUMERICS=per-output. */
enum { SDR_NUMERICS_FAST = 0, SDR_NUMERICS_PER_OUTPUT = 1 };
int sdr_fir_create_numerics(sdr_fir_t** out, const float* taps, size_t ntaps, int sample_kind,
                            size_t channels, int device, int numerics);

/* Replaces `out = kfr::fir(state, in)` (reference libs/kfr/dsp/fir.hpp:167-171,
 * 119-148; call sites blocks/hilbert.cpp:72-73, blocks/gen.cpp:235) for buffers
 * ALREADY IN DEVICE MEMORY:  y[c][i] = sum_j h[j] x[c][i-j], i in [0,n), with
 * x[c][<0] taken from the history carried by the handle (zeros after create /
 * reset).  Channel c starts at in + c*in_stride samples (out likewise).  Exactly
 * n outputs per channel; afterwards the history holds the last ntaps-1 inputs,
 * so results do not depend on how a stream is cut into calls.  Asynchronous on
 * `hip_stream` (a hipStream_t, NULL = the default stream); in and out must not
 * overlap.  Allocates nothing, synchronises nothing, and keeps no position of the
 * stream on the host: which history buffer is current (and the kernels' tile
 * counters) is device state that the kernels read and move on themselves, so a call
 * recorded into a HIP graph replays as the stream's NEXT call, any number of times,
 * and replays mix with ordinary calls (launches on one handle must be stream-ordered).
 * While a stream is still shorter than its filter, calls run the direct form (each
 * output rounded relative to itself, like kfr::fir); a captured call counts as past
 * that onset. */
int sdr_fir_process_device(sdr_fir_t* f, const void* d_in, size_t in_stride, size_t n,
                           void* d_out, size_t out_stride, void* hip_stream);

/* Same contract for HOST buffers (any host memory; pinned memory from
 * sdr_host_alloc makes the copies asynchronous).  Channels are contiguous:
 * channel c at in + c*n samples.  Copies H2D, filters, copies D2H on the
 * handle's own stream and returns when `out` is complete. */
int sdr_fir_process(sdr_fir_t* f, const void* in, size_t n, void* out);

/* Pipelined form of sdr_fir_process for callers that overlap pipe I/O with
 * the GPU (the blocks' main loops): submit queues H2D + filter + D2H for
 * staging slot `slot` (0 or 1) and returns at once; wait blocks until that
 * slot's `out` is complete.  Buffers must stay valid and (for overlap) be
 * pinned until the matching wait.  Submissions execute in order, so the
 * history chains exactly as with sdr_fir_process. */
int sdr_fir_submit(sdr_fir_t* f, int slot, const void* in, size_t n, void* out);
int sdr_fir_wait(sdr_fir_t* f, int slot);

/* No counterpart in the reference: the same filter as sdr_fir_create + one
 * sdr_fir_process_device on a fresh handle (zero history, one channel: y[i] = sum_j h[j] x[i-j],
 * x[<0] = 0), as ONE stateless call that allocates nothing -- no hipMalloc / hipFree, no host
 * heap on the success path -- for callers that cannot allocate: code under HIP graph capture,
 * tight serving loops, harnesses that forbid it.  The taps travel as a kernel argument; the
 * compacted taps and the (zero) history the kernels read live in a fixed 8 KiB device array
 * inside the library (one copy per GPU, held until the process ends).  Limits, refused with
 * SDR_ERR_INVALID: 1..SDR_NOALLOC_MAX_TAPS taps (the direct-form kernels),
 * n <= SDR_NOALLOC_MAX_SAMPLES.  d_in / d_out are DEVICE memory on `device`, must not overlap;
 * asynchronous on hip_stream.  Calls on one GPU share the array: back-to-back calls on one
 * stream are fine, calls that overlap in time on one GPU (two streams or threads) race. */
#define SDR_NOALLOC_MAX_TAPS 255
#define SDR_NOALLOC_MAX_SAMPLES ((size_t)1 << 36)
#define SDR_NOALLOC_SCRATCH_BYTES 8192
int sdr_fir_apply_device_noalloc(const float* taps, size_t ntaps, int sample_kind, const void* d_in,
                                 size_t n, void* d_out, int device, void* hip_stream);

/* Replaces kfr::filter_fir::reset (reference libs/kfr/dsp/fir.hpp:195-199):
 * zero the history of every channel. */
/* (Drains the handle's own streams; work the caller queued on a stream of their own is
 * theirs to finish first, as before sdr_fir_destroy.) */
int sdr_fir_reset(sdr_fir_t* f);
void sdr_fir_destroy(sdr_fir_t* f);

/* Introspection (for tests / benchmarks). */
size_t sdr_fir_ntaps(const sdr_fir_t* f);
/* 1 = dense kernel, 2 = every-other-tap (Hilbert) kernel. */
int sdr_fir_tap_stride(const sdr_fir_t* f);
/* Name of the HIP kernel a process call launches (for profiles). */
const char* sdr_fir_kernel_name(const sdr_fir_t* f);

/* Fused FIR chain ----------------------------------------------------------- */
typedef struct sdr_chain sdr_chain_t;

/* Replaces a pipeline of FIR blocks on one stream, e.g. `hilbert | stream-filter
 * fir=lowpass` (reference: one kfr::fir per process, blocks/hilbert.cpp:72-86, chained
 * through unix pipes): stage k filters with taps[k][0..ntaps[k]) and then drops its
 * first skips[k] outputs (hilbert's group-delay compensation, blocks/hilbert.cpp:46,
 * 75-86; 0 for a plain FIR).  The cascade is evaluated in ONE pass with the
 * convolved taps (spectra multiply), so no intermediate stream touches HBM; the
 * first sum_k>=2 (ntaps[k]-1-skips[k]) outputs, where the stage-by-stage result
 * differs from the single filter (a dropped warm-up sample is not fed to the next
 * stage), are computed stage by stage.  Results equal the staged pipeline within
 * fp32 rounding. */
int sdr_chain_create(sdr_chain_t** out, size_t nstages, const float* const* taps, const size_t* ntaps,
                     const size_t* skips, int sample_kind, size_t channels, int device);
/* n input samples per channel in DEVICE memory -> *n_out output samples per channel
 * (n minus what the skips still swallow at the start of the stream), written from
 * d_out[c*out_stride].  Stream position, histories and pending skips carry from
 * call to call.  Asynchronous on hip_stream.  Past the start of the stream (group-delay
 * skips swallowed, staged warm-up done) a call is one call of the fused filter and may be
 * recorded into a HIP graph like sdr_fir_process_device; inside that region a capture is
 * refused (SDR_ERR_INVALID), because the work depends on the host-side position. */
int sdr_chain_process_device(sdr_chain_t* c, const void* d_in, size_t in_stride, size_t n, void* d_out,
                             size_t out_stride, size_t* n_out, void* hip_stream);
int sdr_chain_reset(sdr_chain_t* c);
/* Multi-GPU sharding of one long stream (no counterpart in the single-process
 * reference): declare that `position` input samples of the stream were consumed
 * elsewhere, so this handle is not at the start of the stream -- no group-delay skip
 * and no warm-up are applied from here on.  Histories are zeroed: feed at least
 * sdr_chain_ntaps()-1 samples preceding the first output you keep (their outputs are
 * to be discarded).  position must be 0 (the same as sdr_chain_reset) or at least
 * sdr_chain_min_seek(c): beyond the start-of-stream region -- max(staged warm-up prefix,
 * total group-delay skip) -- plus sdr_chain_ntaps(c). */
int sdr_chain_seek(sdr_chain_t* c, size_t position);
size_t sdr_chain_min_seek(const sdr_chain_t* c);
void sdr_chain_destroy(sdr_chain_t* c);
/* Taps of the single equivalent filter (for tests). */
size_t sdr_chain_ntaps(const sdr_chain_t* c);

/* Waveform generator ("gen" block producer, fused chains) ------------------ */
enum { SDR_WAVE_COSINE = 0, SDR_WAVE_SINE = 1, SDR_WAVE_SQUARE = 2, SDR_WAVE_TRIANGLE = 3, SDR_WAVE_SAWTOOTH = 4 };
typedef struct sdr_gen sdr_gen_t;

/* Replaces sdr::optimal_block_size (reference include/sdr/signal.hpp:73-89):
 * samples per packet for an element size and sample rate (4 KiB pages). */
size_t sdr_optimal_block_size(uint64_t element_size, uint64_t sample_rate);

/* Replaces the set-up of the reference generator loop (blocks/gen.cpp:117-132):
 * amplitude A, frequency in cycles per sample (convert_freq,
 * include/sdr/options.hpp:120-141), initial phase in degrees, packets of
 * `block_size` samples.  SDR_REAL: all five waveforms (blocks/gen.cpp:178-216);
 * SDR_COMPLEX: cosine and sine as A*cexp(j2pi(..)) (:219-230), square / triangle /
 * sawtooth as waveform + j*hilbert(waveform) (:231-260, see sdr_gen_set_hilbert). */
int sdr_gen_create(sdr_gen_t** out, int waveform, int sample_kind, float amplitude, float cycles_per_sample,
                   float phase_degrees, size_t block_size, int device);
/* Replaces the quadrature filter of the reference generator (blocks/gen.cpp:138
 * `kfr::fir_state<RealSample> hilb(hilbert<RealSample>(hilbert_taps))` and the pre-roll
 * at :147-164): complex square / triangle / sawtooth are the waveform plus +-j times its
 * Hilbert transform from a real FIR of `hilbert_taps` taps (default 127); with
 * preroll != 0 (default; the reference does it when there is no frequency-control input)
 * (hilbert_taps-1)/2 samples run through the filter before the first block.  Call before
 * the first generate. */
int sdr_gen_set_hilbert(sdr_gen_t* g, size_t hilbert_taps, int preroll);
/* New frequency from the next block on (the unit=stream control input,
 * blocks/gen.cpp:167-176). */
int sdr_gen_set_frequency(sdr_gen_t* g, float cycles_per_sample);
/* The phase recurrence of the generator loop (blocks/gen.cpp:185 `phi = fract(phi +
 * phi_incr)`) run `blocks` times without producing samples: positions the generator
 * at packet `blocks` further on, bit-for-bit where the reference would be (a rank that
 * owns a later chunk of the stream starts here).  Not for complex square / triangle /
 * sawtooth, whose quadrature filter history cannot be skipped. */
int sdr_gen_advance(sdr_gen_t* g, size_t blocks);
/* Replaces the body of the generator loop (blocks/gen.cpp:183-195, 224-229):
 * the next `blocks` packets' worth of samples, bit-for-bit the reference's fp32
 * recipe (polynomial sine, per-block phase recurrence), written to DEVICE
 * memory d_out, asynchronously on hip_stream.  When the phase recurrence has a closed
 * form -- phase and per-block increment whole multiples of 2^-23, e.g. 0.1 cycles per
 * sample in 2044-sample packets from phase 0 or 45 degrees: no step of it ever rounds --
 * the generator's position lives in device memory and the kernels move it on: no
 * per-call phase table, and a call recorded into a HIP graph replays as the next
 * packets.  Otherwise each call uploads the host's phase table and refuses capture
 * (SDR_ERR_INVALID), as do the quadrature waveforms.  sdr_gen_set_frequency /
 * sdr_gen_advance wait for the generating launches the library itself queued; finish
 * graph replays before calling them. */
int sdr_gen_generate_device(sdr_gen_t* g, size_t blocks, void* d_out, void* hip_stream);
/* Same into HOST memory; returns when `out` is complete. */
int sdr_gen_generate(sdr_gen_t* g, size_t blocks, void* out);
void sdr_gen_destroy(sdr_gen_t* g);

/* Replaces the whole pipeline `gen ... mode=real | hilbert | stream-filter fir=...`
 * (reference: blocks/gen.cpp:166-195 feeding blocks/hilbert.cpp:53-96 through a unix
 * pipe): the chain's outputs for the generator's next `blocks` packets, with the
 * samples synthesised inside the FIR kernel -- no generated or intermediate stream is
 * written to memory (the few packets that overlap the start-of-stream warm-up are
 * materialised once, in a buffer the first call allocates).  Same results as
 * sdr_gen_generate_device followed by sdr_chain_process_device within fp32 rounding (the
 * fused kernels evaluate the sine behind the bit-exact phase argument with fused
 * multiply-adds: samples an ulp or two from the packets' bits).  Real generator,
 * real one-channel chain, packets of 256..2^23 samples; *n_out outputs are written
 * from d_out (DEVICE memory), asynchronously on hip_stream.  A chain short enough
 * for the direct-form kernel is moved to the overlap-save kernels (where the
 * generator lives) by its first call, which must then come before any
 * sdr_chain_process_device on that chain (or after sdr_chain_reset).  Past the start of
 * the stream, with a closed-form generator (see sdr_gen_generate_device), one call
 * recorded into a HIP graph replays as the stream's next call. */
int sdr_chain_generate_device(sdr_chain_t* c, sdr_gen_t* g, size_t blocks, void* d_out, size_t* n_out,
                              void* hip_stream);

/* Pinned host memory (replaces the reference's SampleAllocator,
 * include/sdr/signal.hpp:41-69, for buffers that feed the GPU). */
void* sdr_host_alloc(size_t bytes);
void sdr_host_free(void* p);


#ifdef __cplusplus
}
#endif
#endif /* SDR_AMD_H */
