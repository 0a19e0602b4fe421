/* sdr_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's sample-stream FIR hot path
 * (SURVEY.md section 8a rows A1..A7).  It is the checker the HIP path is
 * compared against; nothing in the product (sdr_amd/, include/, blocks) may
 * include, link or call it.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it.
 *
 * Parity pin: every function here is validated by tests/test_oracle_golden.py
 * against golden vectors produced by the reference itself compiled in the
 * build container (oracle/Makefile `ref`, fixtures under tests/golden/,
 * generator tests/golden/make_golden.py).  The reference has no tests or
 * fixtures of its own.
 */
#ifndef SDR_ORACLE_H
#define SDR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- tap design ------------------------------------------------------- */

/* kfr::window_blackman<float>(n), symmetric, alpha = 0.16
 * (libs/kfr/dsp/window.hpp:290-311; linspace basic_expressions.hpp:231-255). */
void sdro_window_blackman(float* w, size_t n);

/* sdr::hilbert<float>(n)  (include/sdr/hilbert.hpp:33-63). n >= 2. */
void sdro_hilbert_taps(float* taps, size_t n);

/* kfr::fir_lowpass/highpass/bandpass/bandstop with a Blackman window and
 * normalize=true (libs/kfr/dsp/fir_design.hpp:35-116).
 * kind: 0 lowpass(f1) 1 highpass(f1) 2 bandpass(f1,f2) 3 bandstop(f1,f2). */
void sdro_fir_design(int kind, float* taps, size_t n, float f1, float f2);
/* The same designs under another window (libs/kfr/dsp/window.hpp): 0 blackman, 1 rectangular,
 * 2 hann, 3 hamming (alpha 0.54), 4 blackman_harris.  -1 for an unknown window. */
int sdro_window(int window, float* w, size_t n);
int sdro_fir_design_window(int kind, int window, float* taps, size_t n, float f1, float f2);
/* ... and 5 = kaiser (window.hpp:336-358, modzerobessel.hpp:42-94), param = beta (kfr's default 0.5). */
void sdro_window_kaiser(float* w, size_t n, float beta);
int sdro_window_param(int window, float param, float* w, size_t n);
int sdro_fir_design_window_param(int kind, int window, float param, float* taps, size_t n, float f1, float f2);

/* ---- streaming FIR (kfr::fir_state + kfr::fir, libs/kfr/dsp/fir.hpp:57-68,
 *      119-148): y[n] = sum_j h[j] x[n-j], zero initial history, history
 *      carried from call to call ------------------------------------------ */

typedef struct sdro_fir sdro_fir;

/* complex != 0: samples are interleaved {re,im} float pairs, taps real.
 * acc64 != 0: accumulate in double and round once (tight checker);
 * acc64 == 0: float accumulation in tap order (scalar port of the reference
 * arithmetic; the one timed as cpu_baseline "port"). */
sdro_fir* sdro_fir_create(const float* taps, size_t ntaps, int complex_samples, int acc64);
void sdro_fir_reset(sdro_fir* f);
void sdro_fir_destroy(sdro_fir* f);
/* n samples in -> n samples out (a complex sample is two floats). */
void sdro_fir_process(sdro_fir* f, const float* in, size_t n, float* out);

/* ---- gen (blocks/gen.cpp:117-132, 178-195, 219-230) --------------------- */

/* include/sdr/signal.hpp:73-89 with a 4 KiB page. */
size_t sdro_optimal_block_size(uint64_t element_size, uint64_t sample_rate);

/* kfr fastsin (libs/kfr/base/sin_cos.hpp:183-206), x in [0, 2pi). */
float sdro_fastsin(float x);

/* Real generator, waveform 0 cosine, 1 sine, 2 square, 3 triangle, 4 sawtooth
 * (libs/kfr/dsp/oscillators.hpp:48-116): writes `blocks`
 * blocks of `block_size` samples; *phi is the running block phase (in/out).
 * cps = cycles per sample. */
void sdro_gen_real(float* out, size_t blocks, size_t block_size, float amp,
                   float cps, float* phi, int waveform);

/* Complex cosine (0) / sine (1): A*cexp(j2pi(phi [-0.25] + cps*i)). */
void sdro_gen_complex(float* out, size_t blocks, size_t block_size, float amp,
                      float cps, float* phi, int waveform);

/* Complex square (2) / triangle (3) / sawtooth (4): the waveform plus +-j times its
 * Hilbert transform from a real FIR (blocks/gen.cpp:138, 147-164, 231-260).  `hilb` is
 * a real sdro_fir made from sdro_hilbert_taps(hilbert_taps), hilb_delay =
 * (hilbert_taps - 1) / 2; the pre-roll runs once before the first block. */
void sdro_gen_quad_preroll(sdro_fir* hilb, size_t hilb_delay, float cps, float phi, int waveform);
void sdro_gen_complex_quad(float* out, size_t blocks, size_t block_size, float amp, float cps,
                           float* phi, int waveform, sdro_fir* hilb, size_t hilb_delay);

/* ---- hilbert block (blocks/hilbert.cpp:29-99) over an in-memory packet
 *      stream.  Returns bytes written to out (capacity out_cap), or
 *      (size_t)-1 when out_cap is too small.  Foreign packets are passed
 *      through once (the reference's fall-through twin, SURVEY.md 8b Q1, is
 *      emitted as header + zero payload only when emulate_q1 != 0). -------- */
size_t sdro_hilbert_block(const uint8_t* in, size_t in_bytes, uint8_t* out, size_t out_cap,
                          uint16_t stream_id, size_t ntaps, int delay, int emulate_q1);

/* ---- stream-filter decision (blocks/stream_filter.cpp:59-86).
 * mode 0 pass / 1 drop; ids/contents may be NULL with count 0 = "unset".
 * Returns 1 when the packet is forwarded. */
int sdro_stream_filter_forward(int mode, const uint64_t* ids, size_t nids, int ids_set,
                               const uint16_t* contents, size_t ncontents, int contents_set,
                               uint16_t pkt_id, uint16_t pkt_content);

#ifdef __cplusplus
}
#endif
#endif
