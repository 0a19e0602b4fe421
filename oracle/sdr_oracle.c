/* sdr_oracle.c -- TEST INFRASTRUCTURE ONLY (see sdr_oracle.h).
 *
 * CPU restatement, in plain C, of the arithmetic of the reference's FIR hot
 * path.  Each function cites the reference file:line it follows.  Compiled with
 * -ffp-contract=off so the fp32 operation order below is what executes.
 *
 * Parity pin: tests/test_oracle_golden.py checks every function against golden
 * vectors made by the reference itself (tests/golden/make_golden.py).
 */
#include "sdr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static const float PI_F = 3.14159265358979323846f;
static const float TWO_PI_F = 6.28318530717958647692f;
static const float FOUR_PI_F = 12.56637061435917295385f;

/* ------------------------------------------------------------------ A2 ---
 * libs/kfr/dsp/window.hpp:290-311: a0 - a1*cos(2pi*t) + a2*cos(4pi*t),
 * t_i = 0 + float(i) * ((1-0)/float(n-1))  (basic_expressions.hpp:236-251,
 * endpoint=true for a symmetric window, window.hpp:71-78).
 * KFR evaluates cos with its own fp32 polynomial (sin_cos.hpp:167-181); libm
 * cosf of the same fp32 argument differs by <= 1 ulp of the result, far below
 * the tolerance. */
void sdro_window_blackman(float* w, size_t n) {
    const float alpha = 0.16f;
    const float a0 = (float)((1 - alpha) * 0.5);
    const float a1 = 0.5f;
    const float a2 = (float)(alpha * 0.5);
    const float offset = (1.0f - 0.0f) / (float)(n - 1);
    for (size_t i = 0; i < n; ++i) {
        float t = 0.0f + (float)(int64_t)i * offset;
        float c1 = cosf(TWO_PI_F * t);
        float c2 = cosf(FOUR_PI_F * t);
        /* the -O3 -ffast-math reference build evaluates (a0 + a2*c2) - a1*c1
         * (end points come out exactly 0); same value to 1 ulp otherwise */
        w[i] = (a0 + a2 * c2) - a1 * c1;
    }
}

/* The other windows the FIR designs can take (libs/kfr/dsp/window.hpp:120-139 rectangular,
 * :204-223 hann, :247-267 hamming alpha 0.54, :314-334 blackman_harris), all over the same
 * symmetric linspace 0..1 as the Blackman window above.  window: 0 blackman, 1 rectangular,
 * 2 hann, 3 hamming, 4 blackman_harris, 5 kaiser (param = beta). */
/* kfr::modzerobessel in fp32 (libs/kfr/base/modzerobessel.hpp:42-94): the power series of
 * I0, sum_k ((x/2)^2)^k / (k!)^2, accumulated term by term -- 1 + q, then 20 more terms with
 * coefficients 1/((k+2)!)^2 (rounded to fp32 from their exact values). */
static float modzerobessel_f(float x) {
    const float h = x * 0.5f;
    const float q = h * h;
    float num = q;
    float result = 1.0f + q;
    double fact = 1.0; /* (k+1)! */
    for (int k = 0; k < 20; ++k) {
        fact *= (double)(k + 2);
        const float coef = (float)(1.0 / (fact * fact));
        num *= q;
        result = num * coef + result;
    }
    return result;
}

/* kfr::window_kaiser (libs/kfr/dsp/window.hpp:336-358): I0(beta sqrt(1 - t^2)) / I0(beta) on
 * the plain linspace t_i = -1 + i * (2 / (n - 1)). */
void sdro_window_kaiser(float* w, size_t n, float beta) {
    const float m = 1.0f / modzerobessel_f(beta);
    const float offset = (1.0f - -1.0f) / (float)(n - 1);
    for (size_t i = 0; i < n; ++i) {
        const float t = -1.0f + (float)(int64_t)i * offset;
        w[i] = modzerobessel_f(beta * sqrtf(1.0f - t * t)) * m;
    }
}

int sdro_window(int window, float* w, size_t n) { return sdro_window_param(window, 0.5f, w, n); }

int sdro_window_param(int window, float param, float* w, size_t n) {
    if (window == 0) { sdro_window_blackman(w, n); return 0; }
    if (window == 5) { sdro_window_kaiser(w, n, param); return 0; }
    if (window < 0 || window > 5) return -1;
    const float offset = (1.0f - 0.0f) / (float)(n - 1);
    for (size_t i = 0; i < n; ++i) {
        const float t = 0.0f + (float)(int64_t)i * offset;
        const float x = TWO_PI_F * t;
        switch (window) {
            case 1: w[i] = 1.0f; break;
            case 2: w[i] = 0.5f * (1.0f - cosf(x)); break;
            case 3: w[i] = 0.54f - (1.0f - 0.54f) * cosf(x); break;
            default: w[i] = 0.35875f - 0.48829f * cosf(x) + 0.14128f * cosf(2.0f * x) - 0.01168f * cosf(3.0f * x); break;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ A1 ---
 * include/sdr/hilbert.hpp:33-63. */
void sdro_hilbert_taps(float* taps, size_t n) {
    size_t size = n - !(n & 1); /* :37 largest odd <= n */
    float* wnd = (float*)malloc(sizeof(float) * (size ? size : 1));
    sdro_window_blackman(wnd, size); /* :39-40 */
    memset(taps, 0, n * sizeof(float)); /* univector<T>(size) value-initialised, :78 */

    float gain = 0.0f;
    int64_t half = (int64_t)(size / 2);
    for (int64_t i = 1; i <= half; ++i) { /* :46-55 */
        if (i & 1) {
            float x = 1.0f / (float)i;
            taps[half + i] = x * wnd[half + i];
            taps[half - i] = -x * wnd[half - i];
            gain = taps[half + i] - gain;
        } else {
            taps[half + i] = taps[half - i] = 0.0f;
        }
    }
    if (size < n) taps[n - 1] = 0.0f; /* :57-58 */
    gain = 2 * fabsf(gain); /* :60 */
    for (size_t i = 0; i < n; ++i) taps[i] = taps[i] / gain; /* :61-62 */
    free(wnd);
}

/* kfr::sinc (libs/kfr/base/sin_cos.hpp:228-232). */
static float sinc_f(float x) {
    return fabsf(x) <= 1.1920928955078125e-07f ? 1.0f : sinf(x) / x;
}

/* symmlinspace<T, precise=true>(symsize, n, endpoint=true)
 * (basic_expressions.hpp:257-283): mix(float(i)*invsize, -symsize, +symsize),
 * mix(t,x,y) = (1-t)*x + t*y, invsize = 1.0/float(n-1) stored as float.
 * With x = -y the reference's release build (-O3 -ffast-math, meson.build:36-40)
 * folds this to (t + t - 1)*symsize; that form reproduces the golden taps to
 * 1 ulp, whereas the literal form suffers cancellation near the centre
 * (1e-4 relative at 4095 taps) and does not. */
static float symmlin(float symsize, size_t n, size_t i) {
    float invsize = (float)(1.0 / (double)(float)(n - 1));
    float t = (float)(int64_t)i * invsize;
    return ((t + t) - 1.0f) * symsize;
}

/* ------------------------------------------------------------------ A7 ---
 * libs/kfr/dsp/fir_design.hpp:35-116, Blackman window, normalize=true. */
void sdro_fir_design(int kind, float* taps, size_t n, float f1, float f2) {
    (void)sdro_fir_design_window(kind, 0, taps, n, f1, f2);
}

int sdro_fir_design_window(int kind, int window, float* taps, size_t n, float f1, float f2) {
    return sdro_fir_design_window_param(kind, window, 0.5f, taps, n, f1, f2);
}

int sdro_fir_design_window_param(int kind, int window, float param, float* taps, size_t n, float f1, float f2) {
    float* wnd = (float*)malloc(sizeof(float) * (n ? n : 1));
    if (sdro_window_param(window, param, wnd, n) != 0) { free(wnd); return -1; }
    float sum = 0.0f;
    if (kind == 0 || kind == 1) {
        /* :38-44 / :55-61 */
        const float scale = (float)(2.0 * (kind == 0 ? f1 : -f1));
        const float sym = (float)(n - 1) * f1 * PI_F;
        for (size_t i = 0; i < n; ++i) taps[i] = sinc_f(symmlin(sym, n, i)) * scale * wnd[i];
        if (n & 1) taps[n / 2] = kind == 0 ? scale : (float)(1 - 2.0 * f1);
    } else {
        /* :74-85 / :98-109 */
        const float scale1 = (float)(2.0 * f1);
        const float scale2 = (float)(2.0 * f2);
        const float sc = PI_F * (float)(n - 1);
        const float start1 = sc * f1;
        const float start2 = sc * f2;
        for (size_t i = 0; i < n; ++i) {
            float s1 = sinc_f(symmlin(start1, n, i)) * scale1;
            float s2 = sinc_f(symmlin(start2, n, i)) * scale2;
            taps[i] = (kind == 2 ? (s2 - s1) : (s1 - s2)) * wnd[i];
        }
        if (n & 1) taps[n / 2] = kind == 2 ? 2 * (f2 - f1) : 1 - 2 * (f2 - f1);
    }
    for (size_t i = 0; i < n; ++i) sum += taps[i];
    /* :46-50 (lowpass), :111-115 (bandstop): 1/sum ; :63-67 (highpass),
     * :87-91 (bandpass): 1/(sum+1) -- the reference's own quirk, kept. */
    const float invsum = (kind == 0 || kind == 3) ? 1.0f / sum : 1.0f / (sum + 1);
    for (size_t i = 0; i < n; ++i) taps[i] = taps[i] * invsum;
    free(wnd);
    return 0;
}

/* --------------------------------------------------------------- A3/A4 ---
 * kfr::fir_state (fir.hpp:57-68): zero delay line, persists across calls.
 * expression_fir::operator() (fir.hpp:129-146) evaluates, per input sample,
 * the dot product of the reversed taps with the ring-buffered delay line,
 * i.e. y[n] = sum_{j=0}^{T-1} h[j] x[n-j] with x[<0] = 0.  The ring buffer is
 * restated as a linear history of the last T-1 inputs. */
struct sdro_fir {
    size_t ntaps;
    int cplx;
    int acc64;
    float* taps;
    float* hist; /* (T-1) samples, oldest first */
};

sdro_fir* sdro_fir_create(const float* taps, size_t ntaps, int complex_samples, int acc64) {
    if (!taps || ntaps == 0) return NULL;
    sdro_fir* f = (sdro_fir*)calloc(1, sizeof(*f));
    f->ntaps = ntaps;
    f->cplx = complex_samples != 0;
    f->acc64 = acc64 != 0;
    f->taps = (float*)malloc(ntaps * sizeof(float));
    memcpy(f->taps, taps, ntaps * sizeof(float));
    f->hist = (float*)calloc((ntaps - 1 + 1) * (f->cplx ? 2 : 1), sizeof(float));
    return f;
}

void sdro_fir_reset(sdro_fir* f) { /* filter_fir::reset, fir.hpp:195-199 */
    memset(f->hist, 0, (f->ntaps - 1) * (f->cplx ? 2 : 1) * sizeof(float));
}

void sdro_fir_destroy(sdro_fir* f) {
    if (!f) return;
    free(f->taps);
    free(f->hist);
    free(f);
}

void sdro_fir_process(sdro_fir* f, const float* in, size_t n, float* out) {
    const size_t T = f->ntaps, H = T - 1, w = f->cplx ? 2 : 1;
    if (n == 0) return;
    float* x = (float*)malloc((H + n) * w * sizeof(float));
    memcpy(x, f->hist, H * w * sizeof(float));
    memcpy(x + H * w, in, n * w * sizeof(float));
    const float* h = f->taps;
    if (!f->cplx) {
        for (size_t i = 0; i < n; ++i) {
            const float* p = x + H + i; /* x[n] ; x[n-j] = p[-j] */
            if (f->acc64) {
                double a = 0.0;
                for (size_t j = 0; j < T; ++j) a += (double)h[j] * (double)p[-(ptrdiff_t)j];
                out[i] = (float)a;
            } else {
                float a = 0.0f;
                for (size_t j = 0; j < T; ++j) a += h[j] * p[-(ptrdiff_t)j];
                out[i] = a;
            }
        }
    } else {
        for (size_t i = 0; i < n; ++i) {
            const float* p = x + 2 * (H + i);
            if (f->acc64) {
                double ar = 0.0, ai = 0.0;
                for (size_t j = 0; j < T; ++j) {
                    ar += (double)h[j] * (double)p[-2 * (ptrdiff_t)j];
                    ai += (double)h[j] * (double)p[-2 * (ptrdiff_t)j + 1];
                }
                out[2 * i] = (float)ar;
                out[2 * i + 1] = (float)ai;
            } else {
                float ar = 0.0f, ai = 0.0f;
                for (size_t j = 0; j < T; ++j) {
                    ar += h[j] * p[-2 * (ptrdiff_t)j];
                    ai += h[j] * p[-2 * (ptrdiff_t)j + 1];
                }
                out[2 * i] = ar;
                out[2 * i + 1] = ai;
            }
        }
    }
    /* carry the last T-1 inputs (fir_state::delayline persists, fir.hpp:65-67) */
    memcpy(f->hist, x + n * w, H * w * sizeof(float));
    free(x);
}

/* ------------------------------------------------------------------ A6 ---
 * include/sdr/signal.hpp:73-89, page_size = 4096, sizeof(Packet) = 16. */
static uint64_t gcd_u64(uint64_t a, uint64_t b) {
    while (b) { uint64_t t = a % b; a = b; b = t; }
    return a;
}

size_t sdro_optimal_block_size(uint64_t element_size, uint64_t sample_rate) {
    const uint64_t page = 4096, hdr = 16;
    uint64_t size = (2 * page - hdr) / element_size;
    if (size < 1) size = 1;
    if (sample_rate == 0) return (size_t)size;
    const uint64_t step = (hdr / gcd_u64(hdr, element_size) * element_size) / element_size;
    uint64_t limit = (page - hdr) / element_size;
    if (limit < step) limit = step;
    while (size > limit && (size * 1000000000ull) % sample_rate > 0) size -= step;
    return (size_t)size;
}

/* libs/kfr/base/sin_cos.hpp:183-206 */
float sdro_fastsin(float x) {
    const float c2 = -0.16665853559970855712890625f;
    const float c4 = +8.31427983939647674560546875e-3f;
    const float c6 = -1.85423981747590005397796630859375e-4f;
    const float pi = PI_F;
    float xx = x - pi;
    float y = fabsf(xx);
    y = y > (PI_F * 0.5f) ? pi - y : y;
    y = signbit(xx) ? y : -y; /* y ^ (highbit & ~xx) */
    float y2 = y * y;
    float formula = c6;
    float y3 = y2 * y;
    formula = formula * y2 + c4;
    formula = formula * y2 + c2;
    formula = formula * y3 + y;
    return formula;
}

static float fract_f(float x) { return x - floorf(x); } /* round.hpp:67 */

/* blocks/gen.cpp:178-195: block = A*sinenorm(phi [+0.25f] + cps*counter());
 * phi = fract(phi + phi_incr), phi_incr = fract(cps*block_size) (:124);
 * sinenorm(x) = fastsin(fract(x) * 2pi) (oscillators.hpp:48-57);
 * counter() is the in-block int index converted to float
 * (basic_expressions.hpp:112-116). */
/* the five normalised oscillators of libs/kfr/dsp/oscillators.hpp at argument `arg`
 * (cosine = sine: the caller adds the quarter cycle) */
static float wave_value(int waveform, float arg) {
    switch (waveform) {
        case 2: /* squarenorm, oscillators.hpp:64-73 */
            return fract_f(arg) < 0.5f ? 1.0f : -1.0f;
        case 3: { /* trianglenorm, oscillators.hpp:107-116: 1 - |4 fract(x + 0.25) - 2|.
                   * The reference's release build (-ffast-math) folds the +0.25 out of
                   * the fraction: 4 (x - floor(x + 0.25)) - 1, which keeps the bits the
                   * literal form loses when x + 0.25 is rounded (up to 1.3e-5 of RMS at
                   * x ~ 40); the golden packets pin this form. */
            float fl = floorf(arg + 0.25f);
            return 1.0f - fabsf(4.0f * (arg - fl) - 1.0f);
        }
        case 4: /* sawtoothnorm, oscillators.hpp:80-89: 1 - 2 fract(x) */
            return 1.0f - 2.0f * fract_f(arg);
        default: /* sinenorm, oscillators.hpp:48-57 */
            return sdro_fastsin(fract_f(arg) * TWO_PI_F);
    }
}

void sdro_gen_real(float* out, size_t blocks, size_t block_size, float amp,
                   float cps, float* phi_io, int waveform) {
    float phi = *phi_io;
    const float phi_incr = fract_f(cps * (float)block_size);
    for (size_t b = 0; b < blocks; ++b) {
        float base = waveform == 0 ? phi + 0.25f : phi;
        for (size_t i = 0; i < block_size; ++i) {
            float arg = base + cps * (float)(int)i;
            out[b * block_size + i] = amp * wave_value(waveform, arg);
        }
        phi = fract_f(phi + phi_incr);
    }
    *phi_io = phi;
}

/* blocks/gen.cpp:147-164: before the first block of a complex square / triangle /
 * sawtooth stream without a frequency-control input, hilb_delay samples of the
 * waveform run through the quadrature filter and are discarded. */
void sdro_gen_quad_preroll(sdro_fir* hilb, size_t hilb_delay, float cps, float phi, int waveform) {
    for (size_t i = 0; i < hilb_delay; ++i) {
        float x = wave_value(waveform, phi + cps * (float)(int)i), y;
        sdro_fir_process(hilb, &x, 1, &y);
    }
}

/* blocks/gen.cpp:231-260: block = A*(wave(phi + cps*i) + (+-j)*fir(hilb, wave(phi +
 * (cps*hilb_delay) + cps*i))), -j for a negative frequency; hilb = real FIR state with
 * sdr::hilbert taps (:138), carried from block to block. */
void sdro_gen_complex_quad(float* out, size_t blocks, size_t block_size, float amp, float cps,
                           float* phi_io, int waveform, sdro_fir* hilb, size_t hilb_delay) {
    float phi = *phi_io;
    const float phi_incr = fract_f(cps * (float)block_size);
    const float sign = cps < 0 ? -1.0f : 1.0f;
    for (size_t b = 0; b < blocks; ++b) {
        const float adv = phi + (cps * (float)hilb_delay);
        for (size_t i = 0; i < block_size; ++i) {
            float re = wave_value(waveform, phi + cps * (float)(int)i);
            float x = wave_value(waveform, adv + cps * (float)(int)i), q;
            sdro_fir_process(hilb, &x, 1, &q);
            out[2 * (b * block_size + i)] = amp * re;
            out[2 * (b * block_size + i) + 1] = amp * (sign * q);
        }
        phi = fract_f(phi + phi_incr);
    }
    *phi_io = phi;
}

/* blocks/gen.cpp:219-230: A*cexp(j2pi*(phi + (sine ? -0.25f : 0) + cps*i));
 * cexp (complex.hpp:570-573) = exp(re)*cossin(im), re = 0.  The fp32 angle
 * 2pi_f*((phi+off)+cps*float(i)) is part of the result (no fract here);
 * sin/cos of that fp32 angle are evaluated accurately (double libm), which
 * reproduces KFR's own trig_sincos to ~1 ulp. */
void sdro_gen_complex(float* out, size_t blocks, size_t block_size, float amp,
                      float cps, float* phi_io, int waveform) {
    float phi = *phi_io;
    const float phi_incr = fract_f(cps * (float)block_size);
    for (size_t b = 0; b < blocks; ++b) {
        float base = phi + (waveform == 1 ? -0.25f : 0.0f);
        for (size_t i = 0; i < block_size; ++i) {
            float ang = TWO_PI_F * (base + cps * (float)(int)i);
            out[2 * (b * block_size + i)] = amp * (float)cos((double)ang);
            out[2 * (b * block_size + i) + 1] = amp * (float)sin((double)ang);
        }
        phi = fract_f(phi + phi_incr);
    }
    *phi_io = phi;
}

/* ------------------------------------------------------------------ A5 ---
 * blocks/hilbert.cpp:53-96 over a byte buffer.  Packet header
 * (include/sdr/packet.hpp:28-45): u16 id, u16 content, u32 size, u64 duration,
 * little endian, 16 bytes. */
typedef struct {
    uint16_t id, content;
    uint32_t size;
    uint64_t duration;
} pkt_hdr;

enum { PKT_SIGNAL = 6, PKT_COMPLEX_SIGNAL = 7 };

size_t sdro_hilbert_block(const uint8_t* in, size_t in_bytes, uint8_t* out, size_t out_cap,
                          uint16_t stream_id, size_t ntaps, int delay, int emulate_q1) {
    float* taps = (float*)malloc(ntaps * sizeof(float));
    sdro_hilbert_taps(taps, ntaps);
    /* hilbert.cpp:45: fir_state<float, complex<float>> -- real packets are
     * widened to complex (imag 0) and only .real() is kept (:63-64, :89-92);
     * filtering the real part alone is arithmetically identical. */
    sdro_fir* fr = sdro_fir_create(taps, ntaps, 1, 0);
    size_t hilb_delay = delay ? (ntaps - 1) / 2 : 0; /* :46 */
    size_t ip = 0, op = 0;
    int fail = 0;

    while (!fail && in_bytes - ip >= sizeof(pkt_hdr)) { /* Source::next, stream.cpp:131-136 */
        pkt_hdr pkt;
        memcpy(&pkt, in + ip, sizeof pkt);
        ip += sizeof pkt;
        size_t avail = in_bytes - ip;
        size_t have = pkt.size < avail ? pkt.size : avail; /* short payload => eof after this one */

        if (pkt.id != stream_id || (pkt.content != PKT_SIGNAL && pkt.content != PKT_COMPLEX_SIGNAL)) {
            /* :56-58 source.pass(sink) */
            size_t need = sizeof pkt + have + (emulate_q1 ? sizeof pkt + pkt.size : 0);
            if (op + need > out_cap) { fail = 1; break; }
            memcpy(out + op, &pkt, sizeof pkt);
            op += sizeof pkt;
            memcpy(out + op, in + ip, have);
            op += have;
            if (emulate_q1) { /* fall-through twin: same header, unspecified payload */
                memcpy(out + op, &pkt, sizeof pkt);
                op += sizeof pkt;
                memset(out + op, 0, pkt.size);
                op += pkt.size;
            }
            ip += have;
            continue;
        }

        const int real = pkt.content == PKT_SIGNAL;
        const size_t elt = real ? 4 : 8;
        /* Packet::count<T>() is 0 when size % sizeof(T) != 0 (packet.hpp:47-55);
         * Source::recv returns whole elements actually read (stream.hpp:77-83) */
        size_t count = (pkt.size % elt) ? 0 : pkt.size / elt;
        size_t got = have / elt < count ? have / elt : count;

        float* cin = (float*)calloc(2 * (got ? got : 1), sizeof(float));
        float* cout = (float*)calloc(2 * (got ? got : 1), sizeof(float));
        if (real) {
            for (size_t i = 0; i < got; ++i) memcpy(&cin[2 * i], in + ip + 4 * i, 4);
        } else {
            memcpy(cin, in + ip, got * 8);
        }
        ip += have;
        sdro_fir_process(fr, cin, got, cout); /* :72-73 */

        size_t skip = got < hilb_delay ? got : hilb_delay; /* :75 */
        int swallowed = 0;
        if (skip > 0) { /* :77-86 */
            hilb_delay -= skip;
            pkt.size -= (uint32_t)(skip * elt);
            pkt.duration -= (skip * pkt.duration) / got;
            if (pkt.size == 0) swallowed = 1;
        }
        if (!swallowed) {
            /* Sink::send(pkt, data, count): size rewritten only when count != 0
             * (stream.hpp:133-139) */
            size_t cnt = got - skip;
            if (cnt) pkt.size = (uint32_t)(cnt * elt);
            if (op + sizeof pkt + pkt.size > out_cap) fail = 1;
            else {
                memcpy(out + op, &pkt, sizeof pkt);
                op += sizeof pkt;
                if (cnt) {
                    if (real)
                        for (size_t i = 0; i < cnt; ++i) memcpy(out + op + 4 * i, &cout[2 * (skip + i)], 4);
                    else
                        memcpy(out + op, cout + 2 * skip, cnt * 8);
                    op += cnt * elt;
                } else {
                    memset(out + op, 0, pkt.size); /* unspecified bytes in the reference */
                    op += pkt.size;
                }
            }
        }
        free(cin);
        free(cout);
    }
    sdro_fir_destroy(fr);
    free(taps);
    return fail ? (size_t)-1 : op;
}

/* ------------------------------------------------------------------ A7 ---
 * blocks/stream_filter.cpp:59-86. */
static int in_u64(const uint64_t* v, size_t n, uint64_t x) {
    for (size_t i = 0; i < n; ++i) if (v[i] == x) return 1;
    return 0;
}
static int in_u16(const uint16_t* v, size_t n, uint16_t x) {
    for (size_t i = 0; i < n; ++i) if (v[i] == x) return 1;
    return 0;
}

int sdro_stream_filter_forward(int mode, const uint64_t* ids, size_t nids, int ids_set,
                               const uint16_t* contents, size_t ncontents, int contents_set,
                               uint16_t pkt_id, uint16_t pkt_content) {
    if (mode == 0) { /* :59-68 */
        if (ids_set && !in_u64(ids, nids, pkt_id)) return 0;
        if (contents_set && !in_u16(contents, ncontents, pkt_content)) return 0;
        return 1;
    }
    /* :69-86 */
    if (ids_set) {
        if (in_u64(ids, nids, pkt_id)) {
            if (!contents_set) return 0;
            else if (in_u16(contents, ncontents, pkt_content)) return 0;
        }
    } else if (contents_set) {
        if (in_u16(contents, ncontents, pkt_content)) return 0;
    } else {
        return 0;
    }
    return 1;
}
