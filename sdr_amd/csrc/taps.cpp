// taps.cpp -- host-side tap designers of libsdr_amd (product code, no GPU needed).
//
// The FIR kernels take taps as data; these functions produce the same tap
// vectors the reference blocks design for themselves, following the reference's
// fp32 arithmetic step by step so a drop-in block filters with the same numbers:
//   sdr_hilbert_taps  <- sdr::hilbert<float>(n), include/sdr/hilbert.hpp:33-63
//   sdr_fir_design    <- kfr::fir_lowpass/highpass/bandpass/bandstop,
//                        libs/kfr/dsp/fir_design.hpp:35-116
//   blackman()        <- kfr::window_blackman, libs/kfr/dsp/window.hpp:290-311
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/sdr_amd.h"
#include "errors.h"

namespace {

constexpr float kPi = 3.14159265358979323846f;

// Symmetric Blackman window, alpha = 0.16: a0 - a1 cos(2 pi t) + a2 cos(4 pi t)
// on t_i = i/(n-1) built as float(i)*step like kfr's linspace
// (libs/kfr/base/basic_expressions.hpp:236-251).  Summation order
// (a0 + a2 c2) - a1 c1 is the one the reference's release build uses: it makes
// both end points exactly zero, as they are in the reference.
std::vector<float> blackman(size_t n) {
    std::vector<float> w(n);
    const float alpha = 0.16f;
    const float a0 = static_cast<float>((1 - alpha) * 0.5);
    const float a1 = 0.5f;
    const float a2 = static_cast<float>(alpha * 0.5);
    const float step = 1.0f / static_cast<float>(n - 1);
    for (size_t i = 0; i < n; ++i) {
        const float t = static_cast<float>(static_cast<int64_t>(i)) * step;
        const float c1 = std::cos(2.0f * kPi * t);
        const float c2 = std::cos(4.0f * kPi * t);
        w[i] = (a0 + a2 * c2) - a1 * c1;
    }
    return w;
}

// The other windows a design can take, over the same grid t_i = i/(n-1)
// (libs/kfr/dsp/window.hpp:120-139 rectangular, :204-223 hann, :247-267 hamming with
// alpha 0.54, :314-334 blackman_harris).
// kfr::modzerobessel in fp32 (libs/kfr/base/modzerobessel.hpp:42-94): I0 by its power series
// sum_k ((x/2)^2)^k / (k!)^2, 1 + q and twenty further terms accumulated in fp32.
float bessel_i0(float x) {
    const float half = x * 0.5f, q = half * half;
    float power = q, sum = 1.0f + q;
    double factorial = 1.0;
    for (int k = 2; k < 22; ++k) {
        factorial *= k;
        power *= q;
        sum = power * static_cast<float>(1.0 / (factorial * factorial)) + sum;
    }
    return sum;
}

// kfr::window_kaiser (libs/kfr/dsp/window.hpp:336-358) on t_i = -1 + i * 2/(n-1)
std::vector<float> kaiser(size_t n, float beta) {
    std::vector<float> w(n);
    const float norm = 1.0f / bessel_i0(beta);
    const float step = 2.0f / static_cast<float>(n - 1);
    for (size_t i = 0; i < n; ++i) {
        const float t = -1.0f + static_cast<float>(static_cast<int64_t>(i)) * step;
        w[i] = bessel_i0(beta * std::sqrt(1.0f - t * t)) * norm;
    }
    return w;
}

bool design_window(int window, float param, size_t n, std::vector<float>& w) {
    if (window == SDR_WINDOW_BLACKMAN) {
        w = blackman(n);
        return true;
    }
    if (window == SDR_WINDOW_KAISER) {
        w = kaiser(n, param);
        return true;
    }
    if (window < SDR_WINDOW_BLACKMAN || window > SDR_WINDOW_KAISER) return false;
    w.resize(n);
    const float step = 1.0f / static_cast<float>(n - 1);
    for (size_t i = 0; i < n; ++i) {
        const float x = 2.0f * kPi * (static_cast<float>(static_cast<int64_t>(i)) * step);
        switch (window) {
            case SDR_WINDOW_RECTANGULAR: w[i] = 1.0f; break;
            case SDR_WINDOW_HANN: w[i] = 0.5f * (1.0f - std::cos(x)); break;
            case SDR_WINDOW_HAMMING: w[i] = 0.54f - (1.0f - 0.54f) * std::cos(x); break;
            default:
                w[i] = 0.35875f - 0.48829f * std::cos(x) + 0.14128f * std::cos(2.0f * x) - 0.01168f * std::cos(3.0f * x);
        }
    }
    return true;
}

// kfr::sinc (libs/kfr/base/sin_cos.hpp:228-232)
inline float sinc(float x) {
    return std::fabs(x) <= 1.1920928955078125e-07f ? 1.0f : std::sin(x) / x;
}

// The sinc argument grid: kfr's precise symmetric linspace over [-a, +a]
// (basic_expressions.hpp:257-283), in the algebraic form (2t-1)*a to which the
// reference's -ffast-math release build reduces mix(t, -a, +a).
struct SymGrid {
    float a, inv;
    SymGrid(float a_, size_t n) : a(a_), inv(static_cast<float>(1.0 / static_cast<double>(static_cast<float>(n - 1)))) {}
    float operator()(size_t i) const {
        const float t = static_cast<float>(static_cast<int64_t>(i)) * inv;
        return ((t + t) - 1.0f) * a;
    }
};

} // namespace

extern "C" int sdr_hilbert_taps(float* taps, size_t n) {
    if (!taps) return sdr_amd::fail(SDR_ERR_INVALID, "sdr_hilbert_taps: taps is NULL");
    // n <= 4: window end points are 0, the gain is 0 and the reference yields
    // 0/0 taps; rejected here instead of emitting NaNs.
    if (n < 5) return sdr_amd::fail(SDR_ERR_INVALID, "sdr_hilbert_taps: at least 5 taps are required");
    const size_t odd = (n & 1) ? n : n - 1; // hilbert.hpp:37
    const std::vector<float> w = blackman(odd);
    for (size_t i = 0; i < n; ++i) taps[i] = 0.0f;
    const int64_t mid = static_cast<int64_t>(odd / 2);
    float gain = 0.0f;
    for (int64_t k = 1; k <= mid; k += 2) { // odd offsets only; even ones stay 0 (hilbert.hpp:46-55)
        const float inv = 1.0f / static_cast<float>(k);
        taps[mid + k] = inv * w[mid + k];
        taps[mid - k] = -inv * w[mid - k];
        gain = taps[mid + k] - gain; // alternating running sum
    }
    gain = 2 * std::fabs(gain); // hilbert.hpp:60
    for (size_t i = 0; i < n; ++i) taps[i] = taps[i] / gain;
    return SDR_OK;
}

extern "C" int sdr_fir_design(int kind, float* taps, size_t n, float f1, float f2) {
    return sdr_fir_design_window(kind, SDR_WINDOW_BLACKMAN, 0.0f, taps, n, f1, f2);
}

extern "C" int sdr_fir_design_window(int kind, int window, float window_param, float* taps, size_t n, float f1,
                                     float f2) {
    if (!taps || n < 2) return sdr_amd::fail(SDR_ERR_INVALID, "sdr_fir_design: need taps and n >= 2");
    if (kind < SDR_LOWPASS || kind > SDR_BANDSTOP) return sdr_amd::fail(SDR_ERR_INVALID, "sdr_fir_design: unknown kind");
    std::vector<float> w;
    if (!design_window(window, window_param, n, w)) return sdr_amd::fail(SDR_ERR_INVALID, "sdr_fir_design: unknown window");
    const bool odd = n & 1;
    if (kind == SDR_LOWPASS || kind == SDR_HIGHPASS) {
        // fir_design.hpp:38-44 and :55-61
        const float gain = static_cast<float>(2.0 * (kind == SDR_LOWPASS ? f1 : -f1));
        const SymGrid grid(static_cast<float>(n - 1) * f1 * kPi, n);
        for (size_t i = 0; i < n; ++i) taps[i] = sinc(grid(i)) * gain * w[i];
        if (odd) taps[n / 2] = kind == SDR_LOWPASS ? gain : static_cast<float>(1 - 2.0 * f1);
    } else {
        // fir_design.hpp:74-85 and :98-109
        const float g1 = static_cast<float>(2.0 * f1), g2 = static_cast<float>(2.0 * f2);
        const float span = kPi * static_cast<float>(n - 1);
        const SymGrid grid1(span * f1, n), grid2(span * f2, n);
        for (size_t i = 0; i < n; ++i) {
            const float lo = sinc(grid1(i)) * g1, hi = sinc(grid2(i)) * g2;
            taps[i] = (kind == SDR_BANDPASS ? hi - lo : lo - hi) * w[i];
        }
        if (odd) taps[n / 2] = kind == SDR_BANDPASS ? 2 * (f2 - f1) : 1 - 2 * (f2 - f1);
    }
    float total = 0.0f;
    for (size_t i = 0; i < n; ++i) total += taps[i];
    // normalize=true: 1/sum for lowpass & bandstop, 1/(sum+1) for highpass &
    // bandpass (the reference's own formulae, fir_design.hpp:48,65,89,113).
    const float k = (kind == SDR_LOWPASS || kind == SDR_BANDSTOP) ? 1.0f / total : 1.0f / (total + 1);
    for (size_t i = 0; i < n; ++i) taps[i] *= k;
    return SDR_OK;
}
