// ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// A thin driver (our own code) around the *reference's* header-only KFR and
// sdr::hilbert templates, compiled by oracle/Makefile against the headers where
// they lie under /root/reference, output into oracle/_ref/.  It exposes the
// reference arithmetic that has no executable of its own:
//
//   ref_harness taps hilbert N                       -> N float32 on stdout
//        (include/sdr/hilbert.hpp:33-81)
//   ref_harness taps lowpass|highpass N FC           -> N float32
//   ref_harness taps bandpass|bandstop N F1 F2       -> N float32
//        (libs/kfr/dsp/fir_design.hpp:35-116 with window_blackman, alpha 0.16)
//   ref_harness fir real|complex TAPS.bin BLOCK      -> stdin raw f32 / c64
//        samples filtered by kfr::fir(fir_state&, ...) (libs/kfr/dsp/fir.hpp:167-171)
//        in calls of BLOCK samples (state carried), raw samples on stdout.
//
// Used to generate tests/golden/* (tests/golden/make_golden.py) and as the
// "reference" CPU baseline of bench.py.
#include "hilbert.hpp"
#include "signal.hpp"

#include "kfr/dsp/fir.hpp"
#include "kfr/dsp/fir_design.hpp"
#include "kfr/dsp/window.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using cf = kfr::complex<float>;

static void write_all(const void* p, size_t bytes) {
    if (bytes && fwrite(p, 1, bytes, stdout) != bytes) {
        perror("write");
        exit(2);
    }
}

static std::vector<float> read_taps(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    std::vector<float> t;
    float v;
    while (fread(&v, 4, 1, f) == 1) t.push_back(v);
    fclose(f);
    return t;
}

template <typename U>
static int run_fir(const std::vector<float>& taps, size_t block) {
    kfr::univector<float> t(taps.size());
    for (size_t i = 0; i < taps.size(); ++i) t[i] = taps[i];
    kfr::fir_state<float, U> st(t);
    std::vector<U> in(block), out(block);
    for (;;) {
        size_t n = fread(in.data(), sizeof(U), block, stdin);
        if (n == 0) break;
        kfr::make_univector(out.data(), n) = kfr::fir(st, kfr::make_univector(in.data(), n));
        write_all(out.data(), n * sizeof(U));
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) return 1;
    std::string cmd = argv[1];
    if (cmd == "taps" && argc >= 4) {
        std::string kind = argv[2];
        size_t n = strtoull(argv[3], nullptr, 10);
        kfr::univector<float> taps(n);
        if (kind == "hilbert") {
            taps = sdr::hilbert<float>(n);
        } else {
            float f1 = argc > 4 ? strtof(argv[4], nullptr) : 0.f;
            float f2 = argc > 5 ? strtof(argv[5], nullptr) : 0.f;
            std::string wname = argc > 6 ? argv[6] : "blackman";
            float wparam = argc > 7 ? strtof(argv[7], nullptr) : 0.5f;
            // (each window expression is bound to a named variable before to_pointer: see the
            // note on g++ at the top of this file)
            auto design = [&](auto wexpr) {
                kfr::expression_pointer<float> w = kfr::to_pointer(wexpr);
                if (kind == "lowpass") kfr::fir_lowpass(taps, f1, w, true);
                else if (kind == "highpass") kfr::fir_highpass(taps, f1, w, true);
                else if (kind == "bandpass") kfr::fir_bandpass(taps, f1, f2, w, true);
                else if (kind == "bandstop") kfr::fir_bandstop(taps, f1, f2, w, true);
                else return 1;
                return 0;
            };
            int rc = 1;
            if (wname == "blackman") rc = design(kfr::window_blackman<float>(n));
            else if (wname == "rectangular") rc = design(kfr::internal::expression_rectangular<float>(n));
            else if (wname == "hann") rc = design(kfr::window_hann<float>(n));
            else if (wname == "hamming") rc = design(kfr::window_hamming<float>(n));
            else if (wname == "blackman_harris") rc = design(kfr::window_blackman_harris<float>(n));
            else if (wname == "kaiser") rc = design(kfr::window_kaiser<float>(n, wparam));
            if (rc) return rc;
        }
        write_all(taps.data(), n * sizeof(float));
        return 0;
    }
    if (cmd == "window" && argc >= 4) { // window NAME N [PARAM]: the window itself
        std::string wname = argv[2];
        size_t n = strtoull(argv[3], nullptr, 10);
        float wparam = argc > 4 ? strtof(argv[4], nullptr) : 0.5f;
        kfr::univector<float> w(n);
        if (wname == "blackman") w = kfr::window_blackman<float>(n);
        else if (wname == "rectangular") w = kfr::internal::expression_rectangular<float>(n);
        else if (wname == "hann") w = kfr::window_hann<float>(n);
        else if (wname == "hamming") w = kfr::window_hamming<float>(n);
        else if (wname == "blackman_harris") w = kfr::window_blackman_harris<float>(n);
        else if (wname == "kaiser") w = kfr::window_kaiser<float>(n, wparam);
        else return 1;
        write_all(w.data(), n * sizeof(float));
        return 0;
    }
    if (cmd == "fir" && argc >= 5) {
        auto taps = read_taps(argv[3]);
        size_t block = strtoull(argv[4], nullptr, 10);
        if (!strcmp(argv[2], "real")) return run_fir<float>(taps, block);
        return run_fir<cf>(taps, block);
    }
    return 1;
}
