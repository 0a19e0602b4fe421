// gen -- drop-in for the reference waveform source (blocks/gen.cpp:83-268), sample
// synthesis on the GPU (libsdr_amd's generator restates the reference's fp32 recipe:
// polynomial sine, per-block phase recurrence).
//
//   gen [help] [freq=]FREQ [[unit=](hertz|hz|meters|meter|m|samples|stream)]
//       [[waveform=](cosine|sine|square|triangle|sawtooth)] sample_rate=HERTZ [amp=AMPLITUDE]
//       [phi=PHASE] [mode=(real|complex)] [hilbert_taps=COUNT] [stream=ID]
//
// Runs until its output closes (SIGPIPE), like the reference; with unit=stream (FREQ =
// the stream id of a frequency-control input on stdin, blocks/gen.cpp:61-81, 143-146,
// 167-176) it runs until that input ends, looking at the control input before every packet
// it produces, like the reference (one packet per GPU round trip in this mode).
#include <algorithm>
#include <condition_variable>
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../include/sdr_amd.h"
#include "host/cli.hpp"
#include "host/wire.hpp"

namespace {
const cli::Choices kUnits = {{"hertz", 0}, {"hz", 0}, {"meters", 1}, {"meter", 1}, {"m", 1}, {"samples", 2}, {"stream", 3}};
const cli::Choices kWaveforms = {{"cosine", SDR_WAVE_COSINE}, {"sine", SDR_WAVE_SINE}, {"square", SDR_WAVE_SQUARE},
                                 {"triangle", SDR_WAVE_TRIANGLE}, {"sawtooth", SDR_WAVE_SAWTOOTH}};
const cli::Choices kModes = {{"real", SDR_REAL}, {"complex", SDR_COMPLEX}};

// include/sdr/options.hpp:120-141 of the reference, fp32
float cycles_per_sample(int unit, float f, uint64_t sample_rate) {
    switch (unit) {
        case 1: f = 299792458.0f / f; /* fallthrough */
        case 0: f /= static_cast<float>(sample_rate); break;
        case 2: f = 1.0f / f; break;
        default: break;
    }
    if (unit == 3 || std::isnan(f) || std::isinf(f)) return 0.0f;
    return f;
}

// The frequency-control input of unit=stream (reference blocks/gen.cpp:61-81, 143-146): a reader
// thread takes Frequency / Wavelength / SampleCount packets of stream `id` from stdin (first
// float of the payload) and leaves the latest value, or the end of the input, for the generator
// loop, which looks before every packet it produces (:167-176, :187).
class FrequencyControl {
public:
    // Reader side: runs until stdin ends.
    void read_stdin(uint16_t id, uint64_t sample_rate) {
        host::PacketReader source(0);
        while (source.next()) {
            const host::PacketHeader& h = source.header();
            const int unit = h.content == host::kFrequency ? 0 : h.content == host::kWavelength ? 1
                             : h.content == host::kSampleCount ? 2 : 3;
            if (h.id != id || unit == 3 || h.size < sizeof(float) || h.size % sizeof(float) != 0) continue;
            float f = 0.0f;
            if (source.read_payload(&f, sizeof f) != sizeof f) continue;
            std::lock_guard<std::mutex> lk(mu_);
            value_ = cycles_per_sample(unit, f, sample_rate);
            news_ = true;
        }
        std::lock_guard<std::mutex> lk(mu_);
        ended_ = news_ = true;
    }
    // Generator side: anything new since the last look?  *ended: the input is over (the reference
    // stops at the first look after the end, whatever value arrived just before it).
    bool poll(float* cps, bool* ended) {
        std::lock_guard<std::mutex> lk(mu_);
        if (!news_) return false;
        news_ = false;
        *cps = value_;
        *ended = ended_;
        return true;
    }

private:
    std::mutex mu_;
    float value_ = 0.0f;
    bool news_ = false, ended_ = false;
};
FrequencyControl control;
} // namespace

static int block_main(int argc, char* argv[]) {
    cli::Option freq = cli::Option::real("freq", "FREQ", 0.0f, true);
    cli::Option unit = cli::Option::choice("unit", kUnits, 0);
    cli::Option waveform = cli::Option::choice("waveform", kWaveforms, SDR_WAVE_COSINE);
    cli::Option sample_rate = cli::Option::uint("sample_rate", "HERTZ", 0, true);
    cli::Option amp = cli::Option::real("amp", "AMPLITUDE", 1.0f);
    cli::Option phase = cli::Option::real("phi", "PHASE", 0.0f);
    cli::Option mode = cli::Option::choice("mode", kModes, SDR_COMPLEX);
    cli::Option hilbert_taps = cli::Option::uint("hilbert_taps", "COUNT", 127);
    cli::Option id = cli::Option::uint("stream", "ID", 0);
    const std::vector<cli::Option*> pos = {&freq, &unit, &waveform};
    const std::vector<cli::Option*> kw = {&sample_rate, &amp, &phase, &mode, &hilbert_taps, &id};
    if (!cli::parse(pos, kw, argc, argv, std::cerr)) return -1;

    if (!freq.is_set() || !sample_rate.is_set()) {
        std::cerr << "error: gen: options 'freq' and 'sample_rate' are required" << std::endl;
        cli::usage(argv[0], pos, kw, std::cerr);
        return -1;
    }
    if (unit.as_choice() == 3) {
        const float f = freq.as_float();
        if (!(f >= 0 && f <= 65535 && f == std::floor(f))) {
            std::cerr << "error: gen: " << f << " is not a valid stream id" << std::endl;
            return -1;
        }
    }
    if (!cli::valid_stream_id(id.as_uint())) {
        std::cerr << "error: gen: " << id.as_uint() << " is not a valid stream id" << std::endl;
        return -1;
    }
    const int kind = mode.as_choice();
    const bool no_source = unit.as_choice() != 3;
    const bool quadrature = kind == SDR_COMPLEX && waveform.as_choice() > SDR_WAVE_SINE;
    if (quadrature && (hilbert_taps.as_uint() < 5 || hilbert_taps.as_uint() > SDR_MAX_TAPS)) {
        // the reference's tap design is 0/0 below 5 taps (all-NaN quadrature)
        std::cerr << "error: gen: hilbert_taps must be between 5 and 131072" << std::endl;
        return -1;
    }
    if (sdr_device_count() < 1) {
        std::cerr << "error: gen: no HIP device found (this block has no CPU path)" << std::endl;
        return -1;
    }

    const size_t elt = kind == SDR_REAL ? 4 : 8;
    const uint64_t rate = sample_rate.as_uint();
    const size_t block = sdr_optimal_block_size(elt, rate);                   // gen.cpp:117-118
    const float cps = cycles_per_sample(unit.as_choice(), freq.as_float(), rate); // :122

    host::PacketHeader hdr;                                                   // :128-132
    hdr.id = static_cast<uint16_t>(id.as_uint());
    hdr.content = kind == SDR_REAL ? host::kSignal : host::kComplexSignal;
    hdr.size = static_cast<uint32_t>(block * elt);
    hdr.duration = rate ? block * 1000000000ull / rate : 0;

    sdr_gen_t* g = nullptr;
    const int device = std::getenv("SDR_AMD_DEVICE") ? std::atoi(std::getenv("SDR_AMD_DEVICE")) : 0;
    if (sdr_gen_create(&g, waveform.as_choice(), kind, amp.as_float(), cps, phase.as_float(), block, device) != SDR_OK) {
        std::cerr << "error: gen: " << sdr_last_error() << std::endl;
        return -1;
    }
    if (quadrature && sdr_gen_set_hilbert(g, hilbert_taps.as_uint(), no_source) != SDR_OK) {     // :138, 147-164
        std::cerr << "error: gen: " << sdr_last_error() << std::endl;
        return -1;
    }
    if (!no_source) {                                                         // :143-146
        std::thread thr([&, rate] { control.read_stdin(static_cast<uint16_t>(freq.as_float()), rate); });
        thr.detach();
    }
    host::widen_pipe(1);
    float current = cps;
    // ~4 MiB per GPU round trip with a fixed frequency; ONE packet per round trip with a control
    // input, so that a retune takes effect at the next packet produced, as in the reference
    size_t blocks_per_call = no_source ? 512 : 1;
    if (const char* e = std::getenv("SDR_AMD_GEN_BLOCKS")) blocks_per_call = std::max(1, std::atoi(e));
    // two pinned buffers: with a fixed frequency a producer thread fills one on the GPU while
    // the main thread writes the other to the pipe
    uint8_t* bufs[2] = {static_cast<uint8_t*>(sdr_host_alloc(blocks_per_call * block * elt)),
                        no_source ? static_cast<uint8_t*>(sdr_host_alloc(blocks_per_call * block * elt)) : nullptr};
    if (!bufs[0] || (no_source && !bufs[1])) {
        std::cerr << "error: gen: " << sdr_last_error() << std::endl;
        return -1;
    }
    auto write_packets = [&](const uint8_t* buf) {
        host::GatherWriter out(1);
        for (size_t b = 0; b < blocks_per_call; ++b) {
            out.add(&hdr, sizeof hdr);
            out.add(buf + b * block * elt, block * elt);
        }
        return out.flush();
    };
    int rc = 0;
    bool output_closed = false;
    host::ignore_sigpipe();
    if (no_source) {
        std::mutex mu;
        std::condition_variable cv;
        bool full[2] = {false, false};
        bool stop = false, failed = false;
        std::string error;
        std::thread producer([&] {
            for (int i = 0;; i ^= 1) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return !full[i] || stop; });
                    if (stop) return;
                }
                const bool ok = sdr_gen_generate(g, blocks_per_call, bufs[i]) == SDR_OK;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (!ok) { failed = true; error = sdr_last_error(); }
                    full[i] = true;
                }
                cv.notify_all();
                if (!ok) return;
            }
        });
        for (int i = 0;; i ^= 1) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return full[i]; });
                if (failed) {
                    std::cerr << "error: gen: " << error << std::endl;
                    rc = -1;
                    break;
                }
            }
            if (!write_packets(bufs[i])) { output_closed = true; break; }
            {
                std::lock_guard<std::mutex> lk(mu);
                full[i] = false;
            }
            cv.notify_all();
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        producer.join();
    } else {
        for (;;) {
            float msg = current;
            bool ended = false;
            if (control.poll(&msg, &ended)) {                                                      // :167-176
                if (ended) break;
                if (msg != current) {
                    current = msg;
                    sdr_gen_set_frequency(g, current);
                }
            }
            if (sdr_gen_generate(g, blocks_per_call, bufs[0]) != SDR_OK) {
                std::cerr << "error: gen: " << sdr_last_error() << std::endl;
                rc = -1;
                break;
            }
            if (!write_packets(bufs[0])) { output_closed = true; break; }
        }
    }
    sdr_gen_destroy(g);
    sdr_host_free(bufs[0]);
    sdr_host_free(bufs[1]);
    if (output_closed) host::end_output_closed(); // the GPU is released; end like the reference does
    return rc;
}

// the body runs in a child process, so that a reader that goes away ends the block at once (host/wire.hpp)
int main(int argc, char* argv[]) { return host::supervised_main(argc, argv, block_main); }
