// stream-filter -- drop-in for the reference block (blocks/stream_filter.cpp:35-88):
// forwards or discards whole packets by stream id and content type.
//
//   stream-filter [help] [mode=](pass|drop) [[stream=]ID,...] [[content=](binary|...),...]
//
// With only these options the block never touches a payload byte and needs no GPU.
//
// EXTENSION (not in the reference; BASELINE.json's north_star names this block for
// "the real/complex FIR", which the reference implements nowhere but in its vendored
// kfr::fir + kfr::fir_lowpass/... , libs/kfr/dsp/fir.hpp, fir_design.hpp): the keyword
// options
//   fir=(lowpass|highpass|bandpass|bandstop) [taps=COUNT] cutoff=F [cutoff2=F]
//   [window=(blackman|rectangular|hann|hamming|blackman_harris|kaiser)] [beta=B]
// run every forwarded Signal / ComplexSignal packet through that FIR on the GPU
// (taps designed like kfr::fir_* under that kfr window, Blackman by default, F = Hz / sample_rate; one
// filter state shared by the forwarded sample packets, no group-delay skip).
#include <cstdlib>
#include <iostream>
#include <vector>

#include "host/cli.hpp"
#include "host/fir_pipeline.hpp"
#include "host/wire.hpp"

namespace {

const cli::Choices kModes = {{"pass", 0}, {"drop", 1}};
const cli::Choices kContents = {
    {"binary", host::kBinary}, {"string", host::kString}, {"time", host::kTime},
    {"frequency", host::kFrequency}, {"wavelength", host::kWavelength},
    {"sample_count", host::kSampleCount}, {"signal", host::kSignal},
    {"complex_signal", host::kComplexSignal}, {"spectrum", host::kSpectrum},
    {"complex_spectrum", host::kComplexSpectrum}};
const cli::Choices kFirKinds = {{"none", -1}, {"lowpass", SDR_LOWPASS}, {"highpass", SDR_HIGHPASS},
                                {"bandpass", SDR_BANDPASS}, {"bandstop", SDR_BANDSTOP}};
const cli::Choices kWindows = {{"blackman", SDR_WINDOW_BLACKMAN}, {"rectangular", SDR_WINDOW_RECTANGULAR},
                               {"hann", SDR_WINDOW_HANN}, {"hamming", SDR_WINDOW_HAMMING},
                               {"blackman_harris", SDR_WINDOW_BLACKMAN_HARRIS}, {"kaiser", SDR_WINDOW_KAISER}};

struct Rule {
    bool drop_mode;
    const cli::Option* ids;
    const cli::Option* contents;
    // The reference's two loops (blocks/stream_filter.cpp:59-86) as one predicate.
    bool forwards(const host::PacketHeader& h) const {
        const bool id_hit = ids->is_set() && ids->as_uint_set().count(h.id);
        const bool content_hit = contents->is_set() && contents->as_choice_set().count(h.content);
        if (!drop_mode) {
            if (ids->is_set() && !id_hit) return false;
            if (contents->is_set() && !content_hit) return false;
            return true;
        }
        if (ids->is_set()) return !(id_hit && (!contents->is_set() || content_hit));
        if (contents->is_set()) return !content_hit;
        return false; // drop with no criteria: drop everything
    }
};

bool forward_cb(const host::PacketHeader& h, void* ctx) { return static_cast<Rule*>(ctx)->forwards(h); }

} // namespace

static int block_main(int argc, char* argv[]) {
    cli::Option mode = cli::Option::choice("mode", kModes, 0, /*required=*/true);
    cli::Option ids = cli::Option::uint_set("stream", "ID,...");
    cli::Option content = cli::Option::choice_set("content", kContents);
    // extension options: looked up by key only, never shown in the reference usage line
    cli::Option fir = cli::Option::choice("fir", kFirKinds, -1);
    cli::Option taps = cli::Option::uint("taps", "COUNT", 127);
    cli::Option cutoff = cli::Option::real("cutoff", "F", 0.0f);
    cli::Option cutoff2 = cli::Option::real("cutoff2", "F", 0.0f);
    cli::Option window = cli::Option::choice("window", kWindows, SDR_WINDOW_BLACKMAN);
    cli::Option beta = cli::Option::real("beta", "B", 0.5f); // the Kaiser window's parameter (kfr's default)

    const std::vector<cli::Option*> positional = {&mode, &ids, &content};
    for (int i = 1; i < argc; ++i)
        if (std::string(argv[i]) == "help") {
            cli::usage(argv[0], positional, {}, std::cerr);
            std::cerr << "Extension (GPU FIR on forwarded sample packets): [fir=(lowpass|highpass|bandpass|bandstop)]"
                         " [taps=COUNT] [cutoff=F] [cutoff2=F] [window=(blackman|rectangular|hann|hamming|blackman_harris|kaiser)] [beta=B]" << std::endl;
            return -1;
        }
    if (!cli::parse(positional, {&fir, &taps, &cutoff, &cutoff2, &window, &beta}, argc, argv, std::cerr)) return -1;

    if (!mode.is_set()) {
        std::cerr << "error: stream_filter: option 'mode' is required" << std::endl;
        cli::usage(argv[0], positional, {}, std::cerr);
        return -1;
    }
    for (uint64_t id : ids.as_uint_set())
        if (!cli::valid_stream_id(id)) {
            std::cerr << "error: stream_filter: " << id << " is not a valid stream id" << std::endl;
            return -1;
        }

    Rule rule{mode.as_choice() == 1, &ids, &content};

    if (!fir.is_set() || fir.as_choice() < 0) {
        // the reference block: pure packet routing on the host, payloads never looked at
        (void)host::route_packets(0, 1, forward_cb, &rule);
        host::PacketWriter(1).flush_to_disk();
        if (std::getenv("SDR_AMD_ROUTE_STATS")) {
            const host::RouteStats& st = host::route_stats();
            std::cerr << "route: spliced=" << st.spliced << " sendfile=" << st.sent << " copied=" << st.copied
                      << " skipped=" << st.skipped << std::endl;
        }
        return 0;
    }

    const uint64_t ntaps = taps.as_uint();
    if (ntaps < 2 || ntaps > SDR_MAX_TAPS) {
        std::cerr << "error: stream_filter: taps must be between 2 and 131072" << std::endl;
        return -1;
    }
    if (!cutoff.is_set()) {
        std::cerr << "error: stream_filter: fir= needs cutoff=F (normalised frequency)" << std::endl;
        return -1;
    }
    host::PipelineConfig cfg;
    cfg.taps.resize(ntaps);
    if (sdr_fir_design_window(fir.as_choice(), window.as_choice(), beta.as_float(), cfg.taps.data(), ntaps,
                              cutoff.as_float(), cutoff2.as_float()) != SDR_OK) {
        std::cerr << "error: stream_filter: " << sdr_last_error() << std::endl;
        return -1;
    }
    cfg.select = forward_cb;
    cfg.select_ctx = &rule;
    cfg.forward_other = forward_cb;
    cfg.forward_ctx = &rule;
    if (const char* e = std::getenv("SDR_AMD_BATCH_BYTES")) cfg.batch_bytes = std::strtoull(e, nullptr, 0);
    if (const char* e = std::getenv("SDR_AMD_DEVICE")) cfg.device = std::atoi(e);
    int rc;
    {
        host::FirPipeline pipe(cfg);
        rc = pipe.run(0, 1);
    } // (the GPU state is released here)
    if (rc == host::FirPipeline::kOutputClosed) host::end_output_closed(); // end like the reference: by SIGPIPE
    return rc;
}

// the body runs in a child process, so that a reader that goes away ends the block at once (host/wire.hpp)
int main(int argc, char* argv[]) { return host::supervised_main(argc, argv, block_main); }
