// hilbert -- drop-in for the reference block of the same name (blocks/hilbert.cpp:29-99):
// Hilbert-transformer FIR over one stream id, real or complex samples, optional
// group-delay compensation.  Same argv grammar, same stdin/stdout packet framing,
// same tap design; the kfr::fir inner loop runs on an MI355X through libsdr_amd.
//
//   hilbert [help] [[stream=]ID] [[taps=]TAPS] [delay=(true|1|false|0)]
//
// Deliberate differences from the reference:
//   * a packet that is not filtered is forwarded once (the reference falls through
//     after pass() and emits a second, garbage copy: blocks/hilbert.cpp:56-58);
//   * taps < 5 is rejected (the reference designs all-NaN taps for 2..4 and
//     underflows for 0);
//   * there is no CPU compute path: without a GPU the block exits with an error.
#include <cstdlib>
#include <iostream>
#include <vector>

#include "host/cli.hpp"
#include "host/fir_pipeline.hpp"

static int block_main(int argc, char* argv[]) {
    cli::Option id = cli::Option::uint("stream", "ID", 0);
    cli::Option taps = cli::Option::uint("taps", "TAPS", 127);
    cli::Option delay = cli::Option::boolean("delay", true);
    if (!cli::parse({&id, &taps}, {&delay}, argc, argv, std::cerr)) return -1;

    if (!cli::valid_stream_id(id.as_uint())) {
        // (sic) the reference's message names `gen`
        std::cerr << "error: gen: " << id.as_uint() << " is not a valid stream id" << std::endl;
        return -1;
    }
    const uint64_t ntaps = taps.as_uint();
    if (ntaps < 5 || ntaps > SDR_MAX_TAPS) {
        std::cerr << "error: hilbert: taps must be between 5 and 131072" << std::endl;
        return -1;
    }

    host::PipelineConfig cfg;
    cfg.taps.resize(ntaps);
    if (sdr_hilbert_taps(cfg.taps.data(), ntaps) != SDR_OK) {
        std::cerr << "error: hilbert: " << sdr_last_error() << std::endl;
        return -1;
    }
    cfg.stream_id = static_cast<uint16_t>(id.as_uint());
    cfg.skip = delay.as_bool() ? (ntaps - 1) / 2 : 0; // blocks/hilbert.cpp:46
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
