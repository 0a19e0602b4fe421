// wrap -- drop-in for the reference block (blocks/wrap.cpp:25-102): cuts a raw byte stream
// on stdin into packets on stdout (header synthesis; no GPU work).
//
//   wrap [help] [[content_type=](binary|...|complex_spectrum)] [[stream=]ID] element_size=BYTES
//        [element_count=COUNT] [duration=NANOSECONDS] [sample_rate=HERTZ]
//
// Packet payload size: element_size*element_count, else element_size*(duration*sample_rate/1e9),
// else element_size*optimal_block_size(element_size, sample_rate) (include/sdr/signal.hpp:73-89).
// Difference from the reference on a PIPE input (lib/sdr/stream.cpp:121-158, 251-296: the raw
// source cannot see the end of a pipe coming): the reference writes the last header with the
// full packet size in front of whatever bytes were left -- a bare header when the input ended
// on a packet boundary; here the last header carries the number of bytes that follow it, and
// nothing is written after the input ends (what the reference itself does on a seekable input).
#include <cstdlib>
#include <iostream>
#include <vector>

#include "../include/sdr_amd.h"
#include "host/cli.hpp"
#include "host/wire.hpp"

namespace {

// What one packet of the wrapped stream looks like, or why it cannot be built.
struct Shape {
    uint32_t bytes = 0;       // payload size
    uint64_t nanoseconds = 0; // header duration
    const char* problem = nullptr;
};

constexpr uint64_t kNsPerSecond = 1000000000ull;

// The reference's sizing rules (blocks/wrap.cpp:60-88), as one function of the options:
// an explicit element count wins; else a duration at a sample rate; else the block size the
// generator would use.  A sample rate also fixes the duration written into every header.
Shape packet_shape(uint64_t elt, bool has_count, uint64_t count, bool has_duration, uint64_t duration,
                   bool has_rate, uint64_t rate) {
    Shape sh;
    sh.nanoseconds = duration;
    uint64_t elements;
    if (has_count) {
        if (count == 0) { sh.problem = "element_count must be greater than zero"; return sh; }
        elements = count;
    } else if (has_rate && has_duration) {
        elements = duration * rate / kNsPerSecond;
        if (elements == 0) { sh.problem = "packet duration is too small"; return sh; }
    } else {
        elements = sdr_optimal_block_size(elt, rate);
    }
    sh.bytes = static_cast<uint32_t>(elt * elements);
    if (has_rate) {
        sh.nanoseconds = rate ? (sh.bytes / elt) * kNsPerSecond / rate : 0;
        if (has_duration && has_count && duration != sh.nanoseconds) sh.problem = "duration and element_count do not match";
    }
    return sh;
}

int complain(const char* what) {
    std::cerr << "error: wrap: " << what << std::endl;
    return -1;
}

} // namespace

int main(int argc, char* argv[]) {
    cli::Option content = cli::Option::choice("content_type", host::content_choices(), host::kBinary);
    cli::Option id = cli::Option::uint("stream", "ID", 0);
    cli::Option element_size = cli::Option::uint("element_size", "BYTES", 0, /*required=*/true);
    cli::Option element_count = cli::Option::uint("element_count", "COUNT", 0);
    cli::Option duration = cli::Option::uint("duration", "NANOSECONDS", 0);
    cli::Option sample_rate = cli::Option::uint("sample_rate", "HERTZ", 0);
    const std::vector<cli::Option*> pos = {&content, &id};
    const std::vector<cli::Option*> kw = {&element_size, &element_count, &duration, &sample_rate};
    if (!cli::parse(pos, kw, argc, argv, std::cerr)) return -1;

    if (!element_size.is_set()) {
        complain("option 'element_size' is required");
        cli::usage(argv[0], pos, kw, std::cerr);
        return -1;
    }
    if (element_size.as_uint() == 0) return complain("element_size must be greater than zero");
    if (!cli::valid_stream_id(id.as_uint())) {
        // (the reference names `gen` in this message: blocks/wrap.cpp:51)
        std::cerr << "error: gen: " << id.as_uint() << " is not a valid stream id" << std::endl;
        return -1;
    }
    const Shape shape = packet_shape(element_size.as_uint(), element_count.is_set(), element_count.as_uint(),
                                     duration.is_set(), duration.as_uint(), sample_rate.is_set(), sample_rate.as_uint());
    if (shape.problem) return complain(shape.problem);
    if (!content.is_set())
        std::cerr << "warning: wrap: content type not set, input will be treated as binary data" << std::endl;

    host::PacketHeader every;
    every.id = static_cast<uint16_t>(id.as_uint());
    every.content = static_cast<uint16_t>(content.as_choice());
    every.size = shape.bytes;
    every.duration = shape.nanoseconds;

    // one read per packet; the last one may come up short and says so in its header
    std::vector<uint8_t> payload(shape.bytes);
    host::PacketWriter sink(1);
    size_t got;
    do {
        got = host::read_full(0, payload.data(), payload.size());
        if (got == 0) break;
        host::PacketHeader h = every;
        h.size = static_cast<uint32_t>(got);
        sink.send(h, payload.data());
    } while (got == payload.size());
    sink.flush_to_disk();
    return 0;
}
