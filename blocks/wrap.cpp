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
        std::cerr << "error: wrap: option 'element_size' is required" << std::endl;
        cli::usage(argv[0], pos, kw, std::cerr);
        return -1;
    }
    if (element_size.as_uint() < 1) {
        std::cerr << "error: wrap: element_size must be greater than zero" << std::endl;
        return -1;
    }
    if (!cli::valid_stream_id(id.as_uint())) {
        std::cerr << "error: gen: " << id.as_uint() << " is not a valid stream id" << std::endl; // sic (wrap.cpp:51)
        return -1;
    }
    const uint64_t es = element_size.as_uint(), rate = sample_rate.as_uint(), dur = duration.as_uint();
    host::PacketHeader hdr;
    hdr.id = static_cast<uint16_t>(id.as_uint());
    hdr.content = static_cast<uint16_t>(content.as_choice());
    hdr.size = 0;
    hdr.duration = dur;
    if (element_count.is_set()) {                                               // wrap.cpp:60-79
        if (element_count.as_uint() == 0) {
            std::cerr << "error: wrap: element_count must be greater than zero" << std::endl;
            return -1;
        }
        hdr.size = static_cast<uint32_t>(es * element_count.as_uint());
    } else if (sample_rate.is_set() && duration.is_set()) {
        const uint64_t count = dur * rate / 1000000000ull;
        if (count < 1) {
            std::cerr << "error: wrap: packet duration is too small" << std::endl;
            return -1;
        }
        hdr.size = static_cast<uint32_t>(es * count);
    } else {
        hdr.size = static_cast<uint32_t>(es * sdr_optimal_block_size(es, rate));
    }
    if (sample_rate.is_set()) {                                                 // :81-88
        hdr.duration = rate ? (hdr.size / es) * 1000000000ull / rate : 0;
        if (duration.is_set() && element_count.is_set() && dur != hdr.duration) {
            std::cerr << "error: wrap: duration and element_count do not match" << std::endl;
            return -1;
        }
    }
    if (!content.is_set())
        std::cerr << "warning: wrap: content type not set, input will be treated as binary data" << std::endl;

    std::vector<uint8_t> buf(hdr.size);
    host::PacketWriter sink(1);
    for (;;) {
        const size_t got = host::read_full(0, buf.data(), buf.size());
        if (got == 0) break;
        host::PacketHeader h = hdr;
        h.size = static_cast<uint32_t>(got);
        sink.send(h, buf.data());
        if (got < buf.size()) break;
    }
    sink.flush_to_disk();
    return 0;
}
