// throttle -- drop-in for the reference block (blocks/throttle.cpp:27-52): forwards every
// packet and paces the stream to wall-clock time by the packets' `duration` (of one stream,
// if a stream id is given; packets without a duration are not paced).
//
//   throttle [help] [[stream=]ID]
#include <chrono>
#include <iostream>
#include <thread>
#include <vector>

#include "host/cli.hpp"
#include "host/wire.hpp"

int main(int argc, char* argv[]) {
    cli::Option id = cli::Option::uint("stream", "ID", 0);
    if (!cli::parse({&id}, {}, argc, argv, std::cerr)) return -1;
    if (!cli::valid_stream_id(id.as_uint())) {
        std::cerr << "error: gen: " << id.as_uint() << " is not a valid stream id" << std::endl; // sic (throttle.cpp:34)
        return -1;
    }
    host::PacketReader source(0);
    std::vector<uint8_t> buf;
    auto next_packet = std::chrono::steady_clock::now();
    while (source.next()) {
        const host::PacketHeader h = source.header();
        buf.resize(h.size);
        const size_t got = source.read_payload(buf.data(), buf.size());
        (void)host::write_full(1, &h, sizeof h);
        (void)host::write_full(1, buf.data(), got);
        if (!(id.is_set() && h.id != id.as_uint()) && h.duration) {
            next_packet += std::chrono::nanoseconds(h.duration);
            std::this_thread::sleep_until(next_packet);
        }
    }
    return 0;
}
