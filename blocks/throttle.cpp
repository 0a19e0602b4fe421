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

namespace {

// Wall-clock schedule of the paced stream: every paced packet pushes the deadline on by its
// duration, and the block sleeps until the deadline after sending it.
class Pacer {
public:
    Pacer() : deadline_(std::chrono::steady_clock::now()) {}
    void after_packet(uint64_t duration_ns) {
        if (duration_ns == 0) return; // unknown duration: not paced
        deadline_ += std::chrono::nanoseconds(duration_ns);
        std::this_thread::sleep_until(deadline_);
    }

private:
    std::chrono::steady_clock::time_point deadline_;
};

} // namespace

int main(int argc, char* argv[]) {
    cli::Option id = cli::Option::uint("stream", "ID", 0);
    if (!cli::parse({&id}, {}, argc, argv, std::cerr)) return -1;
    if (!cli::valid_stream_id(id.as_uint())) {
        // (the reference names `gen` in this message: blocks/throttle.cpp:34)
        std::cerr << "error: gen: " << id.as_uint() << " is not a valid stream id" << std::endl;
        return -1;
    }
    const bool one_stream = id.is_set();
    const uint64_t paced_id = id.as_uint();
    host::PacketReader source(0);
    std::vector<uint8_t> payload;
    Pacer pacer;
    while (source.next()) {
        const host::PacketHeader h = source.header();
        payload.resize(h.size);
        const size_t got = source.read_payload(payload.data(), payload.size());
        host::GatherWriter out(1); // header and payload in one write
        out.add(&h, sizeof h);
        out.add(payload.data(), got);
        (void)out.flush();
        if (!one_stream || h.id == paced_id) pacer.after_packet(h.duration);
    }
    return 0;
}
