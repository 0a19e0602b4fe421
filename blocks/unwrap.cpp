// unwrap -- drop-in for the reference block (blocks/unwrap.cpp:24-44): writes the payloads of
// the packets on stdin to stdout, headers removed; with a stream id, only that stream's.
//
//   unwrap [help] [[stream=]ID]
#include <iostream>
#include <vector>

#include "host/cli.hpp"
#include "host/wire.hpp"

int main(int argc, char* argv[]) {
    cli::Option id = cli::Option::uint("stream", "ID", 0);
    if (!cli::parse({&id}, {}, argc, argv, std::cerr)) return -1;
    if (!cli::valid_stream_id(id.as_uint())) {
        std::cerr << "error: gen: " << id.as_uint() << " is not a valid stream id" << std::endl; // sic (unwrap.cpp:31)
        return -1;
    }
    host::PacketReader source(0);
    std::vector<uint8_t> buf(1 << 16);
    while (source.next()) {
        if (id.is_set() && source.header().id != id.as_uint()) continue;
        while (source.remaining() && !source.eof()) {
            const size_t got = source.read_payload(buf.data(), buf.size());
            if (got == 0 || !host::write_full(1, buf.data(), got)) break;
        }
    }
    return 0;
}
