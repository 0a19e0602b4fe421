// inspect -- drop-in for the reference block (blocks/inspect.cpp:26-68): prints the header of
// every packet (of one stream, if a stream id is given) on stderr.
//
//   inspect [help] [[stream=]ID] [pass=(true|1|false|0)] [pass_all=(true|1|false|0)]
//
// pass: packets of other streams are forwarded to stdout; pass_all: the inspected packets too.
#include <iostream>
#include <vector>

#include "host/cli.hpp"
#include "host/wire.hpp"

// operator<<(std::ostream&, Packet::Content) of the reference (include/sdr/packet.hpp:58-85):
// the enumerator's name; nothing for a value outside the enum
static const char* content_name(uint16_t c) {
    static const char* const names[] = {"Binary", "String", "Time", "Frequency", "Wavelength", "SampleCount",
                                        "Signal", "ComplexSignal", "Spectrum", "ComplexSpectrum"};
    return c < host::kContentCount ? names[c] : "";
}

int main(int argc, char* argv[]) {
    cli::Option id = cli::Option::uint("stream", "ID", 0);
    cli::Option pass = cli::Option::boolean("pass", false);
    cli::Option pass_all = cli::Option::boolean("pass_all", false);
    if (!cli::parse({&id}, {&pass, &pass_all}, argc, argv, std::cerr)) return -1;
    if (!cli::valid_stream_id(id.as_uint())) {
        std::cerr << "error: gen: " << id.as_uint() << " is not a valid stream id" << std::endl; // sic (inspect.cpp:35)
        return -1;
    }
    host::PacketReader source(0);
    host::PacketWriter sink(1);
    std::vector<uint8_t> buf;
    while (source.next()) {
        const host::PacketHeader h = source.header();
        const bool foreign = id.is_set() && h.id != id.as_uint();
        if (foreign && !(pass.as_bool() || pass_all.as_bool())) continue;
        buf.assign(h.size, 0);
        const size_t got = source.read_payload(buf.data(), buf.size());
        if (foreign) {
            // forwarded untouched; a payload cut short by the end of the input goes out as it is
            (void)host::write_full(1, &h, sizeof h);
            (void)host::write_full(1, buf.data(), got);
            continue;
        }
        std::cerr << "Packet{ "
                  << "id: " << h.id << ", "
                  << "content: " << content_name(h.content) << ", "
                  << "size: " << h.size << ", "
                  << "duration: " << h.duration << " }"
                  << " " << got << " bytes received" << std::endl;
        if (pass_all.as_bool()) sink.send(h, buf.data()); // the reference sends the whole buffer (inspect.cpp:63-64)
    }
    return 0;
}
