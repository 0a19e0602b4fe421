// wire.hpp -- packet framing and blocking packet I/O for the drop-in blocks.
//
// Wire format (must stay byte-compatible with the reference, which writes the
// raw struct of include/sdr/packet.hpp:28-45): a 16-byte little-endian header
//   off 0  u16 id        stream id
//   off 2  u16 content   0 Binary 1 String 2 Time 3 Frequency 4 Wavelength
//                        5 SampleCount 6 Signal 7 ComplexSignal 8 Spectrum 9 ComplexSpectrum
//   off 4  u32 size      payload bytes
//   off 8  u64 duration  nanoseconds, 0 = unknown
// followed by `size` payload bytes; no padding, magic or trailer.
// Signal = float32 samples, ComplexSignal = {float re, float im} pairs.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

namespace host {

enum Content : uint16_t {
    kBinary = 0, kString, kTime, kFrequency, kWavelength, kSampleCount,
    kSignal, kComplexSignal, kSpectrum, kComplexSpectrum, kContentCount
};

// content-type names of the reference's command lines (include/sdr/options.hpp:220-232)
const std::vector<std::pair<std::string, int>>& content_choices();

struct PacketHeader {
    uint16_t id;
    uint16_t content;
    uint32_t size;
    uint64_t duration;
};
static_assert(sizeof(PacketHeader) == 16, "wire header is 16 bytes");

// Reads until `size` bytes arrived, EOF or error; returns bytes read.
size_t read_full(int fd, void* dst, size_t size);
// Writes everything or returns false on error.
bool write_full(int fd, const void* src, size_t size);
// Best effort: a 1 MiB kernel buffer for a pipe end (no effect on other descriptors).
void widen_pipe(int fd);
// Blocks that hold GPU state.  The reference's blocks die of SIGPIPE when their reader goes away.  A process
// with a live HIP runtime that is KILLED -- or leaves through _exit -- lingers ~0.2 s while the driver tears
// its queues down; only a normal exit(), which runs the runtime's own shutdown, ends at once (measured:
// profiles/r03_c1pipe_startup.txt -- hipDeviceReset before the signal does not help).  So a GPU block runs its
// body in a child process (supervised_main): SIGPIPE is ignored there, a write that fails with EPIPE makes the
// body release its handles and leave through end_output_closed() -- a normal exit with a status of its own --
// and the PARENT, which never touches the GPU and holds neither stdin nor stdout, then ends the way the
// reference ends: killed by SIGPIPE.  `gen | head -c N` takes 65 ms instead of 255.
void ignore_sigpipe();
[[noreturn]] void end_output_closed();
// Runs body(argc, argv) in a child and hands its exit status (or its fatal signal) on; SIGTERM / SIGINT /
// SIGHUP / SIGQUIT to the parent are passed to the child, and the child is killed if the parent dies.  Where
// fork fails the body runs unsupervised.
int supervised_main(int argc, char** argv, int (*body)(int, char**));
// True when a read on fd would not block right now (data or EOF pending).
bool input_ready(int fd);

// Packet routing without touching payloads (the reference's Source::pass / Source::drop,
// lib/sdr/stream.cpp:219-364, which splice packet by packet).  Two regimes:
//   * small packets (the ~8 KiB ones `gen` emits): the stream is read in large pieces, headers
//     are parsed in place, and every run of consecutive forwarded packets leaves in ONE gathered
//     write straight from the read buffer -- two copies per byte but a few system calls per
//     megabyte instead of three per packet;
//   * a payload that extends kZeroCopyFrom bytes or more beyond what has been read is moved
//     descriptor to descriptor like the reference does: splice(2) when an end is a pipe,
//     sendfile(2) from a regular file; a dropped one is seeked over (file) or spliced into
//     /dev/null (pipe).  No byte of it enters user space.
// A short final payload is forwarded as far as it arrived, behind its unchanged header; a short
// final header ends the stream.  Returns false if a write failed.
constexpr uint64_t kZeroCopyFrom = 64u << 10;
bool route_packets(int in_fd, int out_fd, bool (*forwards)(const PacketHeader&, void*), void* ctx);
// Bytes route_packets moved by each mechanism so far in this process (stream-filter prints them
// to stderr when SDR_AMD_ROUTE_STATS is set: tests and curiosity).
struct RouteStats {
    uint64_t spliced = 0, sent = 0, copied = 0, skipped = 0;
};
const RouteStats& route_stats();

// Sequential packet reader over a byte stream (semantics of the reference's
// Source::next/recv/drop, lib/sdr/stream.cpp:121-249: unread payload of the
// previous packet is discarded by next(); a short header is a clean EOF; a short
// payload yields what arrived and ends the stream).
// Reads are buffered: one read(2) fetches whatever the pipe holds (up to 1 MiB) and
// headers and payloads are served from that buffer -- with ~8 KiB packets a system call
// per header and per payload is what bounds a block's throughput otherwise.  A refill
// never waits for more than the first byte, so an interactive pipe sees no added latency.
class PacketReader {
public:
    explicit PacketReader(int fd = 0);
    // Advance to the next packet header.  False at end of stream.
    bool next();
    const PacketHeader& header() const { return hdr_; }
    // Read up to `bytes` of the remaining payload into dst; returns bytes read.
    size_t read_payload(void* dst, size_t bytes);
    size_t remaining() const { return hdr_.size - consumed_; }
    bool eof() const { return eof_; }
    int fd() const { return fd_; }
    // True when the next read would not block: bytes already buffered, or the
    // descriptor readable (data or end of input).
    bool input_pending() const;

private:
    size_t fetch(void* dst, size_t bytes); // from the buffer, then the descriptor; short only at end of input
    void skip_rest();
    int fd_;
    PacketHeader hdr_{};
    size_t consumed_ = 0;
    bool eof_ = false;
    std::vector<uint8_t> buf_;
    size_t pos_ = 0, end_ = 0;
};

// Gathers many small writes (header, payload, header, payload, ...) into few writev(2)
// calls.  The memory handed to add() must stay valid until flush().
class GatherWriter {
public:
    explicit GatherWriter(int fd) : fd_(fd) {}
    size_t pending() const { return parts_.size(); }
    void add(const void* p, size_t n);
    bool flush(); // false on a write error (the rest is dropped)

private:
    int fd_;
    std::vector<std::pair<const void*, size_t>> parts_;
};

class PacketWriter {
public:
    explicit PacketWriter(int fd = 1) : fd_(fd) {}
    // Header + payload (hdr.size bytes).  Write errors are ignored like the
    // reference's Sink::send (lib/sdr/stream.cpp:367-375); a closed pipe ends the
    // process by SIGPIPE.
    void send(const PacketHeader& hdr, const void* payload);
    void flush_to_disk(); // fdatasync when the sink is a regular file
    int fd() const { return fd_; }

private:
    int fd_;
};

} // namespace host
