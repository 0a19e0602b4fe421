#include "wire.hpp"

#include <algorithm>
#include <cerrno>
#include <fcntl.h>
#include <poll.h>
#include <csignal>
#include <sys/uio.h>
#include <sys/stat.h>
#include <unistd.h>

namespace host {

const std::vector<std::pair<std::string, int>>& content_choices() {
    static const std::vector<std::pair<std::string, int>> k = {
        {"binary", kBinary}, {"string", kString}, {"time", kTime}, {"frequency", kFrequency},
        {"wavelength", kWavelength}, {"sample_count", kSampleCount}, {"signal", kSignal},
        {"complex_signal", kComplexSignal}, {"spectrum", kSpectrum}, {"complex_spectrum", kComplexSpectrum}};
    return k;
}

size_t read_full(int fd, void* dst, size_t size) {
    uint8_t* p = static_cast<uint8_t*>(dst);
    size_t got = 0;
    while (got < size) {
        ssize_t r = ::read(fd, p + got, size - got);
        if (r < 0) {
            if (errno == EINTR) continue;
            break;
        }
        if (r == 0) break;
        got += static_cast<size_t>(r);
    }
    return got;
}

bool write_full(int fd, const void* src, size_t size) {
    const uint8_t* p = static_cast<const uint8_t*>(src);
    size_t put = 0;
    while (put < size) {
        ssize_t w = ::write(fd, p + put, size - put);
        if (w < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        if (w == 0) return false;
        put += static_cast<size_t>(w);
    }
    return true;
}

bool input_ready(int fd) {
    struct pollfd p;
    p.fd = fd;
    p.events = POLLIN;
    p.revents = 0;
    int r = ::poll(&p, 1, 0);
    return r != 0; // data, hangup or error: a read will not block
}

// A pipe end asks for a 1 MiB kernel buffer (the default is 64 KiB): fewer wake-ups per
// megabyte for both neighbours.  Best effort; not a FIFO or not permitted = no change.
void widen_pipe(int fd) {
#ifdef F_SETPIPE_SZ
    struct stat st;
    if (::fstat(fd, &st) == 0 && S_ISFIFO(st.st_mode)) (void)::fcntl(fd, F_SETPIPE_SZ, 1 << 20);
#else
    (void)fd;
#endif
}

bool route_packets(int in_fd, int out_fd, bool (*forwards)(const PacketHeader&, void*), void* ctx) {
    widen_pipe(in_fd);
    widen_pipe(out_fd);
    std::vector<uint8_t> buf(4u << 20);
    GatherWriter out(out_fd);
    size_t end = 0;          // valid bytes in buf
    uint64_t payload_left = 0; // bytes of the current packet's payload still to come
    bool forwarding = false;   // ... and whether they are forwarded
    bool ok = true;
    for (;;) {
        ssize_t r = ::read(in_fd, buf.data() + end, buf.size() - end);
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) break; // end of stream (a partial header in buf[0, end) is dropped)
        end += static_cast<size_t>(r);
        size_t pos = 0, run = 0; // buf[run, pos) = forwarded bytes not yet handed to the writer
        auto close_run = [&](size_t upto) {
            if (forwarding && upto > run) out.add(buf.data() + run, upto - run);
        };
        while (pos < end) {
            if (payload_left) {
                const size_t take = static_cast<size_t>(std::min<uint64_t>(payload_left, end - pos));
                pos += take;
                payload_left -= take;
                continue;
            }
            if (end - pos < sizeof(PacketHeader)) break;
            PacketHeader h;
            std::memcpy(&h, buf.data() + pos, sizeof h);
            const bool f = forwards(h, ctx);
            if (f != forwarding) {
                close_run(pos);
                forwarding = f;
                run = pos;
            }
            pos += sizeof h;
            payload_left = h.size;
        }
        close_run(pos);
        if (!out.flush()) { ok = false; break; }
        // keep the partial header (fewer than 16 bytes) for the next read
        const size_t rest = end - pos;
        if (rest) std::memmove(buf.data(), buf.data() + pos, rest);
        end = rest;
        // (the run restarts at the buffer's start: what is forwarded next is new data)
    }
    return ok;
}

void ignore_sigpipe() { (void)::signal(SIGPIPE, SIG_IGN); }

void die_of_sigpipe() {
    (void)::signal(SIGPIPE, SIG_DFL);
    (void)::raise(SIGPIPE);
    ::_exit(141); // not reached unless SIGPIPE is blocked
}

PacketReader::PacketReader(int fd) : fd_(fd), buf_(1u << 20) { widen_pipe(fd); }

size_t PacketReader::fetch(void* dst, size_t bytes) {
    uint8_t* p = static_cast<uint8_t*>(dst);
    size_t got = 0;
    while (got < bytes) {
        if (pos_ < end_) {
            const size_t take = std::min(bytes - got, end_ - pos_);
            if (p) std::memcpy(p + got, buf_.data() + pos_, take);
            pos_ += take;
            got += take;
            continue;
        }
        // buffer empty: a large remainder goes straight to its destination, anything
        // else refills the buffer with whatever one read() returns
        ssize_t r;
        if (p && bytes - got >= buf_.size()) {
            r = ::read(fd_, p + got, bytes - got);
            if (r > 0) got += static_cast<size_t>(r);
        } else {
            r = ::read(fd_, buf_.data(), buf_.size());
            if (r > 0) { pos_ = 0; end_ = static_cast<size_t>(r); }
        }
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) break;
    }
    return got;
}

bool PacketReader::input_pending() const { return pos_ < end_ || input_ready(fd_); }

void PacketReader::skip_rest() {
    if (!eof_ && consumed_ < hdr_.size) {
        const size_t want = hdr_.size - consumed_;
        const size_t got = fetch(nullptr, want);
        consumed_ += got;
        if (got < want) eof_ = true;
    }
}

bool PacketReader::next() {
    skip_rest();
    consumed_ = 0;
    if (eof_) {
        hdr_ = PacketHeader{};
        return false;
    }
    PacketHeader h;
    if (fetch(&h, sizeof h) < sizeof h) {
        hdr_ = PacketHeader{};
        eof_ = true;
        return false;
    }
    hdr_ = h;
    return true;
}

size_t PacketReader::read_payload(void* dst, size_t bytes) {
    size_t left = hdr_.size - consumed_;
    if (bytes > left) bytes = left;
    if (bytes == 0 || eof_) return 0;
    size_t got = fetch(dst, bytes);
    if (got < bytes) eof_ = true;
    consumed_ += got;
    return got;
}

void GatherWriter::add(const void* p, size_t n) {
    if (n) parts_.emplace_back(p, n);
    if (parts_.size() >= 1000) (void)flush();
}

bool GatherWriter::flush() {
    bool ok = true;
    size_t i = 0;
    size_t done_in_first = 0; // bytes of parts_[i] already written
    while (ok && i < parts_.size()) {
        struct iovec iov[512];
        int n = 0;
        for (size_t j = i; j < parts_.size() && n < 512; ++j, ++n) {
            const size_t off = j == i ? done_in_first : 0;
            iov[n].iov_base = const_cast<uint8_t*>(static_cast<const uint8_t*>(parts_[j].first)) + off;
            iov[n].iov_len = parts_[j].second - off;
        }
        ssize_t w = ::writev(fd_, iov, n);
        if (w < 0) {
            if (errno == EINTR) continue;
            ok = false;
            break;
        }
        size_t left = static_cast<size_t>(w);
        while (i < parts_.size() && left >= parts_[i].second - done_in_first) {
            left -= parts_[i].second - done_in_first;
            done_in_first = 0;
            ++i;
        }
        done_in_first += left;
        if (w == 0) { ok = false; break; }
    }
    parts_.clear();
    return ok;
}

void PacketWriter::send(const PacketHeader& hdr, const void* payload) {
    (void)write_full(fd_, &hdr, sizeof hdr);
    if (hdr.size) (void)write_full(fd_, payload, hdr.size);
}

void PacketWriter::flush_to_disk() {
    struct stat st;
    if (::fstat(fd_, &st) == 0 && S_ISREG(st.st_mode)) (void)::fdatasync(fd_);
}

} // namespace host
