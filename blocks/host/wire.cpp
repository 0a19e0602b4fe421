#include "wire.hpp"

#include <cerrno>
#include <poll.h>
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

void PacketReader::skip_rest() {
    uint8_t sink[4096];
    while (!eof_ && consumed_ < hdr_.size) {
        size_t want = hdr_.size - consumed_;
        if (want > sizeof sink) want = sizeof sink;
        size_t got = read_full(fd_, sink, want);
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
    if (read_full(fd_, &h, sizeof h) < sizeof h) {
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
    size_t got = read_full(fd_, dst, bytes);
    if (got < bytes) eof_ = true;
    consumed_ += got;
    return got;
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
