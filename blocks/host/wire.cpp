#ifndef _GNU_SOURCE
#define _GNU_SOURCE // splice(2)
#endif
#include "wire.hpp"

#include <algorithm>
#include <cerrno>
#include <fcntl.h>
#include <poll.h>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <sys/sendfile.h>
#include <sys/uio.h>
#include <sys/prctl.h>
#include <sys/stat.h>
#include <sys/wait.h>
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

namespace {

// What kind of descriptor (decides which zero-copy call applies).
struct FdKind {
    bool fifo = false, regular = false;
};
FdKind kind_of(int fd) {
    FdKind k;
    struct stat st;
    if (::fstat(fd, &st) == 0) {
        k.fifo = S_ISFIFO(st.st_mode);
        k.regular = S_ISREG(st.st_mode);
    }
    return k;
}

RouteStats g_route_stats;

// Moves up to n payload bytes from in_fd to out_fd without bringing them into user space:
// splice(2) when either end is a pipe, sendfile(2) from a regular file (the reference's
// Source::pass, lib/sdr/stream.cpp:251-296); a plain copy loop where neither applies (terminal,
// socket pairs the kernel refuses).  Returns bytes moved (short at end of input); *ok = false on
// a write error.
uint64_t move_payload(int in_fd, int out_fd, uint64_t n, const FdKind& in, const FdKind& out, std::vector<uint8_t>& buf,
                      bool* ok) {
    uint64_t moved = 0;
    bool try_splice = in.fifo || out.fifo, try_sendfile = in.regular;
    while (moved < n) {
        const size_t want = static_cast<size_t>(std::min<uint64_t>(n - moved, 1u << 30));
        ssize_t r = -1;
        if (try_splice) {
            r = ::splice(in_fd, nullptr, out_fd, nullptr, want, SPLICE_F_MOVE | (n - moved > want ? SPLICE_F_MORE : 0));
            if (r < 0 && errno == EINTR) continue;
            if (r < 0 && (errno == EINVAL || errno == ENOSYS || errno == EBADF)) { try_splice = false; continue; }
            if (r > 0) g_route_stats.spliced += static_cast<uint64_t>(r);
        } else if (try_sendfile) {
            r = ::sendfile(out_fd, in_fd, nullptr, want);
            if (r < 0 && errno == EINTR) continue;
            if (r < 0 && (errno == EINVAL || errno == ENOSYS)) { try_sendfile = false; continue; }
            if (r > 0) g_route_stats.sent += static_cast<uint64_t>(r);
        } else {
            const size_t piece = std::min(want, buf.size());
            r = ::read(in_fd, buf.data(), piece);
            if (r < 0 && errno == EINTR) continue;
            if (r > 0) {
                if (!write_full(out_fd, buf.data(), static_cast<size_t>(r))) { *ok = false; return moved; }
                g_route_stats.copied += static_cast<uint64_t>(r);
            }
        }
        if (r < 0) {
            if (errno == EAGAIN || errno == EWOULDBLOCK) {
                // a non-blocking descriptor with nothing to take or no room: wait for either end, like the buffered
                // path's blocking read / write would, and go again
                struct pollfd p[2] = {{in_fd, POLLIN, 0}, {out_fd, POLLOUT, 0}};
                (void)::poll(p, 2, 100);
                continue;
            }
            // any other failure no fallback above has absorbed is an error of the move, never a quiet end of
            // the stream in mid-packet
            *ok = false;
            break;
        }
        if (r == 0) break; // end of input inside the payload
        moved += static_cast<uint64_t>(r);
    }
    return moved;
}

// Discards up to n payload bytes of in_fd (the reference's Source::drop, lib/sdr/stream.cpp:219-249):
// a seek on a regular file, a splice into /dev/null on a pipe, read-and-forget otherwise.
uint64_t skip_payload(int in_fd, uint64_t n, const FdKind& in, std::vector<uint8_t>& buf) {
    if (in.regular) {
        const off_t here = ::lseek(in_fd, 0, SEEK_CUR);
        struct stat st;
        if (here >= 0 && ::fstat(in_fd, &st) == 0) {
            const uint64_t left = st.st_size > here ? static_cast<uint64_t>(st.st_size - here) : 0;
            const uint64_t take = std::min(n, left);
            if (::lseek(in_fd, static_cast<off_t>(take), SEEK_CUR) >= 0) {
                g_route_stats.skipped += take;
                return take;
            }
        }
    }
    static int null_fd = -2;
    if (in.fifo && null_fd == -2) null_fd = ::open("/dev/null", O_WRONLY);
    uint64_t done = 0;
    bool try_splice = in.fifo && null_fd >= 0;
    while (done < n) {
        const size_t want = static_cast<size_t>(std::min<uint64_t>(n - done, 1u << 30));
        ssize_t r;
        if (try_splice) {
            r = ::splice(in_fd, nullptr, null_fd, nullptr, want, SPLICE_F_MOVE);
            if (r < 0 && errno == EINTR) continue;
            if (r < 0) { try_splice = false; continue; }
        } else {
            r = ::read(in_fd, buf.data(), std::min(want, buf.size()));
            if (r < 0 && errno == EINTR) continue;
        }
        if (r <= 0) break;
        done += static_cast<uint64_t>(r);
        g_route_stats.skipped += static_cast<uint64_t>(r);
    }
    return done;
}

} // namespace

const RouteStats& route_stats() { return g_route_stats; }

bool route_packets(int in_fd, int out_fd, bool (*forwards)(const PacketHeader&, void*), void* ctx) {
    widen_pipe(in_fd);
    widen_pipe(out_fd);
    const FdKind in_kind = kind_of(in_fd), out_kind = kind_of(out_fd);
    std::vector<uint8_t> buf(4u << 20);
    GatherWriter out(out_fd);
    size_t end = 0;          // valid bytes in buf
    uint64_t payload_left = 0; // bytes of the current packet's payload still to come
    bool forwarding = false;   // ... and whether they are forwarded
    bool ok = true;
    for (;;) {
        // A payload that reaches well beyond what has been read never enters user space: it is
        // spliced (or sent, or seeked over) descriptor to descriptor -- the reference's zero-copy
        // pass / drop (lib/sdr/stream.cpp:219-296), applied where it pays: packets of ~8 KiB are
        // cheaper through the gathered path below (a few system calls per megabyte instead of
        // three per packet).
        if (end == 0 && payload_left >= kZeroCopyFrom) {
            const uint64_t got = forwarding ? move_payload(in_fd, out_fd, payload_left, in_kind, out_kind, buf, &ok)
                                            : skip_payload(in_fd, payload_left, in_kind, buf);
            if (!ok || got < payload_left) break; // write error, or the input ended inside the payload
            payload_left = 0;
            continue;
        }
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

namespace {
constexpr int kOutputClosedStatus = 141; // child -> parent: "my reader went away" (what a shell shows for SIGPIPE)
bool g_supervised = false;
volatile sig_atomic_t g_child = 0;

[[noreturn]] void die_of_sigpipe() {
    (void)::signal(SIGPIPE, SIG_DFL);
    (void)::raise(SIGPIPE);
    ::_exit(kOutputClosedStatus); // not reached unless SIGPIPE is blocked
}

void pass_signal_on(int sig) {
    if (g_child > 0) (void)::kill(static_cast<pid_t>(g_child), sig);
}
} // namespace

void end_output_closed() {
    std::fflush(nullptr);
    if (g_supervised) std::exit(kOutputClosedStatus); // normal exit: the HIP runtime shuts itself down; the parent dies of SIGPIPE
    die_of_sigpipe();
}

int supervised_main(int argc, char** argv, int (*body)(int, char**)) {
    std::fflush(nullptr);
    const pid_t parent = ::getpid();
    const pid_t pid = ::fork();
    if (pid < 0) return body(argc, argv);
    if (pid == 0) {
        (void)::prctl(PR_SET_PDEATHSIG, SIGKILL);
        if (::getppid() != parent) ::_exit(1); // the parent is gone already
        g_supervised = true;
        const int rc = body(argc, argv);
        std::fflush(nullptr);
        std::exit(rc & 0xff);
    }
    g_child = pid;
    for (int sig : {SIGTERM, SIGINT, SIGHUP, SIGQUIT}) (void)::signal(sig, pass_signal_on);
    (void)::signal(SIGPIPE, SIG_IGN);
    // the streams belong to the child: its reader sees the end of the stream, and its writer the closed pipe,
    // the moment the CHILD goes
    (void)::close(0);
    (void)::close(1);
    int status = 0;
    while (::waitpid(pid, &status, 0) < 0) {
        if (errno != EINTR) return 1;
    }
    if (WIFSIGNALED(status)) {
        (void)::signal(WTERMSIG(status), SIG_DFL);
        (void)::raise(WTERMSIG(status));
        return 128 + WTERMSIG(status);
    }
    if (WEXITSTATUS(status) == kOutputClosedStatus) die_of_sigpipe();
    return WEXITSTATUS(status);
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
