#include "fir_pipeline.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <iostream>

namespace host {

namespace {
constexpr size_t kPoolSize = 4;
int report(const char* what) {
    std::cerr << "error: " << what << ": " << sdr_last_error() << std::endl;
    return -1;
}
} // namespace

FirPipeline::FirPipeline(const PipelineConfig& cfg) : cfg_(cfg), skip_left_(cfg.skip) {
    if (cfg_.batch_bytes < (64u << 10)) cfg_.batch_bytes = 64u << 10;
    cfg_.batch_bytes &= ~size_t(63);
    tail_.assign(2 * (cfg_.taps.size() > 0 ? cfg_.taps.size() - 1 : 0), 0.0f);
}

FirPipeline::~FirPipeline() {
    // order matters: the writer thread reads the batches, and an asynchronous device-to-host copy
    // of an unfinished batch may still target b.out -- end the thread, let sdr_fir_destroy
    // synchronise the handles' streams, and only then free the pinned batches
    if (writer_.joinable()) {
        {
            std::lock_guard<std::mutex> lk(mu_);
            done_ = true;
        }
        cv_.notify_all();
        writer_.join();
    }
    sdr_fir_destroy(fir_[0]);
    sdr_fir_destroy(fir_[1]);
    for (auto& b : pool_) {
        sdr_host_free(b.in);
        sdr_host_free(b.out);
    }
}

// A packet larger than the batch size gets a batch of its own, grown to fit (the wire
// format allows payloads up to 4 GiB; the reference reads any packet whole).
bool FirPipeline::grow(Batch& b, size_t bytes) {
    if (bytes <= b.cap_bytes) return true;
    bytes = (bytes + ((1u << 20) - 1)) & ~size_t((1u << 20) - 1);
    float* in = static_cast<float*>(sdr_host_alloc(bytes));
    float* out = static_cast<float*>(sdr_host_alloc(bytes));
    if (!in || !out) {
        sdr_host_free(in);
        sdr_host_free(out);
        return false;
    }
    sdr_host_free(b.in);
    sdr_host_free(b.out);
    b.in = in;
    b.out = out;
    b.cap_bytes = bytes;
    return true;
}

bool FirPipeline::is_filtered(const PacketHeader& h) const {
    if (h.content != kSignal && h.content != kComplexSignal) return false;
    if (cfg_.select) return cfg_.select(h, cfg_.select_ctx);
    return h.id == cfg_.stream_id;
}

FirPipeline::Batch* FirPipeline::take_free() {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return !free_.empty(); });
    Batch* b = free_.front();
    free_.pop_front();
    b->items.clear();
    b->raw.clear();
    b->samples = 0;
    b->slot = -1;
    return b;
}

void FirPipeline::give_to_writer(Batch* b) {
    {
        std::lock_guard<std::mutex> lk(mu_);
        ready_.push_back(b);
    }
    cv_.notify_all();
}

// The reference keeps ONE delay line for a stream id, whatever the packet
// content: real packets are widened to complex before kfr::fir
// (blocks/hilbert.cpp:60-68).  Two device states (real, complex) are kept here
// for bandwidth; when the content type of the stream changes, the new state is
// primed with the last ntaps-1 inputs so the history continues seamlessly.
void FirPipeline::remember_tail(const Batch& b) {
    const size_t H = cfg_.taps.size() - 1;
    if (H == 0 || b.samples == 0) return;
    const size_t take = std::min(H, b.samples);
    if (take < H) std::memmove(tail_.data(), tail_.data() + 2 * take, 2 * (H - take) * sizeof(float));
    float* dst = tail_.data() + 2 * (H - take);
    if (b.complex_samples) {
        std::memcpy(dst, b.in + 2 * (b.samples - take), 2 * take * sizeof(float));
    } else {
        const float* src = b.in + (b.samples - take);
        for (size_t i = 0; i < take; ++i) {
            dst[2 * i] = src[i];
            dst[2 * i + 1] = 0.0f;
        }
    }
    tail_len_ = std::min(H, tail_len_ + take);
}

int FirPipeline::switch_kind(bool to_complex) {
    const int k = to_complex ? 1 : 0;
    if (!fir_[k]) {
        // SDR_AMD_NUMERICS=per-output: the direct-form kernels whatever the tap count -- kfr::fir's own error
        // character (each output rounded relative to itself, a NaN / Inf sample reaches ntaps outputs) at the
        // direct form's cost; the default picks the quickest kernel family (include/sdr_amd.h)
        const char* nm = std::getenv("SDR_AMD_NUMERICS");
        const int numerics = (nm && std::strcmp(nm, "per-output") == 0) ? SDR_NUMERICS_PER_OUTPUT : SDR_NUMERICS_FAST;
        if (nm && *nm && numerics == SDR_NUMERICS_FAST && std::strcmp(nm, "fast") != 0)
            return report("SDR_AMD_NUMERICS must be fast or per-output");
        if (sdr_fir_create_numerics(&fir_[k], cfg_.taps.data(), cfg_.taps.size(), to_complex ? SDR_COMPLEX : SDR_REAL, 1,
                                    cfg_.device, numerics) != SDR_OK)
            return report("cannot create the GPU filter");
    }
    if (have_kind_ && cur_complex_ != to_complex) {
        // everything queued on the old state has to be finished by the caller
        // before this point; re-create the history on the new state
        const size_t H = cfg_.taps.size() - 1;
        if (sdr_fir_reset(fir_[k]) != SDR_OK) return report("sdr_fir_reset");
        if (H) {
            std::vector<float> in(to_complex ? 2 * H : H), out(to_complex ? 2 * H : H);
            if (to_complex) in = tail_;
            else for (size_t i = 0; i < H; ++i) in[i] = tail_[2 * i];
            if (sdr_fir_process(fir_[k], in.data(), H, out.data()) != SDR_OK) return report("sdr_fir_process");
        }
    }
    have_kind_ = true;
    cur_complex_ = to_complex;
    return 0;
}

int FirPipeline::launch(Batch& b) {
    if (b.samples == 0) return 0;
    b.slot = next_slot_;
    next_slot_ ^= 1;
    if (sdr_fir_submit(fir_[b.complex_samples ? 1 : 0], b.slot, b.in, b.samples, b.out) != SDR_OK)
        return report("sdr_fir_submit");
    remember_tail(b);
    return 0;
}

int FirPipeline::finish(Batch* b) {
    if (b->slot >= 0 && sdr_fir_wait(fir_[b->complex_samples ? 1 : 0], b->slot) != SDR_OK)
        return report("sdr_fir_wait");
    give_to_writer(b);
    return 0;
}

void FirPipeline::writer_loop(int out_fd) {
    PacketWriter w(out_fd);
    widen_pipe(out_fd);
    for (;;) {
        Batch* b;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return !ready_.empty() || done_; });
            if (ready_.empty()) break;
            b = ready_.front();
            ready_.pop_front();
        }
        const size_t elt = b->complex_samples ? 8 : 4;
        // headers of the filtered packets are rewritten: they live here until the flush
        std::vector<PacketHeader> hdrs;
        hdrs.reserve(b->items.size());
        std::vector<uint8_t> zeros;
        GatherWriter out(out_fd);
        for (const Item& it : b->items) {
            if (!it.filtered) {
                // forwarded untouched; a truncated payload stays truncated
                out.add(&it.hdr, sizeof it.hdr);
                out.add(b->raw.data() + it.offset, it.count);
                continue;
            }
            // Group-delay compensation, per packet, with the reference's integer
            // arithmetic (blocks/hilbert.cpp:75-86): skip = min(n, left);
            // size -= skip*elt; duration -= skip*duration/n; empty => swallowed.
            PacketHeader h = it.hdr;
            const size_t n = it.count;
            const size_t skip = std::min(n, skip_left_);
            if (skip > 0) {
                skip_left_ -= skip;
                h.size -= static_cast<uint32_t>(skip * elt);
                h.duration -= (skip * h.duration) / n;
                if (h.size == 0) continue;
            }
            if (n - skip) h.size = static_cast<uint32_t>((n - skip) * elt);
            const uint8_t* payload = reinterpret_cast<const uint8_t*>(b->out) + (it.offset + skip) * elt;
            hdrs.push_back(h);
            out.add(&hdrs.back(), sizeof h);
            if (n - skip) {
                out.add(payload, (n - skip) * elt);
            } else if (h.size) {
                // no whole sample arrived (size not a multiple of the sample size, or
                // EOF right after the header): the reference sends the header with
                // its size unchanged followed by unspecified bytes
                // (include/sdr/stream.hpp:133-139); zeros keep the stream framed.
                if (zeros.size() < h.size) {
                    (void)out.flush(); // `zeros` may move
                    zeros.assign(h.size, 0);
                }
                out.add(zeros.data(), h.size);
            }
        }
        if (!out.flush()) output_closed_.store(true, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(mu_);
            free_.push_back(b);
        }
        cv_.notify_all();
    }
    w.flush_to_disk();
}

int FirPipeline::run(int in_fd, int out_fd) {
    if (sdr_device_count() < 1) {
        std::cerr << "error: no HIP device found (this block has no CPU path)" << std::endl;
        return -1;
    }
    pool_.resize(kPoolSize);
    for (auto& b : pool_) {
        if (!grow(b, cfg_.batch_bytes)) return report("pinned host allocation failed");
        free_.push_back(&b);
    }
    writer_ = std::thread([this, out_fd] { writer_loop(out_fd); });

    PacketReader rd(in_fd);
    Batch* cur = take_free();
    Batch* inflight = nullptr;
    int rc = 0;

    auto flush = [&](bool wait_now) -> int {
        // hand `cur` on: queue it on the GPU (if it has samples), retire the batch
        // that was in flight before it, and start a fresh one
        if (cur->items.empty()) {
            if (wait_now && inflight) {
                if (int r = finish(inflight)) return r;
                inflight = nullptr;
            }
            return 0;
        }
        if (cur->samples) {
            if (inflight && inflight->complex_samples != cur->complex_samples) {
                if (int r = finish(inflight)) return r;
                inflight = nullptr;
            }
            if (!have_kind_ || cur_complex_ != cur->complex_samples)
                if (int r = switch_kind(cur->complex_samples)) return r;
            if (int r = launch(*cur)) return r;
        }
        if (inflight) {
            if (int r = finish(inflight)) return r;
            inflight = nullptr;
        }
        if (wait_now) {
            if (int r = finish(cur)) return r;
        } else {
            inflight = cur;
        }
        cur = take_free();
        return 0;
    };

    ignore_sigpipe();
    while (rd.next()) {
        if (output_closed_.load(std::memory_order_relaxed)) break;
        const PacketHeader h = rd.header();
        if (!is_filtered(h)) {
            const bool fwd = cfg_.forward_other ? cfg_.forward_other(h, cfg_.forward_ctx) : true;
            if (!fwd) continue;
            if (cur->raw.size() + h.size > (64u << 20))
                if ((rc = flush(false))) break;
            Item it{h, false, cur->raw.size(), 0};
            if (!cur->raw.resize(it.offset + h.size)) {
                std::cerr << "error: out of memory forwarding a " << h.size << "-byte packet" << std::endl;
                rc = -1;
                break;
            }
            it.count = rd.read_payload(cur->raw.data() + it.offset, h.size);
            (void)cur->raw.resize(it.offset + it.count);
            cur->items.push_back(it);
        } else {
            const bool cplx = h.content == kComplexSignal;
            const size_t elt = cplx ? 8 : 4;
            // Packet::count<T>() is 0 unless size is a whole number of samples
            const size_t want = (h.size % elt) ? 0 : h.size / elt;
            if (cur->samples && (cur->complex_samples != cplx || (cur->samples + want) * elt > cur->cap_bytes))
                if ((rc = flush(false))) break;
            if (want * elt > cur->cap_bytes) {
                // cur holds no samples here (just flushed, or fresh) -- but it may hold
                // forwarded packets whose order must be kept: hand those on first
                if (!cur->items.empty())
                    if ((rc = flush(false))) break;
                if (!grow(*cur, want * elt)) {
                    std::cerr << "error: cannot allocate a " << want * elt << "-byte pinned buffer for one packet" << std::endl;
                    rc = -1;
                    break;
                }
            }
            if (cur->samples == 0) cur->complex_samples = cplx;
            uint8_t* dst = reinterpret_cast<uint8_t*>(cur->in) + cur->samples * elt;
            const size_t got = rd.read_payload(dst, want * elt) / elt;
            cur->items.push_back(Item{h, true, cur->samples, got});
            cur->samples += got;
        }
        // nothing more to read right now: do not sit on a partial batch
        if (!rd.input_pending())
            if ((rc = flush(true))) break;
    }
    if (!rc) rc = flush(true);
    if (inflight && !rc) rc = finish(inflight);
    if (rc) {
        // error exit with a batch possibly still queued on the GPU: wait for both staging slots
        // before anything that owns their buffers goes away (errors here change nothing)
        for (sdr_fir_t* f : fir_)
            if (f) {
                (void)sdr_fir_wait(f, 0);
                (void)sdr_fir_wait(f, 1);
            }
    }
    {
        std::lock_guard<std::mutex> lk(mu_);
        done_ = true;
    }
    cv_.notify_all();
    writer_.join();
    if (!rc && output_closed_.load()) rc = kOutputClosed;
    return rc;
}

} // namespace host
