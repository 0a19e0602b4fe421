// fir_pipeline.hpp -- packet stream -> GPU FIR -> packet stream, in arrival order.
//
// The reference blocks handle one ~8 KiB packet at a time: read, kfr::fir, write
// (blocks/hilbert.cpp:53-96).  A GPU wants megabytes per launch, so this pipeline
// coalesces consecutive packets of the filtered stream into one batch held in
// pinned host memory, runs it through libsdr_amd (H2D, FIR kernel, D2H, queued
// asynchronously on the handle's stream) while the next batch is being read from
// the pipe, and a writer thread emits finished batches while the GPU works on the
// following one.  Packet headers, the group-delay skip and their integer
// arithmetic stay on the host and are applied per packet exactly like the
// reference does.  Packets that are not filtered (other stream ids / contents)
// are carried through the same queue so the interleaving on output equals the
// interleaving on input.  A batch is flushed when full, when the content type
// changes, when no more input is pending (latency), and at end of stream.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <mutex>
#include <thread>
#include <cstdlib>
#include <vector>

#include "../../include/sdr_amd.h"
#include "wire.hpp"

namespace host {

struct PipelineConfig {
    std::vector<float> taps;
    uint16_t stream_id = 0;
    bool any_stream = false;      // filter every Signal/ComplexSignal packet (stream-filter fir=)
    std::vector<uint16_t> ids;    // with any_stream=false and non-empty: filter these ids
    size_t skip = 0;              // outputs dropped at the start of the stream (group delay)
    size_t batch_bytes = 8u << 20;
    int device = 0;
    // Decides what happens to a packet that is not filtered: true = forward
    // unchanged, false = discard.  Null = forward everything.
    bool (*forward_other)(const PacketHeader&, void*) = nullptr;
    void* forward_ctx = nullptr;
    // Decides whether a Signal/ComplexSignal packet is filtered at all.
    bool (*select)(const PacketHeader&, void*) = nullptr;
    void* select_ctx = nullptr;
};

// Growable byte buffer that does not zero what it grows by (a std::vector would write every
// forwarded payload byte once more before the read overwrites it).
class RawBytes {
public:
    RawBytes() = default;
    RawBytes(const RawBytes&) = delete;
    RawBytes& operator=(const RawBytes&) = delete;
    RawBytes(RawBytes&& o) noexcept : p_(o.p_), size_(o.size_), cap_(o.cap_) { o.p_ = nullptr; o.size_ = o.cap_ = 0; }
    ~RawBytes() { std::free(p_); }
    size_t size() const { return size_; }
    uint8_t* data() { return p_; }
    const uint8_t* data() const { return p_; }
    void clear() { size_ = 0; }
    // false when memory runs out (size unchanged)
    bool resize(size_t n) {
        if (n > cap_) {
            size_t cap = cap_ ? cap_ : (1u << 20);
            while (cap < n) cap *= 2;
            void* q = std::realloc(p_, cap);
            if (!q) return false;
            p_ = static_cast<uint8_t*>(q);
            cap_ = cap;
        }
        size_ = n;
        return true;
    }

private:
    uint8_t* p_ = nullptr;
    size_t size_ = 0, cap_ = 0;
};

class FirPipeline {
public:
    explicit FirPipeline(const PipelineConfig& cfg);
    ~FirPipeline();
    // Runs until end of input.  Returns 0, or -1 after printing an error.
    // 0 = end of input; kOutputClosed = the output went away (the caller ends by SIGPIPE); other = error
    static constexpr int kOutputClosed = 141;
    int run(int in_fd, int out_fd);

private:
    struct Item {
        PacketHeader hdr;
        bool filtered;
        size_t offset; // filtered: first sample in the batch; raw: byte offset in raw
        size_t count;  // filtered: samples read; raw: payload bytes present
    };
    struct Batch {
        float* in = nullptr;  // pinned
        float* out = nullptr; // pinned
        RawBytes raw; // payloads of the forwarded packets, in arrival order
        std::vector<Item> items;
        size_t samples = 0;
        size_t cap_bytes = 0; // size of `in` and of `out`
        bool complex_samples = false;
        int slot = -1;
    };
    bool grow(Batch& b, size_t bytes); // pinned buffers of at least `bytes` (contents dropped)

    bool is_filtered(const PacketHeader& h) const;
    int launch(Batch& b);           // queue on the GPU
    int finish(Batch* b);           // wait for the GPU, hand to the writer
    void remember_tail(const Batch& b);
    int switch_kind(bool to_complex);
    void writer_loop(int out_fd);
    Batch* take_free();
    void give_to_writer(Batch* b);

    PipelineConfig cfg_;
    sdr_fir_t* fir_[2] = {nullptr, nullptr}; // [0] real, [1] complex, created on first use
    bool have_kind_ = false, cur_complex_ = false;
    std::vector<float> tail_; // last ntaps-1 inputs as complex pairs (shared delay line semantics)
    size_t tail_len_ = 0;
    int next_slot_ = 0;

    std::vector<Batch> pool_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Batch*> free_, ready_;
    bool done_ = false;
    std::atomic<bool> output_closed_{false}; // a write failed (EPIPE): stop reading, release the GPU
    size_t skip_left_;
    std::thread writer_;
};

} // namespace host
