// fft_call.h -- internal: what the overlap-save kernels of both tile sizes (fir_fft.hip, fir_fft16k.hip) share at the
// edges of a call: the history carry and the end-of-call epilogue that moves the handle's device state on.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "fir_kernels.h"

namespace sdr_amd {

// ---- history carry: after the call the other history buffer holds the last hist_len
// samples of the virtual stream (old history | this call's input or synthesised samples).
// Reads only what the call's other workgroups read; writes a buffer none of them touches.
// Which buffer is which is device state (FirDevState::cur), flipped by call_done below.
template <bool GEN, int LANES = 256> // LANES: the workgroup's size
__device__ __forceinline__ void carry_history(const FftArgs& a, const float* old_hist, float* new_hist, const int w, const int bid,
                                              const int nblk) {
    const long long total = (long long)a.hist_len * w * a.channels;
    for (long long i = (long long)bid * LANES + threadIdx.x; i < total; i += (long long)nblk * LANES) {
        const long long ch = i / (a.hist_len * w);
        const int r = (int)(i - ch * (a.hist_len * w));
        const int s = r / w, k = r - s * w;
        const long long m = a.n - a.hist_len + s;
        float v;
        if (m < 0) v = old_hist[(ch * a.hist_len + (a.hist_len + m)) * w + k];
        else if constexpr (GEN) v = gen_src_at(a.gen, m);
        else v = a.in[(ch * a.in_stride + m) * w + k];
        new_hist[i] = v;
    }
}

// ---- end of a call (every workgroup of the call's last launch, at its very end).  The workgroup that
// finishes last leaves the handle's device state ready for the next launch on the stream: ticket
// counters back to zero, history buffers swapped.  Nothing of this is host state, so a captured
// graph of one call replays as the stream's next call (tests/test_graph_capture_gpu.py).
// What it needs -- the state and ticket pointers -- it reads again from the kernel-argument segment through a
// pointer the compiler cannot see through, instead of taking the caller's copies: values kept alive across the
// persistent tile loop for this epilogue cost that loop scalar registers it does not have (it then re-loaded
// `channels` from the argument segment in every tile, behind an s_waitcnt that also waits for the LDS: C2
// 0.771 -> 0.798 ms, profiles/r03_device_state_ab.txt).  The kernels take FftArgs as their only parameter.
__device__ __forceinline__ void call_done() {
    __syncthreads(); // every wave of the workgroup is past its reads of the state, the history and its tickets
    if (threadIdx.x == 0) {
        const char __attribute__((address_space(4)))* ka =
            (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        FirDevState* const st = *reinterpret_cast<FirDevState* const __attribute__((address_space(4)))*>(ka + offsetof(FftArgs, state));
        if (last_workgroup_done(st, gridDim.x)) {
            unsigned* const tickets = *reinterpret_cast<unsigned* const __attribute__((address_space(4)))*>(ka + offsetof(FftArgs, tickets));
            if (tickets) {
#pragma unroll
                for (int t = 0; t < FFT_TICKETS; ++t) tickets[FFT_TICKET_STRIDE * t] = 0;
            }
            st->done = 0;
            st->cur = (st->cur & 1u) ^ 1u; // unchanged since the launch began: only this workgroup, here, writes it
            // a fused generator with closed-form phases moves on by this call's blocks (GenSrc::pstate)
            GenDevState* const gs = *reinterpret_cast<GenDevState* const __attribute__((address_space(4)))*>(
                ka + offsetof(FftArgs, gen) + offsetof(GenSrc, pstate));
            if (gs) {
                const unsigned pincr = *reinterpret_cast<const unsigned __attribute__((address_space(4)))*>(
                    ka + offsetof(FftArgs, gen) + offsetof(GenSrc, pincr));
                const unsigned nblocks = *reinterpret_cast<const unsigned __attribute__((address_space(4)))*>(
                    ka + offsetof(FftArgs, gen) + offsetof(GenSrc, nblocks));
                gs->phase = (gs->phase + nblocks * pincr) & 0x7fffffu;
            }
        }
    }
}

} // namespace sdr_amd
