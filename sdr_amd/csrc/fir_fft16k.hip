// fir_fft16k.hip -- overlap-save FIR with an in-LDS 16384-point FFT: the long-filter shape (1026..8193 taps whole -- halo 2048 .. 8192
// in rows of 1024 --, partitioned at 8192 taps beyond: LONG16).  From 1026 taps on a 4096-point tile is half halo (0.42 of HBM peak at 2049 taps).
//
// With 4096-point tiles a filter of more than 2049 taps is partitioned (hop 2048): 4095 taps cost one forward and
// one inverse 4096-point transform plus two spectrum products per 2048 outputs (the spectrum-cache variant of
// fir_fft.hip), 8191 taps four forward transforms and one inverse.  One 16384-point tile holds the whole filter:
// a halo of two to eight rows of 1024 samples, the fewest that hold the taps (four rows up to 4097 taps: 12288 new
// outputs per tile; eight up to 8193 taps: 8192 outputs), ONE spectrum, no partitions -- 0.99 vector instructions
// per output at 4095 taps where the cached 4096-point form executes 1.52, and a quarter of what the partitioned
// form executes at 8191.
//
// One workgroup of 1024 lanes (16 waves, four per SIMD) owns the CU: the tile fills the LDS (16 rows of 1088
// complex = 136 KB, + an 8 KB twiddle table).  Each lane keeps 16 complex points; radix 16 . 16 . 4 . 16:
//
//   n = 1024 a + r,  r = tid = 64 b + l,  l = 16 d + c          k = k0 + 16 k1 + 256 k2 + 1024 k3
//   A : lane r loads x[1024 a + r] (coalesced rows of 1024), DFT16 over a -> k0, times W_16384^(r k0)
//   exchange 1 (the only workgroup-wide one): row k0 of the LDS buffer goes to WAVE k0, which from here on owns
//       one independent 1024-point transform -- everything below is wave-local, no barrier
//   B : lane l: DFT16 over b -> k1, times W_1024^(l k1)                                  (LDS table)
//   exchange 2:  C : lane (q, c) holds (j, d), k1 = 4 q + j: four DFT4 over d -> k2, times W_64^(c k2)
//   exchange 3:  D : lane (k1, k2): DFT16 over c -> k3
//   => lane 64 k0 + 4 k1 + k2 holds X[k0 + 16 k1 + 256 k2 + 1024 k3], k3 = 0..15
// product with the filter spectrum (read from L2 per tile: requested during stage D, ahead of the next tile's
// prefetch, because memory operations return in order), mirrored inverse, stores of the rows past the halo.
// LDS layouts inside a row (complex elements): exchange 1 [r]; exchange 2 [68 k1 + 16 d + c]; exchange 3
// [68 k1 + 17 k2 + c]: each conflict-free for its ds_write_b64 (16-lane groups, 32 banks) and its ds_read_b64
// (32-lane groups, 64 banks), in both directions.
//
// One launch per call: the interior tiles (complex samples, or real samples two stretches at a time) stream through
// persistent workgroups, one per CU; the first tile of every channel (it reads the carried history), a partial last
// one and the history carry go to a few workgroups at the end of the same grid, on CUs of their own, and the last
// workgroup to finish ends the call (history swap).  (First build: those ran as fir_fft4096_short behind the main
// kernel, 16-26 us of a 300 us call.)
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <type_traits>

#include "fft_math.h"
#include "fir_kernels.h"
#include "fft_call.h"

#ifndef SDR16K_FENCE
#define SDR16K_FENCE 1     // scheduling fences behind each batch of LDS reads (1) or the compiler's own interleaving (0)
#endif
#if SDR16K_FENCE
#define SDR_READ_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define SDR_READ_FENCE() ((void)0)
#endif
#ifndef SDR16K_STREAM
#define SDR16K_STREAM 1    // each stage's outputs written as they are produced (1) or the staged form: compute, then sixteen writes (0)
#endif
#ifndef SDR16K_PRIO
#define SDR16K_PRIO 0      // issue priority by wave slot: 0 none, 1 by wave / 4, 2 by wave % 4
#endif
#ifndef SDR16K_TA_REGS
#define SDR16K_TA_REGS 1   // W_16384 twiddles in registers (1) or re-read from the L2-resident table at each use (0)
#endif
#ifndef SDR16K_PF_FIRST
#define SDR16K_PF_FIRST 8  // rows of the next tile requested right after the spectrum product; the rest before the inverse's block barrier
#endif                     // (4 / 12 / 16, and a third share behind the inverse's second stage: level or slower, profiles/r03_fft16k_one_launch.txt)
#ifndef SDR16K_PF_REAL
#define SDR16K_PF_REAL 7
#endif
#ifndef SDR16K_XCD
#define SDR16K_XCD 1       // a round of tiles dealt in contiguous runs per XCD, shared halo rows cacheable (1) or tile = blockIdx, all rows non-temporal (0)
#endif
#ifndef SDR16K_PREFETCH
#define SDR16K_PREFETCH 1  // next tile's loads issued during the inverse transform (1) or at the top of its own tile (0)
#endif

namespace sdr_amd {
namespace {

constexpr int NK = 16384;
constexpr int TK = 1024;   // lanes per workgroup
constexpr int PK = 1088;   // complex elements per k0 row: 1024 for exchange 1, 16 x 68 for the wave-local exchanges

__host__ __device__ constexpr int P16k(int k) { return 4 * (k & 3) + (k >> 2); }

// In-place 16-point DFT, as dft16 of fir_fft.hip: input v[n] natural, X[k] left at v[P16k(k)].
template <bool INV>
__device__ __forceinline__ void dft16k(f2 (&v)[16]) {
#pragma unroll
    for (int b = 0; b < 4; ++b) radix4<INV>(v[b], v[4 + b], v[8 + b], v[12 + b]);
    constexpr float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f;
    constexpr float C2 = 0.70710678118654752440f;
    constexpr float sg = INV ? 1.0f : -1.0f;
    v[5] = cmul_k(v[5], C1, sg * S1);
    v[6] = cmul_k(v[6], C2, sg * C2);
    v[7] = cmul_k(v[7], S1, sg * C1);
    v[9] = cmul_k(v[9], C2, sg * C2);
    {
        const f2 z = {0.0f, 0.0f};
        v[10] = INV ? add_pi(z, v[10]) : add_mi(z, v[10]);
    }
    v[11] = cmul_k(v[11], -C2, sg * C2);
    v[13] = cmul_k(v[13], S1, sg * C1);
    v[14] = cmul_k(v[14], -C2, sg * C2);
    v[15] = cmul_k(v[15], -C1, -sg * S1);
#pragma unroll
    for (int c = 0; c < 4; ++c) radix4<INV>(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
}

// The same transform in two pieces, for stages that hand each output on as soon as it exists: first_pass, then
// last_pass_group(c) for c = 0..3, which leaves X[c + 4 j] at v[4 c + j].
template <bool INV>
__device__ __forceinline__ void first_pass(f2 (&v)[16]) {
#pragma unroll
    for (int b = 0; b < 4; ++b) radix4<INV>(v[b], v[4 + b], v[8 + b], v[12 + b]);
}
template <bool INV>
__device__ __forceinline__ void last_pass_group(f2 (&v)[16], int c) {
    constexpr float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f;
    constexpr float C2 = 0.70710678118654752440f;
    constexpr float sg = INV ? 1.0f : -1.0f;
    if (c == 1) {
        v[5] = cmul_k(v[5], C1, sg * S1);
        v[6] = cmul_k(v[6], C2, sg * C2);
        v[7] = cmul_k(v[7], S1, sg * C1);
    } else if (c == 2) {
        const f2 z = {0.0f, 0.0f};
        v[9] = cmul_k(v[9], C2, sg * C2);
        v[10] = INV ? add_pi(z, v[10]) : add_mi(z, v[10]);
        v[11] = cmul_k(v[11], -C2, sg * C2);
    } else if (c == 3) {
        v[13] = cmul_k(v[13], S1, sg * C1);
        v[14] = cmul_k(v[14], -C2, sg * C2);
        v[15] = cmul_k(v[15], -C1, -sg * S1);
    }
    radix4<INV>(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Per-lane constants: TA[k] = W_16384^(tid k) (wtab[1024 k + tid]), TC[k] = W_64^(c k), c = tid & 15 (wtab[16384 + 1024
// + 16 k + c]); the W_1024^(l k) twiddles come from an LDS table (tabB[64 k + l], filled from wtab[16384 + ...]).
struct Lane16k {
    f2 TA[SDR16K_TA_REGS ? 16 : 1], TC[4];
    const f2* tb; // tabB + l
    const f2* ta_mem;
    unsigned ta_lane;
    __device__ __forceinline__ f2 ta(int k) const { return SDR16K_TA_REGS ? TA[SDR16K_TA_REGS ? k : 0] : (ta_mem + 1024 * k)[ta_lane]; }
    __device__ __forceinline__ void init(int tid, const f2* wtab, f2* tabB) {
        ta_mem = wtab;
        ta_lane = (unsigned)tid;
#pragma unroll
        for (int k = 1; k < (SDR16K_TA_REGS ? 16 : 1); ++k) TA[k] = (wtab + 1024 * k)[(unsigned)tid];
#pragma unroll
        for (int k = 1; k < 4; ++k) TC[k] = wtab[16384 + 1024 + 16 * k + (tid & 15)];
        tabB[tid] = wtab[16384 + tid];
        tb = tabB + (tid & 63);
    }
};

template <class Hook, class HookD>
__device__ __forceinline__ void forward16k(f2 (&v)[16], f2* lds, int tid, const Lane16k& c, Hook&& after_points_left, HookD&& before_stage_d) {
    const int l = tid & 63;
    f2* row = lds + (tid >> 6) * PK;
    // ---- A ----------------------------------------------------------------------------
    dft16k<false>(v);
#pragma unroll
    for (int k = 1; k < 16; ++k) v[P16k(k)] = cmul(v[P16k(k)], c.ta(k));
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[k * PK + tid] = v[P16k(k)];
    after_points_left();
    __syncthreads();
    // ---- B: lane l, over b -> k1, twiddle W_1024^(l k1) ---------------------------------
#pragma unroll
    for (int b = 0; b < 16; ++b) v[b] = row[64 * b + l];
    SDR_READ_FENCE();
    dft16k<false>(v);
    // (table twiddles four at a time: requested as one batch each, but never all fifteen at once -- the kernel has
    // 128 VGPRs for 16 points + 16 prefetched points + 15 per-lane twiddles, and a spilled value comes back
    // through the vector-memory queue BEHIND the prefetch, at HBM latency)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        f2 t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = c.tb[64 * (4 * g4 + i)];
        SDR_READ_FENCE();
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (4 * g4 + i > 0) v[P16k(4 * g4 + i)] = cmul(v[P16k(4 * g4 + i)], t[i]);
        SDR_READ_FENCE();
    }
    wave_sync();
#pragma unroll
    for (int k = 0; k < 16; ++k) row[68 * k + l] = v[P16k(k)];
    wave_sync();
    // ---- C: lane (q, c) = (l >> 4, l & 15) holds (j, d): k1 = 4 q + j; four DFT4 over d -> k2, twiddle W_64^(c k2)
    const int q = l >> 4, cc = l & 15;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int d = 0; d < 4; ++d) v[4 * j + d] = row[68 * (4 * q + j) + 16 * d + cc];
    SDR_READ_FENCE();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        radix4<false>(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
#pragma unroll
        for (int k = 1; k < 4; ++k) v[4 * j + k] = cmul(v[4 * j + k], c.TC[k]);
    }
    wave_sync();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) row[68 * (4 * q + j) + 17 * k + cc] = v[4 * j + k];
    before_stage_d();
    wave_sync();
    // ---- D: lane (k1, k2) = (l >> 2, l & 3), over c -> k3 --------------------------------
    const int k1 = l >> 2, k2 = l & 3;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = row[68 * k1 + 17 * k2 + i];
    SDR_READ_FENCE();
    dft16k<false>(v);
}

template <class Hook>
__device__ __forceinline__ void inverse16k(f2 (&w)[16], f2* lds, int tid, const Lane16k& c, Hook&& before_barrier) {
    const int l = tid & 63;
    f2* row = lds + (tid >> 6) * PK;
    const int k1 = l >> 2, k2 = l & 3, q = l >> 4, cc = l & 15;
    // ---- D' ---------------------------------------------------------------------------
    dft16k<true>(w);
    wave_sync();
#pragma unroll
    for (int i = 0; i < 16; ++i) row[68 * k1 + 17 * k2 + i] = w[P16k(i)];
    wave_sync();
    // ---- C' ---------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) w[4 * j + k] = row[68 * (4 * q + j) + 17 * k + cc];
    SDR_READ_FENCE();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int k = 1; k < 4; ++k) w[4 * j + k] = cmulc(w[4 * j + k], c.TC[k]);
        radix4<true>(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
    }
    wave_sync();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int d = 0; d < 4; ++d) row[68 * (4 * q + j) + 16 * d + cc] = w[4 * j + d];
    wave_sync();
    // ---- B' ---------------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < 16; ++k) w[k] = row[68 * k + l];
    SDR_READ_FENCE();
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        f2 t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = c.tb[64 * (4 * g4 + i)];
        SDR_READ_FENCE();
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (4 * g4 + i > 0) w[4 * g4 + i] = cmulc(w[4 * g4 + i], t[i]);
        SDR_READ_FENCE();
    }
    dft16k<true>(w);
    wave_sync();
#pragma unroll
    for (int b = 0; b < 16; ++b) row[64 * b + l] = w[P16k(b)];
    before_barrier();
    __syncthreads();
    // ---- A' ---------------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < 16; ++k) w[k] = lds[k * PK + tid];
    SDR_READ_FENCE();
#pragma unroll
    for (int k = 1; k < 16; ++k) w[k] = cmulc(w[k], c.ta(k));
    dft16k<true>(w);
}

// The streaming forms.  The LDS serves one wave's instructions in order, and all lanes of a wave issue together:
// a stage's writes may follow its reads as soon as the read INSTRUCTIONS are issued (wave_sync there keeps the
// compiler from moving a write above a read of another lane's element), so each output goes to the LDS the moment
// its butterfly is done and the queue of sixteen writes overlaps the rest of the stage's arithmetic, in this wave
// and in the three that share its SIMD.
template <class HookD>
__device__ __forceinline__ void forward16k_s(f2 (&v)[16], f2* lds, int tid, const Lane16k& c, HookD&& before_stage_d) {
    const int l = tid & 63;
    f2* row = lds + (tid >> 6) * PK;
    // ---- A ----
    first_pass<false>(v);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        last_pass_group<false>(v, g);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = g + 4 * j;
            lds[k * PK + tid] = k ? cmul(v[4 * g + j], c.ta(k)) : v[4 * g + j];
        }
    }
    __syncthreads();
    // ---- B ----
#pragma unroll
    for (int b0 = 0; b0 < 4; ++b0)
#pragma unroll
        for (int a = 0; a < 4; ++a) v[b0 + 4 * a] = row[64 * (b0 + 4 * a) + l];
    wave_sync();
    first_pass<false>(v);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f2 t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = c.tb[64 * (g + 4 * j)];
        last_pass_group<false>(v, g);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = g + 4 * j;
            row[68 * k + l] = k ? cmul(v[4 * g + j], t[j]) : v[4 * g + j];
        }
    }
    wave_sync();
    // ---- C ----
    const int q = l >> 4, cc = l & 15;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int d = 0; d < 4; ++d) v[4 * j + d] = row[68 * (4 * q + j) + 16 * d + cc];
    wave_sync();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        radix4<false>(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
#pragma unroll
        for (int k = 0; k < 4; ++k) row[68 * (4 * q + j) + 17 * k + cc] = k ? cmul(v[4 * j + k], c.TC[k]) : v[4 * j + k];
    }
    before_stage_d();
    wave_sync();
    // ---- D ----  (D' writes every element back to the address this lane read it from: no ordering needed there)
    const int k1 = l >> 2, k2 = l & 3;
#pragma unroll
    for (int b0 = 0; b0 < 4; ++b0)
#pragma unroll
        for (int a = 0; a < 4; ++a) v[b0 + 4 * a] = row[68 * k1 + 17 * k2 + b0 + 4 * a];
    dft16k<false>(v);
}

// emit(r, value): output row r of this lane's column (x[1024 r + tid]), in the order 0 4 8 12 1 5 ...
template <class Hook, class Emit>
__device__ __forceinline__ void inverse16k_s(f2 (&w)[16], f2* lds, int tid, const Lane16k& c, Hook&& before_barrier, Emit&& emit) {
    const int l = tid & 63;
    f2* row = lds + (tid >> 6) * PK;
    const int k1 = l >> 2, k2 = l & 3, q = l >> 4, cc = l & 15;
    // ---- D' ----
    first_pass<true>(w);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        last_pass_group<true>(w, g);
#pragma unroll
        for (int j = 0; j < 4; ++j) row[68 * k1 + 17 * k2 + g + 4 * j] = w[4 * g + j];
    }
    wave_sync();
    // ---- C' ----
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) w[4 * j + k] = row[68 * (4 * q + j) + 17 * k + cc];
    wave_sync();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int k = 1; k < 4; ++k) w[4 * j + k] = cmulc(w[4 * j + k], c.TC[k]);
        radix4<true>(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
#pragma unroll
        for (int d = 0; d < 4; ++d) row[68 * (4 * q + j) + 16 * d + cc] = w[4 * j + d];
    }
    wave_sync();
    // ---- B' ----
#pragma unroll
    for (int b0 = 0; b0 < 4; ++b0) {
        f2 t[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            w[b0 + 4 * a] = row[68 * (b0 + 4 * a) + l];
            t[a] = c.tb[64 * (b0 + 4 * a)];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
            if (b0 + 4 * a > 0) w[b0 + 4 * a] = cmulc(w[b0 + 4 * a], t[a]);
        radix4<true>(w[b0], w[4 + b0], w[8 + b0], w[12 + b0]);
    }
    wave_sync();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        last_pass_group<true>(w, g);
#pragma unroll
        for (int j = 0; j < 4; ++j) row[64 * (g + 4 * j) + l] = w[4 * g + j];
    }
    before_barrier();
    __syncthreads();
    // ---- A' ----
#pragma unroll
    for (int b0 = 0; b0 < 4; ++b0) {
#pragma unroll
        for (int a = 0; a < 4; ++a) w[b0 + 4 * a] = lds[(b0 + 4 * a) * PK + tid];
#pragma unroll
        for (int a = 0; a < 4; ++a)
            if (b0 + 4 * a > 0) w[b0 + 4 * a] = cmulc(w[b0 + 4 * a], c.ta(b0 + 4 * a));
        radix4<true>(w[b0], w[4 + b0], w[8 + b0], w[12 + b0]);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        last_pass_group<true>(w, g);
#pragma unroll
        for (int j = 0; j < 4; ++j) emit(g + 4 * j, w[4 * g + j]);
    }
}

// Edge tiles: the first tile of every channel (its window reaches back into the carried history) and the last one when
// it is partial (outputs or inputs beyond n).  The same transform pair over the virtual stream history | input | 0,
// bounds-checked.  A few workgroups at the END of the grid start on them at once, beside the bulk: a single long stream hides
// its two edge tiles entirely that way, sharing them out by stride.  COUNTER (calls of many channels: two edge tiles each): the
// tiles are drawn from a counter (the handle's first ticket counter: zero between calls) and every interior workgroup that runs
// out of tiles joins in, so that the many edge tiles spread over the chip.  (A kernel of its own -- fir_fft16k<.., HELP> --
// because the helping epilogue costs the single-stream kernel's tile loop 6 %: profiles/r03_fft16k_one_launch.txt.)
template <int HROWS, bool REAL, bool LONG16, bool COUNTER>
__device__ __forceinline__ void edge16k(const FftArgs& a, const float* hist_base, f2* lds, const Lane16k& c, const int tid,
                                        const unsigned bid, const unsigned nblk) {
    constexpr long long HALO = 1024LL * HROWS, L = NK - HALO, PT = REAL ? 2 * L : L;
    constexpr int W = REAL ? 1 : 2;
    const long long tiles_per_ch = (a.n + PT - 1) / PT;
    const long long edge_per_ch = tiles_per_ch - a.big_tiles; // the tiles before the first interior one and whatever follows the last
    const long long total = edge_per_ch * a.channels;
    const long long t_first = a.big_first / PT;                // first interior tile
    const f2* const hbase = reinterpret_cast<const f2*>(a.hperm16k);
    __shared__ unsigned next_edge;
    long long e = bid;
    for (;;) {
        if constexpr (COUNTER) {
            __syncthreads(); // everyone has read the previous draw (and is done with the exchange buffer)
            if (tid == 0) next_edge = atomicAdd(a.tickets, 1u);
            __syncthreads();
            e = next_edge;
        }
        if (e >= total) break;
        const long long ch = e / edge_per_ch;
        long long tt = e - ch * edge_per_ch;
        if (tt >= t_first) tt += a.big_tiles;
        const float* in = a.in + ch * a.in_stride * W;
        const float* hist = hist_base + ch * (long long)a.hist_len * W;
        float* out = a.out + ch * a.out_stride * W;
        const long long o0 = tt * PT; // first output of the tile
        auto sample = [&](long long m, int k) -> float { // virtual stream: history | input | 0
            if (m < 0) { const long long h = a.hist_len + m; return h >= 0 ? hist[h * W + k] : 0.0f; }
            if (m >= a.n) return 0.0f;
            return in[m * W + k];
        };
        // 1024 consecutive samples from m0 (wave-uniform): a row lies whole in the input or whole in the history except
        // where it straddles an end of the stream -- uniform branches to plain coalesced loads, per-lane checks only there
        auto stretch = [&](long long m0, int k) -> float {
            if (m0 >= 0 && m0 + 1024 <= a.n) return in[(m0 + tid) * W + k];
            if (m0 + 1024 <= 0 && a.hist_len + m0 >= 0) return hist[(a.hist_len + m0 + tid) * W + k];
            return sample(m0 + tid, k);
        };
        f2 v[16];
        auto window = [&](long long back) { // the tile's window, `back` samples earlier
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m0 = o0 - HALO - back + 1024 * r;
                if (REAL) v[r] = f2{stretch(m0, 0), stretch(m0 + L, 0)};
                else v[r] = f2{stretch(m0, 0), stretch(m0, 1)};
                asm volatile("" : "+v"(v[r])); // one row at a time: sixteen rows of guarded loads at once spilled a hundred registers
            }
        };
        window(0);
        f2 h[16];
        forward16k_s(v, lds, tid, c, [&] {
#pragma unroll
            for (int k = 0; k < 16; ++k) h[k] = (hbase + 1024 * k)[(unsigned)tid];
        });
        f2 w[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) w[k] = cmul(v[P16k(k)], h[k]);
        if constexpr (LONG16) {
            for (int p = 1; p < a.parts16; ++p) { // Y_b = sum_p X_(b-p) H_p, hop 8192
                window(8192LL * p);
                __syncthreads(); // the previous transform was a forward one too: its last stage still reads rows this one's first stage writes
                forward16k_s(v, lds, tid, c, [] {});
                const f2* const hp = hbase + (size_t)p * NK;
#pragma unroll
                for (int k = 0; k < 16; ++k) w[k] += cmul(v[P16k(k)], (hp + 1024 * k)[(unsigned)tid]);
            }
        }
        inverse16k_s(w, lds, tid, c, [] {}, [&](int r, f2 val) {
            if (r < HROWS) return; // halo positions
            const long long o = o0 - HALO + 1024 * r + tid;
            if (REAL) {
                if (o < a.n) out[o] = val.x;
                if (o + L < a.n) out[o + L] = val.y;
            } else if (o < a.n) {
                *reinterpret_cast<f2*>(out + 2 * o) = val;
            }
        });
        if constexpr (!COUNTER) e += nblk;
    }
}

// HROWS = halo / 1024 (2 .. 8: the smallest number of rows that holds the taps).  One launch per call.  Tiles of a channel: tile t produces outputs [t P, t P + P),
// P = L = 16384 - halo (REAL: 2 L), reading 16384 inputs from t P - halo.  Interior tiles -- every input and output
// inside [0, n): tiles 1 .. a.big_tiles, tile t' = t - 1 starting at a.big_first + t' P with a.big_first = P -- are
// walked by the first G workgroups of the grid, persistently; behind them come a.grid_edge workgroups for the edge
// tiles and a.grid_carry for the history carry (G + edge + carry <= the CUs: one workgroup fills a CU, and every one
// must be resident for the short work to overlap the bulk).  The last workgroup to finish ends the call (call_done).
// REAL: two stretches of a real stream ride as re / im of one complex transform (the filter is real-tap, hence
// linear over that pairing), as in fir_fft4096: the re member produces the first L outputs of the tile, the im member
// the next L (its window starts L samples later).
// LONG16 (HROWS = 8): uniformly partitioned overlap-save for more than 8193 taps -- hop 8192, the filter cut into a.parts16
// pieces of 8192 taps, Y_b = sum_p X_(b-p) H_p: the windows b-1.. are re-read (L2 / Infinity Cache) and re-transformed, the
// spectra H_p come from L2 (a.hperm16k holds them back to back).  (parts + 1) transforms of 16384 points per 8192 outputs where
// the 4096-point partitioned form runs (4 parts + 1) of 4096 points per 2048: 0.35-0.4 of its arithmetic.  No register prefetch
// of the next tile in this variant: the product accumulators take its place.
template <int HROWS, bool REAL, bool LONG16 = false, bool HELP = false>
__global__ __launch_bounds__(1024, 4) void fir_fft16k(FftArgs a) {
    static_assert(!LONG16 || HROWS == 8, "the partitioned form hops 8192 samples");
    __shared__ __align__(16) f2 lds[16 * PK + 1024];
    const int tid = threadIdx.x;
#if SDR16K_PRIO == 1
    switch (tid >> 8) { case 0: __builtin_amdgcn_s_setprio(3); break; case 1: __builtin_amdgcn_s_setprio(2); break; case 2: __builtin_amdgcn_s_setprio(1); break; default: break; }
#elif SDR16K_PRIO == 2
    switch ((tid >> 6) & 3) { case 0: __builtin_amdgcn_s_setprio(3); break; case 1: __builtin_amdgcn_s_setprio(2); break; case 2: __builtin_amdgcn_s_setprio(1); break; default: break; }
#endif
    constexpr long long HALO = 1024LL * HROWS, L = NK - HALO;
    const unsigned G = gridDim.x - (unsigned)a.grid_edge - (unsigned)a.grid_carry; // interior workgroups
    Lane16k c;
    if (blockIdx.x >= G) {
        const unsigned sb = blockIdx.x - G;
        const unsigned cur = a.state->cur & 1u;
        if (sb >= (unsigned)a.grid_edge) {
            carry_history<false, TK>(a, pick(a.hist, cur), pick(a.hist, cur ^ 1u), REAL ? 1 : 2, (int)(sb - (unsigned)a.grid_edge), a.grid_carry);
            call_done();
            return;
        }
        c.init(tid, reinterpret_cast<const f2*>(a.wtab16k), lds + 16 * PK);
        edge16k<HROWS, REAL, LONG16, HELP>(a, pick(a.hist, cur), lds, c, tid, sb, (unsigned)a.grid_edge);
        call_done();
        return;
    }
    // (channel, tile) of this workgroup's current and next tile, advanced without a division: the flat list is walked
    // grid-strided, and stride = dq big_tiles + dr is split once
    const int tiles = (int)a.big_tiles, channels = a.channels;
    // Neighbouring tiles share a halo of 2048 .. 8192 samples (a seventh to all of what a tile reads).  Workgroups go
    // to the eight XCDs round-robin by blockIdx, each XCD with an L2 of its own: with tile = blockIdx every halo was
    // fetched through two L2s -- 1.36x the algorithmic reads at 4095 taps, 1.07x now (profiles/r03_c3_summary.txt).
    // A round of the grid is dealt in eight contiguous runs instead, one per XCD (XCD x holds the workgroups
    // blockIdx = x mod 8: G / 8 of them, one more for x < G mod 8), and the rows two tiles share are loaded with the
    // default cache policy (the rest non-temporal): the neighbour finds them in its own XCD's L2.
    unsigned first = blockIdx.x;
#if SDR16K_XCD
    {
        const unsigned x = blockIdx.x & 7u, q = G >> 3, rem = G & 7u;
        first = x * q + (x < rem ? x : rem) + (blockIdx.x >> 3);
    }
#endif
    int ch = (int)(first / (unsigned)tiles), t = (int)(first % (unsigned)tiles);
    int dq = (int)(G / (unsigned)tiles), dr = (int)(G % (unsigned)tiles);
    asm volatile("" : "+s"(dq), "+s"(dr));
    const int ch0 = ch, t0 = t;
    typedef typename std::conditional<REAL, float, f2>::type S; // one sample in memory
    constexpr long long PER_TILE = REAL ? 2 * L : L;
    auto in_of = [&](int chi, int ti) -> const S* {
        return reinterpret_cast<const S*>(a.in) + chi * a.in_stride + a.big_first + ti * PER_TILE - HALO;
    };
    unsigned lane = (unsigned)threadIdx.x;
    // row r of a window: one complex sample per lane, or the pair (x[1024 r + lane], x[L + 1024 r + lane])
    auto load_row = [&](const S* p, int r) -> f2 {
        const bool shared = SDR16K_XCD && (r < HROWS || r >= 16 - HROWS); // also a row of the previous / next tile
        if constexpr (REAL) {
            const S* q0 = &(p + 1024 * r)[lane];
            const S* q1 = &(p + (L + 1024 * r))[lane];
            return shared ? f2{*q0, *q1} : f2{stream_load(q0), stream_load(q1)};
        } else {
            const S* q = &(p + 1024 * r)[lane];
            return shared ? *q : stream_load(q);
        }
    };
    auto store_row = [&](S* q, int r, f2 val) {
        if constexpr (REAL) {
            stream_store(val.x, &(q + 1024 * r)[lane]);
            stream_store(val.y, &(q + (L + 1024 * r))[lane]);
        } else {
            stream_store(val, &(q + 1024 * r)[lane]);
        }
    };
    constexpr bool PREFETCH = SDR16K_PREFETCH && !LONG16;
    // (real samples: seven rows early -- with eight the allocator spills three registers inside the loop, and their reloads queue
    // behind the prefetch at HBM latency; the count is a property of this compiler's allocation, re-check it with `make asm`)
    constexpr int PF_FIRST = REAL ? SDR16K_PF_REAL : SDR16K_PF_FIRST;
    f2 nx[16];
    if constexpr (PREFETCH) {
        const S* p = in_of(ch, t);
#pragma unroll
        for (int r = 0; r < 16; ++r) nx[r] = load_row(p, r);
    }
    c.init(tid, reinterpret_cast<const f2*>(a.wtab16k), lds + 16 * PK);
    // every global access below is a wave-uniform base + this lane's 32-bit offset: one VGPR of addressing for all
    // sixteen rows (per-lane 64-bit pointers cost two registers per row, and the first build spilled exactly those)
    const f2* const hbase = reinterpret_cast<const f2*>(a.hperm16k);
    for (;;) {
        // (opaque per tile: the spectrum's sixteen addresses are loop-invariant, and hoisted out of the loop they
        // were sixteen 64-bit registers that went straight to scratch)
        asm volatile("" : "+v"(lane));
        f2 v[16];
        if constexpr (!PREFETCH) {
            const S* p = in_of(ch, t);
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = load_row(p, r);
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = nx[i];
        }
        int chn = ch + dq, tn = t + dr;
        if (tn >= tiles) {
            tn -= tiles;
            ++chn;
        }
        const bool more = chn < channels;
        const S* pn = more ? in_of(chn, tn) : in_of(ch0, t0); // past the end: a tile that sits in L2 (nobody consumes it)
        // the spectrum: L2-resident, requested ahead of the forward transform's last stage and AHEAD of the next tile's
        // prefetch (memory operations return in order)
        f2 h[16];
#if SDR16K_STREAM
        forward16k_s(v, lds, tid, c, [&] {
#pragma unroll
            for (int k = 0; k < 16; ++k) h[k] = (hbase + 1024 * k)[lane];
        });
#else
        forward16k(v, lds, tid, c, [] {}, [&] {
#pragma unroll
            for (int k = 0; k < 16; ++k) h[k] = (hbase + 1024 * k)[lane];
        });
#endif
        f2 w[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) w[k] = cmul(v[P16k(k)], h[k]);
        if constexpr (LONG16) {
            for (int p = 1; p < a.parts16; ++p) { // window b-p: every row was read before (by this or a neighbouring tile)
                const S* pw = in_of(ch, t) - 8192LL * p;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if constexpr (REAL) v[r] = f2{(pw + 1024 * r)[lane], (pw + (L + 1024 * r))[lane]};
                    else v[r] = (pw + 1024 * r)[lane];
                }
                // the previous transform was a forward one too: other waves' last stage may still read the rows this one's
                // first stage writes (after an inverse transform its block barrier has seen to that)
                __syncthreads();
                forward16k_s(v, lds, tid, c, [] {});
                const f2* const hp = hbase + (size_t)p * NK;
                // the spectrum a quarter at a time: the accumulators leave no room for sixteen rows of it
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    f2 hq[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) hq[i] = (hp + 1024 * (4 * g4 + i))[lane];
#pragma unroll
                    for (int i = 0; i < 4; ++i) w[4 * g4 + i] += cmul(v[P16k(4 * g4 + i)], hq[i]);
                }
            }
        }
        if constexpr (PREFETCH) {
#pragma unroll
            for (int r = 0; r < PF_FIRST; ++r) nx[r] = load_row(pn, r);
        }
        S* q0 = reinterpret_cast<S*>(a.out) + ch * a.out_stride + a.big_first + t * PER_TILE - HALO;
#if SDR16K_STREAM
        inverse16k_s(w, lds, tid, c, [&] {
            if constexpr (PREFETCH) {
#pragma unroll
                for (int r = PF_FIRST; r < 16; ++r) nx[r] = load_row(pn, r);
            }
        }, [&](int r, f2 val) {
            if (r >= HROWS) store_row(q0, r, val);
        });
#else
        inverse16k(w, lds, tid, c, [&] {
            if constexpr (PREFETCH) {
#pragma unroll
                for (int r = PF_FIRST; r < 16; ++r) nx[r] = load_row(pn, r);
            }
        });
#pragma unroll
        for (int r = HROWS; r < 16; ++r) store_row(q0, r, w[P16k(r)]);
#endif
        ch = chn;
        t = tn;
        if (!more) break;
    }
    // Help with what is left of the edge tiles.  The arguments are read again from the kernel-argument segment through a pointer
    // the compiler cannot see through: anything this epilogue kept alive across the tile loop cost that loop registers (22 scratch
    // operations per tile when it used the kernel's own copy of the arguments).
    if constexpr (HELP) {
        const char __attribute__((address_space(4)))* ka =
            (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        FftArgs a2;
        __builtin_memcpy(&a2, (const void*)ka, sizeof(FftArgs));
        const unsigned cur2 = a2.state->cur & 1u;
        edge16k<HROWS, REAL, LONG16, true>(a2, pick(a2.hist, cur2), lds, c, tid, 0u, 1u);
    }
    call_done();
}

} // namespace

// For a call of n samples per channel: outputs per tile, the first interior tile and how many interior tiles follow it (see
// fir_fft16k; parts16 > 1: the partitioned form, whose interior windows reach (parts16 - 1) * 8192 samples further back)
void fir_fft16k_tiles(long long n, int halo16k, bool real, int parts16, long long* per_tile, long long* first, long long* interior) {
    const long long P = (long long)(NK - halo16k) * (real ? 2 : 1);
    *per_tile = P;
    const long long reach = halo16k + (parts16 > 1 ? (long long)(parts16 - 1) * 8192 : 0);
    const long long t_first = (reach + P - 1) / P; // first tile whose oldest window starts at a sample >= 0
    const long long full = n / P;                  // tiles whose outputs and inputs all lie below n: 0 .. full - 1
    *first = t_first;
    *interior = full > t_first ? full - t_first : 0;
}

// The whole call in one launch: interior, edge and carry workgroups (see fir_fft16k); `cus` = CUs of the device.
hipError_t launch_fir_fft16k(const FftArgs& a0, bool real, int cus, hipStream_t stream) {
    if (a0.n <= 0 || a0.channels <= 0) return hipSuccess;
    if (!a0.hperm16k || !a0.wtab16k || !a0.state || a0.halo16k < 2048 || a0.halo16k > 8192 || a0.halo16k % 1024 != 0) return hipErrorInvalidValue;
    if (a0.parts16 > 1 && a0.halo16k != 8192) return hipErrorInvalidValue;
    FftArgs a = a0;
    long long P, t_first, inner;
    fir_fft16k_tiles(a.n, a.halo16k, real, a.parts16, &P, &t_first, &inner);
    if (inner > 0x7fffffffLL / (a.channels > 0 ? a.channels : 1)) return hipErrorInvalidValue; // 32-bit flat tile indices
    const long long tiles_per_ch = (a.n + P - 1) / P;
    a.big_tiles = inner;
    a.big_first = t_first * P;
    const long long total = inner * a.channels, edge = (tiles_per_ch - inner) * a.channels;
    // the short work runs beside the bulk on CUs of its own: as few workgroups as finish it within the bulk's time
    // (an edge tile costs about two interior ones), at most a quarter of the chip for a call that is all edges
    // (an edge tile costs about three interior ones; with more than a handful of edge tiles -- many channels -- what the
    // dedicated workgroups have not taken when the bulk is through, the interior workgroups finish: fir_fft16k<.., HELP>)
    const long long bulk_rounds = total > 0 ? (total + cus - 1) / cus : 1;
    long long ge = (3 * edge + bulk_rounds - 1) / bulk_rounds;
    if (ge > edge) ge = edge;
    if (ge > cus / 8) ge = cus / 8;
    if (ge < 1) ge = 1;
    const long long hist_elems = (long long)a.hist_len * (real ? 1 : 2) * a.channels;
    long long gc = (hist_elems + TK * 64 - 1) / (TK * 64);
    if (gc > 4) gc = 4;
    if (gc < 1) gc = 1;
    long long G = cus - ge - gc;
    if (G > total) G = total;
    if (G < 0) G = 0;
    a.grid_edge = (int)ge;
    a.grid_carry = (int)gc;
    const dim3 g((unsigned)(G + ge + gc)), b(TK);
    // the helping variant when the edge tiles are more than the dedicated workgroups take in their stride within the bulk's time
    const bool help = a.tickets != nullptr && edge > 8;
    auto go = [&](auto hrows, auto re, auto lng) {
        constexpr int H = decltype(hrows)::value;
        constexpr bool R = decltype(re)::value, LG = decltype(lng)::value;
        if (help) hipLaunchKernelGGL((fir_fft16k<H, R, LG, true>), g, b, 0, stream, a);
        else hipLaunchKernelGGL((fir_fft16k<H, R, LG, false>), g, b, 0, stream, a);
    };
    auto whole = [&](auto hrows) {
        if (real) go(hrows, IC<1>{}, IC<0>{});
        else go(hrows, IC<0>{}, IC<0>{});
    };
    if (a.parts16 > 1) {
        if (real) go(IC<8>{}, IC<1>{}, IC<1>{});
        else go(IC<8>{}, IC<0>{}, IC<1>{});
    } else {
        switch (a.halo16k / 1024) { // the halo in whole rows of 1024 samples: the smallest that holds the taps
            case 2: whole(IC<2>{}); break;
            case 3: whole(IC<3>{}); break;
            case 4: whole(IC<4>{}); break;
            case 5: whole(IC<5>{}); break;
            case 6: whole(IC<6>{}); break;
            case 7: whole(IC<7>{}); break;
            default: whole(IC<8>{}); break;
        }
    }
    return hipGetLastError();
}

// Host tables.  Spectrum H[k] / N in natural order (16384 complex) -> this kernel's lane order:
// lane 64 k0 + 4 k1 + k2 holds bin k0 + 16 k1 + 256 k2 + 1024 k3 at hperm[1024 k3 + lane].
void fir_fft16k_permute_spectrum(const float* h_natural, float* hperm) {
    for (int k = 0; k < NK; ++k) {
        const int k0 = k & 15, k1 = (k >> 4) & 15, k2 = (k >> 8) & 3, k3 = k >> 10;
        const int pos = 1024 * k3 + 64 * k0 + 4 * k1 + k2;
        hperm[2 * pos] = h_natural[2 * k];
        hperm[2 * pos + 1] = h_natural[2 * k + 1];
    }
}

// cs / sn = cos / sin(2 pi m / 16384), m = 0..16383; forward sign exp(-i ...)
void fir_fft16k_fill_twiddles(const double* cs, const double* sn, float* wtab) {
    auto put = [&](int pos, long long m) {
        m &= NK - 1;
        wtab[2 * pos] = (float)cs[m];
        wtab[2 * pos + 1] = (float)-sn[m];
    };
    for (int k = 0; k < 16; ++k)
        for (int t = 0; t < 1024; ++t) put(1024 * k + t, (long long)t * k);          // W_16384^(t k)
    for (int k = 0; k < 16; ++k)
        for (int l = 0; l < 64; ++l) put(16384 + 64 * k + l, 16LL * l * k);           // W_1024^(l k)
    for (int k = 0; k < 4; ++k)
        for (int c = 0; c < 16; ++c) put(16384 + 1024 + 16 * k + c, 256LL * c * k);   // W_64^(c k)
}

} // namespace sdr_amd
