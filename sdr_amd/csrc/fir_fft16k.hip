// fir_fft16k.hip -- overlap-save FIR with an in-LDS 16384-point FFT: the long-filter shape (2050..8193 taps).
//
// With 4096-point tiles a filter of more than 2049 taps is partitioned (hop 2048): 4095 taps cost one forward and
// one inverse 4096-point transform plus two spectrum products per 2048 outputs (the spectrum-cache variant of
// fir_fft.hip), 8191 taps four forward transforms and one inverse.  One 16384-point tile holds the whole filter:
// halo 4096 (up to 4097 taps: 12288 new outputs per tile) or 8192 (up to 8193 taps: 8192 outputs), ONE spectrum, no
// partitions -- 0.99 vector instructions per output at 4095 taps where the cached 4096-point form executes 1.52,
// and a quarter of what the partitioned form executes at 8191.
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
// Interior tiles only (complex samples, or real samples two stretches at a time); the call's carry / edge work runs as fir_fft4096_short behind it on the
// stream -- the 4096-point partitioned form over the few 2048-sample tiles at both ends of every channel -- and
// ends the call (history swap).  Tile boundaries are multiples of 2048 for that reason.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <type_traits>

#include "fft_math.h"
#include "fir_kernels.h"

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

// HROWS = halo / 1024 (4 or 8).  Tiles of a channel: big tile t (0 <= t < a.big_tiles) produces outputs
// [a.big_first + t L, + L), L = 16384 - halo, reading 16384 inputs from a.big_first + t L - halo.  Workgroups walk
// the flat list (channel, tile) grid-strided.
// REAL: two stretches of a real stream ride as re / im of one complex transform (the filter is real-tap, hence
// linear over that pairing), as in fir_fft4096: big tile t produces 2 L outputs from a.big_first + 2 t L, the re
// member the first L of them, the im member the next L (its window starts L samples later).
template <int HROWS, bool REAL>
__global__ __launch_bounds__(1024, 4) void fir_fft16k(FftArgs a) {
    __shared__ __align__(16) f2 lds[16 * PK + 1024];
    const int tid = threadIdx.x;
#if SDR16K_PRIO == 1
    switch (tid >> 8) { case 0: __builtin_amdgcn_s_setprio(3); break; case 1: __builtin_amdgcn_s_setprio(2); break; case 2: __builtin_amdgcn_s_setprio(1); break; default: break; }
#elif SDR16K_PRIO == 2
    switch ((tid >> 6) & 3) { case 0: __builtin_amdgcn_s_setprio(3); break; case 1: __builtin_amdgcn_s_setprio(2); break; case 2: __builtin_amdgcn_s_setprio(1); break; default: break; }
#endif
    constexpr long long HALO = 1024LL * HROWS, L = NK - HALO;
    // (channel, tile) of this workgroup's current and next tile, advanced without a division: the flat list is walked
    // grid-strided, and stride = dq big_tiles + dr is split once
    const int tiles = (int)a.big_tiles, channels = a.channels;
    if ((long long)blockIdx.x >= a.big_tiles * channels) return;
    // Neighbouring tiles share a halo of 4096 or 8192 samples (a third or all of what a tile reads).  Workgroups go
    // to the eight XCDs round-robin by blockIdx, each XCD with an L2 of its own: with tile = blockIdx every halo was
    // fetched through two L2s -- 1.36x the algorithmic reads at 4095 taps (profiles/r03_c3_summary.txt).  A round of
    // the grid is dealt in eight contiguous runs instead, one per XCD, and the rows two tiles share are loaded with
    // the default cache policy (the rest non-temporal): the neighbour finds them in its own XCD's L2.
    unsigned first = blockIdx.x;
#if SDR16K_XCD
    if ((gridDim.x & 7u) == 0) first = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
#endif
    int ch = (int)(first / (unsigned)tiles), t = (int)(first % (unsigned)tiles);
    int dq = (int)(gridDim.x / (unsigned)tiles), dr = (int)(gridDim.x % (unsigned)tiles);
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
    f2 nx[16];
    {
        const S* p = in_of(ch, t);
#pragma unroll
        for (int r = 0; r < 16; ++r) nx[r] = load_row(p, r);
    }
    Lane16k c;
    c.init(tid, reinterpret_cast<const f2*>(a.wtab16k), lds + 16 * PK);
    // every global access below is a wave-uniform base + this lane's 32-bit offset: one VGPR of addressing for all
    // sixteen rows (per-lane 64-bit pointers cost two registers per row, and the first build spilled exactly those)
    const f2* const hbase = reinterpret_cast<const f2*>(a.hperm16k);
    for (;;) {
        // (opaque per tile: the spectrum's sixteen addresses are loop-invariant, and hoisted out of the loop they
        // were sixteen 64-bit registers that went straight to scratch)
        asm volatile("" : "+v"(lane));
        f2 v[16];
#if !SDR16K_PREFETCH
        {
            const S* p = in_of(ch, t);
#pragma unroll
            for (int r = 0; r < 16; ++r) nx[r] = load_row(p, r);
        }
#endif
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = nx[i];
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
#if SDR16K_PREFETCH
#pragma unroll
        for (int r = 0; r < 8; ++r) nx[r] = load_row(pn, r);
#endif
        S* q0 = reinterpret_cast<S*>(a.out) + ch * a.out_stride + a.big_first + t * PER_TILE - HALO;
#if SDR16K_STREAM
        inverse16k_s(w, lds, tid, c, [&] {
#if SDR16K_PREFETCH
#pragma unroll
            for (int r = 8; r < 16; ++r) nx[r] = load_row(pn, r);
#endif
        }, [&](int r, f2 val) {
            if (r >= HROWS) store_row(q0, r, val);
        });
#else
        inverse16k(w, lds, tid, c, [&] {
#if SDR16K_PREFETCH
#pragma unroll
            for (int r = 8; r < 16; ++r) nx[r] = load_row(pn, r);
#endif
        });
#pragma unroll
        for (int r = HROWS; r < 16; ++r) store_row(q0, r, w[P16k(r)]);
#endif
        ch = chn;
        t = tn;
        if (!more) break;
    }
}

} // namespace

hipError_t launch_fir_fft16k_interior(const FftArgs& a, bool real, int grid_blocks, hipStream_t stream) {
    const long long total = a.big_tiles * a.channels;
    if (total <= 0) return hipSuccess;
    if (!a.hperm16k || !a.wtab16k || a.big_tiles > 0x7fffffffLL) return hipErrorInvalidValue;
    const dim3 g((unsigned)(total < grid_blocks ? total : grid_blocks)), b(TK);
    if (a.halo16k == 4096) {
        if (real) hipLaunchKernelGGL((fir_fft16k<4, true>), g, b, 0, stream, a);
        else hipLaunchKernelGGL((fir_fft16k<4, false>), g, b, 0, stream, a);
    } else if (a.halo16k == 8192) {
        if (real) hipLaunchKernelGGL((fir_fft16k<8, true>), g, b, 0, stream, a);
        else hipLaunchKernelGGL((fir_fft16k<8, false>), g, b, 0, stream, a);
    } else {
        return hipErrorInvalidValue;
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
