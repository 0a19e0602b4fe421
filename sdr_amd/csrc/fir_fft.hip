// fir_fft.hip -- overlap-save streaming FIR with an in-LDS 4096-point FFT, gfx950.
//
// Same contract as fir_direct (y[i] = sum_j h[j] x[i-j] over the virtual stream
// history|input), but the arithmetic is a fast convolution: the direct form costs
// ntaps packed FMAs per complex sample and is VALU-bound from ~35 dense taps up
// (DESIGN.md 4.1, 4.2); this kernel costs ~45 packed ops per sample whatever the tap
// count (one transform pair up to 2049 taps, partitioned beyond, up to 131072),
// which puts medium and long filters back under the HBM roofline.
//
// One workgroup (256 lanes) owns tiles of N = 4096 input samples: halo (>= ntaps-1,
// multiple of 64) old samples + L = N - halo new ones; it produces L outputs per tile
// and walks over tiles persistently.  Each lane keeps 16 complex points in VGPRs:
//
//   forward (decimation in frequency, natural order in -> digit-reversed out):
//     n = 256 n2 + 16 n1 + n0,  k = k0 + 16 k1 + 256 k2
//     A: lane r = 16 n1 + n0 loads x[256 n2 + r] straight from HBM (coalesced 8 B/lane),
//        DFT16 over n2 -> k0, times W_4096^(r k0)
//     exchange 1 through LDS ([k0][r], 16-complex pad per k0 row: conflict-free both ways)
//     B: lane (k0,n0): DFT16 over n1 -> k1, times W_256^(n0 k1)
//     exchange 2 (16x16 transpose inside each 16-lane group, row stride 17)
//     C: lane (k0,k1): DFT16 over n0 -> k2        => X[k0 + 16 k1 + 256 k2]
//   pointwise product with the filter spectrum H/N (held in VGPRs, one load per lane
//   per kernel), then the mirrored inverse (C', exchange 2', B', exchange 1', A') with
//   conjugated twiddles, and lane r stores y[256 n2 + r] for the L new positions.
//
// The W_4096 twiddles and the filter spectrum are per-lane constants kept in registers for
// the life of the (persistent) workgroup, the W_256 twiddles sit in a 256-entry LDS table;
// all come from tables computed on the host in double (no sincos on the device).  The next
// tile's loads are issued in shares between the phases of the current tile (SDR_SPREAD).  Real signals ride two-at-a-time as re/im of one complex
// FFT (the filter is real-tap, hence linear over that pairing), like in fir_direct.
//
// fp32 error: ~3e-7 of the tile's RMS (measured against a float64 convolution), inside
// the 1e-5 bar; unlike the direct form it is relative to the tile, not to each output.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include "fir_kernels.h"
#include "fft_math.h"
#include "fft_call.h"

// shares the next tile's 16 row loads are issued in: 0 = one burst, 1 = three, 2 = six, 3 = eight
#ifndef SDR_SPREAD
#define SDR_SPREAD 2
#endif
// Four choices of the spectrum-cache variant (C3), each measured against its alternative on the same box
// (profiles/r02_fft_variants.txt).  The second partition's spectrum in LDS (1) or read from L2 per tile (0):
// C3 0.415 / 0.425 ms, real samples 0.394 / 0.420.
#ifndef SDR_CACHE_H1_LDS
#define SDR_CACHE_H1_LDS 1
#endif
// the next tile's loads in shares like the single-spectrum variants (1) or in one burst (0).
// With its second spectrum in LDS the kernel has the registers for it: C3 0.418 -> 0.396 ms (level 2; 1: 0.404, 3: 0.399)
#ifndef SDR_CACHE_SPREAD
#define SDR_CACHE_SPREAD 1
#endif
// carry and edge workgroups lead the main launch (1) like in the single-spectrum variants, or
// run as a launch of their own behind it (0).  Round 1 measured the shared kernel 20 % slower (its main loop was
// scheduled worse at 252 VGPRs); at 200 it is 1.5-2.5 % faster than two launches (C3 0.390 / 0.395 ms).
#ifndef SDR_CACHE_ONE_LAUNCH
#define SDR_CACHE_ONE_LAUNCH 1
#endif
// W_256 twiddles in registers (0) or read from the LDS table (1).  At two workgroups per CU
// the kernel has the 27 registers, and 30 fewer LDS reads per tile are worth 2 % (C3 0.384 / 0.391 ms).
#ifndef SDR_CACHE_TBL
#define SDR_CACHE_TBL 0
#endif

// An exchange's 16 LDS reads are issued as one batch and waited for one by one (1), or left to the compiler's
// scheduler (0), which strings them out -- read, wait for it, two or three instructions, next read -- and
// exposes the LDS latency some forty times per tile.  Same box: C5 0.8825 -> 0.8738 ms, C3 0.3917 -> 0.3886,
// C2 and the 64-channel chain unchanged (profiles/r02_fft_variants.txt).
#ifndef SDR_READ_BATCH
#define SDR_READ_BATCH 1
#endif
#if SDR_READ_BATCH
#define SDR_READ_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define SDR_READ_FENCE() ((void)0)
#endif
// Fused-generator kernels: how many of a stage's fifteen W_256 twiddles (LDS table) are requested in the batch
// of the stage's data; the rest follow in a second batch behind the butterflies.  1 = each at its use (what the
// other kernels do: they have no registers to spare, batching costs C2 2.8 %).  C5: 1 0.874, 4 0.863, 8 0.855,
// 16 0.860 ms on one box.
#ifndef SDR_TW_BATCH
#define SDR_TW_BATCH 8
#endif
// spectrum-cache variant: the first partition's spectrum in registers (1) or read from L2 per tile (0).
// Memory operations return in order: 16 spectrum loads issued in the middle of a tile queue up behind the
// shares of the next tile's prefetch issued before them, and the product waited out an HBM latency per tile
// for data that sat in L2.  221 VGPRs with the spectrum resident (the kernel runs two workgroups per CU,
// 256 each).  C3 0.389 -> 0.360 ms, real samples 0.380 -> 0.352 (same box; profiles/r02_fft_variants.txt).
#ifndef SDR_CACHE_H0_REGS
#define SDR_CACHE_H0_REGS 1
#endif
// Fused-generator kernels: the sine behind the (bit-exact) phase argument with fused multiply-adds (1,
// gen_device.h: gen_sine_pair_fast, 13 operations per sample) or in the reference's own operation order (0, 18).
#ifndef SDR_GEN_TBL
#define SDR_GEN_TBL 1
#endif
#ifndef SDR_GEN_FAST
#define SDR_GEN_FAST 1
#endif
// Diagnostic builds (make diag): FftArgs::ablate can switch the arithmetic off to expose the
// memory pattern.  The release library compiles the switch out.
#ifdef SDR_AMD_DIAG
#define SDR_ABLATE(a) ((a).ablate)
#define SDR_CONTIG(a) ((a).contig)
#else
#define SDR_ABLATE(a) 0
#define SDR_CONTIG(a) 0
#endif
namespace sdr_amd {
namespace {

constexpr int FFT_N = 4096;
constexpr int ROW = 272; // complex elements per k0 row in LDS (256 + 16 pad)

// In-place 16-point DFT (INV: inverse, unscaled).  Input v[n] in natural order;
// output X[k] is left at v[P16(k)], P16(k) = 4*(k&3) + (k>>2).
__host__ __device__ constexpr int P16(int k) { return 4 * (k & 3) + (k >> 2); }

template <bool INV>
__device__ __forceinline__ void dft16(f2 (&v)[16]) {
    // n = 4a + b: radix-4 over a for every b -> t[b][c] at v[4c + b]
#pragma unroll
    for (int b = 0; b < 4; ++b) radix4<INV>(v[b], v[4 + b], v[8 + b], v[12 + b]);
    // twiddle W16^(b c)
    constexpr float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f; // cos, sin(pi/8)
    constexpr float C2 = 0.70710678118654752440f;
    constexpr float sg = INV ? 1.0f : -1.0f; // forward W = exp(-i..): (cos, -sin)
    v[4 * 1 + 1] = cmul_k(v[4 * 1 + 1], C1, sg * S1);   // W16^1
    v[4 * 1 + 2] = cmul_k(v[4 * 1 + 2], C2, sg * C2);   // W16^2
    v[4 * 1 + 3] = cmul_k(v[4 * 1 + 3], S1, sg * C1);   // W16^3
    v[4 * 2 + 1] = cmul_k(v[4 * 2 + 1], C2, sg * C2);   // W16^2
    {                                                    // W16^4 = -i (fwd) / +i (inv)
        const f2 z = {0.0f, 0.0f};
        v[4 * 2 + 2] = INV ? add_pi(z, v[4 * 2 + 2]) : add_mi(z, v[4 * 2 + 2]);
    }
    v[4 * 2 + 3] = cmul_k(v[4 * 2 + 3], -C2, sg * C2);  // W16^6
    v[4 * 3 + 1] = cmul_k(v[4 * 3 + 1], S1, sg * C1);   // W16^3
    v[4 * 3 + 2] = cmul_k(v[4 * 3 + 2], -C2, sg * C2);  // W16^6
    v[4 * 3 + 3] = cmul_k(v[4 * 3 + 3], -C1, -sg * S1); // W16^9
    // radix-4 over b for every c -> X[c + 4d] at v[4c + d]
#pragma unroll
    for (int c = 0; c < 4; ++c) radix4<INV>(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
}

} // namespace

// Per-lane constants, computed once per (persistent) workgroup and kept in VGPRs:
//   TA[k] = W_4096^(tid k), TB[k] = W_256^((tid&15) k), H[k2] = spectrum/N of bin
//   (tid>>4) + 16 (tid&15) + 256 k2.
// TBL: the W_256 twiddles are read from a 256-entry LDS table instead of 30 VGPRs.
template <bool TBL, int TWB = 1>
struct LaneConstsT {
    static constexpr int tw_batch = TBL ? TWB : 1; // see SDR_TW_BATCH
    f2 TA[16], TB[TBL ? 1 : 16], H[16];
    const f2* tb_lds;
    __device__ __forceinline__ f2 tb(int k) const {
        if constexpr (TBL) return tb_lds[16 * k];
        else return TB[k];
    }
    __device__ __forceinline__ f2 ta(int k) const { return TA[k]; }
    // W_256 twiddles LO..HI-1 of a stage, requested as one batch (TBL: LDS reads)
    template <int LO, int HI>
    __device__ __forceinline__ void tb_some(f2 (&t)[16]) const {
#pragma unroll
        for (int k = LO; k < HI; ++k) t[k] = tb(k);
    }
    // wtab: the per-lane twiddles, laid out by the host (double, rounded to fp32) exactly as the lanes take
    // them: wtab[256 k + tid] = W_4096^(tid k) for k = 0..15, then 256 entries W_256^(n k) at [4096 + 16 k + n].
    // 15 coalesced loads per lane (L2-hot: every workgroup reads the same 34 KB) and one LDS write; no
    // staging through the exchange buffer and no barrier of its own -- the 256-entry table is first read in
    // stage B of a forward transform, behind that transform's block barrier.  (Round 1 copied a 4096-entry
    // exp table into LDS and gathered from it: two barriers and 30 conflicting LDS reads per workgroup, a
    // quarter of a C1-sized call.)  `stage` is unused; `table`: 256 entries for TBL.
    __device__ __forceinline__ void init(int tid, const float* hperm, bool with_spectrum, const f2* wtab, f2* stage,
                                         f2* table = nullptr) {
        (void)stage;
        tb_lds = nullptr;
#pragma unroll
        for (int k = 1; k < 16; ++k) TA[k] = wtab[256 * k + tid];
        if constexpr (TBL) {
            // table[16 k + n] = W_256^(n k); a lane reads column n = tid & 15 (broadcast within 16-lane groups)
            table[tid] = wtab[4096 + tid];
            tb_lds = table + (tid & 15);
        } else {
            (void)table;
#pragma unroll
            for (int k = 1; k < 16; ++k) TB[k] = wtab[4096 + 16 * k + (tid & 15)];
        }
        if (with_spectrum) {
#pragma unroll
            for (int k = 0; k < 16; ++k) H[k] = reinterpret_cast<const f2*>(hperm)[k * 256 + tid];
        }
    }
};

// Forward transform of one tile: v[n2] = x[256 n2 + tid] in, X[(tid>>4) + 16 (tid&15)
// + 256 k2] left at v[P16(k2)].  `after_points_left` runs once the input points have
// left the registers (the slot where the next tile's loads are issued).
// `after_forward`: the previous use of the exchange buffer was another forward transform,
// whose last reads (stage C) cover whole rows -- columns of other waves included -- so
// everyone must be done before this one writes.  After an INVERSE transform no barrier is
// needed here: past its block barrier every wave only reads its own columns (stage A'),
// and the writes below touch only the writer's own columns.
template <class Consts, class Hook, class HookB = NoHook, class HookR = NoHook>
__device__ __forceinline__ void forward_tile(f2 (&v)[16], f2* lds, int tid, const Consts& c, Hook&& after_points_left,
                                             bool after_forward = false, HookB&& after_stage_b = HookB{},
                                             HookR&& after_b_reads = HookR{}) {
    const int l16 = tid & 15;
    f2* row1 = lds + (tid >> 4) * ROW; // this 16-lane group's k0 row
    // ---- A: over n2 -> k0, twiddle W_4096^(r k0) ---------------------------------
    dft16<false>(v);
#pragma unroll
    for (int k = 1; k < 16; ++k) v[P16(k)] = cmul(v[P16(k)], c.ta(k));
    if (after_forward) __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[k * ROW + tid] = v[P16(k)];
    after_points_left();
    __syncthreads();
    // ---- B: lane (k0, n0 = l16), over n1 -> k1, twiddle W_256^(n0 k1) --------------
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) v[n1] = row1[16 * n1 + l16];
    f2 tw[16];
    constexpr int TWN = Consts::tw_batch; // twiddles requested with the data; the rest behind the butterflies
    c.template tb_some<1, TWN>(tw);
    SDR_READ_FENCE();
    after_b_reads();
    dft16<false>(v);
    if (TWN > 1 && TWN < 16) {
        c.template tb_some<TWN, 16>(tw);
        SDR_READ_FENCE();
    }
#pragma unroll
    for (int k = 1; k < 16; ++k) v[P16(k)] = cmul(v[P16(k)], TWN > 1 ? tw[k] : c.tb(k));
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 16; ++k) row1[k * 17 + l16] = v[P16(k)];
    after_stage_b();
    __builtin_amdgcn_wave_barrier();
    // ---- C: lane (k0, k1 = l16), over n0 -> k2 ------------------------------------
#pragma unroll
    for (int n0 = 0; n0 < 16; ++n0) v[n0] = row1[l16 * 17 + n0];
    SDR_READ_FENCE();
    dft16<false>(v);
}

// Inverse transform: w[k2] = Y[(tid>>4) + 16 (tid&15) + 256 k2] in (natural k2 order),
// y[256 n2 + tid] left at w[P16(n2)].
// `before_barrier` runs after the last group-local write, ahead of the block barrier (a
// slot for a share of the next tile's loads).
template <class Consts, class Hook, class HookC = NoHook, class HookR = NoHook>
__device__ __forceinline__ void inverse_tile(f2 (&w)[16], f2* lds, int tid, const Consts& c, Hook&& before_barrier,
                                             HookC&& after_stage_c = HookC{}, HookR&& after_b_reads = HookR{}) {
    const int l16 = tid & 15;
    f2* row1 = lds + (tid >> 4) * ROW;
    // ---- C': over k2 -> n0 --------------------------------------------------------
    dft16<true>(w);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int n0 = 0; n0 < 16; ++n0) row1[l16 * 17 + n0] = w[P16(n0)];
    after_stage_c();
    __builtin_amdgcn_wave_barrier();
    // ---- B': lane (k0, n0 = l16): conj twiddle, over k1 -> n1 ----------------------
#pragma unroll
    for (int k = 0; k < 16; ++k) w[k] = row1[k * 17 + l16];
    f2 tw[16];
    constexpr int TWN = Consts::tw_batch;
    c.template tb_some<1, TWN>(tw);
    SDR_READ_FENCE();
    after_b_reads();
#pragma unroll
    for (int k = 1; k < TWN; ++k) w[k] = cmulc(w[k], tw[k]);
    if (TWN > 1 && TWN < 16) {
        c.template tb_some<TWN, 16>(tw);
        SDR_READ_FENCE();
    }
#pragma unroll
    for (int k = TWN; k < 16; ++k) w[k] = cmulc(w[k], TWN > 1 ? tw[k] : c.tb(k));
    dft16<true>(w);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) row1[16 * n1 + l16] = w[P16(n1)];
    before_barrier();
    __syncthreads();
    // ---- A': lane r: conj twiddle, over k0 -> n2 ----------------------------------
#pragma unroll
    for (int k = 0; k < 16; ++k) w[k] = lds[k * ROW + tid];
    SDR_READ_FENCE();
#pragma unroll
    for (int k = 1; k < 16; ++k) w[k] = cmulc(w[k], c.ta(k));
    dft16<true>(w);
}

// Filter spectrum of partition p, bin (tid>>4) + 16 (tid&15) + 256 k2, from L2.
__device__ __forceinline__ f2 spectrum(const float* hperm, int p, int k2, int tid) {
    // wave-uniform base + 32-bit lane offset: one VGPR of addressing for all 16 loads
    return (reinterpret_cast<const f2*>(hperm) + (p * FFT_N + k2 * 256))[(unsigned)tid];
}

// ---- interior tiles: straight-line, software-pipelined -----------------------------
// Tiles whose inputs and outputs all lie inside [0, n).  halo = 64*HQ is a
// compile-time constant so that loads and stores are unconditional and the compiler's
// vmcnt bookkeeping is exact: the next tile's 16 loads are issued right after the
// current points have left the registers and are only waited for at the top of the
// next iteration, past the current tile's stores.
//
// LONG = uniformly partitioned overlap-save for more than 2049 taps: hop L = 2048,
// the filter is cut into `parts` pieces of 2048 taps, Y_b = sum_p X_(b-p) H_p; the
// windows b-1.. are re-read (L2) and re-transformed, spectra H_p come from L2.
// CACHE (LONG with exactly two partitions, 2050..4096 taps): every workgroup walks a
// contiguous run of tiles and keeps X_(b-1) -- the spectrum it computed for the
// previous tile -- in registers, so a tile costs one forward and one inverse
// transform instead of two forward and one inverse.
// GEN (real signals): the input is synthesised in registers by the reference
// generator's recipe (gen_device.h) instead of being loaded -- the fused
// gen -> hilbert -> FIR chain writes 4 bytes per sample and reads none.
// Workgroup `bid` of `nblk` interior workgroups.
template <bool REAL, int HQ, bool LONG, bool TBL, bool CACHE, int GEN>
__device__ __forceinline__ void interior_tiles(const FftArgs& a0, f2* lds, LaneConstsT<TBL, GEN ? SDR_TW_BATCH : 1>& c, const int bid,
                                               const int nblk) {
    // The persistent loop is short of scalar registers, and the compiler's way out is to re-load a kernel argument
    // from the argument segment wherever it is used -- inside the tile loop, behind an s_waitcnt lgkmcnt(0) that
    // also drains the wave's LDS reads (3.5 % of C2 when it happened to `channels`, profiles/r03_device_state_ab.txt;
    // which argument it picks changes with any edit of FftArgs).  The arguments the loop uses are copied once and
    // made opaque: the copies live in registers (or VGPR lanes), never behind a load.  `make asm` +
    // scripts/asm_loops.py lists the scalar loads inside every loop of a kernel: check it after touching FftArgs.
    // Only where it happens -- the fused-generator kernels: the others keep their arguments in registers by
    // themselves, and the copies cost them 2.5 % (the allocator then spills other values to VGPR lanes).
    FftArgs al = a0;
    if constexpr (GEN != 0) {
        asm volatile("" : "+s"(al.out), "+s"(al.out_stride), "+s"(al.channels), "+s"(al.t_lo), "+s"(al.t_hi), "+s"(al.tickets));
        asm volatile("" : "+s"(al.gen.phi), "+s"(al.gen.phi_incr), "+s"(al.gen.amp), "+s"(al.gen.cps), "+s"(al.gen.pincr));
    }
    if constexpr (GEN == 0 && REAL) asm volatile("" : "+s"(al.in_stride), "+s"(al.channels)); // (the two it re-loads there)
    const FftArgs& a = (GEN != 0 || REAL) ? al : a0;
    static_assert(!LONG || HQ == 32, "partitioned mode uses hop 2048");
    static_assert(HQ >= 1 && HQ <= 32, "halo = 64 * HQ, at most half a tile");
    static_assert(!CACHE || LONG, "the spectrum cache belongs to the partitioned mode");
    static_assert(!GEN || REAL, "the generator source produces real samples");
    const int tid = threadIdx.x;
    constexpr bool SPREAD = (!LONG || (CACHE && SDR_CACHE_SPREAD)) && SDR_SPREAD; // single-spectrum variants: spread the prefetch (see the main loop)
    constexpr int W = REAL ? 1 : 2;
    constexpr int HALO = 64 * HQ;
    constexpr long long L = FFT_N - HALO;
    constexpr long long per_tile = REAL ? 2 * L : L;
    const long long inner = a.t_hi - a.t_lo;             // interior tiles per channel
    const long long total = inner * a.channels;

    // (channel, tile) of this workgroup's current and next tile, advanced without
    // 64-bit divisions; tt counts tiles of the channel from a.t_lo
    // tile -> workgroup: contiguous runs (always with the spectrum cache; a.contig for
    // the others) or grid-strided
    const bool runs = CACHE || SDR_CONTIG(a);
    struct Pos { long long ch, tt; };
    auto advance = [&](Pos p) {
        p.tt += runs ? 1 : nblk;
        while (p.tt >= a.t_hi) { p.tt -= inner; ++p.ch; }
        return p;
    };
    // Real signals ride in pairs (re, im) = (x[o + m], x[o + D + m]).  Normally the pair is
    // the two halves of one 2L-sample tile (D = L).  With the spectrum cache both members
    // must advance by one hop per tile so that the previous tile's spectrum IS X_(b-1):
    // the re member walks the first half of the channel's interior, the im member the
    // second half (D = inner * L), one hop of L samples per tile each.
    const long long D = (REAL && CACHE) ? inner * L : L;
    auto first_out = [&](Pos p) -> long long {            // first output sample of the tile's re member
        if (REAL && CACHE) return a.t_lo * per_tile + (p.tt - a.t_lo) * L;
        return p.tt * per_tile;
    };
    auto in_ptr = [&](Pos p) -> const float* {            // wave-uniform pointer to x[s0]
        return a.in + (p.ch * a.in_stride + first_out(p) - HALO) * W;
    };
    // rows [LO, HI) of a tile's 16 rows of 256 points
    // With the spectrum cache a workgroup walks consecutive hops of 2048 samples: rows 8..15 of
    // this window are rows 0..7 of its next one.  Those are loaded with the default cache policy
    // (they come back out of L2 one tile later); rows read for the last time stay non-temporal.
    // (r01: every row non-temporal -> each input fetched from HBM twice, 1.93x the reads.)
    auto row = [&](const float* p0, int n2) -> f2 {
        const bool again = CACHE && n2 >= 8;
        if (REAL) {
            const float* q = p0 + 256 * n2;
            return again ? f2{q[tid], q[D + tid]} : f2{stream_load(q + tid), stream_load(q + D + tid)};
        }
        const f2* q = reinterpret_cast<const f2*>(p0 + 512 * n2) + tid;
        return again ? *q : stream_load(q);
    };
    auto load_rows = [&](const float* p0, f2 (&dst)[16], auto lo, auto hi) {
#pragma unroll
        for (int n2 = decltype(lo)::value; n2 < decltype(hi)::value; ++n2) dst[n2] = row(p0, n2);
    };
    auto load = [&](const float* p0, f2 (&dst)[16]) {
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) dst[n2] = row(p0, n2);
    };
    // GEN: the pair (x[o + 256 n2 + tid], x[o + D + 256 n2 + tid]) for n2 = 0..15 (o >= 0,
    // wave-uniform).  Every lane walks its own two samples down the 16 rows: in-block index
    // (a float: exact integers below 2^24) and start phase of its block.  A row step adds 256
    // to the index; a lane that runs off its block (B >= 256: at most once per step) moves to
    // the next block's phase by the host's own recurrence (sdr_capi.hip gen_next_table: one
    // fp32 addition, one conditional subtraction -- the same bits as the table's next entry),
    // so the walk touches no memory: only the two phases at its start come from the table.
    // With the usual B of a few thousand samples a wave wraps every B/256 rows: the row step
    // is one packed add and two compares ahead of a wave-uniform branch.  (r01 kept scalar
    // cursors per row and picked phases per lane: 43 VALU instructions per sample, 0.963 ms
    // for C5; profiles/r02_fft_variants.txt.)
    struct Walk { float idx, p, base; };    // base = p (+ 0.25 for cosine): what the sample formula takes
    struct Cursor { long long b; int rb; }; // block and in-block offset of a sample
    const int genB = GEN ? (int)a.gen.block_size : 1;
    const float genBf = (float)genB;
    const float gen_quarter = GEN && a.gen.waveform == 0 ? 0.25f : 0.0f;
    auto cursor_at = [&](long long o) {
        long long b = (long long)((double)o * a.gen.inv_block);
        long long r = o - b * genB;
        if (r < 0) { r += genB; --b; }
        else if (r >= genB) { r -= genB; ++b; }
        return Cursor{b, (int)r};
    };
    const Cursor genD = GEN ? cursor_at(D) : Cursor{0, 0}; // the im member runs D samples ahead of the re member
    // closed-form phases (GenSrc::pstate): the generator's position, read once per workgroup
    // (GEN = 2, a kernel of its own: as a run-time choice the two phase sources cost the persistent loop scalar
    // registers it does not have -- it re-loaded kernel arguments in every tile)
    constexpr bool gen_closed = GEN == 2;
    const unsigned genP0 = gen_closed ? a.gen.pstate->phase : 0u;
    auto walk_open = [&](Cursor c, Walk& s) __attribute__((always_inline)) {
        int i = c.rb + tid;
        const bool wrap = i >= genB;
        i = wrap ? i - genB : i;
        s.idx = (float)i;
        const unsigned blk = (unsigned)c.b + (wrap ? 1u : 0u);
        s.p = gen_closed ? gen_phase_closed(genP0, a.gen.pincr, blk) : a.gen.phi[blk];
        s.base = s.p + gen_quarter;
    };
    auto gen_open = [&](long long o, Walk& s0, Walk& s1) __attribute__((always_inline)) {
        const Cursor c0 = cursor_at(o);
        Cursor c1{c0.b + genD.b, c0.rb + genD.rb};
        if (c1.rb >= genB) { c1.rb -= genB; ++c1.b; }
        walk_open(c0, s0);
        walk_open(c1, s1);
    };
    auto walk_wrap = [&](Walk& s, bool wrap) __attribute__((always_inline)) {
        if (wrap) {
            s.idx -= genBf;
            const float t = s.p + a.gen.phi_incr; // both in [0, 1], 1 only where a fract() rounded up
            s.p = t >= 1.0f ? t - 1.0f : t;
            s.p = s.p >= 1.0f ? s.p - 1.0f : s.p;
            s.base = s.p + gen_quarter;
        }
    };
    // rows [LO, HI) of a window; the walkers stand at row LO - 1 (LO > 0) or at row 0 and are left at row HI - 1.
    // (Variants that cost more instructions on the common path, for the copies with which the compiler merges the
    // two paths of a step: both members in one vector-typed walker, the wrap written with selects, a wave-uniform
    // skip per member, a block of its own for the common path: 28-32 instructions per row where this form has 27.)
    auto gen_rows_of = [&](auto sine, Walk& s0, Walk& s1, f2 (&dst)[16], auto lo, auto hi) __attribute__((always_inline)) {
#pragma unroll
        for (int n2 = decltype(lo)::value; n2 < decltype(hi)::value; ++n2) {
            if (n2 > 0) {
                s0.idx += 256.0f;
                s1.idx += 256.0f;
                const bool w0 = s0.idx >= genBf, w1 = s1.idx >= genBf;
                if (__any(w0 || w1)) {
                    walk_wrap(s0, w0);
                    walk_wrap(s1, w1);
                }
            }
            if constexpr (decltype(sine)::value) { // straight-line polynomial on both members at once
                dst[n2] = SDR_GEN_FAST ? gen_sine_pair_fast(a.gen.amp, a.gen.cps, f2{s0.base, s1.base}, f2{s0.idx, s1.idx})
                                       : gen_sine_pair(a.gen.amp, a.gen.cps, f2{s0.base, s1.base}, f2{s0.idx, s1.idx});
            } else {
                dst[n2] = f2{gen_real_sample_f(a.gen.waveform, a.gen.amp, a.gen.cps, s0.p, s0.idx),
                             gen_real_sample_f(a.gen.waveform, a.gen.amp, a.gen.cps, s1.p, s1.idx)};
            }
            // the row's samples are computed before the walk moves on (the compiler otherwise sinks all
            // sixteen evaluations below the fifteen steps and copies every row's phase and index aside)
            asm volatile("" : "+v"(s0.idx), "+v"(s1.idx) : "v"(dst[n2]));
        }
    };
    auto gen_rows = [&](Walk& s0, Walk& s1, f2 (&dst)[16], auto lo, auto hi) __attribute__((always_inline)) {
        if (a.gen.waveform <= 1) gen_rows_of(IC<1>{}, s0, s1, dst, lo, hi);
        else gen_rows_of(IC<0>{}, s0, s1, dst, lo, hi);
    };
    auto synth = [&](long long o, f2 (&dst)[16]) __attribute__((always_inline)) {
        Walk s0, s1;
        gen_open(o, s0, s1);
        gen_rows(s0, s1, dst, IC<0>{}, IC<16>{});
    };
    // the tile's 4096-point window, `back` samples earlier (previous windows of the
    // partitioned mode): loaded, or synthesised
    auto fetch = [&](Pos p, long long back, f2 (&dst)[16]) __attribute__((always_inline)) {
        if constexpr (GEN) {
            synth(first_out(p) - HALO - back, dst);
        } else {
            load(in_ptr(p) - back * W, dst);
        }
    };

    if (inner <= 0 || (long long)bid >= total) return;
    // Ordered hand-out (single-spectrum and uncached partitioned variants): a workgroup's first two
    // tiles are bid and bid + nblk; every later one comes from a global ticket counter, in order.
    // Hardware dispatch keeps the tiles in flight a dense window of the buffers; grid-strided
    // persistent workgroups drift apart (CUs differ in memory latency), the window widens and the
    // same loop moves 5.0-5.4 TB/s where ordered tickets move 5.7 (profiles/r02_membench_tickets.txt).
    // The ticket for the tile after next is fetched at the top of a tile and published through one
    // LDS word at the inverse transform's block barrier: no exposed latency, no extra barrier.
    // FFT_TICKETS counters (a workgroup uses counter bid % FFT_TICKETS; one counter serialises at
    // ~12 ns per ticket, C2 needs one per 11 ns; more counters = looser order = slower).  Flat index g = channel * inner + tile (host: total < 2^31 in this mode).
    const bool ticketed = !CACHE && !runs && a.tickets != nullptr;
    unsigned* const tbox = reinterpret_cast<unsigned*>(lds + (ROW - 1)); // row 0, column 271: no exchange touches it
    const int tslot = bid % FFT_TICKETS;
    auto pos_of = [&](long long g) -> Pos {
        if (g >= total) return Pos{a.channels, a.t_lo};
        if (a.channels == 1) return Pos{0, a.t_lo + g};
        const unsigned ch = (unsigned)g / (unsigned)inner;
        return Pos{(long long)ch, a.t_lo + (long long)((unsigned)g - ch * (unsigned)inner)};
    };
    Pos cur{0, 0};
    long long left = 0;    // CACHE: tiles of this workgroup's run still to do
    long long prev_ch = -1; // CACHE: channel whose X_(b-1) sits in `prev`
    f2 prev[CACHE ? 16 : 1];
    if (runs) {
        const long long run = (total + nblk - 1) / nblk;
        const long long first = (long long)bid * run;
        if (first >= total) return;
        left = total - first < run ? total - first : run;
        cur.ch = first / inner;
        cur.tt = a.t_lo + (first - cur.ch * inner);
    } else {
        cur.tt = a.t_lo + bid;
        while (cur.tt >= a.t_hi) { cur.tt -= inner; ++cur.ch; }
    }
    Pos nxt = advance(cur); // the tile after `cur`; past the last one: ch >= channels (runs: see `left`)
    f2 nx[GEN ? 1 : 16]; // the next tile's points, loaded one tile ahead
    // GEN: the next tile's walkers instead -- their two phases are what comes from memory.  (Synthesising the
    // next tile's points in the prefetch shares as well, between the transform's phases, measured 3 % slower
    // than synthesising a tile's points at its top: profiles/r02_fft_variants.txt.)
    Walk g0{}, g1{};
    if constexpr (GEN) gen_open(first_out(cur) - HALO, g0, g1);
    else load(in_ptr(cur), nx);
    // the lane constants are set up with the first tile's loads already in flight (every
    // lane of the workgroup gets here or none does: the returns above are uniform)
    c.init(tid, a.hperm, !LONG || (CACHE && SDR_CACHE_H0_REGS), reinterpret_cast<const f2*>(a.wtab), lds, TBL ? lds + 16 * ROW : nullptr);
    if constexpr (CACHE && SDR_CACHE_H1_LDS) { // each lane reads back only what it wrote itself: no barrier needed
#pragma unroll
        for (int k = 0; k < 16; ++k) lds[16 * ROW + (TBL ? 256 : 0) + 256 * k + tid] = spectrum(a.hperm, 1, k, tid);
    }

    auto body = [&]() __attribute__((always_inline)) {
        f2 v[16];
        bool after_forward = false;
        // The ticket for the tile after next: one atomic by lane 0, issued once the prefetched points have left
        // their registers (forward transform's first hook: nothing at the top of the tile then waits behind it), its
        // result first needed when lane 0 publishes it at the inverse transform's barrier.  fir_fft.o is compiled
        // with the AMDGPU atomic optimizer off (Makefile): that pass rewrites the add into a wave reduction whose
        // v_readfirstlane sits right behind the atomic, behind an s_waitcnt vmcnt(0) -- every tile then began by
        // waiting out the atomic AND the previous tile's stores (0.775 ms for the bare copy loop where the
        // skeleton needs 0.726).
        unsigned ticket;
        auto fetch_ticket = [&] {
            if (!CACHE && ticketed && tid == 0) ticket = atomicAdd(a.tickets + FFT_TICKET_STRIDE * tslot, 1u);
        };
        if (CACHE && prev_ch != cur.ch) {
            // start of a run or of a channel: transform window b-1 once
            fetch(cur, 2048, v);
            forward_tile(v, lds, tid, c, [] {});
#pragma unroll
            for (int k = 0; k < 16; ++k) prev[CACHE ? k : 0] = v[P16(k)];
            prev_ch = cur.ch;
            after_forward = true;
        }
        if constexpr (GEN) {
            gen_rows(g0, g1, v, IC<0>{}, IC<16>{});
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = nx[GEN ? 0 : i];
        }
        // past its last tile a workgroup prefetches the first interior tile of the call -- loads
        // that nobody consumes, which keeps the body branch-free; one tile shared by every
        // workgroup stays in L2, where re-loading the workgroup's own tile went out to memory
        // (2304 x 32 KB per C2 launch)
        Pos pf = nxt;
        if (runs ? left <= 1 : nxt.ch >= a.channels) pf = Pos{0, a.t_lo};
        const float* pn = in_ptr(pf);
        if constexpr (GEN) gen_open(first_out(pf) - HALO, g0, g1);
        f2 w[16];
        if (SDR_ABLATE(a) & 1) {
            fetch_ticket();
            if constexpr (!GEN) load(pn, nx);
#pragma unroll
            for (int k = 0; k < 16; ++k) w[P16(k)] = v[k];

        } else {
            // The next tile's 16 row loads go out in shares spread over this tile's phases: the
            // memory system sees a steadier request stream than one burst per tile.  Share
            // boundaries per SDR_SPREAD level (slots: A hook, B reads, B hook, after forward,
            // after the pointwise product, C' hook, B' reads, before the inverse's barrier).
            constexpr int LV = SPREAD ? SDR_SPREAD : 0;
            constexpr int S0 = 0;
            constexpr int S1 = LV == 3 ? 2 : LV == 2 ? 3 : LV == 1 ? 5 : 16; // after the A hook
            constexpr int S2 = LV == 3 ? 4 : S1;                              // after B reads
            constexpr int S3 = LV == 3 ? 6 : LV == 2 ? 6 : S2;                // after the B hook
            constexpr int S4 = LV == 3 ? 8 : LV == 2 ? 9 : LV == 1 ? 10 : 16; // after forward
            constexpr int S5 = LV == 3 ? 10 : LV == 2 ? 11 : S4;              // after the product
            constexpr int S6 = LV == 3 ? 12 : LV == 2 ? 14 : S5;              // after the C' hook
            constexpr int S7 = LV == 3 ? 14 : S6;                             // after B' reads
            auto share = [&](auto lo, auto hi) {
                if constexpr (!GEN && decltype(lo)::value < decltype(hi)::value) load_rows(pn, nx, lo, hi);
            };
            forward_tile(v, lds, tid, c, [&] { fetch_ticket(); share(IC<S0>{}, IC<S1>{}); }, after_forward,
                         [&] { share(IC<S2>{}, IC<S3>{}); }, [&] { share(IC<S1>{}, IC<S2>{}); });
            share(IC<S3>{}, IC<S4>{});
#pragma unroll
            for (int k = 0; k < 16; ++k) w[k] = cmul(v[P16(k)], (LONG && !(CACHE && SDR_CACHE_H0_REGS)) ? spectrum(a.hperm, 0, k, tid) : c.H[k]);
            if (CACHE) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const f2 h1 = SDR_CACHE_H1_LDS ? lds[16 * ROW + (TBL ? 256 : 0) + 256 * k + tid] : spectrum(a.hperm, 1, k, tid);
                    w[k] += cmul(prev[CACHE ? k : 0], h1);
                    prev[CACHE ? k : 0] = v[P16(k)];
                }
            }
            if (LONG && !CACHE) {
                for (int p = 1; p < a.parts; ++p) {
                    fetch(cur, (long long)p * 2048, v); // window b-p (L2 resident)
                    forward_tile(v, lds, tid, c, [] {}, true);
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const f2 t = cmul(v[P16(k)], spectrum(a.hperm, p, k, tid));
                        w[k] += t;
                    }
                }
            }
            share(IC<S4>{}, IC<S5>{});
            inverse_tile(w, lds, tid, c,
                         [&] {
                             share(IC<S7>{}, IC<16>{});
                             if (!CACHE && ticketed && tid == 0) *tbox = ticket;
                         },
                         [&] { share(IC<S5>{}, IC<S6>{}); }, [&] { share(IC<S6>{}, IC<S7>{}); });
        }
        float* q0 = a.out + (cur.ch * a.out_stride + first_out(cur) - HALO) * W; // &y[-halo] of the tile
#pragma unroll
        for (int n2 = HALO / 256; n2 < 16; ++n2) {
            // a halo that ends inside a row (64 or 128 samples): the row's first waves hold
            // halo positions and store nothing -- a wave-uniform predicate.  (One store fewer on
            // one path makes the compiler's wait at the top of the next tile one operation
            // stricter than needed on the storing waves; a duplicate store of the next row in
            // this slot, which evens the counts, measured 0.6 % slower: r02_fft_variants.txt.)
            if (n2 == HALO / 256 && HALO % 256 != 0 && tid < HALO % 256) continue;
            const f2 y = w[P16(n2)];
            if (REAL) {
                float* q = q0 + 256 * n2;
                stream_store(y.x, q + tid);
                stream_store(y.y, q + D + tid);
            } else {
                stream_store(y, reinterpret_cast<f2*>(q0 + 512 * n2) + tid);
            }
        }
        if ((SDR_ABLATE(a) & 1) && !CACHE && ticketed) { // diagnostic builds only: the copy-only loop draws tickets too
            if (tid == 0) *tbox = ticket;
            __syncthreads();
        }
        cur = nxt;
        if (!CACHE && ticketed) {
            const unsigned tk = __builtin_amdgcn_readfirstlane(*tbox); // published before the inverse's barrier
            nxt = pos_of(2LL * nblk + (long long)tk * FFT_TICKETS + tslot);
        } else {
            nxt = advance(nxt);
        }
        --left;
    };
    // First tile peeled so that the loop header sees the same queue of outstanding
    // memory operations from both of its predecessors (16 loads, then this tile's
    // stores): the waits at the top of the loop then cover the loads only.
    body();
    while (runs ? left > 0 : cur.ch < a.channels) body();
}

// ---- edge tiles: the first tile(s) of every channel (they read the carried history)
// and the tiles from the last full one on (partial outputs); also every tile of a
// short call.  Bounds-checked, runtime halo / parts, no pipelining.  Workgroup `bid` of
// `nblk` edge workgroups.
template <bool REAL, bool GEN, class Consts>
__device__ __forceinline__ void edge_tiles(const FftArgs& a, const float* hist_base, f2* lds, const Consts& c, const int bid,
                                           const int nblk) {
    static_assert(!GEN || REAL, "the generator source produces real samples");
    const int tid = threadIdx.x;
    constexpr int W = REAL ? 1 : 2;
    const int halo = a.halo;
    const long long L = FFT_N - halo;
    const long long per_tile = REAL ? 2 * L : L;
    const long long tiles_per_ch = (a.n + per_tile - 1) / per_tile;
    const long long skipped = a.t_hi > a.t_lo ? a.t_hi - a.t_lo : 0; // handled by the interior workgroups
    const long long edge_per_ch = tiles_per_ch - skipped;
    const long long total = edge_per_ch * a.channels;

    for (long long e = bid; e < total; e += nblk) {
        const long long ch = e / edge_per_ch;
        long long tt = e - ch * edge_per_ch;
        if (skipped && tt >= a.t_lo) tt += skipped;
        const float* in = a.in + ch * a.in_stride * W;
        const float* hist = hist_base + ch * (long long)a.hist_len * W;
        float* out = a.out + ch * a.out_stride * W;
        const long long o0 = tt * per_tile; // first output of the tile
        auto sample = [&](long long m, int k) -> float { // virtual stream: history | input | 0
            if (m < 0) { const long long h = a.hist_len + m; return h >= 0 ? hist[h * W + k] : 0.0f; }
            if (m >= a.n) return 0.0f;
            if constexpr (GEN) return gen_src_at(a.gen, m);
            else return in[m * W + k];
        };
        f2 w[16];
        for (int p = 0; p < a.parts; ++p) {
            const long long s0 = o0 - halo - (long long)p * 2048; // first input of window b-p
            f2 v[16];
#pragma unroll
            for (int n2 = 0; n2 < 16; ++n2) {
                const long long m = s0 + 256 * n2 + tid;
                if (REAL) v[n2] = f2{sample(m, 0), sample(m + L, 0)};
                else v[n2] = f2{sample(m, 0), sample(m, 1)};
            }
            forward_tile(v, lds, tid, c, [] {}, p > 0);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const f2 t = cmul(v[P16(k)], spectrum(a.hperm, p, k, tid));
                w[k] = p == 0 ? t : w[k] + t;
            }
        }
        inverse_tile(w, lds, tid, c, [] {});
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) {
            const int m = 256 * n2 + tid - halo;
            if (m < 0) continue;
            const long long o = o0 + m;
            const f2 y = w[P16(n2)];
            if (REAL) {
                if (o < a.n) out[o] = y.x;
                if (o + L < a.n) out[o + L] = y.y;
            } else if (o < a.n) {
                *reinterpret_cast<f2*>(out + 2 * o) = y;
            }
        }
    }
}

// ---- one launch per call.  The first a.grid_carry workgroups carry the history, the
// next a.grid_edge take the edge tiles, the rest are the persistent interior
// workgroups: the few slow, bounds-checked tiles start first and overlap the bulk
// instead of trailing it as separate launches.
template <bool REAL, int HQ, bool LONG, bool TBL, bool CACHE, int GEN>
__global__ __launch_bounds__(256, (TBL && !LONG) ? 3 : 2) void fir_fft4096(FftArgs a) {
    // (padding the LDS to a third of the CU -- which in the copy skeletons moves 4 % more bytes, because 54 KB
    // workgroups end up TWO per CU and fewer bytes in flight suit the memory system -- costs this kernel 3.5 %:
    // the transform needs the third workgroup.  profiles/r02_fft_variants.txt)
    // CACHE: the second partition's spectrum lives in LDS (32 KB more: two workgroups per CU still fit), copied
    // once per workgroup -- the product then waits on 16 L2 loads and 16 LDS reads instead of 32 L2 loads
    __shared__ __align__(16) f2 lds[16 * ROW + (TBL ? 256 : 0) + ((CACHE && SDR_CACHE_H1_LDS) ? 4096 : 0)];
    int bid = blockIdx.x;
    LaneConstsT<TBL, GEN ? SDR_TW_BATCH : 1> c;
    constexpr bool ONE_LAUNCH = !LONG || (CACHE && SDR_CACHE_ONE_LAUNCH);
    // (the uncached partitioned variants keep their short work in a launch of its own, fir_fft4096_short,
    // which also ends the call; their interior workgroups touch neither history nor state)
    if constexpr (ONE_LAUNCH) {
        if (bid < a.grid_carry) {
            const unsigned cur = a.state->cur & 1u;
            carry_history<GEN != 0>(a, pick(a.hist, cur), pick(a.hist, cur ^ 1u), REAL ? 1 : 2, bid, a.grid_carry);
            call_done();
            return;
        }
        bid -= a.grid_carry;
        if (bid < a.grid_edge) {
            const unsigned cur = a.state->cur & 1u;
            c.init(threadIdx.x, a.hperm, false, reinterpret_cast<const f2*>(a.wtab), lds, TBL ? lds + 16 * ROW : nullptr);
            edge_tiles<REAL, GEN != 0>(a, pick(a.hist, cur), lds, c, bid, a.grid_edge);
            call_done();
            return;
        }
        bid -= a.grid_edge;
    }
    interior_tiles<REAL, HQ, LONG, TBL, CACHE, GEN>(a, lds, c, bid,
                                                    ONE_LAUNCH ? (int)gridDim.x - a.grid_carry - a.grid_edge : (int)gridDim.x);
    if constexpr (ONE_LAUNCH) call_done(); // (interior workgroups touch neither history nor state before this)
}

// carry + edge workgroups alone (partitioned filters; calls without interior tiles)
template <bool REAL, bool GEN>
__global__ __launch_bounds__(256, 2) void fir_fft4096_short(FftArgs a) {
    __shared__ __align__(16) f2 lds[16 * ROW + 256];
    const int bid = blockIdx.x;
    // partitioned filters: this launch follows the main one on the stream, is never empty (the history is
    // always carried) and ends the call: its last workgroup zeroes the main launch's tickets and swaps the histories
    const unsigned cur = a.state->cur & 1u;
    if (bid < a.grid_carry) {
        carry_history<GEN>(a, pick(a.hist, cur), pick(a.hist, cur ^ 1u), REAL ? 1 : 2, bid, a.grid_carry);
    } else {
        LaneConstsT<true> c;
        c.init(threadIdx.x, a.hperm, false, reinterpret_cast<const f2*>(a.wtab), lds, lds + 16 * ROW);
        edge_tiles<REAL, GEN>(a, pick(a.hist, cur), lds, c, bid - a.grid_carry, a.grid_edge);
    }
    call_done();
}

// GEN: 0 = samples are loaded, 1 = synthesised with the host's phase table, 2 = with closed-form phases (GenSrc::pstate)
template <bool REAL, int GEN>
static hipError_t launch_tiles(const FftArgs& a, long long grid_int, hipStream_t stream) {
    const dim3 b(256);
    const int n_short = a.grid_carry + a.grid_edge;
    if (a.parts == 2 && SDR_CACHE_ONE_LAUNCH && !(SDR_ABLATE(a) & 4)) {
        hipLaunchKernelGGL((fir_fft4096<REAL, 32, true, SDR_CACHE_TBL != 0, true, GEN>), dim3((unsigned)(grid_int + n_short)), b, 0, stream, a);
        return hipGetLastError();
    }
    if (a.parts > 1) {
        const dim3 g((unsigned)grid_int);
        if (grid_int > 0) {
            if (a.parts == 2 && !(SDR_ABLATE(a) & 4)) hipLaunchKernelGGL((fir_fft4096<REAL, 32, true, true, true, GEN>), g, b, 0, stream, a);
            else hipLaunchKernelGGL((fir_fft4096<REAL, 32, true, false, false, GEN>), g, b, 0, stream, a);
        }
        // (forking this short launch onto a side stream so that it runs beside the main kernel was
        // tried: the two event hand-offs cost more than the 17 us it hides, C3 0.456 -> 0.480 ms)
        if (n_short > 0) hipLaunchKernelGGL((fir_fft4096_short<REAL, GEN != 0>), dim3((unsigned)n_short), b, 0, stream, a);
        return hipGetLastError();
    }
    const dim3 g((unsigned)(grid_int + n_short));
    // W_256 twiddles from the LDS table: 161 VGPRs, three workgroups per CU (+3 % over
    // the all-register build at two per CU, profiles/r01_fft_variants.txt)
    constexpr bool TBLK = GEN == 0 || SDR_GEN_TBL != 0; // W_256 twiddles: LDS table (three workgroups per CU) or registers (two)
    switch (a.halo / 64) {
        case 1: hipLaunchKernelGGL((fir_fft4096<REAL, 1, false, TBLK, false, GEN>), g, b, 0, stream, a); break;
        case 2: hipLaunchKernelGGL((fir_fft4096<REAL, 2, false, TBLK, false, GEN>), g, b, 0, stream, a); break;
        case 3: hipLaunchKernelGGL((fir_fft4096<REAL, 3, false, TBLK, false, GEN>), g, b, 0, stream, a); break;
        case 4: hipLaunchKernelGGL((fir_fft4096<REAL, 4, false, TBLK, false, GEN>), g, b, 0, stream, a); break;
        case 6: hipLaunchKernelGGL((fir_fft4096<REAL, 6, false, TBLK, false, GEN>), g, b, 0, stream, a); break;
        case 8: hipLaunchKernelGGL((fir_fft4096<REAL, 8, false, TBLK, false, GEN>), g, b, 0, stream, a); break;
        case 12: hipLaunchKernelGGL((fir_fft4096<REAL, 12, false, TBLK, false, GEN>), g, b, 0, stream, a); break;
        case 16: hipLaunchKernelGGL((fir_fft4096<REAL, 16, false, TBLK, false, GEN>), g, b, 0, stream, a); break;
        case 32: hipLaunchKernelGGL((fir_fft4096<REAL, 32, false, TBLK, false, GEN>), g, b, 0, stream, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// Tiling for a tap count: halo = 64 * {1,2,3,4,6,8,12,16,32} old samples per tile for up to 2049
// taps (one spectrum; 127 taps -> 128, i.e. 3968 new samples per 4096-point tile; a tile costs the
// same whatever its halo, so the classes in between -- 192, 384, 768 -- are worth their kernels:
// 700 taps 3328 instead of 3072 new samples per tile); beyond that hop 2048 with ceil(ntaps/2048) partitions.
void fir_fft_plan(size_t ntaps, int* halo, int* parts) {
    *parts = 1;
    for (int hq : {1, 2, 3, 4, 6, 8, 12, 16, 32})
        if (ntaps <= (size_t)64 * hq + 1) { *halo = 64 * hq; return; }
    *halo = 2048;
    *parts = (int)((ntaps + 2047) / 2048);
}

hipError_t launch_fir_fft(const FftArgs& a0, bool real, int grid_blocks, hipStream_t stream) {
    if (a0.n == 0 || a0.channels == 0) return hipSuccess;
    if (a0.gen.enabled && (!real || a0.channels != 1 || a0.gen.block_size < 256 || a0.gen.block_size > (1 << 23)))
        return hipErrorInvalidValue;
    FftArgs a = a0;
    const long long L = FFT_N - a.halo;
    const long long per_tile = real ? 2 * L : L;
    const long long tiles_per_ch = (a.n + per_tile - 1) / per_tile;
    // interior tiles: every input (all partitions) at index >= 0, every output < n
    const long long lead = a.parts > 1 ? (long long)a.parts * 2048 : a.halo;
    a.t_lo = (lead + per_tile - 1) / per_tile;
    a.t_hi = a.n / per_tile;
    // 16384-point tiles (two to four partitions' worth of taps: fir_fft16k.hip, the whole call in one launch of its
    // own) when the call fills the chip with them: worth it when the last round of one-tile-per-CU workgroups is full
    // enough -- per sample the big tile costs 0.84 of the cached two-partition form and about half of the three- and
    // four-partition forms (use16k == 2: the caller asked for this path by name, every call takes it)
    if (a.use16k && !a.gen.enabled && (a.parts >= 2 || a.halo == 2048) && a.hperm16k) { // (halo16k / parts16 cover the taps: sdr_capi.hip)
        long long P, first, big;
        fir_fft16k_tiles(a.n, a.halo16k, real, a.parts16 > 1 ? a.parts16 : 1, &P, &first, &big);
        const long long cus = a.cus16k > 0 ? a.cus16k : 256;
        const long long rounds = (big * a.channels + cus - 1) / cus;
        const long long need = a.parts <= 2 ? 84 : 50;
        // (its edge tiles run beside the bulk but cost ~40 us whatever the size: against the cached two-partition form
        // it takes two rounds of tiles to be ahead, profiles/r03_fft16k_one_launch.txt)
        // the three- and four-partition forms lose from 2^18 samples per call on, whatever the fill
        // (profiles/r03_fft16k_one_launch.txt: 8191 taps, 0.033 against 0.042 ms at 2^18 samples)
        const long long min_rounds = a.parts <= 2 ? 2 : 1;
        const bool long_enough = a.parts > 2 && a.n * a.channels >= (1LL << 18);
        // many short channels are all edges -- two bounds-checked tiles per channel on a handful of workgroups: the interior
        // must outweigh them (three to one against the cached two-partition form, one to one against the slower forms)
        const long long edge = ((a.n + P - 1) / P - big) * a.channels;
        const bool edges_small = edge * (a.parts <= 2 ? 3 : 1) <= big * a.channels;
        if (a.use16k == 2 ||
            (edges_small && (long_enough || (rounds >= min_rounds && big * a.channels * 100 >= rounds * cus * need))))
            return launch_fir_fft16k(a, real, (int)cus, stream);
    }
    const long long inner = a.t_hi > a.t_lo ? a.t_hi - a.t_lo : 0;
    if (inner == 0) a.t_lo = a.t_hi = 0;
    // equal shares: with T interior tiles and at most G workgroups every workgroup walks
    // ceil(T/G) tiles (or one fewer), instead of G workgroups of uneven load
    long long grid_int = 0;
    // ordered tickets: every variant but the spectrum cache (its workgroups walk runs) and the diagnostic
    // contiguous mode; 32-bit flat tile indices; every workgroup must be resident from the start (a
    // workgroup that starts late would take its two static tiles at the very end)
    bool ticketed = a.tickets && a.parts != 2 && !SDR_CONTIG(a) && inner > 0 &&
                    inner * a.channels + 4LL * grid_blocks < (1LL << 31);
    if (ticketed && a.resident > 0 && grid_blocks > a.resident) grid_blocks = a.resident;
    if (inner > 0) {
        const long long total = inner * a.channels;
        // a short call: at least two tiles per workgroup (its set-up costs about one), down to
        // the third of the grid that is resident at once
        long long cap = total / 2 > grid_blocks / 3 ? total / 2 : grid_blocks / 3;
        if (cap < 1) cap = 1;
        if (cap < grid_blocks) grid_blocks = (int)cap;
        const long long rounds = (total + grid_blocks - 1) / grid_blocks;
        grid_int = (total + rounds - 1) / rounds;
    }
    // a workgroup draws from counter bid % FFT_TICKETS: with fewer workgroups than counters some counters --
    // and the tiles they stand for -- would have no workgroup.  Such grids (tests, tiny calls) walk grid-strided.
    if (grid_int < FFT_TICKETS) ticketed = false;
    if (!ticketed) a.tickets = nullptr;
    const long long edge = (tiles_per_ch - inner) * a.channels;
    a.grid_edge = (int)(edge < grid_blocks ? edge : grid_blocks);
    const long long hist_elems = (long long)a.hist_len * (real ? 1 : 2) * a.channels;
    a.grid_carry = (int)((hist_elems + 255) / 256 < 64 ? (hist_elems + 255) / 256 : 64); // >= 1: every call carries
    if (a.gen.enabled) return a.gen.pstate ? launch_tiles<true, 2>(a, grid_int, stream) : launch_tiles<true, 1>(a, grid_int, stream);
    return real ? launch_tiles<true, 0>(a, grid_int, stream) : launch_tiles<false, 0>(a, grid_int, stream);
}

const char* fir_fft_kernel_name(bool real) { return real ? "fir_fft4096<REAL>" : "fir_fft4096<COMPLEX>"; }

} // namespace sdr_amd
