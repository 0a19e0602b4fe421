// fir_mfma.hip -- direct-form FIR on the matrix cores: the Toeplitz-GEMM formulation that
// BASELINE.json's north_star names, in exact fp32 (v_mfma_f32_16x16x4_f32: a k-ordered chain of
// fp32 fused multiply-adds, no reduced-precision split).
//
//   y[n] = sum_t g[t] x[n - t]   (g: Tg dense real taps, x: one real "plane")
//
// is evaluated 256 outputs at a time as D = A * B with
//   D[i][j] = y[ob + 16 j + i]                      16 x 16 outputs (one accumulator, 4 VGPRs)
//   A[i][u] = g[i + HB - u]  (0 outside the taps)    16 x K Toeplitz matrix of the taps, K = HB + 16
//   B[u][j] = x[ob + 16 j + u - HB]                  K x 16 Hankel matrix of the samples
// in K/4 steps of the 16x16x4 instruction.  The A operand of step s is one VGPR per lane, the same
// for every tile (loaded once per wave from a table the host builds); the B operand of step s is
// one ds_read_b32 per lane from the staged samples (LDS row stride 18 floats per 16 samples: the
// 64 lanes of a read then hit 64 different banks).  Useful share of the multiply-adds:
// Tg / (HB + 16) = 0.80 at 64 taps.
//
// Planes: a real-tap filter acts on every interleaved component separately, so P planes are the
// re / im parts of complex samples (P = 2), and the even / odd samples of a stream whose filter
// has every other tap zero -- the Hilbert transformer, reference include/sdr/hilbert.hpp:46-55 --
// (x2): its 127-tap design is two dense 64-tap half-rate filters.  P = W * stride in {1, 2, 4}.
//
// One workgroup (256 lanes) = one tile of 4096 floats (all planes), not persistent; each wave owns
// four accumulators (1024 outputs).  fp32 MFMA and packed-fp32 VALU have the same peak, so this
// kernel can only win where it holds the memory floor of small non-persistent tiles (5.7-6.0 TB/s)
// while the overlap-save kernel sits at 5.2: profiles/r02_direct_vs_fft_vs_mfma.txt decides.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fir_kernels.h"

namespace sdr_amd {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f2v lds_f2v; // an LDS-qualified pointer keeps a volatile read a ds_read

namespace {

// virtual stream: history | input | zeros (the same contract as fir_direct's staging)
template <int W>
struct MStream {
    const float* in;
    const float* hist;
    long long n;
    int hist_len;
    bool vec_ok;
    __device__ __forceinline__ float elem(long long m, int k) const {
        if (m < 0) {
            const long long h = hist_len + m;
            return h >= 0 ? hist[h * W + k] : 0.0f;
        }
        return m < n ? in[m * W + k] : 0.0f;
    }
    // 4 consecutive floats starting at float index f (f % 4 == 0) of the virtual stream
    __device__ __forceinline__ f4 load4(long long f) const {
        constexpr int NS = 4 / W;
        const long long m = f / W; // exact: f is a multiple of 4
        if (vec_ok) {
            if (m >= 0 && m + NS <= n) return __builtin_nontemporal_load(reinterpret_cast<const f4*>(in + f));
            if (m + NS <= 0 && m >= -(long long)hist_len && ((hist_len * W) & 3) == 0)
                return *reinterpret_cast<const f4*>(hist + (long long)hist_len * W + f);
        }
        f4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = elem(m + e / W, e % W);
        return v;
    }
};

} // namespace

// waves per SIMD asked of the register allocator: 8 workgroups per CU fit the LDS (18 KB each);
// the taps operand (SMAX registers) sets what the register file allows
constexpr int mfma_waves(int smax) { return smax <= 20 ? 8 : smax <= 28 ? 6 : 5; }

template <int P, int W, int SMAX>
__global__ __launch_bounds__(256, mfma_waves(SMAX)) void fir_mfma(MfmaArgs a) {
    constexpr int HB = 4 * (SMAX - 4);   // history samples per plane ahead of a tile (>= taps - 1)
    constexpr int OT = 4096 / P;         // outputs per plane per tile
    constexpr int G = OT / 1024;         // 256-output column groups per wave per plane
    constexpr int NIDX = OT + HB;        // staged samples per plane
    constexpr int PS = NIDX + 2 * (NIDX / 16) + 2; // padded samples in LDS (row stride 18 per 16 samples), x P floats
    constexpr int NC = (4096 + P * HB) / 4; // 16-byte chunks to stage
    extern __shared__ __align__(16) float lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long ch = blockIdx.y;
    const long long tile = blockIdx.x;

    MStream<W> src;
    src.in = a.in + ch * a.in_stride * W;
    src.hist = pick(a.hist, a.state ? a.state->cur & 1u : 0u) + ch * (long long)a.hist_len * W;
    src.n = a.n;
    src.hist_len = a.hist_len;
    src.vec_ok = a.vec_in;

    // ---- stage the tile and its left halo.  LDS keeps the planes interleaved as in memory:
    // sample idx of plane p sits at float P * (idx + 2 * (idx >> 4)) + p (row stride 18 samples per
    // 16), so the B operands of the P planes of one column group come out of ONE ds_read of 4 P
    // bytes per lane, conflict-free (lane address 18 j + kk samples: the 64 lanes of a b32 / b64
    // read, and the 16-lane groups of a b128 read, fall on different banks).
    const long long F0 = (long long)P * (tile * OT - HB); // first float of the staged range (stream float index)
    constexpr int ROUNDS = (NC + 255) / 256;
    f4 st[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int ci = tid + 256 * r;
        if (ci < NC) st[r] = src.load4(F0 + 4LL * ci);
    }
    // taps operand: one VGPR per step, the same for every tile
    float A[SMAX];
#pragma unroll
    for (int s = 0; s < SMAX; ++s) A[s] = a.atab[s * 64 + lane];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int ci = tid + 256 * r;
        if (ci < NC) {
            const int idx0 = 4 * ci / P;                          // first sample of the chunk (4 / P samples, one 16-group)
            float* dst = lds + P * (idx0 + 2 * (idx0 >> 4));
            if constexpr (P == 1) {                               // 8-byte aligned only
                *reinterpret_cast<f2v*>(dst) = f2v{st[r][0], st[r][1]};
                *reinterpret_cast<f2v*>(dst + 2) = f2v{st[r][2], st[r][3]};
            } else {
                *reinterpret_cast<f4*>(dst) = st[r];
            }
        }
    }
    __syncthreads();

    // ---- 4 accumulators per wave: chain c = (column group gq, plane p) ----------------------
    f4 acc[4];
    const int j = lane & 15, kk = lane >> 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = f4{0.0f, 0.0f, 0.0f, 0.0f};
    static_assert(G * P == 4, "four chains per wave");
    // lane's first B sample of column group gq: group base (multiple of 16) + 16 j + kk
    const float* bbase = lds + P * ((w * G) * 288 + 18 * j + kk); // 256 samples = 288 padded
#pragma unroll
    for (int s = 0; s < SMAX; ++s) {
        const int off = 4 * s + 2 * (s >> 2); // samples; compile-time per step
        if constexpr (P == 1) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                acc[gq] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s], bbase[gq * 288 + off], acc[gq], 0, 0, 0);
        } else if constexpr (P == 2) {
#pragma unroll
            for (int gq = 0; gq < 2; ++gq) {
                // volatile: keeps hipcc from pairing two of these into ds_read2_b64, which is banked
                // modulo 32 (2-way conflicts on this layout) and moves half the bytes per LDS cycle of
                // ds_read_b64 (MI355X_MICROARCH.md, LDS; r02_c1big_mfma_sq_counters.txt: 46 % conflicts)
                const f2v b = *(const volatile lds_f2v*)(bbase + 2 * (gq * 288 + off));
                acc[2 * gq] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s], b.x, acc[2 * gq], 0, 0, 0);
                acc[2 * gq + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s], b.y, acc[2 * gq + 1], 0, 0, 0);
            }
        } else {
            const f4 b = *reinterpret_cast<const f4*>(bbase + 4 * off);
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s], b[p], acc[p], 0, 0, 0);
        }
    }

    // ---- stores.  A lane holds outputs 16 j + 4 kk + r (r = 0..3) of every chain, i.e. with P
    // planes 4 floats out of every 4 P of the stream: stored from the registers, an instruction
    // would fill only 1/P of every cache line it touches.  The accumulators go back through LDS in
    // stream order instead (each wave's 1024 floats are contiguous: a wave-private region once
    // every wave has finished reading the staged samples), and come out as 16 contiguous bytes
    // per lane, 1 KiB per instruction.  Row padding: 4 floats per 64 (keeps 16-byte alignment).
    __syncthreads();
    static_assert(P * PS >= 4096 + 4 * 64, "the staged planes are large enough to hold the tile's outputs");
#pragma unroll
    for (int gq = 0; gq < G; ++gq) {
        const int f0 = P * ((w * G + gq) * 256 + 16 * j + 4 * kk); // float index inside the tile, multiple of 4 P
        if constexpr (P == 1) {
            *reinterpret_cast<f4*>(&lds[f0 + 4 * (f0 >> 6)]) = acc[gq];
        } else if constexpr (P == 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {   // floats f0 + 4 h .. + 3 = samples 2 h, 2 h + 1 of both planes
                const int f = f0 + 4 * h;
                *reinterpret_cast<f4*>(&lds[f + 4 * (f >> 6)]) =
                    f4{acc[2 * gq][2 * h], acc[2 * gq + 1][2 * h], acc[2 * gq][2 * h + 1], acc[2 * gq + 1][2 * h + 1]};
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = f0 + 4 * r;
                *reinterpret_cast<f4*>(&lds[f + 4 * (f >> 6)]) = f4{acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
            }
        }
    }
    // the accumulators were written by one lane and are read back by others of the same wave: the
    // fences order that hand-off by construction (wave_barrier alone only pins the schedule)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float* out = a.out + ch * a.out_stride * W;
    const long long nfl = a.n * W; // floats per channel
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int fi = 1024 * w + 256 * q + 4 * lane;
        const f4 v = *reinterpret_cast<const f4*>(&lds[fi + 4 * (fi >> 6)]);
        const long long f = tile * 4096 + fi;
        if (a.vec_out && f + 4 <= nfl) {
            __builtin_nontemporal_store(v, reinterpret_cast<f4*>(out + f));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (f + e < nfl) out[f + e] = v[e];
        }
    }
}

// ---- launch table ----------------------------------------------------------------------------
int mfma_steps_for(size_t dense_taps) {
    for (int smax : {8, 12, 16, 20, 24, 28, 32, 36})
        if (dense_taps <= (size_t)(4 * smax - 15)) return smax;
    return 0;
}

size_t mfma_hist_samples(int smax, int stride) { return (size_t)4 * (smax - 4) * stride; }

// atab[s * 64 + lane] = g[(lane & 15) + HB - (4 s + (lane >> 4))], zero outside [0, ntaps)
void mfma_fill_atab(const float* g, size_t ntaps, int smax, float* atab) {
    const int HB = 4 * (smax - 4);
    for (int s = 0; s < smax; ++s)
        for (int lane = 0; lane < 64; ++lane) {
            const int t = (lane & 15) + HB - (4 * s + (lane >> 4));
            atab[s * 64 + lane] = (t >= 0 && (size_t)t < ntaps) ? g[t] : 0.0f;
        }
}

template <int P, int W>
static hipError_t launch_pw(const MfmaArgs& a, int smax, dim3 grid, hipStream_t stream) {
    const int HB = 4 * (smax - 4), NIDX = 4096 / P + HB;
    const size_t lds = (size_t)P * (NIDX + 2 * (NIDX / 16) + 2) * sizeof(float);
    switch (smax) {
        case 8: hipLaunchKernelGGL((fir_mfma<P, W, 8>), grid, dim3(256), lds, stream, a); break;
        case 12: hipLaunchKernelGGL((fir_mfma<P, W, 12>), grid, dim3(256), lds, stream, a); break;
        case 16: hipLaunchKernelGGL((fir_mfma<P, W, 16>), grid, dim3(256), lds, stream, a); break;
        case 20: hipLaunchKernelGGL((fir_mfma<P, W, 20>), grid, dim3(256), lds, stream, a); break;
        case 24: hipLaunchKernelGGL((fir_mfma<P, W, 24>), grid, dim3(256), lds, stream, a); break;
        case 28: hipLaunchKernelGGL((fir_mfma<P, W, 28>), grid, dim3(256), lds, stream, a); break;
        case 32: hipLaunchKernelGGL((fir_mfma<P, W, 32>), grid, dim3(256), lds, stream, a); break;
        case 36: hipLaunchKernelGGL((fir_mfma<P, W, 36>), grid, dim3(256), lds, stream, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_fir_mfma(const MfmaArgs& a, bool real, int stride, int smax, size_t channels, hipStream_t stream) {
    if (a.n == 0 || channels == 0) return hipSuccess;
    const int W = real ? 1 : 2;
    const long long nfl = a.n * W;
    const long long tiles = (nfl + 4095) / 4096;
    if (tiles > 0x7fffffffLL || channels > 65535) return hipErrorInvalidValue;
    const dim3 grid((unsigned)tiles, (unsigned)channels, 1);
    if (real && stride == 1) return launch_pw<1, 1>(a, smax, grid, stream);
    if (real && stride == 2) return launch_pw<2, 1>(a, smax, grid, stream);
    if (!real && stride == 1) return launch_pw<2, 2>(a, smax, grid, stream);
    if (!real && stride == 2) return launch_pw<4, 2>(a, smax, grid, stream);
    return hipErrorInvalidValue;
}

const char* fir_mfma_kernel_name(bool real, int stride) {
    if (real) return stride == 2 ? "fir_mfma<REAL,STRIDE=2>" : "fir_mfma<REAL>";
    return stride == 2 ? "fir_mfma<COMPLEX,STRIDE=2>" : "fir_mfma<COMPLEX>";
}

} // namespace sdr_amd
