// fir_kernels.hip -- direct-form streaming FIR for gfx950 (MI355X), hand-written HIP.
//
// Replaces the inner loop of the reference's kfr::fir (libs/kfr/dsp/fir.hpp:129-146:
// per input sample, a ring-buffer write and two strided dot products) with a
// tiled kernel:  y[i] = sum_j h[j] x[i-j].
//
// Work decomposition (one workgroup = 256 lanes = 4 wave64):
//   * a tile is 256*R consecutive complex outputs; lane l owns the R consecutive
//     outputs of "segment" l.  REAL signals are processed as pseudo-complex
//     z[m] = x[t0+m] + i*x[t0+S+m] (S = 256*R): a real-tap FIR is linear, so the
//     real and imaginary lanes of v_pk_fma_f32 carry two independent halves of a
//     2*S real tile and both sample kinds share one inner loop.
//   * the input tile plus its left halo (NB segments >= ntaps-1 samples) is staged
//     global -> registers -> LDS with 16-byte coalesced loads.  Each R-sample
//     segment is padded by 16 bytes in LDS so that the per-lane ds_read_b128 of a
//     whole segment (lane stride 8R+16 bytes) is bank-conflict free.
//   * taps are wave-uniform: read through the scalar cache (s_load) and applied
//     from SGPRs; the sample window lives in VGPRs and slides by whole segments:
//     tap chunk c needs segments (l-c) and (l-c-1), held in two register sets that
//     swap roles each chunk (no register moves).  Each LDS word is read once per
//     R multiply-adds.
//   * STRIDE=2 kernels skip every other tap (the Hilbert transformer's zero
//     taps, include/sdr/hilbert.hpp:46-55 of the reference): half the FMAs.
//   * results go back through LDS so the global stores are 16 B/lane coalesced.
//
// Algorithmic HBM traffic: 8 B/sample real (4 in + 4 out), 16 B/sample complex;
// halo re-reads add NB/256 (6 % at 127 taps, R=8) and normally hit L2.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fir_kernels.h"

namespace sdr_amd {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// One virtual stream per channel: index m < 0 reads the carried history
// (hist[hist_len + m]), 0 <= m < n reads the input, m >= n reads zero.
template <int W> // floats per sample
struct Stream {
    const float* in;
    const float* hist;
    long long n;
    int hist_len;
    bool vec_ok; // in and hist 16-byte aligned

    __device__ __forceinline__ void elem(long long m, float* dst) const {
        if (m < 0) {
            long long h = hist_len + m;
#pragma unroll
            for (int k = 0; k < W; ++k) dst[k] = h >= 0 ? hist[h * W + k] : 0.0f;
        } else if (m < n) {
#pragma unroll
            for (int k = 0; k < W; ++k) dst[k] = in[m * W + k];
        } else {
#pragma unroll
            for (int k = 0; k < W; ++k) dst[k] = 0.0f;
        }
    }
    // 16 bytes = 4/W consecutive samples starting at virtual index m
    __device__ __forceinline__ f4 load16(long long m) const {
        constexpr int NS = 4 / W;
        if (vec_ok && ((m * W) & 3) == 0) {
            if (m >= 0 && m + NS <= n) return *reinterpret_cast<const f4*>(in + m * W);
            if (m + NS <= 0 && m >= -(long long)hist_len && ((hist_len * W) & 3) == 0)
                return *reinterpret_cast<const f4*>(hist + (hist_len + m) * W);
        }
        float t[4];
#pragma unroll
        for (int s = 0; s < NS; ++s) elem(m + s, t + s * W);
        return f4{t[0], t[1], t[2], t[3]};
    }
};

template <int R, int STRIDE>
__device__ __forceinline__ void tap_chunk(f2 (&acc)[R], const f2 (&hi)[R], const f2 (&lo)[R],
                                          const float* __restrict__ taps) {
#pragma unroll
    for (int kk = 0; kk < R / STRIDE; ++kk) {
        const float h = taps[kk];
        const f2 hh = {h, h};
        const int jj = kk * STRIDE;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int e = r - jj;
            const f2 x = e >= 0 ? hi[e] : lo[R + e];
            acc[r] = __builtin_elementwise_fma(hh, x, acc[r]);
        }
    }
}

template <int R>
__device__ __forceinline__ void load_segment(f2 (&dst)[R], const float* p) {
#pragma unroll
    for (int q = 0; q < R / 2; ++q) {
        f4 v = *reinterpret_cast<const f4*>(p + 4 * q);
        dst[2 * q] = f2{v.x, v.y};
        dst[2 * q + 1] = f2{v.z, v.w};
    }
}

template <int R, int STRIDE, bool REAL>
__global__ __launch_bounds__(256) void fir_direct(FirArgs a) {
    constexpr int SEGF = 2 * R;    // floats per segment (R pseudo-complex samples)
    constexpr int SEGS = SEGF + 4; // padded segment stride in floats (+16 B)
    constexpr long long S = 256LL * R;
    extern __shared__ __align__(16) float lds[];

    const int tid = threadIdx.x;
    const int NB = a.nchunks;
    const int halo = NB * R;
    const long long ch = blockIdx.y;
    const long long t0 = (long long)blockIdx.x * (REAL ? 2 * S : S) - a.shift;

    Stream<REAL ? 1 : 2> src;
    src.in = a.in + ch * a.in_stride * (REAL ? 1 : 2);
    src.hist = pick(a.hist, a.state ? a.state->cur & 1u : 0u) + ch * (long long)a.hist_len * (REAL ? 1 : 2);
    src.n = a.n;
    src.hist_len = a.hist_len;
    src.vec_ok = a.vec_in;

    // ---- stage the tile + halo into LDS ------------------------------------
    const int nseg = 256 + NB;
    if (!REAL) {
        const int nchunk = nseg * (R / 2); // 16-byte chunks (2 complex each)
        for (int ci = tid; ci < nchunk; ci += 256) {
            const f4 v = src.load16(t0 - halo + 2LL * ci);
            const int fo = 4 * ci;
            *reinterpret_cast<f4*>(&lds[fo + (fo / SEGF) * 4]) = v;
        }
    } else {
        const int nquad = nseg * (R / 4); // 4 pseudo-complex samples each
        for (int q = tid; q < nquad; q += 256) {
            const long long m = t0 - halo + 4LL * q;
            const f4 A = src.load16(m);
            const f4 B = src.load16(m + S);
            const int fo = 8 * q;
            float* p = &lds[fo + (fo / SEGF) * 4];
            *reinterpret_cast<f4*>(p) = f4{A.x, B.x, A.y, B.y};
            *reinterpret_cast<f4*>(p + 4) = f4{A.z, B.z, A.w, B.w};
        }
    }
    __syncthreads();

    // ---- sliding-window multiply-accumulate ---------------------------------
    // Blocked accumulation: partial sums of at most 16 chunks (16*R tap offsets)
    // are folded into `acc`, so the fp32 rounding error grows with
    // sqrt(ntaps/128) instead of sqrt(ntaps) (matters at 4095 taps).
    f2 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = f2{0.0f, 0.0f};
    f2 P[R], Q[R];
    const float* seg = lds + (tid + NB) * SEGS; // segment holding this lane's outputs
    const float* __restrict__ taps = a.taps;
    load_segment<R>(P, seg);
    for (int c0 = 0; c0 < NB; c0 += 16) {
        const int c1 = c0 + 16 < NB ? c0 + 16 : NB;
        f2 part[R];
#pragma unroll
        for (int r = 0; r < R; ++r) part[r] = f2{0.0f, 0.0f};
        for (int c = c0; c < c1; c += 2) {
            load_segment<R>(Q, seg - (c + 1) * SEGS);
            tap_chunk<R, STRIDE>(part, P, Q, taps + c * (R / STRIDE));
            load_segment<R>(P, seg - (c + 2) * SEGS);
            tap_chunk<R, STRIDE>(part, Q, P, taps + (c + 1) * (R / STRIDE));
        }
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] += part[r];
    }
    __syncthreads();

    // ---- transpose through LDS, coalesced stores ----------------------------
    {
        float* p = lds + tid * SEGS;
#pragma unroll
        for (int q = 0; q < R / 2; ++q)
            *reinterpret_cast<f4*>(p + 4 * q) = f4{acc[2 * q].x, acc[2 * q].y, acc[2 * q + 1].x, acc[2 * q + 1].y};
    }
    __syncthreads();

    const long long o0 = (long long)blockIdx.x * (REAL ? 2 * S : S);
    if (!REAL) {
        float* out = a.out + ch * a.out_stride * 2;
#pragma unroll
        for (int it = 0; it < R / 2; ++it) {
            const int ci = tid + it * 256;
            const int fo = 4 * ci;
            const f4 v = *reinterpret_cast<const f4*>(&lds[fo + (fo / SEGF) * 4]);
            const long long o = o0 + 2LL * ci;
            if (o + 2 <= a.n && a.vec_out) {
                *reinterpret_cast<f4*>(out + 2 * o) = v;
            } else {
                if (o < a.n) { out[2 * o] = v.x; out[2 * o + 1] = v.y; }
                if (o + 1 < a.n) { out[2 * o + 2] = v.z; out[2 * o + 3] = v.w; }
            }
        }
    } else {
        float* out = a.out + ch * a.out_stride;
#pragma unroll
        for (int it = 0; it < R / 4; ++it) {
            const int q = tid + it * 256;
            const int fo = 8 * q;
            const float* p = &lds[fo + (fo / SEGF) * 4];
            const f4 u = *reinterpret_cast<const f4*>(p);
            const f4 w = *reinterpret_cast<const f4*>(p + 4);
            const f4 A = {u.x, u.z, w.x, w.z};
            const f4 B = {u.y, u.w, w.y, w.w};
            const long long oa = o0 + 4LL * q, ob = oa + S;
            if (oa + 4 <= a.n && a.vec_out) {
                *reinterpret_cast<f4*>(out + oa) = A;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) if (oa + k < a.n) out[oa + k] = A[k];
            }
            if (ob + 4 <= a.n && a.vec_out) {
                *reinterpret_cast<f4*>(out + ob) = B;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) if (ob + k < a.n) out[ob + k] = B[k];
            }
        }
    }
}

// new_hist[i] = virtual x[n - hist_len + i]; reads the OLD history (the buffer state->cur names) for
// the part of the window that precedes this call's input (n < hist_len), writes the other buffer.
// This is the last launch of a direct-form / matrix-core call: its last workgroup to finish flips
// state->cur, so the next launch on the stream -- a new call or a replay of this one -- reads on.
__global__ void fir_carry_history(const float* __restrict__ in, long long in_stride, long long n,
                                  float* hist0, float* hist1, FirDevState* state, int hist_len, int w) {
    const long long ch = blockIdx.y;
    const int total = hist_len * w;
    const unsigned cur = state->cur & 1u;
    const float* __restrict__ old_hist = cur ? hist1 : hist0;
    float* __restrict__ new_hist = cur ? hist0 : hist1;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int s = i / w, k = i - s * w;
        const long long m = n - hist_len + s;
        float v;
        if (m >= 0) v = in[(ch * in_stride + m) * w + k];
        else v = old_hist[(ch * hist_len + (hist_len + m)) * w + k];
        new_hist[ch * (long long)total + i] = v;
    }
    __syncthreads(); // every wave of the workgroup has read state->cur
    if (last_workgroup_done(state, gridDim.x * gridDim.y)) {
        state->done = 0;
        state->cur = cur ^ 1u;
    }
}

// ---- launch table ------------------------------------------------------------
typedef void (*fir_fn)(FirArgs);

template <int R, int STRIDE, bool REAL>
static KernelInfo make_info(const char* name) {
    KernelInfo k;
    k.fn = reinterpret_cast<const void*>(&fir_direct<R, STRIDE, REAL>);
    k.name = name;
    k.R = R;
    k.stride = STRIDE;
    k.real = REAL;
    return k;
}

#define SDR_K(R, S, RE) make_info<R, S, RE>("fir_direct<R=" #R ",STRIDE=" #S "," #RE ">")

static const KernelInfo g_kernels[] = {
    SDR_K(4, 1, false), SDR_K(8, 1, false), SDR_K(16, 1, false),
    SDR_K(4, 2, false), SDR_K(8, 2, false), SDR_K(16, 2, false),
    SDR_K(4, 1, true),  SDR_K(8, 1, true),  SDR_K(16, 1, true),
    SDR_K(4, 2, true),  SDR_K(8, 2, true),  SDR_K(16, 2, true),
};

const KernelInfo* find_kernel(int R, int stride, bool real) {
    for (const KernelInfo& k : g_kernels)
        if (k.R == R && k.stride == stride && k.real == real) return &k;
    return nullptr;
}

size_t kernel_lds_bytes(const KernelInfo& k, int nchunks) {
    return (size_t)(256 + nchunks) * (2 * k.R + 4) * sizeof(float);
}

hipError_t launch_fir(const KernelInfo& k, const FirArgs& a, size_t channels, hipStream_t stream) {
    const long long per_tile = 256LL * k.R * (k.real ? 2 : 1);
    const long long tiles = (a.n + per_tile - 1) / per_tile;
    if (tiles == 0 || channels == 0) return hipSuccess;
    if (tiles > 0x7fffffffLL || channels > 65535) return hipErrorInvalidValue;
    const size_t lds = kernel_lds_bytes(k, a.nchunks);
    FirArgs args = a;
    void* params[] = {&args};
    return hipLaunchKernel(k.fn, dim3((unsigned)tiles, (unsigned)channels, 1), dim3(256, 1, 1), params, lds, stream);
}

hipError_t prepare_kernel(const KernelInfo& k, int nchunks) {
    const size_t lds = kernel_lds_bytes(k, nchunks);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024)
        return hipFuncSetAttribute(k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return hipSuccess;
}

hipError_t launch_carry_history(const float* in, long long in_stride, long long n, float* const hist[2],
                                FirDevState* state, int hist_len, int w, size_t channels, hipStream_t stream) {
    if (channels == 0) return hipSuccess;
    const int total = hist_len * w;
    const int blocks = (total + 255) / 256;
    hipLaunchKernelGGL(fir_carry_history, dim3(blocks > 0 ? blocks : 1, (unsigned)channels, 1), dim3(256), 0, stream,
                       in, in_stride, n, hist[0], hist[1], state, hist_len, w);
    return hipGetLastError();
}

} // namespace sdr_amd
