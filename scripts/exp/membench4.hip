// membench4.hip -- does the ORDER in which a persistent kernel walks its tiles matter?
// 2 GiB -> 2 GiB copy in 32 KB tiles (the FFT kernel's shape: 16 loads of 8 B per lane,
// prefetched one tile ahead): grid-strided tiles vs tiles handed out by an atomic counter
// (the active tiles stay a dense window however the workgroups drift), 3 or 6 blocks/CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// PADKB: extra LDS per workgroup, to set how many workgroups fit a CU (160 KB)
template <bool DYN, int SPIN, int XCH = 0, int PADKB = 0, int SPLIT = 0>
__global__ __launch_bounds__(256, 3) void copy_tiles(const f2* __restrict__ in, f2* __restrict__ out, long long tiles,
                                                     unsigned* counter) {
    __shared__ long long next_tile;
    __shared__ f2 xch[XCH ? 16 * 272 : 1];
    __shared__ float pad[PADKB ? PADKB * 256 : 1];
    if (PADKB && in == nullptr) pad[threadIdx.x] = 1.0f; // keep the array
    const int tid = threadIdx.x;
    auto grab = [&]() -> long long {
        if (!DYN) return -1;
        __syncthreads();
        if (tid == 0) next_tile = atomicAdd(counter, 1u);
        __syncthreads();
        return next_tile;
    };
    long long t = DYN ? grab() : blockIdx.x;
    if (t >= tiles) return;
    f2 nx[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) nx[k] = __builtin_nontemporal_load(&in[t * 3840 + 256 * k + tid]);
    while (true) {
        f2 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = nx[k];
        long long tn = DYN ? grab() : t + gridDim.x;
        const bool more = tn < tiles;
        const long long tl = more ? tn : t;
        if (!SPLIT) {
#pragma unroll
            for (int k = 0; k < 16; ++k) nx[k] = __builtin_nontemporal_load(&in[tl * 3840 + 256 * k + tid]);
        }
        // stand-in for the transform: SPIN dependent packed ops per point, XCH exchanges
        // of the whole tile through LDS with a block barrier each
#pragma unroll
        for (int k = 0; k < 16; ++k)
            for (int s = 0; s < SPIN; ++s) v[k] = v[k] * 1.0001f + 0.0001f;
        for (int x = 0; x < XCH; ++x) {
            // XCH < 10: block-wide exchanges (two block barriers each).  XCH = 12: the real
            // kernel's pattern -- exchanges 0 and 5 block-wide with ONE barrier each (columns
            // are private to a wave on one side), exchanges 1-4 inside 16-lane groups.
            // XCH = 11: 2 block-wide + 2 group exchanges (the kernel's actual count)
            const int last = XCH == 11 ? 3 : 5;
            if (SPLIT && x < 4) { // a quarter of the next tile's loads before each of the first four exchanges
#pragma unroll
                for (int k = 4 * x; k < 4 * x + 4; ++k) nx[k] = __builtin_nontemporal_load(&in[tl * 3840 + 256 * k + tid]);
            }
            const bool block_wide = XCH < 10 || x == 0 || x == last;
            if (XCH >= 10 && x > last) break;
            if (XCH < 10) __syncthreads();
            if (block_wide) {
#pragma unroll
                for (int k = 0; k < 16; ++k) xch[k * 272 + tid] = v[k];
                __syncthreads();
#pragma unroll
                for (int k = 0; k < 16; ++k) v[k] = xch[(tid >> 4) * 272 + 16 * k + (tid & 15)] * 1.0001f;
            } else {
                f2* row = xch + (tid >> 4) * 272;
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < 16; ++k) row[k * 17 + (tid & 15)] = v[k];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < 16; ++k) v[k] = row[(tid & 15) * 17 + k] * 1.0001f;
            }
        }
#pragma unroll
        for (int k = 0; k < 15; ++k) __builtin_nontemporal_store(v[k + 1], &out[t * 3840 + 256 * k + tid]);
        if (!more) break;
        t = tn;
    }
}

template <class F> float timeit(F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) f();
    CK(hipEventRecord(a));
    for (int i = 0; i < 20; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / 20;
}

int main() {
    const long long n = 1LL << 28, tiles = n / 3840;   // like C2: 4096 in, 3840 out per tile
    f2 *in, *out; unsigned* ctr;
    CK(hipMalloc(&in, (n + 8192) * 8)); CK(hipMalloc(&out, (n + 8192) * 8)); CK(hipMalloc(&ctr, 4));
    CK(hipMemset(in, 1, (n + 8192) * 8));
    const double gb = 2.0 * n * 8 / 1e9;
    {
        const int grid = 256 * 8;
        float t;
#define RUN(SPIN, XCH, PAD, WHAT) t = timeit([&] { hipLaunchKernelGGL((copy_tiles<false, SPIN, XCH, PAD>), dim3(grid), dim3(256), 0, 0, in, out, tiles, ctr); }); \
        printf("%s: %4d packed ops/lane/tile, %d LDS exchanges: %.3f ms %.0f GB/s\n", WHAT, 16 * SPIN, XCH, t, gb / t * 1e3);
        RUN(40, 11, 17, "3 workgroups/CU, 2 block-wide + 2 group exchanges (the kernel's count)")
        t = timeit([&] { hipLaunchKernelGGL((copy_tiles<false, 40, 11, 17, 1>), dim3(grid), dim3(256), 0, 0, in, out, tiles, ctr); });
        printf("same, next tile's loads spread over the four exchanges: %.3f ms %.0f GB/s\n", t, gb / t * 1e3);
        RUN(43, 11, 17, "3 workgroups/CU, same with 690 ops")
        RUN(40, 12, 17, "3 workgroups/CU, real pattern (2 block-wide + 4 group exchanges)")
        RUN(40, 12, 4, "4 workgroups/CU, real pattern")
        RUN(0, 12, 17, "3 workgroups/CU, real pattern, no ALU work")
        RUN(40, 0, 17, "3 workgroups/CU, ALU work only")
        RUN(40, 6, 17, "3 workgroups/CU, 6 block-wide exchanges")
        RUN(0, 0, 17, "3 workgroups/CU, copy only")
    }
    return 0;
}
