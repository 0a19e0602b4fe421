// membench5.hip -- why does a one-element-per-thread copy reach 6.1 TB/s on this device when
// every persistent / multi-load copy stops at 5.2-5.5?  2 GiB -> 2 GiB copy (2^28 f2 elements),
// workgroups of 256 lanes, a workgroup moves tiles of 256*LOADS elements (LOADS loads of 8 B
// per lane, all issued before the first store -- the FFT kernel's shape is LOADS = 16).
//   PERSIST = false: one tile per workgroup, grid = number of tiles (hardware dispatch order)
//   PERSIST = true : grid-strided loop over the tiles
// Dynamic LDS sets how many workgroups a CU holds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int LOADS, bool PERSIST>
__global__ __launch_bounds__(256) void copy_tiles(const f2* __restrict__ in, f2* __restrict__ out, long long tiles) {
    extern __shared__ float pad[];
    if (in == nullptr) pad[threadIdx.x] = 1.0f;
    const int tid = threadIdx.x;
    for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
        const long long base = t * (256LL * LOADS); // < tiles * 256 * LOADS = n exactly
        f2 v[LOADS];
#pragma unroll
        for (int k = 0; k < LOADS; ++k) v[k] = __builtin_nontemporal_load(&in[base + 256 * k + tid]);
#pragma unroll
        for (int k = 0; k < LOADS; ++k) __builtin_nontemporal_store(v[k], &out[base + 256 * k + tid]);
        if (!PERSIST) break;
    }
}

// overlap-save shape: tiles of 4096 in, the last 4096 - HALO out, tile stride 4096 - HALO
// (both streams lose their 32 KB alignment); PF: next tile's loads issued before the stores
// IS / OS: input / output tile stride in elements (0 = 4096 - HALO, the overlap-save stride)
// XM: workgroup b takes tile slot (b % 8) * (grid / 8) + b / 8 of each grid-sized batch, so
// that the workgroups of one XCD (hardware round-robin: b % 8) cover a contiguous range
template <int HALO, bool PF, int IS = 0, int OS = 0, bool XM = false>
__global__ __launch_bounds__(256) void copy_halo(const f2* __restrict__ in, f2* __restrict__ out, long long tiles) {
    extern __shared__ float pad[];
    if (in == nullptr) pad[threadIdx.x] = 1.0f;
    const int tid = threadIdx.x;
    constexpr long long L = IS ? IS : 4096 - HALO;
    constexpr long long LO = OS ? OS : 4096 - HALO;
    long long t = XM ? (long long)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : blockIdx.x;
    if (t >= tiles) return;
    f2 nx[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) nx[k] = __builtin_nontemporal_load(&in[t * L + 256 * k + tid]);
    for (; t < tiles; t += gridDim.x) {
        f2 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = nx[k];
        const long long tn = t + gridDim.x < tiles ? t + gridDim.x : t;
        if (PF) {
#pragma unroll
            for (int k = 0; k < 16; ++k) nx[k] = __builtin_nontemporal_load(&in[tn * L + 256 * k + tid]);
        }
#pragma unroll
        for (int k = HALO / 256; k < 16; ++k) {
            if (k == HALO / 256 && HALO % 256 != 0 && tid < HALO % 256) continue;
            __builtin_nontemporal_store(v[k], &out[t * LO + 256 * k + tid]);
        }
        if (!PF) {
#pragma unroll
            for (int k = 0; k < 16; ++k) nx[k] = __builtin_nontemporal_load(&in[tn * L + 256 * k + tid]);
        }
    }
}

template <int HALO, bool PF, int IS = 0, int OS = 0, bool XM = false>
static void run_halo(const f2* in, f2* out, long long n, int wg_per_cu, int grid_per_cu) {
    constexpr long long L = 4096 - HALO;
    constexpr long long SMAX = (IS > OS ? IS : OS) > L ? (IS > OS ? IS : OS) : L;
    const long long tiles = (n - 4096) / SMAX + 1; // last tile ends at (tiles-1)*stride + 4096 <= n for both streams
    if ((tiles - 1) * SMAX + 4096 > n) { printf("bad tiling\n"); exit(1); }
    const size_t lds = (size_t)(160 * 1024 / wg_per_cu) & ~(size_t)255;
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)copy_halo<HALO, PF, IS, OS, XM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned grid = 256u * grid_per_cu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((copy_halo<HALO, PF, IS, OS, XM>), dim3(grid), dim3(256), lds, 0, in, out, tiles);
    CK(hipDeviceSynchronize());
    const int reps = 30;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((copy_halo<HALO, PF, IS, OS, XM>), dim3(grid), dim3(256), lds, 0, in, out, tiles);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    if (XM) printf("XCD-contiguous ");
    printf("overlap-save shape, halo %4d, in stride %4d, out stride %4d, %s, wg/CU %d, grid %d/CU: %.3f ms %5.0f GB/s (outputs x 16 B)\n", HALO,
           (int)(IS ? IS : L), (int)(OS ? OS : L), PF ? "prefetch" : "loads after stores", wg_per_cu, grid_per_cu, ms,
           2.0 * tiles * L * sizeof(f2) / ms / 1e6);
    fflush(stdout);
}

template <int LOADS, bool PERSIST>
static void run(const char* what, const f2* in, f2* out, long long n, int wg_per_cu, int grid_per_cu) {
    const long long tiles = n / (256LL * LOADS);
    if (tiles * 256LL * LOADS != n) { printf("size not a multiple of the tile\n"); exit(1); }
    // LDS per workgroup so that exactly wg_per_cu fit in 160 KB (0 = no limit from LDS)
    const size_t lds = wg_per_cu ? (size_t)(160 * 1024 / wg_per_cu) & ~(size_t)255 : 0;
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)copy_tiles<LOADS, PERSIST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long long grid = PERSIST ? 256LL * grid_per_cu : tiles;
    if (grid > 0x7fffffffLL) { printf("grid too large\n"); exit(1); }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((copy_tiles<LOADS, PERSIST>), dim3((unsigned)grid), dim3(256), lds, 0, in, out, tiles);
    CK(hipDeviceSynchronize());
    const int reps = 30;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((copy_tiles<LOADS, PERSIST>), dim3((unsigned)grid), dim3(256), lds, 0, in, out, tiles);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-58s %2d loads/lane, %s wg/CU %2d: %.3f ms %5.0f GB/s\n", what, LOADS, PERSIST ? "persistent    " : "one tile per wg", wg_per_cu,
           ms, 2.0 * n * sizeof(f2) / ms / 1e6);
    fflush(stdout);
}

int main() {
    const long long n = 1LL << 28;
    f2 *in, *out;
    CK(hipMalloc(&in, n * sizeof(f2)));
    CK(hipMalloc(&out, n * sizeof(f2)));
    CK(hipMemset(in, 1, n * sizeof(f2)));
    CK(hipMemset(out, 0, n * sizeof(f2)));
    run<1, false>("8 B per thread, nothing else", in, out, n, 0, 0);
    run<2, false>("", in, out, n, 0, 0);
    run<4, false>("", in, out, n, 0, 0);
    run<16, false>("FFT tile shape, dispatch order, residency unlimited", in, out, n, 0, 0);
    run<16, false>("FFT tile shape, dispatch order", in, out, n, 8, 0);
    run<16, false>("FFT tile shape, dispatch order", in, out, n, 4, 0);
    run<16, false>("FFT tile shape, dispatch order", in, out, n, 3, 0);
    run<16, false>("FFT tile shape, dispatch order", in, out, n, 2, 0);
    run<16, true>("FFT tile shape, grid-stride, grid 3/CU", in, out, n, 3, 3);
    run<16, true>("FFT tile shape, grid-stride, grid 6/CU", in, out, n, 3, 6);
    run<16, true>("FFT tile shape, grid-stride, grid 8/CU", in, out, n, 8, 8);
    run<4, false>("8 KB tiles, dispatch order", in, out, n, 8, 0);
    run<4, false>("8 KB tiles, dispatch order", in, out, n, 3, 0);
    run<4, true>("8 KB tiles, grid-stride, grid 8/CU", in, out, n, 8, 8);
    run<1, true>("2 KB tiles, grid-stride, grid 8/CU", in, out, n, 8, 8);
    run<1, true>("2 KB tiles, grid-stride, grid 32/CU", in, out, n, 0, 32);
    run_halo<0, true>(in, out, n, 3, 6);
    run_halo<0, false>(in, out, n, 3, 6);
    run_halo<128, true>(in, out, n, 3, 6);
    run_halo<128, false>(in, out, n, 3, 6);
    run_halo<256, true>(in, out, n, 3, 6);
    run_halo<512, true>(in, out, n, 3, 6);
    run_halo<2048, true>(in, out, n, 3, 6);
    run_halo<128, true, 0, 4096>(in, out, n, 3, 6);     // reads as overlap-save, writes 32 KB-aligned
    run_halo<128, true, 4096, 0>(in, out, n, 3, 6);     // reads 32 KB-aligned, writes as overlap-save
    run_halo<128, true, 4096, 4096>(in, out, n, 3, 6);  // both aligned, 31 KB written per tile
    run_halo<128, true, 0, 0, true>(in, out, n, 3, 6);
    run_halo<128, true, 0, 0, true>(in, out, n, 3, 3);
    run_halo<256, true, 0, 0, true>(in, out, n, 3, 6);
    run_halo<0, true, 0, 0, true>(in, out, n, 3, 6);
    run_halo<128, true>(in, out, n, 3, 6);
    run_halo<128, true>(in, out, n, 4, 8);
    run_halo<128, true>(in, out, n, 2, 4);
    CK(hipFree(in));
    CK(hipFree(out));
    return 0;
}
