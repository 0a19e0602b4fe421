// membench7.hip -- round 2: which property of the access shape does the memory system price?
// 2 GiB -> 2 GiB copies, ONE tile per workgroup in hardware dispatch order (no persistence), all
// loads of a tile issued before its first store, non-temporal.  Variables:
//   LANES  lanes per workgroup (256 / 512 / 1024)       -> width of a workgroup-wide row
//   ROWS   loads per lane
//   VEC    floats per lane per access (2 = 8 B, 4 = 16 B)
//   LW/SW  loads / stores wave-private: a wave covers a contiguous 1/WAVES share of the tile with
//          ROWS narrow rows of 64 lanes, instead of the workgroup covering it with ROWS wide rows
//   ORDER  tile = blockIdx (compact window) or a 64-way interleave over the buffer (spread window)
//   occupancy via dynamic LDS
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int VEC> struct V;
template <> struct V<2> { typedef float T __attribute__((ext_vector_type(2))); };
template <> struct V<4> { typedef float T __attribute__((ext_vector_type(4))); };

template <int LANES, int ROWS, int VEC, bool LW, bool SW, bool SPREAD>
__global__ __launch_bounds__(LANES) void copy1(const float* __restrict__ in, float* __restrict__ out, long long tiles) {
    typedef typename V<VEC>::T T;
    extern __shared__ float pad[];
    if (in == nullptr) pad[threadIdx.x] = 1.0f;
    long long t = blockIdx.x;
    if (SPREAD) t = (t % 64) * (tiles / 64) + t / 64; // tiles is a multiple of 64
    const T* src = reinterpret_cast<const T*>(in) + t * (long long)(LANES * ROWS);
    T* dst = reinterpret_cast<T*>(out) + t * (long long)(LANES * ROWS);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    T v[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; ++k) v[k] = __builtin_nontemporal_load(&src[LW ? w * (64 * ROWS) + 64 * k + lane : LANES * k + tid]);
    if (LW != SW) {
        // the two mappings hold different elements per lane: a real kernel would exchange through LDS; the
        // copy just stores what it has at the other mapping's addresses (same traffic, permuted data)
    }
#pragma unroll
    for (int k = 0; k < ROWS; ++k) __builtin_nontemporal_store(v[k], &dst[SW ? w * (64 * ROWS) + 64 * k + lane : LANES * k + tid]);
}

template <int LANES, int ROWS, int VEC, bool LW, bool SW, bool SPREAD>
static void run(const float* in, float* out, long long nfloat, int wg_per_cu) {
    const long long per_tile = (long long)LANES * ROWS * VEC;
    const long long tiles = nfloat / per_tile;
    const size_t lds = wg_per_cu ? ((size_t)(160 * 1024 / wg_per_cu)) & ~(size_t)255 : 0;
    auto fn = copy1<LANES, ROWS, VEC, LW, SW, SPREAD>;
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(fn, dim3((unsigned)tiles), dim3(LANES), lds, 0, in, out, tiles);
    CK(hipDeviceSynchronize());
    const int reps = 30;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(fn, dim3((unsigned)tiles), dim3(LANES), lds, 0, in, out, tiles);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("lanes %4d rows %2d x %2d B tile %3lld KB loads %s stores %s %s wg/CU %d: %.3f ms %5.0f GB/s\n", LANES, ROWS, VEC * 4,
           per_tile * 4 / 1024, LW ? "wave-private" : "wg-wide     ", SW ? "wave-private" : "wg-wide     ",
           SPREAD ? "spread " : "compact", wg_per_cu, ms, 2.0 * tiles * per_tile * 4 / ms / 1e6);
    fflush(stdout);
}

int main() {
    const long long nfloat = 1LL << 29; // 2 GiB
    float *in, *out;
    CK(hipMalloc(&in, nfloat * 4));
    CK(hipMalloc(&out, nfloat * 4));
    CK(hipMemset(in, 1, nfloat * 4));
    CK(hipMemset(out, 0, nfloat * 4));
    printf("# row width at a 32 KB tile, occupancy 4 tiles (128 KB) per CU\n");
    run<256, 16, 2, false, false, false>(in, out, nfloat, 4);
    run<512, 8, 2, false, false, false>(in, out, nfloat, 4);
    run<1024, 4, 2, false, false, false>(in, out, nfloat, 4);
    run<256, 8, 4, false, false, false>(in, out, nfloat, 4);
    run<512, 4, 4, false, false, false>(in, out, nfloat, 4);
    printf("# the same at 3 tiles per CU (the FFT kernel's residency)\n");
    run<256, 16, 2, false, false, false>(in, out, nfloat, 3);
    run<512, 8, 2, false, false, false>(in, out, nfloat, 3);
    run<256, 8, 4, false, false, false>(in, out, nfloat, 3);
    printf("# wave-private vs workgroup-wide rows, 32 KB tile, 256 lanes, 8 B\n");
    run<256, 16, 2, true, true, false>(in, out, nfloat, 4);
    run<256, 16, 2, true, false, false>(in, out, nfloat, 4);
    run<256, 16, 2, false, true, false>(in, out, nfloat, 4);
    printf("# compact vs spread window, 32 KB and 8 KB tiles\n");
    run<256, 16, 2, false, false, true>(in, out, nfloat, 4);
    run<256, 4, 2, false, false, false>(in, out, nfloat, 8);
    run<256, 4, 2, false, false, true>(in, out, nfloat, 8);
    printf("# tile size at 256 lanes x 8 B, 4 and 8 workgroups per CU\n");
    run<256, 8, 2, false, false, false>(in, out, nfloat, 4);
    run<256, 8, 2, false, false, false>(in, out, nfloat, 8);
    run<256, 4, 2, false, false, false>(in, out, nfloat, 4);
    run<256, 2, 2, false, false, false>(in, out, nfloat, 8);
    run<256, 4, 4, false, false, false>(in, out, nfloat, 8);
    run<256, 2, 4, false, false, false>(in, out, nfloat, 8);
    CK(hipFree(in));
    CK(hipFree(out));
    return 0;
}
