// membench.hip -- which streaming access pattern reaches the HBM rate on MI355X?
// Persistent copy kernels shaped like the FFT kernel's tile traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// tile = 4096 complex (32 KB). 256 threads. W8: 16 x 8B loads per thread; W16: 8 x 16B.
template <int WIDTH, bool PREFETCH>
__global__ __launch_bounds__(256) void copy_tiles(const float* __restrict__ in, float* __restrict__ out, long long tiles) {
    const int tid = threadIdx.x;
    constexpr int NL = WIDTH == 8 ? 16 : 8;
    auto load = [&](long long t, f4 (&d)[8]) {
        const float* p = in + t * 8192;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            if (WIDTH == 8) { f2 v = reinterpret_cast<const f2*>(p + 512 * i)[tid]; if (i & 1) { d[i/2].z = v.x; d[i/2].w = v.y; } else { d[i/2].x = v.x; d[i/2].y = v.y; } }
            else d[i] = reinterpret_cast<const f4*>(p + 1024 * i)[tid];
        }
    };
    auto store = [&](long long t, f4 (&d)[8]) {
        float* p = out + t * 8192;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            if (WIDTH == 8) { f2 v = (i & 1) ? f2{d[i/2].z, d[i/2].w} : f2{d[i/2].x, d[i/2].y}; reinterpret_cast<f2*>(p + 512 * i)[tid] = v; }
            else reinterpret_cast<f4*>(p + 1024 * i)[tid] = d[i];
        }
    };
    long long t = blockIdx.x;
    if (t >= tiles) return;
    f4 nx[8], v[8];
    if (PREFETCH) load(t, nx);
    for (; t < tiles; t += gridDim.x) {
        if (PREFETCH) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = nx[i];
            long long n = t + gridDim.x < tiles ? t + gridDim.x : t;
            load(n, nx);
        } else load(t, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = v[i] * 1.0001f;
        store(t, v);
    }
}

template <class K> float run(K k, int grid, const float* in, float* out, long long tiles) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, in, out, tiles);
    CK(hipEventRecord(a));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, in, out, tiles);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / 10;
}

int main() {
    const long long tiles = 65536; // 2 GiB in, 2 GiB out
    float *in, *out;
    CK(hipMalloc(&in, tiles * 32768)); CK(hipMalloc(&out, tiles * 32768));
    CK(hipMemset(in, 1, tiles * 32768));
    const double gb = 2.0 * tiles * 32768 / 1e9;
    for (int bpc : {1, 2, 3, 4, 8}) {
        int grid = 256 * bpc;
        float t;
        t = run(copy_tiles<8, false>, grid, in, out, tiles);  printf("8B  noprefetch %d blocks/CU: %.3f ms %.0f GB/s\n", bpc, t, gb / t * 1e3);
        t = run(copy_tiles<8, true>, grid, in, out, tiles);   printf("8B  prefetch   %d blocks/CU: %.3f ms %.0f GB/s\n", bpc, t, gb / t * 1e3);
        t = run(copy_tiles<16, false>, grid, in, out, tiles); printf("16B noprefetch %d blocks/CU: %.3f ms %.0f GB/s\n", bpc, t, gb / t * 1e3);
        t = run(copy_tiles<16, true>, grid, in, out, tiles);  printf("16B prefetch   %d blocks/CU: %.3f ms %.0f GB/s\n", bpc, t, gb / t * 1e3);
    }
    return 0;
}
