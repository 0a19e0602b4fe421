// membench3.hip -- why is a 32 KB-per-workgroup copy slower than a 4 KB-per-workgroup one?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
// MODE 0: block b copies 8 contiguous 4 KB chunks (32 KB contiguous).
// MODE 1: block b copies chunks b, b+G, b+2G.. (interleaved across blocks), all 8 loads in flight.
// MODE 2: like 0 but load->store one chunk at a time (1 load in flight per thread).
// MODE 3: like 0 but the 8 chunks of a block are issued by wave order: wave w loads chunks {w, w+4} * 2 halves
template <int MODE>
__global__ __launch_bounds__(256) void copy8(const f4* __restrict__ in, f4* __restrict__ out, long long nchunks) {
    const long long G = gridDim.x;
    const int t = threadIdx.x;
    f4 v[8];
    if (MODE == 2) {
#pragma unroll 1
        for (int k = 0; k < 8; ++k) { long long c = (long long)blockIdx.x * 8 + k; out[c * 256 + t] = in[c * 256 + t]; }
        return;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { long long c = MODE == 1 ? blockIdx.x + k * G : (long long)blockIdx.x * 8 + k; v[k] = in[c * 256 + t]; }
#pragma unroll
    for (int k = 0; k < 8; ++k) { long long c = MODE == 1 ? blockIdx.x + k * G : (long long)blockIdx.x * 8 + k; out[c * 256 + t] = v[k]; }
}
// dwordx2 version of mode 0 with the FFT kernel's lane pattern (16 x 8 B per thread, stride 2 KB)
typedef float f2 __attribute__((ext_vector_type(2)));
template <bool NT>
__global__ __launch_bounds__(256) void copy16x8(const f2* __restrict__ in, f2* __restrict__ out, long long ntiles) {
    const f2* p = in + (long long)blockIdx.x * 4096; f2* q = out + (long long)blockIdx.x * 4096;
    f2 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = NT ? __builtin_nontemporal_load(&p[k * 256 + threadIdx.x]) : p[k * 256 + threadIdx.x];
#pragma unroll
    for (int k = 0; k < 16; ++k) { if (NT) __builtin_nontemporal_store(v[k], &q[k * 256 + threadIdx.x]); else q[k * 256 + threadIdx.x] = v[k]; }
}
template <class F> float timeit(F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipEventRecord(a));
    for (int i = 0; i < 10; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / 10;
}
int main() {
    const long long bytes = 2LL << 30, nchunks = bytes / 4096;
    f4 *in, *out;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes)); CK(hipMemset(in, 1, bytes));
    const double gb = 2.0 * bytes / 1e9;
    float t;
    t = timeit([&] { hipLaunchKernelGGL((copy8<0>), dim3(nchunks / 8), dim3(256), 0, 0, in, out, nchunks); }); printf("32KB contiguous per block, 8 in flight: %.3f ms %.0f GB/s\n", t, gb / t * 1e3);
    t = timeit([&] { hipLaunchKernelGGL((copy8<1>), dim3(nchunks / 8), dim3(256), 0, 0, in, out, nchunks); }); printf("8 x 4KB interleaved across blocks:      %.3f ms %.0f GB/s\n", t, gb / t * 1e3);
    t = timeit([&] { hipLaunchKernelGGL((copy8<2>), dim3(nchunks / 8), dim3(256), 0, 0, in, out, nchunks); }); printf("32KB contiguous, 1 in flight:           %.3f ms %.0f GB/s\n", t, gb / t * 1e3);
    t = timeit([&] { hipLaunchKernelGGL((copy16x8<false>), dim3(nchunks / 8), dim3(256), 0, 0, (f2*)in, (f2*)out, nchunks / 8); }); printf("16 x 8B per thread (FFT shape):         %.3f ms %.0f GB/s\n", t, gb / t * 1e3);
    t = timeit([&] { hipLaunchKernelGGL((copy16x8<true>), dim3(nchunks / 8), dim3(256), 0, 0, (f2*)in, (f2*)out, nchunks / 8); }); printf("16 x 8B per thread nontemporal:         %.3f ms %.0f GB/s\n", t, gb / t * 1e3);
    // separate buffers far apart vs out = in + offset inside one allocation does not matter here; try in-place
    t = timeit([&] { hipLaunchKernelGGL((copy8<0>), dim3(nchunks / 8), dim3(256), 0, 0, in, in, nchunks); }); printf("32KB contiguous IN PLACE:               %.3f ms %.0f GB/s\n", t, gb / t * 1e3);
    return 0;
}
