// membench2.hip -- simple copies: plain vs nontemporal, 1 or 4 float4 per thread.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int PER, bool NT>
__global__ __launch_bounds__(256) void copy_simple(const f4* __restrict__ in, f4* __restrict__ out, long long n) {
    long long i = ((long long)blockIdx.x * 256 * PER) + threadIdx.x;
    f4 v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) v[k] = NT ? __builtin_nontemporal_load(&in[i + 256 * k]) : in[i + 256 * k];
#pragma unroll
    for (int k = 0; k < PER; ++k) { if (NT) __builtin_nontemporal_store(v[k], &out[i + 256 * k]); else out[i + 256 * k] = v[k]; }
}
template <bool NT>
__global__ __launch_bounds__(256) void copy_persist(const f4* __restrict__ in, f4* __restrict__ out, long long tiles) {
    for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
        const f4* p = in + t * 2048; f4* q = out + t * 2048;
        f4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = NT ? __builtin_nontemporal_load(&p[threadIdx.x + 256 * k]) : p[threadIdx.x + 256 * k];
#pragma unroll
        for (int k = 0; k < 8; ++k) { if (NT) __builtin_nontemporal_store(v[k], &q[threadIdx.x + 256 * k]); else q[threadIdx.x + 256 * k] = v[k]; }
    }
}
__global__ __launch_bounds__(256) void read_only(const f4* __restrict__ in, float* out, long long n) {
    long long i = ((long long)blockIdx.x * 256 * 8) + threadIdx.x;
    f4 s = {0,0,0,0};
#pragma unroll
    for (int k = 0; k < 8; ++k) s += in[i + 256 * k];
    if (s.x == 123.456f) out[0] = s.y;
}
__global__ __launch_bounds__(256) void write_only(f4* out, long long n) {
    long long i = ((long long)blockIdx.x * 256 * 8) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 8; ++k) out[i + 256 * k] = f4{1,2,3,4};
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
    const long long bytes = 2LL << 30, n = bytes / 16;
    f4 *in, *out;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes)); CK(hipMemset(in, 1, bytes));
    const double gb = 2.0 * bytes / 1e9;
    float t;
    t = timeit([&] { hipLaunchKernelGGL((copy_simple<1, false>), dim3(n / 256), dim3(256), 0, 0, in, out, n); }); printf("simple 1x plain: %.3f ms %.0f GB/s\n", t, gb / t * 1e3);
    t = timeit([&] { hipLaunchKernelGGL((copy_simple<4, false>), dim3(n / 1024), dim3(256), 0, 0, in, out, n); }); printf("simple 4x plain: %.3f ms %.0f GB/s\n", t, gb / t * 1e3);
    t = timeit([&] { hipLaunchKernelGGL((copy_simple<8, false>), dim3(n / 2048), dim3(256), 0, 0, in, out, n); }); printf("simple 8x plain: %.3f ms %.0f GB/s\n", t, gb / t * 1e3);
    t = timeit([&] { hipLaunchKernelGGL((copy_simple<4, true>), dim3(n / 1024), dim3(256), 0, 0, in, out, n); }); printf("simple 4x nt:    %.3f ms %.0f GB/s\n", t, gb / t * 1e3);
    t = timeit([&] { hipLaunchKernelGGL((copy_simple<8, true>), dim3(n / 2048), dim3(256), 0, 0, in, out, n); }); printf("simple 8x nt:    %.3f ms %.0f GB/s\n", t, gb / t * 1e3);
    for (int bpc : {2, 4, 8}) {
        t = timeit([&] { hipLaunchKernelGGL((copy_persist<false>), dim3(256 * bpc), dim3(256), 0, 0, in, out, n / 2048); }); printf("persist plain %d/CU: %.3f ms %.0f GB/s\n", bpc, t, gb / t * 1e3);
        t = timeit([&] { hipLaunchKernelGGL((copy_persist<true>), dim3(256 * bpc), dim3(256), 0, 0, in, out, n / 2048); }); printf("persist nt    %d/CU: %.3f ms %.0f GB/s\n", bpc, t, gb / t * 1e3);
    }
    t = timeit([&] { hipLaunchKernelGGL(read_only, dim3(n / 2048), dim3(256), 0, 0, in, (float*)out, n); }); printf("read only: %.3f ms %.0f GB/s\n", t, gb / 2 / t * 1e3);
    t = timeit([&] { hipLaunchKernelGGL(write_only, dim3(n / 2048), dim3(256), 0, 0, out, n); }); printf("write only: %.3f ms %.0f GB/s\n", t, gb / 2 / t * 1e3);
    t = timeit([&] { CK(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0)); }); printf("hipMemcpy D2D: %.3f ms %.0f GB/s\n", t, gb / t * 1e3);
    return 0;
}
