// membench8.hip -- round 2: overlap-save copy skeleton with 16 KB tiles (2048 complex points, 8 loads of
// 8 B per lane in a 256-lane workgroup) against the 32 KB tiles of fir_fft4096.  A tile reads ROWS*256
// points and writes the last ROWS*256 - HALO of them; tile stride = ROWS*256 - HALO.  Modes:
//   K = 0: persistent, grid-strided (grid = gpc workgroups per CU)
//   K > 0: workgroup b walks the K consecutive tiles [bK, bK + K) and exits (hardware dispatch order)
//   PF: the next tile's loads are issued before the current tile's stores
// optional stand-in for the transform: `xch` workgroup-wide LDS exchanges + `ops` packed FMAs per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int ROWS, int HALO, bool PF>
__global__ __launch_bounds__(256) void copy_os(const f2* __restrict__ in, f2* __restrict__ out, long long tiles, int K,
                                               int xch, int ops) {
    extern __shared__ f2 lds[];
    const int tid = threadIdx.x;
    constexpr long long L = ROWS * 256 - HALO;
    long long t = K ? (long long)blockIdx.x * K : blockIdx.x;
    const long long step = K ? 1 : gridDim.x;
    const long long end = K ? (t + K < tiles ? t + K : tiles) : tiles;
    if (t >= end) return;
    f2 nx[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; ++k) nx[k] = __builtin_nontemporal_load(&in[t * L + 256 * k + tid]);
    for (; t < end; t += step) {
        f2 v[ROWS];
#pragma unroll
        for (int k = 0; k < ROWS; ++k) v[k] = nx[k];
        const long long tn = t + step;
        const bool more = tn < end; // no redundant prefetch past the run's last tile (wave-uniform branch)
        if (PF && more) {
#pragma unroll
            for (int k = 0; k < ROWS; ++k) nx[k] = __builtin_nontemporal_load(&in[tn * L + 256 * k + tid]);
        }
        for (int x = 0; x < xch; ++x) {
#pragma unroll
            for (int k = 0; k < ROWS; ++k) lds[k * 264 + tid] = v[k];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < ROWS; ++k) v[k] = lds[(tid >> 5) * 264 + 32 * k + (tid & 31)];
            __syncthreads();
        }
        for (int o = 0; o < ops; o += ROWS) {
#pragma unroll
            for (int k = 0; k < ROWS; ++k) v[k] = __builtin_elementwise_fma(v[k], f2{1.0f, 1.0f}, v[(k + 1) % ROWS] * 1e-30f);
        }
#pragma unroll
        for (int k = HALO / 256; k < ROWS; ++k) {
            if (k == HALO / 256 && HALO % 256 != 0 && tid < HALO % 256) continue;
            __builtin_nontemporal_store(v[k], &out[t * L + 256 * k + tid - HALO]);
        }
        if (!PF && more) {
#pragma unroll
            for (int k = 0; k < ROWS; ++k) nx[k] = __builtin_nontemporal_load(&in[tn * L + 256 * k + tid]);
        }
    }
}

template <int ROWS, int HALO, bool PF>
static void run(const f2* in, f2* out, long long n, int K, int wg_per_cu, int gpc, int xch, int ops) {
    constexpr long long L = ROWS * 256 - HALO;
    const long long tiles = (n - ROWS * 256) / L + 1;
    size_t lds = ((size_t)(160 * 1024 / wg_per_cu)) & ~(size_t)255;
    if (lds < (size_t)ROWS * 264 * 8) { printf("LDS too small\n"); return; }
    auto fn = copy_os<ROWS, HALO, PF>;
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long long grid = K ? (tiles + K - 1) / K : 256LL * gpc;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(256), lds, 0, in, out, tiles, K, xch, ops);
    CK(hipDeviceSynchronize());
    const int reps = 30;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(256), lds, 0, in, out, tiles, K, xch, ops);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    char mode[64];
    if (K) snprintf(mode, sizeof mode, "runs of %3d, dispatch order", K);
    else snprintf(mode, sizeof mode, "persistent, grid %2d/CU     ", gpc);
    printf("tile %2d KB halo %3d %s %s wg/CU %d xch %d ops %3d: %.3f ms %5.0f GB/s (outputs x 16 B)\n", ROWS * 2, HALO, mode,
           PF ? "prefetch" : "no-pf   ", wg_per_cu, xch, ops, ms, 2.0 * tiles * L * sizeof(f2) / ms / 1e6);
    fflush(stdout);
}

int main() {
    const long long n = 1LL << 28;
    f2 *in, *out;
    CK(hipMalloc(&in, n * sizeof(f2)));
    CK(hipMalloc(&out, n * sizeof(f2)));
    CK(hipMemset(in, 1, n * sizeof(f2)));
    CK(hipMemset(out, 0, n * sizeof(f2)));
    printf("# copy only, no redundant prefetch: 16 KB and 32 KB tiles, halo 0 / 128, run length\n");
    for (int K : {1, 2, 4, 8, 32}) {
        run<8, 0, true>(in, out, n, K, 4, 0, 0, 0);
        run<8, 128, true>(in, out, n, K, 4, 0, 0, 0);
        run<8, 128, true>(in, out, n, K, 5, 0, 0, 0);
        run<16, 0, true>(in, out, n, K, 4, 0, 0, 0);
        run<16, 128, true>(in, out, n, K, 4, 0, 0, 0);
        run<16, 128, true>(in, out, n, K, 3, 0, 0, 0);
    }
    run<8, 128, false>(in, out, n, 1, 4, 0, 0, 0);
    run<8, 128, false>(in, out, n, 4, 4, 0, 0, 0);
    run<16, 128, false>(in, out, n, 1, 4, 0, 0, 0);
    run<8, 128, true>(in, out, n, 0, 4, 8, 0, 0);
    run<16, 128, true>(in, out, n, 0, 3, 9, 0, 0);
    printf("# with stand-in work\n");
    for (int K : {1, 4, 8}) {
        run<8, 128, true>(in, out, n, K, 4, 0, 2, 328);
        run<8, 128, true>(in, out, n, K, 5, 0, 2, 328);
        run<16, 128, true>(in, out, n, K, 3, 0, 2, 688);
        run<16, 128, true>(in, out, n, K, 4, 0, 2, 688);
    }
    run<16, 128, true>(in, out, n, 0, 3, 9, 2, 688);
    CK(hipFree(in));
    CK(hipFree(out));
    return 0;
}
