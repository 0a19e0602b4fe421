// membench6.hip -- round 2: what does the memory system deliver to WAVE-PRIVATE small tiles?
// 2 GiB -> 2 GiB copy of f2 elements in the overlap-save shape: a wave64 owns tiles of 1024
// points (16 loads of 8 B per lane, 512 B contiguous per wave-instruction), writes the last
// 1024 - HALO of them, tile stride 1024 - HALO.  No workgroup barrier anywhere: the waves of a
// workgroup are independent.  Variables:
//   WAVES   waves per workgroup (1, 4, 8, 16)
//   K       tiles per wave (runtime); grid = tiles / (WAVES * K)  (hardware dispatch order)
//   RUNS    wave w walks K consecutive tiles / the workgroup's waves interleave (tile = t*WAVES + w)
//   PF      next tile's loads issued before the current tile's stores (32 more VGPRs)
//   WORK    stand-in for the transform: X LDS exchanges (16 ds_write_b64 + 16 ds_read_b64 per lane,
//           wave-private 8 KB buffer) and OPS dependent packed FMAs per lane per tile
//   occupancy is set by the dynamic LDS size (bytes per wave)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int WAVES, bool RUNS, bool PF, int HALO>
__global__ __launch_bounds__(WAVES * 64) void copy_wave_tiles(const f2* __restrict__ in, f2* __restrict__ out,
                                                              long long tiles, int K, int xch, int ops,
                                                              int lds_per_wave_f2) {
    extern __shared__ f2 lds[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f2* buf = lds + (size_t)w * lds_per_wave_f2;
    constexpr long long L = 1024 - HALO;
    const long long wg0 = (long long)blockIdx.x * WAVES * K;
    long long t = RUNS ? wg0 + (long long)w * K : wg0 + w;
    const long long step = RUNS ? 1 : WAVES;
    if (t >= tiles) return;
    f2 nx[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) nx[k] = __builtin_nontemporal_load(&in[t * L + 64 * k + lane]);
    for (int i = 0; i < K && t < tiles; ++i, t += step) {
        f2 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = nx[k];
        const long long tn = (i + 1 < K && t + step < tiles) ? t + step : t;
        if (PF) {
#pragma unroll
            for (int k = 0; k < 16; ++k) nx[k] = __builtin_nontemporal_load(&in[tn * L + 64 * k + lane]);
        }
        for (int x = 0; x < xch; ++x) {
#pragma unroll
            for (int k = 0; k < 16; ++k) buf[k * 64 + (lane ^ (2 * k))] = v[k];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = buf[(lane & 15) * 64 + ((4 * k + (lane >> 4)) ^ (2 * (lane & 15)))];
            __builtin_amdgcn_wave_barrier();
        }
        for (int o = 0; o < ops; o += 16) {
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = __builtin_elementwise_fma(v[k], f2{1.0f, 1.0f}, v[(k + 1) & 15] * 1e-30f);
        }
#pragma unroll
        for (int k = HALO / 64; k < 16; ++k) __builtin_nontemporal_store(v[k], &out[t * L + 64 * k + lane - HALO]);
        if (!PF) {
#pragma unroll
            for (int k = 0; k < 16; ++k) nx[k] = __builtin_nontemporal_load(&in[tn * L + 64 * k + lane]);
        }
    }
}

template <int WAVES, bool RUNS, bool PF, int HALO>
static void run(const f2* in, f2* out, long long n, int K, int waves_per_cu, int xch, int ops) {
    constexpr long long L = 1024 - HALO;
    const long long tiles = (n - 1024) / L + 1;
    const long long grid = (tiles + (long long)WAVES * K - 1) / ((long long)WAVES * K);
    // LDS per wave so that waves_per_cu waves fit in 160 KB (at least the 8 KB exchange buffer)
    size_t per_wave = ((size_t)(160 * 1024 / waves_per_cu)) & ~(size_t)255;
    if (per_wave < 8192) { printf("occupancy too high for an 8 KB exchange buffer\n"); return; }
    const size_t lds = per_wave * WAVES;
    if (lds > 160 * 1024) { printf("workgroup too large\n"); return; }
    auto fn = copy_wave_tiles<WAVES, RUNS, PF, HALO>;
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(WAVES * 64), lds, 0, in, out, tiles, K, xch, ops, (int)(per_wave / 8));
    CK(hipDeviceSynchronize());
    const int reps = 30;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(WAVES * 64), lds, 0, in, out, tiles, K, xch, ops, (int)(per_wave / 8));
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("waves/wg %2d K %5d %s %s halo %3d waves/CU %2d xch %d ops %4d: %.3f ms %5.0f GB/s (outputs x 16 B)\n", WAVES, K,
           RUNS ? "runs       " : "interleaved", PF ? "prefetch" : "no-pf   ", HALO, waves_per_cu, xch, ops, ms,
           2.0 * tiles * L * sizeof(f2) / ms / 1e6);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const long long n = 1LL << 28;
    f2 *in, *out;
    CK(hipMalloc(&in, n * sizeof(f2)));
    CK(hipMalloc(&out, n * sizeof(f2)));
    CK(hipMemset(in, 1, n * sizeof(f2)));
    CK(hipMemset(out, 0, n * sizeof(f2)));
    const int set = argc > 1 ? atoi(argv[1]) : 0;
    if (set == 0 || set == 1) {
        printf("# copy only: tiles per wave, workgroup size, order (16 waves/CU)\n");
        for (int K : {1, 2, 4, 16, 64}) {
            run<4, false, true, 128>(in, out, n, K, 16, 0, 0);
            run<4, true, true, 128>(in, out, n, K, 16, 0, 0);
        }
        for (int K : {1, 4, 16}) {
            run<1, true, true, 128>(in, out, n, K, 16, 0, 0);
            run<8, false, true, 128>(in, out, n, K, 16, 0, 0);
            run<8, true, true, 128>(in, out, n, K, 16, 0, 0);
            run<16, false, true, 128>(in, out, n, K, 16, 0, 0);
        }
        printf("# no prefetch\n");
        for (int K : {1, 4, 16}) {
            run<4, false, false, 128>(in, out, n, K, 16, 0, 0);
            run<8, false, false, 128>(in, out, n, K, 16, 0, 0);
        }
        printf("# occupancy (copy only, K = 4 and 16, 4 and 8 waves per workgroup, interleaved)\n");
        for (int wpc : {8, 12, 16, 20}) {
            run<4, false, true, 128>(in, out, n, 4, wpc, 0, 0);
            run<4, false, true, 128>(in, out, n, 16, wpc, 0, 0);
            if (wpc % 8 == 0) run<8, false, true, 128>(in, out, n, 4, wpc, 0, 0);
        }
        printf("# halo 0 / 256\n");
        run<4, false, true, 0>(in, out, n, 4, 16, 0, 0);
        run<4, false, true, 256>(in, out, n, 4, 16, 0, 0);
        run<8, false, true, 0>(in, out, n, 4, 16, 0, 0);
    }
    if (set == 0 || set == 2) {
        printf("# with stand-in work: 2 exchanges + 480 packed ops per lane per tile (the planned transform)\n");
        for (int wpc : {12, 16}) {
            for (int K : {4, 16}) {
                run<4, false, true, 128>(in, out, n, K, wpc, 2, 480);
                run<4, true, true, 128>(in, out, n, K, wpc, 2, 480);
                run<8, false, true, 128>(in, out, n, K, wpc, 2, 480);
                run<4, false, false, 128>(in, out, n, K, wpc, 2, 480);
            }
        }
        run<4, false, true, 128>(in, out, n, 4, 16, 2, 0);
        run<4, false, true, 128>(in, out, n, 4, 16, 0, 480);
        run<4, false, true, 128>(in, out, n, 4, 16, 4, 640);
        run<4, false, true, 128>(in, out, n, 4, 16, 2, 800);
    }
    CK(hipFree(in));
    CK(hipFree(out));
    return 0;
}
