// membench9.hip -- round 2: skeleton of a ONE-TILE-PER-WORKGROUP overlap-save kernel on 2048-point tiles
// (16 KB in, 16 KB - halo out), before building the transform.  What fir_fft4096_once (4096 points, 256
// lanes, 4 workgroups per CU) lacked was independent chains per CU; here a workgroup is WG = 128 or 256
// lanes (16 or 8 points per lane), so 8-10 fit a CU.  Stand-in for the transform, in the real kernel's
// dependency order: loads -> [ops/4, exchange] x 2 -> product with a per-tile-reloaded 16 KB spectrum
// (L2 resident; `tab` = 1) -> [ops/4, exchange] x 2 -> stores; `tw` extra per-lane table loads stand for
// twiddles that a persistent kernel would keep in registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int WG, int HALO>
__global__ __launch_bounds__(WG) void tile_once(const f2* __restrict__ in, f2* __restrict__ out, const f2* __restrict__ spec,
                                                const f2* __restrict__ twid, long long tiles, int xch, int ops, int tab, int tw) {
    constexpr int ROWS = 2048 / WG;
    constexpr int PITCH = WG + 8;
    extern __shared__ f2 lds[];
    const int tid = threadIdx.x;
    constexpr long long L = 2048 - HALO;
    const long long t = blockIdx.x;
    f2 v[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; ++k) v[k] = __builtin_nontemporal_load(&in[t * L + WG * k + tid]);
    f2 w[4] = {f2{1.f, 0.f}, f2{1.f, 0.f}, f2{1.f, 0.f}, f2{1.f, 0.f}};
    for (int i = 0; i < tw && i < 4; ++i) w[i] = twid[i * WG + tid];
    auto work = [&](int n) {
        for (int o = 0; o < n; o += ROWS) {
#pragma unroll
            for (int k = 0; k < ROWS; ++k) v[k] = __builtin_elementwise_fma(v[k], w[k & 3], v[(k + 1) % ROWS] * 1e-30f);
        }
    };
    auto exchange = [&]() {
#pragma unroll
        for (int k = 0; k < ROWS; ++k) lds[k * PITCH + tid] = v[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ROWS; ++k) v[k] = lds[(tid / (WG / ROWS)) * PITCH + (WG / ROWS) * k + (tid % (WG / ROWS))];
        __syncthreads();
    };
    const int q = ops / 4;
    for (int x = 0; x < xch / 2; ++x) { work(q); exchange(); }
    if (xch == 0) work(2 * q);
    if (tab) {
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
            const f2 h = spec[WG * k + tid];
            v[k] = f2{v[k].x * h.x - v[k].y * h.y, v[k].x * h.y + v[k].y * h.x};
        }
    }
    for (int x = 0; x < xch / 2; ++x) { work(q); exchange(); }
    if (xch == 0) work(2 * q);
#pragma unroll
    for (int k = HALO / WG; k < ROWS; ++k) {
        if (k == HALO / WG && HALO % WG != 0 && tid < HALO % WG) continue;
        __builtin_nontemporal_store(v[k], &out[t * L + WG * k + tid - HALO]);
    }
}

template <int WG, int HALO>
static void run(const f2* in, f2* out, const f2* spec, const f2* twid, long long n, int wg_per_cu, int xch, int ops, int tab, int tw) {
    constexpr long long L = 2048 - HALO;
    constexpr int ROWS = 2048 / WG;
    const long long tiles = (n - 2048) / L + 1;
    size_t lds = ((size_t)(160 * 1024 / wg_per_cu)) & ~(size_t)255;
    if (lds < (size_t)ROWS * (WG + 8) * 8) { printf("LDS too small for %d per CU\n", wg_per_cu); return; }
    auto fn = tile_once<WG, HALO>;
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(fn, dim3((unsigned)tiles), dim3(WG), lds, 0, in, out, spec, twid, tiles, xch, ops, tab, tw);
    CK(hipDeviceSynchronize());
    const int reps = 30;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(fn, dim3((unsigned)tiles), dim3(WG), lds, 0, in, out, spec, twid, tiles, xch, ops, tab, tw);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("2048-point tile, %3d lanes (%2d points per lane), halo %3d, %2d wg/CU by LDS, xch %d ops %4d spectrum %d twiddle loads %d: %.3f ms %5.0f GB/s (outputs x 16 B)\n",
           WG, ROWS, HALO, wg_per_cu, xch, ops, tab, tw, ms, 2.0 * tiles * L * sizeof(f2) / ms / 1e6);
    fflush(stdout);
}

int main() {
    const long long n = 1LL << 28;
    f2 *in, *out, *spec, *twid;
    CK(hipMalloc(&in, n * sizeof(f2)));
    CK(hipMalloc(&out, n * sizeof(f2)));
    CK(hipMalloc(&spec, 2048 * sizeof(f2)));
    CK(hipMalloc(&twid, 4 * 256 * sizeof(f2)));
    CK(hipMemset(in, 1, n * sizeof(f2)));
    CK(hipMemset(out, 0, n * sizeof(f2)));
    CK(hipMemset(spec, 1, 2048 * sizeof(f2)));
    CK(hipMemset(twid, 1, 4 * 256 * sizeof(f2)));
    printf("# 256 lanes x 8 points\n");
    for (int wpc : {4, 5, 6, 8}) {
        run<256, 128>(in, out, spec, twid, n, wpc, 0, 0, 0, 0);
        run<256, 128>(in, out, spec, twid, n, wpc, 2, 328, 0, 0);
        run<256, 128>(in, out, spec, twid, n, wpc, 4, 328, 0, 0);
        run<256, 128>(in, out, spec, twid, n, wpc, 4, 328, 1, 0);
        run<256, 128>(in, out, spec, twid, n, wpc, 4, 328, 1, 4);
        run<256, 128>(in, out, spec, twid, n, wpc, 6, 328, 1, 1);
    }
    printf("# 128 lanes x 16 points\n");
    for (int wpc : {6, 8, 9}) {
        run<128, 128>(in, out, spec, twid, n, wpc, 0, 0, 0, 0);
        run<128, 128>(in, out, spec, twid, n, wpc, 2, 656, 0, 0);
        run<128, 128>(in, out, spec, twid, n, wpc, 4, 656, 0, 0);
        run<128, 128>(in, out, spec, twid, n, wpc, 4, 656, 1, 0);
        run<128, 128>(in, out, spec, twid, n, wpc, 4, 656, 1, 4);
    }
    printf("# halo 256 (257 taps)\n");
    run<256, 256>(in, out, spec, twid, n, 6, 4, 328, 1, 1);
    run<128, 256>(in, out, spec, twid, n, 9, 4, 656, 1, 1);
    CK(hipFree(in));
    CK(hipFree(out));
    return 0;
}
