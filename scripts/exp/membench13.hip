// membench13.hip -- round 3: membench10 with 2 / 3 / 5 / 6 ticket counters (three beat two in fir_fft4096: does the copy skeleton agree?).
// membench10.hip -- round 2: WHY does a looping workgroup move fewer bytes than one-tile workgroups
// (r02_membench_runs.txt: 5.2-5.4 against 5.9-6.0 TB/s)?  Hypothesis: hardware dispatch keeps the tiles in
// flight a dense, ordered window of the buffers; grid-strided persistent workgroups drift apart (CUs differ
// in memory latency), the window widens, DRAM rows and translations are shared by fewer accesses.
// Test: persistent workgroups that take their tiles from a global ticket counter IN ORDER, the ticket for
// the tile after next fetched while the current tile is in flight (no exposed atomic latency; the r01
// "atomic queue" of membench4 had it exposed and two extra barriers per tile).
//   MODE 0: persistent, grid-strided      MODE 1: persistent, ordered tickets      MODE 2: one tile per workgroup
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int ROWS, int HALO, int MODE, int NC>
__global__ __launch_bounds__(256) void copy_os(const f2* __restrict__ in, f2* __restrict__ out, long long tiles,
                                               unsigned* counter, int xch, int ops) {
    extern __shared__ f2 lds[];
    __shared__ unsigned ticket_box;
    const int tid = threadIdx.x;
    constexpr long long L = ROWS * 256 - HALO;
    long long t = blockIdx.x;
    if (t >= tiles) return;
    long long tn = MODE == 2 ? tiles : t + gridDim.x; // the first two tiles of a workgroup are static
    f2 nx[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; ++k) nx[k] = __builtin_nontemporal_load(&in[t * L + 256 * k + tid]);
    while (true) {
        f2 v[ROWS];
#pragma unroll
        for (int k = 0; k < ROWS; ++k) v[k] = nx[k];
        const bool more = tn < tiles;
        if (more) {
#pragma unroll
            for (int k = 0; k < ROWS; ++k) nx[k] = __builtin_nontemporal_load(&in[tn * L + 256 * k + tid]);
        }
        unsigned ticket = 0;
        if (MODE == 1 && tid == 0) ticket = atomicAdd(counter + 64 * (blockIdx.x % NC), 1u); // tile after next; NC counters, 256 B apart
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
        if (!more) break;
        t = tn;
        if (MODE == 1) {
            if (tid == 0) ticket_box = ticket;
            __syncthreads();
            tn = 2LL * gridDim.x + (long long)ticket_box * NC + blockIdx.x % NC;
            __syncthreads();
        } else {
            tn = t + gridDim.x;
        }
    }
}

// pseudo-random fill (the bench's inputs are uniform random floats; a constant byte pattern toggles far fewer lines)
__global__ void fill_random(unsigned* p, long long nwords) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (long long)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ (unsigned)(i >> 32) * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float f = (float)(h >> 8) * (2.0f / 16777216.0f) - 1.0f; // uniform [-1, 1)
        p[i] = __float_as_uint(f);
    }
}

template <int ROWS, int HALO, int MODE, int NC = 8>
static void run(const f2* in, f2* out, unsigned* counter, long long n, int wg_per_cu, int gpc, int xch, int ops) {
    constexpr long long L = ROWS * 256 - HALO;
    const long long tiles = (n - ROWS * 256) / L + 1;
    size_t lds = ((size_t)(160 * 1024 / wg_per_cu) - 64) & ~(size_t)255;
    if (lds < (size_t)ROWS * 264 * 8) { printf("LDS too small\n"); return; }
    auto fn = copy_os<ROWS, HALO, MODE, NC>;
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long long grid = MODE == 2 ? tiles : 256LL * gpc;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto launch = [&]() {
        if (MODE == 1) CK(hipMemsetAsync(counter, 0, 64 * 256, 0));
        hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(256), lds, 0, in, out, tiles, counter, xch, ops);
    };
    for (int i = 0; i < 5; ++i) launch();
    CK(hipDeviceSynchronize());
    const int reps = 30;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const char* names[3] = {"persistent grid-stride ", "persistent ordered tick", "one tile per workgroup "};
    printf("tile %2d KB halo %3d %s (%2d counters) grid %2d/CU resident %d/CU xch %d ops %3d: %.3f ms %5.0f GB/s (outputs x 16 B)\n", ROWS * 2, HALO,
           names[MODE], MODE == 1 ? NC : 0, MODE == 2 ? 0 : gpc, wg_per_cu, xch, ops, ms, 2.0 * tiles * L * sizeof(f2) / ms / 1e6);
    fflush(stdout);
}

int main() {
    const long long n = 1LL << 28;
    f2 *in, *out;
    unsigned* counter;
    CK(hipMalloc(&in, n * sizeof(f2)));
    CK(hipMalloc(&out, n * sizeof(f2)));
    CK(hipMalloc(&counter, 64 * 256));
    CK(hipMemset(out, 0, n * sizeof(f2)));
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (unsigned*)in, 2 * n);
    CK(hipDeviceSynchronize());
    printf("# round 3: counters 2 / 3 / 5 / 6, 32 KB tiles, three workgroups per CU, random floats; copy only, then with stand-in work\n");
    for (int rep = 0; rep < 2; ++rep) {
        run<16, 128, 1, 2>(in, out, counter, n, 3, 3, 0, 0);
        run<16, 128, 1, 3>(in, out, counter, n, 3, 3, 0, 0);
        run<16, 128, 1, 5>(in, out, counter, n, 3, 3, 0, 0);
        run<16, 128, 1, 6>(in, out, counter, n, 3, 3, 0, 0);
        run<16, 128, 1, 2>(in, out, counter, n, 3, 3, 2, 688);
        run<16, 128, 1, 3>(in, out, counter, n, 3, 3, 2, 688);
        run<16, 128, 1, 5>(in, out, counter, n, 3, 3, 2, 688);
        run<16, 128, 1, 2>(in, out, counter, n, 2, 2, 0, 0);
        run<16, 128, 1, 3>(in, out, counter, n, 2, 2, 0, 0);
        run<16, 128, 1, 2>(in, out, counter, n, 2, 2, 2, 688);
        run<16, 128, 1, 3>(in, out, counter, n, 2, 2, 2, 688);
    }
    CK(hipFree(in));
    CK(hipFree(out));
    return 0;
}
