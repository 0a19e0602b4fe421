// membench12.hip -- round 2: would a 512-lane, 8-points-per-lane transform at TWO workgroups per CU carry the
// 5.85 TB/s that two persistent 32 KB-tile workgroups per CU copy (r02_membench_residency.txt)?  Ticketed persistent
// loop on random data; per tile and lane: `ops` packed FMAs in 8 dependent chains, `xb` workgroup-wide exchanges
// (barrier) and `xw` wave-local ones (no barrier), 8 LDS writes + 8 reads each; 256-lane / 16-point version
// (three per CU: the shape of fir_fft4096) beside it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void fill_random(unsigned* p, long long nwords) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (long long)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ (unsigned)(i >> 32) * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = __float_as_uint((float)(h >> 8) * (2.0f / 16777216.0f) - 1.0f);
    }
}

template <int WG, int PER_CU>
__global__ __launch_bounds__(WG, WG == 512 ? 2 : 3) void loop(const f2* __restrict__ in, f2* __restrict__ out, long long tiles,
                                                             unsigned* counter, int ops, int xb, int xw) {
    constexpr int ROWS = 4096 / WG, HALO = 128, NC = 2;
    constexpr int PITCH = WG + 16;
    __shared__ f2 lds[(160 * 1024 / PER_CU - 1024) / 8];
    __shared__ unsigned ticket_box;
    const int tid = threadIdx.x;
    constexpr long long L = 4096 - HALO;
    const int bid = blockIdx.x, nblk = gridDim.x;
    long long t = bid;
    if (t >= tiles) return;
    long long tn = t + nblk;
    f2 nx[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; ++k) nx[k] = __builtin_nontemporal_load(&in[t * L + WG * k + tid]);
    const f2 w = {0.999f, 0.001f * (tid & 7)};
    while (true) {
        f2 v[ROWS];
#pragma unroll
        for (int k = 0; k < ROWS; ++k) v[k] = nx[k];
        const bool more = tn < tiles;
        unsigned ticket = 0;
        const int phases = xb + xw + 1;
        for (int ph = 0; ph < phases; ++ph) {
            for (int o = 0; o < ops / phases; o += ROWS) {
#pragma unroll
                for (int k = 0; k < ROWS; ++k) v[k] = __builtin_elementwise_fma(v[k], w, v[(k + 1) % ROWS]);
            }
            if (ph == 0) {
                if (tid == 0) ticket = atomicAdd(counter + 64 * (bid % NC), 1u);
                if (more) {
#pragma unroll
                    for (int k = 0; k < ROWS; ++k) nx[k] = __builtin_nontemporal_load(&in[tn * L + WG * k + tid]);
                }
            }
            if (ph < xb) { // workgroup-wide exchange
#pragma unroll
                for (int k = 0; k < ROWS; ++k) lds[k * PITCH + tid] = v[k];
                if (ph == xb - 1 && tid == 0) ticket_box = ticket;
                __syncthreads();
#pragma unroll
                for (int k = 0; k < ROWS; ++k) v[k] = lds[(tid / (WG / ROWS)) * PITCH + (WG / ROWS) * k + (tid % (WG / ROWS))];
            } else if (ph < xb + xw) { // inside 8- or 16-lane groups, no barrier
                f2* g = lds + (tid / ROWS) * (ROWS * (ROWS + 1));
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < ROWS; ++k) g[k * (ROWS + 1) + tid % ROWS] = v[k];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < ROWS; ++k) v[k] = g[(tid % ROWS) * (ROWS + 1) + k];
            }
        }
#pragma unroll
        for (int k = HALO / WG; k < ROWS; ++k) {
            if (k == HALO / WG && HALO % WG != 0 && tid < HALO % WG) continue;
            __builtin_nontemporal_store(v[k], &out[t * L + WG * k + tid - HALO]);
        }
        if (!more) break;
        t = tn;
        if (xb == 0) { // no exchange published the ticket: do it here
            if (tid == 0) ticket_box = ticket;
            __syncthreads();
        }
        const unsigned tk = (unsigned)__builtin_amdgcn_readfirstlane((int)ticket_box); // unsigned: a ticket is never negative
        tn = 2LL * nblk + (long long)tk * NC + bid % NC;
        __syncthreads(); // the exchange buffer and the ticket box are free again
    }
}

template <int WG, int PER_CU>
static void run(const f2* in, f2* out, unsigned* counter, long long n, int ops, int xb, int xw) {
    constexpr long long L = 4096 - 128;
    const long long tiles = (n - 4096) / L + 1;
    auto fn = loop<WG, PER_CU>;
    const int grid = 256 * PER_CU;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto launch = [&]() {
        CK(hipMemsetAsync(counter, 0, 64 * 256, 0));
        hipLaunchKernelGGL(fn, dim3(grid), dim3(WG), 0, 0, in, out, tiles, counter, ops, xb, xw);
    };
    for (int i = 0; i < 8; ++i) launch();
    CK(hipDeviceSynchronize());
    const int reps = 40;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%3d lanes x %2d points, %d per CU, %4d packed ops, %d block + %d local exchanges: %.3f ms %5.0f GB/s\n", WG, 4096 / WG, PER_CU,
           ops, xb, xw, ms, 2.0 * tiles * L * sizeof(f2) / ms / 1e6);
    fflush(stdout);
}

int main() {
    const long long n = 1LL << 28;
    f2 *in, *out;
    unsigned* counter;
    CK(hipMalloc(&in, n * sizeof(f2)));
    CK(hipMalloc(&out, n * sizeof(f2)));
    CK(hipMalloc(&counter, 64 * 256));
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (unsigned*)in, 2 * n);
    CK(hipMemset(out, 0, n * sizeof(f2)));
    CK(hipDeviceSynchronize());
    for (int round = 0; round < 2; ++round) {
        run<256, 3>(in, out, counter, n, 0, 0, 0);
        run<512, 2>(in, out, counter, n, 0, 0, 0);
        run<256, 3>(in, out, counter, n, 688, 2, 2);   // fir_fft4096: 2 block-wide + 2 group exchanges, ~690 ops
        run<256, 2>(in, out, counter, n, 688, 2, 2);
        run<512, 2>(in, out, counter, n, 368, 2, 4);   // radix 8.8.8.8: per transform 1 block-wide + 2 wave-local
        run<512, 2>(in, out, counter, n, 368, 2, 2);
        run<512, 2>(in, out, counter, n, 368, 4, 2);
    }
    return 0;
}
