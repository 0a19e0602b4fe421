// valubench.hip -- round 2: what does a packed fp32 VALU instruction cost on gfx950?  The overlap-save transform is
// ~92 % v_pk_{add,mul,fma}_f32; its loop saturates at two workgroups per CU (profiles/r02_fft_variants.txt).
// Each wave runs a loop of 32 independent instructions of one kind; 1..4 waves per SIMD; cycles per instruction
// and SIMD from the wall time and s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int KIND>
__global__ __launch_bounds__(256) void spin(float* out, int iters, long long* cyc) {
    f2 a[16];
    const float s = 1.0f + 1e-7f * threadIdx.x;
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = f2{s + i, s - i};
    const f2 m = {0.999f, 1.001f}, c = {1e-3f, -1e-3f};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
                if (KIND == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x));
                if (KIND == 5) { // two unpacked instructions doing a packed one's work
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].y) : "v"(m.y), "v"(c.y));
                }
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += a[i].x + a[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND>
static void run(const char* name, float* out, long long* cyc, int wg_per_cu) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(spin<KIND>, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, 100, cyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(spin<KIND>, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, iters, cyc);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    long long h;
    CK(hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    const double per_wave = (KIND == 5 ? 64.0 : 32.0) * iters;          // instructions per wave
    const double inst_simd = per_wave * wg_per_cu;                      // per SIMD (one wave of each workgroup)
    printf("%-14s %d waves/SIMD: %.3f ms, %.2f ns per instruction and SIMD, shader clock counter %lld -> %.2f counts per instruction of one wave\n",
           name, wg_per_cu, ms, ms * 1e6 / inst_simd, h, (double)h / per_wave);
    fflush(stdout);
}

int main() {
    float* out;
    long long* cyc;
    CK(hipMalloc(&out, 256 * 256 * 8 * sizeof(float)));
    CK(hipMalloc(&cyc, 8));
    for (int w = 1; w <= 4; ++w) {
        run<0>("v_pk_fma_f32", out, cyc, w);
        run<1>("v_pk_add_f32", out, cyc, w);
        run<2>("v_pk_mul_f32", out, cyc, w);
        run<3>("v_fma_f32", out, cyc, w);
        run<4>("v_add_f32", out, cyc, w);
        run<5>("2 x v_fma_f32", out, cyc, w);
    }
    return 0;
}
