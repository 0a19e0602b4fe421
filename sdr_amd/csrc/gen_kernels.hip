// gen_kernels.hip -- on-device waveform generator restating the reference `gen`
// block's arithmetic (blocks/gen.cpp:117-132, 178-230) so that a fused
// gen -> hilbert -> FIR chain never has to bring generated samples over PCIe.
//
// The reference evaluates, per block k of B samples and in-block index i (an int
// converted to float, libs/kfr/base/basic_expressions.hpp:112-116), all in fp32
// without fused multiply-add (SSE2 build):
//   real:    A * wave(fract(phi_k [+0.25 for cosine] + cps*float(i)))
//            sine/cosine: wave(t) = fastsin(t * 2pi), a 7th-order odd polynomial
//            (libs/kfr/base/sin_cos.hpp:183-206) -- NOT sinf: it differs from a true
//            sine by 1.6e-5, more than the 1e-5 tolerance, so it is restated here;
//   complex: A * (cos(a), sin(a)),  a = 2pi_f * ((phi_k [-0.25 for sine]) + cps*float(i))
//            (no fract: the fp32 rounding of the angle is part of the result).
// phi_k follows the sequential fp32 recurrence phi_{k+1} = fract(phi_k + fract(cps*B));
// it is computed on the host and passed as a table.
// The reference build has no fused multiply-add: this file is compiled with
// -ffp-contract=off (Makefile) and repeats it here, because HIP's __fmul_rn/__fadd_rn
// are plain operators that the compiler would otherwise still contract into v_fma_f32.
#include <hip/hip_runtime.h>
#pragma clang fp contract(off)

#include "gen_kernels.h"

namespace sdr_amd {

__device__ __forceinline__ float fract_f(float x) { return __fsub_rn(x, floorf(x)); }

__device__ __forceinline__ float fastsin_f(float x) {
    const float c2 = -0.16665853559970855712890625f;
    const float c4 = +8.31427983939647674560546875e-3f;
    const float c6 = -1.85423981747590005397796630859375e-4f;
    const float pi = 3.14159265358979323846f;
    const float xx = __fsub_rn(x, pi);
    float y = fabsf(xx);
    y = y > 1.57079632679489661923f ? __fsub_rn(pi, y) : y;
    y = xx >= 0.0f ? -y : y; // sign bit of xx clear => negate (xx = +0 included)
    const float y2 = __fmul_rn(y, y);
    const float y3 = __fmul_rn(y2, y);
    float f = __fadd_rn(__fmul_rn(c6, y2), c4);
    f = __fadd_rn(__fmul_rn(f, y2), c2);
    return __fadd_rn(__fmul_rn(f, y3), y);
}

__device__ __forceinline__ float wave_real(int waveform, float arg) {
    const float two_pi = 6.28318530717958647692f;
    switch (waveform) {
        case 0: // cosine: caller already added 0.25
        case 1: return fastsin_f(__fmul_rn(fract_f(arg), two_pi));
        case 2: return fract_f(arg) < 0.5f ? 1.0f : -1.0f;                                   // oscillators.hpp:64-73
        case 3: return __fsub_rn(1.0f, fabsf(__fsub_rn(__fmul_rn(4.0f, fract_f(__fadd_rn(arg, 0.25f))), 2.0f))); // :107-116
        default: return __fsub_rn(1.0f, __fmul_rn(2.0f, fract_f(arg)));                      // :80-89
    }
}

// grid.x = packet (block) index, grid.y * 256 lanes cover the in-block index: no
// division on the per-sample path, coalesced 4/8-byte stores.
__global__ __launch_bounds__(256) void gen_blocks(GenArgs a) {
    const int i = blockIdx.y * 256 + threadIdx.x;
    if (i >= a.block_size) return;
    for (long long b = blockIdx.x; b < a.blocks; b += gridDim.x) {
        const float phi = a.phi[b];
        const float step = __fmul_rn(a.cps, (float)i);
        const long long s = b * a.block_size + i;
        if (!a.complex_samples) {
            const float base = a.waveform == 0 ? __fadd_rn(phi, 0.25f) : phi;
            a.out[s] = __fmul_rn(a.amp, wave_real(a.waveform, __fadd_rn(base, step)));
        } else {
            const float base = __fadd_rn(phi, a.waveform == 1 ? -0.25f : 0.0f);
            const float ang = __fmul_rn(6.28318530717958647692f, __fadd_rn(base, step));
            float sn, cs;
            sincosf(ang, &sn, &cs);
            reinterpret_cast<float2*>(a.out)[s] = make_float2(__fmul_rn(a.amp, cs), __fmul_rn(a.amp, sn));
        }
    }
}

hipError_t launch_gen(const GenArgs& a, hipStream_t stream) {
    if (a.blocks == 0 || a.block_size == 0) return hipSuccess;
    const unsigned gy = (unsigned)((a.block_size + 255) / 256);
    if (gy > 65535) return hipErrorInvalidValue;
    const long long gx = a.blocks < 2048 ? a.blocks : 2048; // each workgroup walks over many packets
    hipLaunchKernelGGL(gen_blocks, dim3((unsigned)gx, gy), dim3(256), 0, stream, a);
    return hipGetLastError();
}

} // namespace sdr_amd
