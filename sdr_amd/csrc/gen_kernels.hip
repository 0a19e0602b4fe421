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

#include "gen_device.h"
#include "gen_kernels.h"

namespace sdr_amd {

// grid.x = packet (block) index, grid.y * 256 lanes cover the in-block index: no
// division on the per-sample path, coalesced 4/8-byte stores.
__global__ __launch_bounds__(256) void gen_blocks(GenArgs a) {
    const int i = blockIdx.y * 256 + threadIdx.x;
    const unsigned P0 = a.pstate ? a.pstate->phase : 0u;
    for (long long b = blockIdx.x; i < a.block_size && b < a.blocks; b += gridDim.x) {
        const float phi = a.pstate ? gen_phase_closed(P0, a.pincr, (unsigned)b) : a.phi[b];
        const float step = __fmul_rn(a.cps, (float)i);
        const long long s = b * a.block_size + i;
        if (!a.complex_samples) {
            a.out[s] = gen_real_sample(a.waveform, a.amp, a.cps, phi, i);
        } else {
            const float base = __fadd_rn(phi, a.waveform == 1 ? -0.25f : 0.0f);
            const float ang = __fmul_rn(6.28318530717958647692f, __fadd_rn(base, step));
            float sn, cs;
            sincosf(ang, &sn, &cs);
            reinterpret_cast<float2*>(a.out)[s] = make_float2(__fmul_rn(a.amp, cs), __fmul_rn(a.amp, sn));
        }
    }
    if (a.pstate) {
        // the launch's last workgroup to finish moves the generator on (every wave has read the phase by now)
        __syncthreads();
        if (threadIdx.x == 0 &&
            __hip_atomic_fetch_add(&a.pstate->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == gridDim.x * gridDim.y) {
            a.pstate->done = 0;
            a.pstate->phase = (P0 + (unsigned)a.blocks * a.pincr) & 0x7fffffu;
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

__global__ __launch_bounds__(256) void gen_interleave(float2* __restrict__ out, const float* __restrict__ re,
                                                      const float* __restrict__ q, float scale, long long n) {
    for (long long s = (long long)blockIdx.x * 256 + threadIdx.x; s < n; s += (long long)gridDim.x * 256)
        out[s] = make_float2(re[s], __fmul_rn(scale, q[s]));
}

hipError_t launch_gen_interleave(float* out, const float* re, const float* q, float scale, long long n, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    long long grid = (n + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(gen_interleave, dim3((unsigned)grid), dim3(256), 0, stream, reinterpret_cast<float2*>(out), re, q, scale, n);
    return hipGetLastError();
}

} // namespace sdr_amd
