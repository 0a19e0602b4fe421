// gen_device.h -- internal: the reference generator's per-sample arithmetic as device
// functions, shared by the stand-alone generator kernel (gen_kernels.hip) and by the
// FIR kernels that synthesise their own input (fir_fft.hip, fused gen -> FIR chain).
//
// The reference evaluates, per block k of B samples and in-block index i (an int
// converted to float, libs/kfr/base/basic_expressions.hpp:112-116), all in fp32
// without fused multiply-add (SSE2 build):
//   real:    A * wave(fract(phi_k [+0.25 for cosine] + cps*float(i)))
//            sine/cosine: wave(t) = fastsin(t * 2pi), a 7th-order odd polynomial
//            (libs/kfr/base/sin_cos.hpp:183-206) -- NOT sinf: it differs from a true
//            sine by 1.6e-5, more than the 1e-5 tolerance, so it is restated here.
// Every function switches FMA contraction off for its own body: HIP's
// __fmul_rn/__fadd_rn are plain operators that the compiler would otherwise contract
// into v_fma_f32, and the including file may be compiled with contraction on.
#pragma once
#include <hip/hip_runtime.h>

namespace sdr_amd {

// Generator as an input source of a FIR kernel (real samples only).
// Where the generator stands, in DEVICE memory, when its phase recurrence has a closed form (sdr_capi.hip,
// phase_chain_is_exact: phase and increment whole multiples of 2^-23, so no step of phi' = fract(phi + incr)
// ever rounds): the next block's start phase in units of 2^-23.  Kernels read it, the last workgroup of a call
// moves it on -- nothing about the generator's position is host state then, so a captured graph of one
// generating call replays as the stream's next call.
struct GenDevState {
    unsigned phase; // P: the next block starts at phase P / 2^23
    unsigned done;  // stand-alone generator kernel: workgroups of the running launch that have finished
};

struct GenSrc {
    const float* phi;     // start phase of blocks 0..blocks of this call (one spare entry at the end); unused with pstate
    GenDevState* pstate;  // non-NULL: block b of this call starts at phase ((pstate->phase + b * pincr) mod 2^23) / 2^23
    unsigned pincr;       // the per-block increment in units of 2^-23
    unsigned nblocks;     // blocks of this call: what the call's last workgroup advances pstate->phase by
    float phi_incr;       // phase advance per block, in [0, 1): phi[k+1] = phi[k] + phi_incr, less 1 if >= 1 (fp32)
    long long block_size; // B, 256 <= B <= 2^23
    double inv_block;     // 1.0 / B
    float amp;
    float cps;            // cycles per sample
    int waveform;         // 0 cosine 1 sine 2 square 3 triangle 4 sawtooth
    int enabled;
};

// Plain operators under `fp contract(off)`: each is one correctly rounded fp32
// operation (v_mul_f32 / v_add_f32 / v_sub_f32), never a v_fma_f32.
__device__ __forceinline__ float fract_f(float x) {
#pragma clang fp contract(off)
    return x - floorf(x);
}

__device__ __forceinline__ float fastsin_f(float x) {
#pragma clang fp contract(off)
    const float c2 = -0.16665853559970855712890625f;
    const float c4 = +8.31427983939647674560546875e-3f;
    const float c6 = -1.85423981747590005397796630859375e-4f;
    const float pi = 3.14159265358979323846f;
    const float xx = x - pi;
    float y = fabsf(xx);
    y = y > 1.57079632679489661923f ? pi - y : y;
    y = xx >= 0.0f ? -y : y; // sign bit of xx clear => negate (xx = +0 included)
    const float y2 = y * y;
    const float y3 = y2 * y;
    float f = c6 * y2;
    f = f + c4;
    f = f * y2;
    f = f + c2;
    f = f * y3;
    return f + y;
}

// sine and cosine (the caller has added 0.25 for cosine)
__device__ __forceinline__ float wave_sine(float arg) {
#pragma clang fp contract(off)
    const float two_pi = 6.28318530717958647692f;
    const float t = fract_f(arg) * two_pi;
    return fastsin_f(t);
}

__device__ __forceinline__ float wave_real(int waveform, float arg) {
#pragma clang fp contract(off)
    switch (waveform) {
        case 0:
        case 1: return wave_sine(arg);
        case 2: return fract_f(arg) < 0.5f ? 1.0f : -1.0f; // oscillators.hpp:64-73
        case 3: {
            // :107-116, 1 - |4 fract(x + 0.25) - 2| as the reference's release build
            // evaluates it: 4 (x - floor(x + 0.25)) - 1 (pinned by tests/golden/gen.npz)
            const float fl = floorf(arg + 0.25f);
            const float d = arg - fl;
            const float t = 4.0f * d;
            const float u = t - 1.0f;
            return 1.0f - fabsf(u);
        }
        default: {                                          // :80-89
            const float t = 2.0f * fract_f(arg);
            return 1.0f - t;
        }
    }
}

// One real sample of block phase `phi`, in-block index i (blocks/gen.cpp:183-195).
// The in-block index arrives as a float (an exact integer below 2^24).
__device__ __forceinline__ float gen_real_sample_f(int waveform, float amp, float cps, float phi, float i) {
#pragma clang fp contract(off)
    const float base = waveform == 0 ? phi + 0.25f : phi;
    const float step = cps * i;
    const float arg = base + step;
    return amp * wave_real(waveform, arg);
}
__device__ __forceinline__ float gen_real_sample(int waveform, float amp, float cps, float phi, int i) {
    return gen_real_sample_f(waveform, amp, cps, phi, (float)i);
}

// Two sine / cosine samples at once (the re and im members of a real pair), in-block
// indices given as floats (exact small integers), `base` = block phase (+0.25 for
// cosine).  The same roundings in the same order as gen_real_sample's sine on each component:
// vector operands let the compiler issue v_pk_mul_f32 / v_pk_add_f32, which round each
// half exactly like the scalar instructions; the fold y > pi/2 ? pi - y : y is
// min(y, pi - y) on [0, pi]; and because the polynomial is odd (every operation
// commutes with a sign flip) it is evaluated on |y| and the sign -- minus the sign of
// xx -- is put back at the end, through copysign and a multiplication by -A.
typedef float gen_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gen_f2 gen_sine_pair(float amp, float cps, gen_f2 base, gen_f2 idx) {
#pragma clang fp contract(off)
    const float c2 = -0.16665853559970855712890625f;
    const float c4 = +8.31427983939647674560546875e-3f;
    const float c6 = -1.85423981747590005397796630859375e-4f;
    const float pi = 3.14159265358979323846f;
    const float two_pi = 6.28318530717958647692f;
    const gen_f2 step = idx * cps;
    const gen_f2 arg = base + step;
    const gen_f2 fl = {floorf(arg.x), floorf(arg.y)};
    const gen_f2 fr = arg - fl;
    const gen_f2 t = fr * two_pi;
    const gen_f2 xx = t - pi;
    const gen_f2 y = {fminf(fabsf(xx.x), pi - fabsf(xx.x)), fminf(fabsf(xx.y), pi - fabsf(xx.y))};
    const gen_f2 y2 = y * y;
    const gen_f2 y3 = y2 * y;
    gen_f2 f = y2 * c6;
    f = f + c4;
    f = f * y2;
    f = f + c2;
    f = f * y3;
    f = f + y;
    const gen_f2 r = {__builtin_copysignf(f.x, xx.x), __builtin_copysignf(f.y, xx.y)};
    return r * -amp;
}

// The same two samples for the FIR kernels that synthesise their own input and hand it to an fp32 transform
// (fir_fft.hip, GEN): cheaper, and NOT the same bits.  What has to be the reference's own rounding is the phase
// argument base + cps * float(i): half an ulp of it is up to 1e-5 of a cycle, i.e. several times the parity bar,
// so the product and the sum stay two correctly rounded operations.  Everything behind the fract is well
// conditioned -- one rounding more or less moves the sample by ~1e-7 of the amplitude, below what the transform
// that consumes it adds -- and is evaluated with fused multiply-adds: v_fract_f32 (equal to x - floor(x) except
// where that rounds up to 1.0, where it returns the float below), the shift to [-pi, pi) as one fma, the odd
// polynomial y + y^3 (c2 + c4 y^2 + c6 y^4) in Horner form, sign and amplitude folded into one factor.
// 13 operations per sample where gen_sine_pair needs 18 (profiles/r03_gen_fast.txt).
__device__ __forceinline__ gen_f2 gen_sine_pair_fast(float amp, float cps, gen_f2 base, gen_f2 idx) {
    const float c2 = -0.16665853559970855712890625f;
    const float c4 = +8.31427983939647674560546875e-3f;
    const float c6 = -1.85423981747590005397796630859375e-4f;
    const float pi = 3.14159265358979323846f;
    const float two_pi = 6.28318530717958647692f;
    gen_f2 arg;
    {
#pragma clang fp contract(off)
        const gen_f2 step = idx * cps;
        arg = base + step;
    }
    const gen_f2 fr = {__builtin_amdgcn_fractf(arg.x), __builtin_amdgcn_fractf(arg.y)};
    const gen_f2 xx = __builtin_elementwise_fma(fr, gen_f2{two_pi, two_pi}, gen_f2{-pi, -pi});
    const gen_f2 y = {fminf(fabsf(xx.x), pi - fabsf(xx.x)), fminf(fabsf(xx.y), pi - fabsf(xx.y))};
    const gen_f2 y2 = y * y;
    gen_f2 f = __builtin_elementwise_fma(y2, gen_f2{c6, c6}, gen_f2{c4, c4});
    f = __builtin_elementwise_fma(f, y2, gen_f2{c2, c2});
    const gen_f2 g = f * y2;
    const gen_f2 ys = {__builtin_copysignf(y.x, xx.x), __builtin_copysignf(y.y, xx.y)};
    const gen_f2 t = ys * -amp;
    return __builtin_elementwise_fma(g, t, t);
}

// Start phase of block b of the call: exact integer arithmetic (every multiple of 2^-23 below 1 is a float, and
// the scaling by a power of two is exact) or the host's table -- the same bits either way.
__device__ __forceinline__ float gen_phase_closed(unsigned P0, unsigned pincr, unsigned b) {
    return (float)((P0 + b * pincr) & 0x7fffffu) * (1.0f / 8388608.0f);
}
__device__ __forceinline__ float gen_phase(const GenSrc& g, unsigned b) {
    if (g.pstate) return gen_phase_closed(g.pstate->phase, g.pincr, b);
    return g.phi[b];
}

// Sample m (>= 0, counted from the first sample of the call) of a generator source:
// block m / B, index m % B.  The quotient comes from a double multiply with one
// correction step (m < 2^52).
__device__ __forceinline__ float gen_src_at(const GenSrc& g, long long m) {
    long long b = (long long)((double)m * g.inv_block);
    long long r = m - b * g.block_size;
    if (r < 0) { r += g.block_size; --b; }
    else if (r >= g.block_size) { r -= g.block_size; ++b; }
    return gen_real_sample(g.waveform, g.amp, g.cps, gen_phase(g, (unsigned)b), (int)r);
}

} // namespace sdr_amd
