// fft_math.h -- internal: packed-fp32 complex arithmetic shared by the overlap-save kernels
// (fir_fft.hip: 256 lanes x 16 points; fir_fft8.hip: 512 lanes x 8 points).  All complex arithmetic is
// v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 with operand modifiers doing the +-i rotations and swaps.
#pragma once
#include <hip/hip_runtime.h>

// cache policy of the streaming loads / stores (experiments: -DSDR_LD_NT=0 / -DSDR_ST_NT=0)
#ifndef SDR_LD_NT
#define SDR_LD_NT 1
#endif
#ifndef SDR_ST_NT
#define SDR_ST_NT 1
#endif

namespace sdr_amd {

typedef float f2 __attribute__((ext_vector_type(2)));

template <class T>
__device__ __forceinline__ T stream_load(const T* p) {
#if SDR_LD_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
template <class T>
__device__ __forceinline__ void stream_store(T v, T* p) {
#if defined(SDR_ST_SCOPE)
    // experiment: relaxed atomic stores carry the scope bits (workgroup sc0, agent sc1, system sc0 sc1), no nt
    if constexpr (sizeof(T) == 8) __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, SDR_ST_SCOPE);
    else __hip_atomic_store(reinterpret_cast<unsigned*>(p), __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, SDR_ST_SCOPE);
#elif SDR_ST_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

namespace {

// a + (-i) b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ f2 add_mi(f2 a, f2 b) {
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a + (+i) b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ f2 add_pi(f2 a, f2 b) {
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// x * w : two packed ops (broadcast / swap / negate are operand modifiers)
__device__ __forceinline__ f2 cmul(f2 x, f2 w) {
    f2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(x), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(x), "v"(w), "v"(t));
    return r;
}
// x * conj(w)
__device__ __forceinline__ f2 cmulc(f2 x, f2 w) {
    f2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(x), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(x), "v"(w), "v"(t));
    return r;
}
// x * (c, s) with a compile-time constant held in an SGPR pair
__device__ __forceinline__ f2 cmul_k(f2 x, float c, float s) {
    const f2 w = {c, s};
    f2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(x), "s"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(x), "s"(w), "v"(t));
    return r;
}

template <bool INV>
__device__ __forceinline__ void radix4(f2& x0, f2& x1, f2& x2, f2& x3) {
    const f2 s0 = x0 + x2, s1 = x0 - x2, s2 = x1 + x3, s3 = x1 - x3;
    x0 = s0 + s2;
    x2 = s0 - s2;
    if (!INV) { x1 = add_mi(s1, s3); x3 = add_pi(s1, s3); }
    else      { x1 = add_pi(s1, s3); x3 = add_mi(s1, s3); }
}


template <int V> struct IC { static constexpr int value = V; };
struct NoHook { __device__ __forceinline__ void operator()() const {} };

} // namespace
} // namespace sdr_amd
