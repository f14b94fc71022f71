// Split-bf16 operands for v_mfma_f32_16x16x32_bf16 (internal; used by csrc/elg_bwd.hip and csrc/elg_fwd.hip).
// A float x is written as x = x1 + x2 (+ x3), every term a bf16: x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)
// (the residuals are exact in f32) -- 16 (24) significand bits.  A product of two such sums is a handful of bf16 MFMAs with
// f32 accumulation; with three terms what is dropped (a2 b3, a3 b2, a3 b3) is 2^-24 of the product, an f32 rounding.
#pragma once
#include <hip/hip_runtime.h>

namespace elg {

using f32x4_bf = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {       // v_cvt_pk_bf16_f32: a -> bits 0..15, b -> bits 16..31
    f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
// terms of 4 values: t1 = (p[0], p[1]), t2 = (p[2], p[3]), t3 = (p[4], p[5]); each word packs two values
template <int T>
__device__ __forceinline__ void bf_terms(float x0, float x1, float x2, float x3, unsigned (&p)[6]) {
    p[0] = pk_bf16(x0, x1); p[1] = pk_bf16(x2, x3);
    x0 -= bf_lo(p[0]); x1 -= bf_hi(p[0]); x2 -= bf_lo(p[1]); x3 -= bf_hi(p[1]);
    p[2] = pk_bf16(x0, x1); p[3] = pk_bf16(x2, x3);
    if (T >= 3) {
        x0 -= bf_lo(p[2]); x1 -= bf_hi(p[2]); x2 -= bf_lo(p[3]); x3 -= bf_hi(p[3]);
        p[4] = pk_bf16(x0, x1); p[5] = pk_bf16(x2, x3);
    } else {
        p[4] = 0u; p[5] = 0u;
    }
}
__device__ __forceinline__ u32x4 bf_single(float x0, float x1, float x2, float x3) {     // [x1 | x2]
    unsigned p[6];
    bf_terms<2>(x0, x1, x2, x3, p);
    return u32x4{p[0], p[1], p[2], p[3]};
}
__device__ __forceinline__ void bf_dup(float x0, float x1, float x2, float x3, u32x4& d1, u32x4& d2) {   // [x1 | x1], [x2 | x2]
    unsigned p[6];
    bf_terms<2>(x0, x1, x2, x3, p);
    d1 = u32x4{p[0], p[1], p[0], p[1]};
    d2 = u32x4{p[2], p[3], p[2], p[3]};
}
__device__ __forceinline__ f32x4_bf mfma_bf(u32x4 a, u32x4 b, f32x4_bf c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}


}  // namespace elg
