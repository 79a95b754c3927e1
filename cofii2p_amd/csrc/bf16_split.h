// fp32 -> bf16 hi + bf16 lo (round to nearest even both times): x ~ hi + lo with ~2^-17 relative error.  The 3-term product
// hi*hi + hi*lo + lo*hi on the bf16 matrix cores then carries ~2^-16 per product (gemm.hip, transformer_tail.hip, kpconv.hip).
#pragma once
#include "common.h"

typedef __bf16 cofi_bf16x8 __attribute__((ext_vector_type(8)));
union CofiFrag { uint4 u; cofi_bf16x8 v; };

static __device__ __forceinline__ unsigned cofi_cvt_pk_bf16(float a, float b) {  // RNE, a -> low half
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
static __device__ __forceinline__ void cofi_split2(float a, float b, unsigned &hi, unsigned &lo) {
    hi = cofi_cvt_pk_bf16(a, b);
    lo = cofi_cvt_pk_bf16(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}
// fp32 -> three bf16 planes (RNE each time): x == hi + mid + lo exactly for finite fp32 (3 x 8 significant bits).  The 6-term product
// hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi is the fp32-grade "bf16x6" arithmetic (gemm.hip split4x3: the same rounding).
static __device__ __forceinline__ void cofi_split4x3(const f32x4 v, uint2 &hi, uint2 &mid, uint2 &lo) {
    // two-wide float arithmetic (v_pk_add_f32): 18 instructions per four values
    const f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
    hi.x = cofi_cvt_pk_bf16(a[0], a[1]);
    hi.y = cofi_cvt_pk_bf16(b[0], b[1]);
    const f32x2 ha = {__uint_as_float(hi.x << 16), __uint_as_float(hi.x & 0xffff0000u)}, hb = {__uint_as_float(hi.y << 16), __uint_as_float(hi.y & 0xffff0000u)};
    const f32x2 ra = a - ha, rb = b - hb;
    mid.x = cofi_cvt_pk_bf16(ra[0], ra[1]);
    mid.y = cofi_cvt_pk_bf16(rb[0], rb[1]);
    const f32x2 ma = {__uint_as_float(mid.x << 16), __uint_as_float(mid.x & 0xffff0000u)}, mb = {__uint_as_float(mid.y << 16), __uint_as_float(mid.y & 0xffff0000u)};
    const f32x2 sa = ra - ma, sb = rb - mb;
    lo.x = cofi_cvt_pk_bf16(sa[0], sa[1]);
    lo.y = cofi_cvt_pk_bf16(sb[0], sb[1]);
}
