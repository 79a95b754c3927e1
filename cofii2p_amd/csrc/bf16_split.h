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
