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
// 22 instructions per four values.  (The subtractions two-wide - v_pk_add_f32, 18 instructions - measured SLOWER on MI355X: the
// attention kernel 124.8 vs 113.9 us per batch-16 cross launch, the whole forward 698 -> 683 frames/s with it in the GEMM loaders.)
static __device__ __forceinline__ void cofi_split4x3(const f32x4 v, uint2 &hi, uint2 &mid, uint2 &lo) {
    hi.x = cofi_cvt_pk_bf16(v[0], v[1]);
    hi.y = cofi_cvt_pk_bf16(v[2], v[3]);
    const float rx = v[0] - __uint_as_float(hi.x << 16), ry = v[1] - __uint_as_float(hi.x & 0xffff0000u);
    const float rz = v[2] - __uint_as_float(hi.y << 16), rw = v[3] - __uint_as_float(hi.y & 0xffff0000u);
    mid.x = cofi_cvt_pk_bf16(rx, ry);
    mid.y = cofi_cvt_pk_bf16(rz, rw);
    lo.x = cofi_cvt_pk_bf16(rx - __uint_as_float(mid.x << 16), ry - __uint_as_float(mid.x & 0xffff0000u));
    lo.y = cofi_cvt_pk_bf16(rz - __uint_as_float(mid.y << 16), rw - __uint_as_float(mid.y & 0xffff0000u));
}
