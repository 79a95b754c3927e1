// Wave-level (64 lanes) top-k machinery shared by the brute-force (knn.hip) and the cell-grid (knn_grid.hip) searches:
// canonical fp32 distance, 64-bit (distance bits, index) keys, bitonic sort / merge of one key per lane.
// The compare-exchange partners come from DPP row operations (xor 1, 2, 4, 8 and the in-row mirror) and the gfx950 lane-swap
// instructions v_permlane16_swap / v_permlane32_swap (xor 16, 32): plain VALU latency instead of a trip through the LDS
// crossbar (ds_bpermute) per step — a flush of the staging buffer is 39 dependent compare-exchange steps.
#pragma once
#include "common.h"

namespace {

typedef unsigned long long u64;
constexpr u64 KEY_INF = 0x7f8000007fffffffull;  // (+inf, INT_MAX)

__device__ __forceinline__ float canon_sqnorm(float x, float y, float z) { return (x * x + y * y) + z * z; }
__device__ __forceinline__ float canon_dist(float qx, float qy, float qz, float qq, float sx, float sy, float sz, float ss) {
    const float dot = fmaf(qz, sz, fmaf(qy, sy, qx * sx));
    const float d = ((-2.0f * dot) + qq) + ss;
    return d < 1e-12f ? 1e-12f : d;
}

__device__ __forceinline__ u64 umin64(u64 a, u64 b) { return a < b ? a : b; }
__device__ __forceinline__ u64 umax64(u64 a, u64 b) { return a < b ? b : a; }

template <int CTRL>
__device__ __forceinline__ unsigned dpp_mov(unsigned v) {
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);   // every source lane is in range: no `old` value needed
}
// per-lane select by a wave-uniform 64-bit lane mask held in scalar registers: mask bit set -> b, else a (one v_cndmask, no
// per-lane predicate arithmetic: every mask of the sorting network is a compile-time constant or a compare result)
__device__ __forceinline__ unsigned sel_mask(unsigned a, unsigned b, u64 mask) {
    unsigned r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
    return r;
}
constexpr u64 lanes_with_bit(int bit) {   // lanes whose index has `bit` set
    u64 m = 0;
    for (int l = 0; l < 64; ++l)
        if (l & bit) m |= 1ull << l;
    return m;
}
// value of lane (lane ^ J), J a power of two
template <int J>
__device__ __forceinline__ unsigned lane_xor32(unsigned v) {
    if constexpr (J == 1) return dpp_mov<0xB1>(v);                        // quad_perm [1,0,3,2]
    else if constexpr (J == 2) return dpp_mov<0x4E>(v);                   // quad_perm [2,3,0,1]
    else if constexpr (J == 4) return dpp_mov<0x1B>(dpp_mov<0x141>(v));   // row_half_mirror (i -> 7-i), then quad reverse (i -> i^3): i -> i^4
    else if constexpr (J == 8) return dpp_mov<0x128>(v);                  // row_ror:8
    else if constexpr (J == 16) {
        // v_permlane16_swap(a, b): odd rows of a <-> even rows of b.  a = b = v: a' = {r0,r0,r2,r2}, b' = {r1,r1,r3,r3}
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return sel_mask(r[1], r[0], lanes_with_bit(16));
    } else {
        static_assert(J == 32, "power of two below 64");
        // v_permlane32_swap(a, b): upper half of a <-> lower half of b.  a = b = v: a' = {lo,lo}, b' = {hi,hi}
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return sel_mask(r[1], r[0], lanes_with_bit(32));
    }
}
template <int J>
__device__ __forceinline__ u64 lane_xor64(u64 v, int) {
    return ((u64)lane_xor32<J>((unsigned)(v >> 32)) << 32) | lane_xor32<J>((unsigned)v);
}
// value of lane (63 - lane): in-row mirror, then swap the rows pairwise and the halves
__device__ __forceinline__ unsigned lane_rev32(unsigned v) { return lane_xor32<32>(lane_xor32<16>(dpp_mov<0x140>(v))); }
__device__ __forceinline__ u64 lane_rev64(u64 v, int) { return ((u64)lane_rev32((unsigned)(v >> 32)) << 32) | lane_rev32((unsigned)v); }
// value of lane `src` (wave-uniform) in every lane
__device__ __forceinline__ u64 lane_bcast64(u64 v, int src) {
    return ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src);
}

// One compare-exchange step of a bitonic network: lane l keeps min(v, partner) if bit l of MINMASK is set, max otherwise.
// lt = lanes whose partner is smaller; a lane takes the partner's key iff (partner smaller) == (lane keeps the minimum):
// one 64-bit compare into a scalar mask, one scalar xnor with the constant, two selects.
template <int J, u64 MINMASK>
__device__ __forceinline__ u64 exchange_step(u64 v) {
    const unsigned vlo = (unsigned)v, vhi = (unsigned)(v >> 32);
    const unsigned plo = lane_xor32<J>(vlo), phi = lane_xor32<J>(vhi);
    const u64 p = ((u64)phi << 32) | plo;
    const u64 take = ~(__ballot(p < v) ^ MINMASK);
    return ((u64)sel_mask(vhi, phi, take) << 32) | sel_mask(vlo, plo, take);
}
template <int K, int J>
constexpr u64 sort_minmask() {   // ascending runs where (lane & K) == 0: the lower lane of a pair keeps the minimum there
    u64 m = 0;
    for (int l = 0; l < 64; ++l)
        if (((l & J) == 0) == ((l & K) == 0)) m |= 1ull << l;
    return m;
}
template <int K, int J>
__device__ __forceinline__ u64 sort_stage(u64 v) {
    v = exchange_step<J, sort_minmask<K, J>()>(v);
    if constexpr (J > 1) v = sort_stage<K, J / 2>(v);
    return v;
}
// ascending bitonic sort of one key per lane
__device__ __forceinline__ u64 wave_sort(u64 v, int) {
    v = sort_stage<2, 1>(v);
    v = sort_stage<4, 2>(v);
    v = sort_stage<8, 4>(v);
    v = sort_stage<16, 8>(v);
    v = sort_stage<32, 16>(v);
    v = sort_stage<64, 32>(v);   // (lane & 64) == 0 everywhere: the last stage sorts ascending
    return v;
}
template <int J>
__device__ __forceinline__ u64 merge_stage(u64 v) {
    v = exchange_step<J, ~lanes_with_bit(J)>(v);
    if constexpr (J > 1) v = merge_stage<J / 2>(v);
    return v;
}
// sorts a bitonic sequence held one key per lane into ascending order
__device__ __forceinline__ u64 wave_bitonic_merge(u64 v, int) { return merge_stage<32>(v); }

// Sorted best-128 of a query across the wave (rank lane in l0, rank 64+lane in l1) + the admission threshold tau = 128th key.
struct Best128 {
    u64 l0 = KEY_INF, l1 = KEY_INF, tau = KEY_INF;
    // merge n <= 64 unsorted staged keys (lane i holds the i-th, or KEY_INF)
    __device__ __forceinline__ void merge(u64 b, int lane) {
        b = wave_sort(b, lane);
        // 64 smallest of (l1 U b): min(l1[i], b[63-i]) is bitonic
        u64 t = umin64(l1, lane_rev64(b, lane));
        t = wave_bitonic_merge(t, lane);
        // merge sorted l0 with sorted t (128 keys): low / high halves are each bitonic
        const u64 tr = lane_rev64(t, lane);
        const u64 lo = umin64(l0, tr), hi = umax64(l0, tr);
        l0 = wave_bitonic_merge(lo, lane);
        l1 = wave_bitonic_merge(hi, lane);
        tau = lane_bcast64(l1, 63);
    }
    __device__ __forceinline__ u64 kth(int k) const { return k <= 64 ? lane_bcast64(l0, k - 1) : lane_bcast64(l1, k - 65); }
    __device__ __forceinline__ void emit(int q, int k, int S, int lane, int32_t *out_idx, float *out_dist) const {
        if (lane < k) {
            const int id = (int)(unsigned)(l0 & 0xffffffffu);
            out_idx[(size_t)q * k + lane] = id == 0x7fffffff ? S : id;
            if (out_dist) out_dist[(size_t)q * k + lane] = __uint_as_float((unsigned)(l0 >> 32));
        }
        if (64 + lane < k) {
            const int id = (int)(unsigned)(l1 & 0xffffffffu);
            out_idx[(size_t)q * k + 64 + lane] = id == 0x7fffffff ? S : id;
            if (out_dist) out_dist[(size_t)q * k + 64 + lane] = __uint_as_float((unsigned)(l1 >> 32));
        }
    }
};

}  // namespace
