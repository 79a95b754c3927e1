// Layout of the attention partials (attention.hip) shared by the producer, the stand-alone merge kernel and the consumers
// that merge on the fly (transformer_tail.hip).
//
// The T = pairs * P units of a launch (pair = (frame, head, 32-query block), P = ceil(S / 32) key blocks) are numbered
// pair-major and cut into `nwg` contiguous ranges of U units, one per workgroup.  Workgroup w leaves, for every pair its range
// touches, ONE slot {m[32], l[32], O[32][32]} (running max, row sum, un-normalised output, all relative to m) at
//     parts[(pair * maxp + (w - first_wg(pair))) * COFI_ATTN_SLOT_FLOATS],      first_wg(pair) = pair * P / U,
// so a reader only needs (P, U, QB, maxp) - recomputed here from (L, S, H, frames) and the device's CU count - to find and
// combine the slots of a query row:  O = sum_s O_s 2^(m_s - m) / sum_s l_s 2^(m_s - m),  m = max_s m_s  (fixed order).
#pragma once
#include "common.h"

#define COFI_ATTN_SLOT_FLOATS 1088
#define COFI_ATTN_MAX_SEGMENTS 3   /* pairs one workgroup's range may touch: U <= 2 P + 1 */

struct AttnLayout {
    int P;      // key blocks per pair
    int QB;     // query blocks per (frame, head)
    int U;      // units per workgroup
    int T;      // units of the launch
    int maxp;   // slots reserved per pair
    int nwg;    // workgroups
    size_t bytes;
};

static inline int attn_num_cus() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) return v;
        (void)hipGetLastError();
        return 256;   // MI355X
    }();
    return n;
}

static inline AttnLayout attn_layout(int L, int S, int H, int frames) {
    AttnLayout a;
    a.P = cofi_cdiv(S, 32);
    a.QB = cofi_cdiv(L, 32);
    const long pairs = (long)frames * H * a.QB;
    a.T = (int)(pairs * a.P);
    int U = cofi_cdiv(a.T, attn_num_cus());   // one range per CU ...
    if (U > 2 * a.P + 1) U = 2 * a.P + 1;     // ... of at most COFI_ATTN_MAX_SEGMENTS pairs
    if (U < 1) U = 1;
    a.U = U;
    a.nwg = cofi_cdiv(a.T, U);
    a.maxp = (a.P + U - 2) / U + 1;
    a.bytes = (size_t)pairs * a.maxp * COFI_ATTN_SLOT_FLOATS * sizeof(float);
    return a;
}

__host__ __device__ __forceinline__ int attn_first_wg(const AttnLayout &lay, int pair) { return (int)(((long)pair * lay.P) / lay.U); }

// merged, normalised output of query row l (frame f, head h), 16-byte chunk `ch` (d = 4 ch .. 4 ch + 3)
__device__ __forceinline__ float4 attn_merged_chunk(const float *parts, const AttnLayout &lay, int f, int H, int h, int l, int ch) {
    const int q = l & 31;
    const int pair = (f * H + h) * lay.QB + (l >> 5);
    const int w0 = attn_first_wg(lay, pair), w1 = (int)((((long)pair + 1) * lay.P - 1) / lay.U);
    const float *slot = parts + (size_t)pair * lay.maxp * COFI_ATTN_SLOT_FLOATS;
    float mm = -1e30f;
    for (int s = 0; s <= w1 - w0; ++s) mm = fmaxf(mm, slot[(size_t)s * COFI_ATTN_SLOT_FLOATS + q]);
    float lsum = 0.f;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s <= w1 - w0; ++s) {
        const float *sl = slot + (size_t)s * COFI_ATTN_SLOT_FLOATS;
        const float sc = __builtin_amdgcn_exp2f(sl[q] - mm);
        lsum += sl[32 + q] * sc;
        const float4 t = *reinterpret_cast<const float4 *>(sl + 64 + q * 32 + 4 * ch);
        r.x += t.x * sc; r.y += t.y * sc; r.z += t.z * sc; r.w += t.w * sc;
    }
    const float inv = 1.0f / lsum;
    r.x *= inv; r.y *= inv; r.z *= inv; r.w *= inv;
    return r;
}
