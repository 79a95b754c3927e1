// Layout of the attention partials (attention.hip) shared by the producer, the stand-alone merge kernel and the consumers
// that merge on the fly (transformer_tail.hip).
//
// A PAIR = (frame, head, 32-query block).  The P = ceil(S / 32) key blocks of every pair are cut into KS contiguous ranges
// (as even as possible); each range is handled by another workgroup, which leaves ONE slot {m[32], l[32], O[32][32]} (running
// max, row sum, un-normalised output relative to m) at
//     parts[(pair * KS + ks) * COFI_ATTN_SLOT_FLOATS].
// A reader combines the KS slots of a query row in a fixed order:
//     O = sum_s O_s 2^(m_s - m) / sum_s l_s 2^(m_s - m),   m = max_s m_s.
// KS comes from (L, S, H, frames) and the device's CU count (attn_layout): enough ranges to give every CU a workgroup, as few
// as possible beyond that.
#pragma once
#include "common.h"
#include <stdlib.h>

#define COFI_ATTN_SLOT_FLOATS 1088
#define COFI_ATTN_MAX_KS 8
#define COFI_ATTN_QG 2      /* query blocks sharing one workgroup's K / V tiles */
#define COFI_ATTN_KPH 4     /* key blocks a workgroup stages and processes per step (one per wave group) */

struct AttnLayout {
    int P;      // key blocks per pair
    int QB;     // query blocks per (frame, head)
    int QSB;    // 64-query super blocks per (frame, head)
    int KS;     // key ranges = slots per pair
    int nwg;    // workgroups = frames * H * QSB * KS
    int light;  // kernel variant: 1 = two workgroups per CU
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

// Two variants of the attention kernel: the "heavy" one (one workgroup per CU, the wave software-pipelines QK^T of the next unit
// over the softmax of the current one) and the "light" one (<= 128 VGPRs, two workgroups per CU: one workgroup's prologue - first
// K / V tiles, Q-norm fold - and merge run under the other's MFMA phase).  Measured on MI355X (us per launch, heavy / light): one
// KITTI cross-attention call (80 query super blocks) 14.9 / 16.3, the joint self-attention call (160) 28.3 / 24.6, 16 stacked
// frames 158.9 / 150.0 - light wins once there are enough super blocks to give every CU two workgroups without splitting the keys
// further.
static inline AttnLayout attn_layout(int L, int S, int H, int frames) {
    AttnLayout a;
    a.P = cofi_cdiv(S, 32);
    a.QB = cofi_cdiv(L, 32);
    a.QSB = cofi_cdiv(a.QB, COFI_ATTN_QG);
    const long sp = (long)frames * H * a.QSB;   // workgroups per key range
    // Cost of a split in "steps" (one 32 x 32 unit per wave): waves of workgroups the chip runs one after the other, each
    // ceil(blocks / KPH) steps long plus a fixed prologue + merge cost of about 1.5 steps.
    a.light = sp * 2 >= attn_num_cus();
    const int ncu = attn_num_cus() * (a.light ? 2 : 1);
    int best = 1;
    double best_cost = 1e30;
    for (int ks = 1; ks <= COFI_ATTN_MAX_KS && ks <= a.P; ++ks) {
        const double rounds = (double)cofi_cdiv(sp * ks, ncu);
        const double cost = rounds * (cofi_cdiv(cofi_cdiv(a.P, ks), COFI_ATTN_KPH) + 1.5);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = ks; }
    }
    a.KS = best;
    a.nwg = (int)(sp * a.KS);
    a.bytes = (size_t)frames * H * a.QB * a.KS * COFI_ATTN_SLOT_FLOATS * sizeof(float);
    return a;
}

// merged, normalised output of query row l (frame f, head h), 16-byte chunk `ch` (d = 4 ch .. 4 ch + 3).  Every load is
// issued before the first use: one round trip whatever KS is.
__device__ __forceinline__ float4 attn_merged_chunk(const float *parts, const AttnLayout &lay, int f, int H, int h, int l, int ch) {
    const int q = l & 31;
    const int pair = (f * H + h) * lay.QB + (l >> 5);
    const float *slot = parts + (size_t)pair * lay.KS * COFI_ATTN_SLOT_FLOATS;
    float m[COFI_ATTN_MAX_KS], ls[COFI_ATTN_MAX_KS];
    float4 o[COFI_ATTN_MAX_KS];
#pragma unroll
    for (int s = 0; s < COFI_ATTN_MAX_KS; ++s) {
        if (s < lay.KS) {   // uniform
            const float *sl = slot + (size_t)s * COFI_ATTN_SLOT_FLOATS;
            m[s] = sl[q];
            ls[s] = sl[32 + q];
            o[s] = *reinterpret_cast<const float4 *>(sl + 64 + q * 32 + 4 * ch);
        }
    }
    float mm = -1e30f;
#pragma unroll
    for (int s = 0; s < COFI_ATTN_MAX_KS; ++s)
        if (s < lay.KS) mm = fmaxf(mm, m[s]);
    float lsum = 0.f;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int s = 0; s < COFI_ATTN_MAX_KS; ++s)
        if (s < lay.KS) {
            const float sc = __builtin_amdgcn_exp2f(m[s] - mm);
            lsum += ls[s] * sc;
            r.x += o[s].x * sc; r.y += o[s].y * sc; r.z += o[s].z * sc; r.w += o[s].w * sc;
        }
    const float inv = 1.0f / lsum;
    r.x *= inv; r.y *= inv; r.z *= inv; r.w *= inv;
    return r;
}
