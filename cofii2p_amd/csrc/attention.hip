// Flash-style multi-head attention for the I2P transformer on fp32 MFMA (v_mfma_f32_32x32x2_f32).
// Reference: model/transformer/linear_attention.py:56-79 (FullAttention) + the token-axis query
// normalisation of model/transformer/transformer.py:53, folded in as a per-channel scale.
//
// WORK PARTITION.  A unit = one 32-query block x one 32-key block of one (frame, head): 16 + 16 MFMAs.  A PAIR = all P =
// ceil(S/32) units of one (frame, head, query block).  One KITTI cross-attention call is 160 pairs x 40 units: handing whole
// pairs to workgroups fills 160 of 256 CUs (and splitting keys inside a workgroup cannot help: a pair's 40 units then sit on
// ONE CU).  Instead the T units of a launch are numbered pair-major and dealt out in equal contiguous ranges of U units to
// ~one workgroup per CU (25 units each for that call); the dispatcher's round-robin over the 8 XCDs is undone so that an XCD
// works through ONE contiguous eighth of the numbering = the query blocks of (mostly) one head: its L2 pulls only that head's
// K / V.  A workgroup's range covers at most 3 pairs ("segments"); inside it the 8 waves take the units round-robin, each
// running an online softmax in registers per segment, and the waves' states are merged through LDS at the end.  A pair that
// is spread over several workgroups leaves one PARTIAL (running max m, row sum l, un-normalised O) per workgroup in a slot
// table; the slots of a pair are combined by the consumer of the attention output (the fused layer tail, transformer_tail.hip)
// or by attention_merge_kernel - a kernel boundary, not an in-launch hand-off, orders the two.  Everything is a fixed-order
// reduction: bit-reproducible.
//
// Per unit and wave:
//   S^T[key, q] = K_blk . Q^T    16 MFMA, A = K straight from L2 (float4 per lane), B = Q registers
//   online softmax, per lane = per query column (q = lane&31), 16 keys per lane, halves joined by
//   one __shfl_xor(.., 32)
//   O^T[d, q] += V_blk^T . P^T   16 MFMA, A = V[key, d = lane&31] (block fetched as 16-B loads, transposed through a
//                                wave-private LDS tile), B = P
// The MFMA D layout of S^T (lane (q,h) holds keys (r&3)+8(r>>2)+4h) is exactly the B-operand layout
// the second chain needs once the contraction index is allowed to run in that (permuted) key order,
// so P never moves between lanes and O^T keeps "one query per lane": the softmax rescale and the
// final 1/l are plain per-lane multiplies.
#include "attention_parts.h"
#include "common.h"

namespace {

constexpr int D = 32;      // head dimension
constexpr int NW = 8;      // waves per workgroup
constexpr int NSEG = COFI_ATTN_MAX_SEGMENTS;

struct AttnArgs {
    const float *Q, *K, *V, *qs;
    float *parts;
    int ldq, ldk, ldv, L, S, H;
    float scale_log2e;
    // alternative to qs: the column partials of the projection that produced Q (cofi_gemm_f32_colstats: (frames*nslab, ncols, 2)
    // {sum, sum of squares}); the kernel folds them into 1 / max(||Q[:, c]||, eps) itself (transformer.py:53)
    const float *q_colpart;
    int q_nslab, q_ncols;
    float q_eps;
    AttnLayout lay;
};

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// the dispatcher sends block b to XCD b % 8: give every XCD one contiguous eighth of the logical numbering (bijective)
__device__ __forceinline__ int xcd_contiguous_block(int b, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = b & 7, i = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// LDS image of a wave's O^T state: (32 queries, 32 d) with the 16-byte chunks of a row XOR-swizzled by the row, so the
// b128 stores of 8 consecutive query lanes hit 8 different chunk slots (conflict free without padding: 4 KB per state)
__device__ __forceinline__ int so_off(int q, int chunk) { return q * 32 + ((chunk ^ (q & 7)) << 2); }

__global__ __launch_bounds__(64 * NW) void attention_flat_kernel(AttnArgs a) {
    // LDS carve (floats)
    constexpr int SO = NSEG * NW * 1024, SM = NSEG * NW * 32, SV = NW * 32 * (D + 4), SQ = NSEG * 32, SP = 2 * NW * NSEG * 32;
    __shared__ __attribute__((aligned(16))) float lds[SO + 2 * SM + SV + SQ + SP];
    float *s_o = lds, *s_m = s_o + SO, *s_l = s_m + SM, *s_v = s_l + SM, *s_qs = s_v + SV, *s_part = s_qs + SQ;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const AttnLayout &lay = a.lay;
    const int lb = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const int u0 = lb * lay.U, u1 = min(lay.T, u0 + lay.U);
    const int nu = u1 - u0;                      // >= 1
    const int pair0 = u0 / lay.P;
    const int nseg = (u1 - 1) / lay.P - pair0 + 1;   // <= NSEG (host: U <= 2P + 1)

    // pair -> (frame, head, first query row)
    auto pair_fhq = [&](int pair, int &f, int &h, int &q0) {
        const int fh = pair / lay.QB;
        q0 = (pair - fh * lay.QB) * 32;
        f = fh / a.H;
        h = fh - f * a.H;
    };
    // unit j of this wave -> segment, key block, K / V base of its (frame, head)
    struct Unit { int seg, kb; const float *k, *v; };
    auto unit = [&](int j) {
        const int u = u0 + wave + NW * j;
        const int pair = u / lay.P;
        int f, h, q0;
        pair_fhq(pair, f, h, q0);
        Unit r;
        r.seg = pair - pair0;
        r.kb = u - pair * lay.P;
        r.k = a.K + (size_t)f * a.S * a.ldk + h * D;
        r.v = a.V + (size_t)f * a.S * a.ldv + h * D;
        return r;
    };
    const int nj = wave < nu ? (nu - wave + NW - 1) / NW : 0;

    auto load_k = [&](const Unit &un, f32x4(&kf)[4]) {
        const int key = min(un.kb * 32 + li, a.S - 1);
        const float *kp = un.k + (size_t)key * a.ldk + 4 * lh;
#pragma unroll
        for (int c = 0; c < 4; ++c) kf[c] = *reinterpret_cast<const f32x4 *>(kp + 8 * c);
    };
    auto load_v = [&](const Unit &un, f32x4(&vr)[4]) {   // raw rows: lane l holds float4 #(l & 7) of key rows (l >> 3) + 8j
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kr = min(un.kb * 32 + (lane >> 3) + 8 * j, a.S - 1);
            vr[j] = *reinterpret_cast<const f32x4 *>(un.v + (size_t)kr * a.ldv + 4 * (lane & 7));
        }
    };

    // ---- prologue: everything that goes to memory is issued before the first wait
    // raw Q fragments of the (up to 3) segments: lane (q = li, h) holds Q[q][8c+4h+e]
    f32x4 qraw[NSEG][4];
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
        int f, h, q0;
        pair_fhq(pair0 + min(s, nseg - 1), f, h, q0);
        const int q = min(q0 + li, a.L - 1);
        const float *qp = a.Q + ((size_t)f * a.L + q) * a.ldq + h * D + 4 * lh;
#pragma unroll
        for (int c = 0; c < 4; ++c) qraw[s][c] = *reinterpret_cast<const f32x4 *>(qp + 8 * c);
    }
    f32x4 kC[4], kN[4], kN2[4], vR[4];
    Unit uC{}, uN{};
    if (nj > 0) {
        uC = unit(0);
        uN = unit(min(1, nj - 1));
        load_k(uC, kC);
        load_v(uC, vR);
        load_k(uN, kN);
    }
    // token-axis norm of the segments' 32 Q columns (transformer.py:53): from the projection's column partials - thread
    // (phase, column) sums the squares of slabs phase, phase + 2 NW, ...; column threads fold the phases in a fixed order
    {
        const int col = threadIdx.x & 31, ph = threadIdx.x >> 5;   // 2 NW phases
        if (a.q_colpart) {
            for (int s = 0; s < nseg; ++s) {
                int f, h, q0;
                pair_fhq(pair0 + s, f, h, q0);
                const float *cp = a.q_colpart + ((size_t)f * a.q_nslab * a.q_ncols + h * D + col) * 2 + 1;
                float acc = 0.f;
                for (int b = ph; b < a.q_nslab; b += 2 * NW) acc += cp[(size_t)b * a.q_ncols * 2];
                s_part[(ph * NSEG + s) * 32 + col] = acc;
            }
        }
        // neutral state in every (segment, wave) slot: a wave without a unit in a segment contributes nothing to its merge
        if (lh == 0) {
#pragma unroll
            for (int s = 0; s < NSEG; ++s) {
                s_m[(s * NW + wave) * 32 + li] = -1e30f;
                s_l[(s * NW + wave) * 32 + li] = 0.f;
            }
        }
        __syncthreads();
        if (threadIdx.x < 32 * NSEG) {
            const int s = threadIdx.x >> 5;
            float v = 1.0f;
            if (s < nseg) {
                int f, h, q0;
                pair_fhq(pair0 + s, f, h, q0);
                if (a.q_colpart) {
                    float t = 0.f;
#pragma unroll
                    for (int p = 0; p < 2 * NW; ++p) t += s_part[(p * NSEG + s) * 32 + col];
                    v = 1.0f / fmaxf(sqrtf(t), a.q_eps);
                } else if (a.qs) {
                    v = a.qs[((size_t)f * a.H + h) * D + col];
                }
            }
            s_qs[threadIdx.x] = v;
        }
        __syncthreads();
    }
    // scaled Q fragments: q * colscale * softmax scale * log2(e)
    float qf[NSEG][16];
#pragma unroll
    for (int s = 0; s < NSEG; ++s)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(&s_qs[s * 32 + 4 * lh + 8 * c]);
#pragma unroll
            for (int e = 0; e < 4; ++e) qf[s][4 * c + e] = (qraw[s][c][e] * sc[e]) * a.scale_log2e;
        }

    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    float *sv = s_v + wave * 32 * (D + 4);
    auto transpose_v = [&](const f32x4(&vr)[4], float(&vf)[16]) {   // -> lane (d = li, h): V[keyrow(r, h)][d]
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4 *>(&sv[((lane >> 3) + 8 * j) * (D + 4) + 4 * (lane & 7)]) = vr[j];
#pragma unroll
        for (int r = 0; r < 16; ++r) vf[r] = sv[((r & 3) + 8 * (r >> 2) + 4 * lh) * (D + 4) + li];
    };
    auto qk1 = [&](const f32x4(&kf)[4], const float(&q)[16]) {  // S^T = K . Q^T, 16 chained MFMAs
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c][e], q[4 * c + e], s, 0, 0, 0);
        }
        return s;
    };
    auto qk = [&](const f32x4(&kf)[4], int seg) {   // seg is wave-uniform: a scalar branch, no register indexing
        if (seg == 0) return qk1(kf, qf[0]);
        if (seg == 1) return qk1(kf, qf[1]);
        return qk1(kf, qf[2]);
    };
    auto softmax = [&](int kb, f32x16 &s) {  // online softmax (base 2); returns P in s, rescales o
        const int k0 = kb * 32;
        if (k0 + 32 > a.S) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (k0 + (r & 3) + 8 * (r >> 2) + 4 * lh >= a.S) s[r] = -INFINITY;
        }
        float bmax = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) bmax = fmaxf(bmax, s[r]);
        bmax = fmaxf(bmax, __shfl_xor(bmax, 32, 64));
        const float m_new = fmaxf(m_run, bmax);
        const float alpha = fast_exp2(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = fast_exp2(s[r] - m_new);
            psum += s[r];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= alpha;
    };
    auto pv = [&](const f32x16 &p, const float(&vf)[16]) {  // O^T += V^T . P^T in D-layout key order
#pragma unroll
        for (int r = 0; r < 16; ++r) o = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[r], p[r], o, 0, 0, 0);
    };
    auto flush = [&](int seg) {   // this wave's state of segment `seg` -> its LDS slot; fresh state for the next segment
        l_run += __shfl_xor(l_run, 32, 64);   // join the two halves' row sums (same m in both halves by construction)
        const int slot = seg * NW + wave;
        if (lh == 0) {
            s_m[slot * 32 + li] = m_run;
            s_l[slot * 32 + li] = l_run;
        }
        float *so = s_o + slot * 1024;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)   // lane (q, h) holds O[q][8 rq + 4 h .. +3] in regs 4rq .. 4rq+3: chunk 2 rq + h
            *reinterpret_cast<float4 *>(&so[so_off(li, 2 * rq + lh)]) = make_float4(o[4 * rq], o[4 * rq + 1], o[4 * rq + 2], o[4 * rq + 3]);
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
        m_run = -1e30f;
        l_run = 0.f;
    };

    // Software pipeline over this wave's units: the next unit's QK^T MFMA chain is issued before the current unit's softmax
    // (VALU / transcendental work runs under matrix-core time), V of the next unit and K of the unit after it are in flight.
    if (nj > 0) {
        float vC[16];
        f32x16 sC = qk(kC, uC.seg);
        transpose_v(vR, vC);
        for (int j = 0; j + 1 < nj; ++j) {
            const Unit uN2 = unit(min(j + 2, nj - 1));
            load_k(uN2, kN2);
            load_v(uN, vR);                     // raw rows of the next unit fly under this unit's MFMA chains ...
            f32x16 sN = qk(kN, uN.seg);
            softmax(uC.kb, sC);
            pv(sC, vC);
            if (uN.seg != uC.seg) flush(uC.seg);
            transpose_v(vR, vC);                // ... and go through LDS once the PV chain has read the current ones
            sC = sN;
#pragma unroll
            for (int c = 0; c < 4; ++c) kN[c] = kN2[c];
            uC = uN;
            uN = uN2;
        }
        softmax(uC.kb, sC);
        pv(sC, vC);
        flush(uC.seg);
    }
    __syncthreads();

    // ---- merge the waves' states of every segment (fixed order: deterministic) -> the pair's slot of this workgroup
    for (int e = threadIdx.x; e < nseg * 256; e += 64 * NW) {   // per segment: 32 queries x 8 float4 chunks
        const int s = e >> 8, q = (e >> 3) & 31, ch = e & 7;
        float mm = -1e30f;
#pragma unroll
        for (int w = 0; w < NW; ++w) mm = fmaxf(mm, s_m[(s * NW + w) * 32 + q]);
        float l = 0.f;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float lw = s_l[(s * NW + w) * 32 + q];
            if (lw > 0.f) {   // waves without a unit in this segment left the neutral state (and never wrote their O image)
                const float sc = fast_exp2(s_m[(s * NW + w) * 32 + q] - mm);
                l += lw * sc;
                const float4 t = *reinterpret_cast<const float4 *>(&s_o[(s * NW + w) * 1024 + so_off(q, ch)]);
                r.x += t.x * sc; r.y += t.y * sc; r.z += t.z * sc; r.w += t.w * sc;
            }
        }
        const int pair = pair0 + s;
        float *slot = a.parts + ((size_t)pair * lay.maxp + (lb - attn_first_wg(lay, pair))) * COFI_ATTN_SLOT_FLOATS;
        if (ch == 0) {
            slot[q] = mm;
            slot[32 + q] = l;
        }
        *reinterpret_cast<float4 *>(slot + 64 + q * 32 + 4 * ch) = r;
    }
}

// O[row, h*32 + d] = sum_slots O_s * 2^(m_s - m) / sum_slots l_s * 2^(m_s - m): thread = (query row, head, 16-byte chunk)
__global__ __launch_bounds__(256) void attention_merge_kernel(const float *parts, AttnLayout lay, int L, int H, int frames, float *O, int ldo) {
    const size_t total = (size_t)frames * L * H * 8;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(e & 7);
        const int h = (int)((e >> 3) % H);
        const size_t row = (e >> 3) / H;          // frame * L + l
        const int f = (int)(row / L), l = (int)(row - (size_t)f * L);
        const float4 r = attn_merged_chunk(parts, lay, f, H, h, l, ch);
        *reinterpret_cast<float4 *>(O + row * ldo + h * D + 4 * ch) = r;
    }
}

int attention_check(const float *Q, int ldq, const float *K, int ldk, const float *V, int ldv, int L, int S, int H, int D_, int frames) {
    if (!Q || !K || !V || L <= 0 || S <= 0 || H <= 0 || frames <= 0) return COFI_EINVAL;
    if (D_ != D) return COFI_EUNSUPPORTED;
    if ((ldq & 3) || (ldk & 3) || ldq < H * D || ldk < H * D || ldv < H * D) return COFI_EINVAL;
    if ((ldv & 3) || ((uintptr_t)V & 15) || ((uintptr_t)Q & 15) || ((uintptr_t)K & 15)) return COFI_EINVAL;
    return 0;
}

int launch_parts(AttnArgs a, int frames, hipStream_t stream) {
    a.lay = attn_layout(a.L, a.S, a.H, frames);
    hipLaunchKernelGGL(attention_flat_kernel, dim3(a.lay.nwg), dim3(64 * NW), 0, stream, a);
    return cofi_launch_status();
}

}  // namespace

extern "C" size_t cofi_attention_workspace(int L, int S, int H, int D_, int frames) {
    if (L <= 0 || S <= 0 || H <= 0 || D_ != D || frames <= 0) return 0;
    return attn_layout(L, S, H, frames).bytes;
}

extern "C" int cofi_attention_parts(const float *Q, int ldq, const float *K, int ldk, const float *V, int ldv, const float *q_colscale,
                                    const float *q_colpart, int q_nslab, int q_ncols, float q_eps, int L, int S, int H, int D_, float scale,
                                    int frames, void *parts, size_t parts_bytes, cofi_stream_t stream) {
    if (int rc = attention_check(Q, ldq, K, ldk, V, ldv, L, S, H, D_, frames)) return rc;
    if (q_colscale && q_colpart) return COFI_EINVAL;
    if (q_colpart) {
        // the partials' 64-row slabs must be whole per frame
        if (q_nslab <= 0 || (q_nslab % frames) || q_ncols < H * D || q_nslab / frames != cofi_cdiv(L, 64) || (frames > 1 && (L % 64))) return COFI_EINVAL;
    }
    if (!parts || ((uintptr_t)parts & 15) || parts_bytes < attn_layout(L, S, H, frames).bytes) return COFI_EWORKSPACE;
    AttnArgs a{Q, K, V, q_colscale, (float *)parts, ldq, ldk, ldv, L, S, H, scale * 1.4426950408889634f, q_colpart,
               q_colpart ? q_nslab / frames : 0, q_ncols, q_eps, {}};
    return launch_parts(a, frames, cofi_s(stream));
}

extern "C" int cofi_attention_merge(const void *parts, size_t parts_bytes, int L, int S, int H, int D_, int frames, float *O, int ldo,
                                    cofi_stream_t stream) {
    if (!parts || !O || L <= 0 || S <= 0 || H <= 0 || frames <= 0 || D_ != D || (ldo & 3) || ldo < H * D || ((uintptr_t)O & 15)) return COFI_EINVAL;
    const AttnLayout lay = attn_layout(L, S, H, frames);
    if (parts_bytes < lay.bytes) return COFI_EWORKSPACE;
    const size_t total = (size_t)frames * L * H * 8;
    int nb = (int)((total + 255) / 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(attention_merge_kernel, dim3(nb), dim3(256), 0, cofi_s(stream), (const float *)parts, lay, L, H, frames, O, ldo);
    return cofi_launch_status();
}

extern "C" int cofi_attention_fwd(const float *Q, int ldq, const float *K, int ldk, const float *V, int ldv, const float *q_colscale,
                                  float *O, int ldo, int L, int S, int H, int D_, float scale, void *ws, size_t ws_bytes, int frames,
                                  cofi_stream_t stream) {
    if (int rc = cofi_attention_parts(Q, ldq, K, ldk, V, ldv, q_colscale, nullptr, 0, 0, 0.f, L, S, H, D_, scale, frames, ws, ws_bytes, stream)) return rc;
    return cofi_attention_merge(ws, ws_bytes, L, S, H, D_, frames, O, ldo, stream);
}

extern "C" int cofi_attention_fwd_colpart(const float *Q, int ldq, const float *K, int ldk, const float *V, int ldv, const float *q_colpart,
                                          int q_nslab, int q_ncols, float q_eps, float *O, int ldo, int L, int S, int H, int D_,
                                          float scale, void *ws, size_t ws_bytes, int frames, cofi_stream_t stream) {
    if (int rc = cofi_attention_parts(Q, ldq, K, ldk, V, ldv, nullptr, q_colpart, q_nslab, q_ncols, q_eps, L, S, H, D_, scale, frames, ws, ws_bytes,
                                      stream))
        return rc;
    return cofi_attention_merge(ws, ws_bytes, L, S, H, D_, frames, O, ldo, stream);
}
