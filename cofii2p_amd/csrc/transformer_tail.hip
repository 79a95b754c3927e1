// Fused tail of one LoFTR encoder layer (model/transformer/transformer.py:57-64) in ONE kernel:
//     m   = LayerNorm1( msg @ Wm^T )
//     h   = relu( [x | m] @ W0^T )
//     out = x + LayerNorm2( h @ W2^T )
// for d_model = 128 (the reference's only configuration, model/network.py:35), optionally followed - still inside the launch - by what
// the NEXT layers read first (transformer.py:45-47 of layer l + 1, and of the second direction of a cross layer):
//     y_s = out @ Wp_s^T                  up to two projection segments (stacked [Wq; Wk; Wv] blocks of the layers that follow),
//                                         with the column partials {sum, sum of squares} per 32-row slab that the attention kernel folds
//                                         into the token-axis norm of Q (transformer.py:53)
//     out_l2 = F.normalize(out, dim=1)    token-major and / or channel-major (network.py:125-126 after the last layer)
// At batch 1 the GEMMs of this chain are 1280 x {128,256,128,384} problems: as separate launches each is bound by launch + memory
// latency, not by the matrix cores.  Here a workgroup owns 32 token rows end to end; the intermediates m, [x|m], h and the bf16 planes
// of out never leave LDS, and the only global traffic is one read of msg/x, the writes of out / y_s and the weight stream (L2).
//
// Arithmetic (template NPL): NPL = 2: 3-term bf16 split (hi*hi + hi*lo + lo*hi), NPL = 3: the fp32-grade 6-term split
// (hi + mid + lo planes: hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi) on v_mfma_f32_32x32x16_bf16 with fp32 accumulation - the
// same products as cofi_gemm_f32 with COFI_GEMM_BF16X3 / COFI_GEMM_BF16X6.  Weights arrive PRE-SPLIT into bf16 planes (packed once at
// load time), so a wave streams its B fragments straight from L2 into registers (each wave owns distinct weight rows: nothing to share
// through LDS) ahead of the MFMAs; activations are split once when they enter LDS and are shared by the four waves as A fragments.
#include "attention_parts.h"
#include "common.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
union Frag { uint4 u; bf16x8 v; };

constexpr int C = 128;            // d_model
constexpr int R = 32;             // token rows per workgroup
constexpr int S128 = 128 * 2 + 16;  // LDS row stride (bytes) of a 128-deep bf16 plane: 68 dwords
constexpr int S256 = 256 * 2 + 16;  // ... of a 256-deep plane: 132 dwords (both = 4 mod 64: conflict-free b128 reads)
constexpr int FLD = C + 4;          // fp32 staging tile leading dimension
// chunks of weight fragments in flight ahead of the one being multiplied, bf16x6 (three planes): stage 2 / stage 3 / projections.  Deeper
// (3 / 3 / 2: 256 + 76 registers, no spills; 5 / 3 / 3 spills) measured the same batch-1 pipeline rate on one box - 533.5 / 532.7 vs 533.4 /
// 532.5 frames/s, profiles/r06/ab_tail_pf.txt - so the shallower, smaller form stays.
#ifndef TAIL_PF_S2
#define TAIL_PF_S2 2
#endif
#ifndef TAIL_PF_S3
#define TAIL_PF_S3 2
#endif
#ifndef TAIL_PF_PROJ
#define TAIL_PF_PROJ 1
#endif

struct ProjSeg {
    const unsigned short *w[3];   // planes of the (N, 128) stacked projection weight
    float *y, *part;              // y (rows, N) ldy; part (rows / 32, N, 2) or nullptr
    int N, ldy;                   // N in {0, 128, 256, 384}
};

struct TailArgs {
    const float *msg, *x;
    const unsigned short *wm[3], *w0[3], *w2[3];
    const float *n1g, *n1b, *n2g, *n2b;
    float *out;
    int ldm, ldx, ldo, L;
    float eps;
    // msg == nullptr: the attention output is still in the form the attention kernel left it - per-pair partial slots
    // (attention_parts.h) - and the loader below merges + normalises them on the fly (L = frames * Lf rows, H heads of 32)
    const float *parts;
    AttnLayout lay;
    int Lf, H;
    ProjSeg proj[2];
    float *l2, *l2t;   // optional F.normalize(out, dim=1): (rows, 128) ld_l2 and / or channel-major (128, rows) ld_l2t
    int ld_l2, ld_l2t;
};

__device__ __forceinline__ unsigned cvt_pk(float a, float b) {  // RNE, a -> low half
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// planes of a tile live `pstride` bytes apart: hi | (mid |) lo
template <int NPL>
__device__ __forceinline__ void split_store4(const float4 v, unsigned char *p, int pstride) {
    uint2 hi, lo;
    hi.x = cvt_pk(v.x, v.y);
    hi.y = cvt_pk(v.z, v.w);
    const float rx = v.x - __uint_as_float(hi.x << 16), ry = v.y - __uint_as_float(hi.x & 0xffff0000u);
    const float rz = v.z - __uint_as_float(hi.y << 16), rw = v.w - __uint_as_float(hi.y & 0xffff0000u);
    lo.x = cvt_pk(rx, ry);
    lo.y = cvt_pk(rz, rw);
    *reinterpret_cast<uint2 *>(p) = hi;
    *reinterpret_cast<uint2 *>(p + pstride) = lo;   // NPL == 3: this is the mid plane
    if constexpr (NPL == 3) {
        uint2 l3;
        l3.x = cvt_pk(rx - __uint_as_float(lo.x << 16), ry - __uint_as_float(lo.x & 0xffff0000u));
        l3.y = cvt_pk(rz - __uint_as_float(lo.y << 16), rw - __uint_as_float(lo.y & 0xffff0000u));
        *reinterpret_cast<uint2 *>(p + 2 * pstride) = l3;
    }
}
template <int NPL>
__device__ __forceinline__ void split_store1(float v, unsigned char *p, int pstride) {
    const unsigned h = cvt_pk(v, 0.f) & 0xffffu;
    const float r1 = v - __uint_as_float(h << 16);
    const unsigned l = cvt_pk(r1, 0.f) & 0xffffu;
    *reinterpret_cast<unsigned short *>(p) = (unsigned short)h;
    *reinterpret_cast<unsigned short *>(p + pstride) = (unsigned short)l;
    if constexpr (NPL == 3) {
        const unsigned l3 = cvt_pk(r1 - __uint_as_float(l << 16), 0.f) & 0xffffu;
        *reinterpret_cast<unsigned short *>(p + 2 * pstride) = (unsigned short)l3;
    }
}

// One GEMM stage of a wave: acc[t] (32 rows x 32 cols, t < NT) += A(32 x K, bf16 planes in LDS) . W[nbase + 32t + (0..31), 0..K)^T.
// W planes are (N, K) bf16 row-major in global memory.  KSTEPS = K / 16, in chunks of CH steps; PF chunks of B fragments
// are in flight ahead of the one being multiplied.  The workgroup is alone on its CU (40-80 workgroups per launch) and a
// chunk is only ~0.2 us of MFMA work, so the stage is a chain of L2 round trips: PF = all chunks (one round trip per stage)
// where the registers allow it - one wave per SIMD may use the whole 512-entry register file.
// FRAG: the weight planes are stored in fragment order (cofi_loftr_tail_desc_t::w_frag): block (row block T, k-step s) = the wave's 64 x 16 B.
template <int NPL, int NT, int KSTEPS, int PF, int CH, bool FRAG>
__device__ __forceinline__ void gemm_stage(f32x16 (&acc)[NT], const unsigned char *a_pl, int a_pstride, int a_stride,
                                           const unsigned short *const *w, int K, int nbase, int li, int lh) {
    constexpr int NCH = KSTEPS / CH;
    constexpr int SLOTS = PF + 1 < NCH ? PF + 1 : NCH;
    Frag b[SLOTS][NPL][NT][CH];
    auto loadw = [&](int c, int slot) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const size_t roff = FRAG ? ((size_t)((nbase >> 5) + t) * KSTEPS * 64 + (li + 32 * lh)) * 8 : (size_t)(nbase + 32 * t + li) * K + 8 * lh;
#pragma unroll
            for (int s = 0; s < CH; ++s) {
                const int k0 = (FRAG ? 512 : 16) * (c * CH + s);
#pragma unroll
                for (int p = 0; p < NPL; ++p) b[slot][p][t][s].u = *reinterpret_cast<const uint4 *>(w[p] + roff + k0);
            }
        }
    };
#pragma unroll
    for (int c = 0; c < PF && c < NCH; ++c) loadw(c, c % SLOTS);
    // `aoff` (the lane's byte offset into the A planes) doubles as the anchor of compiler-level barriers: every A-fragment
    // read below depends on it, and no load above an anchor may sink below it.  Without the anchors the scheduler moves each
    // weight load down to the MFMA that consumes it (less register pressure) - a chain of exposed L2 round trips.
    unsigned aoff = li * a_stride + lh * 16;
    asm volatile("" : "+v"(aoff) : : "memory");
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c + PF < NCH) {
            loadw(c + PF, (c + PF) % SLOTS);
            asm volatile("" : "+v"(aoff) : : "memory");
        }
        const unsigned char *ap = a_pl + aoff;
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            Frag fa[NPL];
#pragma unroll
            for (int p = 0; p < NPL; ++p) fa[p].u = *reinterpret_cast<const uint4 *>(ap + p * a_pstride + (c * CH + s) * 32);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const Frag(&bb)[NPL][NT][CH] = b[c % SLOTS];
                if constexpr (NPL == 3) {   // smallest terms first: 2^-16 (lo*hi, hi*lo, mid*mid), 2^-8 (mid*hi, hi*mid), 1 (hi*hi)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2].v, bb[0][t][s].v, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0].v, bb[2][t][s].v, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1].v, bb[1][t][s].v, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1].v, bb[0][t][s].v, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0].v, bb[1][t][s].v, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0].v, bb[0][t][s].v, acc[t], 0, 0, 0);
                } else {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1].v, bb[0][t][s].v, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0].v, bb[1][t][s].v, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0].v, bb[0][t][s].v, acc[t], 0, 0, 0);
                }
            }
        }
    }
}

// y = out_tile @ Wp^T for one projection segment of N = 128 NT columns: wave w owns columns [32 NT w, 32 NT (w + 1)); the accumulator
// tiles go straight to memory (a store instruction of a 32 x 32 D tile writes two 128-byte row segments) and each column's {sum, sum
// of squares} over the workgroup's rows - one 32-row slab - to the partials table.
template <int NPL, int NT, bool FRAG>
__device__ __forceinline__ void proj_stage(const ProjSeg &ps, const unsigned char *a_pl, int a_pstride, int r0, int L, int wave, int li, int lh) {
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    gemm_stage<NPL, NT, 8, NPL == 3 ? TAIL_PF_PROJ : 1, NPL == 3 ? 2 : 4, FRAG>(acc, a_pl, a_pstride, S128, ps.w, C, wave * 32 * NT, li, lh);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = wave * 32 * NT + 32 * t + li;
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = r0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row < L) {
                ps.y[(size_t)row * ps.ldy + col] = acc[t][r];
                s += acc[t][r];
                q += acc[t][r] * acc[t][r];
            }
        }
        if (ps.part) {
            s += __shfl_xor(s, 32, 64);
            q += __shfl_xor(q, 32, 64);
            if (lh == 0) {
                float *o = ps.part + ((size_t)blockIdx.x * ps.N + col) * 2;
                o[0] = s;
                o[1] = q;
            }
        }
    }
}

template <int NPL, bool FRAG>
__global__ __launch_bounds__(256) void loftr_tail_kernel(TailArgs a) {
    // LDS carve (bytes): msg planes NPL*R*S128 | cat planes NPL*R*S256 | h planes NPL*R*S256 | fp32 staging R*FLD*4
    constexpr int P128 = R * S128, P256 = R * S256;   // bytes per plane
    constexpr int OFF_MSG = 0, OFF_CAT = OFF_MSG + NPL * P128, OFF_H = OFF_CAT + NPL * P256, OFF_F = OFF_H + NPL * P256;
    __shared__ __attribute__((aligned(16))) unsigned char lds[OFF_F + R * FLD * 4];
    unsigned char *msg_pl = lds + OFF_MSG, *cat_pl = lds + OFF_CAT, *h_pl = lds + OFF_H;
    float *stage = reinterpret_cast<float *>(lds + OFF_F);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int r0 = blockIdx.x * R;

    // ---- stage 0: msg and x tiles -> bf16 planes (32 lanes cover one 512-B row; 8 rows per pass)
    {
        const int lr = tid >> 5, lk = (tid & 31) * 4;
#pragma unroll
        for (int p = 0; p < R / 8; ++p) {
            const int rl = lr + 8 * p, row = min(r0 + rl, a.L - 1);
            float4 mv;
            if (a.parts) {
                const int f = row / a.Lf;
                mv = attn_merged_chunk(a.parts, a.lay, f, a.H, lk >> 5, row - f * a.Lf, (lk & 31) >> 2);
            } else {
                mv = *reinterpret_cast<const float4 *>(a.msg + (size_t)row * a.ldm + lk);
            }
            const float4 xv = *reinterpret_cast<const float4 *>(a.x + (size_t)row * a.ldx + lk);
            split_store4<NPL>(mv, msg_pl + rl * S128 + lk * 2, P128);
            split_store4<NPL>(xv, cat_pl + rl * S256 + lk * 2, P256);
        }
    }
    __syncthreads();

    // D layout of a 32x32 accumulator: row = (r&3) + 8*(r>>2) + 4*lh, col = li
    auto acc_to_stage = [&](const f32x16 &acc, int colbase) {
#pragma unroll
        for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * lh) * FLD + colbase + li] = acc[r];
    };
    // LayerNorm of the staged (R x 128) tile: wave w owns rows 8w .. 8w+7, lane owns columns lane, lane+64
    auto row_layernorm = [&](int rl, const float *g, const float *b, float &o0, float &o1) {
        const float v0 = stage[rl * FLD + lane], v1 = stage[rl * FLD + 64 + lane];
        const float mean = wave_sum(v0 + v1) * (1.0f / C);
        const float d0 = v0 - mean, d1 = v1 - mean;
        const float rstd = 1.0f / sqrtf(wave_sum(d0 * d0 + d1 * d1) * (1.0f / C) + a.eps);
        o0 = d0 * rstd * g[lane] + b[lane];
        o1 = d1 * rstd * g[64 + lane] + b[64 + lane];
    };

    // ---- stage 1: merged = msg @ Wm^T (K = 128, N = 128: one 32x32 tile per wave) -> LN1 -> right half of cat
    {
        f32x16 acc[1];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
        gemm_stage<NPL, 1, 8, 2, 4, FRAG>(acc, msg_pl, P128, S128, a.wm, C, wave * 32, li, lh);
        acc_to_stage(acc[0], wave * 32);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int rl = wave * 8 + i;
        float m0, m1;
        row_layernorm(rl, a.n1g, a.n1b, m0, m1);
        split_store1<NPL>(m0, cat_pl + rl * S256 + (C + lane) * 2, P256);
        split_store1<NPL>(m1, cat_pl + rl * S256 + (C + 64 + lane) * 2, P256);
    }
    __syncthreads();

    // ---- stage 2: h = relu([x|m] @ W0^T) (K = 256, N = 256: two 32x32 tiles per wave) -> h planes
    {
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        gemm_stage<NPL, 2, 16, NPL == 3 ? TAIL_PF_S2 : 2, NPL == 3 ? 2 : 4, FRAG>(acc, cat_pl, P256, S256, a.w0, 2 * C, wave * 64, li, lh);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (r & 3) + 8 * (r >> 2) + 4 * lh, col = wave * 64 + 32 * t + li;
                split_store1<NPL>(fmaxf(acc[t][r], 0.f), h_pl + rl * S256 + col * 2, P256);
            }
    }
    __syncthreads();

    // ---- stage 3: o = h @ W2^T (K = 256, N = 128) -> LN2 -> + x -> out
    {
        f32x16 acc[1];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
        gemm_stage<NPL, 1, 16, NPL == 3 ? TAIL_PF_S3 : 4, 4, FRAG>(acc, h_pl, P256, S256, a.w2, 2 * C, wave * 32, li, lh);
        acc_to_stage(acc[0], wave * 32);
    }
    __syncthreads();
    const bool want_proj = a.proj[0].N > 0, want_l2t = a.l2t != nullptr;   // uniform
    float nv0[8] = {}, nv1[8] = {};   // normalised rows (want_l2t): written to the staging tile once every wave has read its LayerNorm inputs
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int rl = wave * 8 + i, row = r0 + rl;
        float o0, o1;
        row_layernorm(rl, a.n2g, a.n2b, o0, o1);
        const float *xr = a.x + (size_t)min(row, a.L - 1) * a.ldx;
        o0 = xr[lane] + o0;
        o1 = xr[64 + lane] + o1;
        if (row < a.L) {
            float *dst = a.out + (size_t)row * a.ldo;
            dst[lane] = o0;
            dst[64 + lane] = o1;
        }
        if (want_proj) {   // the msg planes are dead since stage 1: they take the planes of out
            split_store1<NPL>(o0, msg_pl + rl * S128 + lane * 2, P128);
            split_store1<NPL>(o1, msg_pl + rl * S128 + (64 + lane) * 2, P128);
        }
        if (a.l2 || want_l2t) {   // F.normalize(out, dim=1): the operations of l2norm_rows_kernel in the same order
            const float inv = 1.0f / fmaxf(sqrtf(wave_sum(o0 * o0 + o1 * o1)), 1e-12f);
            nv0[i] = o0 * inv;
            nv1[i] = o1 * inv;
            if (a.l2 && row < a.L) {
                float *dst = a.l2 + (size_t)row * a.ld_l2;
                dst[lane] = nv0[i];
                dst[64 + lane] = nv1[i];
            }
        }
    }
    if (want_l2t) {
        // the wave's own 8 rows of the staging tile: no other wave reads or writes them before the barrier
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            stage[(wave * 8 + i) * FLD + lane] = nv0[i];
            stage[(wave * 8 + i) * FLD + 64 + lane] = nv1[i];
        }
    }
    if (want_proj || want_l2t) __syncthreads();
    if (want_l2t) {   // channel-major: thread (column c, half hf) writes 16 consecutive rows of column c = 64 bytes
        const int c = tid >> 1, hf = tid & 1;
        float *dst = a.l2t + (size_t)c * a.ld_l2t + r0 + 16 * hf;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (r0 + 16 * hf + j < a.L) dst[j] = stage[(16 * hf + j) * FLD + c];
    }
    // ---- stage 4: the next layers' projections of out
#pragma unroll
    for (int sgi = 0; sgi < 2; ++sgi) {
        const ProjSeg &ps = a.proj[sgi];
        if (ps.N == 128) proj_stage<NPL, 1, FRAG>(ps, msg_pl, P128, r0, a.L, wave, li, lh);
        else if (ps.N == 256) proj_stage<NPL, 2, FRAG>(ps, msg_pl, P128, r0, a.L, wave, li, lh);
        else if (ps.N == 384) proj_stage<NPL, 3, FRAG>(ps, msg_pl, P128, r0, a.L, wave, li, lh);
    }
}

int launch_tail(const TailArgs &a, int planes, hipStream_t s, bool frag = false) {
    const int nwg = cofi_cdiv(a.L, R);
    if (planes == 3) {
        if (frag) hipLaunchKernelGGL((loftr_tail_kernel<3, true>), dim3(nwg), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((loftr_tail_kernel<3, false>), dim3(nwg), dim3(256), 0, s, a);
    } else {
        if (frag) hipLaunchKernelGGL((loftr_tail_kernel<2, true>), dim3(nwg), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((loftr_tail_kernel<2, false>), dim3(nwg), dim3(256), 0, s, a);
    }
    return cofi_launch_status();
}

bool mis16(const void *p) { return ((uintptr_t)p & 15) != 0; }

}  // namespace

extern "C" int cofi_loftr_tail(const cofi_loftr_tail_desc_t *d, cofi_stream_t stream) {
    if (!d || !d->x || !d->wm || !d->w0 || !d->w2 || !d->n1_gamma || !d->n1_beta || !d->n2_gamma || !d->n2_beta || !d->out) return COFI_EINVAL;
    if ((d->planes != 2 && d->planes != 3) || (d->w_frag != 0 && d->w_frag != 1)) return COFI_EUNSUPPORTED;
    if ((d->msg != nullptr) == (d->parts != nullptr)) return COFI_EINVAL;   // exactly one form of the message operand
    if ((d->ldx & 3) || d->ldx < C || d->ldo < C || mis16(d->x) || mis16(d->wm) || mis16(d->w0) || mis16(d->w2)) return COFI_EINVAL;
    TailArgs a{};
    a.x = d->x; a.ldx = d->ldx; a.out = d->out; a.ldo = d->ldo; a.eps = d->eps;
    a.n1g = d->n1_gamma; a.n1b = d->n1_beta; a.n2g = d->n2_gamma; a.n2b = d->n2_beta;
    for (int p = 0; p < d->planes; ++p) {
        a.wm[p] = d->wm + (size_t)p * C * C;
        a.w0[p] = d->w0 + (size_t)p * 2 * C * 2 * C;
        a.w2[p] = d->w2 + (size_t)p * C * 2 * C;
    }
    int frame_rows;
    if (d->msg) {
        if (d->rows <= 0 || (d->ldm & 3) || d->ldm < C || mis16(d->msg)) return COFI_EINVAL;
        a.msg = d->msg; a.ldm = d->ldm; a.L = d->rows;
        frame_rows = d->frames > 0 ? d->rows / d->frames : d->rows;
    } else {
        if (d->L <= 0 || d->S <= 0 || d->frames <= 0 || d->H * 32 != C || mis16(d->parts)) return COFI_EINVAL;
        a.lay = attn_layout(d->L, d->S, d->H, d->frames);
        if (d->parts_bytes < a.lay.bytes) return COFI_EWORKSPACE;
        a.parts = (const float *)d->parts; a.Lf = d->L; a.H = d->H; a.L = d->L * d->frames;
        frame_rows = d->L;
    }
    for (int s = 0; s < 2; ++s) {
        const int N = d->proj_n[s];
        if (N == 0) continue;
        if (s == 1 && d->proj_n[0] == 0) return COFI_EINVAL;
        if ((N != 128 && N != 256 && N != 384) || !d->proj_w[s] || !d->proj_y[s] || d->proj_ldy[s] < N || mis16(d->proj_w[s])) return COFI_EINVAL;
        // statistics slabs = the workgroups' 32-row blocks: they must not straddle frames
        if (d->proj_part[s] && (frame_rows % R)) return COFI_EUNSUPPORTED;
        ProjSeg &ps = a.proj[s];
        for (int p = 0; p < d->planes; ++p) ps.w[p] = d->proj_w[s] + (size_t)p * N * C;
        ps.y = d->proj_y[s]; ps.part = d->proj_part[s]; ps.N = N; ps.ldy = d->proj_ldy[s];
    }
    if ((d->out_l2 && d->ld_l2 < C) || (d->out_l2t && d->ld_l2t < a.L)) return COFI_EINVAL;
    a.l2 = d->out_l2; a.ld_l2 = d->ld_l2; a.l2t = d->out_l2t; a.ld_l2t = d->ld_l2t;
    return launch_tail(a, d->planes, cofi_s(stream), d->w_frag != 0);
}

extern "C" int cofi_loftr_tail_bf16x3(const float *msg, int ldm, const float *x, int ldx, const uint16_t *wm_hi, const uint16_t *wm_lo,
                                      const float *n1_gamma, const float *n1_beta, const uint16_t *w0_hi, const uint16_t *w0_lo,
                                      const uint16_t *w2_hi, const uint16_t *w2_lo, const float *n2_gamma, const float *n2_beta, float eps,
                                      float *out, int ldo, int L, cofi_stream_t stream) {
    if (!msg || !x || !wm_hi || !wm_lo || !w0_hi || !w0_lo || !w2_hi || !w2_lo || !n1_gamma || !n1_beta || !n2_gamma || !n2_beta || !out)
        return COFI_EINVAL;
    if (L <= 0 || (ldm & 3) || (ldx & 3) || ldm < C || ldx < C || ldo < C || ((uintptr_t)msg & 15) || ((uintptr_t)x & 15)) return COFI_EINVAL;
    if (((uintptr_t)wm_hi | (uintptr_t)wm_lo | (uintptr_t)w0_hi | (uintptr_t)w0_lo | (uintptr_t)w2_hi | (uintptr_t)w2_lo) & 15) return COFI_EINVAL;
    TailArgs a{};
    a.msg = msg; a.x = x; a.wm[0] = wm_hi; a.wm[1] = wm_lo; a.w0[0] = w0_hi; a.w0[1] = w0_lo; a.w2[0] = w2_hi; a.w2[1] = w2_lo;
    a.n1g = n1_gamma; a.n1b = n1_beta; a.n2g = n2_gamma; a.n2b = n2_beta; a.out = out; a.ldm = ldm; a.ldx = ldx; a.ldo = ldo; a.L = L; a.eps = eps;
    return launch_tail(a, 2, cofi_s(stream));
}

extern "C" int cofi_loftr_tail_parts_bf16x3(const void *parts, size_t parts_bytes, int L, int S, int H, int frames, const float *x, int ldx,
                                            const uint16_t *wm_hi, const uint16_t *wm_lo, const float *n1_gamma, const float *n1_beta,
                                            const uint16_t *w0_hi, const uint16_t *w0_lo, const uint16_t *w2_hi, const uint16_t *w2_lo,
                                            const float *n2_gamma, const float *n2_beta, float eps, float *out, int ldo, cofi_stream_t stream) {
    if (!parts || !x || !wm_hi || !wm_lo || !w0_hi || !w0_lo || !w2_hi || !w2_lo || !n1_gamma || !n1_beta || !n2_gamma || !n2_beta || !out)
        return COFI_EINVAL;
    if (L <= 0 || S <= 0 || frames <= 0 || H * 32 != C || (ldx & 3) || ldx < C || ldo < C || ((uintptr_t)x & 15) || ((uintptr_t)parts & 15)) return COFI_EINVAL;
    if (((uintptr_t)wm_hi | (uintptr_t)wm_lo | (uintptr_t)w0_hi | (uintptr_t)w0_lo | (uintptr_t)w2_hi | (uintptr_t)w2_lo) & 15) return COFI_EINVAL;
    const AttnLayout lay = attn_layout(L, S, H, frames);
    if (parts_bytes < lay.bytes) return COFI_EWORKSPACE;
    TailArgs a{};
    a.x = x; a.wm[0] = wm_hi; a.wm[1] = wm_lo; a.w0[0] = w0_hi; a.w0[1] = w0_lo; a.w2[0] = w2_hi; a.w2[1] = w2_lo;
    a.n1g = n1_gamma; a.n1b = n1_beta; a.n2g = n2_gamma; a.n2b = n2_beta; a.out = out; a.ldx = ldx; a.ldo = ldo; a.L = L * frames; a.eps = eps;
    a.parts = (const float *)parts; a.lay = lay; a.Lf = L; a.H = H;
    return launch_tail(a, 2, cofi_s(stream));
}
