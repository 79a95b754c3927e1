// Fused tail of one LoFTR encoder layer (model/transformer/transformer.py:57-64) in ONE kernel:
//     m   = LayerNorm1( msg @ Wm^T )
//     h   = relu( [x | m] @ W0^T )
//     out = x + LayerNorm2( h @ W2^T )
// for d_model = 128 (the reference's only configuration, model/network.py:35).  At batch 1 the three GEMMs of this
// chain are 1280 x {128,256,128} problems: as separate launches each is bound by launch + memory latency, not by
// the matrix cores.  Here a workgroup owns 32 token rows end to end; the intermediates m, [x|m] and h never leave
// LDS, and the only global traffic is one read of msg/x, one write of out and the weight stream (served by L2).
//
// Arithmetic: 3-term bf16 split (hi*hi + hi*lo + lo*hi) on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, the
// same as cofi_gemm_f32 with COFI_GEMM_BF16X3.  Weights arrive PRE-SPLIT into bf16 hi / lo planes (packed once at
// load time), so a wave streams its B fragments straight from L2 into registers (each wave owns distinct weight
// rows: nothing to share through LDS) one 4-step chunk ahead of the MFMAs; activations are split once when they
// enter LDS and are shared by the four waves as A fragments.
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

struct TailArgs {
    const float *msg, *x;
    const unsigned short *wm_hi, *wm_lo, *w0_hi, *w0_lo, *w2_hi, *w2_lo;
    const float *n1g, *n1b, *n2g, *n2b;
    float *out;
    int ldm, ldx, ldo, L;
    float eps;
    // msg == nullptr: the attention output is still in the form the attention kernel left it - per-pair partial slots
    // (attention_parts.h) - and the loader below merges + normalises them on the fly (L = frames * Lf rows, H heads of 32)
    const float *parts;
    AttnLayout lay;
    int Lf, H;
};

__device__ __forceinline__ unsigned cvt_pk(float a, float b) {  // RNE, a -> low half
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void split_store4(const float4 v, unsigned char *hi_ptr, unsigned char *lo_ptr) {
    uint2 hi, lo;
    hi.x = cvt_pk(v.x, v.y);
    hi.y = cvt_pk(v.z, v.w);
    lo.x = cvt_pk(v.x - __uint_as_float(hi.x << 16), v.y - __uint_as_float(hi.x & 0xffff0000u));
    lo.y = cvt_pk(v.z - __uint_as_float(hi.y << 16), v.w - __uint_as_float(hi.y & 0xffff0000u));
    *reinterpret_cast<uint2 *>(hi_ptr) = hi;
    *reinterpret_cast<uint2 *>(lo_ptr) = lo;
}
__device__ __forceinline__ void split_store1(float v, unsigned char *hi_ptr, unsigned char *lo_ptr) {
    const unsigned h = cvt_pk(v, 0.f) & 0xffffu;
    const unsigned l = cvt_pk(v - __uint_as_float(h << 16), 0.f) & 0xffffu;
    *reinterpret_cast<unsigned short *>(hi_ptr) = (unsigned short)h;
    *reinterpret_cast<unsigned short *>(lo_ptr) = (unsigned short)l;
}

// One GEMM stage of a wave: acc[t] (32 rows x 32 cols, t < NT) += A(32 x K, bf16 planes in LDS) . W[nbase + 32t + (0..31), 0..K)^T.
// W planes are (N, K) bf16 row-major in global memory.  KSTEPS = K / 16, in chunks of CH = 4 steps; PF chunks of B fragments
// are in flight ahead of the one being multiplied.  The workgroup is alone on its CU (40-80 workgroups per launch) and a
// chunk is only ~0.2 us of MFMA work, so the stage is a chain of L2 round trips: PF = all chunks (one round trip per stage)
// where the registers allow it - one wave per SIMD may use the whole 512-entry register file.
template <int NT, int KSTEPS, int PF>
__device__ __forceinline__ void gemm_stage(f32x16 (&acc)[NT], const unsigned char *a_hi, const unsigned char *a_lo, int a_stride,
                                           const unsigned short *w_hi, const unsigned short *w_lo, int K, int nbase, int li, int lh) {
    constexpr int CH = 4;
    constexpr int NCH = KSTEPS / CH;
    constexpr int SLOTS = PF + 1 < NCH ? PF + 1 : NCH;
    Frag bh[SLOTS][NT][CH], bl[SLOTS][NT][CH];
    auto loadw = [&](int c, int slot) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const size_t roff = (size_t)(nbase + 32 * t + li) * K + 8 * lh;
#pragma unroll
            for (int s = 0; s < CH; ++s) {
                const int k0 = 16 * (c * CH + s);
                bh[slot][t][s].u = *reinterpret_cast<const uint4 *>(w_hi + roff + k0);
                bl[slot][t][s].u = *reinterpret_cast<const uint4 *>(w_lo + roff + k0);
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
        const unsigned char *ah = a_hi + aoff, *al = a_lo + aoff;
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            Frag fa_h, fa_l;
            fa_h.u = *reinterpret_cast<const uint4 *>(ah + (c * CH + s) * 32);
            fa_l.u = *reinterpret_cast<const uint4 *>(al + (c * CH + s) * 32);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_l.v, bh[c % SLOTS][t][s].v, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_h.v, bl[c % SLOTS][t][s].v, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_h.v, bh[c % SLOTS][t][s].v, acc[t], 0, 0, 0);
            }
        }
    }
}

__global__ __launch_bounds__(256) void loftr_tail_kernel(TailArgs a) {
    // LDS carve (bytes): msg planes 2*R*S128 | cat planes 2*R*S256 | h planes 2*R*S256 | fp32 staging R*FLD*4
    constexpr int OFF_MSG = 0, OFF_CAT = OFF_MSG + 2 * R * S128, OFF_H = OFF_CAT + 2 * R * S256, OFF_F = OFF_H + 2 * R * S256;
    __shared__ __attribute__((aligned(16))) unsigned char lds[OFF_F + R * FLD * 4];
    unsigned char *msg_hi = lds + OFF_MSG, *msg_lo = msg_hi + R * S128;
    unsigned char *cat_hi = lds + OFF_CAT, *cat_lo = cat_hi + R * S256;
    unsigned char *h_hi = lds + OFF_H, *h_lo = h_hi + R * S256;
    float *stage = reinterpret_cast<float *>(lds + OFF_F);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int r0 = blockIdx.x * R;

    // ---- stage 0: msg and x tiles -> bf16 hi/lo planes (32 lanes cover one 512-B row; 8 rows per pass)
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
            split_store4(mv, msg_hi + rl * S128 + lk * 2, msg_lo + rl * S128 + lk * 2);
            split_store4(xv, cat_hi + rl * S256 + lk * 2, cat_lo + rl * S256 + lk * 2);
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
        gemm_stage<1, 8, 2>(acc, msg_hi, msg_lo, S128, a.wm_hi, a.wm_lo, C, wave * 32, li, lh);
        acc_to_stage(acc[0], wave * 32);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int rl = wave * 8 + i;
        float m0, m1;
        row_layernorm(rl, a.n1g, a.n1b, m0, m1);
        split_store1(m0, cat_hi + rl * S256 + (C + lane) * 2, cat_lo + rl * S256 + (C + lane) * 2);
        split_store1(m1, cat_hi + rl * S256 + (C + 64 + lane) * 2, cat_lo + rl * S256 + (C + 64 + lane) * 2);
    }
    __syncthreads();

    // ---- stage 2: h = relu([x|m] @ W0^T) (K = 256, N = 256: two 32x32 tiles per wave) -> h planes
    {
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        gemm_stage<2, 16, 2>(acc, cat_hi, cat_lo, S256, a.w0_hi, a.w0_lo, 2 * C, wave * 64, li, lh);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (r & 3) + 8 * (r >> 2) + 4 * lh, col = wave * 64 + 32 * t + li;
                split_store1(fmaxf(acc[t][r], 0.f), h_hi + rl * S256 + col * 2, h_lo + rl * S256 + col * 2);
            }
    }
    __syncthreads();

    // ---- stage 3: o = h @ W2^T (K = 256, N = 128) -> LN2 -> + x -> out
    {
        f32x16 acc[1];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
        gemm_stage<1, 16, 4>(acc, h_hi, h_lo, S256, a.w2_hi, a.w2_lo, 2 * C, wave * 32, li, lh);
        acc_to_stage(acc[0], wave * 32);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int rl = wave * 8 + i, row = r0 + rl;
        float o0, o1;
        row_layernorm(rl, a.n2g, a.n2b, o0, o1);
        if (row < a.L) {
            const float *xr = a.x + (size_t)row * a.ldx;
            float *dst = a.out + (size_t)row * a.ldo;
            dst[lane] = xr[lane] + o0;
            dst[64 + lane] = xr[64 + lane] + o1;
        }
    }
}

void launch_tail(const TailArgs &a, int nwg, hipStream_t s) {
    hipLaunchKernelGGL(loftr_tail_kernel, dim3(nwg), dim3(256), 0, s, a);
}

}  // namespace

extern "C" int cofi_loftr_tail_bf16x3(const float *msg, int ldm, const float *x, int ldx, const uint16_t *wm_hi, const uint16_t *wm_lo,
                                      const float *n1_gamma, const float *n1_beta, const uint16_t *w0_hi, const uint16_t *w0_lo,
                                      const uint16_t *w2_hi, const uint16_t *w2_lo, const float *n2_gamma, const float *n2_beta, float eps,
                                      float *out, int ldo, int L, cofi_stream_t stream) {
    if (!msg || !x || !wm_hi || !wm_lo || !w0_hi || !w0_lo || !w2_hi || !w2_lo || !n1_gamma || !n1_beta || !n2_gamma || !n2_beta || !out)
        return COFI_EINVAL;
    if (L <= 0 || (ldm & 3) || (ldx & 3) || ldm < C || ldx < C || ldo < C || ((uintptr_t)msg & 15) || ((uintptr_t)x & 15)) return COFI_EINVAL;
    if (((uintptr_t)wm_hi | (uintptr_t)wm_lo | (uintptr_t)w0_hi | (uintptr_t)w0_lo | (uintptr_t)w2_hi | (uintptr_t)w2_lo) & 15) return COFI_EINVAL;
    TailArgs a{msg, x, wm_hi, wm_lo, w0_hi, w0_lo, w2_hi, w2_lo, n1_gamma, n1_beta, n2_gamma, n2_beta, out, ldm, ldx, ldo, L, eps, nullptr, {}, 0, 0};
    launch_tail(a, cofi_cdiv(L, R), cofi_s(stream));
    return cofi_launch_status();
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
    const int rows = L * frames;
    TailArgs a{nullptr, x, wm_hi, wm_lo, w0_hi, w0_lo, w2_hi, w2_lo, n1_gamma, n1_beta, n2_gamma, n2_beta, out, 0, ldx, ldo, rows, eps,
               (const float *)parts, lay, L, H};
    launch_tail(a, cofi_cdiv(rows, R), cofi_s(stream));
    return cofi_launch_status();
}
