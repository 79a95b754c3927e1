// Flash-style multi-head attention for the I2P transformer on fp32 MFMA (v_mfma_f32_32x32x2_f32).
// Reference: model/transformer/linear_attention.py:56-79 (FullAttention) + the token-axis query
// normalisation of model/transformer/transformer.py:53, folded in as a per-channel scale.
//
// WORK PARTITION.  A unit = one 32-query block x one 32-key block of one (frame, head): 16 + 16 MFMAs on 8 KB of K / V.  A
// compute unit can pull ~11-16 bytes per clock from L2, its four matrix pipes consume a unit per 512 clocks: a wave that streams
// its own K / V blocks from L2 is bound by that fill rate (measured: 17.5 us for one KITTI cross-attention call whose MFMAs
// take 8.5 us on the busiest SIMD).  So a workgroup = 8 waves = TWO query blocks (64 queries of one frame and head) x FOUR
// key-block phases: every step it stages four K and four V tiles in LDS once (coalesced 16-byte loads, register-staged one step
// ahead and written after the step's MFMAs were issued) and each tile feeds both query blocks - half the L2 traffic per flop.
// One KITTI call has only 80 (frame, head, 64-query block) combinations for 256 CUs, so the P = ceil(S/32) key blocks are cut
// into KS contiguous ranges (attn_layout: KS = 3 there, 240 workgroups of 13-14 key blocks = 7 units per SIMD), each handled by
// its own workgroup; the dispatcher's round-robin over the 8 XCDs is undone so that an XCD works through neighbouring query
// blocks of (mostly) ONE head and its L2 pulls only that head's K / V.  The KS partial results of a query block (running max
// m, row sum l, un-normalised O) go to a slot table (attention_parts.h) and are combined by the consumer of the attention
// output - the fused layer tail (transformer_tail.hip) - or by attention_merge_kernel: a kernel boundary, not an in-launch
// hand-off, orders producer and consumer.  Everything is a fixed-order reduction: bit-reproducible.
//
// Per unit and wave:
//   S^T[key, q] = K_blk . Q^T    16 MFMA, A = K tile rows from LDS (ds_read_b128, row stride 36 dwords: conflict free), B = Q regs
//   online softmax, per lane = per query column (q = lane&31), 16 keys per lane, halves joined by
//   one __shfl_xor(.., 32)
//   O^T[d, q] += V_blk^T . P^T   16 MFMA, A = V[key, d = lane&31] from the LDS tile, B = P
// The MFMA D layout of S^T (lane (q,h) holds keys (r&3)+8(r>>2)+4h) is exactly the B-operand layout
// the second chain needs once the contraction index is allowed to run in that (permuted) key order,
// so P never moves between lanes and O^T keeps "one query per lane": the softmax rescale and the
// final 1/l are plain per-lane multiplies.  The wave issues the QK^T chain of step t before the softmax / PV of step t-1 (whose
// V operand it keeps in registers), so the softmax VALU work runs under matrix-core time.
#include "attention_parts.h"
#include "bf16_split.h"
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int D = 32;                    // head dimension
constexpr int QG = COFI_ATTN_QG;         // query blocks per workgroup
constexpr int KPH = COFI_ATTN_KPH;       // key-block phases per workgroup
constexpr int NW = QG * KPH;             // waves
constexpr int TLD = D + 4;               // LDS tile row stride (floats)
constexpr int TILE = 32 * TLD;           // floats per staged tile

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
    const unsigned char *kvimg;   // bf16x6 kernel only: K / V pre-split by cofi_attention_kv_planes (then K == V == nullptr); attention_x6.inc
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

// LIGHT: two workgroups per CU (<= 128 VGPRs): a wave runs QK^T, softmax and PV of a unit back to back - with four waves per SIMD
// the other waves' MFMAs cover its softmax - and one workgroup's prologue / merge runs under the other's MFMA phase.
template <bool LIGHT>
__global__ __launch_bounds__(64 * NW, LIGHT ? 4 : 2) void attention_flat_kernel(AttnArgs a) {
    // LDS carve (floats): two staging buffers of KPH K tiles + KPH V tiles | running max / row sum of the waves | Q scale | fold
    // scratch.  The waves' final O states (s_o) re-use the staging buffers: they are written after the loop's last barrier.  78 KB
    // in all, so workgroups of other kernels (the other frame streams' GEMMs) still fit on the CU next to this one.
    constexpr int BUF = 2 * KPH * TILE, SO = NW * 1024, SM = NW * 32, SP = 2 * NW * 32;
    static_assert(SO <= 2 * BUF, "the final states must fit in the staging buffers");
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF + 2 * SM + 32 + SP];
    float *s_buf = lds, *s_o = lds, *s_m = s_buf + 2 * BUF, *s_l = s_m + SM, *s_qs = s_l + SM, *s_part = s_qs + 32;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int ph = wave >> 1, qb_w = wave & 1;   // waves w and w + 4 (phases p and p + 2) usually share a SIMD
    const AttnLayout &lay = a.lay;
    const int lb = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const int sp = lb / lay.KS, ks = lb - sp * lay.KS;
    const int fh = sp / lay.QSB, qsb = sp - fh * lay.QSB;
    const int f = fh / a.H, h = fh - f * a.H;
    // key range ks of KS: base blocks each, the first `rem` ranges one more
    const int base = lay.P / lay.KS, rem = lay.P - base * lay.KS;
    const int kb0 = ks * base + min(ks, rem), nblk = base + (ks < rem ? 1 : 0);
    const int nsteps = (nblk + KPH - 1) / KPH;
    const float *Kf = a.K + (size_t)f * a.S * a.ldk + h * D, *Vf = a.V + (size_t)f * a.S * a.ldv + h * D;

    // ---- staging: thread t moves float4 number t + 512 i (i < 4) of a step's image: [K tiles | V tiles] x KPH x 32 rows x 8 chunks
    f32x4 stg[4];
    auto stage_load = [&](int step) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 64 * NW * i;
            const int isv = e >> 10, blk = (e >> 8) & 3, r = (e >> 3) & 31, c4 = e & 7;
            const int kb = min(kb0 + step * KPH + blk, kb0 + nblk - 1);        // blocks past the range: re-read the last one (unused)
            const int row = min(kb * 32 + r, a.S - 1);                          // rows past S: masked in the softmax
            const float *src = isv ? Vf + (size_t)row * a.ldv : Kf + (size_t)row * a.ldk;
            stg[i] = *reinterpret_cast<const f32x4 *>(src + 4 * c4);
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 64 * NW * i;
            const int tile = e >> 8, r = (e >> 3) & 31, c4 = e & 7;             // tile = isv * KPH + blk
            *reinterpret_cast<f32x4 *>(s_buf + buf * BUF + tile * TILE + r * TLD + 4 * c4) = stg[i];
        }
    };

    // ---- prologue: everything that goes to memory is issued before the first wait
    stage_load(0);
    f32x4 qraw[4];   // raw Q fragment: lane (q = li, h) holds Q[q][8c+4h+e]
    const int qblk = qsb * QG + qb_w;            // this wave's query block; past the last one: idle wave (clamped loads, no slot)
    {
        const int q = min(qblk * 32 + li, a.L - 1);
        const float *qp = a.Q + ((size_t)f * a.L + q) * a.ldq + h * D + 4 * lh;
#pragma unroll
        for (int c = 0; c < 4; ++c) qraw[c] = *reinterpret_cast<const f32x4 *>(qp + 8 * c);
    }
    // token-axis norm of this head's 32 Q columns (transformer.py:53) from the projection's column partials: thread
    // (phase, column) sums the squares of slabs phase, phase + 2 NW, ...; column threads fold the phases in a fixed order
    {
        const int col = tid & 31, php = tid >> 5;   // 2 NW phases
        if (a.q_colpart) {
            const float *cp = a.q_colpart + ((size_t)f * a.q_nslab * a.q_ncols + h * D + col) * 2 + 1;
            float acc = 0.f;
            for (int b = php; b < a.q_nslab; b += 2 * NW) acc += cp[(size_t)b * a.q_ncols * 2];
            s_part[php * 32 + col] = acc;
        }
        stage_store(0);
        if (nsteps > 1) stage_load(1);
        __syncthreads();
        if (tid < 32) {
            float v = 1.0f;
            if (a.q_colpart) {
                float t = 0.f;
#pragma unroll
                for (int p = 0; p < 2 * NW; ++p) t += s_part[p * 32 + tid];
                v = 1.0f / fmaxf(sqrtf(t), a.q_eps);
            } else if (a.qs) {
                v = a.qs[((size_t)f * a.H + h) * D + tid];
            }
            s_qs[tid] = v;
        }
        __syncthreads();
    }
    float qf[16];   // q * colscale * softmax scale * log2(e)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const f32x4 sc = *reinterpret_cast<const f32x4 *>(&s_qs[4 * lh + 8 * c]);
#pragma unroll
        for (int e = 0; e < 4; ++e) qf[4 * c + e] = (qraw[c][e] * sc[e]) * a.scale_log2e;
    }

    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    auto qk = [&](const float *kt) {  // S^T = K . Q^T, 16 chained MFMAs; lane (key = li, h) reads K[key][4h + 8c ..]
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 kf = *reinterpret_cast<const f32x4 *>(kt + li * TLD + 4 * lh + 8 * c);
#pragma unroll
            for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[4 * c + e], s, 0, 0, 0);
        }
        return s;
    };
    auto read_v = [&](const float *vt, float(&vf)[16]) {   // lane (d = li, h): V[keyrow(r, h)][d]
#pragma unroll
        for (int r = 0; r < 16; ++r) vf[r] = vt[((r & 3) + 8 * (r >> 2) + 4 * lh) * TLD + li];
    };
    auto mask_tail = [&](int kb, f32x16 &s) {   // keys past S (last block of a pair only) never win the max and weigh 0
        const int k0 = kb * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (k0 + (r & 3) + 8 * (r >> 2) + 4 * lh >= a.S) s[r] = -INFINITY;
    };
    auto softmax = [&](f32x16 &s) {  // online softmax (base 2); returns P in s, rescales o
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

    // ---- main loop: one unit per wave and step.  Step t: the QK^T chain of this step's tile and the softmax of step t-1 are
    // ONE basic block in which the scheduler is told to put the softmax's VALU / transcendental instructions into the shadows
    // of the QK^T MFMAs (64 cycles each: the wave would otherwise sit through 16 of them before its first exp); the PV chain of
    // step t-1 follows (its V operand waits in registers), then the next step's tiles go from registers to the other buffer
    // and the loads of the step after that are issued.  One barrier per step: buffer (t+1)&1 was last read in step t-1.
    // Steps without a tile for this wave (the tail of the range) and the first step take the plain paths below.
    const bool tail_keys = (a.S & 31) != 0;   // the last key block of every pair is partial
    if constexpr (LIGHT) {
        for (int t = 0; t < nsteps; ++t) {
            const float *bt = s_buf + (t & 1) * BUF;
            const int kb = kb0 + t * KPH + ph;
            if (t * KPH + ph < nblk) {   // wave-uniform
                f32x16 sC = qk(bt + ph * TILE);
                float vC[16];
                read_v(bt + (KPH + ph) * TILE, vC);
                if (tail_keys && kb == lay.P - 1) mask_tail(kb, sC);
                softmax(sC);
                pv(sC, vC);
            }
            if (t + 1 < nsteps) {
                stage_store((t + 1) & 1);
                if (t + 2 < nsteps) stage_load(t + 2);
            }
            __syncthreads();
        }
    } else {
    f32x16 sC;
    float vC[16];
    int kbC = -1;   // key block of the pending unit (wave-uniform), -1: none
    for (int t = 0; t < nsteps; ++t) {
        const float *bt = s_buf + (t & 1) * BUF;
        const int kb = kb0 + t * KPH + ph;
        const bool have = t * KPH + ph < nblk;   // wave-uniform
        if (have && kbC >= 0) {
            if (tail_keys && kbC == lay.P - 1) mask_tail(kbC, sC);
            f32x16 sN = qk(bt + ph * TILE);
            softmax(sC);
            // 16 x { 1 MFMA, then VALU work of the softmax }: exp / max / sub / mul go under the matrix pipe's 64-cycle instructions
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
            }
            float vN[16];
            read_v(bt + (KPH + ph) * TILE, vN);
            pv(sC, vC);
            sC = sN;
#pragma unroll
            for (int r = 0; r < 16; ++r) vC[r] = vN[r];
            kbC = kb;
        } else {
            if (kbC >= 0) {
                if (tail_keys && kbC == lay.P - 1) mask_tail(kbC, sC);
                softmax(sC);
                pv(sC, vC);
                kbC = -1;
            }
            if (have) {
                sC = qk(bt + ph * TILE);
                read_v(bt + (KPH + ph) * TILE, vC);
                kbC = kb;
            }
        }
        if (t + 1 < nsteps) {
            stage_store((t + 1) & 1);
            if (t + 2 < nsteps) stage_load(t + 2);
        }
        __syncthreads();
    }
    if (kbC >= 0) {
        if (tail_keys && kbC == lay.P - 1) mask_tail(kbC, sC);
        softmax(sC);
        pv(sC, vC);
    }
    }

    // ---- this wave's state -> LDS; the KPH phase waves of a query block are merged in a fixed order -> the block's slot
    l_run += __shfl_xor(l_run, 32, 64);   // join the two halves' row sums (same m in both halves by construction)
    if (lh == 0) {
        s_m[wave * 32 + li] = m_run;      // a wave without a unit leaves (-1e30, 0): contributes nothing
        s_l[wave * 32 + li] = l_run;
    }
#pragma unroll
    for (int rq = 0; rq < 4; ++rq)   // lane (q, h) holds O[q][8 rq + 4 h .. +3] in regs 4rq .. 4rq+3: chunk 2 rq + h
        *reinterpret_cast<float4 *>(&s_o[wave * 1024 + so_off(li, 2 * rq + lh)]) = make_float4(o[4 * rq], o[4 * rq + 1], o[4 * rq + 2], o[4 * rq + 3]);
    __syncthreads();
    {   // 512 threads = QG query blocks x 32 queries x 8 float4 chunks
        const int qb = tid >> 8, q = (tid >> 3) & 31, ch = tid & 7;
        const int blk = qsb * QG + qb;
        if (blk < lay.QB) {
            float mm = -1e30f;
#pragma unroll
            for (int p = 0; p < KPH; ++p) mm = fmaxf(mm, s_m[(p * QG + qb) * 32 + q]);
            float l = 0.f;
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int p = 0; p < KPH; ++p) {
                const int w = p * QG + qb;
                const float sc = fast_exp2(s_m[w * 32 + q] - mm);
                l += s_l[w * 32 + q] * sc;
                const float4 t = *reinterpret_cast<const float4 *>(&s_o[w * 1024 + so_off(q, ch)]);
                r.x += t.x * sc; r.y += t.y * sc; r.z += t.z * sc; r.w += t.w * sc;
            }
            const int pair = fh * lay.QB + blk;
            float *slot = a.parts + ((size_t)pair * lay.KS + ks) * COFI_ATTN_SLOT_FLOATS;
            if (ch == 0) {
                slot[q] = mm;
                slot[32 + q] = l;
            }
            *reinterpret_cast<float4 *>(slot + 64 + q * 32 + 4 * ch) = r;
        }
    }
}

#include "attention_x6.inc"

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

#ifdef COFI_ATTN_ABLATION
int g_x6_dbg = 0;   // timing experiments (attention_x6.inc DBG; tools/attn_ablate.py): builds without one part of the kernel
#endif

int launch_parts(AttnArgs a, int frames, bool x6, hipStream_t stream) {
    a.lay = attn_layout(a.L, a.S, a.H, frames);
    const dim3 grid(a.lay.nwg), block(64 * NW);
    if (x6) {
#ifdef COFI_ATTN_ABLATION
        switch (g_x6_dbg) {   // wrong results by construction
        case 0: break;
#define COFI_X6_DBG(D) case D: hipLaunchKernelGGL((attention_x6_kernel<D, false>), grid, block, 0, stream, a); return cofi_launch_status();
        COFI_X6_DBG(1) COFI_X6_DBG(2) COFI_X6_DBG(3) COFI_X6_DBG(4) COFI_X6_DBG(8) COFI_X6_DBG(9) COFI_X6_DBG(16) COFI_X6_DBG(32) COFI_X6_DBG(11) COFI_X6_DBG(15)
#undef COFI_X6_DBG
        default: return COFI_EINVAL;
        }
#endif
        if (a.kvimg) hipLaunchKernelGGL((attention_x6_kernel<0, true>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((attention_x6_kernel<0, false>), grid, block, 0, stream, a);
        return cofi_launch_status();
    }
    if (a.lay.light)
        hipLaunchKernelGGL((attention_flat_kernel<true>), grid, block, 0, stream, a);
    else
        hipLaunchKernelGGL((attention_flat_kernel<false>), grid, block, 0, stream, a);
    return cofi_launch_status();
}

}  // namespace

extern "C" size_t cofi_attention_workspace(int L, int S, int H, int D_, int frames) {
    if (L <= 0 || S <= 0 || H <= 0 || D_ != D || frames <= 0) return 0;
    return attn_layout(L, S, H, frames).bytes;
}

static int attention_parts_entry(const float *Q, int ldq, const float *K, int ldk, const float *V, int ldv, const float *q_colscale,
                                 const float *q_colpart, int q_nslab, int q_ncols, float q_eps, int L, int S, int H, int D_, float scale,
                                 int frames, void *parts, size_t parts_bytes, bool x6, cofi_stream_t stream) {
    if (int rc = attention_check(Q, ldq, K, ldk, V, ldv, L, S, H, D_, frames)) return rc;
    if (q_colscale && q_colpart) return COFI_EINVAL;
    if (q_colpart) {
        // the partials' 64-row slabs must be whole per frame
        // (64-row slabs of the GEMM epilogues, or the 32-row slabs of cofi_loftr_tail's fused projections: the fold only sums them)
        if (q_nslab <= 0 || (q_nslab % frames) || q_ncols < H * D) return COFI_EINVAL;
        const bool s64 = q_nslab / frames == cofi_cdiv(L, 64) && (frames == 1 || L % 64 == 0);
        const bool s32 = q_nslab / frames == cofi_cdiv(L, 32) && (frames == 1 || L % 32 == 0);
        if (!s64 && !s32) return COFI_EINVAL;
    }
    if (!parts || ((uintptr_t)parts & 15) || parts_bytes < attn_layout(L, S, H, frames).bytes) return COFI_EWORKSPACE;
    // the split-arithmetic kernel addresses K / V of one frame through buffer resources with 32-bit byte offsets (attention_x6.inc): a
    // frame whose K or V block reaches 4 GB runs on the fp32-instruction kernel, whose addressing is 64-bit
    if (x6 && ((size_t)S * ldk * sizeof(float) >= 0xffffffffull || (size_t)S * ldv * sizeof(float) >= 0xffffffffull)) x6 = false;
    AttnArgs a{Q, K, V, q_colscale, (float *)parts, ldq, ldk, ldv, L, S, H, scale * 1.4426950408889634f, q_colpart,
               q_colpart ? q_nslab / frames : 0, q_ncols, q_eps, {}, nullptr};
    return launch_parts(a, frames, x6, cofi_s(stream));
}

extern "C" int cofi_attention_parts(const float *Q, int ldq, const float *K, int ldk, const float *V, int ldv, const float *q_colscale,
                                    const float *q_colpart, int q_nslab, int q_ncols, float q_eps, int L, int S, int H, int D_, float scale,
                                    int frames, void *parts, size_t parts_bytes, cofi_stream_t stream) {
    return attention_parts_entry(Q, ldq, K, ldk, V, ldv, q_colscale, q_colpart, q_nslab, q_ncols, q_eps, L, S, H, D_, scale, frames, parts,
                                 parts_bytes, false, stream);
}

extern "C" int cofi_attention_parts_bf16x6(const float *Q, int ldq, const float *K, int ldk, const float *V, int ldv, const float *q_colscale,
                                           const float *q_colpart, int q_nslab, int q_ncols, float q_eps, int L, int S, int H, int D_,
                                           float scale, int frames, void *parts, size_t parts_bytes, cofi_stream_t stream) {
    return attention_parts_entry(Q, ldq, K, ldk, V, ldv, q_colscale, q_colpart, q_nslab, q_ncols, q_eps, L, S, H, D_, scale, frames, parts,
                                 parts_bytes, true, stream);
}

extern "C" size_t cofi_attention_kv_planes_bytes(int S, int H, int D_, int frames) {
    if (S <= 0 || H <= 0 || D_ != D || frames <= 0) return 0;
    return (size_t)frames * H * cofi_cdiv(S, 32) * x6::IMG_BLOCK;
}

extern "C" int cofi_attention_kv_planes(const float *K, int ldk, const float *V, int ldv, int S, int H, int D_, int frames, void *planes, size_t planes_bytes,
                                        cofi_stream_t stream) {
    if (!K || !V || S <= 0 || H <= 0 || frames <= 0) return COFI_EINVAL;
    if (D_ != D) return COFI_EUNSUPPORTED;
    if ((ldk & 3) || ldk < H * D || ldv < H * D || ((uintptr_t)K & 15)) return COFI_EINVAL;
    if (!planes || ((uintptr_t)planes & 15) || planes_bytes < cofi_attention_kv_planes_bytes(S, H, D_, frames)) return COFI_EWORKSPACE;
    const int P = cofi_cdiv(S, 32);
    if (H > 65535 || frames > 65535) return COFI_EUNSUPPORTED;
    hipLaunchKernelGGL(attention_kv_planes_kernel, dim3(P, H, frames), dim3(256), 0, cofi_s(stream), K, ldk, V, ldv, S, H, (unsigned char *)planes, P);
    return cofi_launch_status();
}

extern "C" int cofi_attention_parts_planes(const float *Q, int ldq, const void *planes, size_t planes_bytes, const float *q_colscale, const float *q_colpart,
                                           int q_nslab, int q_ncols, float q_eps, int L, int S, int H, int D_, float scale, int frames, void *parts,
                                           size_t parts_bytes, cofi_stream_t stream) {
    if (!Q || !planes || L <= 0 || S <= 0 || H <= 0 || frames <= 0) return COFI_EINVAL;
    if (D_ != D) return COFI_EUNSUPPORTED;
    if ((ldq & 3) || ldq < H * D || ((uintptr_t)Q & 15) || ((uintptr_t)planes & 15)) return COFI_EINVAL;
    if (planes_bytes < cofi_attention_kv_planes_bytes(S, H, D_, frames)) return COFI_EWORKSPACE;
    if (q_colscale && q_colpart) return COFI_EINVAL;
    if (q_colpart) {
        if (q_nslab <= 0 || (q_nslab % frames) || q_ncols < H * D) return COFI_EINVAL;
        const bool s64 = q_nslab / frames == cofi_cdiv(L, 64) && (frames == 1 || L % 64 == 0);
        const bool s32 = q_nslab / frames == cofi_cdiv(L, 32) && (frames == 1 || L % 32 == 0);
        if (!s64 && !s32) return COFI_EINVAL;
    }
    if (!parts || ((uintptr_t)parts & 15) || parts_bytes < attn_layout(L, S, H, frames).bytes) return COFI_EWORKSPACE;
    AttnArgs a{Q, nullptr, nullptr, q_colscale, (float *)parts, ldq, 0, 0, L, S, H, scale * 1.4426950408889634f, q_colpart,
               q_colpart ? q_nslab / frames : 0, q_ncols, q_eps, {}, (const unsigned char *)planes};
    return launch_parts(a, frames, true, cofi_s(stream));
}

#ifdef COFI_ATTN_ABLATION
extern "C" int cofi_tune_attention_x6_debug(int flags) {   // timing experiments only: the results are wrong
    g_x6_dbg = flags;
    return 0;
}
#endif

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
