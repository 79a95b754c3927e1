// Dense fp32 contraction on v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF/s chip peak) with fused epilogues.
//   C[m,n] = act( (sum_k A[m,k] * W[n,k]) / rowdiv[m] + bias[n] )                (plain)
//   C[m,:] = relu?( LayerNorm_n( ... ) * gamma + beta ) + res[m,:]               (fused LayerNorm, N <= 128)
//   colpart[slab, n] = { sum_m C[m,n], sum_m C[m,n]^2 } over the rows of a slab   (fused column statistics:
//        feeds stack-mode GroupNorm / InstanceNorm / the token-axis Q normalisation without another pass)
// Both operands are K-contiguous ("NT"): activations row-major, weights in nn.Linear's (N,K) layout — no
// packing of the reference's Linear weights is needed.
//
// Tile: BM x BN x 32 per workgroup of 4 waves (2x2), each wave TM x TN MFMA tiles of 32x32.
// LDS image: rows of 32 k-values padded to 36 floats.  A lane reads its MFMA operands as ONE
// ds_read_b128 per 4 MFMA steps: lane (i = l&31, h = l>>5) reads k = 8c+4h .. 8c+4h+3 of row i and
// uses element e in step (c,e).  The k-order seen by the MFMA chain is therefore a permutation
// (step (c,e) contracts k = 8c+e and 8c+4+e), identical for A and W, which a sum does not care
// about.  Row stride 36 dwords makes the b128 reads (16-lane groups, bank = dword mod 64) and the
// b128 staging writes (8-lane groups, bank mod 32) conflict free (MI355X_MICROARCH §LDS).
//
// Epilogue: the accumulator tile goes through LDS once (the operand buffers are dead by then), so rows
// are written back as whole 128-B+ segments by consecutive lanes and the row / column reductions of the
// fused epilogues are plain shuffles over row-major data.
//
// Deep-K / small-MN problems are split over K (gridDim.z) into a workspace and reduced in fixed
// order by splitk_epilogue_kernel (same row-wise epilogue): deterministic, no float atomics.
#include "common.h"
#include "stat_fold.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int BK = 32;
constexpr int LDS_LD = 36;

struct GemmArgs {
    const float *A, *W;
    float *C;
    const float *bias, *rowdiv;
    float *ws;
    float *colpart;                    // optional (nslab, N, 2)
    const float *ln_gamma, *ln_beta;   // optional fused LayerNorm over the N columns of a row
    const float *res;                  // optional residual added after the LayerNorm
    int lda, ldw, ldc, ldr, M, N, K, act, ksplit, kchunk, ln_relu;
    float ln_eps;
    int bf16x3;
    int wsplit;        // W is pre-split: W points at the bf16 hi plane (rows of ldw bf16), the lo plane starts w_lo_off elements later
    long w_lo_off;
    int asplit;        // A is pre-split as well (dense GEMM only): A points at the bf16 hi plane (M rows of lda bf16), lo plane a_lo_off elements later
    long a_lo_off;
    // implicit-GEMM convolution (cv_ks > 0): A is an NHWC map (H*W rows of lda floats), row m of the GEMM is
    // output pixel (m / Wo, m % Wo), column k is (tap = k / Cin, channel = k % Cin); K = ks*ks*Cin
    int cv_ks, cv_H, cv_W, cv_Cin, cv_Wo, cv_stride, cv_pad;
    int cv_Pout;  // output pixels per frame (stack mode: GEMM row m is frame m / cv_Pout, pixel m % cv_Pout)
    int xcd;      // 0: hardware tile order; 1 + log2(gridDim.x): XCD-contiguous tile order (gemm_block_id)
    unsigned xcd_rcp_gy;
    int l2n;          // COFI_GEMM_L2NORM: rows are L2-normalised after bias / residual / activation (a tile spans the whole row, N <= 128)
    int act_col0;     // the activation applies to output columns >= act_col0 only (two layers sharing one A operand: [skip | conv1])
    int stat_shift;   // colpart holds one entry per 2^stat_shift adjacent columns: (nslab, N >> stat_shift, 2)
    // pending normalisation of the A operand (stat_fold.h): the loader applies  a -> leaky(a * sc[c] + sh[c])  to every element
    // it stages, c = column of a dense A / input channel of a convolution; an.part == nullptr: none.  an_rows = GEMM rows per
    // frame (tiles never straddle frames: host-checked).
    NormSrc an;
    int an_rows;
    int f16;          // COFI_GEMM_F16X3: a launch that takes the 256 x 128 kernel runs gemm_f16_big_kernel (three fp16 products) instead of the six-product one
    const float *wscale;  // COFI_GEMM_W_F16PRE: the panel scales of the pre-split W (one per 128 rows), stored behind the (N, ldw) matrix; nullptr: W is fp32
    unsigned *fixflags;   // ... one word per workgroup of that launch, behind the split-K partials in the workspace: 1 = the repair launch computes this tile
    int dbg;          // cofi_tune_big_debug (include/cofi_hip_tune.h) of the calling thread; 0 on the product path.  Bit 64: the generic row-wise epilogue instead of the straight-line one (same bits)
};

// Tile coordinates of this workgroup.  The dispatcher hands workgroup L (x fastest, then y, then the K-split slice z) to XCD
// L % 8, so with e.g. 8 column tiles every XCD owns ONE column of tiles and pulls the whole A operand through its private L2
// (8x the algorithmic traffic).  xcd = 1 + log2(gridDim.x) re-numbers the workgroups so that each XCD works through one
// CONTIGUOUS eighth of that order: the column tiles of a row panel of A run on the same XCD at the same time and A crosses the
// fabric once; with split-K an XCD works on (part of) ONE K-slice and reads only that slice of W.
// Power-of-two gridDim.x and an exact multiply-high reciprocal of gridDim.y: no division in the prologue of latency-bound launches.
struct BlockId { int x, y, z; };
__device__ __forceinline__ BlockId gemm_block_id(const GemmArgs &g) {
    if (!g.xcd) return {(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
    const int sh = g.xcd - 1, gy = gridDim.y, nblk = (gy * (int)gridDim.z) << sh;
    const int l = ((blockIdx.z * gy + blockIdx.y) << sh) + blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, c = l & 7, i = l >> 3;   // bijective for any block count
    const int t = (c < r ? c * (q + 1) : r * (q + 1) + (c - r) * q) + i;
    const int yz = t >> sh;
    const int z = g.xcd_rcp_gy ? (int)__umulhi((unsigned)yz, g.xcd_rcp_gy) : yz;   // rcp = ceil(2^32 / gy), exact for yz, gy < 2^16; 0: gy = 1
    return {t & ((1 << sh) - 1), yz - z * gy, z};
}

// A-operand tile loader shared by both MFMA kernels: float4 number j of this thread covers row m0 + lrow + 32j,
// k .. k+3.  Dense: A[row, k].  Convolution: the input pixel under tap k / Cin of output pixel `row`, zero
// outside the image (Cin % 4 == 0, so a float4 never straddles two taps).
template <int A_LD4>
struct ATileLoader {
    int yo[A_LD4], xo[A_LD4], fb[A_LD4];  // output pixel coordinates and first input row of the frame
    __device__ __forceinline__ void init(const GemmArgs &g, int m0, int lrow) {
        if (g.cv_ks) {
#pragma unroll
            for (int j = 0; j < A_LD4; ++j) {
                const int r = min(m0 + lrow + 32 * j, g.M - 1);
                const int f = r / g.cv_Pout, pix = r - f * g.cv_Pout;
                fb[j] = f * g.cv_H * g.cv_W;
                yo[j] = pix / g.cv_Wo;
                xo[j] = pix - yo[j] * g.cv_Wo;
            }
        }
    }
    // Every load is UNCONDITIONAL on a clamped (always valid) address and masked afterwards with a select: predicated
    // loads compile to one exec-masked basic block each, which stops the scheduler from batching the loads of a tile
    // and puts an s_waitcnt vmcnt(0) behind every one of them.
    __device__ __forceinline__ void load(const GemmArgs &g, int m0, int lrow, int k, bool kin, float4 (&ra)[A_LD4]) const {
        const float4 zero = make_float4(0, 0, 0, 0);
        const int kc = kin ? k : 0;
        if (g.cv_ks == 0) {
#pragma unroll
            for (int j = 0; j < A_LD4; ++j) {
                const int r = m0 + lrow + 32 * j;
                const float4 v = *reinterpret_cast<const float4 *>(g.A + (size_t)min(r, g.M - 1) * g.lda + kc);
                ra[j] = (kin && r < g.M) ? v : zero;
            }
        } else {
            const int tap = kc / g.cv_Cin, c = kc - tap * g.cv_Cin;
            const int dy = tap / g.cv_ks, dx = tap - dy * g.cv_ks;
#pragma unroll
            for (int j = 0; j < A_LD4; ++j) {
                const int r = m0 + lrow + 32 * j;
                const int yi = yo[j] * g.cv_stride - g.cv_pad + dy, xi = xo[j] * g.cv_stride - g.cv_pad + dx;
                const bool ok = kin && r < g.M && (unsigned)yi < (unsigned)g.cv_H && (unsigned)xi < (unsigned)g.cv_W;
                const int yc = min(max(yi, 0), g.cv_H - 1), xc = min(max(xi, 0), g.cv_W - 1);
                const float4 v = *reinterpret_cast<const float4 *>(g.A + ((size_t)fb[j] + (size_t)yc * g.cv_W + xc) * g.lda + c);
                ra[j] = ok ? v : zero;
            }
        }
    }
};

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == COFI_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == COFI_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    if (act == COFI_ACT_LEAKY01) return v >= 0.0f ? v : v * 0.1f;
    return v;
}

// apply_act for a workgroup-UNIFORM activation code: ReLU / LeakyReLU are computed branch-free and chosen with scalar-condition
// selects; only the sigmoid (score heads) sits behind a - uniform - branch.  Same expressions as apply_act: identical bits.
__device__ __forceinline__ float apply_act_uniform(float v, int act) {
    if (act == COFI_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    const float relu = fmaxf(v, 0.0f), leaky = v >= 0.0f ? v : v * 0.1f;
    return act == COFI_ACT_RELU ? relu : (act == COFI_ACT_LEAKY01 ? leaky : v);
}

// Column statistics of a slab: fold the per-thread sums of the RPP row phases in a fixed order (deterministic).  red = (RPP, BN, 2) floats
// of LDS that every thread has finished reading (the caller's barrier); all threads of the workgroup call.  One barrier inside.
template <int BN, int RPP>
__device__ __forceinline__ void colstat_fold(const GemmArgs &g, float *red, const float (&cs)[4], const float (&cq)[4], int slab, int n0, int tr, int tc) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[(tr * BN + 4 * tc + e) * 2 + 0] = cs[e];
        red[(tr * BN + 4 * tc + e) * 2 + 1] = cq[e];
    }
    __syncthreads();
    if (threadIdx.x < BN && n0 + (int)threadIdx.x < g.N) {
        float s = 0.f, q = 0.f;
        for (int p = 0; p < RPP; ++p) {
            s += red[(p * BN + threadIdx.x) * 2 + 0];
            q += red[(p * BN + threadIdx.x) * 2 + 1];
        }
        if (g.stat_shift) {
            // one table entry per 2^stat_shift adjacent columns (<= BN, aligned: a group never leaves the tile, and N is a
            // multiple of the width, so the lanes of a group are all active): fp64 butterfly, fixed order
            double ds = s, dq = q;
            for (int o = 1; o < (1 << g.stat_shift); o <<= 1) {
                ds += __shfl_xor(ds, o, 64);
                dq += __shfl_xor(dq, o, 64);
            }
            if ((threadIdx.x & ((1u << g.stat_shift) - 1)) == 0) {
                float *o = g.colpart + ((size_t)slab * (g.N >> g.stat_shift) + ((n0 + threadIdx.x) >> g.stat_shift)) * 2;
                o[0] = (float)ds;
                o[1] = (float)dq;
            }
        } else {
            float *o = g.colpart + ((size_t)slab * g.N + n0 + threadIdx.x) * 2;
            o[0] = s;
            o[1] = q;
        }
    }
}

// The straight-line form of the row-wise epilogue (no fused LayerNorm / L2 normalisation, whole float4 columns inside N, aligned output):
// x[p][e] = value of row row0 + p * RPP, column col + e.  Every option is a workgroup-uniform scalar and guards a whole BLOCK of the
// NP x 4 elements (a guard around a single element gets if-converted: the compiler then runs e.g. the 10-instruction division for every
// element and selects afterwards - measured: 20 us of epilogue for one 256 x 128 tile, 4500 cycles per 64-row slab).  Same operations in
// the same order per element as the generic path: identical bits.
template <int NP, int RPP, bool RES>
__device__ __forceinline__ void epi_fast_apply(const GemmArgs &g, float (&x)[NP][4], const float (&bias4)[4], const float (&rd)[NP],
                                               const float (&rs)[NP][4], int col, int row0, float (&cs)[4], float (&cq)[4]) {
    if (g.rowdiv) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int e = 0; e < 4; ++e) x[p][e] = x[p][e] / rd[p];
    }
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int e = 0; e < 4; ++e) x[p][e] = (x[p][e] + bias4[e]) + (RES ? rs[p][e] : 0.f);   // residual before the activation; the zero keeps -0.0 + 0.0 = +0.0 of the generic path
    const int act = g.act;
    if (act != COFI_ACT_NONE) {
        bool on[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) on[e] = col + e >= g.act_col0;
        if (act == COFI_ACT_RELU) {
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int e = 0; e < 4; ++e) x[p][e] = on[e] ? fmaxf(x[p][e], 0.0f) : x[p][e];
        } else if (act == COFI_ACT_LEAKY01) {
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int e = 0; e < 4; ++e) x[p][e] = (on[e] && !(x[p][e] >= 0.0f)) ? x[p][e] * 0.1f : x[p][e];
        } else {
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int e = 0; e < 4; ++e) x[p][e] = on[e] ? 1.0f / (1.0f + expf(-x[p][e])) : x[p][e];
        }
    }
    if (g.colpart) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
            if (row0 + p * RPP < g.M) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    cs[e] += x[p][e];
                    cq[e] += x[p][e] * x[p][e];
                }
            }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p)
        if (row0 + p * RPP < g.M) *reinterpret_cast<float4 *>(g.C + (size_t)(row0 + p * RPP) * g.ldc + col) = make_float4(x[p][0], x[p][1], x[p][2], x[p][3]);
}

// Row-wise epilogue over a ROWS x BN tile held row-major in LDS (`tile`, leading dimension TLD) or summed
// from split-K partials.  Thread (tr, tc): row phase tr, float4 column chunk tc.  NT threads.
// The operands it reads from global memory come in two groups that a caller working through several slabs of one tile may load AHEAD:
// column-indexed parameters (EpiCols: once per tile) and row-indexed ones (EpiRows: per slab).  vmcnt counts stores too, so a load
// issued behind the stores of the previous slab makes its consumer wait for those stores' round trip: a multi-slab epilogue issues the
// next slab's loads before the current slab's stores (gemm_x6_big_kernel).
struct EpiCols { float bias4[4], gam4[4], bet4[4]; };
template <int NP> struct EpiRows { float rs[NP][4], rd[NP]; };

template <int BN>
__device__ __forceinline__ EpiCols epi_load_cols(const GemmArgs &g, int n0) {
    constexpr int TPR = BN / 4;
    const int col = n0 + 4 * ((int)threadIdx.x % TPR);
    EpiCols c;
#pragma unroll
    for (int e = 0; e < 4; ++e) { c.bias4[e] = 0.f; c.gam4[e] = 1.f; c.bet4[e] = 0.f; }
    if (!g.bias && !g.ln_gamma) return c;   // uniform
    if (n0 + BN <= g.N) {                   // uniform: no column tail in this tile
        if (g.bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) c.bias4[e] = g.bias[col + e];
        }
        if (g.ln_gamma) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { c.gam4[e] = g.ln_gamma[col + e]; c.bet4[e] = g.ln_beta[col + e]; }
        }
        return c;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (col + e < g.N) {
            if (g.bias) c.bias4[e] = g.bias[col + e];
            if (g.ln_gamma) { c.gam4[e] = g.ln_gamma[col + e]; c.bet4[e] = g.ln_beta[col + e]; }
        }
    }
    return c;
}

template <int ROWS, int BN, int NT>
__device__ __forceinline__ EpiRows<ROWS / (NT / (BN / 4))> epi_load_rows(const GemmArgs &g, int m0, int n0) {
    constexpr int TPR = BN / 4, RPP = NT / TPR, NP = ROWS / RPP;
    const int tc = threadIdx.x % TPR, tr = threadIdx.x / TPR;
    const int col = n0 + 4 * tc;
    const bool res_vec = g.res && ((g.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.res) & 15) == 0) && col + 3 < g.N;
    EpiRows<NP> r;
    if (!g.rowdiv && !g.res) {   // uniform: nothing to load
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            r.rd[p] = 1.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e) r.rs[p][e] = 0.f;
        }
        return r;
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int row = m0 + p * RPP + tr;
        const bool rin = row < g.M;
        r.rd[p] = (g.rowdiv && rin) ? g.rowdiv[row] : 1.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) r.rs[p][e] = 0.f;
        if (g.res && rin) {
            const float *rp = g.res + (size_t)row * g.ldr + col;
            if (res_vec) {
                const float4 t = *reinterpret_cast<const float4 *>(rp);
                r.rs[p][0] = t.x; r.rs[p][1] = t.y; r.rs[p][2] = t.z; r.rs[p][3] = t.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (col + e < g.N) r.rs[p][e] = rp[e];
            }
        }
    }
    return r;
}

template <int ROWS, int BN, bool FROM_WS, int NT = 256>
__device__ __forceinline__ void rowwise_epilogue_pre(const GemmArgs &g, const float *tile, int TLD, int m0, int n0, int slab, float *red,
                                                     const EpiCols &ec, const EpiRows<ROWS / (NT / (BN / 4))> &er) {
    constexpr int TPR = BN / 4;        // threads per row
    constexpr int RPP = NT / TPR;      // rows per pass
    constexpr int NP = ROWS / RPP;     // passes
    const int tc = threadIdx.x % TPR, tr = threadIdx.x / TPR;
    const int col = n0 + 4 * tc;
    const bool vec_ok = ((g.N & 3) == 0) && ((g.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.C) & 15) == 0);
    float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
    const float (&bias4)[4] = ec.bias4, (&gam4)[4] = ec.gam4, (&bet4)[4] = ec.bet4;
    const float (&rs)[NP][4] = er.rs, (&rd)[NP] = er.rd;
    // phase 1: the tile values of all passes (split-K: every partial load of all passes is independent, so their latencies overlap)
    const bool fast = !g.ln_gamma && !g.l2n && vec_ok && n0 + BN <= g.N && !(g.dbg & 64);   // dbg 64 (tools only): the generic path
    float v[NP][4];
    if (FROM_WS || !fast) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int rl = p * RPP + tr, row = m0 + rl;
        const bool rin = row < g.M;
        if constexpr (FROM_WS) {
            v[p][0] = v[p][1] = v[p][2] = v[p][3] = 0.f;
            if (rin) {
                const size_t total = (size_t)g.M * g.N;
                for (int z = 0; z < g.ksplit; ++z) {
                    const float *q = g.ws + (size_t)z * total + (size_t)row * g.N + col;
                    if ((g.N & 3) == 0 && col < g.N) {
                        const float4 t = *reinterpret_cast<const float4 *>(q);
                        v[p][0] += t.x; v[p][1] += t.y; v[p][2] += t.z; v[p][3] += t.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (col + e < g.N) v[p][e] += q[e];
                    }
                }
            }
        } else {
            const float4 t = *reinterpret_cast<const float4 *>(tile + rl * TLD + 4 * tc);
            v[p][0] = t.x; v[p][1] = t.y; v[p][2] = t.z; v[p][3] = t.w;
        }
    }
    }
    // phase 2: arithmetic + stores.
    // FAST PATH - no fused LayerNorm / L2 normalisation, whole float4 columns inside N, aligned output: every decision is taken ONCE per
    // workgroup (uniform scalars), the loops below are straight-line code.  The generic path further down decides per element and costs
    // ~10 000 cycles per 64-row slab (measured on MI355X: 20 us of epilogue for one 256 x 128 tile), which was most of a short-K launch.
    // Same operations in the same order as the generic path: identical bits.
    if (fast) {
        // two passes at a time: blocks large enough not to be if-converted, small enough for the register budget of the kernels that
        // rely on two or three workgroups per CU (all NP x 4 values in flight through every stage cost them a workgroup); a tile in LDS
        // is read chunk by chunk
        constexpr int CH = NP >= 2 ? 2 : 1;
#pragma unroll
        for (int c = 0; c < NP / CH; ++c) {
            if constexpr (!FROM_WS) {
#pragma unroll
                for (int p = c * CH; p < (c + 1) * CH; ++p) {
                    const float4 t = *reinterpret_cast<const float4 *>(tile + (p * RPP + tr) * TLD + 4 * tc);
                    v[p][0] = t.x; v[p][1] = t.y; v[p][2] = t.z; v[p][3] = t.w;
                }
            }
            float (&vx)[CH][4] = reinterpret_cast<float (&)[CH][4]>(v[c * CH]);
            const float (&rdx)[CH] = reinterpret_cast<const float (&)[CH]>(rd[c * CH]);
            const float (&rsx)[CH][4] = reinterpret_cast<const float (&)[CH][4]>(rs[c * CH]);
            if (g.res) epi_fast_apply<CH, RPP, true>(g, vx, bias4, rdx, rsx, col, m0 + tr + c * CH * RPP, cs, cq);
            else epi_fast_apply<CH, RPP, false>(g, vx, bias4, rdx, rsx, col, m0 + tr + c * CH * RPP, cs, cq);
        }
    } else {
    // generic path (fused LayerNorm / L2 normalisation, column tails).  The row divisor guards the whole block of NP x 4 elements: around a
    // single element the branch is if-converted and the 10-instruction division runs for every element of every launch
    if (g.rowdiv) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[p][e] = v[p][e] / rd[p];
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int row = m0 + p * RPP + tr;
        const bool rin = row < g.M;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[p][e] = v[p][e] + bias4[e];
        if (g.ln_gamma) {
            // LayerNorm over the N (<= BN) columns of this row: the TPR threads of a row are adjacent lanes
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) s += (col + e < g.N) ? v[p][e] : 0.f;
#pragma unroll
            for (int o = 1; o < TPR; o <<= 1) s += __shfl_xor(s, o, 64);
            const float mean = s / (float)g.N;
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[p][e] - mean;
                q += (col + e < g.N) ? d * d : 0.f;
            }
#pragma unroll
            for (int o = 1; o < TPR; o <<= 1) q += __shfl_xor(q, o, 64);
            const float rstd = 1.0f / sqrtf(q / (float)g.N + g.ln_eps);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = (v[p][e] - mean) * rstd * gam4[e] + bet4[e];
                if (g.ln_relu) x = fmaxf(x, 0.f);
                v[p][e] = x + rs[p][e];
            }
        } else {
            // residual before the activation.  The activation code is workgroup-uniform, the columns it applies to (>= act_col0) are
            // a per-lane mask: a select, not a per-element divergent branch (three exec-masked regions per element made this loop
            // the longest part of a short-K launch)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = v[p][e] + rs[p][e];
                v[p][e] = (col + e >= g.act_col0) ? apply_act_uniform(x, g.act) : x;
            }
            if (g.l2n) {   // F.normalize(row, dim = 1): x / max(|x|, 1e-12); the TPR threads of a row are adjacent lanes
                float q = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) q += (col + e < g.N) ? v[p][e] * v[p][e] : 0.f;
#pragma unroll
                for (int o = 1; o < TPR; o <<= 1) q += __shfl_xor(q, o, 64);
                const float inv = 1.0f / fmaxf(sqrtf(q), 1e-12f);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[p][e] *= inv;
            }
        }
        if (rin) {
            if (g.colpart) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    cs[e] += v[p][e];
                    cq[e] += v[p][e] * v[p][e];
                }
            }
            float *dst = g.C + (size_t)row * g.ldc + col;
            if (vec_ok && col + 3 < g.N) {
                *reinterpret_cast<float4 *>(dst) = make_float4(v[p][0], v[p][1], v[p][2], v[p][3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (col + e < g.N) dst[e] = v[p][e];
            }
        }
    }
    }
    if (g.colpart) {
        __syncthreads();
        colstat_fold<BN, RPP>(g, red, cs, cq, slab, n0, tr, tc);
    }
}

template <int ROWS, int BN, bool FROM_WS, int NT = 256>
__device__ __forceinline__ void rowwise_epilogue(const GemmArgs &g, const float *tile, int TLD, int m0, int n0, int slab, float *red) {
    const EpiCols ec = epi_load_cols<BN>(g, n0);
    const auto er = epi_load_rows<ROWS, BN, NT>(g, m0, n0);
    rowwise_epilogue_pre<ROWS, BN, FROM_WS, NT>(g, tile, TLD, m0, n0, slab, red, ec, er);
}

template <int BM, int BN, int TM, int TN>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    static_assert(BM == 64 * TM && BN == 64 * TN, "2x2 waves");
    constexpr int A_LD4 = BM / 32;  // float4 loads per thread for the A tile
    constexpr int W_LD4 = BN / 32;
    constexpr int TLD = BN + 4;     // epilogue tile leading dimension
    static_assert(BM * TLD <= 2 * (BM + BN) * LDS_LD && (256 / (BN / 4)) * BN * 2 <= BM * TLD, "epilogue tile must fit in the operand buffers");
    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LDS_LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const BlockId bid = gemm_block_id(g);
    const int m0 = bid.y * BM, n0 = bid.x * BN;
    const int kbeg = bid.z * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int ntiles = (kend - kbeg + BK - 1) / BK;

    const int lrow = tid >> 3, lk = (tid & 7) * 4;  // staging map: 8 lanes cover one 128-B row slice
    float4 ra[A_LD4], rw[W_LD4];
    ATileLoader<A_LD4> aload;
    aload.init(g, m0, lrow);

    auto gload = [&](int t) {
        const int k = kbeg + t * BK + lk;
        const bool kin = k < kend;
        aload.load(g, m0, lrow, k, kin, ra);
#pragma unroll
        for (int j = 0; j < W_LD4; ++j) {
            const int r = n0 + lrow + 32 * j;
            const float4 v = *reinterpret_cast<const float4 *>(g.W + (size_t)min(r, g.N - 1) * g.ldw + (kin ? k : 0));
            rw[j] = (kin && r < g.N) ? v : make_float4(0, 0, 0, 0);
        }
    };
    auto sstore = [&](int buf) {
        float *as = lds + buf * (BM + BN) * LDS_LD, *bs = as + BM * LDS_LD;
#pragma unroll
        for (int j = 0; j < A_LD4; ++j) *reinterpret_cast<float4 *>(as + (lrow + 32 * j) * LDS_LD + lk) = ra[j];
#pragma unroll
        for (int j = 0; j < W_LD4; ++j) *reinterpret_cast<float4 *>(bs + (lrow + 32 * j) * LDS_LD + lk) = rw[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    if (ntiles > 0) {
        gload(0);
        sstore(0);
    }
    __syncthreads();

    const int li = lane & 31, lh = lane >> 5;
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) gload(t + 1);
        const float *as = lds + buf * (BM + BN) * LDS_LD + (wm * 32 * TM + li) * LDS_LD + 4 * lh;
        const float *bs = lds + buf * (BM + BN) * LDS_LD + BM * LDS_LD + (wn * 32 * TN + li) * LDS_LD + 4 * lh;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const float4 *>(as + i * 32 * LDS_LD + 8 * c);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const float4 *>(bs + j * 32 * LDS_LD + 8 * c);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (t + 1 < ntiles) sstore(buf ^ 1);
        __syncthreads();
    }

    if (g.ksplit > 1) {
        // raw partial sums straight from the D layout: row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * 32 * TN + j * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row < g.M && col < g.N) g.ws[((size_t)bid.z * g.M + row) * g.N + col] = acc[i][j][r];
                }
            }
        return;
    }
    if constexpr (BM == 128) {
        // two 64-row halves: the statistics slabs are 64 rows for every tile shape (see the bf16x3 kernel)
        for (int half = 0; half < 2; ++half) {
            if (m0 + 64 * half >= g.M) break;   // uniform: no second slab behind the last valid row
            if (wm == half) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int rl = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                            lds[rl * TLD + wn * 32 * TN + j * 32 + li] = acc[i][j][r];
                        }
            }
            __syncthreads();
            rowwise_epilogue<64, BN, false>(g, lds, TLD, m0 + 64 * half, n0, 2 * bid.y + half, lds);
            __syncthreads();
        }
        return;
    }
    // accumulators -> row-major LDS tile (the operand buffers are dead: the loop ended on a barrier)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                lds[rl * TLD + wn * 32 * TN + j * 32 + li] = acc[i][j][r];
            }
    __syncthreads();
    rowwise_epilogue<BM, BN, false>(g, lds, TLD, m0, n0, bid.y, lds);  // `red` aliases the tile: it is written after a barrier
}

// ------------------------------------------------------------------------------------------------------
// bf16x3 variant: every fp32 operand x is split on the fly into hi = bf16(x), lo = bf16(x - hi) and the product
// is formed as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: ~2^-16 relative
// error per product (the dropped lo*lo term) at 16/3 = 5.3x the fp32 MFMA rate.  Same tiling, staging map,
// split-K and epilogue as the fp32 kernel; LDS holds a hi and a lo plane per operand, rows of BK3 bf16 padded
// by 16 B (conflict-free 16-B fragment reads).  Selected with COFI_GEMM_BF16X3 in `act`; static operands (weights) may arrive
// pre-split (COFI_GEMM_W_SPLIT, cofi_split_bf16_planes).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {  // RNE, a -> low half
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void split4(const f32x4 v, uint2 &hi, uint2 &lo) {
    hi.x = cvt_pk_bf16(v[0], v[1]);
    hi.y = cvt_pk_bf16(v[2], v[3]);
    const float rx = v[0] - __uint_as_float(hi.x << 16), ry = v[1] - __uint_as_float(hi.x & 0xffff0000u);
    const float rz = v[2] - __uint_as_float(hi.y << 16), rw = v[3] - __uint_as_float(hi.y & 0xffff0000u);
    lo.x = cvt_pk_bf16(rx, ry);
    lo.y = cvt_pk_bf16(rz, rw);
}
// three planes: hi + mid + lo carry all 24 mantissa bits of an fp32 value (the bf16x6 arithmetic)
__device__ __forceinline__ void split4x3(const f32x4 v, uint2 &hi, uint2 &mid, uint2 &lo) {
    hi.x = cvt_pk_bf16(v[0], v[1]);
    hi.y = cvt_pk_bf16(v[2], v[3]);
    const float rx = v[0] - __uint_as_float(hi.x << 16), ry = v[1] - __uint_as_float(hi.x & 0xffff0000u);
    const float rz = v[2] - __uint_as_float(hi.y << 16), rw = v[3] - __uint_as_float(hi.y & 0xffff0000u);
    mid.x = cvt_pk_bf16(rx, ry);
    mid.y = cvt_pk_bf16(rz, rw);
    lo.x = cvt_pk_bf16(rx - __uint_as_float(mid.x << 16), ry - __uint_as_float(mid.x & 0xffff0000u));
    lo.y = cvt_pk_bf16(rz - __uint_as_float(mid.y << 16), rw - __uint_as_float(mid.y & 0xffff0000u));
}
__device__ __forceinline__ void split4(const float4 v, uint2 &hi, uint2 &lo) {
    hi.x = cvt_pk_bf16(v.x, v.y);
    hi.y = cvt_pk_bf16(v.z, v.w);
    const float rx = v.x - __uint_as_float(hi.x << 16), ry = v.y - __uint_as_float(hi.x & 0xffff0000u);
    const float rz = v.z - __uint_as_float(hi.y << 16), rw = v.w - __uint_as_float(hi.y & 0xffff0000u);
    lo.x = cvt_pk_bf16(rx, ry);
    lo.y = cvt_pk_bf16(rz, rw);
}

// Static operands (weights) are split once: planes (2, N, ldp) bf16 = [hi | lo], rows zero-padded to ldp (multiple of 8).
// Same rounding as split4, so a pre-split launch is bit-identical to splitting on the fly.
__global__ void split_planes_kernel(const float *W, int ldw, int N, int K, unsigned *planes, int ldp, int nplanes) {
    const int hp = ldp >> 1;   // bf16 pairs per row
    const size_t total = (size_t)N * hp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / hp), k = 2 * (int)(i - (size_t)n * hp);
        const float a = k < K ? W[(size_t)n * ldw + k] : 0.f, b = k + 1 < K ? W[(size_t)n * ldw + k + 1] : 0.f;
        const unsigned hi = cvt_pk_bf16(a, b);
        const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
        const unsigned lo = cvt_pk_bf16(ra, rb);   // two planes: lo; three planes: mid
        planes[i] = hi;
        planes[total + i] = lo;
        if (nplanes == 3) planes[2 * total + i] = cvt_pk_bf16(ra - __uint_as_float(lo << 16), rb - __uint_as_float(lo & 0xffff0000u));
    }
}

// K-tile of 128 per iteration (BK3): at batch 1 most problems give <= 1 workgroup per CU, so the loop is bound by
// the L2/HBM round trip per k-tile, not by the matrix cores; a 4x deeper tile keeps 4x the bytes in flight per wave
// and needs 4x fewer barriers.  One LDS buffer + a register-staged next tile:
//   issue global loads (t+1) -> MFMAs on tile t from LDS -> barrier -> split + write tile t+1 -> barrier.
// (The 128x128 tile uses BK3 = 64: 128 would need 139 KB of LDS and the whole VGPR file for one workgroup.)
// 4 waves (2x2 over the tile), up to 2-3 workgroups per CU.  (An 8 / 16-wave variant that split every K-tile over wave groups shortened a
// LONE launch but cost throughput with frames in flight - 563 vs 583 frames/s - and was removed in round 3)
// ANORM: the A operand carries a pending GroupNorm / InstanceNorm (+ affine + LeakyReLU) of the producing layer (GemmArgs::an):
//        the workgroup folds the producer's statistics partials itself while its first tile is in flight, keeps the per-channel
//        scale / shift in LDS and normalises every A element on its way into the bf16 planes - the stand-alone normalisation
//        kernel, its statistics kernel and one round trip of the activation through HBM disappear.
// ASPLIT: the A operand arrives PRE-SPLIT as well (bf16 hi / lo planes written by its producer - the KPConv aggregation, whose
//        (M, 15 C) output has no other reader): A tiles then travel global -> LDS as plain 16-byte copies like the W tiles, the
//        conversion instructions (the bulk of the VALU work of a tile) and their registers disappear.  Dense GEMM only.
//        (An earlier experiment in this slot - two K-tiles in flight in registers - measured 445 vs 457 frames/s and was removed.)
// X6 (COFI_GEMM_BF16X6): THREE bf16 planes per operand - hi + mid + lo hold all 24 mantissa bits of an fp32 value - and the six
//        products hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi (what is dropped is below 2^-24 of |a||b|): fp32-grade results
//        (the error is that of the fp32 accumulation, as in the exact-fp32 kernel) at 16/6 = 2.7x the fp32 MFMA rate.  Operands are
//        split on the fly (no pre-split planes); the normalising loader (ANORM) works as in the 3-term kernel.
template <int BM, int BN, int TM, int TN, int BK3, bool WSPLIT = false, int WPE = 1, bool ANORM = false, bool ASPLIT = false, bool X6 = false>
__global__ __launch_bounds__(256, WPE) void gemm_bf16x3_kernel(GemmArgs g) {
    static_assert(BM == 64 * TM && BN == 64 * TN, "2x2 waves");
    static_assert(!X6 || !ASPLIT, "bf16x6: A is split on the fly (W may arrive as three pre-split planes)");
    constexpr int NPL = X6 ? 3 : 2;   // bf16 planes per operand
    constexpr int NT = 256;
    // bytes per LDS row.  64- / 128-deep K-tiles: bf16 values + 16 B pad (36 / 68 dwords: conflict-free b128 reads, and a 16-lane group of
    // the b64 plane stores covers one row).  32-deep K-tiles (the wide bf16x6 tiles): that padding (20 dwords) lets two rows of a store group
    // overlap mod 32 banks - a third of the LDS cycles of the 128 x 128 bf16x6 kernel were bank conflicts (PMC, round 4) - so the rows stay
    // UNPADDED (64 B) and the four 16-byte chunks of row r are XOR-swizzled by (r >> 2) & 3: the 16 rows a b128 read group touches
    // ({0-3, 12-15, 20-27} and its shifts) land on 16 distinct 4-bank groups, a store group writes two whole rows = all 32 banks once.
    constexpr bool SWZ = BK3 == 32;
    constexpr int BROW3 = SWZ ? 64 : BK3 * 2 + 16;
    auto loff = [](int row, int kbyte) -> int {   // byte offset of (row, byte kbyte of the row's K-slice) inside a plane
        if constexpr (SWZ) return row * 64 + ((((kbyte >> 4) ^ (row >> 2)) & 3) << 4) + (kbyte & 15);
        else return row * BROW3 + kbyte;
    };
    constexpr int LPR = BK3 / 4;          // lanes per row slice (float4 each)
    constexpr int RPP = NT / LPR;         // rows per staging pass
    constexpr int A_LD4 = BM / RPP, W_LD4 = BN / RPP;            // float4 loads per thread and tile
    static_assert(A_LD4 >= 1 && W_LD4 >= 1, "tile too small for 256 threads");
    // pre-split W: 16-B chunks of 8 bf16; chunk c of a plane tile = (row c / CPR, k 8 * (c % CPR))
    constexpr int CPR = BK3 / 8, W_CH = BN * CPR / NT, A_CH = BM * CPR / NT;
    static_assert(!ASPLIT || ((BM * CPR) % NT == 0 && WSPLIT && !ANORM), "pre-split A: plane tile must divide over the threads");
    static_assert(!WSPLIT || (BN * CPR) % NT == 0, "plane tile must divide over the threads");
    constexpr int TLD = BN + 4;
    constexpr int PLANE_A = BM * BROW3, PLANE_W = BN * BROW3;    // bytes
    constexpr int BUF = NPL * (PLANE_A + PLANE_W);               // hi + lo (+ mid) of A and W
    // the epilogue re-uses the operand buffer as a row-major fp32 tile
    constexpr int EPI = (BM == 128 ? 64 : BM) * TLD * 4;   // BM = 128: two 64-row halves
    constexpr int LDS_BYTES = BUF > EPI ? BUF : EPI;
    static_assert((NT / (BN / 4)) * BN * 2 * 4 <= LDS_BYTES, "column-statistics scratch must fit");
    constexpr int AN_MAXC = 512;                                 // channels of a normalised A operand (scale + shift table behind the buffers)
    static_assert(!ANORM || (NT * 32 + AN_MAXC * 8 <= LDS_BYTES), "statistics fold scratch must fit in the operand buffer");
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES + (ANORM ? AN_MAXC * 8 : 0)];
    float *nsc = reinterpret_cast<float *>(lds_raw + LDS_BYTES), *nsh = nsc + AN_MAXC;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;   // position in the 2x2 wave grid
    const BlockId bid = gemm_block_id(g);
    const int m0 = bid.y * BM, n0 = bid.x * BN;
    const int kbeg = bid.z * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int ntiles = (kend - kbeg + BK3 - 1) / BK3;
    const int lrow = tid / LPR, lk = (tid % LPR) * 4;            // LPR lanes cover one row slice; RPP rows per pass
    struct Stage {                                               // one K-tile on its way from global memory to LDS
        f32x4 ra[ASPLIT ? 1 : A_LD4];                            // fp32 A: native vectors, plain 16-B loads, no struct copies
        f32x4 rah[ASPLIT ? A_CH : 1], ral[ASPLIT ? A_CH : 1];   // pre-split A: 16-B chunks (8 bf16) of the hi / lo plane
        f32x4 rw[WSPLIT ? 1 : W_LD4];                            // fp32 W: W_LD4 chunks of 4
        f32x4 rwh[WSPLIT ? W_CH : 1], rwl[WSPLIT ? W_CH : 1];   // pre-split W: 16-B chunks (8 bf16) of the hi / lo plane, as opaque bits
        f32x4 rwm[(WSPLIT && X6) ? W_CH : 1];                    // ... and of the mid plane (bf16x6: planes hi | mid | lo)
        unsigned amask;                                          // validity of the A values (see gload)
        bool afull;                                              // uniform: the A registers hold a full dense tile (no zeroing needed)
        int achan;                                               // ANORM: first of the 4 channels the staged float4s of this thread belong to
    };
    Stage st0;
    const bool conv = g.cv_ks != 0;

    // Addressing: a workgroup-uniform base (scalar registers, advanced by one K-tile per iteration) plus a per-thread 32-bit
    // byte offset that never changes -> the loads of a full tile need NO vector arithmetic at all.  Rows past M / N are
    // clamped to the last row: they only feed output rows / columns that are never stored (MFMA rows and columns are
    // independent), so they need no zeroing; only K tails (and convolution padding) are zeroed, on the A side.
    unsigned aoff[ASPLIT ? A_CH : A_LD4], woff[WSPLIT ? W_CH : W_LD4];
    if constexpr (ASPLIT) {
#pragma unroll
        for (int j = 0; j < A_CH; ++j) {
            const int c = tid + NT * j;
            aoff[j] = ((unsigned)(min(m0 + c / CPR, g.M - 1) - m0) * (unsigned)g.lda + 8 * (c % CPR)) * 2u;
        }
    } else {
#pragma unroll
        for (int j = 0; j < A_LD4; ++j) aoff[j] = ((unsigned)(min(m0 + lrow + RPP * j, g.M - 1) - m0) * (unsigned)g.lda + lk) * 4u;
    }
    if constexpr (WSPLIT) {
#pragma unroll
        for (int j = 0; j < W_CH; ++j) {
            const int c = tid + NT * j;
            woff[j] = ((unsigned)(min(n0 + c / CPR, g.N - 1) - n0) * (unsigned)g.ldw + 8 * (c % CPR)) * 2u;
        }
    } else {
#pragma unroll
        for (int j = 0; j < W_LD4; ++j) woff[j] = ((unsigned)(min(n0 + lrow + RPP * j, g.N - 1) - n0) * (unsigned)g.ldw + lk) * 4u;
    }
    const char *abase = ASPLIT ? reinterpret_cast<const char *>(reinterpret_cast<const uint16_t *>(g.A) + (size_t)m0 * g.lda + kbeg)
                               : reinterpret_cast<const char *>(g.A + (size_t)m0 * g.lda + kbeg);
    const size_t alo = (size_t)g.a_lo_off * 2;
    const char *wbase = WSPLIT ? reinterpret_cast<const char *>(reinterpret_cast<const uint16_t *>(g.W) + (size_t)n0 * g.ldw + kbeg)
                               : reinterpret_cast<const char *>(g.W + (size_t)n0 * g.ldw + kbeg);
    const size_t wlo = (size_t)g.w_lo_off * 2;

    // conv-mode pixel coordinates of this thread's A rows (row = m0 + lrow + RPP*j)
    int yo[A_LD4], xo[A_LD4], fb[A_LD4];
    if (!ASPLIT && conv) {
#pragma unroll
        for (int j = 0; j < A_LD4; ++j) {
            const int r = min(m0 + lrow + RPP * j, g.M - 1);
            const int f = r / g.cv_Pout, pix = r - f * g.cv_Pout;
            fb[j] = f * g.cv_H * g.cv_W;
            yo[j] = pix / g.cv_Wo;
            xo[j] = pix - yo[j] * g.cv_Wo;
        }
    }
    // Validity of the A values is kept as a bit mask and applied when the registers are consumed (sstore), so nothing touches
    // a loaded value - and no s_waitcnt is needed - until after the MFMAs.  Full dense tiles carry no mask at all.
    auto gload = [&](int t, Stage &st) {
        const bool full = kbeg + (t + 1) * BK3 <= kend;   // uniform
        const int k = kbeg + t * BK3 + lk;
        const bool kin = k < kend;
        if constexpr (ASPLIT) {
            const char *at = abase + (size_t)t * (BK3 * 2);
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            if (full) {
#pragma unroll
                for (int j = 0; j < A_CH; ++j) {
                    st.rah[j] = *reinterpret_cast<const f32x4 *>(at + aoff[j]);
                    st.ral[j] = *reinterpret_cast<const f32x4 *>(at + alo + aoff[j]);
                }
            } else {   // K tail: chunks past the K range are ZERO (the W side re-reads finite values there); K is a multiple of 8
#pragma unroll
                for (int j = 0; j < A_CH; ++j) {
                    const int c8 = 8 * ((tid + NT * j) % CPR);
                    const bool cin = kbeg + t * BK3 + c8 < kend;
                    const char *q = cin ? at + aoff[j] : abase + (aoff[j] - 2u * c8);
                    const f32x4 h = *reinterpret_cast<const f32x4 *>(q), l = *reinterpret_cast<const f32x4 *>(q + alo);
                    st.rah[j] = cin ? h : zero;
                    st.ral[j] = cin ? l : zero;
                }
            }
        } else if (!conv) {
            if constexpr (ANORM) st.achan = kin ? k : 0;
            st.afull = full;
            const char *at = abase + (size_t)t * (BK3 * 4);
            if (full) {
#pragma unroll
                for (int j = 0; j < A_LD4; ++j) st.ra[j] = *reinterpret_cast<const f32x4 *>(at + aoff[j]);
            } else {
                st.amask = kin ? ~0u : 0u;   // lanes past the K range re-read the first values of their row (always in range)
#pragma unroll
                for (int j = 0; j < A_LD4; ++j) st.ra[j] = *reinterpret_cast<const f32x4 *>(kin ? at + aoff[j] : abase + (aoff[j] - 4u * lk));
            }
        } else {
            st.afull = false;
            const int kc = kin ? k : 0;
            const int tap = kc / g.cv_Cin, c = kc - tap * g.cv_Cin;
            const int dy = tap / g.cv_ks, dx = tap - dy * g.cv_ks;
            if constexpr (ANORM) st.achan = c;
            st.amask = 0;
#pragma unroll
            for (int j = 0; j < A_LD4; ++j) {
                const int yi = yo[j] * g.cv_stride - g.cv_pad + dy, xi = xo[j] * g.cv_stride - g.cv_pad + dx;
                const bool ok = kin && (unsigned)yi < (unsigned)g.cv_H && (unsigned)xi < (unsigned)g.cv_W;
                const int yc = min(max(yi, 0), g.cv_H - 1), xc = min(max(xi, 0), g.cv_W - 1);
                st.ra[j] = *reinterpret_cast<const f32x4 *>(g.A + ((size_t)fb[j] + (size_t)yc * g.cv_W + xc) * g.lda + c);
                st.amask |= ok ? (1u << j) : 0u;
            }
        }
        // W: lanes past the K range re-read the start of their row (finite values x zeroed A = 0)
        if constexpr (WSPLIT) {
            const char *wt = wbase + (size_t)t * (BK3 * 2);
            if (full) {
#pragma unroll
                for (int j = 0; j < W_CH; ++j) {
                    st.rwh[j] = *reinterpret_cast<const f32x4 *>(wt + woff[j]);
                    st.rwl[j] = *reinterpret_cast<const f32x4 *>(wt + (NPL - 1) * wlo + woff[j]);
                    if constexpr (X6) st.rwm[j] = *reinterpret_cast<const f32x4 *>(wt + wlo + woff[j]);
                }
            } else {  // chunks that start past the K range re-read the first chunk of their row (plane rows are padded to 8 values)
#pragma unroll
                for (int j = 0; j < W_CH; ++j) {
                    const int c8 = 8 * ((tid + NT * j) % CPR);
                    const bool cin = kbeg + t * BK3 + c8 < kend;
                    const char *q = cin ? wt + woff[j] : wbase + (woff[j] - 2u * c8);
                    st.rwh[j] = *reinterpret_cast<const f32x4 *>(q);
                    st.rwl[j] = *reinterpret_cast<const f32x4 *>(q + (NPL - 1) * wlo);
                    if constexpr (X6) st.rwm[j] = *reinterpret_cast<const f32x4 *>(q + wlo);
                }
            }
        } else {
            const char *wt = wbase + (size_t)t * (BK3 * 4);
            if (full) {
#pragma unroll
                for (int j = 0; j < W_LD4; ++j) st.rw[j] = *reinterpret_cast<const f32x4 *>(wt + woff[j]);
            } else {
#pragma unroll
                for (int j = 0; j < W_LD4; ++j) st.rw[j] = *reinterpret_cast<const f32x4 *>(kin ? wt + woff[j] : wbase + (woff[j] - 4u * lk));
            }
        }
    };
    auto sstore = [&](Stage &st) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        if constexpr (ANORM) {
            // the pending normalisation of the producer: y * sc + sh, LeakyReLU as max(v, v * slope) (0 <= slope <= 1) - the
            // same operations in the same order as the stand-alone apply kernel.  Zero padding / K tails are masked afterwards.
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(nsc + st.achan), sh = *reinterpret_cast<const f32x4 *>(nsh + st.achan);
#pragma unroll
            for (int j = 0; j < A_LD4; ++j) {
                f32x4 v = st.ra[j] * sc + sh;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] >= 0.f ? v[e] : v[e] * g.an.slope;
                st.ra[j] = v;
            }
        }
        if constexpr (ASPLIT) {
#pragma unroll
            for (int j = 0; j < A_CH; ++j) {
                const int c = tid + NT * j;
                unsigned char *p = lds_raw + loff(c / CPR, (c % CPR) * 16);
                *reinterpret_cast<f32x4 *>(p) = st.rah[j];
                *reinterpret_cast<f32x4 *>(p + PLANE_A) = st.ral[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < A_LD4; ++j) {
                const f32x4 v = (st.afull || ((st.amask >> j) & 1u)) ? st.ra[j] : zero;
                unsigned char *p = lds_raw + loff(lrow + RPP * j, lk * 2);
                uint2 hi, lo;
                if constexpr (X6) {   // planes: hi | mid | lo
                    uint2 mid;
                    split4x3(v, hi, mid, lo);
                    *reinterpret_cast<uint2 *>(p + PLANE_A) = mid;
                    *reinterpret_cast<uint2 *>(p + 2 * PLANE_A) = lo;
                } else {
                    split4(v, hi, lo);
                    *reinterpret_cast<uint2 *>(p + PLANE_A) = lo;
                }
                *reinterpret_cast<uint2 *>(p) = hi;
            }
        }
        if constexpr (WSPLIT) {
#pragma unroll
            for (int j = 0; j < W_CH; ++j) {
                const int c = tid + NT * j;
                unsigned char *p = lds_raw + NPL * PLANE_A + loff(c / CPR, (c % CPR) * 16);
                *reinterpret_cast<f32x4 *>(p) = st.rwh[j];
                *reinterpret_cast<f32x4 *>(p + (NPL - 1) * PLANE_W) = st.rwl[j];
                if constexpr (X6) *reinterpret_cast<f32x4 *>(p + PLANE_W) = st.rwm[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < W_LD4; ++j) {
                uint2 hi, lo;
                unsigned char *p = lds_raw + NPL * PLANE_A + loff(lrow + RPP * j, lk * 2);
                if constexpr (X6) {
                    uint2 mid;
                    split4x3(st.rw[j], hi, mid, lo);
                    *reinterpret_cast<uint2 *>(p + PLANE_W) = mid;
                    *reinterpret_cast<uint2 *>(p + 2 * PLANE_W) = lo;
                } else {
                    split4(st.rw[j], hi, lo);
                    *reinterpret_cast<uint2 *>(p + PLANE_W) = lo;
                }
                *reinterpret_cast<uint2 *>(p) = hi;
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int li = lane & 31, lh = lane >> 5;
    union Frag { uint4 u; bf16x8 v; };
    constexpr int STEPS = BK3 / 16;   // 16-deep MFMA steps per tile
    // lane (li, lh) reads chunk 2 s2 + lh of its row; every row this lane touches is li mod 32, so the swizzle term is a lane constant
    const unsigned char *as = lds_raw + (wm * 32 * TM + li) * BROW3 + (SWZ ? 0 : lh * 16);
    const unsigned char *bs = lds_raw + NPL * PLANE_A + (wn * 32 * TN + li) * BROW3 + (SWZ ? 0 : lh * 16);
    const int swz = (li >> 2) & 3;
    auto koff = [&](int s2) -> int { return SWZ ? (((2 * s2 + lh) ^ swz) << 4) : s2 * 32; };
    auto compute = [&]() {
#pragma unroll
        for (int s2 = 0; s2 < STEPS; ++s2) {  // lane (i,h) owns k = 16*s2 + 8h .. +7
            Frag ah[TM], al[TM], bh[TN], bl[TN], am[X6 ? TM : 1], bm[X6 ? TN : 1];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i].u = *reinterpret_cast<const uint4 *>(as + i * 32 * BROW3 + koff(s2));
                al[i].u = *reinterpret_cast<const uint4 *>(as + (NPL - 1) * PLANE_A + i * 32 * BROW3 + koff(s2));
                if constexpr (X6) am[i].u = *reinterpret_cast<const uint4 *>(as + PLANE_A + i * 32 * BROW3 + koff(s2));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j].u = *reinterpret_cast<const uint4 *>(bs + j * 32 * BROW3 + koff(s2));
                bl[j].u = *reinterpret_cast<const uint4 *>(bs + (NPL - 1) * PLANE_W + j * 32 * BROW3 + koff(s2));
                if constexpr (X6) bm[j].u = *reinterpret_cast<const uint4 *>(bs + PLANE_W + j * 32 * BROW3 + koff(s2));
            }
            // product-major order: consecutive MFMAs go to DIFFERENT accumulator tiles (TM x TN independent chains), so none waits for
            // the result of the one issued right before it; per accumulator the order of the terms is unchanged (smallest first)
            auto term = [&](auto pa, auto pb) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa(i), pb(j), acc[i][j], 0, 0, 0);
            };
            auto AH = [&](int i) { return ah[i].v; };
            auto AL = [&](int i) { return al[i].v; };
            auto BH = [&](int j) { return bh[j].v; };
            auto BL = [&](int j) { return bl[j].v; };
            if constexpr (X6) {   // smallest terms first: 2^-16 (lo*hi, hi*lo, mid*mid), 2^-8 (mid*hi, hi*mid), 1 (hi*hi)
                auto AM = [&](int i) { return am[i].v; };
                auto BM_ = [&](int j) { return bm[j].v; };
                term(AL, BH);
                term(AH, BL);
                term(AM, BM_);
                term(AM, BH);
                term(AH, BM_);
                term(AH, BH);
            } else {
                term(AL, BH);
                term(AH, BL);
                term(AH, BH);
            }
        }
    };
    // One LDS buffer + a register-staged next tile: issue global loads (t+1) -> MFMAs on tile t -> barrier -> split + write
    // tile t+1 -> barrier.  (Measured on MI355X: deeper register prefetch (2-3 tiles in flight, exact vmcnt) and a second LDS
    // buffer do NOT shorten the ~1 us a lone workgroup spends per 64 KB tile - that is the CU's L2 fill rate; only spreading
    // the tiles over more CUs does, which is what split-K is tuned for.)
    if (ntiles > 0) gload(0, st0);
    if constexpr (ANORM) {
        // statistics of this tile's frame -> scale / shift table in LDS: copied from the finalized vectors (cofi_norm_finalize) or,
        // without them, folded here from the producer's partials (fold scratch lives in the still unused operand buffer)
        const int f = g.an_rows > 0 ? m0 / g.an_rows : 0;
        if (g.an.scsh) {
            const float *src = g.an.scsh + (size_t)f * 2 * g.an.C;
            for (int c = tid * 4; c < g.an.C; c += NT * 4) {
                *reinterpret_cast<f32x4 *>(nsc + c) = *reinterpret_cast<const f32x4 *>(src + c);
                *reinterpret_cast<f32x4 *>(nsh + c) = *reinterpret_cast<const f32x4 *>(src + g.an.C + c);
            }
        } else {
            double *dred = reinterpret_cast<double *>(lds_raw);
            float *sstat = reinterpret_cast<float *>(lds_raw + NT * 32);
            fold_stat_table<NT, 6>(g.an.part + (size_t)f * g.an.nslab * g.an.tcols * 2, g.an.nslab, g.an.tcols, g.an.groups, g.an.count, g.an.eps,
                                   dred, sstat);
            norm_scale_shift<NT>(g.an, sstat, nsc, nsh);
        }
        __syncthreads();
    }
    if (ntiles > 0) sstore(st0);
    __syncthreads();
    {
        for (int t = 0; t < ntiles; ++t) {
            if (t + 1 < ntiles) gload(t + 1, st0);
            compute();
            __syncthreads();              // every wave is done reading tile t
            if (t + 1 < ntiles) {
                sstore(st0);
                __syncthreads();          // tile t+1 visible
            }
        }
    }

    float *lds = reinterpret_cast<float *>(lds_raw);
    if (g.ksplit > 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * 32 * TN + j * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row < g.M && col < g.N) g.ws[((size_t)bid.z * g.M + row) * g.N + col] = acc[i][j][r];
                }
            }
        return;
    }
    if constexpr (BM == 128) {
        // epilogue in two 64-row halves (statistics slabs of 64 rows): half the LDS tile, so the operand buffer - not the epilogue -
        // sizes the workgroup's LDS and a third workgroup fits on the CU
        for (int half = 0; half < 2; ++half) {
            if (m0 + 64 * half >= g.M) break;   // uniform: a last tile with <= 64 valid rows has no second slab (its table entry does not exist)
            if (wm == half) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int rl = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                            lds[rl * TLD + wn * 32 * TN + j * 32 + li] = acc[i][j][r];
                        }
            }
            __syncthreads();
            rowwise_epilogue<64, BN, false>(g, lds, TLD, m0 + 64 * half, n0, 2 * bid.y + half, lds);
            __syncthreads();
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                lds[rl * TLD + wn * 32 * TN + j * 32 + li] = acc[i][j][r];
            }
    __syncthreads();
    rowwise_epilogue<BM, BN, false>(g, lds, TLD, m0, n0, bid.y, lds);
}

#include "gemm_planes.inc"
#include "gemm_x6_big.inc"
#include "gemm_f16_big.inc"
#include "conv_direct.inc"

// split-K tail, same row-wise epilogue reading the partial sums.  Two tilings:
//   64 rows x 32 columns  (plain / column statistics): slabs of 64 rows keep the statistics table small and
//                          even M = 1280, N = 256 still gives 160 workgroups;
//   16 rows x 128 columns (fused LayerNorm): a tile spans whole rows (N <= 128).
constexpr int SK_ROWS = 64, SK_COLS = 32, SKLN_ROWS = 16;
template <int COLS>   // 32; 64 when a statistics table entry spans 64 columns
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(GemmArgs g) {
    __shared__ float red[(256 / (COLS / 4)) * COLS * 2];
    rowwise_epilogue<SK_ROWS, COLS, true>(g, nullptr, 0, blockIdx.y * SK_ROWS, blockIdx.x * COLS, blockIdx.y, red);
}
__global__ __launch_bounds__(256) void splitk_epilogue_ln_kernel(GemmArgs g) {
    rowwise_epilogue<SKLN_ROWS, 128, true>(g, nullptr, 0, blockIdx.y * SKLN_ROWS, blockIdx.x * 128, blockIdx.y, nullptr);
}

struct Plan {
    int bm, bn, ksplit, kchunk;
    int pcfg;   // >= 0: gemm_planes_kernel configuration (kPlanesCfg) - both operands are bf16 planes; -1: the register-staged kernels
    int big;    // 1: gemm_x6_big_kernel (bf16x6, 256 x 128 tile, one workgroup per CU)
};

// Configurations of gemm_planes_kernel: tile, K-tile depth, wave grid, LDS stages, workgroups per CU the register budget allows.
struct PlanesCfg { int bm, bn, bk, wm, wn, nst, minw; };
#define COFI_PLANES_CFGS(X)          \
    X(0, 128, 128, 64, 4, 2, 2, 2)   \
    X(1, 64, 64, 64, 2, 2, 2, 2)     \
    X(2, 256, 128, 32, 4, 2, 3, 2)   \
    X(3, 256, 256, 32, 4, 2, 2, 2)   \
    X(4, 128, 32, 64, 4, 1, 2, 2)    \
    X(5, 128, 64, 64, 4, 2, 2, 2)    \
    X(6, 64, 64, 32, 2, 2, 4, 2)     \
    X(7, 128, 128, 32, 4, 2, 4, 2)   \
    X(8, 64, 128, 64, 2, 2, 2, 1)
static const PlanesCfg kPlanesCfg[] = {
#define X(id, bm, bn, bk, wm, wn, nst, minw) {bm, bn, bk, wm, wn, nst, minw},
    COFI_PLANES_CFGS(X)
#undef X
};
constexpr int kNumPlanesCfg = sizeof(kPlanesCfg) / sizeof(kPlanesCfg[0]);

// Tuned plans for the shapes of the KITTI / nuScenes-shaped forward (tools/tune_gemm.py on MI355X, bf16x3 kernel):
// {M, N, K, bm, bn, ksplit}.  Anything not listed falls through to the heuristic below.
struct TunedPlan { int M, N, K, bm, bn, ks; };
#include "gemm_plans.inc"
// ... and for the bf16x6 kernel (twice the MFMAs and 1.5x the LDS traffic per K-tile move the best split): tools/tune_gemm.py --gemm bf16x6
#include "gemm_plans_x6.inc"

// Tuning / test hooks (include/cofi_hip_tune.h): plan overrides of the CALLING THREAD only - plans are chosen on the thread that enqueues a
// launch, so a tool or a test that forces a plan cannot change the launches of any other thread; the product path never sets them.
thread_local int g_force_bm = 0, g_force_bn = 0, g_force_ks = 0;

Plan finish_plan(int K, int bm, int bn, int ks) {
    Plan p;
    p.pcfg = -1;
    p.big = 0;
    p.bm = bm; p.bn = bn;
    int ktiles = cofi_cdiv(K, BK);
    if (ks < 1) ks = 1;
    int tiles_per = cofi_cdiv(ktiles, ks);
    tiles_per = (tiles_per + 3) & ~3;  // k-chunks in multiples of 128: the bf16x3 kernel steps K by 128
    p.kchunk = tiles_per * BK;
    p.ksplit = cofi_cdiv(K, p.kchunk);
    return p;
}

// Heuristic: largest tile that still gives >= ~1 workgroup per CU, then split K until the chip
// (256 CUs) is covered about twice, keeping >= 2 k-tiles (64 values) per split.
Plan make_plan_base(int M, int N, int K, bool fused_ln, int arith) {
    Plan p;
    p.pcfg = -1;
    p.big = 0;
    if (g_force_bm) {
        p = finish_plan(K, g_force_bm, (g_force_bm == 128 && g_force_bn == 64 && arith != 2) ? 128 : g_force_bn, g_force_ks);   // 128 x 64: bf16x6 only
        if (fused_ln && p.ksplit == 1 && p.bn < N) { p.bm = 64; p.bn = 128; }
        return p;
    }
    static const bool x6_table = !(getenv("COFI_GEMM_X6_PLANS") && atoi(getenv("COFI_GEMM_X6_PLANS")) == 0);   // A/B switch (tools)
    if (arith == 2 && x6_table)
        for (const TunedPlan &t : kTunedPlansX6)
            if (t.M == M && t.N == N && t.K == K) {
                p = finish_plan(K, t.bm, t.bn, t.ks);
                if (fused_ln && p.ksplit == 1 && p.bn < N) { p.bm = 64; p.bn = 128; }
                return p;
            }
    for (const TunedPlan &t : kTunedPlans)
        if (t.M == M && t.N == N && t.K == K) {
            p = finish_plan(K, t.bm, t.bn, t.ks);
            if (fused_ln && p.ksplit == 1 && p.bn < N) { p.bm = 64; p.bn = 128; }
            return p;
        }
    auto blocks = [&](int bm, int bn) { return (long)cofi_cdiv(M, bm) * cofi_cdiv(N, bn); };
    if (N > 64 && M > 64 && blocks(128, 128) >= 200) {
        p.bm = 128; p.bn = 128;
    } else if (N > 64 && blocks(64, 128) >= 200) {
        p.bm = 64; p.bn = 128;
    } else if (N > 32) {
        p.bm = 64; p.bn = 64;
        if (N > 64 && blocks(64, 128) * 2 >= blocks(64, 64) && blocks(64, 64) > 1024) { p.bn = 128; }
    } else {
        p.bm = 64; p.bn = 64;
    }
    long nb = blocks(p.bm, p.bn);
    int ktiles = cofi_cdiv(K, BK);
    int ks = 1;
    // Split K only when the serial k-loop is long: a split costs a second kernel (the reduction epilogue,
    // ~5 us of launch + latency), which a loop of <= 16 k-tiles (K <= 512) cannot win back.
    if (nb < 384 && ktiles > 16) {
        ks = (int)((512 + nb - 1) / nb);
        int maxks = ktiles / 8;  // >= 256 k-values per split
        if (maxks < 1) maxks = 1;
        if (ks > maxks) ks = maxks;
        if (ks > 32) ks = 32;
    }
    int tiles_per = cofi_cdiv(ktiles, ks);
    tiles_per = (tiles_per + 3) & ~3;  // k-chunks in multiples of 128: the bf16x3 kernel steps K by 128
    p.kchunk = tiles_per * BK;
    p.ksplit = cofi_cdiv(K, p.kchunk);
    if (fused_ln && p.ksplit == 1 && p.bn < N) {  // un-split: one tile must span the whole row (N <= 128)
        p.bm = 64;
        p.bn = 128;
    }
    return p;
}

// bf16x6, <= 64 output columns over many rows (stack-mode batches of >= 4 frames): a 128 x 64 tile (waves of 64 x 32) splits a W tile once
// per 128 rows instead of once per 64 and reads 3/4 of the fragment bytes per MFMA.  Measured on the batch-16 shapes (tools/tall_tile_probe.py,
// outputs bit-equal to the 64 x 64 tile's): -12 ... -31 % per launch from 640 tiles up (M >= 81920), +-3 % at 320 tiles, +35 % at 160 - hence
// the threshold.  COFI_GEMM_TALL_TILE=0 switches it off (A/B), any other value is the threshold in tiles.
Plan make_plan(int M, int N, int K, bool fused_ln, int arith = 1) {   // arith: GemmArgs::bf16x3 (2 = bf16x6: its own table first)
    Plan p = make_plan_base(M, N, K, fused_ln, arith);
    static const long tall = getenv("COFI_GEMM_TALL_TILE") ? atol(getenv("COFI_GEMM_TALL_TILE")) : 512;
    if (arith == 2 && !g_force_bm && tall > 0 && p.pcfg < 0 && p.bm == 64 && p.bn == 64 && p.ksplit == 1 && N <= 64 &&
        (long)cofi_cdiv(M, 128) * cofi_cdiv(N, 64) >= tall)
        p.bm = 128;
    return p;
}

// ---- gemm_x6_big_kernel (256 x 128 tiles, ONE workgroup per CU): which shapes take it, and with which K split.
// One workgroup per CU means whole rounds of 256 workgroups: a grid of 320 tiles takes two rounds like one of 512.  The split is chosen
// to minimise  rounds x (k-tiles per workgroup + fixed cost)  plus the partial-sum traffic a split adds (ks x M x N x 4 bytes written
// and read again), in units of one K-tile of the main loop (~1.5 us).
// tuning hook (tools only): g_force_big = 1 forces the kernel on every eligible launch (split g_force_big_ks, 0 = chosen here), -1 disables it
thread_local int g_force_big = 0, g_force_big_ks = 0, g_big_dbg = 0;
thread_local long g_f16_launches = 0;     // launches of the calling thread that took gemm_f16_big_kernel, and their 2 M N K (cofi_tune_f16x3_launch_flops)
thread_local double g_f16_flops = 0.0;
thread_local int g_force_direct = 0;   // tools / tests: 1 = the direct 3 x 3 kernel on every eligible convolution (no tile-count threshold), 2 = ... with 4-row tiles, -1 = never, 0 = default
struct TunedBig { int M, N, K, ks; };   // ks = 0: keep the small-tile kernel for this shape
#include "gemm_plans_big.inc"

bool big_plan(int M, int N, int K, Plan &p, bool f16 = false) {   // f16: the launch will run gemm_f16_big_kernel (COFI_GEMM_F16X3)
    static const int mode = getenv("COFI_GEMM_BIG") ? atoi(getenv("COFI_GEMM_BIG")) : 1;   // A/B switch: 0 = never
    if (g_force_big < 0 || (mode == 0 && g_force_big == 0) || g_force_bm) return false;
    // N <= 64: the 256 x 64 instantiation of the kernel (four waves of 64 x 64) exists in the template and is bit-equal, but LOSES to the
    // 128 x 64 small tiles (round 5 probe, profiles/r05/big_gemm_probe_n64.txt: 327680 x 64 x 576 222 vs 209 us, 81920 x 64 x 576 80 vs 64 us -
    // the split of the A operand is 4.6 VALU instructions per MFMA there, and one workgroup per CU has nobody to overlap them with): not built.
    if ((K % 32) || N < 128 || M < 256) return false;
    const int bn = 128;
    const long tiles = (long)cofi_cdiv(M, 256) * cofi_cdiv(N, bn);
    const int ktiles = K / 32;
    int ks = 0;
    if (g_force_big > 0 && g_force_big_ks > 0) ks = g_force_big_ks;
    if (!ks && g_force_big == 0) {
        bool listed = false;
        static const int f16_table = getenv("COFI_GEMM_F16_TABLE") ? atoi(getenv("COFI_GEMM_F16_TABLE")) : 1;   // A/B switch: 0 = the six-product kernel's table and thresholds
        if (!f16_table) f16 = false;
        if (f16)
            for (const TunedBig &t : kTunedBigF16)
                if (t.M == M && t.N == N && t.K == K) { ks = t.ks; listed = true; break; }
        if (!listed)
            for (const TunedBig &t : kTunedBig)
                if (t.M == M && t.N == N && t.K == K) { ks = t.ks; listed = true; break; }
        if (listed && ks == 0) return false;
        if (!listed && (tiles < 160 || K < (f16 ? 256 : 512))) return false;   // short loops / small grids: two or three small workgroups per CU overlap their prologues and epilogues
        if (!listed && K < 512) ks = 1;
    }
    // The f16x3 kernel never splits K from this many 128 x 128 tiles up (COFI_GEMM_F16_KS1_TILES, 0 = off: the table's / cost model's splits
    // everywhere).  A split pays for a launch that runs ALONE on the chip - it fills idle CUs - and that is how the table was measured; beside
    // the other submissions of a pipeline its partial sums are only extra traffic and a fold launch.  Same box, frames/s with the splits /
    // capped at 80 / never split: batch 16 827 / 837 / 842 (another box 825 at 80, 824 never), batch 1 511 / 515 / 506, stress (one frame,
    // nothing else in flight) 59.5 / - / 58.4 with 59.4 at 160: small grids still need their splits (profiles/r06/ab_ks1*.txt).
    static const long ks1_tiles = getenv("COFI_GEMM_F16_KS1_TILES") ? atol(getenv("COFI_GEMM_F16_KS1_TILES")) : 80;
    if (f16 && ks1_tiles > 0 && g_force_big == 0 && (long)cofi_cdiv(M, 128) * cofi_cdiv(N, 128) >= ks1_tiles) ks = 1;
    if (!ks) {
        double best = 1e30;
        for (int c = 1; c <= 8; ++c) {
            const int chunk = cofi_cdiv(cofi_cdiv(ktiles, c), 4) * 4;   // k-chunks in multiples of 128, as the other kernels
            const int eff = cofi_cdiv(ktiles, chunk);
            if (eff != c || chunk < 8) continue;
            // f16x3 (pre-split W): 128 x 128 tiles, two workgroups per CU; else one 256 x 128 workgroup per CU
            const double rounds = f16 ? (double)cofi_cdiv((long)cofi_cdiv(M, 128) * cofi_cdiv(N, bn) * c, 512) : (double)cofi_cdiv(tiles * c, 256);
            const double partial = c > 1 ? 2.0 * c * (double)M * N * 4.0 / 4.0e12 / 1.5e-6 : 0.0;   // in K-tile units
            const double cost = rounds * (chunk + 6.0) + partial + (c > 1 ? 4.0 : 0.0);
            if (cost < best) { best = cost; ks = c; }
        }
        if (!ks) return false;
    }
    p.pcfg = -1;
    p.big = 1;
    p.bm = 256; p.bn = bn;
    const int chunk = cofi_cdiv(cofi_cdiv(ktiles, ks), 4) * 4;
    p.kchunk = chunk * 32;
    p.ksplit = cofi_cdiv(K, p.kchunk);
    return true;
}

// f16x3 kernel with pre-split W: 128 x 128 tiles, two workgroups per CU, instead of one 256 x 128 workgroup (gemm_f16_big.inc).
// Measured on the large contractions of a batch-16 forward (tools/f16_probe.py, profiles/r06/f16_probe_bm256.txt / _bm128.txt, same K splits):
// 8.81 -> 8.28 ms per submission, 27 of 29 shapes faster.  COFI_GEMM_F16_BM256=1 / cofi_tune_big_debug bit 512 (A/B): the 256-row form.
bool f16_two_per_cu(int M, int N, int K, const Plan &p) {
    (void)M; (void)N; (void)K; (void)p;
    static const int bm256 = getenv("COFI_GEMM_F16_BM256") ? atoi(getenv("COFI_GEMM_F16_BM256")) : 0;
    return !bm256 && (g_big_dbg & 512) == 0;
}

// bytes of the f16x3 kernels' tile-flag table: one word per workgroup of a 256 x 128 launch with `ks` K-slices
size_t f16_flag_bytes(int M, int N, int ks, int bm = 128) { return (size_t)cofi_cdiv(M, bm) * cofi_cdiv(N, 128) * ks * sizeof(unsigned); }   // (sized for the 128-row tiles: covers both forms)

// ---- plans of gemm_planes_kernel (both operands pre-split): {M, N, K, configuration, split-K}, tuned on MI355X by
// tools/tune_gemm.py --planes; anything not listed falls through to the heuristic.
struct TunedPlanes { int M, N, K, cfg, ks; };
#include "gemm_planes_plans.inc"

// tuning hook (tools/tune_gemm.py only): cfg >= 0 forces that configuration and split, -2 sends pre-split operands to the
// register-staged kernel instead (the A/B partner of the bit-identity test), -1 restores table + heuristic
thread_local int g_force_pcfg = -1, g_force_pks = 0;

Plan finish_planes_plan(int K, int cfg, int ks) {
    Plan p;
    p.pcfg = cfg;
    p.big = 0;
    p.bm = kPlanesCfg[cfg].bm;
    p.bn = kPlanesCfg[cfg].bn;
    const int ktiles = cofi_cdiv(K, 128);   // K chunks in multiples of 128, as finish_plan: equal splits give equal bits in both kernels
    if (ks < 1) ks = 1;
    p.kchunk = cofi_cdiv(ktiles, ks) * 128;
    p.ksplit = cofi_cdiv(K, p.kchunk);
    return p;
}

Plan make_planes_plan(int M, int N, int K) {
    if (g_force_pcfg >= 0 && g_force_pcfg < kNumPlanesCfg) return finish_planes_plan(K, g_force_pcfg, g_force_pks);
    for (const TunedPlanes &t : kTunedPlanes)
        if (t.M == M && t.N == N && t.K == K && t.cfg >= 0 && t.cfg < kNumPlanesCfg) return finish_planes_plan(K, t.cfg, t.ks);
    auto nblk = [&](int c) { return (long)cofi_cdiv(M, kPlanesCfg[c].bm) * cofi_cdiv(N, kPlanesCfg[c].bn); };
    const int max_ks = K >= 1024 ? K / 512 : 1;   // >= 512 K values (4 x 128) per split
    int cfg;
    if (N <= 32) cfg = 4;                          // 128 x 32
    else if (N <= 64) cfg = nblk(5) >= 400 ? 5 : 1;
    else if (nblk(3) * max_ks >= 400 && N % 256 == 0 && M >= 8192) cfg = 3;   // 256 x 256: the L2 -> LDS traffic of a tile halves again
    else if (nblk(2) >= 512) cfg = 2;              // 256 x 128
    else if (nblk(0) * max_ks >= 200) cfg = 0;     // 128 x 128
    else cfg = 1;                                  // 64 x 64
    const long nb = nblk(cfg);
    int ks = 1;
    if (nb < 200) {
        ks = (int)((256 + nb - 1) / nb);
        if (ks > max_ks) ks = max_ks;
        if (ks > 32) ks = 32;
    }
    return finish_planes_plan(K, cfg, ks);
}

// Tile order (GemmArgs::xcd): estimate the bytes both orders pull over the fabric — every XCD that touches a row panel of A
// or a column panel of W reads it through its own L2 — and take the cheaper one.
int xcd_order(const GemmArgs &g, const dim3 &grid) {
    const int mode = 2;   // 0 / 1 would force hardware / XCD-contiguous order; 2 picks per launch from the traffic estimate below
    const long gx = grid.x, gy = grid.y, gz = grid.z, nblk = gx * gy * gz;
    if (mode == 0 || (gx & (gx - 1)) || nblk < 16 || gy * gz >= 65536) return 0;
    const int enc = 1 + __builtin_ctzl(gx);
    if (mode == 1) return enc;
    auto gcd = [](long a, long b) { while (b) { long t = a % b; a = b; b = t; } return a; };
    const double abytes = (double)g.M * g.K, wbytes = (double)g.N * g.K;
    // hardware order: workgroup L -> XCD L % 8, L = (z*gy + y)*gx + x
    const double rr = abytes * std::min(gx, 8L) + wbytes * std::min(gy, 8 / gcd(gx % 8 ? gx % 8 : 8, 8));
    // contiguous order: XCD c owns workgroups [c*nblk/8, (c+1)*nblk/8) -> a K-slice of W is shared by ceil(8/gz) XCDs
    const double a_rep = std::min(8.0, std::max(1.0, 8.0 * gx / nblk));
    const double ct = abytes * a_rep + wbytes * std::min(std::min(gy, 8L), (8 + gz - 1) / gz);
    return ct < rr ? enc : 0;
}

int launch(const GemmArgs &g0, const Plan &p, hipStream_t s) {
    GemmArgs g = g0;
    g.ksplit = p.ksplit;
    g.kchunk = p.kchunk;
    dim3 grid(cofi_cdiv(g.N, p.bn), cofi_cdiv(g.M, p.bm), p.ksplit);
    g.xcd = xcd_order(g, grid);
    g.dbg = g_big_dbg;
    g.xcd_rcp_gy = grid.y > 1 ? (unsigned)((0x100000000ULL + grid.y - 1) / grid.y) : 0u;   // 0: gy = 1
    if (p.pcfg >= 0) {
        switch (p.pcfg) {
#define X(id, bm_, bn_, bk_, wm_, wn_, nst_, minw_)                                                                                      \
    case id:                                                                                                                             \
        hipLaunchKernelGGL((gemm_planes_kernel<bm_, bn_, bk_, wm_, wn_, nst_, minw_>), grid, dim3(64 * wm_ * wn_), 0, s, g);              \
        break;
            COFI_PLANES_CFGS(X)
#undef X
        default: return COFI_EINVAL;
        }
    } else if (p.big) {
        const bool cv = g.cv_ks != 0;
#define COFI_LAUNCH_BIG(BN_)                                                                                                  \
    do {                                                                                                                      \
        if (g.an.part && cv) hipLaunchKernelGGL((gemm_x6_big_kernel<true, true, BN_>), grid, dim3(256), 0, s, g);            \
        else if (g.an.part) hipLaunchKernelGGL((gemm_x6_big_kernel<true, false, BN_>), grid, dim3(256), 0, s, g);            \
        else if (cv) hipLaunchKernelGGL((gemm_x6_big_kernel<false, true, BN_>), grid, dim3(256), 0, s, g);                   \
        else hipLaunchKernelGGL((gemm_x6_big_kernel<false, false, BN_>), grid, dim3(256), 0, s, g);                          \
    } while (0)
        if (g.f16) {
            ++g_f16_launches;
            g_f16_flops += 2.0 * g.M * g.N * g.K;
            // the pipelined kernel, then the repair launch over the same grid (its workgroups exit at once unless the first one flagged their tile)
#define COFI_LAUNCH_F16B(ROBUST_, NW_, WPRE_, BM_)                                                                                                  \
    do {                                                                                                                                            \
        if (g.an.part && cv) hipLaunchKernelGGL((gemm_f16_big_kernel<true, true, ROBUST_, NW_, WPRE_, BM_>), grid, dim3(64 * NW_), 0, s, g);       \
        else if (g.an.part) hipLaunchKernelGGL((gemm_f16_big_kernel<true, false, ROBUST_, NW_, WPRE_, BM_>), grid, dim3(64 * NW_), 0, s, g);       \
        else if (cv) hipLaunchKernelGGL((gemm_f16_big_kernel<false, true, ROBUST_, NW_, WPRE_, BM_>), grid, dim3(64 * NW_), 0, s, g);              \
        else hipLaunchKernelGGL((gemm_f16_big_kernel<false, false, ROBUST_, NW_, WPRE_, BM_>), grid, dim3(64 * NW_), 0, s, g);                     \
    } while (0)
#define COFI_LAUNCH_F16(ROBUST_, NW_, WPRE_) COFI_LAUNCH_F16B(ROBUST_, NW_, WPRE_, 256)
            if (g.wscale && p.bm == 128) {   // pre-split W, 128 x 128 tiles: two workgroups per CU
                COFI_LAUNCH_F16B(false, 4, true, 128);
                COFI_LAUNCH_F16B(true, 4, true, 128);
            } else if (g.wscale) {      // pre-split W (eight-wave geometry only)
                COFI_LAUNCH_F16(false, 8, true);
                COFI_LAUNCH_F16(true, 8, true);
            } else if (g.dbg & 256) {   // cofi_tune_big_debug bit 256 (A/B): the four-wave geometry
                COFI_LAUNCH_F16(false, 4, false);
                COFI_LAUNCH_F16(true, 4, false);
            } else {
                COFI_LAUNCH_F16(false, 8, false);
                COFI_LAUNCH_F16(true, 8, false);
            }
#undef COFI_LAUNCH_F16
#undef COFI_LAUNCH_F16B
        } else {
            COFI_LAUNCH_BIG(128);
        }
#undef COFI_LAUNCH_BIG
    } else if (g.bf16x3 == 2) {
        // bf16x6: three planes per operand; K-tiles of 64 (64 x 64 tile: 54 KB of LDS) / 32 (wider tiles: 41 / 60 KB)
#define COFI_LAUNCH_BF16X6(BM_, BN_, TM_, TN_, BK_)                                                                        \
    do {                                                                                                                  \
        if (g.wsplit && g.an.part)                                                                                        \
            hipLaunchKernelGGL((gemm_bf16x3_kernel<BM_, BN_, TM_, TN_, BK_, true, 1, true, false, true>), grid, dim3(256), 0, s, g);    \
        else if (g.wsplit)                                                                                                \
            hipLaunchKernelGGL((gemm_bf16x3_kernel<BM_, BN_, TM_, TN_, BK_, true, 1, false, false, true>), grid, dim3(256), 0, s, g);   \
        else if (g.an.part)                                                                                               \
            hipLaunchKernelGGL((gemm_bf16x3_kernel<BM_, BN_, TM_, TN_, BK_, false, 1, true, false, true>), grid, dim3(256), 0, s, g);   \
        else                                                                                                              \
            hipLaunchKernelGGL((gemm_bf16x3_kernel<BM_, BN_, TM_, TN_, BK_, false, 1, false, false, true>), grid, dim3(256), 0, s, g);  \
    } while (0)
        if (p.bm == 128 && p.bn == 128)
            COFI_LAUNCH_BF16X6(128, 128, 2, 2, 32);
        else if (p.bm == 64 && p.bn == 128)
            COFI_LAUNCH_BF16X6(64, 128, 1, 2, 32);
        else if (p.bm == 128 && p.bn == 64)
            COFI_LAUNCH_BF16X6(128, 64, 2, 1, 32);
        else
            COFI_LAUNCH_BF16X6(64, 64, 1, 1, 64);
#undef COFI_LAUNCH_BF16X6
    } else if (g.bf16x3) {
#define COFI_LAUNCH_BF16X3(BM_, BN_, TM_, TN_, BK_)                                                                        \
    do {                                                                                                                  \
        if (g.an.part)                                                                                                    \
            hipLaunchKernelGGL((gemm_bf16x3_kernel<BM_, BN_, TM_, TN_, BK_, true, 1, true, false>), grid, dim3(256), 0, s, g);    \
        else if (g.asplit)                                                                                                \
            hipLaunchKernelGGL((gemm_bf16x3_kernel<BM_, BN_, TM_, TN_, BK_, true, 1, false, true>), grid, dim3(256), 0, s, g);    \
        else if (g.wsplit)                                                                                                \
            hipLaunchKernelGGL((gemm_bf16x3_kernel<BM_, BN_, TM_, TN_, BK_, true, 1, false, false>), grid, dim3(256), 0, s, g);   \
        else                                                                                                              \
            hipLaunchKernelGGL((gemm_bf16x3_kernel<BM_, BN_, TM_, TN_, BK_, false, 1, false, false>), grid, dim3(256), 0, s, g);  \
    } while (0)
        // K-tile depth 64 for every tile shape (measured: the 64x128 and 64x64 tiles run 1.4x / 1.05x faster with 64-deep K-tiles -
        // 2-3 workgroups per CU instead of 1-2 - than with 128-deep ones; the 128x128 tile is register-bound at 2 waves per SIMD either way)
        if (p.bm == 128 && p.bn == 128)
            COFI_LAUNCH_BF16X3(128, 128, 2, 2, 64);
        else if (p.bm == 64 && p.bn == 128)
            COFI_LAUNCH_BF16X3(64, 128, 1, 2, 64);
        else
            COFI_LAUNCH_BF16X3(64, 64, 1, 1, 64);
#undef COFI_LAUNCH_BF16X3
    } else if (p.bm == 128 && p.bn == 128)
        hipLaunchKernelGGL((gemm_kernel<128, 128, 2, 2>), grid, dim3(256), 0, s, g);
    else if (p.bm == 64 && p.bn == 128)
        hipLaunchKernelGGL((gemm_kernel<64, 128, 1, 2>), grid, dim3(256), 0, s, g);
    else
        hipLaunchKernelGGL((gemm_kernel<64, 64, 1, 1>), grid, dim3(256), 0, s, g);
    if (p.ksplit > 1) {
        if (g.ln_gamma || g.l2n)
            hipLaunchKernelGGL(splitk_epilogue_ln_kernel, dim3(1, cofi_cdiv(g.M, SKLN_ROWS)), dim3(256), 0, s, g);
        else if (g.colpart && g.stat_shift > 5)
            hipLaunchKernelGGL(splitk_epilogue_kernel<64>, dim3(cofi_cdiv(g.N, 64), cofi_cdiv(g.M, SK_ROWS)), dim3(256), 0, s, g);
        else
            hipLaunchKernelGGL(splitk_epilogue_kernel<SK_COLS>, dim3(cofi_cdiv(g.N, SK_COLS), cofi_cdiv(g.M, SK_ROWS)), dim3(256), 0, s, g);
    }
    return cofi_launch_status();
}

int check_common(const float *A, int lda, const float *W, int ldw, float *C, int ldc, int M, int N, int K) {
    if (!A || !W || !C || M < 0 || N <= 0 || K <= 0) return COFI_EINVAL;
    if ((K & 3) || (lda & 3) || (ldw & 3) || lda < K || ldw < K || ldc < N) return COFI_EINVAL;
    if (((uintptr_t)A & 15) || ((uintptr_t)W & 15)) return COFI_EINVAL;
    return 0;
}

// log2 of the statistics table width, or -1 if it cannot be used with N columns
int stat_shift_of(int width, int N) {
    if (width <= 0 || width > 64 || (width & (width - 1)) || (N % width)) return -1;
    int sh = 0;
    while ((1 << sh) < width) ++sh;
    return sh;
}

// the pending normalisation of the A operand; channels = columns of a dense A / input channels of a convolution.
// Runs on the bf16x3 kernels with pre-split weights and on the bf16x6 kernels.
int set_a_norm(GemmArgs &g, const cofi_norm_desc_t *a_norm, int channels, int a_rows_per_frame, int frames, const Plan &p) {
    if (!a_norm) return 0;
    if (!(g.bf16x3 == 1 && g.wsplit) && g.bf16x3 != 2) return COFI_EUNSUPPORTED;   // 3-term kernel: pre-split weights; 6-term kernel: fp32 or pre-split weights
    if (a_norm->channels != channels) return COFI_EINVAL;
    if (int rc = make_norm_src(a_norm, a_rows_per_frame, frames, 512, &g.an)) return rc;
    if (!(g.an.slope >= 0.f && g.an.slope <= 1.f)) return COFI_EINVAL;
    // a tile of GEMM rows must lie inside one frame
    g.an_rows = g.M / frames;
    if (frames > 1 && (g.M % frames || (g.an_rows % p.bm))) return COFI_EUNSUPPORTED;
    return 0;
}

int gemm_entry(const float *A, int lda, const cofi_norm_desc_t *a_norm, const float *W, int ldw, float *C, int ldc, int M, int N, int K,
               const float *bias, const float *rowdiv, int act, float *colpart, int stat_width, void *ws, size_t ws_bytes, int frames,
               cofi_stream_t stream) {
    if (int rc = check_common(A, lda, W, ldw, C, ldc, M, N, K)) return rc;
    if (M == 0) return 0;
    const int bf16x3 = (act & COFI_GEMM_BF16X6) ? 2 : ((act & COFI_GEMM_BF16X3) ? 1 : 0);
    const int wsplit = (act & COFI_GEMM_W_SPLIT) ? 1 : 0;
    const int asplit = (act & COFI_GEMM_A_SPLIT) ? 1 : 0;
    const int l2n = (act & COFI_GEMM_L2NORM) ? 1 : 0;
    const int f16 = (act & COFI_GEMM_F16X3) ? 1 : 0;
    const int wf16 = (act & COFI_GEMM_W_F16PRE) ? 1 : 0;
    act &= ~(COFI_GEMM_BF16X3 | COFI_GEMM_BF16X6 | COFI_GEMM_W_SPLIT | COFI_GEMM_A_SPLIT | COFI_GEMM_L2NORM | COFI_GEMM_F16X3 | COFI_GEMM_W_F16PRE);
    if (wf16 && (!f16 || wsplit)) return COFI_EINVAL;
    if (l2n && (N > 128 || asplit)) return COFI_EUNSUPPORTED;
    if (act < 0 || act > 3 || (wsplit && (bf16x3 == 0 || (ldw & 7))) || frames <= 0) return COFI_EINVAL;
    if (asplit && bf16x3 != 1) return COFI_EINVAL;
    if (asplit && (!wsplit || a_norm || (lda & 7) || (K & 7))) return COFI_EINVAL;
    const int sshift = colpart ? stat_shift_of(stat_width, N) : 0;
    if (sshift < 0) return COFI_EINVAL;
    Plan p = (asplit && g_force_pcfg != -2) ? make_planes_plan(M, N, K) : make_plan(M, N, K, l2n != 0, bf16x3);
    if (bf16x3 == 2 && !wsplit && !asplit && !l2n && !(a_norm && frames > 1 && (M / frames) % 256)) big_plan(M, N, K, p, f16 != 0);   // the 256 x 128 kernel for the large shapes
    if (a_norm && frames > 1 && p.pcfg < 0 && p.bm == 128 && p.bn == 64 && (M / frames) % 128) p.bm = 64;   // a normalising tile stays inside one frame
    if (p.ksplit > 1 && (!ws || ws_bytes < (size_t)p.ksplit * M * N * sizeof(float))) return COFI_EWORKSPACE;
    GemmArgs g{};
    g.A = A; g.W = W; g.C = C; g.bias = bias; g.rowdiv = rowdiv; g.ws = (float *)ws; g.colpart = colpart;
    g.lda = lda; g.ldw = ldw; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.act = act; g.ksplit = 1;
    g.bf16x3 = bf16x3; g.wsplit = wsplit; g.w_lo_off = (long)N * ldw; g.cv_Pout = 1; g.stat_shift = sshift;
    g.asplit = asplit; g.a_lo_off = (long)M * lda;
    g.l2n = l2n;
    if (f16 && p.big && wf16 && f16_two_per_cu(M, N, K, p)) p.bm = 128;
    if (f16 && p.big) {   // the tile flags live behind this plan's split-K partials; a workspace without room for them: the six-product kernel
        const size_t off = p.ksplit > 1 ? (size_t)p.ksplit * M * N * sizeof(float) : 0;
        if (ws && ws_bytes >= off + f16_flag_bytes(M, N, p.ksplit)) {
            g.f16 = 1;
            g.fixflags = reinterpret_cast<unsigned *>(static_cast<char *>(ws) + off);
        }
    }
    if (wf16) {   // a pre-split W is readable by gemm_f16_big_kernel only: the caller asks cofi_gemm_f16x3_eligible first
        if (!g.f16) return COFI_EUNSUPPORTED;
        g.wscale = W + (size_t)N * ldw;
    }
    if (int rc = set_a_norm(g, a_norm, K, M / frames, frames, p)) return rc;
    return launch(g, p, cofi_s(stream));
}

int conv_entry(const float *x, int ldx, const cofi_norm_desc_t *x_norm, int H, int W, int Cin, const float *Wt, int Cout, int ks, int stride,
               int pad, const float *bias, const float *res, int ldr, int act, int act_col0, float *y, int ldy, float *colpart, int stat_width,
               void *ws, size_t ws_bytes, int frames, cofi_stream_t stream) {
    if (!x || !Wt || !y || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (ks != 1 && ks != 3) || stride <= 0 || pad < 0) return COFI_EINVAL;
    if ((Cin & 3) || (ldx & 3) || ldx < Cin || ldy < Cout || (res && ldr < Cout) || ((uintptr_t)x & 15) || ((uintptr_t)Wt & 15)) return COFI_EINVAL;
    const int Ho = (H + 2 * pad - ks) / stride + 1, Wo = (W + 2 * pad - ks) / stride + 1;
    if (frames <= 0) return COFI_EINVAL;
    const int M = Ho * Wo * frames, K = ks * ks * Cin;
    const int bf16x3 = (act & COFI_GEMM_BF16X6) ? 2 : ((act & COFI_GEMM_BF16X3) ? 1 : 0);
    const int wsplit = (act & COFI_GEMM_W_SPLIT) ? 1 : 0;
    const int l2n = (act & COFI_GEMM_L2NORM) ? 1 : 0;
    const int f16 = (act & COFI_GEMM_F16X3) ? 1 : 0;
    const int wf16 = (act & COFI_GEMM_W_F16PRE) ? 1 : 0;
    act &= ~(COFI_GEMM_BF16X3 | COFI_GEMM_BF16X6 | COFI_GEMM_W_SPLIT | COFI_GEMM_L2NORM | COFI_GEMM_F16X3 | COFI_GEMM_W_F16PRE);
    if (wf16 && (!f16 || wsplit)) return COFI_EINVAL;
    if (act < 0 || act > 3 || (wsplit && bf16x3 == 0) || act_col0 < 0 || act_col0 > Cout) return COFI_EINVAL;
    if (l2n && Cout > 128) return COFI_EUNSUPPORTED;
    const int sshift = colpart ? stat_shift_of(stat_width, Cout) : 0;
    if (sshift < 0) return COFI_EINVAL;
    const int ldw = wsplit ? (K + 7) / 8 * 8 : K;   // pre-split planes: rows padded to 8 values
    // narrow 3 x 3 convolutions of a stack-mode batch: the direct kernel (conv_direct.inc: input halo split once, 9 taps read it shifted)
    {
        static const int direct_env = getenv("COFI_CONV_DIRECT") ? atoi(getenv("COFI_CONV_DIRECT")) : 1;   // A/B switch: 0 = never
        const int mode = g_force_direct ? g_force_direct : (direct_env ? 0 : -1);
        const long tiles4 = (long)frames * (H / 4) * (W / 64);
        const int th = (tiles4 >= 768 || g_force_direct == 2) ? 4 : 2;   // 4-row tiles once they fill the chip's workgroup slots (2 per CU) one and a half times
        const long tiles = (long)frames * (H / th) * (W / 64);
        const bool ok = bf16x3 == 2 && !wsplit && !l2n && ks == 3 && stride == 1 && pad == 1 && Cout == 64 && (Cin % 16) == 0 && Cin <= 512 && (W % 64) == 0 &&
                        (H % 4) == 0 && (!x_norm || x_norm->scale_shift) && (size_t)frames * H * W * ldx * sizeof(float) < 0xffffffffull &&
                        (size_t)Cout * K * sizeof(float) < 0xffffffffull;
        if (ok && !wf16 && mode >= 0 && (mode > 0 || tiles >= 256) && !g_force_bm) {
            GemmArgs g{};
            g.A = x; g.W = Wt; g.C = y; g.bias = bias; g.colpart = colpart; g.res = res;
            g.lda = ldx; g.ldw = K; g.ldc = ldy; g.ldr = ldr; g.M = M; g.N = Cout; g.K = K; g.act = act; g.ksplit = 1;
            g.bf16x3 = 2;
            g.cv_ks = 3; g.cv_H = H; g.cv_W = W; g.cv_Cin = Cin; g.cv_Wo = Wo; g.cv_stride = 1; g.cv_pad = 1; g.cv_Pout = Ho * Wo;
            g.stat_shift = sshift;
            g.act_col0 = act_col0;
            Plan dp{};
            dp.bm = 64; dp.bn = 64; dp.ksplit = 1; dp.pcfg = -1;
            if (int rc = set_a_norm(g, x_norm, Cin, H * W, frames, dp)) return rc;
            const dim3 grid(Cout / 64, (unsigned)tiles);
            if (th == 4) {
                if (g.an.part) hipLaunchKernelGGL((conv3x3_direct_kernel<true, 4>), grid, dim3(256), 0, cofi_s(stream), g);
                else hipLaunchKernelGGL((conv3x3_direct_kernel<false, 4>), grid, dim3(256), 0, cofi_s(stream), g);
            } else {
                if (g.an.part) hipLaunchKernelGGL((conv3x3_direct_kernel<true, 2>), grid, dim3(256), 0, cofi_s(stream), g);
                else hipLaunchKernelGGL((conv3x3_direct_kernel<false, 2>), grid, dim3(256), 0, cofi_s(stream), g);
            }
            return cofi_launch_status();
        }
    }
    Plan p = make_plan(M, Cout, K, l2n != 0, bf16x3);
    // the 256 x 128 kernel: 3 x 3 / stride 1 / pad <= 1 convolutions whose K-tiles of 32 lie inside one tap, 32-bit byte offsets into the input
    if (bf16x3 == 2 && !wsplit && !l2n && ks == 3 && stride == 1 && pad <= 1 && (Cin % 32) == 0 &&
        (size_t)frames * H * W * ldx * sizeof(float) < 0xffffffffull && !(x_norm && frames > 1 && (Ho * Wo) % 256))
        big_plan(M, Cout, K, p, f16 != 0);
    if (x_norm && frames > 1 && p.bm == 128 && p.bn == 64 && (Ho * Wo) % 128) p.bm = 64;   // a normalising tile stays inside one frame
    if (p.ksplit > 1 && (!ws || ws_bytes < (size_t)p.ksplit * M * Cout * sizeof(float))) return COFI_EWORKSPACE;
    GemmArgs g{};
    g.A = x; g.W = Wt; g.C = y; g.bias = bias; g.ws = (float *)ws; g.colpart = colpart; g.res = res;
    g.lda = ldx; g.ldw = ldw; g.ldc = ldy; g.ldr = ldr; g.M = M; g.N = Cout; g.K = K; g.act = act; g.ksplit = 1;
    g.bf16x3 = bf16x3; g.wsplit = wsplit; g.w_lo_off = (long)Cout * ldw;
    g.cv_ks = ks; g.cv_H = H; g.cv_W = W; g.cv_Cin = Cin; g.cv_Wo = Wo; g.cv_stride = stride; g.cv_pad = pad; g.cv_Pout = Ho * Wo;
    g.stat_shift = sshift;
    g.act_col0 = act_col0;
    g.l2n = l2n;
    if (f16 && p.big && wf16 && f16_two_per_cu(M, Cout, K, p)) p.bm = 128;
    if (f16 && p.big) {
        const size_t off = p.ksplit > 1 ? (size_t)p.ksplit * M * Cout * sizeof(float) : 0;
        if (ws && ws_bytes >= off + f16_flag_bytes(M, Cout, p.ksplit)) {
            g.f16 = 1;
            g.fixflags = reinterpret_cast<unsigned *>(static_cast<char *>(ws) + off);
        }
    }
    if (wf16) {
        if (!g.f16) return COFI_EUNSUPPORTED;
        g.wscale = Wt + (size_t)Cout * ldw;
    }
    if (int rc = set_a_norm(g, x_norm, Cin, H * W, frames, p)) return rc;   // statistics of the INPUT map: H * W rows per frame
    return launch(g, p, cofi_s(stream));
}

}  // namespace

// Would a COFI_GEMM_BF16X6 | COFI_GEMM_F16X3 launch of this shape run on gemm_f16_big_kernel (with a workspace of cofi_gemm_f32_workspace bytes)?
// The conditions of gemm_entry / conv_entry, restated: a caller holding a pre-split W (COFI_GEMM_W_F16PRE) asks before it passes it.
extern "C" int cofi_gemm_f16x3_eligible(int M, int N, int K, int pending_norm, int frames) {
    if (M <= 0 || N <= 0 || K <= 0 || frames <= 0) return 0;
    if (pending_norm && frames > 1 && (M / frames) % 256) return 0;
    Plan p = make_plan(M, N, K, false, 2);
    return big_plan(M, N, K, p, true) ? 1 : 0;
}

extern "C" int cofi_conv2d_f16x3_eligible(int H, int W, int Cin, int Cout, int ks, int stride, int pad, int ldx, int pending_norm, int frames) {
    if (H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || stride <= 0 || pad < 0 || frames <= 0 || (ks != 1 && ks != 3)) return 0;
    const int Ho = (H + 2 * pad - ks) / stride + 1, Wo = (W + 2 * pad - ks) / stride + 1;
    const int M = Ho * Wo * frames, K = ks * ks * Cin;
    if (!(ks == 3 && stride == 1 && pad <= 1 && (Cin % 32) == 0 && (size_t)frames * H * W * ldx * sizeof(float) < 0xffffffffull)) return 0;
    if (pending_norm && frames > 1 && (Ho * Wo) % 256) return 0;
    Plan p = make_plan(M, Cout, K, false, 2);
    return big_plan(M, Cout, K, p, true) ? 1 : 0;
}

extern "C" size_t cofi_gemm_f32_workspace(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const Plan p = make_plan(M, N, K, false), q = make_planes_plan(M, N, K), r = make_plan(M, N, K, false, 2);   // whichever kernel the operands select
    Plan b = r, bf = r;
    big_plan(M, N, K, b);                            // the six-product 256 x 128 kernel's plan ...
    const bool bigf = big_plan(M, N, K, bf, true);   // ... and the f16x3 kernel's (its own table of shapes and splits)
    const int ks = std::max(std::max(p.ksplit, b.ksplit), std::max(std::max(q.ksplit, r.ksplit), bf.ksplit));
    // split-K partials, then (shapes of the f16x3 kernel) its tile flags behind the partials of THAT plan
    const size_t partials = ks > 1 ? (size_t)ks * M * N * sizeof(float) : 0;
    return bigf ? std::max(partials, (bf.ksplit > 1 ? (size_t)bf.ksplit * M * N * sizeof(float) : 0) + f16_flag_bytes(M, N, bf.ksplit)) : partials;
}

extern "C" int cofi_gemm_f32_stat_slabs(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return cofi_cdiv(M, 64);   // every path (64-row tiles, 128-row tiles in two halves, split-K reduction) emits 64-row slabs
}

extern "C" int cofi_gemm_f32(const float *A, int lda, const float *W, int ldw, float *C, int ldc, int M, int N, int K,
                             const float *bias, const float *rowdiv, int act, void *ws, size_t ws_bytes, cofi_stream_t stream) {
    return gemm_entry(A, lda, nullptr, W, ldw, C, ldc, M, N, K, bias, rowdiv, act, nullptr, 1, ws, ws_bytes, 1, stream);
}

extern "C" int cofi_gemm_f32_colstats(const float *A, int lda, const float *W, int ldw, float *C, int ldc, int M, int N, int K,
                                      const float *bias, const float *rowdiv, int act, float *colpart, void *ws, size_t ws_bytes,
                                      cofi_stream_t stream) {
    return gemm_entry(A, lda, nullptr, W, ldw, C, ldc, M, N, K, bias, rowdiv, act, colpart, 1, ws, ws_bytes, 1, stream);
}

extern "C" int cofi_gemm_f32_fused(const float *A, int lda, const cofi_norm_desc_t *a_norm, const float *W, int ldw, float *C, int ldc,
                                   int M, int N, int K, const float *bias, const float *rowdiv, int act, float *colpart, int stat_width,
                                   void *ws, size_t ws_bytes, int frames, cofi_stream_t stream) {
    return gemm_entry(A, lda, a_norm, W, ldw, C, ldc, M, N, K, bias, rowdiv, act, colpart, stat_width, ws, ws_bytes, frames, stream);
}

extern "C" int cofi_gemm_f32_layernorm(const float *A, int lda, const float *W, int ldw, float *C, int ldc, int M, int N, int K,
                                       const float *bias, const float *gamma, const float *beta, float eps, int relu, const float *res,
                                       int ldr, void *ws, size_t ws_bytes, cofi_stream_t stream) {
    if (int rc = check_common(A, lda, W, ldw, C, ldc, M, N, K)) return rc;
    if (!gamma || !beta || N > 128 || (res && ldr < N)) return COFI_EINVAL;
    if (M == 0) return 0;
    const int bf16x3 = (relu & COFI_GEMM_BF16X6) ? 2 : ((relu & COFI_GEMM_BF16X3) ? 1 : 0);
    Plan p = make_plan(M, N, K, true, bf16x3);
    if (p.ksplit > 1 && (!ws || ws_bytes < (size_t)p.ksplit * M * N * sizeof(float))) return COFI_EWORKSPACE;
    const int wsplit = (relu & COFI_GEMM_W_SPLIT) ? 1 : 0;
    relu &= ~(COFI_GEMM_BF16X3 | COFI_GEMM_BF16X6 | COFI_GEMM_W_SPLIT | COFI_GEMM_F16X3);   // the fused LayerNorm never takes the 256 x 128 kernel
    if (wsplit && (bf16x3 == 0 || (ldw & 7))) return COFI_EINVAL;
    GemmArgs g{};
    g.A = A; g.W = W; g.C = C; g.bias = bias; g.ws = (float *)ws; g.ln_gamma = gamma; g.ln_beta = beta; g.res = res;
    g.lda = lda; g.ldw = ldw; g.ldc = ldc; g.ldr = ldr; g.M = M; g.N = N; g.K = K; g.ksplit = 1; g.ln_relu = relu; g.ln_eps = eps;
    g.bf16x3 = bf16x3; g.wsplit = wsplit; g.w_lo_off = (long)N * ldw; g.cv_Pout = 1;
    return launch(g, p, cofi_s(stream));
}

extern "C" int cofi_conv2d_nhwc(const float *x, int ldx, int H, int W, int Cin, const float *Wt, int Cout, int ks, int stride, int pad,
                                const float *bias, const float *res, int ldr, int act, float *y, int ldy, float *colpart, void *ws,
                                size_t ws_bytes, int frames, cofi_stream_t stream) {
    return conv_entry(x, ldx, nullptr, H, W, Cin, Wt, Cout, ks, stride, pad, bias, res, ldr, act, 0, y, ldy, colpart, 1, ws, ws_bytes, frames, stream);
}

extern "C" int cofi_conv2d_nhwc_fused(const float *x, int ldx, const cofi_norm_desc_t *x_norm, int H, int W, int Cin, const float *Wt, int Cout,
                                      int ks, int stride, int pad, const float *bias, const float *res, int ldr, int act, int act_col0, float *y,
                                      int ldy, float *colpart, int stat_width, void *ws, size_t ws_bytes, int frames, cofi_stream_t stream) {
    return conv_entry(x, ldx, x_norm, H, W, Cin, Wt, Cout, ks, stride, pad, bias, res, ldr, act, act_col0, y, ldy, colpart, stat_width, ws,
                      ws_bytes, frames, stream);
}

extern "C" int cofi_split_bf16_planes(const float *W, int ldw, int N, int K, void *planes, int ldp, int nplanes, cofi_stream_t stream) {
    if (!W || !planes || N <= 0 || K <= 0 || ldw < K || ldp < K || (ldp & 7) || ((uintptr_t)planes & 15) || (nplanes != 2 && nplanes != 3)) return COFI_EINVAL;
    const size_t total = (size_t)N * (ldp >> 1);
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256), 0, cofi_s(stream),
                       W, ldw, N, K, (unsigned *)planes, ldp, nplanes);
    return cofi_launch_status();
}

// ---- tuning / test hooks, declared in include/cofi_hip_tune.h (not part of the drop-in ABI of include/cofi_hip.h).  Every override is
// state of the calling thread.
extern "C" int cofi_tune_force_planes(int cfg, int ksplit) {
    if (cfg < -2 || cfg >= kNumPlanesCfg || ksplit < 0) return COFI_EINVAL;
    g_force_pcfg = cfg; g_force_pks = ksplit;
    return 0;
}

// 256 x 128 bf16x6 kernel: mode 1 = on every eligible launch (ksplit 0 = chosen by big_plan), -1 = never, 0 = table + heuristic
extern "C" int cofi_tune_force_big(int mode, int ksplit) {
    if (mode < -1 || mode > 1 || ksplit < 0 || ksplit > 64) return COFI_EINVAL;
    g_force_big = mode; g_force_big_ks = ksplit;
    return 0;
}

extern "C" int cofi_tune_force_conv_direct(int mode) {
    if (mode < -1 || mode > 2) return COFI_EINVAL;
    g_force_direct = mode;
    return 0;
}

extern "C" int cofi_tune_big_debug(int flags) {   // 64: the generic row-wise epilogue (identical bits); 256: the four-wave f16x3 geometry (identical bits); other bits: unused
    g_big_dbg = flags;
    return 0;
}

// Diagnostic counter of gemm_f16_big_kernel: tiles re-split because they left the fp16 window of their panel's scale (all launches since the
// last reset).  reset != 0 zeroes it after reading.  Synchronises the device.
extern "C" long cofi_tune_f16x3_resplit_events(int reset) {
    unsigned long long v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_f16_resplit_events), sizeof(v), 0, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (reset) {
        const unsigned long long z = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_f16_resplit_events), &z, sizeof(z), 0, hipMemcpyHostToDevice) != hipSuccess) return -1;
    }
    return (long)v;
}

// Host-side census (bench.py's roofline of a launch list that mixes the two arithmetics): contractions the CALLING THREAD enqueued on
// gemm_f16_big_kernel since the last reset -> their count, *flops = the sum of their 2 M N K.
extern "C" long cofi_tune_f16x3_launch_flops(int reset, double *flops) {
    const long n = g_f16_launches;
    if (flops) *flops = g_f16_flops;
    if (reset) { g_f16_launches = 0; g_f16_flops = 0.0; }
    return n;
}

extern "C" int cofi_tune_force_plan(int bm, int bn, int ksplit) {
    const bool ok = (bm == 0 && bn == 0) || ((bm == 64 || bm == 128) && (bn == 64 || bn == 128));   // 128 x 64: bf16x6 only (make_plan widens it otherwise)
    if (!ok || ksplit < 0) return COFI_EINVAL;
    g_force_bm = bm; g_force_bn = bn; g_force_ks = ksplit;
    return 0;
}
