// Backward kernels of the training path: the adjoints of the three operators that carry the gradient
// through the network - KPConv aggregation, attention, convolution (as im2col / col2im around the GEMM) - and of the two neighbour
// gathers.  The GEMMs of the backward are the forward's own kernel on transposed operands (cofii2p_amd/autograd.py).
// Reference: what torch.autograd derives for model/kpconv/kpconv.py:91-116, model/transformer/linear_attention.py:56-79,
// model/kpconv/functional.py:5-21,53-66 when train.py:285 calls loss.backward().
//
// SCATTER-FREE.  Every forward gather  y[m] = f(x[idx[m, h]])  has the adjoint  dx[j] = sum over the pairs (m, h) with idx[m, h] == j.
// Instead of float atomics (order-dependent sums) the host hands over the TRANSPOSED table of idx in CSR form - pair ids m * H + h
// sorted by (j, m, h), offsets per j - and one wave per support row j walks its own list in that fixed order: every gradient row is
// written once, coalesced, and the result is bit-reproducible.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------------------
// KPConv aggregation, adjoint w.r.t. the support features:
//   forward   agg[m, k, c] = sum_h w(m, h, k) f[idx[m, h], c],   w = max(1 - |(s[idx] - q[m]) - kp[k]| / sigma, 0)   (kpconv.py:93-105)
//   adjoint   df[j, c]     = sum_{(m, h): idx[m, h] = j} sum_k w(m, h, k) dagg[m, k, c]
// One wave per (support row j, chunk of 64 channels), lane = channel.  The influence weights of a pair are recomputed from the
// geometry (uniform over the wave); only kernel points with w > 0 - typically 1-3 of 15 - cost a (coalesced, 256-byte) load of dagg.
// C = 32: the two half-waves take alternate pairs and are folded once at the end (fixed order).
struct KpBwdArgs {
    const float *dagg, *q_pts, *s_pts, *kp;
    const int32_t *pairs, *offsets;
    float *dfeats;
    int ldd, ldf, N, C, H;
    float inv_sigma;
};

__global__ __launch_bounds__(256) void kpconv_aggregate_bwd_kernel(KpBwdArgs a) {
    // wave-private LDS: the influence weights of the 64 pairs of a round, [pair][16], and per pair its query row and the bit mask of its
    // non-zero weights.  A wave reads only what it wrote itself (LDS executes one wave's instructions in order): no workgroup barrier.
    __shared__ float s_w[4][64][16];
    __shared__ int s_m[4][64], s_mask[4][64];
    const int wslot = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int CW = a.C >= 64 ? 64 : a.C;          // channels a wave covers per pass (C < 64: 64 / CW pairs side by side)
    const int chunks = (a.C + CW - 1) / CW;
    const int j = wave / chunks, ch = wave - j * chunks;
    if (j >= a.N) return;
    const int NS = 64 / CW, sub = lane / CW, c = ch * CW + (lane - sub * CW);
    const bool c_ok = c < a.C;
    const float sx = a.s_pts[3 * j], sy = a.s_pts[3 * j + 1], sz = a.s_pts[3 * j + 2];
    const int p0 = a.offsets[j], p1 = a.offsets[j + 1];
    float acc = 0.f;
    // 64 pairs per round, lane = pair: one coalesced read of the pair ids, one gather of the query positions, the 15 influence weights of
    // the pair computed in this lane and parked in LDS with the mask of the non-zero ones (typically 1-3 of 15).  The accumulation loop
    // then walks the pairs: per pair one broadcast read of (row, mask) and, per set bit, one broadcast weight read + one coalesced
    // 256-byte load of the dagg row segment - nothing is spent on the kernel points a neighbour does not reach.
    for (int base = p0; base < p1; base += 64) {
        const int cnt = min(64, p1 - base);
        int m = 0, mask = 0;
        if (lane < cnt) {
            m = a.pairs[base + lane] / a.H;
            const float ox = sx - a.q_pts[3 * m], oy = sy - a.q_pts[3 * m + 1], oz = sz - a.q_pts[3 * m + 2];
#pragma unroll
            for (int k = 0; k < 15; ++k) {
                const float dx = ox - a.kp[3 * k], dy = oy - a.kp[3 * k + 1], dz = oz - a.kp[3 * k + 2];
                const float sq = (dx * dx + dy * dy) + dz * dz;
                const float w = fmaxf(1.0f - __builtin_amdgcn_sqrtf(sq) * a.inv_sigma, 0.0f);
                s_w[wslot][lane][k] = w;
                mask |= (w > 0.f) ? (1 << k) : 0;
            }
        }
        s_m[wslot][lane] = m;
        s_mask[wslot][lane] = mask;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int i = sub; i < cnt; i += NS) {      // NS == 1: i is wave-uniform; NS > 1: the lane groups take alternate pairs
            int bits = s_mask[wslot][i];
            const float *d = a.dagg + (size_t)s_m[wslot][i] * a.ldd + c;
            while (bits) {
                const int k = __builtin_ctz(bits);
                bits &= bits - 1;
                if (c_ok) acc = fmaf(s_w[wslot][i][k], d[(size_t)k * a.C], acc);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // fold the NS side-by-side partial sums (lanes with equal channel) in a fixed order
    for (int o = 32; o >= CW; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (c_ok && sub == 0) a.dfeats[(size_t)j * a.ldf + c] = acc;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Neighbour max-pool with the arg-max (functional.py:53-66: x padded with a zero row, max over the H neighbours; the first
// neighbour attaining the maximum receives the gradient).  One wave per (query, 32-channel chunk), as the serving kernel: lane
// (g = l >> 3, c4 = l & 7) reads float4 number c4 of the chunk for neighbours h = 8 i + g - one load instruction fetches 8 neighbour
// rows x 128 contiguous bytes - and keeps (max, first h) per channel; the 8 lane groups are folded with xor-shuffles (larger value,
// then lower h).  arg: (M, C) bytes (H <= 256).
__global__ __launch_bounds__(256) void neighbor_maxpool_arg_kernel(const float *x, int ldx, int N, int C, const int32_t *idx, int M, int H, float *out, int ldo,
                                                                   uint8_t *arg) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int chunks = (C + 31) / 32;
    const int m = wave % M, ch = wave / M;     // chunks are the slow axis: a chunk's source slice stays in L2 while its queries run
    if (ch >= chunks) return;
    const int g = lane >> 3, c0 = ch * 32 + 4 * (lane & 7);
    const bool c_ok = c0 < C;                   // C % 4 == 0 (host-checked)
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bh[4] = {0, 0, 0, 0};
    for (int h = g; h < H; h += 8) {
        const int id = idx[(size_t)m * H + h];
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (id >= 0 && id < N && c_ok) v = *reinterpret_cast<const f32x4 *>(x + (size_t)id * ldx + c0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (v[e] > best[e]) { best[e] = v[e]; bh[e] = h; }
    }
#pragma unroll
    for (int o = 8; o < 64; o <<= 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ov = __shfl_xor(best[e], o, 64);
            const int oh = __shfl_xor(bh[e], o, 64);
            if (ov > best[e] || (ov == best[e] && oh < bh[e])) { best[e] = ov; bh[e] = oh; }
        }
    if (g == 0 && c_ok) {
        *reinterpret_cast<f32x4 *>(out + (size_t)m * ldo + c0) = f32x4{best[0], best[1], best[2], best[3]};
        *reinterpret_cast<uchar4 *>(arg + (size_t)m * C + c0) = make_uchar4((unsigned char)bh[0], (unsigned char)bh[1], (unsigned char)bh[2], (unsigned char)bh[3]);
    }
}

// dx[j, c] = sum over the pairs (m, h) of row j with arg[m, c] == h of dy[m, c]; one wave per (j, 64-channel chunk).  The pair list is
// staged 64 at a time by lane = pair (one coalesced read) into wave-private LDS, so the walk's loads (arg byte, dy) are independent.
__global__ __launch_bounds__(256) void neighbor_maxpool_bwd_kernel(const float *dy, int ldy, const uint8_t *arg, int C, int H, const int32_t *pairs,
                                                                   const int32_t *offsets, int N, float *dx, int ldx) {
    __shared__ int s_pair[4][64];
    const int wslot = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int chunks = (C + 63) / 64;
    const int j = wave / chunks, c = (wave - j * chunks) * 64 + lane;
    if (j >= N) return;
    const bool c_ok = c < C;
    const int p0 = offsets[j], p1 = offsets[j + 1];
    float acc = 0.f;
    for (int base = p0; base < p1; base += 64) {
        const int cnt = min(64, p1 - base);
        s_pair[wslot][lane] = lane < cnt ? pairs[base + lane] : 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (c_ok)
            for (int i = 0; i < cnt; ++i) {
                const int pair = s_pair[wslot][i], m = pair / H, h = pair - m * H;
                if (arg[(size_t)m * C + c] == h) acc += dy[(size_t)m * ldy + c];
            }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (c_ok) dx[(size_t)j * ldx + c] = acc;
}

// adjoint of out[m] = x[idx[m * stride]] (functional.py:5-21 nearest_upsample; any row gather): dx[j] = sum over the rows m of
// list j of dy[m].  pairs = row ids m sorted by (j, m).
__global__ __launch_bounds__(256) void gather_rows_bwd_kernel(const float *dy, int ldy, int C, const int32_t *pairs, const int32_t *offsets, int N,
                                                              float *dx, int ldx) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int chunks = (C + 63) / 64;
    const int j = wave / chunks, c = (wave - j * chunks) * 64 + lane;
    if (j >= N || c >= C) return;
    float acc = 0.f;
    for (int p = offsets[j]; p < offsets[j + 1]; ++p) acc += dy[(size_t)pairs[p] * ldy + c];
    dx[(size_t)j * ldx + c] = acc;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Convolution as a GEMM with an explicit operand (training only: the weight gradient needs the unfolded input anyway).
//   col[(oy, ox), (dy * ks + dx) * C + c] = x[(oy * stride + dy - pad, ox * stride + dx - pad), c]   (0 outside the map)
// and its adjoint in gather form (one thread per input pixel and 4 channels sums the taps that touched it: no atomics).
__global__ void im2col_nhwc_kernel(const float *x, int ldx, int H, int W, int C, int ks, int stride, int pad, int Ho, int Wo, float *col, int ldc) {
    const int C4 = C >> 2, taps = ks * ks;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)Ho * Wo * taps * C4) return;
    const int c4 = (int)(t % C4);
    const int tap = (int)((t / C4) % taps);
    const int m = (int)(t / ((size_t)C4 * taps));
    const int oy = m / Wo, ox = m - oy * Wo, dy = tap / ks, dx = tap - dy * ks;
    const int y = oy * stride + dy - pad, xx = ox * stride + dx - pad;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (y >= 0 && y < H && xx >= 0 && xx < W) v = *reinterpret_cast<const f32x4 *>(x + ((size_t)y * W + xx) * ldx + 4 * c4);
    *reinterpret_cast<f32x4 *>(col + (size_t)m * ldc + (size_t)tap * C + 4 * c4) = v;
}

__global__ void col2im_nhwc_kernel(const float *dcol, int ldc, int H, int W, int C, int ks, int stride, int pad, int Ho, int Wo, float *dx, int ldx) {
    const int C4 = C >> 2;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)H * W * C4) return;
    const int c4 = (int)(t % C4);
    const int pix = (int)(t / C4);
    const int y = pix / W, xx = pix - y * W;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int dy = 0; dy < ks; ++dy) {
        const int ny = y + pad - dy;
        if (ny < 0 || ny % stride) continue;
        const int oy = ny / stride;
        if (oy >= Ho) continue;
        for (int dxx = 0; dxx < ks; ++dxx) {
            const int nx = xx + pad - dxx;
            if (nx < 0 || nx % stride) continue;
            const int ox = nx / stride;
            if (ox >= Wo) continue;
            acc += *reinterpret_cast<const f32x4 *>(dcol + ((size_t)oy * Wo + ox) * ldc + (size_t)(dy * ks + dxx) * C + 4 * c4);
        }
    }
    *reinterpret_cast<f32x4 *>(dx + (size_t)pix * ldx + 4 * c4) = acc;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Attention backward (recompute style: the (L, S) probability matrix is never stored) on the exact fp32 matrix instruction
// v_mfma_f32_32x32x2_f32, head dimension 32.  With P = softmax(scale Q K^T), O = P V:
//   delta_i = dO_i . O_i,   dP = dO V^T,   dS = P o (dP - delta) scale,   dQ = dS K,   dK = dS^T Q,   dV = P^T dO.
// Two kernels, one wave per 32-row block of one head:
//   attention_bwd_dq_kernel   (query block)  pass 1 recomputes the row log-sum-exp, pass 2 accumulates dQ; writes {lse, delta} per
//                                            query to the workspace,
//   attention_bwd_dkv_kernel  (key block)    loops over the query blocks with that workspace and accumulates dK, dV.
// Operand layouts follow the forward kernel (attention.hip): a score tile is computed TRANSPOSED so that one lane owns one row of the
// block it accumulates for (the query in the dQ kernel, the key in the dK/dV kernel) - softmax statistics are per-lane scalars - and
// the MFMA D layout of that tile is directly the B operand of the second product.  MFMA 32x32x2: lane l supplies A[l & 31][l >> 5] and
// B[l >> 5][l & 31]; D register r of lane l is D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31].
constexpr int AD = 32, ALD = AD + 4;   // head dimension; LDS tile row stride (floats): b128 rows and b32 columns both conflict free

struct AttnBwdArgs {
    const float *Q, *K, *V, *O, *dO;
    float *dQ, *dK, *dV, *stat;   // stat (H, L, 2) = {lse in log2 units, delta}
    int ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv, L, S, H;
    float scale, scale_log2e;
};

__device__ __forceinline__ float bw_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// stage rows [r0, r0 + 32) (clamped to nrows - 1) x 32 columns starting at `src` into an LDS tile; one wave
__device__ __forceinline__ void stage_tile(float *tile, const float *src, int ld, int r0, int nrows, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = lane + 64 * i, r = e >> 3, c4 = e & 7;
        const int row = min(r0 + r, nrows - 1);
        *reinterpret_cast<f32x4 *>(tile + r * ALD + 4 * c4) = *reinterpret_cast<const f32x4 *>(src + (size_t)row * ld + 4 * c4);
    }
}

// acc(32 x 32) = T . F^T: T = LDS tile rows (A operand, row = lane & 31), F = per-lane fragments frag[c][e] = X[row(lane & 31)][8c + 4 (lane >> 5) + e]
__device__ __forceinline__ f32x16 tile_dot(const float *tile, const f32x4 (&frag)[4], int li, int lh) {
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const f32x4 t = *reinterpret_cast<const f32x4 *>(tile + li * ALD + 8 * c + 4 * lh);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(t[e], frag[c][e], acc, 0, 0, 0);
    }
    return acc;
}

// acc(32 x 32)[d][col = lane & 31] += sum_t tile[row_t][d] * b[t], row_t = (t & 3) + 8 (t >> 2) + 4 (lane >> 5): the transposed tile times a
// D-layout operand
__device__ __forceinline__ void tile_t_dot_acc(f32x16 &acc, const float *tile, const f32x16 &b, int li, int lh) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int row = (t & 3) + 8 * (t >> 2) + 4 * lh;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tile[row * ALD + li], b[t], acc, 0, 0, 0);
    }
}

__device__ __forceinline__ void load_frag(f32x4 (&frag)[4], const float *row_ptr, int lh) {
#pragma unroll
    for (int c = 0; c < 4; ++c) frag[c] = *reinterpret_cast<const f32x4 *>(row_ptr + 8 * c + 4 * lh);
}

// rows d = 8 g + 4 lh + e of the transposed accumulator are 4 contiguous columns of the output row this lane owns
__device__ __forceinline__ void store_acc_t(float *row_ptr, const f32x16 &acc, int lh) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        *reinterpret_cast<f32x4 *>(row_ptr + 8 * g + 4 * lh) = v;
    }
}

constexpr int ABW = 4;   // waves per workgroup: each takes every ABW-th block of the loop axis; partial results are folded through LDS in wave order

// Wave-private LDS tiles: a wave only reads what it wrote itself, and the LDS executes one wave's instructions in order, so the loops
// need no workgroup barrier (their trip counts differ between waves); this fence only pins the compiler's ordering.
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// sum the ABW waves' transposed accumulators (lane (li, lh) register r = element [d = (r & 3) + 8 (r >> 2) + 4 lh][row li]) through LDS;
// wave 0 returns the total.  red: ABW x 32 x 33 floats.
__device__ __forceinline__ f32x16 fold_waves(float *red, const f32x16 &acc, int wave, int li, int lh) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 33 + li] = acc[r];
    __syncthreads();
    f32x16 tot = acc;
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = (r & 3) + 8 * (r >> 2) + 4 * lh;
            float t = red[d * 33 + li];
#pragma unroll
            for (int w = 1; w < ABW; ++w) t += red[(w * 32 + d) * 33 + li];
            tot[r] = t;
        }
    }
    return tot;
}

__global__ __launch_bounds__(64 * ABW) void attention_bwd_dq_kernel(AttnBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float s_tiles[ABW][2][32 * ALD];
    __shared__ float s_red[ABW * 32 * 33], s_ml[ABW][2][32];
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *s_k = s_tiles[wave][0], *s_v = s_tiles[wave][1];
    const int qb = blockIdx.x, h = blockIdx.y;
    const int q = min(qb * 32 + li, a.L - 1);
    const bool q_ok = qb * 32 + li < a.L;
    f32x4 qf[4], dof[4], of[4];
    load_frag(qf, a.Q + (size_t)q * a.ldq + h * AD, lh);
    load_frag(dof, a.dO + (size_t)q * a.lddo + h * AD, lh);
    load_frag(of, a.O + (size_t)q * a.ldo + h * AD, lh);
    float delta = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) delta = fmaf(dof[c][e], of[c][e], delta);
    delta += __shfl_xor(delta, 32, 64);
    const int nkb = (a.S + 31) / 32;
    // ---- pass 1: row maximum and sum of exponentials (log2 domain) over this wave's key blocks, this lane's 16 keys of each
    float mx = -INFINITY, sum = 0.f;
    for (int kb = wave; kb < nkb; kb += ABW) {
        wave_lds_fence();
        stage_tile(s_k, a.K + h * AD, a.ldk, kb * 32, a.S, lane);
        wave_lds_fence();
        const f32x16 st = tile_dot(s_k, qf, li, lh);
        float sv[16], bm = mx;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            sv[r] = key < a.S ? st[r] * a.scale_log2e : -INFINITY;
            bm = fmaxf(bm, sv[r]);
        }
        if (bm > -INFINITY) {
            sum *= bw_exp2(mx - bm);
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += bw_exp2(sv[r] - bm);
            mx = bm;
        }
    }
    {   // join the two halves of the query's keys, then the waves (fixed order)
        const float om = __shfl_xor(mx, 32, 64), os = __shfl_xor(sum, 32, 64);
        const float m2 = fmaxf(mx, om);
        sum = (mx > -INFINITY ? sum * bw_exp2(mx - m2) : 0.f) + (om > -INFINITY ? os * bw_exp2(om - m2) : 0.f);
        mx = m2;
        if (lh == 0) { s_ml[wave][0][li] = mx; s_ml[wave][1][li] = sum; }
        __syncthreads();
        float mt = s_ml[0][0][li];
#pragma unroll
        for (int w = 1; w < ABW; ++w) mt = fmaxf(mt, s_ml[w][0][li]);
        float stt = 0.f;
#pragma unroll
        for (int w = 0; w < ABW; ++w) stt += s_ml[w][0][li] > -INFINITY ? s_ml[w][1][li] * bw_exp2(s_ml[w][0][li] - mt) : 0.f;
        mx = mt;
        sum = stt;
    }
    const float lse = mx + __builtin_amdgcn_logf(sum);   // v_log_f32 = log2
    if (q_ok && lh == 0 && wave == 0) {
        float *st = a.stat + ((size_t)h * a.L + q) * 2;
        st[0] = lse;
        st[1] = delta;
    }
    // ---- pass 2: dQ^T[d, q] += K^T dS^T over this wave's key blocks
    f32x16 dq = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int kb = wave; kb < nkb; kb += ABW) {
        wave_lds_fence();
        stage_tile(s_k, a.K + h * AD, a.ldk, kb * 32, a.S, lane);
        stage_tile(s_v, a.V + h * AD, a.ldv, kb * 32, a.S, lane);
        wave_lds_fence();
        const f32x16 st = tile_dot(s_k, qf, li, lh);
        const f32x16 dp = tile_dot(s_v, dof, li, lh);
        f32x16 ds;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float p = key < a.S ? bw_exp2(st[r] * a.scale_log2e - lse) : 0.f;
            ds[r] = p * (dp[r] - delta) * a.scale;
        }
        tile_t_dot_acc(dq, s_k, ds, li, lh);
    }
    dq = fold_waves(s_red, dq, wave, li, lh);
    if (q_ok && wave == 0) store_acc_t(a.dQ + (size_t)q * a.lddq + h * AD, dq, lh);
}

__global__ __launch_bounds__(64 * ABW) void attention_bwd_dkv_kernel(AttnBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float s_tiles[ABW][2][32 * ALD];
    __shared__ float s_red[ABW * 32 * 33], s_stats[ABW][64];
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *s_q = s_tiles[wave][0], *s_do = s_tiles[wave][1], *s_stat = s_stats[wave];
    const int kb = blockIdx.x, h = blockIdx.y;
    const int key = min(kb * 32 + li, a.S - 1);
    const bool k_ok = kb * 32 + li < a.S;
    f32x4 kf[4], vf[4];
    load_frag(kf, a.K + (size_t)key * a.ldk + h * AD, lh);
    load_frag(vf, a.V + (size_t)key * a.ldv + h * AD, lh);
    f32x16 dk = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dv = dk;
    const int nqb = (a.L + 31) / 32;
    for (int qb = wave; qb < nqb; qb += ABW) {
        wave_lds_fence();
        stage_tile(s_q, a.Q + h * AD, a.ldq, qb * 32, a.L, lane);
        stage_tile(s_do, a.dO + h * AD, a.lddo, qb * 32, a.L, lane);
        {
            const int qi = min(qb * 32 + (lane >> 1), a.L - 1);
            s_stat[lane] = a.stat[((size_t)h * a.L + qi) * 2 + (lane & 1)];
        }
        wave_lds_fence();
        const f32x16 s = tile_dot(s_q, kf, li, lh);      // S[i][j]: rows = queries, column = this lane's key
        const f32x16 dp = tile_dot(s_do, vf, li, lh);    // dP[i][j] = dO_i . V_j
        f32x16 p, ds;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float pv = qb * 32 + i < a.L ? bw_exp2(s[r] * a.scale_log2e - s_stat[2 * i]) : 0.f;
            p[r] = pv;
            ds[r] = pv * (dp[r] - s_stat[2 * i + 1]) * a.scale;
        }
        tile_t_dot_acc(dv, s_do, p, li, lh);    // dV^T[d, j] += dO^T P
        tile_t_dot_acc(dk, s_q, ds, li, lh);    // dK^T[d, j] += Q^T dS
    }
    dk = fold_waves(s_red, dk, wave, li, lh);
    __syncthreads();
    dv = fold_waves(s_red, dv, wave, li, lh);
    if (k_ok && wave == 0) {
        store_acc_t(a.dK + (size_t)key * a.lddk + h * AD, dk, lh);
        store_acc_t(a.dV + (size_t)key * a.lddv + h * AD, dv, lh);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Column sums of a (M, C) matrix - the bias gradients - in two fixed-order stages: partial sums of row blocks, then their fold.
__global__ __launch_bounds__(256) void col_sum_partial_kernel(const float *x, int ldx, int M, int C, int rows_per_block, float *part) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float s = 0.f;
    if (c < C)
        for (int m = r0 + ph; m < r1; m += 4) s += x[(size_t)m * ldx + c];
    red[ph][cl] = s;
    __syncthreads();
    if (ph == 0 && c < C) part[(size_t)blockIdx.y * C + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

__global__ __launch_bounds__(256) void col_sum_final_kernel(const float *part, int RB, int C, float *out) {   // 64 columns x 4 phases per workgroup
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (c < C)
        for (int b = ph; b < RB; b += 4) s += part[(size_t)b * C + c];
    red[ph][cl] = s;
    __syncthreads();
    if (ph == 0 && c < C) out[c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}


// ------------------------------------------------------------------------------------------------------------------------------
// Backward of F.normalize(x, dim=1) (network.py:83-84, 90, 125-126: descriptors are unit rows), a wave per row:
//   n = ||x_m||;  dx = dy / n - x <dy, x> / n^3   (n >= eps),   dx = dy / eps   (the clamped norm is a constant)
__global__ __launch_bounds__(256) void l2norm_rows_bwd_kernel(const float *x, int ldx, const float *dy, int lddy, int M, int C, float eps, float *dx,
                                                             int lddx) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int lane = threadIdx.x & 63;
    float q = 0.f, t = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float v = x[(size_t)m * ldx + c];
        q = fmaf(v, v, q);
        t = fmaf(dy[(size_t)m * lddy + c], v, t);
    }
    const float n = sqrtf(wave_sum(q));
    t = wave_sum(t);
    const float inv = 1.0f / fmaxf(n, eps);
    const float coef = n >= eps ? t * inv * inv * inv : 0.f;
    for (int c = lane; c < C; c += 64) dx[(size_t)m * lddx + c] = dy[(size_t)m * lddy + c] * inv - x[(size_t)m * ldx + c] * coef;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Adjoint of the bilinear x2 up-sampling (align_corners = False, imagenet.py:433) of a pixel-major (h w, C1) map: the gradient of the
// (2h 2w, C1 [+ C2]) output's first C1 columns gathered per INPUT pixel - no atomics, fixed order.  Along one axis the output sample Y reads
// in[k-1], in[k] with 0.25 / 0.75 (Y = 2k, k >= 1; Y = 0 reads in[0] alone) and in[k], in[min(k+1, n-1)] with 0.75 / 0.25 (Y = 2k + 1), hence
// input i receives from Y = 2i-1 (0.25), 2i (0.75; 1 at i = 0), 2i+1 (0.75; 1 at i = n-1), 2i+2 (0.25).
__device__ inline int up2_taps(int i, int n, int *Y, float *wgt) {
    int t = 0;
    if (i >= 1) { Y[t] = 2 * i - 1; wgt[t++] = 0.25f; }
    Y[t] = 2 * i; wgt[t++] = i == 0 ? 1.0f : 0.75f;
    Y[t] = 2 * i + 1; wgt[t++] = i == n - 1 ? 1.0f : 0.75f;
    if (i + 1 <= n - 1) { Y[t] = 2 * i + 2; wgt[t++] = 0.25f; }
    return t;
}

__global__ __launch_bounds__(256) void upsample2x_bwd_nhwc_kernel(const float *dout, int lddo, int C1, int h, int w, float *dlow, int lddl) {
    const int ct = C1 >> 2, W = 2 * w;
    const size_t total = (size_t)h * w * ct;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % ct) * 4, p = (int)(e / ct), i = p / w, j = p - i * w;
        int Ys[4], Xs[4];
        float wy[4], wx[4];
        const int ny = up2_taps(i, h, Ys, wy), nx = up2_taps(j, w, Xs, wx);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int a = 0; a < ny; ++a) {
            f32x4 row = {0.f, 0.f, 0.f, 0.f};
            for (int b = 0; b < nx; ++b) {
                const f32x4 g = *reinterpret_cast<const f32x4 *>(dout + ((size_t)Ys[a] * W + Xs[b]) * lddo + c);
                for (int k = 0; k < 4; ++k) row[k] = fmaf(wx[b], g[k], row[k]);
            }
            for (int k = 0; k < 4; ++k) acc[k] = fmaf(wy[a], row[k], acc[k]);
        }
        *reinterpret_cast<f32x4 *>(dlow + (size_t)p * lddl + c) = acc;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// F.normalize(x, dim=0) of a (M, C) matrix - transformer.py:53 normalises Q over the tokens - forward and backward, two launches each:
// column partials of a * b over row blocks (a = b = x: sum of squares; a = dy, b = x: <dy, x>), then an apply kernel whose workgroups
// first fold the partials of their 64 columns in a fixed order.  With n_c = ||x[:, c]||, inv_c = 1 / max(n_c, eps):
//   y = x inv_c;   dx = dy inv_c - x (n_c >= eps ? <dy, x>_c inv_c^3 : 0)      (a clamped norm is a constant)
__global__ __launch_bounds__(256) void col_dot_partial_kernel(const float *a, int lda, const float *b, int ldb, int M, int C, int rows_per_block,
                                                              float *part) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float s = 0.f;
    if (c < C)
        for (int m = r0 + ph; m < r1; m += 4) s = fmaf(a[(size_t)m * lda + c], b[(size_t)m * ldb + c], s);
    red[ph][cl] = s;
    __syncthreads();
    if (ph == 0 && c < C) part[(size_t)blockIdx.y * C + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

__device__ inline float fold_col_partials(const float *part, int RB, int C, int c, int cl, int ph, float (*red)[64]) {
    float s = 0.f;
    if (c < C)
        for (int b = ph; b < RB; b += 4) s += part[(size_t)b * C + c];
    red[ph][cl] = s;
    __syncthreads();
    return (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

// stats (2, C): inv | live.  BWD: part holds <dy, x>; stats are read, dx written.
template <bool BWD>
__global__ __launch_bounds__(256) void col_normalize_apply_kernel(const float *x, int ldx, const float *dy, int lddy, const float *part, int RB, int M,
                                                                  int C, float eps, int rows_per_block, float *stats, float *out, int ldo) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const float t = fold_col_partials(part, RB, C, c, cl, ph, red);
    if (c >= C) return;
    float inv, coef = 0.f;
    if (!BWD) {
        const float n = sqrtf(t);
        inv = 1.0f / fmaxf(n, eps);
        if (blockIdx.y == 0 && ph == 0) { stats[c] = inv; stats[C + c] = n >= eps ? 1.0f : 0.0f; }
    } else {
        inv = stats[c];
        coef = stats[C + c] != 0.0f ? t * inv * inv * inv : 0.0f;
    }
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    for (int m = r0 + ph; m < r1; m += 4) {
        const float xv = x[(size_t)m * ldx + c];
        out[(size_t)m * ldo + c] = BWD ? dy[(size_t)m * lddy + c] * inv - xv * coef : xv * inv;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Backward of  y = leaky( gn(x; mean_G, rstd_G) * gamma + beta + res, slope )  - nn.GroupNorm over all rows (modules.py:45-49), with
// groups == C the affine-less InstanceNorm of imagenet.py / network.py:42-43 and the train-mode BatchNorm of imagenet.py:381-394 - in three
// fixed-order stages.  With g = dy * (y > 0 ? 1 : slope), xh = (x - mean) rstd, n = rows * channels per group:
//   dbeta_c = sum_rows g,  dgamma_c = sum_rows g xh,  S1_G = sum_{c in G} gamma_c dbeta_c,  S2_G = sum_{c in G} gamma_c dgamma_c,
//   dx = rstd_G (g gamma_c - S1_G / n - xh S2_G / n)          (constant statistics - eval-mode BatchNorm: dx = rstd g gamma),   dres = g.
struct NormBwdArgs {
    const float *x, *y, *dy, *stats, *gamma;
    float *part, *dgamma, *dbeta, *coef, *dx, *dres;
    int ldx, ldy, lddy, lddx, lddr, M, C, cpg, rows_per_block, RB, const_stats;
    float slope;
};

__global__ __launch_bounds__(256) void norm_bwd_partial_kernel(NormBwdArgs a) {   // grid (C / 64 chunks, RB row blocks)
    __shared__ float red[4][64][2];
    const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int r0 = blockIdx.y * a.rows_per_block, r1 = min(a.M, r0 + a.rows_per_block);
    float sa = 0.f, sb = 0.f;
    if (c < a.C) {
        const float mean = a.stats[2 * (c / a.cpg)], rstd = a.stats[2 * (c / a.cpg) + 1];
        const bool act = a.slope != 1.0f;
        int m = r0 + ph;
        for (; m + 12 < r1; m += 16) {   // four rows in flight per thread (the loop is a chain of L2 round trips otherwise)
            float g[4], xv[4], yv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                g[u] = a.dy[(size_t)(m + 4 * u) * a.lddy + c];
                xv[u] = a.x[(size_t)(m + 4 * u) * a.ldx + c];
                yv[u] = act ? a.y[(size_t)(m + 4 * u) * a.ldy + c] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {   // same order as the scalar tail: row by row
                if (act && !(yv[u] > 0.f)) g[u] *= a.slope;
                sa += g[u];
                sb = fmaf(g[u], (xv[u] - mean) * rstd, sb);
            }
        }
        for (; m < r1; m += 4) {
            float g = a.dy[(size_t)m * a.lddy + c];
            if (act && !(a.y[(size_t)m * a.ldy + c] > 0.f)) g *= a.slope;
            sa += g;
            sb = fmaf(g, (a.x[(size_t)m * a.ldx + c] - mean) * rstd, sb);
        }
    }
    red[ph][cl][0] = sa;
    red[ph][cl][1] = sb;
    __syncthreads();
    if (ph == 0 && c < a.C) {
        float *o = a.part + ((size_t)blockIdx.y * a.C + c) * 2;
        o[0] = (red[0][cl][0] + red[1][cl][0]) + (red[2][cl][0] + red[3][cl][0]);
        o[1] = (red[0][cl][1] + red[1][cl][1]) + (red[2][cl][1] + red[3][cl][1]);
    }
}

// 64 columns per workgroup, 4 waves: wave w folds the row-block partials w, w + 4, ... of its column (fp64), the four are joined through
// LDS in wave order; cpg (power of two <= 64) adjacent lanes of wave 0 = one group
__global__ __launch_bounds__(256) void norm_bwd_finalize_kernel(NormBwdArgs a) {
    __shared__ double red[4][64][2];
    const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double sa = 0.0, sb = 0.0;
    if (c < a.C)
        for (int b = ph; b < a.RB; b += 4) {
            sa += a.part[((size_t)b * a.C + c) * 2];
            sb += a.part[((size_t)b * a.C + c) * 2 + 1];
        }
    red[ph][cl][0] = sa;
    red[ph][cl][1] = sb;
    __syncthreads();
    if (ph != 0) return;
    sa = (red[0][cl][0] + red[1][cl][0]) + (red[2][cl][0] + red[3][cl][0]);
    sb = (red[0][cl][1] + red[1][cl][1]) + (red[2][cl][1] + red[3][cl][1]);
    if (c < a.C) {
        if (a.dbeta) a.dbeta[c] = (float)sa;
        if (a.dgamma) a.dgamma[c] = (float)sb;
    }
    const double gm = (c < a.C && a.gamma) ? (double)a.gamma[c] : 1.0;
    double s1 = c < a.C ? gm * sa : 0.0, s2 = c < a.C ? gm * sb : 0.0;
    for (int o = 1; o < a.cpg; o <<= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    if (c < a.C && (c & (a.cpg - 1)) == 0) {
        const double n = (double)a.M * a.cpg;
        a.coef[2 * (c / a.cpg)] = (float)(s1 / n);
        a.coef[2 * (c / a.cpg) + 1] = (float)(s2 / n);
    }
}

__global__ void norm_bwd_apply_kernel(NormBwdArgs a) {   // one thread per (row, 4 channels)
    const int C4 = a.C >> 2;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)a.M * C4) return;
    const int m = (int)(t / C4), c0 = 4 * (int)(t - (size_t)m * C4);
    f32x4 g = *reinterpret_cast<const f32x4 *>(a.dy + (size_t)m * a.lddy + c0);
    if (a.slope != 1.0f) {
        const f32x4 yv = *reinterpret_cast<const f32x4 *>(a.y + (size_t)m * a.ldy + c0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (!(yv[e] > 0.f)) g[e] *= a.slope;
    }
    if (a.dres) *reinterpret_cast<f32x4 *>(a.dres + (size_t)m * a.lddr + c0) = g;
    const f32x4 xv = *reinterpret_cast<const f32x4 *>(a.x + (size_t)m * a.ldx + c0);
    f32x4 out;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = c0 + e, G = c / a.cpg;
        const float mean = a.stats[2 * G], rstd = a.stats[2 * G + 1];
        const float gg = g[e] * (a.gamma ? a.gamma[c] : 1.0f);
        out[e] = a.const_stats ? rstd * gg : rstd * (gg - a.coef[2 * G] - (xv[e] - mean) * rstd * a.coef[2 * G + 1]);
    }
    *reinterpret_cast<f32x4 *>(a.dx + (size_t)m * a.lddx + c0) = out;
}

inline bool bad_mat(const void *p, int ld, int cols) { return !p || ld < cols || (ld & 3) || ((uintptr_t)p & 15); }

}  // namespace

extern "C" int cofi_kpconv_aggregate_bwd(const float *dagg, int ldd, const float *q_pts, const float *s_pts, const int32_t *pairs,
                                         const int32_t *offsets, int N, int C, int H, const float *kernel_points, float sigma, float *dfeats,
                                         int ldf, cofi_stream_t stream) {
    if (!dagg || !q_pts || !s_pts || !pairs || !offsets || !kernel_points || !dfeats || N <= 0 || C <= 0 || H <= 0 || ldd < 15 * C || ldf < C ||
        !(sigma > 0.f))
        return COFI_EINVAL;
    if (C < 64 && (C & (C - 1))) return COFI_EUNSUPPORTED;   // narrow layers: a power of two (the network has 32)
    KpBwdArgs a{dagg, q_pts, s_pts, kernel_points, pairs, offsets, dfeats, ldd, ldf, N, C, H, 1.0f / sigma};
    const int CW = C >= 64 ? 64 : C, chunks = (C + CW - 1) / CW;
    const long waves = (long)N * chunks;
    hipLaunchKernelGGL(kpconv_aggregate_bwd_kernel, dim3(cofi_cdiv(waves, 4)), dim3(256), 0, cofi_s(stream), a);
    return cofi_launch_status();
}

extern "C" int cofi_neighbor_maxpool_arg(const float *x, int ldx, int N, int C, const int32_t *idx, int M, int H, float *out, int ldo, uint8_t *arg,
                                         cofi_stream_t stream) {
    if (!idx || !arg || N <= 0 || C <= 0 || (C & 3) || M < 0 || H <= 0 || H > 256 || bad_mat(x, ldx, C) || bad_mat(out, ldo, C)) return COFI_EINVAL;
    if (M == 0) return 0;
    const long waves = (long)M * ((C + 31) / 32);
    hipLaunchKernelGGL(neighbor_maxpool_arg_kernel, dim3(cofi_cdiv(waves, 4)), dim3(256), 0, cofi_s(stream), x, ldx, N, C, idx, M, H, out, ldo, arg);
    return cofi_launch_status();
}

extern "C" int cofi_neighbor_maxpool_bwd(const float *dy, int ldy, const uint8_t *arg, int C, int H, const int32_t *pairs, const int32_t *offsets,
                                         int N, float *dx, int ldx, cofi_stream_t stream) {
    if (!dy || !arg || !pairs || !offsets || !dx || N <= 0 || C <= 0 || H <= 0 || ldy < C || ldx < C) return COFI_EINVAL;
    const long waves = (long)N * ((C + 63) / 64);
    hipLaunchKernelGGL(neighbor_maxpool_bwd_kernel, dim3(cofi_cdiv(waves, 4)), dim3(256), 0, cofi_s(stream), dy, ldy, arg, C, H, pairs, offsets, N, dx,
                       ldx);
    return cofi_launch_status();
}

extern "C" int cofi_gather_rows_bwd(const float *dy, int ldy, int C, const int32_t *pairs, const int32_t *offsets, int N, float *dx, int ldx,
                                    cofi_stream_t stream) {
    if (!dy || !pairs || !offsets || !dx || N <= 0 || C <= 0 || ldy < C || ldx < C) return COFI_EINVAL;
    const long waves = (long)N * ((C + 63) / 64);
    hipLaunchKernelGGL(gather_rows_bwd_kernel, dim3(cofi_cdiv(waves, 4)), dim3(256), 0, cofi_s(stream), dy, ldy, C, pairs, offsets, N, dx, ldx);
    return cofi_launch_status();
}

extern "C" int cofi_im2col_nhwc(const float *x, int ldx, int H, int W, int C, int ks, int stride, int pad, float *col, int ldc,
                                cofi_stream_t stream) {
    if (H <= 0 || W <= 0 || C <= 0 || (C & 3) || ks <= 0 || stride <= 0 || pad < 0 || bad_mat(x, ldx, C) || bad_mat(col, ldc, ks * ks * C))
        return COFI_EINVAL;
    const int Ho = (H + 2 * pad - ks) / stride + 1, Wo = (W + 2 * pad - ks) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return COFI_EINVAL;
    const long n = (long)Ho * Wo * ks * ks * (C / 4);
    hipLaunchKernelGGL(im2col_nhwc_kernel, dim3(cofi_cdiv(n, 256)), dim3(256), 0, cofi_s(stream), x, ldx, H, W, C, ks, stride, pad, Ho, Wo, col, ldc);
    return cofi_launch_status();
}

extern "C" int cofi_col2im_nhwc(const float *dcol, int ldc, int H, int W, int C, int ks, int stride, int pad, float *dx, int ldx,
                                cofi_stream_t stream) {
    if (H <= 0 || W <= 0 || C <= 0 || (C & 3) || ks <= 0 || stride <= 0 || pad < 0 || bad_mat(dx, ldx, C) || bad_mat(dcol, ldc, ks * ks * C))
        return COFI_EINVAL;
    const int Ho = (H + 2 * pad - ks) / stride + 1, Wo = (W + 2 * pad - ks) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return COFI_EINVAL;
    const long n = (long)H * W * (C / 4);
    hipLaunchKernelGGL(col2im_nhwc_kernel, dim3(cofi_cdiv(n, 256)), dim3(256), 0, cofi_s(stream), dcol, ldc, H, W, C, ks, stride, pad, Ho, Wo, dx, ldx);
    return cofi_launch_status();
}


static int norm_bwd_blocks(int M) { return M <= 256 ? 1 : (M / 128 > 256 ? 256 : M / 128); }

extern "C" size_t cofi_group_norm_bwd_workspace(int M, int C, int groups) {
    if (M <= 0 || C <= 0 || groups <= 0) return 0;
    return ((size_t)norm_bwd_blocks(M) * C * 2 + (size_t)groups * 2) * sizeof(float);
}

extern "C" int cofi_group_norm_bwd(const float *x, int ldx, const float *y, int ldy, const float *dy, int lddy, int M, int C, int groups,
                                   const float *stats, const float *gamma, float slope, int const_stats, float *dx, int lddx, float *dgamma,
                                   float *dbeta, float *dres, int lddr, void *ws, size_t ws_bytes, cofi_stream_t stream) {
    if (M <= 0 || C <= 0 || groups <= 0 || (C % groups) || (C & 3) || !stats || bad_mat(x, ldx, C) || bad_mat(dy, lddy, C) || bad_mat(dx, lddx, C) ||
        (slope != 1.0f && bad_mat(y, ldy, C)) || (dres && bad_mat(dres, lddr, C)) || !(slope >= 0.f && slope <= 1.f))
        return COFI_EINVAL;
    const int cpg = C / groups;
    if (cpg > 64 || (cpg & (cpg - 1))) return COFI_EUNSUPPORTED;   // a group = a power of two <= 64 of adjacent columns (the network: 1 ... 64)
    if (!ws || ws_bytes < cofi_group_norm_bwd_workspace(M, C, groups)) return COFI_EWORKSPACE;
    NormBwdArgs a{};
    a.x = x; a.y = y; a.dy = dy; a.stats = stats; a.gamma = gamma; a.dx = dx; a.dres = dres; a.dgamma = dgamma; a.dbeta = dbeta;
    a.ldx = ldx; a.ldy = ldy; a.lddy = lddy; a.lddx = lddx; a.lddr = lddr; a.M = M; a.C = C; a.cpg = cpg; a.slope = slope; a.const_stats = const_stats;
    a.RB = norm_bwd_blocks(M);
    a.rows_per_block = cofi_cdiv(M, a.RB);
    a.part = (float *)ws;
    a.coef = a.part + (size_t)a.RB * C * 2;
    hipLaunchKernelGGL(norm_bwd_partial_kernel, dim3(cofi_cdiv(C, 64), a.RB), dim3(256), 0, cofi_s(stream), a);
    hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3(cofi_cdiv(C, 64)), dim3(256), 0, cofi_s(stream), a);
    hipLaunchKernelGGL(norm_bwd_apply_kernel, dim3(cofi_cdiv((long)M * (C / 4), 256)), dim3(256), 0, cofi_s(stream), a);
    return cofi_launch_status();
}

static int col_sum_blocks(int M) { return M <= 256 ? 1 : (M / 128 > 256 ? 256 : M / 128); }

extern "C" size_t cofi_col_sum_workspace(int M, int C) { return (size_t)col_sum_blocks(M) * C * sizeof(float); }

extern "C" int cofi_col_sum(const float *x, int ldx, int M, int C, float *out, void *ws, size_t ws_bytes, cofi_stream_t stream) {
    if (!x || !out || M <= 0 || C <= 0 || ldx < C) return COFI_EINVAL;
    const int RB = col_sum_blocks(M);
    if (!ws || ws_bytes < cofi_col_sum_workspace(M, C)) return COFI_EWORKSPACE;
    hipLaunchKernelGGL(col_sum_partial_kernel, dim3(cofi_cdiv(C, 64), RB), dim3(256), 0, cofi_s(stream), x, ldx, M, C, cofi_cdiv(M, RB), (float *)ws);
    hipLaunchKernelGGL(col_sum_final_kernel, dim3(cofi_cdiv(C, 64)), dim3(256), 0, cofi_s(stream), (const float *)ws, RB, C, out);
    return cofi_launch_status();
}

extern "C" int cofi_l2norm_rows_bwd(const float *x, int ldx, const float *dy, int lddy, int M, int C, float eps, float *dx, int lddx, cofi_stream_t stream) {
    if (!x || !dy || !dx || M <= 0 || C <= 0 || ldx < C || lddy < C || lddx < C) return COFI_EINVAL;
    hipLaunchKernelGGL(l2norm_rows_bwd_kernel, dim3(cofi_cdiv(M, 4)), dim3(256), 0, cofi_s(stream), x, ldx, dy, lddy, M, C, eps, dx, lddx);
    return cofi_launch_status();
}

extern "C" int cofi_upsample2x_bwd_nhwc(const float *dout, int lddo, int C1, int h, int w, float *dlow, int lddl, cofi_stream_t stream) {
    if (!dout || !dlow || C1 <= 0 || (C1 & 3) || h <= 0 || w <= 0 || lddo < C1 || lddl < C1 || (lddo & 3) || (lddl & 3)) return COFI_EINVAL;
    if ((((uintptr_t)dout) | ((uintptr_t)dlow)) & 15) return COFI_EINVAL;
    const size_t total = (size_t)h * w * (C1 >> 2);
    int nb = (int)((total + 255) / 256);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(upsample2x_bwd_nhwc_kernel, dim3(nb), dim3(256), 0, cofi_s(stream), dout, lddo, C1, h, w, dlow, lddl);
    return cofi_launch_status();
}

extern "C" size_t cofi_col_normalize_workspace(int M, int C) { return cofi_col_sum_workspace(M, C); }

// bwd == 0: y = F.normalize(x, dim=0), stats (2, C) written.  bwd != 0: out = dx for the upstream gradient dy, stats read.
extern "C" int cofi_col_normalize(const float *x, int ldx, const float *dy, int lddy, int M, int C, float eps, int bwd, float *stats, float *out, int ldo,
                                  void *ws, size_t ws_bytes, cofi_stream_t stream) {
    if (!x || !stats || !out || M <= 0 || C <= 0 || ldx < C || ldo < C || (bwd && (!dy || lddy < C))) return COFI_EINVAL;
    const int RB = col_sum_blocks(M);
    if (!ws || ws_bytes < cofi_col_normalize_workspace(M, C)) return COFI_EWORKSPACE;
    const float *a = bwd ? dy : x;
    hipLaunchKernelGGL(col_dot_partial_kernel, dim3(cofi_cdiv(C, 64), RB), dim3(256), 0, cofi_s(stream), a, bwd ? lddy : ldx, x, ldx, M, C, cofi_cdiv(M, RB),
                       (float *)ws);
    const int AB = cofi_cdiv(M, 32) > 128 ? 128 : cofi_cdiv(M, 32);
    if (bwd)
        hipLaunchKernelGGL(col_normalize_apply_kernel<true>, dim3(cofi_cdiv(C, 64), AB), dim3(256), 0, cofi_s(stream), x, ldx, dy, lddy, (const float *)ws, RB,
                           M, C, eps, cofi_cdiv(M, AB), stats, out, ldo);
    else
        hipLaunchKernelGGL(col_normalize_apply_kernel<false>, dim3(cofi_cdiv(C, 64), AB), dim3(256), 0, cofi_s(stream), x, ldx, dy, lddy, (const float *)ws, RB,
                           M, C, eps, cofi_cdiv(M, AB), stats, out, ldo);
    return cofi_launch_status();
}

extern "C" size_t cofi_attention_bwd_workspace(int L, int H) { return (size_t)L * H * 2 * sizeof(float); }

extern "C" int cofi_attention_bwd(const float *q, int ldq, const float *k, int ldk, const float *v, int ldv, const float *o, int ldo,
                                  const float *d_o, int lddo, int L, int S, int H, int D, float scale, float *dq, int lddq, float *dk, int lddk,
                                  float *dv, int lddv, void *ws, size_t ws_bytes, cofi_stream_t stream) {
    if (L <= 0 || S <= 0 || H <= 0) return COFI_EINVAL;
    if (D != AD) return COFI_EUNSUPPORTED;
    const int HD = H * D;
    if (bad_mat(q, ldq, HD) || bad_mat(k, ldk, HD) || bad_mat(v, ldv, HD) || bad_mat(o, ldo, HD) || bad_mat(d_o, lddo, HD) || bad_mat(dq, lddq, HD) ||
        bad_mat(dk, lddk, HD) || bad_mat(dv, lddv, HD))
        return COFI_EINVAL;
    if (!ws || ws_bytes < cofi_attention_bwd_workspace(L, H)) return COFI_EWORKSPACE;
    AttnBwdArgs a{q, k, v, o, d_o, dq, dk, dv, (float *)ws, ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv, L, S, H, scale, scale * 1.4426950408889634f};
    hipLaunchKernelGGL(attention_bwd_dq_kernel, dim3((L + 31) / 32, H), dim3(64 * ABW), 0, cofi_s(stream), a);
    hipLaunchKernelGGL(attention_bwd_dkv_kernel, dim3((S + 31) / 32, H), dim3(64 * ABW), 0, cofi_s(stream), a);
    return cofi_launch_status();
}
