// Normalisation / glue kernels: stack-mode GroupNorm (two-level deterministic statistics),
// row LayerNorm, column inverse norms (token-axis Q normalisation), L2 normalisation, transposes
// and the sine position embedding.  All HBM/L2-bound streaming kernels: float4 accesses where the
// layout allows, one pass over the data per kernel.
#include "common.h"
#include <stdlib.h>
#include "stat_fold.h"

namespace {

// ---------------------------------------------------------------------------- GroupNorm stats
// Level 1: block b reduces rows [b*ROWS, (b+1)*ROWS) for every group -> part[b][g] = {sum, sumsq}
// (fp32 within a thread's <= ROWS*cpg values, fp64 across threads).  Level 2: one block combines
// the partials in fp64 in a fixed order -> {mean, rstd}.  No atomics: bit-reproducible.
constexpr int GS_ROWS = 64;

// ACC = float: the serving path (a thread's <= 16 * cpg values in fp32, everything above in fp64).  ACC = double (cofi_group_stats_exact, the
// training path): the sums are exact to fp64, so the variance does not lose the digits E[x^2] - mean^2 cancels when |mean| >> std - the
// backward of a normalisation removes a large component along the normalised input and amplifies a relative error of rstd by its size.
template <typename ACC>
__global__ __launch_bounds__(256) void group_stats_partial_kernel(const float *x, int ldx, int M, int C, int groups, double *part) {
    // grid: x = row slab of GS_ROWS rows, y = 64-column tile (cpg < 64) or group (cpg >= 64), z = frame (M = rows per frame)
    const int cpg = C / groups;
    x += (size_t)blockIdx.z * M * ldx;
    part += (size_t)blockIdx.z * gridDim.x * groups * 2;
    const int r0 = blockIdx.x * GS_ROWS, r1 = min(M, r0 + GS_ROWS);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __shared__ double red[4][64][2];
    if (cpg < 64) {
        // lanes own columns of this tile; a group is cpg adjacent lanes (cpg is a power of two)
        const int c = blockIdx.y * 64 + lane;
        ACC s = 0, q = 0;
        if (c < C)
            for (int r = r0 + wv; r < r1; r += 4) {
                const ACC v = x[(size_t)r * ldx + c];
                s += v;
                q += v * v;
            }
        double ds = s, dq = q;
        for (int o = 1; o < cpg; o <<= 1) {
            ds += __shfl_xor(ds, o, 64);
            dq += __shfl_xor(dq, o, 64);
        }
        red[wv][lane][0] = ds;
        red[wv][lane][1] = dq;
        __syncthreads();
        if (wv == 0 && c < C && (lane % cpg) == 0) {
            const int g = c / cpg;
            part[((size_t)blockIdx.x * groups + g) * 2 + 0] = red[0][lane][0] + red[1][lane][0] + red[2][lane][0] + red[3][lane][0];
            part[((size_t)blockIdx.x * groups + g) * 2 + 1] = red[0][lane][1] + red[1][lane][1] + red[2][lane][1] + red[3][lane][1];
        }
    } else {
        const int g = blockIdx.y;
        ACC s = 0, q = 0;
        for (int r = r0 + wv; r < r1; r += 4)
            for (int c = g * cpg + lane; c < (g + 1) * cpg; c += 64) {
                const ACC v = x[(size_t)r * ldx + c];
                s += v;
                q += v * v;
            }
        const double ds = wave_sum_d((double)s), dq = wave_sum_d((double)q);
        if (lane == 0) {
            red[wv][0][0] = ds;
            red[wv][0][1] = dq;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            part[((size_t)blockIdx.x * groups + g) * 2 + 0] = red[0][0][0] + red[1][0][0] + red[2][0][0] + red[3][0][0];
            part[((size_t)blockIdx.x * groups + g) * 2 + 1] = red[0][0][1] + red[1][0][1] + red[2][0][1] + red[3][0][1];
        }
    }
}

// one wave per group: lanes stride over the row-slab partials, fp64 butterfly (fixed order)
__global__ __launch_bounds__(256) void group_stats_final_kernel(const double *part, int nblk, int groups, double count, float eps,
                                                                float *stats) {
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= groups) return;
    part += (size_t)blockIdx.y * nblk * groups * 2;   // grid.y = frame
    stats += (size_t)blockIdx.y * groups * 2;
    const int lane = threadIdx.x & 63;
    double s = 0.0, q = 0.0;
    for (int b = lane; b < nblk; b += 64) {
        s += part[((size_t)b * groups + g) * 2 + 0];
        q += part[((size_t)b * groups + g) * 2 + 1];
    }
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    if (lane == 0) {
        const double mean = s / count;
        double var = q / count - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[2 * g + 0] = (float)mean;
        stats[2 * g + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// GroupNorm statistics from the GEMM's fused column partials: one workgroup per group, 256 threads stride over
// the (slab, channel-of-group) pairs (channel fastest: contiguous float2 reads) with two independent fp64
// accumulators, then a fixed-order wave butterfly + 4-wave LDS fold.
__global__ __launch_bounds__(256) void group_stats_from_colpart_kernel(const float *colpart, int nslab, int C, int groups, double count,
                                                                        float eps, float *stats) {
    // grid = (groups, frames): nslab / count are PER FRAME; frame f owns slabs [f*nslab, (f+1)*nslab)
    __shared__ double red[4][2];
    const int g = blockIdx.x;
    const int cpg = C / groups;
    const int total = nslab * cpg;
    const float *base = colpart + ((size_t)blockIdx.y * nslab * C + (size_t)g * cpg) * 2;
    stats += (size_t)blockIdx.y * groups * 2;
    double s0 = 0.0, q0 = 0.0, s1 = 0.0, q1 = 0.0;
    int e = threadIdx.x;
    for (; e + 256 < total; e += 512) {
        const int b0 = e / cpg, c0 = e - b0 * cpg, b1 = (e + 256) / cpg, c1 = (e + 256) - b1 * cpg;
        const float2 t0 = *reinterpret_cast<const float2 *>(base + ((size_t)b0 * C + c0) * 2);
        const float2 t1 = *reinterpret_cast<const float2 *>(base + ((size_t)b1 * C + c1) * 2);
        s0 += (double)t0.x; q0 += (double)t0.y;
        s1 += (double)t1.x; q1 += (double)t1.y;
    }
    if (e < total) {
        const int b0 = e / cpg, c0 = e - b0 * cpg;
        const float2 t0 = *reinterpret_cast<const float2 *>(base + ((size_t)b0 * C + c0) * 2);
        s0 += (double)t0.x; q0 += (double)t0.y;
    }
    const double s = wave_sum_d(s0 + s1), q = wave_sum_d(q0 + q1);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = s; red[threadIdx.x >> 6][1] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double ts = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]), tq = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
        const double mean = ts / count;
        double var = tq / count - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[2 * g + 0] = (float)mean;
        stats[2 * g + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// out[c] = 1 / max(sqrt(sum over slabs of colpart[slab, c].sumsq), eps) for the first C of ncols columns;
// 4 slab phases per column folded through LDS in a fixed order
__global__ __launch_bounds__(256) void col_inv_norm_from_colpart_kernel(const float *colpart, int nslab, int ncols, int C, float eps,
                                                                         float *out) {
    // grid = (column tiles, frames): nslab is per frame
    colpart += (size_t)blockIdx.y * nslab * ncols * 2;
    out += (size_t)blockIdx.y * C;
    __shared__ double red[4][64];
    const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double q = 0.0;
    if (c < C)
        for (int b = ph; b < nslab; b += 4) q += (double)colpart[((size_t)b * ncols + c) * 2 + 1];
    red[ph][cl] = q;
    __syncthreads();
    if (ph == 0 && c < C) out[c] = 1.0f / fmaxf((float)sqrt((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])), eps);
}

struct GnApplyArgs {
    const float *x, *stats, *gamma, *beta, *res, *res_stats, *res_gamma, *res_beta;
    float *y;
    int ldx, ldr, ldy, M, C, cpg;   // M = rows PER FRAME; grid.y = frame
    float slope;
    int groups;
    uint8_t *row_pos;   // optional: row_pos[m] = (sum_c y[m,c] > 0) (kpconv.py:113-114 for the next KPConv); needs C <= 256
};

// Row loop of the apply kernels for one float4 column chunk: a thread's rows are taken four at a time with every load issued
// before the first use (x / res / y may alias as far as the compiler knows: written row by row it would serialise one memory
// round trip per row).
__device__ __forceinline__ void gn_apply_rows(const GnApplyArgs &a, int c, const float (&sc)[4], const float (&sh)[4], const float (&rsc)[4],
                                              const float (&rsh)[4], int tr, int rpb, int tpr, int tc) {
    constexpr int RU = 4;
    const int stride = gridDim.x * rpb;
    for (int m0 = blockIdx.x * rpb + tr; m0 < a.M; m0 += RU * stride) {
        f32x4 xv[RU], rv[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int m = min(m0 + u * stride, a.M - 1);
            xv[u] = *reinterpret_cast<const f32x4 *>(a.x + (size_t)m * a.ldx + c);
            if (a.res) rv[u] = *reinterpret_cast<const f32x4 *>(a.res + (size_t)m * a.ldr + c);
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int m = m0 + u * stride;
            if (m < a.M) {   // uniform over the tpr lanes of a row
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = xv[u][i] * sc[i] + sh[i];
                if (a.res) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += rv[u][i] * rsc[i] + rsh[i];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = v[i] >= 0.f ? v[i] : v[i] * a.slope;
                *reinterpret_cast<float4 *>(a.y + (size_t)m * a.ldy + c) = make_float4(v[0], v[1], v[2], v[3]);
                if (a.row_pos) {   // host guarantees tpr <= 64 and one column pass: the tpr lanes of a row are adjacent lanes of one wave
                    float rs = (v[0] + v[1]) + (v[2] + v[3]);
                    for (int o = 1; o < tpr; o <<= 1) rs += __shfl_xor(rs, o, 64);
                    if (tc == 0) a.row_pos[m] = rs > 0.0f ? 1 : 0;
                }
            }
        }
    }
}

// Thread = one float4 column chunk (its 4 channels' scale/shift are folded once: y = x*sc + sh), looping over a
// strided set of rows: the inner loop is load - 4 fma - select - store with whole rows covered by adjacent lanes.
__global__ __launch_bounds__(256) void group_norm_apply_kernel(GnApplyArgs a) {
    const int c4n = a.C >> 2;
    const int tpr = c4n < 256 ? c4n : 256;           // threads per row (C <= 1024 -> one pass over the columns)
    const int rpb = 256 / tpr;                        // rows per block step
    const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr;
    if (tr >= rpb) return;
    {   // frame blockIdx.y: its rows, its statistics
        const size_t f = blockIdx.y;
        a.x += f * a.M * a.ldx;
        a.y += f * a.M * a.ldy;
        a.stats += f * a.groups * 2;
        if (a.res) a.res += f * a.M * a.ldr;
        if (a.res_stats) a.res_stats += f * a.groups * 2;
        if (a.row_pos) a.row_pos += f * a.M;
    }
    for (int cb = tc; cb < c4n; cb += tpr) {
        const int c = cb * 4;
        float sc[4], sh[4], rsc[4], rsh[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = c + i, g = ch / a.cpg;
            const float mean = a.stats[2 * g], rstd = a.stats[2 * g + 1];
            const float ga = a.gamma ? a.gamma[ch] : 1.f, be = a.gamma ? a.beta[ch] : 0.f;
            sc[i] = rstd * ga;
            sh[i] = be - mean * rstd * ga;
            rsc[i] = 1.f; rsh[i] = 0.f;
            if (a.res && a.res_stats) {
                const float rm = a.res_stats[2 * g], rr = a.res_stats[2 * g + 1];
                const float rg = a.res_gamma ? a.res_gamma[ch] : 1.f, rb = a.res_gamma ? a.res_beta[ch] : 0.f;
                rsc[i] = rr * rg;
                rsh[i] = rb - rm * rr * rg;
            }
        }
        gn_apply_rows(a, c, sc, sh, rsc, rsh, tr, rpb, tpr, tc);
    }
}

// Variant that folds the statistics partials of the producing GEMM itself (stat_fold.h: every workgroup repeats the same
// fixed-order fp64 fold, so all of them see identical statistics) instead of waiting for a separate finalize launch.
// Optionally also emits row_pos[m] = (sum_c y[m,c] > 0), the per-row flag of the next KPConv (kpconv.py:113-114).
struct GnFusedArgs {
    GnApplyArgs a;
    NormSrc n, rn;   // statistics source of x / of the shortcut (rn.part == nullptr: none); gamma / beta live in `a`
};

// Statistics partials -> per-channel scale | shift, once per (frame, 64 table columns): a handful of workgroups, so that the
// hundreds of workgroups of the consuming GEMM / apply launch do not each repeat the fold.
__global__ __launch_bounds__(256) void norm_finalize_kernel(NormSrc n, float *scsh) {
    __shared__ double dred[4 * 256];
    __shared__ float sstat[2 * 64];
    const int f = blockIdx.y;
    const int tc0 = blockIdx.x * 64, ntc = min(64, n.tcols - tc0);   // table columns of this workgroup (tcols: power of two)
    const int epg = n.tcols / n.groups, g0 = tc0 / epg, ng = ntc / epg;
    fold_stat_table<256>(n.part + ((size_t)f * n.nslab * n.tcols + tc0) * 2, n.nslab, ntc, ng, n.count, n.eps, dred, sstat, n.tcols);
    const int cpg = n.C / n.groups;
    float *sc = scsh + (size_t)f * 2 * n.C, *sh = sc + n.C;
    for (int i = threadIdx.x; i < ng * cpg; i += 256) {
        const int gl = i / cpg, c = (g0 + gl) * cpg + (i - gl * cpg);
        const float mean = sstat[2 * gl], rstd = sstat[2 * gl + 1];
        const float ga = n.gamma ? n.gamma[c] : 1.f, be = n.gamma ? n.beta[c] : 0.f;
        sc[c] = rstd * ga;
        sh[c] = be - mean * rstd * ga;
    }
}

__global__ __launch_bounds__(256) void group_norm_apply_fused_kernel(GnFusedArgs fa) {
    GnApplyArgs a = fa.a;
    __shared__ double dred[4 * 256];
    __shared__ float sstat[2 * 1024], rstat[2 * 1024];
    const int c4n = a.C >> 2;
    const int tpr = c4n < 256 ? c4n : 256;
    const int rpb = 256 / tpr;
    const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr;
    {   // frame blockIdx.y: its rows, its statistics partials
        const size_t f = blockIdx.y;
        a.x += f * a.M * a.ldx;
        a.y += f * a.M * a.ldy;
        if (a.res) a.res += f * a.M * a.ldr;
        fa.n.part += f * fa.n.nslab * fa.n.tcols * 2;
        if (fa.rn.part) fa.rn.part += f * fa.rn.nslab * fa.rn.tcols * 2;
        if (a.row_pos) a.row_pos += f * a.M;
    }
    const float *nsc = fa.n.scsh ? fa.n.scsh + (size_t)blockIdx.y * 2 * a.C : nullptr;       // finalized scale | shift of this frame
    const float *rsc_g = fa.rn.scsh ? fa.rn.scsh + (size_t)blockIdx.y * 2 * a.C : nullptr;
    if (!nsc) fold_stat_table<256>(fa.n.part, fa.n.nslab, fa.n.tcols, fa.n.groups, fa.n.count, fa.n.eps, dred, sstat);
    if (fa.rn.part && !rsc_g) fold_stat_table<256>(fa.rn.part, fa.rn.nslab, fa.rn.tcols, fa.rn.groups, fa.rn.count, fa.rn.eps, dred, rstat);
    for (int cb = tc; cb < c4n; cb += tpr) {
        const int c = cb * 4;
        float sc[4], sh[4], rsc[4], rsh[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = c + i, g = ch / a.cpg;
            if (nsc) {
                sc[i] = nsc[ch];
                sh[i] = nsc[a.C + ch];
            } else {
                const float mean = sstat[2 * g], rstd = sstat[2 * g + 1];
                const float ga = a.gamma ? a.gamma[ch] : 1.f, be = a.gamma ? a.beta[ch] : 0.f;
                sc[i] = rstd * ga;
                sh[i] = be - mean * rstd * ga;
            }
            rsc[i] = 1.f; rsh[i] = 0.f;
            if (a.res && fa.rn.part) {
                if (rsc_g) {
                    rsc[i] = rsc_g[ch];
                    rsh[i] = rsc_g[a.C + ch];
                } else {
                    const float rm = rstat[2 * g], rr = rstat[2 * g + 1];
                    const float rg = a.res_gamma ? a.res_gamma[ch] : 1.f, rb = a.res_gamma ? a.res_beta[ch] : 0.f;
                    rsc[i] = rr * rg;
                    rsh[i] = rb - rm * rr * rg;
                }
            }
        }
        gn_apply_rows(a, c, sc, sh, rsc, rsh, tr, rpb, tpr, tc);
    }
}

// ---------------------------------------------------------------------------- LayerNorm
// one wave per row, C <= 2048 held in registers (8 float4 per lane)
__global__ __launch_bounds__(256) void layer_norm_kernel(const float *x, int ldx, int M, int C, const float *gamma,
                                                         const float *beta, float eps, float slope, const float *res, int ldr,
                                                         int res_first, float *y, int ldy) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int lane = threadIdx.x & 63;
    const int c4n = C >> 2;
    float4 v[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c4 = lane + 64 * i;
        if (c4 < c4n) {
            v[i] = reinterpret_cast<const float4 *>(x + (size_t)m * ldx)[c4];
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c4 = lane + 64 * i;
        if (c4 < c4n) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c4 = lane + 64 * i;
        if (c4 < c4n) {
            const float4 gm = reinterpret_cast<const float4 *>(gamma)[c4], bt = reinterpret_cast<const float4 *>(beta)[c4];
            float4 o;
            o.x = (v[i].x - mean) * rstd * gm.x + bt.x;
            o.y = (v[i].y - mean) * rstd * gm.y + bt.y;
            o.z = (v[i].z - mean) * rstd * gm.z + bt.z;
            o.w = (v[i].w - mean) * rstd * gm.w + bt.w;
            float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (res) rv = reinterpret_cast<const float4 *>(res + (size_t)m * ldr)[c4];
            if (res_first) { o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w; }
            if (slope != 1.0f) {   // slope 0: max(o, 0) exactly
                o.x = o.x >= 0.f ? o.x : o.x * slope; o.y = o.y >= 0.f ? o.y : o.y * slope;
                o.z = o.z >= 0.f ? o.z : o.z * slope; o.w = o.w >= 0.f ? o.w : o.w * slope;
            }
            if (!res_first) { o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w; }
            reinterpret_cast<float4 *>(y + (size_t)m * ldy)[c4] = o;
        }
    }
}

// ---------------------------------------------------------------------------- column inverse norm
// out[c] = 1 / max(sqrt(sum_m x[m,c]^2), eps).  Block = 16 columns (one float4 per 4 lanes) x 64 row
// phases; fp32 per thread, fp64 fixed-order LDS fold (deterministic).
__global__ __launch_bounds__(256) void col_inv_norm_kernel(const float *x, int ldx, int M, int C, float eps, float *out) {
    __shared__ double red[64][16];
    const int cq = threadIdx.x & 3, rp = threadIdx.x >> 2;  // 4 lanes x float4 = 16 columns; 64 row phases
    const int c = blockIdx.x * 16 + cq * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) {
        for (int m = rp; m < M; m += 64) {
            const float4 v = *reinterpret_cast<const float4 *>(x + (size_t)m * ldx + c);
            acc.x += v.x * v.x; acc.y += v.y * v.y; acc.z += v.z * v.z; acc.w += v.w * v.w;
        }
    }
    red[rp][cq * 4 + 0] = acc.x; red[rp][cq * 4 + 1] = acc.y; red[rp][cq * 4 + 2] = acc.z; red[rp][cq * 4 + 3] = acc.w;
    __syncthreads();
    if (threadIdx.x < 16 && blockIdx.x * 16 + threadIdx.x < C) {
        double t = 0.0;
        for (int i = 0; i < 64; ++i) t += red[i][threadIdx.x];
        out[blockIdx.x * 16 + threadIdx.x] = 1.0f / fmaxf((float)sqrt(t), eps);
    }
}

// ---------------------------------------------------------------------------- L2 norm of rows
__global__ __launch_bounds__(256) void l2norm_rows2_kernel(const float *x, int ldx, int M, int C, float *y, int ldy, float *y2, int ldy2) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int lane = threadIdx.x & 63;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float v = x[(size_t)m * ldx + c];
        q += v * v;
    }
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(q)), 1e-12f);
    for (int c = lane; c < C; c += 64) {
        const float v = x[(size_t)m * ldx + c] * inv;
        y[(size_t)m * ldy + c] = v;
        y2[(size_t)m * ldy2 + c] = v;
    }
}

__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float *x, int ldx, int M, int C, float *y, int ldy, int transpose) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int lane = threadIdx.x & 63;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float v = x[(size_t)m * ldx + c];
        q += v * v;
    }
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(q)), 1e-12f);
    for (int c = lane; c < C; c += 64) {
        const float v = x[(size_t)m * ldx + c] * inv;
        if (transpose)
            y[(size_t)c * ldy + m] = v;
        else
            y[(size_t)m * ldy + c] = v;
    }
}

// out[f, c] = mean over the M rows of frame f of x[:, c]: nn.AdaptiveAvgPool2d(1) on an NHWC map (imagenet.py:145,215).
// Block = 64 columns x 4 row phases, fp32 per thread, fixed-order fold (deterministic).
__global__ __launch_bounds__(256) void col_mean_kernel(const float *x, int ldx, int M, int C, float *out) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    x += (size_t)blockIdx.y * M * ldx;
    float s = 0.f;
    if (c < C)
        for (int m = ph; m < M; m += 4) s += x[(size_t)m * ldx + c];
    red[ph][cl] = s;
    __syncthreads();
    if (ph == 0 && c < C) out[(size_t)blockIdx.y * C + c] = ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) / (float)M;
}

// 64 x 64 tiles, 16-byte accesses on both sides (M, C, ldx, ldy multiples of 4, 16-byte aligned bases: the backward's activation
// transposes - up to 39 MB - ran at 1.1 TB/s through the scalar kernel below)
__global__ __launch_bounds__(256) void transpose_vec_kernel(const float *x, int ldx, int M, int C, float *y, int ldy) {
    __shared__ float tile[64][65];
    x += (size_t)blockIdx.z * M * ldx;   // frames stacked along the rows of x; frame f writes its own (C, M) block of y
    y += (size_t)blockIdx.z * C * ldy;
    const int m0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int q = threadIdx.x & 15, r = threadIdx.x >> 4;   // 16 float4 per tile row, 16 rows per pass
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + r + 16 * i, c = c0 + 4 * q;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < M && c < C) v = *reinterpret_cast<const f32x4 *>(x + (size_t)m * ldx + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[r + 16 * i][4 * q + e] = v[e];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + r + 16 * i, m = m0 + 4 * q;
        if (c < C && m < M) {
            const f32x4 v = {tile[4 * q][r + 16 * i], tile[4 * q + 1][r + 16 * i], tile[4 * q + 2][r + 16 * i], tile[4 * q + 3][r + 16 * i]};
            *reinterpret_cast<f32x4 *>(y + (size_t)c * ldy + m) = v;
        }
    }
}

// Two matrices with the same row count in ONE launch (the backward of a linear layer transposes dY and X for dW = dY^T X): column blocks
// [0, nb1) belong to the first matrix, the rest to the second; a matrix that does not meet the 16-byte rules takes the scalar path.
struct TransposePairArgs {
    const float *x[2];
    float *y[2];
    int ldx[2], ldy[2], C[2], vec[2], M, nb1;
};

__global__ __launch_bounds__(256) void transpose_pair_kernel(TransposePairArgs a) {
    __shared__ float tile[64][65];
    const int w = blockIdx.x >= a.nb1 ? 1 : 0;
    const float *x = a.x[w];
    float *y = a.y[w];
    const int ldx = a.ldx[w], ldy = a.ldy[w], C = a.C[w], M = a.M;
    const int m0 = blockIdx.y * 64, c0 = (blockIdx.x - (w ? a.nb1 : 0)) * 64;
    if (a.vec[w]) {
        const int q = threadIdx.x & 15, r = threadIdx.x >> 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + r + 16 * i, c = c0 + 4 * q;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < M && c < C) v = *reinterpret_cast<const f32x4 *>(x + (size_t)m * ldx + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[r + 16 * i][4 * q + e] = v[e];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = c0 + r + 16 * i, m = m0 + 4 * q;
            if (c < C && m < M) {
                const f32x4 v = {tile[4 * q][r + 16 * i], tile[4 * q + 1][r + 16 * i], tile[4 * q + 2][r + 16 * i], tile[4 * q + 3][r + 16 * i]};
                *reinterpret_cast<f32x4 *>(y + (size_t)c * ldy + m) = v;
            }
        }
    } else {
        const int cl = threadIdx.x & 63, rr = threadIdx.x >> 6;
        for (int i = rr; i < 64; i += 4) {
            const int m = m0 + i, c = c0 + cl;
            tile[i][cl] = (m < M && c < C) ? x[(size_t)m * ldx + c] : 0.f;
        }
        __syncthreads();
        for (int i = rr; i < 64; i += 4) {
            const int c = c0 + i, m = m0 + cl;
            if (c < C && m < M) y[(size_t)c * ldy + m] = tile[cl][i];
        }
    }
}

__global__ void transpose_kernel(const float *x, int ldx, int M, int C, float *y, int ldy) {
    __shared__ float tile[32][33];
    x += (size_t)blockIdx.z * M * ldx;
    y += (size_t)blockIdx.z * C * ldy;
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int m = by + i, c = bx + tx;
        tile[i][tx] = (m < M && c < C) ? x[(size_t)m * ldx + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = bx + i, m = by + tx;
        if (c < C && m < M) y[(size_t)c * ldy + m] = tile[tx][i];
    }
}

// ---------------------------------------------------------------------------- sine position embedding
struct PosArgs {
    const void *coords;
    float *out;
    int coords_are_int, T, n_dim, F, d_model, accumulate, ldo;
    float dim_t[64];
};

// position_encoding.py:39-50: ang = (x * 2pi) / dim_t[i]; even i -> sin, odd i -> cos; precise
// sinf/cosf (arguments reach ~500 rad for metre-scale coordinates).
__global__ void pos_sine_kernel(PosArgs a) {
    const size_t total = (size_t)a.T * a.d_model;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(e / a.d_model), o = (int)(e % a.d_model);
        float v = 0.f;
        if (o < a.n_dim * a.F) {
            const int d = o / a.F, i = o % a.F;
            const float x = a.coords_are_int ? (float)reinterpret_cast<const int32_t *>(a.coords)[(size_t)t * a.n_dim + d]
                                             : reinterpret_cast<const float *>(a.coords)[(size_t)t * a.n_dim + d];
            const float ang = (x * 6.283185307179586f) / a.dim_t[i];
            v = (i & 1) ? cosf(ang) : sinf(ang);
        }
        float *dst = a.out + (size_t)t * a.ldo + o;
        *dst = a.accumulate ? (*dst + v) : v;
    }
}

}  // namespace

extern "C" int cofi_group_stats_from_colpart(const float *colpart, int nslab, int M, int C, int groups, float eps, float *stats,
                                             int frames, cofi_stream_t stream) {
    if (!colpart || !stats || nslab <= 0 || M <= 0 || C <= 0 || groups <= 0 || (C % groups) || frames <= 0) return COFI_EINVAL;
    if ((nslab % frames) || (M % frames)) return COFI_EINVAL;
    if (frames > 1 && ((M / frames) % 64)) return COFI_EINVAL;   // the producer's 64-row slabs must not straddle frames
    if (nslab / frames != cofi_cdiv(M / frames, 64)) return COFI_EINVAL;
    hipLaunchKernelGGL(group_stats_from_colpart_kernel, dim3(groups, frames), dim3(256), 0, cofi_s(stream), colpart, nslab / frames, C,
                       groups, (double)(M / frames) * (C / groups), eps, stats);
    return cofi_launch_status();
}

extern "C" int cofi_col_inv_norm_from_colpart(const float *colpart, int nslab, int M, int ncols, int C, float eps, float *out, int frames,
                                              cofi_stream_t stream) {
    if (!colpart || !out || nslab <= 0 || M <= 0 || C <= 0 || ncols < C || frames <= 0 || (nslab % frames) || (M % frames)) return COFI_EINVAL;
    if (frames > 1 && ((M / frames) % 64)) return COFI_EINVAL;   // the producer's 64-row slabs must not straddle frames
    if (nslab / frames != cofi_cdiv(M / frames, 64)) return COFI_EINVAL;
    hipLaunchKernelGGL(col_inv_norm_from_colpart_kernel, dim3(cofi_cdiv(C, 64), frames), dim3(256), 0, cofi_s(stream), colpart,
                       nslab / frames, ncols, C, eps, out);
    return cofi_launch_status();
}

extern "C" size_t cofi_group_stats_workspace(int M, int C, int groups, int frames) {
    if (M <= 0 || groups <= 0 || frames <= 0 || (M % frames)) return 0;
    return (size_t)cofi_cdiv(M / frames, GS_ROWS) * frames * groups * 2 * sizeof(double);
}

static int group_stats_entry(const float *x, int ldx, int M, int C, int groups, float eps, float *stats, void *ws, size_t ws_bytes, int frames,
                             bool exact, cofi_stream_t stream) {
    if (!x || !stats || M <= 0 || C <= 0 || groups <= 0 || (C % groups) || ldx < C || frames <= 0 || (M % frames)) return COFI_EINVAL;
    const int cpg = C / groups;
    if (cpg < 64 && (cpg & (cpg - 1))) return COFI_EUNSUPPORTED;  // power-of-two group width below one wave
    if (!ws || ws_bytes < cofi_group_stats_workspace(M, C, groups, frames)) return COFI_EWORKSPACE;
    const int Mf = M / frames, nblk = cofi_cdiv(Mf, GS_ROWS);
    hipStream_t s = cofi_s(stream);
    const dim3 grid(nblk, cpg < 64 ? cofi_cdiv(C, 64) : groups, frames);
    if (exact)
        hipLaunchKernelGGL(group_stats_partial_kernel<double>, grid, dim3(256), 0, s, x, ldx, Mf, C, groups, (double *)ws);
    else
        hipLaunchKernelGGL(group_stats_partial_kernel<float>, grid, dim3(256), 0, s, x, ldx, Mf, C, groups, (double *)ws);
    hipLaunchKernelGGL(group_stats_final_kernel, dim3(cofi_cdiv(groups, 4), frames), dim3(256), 0, s, (const double *)ws, nblk, groups,
                       (double)Mf * cpg, eps, stats);
    return cofi_launch_status();
}

extern "C" int cofi_group_stats(const float *x, int ldx, int M, int C, int groups, float eps, float *stats, void *ws,
                                size_t ws_bytes, int frames, cofi_stream_t stream) {
    return group_stats_entry(x, ldx, M, C, groups, eps, stats, ws, ws_bytes, frames, false, stream);
}

extern "C" int cofi_group_stats_exact(const float *x, int ldx, int M, int C, int groups, float eps, float *stats, void *ws,
                                      size_t ws_bytes, int frames, cofi_stream_t stream) {
    return group_stats_entry(x, ldx, M, C, groups, eps, stats, ws, ws_bytes, frames, true, stream);
}

extern "C" int cofi_group_norm_apply(const float *x, int ldx, int M, int C, int groups, const float *stats, const float *gamma,
                                     const float *beta, const float *res, int ldr, const float *res_stats,
                                     const float *res_gamma, const float *res_beta, float slope, float *y, int ldy, uint8_t *row_pos,
                                     int frames, cofi_stream_t stream) {
    if (!x || !stats || !y || M <= 0 || C <= 0 || groups <= 0 || (C % groups) || (C & 3) || (ldx & 3) || (ldy & 3)) return COFI_EINVAL;
    if ((gamma == nullptr) != (beta == nullptr) || (res_gamma == nullptr) != (res_beta == nullptr)) return COFI_EINVAL;
    if (res && (ldr & 3)) return COFI_EINVAL;
    if (frames <= 0 || (M % frames)) return COFI_EINVAL;
    const int Mf = M / frames;
    GnApplyArgs a{x, stats, gamma, beta, res, res_stats, res_gamma, res_beta, y, ldx, ldr, ldy, Mf, C, C / groups, slope, groups, row_pos};
    const int c4n = C >> 2, tpr = c4n < 256 ? c4n : 256, rpb = 256 / tpr;
    if (row_pos && (tpr > 64 || (64 % tpr))) return COFI_EUNSUPPORTED;
    int nb = cofi_cdiv(Mf, rpb * 4);  // ~4 rows per thread
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(group_norm_apply_kernel, dim3(nb, frames), dim3(256), 0, cofi_s(stream), a);
    return cofi_launch_status();
}

extern "C" int cofi_norm_finalize(const cofi_norm_desc_t *norm, int rows, int frames, float *scale_shift, cofi_stream_t stream) {
    if (!norm || !scale_shift || rows <= 0 || frames <= 0 || (rows % frames) || ((uintptr_t)scale_shift & 15)) return COFI_EINVAL;
    NormSrc n;
    if (int rc = make_norm_src(norm, rows / frames, frames, 1 << 20, &n)) return rc;
    const int epg = n.tcols / n.groups;
    if (epg > 64) return COFI_EUNSUPPORTED;   // a group's table columns stay inside one workgroup's 64
    hipLaunchKernelGGL(norm_finalize_kernel, dim3(cofi_cdiv(n.tcols, 64), frames), dim3(256), 0, cofi_s(stream), n, scale_shift);
    return cofi_launch_status();
}

extern "C" int cofi_group_norm_apply_partials(const float *x, int ldx, int M, int C, const cofi_norm_desc_t *norm, const float *res, int ldr,
                                              const cofi_norm_desc_t *res_norm, float *y, int ldy, uint8_t *row_pos, int frames,
                                              cofi_stream_t stream) {
    if (!x || !norm || !y || M <= 0 || C <= 0 || (C & 3) || (ldx & 3) || (ldy & 3) || ldx < C || ldy < C) return COFI_EINVAL;
    if (res && ((ldr & 3) || ldr < C)) return COFI_EINVAL;
    if (frames <= 0 || (M % frames) || (res_norm && !res)) return COFI_EINVAL;
    if (norm->channels != C || (res_norm && res_norm->channels != C) || (res_norm && res_norm->groups != norm->groups)) return COFI_EINVAL;
    const int Mf = M / frames, c4n = C >> 2, tpr = c4n < 256 ? c4n : 256, rpb = 256 / tpr;
    if (row_pos && (tpr > 64 || (64 % tpr))) return COFI_EUNSUPPORTED;
    GnFusedArgs fa{};
    if (int rc = make_norm_src(norm, Mf, frames, 1 << 20, &fa.n)) return rc;
    if (res_norm)
        if (int rc = make_norm_src(res_norm, Mf, frames, 1 << 20, &fa.rn)) return rc;
    if (fa.n.groups > 1024) return COFI_EUNSUPPORTED;   // LDS statistics table
    fa.a = GnApplyArgs{x, nullptr, norm->gamma, norm->beta, res, nullptr, res_norm ? res_norm->gamma : nullptr, res_norm ? res_norm->beta : nullptr,
                       y, ldx, ldr, ldy, Mf, C, C / norm->groups, norm->slope, norm->groups, row_pos};
    int nb = cofi_cdiv(Mf, rpb * 4);
    constexpr int cap = 2048;   // workgroups per launch, grid-strided beyond (round-5 sweep of this cap: bandwidth-bound at every value)
    const int cap_f = cap / frames > 0 ? cap / frames : 1;
    if (nb > cap_f) nb = cap_f;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(group_norm_apply_fused_kernel, dim3(nb, frames), dim3(256), 0, cofi_s(stream), fa);
    return cofi_launch_status();
}

extern "C" int cofi_layer_norm(const float *x, int ldx, int M, int C, const float *gamma, const float *beta, float eps, int relu,
                               const float *res, int ldr, float *y, int ldy, cofi_stream_t stream) {
    if (!x || !y || !gamma || !beta || M <= 0 || C <= 0 || C > 2048 || (C & 3) || (ldx & 3) || (ldy & 3) || (res && (ldr & 3)))
        return COFI_EINVAL;
    hipLaunchKernelGGL(layer_norm_kernel, dim3(cofi_cdiv(M, 4)), dim3(256), 0, cofi_s(stream), x, ldx, M, C, gamma, beta, eps, relu ? 0.0f : 1.0f,
                       res, ldr, 0, y, ldy);
    return cofi_launch_status();
}

extern "C" int cofi_layer_norm_act(const float *x, int ldx, int M, int C, const float *gamma, const float *beta, float eps, float slope, const float *res,
                                   int ldr, int res_first, float *y, int ldy, cofi_stream_t stream) {
    if (!x || !y || !gamma || !beta || M <= 0 || C <= 0 || C > 2048 || (C & 3) || (ldx & 3) || (ldy & 3) || (res && (ldr & 3)) || !(slope >= 0.f && slope <= 1.f))
        return COFI_EINVAL;
    hipLaunchKernelGGL(layer_norm_kernel, dim3(cofi_cdiv(M, 4)), dim3(256), 0, cofi_s(stream), x, ldx, M, C, gamma, beta, eps, slope, res, ldr,
                       res_first ? 1 : 0, y, ldy);
    return cofi_launch_status();
}

extern "C" int cofi_col_inv_norm(const float *x, int ldx, int M, int C, float eps, float *out, cofi_stream_t stream) {
    if (!x || !out || M <= 0 || C <= 0 || ldx < C || (C & 3) || (ldx & 3) || ((uintptr_t)x & 15)) return COFI_EINVAL;
    hipLaunchKernelGGL(col_inv_norm_kernel, dim3(cofi_cdiv(C, 16)), dim3(256), 0, cofi_s(stream), x, ldx, M, C, eps, out);
    return cofi_launch_status();
}

extern "C" int cofi_l2norm_rows(const float *x, int ldx, int M, int C, float *y, int ldy, int transpose, cofi_stream_t stream) {
    if (!x || !y || M <= 0 || C <= 0 || ldx < C || (transpose ? ldy < M : ldy < C)) return COFI_EINVAL;
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3(cofi_cdiv(M, 4)), dim3(256), 0, cofi_s(stream), x, ldx, M, C, y, ldy, transpose);
    return cofi_launch_status();
}

extern "C" int cofi_l2norm_rows2(const float *x, int ldx, int M, int C, float *y, int ldy, float *y2, int ldy2, cofi_stream_t stream) {
    if (!x || !y || !y2 || M <= 0 || C <= 0 || ldx < C || ldy < C || ldy2 < C) return COFI_EINVAL;
    hipLaunchKernelGGL(l2norm_rows2_kernel, dim3(cofi_cdiv(M, 4)), dim3(256), 0, cofi_s(stream), x, ldx, M, C, y, ldy, y2, ldy2);
    return cofi_launch_status();
}

extern "C" int cofi_col_mean(const float *x, int ldx, int M, int C, float *out, int frames, cofi_stream_t stream) {
    if (!x || !out || M <= 0 || C <= 0 || ldx < C || frames <= 0 || (M % frames)) return COFI_EINVAL;
    hipLaunchKernelGGL(col_mean_kernel, dim3(cofi_cdiv(C, 64), frames), dim3(256), 0, cofi_s(stream), x, ldx, M / frames, C, out);
    return cofi_launch_status();
}

extern "C" int cofi_transpose(const float *x, int ldx, int M, int C, float *y, int ldy, int frames, cofi_stream_t stream) {
    if (!x || !y || M <= 0 || C <= 0 || ldx < C || ldy < M || frames <= 0) return COFI_EINVAL;
    if (!((M | C | ldx | ldy) & 3) && !(((uintptr_t)x | (uintptr_t)y) & 15) && (long)M * C >= 64 * 64)
        hipLaunchKernelGGL(transpose_vec_kernel, dim3(cofi_cdiv(C, 64), cofi_cdiv(M, 64), frames), dim3(256), 0, cofi_s(stream), x, ldx, M, C, y, ldy);
    else
        hipLaunchKernelGGL(transpose_kernel, dim3(cofi_cdiv(C, 32), cofi_cdiv(M, 32), frames), dim3(256), 0, cofi_s(stream), x, ldx, M, C, y, ldy);
    return cofi_launch_status();
}

extern "C" int cofi_transpose_pair(const float *x1, int ldx1, int C1, float *y1, int ldy1, const float *x2, int ldx2, int C2, float *y2, int ldy2, int M,
                                   cofi_stream_t stream) {
    if (!x1 || !y1 || !x2 || !y2 || M <= 0 || C1 <= 0 || C2 <= 0 || ldx1 < C1 || ldx2 < C2 || ldy1 < M || ldy2 < M) return COFI_EINVAL;
    TransposePairArgs a;
    a.x[0] = x1; a.x[1] = x2; a.y[0] = y1; a.y[1] = y2;
    a.ldx[0] = ldx1; a.ldx[1] = ldx2; a.ldy[0] = ldy1; a.ldy[1] = ldy2; a.C[0] = C1; a.C[1] = C2; a.M = M;
    a.vec[0] = !((M | C1 | ldx1 | ldy1) & 3) && !(((uintptr_t)x1 | (uintptr_t)y1) & 15);
    a.vec[1] = !((M | C2 | ldx2 | ldy2) & 3) && !(((uintptr_t)x2 | (uintptr_t)y2) & 15);
    a.nb1 = cofi_cdiv(C1, 64);
    hipLaunchKernelGGL(transpose_pair_kernel, dim3(a.nb1 + cofi_cdiv(C2, 64), cofi_cdiv(M, 64)), dim3(256), 0, cofi_s(stream), a);
    return cofi_launch_status();
}

extern "C" int cofi_pos_sine(const void *coords, int coords_are_int, int T, int n_dim, const float *dim_t_host, int F, int d_model,
                             int accumulate, float *out, int ldo, cofi_stream_t stream) {
    if (!coords || !dim_t_host || !out || T <= 0 || n_dim <= 0 || F <= 0 || F > 64 || n_dim * F > d_model || ldo < d_model)
        return COFI_EINVAL;
    PosArgs a;
    a.coords = coords; a.out = out; a.coords_are_int = coords_are_int; a.T = T; a.n_dim = n_dim; a.F = F; a.d_model = d_model;
    a.accumulate = accumulate; a.ldo = ldo;
    for (int i = 0; i < 64; ++i) a.dim_t[i] = i < F ? dim_t_host[i] : 1.0f;
    size_t total = (size_t)T * d_model;
    int nb = (int)((total + 255) / 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(pos_sine_kernel, dim3(nb), dim3(256), 0, cofi_s(stream), a);
    return cofi_launch_status();
}
