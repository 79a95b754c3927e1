// KPConv part 1 (kernel-point influence + neighbour aggregation), neighbour max-pool and
// nearest up-sample gathers.  Reference: model/kpconv/kpconv.py:91-116, functional.py:5-21,53-66.
//
// kpconv_aggregate: ONE WAVE PER QUERY POINT.  The aggregation of one query,
//     agg[k, c] = sum_h w[h,k] * f[idx[h], c]      (15 x H) . (H x C),
// is an MFMA chain of v_mfma_f32_16x16x4_f32: rows = kernel points (15 padded to 16),
// cols = 16 channels, contraction over 4 neighbours per step.  Lane (g = l>>4, j = l&15) owns
// neighbour h = 4s+g in step s and
//   - computes the influence weight of kernel point k = j for that neighbour (A operand), and
//   - loads channel c0+j of that neighbour's feature row (B operand): 16 lanes read 64 contiguous
//     bytes of one row, so the gather is served in coalesced 64-B segments from L2.
// Nothing is staged through LDS and the (M,H,15) influence tensor / (M,H,C) gather of the reference
// never exist in memory.  Channel passes of 16*NACC channels go over gridDim.y.
#include "common.h"

namespace {

struct KpArgs {
    const float *feats, *q_pts, *s_pts, *kp;
    const int32_t *idx;
    const uint8_t *row_pos;
    float *agg, *cnt;
    int ldf, N, C, M, H, ld_agg;
    float sigma;
};

template <int NACC>
__global__ __launch_bounds__(256) void kpconv_aggregate_kernel(KpArgs a) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= a.M) return;
    const int j = lane & 15, g = lane >> 4;
    const int c0 = blockIdx.y * (16 * NACC);
    const float qx = a.q_pts[3 * m], qy = a.q_pts[3 * m + 1], qz = a.q_pts[3 * m + 2];
    const bool kvalid = j < 15;
    const float kx = kvalid ? a.kp[3 * j] : 0.f, ky = kvalid ? a.kp[3 * j + 1] : 0.f, kz = kvalid ? a.kp[3 * j + 2] : 0.f;
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int npos = 0;
    const int32_t *irow = a.idx + (size_t)m * a.H;
    const int steps = a.H >> 2;
    for (int s = 0; s < steps; ++s) {
        const int id = irow[4 * s + g];
        const bool valid = (unsigned)id < (unsigned)a.N;
        float w = 0.f;
        float f[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) f[i] = 0.f;
        if (valid) {
            const float *sp = a.s_pts + 3 * (size_t)id;
            // kpconv.py:93-99: ((s - q) - kp)^2 summed, sqrt, 1 - d/sigma, clamp at 0
            const float dx = (sp[0] - qx) - kx, dy = (sp[1] - qy) - ky, dz = (sp[2] - qz) - kz;
            const float sq = (dx * dx + dy * dy) + dz * dz;
            w = kvalid ? fmaxf(1.0f - sqrtf(sq) / a.sigma, 0.0f) : 0.0f;
            const float *fr = a.feats + (size_t)id * a.ldf + c0 + j;
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                if (c0 + 16 * i + j < a.C) f[i] = fr[16 * i];
            if (j == 0 && blockIdx.y == 0) npos += a.row_pos[id];
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, f[i], acc[i], 0, 0, 0);
    }
    // D layout 16x16: row (kernel point) = 4*g + r, col (channel) = j
    float *orow = a.agg + (size_t)m * a.ld_agg;
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        const int c = c0 + 16 * i + j;
        if (c < a.C) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 4 * g + r;
                if (k < 15) orow[(size_t)k * a.C + c] = acc[i][r];
            }
        }
    }
    if (blockIdx.y == 0) {
        npos += __shfl_xor(npos, 16, 64);
        npos += __shfl_xor(npos, 32, 64);
        if (lane == 0) a.cnt[m] = (float)(npos > 1 ? npos : 1);
    }
}

// row_pos[n] = (sum_c feats[n,c] > 0); one wave per row (kpconv.py:113-114 applied per source row)
__global__ void row_sum_positive_kernel(const float *feats, int ld, int N, int C, uint8_t *row_pos) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += feats[(size_t)n * ld + c];
    s = wave_sum(s);
    if (lane == 0) row_pos[n] = s > 0.0f ? 1 : 0;
}

// out[m,c] = max_h x[idx[m,h], c], zero row behind idx == N.  Block: 64 channels x 4 row groups;
// consecutive lanes read consecutive channels (256 B per neighbour row).
__global__ __launch_bounds__(256) void neighbor_maxpool_kernel(const float *x, int ldx, int N, int C, const int32_t *idx, int M,
                                                               int H, float *out, int ldo) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int m = blockIdx.x * 4 + wv;
    const int c = blockIdx.y * 64 + lane;
    if (m >= M) return;
    const int32_t *irow = idx + (size_t)m * H;
    float best = -INFINITY;
    for (int h0 = 0; h0 < H; h0 += 64) {
        const int myid = (h0 + lane < H) ? irow[h0 + lane] : -1;
        const int cnt = min(64, H - h0);
        for (int h = 0; h < cnt; ++h) {
            const int id = __shfl(myid, h, 64);
            float v = 0.0f;
            if ((unsigned)id < (unsigned)N && c < C) v = x[(size_t)id * ldx + c];
            best = fmaxf(best, v);
        }
    }
    if (c < C) out[(size_t)m * ldo + c] = best;
}

// out[m,:] = x[idx[m*idx_stride], :] (zero row for idx == N)
__global__ void gather_rows_kernel(const float *x, int ldx, int N, int C, const int32_t *idx, int idx_stride, int M, float *out,
                                   int ldo) {
    const int m = blockIdx.x;
    const int id = idx[(size_t)m * idx_stride];
    const bool valid = (unsigned)id < (unsigned)N;
    const int c4 = C >> 2;
    if (((ldx | ldo | C) & 3) == 0) {
        for (int c = threadIdx.x; c < c4; c += blockDim.x) {
            float4 v = valid ? reinterpret_cast<const float4 *>(x + (size_t)id * ldx)[c] : make_float4(0, 0, 0, 0);
            reinterpret_cast<float4 *>(out + (size_t)m * ldo)[c] = v;
        }
    } else {
        for (int c = threadIdx.x; c < C; c += blockDim.x) out[(size_t)m * ldo + c] = valid ? x[(size_t)id * ldx + c] : 0.0f;
    }
}

}  // namespace

extern "C" int cofi_row_sum_positive(const float *feats, int ld, int N, int C, uint8_t *row_pos, cofi_stream_t stream) {
    if (!feats || !row_pos || N < 0 || C <= 0 || ld < C) return COFI_EINVAL;
    if (N == 0) return 0;
    hipLaunchKernelGGL(row_sum_positive_kernel, dim3(cofi_cdiv(N, 4)), dim3(256), 0, cofi_s(stream), feats, ld, N, C, row_pos);
    return cofi_launch_status();
}

extern "C" int cofi_kpconv_aggregate(const float *feats, int ldf, int N, int C, const float *q_pts, const float *s_pts,
                                     const int32_t *idx, int M, int H, const float *kernel_points, float sigma,
                                     const uint8_t *row_pos, float *agg, int ld_agg, float *cnt, cofi_stream_t stream) {
    if (!feats || !q_pts || !s_pts || !idx || !kernel_points || !row_pos || !agg || !cnt) return COFI_EINVAL;
    if (N <= 0 || C <= 0 || M < 0 || H <= 0 || (H & 3) || ldf < C || ld_agg < 15 * C || !(sigma > 0.f)) return COFI_EINVAL;
    if (M == 0) return 0;
    KpArgs a{feats, q_pts, s_pts, kernel_points, idx, row_pos, agg, cnt, ldf, N, C, M, H, ld_agg, sigma};
    hipStream_t s = cofi_s(stream);
    const int mb = cofi_cdiv(M, 4);
    if (C <= 16)
        hipLaunchKernelGGL((kpconv_aggregate_kernel<1>), dim3(mb, 1), dim3(256), 0, s, a);
    else if (C <= 32)
        hipLaunchKernelGGL((kpconv_aggregate_kernel<2>), dim3(mb, 1), dim3(256), 0, s, a);
    else if (C <= 64)
        hipLaunchKernelGGL((kpconv_aggregate_kernel<4>), dim3(mb, 1), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((kpconv_aggregate_kernel<8>), dim3(mb, cofi_cdiv(C, 128)), dim3(256), 0, s, a);
    return cofi_launch_status();
}

extern "C" int cofi_neighbor_maxpool(const float *x, int ldx, int N, int C, const int32_t *idx, int M, int H, float *out, int ldo,
                                     cofi_stream_t stream) {
    if (!x || !idx || !out || N <= 0 || C <= 0 || M < 0 || H <= 0 || ldx < C || ldo < C) return COFI_EINVAL;
    if (M == 0) return 0;
    hipLaunchKernelGGL(neighbor_maxpool_kernel, dim3(cofi_cdiv(M, 4), cofi_cdiv(C, 64)), dim3(256), 0, cofi_s(stream), x, ldx, N,
                       C, idx, M, H, out, ldo);
    return cofi_launch_status();
}

extern "C" int cofi_gather_rows(const float *x, int ldx, int N, int C, const int32_t *idx, int idx_stride, int M, float *out,
                                int ldo, cofi_stream_t stream) {
    if (!x || !idx || !out || N <= 0 || C <= 0 || M < 0 || idx_stride <= 0 || ldx < C || ldo < C) return COFI_EINVAL;
    if (M == 0) return 0;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(M), dim3(C >= 1024 ? 256 : (C >= 256 ? 128 : 64)), 0, cofi_s(stream), x, ldx, N, C,
                       idx, idx_stride, M, out, ldo);
    return cofi_launch_status();
}
