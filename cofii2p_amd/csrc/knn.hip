// Brute-force k-nearest neighbours with a wavefront-level top-k (k <= 128), bit-exact against
// oracle/knn_oracle.c.  Reference: model/kpconv/preprocess_data.py:109-143 (`square_distance` +
// `dist.topk(k, largest=False)`), model/network.py:250-264 (`point2node`).
//
// Canonical fp32 arithmetic (file compiled with -ffp-contract=off; the only fused operations are the
// two explicit fmaf):
//     dot = fmaf(qz, sz, fmaf(qy, sy, qx*sx));  qq = (qx*qx + qy*qy) + qz*qz;  ss likewise
//     d   = max(((-2*dot) + qq) + ss, 1e-12f)
// Order: ascending 64-bit key (float bits of d) << 32 | index  ==  (distance, lowest index first).
//
// Workgroup = 16 waves = 16 queries.  Candidates are staged once per workgroup through LDS as
// (x,y,z,|s|^2) float4 tiles of 1024 points (coalesced 12-B rows -> one ds_read_b128 per lane and
// step), so L2 is read once per 16 queries.  Each wave streams the tile 64 candidates at a time and
// keeps its query's best 128 keys sorted across the wave (2 keys per lane).  A candidate enters a
// 64-entry LDS staging buffer only if it beats the current 128th key (ballot + mbcnt compaction);
// a full buffer is bitonic-sorted across the wave and bitonic-merged into the list.  After the first
// few hundred candidates almost nothing passes the threshold, so the stream runs at ~10 VALU
// instructions per 64 distances.
#include "knn_common.h"

namespace {

constexpr int TILE = 1024;

__global__ __launch_bounds__(1024) void knn_topk_kernel(const float *support, int S, const float *query, int Q, int k,
                                                        int32_t *out_idx, float *out_dist) {
    __shared__ float4 tile[TILE];
    __shared__ u64 stage[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 16 + wave;
    const bool active = q < Q;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (active) {
        qx = query[3 * (size_t)q];
        qy = query[3 * (size_t)q + 1];
        qz = query[3 * (size_t)q + 2];
    }
    const float qq = canon_sqnorm(qx, qy, qz);
    Best128 best;        // sorted best-128 of the query, admission threshold best.tau
    int nstage = 0;      // wave-uniform fill of stage[wave]

    auto flush = [&]() {
        best.merge(lane < nstage ? stage[wave][lane] : KEY_INF, lane);
        nstage = 0;
    };

    for (int t0 = 0; t0 < S; t0 += TILE) {
        __syncthreads();
        {
            const int c = t0 + threadIdx.x;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < S) {
                v.x = support[3 * (size_t)c];
                v.y = support[3 * (size_t)c + 1];
                v.z = support[3 * (size_t)c + 2];
                v.w = canon_sqnorm(v.x, v.y, v.z);
            }
            tile[threadIdx.x] = v;
        }
        __syncthreads();
        if (!active) continue;
        const int nt = min(TILE, S - t0);
        for (int i = 0; i < nt; i += 64) {
            const int c = i + lane;
            const float4 sp = tile[c];
            const float d = canon_dist(qx, qy, qz, qq, sp.x, sp.y, sp.z, sp.w);
            const u64 key = ((u64)__float_as_uint(d) << 32) | (unsigned)(t0 + c);
            const bool pass = (c < nt) && key < best.tau;
            const u64 mask = __ballot(pass);
            if (mask == 0) continue;
            const int n = __popcll(mask);
            if (nstage + n > 64) flush();  // wave-uniform branch
            // a flush lowers tau, but every staged/passing key is still a valid candidate: the list merge
            // keeps only the 128 smallest, so over-admission is harmless
            if (pass) {
                const int pos = nstage + __popcll(mask & ((1ull << lane) - 1ull));
                stage[wave][pos] = key;
            }
            nstage += n;
        }
    }
    if (!active) return;
    if (nstage > 0) flush();
    best.emit(q, k, S, lane, out_idx, out_dist);
}

// k = 1: nearest support row per query, lowest index on ties.  One wave per query; if sel != NULL
// the query rows are points[sel[i]] and the row count is read from count_dev on the device.
__global__ __launch_bounds__(256) void nearest_kernel(const float *nodes, int S, const float *points, const int32_t *sel,
                                                      const int32_t *count_dev, int Q, int32_t *out_idx) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nq = count_dev ? min(*count_dev, Q) : Q;
    if (q >= nq) return;
    const size_t row = sel ? (size_t)sel[q] : (size_t)q;
    const float qx = points[3 * row], qy = points[3 * row + 1], qz = points[3 * row + 2];
    const float qq = canon_sqnorm(qx, qy, qz);
    u64 best = KEY_INF;
    // 4 candidates per lane and round, all 12 loads issued before the first distance: the scan is bound by the L2 round trip
    // per round, not by arithmetic.  The (distance, index) key makes the result independent of the visiting order.
    for (int c0 = lane; c0 < S; c0 += 256) {
        float px[4], py[4], pz[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = min(c0 + 64 * u, S - 1);
            px[u] = nodes[3 * (size_t)c];
            py[u] = nodes[3 * (size_t)c + 1];
            pz[u] = nodes[3 * (size_t)c + 2];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + 64 * u;
            float4 sp;
            sp.x = px[u]; sp.y = py[u]; sp.z = pz[u];
            sp.w = canon_sqnorm(sp.x, sp.y, sp.z);
            const float d = canon_dist(qx, qy, qz, qq, sp.x, sp.y, sp.z, sp.w);
            if (c < S) best = umin64(best, ((u64)__float_as_uint(d) << 32) | (unsigned)c);
        }
    }
    best = umin64(best, lane_xor64<32>(best, lane));
    best = umin64(best, lane_xor64<16>(best, lane));
    best = umin64(best, lane_xor64<8>(best, lane));
    best = umin64(best, lane_xor64<4>(best, lane));
    best = umin64(best, lane_xor64<2>(best, lane));
    best = umin64(best, lane_xor64<1>(best, lane));
    if (lane == 0) out_idx[q] = (int)(unsigned)(best & 0xffffffffu);
}


// ---- nearest stage-(i+1) point of every stage-i point WITHOUT a search (column 0 of upsampling[i], the only column the forward reads:
// model/kpconv/functional.py:20).  Stage i+1 is a selection WITH replacement of stage i (point j = stage-i point sub[j], bit for bit), so
// the nearest selected point of p is the first entry of p's own sorted neighbour row neighbors[i][p] that was selected - exactly, ties
// included: the candidates at the minimal canonical distance are walked and the lowest stage-(i+1) copy index among them wins, the
// (distance, lowest index) order of the search this replaces.  first_copy[q] = lowest j with sub[j] == q (INT_MAX: not selected).
// A row none of whose k entries was selected (degenerate clouds only) scans all stage-(i+1) points instead.
__global__ void first_copy_kernel(const int32_t *sub, int S1, int32_t *first_copy) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < S1) atomicMin(&first_copy[sub[j]], j);   // integer min: order-free, deterministic
}

__global__ void up_nearest_kernel(const float *pts, int N, const int32_t *nbr, int k, const int32_t *first_copy, const int32_t *sub, int S1,
                                  int32_t *out, int ldo) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    const float qx = pts[3 * p], qy = pts[3 * p + 1], qz = pts[3 * p + 2];
    const float qq = canon_sqnorm(qx, qy, qz);
    u64 best = KEY_INF;
    for (int h = 0; h < k; ++h) {
        const int q = nbr[(size_t)p * k + h];
        if (q < 0 || q >= N) break;   // shadow entries close a row
        const float sx = pts[3 * q], sy = pts[3 * q + 1], sz = pts[3 * q + 2];
        const float d = canon_dist(qx, qy, qz, qq, sx, sy, sz, canon_sqnorm(sx, sy, sz));
        if (best != KEY_INF && __float_as_uint(d) > (unsigned)(best >> 32)) break;   // ascending row: nothing closer or equal follows
        const int j = first_copy[q];
        if (j != 0x7fffffff) best = umin64(best, ((u64)__float_as_uint(d) << 32) | (unsigned)j);
    }
    if (best == KEY_INF) {
        for (int j = 0; j < S1; ++j) {
            const int q = sub[j];
            const float sx = pts[3 * q], sy = pts[3 * q + 1], sz = pts[3 * q + 2];
            const float d = canon_dist(qx, qy, qz, qq, sx, sy, sz, canon_sqnorm(sx, sy, sz));
            best = umin64(best, ((u64)__float_as_uint(d) << 32) | (unsigned)j);
        }
    }
    out[(size_t)p * ldo] = (int)(unsigned)(best & 0xffffffffu);
}

__global__ void idx64_to_32_kernel(const int64_t *src, int32_t *dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (int32_t)src[i];
}
__global__ void idx32_to_64_kernel(const int32_t *src, int64_t *dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (int64_t)src[i];
}

}  // namespace

extern "C" int cofi_knn_topk(const float *support, int S, const float *query, int Q, int k, int32_t *out_idx, float *out_dist,
                             cofi_stream_t stream) {
    if (!support || !query || !out_idx || S <= 0 || Q < 0 || k <= 0 || k > 128) return COFI_EINVAL;
    if (Q == 0) return 0;
    hipLaunchKernelGGL(knn_topk_kernel, dim3(cofi_cdiv(Q, 16)), dim3(1024), 0, cofi_s(stream), support, S, query, Q, k, out_idx,
                       out_dist);
    return cofi_launch_status();
}

extern "C" int cofi_nearest_node(const float *nodes, int S, const float *points, int Q, int32_t *out_idx, cofi_stream_t stream) {
    if (!nodes || !points || !out_idx || S <= 0 || Q < 0) return COFI_EINVAL;
    if (Q == 0) return 0;
    hipLaunchKernelGGL(nearest_kernel, dim3(cofi_cdiv(Q, 4)), dim3(256), 0, cofi_s(stream), nodes, S, points,
                       (const int32_t *)nullptr, (const int32_t *)nullptr, Q, out_idx);
    return cofi_launch_status();
}

extern "C" int cofi_nearest_node_sel(const float *nodes, int S, const float *points_all, const int32_t *sel,
                                     const int32_t *count_dev, int max_count, int32_t *out_idx, cofi_stream_t stream) {
    if (!nodes || !points_all || !sel || !count_dev || !out_idx || S <= 0 || max_count < 0) return COFI_EINVAL;
    if (max_count == 0) return 0;
    hipLaunchKernelGGL(nearest_kernel, dim3(cofi_cdiv(max_count, 4)), dim3(256), 0, cofi_s(stream), nodes, S, points_all, sel,
                       count_dev, max_count, out_idx);
    return cofi_launch_status();
}

extern "C" int cofi_knn_up_nearest(const float *points, int N, const int32_t *neighbors, int k, const int32_t *sub, int S1, int32_t *first_copy,
                                   int32_t *out_idx, int ldo, cofi_stream_t stream) {
    if (!points || !neighbors || !sub || !first_copy || !out_idx || N <= 0 || k <= 0 || S1 <= 0 || ldo < 1) return COFI_EINVAL;
    if (hipError_t e = hipMemsetD32Async((hipDeviceptr_t)first_copy, 0x7fffffff, (size_t)N, cofi_s(stream)); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(first_copy_kernel, dim3(cofi_cdiv(S1, 256)), dim3(256), 0, cofi_s(stream), sub, S1, first_copy);
    hipLaunchKernelGGL(up_nearest_kernel, dim3(cofi_cdiv(N, 256)), dim3(256), 0, cofi_s(stream), points, N, neighbors, k, first_copy, sub, S1, out_idx,
                       ldo);
    return cofi_launch_status();
}

extern "C" int cofi_idx64_to_idx32(const int64_t *src, int32_t *dst, size_t n, cofi_stream_t stream) {
    if (!src || !dst) return COFI_EINVAL;
    if (n == 0) return 0;
    int nb = (int)((n + 255) / 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(idx64_to_32_kernel, dim3(nb), dim3(256), 0, cofi_s(stream), src, dst, n);
    return cofi_launch_status();
}

extern "C" int cofi_idx32_to_idx64(const int32_t *src, int64_t *dst, size_t n, cofi_stream_t stream) {
    if (!src || !dst) return COFI_EINVAL;
    if (n == 0) return 0;
    int nb = (int)((n + 255) / 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(idx32_to_64_kernel, dim3(nb), dim3(256), 0, cofi_s(stream), src, dst, n);
    return cofi_launch_status();
}
