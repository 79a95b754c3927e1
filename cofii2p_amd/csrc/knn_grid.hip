// Exact k-nearest neighbours (k <= 128) over a uniform 2-D cell grid of the support points: the same results, bit for bit, as
// the brute-force kernel of knn.hip / oracle/knn_oracle.c — canonical fp32 distance, ascending (distance, lowest index) —
// but a query only visits the cells that can hold one of its k nearest.  Reference: model/kpconv/preprocess_data.py:109-143.
//
// Build (cofi_knn_grid_build): bounding box -> the two axes with the largest extent span a grid of <= 128 x 128 square cells
// (about 12 points per cell) -> counting sort of the support into cell order as (x, y, z, original index) records.
//
// Search (cofi_knn_topk_grid), one wave per query: stream the records of a block of cells around the query through the same
// threshold filter / 64-key staging / wave-bitonic merge as the brute-force kernel (best 128 keys sorted across the wave), then
//   * while fewer than k points were seen: double the block;
//   * once the k-th distance d_k is known: grow the block ONCE to the cells within sqrt(d_k) of the query and stop when every
//     unvisited cell is provably farther than d_k.
// "Provably" covers the rounding of the canonical distance d = ((-2 q.s) + |q|^2) + |s|^2, which deviates from the true squared
// distance D by at most 8u(|q|^2 + |s|^2) + uD (u = 2^-24; three roundings in each dot/norm, two in the final sums): a cell is
// skipped only if (border distance - delta)^2 (1 - 2^-20) - 2^-20 (|q|^2 + max|s|^2 + d_k) > d_k, i.e. with twice that margin;
// delta covers the rounding of the cell binning.  Skipped points therefore have keys strictly above the k-th key, and the keys
// of visited points are computed by the same instructions as in the brute-force kernel: identical output.
#include "knn_common.h"
#include <stdlib.h>

namespace {

constexpr int GRID_MAX_DIM = 128;
constexpr int GRID_MAX_CELLS = GRID_MAX_DIM * GRID_MAX_DIM;

struct GridHeader {       // 64 bytes at the start of the workspace
    unsigned lo[3], hi[3];   // bounding box as order-preserving uints (atomicMin / atomicMax)
    unsigned ss_max;         // max |s|^2 (non-negative floats order like their bits)
    int S;
    float target_occ;        // points per cell the cell size aims at
    float first_block;       // the first block of a query holds about first_block * k points
    int pad[6];
};
static_assert(sizeof(GridHeader) == 64, "header layout");

struct GridParams {
    float x0, z0, h, inv_h, delta, ss_max;
    int nx, nz, a0, a1;
};

__host__ __device__ inline size_t grid_off_start() { return sizeof(GridHeader); }
__host__ __device__ inline size_t grid_off_cursor() { return grid_off_start() + sizeof(int) * (GRID_MAX_CELLS + 4); }
__host__ __device__ inline size_t grid_off_sorted() { return grid_off_cursor() + sizeof(int) * (GRID_MAX_CELLS + 4); }
static_assert((sizeof(GridHeader) + 2 * sizeof(int) * (GRID_MAX_CELLS + 4)) % 16 == 0, "records are 16-byte aligned");

__device__ __forceinline__ unsigned f2ord(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

// Every thread derives the same grid from the header (pure function of the bounding box and S).
__device__ __forceinline__ GridParams grid_params(const GridHeader *hd) {
    GridParams g;
    float lo[3], ext[3];
    float cmax = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = ord2f(hd->lo[a]);
        const float hi = ord2f(hd->hi[a]);
        ext[a] = hi - lo[a];
        cmax = fmaxf(cmax, fmaxf(fabsf(lo[a]), fabsf(hi)));
    }
    // the two widest axes span the grid (the third is left unbinned: a cell is a column)
    int drop = 0;
    if (ext[1] < ext[drop]) drop = 1;
    if (ext[2] < ext[drop]) drop = 2;
    g.a0 = drop == 0 ? 1 : 0;
    g.a1 = drop == 2 ? 1 : 2;
    const float e0 = fmaxf(ext[g.a0], 1e-6f), e1 = fmaxf(ext[g.a1], 1e-6f);
    float h = sqrtf(e0 * e1 * hd->target_occ / (float)max(hd->S, 1));
    h = fmaxf(h, fmaxf(e0, e1) / (float)(GRID_MAX_DIM - 1));   // at most GRID_MAX_DIM cells per axis
    h = fmaxf(h, 1e-6f);
    g.h = h;
    g.inv_h = 1.0f / h;
    g.x0 = lo[g.a0];
    g.z0 = lo[g.a1];
    g.nx = min(GRID_MAX_DIM, (int)(e0 * g.inv_h) + 1);
    g.nz = min(GRID_MAX_DIM, (int)(e1 * g.inv_h) + 1);
    g.delta = 7.62939453125e-6f * cmax + 1e-7f;   // 2^-17 of the largest coordinate: >> the rounding of (p - x0) * inv_h vs x0 + i*h
    g.ss_max = __uint_as_float(hd->ss_max);
    return g;
}

__device__ __forceinline__ int grid_cell(const GridParams &g, float a, float b) {
    const int cu = min(max((int)floorf((a - g.x0) * g.inv_h), 0), g.nx - 1);
    const int cv = min(max((int)floorf((b - g.z0) * g.inv_h), 0), g.nz - 1);
    return cv * g.nx + cu;
}
__device__ __forceinline__ float pick(float x, float y, float z, int a) { return a == 0 ? x : (a == 1 ? y : z); }

// ------------------------------------------------------------------------------------------------ build
__global__ void grid_init_kernel(GridHeader *hd, int *counts, int S, float target_occ, float first_block) {
    for (int i = threadIdx.x; i < GRID_MAX_CELLS + 4; i += blockDim.x) counts[i] = 0;
    if (threadIdx.x < 3) {
        hd->lo[threadIdx.x] = 0xffffffffu;
        hd->hi[threadIdx.x] = 0u;
    }
    if (threadIdx.x == 3) {
        hd->ss_max = 0u;
        hd->S = S;
        hd->target_occ = target_occ;
        hd->first_block = first_block;
    }
}

__global__ __launch_bounds__(256) void grid_bbox_kernel(const float *support, int S, GridHeader *hd) {
    unsigned lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u}, ssm = 0u;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < S; i += gridDim.x * blockDim.x) {
        const float x = support[3 * (size_t)i], y = support[3 * (size_t)i + 1], z = support[3 * (size_t)i + 2];
        const unsigned ox = f2ord(x), oy = f2ord(y), oz = f2ord(z);
        lo[0] = min(lo[0], ox); hi[0] = max(hi[0], ox);
        lo[1] = min(lo[1], oy); hi[1] = max(hi[1], oy);
        lo[2] = min(lo[2], oz); hi[2] = max(hi[2], oz);
        ssm = max(ssm, __float_as_uint(canon_sqnorm(x, y, z)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo[a] = min(lo[a], (unsigned)__shfl_xor((int)lo[a], o, 64));
            hi[a] = max(hi[a], (unsigned)__shfl_xor((int)hi[a], o, 64));
        }
        ssm = max(ssm, (unsigned)__shfl_xor((int)ssm, o, 64));
    }
    __shared__ unsigned red[4][7];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            red[wave][a] = lo[a];
            red[wave][3 + a] = hi[a];
        }
        red[wave][6] = ssm;
    }
    __syncthreads();
    if (threadIdx.x < 7) {   // 7 atomics per workgroup
        const int c = threadIdx.x;
        unsigned v = red[0][c];
        for (int w = 1; w < 4; ++w) v = c < 3 ? min(v, red[w][c]) : max(v, red[w][c]);
        if (c < 3) atomicMin(&hd->lo[c], v);
        else if (c < 6) atomicMax(&hd->hi[c - 3], v);
        else atomicMax(&hd->ss_max, v);
    }
}

__global__ __launch_bounds__(256) void grid_count_kernel(const float *support, int S, const GridHeader *hd, int *counts) {
    const GridParams g = grid_params(hd);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < S; i += gridDim.x * blockDim.x) {
        const float x = support[3 * (size_t)i], y = support[3 * (size_t)i + 1], z = support[3 * (size_t)i + 2];
        atomicAdd(&counts[grid_cell(g, pick(x, y, z, g.a0), pick(x, y, z, g.a1))], 1);
    }
}

// exclusive scan of the grid's cell counts in place (-> cell_start[0 .. ncells], cell_start[ncells] = S) + a copy as scatter cursors
__global__ __launch_bounds__(1024) void grid_scan_kernel(const GridHeader *hd, int *start, int *cursor) {
    __shared__ int wsum[16];
    const GridParams g = grid_params(hd);
    const int n = g.nx * g.nz + 1;                 // the entry behind the last cell holds a zero count: it becomes S
    const int per = (n + 1023) / 1024;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int b = t * per, e = min(b + per, n);
    int sum = 0;
    for (int i = b; i < e; ++i) sum += start[i];
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int run = inc - sum;
    for (int w = 0; w < wave; ++w) run += wsum[w];
    for (int i = b; i < e; ++i) {
        const int c = start[i];
        start[i] = run;
        cursor[i] = run;
        run += c;
    }
}

__global__ __launch_bounds__(256) void grid_scatter_kernel(const float *support, int S, const GridHeader *hd, int *cursor, float4 *sorted) {
    const GridParams g = grid_params(hd);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < S; i += gridDim.x * blockDim.x) {
        const float x = support[3 * (size_t)i], y = support[3 * (size_t)i + 1], z = support[3 * (size_t)i + 2];
        const int pos = atomicAdd(&cursor[grid_cell(g, pick(x, y, z, g.a0), pick(x, y, z, g.a1))], 1);
        sorted[pos] = make_float4(x, y, z, __int_as_float(i));   // order inside a cell is arbitrary: the result does not depend on it
    }
}

// ------------------------------------------------------------------------------------------------ search
constexpr int GW = 4;   // waves (= queries) per workgroup

__global__ __launch_bounds__(64 * GW) void knn_grid_kernel(const GridHeader *__restrict__ hd, const int *__restrict__ cell_start,
                                                           const float4 *__restrict__ sorted, const float *__restrict__ query,
                                                           const int32_t *__restrict__ qorder, int Q, int k, int S,
                                                           int32_t *__restrict__ out_idx, float *__restrict__ out_dist) {
    __shared__ u64 stage[GW][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w = blockIdx.x * GW + wave;
    if (w >= Q) return;   // no workgroup barrier below
    const int q = __builtin_amdgcn_readfirstlane(qorder ? qorder[w] : w);
    const GridParams g = grid_params(hd);
    const float qx = query[3 * (size_t)q], qy = query[3 * (size_t)q + 1], qz = query[3 * (size_t)q + 2];
    const float qq = canon_sqnorm(qx, qy, qz);
    const float qa = pick(qx, qy, qz, g.a0), qb = pick(qx, qy, qz, g.a1);
    const int nx = g.nx, nz = g.nz;

    Best128 best;   // sorted best-128 of the query, admission threshold best.tau = 128th key
    int nstage = 0;
    auto flush = [&]() {
        best.merge(lane < nstage ? stage[wave][lane] : KEY_INF, lane);
        nstage = 0;
    };
    // the records [s, e) of the sorted array, 64 at a time
    auto stream = [&](int s, int e) {
        for (int c0 = s; c0 < e; c0 += 64) {
            const int c = c0 + lane;
            const bool in = c < e;
            const float4 sp = sorted[in ? c : s];
            const float d = canon_dist(qx, qy, qz, qq, sp.x, sp.y, sp.z, canon_sqnorm(sp.x, sp.y, sp.z));
            const u64 key = ((u64)__float_as_uint(d) << 32) | (unsigned)__float_as_int(sp.w);
            const bool pass = in && key < best.tau;
            const u64 mask = __ballot(pass);
            if (mask == 0) continue;
            const int n = __popcll(mask);
            if (nstage + n > 64) flush();
            if (pass) stage[wave][nstage + __popcll(mask & ((1ull << lane) - 1ull))] = key;
            nstage += n;
        }
    };

    // block of cells [ulo, uhi] x [vlo, vhi]; (pulo..pvhi) = the part already visited
    const int cu = __builtin_amdgcn_readfirstlane(min(max((int)floorf((qa - g.x0) * g.inv_h), 0), nx - 1));
    const int cv = __builtin_amdgcn_readfirstlane(min(max((int)floorf((qb - g.z0) * g.inv_h), 0), nz - 1));
    const float occ = (float)S / (float)(nx * nz);
    int r = max(1, (int)ceilf(0.5f * (sqrtf(hd->first_block * (float)k / fmaxf(occ, 1e-3f)) - 1.0f)));
    int ulo = max(cu - r, 0), uhi = min(cu + r, nx - 1), vlo = max(cv - r, 0), vhi = min(cv + r, nz - 1);
    int pulo = 1, puhi = 0, pvlo = 1, pvhi = 0;   // empty
    for (;;) {
        // visit block \ previous block, row by row; the run bounds of 64 rows are fetched by the lanes in one go
        for (int vb = vlo; vb <= vhi; vb += 64) {
            const int v = vb + lane;
            int sA = 0, eA = 0, sB = 0, eB = 0;
            if (v <= vhi) {
                const bool fresh = v < pvlo || v > pvhi;
                const int a_hi = fresh ? uhi : pulo - 1;
                if (a_hi >= ulo) {
                    sA = cell_start[v * nx + ulo];
                    eA = cell_start[v * nx + a_hi + 1];
                }
                if (!fresh && uhi > puhi) {
                    sB = cell_start[v * nx + puhi + 1];
                    eB = cell_start[v * nx + uhi + 1];
                }
            }
            const int nrows = min(64, vhi - vb + 1);
            for (int j = 0; j < nrows; ++j) {
                const int s0 = __builtin_amdgcn_readlane(sA, j), e0 = __builtin_amdgcn_readlane(eA, j);
                const int s1 = __builtin_amdgcn_readlane(sB, j), e1 = __builtin_amdgcn_readlane(eB, j);
                stream(s0, e0);
                stream(s1, e1);
            }
        }
        if (nstage > 0) flush();
        const u64 kth = best.kth(k);
        const float kd = __uint_as_float((unsigned)(kth >> 32));   // +inf while fewer than k points were seen
        // distance from the query to the nearest border behind which unvisited cells exist
        float b = INFINITY;
        if (ulo > 0) b = fminf(b, qa - (g.x0 + (float)ulo * g.h));
        if (uhi < nx - 1) b = fminf(b, (g.x0 + (float)(uhi + 1) * g.h) - qa);
        if (vlo > 0) b = fminf(b, qb - (g.z0 + (float)vlo * g.h));
        if (vhi < nz - 1) b = fminf(b, (g.z0 + (float)(vhi + 1) * g.h) - qb);
        if (b == INFINITY) break;   // the whole grid was visited
        const float eps = 9.5367431640625e-7f * (qq + g.ss_max + kd);   // 2^-20 (...)
        const float bm = fmaxf(b - g.delta, 0.0f);
        if (bm * bm * 0.99999904632568359375f - eps > kd) break;          // (1 - 2^-20): every unvisited point is farther than the k-th
        pulo = ulo; puhi = uhi; pvlo = vlo; pvhi = vhi;
        if (kd == INFINITY) {
            r = 2 * r + 1;   // not enough points yet
            ulo = max(cu - r, 0); uhi = min(cu + r, nx - 1); vlo = max(cv - r, 0); vhi = min(cv + r, nz - 1);
        } else {
            // d_k only shrinks from here on: covering the disc of radius sqrt(d_k) (+ margins) ends the search
            const float R = sqrtf(kd + eps) * 1.000002f + 2.0f * g.delta;
            ulo = min(ulo, max((int)floorf((qa - R - g.x0) * g.inv_h), 0));
            uhi = max(uhi, min((int)floorf((qa + R - g.x0) * g.inv_h), nx - 1));
            vlo = min(vlo, max((int)floorf((qb - R - g.z0) * g.inv_h), 0));
            vhi = max(vhi, min((int)floorf((qb + R - g.z0) * g.inv_h), nz - 1));
            if (ulo == pulo && uhi == puhi && vlo == pvlo && vhi == pvhi) {   // margins only: widen by one ring so the loop advances
                ulo = max(ulo - 1, 0); uhi = min(uhi + 1, nx - 1); vlo = max(vlo - 1, 0); vhi = min(vhi + 1, nz - 1);
            }
        }
        ulo = __builtin_amdgcn_readfirstlane(ulo); uhi = __builtin_amdgcn_readfirstlane(uhi);
        vlo = __builtin_amdgcn_readfirstlane(vlo); vhi = __builtin_amdgcn_readfirstlane(vhi);
    }
    best.emit(q, k, S, lane, out_idx, out_dist);
}

// the cell-sorted order of the support points themselves (for a self search: neighbouring waves work on neighbouring queries)
__global__ void grid_order_kernel(const float4 *sorted, int S, int32_t *order) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < S; i += gridDim.x * blockDim.x) order[i] = __float_as_int(sorted[i].w);
}

}  // namespace

extern "C" size_t cofi_knn_grid_workspace(int S) {
    if (S <= 0) return 0;
    return grid_off_sorted() + sizeof(float4) * (size_t)S;
}

extern "C" int cofi_knn_grid_build(const float *support, int S, void *ws, size_t ws_bytes, int32_t *order_out, cofi_stream_t stream) {
    if (!support || !ws || S <= 0 || ((uintptr_t)ws & 15)) return COFI_EINVAL;
    if (ws_bytes < cofi_knn_grid_workspace(S)) return COFI_EWORKSPACE;
    hipStream_t s = cofi_s(stream);
    char *base = (char *)ws;
    GridHeader *hd = (GridHeader *)base;
    int *start = (int *)(base + grid_off_start()), *cursor = (int *)(base + grid_off_cursor());
    float4 *sorted = (float4 *)(base + grid_off_sorted());
    const int nb = min(cofi_cdiv(S, 256), 1024), nb_box = min(cofi_cdiv(S, 1024), 64);
    // tuned on MI355X (results never depend on them): points per cell, size of a query's first block in units of k
    const float target_occ = 12.0f, first_block = 0.8f;
    hipLaunchKernelGGL(grid_init_kernel, dim3(1), dim3(1024), 0, s, hd, start, S, target_occ, first_block);
    hipLaunchKernelGGL(grid_bbox_kernel, dim3(nb_box), dim3(256), 0, s, support, S, hd);
    hipLaunchKernelGGL(grid_count_kernel, dim3(nb), dim3(256), 0, s, support, S, hd, start);
    hipLaunchKernelGGL(grid_scan_kernel, dim3(1), dim3(1024), 0, s, (const GridHeader *)hd, start, cursor);
    hipLaunchKernelGGL(grid_scatter_kernel, dim3(nb), dim3(256), 0, s, support, S, hd, cursor, sorted);
    if (order_out) hipLaunchKernelGGL(grid_order_kernel, dim3(nb), dim3(256), 0, s, sorted, S, order_out);
    return cofi_launch_status();
}

extern "C" int cofi_knn_topk_grid(const void *ws, size_t ws_bytes, int S, const float *query, const int32_t *qorder, int Q, int k,
                                  int32_t *out_idx, float *out_dist, cofi_stream_t stream) {
    if (!ws || !query || !out_idx || S <= 0 || Q < 0 || k <= 0 || k > 128 || ((uintptr_t)ws & 15)) return COFI_EINVAL;
    if (ws_bytes < cofi_knn_grid_workspace(S)) return COFI_EWORKSPACE;
    if (Q == 0) return 0;
    const char *base = (const char *)ws;
    hipLaunchKernelGGL(knn_grid_kernel, dim3(cofi_cdiv(Q, GW)), dim3(64 * GW), 0, cofi_s(stream), (const GridHeader *)base,
                       (const int *)(base + grid_off_start()), (const float4 *)(base + grid_off_sorted()), query, qorder, Q, k, S,
                       out_idx, out_dist);
    return cofi_launch_status();
}
