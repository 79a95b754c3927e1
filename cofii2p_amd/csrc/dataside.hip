// Data-side preprocessing of one frame on the device (reference: data/kitti.py:259-393, the part of
// kitti_pc_img_dataset.__getitem__ between the disk read and the model call):
//   cofi_voxel_downsample   data/kitti.py:145-166  open3d voxel_down_sample(0.1) of points + intensity ("colors") + normals
//   cofi_gather_transform   data/kitti.py:168-180, 284-288, 293  resample to exactly num_pc rows, random SE(3), feats = [intensity | R n]
//   cofi_resize_crop_image  data/kitti.py:306-322, 375  cv2.resize(INTER_LINEAR) x0.5, crop, / 255, HWC -> CHW
// The KNN pyramid (preprocess_data.py:36-107) is knn_grid.hip; the coarse / fine label projection works on 1280 points and stays
// host-side numpy, exactly as the reference writes it (cofii2p_amd/dataside.py).
//
// Voxel grid: every point gets the key of its voxel (open3d: index = floor((p - (min_bound - voxel/2)) / voxel) per axis), the
// (key, point index) pairs go through a STABLE least-significant-digit radix sort (8-bit digits, 5 passes over the 39-bit key), and
// one thread per voxel averages its run in sorted = original order with fp64 accumulators (open3d accumulates in double):
// bit-reproducible, no float atomics, no hash table.  Output order = ascending key (open3d's is the iteration order of an
// unordered_map: unspecified, parity with it is unpinned).
// A sort pass = three launches.  The input is cut into SORT_TILES contiguous tiles, ONE WAVE per tile:
//   histogram  digit counts per tile -> table [digit][tile]
//   scan       one wave per digit: exclusive scan over the tiles + the digit's total
//   scatter    every workgroup scans the 256 digit totals itself, a wave then walks its tile 64 keys at a time IN ORDER: lanes holding
//              the same digit find each other with 8 ballots, take consecutive slots after the digit's running offset (an LDS counter
//              private to the wave) - stable by construction, no atomics.
// These are latency / HBM-bound integer and streaming kernels on ~120 000 points: no MFMA anywhere.
#include "common.h"

namespace {

constexpr int KEY_BITS = 13;      // voxel index bits per axis (8192 voxels of 0.1 m = 819 m)
constexpr int SORT_PASSES = 5;    // 8-bit digits over 3 * KEY_BITS = 39 bits
constexpr int SORT_TILES = 512;   // one wave each; 4 waves per workgroup
constexpr int SORT_WPB = 4;
constexpr int RED_T = 1024;       // threads of the reduction / head-scan workgroups

struct VoxHeader {
    double minb[3];  // min_bound - voxel / 2 (open3d keeps it in double)
    float imax;      // max intensity
    int nvox;
    int overflow;    // a voxel index did not fit KEY_BITS
};

// bounding box minimum + intensity maximum of one RED_T-row chunk -> part[block] = {min x, min y, min z, max intensity}
__global__ __launch_bounds__(RED_T) void vox_bounds_kernel(const float *pts, int ldp, const float *inten, int ldi, int N, float *part) {
    __shared__ float red[4][RED_T / 64];
    const int i = blockIdx.x * RED_T + threadIdx.x;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx = -INFINITY;
    if (i < N) {
#pragma unroll
        for (int a = 0; a < 3; ++a) mn[a] = pts[(size_t)i * ldp + a];
        mx = inten[(size_t)i * ldi];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int a = 0; a < 3; ++a) mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64));
        mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = mn[0]; red[1][w] = mn[1]; red[2][w] = mn[2]; red[3][w] = mx; }
    __syncthreads();
    if (threadIdx.x < 4) {
        float v = red[threadIdx.x][0];
        for (int k = 1; k < RED_T / 64; ++k) v = threadIdx.x < 3 ? fminf(v, red[threadIdx.x][k]) : fmaxf(v, red[threadIdx.x][k]);
        part[4 * blockIdx.x + threadIdx.x] = v;
    }
}

// every workgroup folds the chunk partials itself (min / max: order-free), then keys its 256 points; block 0 publishes the header
__global__ __launch_bounds__(256) void vox_keys_kernel(const float *pts, int ldp, int N, double voxel, const float *part, int nchunk, VoxHeader *hdr,
                                                      unsigned long long *keys, int *idx) {
    __shared__ float red[4][4];
    __shared__ double s_minb[3];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx = -INFINITY;
    for (int c = threadIdx.x; c < nchunk; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(part + 4 * c);
        mn[0] = fminf(mn[0], v[0]); mn[1] = fminf(mn[1], v[1]); mn[2] = fminf(mn[2], v[2]); mx = fmaxf(mx, v[3]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int a = 0; a < 3; ++a) mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64));
        mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = mn[0]; red[1][w] = mn[1]; red[2][w] = mn[2]; red[3][w] = mx; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        const float m = fminf(fminf(red[a][0], red[a][1]), fminf(red[a][2], red[a][3]));
        s_minb[a] = (double)m - voxel * 0.5;   // open3d: voxel_min_bound = min_bound - voxel_size * 0.5, in double
        if (blockIdx.x == 0) hdr->minb[a] = s_minb[a];
    }
    if (threadIdx.x == 3 && blockIdx.x == 0) hdr->imax = fmaxf(fmaxf(red[3][0], red[3][1]), fmaxf(red[3][2], red[3][3]));
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    unsigned long long key = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        // open3d: floor((point - voxel_min_bound) / voxel_size) in double
        const double r = ((double)pts[(size_t)i * ldp + a] - s_minb[a]) / voxel;
        long long v = (long long)floor(r);
        if (v < 0 || v >= (1 << KEY_BITS)) { hdr->overflow = 1; v = v < 0 ? 0 : (1 << KEY_BITS) - 1; }
        key = (key << KEY_BITS) | (unsigned long long)v;
    }
    keys[i] = key;
    idx[i] = i;
}

// ---- one radix pass ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * SORT_WPB) void sort_hist_kernel(const unsigned long long *kin, int N, int shift, int tile, int *hist) {
    __shared__ int cnt[SORT_WPB][256];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = blockIdx.x * SORT_WPB + wv;
    for (int d = lane; d < 256; d += 64) cnt[wv][d] = 0;
    const int b = min(N, t * tile), e = min(N, b + tile);
    for (int i = b + lane; i < e; i += 64) atomicAdd(&cnt[wv][(int)((kin[i] >> shift) & 255)], 1);   // LDS integer add: order-free
    for (int d = lane; d < 256; d += 64) hist[d * SORT_TILES + t] = cnt[wv][d];
}

// wave per digit: hist[d][*] -> exclusive offsets inside the digit, tot[d] = the digit's count
__global__ __launch_bounds__(256) void sort_scan_kernel(int *hist, int *tot) {
    const int d = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    int carry = 0;
    for (int t0 = 0; t0 < SORT_TILES; t0 += 64) {
        const int v = hist[d * SORT_TILES + t0 + lane];
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o, 64);
            if (lane >= o) incl += u;
        }
        hist[d * SORT_TILES + t0 + lane] = carry + incl - v;
        carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) tot[d] = carry;
}

__global__ __launch_bounds__(64 * SORT_WPB) void sort_scatter_kernel(const unsigned long long *kin, const int *vin, unsigned long long *kout, int *vout,
                                                                    int N, int shift, int tile, const int *hist, const int *tot) {
    __shared__ int cnt[SORT_WPB][256];
    __shared__ int dbase[256];
    __shared__ int wsum[SORT_WPB];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = blockIdx.x * SORT_WPB + wv;
    {   // exclusive scan of the 256 digit totals (threads = digits)
        const int v = tot[threadIdx.x];
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o, 64);
            if (lane >= o) incl += u;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int off = 0;
        for (int w = 0; w < wv; ++w) off += wsum[w];
        dbase[threadIdx.x] = off + incl - v;
        __syncthreads();
    }
    for (int d = lane; d < 256; d += 64) cnt[wv][d] = dbase[d] + hist[d * SORT_TILES + t];
    const int b = min(N, t * tile), e = min(N, b + tile);
    for (int i0 = b; i0 < e; i0 += 64) {
        const int i = i0 + lane;
        const bool valid = i < e;
        const unsigned long long key = valid ? kin[i] : 0ull;
        const int val = valid ? vin[i] : 0;
        const int d = (int)((key >> shift) & 255);
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const bool on = (d >> bit) & 1;
            const unsigned long long bm = __ballot(on);
            peers &= on ? bm : ~bm;
        }
        const int rank = __popcll(peers & ((1ull << lane) - 1ull));
        const int base = cnt[wv][d];            // every lane of a digit reads the same running offset ...
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            kout[base + rank] = key;
            vout[base + rank] = val;
            if (rank == 0) cnt[wv][d] = base + __popcll(peers);   // ... and its first lane advances it (one wave: program order)
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- segment heads of the sorted keys -> voxel ids ------------------------------------------------------------------------------
__global__ __launch_bounds__(RED_T) void vox_head_count_kernel(const unsigned long long *keys, int N, int *chunk_heads) {
    __shared__ int red[RED_T / 64];
    const int i = blockIdx.x * RED_T + threadIdx.x;
    const int h = (i < N && (i == 0 || keys[i] != keys[i - 1])) ? 1 : 0;
    const int c = __popcll(__ballot(h));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int k = 0; k < RED_T / 64; ++k) s += red[k];
        chunk_heads[blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(RED_T) void vox_head_write_kernel(const unsigned long long *keys, int N, const int *chunk_heads, int nchunk, int *head_pos,
                                                              VoxHeader *hdr, int32_t *count_dev) {
    __shared__ int red[RED_T / 64];
    __shared__ int s_before;
    // heads in the chunks before this one (and, for the last chunk, the total)
    int part = 0;
    for (int c = threadIdx.x; c < (int)blockIdx.x; c += RED_T) part += chunk_heads[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int k = 0; k < RED_T / 64; ++k) s += red[k];
        s_before = s;
    }
    __syncthreads();
    const int before = s_before;
    __syncthreads();
    const int i = blockIdx.x * RED_T + threadIdx.x;
    const int h = (i < N && (i == 0 || keys[i] != keys[i - 1])) ? 1 : 0;
    const unsigned long long bal = __ballot(h);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) red[w] = __popcll(bal);
    __syncthreads();
    int woff = before;
    for (int k = 0; k < w; ++k) woff += red[k];
    if (h) head_pos[woff + __popcll(bal & ((1ull << lane) - 1ull))] = i;   // voxel id -> first sorted position
    if ((int)blockIdx.x == nchunk - 1 && threadIdx.x == 0) {
        int total = before;
        for (int k = 0; k < RED_T / 64; ++k) total += red[k];
        hdr->nvox = total;
        count_dev[0] = total;
        count_dev[1] = hdr->overflow;
    }
}

// out row v = [mean xyz | mean(intensity / imax) * imax | mean normal], 8 floats (last one 0)
__global__ void vox_mean_kernel(const float *pts, int ldp, const float *inten, int ldi, const float *nrm, int ldn, const int *sorted_idx, const int *head_pos,
                                const VoxHeader *hdr, int N, int cap, float *out) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    const int nv = hdr->nvox;
    if (v >= nv || v >= cap) return;
    const int b = head_pos[v], e = v + 1 < nv ? head_pos[v + 1] : N;
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    const float imax = hdr->imax;
    for (int s = b; s < e; ++s) {
        const int i = sorted_idx[s];
#pragma unroll
        for (int a = 0; a < 3; ++a) acc[a] += (double)pts[(size_t)i * ldp + a];
        acc[3] += (double)(inten[(size_t)i * ldi] / imax);   // kitti.py:154: float32 intensity / float32 max, stored into open3d's double colours
#pragma unroll
        for (int a = 0; a < 3; ++a) acc[4 + a] += (double)nrm[(size_t)i * ldn + a];
    }
    const double inv = 1.0 / (double)(e - b);
    float *o = out + (size_t)v * 8;
#pragma unroll
    for (int a = 0; a < 3; ++a) o[a] = (float)(acc[a] * inv);
    o[3] = (float)(acc[3] * inv) * imax;       // kitti.py:163: float32 colour * float32 max
#pragma unroll
    for (int a = 0; a < 3; ++a) o[4 + a] = (float)(acc[4 + a] * inv);
    o[7] = 0.f;
}

// raw scan, channel-major (7, N) = [x y z | intensity | normal] as stored on disk  ->  rows (N, 8) = [T p | intensity | R n | 0] with the
// calibration transform T = P_cam * Tr (kitti.py:273-277).  float32 multiply-adds in the order k = 0, 1, 2.
__global__ void pack_transform_kernel(const float *data, int N, const float *P44, float *rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float p[3], nn[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float sp = 0.f, sn = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            sp = sp + P44[4 * a + k] * data[(size_t)k * N + i];
            sn = sn + P44[4 * a + k] * data[(size_t)(4 + k) * N + i];
        }
        p[a] = sp + P44[4 * a + 3];
        nn[a] = sn;
    }
    f32x4 lo = {p[0], p[1], p[2], data[(size_t)3 * N + i]}, hi = {nn[0], nn[1], nn[2], 0.f};
    *reinterpret_cast<f32x4 *>(rows + (size_t)i * 8) = lo;
    *reinterpret_cast<f32x4 *>(rows + (size_t)i * 8 + 4) = hi;
}

// rows choice[i] of the (nvox, 8) table -> points (n,3) = R p + t, feats (n,4) = [intensity | R n]     (kitti.py:284-288, 293)
// The reference multiplies in float64 (np.dot of the float32 4x4 P with float32 arrays promotes nothing: float32 BLAS) - restated as
// sequential float32 multiply-adds in the order k = 0, 1, 2 (cofii2p_amd/dataside.py documents the same order for the oracle).
__global__ void gather_transform_kernel(const float *vox, const int32_t *choice, int n, const float *P44, float *points, float *feats, int feats_are_points) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *r = vox + (size_t)choice[i] * 8;
    float p[3], nn[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float sp = 0.f, sn = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            sp = sp + P44[4 * a + k] * r[k];
            sn = sn + P44[4 * a + k] * r[4 + k];
        }
        p[a] = sp + P44[4 * a + 3];
        nn[a] = sn;
    }
    points[3 * i] = p[0]; points[3 * i + 1] = p[1]; points[3 * i + 2] = p[2];
    // KITTI: [intensity | R n] (kitti.py:293); nuScenes has no normals: [intensity | transformed point] (nuscenes.py:204)
    feats[4 * i] = r[3];
    feats[4 * i + 1] = feats_are_points ? p[0] : nn[0];
    feats[4 * i + 2] = feats_are_points ? p[1] : nn[1];
    feats[4 * i + 3] = feats_are_points ? p[2] : nn[2];
}

// cv2.resize(img, (dw, dh), INTER_LINEAR) on uint8 HWC in OpenCV's fixed-point arithmetic: 11-bit coefficients, horizontal pass exact
// in int32, vertical pass as the uchar specialisation of VResizeLinear writes it - each term TRUNCATED on its own,
//     ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
// (not the single rounding shift by 22 of the generic FixedPtCast: the two differ by one LSB whenever the low bits of the two terms
// carry, e.g. KITTI's 1241 x 376 -> 620 x 188) - then the crop [cy, cy + H) x [cx, cx + W), / 255 and HWC -> CHW (kitti.py:306-322, 375).  One thread per output value.
__global__ void resize_crop_kernel(const uint8_t *src, int sh, int sw, int dh, int dw, int cy, int cx, int H, int W, float *out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 3 * H * W) return;
    const int c = e / (H * W), y = (e / W) % H, x = e % W;
    const int dy = y + cy, dx = x + cx;
    const float scale_x = (float)((double)sw / dw), scale_y = (float)((double)sh / dh);   // OpenCV: double ratio used as float per coordinate
    auto coef = [](int d, float scale, int ssize, int &s0, int &a0, int &a1) {
        float f = (float)((d + 0.5) * (double)scale - 0.5);
        int s = (int)floorf(f);
        f -= s;
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }   // OpenCV clamps the last column/row onto the border pixel
        s0 = s;
        // saturate_cast<short>(v * 2048): round half to even (cvRound)
        a0 = (int)rintf((1.f - f) * 2048.f);
        a1 = (int)rintf(f * 2048.f);
    };
    int sx, ax0, ax1, sy, ay0, ay1;
    coef(dx, scale_x, sw, sx, ax0, ax1);
    coef(dy, scale_y, sh, sy, ay0, ay1);
    const int sx1 = min(sx + 1, sw - 1), sy1 = min(sy + 1, sh - 1);
    auto px = [&](int yy, int xx) { return (int)src[((size_t)yy * sw + xx) * 3 + c]; };
    const int r0 = px(sy, sx) * ax0 + px(sy, sx1) * ax1;     // horizontal pass, scale 2^11
    const int r1 = px(sy1, sx) * ax0 + px(sy1, sx1) * ax1;
    const int v = (((ay0 * (r0 >> 4)) >> 16) + ((ay1 * (r1 >> 4)) >> 16) + 2) >> 2;    // vertical pass, VResizeLinear<uchar, int, short, ...>
    out[e] = (float)min(max(v, 0), 255) / 255.f;
}


// ---- grid sub-sampling of a point set (model/kpconv/ops/grid_subsample.py -> geotransformer.ext.grid_subsampling, the KPConv
// barycentre sub-sampler): origin = floor(min * (1 / dl)) * dl, cell = floor((p - origin) / dl) per axis, output = barycentre of every
// occupied cell, all in float32 with sums in input order - the arithmetic of the published C++ (KPConv-PyTorch
// cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp).  Output order here: ascending (iz, iy, ix) = ascending
// map index of that code; the C++ emits its unordered_map's iteration order (unspecified).  Same sort / head-scan kernels as above.
__global__ __launch_bounds__(256) void grid_keys_kernel(const float *pts, int N, float dl, const float *part, int nchunk, VoxHeader *hdr,
                                                       unsigned long long *keys, int *idx) {
    __shared__ float red[3][4];
    __shared__ float s_org[3];
    float mn[3] = {INFINITY, INFINITY, INFINITY};
    for (int c = threadIdx.x; c < nchunk; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(part + 4 * c);
        mn[0] = fminf(mn[0], v[0]); mn[1] = fminf(mn[1], v[1]); mn[2] = fminf(mn[2], v[2]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int a = 0; a < 3; ++a) mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64));
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = mn[0]; red[1][w] = mn[1]; red[2][w] = mn[2]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        const float m = fminf(fminf(red[a][0], red[a][1]), fminf(red[a][2], red[a][3]));
        s_org[a] = floorf(m * (1.0f / dl)) * dl;
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    unsigned long long key = 0;
#pragma unroll
    for (int a = 2; a >= 0; --a) {   // z is the slowest axis of the map index
        long long v = (long long)floorf((pts[(size_t)i * 3 + a] - s_org[a]) / dl);
        if (v < 0 || v >= (1 << KEY_BITS)) { hdr->overflow = 1; v = v < 0 ? 0 : (1 << KEY_BITS) - 1; }
        key = (key << KEY_BITS) | (unsigned long long)v;
    }
    keys[i] = key;
    idx[i] = i;
}

__global__ void grid_mean_kernel(const float *pts, const int *sorted_idx, const int *head_pos, const VoxHeader *hdr, int N, int cap, float *out) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    const int nv = hdr->nvox;
    if (v >= nv || v >= cap) return;
    const int b = head_pos[v], e = v + 1 < nv ? head_pos[v + 1] : N;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int s = b; s < e; ++s) {
        const float *p = pts + (size_t)sorted_idx[s] * 3;
        sx += p[0]; sy += p[1]; sz += p[2];
    }
    const float inv = (float)(1.0 / (double)(e - b));
    out[(size_t)v * 3] = sx * inv;
    out[(size_t)v * 3 + 1] = sy * inv;
    out[(size_t)v * 3 + 2] = sz * inv;
}

// out[m][j] = idx[m][j] + offset if dist[m][j] < r2 (and the slot is a real neighbour) else fill; max_count[0] = max over rows of the
// number of neighbours kept (integer atomicMax: order-free).  model/kpconv/ops/radius_search.py on top of the sorted k-nearest rows.
__global__ void radius_mask_kernel(const int32_t *idx, const float *dist, int M, int k, int S, float r2, long long offset, long long fill,
                                   long long *out, int32_t *max_count) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= M) return;
    int cnt = 0;
    for (int j = lane; j < k; j += 64) {
        const int id = idx[(size_t)m * k + j];
        const bool keep = (unsigned)id < (unsigned)S && dist[(size_t)m * k + j] < r2;
        out[(size_t)m * k + j] = keep ? (long long)id + offset : fill;
        cnt += __popcll(__ballot(keep));
    }
    if (lane == 0) atomicMax(max_count, cnt);
}

// ---- train-mode image augmentation: torchvision ColorJitter on the PIL image (kitti.py:193-201) -----------------------------------------
// The four operations of torchvision's transforms/_functional_pil.py on the cropped uint8 image, which lives here as the (3, H, W) float
// image of cofi_resize_crop_image (every value k / 255): PIL.ImageEnhance.Brightness / Contrast / Color = Image.blend(degenerate, img,
// factor) in float32 with truncation, and the hue shift through PIL's RGB <-> HSV conversion (float ratios, double literals: the
// operation order of libImaging/Convert.c).  Bit-equal to oracle/dataside_oracle.py::color_jitter, which is pinned to PIL itself.
__device__ __forceinline__ int cj_u8(float x) { return (int)(x * 255.0f + 0.5f); }
__device__ __forceinline__ int cj_gray(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }
__device__ __forceinline__ int cj_blend(int deg, int v, float a, bool inside) {
    float t = (float)deg + a * ((float)v - (float)deg);
    if (!inside) t = fminf(fmaxf(t, 0.f), 255.f);
    return (int)t;
}

__global__ void cj_gray_sum_kernel(const float *img, int P, unsigned long long *sum) {
    unsigned long long s = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x)
        s += (unsigned long long)cj_gray(cj_u8(img[p]), cj_u8(img[P + p]), cj_u8(img[2 * P + p]));
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(sum, s);   // integer sum: order-free
}

__global__ void cj_op_kernel(float *img, int P, int op, float factor, int shift, const unsigned long long *gray_sum) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    int r = cj_u8(img[p]), g = cj_u8(img[P + p]), b = cj_u8(img[2 * P + p]);
    const bool inside = factor >= 0.f && factor <= 1.f;
    if (op == 0) {                       // brightness: blend with black
        r = cj_blend(0, r, factor, inside); g = cj_blend(0, g, factor, inside); b = cj_blend(0, b, factor, inside);
    } else if (op == 1) {                // contrast: blend with the rounded mean of the grey image
        const int mean = (int)((double)*gray_sum / (double)P + 0.5);
        r = cj_blend(mean, r, factor, inside); g = cj_blend(mean, g, factor, inside); b = cj_blend(mean, b, factor, inside);
    } else if (op == 2) {                // saturation: blend with the grey image
        const int l = cj_gray(r, g, b);
        r = cj_blend(l, r, factor, inside); g = cj_blend(l, g, factor, inside); b = cj_blend(l, b, factor, inside);
    } else {                             // hue: RGB -> HSV, H += shift (mod 256), HSV -> RGB
        const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
        int uh = 0, us = 0;
        if (minc != maxc) {
            const float cr = (float)(maxc - minc);
            const float sat = cr / (float)maxc;
            const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
            float h;
            if (r == maxc) h = bc - gc;
            else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
            else h = (float)(4.0 + (double)gc - (double)rc);
            h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
            uh = min(max((int)((double)h * 255.0), 0), 255);
            us = min(max((int)((double)sat * 255.0), 0), 255);
        }
        const int h8 = (uh + shift) & 255, v8 = maxc;
        if (us == 0) {
            r = g = b = v8;
        } else {
            const double hf = (double)h8 * 6.0 / 255.0;
            const int i = (int)floor(hf);
            const double f = (double)(float)(hf - (double)i), fs = (double)(float)((double)us / 255.0), vf = (double)v8;
            const int pp = min(max((int)floor(vf * (1.0 - fs) + 0.5), 0), 255);
            const int qq = min(max((int)floor(vf * (1.0 - fs * f) + 0.5), 0), 255);
            const int tt = min(max((int)floor(vf * (1.0 - fs * (1.0 - f)) + 0.5), 0), 255);
            switch (i % 6) {
                case 0: r = v8; g = tt; b = pp; break;
                case 1: r = qq; g = v8; b = pp; break;
                case 2: r = pp; g = v8; b = tt; break;
                case 3: r = pp; g = qq; b = v8; break;
                case 4: r = tt; g = pp; b = v8; break;
                default: r = v8; g = pp; b = qq; break;
            }
        }
    }
    img[p] = (float)r / 255.f;
    img[P + p] = (float)g / 255.f;
    img[2 * P + p] = (float)b / 255.f;
}

}  // namespace

static inline size_t vox_align(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" size_t cofi_voxel_downsample_workspace(int N) {
    if (N <= 0) return 0;
    // header | chunk partials (bounds: 4 floats, heads: 1 int per 1024 rows) | histogram table + digit totals | keys A, B | idx A, B | heads
    const size_t nchunk = (size_t)cofi_cdiv(N, RED_T);
    return 256 + vox_align(nchunk * 16) + vox_align(nchunk * 4) + vox_align((size_t)(256 * SORT_TILES + 256) * 4) + 2 * vox_align((size_t)N * 8) +
           3 * vox_align((size_t)N * 4);
}

extern "C" int cofi_pack_transform_scan(const float *data7n, int N, const float *P44_dev, float *rows8, cofi_stream_t stream) {
    if (!data7n || !P44_dev || !rows8 || N <= 0 || ((uintptr_t)rows8 & 15)) return COFI_EINVAL;
    hipLaunchKernelGGL(pack_transform_kernel, dim3(cofi_cdiv(N, 256)), dim3(256), 0, cofi_s(stream), data7n, N, P44_dev, rows8);
    return cofi_launch_status();
}

extern "C" int cofi_voxel_downsample(const float *rows8, int N, double voxel, float *out_rows8, int cap, int32_t *count_dev, void *ws, size_t ws_bytes,
                                     cofi_stream_t stream) {
    if (!rows8 || !out_rows8 || !count_dev || N <= 0 || !(voxel > 0.0) || cap <= 0) return COFI_EINVAL;
    if (!ws || ws_bytes < cofi_voxel_downsample_workspace(N) || ((uintptr_t)ws & 15)) return COFI_EWORKSPACE;
    hipStream_t s = cofi_s(stream);
    const float *points = rows8, *intensity = rows8 + 3, *normals = rows8 + 4;
    const int ldp = 8, ldi = 8, ldn = 8;
    const int nchunk = cofi_cdiv(N, RED_T);
    char *w = (char *)ws;
    VoxHeader *hdr = (VoxHeader *)w; w += 256;
    float *bpart = (float *)w; w += vox_align((size_t)nchunk * 16);
    int *hpart = (int *)w; w += vox_align((size_t)nchunk * 4);
    int *hist = (int *)w, *tot = hist + 256 * SORT_TILES; w += vox_align((size_t)(256 * SORT_TILES + 256) * 4);
    unsigned long long *kA = (unsigned long long *)w; w += vox_align((size_t)N * 8);
    unsigned long long *kB = (unsigned long long *)w; w += vox_align((size_t)N * 8);
    int *vA = (int *)w; w += vox_align((size_t)N * 4);
    int *vB = (int *)w; w += vox_align((size_t)N * 4);
    int *heads = (int *)w;
    (void)hipMemsetAsync(hdr, 0, sizeof(VoxHeader), s);
    hipLaunchKernelGGL(vox_bounds_kernel, dim3(nchunk), dim3(RED_T), 0, s, points, ldp, intensity, ldi, N, bpart);
    hipLaunchKernelGGL(vox_keys_kernel, dim3(cofi_cdiv(N, 256)), dim3(256), 0, s, points, ldp, N, voxel, bpart, nchunk, hdr, kA, vA);
    const int tile = cofi_cdiv(N, SORT_TILES);
    for (int pass = 0; pass < SORT_PASSES; ++pass) {
        hipLaunchKernelGGL(sort_hist_kernel, dim3(SORT_TILES / SORT_WPB), dim3(64 * SORT_WPB), 0, s, kA, N, 8 * pass, tile, hist);
        hipLaunchKernelGGL(sort_scan_kernel, dim3(64), dim3(256), 0, s, hist, tot);
        hipLaunchKernelGGL(sort_scatter_kernel, dim3(SORT_TILES / SORT_WPB), dim3(64 * SORT_WPB), 0, s, kA, vA, kB, vB, N, 8 * pass, tile, hist, tot);
        unsigned long long *tk = kA; kA = kB; kB = tk;
        int *tv = vA; vA = vB; vB = tv;
    }
    hipLaunchKernelGGL(vox_head_count_kernel, dim3(nchunk), dim3(RED_T), 0, s, kA, N, hpart);
    hipLaunchKernelGGL(vox_head_write_kernel, dim3(nchunk), dim3(RED_T), 0, s, kA, N, hpart, nchunk, heads, hdr, count_dev);
    hipLaunchKernelGGL(vox_mean_kernel, dim3(cofi_cdiv(N, 256)), dim3(256), 0, s, points, ldp, intensity, ldi, normals, ldn, vA, heads, hdr, N, cap,
                       out_rows8);
    // count_dev[0] = voxels (may exceed cap: only the first cap rows were written), count_dev[1] = 1 if a voxel index overflowed the
    // 13-bit key field (coordinates spread over more than 819 m at this voxel size) - written by the last head-scan workgroup
    return cofi_launch_status();
}

extern "C" int cofi_grid_subsample(const float *points, int N, float voxel, float *out_points, int cap, int32_t *count_dev, void *ws, size_t ws_bytes,
                                   cofi_stream_t stream) {
    if (!points || !out_points || !count_dev || N <= 0 || !(voxel > 0.f) || cap <= 0) return COFI_EINVAL;
    if (!ws || ws_bytes < cofi_voxel_downsample_workspace(N) || ((uintptr_t)ws & 15)) return COFI_EWORKSPACE;
    hipStream_t s = cofi_s(stream);
    const int nchunk = cofi_cdiv(N, RED_T);
    char *w = (char *)ws;
    VoxHeader *hdr = (VoxHeader *)w; w += 256;
    float *bpart = (float *)w; w += vox_align((size_t)nchunk * 16);
    int *hpart = (int *)w; w += vox_align((size_t)nchunk * 4);
    int *hist = (int *)w, *tot = hist + 256 * SORT_TILES; w += vox_align((size_t)(256 * SORT_TILES + 256) * 4);
    unsigned long long *kA = (unsigned long long *)w; w += vox_align((size_t)N * 8);
    unsigned long long *kB = (unsigned long long *)w; w += vox_align((size_t)N * 8);
    int *vA = (int *)w; w += vox_align((size_t)N * 4);
    int *vB = (int *)w; w += vox_align((size_t)N * 4);
    int *heads = (int *)w;
    (void)hipMemsetAsync(hdr, 0, sizeof(VoxHeader), s);
    hipLaunchKernelGGL(vox_bounds_kernel, dim3(nchunk), dim3(RED_T), 0, s, points, 3, points, 3, N, bpart);
    hipLaunchKernelGGL(grid_keys_kernel, dim3(cofi_cdiv(N, 256)), dim3(256), 0, s, points, N, voxel, bpart, nchunk, hdr, kA, vA);
    const int tile = cofi_cdiv(N, SORT_TILES);
    for (int pass = 0; pass < SORT_PASSES; ++pass) {
        hipLaunchKernelGGL(sort_hist_kernel, dim3(SORT_TILES / SORT_WPB), dim3(64 * SORT_WPB), 0, s, kA, N, 8 * pass, tile, hist);
        hipLaunchKernelGGL(sort_scan_kernel, dim3(64), dim3(256), 0, s, hist, tot);
        hipLaunchKernelGGL(sort_scatter_kernel, dim3(SORT_TILES / SORT_WPB), dim3(64 * SORT_WPB), 0, s, kA, vA, kB, vB, N, 8 * pass, tile, hist, tot);
        unsigned long long *tk = kA; kA = kB; kB = tk;
        int *tv = vA; vA = vB; vB = tv;
    }
    hipLaunchKernelGGL(vox_head_count_kernel, dim3(nchunk), dim3(RED_T), 0, s, kA, N, hpart);
    hipLaunchKernelGGL(vox_head_write_kernel, dim3(nchunk), dim3(RED_T), 0, s, kA, N, hpart, nchunk, heads, hdr, count_dev);
    hipLaunchKernelGGL(grid_mean_kernel, dim3(cofi_cdiv(N, 256)), dim3(256), 0, s, points, vA, heads, hdr, N, cap, out_points);
    return cofi_launch_status();
}

extern "C" int cofi_radius_mask(const int32_t *idx, const float *dist, int M, int k, int S, float radius, long long offset, long long fill,
                                long long *out, int32_t *max_count_dev, cofi_stream_t stream) {
    if (!idx || !dist || !out || !max_count_dev || M < 0 || k <= 0 || S <= 0 || !(radius > 0.f)) return COFI_EINVAL;
    if (M == 0) return 0;
    hipLaunchKernelGGL(radius_mask_kernel, dim3(cofi_cdiv(M, 4)), dim3(256), 0, cofi_s(stream), idx, dist, M, k, S, radius * radius, offset, fill, out,
                       max_count_dev);
    return cofi_launch_status();
}

extern "C" int cofi_gather_transform(const float *vox_rows, const int32_t *choice, int n, const float *P44_dev, float *points, float *feats,
                                     int feats_are_points, cofi_stream_t stream) {
    if (!vox_rows || !choice || !P44_dev || !points || !feats || n <= 0) return COFI_EINVAL;
    hipLaunchKernelGGL(gather_transform_kernel, dim3(cofi_cdiv(n, 256)), dim3(256), 0, cofi_s(stream), vox_rows, choice, n, P44_dev, points, feats, feats_are_points);
    return cofi_launch_status();
}

extern "C" int cofi_resize_crop_image(const uint8_t *src_hwc, int src_h, int src_w, int dst_h, int dst_w, int crop_y, int crop_x, int H, int W,
                                      float *out_chw, cofi_stream_t stream) {
    if (!src_hwc || !out_chw || src_h <= 0 || src_w <= 0 || dst_h <= 0 || dst_w <= 0 || H <= 0 || W <= 0) return COFI_EINVAL;
    if (crop_y < 0 || crop_x < 0 || crop_y + H > dst_h || crop_x + W > dst_w) return COFI_EINVAL;
    hipLaunchKernelGGL(resize_crop_kernel, dim3(cofi_cdiv(3 * H * W, 256)), dim3(256), 0, cofi_s(stream), src_hwc, src_h, src_w, dst_h, dst_w, crop_y, crop_x,
                       H, W, out_chw);
    return cofi_launch_status();
}

extern "C" int cofi_color_jitter_chw(float *img, int H, int W, const int *order4, float brightness, float contrast, float saturation, float hue,
                                     void *ws, size_t ws_bytes, cofi_stream_t stream) {
    if (!img || !order4 || H <= 0 || W <= 0 || !ws || ws_bytes < 8 || ((uintptr_t)ws & 7)) return COFI_EINVAL;
    int seen = 0;
    for (int i = 0; i < 4; ++i) {
        if (order4[i] < 0 || order4[i] > 3) return COFI_EINVAL;
        seen |= 1 << order4[i];
    }
    if (seen != 15 || !(brightness >= 0.f) || !(contrast >= 0.f) || !(saturation >= 0.f) || !(hue >= -0.5f && hue <= 0.5f)) return COFI_EINVAL;
    const int P = H * W;
    hipStream_t s = cofi_s(stream);
    const float fac[4] = {brightness, contrast, saturation, 0.f};
    const int shift = (int)(unsigned char)(int)((double)hue * 255.0);   // np.uint8 cast of hue_factor * 255: truncation, modulo 256
    for (int i = 0; i < 4; ++i) {
        const int op = order4[i];
        if (op == 1) {
            if (hipError_t e = hipMemsetAsync(ws, 0, 8, s); e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(cj_gray_sum_kernel, dim3(cofi_cdiv(P, 1024) > 256 ? 256 : cofi_cdiv(P, 1024)), dim3(256), 0, s, img, P, (unsigned long long *)ws);
        }
        hipLaunchKernelGGL(cj_op_kernel, dim3(cofi_cdiv(P, 256)), dim3(256), 0, s, img, P, op, fac[op], shift, (const unsigned long long *)ws);
    }
    return cofi_launch_status();
}
