// Data-side preprocessing of one frame on the device (reference: data/kitti.py:259-393, the part of
// kitti_pc_img_dataset.__getitem__ between the disk read and the model call):
//   cofi_voxel_downsample   data/kitti.py:145-166  open3d voxel_down_sample(0.1) of points + intensity ("colors") + normals
//   cofi_gather_transform   data/kitti.py:168-180, 284-288, 293  resample to exactly num_pc rows, random SE(3), feats = [intensity | R n]
//   cofi_resize_crop_image  data/kitti.py:306-322, 375  cv2.resize(INTER_LINEAR) x0.5, crop, / 255, HWC -> CHW
// The KNN pyramid (preprocess_data.py:36-107) is knn_grid.hip; the coarse / fine label projection works on 1280 points and stays
// host-side numpy, exactly as the reference writes it (cofii2p_amd/dataside.py).
//
// Voxel grid: every point gets the key of its voxel (open3d: index = floor((p - (min_bound - voxel/2)) / voxel) per axis), the
// (key, point index) pairs go through a STABLE least-significant-digit radix sort (4-bit digits; one workgroup, every thread owns a
// contiguous chunk, so equal keys keep their original order), and one thread per voxel averages its run in that order with fp64
// accumulators (open3d accumulates in double): bit-reproducible, no float atomics, no hash table.  Output order = ascending key
// (open3d's is the iteration order of an unordered_map: unspecified, parity with it is unpinned).
// These are HBM / latency-bound integer and streaming kernels on ~120 000 points: no MFMA anywhere.
#include "common.h"

namespace {

constexpr int SORT_T = 1024;      // threads of the sorting workgroup
constexpr int KEY_BITS = 13;      // voxel index bits per axis (8192 voxels of 0.1 m = 819 m)
constexpr int KEY_TOTAL = 3 * KEY_BITS;

struct VoxHeader {
    double minb[3];  // min_bound - voxel / 2 (open3d keeps it in double)
    float imax;      // max intensity
    int nvox;
    int overflow;    // a voxel index did not fit KEY_BITS
};

// bounding box minimum + intensity maximum: one workgroup, fixed-order fold
__global__ __launch_bounds__(1024) void vox_bounds_kernel(const float *pts, int ldp, const float *inten, int ldi, int N, double voxel, VoxHeader *hdr) {
    __shared__ float red[4][16];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx = -INFINITY;
    for (int i = threadIdx.x; i < N; i += 1024) {
#pragma unroll
        for (int a = 0; a < 3; ++a) mn[a] = fminf(mn[a], pts[(size_t)i * ldp + a]);
        mx = fmaxf(mx, inten[(size_t)i * ldi]);
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
    if (threadIdx.x == 0) {
        for (int k = 1; k < 16; ++k) {
            for (int a = 0; a < 3; ++a) red[a][0] = fminf(red[a][0], red[a][k]);
            red[3][0] = fmaxf(red[3][0], red[3][k]);
        }
        // open3d: voxel_min_bound = min_bound - voxel_size * 0.5 (computed in double from the float coordinates)
        for (int a = 0; a < 3; ++a) hdr->minb[a] = (double)red[a][0] - voxel * 0.5;
        hdr->imax = red[3][0];
        hdr->nvox = 0;
        hdr->overflow = 0;
    }
}

__global__ void vox_keys_kernel(const float *pts, int ldp, int N, double voxel, VoxHeader *hdr, unsigned long long *keys, int *idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    unsigned long long key = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        // open3d: floor((point - voxel_min_bound) / voxel_size) in double
        const double r = ((double)pts[(size_t)i * ldp + a] - hdr->minb[a]) / voxel;
        long long v = (long long)floor(r);
        if (v < 0 || v >= (1 << KEY_BITS)) { hdr->overflow = 1; v = v < 0 ? 0 : (1 << KEY_BITS) - 1; }
        key = (key << KEY_BITS) | (unsigned long long)v;
    }
    keys[i] = key;
    idx[i] = i;
}

// One pass of the stable LSD radix sort (4-bit digit at `shift`): thread t owns elements [t*chunk, (t+1)*chunk); per-thread digit
// counts -> LDS table [digit][thread] -> exclusive scan in that (digit-major) order = every thread's first output slot per digit.
__global__ __launch_bounds__(SORT_T) void vox_sort_pass_kernel(const unsigned long long *kin, const int *vin, unsigned long long *kout, int *vout,
                                                              int N, int shift) {
    __shared__ int tab[16 * SORT_T];
    __shared__ int wsum[SORT_T / 64];
    const int t = threadIdx.x;
    const int chunk = (N + SORT_T - 1) / SORT_T;
    const int b = min(N, t * chunk), e = min(N, b + chunk);
    int cnt[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) cnt[d] = 0;
    for (int i = b; i < e; ++i) {
        const int d = (int)((kin[i] >> shift) & 15);
#pragma unroll
        for (int q = 0; q < 16; ++q) cnt[q] += (q == d);
    }
#pragma unroll
    for (int d = 0; d < 16; ++d) tab[d * SORT_T + t] = cnt[d];
    __syncthreads();
    // exclusive scan of the 16 * SORT_T table: thread t scans entries [16 t, 16 t + 16), wave + workgroup offsets on top
    int loc[16], s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { loc[k] = s; s += tab[16 * t + k]; }
    int incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if ((t & 63) >= o) incl += v;
    }
    if ((t & 63) == 63) wsum[t >> 6] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < (t >> 6); ++w) woff += wsum[w];
    const int base = woff + incl - s;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) tab[16 * t + k] = base + loc[k];
    __syncthreads();
    int pos[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) pos[d] = tab[d * SORT_T + t];
    for (int i = b; i < e; ++i) {
        const unsigned long long k = kin[i];
        const int d = (int)((k >> shift) & 15);
        int p = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (q == d) p = pos[q]++;
        kout[p] = k;
        vout[p] = vin[i];
    }
}

// segment heads of the sorted keys -> voxel ids (exclusive scan by one workgroup), then one thread per voxel averages its run
__global__ __launch_bounds__(SORT_T) void vox_heads_kernel(const unsigned long long *keys, int N, int *head_pos, VoxHeader *hdr, int32_t *count_dev) {
    __shared__ int wsum[SORT_T / 64];
    __shared__ int carry;
    const int t = threadIdx.x;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int i0 = 0; i0 < N; i0 += SORT_T) {
        const int i = i0 + t;
        const int h = (i < N && (i == 0 || keys[i] != keys[i - 1])) ? 1 : 0;
        int incl = h;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if ((t & 63) >= o) incl += v;
        }
        if ((t & 63) == 63) wsum[t >> 6] = incl;
        __syncthreads();
        int woff = carry;
        for (int w = 0; w < (t >> 6); ++w) woff += wsum[w];
        if (h) head_pos[woff + incl - 1] = i;   // voxel id -> first sorted position
        __syncthreads();
        if (t == SORT_T - 1) carry = woff + incl;
        __syncthreads();
    }
    if (t == 0) { hdr->nvox = carry; count_dev[0] = carry; count_dev[1] = hdr->overflow; }
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
__global__ void gather_transform_kernel(const float *vox, const int32_t *choice, int n, const float *P44, float *points, float *feats) {
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
    feats[4 * i] = r[3]; feats[4 * i + 1] = nn[0]; feats[4 * i + 2] = nn[1]; feats[4 * i + 3] = nn[2];
}

// cv2.resize(img, (dw, dh), INTER_LINEAR) on uint8 HWC in OpenCV's fixed-point arithmetic (11-bit coefficients, rounding shift by 22),
// then the crop [cy, cy + H) x [cx, cx + W), / 255 and HWC -> CHW (kitti.py:306-322, 375).  One thread per output value.
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
    const int v = (r0 * ay0 + r1 * ay1 + (1 << 21)) >> 22;    // vertical pass + FixedPtCast<int, uchar, 22>
    out[e] = (float)min(max(v, 0), 255) / 255.f;
}

}  // namespace

extern "C" size_t cofi_voxel_downsample_workspace(int N) {
    if (N <= 0) return 0;
    // header | keys A, B (8 N each) | idx A, B (4 N each) | head positions (4 N)
    return 256 + (size_t)N * (8 + 8 + 4 + 4 + 4) + 64;
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
    char *w = (char *)ws;
    VoxHeader *hdr = (VoxHeader *)w;
    unsigned long long *kA = (unsigned long long *)(w + 256), *kB = kA + N;
    int *vA = (int *)(kB + N), *vB = vA + N, *heads = vB + N;
    hipLaunchKernelGGL(vox_bounds_kernel, dim3(1), dim3(1024), 0, s, points, ldp, intensity, ldi, N, voxel, hdr);
    hipLaunchKernelGGL(vox_keys_kernel, dim3(cofi_cdiv(N, 256)), dim3(256), 0, s, points, ldp, N, voxel, hdr, kA, vA);
    for (int shift = 0; shift < KEY_TOTAL; shift += 4) {   // 10 passes: an even count, the result is back in buffer A
        hipLaunchKernelGGL(vox_sort_pass_kernel, dim3(1), dim3(SORT_T), 0, s, kA, vA, kB, vB, N, shift);
        unsigned long long *tk = kA; kA = kB; kB = tk;
        int *tv = vA; vA = vB; vB = tv;
    }
    hipLaunchKernelGGL(vox_heads_kernel, dim3(1), dim3(SORT_T), 0, s, kA, N, heads, hdr, count_dev);
    hipLaunchKernelGGL(vox_mean_kernel, dim3(cofi_cdiv(N, 256)), dim3(256), 0, s, points, ldp, intensity, ldi, normals, ldn, vA, heads, hdr, N, cap,
                       out_rows8);
    // count_dev[0] = voxels (may exceed cap: only the first cap rows were written), count_dev[1] = 1 if a voxel index overflowed the
    // 13-bit key field (coordinates spread over more than 819 m at this voxel size)
    return cofi_launch_status();
}

extern "C" int cofi_gather_transform(const float *vox_rows, const int32_t *choice, int n, const float *P44_dev, float *points, float *feats,
                                     cofi_stream_t stream) {
    if (!vox_rows || !choice || !P44_dev || !points || !feats || n <= 0) return COFI_EINVAL;
    hipLaunchKernelGGL(gather_transform_kernel, dim3(cofi_cdiv(n, 256)), dim3(256), 0, cofi_s(stream), vox_rows, choice, n, P44_dev, points, feats);
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
