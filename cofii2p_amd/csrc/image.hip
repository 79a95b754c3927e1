// Glue kernels of the image branch on NHWC = pixel-major (H*W, C) maps (reference: model/imagenet.py): the 7x7 stem as an
// explicit im2col matrix, the 3x3/2 max-pool and the bilinear x2 up-sampling fused with the channel concatenation.  The
// convolutions are implicit GEMMs on the MFMA kernel (gemm.hip); InstanceNorm + ReLU + residual is the GroupNorm apply
// kernel with one group per channel (norm.hip) or is folded into the next convolution's operand loader.
#include "common.h"

namespace {

// stem: out[p, (dy*7+dx)*3 + c] = img[c, 2*yo - 3 + dy, 2*xo - 3 + dx], zero outside / beyond 147
__global__ void im2col_stem_kernel(const float *img, int H, int W, int Ho, int Wo, int Kpad, float *out, int frames) {
    const size_t total = (size_t)frames * Ho * Wo * Kpad;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(e % Kpad), pg = (int)(e / Kpad), f = pg / (Ho * Wo), p = pg - f * (Ho * Wo);
        float v = 0.f;
        if (k < 147) {
            const int tap = k / 3, c = k - 3 * tap, dy = tap / 7, dx = tap - 7 * dy;
            const int yi = 2 * (p / Wo) - 3 + dy, xi = 2 * (p % Wo) - 3 + dx;
            if ((unsigned)yi < (unsigned)H && (unsigned)xi < (unsigned)W) v = img[(((size_t)f * 3 + c) * H + yi) * W + xi];
        }
        out[e] = v;
    }
}

__global__ void maxpool3x3s2_nhwc_kernel(const float *x0, int H, int W, int C, int Ho, int Wo, float *y, int frames) {
    const int c4n = C >> 2;
    const size_t total = (size_t)frames * Ho * Wo * c4n;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % c4n), pg = (int)(e / c4n), f = pg / (Ho * Wo), pl = pg - f * (Ho * Wo), yo = pl / Wo, xo = pl - yo * Wo;
        const float *x = x0 + (size_t)f * H * W * C;
        const int p = pg;
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        for (int dy = 0; dy < 3; ++dy)
            for (int dx = 0; dx < 3; ++dx) {
                const int yi = 2 * yo - 1 + dy, xi = 2 * xo - 1 + dx;
                if ((unsigned)yi < (unsigned)H && (unsigned)xi < (unsigned)W) {
                    const float4 v = reinterpret_cast<const float4 *>(x + ((size_t)yi * W + xi) * C)[c4];
                    m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
                }
            }
        reinterpret_cast<float4 *>(y + (size_t)p * C)[c4] = m;
    }
}

__global__ void upsample2x_cat_nhwc_kernel(const float *low0, int ldl, int C1, int h, int w, const float *skip, int lds, int C2, float *out,
                                           int ldo, int frames) {
    const int H = 2 * h, W = 2 * w, ct = (C1 + C2) >> 2;
    const size_t total = (size_t)frames * H * W * ct;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % ct) * 4, p = (int)(e / ct), f = p / (H * W), pl = p - f * (H * W), yo = pl / W, xo = pl - yo * W;
        const float *low = low0 + (size_t)f * h * w * ldl;
        float4 v;
        if (c >= C1) {
            v = *reinterpret_cast<const float4 *>(skip + (size_t)p * lds + (c - C1));
        } else {
            const float sy = fmaxf(0.5f * ((float)yo + 0.5f) - 0.5f, 0.f), sx = fmaxf(0.5f * ((float)xo + 0.5f) - 0.5f, 0.f);
            const int y0 = (int)sy, x0 = (int)sx;
            const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
            const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
            const float4 a = *reinterpret_cast<const float4 *>(low + ((size_t)y0 * w + x0) * ldl + c);
            const float4 b = *reinterpret_cast<const float4 *>(low + ((size_t)y0 * w + x1) * ldl + c);
            const float4 cc = *reinterpret_cast<const float4 *>(low + ((size_t)y1 * w + x0) * ldl + c);
            const float4 d = *reinterpret_cast<const float4 *>(low + ((size_t)y1 * w + x1) * ldl + c);
            v.x = hy * (hx * a.x + lx * b.x) + ly * (hx * cc.x + lx * d.x);
            v.y = hy * (hx * a.y + lx * b.y) + ly * (hx * cc.y + lx * d.y);
            v.z = hy * (hx * a.z + lx * b.z) + ly * (hx * cc.z + lx * d.z);
            v.w = hy * (hx * a.w + lx * b.w) + ly * (hx * cc.w + lx * d.w);
        }
        *reinterpret_cast<float4 *>(out + (size_t)p * ldo + c) = v;
    }
}

}  // namespace

extern "C" int cofi_im2col_stem(const float *img_chw, int H, int W, int Kpad, float *out, int frames, cofi_stream_t stream) {
    if (!img_chw || !out || H <= 0 || W <= 0 || Kpad < 147 || (Kpad & 3) || frames <= 0) return COFI_EINVAL;
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    size_t total = (size_t)frames * Ho * Wo * Kpad;
    int nb = (int)((total + 255) / 256);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(im2col_stem_kernel, dim3(nb), dim3(256), 0, cofi_s(stream), img_chw, H, W, Ho, Wo, Kpad, out, frames);
    return cofi_launch_status();
}

extern "C" int cofi_maxpool3x3s2_nhwc(const float *x, int H, int W, int C, float *y, int frames, cofi_stream_t stream) {
    if (!x || !y || H <= 0 || W <= 0 || C <= 0 || (C & 3) || frames <= 0) return COFI_EINVAL;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    size_t total = (size_t)frames * Ho * Wo * (C >> 2);
    int nb = (int)((total + 255) / 256);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(maxpool3x3s2_nhwc_kernel, dim3(nb), dim3(256), 0, cofi_s(stream), x, H, W, C, Ho, Wo, y, frames);
    return cofi_launch_status();
}

extern "C" int cofi_upsample2x_cat_nhwc(const float *low, int ldl, int C1, int h, int w, const float *skip, int lds, int C2, float *out,
                                        int ldo, int frames, cofi_stream_t stream) {
    if (!low || !out || C1 <= 0 || h <= 0 || w <= 0 || C2 < 0 || (C2 && !skip) || (C1 & 3) || (C2 & 3) || (ldl & 3) || (lds & 3) || (ldo & 3))
        return COFI_EINVAL;
    if (frames <= 0) return COFI_EINVAL;
    size_t total = (size_t)frames * 4 * h * w * ((C1 + C2) >> 2);
    int nb = (int)((total + 255) / 256);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(upsample2x_cat_nhwc_kernel, dim3(nb), dim3(256), 0, cofi_s(stream), low, ldl, C1, h, w, skip, lds, C2, out, ldo,
                       frames);
    return cofi_launch_status();
}
