// Glue kernels of the image branch on channel-major (C, H*W) maps (reference: model/imagenet.py).
// The dense convolutions themselves run through MIOpen; everything between them —
// affine-less InstanceNorm (+ReLU, + residual or InstanceNorm'ed residual), folded-BatchNorm bias + ReLU +
// skip add, bilinear x2 up-sampling fused with the channel concatenation — is one pass each here.
#include "common.h"

namespace {

template <int NT>
__device__ __forceinline__ void block_stats(const float *p, int P, float eps, float &mean, float &rstd, double *red) {
    // two-level fp64 fold in a fixed order: deterministic
    float s = 0.f, q = 0.f;
    for (int i = threadIdx.x * 4; i < P; i += NT * 4) {
        if (i + 3 < P) {
            const float4 v = *reinterpret_cast<const float4 *>(p + i);
            s += (v.x + v.y) + (v.z + v.w);
            q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        } else {
            for (int e = i; e < P; ++e) { s += p[e]; q += p[e] * p[e]; }
        }
    }
    double ds = wave_sum_d((double)s), dq = wave_sum_d((double)q);
    const int wv = threadIdx.x >> 6, nw = NT / 64;
    if ((threadIdx.x & 63) == 0) { red[2 * wv] = ds; red[2 * wv + 1] = dq; }
    __syncthreads();
    ds = 0.0; dq = 0.0;
    for (int w = 0; w < nw; ++w) { ds += red[2 * w]; dq += red[2 * w + 1]; }
    __syncthreads();
    const double m = ds / P;
    double var = dq / P - m * m;
    if (var < 0.0) var = 0.0;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
}

// y[c,:] = relu?( IN(x[c,:]) + R ),  R = 0 | res[c,:] | IN(res[c,:])      (imagenet.py:58-73, 199-201)
template <int NT>
__global__ __launch_bounds__(NT) void instance_norm_kernel(const float *x, int P, float eps, const float *res, int res_mode, int relu,
                                                           float *y) {
    __shared__ double red[2 * (NT / 64)];
    const size_t off = (size_t)blockIdx.x * P;
    float mean, rstd, rmean = 0.f, rrstd = 1.f;
    block_stats<NT>(x + off, P, eps, mean, rstd, red);
    if (res_mode == 2) block_stats<NT>(res + off, P, eps, rmean, rrstd, red);
    for (int i = threadIdx.x * 4; i < P; i += NT * 4) {
        float v[4], r[4] = {0.f, 0.f, 0.f, 0.f};
        const int n = min(4, P - i);
        for (int e = 0; e < n; ++e) v[e] = x[off + i + e];
        if (res_mode)
            for (int e = 0; e < n; ++e) r[e] = res[off + i + e];
        for (int e = 0; e < n; ++e) {
            float t = (v[e] - mean) * rstd;
            if (res_mode == 1) t += r[e];
            if (res_mode == 2) t += (r[e] - rmean) * rrstd;
            y[off + i + e] = relu ? fmaxf(t, 0.f) : t;
        }
    }
}

// y = relu?( x + bias[c] + res ) on (C,P) maps, float4 when P % 4 == 0   (imagenet.py:397-411 with folded BN)
__global__ void bias_act_kernel(const float *x, const float *bias, const float *res, const float *res_bias, int C, int P, int relu,
                                float *y) {
    const size_t total = (size_t)C * P;
    for (size_t e = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; e < total; e += (size_t)gridDim.x * blockDim.x * 4) {
        const int c = (int)(e / P);
        float b = bias ? bias[c] : 0.f;
        if (res_bias) b += res_bias[c];
        const float4 v = *reinterpret_cast<const float4 *>(x + e);
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (res) r = *reinterpret_cast<const float4 *>(res + e);
        float4 o = make_float4(v.x + b + r.x, v.y + b + r.y, v.z + b + r.z, v.w + b + r.w);
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        *reinterpret_cast<float4 *>(y + e) = o;
    }
}

// out[0:C1] = bilinear_x2(low) (align_corners=False), out[C1:C1+C2] = skip     (imagenet.py:433,441-443)
__global__ void upsample2x_cat_kernel(const float *low, int C1, int h, int w, const float *skip, int C2, float *out) {
    const int H = 2 * h, W = 2 * w;
    const size_t total = (size_t)(C1 + C2) * H * W;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int xo = (int)(e % W), yo = (int)((e / W) % H), c = (int)(e / ((size_t)W * H));
        float v;
        if (c >= C1) {
            v = skip[((size_t)(c - C1) * H + yo) * W + xo];
        } else {
            // area_pixel_compute_source_index: scale * (dst + 0.5) - 0.5, clamped at 0
            const float sy = fmaxf(0.5f * ((float)yo + 0.5f) - 0.5f, 0.f), sx = fmaxf(0.5f * ((float)xo + 0.5f) - 0.5f, 0.f);
            const int y0 = (int)sy, x0 = (int)sx;
            const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
            const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
            const float *p = low + (size_t)c * h * w;
            v = hy * (hx * p[y0 * w + x0] + lx * p[y0 * w + x1]) + ly * (hx * p[y1 * w + x0] + lx * p[y1 * w + x1]);
        }
        out[e] = v;
    }
}

// ------------------------------------------------------------------------------------ NHWC kernels
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

extern "C" int cofi_instance_norm_nchw(const float *x, int C, int P, float eps, const float *res, int res_mode, int relu, float *y,
                                       cofi_stream_t stream) {
    if (!x || !y || C <= 0 || P <= 0 || res_mode < 0 || res_mode > 2 || (res_mode && !res) || (P & 3)) return COFI_EINVAL;
    if (P >= 8192)
        hipLaunchKernelGGL((instance_norm_kernel<1024>), dim3(C), dim3(1024), 0, cofi_s(stream), x, P, eps, res, res_mode, relu, y);
    else
        hipLaunchKernelGGL((instance_norm_kernel<256>), dim3(C), dim3(256), 0, cofi_s(stream), x, P, eps, res, res_mode, relu, y);
    return cofi_launch_status();
}

extern "C" int cofi_bias_act_nchw(const float *x, const float *bias, const float *res, const float *res_bias, int C, int P, int relu,
                                  float *y, cofi_stream_t stream) {
    if (!x || !y || C <= 0 || P <= 0 || (P & 3)) return COFI_EINVAL;
    size_t total4 = (size_t)C * P / 4;
    int nb = (int)((total4 + 255) / 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(bias_act_kernel, dim3(nb), dim3(256), 0, cofi_s(stream), x, bias, res, res_bias, C, P, relu, y);
    return cofi_launch_status();
}

extern "C" int cofi_upsample2x_cat(const float *low, int C1, int h, int w, const float *skip, int C2, float *out, cofi_stream_t stream) {
    if (!low || !out || C1 <= 0 || h <= 0 || w <= 0 || C2 < 0 || (C2 && !skip)) return COFI_EINVAL;
    size_t total = (size_t)(C1 + C2) * 4 * h * w;
    int nb = (int)((total + 255) / 256);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(upsample2x_cat_kernel, dim3(nb), dim3(256), 0, cofi_s(stream), low, C1, h, w, skip, C2, out);
    return cofi_launch_status();
}
