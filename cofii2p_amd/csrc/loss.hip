// Row f3, first part: the three training losses of the reference (model/loss.py) with their gradients w.r.t. the network outputs.
//   desc_loss         (loss.py:69-93)  coarse descriptor loss: dists = 1 - <img, pc>, weighted log-sum-exp rows / columns, softplus
//   fine_circle_loss  (loss.py:9-51)   circle loss over the 16 pixels of a 4x4 patch against the point descriptor (cosine similarity)
//   overlap_loss      (loss.py:53-60)  binary cross entropy of the in-picture / out-of-picture super-point scores
// Sizes are tiny (num_kpt = 64 / 32 key points, 128 / 64 channels): latency-bound, one small launch per step, no MFMA.  Every reduction
// has a fixed order (wave butterflies, ascending strides): bit-reproducible.  The weights the reference detaches (pos_weight,
// neg_weight, ap, an) are treated as constants in the gradients, exactly as autograd does.
#include "common.h"

namespace {

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }   // F.softplus, threshold 20
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// ---------------------------------------------------------------------------------------- desc_loss
struct DescArgs {
    const float *img, *pc;   // (C, K) rows of ldi / ldp floats: column i = key point i
    const float *mask;       // (K, K) 1 = corresponding pair
    float *dists;            // (K, K)
    float *lse;              // (4, K): lse_pos_row, lse_neg_row, lse_pos_col, lse_neg_col
    float *loss;             // (1)
    int C, K, ldi, ldp;
    float pos_margin, neg_margin, log_scale;
};

// the two exponents of a pair (loss.py:75-86): z_pos = s (pos - pm) max(0, pos - pm), z_neg = s (nm - neg) max(0, nm - neg),
// pos = d - 1e5 (1 - mask), neg = d + 1e5 mask - written exactly as the reference writes them (a non-corresponding pair contributes
// z_pos = s * (-1e5) * 0 = -0, i.e. exp = 1, not 0: kept)
__device__ __forceinline__ void desc_z(const DescArgs &a, float d, float m, float &zp, float &zn, float &wp, float &wn) {
    const float pos = d - 1e5f * (1.f - m), neg = d + 1e5f * m;
    wp = fmaxf(0.f, pos - a.pos_margin);
    wn = fmaxf(0.f, a.neg_margin - neg);
    zp = a.log_scale * (pos - a.pos_margin) * wp;
    zn = a.log_scale * (a.neg_margin - neg) * wn;
}

__global__ void desc_dists_kernel(DescArgs a) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.K * a.K) return;
    const int i = e / a.K, j = e - i * a.K;
    float s = 0.f;
    for (int c = 0; c < a.C; ++c) s += a.img[(size_t)c * a.ldi + i] * a.pc[(size_t)c * a.ldp + j];
    a.dists[e] = 1.f - s;
}

// one wave per row (blockIdx.y == 0) or column (== 1): max and sum-of-exp of both exponent families -> lse
__global__ __launch_bounds__(64) void desc_lse_kernel(DescArgs a) {
    const int r = blockIdx.x, col = blockIdx.y, lane = threadIdx.x;
    float mp = -INFINITY, mn = -INFINITY;
    for (int t = lane; t < a.K; t += 64) {
        const int e = col ? t * a.K + r : r * a.K + t;
        float zp, zn, wp, wn;
        desc_z(a, a.dists[e], a.mask[e], zp, zn, wp, wn);
        mp = fmaxf(mp, zp);
        mn = fmaxf(mn, zn);
    }
    mp = wave_max(mp);
    mn = wave_max(mn);
    float sp = 0.f, sn = 0.f;
    for (int t = lane; t < a.K; t += 64) {
        const int e = col ? t * a.K + r : r * a.K + t;
        float zp, zn, wp, wn;
        desc_z(a, a.dists[e], a.mask[e], zp, zn, wp, wn);
        sp += expf(zp - mp);
        sn += expf(zn - mn);
    }
    sp = wave_sum(sp);
    sn = wave_sum(sn);
    if (lane == 0) {
        a.lse[(2 * col + 0) * a.K + r] = mp + logf(sp);
        a.lse[(2 * col + 1) * a.K + r] = mn + logf(sn);
    }
}

__global__ __launch_bounds__(64) void desc_final_kernel(DescArgs a) {   // loss = mean_i (softplus(row_i) + softplus(col_i)) / s
    float s = 0.f;
    for (int i = threadIdx.x; i < a.K; i += 64)
        s += softplus_f(a.lse[i] + a.lse[a.K + i]) / a.log_scale + softplus_f(a.lse[2 * a.K + i] + a.lse[3 * a.K + i]) / a.log_scale;
    s = wave_sum(s);
    if (threadIdx.x == 0) a.loss[0] = s / (float)a.K;
}

// dL/d dists[i,j] = g / K * ( sigma(row_i) (softmax_pos_row w_pos - softmax_neg_row w_neg) + sigma(col_j) (... the column softmaxes) )
__global__ void desc_grad_dists_kernel(DescArgs a, const float *gout, float *G) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.K * a.K) return;
    const int i = e / a.K, j = e - i * a.K;
    float zp, zn, wp, wn;
    desc_z(a, a.dists[e], a.mask[e], zp, zn, wp, wn);
    const float si = sigmoid_f(a.lse[i] + a.lse[a.K + i]), sj = sigmoid_f(a.lse[2 * a.K + j] + a.lse[3 * a.K + j]);
    const float row = expf(zp - a.lse[i]) * wp - expf(zn - a.lse[a.K + i]) * wn;
    const float colv = expf(zp - a.lse[2 * a.K + j]) * wp - expf(zn - a.lse[3 * a.K + j]) * wn;
    G[e] = gout[0] / (float)a.K * (si * row + sj * colv);
}

// d img[c,i] = - sum_j G[i,j] pc[c,j];  d pc[c,j] = - sum_i G[i,j] img[c,i]      (dists = 1 - img^T pc)
__global__ void desc_grad_feats_kernel(DescArgs a, const float *G, float *gimg, float *gpc, int ldgi, int ldgp) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.C * a.K) return;
    const int c = e / a.K, k = e - c * a.K;
    float si = 0.f, sp = 0.f;
    for (int t = 0; t < a.K; ++t) {
        si += G[(size_t)k * a.K + t] * a.pc[(size_t)c * a.ldp + t];
        sp += G[(size_t)t * a.K + k] * a.img[(size_t)c * a.ldi + t];
    }
    if (gimg) gimg[(size_t)c * ldgi + k] = -si;
    if (gpc) gpc[(size_t)c * ldgp + k] = -sp;
}

// ---------------------------------------------------------------------------------------- fine_circle_loss
// one wave per key point: lane = (pixel p = lane & 15, channel quarter = lane >> 4), as cofi_fine_match
struct CircleArgs {
    const float *patches;   // (K, C, 16)
    const float *pc;        // (K, C) rows of ldp
    const int64_t *rel;     // (K) index of the true pixel
    float *per_kpt;         // (K) log(1 + loss_n loss_p)
    float *gpatches, *gpc;  // optional gradients, same layouts (gpc rows of ldg)
    const float *gout;
    int K, C, ldp, ldg;
    float m, gamma;
};

__global__ __launch_bounds__(64) void circle_kernel(CircleArgs a) {
    const int k = blockIdx.x, lane = threadIdx.x, p = lane & 15, part = lane >> 4;
    float dot = 0.f, nn = 0.f, pp = 0.f;
    for (int c = part; c < a.C; c += 4) {
        const float pv = a.patches[((size_t)k * a.C + c) * 16 + p], fv = a.pc[(size_t)k * a.ldp + c];
        dot += pv * fv;
        nn += pv * pv;
        pp += fv * fv;
    }
    dot += __shfl_xor(dot, 16, 64); dot += __shfl_xor(dot, 32, 64);
    nn += __shfl_xor(nn, 16, 64);   nn += __shfl_xor(nn, 32, 64);
    pp += __shfl_xor(pp, 16, 64);   pp += __shfl_xor(pp, 32, 64);
    const float nx = fmaxf(sqrtf(nn), 1e-8f), ny = fmaxf(sqrtf(pp), 1e-8f);
    const float dist = dot / (nx * ny);                       // torch.cosine_similarity, eps 1e-8
    const float pos = (p == (int)a.rel[k]) ? 1.f : 0.f, neg = 1.f - pos;
    const float sp = dist * pos, sn = dist * neg;
    const float ap = fmaxf(-sp + pos + pos * a.m, 0.f), an = fmaxf(sn + neg * a.m, 0.f);   // detached in the reference
    const float lp = -ap * (sp - pos * (1.f - a.m)) * a.gamma, ln = an * (sn - neg * a.m) * a.gamma;
    float ep = expf(lp) * pos, en = expf(ln) * neg;
    float loss_p = ep, loss_n = en;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {   // over the 16 pixels (the 4 channel quarters hold copies)
        loss_p += __shfl_xor(loss_p, o, 64);
        loss_n += __shfl_xor(loss_n, o, 64);
    }
    if (lane == 0) a.per_kpt[k] = logf(1.f + loss_n * loss_p);
    if (!a.gpatches && !a.gpc) return;
    // d L / d dist[p]: L = mean_k log(1 + loss_n loss_p)
    const float common = a.gout[0] / (float)a.K / (1.f + loss_n * loss_p);
    const float gd = common * (pos * loss_n * ep * (-ap * a.gamma) + neg * loss_p * en * (an * a.gamma));
    // cosine: d dist / d x = y / (nx ny) - dist x / nx^2 (norms above eps), same for y
    const float inx = 1.f / nx, iny = 1.f / ny;
    for (int c = part; c < a.C; c += 4) {
        const float pv = a.patches[((size_t)k * a.C + c) * 16 + p], fv = a.pc[(size_t)k * a.ldp + c];
        if (a.gpatches) a.gpatches[((size_t)k * a.C + c) * 16 + p] = gd * (fv * inx * iny - dist * pv * inx * inx);
        if (a.gpc) {
            float t = gd * (pv * inx * iny - dist * fv * iny * iny);
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);   // sum over the 16 pixels
            if (p == 0) a.gpc[(size_t)k * a.ldg + c] = t;
        }
    }
}

__global__ __launch_bounds__(64) void mean_kernel(const float *x, int n, float *out) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) s += x[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[0] = s / (float)n;
}

// ---------------------------------------------------------------------------------------- overlap_loss (BCELoss, mean)
__global__ __launch_bounds__(64) void bce_kernel(const float *in_s, int n_in, const float *out_s, int n_out, float *loss, const float *gout,
                                                 float *g_in, float *g_out) {
    const int n = n_in + n_out;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) {
        const bool inl = i < n_in;
        const float x = inl ? in_s[i] : out_s[i - n_in];
        // torch clamps the logs at -100
        s += inl ? -fmaxf(logf(x), -100.f) : -fmaxf(log1pf(-x), -100.f);
        if (gout) {   // binary_cross_entropy_backward: (x - y) / max((1 - x) x, 1e-12) / n
            const float g = gout[0] * (x - (inl ? 1.f : 0.f)) / fmaxf((1.f - x) * x, 1e-12f) / (float)n;
            if (inl) { if (g_in) g_in[i] = g; } else if (g_out) g_out[i - n_in] = g;
        }
    }
    s = wave_sum(s);
    if (threadIdx.x == 0 && loss) loss[0] = s / (float)n;
}

}  // namespace

extern "C" size_t cofi_desc_loss_workspace(int K) { return K > 0 ? (size_t)(4 * K + (size_t)K * K) * sizeof(float) : 0; }

extern "C" int cofi_desc_loss(const float *img, int ldi, const float *pc, int ldp, const float *mask, int C, int K, float pos_margin,
                              float neg_margin, float log_scale, float *loss, float *dists, const float *grad_out, float *grad_img, int ldgi,
                              float *grad_pc, int ldgp, void *ws, size_t ws_bytes, cofi_stream_t stream) {
    if (!img || !pc || !mask || !loss || !dists || C <= 0 || K <= 0 || ldi < K || ldp < K || !(log_scale > 0.f)) return COFI_EINVAL;
    if ((grad_img || grad_pc) && !grad_out) return COFI_EINVAL;
    if ((grad_img && ldgi < K) || (grad_pc && ldgp < K)) return COFI_EINVAL;
    if (!ws || ws_bytes < cofi_desc_loss_workspace(K)) return COFI_EWORKSPACE;
    hipStream_t s = cofi_s(stream);
    float *lse = (float *)ws, *G = lse + 4 * K;
    DescArgs a{img, pc, mask, dists, lse, loss, C, K, ldi, ldp, pos_margin, neg_margin, log_scale};
    hipLaunchKernelGGL(desc_dists_kernel, dim3(cofi_cdiv((long)K * K, 256)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(desc_lse_kernel, dim3(K, 2), dim3(64), 0, s, a);
    hipLaunchKernelGGL(desc_final_kernel, dim3(1), dim3(64), 0, s, a);
    if (grad_img || grad_pc) {
        hipLaunchKernelGGL(desc_grad_dists_kernel, dim3(cofi_cdiv((long)K * K, 256)), dim3(256), 0, s, a, grad_out, G);
        hipLaunchKernelGGL(desc_grad_feats_kernel, dim3(cofi_cdiv((long)C * K, 256)), dim3(256), 0, s, a, (const float *)G, grad_img, grad_pc, ldgi, ldgp);
    }
    return cofi_launch_status();
}

extern "C" int cofi_fine_circle_loss(const float *patches, const float *pc, int ldp, const int64_t *relative_index, int K, int C, float m,
                                     float gamma, float *loss, float *per_kpt, const float *grad_out, float *grad_patches, float *grad_pc,
                                     int ldg, cofi_stream_t stream) {
    if (!patches || !pc || !relative_index || !loss || !per_kpt || K <= 0 || C <= 0 || ldp < C) return COFI_EINVAL;
    if ((grad_patches || grad_pc) && !grad_out) return COFI_EINVAL;
    if (grad_pc && ldg < C) return COFI_EINVAL;
    hipStream_t s = cofi_s(stream);
    CircleArgs a{patches, pc, relative_index, per_kpt, grad_patches, grad_pc, grad_out, K, C, ldp, ldg, m, gamma};
    hipLaunchKernelGGL(circle_kernel, dim3(K), dim3(64), 0, s, a);
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(64), 0, s, (const float *)per_kpt, K, loss);
    return cofi_launch_status();
}

extern "C" int cofi_overlap_loss(const float *inline_score, int n_in, const float *outline_score, int n_out, float *loss, const float *grad_out,
                                 float *grad_in, float *grad_outline, cofi_stream_t stream) {
    if (!inline_score || !outline_score || n_in < 0 || n_out < 0 || n_in + n_out <= 0 || (!loss && !grad_out)) return COFI_EINVAL;
    hipLaunchKernelGGL(bce_kernel, dim3(1), dim3(64), 0, cofi_s(stream), inline_score, n_in, outline_score, n_out, loss, grad_out, grad_in,
                       grad_outline);
    return cofi_launch_status();
}
