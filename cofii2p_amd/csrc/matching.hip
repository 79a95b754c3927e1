// Coarse super-point -> super-pixel matching and 4x4 patch fine matching with no host round trip.
// Reference: model/network.py:145-161 (threshold loop), :167-187 (`fine_process`), :206-226
// (`extract_patch`), evaluation/eval_all.py:99-105 (fine matching in the caller).
// The match count stays on the device (count_dev); downstream kernels are launched at capacity and
// read it, so the only host synchronisation of a test-mode forward is the final read of the count.
#include "common.h"
#include "knn_common.h"

namespace {

// pix[n] = argmin_p (1 - sim[n,p]), first index on ties (torch.argmin).  One wave per row.
__global__ __launch_bounds__(256) void row_argmin_1m_kernel(const float *sim, int lds, int N, int P, int32_t *pix) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int lane = threadIdx.x & 63;
    float best = INFINITY;
    int bi = 0x7fffffff;
    for (int p = lane; p < P; p += 64) {
        const float d = 1.0f - sim[(size_t)n * lds + p];
        if (d < best) { best = d; bi = p; }  // strict: keeps the lowest p of this lane
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) pix[n] = bi;
}

struct SelArgs {
    const float *score;
    const int32_t *pix;
    int32_t *sel, *count;
    float *xy;
    int N, W8, H8, n_thr, min_matches, xmax, ymax;
    float thr[64];
};

// one workgroup per frame (stack mode: frame f reads score / pix rows [f N, (f + 1) N) and owns sel[f], xy[f] (2, N), count[f]):
// threshold search + ordered compaction (ascending point index).  256 threads: a 1024-thread workgroup needs a whole CU's wave slots
// and waits for one to drain while other submissions' kernels fill the chip (86 us on average at batch 16, round 4 - for 5 us of work)
constexpr int SEL_NT = 256, SEL_NW = SEL_NT / 64;
__global__ __launch_bounds__(SEL_NT) void select_matches_kernel(SelArgs a) {
    __shared__ int s_cnt[SEL_NW];
    __shared__ int s_thr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f = blockIdx.x;
    a.score += (size_t)f * a.N; a.pix += (size_t)f * a.N; a.sel += (size_t)f * a.N; a.xy += (size_t)f * 2 * a.N; a.count += 2 * f;
    // pass 1: find the first threshold with enough survivors
    if (tid == 0) s_thr = -1;
    __syncthreads();
    for (int t = 0; t < a.n_thr; ++t) {
        int c = 0;
        for (int n = tid; n < a.N; n += SEL_NT) {
            const int p = a.pix[n];
            const int x = p % a.W8, y = p / a.W8;
            const bool ok = a.score[n] >= a.thr[t] && x >= 2 && x <= a.xmax && y >= 2 && y <= a.ymax;
            c += ok ? 1 : 0;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if (lane == 0) s_cnt[wave] = c;
        __syncthreads();
        int tot = 0;
        for (int w = 0; w < SEL_NW; ++w) tot += s_cnt[w];
        __syncthreads();
        if (tot >= a.min_matches) {
            if (tid == 0) s_thr = t;
            break;  // uniform: every thread computed the same tot
        }
    }
    __syncthreads();
    const int tsel = s_thr;
    if (tsel < 0) {
        if (tid == 0) { a.count[0] = 0; a.count[1] = -1; }
        return;
    }
    const float thr = a.thr[tsel];
    // pass 2: ordered compaction in chunks of SEL_NT points
    int base = 0;
    for (int n0 = 0; n0 < a.N; n0 += SEL_NT) {
        const int n = n0 + tid;
        bool ok = false;
        int x = 0, y = 0;
        if (n < a.N) {
            const int p = a.pix[n];
            x = p % a.W8; y = p / a.W8;
            ok = a.score[n] >= thr && x >= 2 && x <= a.xmax && y >= 2 && y <= a.ymax;
        }
        const unsigned long long m = __ballot(ok);
        if (lane == 0) s_cnt[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += s_cnt[w];
        int tot = 0;
        for (int w = 0; w < SEL_NW; ++w) tot += s_cnt[w];
        if (ok) {
            const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
            a.sel[pos] = n;
            a.xy[pos] = (float)x;
            a.xy[a.N + pos] = (float)y;
        }
        base += tot;
        __syncthreads();
    }
    if (tid == 0) { a.count[0] = base; a.count[1] = tsel; }
}

__global__ void gather_points_sel_kernel(const float *pts, const int32_t *sel, const int32_t *count_dev, int cap, float *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = min(*count_dev, cap);
    if (i >= n * 3) return;
    out[i] = pts[3 * (size_t)sel[i / 3] + (i % 3)];
}

// patches[i, c, r*4 + w] = fmap[(4*y - 2 + r) * W2 + 4*x - 2 + w, c]: fmap row (y*W2 + x) holds the C channels of a pixel
__global__ void extract_patches_nhwc_kernel(const float *fmap, int ldf, int C, int H2, int W2, const float *xy, int ldxy, float cscale,
                                            const int32_t *count_dev, int cap, float *patches) {
    const int i = blockIdx.x;
    if (i >= min(*count_dev, cap)) return;
    const int left = (int)floorf(xy[i] * cscale - 2.0f), top = (int)floorf(xy[ldxy + i] * cscale - 2.0f);
    for (int e = threadIdx.x; e < C * 16; e += blockDim.x) {
        const int c = e % C, t = e / C, r = t >> 2, w = t & 3;  // lanes sweep channels: contiguous reads
        const int yy = top + r, xx = left + w;
        float v = 0.f;
        if (yy >= 0 && yy < H2 && xx >= 0 && xx < W2) v = fmap[((size_t)yy * W2 + xx) * ldf + c];
        patches[((size_t)i * C + c) * 16 + t] = v;
    }
}

__global__ void gather_rows_sel_kernel(const float *x, int ldx, int C, const int32_t *row_idx, const int32_t *count_dev, int cap,
                                       float *out, int ldo) {
    const int i = blockIdx.x;
    if (i >= min(*count_dev, cap)) return;
    const size_t r = (size_t)row_idx[i];
    for (int c = threadIdx.x; c < C; c += blockDim.x) out[(size_t)i * ldo + c] = x[r * ldx + c];
}

// one wave per match: 16 cosine similarities (eps 1e-8, torch.cosine_similarity), argmax, fine_xy
__global__ __launch_bounds__(64) void fine_match_kernel(const float *patches, const float *pcf, int ldp, int C, const float *xy,
                                                        int ldxy, float cscale, const int32_t *count_dev, int cap, float *fine_xy,
                                                        int32_t *best) {
    const int i = blockIdx.x;
    if (i >= min(*count_dev, cap)) return;
    const int lane = threadIdx.x;
    const int pxl = lane & 15, part = lane >> 4;  // 4 lanes groups split the channels of one pixel
    float dot = 0.f, nn = 0.f, pp = 0.f;
    for (int c = part; c < C; c += 4) {
        const float pv = patches[((size_t)i * C + c) * 16 + pxl];
        const float fv = pcf[(size_t)i * ldp + c];
        dot += pv * fv;
        nn += pv * pv;
        pp += fv * fv;
    }
    dot += __shfl_xor(dot, 16, 64); dot += __shfl_xor(dot, 32, 64);
    nn += __shfl_xor(nn, 16, 64); nn += __shfl_xor(nn, 32, 64);
    pp += __shfl_xor(pp, 16, 64); pp += __shfl_xor(pp, 32, 64);
    float sim = dot / (fmaxf(sqrtf(nn), 1e-8f) * fmaxf(sqrtf(pp), 1e-8f));
    int bi = pxl;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        const float os = __shfl_xor(sim, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (os > sim || (os == sim && oi < bi)) { sim = os; bi = oi; }
    }
    if (lane == 0) {
        best[i] = bi;
        // eval_all.py:103-105 — x receives idx // 4 and y receives idx % 4 (kept as in the reference)
        fine_xy[i] = (xy[i] * cscale - 2.0f) + (float)(bi / 4);
        fine_xy[cap + i] = (xy[ldxy + i] * cscale - 2.0f) + (float)(bi % 4);
    }
}

// Everything of a test-mode forward that follows the match selection (network.py:153-161 + eval_all.py:99-105), one workgroup per
// accepted match i (count read on the device): the coarse point, its nearest stage-1 node (point2node, network.py:250-264: canonical
// distance, lowest index on ties - the (distance, index) key of nearest_kernel), that node's fine descriptor, the 4 x 4 patch of the
// fine image map under the coarse pixel and the fine matching of the two.  Same per-element arithmetic, in the same order, as the five
// stand-alone kernels above / in knn.hip: bit-identical outputs, four launches fewer on a frame's chain.
struct FinishArgs {
    const float *pts4, *pts1, *fmap, *xy, *fpc;
    const int32_t *sel, *count;
    float *coarse_pts, *patches, *fine_pc, *fine_xy;
    int32_t *best;
    int N1, cap, ldf, C, H2, W2, ldxy, ldfpc, ldo;
    float cscale;
    int N4;   // stage-4 points per frame (stack mode: frame f = blockIdx.x / cap; every per-frame array advances by its own frame size)
};
constexpr int FIN_MAXC = 128;

__global__ __launch_bounds__(256) void match_finish_kernel(FinishArgs a) {
    __shared__ u64 s_key[4];
    __shared__ float s_f[FIN_MAXC], s_p[FIN_MAXC * 16];
    const int f = blockIdx.x / a.cap, i = blockIdx.x - f * a.cap;
    a.count += 2 * f;
    if (i >= min(*a.count, a.cap)) return;
    a.pts4 += (size_t)f * a.N4 * 3; a.pts1 += (size_t)f * a.N1 * 3; a.sel += (size_t)f * a.cap; a.xy += (size_t)f * 2 * a.ldxy;
    a.fmap += (size_t)f * a.H2 * a.W2 * a.ldf; a.fpc += (size_t)f * a.N1 * a.ldfpc;
    a.coarse_pts += (size_t)f * a.cap * 3; a.patches += (size_t)f * a.cap * a.C * 16; a.fine_pc += (size_t)f * a.cap * a.ldo;
    a.fine_xy += (size_t)f * 2 * a.cap; a.best += (size_t)f * a.cap;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t n = (size_t)a.sel[i];
    const float qx = a.pts4[3 * n], qy = a.pts4[3 * n + 1], qz = a.pts4[3 * n + 2];
    if (tid < 3) a.coarse_pts[3 * (size_t)i + tid] = a.pts4[3 * n + tid];
    // nearest stage-1 node
    const float qq = canon_sqnorm(qx, qy, qz);
    u64 bestk = KEY_INF;
    for (int c0 = tid; c0 < a.N1; c0 += 1024) {
        float px[4], py[4], pz[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = min(c0 + 256 * u, a.N1 - 1);
            px[u] = a.pts1[3 * (size_t)c];
            py[u] = a.pts1[3 * (size_t)c + 1];
            pz[u] = a.pts1[3 * (size_t)c + 2];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + 256 * u;
            const float d = canon_dist(qx, qy, qz, qq, px[u], py[u], pz[u], canon_sqnorm(px[u], py[u], pz[u]));
            if (c < a.N1) bestk = umin64(bestk, ((u64)__float_as_uint(d) << 32) | (unsigned)c);
        }
    }
    bestk = umin64(bestk, lane_xor64<32>(bestk, lane));
    bestk = umin64(bestk, lane_xor64<16>(bestk, lane));
    bestk = umin64(bestk, lane_xor64<8>(bestk, lane));
    bestk = umin64(bestk, lane_xor64<4>(bestk, lane));
    bestk = umin64(bestk, lane_xor64<2>(bestk, lane));
    bestk = umin64(bestk, lane_xor64<1>(bestk, lane));
    if (lane == 0) s_key[wave] = bestk;
    // the patch does not depend on the node: gathered while the keys settle
    const float cx = a.xy[i], cy = a.xy[a.ldxy + i];
    const int left = (int)floorf(cx * a.cscale - 2.0f), top = (int)floorf(cy * a.cscale - 2.0f);
    for (int e = tid; e < a.C * 16; e += 256) {
        const int c = e % a.C, t = e / a.C, r = t >> 2, w = t & 3;  // lanes sweep channels: contiguous reads
        const int yy = top + r, xx = left + w;
        float v = 0.f;
        if (yy >= 0 && yy < a.H2 && xx >= 0 && xx < a.W2) v = a.fmap[((size_t)yy * a.W2 + xx) * a.ldf + c];
        a.patches[((size_t)i * a.C + c) * 16 + t] = v;
        s_p[c * 16 + t] = v;
    }
    __syncthreads();
    const size_t node = (size_t)(unsigned)(umin64(umin64(s_key[0], s_key[1]), umin64(s_key[2], s_key[3])) & 0xffffffffu);
    for (int c = tid; c < a.C; c += 256) {
        const float v = a.fpc[node * a.ldfpc + c];
        a.fine_pc[(size_t)i * a.ldo + c] = v;
        s_f[c] = v;
    }
    __syncthreads();
    if (wave != 0) return;
    // fine matching (fine_match_kernel): 16 cosine similarities, 4 lane groups split the channels of one pixel
    const int pxl = lane & 15, part = lane >> 4;
    float dot = 0.f, nn = 0.f, pp = 0.f;
    for (int c = part; c < a.C; c += 4) {
        const float pv = s_p[c * 16 + pxl];
        const float fv = s_f[c];
        dot += pv * fv;
        nn += pv * pv;
        pp += fv * fv;
    }
    dot += __shfl_xor(dot, 16, 64); dot += __shfl_xor(dot, 32, 64);
    nn += __shfl_xor(nn, 16, 64); nn += __shfl_xor(nn, 32, 64);
    pp += __shfl_xor(pp, 16, 64); pp += __shfl_xor(pp, 32, 64);
    float sim = dot / (fmaxf(sqrtf(nn), 1e-8f) * fmaxf(sqrtf(pp), 1e-8f));
    int bi = pxl;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        const float os = __shfl_xor(sim, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (os > sim || (os == sim && oi < bi)) { sim = os; bi = oi; }
    }
    if (lane == 0) {
        a.best[i] = bi;
        a.fine_xy[i] = (cx * a.cscale - 2.0f) + (float)(bi / 4);         // eval_all.py:103-105: x receives idx // 4, y idx % 4
        a.fine_xy[a.cap + i] = (cy * a.cscale - 2.0f) + (float)(bi % 4);
    }
}

}  // namespace

extern "C" int cofi_match_finish(const float *pts4, const float *pts1, int N1, const int32_t *sel, const int32_t *count_dev, int cap,
                                 const float *fmap, int ldf, int C, int H2, int W2, const float *coarse_xy, int ldxy, float center_scale,
                                 const float *fine_pc_all, int ldfpc, float *coarse_pts, float *patches, float *fine_pc, int ldo,
                                 float *fine_xy, int32_t *best, int N4, int frames, cofi_stream_t stream) {
    if (!pts4 || !pts1 || !sel || !count_dev || !fmap || !coarse_xy || !fine_pc_all || !coarse_pts || !patches || !fine_pc || !fine_xy || !best)
        return COFI_EINVAL;
    if (N1 <= 0 || cap <= 0 || C <= 0 || H2 <= 0 || W2 <= 0 || ldf < C || ldfpc < C || ldo < C || N4 <= 0 || frames <= 0) return COFI_EINVAL;
    if (C > FIN_MAXC) return COFI_EUNSUPPORTED;
    FinishArgs a{pts4, pts1, fmap, coarse_xy, fine_pc_all, sel, count_dev, coarse_pts, patches, fine_pc, fine_xy, best,
                 N1, cap, ldf, C, H2, W2, ldxy, ldfpc, ldo, center_scale, N4};
    hipLaunchKernelGGL(match_finish_kernel, dim3(cap * frames), dim3(256), 0, cofi_s(stream), a);
    return cofi_launch_status();
}

extern "C" int cofi_row_argmin_1m(const float *sim, int lds, int N, int P, int32_t *pix, cofi_stream_t stream) {
    if (!sim || !pix || N <= 0 || P <= 0 || lds < P) return COFI_EINVAL;
    hipLaunchKernelGGL(row_argmin_1m_kernel, dim3(cofi_cdiv(N, 4)), dim3(256), 0, cofi_s(stream), sim, lds, N, P, pix);
    return cofi_launch_status();
}

extern "C" int cofi_select_matches(const float *score, const int32_t *pix, int N, int W8, int H8, int x_max, int y_max, const float *thr_host,
                                   int n_thr, int min_matches, int32_t *sel, float *coarse_xy, int32_t *count_dev, int frames, cofi_stream_t stream) {
    if (!score || !pix || !thr_host || !sel || !coarse_xy || !count_dev || N <= 0 || W8 <= 0 || H8 <= 0 || n_thr <= 0 || n_thr > 64 || frames <= 0)
        return COFI_EINVAL;
    SelArgs a;
    a.score = score; a.pix = pix; a.sel = sel; a.count = count_dev; a.xy = coarse_xy;
    a.N = N; a.W8 = W8; a.H8 = H8; a.n_thr = n_thr; a.min_matches = min_matches; a.xmax = x_max; a.ymax = y_max;
    for (int i = 0; i < 64; ++i) a.thr[i] = i < n_thr ? thr_host[i] : 0.f;
    hipLaunchKernelGGL(select_matches_kernel, dim3(frames), dim3(SEL_NT), 0, cofi_s(stream), a);
    return cofi_launch_status();
}

extern "C" int cofi_gather_points_sel(const float *pts, const int32_t *sel, const int32_t *count_dev, int cap, float *out,
                                      cofi_stream_t stream) {
    if (!pts || !sel || !count_dev || !out || cap <= 0) return COFI_EINVAL;
    hipLaunchKernelGGL(gather_points_sel_kernel, dim3(cofi_cdiv(cap * 3, 256)), dim3(256), 0, cofi_s(stream), pts, sel, count_dev, cap,
                       out);
    return cofi_launch_status();
}

extern "C" int cofi_extract_patches_nhwc(const float *fmap, int ldf, int C, int H2, int W2, const float *coarse_xy, int ldxy,
                                         float center_scale, const int32_t *count_dev, int cap, float *patches, cofi_stream_t stream) {
    if (!fmap || !coarse_xy || !count_dev || !patches || C <= 0 || H2 <= 0 || W2 <= 0 || cap <= 0 || ldf < C) return COFI_EINVAL;
    hipLaunchKernelGGL(extract_patches_nhwc_kernel, dim3(cap), dim3(256), 0, cofi_s(stream), fmap, ldf, C, H2, W2, coarse_xy, ldxy,
                       center_scale, count_dev, cap, patches);
    return cofi_launch_status();
}

extern "C" int cofi_gather_rows_sel(const float *x, int ldx, int C, const int32_t *row_idx, const int32_t *count_dev, int cap,
                                    float *out, int ldo, cofi_stream_t stream) {
    if (!x || !row_idx || !count_dev || !out || C <= 0 || cap <= 0 || ldx < C || ldo < C) return COFI_EINVAL;
    hipLaunchKernelGGL(gather_rows_sel_kernel, dim3(cap), dim3(64), 0, cofi_s(stream), x, ldx, C, row_idx, count_dev, cap, out, ldo);
    return cofi_launch_status();
}

extern "C" int cofi_fine_match(const float *patches, const float *pc_feats, int ldp, int C, const float *coarse_xy, int ldxy,
                               float center_scale, const int32_t *count_dev, int cap, float *fine_xy, int32_t *best,
                               cofi_stream_t stream) {
    if (!patches || !pc_feats || !coarse_xy || !count_dev || !fine_xy || !best || C <= 0 || cap <= 0 || ldp < C) return COFI_EINVAL;
    hipLaunchKernelGGL(fine_match_kernel, dim3(cap), dim3(64), 0, cofi_s(stream), patches, pc_feats, ldp, C, coarse_xy, ldxy,
                       center_scale, count_dev, cap, fine_xy, best);
    return cofi_launch_status();
}
