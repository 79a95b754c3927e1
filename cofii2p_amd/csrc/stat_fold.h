// Statistics partials left by a GEMM / convolution epilogue and their in-kernel fold into GroupNorm / InstanceNorm
// statistics (model/kpconv/modules.py:45-48, model/imagenet.py:123, model/network.py:42-43).
//
// A producer writes, per 64-row slab and per TABLE COLUMN (= `width` adjacent output columns, a power of two that
// divides the GroupNorm group width), {sum, sum of squares} of its output: part (frames * nslab, tcols, 2) fp32.
// Every consumer workgroup folds the table of its frame itself (fixed order, fp64: all workgroups see bit-identical
// statistics) instead of waiting for a separate finalize launch: a dependent launch costs >= 1.5 us at one frame, the
// fold is one or two L2 round trips that overlap with the consumer's first operand loads.
#pragma once
#include "common.h"

// device-side view of a pending normalisation  y -> leaky( gn(y) * gamma + beta, slope )
struct NormSrc {
    const float *part;          // (frames * nslab, tcols, 2); nullptr = none
    const float *gamma, *beta;  // (C) or nullptr (affine-less InstanceNorm)
    int nslab;                  // slabs PER FRAME
    int tcols;                  // table columns = C / width (power of two)
    int groups;                 // statistics groups (power of two, <= tcols)
    int C;                      // channels of the activation
    int cshift;                 // log2(C / groups)
    float eps, slope;
    double count;               // rows per frame * C / groups
};

// sstat[2g] = mean, sstat[2g+1] = rstd for g < groups.  NT = threads of the workgroup (all must call),
// dred = 2 * NT doubles of LDS scratch.  Ends on a barrier (sstat visible to every thread).
template <int NT>
__device__ __forceinline__ void fold_stat_table(const float *part, int nslab, int tcols, int groups, double count, float eps,
                                                double *dred, float *sstat) {
    const int epg = tcols / groups;   // table columns per group
    const int tid = threadIdx.x;
    for (int c0 = 0; c0 < tcols; c0 += NT) {
        const int cw = min(tcols - c0, NT);   // power-of-two tcols: cw divides NT or equals it
        const int PH = NT / cw;               // slab phases
        const int c = c0 + tid % cw, ph = tid / cw;
        double s = 0.0, q = 0.0;
        for (int b0 = ph; b0 < nslab; b0 += 8 * PH) {   // 8 independent loads per round: the fold is bound by L2 round trips
            float2 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + u * PH;
                t[u] = *reinterpret_cast<const float2 *>(part + ((size_t)(b < nslab ? b : ph) * tcols + c) * 2);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (b0 + u * PH < nslab) {
                    s += (double)t[u].x;
                    q += (double)t[u].y;
                }
        }
        __syncthreads();   // dred reuse between passes / by the caller
        dred[2 * tid] = s;
        dred[2 * tid + 1] = q;
        __syncthreads();
        const int g0 = c0 / epg, ng = cw / epg;   // groups fully inside this pass (epg <= cw: host-checked)
        for (int g = tid; g < ng; g += NT) {
            double ts = 0.0, tq = 0.0;
            for (int p = 0; p < PH; ++p)
                for (int i = 0; i < epg; ++i) {
                    const int t = p * cw + g * epg + i;
                    ts += dred[2 * t];
                    tq += dred[2 * t + 1];
                }
            const double mean = ts / count;
            double var = tq / count - mean * mean;
            if (var < 0.0) var = 0.0;
            sstat[2 * (g0 + g)] = (float)mean;
            sstat[2 * (g0 + g) + 1] = (float)(1.0 / sqrt(var + (double)eps));
        }
    }
    __syncthreads();
}

// per-channel scale / shift of the pending normalisation: value = y * sc[c] + sh[c] (then the LeakyReLU).  Same
// expression, in the same order, as the stand-alone apply kernel (norm.hip): both paths produce identical bits.
template <int NT>
__device__ __forceinline__ void norm_scale_shift(const NormSrc &n, const float *sstat, float *sc, float *sh) {
    for (int c = threadIdx.x; c < n.C; c += NT) {
        const int g = c >> n.cshift;
        const float mean = sstat[2 * g], rstd = sstat[2 * g + 1];
        const float ga = n.gamma ? n.gamma[c] : 1.f, be = n.gamma ? n.beta[c] : 0.f;
        sc[c] = rstd * ga;
        sh[c] = be - mean * rstd * ga;
    }
}

// Host side: C descriptor (include/cofi_hip.h) -> NormSrc of one launch.  Returns 0 or a COFI_E* code.
// rows_per_frame = rows of the normalised activation per frame.
static inline int make_norm_src(const cofi_norm_desc_t *d, int rows_per_frame, int frames, int max_channels, NormSrc *out) {
    if (!d || !d->partials || d->nslab <= 0 || d->width <= 0 || d->channels <= 0 || d->groups <= 0 || frames <= 0) return COFI_EINVAL;
    if ((d->gamma == nullptr) != (d->beta == nullptr)) return COFI_EINVAL;
    const int C = d->channels, w = d->width, G = d->groups;
    if ((C % w) || (C % G) || (d->nslab % frames)) return COFI_EINVAL;
    const int tcols = C / w, cpg = C / G;
    if ((cpg % w) || (tcols & (tcols - 1)) || (G & (G - 1)) || C > max_channels) return COFI_EUNSUPPORTED;
    // slabs are 64 rows and must not straddle frames (stack mode)
    if (frames > 1 && (rows_per_frame % 64)) return COFI_EINVAL;
    if (d->nslab / frames != (rows_per_frame + 63) / 64) return COFI_EINVAL;
    int cs = 0;
    while ((1 << cs) < cpg) ++cs;
    out->part = d->partials; out->gamma = d->gamma; out->beta = d->beta;
    out->nslab = d->nslab / frames; out->tcols = tcols; out->groups = G; out->C = C; out->cshift = cs;
    out->eps = d->eps; out->slope = d->slope;
    out->count = (double)rows_per_frame * cpg;
    return 0;
}
