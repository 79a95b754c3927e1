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
    const float *scsh;          // optional: FINALIZED per-channel scale | shift, (frames, 2, C) (cofi_norm_finalize): consumers
                                // then read 2 C floats instead of folding the table
};

// sstat[2g] = mean, sstat[2g+1] = rstd for g < groups.  NT = threads of the workgroup (all must call),
// dred = 4 * NT doubles of LDS scratch.  Ends on a barrier (sstat visible to every thread).
//
// The fold is bound by L2 round trips, not by arithmetic: a thread reads 16 bytes (two table columns of one slab) per load and
// issues up to 16 independent loads before the first wait, so a 320-slab x 32-group table (80 KB, the largest GroupNorm table of
// a KITTI frame) takes 256 threads two batches = two round trips.  tcols is even (host-checked: a power of two >= 2).
template <int NT, int UNR = 16>   // UNR = loads in flight per thread (16 bytes each): registers vs round trips
__device__ __forceinline__ void fold_stat_table(const float *part, int nslab, int tcols, int groups, double count, float eps,
                                                double *dred, float *sstat, int tstride = 0) {
    // tstride: columns per table row when `part` points at a column sub-range of a wider table (0: tcols)
    if (tstride == 0) tstride = tcols;
    const int epg = tcols / groups;   // table columns per group
    const int tid = threadIdx.x;
    const int hc = tcols >> 1;        // column pairs (float4 chunks) per slab
    for (int c0 = 0; c0 < hc; c0 += NT) {
        const int cw = min(hc - c0, NT);      // power-of-two tcols: cw divides NT or equals it
        const int PH = NT / cw;               // slab phases
        const int cp = c0 + tid % cw, ph = tid / cw;
        double s0 = 0.0, q0 = 0.0, s1 = 0.0, q1 = 0.0;
        for (int b0 = ph; b0 < nslab; b0 += UNR * PH) {
            f32x4 t[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int b = b0 + u * PH;
                t[u] = *reinterpret_cast<const f32x4 *>(part + ((size_t)(b < nslab ? b : ph) * tstride + 2 * cp) * 2);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u)
                if (b0 + u * PH < nslab) {
                    s0 += (double)t[u][0]; q0 += (double)t[u][1];
                    s1 += (double)t[u][2]; q1 += (double)t[u][3];
                }
        }
        __syncthreads();   // dred reuse between passes / by the caller
        dred[4 * tid] = s0; dred[4 * tid + 1] = q0; dred[4 * tid + 2] = s1; dred[4 * tid + 3] = q1;
        __syncthreads();
        // table columns 2 c0 .. 2 (c0 + cw) of every phase are in dred: entry (phase p, column pair i) = dred[4 (p cw + i)]
        const int col0 = 2 * c0, ncol = 2 * cw;
        if (epg >= 2) {   // a group = epg / 2 adjacent column pairs, all inside this pass
            const int g0 = col0 / epg, ng = ncol / epg, ppg = epg >> 1;
            for (int g = tid; g < ng; g += NT) {
                double ts = 0.0, tq = 0.0;
                for (int p = 0; p < PH; ++p)
                    for (int i = 0; i < ppg; ++i) {
                        const double *e = dred + 4 * (p * cw + g * ppg + i);
                        ts += e[0] + e[2];
                        tq += e[1] + e[3];
                    }
                const double mean = ts / count;
                double var = tq / count - mean * mean;
                if (var < 0.0) var = 0.0;
                sstat[2 * (g0 + g)] = (float)mean;
                sstat[2 * (g0 + g) + 1] = (float)(1.0 / sqrt(var + (double)eps));
            }
        } else {          // one table column per group (InstanceNorm, or GroupNorm with a group-wide table)
            for (int c = tid; c < ncol; c += NT) {
                double ts = 0.0, tq = 0.0;
                for (int p = 0; p < PH; ++p) {
                    const double *e = dred + 4 * (p * cw + (c >> 1)) + 2 * (c & 1);
                    ts += e[0];
                    tq += e[1];
                }
                const double mean = ts / count;
                double var = tq / count - mean * mean;
                if (var < 0.0) var = 0.0;
                sstat[2 * (col0 + c)] = (float)mean;
                sstat[2 * (col0 + c) + 1] = (float)(1.0 / sqrt(var + (double)eps));
            }
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
    if ((cpg % w) || (tcols & (tcols - 1)) || tcols < 2 || (G & (G - 1)) || C > max_channels) return COFI_EUNSUPPORTED;
    // the channel -> group map of the consumers is a shift (cshift): channels per group must be a power of two (C = 96, G = 32 would
    // otherwise pass every check above and normalise with the wrong group); and a group's table columns must fit one fold pass of
    // the smallest consumer workgroup (fold_stat_table: 2 * 256 columns per pass)
    if ((cpg & (cpg - 1)) || tcols / G > 512) return COFI_EUNSUPPORTED;
    // slabs (64 rows unless the producer says otherwise) must not straddle frames (stack mode)
    const int sr = d->slab_rows > 0 ? d->slab_rows : 64;
    if (frames > 1 && (rows_per_frame % sr)) return COFI_EINVAL;
    if (d->nslab / frames != (rows_per_frame + sr - 1) / sr) return COFI_EINVAL;
    int cs = 0;
    while ((1 << cs) < cpg) ++cs;
    out->part = d->partials; out->gamma = d->gamma; out->beta = d->beta;
    out->nslab = d->nslab / frames; out->tcols = tcols; out->groups = G; out->C = C; out->cshift = cs;
    out->eps = d->eps; out->slope = d->slope;
    out->count = (double)rows_per_frame * cpg;
    out->scsh = d->scale_shift;
    if (out->scsh && ((uintptr_t)out->scsh & 15)) return COFI_EINVAL;
    return 0;
}
