/* CPU ORACLE — test infrastructure, NOT the product.
 *
 * Brute-force k-nearest-neighbour search with the reference's expansion-form distance
 * (model/kpconv/preprocess_data.py:109-143 `square_distance` + `knn`; twin at
 * model/network.py:228-264) and a DEFINED tie rule, so integer outputs can be compared
 * bit-for-bit with the HIP kernel `cofi_knn_topk` (cofii2p_amd/csrc/knn.hip).
 *
 * Canonical fp32 arithmetic (every operation rounded to fp32, no contraction except the two
 * explicit fmaf):
 *     dot = fmaf(qz, sz, fmaf(qy, sy, qx * sx))
 *     qq  = (qx*qx + qy*qy) + qz*qz          (likewise ss)
 *     d   = max(((-2 * dot) + qq) + ss, 1e-12f)
 * Order: ascending (d, support index) — ties broken by the LOWEST index.  torch.topk leaves tie
 * order unspecified, so against the reference itself only tie-aware set equality is claimed
 * (tests/test_oracle_golden.py).
 *
 * Build: see oracle/Makefile (-ffp-contract=off is load-bearing).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float d; int32_t i; } cand_t;

static inline int cand_less(cand_t a, cand_t b) { return a.d < b.d || (a.d == b.d && a.i < b.i); }

static inline float sqnorm3(const float *p) { return (p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]; }

float cofi_oracle_sqdist(const float *q, const float *s) {
    float dot = fmaf(q[2], s[2], fmaf(q[1], s[1], q[0] * s[0]));
    float d = ((-2.0f * dot) + sqnorm3(q)) + sqnorm3(s);
    return d < 1e-12f ? 1e-12f : d;
}

static void sift_down(cand_t *h, int n, int i) { /* max-heap on (d,i) */
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && cand_less(h[m], h[l])) m = l;
        if (r < n && cand_less(h[m], h[r])) m = r;
        if (m == i) return;
        cand_t t = h[i]; h[i] = h[m]; h[m] = t; i = m;
    }
}

static int cmp_cand(const void *a, const void *b) {
    cand_t x = *(const cand_t *)a, y = *(const cand_t *)b;
    return cand_less(x, y) ? -1 : (cand_less(y, x) ? 1 : 0);
}

/* support (S,3), query (Q,3) row-major; out_idx (Q,k) int64; out_dist (Q,k) or NULL.
 * If S < k the tail is padded with index S (the "shadow" index) and +inf. */
int cofi_oracle_knn(const float *support, int S, const float *query, int Q, int k, int64_t *out_idx, float *out_dist) {
    if (S < 0 || Q < 0 || k <= 0) return 1;
#pragma omp parallel
    {
        cand_t *heap = (cand_t *)malloc(sizeof(cand_t) * (size_t)k);
        float *ss = NULL;
#pragma omp for schedule(dynamic, 16)
        for (int q = 0; q < Q; ++q) {
            const float *qp = query + 3 * (size_t)q;
            float qq = sqnorm3(qp);
            int n = 0;
            for (int s = 0; s < S; ++s) {
                const float *sp = support + 3 * (size_t)s;
                float dot = fmaf(qp[2], sp[2], fmaf(qp[1], sp[1], qp[0] * sp[0]));
                float d = ((-2.0f * dot) + qq) + sqnorm3(sp);
                if (d < 1e-12f) d = 1e-12f;
                cand_t c = {d, s};
                if (n < k) {
                    heap[n++] = c;
                    if (n == k) for (int i = k / 2 - 1; i >= 0; --i) sift_down(heap, k, i);
                } else if (cand_less(c, heap[0])) {
                    heap[0] = c;
                    sift_down(heap, k, 0);
                }
            }
            qsort(heap, (size_t)n, sizeof(cand_t), cmp_cand);
            for (int j = 0; j < k; ++j) {
                out_idx[(size_t)q * k + j] = j < n ? heap[j].i : S;
                if (out_dist) out_dist[(size_t)q * k + j] = j < n ? heap[j].d : INFINITY;
            }
        }
        free(heap);
        (void)ss;
    }
    return 0;
}

/* nearest single support row (model/network.py:250-264 point2node), lowest index on ties */
int cofi_oracle_nearest(const float *support, int S, const float *query, int Q, int64_t *out_idx) {
    for (int q = 0; q < Q; ++q) {
        cand_t best = {INFINITY, S};
        for (int s = 0; s < S; ++s) {
            cand_t c = {cofi_oracle_sqdist(query + 3 * (size_t)q, support + 3 * (size_t)s), s};
            if (cand_less(c, best)) best = c;
        }
        out_idx[q] = best.i;
    }
    return 0;
}
