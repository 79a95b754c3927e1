// KPConv part 1 (kernel-point influence + neighbour aggregation), neighbour max-pool and
// nearest up-sample gathers.  Reference: model/kpconv/kpconv.py:91-116, functional.py:5-21,53-66.
//
// kpconv_aggregate: ONE WAVE PER QUERY POINT.  The aggregation of one query,
//     agg[k, c] = sum_h w[h,k] * f[idx[h], c]      (15 x H) . (H x C),
// is an MFMA chain of v_mfma_f32_16x16x4_f32: rows = kernel points (15 padded to 16),
// cols = 16 channels, contraction over 4 neighbours per step.  Lane (g = l>>4, j = l&15) owns
// neighbour h = 4s+g in step s and
//   - computes the influence weight of kernel point k = j for that neighbour (A operand), and
//   - loads channel c0+j of that neighbour's feature row (B operand): 16 lanes read 64 contiguous
//     bytes of one row, so the gather is served in coalesced 64-B segments from L2.
// Nothing is staged through LDS and the (M,H,15) influence tensor / (M,H,C) gather of the reference
// never exist in memory.  Channel passes of 16*NACC channels go over gridDim.y.
#include "bf16_split.h"
#include "common.h"
#include <stdlib.h>

namespace {

// XCD-aware, bijective remap of a 1-D block id: the dispatcher sends block b to XCD b % 8; remapping gives every XCD one
// CONTIGUOUS eighth of the (spatially sorted) queries, so its private 4 MB L2 only ever sees the source rows of one
// spatial region (N*C*4/8 bytes <= 1.3 MB) instead of the whole 10.5 MB source.
__device__ __forceinline__ int xcd_contiguous_block(int b, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = b & 7, i = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

struct KpArgs {
    const float *feats, *q_pts, *s_pts, *kp;
    const int32_t *idx;
    const uint8_t *row_pos;
    float *agg, *cnt;
    int ldf, N, C, M, H, ld_agg;   // N = support rows PER FRAME, M = total query rows
    float sigma;
    int Mpf;                       // query rows per frame (stack mode: frame f owns queries [f*Mpf, ..) and support [f*N, ..))
    size_t planes_lo;              // 0: agg is fp32 (M, ld_agg); else agg is a bf16 hi plane (M, ld_agg bf16) and the lo plane starts planes_lo
                                   // elements later - the GEMM that consumes it then needs no conversion (COFI_GEMM_A_SPLIT)
    const int32_t *order;          // optional processing order (frame-local query ids, stacked per frame): wave w of the grid
                                   // handles query order[w].  A spatially sorted order makes the waves resident on one CU work on
                                   // neighbouring queries whose neighbour rows overlap, so the gather is served from L1 instead of L2.
};

// VEC = contiguous channels one lane loads per neighbour (1, 2 or 4 -> dword / dwordx2 / dwordx4), NCH =
// channel chunks of 16*VEC per pass.  MFMA column j of accumulator (ch, a) is channel
// c0 + ch*16*VEC + VEC*j + a: a permutation of the channel axis that turns the B-operand gather into
// 16 lanes x 4*VEC contiguous bytes per neighbour row.
//
// The texture-address unit of a CU is shared by its 4 SIMDs and spends ~16 cycles per wave-wide memory instruction,
// while one 4-neighbour step is only NCH*VEC MFMAs (32 cycles each) per SIMD: the kernel is kept to ONE feature load
// per (step, channel chunk).  Everything else a step needs - the neighbour's offset from the query, its validity, its
// row id - is staged per 64 neighbours with coalesced/gather loads by lane = neighbour, parked in a wave-private LDS
// record, and fetched back per step with one broadcast ds_read_b128 (4 distinct addresses per wave).
constexpr int KP_PHASE_DEFAULT = 128;  // neighbour records staged per wave and phase (H <= 128: one phase)
constexpr int KP_PAD = 64;     // shadow records behind the (compacted) records of a phase: the round-up to 16 neighbours (<= 15) and the
                               // pipeline's look-ahead (<= 7 steps = 28) read them

template <int VEC, int NCH>
struct KpFeat {
    float f[NCH][VEC];
};

// The MFMA / feature-load loop of one wave over the `nin` staged neighbour records of a query (rec: LDS, followed by >= 64 shadow
// records), channels coffb[] of the rows behind fbase: acc += W^T F.
template <int VEC, int NCH>
__device__ __forceinline__ void kp_steps(const char *fbase, const unsigned ldfb, const unsigned (&coffb)[NCH], const float4 *rec, const int nin,
                                         const float kx, const float ky, const float kz, const float inv_sigma, f32x4 (&acc)[NCH][VEC]) {
    const int g = (threadIdx.x & 63) >> 4;
    {
        const int steps = (nin + 3) >> 2;
        const float4 *rg = rec + g;  // record of neighbour 4t+g = rg[4t]; reads past the phase hit later records or the pad
        // The loads ARE the software pipeline.  Each one is followed by a compiler-level memory barrier (no instruction):
        // without it the optimiser folds the loop-carried loaded values into loop-carried ADDRESSES and re-issues every
        // load at its use, which serialises the L2 latency into each step.
        auto fetch = [&](int t) -> float4 {
            const float4 v = rg[4 * t];
            asm volatile("" ::: "memory");
            return v;
        };
        auto issue = [&](const float4 &rc, KpFeat<VEC, NCH> &f) {
            const unsigned id = (unsigned)__float_as_int(rc.w);
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                // byte offset of (row, channel chunk): < 4 GiB with id, ldfb < 2^24 (checked by the host) -> one full-rate mad
                const char *src = fbase + (size_t)(__umul24(id, ldfb) + coffb[ch]);
                if constexpr (VEC == 4) {
                    const f32x4 tt = *reinterpret_cast<const f32x4 *>(src);
                    f.f[ch][0] = tt[0]; f.f[ch][1] = tt[1]; f.f[ch][2] = tt[2]; f.f[ch][3] = tt[3];
                } else if constexpr (VEC == 2) {
                    const f32x2 tt = *reinterpret_cast<const f32x2 *>(src);
                    f.f[ch][0] = tt[0]; f.f[ch][1] = tt[1];
                } else {
                    f.f[ch][0] = *reinterpret_cast<const float *>(src);
                }
            }
            asm volatile("" ::: "memory");
        };
        auto consume = [&](const float4 &rc, const KpFeat<VEC, NCH> &f) {
            // kpconv.py:93-99: ((s - q) - kp)^2 summed, sqrt, 1 - d/sigma, clamp at 0 (straight-line: select, no branch)
            const float dx = rc.x - kx, dy = rc.y - ky, dz = rc.z - kz;
            const float sq = (dx * dx + dy * dy) + dz * dz;
            const float w = fmaxf(1.0f - __builtin_amdgcn_sqrtf(sq) * inv_sigma, 0.0f);
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    acc[ch][v] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, f.f[ch][v], acc[ch][v], 0, 0, 0);
        };
        // software pipeline: slot k fetches the record of step k+3 (LDS), runs the MFMAs of step k behind it, then issues
        // the feature loads of step k+3 (L2) - three feature loads stay in flight per wave
        float4 R0 = fetch(0), R1 = fetch(1), R2 = fetch(2), R3;
        KpFeat<VEC, NCH> F0, F1, F2, F3;
        issue(R0, F0); issue(R1, F1); issue(R2, F2);
        for (int t = 0; t < steps; t += 4) {  // steps in [steps, roundup4) read shadow records and add zero
            R3 = fetch(t + 3); consume(R0, F0); issue(R3, F3);
            R0 = fetch(t + 4); consume(R1, F1); issue(R0, F0);
            R1 = fetch(t + 5); consume(R2, F2); issue(R1, F1);
            R2 = fetch(t + 6); consume(R3, F3); issue(R2, F2);
        }
    }
}

// The aggregation of ONE query m by one wave (see the file header): acc[ch][v] = MFMA accumulators (rows = kernel points,
// column j = channel c0 + ch*16*VEC + VEC*j + v), npos = neighbours whose feature row sums to > 0 (kpconv.py:113-114).
// rec = the wave's private LDS record buffer (PHASE + KP_PAD entries); a is a private copy (frame shift).
template <int VEC, int NCH, int PHASE>
__device__ __forceinline__ void kp_aggregate_query(KpArgs a, const int m, const int c0, float4 *rec, f32x4 (&acc)[NCH][VEC], int &npos) {
    constexpr int KP_PHASE = PHASE;
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    {   // stack mode: shift the support-side bases to this query's frame (indices are frame-local)
        const size_t fo = (size_t)(m / a.Mpf) * a.N;
        a.feats += fo * a.ldf;
        a.s_pts += fo * 3;
        a.row_pos += fo;
    }
    const float qx = a.q_pts[3 * m], qy = a.q_pts[3 * m + 1], qz = a.q_pts[3 * m + 2];
    // MFMA row 15 (lane j == 15) is a padding row that is never stored: it may hold any finite weight
    const int jk = j < 15 ? j : 0;
    const float kx = a.kp[3 * jk], ky = a.kp[3 * jk + 1], kz = a.kp[3 * jk + 2];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[ch][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    npos = 0;
    const int32_t *irow = a.idx + (size_t)m * a.H;
    // per-lane channel byte offsets, clamped so that every load is unconditional (branch-free inner loop).  A lane whose
    // channels are out of range multiplies whatever it loaded into MFMA columns that are never stored; a shadow neighbour
    // has weight 0 (its record points at row 0, finite data).
    unsigned coffb[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int c = c0 + ch * 16 * VEC + VEC * j;
        coffb[ch] = c < a.C ? 4u * c : 0u;
    }
    const char *fbase = reinterpret_cast<const char *>(a.feats);
    const unsigned ldfb = 4u * a.ldf;
    const float inv_sigma = 1.0f / a.sigma;
    // Support of the kernel: the influence max(0, 1 - |d - k| / sigma) of EVERY kernel point k is exactly 0 for a neighbour at |d| >=
    // max|k| + sigma.  The k nearest neighbours of the pyramid reach far beyond that (KITTI-shaped frames: 2 - 26 % of the 128 lie
    // inside), so the staging keeps only the neighbours inside the support (compacted, order preserved) and the MFMA / feature-load
    // loop runs over those: exact - the dropped terms are zeros - up to the regrouping of the fp32 sums.  (The bound carries a 2e-5
    // margin: far more than the rounding of the distances on either side.)
    float rsup;
    {
        float kn = j < 15 ? __builtin_amdgcn_sqrtf((kx * kx + ky * ky) + kz * kz) : 0.f;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) kn = fmaxf(kn, __shfl_xor(kn, o, 64));
        rsup = (kn + a.sigma) * 1.00002f;
    }
    const float rsup2 = rsup * rsup;

    for (int h0 = 0; h0 < a.H; h0 += KP_PHASE) {
        const int nh = a.H - h0 < KP_PHASE ? a.H - h0 : KP_PHASE;  // multiple of 4
        // ---- stage the records of this phase: lane = neighbour; all index loads first, then all gathers
        int idr[KP_PHASE / 64];
#pragma unroll
        for (int r = 0; r < KP_PHASE / 64; ++r) {
            const int hl = r * 64 + lane;
            idr[r] = irow[h0 + (hl < nh ? hl : 0)];
            if (hl >= nh) idr[r] = a.N;
        }
        float px[KP_PHASE / 64], py[KP_PHASE / 64], pz[KP_PHASE / 64];
        int pos[KP_PHASE / 64];
#pragma unroll
        for (int r = 0; r < KP_PHASE / 64; ++r) {
            const int idc = (unsigned)idr[r] < (unsigned)a.N ? idr[r] : 0;
            const float *sp = a.s_pts + 3 * (size_t)idc;
            px[r] = sp[0]; py[r] = sp[1]; pz[r] = sp[2];
            pos[r] = a.row_pos[idc];
        }
        int nin = 0;   // neighbours of this phase inside the kernel's support (wave-uniform)
#pragma unroll
        for (int r = 0; r < KP_PHASE / 64; ++r) {
            const bool valid = (unsigned)idr[r] < (unsigned)a.N;
            npos += __popcll(__ballot(valid && pos[r] != 0));   // kpconv.py:113-115 counts over ALL neighbours
            // kpconv.py:93: neighbours centred on the query
            const float dx = px[r] - qx, dy = py[r] - qy, dz = pz[r] - qz;
            const bool inside = valid && (dx * dx + dy * dy) + dz * dz < rsup2;
            const unsigned long long msk = __ballot(inside);
            if (inside) rec[nin + __popcll(msk & ((1ull << lane) - 1ull))] = make_float4(dx, dy, dz, __int_as_float(idr[r]));
            nin += __popcll(msk);
        }
        // shadow records (1e18 away: influence exactly 0, row 0: finite data) behind the list, so the step loop needs no validity test
        rec[nin + lane] = make_float4(1e18f, 0.f, 0.f, __int_as_float(0));
        __builtin_amdgcn_wave_barrier();
        kp_steps<VEC, NCH>(fbase, ldfb, coffb, rec, nin, kx, ky, kz, inv_sigma, acc);
        __builtin_amdgcn_wave_barrier();
    }
}

template <int VEC, int NCH>
__global__ __launch_bounds__(256) void kpconv_aggregate_kernel(KpArgs a) {
    __shared__ float4 rec_s[4][KP_PHASE_DEFAULT + KP_PAD];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // the query id is wave-uniform: keep it (and everything derived from it) in scalar registers
    int m = __builtin_amdgcn_readfirstlane(xcd_contiguous_block(blockIdx.x, gridDim.x) * 4 + wv);
    if (m >= a.M) return;
    if (a.order) m = (m / a.Mpf) * a.Mpf + a.order[m];
    const int j = lane & 15, g = lane >> 4;
    const int c0 = blockIdx.y * (16 * VEC * NCH);
    f32x4 acc[NCH][VEC];
    int npos;
    kp_aggregate_query<VEC, NCH, KP_PHASE_DEFAULT>(a, m, c0, rec_s[wv], acc, npos);
    // D layout 16x16: row (kernel point) = 4*g + r, col = j  ->  channels c .. c+VEC-1 contiguous
    if (a.planes_lo) {   // bf16 hi / lo planes (same rounding as the GEMM's on-the-fly split: identical products)
        uint16_t *hrow = reinterpret_cast<uint16_t *>(a.agg) + (size_t)m * a.ld_agg;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int c = c0 + ch * 16 * VEC + VEC * j;
            if (c < a.C) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 4 * g + r;
                    if (k < 15) {
                        uint16_t *dst = hrow + (size_t)k * a.C + c;
                        if constexpr (VEC == 4) {
                            uint2 hi, lo;
                            cofi_split2(acc[ch][0][r], acc[ch][1][r], hi.x, lo.x);
                            cofi_split2(acc[ch][2][r], acc[ch][3][r], hi.y, lo.y);
                            *reinterpret_cast<uint2 *>(dst) = hi;
                            *reinterpret_cast<uint2 *>(dst + a.planes_lo) = lo;
                        } else if constexpr (VEC == 2) {
                            unsigned hi, lo;
                            cofi_split2(acc[ch][0][r], acc[ch][1][r], hi, lo);
                            *reinterpret_cast<unsigned *>(dst) = hi;
                            *reinterpret_cast<unsigned *>(dst + a.planes_lo) = lo;
                        } else {
                            unsigned hi, lo;
                            cofi_split2(acc[ch][0][r], 0.f, hi, lo);
                            dst[0] = (uint16_t)hi;
                            dst[a.planes_lo] = (uint16_t)lo;
                        }
                    }
                }
            }
        }
        if (blockIdx.y == 0 && lane == 0) a.cnt[m] = (float)(npos > 1 ? npos : 1);
        return;
    }
    float *orow = a.agg + (size_t)m * a.ld_agg;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int c = c0 + ch * 16 * VEC + VEC * j;
        if (c < a.C) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 4 * g + r;
                if (k < 15) {
                    float *dst = orow + (size_t)k * a.C + c;
                    if constexpr (VEC == 4)
                        *reinterpret_cast<float4 *>(dst) = make_float4(acc[ch][0][r], acc[ch][1][r], acc[ch][2][r], acc[ch][3][r]);
                    else if constexpr (VEC == 2)
                        *reinterpret_cast<float2 *>(dst) = make_float2(acc[ch][0][r], acc[ch][1][r]);
                    else
                        dst[0] = acc[ch][0][r];
                }
            }
        }
    }
    if (blockIdx.y == 0 && lane == 0) a.cnt[m] = (float)(npos > 1 ? npos : 1);
}

// Wide layers (C = 256 / 512: NP = C / 128 channel passes) with few queries (1280 ... 2560 at the deep stages): the NP passes of a
// query are NP waves of ONE workgroup that stage its neighbour records TOGETHER - each wave a 1/NP share of the 128 neighbours,
// survivors concatenated in neighbour order through LDS - instead of every pass repeating the whole staging.  Same records, same
// order as kp_aggregate_query: identical bits.
template <int NP>
__global__ __launch_bounds__(256) void kpconv_aggregate_shared_kernel(KpArgs a) {
    constexpr int VEC = 4, NCH = 2, QW = 4 / NP, SH = 128 / NP;   // queries per workgroup, neighbours staged per wave
    __shared__ float4 rec_s[QW][128 + KP_PAD];
    __shared__ int cnt_s[QW][NP], pos_s[QW][NP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int qs = wv / NP, p = wv % NP;
    const int j = lane & 15, g = lane >> 4;
    int m = __builtin_amdgcn_readfirstlane(xcd_contiguous_block(blockIdx.x, gridDim.x) * QW + qs);   // host: M % QW == 0
    if (a.order) m = (m / a.Mpf) * a.Mpf + a.order[m];
    {
        const size_t fo = (size_t)(m / a.Mpf) * a.N;
        a.feats += fo * a.ldf;
        a.s_pts += fo * 3;
        a.row_pos += fo;
    }
    const float qx = a.q_pts[3 * m], qy = a.q_pts[3 * m + 1], qz = a.q_pts[3 * m + 2];
    const int jk = j < 15 ? j : 0;
    const float kx = a.kp[3 * jk], ky = a.kp[3 * jk + 1], kz = a.kp[3 * jk + 2];
    const float inv_sigma = 1.0f / a.sigma;
    float rsup;
    {
        float kn = j < 15 ? __builtin_amdgcn_sqrtf((kx * kx + ky * ky) + kz * kz) : 0.f;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) kn = fmaxf(kn, __shfl_xor(kn, o, 64));
        rsup = (kn + a.sigma) * 1.00002f;
    }
    const float rsup2 = rsup * rsup;
    // ---- this wave's share of the staging: neighbours [p * SH, (p + 1) * SH), lane = neighbour
    const int hl = p * SH + lane;
    const bool mine = lane < SH && hl < a.H;
    const int id = mine ? a.idx[(size_t)m * a.H + hl] : a.N;
    const bool valid = (unsigned)id < (unsigned)a.N;
    const int idc = valid ? id : 0;
    const float *sp = a.s_pts + 3 * (size_t)idc;
    const float dx = sp[0] - qx, dy = sp[1] - qy, dz = sp[2] - qz;
    const int rp = a.row_pos[idc];
    const bool inside = valid && (dx * dx + dy * dy) + dz * dz < rsup2;
    const unsigned long long msk = __ballot(inside);
    const int npos_w = __popcll(__ballot(valid && rp != 0));
    if (lane == 0) { cnt_s[qs][p] = __popcll(msk); pos_s[qs][p] = npos_w; }
    __syncthreads();
    int off = 0, nin = 0, npos = 0;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int c = cnt_s[qs][q];
        off += q < p ? c : 0;
        nin += c;
        npos += pos_s[qs][q];
    }
    float4 *rec = rec_s[qs];
    if (inside) rec[off + __popcll(msk & ((1ull << lane) - 1ull))] = make_float4(dx, dy, dz, __int_as_float(id));
    if (p == 0) rec[nin + lane] = make_float4(1e18f, 0.f, 0.f, __int_as_float(0));   // shadow records behind the list
    __syncthreads();
    // ---- this wave's channel pass over the shared records
    const int c0 = p * (16 * VEC * NCH);
    unsigned coffb[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) coffb[ch] = 4u * (c0 + ch * 16 * VEC + VEC * j);
    f32x4 acc[NCH][VEC];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[ch][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    kp_steps<VEC, NCH>(reinterpret_cast<const char *>(a.feats), 4u * a.ldf, coffb, rec, nin, kx, ky, kz, inv_sigma, acc);
    if (a.planes_lo) {   // bf16 hi / lo planes, as kpconv_aggregate_kernel writes them
        uint16_t *hrow = reinterpret_cast<uint16_t *>(a.agg) + (size_t)m * a.ld_agg;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int c = c0 + ch * 16 * VEC + VEC * j;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 4 * g + r;
                if (k < 15) {
                    uint16_t *dst = hrow + (size_t)k * a.C + c;
                    uint2 hi, lo;
                    cofi_split2(acc[ch][0][r], acc[ch][1][r], hi.x, lo.x);
                    cofi_split2(acc[ch][2][r], acc[ch][3][r], hi.y, lo.y);
                    *reinterpret_cast<uint2 *>(dst) = hi;
                    *reinterpret_cast<uint2 *>(dst + a.planes_lo) = lo;
                }
            }
        }
        if (p == 0 && lane == 0) a.cnt[m] = (float)(npos > 1 ? npos : 1);
        return;
    }
    float *orow = a.agg + (size_t)m * a.ld_agg;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int c = c0 + ch * 16 * VEC + VEC * j;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 4 * g + r;
            if (k < 15)
                *reinterpret_cast<float4 *>(orow + (size_t)k * a.C + c) = make_float4(acc[ch][0][r], acc[ch][1][r], acc[ch][2][r], acc[ch][3][r]);
        }
    }
    if (p == 0 && lane == 0) a.cnt[m] = (float)(npos > 1 ? npos : 1);
}

// C <= 4 (the first layer of the encoder: [intensity | normal]).  What a neighbour costs in the staging is the number of distinct
// cache lines its data lies in (measured: without the separate position gathers the general kernel's 40 us for this layer drop to
// 29), so this layer reads ONE 32-byte record per neighbour, packed once per frame by kp_pack_c4_kernel: [f0 f1 f2 f3 | x y z | pos],
// pos = (f0 + .. + f3 > 0) (kpconv.py:113-114).  Lane = neighbour gathers the record, parks offset and features in LDS, and the MFMA
// chain (same operands, same order as the general kernel: identical bits) runs from LDS alone.
__global__ void kp_pack_c4_kernel(const float *feats, int ldf, int C, const float *pts, int N, float4 *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float *f = feats + (size_t)i * ldf, *p = pts + (size_t)i * 3;
    const float4 fv = make_float4(f[0], C > 1 ? f[1] : 0.f, C > 2 ? f[2] : 0.f, C > 3 ? f[3] : 0.f);
    const float sum = (fv.x + fv.z) + (fv.y + fv.w);   // the butterfly order of row_sum_positive_kernel (wave_sum) for C <= 4
    out[2 * (size_t)i] = fv;
    out[2 * (size_t)i + 1] = make_float4(p[0], p[1], p[2], sum > 0.0f ? 1.0f : 0.0f);
}

__global__ __launch_bounds__(256) void kpconv_aggregate_c4_kernel(KpArgs a) {   // a.feats = packed records (N, 8)
    constexpr int PH = 128;
    __shared__ float4 rec_s[4][PH + 4];   // + one step of shadow records behind the compacted list
    __shared__ float4 fea_s[4][PH + 4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int m = __builtin_amdgcn_readfirstlane(xcd_contiguous_block(blockIdx.x, gridDim.x) * 4 + wv);
    if (m >= a.M) return;
    if (a.order) m = (m / a.Mpf) * a.Mpf + a.order[m];
    const int j = lane & 15, g = lane >> 4;
    const float4 *recs = reinterpret_cast<const float4 *>(a.feats) + 2 * (size_t)(m / a.Mpf) * a.N;   // stack mode: this query's frame
    const float qx = a.q_pts[3 * m], qy = a.q_pts[3 * m + 1], qz = a.q_pts[3 * m + 2];
    const int jk = j < 15 ? j : 0;
    const float kx = a.kp[3 * jk], ky = a.kp[3 * jk + 1], kz = a.kp[3 * jk + 2];
    const float inv_sigma = 1.0f / a.sigma;
    float rsup;   // support of the kernel, as in kp_aggregate_query: same neighbours kept, same order -> identical bits
    {
        float kn = j < 15 ? __builtin_amdgcn_sqrtf((kx * kx + ky * ky) + kz * kz) : 0.f;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) kn = fmaxf(kn, __shfl_xor(kn, o, 64));
        rsup = (kn + a.sigma) * 1.00002f;
    }
    const float rsup2 = rsup * rsup;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int npos = 0;
    const int32_t *irow = a.idx + (size_t)m * a.H;
    float4 *rec = rec_s[wv], *fea = fea_s[wv];
    const float fmask = j < a.C ? 1.0f : 0.0f;           // MFMA columns past C multiply zeros
    const float *fsel = reinterpret_cast<const float *>(fea) + (j & 3);
    for (int h0 = 0; h0 < a.H; h0 += PH) {
        const int nh = a.H - h0 < PH ? a.H - h0 : PH;    // multiple of 4
        int idr[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int hl = r * 64 + lane;
            idr[r] = hl < nh ? irow[h0 + hl] : a.N;
        }
        float4 fr[2], pp[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int idc = (unsigned)idr[r] < (unsigned)a.N ? idr[r] : 0;
            fr[r] = recs[2 * (size_t)idc];
            pp[r] = recs[2 * (size_t)idc + 1];
        }
        int nin = 0;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const bool valid = (unsigned)idr[r] < (unsigned)a.N;
            npos += __popcll(__ballot(valid && pp[r].w != 0.f));
            const float dx = pp[r].x - qx, dy = pp[r].y - qy, dz = pp[r].z - qz;
            const bool inside = valid && (dx * dx + dy * dy) + dz * dz < rsup2;
            const unsigned long long msk = __ballot(inside);
            if (inside) {
                const int slot = nin + __popcll(msk & ((1ull << lane) - 1ull));
                rec[slot] = make_float4(dx, dy, dz, 0.f);
                fea[slot] = fr[r];
            }
            nin += __popcll(msk);
        }
        if (lane < 4) {   // the last step is filled up with shadow neighbours: influence exactly 0, zero features
            rec[nin + lane] = make_float4(1e18f, 0.f, 0.f, 0.f);
            fea[nin + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __builtin_amdgcn_wave_barrier();
        const int steps = (nin + 3) >> 2;
        for (int t = 0; t < steps; ++t) {
            const float4 rc = rec[4 * t + g];
            const float f = fsel[(4 * t + g) * 4] * fmask;
            const float dx = rc.x - kx, dy = rc.y - ky, dz = rc.z - kz;
            const float sq = (dx * dx + dy * dy) + dz * dz;
            const float w = fmaxf(1.0f - __builtin_amdgcn_sqrtf(sq) * inv_sigma, 0.0f);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w, f, acc, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (j < a.C) {
        float *orow = a.agg + (size_t)m * a.ld_agg;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 4 * g + r;
            if (k < 15) orow[(size_t)k * a.C + j] = acc[r];
        }
    }
    if (lane == 0) a.cnt[m] = (float)(npos > 1 ? npos : 1);
}

// ---- KPConv as ONE kernel for the narrow layers (mid = 32 / 64 channels: the stages with 20 480 ... 5 120 queries) ----------------
// kpconv.py:91-116 end to end:  y[m] = (sum_k agg[m, k, :] W[k]) / max(#neighbours with positive feature sum, 1) + bias,
// plus the GroupNorm statistics partials of y.  The (M, 15 mid) aggregate - 39 MB per layer at these stages, written and read back
// by the two-kernel form - never leaves the CU:
//   a workgroup (8 waves) owns `qt` queries (= one statistics slab: 64 / 32 / 16 rows, so that a layer is >= 256 workgroups) and
//   works through them 16 at a time: every wave aggregates two queries exactly as kpconv_aggregate_kernel does and leaves their
//   (15 x mid) results as bf16 hi / lo planes in an LDS tile (16 x 15 mid); then the 8 waves multiply the tile with W
//   (pre-split bf16 planes streamed from L2, 3-term split on v_mfma_f32_16x16x32_bf16): wave = (16-column tile, K range), partial
//   tiles summed through LDS in a fixed order; epilogue: / count, + bias, store, column sums for the statistics.
// LDS 75 KB (mid 64, records staged 64 neighbours at a time) / 52 KB (mid 32): two workgroups per CU; the gather phase is
// bound by the texture-address unit, not by occupancy (measured: 16 waves per CU run it as fast as 28).
// MEASURED on MI355X (KITTI frame, us per launch, fused vs aggregate + GEMM): mid 32, 20 480 queries 58-91 vs ~65; mid 64, 10 240
// queries 63-87 vs ~52; whole forward 490 vs 503 frames/s.  A workgroup alternates between a gather phase (texture unit busy,
// matrix cores idle) and a GEMM phase (the opposite) with barriers in between, and 320-1280 such workgroups quantise badly on 256 CUs
// x 2 slots, while the two-kernel form spreads 20 480 independent one-wave queries evenly and its GEMM runs at full occupancy.  Kept
// as an opt-in (COFI_KPCONV_FUSED=1 in cofii2p_amd/kpfpn.py): it removes 2 x 39 MB of fabric traffic per layer, which matters once
// the frame is bandwidth-bound; not the default.
struct KpFusedArgs {
    KpArgs a;                       // agg / cnt / ld_agg unused
    const uint16_t *w_hi, *w_lo;    // bf16 planes of W, (mid rows, ldw), K index = kernel point * mid + channel
    int ldw;
    const float *bias;
    float *y;
    int ldy;
    float *colpart;                 // (M / qt, mid >> stat_shift, 2) or nullptr
    int stat_shift;
    int qt;                         // queries per workgroup = rows per statistics slab: 16, 32 or 64
};

template <int MID, int PHASE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void kpconv_fused_kernel(KpFusedArgs fa) {   // two workgroups per CU

    constexpr int VEC = MID / 16, K = 15 * MID, KLD = K + 8, NT = MID / 16, KQ = 8 / NT, KSTEPS = K / 32;
    constexpr int REC = PHASE + KP_PAD;
    constexpr int RG = 512 / MID;   // epilogue: thread = (column, row group); row groups
    constexpr int RPT = 16 / RG;    // rows per thread and 16-query step
    constexpr int PF = 4;           // k-steps of W fragments in flight per wave
    static_assert(REC * 16 >= 256 * 4, "the partial tiles alias the record buffers");
    __shared__ float4 rec_s[8][REC];
    __shared__ __attribute__((aligned(16))) uint16_t t_hi[16 * KLD];
    __shared__ __attribute__((aligned(16))) uint16_t t_lo[16 * KLD];
    __shared__ float cnt_s[16];
    __shared__ int m_s[16];
    float *part_s = reinterpret_cast<float *>(&rec_s[0][0]);   // [8][256]; live only inside the GEMM phase / the final fold
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int b = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const int en = tid % MID, erg = tid / MID;
    const float bias_n = fa.bias ? fa.bias[en] : 0.f;
    float cs = 0.f, cq = 0.f;
    // GEMM-phase role of this wave
    const int nt = wv % NT, kq = wv / NT;
    const int ks0 = kq * KSTEPS / KQ, ks1 = (kq + 1) * KSTEPS / KQ;
    const uint16_t *wh = fa.w_hi + (size_t)(nt * 16 + j) * fa.ldw + g * 8;
    const uint16_t *wl = fa.w_lo + (size_t)(nt * 16 + j) * fa.ldw + g * 8;
    const uint16_t *ah = t_hi + j * KLD + g * 8, *al = t_lo + j * KLD + g * 8;

    for (int st = 0; st < fa.qt; st += 16) {
        // ---- aggregation: two queries per wave
#pragma unroll 1
        for (int u = 0; u < 2; ++u) {
            const int ql = 2 * wv + u;
            const int pos = b * fa.qt + st + ql;
            int m = pos;
            if (fa.a.order) m = (pos / fa.a.Mpf) * fa.a.Mpf + fa.a.order[pos];
            m = __builtin_amdgcn_readfirstlane(m);
            f32x4 acc[1][VEC];
            int npos;
            kp_aggregate_query<VEC, 1, PHASE>(fa.a, m, 0, rec_s[wv], acc, npos);
            // D layout 16x16: row (kernel point) = 4 g + r, column j -> channels VEC j .. VEC j + VEC - 1 of tile row ql
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 4 * g + r;
                if (k < 15) {
                    const int off = ql * KLD + k * MID + VEC * j;
                    if constexpr (VEC == 4) {
                        uint2 hi, lo;
                        cofi_split2(acc[0][0][r], acc[0][1][r], hi.x, lo.x);
                        cofi_split2(acc[0][2][r], acc[0][3][r], hi.y, lo.y);
                        *reinterpret_cast<uint2 *>(t_hi + off) = hi;
                        *reinterpret_cast<uint2 *>(t_lo + off) = lo;
                    } else {
                        unsigned hi, lo;
                        cofi_split2(acc[0][0][r], acc[0][1][r], hi, lo);
                        *reinterpret_cast<unsigned *>(t_hi + off) = hi;
                        *reinterpret_cast<unsigned *>(t_lo + off) = lo;
                    }
                }
            }
            if (lane == 0) {
                cnt_s[ql] = (float)(npos > 1 ? npos : 1);
                m_s[ql] = m;
            }
        }
        // ---- tile (16 x K) . W^T (K x MID): this wave's 16 columns over its K range; the first W fragments are requested before
        //      the barrier (they do not depend on the tile)
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        CofiFrag bh[PF], bl[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int ks = ks0 + i < ks1 ? ks0 + i : ks1 - 1;
            bh[i].u = *reinterpret_cast<const uint4 *>(wh + ks * 32);
            bl[i].u = *reinterpret_cast<const uint4 *>(wl + ks * 32);
        }
        __syncthreads();
#pragma unroll 1
        for (int k0 = ks0; k0 < ks1; k0 += PF) {
            if (k0 != ks0) {
#pragma unroll
                for (int i = 0; i < PF; ++i) {
                    const int ks = k0 + i < ks1 ? k0 + i : ks1 - 1;
                    bh[i].u = *reinterpret_cast<const uint4 *>(wh + ks * 32);
                    bl[i].u = *reinterpret_cast<const uint4 *>(wl + ks * 32);
                }
            }
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                if (k0 + i < ks1) {   // wave-uniform
                    CofiFrag fh, fl;
                    fh.u = *reinterpret_cast<const uint4 *>(ah + (k0 + i) * 32);
                    fl.u = *reinterpret_cast<const uint4 *>(al + (k0 + i) * 32);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl.v, bh[i].v, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh.v, bl[i].v, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh.v, bh[i].v, c, 0, 0, 0);
                }
            }
        }
        // D layout: column j, rows 4 g + r -> part_s[wave][row][column]
#pragma unroll
        for (int r = 0; r < 4; ++r) part_s[wv * 256 + (4 * g + r) * 16 + j] = c[r];
        __syncthreads();
        // ---- fold the K ranges (fixed order), / count, + bias, store, column sums
#pragma unroll
        for (int rr = 0; rr < RPT; ++rr) {
            const int q = erg * RPT + rr;
            float x = 0.f;
#pragma unroll
            for (int z = 0; z < KQ; ++z) x += part_s[(z * NT + (en >> 4)) * 256 + q * 16 + (en & 15)];
            x = x / cnt_s[q] + bias_n;
            fa.y[(size_t)m_s[q] * fa.ldy + en] = x;
            cs += x;
            cq += x * x;
        }
        __syncthreads();   // tile, records (= partial tiles), cnt_s / m_s are rewritten by the next step
    }
    if (fa.colpart) {
        part_s[(erg * MID + en) * 2] = cs;
        part_s[(erg * MID + en) * 2 + 1] = cq;
        __syncthreads();
        if (tid < MID) {
            float s = 0.f, q = 0.f;
            for (int p = 0; p < RG; ++p) {
                s += part_s[(p * MID + tid) * 2];
                q += part_s[(p * MID + tid) * 2 + 1];
            }
            // one table entry per 2^stat_shift adjacent columns (fp64 butterfly, fixed order), as the GEMM epilogue writes it
            double ds = s, dq = q;
            for (int o = 1; o < (1 << fa.stat_shift); o <<= 1) {
                ds += __shfl_xor(ds, o, 64);
                dq += __shfl_xor(dq, o, 64);
            }
            if ((tid & ((1 << fa.stat_shift) - 1)) == 0) {
                float *o = fa.colpart + ((size_t)b * (MID >> fa.stat_shift) + (tid >> fa.stat_shift)) * 2;
                o[0] = (float)ds;
                o[1] = (float)dq;
            }
        }
    }
}

// row_pos[n] = (sum_c feats[n,c] > 0); one wave per row (kpconv.py:113-114 applied per source row)
__global__ void row_sum_positive_kernel(const float *feats, int ld, int N, int C, uint8_t *row_pos) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += feats[(size_t)n * ld + c];
    s = wave_sum(s);
    if (lane == 0) row_pos[n] = s > 0.0f ? 1 : 0;
}

// out[m,c] = max_h x[idx[m,h], c], zero row behind idx == N.  One wave per (query, 32-channel chunk):
// lane (g = l>>3, c4 = l&7) reads float4 #c4 of the chunk for neighbours h = 8i+g, so one load instruction
// fetches 8 neighbour rows x 128 contiguous bytes; the 8 lane groups are folded with 3 xor-shuffles.
// Chunks are the slow grid axis: all queries of one chunk run together and its source slice
// (N x 128 B <= 2.6 MB) stays resident in every XCD's 4 MB L2.
__global__ __launch_bounds__(256) void neighbor_maxpool_kernel(const float *x, int ldx, int N, int C, const int32_t *idx, int M,
                                                               int H, float *out, int ldo, int Mpf, const int32_t *order) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int m = xcd_contiguous_block(blockIdx.x, gridDim.x) * 4 + wv;
    if (m >= M) return;
    if (order) m = (m / Mpf) * Mpf + order[m];  // spatially sorted processing order (see KpArgs::order)
    x += (size_t)(m / Mpf) * N * ldx;  // stack mode: frame-local indices
    const int g = lane >> 3, c = blockIdx.y * 32 + (lane & 7) * 4;
    const int32_t *irow = idx + (size_t)m * H;
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    const bool cin = c < C;
    const float *xc = x + (cin ? c : 0);
    // 64 neighbours per round: the index row is read once, coalesced (lane = neighbour), and handed to the 8 lane groups by
    // shuffle, so the only per-neighbour memory instruction is the 1-KB feature load (8 rows x 128 B); the 8 loads of a round
    // are issued back to back.  Shadow neighbours (idx == N) read row 0 and are replaced by the zero row afterwards.
    for (int h0 = 0; h0 < H; h0 += 64) {
        const int hl = h0 + lane;
        const int myid = hl < H ? irow[hl] : INT_MIN;   // INT_MIN: past the row (ignored); anything else outside [0,N): zero row
        float4 v[8];
        int ids[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            ids[i] = __shfl(myid, 8 * i + g, 64);
            const unsigned row = (unsigned)ids[i] < (unsigned)N ? ids[i] : 0;
            v[i] = *reinterpret_cast<const float4 *>(xc + (size_t)row * ldx);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool real = (unsigned)ids[i] < (unsigned)N;
            const float other = ids[i] == INT_MIN ? -INFINITY : 0.f;
            best.x = fmaxf(best.x, real ? v[i].x : other); best.y = fmaxf(best.y, real ? v[i].y : other);
            best.z = fmaxf(best.z, real ? v[i].z : other); best.w = fmaxf(best.w, real ? v[i].w : other);
        }
    }
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
        best.x = fmaxf(best.x, __shfl_xor(best.x, o, 64));
        best.y = fmaxf(best.y, __shfl_xor(best.y, o, 64));
        best.z = fmaxf(best.z, __shfl_xor(best.z, o, 64));
        best.w = fmaxf(best.w, __shfl_xor(best.w, o, 64));
    }
    if (g == 0 && cin) *reinterpret_cast<float4 *>(out + (size_t)m * ldo + c) = best;
}

// out[m,:] = x[idx[m*idx_stride], :] (zero row for idx == N)
__global__ void gather_rows_kernel(const float *x, int ldx, int N, int C, const int32_t *idx, int idx_stride, int M, float *out,
                                   int ldo, int Mpf) {
    const int m = blockIdx.x;
    x += (size_t)(m / Mpf) * N * ldx;  // stack mode: frame-local indices
    const int id = idx[(size_t)m * idx_stride];
    const bool valid = (unsigned)id < (unsigned)N;
    const int c4 = C >> 2;
    if (((ldx | ldo | C) & 3) == 0) {
        for (int c = threadIdx.x; c < c4; c += blockDim.x) {
            float4 v = valid ? reinterpret_cast<const float4 *>(x + (size_t)id * ldx)[c] : make_float4(0, 0, 0, 0);
            reinterpret_cast<float4 *>(out + (size_t)m * ldo)[c] = v;
        }
    } else {
        for (int c = threadIdx.x; c < C; c += blockDim.x) out[(size_t)m * ldo + c] = valid ? x[(size_t)id * ldx + c] : 0.0f;
    }
}

}  // namespace

extern "C" int cofi_row_sum_positive(const float *feats, int ld, int N, int C, uint8_t *row_pos, cofi_stream_t stream) {
    if (!feats || !row_pos || N < 0 || C <= 0 || ld < C) return COFI_EINVAL;
    if (N == 0) return 0;
    hipLaunchKernelGGL(row_sum_positive_kernel, dim3(cofi_cdiv(N, 4)), dim3(256), 0, cofi_s(stream), feats, ld, N, C, row_pos);
    return cofi_launch_status();
}

extern "C" int cofi_kpconv_aggregate(const float *feats, int ldf, int N, int C, const float *q_pts, const float *s_pts,
                                     const int32_t *idx, int M, int H, const float *kernel_points, float sigma,
                                     const uint8_t *row_pos, float *agg, int ld_agg, int agg_planes, float *cnt, int frames, const int32_t *order,
                                     cofi_stream_t stream) {
    if (!feats || !q_pts || !s_pts || !idx || !kernel_points || !row_pos || !agg || !cnt) return COFI_EINVAL;
    if (N <= 0 || C <= 0 || M < 0 || H <= 0 || (H & 3) || ldf < C || ld_agg < 15 * C || !(sigma > 0.f) || frames <= 0) return COFI_EINVAL;
    if (agg_planes && ((ld_agg & 7) || (C & 3) || ((uintptr_t)agg & 15))) return COFI_EINVAL;
    if ((size_t)N * ldf * 4 >= ((size_t)1 << 32) || N >= (1 << 24) || (size_t)ldf * 4 >= (1u << 24)) return COFI_EUNSUPPORTED;  // 24x24-bit row offsets
    if (M == 0) return 0;
    KpArgs a{feats, q_pts, s_pts, kernel_points, idx, row_pos, agg, cnt, ldf, N, C, M * frames, H, ld_agg, sigma, M, agg_planes ? (size_t)M * frames * ld_agg : 0, order};
    M *= frames;
    hipStream_t s = cofi_s(stream);
    const int mb = cofi_cdiv(M, 4);
    if ((C == 256 || C == 512) && (ldf & 3) == 0 && H <= 128 && M % (C == 256 ? 2 : 1) == 0) {
        if (C == 256)
            hipLaunchKernelGGL(kpconv_aggregate_shared_kernel<2>, dim3(M / 2), dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL(kpconv_aggregate_shared_kernel<4>, dim3(M), dim3(256), 0, s, a);
    } else if ((C & 3) == 0 && (ldf & 3) == 0 && C >= 64) {
        if (C % 128 == 0)
            hipLaunchKernelGGL((kpconv_aggregate_kernel<4, 2>), dim3(mb, C / 128), dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL((kpconv_aggregate_kernel<4, 1>), dim3(mb, cofi_cdiv(C, 64)), dim3(256), 0, s, a);
    } else if ((C & 1) == 0 && (ldf & 1) == 0 && C >= 32) {
        hipLaunchKernelGGL((kpconv_aggregate_kernel<2, 1>), dim3(mb, cofi_cdiv(C, 32)), dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL((kpconv_aggregate_kernel<1, 1>), dim3(mb, cofi_cdiv(C, 16)), dim3(256), 0, s, a);
    }
    return cofi_launch_status();
}

// First-layer form (C <= 4): pack [features | position | positive-sum flag] into 32-byte records once, then aggregate from them.
extern "C" int cofi_kp_pack_c4(const float *feats, int ldf, int C, const float *points, int N, float *records, cofi_stream_t stream) {
    if (!feats || !points || !records || N <= 0 || C <= 0 || C > 4 || ldf < C || ((uintptr_t)records & 15)) return COFI_EINVAL;
    hipLaunchKernelGGL(kp_pack_c4_kernel, dim3(cofi_cdiv(N, 256)), dim3(256), 0, cofi_s(stream), feats, ldf, C, points, N, (float4 *)records);
    return cofi_launch_status();
}

extern "C" int cofi_kpconv_aggregate_c4(const float *records, int N, int C, const float *q_pts, const int32_t *idx, int M, int H,
                                        const float *kernel_points, float sigma, float *agg, int ld_agg, float *cnt, int frames,
                                        const int32_t *order, cofi_stream_t stream) {
    if (!records || !q_pts || !idx || !kernel_points || !agg || !cnt || ((uintptr_t)records & 15)) return COFI_EINVAL;
    if (N <= 0 || C <= 0 || C > 4 || M < 0 || H <= 0 || (H & 3) || ld_agg < 15 * C || !(sigma > 0.f) || frames <= 0) return COFI_EINVAL;
    if (M == 0) return 0;
    KpArgs a{records, q_pts, nullptr, kernel_points, idx, nullptr, agg, cnt, 8, N, C, M * frames, H, ld_agg, sigma, M, 0, order};
    hipLaunchKernelGGL(kpconv_aggregate_c4_kernel, dim3(cofi_cdiv((long)M * frames, 4)), dim3(256), 0, cofi_s(stream), a);
    return cofi_launch_status();
}

// rows per statistics slab (= queries per workgroup) the fused kernel uses for M queries per frame: the largest of 64 / 32 / 16 that
// divides M and still gives >= 256 workgroups; 0 = shape not supported (use cofi_kpconv_aggregate + cofi_gemm_f32_fused)
extern "C" int cofi_kpconv_fused_slab_rows(int C, int M, int frames) {
    if ((C != 32 && C != 64) || M <= 0 || frames <= 0 || (M % 16)) return 0;
    for (int q = 64; q >= 32; q >>= 1)
        if (M % q == 0 && (long)M * frames / q >= 256) return q;
    return 16;
}

extern "C" int cofi_kpconv_fused(const float *feats, int ldf, int N, int C, const float *q_pts, const float *s_pts, const int32_t *idx, int M, int H,
                                 const float *kernel_points, float sigma, const uint8_t *row_pos, const void *w_planes, int ldw, const float *bias,
                                 float *y, int ldy, float *colpart, int stat_width, int frames, const int32_t *order, cofi_stream_t stream) {
    if (!feats || !q_pts || !s_pts || !idx || !kernel_points || !row_pos || !w_planes || !y) return COFI_EINVAL;
    if (N <= 0 || M <= 0 || H <= 0 || (H & 3) || ldf < C || ldy < C || !(sigma > 0.f) || frames <= 0) return COFI_EINVAL;
    if (ldw < 15 * C || (ldw & 7) || ((uintptr_t)w_planes & 15)) return COFI_EINVAL;
    const int qt = cofi_kpconv_fused_slab_rows(C, M, frames);
    if (qt == 0 || (ldf & (C / 16 - 1))) return COFI_EUNSUPPORTED;
    if ((size_t)N * ldf * 4 >= ((size_t)1 << 32) || N >= (1 << 24) || (size_t)ldf * 4 >= (1u << 24)) return COFI_EUNSUPPORTED;  // 24x24-bit row offsets
    int shift = 0;
    if (colpart) {
        if (stat_width <= 0 || (stat_width & (stat_width - 1)) || stat_width > 16 || (C % stat_width)) return COFI_EINVAL;
        while ((1 << shift) < stat_width) ++shift;
    }
    KpFusedArgs fa{};
    fa.a = KpArgs{feats, q_pts, s_pts, kernel_points, idx, row_pos, nullptr, nullptr, ldf, N, C, M * frames, H, 0, sigma, M, 0, order};
    fa.w_hi = (const uint16_t *)w_planes;
    fa.w_lo = fa.w_hi + (size_t)C * ldw;
    fa.ldw = ldw; fa.bias = bias; fa.y = y; fa.ldy = ldy; fa.colpart = colpart; fa.stat_shift = shift; fa.qt = qt;
    const int nwg = (int)((long)M * frames / qt);
    if (C == 64)
        hipLaunchKernelGGL((kpconv_fused_kernel<64, 64>), dim3(nwg), dim3(512), 0, cofi_s(stream), fa);
    else
        hipLaunchKernelGGL((kpconv_fused_kernel<32, 128>), dim3(nwg), dim3(512), 0, cofi_s(stream), fa);
    return cofi_launch_status();
}

extern "C" int cofi_neighbor_maxpool(const float *x, int ldx, int N, int C, const int32_t *idx, int M, int H, float *out, int ldo,
                                     int frames, const int32_t *order, cofi_stream_t stream) {
    if (!x || !idx || !out || N <= 0 || C <= 0 || M < 0 || H <= 0 || ldx < C || ldo < C) return COFI_EINVAL;
    if ((C & 3) || (ldx & 3) || (ldo & 3) || ((uintptr_t)x & 15) || ((uintptr_t)out & 15)) return COFI_EINVAL;
    if (M == 0) return 0;
    if (frames <= 0) return COFI_EINVAL;
    hipLaunchKernelGGL(neighbor_maxpool_kernel, dim3(cofi_cdiv(M * frames, 4), cofi_cdiv(C, 32)), dim3(256), 0, cofi_s(stream), x, ldx, N,
                       C, idx, M * frames, H, out, ldo, M, order);
    return cofi_launch_status();
}

extern "C" int cofi_gather_rows(const float *x, int ldx, int N, int C, const int32_t *idx, int idx_stride, int M, float *out,
                                int ldo, int frames, cofi_stream_t stream) {
    if (!x || !idx || !out || N <= 0 || C <= 0 || M < 0 || idx_stride <= 0 || ldx < C || ldo < C) return COFI_EINVAL;
    if (M == 0) return 0;
    if (frames <= 0) return COFI_EINVAL;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(M * frames), dim3(C >= 1024 ? 256 : (C >= 256 ? 128 : 64)), 0, cofi_s(stream), x, ldx, N,
                       C, idx, idx_stride, M * frames, out, ldo, M);
    return cofi_launch_status();
}
