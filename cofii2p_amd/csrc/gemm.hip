// Dense fp32 contraction on v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF/s chip peak).
//   C[m,n] = act( (sum_k A[m,k] * W[n,k]) / rowdiv[m] + bias[n] )
// Both operands are K-contiguous ("NT"): activations row-major, weights in nn.Linear's (N,K)
// layout — no packing of the reference's Linear weights is needed.
//
// Tile: BM x BN x 32 per workgroup of 4 waves (2x2), each wave TM x TN MFMA tiles of 32x32.
// LDS image: rows of 32 k-values padded to 36 floats.  A lane reads its MFMA operands as ONE
// ds_read_b128 per 4 MFMA steps: lane (i = l&31, h = l>>5) reads k = 8c+4h .. 8c+4h+3 of row i and
// uses element e in step (c,e).  The k-order seen by the MFMA chain is therefore a permutation
// (step (c,e) contracts k = 8c+e and 8c+4+e), identical for A and W, which a sum does not care
// about.  Row stride 36 dwords makes the b128 reads (16-lane groups, bank = dword mod 64) and the
// b128 staging writes (8-lane groups, bank mod 32) conflict free (MI355X_MICROARCH §LDS).
//
// Deep-K / small-MN problems are split over K (gridDim.z) into a workspace and reduced in fixed
// order by splitk_epilogue_kernel: deterministic, no float atomics.
#include "common.h"

namespace {

constexpr int BK = 32;
constexpr int LDS_LD = 36;

struct GemmArgs {
    const float *A, *W;
    float *C;
    const float *bias, *rowdiv;
    float *ws;
    int lda, ldw, ldc, M, N, K, act, ksplit, kchunk;
};

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == COFI_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == COFI_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}

template <int BM, int BN, int TM, int TN>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    static_assert(BM == 64 * TM && BN == 64 * TN, "2x2 waves");
    constexpr int A_LD4 = BM / 32;  // float4 loads per thread for the A tile
    constexpr int W_LD4 = BN / 32;
    __shared__ __attribute__((aligned(16))) float lds[2][(BM + BN) * LDS_LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int ntiles = (kend - kbeg + BK - 1) / BK;

    const int lrow = tid >> 3, lk = (tid & 7) * 4;  // staging map: 8 lanes cover one 128-B row slice
    float4 ra[A_LD4], rw[W_LD4];

    auto gload = [&](int t) {
        const int k = kbeg + t * BK + lk;
        const bool kin = k < kend;
#pragma unroll
        for (int j = 0; j < A_LD4; ++j) {
            const int r = m0 + lrow + 32 * j;
            ra[j] = (kin && r < g.M) ? *reinterpret_cast<const float4 *>(g.A + (size_t)r * g.lda + k) : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < W_LD4; ++j) {
            const int r = n0 + lrow + 32 * j;
            rw[j] = (kin && r < g.N) ? *reinterpret_cast<const float4 *>(g.W + (size_t)r * g.ldw + k) : make_float4(0, 0, 0, 0);
        }
    };
    auto sstore = [&](int buf) {
        float *as = lds[buf], *bs = lds[buf] + BM * LDS_LD;
#pragma unroll
        for (int j = 0; j < A_LD4; ++j) *reinterpret_cast<float4 *>(as + (lrow + 32 * j) * LDS_LD + lk) = ra[j];
#pragma unroll
        for (int j = 0; j < W_LD4; ++j) *reinterpret_cast<float4 *>(bs + (lrow + 32 * j) * LDS_LD + lk) = rw[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    if (ntiles > 0) {
        gload(0);
        sstore(0);
    }
    __syncthreads();

    const int li = lane & 31, lh = lane >> 5;
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) gload(t + 1);
        const float *as = lds[buf] + (wm * 32 * TM + li) * LDS_LD + 4 * lh;
        const float *bs = lds[buf] + BM * LDS_LD + (wn * 32 * TN + li) * LDS_LD + 4 * lh;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const float4 *>(as + i * 32 * LDS_LD + 8 * c);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const float4 *>(bs + j * 32 * LDS_LD + 8 * c);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (t + 1 < ntiles) sstore(buf ^ 1);
        __syncthreads();
    }

    // epilogue.  D layout of 32x32 MFMA: row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * 32 * TN + j * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < g.M && col < g.N) {
                    float v = acc[i][j][r];
                    if (g.ksplit > 1) {
                        g.ws[((size_t)blockIdx.z * g.M + row) * g.N + col] = v;
                    } else {
                        if (g.rowdiv) v = v / g.rowdiv[row];
                        if (g.bias) v += g.bias[col];
                        g.C[(size_t)row * g.ldc + col] = apply_act(v, g.act);
                    }
                }
            }
        }
}

__global__ void splitk_epilogue_kernel(GemmArgs g) {
    const size_t total = (size_t)g.M * g.N;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(e / g.N), col = (int)(e % g.N);
        float v = 0.0f;
        for (int z = 0; z < g.ksplit; ++z) v += g.ws[(size_t)z * total + e];
        if (g.rowdiv) v = v / g.rowdiv[row];
        if (g.bias) v += g.bias[col];
        g.C[(size_t)row * g.ldc + col] = apply_act(v, g.act);
    }
}

struct Plan {
    int bm, bn, ksplit, kchunk;
};

// Heuristic: largest tile that still gives >= ~1 workgroup per CU, then split K until the chip
// (256 CUs) is covered about twice, keeping >= 4 k-tiles (128 values) per split.
Plan make_plan(int M, int N, int K) {
    Plan p;
    auto blocks = [&](int bm, int bn) { return (long)cofi_cdiv(M, bm) * cofi_cdiv(N, bn); };
    if (N > 64 && M > 64 && blocks(128, 128) >= 200) {
        p.bm = 128; p.bn = 128;
    } else if (N > 64 && blocks(64, 128) >= 200) {
        p.bm = 64; p.bn = 128;
    } else if (N > 32) {
        p.bm = 64; p.bn = 64;
        if (N > 64 && blocks(64, 128) * 2 >= blocks(64, 64) && blocks(64, 64) > 1024) { p.bn = 128; }
    } else {
        p.bm = 64; p.bn = 64;
    }
    long nb = blocks(p.bm, p.bn);
    int ktiles = cofi_cdiv(K, BK);
    int ks = 1;
    if (nb < 384) {
        ks = (int)((512 + nb - 1) / nb);
        int maxks = ktiles / 4;
        if (maxks < 1) maxks = 1;
        if (ks > maxks) ks = maxks;
        if (ks > 32) ks = 32;
    }
    int tiles_per = cofi_cdiv(ktiles, ks);
    p.kchunk = tiles_per * BK;
    p.ksplit = cofi_cdiv(K, p.kchunk);
    return p;
}

}  // namespace

extern "C" size_t cofi_gemm_f32_workspace(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    Plan p = make_plan(M, N, K);
    return p.ksplit > 1 ? (size_t)p.ksplit * M * N * sizeof(float) : 0;
}

extern "C" int cofi_gemm_f32(const float *A, int lda, const float *W, int ldw, float *C, int ldc, int M, int N, int K,
                             const float *bias, const float *rowdiv, int act, void *ws, size_t ws_bytes, cofi_stream_t stream) {
    if (!A || !W || !C || M < 0 || N <= 0 || K <= 0) return COFI_EINVAL;
    if (M == 0) return 0;
    if ((K & 3) || (lda & 3) || (ldw & 3) || lda < K || ldw < K || ldc < N) return COFI_EINVAL;
    if (((uintptr_t)A & 15) || ((uintptr_t)W & 15)) return COFI_EINVAL;
    if (act < 0 || act > 2) return COFI_EINVAL;
    Plan p = make_plan(M, N, K);
    if (p.ksplit > 1 && (!ws || ws_bytes < (size_t)p.ksplit * M * N * sizeof(float))) return COFI_EWORKSPACE;
    GemmArgs g{A, W, C, bias, rowdiv, (float *)ws, lda, ldw, ldc, M, N, K, act, p.ksplit, p.kchunk};
    dim3 grid(cofi_cdiv(N, p.bn), cofi_cdiv(M, p.bm), p.ksplit);
    hipStream_t s = cofi_s(stream);
    if (p.bm == 128 && p.bn == 128)
        hipLaunchKernelGGL((gemm_kernel<128, 128, 2, 2>), grid, dim3(256), 0, s, g);
    else if (p.bm == 64 && p.bn == 128)
        hipLaunchKernelGGL((gemm_kernel<64, 128, 1, 2>), grid, dim3(256), 0, s, g);
    else
        hipLaunchKernelGGL((gemm_kernel<64, 64, 1, 1>), grid, dim3(256), 0, s, g);
    if (p.ksplit > 1) {
        size_t total = (size_t)M * N;
        int nb = (int)((total + 255) / 256);
        if (nb > 2048) nb = 2048;
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(nb), dim3(256), 0, s, g);
    }
    return cofi_launch_status();
}
