// Flash-style multi-head attention for the I2P transformer on fp32 MFMA (v_mfma_f32_32x32x2_f32).
// Reference: model/transformer/linear_attention.py:56-79 (FullAttention) + the token-axis query
// normalisation of model/transformer/transformer.py:53, folded in as a per-channel scale.
//
// Workgroup = 8 waves = one (head, 32-query block).  The eight waves take interleaved 32-key blocks
// (split-KV), each running an online softmax in registers, and merge (m, l, O) through LDS at the
// end.  The L x S score matrix never exists in memory (the reference materialises it twice).
//
// Per 32-key block and wave:
//   S^T[key, q] = K_blk . Q^T    16 MFMA, A = K straight from L2 (float4 per lane), B = Q registers
//   online softmax, per lane = per query column (q = lane&31), 16 keys per lane, halves joined by
//   one __shfl_xor(.., 32)
//   O^T[d, q] += V_blk^T . P^T   16 MFMA, A = V[key, d = lane&31] (block fetched as 16-B loads, transposed through a
//                                wave-private LDS tile), B = P
// The MFMA D layout of S^T (lane (q,h) holds keys (r&3)+8(r>>2)+4h) is exactly the B-operand layout
// the second chain needs once the contraction index is allowed to run in that (permuted) key order,
// so P never moves between lanes and O^T keeps "one query per lane": the softmax rescale and the
// final 1/l are plain per-lane multiplies.
#include "common.h"

namespace {

struct AttnArgs {
    const float *Q, *K, *V, *qs;
    float *O;
    int ldq, ldk, ldv, ldo, L, S, H;
    float scale_log2e;
    // alternative to qs: the column partials of the projection that produced Q (cofi_gemm_f32_colstats: (frames*nslab, ncols, 2)
    // {sum, sum of squares}); the kernel folds them into 1 / max(||Q[:, c]||, eps) itself (transformer.py:53)
    const float *q_colpart;
    int q_nslab, q_ncols;
    float q_eps;
};

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// NW = waves per workgroup = key splits.  8 when the grid is small (one frame: 160 workgroups - the split is what fills the
// chip); fewer when frames are stacked: every wave then runs a longer key loop and the fixed per-workgroup cost (Q load,
// first K/V round trip, LDS merge) is paid less often.
template <int NW>
__global__ __launch_bounds__(64 * NW) void attention_fwd_kernel(AttnArgs a) {
    constexpr int D = 32;
    __shared__ __attribute__((aligned(16))) float s_o[NW][32][D + 4];
    __shared__ float s_m[NW][32], s_l[NW][32];
    // wave-private transposition buffer for V: a block is fetched as 4 coalesced 16-B loads per lane (32 rows x 128 B) and
    // read back as the MFMA A operand (lane = column d, 16 key rows).  Fetching the operand layout directly takes 16 dword
    // loads per block, and the CU's texture-address unit (~16 cycles per wave-wide load, shared by the 4 SIMDs) then costs
    // as much time as the MFMA chains themselves.
    __shared__ __attribute__((aligned(16))) float s_v[NW][32][D + 4];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int head = blockIdx.y, q0 = blockIdx.x * 32;
    const int hc = head * D;
    {   // stack mode: frame blockIdx.z owns L query rows and S key rows
        const size_t f = blockIdx.z;
        a.Q += f * a.L * a.ldq;
        a.K += f * a.S * a.ldk;
        a.V += f * a.S * a.ldv;
        a.O += f * a.L * a.ldo;
        if (a.qs) a.qs += f * a.H * D;
    }

    // token-axis norm of this head's 32 Q columns from the projection's column partials: thread (phase, column) sums the
    // squares of slabs phase, phase + NPH, ...; column threads fold the phases in a fixed order (deterministic)
    __shared__ __attribute__((aligned(16))) float s_qs[D];
    if (a.q_colpart) {
        constexpr int NPH = 2 * NW;   // 64 * NW threads = NPH phases x 32 columns
        __shared__ float s_part[NPH][D];
        const int col = threadIdx.x & 31, ph = threadIdx.x >> 5;
        const float *cp = a.q_colpart + ((size_t)blockIdx.z * a.q_nslab * a.q_ncols + hc + col) * 2 + 1;
        float acc = 0.f;
        for (int b = ph; b < a.q_nslab; b += NPH) acc += cp[(size_t)b * a.q_ncols * 2];
        s_part[ph][col] = acc;
        __syncthreads();
        if (threadIdx.x < D) {
            float t = 0.f;
#pragma unroll
            for (int p = 0; p < NPH; ++p) t += s_part[p][threadIdx.x];
            s_qs[threadIdx.x] = 1.0f / fmaxf(sqrtf(t), a.q_eps);
        }
        __syncthreads();
    }
    // Q fragment: lane (q = li, h) holds Q[q][8c+4h+e], pre-multiplied by colscale * scale * log2(e)
    float qf[16];
    {
        const int q = min(q0 + li, a.L - 1);
        const float *qp = a.Q + (size_t)q * a.ldq + hc + 4 * lh;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 v = *reinterpret_cast<const float4 *>(qp + 8 * c);
            float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
            if (a.q_colpart) sc = *reinterpret_cast<const float4 *>(&s_qs[4 * lh + 8 * c]);
            else if (a.qs) sc = *reinterpret_cast<const float4 *>(a.qs + hc + 4 * lh + 8 * c);
            qf[4 * c + 0] = (v.x * sc.x) * a.scale_log2e;
            qf[4 * c + 1] = (v.y * sc.y) * a.scale_log2e;
            qf[4 * c + 2] = (v.z * sc.z) * a.scale_log2e;
            qf[4 * c + 3] = (v.w * sc.w) * a.scale_log2e;
        }
    }

    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    const int nblk = (a.S + 31) >> 5;
    auto load_k = [&](int b, float4(&kf)[4]) {
        const int key = min(b * 32 + li, a.S - 1);
        const float *kp = a.K + (size_t)key * a.ldk + hc + 4 * lh;
#pragma unroll
        for (int c = 0; c < 4; ++c) kf[c] = *reinterpret_cast<const float4 *>(kp + 8 * c);
    };
    auto load_v = [&](int b, float4(&vr)[4]) {   // raw rows: lane l holds float4 #(l & 7) of key rows (l >> 3) + 8j
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kr = min(b * 32 + (lane >> 3) + 8 * j, a.S - 1);
            vr[j] = *reinterpret_cast<const float4 *>(a.V + (size_t)kr * a.ldv + hc + 4 * (lane & 7));
        }
    };
    auto transpose_v = [&](const float4(&vr)[4], float(&vf)[16]) {   // -> lane (d = li, h): V[keyrow(r, h)][d]
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<float4 *>(&s_v[wave][(lane >> 3) + 8 * j][4 * (lane & 7)]) = vr[j];
#pragma unroll
        for (int r = 0; r < 16; ++r) vf[r] = s_v[wave][(r & 3) + 8 * (r >> 2) + 4 * lh][li];
    };
    auto qk = [&](const float4(&kf)[4]) {  // S^T = K . Q^T, 16 chained MFMAs
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c].x, qf[4 * c + 0], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c].y, qf[4 * c + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c].z, qf[4 * c + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c].w, qf[4 * c + 3], s, 0, 0, 0);
        }
        return s;
    };
    auto softmax = [&](int b, f32x16 &s) {  // online softmax (base 2); returns P in s, rescales o
        const int k0 = b * 32;
        if (k0 + 32 > a.S) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (k0 + (r & 3) + 8 * (r >> 2) + 4 * lh >= a.S) s[r] = -INFINITY;
        }
        float bmax = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) bmax = fmaxf(bmax, s[r]);
        bmax = fmaxf(bmax, __shfl_xor(bmax, 32, 64));
        const float m_new = fmaxf(m_run, bmax);
        const float alpha = fast_exp2(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = fast_exp2(s[r] - m_new);
            psum += s[r];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= alpha;
    };
    auto pv = [&](const f32x16 &p, const float(&vf)[16]) {  // O^T += V^T . P^T in D-layout key order
#pragma unroll
        for (int r = 0; r < 16; ++r) o = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[r], p[r], o, 0, 0, 0);
    };

    // Software pipeline over this wave's key blocks b, b+NW, ...: the loop body is ONE basic block in which
    // the next block's QK^T MFMA chain is issued before the current block's softmax (VALU/transcendental
    // work runs under matrix-core time), V of the next block and K of the block after it are in flight.
    {
        int b = wave;
        if (b < nblk) {
            float4 kC[4], kN[4], kN2[4], vR[4];
            float vC[16];
            load_k(b, kC);
            load_v(b, vR);
            load_k(min(b + NW, nblk - 1), kN);
            f32x16 sC = qk(kC);
            transpose_v(vR, vC);
            while (b + NW < nblk) {
                load_k(min(b + 2 * NW, nblk - 1), kN2);
                load_v(b + NW, vR);                 // raw rows of the next block fly under this block's MFMA chains ...
                f32x16 sN = qk(kN);
                softmax(b, sC);
                pv(sC, vC);
                transpose_v(vR, vC);                // ... and go through LDS once the PV chain has read the current ones
                sC = sN;
#pragma unroll
                for (int c = 0; c < 4; ++c) kN[c] = kN2[c];
                b += NW;
            }
            softmax(b, sC);
            pv(sC, vC);
        }
    }
    // join the two halves' row sums (same m in both halves by construction)
    l_run += __shfl_xor(l_run, 32, 64);

    // ---- merge the NW key splits through LDS
    if (lh == 0) {
        s_m[wave][li] = m_run;
        s_l[wave][li] = l_run;
    }
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
        const int d0 = 8 * rq + 4 * lh;  // lane (q, h) holds O[q][d0 .. d0+3] in regs 4rq .. 4rq+3
        *reinterpret_cast<float4 *>(&s_o[wave][li][d0]) = make_float4(o[4 * rq], o[4 * rq + 1], o[4 * rq + 2], o[4 * rq + 3]);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 256; e += 64 * NW) {   // 32 queries x 8 float4 columns
        const int q = e >> 3, d4 = (e & 7) * 4;
        float mm = s_m[0][q];
#pragma unroll
        for (int w = 1; w < NW; ++w) mm = fmaxf(mm, s_m[w][q]);
        float l = 0.f;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < NW; ++w) {  // fixed order: deterministic
            const float sc = fast_exp2(s_m[w][q] - mm);
            l += s_l[w][q] * sc;
            const float4 t = *reinterpret_cast<const float4 *>(&s_o[w][q][d4]);
            r.x += t.x * sc; r.y += t.y * sc; r.z += t.z * sc; r.w += t.w * sc;
        }
        const float inv = 1.0f / l;
        r.x *= inv; r.y *= inv; r.z *= inv; r.w *= inv;
        if (q0 + q < a.L) *reinterpret_cast<float4 *>(a.O + (size_t)(q0 + q) * a.ldo + hc + d4) = r;
    }
}

}  // namespace

extern "C" size_t cofi_attention_workspace(int L, int S, int H, int D) {
    (void)L; (void)S; (void)H; (void)D;
    return 0;  // split-KV partials are merged in LDS
}

static int attention_launch(AttnArgs a, int frames, hipStream_t stream) {
    // >= ~8 waves per CU (2048 in all) with the fewest key splits
    const long wgs = (long)cofi_cdiv(a.L, 32) * a.H * frames;
    const dim3 grid(cofi_cdiv(a.L, 32), a.H, frames);
    if (wgs * 2 >= 2048 && a.S >= 64 * 2)
        hipLaunchKernelGGL(attention_fwd_kernel<2>, grid, dim3(128), 0, stream, a);
    else if (wgs * 4 >= 2048 && a.S >= 64 * 4)
        hipLaunchKernelGGL(attention_fwd_kernel<4>, grid, dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(attention_fwd_kernel<8>, grid, dim3(512), 0, stream, a);
    return cofi_launch_status();
}

static int attention_check(const float *Q, int ldq, const float *K, int ldk, const float *V, int ldv, float *O, int ldo, int L, int S, int H,
                           int D, int frames) {
    if (!Q || !K || !V || !O || L <= 0 || S <= 0 || H <= 0 || frames <= 0) return COFI_EINVAL;
    if (D != 32) return COFI_EUNSUPPORTED;
    if ((ldq & 3) || (ldk & 3) || (ldo & 3) || ldq < H * D || ldk < H * D || ldv < H * D || ldo < H * D) return COFI_EINVAL;
    if ((ldv & 3) || ((uintptr_t)V & 15) || ((uintptr_t)Q & 15) || ((uintptr_t)K & 15) || ((uintptr_t)O & 15)) return COFI_EINVAL;
    return 0;
}

extern "C" int cofi_attention_fwd(const float *Q, int ldq, const float *K, int ldk, const float *V, int ldv, const float *q_colscale,
                                  float *O, int ldo, int L, int S, int H, int D, float scale, void *ws, size_t ws_bytes, int frames,
                                  cofi_stream_t stream) {
    (void)ws; (void)ws_bytes;
    if (int rc = attention_check(Q, ldq, K, ldk, V, ldv, O, ldo, L, S, H, D, frames)) return rc;
    if (q_colscale && ((uintptr_t)q_colscale & 15)) return COFI_EINVAL;
    AttnArgs a{Q, K, V, q_colscale, O, ldq, ldk, ldv, ldo, L, S, H, scale * 1.4426950408889634f, nullptr, 0, 0, 0.f};
    return attention_launch(a, frames, cofi_s(stream));
}

extern "C" int cofi_attention_fwd_colpart(const float *Q, int ldq, const float *K, int ldk, const float *V, int ldv, const float *q_colpart,
                                          int q_nslab, int q_ncols, float q_eps, float *O, int ldo, int L, int S, int H, int D,
                                          float scale, int frames, cofi_stream_t stream) {
    if (int rc = attention_check(Q, ldq, K, ldk, V, ldv, O, ldo, L, S, H, D, frames)) return rc;
    if (!q_colpart || q_nslab <= 0 || (q_nslab % frames) || q_ncols < H * D) return COFI_EINVAL;
    AttnArgs a{Q, K, V, nullptr, O, ldq, ldk, ldv, ldo, L, S, H, scale * 1.4426950408889634f, q_colpart, q_nslab / frames, q_ncols, q_eps};
    return attention_launch(a, frames, cofi_s(stream));
}
