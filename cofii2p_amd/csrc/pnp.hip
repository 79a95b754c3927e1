// Camera pose from the fine 2-D / 3-D matches: RANSAC over P3P hypotheses + Levenberg-Marquardt refit, all on the device.
// Replaces the pose step of the reference's evaluation (evaluation/eval_all.py:107):
//     cv2.solvePnPRansac(cameraMatrix=K, imagePoints=fine_xy.T, objectPoints=coarse_pc_points, iterationsCount=10000, distCoeffs=None)
// i.e. reprojection threshold 8 px, minimal-set hypotheses, best consensus set, iterative (LM) refit on its inliers.  OpenCV is a
// third-party dependency that is absent here: parity with it is UNPINNED; the restatement this kernel is checked against is
// oracle/pnp_oracle.py (same counter-based samples, same algorithm), which is validated geometrically.
//
// Kernel 1: one wave per HPW = 16 hypotheses.  Lane h < 16 draws 4 correspondences, solves P3P (Grunert: the depth ratios
//   satisfy a quartic whose coefficients come from polynomial arithmetic; closed-form Ferrari roots polished by Newton, fp64) on
//   three of them and lets the fourth choose among the up-to-four poses.  Then the wave scores its 16 poses one after the other,
//   lanes striding over the correspondences (fp32), and publishes (inliers << 32 | ~hypothesis) with an integer atomicMax:
//   the winner is the hypothesis with most inliers, lowest id on ties - deterministic.
// Kernel 2: one workgroup refits the winner on its inliers: LM on the reprojection error with left-multiplied se(3) increments,
//   normal equations reduced in a fixed order (fp64), 6x6 solve by one thread.
#include "common.h"

namespace {

constexpr int HPW = 16;  // hypotheses per wave

__device__ __forceinline__ unsigned hash_u32(unsigned seed, unsigned hyp, unsigned j) {
    unsigned x = seed * 0x9E3779B1u + hyp * 0x85EBCA77u + j * 0xC2B2AE3Du + 0x27D4EB2Fu;
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

struct Cam { float fx, fy, cx, cy; };

// real roots of q0 + q1 x + ... + q4 x^4 (q4 != 0): Ferrari, then Newton on the original quartic
__device__ int quartic_real_roots(const double q[5], double out[4]) {
    const double a = q[3] / q[4], b = q[2] / q[4], c = q[1] / q[4], d = q[0] / q[4];
    const double a2 = a * a;
    const double p = b - 0.375 * a2, qq = c - 0.5 * a * b + 0.125 * a2 * a, r = d - 0.25 * a * c + 0.0625 * a2 * b - 0.01171875 * a2 * a2;
    double y[4];
    int n = 0;
    const double scale = fabs(p) + fabs(r) + 1.0;
    if (fabs(qq) < 1e-12 * scale) {  // biquadratic
        const double disc = p * p - 4.0 * r;
        if (disc >= 0.0) {
            const double sd = sqrt(disc);
            const double z1 = 0.5 * (-p + sd), z2 = 0.5 * (-p - sd);
            if (z1 >= 0.0) { y[n++] = sqrt(z1); y[n++] = -sqrt(z1); }
            if (z2 >= 0.0) { y[n++] = sqrt(z2); y[n++] = -sqrt(z2); }
        }
    } else {
        // resolvent cubic m^3 + p m^2 + (p^2/4 - r) m - q^2/8 = 0: largest real root (positive because q != 0)
        const double c2 = p, c1 = 0.25 * p * p - r, c0 = -0.125 * qq * qq;
        const double P = c1 - c2 * c2 / 3.0, Q = 2.0 * c2 * c2 * c2 / 27.0 - c2 * c1 / 3.0 + c0;
        const double disc = 0.25 * Q * Q + P * P * P / 27.0;
        double t;
        if (disc >= 0.0) {
            const double sd = sqrt(disc);
            t = cbrt(-0.5 * Q + sd) + cbrt(-0.5 * Q - sd);
        } else {
            const double rr = sqrt(-P / 3.0);
            double arg = 1.5 * Q / (P * rr);
            arg = arg > 1.0 ? 1.0 : (arg < -1.0 ? -1.0 : arg);
            t = 2.0 * rr * cos(acos(arg) / 3.0);
        }
        double m = t - c2 / 3.0;
        for (int it = 0; it < 3; ++it) {  // polish the cubic root
            const double f = ((m + c2) * m + c1) * m + c0, df = (3.0 * m + 2.0 * c2) * m + c1;
            if (df != 0.0) m -= f / df;
        }
        if (m > 0.0) {
            const double s = sqrt(2.0 * m), h = 0.5 * p + m, g = qq / (2.0 * s);
            const double d1 = s * s - 4.0 * (h + g), d2 = s * s - 4.0 * (h - g);
            if (d1 >= 0.0) { const double sd = sqrt(d1); y[n++] = 0.5 * (s + sd); y[n++] = 0.5 * (s - sd); }
            if (d2 >= 0.0) { const double sd = sqrt(d2); y[n++] = 0.5 * (-s + sd); y[n++] = 0.5 * (-s - sd); }
        }
    }
    for (int i = 0; i < n; ++i) {
        double x = y[i] - 0.25 * a;
        for (int it = 0; it < 3; ++it) {
            const double f = (((q[4] * x + q[3]) * x + q[2]) * x + q[1]) * x + q[0];
            const double df = ((4.0 * q[4] * x + 3.0 * q[3]) * x + 2.0 * q[2]) * x + q[1];
            if (df != 0.0) x -= f / df;
        }
        out[i] = x;
    }
    return n;
}

__device__ __forceinline__ void cross3(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// orthonormal frame of a triangle: columns e1 = (A1-A0)/|.|, e2 = e3 x e1, e3 = e1 x (A2-A0) / |.|
__device__ bool tri_frame(const double A[3][3], double F[3][3]) {
    double e1[3] = {A[1][0] - A[0][0], A[1][1] - A[0][1], A[1][2] - A[0][2]};
    double w[3] = {A[2][0] - A[0][0], A[2][1] - A[0][1], A[2][2] - A[0][2]};
    const double n1 = sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
    double e3[3];
    cross3(e1, w, e3);
    const double n3 = sqrt(e3[0] * e3[0] + e3[1] * e3[1] + e3[2] * e3[2]);
    if (n1 < 1e-12 || n3 < 1e-12) return false;
    for (int i = 0; i < 3; ++i) { e1[i] /= n1; e3[i] /= n3; }
    double e2[3];
    cross3(e3, e1, e2);
    for (int i = 0; i < 3; ++i) { F[i][0] = e1[i]; F[i][1] = e2[i]; F[i][2] = e3[i]; }
    return true;
}

// P3P on (X[0..2], bearing f[0..2]); the 4th correspondence (X4, uv4) picks the pose.  pose = R row-major (9) | t (3)
__device__ bool p3p_pick(const double X[3][3], const double f[3][3], const double X4[3], double u4, double v4, const Cam &cam,
                         double pose[12]) {
    auto d2 = [&](int i, int j) {
        const double x = X[i][0] - X[j][0], y = X[i][1] - X[j][1], z = X[i][2] - X[j][2];
        return x * x + y * y + z * z;
    };
    const double a2 = d2(1, 2), b2 = d2(0, 2), c2 = d2(0, 1);
    if (fmin(a2, fmin(b2, c2)) < 1e-18) return false;
    auto dot = [&](int i, int j) { return f[i][0] * f[j][0] + f[i][1] * f[j][1] + f[i][2] * f[j][2]; };
    const double ca = dot(1, 2), cb = dot(0, 2), cg = dot(0, 1);
    // polynomials in v = s3/s1, lowest degree first (see oracle/pnp_oracle.py::p3p_grunert)
    const double w[3] = {1.0, -2.0 * cb, 1.0};
    const double N[3] = {(a2 - c2) * w[0] + b2, (a2 - c2) * w[1], (a2 - c2) * w[2] - b2};
    const double D[2] = {2.0 * b2 * cg, -2.0 * b2 * ca};
    const double D2[3] = {D[0] * D[0], 2.0 * D[0] * D[1], D[1] * D[1]};
    double N2[5] = {0, 0, 0, 0, 0}, ND[4] = {0, 0, 0, 0}, WD2[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { N2[i + j] += N[i] * N[j]; WD2[i + j] += w[i] * D2[j]; }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 2; ++j) ND[i + j] += N[i] * D[j];
    double q[5];
    for (int k = 0; k < 5; ++k) q[k] = b2 * ((k < 3 ? D2[k] : 0.0) + N2[k] - 2.0 * cg * (k < 4 ? ND[k] : 0.0)) - c2 * WD2[k];
    const double qs = fabs(q[0]) + fabs(q[1]) + fabs(q[2]) + fabs(q[3]) + fabs(q[4]);
    if (!(fabs(q[4]) > 1e-14 * qs)) return false;
    double roots[4];
    const int nr = quartic_real_roots(q, roots);
    double best = 1e300;
    bool found = false;
    for (int i = 0; i < nr; ++i) {
        const double v = roots[i];
        const double den = D[0] + D[1] * v;
        if (!(v > 0.0) || fabs(den) < 1e-12) continue;
        const double u = (N[0] + N[1] * v + N[2] * v * v) / den;
        const double wv = 1.0 - 2.0 * v * cb + v * v;
        if (!(u > 0.0) || !(wv > 0.0)) continue;
        const double s1 = sqrt(b2 / wv), s[3] = {s1, u * s1, v * s1};
        double C[3][3], Fp[3][3], Fc[3][3];
        for (int k = 0; k < 3; ++k)
            for (int e = 0; e < 3; ++e) C[k][e] = s[k] * f[k][e];
        if (!tri_frame(X, Fp) || !tri_frame(C, Fc)) continue;
        double R[9], t[3];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) R[3 * r + c] = Fc[r][0] * Fp[c][0] + Fc[r][1] * Fp[c][1] + Fc[r][2] * Fp[c][2];
        for (int r = 0; r < 3; ++r) t[r] = C[0][r] - (R[3 * r] * X[0][0] + R[3 * r + 1] * X[0][1] + R[3 * r + 2] * X[0][2]);
        // reprojection of the 4th correspondence
        const double y0 = R[0] * X4[0] + R[1] * X4[1] + R[2] * X4[2] + t[0], y1 = R[3] * X4[0] + R[4] * X4[1] + R[5] * X4[2] + t[1],
                     y2 = R[6] * X4[0] + R[7] * X4[1] + R[8] * X4[2] + t[2];
        if (!(y2 > 1e-6)) continue;
        const double eu = cam.fx * y0 / y2 + cam.cx - u4, ev = cam.fy * y1 / y2 + cam.cy - v4;
        const double e = eu * eu + ev * ev;
        if (e < best) {
            best = e;
            found = true;
            for (int k = 0; k < 9; ++k) pose[k] = R[k];
            for (int k = 0; k < 3; ++k) pose[9 + k] = t[k];
        }
    }
    return found;
}

__device__ __forceinline__ bool inlier(const float p[12], const float *X, const float *uv, const Cam &cam, float thr2) {
    const float y0 = p[0] * X[0] + p[1] * X[1] + p[2] * X[2] + p[9], y1 = p[3] * X[0] + p[4] * X[1] + p[5] * X[2] + p[10],
                y2 = p[6] * X[0] + p[7] * X[1] + p[8] * X[2] + p[11];
    const float eu = cam.fx * y0 / y2 + cam.cx - uv[0], ev = cam.fy * y1 / y2 + cam.cy - uv[1];
    return y2 > 1e-6f && eu * eu + ev * ev <= thr2;
}

__global__ __launch_bounds__(64) void pnp_hypotheses_kernel(const float *obj, const float *img, const int32_t *count_dev, int n_max, Cam cam,
                                                            int iters, float thr2, unsigned seed, float *poses,
                                                            unsigned long long *best_key) {
    const int lane = threadIdx.x;
    const int n = count_dev ? min(*count_dev, n_max) : n_max;
    if (n < 4) return;
    const int hyp = blockIdx.x * HPW + lane;
    float pose_f[12];
    bool ok = false;
    if (lane < HPW && hyp < iters) {
        int idx[4];
        for (int j = 0; j < 4; ++j) {
            int c = (int)(hash_u32(seed, (unsigned)hyp, (unsigned)j) % (unsigned)n);
            bool again = true;
            while (again) {
                again = false;
                for (int k = 0; k < j; ++k)
                    if (idx[k] == c) { c = (c + 1) % n; again = true; }
            }
            idx[j] = c;
        }
        double X[3][3], f[3][3];
        for (int k = 0; k < 3; ++k) {
            for (int e = 0; e < 3; ++e) X[k][e] = (double)obj[3 * idx[k] + e];
            const double bx = ((double)img[2 * idx[k]] - cam.cx) / cam.fx, by = ((double)img[2 * idx[k] + 1] - cam.cy) / cam.fy;
            const double inv = 1.0 / sqrt(bx * bx + by * by + 1.0);
            f[k][0] = bx * inv; f[k][1] = by * inv; f[k][2] = inv;
        }
        const double X4[3] = {(double)obj[3 * idx[3]], (double)obj[3 * idx[3] + 1], (double)obj[3 * idx[3] + 2]};
        double pose[12];
        ok = p3p_pick(X, f, X4, (double)img[2 * idx[3]], (double)img[2 * idx[3] + 1], cam, pose);
        if (ok)
            for (int k = 0; k < 12; ++k) {
                pose_f[k] = (float)pose[k];
                poses[(size_t)hyp * 12 + k] = pose_f[k];
            }
    }
    // score the wave's hypotheses one after the other: lanes stride over the correspondences
    unsigned long long wave_best = 0ull;
    for (int h = 0; h < HPW; ++h) {
        if (!__shfl((int)ok, h, 64)) continue;   // uniform
        float p[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) p[k] = __shfl(pose_f[k], h, 64);
        int cnt = 0;
        for (int i = lane; i < n; i += 64) cnt += inlier(p, obj + 3 * i, img + 2 * i, cam, thr2) ? 1 : 0;
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        const unsigned long long key = ((unsigned long long)(unsigned)cnt << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)(blockIdx.x * HPW + h));
        wave_best = key > wave_best ? key : wave_best;
    }
    if (lane == 0 && wave_best) atomicMax(best_key, wave_best);
}

// ---- refit: Levenberg-Marquardt on the inliers of the winning hypothesis
constexpr int NACC = 28;  // 21 (upper JtJ) + 6 (Jt r) + 1 (cost)

__device__ void accumulate(const double R[9], const double t[3], const float *X, const float *uv, const Cam &cam, double acc[NACC]) {
    const double x = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0], y = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1],
                 z = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
    const double iz = 1.0 / z;
    const double r0 = cam.fx * x * iz + cam.cx - uv[0], r1 = cam.fy * y * iz + cam.cy - uv[1];
    // d(pi)/dY (2x3) times [-[Y]x | I] (3x6)
    const double a00 = cam.fx * iz, a02 = -cam.fx * x * iz * iz, a11 = cam.fy * iz, a12 = -cam.fy * y * iz * iz;
    // -[Y]x = [[0, z, -y], [-z, 0, x], [y, -x, 0]]
    const double J0[6] = {a02 * y, a00 * z - a02 * x, -a00 * y, a00, 0.0, a02};
    const double J1[6] = {-a11 * z + a12 * y, -a12 * x, a11 * x, 0.0, a11, a12};
    int k = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) acc[k++] += J0[i] * J0[j] + J1[i] * J1[j];
    for (int i = 0; i < 6; ++i) acc[21 + i] += J0[i] * r0 + J1[i] * r1;
    acc[27] += r0 * r0 + r1 * r1;
}

__global__ __launch_bounds__(256) void pnp_refine_kernel(const float *obj, const float *img, const int32_t *count_dev, int n_max, Cam cam,
                                                         float thr2, const float *poses, const unsigned long long *best_key, int lm_iters,
                                                         float *pose_out, int32_t *result /* [0] success, [1] inliers, [2] hypothesis */,
                                                         uint8_t *mask) {
    __shared__ double s_red[4][NACC];
    __shared__ double s_sys[NACC];     // reduced normal equations of the candidate pose
    __shared__ double s_pose[12], s_cand[12];
    __shared__ int s_cnt[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = count_dev ? min(*count_dev, n_max) : n_max;
    const unsigned long long key = *best_key;
    const int cnt_best = (int)(key >> 32);
    const int hyp = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
    for (int i = tid; i < n_max; i += 256) mask[i] = 0;
    if (n < 4 || key == 0ull || cnt_best < 4) {
        if (tid == 0) {
            result[0] = 0; result[1] = 0; result[2] = -1;
            for (int k = 0; k < 12; ++k) pose_out[k] = (k == 0 || k == 4 || k == 8) ? 1.f : 0.f;
        }
        return;
    }
    float pf[12];
    for (int k = 0; k < 12; ++k) pf[k] = poses[(size_t)hyp * 12 + k];
    if (tid < 12) s_pose[tid] = (double)pf[tid];
    // inlier mask of the RANSAC model (what cv2 returns as `inliers`)
    int c = 0;
    for (int i = tid; i < n; i += 256) {
        const bool in = inlier(pf, obj + 3 * i, img + 2 * i, cam, thr2);
        mask[i] = in ? 1 : 0;
        c += in ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) s_cnt[wv] = c;
    __syncthreads();
    const int n_in = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];

    // normal equations at `pose` over the inliers, reduced in a fixed order: thread-strided partials -> wave butterflies -> 4 waves
    auto reduce_at = [&](const double *pose) {
        double acc[NACC];
        for (int k = 0; k < NACC; ++k) acc[k] = 0.0;
        double R[9], t[3];
        for (int k = 0; k < 9; ++k) R[k] = pose[k];
        for (int k = 0; k < 3; ++k) t[k] = pose[9 + k];
        for (int i = tid; i < n; i += 256)
            if (mask[i]) accumulate(R, t, obj + 3 * i, img + 2 * i, cam, acc);
        for (int k = 0; k < NACC; ++k) {
            const double v = wave_sum_d(acc[k]);
            if (lane == 0) s_red[wv][k] = v;
        }
        __syncthreads();
        if (tid < NACC) s_sys[tid] = (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]);
        __syncthreads();
    };

    __syncthreads();
    reduce_at(s_pose);
    double H[21], g[6], cost = 0.0, lam = 1e-3;   // thread 0's copy of the accepted system
    if (tid == 0) {
        for (int k = 0; k < 21; ++k) H[k] = s_sys[k];
        for (int k = 0; k < 6; ++k) g[k] = s_sys[21 + k];
        cost = s_sys[27];
    }
    for (int it = 0; it < lm_iters; ++it) {
        if (tid == 0) {
            // (H + lam diag(H) + eps I) d = -g by Gaussian elimination with partial pivoting
            double A[6][7];
            int k = 0;
            for (int i = 0; i < 6; ++i)
                for (int j = i; j < 6; ++j) { A[i][j] = H[k]; A[j][i] = H[k]; ++k; }
            for (int i = 0; i < 6; ++i) { A[i][i] += lam * A[i][i] + 1e-12; A[i][6] = -g[i]; }
            bool okk = true;
            for (int col = 0; col < 6; ++col) {
                int piv = col;
                for (int r = col + 1; r < 6; ++r)
                    if (fabs(A[r][col]) > fabs(A[piv][col])) piv = r;
                if (fabs(A[piv][col]) < 1e-300) { okk = false; break; }
                if (piv != col)
                    for (int j = 0; j < 7; ++j) { const double tmp = A[col][j]; A[col][j] = A[piv][j]; A[piv][j] = tmp; }
                for (int r = col + 1; r < 6; ++r) {
                    const double fct = A[r][col] / A[col][col];
                    for (int j = col; j < 7; ++j) A[r][j] -= fct * A[col][j];
                }
            }
            double d[6] = {0, 0, 0, 0, 0, 0};
            if (okk)
                for (int i = 5; i >= 0; --i) {
                    double sacc = A[i][6];
                    for (int j = i + 1; j < 6; ++j) sacc -= A[i][j] * d[j];
                    d[i] = sacc / A[i][i];
                }
            // candidate = exp(omega) * pose, t' = exp(omega) t + delta (Rodrigues)
            const double wx = d[0], wy = d[1], wz = d[2];
            const double th = sqrt(wx * wx + wy * wy + wz * wz);
            double E[9];
            const double W[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
            double W2[9];
            for (int r = 0; r < 3; ++r)
                for (int cc = 0; cc < 3; ++cc) W2[3 * r + cc] = W[3 * r] * W[cc] + W[3 * r + 1] * W[3 + cc] + W[3 * r + 2] * W[6 + cc];
            const double ka = th < 1e-12 ? 1.0 : sin(th) / th, kb = th < 1e-12 ? 0.0 : (1.0 - cos(th)) / (th * th);
            for (int k2 = 0; k2 < 9; ++k2) E[k2] = ((k2 % 4) == 0 ? 1.0 : 0.0) + ka * W[k2] + kb * W2[k2];
            for (int r = 0; r < 3; ++r) {
                for (int cc = 0; cc < 3; ++cc)
                    s_cand[3 * r + cc] = E[3 * r] * s_pose[cc] + E[3 * r + 1] * s_pose[3 + cc] + E[3 * r + 2] * s_pose[6 + cc];
                s_cand[9 + r] = E[3 * r] * s_pose[9] + E[3 * r + 1] * s_pose[10] + E[3 * r + 2] * s_pose[11] + d[3 + r];
            }
        }
        __syncthreads();
        reduce_at(s_cand);
        if (tid == 0) {
            const double cn = s_sys[27];
            if (cn == cn && cn < cost) {   // accept
                for (int k = 0; k < 12; ++k) s_pose[k] = s_cand[k];
                for (int k = 0; k < 21; ++k) H[k] = s_sys[k];
                for (int k = 0; k < 6; ++k) g[k] = s_sys[21 + k];
                cost = cn;
                lam = fmax(lam * 0.1, 1e-9);
            } else {
                lam = fmin(lam * 10.0, 1e6);
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        for (int k = 0; k < 12; ++k) pose_out[k] = (float)s_pose[k];
        result[0] = 1; result[1] = n_in; result[2] = hyp;
    }
}

}  // namespace

extern "C" size_t cofi_pnp_ransac_workspace(int iterations) {
    return iterations > 0 ? 64 + (size_t)iterations * 12 * sizeof(float) : 0;
}

extern "C" int cofi_pnp_ransac(const float *obj, const float *img, const int32_t *count_dev, int n_max, float fx, float fy, float cx,
                               float cy, int iterations, float reproj_err, unsigned seed, int refine_iters, void *ws, size_t ws_bytes,
                               float *pose, int32_t *result, uint8_t *inlier_mask, cofi_stream_t stream) {
    if (!obj || !img || !pose || !result || !inlier_mask || n_max <= 0 || iterations <= 0 || !(reproj_err > 0.f) || !(fx > 0.f) || !(fy > 0.f) ||
        refine_iters < 0)
        return COFI_EINVAL;
    if (!ws || ws_bytes < cofi_pnp_ransac_workspace(iterations) || ((uintptr_t)ws & 15)) return COFI_EWORKSPACE;
    hipStream_t s = cofi_s(stream);
    unsigned long long *key = (unsigned long long *)ws;
    float *poses = (float *)((char *)ws + 64);
    if (hipError_t e = hipMemsetAsync(key, 0, 8, s); e != hipSuccess) return (int)e;
    const Cam cam{fx, fy, cx, cy};
    hipLaunchKernelGGL(pnp_hypotheses_kernel, dim3(cofi_cdiv(iterations, HPW)), dim3(64), 0, s, obj, img, count_dev, n_max, cam, iterations,
                       reproj_err * reproj_err, seed, poses, key);
    hipLaunchKernelGGL(pnp_refine_kernel, dim3(1), dim3(256), 0, s, obj, img, count_dev, n_max, cam, reproj_err * reproj_err, poses, key,
                       refine_iters, pose, result, inlier_mask);
    return cofi_launch_status();
}
