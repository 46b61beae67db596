// icp_solve.hpp - device code shared by the ICP kernels (k_icp.hip, k_icp_lat.hip): TransformationEstimationSVD
// (pcl::umeyama with Eigen's two-sided Jacobi SVD restated in float32) from the 16 fixed-point moment sums, and the
// canonical 4x4 x point product.  Reference: pcl::IterativeClosestPoint as called at
// cuboid_detection/src/iterative_closest_point.cpp:170-178 and object_detection/src/object_pose_detection.cpp:220-228.
#pragma once
#include "common.hpp"

namespace cd {

struct Rot { float c, s; };
__device__ __forceinline__ Rot rot_mul(const Rot& a, const Rot& b) { return {a.c * b.c - a.s * b.s, a.c * b.s + a.s * b.c}; }
__device__ __forceinline__ Rot rot_T(const Rot& a) { return {a.c, -a.s}; }
__device__ __forceinline__ void apply_left(float M[3][3], int p, int q, const Rot& j) {
    for (int i = 0; i < 3; ++i) {
        const float x = M[p][i], y = M[q][i];
        M[p][i] = j.c * x + j.s * y;
        M[q][i] = -j.s * x + j.c * y;
    }
}
__device__ __forceinline__ void apply_right(float M[3][3], int p, int q, const Rot& j) {
    for (int i = 0; i < 3; ++i) {
        const float x = M[i][p], y = M[i][q];
        M[i][p] = j.c * x - j.s * y;
        M[i][q] = j.s * x + j.c * y;
    }
}
__device__ __forceinline__ Rot make_jacobi(float x, float y, float z) {
    if (y == 0.f) return {1.f, 0.f};
    const float tau = (x - z) / (2.0f * fabsf(y));
    const float w = sqrtf(tau * tau + 1.0f);
    const float t = tau > 0.f ? 1.0f / (tau + w) : 1.0f / (tau - w);
    const float sign_t = t > 0.f ? 1.0f : -1.0f;
    const float n = 1.0f / sqrtf(t * t + 1.0f);
    Rot r;
    r.s = -sign_t * (y / fabsf(y)) * fabsf(t) * n;
    r.c = n;
    return r;
}
// Eigen 3.2 JacobiSVD<Matrix3f>(ComputeFullU|ComputeFullV): two-sided Jacobi, float32
__device__ inline void jacobi_svd3(const float A[3][3], float U[3][3], float S[3], float V[3][3]) {
    const float precision = 2.0f * 1.1920928955078125e-07f;
    const float consider_zero = 2.0f * 1.401298464324817e-45f;
    float scale = 0.f;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) scale = fmaxf(scale, fabsf(A[i][j]));
    if (scale == 0.f) scale = 1.f;
    float W[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            W[i][j] = A[i][j] / scale;
            U[i][j] = V[i][j] = (i == j) ? 1.f : 0.f;
        }
    bool finished = false;
    for (int sweep = 0; sweep < 64 && !finished; ++sweep) {
        finished = true;
        for (int p = 1; p < 3; ++p)
            for (int q = 0; q < p; ++q) {
                const float thr = fmaxf(consider_zero, precision * fmaxf(fabsf(W[p][p]), fabsf(W[q][q])));
                if (fabsf(W[p][q]) > thr || fabsf(W[q][p]) > thr) {
                    finished = false;
                    const float m00 = W[p][p], m01 = W[p][q], m10 = W[q][p], m11 = W[q][q];
                    Rot rot1;
                    const float t = m00 + m11, d = m10 - m01;
                    if (t == 0.f) {
                        rot1.c = 0.f;
                        rot1.s = d > 0.f ? 1.f : -1.f;
                    } else {
                        const float u = d / t;
                        rot1.c = 1.0f / sqrtf(1.0f + u * u);
                        rot1.s = rot1.c * u;
                    }
                    const float n00 = rot1.c * m00 + rot1.s * m10, n01 = rot1.c * m01 + rot1.s * m11;
                    const float n11 = -rot1.s * m01 + rot1.c * m11;
                    const Rot j_right = make_jacobi(n00, n01, n11);
                    const Rot j_left = rot_mul(rot1, rot_T(j_right));
                    apply_left(W, p, q, j_left);
                    apply_right(U, p, q, rot_T(j_left));
                    apply_right(W, p, q, j_right);
                    apply_right(V, p, q, j_right);
                }
            }
    }
    for (int i = 0; i < 3; ++i) {
        const float a = fabsf(W[i][i]);
        S[i] = a;
        if (a != 0.f) {
            const float f = W[i][i] / a;
            for (int r = 0; r < 3; ++r) U[r][i] *= f;
        }
    }
    // Eigen's selection sort of the singular values (first maximum of S[i..2] goes to i; stops at a zero maximum), written out
    // with constant indices: a run-time column index would put U, V and S into scratch memory (three dependent round trips
    // per solve)
#define CD_SWAP_COLS(a, b)                                                  \
    {                                                                       \
        float t_ = S[a]; S[a] = S[b]; S[b] = t_;                            \
        _Pragma("unroll") for (int r = 0; r < 3; ++r) {                     \
            t_ = U[r][a]; U[r][a] = U[r][b]; U[r][b] = t_;                  \
            t_ = V[r][a]; V[r][a] = V[r][b]; V[r][b] = t_;                  \
        }                                                                   \
    }
    {
        int pos = 0;
        float mxv = S[0];
        if (S[1] > mxv) { mxv = S[1]; pos = 1; }
        if (S[2] > mxv) { mxv = S[2]; pos = 2; }
        if (mxv != 0.f) {
            if (pos == 1) CD_SWAP_COLS(0, 1)
            else if (pos == 2) CD_SWAP_COLS(0, 2)
            if (S[2] > S[1]) CD_SWAP_COLS(1, 2)   // (a zero maximum here means S[1] == S[2] == 0: nothing to swap either way)
        }
    }
#undef CD_SWAP_COLS
    for (int i = 0; i < 3; ++i) S[i] *= scale;
}
__device__ __forceinline__ float det3(const float m[3][3]) {
    return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
           m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
}
__device__ __forceinline__ double unfix(unsigned long long s, int shift) { return ldexp((double)(long long)s, -shift); }

// pcl::umeyama(src, dst, false) from the fixed-point moments.  A: [0..2] sum p, [3..5] sum q,
// [6..14] sum q_a p_b, [15] sum d2.
__device__ inline void umeyama_from_moments(const unsigned long long* A, int n, float T[16]) {
    double mp[3], mq[3];
    for (int a = 0; a < 3; ++a) {
        mp[a] = unfix(A[a], FIX_SHIFT) / (double)n;
        mq[a] = unfix(A[3 + a], FIX_SHIFT) / (double)n;
    }
    float sigma[3][3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) sigma[a][b] = (float)(unfix(A[6 + 3 * a + b], FIX_SHIFT) / (double)n - mq[a] * mp[b]);
    float U[3][3], S[3], V[3][3];
    jacobi_svd3(sigma, U, S, V);
    float sd[3] = {1.f, 1.f, 1.f};
    if (det3(sigma) < 0.f) sd[2] = -1.f;
    int rank = 0;
    for (int i = 0; i < 3; ++i)
        if (!(fabsf(S[i]) <= fabsf(S[0]) * 1e-5f)) ++rank;
    if (rank == 2) {
        sd[0] = 1.f; sd[1] = 1.f;
        sd[2] = (det3(U) * det3(V) > 0.f) ? 1.f : -1.f;
    }
    float R[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            R[i][j] = ((U[i][0] * sd[0]) * V[j][0] + (U[i][1] * sd[1]) * V[j][1]) + (U[i][2] * sd[2]) * V[j][2];
    const float mpf[3] = {(float)mp[0], (float)mp[1], (float)mp[2]};
    const float mqf[3] = {(float)mq[0], (float)mq[1], (float)mq[2]};
    for (int i = 0; i < 16; ++i) T[i] = 0.f;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[4 * i + j] = R[i][j];
        T[4 * i + 3] = mqf[i] - ((R[i][0] * mpf[0] + R[i][1] * mpf[1]) + R[i][2] * mpf[2]);
    }
    T[15] = 1.f;
}

__device__ __forceinline__ void xform(const float* T, float x, float y, float z, float& ox, float& oy, float& oz) {
    ox = ((T[0] * x + T[1] * y) + T[2] * z) + T[3];
    oy = ((T[4] * x + T[5] * y) + T[6] * z) + T[7];
    oz = ((T[8] * x + T[9] * y) + T[10] * z) + T[11];
}

}  // namespace cd
