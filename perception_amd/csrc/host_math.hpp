// host_math.hpp - the few scalar computations of the path that stay on the host.
//  * pcl::eigen33 (closed-form smallest eigenpair, float32) for the plane refit: it uses
//    atan2/cos/sin, whose device libm results are not guaranteed identical to the host's, and
//    it is one 3x3 problem per frame.
//  * PCL's adaptive RANSAC stop rule (uses log/pow in double).
//  * Matrix4d inverse (icp.cpp:179), tf::Matrix3x3::getRotation (icp.cpp:62-67), bbox corners
//    (icp.cpp:99-110).
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace cd {
namespace hm {

inline double unfix(uint64_t s, int shift) { return std::ldexp((double)(int64_t)s, -shift); }

// largest float f with (double)f <= v / smallest float f with (double)f >= v
inline float fold_le(double v) {
    float f = (float)v;
    if ((double)f > v) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
    return f;
}
inline float fold_ge(double v) {
    float f = (float)v;
    if ((double)f < v) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
    return f;
}

inline void roots2(float b, float c, float r[3]) {
    r[0] = 0.f;
    float d = b * b - 4.0f * c;
    if (d < 0.0f) d = 0.0f;
    const float sd = std::sqrt(d);
    r[2] = 0.5f * (b + sd);
    r[1] = 0.5f * (b - sd);
}
// pcl::computeRoots
inline void roots3(const float m[3][3], float r[3]) {
    const float c0 = m[0][0] * m[1][1] * m[2][2] + 2.0f * m[0][1] * m[0][2] * m[1][2] - m[0][0] * m[1][2] * m[1][2] -
                     m[1][1] * m[0][2] * m[0][2] - m[2][2] * m[0][1] * m[0][1];
    const float c1 = m[0][0] * m[1][1] - m[0][1] * m[0][1] + m[0][0] * m[2][2] - m[0][2] * m[0][2] + m[1][1] * m[2][2] -
                     m[1][2] * m[1][2];
    const float c2 = m[0][0] + m[1][1] + m[2][2];
    if (std::fabs(c0) < std::numeric_limits<float>::epsilon()) {
        roots2(c2, c1, r);
        return;
    }
    const float inv3 = 1.0f / 3.0f;
    const float sqrt3 = std::sqrt(3.0f);
    const float c2_3 = c2 * inv3;
    float a_3 = (c1 - c2 * c2_3) * inv3;
    if (a_3 > 0.0f) a_3 = 0.0f;
    const float half_b = 0.5f * (c0 + c2_3 * (2.0f * c2_3 * c2_3 - c1));
    float q = half_b * half_b + a_3 * a_3 * a_3;
    if (q > 0.0f) q = 0.0f;
    const float rho = std::sqrt(-a_3);
    const float theta = std::atan2(std::sqrt(-q), half_b) * inv3;
    const float ct = std::cos(theta), sn = std::sin(theta);
    r[0] = c2_3 + 2.0f * rho * ct;
    r[1] = c2_3 - rho * (ct + sqrt3 * sn);
    r[2] = c2_3 - rho * (ct - sqrt3 * sn);
    if (r[0] >= r[1]) std::swap(r[0], r[1]);
    if (r[1] >= r[2]) {
        std::swap(r[1], r[2]);
        if (r[0] >= r[1]) std::swap(r[0], r[1]);
    }
    if (r[0] <= 0.0f) roots2(c2, c1, r);
}
inline void cross(const float a[3], const float b[3], float o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
// pcl::eigen33(mat, eigenvalue, eigenvector): eigenvector of the smallest eigenvalue
inline void eigen33_smallest(const float cov[3][3], float v[3]) {
    float scale = 0.f;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) scale = std::fmax(scale, std::fabs(cov[i][j]));
    if (scale <= std::numeric_limits<float>::min()) scale = 1.0f;
    float s[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) s[i][j] = cov[i][j] / scale;
    float r[3];
    roots3(s, r);
    for (int i = 0; i < 3; ++i) s[i][i] -= r[0];
    float v1[3], v2[3], v3[3];
    cross(s[0], s[1], v1);
    cross(s[0], s[2], v2);
    cross(s[1], s[2], v3);
    const float l1 = (v1[0] * v1[0] + v1[1] * v1[1]) + v1[2] * v1[2];
    const float l2 = (v2[0] * v2[0] + v2[1] * v2[1]) + v2[2] * v2[2];
    const float l3 = (v3[0] * v3[0] + v3[1] * v3[1]) + v3[2] * v3[2];
    const float* w;
    float l;
    if (l1 >= l2 && l1 >= l3) { w = v1; l = l1; }
    else if (l2 >= l1 && l2 >= l3) { w = v2; l = l2; }
    else { w = v3; l = l3; }
    const float sl = std::sqrt(l);
    v[0] = w[0] / sl; v[1] = w[1] / sl; v[2] = w[2] / sl;
}

// SampleConsensusModelPlane::optimizeModelCoefficients from the 9 fixed-point moments + count
inline void plane_refit_from_moments(const uint64_t S[10], const float model[4], float out[4]) {
    const uint64_t cnt = S[9];
    if (cnt < 4) { std::memcpy(out, model, 16); return; }
    float a[9];
    const double n = (double)cnt;
    for (int k = 0; k < 9; ++k) a[k] = (float)(unfix(S[k], 32) / n);
    float cov[3][3];
    cov[0][0] = a[0] - a[6] * a[6];
    cov[0][1] = a[1] - a[6] * a[7];
    cov[0][2] = a[2] - a[6] * a[8];
    cov[1][1] = a[3] - a[7] * a[7];
    cov[1][2] = a[4] - a[7] * a[8];
    cov[2][2] = a[5] - a[8] * a[8];
    cov[1][0] = cov[0][1]; cov[2][0] = cov[0][2]; cov[2][1] = cov[1][2];
    float ev[3];
    eigen33_smallest(cov, ev);
    out[0] = ev[0]; out[1] = ev[1]; out[2] = ev[2];
    out[3] = -1.0f * ((out[0] * a[6] + out[1] * a[7]) + out[2] * a[8]);
}

// isModelValid of pcl::SampleConsensusModelPerpendicularPlane / ...ParallelPlane (PCL 1.7.2), evaluated on the
// host so that the CPU restatement and the GPU path share one libm (acos, sin).  type: CD_PLANE* of cuboid_hip.h.
// Float 4-vector dot products / norms use the canonical association (x + y) + z (w = 0).
inline bool plane_model_valid(int type, const float m[4], const float axis[3], double eps_angle) {
    if (type == 0 || !(eps_angle > 0.0)) return true;
    if (type == 1) {   // perpendicular plane: angle(axis, normal) folded into [0, pi/2] must be <= eps
        const float dot = (axis[0] * m[0] + axis[1] * m[1]) + axis[2] * m[2];
        const float n1 = (axis[0] * axis[0] + axis[1] * axis[1]) + axis[2] * axis[2];
        const float n2 = (m[0] * m[0] + m[1] * m[1]) + m[2] * m[2];
        double rad = dot / std::sqrt(n1 * n2);          // pcl::getAngle3D: float expression, clamped as double
        if (rad < -1.0) rad = -1.0; else if (rad > 1.0) rad = 1.0;
        double angle_diff = std::fabs(std::acos(rad));
        angle_diff = std::min(angle_diff, M_PI - angle_diff);
        return !(angle_diff > eps_angle);
    }
    // parallel plane: |axis . normalized(normal)| must be <= |sin(eps)|
    const float nrm = std::sqrt((m[0] * m[0] + m[1] * m[1]) + m[2] * m[2]);
    const float c[3] = {m[0] / nrm, m[1] / nrm, m[2] / nrm};
    const float dot = (axis[0] * c[0] + axis[1] * c[1]) + axis[2] * c[2];
    return !((double)std::fabs(dot) > std::fabs(std::sin(eps_angle)));
}

// surface_normal_estimation.cpp:207-222: make the frame right-handed, put the origin on plane 0's centroid moved
// along normal 0 to the level of plane 1's centroid, assemble Rt = [n2 n1 n0 c].  normals/mids: sorted planes.
// The third normal is flipped IN PLACE (the node publishes the flipped vector's plane as is; only Rt sees it).
inline void surface_frame(const float normals[3][4], const float mids[3][4], float Rt[16]) {
    const float* n0 = normals[0];
    const float* n1 = normals[1];
    float n2[3] = {normals[2][0], normals[2][1], normals[2][2]};
    const float cr[3] = {n1[1] * n0[2] - n1[2] * n0[1], n1[2] * n0[0] - n1[0] * n0[2], n1[0] * n0[1] - n1[1] * n0[0]};   // n1 x n0
    const float trip = (n2[0] * cr[0] + n2[1] * cr[1]) + n2[2] * cr[2];
    if (trip < 0.f) { n2[0] = -n2[0]; n2[1] = -n2[1]; n2[2] = -n2[2]; }
    const float d[3] = {mids[0][0] - mids[1][0], mids[0][1] - mids[1][1], mids[0][2] - mids[1][2]};
    const float proj = (n0[0] * d[0] + n0[1] * d[1]) + n0[2] * d[2];
    const float cen[3] = {mids[0][0] - proj * n0[0], mids[0][1] - proj * n0[1], mids[0][2] - proj * n0[2]};
    for (int r = 0; r < 3; ++r) {
        Rt[4 * r + 0] = n2[r]; Rt[4 * r + 1] = n1[r]; Rt[4 * r + 2] = n0[r]; Rt[4 * r + 3] = cen[r];
    }
    Rt[12] = 0.f; Rt[13] = 0.f; Rt[14] = 0.f; Rt[15] = 1.f;
}

// pcl::RandomSampleConsensus::computeModel's sequential bookkeeping, fed with the batched counts
struct RansacReplay {
    int iterations = 0, skipped = 0, best = -INT32_MAX, best_h = -1, pos = 0;
    double k = 1.0;
    bool finished = false;
    // consume hypotheses [pos, n_avail); returns true when PCL's loop has ended
    bool consume(const int* counts, const int* valid, int n_avail, int n_points, int max_iter, double prob, bool exhausted) {
        const double log_probability = std::log(1.0 - prob);
        const double one_over = 1.0 / (double)n_points;
        const int max_skip = max_iter * 10;
        while (!finished) {
            if (!((double)iterations < k && skipped < max_skip)) { finished = true; break; }
            if (pos >= n_avail) {
                if (exhausted) finished = true;   // "No samples could be selected!" -> break
                break;
            }
            if (!valid[pos]) { ++skipped; ++pos; continue; }
            const int c = counts[pos];
            if (c > best) {
                best = c;
                best_h = pos;
                const double w = (double)best * one_over;
                double p = 1.0 - std::pow(w, 3.0);
                p = std::fmax(std::numeric_limits<double>::epsilon(), p);
                p = std::fmin(1.0 - std::numeric_limits<double>::epsilon(), p);
                k = log_probability / std::log(p);
            }
            ++iterations;
            ++pos;
            if (iterations > max_iter) finished = true;
        }
        return finished;
    }
};

// Eigen::Matrix4d::inverse(): cofactor expansion
inline bool mat4_inverse(const double m[16], double inv[16]) {
    double c[16];
    c[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    c[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    c[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    c[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    c[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    c[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    c[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    c[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    c[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    c[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    c[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    c[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    c[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    c[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    c[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    c[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const double det = m[0] * c[0] + m[1] * c[4] + m[2] * c[8] + m[3] * c[12];
    if (det == 0.0) return false;
    const double id = 1.0 / det;
    for (int i = 0; i < 16; ++i) inv[i] = c[i] * id;
    return true;
}

}  // namespace hm
}  // namespace cd
