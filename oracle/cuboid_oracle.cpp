/*
 * cuboid_oracle.cpp - CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain C++17 restatement of the reference's per-frame point-cloud path (crop -> voxel ->
 * RANSAC plane -> extract -> Euclidean clusters -> point-to-point ICP).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (libcuboid_hip.so) never links, loads or calls it.
 *
 * PARITY UNPINNED against real PCL: the reference's arithmetic for this path lives in the
 * un-vendored, un-pinned system dependency PCL (cuboid_detection/package.xml:57,78
 * `libpcl-all-dev`; inferred PCL 1.7.2 + Eigen 3.2 + FLANN 1.8 from the vendored
 * vision_opencv 1.12.8 = ROS Kinetic) and the reference holds no test, golden output or
 * recorded frame for it.  What follows restates PCL's published algorithms step by step,
 * following the reference's call sites:
 *   gps.cpp = cuboid_detection/src/ground_plane_segmentation.cpp
 *   icp.cpp = cuboid_detection/src/iterative_closest_point.cpp
 *   opd.cpp = object_detection/src/object_pose_detection.cpp
 *   sne.cpp = cuboid_detection/src/surface_normal_estimation.cpp (constrained planes, surface frame)
 *   cuboid_detection/src/bbox_filter.cpp (image-space gate)
 * It is pinned by what the tree does hold (tests/test_oracle_*.py): make_cuboid.py output
 * bytes, the *_ascii.pcd <-> *_ascii_tf.pcd <-> transforms.txt rigid-transform fixtures,
 * std::mt19937 known answers, and analytic known-answer cases.
 *
 * Canonical evaluation rules where PCL's own result is implementation-defined (documented
 * in DESIGN.md "Canonical arithmetic"):
 *   C1 no FMA contraction anywhere (-ffp-contract=off); float32 where PCL uses float32.
 *   C2 VoxelGrid: std::sort tie order -> stable (members summed in ascending input order).
 *   C3 4-wide dot (plane distance): (a*x + b*y) + (c*z + d).
 *   C4 long reductions (plane refit moments, ICP moments, MSE, fitness) are order-free:
 *      every float32 term is converted to fixed point (round-to-nearest-even at 2^-32, or
 *      2^-36 for squared distances) and summed in 64-bit integers.
 *   C5 nearest neighbour ties -> lowest template index; cluster labels = rank by
 *      (size descending, first member index ascending).
 *   C6 3-vector dots / norms of the constrained plane models and the surface frame: (x + y) + z;
 *      compute3DCentroid: sequential float32 sums in point order.
 */
#include <algorithm>
#include <array>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <random>
#include <unordered_map>
#include <vector>

#include "../include/cuboid_hip.h"

namespace {

constexpr int FIX_SHIFT = 32;     // C4: coordinates and products
constexpr int FIX_SHIFT_D2 = 36;  // C4: squared distances

// Arithmetic variants (orc_process_frame_arith).  The canonical rules C2/C4 pick ONE of the results real PCL can produce
// (its summation orders are implementation-defined); these switches evaluate another plausible one, so that the tests can
// MEASURE how far the canonical choice can sit from a PCL execution (real PCL is not available here):
//   ARITH_SEQUENTIAL   the long reductions as PCL/Eigen's scalar code would run them: plane-refit moments as float32
//                      running sums in inlier order (computeMeanAndCovarianceMatrix), ICP means and covariance as
//                      pcl::umeyama does (float32 sums, demeaned float32 products), MSE and fitness as double running
//                      sums of float32 squared distances (calculateMSE, getFitnessScore)
//   ARITH_REVERSE_TIES VoxelGrid's std::sort leaves equal keys in an unspecified order: sum every voxel in DESCENDING
//                      input order instead of ascending
//   ARITH_PROBE        canonical run that ALSO evaluates, from the same inputs at every step, what the sequential arithmetic
//                      would have produced, and records the largest single-step differences in t_probe: [0] Frobenius
//                      difference of the 4x4 ICP step transform, [1] max |difference| of the refitted plane coefficients,
//                      [2] relative difference of the correspondence MSE, [3] of the fitness score
enum { ARITH_SEQUENTIAL = 1, ARITH_REVERSE_TIES = 2, ARITH_PROBE = 4 };
static thread_local int t_arith = 0;
static thread_local double t_probe[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // [4] step transform incl. ill-conditioned steps, [5] their count, [6] steps
static thread_local float t_last_sv[3] = {0, 0, 0};   // singular values of the last umeyama covariance

inline float ldf(const uint8_t* p) {
    float f;
    std::memcpy(&f, p, 4);
    return f;
}
inline uint32_t ldu(const uint8_t* p) {
    uint32_t u;
    std::memcpy(&u, p, 4);
    return u;
}
inline int64_t fix(float v, int shift) {
    // float -> double is exact, scaling by a power of two is exact, llrint rounds to
    // nearest-even (default rounding mode).
    return (int64_t)std::llrint(std::ldexp((double)v, shift));
}
inline double unfix(int64_t s, int shift) { return std::ldexp((double)s, -shift); }

struct Cloud {  // strided xyz view
    const uint8_t* base;
    size_t stride;
    int n;
    float x(int i) const { return ldf(base + (size_t)i * stride); }
    float y(int i) const { return ldf(base + (size_t)i * stride + 4); }
    float z(int i) const { return ldf(base + (size_t)i * stride + 8); }
};

// ------------------------------------------------------------------------------------
// S0  pcl::PassThrough<PCLPointCloud2> (gps.cpp:53-65, opd.cpp:273-289, opd.cpp:331-336)
// PCL: a point is kept iff x,y,z are all finite, the filter field is finite and
// !(v > max) && !(v < min) with the float field promoted to double (limits are double).
// Order is preserved, the output is unorganized.
// ------------------------------------------------------------------------------------
void passthrough(const Cloud& c, const std::vector<int>* in_idx, int field /*0=x,1=y,2=z*/,
                 double lo, double hi, std::vector<int>& out) {
    out.clear();
    const int n = in_idx ? (int)in_idx->size() : c.n;
    for (int k = 0; k < n; ++k) {
        const int i = in_idx ? (*in_idx)[k] : k;
        const float p[3] = {c.x(i), c.y(i), c.z(i)};
        if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
        const double v = (double)p[field];
        if (v > hi || v < lo) continue;
        out.push_back(i);
    }
}

// ------------------------------------------------------------------------------------
// S1  pcl::VoxelGrid<PCLPointCloud2>::applyFilter (gps.cpp:69-73, opd.cpp:293-298)
// ------------------------------------------------------------------------------------
struct VoxelOut {
    std::vector<float> xyz;     // N_v * 3
    std::vector<uint32_t> rgb;  // N_v packed (0 if no rgb)
    int min_b[3] = {0, 0, 0}, div_b[3] = {0, 0, 0};
};

int voxel_grid(const Cloud& c, const std::vector<int>& idx, float leaf, int rgb_off,
               VoxelOut& out) {
    out.xyz.clear();
    out.rgb.clear();
    if (idx.empty()) return CD_OK;
    // getMinMax3D over the (already finite) input
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i : idx) {
        const float p[3] = {c.x(i), c.y(i), c.z(i)};
        for (int a = 0; a < 3; ++a) {
            mn[a] = std::min(mn[a], p[a]);
            mx[a] = std::max(mx[a], p[a]);
        }
    }
    const float inv = 1.0f / leaf;  // Eigen::Array4f::Ones() / leaf_size_.array()
    // "Leaf size is too small for the input dataset. Integer indices would overflow."
    const int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1;
    const int64_t dy = (int64_t)((mx[1] - mn[1]) * inv) + 1;
    const int64_t dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
    if (dx * dy * dz > (int64_t)INT32_MAX) return CD_ERR_LEAF_TOO_SMALL;
    int max_b[3];
    for (int a = 0; a < 3; ++a) {
        out.min_b[a] = (int)std::floor(mn[a] * inv);
        max_b[a] = (int)std::floor(mx[a] * inv);
        out.div_b[a] = max_b[a] - out.min_b[a] + 1;
    }
    // (PCL's check above uses (max-min)*inv; the per-axis divisions can still multiply past
    // int32 in the marginal case, where PCL would silently wrap.  Treated as the same error.)
    if ((int64_t)out.div_b[0] * out.div_b[1] * out.div_b[2] > (int64_t)INT32_MAX) return CD_ERR_LEAF_TOO_SMALL;
    const int mul1 = out.div_b[0], mul2 = out.div_b[0] * out.div_b[1];
    std::vector<std::pair<int, int>> kv(idx.size());  // (voxel idx, cloud point index)
    for (size_t k = 0; k < idx.size(); ++k) {
        const int i = idx[k];
        const int i0 = (int)(std::floor(c.x(i) * inv) - (float)out.min_b[0]);
        const int i1 = (int)(std::floor(c.y(i) * inv) - (float)out.min_b[1]);
        const int i2 = (int)(std::floor(c.z(i) * inv) - (float)out.min_b[2]);
        kv[k] = {i0 + i1 * mul1 + i2 * mul2, i};
    }
    // PCL: std::sort on idx (unstable).  C2: stable.
    std::stable_sort(kv.begin(), kv.end(),
                     [](const std::pair<int, int>& a, const std::pair<int, int>& b) {
                         return a.first < b.first;
                     });
    if (t_arith & ARITH_REVERSE_TIES) {   // the other extreme of what an unstable sort may leave
        for (size_t a0 = 0; a0 < kv.size();) {
            size_t a1 = a0 + 1;
            while (a1 < kv.size() && kv[a1].first == kv[a0].first) ++a1;
            std::reverse(kv.begin() + a0, kv.begin() + a1);
            a0 = a1;
        }
    }
    size_t first = 0;
    while (first < kv.size()) {
        size_t last = first + 1;
        while (last < kv.size() && kv[last].first == kv[first].first) ++last;
        float s[3] = {0.f, 0.f, 0.f}, col[3] = {0.f, 0.f, 0.f};
        for (size_t k = first; k < last; ++k) {
            const int i = kv[k].second;
            s[0] += c.x(i);
            s[1] += c.y(i);
            s[2] += c.z(i);
            if (rgb_off >= 0) {
                const uint32_t u = ldu(c.base + (size_t)i * c.stride + rgb_off);
                col[0] += (float)((u >> 16) & 0xff);
                col[1] += (float)((u >> 8) & 0xff);
                col[2] += (float)(u & 0xff);
            }
        }
        const float cnt = (float)(last - first);
        out.xyz.push_back(s[0] / cnt);
        out.xyz.push_back(s[1] / cnt);
        out.xyz.push_back(s[2] / cnt);
        uint32_t packed = 0;
        if (rgb_off >= 0) {
            const int r = (int)(col[0] / cnt), g = (int)(col[1] / cnt), b = (int)(col[2] / cnt);
            packed = ((uint32_t)r << 16) | ((uint32_t)g << 8) | (uint32_t)b;
        }
        out.rgb.push_back(packed);
        first = last;
    }
    return CD_OK;
}

// ------------------------------------------------------------------------------------
// S2  pcl::SACSegmentation<PointXYZ> PLANE / RANSAC / optimize (gps.cpp:76-93)
// ------------------------------------------------------------------------------------
struct PlaneSampler {  // SampleConsensusModel: rng seeded 12345 per model instance
    std::mt19937 rng{12345u};
    std::vector<int> shuffled;
    explicit PlaneSampler(int n) : shuffled(n) {
        for (int i = 0; i < n; ++i) shuffled[i] = i;
    }
    // boost::uniform_int<>(0, INT_MAX) over mt19937 == mt() >> 1
    int rnd() { return (int)(rng() >> 1); }
    void draw(int s[3]) {  // drawIndexSample
        const size_t n = shuffled.size();
        for (size_t i = 0; i < 3; ++i)
            std::swap(shuffled[i], shuffled[i + ((size_t)rnd() % (n - i))]);
        s[0] = shuffled[0];
        s[1] = shuffled[1];
        s[2] = shuffled[2];
    }
};

inline bool sample_good(const float* P, const int s[3]) {  // SampleConsensusModelPlane::isSampleGood
    const float* p0 = P + 3 * (size_t)s[0];
    const float* p1 = P + 3 * (size_t)s[1];
    const float* p2 = P + 3 * (size_t)s[2];
    const float q0 = (p1[0] - p0[0]) / (p2[0] - p0[0]);
    const float q1 = (p1[1] - p0[1]) / (p2[1] - p0[1]);
    const float q2 = (p1[2] - p0[2]) / (p2[2] - p0[2]);
    return (q0 != q1) || (q2 != q1);
}

inline bool plane_from_sample(const float* P, const int s[3], float m[4]) {
    // SampleConsensusModelPlane::computeModelCoefficients
    const float* p0 = P + 3 * (size_t)s[0];
    const float* p1 = P + 3 * (size_t)s[1];
    const float* p2 = P + 3 * (size_t)s[2];
    const float a[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
    const float b[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
    const float q0 = a[0] / b[0], q1 = a[1] / b[1], q2 = a[2] / b[2];
    if (q0 == q1 && q2 == q1) return false;  // collinear
    m[0] = a[1] * b[2] - a[2] * b[1];
    m[1] = a[2] * b[0] - a[0] * b[2];
    m[2] = a[0] * b[1] - a[1] * b[0];
    const float nrm = std::sqrt((m[0] * m[0] + m[1] * m[1]) + m[2] * m[2]);  // d = 0 here
    m[0] /= nrm;
    m[1] /= nrm;
    m[2] /= nrm;
    m[3] = -1.0f * ((m[0] * p0[0] + m[1] * p0[1]) + m[2] * p0[2]);
    return true;
}

inline float plane_dist(const float m[4], const float* p) {  // C3
    return std::fabs((m[0] * p[0] + m[1] * p[1]) + (m[2] * p[2] + m[3]));
}

int count_within(const float* P, int n, const float m[4], double thr) {
    int c = 0;
    for (int i = 0; i < n; ++i)
        if ((double)plane_dist(m, P + 3 * (size_t)i) < thr) ++c;
    return c;
}
void select_within(const float* P, int n, const float m[4], double thr, std::vector<int>& out) {
    out.clear();
    for (int i = 0; i < n; ++i)
        if ((double)plane_dist(m, P + 3 * (size_t)i) < thr) out.push_back(i);
}

// pcl::eigen33 smallest eigenpair (closed-form roots), float32
void compute_roots2(float b, float c, float roots[3]) {
    roots[0] = 0.f;
    float d = b * b - 4.0f * c;
    if (d < 0.0f) d = 0.0f;
    const float sd = std::sqrt(d);
    roots[2] = 0.5f * (b + sd);
    roots[1] = 0.5f * (b - sd);
}
void compute_roots(const float m[3][3], float roots[3]) {
    const float c0 = m[0][0] * m[1][1] * m[2][2] + 2.0f * m[0][1] * m[0][2] * m[1][2] -
                     m[0][0] * m[1][2] * m[1][2] - m[1][1] * m[0][2] * m[0][2] -
                     m[2][2] * m[0][1] * m[0][1];
    const float c1 = m[0][0] * m[1][1] - m[0][1] * m[0][1] + m[0][0] * m[2][2] -
                     m[0][2] * m[0][2] + m[1][1] * m[2][2] - m[1][2] * m[1][2];
    const float c2 = m[0][0] + m[1][1] + m[2][2];
    if (std::fabs(c0) < std::numeric_limits<float>::epsilon()) {
        compute_roots2(c2, c1, roots);
        return;
    }
    const float s_inv3 = 1.0f / 3.0f;
    const float s_sqrt3 = std::sqrt(3.0f);
    const float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.0f) a_over_3 = 0.0f;
    const float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
    float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (q > 0.0f) q = 0.0f;
    const float rho = std::sqrt(-a_over_3);
    const float theta = std::atan2(std::sqrt(-q), half_b) * s_inv3;
    const float cos_theta = std::cos(theta);
    const float sin_theta = std::sin(theta);
    roots[0] = c2_over_3 + 2.0f * rho * cos_theta;
    roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
    roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
    if (roots[0] >= roots[1]) std::swap(roots[0], roots[1]);
    if (roots[1] >= roots[2]) {
        std::swap(roots[1], roots[2]);
        if (roots[0] >= roots[1]) std::swap(roots[0], roots[1]);
    }
    if (roots[0] <= 0.0f) compute_roots2(c2, c1, roots);
}
inline void cross3(const float a[3], const float b[3], float o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
void eigen33_smallest(const float cov[3][3], float evec[3]) {
    float scale = 0.f;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) scale = std::max(scale, std::fabs(cov[i][j]));
    if (scale <= std::numeric_limits<float>::min()) scale = 1.0f;
    float s[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) s[i][j] = cov[i][j] / scale;
    float roots[3];
    compute_roots(s, roots);
    for (int i = 0; i < 3; ++i) s[i][i] -= roots[0];
    float v1[3], v2[3], v3[3];
    cross3(s[0], s[1], v1);
    cross3(s[0], s[2], v2);
    cross3(s[1], s[2], v3);
    const float l1 = (v1[0] * v1[0] + v1[1] * v1[1]) + v1[2] * v1[2];
    const float l2 = (v2[0] * v2[0] + v2[1] * v2[1]) + v2[2] * v2[2];
    const float l3 = (v3[0] * v3[0] + v3[1] * v3[1]) + v3[2] * v3[2];
    const float* v;
    float l;
    if (l1 >= l2 && l1 >= l3) {
        v = v1;
        l = l1;
    } else if (l2 >= l1 && l2 >= l3) {
        v = v2;
        l = l2;
    } else {
        v = v3;
        l = l3;
    }
    const float sl = std::sqrt(l);
    evec[0] = v[0] / sl;
    evec[1] = v[1] / sl;
    evec[2] = v[2] / sl;
}

// SampleConsensusModelPlane::optimizeModelCoefficients: computeMeanAndCovarianceMatrix
// (single-pass 9 moments) + eigen33.  C4: the 9 sums are fixed-point.
void plane_refit(const float* P, const std::vector<int>& inl, const float model[4], float out[4]) {
    if (inl.size() < 4) {
        std::memcpy(out, model, 16);
        return;
    }
    uint64_t S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i : inl) {
        const float x = P[3 * (size_t)i], y = P[3 * (size_t)i + 1], z = P[3 * (size_t)i + 2];
        const float t[9] = {x * x, x * y, x * z, y * y, y * z, z * z, x, y, z};
        for (int k = 0; k < 9; ++k) S[k] += (uint64_t)fix(t[k], FIX_SHIFT);
    }
    float accu[9];
    const double n = (double)inl.size();
    for (int k = 0; k < 9; ++k) accu[k] = (float)(unfix((int64_t)S[k], FIX_SHIFT) / n);
    if (t_arith & ARITH_PROBE) {
        float alt[4];
        const int keep = t_arith;
        t_arith = (keep & ~ARITH_PROBE) | ARITH_SEQUENTIAL;
        plane_refit(P, inl, model, alt);
        t_arith = keep;
        float can[4];
        t_arith = keep & ~(ARITH_PROBE | ARITH_SEQUENTIAL);
        plane_refit(P, inl, model, can);
        t_arith = keep;
        // eigenvector sign is not oriented: compare up to a common sign
        double d0 = 0, d1 = 0;
        for (int k = 0; k < 4; ++k) { d0 = std::max(d0, (double)std::fabs(can[k] - alt[k])); d1 = std::max(d1, (double)std::fabs(can[k] + alt[k])); }
        t_probe[1] = std::max(t_probe[1], std::min(d0, d1));
    }
    if (t_arith & ARITH_SEQUENTIAL) {   // pcl::computeMeanAndCovarianceMatrix as written: float32 running sums, one division
        float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int i : inl) {
            const float x = P[3 * (size_t)i], y = P[3 * (size_t)i + 1], z = P[3 * (size_t)i + 2];
            acc[0] += x * x; acc[1] += x * y; acc[2] += x * z; acc[3] += y * y; acc[4] += y * z; acc[5] += z * z;
            acc[6] += x; acc[7] += y; acc[8] += z;
        }
        for (int k = 0; k < 9; ++k) accu[k] = acc[k] / (float)inl.size();
    }
    float cov[3][3];
    cov[0][0] = accu[0] - accu[6] * accu[6];
    cov[0][1] = accu[1] - accu[6] * accu[7];
    cov[0][2] = accu[2] - accu[6] * accu[8];
    cov[1][1] = accu[3] - accu[7] * accu[7];
    cov[1][2] = accu[4] - accu[7] * accu[8];
    cov[2][2] = accu[5] - accu[8] * accu[8];
    cov[1][0] = cov[0][1];
    cov[2][0] = cov[0][2];
    cov[2][1] = cov[1][2];
    float ev[3];
    eigen33_smallest(cov, ev);
    out[0] = ev[0];
    out[1] = ev[1];
    out[2] = ev[2];
    out[3] = -1.0f * ((out[0] * accu[6] + out[1] * accu[7]) + out[2] * accu[8]);
}

// Axis constraint of SACMODEL_PERPENDICULAR_PLANE / SACMODEL_PARALLEL_PLANE
// (surface_normal_estimation.cpp:118-123: setModelType, setAxis(plane_normal), setEpsAngle(0.1)).
// PCL 1.7.2 isModelValid of the two models; float 4-vector products use the canonical (x + y) + z (w = 0).
struct PlaneConstraint {
    int type = CD_PLANE;
    float axis[3] = {0, 0, 0};
    double eps = 0.0;
};
bool plane_model_valid(const PlaneConstraint& pc, const float m[4]) {
    if (pc.type == CD_PLANE || !(pc.eps > 0.0)) return true;
    const float* ax = pc.axis;
    if (pc.type == CD_PLANE_PERPENDICULAR) {
        // getAngle3D(axis, coeff): float expression assigned to a double, clamped, acos
        const float dot = (ax[0] * m[0] + ax[1] * m[1]) + ax[2] * m[2];
        const float n1 = (ax[0] * ax[0] + ax[1] * ax[1]) + ax[2] * ax[2];
        const float n2 = (m[0] * m[0] + m[1] * m[1]) + m[2] * m[2];
        double rad = dot / std::sqrt(n1 * n2);
        if (rad < -1.0) rad = -1.0;
        else if (rad > 1.0) rad = 1.0;
        double angle_diff = std::fabs(std::acos(rad));
        angle_diff = std::min(angle_diff, M_PI - angle_diff);
        return !(angle_diff > pc.eps);
    }
    // parallel plane: coeff.normalize(); |axis . coeff| > sin_angle_ -> invalid
    const float nrm = std::sqrt((m[0] * m[0] + m[1] * m[1]) + m[2] * m[2]);
    const float c[3] = {m[0] / nrm, m[1] / nrm, m[2] / nrm};
    const float dot = (ax[0] * c[0] + ax[1] * c[1]) + ax[2] * c[2];
    return !((double)std::fabs(dot) > std::fabs(std::sin(pc.eps)));
}

struct PlaneOut {
    float coeff[4] = {0, 0, 0, 0};
    std::vector<int> inliers;
    int iterations = 0;
    int skipped = 0;
};

// trace hooks (tests): the first hypotheses in sampling order
struct PlaneTrace {
    int cap = 0;
    int n = 0;
    int32_t* triples = nullptr;  // cap*3
    float* models = nullptr;     // cap*4
    int32_t* counts = nullptr;   // cap   (-1 = model invalid / skipped)
};

int segment_plane(const float* P, int n, double thr, int max_iter, double prob, bool optimize,
                  PlaneOut& out, PlaneTrace* tr = nullptr, const PlaneConstraint& pc = PlaneConstraint()) {
    out.inliers.clear();
    out.iterations = 0;
    out.skipped = 0;
    if (n < 3) return CD_ERR_NO_MODEL;  // "Can not select 3 unique points out of n"
    PlaneSampler smp(n);
    int iterations = 0, best = -INT_MAX, skipped = 0;
    double k = 1.0;
    const double log_probability = std::log(1.0 - prob);
    const double one_over_indices = 1.0 / (double)n;
    const int max_skip = max_iter * 10;
    float best_model[4] = {0, 0, 0, 0};
    bool have = false;
    while ((double)iterations < k && skipped < max_skip) {
        int s[3];
        bool good = false;
        for (int it = 0; it < 1000; ++it) {  // max_sample_checks_
            smp.draw(s);
            if (sample_good(P, s)) {
                good = true;
                break;
            }
        }
        if (!good) break;  // "No samples could be selected!"
        float m[4];
        const bool ok = plane_from_sample(P, s, m);
        if (tr && tr->n < tr->cap) {
            std::memcpy(tr->triples + 3 * tr->n, s, 12);
            if (ok) std::memcpy(tr->models + 4 * tr->n, m, 16);
            tr->counts[tr->n] = ok ? count_within(P, n, m, thr) : -1;
            ++tr->n;
        }
        if (!ok) {
            ++skipped;
            continue;
        }
        // countWithinDistance of the constrained models starts with isModelValid
        const int cnt = plane_model_valid(pc, m) ? count_within(P, n, m, thr) : 0;
        if (cnt > best) {
            best = cnt;
            std::memcpy(best_model, m, 16);
            have = true;
            const double w = (double)best * one_over_indices;
            double p_no_outliers = 1.0 - std::pow(w, 3.0);
            p_no_outliers = std::max(std::numeric_limits<double>::epsilon(), p_no_outliers);
            p_no_outliers = std::min(1.0 - std::numeric_limits<double>::epsilon(), p_no_outliers);
            k = log_probability / std::log(p_no_outliers);
        }
        ++iterations;
        if (iterations > max_iter) break;
    }
    out.iterations = iterations;
    out.skipped = skipped;
    if (!have) return CD_ERR_NO_MODEL;
    // selectWithinDistance of the constrained models also starts with isModelValid (-> no inliers)
    if (plane_model_valid(pc, best_model)) select_within(P, n, best_model, thr, out.inliers);
    std::memcpy(out.coeff, best_model, 16);
    if (optimize) {
        float refined[4];
        plane_refit(P, out.inliers, best_model, refined);
        std::memcpy(out.coeff, refined, 16);
        out.inliers.clear();
        if (plane_model_valid(pc, refined)) select_within(P, n, refined, thr, out.inliers);
    }
    return CD_OK;
}

// ------------------------------------------------------------------------------------
// bbox_filter.cpp:30-51 within_bbox(): the projection is accumulated in double (proj_matrix is
// vector<double>, x/y/z float), stored to float, normalised by a float division, and compared
// strictly against the int rectangle (converted to float).
// ------------------------------------------------------------------------------------
inline bool within_bbox(const double P[12], const int rect[4], float x, float y, float z) {
    float u = (float)((((P[0] * x) + (P[1] * y)) + (P[2] * z)) + P[3]);
    float v = (float)((((P[4] * x) + (P[5] * y)) + (P[6] * z)) + P[7]);
    const float w = (float)((((P[8] * x) + (P[9] * y)) + (P[10] * z)) + P[11]);
    u /= w;
    v /= w;
    return ((float)rect[0] < u && u < (float)rect[2]) && ((float)rect[1] < v && v < (float)rect[3]);
}

// ------------------------------------------------------------------------------------
// S5  pcl::EuclideanClusterExtraction (opd.cpp:352-362)
// neighbour predicate: (dx*dx + dy*dy) + dz*dz < (float)(tol*tol), strict.
// ------------------------------------------------------------------------------------
inline float dist2(const float* a, const float* b) {
    const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return (dx * dx + dy * dy) + dz * dz;
}

struct Dsu {
    std::vector<int> p;
    explicit Dsu(int n) : p(n) {
        for (int i = 0; i < n; ++i) p[i] = i;
    }
    int find(int a) {
        while (p[a] != a) {
            p[a] = p[p[a]];
            a = p[a];
        }
        return a;
    }
    void unite(int a, int b) {
        a = find(a);
        b = find(b);
        if (a == b) return;
        if (a < b)
            p[b] = a;
        else
            p[a] = b;  // root = smallest member index
    }
};

struct CellHash {
    size_t operator()(const std::array<int, 3>& k) const {
        return (size_t)k[0] * 73856093u ^ (size_t)k[1] * 19349663u ^ (size_t)k[2] * 83492791u;
    }
};

void cluster(const float* P, int n, double tol, int min_sz, int max_sz, int mode,
             std::vector<int>& labels, std::vector<int>& sizes) {
    labels.assign(n, -1);
    sizes.clear();
    if (n == 0) return;
    const float r2 = (float)(tol * tol);
    Dsu d(n);
    if (mode == 0) {  // definition: all pairs
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < i; ++j)
                if (dist2(P + 3 * (size_t)i, P + 3 * (size_t)j) < r2) d.unite(i, j);
    } else {  // uniform grid, cell edge slightly larger than tol (same partition, tested)
        const double cell = tol * (1.0 + 1.0 / 1024.0);
        std::unordered_map<std::array<int, 3>, std::vector<int>, CellHash> grid;
        grid.reserve((size_t)n);
        auto key = [&](int i) {
            return std::array<int, 3>{(int)std::floor(P[3 * (size_t)i] / cell),
                                      (int)std::floor(P[3 * (size_t)i + 1] / cell),
                                      (int)std::floor(P[3 * (size_t)i + 2] / cell)};
        };
        for (int i = 0; i < n; ++i) grid[key(i)].push_back(i);
        for (int i = 0; i < n; ++i) {
            const auto k = key(i);
            for (int a = -1; a <= 1; ++a)
                for (int b = -1; b <= 1; ++b)
                    for (int c = -1; c <= 1; ++c) {
                        auto it = grid.find({k[0] + a, k[1] + b, k[2] + c});
                        if (it == grid.end()) continue;
                        for (int j : it->second)
                            if (j < i && dist2(P + 3 * (size_t)i, P + 3 * (size_t)j) < r2)
                                d.unite(i, j);
                    }
        }
    }
    std::vector<int> root(n), csize(n, 0);
    for (int i = 0; i < n; ++i) {
        root[i] = d.find(i);
        ++csize[root[i]];
    }
    std::vector<int> kept;  // roots of kept components
    for (int i = 0; i < n; ++i)
        if (root[i] == i && csize[i] >= min_sz && csize[i] <= max_sz) kept.push_back(i);
    // C5: size descending, first member (= root) ascending
    std::stable_sort(kept.begin(), kept.end(), [&](int a, int b) { return csize[a] > csize[b]; });
    std::vector<int> rank(n, -1);
    for (size_t k = 0; k < kept.size(); ++k) {
        rank[kept[k]] = (int)k;
        sizes.push_back(csize[kept[k]]);
    }
    for (int i = 0; i < n; ++i) labels[i] = rank[root[i]];
}

// ------------------------------------------------------------------------------------
// S6  pcl::IterativeClosestPoint<PointXYZ,PointXYZ> (icp.cpp:170-182, opd.cpp:220-235)
// ------------------------------------------------------------------------------------
struct KdTree {  // exact NN, ties -> lowest index; stands in for FLANN's KDTreeSingleIndex
    struct Node {
        int lo, hi;       // leaf: range in perm
        int left, right;  // children or -1
        int dim;
        float split;
    };
    const float* T = nullptr;
    std::vector<int> perm;
    std::vector<Node> nodes;
    void build(const float* tgt, int m) {
        T = tgt;
        perm.resize(m);
        for (int i = 0; i < m; ++i) perm[i] = i;
        nodes.clear();
        nodes.reserve(m / 4 + 8);
        if (m > 0) rec(0, m);
    }
    int rec(int lo, int hi) {
        const int id = (int)nodes.size();
        nodes.push_back({lo, hi, -1, -1, 0, 0.f});
        if (hi - lo <= 8) return id;
        float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (int k = lo; k < hi; ++k)
            for (int a = 0; a < 3; ++a) {
                mn[a] = std::min(mn[a], T[3 * (size_t)perm[k] + a]);
                mx[a] = std::max(mx[a], T[3 * (size_t)perm[k] + a]);
            }
        int dim = 0;
        if (mx[1] - mn[1] > mx[dim] - mn[dim]) dim = 1;
        if (mx[2] - mn[2] > mx[dim] - mn[dim]) dim = 2;
        const int mid = (lo + hi) / 2;
        std::nth_element(perm.begin() + lo, perm.begin() + mid, perm.begin() + hi,
                         [&](int a, int b) { return T[3 * (size_t)a + dim] < T[3 * (size_t)b + dim]; });
        const float split = T[3 * (size_t)perm[mid] + dim];
        const int l = rec(lo, mid);
        const int r = rec(mid, hi);
        nodes[id].left = l;
        nodes[id].right = r;
        nodes[id].dim = dim;
        nodes[id].split = split;
        return id;
    }
    void query(int id, const float* q, float& best, int& bi) const {
        const Node& nd = nodes[id];
        if (nd.left < 0) {
            for (int k = nd.lo; k < nd.hi; ++k) {
                const int j = perm[k];
                const float d = dist2(q, T + 3 * (size_t)j);
                if (d < best || (d == best && j < bi)) {
                    best = d;
                    bi = j;
                }
            }
            return;
        }
        // left holds coords <= split, right holds coords >= split
        const float diff = q[nd.dim] - nd.split;
        const int near = diff < 0.f ? nd.left : nd.right;
        const int far = diff < 0.f ? nd.right : nd.left;
        query(near, q, best, bi);
        // every far point p has |p[dim]-q[dim]| >= |diff| and float rounding is monotone,
        // so its canonical float distance is >= diff*diff: prune only when strictly worse.
        if (!(diff * diff > best)) query(far, q, best, bi);
    }
};

inline void nn_brute(const float* T, int m, const float* q, float& best, int& bi) {
    best = FLT_MAX;
    bi = -1;
    for (int j = 0; j < m; ++j) {
        const float d = dist2(q, T + 3 * (size_t)j);
        if (d < best) {  // strict: the first (lowest) index wins ties
            best = d;
            bi = j;
        }
    }
}

// Eigen 3.2 JacobiSVD<Matrix3f>(ComputeFullU|ComputeFullV), two-sided Jacobi, float32
struct Rot {
    float c, s;
};
inline Rot rot_mul(const Rot& a, const Rot& b) { return {a.c * b.c - a.s * b.s, a.c * b.s + a.s * b.c}; }
inline Rot rot_T(const Rot& a) { return {a.c, -a.s}; }
inline void apply_left(float M[3][3], int p, int q, const Rot& j) {
    for (int i = 0; i < 3; ++i) {
        const float x = M[p][i], y = M[q][i];
        M[p][i] = j.c * x + j.s * y;
        M[q][i] = -j.s * x + j.c * y;
    }
}
inline void apply_right(float M[3][3], int p, int q, const Rot& j) {
    for (int i = 0; i < 3; ++i) {
        const float x = M[i][p], y = M[i][q];
        M[i][p] = j.c * x - j.s * y;
        M[i][q] = j.s * x + j.c * y;
    }
}
inline Rot make_jacobi(float x, float y, float z) {
    if (y == 0.f) return {1.f, 0.f};
    const float tau = (x - z) / (2.0f * std::fabs(y));
    const float w = std::sqrt(tau * tau + 1.0f);
    const float t = tau > 0.f ? 1.0f / (tau + w) : 1.0f / (tau - w);
    const float sign_t = t > 0.f ? 1.0f : -1.0f;
    const float n = 1.0f / std::sqrt(t * t + 1.0f);
    Rot r;
    r.s = -sign_t * (y / std::fabs(y)) * std::fabs(t) * n;
    r.c = n;
    return r;
}
void jacobi_svd3(const float A[3][3], float U[3][3], float S[3], float V[3][3]) {
    const float precision = 2.0f * std::numeric_limits<float>::epsilon();
    const float consider_zero = 2.0f * std::numeric_limits<float>::denorm_min();
    float scale = 0.f;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) scale = std::max(scale, std::fabs(A[i][j]));
    if (scale == 0.f) scale = 1.f;
    float W[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            W[i][j] = A[i][j] / scale;
            U[i][j] = V[i][j] = (i == j) ? 1.f : 0.f;
        }
    bool finished = false;
    for (int sweep = 0; sweep < 64 && !finished; ++sweep) {  // Eigen loops unbounded
        finished = true;
        for (int p = 1; p < 3; ++p)
            for (int q = 0; q < p; ++q) {
                const float thr = std::max(consider_zero, precision * std::max(std::fabs(W[p][p]), std::fabs(W[q][q])));
                if (std::fabs(W[p][q]) > thr || std::fabs(W[q][p]) > thr) {
                    finished = false;
                    // real_2x2_jacobi_svd
                    float m00 = W[p][p], m01 = W[p][q], m10 = W[q][p], m11 = W[q][q];
                    Rot rot1;
                    const float t = m00 + m11, d = m10 - m01;
                    if (t == 0.f) {
                        rot1.c = 0.f;
                        rot1.s = d > 0.f ? 1.f : -1.f;
                    } else {
                        const float u = d / t;
                        rot1.c = 1.0f / std::sqrt(1.0f + u * u);
                        rot1.s = rot1.c * u;
                    }
                    // m.applyOnTheLeft(0,1,rot1)
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
        const float a = std::fabs(W[i][i]);
        S[i] = a;
        if (a != 0.f) {
            const float f = W[i][i] / a;
            for (int r = 0; r < 3; ++r) U[r][i] *= f;
        }
    }
    for (int i = 0; i < 3; ++i) {
        int pos = i;
        float mxv = S[i];
        for (int k = i + 1; k < 3; ++k)
            if (S[k] > mxv) {
                mxv = S[k];
                pos = k;
            }
        if (mxv == 0.f) break;
        if (pos != i) {
            std::swap(S[i], S[pos]);
            for (int r = 0; r < 3; ++r) {
                std::swap(U[r][i], U[r][pos]);
                std::swap(V[r][i], V[r][pos]);
            }
        }
    }
    for (int i = 0; i < 3; ++i) S[i] *= scale;
}
inline float det3(const float m[3][3]) {
    return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) -
           m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
           m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
}

// pcl::umeyama(src, dst, with_scaling=false) from the means and the covariance: SVD, reflection handling, R, t
void umeyama_from_sigma(const float mpf[3], const float mqf[3], const float sigma[3][3], float T[16]) {
    float U[3][3], S[3], V[3][3];
    jacobi_svd3(sigma, U, S, V);
    t_last_sv[0] = S[0]; t_last_sv[1] = S[1]; t_last_sv[2] = S[2];
    float sd[3] = {1.f, 1.f, 1.f};
    if (det3(sigma) < 0.f) sd[2] = -1.f;
    int rank = 0;
    for (int i = 0; i < 3; ++i)
        if (!(std::fabs(S[i]) <= std::fabs(S[0]) * 1e-5f)) ++rank;  // !isMuchSmallerThan
    float R[3][3];
    auto usvt = [&](const float s[3]) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                R[i][j] = ((U[i][0] * s[0]) * V[j][0] + (U[i][1] * s[1]) * V[j][1]) + (U[i][2] * s[2]) * V[j][2];
    };
    if (rank == 2) {
        if (det3(U) * det3(V) > 0.f) {
            const float one[3] = {1.f, 1.f, 1.f};
            usvt(one);
        } else {
            const float s2[3] = {1.f, 1.f, -1.f};
            usvt(s2);
        }
    } else {
        usvt(sd);
    }
    for (int i = 0; i < 16; ++i) T[i] = 0.f;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[4 * i + j] = R[i][j];
        T[4 * i + 3] = mqf[i] - ((R[i][0] * mpf[0] + R[i][1] * mpf[1]) + R[i][2] * mpf[2]);
    }
    T[15] = 1.f;
}

// ... fed with the fixed-point moments (C4)
void umeyama_from_moments(const uint64_t Sp[3], const uint64_t Sq[3], const uint64_t Sqp[9], int n,
                          float T[16]) {
    double mp[3], mq[3];
    for (int a = 0; a < 3; ++a) {
        mp[a] = unfix((int64_t)Sp[a], FIX_SHIFT) / (double)n;
        mq[a] = unfix((int64_t)Sq[a], FIX_SHIFT) / (double)n;
    }
    float sigma[3][3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
            sigma[a][b] = (float)(unfix((int64_t)Sqp[3 * a + b], FIX_SHIFT) / (double)n - mq[a] * mp[b]);
    const float mpf[3] = {(float)mp[0], (float)mp[1], (float)mp[2]};
    const float mqf[3] = {(float)mq[0], (float)mq[1], (float)mq[2]};
    umeyama_from_sigma(mpf, mqf, sigma, T);
}

// ... as pcl::umeyama's own code runs it in scalar float32 (ARITH_SEQUENTIAL): src_mean = src.rowwise().sum() * (1/n),
// the clouds demeaned, sigma = (1/n) * dst_demean * src_demean^T, every sum a float32 running sum in point order
void umeyama_sequential(const float* P, const float* Q, int n, float T[16]) {
    const float one_over_n = 1.0f / (float)n;
    float sp[3] = {0, 0, 0}, sq[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) { sp[a] += P[3 * (size_t)i + a]; sq[a] += Q[3 * (size_t)i + a]; }
    float mp[3], mq[3];
    for (int a = 0; a < 3; ++a) { mp[a] = sp[a] * one_over_n; mq[a] = sq[a] * one_over_n; }
    float acc[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int i = 0; i < n; ++i) {
        float dp[3], dq[3];
        for (int a = 0; a < 3; ++a) { dp[a] = P[3 * (size_t)i + a] - mp[a]; dq[a] = Q[3 * (size_t)i + a] - mq[a]; }
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) acc[a][b] += dq[a] * dp[b];
    }
    float sigma[3][3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) sigma[a][b] = one_over_n * acc[a][b];
    umeyama_from_sigma(mp, mq, sigma, T);
}

inline void xform(const float T[16], const float* p, float* o) {
    const float x = p[0], y = p[1], z = p[2];
    o[0] = ((T[0] * x + T[1] * y) + T[2] * z) + T[3];
    o[1] = ((T[4] * x + T[5] * y) + T[6] * z) + T[7];
    o[2] = ((T[8] * x + T[9] * y) + T[10] * z) + T[11];
}
inline void mat4_mul(const float A[16], const float B[16], float C[16]) {
    float t[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            t[4 * i + j] = ((A[4 * i] * B[j] + A[4 * i + 1] * B[4 + j]) + A[4 * i + 2] * B[8 + j]) + A[4 * i + 3] * B[12 + j];
    std::memcpy(C, t, 64);
}

// general 4x4 inverse by cofactors, double (Eigen::Matrix4d::inverse())
bool mat4_inverse(const double m[16], double inv[16]) {
    double t[16];
    t[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    t[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    t[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    t[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    t[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    t[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    t[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    t[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    t[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    t[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    t[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    t[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    t[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    t[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    t[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    t[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const double det = m[0] * t[0] + m[1] * t[4] + m[2] * t[8] + m[3] * t[12];
    if (det == 0.0) return false;
    const double id = 1.0 / det;
    for (int i = 0; i < 16; ++i) inv[i] = t[i] * id;
    return true;
}

struct IcpOut {
    float T[16];
    int iterations = 0;
    int converged = 0;
    double fitness = 0.0;
    std::vector<float> aligned;
};

// guess (may be NULL = identity, the reference's live path): pcl::Registration::align(output, guess) followed by
// IterativeClosestPoint::computeTransformation - final_transformation_ = guess; if (guess != Identity)
// transformPointCloud(*input_, *input_transformed, guess) (float32, ((a x + b y) + c z) + d per row); the iterations,
// the convergence tests and getFitnessScore (final_transformation_ * original input) are unchanged.  The reference's
// authors prepared this start (icp.cpp:130-134 stores the sne pose, :165-167 would move the template by it) and left
// it commented out: opt-in here as well (cd_params.icp_use_guess).
int icp_align(const float* src, int n, const float* tgt, int m, int nn_mode, int max_iter,
              double trans_eps, double rel_mse, IcpOut& out, const float* guess = nullptr) {
    for (int i = 0; i < 16; ++i) out.T[i] = guess ? guess[i] : ((i % 5 == 0) ? 1.f : 0.f);
    out.iterations = 0;
    out.converged = 0;
    out.fitness = DBL_MAX;
    out.aligned.assign(src, src + 3 * (size_t)n);
    if (m <= 0) return CD_ERR_NO_TEMPLATE;
    if (n < 3) return CD_ERR_FEW_CORRESPONDENCES;  // min_number_correspondences_ = 3
    if (guess)
        for (int i = 0; i < n; ++i) {
            float o[3];
            xform(guess, src + 3 * (size_t)i, o);
            std::memcpy(out.aligned.data() + 3 * (size_t)i, o, 12);
        }
    KdTree kd;
    if (nn_mode == 1) kd.build(tgt, m);
    auto nn = [&](const float* q, float& d, int& j) {
        if (nn_mode == 1) {
            d = FLT_MAX;
            j = INT_MAX;
            kd.query(0, q, d, j);
        } else {
            nn_brute(tgt, m, q, d, j);
        }
    };
    std::vector<float>& X = out.aligned;
    std::vector<float> Qseq;
    double prev_mse = std::numeric_limits<double>::max();
    const double rot_thr = 1.0 - trans_eps;  // setRotationThreshold(1.0 - transformation_epsilon_)
    const double abs_mse_thr = 1e-12;        // DefaultConvergenceCriteria default
    for (;;) {
        uint64_t Sp[3] = {0, 0, 0}, Sq[3] = {0, 0, 0}, Sqp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Sd = 0;
        double mse_seq = 0.0;               // ARITH_SEQUENTIAL: calculateMSE's double running sum
        if (t_arith & (ARITH_SEQUENTIAL | ARITH_PROBE)) Qseq.resize(3 * (size_t)n);
        for (int i = 0; i < n; ++i) {
            const float* p = X.data() + 3 * (size_t)i;
            float d;
            int j;
            nn(p, d, j);
            const float* q = tgt + 3 * (size_t)j;
            if (t_arith & (ARITH_SEQUENTIAL | ARITH_PROBE)) { mse_seq += (double)d; std::memcpy(&Qseq[3 * (size_t)i], q, 12); }
            for (int a = 0; a < 3; ++a) {
                Sp[a] += (uint64_t)fix(p[a], FIX_SHIFT);
                Sq[a] += (uint64_t)fix(q[a], FIX_SHIFT);
                for (int b = 0; b < 3; ++b) Sqp[3 * a + b] += (uint64_t)fix(q[a] * p[b], FIX_SHIFT);
            }
            Sd += (uint64_t)fix(d, FIX_SHIFT_D2);
        }
        float T[16];
        if (t_arith & ARITH_SEQUENTIAL) umeyama_sequential(X.data(), Qseq.data(), n, T);
        else umeyama_from_moments(Sp, Sq, Sqp, n, T);
        if (t_arith & ARITH_PROBE) {   // same correspondences, the other arithmetic
            // a covariance whose second singular value is (numerically) nothing leaves the rotation about the first axis to
            // the rounding noise - in real PCL as much as here (iteration 0 from the identity guess, half a metre away, maps
            // the whole cluster onto one corner of the template): such steps are counted, not compared
            const bool well = t_last_sv[1] > 1.0e-3f * t_last_sv[0] && t_last_sv[0] > 0.f;
            float Ts[16];
            umeyama_sequential(X.data(), Qseq.data(), n, Ts);
            double e = 0;
            for (int i = 0; i < 16; ++i) e += ((double)T[i] - Ts[i]) * ((double)T[i] - Ts[i]);
            t_probe[4] = std::max(t_probe[4], std::sqrt(e));
            t_probe[6] += 1.0;
            if (well) t_probe[0] = std::max(t_probe[0], std::sqrt(e)); else t_probe[5] += 1.0;
            const double m_can = unfix((int64_t)Sd, FIX_SHIFT_D2) / (double)n, m_seq = mse_seq / (double)n;
            if (m_can > 0) t_probe[2] = std::max(t_probe[2], std::fabs(m_can - m_seq) / m_can);
        }
        for (int i = 0; i < n; ++i) {
            float o[3];
            xform(T, X.data() + 3 * (size_t)i, o);
            X[3 * (size_t)i] = o[0];
            X[3 * (size_t)i + 1] = o[1];
            X[3 * (size_t)i + 2] = o[2];
        }
        mat4_mul(T, out.T, out.T);  // final = T * final
        ++out.iterations;
        // DefaultConvergenceCriteria::hasConverged
        if (out.iterations >= max_iter) {
            out.converged = 1;
            break;
        }
        const double cos_angle = 0.5 * (double)(((T[0] + T[5]) + T[10]) - 1.0f);
        const double translation_sqr = (double)((T[3] * T[3] + T[7] * T[7]) + T[11] * T[11]);
        if (cos_angle >= rot_thr && translation_sqr <= trans_eps) {
            out.converged = 1;
            break;
        }
        const double mse = (t_arith & ARITH_SEQUENTIAL) ? mse_seq / (double)n : unfix((int64_t)Sd, FIX_SHIFT_D2) / (double)n;
        if (std::fabs(mse - prev_mse) < abs_mse_thr) {
            out.converged = 1;
            break;
        }
        if (std::fabs(mse - prev_mse) / prev_mse < rel_mse) {
            out.converged = 1;
            break;
        }
        prev_mse = mse;
    }
    // getFitnessScore(): transform the ORIGINAL source by final_transformation_
    uint64_t Sf = 0;
    double fit_seq = 0.0;
    for (int i = 0; i < n; ++i) {
        float o[3], d;
        int j;
        xform(out.T, src + 3 * (size_t)i, o);
        nn(o, d, j);
        Sf += (uint64_t)fix(d, FIX_SHIFT_D2);
        fit_seq += (double)d;
    }
    out.fitness = (t_arith & ARITH_SEQUENTIAL) ? fit_seq / (double)n : unfix((int64_t)Sf, FIX_SHIFT_D2) / (double)n;
    if (t_arith & ARITH_PROBE) {
        const double f_can = unfix((int64_t)Sf, FIX_SHIFT_D2) / (double)n, f_seq = fit_seq / (double)n;
        if (f_can > 0) t_probe[3] = std::max(t_probe[3], std::fabs(f_can - f_seq) / f_can);
    }
    return CD_OK;
}

void fill_cluster_result(const IcpOut& io, int size, double accept, cd_cluster_result* r) {
    r->size = size;
    r->template_slot = 0;
    r->reserved = 0;
    r->iterations = io.iterations;
    r->converged = io.converged;
    r->fitness = io.fitness;
    r->accepted = (io.converged && io.fitness < accept) ? 1 : 0;
    std::memcpy(r->T, io.T, 64);
    double Td[16];
    for (int i = 0; i < 16; ++i) Td[i] = (double)io.T[i];
    if (!mat4_inverse(Td, r->pose))
        for (int i = 0; i < 16; ++i) r->pose[i] = std::numeric_limits<double>::quiet_NaN();
}

}  // namespace

// ======================================================================================
// C interface (ctypes).  Mirrors include/cuboid_hip.h so the parity tests compare
// like with like.  orc_* never touches a GPU.
// ======================================================================================
extern "C" {

int orc_mt19937_stream(uint32_t seed, int n, uint32_t* out) {
    std::mt19937 g(seed);
    for (int i = 0; i < n; ++i) out[i] = (uint32_t)g();
    return CD_OK;
}

int orc_passthrough(const void* pts, size_t stride, int n, int field, double lo, double hi,
                    int32_t* out_idx, int* out_n) {
    Cloud c{(const uint8_t*)pts, stride, n};
    std::vector<int> o;
    passthrough(c, nullptr, field, lo, hi, o);
    std::memcpy(out_idx, o.data(), o.size() * 4);
    *out_n = (int)o.size();
    return CD_OK;
}

int orc_crop_voxel(const void* pts, size_t stride, int n, const cd_params* prm, float* out_xyz,
                   uint32_t* out_rgb, int capacity, int* out_n_cropped, int* out_n_voxels,
                   int32_t* grid_info /* min_b[3], div_b[3] or NULL */) {
    Cloud c{(const uint8_t*)pts, stride, n};
    std::vector<int> a, b;
    passthrough(c, nullptr, 2, prm->crop_z_min, prm->crop_z_max, a);
    passthrough(c, &a, 0, prm->crop_x_min, prm->crop_x_max, b);
    VoxelOut vo;
    const int st = voxel_grid(c, b, prm->leaf_size, prm->rgb_offset, vo);
    *out_n_cropped = (int)b.size();
    *out_n_voxels = 0;
    if (st != CD_OK) return st;
    const int nv = (int)vo.rgb.size();
    if (nv > capacity) return CD_ERR_CAPACITY;
    std::memcpy(out_xyz, vo.xyz.data(), (size_t)nv * 12);
    if (out_rgb) std::memcpy(out_rgb, vo.rgb.data(), (size_t)nv * 4);
    *out_n_voxels = nv;
    if (grid_info) {
        for (int k = 0; k < 3; ++k) {
            grid_info[k] = vo.min_b[k];
            grid_info[3 + k] = vo.div_b[k];
        }
    }
    return CD_OK;
}

static void gather_xyz(const void* pts, size_t stride, int n, std::vector<float>& P) {
    Cloud c{(const uint8_t*)pts, stride, n};
    P.resize(3 * (size_t)n);
    for (int i = 0; i < n; ++i) {
        P[3 * (size_t)i] = c.x(i);
        P[3 * (size_t)i + 1] = c.y(i);
        P[3 * (size_t)i + 2] = c.z(i);
    }
}

int orc_segment_plane(const void* xyz, size_t stride, int n, const cd_params* prm, float coeff[4],
                      int32_t* inliers, int capacity, int* out_n, int* out_iterations) {
    std::vector<float> P;
    gather_xyz(xyz, stride, n, P);
    PlaneOut po;
    PlaneConstraint pc;
    pc.type = prm->plane_model;
    for (int a = 0; a < 3; ++a) pc.axis[a] = prm->plane_axis[a];
    pc.eps = prm->plane_eps_angle;
    const int st = segment_plane(P.data(), n, prm->plane_distance_threshold, prm->plane_max_iterations,
                                 prm->plane_probability, prm->plane_optimize != 0, po, nullptr, pc);
    *out_n = 0;
    if (out_iterations) *out_iterations = po.iterations;
    if (st != CD_OK) return st;
    if ((int)po.inliers.size() > capacity) return CD_ERR_CAPACITY;
    std::memcpy(coeff, po.coeff, 16);
    std::memcpy(inliers, po.inliers.data(), po.inliers.size() * 4);
    *out_n = (int)po.inliers.size();
    return CD_OK;
}

// hypotheses in sampling order (tests of the sampler / model / count stages)
int orc_ransac_trace(const void* xyz, size_t stride, int n, const cd_params* prm, int cap,
                     int32_t* triples, float* models, int32_t* counts, int* out_n) {
    std::vector<float> P;
    gather_xyz(xyz, stride, n, P);
    PlaneOut po;
    PlaneTrace tr;
    tr.cap = cap;
    tr.triples = triples;
    tr.models = models;
    tr.counts = counts;
    // force the loop to consume `cap` hypotheses regardless of the adaptive stop
    PlaneSampler smp(n);
    int got = 0;
    while (got < cap) {
        int s[3];
        bool good = false;
        for (int it = 0; it < 1000; ++it) {
            smp.draw(s);
            if (sample_good(P.data(), s)) {
                good = true;
                break;
            }
        }
        if (!good) break;
        float m[4] = {0, 0, 0, 0};
        const bool ok = plane_from_sample(P.data(), s, m);
        std::memcpy(triples + 3 * got, s, 12);
        std::memcpy(models + 4 * got, m, 16);
        counts[got] = ok ? count_within(P.data(), n, m, prm->plane_distance_threshold) : -1;
        ++got;
    }
    *out_n = got;
    return CD_OK;
}

int orc_plane_refit(const void* xyz, size_t stride, int n, const int32_t* inliers, int n_inl,
                    const float model[4], float out[4]) {
    std::vector<float> P;
    gather_xyz(xyz, stride, n, P);
    std::vector<int> inl(inliers, inliers + n_inl);
    plane_refit(P.data(), inl, model, out);
    return CD_OK;
}

int orc_cluster(const void* xyz, size_t stride, int n, const cd_params* prm, int mode,
                int32_t* labels, int32_t* sizes, int sizes_capacity, int* out_k) {
    std::vector<float> P;
    gather_xyz(xyz, stride, n, P);
    std::vector<int> lab, sz;
    cluster(P.data(), n, prm->cluster_tolerance, prm->cluster_min_size, prm->cluster_max_size, mode, lab, sz);
    std::memcpy(labels, lab.data(), (size_t)n * 4);
    for (int k = 0; k < (int)sz.size() && k < sizes_capacity; ++k) sizes[k] = sz[k];
    *out_k = (int)sz.size();
    return CD_OK;
}

// surface_normal_estimation.cpp:167-234 (callback) with getNormal (:105-165)
int orc_surface_frame(const void* xyz, size_t stride, int n, const float table_normal[3], int invert, const cd_params* prm,
                      cd_surface_frame_result* out) {
    std::memset(out, 0, sizeof(*out));
    std::vector<float> cloud;
    gather_xyz(xyz, stride, n, cloud);
    float normals[3][4], mids[3][4];
    int counts[3];
    for (int i = 0; i < 3; ++i) {
        PlaneConstraint pc;
        pc.type = i == 0 ? CD_PLANE_PERPENDICULAR : CD_PLANE_PARALLEL;   // sne.cpp:188,192
        for (int a = 0; a < 3; ++a) pc.axis[a] = table_normal[a];
        pc.eps = 0.1;                                                    // sne.cpp:123
        const int m = (int)(cloud.size() / 3);
        PlaneOut po;
        const int st = segment_plane(cloud.data(), m, prm->plane_distance_threshold, 1000, prm->plane_probability, true, po, nullptr, pc);
        out->iterations[i] = po.iterations;
        if (st != CD_OK) return st;
        std::vector<char> is_inl((size_t)std::max(m, 1), 0);
        for (int k : po.inliers) is_inl[(size_t)k] = 1;
        std::vector<float> plane_pc, leftover;   // ExtractIndices(negative = !invert) / (negative = invert)
        for (int k = 0; k < m; ++k) {
            std::vector<float>& dst = ((is_inl[(size_t)k] != 0) == (invert != 0)) ? plane_pc : leftover;
            dst.insert(dst.end(), cloud.begin() + 3 * (size_t)k, cloud.begin() + 3 * (size_t)k + 3);
        }
        float cs[3] = {0.f, 0.f, 0.f};           // pcl::compute3DCentroid: sequential float32 sums
        const size_t np = plane_pc.size() / 3;
        for (size_t k = 0; k < np; ++k)
            for (int a = 0; a < 3; ++a) cs[a] += plane_pc[3 * k + a];
        for (int a = 0; a < 3; ++a) mids[i][a] = cs[a] / (float)np;
        mids[i][3] = 0.f;
        std::memcpy(normals[i], po.coeff, 16);
        counts[i] = (int)np;
        cloud.swap(leftover);
    }
    for (int i = 0; i < 3; ++i)                  // sne.cpp:199-212
        for (int j = i; j < 3; ++j)
            if (counts[i] < counts[j]) {
                std::swap(counts[i], counts[j]);
                for (int a = 0; a < 4; ++a) {
                    std::swap(normals[i][a], normals[j][a]);
                    std::swap(mids[i][a], mids[j][a]);
                }
            }
    const float* n0 = normals[0];
    const float* n1 = normals[1];
    float n2[3] = {normals[2][0], normals[2][1], normals[2][2]};
    float cr[3];
    cross3(n1, n0, cr);                          // sne.cpp:207: normals[2].dot(normals[1].cross(normals[0]))
    if ((n2[0] * cr[0] + n2[1] * cr[1]) + n2[2] * cr[2] < 0.f)
        for (int a = 0; a < 3; ++a) n2[a] = -n2[a];
    const float d[3] = {mids[0][0] - mids[1][0], mids[0][1] - mids[1][1], mids[0][2] - mids[1][2]};
    const float proj = (n0[0] * d[0] + n0[1] * d[1]) + n0[2] * d[2];   // sne.cpp:213
    for (int r = 0; r < 3; ++r) {
        out->Rt[4 * r + 0] = n2[r];
        out->Rt[4 * r + 1] = n1[r];
        out->Rt[4 * r + 2] = n0[r];
        out->Rt[4 * r + 3] = mids[0][r] - proj * n0[r];
    }
    out->Rt[15] = 1.f;
    for (int i = 0; i < 3; ++i) {
        out->n_points[i] = counts[i];
        std::memcpy(out->coeff[i], normals[i], 16);
        std::memcpy(out->midpoint[i], mids[i], 16);
    }
    return CD_OK;
}

int orc_bbox_filter(const void* xyz, size_t stride, int n, const double P[12], const int32_t rect[4], int32_t* out_idx, int* out_n) {
    Cloud c{(const uint8_t*)xyz, stride, n};
    int k = 0;
    const int r[4] = {rect[0], rect[1], rect[2], rect[3]};
    for (int i = 0; i < n; ++i)
        if (within_bbox(P, r, c.x(i), c.y(i), c.z(i))) out_idx[k++] = i;
    *out_n = k;
    return CD_OK;
}

int orc_nn(const void* tgt, size_t tstride, int m, const void* q, size_t qstride, int n, int mode,
           int32_t* idx, float* d2) {
    std::vector<float> T, Q;
    gather_xyz(tgt, tstride, m, T);
    gather_xyz(q, qstride, n, Q);
    KdTree kd;
    if (mode == 1) kd.build(T.data(), m);
    for (int i = 0; i < n; ++i) {
        float d = FLT_MAX;
        int j = INT_MAX;
        if (mode == 1)
            kd.query(0, Q.data() + 3 * (size_t)i, d, j);
        else
            nn_brute(T.data(), m, Q.data() + 3 * (size_t)i, d, j);
        idx[i] = j;
        d2[i] = d;
    }
    return CD_OK;
}

int orc_icp(const void* tgt, size_t tstride, int m, const void* src, size_t sstride, int n,
            const cd_params* prm, int nn_mode, cd_cluster_result* out, float* aligned) {
    std::vector<float> T, S;
    gather_xyz(tgt, tstride, m, T);
    gather_xyz(src, sstride, n, S);
    IcpOut io;
    const int st = icp_align(S.data(), n, T.data(), m, nn_mode, prm->icp_max_iterations,
                             prm->icp_transformation_epsilon, prm->icp_euclidean_fitness_epsilon, io,
                             prm->icp_use_guess != CD_GUESS_NONE ? prm->icp_guess : nullptr);
    fill_cluster_result(io, n, prm->icp_accept_fitness, out);
    if (aligned) std::memcpy(aligned, io.aligned.data(), (size_t)n * 12);
    return st;
}

int orc_svd3(const float A[9], float U[9], float S[3], float V[9]) {
    float a[3][3], u[3][3], v[3][3];
    std::memcpy(a, A, 36);
    jacobi_svd3(a, u, S, v);
    std::memcpy(U, u, 36);
    std::memcpy(V, v, 36);
    return CD_OK;
}

int orc_eigen33_smallest(const float cov[9], float evec[3]) {
    float c[3][3];
    std::memcpy(c, cov, 36);
    eigen33_smallest(c, evec);
    return CD_OK;
}

int orc_mat4_inverse(const double m[16], double inv[16]) { return mat4_inverse(m, inv) ? CD_OK : CD_ERR_INVALID_ARG; }

// tf::Matrix3x3::getRotation (icp.cpp:62-67) + position (icp.cpp:59)
void orc_pose_to_position_quaternion(const double H[16], double pos[3], double q[4]) {
    pos[0] = H[3];
    pos[1] = H[7];
    pos[2] = H[11];
    const double m[3][3] = {{H[0], H[1], H[2]}, {H[4], H[5], H[6]}, {H[8], H[9], H[10]}};
    const double trace = m[0][0] + m[1][1] + m[2][2];
    double t[4];
    if (trace > 0.0) {
        double s = std::sqrt(trace + 1.0);
        t[3] = s * 0.5;
        s = 0.5 / s;
        t[0] = (m[2][1] - m[1][2]) * s;
        t[1] = (m[0][2] - m[2][0]) * s;
        t[2] = (m[1][0] - m[0][1]) * s;
    } else {
        const int i = m[0][0] < m[1][1] ? (m[1][1] < m[2][2] ? 2 : 1) : (m[0][0] < m[2][2] ? 2 : 0);
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        double s = std::sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
        t[i] = s * 0.5;
        s = 0.5 / s;
        t[3] = (m[k][j] - m[j][k]) * s;
        t[j] = (m[j][i] + m[i][j]) * s;
        t[k] = (m[k][i] + m[i][k]) * s;
    }
    q[0] = t[0];
    q[1] = t[1];
    q[2] = t[2];
    q[3] = t[3];
}

// publish_bounding_box (icp.cpp:90-110): 8 corners, order of icp.cpp:99-106, H.cast<float>()
void orc_bbox_corners(const double H[16], double l, double w, double h, float out[24]) {
    float Hf[16];
    for (int i = 0; i < 16; ++i) Hf[i] = (float)H[i];
    const double sx[8] = {-1, -1, -1, -1, 1, 1, 1, 1};
    const double sy[8] = {-1, -1, 1, 1, -1, -1, 1, 1};
    const double sz[8] = {-1, 1, -1, 1, -1, 1, -1, 1};
    for (int k = 0; k < 8; ++k) {
        const float p[3] = {(float)(sx[k] * l / 2), (float)(sy[k] * w / 2), (float)(sz[k] * h / 2)};
        xform(Hf, p, out + 3 * k);
    }
}

// Whole chain for one frame: the stage order of opd.cpp:270-413 with the cuboid launch
// parameters.  plane_inliers / labels may be NULL; capacities are n.
// all_clusters (may be NULL): the ICP result of EVERY cluster (opd.cpp:376-413 loops over all of object_cluster_indices;
// the fixed-size record only has room for the CD_MAX_CLUSTERS_PER_FRAME largest), capacity all_cap, count in *n_all.
int orc_process_frame_all(const void* pts, size_t stride, int n, const cd_params* prm, const void* tgt,
                          size_t tstride, int m, int nn_mode, cd_frame_result* res, int32_t* plane_inliers,
                          int32_t* labels, float* voxel_xyz /* n*3 or NULL */, float* object_xyz /* n*3 or NULL */,
                          cd_cluster_result* all_clusters, int all_cap, int* n_all) {
    std::memset(res, 0, sizeof(*res));
    if (n_all) *n_all = 0;
    Cloud c{(const uint8_t*)pts, stride, n};
    std::vector<int> a, b;
    passthrough(c, nullptr, 2, prm->crop_z_min, prm->crop_z_max, a);
    passthrough(c, &a, 0, prm->crop_x_min, prm->crop_x_max, b);
    res->n_cropped = (int)b.size();
    VoxelOut vo;
    int st = voxel_grid(c, b, prm->leaf_size, prm->rgb_offset, vo);
    if (st != CD_OK) {
        res->status = st;
        return st;
    }
    const int nv = (int)vo.rgb.size();
    res->n_voxels = nv;
    if (voxel_xyz) std::memcpy(voxel_xyz, vo.xyz.data(), (size_t)nv * 12);
    if (plane_inliers)
        for (int i = 0; i < n; ++i) plane_inliers[i] = -1;
    if (labels)
        for (int i = 0; i < n; ++i) labels[i] = -1;
    PlaneOut po;
    PlaneConstraint pcon;
    pcon.type = prm->plane_model;
    for (int a = 0; a < 3; ++a) pcon.axis[a] = prm->plane_axis[a];
    pcon.eps = prm->plane_eps_angle;
    st = segment_plane(vo.xyz.data(), nv, prm->plane_distance_threshold, prm->plane_max_iterations,
                       prm->plane_probability, prm->plane_optimize != 0, po, nullptr, pcon);
    res->ransac_iterations = po.iterations;
    std::vector<char> is_inl(nv, 0);
    if (st == CD_OK) {
        res->n_plane = (int)po.inliers.size();
        std::memcpy(res->plane, po.coeff, 16);
        for (int i : po.inliers) is_inl[i] = 1;
        if (plane_inliers) std::memcpy(plane_inliers, po.inliers.data(), po.inliers.size() * 4);
    } else {
        res->status = st;  // PCL: empty inliers/coefficients, node carries on (gps.cpp:93-107)
    }
    // S3 ExtractIndices(negative) + S3b PassThrough z
    std::vector<float> obj;
    for (int i = 0; i < nv; ++i) {
        const bool keep = prm->extract_negative ? !is_inl[i] : is_inl[i];
        if (!keep) continue;
        const float* p = vo.xyz.data() + 3 * (size_t)i;
        if (prm->crop2_enable) {
            const double v = (double)p[2];
            if (v > prm->crop2_z_max || v < prm->crop2_z_min) continue;
        }
        if (prm->bbox_enable && !within_bbox(prm->bbox_P, prm->bbox_rect, p[0], p[1], p[2])) continue;
        obj.insert(obj.end(), p, p + 3);
    }
    const int no = (int)(obj.size() / 3);
    res->n_objects = no;
    if (object_xyz) std::memcpy(object_xyz, obj.data(), obj.size() * 4);
    std::vector<int> lab(no, 0), sz;
    if (prm->cluster_enable) {
        cluster(obj.data(), no, prm->cluster_tolerance, prm->cluster_min_size, prm->cluster_max_size, 1, lab, sz);
    } else if (no > 0) {
        sz.push_back(no);
    }
    res->n_clusters = (int)sz.size();
    res->flags = (int)sz.size() > CD_MAX_CLUSTERS_PER_FRAME ? CD_FRAME_MORE_CLUSTERS : 0;
    if (labels) std::memcpy(labels, lab.data(), (size_t)no * 4);
    const int k_icp = all_clusters ? std::min((int)sz.size(), std::max(all_cap, CD_MAX_CLUSTERS_PER_FRAME)) : std::min((int)sz.size(), CD_MAX_CLUSTERS_PER_FRAME);
    for (int k = 0; k < k_icp; ++k) {
        std::vector<float> src;
        src.reserve((size_t)sz[k] * 3);
        for (int i = 0; i < no; ++i)
            if (lab[i] == k) src.insert(src.end(), obj.begin() + 3 * (size_t)i, obj.begin() + 3 * (size_t)i + 3);
        std::vector<float> T;
        gather_xyz(tgt, tstride, m, T);
        IcpOut io;
        icp_align(src.data(), sz[k], T.data(), m, nn_mode, prm->icp_max_iterations,
                  prm->icp_transformation_epsilon, prm->icp_euclidean_fitness_epsilon, io,
                  prm->icp_use_guess != CD_GUESS_NONE ? prm->icp_guess : nullptr);   // one frame per call: its guess is in prm
        cd_cluster_result cr;
        fill_cluster_result(io, sz[k], prm->icp_accept_fitness, &cr);
        if (k < CD_MAX_CLUSTERS_PER_FRAME) res->clusters[k] = cr;
        if (all_clusters && k < all_cap) { all_clusters[k] = cr; if (n_all) *n_all = k + 1; }
    }
    return CD_OK;
}

// the same chain under an arithmetic variant (ARITH_* bits): how far is the canonical result from another execution PCL allows?
int orc_process_frame_arith(int arith, const void* pts, size_t stride, int n, const cd_params* prm, const void* tgt,
                            size_t tstride, int m, int nn_mode, cd_frame_result* res, int32_t* plane_inliers,
                            int32_t* labels, cd_cluster_result* all_clusters, int all_cap, int* n_all, double* probe /* 8 or NULL */) {
    t_arith = arith;
    for (double& v : t_probe) v = 0.0;
    const int st = orc_process_frame_all(pts, stride, n, prm, tgt, tstride, m, nn_mode, res, plane_inliers, labels, nullptr, nullptr,
                                         all_clusters, all_cap, n_all);
    t_arith = 0;
    if (probe) for (int k = 0; k < 8; ++k) probe[k] = t_probe[k];
    return st;
}

int orc_process_frame(const void* pts, size_t stride, int n, const cd_params* prm, const void* tgt,
                      size_t tstride, int m, int nn_mode, cd_frame_result* res, int32_t* plane_inliers,
                      int32_t* labels, float* voxel_xyz, float* object_xyz) {
    return orc_process_frame_all(pts, stride, n, prm, tgt, tstride, m, nn_mode, res, plane_inliers, labels, voxel_xyz, object_xyz,
                                 nullptr, 0, nullptr);
}

}  // extern "C"
