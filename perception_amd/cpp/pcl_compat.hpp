// pcl_compat.hpp - header-only C++ mirror of the PCL classes the reference's callbacks use, over
// the C-ABI of include/cuboid_hip.h.  Same class/method names, argument meaning and error
// behaviour as the PCL objects at the cited call sites, so a node ports by switching the
// namespace (pcl:: -> pclhip::) and the include.  Point layouts are PCL's: PointXYZ is 16 bytes
// (x,y,z,pad), PointXYZRGB 32 bytes (x,y,z,pad,rgb,pad[3]), so pcl::PointCloud<T>::points.data()
// can also be handed to the C-ABI directly.
//
// Reference call sites: gps.cpp = cuboid_detection/src/ground_plane_segmentation.cpp,
// icp.cpp = cuboid_detection/src/iterative_closest_point.cpp,
// opd.cpp = object_detection/src/object_pose_detection.cpp.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/cuboid_hip.h"

namespace pclhip {

struct alignas(16) PointXYZ {
    float x = 0, y = 0, z = 0, pad = 1.f;
    PointXYZ() = default;
    PointXYZ(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
struct alignas(16) PointXYZRGB {
    float x = 0, y = 0, z = 0, pad = 1.f;
    float rgb = 0;   // packed 0x00RRGGBB viewed as float, as in PCL
    float pad2[3] = {0, 0, 0};
};
static_assert(sizeof(PointXYZ) == 16 && sizeof(PointXYZRGB) == 32, "PCL point layouts");

template <class PointT>
struct PointCloud {
    using Ptr = std::shared_ptr<PointCloud<PointT>>;
    using ConstPtr = std::shared_ptr<const PointCloud<PointT>>;
    std::vector<PointT> points;
    uint32_t width = 0, height = 1;
    bool is_dense = true;
    size_t size() const { return points.size(); }
    void push_back(const PointT& p) { points.push_back(p); width = (uint32_t)points.size(); height = 1; }
    void clear() { points.clear(); width = 0; height = 1; }
};
struct PointIndices {
    using Ptr = std::shared_ptr<PointIndices>;
    std::vector<int> indices;
};
struct ModelCoefficients {
    using Ptr = std::shared_ptr<ModelCoefficients>;
    std::vector<float> values;
};

// One GPU context shared by the objects of a node (the reference's nodes are single-threaded).
class Device {
public:
    static Device& instance(int max_points = 640 * 480, int max_frames = 1, int device_id = 0) {
        static Device d(max_points, max_frames, device_id);
        return d;
    }
    cd_context* ctx() const { return ctx_; }
    ~Device() { cd_destroy(ctx_); }
private:
    Device(int max_points, int max_frames, int device_id) {
        const int st = cd_create(device_id, max_points, max_frames, &ctx_);
        if (st != CD_OK) throw std::runtime_error("cd_create failed: no usable MI355X/HIP device (there is no CPU fallback)");
    }
    cd_context* ctx_ = nullptr;
};

// gps.cpp:53-73 / opd.cpp:273-298: PassThrough("z") + PassThrough("x") + VoxelGrid in one device pass.
template <class PointT>
class CropVoxelGrid {
public:
    void setInputCloud(const typename PointCloud<PointT>::ConstPtr& c) { in_ = c; }
    void setFilterLimitsZ(double lo, double hi) { prm_.crop_z_min = lo; prm_.crop_z_max = hi; }   // pass_z.setFilterLimits
    void setFilterLimitsX(double lo, double hi) { prm_.crop_x_min = lo; prm_.crop_x_max = hi; }   // pass.setFilterLimits
    void setLeafSize(float lx, float, float) { prm_.leaf_size = lx; }                              // downsample.setLeafSize
    // returns false (and an empty cloud) on "Leaf size is too small"
    bool filter(PointCloud<PointT>& out) {
        out.clear();
        if (!in_ || in_->points.empty()) return true;
        const int n = (int)in_->points.size();
        std::vector<float> xyz((size_t)n * 3);
        std::vector<uint32_t> rgb((size_t)n);
        prm_.rgb_offset = sizeof(PointT) >= 32 ? 16 : -1;
        int nc = 0, nv = 0;
        const int st = cd_crop_voxel(Device::instance().ctx(), in_->points.data(), sizeof(PointT), n, &prm_, xyz.data(), rgb.data(), n, &nc, &nv);
        if (st != CD_OK) { std::fprintf(stderr, "[pclhip::VoxelGrid] %s\n", cd_last_error(Device::instance().ctx())); return false; }
        out.points.resize((size_t)nv);
        for (int i = 0; i < nv; ++i) {
            PointT p;
            p.x = xyz[3 * i]; p.y = xyz[3 * i + 1]; p.z = xyz[3 * i + 2];
            if constexpr (sizeof(PointT) >= 32) std::memcpy(&p.rgb, &rgb[i], 4);
            out.points[(size_t)i] = p;
        }
        out.width = (uint32_t)nv;
        return true;
    }
private:
    cd_params prm_ = defaults();
    typename PointCloud<PointT>::ConstPtr in_;
    static cd_params defaults() { cd_params p; cd_default_params(&p); return p; }
};

enum { SACMODEL_PLANE = 0, SACMODEL_PARALLEL_PLANE = 15, SACMODEL_PERPENDICULAR_PLANE = 9 };   // PCL's pcl::SacModel values
enum { SAC_RANSAC = 0 };

// gps.cpp:78-93: pcl::SACSegmentation<PointXYZ>
template <class PointT>
class SACSegmentation {
public:
    SACSegmentation() { cd_default_params(&prm_); }
    void setOptimizeCoefficients(bool b) { prm_.plane_optimize = b ? 1 : 0; }
    // SACMODEL_PLANE (gps.cpp:86), SACMODEL_PERPENDICULAR_PLANE / SACMODEL_PARALLEL_PLANE (sne.cpp:120,188,192)
    void setModelType(int m) {
        if (m == SACMODEL_PLANE) prm_.plane_model = CD_PLANE;
        else if (m == SACMODEL_PERPENDICULAR_PLANE) prm_.plane_model = CD_PLANE_PERPENDICULAR;
        else if (m == SACMODEL_PARALLEL_PLANE) prm_.plane_model = CD_PLANE_PARALLEL;
        else throw std::invalid_argument("unsupported SAC model");
    }
    template <class Vec3> void setAxis(const Vec3& ax) { for (int i = 0; i < 3; ++i) prm_.plane_axis[i] = (float)ax[i]; }   // sne.cpp:122
    void setEpsAngle(double ea) { prm_.plane_eps_angle = ea; }                                                                 // sne.cpp:123
    void setMethodType(int m) { if (m != SAC_RANSAC) throw std::invalid_argument("only SAC_RANSAC"); }
    void setMaxIterations(int n) { prm_.plane_max_iterations = n; }
    void setDistanceThreshold(double t) { prm_.plane_distance_threshold = t; }
    void setProbability(double p) { prm_.plane_probability = p; }
    void setInputCloud(const typename PointCloud<PointT>::ConstPtr& c) { in_ = c; }
    // PCL: on failure inliers and coefficients are left empty and the node carries on (gps.cpp:93-107)
    void segment(PointIndices& inliers, ModelCoefficients& coefficients) {
        inliers.indices.clear();
        coefficients.values.clear();
        if (!in_ || in_->points.empty()) return;
        const int n = (int)in_->points.size();
        std::vector<int32_t> idx((size_t)n);
        float c[4];
        int ni = 0, it = 0;
        const int st = cd_segment_plane(Device::instance().ctx(), in_->points.data(), sizeof(PointT), n, &prm_, c, idx.data(), n, &ni, &it);
        if (st != CD_OK) {
            std::fprintf(stderr, "[pclhip::SACSegmentation::segment] Error segmenting the model! No solution found.\n");
            return;
        }
        inliers.indices.assign(idx.begin(), idx.begin() + ni);
        coefficients.values.assign(c, c + 4);
    }
private:
    cd_params prm_;
    typename PointCloud<PointT>::ConstPtr in_;
};

// gps.cpp:96-101: pcl::ExtractIndices (index bookkeeping only; no arithmetic)
template <class PointT>
class ExtractIndices {
public:
    void setInputCloud(const typename PointCloud<PointT>::ConstPtr& c) { in_ = c; }
    void setIndices(const PointIndices::Ptr& i) { idx_ = i; }
    void setNegative(bool n) { neg_ = n; }
    void filter(PointCloud<PointT>& out) {
        out.clear();
        if (!in_) return;
        std::vector<char> sel(in_->points.size(), 0);
        if (idx_) for (int i : idx_->indices) if (i >= 0 && (size_t)i < sel.size()) sel[(size_t)i] = 1;
        for (size_t i = 0; i < sel.size(); ++i)
            if ((sel[i] != 0) != neg_) out.points.push_back(in_->points[i]);
        out.width = (uint32_t)out.points.size();
    }
private:
    typename PointCloud<PointT>::ConstPtr in_;
    PointIndices::Ptr idx_;
    bool neg_ = false;
};

// opd.cpp:345-362: pcl::search::KdTree + pcl::EuclideanClusterExtraction
template <class PointT>
class EuclideanClusterExtraction {
public:
    EuclideanClusterExtraction() { cd_default_params(&prm_); }
    void setClusterTolerance(double t) { prm_.cluster_tolerance = t; }
    void setMinClusterSize(int n) { prm_.cluster_min_size = n; }
    void setMaxClusterSize(int n) { prm_.cluster_max_size = n; }
    template <class Tree> void setSearchMethod(const Tree&) {}   // the spatial hash replaces the kd-tree
    void setInputCloud(const typename PointCloud<PointT>::ConstPtr& c) { in_ = c; }
    // clusters size-descending, each cluster's indices ascending (as PCL returns them)
    void extract(std::vector<PointIndices>& clusters) {
        clusters.clear();
        if (!in_ || in_->points.empty()) return;
        const int n = (int)in_->points.size();
        std::vector<int32_t> lab((size_t)n), sizes((size_t)n);
        int k = 0;
        const int st = cd_cluster(Device::instance().ctx(), in_->points.data(), sizeof(PointT), n, &prm_, lab.data(), sizes.data(), n, &k);
        if (st != CD_OK) { std::fprintf(stderr, "[pclhip::EuclideanClusterExtraction] %s\n", cd_last_error(Device::instance().ctx())); return; }
        clusters.resize((size_t)k);
        for (int c = 0; c < k; ++c) clusters[(size_t)c].indices.reserve((size_t)sizes[(size_t)c]);
        for (int i = 0; i < n; ++i) if (lab[(size_t)i] >= 0) clusters[(size_t)lab[(size_t)i]].indices.push_back(i);
    }
private:
    cd_params prm_;
    typename PointCloud<PointT>::ConstPtr in_;
};

// icp.cpp:170-182 / opd.cpp:220-235: pcl::IterativeClosestPoint<PointXYZ,PointXYZ>
template <class PointSource, class PointTarget>
class IterativeClosestPoint {
public:
    using Matrix4 = std::array<float, 16>;   // row-major
    IterativeClosestPoint() { cd_default_params(&prm_); final_.fill(0.f); for (int i = 0; i < 4; ++i) final_[5 * i] = 1.f; }
    void setInputSource(const typename PointCloud<PointSource>::ConstPtr& c) { src_ = c; }
    void setInputTarget(const typename PointCloud<PointTarget>::ConstPtr& c) {
        tgt_ = c;
        if (c && !c->points.empty())
            cd_set_template(Device::instance().ctx(), slot_, c->points.data(), sizeof(PointTarget), (int)c->points.size());
    }
    void setTemplateSlot(int s) { slot_ = s; }
    void setMaximumIterations(int n) { prm_.icp_max_iterations = n; }
    void setTransformationEpsilon(double e) { prm_.icp_transformation_epsilon = e; }
    void setEuclideanFitnessEpsilon(double e) { prm_.icp_euclidean_fitness_epsilon = e; }
    void setRANSACOutlierRejectionThreshold(double) {}   // inert in PCL too: no rejector is installed (icp.cpp:177)
    void align(PointCloud<PointSource>& output) {
        output.clear();
        converged_ = false;
        fitness_ = std::numeric_limits<double>::max();
        if (!src_ || !tgt_) return;
        const int n = (int)src_->points.size();
        std::vector<float> al((size_t)std::max(n, 1) * 3);
        cd_cluster_result r;
        const int st = cd_icp(Device::instance().ctx(), slot_, src_->points.data(), sizeof(PointSource), n, &prm_, &r, al.data());
        if (st != CD_OK && st != CD_ERR_FEW_CORRESPONDENCES) { std::fprintf(stderr, "[pclhip::IterativeClosestPoint] %s\n", cd_last_error(Device::instance().ctx())); return; }
        if (st == CD_ERR_FEW_CORRESPONDENCES) std::fprintf(stderr, "[pclhip::IterativeClosestPoint] Not enough correspondences found. Relax your threshold parameters.\n");
        std::memcpy(final_.data(), r.T, 64);
        std::memcpy(pose_.data(), r.pose, 128);
        converged_ = r.converged != 0;
        fitness_ = r.fitness;
        iterations_ = r.iterations;
        output.points.resize((size_t)n);
        for (int i = 0; i < n; ++i) { output.points[(size_t)i] = src_->points[(size_t)i]; output.points[(size_t)i].x = al[3 * i]; output.points[(size_t)i].y = al[3 * i + 1]; output.points[(size_t)i].z = al[3 * i + 2]; }
        output.width = (uint32_t)n;
    }
    Matrix4 getFinalTransformation() const { return final_; }
    // getFinalTransformation().cast<double>().inverse() of icp.cpp:179, computed once by the library
    std::array<double, 16> getFinalTransformationInverse() const { return pose_; }
    bool hasConverged() const { return converged_; }
    double getFitnessScore() const { return fitness_; }
    int getIterations() const { return iterations_; }
private:
    cd_params prm_;
    int slot_ = 0, iterations_ = 0;
    typename PointCloud<PointSource>::ConstPtr src_;
    typename PointCloud<PointTarget>::ConstPtr tgt_;
    Matrix4 final_;
    std::array<double, 16> pose_{};
    bool converged_ = false;
    double fitness_ = 0;
};

// pcl::io::loadPCDFile<PointXYZ> for ASCII/binary PCD v0.7 with float x y z fields (icp.cpp:159)
namespace io {
inline int loadPCDFile(const std::string& path, PointCloud<PointXYZ>& cloud) {
    cloud.clear();
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return -1;
    char line[512];
    std::vector<std::string> fields;
    std::vector<int> sizes;
    long npts = -1;
    bool binary = false, have_data = false;
    while (std::fgets(line, sizeof(line), f)) {
        std::string s(line);
        if (s.rfind("FIELDS", 0) == 0 || s.rfind("SIZE", 0) == 0) {
            const bool is_f = s[0] == 'F';
            size_t p = s.find(' ');
            while (p != std::string::npos) {
                const size_t q = s.find_first_of(" \r\n", p + 1);
                const std::string tok = s.substr(p + 1, q == std::string::npos ? q : q - p - 1);
                if (!tok.empty()) { if (is_f) fields.push_back(tok); else sizes.push_back(std::atoi(tok.c_str())); }
                p = (q == std::string::npos || s[q] != ' ') ? std::string::npos : q;
            }
        } else if (s.rfind("POINTS", 0) == 0) {
            npts = std::atol(s.c_str() + 7);
        } else if (s.rfind("DATA", 0) == 0) {
            binary = s.find("binary") != std::string::npos;
            have_data = true;
            break;
        }
    }
    int ix = -1, iy = -1, iz = -1;
    for (size_t i = 0; i < fields.size(); ++i) { if (fields[i] == "x") ix = (int)i; if (fields[i] == "y") iy = (int)i; if (fields[i] == "z") iz = (int)i; }
    if (!have_data || npts < 0 || ix < 0 || iy < 0 || iz < 0) { std::fclose(f); return -1; }
    cloud.points.resize((size_t)npts);
    if (!binary) {
        std::vector<double> row(fields.size());
        for (long i = 0; i < npts; ++i) {
            for (size_t k = 0; k < fields.size(); ++k) if (std::fscanf(f, "%lf", &row[k]) != 1) { std::fclose(f); return -1; }
            cloud.points[(size_t)i] = PointXYZ((float)row[(size_t)ix], (float)row[(size_t)iy], (float)row[(size_t)iz]);
        }
    } else {
        size_t step = 0;
        std::vector<size_t> off(fields.size());
        for (size_t k = 0; k < fields.size(); ++k) { off[k] = step; step += (size_t)(k < sizes.size() ? sizes[k] : 4); }
        std::vector<unsigned char> rec(step);
        for (long i = 0; i < npts; ++i) {
            if (std::fread(rec.data(), 1, step, f) != step) { std::fclose(f); return -1; }
            PointXYZ p;
            std::memcpy(&p.x, &rec[off[(size_t)ix]], 4); std::memcpy(&p.y, &rec[off[(size_t)iy]], 4); std::memcpy(&p.z, &rec[off[(size_t)iz]], 4);
            cloud.points[(size_t)i] = p;
        }
    }
    cloud.width = (uint32_t)npts;
    std::fclose(f);
    return 0;
}
}  // namespace io

}  // namespace pclhip
