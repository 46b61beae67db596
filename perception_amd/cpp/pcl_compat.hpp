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
#include <functional>
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

namespace detail {
// `points` of a PointCloud: a std::vector<PointT> whose content may still be owed.  PassThrough::filter leaves its output
// DEFERRED - the filter runs (on the device, cd_passthrough) the first time somebody looks at the points - so that the
// reference's sequence PassThrough("z") -> PassThrough("x") -> VoxelGrid (gps.cpp:53-73, opd.cpp:273-298), whose two
// intermediate clouds nobody ever looks at, becomes ONE fused device call (VoxelGrid::filter below recognises the chain).
// Every accessor forces the content first, so code written against pcl::PointCloud<T>::points reads the same.
template <class PointT>
class LazyPoints {
public:
    using value_type = PointT;
    using iterator = typename std::vector<PointT>::iterator;
    using const_iterator = typename std::vector<PointT>::const_iterator;
    size_t size() const { force(); return v_.size(); }
    bool empty() const { force(); return v_.empty(); }
    PointT* data() { force(); return v_.data(); }
    const PointT* data() const { force(); return v_.data(); }
    PointT& operator[](size_t i) { force(); return v_[i]; }
    const PointT& operator[](size_t i) const { force(); return v_[i]; }
    PointT& back() { force(); return v_.back(); }
    const PointT& back() const { force(); return v_.back(); }
    iterator begin() { force(); return v_.begin(); }
    iterator end() { force(); return v_.end(); }
    const_iterator begin() const { force(); return v_.begin(); }
    const_iterator end() const { force(); return v_.end(); }
    void push_back(const PointT& p) { force(); v_.push_back(p); }
    void resize(size_t n) { force(); v_.resize(n); }
    void reserve(size_t n) { force(); v_.reserve(n); }
    void clear() { pending_ = nullptr; v_.clear(); }
    template <class It> void assign(It a, It b) { pending_ = nullptr; v_.assign(a, b); }
    std::vector<PointT>& vector() { force(); return v_; }
    const std::vector<PointT>& vector() const { force(); return v_; }
    operator const std::vector<PointT>&() const { return vector(); }
    // the content is owed: `make` fills the vector at the first access
    void defer(std::function<void(std::vector<PointT>&)> make) { v_.clear(); pending_ = std::move(make); }
    bool deferred() const { return (bool)pending_; }
private:
    void force() const {
        if (!pending_) return;
        auto f = std::move(pending_);
        pending_ = nullptr;
        f(v_);
    }
    mutable std::vector<PointT> v_;
    mutable std::function<void(std::vector<PointT>&)> pending_;
};
struct Crop { int field; double lo, hi; bool negative; };   // one PassThrough: field 0 / 1 / 2 = "x" / "y" / "z", -1 = none
}  // namespace detail

template <class PointT>
struct PointCloud {
    using Ptr = std::shared_ptr<PointCloud<PointT>>;
    using ConstPtr = std::shared_ptr<const PointCloud<PointT>>;
    detail::LazyPoints<PointT> points;
    uint32_t width = 0, height = 1;   // (of a deferred cloud: set when its points are first looked at - use size())
    bool is_dense = true;
    size_t size() const { return points.size(); }
    void push_back(const PointT& p) { points.push_back(p); width = (uint32_t)points.size(); height = 1; }
    void clear() { points.clear(); width = 0; height = 1; crop_root.reset(); crops.clear(); }
    // provenance of a deferred PassThrough output: this cloud == `crops` applied in order to *crop_root (kept alive here)
    ConstPtr crop_root;
    std::vector<detail::Crop> crops;
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

// gps.cpp:53-65 / opd.cpp:273-289, :331-336: pcl::PassThrough.  filter() leaves the output DEFERRED (LazyPoints above): a
// PassThrough whose output only ever feeds the next filter costs nothing by itself; one that is looked at runs cd_passthrough.
template <class PointT>
class PassThrough {
public:
    void setInputCloud(const typename PointCloud<PointT>::ConstPtr& c) { in_ = c; }
    void setFilterFieldName(const std::string& name) {
        field_ = name == "x" ? 0 : (name == "y" ? 1 : (name == "z" ? 2 : -2));
        if (name.empty()) field_ = -1;
        if (field_ == -2) throw std::invalid_argument("PassThrough: only the fields x, y, z");
    }
    std::string getFilterFieldName() const { return field_ < 0 ? std::string() : std::string(1, "xyz"[field_]); }
    void setFilterLimits(double lo, double hi) { lo_ = lo; hi_ = hi; }
    void setFilterLimitsNegative(bool n) { neg_ = n; }
    void filter(PointCloud<PointT>& out) {
        out.clear();
        if (!in_) return;
        // the chain continues when the input is itself a PassThrough output nobody has looked at; otherwise it starts here
        if (in_->crop_root && in_->points.deferred()) { out.crop_root = in_->crop_root; out.crops = in_->crops; }
        else out.crop_root = in_;
        out.crops.push_back(detail::Crop{field_, lo_, hi_, neg_});
        const typename PointCloud<PointT>::ConstPtr root = out.crop_root;
        const std::vector<detail::Crop> crops = out.crops;
        PointCloud<PointT>* self = &out;
        out.points.defer([root, crops, self](std::vector<PointT>& v) {
            v.assign(root->points.begin(), root->points.end());
            for (const detail::Crop& cr : crops) {
                if (v.empty()) break;
                std::vector<PointT> kept(v.size());
                int n = 0;
                const int st = cd_passthrough(Device::instance().ctx(), v.data(), sizeof(PointT), (int)v.size(), cr.field, cr.lo, cr.hi, cr.negative ? 1 : 0,
                                              kept.data(), (int)kept.size(), &n);
                if (st != CD_OK) { std::fprintf(stderr, "[pclhip::PassThrough] %s\n", cd_last_error(Device::instance().ctx())); n = 0; }
                kept.resize((size_t)n);
                v.swap(kept);
            }
            self->width = (uint32_t)v.size();
            self->height = 1;
        });
    }
private:
    typename PointCloud<PointT>::ConstPtr in_;
    int field_ = -1;
    double lo_ = -std::numeric_limits<float>::max(), hi_ = std::numeric_limits<float>::max();   // PCL's defaults: FLT_MIN .. FLT_MAX as "no limit"
    bool neg_ = false;
};

// gps.cpp:69-73 / opd.cpp:293-298: pcl::VoxelGrid.  When the input is the deferred output of PassThrough filters on "x" / "z"
// (the reference's sequence), the crops and the voxel grid run as ONE device call on the chain's root cloud (cd_crop_voxel:
// the same points in the same order reach the grid, so the result is the one the three filters give one after the other).
template <class PointT>
class VoxelGrid {
public:
    void setInputCloud(const typename PointCloud<PointT>::ConstPtr& c) { in_ = c; }
    void setLeafSize(float lx, float ly, float lz) {
        if (lx != ly || lx != lz) throw std::invalid_argument("VoxelGrid: cubic leaves only (the reference sets one voxel_size, gps.cpp:72)");
        prm_.leaf_size = lx;
    }
    // returns false (and an empty cloud) on "Leaf size is too small" (PCL warns and returns the input; the nodes stop here)
    bool filter(PointCloud<PointT>& out) {
        out.clear();
        if (!in_) return true;
        const double inf = std::numeric_limits<double>::max();
        prm_.crop_z_min = prm_.crop_x_min = -inf;
        prm_.crop_z_max = prm_.crop_x_max = inf;
        typename PointCloud<PointT>::ConstPtr src = in_;
        if (in_->crop_root && in_->points.deferred()) {
            bool fusable = true;
            for (const detail::Crop& cr : in_->crops) fusable = fusable && !cr.negative && (cr.field == 0 || cr.field == 2 || cr.field == -1);
            if (fusable) {
                for (const detail::Crop& cr : in_->crops) {   // (two crops on one field: the intersection)
                    if (cr.field == 2) { prm_.crop_z_min = std::max(prm_.crop_z_min, cr.lo); prm_.crop_z_max = std::min(prm_.crop_z_max, cr.hi); }
                    if (cr.field == 0) { prm_.crop_x_min = std::max(prm_.crop_x_min, cr.lo); prm_.crop_x_max = std::min(prm_.crop_x_max, cr.hi); }
                }
                src = in_->crop_root;
                fused_ = true;
            }
        }
        if (src->points.empty()) return true;
        const int n = (int)src->points.size();
        std::vector<float> xyz((size_t)n * 3);
        std::vector<uint32_t> rgb((size_t)n);
        prm_.rgb_offset = sizeof(PointT) >= 32 ? 16 : -1;
        int nc = 0, nv = 0;
        const int st = cd_crop_voxel(Device::instance().ctx(), src->points.data(), sizeof(PointT), n, &prm_, xyz.data(), rgb.data(), n, &nc, &nv);
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
    bool lastFilterWasFused() const { return fused_; }   // (diagnostic: the crops ran inside the voxel call)
private:
    cd_params prm_ = defaults();
    typename PointCloud<PointT>::ConstPtr in_;
    bool fused_ = false;
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
    // align(output, guess): pcl::Registration's second overload (row-major 4x4, scene -> template)
    void align(PointCloud<PointSource>& output, const Matrix4& guess) {
        prm_.icp_use_guess = CD_GUESS_PARAMS;
        std::memcpy(prm_.icp_guess, guess.data(), 64);
        align(output);
        prm_.icp_use_guess = CD_GUESS_NONE;
    }
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

// pcl::io::loadPCDFile<PointXYZ> (icp.cpp:159, opd.cpp:398): PCD v0.7 (and v0.6's COLUMNS alias) in all three DATA
// encodings.  The header's FIELDS / SIZE / TYPE / COUNT lines give the record layout - a field of COUNT c occupies
// c * SIZE bytes (binary) or c tokens (ascii) -, WIDTH x HEIGHT stands in for a missing POINTS line, x / y / z are picked by
// name and converted to float32 from whatever TYPE / SIZE they are stored as (PCL maps F4 directly; F8 and the integer
// types are cast here rather than dropped).  binary_compressed = two uint32 (compressed, uncompressed size) + an LZF stream
// of the fields stored one after the other (all x, then all y, ...), as pcl::PCDWriter::writeBinaryCompressed lays it out.
// Returns 0, or -1 with a message on stderr (PCL_ERROR at the call sites follows).
namespace io {
namespace detail {
// liblzf's decompressor (format: literal runs `000LLLLL` + L+1 bytes; back references `LLLooooo oooooooo`, LLL = 7 adds a
// length byte): returns the number of bytes produced, 0 on a malformed stream
inline size_t lzf_decompress(const unsigned char* in, size_t in_len, unsigned char* out, size_t out_len) {
    const unsigned char* ip = in;
    const unsigned char* const in_end = in + in_len;
    unsigned char* op = out;
    unsigned char* const out_end = out + out_len;
    while (ip < in_end) {
        unsigned ctrl = *ip++;
        if (ctrl < 32) {
            ++ctrl;
            if (op + ctrl > out_end || ip + ctrl > in_end) return 0;
            std::memcpy(op, ip, ctrl);
            op += ctrl; ip += ctrl;
        } else {
            unsigned len = ctrl >> 5;
            if (ip >= in_end) return 0;
            if (len == 7) { len += *ip++; if (ip >= in_end) return 0; }
            const size_t dist = ((size_t)(ctrl & 0x1f) << 8) + (size_t)*ip++ + 1;
            len += 2;
            if (op + len > out_end || dist > (size_t)(op - out)) return 0;
            const unsigned char* ref = op - dist;
            for (unsigned k = 0; k < len; ++k) *op++ = *ref++;   // may overlap: byte by byte
        }
    }
    return (size_t)(op - out);
}
struct PcdField { std::string name; int size = 4; char type = 'F'; int count = 1; size_t offset = 0; };
inline float pcd_value(const unsigned char* p, const PcdField& f) {
    if (f.type == 'F') { if (f.size == 4) { float v; std::memcpy(&v, p, 4); return v; } if (f.size == 8) { double v; std::memcpy(&v, p, 8); return (float)v; } }
    if (f.type == 'I') { if (f.size == 1) { int8_t v; std::memcpy(&v, p, 1); return (float)v; } if (f.size == 2) { int16_t v; std::memcpy(&v, p, 2); return (float)v; }
                         if (f.size == 4) { int32_t v; std::memcpy(&v, p, 4); return (float)v; } if (f.size == 8) { int64_t v; std::memcpy(&v, p, 8); return (float)v; } }
    if (f.type == 'U') { if (f.size == 1) { uint8_t v; std::memcpy(&v, p, 1); return (float)v; } if (f.size == 2) { uint16_t v; std::memcpy(&v, p, 2); return (float)v; }
                         if (f.size == 4) { uint32_t v; std::memcpy(&v, p, 4); return (float)v; } if (f.size == 8) { uint64_t v; std::memcpy(&v, p, 8); return (float)v; } }
    return std::numeric_limits<float>::quiet_NaN();
}
inline std::vector<std::string> pcd_tokens(const std::string& line) {
    std::vector<std::string> t;
    size_t i = 0;
    while (i < line.size()) {
        while (i < line.size() && std::strchr(" \t\r\n", line[i])) ++i;
        size_t j = i;
        while (j < line.size() && !std::strchr(" \t\r\n", line[j])) ++j;
        if (j > i) t.push_back(line.substr(i, j - i));
        i = j;
    }
    return t;
}
}  // namespace detail

inline int loadPCDFile(const std::string& path, PointCloud<PointXYZ>& cloud) {
    using namespace detail;
    cloud.clear();
    auto err = [&](const char* what) { std::fprintf(stderr, "[pclhip::io::loadPCDFile] %s: %s\n", path.c_str(), what); return -1; };
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return err("cannot open the file");
    struct Closer { FILE* f; ~Closer() { std::fclose(f); } } closer{f};
    std::vector<PcdField> fields;
    long npts = -1, width = -1, height = -1;
    std::string data;
    char line[4096];
    while (std::fgets(line, sizeof(line), f)) {
        const std::vector<std::string> t = pcd_tokens(line);
        if (t.empty() || t[0][0] == '#') continue;
        const std::string& key = t[0];
        if (key == "FIELDS" || key == "COLUMNS") {
            fields.assign(t.size() - 1, PcdField());
            for (size_t k = 1; k < t.size(); ++k) fields[k - 1].name = t[k];
        } else if (key == "SIZE" || key == "TYPE" || key == "COUNT") {
            if (t.size() - 1 != fields.size()) return err("SIZE / TYPE / COUNT does not match FIELDS");
            for (size_t k = 1; k < t.size(); ++k) {
                if (key == "SIZE") fields[k - 1].size = std::atoi(t[k].c_str());
                else if (key == "TYPE") fields[k - 1].type = t[k][0];
                else fields[k - 1].count = std::atoi(t[k].c_str());
            }
        } else if (key == "WIDTH" && t.size() > 1) { width = std::atol(t[1].c_str());
        } else if (key == "HEIGHT" && t.size() > 1) { height = std::atol(t[1].c_str());
        } else if (key == "POINTS" && t.size() > 1) { npts = std::atol(t[1].c_str());
        } else if (key == "DATA" && t.size() > 1) { data = t[1]; break; }
    }
    if (data.empty()) return err("no DATA line");
    // The header is not trusted: counts are bounded by what is left of the file BEFORE anything is allocated (a malformed
    // template must give -1, not bad_alloc / length_error out of a function the nodes call unguarded)
    const long data_pos = std::ftell(f);
    long file_end = data_pos;
    if (data_pos < 0 || std::fseek(f, 0, SEEK_END) != 0 || (file_end = std::ftell(f)) < data_pos || std::fseek(f, data_pos, SEEK_SET) != 0)
        return err("cannot size the file");
    const size_t left = (size_t)(file_end - data_pos);
    if (width > (1l << 31) || height > (1l << 31) || npts > (1l << 31)) return err("WIDTH / HEIGHT / POINTS out of range");
    if (npts < 0 && width >= 0) npts = width * (height >= 0 ? height : 1);     // POINTS is optional: WIDTH x HEIGHT (each <= 2^31: no overflow)
    if (npts < 0) return err("neither POINTS nor WIDTH / HEIGHT");
    int ix = -1, iy = -1, iz = -1;
    size_t step = 0, tokens = 0;
    for (size_t k = 0; k < fields.size(); ++k) {
        PcdField& fd = fields[k];
        if (fd.size <= 0 || fd.count < 0 || !std::strchr("FIU", fd.type)) return err("bad SIZE / TYPE / COUNT entry");
        fd.offset = step;
        step += (size_t)fd.size * (size_t)fd.count;
        tokens += (size_t)fd.count;
        if (fd.count >= 1) { if (fd.name == "x") ix = (int)k; if (fd.name == "y") iy = (int)k; if (fd.name == "z") iz = (int)k; }
    }
    if (ix < 0 || iy < 0 || iz < 0) return err("no x / y / z fields");
    if (step == 0 || step > (1u << 20) || tokens == 0) return err("bad record size");
    // a point takes at least `step` bytes (binary), two bytes per token (ascii: a digit and a separator); binary_compressed is
    // checked against its own size words below
    if (data == "binary" && (size_t)npts > left / step) return err("binary data ends early");
    if (data == "ascii" && (size_t)npts > left / (2 * tokens) + 1) return err("ascii data ends early");
    if (data == "binary_compressed" && (size_t)npts > ((size_t)1 << 32) / step) return err("binary_compressed: header larger than the format's 32-bit sizes");
    try { cloud.points.resize((size_t)npts); } catch (const std::exception&) { cloud.clear(); return err("out of memory"); }
    if (data == "ascii") {
        // token positions of x, y, z within a row (a field of COUNT c contributes c tokens)
        size_t tx = 0, ty = 0, tz = 0, acc = 0;
        for (size_t k = 0; k < fields.size(); ++k) { if ((int)k == ix) tx = acc; if ((int)k == iy) ty = acc; if ((int)k == iz) tz = acc; acc += (size_t)fields[k].count; }
        std::vector<double> row(tokens);
        char tok[128];
        for (long i = 0; i < npts; ++i) {
            for (size_t k = 0; k < tokens; ++k) {
                if (std::fscanf(f, "%127s", tok) != 1) return err("ascii data ends early");
                row[k] = std::strtod(tok, nullptr);                           // handles nan / inf
            }
            cloud.points[(size_t)i] = PointXYZ((float)row[tx], (float)row[ty], (float)row[tz]);
        }
    } else if (data == "binary") {
        std::vector<unsigned char> rec(step);
        for (long i = 0; i < npts; ++i) {
            if (std::fread(rec.data(), 1, step, f) != step) return err("binary data ends early");
            cloud.points[(size_t)i] = PointXYZ(pcd_value(&rec[fields[(size_t)ix].offset], fields[(size_t)ix]), pcd_value(&rec[fields[(size_t)iy].offset], fields[(size_t)iy]),
                                               pcd_value(&rec[fields[(size_t)iz].offset], fields[(size_t)iz]));
        }
    } else if (data == "binary_compressed") {
        uint32_t sz[2];
        if (std::fread(sz, 4, 2, f) != 2) return err("binary_compressed: no size words");
        if ((size_t)sz[1] != step * (size_t)npts) return err("binary_compressed: uncompressed size does not match the header");
        if ((size_t)sz[0] > left - (left >= 8 ? 8 : left)) return err("binary_compressed data ends early");
        // (LZF expands a byte to at most 264 / 2 bytes: a stream this short cannot hold that many)
        if ((size_t)sz[1] > (size_t)sz[0] * 132 + 32) return err("binary_compressed: malformed LZF stream");
        std::vector<unsigned char> in, out;
        try { in.resize(sz[0]); out.resize(sz[1]); } catch (const std::exception&) { cloud.clear(); return err("out of memory"); }
        if (sz[0] && std::fread(in.data(), 1, sz[0], f) != sz[0]) return err("binary_compressed data ends early");
        if (sz[1] && lzf_decompress(in.data(), in.size(), out.data(), out.size()) != out.size()) return err("binary_compressed: malformed LZF stream");
        // fields one after the other: field k occupies npts * size * count bytes starting at npts * offset_k
        auto at = [&](int k, long i) { const PcdField& fd = fields[(size_t)k]; return pcd_value(&out[fd.offset * (size_t)npts + (size_t)i * (size_t)fd.size * (size_t)fd.count], fd); };
        for (long i = 0; i < npts; ++i) cloud.points[(size_t)i] = PointXYZ(at(ix, i), at(iy, i), at(iz, i));
    } else {
        return err("unknown DATA encoding");
    }
    cloud.width = width >= 0 && height > 1 ? (uint32_t)width : (uint32_t)npts;
    cloud.height = width >= 0 && height > 1 ? (uint32_t)height : 1;
    return 0;
}
}  // namespace io

}  // namespace pclhip
