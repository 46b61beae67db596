#pragma once
// declarations only - see tests/pcl_stubs/README.md
#include <cstdint>
#include <memory>
#include <string>
#include <vector>
#define PCL_MAJOR_VERSION 0
#define PCL_MINOR_VERSION 0
#define PCL_REVISION_VERSION 0
namespace Eigen {
template <class S> struct Matrix4 {
    S operator()(int, int) const;
    template <class U> Matrix4<U> cast() const;
    Matrix4 inverse() const;
};
typedef Matrix4<float> Matrix4f;
typedef Matrix4<double> Matrix4d;
}  // namespace Eigen
namespace pcl {
struct PointXYZ { float x, y, z; };
struct PointXYZRGB { float x, y, z, rgb; };
template <class P> struct PointCloud {
    typedef std::shared_ptr<PointCloud> Ptr;
    typedef std::shared_ptr<const PointCloud> ConstPtr;
    std::vector<P> points;
    uint32_t width = 0, height = 0;
    bool is_dense = true;
    size_t size() const;
};
struct PCLPointCloud2 { uint32_t width = 0, height = 0; };
typedef std::shared_ptr<const PCLPointCloud2> PCLPointCloud2ConstPtr;
struct PointIndices { typedef std::shared_ptr<PointIndices> Ptr; std::vector<int> indices; };
struct ModelCoefficients { typedef std::shared_ptr<ModelCoefficients> Ptr; std::vector<float> values; };
template <class P> void toPCLPointCloud2(const PointCloud<P>&, PCLPointCloud2&);
template <class P> void fromPCLPointCloud2(const PCLPointCloud2&, PointCloud<P>&);
enum { SACMODEL_PLANE = 0 };
enum { SAC_RANSAC = 0 };
template <class T> struct PassThrough {
    void setInputCloud(const PCLPointCloud2ConstPtr&);
    void setFilterFieldName(const std::string&);
    void setFilterLimits(double, double);
    void filter(PCLPointCloud2&);
};
template <class T> struct VoxelGrid {
    void setInputCloud(const PCLPointCloud2ConstPtr&);
    void setLeafSize(float, float, float);
    void filter(PCLPointCloud2&);
};
template <class T> struct ExtractIndices {
    void setInputCloud(const PCLPointCloud2ConstPtr&);
    void setIndices(const PointIndices::Ptr&);
    void setNegative(bool);
    void filter(PCLPointCloud2&);
};
template <class P> struct SACSegmentation {
    void setOptimizeCoefficients(bool);
    void setModelType(int);
    void setMethodType(int);
    void setMaxIterations(int);
    void setDistanceThreshold(double);
    void setInputCloud(const typename PointCloud<P>::Ptr&);
    void segment(PointIndices&, ModelCoefficients&);
};
namespace search {
template <class P> struct KdTree {
    typedef std::shared_ptr<KdTree> Ptr;
    void setInputCloud(const typename PointCloud<P>::Ptr&);
};
}  // namespace search
template <class P> struct EuclideanClusterExtraction {
    void setClusterTolerance(double);
    void setMinClusterSize(int);
    void setMaxClusterSize(int);
    void setSearchMethod(const typename search::KdTree<P>::Ptr&);
    void setInputCloud(const typename PointCloud<P>::Ptr&);
    void extract(std::vector<PointIndices>&);
};
template <class S, class T> struct IterativeClosestPoint {
    void setInputSource(const typename PointCloud<S>::Ptr&);
    void setInputTarget(const typename PointCloud<T>::Ptr&);
    void setMaximumIterations(int);
    void setTransformationEpsilon(double);
    void setEuclideanFitnessEpsilon(double);
    void setRANSACOutlierRejectionThreshold(double);
    void align(PointCloud<S>&);
    Eigen::Matrix4f getFinalTransformation() const;
    bool hasConverged() const;
    double getFitnessScore();
protected:
    int nr_iterations_;
};
namespace io {
template <class P> int loadPCDFile(const std::string&, PointCloud<P>&);
}
}  // namespace pcl
