// pcl_golden.cpp - a way to PIN the oracle to real PCL.  NOT part of the product, NOT run here.
//
// The reference holds no test, golden output or recorded frame for its point-cloud path, and PCL cannot be built in this
// image - so oracle/ restates PCL's algorithms and its parity with real PCL is "unpinned" (DESIGN.md section 2).  This program
// is what a maintainer WITH PCL (the reference's dependency: libpcl-all-dev, cuboid_detection/package.xml:57,78; PCL 1.7.2 on
// ROS Kinetic, 1.8.1 on Melodic) runs once to close that gap: it executes, on the synthetic frames the tests use, the exact
// PCL calls of object_detection/src/object_pose_detection.cpp:270-413 (the one place where the whole chain runs in one
// function) with cuboid_detection's launch values, and writes tests/golden/pcl_frames_golden.json in the schema of
// tests/golden/frames_golden.json.  tests/test_pcl_golden.py consumes that file when it exists.
//
//   python tools/write_synth_frames.py /tmp/frames 4            # frame_0.bin .. frame_3.bin (x y z rgb float32) + template.pcd
//   g++ -O2 -std=c++14 tools/pcl_golden.cpp -o pcl_golden $(pkg-config --cflags --libs pcl_common pcl_io pcl_filters
//        pcl_segmentation pcl_search pcl_kdtree pcl_registration pcl_sample_consensus)
//   ./pcl_golden /tmp/frames 4 tests/golden/pcl_frames_golden.json              # flavour "chain" (default)
//   ./pcl_golden /tmp/frames 4 tests/golden/pcl_frames_golden_cuboid.json cuboid
//   ./pcl_golden /tmp/frames 2 tests/golden/pcl_frames_golden_object.json object
//
// Flavours (round 5; the parameter sets the tests and bench.py run, each as the reference's own code runs it):
//   chain   object_pose_detection.cpp:270-413 (second z crop, clusters, one ICP per cluster) with cuboid_detection's launch
//           values (leaf 0.005, threshold 0.015): cd_params' defaults, BASELINE configs 1-4
//   cuboid  the cuboid_detection pair of nodes: ground_plane_segmentation.cpp:53-101 (no second crop) and
//           iterative_closest_point.cpp:170-182 - the WHOLE extracted cloud is the one ICP source (no clustering:
//           cd_params.crop2_enable = 0, cluster_enable = 0), accepted iff hasConverged() && getFitnessScore() < 0.0004
//   object  the chain with object_detection.launch:30-38's values: leaf 0.001, threshold 0.01 (bench.py's object_launch leg)
//
// In this repository it is only PARSED (g++ -fsyntax-only against the minimal stand-in headers of tests/pcl_stubs, like the
// ROS node shims against tests/ros_stubs): that keeps it compiling as the code around it changes; it proves nothing about
// parity and is never claimed as such.
//
// What the file pins, and how the test reads it (tolerances of BASELINE.json's north_star):
//   n_cropped, n_voxels            exact (PassThrough / VoxelGrid are deterministic up to the order inside a voxel)
//   plane inlier indices, labels   exact ("plane indices / cluster labels bit-exact"); VoxelGrid's std::sort leaves the order of
//                                  the points INSIDE a voxel unspecified, so centroids may differ in the last bit: the test
//                                  reports a mismatch of the index lists together with the largest centroid difference
//   ICP pose                       Frobenius norm of the difference < 1e-4 per cluster
#include <pcl/ModelCoefficients.h>
#include <pcl/PCLPointCloud2.h>
#include <pcl/conversions.h>
#include <pcl/filters/extract_indices.h>
#include <pcl/filters/passthrough.h>
#include <pcl/filters/voxel_grid.h>
#include <pcl/io/pcd_io.h>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl/registration/icp.h>
#include <pcl/search/kdtree.h>
#include <pcl/segmentation/extract_clusters.h>
#include <pcl/segmentation/sac_segmentation.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

// ---- SHA-256 (FIPS 180-4), so that the file carries the same digests as frames_golden.json -------------------------------
struct Sha256 {
    uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    unsigned char buf[64];
    size_t fill = 0;
    uint64_t bits = 0;
    static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void block(const unsigned char* p) {
        static const uint32_t K[64] = {
            0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu,
            0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau,
            0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u,
            0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u,
            0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu,
            0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
        uint32_t w[64];
        for (int i = 0; i < 16; ++i) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
        for (int i = 16; i < 64; ++i) {
            const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; ++i) {
            const uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
            const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    void add(const void* data, size_t n) {
        const unsigned char* p = (const unsigned char*)data;
        bits += (uint64_t)n * 8;
        while (n) {
            const size_t take = std::min(n, 64 - fill);
            std::memcpy(buf + fill, p, take);
            fill += take; p += take; n -= take;
            if (fill == 64) { block(buf); fill = 0; }
        }
    }
    std::string hex() {
        const uint64_t nbits = bits;
        const unsigned char one = 0x80, zero = 0;
        add(&one, 1);
        while (fill != 56) add(&zero, 1);
        unsigned char len[8];
        for (int i = 0; i < 8; ++i) len[i] = (unsigned char)(nbits >> (56 - 8 * i));
        add(len, 8);
        char out[65];
        for (int i = 0; i < 8; ++i) std::snprintf(out + 8 * i, 9, "%08x", h[i]);
        return std::string(out, 64);
    }
};
template <class T>
std::string sha_of(const std::vector<T>& v) { Sha256 s; if (!v.empty()) s.add(v.data(), v.size() * sizeof(T)); return s.hex(); }

// registration's iteration count is protected in PCL 1.7 (pcl::Registration::nr_iterations_)
struct Icp : pcl::IterativeClosestPoint<pcl::PointXYZ, pcl::PointXYZ> {
    int iterations() const { return nr_iterations_; }
};

// the launch values (ground_plane_segmentation.launch:14-18, iterative_closest_point.launch:42; object_detection.launch:30-38)
// and opd.cpp's constants
struct Flavour {
    const char* name;
    double voxel_size, distance_threshold;
    bool second_crop, clusters;
    const char* params;
};
const Flavour flavours[] = {
    {"chain", 0.005, 0.015, true, true, "opd.cpp:270-413 with the cuboid launch values: leaf 0.005, threshold 0.015, second crop z [0, 0.75], clusters 0.02/200/25000, ICP 5000/1e-9/0.0004"},
    {"cuboid", 0.005, 0.015, false, false, "gps.cpp:53-101 + icp.cpp:170-182: leaf 0.005, threshold 0.015, no second crop, the whole extracted cloud as one ICP source, ICP 5000/1e-9/0.0004"},
    {"object", 0.001, 0.01, true, true, "opd.cpp:270-413 with object_detection.launch's values: leaf 0.001, threshold 0.01, second crop z [0, 0.75], clusters 0.02/200/25000, ICP 5000/1e-9/0.0004"},
};
const double icp_fitness_score = 0.0004;
const bool invert = true;

std::string hexf(double v) { char b[64]; std::snprintf(b, sizeof(b), "%a", v); return b; }

}  // namespace

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: pcl_golden <dir with frame_<i>.bin and template.pcd> <n_frames> <out.json> [chain|cuboid|object]\n"); return 2; }
    const Flavour* fl = &flavours[0];
    if (argc > 4) {
        fl = nullptr;
        for (const Flavour& c : flavours) if (std::strcmp(c.name, argv[4]) == 0) fl = &c;
        if (!fl) { std::fprintf(stderr, "unknown flavour %s\n", argv[4]); return 2; }
    }
    const double voxel_size = fl->voxel_size, distance_threshold = fl->distance_threshold;
    const std::string dir = argv[1];
    const int nf = std::atoi(argv[2]);
    pcl::PointCloud<pcl::PointXYZ>::Ptr template_cuboid(new pcl::PointCloud<pcl::PointXYZ>);
    if (pcl::io::loadPCDFile<pcl::PointXYZ>(dir + "/template.pcd", *template_cuboid) == -1) { std::fprintf(stderr, "no template\n"); return 2; }   // opd.cpp:398
    FILE* out = std::fopen(argv[3], "w");
    if (!out) return 2;
    std::fprintf(out, "{\n \"flavour\": \"%s\",\n \"params\": \"%s\",\n"
                      " \"made_by\": \"tools/pcl_golden.cpp with PCL %d.%d.%d\",\n \"frames\": [\n", fl->name, fl->params, PCL_MAJOR_VERSION, PCL_MINOR_VERSION,
                 PCL_REVISION_VERSION);
    for (int fi = 0; fi < nf; ++fi) {
        // the frame as the D435 driver's PointCloud2 carries it: x y z rgb, float32
        pcl::PointCloud<pcl::PointXYZRGB>::Ptr frame(new pcl::PointCloud<pcl::PointXYZRGB>);
        Sha256 frame_sha;
        {
            FILE* f = std::fopen((dir + "/frame_" + std::to_string(fi) + ".bin").c_str(), "rb");
            if (!f) { std::fprintf(stderr, "no frame %d\n", fi); return 2; }
            float rec[4];
            while (std::fread(rec, 4, 4, f) == 4) {
                frame_sha.add(rec, 16);
                pcl::PointXYZRGB p;
                p.x = rec[0]; p.y = rec[1]; p.z = rec[2]; p.rgb = rec[3];
                frame->points.push_back(p);
            }
            std::fclose(f);
            frame->width = (uint32_t)frame->points.size(); frame->height = 1; frame->is_dense = false;
        }
        pcl::PCLPointCloud2* cloud = new pcl::PCLPointCloud2;
        pcl::PCLPointCloud2ConstPtr cloudPtr(cloud);
        pcl::toPCLPointCloud2(*frame, *cloud);
        // opd.cpp:273-289  PassThrough z [0, 0.9], then x [-0.2, 0.2]
        pcl::PCLPointCloud2* cz = new pcl::PCLPointCloud2; pcl::PCLPointCloud2ConstPtr czPtr(cz);
        pcl::PassThrough<pcl::PCLPointCloud2> pass;
        pass.setInputCloud(cloudPtr); pass.setFilterFieldName("z"); pass.setFilterLimits(0.0, 0.9); pass.filter(*cz);
        pcl::PCLPointCloud2* cx = new pcl::PCLPointCloud2; pcl::PCLPointCloud2ConstPtr cxPtr(cx);
        pass.setInputCloud(czPtr); pass.setFilterFieldName("x"); pass.setFilterLimits(-0.2, 0.2); pass.filter(*cx);
        const int n_cropped = (int)(cx->width * cx->height);
        // opd.cpp:293-298  VoxelGrid
        pcl::PCLPointCloud2* vx = new pcl::PCLPointCloud2; pcl::PCLPointCloud2ConstPtr vxPtr(vx);
        pcl::VoxelGrid<pcl::PCLPointCloud2> sor;
        sor.setInputCloud(cxPtr); sor.setLeafSize((float)voxel_size, (float)voxel_size, (float)voxel_size); sor.filter(*vx);
        pcl::PointCloud<pcl::PointXYZ>::Ptr voxel_ptr(new pcl::PointCloud<pcl::PointXYZ>);
        pcl::fromPCLPointCloud2(*vx, *voxel_ptr);
        // opd.cpp:300-318  SACSegmentation PLANE / RANSAC / optimize / 1000
        pcl::ModelCoefficients::Ptr coefficients(new pcl::ModelCoefficients);
        pcl::PointIndices::Ptr inliers(new pcl::PointIndices);
        pcl::SACSegmentation<pcl::PointXYZ> seg;
        seg.setOptimizeCoefficients(true); seg.setModelType(pcl::SACMODEL_PLANE); seg.setMethodType(pcl::SAC_RANSAC);
        seg.setMaxIterations(1000); seg.setDistanceThreshold(distance_threshold); seg.setInputCloud(voxel_ptr);
        seg.segment(*inliers, *coefficients);
        // opd.cpp:320-326  ExtractIndices negative = invert
        pcl::PCLPointCloud2* np_ = new pcl::PCLPointCloud2; pcl::PCLPointCloud2ConstPtr npPtr(np_);
        pcl::ExtractIndices<pcl::PCLPointCloud2> extract;
        extract.setInputCloud(vxPtr); extract.setIndices(inliers); extract.setNegative(invert); extract.filter(*np_);
        pcl::PointCloud<pcl::PointXYZ>::Ptr objects(new pcl::PointCloud<pcl::PointXYZ>);
        if (fl->second_crop) {   // opd.cpp:331-336  second PassThrough z [0, 0.75]
            pcl::PCLPointCloud2* cl = new pcl::PCLPointCloud2;
            pass.setInputCloud(npPtr); pass.setFilterFieldName("z"); pass.setFilterLimits(0.0, 0.75); pass.filter(*cl);
            pcl::fromPCLPointCloud2(*cl, *objects);
            delete cl;
        } else {                 // gps.cpp:104-112 publishes the extracted cloud as it is; icp.cpp:156 reads it back
            pcl::fromPCLPointCloud2(*np_, *objects);
        }
        std::vector<pcl::PointIndices> clusters;
        if (fl->clusters) {      // opd.cpp:345-362  KdTree + EuclideanClusterExtraction 0.02 / 200 / 25000
            pcl::search::KdTree<pcl::PointXYZ>::Ptr tree(new pcl::search::KdTree<pcl::PointXYZ>);
            tree->setInputCloud(objects);
            pcl::EuclideanClusterExtraction<pcl::PointXYZ> ec;
            ec.setClusterTolerance(0.02); ec.setMinClusterSize(200); ec.setMaxClusterSize(25000); ec.setSearchMethod(tree); ec.setInputCloud(objects);
            ec.extract(clusters);
        } else if (!objects->points.empty()) {   // icp.cpp:171: the whole cloud is the source (label 0 for every point)
            pcl::PointIndices all;
            for (size_t i = 0; i < objects->points.size(); ++i) all.indices.push_back((int)i);
            clusters.push_back(all);
        }
        // canonical labels: rank by (size descending, first member index ascending); PCL's own order is by size
        // with ties unspecified
        std::vector<int> order(clusters.size());
        for (size_t k = 0; k < order.size(); ++k) order[k] = (int)k;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
            if (clusters[a].indices.size() != clusters[b].indices.size()) return clusters[a].indices.size() > clusters[b].indices.size();
            return *std::min_element(clusters[a].indices.begin(), clusters[a].indices.end()) < *std::min_element(clusters[b].indices.begin(), clusters[b].indices.end());
        });
        std::vector<int32_t> labels(objects->size(), fl->clusters ? -1 : 0), inl(inliers->indices.begin(), inliers->indices.end());
        for (size_t r = 0; r < order.size(); ++r) for (int i : clusters[order[r]].indices) labels[(size_t)i] = (int32_t)r;
        std::vector<float> vox;
        for (const auto& p : voxel_ptr->points) { vox.push_back(p.x); vox.push_back(p.y); vox.push_back(p.z); }
        std::fprintf(out, "  {\"index\": %d, \"frame_sha256\": \"%s\", \"n_cropped\": %d, \"n_voxels\": %d, \"n_plane\": %d, \"n_objects\": %d, \"n_clusters\": %d,\n",
                     fi, frame_sha.hex().c_str(), n_cropped, (int)voxel_ptr->size(), (int)inliers->indices.size(), (int)objects->size(), (int)clusters.size());
        std::fprintf(out, "   \"plane_hex\": [");
        for (size_t i = 0; i < coefficients->values.size(); ++i) std::fprintf(out, "%s\"%s\"", i ? ", " : "", hexf(coefficients->values[i]).c_str());
        std::fprintf(out, "],\n   \"voxels_xyz_sha256\": \"%s\", \"plane_inliers_sha256\": \"%s\", \"labels_sha256\": \"%s\",\n   \"clusters\": [",
                     sha_of(vox).c_str(), sha_of(inl).c_str(), sha_of(labels).c_str());
        for (size_t r = 0; r < order.size(); ++r) {   // opd.cpp:376-413: every cluster against the template (one ICP: the <= 11 retries of :215-246 recompute the same answer)
            pcl::PointCloud<pcl::PointXYZ>::Ptr cluster(new pcl::PointCloud<pcl::PointXYZ>);
            std::vector<int> idx = clusters[order[r]].indices;
            std::sort(idx.begin(), idx.end());
            for (int i : idx) cluster->points.push_back(objects->points[(size_t)i]);
            cluster->width = (uint32_t)cluster->points.size(); cluster->height = 1;
            Icp icp;                                                            // opd.cpp:220-228
            icp.setInputSource(cluster); icp.setInputTarget(template_cuboid);
            icp.setMaximumIterations(5000); icp.setTransformationEpsilon(1e-9); icp.setEuclideanFitnessEpsilon(icp_fitness_score);
            icp.setRANSACOutlierRejectionThreshold(1.5);
            pcl::PointCloud<pcl::PointXYZ> aligned;
            icp.align(aligned);
            const Eigen::Matrix4f T = icp.getFinalTransformation();
            const Eigen::Matrix4d pose = T.cast<double>().inverse();            // opd.cpp:229
            const double fitness = icp.getFitnessScore();
            std::fprintf(out, "%s\n    {\"size\": %d, \"iterations\": %d, \"converged\": %d, \"accepted\": %d, \"fitness_hex\": \"%s\",\n     \"T_hex\": [",
                         r ? "," : "", (int)cluster->size(), icp.iterations(), icp.hasConverged() ? 1 : 0, (icp.hasConverged() && fitness < icp_fitness_score) ? 1 : 0,
                         hexf(fitness).c_str());
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) std::fprintf(out, "%s\"%s\"", (i | j) ? ", " : "", hexf(T(i, j)).c_str());
            std::fprintf(out, "],\n     \"pose\": [");
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) std::fprintf(out, "%s%.17g", (i | j) ? ", " : "", pose(i, j));
            std::fprintf(out, "]}");
        }
        std::fprintf(out, "]}%s\n", fi + 1 < nf ? "," : "");
    }
    std::fprintf(out, " ]\n}\n");
    std::fclose(out);
    return 0;
}
