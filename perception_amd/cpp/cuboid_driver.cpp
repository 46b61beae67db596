// cuboid_driver.cpp - ROS-free driver that runs the BODY of the reference's callbacks with the
// pclhip:: classes (pcl_compat.hpp) exactly where the reference uses pcl:: ones.
//   mode gps : cuboid_detection/src/ground_plane_segmentation.cpp:43-113 followed by
//              cuboid_detection/src/iterative_closest_point.cpp:136-203 (ICP on the whole
//              non-plane cloud, acceptance converged && fitness < icp_fitness_score)
//   mode opd : object_detection/src/object_pose_detection.cpp:270-442 (second z crop, clusters,
//              per-cluster ICP, selection by |N_s - M| < 250)
// Input: a raw little-endian float32 file of N records x,y,z,rgb (what the synthetic generator
// writes) and a template .pcd.  Output: one line per result, floats printed as %a (exact).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "pcl_compat.hpp"

using namespace pclhip;

static void print16(const char* tag, const float* T) {
    std::printf("%s", tag);
    for (int i = 0; i < 16; ++i) std::printf(" %a", (double)T[i]);
    std::printf("\n");
}

int main(int argc, char** argv) {
    std::string frame_path, tpl_path, mode = "opd";
    double voxel_size = 0.005, distance_threshold = 0.015, icp_fitness_score = 0.0004;   // launch values
    bool invert = true, unfused = false;   // --unfused 1: look at the cropped cloud, so that the three filters run one by one
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string k = argv[i], v = argv[i + 1];
        if (k == "--frame") frame_path = v;
        else if (k == "--template") tpl_path = v;
        else if (k == "--mode") mode = v;
        else if (k == "--voxel_size") voxel_size = std::atof(v.c_str());
        else if (k == "--distance_threshold") distance_threshold = std::atof(v.c_str());
        else if (k == "--icp_fitness_score") icp_fitness_score = std::atof(v.c_str());
        else if (k == "--unfused") unfused = std::atoi(v.c_str()) != 0;
    }
    if (frame_path.empty() || tpl_path.empty()) { std::fprintf(stderr, "usage: cuboid_driver --frame f.bin --template t.pcd [--mode gps|opd]\n"); return 2; }
    // "message" -> cloud (pcl_conversions::toPCL / fromROSMsg)
    PointCloud<PointXYZRGB>::Ptr cloud(new PointCloud<PointXYZRGB>);
    {
        FILE* f = std::fopen(frame_path.c_str(), "rb");
        if (!f) { std::perror("frame"); return 2; }
        float rec[4];
        while (std::fread(rec, 4, 4, f) == 4) { PointXYZRGB p; p.x = rec[0]; p.y = rec[1]; p.z = rec[2]; p.rgb = rec[3]; cloud->points.push_back(p); }
        std::fclose(f);
        cloud->width = (uint32_t)cloud->points.size();
    }
    try {
        Device::instance((int)cloud->points.size(), 1, 0);
    } catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); return 3; }

    // gps.cpp:53-58  filter the points in z
    PointCloud<PointXYZRGB>::Ptr cloud_filtered_ptr_z(new PointCloud<PointXYZRGB>);
    PointCloud<PointXYZRGB>::Ptr cloud_filtered_ptr(new PointCloud<PointXYZRGB>);
    PassThrough<PointXYZRGB> pass_z;
    pass_z.setInputCloud(cloud);
    pass_z.setFilterFieldName("z");
    pass_z.setFilterLimits(0.0, 0.9);
    pass_z.filter(*cloud_filtered_ptr_z);
    // gps.cpp:61-65  filter the points in x
    PassThrough<PointXYZRGB> pass;
    pass.setInputCloud(cloud_filtered_ptr_z);
    pass.setFilterFieldName("x");
    pass.setFilterLimits(-0.2, 0.2);
    pass.filter(*cloud_filtered_ptr);
    if (unfused) std::printf("cropped %zu\n", cloud_filtered_ptr->size());   // (looking at the cloud runs the two filters one by one)
    // gps.cpp:69-73  downsample
    PointCloud<PointXYZRGB>::Ptr voxel_ptr(new PointCloud<PointXYZRGB>);
    VoxelGrid<PointXYZRGB> downsample;
    downsample.setInputCloud(cloud_filtered_ptr);
    downsample.setLeafSize((float)voxel_size, (float)voxel_size, (float)voxel_size);
    if (!downsample.filter(*voxel_ptr)) return 4;
    std::printf("voxels %zu\n", voxel_ptr->size());
    std::fprintf(stderr, "crops fused into the voxel call: %d\n", downsample.lastFilterWasFused() ? 1 : 0);

    // gps.cpp:76-93  plane segmentation
    ModelCoefficients::Ptr coefficients(new ModelCoefficients);
    PointIndices::Ptr inliers(new PointIndices);
    SACSegmentation<PointXYZRGB> seg;
    seg.setOptimizeCoefficients(true);
    seg.setModelType(SACMODEL_PLANE);
    seg.setMethodType(SAC_RANSAC);
    seg.setMaxIterations(1000);
    seg.setDistanceThreshold(distance_threshold);
    seg.setInputCloud(voxel_ptr);
    seg.segment(*inliers, *coefficients);
    std::printf("plane_inliers %zu\n", inliers->indices.size());
    if (coefficients->values.size() == 4)
        std::printf("coefficients %a %a %a %a\n", (double)coefficients->values[0], (double)coefficients->values[1],
                    (double)coefficients->values[2], (double)coefficients->values[3]);

    // gps.cpp:96-101  extract the non-plane points
    PointCloud<PointXYZRGB>::Ptr plane_cloud_ptr(new PointCloud<PointXYZRGB>);
    ExtractIndices<PointXYZRGB> extract;
    extract.setInputCloud(voxel_ptr);
    extract.setIndices(inliers);
    extract.setNegative(invert);
    extract.filter(*plane_cloud_ptr);
    std::printf("objects %zu\n", plane_cloud_ptr->size());

    // icp.cpp:159  template
    PointCloud<PointXYZ>::Ptr template_cuboid(new PointCloud<PointXYZ>);
    if (io::loadPCDFile(tpl_path, *template_cuboid) == -1) { std::fprintf(stderr, "Couldn't read the template PCL file\n"); return 5; }

    auto run_icp = [&](const PointCloud<PointXYZRGB>::Ptr& src, const char* tag) {
        IterativeClosestPoint<PointXYZRGB, PointXYZ> icp;     // icp.cpp:170-178
        icp.setInputSource(src);
        icp.setInputTarget(template_cuboid);
        icp.setMaximumIterations(5000);
        icp.setTransformationEpsilon(1e-9);
        icp.setEuclideanFitnessEpsilon(icp_fitness_score);
        icp.setRANSACOutlierRejectionThreshold(1.5);
        PointCloud<PointXYZRGB> output_cloud;
        icp.align(output_cloud);
        const auto T = icp.getFinalTransformation();
        const bool ok = icp.hasConverged() && icp.getFitnessScore() < icp_fitness_score;   // icp.cpp:182
        std::printf("%s size %zu iterations %d converged %d accepted %d fitness %a\n", tag, src->size(), icp.getIterations(),
                    icp.hasConverged() ? 1 : 0, ok ? 1 : 0, icp.getFitnessScore());
        print16("T", T.data());
        if (ok) {   // publish_pose / publish_bounding_box (icp.cpp:55-128)
            const auto H = icp.getFinalTransformationInverse();
            double pos[3], q[4];
            float box[24];
            cd_pose_to_position_quaternion(H.data(), pos, q);
            cd_bbox_corners(H.data(), 0.2, 0.1, 0.03, box);
            std::printf("pose %a %a %a quat %a %a %a %a\n", pos[0], pos[1], pos[2], q[0], q[1], q[2], q[3]);
        }
        return (long)src->size();
    };

    if (mode == "gps") {
        run_icp(plane_cloud_ptr, "icp");
        return 0;
    }
    // opd.cpp:331-336  second z crop (done by index bookkeeping on the extracted cloud)
    PointCloud<PointXYZRGB>::Ptr cleaned(new PointCloud<PointXYZRGB>);
    for (const auto& p : plane_cloud_ptr->points)
        if (!((double)p.z > 0.75 || (double)p.z < 0.0)) cleaned->points.push_back(p);
    cleaned->width = (uint32_t)cleaned->points.size();
    // opd.cpp:345-362  clusters
    std::vector<PointIndices> object_cluster_indices;
    EuclideanClusterExtraction<PointXYZRGB> ec;
    ec.setClusterTolerance(0.02);
    ec.setMinClusterSize(200);
    ec.setMaxClusterSize(25000);
    ec.setInputCloud(cleaned);
    ec.extract(object_cluster_indices);
    std::printf("clusters %zu\n", object_cluster_indices.size());
    long best_diff = 1000;
    int argmin = -1, k = 0;
    for (const auto& ci : object_cluster_indices) {   // opd.cpp:376-413
        PointCloud<PointXYZRGB>::Ptr object_cluster(new PointCloud<PointXYZRGB>);
        for (int i : ci.indices) object_cluster->points.push_back(cleaned->points[(size_t)i]);
        object_cluster->width = (uint32_t)object_cluster->points.size();
        const long ns = run_icp(object_cluster, "cluster");
        const long diff = std::labs(ns - (long)template_cuboid->size());
        if (diff < best_diff) { best_diff = diff; argmin = k; }
        ++k;
    }
    std::printf("argmin %d diff %ld success %d\n", argmin, best_diff, best_diff < 250 ? 1 : 0);   // opd.cpp:416-441
    return 0;
}
